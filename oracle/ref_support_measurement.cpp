// ORACLE — TEST INFRASTRUCTURE.  Reference-derived checker (oracle/_ref/support_measurement).
//
// Links the reference's OWN InlierSupportMeasurer, compiled from where it lies
// (/root/reference/src/optim/support_measurement.cc + .h: std-only, built with -I/root/reference/src
// -ffp-contract=off, never copied), and prints what it returns for residual vectors read from a file.
// Pins SURVEY.md §8 row a-10 (count, sequential residual sum, tie-break) to the reference itself.
//
// input (text, every double as a C99 hex float):
//   T <k> <thr> ...             the thresholds
//   V <n> <r_0> ... <r_{n-1}>   one residual vector per line (n may be 0)
// output:
//   E <vector> <threshold> <num_inliers> <residual_sum %a>     InlierSupportMeasurer::Evaluate
//   C <threshold> <row of Compare(support_i, support_j) for all j>   one line per i
//   W <threshold> <winner>      the sequential accept rule of optim/ransac.h:232-236 over the vectors in file
//                               order: a candidate replaces the best one iff Compare(candidate, best), starting
//                               from the default-constructed Support (support_measurement.h:44-50); -1 = none
#include "optim/support_measurement.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s input.txt\n", argv[0]); return 2; }
  std::FILE* f = std::fopen(argv[1], "r");
  if (!f) { std::perror(argv[1]); return 2; }
  std::vector<double> thresholds;
  std::vector<std::vector<double>> vectors;
  char tag[8];
  while (std::fscanf(f, "%7s", tag) == 1) {
    long n = 0;
    if (std::fscanf(f, "%ld", &n) != 1) return 3;
    std::vector<double> v(static_cast<size_t>(n));
    for (long i = 0; i < n; ++i) {
      char tok[64];
      if (std::fscanf(f, "%63s", tok) != 1) return 3;
      v[static_cast<size_t>(i)] = std::strtod(tok, nullptr);
    }
    if (tag[0] == 'T') thresholds = v; else vectors.push_back(v);
  }
  std::fclose(f);
  colmap::InlierSupportMeasurer measurer;
  for (size_t t = 0; t < thresholds.size(); ++t) {
    std::vector<colmap::InlierSupportMeasurer::Support> s(vectors.size());
    for (size_t i = 0; i < vectors.size(); ++i) {
      s[i] = measurer.Evaluate(vectors[i], thresholds[t]);
      std::printf("E %zu %zu %zu %a\n", i, t, s[i].num_inliers, s[i].residual_sum);
    }
    for (size_t i = 0; i < vectors.size(); ++i) {
      std::printf("C %zu", t);
      for (size_t j = 0; j < vectors.size(); ++j) std::printf(" %d", measurer.Compare(s[i], s[j]) ? 1 : 0);
      std::printf("\n");
    }
    colmap::InlierSupportMeasurer::Support best;  // num_inliers 0, residual_sum DBL_MAX
    long winner = -1;
    for (size_t i = 0; i < vectors.size(); ++i)
      if (measurer.Compare(s[i], best)) { best = s[i]; winner = static_cast<long>(i); }
    std::printf("W %zu %ld\n", t, winner);
  }
  return 0;
}
