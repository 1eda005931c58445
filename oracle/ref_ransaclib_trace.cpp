// ORACLE — TEST INFRASTRUCTURE.  Reference-derived checker (oracle/_ref/ransaclib_trace).
//
// Compiles the reference's OWN std-only LO-MSAC driver headers from where they lie
// (/root/reference/lib/RansacLib/RansacLib/{ransac,sampling,utils}.h, included with -I, never copied)
// against a toy line-fitting Solver written here, and prints a trace (iterations, inliers, sample
// stream) used to pin our restatement of that control flow.  Built only in the build container.
#include <RansacLib/ransac.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

struct Line2 { double a, b, c; };

class LineSolver {
 public:
  LineSolver(const std::vector<double>& x, const std::vector<double>& y) : x_(x), y_(y) {}
  int min_sample_size() const { return 2; }
  int non_minimal_sample_size() const { return 6; }
  int num_data() const { return static_cast<int>(x_.size()); }
  int MinimalSolver(const std::vector<int>& s, std::vector<Line2>* models) const {
    models->clear();
    const double dx = x_[s[1]] - x_[s[0]], dy = y_[s[1]] - y_[s[0]];
    const double n = std::sqrt(dx * dx + dy * dy);
    if (n < 1e-12) return 0;
    Line2 l{-dy / n, dx / n, 0};
    l.c = -(l.a * x_[s[0]] + l.b * y_[s[0]]);
    models->push_back(l);
    return 1;
  }
  int NonMinimalSolver(const std::vector<int>& s, Line2* m) const {
    double mx = 0, my = 0;
    for (int i : s) { mx += x_[i]; my += y_[i]; }
    mx /= s.size(); my /= s.size();
    double sxx = 0, sxy = 0, syy = 0;
    for (int i : s) { sxx += (x_[i] - mx) * (x_[i] - mx); sxy += (x_[i] - mx) * (y_[i] - my); syy += (y_[i] - my) * (y_[i] - my); }
    const double th = 0.5 * std::atan2(2 * sxy, sxx - syy);
    m->a = -std::sin(th); m->b = std::cos(th); m->c = -(m->a * mx + m->b * my);
    return 1;
  }
  double EvaluateModelOnPoint(const Line2& m, int i) const { const double d = m.a * x_[i] + m.b * y_[i] + m.c; return d * d; }
  void LeastSquares(const std::vector<int>& s, Line2* m) const { NonMinimalSolver(s, m); }

 private:
  std::vector<double> x_, y_;
};

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 200;
  std::mt19937 g(7);
  std::uniform_real_distribution<double> u(-1, 1);
  std::normal_distribution<double> nz(0, 0.01);
  std::vector<double> x(n), y(n);
  for (int i = 0; i < n; ++i) {
    x[i] = u(g);
    y[i] = (i % 3 == 0) ? u(g) : 0.5 * x[i] + 0.1 + nz(g);
  }
  // the sampler stream of seed 0 (what our restatement must reproduce)
  ransac_lib::UniformSampling sampler(0, n, 2);
  std::printf("samples");
  for (int t = 0; t < 16; ++t) { std::vector<int> s; sampler.Sample(&s); std::printf(" %d %d", s[0], s[1]); }
  std::printf("\n");
  ransac_lib::LORansacOptions opt;
  opt.min_num_iterations_ = 100; opt.max_num_iterations_ = 1000; opt.squared_inlier_threshold_ = 0.03 * 0.03; opt.random_seed_ = 0;
  LineSolver solver(x, y);
  ransac_lib::LocallyOptimizedMSAC<Line2, std::vector<Line2>, LineSolver> lomsac;
  ransac_lib::RansacStatistics st;
  Line2 best{0, 0, 0};
  const int ninl = lomsac.EstimateModel(opt, solver, &best, &st);
  std::printf("inliers %d iterations %d lo %d score %.17g ratio %.17g\n", ninl, st.num_iterations, st.number_lo_iterations,
              st.best_model_score, st.inlier_ratio);
  std::printf("model %.17g %.17g %.17g\n", best.a, best.b, best.c);
  for (double eps : {0.1, 0.25, 0.5, 0.9})
    std::printf("numiter %.2f %u\n", eps, ransac_lib::utils::NumRequiredIterations(eps, 0.0001, 5, 100, 10000));
  return 0;
}
