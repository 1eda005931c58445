// The image order pp_ba_create gives the reduced camera system - host code only (no device is touched).
//
// Ceres' SPARSE_SCHUR reorders the cameras before it factorises the reduced system (the solver the reference selects for 50 < images <= 1000,
// src/optim/bundle_adjustment.cc:279-282).  Here the reduced system is one dense array whose empty 64x64 tiles are skipped and whose independent
// sub-trees of the elimination tree are factorised side by side by several chain workgroups (cholesky.hip "ChainRanges"), so what an ordering has to
// deliver is (a) few non-zero tiles and (b) a short longest dependency path over the block columns:
//   * reverse Cuthill-McKee on the co-visibility graph: a band (a sequence whose image ids are not in capture order gets its block-banded system back);
//   * nested dissection of that band (DissectBand): [part | part | the images that couple them], cuts at tile boundaries;
//   * nested dissection of the GRAPH itself (DissectGraph, round 5): a vertex separator from a level structure, for co-visibility that no order makes a
//     narrow band - photo collections: clusters joined by a few images, a hub with satellites -, the components a separator leaves as parts of their own.
// The candidates are compared by the number of chain steps of their one-launch factorisation (CholeskyPlanSteps: the plan alone, no task list), and
// the winner's task list is verified once (CholeskyChainSteps: plan + list + replay) before it is taken.
//
// Cost (round 5; the mapper builds a new BundleAdjuster per global BA, src/sfm/incremental_mapper.cc:893-936, so this runs per call): the co-visibility is a
// bit matrix filled point by point, and the fill STOPS as soon as the graph is too dense for any order to remove a tenth of the tiles (a clique - the
// headline scene - leaves after ~0.3 ms instead of the 23 ms round 4's candidate plans cost it).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "ba_impl.hpp"
#include "camera_models.hpp"

namespace ppsfm {

namespace {

using Adjacency = std::vector<std::vector<int32_t>>;

// Reverse Cuthill-McKee order of the images on their co-visibility graph (adj: neighbours of every image, no self loops).  Per connected component: a
// pseudo-peripheral start (repeated breadth-first searches from a minimum-degree node of the last level), breadth-first numbering with the neighbours by
// increasing degree, the whole order reversed.  Images without neighbours (constant poses, unobserved images) keep their relative order at the end.
// Returns old_of_new.
// (G: the graph as begin(u) / end(u) / degree(u) - the whole graph's neighbour lists, or a part's sub-graph in two flat arrays)
struct ListsGraph {
  const Adjacency& adj;
  int size() const { return (int)adj.size(); }
  const int32_t* begin(int u) const { return adj[u].data(); }
  const int32_t* end(int u) const { return adj[u].data() + adj[u].size(); }
  size_t degree(int u) const { return adj[u].size(); }
};
struct FlatGraph {
  std::vector<int32_t> off, nb;
  int size() const { return (int)off.size() - 1; }
  const int32_t* begin(int u) const { return nb.data() + off[u]; }
  const int32_t* end(int u) const { return nb.data() + off[u + 1]; }
  size_t degree(int u) const { return (size_t)(off[u + 1] - off[u]); }
};
template <class G>
std::vector<int32_t> ReverseCuthillMcKeeOf(const G& g) {
  const int C = g.size();
  std::vector<int32_t> order; order.reserve(C);
  std::vector<int32_t> level(C, -1);
  std::vector<char> placed(C, 0);
  auto bfs = [&](int start, std::vector<int32_t>* visit) {      // levels from `start` over unplaced nodes; returns the last node of minimum degree in the deepest level
    visit->clear();
    visit->push_back(start); level[start] = 0;
    for (size_t q = 0; q < visit->size(); ++q) {
      const int u = (*visit)[q];
      for (const int32_t* p = g.begin(u); p != g.end(u); ++p) { const int v = *p; if (!placed[v] && level[v] < 0) { level[v] = level[u] + 1; visit->push_back(v); } }
    }
    const int depth = level[visit->back()];
    int best = visit->back();
    for (int v : *visit) if (level[v] == depth && g.degree(v) < g.degree(best)) best = v;
    for (int v : *visit) level[v] = -1;
    return std::make_pair(best, depth);
  };
  std::vector<int32_t> by_degree(C);
  std::iota(by_degree.begin(), by_degree.end(), 0);
  std::stable_sort(by_degree.begin(), by_degree.end(), [&](int a, int b) { return g.degree(a) < g.degree(b); });
  std::vector<int32_t> visit, nb;
  for (int seed : by_degree) {
    if (placed[seed] || g.degree(seed) == 0) continue;
    int start = seed, ecc = -1;
    for (int round = 0; round < 4; ++round) {      // pseudo-peripheral node (two or three searches settle it on these graphs)
      const auto far = bfs(start, &visit);
      if (far.second <= ecc) break;
      ecc = far.second; start = far.first;
    }
    const size_t first = order.size();
    order.push_back(start); placed[start] = 1;
    for (size_t q = first; q < order.size(); ++q) {
      const int u = order[q];
      nb.clear();
      for (const int32_t* p = g.begin(u); p != g.end(u); ++p) { const int v = *p; if (!placed[v]) { placed[v] = 1; nb.push_back(v); } }
      std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return g.degree(a) < g.degree(b); });
      order.insert(order.end(), nb.begin(), nb.end());
    }
  }
  std::reverse(order.begin(), order.end());
  for (int c = 0; c < C; ++c) if (!placed[c]) order.push_back(c);
  return order;
}
std::vector<int32_t> ReverseCuthillMcKee(const Adjacency& adj) { return ReverseCuthillMcKeeOf(ListsGraph{adj}); }

// a vertex set in a band order of its own (Cuthill-McKee on the sub-graph of the set: its images with neighbours inside the set first, the others behind
// them).  The sub-graph is built with LOCAL indices in two flat arrays (the dissections call this for every part they look at: no per-call array of the whole
// graph's size, no vector per vertex); `loc` is scratch, -1 outside every call.
std::vector<int32_t> BandOrderOf(const std::vector<int32_t>& set, const Adjacency& adj, std::vector<int32_t>* loc) {
  const int n = (int)set.size();
  for (int i = 0; i < n; ++i) (*loc)[set[i]] = i;
  FlatGraph sub;
  sub.off.assign((size_t)n + 1, 0);
  size_t bound = 0;
  for (int i = 0; i < n; ++i) bound += adj[set[i]].size();
  sub.nb.resize(bound);
  size_t at = 0;
  for (int i = 0; i < n; ++i) {
    for (int u : adj[set[i]]) if ((*loc)[u] >= 0) sub.nb[at++] = (*loc)[u];
    sub.off[i + 1] = (int32_t)at;
  }
  for (int i = 0; i < n; ++i) (*loc)[set[i]] = -1;
  const std::vector<int32_t> order = ReverseCuthillMcKeeOf(sub);      // (vertices without neighbours inside the set come last, in the set's order)
  std::vector<int32_t> out(n);
  for (int i = 0; i < n; ++i) out[i] = set[order[i]];
  return out;
}

// A chain starts at a 64-column tile boundary and needs three block columns.  With w columns per image (6; 6 + n_v when every image carries its own variable
// intrinsics beside its pose columns, see PrivateIntrinsicsColumns) a part may start every 64 / gcd(64, w) images - 32 for w = 6, 8 for w = 8 - and holds at
// least 192 columns.  Set per call by ChooseImageOrdering.
struct Grain {
  int align = 32, min_leaf = 32, min_right = 33;
  explicit Grain(int columns_per_image = 6) {
    int g = 64, b = columns_per_image;
    while (b) { const int t = g % b; g = b; b = t; }      // gcd(64, columns per image)
    align = 64 / g;
    min_leaf = ((192 + columns_per_image - 1) / columns_per_image + align - 1) / align * align;
    min_right = min_leaf + 1;
  }
};

// ---- nested dissection of a BAND order ------------------------------------------------------------------------------------------------------------
// The one-launch factorisation runs a chain workgroup per independent sub-tree of the elimination tree, so a band of T block columns costs T steps of one
// chain, while [left part | right part | the images that couple them] costs max(left, right) + separator steps of two.  A cut position c of the sequence:
// the separator is every image at a position >= c with a neighbour before c, the right part the rest of [c, n); parts are dissected again (`levels`).
// Cuts are multiples of Grain::align images, and a part starts where its parent started plus such a cut.  A cut is taken when it shortens the sequence's chain
// (max(left, right) + separator) to at most 0.8 of its length.  A PART is first put into a band order of its own when that gives the better cut: the parts
// of a ring folded flat are open bands of half its width.
struct BandCut { int c = -1, cost = 0; std::vector<int32_t> first_nb; };
BandCut BestBandCut(const std::vector<int32_t>& seq, const Adjacency& adj, std::vector<int32_t>* pos, int bias, const Grain& gr) {
  const int n = (int)seq.size();
  BandCut cut;
  cut.cost = n;
  for (int i = 0; i < n; ++i) (*pos)[seq[i]] = i;
  cut.first_nb.resize(n);
  for (int i = 0; i < n; ++i) {
    int m = i;
    for (int v : adj[seq[i]]) if ((*pos)[v] >= 0) m = std::min(m, (int)(*pos)[v]);
    cut.first_nb[i] = m;
  }
  for (int i = 0; i < n; ++i) (*pos)[seq[i]] = -1;
  // separator size per cut position by a sweep: image i belongs to the separator of every cut c with first_nb[i] < c <= i
  std::vector<int32_t> diff(n + 2, 0);
  for (int i = 0; i < n; ++i) if (cut.first_nb[i] < i) { diff[cut.first_nb[i] + 1] += 1; diff[i + 1] -= 1; }
  int sep = 0;
  for (int c = 1; c + gr.min_right <= n; ++c) {
    sep += diff[c];
    if (c < gr.min_leaf || c % gr.align) continue;
    const int right = n - c - sep;
    if (right < gr.min_right) continue;
    // (bias: the left part's chain stops and its contributions reach the separator two steps - 21 images - behind its last column, the right part's
    // chain runs on into the separator: an even split makes that chain wait.  Both are tried.)
    const int cost = std::max(c + bias, right) + sep;
    if (cost < cut.cost) { cut.cost = cost; cut.c = c; }
  }
  return cut;
}
// (the eight trials of ChooseImageOrdering - one to four levels, two balances - look at the same parts again and again: a part's band order and best cut are
// kept per (sequence, balance))
struct BandDissector {
  const Adjacency& adj;
  const Grain gr;
  std::vector<int32_t> pos;
  struct Node { std::vector<int32_t> seq; BandCut cut; };      // seq: the order the cut refers to (the part's own band order, or the one it inherited)
  std::vector<std::pair<uint64_t, Node>> memo;
  BandDissector(const Adjacency& a, const Grain& g) : adj(a), gr(g), pos(a.size(), -1) {}
  static uint64_t Key(const std::vector<int32_t>& seq, bool reorder, int bias) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)(reorder ? 7 : 3) ^ ((uint64_t)bias << 32);
    for (int v : seq) { h ^= (uint64_t)(uint32_t)v; h *= 1099511628211ull; }
    return h ^ seq.size();
  }
  const Node& NodeOf(const std::vector<int32_t>& seq_in, bool reorder, int bias) {
    const uint64_t key = Key(seq_in, reorder, bias);
    for (const auto& m : memo) if (m.first == key && m.second.seq.size() == seq_in.size()) return m.second;
    Node nd;
    nd.cut = BestBandCut(seq_in, adj, &pos, bias, gr);
    if (reorder) {
      std::vector<int32_t> own = BandOrderOf(seq_in, adj, &pos);
      BandCut cut2 = BestBandCut(own, adj, &pos, bias, gr);
      if (cut2.c >= 0 && cut2.cost < nd.cut.cost) { nd.cut = std::move(cut2); nd.seq = std::move(own); }
    }
    if (nd.seq.empty()) nd.seq = seq_in;
    memo.emplace_back(key, std::move(nd));
    return memo.back().second;
  }
  std::vector<int32_t> Dissect(const std::vector<int32_t>& seq_in, int levels, bool reorder, int bias) {
    const int n = (int)seq_in.size();
    if (levels <= 0 || n < gr.min_leaf + gr.min_right + 4) return seq_in;
    std::vector<int32_t> left, right, sep;
    {
      const Node& nd = NodeOf(seq_in, reorder, bias);      // (a reference into memo: not held across the recursive calls below, which may grow it)
      if (nd.cut.c < 0 || nd.cut.cost * 5 > n * 4) return seq_in;
      left.assign(nd.seq.begin(), nd.seq.begin() + nd.cut.c);
      for (int i = nd.cut.c; i < n; ++i) (nd.cut.first_nb[i] < nd.cut.c ? sep : right).push_back(nd.seq[i]);
    }
    std::vector<int32_t> out = Dissect(left, levels - 1, true, bias);
    const std::vector<int32_t> r = Dissect(right, levels - 1, true, bias);
    out.insert(out.end(), r.begin(), r.end());
    out.insert(out.end(), sep.begin(), sep.end());
    return out;
  }
};

// ---- nested dissection of the GRAPH -----------------------------------------------------------------------------------------------------------------
// Order of a vertex set: [part | part | ... | separator].  The set's connected components are parts of their own (nothing couples them: no separator);
// a connected set is cut by a vertex separator taken from the level structure of a pseudo-peripheral vertex: level j, reduced to its vertices with a
// neighbour in level j + 1 (the others join the near side); the far side's components become parts.  The level with the smallest
// max(largest part) + separator is taken if that is at most 0.8 of the set.  Every part but the last must be a multiple of Grain::align images long (the next one
// starts at a tile boundary): the remainder moves into the separator - vertices next to the separator first.  Leaves are put into a band order of their own.
struct GraphDissector {
  const Adjacency& adj;
  const Grain gr;
  std::vector<int32_t> tag, level, comp;      // scratch, -1 outside every call
  GraphDissector(const Adjacency& a, const Grain& g) : adj(a), gr(g), tag(a.size(), -1), level(a.size(), -1), comp(a.size(), -1) {}

  // connected components of a set (tag marks membership during the call)
  std::vector<std::vector<int32_t>> Components(const std::vector<int32_t>& set) {
    std::vector<std::vector<int32_t>> out;
    for (int v : set) tag[v] = 0;
    std::vector<int32_t> stack;
    for (int s : set) {
      if (tag[s] != 0) continue;
      out.emplace_back();
      tag[s] = 1; stack.push_back(s);
      while (!stack.empty()) {
        const int u = stack.back(); stack.pop_back();
        out.back().push_back(u);
        for (int v : adj[u]) if (tag[v] == 0) { tag[v] = 1; stack.push_back(v); }
      }
    }
    for (int v : set) tag[v] = -1;
    return out;
  }
  // breadth-first levels of a connected set from `root`; returns the levels
  std::vector<std::vector<int32_t>> Levels(const std::vector<int32_t>& set, int root) {
    for (int v : set) tag[v] = 0;
    std::vector<std::vector<int32_t>> lv(1, std::vector<int32_t>{root});
    tag[root] = 1;
    for (;;) {
      std::vector<int32_t> next;
      for (int u : lv.back()) for (int v : adj[u]) if (tag[v] == 0) { tag[v] = 1; next.push_back(v); }
      if (next.empty()) break;
      lv.push_back(std::move(next));
    }
    for (int v : set) tag[v] = -1;
    return lv;
  }
  std::vector<int32_t> Leaf(const std::vector<int32_t>& set) { return BandOrderOf(set, adj, &tag); }

  struct Split { std::vector<std::vector<int32_t>> parts; std::vector<int32_t> sep; int cost = 0; };
  // the best level-structure separator of a connected set, or parts.empty()
  Split BestSplit(const std::vector<int32_t>& set) {
    Split best;
    const int n = (int)set.size();
    best.cost = n;
    if (n < gr.min_leaf + gr.min_right + 4) return best;
    // pseudo-peripheral root: repeated searches from a minimum-degree vertex of the last level
    int root = set[0];
    for (int v : set) if (adj[v].size() < adj[root].size()) root = v;
    std::vector<std::vector<int32_t>> lv = Levels(set, root);
    for (int round = 0; round < 6; ++round) {
      int far = lv.back()[0];
      for (int v : lv.back()) if (adj[v].size() < adj[far].size()) far = v;
      std::vector<std::vector<int32_t>> lv2 = Levels(set, far);
      if (lv2.size() <= lv.size()) break;
      lv.swap(lv2);
    }
    const int m = (int)lv.size();
    if (m < 3) return best;
    for (size_t j = 0; j < lv.size(); ++j) for (int v : lv[j]) level[v] = (int)j;
    // per vertex: does it touch the next / the previous level (one pass over the edges); per level: how many do
    std::vector<int32_t> below(m + 1, 0), touch_up(m, 0), touch_down(m, 0);
    for (int j = 0; j < m; ++j) below[j + 1] = below[j] + (int)lv[j].size();
    for (int v : set) {
      bool up = false, down = false;
      for (int u : adj[v]) { const int l = level[u]; if (l < 0) continue; up = up || l == level[v] + 1; down = down || l == level[v] - 1; }
      comp[v] = (up ? 1 : 0) | (down ? 2 : 0);
      touch_up[level[v]] += up ? 1 : 0; touch_down[level[v]] += down ? 1 : 0;
    }
    int best_j = -1, best_dir = 0;
    for (int j = 1; j + 1 < m; ++j) {
      // dir 0: the separator is the part of level j that touches level j + 1 (the rest of the level joins the near side); dir 1: the part that touches
      // level j - 1 (the rest joins the far side)
      for (int dir = 0; dir < 2; ++dir) {
        const int sz = (int)lv[j].size(), sep = dir ? touch_down[j] : touch_up[j];
        const int near = below[j] + (dir ? 0 : sz - sep), far = n - below[j + 1] + (dir ? sz - sep : 0);
        if (near < gr.min_leaf || far < gr.min_right) continue;
        const int cost = std::max(near, far) + sep + (near % gr.align);      // (the far side may fall into components: counted whole here, split below)
        if (cost < best.cost) { best.cost = cost; best_j = j; best_dir = dir; }
      }
    }
    if (best_j >= 0) {
      const int j = best_j, dir = best_dir;
      std::vector<int32_t> near, far;
      for (int v : set) {
        const int l = level[v];
        if (l < j) near.push_back(v);
        else if (l > j) far.push_back(v);
        else if (comp[v] & (dir ? 2 : 1)) best.sep.push_back(v);
        else (dir ? far : near).push_back(v);
      }
      best.parts.push_back(std::move(near));
      best.parts.push_back(std::move(far));
    }
    for (int v : set) comp[v] = -1;
    for (int v : set) level[v] = -1;
    return best;
  }

  // parts -> aligned parts: every part but the last gives its remainder (mod Grain::align) to `tail`, vertices adjacent to `tail`/`sep` first; parts that
  // become shorter than a chain join the tail whole
  void Align(std::vector<std::vector<int32_t>>* parts, std::vector<int32_t>* tail) {
    std::vector<std::vector<int32_t>> out;
    for (int v : *tail) tag[v] = 2;
    for (size_t i = 0; i < parts->size(); ++i) {
      std::vector<int32_t>& p = (*parts)[i];
      const bool last = i + 1 == parts->size();
      if ((int)p.size() < (last ? gr.min_right : gr.min_leaf)) { for (int v : p) { tail->push_back(v); tag[v] = 2; } continue; }
      const int rem = last ? 0 : (int)p.size() % gr.align;
      if (rem) {
        // boundary vertices (a neighbour in the tail) first
        std::stable_partition(p.begin(), p.end(), [&](int v) { for (int u : adj[v]) if (tag[u] == 2) return false; return true; });
        for (int r = 0; r < rem; ++r) { const int v = p.back(); p.pop_back(); tail->push_back(v); tag[v] = 2; }
      }
      out.push_back(std::move(p));
    }
    for (int v : *tail) tag[v] = -1;
    parts->swap(out);
  }

  // (ChooseImageOrdering asks for one to four levels: the deeper calls meet the shallower ones' parts again - a set's cut and its leaf order are kept)
  struct Cut { std::vector<std::vector<int32_t>> parts; std::vector<int32_t> tail; };      // parts empty: the set takes no cut
  std::vector<std::pair<uint64_t, Cut>> cuts;
  std::vector<std::pair<uint64_t, std::vector<int32_t>>> leaves;
  static uint64_t Key(const std::vector<int32_t>& set) {
    uint64_t h = 1469598103934665603ull;
    for (int v : set) { h ^= (uint64_t)(uint32_t)v; h *= 1099511628211ull; }
    return h ^ set.size();
  }
  const std::vector<int32_t>& LeafOf(const std::vector<int32_t>& set) {
    const uint64_t key = Key(set);
    for (const auto& m : leaves) if (m.first == key && m.second.size() == set.size()) return m.second;
    leaves.emplace_back(key, Leaf(set));
    return leaves.back().second;
  }
  size_t CutOf(const std::vector<int32_t>& set) {      // index into cuts (stable across later insertions)
    const uint64_t key = Key(set);
    for (size_t i = 0; i < cuts.size(); ++i) if (cuts[i].first == key) return i;
    Cut c;
    const int n = (int)set.size();
    std::vector<std::vector<int32_t>> parts = Components(set);
    std::vector<int32_t> tail;
    bool cut_found = parts.size() > 1;
    if (!cut_found) {
      Split sp = BestSplit(set);
      if (!sp.parts.empty() && sp.cost * 5 <= n * 4) {
        cut_found = true;
        tail = std::move(sp.sep);
        std::vector<std::vector<int32_t>> far = Components(sp.parts[1]);      // the far side's components are parts of their own
        parts.clear();
        parts.push_back(std::move(sp.parts[0]));
        for (auto& f : far) parts.push_back(std::move(f));
      }
    }
    if (cut_found) {
      // large parts first (their chains are the long ones), the largest LAST: it needs no alignment and its chain runs on into the tail
      std::stable_sort(parts.begin(), parts.end(), [](const std::vector<int32_t>& a, const std::vector<int32_t>& b) { return a.size() > b.size(); });
      if (parts.size() > 1) std::rotate(parts.begin(), parts.begin() + 1, parts.end());
      Align(&parts, &tail);
      if (parts.size() > 1 || (parts.size() == 1 && !tail.empty())) { c.parts = std::move(parts); c.tail = std::move(tail); }
    }
    cuts.emplace_back(key, std::move(c));
    return cuts.size() - 1;
  }
  std::vector<int32_t> Dissect(const std::vector<int32_t>& set, int levels) {
    const int n = (int)set.size();
    if (levels <= 0 || n < gr.min_leaf + gr.min_right + 4) return LeafOf(set);
    const size_t ci = CutOf(set);
    if (cuts[ci].second.parts.empty()) return LeafOf(set);
    std::vector<int32_t> out; out.reserve(n);
    const size_t np = cuts[ci].second.parts.size();
    for (size_t i = 0; i < np; ++i) {
      const std::vector<int32_t> part = cuts[ci].second.parts[i];      // (a copy: the recursion may grow `cuts`)
      const std::vector<int32_t> o = Dissect(part, levels - 1);
      out.insert(out.end(), o.begin(), o.end());
    }
    if (!cuts[ci].second.tail.empty()) { const std::vector<int32_t> tail = cuts[ci].second.tail; const std::vector<int32_t>& t = LeafOf(tail); out.insert(out.end(), t.begin(), t.end()); }
    return out;
  }
};

}  // namespace

// Per-image intrinsics beside their pose columns (round 5).  BundleAdjuster::ParameterizeCameras (src/optim/bundle_adjustment.cc:490-528) makes a camera block
// variable under a refine flag; a reconstruction in which every image has its own camera then has C more parameter blocks, each coupled with the same images as
// its own pose.  Behind all pose columns they are 6 C + ... dense-looking block rows that every part of a dissection waits for (500 images, f and k variable:
// 39 chain steps instead of ~25); beside their image's pose columns they belong to that image's part.  Returns n_v > 0 when EVERY image's camera is variable,
// referenced by that image alone, with the same even number n_v of variable parameters (the pose blocks stay 16-byte aligned in the reduced system) - the
// reduced system then has 6 + n_v columns per image and no tail -, 0 otherwise (variable intrinsics, if any, follow the pose columns).  PPSFM_BA_INTR_LAYOUT=tail: 0.
int PrivateIntrinsicsColumns(const pp_ba_problem_desc* d) {
  if (!d->camera_const_mask) return 0;
  if (const char* e = std::getenv("PPSFM_BA_INTR_LAYOUT")) if (e[0] == 't' || e[0] == 'T') return 0;
  const int C = d->num_poses, K = d->num_cameras;
  std::vector<int32_t> users(K, 0);
  for (int c = 0; c < C; ++c) users[d->pose_camera[c]]++;
  int nv_all = -1;
  for (int c = 0; c < C; ++c) {
    const int k = d->pose_camera[c];
    if (users[k] != 1) return 0;
    const int np = CameraNumParams(d->camera_model[k]);
    int nv = 0;
    for (int j = 0; j < np; ++j) if (!((d->camera_const_mask[k] >> j) & 1)) ++nv;
    if (nv == 0 || (nv & 1) || (nv_all >= 0 && nv != nv_all)) return 0;
    nv_all = nv;
  }
  return nv_all > 0 ? nv_all : 0;
}

// the variable intrinsics columns of a problem (pp_ba_create's layout: block k at intr_off[k])
int CountVariableIntrinsics(const pp_ba_problem_desc* d) {
  if (!d->camera_const_mask) return 0;
  const int C = d->num_poses, K = d->num_cameras;
  std::vector<char> cam_used(K, 0);
  for (int c = 0; c < C; ++c) cam_used[d->pose_camera[c]] = 1;
  int NI = 0;
  for (int k = 0; k < K; ++k) {
    if (!cam_used[k]) continue;
    const int np = CameraNumParams(d->camera_model[k]);
    for (int j = 0; j < np; ++j) if (!((d->camera_const_mask[k] >> j) & 1)) ++NI;
  }
  return NI;
}

// Co-visibility of the variable images (two images are neighbours when a variable point is seen by both) as a bit matrix, filled point by point.
// `stop_at` > 0: the fill stops once that many distinct edges exist (the caller's "too dense for any order to pay"); returns false then.
// `none_fixed`: every image is a node (each carries variable intrinsics of its own: its columns of the reduced system exist whatever its pose is).
static bool FillCoVisibility(const pp_ba_problem_desc* d, std::vector<uint64_t>* bits, int W, int64_t stop_at, int64_t* edges_out, bool none_fixed) {
  const int C = d->num_poses, P = d->num_points;
  const int64_t M = d->num_obs;
  // observations grouped by point: the caller's arrays as they are when obs_point never decreases (BundleAdjuster::SetUp adds a point's track at a time,
  // src/optim/bundle_adjustment.cc:330-338), a counting sort otherwise
  std::vector<int32_t> ps(P + 1, 0), po_sorted;
  bool grouped = true;
  for (int64_t o = 0; o < M; ++o) { ps[d->obs_point[o] + 1]++; grouped = grouped && (o == 0 || d->obs_point[o] >= d->obs_point[o - 1]); }
  for (int p = 0; p < P; ++p) ps[p + 1] += ps[p];
  if (!grouped) { po_sorted.resize(M); std::vector<int32_t> f(ps.begin(), ps.end() - 1); for (int64_t o = 0; o < M; ++o) po_sorted[f[d->obs_point[o]]++] = d->obs_pose[o]; }
  const int32_t* po_base = grouped ? d->obs_pose : po_sorted.data();
  std::vector<uint8_t> fixed(C, 0);
  if (d->pose_const && !none_fixed) for (int c = 0; c < C; ++c) fixed[c] = d->pose_const[c] ? 1 : 0;
  int64_t edges = 0;
  uint64_t* b = bits->data();
  std::vector<int32_t> obs;      // the variable observers of one point
  for (int p = 0; p < P; ++p) {
    if (d->point_const && d->point_const[p]) continue;
    obs.clear();
    for (int e = ps[p]; e < ps[p + 1]; ++e) if (!fixed[po_base[e]]) obs.push_back(po_base[e]);
    const int n = (int)obs.size();
    const int32_t* v = obs.data();
    for (int a = 1; a < n; ++a) {
      const int ca = v[a];
      for (int c = 0; c < a; ++c) {
        const int cb = v[c];
        const int hi = ca > cb ? ca : cb, lo = ca > cb ? cb : ca;      // lower triangle only (an image seen twice by a point: a bit on the diagonal, not counted)
        uint64_t& w = b[(size_t)hi * W + (lo >> 6)];
        const uint64_t m = 1ull << (lo & 63);
        edges += (hi != lo) & !(w & m);
        w |= m;
      }
    }
    if (stop_at > 0 && edges >= stop_at) { *edges_out = edges; return false; }
  }
  *edges_out = edges;
  return true;
}

namespace {
// what decides whether an order is looked for at all (before any graph is built)
struct OrderingSetup {
  bool will_iterate, forced, candidate;
  const char* eo;
  int Tt;
};
OrderingSetup SetupOf(const pp_ba_problem_desc* d, int NI) {
  OrderingSetup u;
  const int C = d->num_poses;
  int ls = d->linear_solver;
  if (const char* e = std::getenv("PPSFM_BA_LINEAR_SOLVER")) ls = (e[0] == 'i' || e[0] == 'I') ? PP_LINEAR_SOLVER_ITERATIVE_SCHUR : ((e[0] == 'd' || e[0] == 'D') ? PP_LINEAR_SOLVER_DIRECT : ls);
  u.will_iterate = ls == PP_LINEAR_SOLVER_ITERATIVE_SCHUR || (ls == PP_LINEAR_SOLVER_AUTO && C > PP_MAX_NUM_IMAGES_DIRECT_SOLVER);      // (as pp_ba_create's h->iterative)
  const char* es = std::getenv("PPSFM_BA_SPARSE");
  u.eo = std::getenv("PPSFM_BA_ORDERING");      // natural | rcm (forced even where it does not pay: tests) | band (no dissection) | unset = by chain steps
  u.forced = u.eo && (u.eo[0] == 'r' || u.eo[0] == 'R');
  u.Tt = ((6 * C + NI + 1 + 63) / 64);
  u.candidate = d->ordering == PP_ORDERING_AUTO && !(u.eo && (u.eo[0] == 'n' || u.eo[0] == 'N')) && !u.will_iterate && C >= 3 &&
                (u.forced || (!(es && std::atoi(es) == 0) && u.Tt >= 8));
  return u;
}
}  // namespace

// true when ChooseImageOrdering will build the co-visibility graph from the observations (pp_ba_create then hands it the graph's bits from the device,
// pair_lists.hip CoVisibilityOnDevice, where the by-point lists are anyway)
bool OrderingReadsObservations(const pp_ba_problem_desc* d, int NI) { return SetupOf(d, NI).candidate && !d->covisibility; }

// `graph_bits` (C x ceil(C / 64) words, bit j of row i for j < i: images i and j share a variable point, both variable) replaces the walk over the
// observations when given.
ImageOrdering ChooseImageOrdering(const pp_ba_problem_desc* d, int NI, const uint64_t* graph_bits) {
  const auto t_begin = std::chrono::steady_clock::now();
  const int C = d->num_poses;
  ImageOrdering out;
  const OrderingSetup setup = SetupOf(d, NI);
  const bool will_iterate = setup.will_iterate, forced = setup.forced, candidate = setup.candidate;
  const char* eo = setup.eo;
  const int Tt = setup.Tt;
  // columns per image and what follows the images (the shared intrinsics blocks; none when every image carries its own)
  const int nv_private = will_iterate ? 0 : PrivateIntrinsicsColumns(d);
  const int W6 = 6 + nv_private, tail0 = W6 * C, NI_tail = NI - nv_private * C;
  const Grain grain(W6);
  const int bias_images = 128 / W6;      // (two chain steps: see BestBandCut)
  auto finish = [&]() {
    out.plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (std::getenv("PPSFM_ORDER_DEBUG")) fprintf(stderr, "ppsfm: image ordering %.3f ms (%d images, %s)\n", out.plan_ms, C, out.dense_exit ? "co-visibility too dense: left early" : (out.old_of_new.empty() ? "caller's order" : "renumbered"));
    return out;
  };
  if (!candidate) return finish();
  const bool dbg = std::getenv("PPSFM_ORDER_DEBUG") != nullptr;
  auto lap = [&, last = t_begin](const char* what) mutable {
    if (!dbg) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "ppsfm:   %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
  };

  // ---- the co-visibility graph ------------------------------------------------------------------------------------------------------------------------
  // A tile of the reduced system holds at least 10 x 10 image pairs, so an order can empty at most (non-edges / 100) tiles: once fewer than a tenth of the
  // tiles could go, no order is taken (the rule below) and the fill stops - a clique leaves after a fraction of its pairs.
  Adjacency adj(C);
  const int W = (C + 63) / 64;
  {
    int64_t variable = 0;
    // (an image whose pose is constant is no node of the graph - it has no columns - unless it carries variable intrinsics of its own beside them)
    auto fixed_image = [&](int c) { return nv_private == 0 && d->pose_const && d->pose_const[c]; };
    for (int c = 0; c < C; ++c) variable += fixed_image(c) ? 0 : 1;
    const int64_t all_pairs = variable * (variable - 1) / 2, total_tiles = (int64_t)Tt * (Tt + 1) / 2;
    const int64_t stop_at = forced ? 0 : std::max<int64_t>(1, all_pairs - 10 * total_tiles + 1);      // non-edges < total_tiles / 10 * 100
    std::vector<uint64_t> bits((size_t)C * W, 0);
    int64_t edges = 0;
    if (d->covisibility) {
      // the caller's (a point-sharded group's UNION) co-visibility: C x C bytes, symmetric
      for (int i = 1; i < C; ++i) {
        if (fixed_image(i)) continue;
        for (int j = 0; j < i; ++j)
          if ((d->covisibility[(size_t)i * C + j] || d->covisibility[(size_t)j * C + i]) && !fixed_image(j)) { bits[(size_t)i * W + (j >> 6)] |= 1ull << (j & 63); ++edges; }
      }
      if (stop_at > 0 && edges >= stop_at) { out.dense_exit = true; return finish(); }
    } else if (graph_bits) {
      for (int i = 1; i < C; ++i)
        for (int w = 0; w <= (i >> 6); ++w) { const uint64_t m = graph_bits[(size_t)i * W + w]; bits[(size_t)i * W + w] = m; edges += __builtin_popcountll(m); }
      if (stop_at > 0 && edges >= stop_at) { out.dense_exit = true; return finish(); }
    } else if (!FillCoVisibility(d, &bits, W, stop_at, &edges, nv_private > 0)) { out.dense_exit = true; return finish(); }
    for (int i = 0; i < C; ++i) adj[i].reserve(16);
    for (int i = 0; i < C; ++i)
      for (int w = 0; w <= (i >> 6); ++w) {
        uint64_t m = bits[(size_t)i * W + w];
        while (m) { const int j = 64 * w + __builtin_ctzll(m); m &= m - 1; if (j != i) { adj[i].push_back(j); adj[j].push_back(i); } }
      }
    for (int i = 0; i < C; ++i) std::sort(adj[i].begin(), adj[i].end());
  }
  lap("co-visibility graph");

  // the tile map of an order (image c at position pos[c]; null: the caller's order), closed under fill-in; returns its non-zero tiles
  auto tiles_map = [&](const std::vector<int32_t>* pos, std::vector<uint8_t>* nz) {
    nz->assign((size_t)Tt * Tt, 0);
    auto at = [&](int c) { return pos ? (*pos)[c] : c; };
    auto mark_t = [&](int r0, int c0) {
      if (r0 < c0) std::swap(r0, c0);
      for (int ti = r0 / 64; ti <= (r0 + W6 - 1) / 64; ++ti)
        for (int tj = c0 / 64; tj <= (c0 + W6 - 1) / 64; ++tj) if (tj <= ti) (*nz)[(size_t)ti * Tt + tj] = 1;
    };
    for (int c = 0; c < C; ++c) { mark_t(W6 * at(c), W6 * at(c)); for (int c2 : adj[c]) if (c2 < c) mark_t(W6 * at(c), W6 * at(c2)); }
    for (int ti = tail0 / 64; ti <= (tail0 + NI_tail) / 64; ++ti)      // the shared intrinsics rows (they couple with every image) and the right-hand side's row
      for (int tj = 0; tj <= ti; ++tj) (*nz)[(size_t)ti * Tt + tj] = 1;
    return SymbolicTileFill(Tt, nz->data());
  };
  // what a candidate costs: the chain steps of its factorisation (the block columns on the longest dependency path when the block-sparse one-launch mode
  // takes it, all of them otherwise) - from the plan alone; the winner's list is verified below
  int last_chains = 1;
  auto steps_of = [&](const std::vector<uint8_t>& nz, int nnz) {
    const bool sparse_path = (int64_t)nnz * 10 <= (int64_t)Tt * (Tt + 1) / 2 * 7;      // (SparseActive's threshold)
    last_chains = 1;
    return sparse_path ? CholeskyPlanSteps(Tt, nz.data(), &last_chains) : Tt;
  };
  double acc_map = 0, acc_steps = 0;
  auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::vector<uint8_t> nzmap;
  std::vector<int32_t> oon = ReverseCuthillMcKee(adj), noo(C);
  for (int i = 0; i < C; ++i) noo[oon[i]] = i;
  out.nnz_natural = tiles_map(nullptr, &nzmap);
  const int natural_steps = steps_of(nzmap, out.nnz_natural);
  out.nnz_ordered = tiles_map(&noo, &nzmap);
  bool identity = true;
  for (int i = 0; i < C; ++i) identity = identity && oon[i] == i;
  // Cuthill-McKee: taken when it removes at least a tenth of the factor's tiles (a dense co-visibility keeps the caller's order: nothing to gain, and the
  // solve stays bit-for-bit what it was)
  lap("cuthill-mckee + two tile maps");
  const bool take_rcm = !identity && (forced || (int64_t)out.nnz_ordered * 10 <= (int64_t)out.nnz_natural * 9);
  const bool no_nd = eo && (eo[0] == 'r' || eo[0] == 'R' || eo[0] == 'b' || eo[0] == 'B');      // rcm (forced) / band (by tile count): the band order only
  struct Candidate { std::vector<int32_t> oon, noo; int steps = 0, nnz = 0, chains = 1; };
  Candidate base;      // the band (or the caller's order)
  base.nnz = take_rcm ? out.nnz_ordered : out.nnz_natural;
  base.steps = take_rcm ? steps_of(nzmap, out.nnz_ordered) : natural_steps;
  base.chains = take_rcm ? last_chains : 1;
  if (take_rcm) { base.oon = oon; base.noo = noo; }
  std::vector<Candidate> ranked;      // dissections that beat the band, best first
  if (!no_nd && Tt <= 128) {
    std::vector<int32_t> band, rest, start(C);
    for (int i = 0; i < C; ++i) start[i] = take_rcm ? oon[i] : i;
    for (int i = 0; i < C; ++i) (adj[start[i]].empty() ? rest : band).push_back(start[i]);
    std::vector<std::vector<int32_t>> tried;
    auto consider = [&](std::vector<int32_t> cand) {
      if (cand == band || std::find(tried.begin(), tried.end(), cand) != tried.end()) return false;      // (the same order as an earlier trial: deeper levels / the other balance found no new cut)
      tried.push_back(cand);
      cand.insert(cand.end(), rest.begin(), rest.end());
      Candidate c;
      c.noo.resize(C);
      for (int i = 0; i < C; ++i) c.noo[cand[i]] = i;
      const double ta = now_ms();
      c.nnz = tiles_map(&c.noo, &nzmap);
      const double tb = now_ms();
      c.steps = steps_of(nzmap, c.nnz);
      acc_map += tb - ta; acc_steps += now_ms() - tb;
      c.chains = last_chains;
      c.oon.swap(cand);
      // (a dissection has MORE tiles than its band - the separators' rows fill - and pays when the chain it shortens is what bounds the factorisation:
      // taken from 0.95 of the band's steps; among dissections the fewest steps, then the most chains)
      if (dbg) fprintf(stderr, "ppsfm:   candidate: %d tiles, %d chains, %d steps (band: %d tiles, %d steps)\n", c.nnz, c.chains, c.steps, base.nnz, base.steps);
      if (c.chains > 1 && c.steps * 100 <= base.steps * 95) ranked.push_back(std::move(c));
      return true;
    };
    BandDissector bd(adj, grain);
    for (int trial = 0; trial < 8; ++trial) {      // the band: one to four levels, cuts balanced evenly / in favour of the part whose chain runs on (the bias)
      std::vector<int32_t> cand = bd.Dissect(band, 1 + trial / 2, false, (trial & 1) ? bias_images : 0);
      if (cand == band) { if (trial & 1) break; continue; }
      consider(std::move(cand));
    }
    lap("band dissections");
    GraphDissector gd(adj, grain);      // the graph itself: separators from level structures (clusters, hubs: what no band order shows)
    // (asked for when the cuts of the band left more than 0.7 of its steps: on sequences, rings and the clustered collections of the tests the band's cuts -
    // parts re-ordered by their own Cuthill-McKee - find the same separators, and the graph's would only cost their milliseconds.  PPSFM_BA_GRAPH_ND=1 / 0: always / never)
    int best_band_steps = base.steps;
    for (const Candidate& c : ranked) best_band_steps = std::min(best_band_steps, c.steps);
    const char* eg = std::getenv("PPSFM_BA_GRAPH_ND");
    const bool graph_nd = eg ? std::atoi(eg) != 0 : best_band_steps * 10 > base.steps * 7;
    for (int levels = 1; graph_nd && levels <= 4; ++levels) if (!consider(gd.Dissect(band, levels)) && levels > 1) break;
    lap("graph dissections");
    std::stable_sort(ranked.begin(), ranked.end(), [](const Candidate& a, const Candidate& b) { return a.steps != b.steps ? a.steps < b.steps : a.chains > b.chains; });
  }
  // the winner's task list is built and replayed ONCE (a list of several chains that fails its replay would run as one chain: the next candidate then)
  Candidate* chosen = &base;
  for (Candidate& c : ranked) {
    (void)tiles_map(&c.noo, &nzmap);
    int chains = 1;
    const int steps = CholeskyChainSteps(Tt, nzmap.data(), &chains);
    if (chains == c.chains && steps == c.steps) { chosen = &c; break; }
  }
  lap("winner's list + replay");
  if (dbg) fprintf(stderr, "ppsfm:   (candidates: tile maps %.3f ms, chain plans %.3f ms)\n", acc_map, acc_steps);
  if (!chosen->oon.empty()) { out.nnz_ordered = chosen->nnz; out.old_of_new.swap(chosen->oon); out.new_of_old.swap(chosen->noo); }
  out.chains = chosen->chains; out.chain_steps = chosen->steps;
  return finish();
}

}  // namespace ppsfm

extern "C" int pp_ba_plan_ordering(const pp_ba_problem_desc* d, int32_t* old_of_new, int32_t* info) try {
  using namespace ppsfm;
  PP_REQUIRE(d && info && d->obs_pose && d->obs_point && d->pose_camera && d->camera_model, "pp_ba_plan_ordering: null argument");
  const int C = d->num_poses, P = d->num_points, K = d->num_cameras;
  PP_REQUIRE(C > 0 && P > 0 && K > 0 && d->num_obs > 0, "pp_ba_plan_ordering: empty problem (poses %d, points %d, cameras %d, obs %lld)", C, P, K, (long long)d->num_obs);      // (as pp_ba_create)
  for (int c = 0; c < C; ++c) PP_REQUIRE(d->pose_camera[c] >= 0 && d->pose_camera[c] < K, "pp_ba_plan_ordering: pose_camera[%d] out of range", c);
  for (int64_t o = 0; o < d->num_obs; ++o)
    PP_REQUIRE(d->obs_pose[o] >= 0 && d->obs_pose[o] < C && d->obs_point[o] >= 0 && d->obs_point[o] < P, "pp_ba_plan_ordering: observation %lld indexes out of range", (long long)o);
  PP_REQUIRE(d->ordering >= PP_ORDERING_DEFAULT && d->ordering <= PP_ORDERING_AUTO, "pp_ba_plan_ordering: unknown ordering %d", d->ordering);
  const int NI = CountVariableIntrinsics(d);
  const ImageOrdering ord = ChooseImageOrdering(d, NI);
  const int Tt = (6 * C + NI + 1 + 63) / 64;
  const bool iterate = d->linear_solver == PP_LINEAR_SOLVER_ITERATIVE_SCHUR || (d->linear_solver == PP_LINEAR_SOLVER_AUTO && C > PP_MAX_NUM_IMAGES_DIRECT_SOLVER);
  const int nvp = iterate ? 0 : PrivateIntrinsicsColumns(d), W6 = 6 + nvp, tail0 = W6 * C, NI_tail = NI - nvp * C;
  // the tile map of the order chosen -> chains and chain steps of its one-launch factorisation
  std::vector<uint8_t> nz((size_t)Tt * Tt, 0);
  {
    std::vector<std::vector<int32_t>> obs_of_point(P);
    for (int64_t o = 0; o < d->num_obs; ++o) if (!(d->point_const && d->point_const[d->obs_point[o]])) obs_of_point[d->obs_point[o]].push_back(d->obs_pose[o]);
    auto at = [&](int c) { return ord.new_of_old.empty() ? c : ord.new_of_old[c]; };
    auto mark = [&](int r0, int c0) {
      if (r0 < c0) std::swap(r0, c0);
      for (int ti = r0 / 64; ti <= (r0 + W6 - 1) / 64; ++ti) for (int tj = c0 / 64; tj <= (c0 + W6 - 1) / 64; ++tj) if (tj <= ti) nz[(size_t)ti * Tt + tj] = 1;
    };
    for (int c = 0; c < C; ++c) mark(W6 * at(c), W6 * at(c));
    for (const auto& v : obs_of_point)
      for (size_t a = 0; a < v.size(); ++a) for (size_t b = 0; b < a; ++b)
        if (nvp > 0 || !(d->pose_const && (d->pose_const[v[a]] || d->pose_const[v[b]]))) mark(W6 * at(v[a]), W6 * at(v[b]));
    if (d->covisibility)
      for (int i = 0; i < C; ++i) for (int j = 0; j < i; ++j)
        if ((d->covisibility[(size_t)i * C + j] || d->covisibility[(size_t)j * C + i]) && (nvp > 0 || !(d->pose_const && (d->pose_const[i] || d->pose_const[j])))) mark(W6 * at(i), W6 * at(j));
    for (int ti = tail0 / 64; ti <= (tail0 + NI_tail) / 64; ++ti) for (int tj = 0; tj <= ti; ++tj) nz[(size_t)ti * Tt + tj] = 1;
  }
  const int nnz = SymbolicTileFill(Tt, nz.data());
  const bool sparse_path = Tt >= 8 && (int64_t)nnz * 10 <= (int64_t)Tt * (Tt + 1) / 2 * 7;
  int chains = 1;
  const int steps = sparse_path && Tt <= 128 ? CholeskyChainSteps(Tt, nz.data(), &chains) : Tt;
  info[0] = ord.old_of_new.empty() ? 0 : 1; info[1] = ord.nnz_natural; info[2] = nnz; info[3] = chains; info[4] = steps; info[5] = Tt; info[6] = sparse_path ? 1 : 0; info[7] = NI;
  if (old_of_new) for (int c = 0; c < C; ++c) old_of_new[c] = ord.old_of_new.empty() ? c : ord.old_of_new[c];
  return PP_OK;
} PP_API_CATCH("pp_ba_plan_ordering")
