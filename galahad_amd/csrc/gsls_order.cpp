// Fill-reducing ordering for the gsls backend.
//
// Stands where SSIDS calls METIS (src/ssids/ssids.f90:305-320, options%ordering = 1); METIS, MC68 and
// MC61 are stubs in the reference tree (src/dum/metis.f, src/dum/hsl_mc68i.f90), so there is no
// reference ordering to match -- any permutation is a legal answer and the oracle is simply given
// the same PERM.  What the MI355X wants from an ordering is different from what a CPU wants: a
// bushy assembly tree (many independent fronts per level, few levels) matters more than the last
// 20 % of fill, because every tree level costs a dependent launch and the chip needs thousands of
// workgroups to fill 256 CUs.  Hence nested dissection all the way down to small leaves:
//
//   automatic nested dissection (George & Liu): level structure rooted at a pseudo-peripheral
//   vertex, the narrowest level in the middle third is the separator (thinned to the vertices that
//   actually touch the far side), recurse on the connected pieces, separators numbered last;
//   leaves are numbered in breadth-first order (a reverse Cuthill-McKee flavour: small bandwidth,
//   so leaf fronts stay narrow).
#include <algorithm>
#include <cstdlib>
#include <numeric>

#include "gsls_internal.hpp"

namespace gsls {
namespace {

struct Work {
  const std::vector<int64_t>& ap;
  const std::vector<int>& ar;
  std::vector<int> tag;     // tag[v] == id  <=> v belongs to the subproblem being processed
  std::vector<int> lvl;     // BFS level (valid for vertices visited in the current search)
  std::vector<int> mark;    // visit stamp
  int stamp = 0;
  Work(const std::vector<int64_t>& a, const std::vector<int>& r, int n)
      : ap(a), ar(r), tag(n, -1), lvl(n, 0), mark(n, 0) {}
};

// BFS inside the subproblem `id` from `root`; returns the visit order, level starts in lptr
void bfs(Work& w, int id, int root, std::vector<int>& order, std::vector<int>& lptr) {
  order.clear();
  lptr.clear();
  ++w.stamp;
  order.push_back(root);
  w.mark[root] = w.stamp;
  w.lvl[root] = 0;
  lptr.push_back(0);
  size_t head = 0;
  int cur = 0;
  while (head < order.size()) {
    const int v = order[head];
    if (w.lvl[v] != cur) {
      cur = w.lvl[v];
      lptr.push_back(int(head));
    }
    ++head;
    for (int64_t k = w.ap[v]; k < w.ap[v + 1]; ++k) {
      const int u = w.ar[k];
      if (w.tag[u] != id || w.mark[u] == w.stamp) continue;
      w.mark[u] = w.stamp;
      w.lvl[u] = w.lvl[v] + 1;
      order.push_back(u);
    }
  }
  lptr.push_back(int(order.size()));
}

int degree_in(const Work& w, int id, int v) {
  int d = 0;
  for (int64_t k = w.ap[v]; k < w.ap[v + 1]; ++k) d += (w.tag[w.ar[k]] == id);
  return d;
}

}  // namespace

void order_nested_dissection(int n, const std::vector<int64_t>& aptr, const std::vector<int>& arow,
                             std::vector<int>& perm) {
  perm.assign(n, -1);
  if (n == 0) return;
  Work w(aptr, arow, n);
  int leaf_size = 128;
  if (const char* e = std::getenv("GSLS_ND_LEAF")) leaf_size = std::max(8, std::atoi(e));   // tuning knob

  struct Sub {
    std::vector<int> verts;
    int hi;  // positions [hi - verts.size(), hi) belong to this subproblem
  };
  std::vector<Sub> stack;
  {
    Sub all;
    all.verts.resize(n);
    std::iota(all.verts.begin(), all.verts.end(), 0);
    all.hi = n;
    stack.push_back(std::move(all));
  }
  int next_id = 0;
  std::vector<int> order, lptr, order2, lptr2;
  while (!stack.empty()) {
    Sub sub = std::move(stack.back());
    stack.pop_back();
    const int id = next_id++;
    const int cnt = int(sub.verts.size());
    if (cnt == 0) continue;
    for (int v : sub.verts) w.tag[v] = id;

    // connected component of the first vertex; anything unreached becomes its own subproblem
    bfs(w, id, sub.verts[0], order, lptr);
    if (int(order.size()) < cnt) {
      Sub rest, comp;
      for (int v : sub.verts) (w.mark[v] == w.stamp ? comp.verts : rest.verts).push_back(v);
      comp.hi = sub.hi;
      rest.hi = sub.hi - int(comp.verts.size());
      stack.push_back(std::move(rest));
      stack.push_back(std::move(comp));
      continue;
    }
    // pseudo-peripheral root: restart from a minimum-degree vertex of the last level while the
    // level structure keeps getting deeper
    for (int it = 0; it < 4; ++it) {
      const int nl = int(lptr.size()) - 1;
      int best = -1, bestdeg = 0;
      for (int i = lptr[nl - 1]; i < lptr[nl]; ++i) {
        const int d = degree_in(w, id, order[i]);
        if (best < 0 || d < bestdeg) {
          best = order[i];
          bestdeg = d;
        }
      }
      bfs(w, id, best, order2, lptr2);
      if (lptr2.size() <= lptr.size()) break;
      order.swap(order2);
      lptr.swap(lptr2);
    }
    for (int j = 0; j + 1 < int(lptr.size()); ++j)      // levels of the structure that was kept
      for (int i = lptr[j]; i < lptr[j + 1]; ++i) w.lvl[order[i]] = j;
    const int nl = int(lptr.size()) - 1;

    bool leaf = (cnt <= leaf_size) || nl < 3;
    int jsep = -1;
    if (!leaf) {
      // narrowest level whose two sides both keep at least 30 % of the vertices
      // Candidates: levels that leave at least 30 % of the vertices on either side.  Among those within
      // 15 % of the narrowest one, take the cut that minimises the DEPTH of the dissection tree it
      // leads to (a side spanning k levels of the structure needs about ceil(log2(k+1)) further
      // cuts), then the better balance.  Every tree level is a chain of dependent launches on the
      // GPU, so depth is worth more than the last few per cent of fill.
      int wmin = -1;
      for (int j = 1; j + 1 < nl; ++j) {
        const int left = lptr[j], right = cnt - lptr[j + 1], wj = lptr[j + 1] - lptr[j];
        if (left < 0.3 * cnt || right < 0.3 * cnt) continue;
        if (wmin < 0 || wj < wmin) wmin = wj;
      }
      auto depth_of = [](int k) { int d = 0; while ((1 << d) - 1 < k) ++d; return d; };
      int best_depth = -1;
      int64_t best_imb = 0;
      for (int j = 1; j + 1 < nl && wmin >= 0; ++j) {
        const int left = lptr[j], right = cnt - lptr[j + 1], wj = lptr[j + 1] - lptr[j];
        if (left < 0.3 * cnt || right < 0.3 * cnt) continue;
        if (wj > wmin + wmin * 0.15) continue;
        const int dep = std::max(depth_of(j), depth_of(nl - j - 1));
        const int64_t imb = std::abs(left - right);
        if (jsep < 0 || dep < best_depth || (dep == best_depth && imb < best_imb)) {
          jsep = j;
          best_depth = dep;
          best_imb = imb;
        }
      }
      if (jsep < 0) {   // no balanced cut: take the level holding the median vertex
        for (int j = 1; j + 1 < nl; ++j)
          if (lptr[j + 1] > cnt / 2) {
            jsep = j;
            break;
          }
      }
      if (jsep < 0) leaf = true;
    }
    if (leaf) {
      // breadth-first numbering, deepest level first so the root of the search is eliminated last
      int pos = sub.hi - cnt;
      for (int i = cnt - 1; i >= 0; --i) perm[order[i]] = pos++;
      continue;
    }
    // separator = vertices of level jsep with a neighbour in level jsep+1
    Sub left, right;
    std::vector<int> sep;
    for (int i = lptr[jsep]; i < lptr[jsep + 1]; ++i) {
      const int v = order[i];
      bool touches = false;
      for (int64_t k = aptr[v]; k < aptr[v + 1] && !touches; ++k) {
        const int u = arow[k];
        touches = (w.tag[u] == id && w.lvl[u] == jsep + 1);
      }
      (touches ? sep : left.verts).push_back(v);
    }
    for (int i = 0; i < lptr[jsep]; ++i) left.verts.push_back(order[i]);
    for (int i = lptr[jsep + 1]; i < cnt; ++i) right.verts.push_back(order[i]);
    int pos = sub.hi - int(sep.size());
    for (size_t i = 0; i < sep.size(); ++i) perm[sep[i]] = pos + int(i);
    for (int v : sep) w.tag[v] = -1;
    right.hi = sub.hi - int(sep.size());
    left.hi = right.hi - int(right.verts.size());
    stack.push_back(std::move(left));
    stack.push_back(std::move(right));
  }
}

}  // namespace gsls
