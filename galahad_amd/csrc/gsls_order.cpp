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
#include <atomic>
#include <cstdlib>
#include <new>
#include <numeric>
#include <thread>

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

namespace {

struct Sub {
  std::vector<int> verts;
  int hi;  // positions [hi - verts.size(), hi) belong to this subproblem
};

// Scratch of one worker: the search arrays (indexed by vertex; a worker only ever touches the vertices of its own
// subproblems, and a vertex outside them reads as "not mine") and the id counter that goes with them.
struct Dissector {
  Work w;
  int next_id = 0;
  std::vector<int> order, lptr, order2, lptr2;
  Dissector(const std::vector<int64_t>& a, const std::vector<int>& r, int n) : w(a, r, n) {}
};

// One step: take a subproblem off the stack, number it (leaf) or cut it and push the pieces.  What a subproblem gets
// depends on its vertices and the graph only -- not on the order in which subproblems are taken, nor on who takes them.
void dissect_one(Dissector& d, std::vector<Sub>& stack, std::vector<int>& perm, int leaf_size) {
  Work& w = d.w;
  std::vector<int>&order = d.order, &lptr = d.lptr, &order2 = d.order2, &lptr2 = d.lptr2;
  const std::vector<int64_t>& aptr = w.ap;
  const std::vector<int>& arow = w.ar;
  {
    Sub sub = std::move(stack.back());
    stack.pop_back();
    const int id = d.next_id++;
    const int cnt = int(sub.verts.size());
    if (cnt == 0) return;
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
      return;
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
      return;
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

}  // namespace

void order_nested_dissection(int n, const std::vector<int64_t>& aptr, const std::vector<int>& arow,
                             std::vector<int>& perm) {
  perm.assign(n, -1);
  if (n == 0) return;
  int leaf_size = 128;
  if (const char* e = std::getenv("GSLS_ND_LEAF")) leaf_size = std::max(8, std::atoi(e));   // tuning knob
  int nthreads = int(std::min(4u, std::max(1u, std::thread::hardware_concurrency())));   // (the top cut is sequential: more do not pay)
  if (const char* e = std::getenv("GSLS_ND_THREADS")) nthreads = std::max(1, std::atoi(e));
  if (n < 200000) nthreads = 1;

  std::vector<Sub> stack;
  {
    Sub all;
    all.verts.resize(n);
    std::iota(all.verts.begin(), all.verts.end(), 0);
    all.hi = n;
    stack.push_back(std::move(all));
  }
  Dissector d0(aptr, arow, n);
  if (nthreads == 1) {
    while (!stack.empty()) dissect_one(d0, stack, perm, leaf_size);
    return;
  }
  // The top of the dissection tree on this thread, always the largest pending subproblem first, until there are enough
  // pieces to share out; then every worker takes pieces (largest first, from a common counter) and finishes each of
  // them on a stack and scratch arrays of its own.  The result is the one-thread result, whatever the schedule.
  const size_t want = size_t(nthreads) * 8;
  while (!stack.empty() && stack.size() < want) {
    size_t big = 0;
    for (size_t k = 1; k < stack.size(); ++k)
      if (stack[k].verts.size() > stack[big].verts.size()) big = k;
    if (stack[big].verts.size() <= size_t(std::max(leaf_size, 4096))) break;
    std::swap(stack[big], stack.back());
    dissect_one(d0, stack, perm, leaf_size);
  }
  std::sort(stack.begin(), stack.end(), [](const Sub& x, const Sub& y) { return x.verts.size() > y.verts.size(); });
  std::atomic<size_t> next(0);
  std::atomic<bool> failed(false);
  auto worker = [&](Dissector* mine) {
    try {
      Dissector local_d(aptr, arow, mine ? 0 : n);
      Dissector& d = mine ? *mine : local_d;
      std::vector<Sub> own;
      for (;;) {
        const size_t k = next.fetch_add(1);
        if (k >= stack.size()) break;
        own.clear();
        own.push_back(std::move(stack[k]));
        while (!own.empty()) dissect_one(d, own, perm, leaf_size);
      }
    } catch (...) {
      failed = true;
    }
  };
  std::vector<std::thread> pool;
  try {
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker, static_cast<Dissector*>(nullptr));
  } catch (...) {          // (no more threads to be had: the ones that started and this one do the work)
  }
  worker(&d0);
  for (auto& th : pool) th.join();
  if (failed) throw std::bad_alloc();
}


// =================================================================================================
// Approximate minimum degree (gsls_options.ordering = GSLS_ORDER_AMD): the ordering GALAHAD's SLS offers as
// control%ordering = 1 through HSL MC68 (src/sls/sls.f90:2263-2330; MC68 is a stub in the reference tree), for irregular
// patterns where level-structure nested dissection finds no small separators.  Quotient-graph elimination in the
// Amestoy-Davis-Duff scheme: a variable keeps its remaining original neighbours A_i and the elements (eliminated
// pivots) E_i it belongs to; eliminating p forms the element L_p = A_p U (U_{e in E_p} L_e) \ {p}, absorbs the elements of
// E_p, and the degree of every i in L_p becomes the approximate external degree
//      d_i = min( n - k,  d_i + |L_p \ i|,  |A_i \ L_p| + |L_p \ i| + sum_{e in E_i \ p} |L_e \ L_p| )
// with |L_e \ L_p| obtained for all e at once by the counting pass over L_p (the "w" trick).  Variables that become
// indistinguishable are not merged (no supervariables) and there is no aggressive absorption: host work that runs once
// per analyse, O(sum of |L_p| * elements touched).
// =================================================================================================
void order_amd(int n, const std::vector<int64_t>& aptr, const std::vector<int>& arow, std::vector<int>& perm) {
  perm.assign(n, -1);
  if (n == 0) return;
  std::vector<std::vector<int>> A(n), E(n), L(n);      // L[e]: variables of element e (e = an eliminated pivot)
  std::vector<int> deg(n), w(n, -1), mark(n, -1);
  std::vector<char> dead(n, 0), elim(n, 0);
  for (int i = 0; i < n; ++i) {
    for (int64_t k = aptr[i]; k < aptr[i + 1]; ++k)
      if (arow[k] != i) A[i].push_back(arow[k]);
    std::sort(A[i].begin(), A[i].end());
    A[i].erase(std::unique(A[i].begin(), A[i].end()), A[i].end());
    deg[i] = int(A[i].size());
  }
  // degree buckets as doubly linked lists
  std::vector<int> head(n + 1, -1), next(n, -1), prev(n, -1);
  auto bucket_insert = [&](int i) {
    const int d = deg[i];
    next[i] = head[d];
    prev[i] = -1;
    if (head[d] >= 0) prev[head[d]] = i;
    head[d] = i;
  };
  auto bucket_remove = [&](int i) {
    const int d = deg[i];
    if (prev[i] >= 0) next[prev[i]] = next[i]; else head[d] = next[i];
    if (next[i] >= 0) prev[next[i]] = prev[i];
  };
  for (int i = 0; i < n; ++i) bucket_insert(i);
  int mindeg = 0;
  std::vector<int> Lp;
  for (int k = 0; k < n; ++k) {
    while (mindeg < n && head[mindeg] < 0) ++mindeg;
    const int p = head[mindeg];
    bucket_remove(p);
    elim[p] = 1;
    perm[p] = k;
    const int kp = k;          // (mass elimination below advances k)
    // ---- the new element ----------------------------------------------------------------------------------
    Lp.clear();
    mark[p] = kp;
    for (int i : A[p])
      if (!elim[i] && mark[i] != kp) { mark[i] = kp; Lp.push_back(i); }
    for (int e : E[p]) {
      if (dead[e]) continue;
      for (int i : L[e])
        if (!elim[i] && mark[i] != kp) { mark[i] = kp; Lp.push_back(i); }
      dead[e] = 1;                       // absorbed into p
      std::vector<int>().swap(L[e]);
    }
    std::vector<int>().swap(A[p]);
    std::vector<int>().swap(E[p]);
    // ---- |L_e \ L_p| for every element adjacent to a variable of L_p -------------------------------------------
    for (int i : Lp)
      for (int e : E[i]) {
        if (dead[e]) continue;
        if (w[e] < 0) w[e] = int(L[e].size());
        --w[e];
      }
    // ---- update the variables of L_p --------------------------------------------------------------------------
    const int lp = int(Lp.size());
    for (int i : Lp) {
      bucket_remove(i);
      // original neighbours now covered by the element (members of L_p, and p itself) leave A_i
      size_t o = 0;
      for (int j : A[i])
        if (!elim[j] && mark[j] != kp) A[i][o++] = j;
      A[i].resize(o);
      // dead elements leave E_i, p joins
      o = 0;
      int ext = 0;
      for (int e : E[i])
        if (!dead[e]) { E[i][o++] = e; ext += std::max(w[e], 0); }
      E[i].resize(o);
      E[i].push_back(p);
      const int d = std::min(std::min(n - kp - 1, deg[i] + lp - 1), int(A[i].size()) + (lp - 1) + ext);
      deg[i] = std::max(d, 0);
    }
    for (int i : Lp)
      for (int e : E[i])
        if (e != p) w[e] = -1;
    // mass elimination: a variable of L_p with no original neighbour left and no other element is adjacent to exactly
    // L_p \ {i}: it can follow p at once without any further fill
    {
      size_t o = 0;
      int gone = 0;
      for (int i : Lp) {
        if (A[i].empty() && E[i].size() == 1) {
          elim[i] = 1;
          perm[i] = ++k;
          std::vector<int>().swap(E[i]);
          ++gone;
        } else {
          Lp[o++] = i;
        }
      }
      Lp.resize(o);
      if (gone)
        for (int i : Lp) deg[i] = std::max(deg[i] - gone, 0);
    }
    for (int i : Lp) {
      bucket_insert(i);
      mindeg = std::min(mindeg, deg[i]);
    }
    L[p] = Lp;
  }
}

}  // namespace gsls
