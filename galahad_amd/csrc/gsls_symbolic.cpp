// Symbolic analysis for the gsls backend: elimination tree, postorder, column counts, relaxed
// supernodes, row lists, A->front map, plus the level-set schedule the HIP kernels run.
//
// The numbers this produces (nnodes, sptr, sparent, rptr, rlist, nptr/nlist, num_factor,
// num_flops, final order) must equal the reference's for the same (pattern, order, nemin), because
// SLS surfaces them (inform%entries_in_factors / flops_elimination, src/sls/sls.f90:1758-1759) and
// bench.py prices throughput with them.  The behaviour followed is that of
//   src/spral/core_analyse.f90:38-151   basic_analyse (driver)
//   :173-223   Liu's elimination tree with path compression
//   :233-352   postorder (children ascending, structurally empty columns last)
//   :387-521   Gilbert/Ng/Peyton column counts
//   :536-853   relaxed supernodes: merge child into parent when no fill is added or both are
//              narrower than nemin; children visited widest-first
//   :911-998   row lists, :1007-1064 sort, :862-902 statistics
//   src/ssids/anal.f90:1129-1231  build_map (A entry -> position in front)
//   src/ssids/anal.f90:37-80      expand_pattern, :147-197 check_order
// It is written from that behaviour, 0-based, on std::vector; integer-only, O(nnz alpha(n)).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <numeric>

#include "gsls_internal.hpp"

namespace gsls {
namespace {

// full symmetric pattern (both triangles) from the lower triangle by columns
void expand_lower(int n, const int64_t* ptr, const int32_t* row, std::vector<int64_t>& aptr,
                  std::vector<int>& arow) {
  aptr.assign(n + 1, 0);
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k) {
      int i = row[k] - 1;
      aptr[i + 1]++;
      if (i != j) aptr[j + 1]++;
    }
  for (int j = 0; j < n; ++j) aptr[j + 1] += aptr[j];
  arow.resize(aptr[n]);
  std::vector<int64_t> fill(aptr.begin(), aptr.end() - 1);
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k) {
      int i = row[k] - 1;
      arow[fill[i]++] = j;
      if (i != j) arow[fill[j]++] = i;
    }
}

// parent[p] for pivot positions p (n = virtual root)
void elimination_tree(int n, const std::vector<int64_t>& aptr, const std::vector<int>& arow,
                      const std::vector<int>& perm, const std::vector<int>& invp,
                      std::vector<int>& parent) {
  parent.assign(n, n);
  std::vector<int> anc(n, n);  // compressed path to the current top of each partial tree
  for (int p = 0; p < n; ++p) {
    int c = invp[p];
    for (int64_t k = aptr[c]; k < aptr[c + 1]; ++k) {
      int j = perm[arow[k]];
      if (j >= p) continue;
      while (anc[j] < p) {
        int nxt = anc[j];
        anc[j] = p;
        j = nxt;
      }
      if (anc[j] == p) continue;  // already hangs below p
      parent[j] = p;
      anc[j] = p;
    }
  }
}

// Relabel pivots in depth-first order: every parent directly after its last child's subtree,
// children kept in ascending order, structurally empty roots moved to the very end.
void postorder(int n, const std::vector<int64_t>& aptr, std::vector<int>& perm,
               std::vector<int>& invp, std::vector<int>& parent, int& realn) {
  std::vector<int> head(n + 1, -1), next(n + 1, -1);
  for (int i = n - 1; i >= 0; --i) {
    next[i] = head[parent[i]];
    head[parent[i]] = i;
  }
  std::vector<int> relabel(n + 1), stack;
  stack.reserve(n + 1);
  realn = n;
  int id = n;
  stack.push_back(n);
  while (!stack.empty()) {
    int v = stack.back();
    stack.pop_back();
    relabel[v] = id--;
    if (v == n) {
      for (int c = head[v]; c != -1; c = next[c])
        if (aptr[invp[c] + 1] != aptr[invp[c]]) stack.push_back(c);
      for (int c = head[v]; c != -1; c = next[c])
        if (aptr[invp[c] + 1] == aptr[invp[c]]) {
          --realn;
          stack.push_back(c);
        }
    } else {
      for (int c = head[v]; c != -1; c = next[c]) stack.push_back(c);
    }
  }
  std::vector<int> old_invp(invp), old_parent(parent);
  for (int i = 0; i < n; ++i) invp[relabel[i]] = old_invp[i];
  for (int i = 0; i < n; ++i) perm[invp[i]] = i;
  for (int i = 0; i < n; ++i) parent[relabel[i]] = relabel[old_parent[i]];
}

int find_top(std::vector<int>& up, int u) {
  int root = u;
  while (up[root] != -1) root = up[root];
  while (up[u] != -1) {  // full path compression
    int nxt = up[u];
    if (nxt != root) up[u] = root;
    u = nxt;
  }
  return root;
}

// cc[p] = number of entries (diagonal included) in column p of L
void column_counts(int n, const std::vector<int64_t>& aptr, const std::vector<int>& arow,
                   const std::vector<int>& perm, const std::vector<int>& invp,
                   const std::vector<int>& parent, std::vector<int>& cc) {
  std::vector<int> first(n + 1);
  std::iota(first.begin(), first.end(), 0);
  cc.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) {
    int p = parent[i];
    first[p] = std::min(first[p], first[i]);
    cc[i] = (first[i] == i) ? 1 : 0;
  }
  std::vector<int> up(n + 1, -1), last_piv(n + 1, -1), last_nbr(n + 1, -1);
  for (int p = 0; p < n; ++p) {
    int c = invp[p];
    for (int64_t k = aptr[c]; k < aptr[c + 1]; ++k) {
      int u = perm[arow[k]];
      if (u <= p) continue;
      if (first[p] > last_nbr[u]) {
        cc[p]++;
        int q = last_piv[u];
        if (q != -1) cc[find_top(up, q)]--;
        last_piv[u] = p;
      }
      last_nbr[u] = p;
    }
    int par = parent[p];
    cc[par] += cc[p] - 1;
    up[p] = par;
  }
}

struct Supernodes {
  int nnodes = 0;
  std::vector<int> sperm;  // old pivot position -> new pivot position
  std::vector<int> sptr, sparent, scc;
};

// keep_branches (our own orderings only -- with a user PERM the reference's partition is reproduced
// exactly): do not let the "no extra fill" rule merge a column into a parent that is a BRANCHING
// point of the elimination tree (two or more sizeable child subtrees).  The merged supernode would
// eliminate the child's columns and the parent's as one block, so the other subtrees -- which only
// feed the parent's own columns -- could no longer overlap with the child's: for nested dissection
// that doubles the number of dependent pivots on the critical path (child separator + parent
// separator at every level instead of one).
// force (may be null): columns that MUST share a supernode with their parent column, whatever the
// amalgamation rules say -- delayed pivots that have to meet the pivot candidates of the front they were
// moved to (gsls_api.cpp: plan_repair).  Always structurally valid: a child's rows are a subset of its
// parent's; the merged front just carries explicit zeros.
void relaxed_supernodes(int n, int realn, const std::vector<int>& parent,
                        const std::vector<int>& cc, int nemin, bool keep_branches, const std::vector<char>* force,
                        Supernodes& out) {
  const int64_t kNever = INT64_MAX;
  std::vector<int> nelim(n + 1, 1), nvert(n + 1, 1), mhead(n + 1, -1), mnext(n + 1, -1);
  std::vector<int64_t> ezero(n + 1, 0);
  std::vector<char> keep(n + 1, 0);
  ezero[n] = kNever;
  nelim[n] = n + 1 + nemin;

  std::vector<int> head(n + 1, -1), next(n + 1, -1);
  for (int i = realn - 1; i >= 0; --i) {
    next[i] = head[parent[i]];
    head[parent[i]] = i;
  }
  std::vector<int> subtree(n + 1, 1);   // vertices in the elimination subtree of each column
  if (keep_branches)
    for (int i = 0; i < realn; ++i) subtree[parent[i]] += subtree[i];
  std::vector<int> kids;
  for (int par = 0; par <= n; ++par) {
    kids.clear();
    for (int c = head[par]; c != -1; c = next[c]) kids.push_back(c);
    std::stable_sort(kids.begin(), kids.end(), [&](int a, int b) { return cc[a] > cc[b]; });
    int big_kids = 0;
    if (keep_branches)
      for (int c : kids) big_kids += (subtree[c] >= 2 * nemin);
    for (int c : kids) {
      bool merge = false;
      if (ezero[par] != kNever)
        merge = (cc[par] == cc[c] - 1 && nelim[par] == 1 && big_kids < 2) ||
                (nelim[par] < nemin && nelim[c] < nemin) || (force && (*force)[c]);
      if (merge) {
        mnext[c] = mhead[par];
        mhead[par] = c;
        ezero[par] += ezero[c] + (int64_t(cc[par]) - 1 + nelim[par] - cc[c] + 1) * nelim[par];
        nelim[par] += nelim[c];
        nvert[par] += nvert[c];
      } else {
        keep[c] = 1;
      }
    }
  }

  out.sperm.assign(n, 0);
  out.sptr.clear();
  out.scc.clear();
  std::vector<int> owner(n + 1, 0), vpar;
  std::vector<int> stack;
  int v = 0, nn = 0;
  for (int node = 0; node < realn; ++node) {
    if (!keep[node]) continue;
    out.sptr.push_back(v);
    vpar.push_back(parent[node]);
    out.scc.push_back(cc[node] + nelim[node] - 1);
    v += nvert[node];
    int k = v;
    stack.assign(1, node);
    while (!stack.empty()) {
      int i = stack.back();
      stack.pop_back();
      out.sperm[i] = --k;
      owner[i] = nn;
      if (mnext[i] != -1) stack.push_back(mnext[i]);
      if (mhead[i] != -1) stack.push_back(mhead[i]);
    }
    ++nn;
  }
  out.sptr.push_back(v);
  out.nnodes = nn;
  owner[n] = nn;
  for (int i = realn; i < n; ++i) out.sperm[i] = i;
  out.sparent.resize(nn);
  for (int s = 0; s < nn; ++s) out.sparent[s] = owner[vpar[s]];
}

}  // namespace

int symbolic_analyse(int n, const int64_t* ptr, const int32_t* row, int32_t* order, int ordering,
                     int nemin, Symbolic& S, const uint8_t* force_var) {
  S = Symbolic();
  S.n = n;
  if (nemin < 1) nemin = 32;
  int flag = GSLS_SUCCESS;
  static const bool dbg_time = getenv("GSLS_DEBUG") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!dbg_time) return;
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[gsls] analyse: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };

  std::vector<int64_t> aptr;
  std::vector<int> arow;
  expand_lower(n, ptr, row, aptr, arow);
  lap("expand pattern");

  // ---- pivot order ---------------------------------------------------------------------------
  S.perm.assign(n, 0);
  S.invp.assign(n, -1);
  if (ordering == GSLS_ORDER_USER) {
    if (!order) return GSLS_ERROR_ORDER;
    for (int i = 0; i < n; ++i) {
      int j = order[i] < 0 ? -order[i] : order[i];
      if (j < 1 || j > n || S.invp[j - 1] != -1) return GSLS_ERROR_ORDER;
      S.invp[j - 1] = i;
      S.perm[i] = j - 1;
    }
  } else {
    if (ordering == GSLS_ORDER_NATURAL)
      std::iota(S.perm.begin(), S.perm.end(), 0);
    else if (ordering == GSLS_ORDER_AMD)
      order_amd(n, aptr, arow, S.perm);
    else
      order_nested_dissection(n, aptr, arow, S.perm);
    for (int i = 0; i < n; ++i) S.invp[S.perm[i]] = i;
  }
  lap("ordering");

  // ---- tree, counts, supernodes -----------------------------------------------------------------
  std::vector<int> parent, cc;
  elimination_tree(n, aptr, arow, S.perm, S.invp, parent);
  postorder(n, aptr, S.perm, S.invp, parent, S.realn);
  if (S.realn != n) flag = GSLS_WARNING_ANAL_SINGULAR;
  column_counts(n, aptr, arow, S.perm, S.invp, parent, cc);
  lap("tree, postorder, counts");
  Supernodes sn;
  std::vector<char> forcecol;
  if (force_var) {
    forcecol.assign(n + 1, 0);
    for (int p = 0; p < n; ++p) forcecol[p] = force_var[S.invp[p]] ? 1 : 0;
  }
  relaxed_supernodes(n, S.realn, parent, cc, nemin, ordering != GSLS_ORDER_USER, force_var ? &forcecol : nullptr, sn);

  // final pivot order = supernode renumbering applied on top of the postorder
  {
    std::vector<int> old_invp(S.invp);
    for (int i = 0; i < n; ++i) S.invp[sn.sperm[i]] = old_invp[i];
    for (int i = 0; i < n; ++i) S.perm[S.invp[i]] = i;
  }
  S.nnodes = sn.nnodes;
  S.sptr = sn.sptr;
  S.sparent = sn.sparent;
  const int nn = S.nnodes;

  lap("supernodes");
  // ---- row lists: own pivots + what the children pass up + new rows from A, then sorted ---------
  S.rptr.assign(nn + 1, 0);
  for (int s = 0; s < nn; ++s) S.rptr[s + 1] = S.rptr[s] + sn.scc[s];
  S.rlist.assign(S.rptr[nn], 0);
  S.cptr.assign(nn + 2, 0);
  for (int s = 0; s < nn; ++s) S.cptr[S.sparent[s] + 1]++;
  for (int s = 0; s <= nn; ++s) S.cptr[s + 1] += S.cptr[s];
  S.clist.resize(nn);
  {
    std::vector<int> fill(S.cptr.begin(), S.cptr.end() - 1);
    for (int s = 0; s < nn; ++s) S.clist[fill[S.sparent[s]]++] = s;
  }
  {
    std::vector<int> seen(n, -1);
    for (int s = 0; s < nn; ++s) {
      int64_t w = S.rptr[s];
      for (int p = S.sptr[s]; p < S.sptr[s + 1]; ++p) {
        seen[p] = s;
        S.rlist[w++] = p;
      }
      for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) {
        int c = S.clist[ci];
        for (int64_t k = S.rptr[c]; k < S.rptr[c + 1]; ++k) {
          int j = S.rlist[k];
          if (j < S.sptr[s] || seen[j] == s) continue;
          seen[j] = s;
          S.rlist[w++] = j;
        }
      }
      for (int p = S.sptr[s]; p < S.sptr[s + 1]; ++p) {
        int c = S.invp[p];
        for (int64_t k = aptr[c]; k < aptr[c + 1]; ++k) {
          int j = S.perm[arow[k]];
          if (j < p || seen[j] == s) continue;
          seen[j] = s;
          S.rlist[w++] = j;
        }
      }
      if (w != S.rptr[s + 1]) return GSLS_ERROR_UNKNOWN;  // column counts and pattern disagree
      std::sort(S.rlist.begin() + S.rptr[s], S.rlist.begin() + S.rptr[s + 1]);
    }
  }

  // ---- statistics (core_analyse.f90:862-902) ----------------------------------------------------
  S.num_factor = 0;
  S.num_flops = 0;
  for (int s = 0; s < nn; ++s) {
    int64_t ne = S.ncol(s), m = S.nrow(s) - ne;
    S.num_factor += ne * (ne + 1) / 2 + ne * m;
    for (int64_t j = 1; j <= ne; ++j) S.num_flops += (m + j) * (m + j);
  }

  // ---- user-visible order: 0 for variables that are never eliminated ---------------------------
  if (order) {
    for (int i = 0; i < n; ++i) order[i] = S.perm[i] + 1;
    for (int p = S.sptr[nn]; p < n; ++p) order[S.invp[p]] = 0;
  }

  // ---- A -> front map (anal.f90:1129-1231): entries of row `col` left of the diagonal first (in
  //      column order), then the entries of column `col` itself ------------------------------------
  {
    const int64_t nz = ptr[n] - 1;
    std::vector<int64_t> tptr(n + 1, 0), origin(nz);
    std::vector<int> tcol(nz);
    for (int j = 0; j < n; ++j)
      for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k)
        if (row[k] - 1 != j) tptr[row[k]]++;
    for (int j = 0; j < n; ++j) tptr[j + 1] += tptr[j];
    {
      std::vector<int64_t> fill(tptr.begin(), tptr.end() - 1);
      for (int j = 0; j < n; ++j)
        for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k) {
          int i = row[k] - 1;
          if (i == j) continue;
          tcol[fill[i]] = j;
          origin[fill[i]++] = k;
        }
    }
    S.nptr.assign(nn + 1, 0);
    S.nlist.clear();
    S.nlist.reserve(2 * nz);
    std::vector<int> local(n, 0);
    for (int s = 0; s < nn; ++s) {
      S.nptr[s] = int64_t(S.nlist.size() / 2);
      const int64_t m = S.nrow(s);
      for (int64_t k = S.rptr[s]; k < S.rptr[s + 1]; ++k) local[S.rlist[k]] = int(k - S.rptr[s]);
      for (int p = S.sptr[s]; p < S.sptr[s + 1]; ++p) {
        int c = S.invp[p];
        for (int64_t k = tptr[c]; k < tptr[c + 1]; ++k) {
          int r = S.perm[tcol[k]];
          if (r < p) continue;
          S.nlist.push_back(origin[k]);
          S.nlist.push_back(int64_t(p - S.sptr[s]) * m + local[r]);
        }
      }
      for (int p = S.sptr[s]; p < S.sptr[s + 1]; ++p) {
        int c = S.invp[p];
        for (int64_t k = ptr[c] - 1; k < ptr[c + 1] - 1; ++k) {
          int r = S.perm[row[k] - 1];
          if (r < p) continue;
          S.nlist.push_back(k);
          S.nlist.push_back(int64_t(p - S.sptr[s]) * m + local[r]);
        }
      }
    }
    S.nptr[nn] = int64_t(S.nlist.size() / 2);
  }

  // ---- tree statistics (anal.f90:1093-1106) -----------------------------------------------------
  {
    std::vector<int> depth(nn + 1, 0);
    S.maxfront = 0;
    S.maxdepth = 0;
    for (int s = nn - 1; s >= 0; --s) {
      depth[s] = depth[S.sparent[s]] + 1;
      S.maxfront = std::max(S.maxfront, S.ncol(s));
      S.maxrow = std::max(S.maxrow, S.nrow(s));
      S.maxdepth = std::max(S.maxdepth, depth[s]);
    }
  }

  // ---- level-set schedule + child->parent row maps + storage layout (ours) ----------------------
  S.level.assign(nn + 1, 0);
  for (int s = 0; s < nn; ++s) {
    int p = S.sparent[s];
    S.level[p] = std::max(S.level[p], S.level[s] + 1);
  }
  S.nlevels = 0;
  for (int s = 0; s < nn; ++s) S.nlevels = std::max(S.nlevels, S.level[s] + 1);
  S.lvlptr.assign(S.nlevels + 1, 0);
  for (int s = 0; s < nn; ++s) S.lvlptr[S.level[s] + 1]++;
  for (int l = 0; l < S.nlevels; ++l) S.lvlptr[l + 1] += S.lvlptr[l];
  S.lvlnodes.resize(nn);
  {
    std::vector<int> fill(S.lvlptr.begin(), S.lvlptr.end() - 1);
    for (int s = 0; s < nn; ++s) S.lvlnodes[fill[S.level[s]]++] = s;
  }
  S.cmapptr.assign(nn + 1, 0);
  for (int s = 0; s < nn; ++s) S.cmapptr[s + 1] = S.cmapptr[s] + (S.nrow(s) - S.ncol(s));
  S.cmap.assign(S.cmapptr[nn], -1);
  for (int s = 0; s < nn; ++s) {
    int p = S.sparent[s];
    if (p >= nn) continue;
    int64_t a = S.rptr[s] + S.ncol(s), ae = S.rptr[s + 1];
    int64_t b = S.rptr[p], be = S.rptr[p + 1];
    int64_t w = S.cmapptr[s];
    for (; a < ae; ++a) {
      while (b < be && S.rlist[b] < S.rlist[a]) ++b;
      if (b == be || S.rlist[b] != S.rlist[a]) return GSLS_ERROR_UNKNOWN;  // child row not in parent
      S.cmap[w++] = int(b - S.rptr[p]);
    }
  }
  S.loff.assign(nn + 1, 0);
  S.ldl.assign(nn, 0);
  for (int s = 0; s < nn; ++s) {
    int64_t m = S.nrow(s), ne = S.ncol(s);
    S.ldl[s] = align_ld(int(m));
    S.loff[s + 1] = S.loff[s] + int64_t(S.ldl[s]) * ne;
  }
  layout_contrib_auto(S);
  lap("row lists, maps, layout");
  return flag;
}

// Contribution blocks ((m-n)^2 doubles per front).  The reference allocates them per front and frees them as the
// parent assembles them (assemble.hxx:347-437, BuddyAllocator.hxx); the static schedule here does the same at plan
// time: levels run in order, a block is allocated when its front's level starts and released once its parent's level
// has pulled it, first fit.  Without that the arena is the sum over ALL fronts -- hundreds of GB for a 3-D problem whose
// factor itself is 30 GB.
void layout_contrib_auto(Symbolic& S) {
  // small trees: one block per front and one clear per factorization (no per-level clears on the launch path)
  int64_t total = 0;
  for (int s = 0; s < S.nnodes; ++s) total += int64_t(S.nrow(s) - S.ncol(s)) * (S.nrow(s) - S.ncol(s));
  layout_contrib(S, total > (int64_t(1) << 27));      // reuse above 1 GiB
}

void layout_contrib(Symbolic& S, bool reuse) {
  const int nn = S.nnodes;
  S.coff.assign(nn + 1, 0);
  S.czptr.assign(S.nlevels + 1, 0);
  S.czoff.clear();
  S.czlen.clear();
  auto clen = [&](int s) { const int64_t cm = S.nrow(s) - S.ncol(s); return cm * cm; };
  if (!reuse) {
    for (int s = 0; s < nn; ++s) S.coff[s + 1] = S.coff[s] + clen(s);
    if (S.nlevels > 0 && S.coff[nn] > 0) {
      S.czoff.push_back(0);
      S.czlen.push_back(S.coff[nn]);
    }
    for (int l = 1; l <= S.nlevels; ++l) S.czptr[l] = int(S.czoff.size());
    return;
  }
  std::vector<std::pair<int64_t, int64_t>> freel;     // (offset, length), sorted by offset, coalesced
  int64_t top = 0;
  auto alloc = [&](int64_t len) -> int64_t {
    for (size_t i = 0; i < freel.size(); ++i)
      if (freel[i].second >= len) {
        const int64_t off = freel[i].first;
        if (freel[i].second == len) freel.erase(freel.begin() + i);
        else { freel[i].first += len; freel[i].second -= len; }
        return off;
      }
    if (!freel.empty() && freel.back().first + freel.back().second == top) {   // grow the last free block
      const int64_t off = freel.back().first;
      top = off + len;
      freel.pop_back();
      return off;
    }
    const int64_t off = top;
    top += len;
    return off;
  };
  auto release = [&](int64_t off, int64_t len) {
    auto it = std::lower_bound(freel.begin(), freel.end(), std::make_pair(off, int64_t(0)));
    it = freel.insert(it, std::make_pair(off, len));
    if (it + 1 != freel.end() && it->first + it->second == (it + 1)->first) {
      it->second += (it + 1)->second;
      freel.erase(it + 1);
    }
    if (it != freel.begin() && (it - 1)->first + (it - 1)->second == it->first) {
      (it - 1)->second += it->second;
      freel.erase(it);
    }
  };
  std::vector<std::pair<int64_t, int64_t>> zr;
  for (int l = 0; l < S.nlevels; ++l) {
    zr.clear();
    for (int i = S.lvlptr[l]; i < S.lvlptr[l + 1]; ++i) {
      const int s = S.lvlnodes[i];
      const int64_t len = clen(s);
      if (len == 0) continue;
      S.coff[s] = alloc(len);
      zr.emplace_back(S.coff[s], len);
    }
    std::sort(zr.begin(), zr.end());
    for (size_t i = 0; i < zr.size(); ++i) {          // coalesced ranges to zero before this level runs
      if (!S.czoff.empty() && int(S.czoff.size()) > S.czptr[l] && S.czoff.back() + S.czlen.back() == zr[i].first)
        S.czlen.back() += zr[i].second;
      else {
        S.czoff.push_back(zr[i].first);
        S.czlen.push_back(zr[i].second);
      }
    }
    S.czptr[l + 1] = int(S.czoff.size());
    for (int i = S.lvlptr[l]; i < S.lvlptr[l + 1]; ++i) {
      const int s = S.lvlnodes[i];
      for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) {
        const int c = S.clist[ci];
        if (clen(c) > 0) release(S.coff[c], clen(c));
      }
    }
  }
  S.coff[nn] = top;
}

// ---------------------------------------------------------------------------------------------------
// Subtree partition for multi-GPU runs.  Same idea as the reference's find_subtree_partition
// (src/ssids/anal.f90:284-459): start from the roots, keep splitting the heaviest subtree at its root
// until there are enough pieces to balance, deal them out by decreasing flops (longest processing time
// first); the split-off ancestors form the top part, which one owner (rank 0) factors after it has
// received the contribution blocks of the cut.
void shard_tree(Symbolic& S, int nranks) {
  const int nn = S.nnodes;
  S.nranks = std::max(1, nranks);
  S.owner.assign(nn, 0);
  S.cutroots.clear();
  for (int s = 0; s < nn; ++s)         // (a per-rank layout of an earlier sharding may be in place: shard_layout)
    S.loff[s + 1] = S.loff[s] + int64_t(S.ldl[s]) * S.ncol(s);
  if (S.nranks > 1) layout_contrib(S, false);      // the cut roots' blocks outlive their level: no reuse across ranks
  else layout_contrib_auto(S);
  if (S.nranks <= 1 || nn == 0) return;
  std::vector<double> w(nn, 0.0), W(nn, 0.0);
  for (int s = 0; s < nn; ++s) {
    const double m = S.nrow(s), ne = S.ncol(s);
    for (int j = 0; j < int(ne); ++j) w[s] += (m - j) * (m - j);
    W[s] += w[s];
    if (S.sparent[s] < nn) W[S.sparent[s]] += W[s];
  }
  std::vector<int> cut;
  std::vector<char> top(nn, 0);
  for (int s = 0; s < nn; ++s)
    if (S.sparent[s] >= nn) cut.push_back(s);
  // split the heaviest piece at its root until there are 4 pieces per rank and none is heavier than
  // half a rank's fair share (cf. the 1.2 imbalance target of anal.f90:284-459)
  double total = 0.0;
  for (int c : cut) total += W[c];
  const size_t want = size_t(4 * S.nranks);
  const double heavy = total / (2.0 * S.nranks);
  for (;;) {
    int best = -1;
    for (size_t i = 0; i < cut.size(); ++i)
      if (S.cptr[cut[i] + 1] > S.cptr[cut[i]] && (best < 0 || W[cut[i]] > W[cut[best]])) best = int(i);
    if (best < 0) break;
    if (cut.size() >= want && W[cut[best]] <= heavy) break;
    const int s = cut[best];
    cut.erase(cut.begin() + best);
    top[s] = 1;
    for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) cut.push_back(S.clist[ci]);
  }
  std::sort(cut.begin(), cut.end(), [&](int a, int b) { return W[a] != W[b] ? W[a] > W[b] : a < b; });
  // the top part runs after the exchange, so it does not count against rank 0's share
  std::vector<double> load(S.nranks, 0.0);
  std::vector<int> rank_of_root(nn, -1);
  for (int c : cut) {
    int r = 0;
    for (int k = 1; k < S.nranks; ++k)
      if (load[k] < load[r]) r = k;
    load[r] += W[c];
    rank_of_root[c] = r;
  }
  // ownership flows down from the cut roots; the top part is -1
  for (int s = nn - 1; s >= 0; --s) {
    if (top[s]) S.owner[s] = -1;
    else if (rank_of_root[s] >= 0) S.owner[s] = rank_of_root[s];
    else S.owner[s] = S.owner[S.sparent[s]];
  }
  std::sort(cut.begin(), cut.end());
  for (int c : cut)
    if (S.sparent[c] < nn) S.cutroots.push_back(c);   // roots of the whole tree send nothing
}

// Per-rank memory of a sharded run (round 3).  shard_tree leaves the single-device layout in place: every rank would
// allocate ALL of L and a side-by-side arena for every front's contribution block -- O(total) per rank, > 500 GB of
// arena for BASELINE configs[4] against 288 GB of HBM.  This lays out, for ONE rank, only what that rank touches:
//   * factors: the fronts it owns (rank 0: and the top part); the others get length 0;
//   * contribution blocks: two regions, each reused by lifetime exactly as layout_contrib does on one device, in the
//     order the rank really runs its fronts --
//       region 1: the fronts of its subtrees, level by level (phase 1).  A cut root's block is never released there
//                 (its parent is in the top part): it is what the rank sends;
//       region 2 (rank 0 only): first, pinned, the blocks of the OTHER ranks' cut roots (they arrive through the
//                 exchange; a block is released when the top front that pulls it has run), then the top part's
//                 own blocks level by level (phase 2).
//     The ranges to clear before a level are kept per (phase, level): czptr has 2 * nlevels + 1 entries, phase 2's
//     levels behind phase 1's; received blocks are never cleared.
// Every front's offsets are only meaningful on the rank that owns it; the symbolic data proper (sptr, rlist, maps,
// statistics) stays identical on all ranks.
void shard_layout(Symbolic& S, int rank) {
  const int nn = S.nnodes, L = S.nlevels;
  if (S.nranks <= 1 || nn == 0 || int(S.owner.size()) != nn) return;
  auto top = [&](int s) { return S.owner[s] < 0; };
  auto mine = [&](int s) { return S.owner[s] == rank || (top(s) && rank == 0); };
  auto clen = [&](int s) { const int64_t cm = S.nrow(s) - S.ncol(s); return cm * cm; };
  S.loff.assign(nn + 1, 0);
  for (int s = 0; s < nn; ++s)
    S.loff[s + 1] = S.loff[s] + (mine(s) ? int64_t(S.ldl[s]) * S.ncol(s) : 0);
  S.coff.assign(nn + 1, 0);
  S.czptr.assign(2 * L + 1, 0);
  S.czoff.clear();
  S.czlen.clear();
  std::vector<std::pair<int64_t, int64_t>> freel;     // (offset, length), sorted by offset, coalesced
  int64_t topoff = 0;
  auto alloc = [&](int64_t len) -> int64_t {
    for (size_t i = 0; i < freel.size(); ++i)
      if (freel[i].second >= len) {
        const int64_t off = freel[i].first;
        if (freel[i].second == len) freel.erase(freel.begin() + i);
        else { freel[i].first += len; freel[i].second -= len; }
        return off;
      }
    if (!freel.empty() && freel.back().first + freel.back().second == topoff) {
      const int64_t off = freel.back().first;
      topoff = off + len;
      freel.pop_back();
      return off;
    }
    const int64_t off = topoff;
    topoff += len;
    return off;
  };
  auto release = [&](int64_t off, int64_t len) {
    auto it = std::lower_bound(freel.begin(), freel.end(), std::make_pair(off, int64_t(0)));
    it = freel.insert(it, std::make_pair(off, len));
    if (it + 1 != freel.end() && it->first + it->second == (it + 1)->first) {
      it->second += (it + 1)->second;
      freel.erase(it + 1);
    }
    if (it != freel.begin() && (it - 1)->first + (it - 1)->second == it->first) {
      (it - 1)->second += it->second;
      freel.erase(it);
    }
  };
  std::vector<std::pair<int64_t, int64_t>> zr;
  auto run_phase = [&](int phase) {
    for (int l = 0; l < L; ++l) {
      const int v = (phase - 1) * L + l;
      zr.clear();
      for (int i = S.lvlptr[l]; i < S.lvlptr[l + 1]; ++i) {
        const int s = S.lvlnodes[i];
        if (!mine(s) || (phase == 1) == top(s)) continue;
        const int64_t len = clen(s);
        if (len == 0) continue;
        S.coff[s] = alloc(len);
        zr.emplace_back(S.coff[s], len);
      }
      std::sort(zr.begin(), zr.end());
      for (size_t i = 0; i < zr.size(); ++i) {
        if (!S.czoff.empty() && int(S.czoff.size()) > S.czptr[v] && S.czoff.back() + S.czlen.back() == zr[i].first)
          S.czlen.back() += zr[i].second;
        else {
          S.czoff.push_back(zr[i].first);
          S.czlen.push_back(zr[i].second);
        }
      }
      S.czptr[v + 1] = int(S.czoff.size());
      for (int i = S.lvlptr[l]; i < S.lvlptr[l + 1]; ++i) {
        const int s = S.lvlnodes[i];
        if (!mine(s) || (phase == 1) == top(s)) continue;
        for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) {
          const int c = S.clist[ci];
          if (clen(c) > 0) release(S.coff[c], clen(c));     // (a top front's children include received cut roots)
        }
      }
    }
  };
  run_phase(1);
  // region 2 starts behind everything phase 1 ever used: this rank's own cut roots stay where phase 1 left them
  freel.clear();
  if (rank == 0) {
    for (int c : S.cutroots)
      if (S.owner[c] != 0 && clen(c) > 0) S.coff[c] = alloc(clen(c));      // pinned until their parent has run
    run_phase(2);
  } else {
    for (int v = L; v < 2 * L; ++v) S.czptr[v + 1] = int(S.czoff.size());
  }
  S.coff[nn] = topoff;
}

}  // namespace gsls
