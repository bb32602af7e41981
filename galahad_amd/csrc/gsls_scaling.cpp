// Matching-based and norm-equilibration scalings of a symmetric matrix, computed on the host from the values handed
// to gsls_factor* when gsls_options.scaling = 1 / 2 / 4 -- what SSIDS does inside ssids_factor
// (src/ssids/ssids.f90:900-1040: options%scaling = 1 hungarian_scale_sym, 2 auction_scale_sym, 4 equilib_scale_sym,
// all from src/spral/scaling.f90).  GALAHAD's SLS reaches 1..3 through control%scaling = -1..-3 (sls.f90:1405-1413).
//
// Integer / graph work on the host, as in the reference; the resulting vector is applied on the device by the
// A -> L scatter (S A S) and by the permutation kernels of the solve.
//
//  * scaling 1: minimum-sum assignment on c_ij = max_i log|a_ij| - log|a_ij| with its dual variables (u, v), found by
//    shortest augmenting paths (the Duff & Koster scheme scaling.f90:938-1194 implements); s_i = exp((u_i + v_i - cmax_i)/2)
//    (scaling.f90:134-170, 597-693).  Any optimal dual pair gives |s_i a_ij s_j| <= 1 with equality on the matching;
//    the pair is not unique, so the vector can differ from the reference's in the unconstrained directions --
//    solutions agree, scale vectors need not.  Structurally singular matrices: error, or with `scale_if_singular` the
//    reference's recipe (match the nonsingular part again, Duff-Pralet for the rest, scaling.f90:695-800).
//  * scaling 2: an auction for the same assignment problem -- own design: synchronous bidding rounds with epsilon-scaling
//    (see auction_scale_sym below); the reference's auction_scale_sym (scaling.f90:1351-1609) is a sequential one.
//  * scaling 4: infinity-norm equilibration (Knight-Ruiz iteration, scaling.f90:480-521).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <queue>
#include <vector>

#include "gsls_internal.hpp"

namespace gsls {

namespace {

struct FullMatrix {              // both triangles by columns, explicit zeros dropped, val = log|a|
  std::vector<int64_t> ptr;
  std::vector<int32_t> row;
  std::vector<double> val;
};

// lower triangle by columns (0-based) -> full symmetric pattern (half_to_full, matrix_util.f90); within a column the
// entries of the upper part (rows < j, ascending) come first, then the lower part in the caller's order
FullMatrix expand_log(int n, const int64_t* ptr, const int32_t* row, const double* val) {
  FullMatrix A;
  A.ptr.assign(n + 1, 0);
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) {
      if (val[k] == 0.0) continue;
      A.ptr[j + 1]++;
      if (row[k] != j) A.ptr[row[k] + 1]++;
    }
  for (int j = 0; j < n; ++j) A.ptr[j + 1] += A.ptr[j];
  A.row.resize(A.ptr[n]);
  A.val.resize(A.ptr[n]);
  std::vector<int64_t> fill(A.ptr.begin(), A.ptr.end() - 1);
  for (int j = 0; j < n; ++j)          // upper parts first: entry (j, i) of column i for every (i, j) below the diagonal
    for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) {
      if (val[k] == 0.0 || row[k] == j) continue;
      const int i = row[k];
      A.row[fill[i]] = j;
      A.val[fill[i]++] = std::log(std::fabs(val[k]));
    }
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) {
      if (val[k] == 0.0) continue;
      A.row[fill[j]] = row[k];
      A.val[fill[j]++] = std::log(std::fabs(val[k]));
    }
  return A;
}

constexpr double RINF = std::numeric_limits<double>::max();

// minimum-sum perfect matching of an n x n sparse cost matrix (by columns, costs >= 0) with dual variables:
//   u_i + v_j <= c_ij everywhere, equality on the matching.   rowmatch[i] = column or -1.  Returns the cardinality.
int min_sum_matching(int n, const std::vector<int64_t>& ptr, const std::vector<int32_t>& row,
                     const std::vector<double>& c, std::vector<int32_t>& rowmatch, std::vector<double>& u,
                     std::vector<double>& v) {
  std::vector<int32_t> colmatch(n, -1);
  rowmatch.assign(n, -1);
  u.assign(n, RINF);
  v.assign(n, 0.0);
  // start: u_i = smallest entry of row i, v_j = smallest reduced cost of column j, tight edges matched greedily
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) u[row[k]] = std::min(u[row[k]], c[k]);
  for (int i = 0; i < n; ++i)
    if (u[i] == RINF) u[i] = 0.0;            // empty row
  int num = 0;
  for (int j = 0; j < n; ++j) {
    double best = RINF;
    for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) best = std::min(best, c[k] - u[row[k]]);
    v[j] = (best == RINF) ? 0.0 : best;
    for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k)
      if (rowmatch[row[k]] < 0 && c[k] - u[row[k]] == v[j]) {
        rowmatch[row[k]] = j;
        colmatch[j] = row[k];
        ++num;
        break;
      }
  }
  // shortest augmenting path from every unmatched column (Dijkstra on the reduced costs)
  std::vector<double> dist(n, RINF);
  std::vector<int32_t> pred(n, -1);          // pred[i]: the column row i was reached from
  std::vector<char> done(n, 0);
  std::vector<int32_t> touched, tree;
  typedef std::pair<double, int32_t> Item;
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
  for (int j0 = 0; j0 < n; ++j0) {
    if (colmatch[j0] >= 0 || ptr[j0] == ptr[j0 + 1]) continue;
    touched.clear();
    tree.clear();
    while (!heap.empty()) heap.pop();
    double best = RINF;
    int isp = -1;
    int j = j0;
    double dj = 0.0;
    while (true) {
      for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) {
        const int i = row[k];
        if (done[i]) continue;
        const double nd = dj + (c[k] - u[i] - v[j]);
        if (nd >= best || nd >= dist[i]) continue;
        if (dist[i] == RINF) touched.push_back(i);
        dist[i] = nd;
        pred[i] = j;
        if (rowmatch[i] < 0) {
          best = nd;
          isp = i;
        } else {
          heap.push(Item(nd, i));
        }
      }
      int inext = -1;
      while (!heap.empty()) {
        const Item it = heap.top();
        if (it.first >= best) break;
        heap.pop();
        if (done[it.second] || it.first > dist[it.second]) continue;     // stale entry
        inext = it.second;
        break;
      }
      if (inext < 0) break;
      done[inext] = 1;
      tree.push_back(inext);
      j = rowmatch[inext];
      dj = dist[inext];
    }
    if (isp >= 0) {
      // duals: the rows of the tree and their columns move by (dist - best), the root column by best
      for (int i : tree) {
        const double delta = dist[i] - best;
        u[i] += delta;
        v[rowmatch[i]] -= delta;
      }
      v[j0] += best;
      // augment along pred
      int i = isp;
      while (true) {
        const int jc = pred[i];
        const int inext = colmatch[jc];
        rowmatch[i] = jc;
        colmatch[jc] = i;
        if (jc == j0) break;
        i = inext;
      }
      ++num;
    }
    for (int i : touched) {
      dist[i] = RINF;
      pred[i] = -1;
      done[i] = 0;
    }
  }
  return num;
}

}  // namespace

// ptr/row/val: lower triangle by columns, 0-based.  Returns 0, 1 (singular, scaled anyway) or -2 (singular, identity).
static int hungarian_core(int n, const int64_t* ptr, const int32_t* row, const double* val, bool scale_if_singular,
                          double* scaling, std::vector<int32_t>* match_out);

int hungarian_scale_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, bool scale_if_singular,
                        double* scaling) {
  return hungarian_core(n, ptr, row, val, scale_if_singular, scaling, nullptr);
}

// match_out (optional): the matching the scaling comes from, match[i] = the column matched to row i, -1 for the rows a
// structurally singular matrix leaves out (then the matching of the non-singular part)
static int hungarian_core(int n, const int64_t* ptr, const int32_t* row, const double* val, bool scale_if_singular,
                          double* scaling, std::vector<int32_t>* match_out) {
  FullMatrix A = expand_log(n, ptr, row, val);
  std::vector<double> cmax(n, 0.0);
  for (int j = 0; j < n; ++j) {
    if (A.ptr[j] == A.ptr[j + 1]) continue;
    double m = -RINF;
    for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k) m = std::max(m, A.val[k]);
    cmax[j] = m;
    for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k) A.val[k] = m - A.val[k];
  }
  std::vector<int32_t> match;
  std::vector<double> u, v;
  const int matched = min_sum_matching(n, A.ptr, A.row, A.val, match, u, v);
  if (matched == n) {
    for (int i = 0; i < n; ++i) scaling[i] = std::exp((u[i] + v[i] - cmax[i]) / 2);
    if (match_out) *match_out = match;
    return 0;
  }
  if (!scale_if_singular) {
    for (int i = 0; i < n; ++i) scaling[i] = 1.0;
    if (match_out) *match_out = match;
    return -2;
  }
  // structurally singular: the variables whose row is unmatched leave (row and column); the rest is matched again
  std::vector<int32_t> o2n(n, -1), n2o;
  for (int i = 0; i < n; ++i)
    if (match[i] >= 0) {
      o2n[i] = int32_t(n2o.size());
      n2o.push_back(i);
    }
  const int nn = int(n2o.size());
  FullMatrix B;
  B.ptr.assign(nn + 1, 0);
  for (int jn = 0; jn < nn; ++jn) {
    const int j = n2o[jn];
    for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k)
      if (o2n[A.row[k]] >= 0) {
        B.row.push_back(o2n[A.row[k]]);
        B.val.push_back(A.val[k]);
      }
    B.ptr[jn + 1] = int64_t(B.row.size());
  }
  std::vector<int32_t> m2;
  std::vector<double> u2, v2;
  (void)min_sum_matching(nn, B.ptr, B.row, B.val, m2, u2, v2);
  if (match_out) {
    match_out->assign(n, -1);
    for (int in = 0; in < nn; ++in)
      if (m2[in] >= 0) (*match_out)[n2o[in]] = n2o[m2[in]];
  }
  const double NONE = -RINF;
  std::vector<double> rs(n, NONE);
  for (int i = 0; i < n; ++i)
    if (o2n[i] >= 0) rs[i] = (u2[o2n[i]] + v2[o2n[i]] - cmax[i]) / 2;
  // Duff & Pralet: for i outside the matched set I, s_i = 1 / max_{k in I} |a_ik s_k| (1 if there is none)
  std::vector<double> cs(rs);
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) {
      if (val[k] == 0.0) continue;
      const int i = row[k];
      const double la = std::log(std::fabs(val[k]));
      if (cs[j] == NONE && cs[i] != NONE) rs[j] = std::max(rs[j], la + rs[i]);
      if (cs[i] == NONE && cs[j] != NONE) rs[i] = std::max(rs[i], la + rs[j]);
    }
  for (int i = 0; i < n; ++i) {
    if (cs[i] != NONE) continue;
    rs[i] = (rs[i] == NONE) ? 0.0 : -rs[i];
  }
  for (int i = 0; i < n; ++i) scaling[i] = std::exp(rs[i]);
  return 1;
}

// Matching-based ordering (SSIDS ordering = 2: match_order_metis, src/spral/match_order.f90:51-208, after Duff & Pralet;
// reached from ssids_analyse when the caller passes the values, ssids.f90:305-320).  The maximum-product matching of the
// scaling pairs every variable with the column that holds "its" large entry; a matched pair (i, j), i /= j, is a 2x2 pivot
// [a_ii a_ij; a_ij a_jj] that threshold pivoting will accept, PROVIDED i and j are neighbours in the elimination order.
// So:  1. matching + scaling (hungarian_core);
//      2. the cycles of the matching permutation are cut into pairs and singletons: walking a cycle
//         i -> match[i] -> match[match[i]] ..., consecutive elements are paired (every such pair is an entry of the
//         matrix), an odd cycle leaves one singleton (mo_split, match_order.f90:220-330);
//      3. the graph is compressed -- a pair becomes ONE vertex with the union of both adjacencies -- and ordered with the
//         handle's fill-reducing ordering (nested dissection or AMD here, METIS in the reference);
//      4. the order is expanded: the two variables of a pair take consecutive positions.
// order[i] = position of variable i (1-based).  Returns what hungarian_scale_sym returns (0, 1 singular; the ordering is
// produced either way, rows without a partner are singletons).
int match_order_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, int ordering, int32_t* order,
                    double* scaling, int32_t* npairs) {
  std::vector<int32_t> match;
  const int sf = hungarian_core(n, ptr, row, val, true, scaling, &match);
  // ---- 2. pairs ----
  std::vector<int32_t> partner(n, -1);
  std::vector<char> seen(n, 0);
  int pairs = 0;
  for (int i0 = 0; i0 < n; ++i0) {
    int cur = i0;
    while (cur >= 0 && !seen[cur]) {
      const int nxt = match[cur];
      seen[cur] = 1;
      if (nxt < 0 || nxt == cur || seen[nxt]) break;        // a singleton (fixed point, unmatched, or the odd one out)
      seen[nxt] = 1;
      partner[cur] = nxt;
      partner[nxt] = cur;
      ++pairs;
      cur = match[nxt];
    }
  }
  if (npairs) *npairs = pairs;
  // ---- 3. the compressed graph ----
  std::vector<int32_t> comp(n, -1), first;
  for (int i = 0; i < n; ++i) {
    if (comp[i] >= 0) continue;
    comp[i] = int32_t(first.size());
    if (partner[i] >= 0) comp[partner[i]] = comp[i];
    first.push_back(i);
  }
  const int nc = int(first.size());
  // full pattern of A (both triangles), by columns
  std::vector<int64_t> aptr(n + 1, 0);
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) {
      if (row[k] == j) continue;
      aptr[j + 1]++;
      aptr[row[k] + 1]++;
    }
  for (int j = 0; j < n; ++j) aptr[j + 1] += aptr[j];
  std::vector<int32_t> arow(static_cast<size_t>(aptr[n]));
  {
    std::vector<int64_t> fill(aptr.begin(), aptr.end() - 1);
    for (int j = 0; j < n; ++j)
      for (int64_t k = ptr[j]; k < ptr[j + 1]; ++k) {
        if (row[k] == j) continue;
        arow[fill[j]++] = row[k];
        arow[fill[row[k]]++] = j;
      }
  }
  std::vector<int64_t> cptr(nc + 1, 0);
  std::vector<int> crow;
  crow.reserve(arow.size());
  std::vector<int32_t> mark(nc, -1);
  for (int c = 0; c < nc; ++c) {
    mark[c] = c;
    const int mem[2] = {first[c], partner[first[c]]};
    for (int q = 0; q < 2; ++q) {
      if (mem[q] < 0) continue;
      for (int64_t k = aptr[mem[q]]; k < aptr[mem[q] + 1]; ++k) {
        const int d = comp[arow[k]];
        if (mark[d] == c) continue;
        mark[d] = c;
        crow.push_back(d);
      }
    }
    cptr[c + 1] = int64_t(crow.size());
  }
  std::vector<int> cperm(nc, 0);
  if (ordering == GSLS_ORDER_NATURAL) std::iota(cperm.begin(), cperm.end(), 0);
  else if (ordering == GSLS_ORDER_AMD) order_amd(nc, cptr, crow, cperm);
  else order_nested_dissection(nc, cptr, crow, cperm);
  // ---- 4. expand ----
  std::vector<int32_t> at(nc);
  for (int c = 0; c < nc; ++c) at[cperm[c]] = c;
  int pos = 0;
  for (int p = 0; p < nc; ++p) {
    const int i = first[at[p]];
    order[i] = ++pos;
    if (partner[i] >= 0) order[partner[i]] = ++pos;
  }
  return sf;
}

// scaling = 2 (SSIDS: auction_scale_sym, src/spral/scaling.f90:1351-1609 -- the same problem, not the same algorithm).
//
// The assignment problem behind the scaling: persons = columns j, objects = rows i, benefit
// b_ij = log|a_ij| - max_k log|a_kj| <= 0; prices p_i on the rows.  For ANY price vector, the column duals
// v_j = max_i (b_ij - p_i) give log|a_ij| - cmax_j - p_i - v_j <= 0 for every entry, i.e. a row scaling exp(-p_i) and a
// column scaling exp(-v_j - cmax_j) under which no entry exceeds 1 and every column attains 1; the symmetric scaling
// is their geometric mean (Duff & Pralet).  What the auction adds is prices under which the ROWS attain (almost) 1
// as well: at the end every assigned pair (i, j) is within epsilon of its column's best.
//
// Own design (round 3; the round-2 version restated the reference's sequential loop and is gone): a SYNCHRONOUS
// (Jacobi) auction with epsilon-scaling, after Bertsekas.  A round has two data-parallel steps with no ordering
// between the items of a step --
//   bid:    every unassigned column looks at its entries once: best and second-best net value b_ij - p_i, and offers
//           the best row a price raise of (best - second) + epsilon;
//   award:  every row that received offers takes the highest (ties: the lowest column index, so the result does not
//           depend on the order the offers are looked at), raises its price by it, and releases its previous column
// -- so a round is two kernel-shaped loops over independent items (the shape a device version would launch), not a
// queue walked in sequence.  Epsilon starts at a fraction of the spread of the benefits and shrinks by 8 per phase down
// to EPS_FINAL; prices are kept across phases, assignments are not (a phase re-assigns under its own epsilon).  A phase
// ends when every column that has entries is assigned, or -- matrices without a perfect matching, price wars -- when
// the number of unassigned columns has not reached a new minimum for STALL rounds.  An approximate matching is all a
// scaling needs: whatever the prices are when the last phase ends, the bounds above hold.
int auction_scale_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, double* scaling) {
  constexpr double EPS_FINAL = 0.01;     // in log units: assigned entries within 1 % of their column's best
  constexpr int STALL = 40;
  FullMatrix A = expand_log(n, ptr, row, val);
  if (A.ptr[n] == 0) {
    for (int i = 0; i < n; ++i) scaling[i] = 1.0;
    return 0;
  }
  std::vector<double> cmax(n, 0.0);
  double spread = 0.0;
  for (int j = 0; j < n; ++j) {
    if (A.ptr[j] == A.ptr[j + 1]) continue;
    double m = -RINF;
    for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k) m = std::max(m, A.val[k]);
    cmax[j] = m;
    for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k) {
      A.val[k] -= m;                                     // the benefit b_ij <= 0
      spread = std::max(spread, -A.val[k]);
    }
  }
  std::vector<double> price(n, 0.0), offer(n);
  std::vector<int32_t> holder(n), target(n), bidder(n), todo;
  std::vector<double> raise(n);
  todo.reserve(n);
  const int max_rounds = 2000 + n / 4;
  for (double eps = std::max(spread / 8, EPS_FINAL);; eps = std::max(eps / 8, EPS_FINAL)) {
    std::fill(holder.begin(), holder.end(), -1);         // row -> column that holds it
    todo.clear();
    for (int j = 0; j < n; ++j)
      if (A.ptr[j] < A.ptr[j + 1]) todo.push_back(j);
    size_t best_left = todo.size();
    int since_best = 0;
    for (int round = 0; !todo.empty() && round < max_rounds; ++round) {
      // ---- bid: independent per unassigned column -----------------------------------------------------------
      for (size_t t = 0; t < todo.size(); ++t) {
        const int j = todo[t];
        double w1 = -RINF, w2 = -RINF;
        int i1 = -1;
        for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k) {
          const double w = A.val[k] - price[A.row[k]];
          if (w > w1) { w2 = w1; w1 = w; i1 = A.row[k]; }
          else if (w > w2) w2 = w;
        }
        target[j] = i1;
        raise[j] = (w2 == -RINF) ? eps : (w1 - w2) + eps;    // a column with one entry bids the minimum
      }
      // ---- award: independent per row (here: a scan of the offers that keeps the best per row) ----------------
      for (size_t t = 0; t < todo.size(); ++t) bidder[target[todo[t]]] = -1;
      for (size_t t = 0; t < todo.size(); ++t) {
        const int j = todo[t], i = target[j];
        if (bidder[i] < 0 || raise[j] > offer[i] || (raise[j] == offer[i] && j < bidder[i])) {
          bidder[i] = j;
          offer[i] = raise[j];
        }
      }
      std::vector<int32_t> next;
      next.reserve(todo.size());
      for (size_t t = 0; t < todo.size(); ++t) {
        const int j = todo[t], i = target[j];
        if (bidder[i] != j) { next.push_back(j); continue; }   // outbid in this round: bids again
        price[i] += offer[i];
        if (holder[i] >= 0) next.push_back(holder[i]);          // the previous holder is released
        holder[i] = j;
      }
      todo.swap(next);
      if (todo.size() < best_left) { best_left = todo.size(); since_best = 0; }
      else if (++since_best >= STALL) break;
    }
    if (eps <= EPS_FINAL) break;
  }
  // row scaling exp(-p_i), column scaling exp(-v_j - cmax_j) with v_j = max_i (b_ij - p_i); symmetric: geometric mean
  for (int j = 0; j < n; ++j) {
    if (A.ptr[j] == A.ptr[j + 1]) { scaling[j] = 1.0; continue; }
    double v = -RINF;
    for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k) v = std::max(v, A.val[k] - price[A.row[k]]);
    scaling[j] = std::exp(-(price[j] + v + cmax[j]) / 2);
  }
  return 0;
}

// scaling.f90:480-521: s_i <- s_i / sqrt(max_j |s_i a_ij s_j|) until every row maximum is within tol of 1
int equilib_scale_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, double* scaling) {
  const int max_iterations = 10;
  const double tol = 1e-8f;
  std::vector<double> maxentry(n);
  for (int i = 0; i < n; ++i) scaling[i] = 1.0;
  for (int itr = 1; itr <= max_iterations; ++itr) {
    std::fill(maxentry.begin(), maxentry.end(), 0.0);
    for (int c = 0; c < n; ++c)
      for (int64_t k = ptr[c]; k < ptr[c + 1]; ++k) {
        const int r = row[k];
        const double v = std::fabs(scaling[r] * val[k] * scaling[c]);
        maxentry[r] = std::max(maxentry[r], v);
        maxentry[c] = std::max(maxentry[c], v);
      }
    for (int i = 0; i < n; ++i)
      if (maxentry[i] > 0) scaling[i] /= std::sqrt(maxentry[i]);
    double dev = 0.0;
    for (int i = 0; i < n; ++i) dev = std::max(dev, std::fabs(1 - maxentry[i]));
    if (dev < tol) break;
  }
  return 0;
}

}  // namespace gsls
