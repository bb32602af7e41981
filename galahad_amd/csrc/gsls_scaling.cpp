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
//  * scaling 2: the auction algorithm, restated step by step (scaling.f90:1351-1489 core, :1504-1609 pre/post).
//  * scaling 4: infinity-norm equilibration (Knight-Ruiz iteration, scaling.f90:480-521).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
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
int hungarian_scale_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, bool scale_if_singular,
                        double* scaling) {
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
    return 0;
  }
  if (!scale_if_singular) {
    for (int i = 0; i < n; ++i) scaling[i] = 1.0;
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

// scaling.f90:1351-1489 (core) and :1504-1609 (pre/post-processing), defaults of type auction_options (:33-38)
int auction_scale_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, double* scaling) {
  const int max_iterations = 30000;
  const int max_unchanged[3] = {10, 100, 100};
  const float min_proportion[3] = {0.90f, 0.0f, 0.0f};
  const float eps_initial = 0.01f;
  FullMatrix A = expand_log(n, ptr, row, val);
  std::vector<double> cmax(n, 0.0);
  double maxentry = -RINF;
  for (int j = 0; j < n; ++j) {
    if (A.ptr[j] == A.ptr[j + 1]) continue;
    double m = -RINF;
    for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k) m = std::max(m, A.val[k]);
    cmax[j] = m;
    for (int64_t k = A.ptr[j]; k < A.ptr[j + 1]; ++k) {
      A.val[k] = m - A.val[k];
      maxentry = std::max(maxentry, A.val[k]);
    }
  }
  if (A.ptr[n] == 0) {
    for (int i = 0; i < n; ++i) scaling[i] = 1.0;
    return 0;
  }
  maxentry = 2 * maxentry + 1;       // prefers matchings of high cardinality
  for (auto& x : A.val) x = maxentry - x;
  std::vector<double> dualu(n, 0.0), dualv(n);
  for (int j = 0; j < n; ++j) dualv[j] = -cmax[j];
  std::vector<int32_t> match(n, 0), owner(n, 0), next(n);      // 1-based partners, 0 = none, -1 = ineligible
  int unmatched = n, prev = -1, nunchanged = 0, tail = n;
  for (int i = 0; i < n; ++i) next[i] = i;
  double eps = eps_initial;
  for (int itr = 1; itr <= max_iterations; ++itr) {
    if (unmatched == 0) break;
    if (unmatched != prev) nunchanged = 0;
    prev = unmatched;
    ++nunchanged;
    bool stop = false;
    for (int t = 0; t < 3; ++t)
      if (nunchanged >= max_unchanged[t] && float(n - unmatched) / float(n) >= min_proportion[t]) stop = true;
    if (stop) break;
    eps = std::min(1.0, eps + 1.0 / (n + 1));
    int insert = 0;
    for (int cp = 0; cp < tail; ++cp) {
      const int col = next[cp];
      if (match[col] != 0) continue;
      if (A.ptr[col] == A.ptr[col + 1]) continue;
      int64_t k = A.ptr[col];
      int bestr = A.row[k];
      double bestu = A.val[k] - dualu[bestr];
      double bestv = -RINF;
      for (k = A.ptr[col] + 1; k < A.ptr[col + 1]; ++k) {
        const double uu = A.val[k] - dualu[A.row[k]];
        if (uu > bestu) {
          bestv = bestu;
          bestr = A.row[k];
          bestu = uu;
        } else if (uu > bestv) {
          bestv = uu;
        }
      }
      if (bestv == -RINF) bestv = 0.0;
      if (bestu > 0) {
        dualu[bestr] += bestu - bestv + eps;
        dualv[col] = bestv - eps;
        match[col] = bestr + 1;
        --unmatched;
        const int kcol = owner[bestr];
        owner[bestr] = col + 1;
        if (kcol != 0) {
          match[kcol - 1] = 0;
          ++unmatched;
          next[insert++] = kcol - 1;
        }
      } else {
        match[col] = -1;
        --unmatched;
      }
    }
    tail = insert;
  }
  // undo the pre-processing (the magnitude adjustment of match_postproc cancels in the symmetric average)
  for (int i = 0; i < n; ++i) {
    const double r = -dualu[i] + maxentry;
    const double c = -dualv[i] - cmax[i];
    scaling[i] = std::exp((r + c) / 2);
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
