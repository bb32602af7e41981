/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement, in plain serial C, of the reference's algorithm for
 * the SLS/SSIDS factorize+solve path.  It exists so that the HIP product can be checked against an
 * independent implementation at sizes and on machines where the real reference (oracle/_ref) may
 * not be available, and as the "port" CPU baseline.  Only tests/, bench.py's cpu_baseline leg and
 * __graft_entry__.smoke() may load it; nothing under galahad_amd/ links or calls it.
 *
 * PINNING: tests/test_oracle.py checks this file against (a) the reference's own known-answer
 * systems (src/sls/slst.f90:29-51, src/sls/slss.f90:17-24) and (b) golden vectors produced by the
 * real reference built from /root/reference (tests/golden/, generator tests/golden/make_golden.py):
 * sptr/sparent/rptr/rlist/nlist/num_factor/num_flops bit-exact, solutions/inertia to tolerance.
 *
 * Each routine cites the reference code it follows (paths relative to the GALAHAD tree):
 *   analyse          src/spral/core_analyse.f90:38-1098, src/ssids/anal.f90:37-80,1129-1231
 *   assemble         src/ssids/cpu/kernels/assemble.hxx:49-79, 139-437
 *   factor (posdef)  src/ssids/cpu/factor.hxx:131-160, kernels/cholesky.cxx:32-188
 *   factor (indef)   src/ssids/cpu/factor.hxx:36-129 with pivot_method = TPP,
 *                    kernels/ldlt_tpp.cxx:20-240, kernels/calc_ld.hxx:43-118
 *   inertia          src/ssids/cpu/NumericSubtree.hxx:242-274
 *   solve            src/ssids/cpu/NumericSubtree.hxx:280-400, kernels/ldlt_tpp.cxx:242-317,
 *                    kernels/cholesky.cxx:191-212, src/ssids/fkeep.F90:229-318
 *   enquire          src/ssids/cpu/NumericSubtree.hxx:418-462
 * The reference's default indefinite kernel is APP (a-posteriori pivoting) which falls back to TPP;
 * this restatement uses TPP throughout, so pivot SEQUENCES may differ from the reference while
 * inertia and residuals must agree.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int n, nnodes, realn;
  int *perm, *invp;      /* perm[var] = pivot position, 0-based */
  int *sptr, *sparent;
  int64_t *rptr;
  int *rlist;
  int64_t *nptr, *nlist; /* pairs (src, dst), 0-based */
  int64_t num_factor, num_flops;
  /* numeric */
  int posdef, factored;
  double **lcol;         /* per node: (m+ndin) x (n+ndin) */
  double **dvec;         /* per node: 2*(n+ndin) inverted pivots */
  int **nperm;           /* per node: pivot positions of the n+ndin fully summed columns */
  double **contrib;      /* per node: (m-n)^2 or NULL */
  int *nelim, *ndin, *ndout;
  int num_neg, num_two, num_zero, num_delay, flag;
} oracle_t;

/* ------------------------------------------------------------------ analyse ------------------ */

static int find_root(int *vf, int u) { /* FIND with path compression, core_analyse.f90:505-521 */
  int cur = u, prev;
  while (vf[cur] != -1) {
    prev = cur;
    cur = vf[cur];
    if (vf[cur] != -1) vf[prev] = vf[cur];
  }
  return cur;
}

/* decreasing order of val[idx[]], stable (core_analyse.f90:712-801 sorts the same way) */
static void sort_desc_stable(int cnt, int *idx, const int *val) {
  for (int a = 1; a < cnt; ++a) {
    int t = idx[a], b = a - 1;
    while (b >= 0 && val[idx[b]] < val[t]) {
      idx[b + 1] = idx[b];
      --b;
    }
    idx[b + 1] = t;
  }
}

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

void oracle_free(oracle_t *o);

/* order[n]: 1-based pivot position of each variable on entry, final order on exit */
oracle_t *oracle_analyse(int n, const int64_t *ptr, const int32_t *row, int32_t *order, int nemin,
                         int *flag_out) {
  oracle_t *o = (oracle_t *)calloc(1, sizeof(oracle_t));
  *flag_out = 0;
  o->n = n;
  if (nemin < 1) nemin = 32;
  int64_t nz = ptr[n] - 1;
  /* expand_pattern (anal.f90:37-80): both triangles */
  int64_t *ap = (int64_t *)calloc(n + 2, sizeof(int64_t));
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k) {
      int i = row[k] - 1;
      ap[i + 1]++;
      if (i != j) ap[j + 1]++;
    }
  for (int j = 0; j < n; ++j) ap[j + 1] += ap[j];
  int *ar = (int *)malloc(sizeof(int) * (size_t)(ap[n] > 0 ? ap[n] : 1));
  int64_t *nx = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
  memcpy(nx, ap, sizeof(int64_t) * (n + 1));
  for (int j = 0; j < n; ++j)
    for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k) {
      int i = row[k] - 1;
      ar[nx[i]++] = j;
      if (i != j) ar[nx[j]++] = i;
    }
  /* check_order (anal.f90:147-197) */
  int *perm = o->perm = (int *)malloc(sizeof(int) * n);
  int *invp = o->invp = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) invp[i] = -1;
  for (int i = 0; i < n; ++i) {
    int j = abs(order[i]);
    if (j < 1 || j > n || invp[j - 1] != -1) {
      *flag_out = -8;
      free(ap); free(ar); free(nx);
      oracle_free(o);
      return NULL;
    }
    invp[j - 1] = i;
    perm[i] = j - 1;
  }
  /* find_etree (core_analyse.f90:173-223) */
  int *parent = (int *)malloc(sizeof(int) * (n + 1));
  int *vf = (int *)malloc(sizeof(int) * (n + 1));
  for (int i = 0; i <= n; ++i) vf[i] = n;
  for (int piv = 0; piv < n; ++piv) {
    int c = invp[piv];
    for (int64_t k = ap[c]; k < ap[c + 1]; ++k) {
      int j = perm[ar[k]];
      if (j >= piv) continue;
      int kk = j;
      while (vf[kk] < piv) {
        int l = vf[kk];
        vf[kk] = piv;
        kk = l;
      }
      if (vf[kk] == piv) continue;
      parent[kk] = piv;
      vf[kk] = piv;
    }
    parent[piv] = n;
  }
  /* find_postorder (core_analyse.f90:233-352) */
  int *chead = (int *)malloc(sizeof(int) * (n + 1));
  int *cnext = (int *)malloc(sizeof(int) * (n + 1));
  int *map = (int *)malloc(sizeof(int) * (n + 1));
  int *stack = (int *)malloc(sizeof(int) * (n + 1));
  for (int i = 0; i <= n; ++i) chead[i] = -1;
  for (int i = n - 1; i >= 0; --i) {
    cnext[i] = chead[parent[i]];
    chead[parent[i]] = i;
  }
  int realn = n, sh = 0, id = n;
  stack[sh++] = n;
  while (sh > 0) {
    int node = stack[--sh];
    map[node] = id--;
    if (node == n) {
      for (int i = chead[node]; i != -1; i = cnext[i])
        if (ap[invp[i] + 1] - ap[invp[i]] != 0) stack[sh++] = i;
      for (int i = chead[node]; i != -1; i = cnext[i])
        if (ap[invp[i] + 1] - ap[invp[i]] == 0) {
          realn--;
          stack[sh++] = i;
        }
    } else {
      for (int i = chead[node]; i != -1; i = cnext[i]) stack[sh++] = i;
    }
  }
  {
    int *tmp = (int *)malloc(sizeof(int) * n);
    memcpy(tmp, invp, sizeof(int) * n);
    for (int i = 0; i < n; ++i) invp[map[i]] = tmp[i];
    for (int i = 0; i < n; ++i) perm[invp[i]] = i;
    for (int i = 0; i < n; ++i) tmp[i] = map[parent[i]];
    for (int i = 0; i < n; ++i) parent[map[i]] = tmp[i];
    free(tmp);
  }
  o->realn = realn;
  if (realn != n) *flag_out = 6; /* SSIDS_WARNING_ANAL_SINGULAR */
  /* find_col_counts (core_analyse.f90:387-501) */
  int *cc = (int *)calloc(n + 1, sizeof(int));
  int *first = (int *)malloc(sizeof(int) * (n + 1));
  int *last_p = (int *)malloc(sizeof(int) * (n + 1));
  int *last_nbr = (int *)malloc(sizeof(int) * (n + 1));
  for (int i = 0; i <= n; ++i) first[i] = i;
  for (int i = 0; i < n; ++i) {
    int par = parent[i];
    if (first[i] < first[par]) first[par] = first[i];
    cc[i] = (first[i] == i) ? 1 : 0;
  }
  cc[n] = n + 1;
  for (int i = 0; i <= n; ++i) {
    vf[i] = -1;
    last_p[i] = -1;
    last_nbr[i] = -1;
  }
  for (int piv = 0; piv < n; ++piv) {
    int c = invp[piv];
    for (int64_t k = ap[c]; k < ap[c + 1]; ++k) {
      int u = perm[ar[k]];
      if (u <= piv) continue;
      if (first[piv] > last_nbr[u]) {
        cc[piv]++;
        int pp = last_p[u];
        if (pp != -1) cc[find_root(vf, pp)]--;
        last_p[u] = piv;
      }
      last_nbr[u] = piv;
    }
    int par = parent[piv];
    cc[par] += cc[piv] - 1;
    vf[piv] = par;
  }
  /* find_supernodes (core_analyse.f90:536-707), do_merge :806-822, merge_nodes :827-853 */
  int *nelim = (int *)malloc(sizeof(int) * (n + 1));
  int *nvert = (int *)malloc(sizeof(int) * (n + 1));
  int *vhead = (int *)malloc(sizeof(int) * (n + 1));
  int *vnext = (int *)malloc(sizeof(int) * (n + 1));
  char *mark = (char *)calloc(n + 1, 1);
  int *child = (int *)malloc(sizeof(int) * (n + 1));
  for (int i = 0; i <= n; ++i) {
    nelim[i] = 1;
    nvert[i] = 1;
    vhead[i] = vnext[i] = -1;
    chead[i] = -1;
  }
  nelim[n] = n + 1 + nemin;
  for (int i = realn - 1; i >= 0; --i) {
    cnext[i] = chead[parent[i]];
    chead[parent[i]] = i;
  }
  for (int par = 0; par <= n; ++par) {
    int nchild = 0;
    for (int nd = chead[par]; nd != -1; nd = cnext[nd]) child[nchild++] = nd;
    sort_desc_stable(nchild, child, cc);
    for (int j = 0; j < nchild; ++j) {
      int nd = child[j];
      int merge = 0;
      if (par != n)
        merge = ((cc[par] == cc[nd] - 1) && nelim[par] == 1) || (nelim[par] < nemin && nelim[nd] < nemin);
      if (merge) {
        vnext[nd] = vhead[par];
        vhead[par] = nd;
        nelim[par] += nelim[nd];
        nvert[par] += nvert[nd];
      } else {
        mark[nd] = 1;
      }
    }
  }
  int *sperm = (int *)malloc(sizeof(int) * (n + 1));
  int *npar = (int *)malloc(sizeof(int) * (n + 1));
  int *scc = (int *)malloc(sizeof(int) * (n + 1));
  o->sptr = (int *)malloc(sizeof(int) * (n + 2));
  int nn = 0, v = 0;
  for (int nd = 0; nd < realn; ++nd) {
    if (!mark[nd]) continue;
    o->sptr[nn] = v;
    npar[nn] = parent[nd];
    scc[nn] = cc[nd] + nelim[nd] - 1;
    v += nvert[nd];
    int k = v;
    sh = 0;
    stack[sh++] = nd;
    while (sh > 0) {
      int i = stack[--sh];
      sperm[i] = --k;
      map[i] = nn;
      if (vnext[i] != -1) stack[sh++] = vnext[i];
      if (vhead[i] != -1) stack[sh++] = vhead[i];
    }
    nn++;
  }
  o->sptr[nn] = v;
  map[n] = nn;
  for (int i = realn; i < n; ++i) sperm[i] = i;
  o->nnodes = nn;
  o->sparent = (int *)malloc(sizeof(int) * (nn + 1));
  for (int s = 0; s < nn; ++s) o->sparent[s] = map[npar[s]];
  /* apply_perm (core_analyse.f90:1069-1098) */
  {
    int *tmp = (int *)malloc(sizeof(int) * n);
    memcpy(tmp, invp, sizeof(int) * n);
    for (int i = 0; i < n; ++i) invp[sperm[i]] = tmp[i];
    for (int i = 0; i < n; ++i) perm[invp[i]] = i;
    free(tmp);
  }
  /* find_row_lists (core_analyse.f90:911-998) + dbl_tr_sort (:1007-1064) */
  o->rptr = (int64_t *)malloc(sizeof(int64_t) * (nn + 1));
  o->rptr[0] = 0;
  for (int s = 0; s < nn; ++s) o->rptr[s + 1] = o->rptr[s] + scc[s];
  o->rlist = (int *)malloc(sizeof(int) * (size_t)(o->rptr[nn] > 0 ? o->rptr[nn] : 1));
  {
    int *seen = (int *)malloc(sizeof(int) * n);
    int *sch = (int *)malloc(sizeof(int) * (nn + 1));
    int *scn = (int *)malloc(sizeof(int) * (nn + 1));
    for (int i = 0; i < n; ++i) seen[i] = -1;
    for (int s = 0; s <= nn; ++s) sch[s] = -1;
    for (int s = nn - 1; s >= 0; --s) {
      scn[s] = sch[o->sparent[s]];
      sch[o->sparent[s]] = s;
    }
    for (int s = 0; s < nn; ++s) {
      int64_t idx = o->rptr[s];
      for (int p = o->sptr[s]; p < o->sptr[s + 1]; ++p) {
        seen[p] = s;
        o->rlist[idx++] = p;
      }
      for (int c = sch[s]; c != -1; c = scn[c])
        for (int64_t k = o->rptr[c]; k < o->rptr[c + 1]; ++k) {
          int j = o->rlist[k];
          if (j < o->sptr[s] || seen[j] == s) continue;
          seen[j] = s;
          o->rlist[idx++] = j;
        }
      for (int p = o->sptr[s]; p < o->sptr[s + 1]; ++p) {
        int c = invp[p];
        for (int64_t k = ap[c]; k < ap[c + 1]; ++k) {
          int j = perm[ar[k]];
          if (j < p || seen[j] == s) continue;
          seen[j] = s;
          o->rlist[idx++] = j;
        }
      }
      qsort(o->rlist + o->rptr[s], (size_t)(o->rptr[s + 1] - o->rptr[s]), sizeof(int), cmp_int);
    }
    free(seen); free(sch); free(scn);
  }
  /* calc_stats (core_analyse.f90:862-902) */
  for (int s = 0; s < nn; ++s) {
    int64_t ne = o->sptr[s + 1] - o->sptr[s], m = scc[s] - ne;
    o->num_factor += ne * (ne + 1) / 2 + ne * m;
    for (int64_t j = 1; j <= ne; ++j) o->num_flops += (m + j) * (m + j);
  }
  for (int i = 0; i < n; ++i) order[i] = perm[i] + 1;
  for (int p = o->sptr[nn]; p < n; ++p) order[invp[p]] = 0;
  /* build_map (anal.f90:1129-1231) */
  {
    int64_t *tp = (int64_t *)calloc(n + 2, sizeof(int64_t));
    int64_t *origin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nz > 0 ? nz : 1));
    int *tc = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    for (int j = 0; j < n; ++j)
      for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k)
        if (row[k] - 1 != j) tp[row[k]]++;
    for (int j = 0; j < n; ++j) tp[j + 1] += tp[j];
    memcpy(nx, tp, sizeof(int64_t) * (n + 1));
    for (int j = 0; j < n; ++j)
      for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k) {
        int i = row[k] - 1;
        if (i == j) continue;
        tc[nx[i]] = j;
        origin[nx[i]++] = k;
      }
    o->nptr = (int64_t *)malloc(sizeof(int64_t) * (nn + 1));
    o->nlist = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(nz > 0 ? nz : 1));
    int64_t pp = 0;
    for (int s = 0; s < nn; ++s) {
      int64_t m = o->rptr[s + 1] - o->rptr[s];
      o->nptr[s] = pp;
      for (int64_t k = o->rptr[s]; k < o->rptr[s + 1]; ++k) map[o->rlist[k]] = (int)(k - o->rptr[s]);
      for (int p = o->sptr[s]; p < o->sptr[s + 1]; ++p) {
        int c = invp[p];
        for (int64_t k = tp[c]; k < tp[c + 1]; ++k) {
          int r = perm[tc[k]];
          if (r < p) continue;
          o->nlist[2 * pp] = origin[k];
          o->nlist[2 * pp + 1] = (int64_t)(p - o->sptr[s]) * m + map[r];
          pp++;
        }
      }
      for (int p = o->sptr[s]; p < o->sptr[s + 1]; ++p) {
        int c = invp[p];
        for (int64_t k = ptr[c] - 1; k < ptr[c + 1] - 1; ++k) {
          int r = perm[row[k] - 1];
          if (r < p) continue;
          o->nlist[2 * pp] = k;
          o->nlist[2 * pp + 1] = (int64_t)(p - o->sptr[s]) * m + map[r];
          pp++;
        }
      }
    }
    o->nptr[nn] = pp;
    free(tp); free(origin); free(tc);
  }
  free(ap); free(ar); free(nx); free(parent); free(vf); free(chead); free(cnext); free(map);
  free(stack); free(cc); free(first); free(last_p); free(last_nbr); free(nelim); free(nvert);
  free(vhead); free(vnext); free(mark); free(child); free(sperm); free(npar); free(scc);
  return o;
}

/* ------------------------------------------------------------------ numeric ------------------ */

static void free_numeric(oracle_t *o) {
  if (!o->lcol) return;
  for (int s = 0; s < o->nnodes; ++s) {
    free(o->lcol[s]); free(o->dvec[s]); free(o->nperm[s]); free(o->contrib[s]);
  }
  free(o->lcol); free(o->dvec); free(o->nperm); free(o->contrib);
  free(o->nelim); free(o->ndin); free(o->ndout);
  o->lcol = NULL;
}

void oracle_free(oracle_t *o) {
  if (!o) return;
  free_numeric(o);
  free(o->perm); free(o->invp); free(o->sptr); free(o->sparent); free(o->rptr); free(o->rlist);
  free(o->nptr); free(o->nlist);
  free(o);
}

/* ---- ldlt_tpp.cxx:20-100 helpers ---- */
static int col_small(int idx, int from, int to, const double *a, int lda, double small) {
  for (int c = from; c < idx; ++c) if (!(fabs(a[(size_t)c * lda + idx]) < small)) return 0;
  for (int r = idx; r < to; ++r) if (!(fabs(a[(size_t)idx * lda + r]) < small)) return 0;
  return 1;
}
static int row_abs_max(int from, int to, const double *a, int lda) {
  if (from >= to) return -1;
  int best = from;
  double bv = fabs(a[(size_t)from * lda]);
  for (int i = from + 1; i < to; ++i)
    if (fabs(a[(size_t)i * lda]) > bv) { best = i; bv = fabs(a[(size_t)i * lda]); }
  return best;
}
static void dswap(double *x, double *y) { double t = *x; *x = *y; *y = t; }
static void swap_cols(int c1, int c2, int m, int *perm, double *a, int lda) {
  if (c1 == c2) return;
  if (c2 < c1) { int t = c1; c1 = c2; c2 = t; }
  { int t = perm[c1]; perm[c1] = perm[c2]; perm[c2] = t; }
  for (int c = 0; c < c1; ++c) dswap(&a[(size_t)c * lda + c1], &a[(size_t)c * lda + c2]);
  for (int i = c1 + 1; i < c2; ++i) dswap(&a[(size_t)c1 * lda + i], &a[(size_t)i * lda + c2]);
  for (int r = c2 + 1; r < m; ++r) dswap(&a[(size_t)c1 * lda + r], &a[(size_t)c2 * lda + r]);
  dswap(&a[(size_t)c1 * lda + c1], &a[(size_t)c2 * lda + c2]);
}
static double rc_abs_max_excl(int col, int nelim, int m, const double *a, int lda, int excl) {
  double best = 0.0;
  for (int c = nelim; c < col; ++c) if (c != excl) best = fmax(best, fabs(a[(size_t)c * lda + col]));
  for (int r = col + 1; r < m; ++r) if (r != excl) best = fmax(best, fabs(a[(size_t)col * lda + r]));
  return best;
}
static int test_2x2(int t, int p, double maxt, double maxp, const double *a, int lda, double u,
                    double small, double *d) {
  double a11 = a[(size_t)t * lda + t], a21 = a[(size_t)t * lda + p], a22 = a[(size_t)p * lda + p];
  double maxpiv = fmax(fabs(a11), fmax(fabs(a21), fabs(a22)));
  if (maxpiv < small) return 0;
  double detscale = 1 / maxpiv;
  double detpiv0 = (a11 * detscale) * a22, detpiv1 = (a21 * detscale) * a21;
  double detpiv = detpiv0 - detpiv1;
  if (fabs(detpiv) < fmax(small, fmax(fabs(detpiv0 / 2), fabs(detpiv1 / 2)))) return 0;
  d[0] = (a22 * detscale) / detpiv;
  d[1] = (-a21 * detscale) / detpiv;
  d[2] = INFINITY;
  d[3] = (a11 * detscale) / detpiv;
  if (fmax(maxp, maxt) < small) return 1;
  double x1 = fabs(d[0]) * maxt + fabs(d[1]) * maxp;
  double x2 = fabs(d[1]) * maxt + fabs(d[3]) * maxp;
  return u * fmax(x1, x2) < 1.0;
}
/* trailing update a(r, c) -= sum_k l(r,k) * ld(c,k) over the remaining fully-summed columns */
static void trail_update(int m, int n, int nelim, int npiv, double *a, int lda, const double *ld, int ldld) {
  for (int c = nelim + npiv; c < n; ++c)
    for (int k = 0; k < npiv; ++k) {
      double w = ld[(size_t)k * ldld + c];
      const double *l = &a[(size_t)(nelim + k) * lda];
      double *dst = &a[(size_t)c * lda];
      for (int r = c; r < m; ++r) dst[r] -= l[r] * w;
    }
}
/* ldlt_tpp_factor, ldlt_tpp.cxx:140-240; returns #eliminated, -1 on singular without action */
static int tpp_factor(int m, int n, int *perm, double *a, int lda, double *d, double *ld, int ldld,
                      int action, double u, double small) {
  int nelim = 0;
  while (nelim < n) {
    if (col_small(nelim, nelim, m, a, lda, small)) {
      if (!action) return -1;
      for (int r = nelim; r < m; ++r) a[(size_t)nelim * lda + r] = 0.0;
      d[2 * nelim] = 0.0; d[2 * nelim + 1] = 0.0;
      nelim++;
      continue;
    }
    int p;
    for (p = nelim + 1; p < n; ++p) {
      if (col_small(p, nelim, m, a, lda, small)) {
        if (!action) return -1;
        swap_cols(p, nelim, m, perm, a, lda);
        for (int r = nelim; r < m; ++r) a[(size_t)nelim * lda + r] = 0.0;
        d[2 * nelim] = 0.0; d[2 * nelim + 1] = 0.0;
        nelim++;
        break;
      }
      int t = row_abs_max(nelim, p, &a[p], lda);
      double maxt = rc_abs_max_excl(t, nelim, m, a, lda, p);
      double maxp = rc_abs_max_excl(p, nelim, m, a, lda, t);
      if (test_2x2(t, p, maxt, maxp, a, lda, u, small, &d[2 * nelim])) {
        swap_cols(t, nelim, m, perm, a, lda);
        swap_cols(p, nelim + 1, m, perm, a, lda);
        double *a1 = &a[(size_t)nelim * lda], *a2 = &a[(size_t)(nelim + 1) * lda];
        a1[nelim] = 1.0; a1[nelim + 1] = 0.0; a2[nelim + 1] = 1.0;
        double d11 = d[2 * nelim], d21 = d[2 * nelim + 1], d22 = d[2 * nelim + 3];
        for (int r = nelim + 2; r < m; ++r) {
          ld[r] = a1[r]; ld[ldld + r] = a2[r];
          a1[r] = d11 * ld[r] + d21 * ld[ldld + r];
          a2[r] = d21 * ld[r] + d22 * ld[ldld + r];
        }
        trail_update(m, n, nelim, 2, a, lda, ld, ldld);
        nelim += 2;
        break;
      }
      maxp = fmax(maxp, fabs(a[(size_t)t * lda + p]));
      if (fabs(a[(size_t)p * lda + p]) >= u * maxp) {
        swap_cols(p, nelim, m, perm, a, lda);
        goto pivot_1x1;
      }
    }
    if (p >= n) {
      p = nelim;
      double maxp = rc_abs_max_excl(p, nelim, m, a, lda, -1);
      if (fabs(a[(size_t)p * lda + p]) >= u * maxp) goto pivot_1x1;
      break; /* out of pivots */
    }
    continue;
  pivot_1x1: {
      double *a1 = &a[(size_t)nelim * lda];
      d[2 * nelim] = 1 / a1[nelim];
      d[2 * nelim + 1] = 0.0;
      a1[nelim] = 1.0;
      double d11 = d[2 * nelim];
      for (int r = nelim + 1; r < m; ++r) { ld[r] = a1[r]; a1[r] *= d11; }
      trail_update(m, n, nelim, 1, a, lda, ld, ldld);
      nelim += 1;
    }
  }
  return nelim;
}

/* unblocked Cholesky of the m x n front; returns -1 ok, else failing column (cholesky.cxx:32-188) */
static int chol_factor(int m, int n, double *a, int lda) {
  for (int j = 0; j < n; ++j) {
    double *cj = &a[(size_t)j * lda];
    if (!(cj[j] > 0.0)) return j;
    double s = sqrt(cj[j]);
    cj[j] = s;
    for (int r = j + 1; r < m; ++r) cj[r] /= s;
    for (int c = j + 1; c < n; ++c) {
      double w = cj[c];
      double *cc = &a[(size_t)c * lda];
      for (int r = c; r < m; ++r) cc[r] -= cj[r] * w;
    }
  }
  return -1;
}

/* factor: returns SSIDS flag (0, 7 warning singular, -5 singular, -6 not posdef) */
int oracle_factor(oracle_t *o, int posdef, const double *aval, const double *scaling, double u,
                  double small, int action) {
  free_numeric(o);
  int nn = o->nnodes, n = o->n;
  o->posdef = posdef; o->factored = 0; o->flag = 0;
  o->num_neg = o->num_two = o->num_zero = o->num_delay = 0;
  o->lcol = (double **)calloc(nn + 1, sizeof(double *));
  o->dvec = (double **)calloc(nn + 1, sizeof(double *));
  o->nperm = (int **)calloc(nn + 1, sizeof(int *));
  o->contrib = (double **)calloc(nn + 1, sizeof(double *));
  o->nelim = (int *)calloc(nn + 1, sizeof(int));
  o->ndin = (int *)calloc(nn + 1, sizeof(int));
  o->ndout = (int *)calloc(nn + 1, sizeof(int));
  int *map = (int *)malloc(sizeof(int) * (n + 1));
  int *ch = (int *)malloc(sizeof(int) * (nn + 1)), *cn = (int *)malloc(sizeof(int) * (nn + 1));
  for (int s = 0; s <= nn; ++s) ch[s] = -1;
  for (int s = nn - 1; s >= 0; --s) { cn[s] = ch[o->sparent[s]]; ch[o->sparent[s]] = s; }
  int rc = 0;
  for (int s = 0; s < nn && rc >= 0; ++s) {
    const int *rl = o->rlist + o->rptr[s];
    int sm = (int)(o->rptr[s + 1] - o->rptr[s]), sn = o->sptr[s + 1] - o->sptr[s];
    /* ---- assemble_pre (assemble.hxx:139-345) ---- */
    int ndin = 0;
    for (int c = ch[s]; c != -1; c = cn[c]) ndin += o->ndout[c];
    o->ndin[s] = ndin;
    int m = sm + ndin, nc = sn + ndin, ldl = m;
    double *L = o->lcol[s] = (double *)calloc((size_t)ldl * nc + 1, sizeof(double));
    double *d = o->dvec[s] = (double *)calloc(2 * (size_t)nc + 2, sizeof(double));
    int *pm = o->nperm[s] = (int *)malloc(sizeof(int) * (nc + 1));
    int cm = sm - sn;
    double *C = o->contrib[s] = cm > 0 ? (double *)calloc((size_t)cm * cm, sizeof(double)) : NULL;
    for (int i = 0; i < sn; ++i) pm[i] = rl[i];
    for (int64_t k = o->nptr[s]; k < o->nptr[s + 1]; ++k) { /* add_a_block :49-79 */
      int64_t dst = o->nlist[2 * k + 1];
      int c = (int)(dst / sm), r = (int)(dst % sm);
      size_t pos = (size_t)c * ldl + r + (r >= sn ? ndin : 0);
      double v = aval[o->nlist[2 * k]];
      if (scaling) v *= scaling[o->invp[rl[r]]] * scaling[o->invp[rl[c]]];
      L[pos] = v;
    }
    for (int i = 0; i < sn; ++i) map[rl[i]] = i;
    for (int i = sn; i < sm; ++i) map[rl[i]] = i + ndin;
    int delay_col = sn;
    for (int c = ch[s]; c != -1; c = cn[c]) {
      const int *crl = o->rlist + o->rptr[c];
      int csm = (int)(o->rptr[c + 1] - o->rptr[c]), csn = o->sptr[c + 1] - o->sptr[c];
      int lds = csm + o->ndin[c];
      const double *cl = o->lcol[c];
      for (int i = 0; i < o->ndout[c]; ++i) { /* delays :244-264 */
        const double *src = &cl[(size_t)(o->nelim[c] + i) * (lds + 1)];
        double *dest = &L[(size_t)delay_col * (ldl + 1)];
        pm[delay_col] = o->nperm[c][o->nelim[c] + i];
        for (int j = 0; j < o->ndout[c] - i; ++j) dest[j] = src[j];
        src = &cl[(size_t)(o->nelim[c] + i) * lds + o->ndin[c]];
        for (int j = csn; j < csm; ++j) {
          int r = map[crl[j]];
          if (r < nc) L[(size_t)r * ldl + delay_col] = src[j];
          else L[(size_t)delay_col * ldl + r] = src[j];
        }
        delay_col++;
      }
      if (o->contrib[c]) { /* assemble_expected :91-107 */
        int ccm = csm - csn;
        for (int i = 0; i < ccm; ++i) {
          int cc = map[crl[csn + i]];
          if (cc >= sn) continue;
          const double *src = &o->contrib[c][(size_t)i * ccm];
          for (int j = i; j < ccm; ++j) L[(size_t)cc * ldl + map[crl[csn + j]]] += src[j];
        }
      }
    }
    /* ---- factor_node (factor.hxx:36-160) ---- */
    if (posdef) {
      int bad = chol_factor(m, nc, L, ldl);
      if (bad != -1) { rc = -6; break; }
      o->nelim[s] = nc;
      o->ndout[s] = 0;
      for (int c = 0; c < cm; ++c)       /* upd = -L2 L2^T, beta = 0 */
        for (int k = 0; k < nc; ++k) {
          double w = L[(size_t)k * ldl + nc + c];
          for (int r = c; r < cm; ++r) C[(size_t)c * cm + r] -= L[(size_t)k * ldl + nc + r] * w;
        }
    } else {
      double *ld = (double *)malloc(sizeof(double) * 2 * (size_t)(m + 1));
      int ne = tpp_factor(m, nc, pm, L, ldl, d, ld, m, action, u, small);
      free(ld);
      if (ne < 0) { rc = -5; break; }
      o->nelim[s] = ne;
      o->ndout[s] = nc - ne;
      o->num_delay += nc - ne;
      if (cm > 0 && ne > 0) { /* calcLD (calc_ld.hxx:43-118) + gemm, factor.hxx:84-99 */
        double *LD = (double *)malloc(sizeof(double) * (size_t)cm * ne);
        for (int col = 0; col < ne;) {
          if (col + 1 == ne || isfinite(d[2 * col + 2])) {
            double d11 = d[2 * col];
            if (d11 != 0.0) d11 = 1 / d11;
            for (int r = 0; r < cm; ++r) LD[(size_t)col * cm + r] = d11 * L[(size_t)col * ldl + nc + r];
            col++;
          } else {
            double d11 = d[2 * col], d21 = d[2 * col + 1], d22 = d[2 * col + 3];
            double det = d11 * d22 - d21 * d21;
            d11 /= det; d21 /= det; d22 /= det;
            for (int r = 0; r < cm; ++r) {
              double a1 = L[(size_t)col * ldl + nc + r], a2 = L[(size_t)(col + 1) * ldl + nc + r];
              LD[(size_t)col * cm + r] = d22 * a1 - d21 * a2;
              LD[(size_t)(col + 1) * cm + r] = -d21 * a1 + d11 * a2;
            }
            col += 2;
          }
        }
        for (int c = 0; c < cm; ++c)
          for (int k = 0; k < ne; ++k) {
            double w = LD[(size_t)k * cm + c];
            for (int r = c; r < cm; ++r) C[(size_t)c * cm + r] -= L[(size_t)k * ldl + nc + r] * w;
          }
        free(LD);
      }
      /* inertia, NumericSubtree.hxx:242-274 */
      for (int i = 0; i < ne;) {
        double a11 = d[2 * i], a21 = d[2 * i + 1];
        if (i + 1 == ne || isfinite(d[2 * i + 2])) {
          if (a11 == 0.0) { o->num_zero++; o->flag = 7; }
          if (a11 < 0.0) o->num_neg++;
          i++;
        } else {
          double a22 = d[2 * i + 3];
          o->num_two++;
          double det = a11 * a22 - a21 * a21, tr = a11 + a22;
          if (det < 0) o->num_neg++;
          else if (tr < 0) o->num_neg += 2;
          i += 2;
        }
      }
    }
    /* ---- assemble_post (assemble.hxx:347-437) ---- */
    for (int c = ch[s]; c != -1; c = cn[c]) {
      if (!o->contrib[c]) continue;
      const int *crl = o->rlist + o->rptr[c];
      int csm = (int)(o->rptr[c + 1] - o->rptr[c]), csn = o->sptr[c + 1] - o->sptr[c];
      int ccm = csm - csn;
      for (int i = 0; i < ccm; ++i) {
        int cc = map[crl[csn + i]];
        if (cc < sn) continue;
        const double *src = &o->contrib[c][(size_t)i * ccm];
        for (int j = i; j < ccm; ++j)
          C[(size_t)(cc - nc) * cm + (map[crl[csn + j]] - nc)] += src[j];
      }
      free(o->contrib[c]);
      o->contrib[c] = NULL;
    }
  }
  free(map); free(ch); free(cn);
  if (rc < 0) { o->flag = rc; return rc; }
  o->factored = 1;
  return o->flag;
}

void oracle_stats(const oracle_t *o, int64_t *num_factor, int64_t *num_flops, int *nnodes,
                  int *num_neg, int *num_two, int *num_zero, int *num_delay) {
  *num_factor = o->num_factor; *num_flops = o->num_flops; *nnodes = o->nnodes;
  *num_neg = o->num_neg; *num_two = o->num_two; *num_zero = o->num_zero; *num_delay = o->num_delay;
}

/* job: 0 all, 1 fwd, 2 diag, 3 bwd, 4 diag+bwd  (NumericSubtree.hxx:280-400, fkeep.F90:229-318) */
int oracle_solve(const oracle_t *o, int job, int nrhs, double *x, int ldx, const double *scaling) {
  if (!o->factored) return -1;
  int n = o->n, nn = o->nnodes;
  double *x2 = (double *)malloc(sizeof(double) * (n + 1));
  double *xl = (double *)malloc(sizeof(double) * (n + 1));
  int *mp = (int *)malloc(sizeof(int) * (n + 1));
  int do_f = (job == 0 || job == 1), do_d = !o->posdef && (job == 0 || job == 2 || job == 4);
  int do_b = (job == 0 || job == 3 || job == 4);
  for (int rh = 0; rh < nrhs; ++rh) {
    double *xx = x + (size_t)rh * ldx;
    for (int p = 0; p < n; ++p) {
      int v = o->invp[p];
      x2[p] = (scaling && do_f) ? xx[v] * scaling[v] : xx[v];
    }
    for (int s = 0; s < nn && do_f; ++s) {
      const int *rl = o->rlist + o->rptr[s];
      int sm = (int)(o->rptr[s + 1] - o->rptr[s]), sn = o->sptr[s + 1] - o->sptr[s];
      int ndin = o->ndin[s], ne = o->nelim[s], m = sm + ndin, ldl = m;
      const double *L = o->lcol[s];
      for (int i = 0; i < sn + ndin; ++i) mp[i] = o->nperm[s][i];
      for (int i = sn; i < sm; ++i) mp[i + ndin] = rl[i];
      for (int i = 0; i < m; ++i) xl[i] = x2[mp[i]];
      for (int k = 0; k < ne; ++k) {
        if (o->posdef) xl[k] /= L[(size_t)k * ldl + k];
        double w = xl[k];
        for (int r = k + 1; r < m; ++r) xl[r] -= L[(size_t)k * ldl + r] * w;
      }
      for (int i = 0; i < m; ++i) x2[mp[i]] = xl[i];
    }
    for (int s = nn - 1; s >= 0 && (do_d || do_b); --s) {
      const int *rl = o->rlist + o->rptr[s];
      int sm = (int)(o->rptr[s + 1] - o->rptr[s]), sn = o->sptr[s + 1] - o->sptr[s];
      int ndin = o->ndin[s], ne = o->nelim[s], m = sm + ndin, ldl = m;
      const double *L = o->lcol[s], *d = o->dvec[s];
      for (int i = 0; i < sn + ndin; ++i) mp[i] = o->nperm[s][i];
      for (int i = sn; i < sm; ++i) mp[i + ndin] = rl[i];
      for (int i = 0; i < m; ++i) xl[i] = x2[mp[i]];
      if (do_d)
        for (int i = 0; i < ne;) {
          if (i + 1 == ne || isfinite(d[2 * i + 2])) { xl[i] *= d[2 * i]; i++; }
          else {
            double d11 = d[2 * i], d21 = d[2 * i + 1], d22 = d[2 * i + 3], x1 = xl[i], xb = xl[i + 1];
            xl[i] = d11 * x1 + d21 * xb; xl[i + 1] = d21 * x1 + d22 * xb; i += 2;
          }
        }
      if (do_b)
        for (int k = ne - 1; k >= 0; --k) {
          double sum = xl[k];
          for (int r = k + 1; r < m; ++r) sum -= L[(size_t)k * ldl + r] * xl[r];
          xl[k] = o->posdef ? sum / L[(size_t)k * ldl + k] : sum;
        }
      for (int i = 0; i < ne; ++i) x2[mp[i]] = xl[i];
    }
    for (int p = 0; p < n; ++p) {
      int v = o->invp[p];
      xx[v] = (scaling && do_b) ? x2[p] * scaling[v] : x2[p];
    }
  }
  free(x2); free(xl); free(mp);
  return 0;
}

/* symbolic arrays, 1-based like the reference's akeep (any pointer may be NULL) */
void oracle_get_symbolic(const oracle_t *o, int32_t *sptr, int32_t *sparent, int64_t *rptr,
                         int32_t *rlist, int64_t *nptr, int64_t *nlist) {
  int nn = o->nnodes;
  if (sptr) for (int i = 0; i <= nn; ++i) sptr[i] = o->sptr[i] + 1;
  if (sparent) for (int i = 0; i < nn; ++i) sparent[i] = o->sparent[i] + 1;
  if (rptr) for (int i = 0; i <= nn; ++i) rptr[i] = o->rptr[i] + 1;
  if (rlist) for (int64_t i = 0; i < o->rptr[nn]; ++i) rlist[i] = o->rlist[i] + 1;
  if (nptr) for (int i = 0; i <= nn; ++i) nptr[i] = o->nptr[i] + 1;
  if (nlist) for (int64_t i = 0; i < 2 * o->nptr[nn]; ++i) nlist[i] = o->nlist[i] + 1;
}
int64_t oracle_rlist_len(const oracle_t *o) { return o->rptr[o->nnodes]; }
int64_t oracle_nlist_len(const oracle_t *o) { return o->nptr[o->nnodes]; }

/* piv_order[var] = +-(1-based position in the final pivot sequence), d(2,n) inverted pivots
 * (NumericSubtree.hxx:428-462) */
void oracle_enquire_indef(const oracle_t *o, int32_t *piv_order, double *dout) {
  int piv = 0;
  for (int s = 0; s < o->nnodes; ++s) {
    const double *d = o->dvec[s];
    int ne = o->nelim[s];
    for (int i = 0; i < ne;) {
      if (i + 1 == ne || isfinite(d[2 * i + 2])) {
        if (piv_order) piv_order[o->invp[o->nperm[s][i]]] = piv + 1;
        if (dout) { dout[2 * piv] = d[2 * i]; dout[2 * piv + 1] = 0.0; }
        piv++; i++;
      } else {
        if (piv_order) {
          piv_order[o->invp[o->nperm[s][i]]] = -(piv + 1);
          piv_order[o->invp[o->nperm[s][i + 1]]] = -(piv + 2);
        }
        if (dout) {
          dout[2 * piv] = d[2 * i]; dout[2 * piv + 1] = d[2 * i + 1];
          dout[2 * piv + 2] = d[2 * i + 3]; dout[2 * piv + 3] = 0.0;
        }
        piv += 2; i += 2;
      }
    }
  }
}
