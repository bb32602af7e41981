// C ABI of the gsls backend (include/gsls.h): handle management, host<->HBM staging, status and
// statistics.  Behavioural model: the SSIDS driver routines SLS calls
//   ssids_analyse  src/ssids/ssids.f90:148-392      ssids_factor  :770-1108
//   ssids_solve    :1114-1249                       ssids_enquire_* :1255-1341
//   ssids_alter    :1347-1384                       ssids_free    :1388-1419
// There is deliberately no CPU numeric path: without a HIP device factor/solve fail with -51.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "gsls_device.hpp"

using namespace gsls;

namespace {

struct Handle {
  Symbolic S;
  DeviceFactor F;
  bool analysed = false, factored = false, posdef = false, dev_ready = false, have_scale = false;
  int learned = 0;            // rounds of folding the in-block pivot sequence of a pivoted factorization into the order
  bool tiny_ready = false;    // the order is learned: tiny fronts may go through the wave-per-front kernel
  int tiny_strikes = 0;       // factorizations in which that kernel met a pivot it could not take
  std::vector<int> tiny_black;   // tiny fronts that kernel gave up on: they stay on the workgroup path
  std::vector<double> scale_host; // the scaling the last factorization computed itself (options.scaling > 0), else empty
  bool shard_fast = false;        // multi-GPU: the last factorization needed no pivoting: tiny fronts on the wave kernels
  bool shard_fast_off = false;    // ... unless that path has failed on this handle before
  int last_fast = 0, last_pivoted = 0, last_passes = 0;   // of the last factorization (gsls_get_factor_stats)
  bool own_order = false;     // analyse chose the elimination order itself (it may be refined when values arrive)
  bool preordered = false;    // ... and that refinement (zero-diagonal variables after their neighbours) has been done
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  gsls_inform last;  // statistics of analyse (+factor), returned again by later phases
  std::vector<int64_t> ptr;   // the pattern given to analyse: needed again when failed pivots
  std::vector<int32_t> row;   // force a repair of the elimination order (see plan_repair)
  std::vector<int64_t> diagpos;   // per column: index of its diagonal entry in row/val, -1 if the pattern has none
  std::vector<uint8_t> tppvar;    // per variable: 1 = the front that holds it is factorized by k_front_tpp
  std::vector<uint8_t> tppfail;   // per variable: failures under whole-front pivoting (plan_repair)
  std::vector<uint8_t> force;     // per variable: moved by a repair -- shares a supernode with its parent column
  std::vector<int32_t> partner;   // per variable: the variable it was paired with by the pre-ordering, or -1
  bool tpp_unflagged = false;     // the flags were dropped once after learning; if failures return they stay
  bool tpp_dirty = false;
  bool keep_match_scale = false;     // (gsls_analyse called from gsls_analyse_matching)
  std::vector<double> match_scale;   // scaling saved by gsls_analyse_matching for factorizations with options.scaling = 3
  bool have_coo = false;          // gsls_set_coo has been called for the analysed pattern
  // gsls_set_value_part: the next gsls_factor_coo takes the matrix values from up to four host arrays laid end to end
  // (SBLS: K = [A-part | H-part | -C-part], sbls.f90:3319-3322) instead of one assembled array
  struct ValuePart { const double* val = nullptr; int64_t len = 0; double mult = 1.0; };
  ValuePart parts[4];
  int nparts = 0;
  bool coo_uploaded = false;      // ... and its structure is on the device
  std::vector<int32_t> coo_row, coo_col, coo_map;
  int shard_repairs = 0;          // repair rounds of the sharded path (plan_repair's pass counter)
  std::vector<int32_t> hint_pairs;   // 2x2 pivots of the last discovered sequence as VARIABLE pairs (v1, v2, v1, v2 ...):
                                     // position hints are rebuilt from them after every re-analysis
  int learn_strikes = 0;          // factorizations in a row whose learned pivot sequence failed on the new values
  int tpp_useless = 0;            // flag-only repairs in a row after which every flagged column failed again
  bool eager_delay = false;       // ... twice: failing columns of one-block fronts go straight to the parent
  // in-library exchange (gsls_comm_*): one RCCL communicator per handle, buffers on the handle's device
  ncclComm_t comm = nullptr;
  int comm_ranks = 0, comm_rank = 0;
  double* cx_factor = nullptr;
  double* cx_solve = nullptr;
  int32_t* stat_pin = nullptr;    // pinned landing place of the factorization counters (read_stat)
  double* cx_hostx = nullptr;     // gsls_comm_solve: the host caller's vector on the device
  int cx_hostx_cap = 0;
  int32_t* cx_fail = nullptr;     // nranks x (1 + GSLS_FAILCAP): every rank's failed pivots (all-gather)
  int64_t cx_factor_elems = 0, cx_solve_elems = 0, cx_cut_elems = 0;         // the device-side front flags no longer match tppvar / the current tree
  int nemin = 32;
  double kt_fwd = 0, kt_diag = 0, kt_bwd = 0;
  bool kt_pending = false;        // the last solve was enqueued only: its event times are read on demand
};

double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int fail_hip(Handle* h, gsls_inform* inf, hipError_t e) {
  if (inf) {
    inf->flag = GSLS_ERROR_HIP;
    inf->hip_error = int(e);
  }
  if (h) h->last.flag = GSLS_ERROR_HIP;
  return GSLS_ERROR_HIP;
}

hipError_t ensure_device(Handle* h, const gsls_options* o) {
  if (h->stream) return hipSuccess;
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess) return e;
  if (cnt <= 0) return hipErrorNoDevice;
  int dev = (o && o->device >= 0) ? o->device : -1;
  if (dev < 0) {
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
  }
  h->device = dev;
  e = hipSetDevice(dev);
  if (e != hipSuccess) return e;
  e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) return e;
  for (auto& ev : h->ev) {
    e = hipEventCreate(&ev);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

struct DeviceGuard {  // leave the caller's current device untouched (cf. gpu_interfaces.f90:439-465)
  int prev = -1;
  bool active = false;
  explicit DeviceGuard(int dev) {
    if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) {
      active = (hipSetDevice(dev) == hipSuccess);
    }
  }
  ~DeviceGuard() {
    if (active) (void)hipSetDevice(prev);
  }
};

void fill_from_symbolic(const Symbolic& S, gsls_inform* inf) {
  inf->matrix_rank = S.nnodes ? S.sptr[S.nnodes] : 0;
  inf->maxdepth = S.maxdepth;
  inf->maxfront = S.maxfront;
  inf->num_factor = S.num_factor;
  inf->num_flops = S.num_flops;
  inf->num_sup = S.nnodes;
  inf->nlevels = S.nlevels;
}

// Delayed pivots, MI355X style.  The reference passes a pivot that fails the threshold test up to the
// parent front at run time (ssids/cpu/kernels/assemble.hxx:244-264): front sizes change during the
// factorization and its GPU path re-plans every level on the host.  Here the schedule is static, so a
// failed pivot becomes, in this order,
//   1. a FLAG on its variable: the front that holds it is factorized by k_front_tpp from now on (threshold
//      partial pivoting across ALL of the front's columns, the role of ldlt_tpp_factor on the columns the
//      blocked kernel could not eliminate, factor.hxx:74-106) -- no change of the elimination order;
//   2. if it fails there as well, a change of the ELIMINATION ORDER: the variable moves to just after the
//      last column of the parent supernode (where more of its row is fully summed) and, should it fail
//      once more, to the end of its tree's root; the symbolic analysis is redone and the factorization
//      repeated.  Numerically this is the reference's delay; the flag travels with the variable, so the front
//      it lands in searches all of its columns too.  A root front never reports failures (k_front_tpp
//      records zero pivots there), so a variable fails at most three times.
// Very wide fronts (> TPP_WIDE columns) first get the cheap repair -- the variable moves to the end of its
// supernode, into the last 64-column block, where the blocked kernel's complete pivoting sees every remaining
// column -- because one workgroup walking a 10^4-column front is slow; after a few passes they are flagged too.
// The repaired order and the flags are kept in the handle, so later factorizations of the same structure
// (the interior-point loop of CQP/SBLS) start from them.
constexpr int TPP_WIDE = 1024;

struct RepairPlan {
  bool reorder = false;              // `order` holds a new elimination order: re-analyse
  bool flagged = false;              // new variables were flagged: rebuild the front flags
  std::vector<int32_t> order;
};

// tppfail[v]: how often variable v has failed under whole-front pivoting (first time: to the parent front;
// again: to the root of its tree, where k_front_tpp cannot fail -- the delayed pivot that rides all the way up);
// partner[v]: the variable v was matched with when the values arrived (zero-diagonal pairs, see
// refine_order_with_values), -1 if none: a pair travels together, or the one left behind fails next.
// eager: whole-front pivoting has not rescued anything on this handle so far (every flagged column failed again): a
//        column that fails in a front of ONE 64-column block -- where the blocked kernel's complete pivoting has
//        already seen every fully-summed column -- is flagged AND moved to the parent in the same repair (one pass
//        per level of a delay cascade instead of two);
// to_root: last resort (the pass budget is used up, or nothing else is left to change): every failing variable goes to
//        the root of its tree, where k_front_tpp accepts what is left as zero pivots -- never an error for a matrix
//        that has a factorization.
RepairPlan plan_repair(const Symbolic& S, const std::vector<int32_t>& failed_pos, std::vector<uint8_t>& tppvar,
                       std::vector<uint8_t>& tppfail, std::vector<uint8_t>& force,
                       const std::vector<int32_t>& partner, int pass, bool eager = false, bool to_root = false) {
  RepairPlan rp;
  const int n = S.n, nn = S.nnodes;
  if (int(tppvar.size()) != n) tppvar.assign(n, 0);
  if (int(tppfail.size()) != n) tppfail.assign(n, 0);
  if (int(force.size()) != n) force.assign(n, 0);
  std::vector<double> key(n);
  for (int p = 0; p < n; ++p) key[p] = double(p);
  std::vector<uint8_t> nodeflag(std::max(nn, 1), 0);
  for (int s = 0; s < nn; ++s)
    for (int p = S.sptr[s]; p < S.sptr[s + 1]; ++p)
      if (tppvar[S.invp[p]]) { nodeflag[s] = 1; break; }
  for (const int p : failed_pos) {
    if (p < 0 || p >= n) continue;
    const int s = int(std::upper_bound(S.sptr.begin(), S.sptr.begin() + nn + 1, p) - S.sptr.begin()) - 1;
    if (s < 0 || s >= nn) continue;
    const int v = S.invp[p];
    if (!nodeflag[s] && !to_root && !(eager && S.ncol(s) <= NB && S.sparent[s] < nn)) {
      // the blocked kernels failed here
      const int blk = (p - S.sptr[s]) / NB, nblk = (S.ncol(s) + NB - 1) / NB;
      if (S.ncol(s) > TPP_WIDE && pass < 6) {
        int target = -1;
        if (blk < nblk - 1) target = S.sptr[s + 1] - 1;
        else if (S.sparent[s] < nn) target = S.sptr[S.sparent[s] + 1] - 1;
        if (target >= 0) {
          key[p] = double(target) + 0.5;
          rp.reorder = true;
          continue;
        }
      }
      tppvar[v] = 1;
      rp.flagged = true;
    } else {
      // whole-front pivoting failed as well (k_front_tpp reports nothing at a root)
      tppvar[v] = 1;
      rp.flagged = true;
      if (S.sparent[s] >= nn) continue;
      int anc = S.sparent[s];
      if (tppfail[v] < 255) ++tppfail[v];
      if (to_root && tppfail[v] < 2) tppfail[v] = 2;
      if (tppfail[v] > 1)
        while (S.sparent[anc] < nn) anc = S.sparent[anc];
      const double k2 = double(S.sptr[anc + 1] - 1) + 0.5;
      key[p] = std::max(key[p], k2);
      force[v] = 1;
      if (getenv("GSLS_DEBUG"))
        fprintf(stderr, "[gsls]   var %d at pos %d (front %d: %d x %d, cols %d..%d, parent %d) failed %d times under whole-front pivoting -> behind pos %d (front %d), partner %d\n",
                v, p, s, S.nrow(s), S.ncol(s), S.sptr[s], S.sptr[s + 1] - 1, S.sparent[s], int(tppfail[v]), S.sptr[anc + 1] - 1, anc,
                (int(partner.size()) == n) ? partner[v] : -1);
      const int w = (int(partner.size()) == n) ? partner[v] : -1;
      if (w >= 0 && double(S.perm[w]) < k2) {
        key[S.perm[w]] = std::max(key[S.perm[w]], k2);
        tppvar[w] = 1;
        force[w] = 1;
      }
      rp.reorder = true;
    }
  }
  if (rp.reorder) {
    std::vector<int> idx(n);
    for (int p = 0; p < n; ++p) idx[p] = p;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] < key[b]; });
    rp.order.assign(n, 0);
    for (int newpos = 0; newpos < n; ++newpos) rp.order[S.invp[idx[newpos]]] = newpos + 1;
  }
  return rp;
}

// the fronts that hold a flagged variable, for dev_set_tpp
std::vector<int> tpp_nodes(const Symbolic& S, const std::vector<uint8_t>& tppvar) {
  std::vector<int> nodes;
  if (int(tppvar.size()) != S.n) return nodes;
  for (int s = 0; s < S.nnodes; ++s)
    for (int p = S.sptr[s]; p < S.sptr[s + 1]; ++p)
      if (tppvar[S.invp[p]]) { nodes.push_back(s); break; }
  return nodes;
}

}  // namespace

extern "C" {

const char* gsls_version(void) { return "gsls 0.1 (gfx950)"; }

int gsls_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}

void gsls_default_options(gsls_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->print_level = -1;
  o->ordering = GSLS_ORDER_ND;
  o->nemin = 32;
  o->scaling = 0;
  o->action = 1;
  o->device = -1;
  o->reserved2 = 0;
  o->u = 0.01;
  o->small = 1e-20;
  o->multiplier = 1.1;
}

int gsls_create(void** handle) {
  if (!handle) return GSLS_ERROR_UNKNOWN;
  Handle* h = new (std::nothrow) Handle();
  if (!h) return GSLS_ERROR_ALLOCATION;
  std::memset(&h->last, 0, sizeof(h->last));
  *handle = h;
  return GSLS_SUCCESS;
}

int gsls_destroy(void** handle) {
  if (!handle || !*handle) return GSLS_SUCCESS;
  Handle* h = static_cast<Handle*>(*handle);
  {
    DeviceGuard g(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    dev_free(h->F);
    dev_free_coo(h->F);
    h->F.pool.reset();          // (the blocks the pool keeps are freed here, on the handle's device)
    (void)gsls_comm_destroy(h);
    for (auto& ev : h->ev)
      if (ev) (void)hipEventDestroy(ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stat_pin) (void)hipHostFree(h->stat_pin);
  }
  delete h;
  *handle = nullptr;
  return GSLS_SUCCESS;
}

int gsls_analyse(void* handle, int32_t n, const int64_t* ptr, const int32_t* row, int32_t* order,
                 const gsls_options* options, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  std::memset(inform, 0, sizeof(*inform));
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  gsls_options defo;
  if (!options) {
    gsls_default_options(&defo);
    options = &defo;
  }
  const double t0 = now();
  h->analysed = h->factored = h->dev_ready = false;
  h->have_coo = h->coo_uploaded = false;
  if (!h->keep_match_scale) h->match_scale.clear();
  h->learned = 0;
  h->shard_fast = h->shard_fast_off = false;
  h->tiny_ready = false;
  h->tiny_strikes = 0;
  h->tiny_black.clear();
  if (n < 0) return inform->flag = GSLS_ERROR_A_N_OOR;
  if (n > 0 && (!ptr || !row)) return inform->flag = GSLS_ERROR_A_PTR;
  if (n > 0 && ptr[0] != 1) return inform->flag = GSLS_ERROR_A_PTR;
  for (int j = 0; j < n; ++j)
    if (ptr[j + 1] < ptr[j]) return inform->flag = GSLS_ERROR_A_PTR;
  if (options->ordering < 0 || options->ordering > 3) return inform->flag = GSLS_ERROR_ORDER;
  int flag;
  try {
    if (n == 0) {
      h->S = Symbolic();
      flag = GSLS_SUCCESS;
    } else {
      for (int64_t k = 0; k < ptr[n] - 1; ++k)
        if (row[k] < 1 || row[k] > n) return inform->flag = GSLS_ERROR_A_ALL_OOR;
      int nemin = options->nemin;
      if (nemin <= 0) {
        // nemin = 0: the backend chooses (what SLS_initialize('gsls') leaves in control%node_amalgamation).  A tree of
        // tens of thousands of tiny fronts -- a KKT / banded-Hessian system with a couple of entries per column --
        // runs a wave per front: 24 keeps them within one wavefront (n <= 32, m <= 64) and adds few explicit zeros (2x
        // faster than 64 on the metric workload).  Everything else gets 64, the setting of rounds 1 and 2 (fewer,
        // fatter fronts for the MFMA updates).  The cheap guess (entries per column) is checked on the tree it gives;
        // if the factor does not sit in fronts of at most 64 rows after all, the analysis is repeated with 64.
        const bool guess_tiny = double(ptr[n] - 1) <= 2.5 * double(n);
        nemin = guess_tiny ? 24 : 64;
        std::vector<int32_t> keep;
        if (order) keep.assign(order, order + n);
        flag = symbolic_analyse(n, ptr, row, order, options->ordering, nemin, h->S);
        if (flag >= 0 && guess_tiny) {
          int64_t small = 0, all = 0;
          for (int sn = 0; sn < h->S.nnodes; ++sn) {
            const int64_t e = int64_t(h->S.nrow(sn)) * h->S.ncol(sn);
            all += e;
            if (h->S.nrow(sn) <= 64) small += e;
          }
          if (small * 10 < all * 9) {
            nemin = 64;
            if (order) std::copy(keep.begin(), keep.end(), order);
            flag = symbolic_analyse(n, ptr, row, order, options->ordering, nemin, h->S);
          }
        }
      } else {
        flag = symbolic_analyse(n, ptr, row, order, options->ordering, nemin, h->S);
      }
      h->ptr.assign(ptr, ptr + n + 1);
      h->row.assign(row, row + (ptr[n] - 1));
      h->diagpos.assign(n, -1);
      for (int j = 0; j < n; ++j)
        for (int64_t k = ptr[j] - 1; k < ptr[j + 1] - 1; ++k)
          if (row[k] == j + 1) { h->diagpos[j] = k; break; }
      h->tppvar.assign(n, 0);
      h->tppfail.assign(n, 0);
      h->force.assign(n, 0);
      h->partner.assign(n, -1);
      h->tpp_unflagged = false;
      h->tpp_dirty = true;
      h->nemin = nemin;
      h->own_order = (options->ordering != GSLS_ORDER_USER);
      h->preordered = false;
    }
  } catch (const std::bad_alloc&) {
    inform->stat = 1;
    return inform->flag = GSLS_ERROR_ALLOCATION;
  }
  inform->flag = flag;
  if (flag < 0) return flag;
  fill_from_symbolic(h->S, inform);
  inform->factor_bytes = 8 * (h->S.nnodes ? h->S.loff[h->S.nnodes] : 0) + 16 * int64_t(n);
  h->analysed = true;
  inform->time_analyse = now() - t0;
  h->last = *inform;
  if (h->comm) (void)gsls_comm_destroy(h);     // a new pattern: the old sharding is gone
  {
    const int cf = gsls_comm_init_env(h, options);   // GSLS_COMM_RANKS / _RANK / _ID_FILE: several GPUs, unchanged caller
    if (cf < 0) return inform->flag = cf;
  }
  return flag;
}

// stat[] of the last factorization pass: copies the counters and folds the binned ones (device side: no
// same-address atomics from tens of thousands of fronts) into st[2] (negative pivots) and st[6] (optimistic fronts)
static hipError_t read_stat(Handle* h, int32_t (&st)[16]) {
  // every factorization pass ends here, with the GPU idle until the host has seen the counters: a pinned destination
  // (one DMA, no staging copy inside the runtime) keeps that round trip short
  if (!h->stat_pin) {
    hipError_t e0 = hipHostMalloc(reinterpret_cast<void**>(&h->stat_pin), NSTAT * sizeof(int32_t), hipHostMallocDefault);
    if (e0 != hipSuccess) { h->stat_pin = nullptr; return e0; }
  }
  int32_t* all = h->stat_pin;
  hipError_t e = hipMemcpyAsync(all, h->F.stat, NSTAT * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return e;
  for (int i = 0; i < 16; ++i) st[i] = all[i];
  for (int b = 0; b < STAT_BINS; ++b) {
    st[2] += all[16 + b];
    st[6] += all[16 + STAT_BINS + b];
  }
  return hipSuccess;
}

// Saddle-point structure: a variable whose diagonal entry is exactly zero (a constraint row of a KKT
// matrix) has no pivot of its own until its neighbours are eliminated, and how soon it gets a usable
// one depends on the VALUES of the others -- an order repaired for today's values fails tomorrow.
// Ordering every such variable after all of its neighbours makes its pivot the full Schur complement
// -(a H^-1 a^T): usable for any positive definite H.  Done once, when values are first seen, and only
// if analyse chose the order itself.  Returns a gsls flag (0 also when there was nothing to do).
static int reanalyse(Handle* h, std::vector<int32_t>& order, gsls_inform* inform);

static int refine_order_with_values(Handle* h, const double* val, bool on_device, gsls_inform* inform) {
  if (!h->own_order || h->preordered) return GSLS_SUCCESS;
  h->preordered = true;
  const int n = h->S.n;
  if (n == 0) return GSLS_SUCCESS;
  const int64_t nzv = h->ptr[n] - 1;
  std::vector<double> hv;
  if (on_device) {
    hv.resize(nzv);
    // on the handle's stream: the values may just have been produced there (gsls_factor_coo maps them on the device)
    hipError_t e = h->stream ? hipMemcpyAsync(hv.data(), val, size_t(nzv) * sizeof(double), hipMemcpyDeviceToHost, h->stream)
                             : hipMemcpy(hv.data(), val, size_t(nzv) * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && h->stream) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    val = hv.data();
  }
  // "zero diagonal": no diagonal entry in the pattern, or one that is negligible beside the column's other
  // entries (a 1e-8 regularisation block would fail every threshold test as a pivot of its own, just like 0)
  std::vector<double> cmax(n, 0.0);
  for (int j = 0; j < n; ++j)
    for (int64_t k = h->ptr[j] - 1; k < h->ptr[j + 1] - 1; ++k) {
      const int i = h->row[k] - 1;
      if (i == j) continue;
      const double av = std::fabs(val[k]);
      cmax[i] = std::max(cmax[i], av);
      cmax[j] = std::max(cmax[j], av);
    }
  std::vector<char> zero(n);
  int nzero = 0;
  for (int j = 0; j < n; ++j) {
    const double dj = (h->diagpos[j] >= 0) ? std::fabs(val[h->diagpos[j]]) : 0.0;
    zero[j] = (dj == 0.0) || (dj <= 1e-8 * cmax[j]);
    nzero += zero[j];
  }
  // quasi-definite saddle points K = [H A^T; A -E] (interior-point systems with slack or regularisation blocks): a
  // variable with a NEGATIVE diagonal entry all of whose neighbours have positive ones is a constraint row as well.  As
  // a pivot of its own, -e_i passes the threshold test only while e_i is large; behind its neighbours its pivot is
  // -(e_i + a H^-1 a^T) whatever the iteration has made of e_i.  The rule looks at signs only, so the order does not
  // depend on the values of the first factorization (CQP's E shrinks by many decades between its iterations).
  std::vector<char> late(n, 0);
  int nlate = 0;
  {
    std::vector<char> negd(n, 0), posd(n, 0), ok(n, 1);
    for (int j = 0; j < n; ++j)
      if (!zero[j] && h->diagpos[j] >= 0) {
        negd[j] = val[h->diagpos[j]] < 0.0;
        posd[j] = val[h->diagpos[j]] > 0.0;
      }
    for (int j = 0; j < n; ++j)
      for (int64_t k = h->ptr[j] - 1; k < h->ptr[j + 1] - 1; ++k) {
        const int i = h->row[k] - 1;
        if (i == j || val[k] == 0.0) continue;
        if (!posd[j]) ok[i] = 0;
        if (!posd[i]) ok[j] = 0;
      }
    for (int j = 0; j < n; ++j) {
      late[j] = negd[j] && ok[j];
      nlate += late[j];
    }
  }
  if (nzero == 0 && nlate == 0) return GSLS_SUCCESS;
  const std::vector<int>& pos = h->S.perm;
  // per variable: last position among its neighbours; for zero-diagonal variables also whether any neighbour
  // has a diagonal entry, and the strongest coupling to another zero-diagonal variable
  std::vector<int> nbmax(pos.begin(), pos.end()), best(n, -1);
  std::vector<char> has_nz(n, 0);
  std::vector<double> bestv(n, 0.0);
  for (int j = 0; j < n; ++j)
    for (int64_t k = h->ptr[j] - 1; k < h->ptr[j + 1] - 1; ++k) {
      const int i = h->row[k] - 1;
      if (i == j) continue;
      nbmax[i] = std::max(nbmax[i], pos[j]);
      nbmax[j] = std::max(nbmax[j], pos[i]);
      const double av = std::fabs(val[k]);
      if (zero[i]) {
        if (!zero[j]) has_nz[i] = 1;
        else if (av > bestv[i]) { bestv[i] = av; best[i] = j; }
      }
      if (zero[j]) {
        if (!zero[i]) has_nz[j] = 1;
        else if (av > bestv[j]) { bestv[j] = av; best[j] = i; }
      }
    }
  std::vector<double> key(n);
  for (int v2 = 0; v2 < n; ++v2) key[pos[v2]] = double(pos[v2]);
  // (1) a zero-diagonal variable next to variables WITH a diagonal (constraint row of a KKT matrix): after all
  //     of its neighbours -- its pivot becomes the full Schur complement, usable for any definite Hessian
  // (2) one whose neighbours all have zero diagonals as well ([0 B; B^T 0]): it can only be eliminated in a
  //     2x2 pivot; it is placed right behind its strongest partner so that the pair shares a diagonal block
  std::vector<char> matched(n, 0);
  int moved = 0;
  for (int p2 = 0; p2 < n; ++p2) {
    const int v2 = h->S.invp[p2];
    if (late[v2]) {
      if (nbmax[v2] > pos[v2]) { key[pos[v2]] = double(nbmax[v2]) + 0.5; ++moved; }
      continue;
    }
    if (!zero[v2] || matched[v2]) continue;
    if (has_nz[v2]) {
      if (nbmax[v2] > pos[v2]) { key[pos[v2]] = double(nbmax[v2]) + 0.5; ++moved; }
    } else if (best[v2] >= 0 && !matched[best[v2]]) {
      const int b2 = best[v2];
      matched[v2] = matched[b2] = 1;
      h->partner[v2] = b2;
      h->partner[b2] = v2;
      const int first = std::min(pos[v2], pos[b2]), second = std::max(pos[v2], pos[b2]);
      key[second] = double(first) + 0.25;       // the later one of the pair moves up behind the earlier one
      h->force[h->S.invp[first]] = 1;           // ... and the two share a supernode (its parent column is the partner)
      ++moved;
    }
  }
  // (3) slack variables of an interior-point KKT system [H+D 0 A^T; 0 D_s -I; A -I 0]: a variable whose ONLY neighbour is
  //     a zero-diagonal one.  Its own entry y_i/s_i runs from huge (active constraint) to ~0 (inactive) over the
  //     iterations, so as a 1x1 pivot far from its constraint row it fails the threshold test sooner or later; placed
  //     right in front of that row, in the same supernode, the pair is a 2x2 pivot whenever the 1x1 is refused --
  //     decided inside one diagonal block, no cross-front repair.  Structural, hence independent of the values.
  {
    std::vector<int> deg(n, 0), only(n, -1);
    for (int j = 0; j < n; ++j)
      for (int64_t k = h->ptr[j] - 1; k < h->ptr[j + 1] - 1; ++k) {
        const int i = h->row[k] - 1;
        if (i == j) continue;
        ++deg[i]; only[i] = j;
        ++deg[j]; only[j] = i;
      }
    std::vector<char> taken(n, 0);
    for (int v2 = 0; v2 < n; ++v2) {
      if (zero[v2] || late[v2] || deg[v2] != 1) continue;
      const int c2 = only[v2];
      if (!zero[c2] || !has_nz[c2] || taken[c2] || matched[c2]) continue;
      taken[c2] = 1;
      const double kc = (key[pos[c2]] != double(pos[c2])) ? key[pos[c2]] : double(pos[c2]);
      if (pos[v2] == pos[c2] - 1 && kc == double(pos[c2])) continue;     // already adjacent
      key[pos[v2]] = kc - 0.25;
      h->partner[v2] = c2;
      h->partner[c2] = v2;
      h->force[v2] = 1;
      ++moved;
    }
  }
  if (moved == 0) return GSLS_SUCCESS;
  std::vector<int> idx(n);
  for (int p2 = 0; p2 < n; ++p2) idx[p2] = p2;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] < key[b]; });
  std::vector<int32_t> order(n);
  for (int newpos = 0; newpos < n; ++newpos) order[h->S.invp[idx[newpos]]] = newpos + 1;
  const int rf = reanalyse(h, order, inform);
  if (rf < 0) return rf;
  if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] %d zero-diagonal and %d negative-diagonal constraint variables ordered after their neighbours\n", nzero, nlate);
  return GSLS_SUCCESS;
}

// Re-analyse in place for a new elimination order (order repair, learned pivot sequence, pre-ordering).  The
// handle is consistent whatever happens: device plans are dropped first, and a failed analysis leaves the
// handle un-analysed instead of half-built.
static int reanalyse(Handle* h, std::vector<int32_t>& order, gsls_inform* inform) {
  h->dev_ready = false;
  h->factored = false;
  int flag2;
  try {
    flag2 = symbolic_analyse(h->S.n, h->ptr.data(), h->row.data(), order.data(), GSLS_ORDER_USER, h->nemin, h->S,
                             (int(h->force.size()) == h->S.n) ? h->force.data() : nullptr);
  } catch (const std::bad_alloc&) {
    h->analysed = false;
    return GSLS_ERROR_ALLOCATION;
  }
  if (flag2 < 0) {
    h->analysed = false;
    return flag2;
  }
  gsls_inform* dst = inform ? inform : &h->last;
  fill_from_symbolic(h->S, dst);
  dst->factor_bytes = 8 * h->S.loff[h->S.nnodes] + 16 * int64_t(h->S.n);
  if (inform) {
    const int keep = h->last.flag;
    h->last = *inform;
    h->last.flag = keep;
  }
  h->tpp_dirty = true;
  return GSLS_SUCCESS;
}

// scale_on_host: where the caller's scale vector lives when that differs from the values (gsls_factor_coo: the values
// were mapped on the device, the scale vector is still the caller's host array); -1 = where `on_device` says
static int factor_common(Handle* h, int posdef, const double* val, const double* scale, bool on_device,
                         const gsls_options* options, gsls_inform* inform, int scale_on_host = -1) {
  gsls_inform local;
  if (!inform) inform = &local;
  if (!h || !h->analysed) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  if (h->comm && h->comm_ranks > 1) {
    // the handle is one rank of a sharded system (gsls_comm_init / gsls_comm_init_env): the whole call is collective
    if (scale) { *inform = h->last; return inform->flag = GSLS_ERROR_UNIMPLEMENTED; }   // (a user scale vector is not sharded)
    return on_device ? gsls_comm_factor_dev(h, posdef, val, options, inform)
                     : gsls_comm_factor(h, posdef, val, options, inform);
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  inform->hip_error = 0;
  gsls_options defo;
  if (!options) {
    gsls_default_options(&defo);
    options = &defo;
  }
  const Symbolic& S = h->S;
  h->factored = false;
  if (S.n == 0) {
    h->factored = true;
    h->posdef = posdef != 0;
    return GSLS_SUCCESS;
  }
  if (!val) return inform->flag = GSLS_ERROR_VAL;
  const double t0 = now();
  hipError_t e = ensure_device(h, options);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  DeviceGuard g(h->device);
  if (!posdef) {
    const int rf = refine_order_with_values(h, val, on_device, inform);
    if (rf < 0) return inform->flag = rf;
  }
  DeviceFactor& F = h->F;
  // ---- the reference's internal scalings (ssids.f90:921-1030), computed on the host from these values ------------
  std::vector<double> own_scale;
  if (!scale && options->scaling > 0) {
    if (options->scaling == 3 && int(h->match_scale.size()) != S.n)
      return inform->flag = GSLS_ERROR_NO_SAVED_SCALING;       // needs gsls_analyse_matching (ssids.f90:991-994)
    const int n = S.n;
    const int64_t nzv = h->ptr[n] - 1;
    std::vector<double> hv;
    const double* v = val;
    if (on_device && options->scaling != 3) {
      hv.resize(size_t(nzv));
      e = hipMemcpyAsync(hv.data(), val, size_t(nzv) * sizeof(double), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);    // (the value map of gsls_factor_coo runs on this stream)
      if (e != hipSuccess) return fail_hip(h, inform, e);
      v = hv.data();
    }
    std::vector<int64_t> p0(n + 1);
    std::vector<int32_t> r0(static_cast<size_t>(nzv));
    for (int j = 0; j <= n; ++j) p0[j] = h->ptr[j] - 1;
    for (int64_t k = 0; k < nzv; ++k) r0[k] = h->row[k] - 1;
    own_scale.resize(n);
    int sf = 0;
    try {
      if (options->scaling == 3) own_scale = h->match_scale;     // saved by the matching-based ordering
      else if (options->scaling == 1) sf = hungarian_scale_sym(n, p0.data(), r0.data(), v, options->action != 0, own_scale.data());
      else if (options->scaling == 2) sf = auction_scale_sym(n, p0.data(), r0.data(), v, own_scale.data());
      else sf = equilib_scale_sym(n, p0.data(), r0.data(), v, own_scale.data());
    } catch (const std::bad_alloc&) {
      return inform->flag = GSLS_ERROR_ALLOCATION;
    }
    if (sf == -2) return inform->flag = GSLS_ERROR_SINGULAR;    // structurally singular and action = false (ssids.f90:944-947)
    scale = own_scale.data();      // a HOST vector whatever `on_device` says about the values (see stage_inputs)
    h->scale_host = own_scale;
  } else if (!scale) {
    h->scale_host.clear();
  }
  const double* d_val = val;
  const double* d_scale = scale;
  auto stage_inputs = [&]() -> hipError_t {
    if (!on_device) {
      // ptr/row are kept from analyse; the value count is the number of scattered entries
      const int64_t nz = F.nscatter;
      if (F.val_cap < nz) {
        if (F.val) (void)hipFree(F.val);
        F.val = nullptr;
        hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&F.val), std::max<int64_t>(nz, 1) * sizeof(double));
        if (e2 != hipSuccess) return e2;
        F.val_cap = nz;
      }
      hipError_t e2 = hipMemcpyAsync(F.val, val, nz * sizeof(double), hipMemcpyHostToDevice, h->stream);
      if (e2 != hipSuccess) return e2;
      d_val = F.val;
    }
    if (scale) {
      if (!F.scale) {
        hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&F.scale), h->S.n * sizeof(double));
        if (e2 != hipSuccess) return e2;
      }
      const bool host_vec = !on_device || !own_scale.empty() || scale_on_host == 1;
      hipError_t e2 = hipMemcpyAsync(F.scale, scale, h->S.n * sizeof(double),
                                     host_vec ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream);
      if (e2 != hipSuccess) return e2;
      d_scale = F.scale;
    }
    return hipSuccess;
  };
  // (re)build everything on the device that depends on the current tree: plans, staged inputs, front flags
  auto sync_device = [&]() -> hipError_t {
    if (!h->dev_ready) {
      hipError_t e2 = dev_upload_symbolic(h->S, F, h->stream);
      if (e2 != hipSuccess) return e2;
      h->dev_ready = true;
      h->tpp_dirty = true;
      e2 = stage_inputs();
      if (e2 != hipSuccess) return e2;
    }
    if (h->tpp_dirty) {
      hipError_t e2 = dev_set_tpp(h->S, F, posdef ? std::vector<int>() : tpp_nodes(h->S, h->tppvar), h->stream);
      if (e2 != hipSuccess) return e2;
      h->tpp_dirty = false;
    }
    return hipSuccess;
  };
  // position hints from the variable pairs of the last discovery (the analysis postorders the tree and renumbers inside
  // merged supernodes, so a pair is looked up where it sits NOW; one that is no longer adjacent loses its hint)
  auto upload_pair_hints = [&]() -> hipError_t {
    const int n = h->S.n;
    if (h->hint_pairs.empty() || n == 0) return hipSuccess;
    std::vector<uint8_t> hints(n, 0);
    int lost = 0;
    bool any2 = false;
    for (size_t k = 0; k + 1 < h->hint_pairs.size(); k += 2) {
      const int p1 = h->S.perm[h->hint_pairs[k]], p2 = h->S.perm[h->hint_pairs[k + 1]];
      if (p2 == p1 + 1) { hints[p1] = 1; any2 = true; }
      else if (p1 == p2 + 1) { hints[p2] = 1; any2 = true; }
      else ++lost;
    }
    if (getenv("GSLS_DEBUG") && lost) fprintf(stderr, "[gsls] %d of %zu 2x2 pairs are not adjacent in the analysed order\n", lost, h->hint_pairs.size() / 2);
    hipError_t e2 = hipMemcpy(F.hint, hints.data(), size_t(n), hipMemcpyHostToDevice);
    if (e2 != hipSuccess) return e2;
    F.any_hint = any2;
    return hipSuccess;
  };
  {
    const bool fresh = !h->dev_ready;
    e = sync_device();
    if (e == hipSuccess && !fresh) e = stage_inputs();
    if (e != hipSuccess) { h->dev_ready = false; return fail_hip(h, inform, e); }
  }
  h->have_scale = (scale != nullptr);
  int32_t st[16];
  int total_moved = 0;
  bool tiny_off = false;
  int tiny_repeats = 0;
  const int max_pass = 60;    // a variable fails at most three times (plan_repair); cascades end long before this
  int pending_flags = 0;      // columns the last repair flagged for whole-front pivoting (without moving them)
  // pivot-order discovery sweeps of this call (dev_discover).  Two by default: what the static kernels still fail after the
  // first (a few dozen columns on CQP's systems) is found in one more sweep where the repair loop needs three re-analyses
  // (41 -> 4 -> 1 -> 0 failing columns); a third sweep costs more than it saves (CQP N = 1e5, SLS_factorize over ten
  // iterations, four runs each: one sweep 0.80 s, two 0.70 s, three 0.95 s)
  int discovered = 0;
  static const int max_discover = getenv("GSLS_MAX_DISCOVER") ? atoi(getenv("GSLS_MAX_DISCOVER")) : 2;
  bool disc_ok = false;       // ... and its sequence was adopted
  bool just_learned = false;  // the previous pass ended in a learning round
  int learn_fail = 0;         // learning rounds of this call after which pivots failed
  for (int pass = 0;; ++pass) {
    // refactorizations of a learned order: tiny fronts whole, a wave each (k_front_tiny); if that kernel
    // meets a pivot it cannot take (stat[13]) the pass is repeated on the workgroup path
    bool any_tpp = false;
    for (const auto& c : F.tpp_cnt[0]) any_tpp |= (c > 0);
    if (any_tpp && disc_ok && h->tiny_ready && !tiny_off && !posdef && !scale) {
      // after a discovery the wave-per-front kernels stay in charge; the few fronts flagged for whole-front pivoting are
      // taken out of their hands (blacklist: assembled in HBM, extend-add and contribution by the workgroup tasks,
      // k_front_tpp for the factorization -- every workgroup kernel skips a flagged front)
      std::vector<int32_t> bl(h->tiny_black);
      for (int sn : tpp_nodes(h->S, h->tppvar)) bl.push_back(sn);
      std::sort(bl.begin(), bl.end());
      bl.erase(std::unique(bl.begin(), bl.end()), bl.end());
      if (bl.size() != h->tiny_black.size()) {
        h->tiny_black.swap(bl);
        e = dev_set_tiny_blacklist(h->S, F, h->tiny_black, h->stream);
        if (e != hipSuccess) return fail_hip(h, inform, e);
      }
      any_tpp = false;
    }
    const bool use_tiny = !posdef && !scale && h->tiny_ready && !tiny_off && !any_tpp;
    e = dev_factor(h->S, F, posdef != 0, d_val, d_scale, options->small, options->u, h->stream, use_tiny);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    e = read_stat(h, st);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    if (use_tiny && st[13] > 0) {
      // fronts the wave-per-front kernel gave up on join the blacklist (they take the workgroup kernels from
      // now on, a short extra task list per level) and the pass is repeated
      const int nf = std::min<int>(st[13], FAILCAP);
      std::vector<int32_t> nodes(nf);
      e = hipMemcpy(nodes.data(), F.tinyfail, nf * sizeof(int32_t), hipMemcpyDeviceToHost);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      h->tiny_black.insert(h->tiny_black.end(), nodes.begin(), nodes.end());
      std::sort(h->tiny_black.begin(), h->tiny_black.end());
      h->tiny_black.erase(std::unique(h->tiny_black.begin(), h->tiny_black.end()), h->tiny_black.end());
      if (getenv("GSLS_DEBUG")) {
        fprintf(stderr, "[gsls] pass %d: %d tiny fronts need pivoting (blacklist %zu), repeating\n", pass, st[13], h->tiny_black.size());
        // what the diagonal looks like now (diagnostic only)
        const int n = h->S.n;
        const int64_t nzv = h->ptr[n] - 1;
        std::vector<double> hv(static_cast<size_t>(nzv));
        if (hipMemcpy(hv.data(), d_val, size_t(nzv) * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess) {
          int nneg = 0, nzer = 0, nnan = 0;
          double amin = 1e300, amax = 0, omax = 0;
          for (int j = 0; j < n; ++j) {
            const double d = h->diagpos[j] >= 0 ? hv[h->diagpos[j]] : 0.0;
            if (d != d) ++nnan;
            if (d < 0) ++nneg;
            if (d == 0) ++nzer; else { amin = std::min(amin, std::fabs(d)); amax = std::max(amax, std::fabs(d)); }
          }
          for (int64_t k = 0; k < nzv; ++k) if (hv[k] == hv[k]) omax = std::max(omax, std::fabs(hv[k]));
          fprintf(stderr, "[gsls]   diagonal: %d negative, %d zero, %d NaN, |d| in [%.3e, %.3e], largest entry %.3e\n", nneg, nzer, nnan, amin, amax, omax);
          std::vector<double> cm(n, 0.0);
          for (int j = 0; j < n; ++j)
            for (int64_t k = h->ptr[j] - 1; k < h->ptr[j + 1] - 1; ++k) {
              const int i = h->row[k] - 1;
              if (i == j) continue;
              cm[i] = std::max(cm[i], std::fabs(hv[k]));
              cm[j] = std::max(cm[j], std::fabs(hv[k]));
            }
          int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int j = 0; j < n; ++j) {
            const double d = h->diagpos[j] >= 0 ? std::fabs(hv[h->diagpos[j]]) : 0.0;
            if (d == 0 || cm[j] == 0) continue;
            const double r = cm[j] / d;          // the multiplier a 1x1 pivot on the ORIGINAL entry would give
            int b = r <= 1 ? 0 : r <= 10 ? 1 : r <= 100 ? 2 : r <= 1e3 ? 3 : r <= 1e4 ? 4 : r <= 1e6 ? 5 : r <= 1e8 ? 6 : 7;
            hist[b]++;
          }
          fprintf(stderr, "[gsls]   max|offdiag|/|diag| per variable: <=1: %d, <=10: %d, <=100: %d, <=1e3: %d, <=1e4: %d, <=1e6: %d, <=1e8: %d, more: %d\n",
                  hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
        }
      }
      if (++tiny_repeats > 4 || st[13] > FAILCAP || h->tiny_black.size() * 10 > size_t(h->S.nnodes)) {
        tiny_off = true;                         // not a pattern for that kernel
        if (++h->tiny_strikes >= 3) h->tiny_ready = false;
      } else {
        e = dev_set_tiny_blacklist(h->S, F, h->tiny_black, h->stream);
        if (e != hipSuccess) return fail_hip(h, inform, e);
      }
      continue;
    }
    if (getenv("GSLS_DEBUG"))
      fprintf(stderr, "[gsls] pass %d (tiny %d ready %d strikes %d): blocks fast %d, pivoted %d, tpp fronts %d (left %d), failed columns %d, 2x2 %d | why: small %d inblock %d straddle %d rej2x2 %d below %d\n", pass, int(use_tiny), int(h->tiny_ready), h->tiny_strikes, st[6], st[7], st[14], st[15], st[4], st[3], st[8], st[9], st[10], st[11], st[12]);
    if (pass == 0 && !posdef) {
      // a learned pivot sequence that the next values break at once (interior-point iterations late in a run) is not
      // worth learning again: two such starts in a row and this handle stops spending passes on it
      if (st[4] > 0 && h->learned > 0) ++h->learn_strikes;
      else if (st[4] == 0) h->learn_strikes = 0;
    }
    if (pending_flags > 0) {
      // the last repair only flagged fronts for whole-front pivoting: did that rescue anything?
      if (st[15] * 10 >= pending_flags * 9) { if (++h->tpp_useless >= 2) h->eager_delay = true; }
      else h->tpp_useless = 0;
      pending_flags = 0;
    }
    if (just_learned) {
      // the learning round before this pass changed the order: if that broke pivots, this call stops learning and
      // takes the first clean pass it gets (a valid factorization; learning only serves later refactorizations)
      if (st[4] > 0) ++learn_fail;
      just_learned = false;
    }
    // (not after a discovery in this call: its sequence is a learned one already, and folding the few blocks that still
    // pivot into it has been seen to undo it -- thousands of fronts back on the blacklist, four or five more passes)
    static const bool learn_after_disc = getenv("GSLS_LEARN_AFTER_DISCOVERY") != nullptr;
    if (!posdef && st[4] == 0 && h->learned < 3 && h->learn_strikes < 2 && learn_fail == 0 && (st[7] > 0 || st[14] > 0) &&
        (!disc_ok || learn_after_disc)) {
      just_learned = true;
      // ---- learn: fold the pivot sequence the pivoting kernels chose inside their blocks / fronts into the
      // elimination order, and remember where they took 2x2 pivots, so that later factorizations of
      // this pattern (the next interior-point iterations) go through the optimistic kernel
      h->learned += 1;
      const int n = h->S.n;
      std::vector<int32_t> gp(n);
      std::vector<double> Dh(2 * size_t(n) + 4);
      e = hipMemcpy(gp.data(), F.gperm, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      e = hipMemcpy(Dh.data(), F.D, Dh.size() * sizeof(double), hipMemcpyDeviceToHost);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      std::vector<uint8_t> hints(n, 0);
      bool moved = false, any2 = false;
      std::vector<int32_t> order(n);
      std::vector<int32_t> seq(n);     // seq[slot] = variable eliminated there
      for (int i = 0; i < n; ++i) {
        seq[i] = h->S.invp[gp[i]];
        moved |= (gp[i] != i);
        if (i + 1 < n && std::isinf(Dh[2 * size_t(i) + 2])) { hints[i] = 1; any2 = true; }
      }
      // a 2x2 pivot must not straddle a 16-column stage of the optimistic kernel: pull it one slot
      // forward past a preceding 1x1 pivot of the same 64-column block
      for (int s = 0; s < h->S.nnodes; ++s)
        for (int pos = h->S.sptr[s]; pos + 1 < h->S.sptr[s + 1]; ++pos) {
          const int rel = pos - h->S.sptr[s];
          if (!hints[pos] || (rel & 15) != 15) continue;
          if ((rel & 63) == 0 || hints[pos - 1]) continue;
          if (rel >= 2 && hints[pos - 2]) continue;          // pos-1 is the second half of a pair
          const int a = seq[pos - 1];
          seq[pos - 1] = seq[pos];
          seq[pos] = seq[pos + 1];
          seq[pos + 1] = a;
          hints[pos - 1] = 1;
          hints[pos] = 0;
          moved = true;
        }
      for (int i = 0; i < n; ++i) order[seq[i]] = i + 1;
      h->hint_pairs.clear();                 // (the learned pairs replace those of an earlier discovery)
      for (int i = 0; i + 1 < n; ++i)
        if (hints[i]) { h->hint_pairs.push_back(seq[i]); h->hint_pairs.push_back(seq[i + 1]); }
      // the learned sequence is tried on the blocked kernels again: whole-front pivoting only where it is
      // needed once more
      const bool had_flags = !h->tpp_unflagged && !h->tppvar.empty() &&
                             std::any_of(h->tppvar.begin(), h->tppvar.end(), [](uint8_t f) { return f != 0; });
      if (had_flags) {
        h->tpp_unflagged = true;
        std::fill(h->tppvar.begin(), h->tppvar.end(), uint8_t(0));
        h->tpp_dirty = true;
      }
      if (moved) {
        h->tiny_black.clear();
        const int rf = reanalyse(h, order, inform);
        if (rf < 0) return inform->flag = rf;
      }
      if (moved || had_flags) {
        e = sync_device();
        if (e != hipSuccess) { h->dev_ready = false; return fail_hip(h, inform, e); }
      }
      if (moved || any2) {
        e = hipMemcpy(F.hint, hints.data(), size_t(n), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail_hip(h, inform, e);
        F.any_hint = any2;
      }
      if (moved) continue;     // factorize once more in the learned order
    }
    if (posdef || st[4] == 0) break;
    // ---- some pivots failed.  First choice (once per factorization, trees of wavefront-sized fronts): let the device
    // find the elimination sequence threshold partial pivoting WITH run-time delays gives for these values
    // (k_front_discover: one bottom-up sweep, failed columns travel to the parent inside it, as in the reference),
    // adopt it as the order + 2x2 hints, re-analyse once and factorize again on the static kernels ---------------------
    if (discovered < max_discover && !scale && !getenv("GSLS_NO_DISCOVER")) {
      ++discovered;
      std::vector<int32_t> seq;
      std::vector<uint8_t> two;
      int dstat = 1, ndel = 0;
      const double td = now();
      e = dev_discover(h->S, F, d_val, options->small, options->u, h->stream, seq, two, dstat, ndel);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      if (dstat == 0) {
        const int n = h->S.n;
        std::vector<int32_t> order(n);
        for (int i = 0; i < n; ++i) order[seq[i]] = i + 1;
        // a variable that was eliminated in another front than the one the analysis gave it must share a supernode
        // with the columns it now sits between (relaxed_supernodes: force)
        if (int(h->force.size()) != n) h->force.assign(n, 0);
        {
          const Symbolic& S0 = h->S;
          // (the new supernode ranges are not known yet: compare every variable's OLD supernode with its neighbour's)
          std::vector<int32_t> oldsn(n);
          for (int sn = 0; sn < S0.nnodes; ++sn)
            for (int pcol = S0.sptr[sn]; pcol < S0.sptr[sn + 1]; ++pcol) oldsn[S0.invp[pcol]] = sn;
          for (int i = 0; i + 1 < n; ++i)
            if (oldsn[seq[i]] < oldsn[seq[i + 1]] && S0.perm[seq[i]] + 1 != S0.perm[seq[i + 1]]) {
              // seq[i] comes from a descendant front and now precedes a column of an ancestor: a delayed pivot
              int a = oldsn[seq[i]];
              bool anc = false;
              while (a < S0.nnodes && !anc) { a = S0.sparent[a]; anc = (a == oldsn[seq[i + 1]]); }
              if (anc) h->force[seq[i]] = 1;
            }
        }
        std::fill(h->tppvar.begin(), h->tppvar.end(), uint8_t(0));
        std::fill(h->tppfail.begin(), h->tppfail.end(), uint8_t(0));
        h->tpp_dirty = true;
        // the next pass takes the wave-per-front kernels: k_front_blk eliminates hinted 2x2 pivots wherever they sit
        // (its panels are cut around them), where the 16-column stages of the workgroup kernel cannot take a pair
        // that straddles two stages -- and a sequence found by threshold pivoting has pairs everywhere
        h->tiny_ready = !getenv("GSLS_DISCOVER_WG");
        tiny_off = false;
        tiny_repeats = 0;
        h->tiny_black.clear();
        h->learned = 0;
        int rf = reanalyse(h, order, inform);
        if (rf < 0) return inform->flag = rf;
        // The analysis postorders the elimination tree, so columns that do not depend on each other may have changed
        // places: the 2x2 hints go by VARIABLE pair to wherever the pair sits now (a pair that is no longer adjacent
        // loses its hint and is left to the repair loop; the wave-per-front kernels that run next take a pair anywhere
        // in a front, so no adjustment to the 16-column stages of the workgroup kernel is made here)
        h->hint_pairs.clear();
        for (int i = 0; i + 1 < n; ++i)
          if (two[i]) { h->hint_pairs.push_back(seq[i]); h->hint_pairs.push_back(seq[i + 1]); }
        disc_ok = true;
        e = sync_device();
        if (e == hipSuccess) e = upload_pair_hints();
        if (e != hipSuccess) { h->dev_ready = false; return fail_hip(h, inform, e); }
        total_moved += ndel;
        if (getenv("GSLS_DEBUG"))
          fprintf(stderr, "[gsls] discovery: %d delayed columns, order + hints adopted in %.3f s\n", ndel, now() - td);
        continue;
      }
      if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] discovery: not applicable to this tree / these values\n");
    }
    // ---- otherwise: flag their fronts for whole-front pivoting, or (second failure) move them up ----
    const int nf = std::min<int>(st[5], FAILCAP);
    std::vector<int32_t> failed(nf);
    if (nf > 0) {
      e = hipMemcpy(failed.data(), F.faillist, nf * sizeof(int32_t), hipMemcpyDeviceToHost);
      if (e != hipSuccess) return fail_hip(h, inform, e);
    }
    std::sort(failed.begin(), failed.end());
    failed.erase(std::unique(failed.begin(), failed.end()), failed.end());
    RepairPlan rp = plan_repair(h->S, failed, h->tppvar, h->tppfail, h->force, h->partner, pass, h->eager_delay);
    if (pass >= max_pass || (!rp.reorder && !rp.flagged)) {
      // last resort: every failing variable to the root of its tree (k_front_tpp takes what cannot be eliminated
      // there as zero pivots: warning 7 and a rank below n, as the reference reports a singular matrix) -- a matrix
      // that has a factorization never leaves here with an error
      rp = plan_repair(h->S, failed, h->tppvar, h->tppfail, h->force, h->partner, pass, h->eager_delay, true);
      if (pass >= max_pass + 8 || (!rp.reorder && !rp.flagged)) {
        inform->num_delay = total_moved;
        inform->flag = GSLS_ERROR_UNKNOWN;         // (an internal inconsistency, not a property of the matrix)
        inform->time_factor = now() - t0;
        return inform->flag;
      }
    }
    if (rp.flagged && !rp.reorder) pending_flags = int(failed.size());
    total_moved += int(failed.size());
    if (!disc_ok) h->tiny_ready = false;      // (after a discovery the wave-per-front kernels keep the rest of the tree)
    if (!disc_ok || rp.reorder) h->tiny_black.clear();
    h->tpp_dirty = true;
    const double ta = now();
    if (rp.reorder) {
      h->learned = 0;       // the order changes: learn the in-block pivot sequence again afterwards
      const int rf = reanalyse(h, rp.order, inform);
      if (rf < 0) return inform->flag = rf;
    }
    const double tb = now();
    e = sync_device();
    if (e == hipSuccess && disc_ok && rp.reorder) e = upload_pair_hints();
    if (e != hipSuccess) { h->dev_ready = false; return fail_hip(h, inform, e); }
    if (getenv("GSLS_DEBUG"))
      fprintf(stderr, "[gsls] repair (%s): re-analysis %.3f s, plan + upload %.3f s\n", rp.reorder ? "order" : "flags", tb - ta, now() - tb);
  }

  if (!posdef) {
    // the order is learned when (almost) every block went through the optimistic kernels
    if (!tiny_off && !h->tiny_ready && h->tiny_strikes < 3 && st[14] == 0) h->tiny_ready = (st[7] * 50 <= st[6] + st[7]);
  }
  h->last_fast = posdef ? 0 : st[6];
  h->last_pivoted = posdef ? 0 : st[7] + st[14];
  h->posdef = posdef != 0;
  inform->num_neg = 0;
  inform->num_two = 0;
  inform->num_delay = 0;
  inform->matrix_rank = h->S.sptr[h->S.nnodes];
  inform->maxfront = std::max(h->S.maxfront, h->S.maxrow);   // cpu_iface.f90:84
  if (posdef) {
    if (st[0] != INT_MAX) {
      inform->flag = GSLS_ERROR_NOT_POS_DEF;
      h->last.flag = inform->flag;
      inform->time_factor = now() - t0;
      return inform->flag;
    }
  } else {
    inform->num_neg = st[2];
    inform->num_two = st[3];
    inform->num_delay = total_moved;
    if (st[1] > 0) {
      inform->matrix_rank -= st[1];
      if (!options->action) {
        inform->flag = GSLS_ERROR_SINGULAR;
        inform->time_factor = now() - t0;
        return inform->flag;
      }
      inform->flag = GSLS_WARNING_FACT_SINGULAR;
    }
  }
  h->factored = true;
  inform->time_factor = now() - t0;
  h->last = *inform;
  return inform->flag;
}

// ssids_analyse with values and options%ordering = 2 (ssids.f90:305-320 -> match_order_metis, spral/match_order.f90)
int gsls_analyse_matching(void* handle, int32_t n, const int64_t* ptr, const int32_t* row, const double* val,
                          int32_t* order, const gsls_options* options, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  std::memset(inform, 0, sizeof(*inform));
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  gsls_options o;
  if (options) o = *options; else gsls_default_options(&o);
  if (n < 0) return inform->flag = GSLS_ERROR_A_N_OOR;
  if (n > 0 && (!ptr || !row)) return inform->flag = GSLS_ERROR_A_PTR;
  if (n > 0 && !val) return inform->flag = GSLS_ERROR_VAL;           // (ssids.f90:306-310: ordering 2 needs val)
  if (n > 0 && !order) return inform->flag = GSLS_ERROR_ORDER;
  if (n > 0 && ptr[0] != 1) return inform->flag = GSLS_ERROR_A_PTR;
  for (int j = 0; j < n; ++j)
    if (ptr[j + 1] < ptr[j]) return inform->flag = GSLS_ERROR_A_PTR;
  if (o.ordering == GSLS_ORDER_USER) o.ordering = GSLS_ORDER_ND;     // (the compressed graph has no user order)
  if (o.ordering < 0 || o.ordering > 3) return inform->flag = GSLS_ERROR_ORDER;
  int sf = 0, npairs = 0;
  std::vector<double> scaling(std::max(n, 1), 1.0);
  if (n > 0) {
    const int64_t nz = ptr[n] - 1;
    for (int64_t k = 0; k < nz; ++k)
      if (row[k] < 1 || row[k] > n) return inform->flag = GSLS_ERROR_A_ALL_OOR;
    try {
      // lower triangle by columns, 0-based, duplicates and upper-triangle entries as the analysis treats them: entries
      // above the diagonal are mirrored (the matching works on |a_ij| of the symmetric matrix)
      std::vector<int64_t> p0(n + 1);
      std::vector<int32_t> r0(static_cast<size_t>(nz));
      for (int j = 0; j <= n; ++j) p0[j] = ptr[j] - 1;
      for (int64_t k = 0; k < nz; ++k) r0[k] = row[k] - 1;
      sf = match_order_sym(n, p0.data(), r0.data(), val, o.ordering, order, scaling.data(), &npairs);
    } catch (const std::bad_alloc&) {
      return inform->flag = GSLS_ERROR_ALLOCATION;
    }
  }
  o.ordering = GSLS_ORDER_USER;
  h->keep_match_scale = true;
  const int flag = gsls_analyse(handle, n, ptr, row, order, &o, inform);
  h->keep_match_scale = false;
  if (flag < 0) { h->match_scale.clear(); return flag; }
  scaling.resize(n);
  h->match_scale.swap(scaling);
  if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] matching-based ordering: %d of %d variables in 2x2 pairs%s\n", 2 * npairs, n,
                                    sf == 1 ? " (structurally singular)" : "");
  if (sf == 1 && inform->flag == GSLS_SUCCESS) inform->flag = GSLS_WARNING_ANAL_SINGULAR;
  h->last = *inform;
  return inform->flag;
}

int gsls_factor(void* handle, int32_t posdef, const double* val, const double* scale,
                const gsls_options* options, gsls_inform* inform) {
  if (handle) (void)gsls_set_value_part(handle, -1, nullptr, 0, 1.0);   // (registered for a gsls_factor_coo that never came)
  return factor_common(static_cast<Handle*>(handle), posdef, val, scale, false, options, inform);
}

int gsls_factor_dev(void* handle, int32_t posdef, const double* d_val, const double* d_scale,
                    const gsls_options* options, gsls_inform* inform) {
  return factor_common(static_cast<Handle*>(handle), posdef, d_val, d_scale, true, options, inform);
}

// ---- the caller's own matrix on the device (include/gsls.h; SLS_factorize's scatter sls.f90:4113-4150, the
// residual of SLS_solve_ir sls.f90:4826-4934) ------------------------------------------------------------------
int gsls_set_coo(void* handle, int64_t ne, const int32_t* row, const int32_t* col, const int32_t* map) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || ne < 0 || (ne > 0 && !map)) return GSLS_ERROR_CALL_SEQUENCE;
  // kept on the host until the first factorization needs it: analyse must work without a device
  try {
    h->coo_map.assign(map, map + ne);
    if (row && col) {
      h->coo_row.assign(row, row + ne);
      h->coo_col.assign(col, col + ne);
    } else {
      h->coo_row.clear();
      h->coo_col.clear();
    }
  } catch (const std::bad_alloc&) {
    return GSLS_ERROR_ALLOCATION;
  }
  h->have_coo = true;
  h->coo_uploaded = false;
  return GSLS_SUCCESS;
}

// upload the caller's matrix structure if that has not happened yet
static hipError_t sync_coo(Handle* h) {
  if (h->coo_uploaded || h->S.n == 0) return hipSuccess;
  hipError_t e = ensure_device(h, nullptr);
  if (e != hipSuccess) return e;
  const int64_t ne = int64_t(h->coo_map.size());
  const bool rc = !h->coo_row.empty() || ne == 0;
  e = dev_set_coo(h->F, h->S.n, h->ptr[h->S.n] - 1, ne, rc ? h->coo_row.data() : nullptr,
                  rc ? h->coo_col.data() : nullptr, h->coo_map.data(), h->stream);
  if (e == hipSuccess) h->coo_uploaded = true;
  return e;
}

static int factor_coo_common(Handle* h, int posdef, const double* val, const double* scale, bool on_device,
                             const gsls_options* options, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  if (!h || !h->analysed || !h->have_coo) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  const int nparts = on_device ? 0 : h->nparts;
  Handle::ValuePart parts[4];
  for (int k = 0; k < 4; ++k) { parts[k] = h->parts[k]; h->parts[k] = Handle::ValuePart(); }
  h->nparts = 0;                    // (a registration serves ONE factorization)
  if (h->S.n == 0) return factor_common(h, posdef, val, scale, on_device, options, inform);
  if (nparts > 0) {
    int64_t tot = 0;
    for (int k = 0; k < nparts; ++k) {
      if (parts[k].len < 0 || (parts[k].len > 0 && !parts[k].val)) tot = -1;
      if (tot >= 0) tot += parts[k].len;
    }
    if (tot != int64_t(h->coo_map.size())) {
      *inform = h->last;
      return inform->flag = GSLS_ERROR_VAL;
    }
  } else if (!val) {
    *inform = h->last;
    return inform->flag = GSLS_ERROR_VAL;
  }
  hipError_t e = ensure_device(h, options);
  if (e != hipSuccess) {
    *inform = h->last;
    return fail_hip(h, inform, e);
  }
  DeviceGuard g(h->device);
  try {
    e = sync_coo(h);
  } catch (const std::bad_alloc&) {
    *inform = h->last;
    return inform->flag = GSLS_ERROR_ALLOCATION;
  }
  if (e != hipSuccess) {
    *inform = h->last;
    return fail_hip(h, inform, e);
  }
  if (nparts > 0) {
    if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] factor_coo: values from %d registered parts\n", nparts);
    int64_t off = 0;
    for (int k = 0; k < nparts && e == hipSuccess; ++k) {
      if (parts[k].len == 0) continue;
      e = hipMemcpyAsync(h->F.coo_val + off, parts[k].val, size_t(parts[k].len) * sizeof(double), hipMemcpyHostToDevice,
                         h->stream);
      if (e == hipSuccess && parts[k].mult != 1.0) e = dev_scale_values(h->F.coo_val + off, parts[k].len, parts[k].mult, h->stream);
      off += parts[k].len;
    }
  } else if (on_device) e = hipMemcpyAsync(h->F.coo_val, val, h->F.coo_ne * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
  else e = hipMemcpyAsync(h->F.coo_val, val, h->F.coo_ne * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = dev_map_values(h->F, h->F.coo_val, h->stream);
  if (e != hipSuccess) {
    *inform = h->last;
    return fail_hip(h, inform, e);
  }
  // The caller's scale vector is handed on where it lies: factor_common stages it into F.scale itself, and does so
  // again after every re-analysis (dev_upload_symbolic frees and rebuilds the device arrays, F.scale included -- an
  // upload made here would be read back from freed memory by the first factorization of every pattern).
  return factor_common(h, posdef, h->F.valcsc, scale, true, options, inform, (scale && !on_device) ? 1 : -1);
}

int gsls_set_value_part(void* handle, int32_t part, const double* val, int64_t len, double mult) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return GSLS_ERROR_CALL_SEQUENCE;
  if (part < 0) {                   // forget the registration
    for (int k = 0; k < 4; ++k) h->parts[k] = Handle::ValuePart();
    h->nparts = 0;
    return GSLS_SUCCESS;
  }
  if (part >= 4 || len < 0 || (len > 0 && !val)) return GSLS_ERROR_VAL;
  h->parts[part].val = val;
  h->parts[part].len = len;
  h->parts[part].mult = mult;
  h->nparts = std::max(h->nparts, part + 1);
  return GSLS_SUCCESS;
}

int gsls_factor_coo(void* handle, int32_t posdef, const double* val, const double* scale,
                    const gsls_options* options, gsls_inform* inform) {
  return factor_coo_common(static_cast<Handle*>(handle), posdef, val, scale, false, options, inform);
}

int gsls_factor_coo_dev(void* handle, int32_t posdef, const double* d_val, const double* d_scale,
                        const gsls_options* options, gsls_inform* inform) {
  return factor_coo_common(static_cast<Handle*>(handle), posdef, d_val, d_scale, true, options, inform);
}

int gsls_residual(void* handle, int32_t nrhs, const double* x, int32_t ldx, const double* b, int32_t ldb,
                  double* r, int32_t ldr, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->have_coo || !h->coo_uploaded || (h->S.n > 0 && !h->F.rs_ptr)) {
    std::memset(inform, 0, sizeof(*inform));      // (needs gsls_set_coo with row / col and a gsls_factor_coo)
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  const int n = h->S.n;
  const double t0r = now();
  if (nrhs < 1 || ldx < n || ldb < n || ldr < n || (n > 0 && (!x || !b || !r))) return inform->flag = GSLS_ERROR_X_SIZE;
  if (n == 0) return GSLS_SUCCESS;
  DeviceGuard g(h->device);
  DeviceFactor& F = h->F;
  const int64_t need = 3 * int64_t(n) * nrhs;
  hipError_t e;
  if (F.rbuf_cap < need) {
    if (F.rbuf) (void)hipFree(F.rbuf);
    F.rbuf = nullptr;
    F.rbuf_cap = 0;
    e = hipMalloc(reinterpret_cast<void**>(&F.rbuf), need * sizeof(double));
    if (e != hipSuccess) return fail_hip(h, inform, e);
    F.rbuf_cap = need;
  }
  double *dx = F.rbuf, *db = F.rbuf + int64_t(n) * nrhs, *dr = F.rbuf + 2 * int64_t(n) * nrhs;
  for (int k = 0; k < nrhs; ++k) {
    e = hipMemcpyAsync(dx + int64_t(k) * n, x + int64_t(k) * ldx, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    e = hipMemcpyAsync(db + int64_t(k) * n, b + int64_t(k) * ldb, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
  }
  e = dev_residual(F, n, nrhs, dx, n, db, n, dr, n, h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  for (int k = 0; k < nrhs; ++k) {
    e = hipMemcpyAsync(r + int64_t(k) * ldr, dr + int64_t(k) * n, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
  }
  e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] residual: %.3f ms\n", (now() - t0r) * 1e3);
  return inform->flag;
}

static int solve_common(Handle* h, int job, int nrhs, double* x, int ldx, bool on_device,
                        gsls_inform* inform, const double* b_dev = nullptr, bool enqueue_only = false);

// SLS_solve_ir (sls.f90:4770-4949) with every vector resident in HBM: b in, x out, nothing else crosses the bus
int gsls_solve_ir(void* handle, double* x, int32_t max_refinements, double residual_absolute,
                  double residual_relative, int32_t* iterations, const gsls_options*, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->factored || !h->have_coo || !h->coo_uploaded || (h->S.n > 0 && !h->F.rs_ptr)) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  if (iterations) *iterations = 0;
  const int n = h->S.n;
  if (n == 0) return GSLS_SUCCESS;
  if (!x) return inform->flag = GSLS_ERROR_X_SIZE;
  DeviceGuard g(h->device);
  DeviceFactor& F = h->F;
  const int64_t need = 3 * int64_t(n) + 2;
  hipError_t e;
  if (F.rbuf_cap < need) {
    if (F.rbuf) (void)hipFree(F.rbuf);
    F.rbuf = nullptr;
    F.rbuf_cap = 0;
    e = hipMalloc(reinterpret_cast<void**>(&F.rbuf), need * sizeof(double));
    if (e != hipSuccess) return fail_hip(h, inform, e);
    F.rbuf_cap = need;
  }
  double *dX = F.rbuf, *dB = F.rbuf + n, *dR = F.rbuf + 2 * int64_t(n);
  unsigned long long* dmax = reinterpret_cast<unsigned long long*>(F.rbuf + 3 * int64_t(n));
  auto max_abs = [&](const double* v, double& out) -> hipError_t {
    hipError_t e2 = dev_max_abs(n, v, dmax, h->stream);
    if (e2 != hipSuccess) return e2;
    unsigned long long bits = 0;
    e2 = hipMemcpyAsync(&bits, dmax, sizeof(bits), hipMemcpyDeviceToHost, h->stream);
    if (e2 != hipSuccess) return e2;
    e2 = hipStreamSynchronize(h->stream);
    std::memcpy(&out, &bits, sizeof(out));
    return e2;
  };
  e = hipMemcpyAsync(dB, x, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(dX, 0, n * sizeof(double), h->stream);
  double residual_zero = 0.0, residual = 0.0;
  // (every max_abs is a host round trip: the norm of b only where a relative tolerance needs it)
  if (e == hipSuccess && residual_relative > 0.0 && max_refinements > 0) e = max_abs(dB, residual_zero);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  for (int iter = 0; iter <= std::max(max_refinements, 0); ++iter) {
    if (iterations) *iterations = iter;
    gsls_inform si;
    // stream-ordered, no host synchronisation: the solves, the update and the residual follow each other on the handle's
    // stream; the first solve reads b where it lies (solve_common: b_dev) and writes into dR, the later ones refine dR
    const int f = solve_common(h, GSLS_SOLVE_JOB_ALL, 1, dR, n, true, &si, iter == 0 ? dB : nullptr, true);
    if (f < 0) {
      *inform = si;
      return f;
    }
    e = dev_vec_add(n, dX, dR, h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    if (iter >= max_refinements) break;          // (no step may follow: its residual would serve nothing)
    e = dev_residual(F, n, 1, dX, n, dB, n, dR, n, h->stream);
    if (e == hipSuccess) e = max_abs(dR, residual);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    if (residual < std::max(residual_absolute, residual_relative * residual_zero)) break;
  }
  e = hipMemcpyAsync(x, dX, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  return inform->flag;
}

static int solve_common(Handle* h, int job, int nrhs, double* x, int ldx, bool on_device,
                        gsls_inform* inform, const double* b_dev, bool enqueue_only) {
  gsls_inform local;
  if (!inform) inform = &local;
  if (!h || !h->analysed || !h->factored) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  const Symbolic& S = h->S;
  if (job < GSLS_SOLVE_JOB_ALL || job > GSLS_SOLVE_JOB_DIAG_BWD) return inform->flag = GSLS_ERROR_JOB_OOR;
  if (h->posdef && (job == GSLS_SOLVE_JOB_DIAG || job == GSLS_SOLVE_JOB_DIAG_BWD))
    return inform->flag = GSLS_ERROR_JOB_OOR;  // ssids.f90:1205-1210
  if (nrhs < 1 || ldx < S.n || (!x && S.n > 0)) return inform->flag = GSLS_ERROR_X_SIZE;
  if (S.n == 0) return GSLS_SUCCESS;
  if (h->comm && h->comm_ranks > 1) {
    // one rank of a sharded system: full solves only, column by column, the whole solution on every rank
    if (job != GSLS_SOLVE_JOB_ALL) return inform->flag = GSLS_ERROR_JOB_OOR;
    if (b_dev && b_dev != x && on_device) {
      DeviceGuard g0(h->device);
      hipError_t e0 = hipMemcpyAsync(x, b_dev, (size_t(ldx) * (nrhs - 1) + S.n) * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
      if (e0 != hipSuccess) return fail_hip(h, inform, e0);
    }
    for (int c = 0; c < nrhs; ++c) {
      double* xc = x + int64_t(c) * ldx;
      int f = on_device ? gsls_comm_solve_dev(h, xc, inform) : gsls_comm_solve(h, xc, inform);
      if (f >= 0 && on_device) f = gsls_comm_collect_dev(h, xc, inform);
      if (f < 0) return f;
    }
    return inform->flag;
  }
  const double t0 = now();
  DeviceGuard g(h->device);
  DeviceFactor& F = h->F;
  hipError_t e;
  double* d_x = x;
  const int64_t xelems = int64_t(ldx) * (nrhs - 1) + S.n;
  if (!on_device) {
    if (F.xhost_cap < xelems) {        // kept between calls: an allocation per solve costs more than the solve
      if (F.xhost) (void)hipFree(F.xhost);
      F.xhost = nullptr;
      F.xhost_cap = 0;
      e = hipMalloc(reinterpret_cast<void**>(&F.xhost), xelems * sizeof(double));
      if (e != hipSuccess) return fail_hip(h, inform, e);
      F.xhost_cap = xelems;
    }
    e = hipMemcpyAsync(F.xhost, x, xelems * sizeof(double), hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    d_x = F.xhost;
  }
  e = dev_solve(S, F, h->posdef, job, nrhs, d_x, ldx, h->have_scale ? F.scale : nullptr, h->stream, h->ev,
                on_device ? b_dev : nullptr);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  if (!on_device) {
    e = hipMemcpyAsync(x, F.xhost, xelems * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
  }
  if (enqueue_only && on_device) {
    // device operands: the solve is stream-ordered work on the handle's stream -- whatever the caller enqueues next on this
    // handle (the next factorization of an interior-point loop) follows it without the GPU waiting for the host in between
    h->kt_pending = true;
    inform->time_solve = now() - t0;
    inform->solve_bytes = 2 * 8 * S.num_factor + (h->posdef ? 0 : 16 * int64_t(S.n)) + 32 * int64_t(S.n);
    return inform->flag;
  }
  e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  h->kt_pending = false;
  float ms = 0;
  static const bool phase_events = getenv("GSLS_SOLVE_PHASES") != nullptr;
  if (phase_events) {
    if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) h->kt_fwd = ms * 1e-3;
    if (hipEventElapsedTime(&ms, h->ev[1], h->ev[2]) == hipSuccess) h->kt_diag = ms * 1e-3;
    if (hipEventElapsedTime(&ms, h->ev[2], h->ev[3]) == hipSuccess) h->kt_bwd = ms * 1e-3;
  } else {                        // the whole sweep in `fwd` (no events between the phases: they cost ~10 us)
    h->kt_diag = h->kt_bwd = 0.0;
    if (hipEventElapsedTime(&ms, h->ev[0], h->ev[3]) == hipSuccess) h->kt_fwd = ms * 1e-3;
  }
  (void)hipGetLastError();      // (an event that was never recorded must not surface as the next call's error)
  inform->time_solve = now() - t0;
  if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] solve job %d nrhs %d: %.3f ms\n", job, nrhs, inform->time_solve * 1e3);
  inform->solve_bytes = 2 * 8 * S.num_factor + (h->posdef ? 0 : 16 * int64_t(S.n)) + 32 * int64_t(S.n);
  return inform->flag;
}

int gsls_solve_dev_rhs(void* handle, int32_t job, int32_t nrhs, const double* d_b, double* d_x, int32_t ldx,
                       const gsls_options*, gsls_inform* inform) {
  return solve_common(static_cast<Handle*>(handle), job, nrhs, d_x, ldx, true, inform, d_b, true);
}

int gsls_solve(void* handle, int32_t job, int32_t nrhs, double* x, int32_t ldx, const gsls_options*,
               gsls_inform* inform) {
  return solve_common(static_cast<Handle*>(handle), job, nrhs, x, ldx, false, inform);
}

int gsls_solve_dev(void* handle, int32_t job, int32_t nrhs, double* d_x, int32_t ldx,
                   const gsls_options*, gsls_inform* inform) {
  return solve_common(static_cast<Handle*>(handle), job, nrhs, d_x, ldx, true, inform);
}

// ---- multi-GPU: elimination-tree sharding (anal.f90:284-459, 569-590) -------------
// Every rank analyses the same matrix, then calls gsls_shard(nranks, rank).  The two exchange steps are
// the caller's (torch.distributed / RCCL): see galahad_amd/shard.py.
int gsls_shard(void* handle, int32_t nranks, int32_t rank, int64_t* xchg_factor_elems,
               int64_t* xchg_solve_elems) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed) return GSLS_ERROR_CALL_SEQUENCE;
  if (nranks < 1 || rank < 0 || rank >= nranks) return GSLS_ERROR_CALL_SEQUENCE;
  shard_tree(h->S, nranks);
  if (nranks > 1 && !getenv("GSLS_SHARD_FULL_LAYOUT")) shard_layout(h->S, rank);   // this rank's fronts and blocks only
  h->F.myrank = rank;
  h->dev_ready = false;   // plans are rebuilt on the next factorization
  h->factored = false;
  int64_t ce = 0, ve = 0;
  for (int c : h->S.cutroots) {
    const int64_t cm = h->S.nrow(c) - h->S.ncol(c);
    ce += cm * cm;
    ve += cm;
  }
  if (xchg_factor_elems) *xchg_factor_elems = ce + 24;   // + three status blocks of 8 (k_stat_to_xchg)
  if (xchg_solve_elems) *xchg_solve_elems = std::max<int64_t>(std::max<int64_t>(ve, h->S.n), 1);
  return GSLS_SUCCESS;
}

// phase 1: factorize the subtrees this rank owns, pack the cut roots' contribution blocks into
// d_xchg (zeros for the other ranks' roots); the caller then SUMs d_xchg over ranks (at least onto
// rank 0).  phase 2: rank 0 factorizes the top part (no-op elsewhere).  inform holds THIS rank's
// counts (num_neg, num_two, matrix_rank deficiency); the caller adds them up.
int gsls_shard_factor_dev(void* handle, int32_t phase, int32_t posdef, const double* d_val, double* d_xchg,
                          const gsls_options* options, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || h->S.nranks < 2 || h->S.owner.empty()) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  inform->hip_error = 0;
  gsls_options defo;
  if (!options) {
    gsls_default_options(&defo);
    options = &defo;
  }
  const Symbolic& S = h->S;
  if (phase == 1) h->factored = false;
  if (S.n == 0) {
    h->factored = true;
    h->posdef = posdef != 0;
    return GSLS_SUCCESS;
  }
  if (!d_val || !d_xchg) return inform->flag = GSLS_ERROR_VAL;
  const double t0 = now();
  hipError_t e = ensure_device(h, options);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  DeviceGuard g(h->device);
  if (!h->dev_ready) {
    e = dev_upload_symbolic(S, h->F, h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    h->dev_ready = true;
    h->tpp_dirty = true;
  }
  if (h->tpp_dirty) {
    e = dev_set_tpp(S, h->F, posdef ? std::vector<int>() : tpp_nodes(S, h->tppvar), h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    h->tpp_dirty = false;
  }
  h->have_scale = false;
  e = dev_shard_factor(S, h->F, phase, posdef != 0, d_val, d_xchg, options->small, options->u, h->stream, h->shard_fast);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  int32_t st[16];
  e = read_stat(h, st);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  h->posdef = posdef != 0;
  inform->num_neg = inform->num_two = inform->num_delay = 0;
  inform->matrix_rank = S.n;
  inform->maxfront = std::max(S.maxfront, S.maxrow);
  inform->time_factor = now() - t0;
  if (posdef) {
    if (st[0] != INT_MAX) return inform->flag = GSLS_ERROR_NOT_POS_DEF;
  } else {
    if (st[4] > 0) {   // pivots want to leave their front: the caller gathers every rank's list
      inform->num_delay = st[4];   // (gsls_shard_failed), repairs the order on all ranks alike
      return inform->flag;         // (gsls_shard_repair) and starts again at phase 1
    }
    inform->num_neg = st[2];
    inform->num_two = st[3];
    if (st[1] > 0) {
      inform->matrix_rank -= st[1];
      if (!options->action) return inform->flag = GSLS_ERROR_SINGULAR;
      inform->flag = GSLS_WARNING_FACT_SINGULAR;
    }
  }
  if (phase == 2) {
    h->factored = true;
    h->last = *inform;
  }
  return inform->flag;
}

// phases 1-4 of one full solve (job 0) for one right-hand side; d_x (n) holds the right-hand side on
// every rank before phase 1 and the solution on every rank after phase 4.  Between the phases the
// caller: SUMs d_xchg[0:V) after 1, BROADCASTs d_xchg[0:n) from rank 0 after 2, SUMs d_xchg[0:n)
// after 3 (V = total contribution-vector length of the cut roots; d_xchg has xchg_solve_elems).
int gsls_shard_solve_dev(void* handle, int32_t phase, double* d_x, double* d_xchg, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->factored || h->S.nranks < 2) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  if (h->S.n == 0) return GSLS_SUCCESS;
  if (!d_x || !d_xchg) return inform->flag = GSLS_ERROR_X_SIZE;
  DeviceGuard g(h->device);
  hipError_t e = dev_shard_solve(h->S, h->F, phase, h->posdef, d_x, d_xchg, h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  return inform->flag;
}

// Pivots that failed inside the fronts this rank factorized (pivot positions of the current
// elimination order, at most GSLS_FAILCAP of them); valid after a factor phase that reported
// inform.num_delay > 0.
int gsls_shard_failed(void* handle, int32_t* nfailed, int32_t* failed) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->dev_ready || !nfailed) return GSLS_ERROR_CALL_SEQUENCE;
  DeviceGuard g(h->device);
  int32_t st[16];
  hipError_t e = hipMemcpy(st, h->F.stat, sizeof(st), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return GSLS_ERROR_HIP;
  const int nf = (st[4] > 0) ? std::min<int>(st[5], FAILCAP) : 0;
  *nfailed = nf;
  if (nf > 0 && failed) {
    e = hipMemcpy(failed, h->F.faillist, nf * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return GSLS_ERROR_HIP;
  }
  return GSLS_SUCCESS;
}

// Apply the elimination-order repair of gsls_factor (failed pivots move to the end of their front or
// behind their parent's columns) for the union of all ranks' failed pivots, re-analyse and re-shard.
// Every rank must pass the same list; the result is identical on every rank.
int gsls_shard_repair(void* handle, int32_t nfailed, const int32_t* failed_in, int64_t* xchg_factor_elems,
                      int64_t* xchg_solve_elems) {
  if (handle) static_cast<Handle*>(handle)->shard_fast = false;      // the order changes: not before a clean pass again
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || h->S.nranks < 2 || nfailed < 0 || (nfailed > 0 && !failed_in))
    return GSLS_ERROR_CALL_SEQUENCE;
  std::vector<int32_t> failed(failed_in, failed_in + nfailed);
  std::sort(failed.begin(), failed.end());
  failed.erase(std::unique(failed.begin(), failed.end()), failed.end());
  const int round = h->shard_repairs++;
  RepairPlan rp = plan_repair(h->S, failed, h->tppvar, h->tppfail, h->force, h->partner, round);
  if (round >= 40 || (!rp.reorder && !rp.flagged))       // last resort, as in gsls_factor: to the roots (zero pivots there)
    rp = plan_repair(h->S, failed, h->tppvar, h->tppfail, h->force, h->partner, round, false, true);
  if (!rp.reorder && !rp.flagged) return GSLS_ERROR_UNKNOWN;
  const int nranks = h->S.nranks, rank = h->F.myrank;
  if (rp.reorder) {
    const int flag = reanalyse(h, rp.order, nullptr);
    if (flag < 0) return flag;
  }
  h->tpp_dirty = true;
  return gsls_shard(handle, nranks, rank, xchg_factor_elems, xchg_solve_elems);
}

// ---- the exchange inside the library: RCCL on the handle's stream (one process per GPU) ---------------------------
// The reference drives every device from ONE ssids_factor call (fkeep.F90:99-174, contribution blocks handed over
// through host memory, contrib.f90:20-33).  Here every rank calls the same gsls_comm_* entry point and the cut
// roots' blocks / vectors go device to device: ncclReduce onto rank 0 (each element is non-zero on exactly one rank:
// the owner's ingest over its xGMI links, not a ring), ncclBroadcast of the cut roots' z-vectors back.  A Fortran
// host needs nothing but a way to hand 128 bytes (the communicator id of rank 0) to the other ranks.
#define NCCLCHK(call)                                                            \
  do {                                                                           \
    ncclResult_t r__ = (call);                                                   \
    if (r__ != ncclSuccess) { if (inform) { inform->flag = GSLS_ERROR_HIP; inform->hip_error = 10000 + int(r__); } return GSLS_ERROR_HIP; } \
  } while (0)

int gsls_comm_unique_id(char* id128) {
  if (!id128) return GSLS_ERROR_CALL_SEQUENCE;
  static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");
  ncclUniqueId uid;
  if (ncclGetUniqueId(&uid) != ncclSuccess) return GSLS_ERROR_HIP;
  std::memcpy(id128, &uid, sizeof(uid));
  return GSLS_SUCCESS;
}

static int comm_buffers(Handle* h) {
  int64_t ce = 0, ve = 0;
  const int f = gsls_shard(h, h->comm_ranks, h->comm_rank, &ce, &ve);
  if (f < 0) return f;
  int64_t cut = 0;
  for (int c : h->S.cutroots) cut += h->S.nrow(c) - h->S.ncol(c);
  h->cx_cut_elems = cut;
  auto grow = [&](double*& p, int64_t& cap, int64_t need) -> bool {
    if (cap >= need) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&p), need * sizeof(double)) != hipSuccess) return false;
    cap = need;
    return true;
  };
  if (!grow(h->cx_factor, h->cx_factor_elems, ce) || !grow(h->cx_solve, h->cx_solve_elems, std::max<int64_t>(ve, 16)))
    return GSLS_ERROR_ALLOCATION;
  if (!h->cx_fail && hipMalloc(reinterpret_cast<void**>(&h->cx_fail), size_t(h->comm_ranks) * (1 + GSLS_FAILCAP) * sizeof(int32_t)) != hipSuccess)
    return GSLS_ERROR_ALLOCATION;
  return GSLS_SUCCESS;
}

// every rank, after gsls_analyse of the same matrix (and, for an order the backend chose itself, gsls_refine_order
// with the same values): joins the communicator and deals the subtrees
int gsls_comm_init(void* handle, int32_t nranks, int32_t rank, const char* id128, const gsls_options* options) {
  Handle* h = static_cast<Handle*>(handle);
  gsls_inform* inform = nullptr;
  if (!h || !h->analysed || !id128 || nranks < 2 || rank < 0 || rank >= nranks) return GSLS_ERROR_CALL_SEQUENCE;
  hipError_t e = ensure_device(h, options);
  if (e != hipSuccess) return fail_hip(h, nullptr, e);
  DeviceGuard g(h->device);
  if (h->comm) { (void)ncclCommDestroy(h->comm); h->comm = nullptr; }
  ncclUniqueId uid;
  std::memcpy(&uid, id128, sizeof(uid));
  NCCLCHK(ncclCommInitRank(&h->comm, nranks, uid, rank));
  h->comm_ranks = nranks;
  h->comm_rank = rank;
  return comm_buffers(h);
}

int gsls_comm_destroy(void* handle) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return GSLS_SUCCESS;
  DeviceGuard g(h->device);
  if (h->comm) (void)ncclCommDestroy(h->comm);
  h->comm = nullptr;
  for (void* p : {static_cast<void*>(h->cx_factor), static_cast<void*>(h->cx_solve), static_cast<void*>(h->cx_fail),
                  static_cast<void*>(h->cx_hostx)})
    if (p) (void)hipFree(p);
  h->cx_factor = h->cx_solve = h->cx_hostx = nullptr;
  h->cx_hostx_cap = 0;
  h->cx_fail = nullptr;
  h->cx_factor_elems = h->cx_solve_elems = 0;
  return GSLS_SUCCESS;
}

// The whole sharded factorization as ONE call per rank: subtrees, reduce of the cut roots' contribution blocks (with
// the counters riding behind them), top part on rank 0, broadcast of the 16 status words; failed pivots are gathered
// and repaired identically on every rank.  inform is the same on every rank (totals).
int gsls_comm_factor_dev(void* handle, int32_t posdef, const double* d_val, const gsls_options* options,
                         gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->comm || h->S.nranks < 2 || h->S.owner.empty()) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  gsls_options defo;
  if (!options) {
    gsls_default_options(&defo);
    options = &defo;
  }
  const double t0 = now();
  DeviceGuard g(h->device);
  int total_moved = 0;
  for (int pass = 0; pass < 60; ++pass) {
    *inform = h->last;
    inform->flag = GSLS_SUCCESS;
    inform->hip_error = 0;
    h->factored = false;
    if (h->S.n == 0) { h->factored = true; h->posdef = posdef != 0; return GSLS_SUCCESS; }
    if (!d_val) return inform->flag = GSLS_ERROR_VAL;
    hipError_t e = hipSuccess;
    if (!h->dev_ready) {
      e = dev_upload_symbolic(h->S, h->F, h->stream);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      h->dev_ready = true;
      h->tpp_dirty = true;
    }
    if (h->tpp_dirty) {
      e = dev_set_tpp(h->S, h->F, posdef ? std::vector<int>() : tpp_nodes(h->S, h->tppvar), h->stream);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      h->tpp_dirty = false;
    }
    h->have_scale = false;
    const int64_t ce = h->F.xchgC_elems;
    e = dev_shard_factor(h->S, h->F, 1, posdef != 0, d_val, h->cx_factor, options->small, options->u, h->stream, h->shard_fast);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    NCCLCHK(ncclReduce(h->cx_factor, h->cx_factor, size_t(ce + 8), ncclDouble, ncclSum, 0, h->comm, h->stream));
    e = dev_shard_factor(h->S, h->F, 2, posdef != 0, d_val, h->cx_factor, options->small, options->u, h->stream, h->shard_fast);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    NCCLCHK(ncclBroadcast(h->cx_factor + ce, h->cx_factor + ce, 16, ncclDouble, 0, h->comm, h->stream));
    double st[16];
    e = hipMemcpyAsync(st, h->cx_factor + ce, sizeof(st), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    h->posdef = posdef != 0;
    inform->matrix_rank = h->S.n;
    inform->maxfront = std::max(h->S.maxfront, h->S.maxrow);
    inform->num_neg = inform->num_two = 0;
    inform->num_delay = total_moved;
    if (!posdef && st[5] + st[13] > 0) {      // a front on the wave-per-front path wanted pivoting: again without it
      h->shard_fast = false;
      h->shard_fast_off = true;
      continue;
    }
    if (posdef) {
      if (st[0] + st[8] > 0) { inform->time_factor = now() - t0; return inform->flag = GSLS_ERROR_NOT_POS_DEF; }
    } else if (st[1] + st[9] > 0) {
      h->shard_fast = false;
      // failed pivots somewhere: every rank contributes its list, all apply the same repair and go again
      int32_t stat[16];
      e = hipMemcpy(stat, h->F.stat, sizeof(stat), hipMemcpyDeviceToHost);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      std::vector<int32_t> mine(1 + GSLS_FAILCAP, 0);
      const int nf = (stat[4] > 0) ? std::min<int>(stat[5], FAILCAP) : 0;
      mine[0] = nf;
      if (nf > 0) {
        e = hipMemcpy(mine.data() + 1, h->F.faillist, nf * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail_hip(h, inform, e);
      }
      int32_t* slot = h->cx_fail + size_t(h->comm_rank) * (1 + GSLS_FAILCAP);
      e = hipMemcpyAsync(slot, mine.data(), mine.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      NCCLCHK(ncclAllGather(slot, h->cx_fail, 1 + GSLS_FAILCAP, ncclInt32, h->comm, h->stream));
      std::vector<int32_t> all(size_t(h->comm_ranks) * (1 + GSLS_FAILCAP));
      e = hipMemcpyAsync(all.data(), h->cx_fail, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) return fail_hip(h, inform, e);
      std::vector<int32_t> failed;
      for (int r = 0; r < h->comm_ranks; ++r) {
        const int32_t* p = all.data() + size_t(r) * (1 + GSLS_FAILCAP);
        failed.insert(failed.end(), p + 1, p + 1 + std::min<int>(p[0], GSLS_FAILCAP));
      }
      int64_t a = 0, b = 0;
      const int rf = gsls_shard_repair(h, int32_t(failed.size()), failed.data(), &a, &b);
      if (rf < 0) { inform->time_factor = now() - t0; return inform->flag = rf; }
      total_moved += int(failed.size());
      const int cb = comm_buffers(h);
      if (cb < 0) return inform->flag = cb;
      continue;
    }
    if (!posdef) {
      // no block needed pivoting anywhere: the next factorization may take the wave-per-front kernels
      h->shard_fast = !h->shard_fast_off && (st[6] + st[14] == 0);
      inform->num_neg = int(st[2] + st[10]);
      inform->num_two = int(st[3] + st[11]);
      const int nzero = int(st[4] + st[12]);
      if (nzero > 0) {
        inform->matrix_rank -= nzero;
        if (!options->action) { inform->time_factor = now() - t0; return inform->flag = GSLS_ERROR_SINGULAR; }
        inform->flag = GSLS_WARNING_FACT_SINGULAR;
      }
    }
    h->factored = true;
    inform->time_factor = now() - t0;
    h->last = *inform;
    return inform->flag;
  }
  inform->time_factor = now() - t0;
  return inform->flag = GSLS_ERROR_UNKNOWN;      // (60 collective repair rounds: gsls_shard_repair sends everything
                                                 //  that still fails to the roots from round 40 on)
}

// One full solve (job 0, one right-hand side): d_x holds b on every rank on entry; on exit the entries of the
// variables this rank eliminated (rank 0: also the top part's) hold the solution.  What crosses the links: the cut
// roots' contribution vectors up (reduce onto rank 0), their z-vectors down (broadcast) -- no O(n) collective.
int gsls_comm_solve_dev(void* handle, double* d_x, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->factored || !h->comm || h->S.nranks < 2) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  if (h->S.n == 0) return GSLS_SUCCESS;
  if (!d_x) return inform->flag = GSLS_ERROR_X_SIZE;
  const double t0 = now();
  DeviceGuard g(h->device);
  const size_t V = size_t(std::max<int64_t>(h->cx_cut_elems, 1));
  hipError_t e = dev_shard_solve(h->S, h->F, 1, h->posdef, d_x, h->cx_solve, h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  NCCLCHK(ncclReduce(h->cx_solve, h->cx_solve, V, ncclDouble, ncclSum, 0, h->comm, h->stream));
  e = dev_shard_solve(h->S, h->F, 2, h->posdef, d_x, h->cx_solve, h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  NCCLCHK(ncclBroadcast(h->cx_solve, h->cx_solve, V, ncclDouble, 0, h->comm, h->stream));
  e = dev_shard_solve(h->S, h->F, 3, h->posdef, d_x, h->cx_solve, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  inform->time_solve = now() - t0;
  return inform->flag;
}

// optional: every rank ends with the WHOLE solution in d_x (one O(n) all-reduce; for callers that need it)
int gsls_comm_collect_dev(void* handle, double* d_x, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->factored || !h->comm || h->S.nranks < 2) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  if (h->S.n == 0) return GSLS_SUCCESS;
  DeviceGuard g(h->device);
  hipError_t e = dev_shard_solve(h->S, h->F, 4, h->posdef, d_x, h->cx_solve, h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  NCCLCHK(ncclAllReduce(h->cx_solve, h->cx_solve, size_t(h->S.n), ncclDouble, ncclSum, h->comm, h->stream));
  e = dev_shard_solve(h->S, h->F, 5, h->posdef, d_x, h->cx_solve, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  return inform->flag;
}

// ---- the same with HOST arrays: what the Fortran binding (GSLS_comm_factor / GSLS_comm_solve) and, through
// gsls_comm_init_env, every SLS caller uses.  val: the sorted lower-by-columns values of gsls_factor; x: b on entry,
// the WHOLE solution on exit on every rank (one extra all-reduce: a host caller wants the vector, not a shard of it).
int gsls_comm_factor(void* handle, int32_t posdef, const double* val, const gsls_options* options, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->comm || h->S.nranks < 2) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  if (h->S.n == 0) return gsls_comm_factor_dev(handle, posdef, nullptr, options, inform);
  if (!val) { *inform = h->last; return inform->flag = GSLS_ERROR_VAL; }
  DeviceGuard g(h->device);
  DeviceFactor& F = h->F;
  const int64_t nz = h->ptr[h->S.n] - 1;
  if (F.val_cap < nz) {
    if (F.val) (void)hipFree(F.val);
    F.val = nullptr;
    F.val_cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&F.val), std::max<int64_t>(nz, 1) * sizeof(double));
    if (e != hipSuccess) return fail_hip(h, inform, e);
    F.val_cap = nz;
  }
  hipError_t e = hipMemcpyAsync(F.val, val, nz * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  return gsls_comm_factor_dev(handle, posdef, F.val, options, inform);
}

int gsls_comm_solve(void* handle, double* x, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !h->factored || !h->comm || h->S.nranks < 2) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  if (h->S.n == 0) { *inform = h->last; return inform->flag = GSLS_SUCCESS; }
  if (!x) { *inform = h->last; return inform->flag = GSLS_ERROR_X_SIZE; }
  DeviceGuard g(h->device);
  const size_t bytes = size_t(h->S.n) * sizeof(double);
  if (h->cx_hostx_cap < h->S.n) {
    if (h->cx_hostx) (void)hipFree(h->cx_hostx);
    h->cx_hostx = nullptr;
    h->cx_hostx_cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->cx_hostx), bytes);
    if (e != hipSuccess) return fail_hip(h, inform, e);
    h->cx_hostx_cap = h->S.n;
  }
  hipError_t e = hipMemcpyAsync(h->cx_hostx, x, bytes, hipMemcpyHostToDevice, h->stream);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  int f = gsls_comm_solve_dev(handle, h->cx_hostx, inform);
  if (f < 0) return f;
  const int f2 = gsls_comm_collect_dev(handle, h->cx_hostx, inform);
  if (f2 < 0) return f2;
  e = hipMemcpy(x, h->cx_hostx, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  return inform->flag = f;
}

// One sparse system over several GPUs WITHOUT any change to the caller: N copies of the same host program (one per
// GPU, any launcher) run with
//     GSLS_COMM_RANKS = N     GSLS_COMM_RANK = 0 .. N-1     GSLS_COMM_ID_FILE = a path all of them can reach
// and every handle joins the communicator after its analyse (rank 0 publishes the 128-byte id through the file);
// from then on gsls_factor / gsls_factor_coo / gsls_solve (one right-hand side, job 0) on that handle are the
// sharded calls above -- the reference drives its devices from one ssids_factor call the same way
// (src/ssids/fkeep.F90:99-174).  Returns 0 when the variables are not set (nothing happens), 1 when joined.
int gsls_comm_init_env(void* handle, const gsls_options* options) {
  const char* er = getenv("GSLS_COMM_RANKS");
  const char* ek = getenv("GSLS_COMM_RANK");
  const char* ef = getenv("GSLS_COMM_ID_FILE");
  if (!er || !ek || !ef) return 0;
  const int nranks = atoi(er), rank = atoi(ek);
  if (nranks < 2) return 0;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || rank < 0 || rank >= nranks || !*ef) return GSLS_ERROR_CALL_SEQUENCE;
  // a handle analysed again (new pattern) joins again under a fresh id: the file name carries a per-handle counter
  static int generation = 0;
  const std::string path = std::string(ef) + "." + std::to_string(generation++);
  char id[128];
  if (rank == 0) {
    const int f = gsls_comm_unique_id(id);
    if (f < 0) return f;
    const std::string tmp = path + ".tmp";
    FILE* fp = fopen(tmp.c_str(), "wb");
    if (!fp) return GSLS_ERROR_CALL_SEQUENCE;
    const size_t w = fwrite(id, 1, sizeof(id), fp);
    fclose(fp);
    if (w != sizeof(id) || rename(tmp.c_str(), path.c_str()) != 0) return GSLS_ERROR_CALL_SEQUENCE;
  } else {
    bool got = false;
    for (int tries = 0; tries < 6000 && !got; ++tries) {       // up to ten minutes for rank 0's analyse
      FILE* fp = fopen(path.c_str(), "rb");
      if (fp) {
        got = fread(id, 1, sizeof(id), fp) == sizeof(id);
        fclose(fp);
      }
      if (!got) usleep(100000);
    }
    if (!got) return GSLS_ERROR_CALL_SEQUENCE;
  }
  const int f = gsls_comm_init(handle, nranks, rank, id, options);
  return f < 0 ? f : 1;
}

// what this handle allocates on its device for the factors and for the contribution-block arena, in doubles: the
// whole tree on one device; after gsls_shard(nranks > 1) only the fronts and blocks of this rank (shard_layout)
int gsls_get_layout_sizes(void* handle, int64_t* factor_elems, int64_t* arena_elems) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed) return GSLS_ERROR_CALL_SEQUENCE;
  const int nn = h->S.nnodes;
  if (factor_elems) *factor_elems = nn ? h->S.loff[nn] : 0;
  if (arena_elems) *arena_elems = nn ? h->S.coff[nn] : 0;
  return GSLS_SUCCESS;
}

// owner rank of every supernode (-1: top part) and the cut roots, for callers that want to inspect
// the partition; either pointer may be NULL
int gsls_shard_get(void* handle, int32_t* owner, int32_t* ncut, int32_t* cutroots) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed) return GSLS_ERROR_CALL_SEQUENCE;
  if (owner)
    for (size_t i = 0; i < h->S.owner.size(); ++i) owner[i] = h->S.owner[i];
  if (ncut) *ncut = int32_t(h->S.cutroots.size());
  if (cutroots)
    for (size_t i = 0; i < h->S.cutroots.size(); ++i) cutroots[i] = h->S.cutroots[i] + 1;
  return GSLS_SUCCESS;
}

// d[i] = diagonal of the Cholesky factor in pivot order (NumericSubtree.hxx:418-427)
int gsls_enquire_posdef(void* handle, double* d, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->factored) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  if (!h->posdef) return inform->flag = GSLS_ERROR_NOT_LLT;
  const Symbolic& S = h->S;
  if (S.n == 0) return GSLS_SUCCESS;
  DeviceGuard g(h->device);
  std::vector<double> Lh(h->F.L_elems);
  hipError_t e = hipMemcpy(Lh.data(), h->F.L, Lh.size() * sizeof(double), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  for (int s = 0; s < S.nnodes; ++s)
    for (int j = 0; j < S.ncol(s); ++j) d[S.sptr[s] + j] = Lh[S.loff[s] + int64_t(j) * S.ldl[s] + j];
  return GSLS_SUCCESS;
}

// piv_order[var] = +-(pivot position), d(2,n) inverted pivots in pivot order
// (NumericSubtree.hxx:428-462, fkeep.F90:321-377)
int gsls_enquire_indef(void* handle, int32_t* piv_order, double* d, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->factored) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  if (h->posdef) return inform->flag = GSLS_ERROR_NOT_LDLT;
  const Symbolic& S = h->S;
  if (S.n == 0) return GSLS_SUCCESS;
  DeviceGuard g(h->device);
  std::vector<double> Dh(2 * size_t(S.n) + 2, 0.0);
  std::vector<int32_t> gp(S.n);
  hipError_t e = hipMemcpy(Dh.data(), h->F.D, 2 * size_t(S.n) * sizeof(double), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  e = hipMemcpy(gp.data(), h->F.gperm, size_t(S.n) * sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  for (int p = 0; p < S.n; ++p) {
    const bool two_first = (p + 1 < S.n) && std::isinf(Dh[2 * size_t(p) + 2]);
    const bool two_second = std::isinf(Dh[2 * size_t(p)]);
    if (piv_order) piv_order[S.invp[gp[p]]] = (two_first || two_second) ? -(p + 1) : (p + 1);
    if (d) {
      d[2 * size_t(p)] = two_second ? Dh[2 * size_t(p) + 1] : Dh[2 * size_t(p)];
      d[2 * size_t(p) + 1] = two_first ? Dh[2 * size_t(p) + 1] : 0.0;
    }
  }
  return GSLS_SUCCESS;
}

// d(2,n) in the layout gsls_enquire_indef returns (NumericSubtree.hxx:465-477)
int gsls_alter(void* handle, const double* d, gsls_inform* inform) {
  gsls_inform local;
  if (!inform) inform = &local;
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->factored) {
    std::memset(inform, 0, sizeof(*inform));
    return inform->flag = GSLS_ERROR_CALL_SEQUENCE;
  }
  *inform = h->last;
  inform->flag = GSLS_SUCCESS;
  if (h->posdef) return inform->flag = GSLS_ERROR_NOT_LDLT;
  const Symbolic& S = h->S;
  if (S.n == 0) return GSLS_SUCCESS;
  DeviceGuard g(h->device);
  std::vector<double> Dh(2 * size_t(S.n));
  hipError_t e = hipMemcpy(Dh.data(), h->F.D, Dh.size() * sizeof(double), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  for (int p = 0; p < S.n; ++p) {
    const bool two_second = std::isinf(Dh[2 * size_t(p)]);
    if (two_second) {
      Dh[2 * size_t(p) + 1] = d[2 * size_t(p)];      // d22 lives next to the inf marker
    } else {
      Dh[2 * size_t(p)] = d[2 * size_t(p)];
      Dh[2 * size_t(p) + 1] = d[2 * size_t(p) + 1];
    }
  }
  e = hipMemcpy(h->F.D, Dh.data(), Dh.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail_hip(h, inform, e);
  return GSLS_SUCCESS;
}

int gsls_get_symbolic_sizes(void* handle, int32_t* nnodes, int64_t* rlist_len, int64_t* nlist_len) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed) return GSLS_ERROR_CALL_SEQUENCE;
  if (nnodes) *nnodes = h->S.nnodes;
  if (rlist_len) *rlist_len = h->S.nnodes ? h->S.rptr[h->S.nnodes] : 0;
  if (nlist_len) *nlist_len = h->S.nnodes ? h->S.nptr[h->S.nnodes] : 0;
  return GSLS_SUCCESS;
}

int gsls_get_symbolic(void* handle, int32_t* sptr, int32_t* sparent, int64_t* rptr, int32_t* rlist,
                      int64_t* nptr, int64_t* nlist) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed) return GSLS_ERROR_CALL_SEQUENCE;
  const Symbolic& S = h->S;
  const int nn = S.nnodes;
  if (nn == 0) return GSLS_SUCCESS;
  if (sptr) for (int i = 0; i <= nn; ++i) sptr[i] = S.sptr[i] + 1;
  if (sparent) for (int i = 0; i < nn; ++i) sparent[i] = S.sparent[i] + 1;
  if (rptr) for (int i = 0; i <= nn; ++i) rptr[i] = S.rptr[i] + 1;
  if (rlist) for (int64_t i = 0; i < S.rptr[nn]; ++i) rlist[i] = S.rlist[i] + 1;
  if (nptr) for (int i = 0; i <= nn; ++i) nptr[i] = S.nptr[i] + 1;
  if (nlist) for (int64_t i = 0; i < 2 * S.nptr[nn]; ++i) nlist[i] = S.nlist[i] + 1;
  return GSLS_SUCCESS;
}

// order[var] = 1-based pivot position in the elimination order currently held by the handle (the one
// given to / computed by analyse, as repaired by later factorizations)
int gsls_get_order(void* handle, int32_t* order) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !order) return GSLS_ERROR_CALL_SEQUENCE;
  for (int i = 0; i < h->S.n; ++i) order[i] = h->S.perm[i] + 1;
  return GSLS_SUCCESS;
}

// host utility: the scaling gsls_factor* would compute for options.scaling = kind (1, 2, 4) from these values, without a
// handle or a device -- hungarian_scale_sym / auction_scale_sym / equilib_scale_sym of src/spral/scaling.f90.
// ptr / row 1-based, lower triangle by columns.  Returns 0, 1 (structurally singular, scaled as the reference does
// with action = true) or GSLS_ERROR_SINGULAR (action = false: identity scaling).
int gsls_scale_sym(int32_t kind, int32_t n, const int64_t* ptr, const int32_t* row, const double* val, int32_t action,
                   double* scaling) {
  if (n < 0 || !ptr || !scaling || (n > 0 && (!row || !val))) return GSLS_ERROR_CALL_SEQUENCE;
  if (kind != 1 && kind != 2 && kind != 4) return GSLS_ERROR_UNIMPLEMENTED;
  try {
    const int64_t nz = ptr[n] - 1;
    std::vector<int64_t> p0(n + 1);
    std::vector<int32_t> r0(static_cast<size_t>(nz));
    for (int j = 0; j <= n; ++j) p0[j] = ptr[j] - 1;
    for (int64_t k = 0; k < nz; ++k) r0[k] = row[k] - 1;
    int sf;
    if (kind == 1) sf = gsls::hungarian_scale_sym(n, p0.data(), r0.data(), val, action != 0, scaling);
    else if (kind == 2) sf = gsls::auction_scale_sym(n, p0.data(), r0.data(), val, scaling);
    else sf = gsls::equilib_scale_sym(n, p0.data(), r0.data(), val, scaling);
    return sf == -2 ? GSLS_ERROR_SINGULAR : sf;
  } catch (const std::bad_alloc&) {
    return GSLS_ERROR_ALLOCATION;
  }
}

// multi-GPU drivers that run the phases themselves (gsls_shard_*): after a factorization whose status words report no
// pivoted block on any rank, `on` = 1 lets the next one take the wave-per-front kernels; 0 switches them off again
// (to be called with the same value on every rank).  gsls_comm_factor_dev does this itself.
int gsls_shard_fast(void* handle, int32_t on) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed) return GSLS_ERROR_CALL_SEQUENCE;
  h->shard_fast = on != 0;
  return GSLS_SUCCESS;
}

// the scaling factors the last factorization computed itself (gsls_options.scaling = 1, 2, 4): what ssids_factor returns
// in its optional `scale` argument (ssids.f90:955-958); GSLS_ERROR_CALL_SEQUENCE if that factorization did not scale
int gsls_get_scaling(void* handle, double* scaling) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->factored || !scaling || int(h->scale_host.size()) != h->S.n) return GSLS_ERROR_CALL_SEQUENCE;
  std::copy(h->scale_host.begin(), h->scale_host.end(), scaling);
  return GSLS_SUCCESS;
}

// Give the handle the matrix values before the first factorization so that it can refine the elimination order it
// chose itself (zero-diagonal variables after their neighbours, see refine_order_with_values).  gsls_factor does this
// on its own; callers that split the tree over several GPUs call it on every rank BEFORE gsls_shard, because the
// partition depends on the order.  d_val: device pointer, CSC order as for gsls_factor_dev.
int gsls_refine_order_dev(void* handle, const double* d_val, gsls_inform* inform) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !d_val) return GSLS_ERROR_CALL_SEQUENCE;
  gsls_inform local = h->last;
  if (!inform) inform = &local;
  else *inform = h->last;
  DeviceGuard g(h->device);
  const int rf = refine_order_with_values(h, d_val, true, inform);
  inform->flag = rf;
  return rf;
}

// the same with the values in host memory (needs no device: the refinement is host integer work)
int gsls_refine_order(void* handle, const double* val, gsls_inform* inform) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed || !val) return GSLS_ERROR_CALL_SEQUENCE;
  gsls_inform local = h->last;
  if (!inform) inform = &local;
  else *inform = h->last;
  const int rf = refine_order_with_values(h, val, false, inform);
  inform->flag = rf;
  return rf;
}

// how the last LDL^T factorization went: diagonal blocks / tiny fronts done optimistically, blocks that needed
// the complete-pivoting kernel, tiny fronts currently kept off the wave-per-front kernel
int gsls_get_factor_stats(void* handle, int32_t* fast_blocks, int32_t* pivoted_blocks, int32_t* tiny_blacklist) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h || !h->analysed) return GSLS_ERROR_CALL_SEQUENCE;
  if (fast_blocks) *fast_blocks = h->last_fast;
  if (pivoted_blocks) *pivoted_blocks = h->last_pivoted;
  if (tiny_blacklist) *tiny_blacklist = int32_t(h->tiny_black.size());
  return GSLS_SUCCESS;
}

void* gsls_get_stream(void* handle) {
  Handle* h = static_cast<Handle*>(handle);
  return h ? static_cast<void*>(h->stream) : nullptr;
}

int gsls_last_solve_kernel_seconds(void* handle, double* fwd, double* diag, double* bwd) {
  Handle* h = static_cast<Handle*>(handle);
  if (!h) return GSLS_ERROR_CALL_SEQUENCE;
  if (h->kt_pending) {            // (an enqueue-only solve: wait for its last event now)
    DeviceGuard g(h->device);
    float ms = 0;
    if (h->ev[3] && hipEventSynchronize(h->ev[3]) == hipSuccess) {
      static const bool phase_events = getenv("GSLS_SOLVE_PHASES") != nullptr;
      if (phase_events) {
        if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) h->kt_fwd = ms * 1e-3;
        if (hipEventElapsedTime(&ms, h->ev[1], h->ev[2]) == hipSuccess) h->kt_diag = ms * 1e-3;
        if (hipEventElapsedTime(&ms, h->ev[2], h->ev[3]) == hipSuccess) h->kt_bwd = ms * 1e-3;
      } else {
        h->kt_diag = h->kt_bwd = 0.0;
        if (hipEventElapsedTime(&ms, h->ev[0], h->ev[3]) == hipSuccess) h->kt_fwd = ms * 1e-3;
      }
    }
    (void)hipGetLastError();
    h->kt_pending = false;
  }
  if (fwd) *fwd = h->kt_fwd;
  if (diag) *diag = h->kt_diag;
  if (bwd) *bwd = h->kt_bwd;
  return GSLS_SUCCESS;
}

}  // extern "C"
