// Device-side data layout and launcher prototypes of the gsls backend (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <unordered_map>
#include <vector>

#include "gsls_internal.hpp"

namespace gsls {

constexpr int NB = 64;   // block-column width of the panel factorization
constexpr int RB = 64;   // row-chunk height handled by one workgroup
constexpr int TS = 64;   // contribution-block tile edge
constexpr int BIG_N = 256;    // fronts wider / taller than this use the blocked multi-launch solve
constexpr int BIG_M = 4096;
constexpr int STAT_BINS = 64;            // per-front counters are spread over this many addresses (no same-address atomics)
constexpr int NSTAT = 16 + 2 * STAT_BINS;  // stat[16..): negative-pivot bins, then optimistic-front bins
constexpr int FAILCAP = 16384;  // capacity of the failed-pivot report of one factorization pass

// One front.  L block: m x n column-major, leading dimension ld, at L + loff.
// Contribution block: (m-n) x (m-n) column-major (lower triangle meaningful) at C + coff.
struct NodeDesc {
  int32_t m, n, ld, sptr;      // sptr = first pivot position of the node
  int32_t cbeg, cend;          // children in clist[cbeg..cend)
  int32_t parent, iblk;        // iblk = index of the node's first 64-column block (L11^-T arena)
  int64_t loff, coff;
  int64_t roff;                // offset of the node's row list in rlist
  int64_t moff;                // offset of the node's (m-n) child->parent map / contribution vector
};

// unit of work of the panel kernel: block column `step` of node `node`, row chunk `chunk`
struct PanelTask {
  int32_t node, step, chunk, pad;
};
// unit of work of the contribution kernel: tile (ti,tj), ti>=tj, of node's contribution block
struct TileTask {
  int32_t node, ti, tj, pad;
};

struct BigStep {
  int b, trsv_begin, trsv_cnt, gemv_begin, gemv_cnt;
};

struct LevelPlan {
  int node_begin, node_end;                 // range in lvlnodes
  std::vector<int> panel_rows;              // per step: tallest panel (rows, <= 128) among the step's diag tasks
  std::vector<int> panel_begin, panel_cnt;  // per step: range in the PanelTask array
  int tile_begin, tile_cnt;                 // range in the TileTask array
  int tinyc_begin, tinyc_cnt;               // fronts with a tiny contribution block (k_contrib_tiny)
  int tf_begin = 0, tf_cnt = 0;             // whole tiny fronts (k_front_wave), only in planT
  int tf_cls_cnt[5] = {0, 0, 0, 0, 0};      // ... by width class (24 / 28 / 32 / 48 / 64 columns), in that order
  int tf_cls_maxm[5] = {0, 0, 0, 0, 0};     // tallest front of each class (sizes the LDS triangle)
  int pull_begin, pull_cnt;                 // extend-add tasks of the level (k_assemble_pull)
  int small_begin, small_cnt, small_maxn, small_maxm;   // solve: one-workgroup fronts
  int tiny32_cnt = 0;                       // ... of which the first tiny32_cnt have n <= 32
  int tiny_cnt;                             // ... of which the first tiny_cnt are tiny (n <= 64, m - n <= 64)
  int big_begin, big_cnt;                                // solve: blocked multi-launch fronts
  std::vector<BigStep> bigsteps;
};

// per level: tiny fronts blacklisted from k_front_tiny (ranges in bl_ptasks / bl_ttasks / bl_tctasks)
struct BlLevel {
  int pbeg = 0, np = 0, tbeg = 0, nt = 0, tcbeg = 0, ntc = 0, rows = 0, pullbeg = 0, npull = 0;
};

// Device blocks of one handle, kept across the re-analyses of a factorization (order repair, learning, discovery rebuild
// some sixty arrays each time): a freed block waits here, by size class, for the next upload of that size instead of
// going through hipFree (a device-wide synchronisation each) and hipMalloc.  Per handle, so a block is only ever reused
// by work that is enqueued later on the handle's own stream.
struct DevPool {
  std::multimap<size_t, void*> idle;            // size class -> block
  std::unordered_map<void*, size_t> size;      // every block handed out or idle
  size_t idle_bytes = 0;
  ~DevPool();
};

struct DeviceFactor {
  std::shared_ptr<DevPool> pool;                // (survives dev_free; released with the handle)
  // symbolic (uploaded once per analyse)
  NodeDesc* nodes = nullptr;
  int32_t* rlist = nullptr;
  int32_t* cmap = nullptr;
  int32_t* clist = nullptr;
  int32_t* lvlnodes = nullptr;
  void* pullsegs = nullptr;        // extend-add: PullSeg / PullTask lists (gsls_device.hip)
  void* pulltasks = nullptr;
  void* tinyctasks = nullptr;
  void* tftasks = nullptr;
  uint8_t* tinyskip = nullptr;     // per node: 1 = not for k_front_tiny (blacklisted)
  int32_t* tinyfail = nullptr;     // nodes k_front_tiny gave up on in the last pass (count in stat[13])
  uint8_t* tppflag = nullptr;      // per node: 1 = factorized by k_front_tpp (pivot search across the whole front)
  int32_t* tpplist = nullptr;      // ... those nodes, per plan kind and level
  std::vector<int> tpp_begin[3], tpp_cnt[3];
  PanelTask* bl_ptasks = nullptr;
  TileTask* bl_ttasks = nullptr;
  void* bl_tctasks = nullptr;
  void* bl_pullsegs = nullptr;     // extend-add of the blacklisted fronts (k_front_wave assembles the others itself)
  void* bl_pulltasks = nullptr;
  int bl_count = 0;
  uint32_t* gdst = nullptr;        // k_front_wave: extend-add as a gather (GatherLists in gsls_device.hip)
  int32_t* gbeg = nullptr;
  int64_t* gsrc = nullptr;
  int32_t* aloc = nullptr;         // ... and the position of every entry of A in its front's LDS triangle
  int64_t* asrc_wg = nullptr;      // A -> L scatter restricted to the workgroup fronts of planT
  int64_t* adst_wg = nullptr;
  int64_t nscatter_wg = 0;
  const double* cur_val = nullptr; // the values of the factorization in flight
  bool any_hint = false;           // the handle holds learned 2x2 pivots (F.hint has non-zero entries)
  double* xp_mr = nullptr;         // multi-column solves: up to 8 permuted vectors, xs_mr elements apart,
  double* cvec_mr = nullptr;       // and 8 copies of the contribution vectors, cs_mr apart
  int64_t xs_mr = 0, cs_mr = 0;
  // several right-hand sides through one launch of the single-column kernels (struct Cols in gsls_device.hip):
  // MC_MAX sets of work vectors, the sets mc_s* elements apart
  static constexpr int MC_MAX = 16;
  double *mc_xp = nullptr, *mc_xs = nullptr, *mc_cvec = nullptr, *mc_ybuf = nullptr, *mc_part = nullptr;
  int64_t mc_sx = 0, mc_scv = 0;
  int64_t part_elems = 0;
  std::vector<BlLevel> bl_level;
  int32_t* smallnodes = nullptr;
  void* stasks = nullptr;          // SolveTask per entry of smallnodes (same indexing)
  int32_t* gth_ptr = nullptr;      // forward solve: per row of every small front, range in gth_src
  int64_t* gth_src = nullptr;      // ... the cvec entries (children's contributions) that add into the row
  int32_t* bignodes = nullptr;
  void* bigtrsv = nullptr;
  void* biggemv = nullptr;
  double* ybuf = nullptr;      // y of the big-front solve path, by pivot slot
  double* part = nullptr;      // partial sums of the transposed GEMV (64 per task)
  int64_t* asrc = nullptr;     // A -> L scatter: source index in val
  int64_t* adst = nullptr;     //                 destination element in L
  int32_t* arow = nullptr;     // pivot positions (row, col) of each scattered entry, for scaling
  int32_t* acol = nullptr;
  PanelTask* ptasks = nullptr;
  TileTask* ttasks = nullptr;
  int32_t* invp = nullptr;     // position -> variable
  int32_t* gperm = nullptr;    // pivot slot -> analyse-time position (numerical pivoting)
  int64_t nscatter = 0;
  std::vector<LevelPlan> plan;    // every front (single device)
  std::vector<LevelPlan> planT;   // LDL^T refactorizations: tiny fronts go to k_front_tiny, the rest as in `plan`
  std::vector<LevelPlan> planW;   // wave tier active: the fronts it does not cover (more than 64 rows)
  std::vector<LevelPlan> planA;   // multi-GPU: the subtrees this rank owns
  std::vector<LevelPlan> planB;   // multi-GPU: the top part (run by rank 0 after the exchange)
  std::vector<LevelPlan> planAT, planBT;   // ... the same with the tiny fronts on the wave-per-front kernels
  bool sharded = false;
  int myrank = 0;
  // exchange buffers: contribution blocks / contribution vectors of the cut roots, packed in
  // S.cutroots order (zeros for roots another rank owns); the buffers themselves belong to the caller
  int64_t xchgC_elems = 0, xchgV_elems = 0;
  void* segC = nullptr;           // Segment lists for pack/unpack
  void* segV = nullptr;
  void* segZ = nullptr;           // ... and the cut roots' rows n..m-1 in rlist (z-vectors of the backward sweep)
  int nseg = 0;
  int32_t* posowner = nullptr;    // pivot position -> owner rank of its front (-1 top)
  // wave tier of the LDL^T solves (fronts of at most 64 rows; gsls_device.hip, "WAVE TIER")
  bool wave = false;
  void* wtasks = nullptr;         // WTask per covered front, group by group, postorder inside a group
  void* wgroups = nullptr;        // WGroup records in launch order (by stage)
  void* wpacks = nullptr;         // WPack per task (pack kernel)
  int wtask_cnt = 0;
  std::vector<int> wstage_begin, wstage_cnt, wstage_narrow;   // per stage: group range, and how many of its first
                                                              // groups hold fronts of at most 32 columns only
  // the last stages, a handful of fronts each, run in one launch (k_wsolve_tail): first such stage (-1: none), its
  // tasks and the element ranges of their images / gather lists
  int wtail_k0 = -1, wtail_tbeg = 0, wtail_tcnt = 0;
  int64_t wtail_lf0 = 0, wtail_lb0 = 0, wtail_gp0 = 0, wtail_gp1 = 0, wtail_gs0 = 0, wtail_gs1 = 0;
  // pivot-order discovery (k_front_discover): built on first use, freed with the rest
  void* disc_tasks = nullptr;
  uint32_t* disc_arc = nullptr;
  int32_t *disc_ddelay = nullptr, *disc_dvar = nullptr, *disc_pseq = nullptr, *disc_pcnt = nullptr, *disc_flags = nullptr;
  uint8_t* disc_ptwo = nullptr;
  double *disc_arena = nullptr, *disc_scratch = nullptr;
  int32_t* disc_list = nullptr;             // per level: wave fronts, then workgroup fronts
  std::vector<int> disc_lvl_wide;           // ... how many of the latter
  std::vector<int64_t> disc_host;           // (poff, mcap) per front
  int64_t disc_pcap = 0;
  int wimg_units = 64;            // LDS staging area per wave of the wide backward launches (16-byte units)
  int32_t* wpull2 = nullptr;      // dense form of the gather lists for fronts with at most two sources per row
  int32_t* wperm_list = nullptr;  // positions the bottom stage's gather launch permutes for the launches behind it
  int wperm_cnt = 0;
  bool pure_state = false;        // the last factorization pass ran on the wave-per-front kernels only (dev_factor)
  std::vector<int> wstage_ndepth; // per stage: LDS slots its narrow runs use (1 + the deepest myslot / pslot)
  std::vector<int> wstage_unit;   // per stage: first task if every run is one front in task order, else -1
  int32_t* wgth_ptr = nullptr;    // gather lists of the covered fronts (rows -> children's contribution entries)
  int64_t* wgth_src = nullptr;
  int32_t* wnont = nullptr;       // fronts the tier does not cover (for the D solve)
  int wnont_cnt = 0;
  double* Lf = nullptr;           // packed forward / backward images of the covered fronts
  double* Lb = nullptr;
  int64_t Lf_elems = 0, Lb_elems = 0;
  double* xs = nullptr;           // forward result of the covered fronts by pivot slot (job ALL)
  int32_t* gvar = nullptr;        // pivot slot -> variable, refreshed by every factorization
  // the caller's own matrix (gsls_set_coo): value map by destination, and the full symmetric matrix by rows
  int64_t coo_ne = 0, coo_nz = 0, nscatter_coo = 0;  // entries of the caller's storage / of the full symmetric row structure
  int64_t* mv_ptr = nullptr;       // per CSC value position: range in mv_src
  int32_t* mv_src = nullptr;       // ... the caller's entries that are placed / added there, in entry order
  int64_t* rs_ptr = nullptr;       // per row of A: range in rs_col / rs_src
  int32_t* rs_col = nullptr;
  int32_t* rs_src = nullptr;       // ... the caller's entry that holds the value
  double* coo_val = nullptr;       // the caller's values of the last gsls_factor_coo
  double* valcsc = nullptr;        // the mapped CSC values
  double* rbuf = nullptr;          // x, b, r staging of gsls_residual
  int64_t rbuf_cap = 0;
  // contribution arena: (offset, length) chunks to zero before each level (Symbolic::czptr / czoff / czlen, split)
  void* cztasks = nullptr;
  std::vector<int> cz_begin, cz_cnt;
  // numeric
  double* L = nullptr;
  double* Linv = nullptr;        // Cholesky only: L11^-T of every 64-column block, nblk64 x 64 x 64
  int64_t nblk64 = 0;
  uint8_t* hint = nullptr;       // LDL^T: 1 at the first position of a 2x2 pivot learned from an earlier factorization
  int32_t* fastok = nullptr;     // LDL^T: 1 = the optimistic block pass succeeded (per 64-column block)
  double* C = nullptr;         // contribution arena
  double* D = nullptr;         // 2*n inverted pivots in pivot order (indefinite)
  double* val = nullptr;       // staging for host-supplied values
  double* scale = nullptr;     // staging for host-supplied scaling (variable order)
  double* xp = nullptr;        // permuted solution / rhs workspace (n * nrhs_cap)
  double* cvec = nullptr;      // per-node contribution vectors for the forward solve
  double* xhost = nullptr;     // staging for host x
  int64_t xhost_cap = 0;
  int32_t* faillist = nullptr; // analyse-time positions of pivots that failed in the last pass
  int32_t* stat = nullptr;     // [0] first failing pivot position+1 (posdef) / flag, [1] zero pivots,
                               // [2] num_neg, [3] num_two, [4] delays
  int64_t L_elems = 0, C_elems = 0, cvec_elems = 0;
  int nrhs_cap = 0;
  int64_t val_cap = 0;
};

// ---- launchers (gsls_device.hip) -----------------------------------------------------------------
hipError_t dev_upload_symbolic(const Symbolic& S, DeviceFactor& F, hipStream_t st);
void dev_free(DeviceFactor& F);
hipError_t dev_factor(const Symbolic& S, DeviceFactor& F, bool posdef, const double* d_val,
                      const double* d_scale, double small, double u, hipStream_t st, bool use_tiny = false);
hipError_t dev_set_coo(DeviceFactor& F, int n, int64_t nzcsc, int64_t ne, const int32_t* row, const int32_t* col,
                       const int32_t* map, hipStream_t st);
hipError_t dev_map_values(DeviceFactor& F, const double* d_val_in, hipStream_t st);
hipError_t dev_scale_values(double* d_val, int64_t n, double mult, hipStream_t st);
hipError_t dev_residual(DeviceFactor& F, int n, int nrhs, const double* d_x, int ldx, const double* d_b, int ldb,
                        double* d_r, int ldr, hipStream_t st);
void dev_free_coo(DeviceFactor& F);
// x += r ; *out (device, 8 bytes, zeroed by the call) = max |v_i| as the bit pattern of a non-negative double
hipError_t dev_vec_add(int n, double* d_x, const double* d_r, hipStream_t st);
hipError_t dev_max_abs(int n, const double* d_v, unsigned long long* d_out, hipStream_t st);
hipError_t dev_set_tpp(const Symbolic& S, DeviceFactor& F, const std::vector<int>& nodes, hipStream_t st);
// the elimination sequence threshold partial pivoting WITH run-time delays finds for these values (gsls_device.hip,
// "PIVOT-ORDER DISCOVERY"): seq[k] = variable of the k-th pivot, two[k] = 1 on the first column of a 2x2 pivot;
// status 0 = found, 1 = not applicable here (a front beyond a wavefront, too many delays), the arrays are then unset
hipError_t dev_discover(const Symbolic& S, DeviceFactor& F, const double* d_val, double small, double u, hipStream_t st,
                        std::vector<int32_t>& seq, std::vector<uint8_t>& two, int& status, int& ndelayed);
hipError_t dev_set_tiny_blacklist(const Symbolic& S, DeviceFactor& F, const std::vector<int>& nodes, hipStream_t st);
hipError_t dev_solve(const Symbolic& S, DeviceFactor& F, bool posdef, int job, int nrhs, double* d_x,
                     int ldx, const double* d_scale, hipStream_t st, hipEvent_t* ev /*4 or null*/, const double* d_b = nullptr);
// multi-GPU phases (see gsls_shard_factor / gsls_shard_solve in include/gsls.h)
hipError_t dev_shard_factor(const Symbolic& S, DeviceFactor& F, int phase, bool posdef, const double* d_val,
                            double* d_xchg, double small, double u, hipStream_t st, bool fast = false);
hipError_t dev_shard_solve(const Symbolic& S, DeviceFactor& F, int phase, bool posdef, double* d_x,
                           double* d_xchg, hipStream_t st);

}  // namespace gsls
