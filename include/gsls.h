/*
 * gsls.h -- C ABI of the MI355X-native sparse symmetric factorize+solve backend ("gsls") that
 * drops in under GALAHAD's SLS / SBLS path.
 *
 * The boundary is modelled on the one the reference already has between SLS and its SSIDS backend:
 * every entry point below names the reference interface it replaces (paths relative to the GALAHAD
 * tree).  SLS reaches these through the ISO_C_BINDING module galahad_amd/fortran/gsls_iface.f90;
 * INTEGRATION.md shows the `CASE ( 'gsls' )` arms a maintainer adds to src/sls/sls.f90.
 *
 * Conventions (same as src/ssids/ssids.f90):
 *   - indices handed in are 1-based (Fortran); `ptr` is 64-bit, `row` is 32-bit;
 *   - the matrix is the LOWER triangle by columns, every diagonal entry present, no duplicates
 *     (SLS pre-sums duplicates and inserts diagonals, src/sls/sls.f90:8409-8578);
 *   - x is column-major with leading dimension ldx >= n;
 *   - D is returned INVERTED: d[2*i] diagonal, d[2*i+1] off-diagonal of a 2x2 (fkeep.F90:357-359);
 *     piv_order[i] < 0 marks a member of a 2x2 pivot (fkeep.F90:353-356);
 *   - every function returns the SSIDS flag space (src/ssids/datatypes.f90:25-59): 0 success,
 *     < 0 error, > 0 warning; the same value is stored in inform->flag.  Nothing throws or aborts
 *     across this ABI.
 *   - caller owns ptr/row/val/x/order; the handle owns symbolic data, factors and device memory.
 *
 * There is NO CPU fallback behind this ABI: without a usable HIP device gsls_analyse keeps working
 * (it is host integer work) but gsls_factor / gsls_solve return GSLS_ERROR_HIP (-51).
 */
#ifndef GSLS_H
#define GSLS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- flags: identical values to src/ssids/datatypes.f90:25-59 ---------------------------------- */
enum {
  GSLS_SUCCESS = 0,
  GSLS_ERROR_CALL_SEQUENCE = -1,
  GSLS_ERROR_A_N_OOR = -2,
  GSLS_ERROR_A_PTR = -3,
  GSLS_ERROR_A_ALL_OOR = -4,
  GSLS_ERROR_SINGULAR = -5,
  GSLS_ERROR_NOT_POS_DEF = -6,
  GSLS_ERROR_PTR_ROW = -7,
  GSLS_ERROR_ORDER = -8,
  GSLS_ERROR_VAL = -9,
  GSLS_ERROR_X_SIZE = -10,
  GSLS_ERROR_JOB_OOR = -11,
  GSLS_ERROR_NOT_LLT = -13,
  GSLS_ERROR_NOT_LDLT = -14,
  GSLS_ERROR_NO_SAVED_SCALING = -15, /* scaling = 3 without the matching-based ordering (ssids.f90:991-994)     */
  GSLS_ERROR_ALLOCATION = -50,
  GSLS_ERROR_HIP = -51,           /* SSIDS_ERROR_CUDA_UNKNOWN */
  GSLS_ERROR_UNIMPLEMENTED = -98,
  GSLS_ERROR_UNKNOWN = -99,
  GSLS_WARNING_ANAL_SINGULAR = 6,
  GSLS_WARNING_FACT_SINGULAR = 7
};

/* solve jobs: src/ssids/datatypes.f90:61-66 */
enum {
  GSLS_SOLVE_JOB_ALL = 0,      /* P L D (P L)^T x = b */
  GSLS_SOLVE_JOB_FWD = 1,      /* P L x = b           */
  GSLS_SOLVE_JOB_DIAG = 2,     /* D x = b  (indefinite only) */
  GSLS_SOLVE_JOB_BWD = 3,      /* (P L)^T x = b       */
  GSLS_SOLVE_JOB_DIAG_BWD = 4  /* D (P L)^T x = b     */
};

/* ordering choices for gsls_options.ordering */
enum {
  GSLS_ORDER_USER = 0,     /* `order` supplied by the caller (ssids ordering=0)            */
  GSLS_ORDER_ND = 1,       /* built-in nested dissection (stands where ssids calls METIS)   */
  GSLS_ORDER_AMD = 2,      /* approximate minimum degree (SLS control%ordering = 1, MC68)    */
  GSLS_ORDER_NATURAL = 3   /* identity                                                      */
};

/* Mirrors the fields of type(ssids_options) that SLS sets (src/sls/sls.f90:1385-1439,
 * src/ssids/datatypes.f90:187-283).  Plain ints/doubles only, so the Fortran bind(C) type is 1:1. */
typedef struct gsls_options {
  int32_t print_level;     /* <0 silent (default -1)                                            */
  int32_t ordering;        /* GSLS_ORDER_*                                                      */
  int32_t nemin;           /* supernode amalgamation, default 32 (core_analyse.f90:806-822)      */
  int32_t scaling;         /* 0 none / user supplied `scale`; 1 Hungarian (MC64), 2 auction, 4 norm     */
                           /* equilibration, computed from the values (ssids.f90:921-1030); 3: the one   */
                           /* gsls_analyse_matching saved (else -15)                                     */
  int32_t action;          /* indefinite: continue on singularity with warning 7 (default 1)     */
  int32_t device;          /* HIP device ordinal, -1 = current device                            */
  int32_t reserved2;       /* (was use_graph: never implemented; launch gaps measure ~0, a graph would buy nothing) */
  int32_t reserved0;
  double u;                /* relative pivot threshold, default 0.01                             */
  double small;            /* absolute pivot tolerance, default 1e-20                            */
  double multiplier;       /* factor-memory head-room for delayed pivots, default 1.1            */
  double reserved1;
} gsls_options;

/* Mirrors type(ssids_inform) (src/ssids/inform.f90:17-44) + phase timings that SLS copies into
 * inform%time%*_external (src/sls/sls.f90:363-429). */
typedef struct gsls_inform {
  int32_t flag;
  int32_t matrix_dup;
  int32_t matrix_missing_diag;
  int32_t matrix_outrange;
  int32_t matrix_rank;
  int32_t maxdepth;
  int32_t maxfront;
  int32_t num_delay;
  int64_t num_factor;
  int64_t num_flops;
  int32_t num_neg;
  int32_t num_sup;
  int32_t num_two;
  int32_t stat;
  int32_t hip_error;       /* hipError_t of the failing runtime call, 0 otherwise                */
  int32_t not_first_pass;
  int32_t nlevels;         /* depth of the level-set schedule                                    */
  int32_t reserved0;
  int64_t factor_bytes;    /* device bytes held by L, D                                          */
  int64_t solve_bytes;     /* algorithmic bytes of one fwd+diag+bwd solve, one rhs               */
  double time_analyse;     /* seconds, wall                                                      */
  double time_factor;
  double time_solve;
  double reserved1;
} gsls_inform;

/* ---- lifecycle ----------------------------------------------------------------------------------- */

/* default options (type(ssids_options) initialisers, datatypes.f90:187-283) */
void gsls_default_options(gsls_options* options);

/* allocate an empty handle; replaces the akeep/fkeep pair SLS embeds (sls.f90:793-796) */
int gsls_create(void** handle);

/* replaces ssids_free(akeep, fkeep, cuda_error)  src/ssids/ssids.f90:1388-1419 */
int gsls_destroy(void** handle);

/* ---- analyse ------------------------------------------------------------------------------------- */

/* replaces ssids_analyse(check=.false., n, ptr, row, akeep, options, inform, order)
 *          src/ssids/ssids.f90:148-392  (called from SLS_analyse, src/sls/sls.f90:3115-3150).
 * order[n]: in: position of variable i in the pivot sequence (1-based) when ordering==GSLS_ORDER_USER;
 *           out: the pivot order actually used (ssids.f90:381). */
int gsls_analyse(void* handle, int32_t n, const int64_t* ptr, const int32_t* row, int32_t* order,
                 const gsls_options* options, gsls_inform* inform);

/* gsls_analyse with the VALUES: matching-based ordering and scaling (ssids_analyse with val and options%ordering = 2,
 * src/ssids/ssids.f90:305-320 -> src/spral/match_order.f90:51-208).  val[ptr[n]-1]: the entries of the lower triangle
 * in the order of row[].  A maximum-product matching pairs variables into 2x2 pivot candidates, the graph compressed by
 * those pairs is ordered with options->ordering (GSLS_ORDER_ND / _AMD / _NATURAL; _USER is read as _ND) and expanded
 * with the partners next to each other; order[n] (output, 1-based positions) is that order, and the matching's scaling
 * is kept in the handle for factorizations with options->scaling = 3 (ssids.f90:991-994; without this call: -15). */
int gsls_analyse_matching(void* handle, int32_t n, const int64_t* ptr, const int32_t* row, const double* val,
                          int32_t* order, const gsls_options* options, gsls_inform* inform);

/* ---- factorize ----------------------------------------------------------------------------------- */

/* replaces ssids_factor(posdef, val, akeep, fkeep, options, inform [, scale], ptr, row)
 *          src/ssids/ssids.f90:770-1108 (called from SLS_factorize, src/sls/sls.f90:4273-4297).
 * val[ptr[n]-1]: lower-triangle values in the order of `row` given to gsls_analyse (host memory).
 * scale: NULL, or n user scaling factors applied as S A S. */
int gsls_factor(void* handle, int32_t posdef, const double* val, const double* scale,
                const gsls_options* options, gsls_inform* inform);

/* same, with `val` (and `scale`) already resident in HBM (device pointers); asynchronous on the
 * handle's stream apart from the final status read-back. */
int gsls_factor_dev(void* handle, int32_t posdef, const double* d_val, const double* d_scale,
                    const gsls_options* options, gsls_inform* inform);

/* ---- the caller's own matrix on the device ------------------------------------
 * SLS keeps the user's matrix (COORDINATE storage: row, col, val of its lower triangle) and a map MAPS from its
 * entries to the sorted lower-CSC values the backend factorizes (SLS_coord_to_sorted_csr, src/sls/sls.f90:8409-8578:
 * map[l] = k > 0: val[l] is placed at position k, k < 0: added to position -k, 0: entry out of range, ignored).
 * With the map resident in HBM, SLS_factorize's host loop over the entries (sls.f90:4113-4150) becomes one
 * transfer of val[ne] and one kernel, and the residual b - A x of SLS's iterative refinement
 * (sls.f90:4826-4934) a sparse matrix-vector product on the device.
 * gsls_set_coo: after gsls_analyse; row / col may be NULL (then gsls_residual is unavailable).  1-based indices. */
int gsls_set_coo(void* handle, int64_t ne, const int32_t* row, const int32_t* col, const int32_t* map);

/* replaces the value scatter + ssids_factor pair of SLS_factorize: val[ne] in the order of the map (host memory;
 * _dev: HBM).  Duplicates are summed in entry order, as the reference does. */
int gsls_factor_coo(void* handle, int32_t posdef, const double* val, const double* scale,
                    const gsls_options* options, gsls_inform* inform);
int gsls_factor_coo_dev(void* handle, int32_t posdef, const double* d_val, const double* d_scale,
                        const gsls_options* options, gsls_inform* inform);

/* The values of the NEXT gsls_factor_coo come from up to four host arrays laid end to end (part 0 first), each times
 * `mult`, instead of from its `val` argument (which may then be NULL): what SBLS_form_n_factorize_explicit assembles on
 * the host as K%val = [ A%val | H%val | -C%val ] (sbls.f90:3319-3322, 3349, 3404, 3967) is assembled in HBM, the three
 * arrays going over the link straight from where the caller keeps them.  The lengths must add up to the `ne` of
 * gsls_set_coo (else gsls_factor_coo returns GSLS_ERROR_VAL); the arrays must stay valid until that call returns; a
 * registration serves one factorization.  part < 0 forgets a registration. */
int gsls_set_value_part(void* handle, int32_t part, const double* val, int64_t len, double mult);

/* r = b - A x for nrhs vectors (host memory, column-major), A = the matrix of the last gsls_factor_coo
 * (the residual step of SLS_solve_ir, sls.f90:4826-4934).  Row by row in a fixed order: reproducible. */
int gsls_residual(void* handle, int32_t nrhs, const double* x, int32_t ldx, const double* b, int32_t ldb,
                  double* r, int32_t ldr, gsls_inform* inform);

/* The whole of SLS_solve_ir (src/sls/sls.f90:4770-4949) on the device, for one right-hand side and the matrix of the
 * last gsls_factor_coo: x holds b on entry and the refined solution on exit (host memory), *iterations the value
 * SLS reports as inform%iterative_refinements.  Same recurrence, same stopping test
 * (max|r| < max(residual_absolute, residual_relative * max|b|)), residuals formed as gsls_residual does. */
int gsls_solve_ir(void* handle, double* x, int32_t max_refinements, double residual_absolute,
                  double residual_relative, int32_t* iterations, const gsls_options* options, gsls_inform* inform);

/* ---- solve --------------------------------------------------------------------------------------- */

/* replaces ssids_solve(x, ...) / ssids_solve(nrhs, x, ldx, ..., job)
 *          src/ssids/ssids.f90:1114-1249 (called from SLS_solve_one_rhs / SLS_solve_multiple_rhs,
 *          src/sls/sls.f90:5392-5397, 5693-5700; SLS_part_solve, sls.f90:6886-6920).
 * x is X(ldx, nrhs) column-major, ldx >= n; rows n..ldx-1 are not touched.  Several columns go through the kernels
 * together (Cholesky: blocks of 8/4/2 columns in one pass over L; otherwise up to 8 columns per launch) and every
 * column carries exactly the bits a call with nrhs = 1 gives it. */
int gsls_solve(void* handle, int32_t job, int32_t nrhs, double* x, int32_t ldx,
               const gsls_options* options, gsls_inform* inform);

/* same with x resident in HBM */
int gsls_solve_dev(void* handle, int32_t job, int32_t nrhs, double* d_x, int32_t ldx,
                   const gsls_options* options, gsls_inform* inform);

/* gsls_solve_dev with the right-hand sides in ANOTHER device array of the same shape (d_b, left untouched); d_x only
 * receives the solution.  A whole solve of one column on the LDL^T wave tier reads d_b directly (no copy); every other
 * case starts with a device-to-device copy and is gsls_solve_dev on d_x.  An interior-point loop that keeps its
 * right-hand side for the residual (SLS_solve_ir, SBLS) needs no copy of it.
 * ENQUEUE ONLY: the call returns when the work is on the handle's stream (gsls_get_stream), not when it has run --
 * the next call on this handle is ordered behind it (so a factorization can be enqueued while the solve runs, no host
 * round trip in between); to read d_x from another stream or from the host, synchronise with the handle's stream
 * first.  Errors of the enqueued kernels surface in the next call that synchronises. */
int gsls_solve_dev_rhs(void* handle, int32_t job, int32_t nrhs, const double* d_b, double* d_x, int32_t ldx,
                       const gsls_options* options, gsls_inform* inform);

/* ---- enquire / alter ----------------------------------------------------------------------------- */

/* replaces ssids_enquire_posdef(akeep, fkeep, options, inform, d)  src/ssids/ssids.f90:1255-1293 */
int gsls_enquire_posdef(void* handle, double* d, gsls_inform* inform);

/* replaces ssids_enquire_indef(akeep, fkeep, options, inform, piv_order, d)  ssids.f90:1299-1341;
 * either output may be NULL. d is (2,n) column-major. */
int gsls_enquire_indef(void* handle, int32_t* piv_order, double* d, gsls_inform* inform);

/* replaces ssids_alter(d, akeep, fkeep, options, inform)  src/ssids/ssids.f90:1347-1384 */
int gsls_alter(void* handle, const double* d, gsls_inform* inform);

/* ---- multi-GPU: elimination-tree sharding, one process (and one handle) per GPU -------------------
 * replaces the reference's subtree partition over NUMA regions / GPUs, find_subtree_partition
 * src/ssids/anal.f90:284-459 and the region assignment :569-590.  Every rank
 * analyses the same matrix, then calls gsls_shard(nranks, rank): independent subtrees are dealt to the
 * ranks, the remaining top of the tree belongs to rank 0.  The exchange steps between the phases are
 * the caller's here (galahad_amd/shard.py with torch.distributed; gsls_comm_* below does them inside the library):
 *   factor: phase 1 | REDUCE onto rank 0 d_xchg[0:xchg_factor_elems) (blocks + counters) | phase 2 |
 *           BROADCAST from rank 0 entries [E-24, E-8), E = xchg_factor_elems (status: 8 sums over the subtrees, then the
 *           top part's 8:
 *           non-positive pivots, failed columns, negative pivots, 2x2 pivots, zero pivots)
 *   solve : phase 1 | REDUCE onto rank 0 d_xchg[0:V) | phase 2 | BROADCAST from rank 0 d_xchg[0:V) | phase 3
 *           (V = total length of the cut roots' contribution vectors; d_x then holds the solution of the variables
 *           this rank eliminated).  Optional, O(n): phase 4 | SUM d_xchg[0:n) | phase 5 -> the whole solution in d_x
 *           on every rank.  d_xchg has xchg_solve_elems >= max(V, n) doubles.
 * Every summed element is non-zero on exactly one rank, so the result is exact and independent of
 * the reduction order.  inform of the factor phases carries THIS rank's num_neg / num_two / rank
 * deficiency; add them over ranks.  Delayed pivots: a factor phase that leaves inform.num_delay > 0 on
 * ANY rank has not completed; gather every rank's gsls_shard_failed list, hand the union to
 * gsls_shard_repair on every rank (same elimination-order repair as gsls_factor, then re-analyse and
 * re-shard; the exchange sizes may change) and start again at phase 1. */
int gsls_shard(void* handle, int32_t nranks, int32_t rank, int64_t* xchg_factor_elems,
               int64_t* xchg_solve_elems);
int gsls_shard_factor_dev(void* handle, int32_t phase, int32_t posdef, const double* d_val, double* d_xchg,
                          const gsls_options* options, gsls_inform* inform);
int gsls_shard_solve_dev(void* handle, int32_t phase, double* d_x, double* d_xchg, gsls_inform* inform);
#define GSLS_FAILCAP 16384
int gsls_shard_failed(void* handle, int32_t* nfailed, int32_t* failed /* GSLS_FAILCAP */);
int gsls_shard_repair(void* handle, int32_t nfailed, const int32_t* failed, int64_t* xchg_factor_elems,
                      int64_t* xchg_solve_elems);
/* the partition: owner[nnodes] (rank, -1 = top part), number of cut roots and their 1-based indices */
int gsls_shard_get(void* handle, int32_t* owner, int32_t* ncut, int32_t* cutroots);
/* device memory of this handle in doubles: the factors and the contribution-block arena.  One device: the whole tree;
 * after gsls_shard with nranks > 1: this rank's fronts (rank 0: and the top part) and blocks only */
int gsls_get_layout_sizes(void* handle, int64_t* factor_elems, int64_t* arena_elems);

/* ---- multi-GPU with the exchange INSIDE the library (RCCL on the handle's stream) -----------------------------------
 * replaces what ssids_factor / ssids_solve do for several devices in one call (src/ssids/fkeep.F90:99-174, 229-318;
 * contribution hand-over contrib.f90:20-33).  One process per GPU; every rank makes the same calls:
 *   rank 0: gsls_comm_unique_id(id) -> the host hands the 128 bytes to the other ranks (MPI, a file, torch.distributed)
 *   all:    gsls_analyse (same matrix) [-> gsls_refine_order with the same values] -> gsls_comm_init(nranks, rank, id)
 *   all:    gsls_comm_factor_dev -> gsls_comm_solve_dev ...            (any number of times)
 * Exchanged per factorization: the cut roots' contribution blocks + 8 counters (ncclReduce onto rank 0) and 16 status
 * words (ncclBroadcast); per solve: the cut roots' contribution vectors (ncclReduce) and their z-vectors
 * (ncclBroadcast).  No O(n) collective on the data path; gsls_comm_collect_dev is the optional all-reduce for callers
 * that want the whole solution on every rank.  Failed pivots: gathered (ncclAllGather) and repaired identically on
 * every rank inside gsls_comm_factor_dev. */
int gsls_comm_unique_id(char* id128);
int gsls_comm_init(void* handle, int32_t nranks, int32_t rank, const char* id128, const gsls_options* options);
int gsls_comm_factor_dev(void* handle, int32_t posdef, const double* d_val, const gsls_options* options,
                         gsls_inform* inform);
int gsls_comm_solve_dev(void* handle, double* d_x, gsls_inform* inform);
int gsls_comm_collect_dev(void* handle, double* d_x, gsls_inform* inform);
int gsls_comm_destroy(void* handle);
/* the same with HOST arrays (what GALAHAD_GSLS_double's GSLS_comm_factor / GSLS_comm_solve bind): val = the sorted
 * lower-by-columns values of gsls_factor; x = b on entry, the WHOLE solution on every rank on exit */
int gsls_comm_factor(void* handle, int32_t posdef, const double* val, const gsls_options* options, gsls_inform* inform);
int gsls_comm_solve(void* handle, double* x, gsls_inform* inform);
/* several GPUs under an UNCHANGED caller (SLS_factorize / SLS_solve, SBLS, CQP ...): start one copy of the host program
 * per GPU with GSLS_COMM_RANKS = N, GSLS_COMM_RANK = 0..N-1, GSLS_COMM_ID_FILE = <path>; gsls_analyse then joins the
 * communicator (this call; rank 0 publishes the id through the file) and gsls_factor* / gsls_solve on the handle become
 * the collective calls above.  Returns 0 if the variables are not set, 1 if joined, < 0 on error. */
int gsls_comm_init_env(void* handle, const gsls_options* options);

/* ---- introspection used by the parity tests and bench (not part of the SSIDS surface) ------------ */

/* symbolic factorization as the reference's akeep holds it (src/ssids/akeep.f90:25-76):
 * sizes first (any output pointer may be NULL), then the arrays, 1-based like the reference. */
int gsls_get_symbolic_sizes(void* handle, int32_t* nnodes, int64_t* rlist_len, int64_t* nlist_len);
int gsls_get_symbolic(void* handle, int32_t* sptr, int32_t* sparent, int64_t* rptr, int32_t* rlist,
                      int64_t* nptr, int64_t* nlist);

/* host utility (no handle, no device): the scaling gsls_factor* computes for options.scaling = kind (1 Hungarian,
 * 2 auction, 4 norm equilibration) -- hungarian_scale_sym / auction_scale_sym / equilib_scale_sym of
 * src/spral/scaling.f90:109-170, 245-309, 381-416.  ptr / row 1-based lower triangle by columns.
 * Returns 0, 1 (structurally singular, scaled anyway because action != 0) or GSLS_ERROR_SINGULAR. */
int gsls_scale_sym(int32_t kind, int32_t n, const int64_t* ptr, const int32_t* row, const double* val, int32_t action,
                   double* scaling);

/* multi-GPU, phase drivers only: status words [5] / [6] of the factor exchange are the fronts the wave-per-front
 * kernels gave up on and the blocks that needed pivoting, summed over the ranks.  After a factorization with [6] = 0
 * call gsls_shard_fast(h, 1) on every rank: the next one takes the wave-per-front kernels; if that one reports
 * [5] > 0, call gsls_shard_fast(h, 0) and repeat it.  (gsls_comm_factor_dev does this itself.) */
int gsls_shard_fast(void* handle, int32_t on);

/* the scaling factors the last factorization computed itself (options.scaling = 1, 2, 4), in the caller's
 * variable order: the optional `scale` output of ssids_factor (src/ssids/ssids.f90:955-958, 983-986, 1021-1025) */
int gsls_get_scaling(void* handle, double* scaling);

/* the elimination order the handle currently holds (order[var] = 1-based pivot position): what analyse was
 * given or computed, as repaired by factorizations that met delayed pivots; a valid PERM for SLS_analyse */
int gsls_get_order(void* handle, int32_t* order);

/* Hand over the values before the first factorization so that an order chosen by analyse can be refined (zero-diagonal
 * variables -- constraint rows of a saddle-point matrix -- after their neighbours: value-independent pivots).
 * gsls_factor[_dev] does this itself on the first indefinite factorization; multi-GPU callers call it on every rank
 * before gsls_shard (the partition depends on the order).  No reference counterpart: SSIDS delays pivots instead. */
int gsls_refine_order_dev(void* handle, const double* d_val, gsls_inform* inform);
/* the same with `val` in host memory; needs no device */
int gsls_refine_order(void* handle, const double* val, gsls_inform* inform);

/* how the last LDL^T factorization went ("optimistic pass"): blocks / tiny fronts that passed the
 * optimistic kernels, blocks redone by the complete-pivoting kernel, tiny fronts kept off the wave-per-front kernel */
int gsls_get_factor_stats(void* handle, int32_t* fast_blocks, int32_t* pivoted_blocks, int32_t* tiny_blacklist);

/* stream the handle launches on (hipStream_t as void*), and kernel timing of the last solve measured with HIP events
 * on that stream (seconds); used by bench.py's roofline block.  The whole sweep comes back in *fwd (*diag = *bwd = 0):
 * events BETWEEN the phases cost the sweep ~10 us and are recorded only when GSLS_SOLVE_PHASES is set in the
 * environment -- then the three values are the forward, diagonal and backward parts. */
void* gsls_get_stream(void* handle);
int gsls_last_solve_kernel_seconds(void* handle, double* fwd, double* diag, double* bwd);

/* library/device probe: returns number of visible HIP devices (0 when none / no driver) */
int gsls_device_count(void);
const char* gsls_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSLS_H */
