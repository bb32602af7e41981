// HIP kernels and launch sequences of the gsls backend -- gfx950 (MI355X, CDNA4) only.
//
// What runs here replaces, behaviourally, the reference's numeric phase:
//   * A -> front scatter           (ssids/cpu/kernels/assemble.hxx:49-79 add_a_block;
//                                   CUDA: ssids/assemble.cu:39-96 cu_load_nodes[_sc])
//   * child -> parent extend-add   (assemble.hxx:91-137, 244-345, 347-437; CUDA assemble.cu:170-230)
//   * dense partial factorization  (ssids/cpu/kernels/cholesky.cxx:32-188; ldlt_app.cxx; CUDA
//                                   dense_factor.cu cu_block_chol / cu_block_ldlt + syrk.cu)
//   * contribution block           (factor.hxx:84-99 calcLD + gemm; CUDA syrk.cu:178-385)
//   * forward / diagonal / backward solves (NumericSubtree.hxx:280-400; CUDA solve.cu, dtrsv.h)
// but is organised MI355X-first: a level-set schedule over the assembly tree fixed at analyse time,
// every level = a handful of batched launches over device-resident task lists, fronts resident in
// HBM for the lifetime of the handle, 64-wide wavefronts, LDS-resident panels, and
// v_mfma_f64_16x16x4_f64 for every L*D*L^T update.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gsls_device.hpp"

namespace gsls {

typedef double double4_t __attribute__((ext_vector_type(4)));

#ifdef GSLS_STAMPS   // diagnostic build only: in-kernel phase stamps (s_memtime), never in the product
__device__ unsigned long long g_stamps[64];
#define STAMP(i) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define PH(k) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); phacc[k] += t__ - pht; pht = t__; } while (0)
#define STAMPN(i) do { __syncthreads(); if (threadIdx.x == 0 && nd.m > 300 && nd.n > 100 && nd.m > nd.n) g_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMPW(i) do { if (lane == 0 && !t.has_contrib) g_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMPW(i) do {} while (0)
#define STAMP(i) do {} while (0)
#define PH(k) do {} while (0)
#define STAMPN(i) do {} while (0)
#endif

#ifndef GSLS_FW_LDSB
#define GSLS_FW_LDSB 0
#endif

#define HIPCHK(call)                    \
  do {                                  \
    hipError_t e__ = (call);            \
    if (e__ != hipSuccess) return e__;  \
  } while (0)

// =================================================================================================
// A -> L scatter
// =================================================================================================
__global__ void k_scatter_a(int64_t cnt, const int64_t* __restrict__ asrc,
                            const int64_t* __restrict__ adst, const double* __restrict__ val,
                            double* __restrict__ L, const double* __restrict__ scale,
                            const int32_t* __restrict__ arow, const int32_t* __restrict__ acol,
                            const int32_t* __restrict__ invp) {
  int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (; i < cnt; i += stride) {
    const int64_t d = adst[i];
    if (d < 0) continue;            // (a sharded run: the entry belongs to a front another rank owns)
    double v = val[asrc[i]];
    if (scale) v *= scale[invp[arow[i]]] * scale[invp[acol[i]]];
    L[d] = v;
  }
}

// zero the contribution blocks a level is about to use (the arena is reused level by level)
struct ZeroTask {
  int64_t off, len;
};
__global__ void __launch_bounds__(256)
k_zero_tasks(const ZeroTask* __restrict__ tasks, double* __restrict__ C) {
  const ZeroTask t = tasks[blockIdx.x];
  double* p = C + t.off;
  for (int64_t i = threadIdx.x; i < t.len; i += 256) p[i] = 0.0;
}

// =================================================================================================
// extend-add: every parent pulls its children's contribution blocks, one child after the other
// (deterministic summation order, no atomics).
// =================================================================================================
// Pull flavour, one launch per level: a workgroup owns PCOLS columns of a parent front and walks the
// children IN ORDER, adding from each the (host-computed) range of its columns that land there --
// deterministic summation order without atomics and without one launch per child rank.
constexpr int PCOLS = 8;
struct PullSeg {       // columns [j0, j1) of one child's contribution block, with everything the adds need
  int32_t j0, j1, cm, pn, pld, ppcm;
  int64_t cmoff, ccoff, ploff, pcoff;
};
struct PullTask {
  int32_t seg_begin, seg_cnt;
};
__global__ void __launch_bounds__(256)
k_assemble_pull(const PullTask* __restrict__ tasks, const PullSeg* __restrict__ segs,
                const int32_t* __restrict__ cmap, double* __restrict__ L, double* __restrict__ C) {
  const PullTask t = tasks[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int U = 4;   // 256 rows per pass
  for (int si = 0; si < t.seg_cnt; ++si) {
    const PullSeg sg = segs[t.seg_begin + si];
    const int cm = sg.cm;
    const int32_t* map = cmap + sg.cmoff;
    const double* src = C + sg.ccoff;
    const int j0 = sg.j0 + 2 * wave;
    if (j0 < sg.j1) {
      const int nc = min(2, sg.j1 - j0);
      for (int base = (j0 / (64 * U)) * (64 * U); base < cm; base += 64 * U) {
        int mi[U];
        double v[2][U];
        int pc[2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + lane + 64 * u;
          mi[u] = (i < cm) ? map[i] : 0;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int j = j0 + c;
          pc[c] = (c < nc) ? map[j] : 0;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int i = base + lane + 64 * u;
            v[c][u] = (c < nc && i >= j && i < cm) ? src[int64_t(j) * cm + i] : 0.0;
          }
        }
        double* dst[2];
        double d[2][U];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          dst[c] = (pc[c] < sg.pn) ? (L + sg.ploff + int64_t(pc[c]) * sg.pld)
                                   : (C + sg.pcoff + int64_t(pc[c] - sg.pn) * sg.ppcm - sg.pn);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int i = base + lane + 64 * u;
            d[c][u] = (c < nc && i >= j0 + c && i < cm) ? dst[c][mi[u]] : 0.0;
          }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int i = base + lane + 64 * u;
            if (c < nc && i >= j0 + c && i < cm) dst[c][mi[u]] = d[c][u] + v[c][u];
          }
      }
    }
    __syncthreads();   // the next child may add to the same entries
  }
}

// =================================================================================================
// MFMA core:  acc(ARxBC tile) += Arows(:,k0:k1) * diag(d) * Brows(:,k0:k1)^T
// A and B are row ranges of the same column-major L block.  v_mfma_f64_16x16x4_f64: lane l feeds
// A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; result reg r of lane l is C[row=(l>>4)+4r][col=l&15].
// =================================================================================================
constexpr int KC = 16;

__device__ __forceinline__ double readlane_f64(double v, int k) {   // k wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}


template <int AROWS, int KCH = KC>
struct Stage {
  static constexpr int AST = AROWS + 16;  // == 16 mod 32 doubles: the two 16-lane halves of a
  static constexpr int BST = 64 + 16;     // ds_read_b64 lane group land on disjoint banks
  double As[KCH][AST];
  double Bs[KCH][BST];
};

// MT x NT 16x16 tiles per wave at (rbase, cbase) of the staged panels
// TRANS: the tile is produced transposed (acc[i][j][r] = C[col = 16j + (l>>4) + 4r][row = 16i + (l&15)]), i.e.
// lanes run along the ROWS of the front: coalesced / conflict-free epilogues for column-major storage
template <int AROWS, int MT, int NT, int KCH = KC, bool TRANS = false>
__device__ __forceinline__ void mfma_panel(const Stage<AROWS, KCH>& sg, int rbase, int cbase, int lane,
                                           double4_t (&acc)[MT][NT]) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int k4 = 0; k4 < KCH; k4 += 4) {
    double a[MT], b[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = sg.As[k4 + lk][rbase + 16 * i + lr];
#pragma unroll
    for (int j = 0; j < NT; ++j) b[j] = sg.Bs[k4 + lk][cbase + 16 * j + lr];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[i][j], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

// stage rows [arow0, arow0+AROWS) (valid < amax) and rows [brow0, brow0+64) (valid < bmax) of
// columns [k0, k0+KC) (valid < kmax) of the front at Lb.  With dinv != nullptr the B panel is
// (L*D): D is held inverted, 1x1 as [d,0], 2x2 as [d11,d21,inf,d22] (ldlt_app.cxx:324-329), and
// L*D is formed as in calc_ld.hxx:43-118.
// Operand staging for the MFMA updates, split in two halves so that the loads of K-chunk k+1 are in
// flight while chunk k is multiplied (register double buffering; the LDS image is single):
//   stage_load : rows [arow0, arow0+AROWS) (valid < amax) and rows [brow0, brow0+64) (valid < bmax)
//                of columns [k0, k0+KC) (valid < kmax) of the front at Lb -> registers.  With
//                dinv != nullptr the B panel is (L*D): D is held inverted, 1x1 as [d,0], 2x2 as
//                [d11,d21,inf,d22] (ldlt_app.cxx:324-329); L*D as in calc_ld.hxx:43-118.
//   stage_store: registers -> LDS.
template <int AROWS, int KCH = KC>
struct StageRegs {
  double va[AROWS * KCH / 256];
  double vb[64 * KCH / 256];
};

template <int AROWS, int KCH = KC>
__device__ __forceinline__ void stage_load(StageRegs<AROWS, KCH>& rg, const double* __restrict__ Lb, int ld,
                                           int arow0, int amax, int brow0, int bmax, int k0, int kmax,
                                           const double* __restrict__ dinv, int tid) {
  constexpr int NA = AROWS * KCH / 256, NBL = 64 * KCH / 256;
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    const int e = tid + 256 * t;
    const int r = e % AROWS, kk = e / AROWS;
    const int gr = arow0 + r, gk = k0 + kk;
    rg.va[t] = (gr < amax && gk < kmax) ? Lb[int64_t(gk) * ld + gr] : 0.0;
  }
#pragma unroll
  for (int t = 0; t < NBL; ++t) {
    const int e = tid + 256 * t;
    const int r = e & 63, kk = e >> 6;
    const int gr = brow0 + r, gk = k0 + kk;
    const bool ok = (gr < bmax && gk < kmax);
    double v = ok ? Lb[int64_t(gk) * ld + gr] : 0.0;
    if (dinv && ok) {
      const double d0 = dinv[2 * gk], d1 = dinv[2 * gk + 1];
      if (isinf(d0)) {                                   // second column of a 2x2 pivot
        const double d11 = dinv[2 * gk - 2], d21 = dinv[2 * gk - 1], d22 = d1;
        const double a1 = Lb[int64_t(gk - 1) * ld + gr];
        v = (-d21 * a1 + d11 * v) / (d11 * d22 - d21 * d21);
      } else if (isinf(dinv[2 * gk + 2])) {              // first column (D has a spare pair at the end)
        const double d22 = dinv[2 * gk + 3];
        const double a2 = Lb[int64_t(gk + 1) * ld + gr];
        v = (d22 * v - d1 * a2) / (d0 * d22 - d1 * d1);
      } else {
        v = (d0 != 0.0) ? v / d0 : 0.0;                  // zero pivots just give zeros
      }
    }
    rg.vb[t] = v;
  }
}

template <int AROWS, int KCH = KC>
__device__ __forceinline__ void stage_store(Stage<AROWS, KCH>& sg, const StageRegs<AROWS, KCH>& rg, int tid) {
  constexpr int NA = AROWS * KCH / 256, NBL = 64 * KCH / 256;
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    const int e = tid + 256 * t;
    sg.As[e / AROWS][e % AROWS] = rg.va[t];
  }
#pragma unroll
  for (int t = 0; t < NBL; ++t) {
    const int e = tid + 256 * t;
    sg.Bs[e >> 6][e & 63] = rg.vb[t];
  }
}

// =================================================================================================
// diag kernel: block column `step` of a front: rows [kb, kb+128) x cols [kb, kb+w)
//   1. left-looking update with the front's earlier block columns (MFMA)
//   2. right-looking factorization of the 128 x w panel in LDS (Cholesky, or LDL^T with 1x1 pivots)
//   3. store L11 (lower), the first row chunk of L21, and D^-1
// stat[0]: smallest failing pivot position (posdef), stat[1]: #zero pivots, stat[2]: #negative
// =================================================================================================
constexpr int PR = 128;  // panel rows handled by the diag kernel
constexpr int LDP = PR + 1;  // LDS panel leading dimension (odd: column-strided access is conflict free)

// ---- Cholesky flavour ----------------------------------------------------------------------------
// The 128-row panel gets 64 extra rows holding the identity: the same block operations that turn A21
// into L21 = A21 L11^-T turn them into L11^-T, which the panel kernel (and the solves) then use
// instead of a 64-step substitution.  The 64 columns are factorized in four 16-column stages:
//   a. wave 0: 16 x 16 Cholesky, one matrix row per lane in registers (static indices, 16 unrolled
//      pivots, lane broadcasts through v_readlane); lanes 16-31 run the SAME instruction stream on the
//      columns of the identity, which leaves them holding the inverse of the 16 x 16 factor;
//   b. every row tile below:  Y = R * L16^-T            (MFMA, K = 16)
//   c. trailing columns:      P -= Y_rows * Y_cols^T     (MFMA, K = 16)
// i.e. 4 x (1 serial stage + 2 MFMA stages) instead of 64 barrier-separated pivots.
constexpr int CK = 32;           // K chunk of the Cholesky kernels' left-looking updates
constexpr int PRX = PR + NB;     // panel rows + identity rows
constexpr int LDQ = PRX + 16;    // == 16 mod 32 doubles: conflict-free MFMA operand reads

// P[row tile rt, col tiles cj0 .. cj0+NC) -= Y[rt] Y[cj]^T with Y = columns jb..jb+16 of the LDS panel
template <int NC>
__device__ __forceinline__ void syrk_row(double* P, int ldq, int jb, int rt, int cj0, int w, int lr, int lq,
                                         const double* ypb = nullptr) {   // ypb: Y' = R L^-T (LDL^T), [16][ldq]
  const int row0 = 16 * rt;
  double yb[4], ya[NC][4];
  double4_t c[NC];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    yb[k] = ypb ? ypb[(4 * k + lq) * ldq + row0 + lr] : P[(jb + 4 * k + lq) * ldq + row0 + lr];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const int col0 = 16 * (cj0 + q);
#pragma unroll
    // columns >= w are padding (unit diagonal): they must not see the real rows' updates
    for (int k = 0; k < 4; ++k) {
      ya[q][k] = (col0 + lr < w) ? -P[(jb + 4 * k + lq) * ldq + col0 + lr] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) c[q][r] = P[(col0 + lq + 4 * r) * ldq + row0 + lr];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int q = 0; q < NC; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[q][k], yb[k], c[q], 0, 0, 0);
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const int col0 = 16 * (cj0 + q);
#pragma unroll
    for (int r = 0; r < 4; ++r) P[(col0 + lq + 4 * r) * ldq + row0 + lr] = c[q][r];
  }
}

// LDL = true: the same pipeline as an OPTIMISTIC pass of the pivoted LDL^T -- pivots taken in the given
// order, 1x1 only, every multiplier checked against 1/u and every pivot against `small`.  If a check
// fails anywhere in the block nothing is stored and fastok[block] = 0: k_diag_ldlt (complete pivoting,
// 2x2 pivots) then redoes that block; blocks that pass are skipped by it.  (Optimistic first, robust
// fallback: the a-posteriori idea of ldlt_app.cxx:303-321 at block granularity.)
// GEMM = false: first block column of a front (nothing to the left): no update code, fewer registers
template <bool LDL, bool GEMM>
__global__ void __launch_bounds__(256, LDL ? (GEMM ? 2 : 3) : 1)
k_diag_fast(const NodeDesc* __restrict__ nodes, const PanelTask* __restrict__ tasks,
            double* __restrict__ L, double* __restrict__ Linv, double* __restrict__ D,
            int32_t* __restrict__ stat, int32_t* __restrict__ fastok, const uint8_t* __restrict__ hint,
            double small, double u, int ldq_arg, int nrt_arg, const uint8_t* __restrict__ tppflag) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Stage<PR, CK>& sg = *reinterpret_cast<Stage<PR, CK>*>(smem_raw);
  double* P = reinterpret_cast<double*>(smem_raw);  // [NB][LDQ], overlays the staging buffers after the GEMM
  // LDL: the panel is only as tall as the launch's tallest front (ldq, nrt from the host): several small
  // fronts share a CU.  Cholesky keeps compile-time constants.
  const int ldq = LDL ? ldq_arg : LDQ;
  const int nrt = LDL ? nrt_arg : PRX / 16;       // row tiles: the identity rows (W) only for Cholesky
  double* YP = P + NB * ldq;                        // LDL: [16][ldq], Y' = R L16^-T of the current stage
  // LDL scratch lives in tiles of P ABOVE the diagonal, which the factorization never touches (LDS is what
  // limits the number of small fronts per CU): Zs[k][n] = (D16^-1 L16^-1)[n][k] in rows 0..15 of columns
  // 48..63, the inverted pivots (layout of D, 128 doubles) in rows 0..15 of columns 32..39
  double* Zs = P + 48 * ldq;
  double* dgp = P + 32 * ldq;
  auto dgs = [&](int i) -> double& { return dgp[(i >> 4) * ldq + (i & 15)]; };
  int negacc = 0, twoacc = 0;                       // LDL: inertia counts, live in the serial-stage wave
  // Xs[k][n] = (L16^-1)[n][k]: LDL keeps it in rows 0..15 of YP, which no stage ever uses (LDS is what
  // limits the number of small fronts per CU)
  double* Xs = YP;
  const int xst = LDL ? ldq : 16;
  if constexpr (!LDL) {
    __shared__ double XsC[16 * 16];
    Xs = XsC;
  }

  const PanelTask t = tasks[blockIdx.x];
  if (LDL && tppflag[t.node]) return;     // the whole front goes through k_front_tpp
  const NodeDesc nd = nodes[t.node];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int kb = t.step * NB;
  const int w = min(NB, nd.n - kb);
  const int pr = min(PR, nd.m - kb);
  double* Lb = L + nd.loff;
  const double* dinv = LDL ? (D + 2 * int64_t(nd.sptr)) : nullptr;
  const double inv_u = (LDL && u > 0.0) ? 1.0 / u : INFINITY;
  bool bad = false;
  int why = 0;   // diagnostics only (GSLS_DEBUG): which test abandoned the block

  STAMP(0);
  // the block's own entries first: their latency hides behind the left-looking update.  Tiles are
  // produced transposed (lanes along the rows of the front), so these loads are 128-byte segments
  // and the LDS panel is written without bank conflicts.
  double g[2][4][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 32 * wave + 16 * i + lr;
        const int col = 16 * j + lq + 4 * r;
        g[i][j][r] = (row < pr && col < w) ? Lb[int64_t(kb + col) * nd.ld + kb + row] : 0.0;
      }
  double4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
  if (GEMM && kb > 0) {
    StageRegs<PR, CK> rg;
    stage_load<PR, CK>(rg, Lb, nd.ld, kb, nd.m, kb, kb + w, 0, kb, dinv, tid);
    for (int k0 = 0; k0 < kb; k0 += CK) {
      __syncthreads();
      stage_store<PR, CK>(sg, rg, tid);
      __syncthreads();
      if (k0 + CK < kb) stage_load<PR, CK>(rg, Lb, nd.ld, kb, nd.m, kb, kb + w, k0 + CK, kb, dinv, tid);
      mfma_panel<PR, 2, 4, CK, true>(sg, 32 * wave, 0, lane, acc);
    }
  }
  __syncthreads();
  STAMP(1);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 32 * wave + 16 * i + lr;
        const int col = 16 * j + lq + 4 * r;
        // columns beyond the front's last one get a unit diagonal: they factorize to the identity
        if (!LDL || row < 16 * nrt)
          P[col * ldq + row] = (row < pr && col < w) ? g[i][j][r] - acc[i][j][r] : ((row == col) ? 1.0 : 0.0);
      }
  if (!LDL)
    for (int e = tid; e < NB * NB; e += 256) {
      const int k = e & 63, n = e >> 6;
      P[n * ldq + PR + k] = (k == n) ? 1.0 : 0.0;
    }
  __syncthreads();
  STAMP(2);

  // ---- a. 16 x 16 factorization + inverse of the diagonal block at jb (one wave) ---------------------
  // LDL: pivots in the given order; hint[] (learned from the handle's previous pivoted factorization)
  // says where a 2x2 pivot (j, j+1) is to be taken.  Every pivot and multiplier is tested; `bad`
  // abandons the block to k_diag_ldlt.
  auto fact16 = [&](int jb) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      v[k] = (lane < 16) ? P[(jb + k) * ldq + jb + lr] : ((lr == k) ? 1.0 : 0.0);
    int failj = 16, nneg = 0, ntwo = 0;
    unsigned hmask = 0;
    if (LDL) {
      const bool h2 = (lane < 16 && jb + lr < w) ? (hint[nd.sptr + kb + jb + lr] != 0) : false;
      hmask = unsigned(__ballot(h2));
    }
    bool second = false;                // wave-uniform: column j is the second of a 2x2 pivot
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (LDL && second) { second = false; continue; }
      const bool want2 = LDL && ((hmask >> j) & 1u);
      if (want2 && (j == 15 || jb + j + 1 >= w)) { bad = true; why |= 4; }      // the pair straddles this stage
      if (LDL && want2 && j < 15 && jb + j + 1 < w) {
        const double a11 = readlane_f64(v[j], j), a21 = readlane_f64(v[j], j + 1);
        const double a22 = readlane_f64(v[j + 1], j + 1);
        // a hinted pair was chosen by one of the pivoting kernels before: accepted unless its determinant
        // cancels (test_2x2 of ldlt_tpp.cxx:99-118, the weaker of the reference's two 2x2 tests; the
        // multipliers are tested below like every other pivot's)
        const double maxpiv = fmax(fabs(a11), fmax(fabs(a21), fabs(a22)));
        if (!(maxpiv >= small)) { bad = true; why |= 8; }
        const double detscale = 1.0 / maxpiv;
        const double detpiv0 = (a11 * detscale) * a22, detpiv1 = (a21 * detscale) * a21;
        const double detpiv = detpiv0 - detpiv1;
        if (!(fabs(detpiv) >= fmax(small, fmax(fabs(detpiv0 / 2), fabs(detpiv1 / 2))))) { bad = true; why |= 8; }
        const double d11 = (a22 * detscale) / detpiv, d22 = (a11 * detscale) / detpiv;
        const double d21 = (-a21 * detscale) / detpiv;
        const double own1 = d11 * v[j] + d21 * v[j + 1], own2 = d21 * v[j] + d22 * v[j + 1];
        if (lane < 16 && lr > j + 1 && !(fabs(own1) <= inv_u && fabs(own2) <= inv_u)) { bad = true; why |= 2; }
        const double u1 = (jb + lr < w) ? v[j] : 0.0, u2 = (jb + lr < w) ? v[j + 1] : 0.0;
#pragma unroll
        for (int k = j + 2; k < 16; ++k) {
          const double l1 = readlane_f64(u1, k), l2 = readlane_f64(u2, k);
          v[k] = fma(-own1, l1, fma(-own2, l2, v[k]));
        }
        if (lane < 16) {
          v[j] = (lr == j) ? 1.0 : ((lr == j + 1) ? 0.0 : own1);
          v[j + 1] = (lr == j + 1) ? 1.0 : own2;
        }
        const double det = a11 * a22 - a21 * a21;
        if (det < 0.0) nneg += 1;
        else if (a11 + a22 < 0.0) nneg += 2;
        ++ntwo;
        if (lane == 0) { dgs(2 * (jb + j)) = d11; dgs(2 * (jb + j) + 1) = d21; dgs(2 * (jb + j) + 2) = INFINITY; dgs(2 * (jb + j) + 3) = d22; }
        second = true;
        continue;
      }
      const double d = readlane_f64(v[j], j);
      double own;
      if (LDL) {
        if (jb + j < w) {
          if (!(fabs(d) >= small)) { bad = true; why |= 1; }
          if (d < 0.0) ++nneg;
        }
        double rd = __builtin_amdgcn_rcp(d);         // reciprocal + two Newton steps: full precision
        rd = fma(fma(-d, rd, 1.0), rd, rd);
        rd = fma(fma(-d, rd, 1.0), rd, rd);
        own = v[j] * rd;                             // L rows: l_rj (lane j: 1); identity lanes: x_j / d
        if (lane < 16 && lr > j && !(fabs(own) <= inv_u)) { bad = true; why |= 2; }   // threshold test inside the block
        if (lane == 0) { dgs(2 * (jb + j)) = rd; dgs(2 * (jb + j) + 1) = 0.0; }
      } else {
        if (!(d > 0.0)) failj = min(failj, j);
        double y = __builtin_amdgcn_rsq(d);          // 1/sqrt(d): hardware estimate + two Newton steps
        const double hd = 0.5 * d;
        y = y * fma(-hd * y, y, 1.5);
        y = y * fma(-hd * y, y, 1.5);
        v[j] *= y;                                   // L rows: l_rj (lane j: sqrt(d)); identity lanes: x_j
        own = v[j];
      }
      // columns >= w are padding (unit diagonal): rows >= w keep their l_rj but must not update them
      const double um = (jb + lr < w) ? v[j] : 0.0;  // LDL: the unscaled a_kj
#pragma unroll
      for (int k = j + 1; k < 16; ++k) {
        const double lkj = readlane_f64(um, k);
        v[k] = fma(-own, lkj, v[k]);
      }
      if (LDL && lane < 16) v[j] = own;              // the identity lanes keep x_j
    }
    if (!LDL && lane == 0 && failj < 16 && jb + failj < w) atomicMin(&stat[0], nd.sptr + kb + jb + failj);
    if (LDL) { negacc += nneg; twoacc += ntwo; }
    if (lane < 16) {
#pragma unroll
      for (int k = 0; k < 16; ++k) P[(jb + k) * ldq + jb + lr] = (k <= lr) ? v[k] : 0.0;
    } else if (lane < 32) {
#pragma unroll
      for (int k = 0; k < 16; ++k) Xs[lr * xst + k] = (k >= lr) ? v[k] : 0.0;
      if (LDL) {   // Z = D16^-1 L16^-1, column lr; the pivots come back from LDS (dgs, written by lane 0)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        double z[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) z[k] = dgs(2 * (jb + k)) * v[k];
#pragma unroll
        for (int k = 0; k + 1 < 16; ++k)
          if (jb + k + 1 < NB && isinf(dgs(2 * (jb + k) + 2))) {   // 2x2 pivot at (k, k+1): [d11, d21, inf, d22]
            const double d11 = dgs(2 * (jb + k)), d21 = dgs(2 * (jb + k) + 1), d22 = dgs(2 * (jb + k) + 3);
            z[k] = d11 * v[k] + d21 * v[k + 1];
            z[k + 1] = d21 * v[k] + d22 * v[k + 1];
          }
#pragma unroll
        for (int k = 0; k < 16; ++k) Zs[lr * ldq + k] = z[k];
      }
    }
  };
  // the wave that runs the serial 16 x 16 stages rotates with the workgroup index, so that workgroups
  // sharing a CU do not queue their serial stages on the same SIMD
  const int fw = blockIdx.x & 3;
  if (wave == fw) fact16(0);
  __syncthreads();
  const int wup = LDL ? ((w + 15) & ~15) : NB;   // LDL: stages beyond the front's last column are pure padding
  for (int jb = 0; jb < wup; jb += 16) {
    STAMP(4 + 3 * (jb >> 4));
    // ---- b. rows below: Y = R * L16^-T [* D16^-1], computed transposed so that lanes run along the rows
    // of P.  Three row tiles per wave, no branches around the MFMAs: a tile index past the end is
    // clamped to the last tile (computed again, not stored).
    {
      const int rt0 = (jb >> 4) + 1;
      double4_t c[3], cz[3];
      double rb[3][4], xa[4], za[4];
      int rtc[3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xa[k] = Xs[(4 * k + lq) * xst + lr];                 // A[i=n][k] = X[n][k]
        za[k] = LDL ? Zs[(4 * k + lq) * ldq + lr] : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        rtc[i] = min(rt0 + wave + 4 * i, nrt - 1);
        c[i] = double4_t{0.0, 0.0, 0.0, 0.0};
        cz[i] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k) rb[i][k] = P[(jb + 4 * k + lq) * ldq + 16 * rtc[i] + lr];   // B[k][j=row]
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[k], rb[i][k], c[i], 0, 0, 0);
          if (LDL) cz[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(za[k], rb[i][k], cz[i], 0, 0, 0);
        }
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (rt0 + wave + 4 * i < nrt) {           // a clamped duplicate must not store: its inputs may be gone
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (LDL) {
              const double yv = cz[i][r];
              if (!(fabs(yv) <= inv_u)) { bad = true; why |= 16; }   // a-posteriori threshold test on the rows below
              P[(jb + lq + 4 * r) * ldq + 16 * rtc[i] + lr] = yv;           // Y = R L^-T D^-1 (the factor)
              YP[(lq + 4 * r) * ldq + 16 * rtc[i] + lr] = c[i][r];          // Y' = R L^-T
            } else {
              P[(jb + lq + 4 * r) * ldq + 16 * rtc[i] + lr] = c[i][r];
            }
          }
        }
    }
    __syncthreads();
    STAMP(5 + 3 * (jb >> 4));
    if (jb + 16 < wup) {
      const double* dsc = LDL ? YP : nullptr;
      // ---- c1. the next block column only, all waves ------------------------------------------------
      const int cj1 = (jb >> 4) + 1;
      for (int rt = cj1 + wave; rt < nrt; rt += 4) syrk_row<1>(P, ldq, jb, rt, cj1, w, lr, lq, dsc);
      __syncthreads();
      // ---- wave 0 factorizes the next diagonal block while the others finish the trailing update -----
      if (wave == fw) {
        fact16(jb + 16);
      } else if (cj1 + 1 < NB / 16) {
        for (int rt = cj1 + 1 + ((wave - fw - 1) & 3); rt < nrt; rt += 3) {
          if (min(NB / 16 - 1, rt) - cj1 == 2) syrk_row<2>(P, ldq, jb, rt, cj1 + 1, w, lr, lq, dsc);
          else syrk_row<1>(P, ldq, jb, rt, cj1 + 1, w, lr, lq, dsc);
        }
      }
      __syncthreads();
    }
    STAMP(6 + 3 * (jb >> 4));
  }
  STAMP(16);
  if (LDL) {
    const bool anybad = __syncthreads_or(bad);
    for (int b2 = 0; b2 < 5; ++b2)
      if (__syncthreads_or((why >> b2) & 1) && tid == 0) atomicAdd(&stat[8 + b2], 1);
    if (tid == 0) {
      fastok[nd.iblk + t.step] = anybad ? 0 : 1;
      // statistics: blocks done optimistically (binned: no same-address atomics) / redone with pivoting (rare)
      if (anybad) atomicAdd(&stat[7], 1);
      else atomicAdd(&stat[16 + STAT_BINS + (blockIdx.x & (STAT_BINS - 1))], 1);
    }
    if (anybad) return;                              // k_diag_ldlt takes the block from the untouched input
    if (tid < w) {
      D[2 * int64_t(nd.sptr + kb + tid)] = dgs(2 * tid);
      D[2 * int64_t(nd.sptr + kb + tid) + 1] = dgs(2 * tid + 1);
    }
    if (wave == fw && lane == 0) {
      if (negacc) atomicAdd(&stat[16 + (blockIdx.x & (STAT_BINS - 1))], negacc);
      if (twoacc) atomicAdd(&stat[3], twoacc);
    }
  }
  // ---- store L (lower trapezoid) and W = L11^-T; both coalesced along rows ---------------------------
  {
    const int row = tid & (PR - 1);
    double* dst = Lb + int64_t(kb) * nd.ld + kb + row;
    for (int col = tid >> 7; col < w; col += 2)
      if (row >= col && row < pr) dst[int64_t(col) * nd.ld] = P[col * ldq + row];
  }
  if (!LDL) {
    double* W = Linv + (int64_t(nd.iblk) + t.step) * (NB * NB);
    for (int e = tid; e < NB * NB; e += 256) W[e] = P[(e >> 6) * ldq + PR + (e & 63)];
  }
  STAMP(17);
}

// =================================================================================================
// Whole-front LDL^T for TINY fronts (n <= 48 pivots, m <= 64 rows), assembly included: one WAVE per front, no
// barriers, no MFMA.  The wave
//   1. builds the front in LDS (lower triangle, packed columns): zero, the entries of A (gathered through the value
//      map -- the rectangle in HBM is never read), then the children's contribution blocks column by column, eight
//      columns' loads in flight at a time, children in the order the extend-add kernel uses (assemble.hxx:347-437 is
//      the reference's version of this step);
//   2. takes row `lane` of the front into registers -- pivot rows and contribution rows alike, all columns -- and runs
//      the pivots as in the serial stage of k_diag_fast (given order, every pivot and multiplier tested), every row
//      below taking its update in the same instruction;
//   3. writes L (or, for the wave tier of the solves, the two packed images), D, and the contribution block
//      C = (assembled part) - L21 D L21^T straight from LDS + registers: no read-modify-write in HBM.
// A KKT tree is tens of thousands of such fronts on a handful of levels: this is ONE launch per level where the
// workgroup kernels need five.  Any failed test only raises stat[13]; the host then repeats the factorization with
// that front on the workgroup path (which has the pivoting fallbacks).  A front with a hinted 2x2 pivot fails at once.
// =================================================================================================
constexpr int TINY_N = 64;
constexpr int TINY_CLASSES = 5;                     // unrolled for 24, 28, 32 and 36 columns; up to 64: k_front_blk
static inline int tiny_class(int n) { return n <= 24 ? 0 : n <= 28 ? 1 : n <= 32 ? 2 : n <= 36 ? 3 : 4; }
struct TinyFrontTask {
  int32_t n, m, ld, sptr;
  int64_t loff, coff;
  int32_t iblk, has_contrib, node, nd;   // nd: entries of the front that children add to
  int64_t lfoff, lboff;      // wave tier: the front's packed images (written instead of the rectangle)
  int64_t d0, s0, a0;        // its ranges in the gather lists (gdst/gbeg, gsrc) and in A's entry lists (asrc/aloc)
  int32_t acnt, pad;
};
// Extend-add as a GATHER: for every entry of the front that receives something, the arena offsets of the children's
// entries that land there, children in clist order (the order the scatter form k_assemble_pull adds them in).
//   gdst[k] = LDS offset of the entry (low 12 bits) | number of sources << 12;  gbeg[k] = first source, relative to s0
struct GatherLists {
  const uint32_t* gdst;
  const int32_t* gbeg;
  const int64_t* gsrc;
};
// step 1 of both wave-per-front kernels: the assembled front (lower triangle, packed columns) in LDS
__device__ __forceinline__ void front_assemble(const TinyFrontTask& t, double* Fr, int lane, const GatherLists g,
                                               const int64_t* __restrict__ asrc, const int32_t* __restrict__ aloc,
                                               const double* __restrict__ val, const double* __restrict__ C) {
  const int m = t.m;
  // (the loads of A's entries are issued before the triangle is zeroed)
  double av[2];
  int al[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = lane + 64 * q;
    const bool ok = e < t.acnt;
    al[q] = ok ? aloc[t.a0 + e] : -1;
    av[q] = ok ? val[asrc[t.a0 + (ok ? e : 0)]] : 0.0;
  }
  const int tot = (m * (m + 1)) >> 1;
  for (int e = lane; e < tot; e += 64) Fr[e] = 0.0;
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (al[q] >= 0) Fr[al[q]] = av[q];
  for (int e = lane + 128; e < t.acnt; e += 64) Fr[aloc[t.a0 + e]] = val[asrc[t.a0 + e]];
  STAMPW(56);
  // the children: a lane per receiving entry, its sources eight at a time (their offsets fetched one round ahead)
  for (int k0 = 0; k0 < t.nd; k0 += 64) {
    const bool ok = k0 + lane < t.nd;
    const int64_t k = t.d0 + (ok ? k0 + lane : t.nd - 1);
    const uint32_t dd = g.gdst[k];
    const int64_t* sp = g.gsrc + t.s0 + g.gbeg[k];
    const int dst = int(dd & 4095u), cnt = ok ? int(dd >> 12) : 0;
    int64_t idx[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) idx[q] = sp[min(q, max(cnt - 1, 0))];
    double acc = ok ? Fr[dst] : 0.0;
    for (int p = 0; __any(p < cnt); p += 8) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = C[idx[q]];
#pragma unroll
      for (int q = 0; q < 8; ++q) idx[q] = sp[min(p + 8 + q, max(cnt - 1, 0))];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += (p + q < cnt) ? v[q] : 0.0;
    }
    if (ok) Fr[dst] = acc;
  }
}

template <int NC, int WPB>
__global__ void __launch_bounds__(64 * WPB)
k_front_wave(const TinyFrontTask* __restrict__ tasks, int ntask, const GatherLists g,
             const int64_t* __restrict__ asrc, const int32_t* __restrict__ aloc,
             const double* __restrict__ val, double* __restrict__ L, double* __restrict__ D, double* __restrict__ C,
             int32_t* __restrict__ stat, int32_t* __restrict__ fastok, const uint8_t* __restrict__ hint,
             const uint8_t* __restrict__ tinyskip, int32_t* __restrict__ tinyfail, double small, double u,
             double* __restrict__ Lf, double* __restrict__ Lbk, int tri, int skip_hinted) {
  extern __shared__ __attribute__((aligned(16))) double fsh[];
  __shared__ double psh[WPB][NC];     // per wave: the pivots d_k (for L*D)
#if GSLS_FW_LDSB
  __shared__ __attribute__((aligned(16))) double ush[WPB][64];    // per wave: the pivot column, for its broadcasts
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (ti >= ntask) return;
  const TinyFrontTask t = tasks[ti];
  if (tinyskip[t.node]) return;      // a front this kernel could not take before: it is on the workgroup path
  const int n = t.n, m = t.m, cm = m - n;
  double* Fr = fsh + wave * tri;     // column c (rows c..m-1) at Fr[c*m - c(c+1)/2 + r]
  double* ps = psh[wave];
  // 1x1 pivots only (a second code path per pivot would double a fully unrolled body): a front with a hinted 2x2 pivot
  // is left to k_front_blk (launched behind this kernel when the handle holds hints), or reported
  const bool h2 = (lane < n) ? (hint[t.sptr + lane] != 0) : false;
  const bool hinted = (__ballot(h2) != 0ull);
  if (hinted && skip_hinted) return;
  STAMPW(49);
  front_assemble(t, Fr, lane, g, asrc, aloc, val, C);
  STAMPW(50);
  const double inv_u = (u > 0.0) ? 1.0 / u : INFINITY;
  // ---- 2. rows into registers, the pivots ----------------------------------------------------------------
  double v[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k)
    v[k] = (lane < m && k < n && lane >= k) ? Fr[k * m - ((k * (k + 1)) >> 1) + lane] : 0.0;
  bool bad = hinted;
  int nneg = 0;
  double myd0 = 0.0;                   // lane j: inverse of pivot j
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    // no break / continue in here: the loop must unroll completely (static register indices)
    if (j < n) {
      const double d = readlane_f64(v[j], j);
      if (!(fabs(d) >= small)) bad = true;
      if (d < 0.0) ++nneg;
      double rd = __builtin_amdgcn_rcp(d);         // reciprocal + two Newton steps: full precision
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      const double own = v[j] * rd;
      if (lane > j && lane < m && !(fabs(own) <= inv_u)) bad = true;   // threshold test, whole column
      const double um = v[j];
#if GSLS_FW_LDSB
      // EXPERIMENT (-DGSLS_FW_LDSB=1, off): the pivot column through LDS -- one 16-byte broadcast read per TWO columns on the
      // LDS port instead of four v_readlane_b32 on the VALU port (600 of ~1700 instructions per 24-column front are these
      // broadcasts).  Same operands, same FMAs, same bits -- and 0.69 against 0.67 ms per step on the metric workload
      // (round 3; round 2 measured "no difference" on this kernel's predecessor): the write -> read round trip through LDS
      // sits on every column's critical path.
      ush[wave][lane] = um;
      typedef double double2_t __attribute__((ext_vector_type(2)));
      const double2_t* u2 = reinterpret_cast<const double2_t*>(ush[wave]);
#pragma unroll
      for (int p2 = (j + 1) >> 1; p2 < NC / 2; ++p2) {
        if (NC > 32 && 2 * p2 >= 32 && 2 * p2 >= n) continue;
        const double2_t l2 = u2[p2];
        if (2 * p2 > j) v[2 * p2] = fma(-own, l2.x, v[2 * p2]);
        v[2 * p2 + 1] = fma(-own, l2.y, v[2 * p2 + 1]);
      }
#else
#pragma unroll
      for (int k = j + 1; k < NC; ++k) {
        if (NC > 32 && k >= 32 && k >= n) continue;      // (uniform; the wide variant skips its idle tail)
        const double lkj = readlane_f64(um, k);
        v[k] = fma(-own, lkj, v[k]);
      }
#endif
      v[j] = (lane == j) ? 1.0 : own;
      if (lane == j) myd0 = rd;
      if (lane == 0) ps[j] = d;
    }
  }
  STAMPW(51);
  if (__ballot(bad) != 0ull) {
    if (lane == 0) {
      const int slot = atomicAdd(&stat[13], 1);
      if (slot < FAILCAP) tinyfail[slot] = t.node;
      fastok[t.iblk] = 0;
    }
    return;
  }
  // ---- 3. factors out: column k of L (rows k..m-1), D, statistics -------------------------------------------
  if (Lf && t.lfoff >= 0) {
    // wave tier: straight into the two packed images the solves read (layout: "WAVE TIER" below); nothing
    // reads the rectangle of such a front again
    typedef double double2_t __attribute__((ext_vector_type(2)));
    double2_t* f = reinterpret_cast<double2_t*>(Lf + t.lfoff);
#pragma unroll
    for (int j = 0; j < NC / 2; ++j) {
      const int r0 = 2 * j + 1;
      if (2 * j < n && lane >= r0 && lane < m) {
        double2_t e;
        e.x = v[2 * j];
        e.y = (lane > r0 && r0 < n) ? v[2 * j + 1] : 0.0;
        f[j * (m - 1) - j * (j - 1) + lane - r0] = e;
      }
    }
    // (the backward sweep reads this image too and transposes it in LDS: no second image, wave_bwd_transpose)
  } else {
    double* Lb = L + t.loff;
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k < n && lane >= k && lane < m) Lb[int64_t(k) * t.ld + lane] = v[k];
  }
  if (lane < n) {
    D[2 * int64_t(t.sptr + lane)] = myd0;
    D[2 * int64_t(t.sptr + lane) + 1] = 0.0;
  }
  if (lane == 0) {
    fastok[t.iblk] = 1;
    // tens of thousands of fronts per launch: one shared counter would serialise them in L2
    atomicAdd(&stat[16 + STAT_BINS + (ti & (STAT_BINS - 1))], 1);
    if (nneg) atomicAdd(&stat[16 + (ti & (STAT_BINS - 1))], nneg);
  }
  STAMPW(52);
  if (!t.has_contrib || cm <= 0) return;
  // ---- contribution block: C(i, j) = assembled(i, j) - sum_k (L D)(i, k) L(j, k), rows i >= j below the pivots ------
  // The rows below the pivots go to LDS (over the part of the triangle that held the pivot columns: those are in
  // registers / written out), then a lane per ENTRY of the block forms its sum: cm (cm + 1) / 2 entries share the 64 lanes
  // instead of cm columns x NC broadcasts for the few lanes that hold contribution rows.
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (lane >= n && lane < m) {
    double* ur = Fr + (lane - n) * n;
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (k < n) ur[k] = v[k];
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const int nent = (cm * (cm + 1)) >> 1;
  double* Cb = C + t.coff;
  for (int e0 = 0; e0 < nent; e0 += 64) {
    const int e = min(e0 + lane, nent - 1);
    int i = int((sqrtf(float(8 * e + 1)) - 1.0f) * 0.5f);
    i += (((i + 1) * (i + 2)) >> 1 <= e) ? 1 : 0;
    i -= ((i * (i + 1)) >> 1 > e) ? 1 : 0;
    const int j = e - ((i * (i + 1)) >> 1);
    const double* ui = Fr + i * n;
    const double* uj = Fr + j * n;
    double acc = 0.0;
#pragma unroll 4
    for (int k = 0; k < n; ++k) acc = fma(ui[k] * ps[k], uj[k], acc);
    const int c = n + j;
    const double base = Fr[c * m - ((c * (c + 1)) >> 1) + n + i];
    if (e0 + lane < nent) Cb[int64_t(j) * cm + i] = base - acc;
  }
}

// ---- LDL^T flavour: complete pivoting (1x1 and 2x2) inside the w x w diagonal block, applied to the
// whole 128-row panel, a-posteriori threshold test |l_ij| <= 1/u on the rows below the block.
// Behavioural model: block_ldlt<32> (ssids/cpu/kernels/block_ldlt.hxx:257-412: largest remaining
// entry picks a 1x1 or a 2x2 pivot, test_2x2 :210-215) inside the a-posteriori scheme of
// ldlt_app.cxx:303-321 (check_threshold).  A column that fails the test is counted in stat[4]; the
// host then abandons this optimistic pass (see gsls_api.cpp).
// stat[1] #zero pivots, stat[2] #negative eigenvalues, stat[3] #2x2 pivots, stat[4] #failed columns
__device__ __forceinline__ void swap_sym(double* P, int32_t* lperm, int pr, int c1, int c2, int tid) {
  if (c1 == c2) return;
  if (c2 < c1) { const int t = c1; c1 = c2; c2 = t; }
  const int i = tid;
  if (i < pr && i != c2) {
    double *x, *y;
    if (i < c1) { x = &P[i * LDP + c1]; y = &P[i * LDP + c2]; }
    else if (i == c1) { x = &P[c1 * LDP + c1]; y = &P[c2 * LDP + c2]; }
    else if (i < c2) { x = &P[c1 * LDP + i]; y = &P[i * LDP + c2]; }
    else { x = &P[c1 * LDP + i]; y = &P[c2 * LDP + i]; }
    const double tv = *x; *x = *y; *y = tv;
  }
  if (tid == 255) { const int t = lperm[c1]; lperm[c1] = lperm[c2]; lperm[c2] = t; }
}

__global__ void __launch_bounds__(256)
k_diag_ldlt(const NodeDesc* __restrict__ nodes, const PanelTask* __restrict__ tasks,
            double* __restrict__ L, double* __restrict__ D, int32_t* __restrict__ gperm,
            int32_t* __restrict__ stat, int32_t* __restrict__ faillist, const int32_t* __restrict__ fastok,
            double small, double u, const uint8_t* __restrict__ tppflag) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Stage<PR>& sg = *reinterpret_cast<Stage<PR>*>(smem_raw);
  double* P = reinterpret_cast<double*>(smem_raw);
  __shared__ double dg[2 * NB + 4];
  __shared__ double w1[PR], w2[PR];
  __shared__ double rv[4];
  __shared__ int32_t ri[4];
  __shared__ int32_t lperm[NB];

  const PanelTask t = tasks[blockIdx.x];
  if (tppflag[t.node]) return;               // the whole front goes through k_front_tpp
  const NodeDesc nd = nodes[t.node];
  if (fastok[nd.iblk + t.step]) return;      // the optimistic pass (k_diag_fast<true>) already did this block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kb = t.step * NB;
  const int w = min(NB, nd.n - kb);
  const int pr = min(PR, nd.m - kb);
  double* Lb = L + nd.loff;
  const double* dinv = D + 2 * int64_t(nd.sptr);

  double4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
  if (kb > 0) {
    StageRegs<PR> rg;
    stage_load<PR>(rg, Lb, nd.ld, kb, nd.m, kb, kb + w, 0, kb, dinv, tid);
    for (int k0 = 0; k0 < kb; k0 += KC) {
      __syncthreads();
      stage_store<PR>(sg, rg, tid);
      __syncthreads();
      if (k0 + KC < kb) stage_load<PR>(rg, Lb, nd.ld, kb, nd.m, kb, kb + w, k0 + KC, kb, dinv, tid);
      mfma_panel<PR, 2, 4>(sg, 32 * wave, 0, lane, acc);
    }
  }
  __syncthreads();
  {
    const int lr = lane & 15, lq = lane >> 4;
    double g[2][4][4];   // all 32 loads in flight before the first LDS store
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 32 * wave + 16 * i + lq + 4 * r;
          const int col = 16 * j + lr;
          g[i][j][r] = (row < pr && col < w) ? Lb[int64_t(kb + col) * nd.ld + kb + row] : 0.0;
        }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 32 * wave + 16 * i + lq + 4 * r;
          const int col = 16 * j + lr;
          P[col * LDP + row] = (row < pr && col < w) ? g[i][j][r] - acc[i][j][r] : 0.0;
        }
  }
  if (tid < NB) lperm[tid] = tid;
  __syncthreads();

  const int r = tid & (PR - 1), kofs = tid >> 7;
  const double inv_u = (u > 0.0) ? 1.0 / u : INFINITY;
  int nfail = 0;      // uniform
  int fail_from = NB; // uniform: columns [fail_from, w) of the block could not be eliminated here
  int bigcol = NB;    // per thread: first column with an |l| above 1/u below the block
  int p = 0;
#ifdef GSLS_STAMPS
  unsigned long long phacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pht = __builtin_amdgcn_s_memtime();
  const unsigned long long ph0 = pht;
#endif
  while (p < w) {
    PH(7);
    // ---- the next pivot in the given order first: accepted as a 1x1 if it passes the threshold test
    // against its whole column (|a_rp| <= |a_pp| / u for every row below, ldlt_app.cxx:303-321) -- one LDS
    // read per thread and one barrier instead of the block-wide search below
    double bv = -1.0;
    int bi = INT_MAX;
    bool natural = false;
    {
      const double app = P[p * LDP + p];
      bool viol = !(fabs(app) >= small);
      if (kofs == 0 && r > p && r < pr) viol |= !(fabs(P[p * LDP + r]) <= fabs(app) * inv_u);
      natural = !__syncthreads_or(viol);
      if (natural) { bv = fabs(app); bi = p * 64 + p; }
    }
    PH(7);
    // ---- otherwise the largest remaining entry of the block (lower triangle, rows/cols p..w-1) ----
    if (!natural) {
      {
      const int rr = p + (tid & 63);
      if (rr < w) {
        double cand[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int c = p + (tid >> 6) + 4 * q;
          cand[q] = (c <= rr) ? fabs(P[c * LDP + rr]) : -2.0;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int id = (p + (tid >> 6) + 4 * q) * 64 + rr;
          if (cand[q] > bv || (cand[q] == bv && id < bi)) { bv = cand[q]; bi = id; }
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_down(bv, o);
      const int oi = __shfl_down(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    PH(0);
    if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
    __syncthreads();
    PH(1);
    bv = rv[0]; bi = ri[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (rv[k] > bv || (rv[k] == bv && ri[k] < bi)) { bv = rv[k]; bi = ri[k]; }
    }

    if (!(bv >= small)) {
      // nothing usable left in the block: zero pivots if the rows below are negligible as well
      bool nz = false;
      if (kofs == 0 && r >= w && r < pr)
        for (int c = p; c < w; ++c) nz |= !(fabs(P[c * LDP + r]) < small);
      if (__syncthreads_or(nz)) { nfail += w - p; fail_from = p; }
      for (int c = p + kofs; c < w; c += 2)
        if (r > c && r < pr) P[c * LDP + r] = 0.0;
      if (tid >= p && tid < w) { dg[2 * tid] = 0.0; dg[2 * tid + 1] = 0.0; }
      __syncthreads();
      for (int c = p + tid; c < w; c += 256) P[c * LDP + c] = 1.0;
      break;
    }
    int mcol = bi >> 6, trow = bi & 63, pivsiz;
    double a11, a21 = 0.0, a22 = 0.0, detpiv = 0.0, detscale = 0.0;
    if (trow == mcol) {
      a11 = P[trow * LDP + trow];
      pivsiz = 1;
    } else {
      a11 = P[mcol * LDP + mcol];
      a22 = P[trow * LDP + trow];
      a21 = P[mcol * LDP + trow];
      detscale = 1.0 / fabs(a21);
      detpiv = (a11 * detscale) * a22 - fabs(a21);
      if (fabs(detpiv) >= fabs(a21) / 2) {
        pivsiz = 2;
      } else if (fabs(a11) > fabs(a22)) {
        pivsiz = (fabs(a11 / a21) < u) ? 0 : 1;
        trow = mcol;
      } else {
        pivsiz = (fabs(a22 / a21) < u) ? 0 : 1;
        a11 = a22;
        mcol = trow;
      }
    }
    PH(2);
    __syncthreads();   // every thread has read the candidates
    PH(3);
    if (pivsiz == 0) { nfail += w - p; fail_from = p; break; }
    if (pivsiz == 1) {
      swap_sym(P, lperm, pr, p, trow, tid);
      __syncthreads();
      PH(4);
      const double d11 = 1.0 / a11;
      if (kofs == 0 && r > p && r < pr) w1[r] = P[p * LDP + r];
      __syncthreads();
      PH(5);
      const double l = (r > p && r < pr) ? w1[r] * d11 : 0.0;
      // rank-1 update of this thread's row, every second column: operands into registers first (the
      // compiler cannot reorder LDS loads across the stores)
      for (int c0 = p + 1 + kofs; c0 < w; c0 += 32) {
        double pv[16], wv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int c = c0 + 2 * q;
          const bool ok = (c < w && r >= c && r < pr);
          pv[q] = ok ? P[c * LDP + r] : 0.0;
          wv[q] = ok ? w1[c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int c = c0 + 2 * q;
          if (c < w && r >= c && r < pr) P[c * LDP + r] = pv[q] - l * wv[q];
        }
      }
      if (kofs == 0) {
        if (r > p && r < pr) {
          P[p * LDP + r] = l;
          if (r >= w && fabs(l) > inv_u) bigcol = min(bigcol, p);
        }
        if (r == p) { P[p * LDP + p] = 1.0; dg[2 * p] = d11; dg[2 * p + 1] = 0.0; }
      }
      __syncthreads();
      PH(6);
      p += 1;
    } else {
      swap_sym(P, lperm, pr, p, mcol, tid);
      __syncthreads();
      swap_sym(P, lperm, pr, p + 1, trow, tid);
      __syncthreads();
      const double d11 = (a22 * detscale) / detpiv;
      const double d22 = (a11 * detscale) / detpiv;
      const double d21 = (-a21 * detscale) / detpiv;
      if (kofs == 0 && r > p + 1 && r < pr) { w1[r] = P[p * LDP + r]; w2[r] = P[(p + 1) * LDP + r]; }
      __syncthreads();
      double l1 = 0.0, l2 = 0.0;
      if (r > p + 1 && r < pr) { l1 = d11 * w1[r] + d21 * w2[r]; l2 = d21 * w1[r] + d22 * w2[r]; }
      for (int c0 = p + 2 + kofs; c0 < w; c0 += 32) {
        double pv[16], wv1[16], wv2[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int c = c0 + 2 * q;
          const bool ok = (c < w && r >= c && r < pr);
          pv[q] = ok ? P[c * LDP + r] : 0.0;
          wv1[q] = ok ? w1[c] : 0.0;
          wv2[q] = ok ? w2[c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int c = c0 + 2 * q;
          if (c < w && r >= c && r < pr) P[c * LDP + r] = pv[q] - (wv1[q] * l1 + wv2[q] * l2);
        }
      }
      if (kofs == 0) {
        if (r > p + 1 && r < pr) {
          P[p * LDP + r] = l1;
          P[(p + 1) * LDP + r] = l2;
          if (r >= w && (fabs(l1) > inv_u || fabs(l2) > inv_u)) bigcol = min(bigcol, p);
        }
        if (r == p) {
          P[p * LDP + p] = 1.0; P[p * LDP + p + 1] = 0.0; P[(p + 1) * LDP + p + 1] = 1.0;
          dg[2 * p] = d11; dg[2 * p + 1] = d21; dg[2 * p + 2] = INFINITY; dg[2 * p + 3] = d22;
        }
      }
      __syncthreads();
      p += 2;
    }
  }
  __syncthreads();
#ifdef GSLS_STAMPS
  if (tid == 0) { for (int q = 0; q < 8; ++q) g_stamps[40 + q] = phacc[q]; g_stamps[48] = pht - ph0; }
#endif
  // first column that failed the a-posteriori test (min over the workgroup, via LDS)
  __shared__ int32_t s_bigcol;
  if (tid == 0) s_bigcol = NB;
  __syncthreads();
  if (bigcol < NB) atomicMin(&s_bigcol, bigcol);
  __syncthreads();
  bigcol = s_bigcol;
  if (bigcol < NB) nfail = max(nfail, 1);

  // ---- store: factors, pivots, the permutation, inertia ----------------------------------------------
  for (int e = tid; e < pr * w; e += 256) {
    const int row = e % pr, col = e / pr;
    if (row >= col) Lb[int64_t(kb + col) * nd.ld + kb + row] = P[col * LDP + row];
  }
  if (tid < w) {
    D[2 * int64_t(nd.sptr + kb + tid)] = dg[2 * tid];
    D[2 * int64_t(nd.sptr + kb + tid) + 1] = dg[2 * tid + 1];
    gperm[nd.sptr + kb + tid] = nd.sptr + kb + lperm[tid];
  }
  // rows of the front's earlier block columns follow the permutation (one wave per column: the
  // gather completes before the same wave's store issues)
  if (kb > 0 && lane < w) {
    const int src = lperm[lane];
    for (int k = wave; k < kb; k += 4) {
      const double v = Lb[int64_t(k) * nd.ld + kb + src];
      Lb[int64_t(k) * nd.ld + kb + lane] = v;
    }
  }
  if (tid == 0) {
    int nneg = 0, ntwo = 0, nzero = 0;
    for (int i = 0; i < w;) {
      const double a11 = dg[2 * i], a21 = dg[2 * i + 1];
      if (i + 1 == w || !isinf(dg[2 * i + 2])) {
        if (a11 == 0.0) ++nzero;
        if (a11 < 0.0) ++nneg;
        i += 1;
      } else {
        const double a22 = dg[2 * i + 3];
        ++ntwo;
        const double det = a11 * a22 - a21 * a21, tr = a11 + a22;
        if (det < 0) nneg += 1;
        else if (tr < 0) nneg += 2;
        i += 2;
      }
    }
    if (nzero) atomicAdd(&stat[1], nzero);
    if (nneg) atomicAdd(&stat[2], nneg);
    if (ntwo) atomicAdd(&stat[3], ntwo);
    if (nfail) {
      atomicAdd(&stat[4], nfail);
      // report which variables (analyse-time pivot positions) must be eliminated later
      const int b2 = (bigcol < NB) ? bigcol + ((bigcol + 1 < w && isinf(dg[2 * bigcol + 2])) ? 2 : 1) : bigcol;
      for (int j = 0; j < w; ++j) {
        const bool bad = (j >= fail_from) || (bigcol < NB && j >= bigcol && j < b2);
        if (!bad) continue;
        const int slot = atomicAdd(&stat[5], 1);
        if (slot < FAILCAP) faillist[slot] = nd.sptr + kb + lperm[j];
      }
    }
  }
}

// =================================================================================================
// Threshold partial pivoting over a WHOLE front -- the fallback that always terminates.  The blocked
// kernels above search for pivots inside one 64-column block only; a front whose pivots need partners
// from another block (K = [0 B; B^T 0]: every elimination is a 2x2 pivot) is flagged by the host and comes
// here instead: ONE workgroup per front, right-looking, in global memory, pivot search across all of the
// front's fully summed columns.  Behavioural model: ldlt_tpp_factor (ssids/cpu/kernels/ldlt_tpp.cxx:140-240;
// candidate order, test_2x2 :99-130, the 1x1 threshold test and the zero-column handling are restated 1:1),
// called by the reference on the columns its blocked APP kernel could not eliminate (factor.hxx:74-106).
// Columns that still fail are delayed: reported in faillist (the host moves them to the parent front,
// gsls_api.cpp), except at a root front (m == n: nothing is left to wait for), where they are recorded as zero
// pivots like the reference's delays out of a root.  Not a fast kernel: it runs on the few fronts that need it.
// The n x n pivot block is mirrored to full symmetric storage first (the rectangle has the room), so that
// a column's entries are contiguous whichever side of the diagonal they lie on.
// =================================================================================================
__device__ __forceinline__ void tpp_sync() {
  __threadfence_block();
  __syncthreads();
}
// block-wide max |.| (every thread gets the result)
__device__ __forceinline__ double tpp_max(double v, double* redv) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) redv[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(redv[0], redv[1]), fmax(redv[2], redv[3]));
}
// block-wide (max, smallest index attaining it)
__device__ __forceinline__ void tpp_argmax(double& v, int& i, double* redv, int* redi) {
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(i, o);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = v; redi[threadIdx.x >> 6] = i; }
  __syncthreads();
  v = redv[0]; i = redi[0];
  for (int k = 1; k < 4; ++k)
    if (redv[k] > v || (redv[k] == v && redi[k] < i)) { v = redv[k]; i = redi[k]; }
}

__global__ void __launch_bounds__(256)
k_front_tpp(const NodeDesc* __restrict__ nodes, const int32_t* __restrict__ list, double* __restrict__ L,
            double* __restrict__ D, int32_t* __restrict__ gperm, int32_t* __restrict__ stat,
            int32_t* __restrict__ faillist, double small, double u, int nnodes) {
  __shared__ double redv[8];
  __shared__ int redi[4];
  __shared__ int cnt[3];
  const NodeDesc nd = nodes[list[blockIdx.x]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = nd.m, n = nd.n;
  const int64_t ld = nd.ld;
  double* A = L + nd.loff;
  double* Dn = D + 2 * int64_t(nd.sptr);
  int32_t* pm = gperm + nd.sptr;
  const bool root = (nd.parent >= nnodes) || (m == n);
  if (tid < 3) cnt[tid] = 0;
  // full symmetric storage of the pivot block
  for (int c = wave; c < n; c += 4)
    for (int i = lane; i < c; i += 64) A[c * ld + i] = A[i * ld + c];
  tpp_sync();

  // symmetric swap of positions x < y of the trailing matrix (rows of the finished columns follow)
  auto swap_sym = [&](int x, int y, int p) {
    if (x == y) return;           // uniform
    if (y < x) { const int t = x; x = y; y = t; }
    for (int c = tid; c < n; c += 256) {
      const double a = A[c * ld + x], b = A[c * ld + y];
      A[c * ld + x] = b;
      A[c * ld + y] = a;
    }
    tpp_sync();
    for (int i = p + tid; i < m; i += 256) {
      const double a = A[x * ld + i], b = A[y * ld + i];
      A[x * ld + i] = b;
      A[y * ld + i] = a;
    }
    if (tid == 0) { const int t = pm[x]; pm[x] = pm[y]; pm[y] = t; }
    tpp_sync();
  };
  // column p has no entry of size `small` or more: a zero pivot (ldlt_tpp.cxx:150-160)
  auto zero_pivot = [&](int p) {
    for (int i = p + tid; i < m; i += 256) A[p * ld + i] = (i == p) ? 1.0 : 0.0;
    if (tid == 0) { Dn[2 * p] = 0.0; Dn[2 * p + 1] = 0.0; }
    tpp_sync();
  };
  auto pivot_1x1 = [&](int p) {
    const double app = A[p * ld + p];
    const double d = 1.0 / app;
    for (int c = p + 1 + wave; c < n; c += 4) {
      const double acp = A[p * ld + c];
      for (int i = p + 1 + lane; i < m; i += 64) A[c * ld + i] -= (A[p * ld + i] * acp) * d;
    }
    tpp_sync();
    for (int i = p + tid; i < m; i += 256) A[p * ld + i] = (i == p) ? 1.0 : A[p * ld + i] * d;
    if (tid == 0) { Dn[2 * p] = d; Dn[2 * p + 1] = 0.0; }
    tpp_sync();
  };
  auto pivot_2x2 = [&](int p, double d11, double d21, double d22) {
    for (int c = p + 2 + wave; c < n; c += 4) {
      const double a1c = A[p * ld + c], a2c = A[(p + 1) * ld + c];
      for (int i = p + 2 + lane; i < m; i += 64) {
        const double a1i = A[p * ld + i], a2i = A[(p + 1) * ld + i];
        A[c * ld + i] -= d11 * (a1i * a1c) + d21 * (a2i * a1c + a1i * a2c) + d22 * (a2i * a2c);
      }
    }
    tpp_sync();
    for (int i = p + tid; i < m; i += 256) {
      const double a1 = A[p * ld + i], a2 = A[(p + 1) * ld + i];
      A[p * ld + i] = (i == p) ? 1.0 : ((i == p + 1) ? 0.0 : d11 * a1 + d21 * a2);
      A[(p + 1) * ld + i] = (i == p) ? 0.0 : ((i == p + 1) ? 1.0 : d21 * a1 + d22 * a2);
    }
    if (tid == 0) { Dn[2 * p] = d11; Dn[2 * p + 1] = d21; Dn[2 * p + 2] = INFINITY; Dn[2 * p + 3] = d22; }
    tpp_sync();
  };
  // max |A(i, col)| over rows [from, m) except rows x1, x2
  auto col_max_excl = [&](int col, int from, int x1, int x2) -> double {
    double v = 0.0;
    for (int i = from + tid; i < m; i += 256)
      if (i != x1 && i != x2) v = fmax(v, fabs(A[col * ld + i]));
    return tpp_max(v, redv);
  };

  int p = 0;
  while (p < n) {
    if (col_max_excl(p, p, -1, -1) < small) { zero_pivot(p); ++p; continue; }
    bool found = false;
    for (int q = p + 1; q < n && !found; ++q) {
      // column q: largest entry overall (negligible column?) and the largest one among the rows [p, q)
      double mall = 0.0, mv = -1.0;
      int mi = INT_MAX;
      for (int i = p + tid; i < m; i += 256) {
        const double a = fabs(A[q * ld + i]);
        mall = fmax(mall, a);
        if (i < q && (a > mv || (a == mv && i < mi))) { mv = a; mi = i; }
      }
      mall = tpp_max(mall, redv);
      if (mall < small) {
        swap_sym(p, q, p);
        zero_pivot(p);
        ++p;
        found = true;
        break;
      }
      tpp_argmax(mv, mi, redv + 4, redi);
      const int t = mi;
      const double maxt = col_max_excl(t, p, t, q);
      double maxq = col_max_excl(q, p, q, t);
      const double a11 = A[t * ld + t], a21 = A[t * ld + q], a22 = A[q * ld + q];
      // test_2x2 (ldlt_tpp.cxx:99-130)
      bool ok2 = false;
      double d11 = 0.0, d21 = 0.0, d22 = 0.0;
      const double maxpiv = fmax(fabs(a11), fmax(fabs(a21), fabs(a22)));
      if (maxpiv >= small) {
        const double detscale = 1.0 / maxpiv;
        const double detpiv0 = (a11 * detscale) * a22, detpiv1 = (a21 * detscale) * a21;
        const double detpiv = detpiv0 - detpiv1;
        if (!(fabs(detpiv) < fmax(small, fmax(fabs(detpiv0 / 2), fabs(detpiv1 / 2))))) {
          d11 = (a22 * detscale) / detpiv;
          d21 = (-a21 * detscale) / detpiv;
          d22 = (a11 * detscale) / detpiv;
          if (fmax(maxq, maxt) < small) ok2 = true;
          else {
            const double x1 = fabs(d11) * maxt + fabs(d21) * maxq;
            const double x2 = fabs(d21) * maxt + fabs(d22) * maxq;
            ok2 = (u * fmax(x1, x2) < 1.0);
          }
        }
      }
      if (ok2) {
        swap_sym(t, p, p);
        swap_sym(q, p + 1, p);
        pivot_2x2(p, d11, d21, d22);
        p += 2;
        found = true;
        break;
      }
      maxq = fmax(maxq, fabs(a21));
      if (fabs(a22) >= u * maxq && fabs(a22) >= small) {
        swap_sym(q, p, p);
        pivot_1x1(p);
        p += 1;
        found = true;
        break;
      }
    }
    if (found) continue;
    // last resort: column p itself as a 1x1 pivot
    const double maxp = col_max_excl(p, p, p, -1);
    const double app = A[p * ld + p];
    if (fabs(app) >= u * maxp && fabs(app) >= small) {
      pivot_1x1(p);
      ++p;
    } else {
      break;                      // no more pivots in this front
    }
  }
  const int nelim = p;
  if (nelim < n && root) {          // nowhere to delay to: zero pivots (cf. delays out of a root, factor.hxx:118-119)
    for (int c = nelim; c < n; ++c) zero_pivot(c);
  }
  const int ndone = (nelim < n && !root) ? nelim : n;
  // inertia of what was eliminated
  {
    int nneg = 0, ntwo = 0, nzero = 0;
    for (int i = tid; i < ndone; i += 256) {
      const double a11 = Dn[2 * i];
      if (isinf(a11)) continue;                                   // second half of a 2x2 pivot
      if (i + 1 < ndone && isinf(Dn[2 * i + 2])) {
        const double a21 = Dn[2 * i + 1], a22 = Dn[2 * i + 3];
        ++ntwo;
        const double det = a11 * a22 - a21 * a21, tr = a11 + a22;
        if (det < 0) nneg += 1;
        else if (tr < 0) nneg += 2;
      } else {
        if (a11 == 0.0) ++nzero;
        if (a11 < 0.0) ++nneg;
      }
    }
    if (nneg) atomicAdd(&cnt[0], nneg);
    if (ntwo) atomicAdd(&cnt[1], ntwo);
    if (nzero) atomicAdd(&cnt[2], nzero);
    __syncthreads();
    if (tid == 0) {
      if (cnt[2]) atomicAdd(&stat[1], cnt[2]);
      if (cnt[0]) atomicAdd(&stat[2], cnt[0]);
      if (cnt[1]) atomicAdd(&stat[3], cnt[1]);
      atomicAdd(&stat[14], 1);
      if (ndone < n) {
        atomicAdd(&stat[4], n - ndone);
        atomicAdd(&stat[15], n - ndone);
      }
    }
  }
  if (ndone < n) {
    for (int j = ndone + tid; j < n; j += 256) {
      const int slot = atomicAdd(&stat[5], 1);
      if (slot < FAILCAP) faillist[slot] = pm[j];
    }
  }
  // the rectangle's upper triangle is nobody's input: leave it clean
  for (int c = 1 + wave; c < n; c += 4)
    for (int i = lane; i < c; i += 64) A[c * ld + i] = 0.0;
}

// =================================================================================================
// panel kernel: row chunk c>=1 of block column `step`:  rows [kb+128+(c-1)*64, +64)
//   R = (A - L[rows,0:kb] D L[kb:kb+w,0:kb]^T) * L11^-T * D11^-1
// =================================================================================================
constexpr int RBP = RB + 1;   // odd LDS stride: column-strided register loads are conflict free

template <bool POSDEF>
__global__ void __launch_bounds__(256)
k_panel(const NodeDesc* __restrict__ nodes, const PanelTask* __restrict__ tasks,
        double* __restrict__ L, const double* __restrict__ D, const int32_t* __restrict__ gperm,
        int32_t* __restrict__ stat, int32_t* __restrict__ faillist, double u, const uint8_t* __restrict__ tppflag,
        double small) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Stage<RB>& sg = *reinterpret_cast<Stage<RB>*>(smem_raw);
  double* Pc = reinterpret_cast<double*>(smem_raw);                  // [w][RBP]
  double* L11 = reinterpret_cast<double*>(smem_raw) + NB * RBP;      // [w][NB] column-major
  __shared__ double dsc[2 * NB + 4];
  __shared__ int32_t lp[NB];

  const PanelTask t = tasks[blockIdx.x];
  if (!POSDEF && tppflag[t.node]) return;
  const NodeDesc nd = nodes[t.node];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kb = t.step * NB;
  const int w = min(NB, nd.n - kb);
  const int r0 = kb + PR + (t.chunk - 1) * RB;
  const int rows = min(RB, nd.m - r0);
  double* Lb = L + nd.loff;
  const double* dinv = POSDEF ? nullptr : (D + 2 * int64_t(nd.sptr));

  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1);
  if (kb > 0) {
    StageRegs<RB> rg;
    stage_load<RB>(rg, Lb, nd.ld, r0, nd.m, kb, kb + w, 0, kb, dinv, tid);
    for (int k0 = 0; k0 < kb; k0 += KC) {
      __syncthreads();
      stage_store<RB>(sg, rg, tid);
      __syncthreads();
      if (k0 + KC < kb) stage_load<RB>(rg, Lb, nd.ld, r0, nd.m, kb, kb + w, k0 + KC, kb, dinv, tid);
      mfma_panel<RB, 2, 2>(sg, wr, wc, lane, acc);
    }
  }
  if (tid < NB) lp[tid] = (POSDEF || tid >= w) ? tid : gperm[nd.sptr + kb + tid] - (nd.sptr + kb);
  __syncthreads();
  {
    const int lr = lane & 15, lq = lane >> 4;
    double g[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr + 16 * i + lq + 4 * r;
          const int col = wc + 16 * j + lr;
          // the block's pivoting permuted its columns: gather column lp[col] of A
          g[i][j][r] = (row < rows && col < w) ? Lb[int64_t(kb + lp[col]) * nd.ld + r0 + row] : 0.0;
        }
    double l11[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = tid + 256 * t;
      const int row = e & 63, col = e >> 6;
      l11[t] = (row < w && col < w && row >= col) ? Lb[int64_t(kb + col) * nd.ld + kb + row] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wr + 16 * i + lq + 4 * r;
          const int col = wc + 16 * j + lr;
          Pc[col * RBP + row] = (row < rows && col < w) ? g[i][j][r] - acc[i][j][r] : 0.0;
        }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int e = tid + 256 * t;
      L11[(e >> 6) * NB + (e & 63)] = l11[t];
    }
  }
  if (POSDEF) {
    if (tid < w) dsc[tid] = 1.0 / Lb[int64_t(kb + tid) * nd.ld + kb + tid];
  } else {
    for (int e = tid; e < 2 * w + 2; e += 256) dsc[e] = (e < 2 * w) ? dinv[2 * kb + e] : 0.0;
  }
  __syncthreads();

  // forward substitution against L11 (unit for LDL^T).  Thread (c, q) owns rows [16q, 16q+16) of
  // column c in registers (static indices, run-time column loop); per column: its owners publish the
  // finished (L D) column, one barrier, the columns to the right take their update.
  const double inv_u = (!POSDEF && u > 0.0) ? 1.0 / u : INFINITY;
  int bigcol = NB;   // first column whose multiplier breaks |l| <= 1/u in this chunk
  {
    const int c = tid & 63, q = tid >> 6;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = Pc[c * RBP + 16 * q + i];
    __syncthreads();
    double* xb = Pc;             // 2 x RB doubles, double buffered
    for (int j = 0; j < w; ++j) {
      double* cb = xb + (j & 1) * RB;
      if (c == j) {
        const double sc = POSDEF ? dsc[j] : 1.0;   // posdef: l = a / l_jj ; indef: keep a = (L D)_j
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          a[i] *= sc;
          cb[16 * q + i] = a[i];
        }
      }
      __syncthreads();
      if (c > j && c < w) {
        const double f = L11[j * NB + c];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] -= cb[16 * q + i] * f;
      }
    }
    __syncthreads();
    // D^-1 scaling (1x1 / 2x2) needs the neighbouring column: go back through LDS
#pragma unroll
    for (int i = 0; i < 16; ++i) Pc[c * RBP + 16 * q + i] = a[i];
    __syncthreads();
    for (int e = tid; e < rows * w; e += 256) {
      const int r = e % rows, j = e / rows;
      double* out = Lb + int64_t(kb + j) * nd.ld + r0 + r;
      const double aj = Pc[j * RBP + r];
      if (POSDEF) {
        *out = aj;
      } else if (isinf(dsc[2 * j])) {
        const double a1 = Pc[(j - 1) * RBP + r];
        const double l2 = dsc[2 * j - 1] * a1 + dsc[2 * j + 1] * aj;
        *out = l2;
        if (fabs(l2) > inv_u) bigcol = min(bigcol, j - 1);
      } else if (j + 1 < w && isinf(dsc[2 * j + 2])) {
        const double a2 = Pc[(j + 1) * RBP + r];
        const double l1 = dsc[2 * j] * aj + dsc[2 * j + 1] * a2;
        *out = l1;
        if (fabs(l1) > inv_u) bigcol = min(bigcol, j);
      } else {
        const double l = aj * dsc[2 * j];
        *out = l;
        if (fabs(l) > inv_u) bigcol = min(bigcol, j);
        // a column the diagonal kernel declared a ZERO pivot (it sees the first 128 rows of the front only) that
        // has a non-negligible entry down here is not a zero column (ldlt_tpp.cxx:250-262 tests the whole column):
        // it failed, like any column whose multipliers are too large
        if (dsc[2 * j] == 0.0 && !(fabs(aj) < small)) bigcol = min(bigcol, j);
      }
    }
  }
  if (!POSDEF) {
    __shared__ int32_t s_bigcol;
    if (tid == 0) s_bigcol = NB;
    __syncthreads();
    if (bigcol < NB) atomicMin(&s_bigcol, bigcol);
    __syncthreads();
    if (tid == 0 && s_bigcol < NB) {
      const int j = s_bigcol;
      const int cnt = (j + 1 < w && isinf(dsc[2 * j + 2])) ? 2 : 1;
      atomicAdd(&stat[4], cnt);
      for (int k = j; k < j + cnt; ++k) {
        const int slot = atomicAdd(&stat[5], 1);
        if (slot < FAILCAP) faillist[slot] = gperm[nd.sptr + kb + k];
      }
    }
  }
}

// =================================================================================================
// panel kernel, Cholesky flavour: row chunk c>=1 of block column `step`
//   R = (A - L[rows,0:kb] L[kb:kb+w,0:kb]^T) * W,   W = L11^-T from the diag kernel
// both products on the matrix cores; no substitution loop.
// =================================================================================================
__global__ void __launch_bounds__(256)
k_panel_chol(const NodeDesc* __restrict__ nodes, const PanelTask* __restrict__ tasks,
             double* __restrict__ L, const double* __restrict__ Linv) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Stage<RB, CK>& sg = *reinterpret_cast<Stage<RB, CK>*>(smem_raw);
  double* Pc = reinterpret_cast<double*>(smem_raw);                  // [k][RBP]: R'[row][k]
  double* Ws = reinterpret_cast<double*>(smem_raw) + NB * RBP;       // [n][RBP]: X[n][k] = W[k + 64 n]

  const PanelTask t = tasks[blockIdx.x];
  const NodeDesc nd = nodes[t.node];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int kb = t.step * NB;
  const int w = min(NB, nd.n - kb);
  const int r0 = kb + PR + (t.chunk - 1) * RB;
  const int rows = min(RB, nd.m - r0);
  double* Lb = L + nd.loff;

  // this chunk's own entries and W first, so that their latency hides behind the left-looking update
  const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1);
  double g[2][2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr + 16 * i + lr;
        const int col = wc + 16 * j + lq + 4 * r;
        g[i][j][r] = (row < rows && col < w) ? Lb[int64_t(kb + col) * nd.ld + r0 + row] : 0.0;
      }
  const double* W = Linv + (int64_t(nd.iblk) + t.step) * (NB * NB);   // written by this step's diag kernel
  double wv[16];
#pragma unroll
  for (int tt = 0; tt < 16; ++tt) wv[tt] = W[tid + 256 * tt];
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
  if (kb > 0) {
    StageRegs<RB, CK> rg;
    stage_load<RB, CK>(rg, Lb, nd.ld, r0, nd.m, kb, kb + w, 0, kb, nullptr, tid);
    for (int k0 = 0; k0 < kb; k0 += CK) {
      __syncthreads();
      stage_store<RB, CK>(sg, rg, tid);
      __syncthreads();
      if (k0 + CK < kb) stage_load<RB, CK>(rg, Lb, nd.ld, r0, nd.m, kb, kb + w, k0 + CK, kb, nullptr, tid);
      mfma_panel<RB, 2, 2, CK, true>(sg, wr, wc, lane, acc);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr + 16 * i + lr;
        const int col = wc + 16 * j + lq + 4 * r;
        Pc[col * RBP + row] = (row < rows && col < w) ? g[i][j][r] - acc[i][j][r] : 0.0;
      }
#pragma unroll
  for (int tt = 0; tt < 16; ++tt) {
    const int e = tid + 256 * tt;
    Ws[(e >> 6) * RBP + (e & 63)] = wv[tt];
  }
  __syncthreads();
  // Y^T = X R'^T: wave -> 16 rows of the chunk, all four column tiles; X is lower triangular, so
  // column tile ct only needs k < 16 (ct + 1)
  {
    const int row0 = 16 * wave;
    double4_t y[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) y[ct] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < NB; k4 += 4) {
      const double rb = Pc[(k4 + lq) * RBP + row0 + lr];            // B[k][j=row] = R'[row][k]
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (k4 < 16 * (ct + 1)) {
          const double xa = Ws[(16 * ct + lr) * RBP + k4 + lq];     // A[i=c][k] = X[c][k]
          y[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, rb, y[ct], 0, 0, 0);
        }
      }
    }
    const int row = row0 + lr;
    if (row < rows) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * ct + lq + 4 * r;
          if (c < w) Lb[int64_t(kb + c) * nd.ld + r0 + row] = y[ct][r];
        }
    }
  }
}

// =================================================================================================
// contribution kernel: tile (ti,tj) of  C -= L21 * D * L21^T   (K = n, MFMA)
// =================================================================================================
template <bool POSDEF>
__global__ void __launch_bounds__(256, POSDEF ? 3 : 2)
k_contrib(const NodeDesc* __restrict__ nodes, const TileTask* __restrict__ tasks,
          const double* __restrict__ L, const double* __restrict__ D, double* __restrict__ C) {
  __shared__ Stage<TS, CK> sg;
  const TileTask t = tasks[blockIdx.x];
  const NodeDesc nd = nodes[t.node];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cm = nd.m - nd.n;
  const double* Lb = L + nd.loff;
  const double* dinv = POSDEF ? nullptr : (D + 2 * int64_t(nd.sptr));
  const int ar0 = nd.n + t.ti * TS, br0 = nd.n + t.tj * TS;
  double* Cb = C + nd.coff;
  const int lr = lane & 15, lq = lane >> 4;
  const int wr = 32 * (wave >> 1), wc = 32 * (wave & 1);

  // tiles are produced transposed: lanes run along the rows of C (column-major), so the
  // read-modify-write of the contribution block moves 128-byte segments; its loads go first
  double cv[2][2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = t.ti * TS + wr + 16 * i + lr;
        const int col = t.tj * TS + wc + 16 * j + lq + 4 * r;
        cv[i][j][r] = (row < cm && col < cm && row >= col) ? Cb[int64_t(col) * cm + row] : 0.0;
      }
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = double4_t{0.0, 0.0, 0.0, 0.0};
  {
    StageRegs<TS, CK> rg;
    stage_load<TS, CK>(rg, Lb, nd.ld, ar0, nd.m, br0, nd.m, 0, nd.n, dinv, tid);
    for (int k0 = 0; k0 < nd.n; k0 += CK) {
      __syncthreads();
      stage_store<TS, CK>(sg, rg, tid);
      __syncthreads();
      if (k0 + CK < nd.n) stage_load<TS, CK>(rg, Lb, nd.ld, ar0, nd.m, br0, nd.m, k0 + CK, nd.n, dinv, tid);
      mfma_panel<TS, 2, 2, CK, true>(sg, wr, wc, lane, acc);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = t.ti * TS + wr + 16 * i + lr;
        const int col = t.tj * TS + wc + 16 * j + lq + 4 * r;
        if (row < cm && col < cm && row >= col) Cb[int64_t(col) * cm + row] = cv[i][j][r] - acc[i][j][r];
      }
}

// Tiny contribution blocks ((m-n) <= 16, n <= 64): one WAVE per front, two fronts per workgroup.  A
// 64 x 64 MFMA tile per such front wasted 15/16 of its work; here L21 and L21*D sit in LDS and every
// lane owns up to four entries of C.
struct TinyContribTask {
  int32_t n, cm, ld, sptr;
  int64_t loff, coff;
};
template <bool POSDEF>
__global__ void __launch_bounds__(128)
k_contrib_tiny(const TinyContribTask* __restrict__ tasks, int ntask, const double* __restrict__ L,
               const double* __restrict__ D, double* __restrict__ C) {
  __shared__ double Ls[2][16 * 65], LDs[2][16 * 65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = blockIdx.x * 2 + wave;
  if (ti >= ntask) return;
  const TinyContribTask t = tasks[ti];
  const int n = t.n, cm = t.cm;
  const double* Lb = L + t.loff;
  const double* dinv = D + 2 * int64_t(t.sptr);
  double* ls = Ls[wave];
  double* lds = LDs[wave];
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int u = 0; u < 16; ++u) {            // row i of L21, columns kq, kq+4, ...
    const int k = kq + 4 * u;
    const bool ok = (i < cm && k < n);
    const double v = ok ? Lb[int64_t(k) * t.ld + n + i] : 0.0;
    double ld = v;
    if (!POSDEF && ok) {                    // (L D)(i,k): D held inverted, calc_ld.hxx:43-118
      const double d0 = dinv[2 * k], d1 = dinv[2 * k + 1];
      if (isinf(d0)) {                      // second column of a 2x2 pivot
        const double d11 = dinv[2 * k - 2], d21 = dinv[2 * k - 1], d22 = d1;
        const double a1 = Lb[int64_t(k - 1) * t.ld + n + i];
        ld = (-d21 * a1 + d11 * v) / (d11 * d22 - d21 * d21);
      } else if (isinf(dinv[2 * k + 2])) {  // first column
        const double d22 = dinv[2 * k + 3];
        const double a2 = Lb[int64_t(k + 1) * t.ld + n + i];
        ld = (d22 * v - d1 * a2) / (d0 * d22 - d1 * d1);
      } else {
        ld = (d0 != 0.0) ? v / d0 : 0.0;
      }
    }
    ls[i * 65 + k] = v;
    lds[i * 65 + k] = ld;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  double* Cb = C + t.coff;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = kq + 4 * u;               // entry (i, j), i >= j
    if (i < cm && j <= i) {
      double acc = 0.0;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) acc = fma(ls[i * 65 + k], lds[j * 65 + k], acc);
      Cb[int64_t(j) * cm + i] -= acc;
    }
  }
}

// Several right-hand sides through ONE launch of a solve kernel: the grid is R times as long and every task is done
// once per column (a sweep that is bound by latency gets R times the waves in flight).  Workgroup b runs on XCD b % 8:
// the R columns of a task are given to workgroups of the SAME XCD, eight dispatch slots apart (a round of 8 R
// workgroups = 8 tasks x R columns), so that L comes from HBM once and from that XCD's L2 R - 1 times.  The columns'
// work vectors lie at fixed strides; R = 1: one column, strides unused.
struct Cols {
  int R, xcd;                            // xcd = 0: plain mapping (column b % R of task b / R)
  int64_t sx, sxs, scv, syb, spt, sio;   // xp, slot vector, contribution vectors, ybuf, part, the caller's columns
};
__device__ __forceinline__ void cols_map(const Cols& cs, int& col, unsigned& bid) {
  const unsigned b = blockIdx.x, R = unsigned(cs.R);
  if (cs.R <= 1) { col = 0; bid = b; return; }
  if (!cs.xcd) { col = int(b % R); bid = b / R; return; }
  const unsigned nb = gridDim.x / R, full = nb & ~7u, base = full * R;
  if (b < base) {
    const unsigned q = b / (8 * R), r = b % (8 * R);
    col = int(r >> 3);
    bid = q * 8 + (r & 7);
  } else {                               // the last, incomplete round
    const unsigned rem = nb - full, r = b - base;
    col = int(r / rem);
    bid = full + r % rem;
  }
}
#define GSLS_COLS      \
  int col_;            \
  unsigned bid;        \
  cols_map(cs, col_, bid)

// permutation kernels for R columns at once: grid (blocks, R)
__global__ void k_permute_in_cols(int n, const int32_t* __restrict__ invp, const double* __restrict__ x, int64_t ldx,
                                  const double* __restrict__ scale, double* __restrict__ xp, int64_t sx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = invp[i];
  x += blockIdx.y * ldx;
  xp[blockIdx.y * sx + i] = scale ? x[v] * scale[v] : x[v];
}
__global__ void k_permute_out_cols(int n, const int32_t* __restrict__ invp, const double* __restrict__ xp, int64_t sx,
                                   const double* __restrict__ scale, double* __restrict__ x, int64_t ldx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = invp[i];
  xp += blockIdx.y * sx;
  x[blockIdx.y * ldx + v] = scale ? xp[i] * scale[v] : xp[i];
}

// =================================================================================================
// Solve kernels (v0: one workgroup per front, level by level)
// =================================================================================================
__global__ void k_permute_in(int n, const int32_t* __restrict__ invp, const double* __restrict__ x,
                             const double* __restrict__ scale, double* __restrict__ xp) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int v = invp[i];
    xp[i] = scale ? x[v] * scale[v] : x[v];
  }
}
__global__ void k_permute_out(int n, const int32_t* __restrict__ invp, const double* __restrict__ xp,
                              const double* __restrict__ scale, double* __restrict__ x) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int v = invp[i];
    x[v] = scale ? xp[i] * scale[v] : xp[i];
  }
}

// ---- 64 x 64 triangular solves by ONE wave, operands in registers -------------------------------
// The diagonal block is staged in LDS as blk[k][i] (= L11(i,k), leading dimension SB); lane i then
// owns row i (forward) or column i (backward) in registers and the 64 dependent steps are a
// v_readlane/shuffle + one FMA each: no memory access and no barrier inside the recurrence.
constexpr int SB = 65;

template <bool UNIT>
__device__ __forceinline__ double wave_trsv_fwd(const double* blk, int nb, int lane, double yv) {
  double row[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) row[k] = blk[k * SB + lane];
  const double rd = UNIT ? 1.0 : 1.0 / blk[lane * SB + lane];   // one division, off the recurrence
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    if (k < nb) {
      if (!UNIT && lane == k) yv *= rd;
      const double yk = readlane_f64(yv, k);
      if (lane > k) yv -= row[k] * yk;
    }
  }
  return yv;
}

template <bool UNIT>
__device__ __forceinline__ double wave_trsv_bwd(const double* blk, int nb, int lane, double yv) {
  double col[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) col[k] = blk[lane * SB + k];   // L11(k, lane), k >= lane meaningful
  const double rd = UNIT ? 1.0 : 1.0 / blk[lane * SB + lane];
#pragma unroll
  for (int k = 63; k >= 0; --k) {
    if (k < nb) {
      if (!UNIT && lane == k) yv *= rd;
      const double yk = readlane_f64(yv, k);
      if (lane < k) yv -= col[k] * yk;
    }
  }
  return yv;
}

__device__ __forceinline__ void stage_block(double* blk, const double* __restrict__ Lb, int ld, int b,
                                            int nb, int tid) {
  double v[16];   // loads first, LDS stores afterwards (see stage_panels)
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int e = tid + 256 * t;
    const int i = e & 63, k = e >> 6;
    v[t] = (i < nb && k < nb && i >= k) ? Lb[int64_t(b + k) * ld + b + i] : (i == k ? 1.0 : 0.0);
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int e = tid + 256 * t;
    blk[(e >> 6) * SB + (e & 63)] = v[t];
  }
}

// s = sum_k row[k*ld] * y[k], k < cnt <= 64, with the loads batched 16 deep
__device__ __forceinline__ double dot_strided(const double* __restrict__ row, int64_t ld,
                                              const double* y, int cnt) {
  double s = 0.0;
  for (int k0 = 0; k0 < cnt; k0 += 16) {
    double v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = (k0 + t < cnt) ? row[int64_t(k0 + t) * ld] : 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += v[t] * ((k0 + t < cnt) ? y[k0 + t] : 0.0);
  }
  return s;
}

// forward substitution on one front: gather children's contribution vectors, solve L11 y = rhs,
// leave my contribution vector cvec = (children pass-through) - L21 y.
// xp is indexed by analyse-time pivot position; gperm[sptr+i] is the position whose variable became
// the front's i-th pivot after numerical pivoting (identity for Cholesky).
template <bool POSDEF>
__global__ void __launch_bounds__(256)
k_solve_fwd(const NodeDesc* __restrict__ nodes, const int32_t* __restrict__ lvl,
            const int32_t* __restrict__ clist, const int32_t* __restrict__ cmap,
            const int32_t* __restrict__ gperm, const double* __restrict__ L,
            double* __restrict__ xp, double* __restrict__ cvec, Cols cs) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  GSLS_COLS;
  xp += col_ * cs.sx;
  cvec += col_ * cs.scv;
  const NodeDesc nd = nodes[lvl[bid]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = nd.n, cm = nd.m - nd.n;
  double* blk = sh;                 // 64 x SB
  double* yo = sh + 64 * SB;        // n, analyse order
  double* y = yo + n;               // n, pivot order
  const double* Lb = L + nd.loff;
  double* mine = cvec + nd.moff;
  STAMPN(8);
  for (int i = tid; i < n; i += 256) yo[i] = xp[nd.sptr + i];
  for (int i = tid; i < cm; i += 256) mine[i] = 0.0;
  __syncthreads();
  for (int ci = nd.cbeg; ci < nd.cend; ++ci) {
    const NodeDesc cn = nodes[clist[ci]];
    const int ccm = cn.m - cn.n;
    const int32_t* map = cmap + cn.moff;
    const double* cv = cvec + cn.moff;
    for (int i = tid; i < ccm; i += 256) {
      const int idx = map[i];
      if (idx < n) yo[idx] += cv[i];
      else mine[idx - n] += cv[i];
    }
    __syncthreads();
  }
  STAMPN(9);
  for (int i = tid; i < n; i += 256) y[i] = POSDEF ? yo[i] : yo[gperm[nd.sptr + i] - nd.sptr];
  // blocked forward substitution
  for (int b = 0; b < n; b += 64) {
    const int nb = min(64, n - b);
    __syncthreads();
    stage_block(blk, Lb, nd.ld, b, nb, tid);
    __syncthreads();
    if (wave == 0) {
      const double v = wave_trsv_fwd<!POSDEF>(blk, nb, lane, (lane < nb) ? y[b + lane] : 0.0);
      if (lane < nb) y[b + lane] = v;
    }
    __syncthreads();
    // y[b+64 ..) -= L[b+64.., b..b+nb) * y[b..b+nb)
    for (int i = b + 64 + tid; i < n; i += 256)
      y[i] -= dot_strided(Lb + int64_t(b) * nd.ld + i, nd.ld, y + b, nb);
  }
  STAMPN(10);
  for (int i = tid; i < n; i += 256) xp[POSDEF ? nd.sptr + i : gperm[nd.sptr + i]] = y[i];
  // cvec -= L21 * y : four threads per row split the columns (coalesced across rows)
  {
    const int q = tid >> 6;   // column quarter handled by this wave
    for (int i0 = 0; i0 < cm; i0 += 64) {
      const int i = i0 + lane;
      double s = 0.0;
      if (i < cm) {   // wave q takes the columns [q*nq, (q+1)*nq)
        const int nq = (n + 3) / 4, kb0 = q * nq, ke0 = min(n, kb0 + nq);
        for (int k0 = kb0; k0 < ke0; k0 += 64)
          s += dot_strided(Lb + int64_t(k0) * nd.ld + n + i, nd.ld, y + k0, min(64, ke0 - k0));
      }
      blk[q * 64 + lane] = s;
      __syncthreads();
      if (q == 0 && i < cm) mine[i] -= (blk[lane] + blk[64 + lane]) + (blk[128 + lane] + blk[192 + lane]);
      __syncthreads();
    }
  }
  STAMPN(11);
}

// x <- D^-1 x in pivot order; D holds inverted pivots, 2x2 blocks as [d11,d21,inf,d22]
// (ldlt_app.cxx:2550-2571 ldlt_app_solve_diag)
__global__ void k_solve_diag(int n, const double* __restrict__ D, const int32_t* __restrict__ gperm,
                             double* __restrict__ xp, Cols cs) {
  GSLS_COLS;
  xp += col_ * cs.sx;
  const int i = bid * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d0 = D[2 * int64_t(i)];
  if (isinf(d0)) return;                        // second of a 2x2: handled by its partner
  const int gi = gperm[i];
  if (i + 1 < n && isinf(D[2 * int64_t(i) + 2])) {
    const int gj = gperm[i + 1];
    const double d21 = D[2 * int64_t(i) + 1], d22 = D[2 * int64_t(i) + 3];
    const double x1 = xp[gi], x2 = xp[gj];
    xp[gi] = fma(d0, x1, d21 * x2);
    xp[gj] = fma(d21, x1, d22 * x2);
  } else {
    xp[gi] *= d0;
  }
}

// backward substitution on one front: y = L11^-T (x1 - L21^T x2)
template <bool POSDEF>
__global__ void __launch_bounds__(256)
k_solve_bwd(const NodeDesc* __restrict__ nodes, const int32_t* __restrict__ lvl,
            const int32_t* __restrict__ rlist, const int32_t* __restrict__ gperm,
            const double* __restrict__ L, double* __restrict__ xp, Cols cs) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  GSLS_COLS;
  xp += col_ * cs.sx;
  const NodeDesc nd = nodes[lvl[bid]];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = nd.n, cm = nd.m - nd.n;
  double* blk = sh;              // 64 x SB (+ 256 doubles of scratch)
  double* y = sh + 64 * SB + 256;  // n
  double* z = y + n;             // cm
  const double* Lb = L + nd.loff;
  const int32_t* rl = rlist + nd.roff + n;
  STAMPN(16);
  for (int i = tid; i < n; i += 256) y[i] = xp[POSDEF ? nd.sptr + i : gperm[nd.sptr + i]];
  for (int i = tid; i < cm; i += 256) z[i] = xp[rl[i]];
  __syncthreads();
  STAMPN(17);
  // y[k] -= sum_i L21[i,k] z[i]: 64 x 64 tiles of L21 go through LDS (coalesced along rows), thread
  // (k, q) then walks 16 rows of column k out of LDS (odd stride: conflict free) -- no cross-lane
  // reduction; the four row-quarters are combined through LDS
  {
    const int kc = tid & 63, q = tid >> 6;
    for (int k0 = 0; k0 < n; k0 += 64) {
      double s = 0.0;
      for (int i0 = 0; i0 < cm; i0 += 64) {
        __syncthreads();
        {
          double v[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int e = tid + 256 * t;
            const int i = e & 63, k = e >> 6;
            v[t] = (i0 + i < cm && k0 + k < n) ? Lb[int64_t(k0 + k) * nd.ld + n + i0 + i] : 0.0;
          }
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            const int e = tid + 256 * t;
            blk[(e >> 6) * SB + (e & 63)] = v[t];
          }
        }
        __syncthreads();
        const double* zz = z + i0 + 16 * q;
        const int lim = min(16, cm - i0 - 16 * q);
#pragma unroll 4
        for (int i = 0; i < lim; ++i) s += blk[kc * SB + 16 * q + i] * zz[i];
      }
      __syncthreads();
      blk[64 * SB + q * 64 + kc] = s;       // scratch behind the tile (blk has 64*SB + 256 doubles)
      __syncthreads();
      if (q == 0 && k0 + kc < n)
        y[k0 + kc] -= (blk[64 * SB + kc] + blk[64 * SB + 64 + kc]) + (blk[64 * SB + 128 + kc] + blk[64 * SB + 192 + kc]);
    }
  }
  STAMPN(18);
  // blocked back substitution, last block first
  for (int b = ((n - 1) / 64) * 64; b >= 0; b -= 64) {
    const int nb = min(64, n - b);
    __syncthreads();
    // y[b..b+nb) -= L[b+64.., b..b+nb)^T y[b+64..)
    for (int k = b + wave; k < b + nb; k += 4) {
      const double* col = Lb + int64_t(k) * nd.ld;
      double s = 0.0;
#pragma unroll 4
      for (int i = b + 64 + lane; i < n; i += 64) s += col[i] * y[i];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
      if (lane == 0) y[k] -= s;
    }
    stage_block(blk, Lb, nd.ld, b, nb, tid);
    __syncthreads();
    if (wave == 0) {
      const double v = wave_trsv_bwd<!POSDEF>(blk, nb, lane, (lane < nb) ? y[b + lane] : 0.0);
      if (lane < nb) y[b + lane] = v;
    }
  }
  STAMPN(19);
  for (int i = tid; i < n; i += 256) xp[POSDEF ? nd.sptr + i : gperm[nd.sptr + i]] = y[i];
  STAMPN(20);
}

// =================================================================================================
// Cholesky solves with the inverted diagonal blocks (W = L11^-T, one 64 x 64 block per 64 pivots,
// written by k_diag_fast): a block's triangular solve is a 64 x 64 matrix-vector product instead of
// 64 dependent steps, and every load of a phase is issued before its first use.
// One workgroup per front; r (LDS) holds the front's m entries: [0,n) pivots, [n,m) contribution rows.
// =================================================================================================
// Everything a solve workgroup needs to know about its front in ONE load (no list -> node -> child
// pointer chase at the head of every launch): the front, and its first two children's vectors.
struct SolveTask {
  int32_t m, n, ld, sptr;
  int64_t loff, roff, moff;
  int32_t iblk, cbeg, cend, ccm0, ccm1, pad;
  int64_t cmoff0, cmoff1;
  int64_t goff;      // first entry of the front's rows in the gather lists (gth_ptr)
};

// rows [i] x columns [b, b+nb) of a front, one row per thread, every load issued before any use
__device__ __forceinline__ void load_row64(double (&l)[64], const double* __restrict__ Lb, int ld, int b,
                                           int nb, int i, int m) {
#pragma unroll
  for (int k = 0; k < 64; ++k) l[k] = (i < m && k < nb) ? Lb[int64_t(b + k) * ld + i] : 0.0;
}

__global__ void __launch_bounds__(256)
k_solve_fwd_chol(const NodeDesc* __restrict__ nodes, const SolveTask* __restrict__ tasks,
                 const int32_t* __restrict__ clist, const int32_t* __restrict__ cmap,
                 const double* __restrict__ L, const double* __restrict__ Linv,
                 double* __restrict__ xp, double* __restrict__ cvec, Cols cs) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  GSLS_COLS;
  xp += col_ * cs.sx;
  cvec += col_ * cs.scv;
  const SolveTask nd = tasks[bid];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int n = nd.n, m = nd.m, cm = m - n;
  double* r = sh;                         // m
  double* part = sh + ((m + 63) & ~63);   // 4 x 64
  const double* Lb = L + nd.loff;
  const double* Wn = Linv + int64_t(nd.iblk) * (NB * NB);
  STAMPN(8);
  // ---- everything that depends on nothing goes out first: W of block 0, the first 256 rows below
  // blocks 0 and 1, the right-hand side, the first two children's contribution vectors
  // X[row][k] = W[k + 64 row]: thread (row = lane, q) owns k in [16q, 16q+16)
  double w[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) w[t] = Wn[64 * lane + 16 * q + t];
  const int nb0 = min(64, n);
  double la[64];
  load_row64(la, Lb, nd.ld, 0, nb0, nb0 + tid, m);
  const int nchild = nd.cend - nd.cbeg;
  const int ccm2[2] = {nd.ccm0, nd.ccm1};
  const int64_t moff2[2] = {nd.cmoff0, nd.cmoff1};
  int mi2[2] = {0, 0};
  double cv2[2] = {0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c < nchild && tid < ccm2[c]) {
      mi2[c] = cmap[moff2[c] + tid];
      cv2[c] = cvec[moff2[c] + tid];
    }
  for (int i = tid; i < n; i += 256) r[i] = xp[nd.sptr + i];
  for (int i = n + tid; i < m; i += 256) r[i] = 0.0;
  __syncthreads();
  STAMPN(23);
  // children's contribution vectors, one child after the other (fixed summation order)
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (c < nchild) {
      if (tid < ccm2[c]) r[mi2[c]] += cv2[c];
      for (int i = tid + 256; i < ccm2[c]; i += 256) r[cmap[moff2[c] + i]] += cvec[moff2[c] + i];
      __syncthreads();
    }
  for (int ci = nd.cbeg + 2; ci < nd.cend; ++ci) {
    const NodeDesc cn = nodes[clist[ci]];
    const int ccm = cn.m - cn.n;
    const int32_t* map = cmap + cn.moff;
    const double* cv = cvec + cn.moff;
    for (int i = tid; i < ccm; i += 256) r[map[i]] += cv[i];
    __syncthreads();
  }
  STAMPN(9);
  auto apply = [&](const double (&l)[64], int b, int nb, int i) {
    double sacc = 0.0;
#pragma unroll
    for (int k = 0; k < 64; ++k) sacc += l[k] * r[b + (k < nb ? k : 0)];
    if (i < m) r[i] -= sacc;
  };
  for (int b = 0; b < n; b += 64) {
    const int nb = min(64, n - b);
    const int below = b + nb;
    // y_b = X_b r_b
    {
      double sacc = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) sacc += w[t] * ((16 * q + t < nb) ? r[b + 16 * q + t] : 0.0);
      part[q * 64 + lane] = sacc;
    }
    __syncthreads();
    STAMPN(24 + 4 * (b >> 6));
    if (b + 64 < n) {   // next block's W
      const double* W2 = Wn + int64_t((b >> 6) + 1) * (NB * NB);
#pragma unroll
      for (int t = 0; t < 16; ++t) w[t] = W2[64 * lane + 16 * q + t];
    }
    if (tid < nb) r[b + tid] = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
    __syncthreads();
    STAMPN(25 + 4 * (b >> 6));
    // r[below ..) -= L[below.., b..b+nb) y_b : one row per thread, 256 rows per pass
    for (int c0 = below; c0 < m; c0 += 256) {
      const int i = c0 + tid;
      if (c0 > below) load_row64(la, Lb, nd.ld, b, nb, i, m);
      apply(la, b, nb, i);
    }
    __syncthreads();
    STAMPN(26 + 4 * (b >> 6));
    // first 256 rows below the next block: in flight while that block is solved
    if (b + 64 < n) load_row64(la, Lb, nd.ld, b + 64, min(64, n - b - 64), b + 64 + min(64, n - b - 64) + tid, m);
  }
  STAMPN(10);
  for (int i = tid; i < n; i += 256) xp[nd.sptr + i] = r[i];
  double* mine = cvec + nd.moff;
  for (int i = tid; i < cm; i += 256) mine[i] = r[n + i];
  STAMPN(11);
}

__global__ void __launch_bounds__(256)
k_solve_bwd_chol(const SolveTask* __restrict__ tasks, const int32_t* __restrict__ rlist,
                 const double* __restrict__ L, const double* __restrict__ Linv, double* __restrict__ xp, Cols cs) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  GSLS_COLS;
  xp += col_ * cs.sx;
  const SolveTask nd = tasks[bid];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int n = nd.n, m = nd.m;
  double* blk = sh;                        // 64 x SB tile of L, transposed access
  double* part = sh + 64 * SB;             // 4 x 64
  double* z = part + 256;                  // 64
  double* r = z + 64;                      // m: [0,n) this front's pivots, [n,m) the ancestors' solution
  const double* Lb = L + nd.loff;
  const double* Wn = Linv + int64_t(nd.iblk) * (NB * NB);
  const int32_t* rl = rlist + nd.roff;
  STAMPN(16);
  for (int i = tid; i < n; i += 256) r[i] = xp[nd.sptr + i];
  for (int i = n + tid; i < m; i += 256) r[i] = xp[rl[i]];
  for (int b = ((n - 1) >> 6) << 6; b >= 0; b -= 64) {
    const int nb = min(64, n - b);
    const int below = b + nb;
    // (W z)[k] = sum_row W[k + 64 row] z[row]: thread (k = lane, q) owns rows [16q, 16q+16)
    double w[16];
    const double* Wb = Wn + int64_t(b >> 6) * (NB * NB);
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = Wb[lane + 64 * (16 * q + t)];
    // s[k] = sum_{i >= below} L[i, b+k] r[i]: 64-row tiles through LDS (coalesced along rows), thread
    // (k, q) then walks 16 rows of column k; the next tile's loads are in flight meanwhile
    double sacc = 0.0;
    double v[16];
    auto load_tile = [&](int i0) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int e = tid + 256 * t;
        const int i = e & 63, k = e >> 6;
        v[t] = (i0 + i < m && k < nb) ? Lb[int64_t(b + k) * nd.ld + i0 + i] : 0.0;
      }
    };
    if (below < m) load_tile(below);
    __syncthreads();                       // r complete (first pass) / previous block's r update visible
    for (int i0 = below; i0 < m; i0 += 64) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int e = tid + 256 * t;
        blk[(e >> 6) * SB + (e & 63)] = v[t];
      }
      __syncthreads();
      if (i0 + 64 < m) load_tile(i0 + 64);
      const int lim = min(16, m - i0 - 16 * q);
#pragma unroll 4
      for (int i = 0; i < lim; ++i) sacc += blk[lane * SB + 16 * q + i] * r[i0 + 16 * q + i];
      __syncthreads();
    }
    part[q * 64 + lane] = sacc;
    __syncthreads();
    if (tid < 64) z[tid] = (tid < nb) ? r[b + tid] - ((part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid])) : 0.0;
    __syncthreads();
    {
      double xacc = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) xacc += w[t] * z[16 * q + t];
      part[q * 64 + lane] = xacc;
    }
    __syncthreads();
    if (tid < nb) r[b + tid] = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
    // the next iteration's first barrier orders this write before any read of r
  }
  __syncthreads();
  STAMPN(19);
  for (int i = tid; i < n; i += 256) xp[nd.sptr + i] = r[i];
  STAMPN(20);
}

// =================================================================================================
// The same two Cholesky solve kernels for R right-hand sides at once (ssids_solve_mult, ssids.f90:1139-1249; the
// reference's kernel takes nrhs columns through trsm/gemm, cholesky.cxx:191-212): L and W are read ONCE for all R
// columns.  The front's vector becomes an m x R panel in LDS, r[i * R + c] (the R values of a row are contiguous: one
// broadcast read feeds R FMAs); xp and cvec hold R copies, `xs` and `cs` elements apart.  Column c of the result is
// bit-identical to the single-column kernels' (same operations in the same order per column).
// =================================================================================================
constexpr size_t MR_LDS_CAP = 150 * 1024;     // dynamic LDS the multi-column kernels may ask for
template <int R>
__global__ void __launch_bounds__(256)
k_solve_fwd_chol_mr(const NodeDesc* __restrict__ nodes, const SolveTask* __restrict__ tasks,
                    const int32_t* __restrict__ clist, const int32_t* __restrict__ cmap,
                    const double* __restrict__ L, const double* __restrict__ Linv,
                    double* __restrict__ xp, double* __restrict__ cvec, int64_t xs, int64_t cs) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const SolveTask nd = tasks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int n = nd.n, m = nd.m, cm = m - n;
  double* r = sh;                               // m x R
  double* part = sh + ((m + 63) & ~63) * R;     // 4 x 64 x R
  const double* Lb = L + nd.loff;
  const double* Wn = Linv + int64_t(nd.iblk) * (NB * NB);
  double w[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) w[t] = Wn[64 * lane + 16 * q + t];
  const int nb0 = min(64, n);
  double la[64];
  load_row64(la, Lb, nd.ld, 0, nb0, nb0 + tid, m);
  for (int i = tid; i < n; i += 256)
#pragma unroll
    for (int c = 0; c < R; ++c) r[i * R + c] = xp[c * xs + nd.sptr + i];
  for (int i = n + tid; i < m; i += 256)
#pragma unroll
    for (int c = 0; c < R; ++c) r[i * R + c] = 0.0;
  __syncthreads();
  for (int ci = nd.cbeg; ci < nd.cend; ++ci) {    // children's contribution vectors, one child after the other
    const NodeDesc cn = nodes[clist[ci]];
    const int ccm = cn.m - cn.n;
    const int32_t* map = cmap + cn.moff;
    const double* cv = cvec + cn.moff;
    for (int i = tid; i < ccm; i += 256) {
      const int mi = map[i];
#pragma unroll
      for (int c = 0; c < R; ++c) r[mi * R + c] += cv[c * cs + i];
    }
    __syncthreads();
  }
  auto apply = [&](const double (&l)[64], int b, int nb, int i) {
    double sacc[R];
#pragma unroll
    for (int c = 0; c < R; ++c) sacc[c] = 0.0;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      const double* rr = r + (b + (k < nb ? k : 0)) * R;
#pragma unroll
      for (int c = 0; c < R; ++c) sacc[c] += l[k] * rr[c];
    }
    if (i < m)
#pragma unroll
      for (int c = 0; c < R; ++c) r[i * R + c] -= sacc[c];
  };
  for (int b = 0; b < n; b += 64) {
    const int nb = min(64, n - b);
    const int below = b + nb;
    {
      double sacc[R];
#pragma unroll
      for (int c = 0; c < R; ++c) sacc[c] = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool on = 16 * q + t < nb;
        const double* rr = r + (b + (on ? 16 * q + t : 0)) * R;
#pragma unroll
        for (int c = 0; c < R; ++c) sacc[c] += w[t] * (on ? rr[c] : 0.0);
      }
#pragma unroll
      for (int c = 0; c < R; ++c) part[(q * 64 + lane) * R + c] = sacc[c];
    }
    __syncthreads();
    if (b + 64 < n) {   // next block's W
      const double* W2 = Wn + int64_t((b >> 6) + 1) * (NB * NB);
#pragma unroll
      for (int t = 0; t < 16; ++t) w[t] = W2[64 * lane + 16 * q + t];
    }
    if (tid < nb)
#pragma unroll
      for (int c = 0; c < R; ++c)
        r[(b + tid) * R + c] = (part[tid * R + c] + part[(64 + tid) * R + c]) +
                               (part[(128 + tid) * R + c] + part[(192 + tid) * R + c]);
    __syncthreads();
    for (int c0 = below; c0 < m; c0 += 256) {
      const int i = c0 + tid;
      if (c0 > below) load_row64(la, Lb, nd.ld, b, nb, i, m);
      apply(la, b, nb, i);
    }
    __syncthreads();
    if (b + 64 < n) load_row64(la, Lb, nd.ld, b + 64, min(64, n - b - 64), b + 64 + min(64, n - b - 64) + tid, m);
  }
  for (int i = tid; i < n; i += 256)
#pragma unroll
    for (int c = 0; c < R; ++c) xp[c * xs + nd.sptr + i] = r[i * R + c];
  double* mine = cvec + nd.moff;
  for (int i = tid; i < cm; i += 256)
#pragma unroll
    for (int c = 0; c < R; ++c) mine[c * cs + i] = r[(n + i) * R + c];
}

template <int R>
__global__ void __launch_bounds__(256)
k_solve_bwd_chol_mr(const SolveTask* __restrict__ tasks, const int32_t* __restrict__ rlist,
                    const double* __restrict__ L, const double* __restrict__ Linv, double* __restrict__ xp,
                    int64_t xs) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const SolveTask nd = tasks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
  const int n = nd.n, m = nd.m;
  double* blk = sh;                        // 64 x SB tile of L, transposed access
  double* part = sh + 64 * SB;             // 4 x 64 x R
  double* z = part + 256 * R;              // 64 x R
  double* r = z + 64 * R;                  // m x R
  const double* Lb = L + nd.loff;
  const double* Wn = Linv + int64_t(nd.iblk) * (NB * NB);
  const int32_t* rl = rlist + nd.roff;
  for (int i = tid; i < n; i += 256)
#pragma unroll
    for (int c = 0; c < R; ++c) r[i * R + c] = xp[c * xs + nd.sptr + i];
  for (int i = n + tid; i < m; i += 256) {
    const int src = rl[i];
#pragma unroll
    for (int c = 0; c < R; ++c) r[i * R + c] = xp[c * xs + src];
  }
  for (int b = ((n - 1) >> 6) << 6; b >= 0; b -= 64) {
    const int nb = min(64, n - b);
    const int below = b + nb;
    double w[16];
    const double* Wb = Wn + int64_t(b >> 6) * (NB * NB);
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = Wb[lane + 64 * (16 * q + t)];
    double sacc[R];
#pragma unroll
    for (int c = 0; c < R; ++c) sacc[c] = 0.0;
    double v[16];
    auto load_tile = [&](int i0) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int e = tid + 256 * t;
        const int i = e & 63, k = e >> 6;
        v[t] = (i0 + i < m && k < nb) ? Lb[int64_t(b + k) * nd.ld + i0 + i] : 0.0;
      }
    };
    if (below < m) load_tile(below);
    __syncthreads();
    for (int i0 = below; i0 < m; i0 += 64) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int e = tid + 256 * t;
        blk[(e >> 6) * SB + (e & 63)] = v[t];
      }
      __syncthreads();
      if (i0 + 64 < m) load_tile(i0 + 64);
      const int lim = min(16, m - i0 - 16 * q);
#pragma unroll 4
      for (int i = 0; i < lim; ++i) {
        const double lv = blk[lane * SB + 16 * q + i];
        const double* rr = r + (i0 + 16 * q + i) * R;
#pragma unroll
        for (int c = 0; c < R; ++c) sacc[c] += lv * rr[c];
      }
      __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < R; ++c) part[(q * 64 + lane) * R + c] = sacc[c];
    __syncthreads();
    if (tid < 64)
#pragma unroll
      for (int c = 0; c < R; ++c)
        z[tid * R + c] = (tid < nb) ? r[(b + tid) * R + c] - ((part[tid * R + c] + part[(64 + tid) * R + c]) +
                                                              (part[(128 + tid) * R + c] + part[(192 + tid) * R + c]))
                                    : 0.0;
    __syncthreads();
    {
      double xacc[R];
#pragma unroll
      for (int c = 0; c < R; ++c) xacc[c] = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const double* zz = z + (16 * q + t) * R;
#pragma unroll
        for (int c = 0; c < R; ++c) xacc[c] += w[t] * zz[c];
      }
#pragma unroll
      for (int c = 0; c < R; ++c) part[(q * 64 + lane) * R + c] = xacc[c];
    }
    __syncthreads();
    if (tid < nb)
#pragma unroll
      for (int c = 0; c < R; ++c)
        r[(b + tid) * R + c] = (part[tid * R + c] + part[(64 + tid) * R + c]) +
                               (part[(128 + tid) * R + c] + part[(192 + tid) * R + c]);
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256)
#pragma unroll
    for (int c = 0; c < R; ++c) xp[c * xs + nd.sptr + i] = r[i * R + c];
}

// =================================================================================================
// LDL^T solves for TINY fronts (n <= 64 pivots, m - n <= 64 rows below): one WAVE per front, four fronts per
// workgroup, no barriers.  Trees of saddle-point systems are tens of thousands of such fronts; a
// workgroup per front left the CU mostly idle.  Row/column of the front live in registers, the
// 64-step recurrences run on v_readlane broadcasts, rows are read coalesced (forward) or as one
// contiguous column per lane (backward).
// =================================================================================================
constexpr int TINY_M = 128;

template <int NMAX>   // NMAX = 32: fronts with at most 32 pivots (half the registers and instructions)
__global__ void __launch_bounds__(256)
k_solve_fwd_tiny(const SolveTask* __restrict__ tasks, int ntask, const int32_t* __restrict__ gth_ptr,
                 const int64_t* __restrict__ gth_src, const int32_t* __restrict__ gperm,
                 const double* __restrict__ L, double* __restrict__ xp, double* __restrict__ cvec, Cols cs) {
  __shared__ double rsh[4][TINY_M];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GSLS_COLS;
  xp += col_ * cs.sx;
  cvec += col_ * cs.scv;
  const int ti = bid * 4 + wave;
  if (ti >= ntask) return;
  const SolveTask nd = tasks[ti];
  const int n = nd.n, m = nd.m, cm = m - n;
  double* r = rsh[wave];
  const double* Lb = L + nd.loff;
  // row `lane` of the unit lower L11 and row n+lane of L21: coalesced along the lanes, all in flight
  double row[NMAX], low[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; ++k) row[k] = (lane < n && k < lane) ? Lb[int64_t(k) * nd.ld + lane] : 0.0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) low[k] = (lane < cm && k < n) ? Lb[int64_t(k) * nd.ld + n + lane] : 0.0;
  const int pslot = (lane < n) ? gperm[nd.sptr + lane] - nd.sptr : 0;   // pivot `lane` sits at this analyse position
  // right-hand side + the children's contribution vectors: every row of the front PULLS its sources in
  // child order (host-built gather lists: fixed summation order, and a front with hundreds of children
  // costs one round trip, not one per child)
  for (int i = lane; i < m; i += 64) {
    double csum = 0.0;                  // contributions first, then the right-hand side (the wave tier's order)
    const int g0 = gth_ptr[nd.goff + i], g1 = gth_ptr[nd.goff + i + 1];
    for (int g = g0; g < g1; ++g) csum += cvec[gth_src[g]];
    r[i] = ((i < n) ? xp[nd.sptr + i] : 0.0) + csum;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  double yv = (lane < n) ? r[pslot] : 0.0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {    // unit lower triangular solve, one pivot per step
    const double yk = readlane_f64(yv, k);
    yv = fma(-row[k], yk, yv);        // row[k] = 0 for k >= lane
  }
  if (lane < n) xp[nd.sptr + pslot] = yv;
  double acc = (lane < cm) ? r[n + lane] : 0.0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) acc = fma(-low[k], readlane_f64(yv, k), acc);
  if (lane < cm) cvec[nd.moff + lane] = acc;
}

template <int NMAX>
__global__ void __launch_bounds__(256)
k_solve_bwd_tiny(const SolveTask* __restrict__ tasks, int ntask, const int32_t* __restrict__ rlist,
                 const int32_t* __restrict__ gperm, const double* __restrict__ L, double* __restrict__ xp, Cols cs) {
  __shared__ double zsh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GSLS_COLS;
  xp += col_ * cs.sx;
  const int ti = bid * 4 + wave;
  if (ti >= ntask) return;
  const SolveTask nd = tasks[ti];
  const int n = nd.n, m = nd.m, cm = m - n;
  double* z = zsh[wave];
  const double* col = L + nd.loff + int64_t(lane) * nd.ld;   // column `lane`: one contiguous stream per lane
  double up[NMAX];
#pragma unroll
  for (int j = 0; j < NMAX; ++j) up[j] = (lane < n && j > lane && j < n) ? col[j] : 0.0;   // L11(j, lane)
  const int pslot = (lane < n) ? gperm[nd.sptr + lane] : 0;
  double xv = (lane < n) ? xp[pslot] : 0.0;
  if (lane < cm) z[lane] = xp[rlist[nd.roff + n + lane]];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (lane < n)       // rows below the pivots, last to first (the order of the wave tier: same bits)
    for (int i = cm - 1; i >= 0; --i) xv = fma(-col[n + i], z[i], xv);
#pragma unroll
  for (int j = NMAX - 1; j >= 0; --j) {   // unit upper triangular solve, last pivot first
    const double xj = readlane_f64(xv, j);
    xv = fma(-up[j], xj, xv);         // up[j] = 0 for j <= lane
  }
  if (lane < n) xp[pslot] = xv;
}

// =================================================================================================
// Solve path for BIG fronts (n > BIG_N or m > BIG_M): the front does not fit one workgroup's LDS and
// one CU cannot stream it fast enough, so every 64-column block becomes two launches -- a 64 x 64
// triangular solve (one workgroup per front) and a GEMV over the rows below it split into 256-row
// chunks (many workgroups per front).  y lives in HBM (ybuf, indexed by pivot slot).  Partial sums
// of the transposed GEMV are combined in a fixed order (no atomics).
// =================================================================================================
struct BigTrsv {
  int32_t node, part_first, part_cnt, pad;
};
struct BigGemv {
  int32_t node, row0;
};

template <bool POSDEF>
__global__ void __launch_bounds__(256)
k_big_fwd_prep(const NodeDesc* __restrict__ nodes, const int32_t* __restrict__ list,
               const int32_t* __restrict__ clist, const int32_t* __restrict__ cmap,
               const int32_t* __restrict__ gperm, double* __restrict__ xp, double* __restrict__ cvec,
               double* __restrict__ ybuf, Cols cs) {
  GSLS_COLS;
  xp += col_ * cs.sx;
  cvec += col_ * cs.scv;
  ybuf += col_ * cs.syb;
  const NodeDesc nd = nodes[list[bid]];
  const int tid = threadIdx.x;
  const int n = nd.n, cm = nd.m - nd.n;
  double* mine = cvec + nd.moff;
  for (int i = tid; i < cm; i += 256) mine[i] = 0.0;
  __syncthreads();
  for (int ci = nd.cbeg; ci < nd.cend; ++ci) {
    const NodeDesc cn = nodes[clist[ci]];
    const int ccm = cn.m - cn.n;
    const int32_t* map = cmap + cn.moff;
    const double* cv = cvec + cn.moff;
    for (int i = tid; i < ccm; i += 256) {
      const int idx = map[i];
      if (idx < n) xp[nd.sptr + idx] += cv[i];
      else mine[idx - n] += cv[i];
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += 256) ybuf[nd.sptr + i] = xp[POSDEF ? nd.sptr + i : gperm[nd.sptr + i]];
}

template <bool POSDEF>
__global__ void __launch_bounds__(256)
k_big_store(const NodeDesc* __restrict__ nodes, const int32_t* __restrict__ list,
            const int32_t* __restrict__ gperm, const double* __restrict__ ybuf,
            double* __restrict__ xp, int load, Cols cs) {
  GSLS_COLS;
  xp += col_ * cs.sx;
  ybuf += col_ * cs.syb;
  const NodeDesc nd = nodes[list[bid]];
  for (int i = threadIdx.x; i < nd.n; i += 256) {
    const int g = POSDEF ? nd.sptr + i : gperm[nd.sptr + i];
    if (load) const_cast<double*>(ybuf)[nd.sptr + i] = xp[g];
    else xp[g] = ybuf[nd.sptr + i];
  }
}

template <bool POSDEF>
__global__ void __launch_bounds__(256)
k_big_fwd_trsv(const NodeDesc* __restrict__ nodes, const BigTrsv* __restrict__ tasks, int b,
               const double* __restrict__ L, double* __restrict__ ybuf, Cols cs) {
  __shared__ double blk[64 * SB];
  GSLS_COLS;
  ybuf += col_ * cs.syb;
  const NodeDesc nd = nodes[tasks[bid].node];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nb = min(64, nd.n - b);
  stage_block(blk, L + nd.loff, nd.ld, b, nb, tid);
  __syncthreads();
  if (tid < 64) {
    const double v = wave_trsv_fwd<!POSDEF>(blk, nb, lane, (lane < nb) ? ybuf[nd.sptr + b + lane] : 0.0);
    if (lane < nb) ybuf[nd.sptr + b + lane] = v;
  }
}

__global__ void __launch_bounds__(256)
k_big_fwd_gemv(const NodeDesc* __restrict__ nodes, const BigGemv* __restrict__ tasks, int b,
               const double* __restrict__ L, double* __restrict__ ybuf, double* __restrict__ cvec, Cols cs) {
  __shared__ double ys[64];
  GSLS_COLS;
  ybuf += col_ * cs.syb;
  cvec += col_ * cs.scv;
  const BigGemv t = tasks[bid];
  const NodeDesc nd = nodes[t.node];
  const int tid = threadIdx.x;
  const int nb = min(64, nd.n - b);
  if (tid < 64) ys[tid] = (tid < nb) ? ybuf[nd.sptr + b + tid] : 0.0;
  __syncthreads();
  const int row = t.row0 + tid;
  if (row < nd.m) {
    const double sum = dot_strided(L + nd.loff + int64_t(b) * nd.ld + row, nd.ld, ys, nb);
    if (row < nd.n) ybuf[nd.sptr + row] -= sum;
    else cvec[nd.moff + row - nd.n] -= sum;
  }
}

__global__ void __launch_bounds__(256)
k_big_bwd_gemvT(const NodeDesc* __restrict__ nodes, const BigGemv* __restrict__ tasks, int b,
                const int32_t* __restrict__ rlist, const double* __restrict__ L,
                const double* __restrict__ ybuf, const double* __restrict__ xp,
                double* __restrict__ part, Cols cs) {
  __shared__ double blk[64 * SB + 256];
  __shared__ double zz[64];
  GSLS_COLS;
  ybuf += col_ * cs.syb;
  xp += col_ * cs.sx;
  part += col_ * cs.spt;
  const BigGemv t = tasks[bid];
  const NodeDesc nd = nodes[t.node];
  const int tid = threadIdx.x, kc = tid & 63, q = tid >> 6;
  const int nb = min(64, nd.n - b);
  const double* Lb = L + nd.loff;
  double s = 0.0;
  for (int i0 = t.row0; i0 < min(t.row0 + 256, nd.m); i0 += 64) {
    __syncthreads();
    {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u;
        const int i = e & 63, k = e >> 6;
        v[u] = (i0 + i < nd.m && k < nb) ? Lb[int64_t(b + k) * nd.ld + i0 + i] : 0.0;
      }
      if (tid < 64) {
        const int row = i0 + tid;
        zz[tid] = (row >= nd.m) ? 0.0 : (row < nd.n ? ybuf[nd.sptr + row] : xp[rlist[nd.roff + row]]);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int e = tid + 256 * u;
        blk[(e >> 6) * SB + (e & 63)] = v[u];
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) s += blk[kc * SB + 16 * q + i] * zz[16 * q + i];
  }
  __syncthreads();
  blk[64 * SB + q * 64 + kc] = s;
  __syncthreads();
  if (q == 0)
    part[int64_t(bid) * 64 + kc] =
        (blk[64 * SB + kc] + blk[64 * SB + 64 + kc]) + (blk[64 * SB + 128 + kc] + blk[64 * SB + 192 + kc]);
}

template <bool POSDEF>
__global__ void __launch_bounds__(256)
k_big_bwd_trsv(const NodeDesc* __restrict__ nodes, const BigTrsv* __restrict__ tasks, int b,
               const double* __restrict__ L, double* __restrict__ ybuf, const double* __restrict__ part, Cols cs) {
  __shared__ double blk[64 * SB];
  GSLS_COLS;
  ybuf += col_ * cs.syb;
  part += col_ * cs.spt;
  const BigTrsv t = tasks[bid];
  const NodeDesc nd = nodes[t.node];
  const int tid = threadIdx.x, lane = tid & 63;
  const int nb = min(64, nd.n - b);
  stage_block(blk, L + nd.loff, nd.ld, b, nb, tid);
  __syncthreads();
  if (tid < 64) {
    double yv = (lane < nb) ? ybuf[nd.sptr + b + lane] : 0.0;
    for (int c = 0; c < t.part_cnt; ++c) yv -= part[int64_t(t.part_first + c) * 64 + lane];   // fixed order
    const double v = wave_trsv_bwd<!POSDEF>(blk, nb, lane, (lane < nb) ? yv : 0.0);
    if (lane < nb) ybuf[nd.sptr + b + lane] = v;
  }
}

// =================================================================================================
// WAVE TIER of the LDL^T solves: fronts of at most 64 rows whose whole subtree consists of such fronts (in a
// saddle-point tree that is every front), one WAVE per front, lane r = row r of the front, pivot rows and
// contribution rows alike.
//   * the factor of such a front is kept in two packed images, each in exactly the order its sweep consumes it,
//     so every load is a 16-byte-per-lane segment of consecutive addresses and no byte that is not part of L is read:
//       forward  image (Lf): for column pair j = (2j, 2j+1): rows r = 2j+1 .. m-1, element (L(r,2j), L(r,2j+1))
//                            [the slot of L(2j+1,2j+1) holds 0]
//       backward image (Lb): for row pair i = (2i, 2i+1): columns k = 0 .. min(2i, n-1), element (L(2i,k), L(2i+1,k))
//                            [the slot of L(2i,2i) holds 0]
//     The rectangle m x n of the factorization kernels stores 1.4x the entries of L for the typical 39 x 24 front,
//     and walking it column by column in the backward sweep drags whole cache lines for 8 bytes each.
//   * the schedule is by SUBTREES, not by levels (the reference's CPU code has the same idea for the same reason:
//     SmallLeafNumericSubtree.hxx): a GROUP is a small subtree that ONE wave walks in postorder (forward) or reverse
//     postorder (backward).  Inside a group nothing goes through memory and nothing waits for another wave: a child's
//     contribution vector is added into its parent's accumulator in LDS (forward), a parent leaves its front's part
//     of the solution in LDS for its children (backward), through the static child-row -> parent-row map.  The
//     groups of one STAGE are independent (one launch); stage k+1 holds the subtrees of what remains of the tree
//     when the groups of stages <= k are removed, and reads their roots' contribution vectors from HBM.  A
//     saddle-point tree of 10 levels becomes 3-4 launches per sweep, and 97 % of the factor sits in stage 0.
//   * the right-hand side is read straight from the caller's vector (x[invp[.]]) and the solution written straight
//     back; D^-1 is applied at the end of the forward step: no separate permutation or diagonal launches.
// Per front the arithmetic is the tiny kernels': right-hand side + (sum of the children's contributions), the
// recurrences column by column.
// =================================================================================================
enum { WT_PULL = 1, WT_INT = 2, WT_PUSH = 4, WT_ZVEC = 8,
       WT_DENSE = 16 };   // (with WT_PULL) every row has at most two sources: goff indexes the dense table wpull2
struct WTask {
  int32_t m, n, sptr, flags;   // WT_PULL: children in earlier stages (gather lists); WT_INT: children in this group
                               // (LDS accumulator); WT_PUSH: the parent is in this group; WT_ZVEC: the parent is in a
                               // later stage of the tier (backward: it leaves its values in this front's cvec)
  int64_t lfoff, lboff;        // element offsets of the front's images in Lf / Lb
  int64_t roff, moff, goff;    // row list, contribution vector / child->parent map, gather lists
  int32_t myslot, pslot;       // LDS slots (depth inside the group) of this front and of its parent
};
static_assert(sizeof(WTask) == 64, "WTask is loaded as one 64-byte record");
struct WGroup {
  int32_t tbeg, tcnt;          // tasks [tbeg, tbeg + tcnt), postorder
};
struct WPack {                 // per task, for the pack kernel
  int64_t loff;
  int32_t node, pad;
};
constexpr int WSLOT = 8;       // a group is at most this deep

__device__ __forceinline__ int wf_pair_off(int j, int m) { return j * (m - 1) - j * (j - 1); }       // 16-byte units
__device__ __forceinline__ int wb_pair_off(int i, int n) {
  const int q = n >> 1;
  return (i <= q) ? i * i : q * q + (i - q) * n;
}
static inline int64_t wf_size(int m, int n) { const int j = (n + 1) / 2; return 2 * (int64_t(j) * (m - 1) - int64_t(j) * (j - 1)); }   // doubles
static inline int64_t wb_size(int m, int n) {
  const int i = (m + 1) / 2, q = n / 2;
  return 2 * ((i <= q) ? int64_t(i) * i : int64_t(q) * q + int64_t(i - q) * n);
}

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));

// D^-1 of pivot slot s applied to the forward result y (lane = slot - sptr); yp / yn: the neighbouring lanes' y.
// Same expressions as k_solve_diag.
__device__ __forceinline__ double wave_apply_d(const double* __restrict__ D, int64_t s, double y, double yp, double yn) {
  const double d0 = D[2 * s], d1 = D[2 * s + 1];
  if (isinf(d0)) return fma(D[2 * s - 1], yp, d1 * y);               // second row of a 2x2: d21 * y1 + d22 * y2
  if (isinf(D[2 * s + 2])) return fma(d0, y, d1 * yn);               // first row:           d11 * y1 + d21 * y2
  return y * d0;
}

// Everything a front's step needs that does not depend on other fronts of its group is loaded ONE FRONT AHEAD (while
// the wave works on the previous front), so that what is left on the wave's critical path is the LDS hand-off and
// the recurrence.  For that to work every such load must be ONE hop (address from the task record and the lane
// only): vector-memory results return in order, so a dependent load inside the look-ahead would make the wave wait
// for everything issued before it, the look-ahead included.  Hence
//   * forward:  the right-hand side comes permuted (xp, by position); the pivots' D^-1 entries ride along;
//   * forward -> backward: the forward result is kept by PIVOT SLOT (xs), which is how the backward step wants it;
//   * backward: a front with children in an earlier stage scatters its part of the solution to THEIR
//     contribution vectors (the forward gather lists read backwards), so a group root finds its ancestors' values
//     in cvec[moff + i], not behind rlist;
//   * backward: the variable index of every pivot slot (gvar = invp o gperm) is refreshed by each factorization.
// The two-hop forms stay for the part solves (slotv == nullptr) and for roots below a front outside the tier.
template <int NN>
struct WFwdPre {
  double2_t lp[NN / 2];
  double rhs, d0, d1, dn, dp;
  int pslot, prow;
};

// (The wide forward launches -- upper stages, one front per wave, latency-bound -- keep this form: the flat stream
//  through LDS that pays in the bottom stage was measured here too and lost 3 us over stages 1-3, the extra LDS
//  round trip costing more than the partial loads at that occupancy.)
// Branch-free on purpose: every load is issued unconditionally from a clamped (always valid) address and masked
// afterwards.  hipcc's wait-count insertion falls back to s_waitcnt vmcnt(0) at control-flow joins, which would make
// the wave wait for the look-ahead loads at the first conditional in the compute step; straight-line code gets exact
// counted waits.  (The arrays indexed by sptr + lane / moff + lane carry 64 elements of padding.)
template <int NN, bool APPLY_D>
__device__ __forceinline__ void wave_fwd_load(const WTask& t, int lane, WFwdPre<NN>& p, const double* __restrict__ Lf,
                                              const double* __restrict__ D, const int32_t* __restrict__ gperm,
                                              const int32_t* __restrict__ cmap, const double* __restrict__ xp) {
  const int m = t.m, n = t.n;
  // the image through a buffer descriptor: lanes outside their range get an out-of-range offset and the hardware
  // returns zeros -- two vector instructions per column pair (compare, select) instead of a dozen for 64-bit
  // addresses and masks (these kernels are bound by instruction issue, not by HBM, until that is trimmed)
  const int npair = (n + 1) >> 1;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<double*>(Lf + t.lfoff), 0, 16 * (npair * (m - 1) - npair * (npair - 1)), 0x00020000);
  const int oob = int(0x80000000);
  const int v0 = (lane >= 1 && lane < m) ? (lane - 1) * 16 : oob;
#pragma unroll
  for (int j = 0; j < NN / 2; ++j) {
    const bool ok = (lane >= 2 * j + 1) & (2 * j < n);
    // element (row lane, pair j) sits at wf_pair_off(j, m) + lane - (2j + 1) = j (m - j - 2) + (lane - 1)
    p.lp[j] = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? v0 : oob, j * (m - j - 2) * 16, 0));
  }
  const int64_t s = int64_t(t.sptr) + lane;
  const bool piv = lane < n;
  const double r = xp[s];
  const int gp = gperm[s];
  const int pr = cmap[t.moff + max(lane - n, 0)];
  p.rhs = piv ? r : 0.0;
  p.pslot = piv ? gp - t.sptr : lane;
  p.prow = pr;
  p.d0 = p.d1 = p.dn = p.dp = 0.0;
  if (APPLY_D) {
    const double d0 = D[2 * s], d1 = D[2 * s + 1], dn = D[2 * s + 2], dp = D[2 * s + 3 - 4 * (s > 0)];
    p.d0 = d0;
    p.d1 = d1;
    p.dn = dn;
    p.dp = dp;                          // D[2s - 1] (unused at s = 0)
  }
}

// The NARROW kernels' form of the same load (round 3).  Measured on the bottom stage of the metric workload
// (tools/wsolve_ubench.hip, profiles/r03/ubench_wsolve.txt): one 16-byte load instruction per column pair costs about as
// much whether it brings 464 bytes or none (16 instructions per front: 54 us for 140 MB; 12: 48 us; the same bytes as 4
// full-wave pieces: 39 us), and the seven small loads beside the image -- issued for all 64 lanes, 4 x 1 KB of D alone --
// another 11 us.  Hence
//   * the image is read as a FLAT stream, ceil(bytes / 1024) loads of 16 bytes per lane over consecutive addresses,
//     and handed to the lanes that own the entries through LDS (the packed layout is unchanged: the writers are);
//   * the small loads are masked to the lanes that use them (buffer descriptors sized to the front: a lane out of
//     range costs no request), D^-1 is ONE 16-byte load per pivot, its neighbours' words come by lane shifts.
// A 2x2 pivot never straddles two fronts, so the word of the next front's first pivot that the unmasked form read
// for the last lane (never inf) can be any finite number: the masked load returns 0.
constexpr int WIMG_BYTES = 6144;      // LDS staging area per wave: the largest image of a narrow front (n <= 32, m <= 40)
constexpr int WACC_NARROW = 40;       // ... and the row length of its LDS accumulators (64 otherwise)
constexpr int WIMG_CHUNKS = WIMG_BYTES / 1024;
constexpr unsigned WBUF_FLAGS = 0x00020000;

template <class T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_rsrc(const T* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), 0, bytes, WBUF_FLAGS);
}
__device__ __forceinline__ double wave_ld_f64(__amdgpu_buffer_rsrc_t rs, int off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
}
template <int AUX = 0>
__device__ __forceinline__ double2_t wave_ld_f64x2(__amdgpu_buffer_rsrc_t rs, int off, int soff) {
  return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, AUX));
}
// Cache policy of the image loads (2 = nt).  The forward image is read once per solve and its 170 MB pass through the
// caches right after the factorization wrote both images; as nt loads they leave the small vectors (right-hand side, D,
// maps, contribution vectors) and the backward image where later kernels find them: bottom stage 46.7 -> 37.9 us,
// whole sweep 187 -> 171 us (profiles/r03/solve_experiments.txt).  On the backward image it makes no difference.
#ifndef GSLS_WS_NT_F
#define GSLS_WS_NT_F 2
#endif
#ifndef GSLS_WS_NT_B
#define GSLS_WS_NT_B 0
#endif

// issue: the flat pieces of the image and the masked small loads
template <bool APPLY_D>
struct WFwdFlat {
  double2_t ch[WIMG_CHUNKS];
  double2_t dd;
  double rhs;
  int gp, pr;
};
// GATHER: the right-hand side comes from the CALLER's vector (variable order) through invp -- its index is the first load
// issued, so that the dependent load can go out while the image is still arriving (results return in order: waiting for
// the oldest load waits for nothing else) -- instead of from the permuted copy xp: the k_permute_in launch disappears
// for the fronts of the bottom stage's narrow launch, which also permutes the remaining 3 % of the positions for the
// launches behind it (trailing workgroups, k_wsolve_fwd).
template <bool APPLY_D, bool GATHER = false>
__device__ __forceinline__ void wave_fwd_issue(const WTask& t, int lane, WFwdFlat<APPLY_D>& q, const double* __restrict__ Lf,
                                               const double* __restrict__ D, const int32_t* __restrict__ gperm,
                                               const int32_t* __restrict__ cmap, const double* __restrict__ xp,
                                               const int32_t* __restrict__ invp = nullptr,
                                               const double* __restrict__ xin = nullptr) {
  const int m = t.m, n = t.n;
  const int npair = (n + 1) >> 1;
  const int nbytes = 16 * (npair * (m - 1) - npair * (npair - 1));
  const int oob = int(0x80000000);
  int iv = 0;
  if (GATHER) iv = __builtin_amdgcn_raw_buffer_load_b32(wave_rsrc(invp + t.sptr, n * 4), lane * 4, 0, 0);
  const __amdgpu_buffer_rsrc_t rs = wave_rsrc(Lf + t.lfoff, nbytes);
#pragma unroll
  for (int c = 0; c < WIMG_CHUNKS; ++c) {
    q.ch[c] = double2_t{0.0, 0.0};
    if (c < 2 || c * 1024 < nbytes) q.ch[c] = wave_ld_f64x2<GSLS_WS_NT_F>(rs, lane * 16, c * 1024);   // (uniform)
  }
  if (!GATHER) q.rhs = wave_ld_f64(wave_rsrc(xp + t.sptr, n * 8), lane * 8);
  q.gp = __builtin_amdgcn_raw_buffer_load_b32(wave_rsrc(gperm + t.sptr, n * 4), lane * 4, 0, 0);
  q.pr = __builtin_amdgcn_raw_buffer_load_b32(wave_rsrc(cmap + t.moff, (m - n) * 4), lane >= n ? (lane - n) * 4 : oob, 0, 0);
  q.dd = double2_t{0.0, 0.0};
  if (APPLY_D) q.dd = wave_ld_f64x2(wave_rsrc(D + 2 * int64_t(t.sptr), n * 16), lane * 16, 0);
  if (GATHER) {
    const double v = xin[lane < n ? iv : 0];     // (iv = 0 from the descriptor for the lanes that are not pivots)
    q.rhs = lane < n ? v : 0.0;
  }
}
// arrival: through LDS to the lane = row, register = column pair form the recurrence wants
template <bool APPLY_D>
__device__ __forceinline__ void wave_fwd_arrive(const WTask& t, int lane, const WFwdFlat<APPLY_D>& q, WFwdPre<32>& p,
                                                double2_t* __restrict__ im) {
  const int m = t.m, n = t.n;
  const int npair = (n + 1) >> 1;
  const int nbytes = 16 * (npair * (m - 1) - npair * (npair - 1));
#pragma unroll
  for (int c = 0; c < WIMG_CHUNKS; ++c)
    if (c < 2 || c * 1024 < nbytes) im[c * 64 + lane] = q.ch[c];
  const bool row = (lane >= 1) & (lane < m);
#pragma unroll
  for (int j4 = 0; j4 < 16; j4 += 4) {
    if (j4 < 8 || 2 * j4 < n) {          // (uniform; the recurrence skips the same groups of columns)
#pragma unroll
      for (int j = j4; j < j4 + 4; ++j) {
        const bool ok = row & (lane >= 2 * j + 1) & (2 * j < n);
        const double2_t v = im[ok ? j * (m - j - 2) + lane - 1 : 0];
        p.lp[j] = ok ? v : double2_t{0.0, 0.0};
      }
    } else {
#pragma unroll
      for (int j = j4; j < j4 + 4; ++j) p.lp[j] = double2_t{0.0, 0.0};
    }
  }
  const bool piv = lane < n;
  p.rhs = q.rhs;                          // (0 from the descriptor for the lanes that are not pivots)
  p.pslot = piv ? q.gp - t.sptr : lane;
  p.prow = q.pr;
  p.d0 = q.dd.x;
  p.d1 = q.dd.y;
  p.dn = p.dp = 0.0;
  if (APPLY_D) {
    p.dn = __shfl_down(q.dd.x, 1);        // D[2s + 2]: the next pivot's first word (inf: this is the first row of a 2x2)
    p.dp = __shfl_up(q.dd.y, 1);          // D[2s - 1]: the previous pivot's second word (d21 of a 2x2 ending here)
  }
}

// slotv != nullptr: the result goes there by pivot slot (job ALL: only the backward wave kernels read it);
// otherwise to xp by position, like every other kernel's
// Pulls with at most two sources per row (WT_DENSE: every front of the metric workload's upper stages): the sources'
// indices sit in a dense table, one 8-byte pair per row at a place the task record gives, so the pair is ONE hop and
// is requested before the image (wave_pull2_issue, ahead of wave_fwd_load), the two contribution entries a second hop
// that overlaps the image's arrival -- against gth_ptr -> gth_src -> cvec, three dependent round trips AFTER it.
typedef int int2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int2_t wave_pull2_issue(const WTask& t, int lane, const int32_t* __restrict__ pull2) {
  int2_t r = {-1, -1};
  if ((t.flags & (WT_PULL | WT_DENSE)) == (WT_PULL | WT_DENSE)) {          // (uniform)
    r = __builtin_bit_cast(int2_t, __builtin_amdgcn_raw_buffer_load_b64(wave_rsrc(pull2 + 2 * t.goff, t.m * 8), lane * 8, 0, 0));
    if (lane >= t.m) r = int2_t{-1, -1};                                   // (the descriptor returned zeros there)
  }
  return r;
}

template <int NN, bool APPLY_D, bool PULLS, int AS = 64>     // AS: row length of the LDS accumulators (>= every m of the launch)
__device__ __forceinline__ void wave_fwd_compute(const WTask& t, int lane, const WFwdPre<NN>& p, double* __restrict__ acc,
                                                 const int32_t* __restrict__ gth_ptr, const int64_t* __restrict__ gth_src,
                                                 double* __restrict__ xp, double* __restrict__ slotv,
                                                 double* __restrict__ cvec, int2_t pull = int2_t{-1, -1}) {
  const int m = t.m, n = t.n;
  // analyse-time row `lane` of the front: right-hand side + (sum of the children's contributions)
  double csum = 0.0;
  if (PULLS && (t.flags & WT_PULL) && (t.flags & WT_DENSE)) {
    const double c0 = cvec[max(pull.x, 0)], c1 = cvec[max(pull.y, 0)];
    csum += (pull.x >= 0) ? c0 : 0.0;      // list order, as the general form below
    csum += (pull.y >= 0) ? c1 : 0.0;
  } else if (PULLS && (t.flags & WT_PULL)) {   // (wave-uniform: every lane walks the loop, rows without sources add nothing)
    const int lr = min(lane, m - 1);
    const int g0 = gth_ptr[t.goff + lr], g1 = (lane < m) ? gth_ptr[t.goff + lr + 1] : g0;
    const int last = max(g1 - 1, g0);   // (g0 is a valid index whenever the front has any source at all)
    // eight sources in flight, their indices fetched one round ahead; summed in list order
    int64_t idx[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) idx[q] = gth_src[min(g0 + q, last)];
    for (int g = g0; __any(g < g1); g += 8) {
      double c[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) c[q] = cvec[idx[q]];
#pragma unroll
      for (int q = 0; q < 8; ++q) idx[q] = gth_src[min(g + 8 + q, last)];
#pragma unroll
      for (int q = 0; q < 8; ++q) csum += (g + q < g1) ? c[q] : 0.0;
    }
  }
  if (t.flags & WT_INT) {
    double* mine = acc + t.myslot * AS;
    if (AS == 64 || lane < AS) {
      csum += mine[lane];
      mine[lane] = 0.0;                 // ready for the next front at this depth
    }
  }
  // pivot slot r holds the analyse-time row gperm[sptr + r] - sptr (numerical pivoting inside the front)
  double x = __shfl(p.rhs + csum, p.pslot);
#pragma unroll
  for (int k4 = 0; k4 < NN; k4 += 4) {  // the image holds zeros from column n on: whole groups of four are skipped
    if (k4 < 16 || k4 < n) {            // (uniform; no memory operation inside, so the waits stay counted)
#pragma unroll
      for (int k = k4; k < k4 + 4; ++k) {
        const double yk = readlane_f64(x, k);
        const double l = (k & 1) ? p.lp[k >> 1].y : p.lp[k >> 1].x;
        x = fma(-l, yk, x);             // l = 0 for the rows up to k and for the columns from n on
      }
    }
  }
  {
    // contribution rows: into the parent's accumulator (inside the group) or out to HBM; the other lanes add 0 to a
    // spare slot / store to the padding behind the vector -- no branch
    const bool crow = (lane >= n) & (lane < m);
    const bool push = (t.flags & WT_PUSH) != 0;
    double* a = acc + ((crow & push) ? t.pslot * AS + p.prow : WSLOT * AS + lane);
    *a += (crow & push) ? x : 0.0;
    if (!push && crow) cvec[t.moff + lane - n] = x;
  }
  if (APPLY_D) {                        // same expressions as k_solve_diag
    const double yp = __shfl_up(x, 1), yn = __shfl_down(x, 1);
    if (isinf(p.d0)) x = fma(p.dp, yp, p.d1 * x);              // second row of a 2x2: d21 * y1 + d22 * y2
    else if (isinf(p.dn)) x = fma(p.d0, x, p.d1 * yn);         // first row:           d11 * y1 + d21 * y2
    else x = x * p.d0;
  }
  if (lane < n) {
    if (slotv) slotv[t.sptr + lane] = x;
    else xp[t.sptr + p.pslot] = x;
  }
}

// wave_fwd_compute for CG columns at once (narrow launches: no pulls): the same operations per column in the same
// order -- every column's result carries the bits of a single-column solve -- with the CG recurrences interleaved, so that
// one column's v_readlane -> FMA dependency is covered by the others' (these launches run at two or three waves per SIMD:
// CG sets of LDS accumulators per wave).  Column c works in acc + c * ACCW, xp + c * sx, slotv + c * sxs, cvec + c * scv.
template <int NN, bool APPLY_D, int AS, int CG>
__device__ __forceinline__ void wave_fwd_compute_cg(const WTask& t, int lane, const WFwdPre<NN>& p, const double (&rhs)[CG],
                                                    double* __restrict__ acc, int accw, double* __restrict__ xp, int64_t sx,
                                                    double* __restrict__ slotv, int64_t sxs, double* __restrict__ cvec,
                                                    int64_t scv) {
  const int m = t.m, n = t.n;
  const int spare = accw - 64;          // the row for the masked lanes sits behind the launch's deepest slot
  double x[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    double csum = 0.0;
    if (t.flags & WT_INT) {
      double* mine = acc + c * accw + t.myslot * AS;
      if (AS == 64 || lane < AS) {
        csum += mine[lane];
        mine[lane] = 0.0;
      }
    }
    x[c] = __shfl(rhs[c] + csum, p.pslot);
  }
#pragma unroll
  for (int k4 = 0; k4 < NN; k4 += 4) {
    if (k4 < 16 || k4 < n) {
#pragma unroll
      for (int k = k4; k < k4 + 4; ++k) {
        const double l = (k & 1) ? p.lp[k >> 1].y : p.lp[k >> 1].x;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          const double yk = readlane_f64(x[c], k);
          x[c] = fma(-l, yk, x[c]);
        }
      }
    }
  }
  const bool crow = (lane >= n) & (lane < m);
  const bool push = (t.flags & WT_PUSH) != 0;
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    double* a = acc + c * accw + ((crow & push) ? t.pslot * AS + p.prow : spare + lane);
    *a += (crow & push) ? x[c] : 0.0;
    if (!push && crow) cvec[c * scv + t.moff + lane - n] = x[c];
    double y = x[c];
    if (APPLY_D) {
      const double yp = __shfl_up(y, 1), yn = __shfl_down(y, 1);
      if (isinf(p.d0)) y = fma(p.dp, yp, p.d1 * y);
      else if (isinf(p.dn)) y = fma(p.d0, y, p.d1 * yn);
      else y = y * p.d0;
    }
    if (lane < n) {
      if (slotv) slotv[c * sxs + t.sptr + lane] = y;
      else xp[c * sx + t.sptr + p.pslot] = y;
    }
  }
}

template <int MM>
struct WBwdPre {
  double2_t up[MM / 2];
  double x;
  int pos, var, prow;
};

// FAST: the one-hop forms only (job ALL on the tier's own data; see above), branch-free like wave_fwd_load
// ONE image serves both sweeps (round 3): the backward step reads the forward image Lf -- column pairs by rows, the layout
// the factorization kernels' registers have -- as the same flat stream and takes its TRANSPOSE out of LDS: lane k =
// column k reads (L(2i,k), L(2i+1,k)) for the row pairs i, two 8-byte LDS reads 16 bytes apart (one ds_read2_b64) where
// the second image had one 16-byte read.  The factorization writes 168 MB less per step, and what both sweeps stream is
// half as large as the Infinity Cache instead of larger than it.
__device__ __forceinline__ int wf_image_bytes(int m, int n) {
  const int npair = (n + 1) >> 1;
  return 16 * (npair * (m - 1) - npair * (npair - 1));
}
// entry (row r, column k), r > k, of the forward image as an index of doubles: 2 * (j (m - j - 2) + r - 1) + (k & 1), j = k / 2
__device__ __forceinline__ int wf_col_base(int k, int m) {
  const int j = k >> 1;
  return 2 * (j * (m - j - 2) - 1) + (k & 1);
}
template <int MM>
__device__ __forceinline__ void wave_bwd_transpose(int m, int n, int lane, const double* __restrict__ imd, WBwdPre<MM>& p) {
  const bool colv = lane < n;
  // one address per lane, compile-time offsets from it (pairs of reads merge into ds_read2_b64): entries that do not
  // exist (row <= column, row >= m, lane >= n) are read from wherever that lands inside the staging area -- never
  // below it: the smallest offset used is base + 2 >= 0 -- and replaced by zero
  const double* q = imd + (colv ? wf_col_base(lane, m) : 0);
#pragma unroll
  for (int i4 = 0; i4 < MM / 2; i4 += 4) {
    if (i4 < 12 || 2 * i4 < m) {          // (uniform; the recurrence skips the same groups of rows)
#pragma unroll
      for (int i = i4; i < i4 + 4; ++i) {
        const bool ok0 = colv & (2 * i > lane) & (2 * i < m), ok1 = colv & (2 * i + 1 > lane) & (2 * i + 1 < m);
        const double d0 = (i == 0) ? 0.0 : q[4 * i];          // (row 0 is above every column's diagonal)
        const double d1 = q[4 * i + 2];
        p.up[i].x = ok0 ? d0 : 0.0;
        p.up[i].y = ok1 ? d1 : 0.0;
      }
    } else {
#pragma unroll
      for (int i = i4; i < i4 + 4; ++i) p.up[i] = double2_t{0.0, 0.0};
    }
  }
}
// the general (wide / part-solve) form: the forward image as a flat stream into the wave's LDS area `im` (any size up to
// a 64 x 64 front's 16 KB; the launch sizes the area), its transpose out of it
template <int MM, bool FAST>
__device__ __forceinline__ void wave_bwd_load(const WTask& t, int lane, WBwdPre<MM>& p, const double* __restrict__ Lf,
                                              double2_t* __restrict__ im,
                                              const int32_t* __restrict__ gperm, const int32_t* __restrict__ gvar,
                                              const int32_t* __restrict__ cmap, const int32_t* __restrict__ rlist,
                                              const double* __restrict__ xp, const double* __restrict__ slotv,
                                              const double* __restrict__ cvec, bool want_var) {
  const int m = t.m, n = t.n;
  {
    const int nbytes = wf_image_bytes(m, n);
    const __amdgpu_buffer_rsrc_t rs = wave_rsrc(Lf + t.lfoff, nbytes);
    for (int c0 = 0; c0 * 1024 < nbytes; c0 += 4) {        // (uniform) four 1 KB pieces in flight at a time
      double2_t ch[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) ch[c] = wave_ld_f64x2(rs, lane * 16, (c0 + c) * 1024);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((c0 + c) * 1024 < nbytes) im[(c0 + c) * 64 + lane] = ch[c];
    }
    wave_bwd_transpose<MM>(m, n, lane, reinterpret_cast<const double*>(im), p);
  }
  const int64_t s = int64_t(t.sptr) + lane;
  if (FAST) {
    const bool piv = lane < n;
    const int gp = gperm[s], gv = gvar[s], pr = cmap[t.moff + max(lane - n, 0)];
    const double xs = slotv[s], z = cvec[t.moff + max(lane - n, 0)];
    p.pos = gp;
    p.var = gv;
    p.prow = pr;
    p.x = piv ? xs : ((lane < m) ? z : 0.0);    // inside a group the parent's values replace z (compute step)
  } else {
    p.pos = p.var = p.prow = 0;
    p.x = 0.0;
    if (lane < n) {
      p.pos = gperm[s];
      p.x = slotv ? slotv[s] : xp[p.pos];       // this front's part of the forward result: nobody else writes it
      if (want_var) p.var = gvar[s];
    } else if (lane < m) {                      // the ancestors' part of the solution
      if (t.flags & WT_PUSH) p.prow = cmap[t.moff + lane - n];       // from the parent, through LDS
      else if (t.flags & WT_ZVEC) p.x = cvec[t.moff + lane - n];     // scattered here by the parent (earlier launch)
      else p.x = xp[rlist[t.roff + lane]];
    }
  }
}

// the NARROW backward kernel's load: flat image + masked small loads, as wave_fwd_issue / wave_fwd_arrive
struct WBwdFlat {
  double2_t ch[WIMG_CHUNKS];
  double xs, z;
  int gp, gv, pr;
};
__device__ __forceinline__ void wave_bwd_issue(const WTask& t, int lane, WBwdFlat& q, const double* __restrict__ Lf,
                                               const int32_t* __restrict__ gperm, const int32_t* __restrict__ gvar,
                                               const int32_t* __restrict__ cmap, const double* __restrict__ slotv,
                                               const double* __restrict__ cvec) {
  const int m = t.m, n = t.n;
  const int nbytes = wf_image_bytes(m, n);
  const int oob = int(0x80000000);
  const __amdgpu_buffer_rsrc_t rs = wave_rsrc(Lf + t.lfoff, nbytes);
#pragma unroll
  for (int c = 0; c < WIMG_CHUNKS; ++c) {
    q.ch[c] = double2_t{0.0, 0.0};
    if (c < 2 || c * 1024 < nbytes) q.ch[c] = wave_ld_f64x2<GSLS_WS_NT_B>(rs, lane * 16, c * 1024);   // (uniform)
  }
  const int co = lane >= n ? (lane - n) : oob;
  q.gp = __builtin_amdgcn_raw_buffer_load_b32(wave_rsrc(gperm + t.sptr, n * 4), lane * 4, 0, 0);
  q.gv = __builtin_amdgcn_raw_buffer_load_b32(wave_rsrc(gvar + t.sptr, n * 4), lane * 4, 0, 0);
  q.pr = __builtin_amdgcn_raw_buffer_load_b32(wave_rsrc(cmap + t.moff, (m - n) * 4), lane >= n ? co * 4 : oob, 0, 0);
  q.xs = wave_ld_f64(wave_rsrc(slotv + t.sptr, n * 8), lane * 8);
  q.z = wave_ld_f64(wave_rsrc(cvec + t.moff, (m - n) * 8), lane >= n ? co * 8 : oob);
}
__device__ __forceinline__ void wave_bwd_arrive(const WTask& t, int lane, const WBwdFlat& q, WBwdPre<40>& p,
                                                double2_t* __restrict__ im) {
  const int m = t.m, n = t.n;
  const int nbytes = wf_image_bytes(m, n);
#pragma unroll
  for (int c = 0; c < WIMG_CHUNKS; ++c)
    if (c < 2 || c * 1024 < nbytes) im[c * 64 + lane] = q.ch[c];
  const bool colv = lane < n;
  wave_bwd_transpose<40>(m, n, lane, reinterpret_cast<const double*>(im), p);
  p.pos = q.gp;
  p.var = q.gv;
  p.prow = q.pr;
  p.x = colv ? q.xs : q.z;                // (z = 0 from the descriptor for the lanes from m on)
}

template <int MM, bool PULLS, int AS = 64>
__device__ __forceinline__ void wave_bwd_compute(const WTask& t, int lane, const WBwdPre<MM>& p, double* __restrict__ xfull,
                                                 const int32_t* __restrict__ gth_ptr, const int64_t* __restrict__ gth_src,
                                                 double* __restrict__ xp, double* __restrict__ xout,
                                                 const double* __restrict__ scale, double* __restrict__ cvec,
                                                 int2_t pull = int2_t{-1, -1}) {
  const int m = t.m, n = t.n;
  double x = p.x;
  {
    const bool arow = ((t.flags & WT_PUSH) != 0) & (lane >= n) & (lane < m);
    const double xv = xfull[arow ? t.pslot * AS + p.prow : WSLOT * AS + lane];
    x = arow ? xv : x;
  }
#pragma unroll
  for (int j4 = MM - 4; j4 >= 0; j4 -= 4) {   // the image holds zeros from row m on: whole groups of four are skipped
    if (j4 < 24 || j4 < m) {                  // (uniform; no memory operation inside, so the waits stay counted)
#pragma unroll
      for (int j = j4 + 3; j >= j4; --j) {
        if (j == 0) continue;
        const double xj = readlane_f64(x, j);
        const double u = (j & 1) ? p.up[j >> 1].y : p.up[j >> 1].x;
        x = fma(-u, xj, x);             // u = 0 for the lanes from j on and for the rows below the pivots
      }
    }
  }
  // pivot slot `lane` holds the analyse-time row gperm[sptr + lane] - sptr (numerical pivoting inside the front); the
  // children address the front by analyse-time rows (cmap, gather lists)
  const int arow = (lane < n) ? p.pos - t.sptr : lane;
  if ((t.flags & WT_INT) && (AS == 64 || lane < m)) xfull[t.myslot * AS + arow] = x;       // for the children inside the group
  if (PULLS && (t.flags & WT_PULL) && (t.flags & WT_DENSE)) {
    // the dense table is by analyse-time ROW (= lane in the load): the value of row r sits in the lane whose pivot
    // slot holds that row -- hand it over (a permutation of the pivot lanes; the other lanes keep theirs)
    const int lo = __builtin_amdgcn_ds_permute(arow * 4, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_permute(arow * 4, __double2hiint(x));
    const double xr = __hiloint2double(hi, lo);
    if (pull.x >= 0) cvec[pull.x] = xr;
    if (pull.y >= 0) cvec[pull.y] = xr;
  } else if (PULLS && (t.flags & WT_PULL)) {                   // ... and for those of earlier stages
    const int lr = min(arow, m - 1);
    const int g0 = gth_ptr[t.goff + lr], g1 = (lane < m) ? gth_ptr[t.goff + lr + 1] : g0;
    const int last = max(g1 - 1, g0);
    for (int g = g0; __any(g < g1); g += 8) {                  // eight destinations' indices in flight at a time
      int64_t idx[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) idx[q] = gth_src[min(g + q, last)];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (g + q < g1) cvec[idx[q]] = x;
    }
  }
  if (lane < n) {
    xp[p.pos] = x;
    if (xout) xout[p.var] = (PULLS && scale) ? x * scale[p.var] : x;     // (the look-ahead kernels run unscaled)
  }
}

// wave_bwd_compute for CG columns at once (narrow launches: no pulls), as wave_fwd_compute_cg
template <int MM, int AS, int CG>
__device__ __forceinline__ void wave_bwd_compute_cg(const WTask& t, int lane, const WBwdPre<MM>& p, const double (&xin)[CG],
                                                    double* __restrict__ xfull, int accw, double* __restrict__ xp, int64_t sx,
                                                    double* __restrict__ xout, int64_t sio) {
  const int m = t.m, n = t.n;
  double x[CG];
  const bool prow = ((t.flags & WT_PUSH) != 0) & (lane >= n) & (lane < m);
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    const double xv = xfull[c * accw + (prow ? t.pslot * AS + p.prow : accw - 64 + lane)];
    x[c] = prow ? xv : xin[c];
  }
#pragma unroll
  for (int j4 = MM - 4; j4 >= 0; j4 -= 4) {
    if (j4 < 24 || j4 < m) {
#pragma unroll
      for (int j = j4 + 3; j >= j4; --j) {
        if (j == 0) continue;
        const double u = (j & 1) ? p.up[j >> 1].y : p.up[j >> 1].x;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          const double xj = readlane_f64(x[c], j);
          x[c] = fma(-u, xj, x[c]);
        }
      }
    }
  }
  const int arow = (lane < n) ? p.pos - t.sptr : lane;
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    if ((t.flags & WT_INT) && (AS == 64 || lane < m)) xfull[c * accw + t.myslot * AS + arow] = x[c];
    if (lane < n) {
      xp[c * sx + p.pos] = x[c];
      if (xout) xout[c * sio + p.var] = x[c];
    }
  }
}

__device__ __forceinline__ WTask wave_task(const WTask* __restrict__ tasks, int idx) {
  return tasks[__builtin_amdgcn_readfirstlane(idx)];
}

// NARROW: every front of the launch has at most 32 pivot columns: the loads of the group's next front are in flight
// while the wave works on the current one (two register sets, A and B).  Otherwise fronts of 33..64 columns are
// among them: one at a time (two 64-column register sets do not fit).
// CG (narrow flat launches, several right-hand sides): ONE wave takes a front for CG columns -- the image, D and the maps
// are loaded once and the recurrence runs CG times on them, each column with its own right-hand side, LDS accumulators
// and vectors (cs.R counts column GROUPS then).  The bottom stage of a KKT tree is bound by the memory system, not by
// the arithmetic: with a launch per column (CG = 1, grid x R) eight columns read the 170 MB image eight times.
// (Tried on top and dropped: the next front's loads in flight during this front's CG recurrences -- 0.60 -> 0.62 ms
// for eight columns.)
struct WGather {                 // GATHER launches: the caller's right-hand side, invp, and the positions this launch permutes
  const double* xin;             // for the launches behind it (workgroups from `nblk` on)
  const int32_t* invp;
  const int32_t* list;
  int cnt, nblk;
};
template <bool APPLY_D, bool NARROW, bool FLAT = false, int CG = 1, bool GATHER = false>
__global__ void __launch_bounds__(256)
k_wsolve_fwd(const WGroup* __restrict__ groups, int ngroup, const WTask* __restrict__ tasks,
             const double* __restrict__ Lf, const double* __restrict__ D, const int32_t* __restrict__ gperm,
             const int32_t* __restrict__ cmap, const int32_t* __restrict__ gth_ptr,
             const int64_t* __restrict__ gth_src, double* __restrict__ xp, double* __restrict__ slotv,
             double* __restrict__ cvec, Cols cs, const int32_t* __restrict__ pull2, int unit_tbeg, int wimg_units,
             WGather wg = WGather{nullptr, nullptr, nullptr, 0, 0}) {
  static_assert(CG == 1 || (NARROW && FLAT), "column groups: narrow flat launches only");
  static_assert(!GATHER || (NARROW && FLAT && CG == 1), "fused input permutation: the single-column narrow flat launch");
  if constexpr (GATHER) {
    if (int(blockIdx.x) >= wg.nblk) {       // trailing workgroups: xp = P x for the positions of every other launch
      const int i = (int(blockIdx.x) - wg.nblk) * 256 + int(threadIdx.x);
      if (i < wg.cnt) {
        const int pos = wg.list[i];
        xp[pos] = wg.xin[wg.invp[pos]];
      }
      return;
    }
  }
  constexpr int AS = (NARROW && FLAT) ? WACC_NARROW : 64;
  // accumulators: a row per LDS slot + a spare row (64 wide) for the masked lanes.  CG = 1: WSLOT rows, static; column
  // groups: the launch's deepest slot (wimg_units carries it), CG sets per wave in dynamic LDS -- what decides how many
  // waves share a CU there
  const int depth = (CG > 1) ? wimg_units : WSLOT;
  const int ACCW = depth * AS + 64;
  __shared__ double accs[CG > 1 ? 1 : 4][CG > 1 ? 1 : WSLOT * AS + 64];
  extern __shared__ __attribute__((aligned(16))) double2_t wdyn[];    // wide launches: wimg_units 16-byte units per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GSLS_COLS;
  xp += col_ * CG * cs.sx;
  cvec += col_ * CG * cs.scv;
  if (slotv) slotv += col_ * CG * cs.sxs;
  const int gi = __builtin_amdgcn_readfirstlane(bid * 4 + wave);
  if (gi >= ngroup) return;
  double* acc = (CG > 1) ? reinterpret_cast<double*>(wdyn) + wave * CG * ACCW : accs[wave];
  for (int i = 0; i < CG * ACCW; i += 64) acc[i + lane] = 0.0;
  // unit_tbeg >= 0: every run of this launch is ONE front, run gi = task unit_tbeg + gi -- no group record to fetch
  // (a stage of the upper tree is a chain of dependent round trips: this is one of them)
  const WGroup g = (unit_tbeg >= 0) ? WGroup{unit_tbeg + gi, 1} : groups[gi];
  const int te = g.tbeg + g.tcnt;
  if constexpr (NARROW) {
    // One register set: under load a round trip to HBM takes several times longer than a front's arithmetic, so a
    // second register set (the next front's loads in flight during this front's recurrence) hides little and
    // halves the number of waves per CU -- and it is the number of fronts in flight per CU that sets the bandwidth.
    // What is fetched ahead is the next front's task record (scalar registers): every load of a front is then one
    // hop behind the previous front's last instruction.
    int ti = g.tbeg;
    WTask ta = wave_task(tasks, ti);
#ifdef GSLS_STAMPS   // where a wave's time goes, summed over the launch (100 MHz ticks): g_stamps[32..36]
    unsigned long long pht = __builtin_amdgcn_s_memrealtime(), ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, nfr = 0;
#define WPH(x) do { const unsigned long long t__ = __builtin_amdgcn_s_memrealtime(); x += t__ - pht; pht = t__; } while (0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WPH(ph3);                             // group record + first task record
#else
#define WPH(x) do {} while (0)
#endif
    while (true) {
      const WTask tb = wave_task(tasks, min(ti + 1, te - 1));
      WFwdPre<32> A;
      double rhsx[CG];                    // the other columns' right-hand sides (rhsx[0] unused)
      if constexpr (FLAT) {
        __shared__ __attribute__((aligned(16))) double2_t wimg[4][WIMG_BYTES / 16];
        WFwdFlat<APPLY_D> Q;
        wave_fwd_issue<APPLY_D, GATHER>(ta, lane, Q, Lf, D, gperm, cmap, xp, wg.invp, wg.xin);
#pragma unroll
        for (int c = 1; c < CG; ++c) rhsx[c] = wave_ld_f64(wave_rsrc(xp + c * cs.sx + ta.sptr, ta.n * 8), lane * 8);
        wave_fwd_arrive<APPLY_D>(ta, lane, Q, A, wimg[wave]);
      } else {
        wave_fwd_load<32, APPLY_D>(ta, lane, A, Lf, D, gperm, cmap, xp);
      }
#ifdef GSLS_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WPH(ph0);                           // image, right-hand side, D, maps: issue -> arrival
      ++nfr;
#endif
      if constexpr (CG > 1) {
        rhsx[0] = A.rhs;
        wave_fwd_compute_cg<32, APPLY_D, AS, CG>(ta, lane, A, rhsx, acc, ACCW, xp, cs.sx, slotv, cs.sxs, cvec, cs.scv);
      } else {
        wave_fwd_compute<32, APPLY_D, false, AS>(ta, lane, A, acc, gth_ptr, gth_src, xp, slotv, cvec);
      }
#ifdef GSLS_STAMPS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WPH(ph1);                           // recurrence, LDS hand-off, stores issued (and the next task record arrived)
#endif
      if (++ti >= te) break;
      ta = tb;
    }
#ifdef GSLS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WPH(ph2);                             // the last stores
    if (lane == 0) {
      atomicAdd(&g_stamps[32], ph0); atomicAdd(&g_stamps[33], ph1); atomicAdd(&g_stamps[34], ph2);
      atomicAdd(&g_stamps[35], ph3); atomicAdd(&g_stamps[36], nfr); atomicAdd(&g_stamps[37], 1ull);
    }
#endif
#undef WPH
  } else {
    for (int ti = g.tbeg; ti < te; ++ti) {
      const WTask t = wave_task(tasks, ti);
      const int2_t pull = wave_pull2_issue(t, lane, pull2);
      if (t.n <= 32) {
        WFwdPre<32> P;
        wave_fwd_load<32, APPLY_D>(t, lane, P, Lf, D, gperm, cmap, xp);
        wave_fwd_compute<32, APPLY_D, true>(t, lane, P, acc, gth_ptr, gth_src, xp, slotv, cvec, pull);
      } else {
        WFwdPre<64> P;
        wave_fwd_load<64, APPLY_D>(t, lane, P, Lf, D, gperm, cmap, xp);
        wave_fwd_compute<64, APPLY_D, true>(t, lane, P, acc, gth_ptr, gth_src, xp, slotv, cvec, pull);
      }
    }
  }
}

// NARROW: every front of the launch has at most 40 rows (two register sets, as in the forward kernel)
template <bool NARROW, bool FLAT = false, int CG = 1>
__global__ void __launch_bounds__(256)
k_wsolve_bwd(const WGroup* __restrict__ groups, int ngroup, const WTask* __restrict__ tasks,
             const double* __restrict__ Lb, const int32_t* __restrict__ gperm, const int32_t* __restrict__ gvar,
             const int32_t* __restrict__ cmap, const int32_t* __restrict__ rlist, const int32_t* __restrict__ gth_ptr,
             const int64_t* __restrict__ gth_src, double* __restrict__ xp, const double* __restrict__ slotv,
             double* __restrict__ xout, const double* __restrict__ scale, double* __restrict__ cvec, Cols cs,
             const int32_t* __restrict__ pull2, int unit_tbeg, int wimg_units) {
  static_assert(CG == 1 || (NARROW && FLAT), "column groups: narrow flat launches only");
  constexpr int AS = (NARROW && FLAT) ? WACC_NARROW : 64;
  const int depth = (CG > 1) ? wimg_units : WSLOT;     // (as in the forward kernel)
  const int ACCW = depth * AS + 64;
  __shared__ double xfs[CG > 1 ? 1 : 4][CG > 1 ? 1 : WSLOT * AS + 64];
  extern __shared__ __attribute__((aligned(16))) double2_t wdyn[];    // wide launches: wimg_units 16-byte units per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  GSLS_COLS;
  xp += col_ * CG * cs.sx;
  cvec += col_ * CG * cs.scv;
  if (slotv) slotv += col_ * CG * cs.sxs;
  if (xout) xout += col_ * CG * cs.sio;
  const int gi = __builtin_amdgcn_readfirstlane(bid * 4 + wave);
  if (gi >= ngroup) return;
  double* xfull = (CG > 1) ? reinterpret_cast<double*>(wdyn) + wave * CG * ACCW : xfs[wave];
  const WGroup g = (unit_tbeg >= 0) ? WGroup{unit_tbeg + gi, 1} : groups[gi];
  const bool wv = xout != nullptr;
  if constexpr (NARROW) {
    int ti = g.tbeg + g.tcnt - 1;
    WTask ta = wave_task(tasks, ti);
    while (true) {
      const WTask tb = wave_task(tasks, max(ti - 1, g.tbeg));
      WBwdPre<40> A;
      double xcol[CG];                    // (xcol[0] unused)
      if constexpr (FLAT) {
        __shared__ __attribute__((aligned(16))) double2_t wimg[4][WIMG_BYTES / 16];
        WBwdFlat Q;
        wave_bwd_issue(ta, lane, Q, Lb, gperm, gvar, cmap, slotv, cvec);
#pragma unroll
        for (int c = 1; c < CG; ++c) {      // the other columns' forward results and ancestors' values (as Q.xs / Q.z)
          const int co = lane >= ta.n ? (lane - ta.n) * 8 : int(0x80000000);
          const double xs = wave_ld_f64(wave_rsrc(slotv + c * cs.sxs + ta.sptr, ta.n * 8), lane * 8);
          const double z = wave_ld_f64(wave_rsrc(cvec + c * cs.scv + ta.moff, (ta.m - ta.n) * 8), co);
          xcol[c] = (lane < ta.n) ? xs : z;
        }
        wave_bwd_arrive(ta, lane, Q, A, wimg[wave]);
      } else {
        wave_bwd_load<40, true>(ta, lane, A, Lb, wdyn + wave * wimg_units, gperm, gvar, cmap, rlist, xp, slotv, cvec, wv);
      }
      if constexpr (CG > 1) {
        xcol[0] = A.x;
        wave_bwd_compute_cg<40, AS, CG>(ta, lane, A, xcol, xfull, ACCW, xp, cs.sx, xout, cs.sio);
      } else {
        wave_bwd_compute<40, false, AS>(ta, lane, A, xfull, gth_ptr, gth_src, xp, xout, scale, cvec);
      }
      if (--ti < g.tbeg) break;
      ta = tb;
    }
  } else {
    for (int ti = g.tbeg + g.tcnt - 1; ti >= g.tbeg; --ti) {
      const WTask t = wave_task(tasks, ti);
      const int2_t pull = wave_pull2_issue(t, lane, pull2);
      if (t.m <= 32) {
        WBwdPre<32> P;
        wave_bwd_load<32, false>(t, lane, P, Lb, wdyn + wave * wimg_units, gperm, gvar, cmap, rlist, xp, slotv, cvec, wv);
        wave_bwd_compute<32, true>(t, lane, P, xfull, gth_ptr, gth_src, xp, xout, scale, cvec, pull);
      } else {
        WBwdPre<64> P;
        wave_bwd_load<64, false>(t, lane, P, Lb, wdyn + wave * wimg_units, gperm, gvar, cmap, rlist, xp, slotv, cvec, wv);
        wave_bwd_compute<64, true>(t, lane, P, xfull, gth_ptr, gth_src, xp, xout, scale, cvec, pull);
      }
    }
  }
}

// The TOP of the tier in one launch (round 3).  The last stages of a tree hold a handful of fronts each (metric
// workload: 4, 1, 1), and a launch per stage and direction costs 4 - 7 us of which the arithmetic is a fraction: a
// chain of dependent round trips to memory (group record -> task record -> gather lists -> contribution vectors) plus
// the launch itself.  Here ONE workgroup of eight waves walks those stages bottom-up, applies D^-1 (the forward
// step does that) and walks them top-down again, with a workgroup barrier where a launch boundary was; the eight
// waves of a stage take its groups in turn.  Vectors still travel through cvec / xp / slotv in memory -- waves of one
// workgroup share the CU's vector cache, so the barrier's workgroup-scope fence is all the coherence this needs (no
// L2 write-back: the cost that ruled out chaining workgroups on different XCDs).  The first thing the kernel does is
// to touch every task record, image and gather list of its stages, so that the dependent accesses of the sweep find
// them in this XCD's L2.
constexpr int WTAIL_STAGES = 8, WTAIL_WAVES = 8;
struct WTail {
  int nst;
  int gbeg[WTAIL_STAGES], gcnt[WTAIL_STAGES];   // group ranges of the tail's stages, bottom-up
  int tbeg, tcnt;                               // their tasks (one range: launch order)
  int64_t lf0, lf1, lb0, lb1;                   // element ranges of their images
  int64_t gp0, gp1, gs0, gs1;                   // ... of their gather pointers / sources
};
template <bool APPLY_D>
__global__ void __launch_bounds__(64 * WTAIL_WAVES)
k_wsolve_tail(WTail tl, const WGroup* __restrict__ groups, const WTask* __restrict__ tasks,
              const double* __restrict__ Lf, const double* __restrict__ Lb, const double* __restrict__ D,
              const int32_t* __restrict__ gperm, const int32_t* __restrict__ gvar, const int32_t* __restrict__ cmap,
              const int32_t* __restrict__ rlist, const int32_t* __restrict__ gth_ptr, const int64_t* __restrict__ gth_src,
              double* __restrict__ xp, double* __restrict__ slotv, double* __restrict__ xout,
              const double* __restrict__ scale, double* __restrict__ cvec, Cols cs, int do_fwd, int do_bwd,
              double* __restrict__ sink, const int32_t* __restrict__ pull2, int wimg_units) {
  __shared__ double accs[WTAIL_WAVES][(WSLOT + 1) * 64];
  extern __shared__ __attribute__((aligned(16))) double2_t wdyn[];    // wimg_units 16-byte units per wave (backward half)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col_ = blockIdx.x;
  xp += col_ * cs.sx;
  cvec += col_ * cs.scv;
  if (slotv) slotv += col_ * cs.sxs;
  if (xout) xout += col_ * cs.sio;
  double* acc = accs[wave];
#ifdef GSLS_STAMPS     // wave 0's timeline (100 MHz ticks) into g_stamps[0..]: tools/stamp_tail.py
  int stk = 0;
#define TSTAMP() do { if (wave == 0 && lane == 0 && blockIdx.x == 0 && stk < 62) g_stamps[stk++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TSTAMPW() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); TSTAMP(); } while (0)
#else
#define TSTAMP() do {} while (0)
#define TSTAMPW() do {} while (0)
#endif
  TSTAMP();
#pragma unroll
  for (int i = 0; i <= WSLOT; ++i) acc[i * 64 + lane] = 0.0;
  {
    // warm this XCD's L2: one 64-byte line per lane and round
    double warm = 0.0;
    const int tid = threadIdx.x, nt = 64 * WTAIL_WAVES;
    const double* tk = reinterpret_cast<const double*>(tasks + tl.tbeg);
    for (int64_t i = int64_t(tid) * 8; i < int64_t(tl.tcnt) * 8; i += int64_t(nt) * 8) warm += tk[i];
    if (do_fwd) for (int64_t i = tl.lf0 + int64_t(tid) * 8; i < tl.lf1; i += int64_t(nt) * 8) warm += Lf[i];
    if (do_bwd && !do_fwd) for (int64_t i = tl.lf0 + int64_t(tid) * 8; i < tl.lf1; i += int64_t(nt) * 8) warm += Lf[i];
    for (int64_t i = tl.gp0 + int64_t(tid) * 16; i < tl.gp1; i += int64_t(nt) * 16) warm += double(gth_ptr[i]);
    for (int64_t i = tl.gs0 + int64_t(tid) * 8; i < tl.gs1; i += int64_t(nt) * 8) warm += double(gth_src[i]);
    // ... and the small vectors each front reads beside its image (right-hand side, D, maps, dense pull pairs): measured
    // (tools/stamp_tail.py) a front's load phase waits 1.1 - 1.5 us for exactly these when only the images are warm
    for (int ti = wave; ti < tl.tcnt; ti += WTAIL_WAVES) {
      const WTask t = wave_task(tasks, tl.tbeg + ti);
      const int64_t sl = int64_t(t.sptr) + lane;
      if (lane < t.n) warm += xp[sl] + D[2 * sl] + double(gperm[sl]) + double(gvar[sl]);
      if (lane < t.m - t.n) warm += double(cmap[t.moff + lane]) + cvec[t.moff + lane];
      if ((t.flags & WT_DENSE) && lane < t.m) {
        const int a = pull2[2 * (t.goff + lane)], b = pull2[2 * (t.goff + lane) + 1];
        warm += cvec[max(a, 0)] + cvec[max(b, 0)];
      }
    }
    if (warm == 1.2345e-300) sink[0] = warm;      // (never: keeps the loads)
  }
  TSTAMPW();
  if (do_fwd)
    for (int sg = 0; sg < tl.nst; ++sg) {
      for (int gi = wave; gi < tl.gcnt[sg]; gi += WTAIL_WAVES) {
        const WGroup g = groups[__builtin_amdgcn_readfirstlane(tl.gbeg[sg] + gi)];
        for (int ti = g.tbeg; ti < g.tbeg + g.tcnt; ++ti) {
          const WTask t = wave_task(tasks, ti);
          TSTAMPW();                        // group + task records
          const int2_t pull = wave_pull2_issue(t, lane, pull2);
          if (t.n <= 32) {
            WFwdPre<32> P;
            wave_fwd_load<32, APPLY_D>(t, lane, P, Lf, D, gperm, cmap, xp);
            TSTAMPW();                      // loads
            wave_fwd_compute<32, APPLY_D, true>(t, lane, P, acc, gth_ptr, gth_src, xp, slotv, cvec, pull);
          } else {
            WFwdPre<64> P;
            wave_fwd_load<64, APPLY_D>(t, lane, P, Lf, D, gperm, cmap, xp);
            TSTAMPW();                      // loads
            wave_fwd_compute<64, APPLY_D, true>(t, lane, P, acc, gth_ptr, gth_src, xp, slotv, cvec, pull);
          }
          TSTAMPW();                        // pulls, recurrence, stores
        }
      }
      __syncthreads();
      TSTAMP();                             // barrier
    }
  if (do_bwd) {
    const bool wv = xout != nullptr;
    for (int sg = tl.nst - 1; sg >= 0; --sg) {
      for (int gi = wave; gi < tl.gcnt[sg]; gi += WTAIL_WAVES) {
        const WGroup g = groups[__builtin_amdgcn_readfirstlane(tl.gbeg[sg] + gi)];
        for (int ti = g.tbeg + g.tcnt - 1; ti >= g.tbeg; --ti) {
          const WTask t = wave_task(tasks, ti);
          const int2_t pull = wave_pull2_issue(t, lane, pull2);
          if (t.m <= 32) {
            WBwdPre<32> P;
            wave_bwd_load<32, false>(t, lane, P, Lf, wdyn + wave * wimg_units, gperm, gvar, cmap, rlist, xp, slotv, cvec, wv);
            wave_bwd_compute<32, true>(t, lane, P, acc, gth_ptr, gth_src, xp, xout, scale, cvec, pull);
          } else {
            WBwdPre<64> P;
            wave_bwd_load<64, false>(t, lane, P, Lf, wdyn + wave * wimg_units, gperm, gvar, cmap, rlist, xp, slotv, cvec, wv);
            wave_bwd_compute<64, true>(t, lane, P, acc, gth_ptr, gth_src, xp, xout, scale, cvec, pull);
          }
        }
      }
      __syncthreads();
    }
  }
}

// =================================================================================================
// PIVOT-ORDER DISCOVERY (round 3): run-time delayed pivots, confined to one kernel.
// The reference carries a column that fails the threshold test into the parent front during the pass
// (assemble.hxx:244-264, factor.hxx:57-106, ldlt_tpp.cxx:140-240).  The product kernels here are statically scheduled
// and sized, so until now a failure cost a host round trip per LEVEL of the delay cascade (flag, move to the parent,
// re-analyse, another pass: 10-20 passes for an interior-point KKT system late in the run).  This kernel does what the
// reference does -- threshold partial pivoting over ALL fully-summed columns of a front, 1x1 and 2x2 pivots, what cannot
// be eliminated handed to the parent together with its rows of the Schur complement -- in ONE bottom-up sweep, for trees
// whose fronts (delays included) fit a wavefront: a wave per front, the front as a full symmetric matrix in LDS, lane =
// row.  It does not produce factors: its output is the ELIMINATION SEQUENCE it found (per front: the variables in pivot
// order, 2x2 pairs marked).  The host turns that into an order + hints, re-analyses ONCE, and the static kernels factorize
// in that order (gsls_api.cpp, pass loop).  A front that would exceed 64 rows, or pass up more than DISC_DCAP columns,
// raises the overflow flag and the old repair loop takes over.
//   work matrix indices: [0, din) columns delayed by the children (child order), [din, din + n) own pivots,
//   [din + n, din + m) contribution rows; candidates = [0, din + n).
// =================================================================================================
constexpr int DISC_LD = 65, DISC_DCAP = 12;      // wave fronts: at most 64 rows with their incoming delays, 12 delays out
constexpr int DISC_WIDE_M = 48;                  // fronts with more rows than this take the workgroup form:
constexpr int DISC_WIDE_IN = 64, DISC_WIDE_OUT = 64, DISC_WIDE_MAX = 512;   // up to 64 delays in / out, 512 rows in all
struct DiscTask {                   // per front, indexed by node
  int32_t m, n, sptr, parent;
  int32_t cbeg, ccnt, moff, qcap;   // children in clist; its child->parent map; row stride of its block in the arena
  int64_t a0, coff;                 // its entries of A (asrc / arc); its block in the discovery arena
  int32_t acnt, dcap;               // ... ; delays it may pass up
  int64_t woff;                     // workgroup form: its work matrix in the scratch (mcap x mcap, column-major)
  int32_t mcap, wide;
  int64_t voff, poff;               // its delayed variables in dvar, its elimination sequence in pseq / ptwo
};
__device__ __forceinline__ double disc_wave_max(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__global__ void __launch_bounds__(256)
k_front_discover(const DiscTask* __restrict__ tasks, const int32_t* __restrict__ list, int cnt,
                 const int32_t* __restrict__ clist, const int32_t* __restrict__ cmap, const int32_t* __restrict__ invp,
                 const int64_t* __restrict__ asrc, const uint32_t* __restrict__ arc, const double* __restrict__ val,
                 double* __restrict__ arena, int32_t* __restrict__ ddelay, int32_t* __restrict__ dvar,
                 int32_t* __restrict__ pseq, uint8_t* __restrict__ ptwo, int32_t* __restrict__ pcnt,
                 int32_t* __restrict__ flags, double small, double u, int nn) {
  extern __shared__ __attribute__((aligned(16))) double dsh[];
  __shared__ int idsh[4][64];
  __shared__ int twosh[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (ti >= cnt) return;
  const int s = list[ti];
  const DiscTask t = tasks[s];
  double* W = dsh + wave * (64 * DISC_LD);
  int* ids = idsh[wave];
  int* two = twosh[wave];
#define DSYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
  int din = 0;
  for (int ci = 0; ci < t.ccnt; ++ci) din += ddelay[clist[t.cbeg + ci]];
  din = __builtin_amdgcn_readfirstlane(din);
  const int M = din + t.m, nc = din + t.n;
  if (M > 64) {                                     // does not fit a wavefront: the whole discovery is void
    if (lane == 0) { atomicOr(&flags[0], 1); ddelay[s] = 0; pcnt[s] = 0; }
    return;
  }
  for (int c = 0; c < M; ++c) W[lane * DISC_LD + c] = 0.0;
  ids[lane] = (lane >= din && lane < nc) ? invp[t.sptr + lane - din] : -1;
  two[lane] = 0;
  DSYNC();
  // own entries of A (lower triangle of the front, both halves of the symmetric work matrix)
  for (int e = lane; e < t.acnt; e += 64) {
    const uint32_t rc = arc[t.a0 + e];
    const int r = din + int(rc & 0xffffu), c = din + int(rc >> 16);
    const double v = val[asrc[t.a0 + e]];
    W[r * DISC_LD + c] = v;
    W[c * DISC_LD + r] = v;
  }
  DSYNC();
  // the children's Schur complements: delayed columns first, then their contribution rows
  {
    int base = 0;
    for (int ci = 0; ci < t.ccnt; ++ci) {
      const int c = clist[t.cbeg + ci];
      const DiscTask tc = tasks[c];
      const int dc = __builtin_amdgcn_readfirstlane(ddelay[c]);
      const int q = dc + tc.m - tc.n;
      if (lane < dc) ids[base + lane] = dvar[tc.voff + lane];
      const int mi = (lane < dc) ? base + lane : ((lane < q) ? din + cmap[tc.moff + lane - dc] : 0);
      const double* Cb = arena + tc.coff;
      for (int j = 0; j < q; ++j) {
        const int mj = __shfl(mi, j);
        if (lane < q) W[mi * DISC_LD + mj] += Cb[int64_t(lane) * tc.qcap + j];
      }
      base += dc;
      DSYNC();
    }
  }
  // ---- threshold partial pivoting over the candidates [0, nc), rows [0, M): the rules of ldlt_tpp.cxx:140-240 ------
  auto colmax = [&](int col, int from, int x1, int x2) -> double {
    const double v = (lane >= from && lane < M && lane != x1 && lane != x2) ? fabs(W[lane * DISC_LD + col]) : 0.0;
    return disc_wave_max(v);
  };
  auto swap_sym = [&](int x, int y) {
    if (x == y) return;
    {
      const double a = W[lane * DISC_LD + x], b = W[lane * DISC_LD + y];
      W[lane * DISC_LD + x] = b;
      W[lane * DISC_LD + y] = a;
    }
    DSYNC();
    {
      const double a = W[x * DISC_LD + lane], b = W[y * DISC_LD + lane];
      W[x * DISC_LD + lane] = b;
      W[y * DISC_LD + lane] = a;
    }
    if (lane == 0) { const int v = ids[x]; ids[x] = ids[y]; ids[y] = v; }
    DSYNC();
  };
  auto pivot_1x1 = [&](int p) {
    const double d = 1.0 / W[p * DISC_LD + p];
    const double lrp = W[lane * DISC_LD + p];
    if (lane > p && lane < M)
      for (int c = p + 1; c < M; ++c) W[lane * DISC_LD + c] -= (lrp * W[c * DISC_LD + p]) * d;
    DSYNC();
  };
  auto pivot_2x2 = [&](int p, double d11, double d21, double d22) {
    const double a1r = W[lane * DISC_LD + p], a2r = W[lane * DISC_LD + p + 1];
    if (lane > p + 1 && lane < M)
      for (int c = p + 2; c < M; ++c) {
        const double a1c = W[c * DISC_LD + p], a2c = W[c * DISC_LD + p + 1];
        W[lane * DISC_LD + c] -= d11 * (a1r * a1c) + d21 * (a2r * a1c + a1r * a2c) + d22 * (a2r * a2c);
      }
    DSYNC();
  };
  int p = 0;
  while (p < nc) {
    if (colmax(p, p, -1, -1) < small) { ++p; continue; }            // a zero pivot (ldlt_tpp.cxx:150-160)
    bool found = false;
    for (int q = p + 1; q < nc && !found; ++q) {
      const double aq = (lane >= p && lane < M) ? fabs(W[lane * DISC_LD + q]) : 0.0;
      const double mall = disc_wave_max(aq);
      if (mall < small) {
        swap_sym(p, q);
        ++p;
        found = true;
        break;
      }
      // largest entry of column q among the rows [p, q): smallest index on ties
      const double mv = disc_wave_max((lane >= p && lane < q) ? aq : -1.0);
      const unsigned long long hit = __ballot((lane >= p && lane < q) && aq == mv);
      const int tt = __builtin_amdgcn_readfirstlane(int(__ffsll((long long)hit)) - 1);
      const double maxt = colmax(tt, p, tt, q);
      double maxq = colmax(q, p, q, tt);
      const double a11 = W[tt * DISC_LD + tt], a21 = W[q * DISC_LD + tt], a22 = W[q * DISC_LD + q];
      bool ok2 = false;
      double d11 = 0.0, d21 = 0.0, d22 = 0.0;
      const double maxpiv = fmax(fabs(a11), fmax(fabs(a21), fabs(a22)));
      if (maxpiv >= small) {                       // test_2x2 (ldlt_tpp.cxx:99-130)
        const double detscale = 1.0 / maxpiv;
        const double detpiv0 = (a11 * detscale) * a22, detpiv1 = (a21 * detscale) * a21;
        const double detpiv = detpiv0 - detpiv1;
        if (!(fabs(detpiv) < fmax(small, fmax(fabs(detpiv0 / 2), fabs(detpiv1 / 2))))) {
          d11 = (a22 * detscale) / detpiv;
          d21 = (-a21 * detscale) / detpiv;
          d22 = (a11 * detscale) / detpiv;
          if (fmax(maxq, maxt) < small) ok2 = true;
          else {
            const double x1 = fabs(d11) * maxt + fabs(d21) * maxq;
            const double x2 = fabs(d21) * maxt + fabs(d22) * maxq;
            ok2 = (u * fmax(x1, x2) < 1.0);
          }
        }
      }
      if (ok2) {
        swap_sym(tt, p);
        swap_sym(q, p + 1);
        pivot_2x2(p, d11, d21, d22);
        if (lane == 0) two[p] = 1;
        p += 2;
        found = true;
        break;
      }
      maxq = fmax(maxq, fabs(a21));
      if (fabs(a22) >= u * maxq && fabs(a22) >= small) {
        swap_sym(q, p);
        pivot_1x1(p);
        p += 1;
        found = true;
        break;
      }
    }
    if (found) continue;
    const double maxp = colmax(p, p, p, -1);
    const double app = W[p * DISC_LD + p];
    if (fabs(app) >= u * maxp && fabs(app) >= small) {
      pivot_1x1(p);
      ++p;
    } else {
      break;                                       // no more pivots in this front: the rest is delayed
    }
  }
  DSYNC();
  const bool root = t.parent >= nn;
  const int nelim = p;
  const int count = root ? nc : nelim;             // (a root keeps what is left: zero pivots)
  const int dout = root ? 0 : nc - nelim;
  if (dout > t.dcap) {
    if (lane == 0) { atomicOr(&flags[0], 2); ddelay[s] = 0; pcnt[s] = 0; }
    return;
  }
  if (lane < count) {
    pseq[t.poff + lane] = ids[lane];
    ptwo[t.poff + lane] = uint8_t(two[lane]);
  }
  if (lane < dout) dvar[t.voff + lane] = ids[nelim + lane];
  if (lane == 0) {
    pcnt[s] = count;
    ddelay[s] = dout;
    if (dout > 0) atomicAdd(&flags[1], dout);
  }
  if (!root) {
    const int q = dout + t.m - t.n;                // what goes to the parent: leftover candidates, then the contribution rows
    double* Cb = arena + t.coff;
    if (lane < q)
      for (int j = 0; j < q; ++j) Cb[int64_t(lane) * t.qcap + j] = W[(nelim + lane) * DISC_LD + nelim + j];
  }
#undef DSYNC
}

// The same for a front beyond a wavefront (up to DISC_WIDE_MAX rows with its incoming delays): one workgroup, the work
// matrix full symmetric and column-major in a global scratch area (few such fronts: the top of a saddle-point tree, or
// what earlier order repairs have gathered at a root), thread = row.  Same rules, same outputs.
__global__ void __launch_bounds__(256)
k_front_discover_wg(const DiscTask* __restrict__ tasks, const int32_t* __restrict__ list, int cnt,
                    const int32_t* __restrict__ clist, const int32_t* __restrict__ cmap, const int32_t* __restrict__ invp,
                    const int64_t* __restrict__ asrc, const uint32_t* __restrict__ arc, const double* __restrict__ val,
                    double* __restrict__ arena, double* __restrict__ scratch, int32_t* __restrict__ ddelay,
                    int32_t* __restrict__ dvar, int32_t* __restrict__ pseq, uint8_t* __restrict__ ptwo,
                    int32_t* __restrict__ pcnt, int32_t* __restrict__ flags, double small, double u, int nn) {
  __shared__ double redv[8];
  __shared__ int redi[4];
  __shared__ int ids[DISC_WIDE_MAX], two[DISC_WIDE_MAX], mapi[DISC_WIDE_MAX];
  const int tid = threadIdx.x;
  const int s = list[blockIdx.x];
  const DiscTask t = tasks[s];
  int din = 0;
  for (int ci = 0; ci < t.ccnt; ++ci) din += ddelay[clist[t.cbeg + ci]];
  const int M = din + t.m, nc = din + t.n;
  if (M > t.mcap) {
    if (tid == 0) { atomicOr(&flags[0], 1); ddelay[s] = 0; pcnt[s] = 0; }
    return;
  }
  const int64_t ld = t.mcap;
  double* W = scratch + t.woff;                     // W(r, c) = W[c * ld + r]
  for (int c = 0; c < M; ++c)
    for (int r = tid; r < M; r += 256) W[c * ld + r] = 0.0;
  for (int i = tid; i < M; i += 256) {
    ids[i] = (i >= din && i < nc) ? invp[t.sptr + i - din] : -1;
    two[i] = 0;
  }
  tpp_sync();
  for (int e = tid; e < t.acnt; e += 256) {
    const uint32_t rc = arc[t.a0 + e];
    const int r = din + int(rc & 0xffffu), c = din + int(rc >> 16);
    const double v = val[asrc[t.a0 + e]];
    W[c * ld + r] = v;
    W[r * ld + c] = v;
  }
  tpp_sync();
  {
    int base = 0;
    for (int ci = 0; ci < t.ccnt; ++ci) {
      const int c = clist[t.cbeg + ci];
      const DiscTask tc = tasks[c];
      const int dc = ddelay[c];
      const int q = dc + tc.m - tc.n;
      for (int i = tid; i < q; i += 256) {
        if (i < dc) ids[base + i] = dvar[tc.voff + i];
        mapi[i] = (i < dc) ? base + i : din + cmap[tc.moff + i - dc];
      }
      tpp_sync();
      const double* Cb = arena + tc.coff;
      for (int64_t e = tid; e < int64_t(q) * q; e += 256) {
        const int i = int(e / q), j = int(e % q);
        W[int64_t(mapi[j]) * ld + mapi[i]] += Cb[int64_t(i) * tc.qcap + j];   // (one thread per target: the map is injective)
      }
      base += dc;
      tpp_sync();
    }
  }
  auto colmax = [&](int col, int from, int x1, int x2) -> double {
    double v = 0.0;
    for (int i = from + tid; i < M; i += 256)
      if (i != x1 && i != x2) v = fmax(v, fabs(W[col * ld + i]));
    return tpp_max(v, redv);
  };
  auto swap_sym = [&](int x, int y) {
    if (x == y) return;
    for (int r = tid; r < M; r += 256) {
      const double a = W[x * ld + r], b = W[y * ld + r];
      W[x * ld + r] = b;
      W[y * ld + r] = a;
    }
    tpp_sync();
    for (int c = tid; c < M; c += 256) {
      const double a = W[c * ld + x], b = W[c * ld + y];
      W[c * ld + x] = b;
      W[c * ld + y] = a;
    }
    if (tid == 0) { const int v = ids[x]; ids[x] = ids[y]; ids[y] = v; }
    tpp_sync();
  };
  auto pivot_1x1 = [&](int p) {
    const double d = 1.0 / W[p * ld + p];
    for (int c = p + 1; c < M; ++c) {
      const double acp = W[p * ld + c];
      for (int r = p + 1 + tid; r < M; r += 256) W[c * ld + r] -= (W[p * ld + r] * acp) * d;
    }
    tpp_sync();
  };
  auto pivot_2x2 = [&](int p, double d11, double d21, double d22) {
    for (int c = p + 2; c < M; ++c) {
      const double a1c = W[p * ld + c], a2c = W[(p + 1) * ld + c];
      for (int r = p + 2 + tid; r < M; r += 256) {
        const double a1r = W[p * ld + r], a2r = W[(p + 1) * ld + r];
        W[c * ld + r] -= d11 * (a1r * a1c) + d21 * (a2r * a1c + a1r * a2c) + d22 * (a2r * a2c);
      }
    }
    tpp_sync();
  };
  int p = 0;
  while (p < nc) {
    if (colmax(p, p, -1, -1) < small) { ++p; continue; }
    bool found = false;
    for (int q = p + 1; q < nc && !found; ++q) {
      double mall = 0.0, mv = -1.0;
      int mi = INT_MAX;
      for (int i = p + tid; i < M; i += 256) {
        const double a = fabs(W[q * ld + i]);
        mall = fmax(mall, a);
        if (i < q && (a > mv || (a == mv && i < mi))) { mv = a; mi = i; }
      }
      mall = tpp_max(mall, redv);
      if (mall < small) {
        swap_sym(p, q);
        ++p;
        found = true;
        break;
      }
      tpp_argmax(mv, mi, redv + 4, redi);
      const int tt = mi;
      const double maxt = colmax(tt, p, tt, q);
      double maxq = colmax(q, p, q, tt);
      const double a11 = W[tt * ld + tt], a21 = W[tt * ld + q], a22 = W[q * ld + q];
      bool ok2 = false;
      double d11 = 0.0, d21 = 0.0, d22 = 0.0;
      const double maxpiv = fmax(fabs(a11), fmax(fabs(a21), fabs(a22)));
      if (maxpiv >= small) {
        const double detscale = 1.0 / maxpiv;
        const double detpiv0 = (a11 * detscale) * a22, detpiv1 = (a21 * detscale) * a21;
        const double detpiv = detpiv0 - detpiv1;
        if (!(fabs(detpiv) < fmax(small, fmax(fabs(detpiv0 / 2), fabs(detpiv1 / 2))))) {
          d11 = (a22 * detscale) / detpiv;
          d21 = (-a21 * detscale) / detpiv;
          d22 = (a11 * detscale) / detpiv;
          if (fmax(maxq, maxt) < small) ok2 = true;
          else {
            const double x1 = fabs(d11) * maxt + fabs(d21) * maxq;
            const double x2 = fabs(d21) * maxt + fabs(d22) * maxq;
            ok2 = (u * fmax(x1, x2) < 1.0);
          }
        }
      }
      if (ok2) {
        swap_sym(tt, p);
        swap_sym(q, p + 1);
        pivot_2x2(p, d11, d21, d22);
        if (tid == 0) two[p] = 1;
        p += 2;
        found = true;
        break;
      }
      maxq = fmax(maxq, fabs(a21));
      if (fabs(a22) >= u * maxq && fabs(a22) >= small) {
        swap_sym(q, p);
        pivot_1x1(p);
        p += 1;
        found = true;
        break;
      }
    }
    if (found) continue;
    const double maxp = colmax(p, p, p, -1);
    const double app = W[p * ld + p];
    if (fabs(app) >= u * maxp && fabs(app) >= small) {
      pivot_1x1(p);
      ++p;
    } else {
      break;
    }
  }
  tpp_sync();
  const bool root = t.parent >= nn;
  const int nelim = p;
  const int count = root ? nc : nelim;
  const int dout = root ? 0 : nc - nelim;
  if (dout > t.dcap) {
    if (tid == 0) { atomicOr(&flags[0], 2); ddelay[s] = 0; pcnt[s] = 0; }
    return;
  }
  for (int i = tid; i < count; i += 256) {
    pseq[t.poff + i] = ids[i];
    ptwo[t.poff + i] = uint8_t(two[i]);
  }
  for (int i = tid; i < dout; i += 256) dvar[t.voff + i] = ids[nelim + i];
  if (tid == 0) {
    pcnt[s] = count;
    ddelay[s] = dout;
    if (dout > 0) atomicAdd(&flags[1], dout);
  }
  if (!root) {
    const int q = dout + t.m - t.n;
    double* Cb = arena + t.coff;
    for (int64_t e = tid; e < int64_t(q) * q; e += 256) {
      const int i = int(e / q), j = int(e % q);
      Cb[int64_t(i) * t.qcap + j] = W[int64_t(nelim + j) * ld + nelim + i];
    }
  }
}

// gvar[slot] = variable eliminated at pivot slot `slot` (after the numerical pivoting of this factorization)
__global__ void k_gvar(int n, const int32_t* __restrict__ gperm, const int32_t* __restrict__ invp,
                       int32_t* __restrict__ gvar) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gvar[i] = invp[gperm[i]];
}

// rectangle -> the two packed images, for the wave-tier fronts that a workgroup kernel factorized (first
// factorizations, blacklisted fronts, fronts of more than TINY_N columns); k_front_wave writes the images itself.
// mode 0: every task; mode 1: only those k_front_wave did not do in this pass.  One wave per front.
__global__ void __launch_bounds__(256)
k_wpack(const WTask* __restrict__ tasks, const WPack* __restrict__ packs, int ntask, int mode,
        const uint8_t* __restrict__ tinyskip, const double* __restrict__ L, double* __restrict__ Lf,
        double* __restrict__ Lb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = blockIdx.x * 4 + wave;
  if (ti >= ntask) return;
  const WTask t = wave_task(tasks, ti);
  const WPack pk = packs[__builtin_amdgcn_readfirstlane(ti)];
  const int m = t.m, n = t.n;
  if (mode == 1 && n <= TINY_N && !tinyskip[pk.node]) return;
  const int ld = (m + 1) & ~1;
  const double* A = L + pk.loff;
  double* f = Lf + t.lfoff;
  const int r = lane;
  if (r >= m) return;
  for (int k = 0; k < n && k < r; ++k) {
    const double v = A[int64_t(k) * ld + r];
    const int j = k >> 1;
    f[2 * (wf_pair_off(j, m) + r - (2 * j + 1)) + (k & 1)] = v;
  }
}

// The same front kernel BLOCKED through LDS: the front stays in its LDS triangle, four pivot columns at a time are taken
// into registers (lane = row), factorized there exactly as k_front_wave does it (given order, every pivot and multiplier
// tested, v_readlane broadcasts -- but only inside the 4-column panel), written out, and the rank-4 update of the
// trailing triangle is done with a lane per ENTRY: all 64 lanes work whatever the front's height, nothing is computed
// above the diagonal, and the loops are rolled -- one instantiation of ~4 KB serves every front of up to 64 columns
// (no width classes, no cold pass through tens of KB of unrolled code on the levels with a handful of fronts).
// After the last panel the trailing triangle IS the contribution block.
template <int WPB>
__global__ void __launch_bounds__(64 * WPB)
k_front_blk(const TinyFrontTask* __restrict__ tasks, int ntask, const GatherLists g,
            const int64_t* __restrict__ asrc, const int32_t* __restrict__ aloc,
            const double* __restrict__ val, double* __restrict__ L, double* __restrict__ D, double* __restrict__ C,
            int32_t* __restrict__ stat, int32_t* __restrict__ fastok, const uint8_t* __restrict__ hint,
            const uint8_t* __restrict__ tinyskip, int32_t* __restrict__ tinyfail, double small, double u,
            double* __restrict__ Lf, double* __restrict__ Lbk, int tri, int mode) {
  typedef double double2_t __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) double fsh[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
  if (ti >= ntask) return;
  const TinyFrontTask t = tasks[ti];
  if (tinyskip[t.node]) return;
  const int n = t.n, m = t.m, cm = m - n;
  // learned 2x2 pivots: bit c = columns (c, c+1) are eliminated together
  const bool h2 = (lane < n) ? (hint[t.sptr + lane] != 0) : false;
  const unsigned long long hm = __ballot(h2);
  // mode 1: the pass behind the unrolled kernels -- only the fronts of at most 36 columns they left alone (hinted ones)
  if (mode == 1 && (n > 36 || hm == 0ull)) return;
  double* Fr = fsh + wave * (tri + 512);        // the triangle, then the panel scratch: L[row][4], (L D)[row][4]
  double2_t* P = reinterpret_cast<double2_t*>(Fr + tri);
  double2_t* PD = P + 128;
  front_assemble(t, Fr, lane, g, asrc, aloc, val, C);
  const double inv_u = (u > 0.0) ? 1.0 / u : INFINITY;
  bool bad = false;
  int nneg = 0, ntwo = 0;
  double myd0 = 0.0, myd1 = 0.0;                // lane c: D entries of pivot c (inverted; 2x2: [d11, d21], [inf, d22])
  const bool in = lane < m;
  // entry e = lane + 64 p of a lower triangle stored row by row (independent of its order): row i, column j
  auto decode = [](int e, int& i, int& j) {
    i = int((sqrtf(float(8 * e + 1)) - 1.0f) * 0.5f);
    i += (((i + 1) * (i + 2)) >> 1 <= e) ? 1 : 0;
    i -= ((i * (i + 1)) >> 1 > e) ? 1 : 0;
    j = e - ((i * (i + 1)) >> 1);
  };
  const bool images = Lf && t.lfoff >= 0;
  double* fimg = Lf + (images ? t.lfoff : 0);
  double* Lb = L + t.loff;
  int w = 0;
  for (int c0 = 0; c0 < n; c0 += w) {
    w = min(4, n - c0);
    if (w == 4 && ((hm >> (c0 + 3)) & 1ull)) w = 3;          // a 2x2 pivot must not straddle two panels
    // ---- the panel: columns c0 .. c0+w-1, row `lane` -------------------------------------------------------
    double a[4], um[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + q;
      a[q] = (q < w && in && lane >= c) ? Fr[c * m - ((c * (c + 1)) >> 1) + lane] : 0.0;
      um[q] = 0.0;
    }
    bool second = false;                // column q is the second of a 2x2 pivot (done with the first)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= w) continue;             // uniform
      if (second) { second = false; continue; }
      const int c = c0 + q;
      const bool want2 = ((hm >> c) & 1ull) != 0ull;
      if (want2 && q + 1 >= w) bad = true;             // (a hint on the last column of the front: no partner)
      if (want2 && q + 1 < w && q < 3) {
        const double a11 = readlane_f64(a[q], c), a21 = readlane_f64(a[q], c + 1);
        const double a22 = readlane_f64(a[q < 3 ? q + 1 : 3], c + 1);
        // a hinted pair was chosen by one of the pivoting kernels before: accepted unless its determinant cancels
        // (test_2x2 of ldlt_tpp.cxx:99-118); the multipliers are tested like every other pivot's
        const double maxpiv = fmax(fabs(a11), fmax(fabs(a21), fabs(a22)));
        if (!(maxpiv >= small)) bad = true;
        const double detscale = 1.0 / maxpiv;
        const double detpiv0 = (a11 * detscale) * a22, detpiv1 = (a21 * detscale) * a21;
        const double detpiv = detpiv0 - detpiv1;
        if (!(fabs(detpiv) >= fmax(small, fmax(fabs(detpiv0 / 2), fabs(detpiv1 / 2))))) bad = true;
        const double d11 = (a22 * detscale) / detpiv, d22 = (a11 * detscale) / detpiv;
        const double d21 = (-a21 * detscale) / detpiv;
        const double u1 = a[q], u2 = a[q < 3 ? q + 1 : 3];
        const double own1 = d11 * u1 + d21 * u2, own2 = d21 * u1 + d22 * u2;
        if (in && lane > c + 1 && !(fabs(own1) <= inv_u && fabs(own2) <= inv_u)) bad = true;
#pragma unroll
        for (int q2 = 2; q2 < 4; ++q2)
          if (q2 > q + 1 && q2 < w)
            a[q2] = fma(-own1, readlane_f64(u1, c0 + q2), fma(-own2, readlane_f64(u2, c0 + q2), a[q2]));
        um[q] = u1;
        um[q < 3 ? q + 1 : 3] = u2;
        a[q] = (lane == c) ? 1.0 : ((lane == c + 1) ? 0.0 : own1);
        a[q < 3 ? q + 1 : 3] = (lane == c + 1) ? 1.0 : own2;
        if (lane == c) { myd0 = d11; myd1 = d21; }
        if (lane == c + 1) { myd0 = INFINITY; myd1 = d22; }
        const double det = a11 * a22 - a21 * a21;
        if (det < 0.0) nneg += 1;
        else if (a11 + a22 < 0.0) nneg += 2;
        ++ntwo;
        second = true;
        continue;
      }
      const double d = readlane_f64(a[q], c);
      if (!(fabs(d) >= small)) bad = true;
      if (d < 0.0) ++nneg;
      double rd = __builtin_amdgcn_rcp(d);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      um[q] = a[q];
      const double own = um[q] * rd;
      if (in && lane > c && !(fabs(own) <= inv_u)) bad = true;
#pragma unroll
      for (int q2 = q + 1; q2 < 4; ++q2)
        if (q2 < w) a[q2] = fma(-own, readlane_f64(um[q], c0 + q2), a[q2]);
      a[q] = (lane == c) ? 1.0 : own;
      if (lane == c) { myd0 = rd; myd1 = 0.0; }
    }
    // ---- the panel out: images (or rectangle), and scratch for the trailing update ------------------------------
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = c0 + q;
      if (q < w) {                      // uniform
        if (images) {
          if (in && lane > k) {
            const int j2 = k >> 1;
            fimg[2 * (wf_pair_off(j2, m) + lane - (2 * j2 + 1)) + (k & 1)] = a[q];
          }
        } else if (in && lane >= k) {
          Lb[int64_t(k) * t.ld + lane] = a[q];
        }
      }
    }
    const int tb = c0 + w, mt = m - tb;            // the trailing triangle: rows / columns tb .. m-1
    if (mt <= 0) break;
    if (in) {                                        // L and L D (= the columns before they were scaled) of the panel
      P[2 * lane] = double2_t{a[0], a[1]};
      P[2 * lane + 1] = double2_t{a[2], a[3]};
      PD[2 * lane] = double2_t{um[0], um[1]};
      PD[2 * lane + 1] = double2_t{um[2], um[3]};
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // ---- F(i, c) -= sum_q L(i, c0+q) (L D)(c, c0+q) over the trailing triangle: a lane per 2 x 2 TILE of entries
    //      (rows i, i+1 x columns c, c+1: four panel rows read for four entries)
    const int T = (mt + 1) >> 1, ntile = (T * (T + 1)) >> 1;
    for (int e0 = 0; e0 < ntile; e0 += 64) {
      int I, J;
      decode(min(e0 + lane, ntile - 1), I, J);
      const bool on = e0 + lane < ntile;
      const int i = tb + 2 * I, c = tb + 2 * J;
      const bool r1 = i + 1 < m;                     // second row exists
      const bool v01 = I > J;                        // (i, c+1) lies below the diagonal
      const int ib = r1 ? i + 1 : i, cb = (c + 1 < m) ? c + 1 : c;
      const double2_t a0 = P[2 * i], a1 = P[2 * i + 1], b0 = P[2 * ib], b1 = P[2 * ib + 1];
      const double2_t x0 = PD[2 * c], x1 = PD[2 * c + 1], y0 = PD[2 * cb], y1 = PD[2 * cb + 1];
      const int oc = c * m - ((c * (c + 1)) >> 1), od = cb * m - ((cb * (cb + 1)) >> 1);
      const int o00 = oc + i, o10 = oc + ib, o01 = v01 ? od + i : o00, o11 = od + ib;
      double f00 = Fr[o00], f10 = Fr[o10], f01 = Fr[o01], f11 = Fr[o11];
      f00 = fma(-a0.x, x0.x, f00); f00 = fma(-a0.y, x0.y, f00); f00 = fma(-a1.x, x1.x, f00); f00 = fma(-a1.y, x1.y, f00);
      f10 = fma(-b0.x, x0.x, f10); f10 = fma(-b0.y, x0.y, f10); f10 = fma(-b1.x, x1.x, f10); f10 = fma(-b1.y, x1.y, f10);
      f01 = fma(-a0.x, y0.x, f01); f01 = fma(-a0.y, y0.y, f01); f01 = fma(-a1.x, y1.x, f01); f01 = fma(-a1.y, y1.y, f01);
      f11 = fma(-b0.x, y0.x, f11); f11 = fma(-b0.y, y0.y, f11); f11 = fma(-b1.x, y1.x, f11); f11 = fma(-b1.y, y1.y, f11);
      if (on) {
        Fr[o00] = f00;
        if (r1) Fr[o10] = f10;
        if (v01) Fr[o01] = f01;
        if (r1) Fr[o11] = f11;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  if (__ballot(bad) != 0ull) {
    if (lane == 0) {
      const int slot = atomicAdd(&stat[13], 1);
      if (slot < FAILCAP) tinyfail[slot] = t.node;
      fastok[t.iblk] = 0;
    }
    return;
  }
  if (lane < n) {
    D[2 * int64_t(t.sptr + lane)] = myd0;
    D[2 * int64_t(t.sptr + lane) + 1] = myd1;
  }
  if (lane == 0) {
    fastok[t.iblk] = 1;
    atomicAdd(&stat[16 + STAT_BINS + (ti & (STAT_BINS - 1))], 1);
    if (nneg) atomicAdd(&stat[16 + (ti & (STAT_BINS - 1))], nneg);
    if (ntwo) atomicAdd(&stat[3], ntwo);
  }
  if (!t.has_contrib || cm <= 0) return;
  double* Cb = C + t.coff;
  const int nent = (cm * (cm + 1)) >> 1;
  for (int e0 = 0; e0 < nent; e0 += 64) {
    int i2, j2;
    decode(min(e0 + lane, nent - 1), i2, j2);
    const int c = n + j2;
    if (e0 + lane < nent) Cb[int64_t(j2) * cm + i2] = Fr[c * m - ((c * (c + 1)) >> 1) + n + i2];
  }
}

// D solve restricted to the pivots of the fronts in `list` (one workgroup each): the fronts the wave tier does not
// cover, when it applies D to its own fronts inside the forward step
__global__ void __launch_bounds__(256)
k_solve_diag_nodes(const NodeDesc* __restrict__ nodes, const int32_t* __restrict__ list, const double* __restrict__ D,
                   const int32_t* __restrict__ gperm, double* __restrict__ xp, Cols cs) {
  GSLS_COLS;
  xp += col_ * cs.sx;
  const NodeDesc nd = nodes[list[bid]];
  for (int k = threadIdx.x; k < nd.n; k += 256) {
    const int64_t i = int64_t(nd.sptr) + k;
    const double d0 = D[2 * i];
    if (isinf(d0)) continue;
    const int gi = gperm[i];
    if (isinf(D[2 * i + 2])) {
      const int gj = gperm[i + 1];
      const double d21 = D[2 * i + 1], d22 = D[2 * i + 3];
      const double x1 = xp[gi], x2 = xp[gj];
      xp[gi] = fma(d0, x1, d21 * x2);
      xp[gj] = fma(d21, x1, d22 * x2);
    } else {
      xp[gi] *= d0;
    }
  }
}

// =================================================================================================
// multi-GPU exchange helpers: pack / unpack the cut roots' blocks, merge / mask solution vectors
// =================================================================================================
struct Segment {
  int64_t src, dst, len;   // element offsets in the arena and in the exchange buffer
  int32_t owner, pad;
};
// dir 0: buf[dst..] = (owner == me ? arena[src..] : 0) ; dir 1: arena[src..] = buf[dst..]
__global__ void k_segments(const Segment* __restrict__ seg, double* __restrict__ arena,
                           double* __restrict__ buf, int dir, int me) {
  const Segment sg = seg[blockIdx.y];
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < sg.len;
       i += int64_t(gridDim.x) * blockDim.x) {
    if (dir == 0) buf[sg.dst + i] = (sg.owner == me) ? arena[sg.src + i] : 0.0;
    else arena[sg.src + i] = buf[sg.dst + i];
  }
}
// mode 0: xp[i] = xq[i] where the position belongs to the top part
// mode 1: xp[i] = 0 where this rank did not compute position i (top counts for rank 0)
__global__ void k_xmask(int n, const int32_t* __restrict__ posowner, double* __restrict__ xp,
                        const double* __restrict__ xq, int mode, int me) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = posowner[i];
  if (mode == 0) { if (o < 0) xp[i] = xq[i]; }
  else if (!(o == me || (o < 0 && me == 0))) xp[i] = 0.0;
}
// z-vectors of the cut roots: the ancestors' part of the solution that the subtree below a cut root needs --
// rows n..m-1 of the root's front, which (elimination-tree property) cover every top-part position any front of that
// subtree refers to.  dir 0 (rank 0, after the top part's backward sweep): buf[dst + i] = xp[rlist[src + i]] for every
// cut root; dir 1 (owner): xp[rlist[src + i]] = buf[dst + i] for the roots this rank owns.
__global__ void k_zvec(const Segment* __restrict__ seg, const int32_t* __restrict__ rlist, double* __restrict__ xp,
                       double* __restrict__ buf, int dir, int me) {
  const Segment sg = seg[blockIdx.y];
  if (dir == 1 && sg.owner != me) return;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < sg.len; i += int64_t(gridDim.x) * blockDim.x) {
    if (dir == 0) buf[sg.dst + i] = xp[rlist[sg.src + i]];
    else xp[rlist[sg.src + i]] = buf[sg.dst + i];
  }
}
// x[var] = xp[pos] for the positions this rank computed (its subtrees; rank 0 also the top part)
__global__ void k_permute_out_owned(int n, const int32_t* __restrict__ invp, const int32_t* __restrict__ posowner,
                                    int me, const double* __restrict__ xp, double* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = posowner[i];
  if (o == me || (o < 0 && me == 0)) x[invp[i]] = xp[i];
}
// the factorization counters of this rank as doubles behind the contribution blocks of the exchange buffer, so that
// ONE reduction carries data and status: [0] ranks that met a non-positive pivot (posdef), [1] failed columns
// (delays), [2] negative pivots, [3] 2x2 pivots, [4] zero pivots, [5] fronts the wave-per-front kernels gave up on
// (fast path: the factorization has to be repeated without it), [6] blocks / fronts that needed pivoting
__global__ void k_stat_to_xchg(const int32_t* __restrict__ stat, double* __restrict__ out,
                               const double* __restrict__ minus) {     // minus: an earlier snapshot to subtract, or null
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int neg = stat[2];
  for (int b = 0; b < STAT_BINS; ++b) neg += stat[16 + b];
  double v[8] = {(stat[0] != INT_MAX) ? 1.0 : 0.0, double(stat[4]), double(neg), double(stat[3]), double(stat[1]),
                 double(stat[13]), double(stat[7] + stat[14]), 0.0};   // [5] fronts the wave kernels gave up on, [6] pivoted
  for (int k = 0; k < 8; ++k) out[k] = v[k] - (minus ? minus[k] : 0.0);
  if (minus && out[0] < 0.0) out[0] = 0.0;
}

// D solve restricted to the positions of one owner class (sel = rank, or -1 for the top part)
__global__ void k_solve_diag_owned(int n, const double* __restrict__ D, const int32_t* __restrict__ gperm,
                                   const int32_t* __restrict__ posowner, int sel, double* __restrict__ xp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || posowner[i] != sel) return;
  const double d0 = D[2 * int64_t(i)];
  if (isinf(d0)) return;
  const int gi = gperm[i];
  if (i + 1 < n && isinf(D[2 * int64_t(i) + 2])) {
    const int gj = gperm[i + 1];
    const double d21 = D[2 * int64_t(i) + 1], d22 = D[2 * int64_t(i) + 3];
    const double x1 = xp[gi], x2 = xp[gj];
    xp[gi] = fma(d0, x1, d21 * x2);
    xp[gj] = fma(d21, x1, d22 * x2);
  } else {
    xp[gi] *= d0;
  }
}

__global__ void k_iota(int n, int32_t* __restrict__ a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}

// =================================================================================================
// host side: upload of the symbolic data and the per-level launch plan
// =================================================================================================
// ---- the handle's block pool (DevPool, gsls_device.hpp) -------------------------------------------------------------------
DevPool::~DevPool() {
  for (auto& kv : idle) (void)hipFree(kv.second);
}
static thread_local DevPool* tl_pool = nullptr;      // the pool of the handle whose arrays are being (re)built
struct PoolScope {
  DevPool* prev;
  explicit PoolScope(DeviceFactor& F) : prev(tl_pool) {
    static const bool off = getenv("GSLS_NO_POOL") != nullptr;     // (A/B knob)
    if (!F.pool) F.pool = std::make_shared<DevPool>();
    tl_pool = off ? nullptr : F.pool.get();
  }
  ~PoolScope() { tl_pool = prev; }
};
static size_t pool_class(size_t bytes) {             // next of 8 steps per power of two (<= 12.5 % head-room), >= 512 B
  size_t c = 512;
  while (c < bytes) c <<= 1;
  if (c > 512) {
    const size_t step = c >> 4;                       // c / 2 < bytes <= c: classes c/2 + k * c/16
    c = (c >> 1) + ((bytes - (c >> 1) + step - 1) / step) * step;
  }
  return c;
}
static hipError_t pool_alloc(void** p, size_t bytes) {
  DevPool* pool = tl_pool;
  if (!pool || bytes > (size_t(256) << 20)) return hipMalloc(p, bytes);     // (large blocks: exact size, never kept)
  const size_t c = pool_class(bytes);
  auto it = pool->idle.find(c);
  if (it != pool->idle.end()) {
    *p = it->second;
    pool->idle.erase(it);
    pool->idle_bytes -= c;
    return hipSuccess;
  }
  hipError_t e = hipMalloc(p, c);
  if (e != hipSuccess && !pool->idle.empty()) {       // out of memory with blocks waiting: give them back and try again
    for (auto& kv : pool->idle) { (void)hipFree(kv.second); pool->size.erase(kv.second); }
    pool->idle.clear();
    pool->idle_bytes = 0;
    e = hipMalloc(p, c);
  }
  if (e == hipSuccess) pool->size[*p] = c;
  return e;
}
// blocks the pool does not know (plain hipMalloc elsewhere) are simply freed; large ones are not kept waiting
static void pool_free(DeviceFactor& F, void* p) {
  if (!p) return;
  DevPool* pool = F.pool.get();
  if (pool) {
    auto it = pool->size.find(p);
    if (it != pool->size.end()) {
      const size_t c = it->second;
      if (c <= (size_t(256) << 20) && pool->idle_bytes + c <= (size_t(2) << 30)) {
        pool->idle.emplace(c, p);
        pool->idle_bytes += c;
        return;
      }
      pool->size.erase(it);
    }
  }
  (void)hipFree(p);
}

template <class T>
static hipError_t upload(T*& dptr, const std::vector<T>& h, hipStream_t st) {
  dptr = nullptr;
  if (h.empty()) return hipSuccess;
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&dptr), h.size() * sizeof(T)));
  return hipMemcpyAsync(dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st);
}

#ifdef GSLS_STAMPS
extern "C" void gsls_debug_stamps(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64);
}
#endif

void dev_free(DeviceFactor& F) {
  void* ptrs[] = {F.nodes, F.rlist, F.cmap, F.clist, F.lvlnodes, F.pullsegs, F.tinyctasks, F.tftasks, F.asrc, F.adst, F.arow,
                  F.acol, F.ptasks, F.ttasks, F.invp, F.L, F.C, F.D, F.val, F.scale, F.xp, F.cvec, F.xp_mr, F.cvec_mr,
                  F.xhost, F.stat, F.gperm, F.pulltasks, F.faillist, F.smallnodes, F.bignodes, F.bigtrsv,
                  F.biggemv, F.ybuf, F.part, F.Linv, F.stasks, F.gth_ptr, F.gth_src, F.fastok, F.hint, F.tinyskip, F.tinyfail, F.bl_ptasks, F.bl_ttasks, F.bl_tctasks, F.segC, F.segV, F.segZ, F.posowner, F.tppflag, F.tpplist,
                  F.cztasks, F.gdst, F.gbeg, F.gsrc, F.aloc, F.asrc_wg, F.adst_wg, F.bl_pullsegs, F.bl_pulltasks, F.wtasks, F.wgroups, F.wpacks, F.wgth_ptr, F.wgth_src, F.wpull2, F.wnont, F.Lf, F.Lb, F.xs, F.gvar,
                  F.disc_tasks, F.disc_arc, F.disc_ddelay, F.disc_dvar, F.disc_pseq, F.disc_pcnt, F.disc_flags, F.disc_ptwo, F.disc_arena,
                  F.disc_scratch, F.disc_list, F.wperm_list};
  for (void* p : ptrs) pool_free(F, p);
  for (void* p : {static_cast<void*>(F.mc_xp), static_cast<void*>(F.mc_xs), static_cast<void*>(F.mc_cvec),
                  static_cast<void*>(F.mc_ybuf), static_cast<void*>(F.mc_part)})
    pool_free(F, p);
  // the caller's matrix (gsls_set_coo) depends on the pattern only, not on the elimination order: it survives
  // the re-analyses of order repair and learning; dev_free_coo releases it
  DeviceFactor keep;
  keep.pool = F.pool;
  keep.coo_ne = F.coo_ne; keep.coo_nz = F.coo_nz; keep.nscatter_coo = F.nscatter_coo;
  keep.mv_ptr = F.mv_ptr; keep.mv_src = F.mv_src; keep.rs_ptr = F.rs_ptr; keep.rs_col = F.rs_col;
  keep.rs_src = F.rs_src; keep.coo_val = F.coo_val; keep.valcsc = F.valcsc; keep.rbuf = F.rbuf;
  keep.rbuf_cap = F.rbuf_cap;
  F = keep;
}

static hipError_t allow_big_lds() {
  const int big = 160 * 1024 - 512;
  // the wave-per-front kernels: four packed triangles of up to 64 x 64 (+ panel scratch) per workgroup
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_blk<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_wave<24, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_wave<28, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_wave<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_wave<36, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_fwd_chol_mr<8>), hipFuncAttributeMaxDynamicSharedMemorySize, int(MR_LDS_CAP)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_fwd_chol_mr<4>), hipFuncAttributeMaxDynamicSharedMemorySize, int(MR_LDS_CAP)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_fwd_chol_mr<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(MR_LDS_CAP)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_chol_mr<8>), hipFuncAttributeMaxDynamicSharedMemorySize, int(MR_LDS_CAP)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_chol_mr<4>), hipFuncAttributeMaxDynamicSharedMemorySize, int(MR_LDS_CAP)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_chol_mr<2>), hipFuncAttributeMaxDynamicSharedMemorySize, int(MR_LDS_CAP)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diag_fast<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diag_fast<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diag_fast<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel_chol), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_diag_ldlt), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_fwd_chol), hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd_chol), hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_solve_bwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, big));
  // the wide backward launches of the wave tier stage one forward image per wave (up to 16 KB) beside their static LDS
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_bwd<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_bwd<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_fwd<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_fwd<false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front_discover), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * DISC_LD * 8));
  // (column groups: CG sets of LDS accumulators per wave, dynamic, next to 24 KB of static image staging)
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_fwd<true, true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_fwd<true, true, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_fwd<false, true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_fwd<false, true, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_bwd<true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_bwd<true, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_tail<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wsolve_tail<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  return hipSuccess;
}

hipError_t dev_upload_symbolic(const Symbolic& S, DeviceFactor& F, hipStream_t st) {
  PoolScope pool_scope(F);
  F.pure_state = false;      // (gperm, D, gvar are rebuilt: dev_factor)
  const int me = F.myrank;
  dev_free(F);
  F.myrank = me;
  HIPCHK(allow_big_lds());
  const int nn = S.nnodes;
  std::vector<NodeDesc> nd(nn);
  int64_t nblk64 = 0;   // 64-column blocks, numbered front by front: slots of the L11^-T arena
  for (int s = 0; s < nn; ++s) {
    NodeDesc& d = nd[s];
    d.m = S.nrow(s);
    d.n = S.ncol(s);
    d.ld = S.ldl[s];
    d.sptr = S.sptr[s];
    d.cbeg = S.cptr[s];
    d.cend = S.cptr[s + 1];
    d.parent = S.sparent[s];
    d.iblk = int32_t(nblk64);
    nblk64 += (S.ncol(s) + NB - 1) / NB;
    d.loff = S.loff[s];
    d.coff = S.coff[s];
    d.roff = S.rptr[s];
    d.moff = S.cmapptr[s];
  }
  // A -> L scatter lists with absolute destinations
  const int64_t nz = S.nptr[nn];
  std::vector<int64_t> asrc(nz), adst(nz);
  std::vector<int32_t> arow(nz), acol(nz), aloc(nz);
  // (a sharded run lays out only the fronts this rank owns, shard_layout: the others' entries are not scattered)
  const bool rank_layout = !S.owner.empty() && S.nranks > 1 && int(S.czptr.size()) == 2 * S.nlevels + 1;
  for (int s = 0; s < nn; ++s) {
    const int64_t m = S.nrow(s);
    const bool here = !rank_layout || S.owner[s] == F.myrank || (S.owner[s] < 0 && F.myrank == 0);
    for (int64_t k = S.nptr[s]; k < S.nptr[s + 1]; ++k) {
      const int64_t dst = S.nlist[2 * k + 1];
      const int64_t c = dst / m, r = dst % m;
      asrc[k] = S.nlist[2 * k];
      adst[k] = here ? S.loff[s] + c * S.ldl[s] + r : -1;
      aloc[k] = (m <= 64) ? int32_t(c * m - c * (c + 1) / 2 + r) : 0;     // k_front_wave's packed triangle in LDS
      arow[k] = S.rlist[S.rptr[s] + r];
      acol[k] = S.sptr[s] + int(c);
    }
  }
  // launch plan
  std::vector<PanelTask> pt;
  std::vector<TileTask> tt;
  std::vector<PullSeg> psg;
  std::vector<TinyContribTask> tct;
  std::vector<TinyFrontTask> tft;
  std::vector<uint32_t> fgdst;      // k_front_wave: extend-add gather lists
  std::vector<int32_t> fgbeg;
  std::vector<int64_t> fgsrc;
  std::vector<std::pair<int32_t, int64_t>> gtmp;
  std::vector<PullTask> ptk;
  std::vector<int32_t> smalln, bign;
  std::vector<BigTrsv> btr;
  std::vector<BigGemv> bgm;
  int64_t part_max = 0;
  // one plan per node subset: everything (single device), my subtrees, the top part (multi-GPU)
  // wg(s): the front is assembled and factorized by the workgroup kernels (else: by k_front_wave, a wave per front)
  auto build_plan = [&](std::vector<LevelPlan>& plan, auto keep, auto wg) {
  plan.assign(S.nlevels, LevelPlan());
  std::vector<int> lvl_nodes;
  for (int l = 0; l < S.nlevels; ++l) {
    LevelPlan& lp = plan[l];
    lvl_nodes.clear();
    for (int i = S.lvlptr[l]; i < S.lvlptr[l + 1]; ++i)
      if (keep(S.lvlnodes[i])) lvl_nodes.push_back(S.lvlnodes[i]);
    lp.node_begin = 0;
    lp.node_end = int(lvl_nodes.size());
    int maxsteps = 0;
    for (int i = lp.node_begin; i < lp.node_end; ++i)
      maxsteps = std::max(maxsteps, (S.ncol(lvl_nodes[i]) + NB - 1) / NB);
    // per step: diag tasks first, then the extra row chunks
    lp.panel_begin.assign(2 * maxsteps, 0);
    lp.panel_cnt.assign(2 * maxsteps, 0);
    lp.panel_rows.assign(maxsteps, 0);
    for (int st_ = 0; st_ < maxsteps; ++st_) {
      lp.panel_begin[2 * st_] = int(pt.size());
      for (int i = lp.node_begin; i < lp.node_end; ++i) {
        const int s = lvl_nodes[i];
        if (S.ncol(s) > st_ * NB && wg(s)) {
          pt.push_back(PanelTask{s, st_, 0, 0});
          lp.panel_rows[st_] = std::max(lp.panel_rows[st_], std::min(PR, S.nrow(s) - st_ * NB));
        }
      }
      lp.panel_cnt[2 * st_] = int(pt.size()) - lp.panel_begin[2 * st_];
      lp.panel_begin[2 * st_ + 1] = int(pt.size());
      for (int i = lp.node_begin; i < lp.node_end; ++i) {
        const int s = lvl_nodes[i];
        if (S.ncol(s) <= st_ * NB || !wg(s)) continue;
        const int rem = S.nrow(s) - st_ * NB - PR;
        for (int c = 1; (c - 1) * RB < rem; ++c) pt.push_back(PanelTask{s, st_, c, 0});
      }
      lp.panel_cnt[2 * st_ + 1] = int(pt.size()) - lp.panel_begin[2 * st_ + 1];
    }
    lp.tile_begin = int(tt.size());
    lp.tinyc_begin = int(tct.size());
    for (int i = lp.node_begin; i < lp.node_end; ++i) {
      const int s = lvl_nodes[i];
      if (!wg(s)) continue;              // k_front_wave forms the contribution block itself
      if (S.sparent[s] >= nn) continue;  // roots have no (used) contribution block
      const int cm = S.nrow(s) - S.ncol(s);
      if (cm <= 16 && S.ncol(s) <= 64) {   // tiny: a wave per front (k_contrib_tiny)
        if (cm > 0) tct.push_back(TinyContribTask{S.ncol(s), cm, S.ldl[s], S.sptr[s], S.loff[s], S.coff[s]});
        continue;
      }
      const int nt = (cm + TS - 1) / TS;
      for (int tj = 0; tj < nt; ++tj)
        for (int ti = tj; ti < nt; ++ti) tt.push_back(TileTask{s, ti, tj, 0});
    }
    lp.tile_cnt = int(tt.size()) - lp.tile_begin;
    lp.tinyc_cnt = int(tct.size()) - lp.tinyc_begin;
    lp.tf_begin = int(tft.size());
    for (int cls = 0; cls < TINY_CLASSES; ++cls) {        // by width class: each has its own unrolled kernel
      lp.tf_cls_cnt[cls] = lp.tf_cls_maxm[cls] = 0;
      for (int i = lp.node_begin; i < lp.node_end; ++i) {
        const int s = lvl_nodes[i];
        if (wg(s) || tiny_class(S.ncol(s)) != cls) continue;
        TinyFrontTask tk{S.ncol(s), S.nrow(s), S.ldl[s], S.sptr[s], S.loff[s], S.coff[s], nd[s].iblk,
                         (S.sparent[s] < nn) ? 1 : 0, s, 0, 0, 0, int64_t(fgdst.size()), int64_t(fgsrc.size()),
                         S.nptr[s], int32_t(S.nptr[s + 1] - S.nptr[s]), 0};
        gtmp.clear();
        const int pm = S.nrow(s);
        for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) {      // children in clist order
          const int c = S.clist[ci];
          const int cmc = S.nrow(c) - S.ncol(c);
          const int32_t* mp = S.cmap.data() + S.cmapptr[c];
          for (int j = 0; j < cmc; ++j)
            for (int i = j; i < cmc; ++i)
              gtmp.emplace_back(mp[j] * pm - mp[j] * (mp[j] + 1) / 2 + mp[i], S.coff[c] + int64_t(j) * cmc + i);
        }
        std::stable_sort(gtmp.begin(), gtmp.end(),
                         [](const std::pair<int32_t, int64_t>& x, const std::pair<int32_t, int64_t>& y) { return x.first < y.first; });
        for (size_t e = 0; e < gtmp.size();) {
          size_t e2 = e;
          while (e2 < gtmp.size() && gtmp[e2].first == gtmp[e].first) ++e2;
          fgdst.push_back(uint32_t(gtmp[e].first) | (uint32_t(e2 - e) << 12));
          fgbeg.push_back(int32_t(int64_t(fgsrc.size()) - tk.s0));
          for (size_t q = e; q < e2; ++q) fgsrc.push_back(gtmp[q].second);
          e = e2;
        }
        tk.nd = int32_t(int64_t(fgdst.size()) - tk.d0);
        tft.push_back(tk);
        lp.tf_cls_maxm[cls] = std::max(lp.tf_cls_maxm[cls], S.nrow(s));
        lp.tf_cls_cnt[cls]++;
      }
    }
    lp.tf_cnt = int(tft.size()) - lp.tf_begin;
    // extend-add: one pull task per PCOLS columns of every parent, children in clist order
    lp.pull_begin = int(ptk.size());
    for (int i = lp.node_begin; i < lp.node_end; ++i) {
      const int s = lvl_nodes[i];
      if (!wg(s)) continue;                     // k_front_wave assembles its front itself
      if (S.cptr[s + 1] == S.cptr[s]) continue;
      const int pm = S.nrow(s), pn = S.ncol(s);
      for (int pc0 = 0; pc0 < pm; pc0 += PCOLS) {
        PullTask tk{int(psg.size()), 0};
        for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) {
          const int c = S.clist[ci];
          const int cmc = S.nrow(c) - S.ncol(c);
          const int32_t* mp = S.cmap.data() + S.cmapptr[c];
          const int j0 = int(std::lower_bound(mp, mp + cmc, pc0) - mp);
          const int j1 = int(std::lower_bound(mp, mp + cmc, pc0 + PCOLS) - mp);
          if (j1 > j0)
            psg.push_back(PullSeg{j0, j1, cmc, pn, S.ldl[s], pm - pn, S.cmapptr[c], S.coff[c], S.loff[s], S.coff[s]});
        }
        tk.seg_cnt = int(psg.size()) - tk.seg_begin;
        if (tk.seg_cnt > 0) ptk.push_back(tk);
      }
    }
    lp.pull_cnt = int(ptk.size()) - lp.pull_begin;
    // solve: small fronts (one workgroup each) and big fronts (blocked multi-launch path)
    lp.small_begin = int(smalln.size());
    lp.big_begin = int(bign.size());
    lp.small_maxn = lp.small_maxm = 0;
    int big_maxn = 0;
    lp.tiny_cnt = lp.tiny32_cnt = 0;
    for (int pass = 0; pass < 3; ++pass)      // tiny fronts first (<= 32 pivots, then <= 64): the LDL^T solves give them a wave each
      for (int i = lp.node_begin; i < lp.node_end; ++i) {
        const int s = lvl_nodes[i];
        const bool big = S.ncol(s) > BIG_N || S.nrow(s) > BIG_M;
        const bool tiny = S.ncol(s) <= 64 && S.nrow(s) - S.ncol(s) <= 64;
        const int cls = !tiny ? 2 : (S.ncol(s) <= 32 ? 0 : 1);
        if (big) {
          if (pass == 0) {
            bign.push_back(s);
            big_maxn = std::max(big_maxn, S.ncol(s));
          }
        } else if (cls == pass) {
          smalln.push_back(s);
          if (tiny) lp.tiny_cnt++;
          if (cls == 0) lp.tiny32_cnt++;
          lp.small_maxn = std::max(lp.small_maxn, S.ncol(s));
          lp.small_maxm = std::max(lp.small_maxm, S.nrow(s));
        }
      }
    lp.small_cnt = int(smalln.size()) - lp.small_begin;
    lp.big_cnt = int(bign.size()) - lp.big_begin;
    lp.bigsteps.clear();
    for (int b = 0; b < big_maxn; b += 64) {
      BigStep bs;
      bs.b = b;
      bs.trsv_begin = int(btr.size());
      bs.gemv_begin = int(bgm.size());
      for (int i = lp.big_begin; i < lp.big_begin + lp.big_cnt; ++i) {
        const int s = bign[i];
        if (S.ncol(s) <= b) continue;
        BigTrsv t{s, int(bgm.size()) - bs.gemv_begin, 0, 0};
        for (int r0 = b + std::min(64, S.ncol(s) - b); r0 < S.nrow(s); r0 += 256) {
          bgm.push_back(BigGemv{s, r0});
          t.part_cnt++;
        }
        btr.push_back(t);
      }
      bs.trsv_cnt = int(btr.size()) - bs.trsv_begin;
      bs.gemv_cnt = int(bgm.size()) - bs.gemv_begin;
      part_max = std::max<int64_t>(part_max, bs.gemv_cnt);
      lp.bigsteps.push_back(bs);
    }
  }
  };
  auto all = [](int) { return true; };
  build_plan(F.plan, all, all);
  // the same schedule with the tiny fronts (n <= TINY_N, m <= 64) handed to k_front_wave (LDL^T refactorizations)
  auto is_wg = [&](int s) { return !(S.ncol(s) <= TINY_N && S.nrow(s) <= 64); };
  build_plan(F.planT, all, is_wg);
  {
    // A -> L scatter of that schedule: only the fronts the workgroup kernels assemble
    std::vector<int64_t> s2, d2;
    for (int s = 0; s < nn; ++s)
      if (is_wg(s))
        for (int64_t k = S.nptr[s]; k < S.nptr[s + 1]; ++k) {
          s2.push_back(asrc[k]);
          d2.push_back(adst[k]);
        }
    F.nscatter_wg = int64_t(s2.size());
    HIPCHK(upload(F.asrc_wg, s2, st));
    HIPCHK(upload(F.adst_wg, d2, st));
  }
  F.sharded = !S.owner.empty() && S.nranks > 1;
  if (F.sharded) {
    build_plan(F.planA, [&](int s) { return S.owner[s] == F.myrank; }, all);
    build_plan(F.planB, [&](int s) { return S.owner[s] < 0; }, all);
    // the same two with the tiny fronts on the wave-per-front kernels (refactorizations that need no pivoting)
    build_plan(F.planAT, [&](int s) { return S.owner[s] == F.myrank; }, is_wg);
    build_plan(F.planBT, [&](int s) { return S.owner[s] < 0; }, is_wg);
  }

  // ---- wave tier of the LDL^T solves: stages of small subtrees, one wave each ------------------------------
  F.wave = false;
  if (!F.sharded && nn > 0 && !getenv("GSLS_NO_WAVE")) {
    std::vector<char> waveT(nn);          // the front and everything below it has at most 64 rows
    int nT = 0;
    for (int s = 0; s < nn; ++s) waveT[s] = (S.nrow(s) <= 64);
    for (int s = 0; s < nn; ++s) {        // supernodes are numbered in postorder: children first
      const int p2 = S.sparent[s];
      if (p2 < nn && !waveT[s]) waveT[p2] = 0;
    }
    for (int s = 0; s < nn; ++s) nT += waveT[s];
    if (nT > 0) {
      const int gmax = getenv("GSLS_GMAX") ? std::max(1, atoi(getenv("GSLS_GMAX"))) : 8;
      // groups: walk the tree bottom-up; a front joins the open subtrees of its children unless that would make the
      // group too big or too deep -- then the children's subtrees are closed (each becomes a group)
      std::vector<int> open(nn, 0), hgt(nn, 0);
      std::vector<char> closed(nn, 0);
      const int upper_lvl = getenv("GSLS_UPPER_LVL") ? atoi(getenv("GSLS_UPPER_LVL")) : 1024;
      for (int s = 0; s < nn; ++s) {
        if (!waveT[s]) {
          for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci)
            if (waveT[S.clist[ci]]) closed[S.clist[ci]] = 1;
          continue;
        }
        int sz = 1, h = 1;
        for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) {
          const int c = S.clist[ci];
          if (closed[c]) continue;
          sz += open[c];
          h = std::max(h, hgt[c] + 1);
        }
        // near the top of the tree a level holds few fronts and a wave walking a group serialises them: there a front
        // only extends a CHAIN (one open child); siblings stay separate groups and run side by side, a stage earlier
        int nopen = 0;
        for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) nopen += closed[S.clist[ci]] ? 0 : 1;
        const bool upper = (S.lvlptr[S.level[s] + 1] - S.lvlptr[S.level[s]]) < upper_lvl;
        if (sz > gmax || h > WSLOT || (upper && nopen > 1)) {
          for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) closed[S.clist[ci]] = 1;
          sz = 1;
          h = 1;
        }
        open[s] = sz;
        hgt[s] = h;
        if (S.sparent[s] >= nn) closed[s] = 1;
      }
      // group root, depth inside the group (parents first), stage of every group (children first)
      std::vector<int> groot(nn, -1), gdepth(nn, 0), gstage(nn, 0);
      for (int s = nn - 1; s >= 0; --s) {
        if (!waveT[s]) continue;
        if (closed[s]) {
          groot[s] = s;
          gdepth[s] = 0;
        } else {
          groot[s] = groot[S.sparent[s]];
          gdepth[s] = gdepth[S.sparent[s]] + 1;
        }
      }
      int nstage = 0;
      for (int s = 0; s < nn; ++s) {
        if (!waveT[s] || !closed[s]) continue;
        nstage = std::max(nstage, gstage[s] + 1);
        const int p2 = S.sparent[s];
        if (p2 < nn && waveT[p2]) gstage[groot[p2]] = std::max(gstage[groot[p2]], gstage[s] + 1);
      }
      // members of every group in postorder
      std::vector<int> gcount(nn, 0);
      for (int s = 0; s < nn; ++s)
        if (waveT[s]) gcount[groot[s]]++;
      struct GInfo { int root, stage, cnt, wide; };
      std::vector<GInfo> gi;
      {
        std::vector<char> gwide(nn, 0);
        for (int s = 0; s < nn; ++s)
          if (waveT[s]) {
            bool pull = false;
            for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) pull |= bool(closed[S.clist[ci]]);
            const bool orphan = closed[s] && S.sparent[s] < nn && !waveT[S.sparent[s]];   // its parent is outside the tier
            if (S.ncol(s) > 32 || S.nrow(s) > 40 || pull || orphan) gwide[groot[s]] = 1;
          }
        for (int s = 0; s < nn; ++s)
          if (waveT[s] && closed[s]) gi.push_back(GInfo{s, gstage[s], gcount[s], gwide[s]});
      }
      // launch order: by stage, narrow groups (every front <= 32 columns and <= 40 rows, no children in earlier stages,
      // the root's parent inside the tier: what the kernels that load one front ahead handle) before wide ones, long
      // groups first
      std::stable_sort(gi.begin(), gi.end(), [](const GInfo& a, const GInfo& b) {
        if (a.stage != b.stage) return a.stage < b.stage;
        if (a.wide != b.wide) return a.wide < b.wide;
        return a.cnt > b.cnt;
      });
      // A wave walks a RUN of groups one after the other (WGroup = the run's task range): what a wave pays before
      // its first front's data arrives -- three dependent round trips -- is paid once per run instead of once per
      // subtree, and the look-ahead carries across the subtree boundaries.  Runs of one (stage, narrow/wide) class
      // are filled to about `wtarget` fronts: enough runs to fill the chip once, no more.
      std::vector<int64_t> gtbeg(nn, 0);      // first task of the group rooted at node
      std::vector<WGroup> wg;
      F.wstage_begin.assign(nstage, 0);
      F.wstage_cnt.assign(nstage, 0);
      F.wstage_narrow.assign(nstage, 0);
      {
        std::vector<int64_t> stage_tasks(nstage, 0);
        for (const GInfo& g : gi) stage_tasks[g.stage] += g.cnt;
        const int wslots = getenv("GSLS_WSLOTS") ? std::max(1, atoi(getenv("GSLS_WSLOTS"))) : 16384;
        int64_t tb = 0;
        int cur_stage = -1, cur_wide = -1, cur_cnt = 0, target = 1;
        for (size_t g = 0; g < gi.size(); ++g) {
          gtbeg[gi[g].root] = tb;
          const bool fresh = gi[g].stage != cur_stage || gi[g].wide != cur_wide || cur_cnt >= target;
          if (fresh) {
            if (gi[g].stage != cur_stage) F.wstage_begin[gi[g].stage] = int(wg.size());
            cur_stage = gi[g].stage;
            cur_wide = gi[g].wide;
            target = getenv("GSLS_WTARGET") ? std::max(1, atoi(getenv("GSLS_WTARGET")))
                                            : int(std::max<int64_t>(1, (stage_tasks[cur_stage] + wslots - 1) / wslots));
            wg.push_back(WGroup{int32_t(tb), 0});
            cur_cnt = 0;
            F.wstage_cnt[cur_stage]++;
            if (!cur_wide) F.wstage_narrow[cur_stage]++;
          }
          wg.back().tcnt += gi[g].cnt;
          cur_cnt += gi[g].cnt;
          tb += gi[g].cnt;
        }
      }
      std::vector<WTask> wt(nT);
      std::vector<WPack> wp(nT);
      std::vector<int32_t> gptr(1, 0);
      std::vector<int64_t> gsrc;
      std::vector<int32_t> pull2;
      {
        std::vector<int64_t> fill(gtbeg);
        for (int s = 0; s < nn; ++s) {        // ascending node number = postorder inside every group
          if (!waveT[s]) continue;
          const int64_t ti = fill[groot[s]]++;
          WTask& t = wt[ti];
          t.m = S.nrow(s);
          t.n = S.ncol(s);
          t.sptr = S.sptr[s];
          t.flags = closed[s] ? ((S.sparent[s] < nn && waveT[S.sparent[s]]) ? WT_ZVEC : 0) : WT_PUSH;
          t.lfoff = t.lboff = 0;
          t.roff = S.rptr[s];
          t.moff = S.cmapptr[s];
          t.goff = 0;
          t.myslot = gdepth[s];
          t.pslot = closed[s] ? 0 : gdepth[S.sparent[s]];
          for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) t.flags |= closed[S.clist[ci]] ? WT_PULL : WT_INT;
          wp[ti] = WPack{S.loff[s], s, 0};
        }
      }
      // fused input permutation (k_wsolve_fwd<.., GATHER>): every position that is NOT a pivot of a front in the narrow
      // runs of stage 0 -- those fronts fetch their right-hand side through invp themselves
      {
        std::vector<char> own(std::max(S.n, 1), 0);
        if (nstage > 0)
          for (int r = F.wstage_begin[0]; r < F.wstage_begin[0] + F.wstage_narrow[0]; ++r)
            for (int64_t ti = wg[r].tbeg; ti < int64_t(wg[r].tbeg) + wg[r].tcnt; ++ti)
              for (int c = 0; c < wt[ti].n; ++c) own[wt[ti].sptr + c] = 1;
        std::vector<int32_t> rest;
        for (int i = 0; i < S.n; ++i)
          if (!own[i]) rest.push_back(i);
        F.wperm_cnt = int(rest.size());
        rest.push_back(0);
        HIPCHK(upload(F.wperm_list, rest, st));
      }
      // deepest LDS slot the narrow runs of each stage use (the column-group launches size their accumulators by it)
      F.wstage_ndepth.assign(nstage, 1);
      for (int k = 0; k < nstage; ++k)
        for (int r = F.wstage_begin[k]; r < F.wstage_begin[k] + F.wstage_narrow[k]; ++r)
          for (int64_t ti = wg[r].tbeg; ti < int64_t(wg[r].tbeg) + wg[r].tcnt; ++ti)
            F.wstage_ndepth[k] = std::max(F.wstage_ndepth[k], std::max(wt[ti].myslot, wt[ti].pslot) + 1);
      // images and gather lists in launch order (run by run; laying the images out in the order the waves of a launch
      // reach them -- the k-th fronts of all runs next to each other -- was measured: no difference)
      int64_t of = 0, ob = 0;
      std::vector<int64_t> nlf(nn, -1), nlb(nn, -1);
      for (int64_t ti = 0; ti < nT; ++ti) {
        WTask& t = wt[ti];
        const int s = wp[ti].node, pm = t.m;
        t.lfoff = of;
        t.lboff = ob;
        nlf[s] = of;
        nlb[s] = ob;
        of += wf_size(t.m, t.n);
        ob += wb_size(t.m, t.n);
        if (!(t.flags & WT_PULL)) continue;
        std::vector<std::vector<int64_t>> rows(pm);
        for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) {
          const int c = S.clist[ci];
          if (!closed[c]) continue;            // children inside the group arrive through LDS
          const int ccm = S.nrow(c) - S.ncol(c);
          for (int k = 0; k < ccm; ++k) rows[S.cmap[S.cmapptr[c] + k]].push_back(S.cmapptr[c] + k);
        }
        {
          // at most two sources per row and 32-bit indices: the dense form (one 8-byte pair per row)
          size_t longest = 0;
          for (int r2 = 0; r2 < pm; ++r2) longest = std::max(longest, rows[r2].size());
          if (longest <= 2 && int64_t(S.cmapptr[nn]) < (int64_t(1) << 31) && !getenv("GSLS_NO_WPULL2")) {
            t.flags |= WT_DENSE;
            t.goff = int64_t(pull2.size() / 2);
            for (int r2 = 0; r2 < pm; ++r2)
              for (size_t q = 0; q < 2; ++q) pull2.push_back(q < rows[r2].size() ? int32_t(rows[r2][q]) : -1);
            continue;
          }
        }
        t.goff = int64_t(gptr.size()) - 1;
        for (int r2 = 0; r2 < pm; ++r2) {
          gsrc.insert(gsrc.end(), rows[r2].begin(), rows[r2].end());
          gptr.push_back(int32_t(gsrc.size()));
        }
        gptr.push_back(int32_t(gsrc.size()));   // one spare entry: every task owns m + 1 consecutive pointers
      }
      if (getenv("GSLS_DEBUG"))
        for (int sg = 0; sg < nstage; ++sg) {
          int runs = F.wstage_cnt[sg], maxrun = 0, maxn = 0, maxm = 0, npull = 0, ntask = 0, maxsrc = 0;
          for (int r = F.wstage_begin[sg]; r < F.wstage_begin[sg] + runs; ++r) {
            maxrun = std::max(maxrun, int(wg[r].tcnt));
            for (int64_t ti = wg[r].tbeg; ti < wg[r].tbeg + wg[r].tcnt; ++ti) {
              ++ntask;
              maxn = std::max(maxn, int(wt[ti].n));
              maxm = std::max(maxm, int(wt[ti].m));
              if (wt[ti].flags & WT_PULL) {
                ++npull;
                if (wt[ti].flags & WT_DENSE) { maxsrc = std::max(maxsrc, 2); continue; }
                for (int r2 = 0; r2 < wt[ti].m; ++r2)
                  maxsrc = std::max(maxsrc, int(gptr[wt[ti].goff + r2 + 1] - gptr[wt[ti].goff + r2]));
              }
            }
          }
          fprintf(stderr, "[gsls] wave stage %d: %d runs (%d narrow), %d fronts, longest run %d, max n %d m %d, %d pulling fronts (longest row list %d)\n",
                  sg, runs, F.wstage_narrow[sg], ntask, maxrun, maxn, maxm, npull, maxsrc);
        }
      for (auto& tk : tft) {
        tk.lfoff = nlf[tk.node];
        tk.lboff = nlb[tk.node];
      }
      {
        int64_t big = 0;      // the wide launches stage one image per wave in LDS: whole 1 KB pieces
        for (int64_t ti = 0; ti < nT; ++ti) big = std::max<int64_t>(big, 8 * wf_size(wt[ti].m, wt[ti].n));
        F.wimg_units = int(((big + 1023) / 1024) * 64);
      }
      // the tail: the longest run of last stages with at most WTAIL_WAVES groups each (never stage 0)
      F.wtail_k0 = -1;
      if (!getenv("GSLS_NO_WTAIL")) {
        int k0 = nstage;
        while (k0 - 1 >= 1 && F.wstage_cnt[k0 - 1] <= WTAIL_WAVES && nstage - (k0 - 1) <= WTAIL_STAGES) --k0;
        if (nstage - k0 >= 2 && int64_t(WTAIL_WAVES) * F.wimg_units * 16 <= 116 * 1024) {   // (one stage = one launch either way)
          F.wtail_k0 = k0;
          F.wtail_tbeg = wg[F.wstage_begin[k0]].tbeg;
          F.wtail_tcnt = nT - F.wtail_tbeg;
          F.wtail_lf0 = wt[F.wtail_tbeg].lfoff;
          F.wtail_lb0 = wt[F.wtail_tbeg].lboff;
          F.wtail_gp0 = F.wtail_gp1 = int64_t(gptr.size()) - 1;
          for (int64_t ti = F.wtail_tbeg; ti < nT; ++ti)
            if ((wt[ti].flags & WT_PULL) && !(wt[ti].flags & WT_DENSE)) F.wtail_gp0 = std::min<int64_t>(F.wtail_gp0, wt[ti].goff);
          F.wtail_gs0 = gptr[F.wtail_gp0];
          F.wtail_gs1 = int64_t(gsrc.size());
        }
      }
      std::vector<int32_t> nont;
      for (int s = 0; s < nn; ++s)
        if (!waveT[s]) nont.push_back(s);
      F.wnont_cnt = int(nont.size());
      F.wtask_cnt = nT;
      F.Lf_elems = of;
      F.Lb_elems = 0;

      {
        WTask* d1 = nullptr;
        WGroup* d2 = nullptr;
        WPack* d3 = nullptr;
        HIPCHK(upload(d1, wt, st));
        HIPCHK(upload(d2, wg, st));
        HIPCHK(upload(d3, wp, st));
        F.wtasks = d1;
        F.wgroups = d2;
        F.wpacks = d3;
      }
      gsrc.insert(gsrc.end(), 16, 0);            // (the pipelined gathers read one clamped index past a row's list)
      HIPCHK(upload(F.wgth_ptr, gptr, st));
      HIPCHK(upload(F.wgth_src, gsrc, st));
      pull2.resize(pull2.size() + 128, -1);      // (the masked 8-byte loads never read it; keeps the buffer non-empty)
      HIPCHK(upload(F.wpull2, pull2, st));
      // stages whose runs are all single fronts (the upper tree): run gi = task tbeg + gi, no group record needed
      F.wstage_unit.assign(nstage, -1);
      for (int k = 0; k < nstage; ++k) {
        bool unit = F.wstage_cnt[k] > 0;
        for (int r = F.wstage_begin[k]; unit && r < F.wstage_begin[k] + F.wstage_cnt[k]; ++r)
          unit = wg[r].tcnt == 1 && wg[r].tbeg == wg[F.wstage_begin[k]].tbeg + (r - F.wstage_begin[k]);
        if (unit) F.wstage_unit[k] = wg[F.wstage_begin[k]].tbeg;
      }
      HIPCHK(upload(F.wnont, nont, st));
      HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.Lf), std::max<int64_t>(of, 2) * sizeof(double)));
      F.Lb = nullptr;       // (one image since round 3: the backward sweep transposes Lf in LDS)
      HIPCHK(hipMemsetAsync(F.Lf, 0, std::max<int64_t>(of, 2) * sizeof(double), st));   // the diagonal slots stay 0
      HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.xs), (std::max(S.n, 1) + 64) * sizeof(double)));
      HIPCHK(hipMemsetAsync(F.xs, 0, (std::max(S.n, 1) + 64) * sizeof(double), st));
      HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.gvar), (std::max(S.n, 1) + 64) * sizeof(int32_t)));
      HIPCHK(hipMemsetAsync(F.gvar, 0, (std::max(S.n, 1) + 64) * sizeof(int32_t), st));
      build_plan(F.planW, [&](int s) { return !waveT[s]; }, all);
      F.wave = true;
      if (getenv("GSLS_DEBUG")) {
        fprintf(stderr, "[gsls] wave tier: %d of %d fronts in %zu groups, %zu runs, %d stages (runs per stage:", nT, nn, gi.size(), wg.size(), nstage);
        for (int k = 0; k < nstage; ++k) fprintf(stderr, " %d", F.wstage_cnt[k]);
        fprintf(stderr, "), images %.1f + %.1f MB\n", of * 8e-6, ob * 8e-6);
      }
    }
  }

  HIPCHK(upload(F.nodes, nd, st));
  HIPCHK(upload(F.rlist, S.rlist, st));
  {
    std::vector<int> cm2(S.cmap);           // + 64: the wave tier reads cmap[moff + lane] unmasked
    cm2.resize(cm2.size() + 64, 0);
    HIPCHK(upload(F.cmap, cm2, st));
  }
  HIPCHK(upload(F.clist, S.clist, st));
  HIPCHK(upload(F.lvlnodes, S.lvlnodes, st));
  HIPCHK(upload(F.smallnodes, smalln, st));
  {
    std::vector<SolveTask> stv(smalln.size());
    for (size_t i = 0; i < smalln.size(); ++i) {
      const int sn = smalln[i];
      SolveTask& t = stv[i];
      t.m = S.nrow(sn);
      t.n = S.ncol(sn);
      t.ld = S.ldl[sn];
      t.sptr = S.sptr[sn];
      t.loff = S.loff[sn];
      t.roff = S.rptr[sn];
      t.moff = S.cmapptr[sn];
      t.iblk = nd[sn].iblk;
      t.cbeg = S.cptr[sn];
      t.cend = S.cptr[sn + 1];
      t.ccm0 = t.ccm1 = t.pad = 0;
      t.cmoff0 = t.cmoff1 = 0;
      if (t.cend - t.cbeg > 0) {
        const int c = S.clist[t.cbeg];
        t.ccm0 = S.nrow(c) - S.ncol(c);
        t.cmoff0 = S.cmapptr[c];
      }
      if (t.cend - t.cbeg > 1) {
        const int c = S.clist[t.cbeg + 1];
        t.ccm1 = S.nrow(c) - S.ncol(c);
        t.cmoff1 = S.cmapptr[c];
      }
    }
    // gather lists: for every row of a small front, the children's contribution-vector entries that add
    // into it, in child order
    std::vector<int32_t> gptr;
    std::vector<int64_t> gsrc;
    gptr.push_back(0);
    for (size_t i = 0; i < smalln.size(); ++i) {
      const int sn = smalln[i];
      const int pm = S.nrow(sn);
      stv[i].goff = int64_t(gptr.size()) - 1;
      std::vector<std::vector<int64_t>> rows(pm);
      for (int ci = S.cptr[sn]; ci < S.cptr[sn + 1]; ++ci) {
        const int c = S.clist[ci];
        const int ccm = S.nrow(c) - S.ncol(c);
        for (int k = 0; k < ccm; ++k) rows[S.cmap[S.cmapptr[c] + k]].push_back(S.cmapptr[c] + k);
      }
      for (int r2 = 0; r2 < pm; ++r2) {
        gsrc.insert(gsrc.end(), rows[r2].begin(), rows[r2].end());
        gptr.push_back(int32_t(gsrc.size()));
      }
    }
    HIPCHK(upload(F.gth_ptr, gptr, st));
    HIPCHK(upload(F.gth_src, gsrc, st));
    SolveTask* d = nullptr;
    HIPCHK(upload(d, stv, st));
    F.stasks = d;
  }
  HIPCHK(upload(F.bignodes, bign, st));
  {
    BigTrsv* d1 = nullptr;
    BigGemv* d2 = nullptr;
    HIPCHK(upload(d1, btr, st));
    HIPCHK(upload(d2, bgm, st));
    F.bigtrsv = d1;
    F.biggemv = d2;
  }
  if (F.sharded) {
    std::vector<Segment> sc, sv;
    int64_t oc = 0, ov = 0;
    for (int c : S.cutroots) {
      const int64_t cm = S.nrow(c) - S.ncol(c);
      sc.push_back(Segment{S.coff[c], oc, cm * cm, S.owner[c], 0});
      sv.push_back(Segment{S.cmapptr[c], ov, cm, S.owner[c], 0});
      oc += cm * cm;
      ov += cm;
    }
    F.nseg = int(sc.size());
    F.xchgC_elems = oc;
    F.xchgV_elems = ov;
    std::vector<Segment> sz;
    {
      int64_t oz = 0;
      for (int c : S.cutroots) {
        const int64_t cm = S.nrow(c) - S.ncol(c);
        sz.push_back(Segment{S.rptr[c] + S.ncol(c), oz, cm, S.owner[c], 0});
        oz += cm;
      }
    }
    Segment *d1 = nullptr, *d2 = nullptr, *d3 = nullptr;
    HIPCHK(upload(d1, sc, st));
    HIPCHK(upload(d2, sv, st));
    HIPCHK(upload(d3, sz, st));
    F.segC = d1;
    F.segV = d2;
    F.segZ = d3;
    std::vector<int32_t> po(S.n, -1);
    for (int s = 0; s < nn; ++s)
      for (int p2 = S.sptr[s]; p2 < S.sptr[s + 1]; ++p2) po[p2] = S.owner[s];
    HIPCHK(upload(F.posowner, po, st));
  }
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.ybuf), std::max(S.n, 1) * sizeof(double)));
  F.part_elems = std::max<int64_t>(part_max, 1) * 64;
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.part), F.part_elems * sizeof(double)));
  {
    PullSeg* d1 = nullptr;
    PullTask* d2 = nullptr;
    HIPCHK(upload(d1, psg, st));
    HIPCHK(upload(d2, ptk, st));
    F.pullsegs = d1;
    F.pulltasks = d2;
    TinyContribTask* d3 = nullptr;
    HIPCHK(upload(d3, tct, st));
    F.tinyctasks = d3;
    TinyFrontTask* d4 = nullptr;
    HIPCHK(upload(d4, tft, st));
    F.tftasks = d4;
    HIPCHK(upload(F.gdst, fgdst, st));
    HIPCHK(upload(F.gbeg, fgbeg, st));
    HIPCHK(upload(F.gsrc, fgsrc, st));
  }
  HIPCHK(upload(F.aloc, aloc, st));
  HIPCHK(upload(F.asrc, asrc, st));
  HIPCHK(upload(F.adst, adst, st));
  HIPCHK(upload(F.arow, arow, st));
  HIPCHK(upload(F.acol, acol, st));
  HIPCHK(upload(F.ptasks, pt, st));
  HIPCHK(upload(F.ttasks, tt, st));
  HIPCHK(upload(F.invp, S.invp, st));
  F.nscatter = nz;
  F.nblk64 = nblk64;
  F.L_elems = S.loff[nn];
  F.C_elems = S.coff[nn];
  {
    std::vector<ZeroTask> zt;
    const int nzl = std::max(int(S.czptr.size()) - 1, S.nlevels);   // (2 * nlevels with a per-rank layout: phase 2 behind phase 1)
    F.cz_begin.assign(nzl, 0);
    F.cz_cnt.assign(nzl, 0);
    const int64_t chunk = int64_t(1) << 13;        // 64 KB per workgroup
    for (int l = 0; l < int(S.czptr.size()) - 1; ++l) {
      F.cz_begin[l] = int(zt.size());
      for (int r = S.czptr[l]; r < S.czptr[l + 1]; ++r)
        for (int64_t o = 0; o < S.czlen[r]; o += chunk) zt.push_back(ZeroTask{S.czoff[r] + o, std::min(chunk, S.czlen[r] - o)});
      F.cz_cnt[l] = int(zt.size()) - F.cz_begin[l];
    }
    ZeroTask* d = nullptr;
    HIPCHK(upload(d, zt, st));
    F.cztasks = d;
  }
  F.cvec_elems = S.cmapptr[nn];
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.L), std::max<int64_t>(F.L_elems, 1) * sizeof(double)));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.C), std::max<int64_t>(F.C_elems, 1) * sizeof(double)));
  // (+ 64 elements / 132 doubles of padding: the wave tier loads [sptr + lane] and [moff + lane] unmasked)
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.D), (2 * int64_t(S.n) + 4 + 132) * sizeof(double)));
  HIPCHK(hipMemsetAsync(F.D, 0, (2 * int64_t(S.n) + 4 + 132) * sizeof(double), st));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.gperm), (std::max(S.n, 1) + 64) * sizeof(int32_t)));
  HIPCHK(hipMemsetAsync(F.gperm, 0, (std::max(S.n, 1) + 64) * sizeof(int32_t), st));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.cvec), (std::max<int64_t>(F.cvec_elems, 1) + 64) * sizeof(double)));
  HIPCHK(hipMemsetAsync(F.cvec, 0, (std::max<int64_t>(F.cvec_elems, 1) + 64) * sizeof(double), st));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.stat), NSTAT * sizeof(int32_t)));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.faillist), FAILCAP * sizeof(int32_t)));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.fastok), std::max<int64_t>(F.nblk64, 1) * sizeof(int32_t)));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.tinyskip), std::max(nn, 1)));
  HIPCHK(hipMemsetAsync(F.tinyskip, 0, std::max(nn, 1), st));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.tinyfail), FAILCAP * sizeof(int32_t)));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.tppflag), std::max(nn, 1)));
  HIPCHK(hipMemsetAsync(F.tppflag, 0, std::max(nn, 1), st));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.hint), std::max(S.n, 1)));
  HIPCHK(hipMemsetAsync(F.hint, 0, std::max(S.n, 1), st));
  HIPCHK(hipStreamSynchronize(st));  // host vectors go out of scope
  return hipSuccess;
}

// -------------------------------------------------------------------------------------------------
template <bool POSDEF>
static hipError_t factor_levels(const Symbolic& S, DeviceFactor& F, const std::vector<LevelPlan>& plan,
                                double small, double u, hipStream_t st, int which = 0, bool zero_c = true) {
  const size_t lds_diag = std::max(sizeof(Stage<PR>), sizeof(double) * LDP * NB);
  const size_t lds_panel = std::max(sizeof(Stage<RB>), sizeof(double) * (NB * RBP + NB * NB));
  const size_t lds_chol = std::max(sizeof(Stage<PR, CK>), sizeof(double) * LDQ * NB);
  const size_t lds_pchol = std::max(sizeof(Stage<RB, CK>), sizeof(double) * (2 * NB * RBP));
  // a sharded run with a per-rank layout keeps phase 2's clear lists behind phase 1's (shard_layout)
  const int zbase = (which == 2 && int(F.cz_cnt.size()) == 2 * S.nlevels) ? S.nlevels : 0;
  if (zbase > 0) zero_c = true;
  for (int l = 0; l < S.nlevels; ++l) {
    const LevelPlan& lp = plan[l];
    // (a pass on the wave-per-front kernels only writes every contribution block whole, once: nothing to clear)
    const bool all_wave = !POSDEF && &plan == &F.planT && F.nscatter_wg == 0 && F.bl_count == 0 && zbase == 0;
    if (zero_c && !all_wave && F.cz_cnt[zbase + l] > 0)     // the contribution blocks this level's fronts own (arena space is reused)
      hipLaunchKernelGGL(k_zero_tasks, dim3(F.cz_cnt[zbase + l]), dim3(256), 0, st,
                         static_cast<const ZeroTask*>(F.cztasks) + F.cz_begin[zbase + l], F.C);
    if (lp.pull_cnt > 0)
      hipLaunchKernelGGL(k_assemble_pull, dim3(lp.pull_cnt), dim3(256), 0, st,
                         static_cast<const PullTask*>(F.pulltasks) + lp.pull_begin,
                         static_cast<const PullSeg*>(F.pullsegs), F.cmap, F.L, F.C);
    const bool bl_here = !POSDEF && &plan == &F.planT && l < int(F.bl_level.size()) && F.bl_level[l].np > 0;
    if (bl_here && F.bl_level[l].npull > 0)    // before k_front_tpp: a blacklisted front may be flagged as well
      hipLaunchKernelGGL(k_assemble_pull, dim3(F.bl_level[l].npull), dim3(256), 0, st,
                         static_cast<const PullTask*>(F.bl_pulltasks) + F.bl_level[l].pullbeg,
                         static_cast<const PullSeg*>(F.bl_pullsegs), F.cmap, F.L, F.C);
    if (!POSDEF && !F.tpp_cnt[which].empty() && F.tpp_cnt[which][l] > 0)   // fronts flagged for whole-front pivoting
      hipLaunchKernelGGL(k_front_tpp, dim3(F.tpp_cnt[which][l]), dim3(256), 0, st, F.nodes,
                         F.tpplist + F.tpp_begin[which][l], F.L, F.D, F.gperm, F.stat, F.faillist, small, u, S.nnodes);
    if (!POSDEF && lp.tf_cnt > 0) {   // only in the tiny-front plan (planT): assembly + factorization, a wave per front
      const GatherLists gl{F.gdst, F.gbeg, F.gsrc};
      // one launch per width class on a wide level; on a narrow one (latency, not throughput) a single launch of
      // the widest class present.  Classes with few fronts ride with the next wider one.
      static const int blk_mode = getenv("GSLS_FRONT_BLK") ? atoi(getenv("GSLS_FRONT_BLK")) : 0;
      if (blk_mode == 1 || (blk_mode == 2 && lp.tf_cnt <= 2048)) {      // every width in one launch of the blocked kernel
        int maxm = 0;
        for (int cls = 0; cls < TINY_CLASSES; ++cls) maxm = std::max(maxm, lp.tf_cls_maxm[cls]);
        const int tri = (maxm * (maxm + 1) / 2 + 1) & ~1, cnt = lp.tf_cnt;
        const TinyFrontTask* tf = static_cast<const TinyFrontTask*>(F.tftasks) + lp.tf_begin;
        hipLaunchKernelGGL((k_front_blk<4>), dim3((cnt + 3) / 4), dim3(256), size_t(4) * (tri + 512) * 8, st, tf, cnt, gl,
                           F.asrc, F.aloc, F.cur_val, F.L, F.D, F.C, F.stat, F.fastok, F.hint, F.tinyskip, F.tinyfail,
                           small, u, F.wave ? F.Lf : nullptr, F.wave ? F.Lb : nullptr, tri, 0);
      } else {
      int beg = lp.tf_begin, cnt = 0, maxm = 0;
      static const int narrow_max = getenv("GSLS_NARROW_MAX") ? atoi(getenv("GSLS_NARROW_MAX")) : 2048;
      // ... and up to 16 384 fronts when the classes present are neighbours (24 | 28, 28 | 32, ...): one launch of the wider
      // body costs the narrower fronts ~1.3x their arithmetic, two launches cost the level a second ramp-up and tail
      // (metric workload: 0.696 -> 0.675 ms per step; a span of two classes measures the same, three is back to 0.687)
      static const int merge_max = getenv("GSLS_MERGE_MAX") ? atoi(getenv("GSLS_MERGE_MAX")) : 16384;
      static const int merge_span = getenv("GSLS_MERGE_SPAN") ? atoi(getenv("GSLS_MERGE_SPAN")) : 1;
      int clo = TINY_CLASSES, chi = -1;
      for (int c2 = 0; c2 < TINY_CLASSES; ++c2)
        if (lp.tf_cls_cnt[c2] > 0) { clo = std::min(clo, c2); chi = std::max(chi, c2); }
      const bool narrow = lp.tf_cnt <= narrow_max || (lp.tf_cnt <= merge_max && chi - clo <= merge_span && chi <= 3);
      for (int cls = 0; cls < TINY_CLASSES; ++cls) {
        cnt += lp.tf_cls_cnt[cls];
        maxm = std::max(maxm, lp.tf_cls_maxm[cls]);
        int later = 0;
        for (int c2 = cls + 1; c2 < TINY_CLASSES; ++c2) later += lp.tf_cls_cnt[c2];
        if (cnt == 0 || (later > 0 && (narrow || cnt < 512))) continue;
        const TinyFrontTask* tf = static_cast<const TinyFrontTask*>(F.tftasks) + beg;
        const int tri = (maxm * (maxm + 1) / 2 + 1) & ~1;
        static const bool dbg_launch = getenv("GSLS_DEBUG_LAUNCH") != nullptr;
        if (dbg_launch)
          fprintf(stderr, "[gsls] level %d: class %d, %d fronts, max m %d, LDS %zu bytes per workgroup\n", l, cls, cnt, maxm,
                  size_t(4) * tri * 8);
#define GSLS_FW_ARGS tf, cnt, gl, F.asrc, F.aloc, F.cur_val, F.L, F.D, F.C, F.stat, F.fastok, F.hint, \
                     F.tinyskip, F.tinyfail, small, u, F.wave ? F.Lf : nullptr, F.wave ? F.Lb : nullptr, tri, \
                     (cls <= 3 ? (F.any_hint ? 1 : 0) : 0)
        switch (cls) {
          case 0:
            hipLaunchKernelGGL((k_front_wave<24, 4>), dim3((cnt + 3) / 4), dim3(256), size_t(4) * tri * 8, st, GSLS_FW_ARGS);
            break;
          case 1:
            hipLaunchKernelGGL((k_front_wave<28, 4>), dim3((cnt + 3) / 4), dim3(256), size_t(4) * tri * 8, st, GSLS_FW_ARGS);
            break;
          case 2:
            hipLaunchKernelGGL((k_front_wave<32, 4>), dim3((cnt + 3) / 4), dim3(256), size_t(4) * tri * 8, st, GSLS_FW_ARGS);
            break;
          case 3:
            hipLaunchKernelGGL((k_front_wave<36, 4>), dim3((cnt + 3) / 4), dim3(256), size_t(4) * tri * 8, st, GSLS_FW_ARGS);
            break;
          default:      // more than 36 columns: the blocked kernel (an unrolled 48-column body is 64 KB of code)
            hipLaunchKernelGGL((k_front_blk<4>), dim3((cnt + 3) / 4), dim3(256), size_t(4) * (tri + 512) * 8, st, GSLS_FW_ARGS);
        }
        if (cls <= 3 && F.any_hint)      // the fronts with learned 2x2 pivots that the unrolled kernel left alone
          hipLaunchKernelGGL((k_front_blk<4>), dim3((cnt + 3) / 4), dim3(256), size_t(4) * (tri + 512) * 8, st, tf, cnt, gl,
                             F.asrc, F.aloc, F.cur_val, F.L, F.D, F.C, F.stat, F.fastok, F.hint, F.tinyskip, F.tinyfail,
                             small, u, F.wave ? F.Lf : nullptr, F.wave ? F.Lb : nullptr, tri, 1);
#undef GSLS_FW_ARGS
        beg += cnt;
        cnt = maxm = 0;
      }
      }
    }
    const int nsteps = int(lp.panel_cnt.size() / 2);
    for (int s = 0; s < nsteps; ++s) {
      if (lp.panel_cnt[2 * s] > 0) {
        if (POSDEF) {
          hipLaunchKernelGGL((k_diag_fast<false, true>), dim3(lp.panel_cnt[2 * s]), dim3(256), lds_chol, st, F.nodes,
                             F.ptasks + lp.panel_begin[2 * s], F.L, F.Linv, F.D, F.stat, F.fastok, F.hint, small, u,
                             LDQ, PRX / 16, F.tppflag);
        } else {
          // the LDS panel is as tall as this launch's tallest front: small fronts share a CU
          const int nrt = std::max(NB / 16, (lp.panel_rows[s] + 15) / 16);
          const int ldq = (16 * nrt) % 32 == 16 ? 16 * nrt : 16 * nrt + 16;
          const size_t lds_fldl = std::max(s > 0 ? sizeof(Stage<PR, CK>) : size_t(0), sizeof(double) * ldq * (NB + 16));
          if (s == 0)
            hipLaunchKernelGGL((k_diag_fast<true, false>), dim3(lp.panel_cnt[2 * s]), dim3(256), lds_fldl, st, F.nodes,
                               F.ptasks + lp.panel_begin[2 * s], F.L, F.Linv, F.D, F.stat, F.fastok, F.hint, small, u,
                               ldq, nrt, F.tppflag);
          else
            hipLaunchKernelGGL((k_diag_fast<true, true>), dim3(lp.panel_cnt[2 * s]), dim3(256), lds_fldl, st, F.nodes,
                               F.ptasks + lp.panel_begin[2 * s], F.L, F.Linv, F.D, F.stat, F.fastok, F.hint, small, u,
                               ldq, nrt, F.tppflag);
          hipLaunchKernelGGL(k_diag_ldlt, dim3(lp.panel_cnt[2 * s]), dim3(256), lds_diag, st, F.nodes,
                             F.ptasks + lp.panel_begin[2 * s], F.L, F.D, F.gperm, F.stat, F.faillist, F.fastok,
                             small, u, F.tppflag);
        }
      }
      if (lp.panel_cnt[2 * s + 1] > 0) {
        if (POSDEF)
          hipLaunchKernelGGL(k_panel_chol, dim3(lp.panel_cnt[2 * s + 1]), dim3(256), lds_pchol, st, F.nodes,
                             F.ptasks + lp.panel_begin[2 * s + 1], F.L, F.Linv);
        else
          hipLaunchKernelGGL(k_panel<false>, dim3(lp.panel_cnt[2 * s + 1]), dim3(256), lds_panel, st, F.nodes,
                             F.ptasks + lp.panel_begin[2 * s + 1], F.L, F.D, F.gperm, F.stat, F.faillist, u, F.tppflag, small);
      }
    }
    if (bl_here) {
      // tiny fronts that k_front_tiny could not take (blacklisted by the host): the workgroup kernels
      const BlLevel& b = F.bl_level[l];
      const int nrt = std::max(NB / 16, (b.rows + 15) / 16);
      const int ldq = (16 * nrt) % 32 == 16 ? 16 * nrt : 16 * nrt + 16;
      const size_t lds_fldl = sizeof(double) * ldq * (NB + 16);
      hipLaunchKernelGGL((k_diag_fast<true, false>), dim3(b.np), dim3(256), lds_fldl, st, F.nodes,
                         F.bl_ptasks + b.pbeg, F.L, F.Linv, F.D, F.stat, F.fastok, F.hint, small, u, ldq, nrt, F.tppflag);
      hipLaunchKernelGGL(k_diag_ldlt, dim3(b.np), dim3(256), lds_diag, st, F.nodes, F.bl_ptasks + b.pbeg, F.L, F.D,
                         F.gperm, F.stat, F.faillist, F.fastok, small, u, F.tppflag);
      if (b.nt > 0)
        hipLaunchKernelGGL(k_contrib<false>, dim3(b.nt), dim3(256), 0, st, F.nodes, F.bl_ttasks + b.tbeg, F.L, F.D,
                           F.C);
      if (b.ntc > 0)
        hipLaunchKernelGGL(k_contrib_tiny<false>, dim3((b.ntc + 1) / 2), dim3(128), 0, st,
                           static_cast<const TinyContribTask*>(F.bl_tctasks) + b.tcbeg, b.ntc, F.L, F.D, F.C);
    }
    if (lp.tile_cnt > 0)
      hipLaunchKernelGGL(k_contrib<POSDEF>, dim3(lp.tile_cnt), dim3(256), 0, st, F.nodes,
                         F.ttasks + lp.tile_begin, F.L, F.D, F.C);
    if (lp.tinyc_cnt > 0)
      hipLaunchKernelGGL(k_contrib_tiny<POSDEF>, dim3((lp.tinyc_cnt + 1) / 2), dim3(128), 0, st,
                         static_cast<const TinyContribTask*>(F.tinyctasks) + lp.tinyc_begin, lp.tinyc_cnt, F.L, F.D,
                         F.C);
  }
  return hipGetLastError();
}

// Tiny fronts that k_front_tiny could not take go (back) to the workgroup kernels: per level a short
// task list, rebuilt and uploaded whenever the blacklist grows (a few entries; no re-analysis).
hipError_t dev_set_tiny_blacklist(const Symbolic& S, DeviceFactor& F, const std::vector<int>& nodes,
                                  hipStream_t st) {
  PoolScope pool_scope(F);
  const int nn = S.nnodes;
  std::vector<uint8_t> skip(std::max(nn, 1), 0);
  std::vector<std::vector<int>> per(S.nlevels);
  for (int s : nodes) {
    // (a front the tiny plan gives to the workgroup kernels anyway has its tasks in the plan: listing it here as well
    // would assemble it twice)
    if (!(S.ncol(s) <= TINY_N && S.nrow(s) <= 64)) continue;
    skip[s] = 1;
    per[S.level[s]].push_back(s);
  }
  std::vector<PanelTask> pt;
  std::vector<TileTask> tt;
  std::vector<TinyContribTask> tc;
  std::vector<PullSeg> psg;
  std::vector<PullTask> ptk;
  F.bl_level.assign(S.nlevels, BlLevel());
  for (int l = 0; l < S.nlevels; ++l) {
    BlLevel& b = F.bl_level[l];
    b.pbeg = int(pt.size());
    b.tbeg = int(tt.size());
    b.tcbeg = int(tc.size());
    b.pullbeg = int(ptk.size());
    for (int s : per[l]) {                   // the extend-add k_front_wave would have done itself
      const int pm = S.nrow(s), pn = S.ncol(s);
      for (int pc0 = 0; pc0 < pm; pc0 += PCOLS) {
        PullTask tk{int(psg.size()), 0};
        for (int ci = S.cptr[s]; ci < S.cptr[s + 1]; ++ci) {
          const int c = S.clist[ci];
          const int cmc = S.nrow(c) - S.ncol(c);
          const int32_t* mp = S.cmap.data() + S.cmapptr[c];
          const int j0 = int(std::lower_bound(mp, mp + cmc, pc0) - mp);
          const int j1 = int(std::lower_bound(mp, mp + cmc, pc0 + PCOLS) - mp);
          if (j1 > j0)
            psg.push_back(PullSeg{j0, j1, cmc, pn, S.ldl[s], pm - pn, S.cmapptr[c], S.coff[c], S.loff[s], S.coff[s]});
        }
        tk.seg_cnt = int(psg.size()) - tk.seg_begin;
        if (tk.seg_cnt > 0) ptk.push_back(tk);
      }
    }
    b.npull = int(ptk.size()) - b.pullbeg;
    for (int s : per[l]) {
      pt.push_back(PanelTask{s, 0, 0, 0});
      b.rows = std::max(b.rows, std::min(PR, S.nrow(s)));
      if (S.sparent[s] >= nn) continue;
      const int cm = S.nrow(s) - S.ncol(s);
      if (cm <= 0) continue;
      if (cm <= 16) tc.push_back(TinyContribTask{S.ncol(s), cm, S.ldl[s], S.sptr[s], S.loff[s], S.coff[s]});
      else tt.push_back(TileTask{s, 0, 0, 0});
    }
    b.np = int(pt.size()) - b.pbeg;
    b.nt = int(tt.size()) - b.tbeg;
    b.ntc = int(tc.size()) - b.tcbeg;
  }
  for (void* p2 : {static_cast<void*>(F.bl_ptasks), static_cast<void*>(F.bl_ttasks), F.bl_tctasks, F.bl_pullsegs,
                   F.bl_pulltasks})
    pool_free(F, p2);
  F.bl_ptasks = nullptr;
  F.bl_ttasks = nullptr;
  F.bl_tctasks = nullptr;
  F.bl_pullsegs = F.bl_pulltasks = nullptr;
  F.bl_count = int(nodes.size());
  {
    PullSeg* d1 = nullptr;
    PullTask* d2 = nullptr;
    HIPCHK(upload(d1, psg, st));
    HIPCHK(upload(d2, ptk, st));
    F.bl_pullsegs = d1;
    F.bl_pulltasks = d2;
  }
  HIPCHK(upload(F.bl_ptasks, pt, st));
  HIPCHK(upload(F.bl_ttasks, tt, st));
  {
    TinyContribTask* d = nullptr;
    HIPCHK(upload(d, tc, st));
    F.bl_tctasks = d;
  }
  HIPCHK(hipMemcpyAsync(F.tinyskip, skip.data(), skip.size(), hipMemcpyHostToDevice, st));
  return hipStreamSynchronize(st);
}

// =================================================================================================
// the caller's own matrix on the device (gsls_set_coo / gsls_factor_coo / gsls_residual)
// =================================================================================================
// VAL(k) = sum of the caller's entries mapped to position k, in entry order (sls.f90:4113-4121)
__global__ void k_map_values(int64_t nz, const int64_t* __restrict__ ptr, const int32_t* __restrict__ src,
                             const double* __restrict__ vin, double* __restrict__ vout) {
  const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= nz) return;
  double v = 0.0;
  for (int64_t e = ptr[k]; e < ptr[k + 1]; ++e) v += vin[src[e]];
  vout[k] = v;
}
// r_i = b_i - sum_e val[src[e]] x[col[e]] over row i of the full symmetric matrix, entries in storage order
__global__ void k_coo_residual(int n, const int64_t* __restrict__ ptr, const int32_t* __restrict__ col,
                               const int32_t* __restrict__ src, const double* __restrict__ val,
                               const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = b[i];
  for (int64_t e = ptr[i]; e < ptr[i + 1]; ++e) v -= val[src[e]] * x[col[e]];
  r[i] = v;
}

__global__ void k_vec_add(int n, double* __restrict__ x, const double* __restrict__ r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += r[i];
}
// max |v_i|: non-negative doubles order like their bit patterns, so an integer atomicMax is exact and deterministic
__global__ void k_max_abs(int n, const double* __restrict__ v, unsigned long long* __restrict__ out) {
  double m = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmax(m, fabs(v[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}
hipError_t dev_vec_add(int n, double* d_x, const double* d_r, hipStream_t st) {
  hipLaunchKernelGGL(k_vec_add, dim3((n + 255) / 256), dim3(256), 0, st, n, d_x, d_r);
  return hipGetLastError();
}
hipError_t dev_max_abs(int n, const double* d_v, unsigned long long* d_out, hipStream_t st) {
  HIPCHK(hipMemsetAsync(d_out, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_max_abs, dim3(std::min((n + 255) / 256, 1024)), dim3(256), 0, st, n, d_v, d_out);
  return hipGetLastError();
}

void dev_free_coo(DeviceFactor& F) {
  void* ptrs[] = {F.mv_ptr, F.mv_src, F.rs_ptr, F.rs_col, F.rs_src, F.coo_val, F.valcsc, F.rbuf};
  for (void* p : ptrs) pool_free(F, p);
  F.mv_ptr = F.rs_ptr = nullptr;
  F.mv_src = F.rs_col = F.rs_src = nullptr;
  F.coo_val = F.valcsc = F.rbuf = nullptr;
  F.coo_ne = F.coo_nz = F.rbuf_cap = 0;
}

hipError_t dev_set_coo(DeviceFactor& F, int n, int64_t nzcsc, int64_t ne, const int32_t* row, const int32_t* col,
                       const int32_t* map, hipStream_t st) {
  PoolScope pool_scope(F);
  dev_free_coo(F);
  // by destination: the caller's entries of every CSC position, in entry order
  std::vector<int64_t> mptr(nzcsc + 1, 0);
  for (int64_t l = 0; l < ne; ++l) {
    const int64_t k = map[l] < 0 ? -int64_t(map[l]) : map[l];
    if (k >= 1 && k <= nzcsc) mptr[k]++;
  }
  for (int64_t k = 0; k < nzcsc; ++k) mptr[k + 1] += mptr[k];
  std::vector<int32_t> msrc(mptr[nzcsc]);
  {
    std::vector<int64_t> fill(mptr.begin(), mptr.end() - 1);
    for (int64_t l = 0; l < ne; ++l) {
      const int64_t k = map[l] < 0 ? -int64_t(map[l]) : map[l];
      if (k >= 1 && k <= nzcsc) msrc[fill[k - 1]++] = int32_t(l);
    }
  }
  HIPCHK(upload(F.mv_ptr, mptr, st));
  HIPCHK(upload(F.mv_src, msrc, st));
  F.coo_ne = ne;
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.coo_val), std::max<int64_t>(ne, 1) * sizeof(double)));
  HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.valcsc), std::max<int64_t>(nzcsc, 1) * sizeof(double)));
  F.nscatter_coo = nzcsc;
  if (row && col) {
    // the full symmetric matrix by rows (both triangles), out-of-range entries dropped (sls.f90:4901)
    std::vector<int64_t> rptr(n + 1, 0);
    auto ok = [&](int64_t l) { return row[l] >= 1 && row[l] <= n && col[l] >= 1 && col[l] <= n; };
    for (int64_t l = 0; l < ne; ++l)
      if (ok(l)) {
        rptr[row[l]]++;
        if (row[l] != col[l]) rptr[col[l]]++;
      }
    for (int i = 0; i < n; ++i) rptr[i + 1] += rptr[i];
    std::vector<int32_t> rcol(rptr[n]), rsrc(rptr[n]);
    std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
    for (int64_t l = 0; l < ne; ++l)
      if (ok(l)) {
        const int i = row[l] - 1, j = col[l] - 1;
        rcol[fill[i]] = j;
        rsrc[fill[i]++] = int32_t(l);
        if (i != j) {
          rcol[fill[j]] = i;
          rsrc[fill[j]++] = int32_t(l);
        }
      }
    HIPCHK(upload(F.rs_ptr, rptr, st));
    HIPCHK(upload(F.rs_col, rcol, st));
    HIPCHK(upload(F.rs_src, rsrc, st));
    F.coo_nz = rptr[n];
  }
  return hipStreamSynchronize(st);
}

__global__ void k_scale_values(int64_t n, double* __restrict__ v, double mult) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= mult;
}
// a stretch of the caller's values times a constant (gsls_set_value_part: SBLS stores -C in K)
hipError_t dev_scale_values(double* d_val, int64_t n, double mult, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scale_values, dim3(unsigned((n + 255) / 256)), dim3(256), 0, st, n, d_val, mult);
  return hipGetLastError();
}

hipError_t dev_map_values(DeviceFactor& F, const double* d_val_in, hipStream_t st) {
  if (F.nscatter_coo > 0)
    hipLaunchKernelGGL(k_map_values, dim3(unsigned((F.nscatter_coo + 255) / 256)), dim3(256), 0, st, F.nscatter_coo,
                       F.mv_ptr, F.mv_src, d_val_in, F.valcsc);
  return hipGetLastError();
}

hipError_t dev_residual(DeviceFactor& F, int n, int nrhs, const double* d_x, int ldx, const double* d_b, int ldb,
                        double* d_r, int ldr, hipStream_t st) {
  for (int k = 0; k < nrhs; ++k)
    hipLaunchKernelGGL(k_coo_residual, dim3((n + 255) / 256), dim3(256), 0, st, n, F.rs_ptr, F.rs_col, F.rs_src,
                       F.coo_val, d_x + int64_t(k) * ldx, d_b + int64_t(k) * ldb, d_r + int64_t(k) * ldr);
  return hipGetLastError();
}

// Fronts that go through k_front_tpp instead of the blocked kernels (host decision, gsls_api.cpp): a flag per
// node for the blocked kernels to skip them, and per level one node list per plan kind (0: all, 1: the
// subtrees this rank owns, 2: the top part).
hipError_t dev_set_tpp(const Symbolic& S, DeviceFactor& F, const std::vector<int>& nodes, hipStream_t st) {
  PoolScope pool_scope(F);
  const int nn = S.nnodes;
  std::vector<uint8_t> flag(std::max(nn, 1), 0);
  for (int s : nodes) flag[s] = 1;
  std::vector<int32_t> list;
  for (int which = 0; which < 3; ++which) {
    F.tpp_begin[which].assign(S.nlevels, 0);
    F.tpp_cnt[which].assign(S.nlevels, 0);
    if (nodes.empty()) { F.tpp_cnt[which].clear(); continue; }
    for (int l = 0; l < S.nlevels; ++l) {
      F.tpp_begin[which][l] = int(list.size());
      for (int i = S.lvlptr[l]; i < S.lvlptr[l + 1]; ++i) {
        const int s = S.lvlnodes[i];
        if (!flag[s]) continue;
        const bool mine = which == 0 || (F.sharded && (which == 1 ? S.owner[s] == F.myrank : S.owner[s] < 0));
        if (mine) list.push_back(s);
      }
      F.tpp_cnt[which][l] = int(list.size()) - F.tpp_begin[which][l];
    }
  }
  pool_free(F, F.tpplist);
  F.tpplist = nullptr;
  HIPCHK(upload(F.tpplist, list, st));
  HIPCHK(hipMemcpyAsync(F.tppflag, flag.data(), flag.size(), hipMemcpyHostToDevice, st));
  return hipStreamSynchronize(st);
}

// arena of the L11^-T blocks the Cholesky kernels exchange (one 64 x 64 block per 64 pivots)
static hipError_t ensure_linv(DeviceFactor& F) {
  if (F.Linv) return hipSuccess;
  return pool_alloc(reinterpret_cast<void**>(&F.Linv), std::max<int64_t>(F.nblk64, 1) * NB * NB * sizeof(double));
}

hipError_t dev_discover(const Symbolic& S, DeviceFactor& F, const double* d_val, double small, double u, hipStream_t st,
                        std::vector<int32_t>& seq, std::vector<uint8_t>& two, int& status, int& ndelayed) {
  PoolScope pool_scope(F);
  const int nn = S.nnodes, n = S.n;
  status = 1;
  ndelayed = 0;
  const int biggest = std::max(S.maxrow, S.maxfront);
  if (nn == 0 || F.sharded || biggest + DISC_WIDE_IN > DISC_WIDE_MAX || !F.asrc) {
    if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] discovery: tree not eligible (largest front %d rows)\n", biggest);
    return hipSuccess;
  }
  if (!F.disc_tasks) {
    std::vector<DiscTask> dt(nn);
    int64_t off = 0, woff = 0, voff = 0, poff = 0;
    for (int s = 0; s < nn; ++s) {
      DiscTask& t = dt[s];
      t.m = S.nrow(s);
      t.n = S.ncol(s);
      t.sptr = S.sptr[s];
      t.parent = S.sparent[s];
      t.cbeg = S.cptr[s];
      t.ccnt = S.cptr[s + 1] - S.cptr[s];
      t.moff = S.cmapptr[s];
      t.wide = t.m > DISC_WIDE_M ? 1 : 0;
      t.dcap = t.wide ? DISC_WIDE_OUT : DISC_DCAP;
      t.mcap = t.wide ? t.m + DISC_WIDE_IN : 64;
      t.qcap = (t.m - t.n) + t.dcap;
      t.a0 = S.nptr[s];
      t.acnt = int32_t(S.nptr[s + 1] - S.nptr[s]);
      t.coff = off;
      off += int64_t(t.qcap) * t.qcap;
      t.woff = woff;
      if (t.wide) woff += int64_t(t.mcap) * t.mcap;
      t.voff = voff;
      voff += t.dcap;
      t.poff = poff;
      poff += t.mcap;
    }
    F.disc_pcap = poff;
    if ((woff + off) * 8 > (int64_t(16) << 30)) {       // (trees of many mid-sized fronts: not what this sweep is for)
      if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] discovery: tree not eligible (%.1f GB of work matrices)\n", (woff + off) * 8e-9);
      return hipSuccess;
    }
    std::vector<uint32_t> arc(std::max<int64_t>(S.nptr[nn], 1));
    for (int s = 0; s < nn; ++s) {
      const int64_t m = S.nrow(s);
      for (int64_t k = S.nptr[s]; k < S.nptr[s + 1]; ++k) {
        const int64_t dst = S.nlist[2 * k + 1];
        arc[k] = uint32_t(dst % m) | (uint32_t(dst / m) << 16);
      }
    }
    // per level: the wave fronts first, then the workgroup fronts
    std::vector<int32_t> dl(S.lvlnodes.size());
    F.disc_lvl_wide.assign(S.nlevels, 0);
    for (int l = 0; l < S.nlevels; ++l) {
      int a = S.lvlptr[l], b = S.lvlptr[l + 1];
      int k = a;
      for (int i = a; i < b; ++i)
        if (!dt[S.lvlnodes[i]].wide) dl[k++] = S.lvlnodes[i];
      F.disc_lvl_wide[l] = b - k;
      for (int i = a; i < b; ++i)
        if (dt[S.lvlnodes[i]].wide) dl[k++] = S.lvlnodes[i];
    }
    DiscTask* d = nullptr;
    HIPCHK(upload(d, dt, st));
    F.disc_tasks = d;
    HIPCHK(upload(F.disc_arc, arc, st));
    HIPCHK(upload(F.disc_list, dl, st));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.disc_ddelay), size_t(nn) * sizeof(int32_t)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.disc_dvar), size_t(std::max<int64_t>(voff, 1)) * sizeof(int32_t)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.disc_pseq), size_t(std::max<int64_t>(poff, 1)) * sizeof(int32_t)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.disc_ptwo), size_t(std::max<int64_t>(poff, 1))));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.disc_pcnt), size_t(nn) * sizeof(int32_t)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.disc_flags), 4 * sizeof(int32_t)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.disc_arena), size_t(std::max<int64_t>(off, 1)) * sizeof(double)));
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.disc_scratch), size_t(std::max<int64_t>(woff, 1)) * sizeof(double)));
    F.disc_host.assign(dt.size() * 2, 0);        // (poff, mcap) per front for the read-back
    for (int s = 0; s < nn; ++s) { F.disc_host[2 * s] = dt[s].poff; F.disc_host[2 * s + 1] = dt[s].mcap; }
  }
  HIPCHK(hipMemsetAsync(F.disc_ddelay, 0, size_t(nn) * sizeof(int32_t), st));
  HIPCHK(hipMemsetAsync(F.disc_pcnt, 0, size_t(nn) * sizeof(int32_t), st));
  HIPCHK(hipMemsetAsync(F.disc_flags, 0, 4 * sizeof(int32_t), st));
  for (int l = 0; l < S.nlevels; ++l) {            // children before parents: the levels of the factorization plan
    const int cnt = S.lvlptr[l + 1] - S.lvlptr[l];
    const int nw = F.disc_lvl_wide[l], nv = cnt - nw;
    if (nv > 0)
      hipLaunchKernelGGL(k_front_discover, dim3((nv + 3) / 4), dim3(256), size_t(4) * 64 * DISC_LD * 8, st,
                         static_cast<const DiscTask*>(F.disc_tasks), F.disc_list + S.lvlptr[l], nv, F.clist, F.cmap, F.invp,
                         F.asrc, F.disc_arc, d_val, F.disc_arena, F.disc_ddelay, F.disc_dvar, F.disc_pseq, F.disc_ptwo,
                         F.disc_pcnt, F.disc_flags, small, u, nn);
    if (nw > 0)
      hipLaunchKernelGGL(k_front_discover_wg, dim3(nw), dim3(256), 0, st,
                         static_cast<const DiscTask*>(F.disc_tasks), F.disc_list + S.lvlptr[l] + nv, nw, F.clist, F.cmap,
                         F.invp, F.asrc, F.disc_arc, d_val, F.disc_arena, F.disc_scratch, F.disc_ddelay, F.disc_dvar,
                         F.disc_pseq, F.disc_ptwo, F.disc_pcnt, F.disc_flags, small, u, nn);
  }
  HIPCHK(hipGetLastError());
  int32_t fl[4];
  HIPCHK(hipMemcpyAsync(fl, F.disc_flags, sizeof(fl), hipMemcpyDeviceToHost, st));
  std::vector<int32_t> pc(nn);
  HIPCHK(hipMemcpyAsync(pc.data(), F.disc_pcnt, size_t(nn) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (fl[0] != 0) {                                // a front beyond its capacity / too many delays: not applicable
    if (getenv("GSLS_DEBUG")) fprintf(stderr, "[gsls] discovery: overflow (flags %d: 1 = a front beyond its row capacity with its delays, 2 = too many delays out of a front)\n", fl[0]);
    return hipSuccess;
  }
  int64_t tot = 0;
  for (int s = 0; s < nn; ++s) tot += pc[s];
  if (tot != n) return hipSuccess;                 // (cannot happen without the overflow flag; never trust it blindly)
  std::vector<int32_t> ps(size_t(F.disc_pcap));
  std::vector<uint8_t> pt(size_t(F.disc_pcap));
  HIPCHK(hipMemcpy(ps.data(), F.disc_pseq, ps.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(pt.data(), F.disc_ptwo, pt.size(), hipMemcpyDeviceToHost));
  seq.resize(n);
  two.assign(n, 0);
  std::vector<char> seen(n, 0);
  int64_t k = 0;
  for (int s = 0; s < nn; ++s) {                   // supernodes are numbered in postorder: children first
    const int64_t po = F.disc_host[2 * s];
    if (pc[s] > F.disc_host[2 * s + 1]) return hipSuccess;
    for (int j = 0; j < pc[s]; ++j) {
      const int v = ps[size_t(po + j)];
      if (v < 0 || v >= n || seen[v]) return hipSuccess;
      seen[v] = 1;
      seq[k] = v;
      two[k] = pt[size_t(po + j)];
      ++k;
    }
  }
  ndelayed = fl[1];
  status = 0;
  return hipSuccess;
}

hipError_t dev_factor(const Symbolic& S, DeviceFactor& F, bool posdef, const double* d_val,
                      const double* d_scale, double small, double u, hipStream_t st, bool use_tiny) {
  // with the wave-per-front plan only the fronts of the workgroup kernels take their entries of A through the
  // rectangle in HBM (k_front_wave gathers its own); blacklisted fronts are workgroup fronts again: full scatter
  const bool part = use_tiny && F.bl_count == 0;
  // pure: every front goes through the wave-per-front kernels (every front has its diagonal entry among A's, so no
  // scatter entries for workgroup fronts = no workgroup fronts).  Those kernels eliminate in the given order: they never
  // write gperm (it stays the identity), they write every pivot's two words of D and every front's image themselves.
  // So on a handle whose last pass was pure as well, the identity in gperm, the zeroing of D, the pack kernel and
  // gvar = invp o gperm are all in place already: four launches less per refactorization of the metric workload.
  const bool pure = part && !posdef && F.nscatter_wg == 0 && F.wave;
  const bool carry = pure && F.pure_state;
  F.cur_val = d_val;
  if (!part || F.nscatter_wg > 0)
    HIPCHK(hipMemsetAsync(F.L, 0, std::max<int64_t>(F.L_elems, 1) * sizeof(double), st));
  if (!carry) HIPCHK(hipMemsetAsync(F.D, 0, (2 * int64_t(S.n) + 4) * sizeof(double), st));
  static const std::vector<int32_t> init = [] { std::vector<int32_t> v(NSTAT, 0); v[0] = INT_MAX; return v; }();
  HIPCHK(hipMemcpyAsync(F.stat, init.data(), NSTAT * sizeof(int32_t), hipMemcpyHostToDevice, st));
  if (!carry) hipLaunchKernelGGL(k_iota, dim3((S.n + 255) / 256), dim3(256), 0, st, S.n, F.gperm);
  const int64_t nsc = part ? F.nscatter_wg : F.nscatter;
  if (nsc > 0) {
    const int blocks = int(std::min<int64_t>((nsc + 255) / 256, 256 * 8));
    hipLaunchKernelGGL(k_scatter_a, dim3(blocks), dim3(256), 0, st, nsc, part ? F.asrc_wg : F.asrc,
                       part ? F.adst_wg : F.adst, d_val, F.L, d_scale, F.arow, F.acol, F.invp);
  }
  if (posdef) {
    HIPCHK(ensure_linv(F));
    return factor_levels<true>(S, F, F.plan, small, u, st);
  }
  hipError_t e = factor_levels<false>(S, F, use_tiny ? F.planT : F.plan, small, u, st);
  if (e != hipSuccess) return e;
  if (F.wave && F.wtask_cnt > 0 && !pure)   // packed images of the wave-tier fronts the workgroup kernels factorized
    hipLaunchKernelGGL(k_wpack, dim3((F.wtask_cnt + 3) / 4), dim3(256), 0, st, static_cast<const WTask*>(F.wtasks),
                       static_cast<const WPack*>(F.wpacks), F.wtask_cnt, use_tiny ? 1 : 0, F.tinyskip, F.L, F.Lf, F.Lb);
  if (F.wave && !carry)
    hipLaunchKernelGGL(k_gvar, dim3((S.n + 255) / 256), dim3(256), 0, st, S.n, F.gperm, F.invp, F.gvar);
  F.pure_state = pure;
  return hipGetLastError();
}

static bool ws_flat_on() {
  static const bool on = !(getenv("GSLS_WS_FLAT") && atoi(getenv("GSLS_WS_FLAT")) == 0);   // (A/B knob)
  return on;
}
// the bottom stage's narrow launch reads the right-hand side from the caller's vector itself (no k_permute_in): whole
// solves of one column on the wave tier, no scaling vector
static bool can_fuse_input(const DeviceFactor& F, int job, bool scaled, int R) {
  static const bool off = getenv("GSLS_NO_FUSE_IN") != nullptr;
  return !off && F.wave && job == GSLS_SOLVE_JOB_ALL && !scaled && R == 1 && ws_flat_on() && F.wperm_list != nullptr &&
         !F.wstage_narrow.empty() && F.wstage_narrow[0] > 0 && !(F.wtail_k0 == 0);
}

// wave: the wave tier (LDL^T fronts of at most 64 rows) runs beside `plan`, which then holds the other fronts only;
// xin / xout: the caller's vector when the tier reads the right-hand side / writes the solution itself
template <bool POSDEF>
static hipError_t solve_sweeps(const Symbolic& S, DeviceFactor& F, const std::vector<LevelPlan>& plan,
                               int job, double* xp, hipStream_t st, hipEvent_t* ev, int diag_sel = -2,
                               bool wave = false, const double* xin = nullptr, double* xout = nullptr,
                               const double* scale = nullptr, const Cols* cols = nullptr) {
  const Cols cs = cols ? *cols : Cols{1, 0, 0, 0, 0, 0, 0, 0};
  const int R = cs.R;
  // the vectors a sweep works in: the handle's own, or (R columns at once) the first of R sets
  double* const w_xs = R > 1 ? F.mc_xs : F.xs;
  double* const w_cvec = R > 1 ? F.mc_cvec : F.cvec;
  double* const w_ybuf = R > 1 ? F.mc_ybuf : F.ybuf;
  double* const w_part = R > 1 ? F.mc_part : F.part;
  const bool do_fwd = (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_FWD);
  const bool do_diag = !POSDEF && (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_DIAG ||
                                   job == GSLS_SOLVE_JOB_DIAG_BWD);
  const bool do_bwd = (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_BWD ||
                       job == GSLS_SOLVE_JOB_DIAG_BWD);
  const BigTrsv* btr = static_cast<const BigTrsv*>(F.bigtrsv);
  const BigGemv* bgm = static_cast<const BigGemv*>(F.biggemv);
  const WGroup* wgr = static_cast<const WGroup*>(F.wgroups);
  const WTask* wtk = static_cast<const WTask*>(F.wtasks);
  const bool fuse_d = wave && do_fwd && do_diag;     // the tier applies D^-1 at the end of its forward step
  // job ALL: the tier's forward result stays by pivot slot (w_xs) for its own backward kernels
  double* slotv = (wave && do_fwd && do_bwd) ? w_xs : nullptr;
  // (experiment knob: unused dynamic LDS per workgroup of the narrow wave kernels = fewer waves per CU)
  static const size_t ws_pad = getenv("GSLS_WS_LDSPAD") ? size_t(atoi(getenv("GSLS_WS_LDSPAD"))) : 0;
  const bool ws_flat = ws_flat_on();
  // columns per wave in the narrow flat launches (several right-hand sides): 4, 2 or 1, whatever divides R
  static const int colgrp_max = getenv("GSLS_WS_COLGRP") ? atoi(getenv("GSLS_WS_COLGRP")) : 4;
  const int colgrp = (R > 1 && colgrp_max >= 4 && R % 4 == 0) ? 4 : (R > 1 && colgrp_max >= 2 && R % 2 == 0) ? 2 : 1;
  auto wave_fwd = [&](int g0, int cnt, bool narrow, int unit = -1, int depth = WSLOT) {
    if (cnt <= 0) return;
#define GSLS_WFWD(D_, N_, F_)                                                                                      \
  hipLaunchKernelGGL((k_wsolve_fwd<D_, N_, F_>), dim3(((cnt + 3) / 4) * R), dim3(256),                                        \
                     (N_) ? ws_pad : 0, st, wgr + g0, cnt, wtk, F.Lf, F.D, \
                     F.gperm, F.cmap, F.wgth_ptr, F.wgth_src, xp, slotv, w_cvec, cs, F.wpull2, unit, F.wimg_units)
    if (narrow && ws_flat && colgrp > 1) {
      // several columns per wave (template parameter CG): cs.R counts the column groups in these launches
      Cols cg = cs;
      cg.R = R / colgrp;
#define GSLS_WFWDG(D_, G_)                                                                                                    \
  hipLaunchKernelGGL((k_wsolve_fwd<D_, true, true, G_>), dim3(((cnt + 3) / 4) * cg.R), dim3(256),                                 \
                     size_t(4) * G_ * (depth * WACC_NARROW + 64) * 8, st, wgr + g0, cnt, wtk,                                     \
                     F.Lf, F.D, F.gperm, F.cmap, F.wgth_ptr, F.wgth_src, xp, slotv, w_cvec, cg, F.wpull2, unit, depth)
      if (colgrp == 4) { if (fuse_d) GSLS_WFWDG(true, 4); else GSLS_WFWDG(false, 4); }
      else { if (fuse_d) GSLS_WFWDG(true, 2); else GSLS_WFWDG(false, 2); }
#undef GSLS_WFWDG
      return;
    }
    if (fuse_d) { if (narrow) { if (ws_flat) GSLS_WFWD(true, true, true); else GSLS_WFWD(true, true, false); } else GSLS_WFWD(true, false, false); }
    else { if (narrow) { if (ws_flat) GSLS_WFWD(false, true, true); else GSLS_WFWD(false, true, false); } else GSLS_WFWD(false, false, false); }
#undef GSLS_WFWD
  };
  auto wave_bwd = [&](int g0, int cnt, bool narrow, int unit = -1, int depth = WSLOT) {
    if (cnt <= 0) return;
    const int wu = F.wimg_units;      // LDS staging area per wave of the wide launches (16-byte units)
    if (narrow && ws_flat && colgrp > 1) {
      Cols cg = cs;
      cg.R = R / colgrp;
      const size_t lds = size_t(4) * colgrp * (depth * WACC_NARROW + 64) * 8;
      if (colgrp == 4)
        hipLaunchKernelGGL((k_wsolve_bwd<true, true, 4>), dim3(((cnt + 3) / 4) * cg.R), dim3(256), lds, st, wgr + g0, cnt, wtk, F.Lf, F.gperm,
                           F.gvar, F.cmap, F.rlist, F.wgth_ptr, F.wgth_src, xp, slotv, xout, scale, w_cvec, cg, F.wpull2, unit, depth);
      else
        hipLaunchKernelGGL((k_wsolve_bwd<true, true, 2>), dim3(((cnt + 3) / 4) * cg.R), dim3(256), lds, st, wgr + g0, cnt, wtk, F.Lf, F.gperm,
                           F.gvar, F.cmap, F.rlist, F.wgth_ptr, F.wgth_src, xp, slotv, xout, scale, w_cvec, cg, F.wpull2, unit, depth);
      return;
    }
    if (narrow && ws_flat)
      hipLaunchKernelGGL((k_wsolve_bwd<true, true>), dim3(((cnt + 3) / 4) * R), dim3(256), ws_pad, st, wgr + g0, cnt, wtk, F.Lf, F.gperm,
                         F.gvar, F.cmap, F.rlist, F.wgth_ptr, F.wgth_src, xp, slotv, xout, scale, w_cvec, cs, F.wpull2, unit, wu);
    else if (narrow)
      hipLaunchKernelGGL(k_wsolve_bwd<true>, dim3(((cnt + 3) / 4) * R), dim3(256), ws_pad + size_t(4) * wu * 16, st, wgr + g0, cnt, wtk, F.Lf, F.gperm,
                         F.gvar, F.cmap, F.rlist, F.wgth_ptr, F.wgth_src, xp, slotv, xout, scale, w_cvec, cs, F.wpull2, unit, wu);
    else
      hipLaunchKernelGGL(k_wsolve_bwd<false>, dim3(((cnt + 3) / 4) * R), dim3(256), size_t(4) * wu * 16, st, wgr + g0, cnt, wtk, F.Lf, F.gperm,
                         F.gvar, F.cmap, F.rlist, F.wgth_ptr, F.wgth_src, xp, slotv, xout, scale, w_cvec, cs, F.wpull2, unit, wu);
  };
  if (ev) HIPCHK(hipEventRecord(ev[0], st));
  const bool ahead = slotv != nullptr && scale == nullptr;  // the look-ahead kernels: job ALL, no user scaling
  // the last stages in one launch (k_wsolve_tail); with nothing above the tier, both directions in the same launch
  const int nstage = int(F.wstage_cnt.size());
  const int ktail = (wave && F.wtail_k0 >= 1) ? F.wtail_k0 : nstage;
  const bool tail_both = ktail < nstage && do_fwd && do_bwd && F.wnont_cnt == 0;
  auto wave_tail = [&](int df, int db) {
    WTail tl;
    tl.nst = nstage - ktail;
    for (int k = 0; k < WTAIL_STAGES; ++k) {
      tl.gbeg[k] = k < tl.nst ? F.wstage_begin[ktail + k] : 0;
      tl.gcnt[k] = k < tl.nst ? F.wstage_cnt[ktail + k] : 0;
    }
    tl.tbeg = F.wtail_tbeg;
    tl.tcnt = F.wtail_tcnt;
    tl.lf0 = F.wtail_lf0; tl.lf1 = F.Lf_elems;
    tl.lb0 = 0; tl.lb1 = 0;
    tl.gp0 = F.wtail_gp0; tl.gp1 = F.wtail_gp1;
    tl.gs0 = F.wtail_gs0; tl.gs1 = F.wtail_gs1;
    if (fuse_d && df)
      hipLaunchKernelGGL(k_wsolve_tail<true>, dim3(R), dim3(64 * WTAIL_WAVES), size_t(WTAIL_WAVES) * F.wimg_units * 16, st, tl, wgr, wtk, F.Lf, F.Lf, F.D, F.gperm,
                         F.gvar, F.cmap, F.rlist, F.wgth_ptr, F.wgth_src, xp, slotv, xout, scale, w_cvec, cs, df, db, F.D, F.wpull2, F.wimg_units);
    else
      hipLaunchKernelGGL(k_wsolve_tail<false>, dim3(R), dim3(64 * WTAIL_WAVES), size_t(WTAIL_WAVES) * F.wimg_units * 16, st, tl, wgr, wtk, F.Lf, F.Lf, F.D, F.gperm,
                         F.gvar, F.cmap, F.rlist, F.wgth_ptr, F.wgth_src, xp, slotv, xout, scale, w_cvec, cs, df, db, F.D, F.wpull2, F.wimg_units);
  };
  if (do_fwd && wave) {
    for (int k = 0; k < ktail; ++k) {                       // stages of small subtrees, bottom-up
      const int nar = ahead ? F.wstage_narrow[k] : 0;
      if (k == 0 && xin && nar > 0) {        // (dev_solve asked can_fuse_input: fuse_d, ahead, one column, flat)
        const int nblk = (nar + 3) / 4;
        const WGather wgi{xin, F.invp, F.wperm_list, F.wperm_cnt, nblk};
        hipLaunchKernelGGL((k_wsolve_fwd<true, true, true, 1, true>), dim3(nblk + (F.wperm_cnt + 255) / 256), dim3(256), ws_pad, st,
                           wgr + F.wstage_begin[0], nar, wtk, F.Lf, F.D, F.gperm, F.cmap, F.wgth_ptr, F.wgth_src, xp, slotv,
                           w_cvec, cs, F.wpull2, -1, F.wimg_units, wgi);
      } else
      wave_fwd(F.wstage_begin[k], nar, true, -1, k < int(F.wstage_ndepth.size()) ? F.wstage_ndepth[k] : WSLOT);
      wave_fwd(F.wstage_begin[k] + nar, F.wstage_cnt[k] - nar, false, nar == 0 ? F.wstage_unit[k] : -1);
    }
    if (ktail < nstage) wave_tail(1, tail_both ? 1 : 0);
  }
  if (do_fwd)
    for (int l = 0; l < S.nlevels; ++l) {
      const LevelPlan& lp = plan[l];
      if (lp.small_cnt > 0) {
        if (POSDEF)
          hipLaunchKernelGGL(k_solve_fwd_chol, dim3(lp.small_cnt * R), dim3(256),
                             sizeof(double) * (((lp.small_maxm + 63) & ~63) + 256), st, F.nodes,
                             static_cast<const SolveTask*>(F.stasks) + lp.small_begin, F.clist, F.cmap, F.L, F.Linv,
                             xp, w_cvec, cs);
        else {
          const SolveTask* stk = static_cast<const SolveTask*>(F.stasks) + lp.small_begin;
          if (lp.tiny32_cnt > 0)
            hipLaunchKernelGGL(k_solve_fwd_tiny<32>, dim3(((lp.tiny32_cnt + 3) / 4) * R), dim3(256), 0, st, stk,
                               lp.tiny32_cnt, F.gth_ptr, F.gth_src, F.gperm, F.L, xp, w_cvec, cs);
          if (lp.tiny_cnt > lp.tiny32_cnt)
            hipLaunchKernelGGL(k_solve_fwd_tiny<64>, dim3(((lp.tiny_cnt - lp.tiny32_cnt + 3) / 4) * R), dim3(256), 0, st,
                               stk + lp.tiny32_cnt, lp.tiny_cnt - lp.tiny32_cnt, F.gth_ptr, F.gth_src, F.gperm, F.L, xp,
                               w_cvec, cs);
          if (lp.small_cnt > lp.tiny_cnt)
            hipLaunchKernelGGL(k_solve_fwd<false>, dim3((lp.small_cnt - lp.tiny_cnt) * R), dim3(256),
                               sizeof(double) * (64 * 65 + 2 * std::max(lp.small_maxn, 1)), st, F.nodes,
                               F.smallnodes + lp.small_begin + lp.tiny_cnt, F.clist, F.cmap, F.gperm, F.L, xp, w_cvec, cs);
        }
      }
      if (lp.big_cnt > 0) {
        hipLaunchKernelGGL(k_big_fwd_prep<POSDEF>, dim3((lp.big_cnt) * R), dim3(256), 0, st, F.nodes,
                           F.bignodes + lp.big_begin, F.clist, F.cmap, F.gperm, xp, w_cvec, w_ybuf, cs);
        for (const BigStep& bs : lp.bigsteps) {
          hipLaunchKernelGGL(k_big_fwd_trsv<POSDEF>, dim3((bs.trsv_cnt) * R), dim3(256), 0, st, F.nodes,
                             btr + bs.trsv_begin, bs.b, F.L, w_ybuf, cs);
          if (bs.gemv_cnt > 0)
            hipLaunchKernelGGL(k_big_fwd_gemv, dim3((bs.gemv_cnt) * R), dim3(256), 0, st, F.nodes,
                               bgm + bs.gemv_begin, bs.b, F.L, w_ybuf, w_cvec, cs);
        }
        hipLaunchKernelGGL(k_big_store<POSDEF>, dim3((lp.big_cnt) * R), dim3(256), 0, st, F.nodes,
                           F.bignodes + lp.big_begin, F.gperm, w_ybuf, xp, 0, cs);
      }
    }
  // the events between the phases cost the sweep ~10 us (a barrier packet each, the queue drains around them): only on
  // request; otherwise ev[0] .. ev[3] bracket the whole sweep
  static const bool phase_events = getenv("GSLS_SOLVE_PHASES") != nullptr;
  if (ev && phase_events) HIPCHK(hipEventRecord(ev[1], st));
  if (fuse_d) {
    if (F.wnont_cnt > 0)
      hipLaunchKernelGGL(k_solve_diag_nodes, dim3((F.wnont_cnt) * R), dim3(256), 0, st, F.nodes, F.wnont, F.D, F.gperm, xp, cs);
  } else if (do_diag) {
    if (diag_sel == -2)
      hipLaunchKernelGGL(k_solve_diag, dim3(((S.n + 255) / 256) * R), dim3(256), 0, st, S.n, F.D, F.gperm, xp, cs);
    else
      hipLaunchKernelGGL(k_solve_diag_owned, dim3((S.n + 255) / 256), dim3(256), 0, st, S.n, F.D, F.gperm,
                         F.posowner, diag_sel, xp);
  }
  if (ev && phase_events) HIPCHK(hipEventRecord(ev[2], st));
  if (do_bwd)
    for (int l = S.nlevels - 1; l >= 0; --l) {
      const LevelPlan& lp = plan[l];
      if (lp.big_cnt > 0) {
        hipLaunchKernelGGL(k_big_store<POSDEF>, dim3((lp.big_cnt) * R), dim3(256), 0, st, F.nodes,
                           F.bignodes + lp.big_begin, F.gperm, w_ybuf, xp, 1, cs);
        for (int k = int(lp.bigsteps.size()) - 1; k >= 0; --k) {
          const BigStep& bs = lp.bigsteps[k];
          if (bs.gemv_cnt > 0)
            hipLaunchKernelGGL(k_big_bwd_gemvT, dim3((bs.gemv_cnt) * R), dim3(256), 0, st, F.nodes,
                               bgm + bs.gemv_begin, bs.b, F.rlist, F.L, w_ybuf, xp, w_part, cs);
          hipLaunchKernelGGL(k_big_bwd_trsv<POSDEF>, dim3((bs.trsv_cnt) * R), dim3(256), 0, st, F.nodes,
                             btr + bs.trsv_begin, bs.b, F.L, w_ybuf, w_part, cs);
        }
        hipLaunchKernelGGL(k_big_store<POSDEF>, dim3((lp.big_cnt) * R), dim3(256), 0, st, F.nodes,
                           F.bignodes + lp.big_begin, F.gperm, w_ybuf, xp, 0, cs);
      }
      if (lp.small_cnt > 0) {
        if (POSDEF)
          hipLaunchKernelGGL(k_solve_bwd_chol, dim3(lp.small_cnt * R), dim3(256),
                             sizeof(double) * (64 * SB + 256 + 64 + std::max(lp.small_maxm, 1)), st,
                             static_cast<const SolveTask*>(F.stasks) + lp.small_begin, F.rlist, F.L, F.Linv, xp, cs);
        else {
          const SolveTask* stk = static_cast<const SolveTask*>(F.stasks) + lp.small_begin;
          if (lp.small_cnt > lp.tiny_cnt)
            hipLaunchKernelGGL(k_solve_bwd<false>, dim3((lp.small_cnt - lp.tiny_cnt) * R), dim3(256),
                               sizeof(double) * (64 * 65 + 256 + std::max(lp.small_maxm, 1)), st, F.nodes,
                               F.smallnodes + lp.small_begin + lp.tiny_cnt, F.rlist, F.gperm, F.L, xp, cs);
          if (lp.tiny_cnt > lp.tiny32_cnt)
            hipLaunchKernelGGL(k_solve_bwd_tiny<64>, dim3(((lp.tiny_cnt - lp.tiny32_cnt + 3) / 4) * R), dim3(256), 0, st,
                               stk + lp.tiny32_cnt, lp.tiny_cnt - lp.tiny32_cnt, F.rlist, F.gperm, F.L, xp, cs);
          if (lp.tiny32_cnt > 0)
            hipLaunchKernelGGL(k_solve_bwd_tiny<32>, dim3(((lp.tiny32_cnt + 3) / 4) * R), dim3(256), 0, st, stk,
                               lp.tiny32_cnt, F.rlist, F.gperm, F.L, xp, cs);
        }
      }
    }
  if (do_bwd && wave && ktail < nstage && !tail_both) wave_tail(0, 1);
  if (do_bwd && wave)
    for (int k = ktail - 1; k >= 0; --k) {
      const int nar = ahead ? F.wstage_narrow[k] : 0;
      wave_bwd(F.wstage_begin[k] + nar, F.wstage_cnt[k] - nar, false, nar == 0 ? F.wstage_unit[k] : -1);
      wave_bwd(F.wstage_begin[k], nar, true, -1, k < int(F.wstage_ndepth.size()) ? F.wstage_ndepth[k] : WSLOT);
    }
  if (ev) HIPCHK(hipEventRecord(ev[3], st));
  return hipGetLastError();
}

// Cholesky sweeps for R right-hand sides at once (every front of the plan on the one-workgroup kernels)
static size_t mr_lds_fwd(int maxm, int R) { return sizeof(double) * (size_t((maxm + 63) & ~63) * R + 256 * R); }
static size_t mr_lds_bwd(int maxm, int R) { return sizeof(double) * (64 * SB + 256 * R + 64 * R + size_t(std::max(maxm, 1)) * R); }
template <int R>
static hipError_t solve_sweeps_mr(const Symbolic& S, DeviceFactor& F, int job, hipStream_t st) {
  const bool do_fwd = (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_FWD);
  const bool do_bwd = (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_BWD || job == GSLS_SOLVE_JOB_DIAG_BWD);
  const int64_t xs = F.xs_mr, cs = F.cs_mr;
  const SolveTask* stk = static_cast<const SolveTask*>(F.stasks);
  if (do_fwd)
    for (int l = 0; l < S.nlevels; ++l) {
      const LevelPlan& lp = F.plan[l];
      if (lp.small_cnt > 0)
        hipLaunchKernelGGL(k_solve_fwd_chol_mr<R>, dim3(lp.small_cnt), dim3(256), mr_lds_fwd(lp.small_maxm, R), st, F.nodes,
                           stk + lp.small_begin, F.clist, F.cmap, F.L, F.Linv, F.xp_mr, F.cvec_mr, xs, cs);
    }
  if (do_bwd)
    for (int l = S.nlevels - 1; l >= 0; --l) {
      const LevelPlan& lp = F.plan[l];
      if (lp.small_cnt > 0)
        hipLaunchKernelGGL(k_solve_bwd_chol_mr<R>, dim3(lp.small_cnt), dim3(256), mr_lds_bwd(lp.small_maxm, R), st,
                           stk + lp.small_begin, F.rlist, F.L, F.Linv, F.xp_mr, xs);
    }
  return hipGetLastError();
}
// widest block the plan admits: 0 if some front needs the blocked multi-launch kernels or the panel does not fit in LDS
static int mr_width(const Symbolic& S, const DeviceFactor& F, int nrhs) {
  int maxm = 0;
  for (int l = 0; l < S.nlevels; ++l) {
    if (F.plan[l].big_cnt > 0) return 0;
    if (F.plan[l].small_cnt > 0) maxm = std::max(maxm, F.plan[l].small_maxm);
  }
  for (int R : {8, 4, 2})
    if (R <= nrhs && mr_lds_fwd(maxm, R) <= MR_LDS_CAP && mr_lds_bwd(maxm, R) <= MR_LDS_CAP) return R;
  return 0;
}

hipError_t dev_solve(const Symbolic& S, DeviceFactor& F, bool posdef, int job, int nrhs, double* d_x,
                     int ldx, const double* d_scale, hipStream_t st, hipEvent_t* ev, const double* d_b) {
  // d_b (optional): the right-hand sides live in another device array of the same shape and d_x only receives the
  // solution.  A whole solve of one column on the wave tier reads them from there directly (its bottom-stage launch
  // gathers the right-hand side anyway); every other path starts with a device-to-device copy and works in place.
  const bool b_direct = d_b && d_b != d_x && nrhs == 1 && !posdef && F.wave && S.n > 0 &&
                        can_fuse_input(F, job, d_scale != nullptr, 1);
  if (d_b && d_b != d_x && !b_direct && S.n > 0)
    HIPCHK(hipMemcpyAsync(d_x, d_b, (size_t(ldx) * (nrhs - 1) + S.n) * sizeof(double), hipMemcpyDeviceToDevice, st));
  int r0 = 0;
  if (posdef && nrhs >= 2 && S.n > 0 && !F.sharded && !getenv("GSLS_NO_MULTIRHS")) {
    // blocks of 8 / 4 / 2 columns through the multi-column kernels: one pass over L per block
    const int blocks = (S.n + 255) / 256;
    const bool scale_in = d_scale && (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_FWD);
    const bool scale_out = d_scale && (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_BWD ||
                                       job == GSLS_SOLVE_JOB_DIAG_BWD);
    while (true) {
      const int R = mr_width(S, F, nrhs - r0);
      if (R == 0) break;
      if (ev && r0 == 0) HIPCHK(hipEventRecord(ev[0], st));
      if (!F.xp_mr) {
        F.xs_mr = ((int64_t(S.n) + 64 + 15) / 16) * 16;
        F.cs_mr = ((std::max<int64_t>(F.cvec_elems, 1) + 64 + 15) / 16) * 16;
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.xp_mr), 8 * F.xs_mr * sizeof(double)));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.cvec_mr), 8 * F.cs_mr * sizeof(double)));
        HIPCHK(hipMemsetAsync(F.xp_mr, 0, 8 * F.xs_mr * sizeof(double), st));
        HIPCHK(hipMemsetAsync(F.cvec_mr, 0, 8 * F.cs_mr * sizeof(double), st));
      }
      for (int c = 0; c < R; ++c)
        hipLaunchKernelGGL(k_permute_in, dim3(blocks), dim3(256), 0, st, S.n, F.invp, d_x + int64_t(r0 + c) * ldx,
                           scale_in ? d_scale : nullptr, F.xp_mr + c * F.xs_mr);
      hipError_t e = R == 8 ? solve_sweeps_mr<8>(S, F, job, st)
                   : R == 4 ? solve_sweeps_mr<4>(S, F, job, st) : solve_sweeps_mr<2>(S, F, job, st);
      if (e != hipSuccess) return e;
      for (int c = 0; c < R; ++c)
        hipLaunchKernelGGL(k_permute_out, dim3(blocks), dim3(256), 0, st, S.n, F.invp, F.xp_mr + c * F.xs_mr,
                           scale_out ? d_scale : nullptr, d_x + int64_t(r0 + c) * ldx);
      r0 += R;
    }
    if (ev && r0 >= nrhs && r0 > 0) {      // (no single column follows: the events bracket the blocked sweeps)
      static const bool phase_events = getenv("GSLS_SOLVE_PHASES") != nullptr;
      for (int k = phase_events ? 1 : 3; k < 4; ++k) HIPCHK(hipEventRecord(ev[k], st));
    }
  }
  if (F.nrhs_cap < 1) {
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.xp), (std::max(S.n, 1) + 64) * sizeof(double)));
    HIPCHK(hipMemsetAsync(F.xp, 0, (std::max(S.n, 1) + 64) * sizeof(double), st));
    F.nrhs_cap = 1;
  }
  const int blocks = (S.n + 255) / 256;
  const bool wave = F.wave && !posdef;
  const bool has_bwd = (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_BWD || job == GSLS_SOLVE_JOB_DIAG_BWD);
  // which side of the permutation/scaling each job touches (fkeep.F90:229-318)
  const bool scale_in = d_scale && (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_FWD);
  const bool scale_out = d_scale && (job == GSLS_SOLVE_JOB_ALL || job == GSLS_SOLVE_JOB_BWD ||
                                     job == GSLS_SOLVE_JOB_DIAG_BWD);
  // the wave tier writes the solution into the caller's vector itself; the separate output permutation is only
  // needed for the fronts it does not cover and for the jobs without a backward sweep
  const bool fuse_out = wave && has_bwd;
  static const int mc_env = [] { const char* e = getenv("GSLS_SOLVE_COLS"); return e ? atoi(e) : 0; }();
  const int mc_max = (F.sharded || getenv("GSLS_NO_MULTIRHS")) ? 1
                     : std::max(1, std::min(mc_env > 0 ? mc_env : int(DeviceFactor::MC_MAX), int(DeviceFactor::MC_MAX)));
  for (int r = r0; r < nrhs && S.n > 0;) {
    const int R = std::min(nrhs - r, mc_max);
    double* x = d_x + int64_t(r) * ldx;
    hipEvent_t* evr = (r == r0) ? ev : nullptr;
    hipError_t e;
    if (R == 1) {
      const bool fuse_in = wave && can_fuse_input(F, job, d_scale != nullptr, 1);
      if (!fuse_in)
        hipLaunchKernelGGL(k_permute_in, dim3(blocks), dim3(256), 0, st, S.n, F.invp, x,
                           scale_in ? d_scale : nullptr, F.xp);
      e = posdef ? solve_sweeps<true>(S, F, F.plan, job, F.xp, st, evr)
          : wave ? solve_sweeps<false>(S, F, F.planW, job, F.xp, st, evr, -2, true, fuse_in ? (b_direct ? d_b : x) : nullptr,
                                       fuse_out ? x : nullptr, scale_out ? d_scale : nullptr)
                 : solve_sweeps<false>(S, F, F.plan, job, F.xp, st, evr);
      if (e != hipSuccess) return e;
      if (!fuse_out || F.wnont_cnt > 0)
        hipLaunchKernelGGL(k_permute_out, dim3(blocks), dim3(256), 0, st, S.n, F.invp, F.xp,
                           scale_out ? d_scale : nullptr, x);
    } else {
      // R columns through every launch (struct Cols): the same kernels, the same bits per column.  The Cholesky path
      // comes here only with the columns its blocked kernels could not take (fronts too tall for their LDS panel).
      if (!F.mc_xp) {
        constexpr int M = DeviceFactor::MC_MAX;
        F.mc_sx = ((int64_t(S.n) + 64 + 15) / 16) * 16;
        F.mc_scv = ((std::max<int64_t>(F.cvec_elems, 1) + 64 + 15) / 16) * 16;
        for (double** q : {&F.mc_xp, &F.mc_xs, &F.mc_ybuf}) {
          HIPCHK(pool_alloc(reinterpret_cast<void**>(q), M * F.mc_sx * sizeof(double)));
          HIPCHK(hipMemsetAsync(*q, 0, M * F.mc_sx * sizeof(double), st));
        }
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.mc_cvec), M * F.mc_scv * sizeof(double)));
        HIPCHK(hipMemsetAsync(F.mc_cvec, 0, M * F.mc_scv * sizeof(double), st));
        HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.mc_part), M * std::max<int64_t>(F.part_elems, 64) * sizeof(double)));
      }
      static const int cols_xcd = getenv("GSLS_COLS_XCD") ? atoi(getenv("GSLS_COLS_XCD")) : 1;
      const Cols cs{R, cols_xcd, F.mc_sx, F.mc_sx, F.mc_scv, F.mc_sx, std::max<int64_t>(F.part_elems, 64), int64_t(ldx)};
      hipLaunchKernelGGL(k_permute_in_cols, dim3(blocks, R), dim3(256), 0, st, S.n, F.invp, x, int64_t(ldx),
                         scale_in ? d_scale : nullptr, F.mc_xp, F.mc_sx);
      e = posdef ? solve_sweeps<true>(S, F, F.plan, job, F.mc_xp, st, evr, -2, false, nullptr, nullptr, nullptr, &cs)
          : wave ? solve_sweeps<false>(S, F, F.planW, job, F.mc_xp, st, evr, -2, true, nullptr,
                                       fuse_out ? x : nullptr, scale_out ? d_scale : nullptr, &cs)
                 : solve_sweeps<false>(S, F, F.plan, job, F.mc_xp, st, evr, -2, false, nullptr, nullptr, nullptr, &cs);
      if (e != hipSuccess) return e;
      if (!fuse_out || F.wnont_cnt > 0)
        hipLaunchKernelGGL(k_permute_out_cols, dim3(blocks, R), dim3(256), 0, st, S.n, F.invp, F.mc_xp, F.mc_sx,
                           scale_out ? d_scale : nullptr, x, int64_t(ldx));
    }
    r += R;
  }
  return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// multi-GPU (elimination-tree sharding).  Every rank holds the whole symbolic structure; rank r
// factorizes the subtrees it owns, the cut roots' contribution blocks are summed across ranks (each
// is non-zero on exactly one rank, so the sum is exact and order independent), rank 0 factorizes
// the top part.
// -------------------------------------------------------------------------------------------------
static void launch_segments(const DeviceFactor& F, const void* seg, double* arena, double* buf, int dir,
                            hipStream_t st) {
  if (F.nseg > 0)
    hipLaunchKernelGGL(k_segments, dim3(64, F.nseg), dim3(256), 0, st, static_cast<const Segment*>(seg), arena,
                       buf, dir, F.myrank);
}

hipError_t dev_shard_factor(const Symbolic& S, DeviceFactor& F, int phase, bool posdef, const double* d_val,
                            double* d_xchg, double small, double u, hipStream_t st, bool fast) {
  if (!F.sharded) return hipErrorInvalidValue;
  fast = fast && !posdef;
  F.pure_state = false;
  F.cur_val = d_val;
  if (phase == 1) {
    HIPCHK(hipMemsetAsync(F.L, 0, std::max<int64_t>(F.L_elems, 1) * sizeof(double), st));
    HIPCHK(hipMemsetAsync(F.D, 0, (2 * int64_t(S.n) + 4) * sizeof(double), st));
    static const std::vector<int32_t> init = [] { std::vector<int32_t> v(NSTAT, 0); v[0] = INT_MAX; return v; }();
    HIPCHK(hipMemcpyAsync(F.stat, init.data(), NSTAT * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_iota, dim3((S.n + 255) / 256), dim3(256), 0, st, S.n, F.gperm);
    if (F.nscatter > 0) {
      const int blocks = int(std::min<int64_t>((F.nscatter + 255) / 256, 256 * 8));
      hipLaunchKernelGGL(k_scatter_a, dim3(blocks), dim3(256), 0, st, F.nscatter, F.asrc, F.adst, d_val, F.L,
                         static_cast<const double*>(nullptr), F.arow, F.acol, F.invp);
    }
    if (posdef) HIPCHK(ensure_linv(F));
    hipError_t e = posdef ? factor_levels<true>(S, F, F.planA, small, u, st, 1)
                          : factor_levels<false>(S, F, fast ? F.planAT : F.planA, small, u, st, 1);
    if (e != hipSuccess) return e;
    launch_segments(F, F.segC, F.C, d_xchg, 0, st);
    hipLaunchKernelGGL(k_stat_to_xchg, dim3(1), dim3(64), 0, st, F.stat, d_xchg + F.xchgC_elems, static_cast<const double*>(nullptr));
    hipLaunchKernelGGL(k_stat_to_xchg, dim3(1), dim3(64), 0, st, F.stat, d_xchg + F.xchgC_elems + 16, static_cast<const double*>(nullptr));   // own copy (not reduced)
  } else if (phase == 2) {
    if (F.myrank != 0) return hipSuccess;
    launch_segments(F, F.segC, F.C, d_xchg, 1, st);
    hipError_t e = posdef ? factor_levels<true>(S, F, F.planB, small, u, st, 2, false)
                          : factor_levels<false>(S, F, fast ? F.planBT : F.planB, small, u, st, 2, false);
    if (e != hipSuccess) return e;
    // the top part's own counters = rank 0's counters now minus what they were after its subtrees
    hipLaunchKernelGGL(k_stat_to_xchg, dim3(1), dim3(64), 0, st, F.stat, d_xchg + F.xchgC_elems + 8,
                       static_cast<const double*>(d_xchg + F.xchgC_elems + 16));
    return hipGetLastError();
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// One full solve across the ranks; what crosses between them are the cut roots' vectors only:
// phase 1 (all): permute in, forward (+D) over my subtrees, pack the cut roots' contribution vectors
//          -> REDUCE d_xchg[0 : xchgV_elems) onto rank 0 (each entry is non-zero on one rank: exact, order-free)
// phase 2 (rank 0): unpack, forward / D / backward over the top part, pack the cut roots' z-vectors (the ancestors'
//          part of the solution below each cut) -> BROADCAST d_xchg[0 : xchgV_elems) from rank 0
// phase 3 (all): scatter my roots' z-vectors into xp, backward over my subtrees, write the solution of the
//          positions this rank computed into d_x (rank 0: also the top part's)
// phase 4 (optional, O(n), for callers that want the whole vector everywhere): d_xchg[0:n) = my part, zeros elsewhere
//          -> SUM d_xchg[0:n) over ranks, then phase 5: permute out d_xchg[0:n) -> d_x
hipError_t dev_shard_solve(const Symbolic& S, DeviceFactor& F, int phase, bool posdef, double* d_x,
                           double* d_xchg, hipStream_t st) {
  if (!F.sharded) return hipErrorInvalidValue;
  if (S.n == 0) return hipSuccess;
  if (F.nrhs_cap < 1) {
    HIPCHK(pool_alloc(reinterpret_cast<void**>(&F.xp), (std::max(S.n, 1) + 64) * sizeof(double)));
    HIPCHK(hipMemsetAsync(F.xp, 0, (std::max(S.n, 1) + 64) * sizeof(double), st));
    F.nrhs_cap = 1;
  }
  const int blocks = (S.n + 255) / 256;
  hipError_t e = hipSuccess;
  switch (phase) {
    case 1:
      hipLaunchKernelGGL(k_permute_in, dim3(blocks), dim3(256), 0, st, S.n, F.invp, d_x,
                         static_cast<const double*>(nullptr), F.xp);
      e = posdef ? solve_sweeps<true>(S, F, F.planA, GSLS_SOLVE_JOB_FWD, F.xp, st, nullptr)
                 : solve_sweeps<false>(S, F, F.planA, GSLS_SOLVE_JOB_FWD, F.xp, st, nullptr);
      if (e != hipSuccess) return e;
      if (!posdef)
        hipLaunchKernelGGL(k_solve_diag_owned, dim3(blocks), dim3(256), 0, st, S.n, F.D, F.gperm, F.posowner,
                           F.myrank, F.xp);
      launch_segments(F, F.segV, F.cvec, d_xchg, 0, st);
      break;
    case 2:
      if (F.myrank != 0) return hipSuccess;
      launch_segments(F, F.segV, F.cvec, d_xchg, 1, st);
      e = posdef ? solve_sweeps<true>(S, F, F.planB, GSLS_SOLVE_JOB_ALL, F.xp, st, nullptr, -1)
                 : solve_sweeps<false>(S, F, F.planB, GSLS_SOLVE_JOB_ALL, F.xp, st, nullptr, -1);
      if (e != hipSuccess) return e;
      if (F.nseg > 0)
        hipLaunchKernelGGL(k_zvec, dim3(8, F.nseg), dim3(256), 0, st, static_cast<const Segment*>(F.segZ), F.rlist,
                           F.xp, d_xchg, 0, F.myrank);
      break;
    case 3:
      if (F.nseg > 0)
        hipLaunchKernelGGL(k_zvec, dim3(8, F.nseg), dim3(256), 0, st, static_cast<const Segment*>(F.segZ), F.rlist,
                           F.xp, d_xchg, 1, F.myrank);
      e = posdef ? solve_sweeps<true>(S, F, F.planA, GSLS_SOLVE_JOB_BWD, F.xp, st, nullptr)
                 : solve_sweeps<false>(S, F, F.planA, GSLS_SOLVE_JOB_BWD, F.xp, st, nullptr);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_permute_out_owned, dim3(blocks), dim3(256), 0, st, S.n, F.invp, F.posowner, F.myrank, F.xp,
                         d_x);
      break;
    case 4:
      HIPCHK(hipMemcpyAsync(d_xchg, F.xp, S.n * sizeof(double), hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(k_xmask, dim3(blocks), dim3(256), 0, st, S.n, F.posowner, d_xchg,
                         static_cast<const double*>(nullptr), 1, F.myrank);
      break;
    case 5:
      hipLaunchKernelGGL(k_permute_out, dim3(blocks), dim3(256), 0, st, S.n, F.invp, d_xchg,
                         static_cast<const double*>(nullptr), d_x);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace gsls
