// Prototype for DESIGN.md section f-3 (not part of the product): LDL^T of a 32 x 32 symmetric matrix by ONE wave with the
// matrix held as 4 x 4 tiles in the operand layout of v_mfma_f64_4x4x4_4b_f64, against the lane-per-row elimination that
// k_front_wave uses today (two v_readlane + one FMA per updated entry).  Both kernels repeat the factorization REP times
// on registers (no memory traffic inside the timed loop): what is compared is the instruction stream per front.
//
// Layout ("transposed D"): register R[J][h] (tile column J = 0..7, h = 0, 1) holds the four row tiles I = 4h + b,
// b = 0..3; lane 16 i + 4 b + j holds element (row 4 I + j, column 4 J + i).  With that
//   * R[p][h] IS the B operand of the panel solve and of the trailing update (B(k, j) at lane 16 k + 4 b + j),
//   * the result lands in the same layout (D(i, j) at lane 16 i + 4 b + j),
//   * the A operand is a 4 x 4 matrix shared by the four blocks: M = D^-1 L_pp^-1 for the panel (from the scalar 4 x 4
//     factorization), -W_Jp = -L_Jp D for the update of tile column J (block b_J of R[p][h_J] replicated to all four
//     blocks: two ds_bpermute_b32).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_front_proto.hip -o tools/ubench_mfma_front ; run: tools/ubench_mfma_front
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s:%d\n", int(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double readlane_f64(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double inv_f64(double d) {      // reciprocal + two Newton steps, as k_front_wave
  double rd = __builtin_amdgcn_rcp(d);
  rd = fma(fma(-d, rd, 1.0), rd, rd);
  return fma(fma(-d, rd, 1.0), rd, rd);
}
__device__ __forceinline__ double bperm_f64(double v, int srclane) {
  const int lo = __builtin_amdgcn_ds_bpermute(srclane * 4, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(srclane * 4, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// ---- today's form: lane = row, the 32 columns in registers, pivots and multipliers by v_readlane ----
template <int REP>
__global__ void __launch_bounds__(256) k_rows(const double* __restrict__ A, double* __restrict__ Lo, double* __restrict__ Do, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * 4 + wave;
  if (f >= N) return;
  // the matrix comes from memory in every repetition (no register copy: the product kernel has none either); the same
  // 64 matrices for all waves, so the loads hit in L2
  double v[32], myd = 0.0, chk = 0.0;
  for (int rep = 0; rep < REP; ++rep) {
    const double* a = A + size_t((f + rep) & 63) * 1024;     // a different matrix per repetition
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = (lane < 32 && lane >= k) ? a[k * 32 + lane] : 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const double d = readlane_f64(v[j], j);
      const double rd = inv_f64(d);
      const double own = v[j] * rd, um = v[j];
#pragma unroll
      for (int k = j + 1; k < 32; ++k) v[k] = fma(-own, readlane_f64(um, k), v[k]);
      v[j] = own;
      if (lane == j) myd = d;
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) chk += v[k];      // (every repetition must be computed in full)
    chk += myd;
  }
  if (lane < 32) {
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (lane > k) Lo[size_t(f) * 1024 + k * 32 + lane] = v[k];
    Do[size_t(f) * 32 + lane] = myd + ((chk == 12345.678) ? 1.0 : 0.0);
  }
}

// ---- tiles in MFMA layout ----
template <int REP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_tiles(const double* __restrict__ A, double* __restrict__ Lo, double* __restrict__ Do, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * 4 + wave;
  if (f >= N) return;
  const int li = lane >> 4, lb = (lane >> 2) & 3, lj = lane & 3;
  double R[8][2], dsave[8] = {0, 0, 0, 0, 0, 0, 0, 0}, chk = 0.0;   // dsave[p]: lane k < 4 holds pivot 4 p + k
  for (int rep = 0; rep < REP; ++rep) {
    const double* a = A + size_t((f + rep) & 63) * 1024;
#pragma unroll
    for (int J = 0; J < 8; ++J)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = 4 * (4 * h + lb) + lj, col = 4 * J + li;
        R[J][h] = ((row >= col) ? a[col * 32 + row] : a[row * 32 + col]);
      }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int hp = p >> 2, bp = p & 3;
      const double T = R[p][hp];
      // the diagonal tile, lower triangle: element (r, c) of the tile sits at lane 16 c + 4 bp + r
      const double t00 = readlane_f64(T, 4 * bp + 0), t10 = readlane_f64(T, 4 * bp + 1), t20 = readlane_f64(T, 4 * bp + 2),
                   t30 = readlane_f64(T, 4 * bp + 3), t11 = readlane_f64(T, 16 + 4 * bp + 1), t21 = readlane_f64(T, 16 + 4 * bp + 2),
                   t31 = readlane_f64(T, 16 + 4 * bp + 3), t22 = readlane_f64(T, 32 + 4 * bp + 2), t32 = readlane_f64(T, 32 + 4 * bp + 3),
                   t33 = readlane_f64(T, 48 + 4 * bp + 3);
      // scalar LDL^T of the 4 x 4 tile (wave-uniform values)
      const double d0 = t00, r0 = inv_f64(d0);
      const double l10 = t10 * r0, l20 = t20 * r0, l30 = t30 * r0;
      const double d1 = fma(-l10, t10, t11), r1 = inv_f64(d1);
      const double u21 = fma(-l20, t10, t21), u31 = fma(-l30, t10, t31);
      const double l21 = u21 * r1, l31 = u31 * r1;
      const double d2 = fma(-l21, u21, fma(-l20, t20, t22)), r2 = inv_f64(d2);
      const double u32 = fma(-l31, u21, fma(-l30, t20, t32));
      const double l32 = u32 * r2;
      const double d3 = fma(-l32, u32, fma(-l31, u31, fma(-l30, t30, t33))), r3 = inv_f64(d3);
      // inverse of the unit lower factor, M = D^-1 L^-1
      const double i10 = -l10, i21 = -l21, i32 = -l32;
      const double i20 = fma(-l21, i10, -l20), i31 = fma(-l32, i21, -l31);
      const double i30 = fma(-l32, i20, fma(-l31, i10, -l30));
      // A operand of the panel solve: lane 16 k + 4 b + i holds M(i, k)   (i = lj, k = li)
      double Mrow_k0 = (lj == 0) ? r0 : (lj == 1) ? i10 * r1 : (lj == 2) ? i20 * r2 : i30 * r3;
      double Mrow_k1 = (lj == 0) ? 0.0 : (lj == 1) ? r1 : (lj == 2) ? i21 * r2 : i31 * r3;
      double Mrow_k2 = (lj <= 1) ? 0.0 : (lj == 2) ? r2 : i32 * r3;
      double Mrow_k3 = (lj <= 2) ? 0.0 : r3;
      const double Mop = (li == 0) ? Mrow_k0 : (li == 1) ? Mrow_k1 : (li == 2) ? Mrow_k2 : Mrow_k3;
      const double dk = (li == 0) ? d0 : (li == 1) ? d1 : (li == 2) ? d2 : d3;   // d_k for k = lane / 16
      dsave[p] = (lane == 0) ? d0 : (lane == 1) ? d1 : (lane == 2) ? d2 : d3;
      // panel: (L_Ip)^T = M (F_Ip)^T for the row tiles of tile column p (those above the diagonal come out as by-products)
#pragma unroll
      for (int h = hp; h < 2; ++h) R[p][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(Mop, R[p][h], 0.0, 0, 0, 0);
      // trailing update, tile column by tile column: R[J][h] -= (W_Jp)(L_Ip)^T transposed, W = L D
#pragma unroll
      for (int J = p + 1; J < 8; ++J) {
        const int hJ = J >> 2, bJ = J & 3;
        const double Aop = -bperm_f64(R[p][hJ], (lane & 0x33) | (bJ << 2)) * dk;
#pragma unroll
        for (int h = hJ; h < 2; ++h) R[J][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(Aop, R[p][h], R[J][h], 0, 0, 0);
      }
    }
#pragma unroll
    for (int J = 0; J < 8; ++J) chk += R[J][0] + R[J][1] + dsave[J];
  }
#pragma unroll
  for (int J = 0; J < 8; ++J)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 4 * (4 * h + lb) + lj, col = 4 * J + li;
      if (row > col) Lo[size_t(f) * 1024 + col * 32 + row] = R[J][h];
    }
  if (lane < 4) {
#pragma unroll
    for (int p = 0; p < 8; ++p) Do[size_t(f) * 32 + 4 * p + lane] = dsave[p] + ((chk == 12345.678) ? 1.0 : 0.0);
  }
}

// ---- tiles in MFMA layout, the shared A operand through a 20-entry LDS table that one lane writes (no selects) ----
template <int REP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k_tiles_lds(const double* __restrict__ A, double* __restrict__ Lo, double* __restrict__ Do, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.x * 4 + wave;
  if (f >= N) return;
  const int li = lane >> 4, lb = (lane >> 2) & 3, lj = lane & 3;
  __shared__ __attribute__((aligned(16))) double tabs[4][24];   // per wave: M(i, k) at [4 k + i], d_k at [16 + k]
  double* tab = tabs[wave];
  if (lane < 24) tab[lane] = 0.0;
  double R[8][2], dsave[8] = {0, 0, 0, 0, 0, 0, 0, 0}, chk = 0.0;   // dsave[p]: lane k < 4 holds pivot 4 p + k
  for (int rep = 0; rep < REP; ++rep) {
    const double* a = A + size_t((f + rep) & 63) * 1024;
#pragma unroll
    for (int J = 0; J < 8; ++J)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = 4 * (4 * h + lb) + lj, col = 4 * J + li;
        R[J][h] = ((row >= col) ? a[col * 32 + row] : a[row * 32 + col]);
      }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int hp = p >> 2, bp = p & 3;
      const double T = R[p][hp];
      // the diagonal tile, lower triangle: element (r, c) of the tile sits at lane 16 c + 4 bp + r
      const double t00 = readlane_f64(T, 4 * bp + 0), t10 = readlane_f64(T, 4 * bp + 1), t20 = readlane_f64(T, 4 * bp + 2),
                   t30 = readlane_f64(T, 4 * bp + 3), t11 = readlane_f64(T, 16 + 4 * bp + 1), t21 = readlane_f64(T, 16 + 4 * bp + 2),
                   t31 = readlane_f64(T, 16 + 4 * bp + 3), t22 = readlane_f64(T, 32 + 4 * bp + 2), t32 = readlane_f64(T, 32 + 4 * bp + 3),
                   t33 = readlane_f64(T, 48 + 4 * bp + 3);
      // scalar LDL^T of the 4 x 4 tile (wave-uniform values)
      const double d0 = t00, r0 = inv_f64(d0);
      const double l10 = t10 * r0, l20 = t20 * r0, l30 = t30 * r0;
      const double d1 = fma(-l10, t10, t11), r1 = inv_f64(d1);
      const double u21 = fma(-l20, t10, t21), u31 = fma(-l30, t10, t31);
      const double l21 = u21 * r1, l31 = u31 * r1;
      const double d2 = fma(-l21, u21, fma(-l20, t20, t22)), r2 = inv_f64(d2);
      const double u32 = fma(-l31, u21, fma(-l30, t20, t32));
      const double l32 = u32 * r2;
      const double d3 = fma(-l32, u32, fma(-l31, u31, fma(-l30, t30, t33))), r3 = inv_f64(d3);
      // inverse of the unit lower factor, M = D^-1 L^-1
      const double i10 = -l10, i21 = -l21, i32 = -l32;
      const double i20 = fma(-l21, i10, -l20), i31 = fma(-l32, i21, -l31);
      const double i30 = fma(-l32, i20, fma(-l31, i10, -l30));
      // A operand of the panel solve: lane 16 k + 4 b + i holds M(i, k)   (i = lj, k = li): through the table
      if (lane == 0) {
        tab[0] = r0; tab[1] = i10 * r1; tab[2] = i20 * r2; tab[3] = i30 * r3;
        tab[5] = r1; tab[6] = i21 * r2; tab[7] = i31 * r3;
        tab[10] = r2; tab[11] = i32 * r3;
        tab[15] = r3;
        tab[16] = d0; tab[17] = d1; tab[18] = d2; tab[19] = d3;
      }
      __builtin_amdgcn_wave_barrier();
      const double Mop = tab[4 * li + lj];
      const double dk = tab[16 + li];
      __builtin_amdgcn_wave_barrier();
      dsave[p] = tab[16 + (lane & 3)];
      // panel: (L_Ip)^T = M (F_Ip)^T for the row tiles of tile column p (those above the diagonal come out as by-products)
#pragma unroll
      for (int h = hp; h < 2; ++h) R[p][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(Mop, R[p][h], 0.0, 0, 0, 0);
      // trailing update, tile column by tile column: R[J][h] -= (W_Jp)(L_Ip)^T transposed, W = L D
#pragma unroll
      for (int J = p + 1; J < 8; ++J) {
        const int hJ = J >> 2, bJ = J & 3;
        const double Aop = -bperm_f64(R[p][hJ], (lane & 0x33) | (bJ << 2)) * dk;
#pragma unroll
        for (int h = hJ; h < 2; ++h) R[J][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(Aop, R[p][h], R[J][h], 0, 0, 0);
      }
    }
#pragma unroll
    for (int J = 0; J < 8; ++J) chk += R[J][0] + R[J][1] + dsave[J];
  }
#pragma unroll
  for (int J = 0; J < 8; ++J)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 4 * (4 * h + lb) + lj, col = 4 * J + li;
      if (row > col) Lo[size_t(f) * 1024 + col * 32 + row] = R[J][h];
    }
  if (lane < 4) {
#pragma unroll
    for (int p = 0; p < 8; ++p) Do[size_t(f) * 32 + 4 * p + lane] = dsave[p] + ((chk == 12345.678) ? 1.0 : 0.0);
  }
}

static void cpu_ldlt(const double* a, std::vector<double>& L, std::vector<double>& D) {
  std::vector<double> w(a, a + 1024);
  L.assign(1024, 0.0); D.assign(32, 0.0);
  for (int j = 0; j < 32; ++j) {
    const double d = w[j * 32 + j];
    D[j] = d;
    for (int i = j + 1; i < 32; ++i) L[j * 32 + i] = w[j * 32 + i] / d;
    for (int k = j + 1; k < 32; ++k)
      for (int i = k; i < 32; ++i) w[k * 32 + i] -= L[j * 32 + i] * w[j * 32 + k];
  }
}

template <typename K>
static double run(K kern, const char* name, const double* dA, double* dL, double* dD, int N, int rep, const std::vector<double>& hA) {
  CHK(hipMemset(dL, 0, size_t(N) * 1024 * 8));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3((N + 3) / 4), dim3(256), 0, 0, dA, dL, dD, N);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0, 0));
  for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3((N + 3) / 4), dim3(256), 0, 0, dA, dL, dD, N);
  CHK(hipEventRecord(e1, 0));
  CHK(hipDeviceSynchronize());
  float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  // check the first eight fronts
  std::vector<double> hL(size_t(8) * 1024), hD(8 * 32), L, D;
  CHK(hipMemcpy(hL.data(), dL, hL.size() * 8, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(hD.data(), dD, hD.size() * 8, hipMemcpyDeviceToHost));
  double err = 0;
  for (int f = 0; f < 8; ++f) {
    cpu_ldlt(hA.data() + size_t((f + rep - 1) & 63) * 1024, L, D);     // the matrix of the last repetition
    for (int j = 0; j < 32; ++j) {
      err = std::fmax(err, std::fabs(hD[f * 32 + j] - D[j]) / std::fabs(D[j]));
      for (int i = j + 1; i < 32; ++i) err = std::fmax(err, std::fabs(hL[size_t(f) * 1024 + j * 32 + i] - L[j * 32 + i]));
    }
  }
  const double per = double(ms) * 1e3 / (double(N) * rep);      // us of whole-GPU time per front
  printf("%-8s %8.3f ms for %d fronts x %d repetitions: %.4f ns of GPU time per front = %.0f SIMD-cycles per front at 2.4 GHz and 1024 SIMDs; max error vs CPU %.2e\n",
         name, ms, N, rep, per * 1e3, per * 1e-6 * 2.4e9 * 1024, err);
  return per;
}

int main() {
  const int N = 20480, REP = 32;        // 20 waves per CU on 256 CUs = 5120 at a time: four full rounds
  std::vector<double> hA(size_t(N) * 1024);
  srand(7);
  for (int f = 0; f < N; ++f) {
    double* a = hA.data() + size_t(f) * 1024;
    for (int j = 0; j < 32; ++j)
      for (int i = j; i < 32; ++i) {
        const double v = (i == j) ? ((j % 3 == 1) ? -1.0 : 1.0) * (8.0 + rand() / double(RAND_MAX)) : (rand() / double(RAND_MAX) - 0.5) * 0.5;
        a[j * 32 + i] = v; a[i * 32 + j] = v;
      }
  }
  double *dA, *dL, *dD;
  CHK(hipMalloc(&dA, hA.size() * 8)); CHK(hipMalloc(&dL, hA.size() * 8)); CHK(hipMalloc(&dD, size_t(N) * 32 * 8));
  CHK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
  const double t1 = run(k_rows<REP>, "rows", dA, dL, dD, N, REP, hA);
  const double t2 = run(k_tiles<REP>, "tiles", dA, dL, dD, N, REP, hA);
  const double t3 = run(k_tiles_lds<REP>, "tiles+lds", dA, dL, dD, N, REP, hA);
  printf("tiles / rows = %.2f, tiles with the LDS table / rows = %.2f\n", t2 / t1, t3 / t1);
  return 0;
}
