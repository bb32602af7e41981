// Micro-benchmark for the wave tier's bottom stage (round 3): which part of a front's load shape limits the
// forward sweep to ~2.6 TB/s?  Synthetic fronts of the metric workload's shape (n = 24 pivots, m = 26 / 30 rows,
// four fronts per run, 11 360 runs), the same instruction forms as k_wsolve_fwd<true,true>, and variants of
//   IMG   0: 16 partial 16-byte buffer loads per front, one per column pair (today)
//         1: 12 partial loads (no instruction for the column pairs beyond n)
//         2: the image as a flat stream -- ceil(bytes / 1024) full-wave 16-byte loads -- redistributed through LDS
//         3: flat stream, no redistribution (memory side alone; the arithmetic runs on whatever arrived)
//   SMALL 0: right-hand side, gperm, cmap, four D words, all 64 lanes unmasked (today)
//         1: masked to the lanes that use them, D as ONE 16-byte load
//         2: none
// Build: hipcc --offload-arch=gfx950 -O3 tools/wsolve_ubench.hip -o tools/ubench_wsolve
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double double2_t __attribute__((ext_vector_type(2)));
struct Task { int m, n, sptr, moff; long long lfoff; int pad[2]; };
static_assert(sizeof(Task) == 32, "");

__device__ __forceinline__ double readlane_f64(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

template <int IMG, int SMALL>
__global__ void __launch_bounds__(256)
k_fwd(const Task* __restrict__ tasks, int nrun, int runlen, const double* __restrict__ Lf,
      const double* __restrict__ D, const int* __restrict__ gperm, const int* __restrict__ cmap,
      const double* __restrict__ xp, double* __restrict__ slotv, double* __restrict__ cvec) {
  __shared__ __attribute__((aligned(16))) double img[4][IMG == 2 ? 512 : 2];
  __shared__ double accs[4][9 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ri = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (ri >= nrun) return;
  double* acc = accs[wave];
#pragma unroll
  for (int i = 0; i <= 8; ++i) acc[i * 64 + lane] = 0.0;
  const int oob = int(0x80000000);
  for (int ti = ri * runlen; ti < (ri + 1) * runlen; ++ti) {
    const Task t = tasks[__builtin_amdgcn_readfirstlane(ti)];
    const int m = t.m, n = t.n;
    const int npair = (n + 1) >> 1;
    const int nbytes = 16 * (npair * (m - 1) - npair * (npair - 1));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Lf + t.lfoff), 0, nbytes, 0x00020000);
    double2_t lp[16];
    if constexpr (IMG == 0 || IMG == 1) {
      const int v0 = (lane >= 1 && lane < m) ? (lane - 1) * 16 : oob;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (IMG == 1 && j >= 12) { lp[j] = double2_t{0.0, 0.0}; continue; }
        const bool ok = (lane >= 2 * j + 1) & (2 * j < n);
        lp[j] = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? v0 : oob, j * (m - j - 2) * 16, 0));
      }
    } else {
      // flat: chunk c covers bytes [1024 c, 1024 c + 1024); the descriptor's range check masks the tail
#pragma unroll
      for (int c = 0; c < 4; ++c)
        lp[c] = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, c * 1024, 0));
    }
    const long long s = (long long)t.sptr + lane;
    const bool piv = lane < n;
    double rhs = 0.0, d0 = 1.0, d1 = 0.0, dn = 0.0, dp = 0.0;
    int pslot = lane, prow = 0;
    if constexpr (SMALL == 0) {
      const double r = xp[s];
      const int gp = gperm[s];
      prow = cmap[t.moff + max(lane - n, 0)];
      rhs = piv ? r : 0.0;
      pslot = piv ? gp - t.sptr : lane;
      d0 = D[2 * s]; d1 = D[2 * s + 1]; dn = D[2 * s + 2]; dp = D[2 * s + 3 - 4 * (s > 0)];
    } else if constexpr (SMALL == 1) {
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(xp + t.sptr), 0, n * 8, 0x00020000);
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(gperm + t.sptr), 0, n * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(cmap + t.moff), 0, (m - n) * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(D + 2ll * t.sptr), 0, n * 16, 0x00020000);
      typedef unsigned int u2 __attribute__((ext_vector_type(2)));
      rhs = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, lane * 8, 0, 0));
      const int gp = __builtin_amdgcn_raw_buffer_load_b32(rg, lane * 4, 0, 0);
      prow = __builtin_amdgcn_raw_buffer_load_b32(rc, (lane - n) * 4, 0, 0);
      const double2_t dd = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rd, lane * 16, 0, 0));
      pslot = piv ? gp - t.sptr : lane;
      d0 = dd.x; d1 = dd.y;
      dn = __shfl_down(d0, 1);
      dp = __shfl_up(d1, 1);
    } else {
      rhs = piv ? 1.0 : 0.0;
    }
    if constexpr (IMG == 2) {
      double2_t* im = reinterpret_cast<double2_t*>(img[wave]);
#pragma unroll
      for (int c = 0; c < 4; ++c) im[c * 64 + lane] = lp[c];
      const int r1 = max(lane - 1, 0);
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const bool ok = (lane >= 2 * j + 1) & (2 * j < n) & (lane < m);
        const double2_t v = im[min(j * (m - j - 2) + r1, 255)];
        lp[j] = ok ? v : double2_t{0.0, 0.0};
      }
#pragma unroll
      for (int j = 12; j < 16; ++j) lp[j] = double2_t{0.0, 0.0};
    }
    double csum = acc[(ti & 7) * 64 + lane];
    acc[(ti & 7) * 64 + lane] = 0.0;
    double x = __shfl(rhs + csum, pslot);
    constexpr int NK = (IMG == 3) ? 8 : 32;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const double yk = readlane_f64(x, k);
      const double l = (k & 1) ? lp[k >> 1].y : lp[k >> 1].x;
      x = fma(-l, yk, x);
    }
    const bool crow = (lane >= n) & (lane < m);
    acc[((ti + 1) & 7) * 64 + (crow ? (prow & 63) : lane)] += crow ? x : 0.0;
    if (crow) cvec[t.moff + lane - n] = x;
    {
      const double yp = __shfl_up(x, 1), yn = __shfl_down(x, 1);
      if (isinf(d0)) x = fma(dp, yp, d1 * x);
      else if (isinf(dn)) x = fma(d0, x, d1 * yn);
      else x = x * d0;
    }
    if (lane < n) slotv[t.sptr + lane] = x;
  }
}

__global__ void k_readall(const double2_t* __restrict__ p, size_t n2, double* out) {
  double acc = 0;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += size_t(gridDim.x) * blockDim.x) { double2_t v = p[i]; acc += v.x + v.y; }
  if (acc == 12345.678) out[0] = acc;
}
static int g_mode = 0;
static void* g_flush = nullptr;
static size_t g_flush_bytes = size_t(768) << 20;
template <int IMG, int SMALL>
static void run(const char* name, const Task* tk, int nrun, int runlen, const double* Lf, const double* D, const int* gperm,
                const int* cmap, const double* xp, double* slotv, double* cvec, double bytes, int pad) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f, sum = 0;
  for (int rep = 0; rep < 7; ++rep) {
    // what the 256 MiB Infinity Cache holds when the sweep starts:
    //   1: 768 MB of OTHER data, dirty (memset)           5: 768 MB of other data, clean (read by a kernel)
    //   2: the image written (dirty), then 140 MB of other data written after it   (two images, the other one last)
    //   3: 140 MB of other data written, then the image written                  (two images, this one last)
    //   4: the image written and nothing else                                     (one image)
    if (g_mode == 1) hipMemsetAsync(g_flush, rep, g_flush_bytes, 0);
    if (g_mode == 5) hipLaunchKernelGGL(k_readall, dim3(4096), dim3(256), 0, 0, (const double2_t*)g_flush, g_flush_bytes / 16, (double*)g_flush);
    if (g_mode == 2) { hipMemsetAsync((void*)Lf, 0, (size_t)bytes, 0); hipMemsetAsync(g_flush, rep, (size_t)bytes, 0); }
    if (g_mode == 3) { hipMemsetAsync(g_flush, rep, (size_t)bytes, 0); hipMemsetAsync((void*)Lf, 0, (size_t)bytes, 0); }
    if (g_mode == 4) hipMemsetAsync((void*)Lf, 0, (size_t)bytes, 0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_fwd<IMG, SMALL>), dim3((nrun + 3) / 4), dim3(256), IMG == 2 ? 0 : pad, 0, tk, nrun, runlen, Lf, D, gperm, cmap, xp, slotv, cvec);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2) { best = std::min(best, ms); sum += ms; }
  }
  printf("%-28s best %7.1f us  mean %7.1f us   image stream %6.2f TB/s\n", name, best * 1e3, sum / 5 * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int runlen = argc > 1 ? atoi(argv[1]) : 4;
  const int nrun = argc > 2 ? atoi(argv[2]) : 11360;
  const int pad = argc > 3 ? atoi(argv[3]) : 0;   // unused dynamic LDS for the variants without the LDS image (equal occupancy)
  g_mode = argc > 4 ? atoi(argv[4]) : 0;
  if (g_mode) { hipMalloc(&g_flush, g_flush_bytes); hipMemset(g_flush, 0, g_flush_bytes); }
  const int nf = nrun * runlen;
  std::vector<Task> tk(nf);
  long long of = 0;
  int sp = 0, mo = 0;
  for (int i = 0; i < nf; ++i) {
    Task& t = tk[i];
    t.n = 24;
    t.m = (i & 1) ? 30 : 26;
    t.sptr = sp;
    t.moff = mo;
    t.lfoff = of;
    const int npair = 12;
    of += 2ll * (npair * (t.m - 1) - npair * (npair - 1));
    sp += t.n;
    mo += t.m - t.n;
  }
  double *Lf, *D, *xp, *slotv, *cvec;
  int *gperm, *cmap;
  Task* dtk;
  hipMalloc(&Lf, (of + 1024) * 8);
  hipMalloc(&D, (2ll * sp + 256) * 8);
  hipMalloc(&xp, (sp + 128) * 8);
  hipMalloc(&slotv, (sp + 128) * 8);
  hipMalloc(&cvec, (mo + 128) * 8);
  hipMalloc(&gperm, (sp + 128) * 4);
  hipMalloc(&cmap, (mo + 128) * 4);
  hipMalloc(&dtk, sizeof(Task) * nf);
  hipMemset(Lf, 0, (of + 1024) * 8);
  std::vector<double> hd(2ll * sp + 256, 1.0);
  for (long long i = 0; i < sp; ++i) hd[2 * i + 1] = 0.0;
  hipMemcpy(D, hd.data(), hd.size() * 8, hipMemcpyHostToDevice);
  std::vector<int> hg(sp + 128);
  for (int i = 0; i < sp + 128; ++i) hg[i] = i;
  hipMemcpy(gperm, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
  std::vector<int> hc(mo + 128);
  for (int i = 0; i < mo + 128; ++i) hc[i] = 24 + (i % 6);
  hipMemcpy(cmap, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
  hipMemset(xp, 0, (sp + 128) * 8);
  hipMemcpy(dtk, tk.data(), sizeof(Task) * nf, hipMemcpyHostToDevice);
  const double bytes = of * 8.0;
  printf("fronts %d, runs %d x %d, image %.1f MB, unknowns %d, LDS pad %d, cache mode %d\n", nf, nrun, runlen, bytes / 1e6, sp, pad, g_mode);
#define RUN(I, S) run<I, S>("IMG " #I " SMALL " #S, dtk, nrun, runlen, Lf, D, gperm, cmap, xp, slotv, cvec, bytes, pad)
  RUN(0, 0); RUN(0, 1); RUN(0, 2);
  RUN(1, 0); RUN(1, 1); RUN(1, 2);
  RUN(2, 0); RUN(2, 1); RUN(2, 2);
  RUN(3, 0); RUN(3, 1); RUN(3, 2);
  hipDeviceSynchronize();
  return 0;
}
