// micro-benchmark (not part of the product): f64 MFMA issue rate / latency and vector FMA rate on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
  double4_t c[CHAINS];
  for (int i = 0; i < CHAINS; ++i) c[i] = double4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < CHAINS; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters) {
  double c[CHAINS];
  for (int i = 0; i < CHAINS; ++i) c[i] = i;
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) c[i] = fma(c[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < CHAINS; ++i) s += c[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; hipMalloc(&d, 8 * 256 * 4096);
  const int iters = 4096;
  for (int blocks : {1, 256, 1024}) {
    float m1 = timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, d, iters); });
    float m8 = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, d, iters); });
    float f8 = timeit([&] { hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(256), 0, 0, d, iters); });
    const double fl_m8 = double(blocks) * 4 * 8 * iters * 2048.0, fl_f = double(blocks) * 256 * 16 * iters * 2.0;
    printf("blocks %4d: mfma dep-chain %.1f ns/op | 8 chains: %.1f ns per mfma per wave, %.2f TF/s | vector fma 16 chains: %.2f TF/s\n",
           blocks, m1 * 1e6 / iters, m8 * 1e6 / (iters * 8), fl_m8 / (m8 * 1e-3) / 1e12, fl_f / (f8 * 1e-3) / 1e12);
  }
  return 0;
}
