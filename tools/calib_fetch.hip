// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes the gsls kernels use:
// known byte counts, streamed from a buffer far larger than the 256 MiB Infinity Cache.
//   k_read16 : 16 B per lane, consecutive lanes consecutive addresses   (the guide's calibrated case: counter = 1/2)
//   k_read8  :  8 B per lane, consecutive
//   k_gather8:  8 B per lane through an index array (random inside 4 KiB pages: every byte of the buffer once)
//   k_seg16  : 16 B per lane, 30 of 64 lanes active, segments starting at 16-B (not 128-B) aligned offsets --
//              the shape of the wave tier's image loads
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o tools/ubench_calib ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
typedef double double2_t __attribute__((ext_vector_type(2)));
__global__ void k_read16(const double2_t* __restrict__ p, size_t n2, double* out) {
  double acc = 0;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += size_t(gridDim.x) * blockDim.x) { double2_t v = p[i]; acc += v.x + v.y; }
  if (acc == 12345.678) out[0] = acc;
}
__global__ void k_read8(const double* __restrict__ p, size_t n, double* out) {
  double acc = 0;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc += p[i];
  if (acc == 12345.678) out[0] = acc;
}
__global__ void k_gather8(const double* __restrict__ p, const int* __restrict__ idx, size_t n, double* out) {
  double acc = 0;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) acc += p[(i & ~size_t(511)) + idx[i & 511]];
  if (acc == 12345.678) out[0] = acc;
}
// chunks of 30 double2 (480 B) back to back: wave w reads chunk w with lanes 0..29
__global__ void k_seg16(const double2_t* __restrict__ p, size_t nchunk, double* out) {
  double acc = 0;
  const int lane = threadIdx.x & 63;
  for (size_t c = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; c < nchunk; c += (size_t(gridDim.x) * blockDim.x) >> 6)
    if (lane < 30) { double2_t v = p[c * 30 + lane]; acc += v.x + v.y; }
  if (acc == 12345.678) out[0] = acc;
}
int main() {
  const size_t n = size_t(1) << 27;   // 1 GiB of doubles
  double *d, *out; int* idx;
  hipMalloc(&d, n * 8); hipMalloc(&out, 8); hipMalloc(&idx, 512 * 4);
  hipMemset(d, 0, n * 8);
  std::vector<int> h(512); std::iota(h.begin(), h.end(), 0); std::shuffle(h.begin(), h.end(), std::mt19937(1));
  hipMemcpy(idx, h.data(), 512 * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_read16, dim3(8192), dim3(256), 0, 0, (const double2_t*)d, n / 2, out);
    hipLaunchKernelGGL(k_read8, dim3(8192), dim3(256), 0, 0, d, n, out);
    hipLaunchKernelGGL(k_gather8, dim3(8192), dim3(256), 0, 0, d, idx, n, out);
    hipLaunchKernelGGL(k_seg16, dim3(8192), dim3(256), 0, 0, (const double2_t*)d, n / 60, out);
  }
  hipDeviceSynchronize();
  printf("bytes per kernel: read16 %zu read8 %zu gather8 %zu seg16 %zu\n", n * 8, n * 8, n * 8, (n / 60) * 480);
  return 0;
}
