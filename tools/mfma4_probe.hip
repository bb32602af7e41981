// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: which lane holds which element of A, B, D in each of the four blocks,
// and what cbsz / abid do.  One wave per (la, lb) pair: A = e_la, B = e_lb, C = 0 -> the lane(s) of D that become 1.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma4_probe.hip -o tools/ubench_mfma4 ; run: tools/ubench_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void probe(int* out) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x;          // la * 64 + lb
  const int la = pair >> 6, lb = pair & 63;
  const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
  unsigned long long m = __ballot(d != 0.0);
  if (lane == 0) { out[2 * pair] = int(m & 0xffffffffu); out[2 * pair + 1] = int(m >> 32); }
}
template <int CBSZ, int ABID>
void run(const char* tag) {
  int* d; hipMalloc(&d, 4096 * 2 * sizeof(int));
  hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(4096), dim3(64), 0, 0, d);
  std::vector<int> h(8192); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  printf("== %s\n", tag);
  // for the first block print, per A lane, which (B lane -> D lane) pairs exist
  for (int la = 0; la < 64; ++la) {
    int cnt = 0;
    for (int lb = 0; lb < 64; ++lb) {
      unsigned long long m = (unsigned long long)(unsigned)h[2 * (la * 64 + lb)] | ((unsigned long long)(unsigned)h[2 * (la * 64 + lb) + 1] << 32);
      if (!m) continue;
      if (la < 20 || la % 16 == 0) {
        printf("A lane %2d x B lane %2d -> D lanes:", la, lb);
        for (int l = 0; l < 64; ++l) if (m >> l & 1) printf(" %d", l);
        printf("\n");
      }
      ++cnt;
    }
    if (la >= 20 && la % 16 != 0) continue;
  }
  hipFree(d);
}
int main() {
  run<0, 0>("cbsz 0 abid 0");
  run<2, 1>("cbsz 2 abid 1 (A of block 1 broadcast to all four?)");
  return 0;
}
