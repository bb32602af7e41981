// Host-only harness: the threaded nested dissection (galahad_amd/csrc/gsls_order.cpp) on a 2-D grid under the sanitizers
// (the GPU pool has none; the host code is where the threads are).  One thread against six: same permutation, no reports.
//   g++ -std=c++17 -O1 -g -fsanitize=thread -Igalahad_amd/csrc tools/nd_sanitize.cpp galahad_amd/csrc/gsls_order.cpp -o /tmp/nd_tsan -pthread && /tmp/nd_tsan 700
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Igalahad_amd/csrc tools/nd_sanitize.cpp galahad_amd/csrc/gsls_order.cpp -o /tmp/nd_asan -pthread && /tmp/nd_asan 520
// Round 2: both clean at n = 250 000 ... 490 000.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gsls_internal.hpp"
using namespace gsls;
int main(int argc, char** argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 500, ny = nx, n = nx * ny;
  std::vector<int64_t> ap(n + 1, 0);
  std::vector<int> ar;
  for (int y = 0; y < ny; ++y)
    for (int x = 0; x < nx; ++x) {
      const int v = y * nx + x;
      if (x > 0) ar.push_back(v - 1);
      if (x + 1 < nx) ar.push_back(v + 1);
      if (y > 0) ar.push_back(v - nx);
      if (y + 1 < ny) ar.push_back(v + nx);
      ap[v + 1] = int64_t(ar.size());
    }
  std::vector<int> p1, p4;
  setenv("GSLS_ND_THREADS", "1", 1);
  order_nested_dissection(n, ap, ar, p1);
  setenv("GSLS_ND_THREADS", "6", 1);
  order_nested_dissection(n, ap, ar, p4);
  std::vector<char> seen(n, 0);
  for (int v = 0; v < n; ++v) {
    if (p4[v] < 0 || p4[v] >= n || seen[p4[v]]) { printf("not a permutation at %d\n", v); return 1; }
    seen[p4[v]] = 1;
  }
  printf("n %d: %s\n", n, p1 == p4 ? "identical" : "DIFFERENT");
  return p1 == p4 ? 0 : 2;
}
