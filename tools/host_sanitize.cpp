// Host-only harness: symbolic analysis (three orderings, amalgamation 1..64, sharding), contribution-arena layouts and
// the three scalings on seeded random symmetric patterns under AddressSanitizer + UBSan (the GPU pool has no sanitizer):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Igalahad_amd/csrc tools/host_sanitize.cpp galahad_amd/csrc/gsls_order.cpp \
//       galahad_amd/csrc/gsls_symbolic.cpp galahad_amd/csrc/gsls_scaling.cpp -o /tmp/host_asan -pthread && /tmp/host_asan 60
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>
#include "gsls_internal.hpp"
using namespace gsls;
int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 40;
  std::mt19937 rng(12345);
  int bad = 0;
  for (int it = 0; it < cases; ++it) {
    const int n = 1 + int(rng() % (it % 7 == 0 ? 3000 : 400));
    const int deg = 1 + int(rng() % 6);
    const bool drop_diag = (it % 5 == 3);                 // some structurally zero diagonals / empty columns
    std::vector<std::set<int>> colrows(n);
    for (int j = 0; j < n; ++j) {
      if (!drop_diag || rng() % 4) colrows[j].insert(j);
      for (int k = 0; k < deg; ++k) {
        const int i = int(rng() % n);
        if (i > j) colrows[j].insert(i); else if (i < j) colrows[i].insert(j);
      }
    }
    std::vector<int64_t> ptr(n + 1, 1);
    std::vector<int32_t> row;
    std::vector<double> val;
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int j = 0; j < n; ++j) {
      for (int i : colrows[j]) { row.push_back(i + 1); val.push_back(i == j ? 4.0 + U(rng) : U(rng) * std::pow(10.0, 4 * U(rng))); }
      ptr[j + 1] = int64_t(row.size()) + 1;
    }
    for (int ordering : {GSLS_ORDER_ND, GSLS_ORDER_AMD, GSLS_ORDER_NATURAL}) {
      Symbolic S;
      std::vector<int32_t> order(n, 0);
      const int nemin = 1 << (rng() % 7);
      const int flag = symbolic_analyse(n, ptr.data(), row.data(), order.data(), ordering, nemin, S, nullptr);
      if (flag < 0) { printf("case %d ordering %d: flag %d\n", it, ordering, flag); ++bad; continue; }
      std::vector<char> seen(n, 0);
      for (int i = 0; i < n; ++i) {
        const int p = S.invp[i];
        if (p < 0 || p >= n || seen[p]) { printf("case %d ordering %d: invp is not a permutation\n", it, ordering); ++bad; break; }
        seen[p] = 1;
      }
      layout_contrib(S, true);
      layout_contrib(S, false);
      layout_contrib_auto(S);
      if (S.nnodes > 3) shard_tree(S, 2 + int(rng() % 3));
    }
    std::vector<double> sc(n, 0.0);
    std::vector<int64_t> ptr0(ptr);
    std::vector<int32_t> row0(row);
    for (auto& x : ptr0) --x;                              // the scalings take 0-based arrays
    for (auto& x : row0) --x;
    hungarian_scale_sym(n, ptr0.data(), row0.data(), val.data(), true, sc.data());
    auction_scale_sym(n, ptr0.data(), row0.data(), val.data(), sc.data());
    equilib_scale_sym(n, ptr0.data(), row0.data(), val.data(), sc.data());
    for (int i = 0; i < n; ++i)
      if (!(sc[i] > 0.0) || !std::isfinite(sc[i])) { printf("case %d: scaling entry %d = %g\n", it, i, sc[i]); ++bad; break; }
  }
  printf("host_sanitize: %d cases, %d failures\n", cases, bad);
  return bad != 0;
}
