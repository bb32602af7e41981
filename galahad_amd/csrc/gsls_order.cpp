// Fill-reducing ordering for the gsls backend (stands where SSIDS calls METIS,
// src/ssids/ssids.f90:305-320; METIS itself is a stub in the reference tree, src/dum/metis.f).
#include <numeric>

#include "gsls_internal.hpp"

namespace gsls {

void order_nested_dissection(int n, const std::vector<int64_t>& aptr, const std::vector<int>& arow,
                             std::vector<int>& perm) {
  (void)aptr;
  (void)arow;
  perm.resize(n);
  std::iota(perm.begin(), perm.end(), 0);
}

}  // namespace gsls
