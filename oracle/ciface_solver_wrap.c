/* TEST INFRASTRUCTURE ONLY.  The reference's C test src/sls/C/slst.c names its solver in a string literal
 * (sls_initialize( "sils", ... ), slst.c:46).  The test is compiled from where it lies, unmodified; this shim is
 * linked with -Wl,--wrap=sls_initialize and substitutes the solver named by $GSLS_CTEST_SOLVER (oracle/build_ref.sh,
 * tests/test_sls_dropin.py::test_reference_c_interface_test_with_gsls). */
#include <stdlib.h>

void __real_sls_initialize(const char solver[], void **data, void *control, int *status);

void __wrap_sls_initialize(const char solver[], void **data, void *control, int *status) {
  const char *s = getenv("GSLS_CTEST_SOLVER");
  __real_sls_initialize(s ? s : solver, data, control, status);
}
