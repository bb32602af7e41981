/* TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Second CPU baseline: the reference
 * library oracle/_ref/libgalahad_ref.so carries GALAHAD's vendored reference BLAS / LAPACK (src/lapack/blas.f,
 * lapack.f).  Preloaded (LD_PRELOAD=oracle/_ref/libblas_shim.so), this shim takes over the seven routines SSIDS' CPU
 * kernels call (src/ssids/cpu/kernels/wrappers.cxx:11-19) and hands them to the optimised OpenBLAS that ships inside
 * scipy (scipy.libs/libscipy_openblas*.so exports them with a scipy_ prefix) -- the only optimised BLAS on this image.
 * Nothing under galahad_amd/ links or loads this. */
void scipy_dgemm_(char*, char*, int*, int*, int*, double*, const double*, int*, const double*, int*, double*, double*, int*);
void scipy_dpotrf_(char*, int*, double*, int*, int*);
void scipy_dsytrf_(char*, int*, double*, int*, int*, double*, int*, int*);
void scipy_dtrsm_(char*, char*, char*, char*, int*, int*, const double*, const double*, int*, double*, int*);
void scipy_dsyrk_(char*, char*, int*, int*, double*, const double*, int*, double*, double*, int*);
void scipy_dtrsv_(char*, char*, char*, int*, const double*, int*, double*, int*);
void scipy_dgemv_(char*, int*, int*, const double*, const double*, int*, const double*, int*, const double*, double*, int*);

void dgemm_(char* ta, char* tb, int* m, int* n, int* k, double* alpha, const double* a, int* lda, const double* b,
            int* ldb, double* beta, double* c, int* ldc) {
  scipy_dgemm_(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
}
void dpotrf_(char* uplo, int* n, double* a, int* lda, int* info) { scipy_dpotrf_(uplo, n, a, lda, info); }
void dsytrf_(char* uplo, int* n, double* a, int* lda, int* ipiv, double* work, int* lwork, int* info) {
  scipy_dsytrf_(uplo, n, a, lda, ipiv, work, lwork, info);
}
void dtrsm_(char* side, char* uplo, char* ta, char* diag, int* m, int* n, const double* alpha, const double* a, int* lda,
            double* b, int* ldb) {
  scipy_dtrsm_(side, uplo, ta, diag, m, n, alpha, a, lda, b, ldb);
}
void dsyrk_(char* uplo, char* trans, int* n, int* k, double* alpha, const double* a, int* lda, double* beta, double* c,
            int* ldc) {
  scipy_dsyrk_(uplo, trans, n, k, alpha, a, lda, beta, c, ldc);
}
void dtrsv_(char* uplo, char* trans, char* diag, int* n, const double* a, int* lda, double* x, int* incx) {
  scipy_dtrsv_(uplo, trans, diag, n, a, lda, x, incx);
}
void dgemv_(char* trans, int* m, int* n, const double* alpha, const double* a, int* lda, const double* x, int* incx,
            const double* beta, double* y, int* incy) {
  scipy_dgemv_(trans, m, n, alpha, a, lda, x, incx, beta, y, incy);
}
