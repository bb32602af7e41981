#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *real* reference (GALAHAD SLS + SPRAL SSIDS CPU backend)
# from the sources where they lie under /root/reference, for use as the parity checker and as the
# "reference" CPU baseline.  Nothing under galahad_amd/ may link or call what this produces.
#
# * sources are compiled in place (never copied into the repo); the throw-away header layout the
#   C++ files expect (the same one src/spral/makemaster:270-300 stages with cp + seds/have_hwloc_undef.sed)
#   lives in a scratch dir under /tmp and is deleted afterwards;
# * outputs go ONLY to oracle/_ref/ (git-ignored, but shipped to the GPU box by gpurun):
#       oracle/_ref/libgalahad_ref.so   reference SLS/SSIDS/LAPACK objects
#       oracle/_ref/ref_driver          our driver (oracle/ref_driver.f90) linked against it
# * the reference's own build system (bin/install_galahad, makemaster files) is NOT run.
#
# Toolchain notes: g++ for the C++11/OpenMP files (clang rejects default(none)
# uses in assemble.hxx), amdflang for Fortran; fkeep.F90 is a preprocessed source, so the `untied`
# task clause that makes amdflang-22 ICE is dropped with -Duntied= (semantics unchanged: an untied
# task may simply not migrate between threads).
set -euo pipefail

REF=${GSLS_REFERENCE_ROOT:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$REF/src/ssids" ]; then
  echo "build_ref: $REF not present -- keeping whatever prebuilt files are in $OUT" >&2
  exit 0
fi
# (development aid: GSLS_REF_SCRATCH=<dir> keeps the scratch directory and reuses the reference objects in it, so that
#  a change to the drop-in part below does not recompile LAPACK and SSIDS)
if [ -n "${GSLS_REF_SCRATCH:-}" ]; then
  W=$GSLS_REF_SCRATCH
  mkdir -p "$W"
else
  W=$(mktemp -d /tmp/gsls_ref_build.XXXXXX)
  trap 'rm -rf "$W"' EXIT
fi
mkdir -p "$OUT" "$W/inc/ssids/cpu/kernels" "$W/inc/hw_topology" "$W/mod" "$W/obj"
S=$REF/src

FC=${FC:-amdflang}
CXX=${CXX:-g++}
OPT=${GSLS_REF_OPT:--O2}
FFLAGS="$OPT -fopenmp -fPIC -module-dir $W/mod -I$W/mod"
CXXFLAGS="-std=c++11 $OPT -fopenmp -fPIC -I$W/inc"

if [ ! -f "$W/.reference_objects_done" ]; then
# ---- header layout (scratch only) -------------------------------------------------------------
cp $S/ssids/cpu/*.hxx            $W/inc/ssids/cpu/
cp $S/ssids/cpu/kernels/*.hxx    $W/inc/ssids/cpu/kernels/
cp $S/ssids/profile.hxx $S/ssids/contrib.h $W/inc/ssids/
cp $S/spral/omp.hxx $S/spral/compat.hxx   $W/inc/
cp $S/spral/guess_topology.hxx $S/spral/hwloc_wrapper.hxx $W/inc/hw_topology/
sed -f $REF/seds/have_hwloc_undef.sed $S/spral/config.h > $W/inc/config.h

# ---- C++ (SSIDS CPU numeric backend) -----------------------------------------------------------
pids=()
for f in spral/compat spral/omp spral/guess_topology ssids/profile \
         ssids/cpu/NumericSubtree ssids/cpu/SymbolicSubtree ssids/cpu/ThreadStats \
         ssids/cpu/kernels/cholesky ssids/cpu/kernels/ldlt_app ssids/cpu/kernels/ldlt_nopiv \
         ssids/cpu/kernels/ldlt_tpp ssids/cpu/kernels/wrappers ; do
  o=$W/obj/cxx_$(basename $f).o
  # -iquote- style trick: compile from the scratch inc dir so that quote-includes resolve to the
  # staged headers (and the hwloc-less config.h), not to the ones next to the source file.
  ( cd $W/inc && $CXX $CXXFLAGS -c -o $o -x c++ - < $S/$f.cxx ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done

# ---- Fortran, in dependency order ---------------------------------------------------------------
fc() { # fc <path relative to src> [extra flags]
  local src=$1; shift
  local o=$W/obj/f_$(echo $src | tr '/.' '__').o
  $FC $FFLAGS "$@" -c -o $o $S/$src
}
ff() { # fixed-form F77, compiled in parallel (no modules produced)
  local src=$1
  local o=$W/obj/f_$(echo $src | tr '/.' '__').o
  $FC $OPT -fPIC -module-dir $W/mod -I$W/mod -c -o $o $S/$src
}
# big F77 files first, in the background
ff lapack/blas.f & p1=$!
ff lapack/lapack.f & p2=$!
ff lapack/ieeeck.f & p3=$!
for f in spral/matrix_util.f90 spral/random.f90 spral/rutherford_boeing.f90 spral/cuda_nocuda.f90 \
         spral/hw_topology.f90 spral/metis4_wrapper.f90 spral/scaling.f90 spral/match_order.f90 \
         ssids/datatypes.f90 spral/core_analyse.f90 spral/pgm.f90 \
         ssids/inform.f90 ssids/contrib.f90 ssids/subtree.f90 ssids/akeep.f90 ssids/cpu_iface.f90 \
         ssids/cpu_subtree.f90 ssids/profile_iface.f90 ssids/gpu_subtree_no_cuda.f90 ssids/anal.f90 ; do
  fc $f
done
fc ssids/fkeep.F90 -cpp -Duntied=
fc ssids/ssids.f90
fc ssids/contrib_free.f90
for p in symbols clock string zd11 smt space specfile sort ; do fc $p/$p.f90 ; done
for f in dum/ma27d.f dum/mc61d.f dum/mc77d.f dum/mc64d.f dum/metis.f ; do ff $f ; done
fc lapack/blas_interface.f90
fc lapack/lapack_interface.f90
fc sils/sils.f90
for f in hsl_zb01i hsl_of01i hsl_of01d hsl_mc78i hsl_mc34d hsl_ma57d hsl_ma77d hsl_ma86d hsl_ma87d \
         hsl_ma97d hsl_mc64d hsl_mc68i ; do fc dum/$f.f90 ; done
fc non-free/mkl/mkl_pardiso_interface.f90
fc dum/mkl_pardiso.f90
fc dum/pardiso.f90
fc dum/wsmp.f90
fc sls/sls.f90
# SBLS (saddle-point / KKT layer above SLS) and TRS (trust-region subproblem) and what they USE;
# ULS/GLS only for SBLS's implicit variants
for f in lmt/lmt.f90 qpt/qpt.f90 roots/roots.f90 norms/norms.f90 gls/gls.f90 dum/hsl_ma48d.f90 \
         uls/uls.f90 sbls/sbls.f90 rand/rand.f90 ir/ir.f90 mop/mop.f90 trs/trs.f90 ; do fc $f ; done
for f in dum/ma33d.f dum/mc13d.f dum/mc21d.f dum/mc22d.f dum/mc23d.f dum/mc29d.f \
         dum/mc30d.f ; do ff $f ; done
wait $p1 $p2 $p3
touch "$W/.reference_objects_done"
fi

# ---- link ------------------------------------------------------------------------------------------
$FC -fopenmp -shared -o $OUT/libgalahad_ref.so $W/obj/*.o -lstdc++
$FC $FFLAGS -o $OUT/ref_driver $HERE/ref_driver.f90 -L$OUT -lgalahad_ref -Wl,-rpath,'$ORIGIN' -lstdc++
$FC $FFLAGS -o $OUT/sbls_driver $HERE/sbls_driver.f90 -L$OUT -lgalahad_ref -Wl,-rpath,'$ORIGIN' -lstdc++
$FC $FFLAGS -o $OUT/trs_driver $HERE/trs_driver.f90 -L$OUT -lgalahad_ref -Wl,-rpath,'$ORIGIN' -lstdc++
# second CPU baseline: the same reference library with the optimised OpenBLAS inside scipy behind SSIDS' seven BLAS /
# LAPACK calls (oracle/blas_shim.c, used through LD_PRELOAD by oracle/refio.py run(..., blas="openblas"))
OB=$(python3 - <<'PYEOF'
import glob, os
try:
    import scipy
    c = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    print(os.path.realpath(c[0]) if c else "")
except Exception:
    print("")
PYEOF
)
if [ -n "$OB" ]; then
  gcc -O2 -shared -fPIC -o $OUT/libblas_shim.so $HERE/blas_shim.c "$OB" -Wl,-rpath,"$(dirname "$OB")"
fi
echo "build_ref: wrote $OUT/libgalahad_ref.so and $OUT/ref_driver"

# ---- drop-in build: the REAL SLS facade with the gsls arms of INTEGRATION.md, linked to the MI355X
#      backend.  The patched sls.f90 exists only in the scratch dir. ---------------------------------
GSLS_LIB=$HERE/../galahad_amd/libgsls.so
if [ -f "$GSLS_LIB" ]; then
  mkdir -p $W/mod2
  python3 $HERE/../integration/patch_sls.py $S/sls/sls.f90 $W/sls_gsls_0.f90
  # the reference's default configuration (src/makedefs/packages.default:175, MA86_VERSION = ma86v2) passes sls.f90 and
  # the MA86 dummy through seds/ma86v2.sed; the C interface's dummy (dum/C/hsl_ma86d_ciface.f90) needs that variant
  sed -f $REF/seds/ma86v2.sed $W/sls_gsls_0.f90 > $W/sls_gsls.f90
  sed -f $REF/seds/ma86v2.sed $S/dum/hsl_ma86d.f90 > $W/hsl_ma86d_v2.f90
  $FC $OPT -fPIC -module-dir $W/mod2 -c -o $W/obj2_gsls_iface.o $HERE/../galahad_amd/fortran/gsls_iface.f90
  F2="$OPT -fopenmp -fPIC -module-dir $W/mod2 -I$W/mod2 -I$W/mod"
  $FC $F2 -c -o $W/obj2_hsl_ma86d_v2.o $W/hsl_ma86d_v2.f90
  $FC $F2 -c -o $W/obj2_sls_gsls.o $W/sls_gsls.f90
  $FC $F2 -o $OUT/sls_gsls_driver $HERE/ref_driver.f90 \
      $W/obj2_sls_gsls.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  # SBLS above the patched SLS: SBLS_solve_explicit leaves its refinement loop to the backend (integration/patch_sbls.py;
  # the patched copy exists only in the scratch dir)
  python3 $HERE/../integration/patch_sbls.py $S/sbls/sbls.f90 $W/sbls_gsls.f90
  $FC $F2 -c -o $W/obj2_sbls.o $W/sbls_gsls.f90
  $FC $F2 -o $OUT/sbls_gsls_driver $HERE/sbls_driver.f90 $W/obj2_sbls.o \
      $W/obj2_sls_gsls.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  # TRS (and the IR it calls) above the patched SLS
  $FC $F2 -c -o $W/obj2_ir.o $S/ir/ir.f90
  $FC $F2 -c -o $W/obj2_trs.o $S/trs/trs.f90
  $FC $F2 -o $OUT/trs_gsls_driver $HERE/trs_driver.f90 $W/obj2_trs.o $W/obj2_ir.o \
      $W/obj2_sls_gsls.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  # BASELINE.json configs[0]: CQP (interior point) -> SBLS -> SLS on QPBAND, above the patched facade (the solver is
  # chosen by name at run time: the reference's dense 'sytr' arm or 'gsls'); CQP's own dependencies in USE order
  for f in checkpoint/checkpoint lms/lms scu/scu cro/cro fdc/fdc fit/fit gltr/gltr lpqp/lpqp presolve/presolve \
           qpp/qpp trans/trans scale/scale qpd/qpd rpd/rpd cqp/cqp ; do
    $FC $F2 -c -o $W/obj2_q_$(basename $f).o $S/$f.f90
  done
  $FC $F2 -o $OUT/cqp_gsls_driver $HERE/cqp_driver.f90 $W/obj2_q_*.o $W/obj2_sbls.o \
      $W/obj2_sls_gsls.o $W/obj2_hsl_ma86d_v2.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  # the C interface of SLS (include/sls.h, src/sls/C/sls_ciface.f90) above the patched facade, and the reference's own C
  # test of it (src/sls/C/slst.c, compiled where it lies).  galahad_precision.h is staged exactly as the reference's
  # makefiles do (src/sls/makemaster:149: cp include/galahad_double.h $(OBJ)/galahad_precision.h).
  mkdir -p $W/cinc
  cp $REF/include/galahad_double.h $W/cinc/galahad_precision.h
  for f in common/C/common_ciface sils/C/sils_ciface dum/C/hsl_ma57d_ciface dum/C/hsl_ma77d_ciface \
           dum/C/hsl_ma86d_ciface dum/C/hsl_ma87d_ciface dum/C/hsl_ma97d_ciface dum/C/hsl_mc64d_ciface \
           dum/C/hsl_mc68i_ciface dum/C/ssids_ciface ; do
    $FC $F2 -c -o $W/obj2_$(basename $f).o $S/$f.f90
  done
  $FC $F2 -c -o $W/obj2_sls_ciface.o $S/sls/C/sls_ciface.f90
  gcc -O1 -c -o $W/obj2_wrap.o $HERE/ciface_solver_wrap.c
  for t in slst slstf ; do       # C (0-based) and Fortran (1-based) indexing variants of the reference's test
  gcc -O1 -I$W/cinc -I$REF/include -c -o $W/obj2_slst_c.o $S/sls/C/$t.c
  $FC $F2 -o $OUT/${t}_c_gsls $W/obj2_slst_c.o $W/obj2_wrap.o $W/obj2_sls_ciface.o $W/obj2_common_ciface.o \
      $W/obj2_sils_ciface.o $W/obj2_hsl_ma57d_ciface.o $W/obj2_hsl_ma77d_ciface.o $W/obj2_hsl_ma86d_ciface.o \
      $W/obj2_hsl_ma87d_ciface.o $W/obj2_hsl_ma97d_ciface.o $W/obj2_hsl_mc64d_ciface.o $W/obj2_hsl_mc68i_ciface.o \
      $W/obj2_ssids_ciface.o $W/obj2_sls_gsls.o $W/obj2_hsl_ma86d_v2.o $W/obj2_gsls_iface.o -Wl,--wrap=sls_initialize \
      -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' \
      -lstdc++ -lm 2>$W/link.log || { cat $W/link.log; exit 1; }
  done
  # SBLS's interface layer.  The C interface itself (src/sbls/C/sbls_ciface.f90 behind include/sbls.h) cannot be built
  # in this tree: it USEs GALAHAD_ULS_double_ciface -> HSL_MA48_double_ciface (src/dum/C/hsl_ma48d_ciface.f90:15,20),
  # which imports ma48_get_perm and ma48_determinant -- procedures the tree's own dummy src/dum/hsl_ma48d.f90 does not
  # define (HSL MA48 itself is not here, and no stand-in is written).  What sbls_ciface.f90 wraps, one call each, is
  # SBLS's "simple" interface (SBLS_import / SBLS_factorize_matrix / SBLS_solve_system / SBLS_information on
  # SBLS_full_data_type); the reference's own test of THAT layer, src/sbls/sblsti.f90 -- the same seven storage
  # schemes and data as src/sbls/C/sblst.c, expected residuals in src/sbls/sblsdt.output -- is built above the
  # patched SBLS and SLS, the solver named through oracle/select_solver.py.
  python3 $HERE/select_solver.py $S/sbls/sblsti.f90 $W/sblsti_sel.f90 SBLS
  $FC $F2 -o $OUT/sblsti_gsls $W/sblsti_sel.f90 $W/obj2_sbls.o \
      $W/obj2_sls_gsls.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  # The callers' own spec-sheet examples, whose stored outputs the reference holds (tests/golden/*.output are copies of
  # those data files): src/cqp/cqps.f90 (cqpds.output), src/rqs/rqss.f90 (rqsds.output), src/trs/trss.f90
  # (trsds.output), compiled where they lie with the solver named through oracle/select_solver.py
  $FC $F2 -c -o $W/obj2_rqs.o $S/rqs/rqs.f90
  python3 $HERE/select_solver.py $S/cqp/cqps.f90 $W/cqps_sel.f90 CQP
  python3 $HERE/select_solver.py $S/rqs/rqss.f90 $W/rqss_sel.f90 RQS
  python3 $HERE/select_solver.py $S/trs/trss.f90 $W/trss_sel.f90 TRS
  $FC $F2 -o $OUT/cqps_gsls $W/cqps_sel.f90 $W/obj2_q_*.o $W/obj2_sbls.o \
      $W/obj2_sls_gsls.o $W/obj2_hsl_ma86d_v2.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  $FC $F2 -o $OUT/rqss_gsls $W/rqss_sel.f90 $W/obj2_rqs.o $W/obj2_ir.o \
      $W/obj2_sls_gsls.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  $FC $F2 -o $OUT/trss_gsls $W/trss_sel.f90 $W/obj2_trs.o $W/obj2_ir.o \
      $W/obj2_sls_gsls.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  # BASELINE.json configs[0] as the reference runs it (src/cqp/makemaster:57 QPLIB_EXAMPLE = QPBAND.qplib; bin/dgal):
  # the reference's main program for QPLIB input, src/cqp/incqp.f90 (RPD reads the problem from unit 5, the solvers
  # come from the spec file RUNCQP.SPC in the working directory), compiled where it lies above the patched facade
  for f in copyright/copyright scaling/scaling ; do
    $FC $F2 -c -o $W/obj2_q2_$(basename $f).o $S/$f.f90
  done
  $FC $F2 -o $OUT/runcqp_qplib_gsls $S/cqp/incqp.f90 $W/obj2_q2_*.o $W/obj2_q_*.o $W/obj2_sbls.o \
      $W/obj2_sls_gsls.o $W/obj2_hsl_ma86d_v2.o $W/obj2_gsls_iface.o -L$OUT -lgalahad_ref -L$HERE/../galahad_amd -lgsls \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../galahad_amd' -lstdc++
  echo "build_ref: wrote $OUT/sls_gsls_driver, sbls_gsls_driver, trs_gsls_driver, slst_c_gsls, sblsti_gsls (GALAHAD SLS/SBLS/TRS + gsls backend)"
fi
