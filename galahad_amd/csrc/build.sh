#!/usr/bin/env bash
# Builds libgsls.so (HIP kernels + host orchestration + C ABI) for gfx950, in-tree.
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../libgsls.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="${GSLS_EXTRA:-} -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
mkdir -p $HERE/obj
pids=()
for f in gsls_symbolic.cpp gsls_order.cpp gsls_scaling.cpp gsls_api.cpp ; do
  $HIPCC $FLAGS -x hip -c $HERE/$f -o $HERE/obj/${f%.cpp}.o & pids+=($!)
done
$HIPCC $FLAGS -c $HERE/gsls_device.hip -o $HERE/obj/gsls_device.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC -shared -fPIC --offload-arch=gfx950 -o $OUT $HERE/obj/gsls_symbolic.o $HERE/obj/gsls_order.o $HERE/obj/gsls_scaling.o \
    $HERE/obj/gsls_api.o $HERE/obj/gsls_device.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built $OUT"
