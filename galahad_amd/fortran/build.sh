#!/usr/bin/env bash
# Compiles the ISO_C_BINDING module and its KAT driver against libgsls.so (amdflang).
set -euo pipefail
HERE=$(cd "$(dirname "$0")" && pwd)
FC=${FC:-amdflang}
mkdir -p $HERE/obj
$FC -O2 -fPIC -module-dir $HERE/obj -c $HERE/gsls_iface.f90 -o $HERE/obj/gsls_iface.o
$FC -O2 -I$HERE/obj -o $HERE/gsls_kat $HERE/gsls_kat.f90 $HERE/obj/gsls_iface.o \
    -L$HERE/.. -lgsls -Wl,-rpath,'$ORIGIN/..'
echo "built $HERE/gsls_kat"
