#!/bin/bash
# Developer A/B helper: a variant of the library that differs in some translation units built with extra flags.
#   tools/ab_variant.sh NAME "api bp_streamed" -DACG_RING_SLOTS=3
# -> acg_alp_ldpc_amd/lib/variants/libacg_NAME.so ; select it with ACG_LDPC_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
name=$1; tus=$2; shift 2
src=acg_alp_ldpc_amd/csrc
mkdir -p acg_alp_ldpc_amd/lib/variants $src/_obj
objs=""
skip=""
for tu in $tus; do
  f=$src/$tu.hip; x=""
  if [ ! -f $f ]; then f=$src/$tu.cpp; x="-x hip"; fi
  extra=""; [ $tu = admm_kernels ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc $x -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Iinclude $extra "$@" -c $f -o $src/_obj/${tu}__$name.o &
  objs="$objs $src/_obj/${tu}__$name.o"
  skip="$skip|/$tu.o"
done
wait
rest=$(ls $src/_obj/*.o | grep -v "__" | grep -Ev "${skip#|}")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o acg_alp_ldpc_amd/lib/variants/libacg_$name.so $rest $objs
echo acg_alp_ldpc_amd/lib/variants/libacg_$name.so
