#!/bin/bash
# Developer A/B helper: builds a variant of the library that differs only in the QP-ADMM translation unit.
#   tools/ab_build.sh NAME [extra hipcc flags, e.g. -DADMM_MINBLK=3]
# -> acg_alp_ldpc_amd/lib/variants/libacg_NAME.so ; select it with ACG_LDPC_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=acg_alp_ldpc_amd/csrc
mkdir -p acg_alp_ldpc_amd/lib/variants $src/_obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Iinclude -ffp-contract=off "$@" \
    -c $src/admm_kernels.hip -o $src/_obj/admm_kernels_$name.o
objs=$(ls $src/_obj/*.o | grep -v admm_kernels)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o acg_alp_ldpc_amd/lib/variants/libacg_$name.so $objs $src/_obj/admm_kernels_$name.o
echo acg_alp_ldpc_amd/lib/variants/libacg_$name.so
