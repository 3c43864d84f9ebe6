#!/bin/bash
# Diagnostic (timing-only / A-B) builds of libathena_mp: one source recompiled with an extra define.
#   scripts/build_variants.sh <source.hip> <tag> <-DNAME=value ...>   ->  variants/libathena_mp_<tag>.so
# Use: ATHENA_MP_LIB=$PWD/variants/libathena_mp_<tag>.so python bench.py ...
set -e
src=$1; tag=$2; shift 2
cd "$(dirname "$0")/../athena_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
mkdir -p ../../variants ../../build/obj_variants
/opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o ../../build/obj_variants/${src%.hip}_$tag.o
objs=$(ls ../../build/obj/*.o | grep -v "/${src%.hip}\.o$" | tr '\n' ' ')     # every stock object but this source's
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../build/obj_variants/${src%.hip}_$tag.o -o ../../variants/libathena_mp_$tag.so
echo "built variants/libathena_mp_$tag.so"
