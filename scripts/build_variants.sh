#!/bin/bash
# Diagnostic (timing-only) builds of libathena_mp with -DDUV_VARIANT=n; output build/libathena_mp_v<n>.so.
# Use: ATHENA_MP_LIB=build/libathena_mp_v1.so python scripts/bench_configs.py --config c3 --no-cpu
set -e
cd "$(dirname "$0")/../athena_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DDUV_VARIANT=$v -c duv_mfma.hip -o ../../build/obj/duv_mfma_v$v.o
  objs=$(ls ../../build/obj/*.o | grep -v "duv_mfma" | tr '\n' ' ')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../build/obj/duv_mfma_v$v.o -o ../../build/libathena_mp_v$v.so
done
