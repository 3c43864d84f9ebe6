#!/bin/bash
# Builds athena_amd/libathena_mp.so for gfx950 (MI355X) -- no other target.
# -ffp-contract=off: a*b+c stays a rounded multiply + rounded add unless the source says fmaf/MFMA,
# so the aggregation kernels reproduce the reference's strict fp32 sums bit for bit.
set -e
cd "$(dirname "$0")"
OUT=../libathena_mp.so
SRCS="capi.hip graph_build.hip agg.hip banded_fused.hip gemm.hip gemm_tiled.hip fused.hip fused_dw.hip elementwise.hip duvenaud.hip duv_mfma.hip readout.hip gno.hip train.hip host.hip comm.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
mkdir -p ../../build/obj
objs=""
pids=""
for s in $SRCS; do
  [ -f "$s" ] || continue
  o=../../build/obj/${s%.hip}.o
  objs="$objs $o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ transport.h -nt "$o" ] || [ radix_sort.h -nt "$o" ] || [ ../../include/athena_mp.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc $FLAGS -c "$s" -o "$o" &
    pids="$pids $!"
  fi
done
fail=0
for p in $pids; do wait $p || fail=1; done     # a failed compile must not link the stale object of the last good build
[ $fail -eq 0 ] || { echo "build.sh: compilation failed" >&2; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
# the TEST transport of comm.hip: a plugin of its own, never linked into the product library
if [ ! -f ../libathena_mp_testcomm.so ] || [ test_transport.cpp -nt ../libathena_mp_testcomm.so ] || [ transport.h -nt ../libathena_mp_testcomm.so ]; then
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O2 -std=c++17 -fPIC -shared test_transport.cpp -o ../libathena_mp_testcomm.so
fi
echo "built $(readlink -f $OUT)"
