#!/bin/bash
# A/B of a secondary configuration: the stock library against variants/libathena_mp_<tag>.so
#   bash scripts/gpu_ab.sh c3 r3base [reps]
CFG=${1:-c3}; TAG=${2:-r3base}; REPS=${3:-20}
echo "== stock"; python3 scripts/bench_configs.py --config $CFG --reps $REPS --no-cpu
echo "== $TAG"; ATHENA_MP_LIB=../variants/libathena_mp_$TAG.so python3 scripts/bench_configs.py --config $CFG --reps $REPS --no-cpu
echo "== stock again"; python3 scripts/bench_configs.py --config $CFG --reps $REPS --no-cpu
