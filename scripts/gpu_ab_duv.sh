#!/bin/bash
# the Duvenaud update family: stock library against variants/libathena_mp_<tag>.so, alternating on one box
TAG=${1:-depth3}
for k in 1 2; do python3 scripts/gpu_duv_variants.py; ATHENA_MP_LIB=../variants/libathena_mp_$TAG.so python3 scripts/gpu_duv_variants.py; done
