#!/bin/bash
# A/B of the cache bits on the per-entry partials' stores of gno_dh_pc_kernel<PX> (configs[3]): stock (0) against
# variants built with scripts/build_variants.sh gno.hip pxnt -DGNO_PX_AUX=2 (nt) and pxntsc -DGNO_PX_AUX=19 (sc0 sc1 nt)
mkdir -p gpurun_out/px_ab
for i in 1 2; do
  for tag in stock pxnt pxntsc; do
    if [ $tag = stock ]; then unset ATHENA_MP_LIB; else export ATHENA_MP_LIB=$PWD/variants/libathena_mp_$tag.so; fi
    python3 scripts/bench_secondary.py --config c4 > gpurun_out/px_ab/${tag}_$i.json 2> gpurun_out/px_ab/${tag}_$i.err
  done
done
python3 - <<PY | tee gpurun_out/px_ab/summary.txt
import json, glob
print("run        step_ms  reverse_pass_ms  parity")
for f in sorted(glob.glob("gpurun_out/px_ab/*.json")):
    for l in open(f):
        try: j = json.loads(l)
        except Exception: continue
        print(f.split("/")[-1][:-5].ljust(10), j["step_ms"], j["ops"]["bwd_x+theta_one_contraction"]["ms"], j["parity"]["ok"])
PY
