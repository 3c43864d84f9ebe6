#!/bin/bash
# configs[3] dtheta with the two workgroup orders of gno_stg_kernel (same build, same box)
for o in grouped spread grouped spread; do echo "== $o"; ATHENA_MP_GNO_STG_ORDER=$o python3 scripts/bench_configs.py --config c4 --reps 3 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms'])"; done
