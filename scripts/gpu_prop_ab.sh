#!/bin/bash
# duvenaud_propagate at configs[2] size: the lean buffer-load kernel against round 2's (ATHENA_MP_SHORT_ROWS_V1), alternating
for r in new old new old; do echo "== $r"; if [ $r = old ]; then export ATHENA_MP_SHORT_ROWS_V1=1; else unset ATHENA_MP_SHORT_ROWS_V1; fi; python3 scripts/bench_configs.py --config c3 --reps 20 --no-cpu | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms']['propagate'], d['total_ms'])"; done
