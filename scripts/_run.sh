timeout 1200 python -m pytest tests -m gpu -x -q -k "duvenaud or c3 or fuzz or network or layer" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 600 python3 scripts/bench_configs.py --config c3 --no-cpu
ATHENA_MP_NO_SHORT_ROWS=1 timeout 600 python3 scripts/bench_configs.py --config c3 --no-cpu
