mkdir -p gpurun_out/r2g
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -5
timeout 300 python3 bench.py > gpurun_out/r2g/bench1.json 2> gpurun_out/r2g/bench1.err; python3 -c "
import json; d=json.load(open('gpurun_out/r2g/bench1.json')); print(d['ms_per_step'], d['value']/1e9, d['roofline']['frac'], d['parity']['ok'], d['cpu_baseline']['value'])"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
