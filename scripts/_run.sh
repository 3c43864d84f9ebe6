bash scripts/profile_round.sh r02b > gpurun_out/prof_r02b.log 2>&1
tail -3 gpurun_out/prof_r02b.log; cat gpurun_out/prof_r02b/pmc_summary.txt | head -4; head -4 gpurun_out/prof_r02b/bench_kernel_stats.csv | cut -c1-60,180-260; tail -4 gpurun_out/prof_r02b/f256.txt; python3 -c "
import json; d=json.load(open('gpurun_out/prof_r02b/bench.json')); print(d['ms_per_step'], d['value']/1e9, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['parity'])
d=json.load(open('gpurun_out/prof_r02b/bench_profiled.json')); print('profiled', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
cat gpurun_out/prof_r02b/fortran_bench.txt | grep driver | cut -c1-200
