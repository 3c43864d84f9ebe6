mkdir -p gpurun_out/r2f
free -g | head -2
( timeout 1500 python3 bench.py --config c5 --steps 10 --warmup 2 --cpu-sample-rows 400000 > gpurun_out/r2f/bench_c5_n1.json 2> gpurun_out/r2f/bench_c5_n1.err ; echo rc=$? )
cut -c1-1500 gpurun_out/r2f/bench_c5_n1.json; tail -3 gpurun_out/r2f/bench_c5_n1.err
