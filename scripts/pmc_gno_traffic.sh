#!/bin/bash
# HBM traffic counters of the fused GNO kernels at configs[3], one pass each: gpurun_out/pmc_gno_traffic.txt
OUT=$PWD/gpurun_out; mkdir -p $OUT; : > $OUT/pmc_gno_traffic.txt
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=/tmp/pmc_gt_$(echo $C | tr ' ' '_'); rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $OLDPWD/scripts/bench_configs.py --config c4 --no-cpu --reps 1 > /dev/null 2>> $OUT/pmc_gno.err)
  python3 scripts/pmc_summarise.py $D gno_ >> $OUT/pmc_gno_traffic.txt
done
cat $OUT/pmc_gno_traffic.txt
