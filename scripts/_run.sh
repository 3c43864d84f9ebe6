export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_c3; mkdir -p $OUT
HERE=$PWD
cd /tmp; rm -rf /tmp/p1 /tmp/p2
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d /tmp/p1 -- python3 $HERE/scripts/bench_configs.py --config c3 --reps 2 --no-cpu > /dev/null 2> $OUT/e1.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $HERE/scripts/bench_configs.py --config c3 --reps 5 --no-cpu > /dev/null 2>> $OUT/e1.txt
cd $HERE
python3 scripts/pmc_summarise.py /tmp/p1 > $OUT/c3_pmc.txt
find /tmp/p2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c3_kernel_stats.csv
grep -E "duv_|readout|csr_gather" $OUT/c3_pmc.txt | cut -c1-140
head -12 $OUT/c3_kernel_stats.csv | cut -c1-150
