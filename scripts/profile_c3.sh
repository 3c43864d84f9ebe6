#!/bin/bash
# configs[2] (Duvenaud) evidence: kernel trace + SQ counters + HBM traffic, one rocprofv3 pass each -> gpurun_out/prof_c3_$TAG/
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_c3_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
python3 scripts/bench_configs.py --config c3 --reps 20 --no-cpu > $OUT/c3_config.json 2> $OUT/err.txt
python3 scripts/gpu_stream.py > $OUT/stream_ceiling.txt 2>> $OUT/err.txt
rm -rf /tmp/kt3 && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python3 $OLDPWD/scripts/bench_configs.py --config c3 --reps 10 --no-cpu > /dev/null 2>> $OUT/err.txt)
find /tmp/kt3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c3_kernel_stats.csv
: > $OUT/c3_pmc.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
  D=/tmp/pmc3_$(echo $C | tr ' ' '_' | cut -c1-40); rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $OLDPWD/scripts/bench_configs.py --config c3 --reps 2 --no-cpu > /dev/null 2>> $OUT/err.txt)
  python3 scripts/pmc_summarise.py $D duv_ >> $OUT/c3_pmc.txt
  python3 scripts/pmc_summarise.py $D csr_gather >> $OUT/c3_pmc.txt
done
cat $OUT/c3_pmc.txt
