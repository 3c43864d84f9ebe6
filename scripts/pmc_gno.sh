#!/bin/bash
# PMC passes over one leg of C4 (scripts/bench_configs.py --only <leg>): gpurun_out/pmc_gno_<leg>.txt
leg=${1:-fwd}; kern=${2:-gno_}
OUT=$PWD/gpurun_out; mkdir -p $OUT; : > $OUT/pmc_gno_$leg.txt
export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  D=/tmp/pmc_gno_$(echo $C | tr ' ' '_' | cut -c1-40); rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $OLDPWD/scripts/bench_configs.py --config c4 --no-cpu --reps 1 --only $leg > /dev/null 2>> $OUT/pmc_gno.err)
  python3 scripts/pmc_summarise.py $D $kern >> $OUT/pmc_gno_$leg.txt
done
cat $OUT/pmc_gno_$leg.txt
