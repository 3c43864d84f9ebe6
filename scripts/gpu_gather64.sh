#!/bin/bash
# what FETCH_SIZE tallies for 64-byte gathers (scripts/ubench/gather64.hip): timings, then one counter pass each
cd "$(dirname "$0")/.."; OUT=$PWD/gpurun_out/gather64; mkdir -p $OUT; export TMPDIR=/tmp
./scripts/ubench/gather64 > $OUT/times.txt 2>&1; cat $OUT/times.txt
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/counters.txt; cat $OUT/counters.txt; echo
: > $OUT/pmc.txt
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  D=/tmp/pmc_g64; rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -- $OLDPWD/scripts/ubench/gather64 > /dev/null 2>> $OUT/err.txt)
  python3 scripts/pmc_summarise.py $D k_ >> $OUT/pmc.txt
done
cat $OUT/pmc.txt
