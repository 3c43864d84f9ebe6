export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  D=/tmp/pmc_c3g; rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $D -- python3 /root/repo/scripts/bench_configs.py --config c3 --reps 2 --no-cpu > /dev/null 2>&1)
  python3 /root/repo/scripts/pmc_summarise.py $D csr_gather
  python3 /root/repo/scripts/pmc_summarise.py $D readout_bwd
done
