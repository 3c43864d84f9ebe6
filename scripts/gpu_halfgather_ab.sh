for i in 1 2; do
python3 scripts/bench_configs.py --config c4 --no-cpu --reps 3 --only fwd 2>&1 | grep '^{'
ATHENA_MP_LIB=$PWD/variants/libathena_mp_halfgather.so python3 scripts/bench_configs.py --config c4 --no-cpu --reps 3 --only fwd 2>&1 | grep '^{'
python3 scripts/bench_configs.py --config c4 --no-cpu --reps 3 --only fwd_save 2>&1 | grep '^{'
ATHENA_MP_LIB=$PWD/variants/libathena_mp_halfgather.so python3 scripts/bench_configs.py --config c4 --no-cpu --reps 3 --only fwd_save 2>&1 | grep '^{'
done
