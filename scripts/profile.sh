#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun):  bash scripts/profile.sh r04
# Outputs under gpurun_out/prof_<tag>/ ; condense them with scripts/profile_summary.py into profiles/.
#  1. rocprofv3 --kernel-trace --stats of the default bench command (and of the n=21 variant)
#  2. separate --pmc passes (FETCH_SIZE, then WRITE_SIZE; never combined with trace domains) of the bench command,
#     the cache-busting 1M-filter run, the n=21 run, and the calibration copy (known bytes, same access pattern)
#  3. the adjacent kernels, whole configurations and the smoother (trace + SQ/LDS counters); copy ceilings and sweeps
set -o pipefail
TAG=${1:-r05}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
ONLY=${2:-all}   # "smoother": refresh the smoother's trace and counters only (merged into the same prof_<tag>/)
cd $ROOT
smoother_trace() {
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smoother -- python3 scripts/smooth_rate.py > $OUT/smoother.txt 2> $OUT/smoother.err || exit 23
  # the kernels it replaced, same box, for the record: k_smooth_lane for 15 states as well (the default until the second half of round 5),
  # k_smooth_reg (16 / 32 lanes per filter)
  PRONTO_SMOOTH_KERNEL=lane python3 scripts/smooth_rate.py > $OUT/smoother_lane.txt 2>> $OUT/smoother.err || exit 23
  PRONTO_SMOOTH_KERNEL=reg python3 scripts/smooth_rate.py > $OUT/smoother_reg.txt 2>> $OUT/smoother.err || exit 23
}
smoother_pmc() {
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/smooth_pmc_a -- python3 scripts/smooth_rate.py > $OUT/smooth_pmc_a.txt 2>&1 || exit 24
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/smooth_pmc_b -- python3 scripts/smooth_rate.py > $OUT/smooth_pmc_b.txt 2>&1 || exit 25
  # HBM traffic of the smoother step (one counter per pass, as for the hot step)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/smooth_pmc_fetch -- python3 scripts/smooth_rate.py > $OUT/smooth_pmc_fetch.txt 2>&1 || exit 26
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/smooth_pmc_write -- python3 scripts/smooth_rate.py > $OUT/smooth_pmc_write.txt 2>&1 || exit 27
}
if [ "$ONLY" = smoother ]; then
  mkdir -p $OUT; rm -rf $OUT/smoother $OUT/smooth_pmc_a $OUT/smooth_pmc_b $OUT/smooth_pmc_fetch $OUT/smooth_pmc_write
  smoother_trace; smoother_pmc
  echo "smoother profile done"
  exit 0
fi
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace64k -- python3 bench.py --steps 200 --warmup 20 --min-timed-ms 1000 --no-cpu-baseline --no-cache-busting > $OUT/trace64k.json 2> $OUT/trace64k.err || exit 11
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace64k_n21 -- python3 bench.py --steps 100 --warmup 10 --min-timed-ms 1000 --no-cpu-baseline --no-cache-busting --n-states 21 > $OUT/trace64k_n21.json 2> $OUT/trace64k_n21.err || exit 11
# the true-HBM leg: the same step on 1 M filters (state 1.17 GB), kernel trace of its own -- step by step over the WHOLE batch
# (PRONTO_BATCH_BLOCKED=0: the library's default launch order for such a state keeps blocks cache-resident, which is not what this leg is for)
PRONTO_BATCH_BLOCKED=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1m -- python3 bench.py --batch-per-gpu 1048576 --steps 20 --warmup 5 --min-timed-ms 300 --no-cpu-baseline --no-cache-busting > $OUT/trace1m.json 2> $OUT/trace1m.err || exit 11
echo "traces done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc64k_$C -- python3 bench.py --steps 20 --warmup 5 --min-timed-ms 0 --no-cpu-baseline --no-cache-busting > $OUT/pmc64k_$C.json 2> $OUT/pmc64k_$C.err || exit 12
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc64k_n21_$C -- python3 bench.py --steps 20 --warmup 5 --min-timed-ms 0 --no-cpu-baseline --no-cache-busting --n-states 21 > $OUT/pmc64k_n21_$C.json 2> $OUT/pmc64k_n21_$C.err || exit 12
  PRONTO_BATCH_BLOCKED=0 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc1m_$C -- python3 bench.py --batch-per-gpu 1048576 --steps 10 --warmup 2 --min-timed-ms 0 --no-cpu-baseline > $OUT/pmc1m_$C.json 2> $OUT/pmc1m_$C.err || exit 13
  rocprofv3 --pmc $C --output-format csv -d $OUT/calib_$C -- python3 scripts/calib_copy.py > $OUT/calib_$C.txt 2> $OUT/calib_$C.err || exit 14
  echo "pmc $C done"
done
PRONTO_BATCH_BLOCKED=0 python3 bench.py --batch-per-gpu 1048576 --steps 20 --warmup 5 --min-timed-ms 0 --no-cpu-baseline --no-cache-busting > $OUT/bench1m.json 2> $OUT/bench1m.err
python3 scripts/calib_copy.py > $OUT/calib_plain.txt 2>&1
echo "hot path done"
# the kernels next to the hot step, the whole configurations, the smoother (kernel trace, then two SQ / LDS counter passes)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/others -- python3 scripts/kernel_rates.py > $OUT/others.txt 2> $OUT/others.err || exit 21
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs -- python3 scripts/config_rates.py > $OUT/configs.txt 2> $OUT/configs.err || exit 22
smoother_trace
# leg kinematic odometry: per-filter inputs (kernel rates), then the handler API end to end (one log for every filter)
python3 scripts/leg_rates.py > $OUT/leg_rates.txt 2> $OUT/leg_rates.err
bash scripts/shim_rate.sh $OUT/shim > $OUT/shim.log 2>&1
python3 scripts/smooth_log_rate.py > $OUT/smooth_log.txt 2>&1
bash scripts/segment_rate.sh $OUT/segment_rate.txt > $OUT/segment_rate.log 2>&1
echo "adjacent kernels done"
smoother_pmc
echo "smoother counters done"
# ceilings and sweeps, not under the profiler
hipcc -O3 --offload-arch=gfx950 -o /tmp/copybench scripts/copybench.hip && /tmp/copybench > $OUT/copybench.txt 2>&1
# step by step over the whole batch (the rate a single launch gets at each size), then the launch ORDER of bulk replays beyond the
# memory-side cache: the library's default (blocked) against step by step, with 64- and 256-step streams
SWEEP_VARIANTS=unblocked python3 scripts/batch_sweep.py 15 21 > $OUT/batch_sweep.txt 2>&1
for K in 64 256; do
  SWEEP_STEPS=$K SWEEP_SIZES=524288,1048576 SWEEP_VARIANTS=unblocked,default python3 scripts/batch_sweep.py 15 >> $OUT/batch_sweep.txt 2>&1
  SWEEP_STEPS=$K SWEEP_SIZES=262144,524288 SWEEP_VARIANTS=unblocked,default python3 scripts/batch_sweep.py 21 >> $OUT/batch_sweep.txt 2>&1
done
bash scripts/input_footprint.sh > $OUT/n21_input_footprint.txt 2>&1
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
# round 4: forward passes that keep every posterior (per-message into slots vs the write-through replay), the smoother with
# Eigen's pivoting for comparison, the pair kernels A/B (min of 3)  (independent log segments: scripts/segment_rate.sh, above)
python3 scripts/checkpoint_rate.py 2>/dev/null | grep "n=" > $OUT/checkpoint_rate.txt
PRONTO_SMOOTH_PIVOT=1 python3 scripts/smooth_rate.py 2>/dev/null | grep smoother | sed "s/^/PRONTO_SMOOTH_PIVOT=1 /" > $OUT/smoother_pivoted.txt
bash scripts/leg_ab.sh 3 - PRONTO_BATCH_LEG21_TWO=1 > $OUT/leg_ab.txt 2>&1
echo "profile done"
