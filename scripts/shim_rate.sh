#!/bin/bash
# The drop-in path end to end (examples/shim_sweep_rate.cpp: BotParam keys -> InsHandler / LegOdoHandler -> FrontEnd ->
# MavStateEstimator::addUpdate -> kernels), 64k filters fed by ONE robot's IMU + foot-state log: wall-clock rates for the
# configurations DESIGN.md 6 quotes, then the same 15-state run under rocprofv3 --kernel-trace --stats.
#   bash scripts/shim_rate.sh <out dir>      (on the GPU box; the example is built by tests/test_cpp_shim.py's recipe)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/shim_rate}
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, 'tests'); import test_cpp_shim as t; print(t.build_shim_sweep_rate())" > $OUT/build.log 2>&1 || exit 11
EXE=tests/build/shim_sweep_rate
{
  $EXE 65536 2000 15 0
  $EXE 65536 2000 15 0 1000000 0 feet two
  $EXE 65536 2000 15 0 1000000 0 joints
  $EXE 65536 2000 15 0 1000000 0 joints two
  $EXE 65536 2000 15 32
  $EXE 65536 2000 15 32 1000000 0 joints
  $EXE 65536 2000 15 0 1000000 32
  # the history as every reference .cfg gets it (only utime_history_span set: 32 slots spaced over the span)
  $EXE 65536 2000 15 derived 1000000 0 joints
  $EXE 65536 2000 21 derived 1000000 0 joints
  $EXE 65536 2000 21 0
  $EXE 65536 2000 21 0 1000000 0 joints
  $EXE 65536 2000 21 32
  # LegOdoCommon's six-row modes: formed and applied inside the pair kernel (round 4; rbis_legstep.hpp, SIX)
  $EXE 65536 2000 15 0 1000000 0 joints one lin_rot_rate
  $EXE 65536 2000 15 0 1000000 0 joints one pos_and_lin_rate
  $EXE 65536 2000 21 0 1000000 0 joints one lin_rot_rate
  $EXE 65536 2000 21 0 1000000 0 joints one pos_and_lin_rate
} > $OUT/shim_sweep.txt 2>&1 || exit 12
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $EXE 65536 2000 15 0 1000000 0 joints > $OUT/trace.txt 2> $OUT/trace.err || exit 13
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace21 -- $EXE 65536 2000 21 0 1000000 0 joints > $OUT/trace21.txt 2> $OUT/trace21.err || exit 14
echo "shim rate done"
