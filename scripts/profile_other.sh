#!/bin/bash
# rocprofv3 --kernel-trace --stats of the kernels next to the hot step (scripts/kernel_rates.py, scripts/smooth_rate.py,
# scripts/config_rates.py); summaries are copied to profiles/<tag>_kernel_stats_{others,smoother,configs}.csv by hand:
#   bash scripts/profile_other.sh r01
set -o pipefail
TAG=${1:-r01}
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_other_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/others -- python3 scripts/kernel_rates.py > $OUT/others.txt 2> $OUT/others.err || exit 11
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smoother -- python3 scripts/smooth_rate.py > $OUT/smoother.txt 2> $OUT/smoother.err || exit 12
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs -- python3 scripts/config_rates.py > $OUT/configs.txt 2> $OUT/configs.err || exit 13
echo "done"
