mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_smoother.py -m gpu -x -q < /dev/null > gpurun_out/r04/smooth_lane_tests.txt 2>&1; tail -3 gpurun_out/r04/smooth_lane_tests.txt
{
timeout -k 10 200 python3 scripts/smooth_rate.py < /dev/null
} > gpurun_out/r04/smooth_lane_rate.txt 2>&1
grep smoother gpurun_out/r04/smooth_lane_rate.txt
