set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r04/gputest_b.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/gputest_b.log
tail -4 gpurun_out/r04/gputest_b.log
timeout -k 10 300 python tools/configs_bench.py --only cfg4_loop8192,cfg4_cutoff_sweep,cfg4_delay_sweep > gpurun_out/r04/configs_cfg4.txt 2>&1; cat gpurun_out/r04/configs_cfg4.txt | tail -4
