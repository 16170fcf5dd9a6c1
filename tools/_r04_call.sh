set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_js_host.py tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q -s -k "js_render or one_long_filter or segments_that_warm_up or cutoff_sweeps or filter_circuits_in_full or time_split" > gpurun_out/r04/gputest_c.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/gputest_c.log
grep -v "^\.\|^$" gpurun_out/r04/gputest_c.log | tail -30
