set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r04/gputest_d.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/gputest_d.log
grep -v "^\.\|^$" gpurun_out/r04/gputest_d.log | tail -30 | cut -c1-300
