set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04/gputest_e.log 2>&1
echo "pytest rc $?" >> gpurun_out/r04/gputest_e.log
grep -v "^\.\|^$" gpurun_out/r04/gputest_e.log | tail -22 | cut -c1-300
