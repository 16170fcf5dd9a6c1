#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python bench.py --config cfg4 --steps 5 --warmup 2 > gpurun_out/cfg4_a.json 2> gpurun_out/cfg4_a.err && cat gpurun_out/cfg4_a.json && \
python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err && cat gpurun_out/bench_a.json && \
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/gputest_a.txt 2>&1; tail -5 gpurun_out/gputest_a.txt
