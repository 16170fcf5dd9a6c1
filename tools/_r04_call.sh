mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py -m gpu -q -s -k "random_filter_circuits_in_warming or cut_into_segments" 2>&1 | grep -v "^$" | grep "^E \|one chain\|passed\|failed" | cut -c1-300
