mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -q -s -k "cut_into_segments_that_warm_up or segments_that_warm_up or a_second_process" 2>&1 | grep -v "^$" | grep "^E \|one chain\|cut into\|passed\|failed" | cut -c1-300
