mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -q -s -k "random_filter_circuits_in_warming or cut_into_segments or segments_that_warm_up or time_split_wave_render_equals_unsplit" 2>&1 | grep -v "^$" | grep "^E \|one chain\|passed\|failed" | cut -c1-300
