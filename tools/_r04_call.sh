mkdir -p gpurun_out/r04
for S in 64 128; do
  echo "== DUSP_JIT_SPILL=$S"
  DUSP_JIT_SPILL=$S timeout -k 10 400 python tools/wave_ops.py 2>&1 | grep -v amdgpu.ids | grep " ms " | cut -c1-150
done
