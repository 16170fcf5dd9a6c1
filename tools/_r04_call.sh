mkdir -p gpurun_out/r04
export DUSP_JIT_CACHE=/tmp/empty_cache_$$
DUSP_JIT_LOG=1 timeout -k 10 400 python tools/wave_ops.py "--only=filter(osc);filter(osc) * ramp;filter(osc, lfo);filter(filter(osc));delay(osc, lfo);patch_multitap x 256" --json=gpurun_out/r04/first_call.json 2>&1 | grep -v amdgpu.ids | grep " ms \|dusp jit" | cut -c1-170
python -c "
import json
for r in json.load(open('gpurun_out/r04/first_call.json')): print(r['graph'], r['first_render_ms_compile_inclusive'], r['avg_ms'], r['shape'])"
