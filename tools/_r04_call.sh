timeout -k 10 600 python tools/_exp_loop.py 2>&1 | grep -v amdgpu.ids | tail -20
