mkdir -p gpurun_out/r04
timeout -k 10 300 python tools/_dbg.py > gpurun_out/r04/dbg.txt 2>&1; tail -40 gpurun_out/r04/dbg.txt
