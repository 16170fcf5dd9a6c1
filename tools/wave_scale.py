import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import dusp_amd as d
from dusp_amd import descriptor, runtime
sr = 48000
d.configure(sr)
ctx = runtime.Context(0, sr)
stream = torch.cuda.current_stream().cuda_stream
def fm(k):
    return d.Multiply(d.Osc(d.Sum(d.Multiply(d.Osc(3.0 + k / 100), 40), 220 + k / 4)), d.Ramp(sr, 1, 0).trigger())
for V, n in [(64, 48000), (1024, 48000), (4096, 48000), (16384, 48000)]:
    uni = descriptor.unify([descriptor.extract(fm(k)) for k in (0, 1)])
    ks = np.arange(V)
    params = np.stack([(3.0 + ks / 100), (220 + ks / 4)]).astype(np.float32)
    assert uni.n_params == 2
    prog = ctx.build(uni.words)
    out = torch.empty((V, 1, n), dtype=torch.float32, device="cuda")
    dp = torch.from_numpy(params).cuda()
    ts = []
    for r in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), stream); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = float(np.median(ts))
    print("FM voice (6 units) engine=%s: %6d instances x 1 s: %8.3f ms  %10.1f Msamples/s  %7.1f GB/s" % (prog.engine, V, ms, V * n / ms / 1e3, 4.0 * V * n / ms / 1e6), flush=True)
    prog.close()
