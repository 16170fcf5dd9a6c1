import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DUSP_WAVE_JIT", "2")
import numpy as np, torch
import dusp_amd as d
from dusp_amd import descriptor, runtime
from oracle import oracle
d.configure(48000)
def loop(k):
    s = d.Sum(d.Osc(110 + k / 64), 0)
    f = d.Filter(d.Delay(s, d.Sum(d.Multiply(d.Osc(2), 40), 300), 4096), 2000)
    s.B = d.Multiply(f, 0.5)
    return f
uni = descriptor.unify([descriptor.extract(loop(k)) for k in (0, 64)])
V, n = 8192, 48000
params = (110 + np.arange(V) / 64.0).astype(np.float32).reshape(1, V)
ctx = runtime.Context(0, 48000)
dp = torch.from_numpy(params).cuda()
out = torch.empty((V, 1, n), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
res = {}
for name, eng in (("auto", runtime.ENGINE_AUTO), ("wave", runtime.ENGINE_WAVE), ("chunk", runtime.ENGINE_CHUNK)):
    try:
        prog = ctx.build(uni.words, eng)
    except runtime.DuspHipError as e:
        print(name, "refused:", e.message); continue
    ts = []
    for r in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); prog.render_device(n, V, dp.data_ptr(), out.data_ptr(), stream); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    res[name] = out[[0, 1, 4095, 8191]].cpu().numpy().copy()
    print("%-6s %-40s %8.2f ms for 8192 x 1 s" % (name, prog.engine + ": " + prog.read_shape(), float(np.median(ts))))
    prog.close()
for i, inst in enumerate((0, 1, 4095, 8191)):
    want = oracle.render(uni.words, n, params=params, n_instances=V, instance=inst)
    for name in res:
        err = float(np.max(np.abs(res[name][i].astype(np.float64) - want))) / float(np.max(np.abs(want)))
        print("   instance %5d %-6s max err vs oracle %.3g of scale, equal to chunk: %s" % (inst, name, err, np.array_equal(res[name][i], res["chunk"][i]) if "chunk" in res else "-"))
