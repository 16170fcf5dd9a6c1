#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of tools/profile.sh into profiles/<tag>_*.{csv,md,json}:
    python tools/profile_summary.py <gpurun_out/profile_<tag>> <profiles> <tag>
Machine-readable per-config evidence (one record per BASELINE config and per tools/wave_ops.py graph: kernel, avg_ms, algorithmic bytes,
fraction of 8 TB/s, fraction of the same box's fill kernel) out of the JSON records tools/configs_bench.py --json / tools/wave_ops.py --json wrote:
    python tools/profile_summary.py configs <configs.json> <wave_ops.json> <profiles/rNN_configs.json>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

if sys.argv[1] == "configs":
    out = {"note": "kernel time by HIP events around the launch (median of 3 renders; `first_render_ms_compile_inclusive` is the first of them: for a new "
                   "circuit structure it contains the hiprtc compile unless the code-object cache held it), 1 x MI355X; algorithmic_bytes = 4 B x instances x "
                   "channels x samples written, 0 read; frac_of_fill = against dusp_fill_device on the same box in the same process",
           "configs": json.load(open(sys.argv[2])), "wave_ops": json.load(open(sys.argv[3]))}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print("%d configs, %d graphs -> %s" % (len(out["configs"]), len(out["wave_ops"]), sys.argv[4]))
    sys.exit(0)
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
lines = ["# rocprofv3 summary (%s)" % tag, "",
         "Command: `python3 bench.py --cpu-seconds 0` (default --steps 10 --warmup 3) (1 GPU, 1024 voices x 60 s @ 48 kHz).", ""]
kernel_ms = None
def newest(pattern):
    """gpurun merges every call's files into the same directory: keep the most recent capture of a pass."""
    found = sorted(glob.glob(pattern), key=os.path.getmtime)
    return found[-1:]


for f in newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "%s_kernel_stats.csv" % tag))
    lines += ["## --kernel-trace --stats", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        lines.append("| `%s` | %s | %.4f | %.4f | %.4f | %s |" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                              float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
        if "dusp_fused_kernel" in r["Name"] or "dusp_chunk_kernel" in r["Name"]:
            kernel_ms = float(r["AverageNs"]) / 1e6
    lines.append("")
counters = collections.OrderedDict()
meta = {}
for f in [g for d in sorted(glob.glob(os.path.join(src, "pmc_*"))) for g in newest(os.path.join(d, "*", "*_counter_collection.csv"))]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "dusp_fused_kernel" in r["Kernel_Name"] or "dusp_chunk_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
    for k, v in agg.items():
        counters[k] = sum(v) / len(v)
lines += ["## --pmc (separate passes; mean per launch of the render kernel)", "", "| counter | per launch |", "|---|---|"]
for k, v in counters.items():
    lines.append("| %s | %.6g |" % (k, v))
lines.append("")
algo = 4.0 * 1024 * 2880000
traffic = {}
if "WRITE_SIZE" in counters:
    wb = counters["WRITE_SIZE"] * 1024.0  # WRITE_SIZE is in KiB; exact for 16-B/lane streaming stores on gfx950
    rb = counters.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0  # FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950
    lines += ["## HBM traffic", "",
              "- WRITE_SIZE x 1024 B = **%.4g B** written per launch; algorithmic bytes (4 B x 1024 voices x 2 880 000 samples) = %.4g B -> ratio %.4f"
              % (wb, algo, wb / algo),
              "- FETCH_SIZE x 1024 B x 2 (gfx950 correction) = %.4g B read per launch (wave table + parameters)" % rb, ""]
    traffic = {"n_voices": 1024, "n_samples": 2880000, "engine": "fused", "write_bytes_per_launch": wb,
               "read_bytes_per_launch": rb, "algorithmic_bytes_per_launch": algo, "kernel_ms_rocprof": kernel_ms,
               "source": "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes, MI355X_MICROARCH.md HBM corrections"}
    json.dump(traffic, open(os.path.join(dst, "traffic_%s.json" % tag), "w"), indent=1)
if kernel_ms and "GRBM_GUI_ACTIVE" in counters:
    lines += ["- effective shader clock = GRBM_GUI_ACTIVE / 8 / kernel time = %.3f GHz (profiled pass)" % (counters["GRBM_GUI_ACTIVE"] / 8 / (kernel_ms * 1e-3) / 1e9), ""]
if "SQ_INSTS_VALU" in counters:
    groups = 1024 * 2880000 / 64.0
    lines += ["- VALU wave-instructions per rendered sample (per lane): %.2f; LDS instructions per sample: %.2f"
              % (counters["SQ_INSTS_VALU"] / groups, counters.get("SQ_INSTS_LDS", 0) / groups)]
if "SQ_LDS_BANK_CONFLICT" in counters and counters.get("SQ_LDS_IDX_ACTIVE"):
    lines += ["- LDS bank-conflict cycles / LDS active cycles: %.1f %%" % (100 * counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"])]
lines += ["", "Dispatch: %s" % json.dumps(meta), ""]
for f in newest(os.path.join(src, "configs", "*", "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "%s_configs_kernel_stats.csv" % tag))
    lines += ["## other BASELINE configs: `python3 tools/configs_bench.py --rounds 2` under --kernel-trace --stats", "",
              "(kernel durations from the trace: launches only — the lines of `configs.log` below are HIP-event medians over the rounds, so a round-1 "
              "render of a NEW circuit structure, which contains its hiprtc compile, does not show in them; `first_render_ms_compile_inclusive` in "
              "`%s_configs.json` keeps that figure apart)" % tag, "",
              "| kernel | calls | avg ms | min ms | max ms |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        if "dusp_" in r["Name"]:
            lines.append("| `%s` | %s | %.4f | %.4f | %.4f |" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                              float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
    lines.append("")
    log = os.path.join(src, "configs.log")
    if os.path.exists(log):
        lines += ["```"] + [l.rstrip() for l in open(log) if l.startswith("cfg")] + ["```", ""]
slog = os.path.join(src, "stats.log")
if os.path.exists(slog):
    for l in open(slog):
        if l.startswith("{") and '"metric"' in l:
            open(os.path.join(dst, "%s_bench_line_profiled.json" % tag), "w").write(l)
            j = json.loads(l)
            lines += ["## bench.py's own line from the --kernel-trace pass (HIP events inside bench.py)", "",
                      "- kernel_ms (HIP events, timed steps only) = %.4f; rocprofv3 AverageNs over all launches incl. warm-up = %.4f ms"
                      % (j["roofline"]["kernel_ms"], kernel_ms or float("nan")),
                      "- roofline.frac in that pass = %.4f (clock state varies between processes on the same box; see %s_bench_line.json for the unprofiled run of the same call)"
                      % (j["roofline"]["frac"], tag), ""]
open(os.path.join(dst, "%s_summary.md" % tag), "w").write("\n".join(lines))
print("\n".join(lines))
