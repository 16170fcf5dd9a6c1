#!/usr/bin/env python3
"""Mean per launch of every counter in rocprofv3 --pmc CSVs, for the kernels whose name contains a pattern.
  python tools/pmc_table.py <dir with pmc_*/.../*_counter_collection.csv> <kernel name pattern> [samples per launch [traffic.json]]
With WRITE_SIZE / FETCH_SIZE among the counters the HBM traffic per launch is printed against the algorithmic bytes (4 B per sample written, 0 read) with the
gfx950 corrections of MI355X_MICROARCH.md (WRITE_SIZE in KiB, exact for 16-byte-per-lane stores; FETCH_SIZE in KiB, HALF the bytes of wide coalesced reads), and
written as a record bench.py's roofline.traffic picks up (profiles/traffic_*.json)."""
import json
import collections
import csv
import glob
import os
import sys

src, pat = sys.argv[1], sys.argv[2]
samples = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
counters, meta = collections.OrderedDict(), {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
    for k, v in agg.items():
        counters[k] = sum(v) / len(v)
print("| counter | per launch |" + (" per sample |" if samples else ""))
print("|---|---|" + ("---|" if samples else ""))
for k, v in counters.items():
    print("| %s | %.6g |" % (k, v) + (" %.4g |" % (v * (64.0 if k.startswith("SQ_INSTS") else 1.0) / samples) if samples else ""))
print()
print("dispatch: %s" % meta)
if samples and "WRITE_SIZE" in counters:
    wb = counters["WRITE_SIZE"] * 1024.0
    rb = counters.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
    algo = 4.0 * samples
    print("HBM traffic per launch: WRITE_SIZE x 1024 B = %.4g B written (%.3f x the algorithmic %.4g B); FETCH_SIZE x 1024 B x 2 = %.4g B read; together %.4g B = %.2f x algorithmic"
          % (wb, wb / algo, algo, rb, wb + rb, (wb + rb) / algo))
    if len(sys.argv) > 4:
        json.dump({"n_voices": 8192, "n_samples": int(samples / 8192), "engine": "wave", "write_bytes_per_launch": wb, "read_bytes_per_launch": rb, "hbm_bytes_per_launch": wb + rb,
                   "algorithmic_bytes_per_launch": algo, "source": "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE on `bench.py --config cfg4`, separate passes, MI355X_MICROARCH.md HBM corrections "
                   "(FETCH_SIZE doubled); earlier box"}, open(sys.argv[4], "w"), indent=1)
if "SQ_LDS_BANK_CONFLICT" in counters and counters.get("SQ_LDS_IDX_ACTIVE"):
    print("LDS bank-conflict cycles / LDS active cycles: %.1f %%" % (100.0 * counters["SQ_LDS_BANK_CONFLICT"] / counters["SQ_LDS_IDX_ACTIVE"]))
if "SQ_WAIT_INST_ANY" in counters and counters.get("SQ_WAVE_CYCLES"):
    print("SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: %.1f %%   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: %.1f %%" %
          (100.0 * counters["SQ_WAIT_INST_ANY"] / counters["SQ_WAVE_CYCLES"], 100.0 * counters.get("SQ_ACTIVE_INST_VALU", 0) / counters["SQ_WAVE_CYCLES"]))
