#!/usr/bin/env python3
"""profiles/summarize_sq.py <tag> [round]: condense gpurun_out/sq_<tag>/p*/ (profiles/run_sq.sh) into
profiles/<round>_sq_counters_<tag>.json -- per kernel the average counter value per launch and its ratio to SQ_WAVE_CYCLES."""
import csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
acc = {}
for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"sq_{tag}", "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"(k_[a-z_0-9]+)(<[^(]*>)?\(", name) or re.search(r"(k_[a-z_0-9]+)", name)
        if not m:
            continue
        k = m.group(1) + (m.group(2) or "")
        d = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
        d[0] += float(r["Counter_Value"]); d[1] += 1
out = {"how": "profiles/run_sq.sh (four rocprofv3 --pmc passes, no tracing) of python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-auto "
              "--no-extras --no-verify; per kernel the AVERAGE counter value per launch, and its ratio to SQ_WAVE_CYCLES (per resident wavefront)",
       "kernels": {}}
for k, cs in sorted(acc.items()):
    per = {c: v[0] / v[1] for c, v in cs.items()}
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    out["kernels"][k] = {"launches": max(v[1] for v in cs.values()), "per_launch": per,
                         "per_wave_cycle": {c: round(x / wc, 4) for c, x in per.items()} if wc else {}}
dst = os.path.join(ROOT, "profiles", f"{rnd}_sq_counters_{tag}.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k, v in out["kernels"].items():
    if "join_ct" in k or "scatter_wcn" in k:
        p = v["per_wave_cycle"]
        print(k[:70], {c: p.get(c) for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
              "VALU insts", v["per_launch"].get("SQ_INSTS_VALU"), "LDS insts", v["per_launch"].get("SQ_INSTS_LDS"),
              "bank conflict / idx active", round(v["per_launch"].get("SQ_LDS_BANK_CONFLICT", 0) / max(v["per_launch"].get("SQ_LDS_IDX_ACTIVE", 1), 1), 3))
