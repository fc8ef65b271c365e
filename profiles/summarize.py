#!/usr/bin/env python3
"""profiles/summarize.py <tag> [round]: condense gpurun_out/prof_<tag>/ (written by run_profile.sh)
into tracked files under profiles/:
  <round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, kernel names shortened
  <round>_<tag>_pmc.json           per-kernel average FETCH_SIZE / WRITE_SIZE per launch and the HBM
                                   bytes derived as MI355X_MICROARCH.md §HBM prescribes:
                                   FETCH_SIZE, WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
                                   half of the bytes of a wide (16 B/lane) coalesced streaming read,
                                   so read bytes = 2 * FETCH_SIZE * 1024 for the streaming kernels here.
  traffic.json                     what bench.py reports as roofline.traffic (scatter kernel, per launch)
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")


def short(name):
    m = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_[A-Za-z]+)", name)
    if not m:
        return name[:40]
    k = m.group(1)
    t = re.search(re.escape(k) + r"<([^>]*)>", name)
    if k == "k_scatter_wcn" and t:                       # input format: 16-byte tuples or narrow arrays
        k += "<in_narrow=%s>" % t.group(1).strip()
    if k == "k_join_ct" and t:
        a = [x.strip() for x in t.group(1).split(",")]
        k += "<%s threads%s>" % (a[0], ", narrow" if a[-1] == "true" else "")
    return k


# FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section): x2 for the streaming reads of these kernels.  It holds for
# the narrow-format kernels (8 B + 4 B per lane) as well: k_scatter_wcn<in_narrow=true> must read its whole 12.0 GB
# input (far beyond every cache) and the raw counter says 6.01 GB.


stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(ROOT, "profiles", f"{rnd}_{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

pmc = {}
for counter, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    files = glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    per = {}
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        per.setdefault(k, []).append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in per.items():
        # skip warm-up-sized outliers: average over all launches of the kernel
        pmc.setdefault(k, {})[counter + "_KiB_avg"] = sum(x[0] for x in v) / len(v)
        pmc[k][counter + "_launches"] = len(v)
        pmc[k]["avg_ns_under_" + counter] = sum(x[1] for x in v) / len(v)
for k, v in pmc.items():
    f_, w_ = v.get("FETCH_SIZE_KiB_avg"), v.get("WRITE_SIZE_KiB_avg")
    if f_ is not None and w_ is not None:
        corr = 2
        v["fetch_size_correction"] = corr
        v["hbm_read_bytes_per_launch"] = corr * f_ * 1024   # gfx950 correction for 16 B/lane streams
        v["hbm_write_bytes_per_launch"] = w_ * 1024
        v["hbm_bytes_per_launch"] = v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]
bench = {}
for line in open(os.path.join(src, "bench_trace.log")):
    if line.startswith("{"):
        bench = json.loads(line)
out = {"tag": tag, "bench_line_under_kernel_trace": bench, "kernels": pmc}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_{tag}_pmc.json"), "w"), indent=1)
sc = [v for k, v in pmc.items() if k.startswith("k_scatter_wc") and "hbm_bytes_per_launch" in v]
if sc:
    nl = sum(v["FETCH_SIZE_launches"] for v in sc)
    per_launch = sum(v["hbm_bytes_per_launch"] * v["FETCH_SIZE_launches"] for v in sc) / nl
    m = re.search(r"^(\d+) x", bench.get("config", {}).get("workload", ""))
    b = re.search(r"\((\d+)\+(\d+) bit\)", bench.get("config", {}).get("workload", ""))
    sys.path.insert(0, ROOT)
    from bench import source_blobs                      # the kernel sources these counters were measured on
    def one(flag):
        v = [x for k, x in pmc.items() if k.startswith("k_scatter_wcn<in_narrow=%s" % flag) and "hbm_bytes_per_launch" in x]
        return v[0]["hbm_bytes_per_launch"] if len(v) == 1 else None
    json.dump({"tuples": int(m.group(1)) if m else None, "bits": [int(b.group(1)), int(b.group(2))] if b else None,
               "scatter_hbm_bytes_per_launch": per_launch,
               "scatter_pass1_hbm_bytes_per_launch": one("false"), "scatter_pass2_hbm_bytes_per_launch": one("true"),
               "kernels": sorted(k for k in pmc if k.startswith("k_scatter_wc")),
               "source": f"profiles/{rnd}_{tag}_pmc.json", "source_blobs": source_blobs()},
              open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
for r in rows[:8]:
    print(f'{short(r["Name"]):40s} calls={r["Calls"]:>4s} avg={float(r["AverageNs"])/1e6:9.3f} ms  {r["Percentage"]}%')
for k, v in pmc.items():
    if "hbm_bytes_per_launch" in v:
        print(f'{k:40s} read={v["hbm_read_bytes_per_launch"]/1e9:8.2f} GB write={v["hbm_write_bytes_per_launch"]/1e9:8.2f} GB per launch')
