#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile_bench.sh) into profiles/:
  profiles/<tag>_kernel_stats.csv   top rows of rocprofv3 --kernel-trace --stats
  profiles/<tag>_bench.log          the bench JSON line of the same command
  profiles/<tag>_hbm_traffic.json   per-launch FETCH_SIZE / WRITE_SIZE averages (separate --pmc passes)
  profiles/hbm_traffic.json         what bench.py reports as roofline.traffic
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled for wide reads on gfx950
(MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
write_default = "--not-default" not in sys.argv   # profiles/hbm_traffic.json is what bench.py reads for the DEFAULT workload
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)

stats_files = glob.glob(os.path.join(base, "trace", "*", "*_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats_files[0])))
with open(os.path.join(out_dir, tag + "_kernel_stats.csv"), "w", newline="") as handle:
    writer = csv.writer(handle)
    writer.writerow(rows[0].keys())
    for row in rows[:10]:
        writer.writerow(row.values())

for line in open(os.path.join(base, "bench_trace.log")):
    if line.startswith("{"):
        open(os.path.join(out_dir, tag + "_bench.log"), "w").write(line)

traffic = {}
for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(base, "pmc_" + name, "*", "*_counter_collection.csv"))
    sums = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] != counter:
            continue
        kernel = row["Kernel_Name"].split("(")[0]
        sums[kernel][0] += 1
        sums[kernel][1] += float(row["Counter_Value"])
    for kernel, (count, total) in sums.items():
        counting_variant = ("k_trace_small<true>" in kernel) or ("k_trace<" in kernel and kernel.rstrip().endswith("true>"))
        if "pathed::" in kernel and not counting_variant and count > 4:
            traffic.setdefault(kernel, {})[counter + "_KiB_avg"] = total / count
            traffic[kernel]["launches_" + name] = count

summary = {"tag": tag, "kernels": traffic}
trace_kernel = next((k for k in traffic if "k_trace" in k), None)
if trace_kernel:
    fetch = traffic[trace_kernel].get("FETCH_SIZE_KiB_avg", 0.0)
    write = traffic[trace_kernel].get("WRITE_SIZE_KiB_avg", 0.0)
    summary["trace_kernel"] = trace_kernel
    summary["trace_hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
    summary["note"] = "2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, averaged over the launches of the pmc passes"
json.dump(summary, open(os.path.join(out_dir, tag + "_hbm_traffic.json"), "w"), indent=1)
if write_default:
    json.dump(summary, open(os.path.join(out_dir, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
print(open(os.path.join(out_dir, tag + "_kernel_stats.csv")).read()[:1200])
