#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/pmc_cornell.sh / tools/pmc_dragon.sh
(gpurun_out/pmc_<tag>/p*/) into profiles/<name>.json: per kernel, the per-launch average of every
counter collected, plus the kernel's average duration from the same passes' kernel trace.
Usage: tools/summarize_pmc.py <tag> <name>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
base = os.path.join(ROOT, "gpurun_out", "pmc_" + tag)
kernels = collections.defaultdict(dict)
for directory in sorted(glob.glob(os.path.join(base, "p*/"))):
    for path in glob.glob(os.path.join(directory, "*", "*_counter_collection.csv")):
        sums = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(path)):
            kernel = row["Kernel_Name"].split("(")[0]
            if "pathed::" not in kernel:
                continue
            entry = sums[(kernel, row["Counter_Name"])]
            entry[0] += 1
            entry[1] += float(row["Counter_Value"])
        for (kernel, counter), (count, total) in sums.items():
            kernels[kernel][counter] = total / count
            kernels[kernel]["launches_in_pass"] = count
    for path in glob.glob(os.path.join(directory, "*", "*_kernel_trace.csv")):
        durations = collections.defaultdict(list)
        for row in csv.DictReader(open(path)):
            kernel = row["Kernel_Name"].split("(")[0]
            if "pathed::" in kernel:
                durations[kernel].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for kernel, values in durations.items():
            kernels[kernel].setdefault("avg_duration_us_under_pmc", round(sum(values) / len(values) / 1e3, 2))
summary = {
    "tag": tag,
    "note": "per-launch averages; one rocprofv3 --pmc pass per counter group (no --stats / tracing domains mixed in); "
            "SQ_INSTS_VALU counts wave instructions (a wave64 VALU instruction occupies its SIMD for 4 cycles)",
    "kernels": {kernel: dict(sorted(values.items())) for kernel, values in sorted(kernels.items())},
}
out = os.path.join(ROOT, "profiles", name + ".json")
json.dump(summary, open(out, "w"), indent=1)
print(out)
for kernel, values in summary["kernels"].items():
    if "SQ_INSTS_VALU" in values and "SQ_WAVES" in values:
        print("%-50s VALU/wave %.0f  waves %.0f  wait %.2f  dur %.1f us" % (
            kernel[-50:], values["SQ_INSTS_VALU"] / values["SQ_WAVES"], values["SQ_WAVES"],
            values.get("SQ_WAIT_ANY", 0) / max(values.get("SQ_WAVE_CYCLES", 1), 1), values.get("avg_duration_us_under_pmc", 0)))
