#!/usr/bin/env python3
"""Condenses the passes of tools/pmc_per_sample.sh (gpurun_out/pmcps_<tag>/) into profiles/pmc_per_sample.json:
per workload and per camera sample, the VALU wave-instructions issued and the HBM bytes moved (2 x FETCH_SIZE +
WRITE_SIZE, KiB -> bytes: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md
§HBM), split into the trace kernels, the fused path kernel ("path") and everything else of the pipeline ("shade": shade, init and resolve kernels).
Also keeps the per-kernel per-launch averages the totals come from.  Usage: tools/summarize_pmc_per_sample.py <tag>"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
base = os.path.join(ROOT, "gpurun_out", "pmcps_" + tag)


def digest():
    sha = hashlib.sha1()
    directory = os.path.join(ROOT, "pathed_amd", "csrc")
    for name in sorted(os.listdir(directory)):
        sha.update(open(os.path.join(directory, name), "rb").read())
    return sha.hexdigest()[:16]


def totals(name):
    """{counter: {kernel: [launches, total]}} over the pathed:: kernels of one pass"""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    paths = glob.glob(os.path.join(base, name, "*", "*_counter_collection.csv"))
    if len(paths) > 1:
        # gpurun MERGES a call's files into gpurun_out/: an earlier collection's file (another pid) may still lie here
        raise SystemExit("%s: %d counter files in one pass directory -- remove gpurun_out/pmcps_* before collecting again" % (name, len(paths)))
    for path in paths:
        for row in csv.DictReader(open(path)):
            kernel = row["Kernel_Name"].split("(")[0]
            if "pathed::" not in kernel:
                continue
            entry = out[row["Counter_Name"]][kernel]
            entry[0] += 1
            entry[1] += float(row["Counter_Value"])
    return out


def kernel_class(kernel):
    """trace / path (fused) / shade (shade kernels + k_init + k_resolve: the render loop's own traffic) / setup (scene_create's
    kernels -- shading records, device BVH builders -- which run once per scene and are not part of a camera sample)"""
    if "k_trace" in kernel:
        return "trace"
    if "k_path" in kernel:
        return "path"
    if "k_build_tri_shade" in kernel or "k_lbvh" in kernel or "k_ploc" in kernel:
        return "setup"
    return "shade"


def workload(prefix, samples, with_valu):
    entry = {"samples": samples, "kernel_sources": digest()}
    per_kernel = collections.defaultdict(dict)
    if with_valu:
        valu = totals(prefix + "_valu")
        entry["valu_wave_instructions_per_sample"] = sum(v[1] for v in valu["SQ_INSTS_VALU"].values()) / samples
        lanes = sum(v[1] for v in valu["SQ_THREAD_CYCLES_VALU"].values())
        active = sum(v[1] for v in valu["SQ_ACTIVE_INST_VALU"].values())
        if active > 0:
            # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = lanes active per VALU issue cycle group: lane utilisation of 64
            entry["valu_lane_utilisation"] = lanes / active / 64.0
        for counter, kernels in valu.items():
            for kernel, (launches, total) in kernels.items():
                per_kernel[kernel][counter + "_per_launch"] = total / launches
                per_kernel[kernel]["launches"] = launches
        # ... and by kernel class (the wavefront: what its trace and its shade side issue per camera sample, and at what lane utilisation)
        by_class = collections.defaultdict(lambda: collections.defaultdict(float))
        for counter, kernels in valu.items():
            for kernel, (launches, total) in kernels.items():
                by_class[kernel_class(kernel)][counter] += total
        entry["valu_by_class"] = {
            name: {"wave_instructions_per_sample": c["SQ_INSTS_VALU"] / samples,
                   "lane_utilisation": (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] / 64.0) if c["SQ_ACTIVE_INST_VALU"] > 0 else None}
            for name, c in sorted(by_class.items()) if name != "setup"}
    fetch = totals(prefix + "_fetch")["FETCH_SIZE"]
    write = totals(prefix + "_write")["WRITE_SIZE"]
    classes = collections.defaultdict(float)
    for kernel, (launches, total) in fetch.items():
        classes[kernel_class(kernel)] += 2.0 * total * 1024.0
        per_kernel[kernel]["FETCH_SIZE_KiB_per_launch"] = total / launches
        per_kernel[kernel]["launches"] = launches
    for kernel, (launches, total) in write.items():
        classes[kernel_class(kernel)] += total * 1024.0
        per_kernel[kernel]["WRITE_SIZE_KiB_per_launch"] = total / launches
    entry["hbm_bytes_per_sample"] = {name: value / samples for name, value in sorted(classes.items()) if name != "setup"}
    entry["setup_hbm_bytes"] = classes.get("setup", 0.0)
    entry["kernels"] = {kernel: dict(sorted(values.items())) for kernel, values in sorted(per_kernel.items())}
    return entry


summary = {
    "note": "rocprofv3 --pmc passes of tools/pmc_per_sample.sh, one counter group per pass, over exactly the timed path's kernels; "
            "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); large_bvh is ONE 64-spp call "
            "(ramp-up and drain of the slot pool included: 61 % of the slot visits carry a ray, a long render does better)",
    "tag": tag,
    "cornell_1024": workload("cornell", 1024 * 1024 * 256, True),
    "large_bvh": workload("dragon", 1920 * 1080 * 64, os.path.isdir(os.path.join(base, "dragon_valu"))),
}
out = os.path.join(ROOT, "profiles", "pmc_per_sample.json")
json.dump(summary, open(out, "w"), indent=1)
for name in ("cornell_1024", "large_bvh"):
    entry = summary[name]
    print(name, {k: v for k, v in entry.items() if k not in ("kernels",)})
