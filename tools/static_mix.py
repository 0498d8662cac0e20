#!/usr/bin/env python3
"""Static instruction mix of the device kernels, from the compiler's own assembly (no GPU needed).

    tools/static_mix.py [--asm /tmp/pathed_hip.s] [--kernel REGEX] [--json out.json]

Compiles pathed_amd/csrc/pathed_hip.hip to gfx950 assembly (hipcc -S --offload-device-only; about two minutes) unless --asm names
an existing file, then prints per kernel: VALU instructions by class -- packed fp32 (v_pk_*), transcendental (v_rcp / v_rsq /
v_sqrt / v_exp / v_log / v_sin / v_cos: quarter rate), 64-bit / double, everything else ("plain") --, SALU, LDS, vector memory,
scalar memory, and the instructions that make up IEEE divisions (v_div_scale / v_div_fmas / v_div_fixup).  With --loops the same
per basic block that ends in a backward branch (the innermost loops), which with the scene's trip counts gives the dynamic mix
bench.py's roofline block quotes.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
        return "valu_div_helpers"
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_") and ("_f64" in op or "_u64" in op or "_i64" in op or "_b64" in op):
        return "valu_64"
    if op.startswith("v_"):
        return "valu_plain"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def parse(path):
    kernels = collections.OrderedDict()
    name, blocks, label = None, None, None
    instruction = re.compile(r"^\s+([a-z][a-z0-9_]+)\b(.*)$")
    for line in open(path, errors="replace"):
        if line.startswith("_Z") and ":" in line:
            name = line.split(":", 1)[0].strip()
            blocks = collections.OrderedDict()
            label = "entry"
            blocks[label] = {"ops": collections.Counter(), "targets": []}
            kernels[name] = blocks
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        if line.startswith(".LBB") and ":" in line:
            label = line.split(":", 1)[0].strip()
            blocks[label] = {"ops": collections.Counter(), "targets": []}
            continue
        found = instruction.match(line)
        if not found or line.lstrip().startswith((".", ";")):
            continue
        op, rest = found.group(1), found.group(2)
        blocks[label]["ops"][classify(op)] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            target = rest.strip().split()[0] if rest.strip() else ""
            blocks[label]["targets"].append(target)
    return kernels


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--asm", default="")
    parser.add_argument("--kernel", default="k_path_small|k_path_hybrid|k_path_wave|k_shade|k_trace<")
    parser.add_argument("--loops", action="store_true")
    parser.add_argument("--json", default="")
    args = parser.parse_args()
    path = args.asm
    if not path:
        path = "/tmp/pathed_hip.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Iinclude",
                        "--offload-device-only", "-S", "-o", path, "pathed_amd/csrc/pathed_hip.hip"], cwd=ROOT, check=True)
    kernels = parse(path)
    pretty = demangle(list(kernels))
    out = {}
    for name, blocks in kernels.items():
        shown = pretty[name].replace("pathed::", "").replace("void ", "")
        if not re.search(args.kernel, shown):
            continue
        total = collections.Counter()
        for block in blocks.values():
            total.update(block["ops"])
        valu = sum(v for k, v in total.items() if k.startswith("valu"))
        row = dict(total)
        row["valu_total"] = valu
        out[shown] = row
        print("%s\n    VALU %d = plain %d + packed %d + transcendental %d + division helpers %d + 64-bit %d | SALU %d  LDS %d  VMEM %d  SMEM %d  branches %d" % (
            shown[:200], valu, total["valu_plain"], total["valu_packed"], total["valu_trans"], total["valu_div_helpers"], total["valu_64"],
            total["salu"], total["lds"], total["vmem"], total["smem"], total["branch"]))
        if args.loops:
            order = list(blocks)
            index = {label: i for i, label in enumerate(order)}
            for i, label in enumerate(order):
                for target in blocks[label]["targets"]:
                    if target in index and index[target] <= i:
                        body = collections.Counter()
                        for inner in order[index[target]:i + 1]:
                            body.update(blocks[inner]["ops"])
                        bv = sum(v for k, v in body.items() if k.startswith("valu"))
                        if bv >= 40:
                            print("      loop %s..%s: VALU %d (packed %d, transcendental %d, division helpers %d), LDS %d, VMEM %d, SMEM %d" % (
                                target, label, bv, body["valu_packed"], body["valu_trans"], body["valu_div_helpers"], body["lds"], body["vmem"], body["smem"]))
    if args.json:
        with open(args.json, "w") as handle:
            json.dump(out, handle, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
