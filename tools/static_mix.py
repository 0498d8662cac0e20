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
    parser.add_argument("--bench-json", default="", help="write the trip-count-weighted mix of the Cornell instantiation of k_path_small (what bench.py times) here")
    parser.add_argument("--quad-pairs", type=int, default=9, help="trips of the parallelogram-pair loop per pass (Cornell: 17 parallelograms)")
    parser.add_argument("--lone-pairs", type=int, default=1, help="trips of the lone-triangle-pair loop per pass (Cornell: 2 triangles)")
    args = parser.parse_args()
    path = args.asm
    if not path:
        path = "/tmp/pathed_hip.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Iinclude", "-mllvm", "-instcombine-max-copied-from-constant-users=4000",
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
    if args.bench_json:
        write_bench_mix(kernels, pretty, args)
    return 0


def sources_digest():
    import hashlib
    sha = hashlib.sha1()
    directory = os.path.join(ROOT, "pathed_amd", "csrc")
    for name in sorted(os.listdir(directory)):
        sha.update(open(os.path.join(directory, name), "rb").read())
    return sha.hexdigest()[:16]


def write_bench_mix(kernels, pretty, args):
    """One iteration of the fused kernel's loop as the compiler laid it out, every block once, the two phase-1 loops times
    their trip counts on the benchmarked scene: the mix bench.py prices against the box's probed issue rates."""
    wanted = r"k_path_small<true, false, SceneTraits<1u, false, true, false, false, true, 3u>, false, true>"
    for name, blocks in kernels.items():
        shown = pretty[name].replace("pathed::", "").replace("void ", "")
        if not re.search(wanted, shown):
            continue
        total = collections.Counter()
        for block in blocks.values():
            total.update(block["ops"])
        loops = []
        for label, block in blocks.items():
            if label in block["targets"] and block["ops"]["valu_packed"] >= 40:
                loops.append((block["ops"]["valu_packed"], label))
        loops.sort(reverse=True)
        weights = {}
        if len(loops) >= 1:
            weights[loops[0][1]] = args.quad_pairs
        if len(loops) >= 2:
            weights[loops[1][1]] = args.lone_pairs
        weighted = collections.Counter()
        for label, block in blocks.items():
            for key, count in block["ops"].items():
                weighted[key] += count * weights.get(label, 1)
        valu = {k: v for k, v in weighted.items() if k.startswith("valu")}
        result = {
            "kernel": shown, "kernel_sources": sources_digest(),
            "method": "compiler assembly (hipcc -S), every basic block once, the parallelogram-pair loop x %d and the lone-pair loop x %d "
                      "(their trip counts on scenes/cornell.json); v_div_scale / v_div_fmas / v_div_fixup counted with the plain instructions" % (args.quad_pairs, args.lone_pairs),
            "loops": {label: {"trips": trips, "valu": sum(v for k, v in blocks[label]["ops"].items() if k.startswith("valu")),
                              "packed": blocks[label]["ops"]["valu_packed"]} for label, trips in weights.items()},
            "valu_per_iteration": sum(valu.values()),
            "classes": {"plain": weighted["valu_plain"] + weighted["valu_div_helpers"], "packed": weighted["valu_packed"],
                        "transcendental": weighted["valu_trans"], "wide": weighted["valu_64"]},
            "static_valu": sum(v for k, v in total.items() if k.startswith("valu")),
        }
        with open(args.bench_json, "w") as handle:
            json.dump(result, handle, indent=1)
        print("bench mix: %s" % json.dumps(result["classes"]))
        return
    raise SystemExit("static_mix.py: the Cornell instantiation of k_path_small was not found in the assembly")


if __name__ == "__main__":
    sys.exit(main())
