"""Register budgets the two-pool overlap depends on ("Two slot pools"): two 96-VGPR trace waves and
three k_shade waves (round 1: three and two) share a SIMD only while k_shade stays at 104 VGPRs or fewer; one register
more cost 6-8 % on the mesh scenes every time it was tried (profiles/r1s2_ab_*.log).  Compile-only: hipcc reports the
usage per kernel."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def resource_usage(tmp_path, flags=()):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    compiler = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
    result = subprocess.run(
        [compiler, "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Iinclude", "-mllvm", "-instcombine-max-copied-from-constant-users=4000", *flags,
         "-Rpass-analysis=kernel-resource-usage", "-c", "pathed_amd/csrc/pathed_hip.hip", "-o", str(tmp_path / "probe.o")],
        cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stderr[-2000:]
    usage = {}
    name = None
    for line in result.stderr.splitlines():
        found = re.search(r"Function Name: (\S+)", line)
        if found:
            name = found.group(1)
            usage[name] = {}
        for key in ("VGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]", "LDS Size \\[bytes/block\\]"):
            value = re.search(key + r": (\d+)", line)
            if value and name:
                usage[name][key.split(" ")[0]] = int(value.group(1))
    return usage


def test_vgpr_budgets_of_the_hot_kernels(tmp_path):
    usage = resource_usage(tmp_path)
    shade = [v for k, v in usage.items() if "k_shadeILb1" in k]      # generic and ENV_ONLY instantiations
    assert len(shade) == 2 and all(v["VGPRs"] <= 104 and v["ScratchSize"] == 0 for v in shade), shade
    # [r5] the shade kernel of environment-lit scenes (kernels.h: k_shade_env) reads its 1.6 KB of arguments from the kernel-argument
    # segment: no copy of them in scratch (without the Makefile's -instcombine-max-copied-from-constant-users every lane held a copy)
    env = [v for k, v in usage.items() if "k_shade_env" in k]
    assert len(env) == 2 and all(v["VGPRs"] <= 96 and v["ScratchSize"] <= 32 and v["Occupancy"] == 5 for v in env), env   # (capped at five waves: 12 bytes of spills)
    traces = [v for k, v in usage.items() if re.search(r"k_traceILi\d+ELb[01]ELb0ELb[01]", k)]   # the non-counting variants (plain, LDS-resident tree, sphere-free)
    assert len(traces) == 9 and all(v["VGPRs"] <= 96 for v in traces), traces
    small = [v for k, v in usage.items() if "k_trace_smallILb0" in k]
    assert small and small[0]["VGPRs"] <= 64 and small[0]["ScratchSize"] == 0, small
    # the fused path kernel lives at four waves per SIMD (128 VGPRs; three waves without spills measured 7 % slower):
    # what it spills stays within a couple of dozen dwords.  Template tail: ..., MFMA, QUADS (kernels.h)
    fused = {k: v for k, v in usage.items() if re.search(r"k_path_smallILb1ELb0E.*ELb0ELb0EEEv", k)}   # pair-of-triangles phase 1
    assert len(fused) == 3 and all(v["VGPRs"] <= 128 and v["ScratchSize"] <= 96 and v["Occupancy"] == 4 for v in fused.values()), fused
    # the instantiation narrowed to Cornell-like scenes (Lambertian, triangle lights: shading.h SceneTraits): the code it does
    # not contain is what used to spill (92 bytes of scratch in the generic one)
    narrow = [v for k, v in fused.items() if "SceneTraitsILj1E" in k]
    assert len(narrow) == 1 and narrow[0]["ScratchSize"] <= 16, fused
    # ... and with the parallelogram phase 1 (small_items.h), which is what the headline runs on: the path state that waits
    # in LDS across the pass (24 KiB per block) keeps the spills of the Cornell instantiation at a handful
    quads = {k: v for k, v in usage.items() if re.search(r"k_path_smallILb1ELb0E.*ELb0ELb1EEEv", k)}
    # (Cornell-like, Veach-like, [r5] rough BSDFs over Beckmann / over GGX, Lambertian + glass + mirror, any BSDF under triangle lights, generic)
    assert len(quads) == 7 and all(v["VGPRs"] <= 128 and v["Occupancy"] == 4 for v in quads.values()), quads
    # [r5] the ladder's rungs spill no more than the any-BSDF instantiation they stand in for (shading.h: TraitsRough*, TraitsSmooth)
    any_bsdf = [v for k, v in quads.items() if "SceneTraitsILj63E" in k][0]
    rungs = [v for k, v in quads.items() if "SceneTraitsILj15E" in k or "SceneTraitsILj49E" in k]
    assert len(rungs) == 3 and all(v["ScratchSize"] <= any_bsdf["ScratchSize"] for v in rungs), quads
    narrow_quads = [v for k, v in quads.items() if "SceneTraitsILj1E" in k]
    assert len(narrow_quads) == 1 and narrow_quads[0]["ScratchSize"] <= 160, quads
    # static LDS: the stash + the lists of the shared phase 2 (kernels.h smallResolveShared); with the material table of a
    # scene that pairs triangles (<= 64 materials, 6 KiB) four blocks must fit a CU's 160 KiB
    assert all(v["LDS"] + 64 * 96 <= 160 * 1024 // 4 for v in quads.values()), quads

    # [r5] the hybrid kernel (path_hybrid.h: direct set + a tree of the rest) keeps k_path_small's budget: four waves per SIMD at
    # 128 VGPRs, and its static LDS (the stash + 4.4 KiB per wave) leaves room for four blocks per CU beside a material table
    hybrid = {k: v for k, v in usage.items() if "k_path_hybrid" in k}
    assert len(hybrid) == 4 and all(v["VGPRs"] <= 128 and v["Occupancy"] == 4 for v in hybrid.values()), hybrid
    assert all(v["LDS"] + 16 * 96 <= 160 * 1024 // 4 for v in hybrid.values()), hybrid
    smooth = [v for k, v in hybrid.items() if "SceneTraitsILj49E" in k]
    assert len(smooth) == 1 and smooth[0]["ScratchSize"] <= 192, hybrid   # (parked-ray words and the burst state: 164 bytes today)


@pytest.mark.skipif(not os.environ.get("PATHED_TEST_EXPERIMENTS"), reason="compiles the experiments build (minutes): set PATHED_TEST_EXPERIMENTS=1")
def test_vgpr_budgets_of_the_experimental_kernels(tmp_path):
    """The measured-and-rejected organisations (kernels_experiments.h, `make experiments`) keep the budgets they were measured at."""
    usage = resource_usage(tmp_path, ("-DPATHED_EXPERIMENTS=1",))
    # the variant over compressed nodes holds 16 dwords of node instead of 28: nothing spills; the 8-wide one holds 28 and
    # eight keys, refs and ranks: it spills (one of the reasons it loses)
    packed = [v for k, v in usage.items() if re.search(r"k_traceILi\d+ELb0ELb0ELb0ELb0ELi1E", k)]
    assert len(packed) == 3 and all(v["ScratchSize"] <= 8 for v in packed), packed
    packed8 = [v for k, v in usage.items() if re.search(r"k_traceILi\d+ELb0ELb0ELb0ELb0ELi2E", k)]
    assert len(packed8) == 3 and all(v["ScratchSize"] <= 96 for v in packed8), packed8
    # the list-writing variant must not spill more than a few dwords beyond the plain one: every value its list code kept
    # alive across the traversal loop was a reload inside it (+45 % kernel time, profiles/r3_ab_split_shade.log)
    plain = [v for k, v in usage.items() if re.search(r"k_traceILi22ELb0ELb0ELb0ELb1ELi0E", k)][0]
    lists = [v for k, v in usage.items() if re.search(r"k_traceILi22ELb0ELb0ELb1ELb1ELi0E", k)][0]
    assert lists["ScratchSize"] <= plain["ScratchSize"] + 16, (plain, lists)
    split = [v for k, v in usage.items() if "k_vertexILb1" in k or "k_regen" in k]
    assert len(split) == 2 and all(v["VGPRs"] <= 128 and v["ScratchSize"] <= 32 for v in split), split
    staged = [v for k, v in usage.items() if "k_shade_stagedILb1" in k]
    assert len(staged) == 2 and all(v["VGPRs"] <= 128 and v["ScratchSize"] == 0 for v in staged), staged
