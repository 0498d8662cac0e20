#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace directory and prints, for the path kernels of the LAST step in it: launches, mean and
summed duration per kernel, the wall time their union covers, the time two path kernels run at once, and the idle time
between kernels.  Usage: summarize_timeline.py <dir> [ms_per_step [N]]"""
import csv
import glob
import sys


def main():
    rows = []
    for name in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(name)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pathed::", "")))
    rows.sort()
    # a step starts with k_init launches; take everything after the last large idle gap before a k_init run
    inits = [i for i, r in enumerate(rows) if r[2].startswith("k_init")]
    if not inits:
        print("no k_init in trace")
        return
    # group k_init launches into passes, keep the passes of the last step: the last third of the trace is enough
    resolves = [i for i, r in enumerate(rows) if r[2].startswith("k_resolve")]
    last_resolve = resolves[-1]
    # the window of the last step: its duration as bench.py printed it (ms), counted back from the last k_resolve
    window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else (rows[last_resolve][1] - rows[0][0]) / 3
    begin_time = rows[last_resolve][1] - int(window)
    step = [r for r in rows[:last_resolve + 1] if r[0] >= begin_time]
    t0, t1 = step[0][0], max(r[1] for r in step)
    print("  last step: %.2f ms, %d launches" % ((t1 - t0) / 1e6, len(step)))
    by = {}
    for s, e, k in step:
        by.setdefault(k, []).append(e - s)
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print("  %-34s n=%5d mean %8.1f us  sum %8.2f ms" % (k[:34], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
    if len(sys.argv) > 3:   # series: every Nth launch of the two path kernels, "offset ms: duration us"
        every = int(sys.argv[3])
        for kernel in ("k_trace", "k_shade"):
            launches = [(s, e) for s, e, k in step if k.startswith(kernel)]
            print("  %s series (offset ms:duration us): %s" % (kernel, " ".join(
                "%.1f:%.0f" % ((s - t0) / 1e6, (e - s) / 1e3) for s, e in launches[::every])))
    events = []
    for s, e, k in step:
        events.append((s, 1))
        events.append((e, -1))
    events.sort()
    depth, last, cover, overlap = 0, t0, 0, 0
    gaps = []
    for t, d in events:
        if depth >= 1:
            cover += t - last
        if depth >= 2:
            overlap += t - last
        if depth == 0 and t > last:
            gaps.append(t - last)
        depth += d
        last = t
    print("  covered %.2f ms, two or more kernels at once %.2f ms, idle %.2f ms in %d gaps (mean %.1f us, max %.1f us)" % (
        cover / 1e6, overlap / 1e6, sum(gaps) / 1e6, len(gaps), (sum(gaps) / max(1, len(gaps))) / 1e3, max(gaps + [0]) / 1e3))


if __name__ == "__main__":
    main()
