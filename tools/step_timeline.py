#!/usr/bin/env python3
"""Timeline of one graph-replayed denoising step from a rocprofv3 kernel trace: every kernel's duration and the idle gap in front of it.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl -o t -- python3 $GRAFT_REPO_ROOT/bench.py \
        --steps 40 --warmup 5 --no-cpu-baseline --train-steps 0 --no-vae --no-roofline --no-full-call --no-phosc
    python3 tools/step_timeline.py gpurun_out/tl/*/t_kernel_trace.csv

The step is found as the repeating sequence between two launches of the first kernel of the plan (wd_select_rows); the median step
of the trace is printed (kernel time, gap time, per-kernel table)."""
import csv
import re
import sys
from statistics import median


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def main():
    rows = []
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("select_rows_kernel") or "select_rows" in r[2]]
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a:b]
        if len(seg) < 20:
            continue
        steps.append(seg)
    if not steps:
        print("no steps found")
        return
    n = max(set(len(s) for s in steps), key=lambda k: sum(1 for s in steps if len(s) == k))
    steps = [s for s in steps if len(s) == n]
    spans = [s[-1][1] - s[0][0] for s in steps]
    mid = sorted(range(len(steps)), key=lambda i: spans[i])[len(steps) // 2]
    seg = steps[mid]
    nxt_start = None
    print(f"{len(steps)} steps of {n} kernels; median span {median(spans) / 1e3:.1f} us")
    tk = tg = 0.0
    prev_end = seg[0][0]
    for s, e, name in seg:
        gap = (s - prev_end) / 1e3
        dur = (e - s) / 1e3
        tk += dur
        tg += max(gap, 0.0)
        print(f"  gap {gap:6.2f} us   {dur:8.2f} us   {name}")
        prev_end = max(prev_end, e)
    print(f"kernels {tk:.1f} us, gaps {tg:.1f} us, sum {tk + tg:.1f} us")


if __name__ == "__main__":
    main()
