#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes -> profiles/rNN_pmc_hbm_traffic.json (bench.py `roofline.traffic`).

FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 (MI355X_MICROARCH.md, "rocprofv3 PMC slots"), and PMC passes must not be
combined with the sys / runtime trace domains, so on the GPU box:

    cd /tmp && export TMPDIR=/tmp
    B="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --train-steps 0 --no-vae --no-roofline"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o f -- $B
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o w -- $B
    python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch/f_counter_collection.csv \
            $GRAFT_REPO_ROOT/gpurun_out/pmc_write/w_counter_collection.csv > $GRAFT_REPO_ROOT/gpurun_out/pmc_hbm_traffic.json

Units and the gfx950 correction as that guide's HBM section prescribes: both counters are in KB; FETCH_SIZE tallies the 128-byte
requests of wide streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  Launches are
grouped by kernel and grid size (one kernel name covers several layer shapes); the group with the most launches of a kernel gets
the bare name as its key, the others "name (grid N)".
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return {k: (n, s / n) for k, (n, s) in acc.items()}


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    by_name = defaultdict(list)
    for (name, grid), (n, kb) in fetch.items():
        by_name[name].append((n, grid, kb))
    out = {}
    for name, groups in sorted(by_name.items(), key=lambda kv: -sum(g[0] * g[2] for g in kv[1])):
        groups.sort(reverse=True)
        for i, (n, grid, kb) in enumerate(groups):
            key = name if i == 0 else f"{name} (grid {grid})"
            wn, wkb = write.get((name, grid), (0, 0.0))
            out[key] = dict(launches=n, grid=grid, fetch_kb_raw=kb, fetch_mb_corrected=2.0 * kb * 1024 / 1e6,
                            write_mb=wkb * 1024 / 1e6)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
