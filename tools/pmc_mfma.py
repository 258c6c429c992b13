#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass -> profiles/rNN_pmc_mfma_util.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o m -- \
        python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --train-steps 0 --no-vae --no-roofline --no-full-call --no-phosc
    python3 tools/pmc_mfma.py $OUT/pmc_mfma/*/m_counter_collection.csv > $OUT/pmc_mfma_util.json

The formula is rocprofv3's own derived metric (`rocprofv3 -L`: MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100);
the CSV carries GRBM_GUI_ACTIVE summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), hence the / 8; SIMD_NUM = 256 CUs x 4.
SQ_VALU_MFMA_BUSY_CYCLES counts 16 cycles per v_mfma_f32_16x16x32_bf16: with the three passes of a split-bf16 product, utilisation x 2.5 PF
x (clock / 2.4 GHz) is the MFMA ISSUE rate, three times the algorithmic one."""
import csv
import json
import re
import sys
from collections import defaultdict

SIMDS = 256 * 4


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            e = acc[k][r["Counter_Name"]]
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    out = {}
    by_name = defaultdict(list)
    for (name, grid), d in acc.items():
        n, busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])
        n2, act = d.get("GRBM_GUI_ACTIVE", [0, 0.0])
        if not n or not act:
            continue
        by_name[name].append((n, grid, busy / n, act / n2))
    for name, groups in sorted(by_name.items(), key=lambda kv: -sum(g[0] * g[2] for g in kv[1])):
        groups.sort(reverse=True)
        for i, (n, grid, busy, act) in enumerate(groups):
            key = name if i == 0 else f"{name} (grid {grid})"
            out[key] = dict(launches=n, grid=grid, mfma_busy_cycles=busy, gui_active_cycles_per_xcd=act / 8,
                            mfma_util_percent=100.0 * busy / ((act / 8) * SIMDS))
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
