#!/usr/bin/env python3
"""Condense two rocprofv3 --pmc passes of SQ counters into per-kernel ratios (Markdown).

    python profiles/summarize_sq.py <pass1 counter_collection.csv> <pass2 counter_collection.csv>
"""
import csv
import re
import sys
from collections import defaultdict


def agg(path):
    d = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            if not name.startswith("vof::"):
                continue
            d[name][r["Counter_Name"]] += float(r["Counter_Value"])
            n[name].add(r["Dispatch_Id"])
    return d, {k: len(v) for k, v in n.items()}


def main():
    a, na = agg(sys.argv[1])
    b, _ = agg(sys.argv[2])
    names = sorted(a, key=lambda k: -a[k].get("SQ_BUSY_CYCLES", 0))[:8]
    print("| kernel | launches | waves | wave cycles waiting on any instr. | ... issuing | VALU / wave | SALU / wave | LDS / wave | VMEM rd / wave | VALU-active share of busy | LDS wait share |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for k in names:
        x, y = a[k], b.get(k, {})
        w = x.get("SQ_WAVES", 0) or 1.0
        wc = x.get("SQ_WAVE_CYCLES", 0) or 1.0
        w2 = w   # same launches in both passes
        print(f"| `{k[:60]}` | {na[k]} | {w:.3g} | {x.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | {x.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | "
              f"{x.get('SQ_INSTS_VALU', 0) / w:.0f} | {x.get('SQ_INSTS_SALU', 0) / w:.0f} | {y.get('SQ_INSTS_LDS', 0) / w2:.0f} | "
              f"{y.get('SQ_INSTS_VMEM_RD', 0) / w2:.0f} | {y.get('SQ_ACTIVE_INST_VALU', 0) / (x.get('SQ_BUSY_CYCLES', 0) or 1):.2f} | "
              f"{y.get('SQ_WAIT_INST_LDS', 0) / wc:.2f} |")


if __name__ == "__main__":
    main()
