#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV into a per-(kernel, grid) table (Markdown).
The multigrid levels share one kernel symbol (k_gs<double> runs on levels 1..6), so the stock
--stats summary mixes them; the grid size separates the levels.

    python profiles/summarize_rocprof.py <kernel_trace.csv> [<kernel_stats.csv>] > profiles/rNN_summary.md
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").strip()


def main():
    trace = sys.argv[1]
    rows = defaultdict(lambda: [0, 0, 10 ** 18, 0, 0])
    total = 0
    with open(trace) as f:
        for r in csv.DictReader(f):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            key = (short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
            e = rows[key]
            e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
            e[4] = int(r["VGPR_Count"])
            total += d
    print("| kernel | grid (threads) | calls | total ms | avg us | min us | max us | % | VGPR |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|")
    for key, e in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
        if not key[0].startswith("vof::") and e[1] < 0.005 * total:
            continue
        print(f"| `{key[0][:60]}` | {key[1]}x{key[2]}x{key[3]} | {e[0]} | {e[1] / 1e6:.3f} | {e[1] / e[0] / 1e3:.2f} | "
              f"{e[2] / 1e3:.2f} | {e[3] / 1e3:.2f} | {100 * e[1] / total:.1f} | {e[4]} |")
    print(f"\nTotal kernel time {total / 1e6:.1f} ms over {sum(e[0] for e in rows.values())} dispatches.")
    if len(sys.argv) > 2:
        print("\nStock `--stats` summary (top rows, per kernel symbol):\n")
        print("| kernel | calls | total ms | avg us | % |")
        print("|---|---:|---:|---:|---:|")
        with open(sys.argv[2]) as f:
            for i, r in enumerate(csv.DictReader(f)):
                if i >= 14:
                    break
                print(f"| `{short(r['Name'])[:60]}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | "
                      f"{float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |")


if __name__ == "__main__":
    main()
