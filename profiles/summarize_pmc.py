#!/usr/bin/env python3
"""HBM traffic per launch of the hot kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes as MI355X_MICROARCH.md prescribes: both do not fit the TCC slots of one pass).

Units/corrections (guide, section HBM): counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a coalesced streaming read, so  traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
(Calibration on our own access pattern: k_update_s reads 4 and writes 2 float64 vectors of 3*npts per active
pair; the corrected read figure matches the byte count to ~1 %, the write figure is exact.)

    python profiles/summarize_pmc.py <fetch/counter_collection.csv> <write/counter_collection.csv> <size> <frames> \
           [--json profiles/traffic.json] > profiles/rNN_pmc_summary.md
"""
import csv
import json
import re
import sys
from collections import defaultdict


def agg(path, counter):
    d = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            key = (name, int(r["Grid_Size"]))
            d[key][0] += 1
            d[key][1] += float(r["Counter_Value"])
    return d


def main():
    fetch = agg(sys.argv[1], "FETCH_SIZE")
    write = agg(sys.argv[2], "WRITE_SIZE")
    size, frames = int(sys.argv[3]), int(sys.argv[4])
    jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    print("| kernel | grid (threads) | launches | FETCH_SIZE x2 /launch (MB) | WRITE_SIZE /launch (MB) | HBM traffic /launch (MB) |")
    print("|---|---:|---:|---:|---:|---:|")
    rows = []
    for key in sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, [0, 0])[1])):
        if not key[0].startswith("vof::"):
            continue
        n, fk = fetch[key]
        wn, wk = write.get(key, [0, 0.0])
        rd = 2 * fk / n * 1024
        wr = (wk / wn * 1024) if wn else 0.0
        rows.append((key, n, rd, wr))
    for key, n, rd, wr in rows[:24]:
        print(f"| `{key[0][:70]}` | {key[1]} | {n} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | {(rd + wr) / 1e6:.1f} |")
    if jpath:
        # level of a sweep / residual kernel = rank of its grid size among the grids of the same symbol
        out = {}
        by_sym = defaultdict(list)
        for key, n, rd, wr in rows:
            by_sym[key[0]].append((key[1], n, rd, wr))
        # Level-0-only kernels: the band height adapts to the number of active pairs, so one solve launches the same
        # symbol with several grid sizes -> per-launch mean over all of them (what bench.py's per-launch figure is).
        # Stored-level kernels share a symbol across levels: label the grid with the largest total traffic (level 1).
        # bench.py's classes: gs0 = every level-0 smoother pass (k_sweep0r / k_sweep0m / k_sweep0 / k_sweep<SweepFine> instantiations),
        # apply0 = the level-0 operator kernels.  Their per-launch figure is the mean over ALL launches of the class.
        merged = {"gs0": ("k_sweep0r<", "k_sweep0m<", "k_sweep0<", "k_sweep<vof::SweepFine"), "apply0": ("k_stream_apply0<", "k_stream_resrestrict0<")}
        for nm, pats in merged.items():
            lst = [(sym, t) for sym, l2 in by_sym.items() if any(pt in sym for pt in pats) for t in l2]
            if not lst:
                continue
            n = sum(t[1] for _, t in lst)
            rd = sum(t[1] * t[2] for _, t in lst) / n
            wr = sum(t[1] * t[3] for _, t in lst) / n
            out[f"{nm}_L0_{size}x{size}x{frames}"] = {"hbm_bytes_per_launch": rd + wr, "fetch_x2_bytes": rd, "write_bytes": wr, "launches": n,
                                                     "kernel": sorted({sym for sym, _ in lst}), "grid": sorted({t[0] for _, t in lst})}
        # stored-level kernels share a symbol across levels: label the grid with the largest total traffic (level 1)
        names = {"k_sweep<vof::SweepStored": ("gs", 1), "k_sweep_st<": ("gs", 1), "k_apply<": ("residual", 1),
                 "k_resrestrict_u<": ("residual", 1)}
        for sym, lst in by_sym.items():
            for pat, (nm, level) in names.items():
                if pat not in sym:
                    continue
                grid, n, rd, wr = max(lst, key=lambda t: t[1] * (t[2] + t[3]))
                k = f"{nm}_L{level}_{size}x{size}x{frames}"
                if k in out and out[k]["launches"] * out[k]["hbm_bytes_per_launch"] >= n * (rd + wr):
                    continue       # several template instantiations of one class: keep the busiest
                out[k] = {"hbm_bytes_per_launch": rd + wr, "fetch_x2_bytes": rd, "write_bytes": wr,
                          "launches": n, "kernel": sym, "grid": grid}
        json.dump(out, open(jpath, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
