#!/usr/bin/env python3
"""HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/<round>/pmc_traffic.json + .txt.

    python tools/pmc_traffic.py <out_dir> <label> <kernel-filter> <grid or -> <fetch_dir> <write_dir> [<label> ...]

FETCH_SIZE / WRITE_SIZE are in KB (1024 B). On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section): corrected = x2. WRITE_SIZE is exact for 16-byte streaming stores."""
import collections, csv, glob, json, os, re, sys


def avg(root, flt, grid, ctr):
    n, tot = 0, 0.0
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr or flt not in r["Kernel_Name"]:
                continue
            if grid != "-" and r.get("Grid_Size", "") != grid:
                continue
            n += 1
            tot += float(r["Counter_Value"])
    return (tot / n if n else None), n


out_dir = sys.argv[1]
args = sys.argv[2:]
table, lines = {}, []
for i in range(0, len(args), 5):
    label, flt, grid, fdir, wdir = args[i:i + 5]
    f, nf = avg(fdir, flt, grid, "FETCH_SIZE")
    w, nw = avg(wdir, flt, grid, "WRITE_SIZE")
    if f is None or w is None:
        lines.append(f"{label}: no samples for '{flt}' grid {grid} ({nf} fetch / {nw} write rows)")
        continue
    fetch_b, write_b = 2.0 * f * 1024.0, w * 1024.0
    table[label] = {"bytes": round(fetch_b + write_b), "fetch_bytes_x2_corrected": round(fetch_b), "write_bytes": round(write_b),
                    "launches": [nf, nw], "kernel": flt, "grid": grid,
                    "source": f"{os.path.basename(out_dir.rstrip('/'))}/pmc_traffic.json"}
    lines.append(f"{label:28s} kernel~'{flt}' grid {grid:>8s}: FETCH_SIZE {f:10.0f} KB (x2 = {fetch_b / 1e6:7.1f} MB)  WRITE_SIZE {w:10.0f} KB "
                 f"({write_b / 1e6:7.1f} MB)  total {(fetch_b + write_b) / 1e6:7.1f} MB per launch  [{nf}/{nw} launches]")
os.makedirs(out_dir, exist_ok=True)
for v in table.values():
    v["source"] = "profiles/" + v["source"] + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/profile_round.sh pmc)"
json.dump(table, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
open(os.path.join(out_dir, "pmc_traffic.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
