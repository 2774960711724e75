#!/usr/bin/env python3
"""Summarise `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over bench.py into the record bench.py
reports as `roofline.traffic` (profiles/traffic.json) — MI355X_MICROARCH.md §HBM:

  * FETCH_SIZE and WRITE_SIZE do not fit one pass (3 + 2 of the 4 TCC slots) -> two passes;
  * both are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of a streaming read at 64 B, i.e.
    reports exactly half of the bytes -> x2 (checked here against the input bytes, which every cache is too
    small to hold: fetched bytes cannot be below them);
  * WRITE_SIZE is exact for 16-byte-per-lane streaming stores.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 1 ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write --output-format csv -- python3 /root/repo/bench.py --steps 5 --warmup 1 ...
  python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write --kernel k_binop_direct --cells 268435456 \
         --bytes-per-cell 11 --input-bytes-per-cell 3 --commit $(git rev-parse --short HEAD) --round 2 \
         --summary profiles/r02/pmc_fetch_write_summary.json --traffic profiles/traffic.json
"""
import argparse
import csv
import glob
import json
import os
import sys


def counter_rows(d, counter, kernel_substr, grid):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter or kernel_substr not in r.get("Kernel_Name", ""):
                    continue
                if grid and int(r.get("Grid_Size", 0)) != grid:
                    continue
                rows.append(r)
    return rows


def summarise(rows, counter):
    # one row per (dispatch, counter) — or one per XCD/instance on some builds: sum by dispatch first
    per = {}
    for r in rows:
        per.setdefault(r["Dispatch_Id"], 0.0)
        per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = list(per.values())
    r0 = rows[0]
    return {"counter": counter, "kernel": r0["Kernel_Name"], "dispatches": len(vals),
            "mean_KiB": sum(vals) / len(vals), "min_KiB": min(vals), "max_KiB": max(vals),
            "VGPR_Count": r0.get("VGPR_Count"), "SGPR_Count": r0.get("SGPR_Count"),
            "LDS_Block_Size": r0.get("LDS_Block_Size"), "Grid_Size": r0.get("Grid_Size"),
            "Workgroup_Size": r0.get("Workgroup_Size")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--kernel", default="k_binop_direct")
    ap.add_argument("--cells", type=int, default=268435456)
    ap.add_argument("--cells-per-workgroup", type=int, default=1024)
    ap.add_argument("--bytes-per-cell", type=int, default=11)
    ap.add_argument("--input-bytes-per-cell", type=int, default=3)
    ap.add_argument("--key", default="binop_div_u8_u16")
    ap.add_argument("--commit", default="?")
    ap.add_argument("--round", type=int, default=2)
    ap.add_argument("--summary")
    ap.add_argument("--traffic")
    a = ap.parse_args()

    grid = a.cells // a.cells_per_workgroup * 256 if a.cells_per_workgroup else 0
    f = counter_rows(a.fetch_dir, "FETCH_SIZE", a.kernel, grid)
    w = counter_rows(a.write_dir, "WRITE_SIZE", a.kernel, grid)
    if not f or not w:
        f = f or counter_rows(a.fetch_dir, "FETCH_SIZE", a.kernel, 0)
        w = w or counter_rows(a.write_dir, "WRITE_SIZE", a.kernel, 0)
    if not f or not w:
        sys.exit(f"no {a.kernel} dispatches with FETCH_SIZE ({len(f)}) / WRITE_SIZE ({len(w)}) rows found")
    sf, sw = summarise(f, "FETCH_SIZE"), summarise(w, "WRITE_SIZE")
    fetch = sf["mean_KiB"] * 1024 * 2   # gfx950: FETCH_SIZE = half the streamed bytes
    write = sw["mean_KiB"] * 1024
    alg = a.bytes_per_cell * a.cells
    inp = a.input_bytes_per_cell * a.cells
    rec = {"cells_per_launch": a.cells, "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch,
           "write_bytes": write, "algorithmic_bytes": alg, "ratio_to_algorithmic": (fetch + write) / alg,
           "fetch_over_input_bytes": fetch / inp, "kernel": sf["kernel"], "kernel_write_pass": sw["kernel"],
           "commit": a.commit, "round": a.round,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (+ --kernel-trace) in separate passes over "
                     "`python3 bench.py --steps 5 --warmup 1`, program directly after `--`; FETCH_SIZE (KiB) x1024 x2 per "
                     "MI355X_MICROARCH.md §HBM (gfx950 tallies 128-B read requests at 64 B; calibration: fetch_over_input_bytes "
                     "must be >= 1 because the inputs exceed every cache), WRITE_SIZE (KiB) x1024; mean over the dispatches "
                     "of the kernel in each pass"}
    print(json.dumps({"fetch": sf, "write": sw, "record": rec}, indent=1))
    if a.summary:
        os.makedirs(os.path.dirname(a.summary), exist_ok=True)
        json.dump([sf, sw], open(a.summary, "w"), indent=1)
    if a.traffic:
        try:
            cur = json.load(open(a.traffic))
        except Exception:
            cur = {}
        cur[a.key] = rec
        json.dump(cur, open(a.traffic, "w"), indent=1)


if __name__ == "__main__":
    main()
