#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/jobs/r03pmc.sh: one directory per (workload, counter group), named
<workload>__<group>, each holding a *counter_collection.csv.

Per workload and kernel: dispatches, and every counter's mean per dispatch (summed over the TCC instances / XCDs the
CSV lists separately).  Per workload: HBM bytes per STEP (one step = every kernel of the workload once) from
FETCH_SIZE / WRITE_SIZE as MI355X_MICROARCH.md §HBM prescribes — FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies a 128-byte
read request at 64 B), WRITE_SIZE (KiB) x 1024 — next to the algorithmic bytes, and where the request counters were
taken the same bytes WITHOUT any correction factor: TCC_EA0_RDREQ_128B x 128 + (RDREQ - RDREQ_128B - RDREQ_32B) x 64 +
RDREQ_32B x 32 read, WRREQ_64B x 64 + (WRREQ - WRREQ_64B) x 32 written.

    python tools/pmc_summary.py gpurun_out/r03pmc --steps 6 --out-json ... --out-md ... [--traffic profiles/traffic.json --commit abc --round 3]
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

CELLS = 16384 * 16384
# workload -> (algorithmic bytes per cell per step, input bytes per cell per step, kernels that make up one step)
WORKLOADS = {
    "div": (11, 3, ["k_binop_direct"]),
    "masked_chain": (42, 24, ["k_masked_binop"]),
    "masked_chain_fused": (24, 15, ["k_fused_any", "k_fused_same"]),
    "minmax": (2, 2, ["k_min_max_partials", "k_min_max_finalize"]),
    "ndvi_fused": (12, 4, ["k_fused_any", "k_fused_same"]),
    "ndvi_fused_mixed": (14, 6, ["k_fused_any", "k_fused_mixed"]),
    "binop_add_u16_u16": (12, 4, ["k_binop_direct"]),
    "binop_add_f32_f32": (16, 8, ["k_binop_direct"]),
    "evi_fused": (14, 6, ["k_expr"]),
    "evi_fused_compiled": (14, 6, ["ec_expr_jit"]),
    "evi_fused_builtin": (14, 6, ["k_expr_fixed"]),
    "evi": (130, 66, ["k_binop_direct", "k_binop_scalar"]),
    "binop_add_f64_u16_rule": (18, 10, ["k_binop_lds"]),      # the LDS-staged kernel by rule (round 4)
    "binop_add_f64_u16_direct": (18, 10, ["k_binop_direct"]),  # the same launch with binop_variant = 0
}
TRAFFIC_KEY = {"div": "binop_div_u8_u16"}


def read_dir(d, kernels):
    """-> {kernel name: {counter: [per-dispatch totals]}}"""
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = r.get("Kernel_Name", "")
                if not any(s in k for s in kernels):
                    continue
                per[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: list(v.values()) for c, v in cs.items()} for k, cs in per.items()}


def short(k):
    k = re.sub(r"^void ecd::", "", k)
    return re.sub(r"\(.*$", "", k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--steps", type=int, default=6, help="steps each pass executed (warm-up + timed)")
    ap.add_argument("--out-json")
    ap.add_argument("--out-md")
    ap.add_argument("--traffic", help="profiles/traffic.json to update")
    ap.add_argument("--commit", default="?")
    ap.add_argument("--round", type=int, default=3)
    a = ap.parse_args()
    out = {}
    md = ["| workload | kernel | dispatches | counter | mean per dispatch |", "|---|---|---:|---|---:|"]
    for wl, (bpc, inb, kernels) in WORKLOADS.items():
        rec = {"kernels": {}, "algorithmic_bytes_per_step": bpc * CELLS, "input_bytes_per_step": inb * CELLS}
        totals = collections.defaultdict(float)   # counter -> sum over all dispatches of all kernels of the workload
        for d in sorted(glob.glob(os.path.join(a.root, wl + "__*"))):
            if not os.path.isdir(d):
                continue
            for k, cs in read_dir(d, kernels).items():
                kr = rec["kernels"].setdefault(short(k), {"full_name": k})
                for c, vals in cs.items():
                    kr[c] = {"dispatches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
                    totals[c] += sum(vals)
                    md.append(f"| {wl} | `{short(k)}` | {len(vals)} | {c} | {sum(vals) / len(vals):,.0f} |")
        if not rec["kernels"]:
            continue
        per_step = {c: v / a.steps for c, v in totals.items()}
        rec["per_step"] = per_step
        if "FETCH_SIZE" in per_step and "WRITE_SIZE" in per_step:
            fetch, write = per_step["FETCH_SIZE"] * 1024 * 2, per_step["WRITE_SIZE"] * 1024
            rec["hbm_bytes_per_step_fetch_write"] = fetch + write
            rec["fetch_bytes_corrected"] = fetch
            rec["write_bytes"] = write
            rec["ratio_to_algorithmic"] = (fetch + write) / (bpc * CELLS)
            rec["fetch_over_input_bytes"] = fetch / (inb * CELLS)
        if "TCC_EA0_RDREQ_sum" in per_step:
            rd, rd128 = per_step["TCC_EA0_RDREQ_sum"], per_step.get("TCC_EA0_RDREQ_128B_sum", 0.0)
            rd32 = per_step.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            wr, wr64 = per_step["TCC_EA0_WRREQ_sum"], per_step.get("TCC_EA0_WRREQ_64B_sum", 0.0)
            rb = rd128 * 128 + max(0.0, rd - rd128 - rd32) * 64 + rd32 * 32
            wb = wr64 * 64 + max(0.0, wr - wr64) * 32
            rec["hbm_bytes_per_step_requests"] = rb + wb
            rec["requests_ratio_to_algorithmic"] = (rb + wb) / (bpc * CELLS)
            rec["read_requests_full_128B_share"] = rd128 / rd if rd else None
            rec["write_requests_64B_share"] = wr64 / wr if wr else None
            if "TCC_EA0_RDREQ_LEVEL_sum" in per_step and rd:
                rec["read_latency_cycles"] = per_step["TCC_EA0_RDREQ_LEVEL_sum"] / rd
            if "TCC_EA0_WRREQ_LEVEL_sum" in per_step and wr:
                rec["write_residency_cycles"] = per_step["TCC_EA0_WRREQ_LEVEL_sum"] / wr
            if "GRBM_GUI_ACTIVE" in per_step:
                act = per_step["GRBM_GUI_ACTIVE"]  # summed over the 8 XCDs
                rec["gui_active_cycles_sum_over_xcds"] = act
                for c in ("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum"):
                    if c in per_step:
                        # 16 TCC instances per XCD share the XCD's active cycles: stall share = stalls / (16 x active)
                        rec[c.replace("_sum", "") + "_share_of_active"] = per_step[c] / (16.0 * act) if act else None
        out[wl] = rec
    if a.out_json:
        json.dump(out, open(a.out_json, "w"), indent=1)
    lines = ["| workload | alg. bytes/step | FETCH x2 + WRITE | ratio | request-counter bytes | ratio | 128-B reads | 64-B writes | read latency (cyc) | WR credit stall | RD credit stall |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for wl, r in out.items():
        f = lambda x, fmt: (fmt % x) if x is not None else "—"
        lines.append("| %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
            wl, f"{r['algorithmic_bytes_per_step']:,}", f(r.get("hbm_bytes_per_step_fetch_write"), "%.0f"), f(r.get("ratio_to_algorithmic"), "%.4f"),
            f(r.get("hbm_bytes_per_step_requests"), "%.0f"), f(r.get("requests_ratio_to_algorithmic"), "%.4f"),
            f(r.get("read_requests_full_128B_share"), "%.4f"), f(r.get("write_requests_64B_share"), "%.4f"),
            f(r.get("read_latency_cycles"), "%.0f"), f(r.get("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_share_of_active"), "%.3f"),
            f(r.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_share_of_active"), "%.3f")))
    text = "\n".join(lines) + "\n\n" + "\n".join(md) + "\n"
    if a.out_md:
        open(a.out_md, "w").write(text)
    else:
        print(text)
    if a.traffic:
        try:
            cur = json.load(open(a.traffic))
        except Exception:
            cur = {}
        for wl, r in out.items():
            if "hbm_bytes_per_step_fetch_write" not in r:
                continue
            key = TRAFFIC_KEY.get(wl, wl)
            cur[key] = {
                "cells_per_launch": CELLS, "hbm_bytes_per_launch": r["hbm_bytes_per_step_fetch_write"],
                "fetch_bytes_corrected": r["fetch_bytes_corrected"], "write_bytes": r["write_bytes"],
                "algorithmic_bytes": r["algorithmic_bytes_per_step"], "ratio_to_algorithmic": r["ratio_to_algorithmic"],
                "fetch_over_input_bytes": r["fetch_over_input_bytes"],
                "request_counter_bytes": r.get("hbm_bytes_per_step_requests"),
                "kernels": [k["full_name"] for k in r["kernels"].values()], "commit": a.commit, "round": a.round,
                "method": "per STEP of the workload (every kernel of the step once): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                          "(+ --kernel-trace) in separate passes over `python3 bench.py --workload ... --steps 5 --warmup 1 --ramp 0`, program "
                          "directly after `--`; FETCH_SIZE (KiB) x1024 x2 per MI355X_MICROARCH.md §HBM, WRITE_SIZE (KiB) x1024; totals over the "
                          "workload's kernels / 6 steps; request_counter_bytes: TCC_EA0_RDREQ/WRREQ by request size, no correction factor "
                          "(tools/pmc_summary.py, tools/jobs/r03pmc.sh / r04pmc.sh)"}
        json.dump(cur, open(a.traffic, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
