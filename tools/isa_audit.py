#!/usr/bin/env python3
"""isa_audit.py — does every full-tile load and store of the library carry the non-temporal modifier?

Rounds 1-2 declared every streaming access non-temporal, but hipcc (ROCm 7.2) silently drops the flag when it
legalises loads of <N x i8> vectors: the u8 stream of the headline divide, every u8/i8 convert source and every mask
byte stream of the reductions went out as plain `global_load_*` (found by disassembly in the round-2 review).  The
kernels now move 1-byte cells as unsigned words (`cells<T, N>`, ec_device.hpp); this tool keeps it that way.

It compiles the given translation units of erased-cells_amd/csrc to gfx950 assembly (device side only, no GPU
needed) and checks the invariant the kernels are written to.  In a streaming kernel
  * every global STORE of caller data carries `nt`;
  * every global LOAD carries `nt` — the full tiles, the guarded tail tiles and the single head / tail cells alike
    (`load_cells`, `ld_cell` / `st_cell`, ec_device.hpp) — EXCEPT the deliberate default-policy twin of a full tile's
    operand loads: the host may mark an operand stream that fits the Infinity Cache "cacheable" for a launch
    (`cache_plan`, ec_runtime.hpp), and `load_stream` then takes a wave-uniform branch to the same loads without `nt`.
    So per kernel and per load opcode the plain loads may not outnumber the nt loads (each plain load has its nt twin).
Exempt by name: the cell-wise comparison kernels and the one-workgroup finalize kernels (they read a few KB of partials
that were just written: cached on purpose); in the reduction kernels the STORES are exempt (per-workgroup partials and
the result words, re-read at once).

    python tools/isa_audit.py                       # the default TU list, prints a summary, exit 1 on a finding
    python tools/isa_audit.py ec_binop_div.hip -v   # per kernel
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "erased-cells_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
DEFAULT_TUS = ["ec_abi.hip", "ec_binop_div.hip", "ec_fusedany_c1.hip", "ec_fusedany_c4.hip", "ec_expr_c1.hip", "ec_expr_c2.hip"]
SKIP = re.compile(r"cellwise|finalize")
MEM = re.compile(r"^\s+((?:global|buffer)_(load|store)_\w+)\s+(.*)$")
KERNEL = re.compile(r"^(_Z\S+):")
REDUCTION = re.compile(r"k_min_max_partials|k_mask_count_partials|k_first_diff_partials")


def compile_asm(tu, out):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include", "-Wno-unused-function",
           "-S", "--cuda-device-only", os.path.join(CSRC, tu), "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def audit(asm_path):
    """-> ({kernel: [(kind, n_access, n_without_nt, example), ...]} for kernels that break the rule,
    {kernel: audited access count})"""
    findings = collections.OrderedDict()
    audited = collections.Counter()
    per_kernel = collections.OrderedDict()
    kernel = None
    with open(asm_path) as f:
        for line in f:
            m = KERNEL.match(line)
            if m:
                kernel = m.group(1)
                continue
            m = MEM.match(line)
            if m and kernel:
                has_nt = re.search(r"(^|\s)nt(\s|$)", m.group(3)) is not None
                per_kernel.setdefault(kernel, []).append((m.group(2), has_nt, line.strip()))
    for kernel, accesses in per_kernel.items():
        if SKIP.search(kernel):
            continue
        for kind in ("load", "store"):
            if kind == "store" and REDUCTION.search(kernel):
                continue
            acc = [a for a in accesses if a[0] == kind]
            audited[kernel] += len(acc)
            if kind == "store":
                bad = [a for a in acc if not a[1]]
                if bad:
                    findings.setdefault(kernel, []).append((kind, len(acc), len(bad), bad[0][2]))
                continue
            by_op = collections.defaultdict(lambda: [0, 0, None])  # opcode -> [nt, plain, example of a plain one]
            for a in acc:
                op = a[2].split()[0]
                by_op[op][0 if a[1] else 1] += 1
                if not a[1]:
                    by_op[op][2] = a[2]
            for op, (n_nt, n_plain, ex) in by_op.items():
                if n_plain > n_nt:
                    findings.setdefault(kernel, []).append((f"load ({op}: {n_nt} nt)", n_nt + n_plain, n_plain, ex))
    return findings, audited


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.split("\n")))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tus", nargs="*", default=DEFAULT_TUS)
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--keep", help="directory to keep the .s files in")
    args = ap.parse_args()
    total_bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        outdir = args.keep or tmp
        os.makedirs(outdir, exist_ok=True)
        for tu in args.tus:
            asm = os.path.join(outdir, os.path.splitext(os.path.basename(tu))[0] + ".s")
            compile_asm(tu, asm)
            findings, audited = audit(asm)
            names = demangle(list(audited))
            n_acc = sum(audited.values())
            print(f"{tu}: {len(audited)} streaming kernels, {n_acc} global accesses audited, "
                  f"{len(findings)} kernels with an access lacking nt")
            if args.verbose:
                for k, n in audited.items():
                    print(f"    {n:4d}  {names[k][:140]}")
            for k, rows in findings.items():
                total_bad += 1
                print(f"  MISSING nt  {names.get(k, k)[:160]}")
                for kind, n, nbad, example in rows:
                    print(f"      {nbad} of {n} {kind}s, e.g. `{example}`")
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
