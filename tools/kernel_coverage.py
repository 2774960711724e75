#!/usr/bin/env python3
"""Which kernel instantiations of liberased_cells_hip.so does the GPU test suite execute?  (dev tool)

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $OUT/cov --output-format csv -- python3 -m pytest /root/repo/tests -q -m gpu -k "<no subprocess tests>"
  python tools/kernel_coverage.py $OUT/cov > profiles/r02/kernel_instantiation_coverage.md

Built = the device stubs in the library (`nm -C | grep __device_stub__`); executed = the kernel names in the run's
kernel stats.  Names are compared up to the end of their template argument list."""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "erased-cells_amd", "liberased_cells_hip.so")


def head(name: str) -> str:
    """`ecd::k_x<a, b<c>>(args)` -> `k_x<a,b<c>>` (name + template arguments, no parameter list, no spaces)."""
    name = name.replace("void ", "").replace("__device_stub__", "").strip()
    name = re.sub(r"^ecd::", "", name)
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return re.sub(r"\s+", "", "".join(out)).replace("ecd::", "")


def main():
    prof = sys.argv[1]
    nm = subprocess.run(["nm", "-C", SO], capture_output=True, text=True).stdout
    built = {head(ln.split(" ", 2)[2]) for ln in nm.splitlines() if "__device_stub__" in ln}
    ran = set()
    for path in glob.glob(os.path.join(prof, "**", "*kernel_stats.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                ran.add(head(r["Name"]))
    fam = lambda n: n.split("<", 1)[0]
    families = sorted({fam(n) for n in built})
    print("Kernel-instantiation coverage of the GPU test suite: kernel names of `rocprofv3 --kernel-trace --stats -- python3 -m pytest "
          "tests -m gpu` against the device stubs in `liberased_cells_hip.so` (`tools/kernel_coverage.py`).\n")
    print("| kernel family | built | executed by the tests |")
    print("|---|---:|---:|")
    tb = tr = 0
    missing = []
    for f_ in families:
        b = {n for n in built if fam(n) == f_}
        r = b & ran
        tb, tr = tb + len(b), tr + len(r)
        missing += sorted(b - r)
        print(f"| `{f_}` | {len(b)} | {len(r)} |")
    print(f"| total | {tb} | {tr} |")
    if missing:
        print("\nNot executed:\n")
        for n in missing[:60]:
            print(f"* `{n}`")
        if len(missing) > 60:
            print(f"* ... and {len(missing) - 60} more")


if __name__ == "__main__":
    main()
