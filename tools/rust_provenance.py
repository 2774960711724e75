#!/usr/bin/env python3
"""rust_provenance.py — which lines of the Rust crate are the reference's text, and how similar is the rest?

The crate (erased-cells_amd/rust/erased-cells-hip) keeps the reference's public surface: names, signatures, trait and
operator impl headers, the `with_ct!` table.  Those declarations are necessarily the reference's text; they are marked in
the source with comment pairs

    // api-surface(src/encoding.rs:9-40): what it is
    ...
    // end api-surface

This tool lists the marked ranges (the PROVENANCE table of INTEGRATION.md §2 is its `--table` output) and measures, per
crate file, difflib similarity against every file of the reference — with the marked ranges removed and with them in —
by characters and by lines, on the non-test part of both sides (the measure the round-2 review used).

    python tools/rust_provenance.py --table          # markdown table of the marked ranges
    python tools/rust_provenance.py --similarity     # needs /root/reference
"""
import argparse
import difflib
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "erased-cells_amd", "rust", "erased-cells-hip", "src")
REF = "/root/reference/src"
OPEN = re.compile(r"^\s*// api-surface\(([^)]*)\):\s*(.*)$")
CLOSE = re.compile(r"^\s*// end api-surface\s*$")


def marked_ranges(path):
    """[(first_line, last_line, reference cite, what)] — 1-based, markers included"""
    out, cur = [], None
    for i, line in enumerate(open(path).read().split("\n"), 1):
        m = OPEN.match(line)
        if m:
            assert cur is None, f"{path}:{i}: nested api-surface marker"
            cur = (i, m.group(1), m.group(2))
        elif CLOSE.match(line):
            assert cur is not None, f"{path}:{i}: end marker without a start"
            out.append((cur[0], i, cur[1], cur[2]))
            cur = None
    assert cur is None, f"{path}: unterminated api-surface marker"
    return out


def nontest(text):
    i = text.find("#[cfg(test)]")
    return text[:i] if i >= 0 else text


def without_marked(path):
    lines = open(path).read().split("\n")
    drop = set()
    for a, b, _, _ in marked_ranges(path):
        drop.update(range(a, b + 1))
    return "\n".join(l for i, l in enumerate(lines, 1) if i not in drop)


def line_ratio(a, b):
    return difflib.SequenceMatcher(None, a.splitlines(), b.splitlines(), autojunk=False).ratio()


def char_ratio(a, b):
    return difflib.SequenceMatcher(None, a, b, autojunk=False).ratio()


def best_match(text, refs, exhaustive):
    """(similarity, reference file): max over the reference files of max(char ratio, line ratio).  The character ratio is
    quadratic in the file sizes (minutes for the whole cross product), so unless `exhaustive` it is computed for the
    two reference files with the highest line ratio — the candidates for "the same text" — and the line ratio stands
    for the others."""
    by_lines = sorted(((line_ratio(text, rt), rn) for rn, rt in refs.items()), reverse=True)
    best = by_lines[0]
    for l, rn in (by_lines if exhaustive else by_lines[:2]):
        c = char_ratio(text, refs[rn])
        if max(c, l) > best[0]:
            best = (max(c, l), rn)
    return best


def crate_files():
    return sorted(glob.glob(os.path.join(CRATE, "*.rs")))


def table():
    rows = ["| crate file | lines | reference text it reproduces (erased-cells 0.1.1, MIT, © 2023 Astraea, Inc.) | what |", "|---|---|---|---|"]
    for p in crate_files():
        for a, b, cite, what in marked_ranges(p):
            rows.append(f"| `src/{os.path.basename(p)}` | {a}-{b} | `{cite}` | {what} |")
    return "\n".join(rows)


def similarities(exhaustive=False, with_full=True):
    refs = {os.path.relpath(p, REF): nontest(open(p).read()) for p in glob.glob(os.path.join(REF, "**", "*.rs"), recursive=True)}
    out = []
    for p in crate_files():
        full, stripped = nontest(open(p).read()), nontest(without_marked(p))
        out.append((os.path.basename(p), best_match(stripped, refs, exhaustive), best_match(full, refs, exhaustive) if with_full else (None, None)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", action="store_true")
    ap.add_argument("--similarity", action="store_true")
    ap.add_argument("--exhaustive", action="store_true", help="character ratio against every reference file (minutes)")
    a = ap.parse_args()
    if a.table or not a.similarity:
        print(table())
    if a.similarity:
        if not os.path.isdir(REF):
            sys.exit("the reference tree is not present on this box")
        print("\n| crate file | most similar reference file, marked ranges removed | with them in |\n|---|---|---|")
        for name, (s, rn), (f, fn) in similarities(a.exhaustive):
            print(f"| `src/{name}` | {s:.2f} `{rn}` | {f:.2f} `{fn}` |")


if __name__ == "__main__":
    main()
