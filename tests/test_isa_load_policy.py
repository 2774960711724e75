"""The load policy the kernels are written to, checked in the machine code (no GPU needed: hipcc cross-compiles).

Round 2 shipped `nt` on every load in the SOURCE, but hipcc drops the flag from loads of <N x i8> vectors, so the u8 / i8 /
mask streams went out as plain loads; round 3's first fix (a run-time branch between an nt and a default-policy load) was
silently merged back into one plain load by the optimiser.  Neither shows in a result — only in the ISA.  This test
compiles the translation unit of the headline kernel family to gfx950 assembly and holds tools/isa_audit.py's rule:
every store nt; every load nt unless it is the default-policy twin of an nt load in another arm of the launch's policy
switch (so plain loads never outnumber nt loads of the same width in a kernel).  `python tools/isa_audit.py` audits the
other translation units the same way.
"""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_audit  # noqa: E402


@pytest.mark.skipif(not os.path.exists(isa_audit.HIPCC), reason="hipcc is not installed here")
@pytest.mark.timeout(600)
def test_binop_kernels_carry_the_load_policy(tmp_path):
    asm = str(tmp_path / "ec_binop_div.s")
    isa_audit.compile_asm("ec_binop_div.hip", asm)
    findings, audited = isa_audit.audit(asm)
    assert len(audited) >= 400 and sum(audited.values()) > 5000, "the audit did not see the kernels"
    assert not findings, "\n".join(f"{k}: {v}" for k, v in list(findings.items())[:5])
    # the headline kernel: both policies of both operand streams are present (four arms), the u8 stream included
    text = open(asm).read()
    name = "_ZN3ecd14k_binop_directIhtLi3ELi2ELb1ELb1EEEvPKT_PKT0_Pdmj"
    body = text[text.index(name + ":"):]
    body = body[:body.index("s_endpgm")]
    ushort = [l for l in body.split("\n") if "global_load_ushort" in l]
    dword = [l for l in body.split("\n") if "global_load_dword " in l]
    nt = lambda ls: sum(1 for l in ls if l.rstrip().endswith(" nt") or " nt " in l)  # noqa: E731
    assert nt(ushort) >= 4 and len(ushort) - nt(ushort) >= 4, ushort   # 2 loads x 2 arms each way
    assert nt(dword) >= 4 and len(dword) - nt(dword) >= 4, dword
    for arm in (1, 2, 3):
        assert f"; load-policy arm {arm}" in body
