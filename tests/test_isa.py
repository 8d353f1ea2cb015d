"""Properties of the generated gfx950 ISA that the kernels rely on (hipcc cross-compiles here; no GPU)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_stream4_kernel_keeps_agprs_to_its_asm_statements():
    """stream4_kernel holds its weight ring, operands and accumulators in AGPRs under fixed names
    ACROSS asm statements (csrc/seg_asm.inc): the C++ around them must never be compiled into
    anything that touches an AGPR, and must not spill.  Also: the committed seg_asm.inc is what
    tools/gen_seg_asm.py generates."""
    gen = subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_seg_asm.py")], capture_output=True, text=True, check=True)
    assert gen.stdout == open(os.path.join(ROOT, "deeprecsys_amd", "csrc", "seg_asm.inc")).read()
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "deeprecsys_amd", "csrc"), "check-agpr"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    # six instantiations (16 / 32 rows, summed input, two per CU, and the two column-split forms), one of them two per CU
    # (the report appears twice when make had to rebuild the listing first)
    assert r.stdout.count("AGPR uses outside asm: 0") in (6, 12) and "occupancy 2" in r.stdout
    assert r.stdout.count("scratch 0,") == r.stdout.count("AGPR uses outside asm: 0")


def test_generated_segment_streams_wait_for_exactly_what_they_consume():
    """tools/check_seg_asm.py interprets every generated statement of stream4_kernel (1 / 2 / 4 tiles,
    16 / 32 rows, segments of 1..7 chunks, either entry slot) with in-order retirement of loads and LDS
    reads: every MFMA must find operands that have retired and hold the data of its place in the
    k-ordered chain, and the next segment's first chunk must be requested into the slot left free."""
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "check_seg_asm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "84 cases" in r.stdout
