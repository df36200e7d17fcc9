"""The hand-scheduled kernels issue their MFMAs as inline asm, which hides two things from the compiler:
that an MFMA writes its destination dozens of cycles after issue, and that the fp8 attention kernel's
fragment ring (named accumulator registers) carries data between statements.  tools/check_mfma_hazards.py
reads the compiler's assembly for instructions of its own making that run into either; here it is
checked on synthetic listings and then run over the real kernels (`make asmcheck`, about a minute)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_mfma_hazards as chk  # noqa: E402

LISTING = """
\t.text
_Z6kernelv:
\t;;#ASMSTART
\tv_mfma_f32_16x16x32_bf16 v[12:15], v[74:77], a[44:47], v[12:15]
\t;;#ASMEND
{between}
\t;;#ASMSTART
\tv_mfma_f32_16x16x32_bf16 v[16:19], v[74:77], a[44:47], v[16:19]
\t;;#ASMEND
{after}
\ts_endpgm
"""


def _run(tmp_path, between="", after="", ring_base=None):
    p = tmp_path / "k.s"
    p.write_text(LISTING.format(between=between, after=after))
    return chk.check(str(p), "", ring_base)


def test_checker_flags_a_copy_right_behind_an_asm_mfma(tmp_path):
    assert _run(tmp_path) == 0
    assert _run(tmp_path, between="\tv_mov_b64_e32 v[92:93], v[14:15]") == 1      # reads the destination too early
    assert _run(tmp_path, between="\tv_mov_b32_e32 v13, 0") == 1                   # writes it
    assert _run(tmp_path, between="\tv_mov_b32_e32 v20, v21") == 0                 # other registers
    assert _run(tmp_path, after="\tv_mul_f32_e32 v3, v14, v36") == 0               # one MFMA later: a note only
    # the same instruction inside an asm block is the kernel author's own
    assert _run(tmp_path, between="\t;;#ASMSTART\n\tv_mov_b32_e32 v20, v14\n\t;;#ASMEND") == 0


def test_checker_flags_compiler_traffic_through_the_ring(tmp_path):
    assert _run(tmp_path, between="\tv_accvgpr_write_b32 a248, v11", ring_base=224) == 1
    assert _run(tmp_path, between="\tv_accvgpr_write_b32 a200, v11", ring_base=224) == 0
    assert _run(tmp_path, between="\t;;#ASMSTART\n\tds_read_b128 a[248:251], v1\n\t;;#ASMEND", ring_base=224) == 0
    # the exp-free fp8 instances carry scale dwords by name in a216-a218 as well: checked with the lower base
    assert _run(tmp_path, between="\tv_accvgpr_read_b32 v9, a217", ring_base=224) == 0
    assert _run(tmp_path, between="\tv_accvgpr_read_b32 v9, a217", ring_base=216) == 1
    p = tmp_path / "k.s"
    p.write_text(LISTING.format(between="\tv_accvgpr_read_b32 v9, a217", after=""))
    assert chk.check(str(p), "no_such_kernel", 216) == 0      # --kernel restricts the check to matching symbols
    assert chk.check(str(p), "kernel", 216) == 1


def test_checker_flags_an_lds_read_destination_touched_before_a_wait(tmp_path):
    read = "\t;;#ASMSTART\n\tds_read_b32 v11, v10 offset:0x6600\n\t;;#ASMEND\n"
    wait = "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(4)\n\t;;#ASMEND\n"
    assert _run(tmp_path, between=read + "\tv_accvgpr_write_b32 a221, v11") == 1            # parked before the data has landed
    assert _run(tmp_path, between=read + wait + "\tv_accvgpr_write_b32 a221, v11") == 0     # behind a wait
    assert _run(tmp_path, between=read + "\tv_mov_b32_e32 v20, v21") == 0                   # other registers


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_kernels_with_inline_asm_mfmas_are_clean():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "lightx2v_amd", "csrc"), "-j4", "asmcheck"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
