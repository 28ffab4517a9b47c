"""Static check of the compiled GEMM kernels (no GPU needed: hipcc cross-compiles gfx950): the pipelined kernels read
LDS fragments through inline assembly, which the compiler treats as complete when issued.  An instruction that touches
such a read's destination registers before the next wait covering LDS reads stale data or, if it writes them, is
overwritten when the data lands -- that is how a refill descriptor once lost its byte count and a launch faulted."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_no_instruction_touches_an_inflight_fragment_register():
    # (the default `make` target runs the same check before it links the library; this asks for its report)
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "image_caption_amd", "csrc"), "-j4", "check-asm"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "gemm_bf16_pipe_kernel" in r.stdout          # the kernels with inline-assembly reads were found and walked
    assert "ppipe_kernel" in r.stdout and "gemm_dev.s: no scratch instruction inside an MFMA loop" in r.stdout


def _run_checker(tmp_path, body):
    import sys
    src = tmp_path / "k.s"
    src.write_text("_Z6kernelv:\n" + body + "\ts_endpgm\n")
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_loads.py"), str(src)], capture_output=True, text=True)


def test_checker_finds_a_reused_register_and_follows_branches(tmp_path):
    """The checker itself: a register written while an inline-assembly read into it is in flight is reported, also when
    the write sits behind a branch; a path on which the wait has happened is not."""
    read = "\t;;#ASMSTART\n\tds_read_b64_tr_b16 v[10:11], v12\n\t;;#ASMEND\n"
    r = _run_checker(tmp_path, read + "\tv_mov_b32_e32 v11, v3\n\ts_waitcnt lgkmcnt(0)\n")
    assert r.returncode == 1 and "1 instruction(s)" in r.stdout
    r = _run_checker(tmp_path, read + "\ts_cbranch_scc1 .LBB0_2\n\ts_waitcnt lgkmcnt(0)\n\ts_branch .LBB0_3\n.LBB0_2:\n\tv_add_u32_e32 v10, v1, v2\n"
                     ".LBB0_3:\n\ts_waitcnt lgkmcnt(0)\n")
    assert r.returncode == 1 and "1 instruction(s)" in r.stdout
    r = _run_checker(tmp_path, read + "\ts_waitcnt lgkmcnt(0)\n\ts_cbranch_scc1 .LBB0_2\n\ts_branch .LBB0_3\n.LBB0_2:\n\tv_mov_b32_e32 v11, v3\n.LBB0_3:\n")
    assert r.returncode == 0 and ": 0 instruction(s)" in r.stdout
    # a 16-byte read between the assembly markers counts, the compiler's own (which it waits for itself) does not
    r = _run_checker(tmp_path, "\t;;#ASMSTART\n\tds_read_b128 v[4:7], v12\n\t;;#ASMEND\n\tv_mov_b32_e32 v5, v3\n\ts_waitcnt lgkmcnt(0)\n"
                     + read + "\ts_waitcnt lgkmcnt(0)\n")
    assert r.returncode == 1
    r = _run_checker(tmp_path, "\tds_read_b128 v[4:7], v12\n\tv_mov_b32_e32 v5, v3\n" + read + "\ts_waitcnt lgkmcnt(0)\n")
    assert r.returncode == 0


def test_mfma_loop_checker_finds_a_spill_inside_the_loop(tmp_path):
    """tools/check_mfma_loops.py: a scratch instruction inside a loop of MFMAs is reported, one outside is not."""
    import sys
    mf = "".join("\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]\n" for _ in range(16))
    tool = os.path.join(ROOT, "tools", "check_mfma_loops.py")
    for spill_inside, want in ((True, 1), (False, 0)):
        body = "_Z6kernelv:\n\tscratch_store_dword off, v1, off\n.LBB0_1:\n" + mf + ("\tscratch_load_dword v1, off, off\n" if spill_inside else "") + \
               "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n"
        src = tmp_path / f"k{want}.s"
        src.write_text(body)
        r = subprocess.run([sys.executable, tool, str(src)], capture_output=True, text=True)
        assert r.returncode == want, r.stdout
