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
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "image_caption_amd", "csrc"), "check-asm"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "gemm_bf16_pipe_kernel" in r.stdout          # the kernels with inline-assembly reads were found and walked
