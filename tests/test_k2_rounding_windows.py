"""k2_mcu_lanes converts a pixel row's 24 floats to bytes between two s_setreg that switch the f32 rounding mode to
toward-zero and back (jd_kernels.hip, k2_rtz_on / k2_rtz_off).  The compiler does not know about the mode register, so
the source fences the arithmetic out of those windows; this test reads the generated gfx950 assembly and checks that it
worked: inside a window there is no floating-point instruction other than v_cvt_pk_u8_f32, no branch and no branch
target, and every window is closed before the next one opens.  (CPU only: hipcc cross-compiles.)"""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "jpeg-decoder_amd", "csrc", "jd_kernels.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_only_conversions_inside_rtz_windows(tmp_path):
    out = tmp_path / "jd_kernels.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-S", "--cuda-device-only",
                    "-o", str(out), SRC], check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = [f for f in re.split(r"\n(?=_Z\w+:)", text) if f.startswith("_Z12k2_mcu_lanes")]
    assert len(kernels) == 15                                         # one per sampling mode and coefficient form (int16, bytes, bytes in staging)
    windows = 0
    for k in kernels:
        inside = False
        for line in k.split("\n"):
            if re.match(r"^\.?LBB\w*:", line):
                assert not inside, "branch target inside a rounding-mode window: " + line
                continue
            m = re.match(r"\s+([a-z_0-9]+)(\s|$)", line)
            if not m:
                continue
            op = m.group(1)
            if op == "s_setreg_imm32_b32":
                assert "HW_REG_MODE, 0, 2" in line
                on = line.rstrip().endswith(", 3")
                assert on != inside, "rounding-mode writes do not alternate: " + line
                inside = on
                windows += on
                continue
            if op == "s_endpgm":
                assert not inside
            if inside:
                assert not op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call")), line
                if re.search(r"f16|f32|f64", op):
                    assert op == "v_cvt_pk_u8_f32", "floating-point instruction inside a rounding-mode window: " + line
        assert not inside
    assert windows >= 10 * 8                                           # eight pixel rows per luma block at least
