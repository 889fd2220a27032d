"""Host logic: the Huffman table packers (jpeg-decoder_amd/csrc/jd_tables.cpp) against the canonical code assignment of
the reference (generateCodes, main.go:1156-1169) -- every code of the Annex K tables through the u16 tables (10/11/12
primary bits), the writer's 32-bit tables (value arithmetic included) and the sync pass's multi-symbol tables."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "jpeg-decoder_amd", "csrc")


def test_table_packers(tmp_path):
    exe = str(tmp_path / "test_tables")
    subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + CSRC,
                    "-I" + os.path.join(HERE, "..", "include"), os.path.join(HERE, "native", "test_tables.cpp"),
                    os.path.join(CSRC, "jd_tables.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "ok (0 failures)" in r.stdout
