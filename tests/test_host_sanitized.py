"""CPU: the host-side C++ that reads untrusted bytes -- jd_parse / jd_parse_scans (the twin of decodeJPEG's header work,
main.go:830-1154), jd_split_bands, the Huffman table packers, jd_bmp_encode -- compiled with AddressSanitizer and
UndefinedBehaviorSanitizer into tests/host_fuzz/host_fuzz.cpp's driver and fed the golden files plus synthetic DRI files,
as they are and under random mutations (sanitizers on the host code; the GPU side has none on this pool).
A longer run: `host_fuzz 1500 SEED files...` (≈ 390,000 mutated files over seven seeds were clean when this was written)."""
import glob
import os
import shutil
import subprocess

import pytest

import jpeg_decoder_amd as jd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "jpeg-decoder_amd", "csrc")


def test_parser_splitter_and_table_packers_under_sanitizers(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "host_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-static-libasan", "-o", exe,
           os.path.join(HERE, "host_fuzz", "host_fuzz.cpp"), os.path.join(CSRC, "jd_host.cpp"), os.path.join(CSRC, "jd_tables.cpp"),
           "-L" + os.path.dirname(jd.LIB_PATH), "-ljpeghip", "-Wl,-rpath," + os.path.dirname(jd.LIB_PATH)]   # (jd_frame_geometry lives in the library)
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and ("asan" in built.stderr.lower() or "sanitize" in built.stderr.lower()):
        pytest.skip("this g++ has no sanitizer runtime: " + built.stderr[-200:])
    assert built.returncode == 0, built.stderr[-2000:]
    files = sorted(glob.glob(os.path.join(HERE, "golden", "ref_test", "**", "*.jpg"), recursive=True)) + \
        sorted(glob.glob(os.path.join(HERE, "golden", "prog_twins", "*.jpg")))
    assert len(files) >= 13
    S = jd.SynthImage
    for k, (w, h, hs, vs, nc, ri, fl) in enumerate([(64, 48, 2, 2, 3, 4, 0), (200, 120, 2, 1, 3, 25, 0), (333, 129, 1, 1, 3, 42, S.SHARED_TABLES),
                                                    (96, 64, 1, 1, 1, 3, 0), (160, 96, 1, 2, 3, 20, S.DQT_16BIT), (64, 64, 2, 2, 3, 0, S.ZERO_BASED_IDS)]):
        path = tmp_path / ("synth%d.jpg" % k)
        path.write_bytes(S(w, h, hs, vs, nc, quality=85, seed=k, detail=2).encode(ri, fl))
        files.append(str(path))
    for k in range(4):                                          # ... and files with random valid Huffman tables (codes of 1..16 bits)
        path = tmp_path / ("random_tables%d.jpg" % k)
        path.write_bytes(S(72 + 8 * k, 40 + k, 2, 2, 3, quality=85, seed=k, detail=2).encode(4 * (k % 2), S.RANDOM_TABLES | (k << 8)))
        files.append(str(path))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe, "60", "7"] + files, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "bands cut" in run.stdout and " 0 bands cut" not in run.stdout, run.stdout
