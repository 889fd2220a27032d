"""Static instruction census of K2 (k2_mcu_lanes<4, true>: 4:2:0, byte-form coefficients) from the ISA hipcc emits, by
mnemonic class, and the per-MCU dynamic attribution it implies (checked against SQ_INSTS_VALU per wave = one MCU per lane).
usage: python profiles/tools/k2_census.py > profiles/r03_k2_census.md   (needs hipcc; no GPU)"""
import collections, json, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "jpeg-decoder_amd", "csrc")
with tempfile.TemporaryDirectory() as d:
    s = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-I" + src, "-o", s, os.path.join(src, "jd_kernels.hip")], stderr=subprocess.DEVNULL)
    lines = open(s).read().split("\n")
a = [i for i, l in enumerate(lines) if l.startswith("_Z12k2_mcu_lanesILi4ELb1ELb0EE")][0]
b = [i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
cnt = collections.Counter()
for l in lines[a:b]:
    m = re.match(r"^\s+(v_\w+)", l)
    if m:
        cnt[m.group(1)] += 1
CLASSES = [
    ("fp64 butterfly: v_add_f64, v_mul_f64", lambda k: k in ("v_add_f64", "v_mul_f64")),
    ("int <-> f64 conversions around the passes", lambda k: k.startswith("v_cvt_f64_i32") or k.startswith("v_cvt_i32_f64")),
    ("dequantise: v_mul_i32_i24 (byte operands through SDWA)", lambda k: k.startswith("v_mul_i32_i24")),
    ("colour: f32 adds (packed: the + 128; plain: the sums with the chroma products)", lambda k: k.startswith("v_pk_add_f32") or k.startswith("v_add_f32") or k.startswith("v_sub_f32")),
    ("colour: f32 -> u8 conversions (v_cvt_pk_u8_f32)", lambda k: k.startswith("v_cvt_pk_u8_f32")),
    ("colour: int -> f32 (luma samples, chroma samples)", lambda k: k.startswith("v_cvt_f32_i32")),
    ("colour: chroma products (v_mul_f32)", lambda k: k.startswith("v_mul_f32")),
    ("chroma samples: pack to int16 pairs + range watch (cvt_pk_i16, max3 / min3)", lambda k: k.startswith("v_cvt_pk_i16") or k.startswith("v_max3") or k.startswith("v_min3")),
]
print("# K2 instruction census (round 3): `k2_mcu_lanes<4, true>`, 4:2:0, byte-form coefficients\n")
print("Static VALU instructions in the kernel's ISA (one block-loop body holds: normal and wide dequantisation, 8 column passes, "
      "8 chroma row passes, 8 luma row passes with the colour tail):\n")
print("| class | static count | per block that executes it |\n|---|---|---|")
tot, named = sum(cnt.values()), 0
per = {0: "30 + 13 per pass x 24 passes in the body; 43 x 16 = 688 per block", 1: "16 per pass; 256 per block", 2: "64 per block (the wide path holds a second copy)",
       3: "12 packed + 32 plain per pixel row: 352 per luma block", 4: "24 per pixel row: 192 per luma block", 5: "8 per pixel row + 8 per chroma row pair: 96 per luma block",
       6: "16 per chroma row: 64 per luma block", 7: "4 + 8 per chroma row: 96 per chroma block"}
for i, (name, f) in enumerate(CLASSES):
    n = sum(v for k, v in cnt.items() if f(k))
    named += n
    print("| %s | %d | %s |" % (name, n, per[i]))
print("| everything else (moves, selects, shifts, address arithmetic, compares) | %d | |" % (tot - named))
print("| all VALU | %d | |\n" % tot)
import glob
sq_path = sorted(glob.glob(os.path.join(root, "profiles", "r03_?_sq_counters.json")))[-1]
sq = json.load(open(sq_path))["workloads"]["c3"]["k2_mcu_lanes"]
dyn = sq["valu_insts_per_wave"]
fp, cv, dq = 6 * 688, 6 * 256, 6 * 64
col = 4 * (352 + 192 + 96 + 64)
pk = 2 * 96
rest = dyn - fp - cv - dq - col - pk
print("Dynamic, per wave = per MCU of 6 blocks (SQ_INSTS_VALU / SQ_WAVES = %.0f, `profiles/%s`):\n" % (dyn, os.path.basename(sq_path)))
print("| section | VALU instructions | share |\n|---|---|---|")
for name, n in (("fp64 butterfly (16 one-dimensional passes per block)", fp), ("int <-> f64 conversions (F1: truncation after each pass)", cv),
                ("dequantisation multiplies", dq), ("colour tail of the 4 luma blocks (cvt, chroma products, packed adds, u8 conversions)", col),
                ("chroma sample packing + range watch", pk), ("unpacking, stores, LDS parking, addresses, loop control", rest)):
    print("| %s | %d | %.1f %% |" % (name, n, 100.0 * n / dyn))
print("\nThe butterfly and its conversions are %.1f %% of the instructions" % (100.0 * (fp + cv) / dyn), end="")
print(" (round 3, after the packed sums with broadcast chroma products became plain adds: 512 fewer v_mov per MCU, 448 more adds -- same time, see DESIGN.md section 7)", end="")
print(" and every one of them issues at >= 4.3 clk per wave "
      "(profiles/microbench/RESULTS.md); the colour tail is a quarter; what the round-2 verdict called the non-butterfly third is "
      "%.0f instructions per MCU, of which the colour tail's %d are fixed by the f32 operation order of main.go:679-703." % (dyn - fp - cv, col))
