// issue rate of the integer / f32 instructions the Huffman kernels are made of, at 1, 2 and 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4000
#define R8(ins, tail) asm volatile(ins " %0, %0" tail "\n" ins " %1, %1" tail "\n" ins " %2, %2" tail "\n" ins " %3, %3" tail "\n" \
                                   ins " %4, %4" tail "\n" ins " %5, %5" tail "\n" ins " %6, %6" tail "\n" ins " %7, %7" tail "\n" \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k1), "v"(k2))
template <int OP> __global__ void __launch_bounds__(1024) k(uint64_t* out, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i * 77 + threadIdx.x;
    uint32_t k1 = seed | 3, k2 = seed * 5 + 1;
    __shared__ unsigned long long mx;
    if (threadIdx.x == 0) mx = 0;
    __syncthreads();
    const uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) R8("v_add_u32", ", %8");
        if (OP == 1) R8("v_and_b32", ", %8");
        if (OP == 2) R8("v_lshlrev_b32", ", %8");
        if (OP == 3) R8("v_cndmask_b32", ", %8, vcc");
        if (OP == 4) R8("v_alignbit_b32", ", %8, %9");
        if (OP == 5) R8("v_bfe_u32", ", %8, %9");
        if (OP == 6) R8("v_perm_b32", ", %8, %9");
        if (OP == 7) R8("v_mul_f32", ", %8");
        if (OP == 8) R8("v_fma_f32", ", %8, %9");
        if (OP == 9) R8("v_add3_u32", ", %8, %9");
        if (OP == 10) R8("v_lshl_add_u32", ", %8, %9");
        if (OP == 11) R8("v_and_or_b32", ", %8, %9");
        if (OP == 12) R8("v_mul_u32_u24", ", %8");
        if (OP == 13) R8("v_bcnt_u32_b32", ", %8");
        if (OP == 14) R8("v_max_u32", ", %8");
        if (OP == 15) asm volatile("v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0\n"
                                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        if (OP == 16) asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cmp_lt_u32 vcc, %1, %8\nv_cmp_lt_u32 vcc, %2, %8\nv_cmp_lt_u32 vcc, %3, %8\n"
                                   "v_cmp_lt_u32 vcc, %4, %8\nv_cmp_lt_u32 vcc, %5, %8\nv_cmp_lt_u32 vcc, %6, %8\nv_cmp_lt_u32 vcc, %7, %8\n"
                                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k1) : "vcc");
        if (OP == 17) R8("v_sub_u32", ", %8");
        if (OP == 18) R8("v_xor_b32", ", %8");
        if (OP == 19) R8("v_lshrrev_b32", ", %8");
    }
    const uint64_t t1 = clock64();
    uint32_t s = 0; for (int i = 0; i < 8; i++) s += a[i];
    if (s == 0x12345u) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) atomicMax(&mx, (unsigned long long)(t1 - t0));
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = mx;
}
template <int OP> void run(const char* name, uint64_t* d) {
    printf("%-16s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        k<OP><<<256, wps * 256>>>(d, 12345u);
        uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("  %d w/SIMD: %.2f", wps, (double)h / (ITERS * 8.0 * wps));
    }
    printf("   clk per wave-instruction\n");
}
int main() {
    uint64_t* d; hipMalloc(&d, 16384);
    run<0>("v_add_u32", d); run<17>("v_sub_u32", d); run<1>("v_and_b32", d); run<18>("v_xor_b32", d); run<2>("v_lshlrev_b32", d); run<19>("v_lshrrev_b32", d);
    run<3>("v_cndmask_b32", d); run<4>("v_alignbit_b32", d); run<5>("v_bfe_u32", d); run<6>("v_perm_b32", d); run<7>("v_mul_f32", d);
    run<8>("v_fma_f32", d); run<9>("v_add3_u32", d); run<10>("v_lshl_add_u32", d); run<11>("v_and_or_b32", d); run<12>("v_mul_u32_u24", d);
    run<13>("v_bcnt_u32_b32", d); run<14>("v_max_u32", d); run<15>("v_mov_b32", d); run<16>("v_cmp_lt_u32", d);
    return 0;
}
