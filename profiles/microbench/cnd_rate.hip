// v_cndmask_b32 issue rate: mask in vcc set once / set by a v_cmp right before / mask in an SGPR pair
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4000
template <int OP> __global__ void __launch_bounds__(1024) k(uint64_t* out, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i * 77 + threadIdx.x;
    uint32_t k1 = seed | 3;
    __shared__ unsigned long long mx;
    if (threadIdx.x == 0) mx = 0;
    __syncthreads();
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a[0]), "v"(k1) : "vcc");
    const uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cndmask_b32_e32 %3, %3, %8, vcc\n"
                                  "v_cndmask_b32_e32 %4, %4, %8, vcc\nv_cndmask_b32_e32 %5, %5, %8, vcc\nv_cndmask_b32_e32 %6, %6, %8, vcc\nv_cndmask_b32_e32 %7, %7, %8, vcc\n"
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k1) : "vcc");
        if (OP == 1) asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cmp_lt_u32 vcc, %2, %8\nv_cndmask_b32_e32 %3, %3, %8, vcc\n"
                                  "v_cmp_lt_u32 vcc, %4, %8\nv_cndmask_b32_e32 %5, %5, %8, vcc\nv_cmp_lt_u32 vcc, %6, %8\nv_cndmask_b32_e32 %7, %7, %8, vcc\n"
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k1) : "vcc");
        if (OP == 2) asm volatile("v_cmp_lt_u32 vcc, %0, %8\nv_cndmask_b32_e32 %1, %1, %8, vcc\nv_cndmask_b32_e32 %2, %2, %8, vcc\nv_cndmask_b32_e32 %3, %3, %8, vcc\n"
                                  "v_cndmask_b32_e32 %4, %4, %8, vcc\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k1) : "vcc");
        if (OP == 3) asm volatile("v_cmp_lt_u32 s[20:21], %0, %8\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\nv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                                  "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                                  : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k1) : "s20", "s21");
    }
    const uint64_t t1 = clock64();
    uint32_t s = 0; for (int i = 0; i < 8; i++) s += a[i];
    if (s == 0x12345u) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) atomicMax(&mx, (unsigned long long)(t1 - t0));
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = mx;
}
template <int OP> void run(const char* name, uint64_t* d) {
    printf("%-44s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        k<OP><<<256, wps * 256>>>(d, 12345u);
        uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("  %d w/SIMD: %.2f", wps, (double)h / (ITERS * 8.0 * wps));
    }
    printf("   clk per wave-instruction (8 per group)\n");
}
int main() {
    uint64_t* d; hipMalloc(&d, 16384);
    run<0>("8 x cndmask (vcc set once)", d);
    run<1>("4 x (cmp vcc; cndmask)", d);
    run<2>("cmp vcc; 4 x cndmask; 3 x add", d);
    run<3>("cmp sgpr; 4 x cndmask_e64; 3 x add", d);
    return 0;
}
