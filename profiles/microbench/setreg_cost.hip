// what does an s_setreg pair (rounding-mode window) cost around a group of conversions?  two waves per SIMD as in K2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4000
template <int WIN> __global__ void __launch_bounds__(512) k(uint64_t* out, float seed) {
    float f[8]; uint32_t d[8]; double a[8];
    for (int i = 0; i < 8; i++) { f[i] = seed + i + threadIdx.x; d[i] = 0; a[i] = f[i]; }
    const double kk = seed;
    __syncthreads();
    const uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        // 32 f64 adds (stand-in for the row pass), then the window
        for (int r = 0; r < 4; r++)
            asm volatile("v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\n"
                         "v_add_f64 %4, %4, %8\nv_add_f64 %5, %5, %8\nv_add_f64 %6, %6, %8\nv_add_f64 %7, %7, %8\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(kk));
        if (WIN) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3");
        for (int r = 0; r < 3; r++)
            asm volatile("v_cvt_pk_u8_f32 %0, %8, 0, %0\nv_cvt_pk_u8_f32 %1, %9, 1, %1\nv_cvt_pk_u8_f32 %2, %10, 2, %2\nv_cvt_pk_u8_f32 %3, %11, 3, %3\n"
                         "v_cvt_pk_u8_f32 %4, %12, 0, %4\nv_cvt_pk_u8_f32 %5, %13, 1, %5\nv_cvt_pk_u8_f32 %6, %14, 2, %6\nv_cvt_pk_u8_f32 %7, %15, 3, %7\n"
                         : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])
                         : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]));
        if (WIN) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
    }
    const uint64_t t1 = clock64();
    double s = 0; uint32_t si = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; si += d[i]; }
    if (s == 12345.678 && si == 77) out[1000] = 1;
    __shared__ unsigned long long mx;
    if (threadIdx.x == 0) mx = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicMax(&mx, (unsigned long long)(t1 - t0));
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = mx;
}
int main() {
    uint64_t* d; hipMalloc(&d, 16384); uint64_t h;
    for (int rep = 0; rep < 2; rep++) {
        k<0><<<256, 512>>>(d, 1.25f); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        const double base = (double)h / ITERS;
        k<1><<<256, 512>>>(d, 1.25f); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        const double win = (double)h / ITERS;
        printf("per iteration (56 VALU, 2 waves/SIMD): %.1f clk without the window, %.1f clk with it: %.1f clk per s_setreg pair per wave pair\n", base, win, win - base);
    }
    return 0;
}
