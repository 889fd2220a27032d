// cycles per wave-instruction of the fp64 / conversion ops K2 is made of (one or two waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4000
#define OP8_D(ins) \
    asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" \
                 ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n" \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k))
template <int OP> __global__ void __launch_bounds__(512) rate(uint64_t* out, double seed) {
    double a[8]; int ia[8]; float fa[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; ia[i] = (int)a[i]; fa[i] = (float)a[i]; }
    const double k = seed * 0.5 + 1.0; const float kf = (float)k;
    __syncthreads();
    const uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) OP8_D("v_add_f64");
        if (OP == 1) OP8_D("v_mul_f64");
        if (OP == 2) asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\n"
                                  "v_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k));
        if (OP == 3) asm volatile("v_cvt_f64_i32 %0, %8\nv_cvt_f64_i32 %1, %9\nv_cvt_f64_i32 %2, %10\nv_cvt_f64_i32 %3, %11\n"
                                  "v_cvt_f64_i32 %4, %12\nv_cvt_f64_i32 %5, %13\nv_cvt_f64_i32 %6, %14\nv_cvt_f64_i32 %7, %15\n"
                     : "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7])
                     : "v"(ia[0]), "v"(ia[1]), "v"(ia[2]), "v"(ia[3]), "v"(ia[4]), "v"(ia[5]), "v"(ia[6]), "v"(ia[7]));
        if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %8\nv_cvt_i32_f64 %1, %9\nv_cvt_i32_f64 %2, %10\nv_cvt_i32_f64 %3, %11\n"
                                  "v_cvt_i32_f64 %4, %12\nv_cvt_i32_f64 %5, %13\nv_cvt_i32_f64 %6, %14\nv_cvt_i32_f64 %7, %15\n"
                     : "=v"(ia[0]), "=v"(ia[1]), "=v"(ia[2]), "=v"(ia[3]), "=v"(ia[4]), "=v"(ia[5]), "=v"(ia[6]), "=v"(ia[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
        if (OP == 5) asm volatile("v_trunc_f64 %0, %0\nv_trunc_f64 %1, %1\nv_trunc_f64 %2, %2\nv_trunc_f64 %3, %3\n"
                                  "v_trunc_f64 %4, %4\nv_trunc_f64 %5, %5\nv_trunc_f64 %6, %6\nv_trunc_f64 %7, %7\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        if (OP == 6) asm volatile("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\n"
                                  "v_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_add_f32 %7, %7, %8\n"
                     : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]) : "v"(kf));
        if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n"
                                  "v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(k));
        if (OP == 8) asm volatile("v_cvt_pk_u8_f32 %0, %8, 0, %0\nv_cvt_pk_u8_f32 %1, %9, 1, %1\nv_cvt_pk_u8_f32 %2, %10, 2, %2\nv_cvt_pk_u8_f32 %3, %11, 3, %3\n"
                                  "v_cvt_pk_u8_f32 %4, %12, 0, %4\nv_cvt_pk_u8_f32 %5, %13, 1, %5\nv_cvt_pk_u8_f32 %6, %14, 2, %6\nv_cvt_pk_u8_f32 %7, %15, 3, %7\n"
                     : "+v"(ia[0]), "+v"(ia[1]), "+v"(ia[2]), "+v"(ia[3]), "+v"(ia[4]), "+v"(ia[5]), "+v"(ia[6]), "+v"(ia[7])
                     : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "v"(fa[4]), "v"(fa[5]), "v"(fa[6]), "v"(fa[7]));
        if (OP == 9) asm volatile("v_mul_i32_i24 %0, %0, %8\nv_mul_i32_i24 %1, %1, %8\nv_mul_i32_i24 %2, %2, %8\nv_mul_i32_i24 %3, %3, %8\n"
                                  "v_mul_i32_i24 %4, %4, %8\nv_mul_i32_i24 %5, %5, %8\nv_mul_i32_i24 %6, %6, %8\nv_mul_i32_i24 %7, %7, %8\n"
                     : "+v"(ia[0]), "+v"(ia[1]), "+v"(ia[2]), "+v"(ia[3]), "+v"(ia[4]), "+v"(ia[5]), "+v"(ia[6]), "+v"(ia[7]) : "v"(ia[0] | 3));
        if (OP == 10) asm volatile("v_cvt_f32_i32 %0, %8\nv_cvt_f32_i32 %1, %9\nv_cvt_f32_i32 %2, %10\nv_cvt_f32_i32 %3, %11\n"
                                  "v_cvt_f32_i32 %4, %12\nv_cvt_f32_i32 %5, %13\nv_cvt_f32_i32 %6, %14\nv_cvt_f32_i32 %7, %15\n"
                     : "=v"(fa[0]), "=v"(fa[1]), "=v"(fa[2]), "=v"(fa[3]), "=v"(fa[4]), "=v"(fa[5]), "=v"(fa[6]), "=v"(fa[7])
                     : "v"(ia[0]), "v"(ia[1]), "v"(ia[2]), "v"(ia[3]), "v"(ia[4]), "v"(ia[5]), "v"(ia[6]), "v"(ia[7]));
    }
    const uint64_t t1 = clock64();
    double s = 0; int si = 0; float sf = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; si += ia[i]; sf += fa[i]; }
    if (s == 12345.678 && si == 77 && sf == 3.f) out[1000] = 1;       // keep the values alive
    __shared__ unsigned long long mx;
    if (threadIdx.x == 0) mx = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) atomicMax(&mx, (unsigned long long)(t1 - t0));
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = mx;
}
template <int OP> void run(const char* name, uint64_t* d) {
    for (int waves = 4; waves <= 8; waves += 4) {          // one / two waves per SIMD, one workgroup per CU
        hipMemset(d, 0, 8192);
        rate<OP><<<256, waves * 64>>>(d, 1.25);
        uint64_t h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("%-16s %d waves/SIMD: %.2f clk per wave-instruction issued by the SIMD (longest wave of the workgroup)\n", name, waves / 4,
               (double)h[0] / (ITERS * 8.0) / (waves / 4));
    }
}
int main() {
    uint64_t* d; hipMalloc(&d, 16384);
    int clk = 0, wall = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    printf("clock %d kHz, wall clock %d kHz\n", clk, wall);
    run<0>("v_add_f64", d); run<1>("v_mul_f64", d); run<2>("v_fma_f64", d); run<3>("v_cvt_f64_i32", d);
    run<4>("v_cvt_i32_f64", d); run<5>("v_trunc_f64", d); run<6>("v_add_f32", d); run<7>("v_pk_add_f32", d);
    run<8>("v_cvt_pk_u8_f32", d); run<9>("v_mul_i32_i24", d); run<10>("v_cvt_f32_i32", d);
    // s_memtime may tick at a fixed rate: calibrate with a timed launch
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); rate<6><<<256, 256>>>(d, 1.25); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("calibration: v_add_f32 launch %.3f ms wall, %llu ticks => %.1f MHz tick rate (upper bound: launch overhead included)\n",
           ms, (unsigned long long)h, h / (ms * 1e3));
    return 0;
}
