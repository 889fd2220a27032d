// VALU issue ceiling per SIMD against waves per SIMD: independent v_add_f32 / v_add_f64 streams, longest wave of the workgroup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4000
template <int F64> __global__ void __launch_bounds__(1024) k(uint64_t* out, float seed) {
    float f[8]; double a[8];
    for (int i = 0; i < 8; i++) { f[i] = seed + i + threadIdx.x; a[i] = f[i]; }
    const double kk = seed; const float kf = seed;
    __shared__ unsigned long long mx;
    if (threadIdx.x == 0) mx = 0;
    __syncthreads();
    const uint64_t t0 = clock64();
    for (int it = 0; it < ITERS; it++) {
        if (F64) asm volatile("v_add_f64 %0, %0, %8\nv_add_f64 %1, %1, %8\nv_add_f64 %2, %2, %8\nv_add_f64 %3, %3, %8\n"
                         "v_add_f64 %4, %4, %8\nv_add_f64 %5, %5, %8\nv_add_f64 %6, %6, %8\nv_add_f64 %7, %7, %8\n"
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(kk));
        else asm volatile("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_add_f32 %7, %7, %8\n"
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(kf));
    }
    const uint64_t t1 = clock64();
    double s = 0; for (int i = 0; i < 8; i++) s += a[i] + f[i];
    if (s == 12345.678) out[1000] = 1;
    if ((threadIdx.x & 63) == 0) atomicMax(&mx, (unsigned long long)(t1 - t0));
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = mx;
}
int main() {
    uint64_t* d; hipMalloc(&d, 16384); uint64_t h;
    for (int f64 = 0; f64 < 2; f64++)
        for (int wps = 1; wps <= 4; wps++) {                    // waves per SIMD (one workgroup per CU)
            if (f64) k<1><<<256, wps * 256>>>(d, 1.25f); else k<0><<<256, wps * 256>>>(d, 1.25f);
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("%s, %d wave(s) per SIMD: %.2f clk per wave-instruction issued by the SIMD\n", f64 ? "v_add_f64" : "v_add_f32", wps,
                   (double)h / (ITERS * 8.0 * wps));
        }
    return 0;
}
