// does ds_read_b128 / ds_read_b64 honour byte addresses that are not multiples of 16 / 8 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint8_t* out, int sh) {
    __shared__ __align__(16) uint8_t buf[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = (uint8_t)(i * 7 + 1);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)buf + 16u * threadIdx.x + (uint32_t)sh;
    u32x4 r; u32x2 r2;
    asm volatile("ds_read_b128 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(r2) : "v"(addr) : "memory");
    ((u32x4*)out)[threadIdx.x] = r;
    ((u32x2*)(out + 1024))[threadIdx.x] = r2;
}
int main() {
    uint8_t* d; hipMalloc(&d, 2048);
    for (int sh = 0; sh < 16; sh++) {
        probe<<<1, 64>>>(d, sh);
        uint8_t h[2048]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
        int bad128 = 0, bad64 = 0;
        for (int L = 0; L < 64; L++) {
            for (int j = 0; j < 16; j++) if (h[16 * L + j] != (uint8_t)((16 * L + sh + j) * 7 + 1)) bad128++;
            for (int j = 0; j < 8; j++) if (h[1024 + 8 * L + j] != (uint8_t)((16 * L + sh + j) * 7 + 1)) bad64++;
        }
        printf("shift %2d: b128 %s (%d)  b64 %s (%d)\n", sh, bad128 ? "WRONG" : "exact", bad128, bad64 ? "WRONG" : "exact", bad64);
    }
    return 0;
}
