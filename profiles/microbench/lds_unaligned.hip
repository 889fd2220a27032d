// does ds_write_b32 / ds_read_b32 honour byte addresses that are not multiples of 4 on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint8_t* out) {
    __shared__ __align__(16) uint8_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) buf[i] = 0xEE;
    __syncthreads();
    // lane L writes the dword 0x04030201 + L*0x10101010 at byte offset 5 * L + 1
    const uint32_t addr = (uint32_t)(uintptr_t)buf + 5u * threadIdx.x + 1u, v = 0x04030201u + threadIdx.x * 0x10101010u;
    asm volatile("ds_write_b32 %0, %1\ns_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    ((uint32_t*)(out + 512))[threadIdx.x] = r;
}
int main() {
    uint8_t* d; hipMalloc(&d, 1024); probe<<<1, 64>>>(d);
    uint8_t h[1024]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int L = 0; L < 64; L++) {
        for (int j = 0; j < 4; j++) { const uint8_t want = (uint8_t)((j + 1) + L * 0x10); if (h[5 * L + 1 + j] != want) bad++; }
        if (h[5 * L] != 0xEE) bad++;
        if (((uint32_t*)(h + 512))[L] != 0x04030201u + L * 0x10101010u) bad++;
    }
    printf("unaligned ds_write_b32/ds_read_b32: %s (%d mismatches)\n", bad ? "NOT byte-addressed" : "byte-addressed, exact", bad);
    for (int i = 0; i < 24; i++) printf("%02x ", h[i]); printf("\n");
    return 0;
}
