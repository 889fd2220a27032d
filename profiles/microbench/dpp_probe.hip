// semantics of the DPP controls / dot / perm the K0 rewrite relies on (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t dpp_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
__global__ void probe(uint32_t* out) {
    const uint32_t lane = threadIdx.x, v = lane * 7u + 3u;
    out[lane] = (uint32_t)__builtin_amdgcn_update_dpp(1000, (int)v, 0x138, 0xF, 0xF, false);        // wave_shr:1
    out[64 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(2000, (int)v, 0x130, 0xF, 0xF, false);   // wave_shl:1
    out[128 + lane] = dpp_scan(v);
    const uint32_t f = (lane & 1 ? 0x80u : 0u) | (lane & 2 ? 0x8000u : 0u) | (lane & 4 ? 0x800000u : 0u) | (lane & 8 ? 0x80000000u : 0u);
    out[192 + lane] = __builtin_amdgcn_udot4(f, 0x08040201u, 5u, false);
    out[256 + lane] = __builtin_amdgcn_perm(0xA7A6A5A4u, 0xA3A2A1A0u, 0x0C050300u + (lane & 3));
}
int main() {
    uint32_t* d; hipMalloc(&d, 4096); probe<<<1, 64>>>(d);
    uint32_t h[320]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0; uint32_t run = 0;
    for (uint32_t l = 0; l < 64; l++) {
        const uint32_t v = l * 7 + 3; run += v;
        if (h[l] != (l ? (l - 1) * 7 + 3 : 1000u)) { bad++; printf("wave_shr lane %u: %u\n", l, h[l]); }
        if (h[64 + l] != (l < 63 ? (l + 1) * 7 + 3 : 2000u)) { bad++; printf("wave_shl lane %u: %u\n", l, h[64 + l]); }
        if (h[128 + l] != run) { bad++; printf("scan lane %u: %u want %u\n", l, h[128 + l], run); }
        if (h[192 + l] != 5u + 128u * (l & 15)) { bad++; printf("udot4 lane %u: %u\n", l, h[192 + l]); }
    }
    printf("perm(S0=A7..A4, S1=A3..A0, sel 0C 05 03 0x): %08x %08x %08x %08x\n", h[256], h[257], h[258], h[259]);
    printf("%s (%d)\n", bad ? "MISMATCH" : "dpp/udot4 ok", bad);
    return 0;
}
