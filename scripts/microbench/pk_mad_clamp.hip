// Semantics check: does the clamp bit of v_pk_mad_u16 saturate the full-width result a*b+c (both halves)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* a, const uint32_t* c, uint32_t* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    uint32_t r;
    const uint32_t cc = __builtin_amdgcn_readfirstlane(c[0]);
    asm volatile("v_pk_mad_u16 %0, %1, 64, %2 op_sel_hi:[1,0,1] clamp" : "=v"(r) : "v"(a[i]), "s"(cc));
    out[i] = r;
}
int main() {
    const uint32_t ha[8] = {0x00010002u, 0x03FE03FFu, 0x04000400u, 0x40004000u, 0x42FD0005u, 0xFFFF0000u, 0x08000401u, 0x000003FFu};
    for (uint32_t hc : {0x00050005u, 0xFFFF0003u, 0x003FFFFFu}) {
        uint32_t *a, *c, *o, ho[8];
        hipMalloc(&a, 32); hipMalloc(&c, 4); hipMalloc(&o, 32);
        hipMemcpy(a, ha, 32, hipMemcpyHostToDevice); hipMemcpy(c, &hc, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, c, o, 8);
        hipMemcpy(ho, o, 32, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 8; i++) {
            auto f = [&](uint32_t x, uint32_t y) { uint32_t v = x * 64u + y; return v > 0xFFFFu ? 0xFFFFu : v; };
            const uint32_t want = f(ha[i] & 0xFFFF, hc & 0xFFFF) | (f(ha[i] >> 16, hc >> 16) << 16);
            printf("a=%08x c=%08x -> %08x want %08x %s\n", ha[i], hc, ho[i], want, ho[i] == want ? "ok" : "MISMATCH");
            bad += ho[i] != want;
        }
        printf("mismatches: %d\n", bad);
    }
    return 0;
}
