// Microbenchmark: issue rate of the integer VALU / LDS instructions the assign kernel is built from (gfx950).
// Each kernel runs ITER iterations of 16 independent copies of one instruction; cycles are taken with
// s_memtime inside one wave and with HIP events over a chip-filling grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 512
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
    __shared__ uint32_t s_tab[4096];
    uint32_t v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed * (threadIdx.x + 1) + i * 77;
    for (int i = threadIdx.x; i < 4096; i += 256) s_tab[i] = i * seed;
    __syncthreads();
    const uint32_t s1 = seed | 0x01010101u, s2 = seed + 3;
    uint32_t addr = (threadIdx.x * 4) & 0x3ffc;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(v[i]) : "s"(s1), "v"(v[(i + 1) & 15]));
            if (OP == 1) asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 2) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 6, %1" : "+v"(v[i]) : "s"(s2));
            if (OP == 4) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 5) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "s"(s2));
            if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "s"(s2), "v"(v[(i + 1) & 15]));
            if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
            if (OP == 9) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(v[i]) : "s"(s1), "v"(v[(i + 1) & 15]));
            if (OP == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "s"(s1));
            if (OP == 11) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
            if (OP == 12) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
            if (OP == 13) asm volatile("v_sad_hi_u8 %0, %0, %1, %2" : "+v"(v[i]) : "s"(s1), "v"(v[(i + 1) & 15]));
            if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 15) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 16) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 17) asm volatile("v_pk_lshlrev_b16 %0, 6, %0" : "+v"(v[i]));
            if (OP == 18) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 19) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            if (OP == 20) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(v[i]), "v"(v[(i + 1) & 15]) : "vcc");
            if (OP == 21) asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(v[i]) : "s20");
            if (OP == 22) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
            // LDS: independent reads, one wait per 16
            if (OP == 30) asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(i * 64));
            if (OP == 31) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"(addr), "n"(i * 64));
            if (OP == 32) asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "+v"(v[i]) : "v"(addr), "n"(i * 64));
        }
        if (OP >= 30) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (OP == 33) {   // 4 x ds_read_b128 per iteration (counts as 4)
            uint4 a, b, c, d;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr & 0x3ff0) : "memory");
            v[0] += a.x + b.y + c.z + d.w;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
    for (int i = 0; i < 16; ++i) acc ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Op { int id; const char* name; int per_iter; };

template <int OP>
static int run(const char* name, int per_iter, uint32_t* d_out, unsigned long long* d_cyc) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int blocks_per_cu : {1, 2, 4, 8}) {          // 1, 2, 4, 8 waves per SIMD
        const int grid = 256 * blocks_per_cu;
        hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 12345u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 12345u);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid);
        CHK(hipMemcpy(h.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost));
        double avg = 0;
        for (auto c : h) avg += (double)c;
        avg /= grid;
        const double n = (double)ITER * per_iter;
        // per-SIMD issue cost: waves/SIMD * instrs per wave / cycles
        printf("%-22s waves/SIMD=%d  ticks/instr/wave=%7.2f  ticks/instr/SIMD=%6.2f  kernel=%8.2f us  (%.1f Ginstr/s chip)\n",
               name, blocks_per_cu, avg / n, avg / n / blocks_per_cu, ms * 1e3,
               n * grid * 4 / (ms * 1e-3) / 1e9);
    }
    return 0;
}

int main() {
    uint32_t* d_out; unsigned long long* d_cyc;
    CHK(hipMalloc(&d_out, 256 * 8 * 256 * 4)); CHK(hipMalloc(&d_cyc, 256 * 8 * 8));
    run<8>("v_fma_f32", 16, d_out, d_cyc);
    run<7>("v_add_u32", 16, d_out, d_cyc);
    run<0>("v_sad_u8", 16, d_out, d_cyc);
    run<13>("v_sad_hi_u8", 16, d_out, d_cyc);
    run<9>("v_sad_u16", 16, d_out, d_cyc);
    run<1>("v_pk_add_u16 clamp", 16, d_out, d_cyc);
    run<15>("v_pk_sub_u16 clamp", 16, d_out, d_cyc);
    run<2>("v_pk_min_u16", 16, d_out, d_cyc);
    run<16>("v_pk_max_u16", 16, d_out, d_cyc);
    run<12>("v_pk_mad_u16", 16, d_out, d_cyc);
    run<17>("v_pk_lshlrev_b16", 16, d_out, d_cyc);
    run<3>("v_lshl_add_u32", 16, d_out, d_cyc);
    run<4>("v_min_u32", 16, d_out, d_cyc);
    run<11>("v_min3_u32", 16, d_out, d_cyc);
    run<5>("v_and_or_b32", 16, d_out, d_cyc);
    run<6>("v_mad_u32_u24", 16, d_out, d_cyc);
    run<18>("v_mul_u32_u24", 16, d_out, d_cyc);
    run<19>("v_mul_lo_u32", 16, d_out, d_cyc);
    run<10>("v_perm_b32", 16, d_out, d_cyc);
    run<14>("v_cndmask_b32", 16, d_out, d_cyc);
    run<20>("v_cmp_lt_u32", 16, d_out, d_cyc);
    run<21>("v_readfirstlane_b32", 16, d_out, d_cyc);
    run<22>("v_mov_b32_dpp", 16, d_out, d_cyc);
    run<30>("ds_read_u16", 16, d_out, d_cyc);
    run<32>("ds_read_u16_d16_hi", 16, d_out, d_cyc);
    run<31>("ds_read_b32", 16, d_out, d_cyc);
    run<33>("ds_read_b128", 4, d_out, d_cyc);
    return 0;
}
