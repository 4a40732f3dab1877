// tools/ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (design input for ff.cuh).
// Measures cycles per wave-instruction for the integer / fp64 multiply candidates a 254-bit
// Montgomery multiplier can be built from.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 256

// Each kernel runs ITERS * 64 instructions of one kind per wave on 4 independent chains.
#define DEF_KERNEL(NAME, ASM4)                                                                  \
    __global__ void NAME(uint64_t* out, uint32_t seed) {                                       \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;                                     \
        uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b;                                       \
        double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, e = 1.0000001;                              \
        uint64_t t0 = __builtin_readcyclecounter();                                            \
        for (int i = 0; i < ITERS; ++i) { REP16(ASM4) }                                        \
        uint64_t t1 = __builtin_readcyclecounter();                                            \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                        \
        if (c0 + c1 + c2 + c3 == 0x1234567 && d0 + d1 + d2 + d3 == 1.25) out[0] = 0;            \
    }

DEF_KERNEL(k_mad_u64_u32,
    asm volatile("v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n v_mad_u64_u32 %1, s[20:21], %4, %5, %1\n"
                 "v_mad_u64_u32 %2, s[20:21], %4, %5, %2\n v_mad_u64_u32 %3, s[20:21], %4, %5, %3\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "s20", "s21");)
DEF_KERNEL(k_mul_lo_u32,
    asm volatile("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %4, %1\n v_mul_lo_u32 %2, %4, %2\n v_mul_lo_u32 %3, %4, %3\n"
                 : "+v"(*(uint32_t*)&c0), "+v"(*(uint32_t*)&c1), "+v"(*(uint32_t*)&c2), "+v"(*(uint32_t*)&c3) : "v"(a));)
DEF_KERNEL(k_mul_hi_u32,
    asm volatile("v_mul_hi_u32 %0, %4, %0\n v_mul_hi_u32 %1, %4, %1\n v_mul_hi_u32 %2, %4, %2\n v_mul_hi_u32 %3, %4, %3\n"
                 : "+v"(*(uint32_t*)&c0), "+v"(*(uint32_t*)&c1), "+v"(*(uint32_t*)&c2), "+v"(*(uint32_t*)&c3) : "v"(a));)
DEF_KERNEL(k_fma_f64,
    asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e), "v"(d2));)
DEF_KERNEL(k_add_f64,
    asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));)
DEF_KERNEL(k_lshl_add_u64,
    asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(c2));)
DEF_KERNEL(k_add_u32,
    asm volatile("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3\n"
                 : "+v"(*(uint32_t*)&c0), "+v"(*(uint32_t*)&c1), "+v"(*(uint32_t*)&c2), "+v"(*(uint32_t*)&c3) : "v"(a));)
DEF_KERNEL(k_addc_u32,
    asm volatile("v_add_co_u32 %0, vcc, %4, %0\n v_addc_co_u32 %1, vcc, %4, %1, vcc\n v_addc_co_u32 %2, vcc, %4, %2, vcc\n v_addc_co_u32 %3, vcc, %4, %3, vcc\n"
                 : "+v"(*(uint32_t*)&c0), "+v"(*(uint32_t*)&c1), "+v"(*(uint32_t*)&c2), "+v"(*(uint32_t*)&c3) : "v"(a) : "vcc");)
DEF_KERNEL(k_mad_u32_u24,
    asm volatile("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3\n"
                 : "+v"(*(uint32_t*)&c0), "+v"(*(uint32_t*)&c1), "+v"(*(uint32_t*)&c2), "+v"(*(uint32_t*)&c3) : "v"(a), "v"(b));)
DEF_KERNEL(k_mov_b32,
    asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n"
                 : "+v"(*(uint32_t*)&c0), "+v"(*(uint32_t*)&c1), "+v"(*(uint32_t*)&c2), "+v"(*(uint32_t*)&c3) : "v"(a));)
DEF_KERNEL(k_fma_f32,
    asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n"
                 : "+v"(*(float*)&c0), "+v"(*(float*)&c1), "+v"(*(float*)&c2), "+v"(*(float*)&c3) : "v"(*(float*)&a));)
DEF_KERNEL(k_cvt_f64_u32,
    asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %4\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %4\n"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));)

typedef void (*kern_t)(uint64_t*, uint32_t);
struct K { const char* name; kern_t fn; };

int main() {
    K ks[] = {{"v_mad_u64_u32", k_mad_u64_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
              {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_lshl_add_u64", k_lshl_add_u64},
              {"v_add_u32", k_add_u32}, {"v_addc_co_u32", k_addc_u32}, {"v_mad_u32_u24", k_mad_u32_u24},
              {"v_mov_b32", k_mov_b32}, {"v_fma_f32", k_fma_f32}, {"v_cvt_f64_u32", k_cvt_f64_u32}};
    uint64_t* d;
    hipMalloc(&d, 4096 * sizeof(uint64_t));
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    // waves per SIMD: 1, 2, 4  (block = 256 threads = 4 waves = 1 per SIMD; blocks per CU via grid)
    for (auto& k : ks) {
        for (int wps : {1, 2, 4}) {
            int blocks = prop.multiProcessorCount * wps;  // each 256-thread block puts 1 wave on each SIMD
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k.fn<<<blocks, 256>>>(d, 7);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k.fn<<<blocks, 256>>>(d, 7);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> h(blocks);
            hipMemcpy(h.data(), d, blocks * sizeof(uint64_t), hipMemcpyDeviceToHost);
            double cyc = 0; for (auto v : h) cyc += v; cyc /= blocks;
            double ninst = (double)ITERS * 64;
            // s_memtime/readcyclecounter ticks at a constant 100 MHz on gfx9; report wall-derived rate too
            double inst_per_s = ninst * wps * 4 * prop.multiProcessorCount / (ms * 1e-3);
            printf("%-16s waves/SIMD=%d  ticks/inst=%.3f  wall=%.3f ms  => %.2f Ginst/s chip, %.2f cyc/inst/SIMD @2.4GHz\n",
                   k.name, wps, cyc / ninst, ms, inst_per_s * 1e-9, 2.4e9 * 4 * prop.multiProcessorCount / inst_per_s);
        }
    }
    return 0;
}
