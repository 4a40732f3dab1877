// tools/ubench_valu.hip -- VALU issue-rate ceiling of gfx950 for the instruction classes the 254-bit
// Montgomery multiplier (zksnark_rs_amd/csrc/lazy29.cuh) is built from.  Design input for DESIGN.md 4a and the
// peak behind bench.py's `roofline.alu`.
//
// Round 1's version timed 40-140 us kernels with HIP events: launch overhead and the clock ramp sat inside the
// measurement, and its wall-clock "cycles @2.4 GHz" (5.09 for v_mad_u64_u32) contradicted its own in-kernel cycle
// counter (2.9).  This version
//   * runs every point for >= 5 ms (the repeat count is calibrated per kernel), after a 50 ms warm-up;
//   * reads BOTH clocks inside the kernel: clock64() = s_memtime (shader cycles) and wall_clock64() =
//     s_memrealtime (constant 100 MHz), so the effective shader clock under THIS load is measured, not assumed;
//   * sweeps waves/SIMD 1, 2, 3, 4, 6, 8 and 1 / 2 / 4 / 8 independent dependency chains per wave;
//   * prints machine-readable rows; rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
//     GRBM_GUI_ACTIVE over the same binary cross-checks the instruction counts and the clock.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int UNROLL = 64;   // instructions per loop trip (per wave)

struct Stamp { uint64_t cyc, wall; };

// Every measured loop body is ONE asm statement of UNROLL instructions: between separate asm statements hipcc's hazard
// recogniser inserts an s_nop per instruction (it cannot see inside them), which compiler-generated multiplier code
// does not contain (checked in the disassembly of k_msm_accumulate: dependent v_mad_u64_u32 back to back).
// Operands %0..%7 = chain registers, %8 / %9 = loop-invariant sources.
#define R2(x) x x
#define R4(x) R2(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)
#define R64(x) R32(x) R32(x)
#define BODY1(T) R64(T(0))
#define BODY2(T) R32(T(0) T(1))
#define BODY4(T) R16(T(0) T(1) T(2) T(3))
#define BODY8(T) R8(T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7))

#define DEF_OP(NAME, TYPE, T, CLOB...)                                                                         \
    template <int CHAINS>                                                                                      \
    __global__ __launch_bounds__(256) void NAME(Stamp* out, uint32_t trips, uint32_t seed) {                   \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;                                                     \
        TYPE c0 = a + b, c1 = a * 2 + b, c2 = a * 3 + b, c3 = a * 4 + b, c4 = a * 5 + b, c5 = a * 6 + b, c6 = a * 7 + b, c7 = a * 8 + b; \
        const uint64_t t0 = clock64(), w0 = wall_clock64();                                                    \
        for (uint32_t i = 0; i < trips; ++i) {                                                                 \
            if (CHAINS == 1) asm volatile(BODY1(T) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : CLOB); \
            if (CHAINS == 2) asm volatile(BODY2(T) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : CLOB); \
            if (CHAINS == 4) asm volatile(BODY4(T) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : CLOB); \
            if (CHAINS == 8) asm volatile(BODY8(T) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : CLOB); \
        }                                                                                                      \
        const uint64_t t1 = clock64(), w1 = wall_clock64();                                                    \
        TYPE sink = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;                                                     \
        if ((threadIdx.x & 63) == 0) {                                                                         \
            Stamp s{t1 - t0, w1 - w0};                                                                         \
            if ((uint64_t)sink == 0x123456789abcull) s.cyc = 0;                                                \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;                                                      \
        }                                                                                                      \
    }

#define T_MAD_U64(N) "v_mad_u64_u32 %" #N ", s[20:21], %8, %9, %" #N "\n"
#define T_MAD_I64(N) "v_mad_i64_i32 %" #N ", s[20:21], %8, %9, %" #N "\n"
#define T_LSHL_ADD(N) "v_lshl_add_u64 %" #N ", %" #N ", 0, %" #N "\n"
#define T_ASHR64(N) "v_ashrrev_i64 %" #N ", 1, %" #N "\n"
#define T_FMA64(N) "v_fma_f64 %" #N ", %" #N ", %" #N ", %" #N "\n"
#define T_MUL_LO(N) "v_mul_lo_u32 %" #N ", %8, %" #N "\n"
#define T_MUL_HI(N) "v_mul_hi_u32 %" #N ", %8, %" #N "\n"
#define T_ADD32(N) "v_add_u32 %" #N ", %8, %" #N "\n"
#define T_AND32(N) "v_and_b32 %" #N ", %8, %" #N "\n"
#define T_ADDC(N) "v_addc_co_u32 %" #N ", vcc, %8, %" #N ", vcc\n"
#define T_MOV(N) "v_mov_b32 %" #N ", %8\n"
DEF_OP(k_mad_u64_u32, uint64_t, T_MAD_U64, "s20", "s21")
DEF_OP(k_mad_i64_i32, uint64_t, T_MAD_I64, "s20", "s21")
DEF_OP(k_lshl_add_u64, uint64_t, T_LSHL_ADD, "s20")
DEF_OP(k_ashrrev_i64, uint64_t, T_ASHR64, "s20")
DEF_OP(k_fma_f64, uint64_t, T_FMA64, "s20")
DEF_OP(k_mul_lo_u32, uint32_t, T_MUL_LO, "s20")
DEF_OP(k_mul_hi_u32, uint32_t, T_MUL_HI, "s20")
DEF_OP(k_add_u32, uint32_t, T_ADD32, "s20")
DEF_OP(k_and_b32, uint32_t, T_AND32, "s20")
DEF_OP(k_addc_co_u32, uint32_t, T_ADDC, "vcc")
DEF_OP(k_mov_b32, uint32_t, T_MOV, "s20")

// The multiplier's own column shape: per trip 18 multiply-adds into one 64-bit accumulator (9 products + 9 reduction
// terms), then mul_lo + and (m_k), one more multiply-add, the 64-bit arithmetic shift of the carry and the limb mask
// -- 19 mad + mul_lo + 2 and + ashr = 23 instructions, ONE dependency chain as in a column of FpR::mont (the compiler
// interleaves at most two such chains in k_msm_accumulate).
template <int COLS>
__global__ __launch_bounds__(256) void k_column(Stamp* out, uint32_t trips, uint32_t seed) {
    int32_t a[9], b[9], p[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { a[i] = (seed * (i + 3) + threadIdx.x) & 0x1fffffff; b[i] = (seed * (i + 7) + 5) & 0x1fffffff; p[i] = (0x12345 * (i + 1)) & 0x1fffffff; }
    int64_t carry[COLS];
    int32_t lim[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) { carry[j] = j; lim[j] = 0; }
    const uint64_t t0 = clock64(), w0 = wall_clock64();
    for (uint32_t i = 0; i < trips; ++i) {
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
            int64_t acc = carry[j];
            // two asm statements: the low half of the accumulator pair has no operand syntax of its own (hipcc puts one s_nop between them)
            asm volatile(
                "v_mad_i64_i32 %0, s[20:21], %1, %18, %0\n v_mad_i64_i32 %0, s[20:21], %2, %17, %0\n v_mad_i64_i32 %0, s[20:21], %3, %16, %0\n"
                "v_mad_i64_i32 %0, s[20:21], %4, %15, %0\n v_mad_i64_i32 %0, s[20:21], %5, %14, %0\n v_mad_i64_i32 %0, s[20:21], %6, %13, %0\n"
                "v_mad_i64_i32 %0, s[20:21], %7, %12, %0\n v_mad_i64_i32 %0, s[20:21], %8, %11, %0\n v_mad_i64_i32 %0, s[20:21], %9, %10, %0\n"
                "v_mad_i64_i32 %0, s[20:21], %19, %9, %0\n v_mad_i64_i32 %0, s[20:21], %20, %8, %0\n v_mad_i64_i32 %0, s[20:21], %21, %7, %0\n"
                "v_mad_i64_i32 %0, s[20:21], %22, %6, %0\n v_mad_i64_i32 %0, s[20:21], %23, %5, %0\n v_mad_i64_i32 %0, s[20:21], %24, %4, %0\n"
                "v_mad_i64_i32 %0, s[20:21], %25, %3, %0\n v_mad_i64_i32 %0, s[20:21], %26, %2, %0\n v_mad_i64_i32 %0, s[20:21], %27, %1, %0\n"
                : "+v"(acc)
                : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]),
                  "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]),
                  "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8])
                : "s20", "s21");
            uint32_t m = (uint32_t)acc, lo;
            asm volatile(
                "v_mul_lo_u32 %1, %1, %4\n v_and_b32 %1, 0x1fffffff, %1\n v_mad_i64_i32 %0, s[20:21], %1, %3, %0\n"
                "v_mov_b32 %2, 0x1fffffff\n v_ashrrev_i64 %0, 29, %0\n"
                : "+v"(acc), "+v"(m), "=&v"(lo) : "v"(p[0]), "v"(0x0fffffffu) : "s20", "s21");
            lo &= (uint32_t)acc;
            lim[j] ^= (int32_t)lo;
            carry[j] = acc;
        }
    }
    const uint64_t t1 = clock64(), w1 = wall_clock64();
    int64_t sink = 0;
#pragma unroll
    for (int j = 0; j < COLS; ++j) sink += carry[j] + lim[j];
    if ((threadIdx.x & 63) == 0) {
        Stamp s{t1 - t0, w1 - w0};
        if (sink == 0x123456789abcll) s.cyc = 0;
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
    }
}

typedef void (*kern_t)(Stamp*, uint32_t, uint32_t);
struct Case { std::string name; int chains; kern_t fn; double inst_per_trip; double mad_share; };

#define ADD_OP(v, K) do { v.push_back({&#K[2], 1, K<1>, UNROLL, 0}); v.push_back({&#K[2], 2, K<2>, UNROLL, 0}); \
                          v.push_back({&#K[2], 4, K<4>, UNROLL, 0}); v.push_back({&#K[2], 8, K<8>, UNROLL, 0}); } while (0)

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "--quick");
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s CUs %d nominal clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    std::vector<Case> cases;
    ADD_OP(cases, k_mad_u64_u32); ADD_OP(cases, k_mad_i64_i32); ADD_OP(cases, k_mul_lo_u32); ADD_OP(cases, k_add_u32);
    ADD_OP(cases, k_and_b32); ADD_OP(cases, k_lshl_add_u64); ADD_OP(cases, k_ashrrev_i64);
    if (!quick) { ADD_OP(cases, k_mul_hi_u32); ADD_OP(cases, k_addc_co_u32); ADD_OP(cases, k_fma_f64); ADD_OP(cases, k_mov_b32); }
    cases.push_back({"mont_column(19mad+6other)", 1, k_column<1>, 25, 19.0 / 25});
    Stamp* d;
    const int max_blocks = cus * 8;
    HIPCHECK(hipMalloc(&d, (size_t)max_blocks * 4 * sizeof(Stamp)));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    // warm-up: ~50 ms of multiply-adds on the whole chip so that the first measured point does not see the clock ramp
    k_mad_u64_u32<4><<<cus * 4, 256>>>(d, 400000, 7);
    HIPCHECK(hipDeviceSynchronize());
    printf("# columns: op chains waves_per_simd  cyc_per_inst_per_simd(in-kernel shader clock)  eff_clock_MHz  Ginst_per_s_chip(events)  ms\n");
    const double target_ms = quick ? 3.0 : 6.0;
    for (auto& c : cases) {
        for (int wps : {1, 2, 3, 4, 6, 8}) {
            const int blocks = cus * wps;   // a 256-thread block puts one wave on each SIMD of a CU
            // calibrate the trip count on a short run, then measure one >= target_ms launch
            uint32_t trips = 2000;
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                HIPCHECK(hipEventRecord(e0));
                c.fn<<<blocks, 256>>>(d, trips, 7);
                HIPCHECK(hipEventRecord(e1));
                HIPCHECK(hipDeviceSynchronize());
                HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
                if (pass == 0) trips = (uint32_t)(trips * target_ms / (ms > 1e-3 ? ms : 1e-3)) + 1;
            }
            std::vector<Stamp> h((size_t)blocks * 4);
            HIPCHECK(hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
            double cyc = 0, wall = 0;
            for (auto& s : h) { cyc += (double)s.cyc; wall += (double)s.wall; }
            cyc /= h.size(); wall /= h.size();
            const double inst_per_wave = (double)trips * c.inst_per_trip;
            const double cyc_per_inst_simd = cyc / inst_per_wave / wps;         // wps waves share a SIMD
            const double eff_mhz = wall > 0 ? cyc / wall * 100.0 : 0;           // wall ticks at 100 MHz
            const double ginst = inst_per_wave * wps * 4.0 * cus / (ms * 1e-3) * 1e-9;
            printf("%-34s %d %d  %.3f  %.0f  %.1f  %.2f\n", c.name.c_str(), c.chains, wps, cyc_per_inst_simd, eff_mhz, ginst, ms);
            fflush(stdout);
        }
    }
    return 0;
}
