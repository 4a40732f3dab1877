// tools/ubench_power.hip -- SUSTAINED clock (and, with rocm-smi beside it, power) of gfx950 under one VALU instruction class.
// The prover runs at the package power limit (profiles/r4_power.txt), so what an instruction costs is energy, not only an issue
// slot: this prints the shader clock the chip settles at after seconds of multiply-adds, of 32-bit adds, and of mixes of the two.
//   tools/_bin/ubench_power <seconds per case> [waves per SIMD]
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_power.hip -o tools/_bin/ubench_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Stamp { uint64_t cyc, wall; };
#define MAD(N) "v_mad_i64_i32 %" #N ", s[20:21], %8, %9, %" #N "\n"
#define ADD(N) "v_add_u32 %" #N ", %8, %" #N "\n"
#define XOR(N) "v_xor_b32 %" #N ", %8, %" #N "\n"
// MIX = multiply-adds per group of 8 instructions (the rest are 32-bit adds on the low halves of other chains)
template <int MIX>
__global__ __launch_bounds__(256) void k_mix(Stamp* out, uint32_t trips, uint32_t seed) {
    uint32_t a = (seed * 2654435761u + threadIdx.x * 40503u) & 0x1fffffff, b = (seed * 97u + threadIdx.x * 7919u + 12345u) & 0x1fffffff;
    uint64_t c0 = a + b, c1 = a * 2 + b, c2 = a * 3 + b, c3 = a * 4 + b;
    uint32_t d0 = a, d1 = b, d2 = a ^ b, d3 = a + 1;
    const uint64_t t0 = clock64(), w0 = wall_clock64();
    for (uint32_t i = 0; i < trips; ++i) {
#define G8(I0, I1, I2, I3, I4, I5, I6, I7) I0 I1 I2 I3 I4 I5 I6 I7
#define REP8(x) x x x x x x x x
        if (MIX == 8) asm volatile(REP8(G8(MAD(0), MAD(1), MAD(2), MAD(3), MAD(0), MAD(1), MAD(2), MAD(3))) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "s20", "s21");
        if (MIX == 6) asm volatile(REP8(G8(MAD(0), MAD(1), MAD(2), ADD(4), MAD(3), MAD(0), MAD(1), ADD(5))) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "s20", "s21");
        if (MIX == 4) asm volatile(REP8(G8(MAD(0), ADD(4), MAD(1), ADD(5), MAD(2), ADD(6), MAD(3), ADD(7))) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "s20", "s21");
        if (MIX == 0) asm volatile(REP8(G8(ADD(4), ADD(5), ADD(6), ADD(7), XOR(4), XOR(5), XOR(6), XOR(7))) : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b) : "s20", "s21");
    }
    const uint64_t t1 = clock64(), w1 = wall_clock64();
    uint64_t sink = c0 + c1 + c2 + c3 + d0 + d1 + d2 + d3;
    if ((threadIdx.x & 63) == 0) {
        Stamp s{t1 - t0, w1 - w0};
        if (sink == 0x123456789abcull) s.cyc = 0;
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
    }
}
// Random 29-bit operands that differ from instruction to instruction (what a field multiplication feeds the multiplier array):
// per trip MADS multiply-adds over 9 x 9 limb registers into four accumulators, and ADDS 32-bit adds / xors over the same limbs.
template <int MADS, int ADDS>
__global__ __launch_bounds__(256) void k_limbs(Stamp* out, uint32_t trips, uint32_t seed) {
    int32_t a[9], b[9];
    uint32_t h = seed * 2654435761u + (blockIdx.x * 256 + threadIdx.x) * 40503u;
#pragma unroll
    for (int i = 0; i < 9; ++i) { h = h * 1664525u + 1013904223u; a[i] = (int32_t)(h >> 3) - (1 << 28); h = h * 1664525u + 1013904223u; b[i] = (int32_t)(h >> 3) - (1 << 28); }
    int64_t acc[4] = {1, 2, 3, 4};
    const uint64_t t0 = clock64(), w0 = wall_clock64();
    for (uint32_t t = 0; t < trips; ++t) {
#pragma unroll
        for (int k = 0; k < MADS; ++k) acc[k & 3] += (int64_t)a[k % 9] * b[(k * 4 + k / 9) % 9];
#pragma unroll
        for (int k = 0; k < ADDS; ++k) {
            if (k & 1) a[k % 9] ^= b[(k + 3) % 9]; else b[k % 9] += a[(k + 5) % 9];
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
        for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
    }
    const uint64_t t1 = clock64(), w1 = wall_clock64();
    int64_t sink = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) sink += a[i] + b[i];
    if ((threadIdx.x & 63) == 0) {
        Stamp s{t1 - t0, w1 - w0};
        if (sink == 0x123456789abcll) s.cyc = 0;
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
    }
}
typedef void (*kern_t)(Stamp*, uint32_t, uint32_t);
int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const int wps = argc > 2 ? atoi(argv[2]) : 3;
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * wps;
    Stamp* d;
    HIPCHECK(hipMalloc(&d, (size_t)blocks * 4 * sizeof(Stamp)));
    struct Case { const char* name; kern_t fn; int mads; int per_trip; } cases[] = {
        {"8 mad / 8 (fixed operands)", k_mix<8>, 8, 64}, {"8 add,xor / 8 (fixed)", k_mix<0>, 0, 64},
        {"64 mad, random limbs", k_limbs<64, 0>, 8, 64}, {"48 mad + 16 add, random", k_limbs<48, 16>, 6, 64}, {"32 mad + 32 add, random", k_limbs<32, 32>, 4, 64},
        {"64 add/xor, random limbs", k_limbs<0, 64>, 0, 64}};
    printf("# %d CUs, %d waves per SIMD, %.1f s per case; columns: case | launches | last-second shader clock MHz | G wave-instr/s | G wave-mad/s\n", cus, wps, seconds);
    for (auto& c : cases) {
        const uint32_t trips = 40000;   // 64 instructions per trip
        auto t0 = std::chrono::steady_clock::now();
        int launches = 0;
        double mhz = 0, rate = 0;
        printf("BEGIN %s\n", c.name); fflush(stdout);
        while (true) {
            auto l0 = std::chrono::steady_clock::now();
            c.fn<<<blocks, 256>>>(d, trips, 7 + launches);
            HIPCHECK(hipDeviceSynchronize());
            auto l1 = std::chrono::steady_clock::now();
            ++launches;
            const double el = std::chrono::duration<double>(l1 - t0).count();
            if (el >= seconds) {
                std::vector<Stamp> h((size_t)blocks * 4);
                HIPCHECK(hipMemcpy(h.data(), d, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
                double cyc = 0, wall = 0;
                for (auto& s : h) { cyc += (double)s.cyc; wall += (double)s.wall; }
                mhz = cyc / wall * 100.0;
                rate = (double)trips * 64 * wps * 4.0 * cus / std::chrono::duration<double>(l1 - l0).count() * 1e-9;
                break;
            }
        }
        printf("END %-20s | %d | %.0f | %.1f | %.1f\n", c.name, launches, mhz, rate, rate * c.mads / 8);
        fflush(stdout);
    }
    return 0;
}
