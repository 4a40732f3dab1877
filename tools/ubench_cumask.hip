// ubench_cumask.hip -- which compute units does a stream created with hipExtStreamCreateWithCUMask use?
// Prints, for a few masks, the number of distinct (XCC, SE, SH, CU) places the workgroups of a chip-filling launch ran on, per XCC.
// Answers: is bit i of the mask compute unit i / 8 of XCC i % 8 (so that clearing bits 0..7 takes ONE unit from every XCD)?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_cumask.hip -o tools/ubench_cumask && tools/ubench_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(uint32_t* out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, words = (cus + 31) / 32, blocks = 8192;
    printf("%s: %d compute units\n", prop.gcnArchName, cus);
    uint32_t* d;
    CK(hipMalloc(&d, blocks * 8));
    std::vector<uint32_t> h(2 * blocks);
    struct Case { const char* name; int lo, hi; };   // bits [lo, hi) CLEARED
    const Case cases[] = {{"all units", 0, 0}, {"bits 0..7 cleared", 0, 8}, {"bits 0..15 cleared", 0, 16}, {"bits 0..31 cleared", 0, 32},
                          {"bits 8..15 cleared", 8, 16}, {"bits 248..255 cleared", 248, 256}};
    for (const Case& c : cases) {
        std::vector<uint32_t> mask(words, 0u);
        for (int i = 0; i < cus; ++i)
            if (i < c.lo || i >= c.hi) mask[i / 32] |= 1u << (i % 32);
        hipStream_t st;
        CK(hipExtStreamCreateWithCUMask(&st, words, mask.data()));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, st, d, 40000);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
        std::map<uint32_t, std::set<uint32_t>> per;
        for (int b = 0; b < blocks; ++b) {
            const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 15u;
            per[xcc].insert((hw >> 8) & 0xffu);   // CU_ID 11:8, SH_ID 12, SE_ID 15:13
        }
        size_t total = 0;
        printf("%-24s %.3f ms  units per XCC:", c.name, ms);
        for (auto& kv : per) { printf(" %u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
        printf("  total %zu\n", total);
        CK(hipStreamDestroy(st));
    }
    return 0;
}
