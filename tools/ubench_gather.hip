// tools/ubench_gather.hip -- calibrates rocprofv3's FETCH_SIZE for the MSM accumulation's access pattern:
// every lane gathers one 64-byte (G1) or 128-byte (G2) element at a pseudo-random index of a table much
// larger than L2 + Infinity Cache, as four / eight 16-byte loads.  The byte count is known (gathers x size),
// so FETCH_SIZE / known gives the factor to apply to the counter for k_msm_accumulate
// (MI355X_MICROARCH.md: the x2 correction is calibrated for wide coalesced streams only).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o tools/ubench_gather
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/ubench_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct E64 { uint4 a, b, c, d; };
struct E128 { uint4 a, b, c, d, e, f, g, h; };

template <class E>
__global__ void k_gather(const E* __restrict__ table, uint32_t mask, int per_lane, uint32_t* __restrict__ out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = tid * 2654435761u + 12345u, acc = 0;
    for (int k = 0; k < per_lane; ++k) {
        x = x * 1664525u + 1013904223u;
        const E e = table[(x >> 4) & mask];
        acc ^= e.a.x ^ e.b.y ^ e.c.z ^ e.d.w;
        if (sizeof(E) == 128) { const E128& g = reinterpret_cast<const E128&>(e); acc ^= g.e.x ^ g.f.y ^ g.g.z ^ g.h.w; }
    }
    out[tid] = acc;
}
__global__ void k_stream(const uint4* __restrict__ in, size_t n, uint32_t* __restrict__ out) {
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (size_t i = tid; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = in[i]; acc ^= v.x ^ v.w; }
    out[tid] = acc;
}

int main() {
    const size_t bytes = (size_t)4 << 30;   // 4 GiB table
    void* table;
    uint32_t* out;
    hipMalloc(&table, bytes);
    hipMemset(table, 1, bytes);
    const int lanes = 1 << 20, per_lane = 32;
    hipMalloc(&out, lanes * sizeof(uint32_t));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_gather<E64>, dim3(lanes / 256), dim3(256), 0, 0, (const E64*)table, (uint32_t)(bytes / 64 - 1), per_lane, out);
        hipLaunchKernelGGL(k_gather<E128>, dim3(lanes / 256), dim3(256), 0, 0, (const E128*)table, (uint32_t)(bytes / 128 - 1), per_lane, out);
        hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const uint4*)table, bytes / 16, out);
    }
    hipDeviceSynchronize();
    std::printf("known bytes per launch: gather64 %zu, gather128 %zu, stream %zu\n", (size_t)lanes * per_lane * 64, (size_t)lanes * per_lane * 128, bytes);
    return 0;
}
