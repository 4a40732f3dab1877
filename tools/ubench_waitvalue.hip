// Does hipStreamWaitValue32 work on this stack, and how soon after the flag store does the waiting stream's kernel start?
// Kernel A: G workgroups of busy work; workgroup `rel` stores the flag when it STARTS.  Stream B: wait (flag >= seq), then kernel B,
// whose first workgroup records the wall clock.  Prints: A start, flag store, B start, A end (us, relative to A's first workgroup).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_waitvalue tools/ubench_waitvalue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void busy(unsigned long long* t, unsigned* flag, unsigned seq, int rel, int iters) {
    if (blockIdx.x == 0 && threadIdx.x == 0) t[0] = wall_clock64();
    if ((int)blockIdx.x == rel && threadIdx.x == 0) { t[1] = wall_clock64(); __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 7u) t[7] = x;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) t[3] = wall_clock64();
}
__global__ void after(unsigned long long* t) { if (blockIdx.x == 0 && threadIdx.x == 0) t[2] = wall_clock64(); }
int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned* flag = nullptr;
    CK(hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory));
    CK(hipMemset(flag, 0, 64));
    unsigned long long *t, h[8];
    CK(hipMalloc(&t, 64));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const int G = 12288, rel = G - 3072;   // four rounds of 3072 resident workgroups (256 CUs x 12 waves of 64 lanes)
    for (unsigned seq = 1; seq <= 3; ++seq) {
        CK(hipMemset(t, 0, 64));
        CK(hipDeviceSynchronize());
        CK(hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(after, dim3(1), dim3(64), 0, b, t);
        hipLaunchKernelGGL(busy, dim3(G), dim3(64), 0, a, t, flag, seq, rel, 200000);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, t, 64, hipMemcpyDeviceToHost));
        printf("seq %u: flag store at %.1f us, kernel B started at %.1f us, last workgroup of A at %.1f us (100 MHz wall clock)\n", seq,
               (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0, (h[3] - h[0]) / 100.0);
    }
    return 0;
}
