// Micro-benchmark: cost of accumulating a 16 x 128 tile per wave into global memory with float atomics (4 workgroups add to the
// same tile), against plain 16-byte stores of the same tile.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void atb(float* __restrict__ dst, unsigned long long* cyc, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 15, hi = lane >> 4;
    const int inst = blockIdx.x >> 2, c = blockIdx.x & 3;
    if (wave >= 7) return;
    float* base = dst + ((size_t)inst * 112 + 16 * wave) * 128;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {                 // contiguous 256 B per instruction
#pragma unroll
        for (int k = 0; k < 32; ++k) atomicAdd(base + (k >> 1) * 128 + 64 * (k & 1) + lane, 1.0f + k);
    } else if (mode == 1) {          // fragment layout: row lo, column 16 t + 4 hi + i
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(base + lo * 128 + 16 * t + 4 * hi + i, 1.0f + t);
    } else {                         // plain stores into the part's own buffer (fragment layout, 16 B per lane)
        float* p = base + (size_t)c * 64 * 112 * 128;
#pragma unroll
        for (int t = 0; t < 8; ++t) *reinterpret_cast<float4*>(p + lo * 128 + 16 * t + 4 * hi) = make_float4(1.f, 2.f, 3.f, 4.f + t);
    }
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): the atomics / stores have been accepted
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    float* dst; unsigned long long* cyc;
    const size_t n = (size_t)4 * 64 * 112 * 128;
    if (hipMalloc(&dst, n * 4) != hipSuccess || hipMalloc(&cyc, 256 * 8 * 8) != hipSuccess) return 1;
    std::vector<unsigned long long> h(256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float ms = 0; double a = 0;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipMemset(dst, 0, n * 4);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(atb, dim3(256), dim3(512), 0, 0, dst, cyc, mode);
            hipEventRecord(e1, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("fail\n"); return 1; }
            hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            a = 0; for (int w = 0; w < 256; ++w) for (int k = 0; k < 7; ++k) a += h[w * 8 + k];
            a /= 256 * 7;
        }
        printf("mode %d: wave issue->accepted %.0f cyc, kernel %.1f us\n", mode, a, ms * 1e3);
    }
    return 0;
}
