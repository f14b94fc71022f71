// Micro-benchmark: per-CU load rate of a 16 x 128-float tile per wave (the fused encoder's activation fragments) under
// different lane->address maps, sharing between workgroups and cache states.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// mode 0: lane (lo, hi): row lo, 16 B at column 16 t + 4 hi (64-byte pieces of 16 rows per instruction)
// mode 1: contiguous 1 KB per instruction (lane l: 16 B at (64 t + l) * 16)
// mode 2: lane (lo, hi): row lo, 32 B (two loads) at column 32 t' + 8 hi -> 128-byte pieces of 16 rows per instruction pair
__global__ __launch_bounds__(512) void ldbench(const float* __restrict__ src, float* __restrict__ out, unsigned long long* cyc, int mode,
                                               int nbuf, long bstride, int share, int rot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 15, hi = lane >> 4;
    const int inst = share ? blockIdx.x / share : blockIdx.x;
    const int c = share ? blockIdx.x % share : 0;
    const float* base = src + (size_t)inst * 112 * 128;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave >= 7) return;
    for (int pass = 0; pass < 2; ++pass) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        float4 v[5][8];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int qq = rot ? (q + c) % nbuf : q;
            const float* p = base + (size_t)min(qq, nbuf - 1) * bstride;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (mode == 0) v[q][t] = ld4(p + (16 * wave + lo) * 128 + 16 * t + 4 * hi);
                else if (mode == 1) v[q][t] = ld4(p + 16 * wave * 128 + (64 * t + lane) * 4);
                else v[q][t] = ld4(p + (16 * wave + lo) * 128 + 32 * (t >> 1) + 8 * hi + 4 * (t & 1));
            }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float wq = q < nbuf ? 1.f : 0.f;
                acc.x = fmaf(wq, v[q][t].x, acc.x); acc.y += v[q][t].y * wq; acc.z += v[q][t].z * wq; acc.w += v[q][t].w * wq;
            }
        // the sum depends on every load: the stamp below is after the last arrival
        const float tot = acc.x + acc.y + acc.z + acc.w;
        const unsigned long long t1 = __builtin_amdgcn_s_memtime() + (tot == 1234.5f ? 1 : 0);
        if (lane == 0) cyc[((size_t)blockIdx.x * 8 + wave) * 2 + pass] = t1 - t0;
    }
    *reinterpret_cast<float4*>(out + ((size_t)blockIdx.x * 512 + tid) * 4) = acc;
}
__global__ void fill(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023) * 1e-3f;
}
int main() {
    const int NWG = 256;
    const long bstride = 64L * 112 * 128 * 4;            // floats between partial buffers (>= all instances), generous
    const size_t n = (size_t)bstride * 6;
    float *src, *out; unsigned long long* cyc;
    if (hipMalloc(&src, n * 4) != hipSuccess || hipMalloc(&out, (size_t)NWG * 512 * 16) != hipSuccess || hipMalloc(&cyc, NWG * 8 * 2 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    printf("src %p out %p cyc %p\n", (void*)src, (void*)out, (void*)cyc); fflush(stdout);
    std::vector<unsigned long long> h(NWG * 8 * 2);
    // (every configuration issues 5 x 8 loads per lane: nbuf = 1 reads ONE tile five times, nbuf = 5 five different tiles)
    for (int mode = 0; mode < 2; ++mode)
        for (int share : {0, 4})
            for (int rot : {0})
                for (int nbuf : {1, 5}) {
                    double s0 = 0, s1 = 0;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, src, n);       // rewrite: the reads below start L2-cold
                        hipLaunchKernelGGL(ldbench, dim3(NWG), dim3(512), 0, 0, src, out, cyc, mode, nbuf, bstride, share, rot);
                        if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 1; }
                        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
                        double a = 0, b = 0;
                        for (int w = 0; w < NWG; ++w) for (int k = 0; k < 7; ++k) { a += h[(w * 8 + k) * 2]; b += h[(w * 8 + k) * 2 + 1]; }
                        s0 = a / (NWG * 7); s1 = b / (NWG * 7);
                    }
                    const double bytes = 5 * 7 * 16 * 512.0;           // requested per workgroup (40 loads per lane)
                    fflush(stdout); printf("mode %d share %d rot %d nbuf %d: cold %.0f cyc (%.1f B/clk/CU)  warm %.0f cyc (%.1f B/clk/CU)\n", mode, share, rot,
                           nbuf, s0, bytes / s0, s1, bytes / s1);
                }
    return 0;
}
