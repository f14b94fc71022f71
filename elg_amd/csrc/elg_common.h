// ELG-POMO rollout engine for MI355X (gfx950 / CDNA4): shared device helpers.
// Wave = 64 lanes everywhere; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ELG_WAVE 64
#define ELG_E 128          // embedding dim            (CVRP/config.yml:41)
#define ELG_H 8            // decoder heads            (:43)
#define ELG_DK 16          // decoder qkv dim          (:44)
#define ELG_LE 32          // local_att_hidden_dim     (:47)
#define ELG_LH 4           // local_att_head_num       (:48)
#define ELG_LDK 8          // local_att_qkv_dim        (:49)

#define ELG_NEG_INF (-__builtin_huge_valf())

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: the opt-in is remembered per (kernel
// instantiation, device) -- one `static elg::DynLds` per launch site.  Racing host threads at worst set it twice.
#include <atomic>
namespace elg {
struct DynLds {
    static constexpr int MAXDEV = 32;
    std::atomic<size_t> bytes[MAXDEV] = {};
    // true when the kernel may be launched with `want` bytes of dynamic LDS on the current device
    bool opt_in(const void* kern, size_t want) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const bool tracked = dev >= 0 && dev < MAXDEV;
        if (tracked && bytes[dev].load(std::memory_order_relaxed) >= want) return true;
        (void)hipGetLastError();
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) != hipSuccess) return false;
        if (tracked) bytes[dev].store(want, std::memory_order_relaxed);
        return true;
    }
};
}  // namespace elg

namespace elg {

__device__ __forceinline__ int f2i(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float i2f(int v) { return __builtin_bit_cast(float, v); }

// ---- DPP quad permutes (no LDS traffic) ------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return i2f(__builtin_amdgcn_mov_dpp(f2i(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor1(float v) { return dpp<0xB1>(v); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float quad_xor2(float v) { return dpp<0x4E>(v); }   // quad_perm [2,3,0,1]
template <int J>
__device__ __forceinline__ float quad_bcast(float v) { return dpp<J * 0x55>(v); }   // quad_perm [J,J,J,J]

__device__ __forceinline__ float shfl_xor(float v, int m) { return __shfl_xor(v, m, ELG_WAVE); }
__device__ __forceinline__ int shfl_xor(int v, int m) { return __shfl_xor(v, m, ELG_WAVE); }

// lane i <-> lane i ^ 16 / i ^ 32 all-reduce steps on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap: with both
// operands = v, the pair returned holds {rows 0,0,2,2 | rows 1,1,3,3} resp. {lower half twice | upper half twice}); the
// ds_bpermute route of __shfl_xor costs an LDS round trip per step
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float x16_sum(float v) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap((unsigned)f2i(v), (unsigned)f2i(v), false, false);
    return i2f((int)r[0]) + i2f((int)r[1]);
}
__device__ __forceinline__ float x32_sum(float v) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap((unsigned)f2i(v), (unsigned)f2i(v), false, false);
    return i2f((int)r[0]) + i2f((int)r[1]);
}
__device__ __forceinline__ float x16_max(float v) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap((unsigned)f2i(v), (unsigned)f2i(v), false, false);
    return fmaxf(i2f((int)r[0]), i2f((int)r[1]));
}
__device__ __forceinline__ float x32_max(float v) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap((unsigned)f2i(v), (unsigned)f2i(v), false, false);
    return fmaxf(i2f((int)r[0]), i2f((int)r[1]));
}
// sum / max over the four lanes {i, i ^ 16, i ^ 32, i ^ 48} (the four "quarters" holding one MFMA column)
__device__ __forceinline__ float quarters_sum(float v) { return x32_sum(x16_sum(v)); }
__device__ __forceinline__ float quarters_max(float v) { return x32_max(x16_max(v)); }

// row (16-lane) all-reduce with DPP only: quad butterflies, then row_half_mirror / row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += quad_xor1(v);
    v += quad_xor2(v);
    v += dpp<0x141>(v);          // row_half_mirror
    v += dpp<0x140>(v);          // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, quad_xor1(v));
    v = fmaxf(v, quad_xor2(v));
    v = fmaxf(v, dpp<0x141>(v));
    v = fmaxf(v, dpp<0x140>(v));
    return v;
}
// 8-lane (aligned group) all-reduce
__device__ __forceinline__ float oct_sum(float v) {
    v += quad_xor1(v);
    v += quad_xor2(v);
    v += dpp<0x141>(v);
    return v;
}
// wave all-reduce, result wave-uniform: four row totals are combined through SGPRs (no LDS crossbar)
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    const float a = i2f(__builtin_amdgcn_readlane(f2i(v), 0)), b = i2f(__builtin_amdgcn_readlane(f2i(v), 16));
    const float c = i2f(__builtin_amdgcn_readlane(f2i(v), 32)), d = i2f(__builtin_amdgcn_readlane(f2i(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const float a = i2f(__builtin_amdgcn_readlane(f2i(v), 0)), b = i2f(__builtin_amdgcn_readlane(f2i(v), 16));
    const float c = i2f(__builtin_amdgcn_readlane(f2i(v), 32)), d = i2f(__builtin_amdgcn_readlane(f2i(v), 48));
    return (a + b) + (c + d);
}
// sum over lanes 0..31 only (lanes >= 32 are ignored), wave-uniform
__device__ __forceinline__ float half_sum_lo(float v) {
    v = row16_sum(v);
    const float a = i2f(__builtin_amdgcn_readlane(f2i(v), 0)), b = i2f(__builtin_amdgcn_readlane(f2i(v), 16));
    return a + b;
}
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each 16-lane row with DPP row_shr
// (out-of-row sources read 0), then the row totals are added through SGPRs
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
    v += dpp<0x111>(v);          // row_shr:1
    v += dpp<0x112>(v);          // row_shr:2
    v += dpp<0x114>(v);          // row_shr:4
    v += dpp<0x118>(v);          // row_shr:8
    const float t0 = i2f(__builtin_amdgcn_readlane(f2i(v), 15));
    const float t1 = i2f(__builtin_amdgcn_readlane(f2i(v), 31));
    const float t2 = i2f(__builtin_amdgcn_readlane(f2i(v), 47));
    const int row = lane >> 4;
    float off = 0.f;
    if (row >= 1) off = t0;
    if (row >= 2) off += t1;
    if (row >= 3) off += t2;
    return v + off;
}
__device__ __forceinline__ float readlane(float v, int l) { return i2f(__builtin_amdgcn_readlane(f2i(v), l)); }
__device__ __forceinline__ int readlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int lanes_below(unsigned long long bal) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
}
// wave-local LDS hand-off: DS ops of one wave execute in issue order; stop the compiler reordering.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tanh through one v_exp + one v_rcp: 1 - 2/(e^{2x}+1).  Exact limits (+-1) for |x| large, absolute error
// ~1e-7 (the clipped logit 50*tanh moves by < 1e-5, far inside the 1e-4 relative parity bar).
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// lane i <- lane i^8 / i^4 inside each 16-lane row, DPP only (row_ror; no LDS crossbar)
__device__ __forceinline__ float row_xor8(float v) { return dpp<0x128>(v); }            // row_ror:8
__device__ __forceinline__ float row_xor4(float v, bool bit2) {
    const float up = dpp<0x12C>(v);      // row_ror:12 : lane i <- lane (i+4) & 15
    const float dn = dpp<0x124>(v);      // row_ror:4  : lane i <- lane (i-4) & 15
    return bit2 ? dn : up;
}

// packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32: two lanes of work per VALU issue slot)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 lo2(const float4 v) { return f32x2{v.x, v.y}; }
__device__ __forceinline__ f32x2 hi2(const float4 v) { return f32x2{v.z, v.w}; }
// a . b with two packed ops + one add
__device__ __forceinline__ float dot4p(const float4 a, const float4 b) {
    f32x2 t = lo2(a) * lo2(b);
    t = __builtin_elementwise_fma(hi2(a), hi2(b), t);
    return t.x + t.y;
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}

// Euclidean distance exactly as a plain fp32 evaluation: sqrt(dx*dx + dy*dy), every op rounded once
// (no fma contraction) -- CVRPEnv.py:148 / :261.
__device__ __forceinline__ float dist2d(float ax, float ay, float bx, float by) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by);
    // sqrtf, not __fsqrt_rn: this toolchain lowers __fsqrt_rn to the bare v_sqrt_f32 (1 ulp), sqrtf to the correctly rounded
    // sequence -- and a 1-ulp distance reorders near-equidistant neighbours (a different k-NN set than the reference's)
    return sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
}

// Philox-4x32-10 counter RNG: one uniform in [0,1) per (seed, trajectory, step).
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned traj, unsigned step) {
    unsigned c0 = traj, c1 = step, c2 = 0x243F6A88u, c3 = 0x85A308D3u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

}  // namespace elg
