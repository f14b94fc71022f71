// Cooperative rollout kernels (N1 <= 112): shared device pieces -- the phase-clock macros, LDS pitches, the local policy on the
// matrix cores (head units / tail), and the owners' phases for four trajectories per wave (co_finish4 / co_advance4).
// Included by csrc/elg_fwd_coop.hip (the cooperative kernels) and csrc/elg_fwd.hip (the streaming kernels use the local-policy
// stages and their LDS table image).
#pragma once
#include "elg_rollout.h"
#include "elg_bf16.h"
#include <string>
#ifndef ELG_EXP_SKIP
#define ELG_EXP_SKIP 0
#endif

namespace elg {
using f32x4c = __attribute__((ext_vector_type(4))) float;
int fail(int code, const std::string& msg);
int launch_status(const char* what);
void note_kernel(int id);        // which construction kernel the calling thread launched last (elg_rollout_last_kernel)
// the cooperative rollout (csrc/elg_fwd_coop.hip): tsp / train select the instantiation, A.precision the arithmetic
int launch_fwd_coop_any(const elg_rollout_args& A, hipStream_t stream, bool tsp, bool train, int split);

// In-kernel phase clock of the cooperative kernel: only in the diagnostic build (-DELG_STAMPS, tools/stamp_coop.py); the
// shipped library executes no stamp.  Segment sums leave through elg_rollout_args.scratch (unused at this size).
#ifdef ELG_STAMPS
struct StampCtx { unsigned long long last; float acc[16]; };
__device__ __forceinline__ void stamp_at(StampCtx& c, float& slot) {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    slot += (float)(unsigned)(t - c.last);
    c.last = t;
}
#define ELG_STAMP(c, i) stamp_at(c, (c).acc[i])
#elif defined(ELG_MARKS)                 // ISA listing with the phase boundaries as comments (tools/isa_phase_mix.py); never shipped
struct StampCtx {};
#define ELG_STAMP(c, i) asm volatile("; ELG_PHASE_MARK " #i)
#else
struct StampCtx {};
#define ELG_STAMP(c, i)
#endif

#ifndef ELG_CO_QP
#define ELG_CO_QP 132
#endif
#ifndef ELG_CO_SP
#define ELG_CO_SP 116
#endif
#ifndef ELG_CO_XPAD
#define ELG_CO_XPAD 4
#endif
#ifndef ELG_CL_P
#define ELG_CL_P 36
#endif
#ifndef ELG_CL_Q
#define ELG_CL_Q 52
#endif
// load through a uniform base pointer + a 32-bit BYTE offset: base + zext(offset) is what the scalar-base + vector-offset form of
// global_load takes; an ELEMENT offset (shifted left in 64 bits) costs a 64-bit shift-add per load
template <typename T>
__device__ __forceinline__ T ld_off(const void* base, unsigned byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
constexpr int CO_QP = ELG_CO_QP;      // pitch of the query / glimpse-output exchange rows (conflict-free column reads)
constexpr int CO_SP = ELG_CO_SP;      // pitch of the score exchange rows
constexpr int CO_NT = 7;        // node tiles of 16
constexpr int CO_MAXTR = 32;    // trajectories in lockstep per workgroup

// ---- slot exchange block of a trajectory (floats): f0 | f1 | f2 | slot code | penalty | u   (48 each)
// (block pitch 288 + 4: with 288 = 9 x 32 floats every trajectory's block started on the same LDS bank, and the reads that put
// the TRAJECTORY on the lane -- the local policy's feature / slot-code fragments, 16 lanes x ds_read_b128 -- were 16-way bank
// conflicts; 292 = 4 (mod 32) spreads the sixteen 16-byte pieces over all 64 banks.  PMC: SQ_LDS_BANK_CONFLICT was 45 % of
// SQ_LDS_IDX_ACTIVE in the cooperative kernel.)
constexpr int CO_XP = 6 * ELG_SLOT_STRIDE + ELG_CO_XPAD;
constexpr int CO_XF = 0, CO_XS = 3 * ELG_SLOT_STRIDE, CO_XPEN = 4 * ELG_SLOT_STRIDE, CO_XU = 5 * ELG_SLOT_STRIDE;
// ---- folded local-policy tables staged in LDS with conflict-free pitches (same images as csrc/elg_local.hip)
// Every MFMA operand that comes from a table is one ds_read_b128 (four k-steps at once): lcv is kept transposed ([d][52]: a
// lane's four values are consecutive slots), lpe / lwc row-major with pitch 36 (consecutive channels), lt transposed per head,
// lAv / lWe padded to four floats per channel.  Pitches 36 / 52 keep the 16 lanes of a b128 group on distinct banks.
constexpr int CL_P = ELG_CL_P, CL_Q = ELG_CL_Q;
constexpr int CL_LCVT = 0;                              // [32 d][52]   lcv[j][d] transposed
constexpr int CL_LPE = CL_LCVT + 32 * CL_Q;             // [48 j][36]
constexpr int CL_LWC = CL_LPE + 48 * CL_P;              // [32][36]
constexpr int CL_LTT = CL_LWC + 32 * CL_P;              // [4 heads][48]
constexpr int CL_LAV = CL_LTT + 4 * 48;                 // [32][4]
constexpr int CL_LWE = CL_LAV + 32 * 4;                 // [32][4]
constexpr int CL_LBC = CL_LWE + 32 * 4;                 // [32]
constexpr int CL_LA = CL_LBC + 32;                      // [4 heads][4]  la[h][k], k < 3
constexpr int CL_SIZE = CL_LA + 16;                     // = 5040 floats

__device__ __forceinline__ void co_stage_local(const float* __restrict__ loc, float* sT, int tid, int nthreads) {
    for (int i = tid; i < 48 * 32; i += nthreads) {
        const int j = i >> 5, d = i & 31;
        sT[CL_LCVT + d * CL_Q + j] = loc[ELG_LOC_LCV + i];
        sT[CL_LPE + j * CL_P + d] = loc[ELG_LOC_LPE + i];
    }
    for (int i = tid; i < 32 * 32; i += nthreads) sT[CL_LWC + (i >> 5) * CL_P + (i & 31)] = loc[ELG_LOC_LWC + i];
    for (int i = tid; i < 48 * 4; i += nthreads) sT[CL_LTT + (i & 3) * 48 + (i >> 2)] = loc[ELG_LOC_LT + i];
    for (int i = tid; i < 128; i += nthreads) {
        const int d = i >> 2, k = i & 3;
        sT[CL_LAV + i] = k < 3 ? loc[ELG_LOC_LAV + 3 * d + k] : 0.f;
        sT[CL_LWE + i] = k < 3 ? loc[ELG_LOC_LWE + 3 * d + k] : 0.f;
    }
    for (int i = tid; i < 32; i += nthreads) sT[CL_LBC + i] = loc[ELG_LOC_LBC + i];
    for (int i = tid; i < 16; i += nthreads) sT[CL_LA + i] = (i & 3) < 3 ? loc[ELG_LOC_LA + 3 * (i >> 2) + (i & 3)] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// The local policy of 16 lockstep trajectories (models.py:133-166, folded as in elg_rollout.h::local_policy) on the matrix cores, in two
// stages -- feature-major tiles X[slot 16 jt + 4 hi + v][trajectory lo]; the three table contractions (alpha -> o', o' -> g', g' -> u)
// are MFMAs whose D tiles are the next B operands.  (A single-wave chain of all of it, ~12 K cycles, was the critical path of the
// pointer phase in every kernel that had it.)
//   head units      one (head h, 16-trajectory group) unit: attention of head h over the slots and its 8 channels of o'
//                   (12 MFMAs).  The eight units of a workgroup run on the eight waves inside the GLIMPSE phase, their stages
//                   written between the glimpse's MFMA loops (rollout_fwd_coop_kernel: lh_score / lh_exp / lh_norm / lh_mfma),
//                   o' goes to LDS.
//   co_local_tail   g' = Wc o' + bc, w = g' . Lwe, u_j = Lpe_j . g' + w . f_j for one group (40 MFMAs): one wave per group in
//                   the pointer phase, ~1/3 of the old chain.
// sO1 layout per group (floats): [dt][lane][4] = o'[16 dt + 4 hi + v][trajectory lo]  (the tail's B operands, one b128 each).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void co_local_tail(const float* __restrict__ sT, const float* sXrows, const float* sO1, float* sUrows,
                                              int upitch, int lo, int hi) {
    constexpr int JT = 3;
    const float* X = sXrows + lo * CO_XP;
    const f32x4c z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4c o1[2], g1[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const float4 t = *reinterpret_cast<const float4*>(sO1 + (dt * 64 + 16 * hi + lo) * 4);
        o1[dt] = f32x4c{t.x, t.y, t.z, t.w};
    }
    float w[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int dq = 0; dq < 2; ++dq) {
        const float4 bc4 = *reinterpret_cast<const float4*>(sT + CL_LBC + 16 * dq + 4 * hi);
        f32x4c acc = {bc4.x, bc4.y, bc4.z, bc4.w};
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const float4 w4 = *reinterpret_cast<const float4*>(sT + CL_LWC + (16 * dq + lo) * CL_P + 16 * dt + 4 * hi);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, o1[dt][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, o1[dt][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, o1[dt][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, o1[dt][3], acc, 0, 0, 0);
        }
        g1[dq] = acc;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 we = *reinterpret_cast<const float4*>(sT + CL_LWE + 4 * (16 * dq + 4 * hi + v));
            w[0] = fmaf(acc[v], we.x, w[0]); w[1] = fmaf(acc[v], we.y, w[1]); w[2] = fmaf(acc[v], we.z, w[2]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { w[k] = quarters_sum(w[k]); }
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        f32x4c acc = z4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const float4 p4 = *reinterpret_cast<const float4*>(sT + CL_LPE + (16 * jt + lo) * CL_P + 16 * dt + 4 * hi);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p4.x, g1[dt][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p4.y, g1[dt][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p4.z, g1[dt][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(p4.w, g1[dt][3], acc, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(X + CO_XF + k * ELG_SLOT_STRIDE + 16 * jt + 4 * hi);
            acc[0] = fmaf(w[k], t.x, acc[0]); acc[1] = fmaf(w[k], t.y, acc[1]);
            acc[2] = fmaf(w[k], t.z, acc[2]); acc[3] = fmaf(w[k], t.w, acc[3]);
        }
        *reinterpret_cast<float4*>(sUrows + lo * upitch + 16 * jt + 4 * hi) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// The two stages as out-of-line calls for rollout_fwd_mt_kernel (round 4): head unit (h, group) -- attention of local head h over the
// slots of a 16-trajectory group and its 8 channels of o' (the stages lh_score / lh_exp / lh_norm / lh_mfma of the cooperative kernel
// in a row) -- at the start of the glimpse phase on the waves that finish that phase first, the tail on one wave per group in the
// pointer phase (out of line: their ~250 live registers must not shape the register allocation of the streaming loops).
static __device__ __attribute__((noinline)) void co_local_head_call(const float* sT, const float* sXrows, float* sO1, int h, int lo, int hi) {
    const float* LX = sXrows + lo * CO_XP;
    const int dt = h >> 1;
    f32x4c lf[3][3], lal[3];
    bool lmsk[3][4];
    float lmx = ELG_NEG_INF, lden = 0.f, lF[3] = {0.f, 0.f, 0.f};
    f32x4c lP = {0.f, 0.f, 0.f, 0.f};
    const float4 la4 = *reinterpret_cast<const float4*>(sT + CL_LA + 4 * h);
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
        const int4 sl = *reinterpret_cast<const int4*>(LX + CO_XS + 16 * jt + 4 * hi);
        lmsk[jt][0] = sl.x < 0; lmsk[jt][1] = sl.y < 0; lmsk[jt][2] = sl.z < 0; lmsk[jt][3] = sl.w < 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 tq = *reinterpret_cast<const float4*>(LX + CO_XF + k * ELG_SLOT_STRIDE + 16 * jt + 4 * hi);
            lf[k][jt] = f32x4c{tq.x, tq.y, tq.z, tq.w};
        }
        const float4 lt4 = *reinterpret_cast<const float4*>(sT + CL_LTT + h * 48 + 16 * jt + 4 * hi);
        const float ltv[4] = {lt4.x, lt4.y, lt4.z, lt4.w};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float x = ltv[v];
            x = fmaf(la4.x, lf[0][jt][v], x);
            x = fmaf(la4.y, lf[1][jt][v], x);
            x = fmaf(la4.z, lf[2][jt][v], x);
            x = lmsk[jt][v] ? ELG_NEG_INF : x;
            lal[jt][v] = x;
            lmx = fmaxf(lmx, x);
        }
    }
    lmx = quarters_max(lmx);
#pragma unroll
    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float e = lmsk[jt][v] ? 0.f : __expf(lal[jt][v] - lmx);
            lal[jt][v] = e;
            lden += e;
        }
    lden = quarters_sum(lden);
    const float lrden = lden > 0.f ? 1.0f / lden : 0.f;
#pragma unroll
    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float a = lal[jt][v] * lrden;
            lal[jt][v] = a;
#pragma unroll
            for (int k = 0; k < 3; ++k) lF[k] = fmaf(a, lf[k][jt][v], lF[k]);
        }
#pragma unroll
    for (int k = 0; k < 3; ++k) lF[k] = quarters_sum(lF[k]);
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
        const float4 a4 = *reinterpret_cast<const float4*>(sT + CL_LCVT + (16 * dt + lo) * CL_Q + 16 * jt + 4 * hi);
        lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, lal[jt][0], lP, 0, 0, 0);
        lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, lal[jt][1], lP, 0, 0, 0);
        lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, lal[jt][2], lP, 0, 0, 0);
        lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, lal[jt][3], lP, 0, 0, 0);
    }
    // rows 4 hi + v of the 16-channel tile dt: channels 8 (h & 1) .. + 7 belong to head h
    if ((hi >= 2) == bool(h & 1)) {
        float xo[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 lav = *reinterpret_cast<const float4*>(sT + CL_LAV + 4 * (16 * dt + 4 * hi + v));
            xo[v] = fmaf(lav.z, lF[2], fmaf(lav.y, lF[1], fmaf(lav.x, lF[0], lP[v])));
        }
        *reinterpret_cast<float4*>(sO1 + (dt * 64 + 16 * hi + lo) * 4) = make_float4(xo[0], xo[1], xo[2], xo[3]);
    }
}
static __device__ __attribute__((noinline)) void co_local_tail_call(const float* sT, float* sXrows, const float* sO1, int lo, int hi) {
    co_local_tail(sT, sXrows, sO1, sXrows + CO_XU, CO_XP, lo, hi);
}

template <bool TSP>
__device__ __forceinline__ void co_store_state(int* sS, const Traj<2>& st, int lane) {
    if (lane == 0) {
        sS[0] = st.cur; sS[1] = st.first; sS[2] = st.cnt; sS[3] = st.fin;
        sS[4] = f2i(st.load); sS[5] = f2i(st.len); sS[6] = f2i(st.cx); sS[7] = f2i(st.cy);
        sS[8] = (int)(unsigned)st.vis[0]; sS[9] = (int)(unsigned)(st.vis[0] >> 32);
        sS[10] = (int)(unsigned)st.vis[1]; sS[11] = (int)(unsigned)(st.vis[1] >> 32);
    }
}
// 16-lane (DPP row) integer min / max all-reduce
__device__ __forceinline__ int row16_min_i(int v) {
    v = min(v, f2i(quad_xor1(i2f(v)))); v = min(v, f2i(quad_xor2(i2f(v))));
    v = min(v, f2i(dpp<0x141>(i2f(v)))); v = min(v, f2i(dpp<0x140>(i2f(v))));
    return v;
}
__device__ __forceinline__ int row16_max_i(int v) {
    v = max(v, f2i(quad_xor1(i2f(v)))); v = max(v, f2i(quad_xor2(i2f(v))));
    v = max(v, f2i(dpp<0x141>(i2f(v)))); v = max(v, f2i(dpp<0x140>(i2f(v))));
    return v;
}

// Clip / mask / softmax / choice (models.py:405-420, CVRPModel.py:53-70) of the FOUR trajectories a wave owns at
// once: 16 lanes per trajectory (row tq = lane >> 4), node n = lo + 16 k in register k (7 registers cover 112
// nodes).  Softmax reductions are 16-lane DPP row reductions, the inverse-CDF sample a DPP row scan per register
// chunk (node order = k-major), the arg-max a (value desc, node asc) row reduction.  One pass of ~250 VALU
// instructions for four trajectories instead of ~400 per trajectory with a whole wavefront each.
// Results (chosen node, its probability) go to dwords 12 / 13 of the trajectory's state block.
// LEAN: the production launch (sampled or greedy choice from the kernel's own Philox stream; no teacher forcing, no external
// uniforms, no probability dump): the test / diagnostic branches and their pointers are compiled out of the step loop.
template <bool TSP, bool TRAIN, bool LEAN = false>
__device__ __forceinline__ void co_finish4(const elg_rollout_args& A, int N1, int lane, int wave, int ntraj, int t, int g_lo,
                                           size_t b, size_t Rcap, float* sSc, const unsigned long long* sMask,
                                           const float* sX, int* sState, int fin_row, int& sel_out, float& p_out, float& ubuf,
                                           StampCtx& sc) {
    constexpr int NK = CO_NT;
    const int tq = lane >> 4, lo = lane & 15;
    const int q = 4 * wave + tq;
    const float dflt = (LEAN || A.has_penalty) ? A.xi : 0.f;
    // ---- slot terms (penalty + local policy) scattered into the score rows: every row's 16 lanes take three slots each of
    // the row's own trajectory; all reads are issued before the writes (the slots of a trajectory are distinct nodes), so the
    // read-modify-write costs one LDS round trip instead of four serialised ones
    {
        const float* X = sX + q * CO_XP;
        const bool qok = q < ntraj;
        int sn[3];
        float add[3], cur[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {                 // (unconditional reads, selected afterwards: all 32 slot blocks exist)
            const int j = lo + 16 * c3;
            const int code_r = reinterpret_cast<const int*>(X)[CO_XS + j];
            const float add_r = X[CO_XPEN + j] + X[CO_XU + j] * A.inv_ens - dflt;
            const int code = qok ? code_r : -1;
            sn[c3] = (code == -2) ? 0 : code;
            add[c3] = qok ? add_r : 0.f;
        }
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            const float c_r = sSc[q * CO_SP + max(sn[c3], 0)];
            cur[c3] = sn[c3] >= 0 ? c_r : 0.f;
        }
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3)
            if (sn[c3] >= 0) sSc[q * CO_SP + sn[c3]] = cur[c3] + add[c3];
    }
    wave_lds_fence();
    ELG_STAMP(sc, 4);
    const bool act = q < ntraj && (fin_row >= 0 ? fin_row == 0 : sState[16 * q + 3] == 0);   // decoding this step (row-uniform)
    const unsigned long long w0 = sMask[2 * q], w1 = sMask[2 * q + 1];
    const int m = g_lo + q;
    const size_t bm = b * A.M + m;
    const size_t r = (size_t)t * A.M + m;
    float e[NK], th[NK];
    float mx = ELG_NEG_INF;
    {
        // branch-free (round 4): the closed bit of node lo + 16 k as a 0 / -1 word (two 64-bit shifts per lane instead of seven;
        // nodes past N1 are closed in the mask words), all score reads issued together, tanh for every lane and the closed ones
        // masked afterwards -- `if (!masked)` around the LDS read + tanh chain was seven serialised divergent regions.  Same values
        // for the open nodes; a closed node gets th = 0, x = -inf as before (whatever its score slot holds).
        const unsigned long long x0 = w0 >> lo, x1 = w1 >> lo;
        const int wd[4] = {(int)(unsigned)x0, (int)(unsigned)(x0 >> 32), (int)(unsigned)x1, (int)(unsigned)(x1 >> 32)};
        float sv[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) sv[k] = sSc[q * CO_SP + lo + 16 * k];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int closed = __builtin_amdgcn_sbfe(wd[k >> 1], 16 * (k & 1), 1);          // bit lo + 16 k of (w0, w1)
            const float t = fast_tanh(sv[k] + dflt);
            th[k] = i2f(f2i(t) & ~closed);
            const float x = i2f((f2i(A.clip * t) & ~closed) | (closed & (int)0xff800000u));
            e[k] = x;
            mx = fmaxf(mx, x);
        }
    }
    mx = row16_max(mx);
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        e[k] = (e[k] > ELG_NEG_INF) ? __expf(e[k] - mx) : 0.f;
        part += e[k];
    }
    const float tot = row16_sum(part);
    const float inv = tot > 0.f ? 1.0f / tot : 0.f;
    ELG_STAMP(sc, 5);
    if (!LEAN && A.full_probs && t < A.dump_T && act) {
        float* frow = A.full_probs + (bm * A.dump_T + t) * N1;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int n = lo + 16 * k;
            if (n < N1) {
                float v = e[k] * inv;                                   // probabilities
                if (A.dump_logits) {                                    // 1: clipped logits, 2: the score before the clip
                    const bool masked = (((k < 4) ? w0 : w1) >> (n & 63)) & 1ull;
                    v = masked ? ELG_NEG_INF : (A.dump_logits == 2 ? sSc[q * CO_SP + n] + dflt : A.clip * th[k]);
                }
                frow[n] = v;
            }
        }
    }
    // ---- choose
    int sel = 0;
    if (!LEAN && A.mode == ELG_MODE_FORCED) {
        sel = (act && A.forced && t < A.Tforced) ? A.forced[bm * A.Tforced + t] : 0;
    } else if (A.mode == ELG_MODE_GREEDY) {
        // argmax of the trajectory's row (16 lanes), ties -> lowest node index: row maximum by DPP, then the first node that
        // attains it from one ballot per 16-node slice (the (value, index) butterfly went through ds_bpermute: 8 LDS round trips)
        float pvk[NK];
        float bv = -1.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            pvk[k] = lo + 16 * k < N1 ? e[k] * inv : -1.f;
            bv = fmaxf(bv, pvk[k]);
        }
        bv = row16_max(bv);
        int bn = 0;
        bool got = false;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const unsigned long long hit = __ballot(pvk[k] == bv);
            const unsigned seg = (unsigned)(hit >> (lane & 48)) & 0xffffu;       // this trajectory's 16 lanes
            if (!got && seg) { bn = 16 * k + __builtin_ctz(seg); got = true; }
        }
        sel = bn;
    } else {
        float uni = 0.f;
        if (!LEAN && A.uniforms) {
            if (act) uni = A.uniforms[bm * A.Tmax + t];
        } else {
            // philox_uniform(seed, trajectory, step) as everywhere, but drawn 16 steps at a time: lane lo of the trajectory's
            // row holds the uniform of step (t & ~15) + lo (one Philox evaluation per 16 steps instead of one per step)
            if ((t & 15) == 0 || t == (TSP ? 1 : 2)) ubuf = philox_uniform(A.seed, (unsigned)bm, (unsigned)((t & ~15) + lo));
            uni = __shfl(ubuf, (lane & 48) | (t & 15), ELG_WAVE);
        }
        const float target = uni * tot;
        // Inverse CDF in node order (node lo + 16 k: chunk-major), in two levels (round 6): the 16-node chunk the target falls into
        // from the chunks' row totals, then ONE row scan inside that chunk -- a scan of every chunk was 7 x (4 + 4) dependent DPP
        // steps per row.  `run` is the same running sum of row totals as before; where rounding leaves no node of the chunk above
        // the target (or the target above the last total) the chunk's last open node is taken, as the old scan's `lastpos` did.
        float run = 0.f, base = 0.f;
        int kstar = -1, klast = 0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float tk = row16_sum(e[k]);
            const float nr = run + tk;
            klast = tk > 0.f ? k : klast;
            const bool here = kstar < 0 && nr > target;
            base = here ? run : base;
            kstar = here ? k : kstar;
            run = nr;
        }
        const bool fb = kstar < 0;                                   // target beyond the last total: last open node overall
        const int kc = fb ? klast : kstar;
        float ek = e[0];
#pragma unroll
        for (int k = 1; k < NK; ++k) ek = (kc == k) ? e[k] : ek;
        float c = ek;
        c += dpp<0x111>(c); c += dpp<0x112>(c); c += dpp<0x114>(c); c += dpp<0x118>(c);       // row inclusive scan
        c += base;
        const bool open = ek > 0.f;
        const unsigned long long bh = __ballot(open && !fb && c > target), bo = __ballot(open);
        const unsigned hit = (unsigned)(bh >> (lane & 48)) & 0xffffu, opn = (unsigned)(bo >> (lane & 48)) & 0xffffu;
        const int pos = hit ? __builtin_ctz(hit) : (opn ? 31 - __builtin_clz(opn) : 0);
        sel = 16 * kc + pos;
    }
    ELG_STAMP(sc, 6);
    // probability (and clip Jacobian) of the chosen node: held by lane (sel & 15), register sel >> 4
    const bool mine = (sel & 15) == lo;
    float pe = 0.f, pj = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
        if (mine && (sel >> 4) == k) { pe = e[k] * inv; pj = A.clip * (1.f - th[k] * th[k]); }
    pe = row16_sum(pe);
    if (TRAIN) {
        pj = row16_sum(pj);
        if (act && !(ELG_EXP_SKIP & 8)) {
            float* rPC = A.trPC + (b * Rcap + r) * N1;
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (lo + 16 * k < N1) rPC[lo + 16 * k] = e[k] * inv * (A.clip * (1.f - th[k] * th[k]));
            if (lo == 0) A.trCsel[b * Rcap + r] = pj;
        }
    }
    if (act && lo == 0) { sState[16 * q + 12] = sel; sState[16 * q + 13] = f2i(pe); }
    wave_lds_fence();
    sel_out = sel;
    p_out = pe;
    ELG_STAMP(sc, 7);
}

// State of the trajectory a 16-lane row works on (identical in the row's lanes, different between rows).
struct CoRow {
    int cur, first, cnt, fin;
    float load, len, cx, cy;
    unsigned long long v0, v1;
};
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }

// Environment transition (CVRPEnv.py:195-232, TSPEnv.py:108-124; same roundings as env_update()) and the next step's
// mask / query / k-NN slots (build_mask(), slot_setup()) for the wave's four trajectories at once, 16 lanes per trajectory.
template <bool TSP, bool TRAIN, bool LEAN = false>
__device__ __forceinline__ void co_advance4(const elg_rollout_args& A, const Inst& I, int N1, int lane, int wave, int ntraj,
                                            int t, int g_lo, size_t b, size_t Rcap, CoRow& st, int sel, bool active,
                                            unsigned long long* sMask, float* sQ, float* sX, float* sSc, StampCtx& sc) {
    const int tq = lane >> 4, lo = lane & 15;
    const int q = 4 * wave + tq;
    const int m = g_lo + q;
    // ---------------- every global load of the phase depends on the chosen node only: request them all up front (the query
    // row of the node, its sorted neighbour row), one exposed L2 round trip for the whole phase instead of two in series
    const int cur_n = active ? sel : st.cur;
    const int first_n = (TSP && active && st.cnt == 0) ? sel : st.first;
    const bool want_nbr = LEAN || A.has_penalty || A.has_local;
    float4 q1a, q1c, q2a = make_float4(0.f, 0.f, 0.f, 0.f), q2c = q2a;
    {
        // (32-bit element offsets from the instance's uniform bases: scalar base + vector offset addressing instead of a 64-bit
        // multiply-add chain per load -- an instance's tables are far below 4 GB)
        const unsigned o1 = 4u * (unsigned)(cur_n * ELG_E + 8 * lo);
        q1a = ld_off<float4>(I.Q1, o1);
        q1c = ld_off<float4>(I.Q1, o1 + 16u);
        if (TSP) {
            const unsigned o2 = 4u * (unsigned)(first_n * ELG_E + 8 * lo);
            q2a = ld_off<float4>(I.Q2, o2);
            q2c = ld_off<float4>(I.Q2, o2 + 16u);
        }
    }
    int nb_id[CO_NT];
    float nb_d[CO_NT], nb_th[CO_NT];
    if (want_nbr) {
        const unsigned row = (unsigned)(cur_n * N1);
#pragma unroll
        for (int k = 0; k < CO_NT; ++k) {
            const unsigned e = 4u * (row + (unsigned)min(lo + 16 * k, N1 - 1));
            nb_id[k] = ld_off<int>(I.nidx, e);
            nb_d[k] = ld_off<float>(I.ndist, e);
            nb_th[k] = ld_off<float>(I.ntheta, e);
        }
    } else {
#pragma unroll
        for (int k = 0; k < CO_NT; ++k) { nb_id[k] = 0; nb_d[k] = 0.f; nb_th[k] = 0.f; }
    }
    // ---------------- transition
    if (active) {
        const float sx = I.xy[2 * sel], sy = I.xy[2 * sel + 1];
        if (st.cnt > 0) st.len += dist2d(st.cx, st.cy, sx, sy);
        st.cx = sx; st.cy = sy;
        if (TSP) { if (st.cnt == 0) st.first = sel; }
        else st.load = (sel == 0) ? 1.0f : __fsub_rn(st.load, I.dem[sel]);
        if (sel < 64) st.v0 |= 1ull << sel; else st.v1 |= 1ull << (sel - 64);
        if (!TSP) { if (sel == 0) st.v0 |= 1ull; else st.v0 &= ~1ull; }
        st.cur = sel;
        st.cnt += 1;
        if (TSP) {
            if (st.cnt == N1) {
                st.len += dist2d(sx, sy, I.xy[2 * st.first], I.xy[2 * st.first + 1]);
                st.fin = 1;
            }
        } else {
            const unsigned long long f0 = N1 >= 64 ? ~0ull : ((1ull << N1) - 1ull);
            const unsigned long long f1 = N1 <= 64 ? 0ull : (N1 >= 128 ? ~0ull : ((1ull << (N1 - 64)) - 1ull));
            if ((st.v0 & f0) == f0 && (st.v1 & f1) == f1) st.fin = 1;
        }
    }
    // ---------------- the step decoded next (t + 1): mask words
    const bool nxt = q < ntraj && !st.fin && (t + 1 < A.Tmax) && (TSP ? (t + 1 >= 1) : (t + 1 >= 2));
    const size_t r1 = (size_t)(t + 1) * A.M + m;
    const float lim = __fadd_rn(st.load, 1e-6f);
    unsigned long long w0 = 0ull, w1 = 0ull;
    // (branch-free: the visited bits of the lane's seven nodes from two 64-bit shifts, the demands read up front -- a node past N1
    // reads LDS behind the demand row and is closed whatever it finds; `nxt` implies the trajectory is not finished, so the
    // finished-depot exception of build_mask() cannot apply here)
    const unsigned long long y0 = st.v0 >> lo, y1 = st.v1 >> lo;
    const unsigned vw[4] = {(unsigned)y0, (unsigned)(y0 >> 32), (unsigned)y1, (unsigned)(y1 >> 32)};
    float dk[CO_NT];
#pragma unroll
    for (int k = 0; k < CO_NT; ++k) dk[k] = TSP ? 0.f : I.dem[lo + 16 * k];
#pragma unroll
    for (int k = 0; k < CO_NT; ++k) {
        const int n = lo + 16 * k;
        bool m1 = (vw[k >> 1] >> (16 * (k & 1))) & 1u;
        if (!TSP) m1 = m1 || (lim < dk[k]);
        const bool mm = !(n < N1 && nxt) || m1;
        // the glimpse takes the mask as the C operand of its S = K q^T MFMAs (0 for an open node, -inf for a closed one):
        // written into the trajectory's SCORE row, which is free from here until the pointer phase of the next step
        // refills it (one select + one LDS store per lane and chunk instead of three VALU per score in every head's wave)
        sSc[q * CO_SP + n] = mm ? ELG_NEG_INF : 0.f;
        const unsigned long long bal = __ballot(mm);
        const unsigned long long rowbits = (bal >> (16 * tq)) & 0xFFFFull;
        if (k < 4) w0 |= rowbits << (16 * (k & 3)); else w1 |= rowbits << (16 * (k & 3));
    }
    if (N1 <= 64) w1 = ~0ull;                                           // nodes past N1 are closed, as in build_mask()
    else w1 |= 0xFFFF000000000000ull;                                   // (nodes 112..127)
    if (lo == 0) {
        sMask[2 * q] = w0; sMask[2 * q + 1] = w1;
        if (TRAIN && nxt && (LEAN || A.trMask)) { A.trMask[(b * Rcap + r1) * 2] = w0; A.trMask[(b * Rcap + r1) * 2 + 1] = w1; }
        if (TRAIN && nxt && ((LEAN && !TSP) || (!LEAN && A.trLoad))) A.trLoad[b * Rcap + r1] = st.load;
    }
    ELG_STAMP(sc, 8);
    // ---------------- query row: 8 channels per lane
    {
        float4 a = q1a, c = q1c;
        if (TSP) {
            a.x += q2a.x; a.y += q2a.y; a.z += q2a.z; a.w += q2a.w; c.x += q2c.x; c.y += q2c.y; c.z += q2c.z; c.w += q2c.w;
        } else {
            const float4 wa = *reinterpret_cast<const float4*>(I.wl + 8 * lo), wc = *reinterpret_cast<const float4*>(I.wl + 8 * lo + 4);
            a.x = fmaf(st.load, wa.x, a.x); a.y = fmaf(st.load, wa.y, a.y); a.z = fmaf(st.load, wa.z, a.z); a.w = fmaf(st.load, wa.w, a.w);
            c.x = fmaf(st.load, wc.x, c.x); c.y = fmaf(st.load, wc.y, c.y); c.z = fmaf(st.load, wc.z, c.z); c.w = fmaf(st.load, wc.w, c.w);
        }
        if (!nxt) { a = make_float4(0.f, 0.f, 0.f, 0.f); c = a; }
        *reinterpret_cast<float4*>(sQ + q * CO_QP + 8 * lo) = a;
        *reinterpret_cast<float4*>(sQ + q * CO_QP + 8 * lo + 4) = c;
        if (TRAIN && nxt && !(ELG_EXP_SKIP & 1)) {
            *reinterpret_cast<float4*>(A.trQ + (b * Rcap + r1) * ELG_E + 8 * lo) = a;
            *reinterpret_cast<float4*>(A.trQ + (b * Rcap + r1) * ELG_E + 8 * lo + 4) = c;
        }
    }
    ELG_STAMP(sc, 9);
    // ---------------- k-NN slots: first K open customers of cur's sorted neighbour row, rank = row scan
    constexpr int S0 = TSP ? 0 : 1;
    float* X = sX + q * CO_XP;
    int* Xi = reinterpret_cast<int*>(X);
    int kk = 0;
    if (want_nbr) {
        // candidate flags, their in-row ranks (DPP scan) and the row totals of all seven chunks are independent of each other:
        // only the running offset `found` chains them (two integer adds per chunk)
        // (round 4: the in-row rank and the row total from ONE ballot per chunk -- the row's 16 bits of it, two population counts --
        // instead of a DPP scan and a DPP all-reduce, 16 operations per chunk)
        int cI[CO_NT], exc[CO_NT], tot[CO_NT];
        const unsigned below = (1u << lo) - 1u;
#pragma unroll
        for (int k = 0; k < CO_NT; ++k) {
            const int i = lo + 16 * k;
            const bool valid = i < N1 && nxt;
            const int nid = nb_id[k];
            bool cand = valid && !(((nid < 64 ? w0 : w1) >> (nid & 63)) & 1ull);
            if (!TSP) cand = cand && (nid != 0);
            cI[k] = cand ? 1 : 0;
            const unsigned long long bal = __ballot(cand);
            const unsigned rowbits = (unsigned)(bal >> (16 * tq)) & 0xFFFFu;
            exc[k] = __popc(rowbits & below);
            tot[k] = __popc(rowbits);
        }
        int found = 0;
#pragma unroll
        for (int k = 0; k < CO_NT; ++k) {
            const int rank = found + exc[k];
            if (cI[k] && rank < A.K) {
                X[CO_XF + S0 + rank] = nb_d[k];
                X[CO_XF + ELG_SLOT_STRIDE + S0 + rank] = nb_th[k];
                Xi[CO_XS + S0 + rank] = nb_id[k];
            }
            found += tot[k];
        }
        kk = min(found, A.K);
    }
    wave_lds_fence();
    const float dmax = (kk > 0) ? X[CO_XF + S0 + kk - 1] : 0.f;
    wave_lds_fence();
    ELG_STAMP(sc, 10);
    const float nf = dmax + 1e-6f;
    const bool depot_closed = w0 & 1ull;
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) {
        const int j = lo + 16 * c3;
        const bool cust = nxt && (j >= S0) && (j < S0 + kk);
        // (branch-free: the slot's three words are read whether or not the lane holds a customer and the quotients formed for every
        // lane -- selected afterwards; three divergent regions with LDS reads and IEEE divisions inside serialised the slots)
        const float sd_r = X[CO_XF + j], sth_r = X[CO_XF + ELG_SLOT_STRIDE + j];
        const int snid_r = Xi[CO_XS + j];
        const float sd = cust ? sd_r : 0.f, sth = cust ? sth_r : 0.f;
        int snid = cust ? snid_r : -1;
        if (!TSP && j == 0 && nxt && (LEAN || A.has_penalty || A.has_local)) snid = 0;          // depot slot
        float pen = 0.f;
        if (LEAN || A.has_penalty) {
            float pq;
            if (TSP) pq = -(sd / (dmax + 1e-6f));
            else pq = (dmax != 0.f) ? -(sd / dmax) : -sd;
            pen = cust ? pq : 0.f;
        }
        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
        {
            const int sidx = cust ? snid_r : 0;                          // (a node of the instance whatever the slot holds)
            float g0 = sd / nf, g1 = sth;
            if (!LEAN && A.euclidean) {                                 // models.py:95-125: relative (x, y) / norm
                g0 = __fsub_rn(I.xy[2 * sidx], st.cx) / nf;
                g1 = __fsub_rn(I.xy[2 * sidx + 1], st.cy) / nf;
            }
            const float g2 = TSP ? 0.f : I.dem[sidx] / st.load;
            f0 = cust ? g0 : 0.f; f1 = cust ? g1 : 0.f; f2 = cust ? g2 : 0.f;
        }
        bool smask = !cust;
        if (!TSP && j == 0) smask = depot_closed;
        const int ssave = (smask && snid >= 0) ? -2 : snid;
        X[CO_XF + j] = f0; X[CO_XF + ELG_SLOT_STRIDE + j] = f1; X[CO_XF + 2 * ELG_SLOT_STRIDE + j] = f2;
        Xi[CO_XS + j] = ssave;
        X[CO_XPEN + j] = pen;
        X[CO_XU + j] = 0.f;
        if (TRAIN && nxt && (LEAN || A.trSlot) && !(ELG_EXP_SKIP & 2)) {
            A.trSlot[(b * Rcap + r1) * ELG_SLOT_STRIDE + j] = ssave;
            if (LEAN || A.trF) {
                float* fr = A.trF + (b * Rcap + r1) * (3 * ELG_SLOT_STRIDE) + j;
                fr[0] = f0; fr[ELG_SLOT_STRIDE] = f1; fr[2 * ELG_SLOT_STRIDE] = f2;
            }
        }
    }
    wave_lds_fence();
}

}  // namespace elg
