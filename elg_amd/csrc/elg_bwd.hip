// Backward of the POMO construction: replay kernels.
//
// The reference differentiates through a Python tape of ~1000 aten ops per step (CVRP/train.py:112-125
// over utils.py:14-21).  Here the recorded actions are replayed: per-step state is a pure function
// of the action prefix, so each step is recomputed and differentiated inside one wavefront.
//   rollout_bwd_kernel : glimpse / pointer / softmax backward, emits the row factors of the
//                        per-instance gradient contractions (see include/elg_hip.h) + d u_slot
//   local_bwd_kernel   : k-NN + local-policy replay, gradient of the folded local tables reduced
//                        in LDS accumulators, one flush per workgroup
#include "elg_rollout.h"
#include "elg_bwd_internal.h"
#include "elg_bf16.h"
#include <cstdlib>
#include <string>

namespace elg {
int fail(int code, const std::string& msg);
int launch_status(const char* what);

// ---------------------------------------------------------------------------------------------
// common prologue: instance pointers + LDS staging (same carve as the forward kernel)
// ---------------------------------------------------------------------------------------------
template <bool TSP, bool LDSK, int NT>
__device__ __forceinline__ float* stage_instance(const elg_rollout_args& A, int b, float* lds, Inst& I) {
    const int N1 = A.N1, NE = N1 * ELG_E;
    float* p = lds;
    float *sK = nullptr, *sV = nullptr, *sPK = nullptr;
    if (LDSK) { sK = p; sV = p + NE; sPK = p + 2 * NE; p += 3 * NE; }
    float* sdem = p; p += (N1 + 3) & ~3;
    float* sxy = p; if (LDSK) p += (2 * N1 + 3) & ~3;
    p += 4;
    const float* gK = A.Kmat + (size_t)b * NE;
    const float* gV = A.Vmat + (size_t)b * NE;
    const float* gPK = A.PK + (size_t)b * NE;
    if (LDSK) {
        for (int i = threadIdx.x; i < NE / 4; i += NT) {
            reinterpret_cast<float4*>(sK)[i] = reinterpret_cast<const float4*>(gK)[i];
            reinterpret_cast<float4*>(sV)[i] = reinterpret_cast<const float4*>(gV)[i];
            const int n = i >> 5, c4 = i & 31;
            reinterpret_cast<float4*>(sPK)[n * 32 + (c4 ^ (n & 31))] = reinterpret_cast<const float4*>(gPK)[i];
        }
    }
    if (!TSP)
        for (int i = threadIdx.x; i < N1; i += NT) sdem[i] = A.demand[(size_t)b * N1 + i];
    if (LDSK)
        for (int i = threadIdx.x; i < 2 * N1; i += NT) sxy[i] = A.xy[(size_t)b * N1 * 2 + i];
    I.K = LDSK ? sK : gK;
    I.V = LDSK ? sV : gV;
    I.PK = LDSK ? sPK : gPK;
    I.pb = A.pb + (size_t)b * N1;
    I.Q1 = A.Q1 + (size_t)b * NE;
    I.Q2 = TSP ? A.Q2 + (size_t)b * NE : nullptr;
    I.wl = A.wl;
    I.xy = LDSK ? sxy : A.xy + (size_t)b * N1 * 2;
    I.dem = sdem;
    I.nidx = A.nbr_idx + (size_t)b * N1 * N1;
    I.ndist = A.nbr_dist + (size_t)b * N1 * N1;
    I.ntheta = A.nbr_theta + (size_t)b * N1 * N1;
    I.loc = A.loc;
    return p;
}

__device__ __forceinline__ void unit_of_block(const elg_rollout_args& A, int& b, int& m_lo, int& m_hi) {
    const int G = gridDim.x;
    int u = blockIdx.x;
    if ((G & 7) == 0) u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    b = u / A.tiles;
    const int tile = u % A.tiles;
    const int tile_m = (A.M + A.tiles - 1) / A.tiles;
    m_lo = tile * tile_m;
    m_hi = min(A.M, m_lo + tile_m);
}

// ---------------------------------------------------------------------------------------------
// one decode step, forward recomputed + backward, for the trajectory of this wave
// ---------------------------------------------------------------------------------------------
template <int NCH, bool TSP, bool LDSK, bool SMALL>
__device__ __forceinline__ void bwd_step(const elg_bwd_args& BA, const Inst& I, const Traj<NCH>& st, int lane,
                                         float* sb, int sel, float gp, size_t b, size_t r, size_t R) {
    const elg_rollout_args& A = BA.fwd;
    constexpr int NG = GlimpseGroups<NCH, SMALL>::value;
    const int N1 = A.N1;
    const int half = lane >> 5, hq = lane & 31, ql = lane & 3, cb = hq * 4, head = hq >> 2;
    const int rr = 2 * ql + half;

    unsigned long long mk[NCH];
    build_mask<NCH, TSP>(st, I, N1, lane, mk);

    float4 q4 = *reinterpret_cast<const float4*>(I.Q1 + (size_t)st.cur * ELG_E + cb);
    if (TSP) {
        const float4 qf = *reinterpret_cast<const float4*>(I.Q2 + (size_t)st.first * ELG_E + cb);
        q4.x += qf.x; q4.y += qf.y; q4.z += qf.z; q4.w += qf.w;
    } else {
        const float4 w = *reinterpret_cast<const float4*>(I.wl + cb);
        q4.x = fmaf(st.load, w.x, q4.x); q4.y = fmaf(st.load, w.y, q4.y);
        q4.z = fmaf(st.load, w.z, q4.z); q4.w = fmaf(st.load, w.w, q4.w);
    }

    // ---- forward: slots, local policy, glimpse, pointer, softmax
    float addval = 0.f;
    int snid = -1;
    if (A.has_penalty || A.has_local) {
        const Slots S = slot_setup<NCH, TSP>(I, N1, A.K, A.has_penalty != 0, st, lane, mk, sb, nullptr, A.euclidean != 0, A.ens, A.Kens);
        snid = S.snid;
        float u = 0.f;
        if (A.has_local) u = local_ensemble<TSP>(A, I.loc, lane, S);
        addval = slot_penalty(A, S) + u * A.inv_ens;
    }
    GlimpseSave<NG> gs;
    const float4 o4 = glimpse<NCH, LDSK, NG>(I, N1, lane, q4, mk, &gs);
    float s[NCH];
    pointer_scores<NCH, LDSK>(I, N1, lane, o4, sb, s);

    const float dflt = A.has_penalty ? A.xi : 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int n = lane + 64 * ch;
        if (n < N1) sb[n] = dflt;
    }
    wave_lds_fence();
    if (snid >= 0) sb[snid] = addval;
    wave_lds_fence();
    float th[NCH], lg[NCH];
    float mx = ELG_NEG_INF;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int n = lane + 64 * ch;
        const bool masked = (mk[ch] >> lane) & 1ull;
        th[ch] = 0.f;
        lg[ch] = ELG_NEG_INF;
        if (n < N1 && !masked) {
            th[ch] = fast_tanh(s[ch] + sb[n]);
            lg[ch] = A.clip * th[ch];
        }
        mx = fmaxf(mx, lg[ch]);
    }
    wave_lds_fence();
    mx = wave_max(mx);
    float pn[NCH];
    float part = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        pn[ch] = (lg[ch] > ELG_NEG_INF) ? __expf(lg[ch] - mx) : 0.f;
        part += pn[ch];
    }
    const float inv = 1.0f / wave_sum(part);
    float psel = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        pn[ch] *= inv;
        if ((sel >> 6) == ch) psel = readlane(pn[ch], sel & 63);
    }

    // ---- softmax + clip backward: d s[n]   (models.py:416-420)
    const float gsel = gp * psel;
    float* rDL = BA.rowDL + (b * R + r) * N1;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int n = lane + 64 * ch;
        if (n < N1) {
            const float dlogit = gsel * ((n == sel ? 1.f : 0.f) - pn[ch]);
            const float ds = (lg[ch] > ELG_NEG_INF) ? dlogit * A.clip * (1.f - th[ch] * th[ch]) : 0.f;
            rDL[n] = ds;
            sb[n] = ds;
        }
    }
    wave_lds_fence();
    // d u_slot (local policy output), consumed by local_bwd_kernel
    const int dus = slot_stride_of(kmax_of(A));                  // rowDU of the replay: 48 floats per row up to local_size 47, 64 above
    if (BA.rowDU && lane < dus) {
        float du = 0.f;
        if (snid >= 0) du = sb[snid] * A.inv_ens;
        BA.rowDU[(b * R + r) * dus + lane] = du;
    }
    // rows consumed by the dense part of the backward (host: batched GEMMs, see engine.py):
    // a_h[n] (glimpse attention), q and o of this step, the load seen by the query
    {
        float* rA = BA.rowA + ((b * ELG_H + head) * R + r) * N1;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int row = 8 * k + rr;
            if (row < N1) rA[row] = gs.e[k];
        }
    }
    if (lane < 32) {
        const size_t off = (b * R + r) * ELG_E + cb;
        *reinterpret_cast<float4*>(BA.rowQ + off) = q4;
        *reinterpret_cast<float4*>(BA.rowO + off) = o4;
    }
    if (lane == 0 && BA.rowLoad) BA.rowLoad[b * R + r] = st.load;
}

// rows of a step that is not decoded (first moves, finished trajectories): all zero
template <int NCH>
__device__ __forceinline__ void zero_rows(const elg_bwd_args& BA, int lane, size_t b, size_t r, size_t R) {
    const int N1 = BA.fwd.N1;
    for (int h = 0; h < ELG_H; ++h) {
        float* rA = BA.rowA + ((b * ELG_H + h) * R + r) * N1;
        for (int n = lane; n < N1; n += 64) rA[n] = 0.f;
    }
    float* rDL = BA.rowDL + (b * R + r) * N1;
    for (int n = lane; n < N1; n += 64) rDL[n] = 0.f;
    const size_t off = (b * R + r) * ELG_E;
    for (int c = lane; c < ELG_E; c += 64) { BA.rowQ[off + c] = 0.f; BA.rowO[off + c] = 0.f; }
    const int dus = slot_stride_of(kmax_of(BA.fwd));
    if (BA.rowDU && lane < dus) BA.rowDU[(b * R + r) * dus + lane] = 0.f;
    if (lane == 0 && BA.rowLoad) BA.rowLoad[b * R + r] = 0.f;
}

template <int NCH, bool TSP, bool LDSK, int WAVES, bool SMALL>
__global__ __launch_bounds__(WAVES * 64) void rollout_bwd_kernel(const elg_bwd_args BA) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const elg_rollout_args& A = BA.fwd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N1 = A.N1, T = BA.T;
    int bi, m_lo, m_hi;
    unit_of_block(A, bi, m_lo, m_hi);
    Inst I;
    float* p = stage_instance<TSP, LDSK, WAVES * 64>(A, bi, lds, I);
    float* sb = p + wave * sb_floats_of(NCH, kmax_of(A));
    __syncthreads();
    const size_t b = bi, R = (size_t)A.M * T;

    for (int m = m_lo + wave; m < m_hi; m += WAVES) {
        const size_t bm = b * A.M + m;
        Traj<NCH> st;
        st.cur = 0; st.first = 0; st.cnt = 0; st.fin = 0; st.load = 1.0f; st.len = 0.f; st.cx = 0.f; st.cy = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) st.vis[c] = 0ull;
        for (int t = 0; t < T; ++t) {
            const size_t r = (size_t)m * T + t;
            const int sel = __builtin_amdgcn_readfirstlane(A.forced[bm * A.Tforced + t]);
            const bool first_move = (!TSP && t <= 1) || (TSP && t == 0);
            if (st.fin || first_move) {
                zero_rows<NCH>(BA, lane, b, r, R);
            } else {
                const float gp = i2f(__builtin_amdgcn_readfirstlane(f2i(BA.gprob[(b * T + t) * A.M + m])));
                bwd_step<NCH, TSP, LDSK, SMALL>(BA, I, st, lane, sb, sel, gp, b, r, R);
            }
            if (!st.fin) env_update<NCH, TSP>(st, I, N1, sel);
        }
    }
}

// =============================================================================================
// local-policy replay: gradient of the folded tables (layout ELG_LOC_*).  Every lane owns the rows
// of the tables it reads (slot rows j = lane, channel rows d = lane & 31) and keeps their gradient
// in registers for the whole launch; one LDS reduction + one global flush per workgroup at the end.
// =============================================================================================
struct LocAcc {
    float lpe[32];   // d lpe[j][*]      (lane j)
    float lcv[32];   // d lcv[j][*]      (lane j)
    float lwc[32];   // d lWc[d'][*]     (lane d' < 32)
    float lt[ELG_LH];
    float lwe[3];    // d lWe[d][*]      (lane d < 32)
    float lav[3];    // d lAv[d][*]      (lane d < 32)
    float lbc;       // d lbc[d]         (lane d < 32)
    float la;        // d la[3h+f]       (lane 3h+f < 12)
};

template <bool TSP>
__device__ __forceinline__ void local_bwd_step(const float* __restrict__ loc, LocAcc& A, int lane,
                                               const Slots& S, float du) {
    const int j = lane, dd = lane & 31, hh = dd >> 3;
    constexpr int NF = TSP ? 2 : 3;
    LocalSave sv;
    (void)local_policy<TSP>(loc, lane, S.f0, S.f1, S.f2, S.smask, &sv);
    const float f[3] = {S.f0, S.f1, TSP ? 0.f : S.f2};
    const float* lpe = loc + ELG_LOC_LPE + 32 * j;
    const float* lcv = loc + ELG_LOC_LCV + 32 * j;
    const float* lWe = loc + ELG_LOC_LWE + 3 * dd;
    const float* lAv = loc + ELG_LOC_LAV + 3 * dd;
    const float* wrow = loc + ELG_LOC_LWC + 32 * dd;
    const bool lo = lane < 32;

    // u_j = sum_d g'[d] lpe[j][d] + (sum_d g'[d] lWe[d]) . f_j
    float Sf[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NF; ++k) Sf[k] = wave_sum(du * f[k]);
    float dg;                                                       // d g'[dd]
    {
        float c[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) { c[d] = du * lpe[d]; A.lpe[d] = fmaf(du, readlane(sv.g, d), A.lpe[d]); }
        const float glo_ = reduce_scatter16(c, lane);
#pragma unroll
        for (int d = 0; d < 16; ++d) { c[d] = du * lpe[16 + d]; A.lpe[16 + d] = fmaf(du, readlane(sv.g, 16 + d), A.lpe[16 + d]); }
        const float ghi_ = reduce_scatter16(c, lane);
        dg = (lane & 16) ? ghi_ : glo_;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            dg = fmaf(lWe[k], Sf[k], dg);
            A.lwe[k] = fmaf(sv.g, Sf[k], A.lwe[k]);
        }
    }
    // g' = lWc o' + lbc
    A.lbc += dg;
    float dop;                                                      // d o'[dd]
    {
        const float dgl = lo ? dg : 0.f;
        float c[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) { c[d] = wrow[d] * dgl; A.lwc[d] = fmaf(dg, readlane(sv.op, d), A.lwc[d]); }
        const float plo = reduce_scatter16(c, lane);
#pragma unroll
        for (int d = 0; d < 16; ++d) { c[d] = wrow[16 + d] * dgl; A.lwc[16 + d] = fmaf(dg, readlane(sv.op, 16 + d), A.lwc[16 + d]); }
        const float phi = reduce_scatter16(c, lane);
        dop = (lane & 16) ? phi : plo;
    }
    // o'[d] = P[d] + lAv[d] . F_{h(d)}
    float G[ELG_LH][3];                                             // G[h][f] = sum_{d in h} do'[d] lAv[d][f]  (uniform)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < NF) {
            A.lav[k] = fmaf(dop, __shfl(sv.Ftot, 3 * hh + k, ELG_WAVE), A.lav[k]);
            const float t = oct_sum(dop * lAv[k]);
#pragma unroll
            for (int h = 0; h < ELG_LH; ++h) G[h][k] = readlane(t, 8 * h);
        } else {
#pragma unroll
            for (int h = 0; h < ELG_LH; ++h) G[h][k] = 0.f;
        }
    }
    float dal[ELG_LH];
#pragma unroll
    for (int h = 0; h < ELG_LH; ++h) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < NF; ++k) a = fmaf(f[k], G[h][k], a);
        dal[h] = a;
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) {
        const float dopd = readlane(dop, d);
        dal[d >> 3] = fmaf(dopd, lcv[d], dal[d >> 3]);
        A.lcv[d] = fmaf(sv.al[d >> 3], dopd, A.lcv[d]);
    }
    // softmax backward, then la / lt
    float sf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sf[i] = 0.f;
#pragma unroll
    for (int h = 0; h < ELG_LH; ++h) {
        const float tsum = wave_sum(sv.al[h] * dal[h]);
        const float dsc = sv.al[h] * (dal[h] - tsum);
        A.lt[h] += dsc;
#pragma unroll
        for (int k = 0; k < NF; ++k) sf[3 * h + k] = dsc * f[k];
    }
    A.la += reduce_scatter16(sf, lane);
}

template <int NCH, bool TSP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void local_bwd_kernel(const elg_bwd_args BA) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const elg_rollout_args& A = BA.fwd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N1 = A.N1, T = BA.T;
    int bi, m_lo, m_hi;
    unit_of_block(A, bi, m_lo, m_hi);
    Inst I;
    float* p = stage_instance<TSP, false, WAVES * 64>(A, bi, lds, I);
    float* acc = p; p += ELG_LOC_SIZE;
    float* sb = p + wave * sb_floats_of(NCH, kmax_of(A));
    for (int i = threadIdx.x; i < ELG_LOC_SIZE; i += WAVES * 64) acc[i] = 0.f;
    __syncthreads();
    const size_t b = bi, R = BA.row_stride > 0 ? (size_t)BA.row_stride : (size_t)A.M * T;
    LocAcc LA;
#pragma unroll
    for (int d = 0; d < 32; ++d) { LA.lpe[d] = 0.f; LA.lcv[d] = 0.f; LA.lwc[d] = 0.f; }
#pragma unroll
    for (int h = 0; h < ELG_LH; ++h) LA.lt[h] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { LA.lwe[k] = 0.f; LA.lav[k] = 0.f; }
    LA.lbc = 0.f; LA.la = 0.f;
    // rowDU rows: the replay's own (slot stride by local_size) or the saved-rows path's (row_stride > 0: always ELG_SLOT_STRIDE)
    const int dus = BA.row_stride > 0 ? ELG_SLOT_STRIDE : slot_stride_of(kmax_of(A));

    for (int m = m_lo + wave; m < m_hi; m += WAVES) {
        const size_t bm = b * A.M + m;
        Traj<NCH> st;
        st.cur = 0; st.first = 0; st.cnt = 0; st.fin = 0; st.load = 1.0f; st.len = 0.f; st.cx = 0.f; st.cy = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) st.vis[c] = 0ull;
        for (int t = 0; t < T; ++t) {
            const size_t r = BA.time_major ? (size_t)t * A.M + m : (size_t)m * T + t;
            const int sel = __builtin_amdgcn_readfirstlane(A.forced[bm * A.Tforced + t]);
            const bool first_move = (!TSP && t <= 1) || (TSP && t == 0);
            if (!st.fin && !first_move) {
                const float du = (lane < dus) ? BA.rowDU[(b * R + r) * dus + lane] : 0.f;
                if (__ballot(du != 0.f)) {
                    unsigned long long mk[NCH];
                    build_mask<NCH, TSP>(st, I, N1, lane, mk);
                    Slots S = slot_setup<NCH, TSP>(I, N1, A.K, A.has_penalty != 0, st, lane, mk, sb, nullptr, A.euclidean != 0, A.ens, A.Kens);
                    float dui = du;
                    if (A.ens > 1) {        // this launch's member: its features, its slots' cotangents
                        const bool in = member_slot<TSP>(S, BA.member, lane, A.euclidean != 0, S.f0, S.f1, S.f2, S.smask);
                        dui = in ? du : 0.f;
                    }
                    local_bwd_step<TSP>(I.loc + (size_t)BA.member * ELG_LOC_SIZE, LA, lane, S, dui);
                }
            }
            if (!st.fin) env_update<NCH, TSP>(st, I, N1, sel);
        }
    }
    // reduce the per-lane accumulators of the workgroup in LDS, then one flush to global memory
    {
        const int j = lane, dd = lane & 31;
#pragma unroll
        for (int d = 0; d < 32; ++d) {
            atomicAdd(acc + ELG_LOC_LPE + 32 * j + d, LA.lpe[d]);
            atomicAdd(acc + ELG_LOC_LCV + 32 * j + d, LA.lcv[d]);
            if (lane < 32) atomicAdd(acc + ELG_LOC_LWC + 32 * dd + d, LA.lwc[d]);
        }
#pragma unroll
        for (int h = 0; h < ELG_LH; ++h) atomicAdd(acc + ELG_LOC_LT + 4 * j + h, LA.lt[h]);
        if (lane < 32) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                atomicAdd(acc + ELG_LOC_LWE + 3 * dd + k, LA.lwe[k]);
                atomicAdd(acc + ELG_LOC_LAV + 3 * dd + k, LA.lav[k]);
            }
            atomicAdd(acc + ELG_LOC_LBC + dd, LA.lbc);
        }
        if (lane < 12) atomicAdd(acc + ELG_LOC_LA + lane, LA.la);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ELG_LOC_SIZE; i += WAVES * 64) {
        const float v = acc[i];
        if (v != 0.f) atomicAdd(BA.gloc + (size_t)BA.member * ELG_LOC_SIZE + i, v);
    }
}

// =============================================================================================
// dense-row part of the glimpse backward that is not a plain GEMM: per decode row r and head h
//     dA[n] = sum_d dO[r,h,d] V[n,h,d]            (K = 16: cheaper on the fly than as a GEMM + a pass)
//     dS[n] = a[n] (dA[n] - <dO_h, O_h>) / 4      softmax backward, /sqrt(qkv_dim)
//     dQ[d] = sum_n dS[n] K[n,h,d]
// rowA (B,8,R,N1) is read once and dS written once (both streamed, coalesced); V_h/K_h of the
// (instance, head) pair sit transposed in LDS.  One wavefront per row, lanes over nodes.
// =============================================================================================
template <int NCH>
__global__ __launch_bounds__(256) void glimpse_rows_kernel(const float* __restrict__ rowA, const float* __restrict__ dO,
                                                           const float* __restrict__ rowO, const float* __restrict__ Kmat,
                                                           const float* __restrict__ Vmat, float* __restrict__ dS,
                                                           float* __restrict__ dQ, int R, int N1, int rows_per_block,
                                                           size_t rowA_rows, size_t rowO_rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh >> 3, h = bh & 7;
    // every lane owns the same nodes n = lane + 64 c for all rows: keep their K_h / V_h rows in registers
    float vreg[NCH][16], kreg[NCH][16];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int n = lane + 64 * c;
        const size_t off = ((size_t)b * N1 + (n < N1 ? n : 0)) * ELG_E + h * 16;
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            const float4 v = *reinterpret_cast<const float4*>(Vmat + off + 4 * d4);
            const float4 k = *reinterpret_cast<const float4*>(Kmat + off + 4 * d4);
            const bool ok = n < N1;
            vreg[c][4 * d4 + 0] = ok ? v.x : 0.f; vreg[c][4 * d4 + 1] = ok ? v.y : 0.f;
            vreg[c][4 * d4 + 2] = ok ? v.z : 0.f; vreg[c][4 * d4 + 3] = ok ? v.w : 0.f;
            kreg[c][4 * d4 + 0] = ok ? k.x : 0.f; kreg[c][4 * d4 + 1] = ok ? k.y : 0.f;
            kreg[c][4 * d4 + 2] = ok ? k.z : 0.f; kreg[c][4 * d4 + 3] = ok ? k.w : 0.f;
        }
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(R, r0 + rows_per_block);
    for (int r = r0 + wave; r < r1; r += 4) {
        const size_t rowoff = ((size_t)bh * R + r) * N1;                 // dS (dense, R rows)
        const size_t aoff = ((size_t)bh * rowA_rows + r) * N1;           // rowA (rowA_rows rows per (b,h))
        const size_t voff = ((size_t)b * R + r) * ELG_E + h * 16;       // dO, dQ
        const size_t ooff = ((size_t)b * rowO_rows + r) * ELG_E + h * 16;
        float a[NCH];
        bool any = false;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int n = lane + 64 * c;
            a[c] = (n < N1) ? rowA[aoff + n] : 0.f;
            any = any || (a[c] != 0.f);
        }
        if (!__ballot(any)) {                      // inactive row (first moves / finished): all zero
#pragma unroll
            for (int c = 0; c < NCH; ++c) { const int n = lane + 64 * c; if (n < N1) dS[rowoff + n] = 0.f; }
            if (lane < 16) dQ[voff + lane] = 0.f;
            continue;
        }
        // dO_h, O_h of the row: 16 values each, loaded by lanes 0..15 and broadcast through SGPRs
        // (plain uniform loads were 2x slower here: every row waited on its own scalar-cache round trip)
        const float dol = (lane < 16) ? dO[voff + lane] : 0.f;
        const float ol = (lane < 16) ? rowO[ooff + lane] : 0.f;
        float dov[16], doto = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            dov[d] = readlane(dol, d);
            doto = fmaf(dov[d], readlane(ol, d), doto);
        }
        float part[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) part[d] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int n = lane + 64 * c;
            float da = 0.f;
#pragma unroll
            for (int d = 0; d < 16; ++d) da = fmaf(dov[d], vreg[c][d], da);
            const float ds = 0.25f * a[c] * (da - doto);
            if (n < N1) dS[rowoff + n] = ds;
#pragma unroll
            for (int d = 0; d < 16; ++d) part[d] = fmaf(ds, kreg[c][d], part[d]);
        }
        const float tot = reduce_scatter16(part, lane);
        if (lane < 16) dQ[voff + lane] = tot;
    }
}

// =============================================================================================
// launchers
// =============================================================================================
template <int NCH, bool TSP, bool LDSK, int WAVES, bool SMALL>
static int launch_bwd_impl(const elg_bwd_args& BA, hipStream_t stream);

template <int NCH, bool TSP, bool LDSK, int WAVES>
static int launch_bwd(const elg_bwd_args& BA, hipStream_t stream) {
    if (NCH == 2 && BA.fwd.N1 <= 104) return launch_bwd_impl<NCH, TSP, LDSK, WAVES, (NCH == 2)>(BA, stream);
    return launch_bwd_impl<NCH, TSP, LDSK, WAVES, false>(BA, stream);
}

template <int NCH, bool TSP, bool LDSK, int WAVES, bool SMALL>
static int launch_bwd_impl(const elg_bwd_args& BA, hipStream_t stream) {
    const elg_rollout_args& A = BA.fwd;
    size_t lds = 0;
    if (LDSK) lds += (size_t)3 * A.N1 * ELG_E * 4;
    lds += (size_t)((A.N1 + 3) & ~3) * 4 + 16 + (size_t)WAVES * sb_floats_of(NCH, kmax_of(A)) * 4;
    if (LDSK) lds += (size_t)((2 * A.N1 + 3) & ~3) * 4;
    if (lds > 163840) return fail(ELG_EINVAL, "rollout_bwd: LDS budget exceeded");
    auto kern = rollout_bwd_kernel<NCH, TSP, LDSK, WAVES, SMALL>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), 163840)) return fail(ELG_ELAUNCH, "hipFuncSetAttribute failed");
    if (!BA.local_only) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(kern, dim3(A.B * A.tiles), dim3(WAVES * 64), lds, stream, BA);
        if (launch_status("rollout_bwd") != ELG_OK) return ELG_ELAUNCH;
    }
    if (A.has_local) {
        constexpr int LW = 8;
        size_t l2 = (size_t)((A.N1 + 3) & ~3) * 4 + 16 + (size_t)ELG_LOC_SIZE * 4 + (size_t)LW * sb_floats_of(NCH, kmax_of(A)) * 4;
        auto k2 = local_bwd_kernel<NCH, TSP, LW>;
        static DynLds optin2;
        if (!optin2.opt_in(reinterpret_cast<const void*>(k2), 163840)) return fail(ELG_ELAUNCH, "hipFuncSetAttribute failed");
        elg_bwd_args BM = BA;          // one replay launch per ensemble member (register-resident table gradients)
        for (int i = 0; i < (A.ens > 1 ? A.ens : 1); ++i) {
            BM.member = i;
            (void)hipGetLastError();
            hipLaunchKernelGGL(k2, dim3(A.B * A.tiles), dim3(LW * 64), l2, stream, BM);
            if (launch_status("local_bwd") != ELG_OK) return ELG_ELAUNCH;
        }
    }
    return ELG_OK;
}

template <bool TSP>
static int dispatch_bwd(const elg_bwd_args& BA, hipStream_t stream) {
    const elg_rollout_args& A = BA.fwd;
    const int nch = (A.N1 + 63) / 64;
    if (A.lds_stage && A.N1 > 112) return fail(ELG_EINVAL, "lds_stage needs N1 <= 112");
    const bool lds = A.lds_stage != 0 && A.N1 <= 104;          // 105..112: tables from L2 (see dispatch_fwd)
    if (nch == 1) { if (lds) return launch_bwd<1, TSP, true, 8>(BA, stream); return launch_bwd<1, TSP, false, 8>(BA, stream); }
    if (nch == 2) { if (lds) return launch_bwd<2, TSP, true, 8>(BA, stream); return launch_bwd<2, TSP, false, 8>(BA, stream); }
    if (lds) return fail(ELG_EINVAL, "lds_stage needs N1 <= 104");
    if (nch <= 4) return launch_bwd<4, TSP, false, 8>(BA, stream);
    if (nch <= 8) return launch_bwd<8, TSP, false, 8>(BA, stream);
    if (nch <= 16) return launch_bwd<16, TSP, false, 8>(BA, stream);
    return fail(ELG_ENOTIMPL, "rollout_bwd: N1 > 1024 not built");
}


// =============================================================================================
// Fused glimpse backward on the matrix cores (v_mfma_f32_16x16x4_f32: exact f32, fmaf-chain numerics).
// One workgroup per (instance, head) [x row split], a wavefront per tile of 16 decode rows:
//     dA  = dO_h V_h^T                       (16 rows x N1)      4 MFMAs per 16 nodes
//     dS  = a (dA - <dO_h, O_h>) / 4         softmax backward, never written to memory
//     dQ_h = dS K_h                          contraction over nodes
//     dK_h += dS^T Q_h ,  dV_h += a^T dO_h   contractions over rows, accumulated in registers
// The node contraction wants dS with the ROW on the lane (B operand), the row contractions want the NODE on the
// lane (A operand): each 16 x 16 tile of dS is transposed once through per-wave LDS (ds_write_b128 + 4 ds_read_b32,
// 20-float pitch, conflict-free); the weights a are read from HBM exactly once.  Operand maps (lane l, lo = l & 15,
// hi = l >> 4):
//     A[i = lo][k = hi]   B[k = hi][j = lo]   D[i = 4 hi + reg][j = lo]
// and a k-slot may stand for any node / row as long as both operands agree (node 16 nt + 4 hi + v for
// the dQ product, row 4 hi + v for dK / dV), which is what lets D tiles feed the next MFMA directly.
// =============================================================================================
using f32x4 = __attribute__((ext_vector_type(4))) float;
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

// Node order inside a 16-node chunk nt: position p = 4 q + e (q = p >> 2, e = p & 3) stands for node
// min(16 nt + 4 q, N1 - 4) + e, i.e. the natural order except that the 4-node groups that would cross the end of
// the row are pulled back to end exactly at N1 - 1.  Every group is then a full in-bounds 16-byte run of the
// row (one unguarded dwordx4 load), at the price of re-visiting up to 3 nodes: the re-visits ("non-owners",
// node < the group's natural start) get a zero K operand and their dK / dV rows are not stored.
__device__ __forceinline__ int grp_start(int nt, int q, int N1) { return min(16 * nt + 4 * q, N1 - 4); }

// RECOMP: the weights a are not read (3.3 GB at the bench shape, written by the forward and read here) but recomputed
// per tile from the saved query rows and the rows' feasibility mask words: S = q_h K_h^T / 4 (28 more MFMAs per tile),
// masked softmax over the row's nodes (16-lane DPP reductions).
template <int NT, bool RECOMP, bool SEG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void glimpse_bwd_mfma_kernel(
    const float* __restrict__ rowA, const unsigned long long* __restrict__ rowMask, const float* __restrict__ dO,
    const float* __restrict__ rowO,
    const float* __restrict__ rowQ, const float* __restrict__ Kmat, const float* __restrict__ Vmat,
    float* __restrict__ dQ, float* __restrict__ dKp, float* __restrict__ dVp, int B, int R, int N1,
    size_t rowA_rows, size_t rowO_rows, size_t rowQ_rows, int splits, const GlimpseSeg seg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (SEG && seg.T_dev) R = min(R, seg.T_dev[0] * seg.M);           // the host has not read the rollout's length yet
    // LDS: [K operand image: NT*4 x 64] then the 4 x 2 x NT*256 reduction buffer
    float* sK = lds;
    float* sRed = lds + NT * 256;                                          // [2][2 NT 256] reduction buffer
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    float* sTr = lds + NT * 256 + 2 * 2 * NT * 256 + wave * (2 * 320);    // two 16 x 16 transpose tiles (pitch 20)
    float* sK2 = lds + NT * 256 + 2 * 2 * NT * 256 + 4 * 2 * 320;          // RECOMP: K as the B operand of q K^T
    float* sSeg1 = sRed;       // [(16 NT + 1)][16] d Q1 (+ the load row): aliases the reduction buffer, which is only
                               // used after the row loop (more LDS would halve the workgroups per CU)
    // channel-major [16][SNP] with an odd pitch: the 64 lanes of one atomic (fixed i: channels 4 hi + i, nodes of 16 rows)
    // spread over the banks (node-major [node][16] put them on 8 banks)
    constexpr int SNP = 16 * NT + 9;
    float* sSeg2 = sSeg1 + 16 * SNP;                                       // d Q2
    if (SEG)
        for (int i = threadIdx.x; i < 2 * 16 * SNP; i += 256) sSeg1[i] = 0.f;
    // SEG: the gather indices / load of the tile's rows are fetched with the tile (a load issued after dq is known would
    // expose a full memory latency per tile)
    // RECOMP with the forward's per-(row, head) log2-sum-exp: a = exp2(s log2(e) / 4 - lse), no max / sum passes
    const bool has_lse = RECOMP && seg.lse != nullptr;
    const float* lsep = has_lse ? seg.lse : rowQ;                          // (always a readable address: loads are unguarded)
    const size_t lse_pitch = has_lse ? 8 : 1;
    const int* seg_first = (SEG && seg.idx_first) ? seg.idx_first : seg.idx_prev;
    const float* seg_load = (SEG && seg.load) ? seg.load : reinterpret_cast<const float*>(seg.idx_prev);
    const size_t seg_lrows = (SEG && seg.load) ? (size_t)seg.load_rows : (size_t)R;
    const int bh = blockIdx.y, b = bh >> 3, h = bh & 7;
    const int split = blockIdx.x;

    // V operand (registers; A operand of V dO^T and B operand of dO V^T): node of position lo, channel 4 kk + hi
    float vop[NT][4];
    unsigned gl[NT];                                               // node of position lo per chunk
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        gl[nt] = (unsigned)(grp_start(nt, lo >> 2, N1) + (lo & 3));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            vop[nt][kk] = Vmat[((size_t)b * N1 + gl[nt]) * ELG_E + h * 16 + 4 * kk + hi];
    }
    // K operand image in LDS (A operand of K^T dS^T): entry (nt, v) of lane (hi, lo) = K[node of (hi, v)][lo],
    // zero for non-owners.  Written by wave 0 .. 3 in slices, read by every wave once per tile.
    for (int e = wave; e < NT * 4; e += 4) {
        const int nt = e >> 2, v = e & 3;
        const int g = grp_start(nt, hi, N1) + v;
        const bool own = g >= 16 * nt + 4 * hi;
        const float x = Kmat[((size_t)b * N1 + g) * ELG_E + h * 16 + lo];
        sK[(nt * 64 + lane) * 4 + v] = own ? x : 0.f;                 // [chunk][lane][k-step]: one ds_read_b128 per chunk
        if (RECOMP) {      // entry (nt, kk): K[node of position lo][4 kk + hi]
            const int g2 = grp_start(nt, lo >> 2, N1) + (lo & 3);
            sK2[(nt * 64 + lane) * 4 + v] = Kmat[((size_t)b * N1 + g2) * ELG_E + h * 16 + 4 * v + hi];
        }
    }
    __syncthreads();
    bool own_lo[NT];                                               // position lo of chunk nt is the node's first visit
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) own_lo[nt] = (int)gl[nt] >= 16 * nt + 4 * (lo >> 2);
    f32x4 dKacc[NT], dVacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { dKacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dVacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    int tile_first = 0, ntile = (R + 15) >> 4;
    if (SEG && seg.tlen) {
        // same live range as pointer_bwd_kernel: decode steps t0 .. max_m tlen[b,m] - 1 of this instance
        __shared__ int sTb;
        if (threadIdx.x == 0) sTb = 0;
        __syncthreads();
        int mx = 0;
        for (int m = threadIdx.x; m < seg.M; m += 256) mx = max(mx, seg.tlen[(size_t)b * seg.M + m]);
        mx = (int)wave_max((float)mx);
        if (lane == 0) atomicMax(&sTb, mx);
        __syncthreads();
        tile_first = live_tile_first(seg.t0, seg.M);
        ntile = (min(R, sTb * seg.M) + 15) >> 4;
    }
    const int per = (max(ntile - tile_first, 0) + splits - 1) / splits;
    const int t_lo = tile_first + split * per, t_hi = min(ntile, t_lo + per);
    const float* Abh = rowA + (size_t)bh * rowA_rows * N1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // ---- tile loads.  No load is guarded: the compiler turns `ok ? load : 0` (and any if / else around loads) into
    // branches with their own vmcnt(0) waits, which serialises the ~45 loads of a tile (27 us per tile instead of one
    // memory latency).  Rows past R are clamped to the last valid row (finite duplicates): they meet zeroed Q / dO
    // operands in dK / dV and an unstored dQ column.  Q / dO rows past R must be exact zeros: 0/1 mask multiply.
    // The loads of tile i + 1 are issued before tile i is consumed (software prefetch, 44 registers).
#define ELG_GB_LOAD(TILE, A1, DOA, OA, DOB, QB, QA, MW, SP, SF, SL, LS)                                           \
    {                                                                                                             \
        const int r0_ = (TILE) << 4;                                                                              \
        const int rleft_ = R - 1 - r0_;                                                                           \
        if (SEG) {                                                                                                \
            const size_t rr_ = (size_t)(r0_ + min(lo, rleft_));                                                   \
            SP = seg.idx_prev[(size_t)b * R + rr_];                                                               \
            SF = seg_first[(size_t)b * R + rr_];                                                                  \
            SL = seg_load[(size_t)b * seg_lrows + rr_];                                                           \
        }                                                                                                         \
        if (!RECOMP) {                                                                                            \
            const float* __restrict__ At = Abh + (size_t)r0_ * N1;                                                \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                       \
                const unsigned off1 = (unsigned)(min(4 * hi + v, rleft_) * N1);                                   \
                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) A1[nt][v] = At[off1 + gl[nt]];                  \
            }                                                                                                     \
        } else {                                                                                                  \
            const unsigned long long* __restrict__ Mt = rowMask + ((size_t)b * rowQ_rows + r0_) * 2;              \
            _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                       \
                const unsigned offm = (unsigned)(min(4 * hi + v, rleft_) * 2);                                    \
                MW[v][0] = Mt[offm]; MW[v][1] = Mt[offm + 1];                                                     \
                LS[v] = lsep[((size_t)b * rowQ_rows + r0_ + min(4 * hi + v, rleft_)) * lse_pitch + (has_lse ? h : 0)]; \
            }                                                                                                     \
            const float* __restrict__ Qa = rowQ + ((size_t)b * rowQ_rows + r0_) * ELG_E + h * 16;                 \
            const unsigned offq = (unsigned)(min(lo, rleft_) * ELG_E + hi);                                       \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) QA[kk] = Qa[offq + 4 * kk];                          \
        }                                                                                                         \
        const float* __restrict__ dOt = dO + ((size_t)b * R + r0_) * ELG_E + h * 16;                              \
        const float* __restrict__ Ot = rowO + ((size_t)b * rowO_rows + r0_) * ELG_E + h * 16;                     \
        const float* __restrict__ Qt = rowQ + ((size_t)b * rowQ_rows + r0_) * ELG_E + h * 16;                     \
        const unsigned offA = (unsigned)(min(lo, rleft_) * ELG_E + hi);                                           \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { DOA[kk] = dOt[offA + 4 * kk]; OA[kk] = Ot[offA + 4 * kk]; } \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                           \
            const int rr = 4 * hi + v;                                                                            \
            const unsigned offB = (unsigned)(min(rr, rleft_) * ELG_E + lo);                                       \
            const float mk = (rr <= rleft_) ? 1.f : 0.f;                                                          \
            DOB[v] = dOt[offB] * mk;                                                                              \
            QB[v] = Qt[offB] * mk;                                                                                \
        }                                                                                                         \
    }
    float a1[NT][4], doA[4], oA[4], doB[4], qB[4], qA[4];
    unsigned long long mw[4][2];
    const int tile0 = t_lo + wave_u;
    int sgp = 0, sgf = 0;
    float sgl = 0.f;
    float lsv[4] = {0.f, 0.f, 0.f, 0.f};
    if (tile0 < t_hi) ELG_GB_LOAD(tile0, a1, doA, oA, doB, qB, qA, mw, sgp, sgf, sgl, lsv)
    for (int tile = tile0; tile < t_hi; tile += 4) {
        const int r0 = tile << 4;                                  // wave-uniform
        const int rT = r0 + lo;                                    // row of this lane in the row-on-lane layout
        float a1n[NT][4], doAn[4], oAn[4], doBn[4], qBn[4], qAn[4];
        unsigned long long mwn[4][2];
        int sgpn = 0, sgfn = 0;
        float sgln = 0.f, lsvn[4] = {0.f, 0.f, 0.f, 0.f};
        {
            const int tn = min(tile + 4, t_hi - 1);                // the last prefetch re-reads a valid tile, unused
            ELG_GB_LOAD(tn, a1n, doAn, oAn, doBn, qBn, qAn, mwn, sgpn, sgfn, sgln, lsvn)
        }
        if (RECOMP && has_lse) {
            const float cs = 0.25f * 1.4426950408889634f;
            // (the K operands of chunk nt + 1 are read from LDS while chunk nt's MFMAs issue: one ds_read_b128 per chunk; with
            // a scalar LDS read in front of every MFMA this block was half of the kernel's time)
            float4 kc = *reinterpret_cast<const float4*>(sK2 + lane * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 kn = *reinterpret_cast<const float4*>(sK2 + ((nt + 1 < NT ? nt + 1 : nt) * 64 + lane) * 4);
                f32x4 S = {0.f, 0.f, 0.f, 0.f};
                S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[0], kc.x, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[1], kc.y, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[2], kc.z, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[3], kc.w, S, 0, 0, 0);
                const unsigned node = gl[nt];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const unsigned long long w = (node < 64u) ? mw[v][0] : mw[v][1];
                    const bool closed = !own_lo[nt] || ((w >> (node & 63u)) & 1ull);
                    a1[nt][v] = closed ? 0.f : __builtin_amdgcn_exp2f(fmaf(S[v], cs, -lsv[v]));
                }
                kc = kn;
            }
        } else if (RECOMP) {
            // a_h[row 4 hi + v][position lo] = softmax over the row's open nodes of q_h . K_h[node] / 4
            float mx[4] = {ELG_NEG_INF, ELG_NEG_INF, ELG_NEG_INF, ELG_NEG_INF};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[kk], sK2[(nt * 64 + lane) * 4 + kk], S, 0, 0, 0);
                const unsigned node = gl[nt];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const unsigned long long w = (node < 64u) ? mw[v][0] : mw[v][1];
                    const bool closed = !own_lo[nt] || ((w >> (node & 63u)) & 1ull);
                    const float x = closed ? ELG_NEG_INF : S[v] * 0.25f;
                    a1[nt][v] = x;
                    mx[v] = fmaxf(mx[v], x);
                }
            }
            float den[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) { mx[v] = row16_max(mx[v]); den[v] = 0.f; }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float e = (a1[nt][v] > ELG_NEG_INF) ? __expf(a1[nt][v] - mx[v]) : 0.f;
                    a1[nt][v] = e;
                    den[v] += e;
                }
#pragma unroll
            for (int v = 0; v < 4; ++v) { den[v] = row16_sum(den[v]); den[v] = den[v] > 0.f ? 1.0f / den[v] : 0.f; }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) a1[nt][v] *= den[v];
        }
        // <dO_h, O_h> per row: partial over this lane's 4 channels, summed over the 4 lane groups
        float doto = doA[0] * oA[0];
        doto = fmaf(doA[1], oA[1], doto); doto = fmaf(doA[2], oA[2], doto); doto = fmaf(doA[3], oA[3], doto);
        doto = quarters_sum(doto);                                // row lo, in every lane
        float dv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) dv[v] = __shfl(doto, 4 * hi + v);   // row 4 hi + v
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 dA = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dA = __builtin_amdgcn_mfma_f32_16x16x4f32(doA[kk], vop[nt][kk], dA, 0, 0, 0);
            float ds1[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                ds1[v] = 0.25f * a1[nt][v] * (dA[v] - dv[v]);                 // position lo, row 4 hi + v
                dKacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds1[v], qB[v], dKacc[nt], 0, 0, 0);
                dVacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[nt][v], doB[v], dVacc[nt], 0, 0, 0);
            }
            // dS with the row on the lane (the B operand of dQ^T = K^T dS^T) is the 16 x 16 transpose of the tile just
            // formed: through per-wave LDS (one ds_write_b128 + four ds_read_b32).  Earlier versions re-read the
            // weights in the second layout (doubled the HBM traffic: PMC 6.7 GB for 3.3 GB of weights) and formed
            // dA a second time (28 more MFMAs per tile).
            // (software-pipelined by one tile: the transposed tile nt - 1 is read back and consumed while tile nt's
            // products issue, so the LDS round trip is off the MFMA dependency chain)
            *reinterpret_cast<float4*>(sTr + (nt & 1) * 320 + lo * 20 + 4 * hi) = make_float4(ds1[0], ds1[1], ds1[2], ds1[3]);
            wave_lds_fence();
            if (nt > 0) {
                const float* pb = sTr + ((nt - 1) & 1) * 320;
                const float4 kq = *reinterpret_cast<const float4*>(sK + ((nt - 1) * 64 + lane) * 4);
                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kq.x, pb[(4 * hi + 0) * 20 + lo], dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kq.y, pb[(4 * hi + 1) * 20 + lo], dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kq.z, pb[(4 * hi + 2) * 20 + lo], dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kq.w, pb[(4 * hi + 3) * 20 + lo], dq, 0, 0, 0);
            }
        }
        {
            const float* pb = sTr + ((NT - 1) & 1) * 320;
            const float4 kq = *reinterpret_cast<const float4*>(sK + ((NT - 1) * 64 + lane) * 4);
            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kq.x, pb[(4 * hi + 0) * 20 + lo], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kq.y, pb[(4 * hi + 1) * 20 + lo], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kq.z, pb[(4 * hi + 2) * 20 + lo], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kq.w, pb[(4 * hi + 3) * 20 + lo], dq, 0, 0, 0);
        }
        wave_lds_fence();
        if (SEG && seg.load) {
            // d wl: every row adds to the same 16 cells -- reduce over the tile's rows first.  The DPP row sums need all
            // lanes active (a disabled lane's register is read as is): rows past R enter with weight 0, outside the branch
            const float lw = rT < R ? sgl : 0.f;
            const float l0 = row16_sum(lw * dq[0]), l1 = row16_sum(lw * dq[1]);
            const float l2 = row16_sum(lw * dq[2]), l3 = row16_sum(lw * dq[3]);
            if (lo == 0) {
                float* pl = sSeg1 + 4 * hi * SNP + 16 * NT;
                atomicAdd(pl, l0); atomicAdd(pl + SNP, l1); atomicAdd(pl + 2 * SNP, l2); atomicAdd(pl + 3 * SNP, l3);
            }
        }
        if (rT < R) {
            if (dQ) *reinterpret_cast<float4*>(dQ + ((size_t)b * R + rT) * ELG_E + h * 16 + 4 * hi) =
                        make_float4(dq[0], dq[1], dq[2], dq[3]);
            if (SEG) {
                float* p1 = sSeg1 + 4 * hi * SNP + sgp;
                atomicAdd(p1, dq[0]); atomicAdd(p1 + SNP, dq[1]); atomicAdd(p1 + 2 * SNP, dq[2]); atomicAdd(p1 + 3 * SNP, dq[3]);
                if (seg.idx_first) {
                    float* p2 = sSeg2 + 4 * hi * SNP + sgf;
                    atomicAdd(p2, dq[0]); atomicAdd(p2 + SNP, dq[1]); atomicAdd(p2 + 2 * SNP, dq[2]); atomicAdd(p2 + 3 * SNP, dq[3]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a1[nt][v] = a1n[nt][v];
            doA[v] = doAn[v]; oA[v] = oAn[v]; doB[v] = doBn[v]; qB[v] = qBn[v];
            qA[v] = qAn[v]; mw[v][0] = mwn[v][0]; mw[v][1] = mwn[v][1];
        }
        sgp = sgpn; sgf = sgfn; sgl = sgln;
#pragma unroll
        for (int v = 0; v < 4; ++v) lsv[v] = lsvn[v];
    }
#undef ELG_GB_LOAD
    if (SEG) {
        __syncthreads();                                            // every wave's LDS atomics are done
        for (int i = threadIdx.x; i < N1 * 16; i += 256) {
            const int n = i >> 4, d = i & 15;
            const float v1 = sSeg1[d * SNP + n];
            if (v1 != 0.f) atomicAdd(seg.dQ1 + ((size_t)b * N1 + n) * ELG_E + h * 16 + d, v1);
            if (seg.dQ2) {
                const float v2 = sSeg2[d * SNP + n];
                if (v2 != 0.f) atomicAdd(seg.dQ2 + ((size_t)b * N1 + n) * ELG_E + h * 16 + d, v2);
            }
        }
        if (seg.dwl && threadIdx.x < 16) atomicAdd(seg.dwl + h * 16 + threadIdx.x, sSeg1[threadIdx.x * SNP + 16 * NT]);
        __syncthreads();                                            // the accumulators are dead: the buffer is reused below
    }
    // ---- sum the four waves' dK_h / dV_h (pairwise through LDS: a buffer for all four at once cost a workgroup per CU) and
    // write this split's partial (B,N1,128) image.  D rows are positions 4 hi + v of chunk nt; position -> node, owners only.
    auto dump = [&](float* dst) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int idx = (16 * nt + 4 * hi + v) * 16 + lo;
                dst[idx] = dKacc[nt][v];
                dst[NT * 256 + idx] = dVacc[nt][v];
            }
    };
    auto addin = [&](const float* src) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int idx = (16 * nt + 4 * hi + v) * 16 + lo;
                dKacc[nt][v] += src[idx];
                dVacc[nt][v] += src[NT * 256 + idx];
            }
    };
    if (wave >= 2) dump(sRed + (size_t)(wave - 2) * (2 * NT * 256));
    __syncthreads();
    if (wave < 2) addin(sRed + (size_t)wave * (2 * NT * 256));
    __syncthreads();
    if (wave == 1) dump(sRed);
    __syncthreads();
    if (wave == 0) addin(sRed);
    __syncthreads();
    if (wave == 0) dump(sRed);
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * NT * 256; i += 256) {
        const float sum = sRed[i];
        const int which = i / (NT * 256), idx = i % (NT * 256);
        const int pos = idx >> 4, d = idx & 15;
        const int nt = pos >> 4, q = (pos >> 2) & 3;
        const int g = grp_start(nt, q, N1) + (pos & 3);
        if (g >= 16 * nt + 4 * q) {                                 // owner (implies g < N1)
            float* out = which ? dVp : dKp;
            if (seg.accumulate) atomicAdd(out + ((size_t)b * N1 + g) * ELG_E + h * 16 + d, sum);
            else out[(((size_t)split * B + b) * N1 + g) * ELG_E + h * 16 + d] = sum;
        }
    }
}


// =============================================================================================
// The same backward with split-bf16 products (elg_decoder_bwd_args.mfma_mode 1 / 2; mask-row mode with the saved
// log2-sum-exp and the gather epilogue = the training path).  A float x is written as x = x1 + x2 (+ x3), every term a
// bf16 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); the residuals are exact in f32): 16 (24) significand bits.
// Every contraction of this kernel has length 16 (head channels, the 16 rows of a tile, the 16 nodes of a chunk), half of
// the K = 32 of v_mfma_f32_16x16x32_bf16, so one instruction carries TWO term products: the lane's 8 k-slots are its 4
// contraction indices twice, once per term.  One operand of a product is packed "single" [a1 | a2], the other "dup"
// [b1 | b1], [b2 | b2]:
//     [a1 | a2] [b1 | b1] + [a1 | a2] [b2 | b2] = (a1 + a2)(b1 + b2)          2 MFMAs = 32 cycles
// against 4 v_mfma_f32_16x16x4_f32 = 128 cycles.  TS = 3 adds [a1 | a3] [b3 | b1] to the score product q K^T (the one whose
// error is amplified by exp2): what is dropped there is 2^-24 of the product.  Operands that live for a whole tile (dO, q
// with the row on the k-slot) or for the whole launch (the K images, in LDS) carry the dup forms, operands formed per chunk
// (a, dS, dS^T) the single form, so no register is copied to build a 4-register operand.
// Differences to the f32 kernel above: nodes are in natural order (position p of chunk nt = node 16 nt + p; nodes past
// N1 - 1 have zero K / V operands and their dK / dV rows are not stored, their weights may be anything finite); the rows'
// closed bits are compacted to one register per row at the head of a tile; the weights of a chunk are recomputed inside
// the chunk loop.
// =============================================================================================
template <int NT, int TS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void glimpse_bwd_bf16_kernel(
    const unsigned long long* __restrict__ rowMask, const float* __restrict__ dO, const float* __restrict__ rowO,
    const float* __restrict__ rowQ, const float* __restrict__ Kmat, const float* __restrict__ Vmat,
    float* __restrict__ dKp, float* __restrict__ dVp, int B, int R, int N1, size_t rowO_rows, size_t rowQ_rows, int splits,
    const GlimpseSeg seg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (seg.T_dev) R = min(R, seg.T_dev[0] * seg.M);                  // the host has not read the rollout's length yet
    constexpr int NP2 = TS >= 3 ? 3 : 2;                              // planes of the score-product K image
    constexpr int SNP = 16 * NT + 9;
    // LDS (words): K image of dq [2][NT][64][4] | K image of q K^T [NP2][NT][64][4] | V image [NT][64][4] | transpose tiles |
    // d Q2 accumulator (TSP).  The cross-wave reduction buffer of the epilogue (4 NT 256) aliases the images.
    unsigned* sKd = reinterpret_cast<unsigned*>(lds);
    unsigned* sK2 = sKd + 2 * NT * 256;
    unsigned* sV = sK2 + NP2 * NT * 256;
    float* sRed = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    float* sTr = lds + (3 + NP2) * NT * 256 + wave * (2 * 320);       // two 16 x 16 transpose tiles (pitch 20)
    float* sSeg2 = lds + (3 + NP2) * NT * 256 + 4 * 2 * 320;           // [16][SNP] d Q2, channel-major (odd pitch: see above)
    const bool tsp = seg.idx_first != nullptr;
    if (tsp)
        for (int i = threadIdx.x; i < 16 * SNP; i += 256) sSeg2[i] = 0.f;
    const int* seg_first = tsp ? seg.idx_first : seg.idx_prev;
    const float* seg_load = seg.load ? seg.load : reinterpret_cast<const float*>(seg.idx_prev);
    const size_t seg_lrows = seg.load ? (size_t)seg.load_rows : (size_t)R;
    const float load_on = seg.load ? 1.f : 0.f;
    const int bh = blockIdx.y, b = bh >> 3, h = bh & 7;
    const int split = blockIdx.x;
    const float cs = 0.25f * 1.4426950408889634f;

    // operand images, one chunk per wave and round.  K (dq: dup, node of (hi, v) x channel lo), K (q K^T: dup, node lo x
    // channel 4 v + hi), V (single, node lo x channel 4 v + hi); nodes past N1 - 1 are zeros
    for (int nt = wave; nt < NT; nt += 4) {
        float kv[4], k2[4], vv[4];
        const int n2 = 16 * nt + lo;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = 16 * nt + 4 * hi + v;
            // (the 1/4 of the softmax backward rides on K here and on q below -- exact; the scores' log2(e)/4 on K)
            kv[v] = n < N1 ? 0.25f * Kmat[((size_t)b * N1 + n) * ELG_E + h * 16 + lo] : 0.f;
            // (TS == 1: the scores of a bf16 forward -- bf16(q) . bf16(K), the scale applied to the f32 sum afterwards, as there)
            k2[v] = n2 < N1 ? (TS == 1 ? 1.f : cs) * Kmat[((size_t)b * N1 + n2) * ELG_E + h * 16 + 4 * v + hi] : 0.f;
            vv[v] = n2 < N1 ? Vmat[((size_t)b * N1 + n2) * ELG_E + h * 16 + 4 * v + hi] : 0.f;
        }
        unsigned p[6];
        bf_terms<2>(kv[0], kv[1], kv[2], kv[3], p);
        *reinterpret_cast<uint4*>(sKd + (nt * 64 + lane) * 4) = make_uint4(p[0], p[1], p[0], p[1]);
        *reinterpret_cast<uint4*>(sKd + ((NT + nt) * 64 + lane) * 4) = make_uint4(p[2], p[3], p[2], p[3]);
        bf_terms<(TS < 2 ? 2 : TS)>(k2[0], k2[1], k2[2], k2[3], p);
        *reinterpret_cast<uint4*>(sK2 + (nt * 64 + lane) * 4) = make_uint4(p[0], p[1], p[0], p[1]);
        *reinterpret_cast<uint4*>(sK2 + ((NT + nt) * 64 + lane) * 4) = make_uint4(p[2], p[3], p[2], p[3]);
        if (TS >= 3) *reinterpret_cast<uint4*>(sK2 + ((2 * NT + nt) * 64 + lane) * 4) = make_uint4(p[4], p[5], p[0], p[1]);
        bf_terms<2>(vv[0], vv[1], vv[2], vv[3], p);
        *reinterpret_cast<uint4*>(sV + (nt * 64 + lane) * 4) = make_uint4(p[0], p[1], p[2], p[3]);
    }
    // accumulators over the wave's tiles: dK_h, dV_h and d Q1_h (rows = nodes 16 nt + 4 hi + v, column = channel lo); d wl
    f32x4 dKacc[NT], dVacc[NT], dGacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        dKacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dVacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dGacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float dwl_acc = 0.f;

    int tile_first = 0, ntile = (R + 15) >> 4;
    __shared__ int sTb;
    if (threadIdx.x == 0) sTb = 0;
    __syncthreads();
    if (seg.tlen) {
        // same live range as pointer_bwd_kernel: decode steps t0 .. max_m tlen[b,m] - 1 of this instance
        int mx = 0;
        for (int m = threadIdx.x; m < seg.M; m += 256) mx = max(mx, seg.tlen[(size_t)b * seg.M + m]);
        mx = (int)wave_max((float)mx);
        if (lane == 0) atomicMax(&sTb, mx);
        __syncthreads();
        tile_first = live_tile_first(seg.t0, seg.M);
        ntile = (min(R, sTb * seg.M) + 15) >> 4;
    }
    const int per = (max(ntile - tile_first, 0) + splits - 1) / splits;
    const int t_lo = tile_first + split * per, t_hi = min(ntile, t_lo + per);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // ---- tile loads (unguarded, rows past R clamped to the last valid row; see the f32 kernel), one tile ahead.  The gather
    // indices / loads are fetched for the rows 4 hi + v of the lane's k-slots.
#define ELG_GBF_LOAD(TILE, DOA, OA, DOB, QB, QA, MW, SP, SF, SL, LS)                                              \
    {                                                                                                             \
        const int r0_ = (TILE) << 4;                                                                              \
        const int rleft_ = R - 1 - r0_;                                                                           \
        const unsigned long long* __restrict__ Mt = rowMask + ((size_t)b * rowQ_rows + r0_) * 2;                  \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                           \
            const int rv_ = min(4 * hi + v, rleft_);                                                              \
            MW[v][0] = Mt[rv_ * 2]; MW[v][1] = Mt[rv_ * 2 + 1];                                                   \
            LS[v] = seg.lse[((size_t)b * rowQ_rows + r0_ + rv_) * 8 + h];                                         \
            SP[v] = seg.idx_prev[(size_t)b * R + r0_ + rv_];                                                      \
            SF[v] = seg_first[(size_t)b * R + r0_ + rv_];                                                         \
            SL[v] = seg_load[(size_t)b * seg_lrows + r0_ + rv_];                                                  \
        }                                                                                                         \
        const float* __restrict__ dOt = dO + ((size_t)b * R + r0_) * ELG_E + h * 16;                              \
        const float* __restrict__ Ot = rowO + ((size_t)b * rowO_rows + r0_) * ELG_E + h * 16;                     \
        const float* __restrict__ Qt = rowQ + ((size_t)b * rowQ_rows + r0_) * ELG_E + h * 16;                     \
        const unsigned offA = (unsigned)(min(lo, rleft_) * ELG_E + hi);                                           \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                        \
            DOA[kk] = dOt[offA + 4 * kk]; OA[kk] = Ot[offA + 4 * kk]; QA[kk] = Qt[offA + 4 * kk];                 \
        }                                                                                                         \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                           \
            const int rr = 4 * hi + v;                                                                            \
            const unsigned offB = (unsigned)(min(rr, rleft_) * ELG_E + lo);                                       \
            const float mk = (rr <= rleft_) ? 1.f : 0.f;                                                          \
            DOB[v] = dOt[offB] * mk;                                                                              \
            QB[v] = Qt[offB] * mk;                                                                                \
        }                                                                                                         \
    }
    float doA[4], oA[4], doB[4], qB[4], qA[4], lsv[4], sgl[4];
    int sgp[4], sgf[4];
    unsigned long long mw[4][2];
    const int tile0 = t_lo + wave_u;
    if (tile0 < t_hi) ELG_GBF_LOAD(tile0, doA, oA, doB, qB, qA, mw, sgp, sgf, sgl, lsv)
    for (int tile = tile0; tile < t_hi; tile += 4) {
        const int r0 = tile << 4;                                  // wave-uniform
        const int rleft = R - 1 - r0;
        float doAn[4], oAn[4], doBn[4], qBn[4], qAn[4], lsvn[4], sgln[4];
        int sgpn[4], sgfn[4];
        unsigned long long mwn[4][2];
        {
            const int tn = min(tile + 4, t_hi - 1);                // the last prefetch re-reads a valid tile, unused
            ELG_GBF_LOAD(tn, doAn, oAn, doBn, qBn, qAn, mwn, sgpn, sgfn, sgln, lsvn)
        }
        // closed bits of this lane's nodes 16 nt + lo, one register per row 4 hi + v: chunk nt at bit (nt >> 1) + 16 (nt & 1)
        unsigned cm[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const unsigned long long x0 = mw[v][0] >> lo, x1 = mw[v][1] >> lo;
            cm[v] = ((unsigned)x0 & 0x10001u) | (((unsigned)(x0 >> 32) & 0x10001u) << 1) | (((unsigned)x1 & 0x10001u) << 2) |
                    (((unsigned)(x1 >> 32) & 0x10001u) << 3);
        }
        // <dO_h, O_h> per row: partial over this lane's 4 channels, summed over the 4 lane groups
        float doto = doA[0] * oA[0];
        doto = fmaf(doA[1], oA[1], doto); doto = fmaf(doA[2], oA[2], doto); doto = fmaf(doA[3], oA[3], doto);
        doto = quarters_sum(doto);                                // row lo, in every lane
        f32x4 ndv;                                                  // - <dO, O> of row 4 hi + v: the accumulator input of dA
#pragma unroll
        for (int v = 0; v < 4; ++v) ndv[v] = -__shfl(doto, 4 * hi + v);
        // per-tile operands: q single (scores), dO / dO^T / q^T dup
        unsigned qt[6];
        bf_terms<(TS < 2 ? 2 : TS)>(qA[0], qA[1], qA[2], qA[3], qt);
        const u32x4 q12 = {qt[0], qt[1], TS == 1 ? 0u : qt[2], TS == 1 ? 0u : qt[3]};
        const u32x4 q13 = {qt[0], qt[1], qt[4], qt[5]};
        u32x4 do1, do2, dob1, dob2, qb1, qb2;
        bf_dup(doA[0], doA[1], doA[2], doA[3], do1, do2);
        bf_dup(doB[0], doB[1], doB[2], doB[3], dob1, dob2);
        bf_dup(0.25f * qB[0], 0.25f * qB[1], 0.25f * qB[2], 0.25f * qB[3], qb1, qb2);
        // dq^T[row 4 hi + v][channel lo] = sum over the nodes of dS[row][node] K[node][channel]
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        auto dq_chunk = [&](int c) {                               // from the transposed dS tile of chunk c
            const float* pb = sTr + (c & 1) * 320;
            const uint4 k1 = *reinterpret_cast<const uint4*>(sKd + (c * 64 + lane) * 4);
            const uint4 k2 = *reinterpret_cast<const uint4*>(sKd + ((NT + c) * 64 + lane) * 4);
            const u32x4 tq = bf_single(pb[(4 * hi + 0) * 20 + lo], pb[(4 * hi + 1) * 20 + lo], pb[(4 * hi + 2) * 20 + lo],
                                       pb[(4 * hi + 3) * 20 + lo]);
            dq = mfma_bf(tq, u32x4{k1.x, k1.y, k1.z, k1.w}, dq);
            dq = mfma_bf(tq, u32x4{k2.x, k2.y, k2.z, k2.w}, dq);
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // a_h[row 4 hi + v][node 16 nt + lo] = closed ? 0 : exp2(q_h . K_h[node] log2(e) / 4 - lse)
            const uint4 ka = *reinterpret_cast<const uint4*>(sK2 + (nt * 64 + lane) * 4);
            const uint4 kb = *reinterpret_cast<const uint4*>(sK2 + ((NT + nt) * 64 + lane) * 4);
            const uint4 vw = *reinterpret_cast<const uint4*>(sV + (nt * 64 + lane) * 4);
            const u32x4 vq = {vw.x, vw.y, vw.z, vw.w};
            // accumulator inputs: - lse (or - inf for a closed node: exp2 -> exactly 0) for the scores, - <dO, O> for dA
            const int bit = (nt >> 1) + 16 * (nt & 1);                 // (compile-time after unrolling)
            f32x4 S, dA = ndv;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const unsigned closed = (unsigned)((int)(cm[v] << (31 - bit)) >> 31);    // ~0 for a closed node, 0 for an open one
                S[v] = __uint_as_float((closed & 0xff800000u) | (~closed & __float_as_uint(TS == 1 ? 0.f : -lsv[v])));
            }
            S = mfma_bf(q12, u32x4{ka.x, ka.y, ka.z, ka.w}, S);
            dA = mfma_bf(do1, vq, dA);
            if (TS >= 2) S = mfma_bf(q12, u32x4{kb.x, kb.y, kb.z, kb.w}, S);
            dA = mfma_bf(do2, vq, dA);
            if (TS >= 3) {
                const uint4 kc = *reinterpret_cast<const uint4*>(sK2 + ((2 * NT + nt) * 64 + lane) * 4);
                S = mfma_bf(q13, u32x4{kc.x, kc.y, kc.z, kc.w}, S);
            }
            float aw[4], ds1[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                // (argument clamped at 0: a live row has s log2(e)/4 - lse <= 0; a row the rollout did not write at this step keeps q / lse
                //  of an earlier batch, its exp2 could overflow and inf * 0 -- its dA and dO are 0 -- would poison dK / dV / dQ)
                aw[v] = __builtin_amdgcn_exp2f(fminf(TS == 1 ? fmaf(S[v], cs, -lsv[v]) : S[v], 0.f));
                ds1[v] = aw[v] * dA[v];                            // 4 dS: node 16 nt + lo, row 4 hi + v
            }
            const u32x4 dsq = bf_single(ds1[0], ds1[1], ds1[2], ds1[3]);
            const u32x4 aq = bf_single(aw[0], aw[1], aw[2], aw[3]);
            dKacc[nt] = mfma_bf(dsq, qb1, dKacc[nt]);
            dVacc[nt] = mfma_bf(aq, dob1, dVacc[nt]);
            dKacc[nt] = mfma_bf(dsq, qb2, dKacc[nt]);
            dVacc[nt] = mfma_bf(aq, dob2, dVacc[nt]);
            // dS with the row on the lane: transposed through per-wave LDS, consumed one chunk later so the round trip is off
            // the dependency chain
            *reinterpret_cast<float4*>(sTr + (nt & 1) * 320 + lo * 20 + 4 * hi) = make_float4(ds1[0], ds1[1], ds1[2], ds1[3]);
            wave_lds_fence();
            if (nt > 0) dq_chunk(nt - 1);
        }
        dq_chunk(NT - 1);
        wave_lds_fence();
        // ---- the query-gather backward of the tile: d Q1[node] += sum over the rows gathered at that node of dq[row], as a
        // product with the one-hot matrix P[node][row] = [prev(row) = node] (exact in bf16, so one instruction per chunk:
        // [P | P] [dq1 | dq2]); LDS float atomics on a (node, channel) accumulator took 31 % of this kernel.  Rows past R: -1.
        {
            const u32x4 dqs = bf_single(dq[0], dq[1], dq[2], dq[3]);
            int pv[4];
            float lw = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bool live = 4 * hi + v <= rleft;
                pv[v] = live ? sgp[v] - lo : -1;                                   // == 16 nt  <=>  prev(row) == 16 nt + lo
                lw = fmaf(live ? sgl[v] : 0.f, dq[v], lw);
            }
            dwl_acc = fmaf(load_on, lw, dwl_acc);                                   // d wl[channel lo], this lane group's rows
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned e0 = pv[0] == 16 * nt ? 0x3f80u : 0u, e1 = pv[1] == 16 * nt ? 0x3f800000u : 0u;
                const unsigned e2 = pv[2] == 16 * nt ? 0x3f80u : 0u, e3 = pv[3] == 16 * nt ? 0x3f800000u : 0u;
                const unsigned w0 = e0 | e1, w1 = e2 | e3;
                dGacc[nt] = mfma_bf(u32x4{w0, w1, w0, w1}, dqs, dGacc[nt]);
            }
            if (tsp) {                                                              // d Q2: the tour's first node (LDS atomics)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (4 * hi + v <= rleft) atomicAdd(sSeg2 + lo * SNP + sgf[v], dq[v]);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            doA[v] = doAn[v]; oA[v] = oAn[v]; doB[v] = doBn[v]; qB[v] = qBn[v]; qA[v] = qAn[v]; lsv[v] = lsvn[v];
            mw[v][0] = mwn[v][0]; mw[v][1] = mwn[v][1];
            sgp[v] = sgpn[v]; sgf[v] = sgfn[v]; sgl[v] = sgln[v];
        }
    }
#undef ELG_GBF_LOAD
    // d wl[h 16 + lo]: the four lane groups hold disjoint rows
    if (seg.dwl) {
        const float t = quarters_sum(dwl_acc);
        if (hi == 0 && t != 0.f) atomicAdd(seg.dwl + h * 16 + lo, t);
    }
    __syncthreads();                                                // every wave's LDS atomics are done, the images are dead
    if (tsp)
        for (int i = threadIdx.x; i < N1 * 16; i += 256) {
            const int n = i >> 4, d = i & 15;
            const float v2 = sSeg2[d * SNP + n];
            if (v2 != 0.f) atomicAdd(seg.dQ2 + ((size_t)b * N1 + n) * ELG_E + h * 16 + d, v2);
        }
    // ---- sum the four waves' accumulators pairwise through LDS (over the dead images): first (dK_h, dV_h), then d Q1_h.
    // D rows are nodes 16 nt + 4 hi + v.
    auto reduce_pair = [&](f32x4 (&X)[NT], f32x4 (&Y)[NT]) {
        auto dump = [&](float* dst) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int idx = (16 * nt + 4 * hi + v) * 16 + lo;
                    dst[idx] = X[nt][v];
                    dst[NT * 256 + idx] = Y[nt][v];
                }
        };
        auto addin = [&](const float* src) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int idx = (16 * nt + 4 * hi + v) * 16 + lo;
                    X[nt][v] += src[idx];
                    Y[nt][v] += src[NT * 256 + idx];
                }
        };
        if (wave >= 2) dump(sRed + (size_t)(wave - 2) * (2 * NT * 256));
        __syncthreads();
        if (wave < 2) addin(sRed + (size_t)wave * (2 * NT * 256));
        __syncthreads();
        if (wave == 1) dump(sRed);
        __syncthreads();
        if (wave == 0) addin(sRed);
        __syncthreads();
        if (wave == 0) dump(sRed);
        __syncthreads();
    };
    reduce_pair(dKacc, dVacc);
    for (int i = threadIdx.x; i < 2 * NT * 256; i += 256) {
        const float sum = sRed[i];
        const int which = i / (NT * 256), idx = i % (NT * 256);
        const int n = idx >> 4, d = idx & 15;
        if (n < N1) {
            float* out = which ? dVp : dKp;
            if (seg.accumulate) atomicAdd(out + ((size_t)b * N1 + n) * ELG_E + h * 16 + d, sum);
            else out[(((size_t)split * B + b) * N1 + n) * ELG_E + h * 16 + d] = sum;
        }
    }
    __syncthreads();
    reduce_pair(dGacc, dVacc);                                      // (the second array rides along, unused)
    for (int i = threadIdx.x; i < NT * 256; i += 256) {
        const float sum = sRed[i];
        const int n = i >> 4, d = i & 15;
        if (n < N1 && sum != 0.f) atomicAdd(seg.dQ1 + ((size_t)b * N1 + n) * ELG_E + h * 16 + d, sum);
    }
}

// =============================================================================================
// The f32 kernel of the training path (mfma_mode 0; mask rows + saved log2-sum-exp + gather epilogue), restructured like the
// split-bf16 kernel above -- nodes in natural order, the rows' closed bits compacted to one register per row, the weights of a
// chunk recomputed inside the chunk loop, - lse (or - inf for a closed node) and - <dO, o> as accumulator inputs, the V operand
// in LDS, dq produced row-major so that the query-gather backward is a one-hot product instead of LDS float atomics -- but with
// every arithmetic product on v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation).
// =============================================================================================
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void glimpse_bwd_f32n_kernel(
    const unsigned long long* __restrict__ rowMask, const float* __restrict__ dO, const float* __restrict__ rowO,
    const float* __restrict__ rowQ, const float* __restrict__ Kmat, const float* __restrict__ Vmat,
    float* __restrict__ dKp, float* __restrict__ dVp, int B, int R, int N1, size_t rowO_rows, size_t rowQ_rows, int splits,
    const GlimpseSeg seg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (seg.T_dev) R = min(R, seg.T_dev[0] * seg.M);                  // the host has not read the rollout's length yet
    constexpr int SNP = 16 * NT + 9;
    // LDS (floats): K image of dq [NT][64][4] | K image of q K^T [NT][64][4] | V image [NT][64][4] | (pad to 4 NT 256: the
    // cross-wave reduction buffer of the epilogue aliases the images) | transpose tiles | d Q2 accumulator of the TSP decoder
    float* sKd = lds;
    float* sK2 = sKd + NT * 256;
    float* sV = sK2 + NT * 256;
    float* sRed = lds;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    float* sTr = lds + 4 * NT * 256 + wave * (2 * 320);       // two 16 x 16 transpose tiles (pitch 20)
    float* sSeg2 = lds + 4 * NT * 256 + 4 * 2 * 320;           // [16][SNP] d Q2, channel-major (odd pitch: see above)
    const bool tsp = seg.idx_first != nullptr;
    if (tsp)
        for (int i = threadIdx.x; i < 16 * SNP; i += 256) sSeg2[i] = 0.f;
    const int* seg_first = tsp ? seg.idx_first : seg.idx_prev;
    const float* seg_load = seg.load ? seg.load : reinterpret_cast<const float*>(seg.idx_prev);
    const size_t seg_lrows = seg.load ? (size_t)seg.load_rows : (size_t)R;
    const float load_on = seg.load ? 1.f : 0.f;
    const int bh = blockIdx.y, b = bh >> 3, h = bh & 7;
    const int split = blockIdx.x;
    const float cs = 0.25f * 1.4426950408889634f;

    // operand images, one chunk per wave and round.  K (dq: node of (hi, v) x channel lo), K (q K^T: node lo x channel 4 v + hi),
    // V (node lo x channel 4 v + hi); nodes past N1 - 1 are zeros
    for (int nt = wave; nt < NT; nt += 4) {
        float kv[4], k2[4], vv[4];
        const int n2 = 16 * nt + lo;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = 16 * nt + 4 * hi + v;
            // (the 1/4 of the softmax backward rides on K here and on q below -- exact; the scores' log2(e)/4 on K)
            kv[v] = n < N1 ? 0.25f * Kmat[((size_t)b * N1 + n) * ELG_E + h * 16 + lo] : 0.f;
            k2[v] = n2 < N1 ? cs * Kmat[((size_t)b * N1 + n2) * ELG_E + h * 16 + 4 * v + hi] : 0.f;
            vv[v] = n2 < N1 ? Vmat[((size_t)b * N1 + n2) * ELG_E + h * 16 + 4 * v + hi] : 0.f;
        }
        *reinterpret_cast<float4*>(sKd + (nt * 64 + lane) * 4) = make_float4(kv[0], kv[1], kv[2], kv[3]);
        *reinterpret_cast<float4*>(sK2 + (nt * 64 + lane) * 4) = make_float4(k2[0], k2[1], k2[2], k2[3]);
        *reinterpret_cast<float4*>(sV + (nt * 64 + lane) * 4) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    // accumulators over the wave's tiles: dK_h, dV_h and d Q1_h (rows = nodes 16 nt + 4 hi + v, column = channel lo); d wl
    f32x4 dKacc[NT], dVacc[NT], dGacc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        dKacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dVacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dGacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float dwl_acc = 0.f;

    int tile_first = 0, ntile = (R + 15) >> 4;
    __shared__ int sTb;
    if (threadIdx.x == 0) sTb = 0;
    __syncthreads();
    if (seg.tlen) {
        // same live range as pointer_bwd_kernel: decode steps t0 .. max_m tlen[b,m] - 1 of this instance
        int mx = 0;
        for (int m = threadIdx.x; m < seg.M; m += 256) mx = max(mx, seg.tlen[(size_t)b * seg.M + m]);
        mx = (int)wave_max((float)mx);
        if (lane == 0) atomicMax(&sTb, mx);
        __syncthreads();
        tile_first = live_tile_first(seg.t0, seg.M);
        ntile = (min(R, sTb * seg.M) + 15) >> 4;
    }
    const int per = (max(ntile - tile_first, 0) + splits - 1) / splits;
    const int t_lo = tile_first + split * per, t_hi = min(ntile, t_lo + per);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // ---- tile loads (unguarded, rows past R clamped to the last valid row; see the f32 kernel), one tile ahead.  The gather
    // indices / loads are fetched for the rows 4 hi + v of the lane's k-slots.
#define ELG_GBF_LOAD(TILE, DOA, OA, DOB, QB, QA, MW, SP, SF, SL, LS)                                              \
    {                                                                                                             \
        const int r0_ = (TILE) << 4;                                                                              \
        const int rleft_ = R - 1 - r0_;                                                                           \
        const unsigned long long* __restrict__ Mt = rowMask + ((size_t)b * rowQ_rows + r0_) * 2;                  \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                           \
            const int rv_ = min(4 * hi + v, rleft_);                                                              \
            MW[v][0] = Mt[rv_ * 2]; MW[v][1] = Mt[rv_ * 2 + 1];                                                   \
            LS[v] = seg.lse[((size_t)b * rowQ_rows + r0_ + rv_) * 8 + h];                                         \
            SP[v] = seg.idx_prev[(size_t)b * R + r0_ + rv_];                                                      \
            SF[v] = seg_first[(size_t)b * R + r0_ + rv_];                                                         \
            SL[v] = seg_load[(size_t)b * seg_lrows + r0_ + rv_];                                                  \
        }                                                                                                         \
        const float* __restrict__ dOt = dO + ((size_t)b * R + r0_) * ELG_E + h * 16;                              \
        const float* __restrict__ Ot = rowO + ((size_t)b * rowO_rows + r0_) * ELG_E + h * 16;                     \
        const float* __restrict__ Qt = rowQ + ((size_t)b * rowQ_rows + r0_) * ELG_E + h * 16;                     \
        const unsigned offA = (unsigned)(min(lo, rleft_) * ELG_E + hi);                                           \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                        \
            DOA[kk] = dOt[offA + 4 * kk]; OA[kk] = Ot[offA + 4 * kk]; QA[kk] = Qt[offA + 4 * kk];                 \
        }                                                                                                         \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                           \
            const int rr = 4 * hi + v;                                                                            \
            const unsigned offB = (unsigned)(min(rr, rleft_) * ELG_E + lo);                                       \
            const float mk = (rr <= rleft_) ? 1.f : 0.f;                                                          \
            DOB[v] = dOt[offB] * mk;                                                                              \
            QB[v] = Qt[offB] * mk;                                                                                \
        }                                                                                                         \
    }
    float doA[4], oA[4], doB[4], qB[4], qA[4], lsv[4], sgl[4];
    int sgp[4], sgf[4];
    unsigned long long mw[4][2];
    const int tile0 = t_lo + wave_u;
    if (tile0 < t_hi) ELG_GBF_LOAD(tile0, doA, oA, doB, qB, qA, mw, sgp, sgf, sgl, lsv)
    for (int tile = tile0; tile < t_hi; tile += 4) {
        const int r0 = tile << 4;                                  // wave-uniform
        const int rleft = R - 1 - r0;
        float doAn[4], oAn[4], doBn[4], qBn[4], qAn[4], lsvn[4], sgln[4];
        int sgpn[4], sgfn[4];
        unsigned long long mwn[4][2];
        {
            const int tn = min(tile + 4, t_hi - 1);                // the last prefetch re-reads a valid tile, unused
            ELG_GBF_LOAD(tn, doAn, oAn, doBn, qBn, qAn, mwn, sgpn, sgfn, sgln, lsvn)
        }
        // closed bits of this lane's nodes 16 nt + lo, one register per row 4 hi + v: chunk nt at bit (nt >> 1) + 16 (nt & 1)
        unsigned cm[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const unsigned long long x0 = mw[v][0] >> lo, x1 = mw[v][1] >> lo;
            cm[v] = ((unsigned)x0 & 0x10001u) | (((unsigned)(x0 >> 32) & 0x10001u) << 1) | (((unsigned)x1 & 0x10001u) << 2) |
                    (((unsigned)(x1 >> 32) & 0x10001u) << 3);
        }
        // <dO_h, O_h> per row: partial over this lane's 4 channels, summed over the 4 lane groups
        float doto = doA[0] * oA[0];
        doto = fmaf(doA[1], oA[1], doto); doto = fmaf(doA[2], oA[2], doto); doto = fmaf(doA[3], oA[3], doto);
        doto = quarters_sum(doto);                                // row lo, in every lane
        f32x4 ndv;                                                  // - <dO, O> of row 4 hi + v: the accumulator input of dA
#pragma unroll
        for (int v = 0; v < 4; ++v) ndv[v] = -__shfl(doto, 4 * hi + v);
        // (the 1/4 of the softmax backward rides on q^T here and on the K image of dq -- exact)
        const float qb[4] = {0.25f * qB[0], 0.25f * qB[1], 0.25f * qB[2], 0.25f * qB[3]};
        // dq^T[row 4 hi + v][channel lo] = sum over the nodes of dS[row][node] K[node][channel]
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        auto dq_chunk = [&](int c) {                               // from the transposed dS tile of chunk c
            const float* pb = sTr + (c & 1) * 320;
            const float4 k1 = *reinterpret_cast<const float4*>(sKd + (c * 64 + lane) * 4);
            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[(4 * hi + 0) * 20 + lo], k1.x, dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[(4 * hi + 1) * 20 + lo], k1.y, dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[(4 * hi + 2) * 20 + lo], k1.z, dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[(4 * hi + 3) * 20 + lo], k1.w, dq, 0, 0, 0);
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // a_h[row 4 hi + v][node 16 nt + lo] = closed ? 0 : exp2(q_h . K_h[node] log2(e) / 4 - lse)
            const float4 ka = *reinterpret_cast<const float4*>(sK2 + (nt * 64 + lane) * 4);
            const float4 vw = *reinterpret_cast<const float4*>(sV + (nt * 64 + lane) * 4);
            // accumulator inputs: - lse (or - inf for a closed node: exp2 -> exactly 0) for the scores, - <dO, O> for dA
            const int bit = (nt >> 1) + 16 * (nt & 1);                 // (compile-time after unrolling)
            f32x4 S, dA = ndv;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const unsigned closed = (unsigned)((int)(cm[v] << (31 - bit)) >> 31);    // ~0 for a closed node, 0 for an open one
                S[v] = __uint_as_float((closed & 0xff800000u) | (~closed & __float_as_uint(-lsv[v])));
            }
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[0], ka.x, S, 0, 0, 0);
            dA = __builtin_amdgcn_mfma_f32_16x16x4f32(doA[0], vw.x, dA, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[1], ka.y, S, 0, 0, 0);
            dA = __builtin_amdgcn_mfma_f32_16x16x4f32(doA[1], vw.y, dA, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[2], ka.z, S, 0, 0, 0);
            dA = __builtin_amdgcn_mfma_f32_16x16x4f32(doA[2], vw.z, dA, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA[3], ka.w, S, 0, 0, 0);
            dA = __builtin_amdgcn_mfma_f32_16x16x4f32(doA[3], vw.w, dA, 0, 0, 0);
            float aw[4], ds1[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                aw[v] = __builtin_amdgcn_exp2f(fminf(S[v], 0.f));   // (clamped: rows of an earlier batch must stay finite, see the bf16 kernel)
                ds1[v] = aw[v] * dA[v];                            // 4 dS: node 16 nt + lo, row 4 hi + v
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                dKacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds1[v], qb[v], dKacc[nt], 0, 0, 0);
                dVacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[v], doB[v], dVacc[nt], 0, 0, 0);
            }
            // dS with the row on the lane: transposed through per-wave LDS, consumed one chunk later so the round trip is off
            // the dependency chain
            *reinterpret_cast<float4*>(sTr + (nt & 1) * 320 + lo * 20 + 4 * hi) = make_float4(ds1[0], ds1[1], ds1[2], ds1[3]);
            wave_lds_fence();
            if (nt > 0) dq_chunk(nt - 1);
        }
        dq_chunk(NT - 1);
        wave_lds_fence();
        // ---- the query-gather backward of the tile: d Q1[node] += sum over the rows gathered at that node of dq[row], as a
        // product with the one-hot matrix P[node][row] = [prev(row) = node] on v_mfma_f32_16x16x32_bf16.  This is a SCATTER, not
        // reduced-precision arithmetic: P is 0 / 1 (exact in bf16), dq enters as its three bf16 terms dq1 + dq2 + dq3, which
        // reproduce the f32 value to 2^-27 (elg_bf16.h), every product is exact and the accumulation is f32 -- the same result,
        // to f32 rounding, as the LDS float atomics on a (node, channel) accumulator that took 31 % of the kernel before.
        // [P | P] [dq1 | dq2] + [P | 0] [dq3 | 0]: two instructions (32 cycles) per chunk.  Rows past R: -1.
        {
            unsigned dt[6];
            bf_terms<3>(dq[0], dq[1], dq[2], dq[3], dt);
            const u32x4 dqs = {dt[0], dt[1], dt[2], dt[3]};
            const u32x4 dq3 = {dt[4], dt[5], 0u, 0u};
            int pv[4];
            float lw = 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const bool live = 4 * hi + v <= rleft;
                pv[v] = live ? sgp[v] - lo : -1;                                   // == 16 nt  <=>  prev(row) == 16 nt + lo
                lw = fmaf(live ? sgl[v] : 0.f, dq[v], lw);
            }
            dwl_acc = fmaf(load_on, lw, dwl_acc);                                   // d wl[channel lo], this lane group's rows
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned e0 = pv[0] == 16 * nt ? 0x3f80u : 0u, e1 = pv[1] == 16 * nt ? 0x3f800000u : 0u;
                const unsigned e2 = pv[2] == 16 * nt ? 0x3f80u : 0u, e3 = pv[3] == 16 * nt ? 0x3f800000u : 0u;
                const unsigned w0 = e0 | e1, w1 = e2 | e3;
                dGacc[nt] = mfma_bf(u32x4{w0, w1, w0, w1}, dqs, dGacc[nt]);
                dGacc[nt] = mfma_bf(u32x4{w0, w1, 0u, 0u}, dq3, dGacc[nt]);
            }
            if (tsp) {                                                              // d Q2: the tour's first node (LDS atomics)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (4 * hi + v <= rleft) atomicAdd(sSeg2 + lo * SNP + sgf[v], dq[v]);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            doA[v] = doAn[v]; oA[v] = oAn[v]; doB[v] = doBn[v]; qB[v] = qBn[v]; qA[v] = qAn[v]; lsv[v] = lsvn[v];
            mw[v][0] = mwn[v][0]; mw[v][1] = mwn[v][1];
            sgp[v] = sgpn[v]; sgf[v] = sgfn[v]; sgl[v] = sgln[v];
        }
    }
#undef ELG_GBF_LOAD
    // d wl[h 16 + lo]: the four lane groups hold disjoint rows
    if (seg.dwl) {
        const float t = quarters_sum(dwl_acc);
        if (hi == 0 && t != 0.f) atomicAdd(seg.dwl + h * 16 + lo, t);
    }
    __syncthreads();                                                // every wave's LDS atomics are done, the images are dead
    if (tsp)
        for (int i = threadIdx.x; i < N1 * 16; i += 256) {
            const int n = i >> 4, d = i & 15;
            const float v2 = sSeg2[d * SNP + n];
            if (v2 != 0.f) atomicAdd(seg.dQ2 + ((size_t)b * N1 + n) * ELG_E + h * 16 + d, v2);
        }
    // ---- sum the four waves' accumulators pairwise through LDS (over the dead images): first (dK_h, dV_h), then d Q1_h.
    // D rows are nodes 16 nt + 4 hi + v.
    auto reduce_pair = [&](f32x4 (&X)[NT], f32x4 (&Y)[NT]) {
        auto dump = [&](float* dst) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int idx = (16 * nt + 4 * hi + v) * 16 + lo;
                    dst[idx] = X[nt][v];
                    dst[NT * 256 + idx] = Y[nt][v];
                }
        };
        auto addin = [&](const float* src) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int idx = (16 * nt + 4 * hi + v) * 16 + lo;
                    X[nt][v] += src[idx];
                    Y[nt][v] += src[NT * 256 + idx];
                }
        };
        if (wave >= 2) dump(sRed + (size_t)(wave - 2) * (2 * NT * 256));
        __syncthreads();
        if (wave < 2) addin(sRed + (size_t)wave * (2 * NT * 256));
        __syncthreads();
        if (wave == 1) dump(sRed);
        __syncthreads();
        if (wave == 0) addin(sRed);
        __syncthreads();
        if (wave == 0) dump(sRed);
        __syncthreads();
    };
    reduce_pair(dKacc, dVacc);
    for (int i = threadIdx.x; i < 2 * NT * 256; i += 256) {
        const float sum = sRed[i];
        const int which = i / (NT * 256), idx = i % (NT * 256);
        const int n = idx >> 4, d = idx & 15;
        if (n < N1) {
            float* out = which ? dVp : dKp;
            if (seg.accumulate) atomicAdd(out + ((size_t)b * N1 + n) * ELG_E + h * 16 + d, sum);
            else out[(((size_t)split * B + b) * N1 + n) * ELG_E + h * 16 + d] = sum;
        }
    }
    __syncthreads();
    reduce_pair(dGacc, dVacc);                                      // (the second array rides along, unused)
    for (int i = threadIdx.x; i < NT * 256; i += 256) {
        const float sum = sRed[i];
        const int n = i >> 4, d = i & 15;
        if (n < N1 && sum != 0.f) atomicAdd(seg.dQ1 + ((size_t)b * N1 + n) * ELG_E + h * 16 + d, sum);
    }
}

template <int NT>
static int launch_glimpse_bwd_f32n(const unsigned long long* rowMask, const float* dO, const float* rowO, const float* rowQ,
                                   const float* Kmat, const float* Vmat, float* dKp, float* dVp, int B, int R, int N1, size_t ro,
                                   size_t rq, int splits, const GlimpseSeg& seg, hipStream_t stream) {
    const size_t lds = (size_t)(4 * NT * 256 + 4 * 2 * 320 + 16 * (16 * NT + 9)) * sizeof(float);
    auto kern = glimpse_bwd_f32n_kernel<NT>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, "glimpse_bwd (f32): hipFuncSetAttribute failed");
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(splits, B * 8), dim3(256), lds, stream, rowMask, dO, rowO, rowQ, Kmat, Vmat, dKp, dVp, B, R, N1,
                       ro, rq, splits, seg);
    return launch_status("glimpse_bwd_f32n");
}

template <int NT, int TS>
static int launch_glimpse_bwd_bf16(const unsigned long long* rowMask, const float* dO, const float* rowO, const float* rowQ,
                                   const float* Kmat, const float* Vmat, float* dKp, float* dVp, int B, int R, int N1, size_t ro,
                                   size_t rq, int splits, const GlimpseSeg& seg, hipStream_t stream) {
    constexpr int NP2 = TS >= 3 ? 3 : 2;
    const size_t lds = (size_t)((3 + NP2) * NT * 256 + 4 * 2 * 320 + 16 * (16 * NT + 9)) * sizeof(float);
    auto kern = glimpse_bwd_bf16_kernel<NT, TS>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, "glimpse_bwd (split-bf16): hipFuncSetAttribute failed");
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(splits, B * 8), dim3(256), lds, stream, rowMask, dO, rowO, rowQ, Kmat, Vmat, dKp, dVp, B, R, N1,
                       ro, rq, splits, seg);
    return launch_status("glimpse_bwd_bf16");
}

template <int NT, bool RECOMP, bool SEG>
static int launch_glimpse_bwd_mfma_t(const float* rowA, const unsigned long long* rowMask, const float* dO, const float* rowO,
                                   const float* rowQ, const float* Kmat, const float* Vmat, float* dQ, float* dKp,
                                   float* dVp, int B, int R, int N1, size_t ra, size_t ro, size_t rq, int splits,
                                   const GlimpseSeg& seg, hipStream_t stream) {
    const size_t lds = (size_t)(NT * 256 + 2 * 2 * NT * 256 + 4 * 2 * 320 + (RECOMP ? NT * 256 : 0)) * sizeof(float);
    auto kern = glimpse_bwd_mfma_kernel<NT, RECOMP, SEG>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), (NT * 256 * 10 + 4 * 2 * 320 + (32 * NT + 1) * 16) * sizeof(float)))
        return fail(ELG_ELAUNCH, "glimpse_bwd_fused: hipFuncSetAttribute failed");
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(splits, B * 8), dim3(256), lds, stream, rowA, rowMask, dO, rowO, rowQ, Kmat, Vmat, dQ,
                       dKp, dVp, B, R, N1, ra, ro, rq, splits, seg);
    return launch_status("glimpse_bwd_fused");
}

}  // namespace elg

using namespace elg;

namespace elg {
template <int NT, bool RECOMP>
static int launch_glimpse_bwd_mfma(const float* rowA, const unsigned long long* rowMask, const float* dO, const float* rowO,
                                   const float* rowQ, const float* Kmat, const float* Vmat, float* dQ, float* dKp,
                                   float* dVp, int B, int R, int N1, size_t ra, size_t ro, size_t rq, int splits,
                                   const GlimpseSeg& seg, hipStream_t stream) {
    if (seg.idx_prev)
        return launch_glimpse_bwd_mfma_t<NT, RECOMP, true>(rowA, rowMask, dO, rowO, rowQ, Kmat, Vmat, dQ, dKp, dVp, B, R, N1, ra,
                                                           ro, rq, splits, seg, stream);
    return launch_glimpse_bwd_mfma_t<NT, RECOMP, false>(rowA, rowMask, dO, rowO, rowQ, Kmat, Vmat, dQ, dKp, dVp, B, R, N1, ra, ro,
                                                        rq, splits, seg, stream);
}

int glimpse_bwd_launch(const float* rowA, const unsigned long long* mk, const float* dO, const float* rowO, const float* rowQ,
                       const float* Kmat, const float* Vmat, float* dQ, float* dK_part, float* dV_part, int B, int R, int N1,
                       long long rowA_rows, long long rowO_rows, long long rowQ_rows, int splits, const GlimpseSeg& seg,
                       hipStream_t s) {
    if ((!mk && rowA_rows < R) || rowO_rows < R || rowQ_rows < R)
        return fail(ELG_EINVAL, "glimpse_bwd_fused: row strides smaller than R");
    if (!rowA && !mk) return fail(ELG_EINVAL, "glimpse_bwd_fused: neither weights nor mask rows given");
    if (B <= 0 || R <= 0 || N1 < 4 || splits <= 0) return fail(ELG_EINVAL, "glimpse_bwd_fused: bad sizes");
    if (N1 > 128) return fail(ELG_ENOTIMPL, "glimpse_bwd_fused: N1 > 128 not built (use elg_glimpse_rows_bwd)");
    if (!dQ && !seg.idx_prev) return fail(ELG_EINVAL, "glimpse_bwd_fused: neither dQ nor the gather epilogue requested");
    const int nt = (N1 + 15) / 16;
    static const bool old_f32 = getenv("ELG_GLIMPSE_F32_OLD") != nullptr;      // A/B timing against the previous f32 kernel
    if (seg.mfma_mode == 0 && mk && seg.lse && seg.idx_prev && !dQ && !old_f32) {
#define ELG_GBN(NT) return launch_glimpse_bwd_f32n<NT>(mk, dO, rowO, rowQ, Kmat, Vmat, dK_part, dV_part, B, R, N1, (size_t)rowO_rows, \
                                                        (size_t)rowQ_rows, splits, seg, s);
        if (nt <= 2) ELG_GBN(2)
        if (nt <= 4) ELG_GBN(4)
        if (nt <= 7) ELG_GBN(7)
        ELG_GBN(8)
#undef ELG_GBN
    }
    if (seg.mfma_mode > 0 && mk && seg.lse && seg.idx_prev && !dQ) {
        // split-bf16 products (mask rows + saved normaliser + gather epilogue: the training path)
#define ELG_GBF(NT)                                                                                                         \
    {                                                                                                                       \
        if (seg.mfma_mode == 3)                                                                                             \
            return launch_glimpse_bwd_bf16<NT, 1>(mk, dO, rowO, rowQ, Kmat, Vmat, dK_part, dV_part, B, R, N1, (size_t)rowO_rows, \
                                                  (size_t)rowQ_rows, splits, seg, s);                                       \
        if (seg.mfma_mode == 1)                                                                                             \
            return launch_glimpse_bwd_bf16<NT, 2>(mk, dO, rowO, rowQ, Kmat, Vmat, dK_part, dV_part, B, R, N1, (size_t)rowO_rows, \
                                                  (size_t)rowQ_rows, splits, seg, s);                                       \
        return launch_glimpse_bwd_bf16<NT, 3>(mk, dO, rowO, rowQ, Kmat, Vmat, dK_part, dV_part, B, R, N1, (size_t)rowO_rows,    \
                                              (size_t)rowQ_rows, splits, seg, s);                                           \
    }
        if (nt <= 2) ELG_GBF(2)
        if (nt <= 4) ELG_GBF(4)
        if (nt <= 7) ELG_GBF(7)
        ELG_GBF(8)
#undef ELG_GBF
    }
#define ELG_GB(NT)                                                                                                       \
    {                                                                                                                    \
        if (mk) return launch_glimpse_bwd_mfma<NT, true>(rowA, mk, dO, rowO, rowQ, Kmat, Vmat, dQ, dK_part, dV_part, B,  \
                                                         R, N1, (size_t)rowA_rows, (size_t)rowO_rows, (size_t)rowQ_rows, \
                                                         splits, seg, s);                                                \
        return launch_glimpse_bwd_mfma<NT, false>(rowA, mk, dO, rowO, rowQ, Kmat, Vmat, dQ, dK_part, dV_part, B, R, N1,  \
                                                  (size_t)rowA_rows, (size_t)rowO_rows, (size_t)rowQ_rows, splits, seg, s); \
    }
    if (nt <= 2) ELG_GB(2)
    if (nt <= 4) ELG_GB(4)
    if (nt <= 7) ELG_GB(7)
    ELG_GB(8)
#undef ELG_GB
}
}  // namespace elg

extern "C" int elg_glimpse_bwd_fused(const float* rowA, const uint64_t* rowMask, const float* dO, const float* rowO,
                                     const float* rowQ, const float* Kmat, const float* Vmat, float* dQ, float* dK_part,
                                     float* dV_part, int B, int R, int N1, int64_t rowA_rows, int64_t rowO_rows,
                                     int64_t rowQ_rows, int splits, void* stream) {
    GlimpseSeg seg{};
    return glimpse_bwd_launch(rowA, reinterpret_cast<const unsigned long long*>(rowMask), dO, rowO, rowQ, Kmat, Vmat, dQ,
                              dK_part, dV_part, B, R, N1, rowA_rows, rowO_rows, rowQ_rows, splits, seg, (hipStream_t)stream);
}

extern "C" int elg_glimpse_rows_bwd(const float* rowA, const float* dO, const float* rowO, const float* Kmat,
                                    const float* Vmat, float* dS, float* dQ, int B, int R, int N1,
                                    int64_t rowA_rows, int64_t rowO_rows, void* stream) {
    if (rowA_rows < R || rowO_rows < R) return fail(ELG_EINVAL, "glimpse_rows_bwd: row strides smaller than R");
    if (B <= 0 || R <= 0 || N1 <= 1) return fail(ELG_EINVAL, "glimpse_rows_bwd: bad sizes");
    const int nch = (N1 + 63) / 64;
    const int rpb = 256;
    dim3 grid((R + rpb - 1) / rpb, B * 8);
    const size_t lds = 0;
    (void)hipGetLastError();
    if (nch == 1) hipLaunchKernelGGL(glimpse_rows_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, rowA, dO, rowO, Kmat, Vmat, dS, dQ, R, N1, rpb, (size_t)rowA_rows, (size_t)rowO_rows);
    else if (nch == 2) hipLaunchKernelGGL(glimpse_rows_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, rowA, dO, rowO, Kmat, Vmat, dS, dQ, R, N1, rpb, (size_t)rowA_rows, (size_t)rowO_rows);
    else if (nch <= 4) hipLaunchKernelGGL(glimpse_rows_kernel<4>, grid, dim3(256), lds, (hipStream_t)stream, rowA, dO, rowO, Kmat, Vmat, dS, dQ, R, N1, rpb, (size_t)rowA_rows, (size_t)rowO_rows);
    else return fail(ELG_ENOTIMPL, "glimpse_rows_bwd: N1 > 256 (the K_h / V_h rows of a lane no longer fit its registers): use the "
                                   "batched products of elg_gemm_f32_batched");
    return launch_status("glimpse_rows_bwd");
}

extern "C" int elg_rollout_bwd(const elg_bwd_args* a, void* stream) {
    if (!a) return fail(ELG_EINVAL, "null args");
    const elg_bwd_args& BA = *a;
    const elg_rollout_args& A = BA.fwd;
    if (A.B <= 0 || A.M <= 0 || A.N1 <= 1 || A.tiles <= 0 || BA.T <= 0) return fail(ELG_EINVAL, "rollout_bwd: bad sizes");
    if (!A.forced || A.Tforced < BA.T) return fail(ELG_EINVAL, "rollout_bwd: recorded actions missing");
    if (!BA.local_only && (!BA.gprob || !BA.rowA || !BA.rowDL || !BA.rowQ || !BA.rowO))
        return fail(ELG_EINVAL, "rollout_bwd: missing row buffers");
    if (A.has_local && (!A.loc || !BA.rowDU || !BA.gloc)) return fail(ELG_EINVAL, "rollout_bwd: local buffers missing");
    if (A.K + 1 > ELG_SLOT_MAX) return fail(ELG_EINVAL, "rollout_bwd: local_size must be <= 63");
    if (A.ens > 1) {
        if (A.ens > ELG_MAX_ENS || A.problem != ELG_PROBLEM_CVRP || A.Kens[0] != A.K) return fail(ELG_EINVAL, "rollout_bwd: bad ensemble");
        for (int i = 0; i < A.ens; ++i)
            if (A.Kens[i] < 0 || A.Kens[i] + 1 > ELG_SLOT_MAX) return fail(ELG_EINVAL, "rollout_bwd: local_size must be <= 63");
    }
    if (A.problem == ELG_PROBLEM_CVRP) return dispatch_bwd<false>(BA, (hipStream_t)stream);
    if (A.problem == ELG_PROBLEM_TSP) return dispatch_bwd<true>(BA, (hipStream_t)stream);
    return fail(ELG_EINVAL, "rollout_bwd: unknown problem");
}
