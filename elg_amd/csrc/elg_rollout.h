// Device-side pieces of one ELG construction step, shared by the forward (elg_fwd.hip) and the
// backward replay (elg_bwd.hip).  One wavefront owns one trajectory; all per-trajectory state is
// wave-uniform (cur, load, visited words) and lives in SGPRs.
//
// Lane layouts used inside a step:
//   node layout   : lane l owns nodes n = l + 64*ch, ch < NCH           (masks, pointer scores, softmax)
//   quad layout   : lane l = (half = l>>5, hq = l&31): channels cb = 4*hq .. +3 (head hq>>2),
//                   rows 2c+half of read c; K/V rows are consumed 1 KiB per wave-instruction,
//                   i.e. two whole 128-float rows, perfectly linear in LDS (glimpse attention)
//   slot layout   : lane j owns k-NN slot j (CVRP: slot 0 = depot)        (local policy)
//   sorted layout : lane l owns position l + 64*ch of the current node's neighbour list
#pragma once
#include "elg_common.h"
#include "../../include/elg_hip.h"

#define ELG_SLOT_STRIDE 48      // slot stride of the saved training rows, of the cooperative / streaming kernels' slot blocks and of the
                                // row backward (K + 1 <= 48: the reference's defaults are 40 / 30)
#define ELG_SLOT_MAX 64         // one slot per lane: what the one-wavefront kernels (forward, replay backward) take (K + 1 <= 64)
#define ELG_SB_MIN 144          // per-wave LDS scratch floats (>= 3*ELG_SLOT_STRIDE, >= 128)

namespace elg {

// per-wave LDS scratch: slot compaction (3*48), o broadcast (128), node-indexed terms (64*NCH)
template <int NCH>
struct SbSize { static constexpr int value = (64 * NCH > ELG_SB_MIN) ? 64 * NCH : ELG_SB_MIN; };
// One-wavefront kernels: slot stride of the per-wave scratch and of the replay backward's rowDU rows (48 up to local_size 47,
// so that nothing changes there; 64 above), and the scratch floats per wave (d | theta | node id of the slots; >= 128, >= 64 NCH)
__host__ __device__ inline int slot_stride_of(int K) { return (K + 1 > ELG_SLOT_STRIDE) ? ELG_SLOT_MAX : ELG_SLOT_STRIDE; }
__host__ __device__ inline int kmax_of(const elg_rollout_args& A) {      // the widest neighbourhood of the launch (ensembles: max_i Kens[i])
    int k = A.K;
    if (A.ens > 1)
        for (int i = 0; i < ELG_MAX_ENS; ++i) if (i < A.ens && A.Kens[i] > k) k = A.Kens[i];
    return k;
}
__host__ __device__ inline int sb_floats_of(int nch, int K) {
    const int a = 64 * nch, b = 3 * slot_stride_of(K);
    return a > b ? a : b;
}

struct Inst {                 // per-instance table pointers (global or LDS)
    const float* K;           // [N1][128]
    const float* V;           // [N1][128]
    const float* PK;          // [N1][128]  (LDS copy is XOR-swizzled per 16-B chunk: chunk ^ (n & 31))
    const float* pb;          // [N1] global
    const float* Q1;          // [N1][128] global
    const float* Q2;          // [N1][128] global (TSP) or null
    const float* wl;          // [128] global (CVRP)
    const float* xy;          // [N1][2]   LDS when K/V/PK are staged, else global (one uniform read per step)
    const float* dem;         // [N1]      LDS (CVRP)
    const int* nidx;          // [N1][N1] global
    const float* ndist;
    const float* ntheta;
    const float* loc;         // folded local tables (global)
};

template <int NCH>
struct Traj {                 // wave-uniform trajectory state
    int cur;
    int first;                // TSP: first node
    int cnt;                  // steps taken so far
    int fin;
    float load;
    float len;                // running tour length (closed at the end for TSP)
    float cx, cy;             // coordinates of `cur` (valid once cnt > 0)
    unsigned long long vis[NCH];
};

template <int NCH>
__device__ __forceinline__ bool test_bit(const unsigned long long (&w)[NCH], int n) {
    // n is per-lane; NCH is small in the LDS-staged configuration, so a select chain is fine
    unsigned long long x = w[0];
#pragma unroll
    for (int c = 1; c < NCH; ++c) x = ((n >> 6) == c) ? w[c] : x;
    return (x >> (n & 63)) & 1ull;
}

// ---------------------------------------------------------------------------------------------
// Feasibility mask of the current state, as ballot words in node layout.
// CVRPEnv.py:214-232: visited | (load + 1e-6 < demand); depot re-opened once finished.
// TSPEnv.py:120: visited only.
// ---------------------------------------------------------------------------------------------
template <int NCH, bool TSP>
__device__ __forceinline__ void build_mask(const Traj<NCH>& st, const Inst& I, int N1, int lane,
                                           unsigned long long (&mk)[NCH]) {
    const float lim = __fadd_rn(st.load, 1e-6f);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int n = lane + 64 * ch;
        bool m = true;
        if (n < N1) {
            m = (st.vis[ch] >> lane) & 1ull;
            if (!TSP) {
                m = m || (lim < I.dem[n]);
                if (n == 0 && st.fin) m = false;
            }
        }
        mk[ch] = __ballot(m);
    }
}

// ---------------------------------------------------------------------------------------------
// Environment transition for the chosen node (CVRPEnv.py:195-232, TSPEnv.py:108-124).
// load is updated with exactly one rounded fp32 subtraction, as the reference does.
// ---------------------------------------------------------------------------------------------
template <int NCH, bool TSP>
__device__ __forceinline__ void env_update(Traj<NCH>& st, const Inst& I, int N1, int sel) {
    const float2 sxy = *reinterpret_cast<const float2*>(I.xy + 2 * sel);
    const float sx = i2f(__builtin_amdgcn_readfirstlane(f2i(sxy.x))), sy = i2f(__builtin_amdgcn_readfirstlane(f2i(sxy.y)));
    if (st.cnt > 0) st.len += dist2d(st.cx, st.cy, sx, sy);
    st.cx = sx; st.cy = sy;
    if (TSP) {
        if (st.cnt == 0) st.first = sel;
    } else {
        st.load = (sel == 0) ? 1.0f : __fsub_rn(st.load, I.dem[sel]);
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
        if ((sel >> 6) == ch) st.vis[ch] |= 1ull << (sel & 63);
    if (!TSP) {
        // depot counts as visited exactly while the trajectory stands on it (CVRPEnv.py:214-216)
        if (sel == 0) st.vis[0] |= 1ull; else st.vis[0] &= ~1ull;
    }
    st.cur = sel;
    st.cnt += 1;
    bool all = true;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int rem = N1 - 64 * ch;
        if (rem > 0) {
            const unsigned long long full = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
            all = all && ((st.vis[ch] & full) == full);
        }
    }
    if (TSP) {
        if (st.cnt == N1) {                    // close the tour (TSPEnv.py:166 roll(-1))
            const float fx = I.xy[2 * st.first], fy = I.xy[2 * st.first + 1];
            st.len += dist2d(sx, sy, fx, fy);
            st.fin = 1;
        }
    } else {
        if (all) st.fin = 1;                   // CVRPEnv.py:226-228
    }
}

// ---------------------------------------------------------------------------------------------
// k-NN slots of the current node: walk its sorted neighbour list, keep the first K unmasked
// customers (total order (dist, index) is baked into the list).  Selected entries are compacted
// to slot order through the per-wave LDS scratch `sb` (3*ELG_SLOT_STRIDE floats): d | theta | node id.
// Returns k (number of valid slots, without the depot slot).        models.py:55-90,355-391
// ---------------------------------------------------------------------------------------------
template <int NCH, bool TSP>
__device__ __forceinline__ int knn_slots(const Inst& I, int N1, int K, int cur, int lane,
                                         const unsigned long long (&mk)[NCH], float* sb,
                                         const unsigned long long* lds_mk = nullptr, int ss = ELG_SLOT_STRIDE) {
    constexpr int S0 = TSP ? 0 : 1;
    int found = 0;
    const size_t row = (size_t)cur * N1;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (64 * ch < N1 && found < K) {                 // wave-uniform early exit
            const int i = lane + 64 * ch;
            const bool valid = i < N1;
            const int nid = valid ? I.nidx[row + i] : 0;
            const float nd = valid ? I.ndist[row + i] : 0.f;        // issued together with the index load
            const float nth = valid ? I.ntheta[row + i] : 0.f;
            // (many mask words: a per-lane select chain would put mk[] in scratch memory; the caller keeps a copy in LDS)
            bool cand = valid && !(lds_mk ? (bool)((lds_mk[nid >> 6] >> (nid & 63)) & 1ull) : test_bit<NCH>(mk, nid));
            if (!TSP) cand = cand && (nid != 0);
            const unsigned long long bal = __ballot(cand);
            const int rank = found + lanes_below(bal);
            if (cand && rank < K) {
                sb[S0 + rank] = nd;
                sb[ss + S0 + rank] = nth;
                sb[2 * ss + S0 + rank] = i2f(nid);
            }
            found += __popcll(bal);
        }
    }
    return found < K ? found : K;
}

// Everything the slot lanes need for one step: k-NN slot contents, penalty, local features, mask.
struct Slots {
    int k;            // valid neighbour slots of the walk (uniform)
    float dmax;       // distance of the k-th neighbour of the penalty / first local policy (uniform value)
    bool cust;        // this lane holds a real neighbour slot of the walk
    bool pcust;       // ... and the slot belongs to the first min(k, K) of them (always == cust without an ensemble)
    int snid;         // node of this lane's slot (-1: none; CVRP lane 0: depot)
    float pen;        // distance penalty of the slot (0 if none)
    float f0, f1, f2; // local-policy features (first member)
    bool smask;       // slot masked / absent for the local attention (first member)
    // ensemble_size > 1 (models.py:296-298): what the other members' features are made of
    float sd, sth, rx, ry, f2raw;
    int km[ELG_MAX_ENS];      // min(k, local_size[i])   (uniform)
    float dm[ELG_MAX_ENS];    // distance of member i's km[i]-th neighbour (uniform value)
};

// `K` = local_size[0]: the distance penalty's and the first local policy's neighbourhood.  With ens > 1 the walk keeps the first
// max_i Kens[i] open customers; member i's set is the first min(k, Kens[i]) of them (a prefix: the list is sorted).
template <int NCH, bool TSP, class ST>     // ST: anything with .cur and .load
__device__ __forceinline__ Slots slot_setup(const Inst& I, int N1, int K, bool has_penalty, const ST& st,
                                            int lane, const unsigned long long (&mk)[NCH], float* sb,
                                            const unsigned long long* lds_mk = nullptr, bool euclid = false,
                                            int ens = 1, const int32_t* Kens = nullptr) {
    constexpr int S0 = TSP ? 0 : 1;
    Slots S;
    int Kw = K;
    if (ens > 1) {
#pragma unroll
        for (int i = 0; i < ELG_MAX_ENS; ++i) if (i < ens) Kw = max(Kw, Kens[i]);
    }
    const int ss = slot_stride_of(Kw);                                // (the scratch is sized by the same rule: sb_floats_of)
    S.k = knn_slots<NCH, TSP>(I, N1, Kw, st.cur, lane, mk, sb, lds_mk, ss);
    wave_lds_fence();
    const int j = lane;
    const int kp = min(S.k, K);
    S.cust = (j >= S0) && (j < S0 + S.k);
    S.pcust = (j >= S0) && (j < S0 + kp);
    float sd = 0.f, sth = 0.f;
    S.snid = -1;
    if (S.cust) {
        sd = sb[j];
        sth = sb[ss + j];
        S.snid = f2i(sb[2 * ss + j]);
    }
    S.dmax = (kp > 0) ? sb[S0 + kp - 1] : 0.f;
#pragma unroll
    for (int i = 0; i < ELG_MAX_ENS; ++i) {
        S.km[i] = 0; S.dm[i] = 0.f;
        if (ens > 1 && i < ens) {
            S.km[i] = min(S.k, Kens[i]);
            S.dm[i] = (S.km[i] > 0) ? sb[S0 + S.km[i] - 1] : 0.f;
        }
    }
    wave_lds_fence();
    if (!TSP && j == 0) S.snid = 0;                                   // depot slot
    S.pen = 0.f;
    if (has_penalty && S.pcust) {
        if (TSP) S.pen = -(sd / (S.dmax + 1e-6f));                    // TSP/models.py:290
        else S.pen = (S.dmax != 0.f) ? -(sd / S.dmax) : -sd;          // models.py:379-405 (no epsilon)
    }
    const float nf = S.dmax + 1e-6f;                                  // models.py:79 / TSP :72
    S.f0 = S.f1 = S.f2 = 0.f;
    S.sd = sd; S.sth = sth; S.rx = S.ry = S.f2raw = 0.f;
    if (S.cust) {
        if (euclid) {                                                 // models.py:95-125: relative (x, y) / norm (CVRPEnv.py:303)
            const float cx = I.xy[2 * st.cur], cy = I.xy[2 * st.cur + 1];
            S.rx = __fsub_rn(I.xy[2 * S.snid], cx);
            S.ry = __fsub_rn(I.xy[2 * S.snid + 1], cy);
        }
        if (!TSP) S.f2raw = I.dem[S.snid] / st.load;                  // CVRPEnv.py:315-316
    }
    if (S.pcust) {
        S.f0 = euclid ? S.rx / nf : sd / nf;
        S.f1 = euclid ? S.ry / nf : sth;
        S.f2 = S.f2raw;
    }
    S.smask = !S.pcust;
    if (!TSP && j == 0) S.smask = mk[0] & 1ull;                       // depot slot carries the depot's mask
    return S;
}

// Features / mask of ensemble member i on this lane's slot, and whether the member's output lands on a node at all
// (its depot slot and its own min(k, Kens[i]) neighbours; models.py:64-76,168-172).
template <bool TSP>
__device__ __forceinline__ bool member_slot(const Slots& S, int i, int lane, bool euclid, float& f0, float& f1, float& f2,
                                            bool& smask) {
    constexpr int S0 = TSP ? 0 : 1;
    int km = S.km[0];
    float dm = S.dm[0];
#pragma unroll
    for (int q = 1; q < ELG_MAX_ENS; ++q) if (i == q) { km = S.km[q]; dm = S.dm[q]; }
    const bool c = S.cust && (lane < S0 + km);
    const float nf = dm + 1e-6f;
    f0 = c ? (euclid ? S.rx / nf : S.sd / nf) : 0.f;
    f1 = c ? (euclid ? S.ry / nf : S.sth) : 0.f;
    f2 = c ? S.f2raw : 0.f;
    const bool depot = !TSP && lane == 0;
    smask = depot ? S.smask : !c;
    return c || depot;
}

// 32 per-lane values -> lane l ends with the wave-wide sum of element (l & 31)
__device__ __forceinline__ float reduce_scatter32(float (&c)[32], int lane) {
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float keep = b4 ? c[i + 16] : c[i], send = b4 ? c[i] : c[i + 16];
        c[i] = keep + shfl_xor(send, 16);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b3 ? c[i + 8] : c[i], send = b3 ? c[i] : c[i + 8];
        c[i] = keep + shfl_xor(send, 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b2 ? c[i + 4] : c[i], send = b2 ? c[i] : c[i + 4];
        c[i] = keep + shfl_xor(send, 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b1 ? c[i + 2] : c[i], send = b1 ? c[i] : c[i + 2];
        c[i] = keep + quad_xor2(send);
    }
    {
        const float keep = b0 ? c[1] : c[0], send = b0 ? c[0] : c[1];
        c[0] = keep + quad_xor1(send);
    }
    return x32_sum(c[0]);
}

// 16 per-lane values -> lane l ends with the wave-wide sum of element (l & 15).
// The three in-row butterfly stages are DPP (row_ror / quad_perm); only the two cross-row adds use the
// LDS crossbar.
__device__ __forceinline__ float reduce_scatter16(float (&c)[16], int lane) {
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b3 ? c[i + 8] : c[i], send = b3 ? c[i] : c[i + 8];
        c[i] = keep + row_xor8(send);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b2 ? c[i + 4] : c[i], send = b2 ? c[i] : c[i + 4];
        c[i] = keep + row_xor4(send, b2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b1 ? c[i + 2] : c[i], send = b1 ? c[i] : c[i + 2];
        c[i] = keep + quad_xor2(send);
    }
    {
        const float keep = b0 ? c[1] : c[0], send = b0 ? c[0] : c[1];
        c[0] = keep + quad_xor1(send);
    }
    return quarters_sum(c[0]);
}

// Saved intermediates of the local policy for the backward pass.
struct LocalSave {
    float al[ELG_LH];        // attention weights alpha_h of this slot
    float op;                // o'[lane & 31]
    float g;                 // g'[lane & 31]
    float Ftot;              // element (lane & 15) of F_h[f] = sum_j alpha_hj f_j[f]   (index 3h+f)
};

// ---------------------------------------------------------------------------------------------
// Local policy on the k-NN slots (slot layout: lane j = slot j).  Returns u_j of this lane's slot.
// models.py:133-166 with every projection that does not depend on the features folded on the host:
//   sc_h   = la[h].f_j + lt[j][h]                        (q is one learned vector for all trajectories)
//   alpha  = softmax_j(sc_h + mask)
//   o'[d]  = sum_j alpha_{h(d),j} lcv[j][d]  +  lAv[d] . F_{h(d)},     F_h = sum_j alpha_{h,j} f_j
//   g'     = lWc o' + lbc
//   u_j    = sum_d g'[d] lpe[j][d]  +  w . f_j,                        w = sum_d g'[d] lWe[d]
// (1/sqrt(8) is folded into la/lt, 1/sqrt(32) into lWe/lpe).  All cross-lane sums are DPP/readlane.
// ---------------------------------------------------------------------------------------------
template <bool TSP>
__device__ __forceinline__ float local_policy(const float* __restrict__ loc, int lane, float f0, float f1,
                                              float f2, bool smask, LocalSave* save) {
    const int j = lane, dd = lane & 31, hh = dd >> 3;
    constexpr int NF = TSP ? 2 : 3;
    const float f[3] = {f0, f1, TSP ? 0.f : f2};
    const float* la = loc + ELG_LOC_LA;
    const float4 lt = *reinterpret_cast<const float4*>(loc + ELG_LOC_LT + 4 * j);
    const float sc0[ELG_LH] = {lt.x, lt.y, lt.z, lt.w};
    float al[ELG_LH];
#pragma unroll
    for (int h = 0; h < ELG_LH; ++h) {
        float s = sc0[h];
#pragma unroll
        for (int k = 0; k < NF; ++k) s = fmaf(la[3 * h + k], f[k], s);
        s = smask ? ELG_NEG_INF : s;
        const float mx = wave_max(s);
        const float e = smask ? 0.f : __expf(s - mx);
        const float den = wave_sum(e);
        al[h] = den > 0.f ? e / den : 0.f;
    }
    // F_h[f] = sum_j alpha_hj f_j[f]  -> lane (l & 15) holds element 3h+f
    float fr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) fr[i] = 0.f;
#pragma unroll
    for (int h = 0; h < ELG_LH; ++h)
#pragma unroll
        for (int k = 0; k < NF; ++k) fr[3 * h + k] = al[h] * f[k];
    const float Ftot = reduce_scatter16(fr, lane);
    // P[d] = sum_j alpha_{h(d),j} lcv[j][d]  -> lane (l & 31) holds element d, two halves of 16
    const float* lcv = loc + ELG_LOC_LCV + 32 * j;
    float P;
    {
        float c[16];
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            const float4 cv = *reinterpret_cast<const float4*>(lcv + 4 * d4);
            c[4 * d4 + 0] = al[(4 * d4) >> 3] * cv.x; c[4 * d4 + 1] = al[(4 * d4) >> 3] * cv.y;
            c[4 * d4 + 2] = al[(4 * d4) >> 3] * cv.z; c[4 * d4 + 3] = al[(4 * d4) >> 3] * cv.w;
        }
        const float plo = reduce_scatter16(c, lane);               // element (l & 15) of d = 0..15
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            const float4 cv = *reinterpret_cast<const float4*>(lcv + 16 + 4 * d4);
            c[4 * d4 + 0] = al[2 + ((4 * d4) >> 3)] * cv.x; c[4 * d4 + 1] = al[2 + ((4 * d4) >> 3)] * cv.y;
            c[4 * d4 + 2] = al[2 + ((4 * d4) >> 3)] * cv.z; c[4 * d4 + 3] = al[2 + ((4 * d4) >> 3)] * cv.w;
        }
        const float phi = reduce_scatter16(c, lane);               // element (l & 15) of d = 16..31
        P = (lane & 16) ? phi : plo;
    }
    float op = P;                                                   // o'[dd]
    {
        const float* lAv = loc + ELG_LOC_LAV + 3 * dd;
#pragma unroll
        for (int k = 0; k < NF; ++k) op = fmaf(lAv[k], __shfl(Ftot, 3 * hh + k, ELG_WAVE), op);
    }
    // g' = lWc o' + lbc  (lane d' computes g'[d'])
    const float* wrow = loc + ELG_LOC_LWC + 32 * dd;
    float g = loc[ELG_LOC_LBC + dd];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
        const float4 w = *reinterpret_cast<const float4*>(wrow + 4 * d4);
        g = fmaf(w.x, readlane(op, 4 * d4 + 0), g);
        g = fmaf(w.y, readlane(op, 4 * d4 + 1), g);
        g = fmaf(w.z, readlane(op, 4 * d4 + 2), g);
        g = fmaf(w.w, readlane(op, 4 * d4 + 3), g);
    }
    // u_j = sum_d g'[d] lpe[j][d] + w . f_j
    const float* lpe = loc + ELG_LOC_LPE + 32 * j;
    float u = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
        const float4 pe = *reinterpret_cast<const float4*>(lpe + 4 * d4);
        u = fmaf(readlane(g, 4 * d4 + 0), pe.x, u);
        u = fmaf(readlane(g, 4 * d4 + 1), pe.y, u);
        u = fmaf(readlane(g, 4 * d4 + 2), pe.z, u);
        u = fmaf(readlane(g, 4 * d4 + 3), pe.w, u);
    }
    {
        const float* lWe = loc + ELG_LOC_LWE + 3 * dd;
        const float glo = (lane < 32) ? g : 0.f;
#pragma unroll
        for (int k = 0; k < NF; ++k) u = fmaf(half_sum_lo(glo * lWe[k]), f[k], u);
    }
    if (save) {
#pragma unroll
        for (int h = 0; h < ELG_LH; ++h) save->al[h] = al[h];
        save->op = op; save->g = g; save->Ftot = Ftot;
    }
    return u;
}

// Local-policy term of this lane's slot: one policy, or the sum over the ensemble's members (models.py:409-413; the caller
// scales by 1 / ensemble_size).  A member contributes only to its own slots.
template <bool TSP>
__device__ __forceinline__ float local_ensemble(const elg_rollout_args& A, const float* __restrict__ loc, int lane,
                                                const Slots& S) {
    if (A.ens <= 1) return local_policy<TSP>(loc, lane, S.f0, S.f1, S.f2, S.smask, nullptr);
    float u = 0.f;
#pragma unroll 1
    for (int i = 0; i < A.ens; ++i) {
        float f0, f1, f2;
        bool sm;
        const bool in = member_slot<TSP>(S, i, lane, A.euclidean != 0, f0, f1, f2, sm);
        const float ui = local_policy<TSP>(loc + (size_t)i * ELG_LOC_SIZE, lane, f0, f1, f2, sm, nullptr);
        u += in ? ui : 0.f;
    }
    return u;
}

// Additive slot term (distance penalty + local score).  With an ensemble whose members look further than the penalty's
// local_size[0], the slots past the penalty's set keep the default xi (models.py:405-407).
__device__ __forceinline__ float slot_penalty(const elg_rollout_args& A, const Slots& S) {
    return (A.ens > 1 && A.has_penalty && S.cust && !S.pcust) ? A.xi : S.pen;
}

// Saved intermediates of the glimpse for the backward pass (quad layout).
template <int NG>
struct GlimpseSave {
    float e[NG];             // normalised attention weights a_h[row] of this lane's rows
    float4 q4;               // q[cb..cb+3]
};

// number of 8-row groups the glimpse walks (static: the loops must be branch-free so that the compiler can
// keep many LDS reads in flight): 13 covers N1 <= 104 (CVRP-100), otherwise whole 64-node chunks
template <int NCH, bool SMALL>
struct GlimpseGroups { static constexpr int value = (NCH == 2 && SMALL) ? 13 : 8 * NCH; };

// ---------------------------------------------------------------------------------------------
// Glimpse: 8-head attention of the trajectory's query over all nodes (models.py:330-341,455-503).
// K and V are consumed two whole rows (1 KiB) per wave-instruction; the per-head 16-wide dot
// products are finished with a quad reduce-scatter so that every lane ends up with the scores of
// its own rows (row = 8k + 2*(lane&3) + (lane>>5), k = 0..NG-1) for head (lane&31)>>2.
// Returns o[cb..cb+3] (cb = 4*(lane&31)), identical in both half-waves.
// All loops are straight-line (rows past N1 are clamped / masked, never branched around).
// ---------------------------------------------------------------------------------------------
template <int NCH, bool LDSK, int NG>
__device__ __forceinline__ float4 glimpse(const Inst& I, int N1, int lane, const float4 q4,
                                          const unsigned long long (&mk)[NCH], GlimpseSave<NG>* save) {
    const int half = lane >> 5, hq = lane & 31, ql = lane & 3, cb = hq * 4;
    const bool b0 = ql & 1, b1 = ql & 2;
    const int r = 2 * ql + half;                   // row offset of this lane inside a group
    const float* Kp = I.K + cb;
    const float* Vp = I.V + cb;
    float m_run = ELG_NEG_INF, l_run = 0.f;
    f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    constexpr int GB = NG < 16 ? NG : 16;          // groups per online-softmax block
    static_assert(NG % GB == 0, "group count must be a multiple of the block size");
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += GB) {
        if (NG > GB && 8 * g0 >= N1) {             // block-level skip (only for multi-block, large-N builds)
            if (save) {
#pragma unroll
                for (int k = 0; k < GB; ++k) save->e[g0 + k] = 0.f;
            }
            continue;
        }
        float sc[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            const int g = g0 + k;
            sc[k] = ELG_NEG_INF;
            if (8 * g >= N1) continue;                           // wave-uniform (scalar branch)
            float4 kv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                int row = 8 * g + 2 * jj + half;
                if (!LDSK) row = row < N1 ? row : N1 - 1;      // LDS copy: rows past N1 read the next table (masked)
                kv[jj] = *reinterpret_cast<const float4*>(Kp + (size_t)row * ELG_E);
            }
            const float p0 = dot4p(kv[0], q4), p1 = dot4p(kv[1], q4);
            const float p2 = dot4p(kv[2], q4), p3 = dot4p(kv[3], q4);
            const float s0 = b0 ? p1 : p0, t0 = b0 ? p0 : p1;
            const float s1 = b0 ? p3 : p2, t1 = b0 ? p2 : p3;
            const float a0 = s0 + quad_xor1(t0), a1 = s1 + quad_xor1(t1);
            const float keep = b1 ? a1 : a0, send = b1 ? a0 : a1;
            const float dotv = keep + quad_xor2(send);
            const int myrow = 8 * g + r;
            const unsigned byte = (unsigned)(mk[g >> 3] >> (8 * (g & 7))) & 0xffu;   // uniform
            const bool masked = (myrow >= N1) || ((byte >> r) & 1u);
            sc[k] = masked ? ELG_NEG_INF : dotv * 0.25f;          // / sqrt(qkv_dim)
        }
        float mb = sc[0];
#pragma unroll
        for (int k = 1; k < GB; ++k) mb = fmaxf(mb, sc[k]);
        mb = fmaxf(mb, quad_xor1(mb));
        mb = fmaxf(mb, quad_xor2(mb));
        mb = x32_max(mb);
        const float m_new = fmaxf(m_run, mb);
        // m_new == -inf only if every node seen so far is masked (same for all lanes): nothing to add
        const bool live = m_new > ELG_NEG_INF;
        const float scale = (live && m_run > ELG_NEG_INF) ? __expf(m_run - m_new) : 0.f;
        l_run *= scale;
        acc01 *= scale; acc23 *= scale;
        if (save) {
#pragma unroll
            for (int k = 0; k < NG; ++k) if (k < g0) save->e[k] *= scale;
        }
        m_run = m_new;
        float e[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            e[k] = (live && sc[k] > ELG_NEG_INF) ? __expf(sc[k] - m_new) : 0.f;
            l_run += e[k];
            if (save) save->e[g0 + k] = e[k];
        }
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            const int g = g0 + k;
            if (8 * g >= N1) continue;                           // wave-uniform (scalar branch)
            float4 vv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                int row = 8 * g + 2 * jj + half;
                if (!LDSK) row = row < N1 ? row : N1 - 1;
                vv[jj] = *reinterpret_cast<const float4*>(Vp + (size_t)row * ELG_E);
            }
#define ELG_VACC(JJ)                                                                             \
    {                                                                                            \
        /* rows past N1 read finite bytes of the next table (or a clamped row): weight 0 */      \
        const bool ok = (8 * g + 2 * JJ + half) < N1;                                            \
        float a = quad_bcast<JJ>(e[k]);                                                          \
        a = ok ? a : 0.f;                                                                        \
        const f32x2 a2 = {a, a};                                                                 \
        acc01 = __builtin_elementwise_fma(a2, lo2(vv[JJ]), acc01);                               \
        acc23 = __builtin_elementwise_fma(a2, hi2(vv[JJ]), acc23);                               \
    }
            ELG_VACC(0) ELG_VACC(1) ELG_VACC(2) ELG_VACC(3)
        }
#undef ELG_VACC
    }
    // denominators: lanes of one head = the quad in both half-waves
    float l = l_run;
    l += quad_xor1(l);
    l += quad_xor2(l);
    l = x32_sum(l);
    const float inv = 1.0f / l;
    float4 acc = make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
    acc.x = x32_sum(acc.x); acc.y = x32_sum(acc.y);
    acc.z = x32_sum(acc.z); acc.w = x32_sum(acc.w);
    if (save) {
#pragma unroll
        for (int k = 0; k < NG; ++k) save->e[k] *= inv;
        save->q4 = q4;
    }
    return make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

// ---------------------------------------------------------------------------------------------
// Pointer scores s[n] = o . PK[n] + pb[n]   (models.py:341-352 with combine folded into PK).
// o is broadcast through the wave's LDS scratch; every lane walks its own rows of PK
// (branch-free: rows past N1 are clamped and their result discarded).
// ---------------------------------------------------------------------------------------------
template <int NCH, bool LDSK>
__device__ __forceinline__ void pointer_scores(const Inst& I, int N1, int lane, const float4 o4, float* sb,
                                               float (&s)[NCH]) {
    if (lane < 32) *reinterpret_cast<float4*>(sb + 4 * lane) = o4;
    wave_lds_fence();
    f32x2 acc[NCH];
    const float* rowp[NCH];
    int sw[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        acc[ch] = f32x2{0.f, 0.f};
        const int n = lane + 64 * ch;
        const int nc = n < N1 ? n : N1 - 1;
        rowp[ch] = I.PK + (size_t)nc * ELG_E;
        sw[ch] = LDSK ? (nc & 31) : 0;
    }
#pragma unroll 8
    for (int c4 = 0; c4 < 32; ++c4) {
        const float4 o = *reinterpret_cast<const float4*>(sb + 4 * c4);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const float4 pk = *reinterpret_cast<const float4*>(rowp[ch] + 4 * (c4 ^ sw[ch]));
            acc[ch] = __builtin_elementwise_fma(lo2(o), lo2(pk), acc[ch]);
            acc[ch] = __builtin_elementwise_fma(hi2(o), hi2(pk), acc[ch]);
        }
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int n = lane + 64 * ch;
        s[ch] = (n < N1) ? (acc[ch].x + acc[ch].y) + I.pb[n] : 0.f;
    }
    wave_lds_fence();
}

}  // namespace elg
