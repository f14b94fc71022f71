// Forward POMO construction kernels (persistent: one launch runs every trajectory of the batch to completion)
// + the small per-batch kernels (neighbour tables, distance matrix, aug8, route length).
//   rollout_fwd_coop_kernel   N1 <= 112: lockstep trajectories, glimpse / pointer / local policy on fp32 MFMA,
//                             clip / softmax / choice / transition / k-NN for four trajectories per wave
//   rollout_fwd_mt_kernel     128 < N1 <= 1024: 16 lockstep trajectories per workgroup, glimpse / pointer on fp32 MFMA with
//                             the K / V / PK fragments streamed from L2
//   rollout_fwd_xl_kernel     N1 > 1024 (up to 8192): node-indexed state in LDS words + global scratch rows
//   rollout_fwd_kernel        one wavefront per trajectory: the step-wise protocol (CVRPEnv.step, one_step_rollout),
//                             explicit geometries, A/B reference
//
// Replaces, for gaocrr/ELG: CVRP/utils.py:7-29 (rollout loop), CVRPEnv.py:152-318 (reset / step /
// get_cur_feature / _get_reward), CVRPModel.py:36-75 (one_step_rollout), models.py:51-175,322-423
// (local policy + decoder); and the TSP counterparts (TSP/utils.py:7-26, TSPEnv.py, TSPModel.py,
// TSP/models.py:48-110,244-303).
#include <string>
#include "elg_coop.h"
// diagnostic builds only (tools/exp_rollout_stores.sh): -DELG_EXP_SKIP=<bits> compiles saved-row stores of the cooperative kernel out
// (1 trQ, 2 trSlot / trF, 4 trO, 8 trPC / trCsel) to TIME the launch without them; the shipped library is built with 0.
#ifndef ELG_EXP_SKIP
#define ELG_EXP_SKIP 0
#endif
#include <string>

namespace elg {

static thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
const char* last_error() { return g_err.c_str(); }
static thread_local int g_kernel = 0;
void note_kernel(int id) { g_kernel = id; }
int launch_status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return ELG_OK;
    return fail(ELG_ELAUNCH, std::string(what) + " launch failed: " + hipGetErrorString(e));
}

// =============================================================================================
// small kernels
// =============================================================================================
__global__ void aug8_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over B*N points
    if (i >= B * N) return;
    const float x = in[2 * i], y = in[2 * i + 1];
    const float mx = __fsub_rn(1.f, x), my = __fsub_rn(1.f, y);
    const size_t s = (size_t)B * N * 2;
    float* o = out + 2 * (size_t)i;
    o[0] = x;          o[1] = y;
    o[s] = mx;         o[s + 1] = y;
    o[2 * s] = x;      o[2 * s + 1] = my;
    o[3 * s] = mx;     o[3 * s + 1] = my;
    o[4 * s] = y;      o[4 * s + 1] = x;
    o[5 * s] = my;     o[5 * s + 1] = x;
    o[6 * s] = y;      o[6 * s + 1] = mx;
    o[7 * s] = my;     o[7 * s + 1] = mx;
}

__global__ void dist_matrix_kernel(const float* __restrict__ xy, float* __restrict__ dist, int N) {
    const int b = blockIdx.y, i = blockIdx.x;
    const float* p = xy + (size_t)b * N * 2;
    const float ax = p[2 * i], ay = p[2 * i + 1];
    float* row = dist + ((size_t)b * N + i) * N;
    for (int j = threadIdx.x; j < N; j += blockDim.x) row[j] = dist2d(ax, ay, p[2 * j], p[2 * j + 1]);
}

// one workgroup per (instance, centre node): bitonic sort of (dist bits << 32 | index) in LDS
__global__ void nbr_tables_kernel(const float* __restrict__ xy, int* __restrict__ idx, float* __restrict__ dist,
                                  float* __restrict__ theta, int N, int NP) {
    extern __shared__ unsigned long long keys[];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const float* p = xy + (size_t)b * N * 2;
    const float cx = p[2 * c], cy = p[2 * c + 1];
    for (int i = tid; i < NP; i += nt) {
        unsigned long long k = ~0ull;
        if (i < N) {
            const float d = dist2d(p[2 * i], p[2 * i + 1], cx, cy);
            k = ((unsigned long long)(unsigned)f2i(d) << 32) | (unsigned)i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NP; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], bb = keys[ixj];
                    const bool asc = (i & k) == 0;
                    if ((a > bb) == asc) { keys[i] = bb; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    const size_t row = ((size_t)b * N + c) * N;
    for (int i = tid; i < N; i += nt) {
        const unsigned long long k = keys[i];
        const int n = (int)(k & 0xffffffffull);
        idx[row + i] = n;
        dist[row + i] = i2f((int)(k >> 32));
        theta[row + i] = atan2f(__fsub_rn(p[2 * n + 1], cy), __fsub_rn(p[2 * n], cx));   // CVRPEnv.py:302-311
    }
}

__global__ void route_length_kernel(const float* __restrict__ xy, const long long* __restrict__ tour,
                                    float* __restrict__ out, int B, int M, int T, int N, int rounding) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over B*M
    if (i >= B * M) return;
    const int b = i / M;
    const float* p = xy + (size_t)b * N * 2;
    const long long* t = tour + (size_t)i * T;
    float len = 0.f;
    int prev = (int)t[0];
    const int first = prev;
    for (int s = 1; s <= T; ++s) {
        const int nx = (s < T) ? (int)t[s] : first;                 // roll(-1): last -> first
        float d = dist2d(p[2 * prev], p[2 * prev + 1], p[2 * nx], p[2 * nx + 1]);
        if (rounding) d = rintf(d);                                 // torch.round = half to even
        len += d;
        prev = nx;
    }
    out[i] = len;
}

// =============================================================================================
// the rollout kernel
// =============================================================================================
struct FwdOut {
    int sel;
    float p;
};

// Tail of a decode step, shared by the resident-table and the node-tiled kernels: scatter the slot terms,
// clip / mask / softmax in node layout, choose, (training) save the softmax x clip Jacobian.
template <int NCH, bool TSP, bool TRAIN>
__device__ __forceinline__ FwdOut finish_step(const elg_rollout_args& A, int N1, int lane, float* sb,
                                              const unsigned long long (&mk)[NCH], const float (&s)[NCH], int snid,
                                              float addval, int forced_sel, float uni, float* full_row, size_t b,
                                              size_t r, size_t Rcap, float* pcj_out = nullptr) {
    // ---- scatter the slot terms to node order (xi everywhere else)   models.py:405-413
    const float dflt = A.has_penalty ? A.xi : 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int n = lane + 64 * ch;
        if (n < N1) sb[n] = dflt;
    }
    wave_lds_fence();
    if (snid >= 0) sb[snid] = addval;
    wave_lds_fence();

    // ---- clip, mask, softmax (node layout)   models.py:416-420
    float lg[NCH], th[NCH];
    float mx = ELG_NEG_INF;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int n = lane + 64 * ch;
        const bool masked = (mk[ch] >> lane) & 1ull;
        float x = ELG_NEG_INF;
        th[ch] = 0.f;
        if (n < N1 && !masked) { th[ch] = fast_tanh(s[ch] + sb[n]); x = A.clip * th[ch]; }
        lg[ch] = x;
        mx = fmaxf(mx, x);
    }
    wave_lds_fence();
    mx = wave_max(mx);
    float e[NCH];
    float part = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        e[ch] = (lg[ch] > ELG_NEG_INF) ? __expf(lg[ch] - mx) : 0.f;
        part += e[ch];
    }
    const float tot = wave_sum(part);
    const float inv = 1.0f / tot;
    if (full_row) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int n = lane + 64 * ch;
            if (n < N1) {
                float v = e[ch] * inv;                                   // probabilities
                if (A.dump_logits) {                                     // 1: clipped logits, 2: the score before the clip
                    const bool open = lg[ch] > ELG_NEG_INF;
                    v = !open ? ELG_NEG_INF : (A.dump_logits == 2 ? s[ch] + sb[n] : lg[ch]);
                }
                full_row[n] = v;
            }
        }
    }

    // ---- choose
    int sel = 0;
    if (A.mode == ELG_MODE_FORCED) {
        sel = forced_sel;
    } else if (A.mode == ELG_MODE_GREEDY) {
        // argmax over probabilities, ties -> lowest node index (torch.argmax): the maximum by a DPP all-reduce, then the first
        // node that attains it by one ballot per 64-node chunk (the butterfly on (value, index) pairs this replaces went
        // through ds_bpermute: 12 LDS round trips in a row)
        float pv[NCH];
        float bv = -1.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int n = lane + 64 * ch;
            pv[ch] = n < N1 ? e[ch] * inv : -1.f;
            bv = fmaxf(bv, pv[ch]);
        }
        bv = wave_max(bv);
        int found = -1;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const unsigned long long hit = __ballot(pv[ch] == bv);
            if (found < 0 && hit) found = 64 * ch + (int)__builtin_ctzll(hit);
        }
        sel = found < 0 ? 0 : found;
    } else {
        // inverse-CDF sample in node order
        float cs[NCH];
        float run = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            cs[ch] = wave_scan_incl(e[ch], lane) + run;
            run = readlane(cs[ch], 63);
        }
        const float target = uni * run;
        int found = -1, lastpos = 0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const unsigned long long pos = __ballot(e[ch] > 0.f);
            const unsigned long long hit = __ballot(e[ch] > 0.f && cs[ch] > target);
            if (found < 0 && hit) found = 64 * ch + (int)__builtin_ctzll(hit);
            if (pos) lastpos = 64 * ch + 63 - (int)__builtin_clzll(pos);
        }
        sel = found >= 0 ? found : lastpos;
    }
    // probability of the chosen node
    sel = __builtin_amdgcn_readfirstlane(sel);
    float ps = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
        if ((sel >> 6) == ch) ps = readlane(e[ch], sel & 63) * inv;
    if (TRAIN) {
        // p[n] clip (1 - tanh^2): everything the backward needs of the clip / softmax Jacobian
        float* rPC = A.trPC + (b * Rcap + r) * N1;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int n = lane + 64 * ch;
            const float cj = A.clip * (1.f - th[ch] * th[ch]);
            if (n < N1) rPC[n] = e[ch] * inv * cj;
            if (n == sel) A.trCsel[b * Rcap + r] = cj;
        }
    }
    if (pcj_out) {       // the caller stores the Jacobian row itself: pcj_out[0 .. NCH) = p clip (1 - tanh^2) of this lane's nodes,
                         // pcj_out[NCH] = clip (1 - tanh^2) at the chosen node (uniform)
        float cs_ = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const float cj = A.clip * (1.f - th[ch] * th[ch]);
            pcj_out[ch] = e[ch] * inv * cj;
            if ((sel >> 6) == ch) cs_ = readlane(cj, sel & 63);
        }
        pcj_out[NCH] = cs_;
    }
    FwdOut o;
    o.sel = sel;
    o.p = ps;
    return o;
}

// One decode step for the trajectory held by this wave.  Returns the chosen node and its probability.
template <int NCH, bool TSP, bool LDSK, bool TRAIN, bool SMALL>
__device__ __forceinline__ FwdOut decode_step(const elg_rollout_args& A, const Inst& I, const Traj<NCH>& st,
                                              int lane, float* sb, int forced_sel, float uni,
                                              float* full_row /* (N1) or null */, size_t b, size_t r, size_t Rcap) {
    const int N1 = A.N1;
    unsigned long long mk[NCH];
    build_mask<NCH, TSP>(st, I, N1, lane, mk);

    // ---- glimpse query (prefetch early): q = Wq_last [enc[cur]; load]   (models.py:330-333)
    const int cb = (lane & 31) * 4;
    float4 q4 = *reinterpret_cast<const float4*>(I.Q1 + (size_t)st.cur * ELG_E + cb);
    if (TSP) {
        const float4 qf = *reinterpret_cast<const float4*>(I.Q2 + (size_t)st.first * ELG_E + cb);
        q4.x += qf.x; q4.y += qf.y; q4.z += qf.z; q4.w += qf.w;          // TSP/models.py:252-255
    } else {
        const float4 w = *reinterpret_cast<const float4*>(I.wl + cb);
        q4.x = fmaf(st.load, w.x, q4.x); q4.y = fmaf(st.load, w.y, q4.y);
        q4.z = fmaf(st.load, w.z, q4.z); q4.w = fmaf(st.load, w.w, q4.w);
    }

    // ---- k-NN slots + distance penalty + local policy (slot layout)
    float addval = 0.f;          // per-slot additive term (penalty + local score)
    int snid = -1;               // node of this lane's slot
    int ssave = -1;              // what the backward needs to know about the slot
    float sf0 = 0.f, sf1 = 0.f, sf2 = 0.f;
    if (A.has_penalty || A.has_local) {
        const Slots S = slot_setup<NCH, TSP>(I, N1, A.K, A.has_penalty != 0, st, lane, mk, sb, nullptr, A.euclidean != 0, A.ens, A.Kens);
        snid = S.snid;
        ssave = (S.smask && S.snid >= 0) ? -2 : S.snid;              // present but masked (the CVRP depot slot)
        sf0 = S.f0; sf1 = S.f1; sf2 = S.f2;
        float u = 0.f;
        if (A.has_local) u = local_ensemble<TSP>(A, I.loc, lane, S);
        addval = slot_penalty(A, S) + u * A.inv_ens;
    }
    if (TRAIN && A.trSlot && lane < ELG_SLOT_STRIDE) {
        A.trSlot[(b * Rcap + r) * ELG_SLOT_STRIDE + lane] = ssave;
        if (A.trF) {
            float* fr = A.trF + (b * Rcap + r) * (3 * ELG_SLOT_STRIDE) + lane;
            fr[0] = sf0; fr[ELG_SLOT_STRIDE] = sf1; fr[2 * ELG_SLOT_STRIDE] = sf2;
        }
    }

    // ---- glimpse + pointer
    constexpr int NG = GlimpseGroups<NCH, SMALL>::value;
    GlimpseSave<NG> gs;
    float4 o4 = q4;
    o4 = glimpse<NCH, LDSK, NG>(I, N1, lane, q4, mk, TRAIN ? &gs : nullptr);
    if (TRAIN) {
        const int half = lane >> 5, hq = lane & 31, rr = 2 * (lane & 3) + half;
        float* rA = A.trA + ((b * ELG_H + (hq >> 2)) * Rcap + r) * N1;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int row = 8 * k + rr;
            if (row < N1) rA[row] = gs.e[k];
        }
        if (lane < 32) {
            const size_t off = (b * Rcap + r) * ELG_E + cb;
            *reinterpret_cast<float4*>(A.trQ + off) = q4;
            *reinterpret_cast<float4*>(A.trO + off) = o4;
        }
        if (lane == 0 && A.trLoad) A.trLoad[b * Rcap + r] = st.load;
    }
    float s[NCH];
    pointer_scores<NCH, LDSK>(I, N1, lane, o4, sb, s);

    return finish_step<NCH, TSP, TRAIN>(A, N1, lane, sb, mk, s, snid, addval, forced_sel, uni, full_row, b, r, Rcap);
}

template <int NCH, bool TSP, bool LDSK, int WAVES, bool TRAIN, bool SMALL>
__global__ __launch_bounds__(WAVES * 64) void rollout_fwd_kernel(const elg_rollout_args A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N1 = A.N1;
    // XCD-aware unit mapping: the `tiles` workgroups of one instance share an XCD (L2 locality of the
    // instance's Q1 / neighbour tables); placement only affects speed.
    const int G = gridDim.x;
    int u = blockIdx.x;
    if ((G & 7) == 0) u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int b = u / A.tiles, tile = u % A.tiles;
    const int tile_m = (A.M + A.tiles - 1) / A.tiles;
    const int m_lo = tile * tile_m, m_hi = min(A.M, m_lo + tile_m);

    // ---- LDS carve: [K | V | PK(swizzled)] | dem | counter | per-wave scratch
    const int NE = N1 * ELG_E;
    float* p = lds;
    float *sK = nullptr, *sV = nullptr, *sPK = nullptr;
    if (LDSK) { sK = p; sV = p + NE; sPK = p + 2 * NE; p += 3 * NE; }
    float* sdem = p; p += (N1 + 3) & ~3;
    float* sxy = p; if (LDSK) p += (2 * N1 + 3) & ~3;
    p += 4;
    float* sb = p + wave * sb_floats_of(NCH, kmax_of(A));

    const float* gK = A.Kmat + (size_t)b * NE;
    const float* gV = A.Vmat + (size_t)b * NE;
    const float* gPK = A.PK + (size_t)b * NE;
    if (LDSK) {
        const int nt = WAVES * 64;
        for (int i = threadIdx.x; i < NE / 4; i += nt) {
            reinterpret_cast<float4*>(sK)[i] = reinterpret_cast<const float4*>(gK)[i];
            reinterpret_cast<float4*>(sV)[i] = reinterpret_cast<const float4*>(gV)[i];
            const int n = i >> 5, c4 = i & 31;
            reinterpret_cast<float4*>(sPK)[n * 32 + (c4 ^ (n & 31))] = reinterpret_cast<const float4*>(gPK)[i];
        }
    }
    if (!TSP)
        for (int i = threadIdx.x; i < N1; i += WAVES * 64) sdem[i] = A.demand[(size_t)b * N1 + i];
    if (LDSK)
        for (int i = threadIdx.x; i < 2 * N1; i += WAVES * 64) sxy[i] = A.xy[(size_t)b * N1 * 2 + i];
    __syncthreads();

    Inst I;
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

    constexpr int NW = NCH;
    // static round-robin of the tile's trajectories over the waves (all control flow below is
    // wave-uniform: every trajectory-state value is forced into SGPRs with readfirstlane)
    for (int m = m_lo + wave; m < m_hi; m += WAVES) {
        const size_t bm = (size_t)b * A.M + m;

        Traj<NCH> st;
        if (A.use_state) {
            st.cur = __builtin_amdgcn_readfirstlane(A.st_cur[bm]);
            st.cnt = __builtin_amdgcn_readfirstlane(A.st_cnt[bm]);
            st.fin = __builtin_amdgcn_readfirstlane(A.st_fin[bm]);
            st.first = TSP ? __builtin_amdgcn_readfirstlane(A.st_first[bm]) : 0;
            st.load = i2f(__builtin_amdgcn_readfirstlane(f2i(A.st_load[bm])));
            st.len = i2f(__builtin_amdgcn_readfirstlane(f2i(A.st_len[bm])));
            st.cx = i2f(__builtin_amdgcn_readfirstlane(f2i(A.xy[((size_t)b * N1 + st.cur) * 2])));
            st.cy = i2f(__builtin_amdgcn_readfirstlane(f2i(A.xy[((size_t)b * N1 + st.cur) * 2 + 1])));
#pragma unroll
            for (int c = 0; c < NW; ++c) {
                const unsigned long long v = A.st_vis[bm * NW + c];
                const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
                const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
                st.vis[c] = ((unsigned long long)hi << 32) | lo;
            }
        } else {
            st.cur = 0; st.first = 0; st.cnt = 0; st.fin = 0; st.load = 1.0f; st.len = 0.f; st.cx = 0.f; st.cy = 0.f;
#pragma unroll
            for (int c = 0; c < NW; ++c) st.vis[c] = 0ull;
        }

        const int step_cap = A.max_steps > 0 ? A.max_steps : 2 * N1 + 2;      // hard bound: never spin
        for (int steps = 0; steps < step_cap; ++steps) {
            if (A.max_steps <= 0 && st.fin != 0) break;
            const int t = st.cnt;
            if (!A.use_state && t >= A.Tmax) break;
            int sel = 0;
            float pr = 1.0f;
            int fsel = 0;
            const int tf = A.use_state ? steps : t;      // step-wise protocol: `forced` holds this call's actions
            if (A.forced && tf < A.Tforced) fsel = __builtin_amdgcn_readfirstlane(A.forced[bm * A.Tforced + tf]);
            const bool first_move = (!TSP && t <= 1) || (TSP && t == 0);
            if (!A.do_decode) {
                sel = fsel;                                          // CVRPEnv.step with a given action
            } else if (st.fin) {
                sel = 0;                                             // finished: stay at the depot, prob 1
            } else if (first_move) {
                // CVRPModel.py:42-51 (depot, then POMO start) / TSPModel.py:30-34 (POMO start)
                if (A.mode == ELG_MODE_FORCED) sel = fsel;
                else sel = (!TSP && t == 0) ? 0 : __builtin_amdgcn_readfirstlane(A.starts[m]);
            } else {
                float uni = 0.f;
                if (A.mode == ELG_MODE_SAMPLE)
                    uni = A.uniforms ? A.uniforms[bm * A.Tmax + t] : philox_uniform(A.seed, (unsigned)bm, (unsigned)t);
                float* frow = (A.full_probs && t < A.dump_T) ? A.full_probs + (bm * A.dump_T + t) * N1 : nullptr;
                const FwdOut o = decode_step<NCH, TSP, LDSK, TRAIN, SMALL>(A, I, st, lane, sb, fsel, uni, frow, (size_t)b,
                                                                    (size_t)t * A.M + m, (size_t)A.Tmax * A.M);
                sel = __builtin_amdgcn_readfirstlane(o.sel);
                pr = i2f(__builtin_amdgcn_readfirstlane(f2i(o.p)));
            }
            // time index of the outputs: absolute step, or call-relative in the step-wise protocol
            const int tout = A.use_state ? steps : t;
            if (lane == 0 && tout < A.Tmax) {
                if (A.actions) A.actions[bm * A.Tmax + tout] = sel;
                if (A.probs) A.probs[((size_t)b * A.Tmax + tout) * A.M + m] = pr;
            }
            if (A.do_update) env_update<NCH, TSP>(st, I, N1, sel);
            else break;
        }

        if (lane == 0) {
            if (A.reward) A.reward[bm] = -st.len;
            if (A.tlen) A.tlen[bm] = st.cnt;
            if (A.use_state && A.do_update) {
                A.st_cur[bm] = st.cur; A.st_cnt[bm] = st.cnt; A.st_fin[bm] = st.fin;
                if (TSP) A.st_first[bm] = st.first;
                A.st_load[bm] = st.load; A.st_len[bm] = st.len;
#pragma unroll
                for (int c = 0; c < NW; ++c) A.st_vis[bm * NW + c] = st.vis[c];
            }
        }
    }
}

template <int NCH, bool TSP, bool LDSK, int WAVES, bool TRAIN, bool SMALL>
static int launch_fwd_impl(const elg_rollout_args& A, hipStream_t stream) {
    size_t lds = 0;
    if (LDSK) lds += (size_t)3 * A.N1 * ELG_E * 4;
    lds += (size_t)((A.N1 + 3) & ~3) * 4 + 16 + (size_t)WAVES * sb_floats_of(NCH, kmax_of(A)) * 4;
    if (LDSK) lds += (size_t)((2 * A.N1 + 3) & ~3) * 4;
    if (lds > 163840) return fail(ELG_EINVAL, "rollout: LDS budget exceeded");
    auto kern = rollout_fwd_kernel<NCH, TSP, LDSK, WAVES, TRAIN, SMALL>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), 163840)) return fail(ELG_ELAUNCH, "hipFuncSetAttribute failed");
    dim3 grid(A.B * A.tiles), block(WAVES * 64);
    (void)hipGetLastError();
    note_kernel(ELG_KERNEL_WAVE);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, A);
    if (launch_status("rollout_fwd") != ELG_OK) return ELG_ELAUNCH;
    return ELG_OK;
}

// (the cooperative kernel for N1 <= 112 lives in csrc/elg_fwd_coop.hip, its shared device pieces in csrc/elg_coop.h)
// =============================================================================================
// One LDS-staged tile of the glimpse for a one-wavefront trajectory (online softmax over node tiles, flash-attention
// style, same quad layout as the resident kernel): used by rollout_fwd_xl_kernel, whose instances are too large
// for anything node-indexed to live in registers.
// =============================================================================================
template <int NGT>
__device__ __forceinline__ void glimpse_tile(const float* __restrict__ sK, const float* __restrict__ sV, int row0, int N1,
                                             int lane, const float4 q4, unsigned long long w0, unsigned long long w1,
                                             float& m_run, float& l_run, f32x2& acc01, f32x2& acc23) {
    const int half = lane >> 5, hq = lane & 31, ql = lane & 3, cb = hq * 4;
    const bool b0 = ql & 1, b1 = ql & 2;
    const int r = 2 * ql + half;
    const float* Kp = sK + cb;
    const float* Vp = sV + cb;
    float sc[NGT];
#pragma unroll
    for (int k = 0; k < NGT; ++k) {
        sc[k] = ELG_NEG_INF;
        if (row0 + 8 * k >= N1) continue;                       // wave-uniform
        float4 kv[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) kv[jj] = *reinterpret_cast<const float4*>(Kp + (size_t)(8 * k + 2 * jj + half) * ELG_E);
        const float p0 = dot4p(kv[0], q4), p1 = dot4p(kv[1], q4), p2 = dot4p(kv[2], q4), p3 = dot4p(kv[3], q4);
        const float s0 = b0 ? p1 : p0, t0 = b0 ? p0 : p1;
        const float s1 = b0 ? p3 : p2, t1 = b0 ? p2 : p3;
        const float a0 = s0 + quad_xor1(t0), a1 = s1 + quad_xor1(t1);
        const float keep = b1 ? a1 : a0, send = b1 ? a0 : a1;
        const float dotv = keep + quad_xor2(send);
        const unsigned long long w = (k < 8) ? w0 : w1;
        const unsigned byte = (unsigned)(w >> (8 * (k & 7))) & 0xffu;
        const bool masked = (row0 + 8 * k + r >= N1) || ((byte >> r) & 1u);
        sc[k] = masked ? ELG_NEG_INF : dotv * 0.25f;
    }
    float mb = sc[0];
#pragma unroll
    for (int k = 1; k < NGT; ++k) mb = fmaxf(mb, sc[k]);
    mb = fmaxf(mb, quad_xor1(mb));
    mb = fmaxf(mb, quad_xor2(mb));
    mb = x32_max(mb);
    const float m_new = fmaxf(m_run, mb);
    const bool live = m_new > ELG_NEG_INF;
    const float scale = (live && m_run > ELG_NEG_INF) ? __expf(m_run - m_new) : 0.f;
    l_run *= scale;
    acc01 *= scale; acc23 *= scale;
    m_run = m_new;
    float e[NGT];
#pragma unroll
    for (int k = 0; k < NGT; ++k) {
        e[k] = (live && sc[k] > ELG_NEG_INF) ? __expf(sc[k] - m_new) : 0.f;
        l_run += e[k];
    }
#pragma unroll
    for (int k = 0; k < NGT; ++k) {
        if (row0 + 8 * k >= N1) continue;
        float4 vv[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) vv[jj] = *reinterpret_cast<const float4*>(Vp + (size_t)(8 * k + 2 * jj + half) * ELG_E);
#define ELG_VT(JJ)                                                                               \
    {                                                                                            \
        const float a = quad_bcast<JJ>(e[k]);       /* rows past N1 are zero-filled, weight 0 */  \
        const f32x2 a2 = {a, a};                                                                 \
        acc01 = __builtin_elementwise_fma(a2, lo2(vv[JJ]), acc01);                               \
        acc23 = __builtin_elementwise_fma(a2, hi2(vv[JJ]), acc23);                               \
    }
        ELG_VT(0) ELG_VT(1) ELG_VT(2) ELG_VT(3)
#undef ELG_VT
    }
}


// =============================================================================================
// Fragment-major copies of an instance's K / V / PK for rollout_fwd_mt_kernel: every MFMA operand fragment of a 16-node
// tile is one 16-byte load per lane out of a contiguous 1 KB block, so the per-step stream out of L2 moves whole cache
// lines (the row-major tables give 64-byte pieces of 512-byte rows).  NP = 64 NCH padded nodes, NT = NP / 16 tiles;
// lane = 16 hi + lo.  Per instance, in floats / 32-bit words:
//   Kb [h][tile][form][lane][w]: bf16 terms of K[16 tile + lo][16 h + 4 hi + j], j < 4, as the A operand of the score product
//                                S^T = K_h q^T: form 0 = [k1 | k2], form 1 = [k1 | k3] (two values per word)   at 0   (NP * 256 words)
//   Vf [h][tile][lane][j] = V [16 tile + 4 hi + j][16 h + lo]  (f32)                                          at NP * 256
//   PKb[tile][kb][term][lane][w] = bf16 terms 1 .. 3 of PK[16 tile + lo][32 kb + 8 hi + 2 w, + 1]            at NP * 384  (NP * 192 words)
// Both products with a per-launch constant operand -- the glimpse scores K q^T and the pointer scores PK o -- run on
// v_mfma_f32_16x16x32_bf16 over three bf16 terms per operand (csrc/elg_bf16.h: what is dropped is 2^-24 of a product, an f32
// rounding): 3 instructions of 16 cycles per (node tile, head) where the f32 form took 4 of 32, 24 per (node tile, 128 channels)
// where it took 32 of 32.  Both phases kept the matrix pipe ~70 % busy.  The weights x values product stays on f32 MFMAs (its
// left operand is formed per step: splitting it would cost what the faster instruction saves).
// Rows past N1 are zero (their nodes are masked).
// =============================================================================================
constexpr int MT_KB_WORDS = 256;                              // words of Kb per padded node
constexpr int MT_PKB_WORDS = 192;                             // words of PKb per padded node
__global__ __launch_bounds__(256) void mt_repack_kernel(const float* __restrict__ K, const float* __restrict__ V,
                                                        const float* __restrict__ PK, float* __restrict__ F, int N1, int NP) {
    const int b = blockIdx.y;
    const int per = NP * 32;                                  // float4s per f32 table
    const int o = blockIdx.x * 256 + threadIdx.x;
    const size_t NE = (size_t)N1 * ELG_E;
    float* Fb = F + (size_t)b * NP * (MT_KB_WORDS + ELG_E + MT_PKB_WORDS);
    if (o >= 2 * per) {
        // PKb: one thread per (tile, kb, lane) writes the three term quads
        const int r = o - 2 * per;
        if (r >= NP * 16) return;
        const int lane = r & 63, lo = lane & 15, hi = lane >> 4, kb = (r >> 6) & 3, tile = r >> 8, n = 16 * tile + lo;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = n < N1 ? PK[b * NE + (size_t)n * ELG_E + 32 * kb + 8 * hi + j] : 0.f;
        unsigned pa[6], pb[6];
        bf_terms<3>(x[0], x[1], x[2], x[3], pa);
        bf_terms<3>(x[4], x[5], x[6], x[7], pb);
        uint4* dst = reinterpret_cast<uint4*>(Fb + (size_t)NP * (MT_KB_WORDS + ELG_E)) + ((size_t)(tile * 4 + kb) * 3) * 64 + lane;
#pragma unroll
        for (int t = 0; t < 3; ++t) dst[t * 64] = make_uint4(pa[2 * t], pa[2 * t + 1], pb[2 * t], pb[2 * t + 1]);
        return;
    }
    const int which = o / per, r = o % per;
    const int lane = r & 63, lo = lane & 15, hi = lane >> 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (which == 0) {
        const int NT = NP >> 4, h = (r >> 6) / NT, tile = (r >> 6) % NT, n = 16 * tile + lo;
        if (n < N1) v = *reinterpret_cast<const float4*>(K + b * NE + (size_t)n * ELG_E + 16 * h + 4 * hi);
        unsigned pt[6];
        bf_terms<3>(v.x, v.y, v.z, v.w, pt);
        uint4* dst = reinterpret_cast<uint4*>(Fb) + ((size_t)(h * NT + tile) * 2) * 64 + lane;
        dst[0] = make_uint4(pt[0], pt[1], pt[2], pt[3]);
        dst[64] = make_uint4(pt[0], pt[1], pt[4], pt[5]);
        return;
    }
    {
        const int NT = NP >> 4, h = (r >> 6) / NT, tile = (r >> 6) % NT, n = 16 * tile + 4 * hi;
        const float* src = V + b * NE + 16 * h + lo;
        if (n < N1) v.x = src[(size_t)n * ELG_E];
        if (n + 1 < N1) v.y = src[(size_t)(n + 1) * ELG_E];
        if (n + 2 < N1) v.z = src[(size_t)(n + 2) * ELG_E];
        if (n + 3 < N1) v.w = src[(size_t)(n + 3) * ELG_E];
    }
    reinterpret_cast<float4*>(Fb + (size_t)NP * MT_KB_WORDS)[r] = v;
}

// bf16 throughput mode (elg_rollout_args.precision = 1) of the streaming kernel: one bf16 term per table entry, per instance
// (32-bit words; NT = NP / 16 node tiles):
//   Kb1[h][tile][lane][2]  = K[16 tile + lo][16 h + 4 hi + j], j < 4 (two values per word): k-slots (hi, 0..3) of the A operand
//                            of S^T = K_h q^T, k-slots (hi, 4..7) are zero (16 channels per head, the instruction contracts 32)
//   Vb [h][pair][lane][4]  = V[32 pair + 4 hi + j][16 h + lo] (j < 4) | V[32 pair + 16 + 4 hi + j][16 h + lo]: the A operand of
//                            O^T += V_h^T P^T over TWO node tiles, whose score D tiles side by side are the B operand
//   PKb1[tile][kb][lane][4] = PK[16 tile + lo][32 kb + 8 hi + j], j < 8
// 12 KB per node tile and step out of L2 instead of 36 KB; 1 + 1/2 + 4 instructions of 16 cycles per (node tile, head / 128
// channels) instead of 3 x 16 + 4 x 32 + 24 x 16 cycles.  Inference only (a training forward at these sizes computes in f32).
constexpr int MT_BF_WORDS = 192;                              // words per padded node: 64 (K) + 64 (V) + 64 (PK)
__global__ __launch_bounds__(256) void mt_repack_bf16_kernel(const float* __restrict__ K, const float* __restrict__ V,
                                                             const float* __restrict__ PK, unsigned* __restrict__ F, int N1, int NP) {
    const int b = blockIdx.y;
    const int NT = NP >> 4;
    const int o = blockIdx.x * 256 + threadIdx.x;
    const size_t NE = (size_t)N1 * ELG_E;
    unsigned* Fb = F + (size_t)b * NP * MT_BF_WORDS;
    const int lane = o & 63, lo = lane & 15, hi = lane >> 4;
    const int nK = 8 * NT * 64, nV = 8 * (NT / 2) * 64, nP = NT * 4 * 64;
    if (o < nK) {
        const int h = (o >> 6) / NT, tile = (o >> 6) % NT, n = 16 * tile + lo;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N1) v = *reinterpret_cast<const float4*>(K + b * NE + (size_t)n * ELG_E + 16 * h + 4 * hi);
        reinterpret_cast<uint2*>(Fb)[o] = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
    } else if (o < nK + nV) {
        const int r = o - nK, h = (r >> 6) / (NT / 2), pair = (r >> 6) % (NT / 2);
        const float* src = V + b * NE + 16 * h + lo;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = 32 * pair + 16 * (j >> 2) + 4 * hi + (j & 3);
            x[j] = n < N1 ? src[(size_t)n * ELG_E] : 0.f;
        }
        reinterpret_cast<uint4*>(Fb + (size_t)NP * 64)[r] = make_uint4(pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3]), pk_bf16(x[4], x[5]), pk_bf16(x[6], x[7]));
    } else if (o < nK + nV + nP) {
        const int r = o - nK - nV, kb = (r >> 6) & 3, tile = r >> 8, n = 16 * tile + lo;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = n < N1 ? PK[b * NE + (size_t)n * ELG_E + 32 * kb + 8 * hi + j] : 0.f;
        reinterpret_cast<uint4*>(Fb + (size_t)NP * 128)[r] = make_uint4(pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3]), pk_bf16(x[4], x[5]), pk_bf16(x[6], x[7]));
    }
}

// =============================================================================================
// rollout_fwd_mt_kernel: 128 < N1 <= 1024 (TSP-200/500, VRPLIB X).  16 lockstep trajectories per workgroup.
//   owners   (wave w owns trajectories w and w + 8; one wavefront per trajectory, the code of rollout_fwd_kernel):
//            feasibility mask, query row, k-NN slots, local policy  ->  exchange rows in LDS
//   glimpse  (wave h = head h, v_mfma_f32_16x16x4_f32): S^T = K_h q^T tile by tile over the nodes with the K / V operand
//            fragments read straight from L2 (16-byte K loads, no LDS staging, no barrier per tile), online softmax in the
//            exp2 domain over pairs of node tiles, O^T += V_h^T P^T -- the D tile of the first product is the B operand of
//            the second
//   pointer  (wave w: node tiles w, w + 8, ...): s^T = PK o^T, 32 MFMAs per node tile, scores to an LDS row per trajectory
//   owners   clip, mask, softmax, choice, environment transition
// The node-tiled kernel this replaces shared LDS-staged K / V / PK tiles between 8 one-wavefront trajectories and did the
// glimpse and pointer products on the VALU: two barriers + a global->LDS round trip per tile, 65 us per step of 8.
// =============================================================================================
// wave-uniform scalar state of a trajectory of rollout_fwd_mt_kernel (Traj<> without the visited words, which live in LDS)
struct MtTraj {
    int cur, first, cnt, fin;
    float load, len, cx, cy;
};

// build_mask() with the visited words in LDS
// (also leaves the mask in ADDITIVE form -- 0 open, -inf closed, nodes past N1 closed -- in the trajectory's score row `srow`,
// which is free until this step's pointer phase refills it: the glimpse takes it as the accumulator input of its S = K q^T MFMAs)
template <int NCH, bool TSP>
__device__ __forceinline__ void mt_build_mask(const MtTraj& st, const unsigned long long* vis, const Inst& I, int N1, int lane,
                                              unsigned long long (&mk)[NCH], float* srow) {
    const float lim = __fadd_rn(st.load, 1e-6f);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int n = lane + 64 * ch;
        bool m = true;
        if (n < N1) {
            m = (vis[ch] >> lane) & 1ull;
            if (!TSP) {
                m = m || (lim < I.dem[n]);
                if (n == 0 && st.fin) m = false;
            }
        }
        srow[n] = m ? ELG_NEG_INF : 0.f;
        mk[ch] = __ballot(m);
    }
}

// The mask of a step as words (round 4; mt_build_mask above is the lane-per-node form the first version of the kernel used):
// lane c < NCH returns word c -- visited | (CVRP) demand above the load | nodes past N1 -- with the depot open on a finished
// trajectory (CVRPEnv.py:214-232).  One ballot + two selects per 64 nodes for CVRP, nothing per node for TSP.
template <int NCH, bool TSP>
__device__ __forceinline__ unsigned long long mt_mask_words(const MtTraj& st, const unsigned long long* vis, const float* sdem,
                                                            int N1, int lane) {
    unsigned long long w = lane < NCH ? vis[lane] : 0ull;
    if (!TSP) {
        // the ballot of chunk ch is selected into lane ch.  `lane` is made opaque per call: shared between the wave's trajectories,
        // the NCH lane tests become NCH scalar-register pairs that live across the calls and spill.  Nodes past N1 read LDS behind
        // the demand row (the per-wave scratch: inside the allocation) -- their bits are closed below whatever the compare says.
        const float lim = __fadd_rn(st.load, 1e-6f);
        int lc = lane;
        asm volatile("" : "+v"(lc));
        unsigned lo = 0u, hi = 0u;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const unsigned long long bal = __ballot(lim < sdem[lc + 64 * ch]);
            const bool mine = lc == ch;
            lo = mine ? (unsigned)bal : lo;
            hi = mine ? (unsigned)(bal >> 32) : hi;
        }
        w |= ((unsigned long long)hi << 32) | lo;
    }
    const int rem = N1 - 64 * lane;                                   // nodes of word `lane` that exist
    w |= rem >= 64 ? 0ull : (rem <= 0 ? ~0ull : ~((1ull << rem) - 1ull));
    if (!TSP && lane == 0 && st.fin) w &= ~1ull;
    return w;
}
// additive mask row (0 open / -inf closed) from the words: lane l writes the nodes 256 c + 4 l .. + 3 with one 16-byte store
template <int NCH>
__device__ __forceinline__ void mt_fill_additive(const unsigned long long* mkw, float* srow, int lane) {
#pragma unroll
    for (int c = 0; c < NCH / 4; ++c) {
        const int n0 = 256 * c + 4 * lane;
        const unsigned bits = (unsigned)(mkw[n0 >> 6] >> (n0 & 63));
        *reinterpret_cast<float4*>(srow + n0) = make_float4((bits & 1u) ? ELG_NEG_INF : 0.f, (bits & 2u) ? ELG_NEG_INF : 0.f,
                                                            (bits & 4u) ? ELG_NEG_INF : 0.f, (bits & 8u) ? ELG_NEG_INF : 0.f);
    }
}

// env_update() with the visited words in LDS (same roundings: one fp32 subtraction for the load, dist2d for the length)
// (sxy = xy[sel]: requested by the caller for all of the wave's trajectories before the first of them is updated)
template <int NCH, bool TSP>
__device__ __forceinline__ void mt_env_update(MtTraj& st, unsigned long long* vis, const Inst& I, int N1, int sel, int lane, float2 sxy) {
    const float sx = i2f(__builtin_amdgcn_readfirstlane(f2i(sxy.x))), sy = i2f(__builtin_amdgcn_readfirstlane(f2i(sxy.y)));
    if (st.cnt > 0) st.len += dist2d(st.cx, st.cy, sx, sy);
    st.cx = sx; st.cy = sy;
    if (TSP) {
        if (st.cnt == 0) st.first = sel;
    } else {
        st.load = (sel == 0) ? 1.0f : __fsub_rn(st.load, I.dem[sel]);
    }
    if (lane == 0) {
        if (TSP) {
            vis[sel >> 6] |= 1ull << (sel & 63);
        } else {
            // depot counts as visited exactly while the trajectory stands on it (CVRPEnv.py:214-216)
            if (sel >= 64) vis[sel >> 6] |= 1ull << (sel & 63);
            unsigned long long w0 = vis[0];
            if (sel < 64) w0 |= 1ull << sel;
            if (sel == 0) w0 |= 1ull; else w0 &= ~1ull;
            vis[0] = w0;
        }
    }
    st.cur = sel;
    st.cnt += 1;
    if (TSP) {
        if (st.cnt == N1) {                    // close the tour (TSPEnv.py:166 roll(-1))
            const float fx = I.xy[2 * st.first], fy = I.xy[2 * st.first + 1];
            st.len += dist2d(sx, sy, fx, fy);
            st.fin = 1;
        }
    } else {
        wave_lds_fence();
        bool ok = true;
        if (lane < NCH) {
            const int rem = N1 - 64 * lane;
            if (rem > 0) {
                const unsigned long long full = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
                ok = (vis[lane] & full) == full;
            }
        }
        if (__ballot(ok) == ~0ull) st.fin = 1;  // CVRPEnv.py:226-228
    }
}

// TRAIN (round 3): the rows a backward over saved rows needs, time-major (r = t M + m) as the cooperative kernel writes them:
// query q and glimpse output o, the glimpse's log2-sum-exp per head (in units of s log2(e) / 4), the row's mask words
// (NCH 64-bit words), load, k-NN slot ids + slot features, the softmax x clip Jacobian row and its value at the chosen node.
template <int NCH, bool TSP, int NG, bool TRAIN, bool BF = false>
__global__ __launch_bounds__(512) void rollout_fwd_mt_kernel(const elg_rollout_args A) {
    static_assert(!(BF && TRAIN), "the bf16 mode of the streaming kernel is inference only");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NTR = 16 * NG, NOWN = 2 * NG, QP = 132, SP = 64 * NCH + 4;
    constexpr int OBP = 68;                                 // words of a bf16 term row of o (128 channels + 8: conflict-free b128 reads)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int N1 = A.N1;
    const int NTn = (N1 + 15) >> 4;
    const int G = gridDim.x;
    int u = blockIdx.x;
    if ((G & 7) == 0) u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int b = u / A.tiles, tile = u % A.tiles;
    const int m_base = tile * NTR;
    // LDS: query / glimpse-output rows | score rows | mask words | visited words | slot blocks | local tables | demand |
    //      per-wave slot scratch
    float* sQ = lds;                                       // query rows; after the glimpse: the three bf16 term planes of o
    unsigned* sOb = reinterpret_cast<unsigned*>(lds);      //   [term][trajectory][OBP]
    float* sSc = sQ + NTR * 3 * OBP;                       // score rows (pointer -> choice)
    unsigned long long* sMaskW = reinterpret_cast<unsigned long long*>(sSc + NTR * SP);
    unsigned long long* sVis = sMaskW + NTR * NCH;
    unsigned long long* sSlotW = sVis + NTR * NCH;                      // bit n: node n carries a k-NN slot of this step (choice phase)
    float* sX = reinterpret_cast<float*>(sSlotW + NTR * NCH);           // slot blocks (owners -> local policy -> owners)
    float* sT = sX + (A.has_local ? NTR * CO_XP : 0);                   // folded local-policy tables
    float* sdem = sT + (A.has_local ? CL_SIZE : 0);
    float* sb = sdem + ((N1 + 3) & ~3) + wave * ELG_SB_MIN;
    if (!TSP)
        for (int i = tid; i < N1; i += 512) sdem[i] = A.demand[(size_t)b * N1 + i];
    for (int i = tid; i < 2 * NTR * NCH; i += 512) sVis[i] = 0ull;      // visited words | slot words
    if (A.has_local) co_stage_local(A.loc, sT, tid, 512);
    __syncthreads();
    const size_t NE = (size_t)N1 * ELG_E;
    Inst I;
    I.K = nullptr; I.V = nullptr; I.PK = nullptr;
    I.pb = A.pb + (size_t)b * N1;
    I.Q1 = A.Q1 + b * NE;
    I.Q2 = TSP ? A.Q2 + b * NE : nullptr;
    I.wl = A.wl;
    I.xy = A.xy + (size_t)b * N1 * 2;
    I.dem = sdem;
    I.nidx = A.nbr_idx + (size_t)b * N1 * N1;
    I.ndist = A.nbr_dist + (size_t)b * N1 * N1;
    I.ntheta = A.nbr_theta + (size_t)b * N1 * N1;
    I.loc = A.loc;
    constexpr int NP = 64 * NCH, NT = 4 * NCH;              // padded nodes / tiles of the fragment-major tables
    const float* gF = A.scratch + (size_t)b * NP * (BF ? MT_BF_WORDS : MT_KB_WORDS + ELG_E + MT_PKB_WORDS);
    // bf16 mode: [h][tile][lane] uint2 | [h][pair][lane] uint4 | [tile][kb][lane] uint4
    const uint2* gKb = reinterpret_cast<const uint2*>(gF) + (size_t)wave * NT * 64 + lane;
    const uint4* gVb = reinterpret_cast<const uint4*>(gF + (size_t)NP * 64) + (size_t)wave * (NT / 2) * 64 + lane;
    const uint4* gPb = reinterpret_cast<const uint4*>(gF + (size_t)NP * 128) + lane;
    constexpr int TU = NG == 1 ? 4 : 2;                    // node tiles per softmax update of the glimpse
    const uint4* gK = reinterpret_cast<const uint4*>(gF) + (size_t)wave * NT * 2 * 64 + lane;                // head = wave: [tile][form][lane]
    const float4* gV = reinterpret_cast<const float4*>(gF + (size_t)NP * MT_KB_WORDS) + (size_t)wave * NT * 64 + lane;
    const uint4* gPK = reinterpret_cast<const uint4*>(gF + (size_t)NP * (MT_KB_WORDS + ELG_E)) + lane;     // [tile][kb][term][lane]
    const int step_cap = TSP ? N1 : 2 * N1 + 2;
    const size_t Rcap = (size_t)A.Tmax * A.M;

    // the wave's trajectories: q = wave + 8 j (row q % 16 of MFMA group q / 16); every loop over j is unrolled, so st[] stays
    // in (scalar) registers
    MtTraj st[NOWN];
    bool has[NOWN];
    size_t bm[NOWN];
    StampCtx sc_;
#ifdef ELG_STAMPS
    for (int i = 0; i < 16; ++i) sc_.acc[i] = 0.f;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc_.last) :: "memory");
#endif
#pragma unroll
    for (int j = 0; j < NOWN; ++j) {
        has[j] = m_base + wave + 8 * j < A.M;
        bm[j] = (size_t)b * A.M + (has[j] ? m_base + wave + 8 * j : 0);
        st[j].cur = 0; st[j].first = 0; st[j].cnt = 0; st[j].fin = has[j] ? 0 : 1;
        st[j].load = 1.0f; st[j].len = 0.f; st[j].cx = 0.f; st[j].cy = 0.f;
    }
    for (int t = 0; t < step_cap && t < A.Tmax; ++t) {
        const bool first_move = (!TSP && t <= 1) || (TSP && t == 0);
        bool act[NOWN], dec[NOWN];
        bool any = false;
#pragma unroll
        for (int j = 0; j < NOWN; ++j) {
            act[j] = has[j] && !st[j].fin;
            dec[j] = act[j] && !first_move;
            any = any || act[j];
        }
        if (!__syncthreads_or(any ? 1 : 0)) break;
        ELG_STAMP(sc_, 8);
        int sel[NOWN], snid[NOWN];
        float pr[NOWN], addv[NOWN];
#pragma unroll
        for (int j = 0; j < NOWN; ++j) { sel[j] = 0; snid[j] = -1; pr[j] = 1.0f; addv[j] = 0.f; }
        if (!first_move) {
            // ================= owners: masks, query rows, k-NN slots =================
            // Written in STAGES over the wave's trajectories (round 4): every stage is an LDS or L2 round trip, and the trajectories'
            // chains are independent -- one after the other they exposed every latency NOWN times.
            {
                int ln = lane;                                  // (opaque per step: addresses derived from it are re-formed here instead of
                asm volatile("" : "+v"(ln));                    //  living in registers across the matrix phases, where there are none to spare)
                constexpr int S0 = TSP ? 0 : 1;
                const int cb = (ln & 31) * 4;
                // ---- stage 0: everything that comes from L2 / HBM and depends on the current node only is requested first -- the
                // query rows (q = Wq_last [enc[cur]; load], models.py:330-333; combined at the end of the phase: a use here would wait
                // for the loads before the mask work) and the first 64 entries of cur's sorted neighbour list
                float4 qa[NOWN], qb[NOWN];
                int nid0[NOWN];
                float nd0[NOWN], nth0[NOWN];
                const bool walk = A.has_local || A.has_penalty;
#pragma unroll
                for (int j = 0; j < NOWN; ++j) {
                    qa[j] = make_float4(0.f, 0.f, 0.f, 0.f); qb[j] = qa[j];
                    nid0[j] = 0; nd0[j] = 0.f; nth0[j] = 0.f;
                    if (!dec[j]) continue;
                    qa[j] = *reinterpret_cast<const float4*>(I.Q1 + (size_t)st[j].cur * ELG_E + cb);
                    qb[j] = TSP ? *reinterpret_cast<const float4*>(I.Q2 + (size_t)st[j].first * ELG_E + cb) : *reinterpret_cast<const float4*>(I.wl + cb);
                    if (walk && ln < N1) {
                        const size_t e = (size_t)st[j].cur * N1 + ln;
                        nid0[j] = I.nidx[e]; nd0[j] = I.ndist[e]; nth0[j] = I.ntheta[e];
                    }
                }
                // ---- stage 1: mask words (CVRPEnv.py:214-232 / TSPEnv.py:120): lane c < NCH assembles word c = visited | demand >
                // load | nodes past N1
                unsigned mk0[NOWN];                             // low half of word 0 (the depot's bit)
#pragma unroll
                for (int j = 0; j < NOWN; ++j) {
                    const int q = wave + 8 * j;
                    unsigned long long mword = ~0ull;           // not decoding: every node closed
                    if (dec[j]) mword = mt_mask_words<NCH, TSP>(st[j], sVis + q * NCH, sdem, N1, ln);
                    if (ln < NCH) sMaskW[q * NCH + ln] = mword;
                    mk0[j] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mword);
                    if (TRAIN && dec[j] && ln < NCH) A.trMask[((size_t)b * Rcap + (size_t)t * A.M + m_base + q) * NCH + ln] = mword;
                }
                wave_lds_fence();
                // ---- stage 2: the additive form (0 open / -inf closed) goes to the score rows 16 bytes per lane
#pragma unroll
                for (int j = 0; j < NOWN; ++j) mt_fill_additive<NCH>(sMaskW + (wave + 8 * j) * NCH, sSc + (wave + 8 * j) * SP, ln);
                ELG_STAMP(sc_, 9);
                // ---- stages 3 + 4: k-NN slots (knn_slots / slot_setup of elg_rollout.h for the trajectories jlo .. jhi - 1 side by
                // side): the first K open customers of cur's sorted neighbour list, compacted to slot order through the
                // trajectory's slot block (with the local policy: the regions its features do not use yet) or the wave's scratch
                auto slots_for = [&](int jlo, int jhi) {
                    int found[NOWN];
                    bool go[NOWN];
                    int W[NOWN];                                // float offset of the scratch in the workgroup's LDS (not a pointer: a select of
                                                                // LDS pointers turns generic and trips the compiler's address-space cast)
                    int oD, oT, oN;                             // where distance | theta | node id of slot i wait for the slot lanes
                    if (A.has_local) { oD = CO_XPEN; oT = CO_XU; oN = CO_XS; } else { oD = 0; oT = ELG_SLOT_STRIDE; oN = 2 * ELG_SLOT_STRIDE; }
#pragma unroll
                    for (int j = 0; j < NOWN; ++j) {
                        found[j] = 0;
                        go[j] = dec[j] && j >= jlo && j < jhi;
                        W[j] = A.has_local ? (int)(sX - lds) + (wave + 8 * j) * CO_XP : (int)(sb - lds);
                    }
#pragma unroll 1
                    for (int ch = 0; 64 * ch < N1; ++ch) {
                        bool need[NOWN], any = false;
#pragma unroll
                        for (int j = 0; j < NOWN; ++j) { need[j] = go[j] && found[j] < A.K; any = any || need[j]; }
                        if (!any) break;
                        const int i = ln + 64 * ch;
                        const bool valid = i < N1;
                        int nid[NOWN];
                        float nd[NOWN], nth[NOWN];
#pragma unroll
                        for (int j = 0; j < NOWN; ++j) {
                            nid[j] = nid0[j]; nd[j] = nd0[j]; nth[j] = nth0[j];            // chunk 0: requested at the top of the phase
                            if (ch > 0 && need[j] && valid) {
                                const size_t e = (size_t)st[j].cur * N1 + i;
                                nid[j] = I.nidx[e]; nd[j] = I.ndist[e]; nth[j] = I.ntheta[e];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < NOWN; ++j) {
                            if (!need[j]) continue;
                            bool cand = valid && !((sMaskW[(wave + 8 * j) * NCH + (nid[j] >> 6)] >> (nid[j] & 63)) & 1ull);
                            if (!TSP) cand = cand && (nid[j] != 0);
                            const unsigned long long bal = __ballot(cand);
                            const int rank = found[j] + lanes_below(bal);
                            if (cand && rank < A.K) {
                                lds[W[j] + oD + S0 + rank] = nd[j];
                                lds[W[j] + oT + S0 + rank] = nth[j];
                                reinterpret_cast<int*>(lds)[W[j] + oN + S0 + rank] = nid[j];
                            }
                            found[j] += __popcll(bal);
                        }
                    }
                    wave_lds_fence();
                    float sd[NOWN], sth[NOWN], dmax[NOWN];
                    bool cust[NOWN];
#pragma unroll
                    for (int j = 0; j < NOWN; ++j) {
                        sd[j] = 0.f; sth[j] = 0.f; dmax[j] = 0.f; cust[j] = false;
                        if (!go[j]) continue;
                        const int k = found[j] < A.K ? found[j] : A.K;
                        cust[j] = (ln >= S0) && (ln < S0 + k);
                        if (cust[j]) { sd[j] = lds[W[j] + oD + ln]; sth[j] = lds[W[j] + oT + ln]; snid[j] = reinterpret_cast<const int*>(lds)[W[j] + oN + ln]; }
                        if (k > 0) dmax[j] = lds[W[j] + oD + S0 + k - 1];
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int j = 0; j < NOWN; ++j) {
                        if (!go[j]) continue;
                        const int q = wave + 8 * j;
                        if (!TSP && ln == 0) snid[j] = 0;                                 // depot slot
                        float pen = 0.f;
                        if (A.has_penalty && cust[j]) {
                            if (TSP) pen = -(sd[j] / (dmax[j] + 1e-6f));                  // TSP/models.py:290
                            else pen = (dmax[j] != 0.f) ? -(sd[j] / dmax[j]) : -sd[j];    // models.py:379-405 (no epsilon)
                        }
                        addv[j] = pen;
                        const float nf = dmax[j] + 1e-6f;                                 // models.py:79 / TSP :72
                        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
                        if (cust[j]) {
                            if (A.euclidean) {                                            // models.py:95-125: relative (x, y) / norm
                                const float cx = I.xy[2 * st[j].cur], cy = I.xy[2 * st[j].cur + 1];
                                f0 = __fsub_rn(I.xy[2 * snid[j]], cx) / nf;
                                f1 = __fsub_rn(I.xy[2 * snid[j] + 1], cy) / nf;
                            } else { f0 = sd[j] / nf; f1 = sth[j]; }
                            if (!TSP) f2 = sdem[snid[j]] / st[j].load;                    // CVRPEnv.py:315-316
                        }
                        bool smask = !cust[j];
                        if (!TSP && ln == 0) smask = mk0[j] & 1u;                         // depot slot carries the depot's mask
                        const int code = smask ? (snid[j] >= 0 ? -2 : -1) : snid[j];
                        if (A.has_local && ln < ELG_SLOT_STRIDE) {        // slot block of the local policy (layout of the coop kernel)
                            float* X = sX + q * CO_XP;
                            X[CO_XF + ln] = f0; X[CO_XF + ELG_SLOT_STRIDE + ln] = f1; X[CO_XF + 2 * ELG_SLOT_STRIDE + ln] = f2;
                            reinterpret_cast<int*>(X)[CO_XS + ln] = code;
                        }
                        if (TRAIN && A.trSlot && ln < ELG_SLOT_STRIDE) {
                            const size_t rrow = (size_t)b * Rcap + (size_t)t * A.M + m_base + q;
                            A.trSlot[rrow * ELG_SLOT_STRIDE + ln] = (smask && snid[j] >= 0) ? -2 : snid[j];
                            if (A.trF) {
                                float* fr = A.trF + rrow * (3 * ELG_SLOT_STRIDE) + ln;
                                fr[0] = f0; fr[ELG_SLOT_STRIDE] = f1; fr[2 * ELG_SLOT_STRIDE] = f2;
                            }
                        }
                    }
                };
                ELG_STAMP(sc_, 0);
                if (A.has_local || A.has_penalty) {
                    // (without the local policy there are no slot blocks: one scratch per wave, one trajectory at a time.  ONE call
                    // site: a lambda that is not inlined receives its captured LDS pointers as generic ones)
                    const int npass = A.has_local ? 1 : NOWN;
#pragma unroll 1
                    for (int ps = 0; ps < npass; ++ps) {
                        slots_for(A.has_local ? 0 : ps, A.has_local ? NOWN : ps + 1);
                        wave_lds_fence();
                    }
                }
                ELG_STAMP(sc_, 10);
#pragma unroll
                for (int j = 0; j < NOWN; ++j) {
                    const int q = wave + 8 * j;
                    if (!dec[j] && A.has_local && ln < ELG_SLOT_STRIDE) {
                        float* X = sX + q * CO_XP;
                        X[CO_XF + ln] = 0.f; X[CO_XF + ELG_SLOT_STRIDE + ln] = 0.f; X[CO_XF + 2 * ELG_SLOT_STRIDE + ln] = 0.f;
                        reinterpret_cast<int*>(X)[CO_XS + ln] = -1;
                    }
                    float4 q4 = qa[j];
                    if (TSP) { q4.x += qb[j].x; q4.y += qb[j].y; q4.z += qb[j].z; q4.w += qb[j].w; }       // TSP/models.py:252-255
                    else {
                        q4.x = fmaf(st[j].load, qb[j].x, q4.x); q4.y = fmaf(st[j].load, qb[j].y, q4.y);
                        q4.z = fmaf(st[j].load, qb[j].z, q4.z); q4.w = fmaf(st[j].load, qb[j].w, q4.w);
                    }
                    if (!dec[j]) q4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ln < 32) *reinterpret_cast<float4*>(sQ + q * QP + 4 * ln) = q4;
                    if (TRAIN && dec[j]) {
                        const size_t rrow = (size_t)b * Rcap + (size_t)t * A.M + m_base + q;                 // this trajectory's row of step t
                        if (ln < 32) *reinterpret_cast<float4*>(A.trQ + rrow * ELG_E + cb) = q4;
                        if (ln == 0 && A.trLoad) A.trLoad[rrow] = st[j].load;
                    }
                }
            }
            ELG_STAMP(sc_, 0);
            __syncthreads();
            ELG_STAMP(sc_, 1);
            // ================= glimpse: wave = head; every K / V fragment serves the NG groups of 16 trajectories =================
            // (waves 4-7 are the younger wave of their SIMD and lose the issue arbitration for the whole phase -- phase clock: 51 % of a
            // step against 35 % for waves 0-3, which then idle at the barrier.  A static s_setprio 1 / 2 for the younger half through
            // this phase was measured in round 6: TSP-500 19.80 -> 19.84 / 19.86 ms, X-n1001-sized CVRP 163.1 -> 165.4 / 165.7 ms:
            // the phase is bound by the SIMD's matrix pipe, priority only moves the wait from one wave to the other.)
            {
                const float cs = 0.25f * 1.4426950408889634f;
                u32x4 q11[NG], q22[NG], q31[NG];            // the query's bf16 terms as the B operand: [q1 | q1], [q2 | q2], [q3 | q1]
                float mrun[NG], lrun[NG];
                f32x4c o[NG], o2[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float4 qv = *reinterpret_cast<const float4*>(sQ + (16 * g + lo) * QP + 16 * wave + 4 * hi);     // trajectory 16 g + lo
                    unsigned pt[6];
                    bf_terms<3>(qv.x, qv.y, qv.z, qv.w, pt);
                    q11[g] = u32x4{pt[0], pt[1], pt[0], pt[1]};
                    q22[g] = u32x4{pt[2], pt[3], pt[2], pt[3]};
                    q31[g] = u32x4{pt[4], pt[5], pt[0], pt[1]};
                    mrun[g] = -1e30f; lrun[g] = 0.f;
                    o[g] = f32x4c{0.f, 0.f, 0.f, 0.f}; o2[g] = f32x4c{0.f, 0.f, 0.f, 0.f};
                }
                __syncthreads();        // every head has its queries: the rows are free for the term planes of o (cheap: the waves
                                        // left the previous barrier a few instructions ago)
                // local policy, stage 1: head unit (group w >> 2, head w & 3) on wave w; o' waits in the per-wave scratch area
                // (free between the owners' phases) for the tail in the pointer phase
                if (A.has_local && wave < 4 * NG)
                    co_local_head_call(sT, sX + (wave >> 2) * 16 * CO_XP, sdem + ((N1 + 3) & ~3) + (wave >> 2) * 512, wave & 3, lo, hi);
                if constexpr (BF) {
                    // ---- bf16 mode: one instruction per score tile, one per PAIR of node tiles for the output
                    u32x4 qb1[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) qb1[g] = u32x4{q11[g][0], q11[g][1], 0u, 0u};
                    uint2 kf[TU], kn[TU];
                    uint4 vf[TU / 2], vn[TU / 2];
                    auto loadb = [&](int nt0, uint2 (&kk)[TU], uint4 (&vv)[TU / 2]) {
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) kk[u4] = gKb[(nt0 + u4) * 64];
#pragma unroll
                        for (int u2 = 0; u2 < TU / 2; ++u2) vv[u2] = gVb[(nt0 / 2 + u2) * 64];
                    };
                    loadb(0, kf, vf);
#pragma unroll 1
                    for (int nt = 0; nt < NTn; nt += TU) {
                        loadb(min(nt + TU, NT - TU), kn, vn);
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4c S[NG][TU];
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int u4 = 0; u4 < TU; ++u4) {
                                const float4 m4 = *reinterpret_cast<const float4*>(sSc + (16 * g + lo) * SP + 16 * (nt + u4) + 4 * hi);
                                S[g][u4] = mfma_bf(u32x4{kf[u4].x, kf[u4].y, 0u, 0u}, qb1[g], f32x4c{m4.x, m4.y, m4.z, m4.w});
                            }
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            float tm = ELG_NEG_INF;
#pragma unroll
                            for (int u4 = 0; u4 < TU; ++u4)
#pragma unroll
                                for (int i = 0; i < 4; ++i) tm = fmaxf(tm, S[g][u4][i]);
                            tm = quarters_max(tm);
                            const float mnew = fmaxf(mrun[g], tm);
                            const float sc = __builtin_amdgcn_exp2f((mrun[g] - mnew) * cs);
                            mrun[g] = mnew;
                            lrun[g] *= sc;
#pragma unroll
                            for (int i = 0; i < 4; ++i) { o[g][i] *= sc; o2[g][i] *= sc; }
                            const float cm = -mnew * cs;
#pragma unroll
                            for (int u4 = 0; u4 < TU; ++u4)
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    S[g][u4][i] = __builtin_amdgcn_exp2f(fmaf(S[g][u4][i], cs, cm));
                                    lrun[g] += S[g][u4][i];
                                }
                        }
#pragma unroll
                        for (int g = 0; g < NG; ++g)
#pragma unroll
                            for (int u2 = 0; u2 < TU / 2; ++u2) {
                                const u32x4 pb_ = {pk_bf16(S[g][2 * u2][0], S[g][2 * u2][1]), pk_bf16(S[g][2 * u2][2], S[g][2 * u2][3]),
                                                   pk_bf16(S[g][2 * u2 + 1][0], S[g][2 * u2 + 1][1]), pk_bf16(S[g][2 * u2 + 1][2], S[g][2 * u2 + 1][3])};
                                f32x4c& acc = (u2 & 1) ? o2[g] : o[g];
                                acc = mfma_bf(u32x4{vf[u2].x, vf[u2].y, vf[u2].z, vf[u2].w}, pb_, acc);
                            }
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) kf[u4] = kn[u4];
#pragma unroll
                        for (int u2 = 0; u2 < TU / 2; ++u2) vf[u2] = vn[u2];
                    }
                } else {
                // TU node tiles per softmax update: independent S chains on the matrix cores, one running-max rescale per 16 TU
                // nodes; with 32 trajectories per workgroup two tiles (four spill registers)
                uint4 kf[2 * TU], kn[2 * TU];                // [tile u4][form]
                float4 vf[TU], vn[TU];
                auto load4 = [&](int nt0, uint4 (&kk)[2 * TU], float4 (&vv)[TU]) {
#pragma unroll
                    for (int u4 = 0; u4 < TU; ++u4) {
                        kk[2 * u4] = gK[(nt0 + u4) * 128];
                        kk[2 * u4 + 1] = gK[(nt0 + u4) * 128 + 64];
                        vv[u4] = gV[(nt0 + u4) * 64];
                    }
                };
                load4(0, kf, vf);
#pragma unroll 1
                for (int nt = 0; nt < NTn; nt += TU) {
                    load4(min(nt + TU, NT - TU), kn, vn);
                    __builtin_amdgcn_sched_barrier(0);            // the loads stay here: in flight under this iteration's MFMAs
                    f32x4c S[NG][TU];
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) {              // accumulator input = the additive mask of the tile's nodes
                            const float4 m4 = *reinterpret_cast<const float4*>(sSc + (16 * g + lo) * SP + 16 * (nt + u4) + 4 * hi);
                            S[g][u4] = f32x4c{m4.x, m4.y, m4.z, m4.w};
                        }
                    // S^T += K_h q^T: [k1 | k2] [q1 | q1] + [k1 | k2] [q2 | q2] + [k1 | k3] [q3 | q1]
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4)
                            S[g][u4] = mfma_bf(u32x4{kf[2 * u4].x, kf[2 * u4].y, kf[2 * u4].z, kf[2 * u4].w}, q11[g], S[g][u4]);
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4)
                            S[g][u4] = mfma_bf(u32x4{kf[2 * u4].x, kf[2 * u4].y, kf[2 * u4].z, kf[2 * u4].w}, q22[g], S[g][u4]);
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4)
                            S[g][u4] = mfma_bf(u32x4{kf[2 * u4 + 1].x, kf[2 * u4 + 1].y, kf[2 * u4 + 1].z, kf[2 * u4 + 1].w}, q31[g], S[g][u4]);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        float tm = ELG_NEG_INF;
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4)
#pragma unroll
                            for (int i = 0; i < 4; ++i) tm = fmaxf(tm, S[g][u4][i]);
                        tm = quarters_max(tm);
                        const float mnew = fmaxf(mrun[g], tm);
                        const float sc = __builtin_amdgcn_exp2f((mrun[g] - mnew) * cs);
                        mrun[g] = mnew;
                        lrun[g] *= sc;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { o[g][i] *= sc; o2[g][i] *= sc; }
                        const float cm = -mnew * cs;
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                S[g][u4][i] = __builtin_amdgcn_exp2f(fmaf(S[g][u4][i], cs, cm));
                                lrun[g] += S[g][u4][i];
                            }
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) {             // O^T += V_h^T P^T, two accumulator chains
                            f32x4c& acc = (u4 & 1) ? o2[g] : o[g];
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u4].x, S[g][u4][0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u4].y, S[g][u4][1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u4].z, S[g][u4][2], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u4].w, S[g][u4][3], acc, 0, 0, 0);
                        }
#pragma unroll
                    for (int u4 = 0; u4 < TU; ++u4) { kf[2 * u4] = kn[2 * u4]; kf[2 * u4 + 1] = kn[2 * u4 + 1]; vf[u4] = vn[u4]; }
                }
                }                                                   // !BF
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float l = quarters_sum(lrun[g]);
                    const float inv = l > 0.f ? 1.0f / l : 0.f;
                    const float4 ov = make_float4((o[g][0] + o2[g][0]) * inv, (o[g][1] + o2[g][1]) * inv, (o[g][2] + o2[g][2]) * inv,
                                                  (o[g][3] + o2[g][3]) * inv);
                    {   // o as bf16 terms: the B operand of the pointer product (channels 16 wave + 4 hi .. + 3 of trajectory 16 g + lo)
                        unsigned pt[6];
                        bf_terms<3>(ov.x, ov.y, ov.z, ov.w, pt);
#pragma unroll
                        for (int tm = 0; tm < 3; ++tm)
                            *reinterpret_cast<uint2*>(sOb + (tm * NTR + 16 * g + lo) * OBP + 8 * wave + 2 * hi) = make_uint2(pt[2 * tm], pt[2 * tm + 1]);
                    }
                    if (TRAIN && l > 0.f && m_base + 16 * g + lo < A.M) {       // a decoding trajectory: a_h[n] = exp2(s cs - lse)
                        const size_t rrow = (size_t)b * Rcap + (size_t)t * A.M + m_base + 16 * g + lo;
                        *reinterpret_cast<float4*>(A.trO + rrow * ELG_E + 16 * wave + 4 * hi) = ov;
                        if (hi == 0) A.trLse[rrow * ELG_H + wave] = __log2f(l) + mrun[g] * cs;
                    }
                }
            }
            ELG_STAMP(sc_, 2);
            __syncthreads();
            ELG_STAMP(sc_, 3);
            // ================= pointer: node tiles over the waves; local-policy tail of group g: wave 7 - g =================
            // (the tail costs about what one node tile does: its wave joins the tile round-robin one round late)
            const int nloc = A.has_local ? NG : 0, W0 = 8 - nloc, skip = nloc ? W0 : 0;
            if (wave >= W0) co_local_tail_call(sT, sX + (7 - wave) * 16 * CO_XP, sdem + ((N1 + 3) & ~3) + (7 - wave) * 512, lo, hi);
            {
                // s^T[node][trajectory] = sum over the channels of PK[node][c] o[trajectory][c] on v_mfma_f32_16x16x32_bf16:
                // per 32 channels the six term products a1 b1, a1 b2, a2 b1, a2 b2, a1 b3, a3 b1 (24 instructions of 16 cycles per
                // node tile and group where the f32 form took 32 of 32 cycles; this phase kept the matrix pipe ~70 % busy)
                int nt = (wave < W0 ? wave : skip + wave);
                if constexpr (BF) {
                    // ---- bf16 mode: one term per operand: four instructions per node tile and group
                    uint4 pk[4], pkn[4];
                    auto loadpb = [&](int nt_, uint4 (&d)[4]) {
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) d[kb] = gPb[(nt_ * 4 + kb) * 64];
                    };
                    if (nt < NTn) loadpb(nt, pk);
#pragma unroll 1
                    while (nt < NTn) {
                        const int nxt = nt + (nt < skip ? W0 : 8);
                        loadpb(min(nxt, NTn - 1), pkn);
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4c a0[NG], a1[NG];
#pragma unroll
                        for (int g = 0; g < NG; ++g) { a0[g] = f32x4c{0.f, 0.f, 0.f, 0.f}; a1[g] = f32x4c{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            const u32x4 A1 = {pk[kb].x, pk[kb].y, pk[kb].z, pk[kb].w};
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                const uint4 q1 = *reinterpret_cast<const uint4*>(sOb + (16 * g + lo) * OBP + 16 * kb + 4 * hi);
                                const u32x4 B1 = {q1.x, q1.y, q1.z, q1.w};
                                if (kb & 1) a1[g] = mfma_bf(A1, B1, a1[g]);
                                else a0[g] = mfma_bf(A1, B1, a0[g]);
                            }
                        }
                        const int nb = 16 * nt + 4 * hi;
                        const float p0 = I.pb[min(nb, N1 - 1)], p1 = I.pb[min(nb + 1, N1 - 1)], p2 = I.pb[min(nb + 2, N1 - 1)], p3 = I.pb[min(nb + 3, N1 - 1)];
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            *reinterpret_cast<float4*>(sSc + (16 * g + lo) * SP + nb) =
                                make_float4(a0[g][0] + a1[g][0] + p0, a0[g][1] + a1[g][1] + p1, a0[g][2] + a1[g][2] + p2, a0[g][3] + a1[g][3] + p3);
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) pk[kb] = pkn[kb];
                        nt = nxt;
                    }
                } else {
                uint4 pk[12], pkn[12];
                auto loadpk = [&](int nt, uint4 (&d)[12]) {
#pragma unroll
                    for (int s4 = 0; s4 < 12; ++s4) d[s4] = gPK[(nt * 12 + s4) * 64];
                };
                if (nt < NTn) loadpk(nt, pk);
#pragma unroll 1
                while (nt < NTn) {
                    const int nxt = nt + (nt < skip ? W0 : 8);
                    loadpk(min(nxt, NTn - 1), pkn);              // the wave's next tile, in flight under this tile's MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4c a0[NG], a1[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) { a0[g] = f32x4c{0.f, 0.f, 0.f, 0.f}; a1[g] = f32x4c{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        const u32x4 A1 = {pk[3 * kb].x, pk[3 * kb].y, pk[3 * kb].z, pk[3 * kb].w};
                        const u32x4 A2 = {pk[3 * kb + 1].x, pk[3 * kb + 1].y, pk[3 * kb + 1].z, pk[3 * kb + 1].w};
                        const u32x4 A3 = {pk[3 * kb + 2].x, pk[3 * kb + 2].y, pk[3 * kb + 2].z, pk[3 * kb + 2].w};
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            const unsigned* ob = sOb + (16 * g + lo) * OBP + 16 * kb + 4 * hi;      // channels 32 kb + 8 hi .. + 7
                            const uint4 q1 = *reinterpret_cast<const uint4*>(ob);
                            const uint4 q2 = *reinterpret_cast<const uint4*>(ob + NTR * OBP);
                            const uint4 q3 = *reinterpret_cast<const uint4*>(ob + 2 * NTR * OBP);
                            const u32x4 B1 = {q1.x, q1.y, q1.z, q1.w}, B2 = {q2.x, q2.y, q2.z, q2.w}, B3 = {q3.x, q3.y, q3.z, q3.w};
                            a0[g] = mfma_bf(A1, B1, a0[g]);
                            a1[g] = mfma_bf(A1, B2, a1[g]);
                            a0[g] = mfma_bf(A2, B1, a0[g]);
                            a1[g] = mfma_bf(A2, B2, a1[g]);
                            a0[g] = mfma_bf(A1, B3, a0[g]);
                            a1[g] = mfma_bf(A3, B1, a1[g]);
                        }
                    }
                    const int nb = 16 * nt + 4 * hi;
                    const float p0 = I.pb[min(nb, N1 - 1)], p1 = I.pb[min(nb + 1, N1 - 1)], p2 = I.pb[min(nb + 2, N1 - 1)], p3 = I.pb[min(nb + 3, N1 - 1)];
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        *reinterpret_cast<float4*>(sSc + (16 * g + lo) * SP + nb) =
                            make_float4(a0[g][0] + a1[g][0] + p0, a0[g][1] + a1[g][1] + p1, a0[g][2] + a1[g][2] + p2, a0[g][3] + a1[g][3] + p3);
#pragma unroll
                    for (int s4 = 0; s4 < 12; ++s4) pk[s4] = pkn[s4];
                    nt = nxt;
                }
                }                                                   // !BF
            }
            ELG_STAMP(sc_, 4);
            __syncthreads();
            ELG_STAMP(sc_, 5);
            // ================= owners: clip, mask, softmax, choice =================
            {
                // (round 4) ONE read of the score row, 16 bytes per lane (lane l: nodes 256 c + 4 l .. + 3): the clipped logits
                // x = clip tanh(s + xi | s + slot term) of a trajectory stay in registers, from which the maximum, the first arg max, the
                // normaliser and (training) the softmax x clip Jacobian row are taken.  The slot lanes put s + (penalty + local score)
                // into the row themselves and flag the node in the slot words, so every value is the sum finish_step forms; the
                // first maximum in node order wins (models.py:405-420).  Written in STAGES over the wave's trajectories.
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const float dflt = A.has_penalty ? A.xi : 0.f;
                // a greedy construction whose caller does not ask for the chosen probabilities (elg_rollout_args.probs == NULL: the
                // reference's greedy rollout returns none, CVRPModel.py:70-73 / utils.py:24-25) needs the arg max only: no normaliser
                const bool want_p = A.probs != nullptr || A.full_probs != nullptr || A.mode != ELG_MODE_GREEDY;
                // ---- stage 1: slot terms into the rows
#pragma unroll
                for (int j = 0; j < NOWN; ++j) {
                    if (!dec[j]) continue;
                    const int q = wave + 8 * j;
                    float* scr = sSc + q * SP;
                    if (A.has_local && ln < ELG_SLOT_STRIDE) addv[j] += sX[q * CO_XP + CO_XU + ln] * A.inv_ens;
                    if (snid[j] >= 0) {
                        scr[snid[j]] += addv[j];
                        atomicOr(sSlotW + q * NCH + (snid[j] >> 6), 1ull << (snid[j] & 63));
                    }
                }
                wave_lds_fence();
                ELG_STAMP(sc_, 11);
                // ---- stage 2: the pass.  x stays in registers (NCH values per lane and trajectory); masks and the slot flag enter
                // as bit fields turned into 0 / -1 words (v_bfe_i32) -- no compare, no branch: a select on the closed bit invites the
                // compiler to branch around the tanh per element
                float xv[NOWN][NCH], mrun_l[NOWN];
                float thv[NOWN][TRAIN ? NCH : 1];                               // tanh of the node's score: the training rows' clip Jacobian
                float clipv = A.clip;
                asm volatile("" : "+v"(clipv));                                 // (a vector register: the scalar one is spilled)
#pragma unroll
                for (int j = 0; j < NOWN; ++j) {
                    mrun_l[j] = ELG_NEG_INF;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) xv[j][c] = ELG_NEG_INF;
                    if (!dec[j]) continue;
                    const int q = wave + 8 * j;
                    const float* scr = sSc + q * SP;
                    const unsigned long long* mkw = sMaskW + q * NCH;
                    const unsigned long long* slw = sSlotW + q * NCH;
#pragma unroll
                    for (int c = 0; c < NCH / 4; ++c) {
                        const int n0 = 256 * c + 4 * ln;
                        const float4 sv = *reinterpret_cast<const float4*>(scr + n0);
                        const unsigned mb = (unsigned)(mkw[n0 >> 6] >> (n0 & 63)), sbt = (unsigned)(slw[n0 >> 6] >> (n0 & 63));
                        const float svv[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int closed = __builtin_amdgcn_sbfe((int)mb, i, 1), slot = __builtin_amdgcn_sbfe((int)sbt, i, 1);     // 0 / -1
                            const float v = svv[i] + i2f(~slot & f2i(dflt));                  // s + xi, or the slot lanes' s + slot term
                            const float th = fast_tanh(v);
                            const float x = fmaf(clipv, th, i2f(closed & (int)0xff800000u));             // + (-inf) on a closed node
                            if (TRAIN) thv[j][4 * c + i] = th;
                            xv[j][4 * c + i] = x;
                            mrun_l[j] = fmaxf(mrun_l[j], x);
                        }
                    }
                }
                ELG_STAMP(sc_, 12);
                // ---- stage 3: merge the lanes: the maximum, the first node that attains it, the normaliser
                float gmx[NOWN], tot[NOWN];
                int bn[NOWN];
#pragma unroll
                for (int j = 0; j < NOWN; ++j) gmx[j] = wave_max(mrun_l[j]);
#pragma unroll
                for (int j = 0; j < NOWN; ++j) {
                    float part = 0.f;
                    int first = 0x7fffffff;
#pragma unroll
                    for (int c = NCH / 4 - 1; c >= 0; --c)
#pragma unroll
                        for (int i = 3; i >= 0; --i) {
                            first = (xv[j][4 * c + i] == gmx[j]) ? 256 * c + 4 * ln + i : first;
                            if (want_p) part += __expf(xv[j][4 * c + i] - gmx[j]);
                        }
                    // (node indices are exact in f32; a trajectory that does not decode has gmx = -inf = every x: its choice is unused)
                    const float firstn = -wave_max(first != 0x7fffffff ? -(float)first : -3.0e38f);
                    bn[j] = firstn < 1.0e9f ? (int)firstn : 0;
                    tot[j] = want_p ? wave_sum(part) : 1.0f;
                }
                ELG_STAMP(sc_, 13);
                // ---- stage 4: the choice and its probability
#pragma unroll
                for (int j = 0; j < NOWN; ++j) {
                    if (!dec[j]) continue;
                    const int q = wave + 8 * j;
                    const float* scr = sSc + q * SP;
                    const unsigned long long* mkw = sMaskW + q * NCH;
                    const unsigned long long* slw = sSlotW + q * NCH;
                    const float inv = 1.0f / tot[j];
                    float* frow = (A.full_probs && t < A.dump_T) ? A.full_probs + (bm[j] * A.dump_T + t) * N1 : nullptr;
                    if (frow)                                                   // tests: probabilities / clipped logits / scores before the clip
                        for (int n = ln; n < N1; n += 64) {
                            const bool masked = (mkw[n >> 6] >> (n & 63)) & 1ull, slot = (slw[n >> 6] >> (n & 63)) & 1ull;
                            const float sv = scr[n] + (slot ? 0.f : dflt);
                            const float x = masked ? ELG_NEG_INF : A.clip * fast_tanh(sv);
                            frow[n] = A.dump_logits == 2 ? (masked ? ELG_NEG_INF : sv) : A.dump_logits == 1 ? x : (masked ? 0.f : __expf(x - gmx[j]) * inv);
                        }
                    int s_ = 0;
                    if (A.mode == ELG_MODE_FORCED) s_ = (A.forced && t < A.Tforced) ? A.forced[bm[j] * A.Tforced + t] : 0;
                    else if (A.mode == ELG_MODE_GREEDY) s_ = bn[j];
                    else {
                        // inverse CDF in node order (second pass, lane = node of a 64-node chunk so that the scan runs in node order)
                        const float uni = A.uniforms ? A.uniforms[bm[j] * A.Tmax + t] : philox_uniform(A.seed, (unsigned)bm[j], (unsigned)t);
                        const float target = uni * tot[j];
                        float run = 0.f;
                        int found = -1, lastpos = 0;
#pragma unroll 1
                        for (int ch = 0; ch < NCH && found < 0; ++ch) {
                            const int n = ln + 64 * ch;
                            float e = 0.f;
                            if (!((mkw[ch] >> ln) & 1ull)) e = __expf(A.clip * fast_tanh(scr[n] + (((slw[ch] >> ln) & 1ull) ? 0.f : dflt)) - gmx[j]);
                            const float cs_ = wave_scan_incl(e, ln) + run;
                            run = readlane(cs_, 63);
                            const unsigned long long pos = __ballot(e > 0.f);
                            const unsigned long long hit = __ballot(e > 0.f && cs_ > target);
                            if (hit) found = 64 * ch + (int)__builtin_ctzll(hit);
                            if (pos) lastpos = 64 * ch + 63 - (int)__builtin_clzll(pos);
                        }
                        s_ = found >= 0 ? found : lastpos;
                    }
                    s_ = __builtin_amdgcn_readfirstlane(s_);
                    if (want_p) {
                        if (A.mode == ELG_MODE_GREEDY) pr[j] = inv;         // the arg max: x = gmx, exp(0) = 1 (an open node always exists)
                        else {
                            const bool smasked = (mkw[s_ >> 6] >> (s_ & 63)) & 1ull, sslot = (slw[s_ >> 6] >> (s_ & 63)) & 1ull;
                            const float xs = smasked ? ELG_NEG_INF : A.clip * fast_tanh(scr[s_] + (sslot ? 0.f : dflt));
                            const float p_ = (xs > ELG_NEG_INF) ? __expf(xs - gmx[j]) * inv : 0.f;
                            pr[j] = i2f(__builtin_amdgcn_readfirstlane(f2i(p_)));
                        }
                    }
                    sel[j] = s_;
                    if (TRAIN) {
                        // p[n] clip (1 - tanh^2): everything the backward needs of the clip / softmax Jacobian (row r = t M + m), and
                        // the clip Jacobian at the chosen node
                        const size_t rrow = (size_t)b * Rcap + (size_t)t * A.M + m_base + q;
                        float* rPC = A.trPC + rrow * N1;
#pragma unroll
                        for (int c = 0; c < NCH / 4; ++c)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int n = 256 * c + 4 * ln + i;
                                const float th = thv[j][4 * c + i];
                                if (n < N1) rPC[n] = __expf(xv[j][4 * c + i] - gmx[j]) * inv * (A.clip * (1.f - th * th));
                            }
                        const bool sslot = (slw[s_ >> 6] >> (s_ & 63)) & 1ull;
                        const float ths = fast_tanh(scr[s_] + (sslot ? 0.f : dflt));
                        if (ln == 0) A.trCsel[rrow] = A.clip * (1.f - ths * ths);
                    }
                }
                wave_lds_fence();
                ELG_STAMP(sc_, 14);
                // ---- stage 5: the slot words are all-zero between steps
#pragma unroll
                for (int j = 0; j < NOWN; ++j)
                    if (dec[j] && snid[j] >= 0) atomicAnd(sSlotW + (wave + 8 * j) * NCH + (snid[j] >> 6), ~(1ull << (snid[j] & 63)));
            }
            ELG_STAMP(sc_, 6);
        }
        // environment transition (in stages over the wave's trajectories: the coordinates of the chosen nodes come from L2)
        float2 sxy[NOWN];
#pragma unroll
        for (int j = 0; j < NOWN; ++j) {
            sxy[j] = make_float2(0.f, 0.f);
            if (!act[j]) continue;
            if (first_move) {
                if (A.mode == ELG_MODE_FORCED) sel[j] = (A.forced && t < A.Tforced) ? __builtin_amdgcn_readfirstlane(A.forced[bm[j] * A.Tforced + t]) : 0;
                else sel[j] = (!TSP && t == 0) ? 0 : __builtin_amdgcn_readfirstlane(A.starts[m_base + wave + 8 * j]);
            }
            sxy[j] = *reinterpret_cast<const float2*>(I.xy + 2 * sel[j]);
        }
#pragma unroll
        for (int j = 0; j < NOWN; ++j) {
            if (!act[j]) continue;
            if (lane == 0) {
                if (A.actions) A.actions[bm[j] * A.Tmax + t] = sel[j];
                if (A.probs) A.probs[((size_t)b * A.Tmax + t) * A.M + m_base + wave + 8 * j] = pr[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NOWN; ++j)
            if (act[j]) mt_env_update<NCH, TSP>(st[j], sVis + (wave + 8 * j) * NCH, I, N1, sel[j], lane, sxy[j]);
        ELG_STAMP(sc_, 7);
    }
#ifdef ELG_STAMPS
    if (A.full_probs == nullptr && A.uniforms && lane == 0)        // diagnostic build only: sums leave through the (unused) uniforms pointer
        for (int i = 0; i < 16; ++i) const_cast<float*>(A.uniforms)[((size_t)blockIdx.x * 8 + wave) * 16 + i] = sc_.acc[i];
#endif
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NOWN; ++j)
            if (has[j]) { if (A.reward) A.reward[bm[j]] = -st[j].len; if (A.tlen) A.tlen[bm[j]] = st[j].cnt; }
    }
}

template <int NCH, bool TSP, int NG, bool TRAIN, bool BF = false>
static int launch_fwd_mt_g(const elg_rollout_args& A, hipStream_t stream) {
    constexpr int NTR = 16 * NG;
    const size_t lds = ((size_t)NTR * 3 * 68 + (size_t)NTR * (64 * NCH + 4) + (size_t)NTR * NCH * 6 + (A.has_local ? NTR * CO_XP + CL_SIZE : 0) +
                        ((A.N1 + 3) & ~3) + (size_t)8 * ELG_SB_MIN) * 4;
    auto kern = rollout_fwd_mt_kernel<NCH, TSP, NG, TRAIN, BF>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, "mt rollout: hipFuncSetAttribute failed");
    if (!A.scratch) return fail(ELG_EINVAL, "rollout: 128 < N1 <= 1024 needs the scratch workspace (elg_rollout_scratch_floats)");
    elg_rollout_args B2 = A;
    B2.tiles = (A.M + NTR - 1) / NTR;                      // this kernel's geometry: 16 NG trajectories per workgroup
    (void)hipGetLastError();
    constexpr int NP = 64 * NCH;
    if (BF) {
        constexpr int NTt = NP / 16;
        hipLaunchKernelGGL(mt_repack_bf16_kernel, dim3(((8 * NTt + 4 * NTt + 4 * NTt) * 64 + 255) / 256, A.B), dim3(256), 0, stream, A.Kmat, A.Vmat,
                           A.PK, reinterpret_cast<unsigned*>(A.scratch), A.N1, NP);
    } else
        hipLaunchKernelGGL(mt_repack_kernel, dim3((2 * NP * 32 + NP * 16 + 255) / 256, A.B), dim3(256), 0, stream, A.Kmat, A.Vmat, A.PK, A.scratch, A.N1, NP);
    note_kernel(ELG_KERNEL_STREAM);
    hipLaunchKernelGGL(kern, dim3(B2.B * B2.tiles), dim3(512), lds, stream, B2);
    return launch_status("rollout_fwd_mt");
}

// 32 trajectories per workgroup (every K / V / PK fragment read from L2 serves two MFMA column groups) once 16 per workgroup
// would put more than one workgroup on a CU anyway; the score rows of 32 trajectories fit LDS up to 512 nodes
template <int NCH, bool TSP>
static int launch_fwd_mt(const elg_rollout_args& A, hipStream_t stream) {
    const bool two = NCH <= 8 && (long long)A.B * ((A.M + 15) / 16) > 256;
    if (A.trMask) {     // training forward: saves the backward rows
        if (!A.trPC || !A.trCsel || !A.trQ || !A.trO || !A.trLse) return fail(ELG_EINVAL, "rollout: incomplete training rows (128 < N1 <= 1024 needs trMask, trLse, trPC, trCsel, trQ, trO)");
        if (two) return launch_fwd_mt_g<NCH, TSP, (NCH <= 8 ? 2 : 1), true>(A, stream);
        return launch_fwd_mt_g<NCH, TSP, 1, true>(A, stream);
    }
    if (A.precision == 1) {
        if (two) return launch_fwd_mt_g<NCH, TSP, (NCH <= 8 ? 2 : 1), false, true>(A, stream);
        return launch_fwd_mt_g<NCH, TSP, 1, false, true>(A, stream);
    }
    if (two) return launch_fwd_mt_g<NCH, TSP, (NCH <= 8 ? 2 : 1), false>(A, stream);
    return launch_fwd_mt_g<NCH, TSP, 1, false>(A, stream);
}

// =============================================================================================
// rollout_fwd_xl_kernel: instances beyond the register-resident node layouts (N1 > 1024: Vrp-Set-XXL, N1 up to 7001).
// Same structure as the node-tiled kernel -- 8 lockstep trajectories per workgroup share the K / V / PK tiles staged in
// LDS -- but everything that is an NCH-sized register array there is a runtime loop here: the visited set and the mask
// words live in per-wave LDS, the pointer scores of a trajectory in a global scratch row (N1 floats, L2-resident; read
// back with device-scope loads), clip / softmax / choice stream over that row.  Inference only (greedy, sample, forced).
// =============================================================================================
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool TSP>
__global__ __launch_bounds__(512) void rollout_fwd_xl_kernel(const elg_rollout_args A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int WAVES = 8, NT = WAVES * 64, TR = 64, NGT = TR / 8;
    constexpr int S0 = TSP ? 0 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N1 = A.N1, NW = (N1 + 63) >> 6;
    const int G = gridDim.x;
    int u = blockIdx.x;
    if ((G & 7) == 0) u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int b = u / A.tiles, tile = u % A.tiles;
    const int tile_m = (A.M + A.tiles - 1) / A.tiles;
    const int m_lo = tile * tile_m, m_hi = min(A.M, m_lo + tile_m);
    const int ntiles = (N1 + TR - 1) / TR;
    // LDS: tile A | tile B | demand | per wave { visited words | mask words | slot scratch (144) | o row (128) }
    float* sTA = lds;
    float* sTB = lds + TR * ELG_E;
    float* p = lds + 2 * TR * ELG_E;
    float* sdem = p; p += (N1 + 3) & ~3;
    unsigned long long* svis = reinterpret_cast<unsigned long long*>(p) + (size_t)wave * 2 * NW;
    unsigned long long* smk = svis + NW;
    p += (size_t)WAVES * 4 * NW;
    const int ss = slot_stride_of(A.K), sbf = 3 * ss;              // slot scratch: 144 floats up to local_size 47, 192 above
    float* sb = p + wave * (sbf + ELG_E);
    float* so = sb + sbf;
    if (!TSP)
        for (int i = tid; i < N1; i += NT) sdem[i] = A.demand[(size_t)b * N1 + i];
    __syncthreads();
    const size_t NE = (size_t)N1 * ELG_E;
    const float4* gK = reinterpret_cast<const float4*>(A.Kmat + b * NE);
    const float4* gV = reinterpret_cast<const float4*>(A.Vmat + b * NE);
    const float4* gPK = reinterpret_cast<const float4*>(A.PK + b * NE);
    const float* pbv = A.pb + (size_t)b * N1;
    const float* Q1 = A.Q1 + b * NE;
    const float* Q2 = TSP ? A.Q2 + b * NE : nullptr;
    const float* xy = A.xy + (size_t)b * N1 * 2;
    const int* nidx = A.nbr_idx + (size_t)b * N1 * N1;
    const float* ndist = A.nbr_dist + (size_t)b * N1 * N1;
    const float* ntheta = A.nbr_theta + (size_t)b * N1 * N1;
    const int step_cap = TSP ? N1 : 2 * N1 + 2;
    const float dflt = A.has_penalty ? A.xi : 0.f;

    for (int m_base = m_lo; m_base < m_hi; m_base += WAVES) {
        const int m = m_base + wave;
        const bool has = m < m_hi;
        const size_t bm = (size_t)b * A.M + (has ? m : m_lo);
        float* scr = A.scratch + bm * (size_t)N1;
        int cur = 0, first = 0, cnt = 0, fin = has ? 0 : 1, nvis = 0;     // nvis: set bits of the visited set
        float load = 1.0f, len = 0.f, cx = 0.f, cy = 0.f;
        for (int i = lane; i < NW; i += 64) svis[i] = 0ull;
        wave_lds_fence();
        for (int step = 0; step < step_cap; ++step) {
            const bool active = has && !fin && cnt < A.Tmax;
            if (!__syncthreads_or(active ? 1 : 0)) break;
            const int t = cnt;
            const bool first_move = (!TSP && t <= 1) || (TSP && t == 0);
            const bool dec = active && !first_move;
            int fsel = 0;
            if (active && A.forced && t < A.Tforced) fsel = __builtin_amdgcn_readfirstlane(A.forced[bm * A.Tforced + t]);
            int sel = 0;
            float pr = 1.0f;
            if (__syncthreads_or(dec ? 1 : 0)) {
                float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
                float addval = 0.f;
                int snid = -1;
                if (dec) {
                    // ---- feasibility mask words (CVRPEnv.py:214-232 / TSPEnv.py:120)
                    const float lim = __fadd_rn(load, 1e-6f);
                    for (int ch = 0; ch < NW; ++ch) {
                        const int n = lane + 64 * ch;
                        bool mm = true;
                        if (n < N1) {
                            mm = (svis[ch] >> lane) & 1ull;
                            if (!TSP) {
                                mm = mm || (lim < sdem[n]);
                                if (n == 0 && fin) mm = false;
                            }
                        }
                        const unsigned long long bal = __ballot(mm);
                        if (lane == 0) smk[ch] = bal;
                    }
                    wave_lds_fence();
                    // ---- query row
                    const int cb = (lane & 31) * 4;
                    q4 = *reinterpret_cast<const float4*>(Q1 + (size_t)cur * ELG_E + cb);
                    if (TSP) {
                        const float4 qf = *reinterpret_cast<const float4*>(Q2 + (size_t)first * ELG_E + cb);
                        q4.x += qf.x; q4.y += qf.y; q4.z += qf.z; q4.w += qf.w;
                    } else {
                        const float4 w = *reinterpret_cast<const float4*>(A.wl + cb);
                        q4.x = fmaf(load, w.x, q4.x); q4.y = fmaf(load, w.y, q4.y);
                        q4.z = fmaf(load, w.z, q4.z); q4.w = fmaf(load, w.w, q4.w);
                    }
                    // ---- k-NN slots: the first K open customers of cur's sorted neighbour list (knn_slots / slot_setup)
                    if (A.has_penalty || A.has_local) {
                        int found = 0;
                        const size_t row = (size_t)cur * N1;
                        for (int ch = 0; ch < NW && found < A.K; ++ch) {
                            const int i = lane + 64 * ch;
                            const bool valid = i < N1;
                            const int nid = valid ? nidx[row + i] : 0;
                            const float nd = valid ? ndist[row + i] : 0.f;
                            const float nth = valid ? ntheta[row + i] : 0.f;
                            bool cand = valid && !((smk[nid >> 6] >> (nid & 63)) & 1ull);
                            if (!TSP) cand = cand && (nid != 0);
                            const unsigned long long bal = __ballot(cand);
                            const int rank = found + lanes_below(bal);
                            if (cand && rank < A.K) {
                                sb[S0 + rank] = nd;
                                sb[ss + S0 + rank] = nth;
                                sb[2 * ss + S0 + rank] = i2f(nid);
                            }
                            found += __popcll(bal);
                        }
                        const int k = found < A.K ? found : A.K;
                        wave_lds_fence();
                        const int j = lane;
                        const bool cust = (j >= S0) && (j < S0 + k);
                        float sd = 0.f, sth = 0.f;
                        if (cust) { sd = sb[j]; sth = sb[ss + j]; snid = f2i(sb[2 * ss + j]); }
                        const float dmax = (k > 0) ? sb[S0 + k - 1] : 0.f;
                        wave_lds_fence();
                        if (!TSP && j == 0) snid = 0;
                        float pen = 0.f;
                        if (A.has_penalty && cust) {
                            if (TSP) pen = -(sd / (dmax + 1e-6f));
                            else pen = (dmax != 0.f) ? -(sd / dmax) : -sd;
                        }
                        const float nf = dmax + 1e-6f;
                        float f0 = 0.f, f1 = 0.f, f2 = 0.f;
                        if (cust) {
                            f0 = sd / nf; f1 = sth;
                            if (A.euclidean) { f0 = __fsub_rn(xy[2 * snid], cx) / nf; f1 = __fsub_rn(xy[2 * snid + 1], cy) / nf; }
                            if (!TSP) f2 = sdem[snid] / load;
                        }
                        bool smask = !cust;
                        if (!TSP && j == 0) smask = smk[0] & 1ull;
                        float uu = 0.f;
                        if (A.has_local) uu = local_policy<TSP>(A.loc, lane, f0, f1, f2, smask, nullptr);
                        addval = pen + uu * A.inv_ens;
                    }
                }
                // ---- glimpse: online softmax over the K / V tiles
                float m_run = ELG_NEG_INF, l_run = 0.f;
                f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
                for (int tt = 0; tt < ntiles; ++tt) {
                    __syncthreads();
                    const int row0 = tt * TR;
                    for (int i = tid; i < TR * 32; i += NT) {
                        const int grow = row0 + (i >> 5);
                        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
                        if (grow < N1) { kv = gK[(size_t)grow * 32 + (i & 31)]; vv = gV[(size_t)grow * 32 + (i & 31)]; }
                        reinterpret_cast<float4*>(sTA)[i] = kv;
                        reinterpret_cast<float4*>(sTB)[i] = vv;
                    }
                    __syncthreads();
                    if (dec) glimpse_tile<NGT>(sTA, sTB, row0, N1, lane, q4, smk[tt], 0ull, m_run, l_run, acc01, acc23);
                }
                if (dec) {
                    float l = l_run;
                    l += quad_xor1(l); l += quad_xor2(l); l = x32_sum(l);
                    const float inv = 1.0f / l;
                    float4 o4 = make_float4(acc01.x, acc01.y, acc23.x, acc23.y);
                    o4.x = x32_sum(o4.x); o4.y = x32_sum(o4.y);
                    o4.z = x32_sum(o4.z); o4.w = x32_sum(o4.w);
                    o4.x *= inv; o4.y *= inv; o4.z *= inv; o4.w *= inv;
                    if (lane < 32) *reinterpret_cast<float4*>(so + 4 * lane) = o4;
                }
                // ---- pointer scores, PK tile by tile, into the trajectory's scratch row
                for (int tt = 0; tt < ntiles; ++tt) {
                    __syncthreads();
                    const int row0 = tt * TR;
                    for (int i = tid; i < TR * 32; i += NT) {
                        const int lrow = i >> 5, c4 = i & 31, grow = row0 + lrow;
                        float4 pk = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (grow < N1) pk = gPK[(size_t)grow * 32 + c4];
                        reinterpret_cast<float4*>(sTA)[lrow * 32 + (c4 ^ (lrow & 31))] = pk;
                    }
                    __syncthreads();
                    if (dec) {
                        f32x2 acc = {0.f, 0.f};
#pragma unroll 8
                        for (int c4 = 0; c4 < 32; ++c4) {
                            const float4 o = *reinterpret_cast<const float4*>(so + 4 * c4);
                            const float4 pk = *reinterpret_cast<const float4*>(sTA + (size_t)lane * ELG_E + 4 * (c4 ^ (lane & 31)));
                            acc = __builtin_elementwise_fma(lo2(o), lo2(pk), acc);
                            acc = __builtin_elementwise_fma(hi2(o), hi2(pk), acc);
                        }
                        const int n = row0 + lane;
                        if (n < N1) st_dev(scr + n, (acc.x + acc.y) + pbv[n]);
                    }
                }
                // ---- clip, mask, softmax, choice: streaming over the scratch row   (models.py:405-420)
                if (dec) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (snid >= 0) st_dev(scr + snid, ld_dev(scr + snid) + (addval - dflt));     // penalty + local terms, slot nodes
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                    float* frow = (A.full_probs && t < A.dump_T) ? A.full_probs + (bm * A.dump_T + t) * N1 : nullptr;
                    // pass 1: clipped logits back into the row, their maximum and (greedy) the first arg max
                    float mx = ELG_NEG_INF;
                    int bn = 0x7fffffff;
                    for (int ch = 0; ch < NW; ++ch) {
                        const int n = lane + 64 * ch;
                        float x = ELG_NEG_INF;
                        if (n < N1) {
                            const bool masked = (smk[ch] >> lane) & 1ull;
                            const float sv = ld_dev(scr + n) + dflt;
                            if (!masked) x = A.clip * fast_tanh(sv);
                            if (frow && A.dump_logits == 2) frow[n] = masked ? ELG_NEG_INF : sv;
                            st_dev(scr + n, x);
                        }
                        if (x > mx) { mx = x; bn = n; }
                    }
                    {   // wave argmax, ties -> lowest node: DPP all-reduce of the value, then of the (negated) index among its holders
                        const float gmx = wave_max(mx);
                        const float cand = (mx == gmx && bn != 0x7fffffff) ? -(float)bn : -3.0e38f;     // node indices < 2^24: exact in f32
                        const float first = -wave_max(cand);
                        bn = first < 1.0e9f ? (int)first : 0;
                        mx = gmx;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                    // pass 2: normaliser
                    float part = 0.f;
                    for (int ch = 0; ch < NW; ++ch) {
                        const int n = lane + 64 * ch;
                        if (n < N1) {
                            const float x = ld_dev(scr + n);
                            part += (x > ELG_NEG_INF) ? __expf(x - mx) : 0.f;
                        }
                    }
                    const float tot = wave_sum(part);
                    const float inv = 1.0f / tot;
                    if (frow && A.dump_logits != 2)
                        for (int n = lane; n < N1; n += 64) {
                            const float x = ld_dev(scr + n);
                            frow[n] = A.dump_logits == 1 ? x : ((x > ELG_NEG_INF) ? __expf(x - mx) * inv : 0.f);
                        }
                    if (A.mode == ELG_MODE_FORCED) sel = fsel;
                    else if (A.mode == ELG_MODE_GREEDY) sel = bn;
                    else {
                        // inverse CDF in node order
                        const float uni = A.uniforms ? A.uniforms[bm * A.Tmax + t] : philox_uniform(A.seed, (unsigned)bm, (unsigned)t);
                        const float target = uni * tot;
                        float run = 0.f;
                        int found = -1, lastpos = 0;
                        for (int ch = 0; ch < NW && found < 0; ++ch) {
                            const int n = lane + 64 * ch;
                            float e = 0.f;
                            if (n < N1) { const float x = ld_dev(scr + n); e = (x > ELG_NEG_INF) ? __expf(x - mx) : 0.f; }
                            const float cs = wave_scan_incl(e, lane) + run;
                            run = readlane(cs, 63);
                            const unsigned long long pos = __ballot(e > 0.f);
                            const unsigned long long hit = __ballot(e > 0.f && cs > target);
                            if (hit) found = 64 * ch + (int)__builtin_ctzll(hit);
                            if (pos) lastpos = 64 * ch + 63 - (int)__builtin_clzll(pos);
                        }
                        sel = found >= 0 ? found : lastpos;
                    }
                    sel = __builtin_amdgcn_readfirstlane(sel);
                    const float xs = ld_dev(scr + sel);
                    pr = (xs > ELG_NEG_INF) ? __expf(xs - mx) * inv : 0.f;
                    pr = i2f(__builtin_amdgcn_readfirstlane(f2i(pr)));
                }
            }
            if (active) {
                if (first_move) {
                    if (A.mode == ELG_MODE_FORCED) sel = fsel;
                    else sel = (!TSP && t == 0) ? 0 : __builtin_amdgcn_readfirstlane(A.starts[m]);
                }
                if (lane == 0) {
                    if (A.actions) A.actions[bm * A.Tmax + t] = sel;
                    if (A.probs) A.probs[((size_t)b * A.Tmax + t) * A.M + m] = pr;
                }
                // ---- environment transition (env_update with the visited set in LDS)
                const float sx = xy[2 * sel], sy = xy[2 * sel + 1];
                if (cnt > 0) len += dist2d(cx, cy, sx, sy);
                cx = sx; cy = sy;
                if (TSP) { if (cnt == 0) first = sel; }
                else load = (sel == 0) ? 1.0f : __fsub_rn(load, sdem[sel]);
                if (lane == 0) {
                    unsigned long long w = svis[sel >> 6];
                    const unsigned long long bit = 1ull << (sel & 63);
                    int d = (w & bit) ? 0 : 1;
                    w |= bit;
                    svis[sel >> 6] = w;
                    if (!TSP) {
                        // depot counts as visited exactly while the trajectory stands on it (CVRPEnv.py:214-216)
                        unsigned long long w0 = svis[0];
                        if (sel != 0 && (w0 & 1ull)) { w0 &= ~1ull; d -= 1; svis[0] = w0; }
                    }
                    sb[0] = i2f(d);
                }
                wave_lds_fence();
                nvis += f2i(sb[0]);
                wave_lds_fence();
                cur = sel;
                cnt += 1;
                if (TSP) {
                    if (cnt == N1) { len += dist2d(sx, sy, xy[2 * first], xy[2 * first + 1]); fin = 1; }
                } else if (nvis == N1) fin = 1;
            }
        }
        if (has && lane == 0) {
            if (A.reward) A.reward[bm] = -len;
            if (A.tlen) A.tlen[bm] = cnt;
        }
        __syncthreads();
    }
}

template <bool TSP>
static int launch_fwd_xl(const elg_rollout_args& A, hipStream_t stream) {
    if (!A.scratch) return fail(ELG_EINVAL, "rollout: N1 > 1024 needs the (B,M,N1) scratch rows");
    if (A.N1 > 8192) return fail(ELG_ENOTIMPL, "rollout: N1 > 8192 not built");
    const int NW = (A.N1 + 63) / 64;
    const size_t lds = ((size_t)2 * 64 * ELG_E + ((A.N1 + 3) & ~3) + (size_t)8 * 4 * NW + (size_t)8 * (3 * slot_stride_of(A.K) + ELG_E)) * 4;
    if (lds > 163840 - 256) return fail(ELG_EINVAL, "xl rollout: LDS budget exceeded");
    auto kern = rollout_fwd_xl_kernel<TSP>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, "xl rollout: hipFuncSetAttribute failed");
    (void)hipGetLastError();
    note_kernel(ELG_KERNEL_XL);
    hipLaunchKernelGGL(kern, dim3(A.B * A.tiles), dim3(512), lds, stream, A);
    return launch_status("rollout_fwd_xl");
}

// =============================================================================================
// rollout_fwd_xm_kernel (round 4): N1 > 1024 on the matrix cores.  The phase structure and the matrix phases of the streaming
// kernel above -- 16 lockstep trajectories per workgroup, wave = head in the glimpse, node tiles over the waves in the pointer
// phase, K / V / PK streamed in MFMA-fragment order (mt_repack_kernel / mt_repack_bf16_kernel with NP = 64 ceil(N1 / 64)) -- with
// the owners of rollout_fwd_xl_kernel: every node-indexed quantity is a runtime loop over 64-node chunks, the visited and mask
// words of a trajectory live in LDS, and its score row -- which also carries the ADDITIVE mask (0 / -inf) from the owners into the
// glimpse, as the accumulator input of the score product -- in a global scratch row of NP floats (L2; the waves of a workgroup
// share their CU's vector L1, so plain loads / stores ordered by the workgroup barriers suffice).  Inference only.
// scratch = [B][NP x (tables)] | [B x tiles x 16][NP] score rows  (elg_rollout_scratch_floats, variant 0, N1 > 1024).
// rollout_fwd_xl_kernel (variant 2) remains: the one-wavefront-per-trajectory reference of the tests at these sizes.
// =============================================================================================
struct XmTraj {
    int cur, first, cnt, fin, nvis;
    float load, len, cx, cy;
};

template <bool TSP, bool BF>
__global__ __launch_bounds__(512) void rollout_fwd_xm_kernel(const elg_rollout_args A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NTR = 16, NOWN = 2, QP = 132, OBP = 68, TU = 4;
    constexpr int S0 = TSP ? 0 : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int N1 = A.N1, NW = (N1 + 63) >> 6, NP = 64 * NW, NT = NP >> 4, NTn = (N1 + 15) >> 4;
    const int G = gridDim.x;
    int u = blockIdx.x;
    if ((G & 7) == 0) u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int b = u / A.tiles, tile = u % A.tiles;
    const int m_base = tile * NTR;
    // LDS: query rows / bf16 term planes of o | mask words | visited words | slot blocks | local tables | demand | slot scratch
    float* sQ = lds;
    unsigned* sOb = reinterpret_cast<unsigned*>(lds);
    unsigned long long* sMaskW = reinterpret_cast<unsigned long long*>(sQ + NTR * 3 * OBP);
    unsigned long long* sVis = sMaskW + NTR * NW;
    float* sX = reinterpret_cast<float*>(sVis + NTR * NW);
    float* sT = sX + (A.has_local ? NTR * CO_XP : 0);
    float* sdem = sT + (A.has_local ? CL_SIZE : 0);
    float* sb = sdem + ((N1 + 3) & ~3) + wave * ELG_SB_MIN;
    if (!TSP)
        for (int i = tid; i < N1; i += 512) sdem[i] = A.demand[(size_t)b * N1 + i];
    for (int i = tid; i < NTR * NW; i += 512) sVis[i] = 0ull;
    if (A.has_local) co_stage_local(A.loc, sT, tid, 512);
    __syncthreads();
    const size_t NE = (size_t)N1 * ELG_E;
    const float* pbv = A.pb + (size_t)b * N1;
    const float* Q1 = A.Q1 + b * NE;
    const float* Q2 = TSP ? A.Q2 + b * NE : nullptr;
    const float* xy = A.xy + (size_t)b * N1 * 2;
    const int* nidx = A.nbr_idx + (size_t)b * N1 * N1;
    const float* ndist = A.nbr_dist + (size_t)b * N1 * N1;
    const float* ntheta = A.nbr_theta + (size_t)b * N1 * N1;
    const float dflt = A.has_penalty ? A.xi : 0.f;
    constexpr int TW = BF ? MT_BF_WORDS : MT_KB_WORDS + ELG_E + MT_PKB_WORDS;
    const float* gF = A.scratch + (size_t)b * NP * TW;
    float* rows = A.scratch + (size_t)A.B * NP * (MT_KB_WORDS + ELG_E + MT_PKB_WORDS) + ((size_t)b * A.tiles + tile) * NTR * (size_t)NP;
    // f32-parity mode (split-bf16 scores / pointer, f32 weights x values): [h][tile][form][lane] uint4 | [h][tile][lane] float4 | [tile][kb][term][lane] uint4
    const uint4* gK = reinterpret_cast<const uint4*>(gF) + (size_t)wave * NT * 2 * 64 + lane;
    const float4* gV = reinterpret_cast<const float4*>(gF + (size_t)NP * MT_KB_WORDS) + (size_t)wave * NT * 64 + lane;
    const uint4* gPK = reinterpret_cast<const uint4*>(gF + (size_t)NP * (MT_KB_WORDS + ELG_E)) + lane;
    // bf16 mode: [h][tile][lane] uint2 | [h][pair][lane] uint4 | [tile][kb][lane] uint4
    const uint2* gKb = reinterpret_cast<const uint2*>(gF) + (size_t)wave * NT * 64 + lane;
    const uint4* gVb = reinterpret_cast<const uint4*>(gF + (size_t)NP * 64) + (size_t)wave * (NT / 2) * 64 + lane;
    const uint4* gPb = reinterpret_cast<const uint4*>(gF + (size_t)NP * 128) + lane;
    const int step_cap = TSP ? N1 : 2 * N1 + 2;

    XmTraj st[NOWN];
    bool has[NOWN];
    size_t bm[NOWN];
#pragma unroll
    for (int j = 0; j < NOWN; ++j) {
        has[j] = m_base + wave + 8 * j < A.M;
        bm[j] = (size_t)b * A.M + (has[j] ? m_base + wave + 8 * j : 0);
        st[j].cur = 0; st[j].first = 0; st[j].cnt = 0; st[j].fin = has[j] ? 0 : 1; st[j].nvis = 0;
        st[j].load = 1.0f; st[j].len = 0.f; st[j].cx = 0.f; st[j].cy = 0.f;
    }
    const int lane_k = lane;
    for (int t = 0; t < step_cap && t < A.Tmax; ++t) {
        // (opaque per step: what the owners' phases derive from the lane id is re-formed here instead of living in registers across
        // the matrix phases, where the f32-parity instantiation has none to spare)
        int lane = lane_k;
        asm volatile("" : "+v"(lane));
        const bool first_move = (!TSP && t <= 1) || (TSP && t == 0);
        bool act[NOWN], dec[NOWN];
        bool any = false;
#pragma unroll
        for (int j = 0; j < NOWN; ++j) {
            act[j] = has[j] && !st[j].fin;
            dec[j] = act[j] && !first_move;
            any = any || act[j];
        }
        if (!__syncthreads_or(any ? 1 : 0)) break;
        int sel[NOWN], snid[NOWN];
        float pr[NOWN], addv[NOWN];
#pragma unroll
        for (int j = 0; j < NOWN; ++j) { sel[j] = 0; snid[j] = -1; pr[j] = 1.0f; addv[j] = 0.f; }
        if (!first_move) {
            // ================= owners: mask words + additive mask row, query row, k-NN slots, slot features =================
            auto prepare = [&](const XmTraj& s1, bool dc, int q, float& addval, int& sn) {
                float* srow = rows + (size_t)q * NP;
                unsigned long long* mkw = sMaskW + q * NW;
                const unsigned long long* vis = sVis + q * NW;
                float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
                float f0 = 0.f, f1 = 0.f, f2 = 0.f;
                int scode = -1;
                if (dc) {
                    // the mask of the step as words (the streaming kernel's mt_mask_words with runtime loops, NW <= 128: lane c
                    // assembles the words c and c + 64 = visited | ballot(demand > load) | nodes past N1; CVRPEnv.py:214-232 /
                    // TSPEnv.py:120), then the additive row (0 open / -inf closed) 16 bytes per lane
                    int lc = lane;
                    asm volatile("" : "+v"(lc));
                    unsigned long long w0 = lc < NW ? vis[lc] : 0ull, w1 = lc + 64 < NW ? vis[lc + 64] : 0ull;
                    if (!TSP) {
                        const float lim = __fadd_rn(s1.load, 1e-6f);
                        unsigned l0 = 0u, h0 = 0u, l1 = 0u, h1 = 0u;
#pragma unroll 2
                        for (int ch = 0; ch < NW; ++ch) {               // (nodes past N1 read the scratch behind the demand row: closed below)
                            const unsigned long long bal = __ballot(lim < sdem[lc + 64 * ch]);
                            const bool mine = lc == (ch & 63);
                            if (ch < 64) { l0 = mine ? (unsigned)bal : l0; h0 = mine ? (unsigned)(bal >> 32) : h0; }
                            else { l1 = mine ? (unsigned)bal : l1; h1 = mine ? (unsigned)(bal >> 32) : h1; }
                        }
                        w0 |= ((unsigned long long)h0 << 32) | l0;
                        w1 |= ((unsigned long long)h1 << 32) | l1;
                    }
                    {
                        const int r0 = N1 - 64 * lc, r1 = r0 - 4096;    // nodes of the words lc / lc + 64 that exist
                        w0 |= r0 >= 64 ? 0ull : (r0 <= 0 ? ~0ull : ~((1ull << r0) - 1ull));
                        w1 |= r1 >= 64 ? 0ull : (r1 <= 0 ? ~0ull : ~((1ull << r1) - 1ull));
                    }
                    if (!TSP && lc == 0 && s1.fin) w0 &= ~1ull;
                    if (lc < NW) mkw[lc] = w0;
                    if (lc + 64 < NW) mkw[lc + 64] = w1;
                    wave_lds_fence();
                    for (int n0 = 4 * lc; n0 < NP; n0 += 256) {
                        const unsigned bits = (unsigned)(mkw[n0 >> 6] >> (n0 & 63));
                        *reinterpret_cast<float4*>(srow + n0) = make_float4((bits & 1u) ? ELG_NEG_INF : 0.f, (bits & 2u) ? ELG_NEG_INF : 0.f,
                                                                            (bits & 4u) ? ELG_NEG_INF : 0.f, (bits & 8u) ? ELG_NEG_INF : 0.f);
                    }
                    const int cb = (lane & 31) * 4;
                    q4 = *reinterpret_cast<const float4*>(Q1 + (size_t)s1.cur * ELG_E + cb);
                    if (TSP) {
                        const float4 qf = *reinterpret_cast<const float4*>(Q2 + (size_t)s1.first * ELG_E + cb);
                        q4.x += qf.x; q4.y += qf.y; q4.z += qf.z; q4.w += qf.w;
                    } else {
                        const float4 w = *reinterpret_cast<const float4*>(A.wl + cb);
                        q4.x = fmaf(s1.load, w.x, q4.x); q4.y = fmaf(s1.load, w.y, q4.y);
                        q4.z = fmaf(s1.load, w.z, q4.z); q4.w = fmaf(s1.load, w.w, q4.w);
                    }
                    if (A.has_penalty || A.has_local) {                 // knn_slots / slot_setup with runtime chunk loops
                        int found = 0;
                        const size_t row = (size_t)s1.cur * N1;
                        for (int ch = 0; ch < NW && found < A.K; ++ch) {
                            const int i = lane + 64 * ch;
                            const bool valid = i < N1;
                            const int nid = valid ? nidx[row + i] : 0;
                            const float nd = valid ? ndist[row + i] : 0.f;
                            const float nth = valid ? ntheta[row + i] : 0.f;
                            bool cand = valid && !((mkw[nid >> 6] >> (nid & 63)) & 1ull);
                            if (!TSP) cand = cand && (nid != 0);
                            const unsigned long long bal = __ballot(cand);
                            const int rank = found + lanes_below(bal);
                            if (cand && rank < A.K) {
                                sb[S0 + rank] = nd;
                                sb[ELG_SLOT_STRIDE + S0 + rank] = nth;
                                sb[2 * ELG_SLOT_STRIDE + S0 + rank] = i2f(nid);
                            }
                            found += __popcll(bal);
                        }
                        const int k = found < A.K ? found : A.K;
                        wave_lds_fence();
                        const int jl = lane;
                        const bool cust = (jl >= S0) && (jl < S0 + k);
                        float sd = 0.f, sth = 0.f;
                        if (cust) { sd = sb[jl]; sth = sb[ELG_SLOT_STRIDE + jl]; sn = f2i(sb[2 * ELG_SLOT_STRIDE + jl]); }
                        const float dmax = (k > 0) ? sb[S0 + k - 1] : 0.f;
                        wave_lds_fence();
                        if (!TSP && jl == 0) sn = 0;
                        float pen = 0.f;
                        if (A.has_penalty && cust) {
                            if (TSP) pen = -(sd / (dmax + 1e-6f));
                            else pen = (dmax != 0.f) ? -(sd / dmax) : -sd;
                        }
                        const float nf = dmax + 1e-6f;
                        if (cust) {
                            f0 = sd / nf; f1 = sth;
                            if (A.euclidean) { f0 = __fsub_rn(xy[2 * sn], s1.cx) / nf; f1 = __fsub_rn(xy[2 * sn + 1], s1.cy) / nf; }
                            if (!TSP) f2 = sdem[sn] / s1.load;
                        }
                        bool smask = !cust;
                        if (!TSP && jl == 0) smask = mkw[0] & 1ull;
                        scode = smask ? (sn >= 0 ? -2 : -1) : sn;
                        addval = pen;
                    }
                } else {
                    for (int n0 = 4 * lane; n0 < NP; n0 += 256)                             // not decoding: every node closed
                        *reinterpret_cast<float4*>(srow + n0) = make_float4(ELG_NEG_INF, ELG_NEG_INF, ELG_NEG_INF, ELG_NEG_INF);
                    for (int c = lane; c < NW; c += 64) mkw[c] = ~0ull;
                }
                if (A.has_local && lane < ELG_SLOT_STRIDE) {            // slot block of the local policy (layout of the cooperative kernel)
                    float* X = sX + q * CO_XP;
                    X[CO_XF + lane] = f0; X[CO_XF + ELG_SLOT_STRIDE + lane] = f1; X[CO_XF + 2 * ELG_SLOT_STRIDE + lane] = f2;
                    reinterpret_cast<int*>(X)[CO_XS + lane] = scode;
                }
                if (lane < 32) *reinterpret_cast<float4*>(sQ + q * QP + 4 * lane) = q4;
            };
#pragma unroll
            for (int j = 0; j < NOWN; ++j) prepare(st[j], dec[j], wave + 8 * j, addv[j], snid[j]);
            __syncthreads();
            // ================= glimpse: wave = head =================
            {
                const float cs = 0.25f * 1.4426950408889634f;
                const float* mrow = rows + (size_t)lo * NP + 4 * hi;          // additive mask of trajectory lo
                const float4 qv = *reinterpret_cast<const float4*>(sQ + lo * QP + 16 * wave + 4 * hi);
                unsigned pt[6];
                bf_terms<3>(qv.x, qv.y, qv.z, qv.w, pt);
                const u32x4 q11 = {pt[0], pt[1], pt[0], pt[1]}, q22 = {pt[2], pt[3], pt[2], pt[3]}, q31 = {pt[4], pt[5], pt[0], pt[1]};
                const u32x4 qb1 = {pt[0], pt[1], 0u, 0u};
                float mrun = -1e30f, lrun = 0.f;
                f32x4c o = {0.f, 0.f, 0.f, 0.f}, o2 = {0.f, 0.f, 0.f, 0.f};
                __syncthreads();        // every head has its queries: the rows are free for the term planes of o
                // local policy, stage 1: the four head units on the waves that finish the glimpse first (as in the streaming kernel)
                if (A.has_local && wave < 4) co_local_head_call(sT, sX, sdem + ((N1 + 3) & ~3), wave & 3, lo, hi);
                auto softmax_update = [&](f32x4c (&S)[TU]) {
                    float tm = ELG_NEG_INF;
#pragma unroll
                    for (int u4 = 0; u4 < TU; ++u4)
#pragma unroll
                        for (int i = 0; i < 4; ++i) tm = fmaxf(tm, S[u4][i]);
                    tm = quarters_max(tm);
                    const float mnew = fmaxf(mrun, tm);
                    const float sc = __builtin_amdgcn_exp2f((mrun - mnew) * cs);
                    mrun = mnew;
                    lrun *= sc;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { o[i] *= sc; o2[i] *= sc; }
                    const float cm = -mnew * cs;
#pragma unroll
                    for (int u4 = 0; u4 < TU; ++u4)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            S[u4][i] = __builtin_amdgcn_exp2f(fmaf(S[u4][i], cs, cm));
                            lrun += S[u4][i];
                        }
                };
                if constexpr (BF) {
                    uint2 kf[TU], kn[TU];
                    uint4 vf[TU / 2], vn[TU / 2];
                    auto loadb = [&](int nt0, uint2 (&kk)[TU], uint4 (&vv)[TU / 2]) {
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) kk[u4] = gKb[(size_t)(nt0 + u4) * 64];
#pragma unroll
                        for (int u2 = 0; u2 < TU / 2; ++u2) vv[u2] = gVb[(size_t)(nt0 / 2 + u2) * 64];
                    };
                    loadb(0, kf, vf);
#pragma unroll 1
                    for (int nt = 0; nt < NTn; nt += TU) {
                        loadb(min(nt + TU, NT - TU), kn, vn);
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4c S[TU];
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) {
                            const float4 m4 = *reinterpret_cast<const float4*>(mrow + 16 * (nt + u4));
                            S[u4] = mfma_bf(u32x4{kf[u4].x, kf[u4].y, 0u, 0u}, qb1, f32x4c{m4.x, m4.y, m4.z, m4.w});
                        }
                        softmax_update(S);
#pragma unroll
                        for (int u2 = 0; u2 < TU / 2; ++u2) {
                            const u32x4 pb_ = {pk_bf16(S[2 * u2][0], S[2 * u2][1]), pk_bf16(S[2 * u2][2], S[2 * u2][3]),
                                               pk_bf16(S[2 * u2 + 1][0], S[2 * u2 + 1][1]), pk_bf16(S[2 * u2 + 1][2], S[2 * u2 + 1][3])};
                            f32x4c& acc = (u2 & 1) ? o2 : o;
                            acc = mfma_bf(u32x4{vf[u2].x, vf[u2].y, vf[u2].z, vf[u2].w}, pb_, acc);
                        }
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) kf[u4] = kn[u4];
#pragma unroll
                        for (int u2 = 0; u2 < TU / 2; ++u2) vf[u2] = vn[u2];
                    }
                } else {
                    uint4 kf[2 * TU], kn[2 * TU];
                    float4 vf[TU], vn[TU];
                    auto load4 = [&](int nt0, uint4 (&kk)[2 * TU], float4 (&vv)[TU]) {
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) {
                            kk[2 * u4] = gK[(size_t)(nt0 + u4) * 128];
                            kk[2 * u4 + 1] = gK[(size_t)(nt0 + u4) * 128 + 64];
                            vv[u4] = gV[(size_t)(nt0 + u4) * 64];
                        }
                    };
                    load4(0, kf, vf);
#pragma unroll 1
                    for (int nt = 0; nt < NTn; nt += TU) {
                        load4(min(nt + TU, NT - TU), kn, vn);
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4c S[TU];
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) {
                            const float4 m4 = *reinterpret_cast<const float4*>(mrow + 16 * (nt + u4));
                            S[u4] = f32x4c{m4.x, m4.y, m4.z, m4.w};
                        }
                        // S^T += K_h q^T: [k1 | k2] [q1 | q1] + [k1 | k2] [q2 | q2] + [k1 | k3] [q3 | q1]
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) S[u4] = mfma_bf(u32x4{kf[2 * u4].x, kf[2 * u4].y, kf[2 * u4].z, kf[2 * u4].w}, q11, S[u4]);
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) S[u4] = mfma_bf(u32x4{kf[2 * u4].x, kf[2 * u4].y, kf[2 * u4].z, kf[2 * u4].w}, q22, S[u4]);
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4)
                            S[u4] = mfma_bf(u32x4{kf[2 * u4 + 1].x, kf[2 * u4 + 1].y, kf[2 * u4 + 1].z, kf[2 * u4 + 1].w}, q31, S[u4]);
                        softmax_update(S);
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) {                 // O^T += V_h^T P^T, two accumulator chains
                            f32x4c& acc = (u4 & 1) ? o2 : o;
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u4].x, S[u4][0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u4].y, S[u4][1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u4].z, S[u4][2], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u4].w, S[u4][3], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int u4 = 0; u4 < TU; ++u4) { kf[2 * u4] = kn[2 * u4]; kf[2 * u4 + 1] = kn[2 * u4 + 1]; vf[u4] = vn[u4]; }
                    }
                }
                const float l = quarters_sum(lrun);
                const float inv = l > 0.f ? 1.0f / l : 0.f;
                const float4 ov = make_float4((o[0] + o2[0]) * inv, (o[1] + o2[1]) * inv, (o[2] + o2[2]) * inv, (o[3] + o2[3]) * inv);
                unsigned po[6];
                bf_terms<3>(ov.x, ov.y, ov.z, ov.w, po);
#pragma unroll
                for (int tm = 0; tm < 3; ++tm)
                    *reinterpret_cast<uint2*>(sOb + (tm * NTR + lo) * OBP + 8 * wave + 2 * hi) = make_uint2(po[2 * tm], po[2 * tm + 1]);
            }
            __syncthreads();
            // ================= pointer: node tiles over the waves; the local policy's tail on wave 7 =================
            const int nloc = A.has_local ? 1 : 0, W0 = 8 - nloc, skip = nloc ? W0 : 0;
            if (wave >= W0) co_local_tail_call(sT, sX, sdem + ((N1 + 3) & ~3), lo, hi);
            {
                int nt = (wave < W0 ? wave : skip + wave);
                float* orow = rows + (size_t)lo * NP + 4 * hi;
                if constexpr (BF) {
                    uint4 pk[4], pkn[4];
                    auto loadpb = [&](int nt_, uint4 (&d)[4]) {
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) d[kb] = gPb[(size_t)(nt_ * 4 + kb) * 64];
                    };
                    if (nt < NTn) loadpb(nt, pk);
#pragma unroll 1
                    while (nt < NTn) {
                        const int nxt = nt + (nt < skip ? W0 : 8);
                        loadpb(min(nxt, NTn - 1), pkn);
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4c a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            const uint4 q1 = *reinterpret_cast<const uint4*>(sOb + lo * OBP + 16 * kb + 4 * hi);
                            const u32x4 A1 = {pk[kb].x, pk[kb].y, pk[kb].z, pk[kb].w}, B1 = {q1.x, q1.y, q1.z, q1.w};
                            if (kb & 1) a1 = mfma_bf(A1, B1, a1); else a0 = mfma_bf(A1, B1, a0);
                        }
                        const int nb = 16 * nt + 4 * hi;
                        const float p0 = pbv[min(nb, N1 - 1)], p1 = pbv[min(nb + 1, N1 - 1)], p2 = pbv[min(nb + 2, N1 - 1)], p3 = pbv[min(nb + 3, N1 - 1)];
                        *reinterpret_cast<float4*>(orow + 16 * nt) = make_float4(a0[0] + a1[0] + p0, a0[1] + a1[1] + p1, a0[2] + a1[2] + p2, a0[3] + a1[3] + p3);
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) pk[kb] = pkn[kb];
                        nt = nxt;
                    }
                } else {
                    uint4 pk[12], pkn[12];
                    auto loadpk = [&](int nt_, uint4 (&d)[12]) {
#pragma unroll
                        for (int s4 = 0; s4 < 12; ++s4) d[s4] = gPK[(size_t)(nt_ * 12 + s4) * 64];
                    };
                    if (nt < NTn) loadpk(nt, pk);
#pragma unroll 1
                    while (nt < NTn) {
                        const int nxt = nt + (nt < skip ? W0 : 8);
                        loadpk(min(nxt, NTn - 1), pkn);
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4c a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            const u32x4 A1 = {pk[3 * kb].x, pk[3 * kb].y, pk[3 * kb].z, pk[3 * kb].w};
                            const u32x4 A2 = {pk[3 * kb + 1].x, pk[3 * kb + 1].y, pk[3 * kb + 1].z, pk[3 * kb + 1].w};
                            const u32x4 A3 = {pk[3 * kb + 2].x, pk[3 * kb + 2].y, pk[3 * kb + 2].z, pk[3 * kb + 2].w};
                            const unsigned* ob = sOb + lo * OBP + 16 * kb + 4 * hi;
                            const uint4 q1 = *reinterpret_cast<const uint4*>(ob);
                            const uint4 q2 = *reinterpret_cast<const uint4*>(ob + NTR * OBP);
                            const uint4 q3 = *reinterpret_cast<const uint4*>(ob + 2 * NTR * OBP);
                            const u32x4 B1 = {q1.x, q1.y, q1.z, q1.w}, B2 = {q2.x, q2.y, q2.z, q2.w}, B3 = {q3.x, q3.y, q3.z, q3.w};
                            a0 = mfma_bf(A1, B1, a0);
                            a1 = mfma_bf(A1, B2, a1);
                            a0 = mfma_bf(A2, B1, a0);
                            a1 = mfma_bf(A2, B2, a1);
                            a0 = mfma_bf(A1, B3, a0);
                            a1 = mfma_bf(A3, B1, a1);
                        }
                        const int nb = 16 * nt + 4 * hi;
                        const float p0 = pbv[min(nb, N1 - 1)], p1 = pbv[min(nb + 1, N1 - 1)], p2 = pbv[min(nb + 2, N1 - 1)], p3 = pbv[min(nb + 3, N1 - 1)];
                        *reinterpret_cast<float4*>(orow + 16 * nt) = make_float4(a0[0] + a1[0] + p0, a0[1] + a1[1] + p1, a0[2] + a1[2] + p2, a0[3] + a1[3] + p3);
#pragma unroll
                        for (int s4 = 0; s4 < 12; ++s4) pk[s4] = pkn[s4];
                        nt = nxt;
                    }
                }
            }
            __syncthreads();
            // ================= owners: clip, mask, softmax, choice (models.py:405-420) =================
            // ONE read pass over the score row per trajectory (a second one only when sampling): lane l takes the nodes 4 l .. 4 l + 3
            // of every 256-node block (16-byte loads, two blocks in flight), clipped logit x = clip tanh(s + slot terms) on the fly,
            // per-lane online softmax (running maximum m_l, sum of exp(x - m_l), first arg max), merged over the wave at the end:
            // tot = sum_l sum_l exp(m_l - max).  Nothing is written back to the row.
            auto choose = [&](bool dc, int q, size_t bmq, int sn, float addval, int& sl, float& pp) {
                if (!dc) return;
                float* scr = rows + (size_t)q * NP;
                const unsigned long long* mkw = sMaskW + q * NW;
                if (A.has_local && lane < ELG_SLOT_STRIDE) addval += sX[q * CO_XP + CO_XU + lane] * A.inv_ens;
                if (sn >= 0) scr[sn] += addval - dflt;                        // penalty + local terms on the slot nodes (distinct nodes)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                auto logits4 = [&](int blk, float (&x)[4]) {                   // clipped, masked logits of nodes 256 blk + 4 lane + i
                    const int n0 = 256 * blk + 4 * lane;
                    const float4 sv = *reinterpret_cast<const float4*>(scr + n0);
                    const unsigned bits = (unsigned)(mkw[n0 >> 6] >> (n0 & 63)) & 0xFu;     // NP is a multiple of 64: n0 < NP
                    const float svv[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)       // the closed bit as a 0 / -1 word -> + (-inf): no select for the compiler to branch on
                        x[i] = fmaf(A.clip, fast_tanh(svv[i] + dflt), i2f(__builtin_amdgcn_sbfe((int)bits, i, 1) & (int)0xff800000u));
                };
                const int nblk = NP >> 8, tail = (NP & 255) ? 1 : 0;          // (NP = 64 NW: a last partial block of 64 / 128 / 192 nodes)
                float mrun_l = ELG_NEG_INF, srun = 0.f;
                int bn = 0x7fffffff;
                // (a greedy construction without the chosen probabilities -- elg_rollout_args.probs == NULL -- needs the arg max only)
                const bool want_p = A.probs != nullptr || A.full_probs != nullptr || A.mode != ELG_MODE_GREEDY;
                for (int blk = 0; blk < nblk + tail; ++blk) {
                    float x[4] = {ELG_NEG_INF, ELG_NEG_INF, ELG_NEG_INF, ELG_NEG_INF};
                    if (256 * blk + 4 * lane < NP) logits4(blk, x);
                    float bm_ = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
                    if (bm_ > mrun_l) {                                         // strictly greater: the first maximum wins (node order)
#pragma unroll
                        for (int i = 3; i >= 0; --i) if (x[i] == bm_) bn = 256 * blk + 4 * lane + i;
                        if (want_p) srun *= __expf(mrun_l - bm_);               // exp(-inf) = 0 for the first finite block
                        mrun_l = bm_;
                    }
                    if (want_p && mrun_l > ELG_NEG_INF) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) srun += __expf(x[i] - mrun_l);
                    }
                }
                // merge the lanes: global maximum, first node that attains it, normaliser
                const float gmx = wave_max(mrun_l);
                {
                    const float cand = (mrun_l == gmx && bn != 0x7fffffff) ? -(float)bn : -3.0e38f;      // node indices < 2^24: exact in f32
                    const float firstn = -wave_max(cand);
                    bn = firstn < 1.0e9f ? (int)firstn : 0;
                }
                const float tot = want_p ? wave_sum(mrun_l > ELG_NEG_INF ? srun * __expf(mrun_l - gmx) : 0.f) : 1.0f;
                const float inv = 1.0f / tot;
                float* frow = (A.full_probs && t < A.dump_T) ? A.full_probs + (bmq * A.dump_T + t) * N1 : nullptr;
                if (frow)                                                       // tests: probabilities / clipped logits / scores before the clip
                    for (int n = lane; n < N1; n += 64) {
                        const bool masked = (mkw[n >> 6] >> (n & 63)) & 1ull;
                        const float sv = scr[n] + dflt;
                        const float x = masked ? ELG_NEG_INF : A.clip * fast_tanh(sv);
                        frow[n] = A.dump_logits == 2 ? (masked ? ELG_NEG_INF : sv) : A.dump_logits == 1 ? x : (masked ? 0.f : __expf(x - gmx) * inv);
                    }
                int s_ = 0;
                if (A.mode == ELG_MODE_FORCED) s_ = (A.forced && t < A.Tforced) ? A.forced[bmq * A.Tforced + t] : 0;
                else if (A.mode == ELG_MODE_GREEDY) s_ = bn;
                else {
                    // inverse CDF in node order (second pass; 64-node chunks so that the scan runs in node order)
                    const float uni = A.uniforms ? A.uniforms[bmq * A.Tmax + t] : philox_uniform(A.seed, (unsigned)bmq, (unsigned)t);
                    const float target = uni * tot;
                    float run = 0.f;
                    int found = -1, lastpos = 0;
                    for (int ch = 0; ch < NW && found < 0; ++ch) {
                        const int n = lane + 64 * ch;
                        float e = 0.f;
                        if (n < N1 && !((mkw[ch] >> lane) & 1ull)) e = __expf(A.clip * fast_tanh(scr[n] + dflt) - gmx);
                        const float cs_ = wave_scan_incl(e, lane) + run;
                        run = readlane(cs_, 63);
                        const unsigned long long pos = __ballot(e > 0.f);
                        const unsigned long long hit = __ballot(e > 0.f && cs_ > target);
                        if (hit) found = 64 * ch + (int)__builtin_ctzll(hit);
                        if (pos) lastpos = 64 * ch + 63 - (int)__builtin_clzll(pos);
                    }
                    s_ = found >= 0 ? found : lastpos;
                }
                s_ = __builtin_amdgcn_readfirstlane(s_);
                const bool smasked = (mkw[s_ >> 6] >> (s_ & 63)) & 1ull;
                const float xs = smasked ? ELG_NEG_INF : A.clip * fast_tanh(scr[s_] + dflt);
                const float p_ = (xs > ELG_NEG_INF) ? __expf(xs - gmx) * inv : 0.f;
                pp = i2f(__builtin_amdgcn_readfirstlane(f2i(p_)));
                sl = s_;
            };
#pragma unroll
            for (int j = 0; j < NOWN; ++j) choose(dec[j], wave + 8 * j, bm[j], snid[j], addv[j], sel[j], pr[j]);
        }
        auto advance = [&](XmTraj& s1, bool ac, int q, size_t bmq, int sl, float pp) {
            if (!ac) return;
            const int m = m_base + q;
            if (first_move) {
                if (A.mode == ELG_MODE_FORCED) sl = (A.forced && t < A.Tforced) ? __builtin_amdgcn_readfirstlane(A.forced[bmq * A.Tforced + t]) : 0;
                else sl = (!TSP && t == 0) ? 0 : __builtin_amdgcn_readfirstlane(A.starts[m]);
            }
            if (lane == 0) {
                if (A.actions) A.actions[bmq * A.Tmax + t] = sl;
                if (A.probs) A.probs[((size_t)b * A.Tmax + t) * A.M + m] = pp;
            }
            unsigned long long* vis = sVis + q * NW;
            const float sx = xy[2 * sl], sy = xy[2 * sl + 1];
            if (s1.cnt > 0) s1.len += dist2d(s1.cx, s1.cy, sx, sy);
            s1.cx = sx; s1.cy = sy;
            if (TSP) { if (s1.cnt == 0) s1.first = sl; }
            else s1.load = (sl == 0) ? 1.0f : __fsub_rn(s1.load, sdem[sl]);
            if (lane == 0) {
                unsigned long long w = vis[sl >> 6];
                const unsigned long long bit = 1ull << (sl & 63);
                int d = (w & bit) ? 0 : 1;
                w |= bit;
                vis[sl >> 6] = w;
                if (!TSP) {
                    // depot counts as visited exactly while the trajectory stands on it (CVRPEnv.py:214-216)
                    unsigned long long w0 = vis[0];
                    if (sl != 0 && (w0 & 1ull)) { w0 &= ~1ull; d -= 1; vis[0] = w0; }
                }
                sb[0] = i2f(d);
            }
            wave_lds_fence();
            s1.nvis += f2i(sb[0]);
            wave_lds_fence();
            s1.cur = sl;
            s1.cnt += 1;
            if (TSP) {
                if (s1.cnt == N1) { s1.len += dist2d(sx, sy, xy[2 * s1.first], xy[2 * s1.first + 1]); s1.fin = 1; }
            } else if (s1.nvis == N1) s1.fin = 1;
        };
#pragma unroll
        for (int j = 0; j < NOWN; ++j) advance(st[j], act[j], wave + 8 * j, bm[j], sel[j], pr[j]);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NOWN; ++j)
            if (has[j]) { if (A.reward) A.reward[bm[j]] = -st[j].len; if (A.tlen) A.tlen[bm[j]] = st[j].cnt; }
    }
}

template <bool TSP, bool BF>
static int launch_fwd_xm(const elg_rollout_args& A, hipStream_t stream) {
    if (!A.scratch) return fail(ELG_EINVAL, "rollout: N1 > 1024 needs the scratch workspace (elg_rollout_scratch_floats)");
    if (A.N1 > 8192) return fail(ELG_ENOTIMPL, "rollout: N1 > 8192 not built");
    if (A.trA || A.trMask) return fail(ELG_ENOTIMPL, "rollout: no training rows for N1 > 1024");
    const int NW = (A.N1 + 63) / 64, NP = 64 * NW, NTt = NP / 16;
    const size_t lds = ((size_t)16 * 3 * 68 + (size_t)16 * NW * 4 + (A.has_local ? 16 * CO_XP + CL_SIZE : 0) + ((A.N1 + 3) & ~3) +
                        (size_t)8 * ELG_SB_MIN) * 4;
    if (lds > 163840 - 256) return fail(ELG_EINVAL, "xm rollout: LDS budget exceeded");
    auto kern = rollout_fwd_xm_kernel<TSP, BF>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, "xm rollout: hipFuncSetAttribute failed");
    elg_rollout_args B2 = A;
    B2.tiles = (A.M + 15) / 16;
    (void)hipGetLastError();
    if (BF)
        hipLaunchKernelGGL(mt_repack_bf16_kernel, dim3(((8 * NTt + 4 * NTt + 4 * NTt) * 64 + 255) / 256, A.B), dim3(256), 0, stream, A.Kmat, A.Vmat,
                           A.PK, reinterpret_cast<unsigned*>(A.scratch), A.N1, NP);
    else
        hipLaunchKernelGGL(mt_repack_kernel, dim3((2 * NP * 32 + NP * 16 + 255) / 256, A.B), dim3(256), 0, stream, A.Kmat, A.Vmat, A.PK, A.scratch, A.N1, NP);
    note_kernel(ELG_KERNEL_XM);
    hipLaunchKernelGGL(kern, dim3(B2.B * B2.tiles), dim3(512), lds, stream, B2);
    return launch_status("rollout_fwd_xm");
}

// SMALL: N1 <= 104 in the two-chunk build (13 row groups instead of 16)
template <int NCH, bool TSP, bool LDSK, int WAVES, bool TRAIN = false>
static int launch_fwd(const elg_rollout_args& A, hipStream_t stream) {
    if (NCH == 2 && A.N1 <= 104) return launch_fwd_impl<NCH, TSP, LDSK, WAVES, TRAIN, (NCH == 2)>(A, stream);
    return launch_fwd_impl<NCH, TSP, LDSK, WAVES, TRAIN, false>(A, stream);
}

template <bool TSP>
static int dispatch_fwd(const elg_rollout_args& A, hipStream_t stream) {
    const int nch = (A.N1 + 63) / 64;
    if (A.lds_stage && A.N1 > 16 * CO_NT) return fail(ELG_EINVAL, "lds_stage needs N1 <= 112");
    if (A.waves != 8) return fail(ELG_EINVAL, "waves must be 8");
    // lds_stage = "keep the instance's tables on chip": MFMA operand images (cooperative kernel, N1 <= 112) or the
    // LDS copies of the one-wavefront-per-trajectory kernel (N1 <= 104; 105..112 fall back to its L2 variant)
    // local_size > 47 (K + 1 > ELG_SLOT_STRIDE slots): the one-wavefront kernels only -- one slot per lane, up to 64; the
    // cooperative / streaming / N1 > 1024 matrix kernels and the saved training rows are built on 48-wide slot blocks.  Their
    // LDS-staged form leaves no room for the wider slot scratch at the largest sizes: its tables then stay in L2.
    const bool wide = kmax_of(A) + 1 > ELG_SLOT_STRIDE;
    if (wide && (A.trA || A.trMask)) return fail(ELG_ENOTIMPL, "rollout: training rows are built for local_size <= 47 (replay backward above)");
    if (wide && A.variant != 0 && A.variant != 1 && A.variant != 2) return fail(ELG_ENOTIMPL, "rollout: local_size > 47 runs the one-wavefront kernels (variant 0, 1 or 2)");
    // (LDS-staged one-wavefront kernel: K | V | PK + demands + coordinates + the per-wave scratch must fit the CU's 160 KB: launch_fwd_impl)
    const size_t ldsk_bytes = (size_t)3 * A.N1 * ELG_E * 4 + (size_t)((A.N1 + 3) & ~3) * 4 + 16 + (size_t)8 * sb_floats_of(nch, kmax_of(A)) * 4 +
                              (size_t)((2 * A.N1 + 3) & ~3) * 4;
    const bool lds = A.lds_stage != 0 && A.N1 <= 104 && ldsk_bytes <= 163840;
#define ELG_GO(NCHV, L, W) return launch_fwd<NCHV, TSP, L, W>(A, stream)
    if (A.ens > 1) {        // ensemble_size > 1: the one-wavefront-per-trajectory kernel walks the members
        if (A.trA || A.trMask) return fail(ELG_ENOTIMPL, "rollout: training rows are not built for ensemble_size > 1 (replay backward)");
        if (nch > 16) return fail(ELG_ENOTIMPL, "rollout: ensemble_size > 1 is built for N1 <= 1024");
        if (nch == 1) { if (lds) ELG_GO(1, true, 8); else ELG_GO(1, false, 8); }
        if (nch == 2) { if (lds) ELG_GO(2, true, 8); else ELG_GO(2, false, 8); }
        if (nch <= 4) { ELG_GO(4, false, 8); }
        if (nch <= 8) { ELG_GO(8, false, 8); }
        ELG_GO(16, false, 8);
    }
    if (A.lds_stage && A.waves == 8 && A.N1 >= 4 && A.N1 <= 16 * CO_NT && !A.use_state && A.do_decode && A.do_update &&
        A.max_steps <= 0 && (A.variant == 0 || A.variant == 4 || A.variant == 5) && !wide) {
        // fused rollout at the training scale: lockstep trajectories, tables as MFMA operands in registers (variant 4: the
        // split-group form of the same kernel, two independent 4-wave groups per workgroup -- same results bit for bit)
        const bool train = A.trA || A.trMask;      // training forward (glimpse weights saved, or recomputed from the mask rows)
        if (train && (!A.trPC || !A.trCsel || !A.trQ || !A.trO)) return fail(ELG_EINVAL, "rollout: incomplete training rows");
        return launch_fwd_coop_any(A, stream, TSP, train, A.variant == 4 ? 1 : A.variant == 5 ? 2 : 0);
    }
    if (A.variant == 4 || A.variant == 5) return fail(ELG_EINVAL, "rollout: variants 4 / 5 (split-group cooperative kernels) need lds_stage, 4 <= N1 <= 112, a fused rollout");
    // variants 2 / 3 name a kernel, whatever the size (the tests' A/B references): checked before the shape dispatch
    if ((A.variant == 2 || A.variant == 3) && !A.use_state && A.do_decode && A.do_update && A.max_steps <= 0) {
        if (A.trA || A.trMask) return fail(ELG_ENOTIMPL, "rollout: training rows for N1 > 1024 not built");
        if (A.variant == 3) return A.precision == 1 ? launch_fwd_xm<TSP, true>(A, stream) : launch_fwd_xm<TSP, false>(A, stream);
        return launch_fwd_xl<TSP>(A, stream);
    }
    const bool mt_shape = !lds && nch > 2 && nch <= 16 && !A.use_state && A.do_decode && A.do_update && A.max_steps <= 0 && A.variant == 0 && !wide;
    if (A.trMask && !A.trA && !mt_shape) return fail(ELG_EINVAL, "rollout: this kernel needs trA (mask-only rows: cooperative / streaming kernels)");
    if (A.trA) {        // training forward: saves the backward rows; built for N1 <= 128, 8 waves
        if (!A.trPC || !A.trCsel || !A.trQ || !A.trO) return fail(ELG_EINVAL, "rollout: incomplete training rows");
        if (A.use_state) return fail(ELG_EINVAL, "rollout: training rows need the fused rollout");
        if (nch == 1) { if (lds) return launch_fwd<1, TSP, true, 8, true>(A, stream); return launch_fwd<1, TSP, false, 8, true>(A, stream); }
        if (nch == 2) {
            if (lds) return launch_fwd<2, TSP, true, 8, true>(A, stream);
            return launch_fwd<2, TSP, false, 8, true>(A, stream);
        }
        return fail(ELG_ENOTIMPL, "rollout: training rows for N1 > 128 not built");
    }
    if (nch == 1) { if (lds) ELG_GO(1, true, 8); else ELG_GO(1, false, 8); }
    if (nch == 2) {
        if (lds) ELG_GO(2, true, 8);
        else ELG_GO(2, false, 8);
    }
    if (lds) return fail(ELG_EINVAL, "lds_stage needs N1 <= 104");
    if (!A.use_state && A.do_decode && A.do_update && A.max_steps <= 0 && A.variant == 0 && !wide) {
        // fused rollout of a large instance: node-tiled kernel (tables shared through LDS tiles)
        if (nch <= 4) return launch_fwd_mt<4, TSP>(A, stream);
        if (nch <= 8) return launch_fwd_mt<8, TSP>(A, stream);
        if (nch <= 16) return launch_fwd_mt<16, TSP>(A, stream);
    }
    const bool fused = !A.use_state && A.do_decode && A.do_update && A.max_steps <= 0;
    if (fused && !wide && (A.variant == 3 || (A.variant == 0 && nch > 16))) {
        // Vrp-Set-XXL scale (or variant 3: the same kernel at any size, for the tests): matrix phases of the streaming kernel,
        // runtime chunk loops in the owners, score rows in the scratch
        if (A.trA || A.trMask) return fail(ELG_ENOTIMPL, "rollout: training rows for N1 > 1024 not built");
        if (A.ens > 1) return fail(ELG_ENOTIMPL, "rollout: ensemble_size > 1 is built for N1 <= 1024");
        return A.precision == 1 ? launch_fwd_xm<TSP, true>(A, stream) : launch_fwd_xm<TSP, false>(A, stream);
    }
    if (fused && (A.variant == 2 || nch > 16)) {
        if (A.trA || A.trMask) return fail(ELG_ENOTIMPL, "rollout: training rows for N1 > 1024 not built");
        return launch_fwd_xl<TSP>(A, stream);                 // one wavefront per trajectory, runtime node loops (the tests' reference)
    }
    if (nch <= 4) { ELG_GO(4, false, 8); }
    if (nch <= 8) { ELG_GO(8, false, 8); }
    if (nch <= 16) { ELG_GO(16, false, 8); }
#undef ELG_GO
    return fail(ELG_ENOTIMPL, "N1 > 1024: only the fused rollout is built (step-wise protocol up to 1024 nodes)");
}

}  // namespace elg

using namespace elg;

extern "C" {
int64_t elg_rollout_scratch_floats(int32_t B, int32_t M, int32_t N1, int32_t variant) {
    if (B <= 0 || M <= 0 || N1 <= 0) return 0;
    if (variant == 2) return (int64_t)B * M * N1;                                  // score rows of rollout_fwd_xl_kernel
    if (variant == 3 || (variant == 0 && N1 > 1024)) {                             // rollout_fwd_xm_kernel: tables + score rows
        const int64_t NP = 64 * (int64_t)((N1 + 63) / 64);
        return (int64_t)B * NP * (MT_KB_WORDS + ELG_E + MT_PKB_WORDS) + (int64_t)B * ((M + 15) / 16) * 16 * NP;
    }
    if (N1 > 1024) return (int64_t)B * M * N1;
    if (N1 > 128 && variant == 0) {                                                // fragment-major K / V / PK copies
        const int nch = (N1 + 63) / 64, NP = 64 * (nch <= 4 ? 4 : nch <= 8 ? 8 : 16);
        return (int64_t)B * NP * (MT_KB_WORDS + ELG_E + MT_PKB_WORDS);
    }
    return 0;
}


const char* elg_version(void) { return "elg-hip 0.1 (gfx950)"; }
int elg_rollout_last_kernel(void) { return elg::g_kernel; }
const char* elg_last_error(void) { return elg::last_error(); }

int elg_aug8(const float* xy_in, float* xy_out, int B, int N, void* stream) {
    if (B <= 0 || N <= 0) return fail(ELG_EINVAL, "aug8: empty input");
    const int n = B * N;
    (void)hipGetLastError();
    hipLaunchKernelGGL(aug8_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, xy_in, xy_out, B, N);
    return launch_status("aug8");
}

int elg_dist_matrix(const float* xy, float* dist, int B, int N, void* stream) {
    if (B <= 0 || N <= 0) return fail(ELG_EINVAL, "dist_matrix: empty input");
    (void)hipGetLastError();
    hipLaunchKernelGGL(dist_matrix_kernel, dim3(N, B), dim3(128), 0, (hipStream_t)stream, xy, dist, N);
    return launch_status("dist_matrix");
}

int elg_nbr_tables(const float* xy, int32_t* nbr_idx, float* nbr_dist, float* nbr_theta, int B, int N, void* stream) {
    if (B <= 0 || N <= 0) return fail(ELG_EINVAL, "nbr_tables: empty input");
    if (N > 8192) return fail(ELG_ENOTIMPL, "nbr_tables: N > 8192");
    int NP = 64;
    while (NP < N) NP <<= 1;
    const int threads = NP >= 512 ? 256 : 64;
    (void)hipGetLastError();
    hipLaunchKernelGGL(nbr_tables_kernel, dim3(N, B), dim3(threads), (size_t)NP * 8, (hipStream_t)stream, xy, nbr_idx,
                       nbr_dist, nbr_theta, N, NP);
    return launch_status("nbr_tables");
}

int elg_route_length(const float* xy, const int64_t* tour, float* out, int B, int M, int T, int N, int rounding,
                     void* stream) {
    if (B <= 0 || M <= 0 || T <= 0) return fail(ELG_EINVAL, "route_length: empty input");
    const int n = B * M;
    (void)hipGetLastError();
    hipLaunchKernelGGL(route_length_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, xy,
                       reinterpret_cast<const long long*>(tour), out, B, M, T, N, rounding);
    return launch_status("route_length");
}

int elg_rollout_fwd(const elg_rollout_args* a, void* stream) {
    if (!a) return fail(ELG_EINVAL, "null args");
    const elg_rollout_args& A = *a;
    if (A.B <= 0 || A.M <= 0 || A.N1 <= 1 || A.tiles <= 0) return fail(ELG_EINVAL, "rollout: bad sizes");
    if (A.K < 0 || A.K + 1 > ELG_SLOT_MAX) return fail(ELG_EINVAL, "rollout: local_size must be <= 63");
    if (A.has_local && !A.loc) return fail(ELG_EINVAL, "rollout: has_local without tables");
    if (A.ens > 1) {
        if (A.ens > ELG_MAX_ENS) return fail(ELG_ENOTIMPL, "rollout: ensemble_size > 4 not built");
        if (A.problem != ELG_PROBLEM_CVRP) return fail(ELG_EINVAL, "rollout: the TSP decoder has one local policy (TSP/models.py:223-225)");
        if (A.Kens[0] != A.K) return fail(ELG_EINVAL, "rollout: Kens[0] must equal K (local_size[0])");
        for (int i = 0; i < A.ens; ++i)
            if (A.Kens[i] < 0 || A.Kens[i] + 1 > ELG_SLOT_MAX) return fail(ELG_EINVAL, "rollout: local_size must be <= 63");
    }
    if (A.mode == ELG_MODE_FORCED && !A.forced) return fail(ELG_EINVAL, "rollout: forced mode without actions");
    if (A.precision != 0 && A.precision != 1) return fail(ELG_EINVAL, "rollout: precision 0 (f32) or 1 (bf16 table products)");
    if (A.problem == ELG_PROBLEM_CVRP) return dispatch_fwd<false>(A, (hipStream_t)stream);
    if (A.problem == ELG_PROBLEM_TSP) return dispatch_fwd<true>(A, (hipStream_t)stream);
    return fail(ELG_EINVAL, "rollout: unknown problem");
}

}  // extern "C"
