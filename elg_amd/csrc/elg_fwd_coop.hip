// The cooperative rollout kernel for N1 <= 112 (the CVRP/TSP-100 training and evaluation shape); split out of csrc/elg_fwd.hip
// in round 6 (its own translation unit: the file compiles beside the streaming kernels).
#include "elg_coop.h"
#include <string>
#include <cstdlib>

namespace elg {

// =============================================================================================
// Cooperative rollout kernel for N1 <= 112 (the CVRP/TSP-100 training shape): lockstep + matrix cores.
//
// rollout_fwd_kernel gives every trajectory its own wavefront for the whole decode step; its glimpse and pointer
// stages are then LDS-bandwidth bound (each trajectory re-reads the instance's K, V and PK -- 155 KB -- every
// step).  Here the <= 32 trajectories a workgroup owns advance in lockstep, and the three table contractions of
// a step run once for all of them on v_mfma_f32_16x16x4_f32 with the tables as step-invariant operands held in
// REGISTERS: wave h keeps K_h and V_h (its head's 16 channels of every node: 56 VGPRs) and wave w < 7 keeps the
// 16-node slice w of PK (32 VGPRs).  LDS only carries the per-step exchange: queries in, glimpse outputs back,
// pointer scores out.  Per step:
//   owners   (wave w owns trajectories 4 w .. 4 w + 3): mask, query q, k-NN slots -> LDS
//   glimpse  (wave h = head h): S^T[n][traj] = K_h[n] . q_h[traj] for 2 x 7 tiles, softmax over n (registers +
//            two cross-quarter shuffles), O^T[d][traj] = sum_n V_h[n][d] P^T[n][traj]; the D tile of the first
//            product is the B operand of the second (node on the k-slot), nothing is transposed
//   pointer  (wave w < 7 = node tile w): s^T[n][traj] = sum_c PK[n][c] o[traj][c] + pb[n] -> LDS
//   owners:  the wave's four trajectories side by side, 16 lanes each (state in the row's registers): clip / mask /
//            softmax / choice (DPP row reductions and scans), environment transition, then the next step's mask
//            words (row slices of ballots), query row and k-NN slots (rank = DPP row scan over the sorted neighbours).
// Trajectory state lives in LDS between phases (12 dwords), wave-uniform in SGPRs while a wave works on it.
// =============================================================================================

// BF: the bf16 throughput mode (elg_rollout_args.precision = 1; BASELINE configs[1] names it): the three table products of a
// step -- glimpse scores K_h q^T, glimpse output V_h^T P^T, pointer scores PK o^T -- run on v_mfma_f32_16x16x32_bf16 with the
// operands rounded to bf16 (tables, query, softmax numerators, glimpse output) and f32 accumulation; masks, softmax, clip,
// choice, the local policy and the environment stay f32.  The f32 instantiation is the parity mode and the default.
template <bool TSP, bool TRAIN, bool BF, bool LEAN>
__global__ __launch_bounds__(512) void rollout_fwd_coop_kernel(const elg_rollout_args A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int N1 = A.N1;
    const int G = gridDim.x;
    int u = blockIdx.x;
    if ((G & 7) == 0) u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int b = u / A.tiles, tile = u % A.tiles;
    const int tile_m = (A.M + A.tiles - 1) / A.tiles;
    const int m_lo = tile * tile_m, m_hi = min(A.M, m_lo + tile_m);
    const size_t NE = (size_t)N1 * ELG_E;
    const size_t Rcap = (size_t)A.Tmax * A.M;

    // ---- LDS: exchange rows | scores | masks | states | dem | xy | per-wave scratch
    float* sQ = lds;                                                  // [32][CO_QP]  q in, glimpse output back
    float* sSc = sQ + CO_MAXTR * CO_QP;                               // [32][CO_SP]
    unsigned long long* sMask = reinterpret_cast<unsigned long long*>(sSc + CO_MAXTR * CO_SP);   // [32][2]
    int* sState = reinterpret_cast<int*>(sMask + 2 * CO_MAXTR);       // [32][16]
    float* sdem = reinterpret_cast<float*>(sState + 16 * CO_MAXTR);
    float* sxy = sdem + ((N1 + 3) & ~3);
    float* sX = sxy + ((2 * N1 + 3) & ~3);                            // [32][CO_XP] slot exchange blocks
    float* sT = sX + CO_MAXTR * CO_XP;                                // local-policy tables
    float* sP = sT + CL_SIZE;                                         // [7][32][64] PK operand image
    float* sPb = sP + CO_NT * 32 * 64;                                // [112] pointer bias
    float* sb = sPb + 16 * CO_NT + wave * SbSize<2>::value;
    float* sO1 = sPb + 16 * CO_NT + 8 * SbSize<2>::value;             // [2 groups][2][64][4] o' of the local policy (head units -> tail)
    if (!TSP)
        for (int i = tid; i < N1; i += 512) sdem[i] = A.demand[(size_t)b * N1 + i];
    for (int i = tid; i < 2 * N1; i += 512) sxy[i] = A.xy[(size_t)b * N1 * 2 + i];
    if (LEAN || A.has_local) co_stage_local(A.loc, sT, tid, 512);

    Inst I;
    I.K = nullptr; I.V = nullptr; I.PK = nullptr;
    I.pb = A.pb + (size_t)b * N1;
    I.Q1 = A.Q1 + b * NE;
    I.Q2 = TSP ? A.Q2 + b * NE : nullptr;
    I.wl = A.wl;
    I.xy = sxy;
    I.dem = sdem;
    I.nidx = A.nbr_idx + (size_t)b * N1 * N1;
    I.ndist = A.nbr_dist + (size_t)b * N1 * N1;
    I.ntheta = A.nbr_theta + (size_t)b * N1 * N1;
    I.loc = A.loc;

    // ---- step-invariant MFMA operands in registers
    // glimpse: wave = head h.  kop[nt][kk] = K[n = 16 nt + lo][16 h + 4 hi + kk]   (A operand of S^T = K Q^T)
    //                         vop[nt][v]  = V[n = 16 nt + 4 hi + v][16 h + lo]     (A operand of O^T = V^T P^T)
    // pointer: wave = node tile w (w < 7).  pop[s] = PK[n = 16 w + lo][4 s + hi]   (A operand of s^T = PK o^T)
    const float* gK = A.Kmat + b * NE + wave * 16;
    const float* gV = A.Vmat + b * NE + wave * 16;
#define ELG_CO_LOAD_KV()                                                                                   \
    _Pragma("unroll") for (int nt = 0; nt < CO_NT; ++nt) {                                                  \
        const int n = 16 * nt + lo_t;                                                                       \
        {   /* one 16-byte load: channels 4 hi .. 4 hi + 3 (the MFMA visits the 16 channels in the order 4 hi + kk) */ \
            /* rows past N1 re-read row N1 - 1 (finite): their nodes are closed in every mask, so S is replaced by -inf */ \
            /* and the weight that multiplies the V row is exactly 0 -- no zeroing of the operands needed              */ \
            const float4 k4_ = ld_off<float4>(gK, 4u * (unsigned)(min(n, N1 - 1) * ELG_E + 4 * hi_t));                     \
            kop[nt][0] = k4_.x; kop[nt][1] = k4_.y; kop[nt][2] = k4_.z; kop[nt][3] = k4_.w;                 \
        }                                                                                                   \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                     \
            const int n2 = 16 * nt + 4 * hi_t + v;                                                          \
            vop[nt][v] = ld_off<float>(gV, 4u * (unsigned)(min(n2, N1 - 1) * ELG_E + lo_t));                               \
        }                                                                                                   \
    }
    // bf16 mode: kopb[nt] = K[n = 16 nt + lo][16 h + 4 hi + j], j < 4, in k-slots (hi, 0..3); k-slots (hi, 4..7) are zero (the head
    // has 16 channels, the instruction contracts 32).  vopb[p] = V[n][16 h + lo] for the eight nodes n = 32 p + 4 hi + j (j < 4)
    // and 32 p + 16 + 4 hi + j - 4 (j >= 4): the D tiles of the score product for node tiles 2 p and 2 p + 1, side by side, are
    // the B operand (rows past N1 - 1 re-read row N1 - 1: their weights are exactly 0).
#define ELG_CO_LOAD_KV_BF()                                                                                 \
    _Pragma("unroll") for (int nt = 0; nt < CO_NT; ++nt) {                                                  \
        const int n = 16 * nt + lo_t;                                                                       \
        const float4 k4_ = ld_off<float4>(gK, 4u * (unsigned)(min(n, N1 - 1) * ELG_E + 4 * hi_t));                     \
        kopb[nt] = u32x4{pk_bf16(k4_.x, k4_.y), pk_bf16(k4_.z, k4_.w), 0u, 0u};                             \
    }                                                                                                       \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                         \
        float va_[4], vb_[4];                                                                               \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                     \
            va_[v] = ld_off<float>(gV, 4u * (unsigned)(min(32 * p + 4 * hi_t + v, N1 - 1) * ELG_E + lo_t));                \
            vb_[v] = ld_off<float>(gV, 4u * (unsigned)(min(32 * p + 16 + 4 * hi_t + v, N1 - 1) * ELG_E + lo_t));                \
        }                                                                                                   \
        vopb[p] = u32x4{pk_bf16(va_[0], va_[1]), pk_bf16(va_[2], va_[3]), pk_bf16(vb_[0], vb_[1]), pk_bf16(vb_[2], vb_[3])}; \
    }
    {
        // the PK operand image goes to LDS ([tile][k-step][lane], read back conflict-free): with it in registers
        // too, the batched local policy no longer fits the 256-VGPR budget of 2 waves/SIMD
        const int np = 16 * wave + lo;
        const float* gP = A.PK + b * NE + (size_t)min(np, N1 - 1) * ELG_E;
        // [tile][channel group g][lane][j] = PK[16 tile + lo][16 g + 4 hi + j]: the four k-steps of a channel group are one
        // ds_read_b128 (k-slot (g, j, hi) stands for channel 16 g + 4 hi + j in both operands)
        if (!BF && wave < CO_NT)
            for (int g = 0; g < 8; ++g) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (np < N1) x = *reinterpret_cast<const float4*>(gP + 16 * g + 4 * hi);
                *reinterpret_cast<float4*>(sP + ((wave * 8 + g) * 64 + lane) * 4) = x;
            }
        // bf16: [tile][32-channel block g][lane] = the eight channels 32 g + 8 hi .. + 7 of PK[16 tile + lo] as one 16-byte
        // A operand of v_mfma_f32_16x16x32_bf16 (k-slot (hi, j) stands for channel 32 g + 8 hi + j in both operands)
        if (BF && wave < CO_NT)
            for (int g = 0; g < 4; ++g) {
                uint4 x = make_uint4(0u, 0u, 0u, 0u);
                if (np < N1) {
                    const float4 a = *reinterpret_cast<const float4*>(gP + 32 * g + 8 * hi);
                    const float4 c = *reinterpret_cast<const float4*>(gP + 32 * g + 8 * hi + 4);
                    x = make_uint4(pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(c.x, c.y), pk_bf16(c.z, c.w));
                }
                *reinterpret_cast<uint4*>(sP + ((wave * 4 + g) * 64 + lane) * 4) = x;
            }
    }
    for (int i = tid; i < 16 * CO_NT; i += 512) sPb[i] = (i < N1) ? I.pb[i] : 0.f;
    __syncthreads();

    // groups of <= 32 lockstep trajectories, evenly sized (100 trajectories = 4 x 25, not 3 x 32 + 4: a group of 4 costs
    // as many barriers per step as a group of 32)
    const int n_groups = (m_hi - m_lo + CO_MAXTR - 1) / CO_MAXTR;
    const int g_size = n_groups > 0 ? (m_hi - m_lo + n_groups - 1) / n_groups : CO_MAXTR;
    for (int g_lo = m_lo; g_lo < m_hi; g_lo += g_size) {
        const int ntraj = min(g_size, m_hi - g_lo);
        const bool two_rt = ntraj > 16;
        // ---- reset: every trajectory at the depot / nowhere, step 0 (nothing to decode at t = 0)
        for (int q = wave; q < CO_MAXTR; q += 8) {
            Traj<2> st;
            st.cur = 0; st.first = 0; st.cnt = 0; st.fin = (q < ntraj) ? 0 : 1; st.load = 1.0f; st.len = 0.f;
            st.cx = 0.f; st.cy = 0.f; st.vis[0] = 0ull; st.vis[1] = 0ull;
            co_store_state<TSP>(sState + 16 * q, st, lane);
            if (lane == 0) { sMask[2 * q] = ~0ull; sMask[2 * q + 1] = ~0ull; }
            if (lane < 33) *reinterpret_cast<float4*>(sQ + q * CO_QP + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = lane; i < CO_XP; i += 64) sX[q * CO_XP + i] = (i >= CO_XS && i < CO_XPEN) ? i2f(-1) : 0.f;
        }
        __syncthreads();
        const int step_cap = TSP ? N1 : 2 * N1 + 2;
        CoRow row;                                                  // batched owners: the row's trajectory (registers)
        float ubuf = 0.f;                                           // the row's next 16 sampling uniforms, one per lane
        row.cur = 0; row.first = 0; row.cnt = 0; row.fin = (4 * wave + (lane >> 4) < ntraj) ? 0 : 1;
        row.load = 1.0f; row.len = 0.f; row.cx = 0.f; row.cy = 0.f; row.v0 = 0ull; row.v1 = 0ull;
        StampCtx sc;
#ifdef ELG_STAMPS
        for (int i = 0; i < 16; ++i) sc.acc[i] = 0.f;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc.last) :: "memory");
#endif
        float kop[CO_NT][4], vop[CO_NT][4];                         // (re)loaded at the end of every owners' phase
        u32x4 kopb[CO_NT], vopb[4];                                 // their bf16 forms (BF)
#pragma unroll
        for (int nt = 0; nt < CO_NT; ++nt) {
#pragma unroll
            for (int v = 0; v < 4; ++v) { kop[nt][v] = 0.f; vop[nt][v] = 0.f; }
            kopb[nt] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) vopb[p] = u32x4{0u, 0u, 0u, 0u};
        for (int t = 0; t < step_cap && t < A.Tmax; ++t) {
            // Opaque per-iteration copy of the lane id: every address below is a function of it, so the compiler cannot
            // hoist the (loop-invariant) address arithmetic of ~300 loads out of the step loop -- it did, and then
            // spilled ~250 VGPRs of precomputed 64-bit addresses; recomputing them costs a few hundred VALU ops a step.
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            const int lo_t = lane_t & 15, hi_t = lane_t >> 4;
            const bool decode_step = TSP ? (t >= 1) : (t >= 2);     // uniform over the workgroup: lockstep
            ELG_STAMP(sc, 15);
            if (decode_step) {
                // =============== glimpse: wave = head ===============
                // (kop / vop: this head's K_h / V_h operand images, 56 registers, re-read from L2 for every step right
                // after the owners' phase so the latency hides behind the barrier.  The instance's 103 KB of K / V stay
                // L2-resident; keeping the images live across the whole step does not fit beside the batched local policy.)
                // local policy, stage 1: wave w = (group w >> 2, head w & 3) -- independent of the glimpse, under whose MFMAs it runs
                // (its five stages are spread over the four MFMA loops of the glimpse below, in program order, so that the scheduler has
                // independent VALU to put into the MFMA shadows; a wave without a unit -- one trajectory group only -- computes on
                // its own group's blocks and stores nothing)
                const bool lh_on = (LEAN || A.has_local) && (wave < 4 || two_rt);
                const int lh_h = wave & 3, lh_dt = lh_h >> 1;
                const float* LX = sX + ((lh_on ? (wave >> 2) : 0) * 16 + lo_t) * CO_XP;
                f32x4c lf[3][3], lal[3];
                bool lmsk[3][4];
                float lmx = ELG_NEG_INF, lden = 0.f, lF[3] = {0.f, 0.f, 0.f};
                f32x4c lP = {0.f, 0.f, 0.f, 0.f};
                const float4 la4 = *reinterpret_cast<const float4*>(sT + CL_LA + 4 * lh_h);
                auto lh_score = [&](int jt) {
                    const int4 sl = *reinterpret_cast<const int4*>(LX + CO_XS + 16 * jt + 4 * hi_t);
                    lmsk[jt][0] = sl.x < 0; lmsk[jt][1] = sl.y < 0; lmsk[jt][2] = sl.z < 0; lmsk[jt][3] = sl.w < 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float4 tq = *reinterpret_cast<const float4*>(LX + CO_XF + k * ELG_SLOT_STRIDE + 16 * jt + 4 * hi_t);
                        lf[k][jt] = f32x4c{tq.x, tq.y, tq.z, tq.w};
                    }
                    const float4 lt4 = *reinterpret_cast<const float4*>(sT + CL_LTT + lh_h * 48 + 16 * jt + 4 * hi_t);
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
                };
                auto lh_exp = [&](int jt) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float e = lmsk[jt][v] ? 0.f : __expf(lal[jt][v] - lmx);
                        lal[jt][v] = e;
                        lden += e;
                    }
                };
                float lrden = 0.f;
                auto lh_norm = [&](int jt) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float a = lal[jt][v] * lrden;
                        lal[jt][v] = a;
#pragma unroll
                        for (int k = 0; k < 3; ++k) lF[k] = fmaf(a, lf[k][jt][v], lF[k]);
                    }
                };
                auto lh_mfma = [&](int jt) {
                    const float4 a4 = *reinterpret_cast<const float4*>(sT + CL_LCVT + (16 * lh_dt + lo_t) * CL_Q + 16 * jt + 4 * hi_t);
                    lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, lal[jt][0], lP, 0, 0, 0);
                    lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, lal[jt][1], lP, 0, 0, 0);
                    lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, lal[jt][2], lP, 0, 0, 0);
                    lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, lal[jt][3], lP, 0, 0, 0);
                };
                // Both trajectory tiles (rt = 0: trajectories 0-15, rt = 1: 16-31) are in flight together so that the VALU of
                // one hides under the MFMAs of the other (in program order: S(0) | S(1) with exp(0) | O(0) with exp(1) | O(1));
                // the O accumulators take the unnormalised weights and are scaled by 1 / den once.
                {
                    const float cs = 0.25f * 1.4426950408889634f;
                    float qb[2][4];
                    u32x4 qbb[2];                                       // bf16: the query's four channels in k-slots (hi, 0..3)
                    f32x4c sc[2][CO_NT];
                    float mx[2], cm[2], den[2] = {0.f, 0.f};
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const int traj = 16 * rt + lo_t;
                        const float4 q4 = *reinterpret_cast<const float4*>(sQ + traj * CO_QP + 16 * wave + 4 * hi_t);   // channels 4 hi + kk, as kop
                        qb[rt][0] = q4.x; qb[rt][1] = q4.y; qb[rt][2] = q4.z; qb[rt][3] = q4.w;
                        qbb[rt] = u32x4{pk_bf16(q4.x, q4.y), pk_bf16(q4.z, q4.w), 0u, 0u};
                        mx[rt] = -1e30f;                                // finite floor: a fully closed row gives exp2(-inf) = 0
                    }
                    auto s_tile = [&](int rt, int nt) {                // S^T tile: the additive mask (0 / -inf, left in the score row by
                        // the owners; nodes past N1 and missing trajectories are -inf) is the accumulator input of the 4 MFMAs
                        const float4 m4 = *reinterpret_cast<const float4*>(sSc + (16 * rt + lo_t) * CO_SP + 16 * nt + 4 * hi_t);
                        f32x4c acc = {m4.x, m4.y, m4.z, m4.w};
                        if (BF) acc = mfma_bf(kopb[nt], qbb[rt], acc);
                        else {
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kop[nt][kk], qb[rt][kk], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v) mx[rt] = fmaxf(mx[rt], acc[v]);
                        sc[rt][nt] = acc;
                    };
                    auto e_tile = [&](int rt, int nt) {                // softmax numerators: exp2((s - max) log2(e) / 4), one fma + v_exp
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(sc[rt][nt][v], cs, cm[rt]));
                            sc[rt][nt][v] = e;
                            den[rt] += e;
                        }
                    };
                    f32x4c o[2], o2[2];                                // two chains per trajectory tile: dependent MFMAs stall
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) { o[rt] = f32x4c{0.f, 0.f, 0.f, 0.f}; o2[rt] = f32x4c{0.f, 0.f, 0.f, 0.f}; }
                    auto o_tile = [&](int rt, int nt) {
                        if (BF) {
                            // node tiles nt - 1 and nt in one instruction, issued once the odd tile's numerators exist (and for the
                            // unpaired last tile with an empty upper half)
                            if (nt & 1) {
                                const u32x4 pb_ = {pk_bf16(sc[rt][nt - 1][0], sc[rt][nt - 1][1]), pk_bf16(sc[rt][nt - 1][2], sc[rt][nt - 1][3]),
                                                   pk_bf16(sc[rt][nt][0], sc[rt][nt][1]), pk_bf16(sc[rt][nt][2], sc[rt][nt][3])};
                                if (nt & 2) o2[rt] = mfma_bf(vopb[nt >> 1], pb_, o2[rt]);
                                else o[rt] = mfma_bf(vopb[nt >> 1], pb_, o[rt]);
                            } else if (nt == CO_NT - 1) {
                                const u32x4 pb_ = {pk_bf16(sc[rt][nt][0], sc[rt][nt][1]), pk_bf16(sc[rt][nt][2], sc[rt][nt][3]), 0u, 0u};
                                o2[rt] = mfma_bf(vopb[nt >> 1], pb_, o2[rt]);
                            }
                            return;
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            if (v & 1) o2[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vop[nt][v], sc[rt][nt][v], o2[rt], 0, 0, 0);
                            else o[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vop[nt][v], sc[rt][nt][v], o[rt], 0, 0, 0);
                        }
                    };
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) { s_tile(0, nt); if (nt < 3) lh_score(nt); }
                    mx[0] = quarters_max(mx[0]);
                    cm[0] = -mx[0] * cs;
                    lmx = quarters_max(lmx);
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) { s_tile(1, nt); e_tile(0, nt); if (nt < 3) lh_exp(nt); }
                    mx[1] = quarters_max(mx[1]);
                    cm[1] = -mx[1] * cs;
                    lden = quarters_sum(lden);
                    lrden = lden > 0.f ? 1.0f / lden : 0.f;
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) { o_tile(0, nt); e_tile(1, nt); if (nt < 3) lh_norm(nt); }
#pragma unroll
                    for (int k = 0; k < 3; ++k) lF[k] = quarters_sum(lF[k]);
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) { o_tile(1, nt); if (nt < 3) lh_mfma(nt); }
                    // rows 4 hi + v of the 16-channel tile lh_dt: channels 8 (h & 1) .. + 7 belong to head lh_h
                    if (lh_on && ((hi_t >= 2) == bool(lh_h & 1))) {
                        float xo[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float4 lav = *reinterpret_cast<const float4*>(sT + CL_LAV + 4 * (16 * lh_dt + 4 * hi_t + v));
                            xo[v] = fmaf(lav.z, lF[2], fmaf(lav.y, lF[1], fmaf(lav.x, lF[0], lP[v])));
                        }
                        *reinterpret_cast<float4*>(sO1 + (wave >> 2) * 512 + (lh_dt * 64 + 16 * hi_t + lo_t) * 4) = make_float4(xo[0], xo[1], xo[2], xo[3]);
                    }
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const int traj = 16 * rt + lo_t;
                        const float dn = quarters_sum(den[rt]);
                        const bool live = dn > 0.f;                    // a decoding trajectory has an open node
                        const size_t r = (size_t)t * A.M + g_lo + traj;
                        if (TRAIN && (LEAN || A.trLse) && hi_t == 0 && live)
                            A.trLse[((size_t)b * Rcap + r) * ELG_H + wave] = __log2f(dn) - cm[rt];
                        const float inv = live ? 1.0f / dn : 0.f;
                        if (TRAIN && live && !LEAN && A.trA) {
#pragma unroll
                            for (int nt = 0; nt < CO_NT; ++nt) {
                                float* rA = A.trA + (((size_t)b * ELG_H + wave) * Rcap + r) * N1 + 16 * nt + 4 * hi_t;
                                // (scalar dword stores: rows are only 4-byte aligned, and one unaligned 16-byte store per
                                // group measured 4 % slower for the whole launch)
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    if (16 * nt + 4 * hi_t + v < N1) rA[v] = sc[rt][nt][v] * inv;
                            }
                        }
                        const float4 ov = make_float4((o[rt][0] + o2[rt][0]) * inv, (o[rt][1] + o2[rt][1]) * inv,
                                                      (o[rt][2] + o2[rt][2]) * inv, (o[rt][3] + o2[rt][3]) * inv);
                        // O^T[d = 4 hi_t + v][traj = lo_t] -> this head's 16 channels of the trajectory's exchange row
                        *reinterpret_cast<float4*>(sQ + traj * CO_QP + 16 * wave + 4 * hi_t) = ov;
                        if (TRAIN && live && !(ELG_EXP_SKIP & 4)) *reinterpret_cast<float4*>(A.trO + ((size_t)b * Rcap + r) * ELG_E + 16 * wave + 4 * hi_t) = ov;
                    }
                }
                ELG_STAMP(sc, 0);
                __syncthreads();
                ELG_STAMP(sc, 1);
                // =============== pointer (waves 0-5: 7 node tiles x 2 trajectory tiles) || local policy (waves 6, 7) ====
                if (wave < 6) {
#pragma unroll 1
                    for (int un = wave; un < (two_rt ? 2 * CO_NT : CO_NT); un += 6) {
                        const int nt = un % CO_NT, rt = un / CO_NT;
                        const int traj = 16 * rt + lo_t;
                        if (BF) {
                            const float* orow8 = sQ + traj * CO_QP + 8 * hi_t;
                            const unsigned* popb = reinterpret_cast<const unsigned*>(sP) + (nt * 4 * 64 + lane_t) * 4;
                            const float4 pb4b = *reinterpret_cast<const float4*>(sPb + 16 * nt + 4 * hi_t);
                            f32x4c b0 = {pb4b.x, pb4b.y, pb4b.z, pb4b.w}, b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const uint4 pk8 = *reinterpret_cast<const uint4*>(popb + g * 256);
                                const float4 oa = *reinterpret_cast<const float4*>(orow8 + 32 * g);
                                const float4 oc = *reinterpret_cast<const float4*>(orow8 + 32 * g + 4);
                                const u32x4 ob_ = {pk_bf16(oa.x, oa.y), pk_bf16(oa.z, oa.w), pk_bf16(oc.x, oc.y), pk_bf16(oc.z, oc.w)};
                                if (g & 1) b1 = mfma_bf(u32x4{pk8.x, pk8.y, pk8.z, pk8.w}, ob_, b1);
                                else b0 = mfma_bf(u32x4{pk8.x, pk8.y, pk8.z, pk8.w}, ob_, b0);
                            }
                            *reinterpret_cast<float4*>(sSc + traj * CO_SP + 16 * nt + 4 * hi_t) =
                                make_float4(b0[0] + b1[0], b0[1] + b1[1], b0[2] + b1[2], b0[3] + b1[3]);
                            continue;
                        }
                        const float* orow = sQ + traj * CO_QP + 4 * hi_t;
                        const float* pop = sP + (nt * 8 * 64 + lane_t) * 4;
                        const float4 pb4 = *reinterpret_cast<const float4*>(sPb + 16 * nt + 4 * hi_t);
                        f32x4c a0 = {pb4.x, pb4.y, pb4.z, pb4.w}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            const float4 pk4 = *reinterpret_cast<const float4*>(pop + g * 256);
                            const float4 ov = *reinterpret_cast<const float4*>(orow + 16 * g);
                            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.x, ov.x, a0, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.y, ov.y, a1, 0, 0, 0);
                            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.z, ov.z, a0, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.w, ov.w, a1, 0, 0, 0);
                        }
                        *reinterpret_cast<float4*>(sSc + traj * CO_SP + 16 * nt + 4 * hi_t) =
                            make_float4(a0[0] + a1[0], a0[1] + a1[1], a0[2] + a1[2], a0[3] + a1[3]);
                    }
                } else if ((LEAN || A.has_local) && (wave == 6 || two_rt)) {
                    // local policy, stage 2: wave 6: trajectories 0-15, wave 7: trajectories 16-31
                    co_local_tail(sT, sX + (wave - 6) * 16 * CO_XP, sO1 + (wave - 6) * 512, sX + (wave - 6) * 16 * CO_XP + CO_XU, CO_XP,
                                  lo_t, hi_t);
                }
                ELG_STAMP(sc, 2);
                __syncthreads();
                ELG_STAMP(sc, 3);
            }
            // =============== owners: finish this step, advance, prepare the next ===============
            int any_left = 0;
            // the wave's four trajectories side by side, 16 lanes each: choice, transition, next step's inputs
            const int q4 = 4 * wave + (lane_t >> 4);
            const int m4 = g_lo + min(q4, ntraj - 1);
            const size_t bm4 = (size_t)b * A.M + m4;
            const bool active = q4 < ntraj && !row.fin;
            int sel = 0;
            float pr = 1.0f;
            // (every wave runs the phase, also one without trajectories: it re-closes its rows' additive masks, which the
            // pointer phase has overwritten with scores)
            {
                if (decode_step) {
                    co_finish4<TSP, TRAIN, LEAN>(A, N1, lane_t, wave, ntraj, t, g_lo, (size_t)b, Rcap, sSc, sMask, sX, sState,
                                           q4 < ntraj ? row.fin : 1, sel, pr, ubuf, sc);
                } else if (!LEAN && A.mode == ELG_MODE_FORCED) {
                    sel = (A.forced && t < A.Tforced) ? A.forced[bm4 * A.Tforced + t] : 0;
                } else {
                    sel = (!TSP && t == 0) ? 0 : A.starts[m4];
                }
                if (active && (lane_t & 15) == 0) {
                    if (LEAN || A.actions) A.actions[bm4 * A.Tmax + t] = sel;
                    if (LEAN || A.probs) A.probs[((size_t)b * A.Tmax + t) * A.M + m4] = pr;
                }
                co_advance4<TSP, TRAIN, LEAN>(A, I, N1, lane_t, wave, ntraj, t, g_lo, (size_t)b, Rcap, row, sel, active, sMask, sQ, sX, sSc, sc);
            }
            any_left = (q4 < ntraj && !row.fin) ? 1 : 0;
            ELG_STAMP(sc, 11);
            if (BF) { ELG_CO_LOAD_KV_BF() } else { ELG_CO_LOAD_KV() }  // next step's glimpse operands, in flight over the barrier
            ELG_STAMP(sc, 12);
            const int go_on = __syncthreads_or(any_left);            // also orders the exchange rows for the next step
            ELG_STAMP(sc, 13);
            if (!go_on) break;
        }
#ifdef ELG_STAMPS
        if (A.scratch && lane == 0)
            for (int i = 0; i < 16; ++i) A.scratch[((size_t)blockIdx.x * 8 + wave) * 16 + i] = sc.acc[i];
#endif
        // ---- results of the group
        {
            const int q4 = 4 * wave + (lane >> 4);
            if (q4 < ntraj && (lane & 15) == 0) {
                const size_t bm = (size_t)b * A.M + g_lo + q4;
                if (A.reward) A.reward[bm] = -row.len;
                if (A.tlen) A.tlen[bm] = row.cnt;
            }
        }
        __syncthreads();
    }
}

template <bool TSP, bool TRAIN, bool BF, bool LEAN>
static int launch_fwd_coop_l(const elg_rollout_args& A, hipStream_t stream) {
    const size_t lds = ((size_t)CO_MAXTR * CO_QP + (size_t)CO_MAXTR * CO_SP + 4 * CO_MAXTR + 16 * CO_MAXTR +
                        ((A.N1 + 3) & ~3) + ((2 * A.N1 + 3) & ~3) + (size_t)CO_MAXTR * CO_XP + CL_SIZE + CO_NT * 32 * 64 + 16 * CO_NT +
                        8 * SbSize<2>::value + 2 * 512) * 4 + 64;
    auto kern = rollout_fwd_coop_kernel<TSP, TRAIN, BF, LEAN>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, "coop rollout: hipFuncSetAttribute failed");
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(A.B * A.tiles), dim3(512), lds, stream, A);
    return launch_status("rollout_fwd_coop");
}
template <bool TSP, bool TRAIN, bool BF>
static int launch_fwd_coop(const elg_rollout_args& A, hipStream_t stream) {
    // LEAN = the production configuration (reference config.yml: ensemble + distance penalty on, polar features), all outputs
    // requested, mask-row training rows: every test / ablation branch of the step loop is compiled out
    const bool lean = A.mode != ELG_MODE_FORCED && !A.forced && !A.uniforms && !A.full_probs && A.has_local && A.has_penalty &&
                      !A.euclidean && A.actions && A.probs &&
                      (!TRAIN || (!A.trA && A.trMask && A.trLse && A.trSlot && A.trF && (TSP || A.trLoad)));
    return lean ? launch_fwd_coop_l<TSP, TRAIN, BF, true>(A, stream) : launch_fwd_coop_l<TSP, TRAIN, BF, false>(A, stream);
}

// =============================================================================================
// Split-group cooperative kernel (round 6).  Same arithmetic, same LDS exchange rows and the same owners' phases as
// rollout_fwd_coop_kernel -- every product is formed by the same instruction sequence on the same operands, so tours, probabilities
// and saved rows are bit-identical -- but the workgroup's eight waves are TWO independent groups of four (group = wave >> 2), each
// advancing its own tile of <= 16 trajectories through glimpse -> pointer || local tail -> owners with barriers of its own (an LDS
// arrival counter per group; s_barrier would couple the groups).  Waves w and w + 4 share a SIMD: with the groups half a step apart
// the SIMD holds one wave in a matrix phase (MFMA pipe) and one in the owners' phase (vector issue, LDS, L2 gathers) instead of two
// waves that want the same pipe at the same time.  In a group, wave wg carries heads wg and wg + 4 of its tile (the lockstep kernel:
// one head, two tiles -- the same 124 MFMAs per wave and step): K of both heads is requested at the end of the owners' phase, V of a
// head while the other head's scores are formed, so at most two 28-register operand images are live, as before.
// =============================================================================================
__device__ __forceinline__ void grp_barrier(unsigned* cnt, unsigned& target, int lane) {
    // DS operations of one wave execute in issue order: the arrival add is behind the wave's LDS stores, a load issued after the
    // poll has matched is behind every other wave's stores that preceded its arrival.  The fences stop the compiler.
    target += 4u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
        const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(v - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool TSP, bool TRAIN, bool BF, bool LEAN>
__global__ __launch_bounds__(512) void rollout_fwd_coop2_kernel(const elg_rollout_args A, const int stagger) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;
    const int lo = lane & 15, hi = lane >> 4;
    const int N1 = A.N1;
    const int G = gridDim.x;
    int u = blockIdx.x;
    if ((G & 7) == 0) u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int b = u / A.tiles, tile = u % A.tiles;
    const int tile_m = (A.M + A.tiles - 1) / A.tiles;
    const int m_lo = tile * tile_m, m_hi = min(A.M, m_lo + tile_m);
    const size_t NE = (size_t)N1 * ELG_E;
    const size_t Rcap = (size_t)A.Tmax * A.M;

    // ---- LDS: as rollout_fwd_coop_kernel (rows 0-15 of every exchange array belong to group 0, rows 16-31 to group 1) + sync words
    float* sQ = lds;
    float* sSc = sQ + CO_MAXTR * CO_QP;
    unsigned long long* sMask = reinterpret_cast<unsigned long long*>(sSc + CO_MAXTR * CO_SP);
    int* sState = reinterpret_cast<int*>(sMask + 2 * CO_MAXTR);
    float* sdem = reinterpret_cast<float*>(sState + 16 * CO_MAXTR);
    float* sxy = sdem + ((N1 + 3) & ~3);
    float* sX = sxy + ((2 * N1 + 3) & ~3);
    float* sT = sX + CO_MAXTR * CO_XP;
    float* sP = sT + CL_SIZE;
    float* sPb = sP + CO_NT * 32 * 64;
    float* sO1 = sPb + 16 * CO_NT;                                     // [2 groups][2][64][4]
    unsigned* sSync = reinterpret_cast<unsigned*>(sO1 + 2 * 512);      // [2 groups][16]: arrival counter | 2 x 4 "any left" words
    if (!TSP)
        for (int i = tid; i < N1; i += 512) sdem[i] = A.demand[(size_t)b * N1 + i];
    for (int i = tid; i < 2 * N1; i += 512) sxy[i] = A.xy[(size_t)b * N1 * 2 + i];
    if (LEAN || A.has_local) co_stage_local(A.loc, sT, tid, 512);
    if (tid < 32) sSync[tid] = 0u;

    Inst I;
    I.K = nullptr; I.V = nullptr; I.PK = nullptr;
    I.pb = A.pb + (size_t)b * N1;
    I.Q1 = A.Q1 + b * NE;
    I.Q2 = TSP ? A.Q2 + b * NE : nullptr;
    I.wl = A.wl;
    I.xy = sxy;
    I.dem = sdem;
    I.nidx = A.nbr_idx + (size_t)b * N1 * N1;
    I.ndist = A.nbr_dist + (size_t)b * N1 * N1;
    I.ntheta = A.nbr_theta + (size_t)b * N1 * N1;
    I.loc = A.loc;

    const int h0 = wg, h1 = wg + 4;                                    // the wave's two glimpse heads
    const float* gK = A.Kmat + b * NE;
    const float* gV = A.Vmat + b * NE;
    // operand images as in the lockstep kernel: K: one 16-byte load per node tile (channels 4 hi .. + 3 of the head), V: four dwords
#define ELG_C2_LOAD_K(dst, H)                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < CO_NT; ++nt) {                                                  \
        const float4 k4_ = ld_off<float4>(gK, 4u * (unsigned)(min(16 * nt + lo_t, N1 - 1) * ELG_E + 16 * (H) + 4 * hi_t)); \
        dst[nt][0] = k4_.x; dst[nt][1] = k4_.y; dst[nt][2] = k4_.z; dst[nt][3] = k4_.w;                     \
    }
#define ELG_C2_LOAD_V(dst, H)                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < CO_NT; ++nt)                                                    \
        _Pragma("unroll") for (int v = 0; v < 4; ++v)                                                       \
            dst[nt][v] = ld_off<float>(gV, 4u * (unsigned)(min(16 * nt + 4 * hi_t + v, N1 - 1) * ELG_E + 16 * (H) + lo_t));
#define ELG_C2_LOAD_K_BF(dst, H)                                                                            \
    _Pragma("unroll") for (int nt = 0; nt < CO_NT; ++nt) {                                                  \
        const float4 k4_ = ld_off<float4>(gK, 4u * (unsigned)(min(16 * nt + lo_t, N1 - 1) * ELG_E + 16 * (H) + 4 * hi_t)); \
        dst[nt] = u32x4{pk_bf16(k4_.x, k4_.y), pk_bf16(k4_.z, k4_.w), 0u, 0u};                              \
    }
#define ELG_C2_LOAD_V_BF(dst, H)                                                                            \
    _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                                         \
        float va_[4], vb_[4];                                                                               \
        _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                                     \
            va_[v] = ld_off<float>(gV, 4u * (unsigned)(min(32 * p + 4 * hi_t + v, N1 - 1) * ELG_E + 16 * (H) + lo_t));      \
            vb_[v] = ld_off<float>(gV, 4u * (unsigned)(min(32 * p + 16 + 4 * hi_t + v, N1 - 1) * ELG_E + 16 * (H) + lo_t)); \
        }                                                                                                   \
        dst[p] = u32x4{pk_bf16(va_[0], va_[1]), pk_bf16(va_[2], va_[3]), pk_bf16(vb_[0], vb_[1]), pk_bf16(vb_[2], vb_[3])}; \
    }
    {
        // PK operand image in LDS, shared by both groups (rollout_fwd_coop_kernel's layout)
        const int np = 16 * wave + lo;
        const float* gP = A.PK + b * NE + (size_t)min(np, N1 - 1) * ELG_E;
        if (!BF && wave < CO_NT)
            for (int g = 0; g < 8; ++g) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (np < N1) x = *reinterpret_cast<const float4*>(gP + 16 * g + 4 * hi);
                *reinterpret_cast<float4*>(sP + ((wave * 8 + g) * 64 + lane) * 4) = x;
            }
        if (BF && wave < CO_NT)
            for (int g = 0; g < 4; ++g) {
                uint4 x = make_uint4(0u, 0u, 0u, 0u);
                if (np < N1) {
                    const float4 a = *reinterpret_cast<const float4*>(gP + 32 * g + 8 * hi);
                    const float4 c = *reinterpret_cast<const float4*>(gP + 32 * g + 8 * hi + 4);
                    x = make_uint4(pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(c.x, c.y), pk_bf16(c.z, c.w));
                }
                *reinterpret_cast<uint4*>(sP + ((wave * 4 + g) * 64 + lane) * 4) = x;
            }
    }
    for (int i = tid; i < 16 * CO_NT; i += 512) sPb[i] = (i < N1) ? I.pb[i] : 0.f;
    __syncthreads();                                                   // the only workgroup barrier of the launch

    unsigned* gcnt = sSync + 16 * grp;
    unsigned* gany = sSync + 16 * grp + 4;                             // [2 parities][4 waves]
    unsigned bar_target = 0u, or_epoch = 0u;
    // group 1 starts half a step behind (its partner waves then sit in the other kind of phase; nothing depends on it)
    if (grp == 1)
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(64);

    // tiles of <= 16 trajectories, evenly sized; tile i belongs to group i & 1
    const int n_my = m_hi - m_lo;
    const int n_tiles = (n_my + 15) / 16;
    const int t_size = n_tiles > 0 ? (n_my + n_tiles - 1) / n_tiles : 16;
    for (int ti = grp; ti < n_tiles; ti += 2) {
        const int g_lo_real = m_lo + ti * t_size;
        const int nreal = min(t_size, m_hi - g_lo_real);
        // the owners' code addresses slot q = 4 wave + (lane >> 4) of the 32-row exchange arrays and trajectory g_lo + q: for
        // group 1 (slots 16-31) both bounds are shifted by 16
        const int ntraj = 16 * grp + nreal;
        const int g_lo = g_lo_real - 16 * grp;
        // ---- reset of the group's 16 slots
        for (int q = 16 * grp + wg; q < 16 * grp + 16; q += 4) {
            Traj<2> st;
            st.cur = 0; st.first = 0; st.cnt = 0; st.fin = (q < ntraj) ? 0 : 1; st.load = 1.0f; st.len = 0.f;
            st.cx = 0.f; st.cy = 0.f; st.vis[0] = 0ull; st.vis[1] = 0ull;
            co_store_state<TSP>(sState + 16 * q, st, lane);
            if (lane == 0) { sMask[2 * q] = ~0ull; sMask[2 * q + 1] = ~0ull; }
            if (lane < 33) *reinterpret_cast<float4*>(sQ + q * CO_QP + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = lane; i < CO_XP; i += 64) sX[q * CO_XP + i] = (i >= CO_XS && i < CO_XPEN) ? i2f(-1) : 0.f;
        }
        grp_barrier(gcnt, bar_target, lane);
        const int step_cap = TSP ? N1 : 2 * N1 + 2;
        CoRow row;
        float ubuf = 0.f;
        row.cur = 0; row.first = 0; row.cnt = 0; row.fin = (4 * wave + (lane >> 4) < ntraj) ? 0 : 1;
        row.load = 1.0f; row.len = 0.f; row.cx = 0.f; row.cy = 0.f; row.v0 = 0ull; row.v1 = 0ull;
        StampCtx sc;
#ifdef ELG_STAMPS
        for (int i = 0; i < 16; ++i) sc.acc[i] = 0.f;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc.last) :: "memory");
#endif
        float kA[CO_NT][4], kB[CO_NT][4];                           // K images of heads h0 / h1, (re)loaded at the end of the owners' phase
        u32x4 kAb[CO_NT], kBb[CO_NT];
#pragma unroll
        for (int nt = 0; nt < CO_NT; ++nt) {
#pragma unroll
            for (int v = 0; v < 4; ++v) { kA[nt][v] = 0.f; kB[nt][v] = 0.f; }
            kAb[nt] = u32x4{0u, 0u, 0u, 0u}; kBb[nt] = u32x4{0u, 0u, 0u, 0u};
        }
        for (int t = 0; t < step_cap && t < A.Tmax; ++t) {
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            const int lo_t = lane_t & 15, hi_t = lane_t >> 4;
            const bool decode_step = TSP ? (t >= 1) : (t >= 2);     // uniform over the group: lockstep
            ELG_STAMP(sc, 15);
            if (decode_step) {
                // =============== glimpse: wave wg = heads wg, wg + 4 of the group's tile; local head unit (group, wg) ===============
                const bool lh_on = LEAN || A.has_local;
                const int lh_h = wg, lh_dt = lh_h >> 1;
                const float* LX = sX + (16 * grp + lo_t) * CO_XP;
                f32x4c lf[3][3], lal[3];
                bool lmsk[3][4];
                float lmx = ELG_NEG_INF, lden = 0.f, lF[3] = {0.f, 0.f, 0.f};
                f32x4c lP = {0.f, 0.f, 0.f, 0.f};
                const float4 la4 = *reinterpret_cast<const float4*>(sT + CL_LA + 4 * lh_h);
                auto lh_score = [&](int jt) {
                    const int4 sl = *reinterpret_cast<const int4*>(LX + CO_XS + 16 * jt + 4 * hi_t);
                    lmsk[jt][0] = sl.x < 0; lmsk[jt][1] = sl.y < 0; lmsk[jt][2] = sl.z < 0; lmsk[jt][3] = sl.w < 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float4 tq = *reinterpret_cast<const float4*>(LX + CO_XF + k * ELG_SLOT_STRIDE + 16 * jt + 4 * hi_t);
                        lf[k][jt] = f32x4c{tq.x, tq.y, tq.z, tq.w};
                    }
                    const float4 lt4 = *reinterpret_cast<const float4*>(sT + CL_LTT + lh_h * 48 + 16 * jt + 4 * hi_t);
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
                };
                auto lh_exp = [&](int jt) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float e = lmsk[jt][v] ? 0.f : __expf(lal[jt][v] - lmx);
                        lal[jt][v] = e;
                        lden += e;
                    }
                };
                float lrden = 0.f;
                auto lh_norm = [&](int jt) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float a = lal[jt][v] * lrden;
                        lal[jt][v] = a;
#pragma unroll
                        for (int k = 0; k < 3; ++k) lF[k] = fmaf(a, lf[k][jt][v], lF[k]);
                    }
                };
                auto lh_mfma = [&](int jt) {
                    const float4 a4 = *reinterpret_cast<const float4*>(sT + CL_LCVT + (16 * lh_dt + lo_t) * CL_Q + 16 * jt + 4 * hi_t);
                    lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, lal[jt][0], lP, 0, 0, 0);
                    lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, lal[jt][1], lP, 0, 0, 0);
                    lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, lal[jt][2], lP, 0, 0, 0);
                    lP = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, lal[jt][3], lP, 0, 0, 0);
                };
                // The wave's two heads are in flight together (program order: S(h0) | S(h1) with exp(h0) | O(h0) with exp(h1) | O(h1)),
                // exactly the lockstep kernel's pipeline over its two trajectory tiles.
                {
                    const float cs = 0.25f * 1.4426950408889634f;
                    const int traj = 16 * grp + lo_t;
                    float qb[2][4];
                    u32x4 qbb[2];
                    f32x4c scr[2][CO_NT];
                    float mx[2], cm[2], den[2] = {0.f, 0.f};
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) {
                        const float4 q4 = *reinterpret_cast<const float4*>(sQ + traj * CO_QP + 16 * (wg + 4 * hp) + 4 * hi_t);
                        qb[hp][0] = q4.x; qb[hp][1] = q4.y; qb[hp][2] = q4.z; qb[hp][3] = q4.w;
                        qbb[hp] = u32x4{pk_bf16(q4.x, q4.y), pk_bf16(q4.z, q4.w), 0u, 0u};
                        mx[hp] = -1e30f;
                    }
                    float vA[CO_NT][4], vB[CO_NT][4];
                    u32x4 vAb[4], vBb[4];
                    auto s_tile = [&](int hp, int nt) {
                        const float4 m4 = *reinterpret_cast<const float4*>(sSc + traj * CO_SP + 16 * nt + 4 * hi_t);
                        f32x4c acc = {m4.x, m4.y, m4.z, m4.w};
                        if (BF) acc = mfma_bf(hp ? kBb[nt] : kAb[nt], qbb[hp], acc);
                        else {
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk)
                                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hp ? kB[nt][kk] : kA[nt][kk], qb[hp][kk], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v) mx[hp] = fmaxf(mx[hp], acc[v]);
                        scr[hp][nt] = acc;
                    };
                    auto e_tile = [&](int hp, int nt) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(scr[hp][nt][v], cs, cm[hp]));
                            scr[hp][nt][v] = e;
                            den[hp] += e;
                        }
                    };
                    f32x4c o[2], o2[2];
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) { o[hp] = f32x4c{0.f, 0.f, 0.f, 0.f}; o2[hp] = f32x4c{0.f, 0.f, 0.f, 0.f}; }
                    auto o_tile = [&](int hp, int nt) {
                        if (BF) {
                            if (nt & 1) {
                                const u32x4 pb_ = {pk_bf16(scr[hp][nt - 1][0], scr[hp][nt - 1][1]), pk_bf16(scr[hp][nt - 1][2], scr[hp][nt - 1][3]),
                                                   pk_bf16(scr[hp][nt][0], scr[hp][nt][1]), pk_bf16(scr[hp][nt][2], scr[hp][nt][3])};
                                if (nt & 2) o2[hp] = mfma_bf(hp ? vBb[nt >> 1] : vAb[nt >> 1], pb_, o2[hp]);
                                else o[hp] = mfma_bf(hp ? vBb[nt >> 1] : vAb[nt >> 1], pb_, o[hp]);
                            } else if (nt == CO_NT - 1) {
                                const u32x4 pb_ = {pk_bf16(scr[hp][nt][0], scr[hp][nt][1]), pk_bf16(scr[hp][nt][2], scr[hp][nt][3]), 0u, 0u};
                                o2[hp] = mfma_bf(hp ? vBb[nt >> 1] : vAb[nt >> 1], pb_, o2[hp]);
                            }
                            return;
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float vv = hp ? vB[nt][v] : vA[nt][v];
                            if (v & 1) o2[hp] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, scr[hp][nt][v], o2[hp], 0, 0, 0);
                            else o[hp] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, scr[hp][nt][v], o[hp], 0, 0, 0);
                        }
                    };
                    if (BF) { ELG_C2_LOAD_V_BF(vAb, h0) } else { ELG_C2_LOAD_V(vA, h0) }
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) { s_tile(0, nt); if (nt < 3) lh_score(nt); }
                    mx[0] = quarters_max(mx[0]);
                    cm[0] = -mx[0] * cs;
                    lmx = quarters_max(lmx);
                    if (BF) { ELG_C2_LOAD_V_BF(vBb, h1) } else { ELG_C2_LOAD_V(vB, h1) }
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) { s_tile(1, nt); e_tile(0, nt); if (nt < 3) lh_exp(nt); }
                    mx[1] = quarters_max(mx[1]);
                    cm[1] = -mx[1] * cs;
                    lden = quarters_sum(lden);
                    lrden = lden > 0.f ? 1.0f / lden : 0.f;
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) { o_tile(0, nt); e_tile(1, nt); if (nt < 3) lh_norm(nt); }
#pragma unroll
                    for (int k = 0; k < 3; ++k) lF[k] = quarters_sum(lF[k]);
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) { o_tile(1, nt); if (nt < 3) lh_mfma(nt); }
                    if (lh_on && ((hi_t >= 2) == bool(lh_h & 1))) {
                        float xo[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float4 lav = *reinterpret_cast<const float4*>(sT + CL_LAV + 4 * (16 * lh_dt + 4 * hi_t + v));
                            xo[v] = fmaf(lav.z, lF[2], fmaf(lav.y, lF[1], fmaf(lav.x, lF[0], lP[v])));
                        }
                        *reinterpret_cast<float4*>(sO1 + grp * 512 + (lh_dt * 64 + 16 * hi_t + lo_t) * 4) = make_float4(xo[0], xo[1], xo[2], xo[3]);
                    }
                    const size_t r = (size_t)t * A.M + g_lo + traj;
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) {
                        const int head = wg + 4 * hp;
                        const float dn = quarters_sum(den[hp]);
                        const bool live = dn > 0.f;
                        if (TRAIN && (LEAN || A.trLse) && hi_t == 0 && live)
                            A.trLse[((size_t)b * Rcap + r) * ELG_H + head] = __log2f(dn) - cm[hp];
                        const float inv = live ? 1.0f / dn : 0.f;
                        if (TRAIN && live && !LEAN && A.trA) {
#pragma unroll
                            for (int nt = 0; nt < CO_NT; ++nt) {
                                float* rA = A.trA + (((size_t)b * ELG_H + head) * Rcap + r) * N1 + 16 * nt + 4 * hi_t;
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    if (16 * nt + 4 * hi_t + v < N1) rA[v] = scr[hp][nt][v] * inv;
                            }
                        }
                        const float4 ov = make_float4((o[hp][0] + o2[hp][0]) * inv, (o[hp][1] + o2[hp][1]) * inv,
                                                      (o[hp][2] + o2[hp][2]) * inv, (o[hp][3] + o2[hp][3]) * inv);
                        *reinterpret_cast<float4*>(sQ + traj * CO_QP + 16 * head + 4 * hi_t) = ov;
                        if (TRAIN && live && !(ELG_EXP_SKIP & 4)) *reinterpret_cast<float4*>(A.trO + ((size_t)b * Rcap + r) * ELG_E + 16 * head + 4 * hi_t) = ov;
                    }
                }
                ELG_STAMP(sc, 0);
                grp_barrier(gcnt, bar_target, lane);
                ELG_STAMP(sc, 1);
                // =============== pointer (waves 0-2 of the group: node tiles 0,3,6 | 1,4 | 2,5) || local tail (wave 3) ===============
                if (wg < 3) {
#pragma unroll 1
                    for (int nt = wg; nt < CO_NT; nt += 3) {
                        const int traj = 16 * grp + lo_t;
                        if (BF) {
                            const float* orow8 = sQ + traj * CO_QP + 8 * hi_t;
                            const unsigned* popb = reinterpret_cast<const unsigned*>(sP) + (nt * 4 * 64 + lane_t) * 4;
                            const float4 pb4b = *reinterpret_cast<const float4*>(sPb + 16 * nt + 4 * hi_t);
                            f32x4c b0 = {pb4b.x, pb4b.y, pb4b.z, pb4b.w}, b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const uint4 pk8 = *reinterpret_cast<const uint4*>(popb + g * 256);
                                const float4 oa = *reinterpret_cast<const float4*>(orow8 + 32 * g);
                                const float4 oc = *reinterpret_cast<const float4*>(orow8 + 32 * g + 4);
                                const u32x4 ob_ = {pk_bf16(oa.x, oa.y), pk_bf16(oa.z, oa.w), pk_bf16(oc.x, oc.y), pk_bf16(oc.z, oc.w)};
                                if (g & 1) b1 = mfma_bf(u32x4{pk8.x, pk8.y, pk8.z, pk8.w}, ob_, b1);
                                else b0 = mfma_bf(u32x4{pk8.x, pk8.y, pk8.z, pk8.w}, ob_, b0);
                            }
                            *reinterpret_cast<float4*>(sSc + traj * CO_SP + 16 * nt + 4 * hi_t) =
                                make_float4(b0[0] + b1[0], b0[1] + b1[1], b0[2] + b1[2], b0[3] + b1[3]);
                            continue;
                        }
                        const float* orow = sQ + traj * CO_QP + 4 * hi_t;
                        const float* pop = sP + (nt * 8 * 64 + lane_t) * 4;
                        const float4 pb4 = *reinterpret_cast<const float4*>(sPb + 16 * nt + 4 * hi_t);
                        f32x4c a0 = {pb4.x, pb4.y, pb4.z, pb4.w}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            const float4 pk4 = *reinterpret_cast<const float4*>(pop + g * 256);
                            const float4 ov = *reinterpret_cast<const float4*>(orow + 16 * g);
                            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.x, ov.x, a0, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.y, ov.y, a1, 0, 0, 0);
                            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.z, ov.z, a0, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.w, ov.w, a1, 0, 0, 0);
                        }
                        *reinterpret_cast<float4*>(sSc + traj * CO_SP + 16 * nt + 4 * hi_t) =
                            make_float4(a0[0] + a1[0], a0[1] + a1[1], a0[2] + a1[2], a0[3] + a1[3]);
                    }
                } else if (LEAN || A.has_local) {
                    co_local_tail(sT, sX + grp * 16 * CO_XP, sO1 + grp * 512, sX + grp * 16 * CO_XP + CO_XU, CO_XP, lo_t, hi_t);
                }
                ELG_STAMP(sc, 2);
                grp_barrier(gcnt, bar_target, lane);
                ELG_STAMP(sc, 3);
            }
            // =============== owners: finish this step, advance, prepare the next (rollout_fwd_coop_kernel's, unchanged) ===============
            const int q4 = 4 * wave + (lane_t >> 4);
            const int m4 = g_lo + min(q4, ntraj - 1);
            const size_t bm4 = (size_t)b * A.M + m4;
            const bool active = q4 < ntraj && !row.fin;
            int sel = 0;
            float pr = 1.0f;
            if (decode_step) {
                co_finish4<TSP, TRAIN, LEAN>(A, N1, lane_t, wave, ntraj, t, g_lo, (size_t)b, Rcap, sSc, sMask, sX, sState,
                                       q4 < ntraj ? row.fin : 1, sel, pr, ubuf, sc);
            } else if (!LEAN && A.mode == ELG_MODE_FORCED) {
                sel = (A.forced && t < A.Tforced) ? A.forced[bm4 * A.Tforced + t] : 0;
            } else {
                sel = (!TSP && t == 0) ? 0 : A.starts[m4];
            }
            if (active && (lane_t & 15) == 0) {
                if (LEAN || A.actions) A.actions[bm4 * A.Tmax + t] = sel;
                if (LEAN || A.probs) A.probs[((size_t)b * A.Tmax + t) * A.M + m4] = pr;
            }
            co_advance4<TSP, TRAIN, LEAN>(A, I, N1, lane_t, wave, ntraj, t, g_lo, (size_t)b, Rcap, row, sel, active, sMask, sQ, sX, sSc, sc);
            const int any_left = (q4 < ntraj && !row.fin) ? 1 : 0;
            ELG_STAMP(sc, 11);
            if (BF) { ELG_C2_LOAD_K_BF(kAb, h0) ELG_C2_LOAD_K_BF(kBb, h1) } else { ELG_C2_LOAD_K(kA, h0) ELG_C2_LOAD_K(kB, h1) }
            ELG_STAMP(sc, 12);
            // group-wide OR of "a trajectory is left": every wave leaves its word (double-buffered by the epoch's parity), the
            // barrier orders them, everybody reads the four
            const unsigned wany = __ballot(any_left) != 0ull ? 1u : 0u;
            unsigned* slot = gany + 4 * (or_epoch & 1u);
            if (lane == 0) __hip_atomic_store(slot + wg, wany, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            grp_barrier(gcnt, bar_target, lane);
            const uint4 a4 = *reinterpret_cast<const uint4*>(slot);
            or_epoch += 1u;
            ELG_STAMP(sc, 13);
            if (!__builtin_amdgcn_readfirstlane((int)(a4.x | a4.y | a4.z | a4.w))) break;
        }
#ifdef ELG_STAMPS
        if (A.scratch && lane == 0)
            for (int i = 0; i < 16; ++i) A.scratch[((size_t)blockIdx.x * 8 + wave) * 16 + i] = sc.acc[i];
#endif
        {
            const int q4 = 4 * wave + (lane >> 4);
            if (q4 < ntraj && (lane & 15) == 0) {
                const size_t bm = (size_t)b * A.M + g_lo + q4;
                if (A.reward) A.reward[bm] = -row.len;
                if (A.tlen) A.tlen[bm] = row.cnt;
            }
        }
        grp_barrier(gcnt, bar_target, lane);
    }
}

template <bool TSP, bool TRAIN, bool BF, bool LEAN>
static int launch_fwd_coop2_l(const elg_rollout_args& A, hipStream_t stream, int stagger) {
    const size_t lds = ((size_t)CO_MAXTR * CO_QP + (size_t)CO_MAXTR * CO_SP + 4 * CO_MAXTR + 16 * CO_MAXTR +
                        ((A.N1 + 3) & ~3) + ((2 * A.N1 + 3) & ~3) + (size_t)CO_MAXTR * CO_XP + CL_SIZE + CO_NT * 32 * 64 + 16 * CO_NT +
                        2 * 512 + 32) * 4 + 64;
    auto kern = rollout_fwd_coop2_kernel<TSP, TRAIN, BF, LEAN>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, "coop rollout: hipFuncSetAttribute failed");
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(A.B * A.tiles), dim3(512), lds, stream, A, stagger);
    return launch_status("rollout_fwd_coop2");
}
template <bool TSP, bool TRAIN, bool BF>
static int launch_fwd_coop2(const elg_rollout_args& A, hipStream_t stream, int stagger) {
    const bool lean = A.mode != ELG_MODE_FORCED && !A.forced && !A.uniforms && !A.full_probs && A.has_local && A.has_penalty &&
                      !A.euclidean && A.actions && A.probs &&
                      (!TRAIN || (!A.trA && A.trMask && A.trLse && A.trSlot && A.trF && (TSP || A.trLoad)));
    return lean ? launch_fwd_coop2_l<TSP, TRAIN, BF, true>(A, stream, stagger) : launch_fwd_coop2_l<TSP, TRAIN, BF, false>(A, stream, stagger);
}

// =============================================================================================
// Split-group kernel at FOUR waves per SIMD (round 6, `variant = 5`): 16 waves per workgroup = two independent groups of EIGHT.
// The owners' phase needs 81 VGPRs, so once the matrix phases fit 128 registers the whole kernel runs at four waves per SIMD:
// in a group, wave wg = head wg for the group's ONE tile of <= 16 trajectories (62 f32 MFMAs per wave and step -- one
// 28-register K image, V streamed in while the scores form), waves 0-6 one node tile of the pointer each, wave 7 the local tail,
// waves 4-7 the four local head units (after their glimpse unit: registers reused), waves 0-3 the owners' phase of four
// trajectories each.  Group barriers through LDS arrival counters as in rollout_fwd_coop2_kernel; same arithmetic, same results.
// =============================================================================================
__device__ __forceinline__ void grp8_barrier(unsigned* cnt, unsigned& target, int lane) {
    target += 8u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
        const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(v - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <bool TSP, bool TRAIN, bool BF, bool LEAN>
__global__ __launch_bounds__(1024) void rollout_fwd_coop3_kernel(const elg_rollout_args A, const int stagger) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 3, wg = wave & 7;
    const int lo = lane & 15, hi = lane >> 4;
    const int N1 = A.N1;
    const int G = gridDim.x;
    int u = blockIdx.x;
    if ((G & 7) == 0) u = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int b = u / A.tiles, tile = u % A.tiles;
    const int tile_m = (A.M + A.tiles - 1) / A.tiles;
    const int m_lo = tile * tile_m, m_hi = min(A.M, m_lo + tile_m);
    const size_t NE = (size_t)N1 * ELG_E;
    const size_t Rcap = (size_t)A.Tmax * A.M;

    float* sQ = lds;
    float* sSc = sQ + CO_MAXTR * CO_QP;
    unsigned long long* sMask = reinterpret_cast<unsigned long long*>(sSc + CO_MAXTR * CO_SP);
    int* sState = reinterpret_cast<int*>(sMask + 2 * CO_MAXTR);
    float* sdem = reinterpret_cast<float*>(sState + 16 * CO_MAXTR);
    float* sxy = sdem + ((N1 + 3) & ~3);
    float* sX = sxy + ((2 * N1 + 3) & ~3);
    float* sT = sX + CO_MAXTR * CO_XP;
    float* sP = sT + CL_SIZE;
    float* sPb = sP + CO_NT * 32 * 64;
    float* sO1 = sPb + 16 * CO_NT;                                     // [2 groups][2][64][4]
    unsigned* sSync = reinterpret_cast<unsigned*>(sO1 + 2 * 512);      // [2 groups][32]: arrival counter | 2 x 8 "any left" words
    if (!TSP)
        for (int i = tid; i < N1; i += 1024) sdem[i] = A.demand[(size_t)b * N1 + i];
    for (int i = tid; i < 2 * N1; i += 1024) sxy[i] = A.xy[(size_t)b * N1 * 2 + i];
    if (LEAN || A.has_local) co_stage_local(A.loc, sT, tid, 1024);
    if (tid < 64) sSync[tid] = 0u;

    Inst I;
    I.K = nullptr; I.V = nullptr; I.PK = nullptr;
    I.pb = A.pb + (size_t)b * N1;
    I.Q1 = A.Q1 + b * NE;
    I.Q2 = TSP ? A.Q2 + b * NE : nullptr;
    I.wl = A.wl;
    I.xy = sxy;
    I.dem = sdem;
    I.nidx = A.nbr_idx + (size_t)b * N1 * N1;
    I.ndist = A.nbr_dist + (size_t)b * N1 * N1;
    I.ntheta = A.nbr_theta + (size_t)b * N1 * N1;
    I.loc = A.loc;

    const float* gK = A.Kmat + b * NE;
    const float* gV = A.Vmat + b * NE;
    {
        // PK operand image in LDS, shared by both groups (rollout_fwd_coop_kernel's layout); waves 0-6 of group 0 write it
        const int np = 16 * wave + lo;
        const float* gP = A.PK + b * NE + (size_t)min(np, N1 - 1) * ELG_E;
        if (!BF && wave < CO_NT)
            for (int g = 0; g < 8; ++g) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (np < N1) x = *reinterpret_cast<const float4*>(gP + 16 * g + 4 * hi);
                *reinterpret_cast<float4*>(sP + ((wave * 8 + g) * 64 + lane) * 4) = x;
            }
        if (BF && wave < CO_NT)
            for (int g = 0; g < 4; ++g) {
                uint4 x = make_uint4(0u, 0u, 0u, 0u);
                if (np < N1) {
                    const float4 a = *reinterpret_cast<const float4*>(gP + 32 * g + 8 * hi);
                    const float4 c = *reinterpret_cast<const float4*>(gP + 32 * g + 8 * hi + 4);
                    x = make_uint4(pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(c.x, c.y), pk_bf16(c.z, c.w));
                }
                *reinterpret_cast<uint4*>(sP + ((wave * 4 + g) * 64 + lane) * 4) = x;
            }
    }
    for (int i = tid; i < 16 * CO_NT; i += 1024) sPb[i] = (i < N1) ? I.pb[i] : 0.f;
    __syncthreads();                                                   // the only workgroup barrier of the launch

    unsigned* gcnt = sSync + 32 * grp;
    unsigned* gany = sSync + 32 * grp + 8;                             // [2 parities][8 waves]
    unsigned bar_target = 0u, or_epoch = 0u;
    if (grp == 1)
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(64);

    const int n_my = m_hi - m_lo;
    const int n_tiles = (n_my + 15) / 16;
    const int t_size = n_tiles > 0 ? (n_my + n_tiles - 1) / n_tiles : 16;
    const bool owner = wg < 4;                                          // waves 0-3 of a group own four trajectories each
    const int ow = 4 * grp + wg;                                        // the owners' "wave" index: slots 4 ow .. 4 ow + 3
    for (int ti = grp; ti < n_tiles; ti += 2) {
        const int g_lo_real = m_lo + ti * t_size;
        const int nreal = min(t_size, m_hi - g_lo_real);
        const int ntraj = 16 * grp + nreal;
        const int g_lo = g_lo_real - 16 * grp;
        for (int q = 16 * grp + wg; q < 16 * grp + 16; q += 8) {
            Traj<2> st;
            st.cur = 0; st.first = 0; st.cnt = 0; st.fin = (q < ntraj) ? 0 : 1; st.load = 1.0f; st.len = 0.f;
            st.cx = 0.f; st.cy = 0.f; st.vis[0] = 0ull; st.vis[1] = 0ull;
            co_store_state<TSP>(sState + 16 * q, st, lane);
            if (lane == 0) { sMask[2 * q] = ~0ull; sMask[2 * q + 1] = ~0ull; }
            if (lane < 33) *reinterpret_cast<float4*>(sQ + q * CO_QP + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i = lane; i < CO_XP; i += 64) sX[q * CO_XP + i] = (i >= CO_XS && i < CO_XPEN) ? i2f(-1) : 0.f;
        }
        grp8_barrier(gcnt, bar_target, lane);
        const int step_cap = TSP ? N1 : 2 * N1 + 2;
        CoRow row;
        float ubuf = 0.f;
        row.cur = 0; row.first = 0; row.cnt = 0; row.fin = (owner && 4 * ow + (lane >> 4) < ntraj) ? 0 : 1;
        row.load = 1.0f; row.len = 0.f; row.cx = 0.f; row.cy = 0.f; row.v0 = 0ull; row.v1 = 0ull;
        StampCtx sc;
#ifdef ELG_STAMPS
        for (int i = 0; i < 16; ++i) sc.acc[i] = 0.f;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sc.last) :: "memory");
#endif
        float kA[CO_NT][4];                                          // K image of head wg, (re)loaded at the end of the step
        u32x4 kAb[CO_NT];
#pragma unroll
        for (int nt = 0; nt < CO_NT; ++nt) {
#pragma unroll
            for (int v = 0; v < 4; ++v) kA[nt][v] = 0.f;
            kAb[nt] = u32x4{0u, 0u, 0u, 0u};
        }
        for (int t = 0; t < step_cap && t < A.Tmax; ++t) {
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            const int lo_t = lane_t & 15, hi_t = lane_t >> 4;
            const bool decode_step = TSP ? (t >= 1) : (t >= 2);
            ELG_STAMP(sc, 15);
            if (decode_step) {
                // =============== glimpse: wave wg = head wg of the group's tile ===============
                {
                    const float cs = 0.25f * 1.4426950408889634f;
                    const int traj = 16 * grp + lo_t;
                    const float4 q4 = *reinterpret_cast<const float4*>(sQ + traj * CO_QP + 16 * wg + 4 * hi_t);
                    const float qb[4] = {q4.x, q4.y, q4.z, q4.w};
                    const u32x4 qbb = {pk_bf16(q4.x, q4.y), pk_bf16(q4.z, q4.w), 0u, 0u};
                    float vA[CO_NT][4];
                    u32x4 vAb[4];
                    if (BF) { ELG_C2_LOAD_V_BF(vAb, wg) } else { ELG_C2_LOAD_V(vA, wg) }
                    f32x4c scr[CO_NT];
                    float mx = -1e30f, den = 0.f;
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) {
                        const float4 m4 = *reinterpret_cast<const float4*>(sSc + traj * CO_SP + 16 * nt + 4 * hi_t);
                        f32x4c acc = {m4.x, m4.y, m4.z, m4.w};
                        if (BF) acc = mfma_bf(kAb[nt], qbb, acc);
                        else {
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kA[nt][kk], qb[kk], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int v = 0; v < 4; ++v) mx = fmaxf(mx, acc[v]);
                        scr[nt] = acc;
                    }
                    mx = quarters_max(mx);
                    const float cm = -mx * cs;
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float e = __builtin_amdgcn_exp2f(fmaf(scr[nt][v], cs, cm));
                            scr[nt][v] = e;
                            den += e;
                        }
                    f32x4c o = {0.f, 0.f, 0.f, 0.f}, o2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int nt = 0; nt < CO_NT; ++nt) {
                        if (BF) {
                            if (nt & 1) {
                                const u32x4 pb_ = {pk_bf16(scr[nt - 1][0], scr[nt - 1][1]), pk_bf16(scr[nt - 1][2], scr[nt - 1][3]),
                                                   pk_bf16(scr[nt][0], scr[nt][1]), pk_bf16(scr[nt][2], scr[nt][3])};
                                if (nt & 2) o2 = mfma_bf(vAb[nt >> 1], pb_, o2);
                                else o = mfma_bf(vAb[nt >> 1], pb_, o);
                            } else if (nt == CO_NT - 1) {
                                const u32x4 pb_ = {pk_bf16(scr[nt][0], scr[nt][1]), pk_bf16(scr[nt][2], scr[nt][3]), 0u, 0u};
                                o2 = mfma_bf(vAb[nt >> 1], pb_, o2);
                            }
                        } else {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                if (v & 1) o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[nt][v], scr[nt][v], o2, 0, 0, 0);
                                else o = __builtin_amdgcn_mfma_f32_16x16x4f32(vA[nt][v], scr[nt][v], o, 0, 0, 0);
                            }
                        }
                    }
                    const size_t r = (size_t)t * A.M + g_lo + traj;
                    const float dn = quarters_sum(den);
                    const bool live = dn > 0.f;
                    if (TRAIN && (LEAN || A.trLse) && hi_t == 0 && live)
                        A.trLse[((size_t)b * Rcap + r) * ELG_H + wg] = __log2f(dn) - cm;
                    const float inv = live ? 1.0f / dn : 0.f;
                    if (TRAIN && live && !LEAN && A.trA) {
#pragma unroll
                        for (int nt = 0; nt < CO_NT; ++nt) {
                            float* rA = A.trA + (((size_t)b * ELG_H + wg) * Rcap + r) * N1 + 16 * nt + 4 * hi_t;
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                if (16 * nt + 4 * hi_t + v < N1) rA[v] = scr[nt][v] * inv;
                        }
                    }
                    const float4 ov = make_float4((o[0] + o2[0]) * inv, (o[1] + o2[1]) * inv, (o[2] + o2[2]) * inv, (o[3] + o2[3]) * inv);
                    *reinterpret_cast<float4*>(sQ + traj * CO_QP + 16 * wg + 4 * hi_t) = ov;
                    if (TRAIN && live && !(ELG_EXP_SKIP & 4)) *reinterpret_cast<float4*>(A.trO + ((size_t)b * Rcap + r) * ELG_E + 16 * wg + 4 * hi_t) = ov;
                }
                // local policy, stage 1: the group's four head units on waves 4-7 (which carry no trajectories), after their glimpse unit
                if ((LEAN || A.has_local) && wg >= 4) co_local_head_call(sT, sX + grp * 16 * CO_XP, sO1 + grp * 512, wg - 4, lo_t, hi_t);
                ELG_STAMP(sc, 0);
                grp8_barrier(gcnt, bar_target, lane);
                ELG_STAMP(sc, 1);
                // =============== pointer (waves 0-6: node tile wg) || local tail (wave 7) ===============
                if (wg < CO_NT) {
                    const int nt = wg;
                    const int traj = 16 * grp + lo_t;
                    if (BF) {
                        const float* orow8 = sQ + traj * CO_QP + 8 * hi_t;
                        const unsigned* popb = reinterpret_cast<const unsigned*>(sP) + (nt * 4 * 64 + lane_t) * 4;
                        const float4 pb4b = *reinterpret_cast<const float4*>(sPb + 16 * nt + 4 * hi_t);
                        f32x4c b0 = {pb4b.x, pb4b.y, pb4b.z, pb4b.w}, b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const uint4 pk8 = *reinterpret_cast<const uint4*>(popb + g * 256);
                            const float4 oa = *reinterpret_cast<const float4*>(orow8 + 32 * g);
                            const float4 oc = *reinterpret_cast<const float4*>(orow8 + 32 * g + 4);
                            const u32x4 ob_ = {pk_bf16(oa.x, oa.y), pk_bf16(oa.z, oa.w), pk_bf16(oc.x, oc.y), pk_bf16(oc.z, oc.w)};
                            if (g & 1) b1 = mfma_bf(u32x4{pk8.x, pk8.y, pk8.z, pk8.w}, ob_, b1);
                            else b0 = mfma_bf(u32x4{pk8.x, pk8.y, pk8.z, pk8.w}, ob_, b0);
                        }
                        *reinterpret_cast<float4*>(sSc + traj * CO_SP + 16 * nt + 4 * hi_t) =
                            make_float4(b0[0] + b1[0], b0[1] + b1[1], b0[2] + b1[2], b0[3] + b1[3]);
                    } else {
                        const float* orow = sQ + traj * CO_QP + 4 * hi_t;
                        const float* pop = sP + (nt * 8 * 64 + lane_t) * 4;
                        const float4 pb4 = *reinterpret_cast<const float4*>(sPb + 16 * nt + 4 * hi_t);
                        f32x4c a0 = {pb4.x, pb4.y, pb4.z, pb4.w}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            const float4 pk4 = *reinterpret_cast<const float4*>(pop + g * 256);
                            const float4 ov = *reinterpret_cast<const float4*>(orow + 16 * g);
                            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.x, ov.x, a0, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.y, ov.y, a1, 0, 0, 0);
                            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.z, ov.z, a0, 0, 0, 0);
                            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pk4.w, ov.w, a1, 0, 0, 0);
                        }
                        *reinterpret_cast<float4*>(sSc + traj * CO_SP + 16 * nt + 4 * hi_t) =
                            make_float4(a0[0] + a1[0], a0[1] + a1[1], a0[2] + a1[2], a0[3] + a1[3]);
                    }
                } else if (LEAN || A.has_local) {
                    co_local_tail_call(sT, sX + grp * 16 * CO_XP, sO1 + grp * 512, lo_t, hi_t);
                }
                ELG_STAMP(sc, 2);
                grp8_barrier(gcnt, bar_target, lane);
                ELG_STAMP(sc, 3);
            }
            // =============== owners (waves 0-3 of the group): rollout_fwd_coop_kernel's, unchanged ===============
            int any_left = 0;
            if (owner) {
                const int q4 = 4 * ow + (lane_t >> 4);
                const int m4 = g_lo + min(q4, ntraj - 1);
                const size_t bm4 = (size_t)b * A.M + m4;
                const bool active = q4 < ntraj && !row.fin;
                int sel = 0;
                float pr = 1.0f;
                if (decode_step) {
                    co_finish4<TSP, TRAIN, LEAN>(A, N1, lane_t, ow, ntraj, t, g_lo, (size_t)b, Rcap, sSc, sMask, sX, sState,
                                           q4 < ntraj ? row.fin : 1, sel, pr, ubuf, sc);
                } else if (!LEAN && A.mode == ELG_MODE_FORCED) {
                    sel = (A.forced && t < A.Tforced) ? A.forced[bm4 * A.Tforced + t] : 0;
                } else {
                    sel = (!TSP && t == 0) ? 0 : A.starts[m4];
                }
                if (active && (lane_t & 15) == 0) {
                    if (LEAN || A.actions) A.actions[bm4 * A.Tmax + t] = sel;
                    if (LEAN || A.probs) A.probs[((size_t)b * A.Tmax + t) * A.M + m4] = pr;
                }
                co_advance4<TSP, TRAIN, LEAN>(A, I, N1, lane_t, ow, ntraj, t, g_lo, (size_t)b, Rcap, row, sel, active, sMask, sQ, sX, sSc, sc);
                any_left = (q4 < ntraj && !row.fin) ? 1 : 0;
            }
            ELG_STAMP(sc, 11);
            if (BF) { ELG_C2_LOAD_K_BF(kAb, wg) } else { ELG_C2_LOAD_K(kA, wg) }
            ELG_STAMP(sc, 12);
            const unsigned wany = __ballot(any_left) != 0ull ? 1u : 0u;
            unsigned* slot = gany + 8 * (or_epoch & 1u);
            if (lane == 0) __hip_atomic_store(slot + wg, wany, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            grp8_barrier(gcnt, bar_target, lane);
            const uint4 a4 = *reinterpret_cast<const uint4*>(slot);          // (only the owner waves 0-3 can have anything left)
            or_epoch += 1u;
            ELG_STAMP(sc, 13);
            if (!__builtin_amdgcn_readfirstlane((int)(a4.x | a4.y | a4.z | a4.w))) break;
        }
#ifdef ELG_STAMPS
        if (A.scratch && lane == 0 && wg < 4)
            for (int i = 0; i < 16; ++i) A.scratch[((size_t)blockIdx.x * 8 + ow) * 16 + i] = sc.acc[i];
#endif
        if (owner) {
            const int q4 = 4 * ow + (lane >> 4);
            if (q4 < ntraj && (lane & 15) == 0) {
                const size_t bm = (size_t)b * A.M + g_lo + q4;
                if (A.reward) A.reward[bm] = -row.len;
                if (A.tlen) A.tlen[bm] = row.cnt;
            }
        }
        grp8_barrier(gcnt, bar_target, lane);
    }
}

template <bool TSP, bool TRAIN, bool BF, bool LEAN>
static int launch_fwd_coop3_l(const elg_rollout_args& A, hipStream_t stream, int stagger) {
    const size_t lds = ((size_t)CO_MAXTR * CO_QP + (size_t)CO_MAXTR * CO_SP + 4 * CO_MAXTR + 16 * CO_MAXTR +
                        ((A.N1 + 3) & ~3) + ((2 * A.N1 + 3) & ~3) + (size_t)CO_MAXTR * CO_XP + CL_SIZE + CO_NT * 32 * 64 + 16 * CO_NT +
                        2 * 512 + 64) * 4 + 64;
    auto kern = rollout_fwd_coop3_kernel<TSP, TRAIN, BF, LEAN>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, "coop rollout: hipFuncSetAttribute failed");
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(A.B * A.tiles), dim3(1024), lds, stream, A, stagger);
    return launch_status("rollout_fwd_coop3");
}
template <bool TSP, bool TRAIN, bool BF>
static int launch_fwd_coop3(const elg_rollout_args& A, hipStream_t stream, int stagger) {
    const bool lean = A.mode != ELG_MODE_FORCED && !A.forced && !A.uniforms && !A.full_probs && A.has_local && A.has_penalty &&
                      !A.euclidean && A.actions && A.probs &&
                      (!TRAIN || (!A.trA && A.trMask && A.trLse && A.trSlot && A.trF && (TSP || A.trLoad)));
    return lean ? launch_fwd_coop3_l<TSP, TRAIN, BF, true>(A, stream, stagger) : launch_fwd_coop3_l<TSP, TRAIN, BF, false>(A, stream, stagger);
}

int launch_fwd_coop_any(const elg_rollout_args& A, hipStream_t stream, bool tsp, bool train, int split) {
    // split: the split-group kernel (elg_rollout_args.variant = 4; ELG_COOP_KERNEL=split makes it what variant 0 runs, for A/B
    // timing of whole steps; ELG_COOP_STAGGER = group 1's start delay in units of s_sleep 64).  Same results bit for bit.
    static const int env_split = [] { const char* e = getenv("ELG_COOP_KERNEL"); return (e && e[0] == 's') ? 1 : (e && e[0] == 'w') ? 2 : 0; }();
    static const int stagger = [] { const char* e = getenv("ELG_COOP_STAGGER"); return e ? atoi(e) : 3; }();
    if (split == 0) split = env_split;
    if (split == 2) {                           // ELG_COOP_KERNEL=wide / variant 5: two groups of eight waves, four waves per SIMD
        note_kernel(ELG_KERNEL_COOP_WIDE);
        if (A.precision == 1) {
            if (tsp) return train ? launch_fwd_coop3<true, true, true>(A, stream, stagger) : launch_fwd_coop3<true, false, true>(A, stream, stagger);
            return train ? launch_fwd_coop3<false, true, true>(A, stream, stagger) : launch_fwd_coop3<false, false, true>(A, stream, stagger);
        }
        if (tsp) return train ? launch_fwd_coop3<true, true, false>(A, stream, stagger) : launch_fwd_coop3<true, false, false>(A, stream, stagger);
        return train ? launch_fwd_coop3<false, true, false>(A, stream, stagger) : launch_fwd_coop3<false, false, false>(A, stream, stagger);
    }
    if (split == 1) {
        note_kernel(ELG_KERNEL_COOP_SPLIT);
        if (A.precision == 1) {
            if (tsp) return train ? launch_fwd_coop2<true, true, true>(A, stream, stagger) : launch_fwd_coop2<true, false, true>(A, stream, stagger);
            return train ? launch_fwd_coop2<false, true, true>(A, stream, stagger) : launch_fwd_coop2<false, false, true>(A, stream, stagger);
        }
        if (tsp) return train ? launch_fwd_coop2<true, true, false>(A, stream, stagger) : launch_fwd_coop2<true, false, false>(A, stream, stagger);
        return train ? launch_fwd_coop2<false, true, false>(A, stream, stagger) : launch_fwd_coop2<false, false, false>(A, stream, stagger);
    }
    note_kernel(ELG_KERNEL_COOP);
    if (A.precision == 1) {
        if (tsp) return train ? launch_fwd_coop<true, true, true>(A, stream) : launch_fwd_coop<true, false, true>(A, stream);
        return train ? launch_fwd_coop<false, true, true>(A, stream) : launch_fwd_coop<false, false, true>(A, stream);
    }
    if (tsp) return train ? launch_fwd_coop<true, true, false>(A, stream) : launch_fwd_coop<true, false, false>(A, stream);
    return train ? launch_fwd_coop<false, true, false>(A, stream) : launch_fwd_coop<false, false, false>(A, stream);
}

}  // namespace elg
