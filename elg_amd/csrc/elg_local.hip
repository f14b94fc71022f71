// Local-policy backward over independent decode rows, on the matrix cores.
//
// The training forward saves, per decode row r = (trajectory, step), the features f_j and the node of each k-NN
// slot j; elg_rows_prep turns the loss gradient into du_j = d loss / d u_j.  Given those, rows are independent,
// so instead of replaying every trajectory step by step (one wavefront per trajectory, ~600 dependent cross-lane
// operations per row) a wavefront takes 16 rows at once and every contraction with a shared folded table becomes
// v_mfma_f32_16x16x4_f32 (exact f32).  Math (reference models.py:133-166, folded as in elg_rollout.h):
//   sc_h[j]  = la[h].f_j + lt[j][h]            alpha_h = softmax_j(sc_h)          F_h = sum_j alpha_h[j] f_j
//   o'[d]    = sum_j alpha_{h(d)}[j] lcv[j][d] + lAv[d].F_{h(d)}                  g' = lWc o' + lbc
//   u_j      = lpe[j].g' + (sum_d g'[d] lWe[d]).f_j
// backward for the tables only (the features carry no gradient):
//   dw = sum_j du_j f_j      dg' = lpe^T du + lWe dw      do' = lWc^T dg'      dF_h = sum_{d in h} do'[d] lAv[d]
//   dalpha_h[j] = sum_{d in h} lcv[j][d] do'[d] + dF_h.f_j      dsc_h = alpha_h (dalpha_h - <alpha_h, dalpha_h>)
//   d lpe = du (x) g'   d lWe = g' (x) dw   d lbc = dg'   d lWc = dg' (x) o'   d lAv[d] = do'[d] F_{h(d)}
//   d lcv[j][d] = alpha_{h(d)}[j] do'[d]   d lt[j][h] = dsc_h[j]   d la[h] = sum_j dsc_h[j] f_j       (summed over rows)
//
// Layouts (lane l: lo = l & 15, hi = l >> 4).  MFMA: A[i = lo][k = hi], B[k = hi][j = lo], D[i = 4 hi + reg][j = lo].
//   L1 "feature-major": X1[t][v] = X[feature 16 t + 4 hi + v][row lo]   -- what a D tile looks like when the row is
//       the B operand's column; it feeds the next feature contraction directly as a B operand (k-slot hi of step
//       v <-> feature 16 t + 4 hi + v).  The whole per-row chain runs in L1.
//   L2 "row-major":     X2[t][v] = X[row 4 hi + v][feature 16 t + lo]   -- both operands of a contraction over rows
//       (the table gradients) must look like this.  L2 copies come from a 16 x 16 transpose through per-wave LDS
//       (one ds_write_b128 + four ds_read_b32 per tile, conflict-free with a 20-float pitch).
#include "elg_common.h"
#include "elg_rollout.h"
#include "../../include/elg_hip.h"
#include <cstdlib>
#include <string>

namespace elg {
int fail(int code, const std::string& msg);
int launch_status(const char* what);

using f32x4 = __attribute__((ext_vector_type(4))) float;
#define ELG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Table images in LDS.  Every MFMA operand that comes from a table is one ds_read_b128 (four k-steps / four D rows at once):
// a table is kept row-major ([row][36]) where a lane's four values are consecutive columns, and transposed ([col][52] /
// [col][36]) where they are consecutive rows.  Pitches 36 and 52 keep the 16 lanes of a b128 group on distinct banks.  (With one
// wave per SIMD -- 512 registers -- every LDS round trip in front of an MFMA is exposed: scalar operand reads made this kernel
// latency bound.)
constexpr int LP = 36;                       // pitch of the row-major 32-wide tables
constexpr int LQ = 52;                       // pitch of the transposed 48-wide tables
constexpr int TP = 20;                       // pitch of a transpose tile
constexpr int S_LCV = 0;                     // [48 slots][36]   lcv[j][d]
constexpr int S_LCVT = S_LCV + 48 * LP;      // [32 d][52]       lcv[j][d] transposed
constexpr int S_LPET = S_LCVT + 32 * LQ;     // [32 d][52]       lpe[j][d] transposed
constexpr int S_LWC = S_LPET + 32 * LQ;      // [32][36]         lwc[d'][d]
constexpr int S_LWCT = S_LWC + 32 * LP;      // [32][36]         lwc transposed
constexpr int S_LTT = S_LWCT + 32 * LP;      // [4 heads][48]    lt[j][h] transposed
constexpr int S_LAV = S_LTT + 4 * 48;        // [32][4]          lAv[d][k], k < 3
constexpr int S_LWE = S_LAV + 32 * 4;        // [32][4]          lWe[d][k], k < 3
constexpr int S_LBC = S_LWE + 32 * 4;        // [32]
constexpr int S_TABLES = S_LBC + 32;         // = 7840 floats
constexpr int S_TR = 16 * TP;                // one transpose tile (320 floats)
constexpr int NTRB = 8;                      // transpose buffers per wave (a batch of transposes = ONE LDS round trip)

// X (L1 or L2 tile in registers) -> X^T in the same register layout, through the per-wave buffer `buf`.
__device__ __forceinline__ f32x4 transpose16(f32x4 x, float* buf, int lo, int hi) {
    *reinterpret_cast<float4*>(buf + lo * TP + 4 * hi) = make_float4(x[0], x[1], x[2], x[3]);
    wave_lds_fence();
    f32x4 y;
#pragma unroll
    for (int v = 0; v < 4; ++v) y[v] = buf[(4 * hi + v) * TP + lo];
    wave_lds_fence();
    return y;
}

// N tiles at once: all stores, one fence, all loads, one fence.  (At one wave per SIMD every LDS round trip is exposed: the 34
// transposes of a row tile, one round trip each, were a fifth of the tile's cycles; batched they are six.)
template <int N>
__device__ __forceinline__ void transpose16_batch(const f32x4 (&x)[N], f32x4 (&y)[N], float* bufs, int lo, int hi) {
    static_assert(N <= NTRB, "transpose buffers");
#pragma unroll
    for (int i = 0; i < N; ++i) *reinterpret_cast<float4*>(bufs + i * S_TR + lo * TP + 4 * hi) = make_float4(x[i][0], x[i][1], x[i][2], x[i][3]);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) y[i][v] = bufs[i * S_TR + (4 * hi + v) * TP + lo];
    wave_lds_fence();
}

template <int JT>
__global__ __launch_bounds__(256) void local_bwd_rows_kernel(const float* __restrict__ loc, const float* __restrict__ trF,
                                                             const int* __restrict__ trSlot,
                                                             const float* __restrict__ rowDU, float* __restrict__ gloc,
                                                             int B, int R, long long Rcap, const int* __restrict__ T_dev,
                                                             int M) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (T_dev) R = min(R, T_dev[0] * M);                        // step count still on the device (elg_decoder_bwd_args.T_dev)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    float* sT = lds;                                            // tables
    float* trb = lds + S_TABLES + wave * (NTRB * S_TR);         // this wave's transpose buffers
    float* sAcc = lds + S_TABLES + 4 * NTRB * S_TR;             // ELG_LOC_SIZE accumulators of the workgroup
    // ---- stage the tables (padded pitches) and clear the accumulators
    for (int i = tid; i < 48 * 32; i += 256) {
        const int j = i >> 5, d = i & 31;
        const float cv = loc[ELG_LOC_LCV + i];
        sT[S_LCV + j * LP + d] = cv;
        sT[S_LCVT + d * LQ + j] = cv;
        sT[S_LPET + d * LQ + j] = loc[ELG_LOC_LPE + i];
    }
    for (int i = tid; i < 32 * 32; i += 256) {
        const float wv = loc[ELG_LOC_LWC + i];
        sT[S_LWC + (i >> 5) * LP + (i & 31)] = wv;
        sT[S_LWCT + (i & 31) * LP + (i >> 5)] = wv;
    }
    for (int i = tid; i < 48 * 4; i += 256) sT[S_LTT + (i & 3) * 48 + (i >> 2)] = loc[ELG_LOC_LT + i];
    for (int i = tid; i < 128; i += 256) {
        const int d = i >> 2, k = i & 3;
        sT[S_LAV + i] = k < 3 ? loc[ELG_LOC_LAV + 3 * d + k] : 0.f;
        sT[S_LWE + i] = k < 3 ? loc[ELG_LOC_LWE + 3 * d + k] : 0.f;
    }
    for (int i = tid; i < 32; i += 256) sT[S_LBC + i] = loc[ELG_LOC_LBC + i];
    for (int i = tid; i < ELG_LOC_SIZE; i += 256) sAcc[i] = 0.f;
    float la[ELG_LH][3];                                         // uniform
#pragma unroll
    for (int h = 0; h < ELG_LH; ++h)
#pragma unroll
        for (int k = 0; k < 3; ++k) la[h][k] = loc[ELG_LOC_LA + 3 * h + k];
    __syncthreads();

    // ---- table-gradient accumulators
    f32x4 aLpe[JT][2], aLcv[JT][2], aLwc[2][2], aLav[2], aLwe[2];
    float aLbc[2], aLt[ELG_LH][JT], aLa[ELG_LH][3];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) { aLpe[jt][0] = z4; aLpe[jt][1] = z4; aLcv[jt][0] = z4; aLcv[jt][1] = z4; }
#pragma unroll
    for (int a = 0; a < 2; ++a) { aLwc[a][0] = z4; aLwc[a][1] = z4; aLav[a] = z4; aLwe[a] = z4; aLbc[a] = 0.f; }
#pragma unroll
    for (int h = 0; h < ELG_LH; ++h) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) aLt[h][jt] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) aLa[h][k] = 0.f;
    }

    const int tiles_per_b = (R + 15) >> 4;
    const long long ntiles = (long long)B * tiles_per_b;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // tile loads (cotangents, slot codes, slot features of row lo; rows past R are clamped and get du = 0, so they
    // contribute nothing); the loads of the wave's next tile are issued before the current one is consumed
#define ELG_LB_LOAD(TILE, DU, SL, FF)                                                                              \
    {                                                                                                              \
        const int b_ = (int)((TILE) / tiles_per_b);                                                                \
        const int row0_ = (int)((TILE) % tiles_per_b) << 4;                                                        \
        const int rl_ = min(lo, R - 1 - row0_);                                                                    \
        const float live_ = (lo <= R - 1 - row0_) ? 1.f : 0.f;                                                     \
        const float* duRow = rowDU + ((size_t)b_ * R + row0_ + rl_) * ELG_SLOT_STRIDE + 4 * hi;                    \
        const size_t src_ = (size_t)b_ * Rcap + row0_ + rl_;                                                       \
        const int* slRow = trSlot + src_ * ELG_SLOT_STRIDE + 4 * hi;                                               \
        const float* fRow = trF + src_ * (3 * ELG_SLOT_STRIDE) + 4 * hi;                                           \
        _Pragma("unroll") for (int jt = 0; jt < JT; ++jt) {                                                        \
            const float4 t = *reinterpret_cast<const float4*>(duRow + 16 * jt);                                    \
            DU[jt] = f32x4{t.x * live_, t.y * live_, t.z * live_, t.w * live_};                                    \
            SL[jt] = *reinterpret_cast<const int4*>(slRow + 16 * jt);                                              \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) {                                                        \
                const float4 u = *reinterpret_cast<const float4*>(fRow + k * ELG_SLOT_STRIDE + 16 * jt);           \
                FF[k][jt] = f32x4{u.x, u.y, u.z, u.w};                                                             \
            }                                                                                                      \
        }                                                                                                          \
    }
    const long long tstride = (long long)gridDim.x * 4;
    const long long tile0 = (long long)blockIdx.x * 4 + wave_u;
    f32x4 du1[JT], f1[3][JT];
    int4 sl1[JT];
    if (tile0 < ntiles) ELG_LB_LOAD(tile0, du1, sl1, f1)
    for (long long tile = tile0; tile < ntiles; tile += tstride) {
        const int b = (int)(tile / tiles_per_b);
        const int row0 = (int)(tile % tiles_per_b) << 4;
        const int rleft = R - 1 - row0;                          // last valid row of the tile, relative (>= 0)
        f32x4 du1n[JT], f1n[3][JT];
        int4 sl1n[JT];
        {
            const long long tn = (tile + tstride < ntiles) ? tile + tstride : tile;      // last prefetch: a valid re-read
            ELG_LB_LOAD(tn, du1n, sl1n, f1n)
        }
        bool any = false;
        bool msk[JT][4];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            any = any || du1[jt][0] != 0.f || du1[jt][1] != 0.f || du1[jt][2] != 0.f || du1[jt][3] != 0.f;
            msk[jt][0] = sl1[jt].x < 0; msk[jt][1] = sl1[jt].y < 0; msk[jt][2] = sl1[jt].z < 0; msk[jt][3] = sl1[jt].w < 0;
        }
        if (__ballot(any)) {                                     // else: first moves / finished trajectories / padding
        // ---- forward recompute: attention weights, F, o', g'   (L1)
        f32x4 al[ELG_LH][JT];
        float F[ELG_LH][3];
#pragma unroll
        for (int h = 0; h < ELG_LH; ++h) {
            float mx = ELG_NEG_INF;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const float4 lt4 = *reinterpret_cast<const float4*>(sT + S_LTT + h * 48 + 16 * jt + 4 * hi);
                const float ltv[4] = {lt4.x, lt4.y, lt4.z, lt4.w};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float s = ltv[v];
                    s = fmaf(la[h][0], f1[0][jt][v], s);
                    s = fmaf(la[h][1], f1[1][jt][v], s);
                    s = fmaf(la[h][2], f1[2][jt][v], s);
                    s = msk[jt][v] ? ELG_NEG_INF : s;
                    al[h][jt][v] = s;
                    mx = fmaxf(mx, s);
                }
            }
            mx = quarters_max(mx);
            float den = 0.f;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float e = msk[jt][v] ? 0.f : __expf(al[h][jt][v] - mx);
                    al[h][jt][v] = e;
                    den += e;
                }
            den = quarters_sum(den);
            const float inv = den > 0.f ? 1.0f / den : 0.f;
            float fk[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float a = al[h][jt][v] * inv;
                    al[h][jt][v] = a;
#pragma unroll
                    for (int k = 0; k < 3; ++k) fk[k] = fmaf(a, f1[k][jt][v], fk[k]);
                }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                fk[k] = quarters_sum(fk[k]);
                F[h][k] = fk[k];
            }
        }
        float dw[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int k = 0; k < 3; ++k) dw[k] = fmaf(du1[jt][v], f1[k][jt][v], dw[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) { dw[k] = quarters_sum(dw[k]); }

        const bool up = hi >= 2;                                 // rows 4 hi + v of a d-tile belong to head 2 dt + up
        f32x4 o1[2], g1[2], dg1[2], do1[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x4 Pa = z4, Pb = z4;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const float4 a4 = *reinterpret_cast<const float4*>(sT + S_LCVT + (16 * dt + lo) * LQ + 16 * jt + 4 * hi);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    Pa = ELG_MFMA(av[v], al[2 * dt][jt][v], Pa);
                    Pb = ELG_MFMA(av[v], al[2 * dt + 1][jt][v], Pb);
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int d = 16 * dt + 4 * hi + v;
                float x = up ? Pb[v] : Pa[v];
                const float4 lav = *reinterpret_cast<const float4*>(sT + S_LAV + 4 * d);
                x = fmaf(lav.x, up ? F[2 * dt + 1][0] : F[2 * dt][0], x);
                x = fmaf(lav.y, up ? F[2 * dt + 1][1] : F[2 * dt][1], x);
                x = fmaf(lav.z, up ? F[2 * dt + 1][2] : F[2 * dt][2], x);
                o1[dt][v] = x;
            }
        }
#pragma unroll
        for (int dq = 0; dq < 2; ++dq) {                         // g'[d'] tile dq
            const float4 bc4 = *reinterpret_cast<const float4*>(sT + S_LBC + 16 * dq + 4 * hi);
            f32x4 acc = {bc4.x, bc4.y, bc4.z, bc4.w};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float4 w4 = *reinterpret_cast<const float4*>(sT + S_LWC + (16 * dq + lo) * LP + 16 * dt + 4 * hi);
                acc = ELG_MFMA(w4.x, o1[dt][0], acc);
                acc = ELG_MFMA(w4.y, o1[dt][1], acc);
                acc = ELG_MFMA(w4.z, o1[dt][2], acc);
                acc = ELG_MFMA(w4.w, o1[dt][3], acc);
            }
            g1[dq] = acc;
        }
        // ---- backward chain (L1)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x4 acc = z4;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const float4 p4 = *reinterpret_cast<const float4*>(sT + S_LPET + (16 * dt + lo) * LQ + 16 * jt + 4 * hi);
                acc = ELG_MFMA(p4.x, du1[jt][0], acc);
                acc = ELG_MFMA(p4.y, du1[jt][1], acc);
                acc = ELG_MFMA(p4.z, du1[jt][2], acc);
                acc = ELG_MFMA(p4.w, du1[jt][3], acc);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 we = *reinterpret_cast<const float4*>(sT + S_LWE + 4 * (16 * dt + 4 * hi + v));
                acc[v] = fmaf(we.x, dw[0], acc[v]);
                acc[v] = fmaf(we.y, dw[1], acc[v]);
                acc[v] = fmaf(we.z, dw[2], acc[v]);
            }
            dg1[dt] = acc;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x4 acc = z4;
#pragma unroll
            for (int dq = 0; dq < 2; ++dq) {
                const float4 w4 = *reinterpret_cast<const float4*>(sT + S_LWCT + (16 * dt + lo) * LP + 16 * dq + 4 * hi);
                acc = ELG_MFMA(w4.x, dg1[dq][0], acc);
                acc = ELG_MFMA(w4.y, dg1[dq][1], acc);
                acc = ELG_MFMA(w4.z, dg1[dq][2], acc);
                acc = ELG_MFMA(w4.w, dg1[dq][3], acc);
            }
            do1[dt] = acc;
        }
        float dF[ELG_LH][3];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            float lav[4][3];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t4 = *reinterpret_cast<const float4*>(sT + S_LAV + 4 * (16 * dt + 4 * hi + v));
                lav[v][0] = t4.x; lav[v][1] = t4.y; lav[v][2] = t4.z;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float part = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) part = fmaf(do1[dt][v], lav[v][k], part);
                part = x16_sum(part);                      // the two hi groups of one head
                dF[2 * dt][k] = __shfl(part, lo, ELG_WAVE);       // held by hi = 0, 1
                dF[2 * dt + 1][k] = __shfl(part, lo + 32, ELG_WAVE);   // held by hi = 2, 3
            }
        }
        // per head: dalpha, dsc; d la in place; dsc / alpha transposed for d lt / d lcv
        f32x4 o2[2], g2[2], dg2[2], do2[2];
        {
            const f32x4 tin[8] = {o1[0], o1[1], g1[0], g1[1], dg1[0], dg1[1], do1[0], do1[1]};
            f32x4 tout[8];
            transpose16_batch<8>(tin, tout, trb, lo, hi);
            o2[0] = tout[0]; o2[1] = tout[1]; g2[0] = tout[2]; g2[1] = tout[3];
            dg2[0] = tout[4]; dg2[1] = tout[5]; do2[0] = tout[6]; do2[1] = tout[7];
        }
#pragma unroll
        for (int h = 0; h < ELG_LH; ++h) {
            const int dt = h >> 1;
            const bool mine = up == bool(h & 1);                 // this lane's d rows belong to head h
            f32x4 dob;
#pragma unroll
            for (int v = 0; v < 4; ++v) dob[v] = mine ? do1[dt][v] : 0.f;
            f32x4 dal[JT];
            float ts = 0.f;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                f32x4 acc = z4;
                {
                    const float4 c4 = *reinterpret_cast<const float4*>(sT + S_LCV + (16 * jt + lo) * LP + 16 * dt + 4 * hi);
                    acc = ELG_MFMA(c4.x, dob[0], acc);
                    acc = ELG_MFMA(c4.y, dob[1], acc);
                    acc = ELG_MFMA(c4.z, dob[2], acc);
                    acc = ELG_MFMA(c4.w, dob[3], acc);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[v] = fmaf(dF[h][k], f1[k][jt][v], acc[v]);
                    ts = fmaf(al[h][jt][v], acc[v], ts);
                }
                dal[jt] = acc;
            }
            ts = quarters_sum(ts);
            const bool lane_lo_half = lo < 8;
            f32x4 tin[2 * JT], tout[2 * JT];                    // dsc | alpha of the head, every slot tile: one round trip
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                f32x4 dsc;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    dsc[v] = al[h][jt][v] * (dal[jt][v] - ts);
#pragma unroll
                    for (int k = 0; k < 3; ++k) aLa[h][k] = fmaf(dsc[v], f1[k][jt][v], aLa[h][k]);
                }
                tin[jt] = dsc;
                tin[JT + jt] = al[h][jt];
            }
            transpose16_batch<2 * JT>(tin, tout, trb, lo, hi);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const f32x4 dsc2 = tout[jt];                                       // [row 4 hi + v][slot 16 jt + lo]
                aLt[h][jt] += (dsc2[0] + dsc2[1]) + (dsc2[2] + dsc2[3]);
                const f32x4 al2 = tout[JT + jt];
                // d lcv[j][d] += alpha_h[r][j] do'[r][d] for the 8 channels d of head h (columns lo of tile dt)
                const bool col = lane_lo_half == !(h & 1);
#pragma unroll
                for (int v = 0; v < 4; ++v) aLcv[jt][dt] = ELG_MFMA(al2[v], col ? do2[dt][v] : 0.f, aLcv[jt][dt]);
            }
        }
        // ---- remaining table gradients: contractions over the 16 rows (L2 operands)
        (void)b; (void)rleft;
        // F and dw as feature-major tiles (column c = 3 h + k of F, c = k of dw), the cotangents du1 (rows past R carry zeros):
        // their row-major forms in one batch -- du2[v] = du[row 4 hi + v][slot 16 jt + lo] used to be a second, dependent global read
        f32x4 Ft, dwt;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float x0 = 0.f;
#pragma unroll
            for (int c = 0; c < 12; ++c) x0 = (4 * hi + v == c) ? F[c / 3][c % 3] : x0;
            Ft[v] = x0;
            dwt[v] = (hi == 0 && v < 3) ? dw[v] : 0.f;
        }
        f32x4 tin2[2 + JT], tout2[2 + JT];
        tin2[0] = Ft; tin2[1] = dwt;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) tin2[2 + jt] = du1[jt];
        transpose16_batch<2 + JT>(tin2, tout2, trb, lo, hi);
        const f32x4 F2 = tout2[0], dw2 = tout2[1];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const f32x4 du2 = tout2[2 + jt];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int v = 0; v < 4; ++v) aLpe[jt][dt] = ELG_MFMA(du2[v], g2[dt][v], aLpe[jt][dt]);
        }
#pragma unroll
        for (int dq = 0; dq < 2; ++dq)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int v = 0; v < 4; ++v) aLwc[dq][dt] = ELG_MFMA(dg2[dq][v], o2[dt][v], aLwc[dq][dt]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                aLav[dt] = ELG_MFMA(do2[dt][v], F2[v], aLav[dt]);                // D[d = 16 dt + 4 hi + v'][c = lo]
                aLwe[dt] = ELG_MFMA(g2[dt][v], dw2[v], aLwe[dt]);
            }
            aLbc[dt] += (dg2[dt][0] + dg2[dt][1]) + (dg2[dt][2] + dg2[dt][3]);  // column d = 16 dt + lo, rows 4 hi + v
        }
        }                                                        // if (any)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            du1[jt] = du1n[jt]; sl1[jt] = sl1n[jt];
#pragma unroll
            for (int k = 0; k < 3; ++k) f1[k][jt] = f1n[k][jt];
        }
    }
#undef ELG_LB_LOAD

    // ---- fold this wave's accumulators into the workgroup's image of the table gradient, then one flush
#pragma unroll
    for (int h = 0; h < ELG_LH; ++h) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) { aLt[h][jt] = quarters_sum(aLt[h][jt]); }
#pragma unroll
        for (int k = 0; k < 3; ++k) aLa[h][k] = wave_sum(aLa[h][k]);
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) { aLbc[dt] = quarters_sum(aLbc[dt]); }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int j = 16 * jt + 4 * hi + v, d = 16 * dt + lo;
                        sAcc[ELG_LOC_LPE + 32 * j + d] += aLpe[jt][dt][v];
                        sAcc[ELG_LOC_LCV + 32 * j + d] += aLcv[jt][dt][v];
                    }
#pragma unroll
            for (int dq = 0; dq < 2; ++dq)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        sAcc[ELG_LOC_LWC + 32 * (16 * dq + 4 * hi + v) + 16 * dt + lo] += aLwc[dq][dt][v];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int d = 16 * dt + 4 * hi + v, h = d >> 3;
                    // D[d][c]: d lAv[d][k] sits at column c = 3 h(d) + k, d lWe[d][k] at column c = k
                    if (lo >= 3 * h && lo < 3 * h + 3) sAcc[ELG_LOC_LAV + 3 * d + (lo - 3 * h)] += aLav[dt][v];
                    if (lo < 3) sAcc[ELG_LOC_LWE + 3 * d + lo] += aLwe[dt][v];
                }
            if (hi == 0) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) sAcc[ELG_LOC_LBC + 16 * dt + lo] += aLbc[dt];
#pragma unroll
                for (int h = 0; h < ELG_LH; ++h)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) sAcc[ELG_LOC_LT + 4 * (16 * jt + lo) + h] += aLt[h][jt];
            }
            if (lane == 0) {
#pragma unroll
                for (int h = 0; h < ELG_LH; ++h)
#pragma unroll
                    for (int k = 0; k < 3; ++k) sAcc[ELG_LOC_LA + 3 * h + k] += aLa[h][k];
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < ELG_LOC_SIZE; i += 256) {
        const float v = sAcc[i];
        if (v != 0.f) atomicAdd(gloc + i, v);
    }
}

}  // namespace elg

using namespace elg;

extern "C" int elg_local_bwd_rows(const float* loc, const float* trF, const int32_t* trSlot, const float* rowDU,
                                  float* gloc, int B, int R, int64_t Rcap, int n_slots, const int32_t* T_dev, int M,
                                  int max_workgroups, void* stream) {
    if (!loc || !trF || !trSlot || !rowDU || !gloc) return fail(ELG_EINVAL, "local_bwd_rows: null buffer");
    if (B <= 0 || R <= 0 || Rcap < R) return fail(ELG_EINVAL, "local_bwd_rows: bad sizes");
    if (T_dev && M <= 0) return fail(ELG_EINVAL, "local_bwd_rows: T_dev needs M");
    if (n_slots <= 0 || n_slots > ELG_SLOT_STRIDE) return fail(ELG_EINVAL, "local_bwd_rows: local_size must be <= 47");
    const size_t lds = (size_t)(S_TABLES + 4 * NTRB * S_TR + ELG_LOC_SIZE) * sizeof(float);
    const long long ntiles = (long long)B * ((R + 15) / 16);
    int grid = (int)std::min<long long>((ntiles + 3) / 4, 256);         // 464 registers: one workgroup per CU
    if (max_workgroups > 0) grid = std::min(grid, max_workgroups);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    if (n_slots <= 32) {
        static DynLds optin;
        if (!optin.opt_in(reinterpret_cast<const void*>(local_bwd_rows_kernel<2>), lds)) return fail(ELG_ELAUNCH, "local_bwd_rows: hipFuncSetAttribute failed");
        hipLaunchKernelGGL(local_bwd_rows_kernel<2>, dim3(grid), dim3(256), lds, s, loc, trF, trSlot, rowDU, gloc, B, R,
                           (long long)Rcap, T_dev, M);
    } else {
        static DynLds optin;
        if (!optin.opt_in(reinterpret_cast<const void*>(local_bwd_rows_kernel<3>), lds)) return fail(ELG_ELAUNCH, "local_bwd_rows: hipFuncSetAttribute failed");
        hipLaunchKernelGGL(local_bwd_rows_kernel<3>, dim3(grid), dim3(256), lds, s, loc, trF, trSlot, rowDU, gloc, B, R,
                           (long long)Rcap, T_dev, M);
    }
    return launch_status("local_bwd_rows");
}
