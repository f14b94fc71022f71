// Decoder backward over the rows a training forward saved (time-major r = t*M + m): elg_decoder_bwd.
//
// autograd of reference CVRP/models.py:322-423 (pointer scores, clip, softmax, chosen probability) w.r.t. the decoder
// tables, for the steps of train.py:112-125.  Two launches:
//   pointer_bwd_kernel   d(pre-clip score) rows  dl[r][n] = w_r (c_sel [n == a_r] - p_n c_n)   (never stored),
//                        dO = dl PK (the glimpse output's cotangent), dPK += dl^T O, dpb += sum_r dl,
//                        dU[r][j] = dl[r][slot_j] / ensemble (local policy cotangent), query-gather indices
//   glimpse_bwd_mfma     (csrc/elg_bwd.hip) glimpse attention backward -> dK, dV and, through its LDS epilogue,
//                        dQ1 / dQ2 / d wl
// This replaces a row-prep kernel that wrote dl and two one-hot matrices (1.0 GB), three library GEMMs (dO, dPK, the
// one-hot gather) and three framework reductions.
#include "elg_common.h"
#include "../../include/elg_hip.h"
#include "elg_bwd_internal.h"
#include "elg_bf16.h"
#include <string>

namespace elg {
int fail(int code, const std::string& msg);
int launch_status(const char* what);

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct PtrBwd {
    const float* gprob; const float* pval; const int* tlen; const int* actions;
    const float* PC; const float* Csel; const int* Slot; const float* rowO; const float* PK;
    float* dO; float* dPK; float* dpb; float* rowDU; int* idx_prev; int* idx_first; float4* rowW;
    int B, T, M, N1, Tcap_act, t0, splits;
    long long Rcap;
    float inv_ens;
    const int* T_dev;       // device-resident step count (<= T), or NULL
    int gT;                 // time extent of gprob / pval
};

// decode steps covered: the host's value, or the rollout's own count when the host has not read it yet
__device__ __forceinline__ int eff_T(const PtrBwd& a) { return a.T_dev ? min(a.T, a.T_dev[0]) : a.T; }

// per decode row: weight w = gprob * pval * valid, w * c_sel, the chosen node, and the nodes its query was gathered at
__global__ __launch_bounds__(256) void row_weights_kernel(const PtrBwd a) {
    const long long R = (long long)eff_T(a) * a.M;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.B * R) return;
    const int b = (int)(i / R), r = (int)(i - (long long)b * R);
    const int t = r / a.M, m = r - t * a.M;
    const size_t bm = (size_t)b * a.M + m;
    const bool valid = t >= a.t0 && t < a.tlen[bm];
    const size_t gi = ((size_t)b * a.gT + t) * a.M + m;
    const float w = valid ? a.gprob[gi] * a.pval[gi] : 0.f;
    const int* act = a.actions + bm * a.Tcap_act;
    const int sel = act[t];
    a.rowW[i] = make_float4(w, w * a.Csel[(size_t)b * a.Rcap + r], i2f(sel), 0.f);
    a.idx_prev[i] = t > 0 ? act[t - 1] : 0;
    if (a.idx_first) a.idx_first[i] = act[0];
}

// grid (splits, B), 512 threads: wave h owns channels 16 h .. 16 h + 15 of dO / dPK.  A thread stages 4 nodes of one row
// of the dl tile (32 threads per row) and one 16-byte piece of the O tile; the next tile's values are fetched into
// registers before the current tile's MFMAs.
// BF (elg_decoder_bwd mfma_mode 3, the backward of a bf16 rollout): both products on v_mfma_f32_16x16x32_bf16 -- dl, PK and O
// rounded to bf16 as operands, f32 accumulation (oracle: _PtrBF).  dO contracts over the nodes, two node tiles per instruction
// (k-slot (hi, j) = node 16 (2 p) + 4 hi + j, (hi, 4 + j) = node 16 (2 p + 1) + 4 hi + j); dPK contracts over the tile's 16 rows,
// k-slots (hi, 4..7) empty.  11 instructions of 16 cycles per tile instead of 56 of 32: the kernel is then bound by its 1.3 GB of rows.
template <int NT, bool BF>
__global__ __launch_bounds__(512, 4) void pointer_bwd_kernel(const PtrBwd a) {
    // pitches = 4 x odd (mod 64 floats): the fragment reads / stores that put the ROW on the lane (16 lanes x 16 bytes at one
    // pitch apart) then cover all 64 banks; 120 and 144 were 2- and 4-way conflicts (PMC: 43 % of the LDS-active cycles)
    constexpr int DLP = 16 * NT + 4;                                    // pitch of a dl row in LDS (floats)
    constexpr int OP = 148;                                             // pitch of an O / dO row
    constexpr int NPT = (16 * NT + 31) / 32;                            // nodes per thread and row
    constexpr int TPD = 20;                                             // pitch of the transposed dl tile (conflict-free b128 reads)
    __shared__ __attribute__((aligned(16))) float sDL[16 * DLP], sDLT[16 * NT * TPD], sO[16 * OP], sDO[16 * OP];
    const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int b = blockIdx.y, N1 = a.N1;
    const int R = eff_T(a) * a.M;
    // live rows of this instance: the rows are time-major, so the decode steps first_decode_step .. max_m tlen[b,m] - 1 are
    // the contiguous range [t0 M, Tb M); every row outside has weight 0 (its dO / rowDU are exact zeros, written below
    // without any arithmetic; the glimpse backward walks the same range)
    __shared__ int sTb;
    if (tid == 0) sTb = 0;
    __syncthreads();
    {
        int mx = 0;
        for (int m = tid; m < a.M; m += 512) mx = max(mx, a.tlen[(size_t)b * a.M + m]);
        mx = (int)wave_max((float)mx);
        if (lane == 0) atomicMax(&sTb, mx);
    }
    __syncthreads();
    const int Rb = min(R, sTb * a.M);
    const int tile_first = live_tile_first(a.t0, a.M), ntile = (Rb + 15) >> 4;
    if (blockIdx.x == 0 && a.rowDU) {
        // rows [0, 16 tile_first) and [16 ntile, R): elg_local_bwd_rows reads them (and skips their tiles on du == 0)
        const int head = min(tile_first << 4, R), tail0 = min(ntile << 4, R);
        float* du = a.rowDU + (size_t)b * R * 48;
        for (int i = tid; i < head * 48; i += 512) du[i] = 0.f;
        for (int i = tail0 * 48 + tid; i < R * 48; i += 512) du[i] = 0.f;
    }
    const int per = (max(ntile - tile_first, 0) + a.splits - 1) / a.splits;
    const int t_lo = tile_first + blockIdx.x * per, t_hi = min(ntile, t_lo + per);
    if (t_lo >= t_hi) return;
    // PK operand image: value (nt, j) of lane (lo, hi) = PK[node 16 nt + 4 hi + j][16 h + lo] (0 past the last node)
    float pk[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 16 * nt + 4 * hi + j;
            pk[nt][j] = n < N1 ? a.PK[((size_t)b * N1 + n) * ELG_E + h * 16 + lo] : 0.f;
        }
    f32x4 dpk[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dpk[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dpb_acc = 0.f;                                               // thread n < N1: column sum of dl
    const size_t brow = (size_t)b * a.Rcap;
    const int srow = tid >> 5, sl = tid & 31;                           // staging: row of the tile, lane within the row
    // ---- staging loads of a tile into registers (all unguarded: rows past R are clamped and get w = 0)
    float pc[NPT];
    float4 rw, o4;
#define PB_LOAD(TILE, PCV, RWV, OV, SLV)                                                                       \
    {                                                                                                      \
        const int rr_ = min(((TILE) << 4) + srow, R - 1);                                                  \
        const float* p_ = a.PC + (brow + rr_) * N1;                                                        \
        _Pragma("unroll") for (int k = 0; k < NPT; ++k) PCV[k] = p_[min(sl + 32 * k, N1 - 1)];             \
        RWV = a.rowW[(size_t)b * R + rr_];                                                                 \
        if (((TILE) << 4) + srow >= R) { RWV.x = 0.f; RWV.y = 0.f; }                                       \
        OV = *reinterpret_cast<const float4*>(a.rowO + (brow + rr_) * ELG_E + 4 * sl);                     \
        /* slot indices of the tile's 16 x 48 cotangent cells: fetched with the tile, a load issued when the dl tile is   \
         * in LDS exposed a full memory latency per tile */                                                \
        if (a.rowDU) {                                                                                     \
            SLV[0] = slot_p[min((size_t)(brow + ((TILE) << 4)) * 48 + tid, slot_last)];                    \
            SLV[1] = slot_p[min((size_t)(brow + ((TILE) << 4)) * 48 + 512 + (tid & 255), slot_last)];      \
        }                                                                                                  \
    }
    int slv[2] = {-1, -1};
    const int* slot_p = a.rowDU ? a.Slot : reinterpret_cast<const int*>(a.rowW);
    const size_t slot_last = (brow + R) * 48 - 1;
    PB_LOAD(t_lo, pc, rw, o4, slv)
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int r0 = tile << 4;
        // dl tile -> LDS (nodes past N1 are exact zeros), O tile -> LDS
        {
            const int sel = f2i(rw.z);
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                const int n = sl + 32 * k;
                float v = -rw.x * pc[k];
                if (n == sel) v += rw.y;
                if (n < 16 * NT) {
                    const float x = n < N1 ? v : 0.f;
                    sDL[srow * DLP + n] = x;                            // [row][node]: operand of dO^T, column sums
                    sDLT[n * TPD + srow] = x;                           // [node][row]: operand of dPK (one ds_read_b128 per chunk)
                }
            }
            *reinterpret_cast<float4*>(sO + srow * OP + 4 * sl) = o4;
        }
        __syncthreads();
        int slvn[2] = {-1, -1};
        {
            const int tn = min(tile + 1, t_hi - 1);                     // the last prefetch re-reads its own tile, unused
            PB_LOAD(tn, pc, rw, o4, slvn)
        }
        if (tid < N1) {
#pragma unroll
            for (int row = 0; row < 16; ++row) dpb_acc += sDL[row * DLP + tid];
        }
        if (a.rowDU) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = q ? 512 + (tid & 255) : tid;
                const int row = idx / 48;
                if ((q == 0 || tid < 256) && r0 + row < R) {
                    const int s = slv[q];
                    a.rowDU[((size_t)b * R + r0) * 48 + idx] = s >= 0 ? sDL[row * DLP + s] * a.inv_ens : 0.f;
                }
            }
        }
        slv[0] = slvn[0]; slv[1] = slvn[1];
        // dO^T[d][row] = sum_n PK[n][d] dl[row][n]   (D: lane holds d = 4 hi + i of row lo)
        f32x4 dot = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
#pragma unroll
            for (int pr = 0; pr < (NT + 1) / 2; ++pr) {
                const int t0 = 2 * pr, t1 = 2 * pr + 1;
                const float4 d0 = *reinterpret_cast<const float4*>(sDL + lo * DLP + 16 * t0 + 4 * hi);
                const float4 d1 = t1 < NT ? *reinterpret_cast<const float4*>(sDL + lo * DLP + 16 * (t1 < NT ? t1 : t0) + 4 * hi)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
                const u32x4 aop = {pk_bf16(pk[t0][0], pk[t0][1]), pk_bf16(pk[t0][2], pk[t0][3]),
                                   t1 < NT ? pk_bf16(pk[t1 < NT ? t1 : t0][0], pk[t1 < NT ? t1 : t0][1]) : 0u,
                                   t1 < NT ? pk_bf16(pk[t1 < NT ? t1 : t0][2], pk[t1 < NT ? t1 : t0][3]) : 0u};
                const u32x4 bop = {pk_bf16(d0.x, d0.y), pk_bf16(d0.z, d0.w), pk_bf16(d1.x, d1.y), pk_bf16(d1.z, d1.w)};
                dot = mfma_bf(aop, bop, dot);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 d4 = *reinterpret_cast<const float4*>(sDL + lo * DLP + 16 * nt + 4 * hi);
                dot = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[nt][0], d4.x, dot, 0, 0, 0);
                dot = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[nt][1], d4.y, dot, 0, 0, 0);
                dot = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[nt][2], d4.z, dot, 0, 0, 0);
                dot = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[nt][3], d4.w, dot, 0, 0, 0);
            }
        }
        *reinterpret_cast<float4*>(sDO + lo * OP + h * 16 + 4 * hi) = make_float4(dot[0], dot[1], dot[2], dot[3]);
        // dPK[n][d] += sum_row dl[row][n] O[row][d]   (D: lane holds node 16 nt + 4 hi + i, d = lo)
        // (k-slot (step j, group hi) stands for row 4 hi + j in both operands: a lane's four steps of a chunk are contiguous in
        // the transposed tile)
        {
            float ov[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = sO[(4 * hi + j) * OP + h * 16 + lo];
            const u32x4 ob = {pk_bf16(ov[0], ov[1]), pk_bf16(ov[2], ov[3]), 0u, 0u};          // (BF)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 dT = *reinterpret_cast<const float4*>(sDLT + (16 * nt + lo) * TPD + 4 * hi);
                if constexpr (BF) {
                    dpk[nt] = mfma_bf(u32x4{pk_bf16(dT.x, dT.y), pk_bf16(dT.z, dT.w), 0u, 0u}, ob, dpk[nt]);
                } else {
                    dpk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT.x, ov[0], dpk[nt], 0, 0, 0);
                    dpk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT.y, ov[1], dpk[nt], 0, 0, 0);
                    dpk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT.z, ov[2], dpk[nt], 0, 0, 0);
                    dpk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT.w, ov[3], dpk[nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (r0 + srow < R)
            *reinterpret_cast<float4*>(a.dO + ((size_t)b * R + r0 + srow) * ELG_E + 4 * sl) =
                *reinterpret_cast<const float4*>(sDO + srow * OP + 4 * sl);
    }
#undef PB_LOAD
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = 16 * nt + 4 * hi + i;
            if (n < N1) atomicAdd(a.dPK + ((size_t)b * N1 + n) * ELG_E + h * 16 + lo, dpk[nt][i]);
        }
    if (tid < N1) atomicAdd(a.dpb + (size_t)b * N1 + tid, dpb_acc);
}


// =============================================================================================
// 128 < N1 <= 1024: the same backward over the rows the streaming rollout kernel saved, as batched f32 MFMA GEMMs
// (csrc/elg_gemm.hip) around three row kernels (one wavefront per row, lanes over the nodes).
// =============================================================================================
// d s[n] = w (c_sel [n = a] - p c[n]) in place over the saved p c row; d u_slot for the local policy; column sums -> d pb
__global__ __launch_bounds__(256) void rows_dl_kernel(const PtrBwd a, float* __restrict__ PCrw, int rows_per_block) {
    extern __shared__ float sAcc[];                                       // [N1] column sums of the block's rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, N1 = a.N1;
    const int R = a.T * a.M;
    for (int i = threadIdx.x; i < N1; i += 256) sAcc[i] = 0.f;
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    for (int r = r0 + wave; r < r1; r += 4) {
        const float4 rw = a.rowW[(size_t)b * R + r];
        const float w = rw.x, wc = rw.y;
        const int sel = f2i(rw.z);
        float* row = PCrw + ((size_t)b * a.Rcap + r) * N1;
        if (a.rowDU && lane < 48) {
            const int s = a.Slot[((size_t)b * a.Rcap + r) * 48 + lane];
            float du = 0.f;
            if (s >= 0 && w != 0.f) du = (s == sel ? wc : 0.f) - w * row[s];
            a.rowDU[((size_t)b * R + r) * 48 + lane] = du * a.inv_ens;
        }
        __builtin_amdgcn_wave_barrier();                                    // the slot reads above precede the in-place stores below
        if (w == 0.f) {                                                     // not a decoded row: exact zeros
            for (int n = lane; n < N1; n += 64) row[n] = 0.f;
            continue;
        }
        for (int n = lane; n < N1; n += 64) {
            const float v = (n == sel ? wc : 0.f) - w * row[n];
            row[n] = v;
            atomicAdd(sAcc + n, v);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N1; i += 256) {
        const float v = sAcc[i];
        if (v != 0.f) atomicAdd(a.dpb + (size_t)b * N1 + i, v);
    }
}

// <dO_h, O_h> per (row, head)                                                              grid (R / 16, B): 16 lanes per row
__global__ __launch_bounds__(256) void rows_doto_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                        float* __restrict__ doto, int R, long long Rdo, long long Rcap) {
    const int b = blockIdx.y, r = blockIdx.x * 16 + (threadIdx.x >> 4), c8 = (threadIdx.x & 15) * 8;
    if (r >= R) return;
    const float4 a0 = *reinterpret_cast<const float4*>(dO + ((size_t)b * Rdo + r) * ELG_E + c8);
    const float4 a1 = *reinterpret_cast<const float4*>(dO + ((size_t)b * Rdo + r) * ELG_E + c8 + 4);
    const float4 o0 = *reinterpret_cast<const float4*>(O + ((size_t)b * Rcap + r) * ELG_E + c8);
    const float4 o1 = *reinterpret_cast<const float4*>(O + ((size_t)b * Rcap + r) * ELG_E + c8 + 4);
    float d = a0.x * o0.x + a0.y * o0.y + a0.z * o0.z + a0.w * o0.w + a1.x * o1.x + a1.y * o1.y + a1.z * o1.z + a1.w * o1.w;
    d += dpp<0xB1>(d);                                    // the head's two 8-channel halves sit in neighbouring lanes
    if (!(threadIdx.x & 1)) doto[((size_t)b * R + r) * 8 + (threadIdx.x & 15) / 2] = d;
}

// The two (rows x nodes) products with a 16-deep contraction -- the glimpse scores S_h = q_h K_h^T and dA_h = dO_h V_h^T -- fused
// with their element-wise consumers, one pass over the (8, R, N1) buffer each instead of a GEMM + a read-modify-write pass:
//   SCORE:  a[r][n]  = closed(r, n) ? 0 : exp2(S[r][n] log2(e) / 4 - lse[r])                      (writes a)
//   !SCORE: dS[r][n] = a[r][n] (dA[r][n] - <dO_h, O_h>[r]) / 4                                     (reads a, writes dS)
// grid (rows / 64, B * 8), 4 waves x 16 rows; the table operand (K_h or V_h: N1 x 16) is staged once per workgroup in LDS in
// MFMA-fragment order ([chunk][lane][4]: node 16 c + lo, channels 4 kk + hi); v_mfma_f32_16x16x4_f32 (exact f32 products).
template <bool SCORE>
__global__ __launch_bounds__(256) void rows_tile_kernel(const float* __restrict__ X, long long x_rows, const float* __restrict__ Tab,
                                                        const unsigned long long* __restrict__ mask, const float* __restrict__ lse,
                                                        const float* __restrict__ doto, const float* __restrict__ Ain,
                                                        float* __restrict__ Out, int R, long long Rcap, int N1, int W,
                                                        const float4* __restrict__ rowW, long long w_rows) {
    extern __shared__ __attribute__((aligned(16))) float sTab[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 15, hi = lane >> 4;
    const int bh = blockIdx.y, b = bh >> 3, h = bh & 7;
    const int NTc = (N1 + 15) >> 4;
    for (int i = threadIdx.x; i < NTc * 64; i += 256) {
        const int c = i >> 6, l = i & 63, n = 16 * c + (l & 15), k = l >> 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N1) {
            const float* t = Tab + ((size_t)b * N1 + n) * ELG_E + h * 16 + k;
            v = make_float4(t[0], t[4], t[8], t[12]);
        }
        *reinterpret_cast<float4*>(sTab + (size_t)i * 4) = v;
    }
    __syncthreads();
    const int r0 = blockIdx.x * 64 + 16 * wave;
    if (r0 >= R) return;
    const int rleft = R - 1 - r0;
    // A operand: row r0 + lo, channels 4 kk + hi (rows past R clamped: their outputs are not stored)
    const float* xr = X + ((size_t)b * x_rows + r0 + min(lo, rleft)) * ELG_E + h * 16 + hi;
    const float xa[4] = {xr[0], xr[4], xr[8], xr[12]};
    float rs[4];                                   // per D row 4 hi + v: lse (SCORE) or <dO, O> (!SCORE)
    size_t mrow[4];
    bool dead[4];                                  // rows the rollout did not decode at this step (weight exactly 0): the forward wrote
                                                   // neither q, lse nor the mask words for them -- whatever an earlier batch left there
                                                   // must not reach exp2 (an overflow would turn 0 * inf into NaN downstream): a = 0
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int rv = r0 + min(4 * hi + v, rleft);
        mrow[v] = ((size_t)b * Rcap + rv) * W;
        rs[v] = SCORE ? lse[((size_t)b * Rcap + rv) * 8 + h] : doto[((size_t)b * R + rv) * 8 + h];
        dead[v] = SCORE && rowW && rowW[(size_t)b * w_rows + rv].x == 0.f;
    }
    const float cs = 0.25f * 1.4426950408889634f;
    // 64 nodes (four chunks) at a time through a per-wave LDS tile [16 rows][68]: the D tiles hold a row's nodes 16 apart per
    // lane, memory wants 64 consecutive floats of a row per wavefront access (rows_tile_kernel moved 1.4 TB/s with 64-byte pieces)
    float* sSt = sTab + (size_t)NTc * 256 + wave * (16 * 68);
    for (int g = 0; 64 * g < N1; ++g) {
        const int nn = 64 * g + lane;                      // this lane's node in the row-wise accesses
        if (!SCORE) {
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr)
                sSt[rr * 68 + lane] = (rr <= rleft && nn < N1) ? Ain[((size_t)bh * R + r0 + rr) * N1 + nn] : 0.f;
            wave_lds_fence();
        }
        unsigned long long mw[4] = {0ull, 0ull, 0ull, 0ull};
        if (SCORE) {
#pragma unroll
            for (int v = 0; v < 4; ++v) mw[v] = mask[mrow[v] + g];
        }
        float res[4][4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int c = 4 * g + cc;
            f32x4 D = {0.f, 0.f, 0.f, 0.f};
            if (c < NTc) {
                const float4 tb = *reinterpret_cast<const float4*>(sTab + ((size_t)c * 64 + lane) * 4);
                D = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0], tb.x, D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[1], tb.y, D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[2], tb.z, D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[3], tb.w, D, 0, 0, 0);
            }
            const int nb_ = 16 * cc + lo;                  // node within the group
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (SCORE) res[cc][v] = (dead[v] || ((mw[v] >> nb_) & 1ull)) ? 0.f : __builtin_amdgcn_exp2f(fmaf(D[v], cs, -rs[v]));
                else res[cc][v] = 0.25f * sSt[(4 * hi + v) * 68 + nb_] * (D[v] - rs[v]);
            }
        }
        if (!SCORE) wave_lds_fence();                      // every lane has read its weights: the tile is reused for the results
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int v = 0; v < 4; ++v) sSt[(4 * hi + v) * 68 + 16 * cc + lo] = res[cc][v];
        wave_lds_fence();
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr)
            if (rr <= rleft && nn < N1) Out[((size_t)bh * R + r0 + rr) * N1 + nn] = sSt[rr * 68 + lane];
        wave_lds_fence();
    }
}

// backward of the query gather: d Q1[prev[r]] += dQ[r], d Q2[first[r]] += dQ[r] (TSP), d wl += load[r] dQ[r] (CVRP)
__global__ __launch_bounds__(256) void rows_qgather_bwd_kernel(const float* __restrict__ dQ, const int* __restrict__ prev,
                                                               const int* __restrict__ first, const float* __restrict__ load,
                                                               float* __restrict__ dQ1, float* __restrict__ dQ2,
                                                               float* __restrict__ dwl, int R, long long Ridx,
                                                               long long Rcap, int N1, int rows_per_block) {
    __shared__ float swl[ELG_E];
    const int c = threadIdx.x & (ELG_E - 1), half = threadIdx.x >> 7;
    const int b = blockIdx.y;
    if (threadIdx.x < ELG_E) swl[threadIdx.x] = 0.f;
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float wl = 0.f;
    for (int r = r0 + half; r < r1; r += 2) {
        const float v = dQ[((size_t)b * R + r) * ELG_E + c];
        if (v != 0.f) {
            atomicAdd(dQ1 + ((size_t)b * N1 + prev[(size_t)b * Ridx + r]) * ELG_E + c, v);
            if (dQ2) atomicAdd(dQ2 + ((size_t)b * N1 + first[(size_t)b * Ridx + r]) * ELG_E + c, v);
            if (dwl) wl = fmaf(load[(size_t)b * Rcap + r], v, wl);
        }
    }
    if (dwl) {
        atomicAdd(swl + c, wl);
        __syncthreads();
        if (threadIdx.x < ELG_E && swl[threadIdx.x] != 0.f) atomicAdd(dwl + threadIdx.x, swl[threadIdx.x]);
    }
}

}  // namespace elg

using namespace elg;

extern "C" int elg_gemm_f32_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                    int transA, int transB, int n_outer, int n_inner, int64_t sA_outer, int64_t sA_inner,
                                    int64_t sB_outer, int64_t sB_inner, int64_t sC_outer, int64_t sC_inner, float alpha,
                                    void* stream);

extern "C" int64_t elg_decoder_bwd_ws_floats(int32_t B, int64_t R, int32_t N1) {
    if (B <= 0 || R <= 0 || N1 <= 128) return 0;
    return (int64_t)B * R * (2 * 8 * (int64_t)N1 + ELG_E);
}

static int decoder_bwd_large(const elg_decoder_bwd_args* p, PtrBwd& a, hipStream_t s) {
    const int B = p->B, N1 = p->N1, W = p->mask_words;
    const long long R = (long long)p->T * p->M, Rcap = p->Rcap;
    const bool tsp = p->problem == ELG_PROBLEM_TSP;
    if (p->T_dev) return fail(ELG_ENOTIMPL, "decoder_bwd: N1 > 128 needs the step count on the host (T_dev = NULL)");
    if (!p->trMask || !p->trLse) return fail(ELG_EINVAL, "decoder_bwd: N1 > 128 needs the mask words and the log-sum-exp rows");
    if (W * 64 < N1 || W > 16) return fail(ELG_EINVAL, "decoder_bwd: mask_words does not cover N1");
    const int64_t per1 = elg_decoder_bwd_ws_floats(1, R, N1);
    if (!p->ws || p->ws_floats < per1) return fail(ELG_EINVAL, "decoder_bwd: scratch smaller than one instance (elg_decoder_bwd_ws_floats(1, R, N1))");
    // rows before the first decode step were never written by the rollout: everything after rows_dl runs over [r_lo, R)
    const long long r_lo = min(R, (long long)p->first_decode_step * p->M), Rl = R - r_lo;
    float* PCw = const_cast<float*>(p->trPC);
    void* st = (void*)s;
    const int rpb = 64;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rows_dl_kernel, dim3((unsigned)((R + rpb - 1) / rpb), B), dim3(256), (size_t)N1 * 4, s, a, PCw, rpb);
    int rc = launch_status("rows_dl");
    if (rc != ELG_OK || Rl == 0) return rc;
    // the scratch holds Bc instances: the batch is walked in chunks of Bc
    const int Bc = (int)min((long long)B, (long long)(p->ws_floats / per1));
    float* Sbuf = p->ws;                                  // (Bc,8,Rl,N1)  s -> a
    float* Dbuf = Sbuf + (size_t)Bc * 8 * R * N1;         // (Bc,8,Rl,N1)  dA -> dS
    float* dQr = Dbuf + (size_t)Bc * 8 * R * N1;          // (Bc,Rl,128)
#define DB_TRY(x) { rc = (x); if (rc != ELG_OK) return rc; }
    for (int b0 = 0; b0 < B; b0 += Bc) {
        const int nb = min(Bc, B - b0);
        const float* dl = PCw + ((size_t)b0 * Rcap + r_lo) * N1;
        const float* Ol = p->trO + ((size_t)b0 * Rcap + r_lo) * ELG_E;
        const float* Ql = p->trQ + ((size_t)b0 * Rcap + r_lo) * ELG_E;
        float* dOl = p->dO + ((size_t)b0 * R + r_lo) * ELG_E;
        const size_t tb = (size_t)b0 * N1 * ELG_E;        // this chunk's tables / table gradients
        // d o = d s PK ;  d PK = d s^T o
        DB_TRY(elg_gemm_f32_batched(dl, p->PK + tb, dOl, (int)Rl, ELG_E, N1, N1, ELG_E, ELG_E, 0, 0, nb, 1, Rcap * N1, 0,
                                    (int64_t)N1 * ELG_E, 0, R * ELG_E, 0, 1.f, st))
        // (the three reductions over the rows -- dPK, dK, dV -- split the rows over enough workgroups to fill the chip and
        // accumulate into the caller-zeroed outputs)
        const int mt_ = (N1 + 63) / 64;
        const int sk1 = (int)max(1LL, min(16LL, min(Rl / 512, 2048LL / ((long long)mt_ * 2 * nb))));
        const int sk8 = (int)max(1LL, min(16LL, min(Rl / 512, 2048LL / ((long long)mt_ * 8 * nb))));
        DB_TRY(gemm_f32_batched_splitk(dl, Ol, p->dPK + tb, N1, ELG_E, (int)Rl, N1, ELG_E, ELG_E, 1, 0, nb, 1, Rcap * N1, 0, Rcap * ELG_E, 0,
                                       (long)N1 * ELG_E, 0, 1.f, sk1, st))
        if (p->tables_frozen) continue;
        // a_h = masked softmax weights from q_h K_h^T, the mask words and the saved normaliser (one pass)
        const size_t tab_lds = ((size_t)((N1 + 15) / 16) * 64 * 4 + 4 * 16 * 68) * sizeof(float);
        const dim3 tgrid((unsigned)((Rl + 63) / 64), nb * 8);
        static DynLds optin_s, optin_d;                    // table (<= 64 KB at 1024 nodes) + the four staging tiles
        if (!optin_s.opt_in(reinterpret_cast<const void*>(rows_tile_kernel<true>), 96 * 1024) ||
            !optin_d.opt_in(reinterpret_cast<const void*>(rows_tile_kernel<false>), 96 * 1024))
            return fail(ELG_ELAUNCH, "decoder_bwd: hipFuncSetAttribute failed");
        hipLaunchKernelGGL((rows_tile_kernel<true>), tgrid, dim3(256), tab_lds, s, Ql, Rcap, p->Kmat + tb,
                           reinterpret_cast<const unsigned long long*>(p->trMask) + ((size_t)b0 * Rcap + r_lo) * W,
                           p->trLse + ((size_t)b0 * Rcap + r_lo) * 8, nullptr, nullptr, Sbuf, (int)Rl, Rcap, N1, W,
                           reinterpret_cast<const float4*>(p->rowW) + (size_t)b0 * R + r_lo, R);
        DB_TRY(launch_status("rows_tile<score>"))
        // dS_h = a_h (dO_h V_h^T - <dO_h, O_h>) / 4 (one pass; <dO, O> per (row, head) first, in the place of dQ)
        hipLaunchKernelGGL(rows_doto_kernel, dim3((unsigned)((Rl + 15) / 16), nb), dim3(256), 0, s, dOl, Ol, dQr, (int)Rl, R, Rcap);
        hipLaunchKernelGGL((rows_tile_kernel<false>), tgrid, dim3(256), tab_lds, s, dOl, R, p->Vmat + tb, nullptr, nullptr, dQr, Sbuf,
                           Dbuf, (int)Rl, Rcap, N1, W, nullptr, 0LL);
        DB_TRY(launch_status("rows_tile<dscore>"))
        // d q_h = dS_h K_h ;  d K_h = dS_h^T q_h ;  d V_h = a_h^T dO_h
        DB_TRY(elg_gemm_f32_batched(Dbuf, p->Kmat + tb, dQr, (int)Rl, 16, N1, N1, ELG_E, ELG_E, 0, 0, nb, 8, 8 * Rl * N1, Rl * N1,
                                    (int64_t)N1 * ELG_E, 16, Rl * ELG_E, 16, 1.f, st))
        DB_TRY(gemm_f32_batched_splitk(Dbuf, Ql, p->dK + tb, N1, 16, (int)Rl, N1, ELG_E, ELG_E, 1, 0, nb, 8, 8 * Rl * N1, Rl * N1,
                                       Rcap * ELG_E, 16, (long)N1 * ELG_E, 16, 1.f, sk8, st))
        DB_TRY(gemm_f32_batched_splitk(Sbuf, dOl, p->dV + tb, N1, 16, (int)Rl, N1, ELG_E, ELG_E, 1, 0, nb, 8, 8 * Rl * N1, Rl * N1,
                                       R * ELG_E, 16, (long)N1 * ELG_E, 16, 1.f, sk8, st))
        hipLaunchKernelGGL(rows_qgather_bwd_kernel, dim3((unsigned)((Rl + rpb - 1) / rpb), nb), dim3(256), 0, s, dQr,
                           p->idx_prev + (size_t)b0 * R + r_lo, tsp ? p->idx_first + (size_t)b0 * R + r_lo : nullptr,
                           tsp ? nullptr : p->trLoad + (size_t)b0 * Rcap + r_lo, p->dQ1 + tb, tsp ? p->dQ2 + tb : nullptr,
                           tsp ? nullptr : p->dwl, (int)Rl, R, Rcap, N1, rpb);
        DB_TRY(launch_status("rows_qgather_bwd"))
    }
#undef DB_TRY
    return ELG_OK;
}

extern "C" int elg_decoder_bwd(const elg_decoder_bwd_args* p, void* stream) {
    if (!p) return fail(ELG_EINVAL, "decoder_bwd: null args");
    const int B = p->B, M = p->M, N1 = p->N1, T = p->T;
    if (B <= 0 || M <= 0 || T <= 0 || N1 < 4) return fail(ELG_EINVAL, "decoder_bwd: bad sizes");
    if (N1 > 1024) return fail(ELG_ENOTIMPL, "decoder_bwd: N1 > 1024 not built");
    const long long R = (long long)T * M;
    if (p->Rcap < R || p->Tcap_actions < T) return fail(ELG_EINVAL, "decoder_bwd: row capacity smaller than T*M");
    if (!p->gprob || !p->pval || !p->tlen || !p->actions || !p->trPC || !p->trCsel || !p->trQ || !p->trO || !p->Kmat ||
        !p->Vmat || !p->PK || !p->dK || !p->dV || !p->dPK || !p->dpb || !p->dQ1 || !p->dO || !p->idx_prev || !p->rowW)
        return fail(ELG_EINVAL, "decoder_bwd: null buffer");
    if (!p->trA && !p->trMask) return fail(ELG_EINVAL, "decoder_bwd: neither glimpse weights nor mask rows saved");
    if (p->rowDU && !p->trSlot) return fail(ELG_EINVAL, "decoder_bwd: rowDU needs the slot rows");
    const bool tsp = p->problem == ELG_PROBLEM_TSP;
    if (tsp && (!p->dQ2 || !p->idx_first)) return fail(ELG_EINVAL, "decoder_bwd: TSP needs dQ2 / idx_first");
    if (!tsp && (!p->dwl || !p->trLoad)) return fail(ELG_EINVAL, "decoder_bwd: CVRP needs dwl / the saved loads");
    hipStream_t s = (hipStream_t)stream;
    PtrBwd a{};
    a.gprob = p->gprob; a.pval = p->pval; a.tlen = p->tlen; a.actions = p->actions; a.PC = p->trPC; a.Csel = p->trCsel;
    a.Slot = p->trSlot; a.rowO = p->trO; a.PK = p->PK; a.dO = p->dO; a.dPK = p->dPK; a.dpb = p->dpb; a.rowDU = p->rowDU;
    a.idx_prev = p->idx_prev; a.idx_first = tsp ? p->idx_first : nullptr; a.rowW = reinterpret_cast<float4*>(p->rowW);
    a.B = B; a.T = T; a.M = M; a.N1 = N1; a.Tcap_act = p->Tcap_actions; a.t0 = p->first_decode_step; a.Rcap = p->Rcap;
    a.inv_ens = p->inv_ens;
    a.T_dev = p->T_dev; a.gT = p->T_dev ? p->gprob_T : T;
    if (p->T_dev && p->gprob_T < T) return fail(ELG_EINVAL, "decoder_bwd: gprob_T smaller than T");
    a.splits = (int)max(1LL, min(16LL, min((R + 15) / 16, (long long)((512 + B - 1) / B))));
    const int nt = (N1 + 15) / 16;
    (void)hipGetLastError();
    hipLaunchKernelGGL(row_weights_kernel, dim3((unsigned)(((long long)B * R + 255) / 256)), dim3(256), 0, s, a);
    int rc = launch_status("row_weights");
    if (rc != ELG_OK) return rc;
    if (N1 > 128) return decoder_bwd_large(p, a, s);
    dim3 grid(a.splits, B), block(512);
    if (p->mfma_mode == 3) {            // the backward of a bf16 rollout: the pointer's two products on bf16 operands too
        if (nt <= 2) hipLaunchKernelGGL((pointer_bwd_kernel<2, true>), grid, block, 0, s, a);
        else if (nt <= 4) hipLaunchKernelGGL((pointer_bwd_kernel<4, true>), grid, block, 0, s, a);
        else if (nt <= 7) hipLaunchKernelGGL((pointer_bwd_kernel<7, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((pointer_bwd_kernel<8, true>), grid, block, 0, s, a);
    } else {
        if (nt <= 2) hipLaunchKernelGGL((pointer_bwd_kernel<2, false>), grid, block, 0, s, a);
        else if (nt <= 4) hipLaunchKernelGGL((pointer_bwd_kernel<4, false>), grid, block, 0, s, a);
        else if (nt <= 7) hipLaunchKernelGGL((pointer_bwd_kernel<7, false>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((pointer_bwd_kernel<8, false>), grid, block, 0, s, a);
    }
    rc = launch_status("pointer_bwd");
    if (rc != ELG_OK) return rc;
    if (p->tables_frozen) return ELG_OK;          // nothing upstream of the pointer scores needs a gradient
    GlimpseSeg seg{};
    seg.idx_prev = p->idx_prev; seg.idx_first = tsp ? p->idx_first : nullptr; seg.load = tsp ? nullptr : p->trLoad;
    seg.dQ1 = p->dQ1; seg.dQ2 = tsp ? p->dQ2 : nullptr; seg.dwl = tsp ? nullptr : p->dwl; seg.load_rows = p->Rcap;
    seg.accumulate = 1;
    seg.lse = p->trMask ? p->trLse : nullptr;
    seg.T_dev = p->T_dev; seg.M = M; seg.tlen = p->tlen; seg.t0 = p->first_decode_step;
    if (p->mfma_mode < 0 || p->mfma_mode > 3) return fail(ELG_EINVAL, "decoder_bwd: mfma_mode 0 .. 3");
    seg.mfma_mode = p->mfma_mode;
    const int splits = max(1, min(8, 1024 / (B * 8)));
    return glimpse_bwd_launch(p->trMask ? nullptr : p->trA, reinterpret_cast<const unsigned long long*>(p->trMask), p->dO, p->trO,
                              p->trQ, p->Kmat, p->Vmat, nullptr, p->dK, p->dV, B, (int)R, N1, p->Rcap, p->Rcap, p->Rcap, splits, seg, s);
}
