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
template <int NT>
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
#define PB_LOAD(TILE, PCV, RWV, OV)                                                                        \
    {                                                                                                      \
        const int rr_ = min(((TILE) << 4) + srow, R - 1);                                                  \
        const float* p_ = a.PC + (brow + rr_) * N1;                                                        \
        _Pragma("unroll") for (int k = 0; k < NPT; ++k) PCV[k] = p_[min(sl + 32 * k, N1 - 1)];             \
        RWV = a.rowW[(size_t)b * R + rr_];                                                                 \
        if (((TILE) << 4) + srow >= R) { RWV.x = 0.f; RWV.y = 0.f; }                                       \
        OV = *reinterpret_cast<const float4*>(a.rowO + (brow + rr_) * ELG_E + 4 * sl);                     \
    }
    PB_LOAD(t_lo, pc, rw, o4)
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
        {
            const int tn = min(tile + 1, t_hi - 1);                     // the last prefetch re-reads its own tile, unused
            PB_LOAD(tn, pc, rw, o4)
        }
        if (tid < N1) {
#pragma unroll
            for (int row = 0; row < 16; ++row) dpb_acc += sDL[row * DLP + tid];
        }
        if (a.rowDU)
            for (int idx = tid; idx < 16 * 48; idx += 512) {
                const int row = idx / 48, j = idx - row * 48;
                if (r0 + row < R) {
                    const int s = a.Slot[(brow + r0 + row) * 48 + j];
                    a.rowDU[((size_t)b * R + r0 + row) * 48 + j] = s >= 0 ? sDL[row * DLP + s] * a.inv_ens : 0.f;
                }
            }
        // dO^T[d][row] = sum_n PK[n][d] dl[row][n]   (D: lane holds d = 4 hi + i of row lo)
        f32x4 dot = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float4 d4 = *reinterpret_cast<const float4*>(sDL + lo * DLP + 16 * nt + 4 * hi);
            dot = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[nt][0], d4.x, dot, 0, 0, 0);
            dot = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[nt][1], d4.y, dot, 0, 0, 0);
            dot = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[nt][2], d4.z, dot, 0, 0, 0);
            dot = __builtin_amdgcn_mfma_f32_16x16x4f32(pk[nt][3], d4.w, dot, 0, 0, 0);
        }
        *reinterpret_cast<float4*>(sDO + lo * OP + h * 16 + 4 * hi) = make_float4(dot[0], dot[1], dot[2], dot[3]);
        // dPK[n][d] += sum_row dl[row][n] O[row][d]   (D: lane holds node 16 nt + 4 hi + i, d = lo)
        // (k-slot (step j, group hi) stands for row 4 hi + j in both operands: a lane's four steps of a chunk are contiguous in
        // the transposed tile)
        {
            float ov[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = sO[(4 * hi + j) * OP + h * 16 + lo];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 dT = *reinterpret_cast<const float4*>(sDLT + (16 * nt + lo) * TPD + 4 * hi);
                dpk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT.x, ov[0], dpk[nt], 0, 0, 0);
                dpk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT.y, ov[1], dpk[nt], 0, 0, 0);
                dpk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT.z, ov[2], dpk[nt], 0, 0, 0);
                dpk[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dT.w, ov[3], dpk[nt], 0, 0, 0);
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

}  // namespace elg

using namespace elg;

extern "C" int elg_decoder_bwd(const elg_decoder_bwd_args* p, void* stream) {
    if (!p) return fail(ELG_EINVAL, "decoder_bwd: null args");
    const int B = p->B, M = p->M, N1 = p->N1, T = p->T;
    if (B <= 0 || M <= 0 || T <= 0 || N1 < 4) return fail(ELG_EINVAL, "decoder_bwd: bad sizes");
    if (N1 > 128) return fail(ELG_ENOTIMPL, "decoder_bwd: N1 > 128 not built (use elg_rollout_bwd)");
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
    dim3 grid(a.splits, B), block(512);
    if (nt <= 2) hipLaunchKernelGGL(pointer_bwd_kernel<2>, grid, block, 0, s, a);
    else if (nt <= 4) hipLaunchKernelGGL(pointer_bwd_kernel<4>, grid, block, 0, s, a);
    else if (nt <= 7) hipLaunchKernelGGL(pointer_bwd_kernel<7>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(pointer_bwd_kernel<8>, grid, block, 0, s, a);
    rc = launch_status("pointer_bwd");
    if (rc != ELG_OK) return rc;
    if (p->tables_frozen) return ELG_OK;          // nothing upstream of the pointer scores needs a gradient
    GlimpseSeg seg{};
    seg.idx_prev = p->idx_prev; seg.idx_first = tsp ? p->idx_first : nullptr; seg.load = tsp ? nullptr : p->trLoad;
    seg.dQ1 = p->dQ1; seg.dQ2 = tsp ? p->dQ2 : nullptr; seg.dwl = tsp ? nullptr : p->dwl; seg.load_rows = p->Rcap;
    seg.accumulate = 1;
    seg.lse = p->trMask ? p->trLse : nullptr;
    seg.T_dev = p->T_dev; seg.M = M; seg.tlen = p->tlen; seg.t0 = p->first_decode_step;
    const int splits = max(1, min(8, 1024 / (B * 8)));
    return glimpse_bwd_launch(p->trMask ? nullptr : p->trA, reinterpret_cast<const unsigned long long*>(p->trMask), p->dO, p->trO,
                              p->trQ, p->Kmat, p->Vmat, nullptr, p->dK, p->dV, B, (int)R, N1, p->Rcap, p->Rcap, p->Rcap, splits, seg, s);
}
