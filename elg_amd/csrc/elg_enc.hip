// The attention encoder and the decoder's per-instance tables (reference CVRP/models.py:199-269 CVRP_Encoder /
// EncoderLayer, :455-503 multi_head_attention, :506-527 AddAndInstanceNormalization, :550-561 FeedForward, :300-308
// CVRP_Decoder.set_kv; TSP/models.py:134-194,231-243), forward AND backward, as hand-written CDNA4 kernels behind
// elg_encoder_fwd / elg_encoder_bwd.  No library GEMM, no framework attention kernel.
//
// Everything dense runs on v_mfma_f32_16x16x4_f32 (exact f32, the fmaf-chain numerics of the reference's fp32
// path).  One GEMM kernel serves every x W^T / dY W product of the layer:
//   * the activation operand and the nn.Linear weight are both k-contiguous in memory, and an MFMA does not care
//     in which order the contraction index is visited, so lane (i = lane & 15, q = lane >> 4) feeds the four MFMAs of
//     a 16-wide k chunk from ONE 16-byte load per operand (k = 4 q + j at step j) -- no LDS staging, no transposes;
//   * a wavefront owns 16 output channels for ALL rows of its row block (RT tiles of 16 rows).  With the row block =
//     one instance (N1 <= 128) the per-(instance, channel) statistics of InstanceNorm1d are a register reduction +
//     two cross-quarter shuffles, so bias + residual + instance norm are the epilogue of the combine / FFN-2 GEMMs;
//   * K can be split over the waves of a workgroup (partial tiles summed through LDS) so that the narrow GEMMs
//     (128 output channels = 8 waves per instance) still put >= 1024 waves on the chip's 1024 matrix cores.
// Self-attention is one workgroup per (instance, head, 64 query rows): S^T = K Q^T tiles whose D registers are directly
// the B operand of O^T = V^T P^T (online softmax over chunks of 128 keys, any N1).  Its backward (N1 <= 128) keeps
// Q, K, V, dO of the (instance, head) in LDS and forms the score tile in both orientations, so that dQ (rows on lanes)
// and dK / dV (keys on lanes) each accumulate in registers of the wave that owns them: no atomics, no transposes.
#include "elg_common.h"
#include "../../include/elg_hip.h"
#include <string>

extern "C" __attribute__((visibility("hidden"))) int elg_gemm_f32_alpha(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                                  int ldb, int ldc, int transA, int transB, int relu, int split_k, float* a_rowsum,
                                  float alpha, void* stream);

namespace elg {
int fail(int code, const std::string& msg);
int launch_status(const char* what);

using f32x4 = __attribute__((ext_vector_type(4))) float;

enum { EPI_STORE = 0, EPI_RELU = 1, EPI_NORM = 2, EPI_ADD = 3, EPI_RELUMASK = 4, EPI_ADDBIAS = 5 };
enum { W_NK = 0, W_KN = 1, W_NK_SCALAR = 2 };

struct EncGemm {
    const float* A; int lda;            // (rows, K) activations
    const float* W[6]; int ldw, wblk, wmode;   // W_NK: W[n / wblk] is (wblk, K) ; W_KN: W[k / wblk] is (wblk, N)
    float* C[6]; int ldc, cblk;         // output column n goes to C[n / cblk][row * ldc + n % cblk]
    const float* bias;                  // (N) or NULL
    const float* R; int ldr;            // residual / mask source, indexed [row][n]
    int N, K, rows_total, blk_rows, blk_stride;
    int epi; float alpha;
    const float* gamma; const float* beta; float* xhat; float* rstd; float eps;   // EPI_NORM
    const float* rv; const float* cv; float alpha2;                               // EPI_ADD: + alpha2 rv[row] cv[n]
    int dbg;
};

// wave-uniform pick from a kernel-argument pointer table (a select chain: a dynamic index would move the whole
// argument struct to scratch memory)
template <typename T>
__device__ __forceinline__ T* pick6(T* const (&p)[6], int i) {
    T* r = p[0];
    r = i == 1 ? p[1] : r;
    r = i == 2 ? p[2] : r;
    r = i == 3 ? p[3] : r;
    r = i == 4 ? p[4] : r;
    r = i == 5 ? p[5] : r;
    return r;
}

// ------------------------------------------------------------------------------------------------------------------
// C[row][n] = epilogue( alpha * sum_k A[row][k] W(n, k) )
// grid (N / (16 NWC), row blocks), 256 threads = NWC column tiles x KS k-splits.
// The row block's activations go through LDS in slabs of 128 k (coalesced 16-byte global loads, pitch 136 floats: the
// fragment reads ds_read_b128 [row lo][k 4 hi ..] are bank-conflict free), shared by the workgroup's column tiles; the
// next slab and its weight fragments are fetched while the current one is multiplied.  Feeding the MFMAs straight from
// global memory (first version) was bound by the vector-memory path: every column tile re-read the whole row block.
constexpr int ENC_KS = 128;         // k per slab
constexpr int ENC_AP = 136;         // LDS pitch of a slab row (floats)

template <int RT, int KS>
__global__ __launch_bounds__(256) void enc_gemm_kernel(const EncGemm g) {
    constexpr int NWC = 4 / KS;
    constexpr int ROWS = RT * 16;
    constexpr int NST = ROWS * 32 / 256;          // 16-byte staging loads per thread and slab
    constexpr int NCH = ENC_KS / 16 / KS;         // 16-k chunks of a slab per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int wc = wave % NWC, ks = wave / NWC;
    const int n0 = (blockIdx.x * NWC + wc) * 16;
    const int row0 = blockIdx.y * g.blk_stride;
    const int nrows = min(g.blk_rows, g.rows_total - row0);
    const int nslab = g.K / ENC_KS;
    const int ncol = n0 + lo;
    const int nwb = __builtin_amdgcn_readfirstlane(n0 / g.wblk);        // 16 | wblk: one block per wave
    const float* wp = nullptr;
    if (g.wmode != W_KN) wp = pick6(g.W, nwb) + (size_t)(ncol - nwb * g.wblk) * g.ldw + 4 * hi;

    // staging addresses: thread -> (row, 16-byte column) of the slab, rows past the block clamped (finite duplicates)
    const float* sp[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int idx = tid + i * 256, row = idx >> 5, c4 = idx & 31;
        sp[i] = g.A + (size_t)(row0 + min(row, nrows - 1)) * g.lda + 4 * c4;
    }
    const int sdst = (tid >> 5) * ENC_AP + 4 * (tid & 31);              // + i * 8 rows

#define ENC_WLOAD(K0, WV)                                                                                         \
    {                                                                                                             \
        _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                         \
            const int kk_ = (K0) + 16 * c;                                                                        \
            if (g.wmode == W_NK) WV[c] = *reinterpret_cast<const float4*>(wp + kk_);                              \
            else if (g.wmode == W_NK_SCALAR) WV[c] = make_float4(wp[kk_], wp[kk_ + 1], wp[kk_ + 2], wp[kk_ + 3]); \
            else {                                                                                                \
                const int kb_ = __builtin_amdgcn_readfirstlane(kk_ / g.wblk), kr_ = kk_ - kb_ * g.wblk + 4 * hi;  \
                const float* p_ = pick6(g.W, kb_) + (size_t)kr_ * g.ldw + ncol;                                   \
                WV[c] = make_float4(p_[0], p_[g.ldw], p_[2 * g.ldw], p_[3 * g.ldw]);                              \
            }                                                                                                     \
        }                                                                                                         \
    }
    f32x4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 w[NCH];
    {
        float4 st[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) st[i] = (g.dbg & 4) ? make_float4(1.f, 1.f, 1.f, 1.f) : *reinterpret_cast<const float4*>(sp[i]);
        ENC_WLOAD(ks * (ENC_KS / KS), w)
#pragma unroll
        for (int i = 0; i < NST; ++i) *reinterpret_cast<float4*>(lds + sdst + i * 8 * ENC_AP) = st[i];
    }
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
        const float* sA = lds + (sl & 1) * (ROWS * ENC_AP) + lo * ENC_AP + ks * (ENC_KS / KS) + 4 * hi;
        const bool more = sl + 1 < nslab;               // uniform
        // unconditional (the last iteration re-reads its own slab): a conditional prefetch lands in scratch memory
        float4 st[NST], wn[NCH];
        const int sn = min(sl + 1, nslab - 1);
#pragma unroll
        for (int i = 0; i < NST; ++i) st[i] = *reinterpret_cast<const float4*>(sp[i] + sn * ENC_KS);
        ENC_WLOAD(sn * ENC_KS + ks * (ENC_KS / KS), wn)
        float4 a[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const float4*>(sA + t * 16 * ENC_AP);
#pragma unroll
        for (int c = 0; c < ((g.dbg & 2) ? 1 : NCH); ++c) {
            float4 an[RT];
            if (c + 1 < NCH) {
#pragma unroll
                for (int t = 0; t < RT; ++t) an[t] = *reinterpret_cast<const float4*>(sA + t * 16 * ENC_AP + 16 * (c + 1));
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, w[c].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, w[c].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, w[c].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, w[c].w, acc[t], 0, 0, 0);
            if (c + 1 < NCH) {
#pragma unroll
                for (int t = 0; t < RT; ++t) a[t] = an[t];
            }
        }
        if (more) {
            float* d = lds + ((sl + 1) & 1) * (ROWS * ENC_AP) + sdst;
#pragma unroll
            for (int i = 0; i < NST; ++i) *reinterpret_cast<float4*>(d + i * 8 * ENC_AP) = st[i];
#pragma unroll
            for (int c = 0; c < NCH; ++c) w[c] = wn[c];
        }
        __syncthreads();
    }
#undef ENC_WLOAD
    if (KS > 1) {
        float* red = lds;                                 // the slabs are dead: reuse
        if (ks > 0) {
            float* my = red + (size_t)((ks - 1) * NWC + wc) * RT * 4 * 64;
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) my[(t * 4 + i) * 64 + lane] = acc[t][i];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int s = 0; s < KS - 1; ++s) {
            const float* o = red + (size_t)(s * NWC + wc) * RT * 4 * 64;
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t][i] += o[(t * 4 + i) * 64 + lane];
        }
    }
    if ((g.dbg & 1) && acc[0][0] != 12345.f) return;
    // D layout: acc[t][i] = C[row 16 t + 4 hi + i][column n0 + lo]
    const int ncb = __builtin_amdgcn_readfirstlane(n0 / g.cblk);
    float* cp = pick6(g.C, ncb) + (ncol - ncb * g.cblk);
    const float bv = g.bias ? g.bias[ncol] : 0.f;
    if (g.epi == EPI_NORM) {
        // s = residual + x W^T + b over the instance's rows; InstanceNorm1d statistics per channel (= per lane column)
        float s[RT][4];
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * t + 4 * hi + i;
                const bool ok = r < nrows;
                const float res = g.R[(size_t)(row0 + min(r, nrows - 1)) * g.ldr + ncol];
                s[t][i] = ok ? (acc[t][i] + bv) + res : 0.f;
                sum += s[t][i];
            }
        sum += shfl_xor(sum, 16);
        sum += shfl_xor(sum, 32);
        const float mean = sum / (float)nrows;
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * t + 4 * hi + i;
                const float d = (r < nrows) ? s[t][i] - mean : 0.f;
                s[t][i] = d;
                sq = fmaf(d, d, sq);
            }
        sq += shfl_xor(sq, 16);
        sq += shfl_xor(sq, 32);
        const float rs = 1.0f / sqrtf(sq / (float)nrows + g.eps);
        const float ga = g.gamma[ncol], be = g.beta[ncol];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * t + 4 * hi + i;
                if (r < nrows) {
                    const float xh = s[t][i] * rs;
                    cp[(size_t)(row0 + r) * g.ldc] = fmaf(xh, ga, be);
                    if (g.xhat) g.xhat[(size_t)(row0 + r) * g.N + ncol] = xh;
                }
            }
        if (hi == 0 && g.rstd) g.rstd[(size_t)blockIdx.y * g.N + ncol] = rs;
        return;
    }
    const float cvv = (g.epi == EPI_ADD && g.cv) ? g.cv[ncol] * g.alpha2 : 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * t + 4 * hi + i;
            if (r < nrows) {
                const size_t grow = (size_t)(row0 + r);
                float v = acc[t][i] * g.alpha;
                if (g.epi == EPI_STORE) v += bv;
                else if (g.epi == EPI_RELU) v = fmaxf(v + bv, 0.f);
                else if (g.epi == EPI_ADD) {
                    v += g.R[grow * g.ldr + ncol];
                    if (g.cv) v = fmaf(g.rv[grow], cvv, v);
                } else if (g.epi == EPI_RELUMASK) v = (g.R[grow * g.ldr + ncol] > 0.f) ? v : 0.f;
                else if (g.epi == EPI_ADDBIAS) v = (v + bv) + g.R[grow * g.ldr + ncol];
                cp[grow * g.ldc] = v;
            }
        }
}

template <int RT, int KS>
static int launch_gemm_t(const EncGemm& g, dim3 grid, hipStream_t s) {
    const int nbuf = g.K > ENC_KS ? 2 : 1;
    const size_t lds = (size_t)nbuf * RT * 16 * ENC_AP * sizeof(float);
    auto kern = enc_gemm_kernel<RT, KS>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipGetLastError();
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(2 * RT * 16 * ENC_AP * sizeof(float))) != hipSuccess)
            return fail(ELG_ELAUNCH, "encoder gemm: hipFuncSetAttribute failed");
        attr_done = true;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);
    return launch_status("enc_gemm");
}

static int launch_gemm(const EncGemm& g_in, hipStream_t s) {
    EncGemm g = g_in;
    { const char* e = getenv("ELG_ENC_DBG"); g.dbg = e ? atoi(e) : 0; }
    if ((g.N & 15) || g.N <= 0) return fail(ELG_EINVAL, "encoder gemm: N must be a multiple of 16");
    const int nblk = (g.rows_total + g.blk_stride - 1) / g.blk_stride;
    const int ctiles = g.N / 16;
    if ((g.wblk % 16) || (g.cblk % 16) || (g.K % ENC_KS)) return fail(ELG_EINVAL, "encoder gemm: K must be a multiple of 128, weight / output blocks of 16");
    // K split over the waves of a workgroup: enough waves for the chip's 1024 SIMDs
    int ks = 1;
    if ((long)ctiles * nblk < 1024) ks = 2;
    while (ks < 4 && (ctiles % (4 / ks))) ks *= 2;
    dim3 grid(ctiles / (4 / ks), nblk);
#define ENC_GO(RT)                                                \
    {                                                             \
        if (ks == 1) return launch_gemm_t<RT, 1>(g, grid, s);     \
        if (ks == 2) return launch_gemm_t<RT, 2>(g, grid, s);     \
        return launch_gemm_t<RT, 4>(g, grid, s);                  \
    }
    if (g.blk_rows <= 32) ENC_GO(2)
    if (g.blk_rows <= 64) ENC_GO(4)
    if (g.blk_rows <= 112) ENC_GO(7)
    if (g.blk_rows <= 128) ENC_GO(8)
#undef ENC_GO
    return fail(ELG_EINVAL, "encoder gemm: row block > 128");
}

// ------------------------------------------------------------------------------------------------------------------
// input embedding (models.py:206-217: depot Linear(2,128), customers Linear(3,128) on (x, y, demand); TSP: Linear(2,128))
__global__ __launch_bounds__(256) void enc_embed_kernel(const float* __restrict__ xy, const float* __restrict__ demand,
                                                        const float* __restrict__ Wd, const float* __restrict__ bd,
                                                        const float* __restrict__ Wn, const float* __restrict__ bn,
                                                        float* __restrict__ X, int N1, long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx & (ELG_E - 1));
    const long row = idx >> 7;
    const int n = (int)(row % N1);
    const float x = xy[row * 2], y = xy[row * 2 + 1];
    float v;
    if (Wd && n == 0) v = fmaf(Wd[c * 2 + 1], y, fmaf(Wd[c * 2], x, 0.f)) + bd[c];
    else if (demand) v = fmaf(Wn[c * 3 + 2], demand[row], fmaf(Wn[c * 3 + 1], y, fmaf(Wn[c * 3], x, 0.f))) + bn[c];
    else v = fmaf(Wn[c * 2 + 1], y, fmaf(Wn[c * 2], x, 0.f)) + bn[c];
    X[idx] = v;
}

// d embedding weights: grid over row chunks, thread = channel; partial sums flushed with one atomic each
__global__ __launch_bounds__(128) void enc_embed_bwd_kernel(const float* __restrict__ xy, const float* __restrict__ demand,
                                                            const float* __restrict__ dX, float* gWd, float* gbd, float* gWn,
                                                            float* gbn, int N1, long rows, int rows_per_block) {
    const int c = threadIdx.x;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float nw0 = 0.f, nw1 = 0.f, nw2 = 0.f, nb = 0.f, dw0 = 0.f, dw1 = 0.f, db = 0.f;
    for (long r = r0; r < r1; ++r) {
        const float gx = dX[r * ELG_E + c];
        const float x = xy[r * 2], y = xy[r * 2 + 1];
        if (gWd && (r % N1) == 0) { dw0 = fmaf(gx, x, dw0); dw1 = fmaf(gx, y, dw1); db += gx; }
        else { nw0 = fmaf(gx, x, nw0); nw1 = fmaf(gx, y, nw1); if (demand) nw2 = fmaf(gx, demand[r], nw2); nb += gx; }
    }
    if (demand) { atomicAdd(gWn + c * 3, nw0); atomicAdd(gWn + c * 3 + 1, nw1); atomicAdd(gWn + c * 3 + 2, nw2); }
    else { atomicAdd(gWn + c * 2, nw0); atomicAdd(gWn + c * 2 + 1, nw1); }
    atomicAdd(gbn + c, nb);
    if (gWd) { atomicAdd(gWd + c * 2, dw0); atomicAdd(gWd + c * 2 + 1, dw1); atomicAdd(gbd + c, db); }
}

// pb[r] = enc[r] . bc / sqrt(E)  (one wave per row) ; wl = Wq_last[:, 128] (CVRP load column)
__global__ __launch_bounds__(256) void enc_pb_wl_kernel(const float* __restrict__ enc, const float* __restrict__ bc,
                                                        const float* __restrict__ Wq_last, float* __restrict__ pb,
                                                        float* __restrict__ wl, long rows, float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && wl && threadIdx.x < ELG_E) wl[threadIdx.x] = Wq_last[threadIdx.x * (ELG_E + 1) + ELG_E];
    if (r >= rows) return;
    const float2 e = *reinterpret_cast<const float2*>(enc + r * ELG_E + 2 * lane);
    const float2 b2 = *reinterpret_cast<const float2*>(bc + 2 * lane);
    const float v = wave_sum(fmaf(e.x, b2.x, e.y * b2.y));
    if (lane == 0) pb[r] = v * scale;
}

// d bc[e] += sum_r enc[r][e] gpb[r] * scale ; d Wq_last[:, 128] += gwl
__global__ __launch_bounds__(128) void enc_fold_small_bwd_kernel(const float* __restrict__ enc, const float* __restrict__ gpb,
                                                                 const float* __restrict__ gwl, float* gbc, float* gWq_last,
                                                                 long rows, int rows_per_block, float scale) {
    const int c = threadIdx.x;
    if (blockIdx.x == 0 && gwl && gWq_last) atomicAdd(gWq_last + c * (ELG_E + 1) + ELG_E, gwl[c]);
    if (!gpb) return;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float acc = 0.f;
    for (long r = r0; r < r1; ++r) acc = fmaf(enc[r * ELG_E + c], gpb[r], acc);
    atomicAdd(gbc + c, acc * scale);
}

// ------------------------------------------------------------------------------------------------------------------
// self-attention forward: softmax(Q_h K_h^T / 4) V_h per (instance, head), heads = channels h*16..h*16+15 of the
// (rows, 384) QKV buffer (Q | K | V).  grid (ceil(N1 / 64), B * 8), one wave per 16 query rows.
__global__ __launch_bounds__(256) void enc_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ O,
                                                           float* __restrict__ lse, int N1) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int b = blockIdx.y >> 3, h = blockIdx.y & 7;
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= N1) return;
    constexpr int LD = 3 * ELG_E;
    const float* base = qkv + (size_t)b * N1 * LD + h * 16;
    const float4 q4 = *reinterpret_cast<const float4*>(base + (size_t)min(row0 + lo, N1 - 1) * LD + 4 * hi);
    const int nkt = (N1 + 15) >> 4;
    float m = ELG_NEG_INF, l = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int kt0 = 0; kt0 < nkt; kt0 += 8) {
        f32x4 S[8];
        float v[8][4];
        float cmax = ELG_NEG_INF;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int kb = 16 * (kt0 + c);
            const float4 k4 = *reinterpret_cast<const float4*>(base + (size_t)min(kb + lo, N1 - 1) * LD + ELG_E + 4 * hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[c][i] = base[(size_t)min(kb + 4 * hi + i, N1 - 1) * LD + 2 * ELG_E + lo];
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.x, q4.x, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.y, q4.y, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.z, q4.z, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(k4.w, q4.w, s, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i] = (kb + 4 * hi + i < N1) ? s[i] * 0.25f : ELG_NEG_INF;     // key 16 kt + 4 hi + i, query row lo
                cmax = fmaxf(cmax, s[i]);
            }
            S[c] = s;
        }
        cmax = fmaxf(cmax, shfl_xor(cmax, 16));
        cmax = fmaxf(cmax, shfl_xor(cmax, 32));
        const float mn = fmaxf(m, cmax);
        const float sc = __expf(m - mn);                 // 0 on the first chunk (m = -inf)
        m = mn;
        l *= sc;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] *= sc;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __expf(S[c][i] - m);
                l += p;
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c][i], p, o, 0, 0, 0);
            }
    }
    l += shfl_xor(l, 16);
    l += shfl_xor(l, 32);
    const float inv = 1.0f / l;
    const int row = row0 + lo;
    if (row < N1) {
        *reinterpret_cast<float4*>(O + ((size_t)b * N1 + row) * ELG_E + h * 16 + 4 * hi) =
            make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
        if (hi == 0 && lse) lse[(size_t)blockIdx.y * N1 + row] = m + __logf(l);
    }
}

// self-attention backward, N1 <= 16 NT <= 128.  grid (B * 8), 4 waves.
template <int NT>
__global__ __launch_bounds__(256) void enc_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dOg,
                                                           const float* __restrict__ Og, const float* __restrict__ lse,
                                                           float* __restrict__ dqkv, int N1) {
    constexpr int P = 20, ROWS = NT * 16, LD = 3 * ELG_E;
    __shared__ __attribute__((aligned(16))) float sQ[ROWS * P], sK[ROWS * P], sV[ROWS * P], sD[ROWS * P];
    __shared__ float sL[ROWS], sDel[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int b = blockIdx.x >> 3, h = blockIdx.x & 7;
    const float* base = qkv + (size_t)b * N1 * LD + h * 16;
    const float* dOb = dOg + (size_t)b * N1 * ELG_E + h * 16;
    const float* Ob = Og + (size_t)b * N1 * ELG_E + h * 16;
    for (int idx = tid; idx < ROWS * 4; idx += 256) {
        const int row = idx >> 2, c4 = (idx & 3) * 4;
        const int rr = min(row, N1 - 1);
        const float mk = row < N1 ? 1.f : 0.f;
        float4 q = *reinterpret_cast<const float4*>(base + (size_t)rr * LD + c4);
        float4 k = *reinterpret_cast<const float4*>(base + (size_t)rr * LD + ELG_E + c4);
        float4 v = *reinterpret_cast<const float4*>(base + (size_t)rr * LD + 2 * ELG_E + c4);
        float4 d = *reinterpret_cast<const float4*>(dOb + (size_t)rr * ELG_E + c4);
        q.x *= mk; q.y *= mk; q.z *= mk; q.w *= mk;
        d.x *= mk; d.y *= mk; d.z *= mk; d.w *= mk;
        *reinterpret_cast<float4*>(sQ + row * P + c4) = q;
        *reinterpret_cast<float4*>(sK + row * P + c4) = k;
        *reinterpret_cast<float4*>(sV + row * P + c4) = v;
        *reinterpret_cast<float4*>(sD + row * P + c4) = d;
    }
    if (tid < ROWS) {
        const int rr = min(tid, N1 - 1);
        float acc = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 16; c4 += 4) {
            const float4 d = *reinterpret_cast<const float4*>(dOb + (size_t)rr * ELG_E + c4);
            const float4 o = *reinterpret_cast<const float4*>(Ob + (size_t)rr * ELG_E + c4);
            acc = dot4(d, o, acc);
        }
        sDel[tid] = tid < N1 ? acc : 0.f;
        sL[tid] = tid < N1 ? lse[(size_t)blockIdx.x * N1 + rr] : __builtin_huge_valf();    // exp(s - inf) = 0
    }
    __syncthreads();
    // ---- rows on lanes: dQ of the wave's row tiles
    for (int rt = wave; rt < NT; rt += 4) {
        const float4 qB = *reinterpret_cast<const float4*>(sQ + (16 * rt + lo) * P + 4 * hi);
        const float4 dB = *reinterpret_cast<const float4*>(sD + (16 * rt + lo) * P + 4 * hi);
        const float lr = sL[16 * rt + lo], del = sDel[16 * rt + lo];
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const float4 kA = *reinterpret_cast<const float4*>(sK + (16 * kt + lo) * P + 4 * hi);
            const float4 vA = *reinterpret_cast<const float4*>(sV + (16 * kt + lo) * P + 4 * hi);
            f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(kA.x, qB.x, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_16x16x4f32(vA.x, dB.x, dP, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(kA.y, qB.y, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_16x16x4f32(vA.y, dB.y, dP, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(kA.z, qB.z, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_16x16x4f32(vA.z, dB.z, dP, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(kA.w, qB.w, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_16x16x4f32(vA.w, dB.w, dP, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = 16 * kt + 4 * hi + i;
                const float p = key < N1 ? __expf(S[i] * 0.25f - lr) : 0.f;
                const float ds = p * (dP[i] - del) * 0.25f;
                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(sK[key * P + lo], ds, dq, 0, 0, 0);
            }
        }
        const int row = 16 * rt + lo;
        if (row < N1)
            *reinterpret_cast<float4*>(dqkv + ((size_t)b * N1 + row) * LD + h * 16 + 4 * hi) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    }
    // ---- keys on lanes: dK, dV of the wave's key tiles
    for (int kt = wave; kt < NT; kt += 4) {
        const float4 kB = *reinterpret_cast<const float4*>(sK + (16 * kt + lo) * P + 4 * hi);
        const float4 vB = *reinterpret_cast<const float4*>(sV + (16 * kt + lo) * P + 4 * hi);
        const bool kok = 16 * kt + lo < N1;
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            const float4 qA = *reinterpret_cast<const float4*>(sQ + (16 * rt + lo) * P + 4 * hi);
            const float4 dA = *reinterpret_cast<const float4*>(sD + (16 * rt + lo) * P + 4 * hi);
            f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA.x, kB.x, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_16x16x4f32(dA.x, vB.x, dP, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA.y, kB.y, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_16x16x4f32(dA.y, vB.y, dP, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA.z, kB.z, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_16x16x4f32(dA.z, vB.z, dP, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_16x16x4f32(qA.w, kB.w, S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_16x16x4f32(dA.w, vB.w, dP, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * rt + 4 * hi + i;
                const float p = kok ? __expf(S[i] * 0.25f - sL[row]) : 0.f;
                const float ds = p * (dP[i] - sDel[row]) * 0.25f;
                dv = __builtin_amdgcn_mfma_f32_16x16x4f32(sD[row * P + lo], p, dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x4f32(sQ[row * P + lo], ds, dk, 0, 0, 0);
            }
        }
        const int key = 16 * kt + lo;
        if (key < N1) {
            float* o = dqkv + ((size_t)b * N1 + key) * LD + h * 16 + 4 * hi;
            *reinterpret_cast<float4*>(o + ELG_E) = make_float4(dk[0], dk[1], dk[2], dk[3]);
            *reinterpret_cast<float4*>(o + 2 * ELG_E) = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// workspace layout (floats)
struct EncWs {
    long R, X0, tmp, layer0, layer_stride;
    long oQKV, oO, oLSE, oXH1, oRS1, oX1, oH, oXH2, oRS2, oXout, total;
};
static EncWs enc_ws(int B, int N1, int n_layers, int ff, int save) {
    EncWs w;
    if (n_layers == 0) { w = EncWs{}; w.R = (long)B * N1; return w; }
    w.R = (long)B * N1;
    long o = 0;
    w.X0 = o; o += w.R * ELG_E;
    w.tmp = o; o += w.R * ELG_E;
    w.layer0 = o;
    long p = 0;
    w.oQKV = p; p += w.R * 3 * ELG_E;
    w.oO = p; p += w.R * ELG_E;
    w.oLSE = p; p += ((long)B * 8 * N1 + 3) / 4 * 4;
    w.oXH1 = p; p += w.R * ELG_E;
    w.oRS1 = p; p += (long)B * ELG_E;
    w.oX1 = p; p += w.R * ELG_E;
    w.oH = p; p += w.R * ff;
    w.oXH2 = p; p += w.R * ELG_E;
    w.oRS2 = p; p += (long)B * ELG_E;
    w.oXout = p; p += w.R * ELG_E;
    w.layer_stride = save ? p : 0;
    w.total = w.layer0 + (save ? p * n_layers : p);
    return w;
}

static EncGemm gemm_base(const float* A, int lda, int N, int K, long rows, int N1, bool aligned) {
    EncGemm g{};
    g.A = A; g.lda = lda; g.N = N; g.K = K; g.rows_total = (int)rows;
    if (aligned) { g.blk_rows = N1; g.blk_stride = N1; }
    else { g.blk_rows = 128; g.blk_stride = 128; }
    g.alpha = 1.f; g.cblk = N; g.wblk = N; g.wmode = W_NK; g.epi = EPI_STORE;
    return g;
}

}  // namespace elg

using namespace elg;

extern "C" int elg_add_instnorm_fwd(const float*, const float*, const float*, const float*, float*, float*, float*, int, int,
                                    int, float, void*);
extern "C" int elg_add_instnorm_bwd(const float*, const float*, const float*, const float*, float*, float*, float*, int, int,
                                    int, void*);

extern "C" int64_t elg_encoder_ws_floats(int B, int N1, int n_layers, int ff_hidden, int save) {
    if (B <= 0 || N1 <= 0 || n_layers < 0 || ff_hidden <= 0) return 0;
    return enc_ws(B, N1, n_layers, ff_hidden, save).total;
}

extern "C" int64_t elg_encoder_bwd_ws_floats(int B, int N1, int ff_hidden) {
    if (B <= 0 || N1 <= 0 || ff_hidden <= 0) return 0;
    return (int64_t)B * N1 * (4 * ELG_E + ff_hidden + 3 * ELG_E);
}

static int check_enc_args(const elg_encoder_args* a) {
    if (!a) return fail(ELG_EINVAL, "encoder: null args");
    if (a->B <= 0 || a->N1 < 4) return fail(ELG_EINVAL, "encoder: need B > 0, N1 >= 4");
    if (a->n_layers < 0 || a->n_layers > ELG_ENC_MAX_LAYERS) return fail(ELG_EINVAL, "encoder: 0 .. 8 layers");
    if (a->n_layers == 0) {          // set_kv only: `enc` is an input
        if (!a->enc || !a->K) return fail(ELG_EINVAL, "encoder: n_layers = 0 needs enc (input) and the table buffers");
        return ELG_OK;
    }
    if (a->ff_hidden <= 0 || (a->ff_hidden % 128)) return fail(ELG_EINVAL, "encoder: ff_hidden must be a multiple of 128");
    if ((long)a->B * a->N1 > 0x7fffffffL / (4 * ELG_E)) return fail(ELG_EINVAL, "encoder: batch * nodes too large");
    if (!a->xy || !a->enc || !a->ws) return fail(ELG_EINVAL, "encoder: null buffer");
    if (a->problem == ELG_PROBLEM_CVRP && (!a->demand || !a->W.emb_depot_w || !a->W.emb_depot_b))
        return fail(ELG_EINVAL, "encoder: CVRP needs demand and the depot embedding");
    if (!a->W.emb_w || !a->W.emb_b) return fail(ELG_EINVAL, "encoder: null embedding");
    const EncWs w = enc_ws(a->B, a->N1, a->n_layers, a->ff_hidden, a->save);
    if (a->ws_floats < w.total) return fail(ELG_EINVAL, "encoder: workspace too small (elg_encoder_ws_floats)");
    return ELG_OK;
}

#define ENC_TRY(x)                \
    {                             \
        const int rc_ = (x);      \
        if (rc_ != ELG_OK) return rc_; \
    }

extern "C" int elg_encoder_fwd(const elg_encoder_args* a, void* stream) {
    ENC_TRY(check_enc_args(a))
    hipStream_t s = (hipStream_t)stream;
    const int B = a->B, N1 = a->N1, FF = a->ff_hidden;
    const bool tsp = a->problem == ELG_PROBLEM_TSP;
    const EncWs w = enc_ws(B, N1, a->n_layers, FF, a->save);
    const long R = w.R;
    const bool aligned = N1 <= 128;
    float* ws = a->ws;
    float* X0 = ws + w.X0;
    (void)hipGetLastError();
    if (a->n_layers > 0) {
        const long total = R * ELG_E;
        hipLaunchKernelGGL(enc_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a->xy,
                           tsp ? nullptr : a->demand, tsp ? nullptr : a->W.emb_depot_w, tsp ? nullptr : a->W.emb_depot_b,
                           a->W.emb_w, a->W.emb_b, X0, N1, total);
        ENC_TRY(launch_status("enc_embed"))
    }
    const float* Xin = X0;
    for (int l = 0; l < a->n_layers; ++l) {
        const elg_enc_layer& L = a->W.layer[l];
        float* lb = ws + w.layer0 + w.layer_stride * l;
        float *QKV = lb + w.oQKV, *O = lb + w.oO, *LSE = lb + w.oLSE, *XH1 = lb + w.oXH1, *RS1 = lb + w.oRS1;
        float *X1 = lb + w.oX1, *H = lb + w.oH, *XH2 = lb + w.oXH2, *RS2 = lb + w.oRS2;
        float* Xout = (l == a->n_layers - 1) ? a->enc : (a->save ? lb + w.oXout : X0);
        {   // Q | K | V = x [Wq; Wk; Wv]^T   (models.py:253-255)
            EncGemm g = gemm_base(Xin, ELG_E, 3 * ELG_E, ELG_E, R, N1, aligned);
            g.W[0] = L.Wq; g.W[1] = L.Wk; g.W[2] = L.Wv; g.wblk = ELG_E; g.ldw = ELG_E;
            g.C[0] = QKV; g.ldc = 3 * ELG_E;
            ENC_TRY(launch_gemm(g, s))
        }
        (void)hipGetLastError();
        hipLaunchKernelGGL(enc_attn_fwd_kernel, dim3((N1 + 63) / 64, B * 8), dim3(256), 0, s, QKV, O, LSE, N1);
        ENC_TRY(launch_status("enc_attn_fwd"))
        {   // x1 = InstanceNorm(x + combine(att))   (models.py:262-264)
            EncGemm g = gemm_base(O, ELG_E, ELG_E, ELG_E, R, N1, aligned);
            g.W[0] = L.Wc; g.ldw = ELG_E; g.bias = L.bc; g.R = Xin; g.ldr = ELG_E; g.ldc = ELG_E;
            if (aligned) {
                g.C[0] = X1; g.epi = EPI_NORM; g.gamma = L.g1; g.beta = L.b1; g.xhat = a->save ? XH1 : nullptr;
                g.rstd = a->save ? RS1 : nullptr; g.eps = a->eps;
                ENC_TRY(launch_gemm(g, s))
            } else {
                g.C[0] = ws + w.tmp; g.epi = EPI_ADDBIAS;
                ENC_TRY(launch_gemm(g, s))
                ENC_TRY(elg_add_instnorm_fwd(ws + w.tmp, nullptr, L.g1, L.b1, X1, XH1, RS1, B, N1, ELG_E, a->eps, stream))
            }
        }
        {   // h = relu(x1 W1^T + b1)   (models.py:559-560)
            EncGemm g = gemm_base(X1, ELG_E, FF, ELG_E, R, N1, aligned);
            g.W[0] = L.W1; g.ldw = ELG_E; g.bias = L.bf1; g.C[0] = H; g.ldc = FF; g.epi = EPI_RELU;
            ENC_TRY(launch_gemm(g, s))
        }
        {   // out = InstanceNorm(x1 + h W2^T + b2)
            EncGemm g = gemm_base(H, FF, ELG_E, FF, R, N1, aligned);
            g.W[0] = L.W2; g.ldw = FF; g.bias = L.bf2; g.R = X1; g.ldr = ELG_E; g.ldc = ELG_E;
            if (aligned) {
                g.C[0] = Xout; g.epi = EPI_NORM; g.gamma = L.g2; g.beta = L.b2; g.xhat = a->save ? XH2 : nullptr;
                g.rstd = a->save ? RS2 : nullptr; g.eps = a->eps;
                ENC_TRY(launch_gemm(g, s))
            } else {
                g.C[0] = ws + w.tmp; g.epi = EPI_ADDBIAS;
                ENC_TRY(launch_gemm(g, s))
                ENC_TRY(elg_add_instnorm_fwd(ws + w.tmp, nullptr, L.g2, L.b2, Xout, XH2, RS2, B, N1, ELG_E, a->eps, stream))
            }
        }
        Xin = Xout;
    }
    if (!a->K) return ELG_OK;          // encoder only
    // ---- decoder tables (set_kv + the folds of engine.fold_decoder_tables)
    if (!a->V || !a->PK || !a->pb || !a->Q1 || !a->W.dec_Wk || !a->W.dec_Wv || !a->W.dec_Wc || !a->W.dec_bc || !a->W.dec_Wq_last)
        return fail(ELG_EINVAL, "encoder: decoder tables requested but a buffer / weight is null");
    if (tsp && (!a->Q2 || !a->W.dec_Wq_first)) return fail(ELG_EINVAL, "encoder: TSP needs Q2 / Wq_first");
    const float inv_sqrt_e = 0.08838834764831845f;
    {   // K | V (| Q1 | Q2 for TSP: all square weights) in one launch
        const int nb = tsp ? 4 : 2;
        EncGemm g = gemm_base(a->enc, ELG_E, nb * ELG_E, ELG_E, R, N1, aligned);
        g.W[0] = a->W.dec_Wk; g.W[1] = a->W.dec_Wv; g.W[2] = a->W.dec_Wq_last; g.W[3] = a->W.dec_Wq_first;
        g.wblk = ELG_E; g.ldw = ELG_E;
        g.C[0] = a->K; g.C[1] = a->V; g.C[2] = a->Q1; g.C[3] = a->Q2; g.cblk = ELG_E; g.ldc = ELG_E;
        ENC_TRY(launch_gemm(g, s))
    }
    if (!tsp) {   // Q1 = enc Wq_last[:, :128]^T (row pitch 129: scalar weight loads)
        EncGemm g = gemm_base(a->enc, ELG_E, ELG_E, ELG_E, R, N1, aligned);
        g.W[0] = a->W.dec_Wq_last; g.ldw = ELG_E + 1; g.wmode = W_NK_SCALAR; g.C[0] = a->Q1; g.ldc = ELG_E;
        ENC_TRY(launch_gemm(g, s))
    }
    {   // PK = enc Wc / sqrt(E)
        EncGemm g = gemm_base(a->enc, ELG_E, ELG_E, ELG_E, R, N1, aligned);
        g.W[0] = a->W.dec_Wc; g.ldw = ELG_E; g.wmode = W_KN; g.wblk = ELG_E; g.C[0] = a->PK; g.ldc = ELG_E; g.alpha = inv_sqrt_e;
        ENC_TRY(launch_gemm(g, s))
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(enc_pb_wl_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, s, a->enc, a->W.dec_bc,
                       tsp ? nullptr : a->W.dec_Wq_last, a->pb, tsp ? nullptr : a->wl, R, inv_sqrt_e);
    return launch_status("enc_pb_wl");
}

static int dw_gemm(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, int M, int N, long rows, float* dbias,
                   float alpha, void* stream) {
    // dW[M,N] += alpha dY^T X over `rows` rows, bias gradient = column sums of dY (split-K MFMA GEMM, csrc/elg_gemm.hip)
    const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
    int split = (int)max(1L, min(64L, min(rows / 128, (long)max(4, 512 / tiles))));
    return elg_gemm_f32_alpha(dY, X, dW, nullptr, M, N, (int)rows, ldy, ldx, ldw, 1, 0, 0, split, dbias, alpha, stream);
}

extern "C" int elg_encoder_bwd(const elg_encoder_bwd_args* ba, void* stream) {
    if (!ba) return fail(ELG_EINVAL, "encoder bwd: null args");
    const elg_encoder_args* a = &ba->fwd;
    ENC_TRY(check_enc_args(a))
    if (!a->save || a->n_layers < 1) return fail(ELG_EINVAL, "encoder bwd: the forward must have run with save = 1");
    if (a->N1 > 128) return fail(ELG_ENOTIMPL, "encoder bwd: N1 > 128 not built");
    hipStream_t s = (hipStream_t)stream;
    const int B = a->B, N1 = a->N1, FF = a->ff_hidden;
    const bool tsp = a->problem == ELG_PROBLEM_TSP;
    const EncWs w = enc_ws(B, N1, a->n_layers, FF, 1);
    const long R = w.R;
    if (!ba->ws2 || ba->ws2_floats < elg_encoder_bwd_ws_floats(B, N1, FF)) return fail(ELG_EINVAL, "encoder bwd: scratch too small");
    const elg_enc_weights& G = ba->G;
    float* gX = ba->ws2;
    float* gS = gX + R * ELG_E;
    float* gY = gS + R * ELG_E;
    float* gO = gY + R * ELG_E;
    float* gH = gO + R * ELG_E;
    float* dQKV = gH + R * FF;
    float* ws = a->ws;
    const float inv_sqrt_e = 0.08838834764831845f;
    const bool aligned = true;
    // ---- d enc from the decoder tables (autograd of set_kv / fold_decoder_tables)
    bool have = false;          // gX holds a value
    if (ba->g_enc) {
        if (hipMemcpyAsync(gX, ba->g_enc, sizeof(float) * R * ELG_E, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(ELG_ELAUNCH, "encoder bwd: copy failed");
        have = true;
    }
    auto acc_into_gX = [&](const float* gT, const float* Wt, int ldw, int wmode, float alpha) -> int {
        if (!have) {
            if (hipMemsetAsync(gX, 0, sizeof(float) * R * ELG_E, s) != hipSuccess) return fail(ELG_ELAUNCH, "encoder bwd: memset failed");
            have = true;
        }
        EncGemm g = gemm_base(gT, ELG_E, ELG_E, ELG_E, R, N1, aligned);
        g.W[0] = Wt; g.ldw = ldw; g.wmode = wmode; g.wblk = ELG_E; g.C[0] = gX; g.ldc = ELG_E; g.epi = EPI_ADD; g.R = gX;
        g.ldr = ELG_E; g.alpha = alpha;
        return launch_gemm(g, s);
    };
    auto need = [&](const float* p, const char* what) -> int { return p ? ELG_OK : fail(ELG_EINVAL, std::string("encoder bwd: null ") + what); };
    if (ba->gK) {
        ENC_TRY(need(G.dec_Wk, "d Wk"))
        ENC_TRY(acc_into_gX(ba->gK, a->W.dec_Wk, ELG_E, W_KN, 1.f))
        ENC_TRY(dw_gemm(ba->gK, ELG_E, a->enc, ELG_E, (float*)G.dec_Wk, ELG_E, ELG_E, ELG_E, R, nullptr, 1.f, stream))
    }
    if (ba->gV) {
        ENC_TRY(need(G.dec_Wv, "d Wv"))
        ENC_TRY(acc_into_gX(ba->gV, a->W.dec_Wv, ELG_E, W_KN, 1.f))
        ENC_TRY(dw_gemm(ba->gV, ELG_E, a->enc, ELG_E, (float*)G.dec_Wv, ELG_E, ELG_E, ELG_E, R, nullptr, 1.f, stream))
    }
    if (ba->gQ1) {
        ENC_TRY(need(G.dec_Wq_last, "d Wq_last"))
        const int ldq = tsp ? ELG_E : ELG_E + 1;
        ENC_TRY(acc_into_gX(ba->gQ1, a->W.dec_Wq_last, ldq, W_KN, 1.f))
        ENC_TRY(dw_gemm(ba->gQ1, ELG_E, a->enc, ELG_E, (float*)G.dec_Wq_last, ldq, ELG_E, ELG_E, R, nullptr, 1.f, stream))
    }
    if (tsp && ba->gQ2) {
        ENC_TRY(need(G.dec_Wq_first, "d Wq_first"))
        ENC_TRY(acc_into_gX(ba->gQ2, a->W.dec_Wq_first, ELG_E, W_KN, 1.f))
        ENC_TRY(dw_gemm(ba->gQ2, ELG_E, a->enc, ELG_E, (float*)G.dec_Wq_first, ELG_E, ELG_E, ELG_E, R, nullptr, 1.f, stream))
    }
    if (ba->gPK) {
        ENC_TRY(need(G.dec_Wc, "d Wc"))
        // PK = enc Wc / sqrt(E):  d enc += gPK Wc^T / sqrt(E) ;  d Wc = enc^T gPK / sqrt(E)
        if (!have) {
            if (hipMemsetAsync(gX, 0, sizeof(float) * R * ELG_E, s) != hipSuccess) return fail(ELG_ELAUNCH, "encoder bwd: memset failed");
            have = true;
        }
        EncGemm g = gemm_base(ba->gPK, ELG_E, ELG_E, ELG_E, R, N1, aligned);
        g.W[0] = a->W.dec_Wc; g.ldw = ELG_E; g.wmode = W_NK; g.C[0] = gX; g.ldc = ELG_E; g.epi = EPI_ADD; g.R = gX; g.ldr = ELG_E;
        g.alpha = inv_sqrt_e;
        if (ba->gpb) { g.rv = ba->gpb; g.cv = a->W.dec_bc; g.alpha2 = inv_sqrt_e; }
        ENC_TRY(launch_gemm(g, s))
        ENC_TRY(dw_gemm(a->enc, ELG_E, ba->gPK, ELG_E, (float*)G.dec_Wc, ELG_E, ELG_E, ELG_E, R, nullptr, inv_sqrt_e, stream))
    } else if (ba->gpb) return fail(ELG_EINVAL, "encoder bwd: gpb without gPK");
    if (ba->gpb || (ba->gwl && !tsp)) {
        if (ba->gpb) ENC_TRY(need(G.dec_bc, "d bc"))
        const int rpb = 64;
        (void)hipGetLastError();
        hipLaunchKernelGGL(enc_fold_small_bwd_kernel, dim3(ba->gpb ? (unsigned)((R + rpb - 1) / rpb) : 1u), dim3(128), 0, s, a->enc,
                           ba->gpb, tsp ? nullptr : ba->gwl, (float*)G.dec_bc, (float*)G.dec_Wq_last, R, rpb, inv_sqrt_e);
        ENC_TRY(launch_status("enc_fold_small_bwd"))
    }
    if (!have) return fail(ELG_EINVAL, "encoder bwd: no cotangent given");
    // ---- layers, last to first
    for (int l = a->n_layers - 1; l >= 0; --l) {
        const elg_enc_layer& L = a->W.layer[l];
        const elg_enc_layer& GL = G.layer[l];
        float* lb = ws + w.layer0 + w.layer_stride * l;
        const float *QKV = lb + w.oQKV, *O = lb + w.oO, *LSE = lb + w.oLSE, *XH1 = lb + w.oXH1, *RS1 = lb + w.oRS1;
        const float *X1 = lb + w.oX1, *H = lb + w.oH, *XH2 = lb + w.oXH2, *RS2 = lb + w.oRS2;
        const float* Xin = (l == 0) ? ws + w.X0 : ws + w.layer0 + w.layer_stride * (l - 1) + w.oXout;
        // second add & norm
        ENC_TRY(elg_add_instnorm_bwd(gX, XH2, RS2, L.g2, gS, (float*)GL.g2, (float*)GL.b2, B, N1, ELG_E, stream))
        ENC_TRY(dw_gemm(gS, ELG_E, H, FF, (float*)GL.W2, FF, ELG_E, FF, R, (float*)GL.bf2, 1.f, stream))
        {   // dH = (dS2 W2) * [h > 0]
            EncGemm g = gemm_base(gS, ELG_E, FF, ELG_E, R, N1, aligned);
            g.W[0] = L.W2; g.ldw = FF; g.wmode = W_KN; g.wblk = ELG_E; g.C[0] = gH; g.ldc = FF; g.epi = EPI_RELUMASK; g.R = H; g.ldr = FF;
            ENC_TRY(launch_gemm(g, s))
        }
        ENC_TRY(dw_gemm(gH, FF, X1, ELG_E, (float*)GL.W1, ELG_E, FF, ELG_E, R, (float*)GL.bf1, 1.f, stream))
        {   // d x1 = dS2 + dH W1
            EncGemm g = gemm_base(gH, FF, ELG_E, FF, R, N1, aligned);
            g.W[0] = L.W1; g.ldw = ELG_E; g.wmode = W_KN; g.wblk = FF; g.C[0] = gS; g.ldc = ELG_E; g.epi = EPI_ADD; g.R = gS; g.ldr = ELG_E;
            ENC_TRY(launch_gemm(g, s))
        }
        // first add & norm
        ENC_TRY(elg_add_instnorm_bwd(gS, XH1, RS1, L.g1, gY, (float*)GL.g1, (float*)GL.b1, B, N1, ELG_E, stream))
        ENC_TRY(dw_gemm(gY, ELG_E, O, ELG_E, (float*)GL.Wc, ELG_E, ELG_E, ELG_E, R, (float*)GL.bc, 1.f, stream))
        {   // d att = dY Wc
            EncGemm g = gemm_base(gY, ELG_E, ELG_E, ELG_E, R, N1, aligned);
            g.W[0] = L.Wc; g.ldw = ELG_E; g.wmode = W_KN; g.wblk = ELG_E; g.C[0] = gO; g.ldc = ELG_E;
            ENC_TRY(launch_gemm(g, s))
        }
        (void)hipGetLastError();
        {
            const int nt = (N1 + 15) / 16;
            if (nt <= 2) hipLaunchKernelGGL((enc_attn_bwd_kernel<2>), dim3(B * 8), dim3(256), 0, s, QKV, gO, O, LSE, dQKV, N1);
            else if (nt <= 4) hipLaunchKernelGGL((enc_attn_bwd_kernel<4>), dim3(B * 8), dim3(256), 0, s, QKV, gO, O, LSE, dQKV, N1);
            else if (nt <= 7) hipLaunchKernelGGL((enc_attn_bwd_kernel<7>), dim3(B * 8), dim3(256), 0, s, QKV, gO, O, LSE, dQKV, N1);
            else hipLaunchKernelGGL((enc_attn_bwd_kernel<8>), dim3(B * 8), dim3(256), 0, s, QKV, gO, O, LSE, dQKV, N1);
        }
        ENC_TRY(launch_status("enc_attn_bwd"))
        ENC_TRY(dw_gemm(dQKV, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wq, ELG_E, ELG_E, ELG_E, R, nullptr, 1.f, stream))
        ENC_TRY(dw_gemm(dQKV + ELG_E, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wk, ELG_E, ELG_E, ELG_E, R, nullptr, 1.f, stream))
        ENC_TRY(dw_gemm(dQKV + 2 * ELG_E, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wv, ELG_E, ELG_E, ELG_E, R, nullptr, 1.f, stream))
        {   // d x = dY + dQ Wq + dK Wk + dV Wv
            EncGemm g = gemm_base(dQKV, 3 * ELG_E, ELG_E, 3 * ELG_E, R, N1, aligned);
            g.W[0] = L.Wq; g.W[1] = L.Wk; g.W[2] = L.Wv; g.ldw = ELG_E; g.wmode = W_KN; g.wblk = ELG_E; g.C[0] = gX; g.ldc = ELG_E;
            g.epi = EPI_ADD; g.R = gY; g.ldr = ELG_E;
            ENC_TRY(launch_gemm(g, s))
        }
    }
    // ---- input embeddings
    {
        const int rpb = 32;
        (void)hipGetLastError();
        hipLaunchKernelGGL(enc_embed_bwd_kernel, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(128), 0, s, a->xy,
                           tsp ? nullptr : a->demand, gX, tsp ? nullptr : (float*)G.emb_depot_w, tsp ? nullptr : (float*)G.emb_depot_b,
                           (float*)G.emb_w, (float*)G.emb_b, N1, R, rpb);
        ENC_TRY(launch_status("enc_embed_bwd"))
    }
    return ELG_OK;
}
