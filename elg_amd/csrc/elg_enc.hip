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
#include "elg_enc_internal.h"
#include "elg_bf16.h"
#include <cstdlib>

extern "C" __attribute__((visibility("hidden"))) int elg_gemm_f32_alpha(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                                  int ldb, int ldc, int transA, int transB, int relu, int split_k, float* a_rowsum,
                                  float alpha, void* stream);

namespace elg {

enum { EPI_STORE = 0, EPI_RELU = 1, EPI_NORM = 2, EPI_ADD = 3, EPI_RELUMASK = 4, EPI_ADDBIAS = 5, EPI_ADD_NORMBWD = 6 };
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
    float* dgamma; float* dbeta;        // EPI_ADD_NORMBWD: d = residual + x W; C = backward of the add & norm that produced the
                                        // GEMM's consumer (xhat, rstd, gamma as saved by the forward), dgamma / dbeta accumulated
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

template <int RT, int KS, int WMODE>
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
    if (WMODE != W_KN) wp = pick6(g.W, nwb) + (size_t)(ncol - nwb * g.wblk) * g.ldw + 4 * hi;

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
            if (WMODE == W_NK) WV[c] = *reinterpret_cast<const float4*>(wp + kk_);                                \
            else if (WMODE == W_NK_SCALAR) WV[c] = make_float4(wp[kk_], wp[kk_ + 1], wp[kk_ + 2], wp[kk_ + 3]); \
            else {                                                                                                \
                const int kb_ = __builtin_amdgcn_readfirstlane(kk_ / g.wblk), kr_ = kk_ - kb_ * g.wblk + 4 * hi;  \
                const float* p_ = pick6(g.W, kb_) + (size_t)kr_ * g.ldw + ncol;                                   \
                WV[c] = make_float4(p_[0], p_[g.ldw], p_[2 * g.ldw], p_[3 * g.ldw]);                              \
            }                                                                                                     \
        }                                                                                                         \
    }
    // epilogue operands (residual / mask rows, the saved xhat of a fused norm backward) are requested before the main loop:
    // these launches are latency-bound, a load issued in the epilogue is a full memory round trip added to every kernel.
    // (g.R may alias the output: every element is read and written by the same thread)
    constexpr int TC = 16 * NWC, TP = TC + 4, CG = TC / 4, RPP = 256 / CG;
    constexpr int NP = (ROWS + RPP - 1) / RPP;
    const int cg = tid % CG, rl = tid / CG;
    const int c0 = blockIdx.x * TC;                       // first output column of the workgroup
    const int gc = c0 + 4 * cg;                           // global column of this thread's 4 values
    float4 rpre[NP], xpre[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rr = min(rl + p * RPP, nrows - 1);
        rpre[p] = g.R ? *reinterpret_cast<const float4*>(g.R + (size_t)(row0 + rr) * g.ldr + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
        xpre[p] = g.epi == EPI_ADD_NORMBWD ? *reinterpret_cast<const float4*>(g.xhat + (size_t)(row0 + rr) * g.N + gc)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 w[NCH];
    {
        float4 st[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) st[i] = *reinterpret_cast<const float4*>(sp[i]);
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
        for (int c = 0; c < NCH; ++c) {
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
    // ---- epilogue.  The D tiles (lane = one column, four rows) would be written as 64-byte row pieces (measured: 0.4 TB/s,
    // the dominant cost of the first version); instead the workgroup's TC = 16 NWC output columns are transposed through LDS
    // and written / combined with the residual as whole 16-byte-per-lane row segments (128 or 256 contiguous bytes per row).
    float* red = lds;                                     // the slabs are dead: reuse
    float* T = lds + ((KS > 1) ? (KS - 1) * NWC * RT * 4 * 64 : 0);
    if (KS > 1) {
        if (ks > 0) {
            float* my = red + (size_t)((ks - 1) * NWC + wc) * RT * 4 * 64;
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) my[(t * 4 + i) * 64 + lane] = acc[t][i];
        }
        __syncthreads();
    }
    if (ks == 0) {
        if (KS > 1) {
#pragma unroll
            for (int s = 0; s < KS - 1; ++s) {
                const float* o = red + (size_t)(s * NWC + wc) * RT * 4 * 64;
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][i] += o[(t * 4 + i) * 64 + lane];
            }
        }
        // D layout: acc[t][i] = C[row 16 t + 4 hi + i][column n0 + lo]
        const float bv = (g.bias && g.epi != EPI_ADD && g.epi != EPI_RELUMASK) ? g.bias[ncol] : 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = fmaf(acc[t][i], g.alpha, bv);
                if (g.epi == EPI_RELU) v = fmaxf(v, 0.f);
                T[(16 * t + 4 * hi + i) * TP + 16 * wc + lo] = v;
            }
    }
    __syncthreads();
    const int ncb = __builtin_amdgcn_readfirstlane(c0 / g.cblk);
    float* cbase = pick6(g.C, ncb) + (c0 - ncb * g.cblk) + 4 * cg;
    if (g.epi == EPI_NORM) {
        // s = residual + (x W^T + b); InstanceNorm1d statistics per channel over the block's rows (two passes, like the
        // stand-alone kernel): column partials per thread, summed over the threads of a column group
        float* stat = T + ROWS * TP;                      // [4 waves][TC]
        auto colsum = [&](float4 v) -> float4 {
#pragma unroll
            for (int m = CG; m < 64; m <<= 1) {
                v.x += shfl_xor(v.x, m); v.y += shfl_xor(v.y, m); v.z += shfl_xor(v.z, m); v.w += shfl_xor(v.w, m);
            }
            if (lane < CG) *reinterpret_cast<float4*>(stat + wave * TC + 4 * lane) = v;
            __syncthreads();
            float4 r = *reinterpret_cast<const float4*>(stat + 4 * cg);
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) {
                const float4 o = *reinterpret_cast<const float4*>(stat + w2 * TC + 4 * cg);
                r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            __syncthreads();
            return r;
        };
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = rl + p * RPP;
            if (r < nrows) {
                float4 v = *reinterpret_cast<const float4*>(T + r * TP + 4 * cg);
                const float4 res = rpre[p];
                v.x += res.x; v.y += res.y; v.z += res.z; v.w += res.w;
                *reinterpret_cast<float4*>(T + r * TP + 4 * cg) = v;
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        }
        sum = colsum(sum);
        const float inv_n = 1.0f / (float)nrows;
        const float4 mean = make_float4(sum.x * inv_n, sum.y * inv_n, sum.z * inv_n, sum.w * inv_n);
        float4 sq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = rl + p * RPP;
            if (r < nrows) {
                const float4 v = *reinterpret_cast<const float4*>(T + r * TP + 4 * cg);
                const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
                sq.x = fmaf(dx, dx, sq.x); sq.y = fmaf(dy, dy, sq.y); sq.z = fmaf(dz, dz, sq.z); sq.w = fmaf(dw, dw, sq.w);
            }
        }
        sq = colsum(sq);
        const float4 rs = make_float4(1.0f / sqrtf(sq.x * inv_n + g.eps), 1.0f / sqrtf(sq.y * inv_n + g.eps),
                                      1.0f / sqrtf(sq.z * inv_n + g.eps), 1.0f / sqrtf(sq.w * inv_n + g.eps));
        const float4 ga = *reinterpret_cast<const float4*>(g.gamma + gc);
        const float4 be = *reinterpret_cast<const float4*>(g.beta + gc);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = rl + p * RPP;
            if (r < nrows) {
                const float4 v = *reinterpret_cast<const float4*>(T + r * TP + 4 * cg);
                const float4 xh = make_float4((v.x - mean.x) * rs.x, (v.y - mean.y) * rs.y, (v.z - mean.z) * rs.z, (v.w - mean.w) * rs.w);
                *reinterpret_cast<float4*>(cbase + (size_t)(row0 + r) * g.ldc) =
                    make_float4(fmaf(xh.x, ga.x, be.x), fmaf(xh.y, ga.y, be.y), fmaf(xh.z, ga.z, be.z), fmaf(xh.w, ga.w, be.w));
                if (g.xhat) *reinterpret_cast<float4*>(g.xhat + (size_t)(row0 + r) * g.N + gc) = xh;
            }
        }
        if (rl == 0 && g.rstd) *reinterpret_cast<float4*>(g.rstd + (size_t)blockIdx.y * g.N + gc) = rs;
        return;
    }
    if (g.epi == EPI_ADD_NORMBWD) {
        // d = residual + x W is the cotangent of an add & instance norm output y = xhat gamma + beta (per instance and
        // channel over the block's rows): ds = gamma rstd (d - mean(d) - xhat mean(d xhat)), dbeta += sum d, dgamma += sum d xhat.
        // The workgroup owns whole columns of the instance, so the stand-alone norm-backward launch (one per norm) folds
        // into this epilogue.
        float* stat = T + ROWS * TP;                      // [4 waves][TC]
        auto colsum = [&](float4 v) -> float4 {
#pragma unroll
            for (int m = CG; m < 64; m <<= 1) {
                v.x += shfl_xor(v.x, m); v.y += shfl_xor(v.y, m); v.z += shfl_xor(v.z, m); v.w += shfl_xor(v.w, m);
            }
            if (lane < CG) *reinterpret_cast<float4*>(stat + wave * TC + 4 * lane) = v;
            __syncthreads();
            float4 r = *reinterpret_cast<const float4*>(stat + 4 * cg);
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) {
                const float4 o = *reinterpret_cast<const float4*>(stat + w2 * TC + 4 * cg);
                r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            __syncthreads();
            return r;
        };
        float4 xh[NP];
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = rl + p * RPP;
            xh[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nrows) {
                float4 v = *reinterpret_cast<const float4*>(T + r * TP + 4 * cg);
                const float4 res = rpre[p];
                v.x += res.x; v.y += res.y; v.z += res.z; v.w += res.w;
                *reinterpret_cast<float4*>(T + r * TP + 4 * cg) = v;
                xh[p] = xpre[p];
                s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
                s2.x = fmaf(v.x, xh[p].x, s2.x); s2.y = fmaf(v.y, xh[p].y, s2.y);
                s2.z = fmaf(v.z, xh[p].z, s2.z); s2.w = fmaf(v.w, xh[p].w, s2.w);
            }
        }
        s1 = colsum(s1);
        s2 = colsum(s2);
        if (rl == 0) {
            atomicAdd(g.dbeta + gc, s1.x); atomicAdd(g.dbeta + gc + 1, s1.y); atomicAdd(g.dbeta + gc + 2, s1.z); atomicAdd(g.dbeta + gc + 3, s1.w);
            atomicAdd(g.dgamma + gc, s2.x); atomicAdd(g.dgamma + gc + 1, s2.y); atomicAdd(g.dgamma + gc + 2, s2.z); atomicAdd(g.dgamma + gc + 3, s2.w);
        }
        const float fn = (float)nrows;                    // (true divisions: the means of ~1e2 values must be as exact as f32 allows,
                                                          //  the bias gradients upstream are sums of the ds that cancel to 0)
        const float4 ga = *reinterpret_cast<const float4*>(g.gamma + gc);
        const float4 rs = *reinterpret_cast<const float4*>(g.rstd + (size_t)blockIdx.y * g.N + gc);
        const float4 k = make_float4(ga.x * rs.x, ga.y * rs.y, ga.z * rs.z, ga.w * rs.w);
        const float4 m1 = make_float4(s1.x / fn, s1.y / fn, s1.z / fn, s1.w / fn);
        const float4 m2 = make_float4(s2.x / fn, s2.y / fn, s2.z / fn, s2.w / fn);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = rl + p * RPP;
            if (r < nrows) {
                const float4 v = *reinterpret_cast<const float4*>(T + r * TP + 4 * cg);
                *reinterpret_cast<float4*>(cbase + (size_t)(row0 + r) * g.ldc) =
                    make_float4(k.x * (v.x - m1.x - xh[p].x * m2.x), k.y * (v.y - m1.y - xh[p].y * m2.y),
                                k.z * (v.z - m1.z - xh[p].z * m2.z), k.w * (v.w - m1.w - xh[p].w * m2.w));
            }
        }
        return;
    }
    float4 cv4 = make_float4(0.f, 0.f, 0.f, 0.f), bias4 = cv4;
    if (g.epi == EPI_ADD && g.cv) {
        cv4 = *reinterpret_cast<const float4*>(g.cv + gc);
        cv4.x *= g.alpha2; cv4.y *= g.alpha2; cv4.z *= g.alpha2; cv4.w *= g.alpha2;
    }
    (void)bias4;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = rl + p * RPP;
        if (r < nrows) {
            const size_t grow = (size_t)(row0 + r);
            float4 v = *reinterpret_cast<const float4*>(T + r * TP + 4 * cg);
            if (g.epi == EPI_ADD || g.epi == EPI_ADDBIAS) {
                const float4 res = rpre[p];
                v.x += res.x; v.y += res.y; v.z += res.z; v.w += res.w;
                if (g.epi == EPI_ADD && g.cv) {
                    const float rvv = g.rv[grow];
                    v.x = fmaf(rvv, cv4.x, v.x); v.y = fmaf(rvv, cv4.y, v.y); v.z = fmaf(rvv, cv4.z, v.z); v.w = fmaf(rvv, cv4.w, v.w);
                }
            } else if (g.epi == EPI_RELUMASK) {
                const float4 m = rpre[p];
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4*>(cbase + grow * g.ldc) = v;
        }
    }
}

template <int RT, int KS, int WMODE>
static int launch_gemm_t(const EncGemm& g, dim3 grid, hipStream_t s) {
    const int nbuf = g.K > ENC_KS ? 2 : 1;
    const size_t lds = (size_t)nbuf * RT * 16 * ENC_AP * sizeof(float);
    auto kern = enc_gemm_kernel<RT, KS, WMODE>;
    static DynLds optin;
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), 2 * RT * 16 * ENC_AP * sizeof(float)))
        return fail(ELG_ELAUNCH, "encoder gemm: hipFuncSetAttribute failed");
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g);
    return launch_status("enc_gemm");
}

static int launch_gemm(const EncGemm& g, hipStream_t s) {
    if ((g.N & 15) || g.N <= 0) return fail(ELG_EINVAL, "encoder gemm: N must be a multiple of 16");
    const int nblk = (g.rows_total + g.blk_stride - 1) / g.blk_stride;
    const int ctiles = g.N / 16;
    if ((g.wblk % 16) || (g.cblk % 16) || (g.K % ENC_KS)) return fail(ELG_EINVAL, "encoder gemm: K must be a multiple of 128, weight / output blocks of 16");
    // K split over the waves of a workgroup: enough waves for the chip's 1024 SIMDs
    if (g.N & 31) return fail(ELG_EINVAL, "encoder gemm: N must be a multiple of 32");
    int ks = 1;
    if (((long)ctiles * nblk < 1024) || (ctiles & 3)) ks = 2;
    dim3 grid(ctiles / (4 / ks), nblk);
#define ENC_GO2(RT, KSV)                                                                  \
    {                                                                                     \
        if (g.wmode == W_NK) return launch_gemm_t<RT, KSV, W_NK>(g, grid, s);             \
        if (g.wmode == W_KN) return launch_gemm_t<RT, KSV, W_KN>(g, grid, s);             \
        return launch_gemm_t<RT, KSV, W_NK_SCALAR>(g, grid, s);                           \
    }
#define ENC_GO(RT)                  \
    {                               \
        if (ks == 1) ENC_GO2(RT, 1) \
        ENC_GO2(RT, 2)              \
    }
    if (g.blk_rows <= 32) ENC_GO(2)
    if (g.blk_rows <= 64) ENC_GO(4)
    if (g.blk_rows <= 112) ENC_GO(7)
    if (g.blk_rows <= 128) ENC_GO(8)
#undef ENC_GO
#undef ENC_GO2
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
__global__ __launch_bounds__(512) void enc_embed_bwd_kernel(const float* __restrict__ xy, const float* __restrict__ demand,
                                                            const float* __restrict__ dX, float* gWd, float* gbd, float* gWn,
                                                            float* gbn, int N1, long rows, int rows_per_block) {
    // 512 threads = 128 channels x 4 row phases, partial sums meet in LDS
    __shared__ float part[3][7][ELG_E];
    const int c = threadIdx.x & (ELG_E - 1), q = threadIdx.x >> 7;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // node w0 w1 w2 b | depot w0 w1 b
    for (long r = r0 + q; r < r1; r += 4) {
        const float gx = dX[r * ELG_E + c];
        const float x = xy[r * 2], y = xy[r * 2 + 1];
        if (gWd && (r % N1) == 0) { v[4] = fmaf(gx, x, v[4]); v[5] = fmaf(gx, y, v[5]); v[6] += gx; }
        else { v[0] = fmaf(gx, x, v[0]); v[1] = fmaf(gx, y, v[1]); if (demand) v[2] = fmaf(gx, demand[r], v[2]); v[3] += gx; }
    }
    if (q) {
#pragma unroll
        for (int k = 0; k < 7; ++k) part[q - 1][k][c] = v[k];
    }
    __syncthreads();
    if (q) return;
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] += part[0][k][c] + part[1][k][c] + part[2][k][c];
    if (demand) { atomicAdd(gWn + c * 3, v[0]); atomicAdd(gWn + c * 3 + 1, v[1]); atomicAdd(gWn + c * 3 + 2, v[2]); }
    else { atomicAdd(gWn + c * 2, v[0]); atomicAdd(gWn + c * 2 + 1, v[1]); }
    atomicAdd(gbn + c, v[3]);
    if (gWd) { atomicAdd(gWd + c * 2, v[4]); atomicAdd(gWd + c * 2 + 1, v[5]); atomicAdd(gbd + c, v[6]); }
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
__global__ __launch_bounds__(512) void enc_fold_small_bwd_kernel(const float* __restrict__ enc, const float* __restrict__ gpb,
                                                                 const float* __restrict__ gwl, float* gbc, float* gWq_last,
                                                                 long rows, int rows_per_block, float scale) {
    // 512 threads = 128 channels x 4 row phases (a thread walks every fourth row of the block; partial sums meet in LDS)
    __shared__ float part[3][ELG_E];
    const int c = threadIdx.x & (ELG_E - 1), q = threadIdx.x >> 7;
    if (blockIdx.x == 0 && q == 0 && gwl && gWq_last) atomicAdd(gWq_last + c * (ELG_E + 1) + ELG_E, gwl[c]);
    if (!gpb) return;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float acc = 0.f;
    for (long r = r0 + q; r < r1; r += 4) acc = fmaf(enc[r * ELG_E + c], gpb[r], acc);
    if (q) part[q - 1][c] = acc;
    __syncthreads();
    if (q == 0) atomicAdd(gbc + c, (acc + part[0][c] + part[1][c] + part[2][c]) * scale);
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
        cmax = quarters_max(cmax);
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
    l = quarters_sum(l);
    const float inv = 1.0f / l;
    const int row = row0 + lo;
    if (row < N1) {
        *reinterpret_cast<float4*>(O + ((size_t)b * N1 + row) * ELG_E + h * 16 + 4 * hi) =
            make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
        if (hi == 0 && lse) lse[(size_t)blockIdx.y * N1 + row] = m + __logf(l);
    }
}

// self-attention backward, N1 <= 16 NT <= 128.  grid (B * 8), 8 waves (one row tile / key tile each: with 4 waves the 7 tiles of
// N1 = 101 took two rounds).
template <int NT>
__global__ __launch_bounds__(512) void enc_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dOg,
                                                           const float* __restrict__ Og, const float* __restrict__ lse,
                                                           float* __restrict__ dqkv, int N1) {
    constexpr int P = 20, ROWS = NT * 16, LD = 3 * ELG_E;
    // channel-major copies [channel][row] (pitch PT = 20 mod 32: conflict-free b128 reads): the operands "rows 4 hi .. 4 hi + 3 of
    // channel lo" of the dQ / dK / dV products are one ds_read_b128 instead of four scalar reads in front of four MFMAs
    constexpr int PT = ROWS + 4;
    __shared__ __attribute__((aligned(16))) float sQ[ROWS * P], sK[ROWS * P], sV[ROWS * P], sD[ROWS * P];
    __shared__ __attribute__((aligned(16))) float sQT[16 * PT], sKT[16 * PT], sDT[16 * PT];
    __shared__ __attribute__((aligned(16))) float sL[ROWS], sDel[ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int b = blockIdx.x >> 3, h = blockIdx.x & 7;
    const float* base = qkv + (size_t)b * N1 * LD + h * 16;
    const float* dOb = dOg + (size_t)b * N1 * ELG_E + h * 16;
    const float* Ob = Og + (size_t)b * N1 * ELG_E + h * 16;
    for (int idx = tid; idx < ROWS * 4; idx += 512) {
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
        sQT[(c4 + 0) * PT + row] = q.x; sQT[(c4 + 1) * PT + row] = q.y; sQT[(c4 + 2) * PT + row] = q.z; sQT[(c4 + 3) * PT + row] = q.w;
        sKT[(c4 + 0) * PT + row] = k.x; sKT[(c4 + 1) * PT + row] = k.y; sKT[(c4 + 2) * PT + row] = k.z; sKT[(c4 + 3) * PT + row] = k.w;
        sDT[(c4 + 0) * PT + row] = d.x; sDT[(c4 + 1) * PT + row] = d.y; sDT[(c4 + 2) * PT + row] = d.z; sDT[(c4 + 3) * PT + row] = d.w;
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
    for (int rt = wave; rt < NT; rt += 8) {
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
            const float4 kT = *reinterpret_cast<const float4*>(sKT + lo * PT + 16 * kt + 4 * hi);
            const float kTv[4] = {kT.x, kT.y, kT.z, kT.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = 16 * kt + 4 * hi + i;
                const float p = key < N1 ? __expf(S[i] * 0.25f - lr) : 0.f;
                const float ds = p * (dP[i] - del) * 0.25f;
                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kTv[i], ds, dq, 0, 0, 0);
            }
        }
        const int row = 16 * rt + lo;
        if (row < N1)
            *reinterpret_cast<float4*>(dqkv + ((size_t)b * N1 + row) * LD + h * 16 + 4 * hi) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    }
    // ---- keys on lanes: dK, dV of the wave's key tiles
    for (int kt = wave; kt < NT; kt += 8) {
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
            const float4 l4 = *reinterpret_cast<const float4*>(sL + 16 * rt + 4 * hi);
            const float4 e4 = *reinterpret_cast<const float4*>(sDel + 16 * rt + 4 * hi);
            const float4 dT = *reinterpret_cast<const float4*>(sDT + lo * PT + 16 * rt + 4 * hi);
            const float4 qT = *reinterpret_cast<const float4*>(sQT + lo * PT + 16 * rt + 4 * hi);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
            const float dTv[4] = {dT.x, dT.y, dT.z, dT.w}, qTv[4] = {qT.x, qT.y, qT.z, qT.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = kok ? __expf(S[i] * 0.25f - lv[i]) : 0.f;
                const float ds = p * (dP[i] - ev[i]) * 0.25f;
                dv = __builtin_amdgcn_mfma_f32_16x16x4f32(dTv[i], p, dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x4f32(qTv[i], ds, dk, 0, 0, 0);
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

// self-attention backward for N1 > 128 (training at CVRP/TSP-200 ... 1000: TSP/train.py:101-122 trains any size through one
// tape).  Same products and operand maps as enc_attn_bwd_kernel, but the (instance, head)'s Q, K, V, dO no longer fit LDS:
// a wave owns one 16-row tile (dQ) or one 16-key tile (dK, dV) and walks the other index in tiles of 16 whose operands it
// reads from global memory (the instance's 3 x N1 x 128 floats are L2-resident).  delta[row] = dO_h[row] . O_h[row] comes
// from enc_attn_delta_kernel.  grid (ceil(N1 / 64), B * 8), one wave per tile.
__global__ __launch_bounds__(256) void enc_attn_delta_kernel(const float* __restrict__ dOg, const float* __restrict__ Og,
                                                             float* __restrict__ delta, int N1, long rows) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // (row, head)
    if (idx >= rows * 8) return;
    const long row = idx >> 3;
    const int h = (int)(idx & 7);
    const float* d = dOg + row * ELG_E + h * 16;
    const float* o = Og + row * ELG_E + h * 16;
    float acc = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 16; c4 += 4)
        acc = dot4(*reinterpret_cast<const float4*>(d + c4), *reinterpret_cast<const float4*>(o + c4), acc);
    const long b = row / N1, n = row - b * N1;
    delta[(b * 8 + h) * N1 + n] = acc;
}

template <bool KV>
__global__ __launch_bounds__(256) void enc_attn_bwd_large_kernel(const float* __restrict__ qkv, const float* __restrict__ dOg,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta,
                                                                 float* __restrict__ dqkv, int N1) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int b = blockIdx.y >> 3, h = blockIdx.y & 7;
    const int t0 = (blockIdx.x * 4 + wave) * 16;                    // first row (dQ) / key (dK, dV) of this wave's tile
    if (t0 >= N1) return;
    constexpr int LD = 3 * ELG_E;
    const float* base = qkv + (size_t)b * N1 * LD + h * 16;
    const float* dOb = dOg + (size_t)b * N1 * ELG_E + h * 16;
    const float* lb = lse + (size_t)blockIdx.y * N1;
    const float* db = delta + (size_t)blockIdx.y * N1;
    const int nt = (N1 + 15) >> 4;
    if (!KV) {
        // ---- rows on lanes: dQ of the tile
        const int row = t0 + lo, rc = min(row, N1 - 1);
        const float mk = row < N1 ? 1.f : 0.f;
        float4 qB = *reinterpret_cast<const float4*>(base + (size_t)rc * LD + 4 * hi);
        float4 dB = *reinterpret_cast<const float4*>(dOb + (size_t)rc * ELG_E + 4 * hi);
        qB.x *= mk; qB.y *= mk; qB.z *= mk; qB.w *= mk;
        dB.x *= mk; dB.y *= mk; dB.z *= mk; dB.w *= mk;
        const float lr = row < N1 ? lb[rc] : __builtin_huge_valf(), del = row < N1 ? db[rc] : 0.f;
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nt; ++kt) {
            const int kc = min(16 * kt + lo, N1 - 1);
            const float4 kA = *reinterpret_cast<const float4*>(base + (size_t)kc * LD + ELG_E + 4 * hi);
            const float4 vA = *reinterpret_cast<const float4*>(base + (size_t)kc * LD + 2 * ELG_E + 4 * hi);
            float kTv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) kTv[i] = base[(size_t)min(16 * kt + 4 * hi + i, N1 - 1) * LD + ELG_E + lo];
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
                dq = __builtin_amdgcn_mfma_f32_16x16x4f32(kTv[i], ds, dq, 0, 0, 0);
            }
        }
        if (row < N1)
            *reinterpret_cast<float4*>(dqkv + ((size_t)b * N1 + row) * LD + h * 16 + 4 * hi) = make_float4(dq[0], dq[1], dq[2], dq[3]);
    } else {
        // ---- keys on lanes: dK, dV of the tile
        const int key = t0 + lo, kc = min(key, N1 - 1);
        const bool kok = key < N1;
        const float4 kB = *reinterpret_cast<const float4*>(base + (size_t)kc * LD + ELG_E + 4 * hi);
        const float4 vB = *reinterpret_cast<const float4*>(base + (size_t)kc * LD + 2 * ELG_E + 4 * hi);
        f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
        for (int rt = 0; rt < nt; ++rt) {
            const int r = 16 * rt + lo, rc = min(r, N1 - 1);
            const float mk = r < N1 ? 1.f : 0.f;
            float4 qA = *reinterpret_cast<const float4*>(base + (size_t)rc * LD + 4 * hi);
            float4 dA = *reinterpret_cast<const float4*>(dOb + (size_t)rc * ELG_E + 4 * hi);
            qA.x *= mk; qA.y *= mk; qA.z *= mk; qA.w *= mk;
            dA.x *= mk; dA.y *= mk; dA.z *= mk; dA.w *= mk;
            float lv[4], ev[4], dTv[4], qTv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r2 = 16 * rt + 4 * hi + i, r2c = min(r2, N1 - 1);
                const bool ok = r2 < N1;
                lv[i] = ok ? lb[r2c] : __builtin_huge_valf();
                ev[i] = ok ? db[r2c] : 0.f;
                dTv[i] = ok ? dOb[(size_t)r2c * ELG_E + lo] : 0.f;
                qTv[i] = ok ? base[(size_t)r2c * LD + lo] : 0.f;
            }
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
                const float p = kok ? __expf(S[i] * 0.25f - lv[i]) : 0.f;
                const float ds = p * (dP[i] - ev[i]) * 0.25f;
                dv = __builtin_amdgcn_mfma_f32_16x16x4f32(dTv[i], p, dv, 0, 0, 0);
                dk = __builtin_amdgcn_mfma_f32_16x16x4f32(qTv[i], ds, dk, 0, 0, 0);
            }
        }
        if (kok) {
            float* o = dqkv + ((size_t)b * N1 + key) * LD + h * 16 + 4 * hi;
            *reinterpret_cast<float4*>(o + ELG_E) = make_float4(dk[0], dk[1], dk[2], dk[3]);
            *reinterpret_cast<float4*>(o + 2 * ELG_E) = make_float4(dv[0], dv[1], dv[2], dv[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// All weight gradients of the encoder in ONE launch: dW_j[M,N] += alpha_j dY_j^T X_j over the batch*nodes rows, for every
// nn.Linear of every layer (and the decoder tables' projections), bias gradients = column sums of dY_j.
// Both operands are row-major with the contraction index (the row) outermost, which is exactly the operand order of
// v_mfma_f32_32x32x2_f32: lane (i = lane & 31, q = lane >> 5) feeds A[i][k = q] = dY[row + q][m] and B[k = q][j = i] =
// X[row + q][n] -- coalesced global loads, no LDS.  A lane loads two neighbouring m (and n) per row, so a wave owns a
// 64 x 64 block of dW as 2 x 2 MFMA tiles (rows / columns interleaved by two), a workgroup 128 x 128, and the row range is
// split over gridDim.y workgroups that accumulate with f32 atomics (the per-layer split-K GEMMs this replaces issued ~4x
// more atomics and ran 40 launches of ~22 us).
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(256) void enc_dw_kernel(const DwBatch bt) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, kq = lane >> 5;
    int j = 0;
    for (int t = 1; t < bt.njobs; ++t) j = (int)blockIdx.x >= bt.job[t].tile0 ? t : j;
    // copy the job out of the (SGPR-resident) argument array with a uniform select chain
    const float* dY = nullptr; const float* X = nullptr; float* dW = nullptr; float* db = nullptr;
    int ldy = 0, ldx = 0, ldw = 0, N = 0, tile0 = 0; float alpha = 1.f;
#pragma unroll
    for (int t = 0; t < DW_MAX_JOBS; ++t)
        if (t == j) {
            dY = bt.job[t].dY; X = bt.job[t].X; dW = bt.job[t].dW; db = bt.job[t].db; ldy = bt.job[t].ldy; ldx = bt.job[t].ldx;
            ldw = bt.job[t].ldw; N = bt.job[t].N; tile0 = bt.job[t].tile0; alpha = bt.job[t].alpha;
        }
    const int tile = blockIdx.x - tile0, ntn = N >> 7;
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int mw = (wave >> 1) * 64, nw = (wave & 1) * 64;             // the wave's 64 x 64 block inside the 128 x 128 tile
    const int m0 = mt * 128 + mw, n0 = nt * 128 + nw;
    const int kbeg = blockIdx.y * bt.rows_per_split, kend = min(bt.rows, kbeg + bt.rows_per_split);
    if (kbeg >= kend) return;
    // Round 5: the operands go through LDS in chunks of 16 rows.  Read straight from global memory (round 2), the two waves that
    // share a 64-column block of an operand both fetched it, and a CU's vector-memory path delivers cache LINES at a fixed pace
    // (~7 cycles per 128-byte line from L2, ~14 from the memory side: DESIGN 4.2) -- 32 line touches per k-step of a workgroup
    // against 4 x 64 cycles of MFMA per wave, with 4 workgroups per CU: the kernel sat at 37 % of the matrix pipe whatever the
    // prefetch depth.  Now a chunk's 16 x 128 floats per operand are loaded once per workgroup as whole lines (two 16-byte loads per
    // thread and operand, issued one chunk ahead), and a wave's fragments (rows 2 s + q, columns 2 li, 2 li + 1 of its block)
    // are conflict-free ds_read_b64.
    __shared__ __attribute__((aligned(16))) float sA[2][16 * 128], sB[2][16 * 128];
    const int tid = threadIdx.x;
    const float* gA = dY + mt * 128;
    const float* gB = X + nt * 128;
    float4 ra[2], rb[2];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, r = r0 + (idx >> 5), rc = min(r, kend - 1);
            float4 a_ = *reinterpret_cast<const float4*>(gA + (size_t)rc * ldy + 4 * (idx & 31));
            if (r >= kend) a_ = make_float4(0.f, 0.f, 0.f, 0.f);          // rows past the split: zero dY (X may be anything finite)
            ra[i] = a_;
            rb[i] = *reinterpret_cast<const float4*>(gB + (size_t)rc * ldx + 4 * (idx & 31));
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(&sA[buf][(idx >> 5) * 128 + 4 * (idx & 31)]) = ra[i];
            *reinterpret_cast<float4*>(&sB[buf][(idx >> 5) * 128 + 4 * (idx & 31)]) = rb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum0 = 0.f, bsum1 = 0.f;
    fetch(kbeg);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int k = kbeg; k < kend; k += 16) {
        const bool more = k + 16 < kend;                               // (uniform)
        if (more) fetch(k + 16);
        const float* pa = &sA[buf][kq * 128 + mw + 2 * li];
        const float* pb = &sB[buf][kq * 128 + nw + 2 * li];
        if (bt.bf16) {
            // bf16 mode (elg_encoder_args.precision = 1): the chunk's 16 rows are ONE k-step of v_mfma_f32_32x32x16_bf16 -- lane (li, q)
            // supplies rows 8 q .. 8 q + 7 of its two columns per operand, rounded to bf16; f32 accumulation; the bias sums stay f32
            const float* qa = &sA[buf][(8 * kq) * 128 + mw + 2 * li];
            const float* qb = &sB[buf][(8 * kq) * 128 + nw + 2 * li];
            float2 av[8], bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                av[j] = *reinterpret_cast<const float2*>(qa + j * 128);
                bv[j] = *reinterpret_cast<const float2*>(qb + j * 128);
                bsum0 += av[j].x; bsum1 += av[j].y;
            }
            const u32x4 a0 = {pk_bf16(av[0].x, av[1].x), pk_bf16(av[2].x, av[3].x), pk_bf16(av[4].x, av[5].x), pk_bf16(av[6].x, av[7].x)};
            const u32x4 a1 = {pk_bf16(av[0].y, av[1].y), pk_bf16(av[2].y, av[3].y), pk_bf16(av[4].y, av[5].y), pk_bf16(av[6].y, av[7].y)};
            const u32x4 b0 = {pk_bf16(bv[0].x, bv[1].x), pk_bf16(bv[2].x, bv[3].x), pk_bf16(bv[4].x, bv[5].x), pk_bf16(bv[6].x, bv[7].x)};
            const u32x4 b1 = {pk_bf16(bv[0].y, bv[1].y), pk_bf16(bv[2].y, bv[3].y), pk_bf16(bv[4].y, bv[5].y), pk_bf16(bv[6].y, bv[7].y)};
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), acc[1][1], 0, 0, 0);
        } else {
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const float2 av = *reinterpret_cast<const float2*>(pa + s2 * 256);
                const float2 bv = *reinterpret_cast<const float2*>(pb + s2 * 256);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
                bsum0 += av.x; bsum1 += av.y;
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // C/D layout of 32x32: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); tile (a, b) holds
    // dW[m0 + 2 row + a][n0 + 2 col + b].  The split's 128 x 128 partial goes to the scratch tile [split][tile][128][128] with
    // plain stores (a lane's two neighbouring columns as one 8-byte store: 256 contiguous bytes per row and half wave);
    // enc_dw_reduce_kernel adds the splits in a fixed order.  (Rounds 2 - 4 added the partials with f32 atomics: 17.5 M of them
    // per launch, each 4 bytes at an 8-byte stride -- ~2.2 M memory-side requests that a micro-benchmark prices at ~0.1 ms of the
    // kernel's 0.27, and a summation order that changed from run to run.)
    float* sc = bt.scratch + ((size_t)blockIdx.y * bt.ntiles + blockIdx.x) * (128 * 128);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kq;
            *reinterpret_cast<float2*>(sc + (mw + 2 * row + a) * 128 + nw + 2 * li) = make_float2(acc[a][0][r] * alpha, acc[a][1][r] * alpha);
        }
    if (db && nt == 0 && (wave & 1) == 0) {
        bsum0 = x32_sum(bsum0);
        bsum1 = x32_sum(bsum1);
        if (kq == 0) { atomicAdd(db + m0 + 2 * li, bsum0); atomicAdd(db + m0 + 2 * li + 1, bsum1); }
    }
}

// dW tile += sum over the row splits of the scratch tiles, split 0 first.  grid (ntiles, 16), 256 threads x 16 bytes.
__global__ __launch_bounds__(256) void enc_dw_reduce_kernel(const DwBatch bt, const int splits) {
    int j = 0;
    for (int t = 1; t < bt.njobs; ++t) j = (int)blockIdx.x >= bt.job[t].tile0 ? t : j;
    float* dW = nullptr; int ldw = 0, N = 0, tile0 = 0;
#pragma unroll
    for (int t = 0; t < DW_MAX_JOBS; ++t)
        if (t == j) { dW = bt.job[t].dW; ldw = bt.job[t].ldw; N = bt.job[t].N; tile0 = bt.job[t].tile0; }
    const int tile = blockIdx.x - tile0, ntn = N >> 7;
    const int mt = tile / ntn, nt = tile - mt * ntn;
    const int e = (blockIdx.y * 256 + threadIdx.x) * 4;                 // element of the 128 x 128 tile
    const int row = e >> 7, col = e & 127;
    const float* sc = bt.scratch + (size_t)blockIdx.x * (128 * 128) + e;
    const size_t stride = (size_t)bt.ntiles * (128 * 128);
    float4 acc = *reinterpret_cast<const float4*>(sc);
    for (int sp = 1; sp < splits; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(sc + sp * stride);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* o = dW + (size_t)(mt * 128 + row) * ldw + nt * 128 + col;    // (ldw may be odd: Wq_last is (128,129))
    o[0] += acc.x; o[1] += acc.y; o[2] += acc.z; o[3] += acc.w;
}

int DwList::add(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, int M, int N, float* db, float alpha) {
    if (bt.njobs >= DW_MAX_JOBS) {
        const int rc = launch();
        if (rc != ELG_OK) return rc;
    }
    // (enc_dw_kernel stages both operands with 16-byte loads: leading dimensions in multiples of 4 floats, 16-byte aligned bases)
    if ((M & 127) || (N & 127)) return fail(ELG_EINVAL, "encoder bwd: dW shapes must be multiples of 128");
    if ((ldy & 3) || (ldx & 3) || (reinterpret_cast<uintptr_t>(dY) & 15) || (reinterpret_cast<uintptr_t>(X) & 15))
        return fail(ELG_EINVAL, "encoder bwd: dW operands need leading dimensions in multiples of 4 floats and 16-byte aligned bases");
    DwJob& j = bt.job[bt.njobs++];
    j.dY = dY; j.X = X; j.dW = dW; j.db = db; j.ldy = ldy; j.ldx = ldx; j.ldw = ldw; j.M = M; j.N = N; j.alpha = alpha;
    j.tile0 = bt.ntiles;
    bt.ntiles += (M >> 7) * (N >> 7);
    return ELG_OK;
}
int DwList::launch() {
    if (bt.njobs == 0) return ELG_OK;
    // row split: ~3 workgroups per CU overall, at least 128 rows each (measured at the bench shape with the partial tiles going to
    // scratch: 256 workgroups 272 us, 486 / 512 -- one generation of the two that fit a CU -- 221 / 223, 768 191, 1024 194; the
    // kernel holds the matrix pipe at ~60 % whichever way the rows are cut)
    static const int target = [] {
        const char* e = std::getenv("ELG_DW_WGS");
        if (e && atoi(e) > 0) return atoi(e);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 3 * cus;
    }();
    int splits = (int)max(1L, min(rows / 128, (long)(target / bt.ntiles)));
    int rps = (int)((rows + splits - 1) / splits);
    rps = (rps + 15) / 16 * 16;
    splits = (int)((rows + rps - 1) / rps);
    bt.rows = (int)rows; bt.rows_per_split = rps;
    if (!scratch || (long)bt.ntiles * (128 * 128) > scratch_floats) return fail(ELG_EINVAL, "encoder bwd: weight-gradient scratch too small");
    // (an ELG_DW_WGS beyond what the scratch holds: fewer, longer row splits instead of a failed backward)
    const int cap = (int)(scratch_floats / ((long)bt.ntiles * (128 * 128)));
    if (splits > cap) {
        splits = cap;
        rps = (int)((rows + splits - 1) / splits);
        rps = (rps + 15) / 16 * 16;
        splits = (int)((rows + rps - 1) / rps);
        bt.rows_per_split = rps;
    }
    bt.scratch = scratch;
    (void)hipGetLastError();
    hipLaunchKernelGGL(enc_dw_kernel, dim3(bt.ntiles, splits), dim3(256), 0, s, bt);
    hipLaunchKernelGGL(enc_dw_reduce_kernel, dim3(bt.ntiles, 16), dim3(256), 0, s, bt, splits);
    bt.njobs = 0; bt.ntiles = 0;
    return launch_status("enc_dw");
}

// ------------------------------------------------------------------------------------------------------------------
// workspace layout (floats)
EncWs enc_ws(int B, int N1, int n_layers, int ff, int save) {
    EncWs w;
    if (n_layers == 0) { w = EncWs{}; w.R = (long)B * N1; return w; }
    w.R = (long)B * N1;
    long o = 0;
    w.X0 = o; o += w.R * ELG_E;
    w.tmp = o; o += w.R * ELG_E;
    w.P = o; o += (N1 <= 128) ? w.R * ELG_E * (ff >> 7) : 0;        // FFN2 partial sums of the fused path (one per hidden slice)
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

EncWs2 enc_ws2(int B, int N1, int n_layers, int ff) {
    EncWs2 w{};
    w.R = (long)B * N1;
    long o = 0;
    w.gX = o; o += w.R * ELG_E;
    w.gO = o; o += w.R * ELG_E;
    w.gT = o; o += w.R * ELG_E;
    w.lay0 = o; w.lay_stride = w.R * (5 * ELG_E + ff); o += w.lay_stride * n_layers;
    w.delta = o; o += (N1 > 128) ? (long)B * 8 * N1 : 0;
    if (N1 <= 128) {
        w.PX = o; o += ENC_PX * w.R * ELG_E;
        w.P1 = o; o += (long)(ff >> 7) * w.R * ELG_E;
        w.WT = o; w.wt_stride = 4L * ELG_E * ELG_E + 2L * ELG_E * ff; o += w.wt_stride * n_layers;
    }
    // partial tiles of the grouped weight-gradient launch: splits * ntiles <= max(3 CUs, ntiles) by default, ntiles = 4 + 2 ff / 128
    // per layer (12 at ff = 512, 20 at ff = 1024) + 5 table jobs, at most DW_MAX_JOBS jobs per launch; DwList::launch clamps the
    // row splits to what fits here
    w.DW = o; w.dw_floats = (1024L + 12L * n_layers + 5) * (128 * 128); o += w.dw_floats;
    w.total = o;
    return w;
}

int launch_fold_small_bwd(const float* enc, const float* gpb, const float* gwl, float* gbc, float* gWq_last, long rows, float scale,
                          hipStream_t s) {
    const int rpb = 64;
    (void)hipGetLastError();
    hipLaunchKernelGGL(enc_fold_small_bwd_kernel, dim3(gpb ? (unsigned)((rows + rpb - 1) / rpb) : 1u), dim3(512), 0, s, enc, gpb, gwl, gbc,
                       gWq_last, rows, rpb, scale);
    return launch_status("enc_fold_small_bwd");
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

extern "C" int64_t elg_encoder_bwd_ws_floats(int B, int N1, int n_layers, int ff_hidden) {
    if (B <= 0 || N1 <= 0 || n_layers <= 0 || ff_hidden <= 0) return 0;
    // d x / d att / d x1 (shared) + per layer: dS2, dH, dS1, dQKV (kept until the grouped weight-gradient launch)
    // + delta[b, h, n] of the N1 > 128 attention backward | the fused path's partial sums and transposed weights
    return enc_ws2(B, N1, n_layers, ff_hidden).total;
}

static int check_enc_args(const elg_encoder_args* a) {
    if (!a) return fail(ELG_EINVAL, "encoder: null args");
    if (a->B <= 0 || a->N1 < 4) return fail(ELG_EINVAL, "encoder: need B > 0, N1 >= 4");
    if (a->n_layers < 0 || a->n_layers > ELG_ENC_MAX_LAYERS) return fail(ELG_EINVAL, "encoder: 0 .. 8 layers");
    if (a->precision != 0 && a->precision != 1) return fail(ELG_EINVAL, "encoder: precision 0 (f32) or 1 (bf16 GEMM operands)");
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
    if (enc_fused_ok(a)) return enc_fused_fwd(a, s);
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

extern "C" int elg_encoder_bwd(const elg_encoder_bwd_args* ba, void* stream) {
    if (!ba) return fail(ELG_EINVAL, "encoder bwd: null args");
    const elg_encoder_args* a = &ba->fwd;
    ENC_TRY(check_enc_args(a))
    if (!a->save || a->n_layers < 1) return fail(ELG_EINVAL, "encoder bwd: the forward must have run with save = 1");
    hipStream_t s = (hipStream_t)stream;
    const int B = a->B, N1 = a->N1, FF = a->ff_hidden;
    const bool tsp = a->problem == ELG_PROBLEM_TSP;
    const EncWs w = enc_ws(B, N1, a->n_layers, FF, 1);
    const long R = w.R;
    if (!ba->ws2 || ba->ws2_floats < elg_encoder_bwd_ws_floats(B, N1, a->n_layers, FF)) return fail(ELG_EINVAL, "encoder bwd: scratch too small");
    const elg_enc_weights& G = ba->G;
    const EncWs2 w2 = enc_ws2(B, N1, a->n_layers, FF);
    float* gX = ba->ws2 + w2.gX;
    float* gO = ba->ws2 + w2.gO;
    float* gT = ba->ws2 + w2.gT;
    float* lay2 = ba->ws2 + w2.lay0;
    const long lay2_stride = w2.lay_stride;
    float* ws = a->ws;
    DwList dw(R, s, ba->ws2 + w2.DW, w2.dw_floats);
    dw.bt.bf16 = (a->precision == 1 && enc_fused_ok(a)) ? 1 : 0;          // the bf16 mode's weight gradients (N1 <= 128)
    if (enc_fused_ok(a)) return enc_fused_bwd(ba, dw, s);
    const float inv_sqrt_e = 0.08838834764831845f;
    // N1 <= 128: row block = instance, the norm backwards ride in GEMM epilogues.  N1 > 128: 128-row blocks, the add & norm
    // backwards are the stand-alone kernel (per (instance, channel) over the node axis), attention backward from global memory
    const bool aligned = N1 <= 128;
    float* delta = ba->ws2 + w2.delta;
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
        ENC_TRY(dw.add(ba->gK, ELG_E, a->enc, ELG_E, (float*)G.dec_Wk, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
    }
    if (ba->gV) {
        ENC_TRY(need(G.dec_Wv, "d Wv"))
        ENC_TRY(acc_into_gX(ba->gV, a->W.dec_Wv, ELG_E, W_KN, 1.f))
        ENC_TRY(dw.add(ba->gV, ELG_E, a->enc, ELG_E, (float*)G.dec_Wv, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
    }
    if (ba->gQ1) {
        ENC_TRY(need(G.dec_Wq_last, "d Wq_last"))
        const int ldq = tsp ? ELG_E : ELG_E + 1;
        ENC_TRY(acc_into_gX(ba->gQ1, a->W.dec_Wq_last, ldq, W_KN, 1.f))
        ENC_TRY(dw.add(ba->gQ1, ELG_E, a->enc, ELG_E, (float*)G.dec_Wq_last, ldq, ELG_E, ELG_E, nullptr, 1.f))
    }
    if (tsp && ba->gQ2) {
        ENC_TRY(need(G.dec_Wq_first, "d Wq_first"))
        ENC_TRY(acc_into_gX(ba->gQ2, a->W.dec_Wq_first, ELG_E, W_KN, 1.f))
        ENC_TRY(dw.add(ba->gQ2, ELG_E, a->enc, ELG_E, (float*)G.dec_Wq_first, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
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
        ENC_TRY(dw.add(a->enc, ELG_E, ba->gPK, ELG_E, (float*)G.dec_Wc, ELG_E, ELG_E, ELG_E, nullptr, inv_sqrt_e))
    } else if (ba->gpb) return fail(ELG_EINVAL, "encoder bwd: gpb without gPK");
    if (ba->gpb || (ba->gwl && !tsp)) {
        if (ba->gpb) ENC_TRY(need(G.dec_bc, "d bc"))
        ENC_TRY(launch_fold_small_bwd(a->enc, ba->gpb, tsp ? nullptr : ba->gwl, (float*)G.dec_bc, (float*)G.dec_Wq_last, R, inv_sqrt_e, s))
    }
    if (!have) return fail(ELG_EINVAL, "encoder bwd: no cotangent given");
    // ---- layers, last to first
    for (int l = a->n_layers - 1; l >= 0; --l) {
        const elg_enc_layer& L = a->W.layer[l];
        const elg_enc_layer& GL = G.layer[l];
        float* gS = lay2 + lay2_stride * l;
        float* gH = gS + R * ELG_E;
        float* gY = gH + R * FF;
        float* dQKV = gY + R * ELG_E;
        float* lb = ws + w.layer0 + w.layer_stride * l;
        const float *QKV = lb + w.oQKV, *O = lb + w.oO, *LSE = lb + w.oLSE, *XH1 = lb + w.oXH1, *RS1 = lb + w.oRS1;
        const float *X1 = lb + w.oX1, *H = lb + w.oH, *XH2 = lb + w.oXH2, *RS2 = lb + w.oRS2;
        const float* Xin = (l == 0) ? ws + w.X0 : ws + w.layer0 + w.layer_stride * (l - 1) + w.oXout;
        // second add & norm (the top layer's: the layers below get theirs from the epilogue of the GEMM that produces d x)
        if (l == a->n_layers - 1)
            ENC_TRY(elg_add_instnorm_bwd(gX, XH2, RS2, L.g2, gS, (float*)GL.g2, (float*)GL.b2, B, N1, ELG_E, stream))
        ENC_TRY(dw.add(gS, ELG_E, H, FF, (float*)GL.W2, FF, ELG_E, FF, (float*)GL.bf2, 1.f))
        {   // dH = (dS2 W2) * [h > 0]
            EncGemm g = gemm_base(gS, ELG_E, FF, ELG_E, R, N1, aligned);
            g.W[0] = L.W2; g.ldw = FF; g.wmode = W_KN; g.wblk = ELG_E; g.C[0] = gH; g.ldc = FF; g.epi = EPI_RELUMASK; g.R = H; g.ldr = FF;
            ENC_TRY(launch_gemm(g, s))
        }
        ENC_TRY(dw.add(gH, FF, X1, ELG_E, (float*)GL.W1, ELG_E, FF, ELG_E, (float*)GL.bf1, 1.f))
        {   // d x1 = dS2 + dH W1, then the first add & norm's backward in the epilogue: dY
            EncGemm g = gemm_base(gH, FF, ELG_E, FF, R, N1, aligned);
            g.W[0] = L.W1; g.ldw = ELG_E; g.wmode = W_KN; g.wblk = FF; g.C[0] = gY; g.ldc = ELG_E; g.epi = EPI_ADD_NORMBWD; g.R = gS; g.ldr = ELG_E;
            g.xhat = const_cast<float*>(XH1); g.rstd = const_cast<float*>(RS1); g.gamma = L.g1;
            g.dgamma = (float*)GL.g1; g.dbeta = (float*)GL.b1;
            if (!aligned) { g.epi = EPI_ADD; g.C[0] = gT; }
            ENC_TRY(launch_gemm(g, s))
            if (!aligned) ENC_TRY(elg_add_instnorm_bwd(gT, XH1, RS1, L.g1, gY, (float*)GL.g1, (float*)GL.b1, B, N1, ELG_E, stream))
        }
        ENC_TRY(dw.add(gY, ELG_E, O, ELG_E, (float*)GL.Wc, ELG_E, ELG_E, ELG_E, (float*)GL.bc, 1.f))
        {   // d att = dY Wc
            EncGemm g = gemm_base(gY, ELG_E, ELG_E, ELG_E, R, N1, aligned);
            g.W[0] = L.Wc; g.ldw = ELG_E; g.wmode = W_KN; g.wblk = ELG_E; g.C[0] = gO; g.ldc = ELG_E;
            ENC_TRY(launch_gemm(g, s))
        }
        (void)hipGetLastError();
        if (!aligned) {
            hipLaunchKernelGGL(enc_attn_delta_kernel, dim3((unsigned)((R * 8 + 255) / 256)), dim3(256), 0, s, gO, O, delta, N1, R);
            hipLaunchKernelGGL((enc_attn_bwd_large_kernel<false>), dim3((N1 + 63) / 64, B * 8), dim3(256), 0, s, QKV, gO, LSE, delta, dQKV, N1);
            hipLaunchKernelGGL((enc_attn_bwd_large_kernel<true>), dim3((N1 + 63) / 64, B * 8), dim3(256), 0, s, QKV, gO, LSE, delta, dQKV, N1);
        } else {
            const int nt = (N1 + 15) / 16;
            if (nt <= 2) hipLaunchKernelGGL((enc_attn_bwd_kernel<2>), dim3(B * 8), dim3(512), 0, s, QKV, gO, O, LSE, dQKV, N1);
            else if (nt <= 4) hipLaunchKernelGGL((enc_attn_bwd_kernel<4>), dim3(B * 8), dim3(512), 0, s, QKV, gO, O, LSE, dQKV, N1);
            else if (nt <= 7) hipLaunchKernelGGL((enc_attn_bwd_kernel<7>), dim3(B * 8), dim3(512), 0, s, QKV, gO, O, LSE, dQKV, N1);
            else hipLaunchKernelGGL((enc_attn_bwd_kernel<8>), dim3(B * 8), dim3(512), 0, s, QKV, gO, O, LSE, dQKV, N1);
        }
        ENC_TRY(launch_status("enc_attn_bwd"))
        ENC_TRY(dw.add(dQKV, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wq, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        ENC_TRY(dw.add(dQKV + ELG_E, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wk, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        ENC_TRY(dw.add(dQKV + 2 * ELG_E, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wv, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        {   // d x = dY + dQ Wq + dK Wk + dV Wv; below the first layer d x is the cotangent of layer l - 1's second add & norm,
            // whose backward runs in the epilogue (-> that layer's dS2)
            EncGemm g = gemm_base(dQKV, 3 * ELG_E, ELG_E, 3 * ELG_E, R, N1, aligned);
            g.W[0] = L.Wq; g.W[1] = L.Wk; g.W[2] = L.Wv; g.ldw = ELG_E; g.wmode = W_KN; g.wblk = ELG_E; g.C[0] = gX; g.ldc = ELG_E;
            g.epi = EPI_ADD; g.R = gY; g.ldr = ELG_E;
            if (l > 0 && aligned) {
                const float* lbp = ws + w.layer0 + w.layer_stride * (l - 1);
                g.epi = EPI_ADD_NORMBWD; g.C[0] = lay2 + lay2_stride * (l - 1);
                g.xhat = const_cast<float*>(lbp + w.oXH2); g.rstd = const_cast<float*>(lbp + w.oRS2); g.gamma = a->W.layer[l - 1].g2;
                g.dgamma = (float*)G.layer[l - 1].g2; g.dbeta = (float*)G.layer[l - 1].b2;
            }
            ENC_TRY(launch_gemm(g, s))
            if (l > 0 && !aligned) {
                const float* lbp = ws + w.layer0 + w.layer_stride * (l - 1);
                ENC_TRY(elg_add_instnorm_bwd(gX, lbp + w.oXH2, lbp + w.oRS2, a->W.layer[l - 1].g2, lay2 + lay2_stride * (l - 1),
                                             (float*)G.layer[l - 1].g2, (float*)G.layer[l - 1].b2, B, N1, ELG_E, stream))
            }
        }
    }
    // ---- input embeddings
    {
        const int rpb = 32;
        (void)hipGetLastError();
        hipLaunchKernelGGL(enc_embed_bwd_kernel, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(512), 0, s, a->xy,
                           tsp ? nullptr : a->demand, gX, tsp ? nullptr : (float*)G.emb_depot_w, tsp ? nullptr : (float*)G.emb_depot_b,
                           (float*)G.emb_w, (float*)G.emb_b, N1, R, rpb);
        ENC_TRY(launch_status("enc_embed_bwd"))
    }
    return dw.launch();          // every weight gradient of the call in one grouped launch
}
