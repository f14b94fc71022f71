// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, fmaf-chain numerics) for the
// encoder's projections / feed-forward layers and their backward passes (reference CVRP/models.py:240-269,
// 550-561 executed as nn.Linear there).
//
//     C[M,N] (+)= op(A)[M,K] * op(B)[K,N]  (+ bias[N]) (ReLU)
//     op(A) = A (M x K, row-major, lda)  or  A^T (A stored K x M)        -- transA
//     op(B) = B (K x N, row-major, ldb)  or  B^T (B stored N x K)        -- transB  (nn.Linear weights)
//
// Tiling: 64 x 64 x 32 per workgroup of 4 waves; every wave owns a 32 x 32 accumulator (16 AGPR/VGPR),
// 16 MFMAs per K-tile.  Both operands are staged k-major in LDS ([32][64+pad]) so that a fragment read
// is 64 consecutive floats per k (conflict-free ds_read_b32): lane l feeds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31].  Split-K (gridDim.z) with f32 atomics covers the weight-gradient shape
// (K = batch*nodes = 6464, M,N = 128..512).
#include "elg_common.h"
#include "../../include/elg_hip.h"
#include <string>

namespace elg {
int fail(int code, const std::string& msg);
int launch_status(const char* what);

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmBatch {          // two-level batch: element (b1, b2) at A + b1 sA1 + b2 sA2 (likewise B, C); n_inner = extent of b2
    int nsplit, n_inner;
    long sA1, sA2, sB1, sB2, sC1, sC2;
};

constexpr int BM = 64, BN = 64, BK = 32, LDT = 68;   // LDT: padded leading dimension of the k-major tiles

template <bool TRANS>
__device__ __forceinline__ void stage_tile(const float* __restrict__ G, int ld, int r0, int k0, int R, int K,
                                           float* __restrict__ S, int tid, bool vec) {
    // vec: rows are 16-byte aligned (ld % 4 == 0 and an aligned base); otherwise four scalar loads per group
    // fills S[k][r] (k < BK, r < 64) with op(G)[r0 + r][k0 + k]; out-of-range elements are zero
    if (!TRANS) {
        // G is (R x K) row-major: a thread reads 4 consecutive k of one row, writes them k-major
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * 256;            // 512 float4 slots = 64 rows x 8
            const int r = idx >> 3, kq = (idx & 7) * 4;
            const int gr = r0 + r, gk = k0 + kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < R) {
                if (vec && gk + 3 < K) v = *reinterpret_cast<const float4*>(G + (size_t)gr * ld + gk);
                else {
                    if (gk + 3 < K) v.w = G[(size_t)gr * ld + gk + 3];
                    if (gk < K) v.x = G[(size_t)gr * ld + gk];
                    if (gk + 1 < K) v.y = G[(size_t)gr * ld + gk + 1];
                    if (gk + 2 < K) v.z = G[(size_t)gr * ld + gk + 2];
                }
            }
            S[(kq + 0) * LDT + r] = v.x; S[(kq + 1) * LDT + r] = v.y;
            S[(kq + 2) * LDT + r] = v.z; S[(kq + 3) * LDT + r] = v.w;
        }
    } else {
        // G is (K x R) row-major: rows of the tile are contiguous along r
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * 256;            // 512 float4 slots = 32 k x 16
            const int k = idx >> 4, rq = (idx & 15) * 4;
            const int gk = k0 + k, gr = r0 + rq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < K) {
                if (vec && gr + 3 < R) v = *reinterpret_cast<const float4*>(G + (size_t)gk * ld + gr);
                else {
                    if (gr + 3 < R) v.w = G[(size_t)gk * ld + gr + 3];
                    if (gr < R) v.x = G[(size_t)gk * ld + gr];
                    if (gr + 1 < R) v.y = G[(size_t)gk * ld + gr + 1];
                    if (gr + 2 < R) v.z = G[(size_t)gk * ld + gr + 2];
                }
            }
            *reinterpret_cast<float4*>(S + k * LDT + rq) = v;
        }
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, const float* __restrict__ bias,
                                                       int M, int N, int K, int lda, int ldb, int ldc, int relu,
                                                       int k_per_split, float* __restrict__ a_rowsum, float alpha,
                                                       GemmBatch gb) {
    __shared__ __attribute__((aligned(16))) float sA[BK * LDT];
    __shared__ __attribute__((aligned(16))) float sB[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // blockIdx.z = (batch index) * nsplit + k split; the batch index is two-level (outer, inner) with its own strides
    const int zs = blockIdx.z % gb.nsplit, zb = blockIdx.z / gb.nsplit;
    {
        const int b1 = zb / gb.n_inner, b2 = zb - b1 * gb.n_inner;
        A += b1 * gb.sA1 + b2 * gb.sA2;
        B += b1 * gb.sB1 + b2 * gb.sB2;
        C += b1 * gb.sC1 + b2 * gb.sC2;
    }
    const int kbeg = zs * k_per_split, kend = min(K, kbeg + k_per_split);
    const bool vecA = !(lda & 3) && !((uintptr_t)A & 15), vecB = !(ldb & 3) && !((uintptr_t)B & 15);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // optional: row sums of op(A) over this split's k range (= the bias gradient when C = dY^T X), taken from the
    // staged tiles by the workgroups of the first column tile
    const bool do_sum = a_rowsum != nullptr && blockIdx.x == 0;
    float rsum = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        // op(A)[m][k]: !TA -> A stored (M x K) ; TA -> A stored (K x M)
        stage_tile<TA>(A, lda, m0, k0, M, kend, sA, tid, vecA);
        // op(B)[k][n]: !TB -> B stored (K x N) = "k-major rows" -> the transposed-source path; TB -> (N x K)
        stage_tile<!TB>(B, ldb, n0, k0, N, kend, sB, tid, vecB);
        __syncthreads();
        if (do_sum && tid < BM) {
#pragma unroll 8
            for (int kk = 0; kk < BK; ++kk) rsum += sA[kk * LDT + tid];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = sA[(kk + (lane >> 5)) * LDT + wm + (lane & 31)];
            const float b = sB[(kk + (lane >> 5)) * LDT + wn + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (do_sum && tid < BM && m0 + tid < M) atomicAdd(a_rowsum + m0 + tid, rsum);
    // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int col = n0 + wn + (lane & 31);
    const float bv = (bias && col < N && zs == 0) ? bias[col] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = m0 + wm + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
            float v = acc[reg] * alpha + bv;
            if (gb.nsplit > 1) atomicAdd(C + (size_t)row * ldc + col, v);
            else {
                if (relu) v = fmaxf(v, 0.f);
                C[(size_t)row * ldc + col] = v;
            }
        }
    }
}

}  // namespace elg

using namespace elg;

static int gemm_launch(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                       int ldc, int transA, int transB, int relu, int split_k, float* a_rowsum, float alpha, int n_outer,
                       int n_inner, const long (&st)[6], void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || n_outer <= 0 || n_inner <= 0) return fail(ELG_EINVAL, "gemm: empty problem");
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && relu) return fail(ELG_EINVAL, "gemm: ReLU epilogue needs split_k == 1");
    // (rows that are not 16-byte aligned -- odd leading dimensions such as N + 1 nodes -- are staged with scalar loads)
    int kps = (K + split_k - 1) / split_k;
    kps = (kps + BK - 1) / BK * BK;
    const int splits = (K + kps - 1) / kps;
    const long nz = (long)splits * n_outer * n_inner;
    if (nz > 65535) return fail(ELG_EINVAL, "gemm: too many batches x splits for one launch");
    GemmBatch gb{splits, n_inner, st[0], st[1], st[2], st[3], st[4], st[5]};
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, (unsigned)nz), block(256);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, A, B, C, bias, M, N, K, lda, ldb, ldc, relu, kps, a_rowsum, alpha, gb);
    else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, A, B, C, bias, M, N, K, lda, ldb, ldc, relu, kps, a_rowsum, alpha, gb);
    else if (transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, A, B, C, bias, M, N, K, lda, ldb, ldc, relu, kps, a_rowsum, alpha, gb);
    else hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, A, B, C, bias, M, N, K, lda, ldb, ldc, relu, kps, a_rowsum, alpha, gb);
    return launch_status("gemm_f32");
}

extern "C" __attribute__((visibility("hidden"))) int elg_gemm_f32_alpha(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                                  int lda, int ldb, int ldc, int transA, int transB, int relu, int split_k, float* a_rowsum,
                                  float alpha, void* stream) {
    const long st[6] = {0, 0, 0, 0, 0, 0};
    return gemm_launch(A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, relu, split_k, a_rowsum, alpha, 1, 1, st, stream);
}

namespace elg {
// internal (csrc/elg_bwd_internal.h): the batched product with the contraction split over `split_k` workgroups that ACCUMULATE into
// a caller-zeroed C with f32 atomics -- the row reductions of the large-instance decoder backward (K = decode rows, M, N small)
int gemm_f32_batched_splitk(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                            int transB, int n_outer, int n_inner, long sA_outer, long sA_inner, long sB_outer, long sB_inner,
                            long sC_outer, long sC_inner, float alpha, int split_k, void* stream) {
    const long st[6] = {sA_outer, sA_inner, sB_outer, sB_inner, sC_outer, sC_inner};
    return gemm_launch(A, B, C, nullptr, M, N, K, lda, ldb, ldc, transA, transB, 0, split_k, nullptr, alpha, n_outer, n_inner, st, stream);
}
}  // namespace elg

extern "C" int elg_gemm_f32_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                    int transA, int transB, int n_outer, int n_inner, int64_t sA_outer, int64_t sA_inner,
                                    int64_t sB_outer, int64_t sB_inner, int64_t sC_outer, int64_t sC_inner, float alpha,
                                    void* stream) {
    const long st[6] = {(long)sA_outer, (long)sA_inner, (long)sB_outer, (long)sB_inner, (long)sC_outer, (long)sC_inner};
    return gemm_launch(A, B, C, nullptr, M, N, K, lda, ldb, ldc, transA, transB, 0, 1, nullptr, alpha, n_outer, n_inner, st, stream);
}

extern "C" int elg_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                            int lda, int ldb, int ldc, int transA, int transB, int relu, int split_k, float* a_rowsum,
                            void* stream) {
    return elg_gemm_f32_alpha(A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, relu, split_k, a_rowsum, 1.0f, stream);
}
