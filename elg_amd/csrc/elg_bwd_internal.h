// Internal (not part of the C ABI): pieces of the decoder backward shared between csrc/elg_bwd.hip and csrc/elg_dbwd.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace elg {

// Optional epilogue: the query-gather backward (d Q1[node] = sum of dQ over the rows whose query was gathered at that node,
// d Q2 likewise for the tour's first node, d wl = sum_r load_r dQ_r) taken from the dQ tile while it is still in
// registers: LDS float atomics into a (N1 + 1) x 16 accumulator per (instance, head), flushed once per workgroup with
// global atomics.  dQ then never reaches memory (it was 0.4 GB written + a one-hot GEMM over (B,R,N1+1) to read it back).
struct GlimpseSeg {
    const int* idx_prev;    // (B,R) node the row's query was gathered at, or NULL: no epilogue
    const int* idx_first;   // (B,R) TSP: first node of the tour, or NULL
    const float* load;      // (B,load_rows) CVRP: vehicle load of the row, or NULL
    float* dQ1;             // (B,N1,128) accumulated (caller zeroes)
    float* dQ2;             // (B,N1,128) or NULL
    float* dwl;             // (128) or NULL
    long long load_rows;
    const float* lse;       // (B,rowQ_rows,8) log2-sum-exp of the glimpse scores per (row, head) (mask-row mode), or NULL
    int accumulate;         // 1: dK / dV are added to dKp / dVp (B,N1,128, caller zeroes) instead of written per split
    const int* T_dev;       // device-resident number of decode steps: R = min(R, T_dev[0] * M) inside the kernel, or NULL
    int M;                  // trajectories per instance (with T_dev / tlen)
    const int* tlen;        // (B,M) steps per trajectory, or NULL: with it (time-major rows) only the tiles of the decode steps
    int t0;                 //   t0 .. max_m tlen[b,m] - 1 are walked -- the rows outside carry dO = 0 and contribute nothing
    int mfma_mode;          // 0: f32 MFMAs; split-bf16 products (mask rows + lse + epilogue): 1 = 2 terms, 2 = 3-term scores + 2-term linear, 3 = 1-term (bf16-forward) scores + 2-term linear
};

// first 16-row tile that holds a row of decode step t0 (rows r = t M + m)
__host__ __device__ inline int live_tile_first(int t0, int M) { return (t0 * M) >> 4; }


int glimpse_bwd_launch(const float* rowA, const unsigned long long* mk, const float* dO, const float* rowO, const float* rowQ,
                       const float* Kmat, const float* Vmat, float* dQ, float* dK_part, float* dV_part, int B, int R, int N1,
                       long long rowA_rows, long long rowO_rows, long long rowQ_rows, int splits, const GlimpseSeg& seg,
                       hipStream_t s);

// csrc/elg_gemm.hip: elg_gemm_f32_batched with the contraction split over `split_k` workgroups accumulating into a zeroed C
int gemm_f32_batched_splitk(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int transA,
                            int transB, int n_outer, int n_inner, long sA_outer, long sA_inner, long sB_outer, long sB_inner,
                            long sC_outer, long sC_inner, float alpha, int split_k, void* stream);

}  // namespace elg
