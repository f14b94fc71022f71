// Training-step glue kernels: the REINFORCE/POMO loss (reference CVRP/train.py:112-121, TSP/train.py:107-118),
// the per-row cotangents the decoder backward consumes, and the Adam update (reference train.py:101 torch.optim.Adam
// with weight_decay) on flat parameter / gradient buffers.  Each replaces a chain of 10-30 tiny framework kernels
// whose launch overhead left the GPU idle between the rollout and its backward.
#include "elg_common.h"
#include "../../include/elg_hip.h"
#include <string>

namespace elg {
int fail(int code, const std::string& msg);
int launch_status(const char* what);

__device__ __forceinline__ float block_sum(float v, float* red, int tid, int nthreads) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (nthreads >> 6); ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red, int tid, int nthreads) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = red[0];
    for (int w = 1; w < (nthreads >> 6); ++w) s = fmaxf(s, red[w]);
    return s;
}

// One workgroup per instance.  advantage = reward - mean_m reward;  J[b,m] = -advantage * sum_t log p[b,t,m];
// scaled variant divides by max_m advantage.  Outputs per instance: sum_m J (raw and scaled), the max, and
// coef[b,m] = d(sum J)/d(log-prob sum) = -advantage (/ max) for both variants; the caller picks the variant
// (CVRP: always scaled; TSP: scaled only if no instance has a zero max) and applies the 1/(B*M) of the mean.
__global__ __launch_bounds__(256) void pomo_loss_kernel(const float* __restrict__ probs, const float* __restrict__ reward,
                                                        int T, int M, long long p_bstride, long long p_tstride,
                                                        float* __restrict__ Jraw, float* __restrict__ Jscaled,
                                                        float* __restrict__ amax_out, float* __restrict__ coef_raw,
                                                        float* __restrict__ coef_scaled) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pb = probs + (size_t)b * p_bstride;
    float rsum = 0.f;
    for (int m = tid; m < M; m += 256) rsum += reward[(size_t)b * M + m];
    const float mean = block_sum(rsum, red, tid, 256) / (float)M;
    float amax = ELG_NEG_INF;
    for (int m = tid; m < M; m += 256) amax = fmaxf(amax, reward[(size_t)b * M + m] - mean);
    amax = block_max(amax, red, tid, 256);
    float jr = 0.f;
    for (int m = tid; m < M; m += 256) {
        const float adv = reward[(size_t)b * M + m] - mean;
        float lp = 0.f;
        for (int t = 0; t < T; ++t) lp += logf(pb[(size_t)t * p_tstride + m]);
        jr = fmaf(-adv, lp, jr);
        coef_raw[(size_t)b * M + m] = -adv;
        coef_scaled[(size_t)b * M + m] = -adv / amax;
    }
    jr = block_sum(jr, red, tid, 256);
    if (tid == 0) {
        Jraw[b] = jr;
        Jscaled[b] = jr / amax;
        amax_out[b] = amax;
    }
}

// Cotangent rows for the decoder backward, from the rows the training forward saved (time-major r = t*M + m).
// One wavefront per row.  w = gprob * p * valid (valid: decoded step of an unfinished trajectory);
//   rowDL[r,n]  = w (c_sel [n == sel] - p_n c_n)           d loss / d (pre-clip score)      (PC holds p_n c_n)
//   rowDU[r,j]  = rowDL[r, slot_j] / ensemble_size         d loss / d (local policy output of slot j)
//   onehotP[r,n] = [n == previous node]   onehotF[r,n] = [n == first node]   (query-gather scatter matrices;
//   with `load` given, onehotP has one more column holding the vehicle load of the row, so that the same
//   GEMM onehotP^T dQ also yields d wl = sum_r load_r dQ_r)
__global__ __launch_bounds__(256) void rows_prep_kernel(
    const float* __restrict__ gprob, const float* __restrict__ pval, const int* __restrict__ tlen,
    const int* __restrict__ actions, const float* __restrict__ PC, const float* __restrict__ Csel,
    const int* __restrict__ Slot, const float* __restrict__ load, float* __restrict__ rowDL, float* __restrict__ rowDU,
    float* __restrict__ onehotP, float* __restrict__ onehotF, int* __restrict__ idxP, int* __restrict__ idxF, int B,
    int T, int M, int N1, int Tcap_act, long long Rcap, int t0, float inv_ens) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);      // over B * R
    const long long R = (long long)T * M;
    if (row >= (long long)B * R) return;
    const int b = (int)(row / R);
    const int r = (int)(row % R);
    const int t = r / M, m = r % M;
    const size_t bm = (size_t)b * M + m;
    const bool valid = t >= t0 && t < tlen[bm];
    const size_t gi = ((size_t)b * T + t) * M + m;
    const float w = valid ? gprob[gi] * pval[gi] : 0.f;
    const int* act = actions + bm * Tcap_act;
    const int sel = act[t];
    const int prev = t > 0 ? act[t - 1] : 0;
    const int first = act[0];
    const size_t src = (size_t)b * Rcap + r;
    const float wc = w * Csel[src];
    const float* pc = PC + src * N1;
    float* dl = rowDL + (size_t)row * N1;
    const int pw = load ? N1 + 1 : N1;                                  // pitch of onehotP
    for (int n = lane; n < N1; n += 64) {
        float v = -w * pc[n];
        if (n == sel) v += wc;
        dl[n] = v;
        if (onehotP) onehotP[(size_t)row * pw + n] = (n == prev) ? 1.f : 0.f;
        if (onehotF) onehotF[(size_t)row * N1 + n] = (n == first) ? 1.f : 0.f;
    }
    if (onehotP && load && lane == 0) onehotP[(size_t)row * pw + N1] = load[src];
    if (lane == 0) {
        if (idxP) idxP[row] = prev;
        if (idxF) idxF[row] = first;
    }
    if (rowDU && lane < 48) {
        const int s = Slot[src * 48 + lane];
        float v = 0.f;
        if (s >= 0) {
            v = -w * pc[s];
            if (s == sel) v += wc;
            v *= inv_ens;
        }
        rowDU[(size_t)row * 48 + lane] = v;
    }
}

// Adam with L2 weight decay folded into the gradient (torch.optim.Adam semantics, reference train.py:101):
//   g += wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / (1-b1^t) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// Gradients and moments are flat; the parameters either are flat too (param != NULL) or stay where the framework
// allocated them and are reached through a pointer table: element i belongs to tensor k with
// offsets[k] <= i < offsets[k+1] (binary search, <= 8 probes of a cached table).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, float* const* __restrict__ table,
                                                   const long long* __restrict__ offsets, int n_tensors,
                                                   const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* pp;
    if (param) pp = param + i;
    else {
        int lo = 0, hi = n_tensors;                                // invariant: offsets[lo] <= i < offsets[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (offsets[mid] <= i) lo = mid; else hi = mid;
        }
        pp = table[lo] + (i - offsets[lo]);
    }
    const float pi = *pp;
    const float gi = fmaf(wd, pi, g[i] * gscale);
    const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
    const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    *pp = pi - (lr / bc1) * (mi / denom);
}


// out[b, n, :] = sum over the rows r of instance b with idx[b, r] == n of X[b, r, :]   (+ row `wrow`: sum_r w[b, r] X[b, r, :])
// = onehot^T X without the (B, R, N) one-hot matrix (316 MB at the bench shape): the one-hot is the A operand of
// v_mfma_f32_16x16x4_f32, built in registers from the row's index (A[i = node 16 nt + lo][k = row r0 + hi] = [idx[r] == node]).
// Workgroup = (row split, instance); wave w owns channels 32 w .. 32 w + 31 (2 column tiles x NT node tiles of accumulators).
using f32x4t = __attribute__((ext_vector_type(4))) float;
template <int NT>
__global__ __launch_bounds__(256) void rows_segsum_kernel(const float* __restrict__ X, const int* __restrict__ idx,
                                                          const float* __restrict__ w, float* __restrict__ outp, int B, int R,
                                                          int NO, int wrow, long long w_stride, int splits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = lane & 15, hi = lane >> 4;
    const int b = blockIdx.y, split = blockIdx.x;
    const int nsteps = (R + 3) >> 2;
    const int per = (nsteps + splits - 1) / splits;
    const int s_lo = split * per, s_hi = min(nsteps, s_lo + per);
    f32x4t acc[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { acc[nt][0] = f32x4t{0.f, 0.f, 0.f, 0.f}; acc[nt][1] = f32x4t{0.f, 0.f, 0.f, 0.f}; }
    const float* Xb = X + (size_t)b * R * 128 + 32 * wave + lo;
    const int* ib = idx + (size_t)b * R;
    const float* wb = w ? w + (size_t)b * w_stride : nullptr;
#pragma unroll 4
    for (int st = s_lo; st < s_hi; ++st) {
        const int r = 4 * st + hi;                                   // k-slot hi <-> row r
        const int rc = min(r, R - 1);
        const float live = (r < R) ? 1.f : 0.f;
        const int id = ib[rc];
        const float x0 = Xb[(size_t)rc * 128] * live, x1 = Xb[(size_t)rc * 128 + 16] * live;
        const float wv = wb ? wb[rc] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int node = 16 * nt + lo;
            float a = (id == node) ? 1.f : 0.f;
            if (node == wrow) a = wv;
            acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x0, acc[nt][0], 0, 0, 0);
            acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x1, acc[nt][1], 0, 0, 0);
        }
    }
    float* ob = outp + ((size_t)split * B + b) * NO * 128 + 32 * wave + lo;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int node = 16 * nt + 4 * hi + v;                   // D rows
            if (node < NO) { ob[(size_t)node * 128] = acc[nt][0][v]; ob[(size_t)node * 128 + 16] = acc[nt][1][v]; }
        }
}


// utils.check_feasible (reference CVRP/utils.py:90-119, TSP/utils.py:72-78) for the M tours of one instance: one
// workgroup per tour.  flags[0] |= 1: some customer not visited exactly once (or an entry outside 0..N);
// flags[1] |= 1: the reference's sequential scan `used += d; used[used < 0] = 0; used <= 1 + 1e-4` failed (same fp32
// additions in the same order; a depot visit adds -1).  demand == NULL: TSP (every node 0..N-1 exactly once, no depot).
__global__ __launch_bounds__(256) void check_feasible_kernel(const long long* __restrict__ pi, long long m_stride,
                                                             const float* __restrict__ demand, int T, int N, int* flags) {
    extern __shared__ int sm[];
    const bool tsp = demand == nullptr;
    const int nn = tsp ? N : N + 1;                     // node ids
    int* cnt = sm;
    float* dseq = reinterpret_cast<float*>(sm + nn);
    const long long* row = pi + (long long)blockIdx.x * m_stride;
    const int tid = threadIdx.x;
    for (int i = tid; i < nn; i += 256) cnt[i] = 0;
    __syncthreads();
    bool bad = false;
    for (int t = tid; t < T; t += 256) {
        const long long v = row[t];
        if (v < 0 || v >= nn) { bad = true; if (!tsp) dseq[t] = 0.f; continue; }
        atomicAdd(cnt + (int)v, 1);
        if (!tsp) dseq[t] = v == 0 ? -1.0f : demand[v - 1];
    }
    __syncthreads();
    for (int i = tid + (tsp ? 0 : 1); i < nn; i += 256) bad |= cnt[i] != 1;
    if (bad) atomicOr(flags, 1);
    if (!tsp && tid == 0) {
        float used = 0.f;
        bool over = false;
        for (int t = 0; t < T; ++t) {
            used = __fadd_rn(used, dseq[t]);
            if (used < 0.f) used = 0.f;
            over |= !(used <= 1.0001f);
        }
        if (over) atomicOr(flags + 1, 1);
    }
}

// stats[0] = max tlen, stats[1] = 1 if some chosen probability of a decoded step is exactly 0 (CVRPModel.py:67-68)
__global__ __launch_bounds__(256) void rollout_stats_kernel(const int* __restrict__ tlen, const float* __restrict__ probs,
                                                            long long n_traj, int M, int Tcap, int* stats) {
    __shared__ int smax[4], szero[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    int tl = 0, z = 0;
    if (i < n_traj) {
        tl = tlen[i];
        const long long b = i / M, m = i % M;
        const float* p = probs + b * (long long)Tcap * M + m;
        for (int t = 0; t < tl; ++t) z |= p[(long long)t * M] == 0.f;
    }
    tl = (int)wave_max((float)tl);
    z = __any(z);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { smax[w] = tl; szero[w] = z; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(stats, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
        if (szero[0] | szero[1] | szero[2] | szero[3]) atomicOr(stats + 1, 1);
    }
}

}  // namespace elg

using namespace elg;

extern "C" int elg_check_feasible(const int64_t* pi, int64_t m_stride, const float* demand, int M, int T, int N, int32_t* flags,
                                  void* stream) {
    if (!pi || !flags || M <= 0 || T <= 0 || N <= 0) return fail(ELG_EINVAL, "check_feasible: bad arguments");
    const size_t lds = (size_t)(N + 1 + T) * 4;
    if (lds > 64 * 1024) return fail(ELG_ENOTIMPL, "check_feasible: tour too long for the on-chip counters");
    (void)hipGetLastError();
    hipLaunchKernelGGL(check_feasible_kernel, dim3(M), dim3(256), lds, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(pi), (long long)m_stride, demand, T, N, flags);
    return launch_status("check_feasible");
}

extern "C" int elg_rollout_stats(const int32_t* tlen, const float* probs, int B, int M, int Tcap, int32_t* stats, void* stream) {
    if (!tlen || !probs || !stats || B <= 0 || M <= 0 || Tcap <= 0) return fail(ELG_EINVAL, "rollout_stats: bad arguments");
    const long long n = (long long)B * M;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rollout_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tlen, probs, n,
                       M, Tcap, stats);
    return launch_status("rollout_stats");
}

extern "C" int elg_pomo_loss(const float* probs, const float* reward, int B, int T, int M, int64_t probs_bstride,
                             int64_t probs_tstride, float* J_raw, float* J_scaled, float* adv_max, float* coef_raw,
                             float* coef_scaled, void* stream) {
    if (B <= 0 || T <= 0 || M <= 0) return fail(ELG_EINVAL, "pomo_loss: bad sizes");
    if (!probs || !reward || !J_raw || !J_scaled || !adv_max || !coef_raw || !coef_scaled)
        return fail(ELG_EINVAL, "pomo_loss: null buffer");
    (void)hipGetLastError();
    hipLaunchKernelGGL(pomo_loss_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, probs, reward, T, M,
                       (long long)probs_bstride, (long long)probs_tstride, J_raw, J_scaled, adv_max, coef_raw, coef_scaled);
    return launch_status("pomo_loss");
}

extern "C" int elg_rows_prep(const float* gprob, const float* pval, const int32_t* tlen, const int32_t* actions,
                             const float* PC, const float* Csel, const int32_t* Slot, const float* load, float* rowDL,
                             float* rowDU, float* onehot_prev, float* onehot_first, int32_t* idx_prev, int32_t* idx_first,
                             int B, int T, int M, int N1, int Tcap_actions,
                             int64_t Rcap, int first_decode_step, float inv_ens, void* stream) {
    if (B <= 0 || T <= 0 || M <= 0 || N1 <= 1) return fail(ELG_EINVAL, "rows_prep: bad sizes");
    if (Rcap < (int64_t)T * M || Tcap_actions < T) return fail(ELG_EINVAL, "rows_prep: row capacity smaller than T*M");
    if (!gprob || !pval || !tlen || !actions || !PC || !Csel || !rowDL) return fail(ELG_EINVAL, "rows_prep: null buffer");
    if (rowDU && !Slot) return fail(ELG_EINVAL, "rows_prep: rowDU needs the slot rows");
    const long long rows = (long long)B * T * M;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rows_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, gprob, pval,
                       tlen, actions, PC, Csel, Slot, load, rowDL, rowDU, onehot_prev, onehot_first, idx_prev, idx_first, B, T, M, N1,
                       Tcap_actions,
                       (long long)Rcap, first_decode_step, inv_ens);
    return launch_status("rows_prep");
}

extern "C" int elg_adam_step(float* param, float* const* param_table, const int64_t* offsets, int n_tensors,
                             const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    if (n <= 0 || step <= 0) return fail(ELG_EINVAL, "adam_step: bad sizes");
    if (!grad || !exp_avg || !exp_avg_sq) return fail(ELG_EINVAL, "adam_step: null buffer");
    if (!param && (!param_table || !offsets || n_tensors <= 0)) return fail(ELG_EINVAL, "adam_step: no parameters");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    (void)hipGetLastError();
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param,
                       param_table, reinterpret_cast<const long long*>(offsets), n_tensors, grad, exp_avg, exp_avg_sq,
                       (long long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    return launch_status("adam_step");
}

extern "C" int elg_rows_segsum(const float* X, const int32_t* idx, const float* w, float* out_part, int B, int R, int NO,
                               int wrow, int64_t w_stride, int splits, void* stream) {
    if (B <= 0 || R <= 0 || NO <= 0 || NO > 128 || splits <= 0) return fail(ELG_EINVAL, "rows_segsum: bad sizes (NO <= 128)");
    if (!X || !idx || !out_part) return fail(ELG_EINVAL, "rows_segsum: null buffer");
    if (wrow >= 0 && (!w || wrow >= NO)) return fail(ELG_EINVAL, "rows_segsum: weighted row without weights");
    const int nt = (NO + 15) / 16;
    dim3 grid(splits, B);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
#define ELG_SS(NT) hipLaunchKernelGGL(rows_segsum_kernel<NT>, grid, dim3(256), 0, s, X, idx, w, out_part, B, R, NO, wrow, (long long)w_stride, splits)
    if (nt <= 2) ELG_SS(2);
    else if (nt <= 4) ELG_SS(4);
    else if (nt <= 7) ELG_SS(7);
    else ELG_SS(8);
#undef ELG_SS
    return launch_status("rows_segsum");
}
