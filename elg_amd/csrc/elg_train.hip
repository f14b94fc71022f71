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
    float* __restrict__ onehotP, float* __restrict__ onehotF, int B, int T, int M, int N1, int Tcap_act, long long Rcap,
    int t0, float inv_ens) {
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

}  // namespace elg

using namespace elg;

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
                             float* rowDU, float* onehot_prev, float* onehot_first, int B, int T, int M, int N1, int Tcap_actions,
                             int64_t Rcap, int first_decode_step, float inv_ens, void* stream) {
    if (B <= 0 || T <= 0 || M <= 0 || N1 <= 1) return fail(ELG_EINVAL, "rows_prep: bad sizes");
    if (Rcap < (int64_t)T * M || Tcap_actions < T) return fail(ELG_EINVAL, "rows_prep: row capacity smaller than T*M");
    if (!gprob || !pval || !tlen || !actions || !PC || !Csel || !rowDL) return fail(ELG_EINVAL, "rows_prep: null buffer");
    if (rowDU && !Slot) return fail(ELG_EINVAL, "rows_prep: rowDU needs the slot rows");
    const long long rows = (long long)B * T * M;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rows_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, gprob, pval,
                       tlen, actions, PC, Csel, Slot, load, rowDL, rowDU, onehot_prev, onehot_first, B, T, M, N1, Tcap_actions,
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
