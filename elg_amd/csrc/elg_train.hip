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
__global__ __launch_bounds__(1024) void pomo_loss_kernel(const float* __restrict__ probs, const float* __restrict__ reward,
                                                         int T, int M, long long p_bstride, long long p_tstride,
                                                         float* __restrict__ Jraw, float* __restrict__ Jscaled,
                                                         float* __restrict__ amax_out, float* __restrict__ coef_raw,
                                                         float* __restrict__ coef_scaled) {
    // 1024 threads = G step groups x MP trajectory columns (MP = power of two >= min(M, 1024)): group g sums log p over
    // the steps t = g, g + G, ...; the groups' partial sums are added in group order (fixed order: deterministic)
    __shared__ float red[16];
    __shared__ float part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pb = probs + (size_t)b * p_bstride;
    int MP = 1;
    while (MP < M && MP < 1024) MP <<= 1;
    const int G = 1024 / MP, g = tid / MP, mc = tid - g * MP;
    float rsum = 0.f;
    for (int m = tid; m < M; m += 1024) rsum += reward[(size_t)b * M + m];
    const float mean = block_sum(rsum, red, tid, 1024) / (float)M;
    float amax = ELG_NEG_INF;
    for (int m = tid; m < M; m += 1024) amax = fmaxf(amax, reward[(size_t)b * M + m] - mean);
    amax = block_max(amax, red, tid, 1024);
    float jr = 0.f;
    for (int m0 = 0; m0 < M; m0 += MP) {
        const int m = m0 + mc;
        float lp = 0.f;
        if (m < M)
            for (int t = g; t < T; t += G) lp += logf(pb[(size_t)t * p_tstride + m]);
        __syncthreads();
        part[tid] = lp;
        __syncthreads();
        if (g == 0 && m < M) {
            float tot = part[mc];
            for (int k = 1; k < G; ++k) tot += part[k * MP + mc];
            const float adv = reward[(size_t)b * M + m] - mean;
            jr = fmaf(-adv, tot, jr);
            coef_raw[(size_t)b * M + m] = -adv;
            coef_scaled[(size_t)b * M + m] = -adv / amax;
        }
    }
    jr = block_sum(jr, red, tid, 1024);
    if (tid == 0) {
        Jraw[b] = jr;
        Jscaled[b] = jr / amax;
        amax_out[b] = amax;
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

// stats[0] = max tlen, stats[1] = 1 if some chosen probability of a decoded step is exactly 0 (CVRPModel.py:67-68),
// zero_steps[t] = 1 for those steps.  grid (B, slices of the time axis): a workgroup scans its (steps x M) block of one
// instance with the trajectories along the lanes.
__global__ __launch_bounds__(256) void rollout_stats_kernel(const int* __restrict__ tlen, const float* __restrict__ probs,
                                                            int M, int Tcap, int tper, int* stats, int* __restrict__ zero_steps) {
    __shared__ int smax[4];
    const int b = blockIdx.x, t0 = blockIdx.y * tper, t1 = min(Tcap, t0 + tper);
    const int* tl = tlen + (size_t)b * M;
    const float* p = probs + (size_t)b * Tcap * M;
    if (blockIdx.y == 0) {
        int mx = 0;
        for (int m = threadIdx.x; m < M; m += 256) mx = max(mx, tl[m]);
        mx = (int)wave_max((float)mx);
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(stats, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
    }
    if (!probs) return;                                    // a greedy rollout without probabilities: the length only
    for (int m = threadIdx.x; m < M; m += 256) {
        const int len = min(tl[m], t1);
        for (int t = t0; t < len; ++t)
            if (p[(size_t)t * M + m] == 0.f) {
                atomicOr(stats + 1, 1);
                if (zero_steps) atomicOr(zero_steps + t, 1);
            }
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

extern "C" int elg_rollout_stats(const int32_t* tlen, const float* probs, int B, int M, int Tcap, int32_t* stats,
                                 int32_t* zero_steps, void* stream) {
    if (!tlen || !stats || B <= 0 || M <= 0 || Tcap <= 0) return fail(ELG_EINVAL, "rollout_stats: bad arguments");
    // ~1024 workgroups: the time axis in slices of >= 8 steps
    const int slices = max(1, min((Tcap + 7) / 8, (1024 + B - 1) / B));
    const int tper = (Tcap + slices - 1) / slices;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rollout_stats_kernel, dim3((unsigned)B, (unsigned)((Tcap + tper - 1) / tper)), dim3(256), 0, (hipStream_t)stream,
                       tlen, probs, M, Tcap, tper, stats, zero_steps);
    return launch_status("rollout_stats");
}

// The scaled POMO loss of a CVRP training step AND its gradient w.r.t. the chosen probabilities in one launch (elg_pomo_loss_grad):
// p' = p + 1e-6 [a chosen probability of step t was exactly 0]   (CVRPModel.py:67-68, as rollout_train applies it),
// J_b = inv_count sum_m (-adv / max adv) sum_t log p',   g[b,t,m] = inv_count (-adv / max adv) / p'.
// Same arithmetic and summation order as pomo_loss_kernel + the element-wise chain it replaces (the +1e-6 add, J's scale, the three
// element-wise kernels of the autograd backward: five launches of 6 - 12 us in front of the decoder backward).
__global__ __launch_bounds__(1024) void pomo_loss_grad_kernel(const float* __restrict__ probs, const float* __restrict__ reward,
                                                              const int* __restrict__ zsteps, int T, int M, long long p_bstride,
                                                              long long p_tstride, float inv_count, float* __restrict__ Jterm,
                                                              float* __restrict__ gprob, const int* __restrict__ T_dev,
                                                              float* __restrict__ Jtotal, int* __restrict__ ticket) {
    __shared__ float red[16];
    __shared__ float part[1024];
    __shared__ float coef[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* pb = probs + (size_t)b * p_bstride;
    // steps past the rollout's longest trajectory hold probability 1 and no zero flag: log p' = 0 and g = coefficient / 1, exactly
    // what the arithmetic below gives -- written without reading them (the buffer is padded to 2 N + 1 steps, a tour takes ~1.15 N)
    const int Te = T_dev ? min(T, max(T_dev[0], 0)) : T;
    int MP = 1;
    while (MP < M && MP < 1024) MP <<= 1;
    const int G = 1024 / MP, g = tid / MP, mc = tid - g * MP;
    float rsum = 0.f;
    for (int m = tid; m < M; m += 1024) rsum += reward[(size_t)b * M + m];
    const float mean = block_sum(rsum, red, tid, 1024) / (float)M;
    float amax = ELG_NEG_INF;
    for (int m = tid; m < M; m += 1024) amax = fmaxf(amax, reward[(size_t)b * M + m] - mean);
    amax = block_max(amax, red, tid, 1024);
    float jr = 0.f;
    for (int m0 = 0; m0 < M; m0 += MP) {
        const int m = m0 + mc;
        float lp = 0.f;
        if (m < M)
            for (int t = g; t < Te; t += G) lp += logf(pb[(size_t)t * p_tstride + m] + (zsteps && zsteps[t] ? 1e-6f : 0.f));
        __syncthreads();
        part[tid] = lp;
        __syncthreads();
        if (g == 0 && m < M) {
            float tot = part[mc];
            for (int k = 1; k < G; ++k) tot += part[k * MP + mc];
            const float adv = reward[(size_t)b * M + m] - mean;
            jr = fmaf(-adv, tot, jr);
            coef[mc] = -adv / amax;
        }
        __syncthreads();
        // the gradient rows of these MP trajectories
        if (m < M) {
            const float cf = coef[mc] * inv_count;
            for (int t = g; t < Te; t += G) {
                const float pv = pb[(size_t)t * p_tstride + m] + (zsteps && zsteps[t] ? 1e-6f : 0.f);
                gprob[((size_t)b * T + t) * M + m] = cf / pv;
            }
            for (int t = Te + g; t < T; t += G) gprob[((size_t)b * T + t) * M + m] = cf;
        }
    }
    jr = block_sum(jr, red, tid, 1024);
    if (tid == 0) __hip_atomic_store(Jterm + b, (jr / amax) * inv_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (Jtotal) {
        // the last workgroup to arrive adds the B terms in index order (the same bits whatever the arrival order) and re-arms the ticket
        __shared__ int last;
        if (tid == 0) {
            __threadfence();
            last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
        }
        __syncthreads();
        if (last && tid == 0) {
            __threadfence();
            float tot = 0.f;
            for (int i = 0; i < (int)gridDim.x; ++i) tot += __hip_atomic_load(Jterm + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the workgroups sit on eight L2s)
            Jtotal[0] = tot;
            *ticket = 0;
        }
    }
}

extern "C" int elg_pomo_loss(const float* probs, const float* reward, int B, int T, int M, int64_t probs_bstride,
                             int64_t probs_tstride, float* J_raw, float* J_scaled, float* adv_max, float* coef_raw,
                             float* coef_scaled, void* stream) {
    if (B <= 0 || T <= 0 || M <= 0) return fail(ELG_EINVAL, "pomo_loss: bad sizes");
    if (!probs || !reward || !J_raw || !J_scaled || !adv_max || !coef_raw || !coef_scaled)
        return fail(ELG_EINVAL, "pomo_loss: null buffer");
    (void)hipGetLastError();
    hipLaunchKernelGGL(pomo_loss_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, probs, reward, T, M,
                       (long long)probs_bstride, (long long)probs_tstride, J_raw, J_scaled, adv_max, coef_raw, coef_scaled);
    return launch_status("pomo_loss");
}

extern "C" int elg_pomo_loss_grad(const float* probs, const float* reward, const int32_t* zero_steps, int B, int T, int M,
                                  int64_t probs_bstride, int64_t probs_tstride, float inv_count, float* J_terms, float* gprob,
                                  const int32_t* T_dev, float* J_total, int32_t* ticket, void* stream) {
    if (B <= 0 || T <= 0 || M <= 0) return fail(ELG_EINVAL, "pomo_loss_grad: bad sizes");
    if (!probs || !reward || !J_terms || !gprob) return fail(ELG_EINVAL, "pomo_loss_grad: null buffer");
    if ((J_total != nullptr) != (ticket != nullptr)) return fail(ELG_EINVAL, "pomo_loss_grad: J_total needs a ticket word (and vice versa)");
    (void)hipGetLastError();
    hipLaunchKernelGGL(pomo_loss_grad_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, probs, reward, zero_steps, T, M,
                       (long long)probs_bstride, (long long)probs_tstride, inv_count, J_terms, gprob, T_dev, J_total, ticket);
    return launch_status("pomo_loss_grad");
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
