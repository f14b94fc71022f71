// Folding the local policy's projections into the per-slot tables the rollout kernels read (layout ELG_LOC_* in
// include/elg_hip.h), forward and backward, one workgroup each.
//
// reference CVRP/models.py:8-49 (parameters, sinusoid table), :133-166 (forward); TSP/models.py:12-33,76-100:
//   e_j = We f_j + be + PE[j];  q = Wq c;  k_j = Wk e_j;  v_j = Wv e_j;
//   u_j = (Wc softmax_j(q . k_j / sqrt 8) v + bc) . e_j / sqrt 32
// Everything that does not depend on the features f_j is a table over the slot index j:
//   qk[h] = sum_d q[h,d] Wk[h*8+d, :] / sqrt 8        (4 x 32)
//   LA[h][f] = qk[h] . We[:, f]        LT[j][h] = qk[h] . base[j]          base[j] = be + PE[j]
//   LAV = Wv We                        LCV[j] = Wv base[j]
//   LWC = Wc   LBC = bc   LWE = We / sqrt 32   LPE[j] = base[j] / sqrt 32
// The backward is the exact adjoint of these (a few thousand multiply-adds: one workgroup, LDS scratch).
#include "elg_common.h"
#include "../../include/elg_hip.h"
#include <string>

namespace elg {
int fail(int code, const std::string& msg);
int launch_status(const char* what);

constexpr int LE = ELG_LE, LH = ELG_LH, LDKc = ELG_LDK, LROWS = ELG_LOC_ROWS;

// PE[j][i]: sin block then cos block, 16 timescales exp(-i ln(10000) / 15)   (models.py:28-49)
__device__ __forceinline__ float pos_enc(int j, int i, int positional) {
    if (!positional) return 0.f;
    const int ts = i & 15;
    const float inc = 9.210340371976184f / 15.0f;              // ln(10000) / (16 - 1), fp32 like the reference
    const float inv = expf((float)ts * -inc);
    const float st = (float)j * inv;
    return i < 16 ? sinf(st) : cosf(st);
}

struct LocalW {
    const float *We, *be, *c, *Wq, *Wk, *Wv, *Wc, *bc;
};
struct LocalG {
    float *We, *be, *c, *Wq, *Wk, *Wv, *Wc, *bc;
};

__global__ __launch_bounds__(256) void local_fold_fwd_kernel(const LocalW P, int F, int L, int positional, float* __restrict__ loc) {
    __shared__ float sq[LE], sqk[LH][LE], sbase[LROWS][LE + 1], sWe[LE][3];
    const int tid = threadIdx.x;
    for (int i = tid; i < ELG_LOC_SIZE; i += 256) loc[i] = 0.f;
    if (tid < LE) {
        float a = 0.f;
        for (int e = 0; e < LE; ++e) a = fmaf(P.Wq[tid * LE + e], P.c[e], a);
        sq[tid] = a;
        for (int f = 0; f < 3; ++f) sWe[tid][f] = f < F ? P.We[tid * F + f] : 0.f;
    }
    for (int i = tid; i < LROWS * LE; i += 256) {
        const int j = i / LE, e = i % LE;
        sbase[j][e] = j < L ? P.be[e] + pos_enc(j, e, positional) : 0.f;
    }
    __syncthreads();
    if (tid < LH * LE) {
        const int h = tid / LE, e = tid % LE;
        float a = 0.f;
        for (int d = 0; d < LDKc; ++d) a = fmaf(sq[h * LDKc + d], P.Wk[(h * LDKc + d) * LE + e], a);
        sqk[h][e] = a * 0.35355339059327373f;                   // 1 / sqrt(8)
    }
    __syncthreads();
    if (tid < LH * 3) {                                         // LA [4][3]
        const int h = tid / 3, f = tid % 3;
        float a = 0.f;
        for (int e = 0; e < LE; ++e) a = fmaf(sqk[h][e], sWe[e][f], a);
        loc[ELG_LOC_LA + tid] = a;
    }
    {                                                           // LT [64][4]
        const int j = tid >> 2, h = tid & 3;
        float a = 0.f;
        if (j < L)
            for (int e = 0; e < LE; ++e) a = fmaf(sqk[h][e], sbase[j][e], a);
        loc[ELG_LOC_LT + tid] = a;
    }
    if (tid < LE * 3) {                                         // LAV [32][3], LWE [32][3]
        const int i = tid / 3, f = tid % 3;
        float a = 0.f;
        for (int e = 0; e < LE; ++e) a = fmaf(P.Wv[i * LE + e], sWe[e][f], a);
        loc[ELG_LOC_LAV + tid] = a;
        loc[ELG_LOC_LWE + tid] = sWe[i][f] * 0.17677669529663687f;      // 1 / sqrt(32)
    }
    for (int idx = tid; idx < LROWS * LE; idx += 256) {         // LCV, LPE [64][32]
        const int j = idx / LE, i = idx % LE;
        float a = 0.f;
        if (j < L)
            for (int e = 0; e < LE; ++e) a = fmaf(P.Wv[i * LE + e], sbase[j][e], a);
        loc[ELG_LOC_LCV + idx] = a;
        loc[ELG_LOC_LPE + idx] = j < L ? sbase[j][i] * 0.17677669529663687f : 0.f;
    }
    for (int idx = tid; idx < LE * LE; idx += 256) loc[ELG_LOC_LWC + idx] = P.Wc[idx];
    if (tid < LE) loc[ELG_LOC_LBC + tid] = P.bc[tid];
}

__global__ __launch_bounds__(256) void local_fold_bwd_kernel(const LocalW P, int F, int L, int positional,
                                                             const float* __restrict__ g, const LocalG G) {
    __shared__ float sq[LE], sqk[LH][LE], sbase[LROWS][LE + 1], sWe[LE][3], sgb[LROWS][LE + 1], sgqk[LH][LE], sgq[LE];
    const int tid = threadIdx.x;
    const float r8 = 0.35355339059327373f, r32 = 0.17677669529663687f;
    if (tid < LE) {
        float a = 0.f;
        for (int e = 0; e < LE; ++e) a = fmaf(P.Wq[tid * LE + e], P.c[e], a);
        sq[tid] = a;
        for (int f = 0; f < 3; ++f) sWe[tid][f] = f < F ? P.We[tid * F + f] : 0.f;
    }
    for (int i = tid; i < LROWS * LE; i += 256) {
        const int j = i / LE, e = i % LE;
        sbase[j][e] = j < L ? P.be[e] + pos_enc(j, e, positional) : 0.f;
    }
    __syncthreads();
    if (tid < LH * LE) {
        const int h = tid / LE, e = tid % LE;
        float a = 0.f;
        for (int d = 0; d < LDKc; ++d) a = fmaf(sq[h * LDKc + d], P.Wk[(h * LDKc + d) * LE + e], a);
        sqk[h][e] = a * r8;
    }
    __syncthreads();
    // d base[j][e] = g_LPE / sqrt32 + sum_i g_LCV[j][i] Wv[i][e] + sum_h g_LT[j][h] qk[h][e]
    for (int idx = tid; idx < LROWS * LE; idx += 256) {
        const int j = idx / LE, e = idx % LE;
        float a = 0.f;
        if (j < L) {
            a = g[ELG_LOC_LPE + idx] * r32;
            for (int i = 0; i < LE; ++i) a = fmaf(g[ELG_LOC_LCV + j * LE + i], P.Wv[i * LE + e], a);
            for (int h = 0; h < LH; ++h) a = fmaf(g[ELG_LOC_LT + j * 4 + h], sqk[h][e], a);
        }
        sgb[j][e] = a;
    }
    // d qk[h][e] = sum_j g_LT[j][h] base[j][e] + sum_f g_LA[h][f] We[e][f]
    if (tid < LH * LE) {
        const int h = tid / LE, e = tid % LE;
        float a = 0.f;
        for (int j = 0; j < L; ++j) a = fmaf(g[ELG_LOC_LT + j * 4 + h], sbase[j][e], a);
        for (int f = 0; f < 3; ++f) a = fmaf(g[ELG_LOC_LA + h * 3 + f], sWe[e][f], a);
        sgqk[h][e] = a;
    }
    __syncthreads();
    if (tid < LE) {                                             // d be, d q, d bc
        float a = 0.f;
        for (int j = 0; j < L; ++j) a += sgb[j][tid];
        G.be[tid] = a;
        const int h = tid / LDKc;
        float b = 0.f;
        for (int e = 0; e < LE; ++e) b = fmaf(sgqk[h][e], P.Wk[tid * LE + e], b);
        sgq[tid] = b * r8;
        G.bc[tid] = g[ELG_LOC_LBC + tid];
    }
    __syncthreads();
    for (int idx = tid; idx < LE * LE; idx += 256) {
        const int i = idx / LE, e = idx % LE;
        G.Wc[idx] = g[ELG_LOC_LWC + idx];
        G.Wq[idx] = sgq[i] * P.c[e];
        G.Wk[idx] = sq[i] * sgqk[i / LDKc][e] * r8;
        // d Wv[i][e] = sum_j g_LCV[j][i] base[j][e] + sum_f g_LAV[i][f] We[e][f]
        float a = 0.f;
        for (int j = 0; j < L; ++j) a = fmaf(g[ELG_LOC_LCV + j * LE + i], sbase[j][e], a);
        for (int f = 0; f < 3; ++f) a = fmaf(g[ELG_LOC_LAV + i * 3 + f], sWe[e][f], a);
        G.Wv[idx] = a;
    }
    if (tid < LE) {                                             // d c[e] = sum_i gq[i] Wq[i][e]
        float a = 0.f;
        for (int i = 0; i < LE; ++i) a = fmaf(sgq[i], P.Wq[i * LE + tid], a);
        G.c[tid] = a;
    }
    if (tid < LE * F) {                                         // d We[e][f]
        const int e = tid / F, f = tid % F;
        float a = g[ELG_LOC_LWE + e * 3 + f] * r32;
        for (int i = 0; i < LE; ++i) a = fmaf(P.Wv[i * LE + e], g[ELG_LOC_LAV + i * 3 + f], a);
        for (int h = 0; h < LH; ++h) a = fmaf(sqk[h][e], g[ELG_LOC_LA + h * 3 + f], a);
        G.We[tid] = a;
    }
}

}  // namespace elg

using namespace elg;

static int check_local(const elg_local_weights* w, int nfeat, int n_slots) {
    if (!w || !w->init_emb_w || !w->init_emb_b || !w->cur_token_emb || !w->Wq || !w->Wk || !w->Wv || !w->combine_w || !w->combine_b)
        return fail(ELG_EINVAL, "local fold: null parameter");
    if (nfeat < 2 || nfeat > 3) return fail(ELG_EINVAL, "local fold: 2 (TSP) or 3 (CVRP) slot features");
    if (n_slots < 1 || n_slots > ELG_LOC_ROWS) return fail(ELG_EINVAL, "local fold: 1 .. 64 slots");
    return ELG_OK;
}

extern "C" int elg_local_fold_fwd(const elg_local_weights* w, int nfeat, int n_slots, int positional, float* loc, void* stream) {
    const int rc = check_local(w, nfeat, n_slots);
    if (rc != ELG_OK) return rc;
    if (!loc) return fail(ELG_EINVAL, "local fold: null output");
    LocalW P{w->init_emb_w, w->init_emb_b, w->cur_token_emb, w->Wq, w->Wk, w->Wv, w->combine_w, w->combine_b};
    (void)hipGetLastError();
    hipLaunchKernelGGL(local_fold_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, P, nfeat, n_slots, positional, loc);
    return launch_status("local_fold_fwd");
}

extern "C" int elg_local_fold_bwd(const elg_local_weights* w, int nfeat, int n_slots, int positional, const float* gloc,
                                  const elg_local_weights* grads, void* stream) {
    int rc = check_local(w, nfeat, n_slots);
    if (rc != ELG_OK) return rc;
    rc = check_local(grads, nfeat, n_slots);
    if (rc != ELG_OK) return rc;
    if (!gloc) return fail(ELG_EINVAL, "local fold: null cotangent");
    LocalW P{w->init_emb_w, w->init_emb_b, w->cur_token_emb, w->Wq, w->Wk, w->Wv, w->combine_w, w->combine_b};
    LocalG G{(float*)grads->init_emb_w, (float*)grads->init_emb_b, (float*)grads->cur_token_emb, (float*)grads->Wq,
             (float*)grads->Wk, (float*)grads->Wv, (float*)grads->combine_w, (float*)grads->combine_b};
    (void)hipGetLastError();
    hipLaunchKernelGGL(local_fold_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, P, nfeat, n_slots, positional, gloc, G);
    return launch_status("local_fold_bwd");
}
