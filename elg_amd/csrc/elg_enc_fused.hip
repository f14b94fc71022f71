// The attention encoder for N1 <= 128 as per-instance FUSED layer kernels (reference CVRP/models.py:199-269 CVRP_Encoder /
// EncoderLayer, :455-503 multi_head_attention, :506-527 AddAndInstanceNormalization, :550-561 FeedForward, :300-308 set_kv;
// TSP/models.py:134-194,231-243), forward and backward.  Behind elg_encoder_fwd / elg_encoder_bwd (elg_enc.hip dispatches).
//
// Why: with one kernel per GEMM a layer was 5 launches forward and 5 backward, each a single generation of workgroups whose
// duration is its slowest wave's latency chain (load the row block, 3.5 us of MFMA, epilogue) -- 15-36 us per launch for
// 1-4 us of matrix work.  Here a WAVE owns one tile of 16 rows (nodes) of one instance for a whole sub-layer and
// every product is formed TRANSPOSED, D^T = W x^T: the weight rows are the MFMA's A operand (lane lo = output channel; staged
// through LDS for the whole workgroup, see below), the wave's activation tile is the B operand (lane lo = row).  The D registers of
// such a product, acc[i] = y[row lo][channel 16 ct + 4 hi + i], are exactly the B-operand fragment of chunk ct of the NEXT
// product -- so QKV -> attention, combine -> norm -> FFN1 -> ReLU -> FFN2 and their backward chains run from registers: no
// activation goes through LDS or memory between the GEMMs of a kernel.  What crosses waves is only what the math makes
// cross rows: K / V of the attention (LDS), and the per-(instance, channel) statistics of InstanceNorm1d (a DPP row
// reduction + one 4 KB LDS exchange).
//
// An instance is split over S workgroups so that 64 instances fill 256 CUs: by head pair for the attention block (S = 4), by
// 128-wide slice of the hidden layer for the feed-forward block (S = ff / 128); the slices' partial sums of FFN2 (and of the
// input gradients in the backward) are written to S buffers and summed in the prologue of the kernel that consumes them,
// where the add & instance norm (resp. its backward) runs as well -- every workgroup of the instance repeats that cheap
// element-wise prologue (the norm needs all rows and the next GEMM all channels), the workgroup c = 0 writes its result
// for the backward.  The combine GEMM in front of the first norm is repeated by the S feed-forward workgroups for the same
// reason (+1/3 of that kernel's MFMAs, -1 launch and one round trip per layer).  Partial 0 also carries the residual path (its
// producer has x1 / dS2 / dY in registers), so a consumer fetches S tiles, not S + 1.
// precision = 1 (template parameter BF): the same kernels with every MFMA operand rounded to bf16 (see mma_lds).
//
// Launches per training step at 6 layers: forward 13 (was 36), backward 16 + the grouped weight-gradient launch (was 39).
#include "elg_enc_internal.h"
#include "elg_bf16.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace elg {

// In-kernel phase clock: only in the diagnostic build (-DELG_STAMPS, tools/stamp_enc.py); the shipped library executes no stamp.
#ifdef ELG_STAMPS
__device__ unsigned long long* g_enc_stamps = nullptr;
#define STAMP(KID, K)                                                                                          \
    if (lane == 0 && g_enc_stamps)                                                                             \
        g_enc_stamps[(((size_t)(KID) * 2048 + blockIdx.x) * 8 + wave) * 16 + (K)] = __builtin_amdgcn_s_memtime();
#else
#define STAMP(KID, K)
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// blockIdx -> (instance b, part c of S): the S workgroups of an instance get ids that are equal modulo 8, i.e. (with the
// round-robin placement of consecutive ids over the 8 XCDs) they share an L2.  Placement is for speed only.
__device__ __forceinline__ void map_block(int bid, int B, int S, int& b, int& c) {
    const int full = B & ~7;
    if (bid < full * S) {
        const int grp = bid / (8 * S), r = bid - grp * 8 * S;
        c = r >> 3;
        b = grp * 8 + (r & 7);
    } else {
        const int idx = bid - full * S, rem = B - full;
        c = idx / rem;
        b = full + idx - c * rem;
    }
}

// ---- weights go through LDS.  Every wave of the workgroup needs the same weight rows (its activation rows are what
// differs), and fetched per wave from global memory they were the kernels' bound (first version: 7 waves x 64 KB per
// product through the vector-memory path, 18-20 us per 128 x 128 product for 3.4 us of MFMA).  A STAGE is a block of weight
// rows (<= 64 output channels x K) copied by all 512 threads: global -> registers (issued one stage ahead, under the
// current stage's MFMAs) -> LDS image [row][K + 8] (row pitch = 8 mod 64 floats: the fragment read ds_read_b128
// [row lo][16 kc + 4 hi ..] is bank-conflict free) -> one workgroup barrier.
constexpr int WP128 = 136;          // LDS row pitch (floats) of a K = 128 weight image
constexpr int WP96 = 104;           //                           K = 96

// ---- bf16 mode (elg_encoder_args.precision = 1; BASELINE configs[1] "bf16"): every GEMM of the encoder and of its backward on
// v_mfma_f32_16x16x32_bf16 -- both operands rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32), f32 accumulation;
// bias, residual, instance norm, softmax and everything saved for the backward stay f32 (oracle: oracle/elg_oracle.py
// encoder_forward(precision="bf16")).  The instruction contracts 32 k per issue, 8 per lane: k-slot (hi, j).  The D^T tiles that
// chain the GEMMs hold 4 consecutive channels per lane and tile, so a 32-wide k chunk = two tiles, and slot (hi, j) stands for
// channel 32 kc + 4 hi + j (j < 4, tile 2 kc) resp. 32 kc + 16 + 4 hi + j - 4 (j >= 4, tile 2 kc + 1) in BOTH operands: the
// weight image is written in that order (one ds_read_b128 per lane, tile and chunk), the activation operand is the two tiles'
// registers packed.  Image row = K bf16 + 16 (pitch = 8 mod 64 dwords: conflict-free fragment reads).
constexpr int WPB128 = 72;          // dwords per row of a K = 128 bf16 weight image
constexpr int WPB96 = 56;           //                      K = 96
// dword offset inside an image row of the 4 channels 4 p .. 4 p + 3 (p = 16-byte piece of the f32 row)
__device__ __forceinline__ int bf_piece_off(const int p) { return 16 * (p >> 3) + 4 * (p & 3) + 2 * ((p >> 2) & 1); }
__device__ __forceinline__ u32x4 pack8(const float4 a, const float4 b) {
    return u32x4{pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(b.x, b.y), pk_bf16(b.z, b.w)};
}

// out[j] (D^T tiles: out[j][i] = y[row lo][16 j + 4 hi + i]) = sum_k W(16 j + lo, k) x[row lo][k]: NCT column tiles from the
// LDS image sW, the wave's activation tile `in` held as KC chunks of 16 k (in[kc] = x[row lo][16 kc + 4 hi ..]).
template <bool BF, int NCT, int KC, int PITCH>
__device__ __forceinline__ void mma_lds(const float* sW, const float4 (&in)[KC], f32x4* out, const int lo, const int hi) {
#pragma unroll
    for (int j = 0; j < NCT; ++j) out[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (BF) {
        static_assert((KC & 1) == 0, "bf16: k in chunks of 32");
        constexpr int PB = PITCH == WP128 ? WPB128 : WPB96;
        const unsigned* p = reinterpret_cast<const unsigned*>(sW) + lo * PB + 4 * hi;
#pragma unroll
        for (int kc = 0; kc < KC / 2; ++kc) {
            const u32x4 bop = pack8(in[2 * kc], in[2 * kc + 1]);
            u32x4 a[NCT];
#pragma unroll
            for (int j = 0; j < NCT; ++j) a[j] = *reinterpret_cast<const u32x4*>(p + j * (16 * PB) + 16 * kc);
#pragma unroll
            for (int j = 0; j < NCT; ++j) out[j] = mfma_bf(a[j], bop, out[j]);
        }
    } else {
        const float* p = sW + lo * PITCH + 4 * hi;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            float4 a[NCT];
#pragma unroll
            for (int j = 0; j < NCT; ++j) a[j] = ld4(p + j * (16 * PITCH) + 16 * kc);
#pragma unroll
            for (int j = 0; j < NCT; ++j) out[j] = mfma4(a[j].x, in[kc].x, out[j]);
#pragma unroll
            for (int j = 0; j < NCT; ++j) out[j] = mfma4(a[j].y, in[kc].y, out[j]);
#pragma unroll
            for (int j = 0; j < NCT; ++j) out[j] = mfma4(a[j].z, in[kc].z, out[j]);
#pragma unroll
            for (int j = 0; j < NCT; ++j) out[j] = mfma4(a[j].w, in[kc].w, out[j]);
        }
    }
}

// one piece (4 channels of a row) of a weight stage into the image
template <bool BF, int PITCH>
__device__ __forceinline__ void image_put(float* dst, const int row, const int piece, const float4 v) {
    if constexpr (BF) {
        constexpr int PB = PITCH == WP128 ? WPB128 : WPB96;
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned*>(dst) + row * PB + bf_piece_off(piece)) =
            make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
    } else {
        st4(dst + row * PITCH + 4 * piece, v);
    }
}

// stage copy of NROWS x 128 floats, rows `ld` apart and 16-byte aligned: NF4 = NROWS / 16 float4 per thread
template <int NF4>
__device__ __forceinline__ void stage_fetch(float4 (&r)[NF4], const float* src, const int ld, const int tid) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        const int idx = tid + 512 * i;
        r[i] = ld4(src + (size_t)(idx >> 5) * ld + 4 * (idx & 31));
    }
}
template <bool BF, int NF4>
__device__ __forceinline__ void stage_commit(const float4 (&r)[NF4], float* dst, const int tid) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        const int idx = tid + 512 * i;
        image_put<BF, WP128>(dst, idx >> 5, idx & 31, r[i]);
    }
}
// the same for a weight given as W(m, k) = src[m sm + k sk] (the decoder tables and their backward: nn.Linear weights used
// in both orientations, and the (128,129) Wq_last): 64 rows x 128 k.  sk = 1: rows contiguous (16-byte loads when aligned);
// else sm = 1: lanes walk m (coalesced), the image is written transposed.
__device__ __forceinline__ void stage_fetch_any(float4 (&r)[4], const float* src, const int sm, const int sk, const int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i;
        if (sk == 1) {
            const float* p = src + (size_t)(idx >> 5) * sm + 4 * (idx & 31);
            if ((sm & 3) == 0) r[i] = ld4(p);
            else r[i] = make_float4(p[0], p[1], p[2], p[3]);
        } else {
            const float* p = src + (size_t)(4 * (idx >> 6)) * sk + (size_t)(idx & 63) * sm;
            r[i] = make_float4(p[0], p[sk], p[2 * sk], p[3 * sk]);
        }
    }
}
template <bool BF>
__device__ __forceinline__ void stage_commit_any(const float4 (&r)[4], float* dst, const int sk, const int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 512 * i;
        if (sk == 1) image_put<BF, WP128>(dst, idx >> 5, idx & 31, r[i]);
        else image_put<BF, WP128>(dst, idx & 63, idx >> 6, r[i]);
    }
}

// ---- per-(instance, channel) sums over the node axis.  v[t] = the lane's row, channels 16 t + 4 hi ..; rows that do not
// exist must hold 0.  put: DPP all-reduce over the 16 rows of the tile, lane lo = 0 of each quarter writes stat[wave][128];
// (barrier); get: every lane adds the tiles' partials of its 32 channels.
__device__ __forceinline__ void colsum_put(const float4 (&v)[8], float* stat, int wave, int lo, int hi) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float4 r;
        r.x = row16_sum(v[t].x); r.y = row16_sum(v[t].y); r.z = row16_sum(v[t].z); r.w = row16_sum(v[t].w);
        if (lo == 0) st4(stat + wave * ELG_E + 16 * t + 4 * hi, r);
    }
}
__device__ __forceinline__ void colsum_get(float4 (&r)[8], const float* stat, int nact, int hi) {
#pragma unroll
    for (int t = 0; t < 8; ++t) r[t] = ld4(stat + 16 * t + 4 * hi);
    for (int w = 1; w < nact; ++w) {
#pragma unroll
        for (int t = 0; t < 8; ++t) r[t] = add4(r[t], ld4(stat + w * ELG_E + 16 * t + 4 * hi));
    }
}

// ---- activation tiles come from memory in LOAD layout and are turned into the MFMA fragment layout through LDS.
// Measured (tools/_bench/ldbench.hip): a CU's vector-memory path delivers cache LINES, not bytes -- ~7 cycles per 128-byte line
// touched when the line is in L2, ~14 when it comes from the memory side (everything another kernel has just written), whatever
// the lanes use of it.  The fragment map (lane (lo, hi): row lo, 16 bytes at column 16 t + 4 hi) touches 16 half lines per
// instruction, i.e. every line of the tile twice: 5 - 7 tiles per prologue cost 35 - 42 K cycles in the first version, more than
// the kernels' MFMA time.  Load layout: instruction t, lane l: row 2 t + (l >> 5), 16 bytes at column 4 (l & 31) -- two whole
// rows = 8 whole lines per instruction.  Sums of tiles (partials + residual + bias) are formed in load layout (element-wise), then
// ONE pass through a wave-private LDS buffer (16 x TP floats, written as rows, read back as fragments) re-maps the result.
constexpr int TP = 136;             // row pitch of the transpose buffer (= 8 mod 64: conflict-free fragment reads)
constexpr int TBUF = 16 * TP;       // floats per wave

struct TileAddr {                   // per-lane element offsets of the 8 load-layout slots of the wave's tile (rows past N1 clamped)
    size_t off[8];
};
__device__ __forceinline__ TileAddr tile_addr(const int b, const int N1, const int wave, const int lane, const int ld) {
    TileAddr a;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int r = min(16 * wave + 2 * t + (lane >> 5), N1 - 1);
        a.off[t] = ((size_t)b * N1 + r) * ld + 4 * (lane & 31);
    }
    return a;
}
__device__ __forceinline__ void tile_ld(float4 (&v)[8], const float* src, const TileAddr& a) {
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = ld4(src + a.off[t]);
}
__device__ __forceinline__ void tile_add(float4 (&acc)[8], const float4 (&v)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = add4(acc[t], v[t]);
}
// acc += sum of np tiles pstride floats apart; up to four in flight together (a loop that waits per tile costs one exposed round
// trip each); slots past np re-read the last tile with weight 0
__device__ __forceinline__ void tile_add_partials(float4 (&acc)[8], const float* P, const long pstride, const int np, const TileAddr& a) {
    for (int q0 = 0; q0 < np; q0 += 4) {
        float4 v[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) tile_ld(v[j], P + (size_t)min(q0 + j, np - 1) * pstride, a);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float wq = q0 + j < np ? 1.f : 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc[t].x = fmaf(wq, v[j][t].x, acc[t].x); acc[t].y = fmaf(wq, v[j][t].y, acc[t].y);
                acc[t].z = fmaf(wq, v[j][t].z, acc[t].z); acc[t].w = fmaf(wq, v[j][t].w, acc[t].w);
            }
        }
    }
}
// load layout -> fragment layout (x[t] = row lo, columns 16 t + 4 hi ..) through the wave's own buffer sT
__device__ __forceinline__ void to_frag(const float4 (&v)[8], float4 (&x)[8], float* sT, const int lane, const int lo, const int hi) {
#pragma unroll
    for (int t = 0; t < 8; ++t) st4(sT + (2 * t + (lane >> 5)) * TP + 4 * (lane & 31), v[t]);
    wave_lds_fence();
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = ld4(sT + lo * TP + 16 * t + 4 * hi);
    wave_lds_fence();
}
// 1 / sqrt(v): v_rsq_f32 (1 ulp) -- the per-channel statistics are recomputed by every lane that holds the channel, and the
// correctly rounded sqrt + division sequences were ~700 VALU instructions per lane and norm
__device__ __forceinline__ float rsq(float v) { return __builtin_amdgcn_rsqf(v); }
// s / n for small integer n given inv = 1/n rounded: product + one residual correction (correctly rounded except in rare ties;
// the means of ~1e2 values must be as exact as f32 allows, the bias gradients upstream are sums of ds that cancel to 0)
__device__ __forceinline__ float div_n(float s, float fn, float inv) {
    const float m = s * inv;
    return fmaf(fmaf(-m, fn, s), inv, m);
}

// ---- prologue of the forward kernels: the wave's 16 x 128 tile of the layer input.
//   mode 0: the input embedding (models.py:206-217: depot Linear(2,128), customers Linear(3,128) on (x, y, demand); TSP Linear(2,128))
//   mode 1: the previous layer's second add & instance norm on x1 + (sum of the np FFN2 partials + bias)   (models.py:266-268,
//           :506-527: per (instance, channel) mean / biased variance over the node axis, two passes)
//   mode 2: a plain load (set_kv on given encodings)
struct EncPro {
    int mode, np;
    const float *xy, *demand, *Wd, *bd, *Wn, *bn;
    const float* P; long pstride;
    const float *res, *bias, *gamma, *beta;
    float *xhat, *rstd, *xout;          // written by the workgroup c = 0 (each may be NULL)
    float eps;
};

// all waves of the workgroup call this (it contains barriers); `stat` = 2 x 8 x 128 floats of LDS
__device__ __forceinline__ void enc_prologue(const EncPro& p, float4 (&x)[8], float* stat, float* sT, const int b, const int N1,
                                             const bool writer, const int wave, const int lane, const bool act, const int nact) {
    const int lo = lane & 15, hi = lane >> 4;
    const int row = 16 * wave + lo;
    const bool rvalid = row < N1;
    const size_t grow = (size_t)b * N1 + min(row, N1 - 1);
    if (p.mode != 1) {
        if (!act) return;
        if (p.mode == 0) {
            const float2 pt = *reinterpret_cast<const float2*>(p.xy + grow * 2);
            const float xv = pt.x, yv = pt.y;
            const bool depot = p.Wd && row == 0;
            const float dv = p.demand ? p.demand[grow] : 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int c0 = 16 * t + 4 * hi;
                float o[4];
                if (p.demand) {            // CVRP: customers Linear(3,128), depot Linear(2,128)
                    const float4 w0 = ld4(p.Wn + c0 * 3), w1 = ld4(p.Wn + c0 * 3 + 4), w2 = ld4(p.Wn + c0 * 3 + 8);
                    const float4 d0 = ld4(p.Wd + c0 * 2), d1 = ld4(p.Wd + c0 * 2 + 4);
                    const float4 bn = ld4(p.bn + c0), bd = ld4(p.bd + c0);
                    const float wn[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
                    const float wd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                    const float bnv[4] = {bn.x, bn.y, bn.z, bn.w}, bdv[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float vn = fmaf(wn[3 * j + 2], dv, fmaf(wn[3 * j + 1], yv, fmaf(wn[3 * j], xv, 0.f))) + bnv[j];
                        const float vd = fmaf(wd[2 * j + 1], yv, fmaf(wd[2 * j], xv, 0.f)) + bdv[j];
                        o[j] = depot ? vd : vn;
                    }
                } else {
                    const float4 w0 = ld4(p.Wn + c0 * 2), w1 = ld4(p.Wn + c0 * 2 + 4);
                    const float4 bn = ld4(p.bn + c0);
                    const float wn[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                    const float bnv[4] = {bn.x, bn.y, bn.z, bn.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(wn[2 * j + 1], yv, fmaf(wn[2 * j], xv, 0.f)) + bnv[j];
                }
                x[t] = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
            float4 v[8];
            tile_ld(v, p.xout, tile_addr(b, N1, wave, lane, ELG_E));
            to_frag(v, x, sT, lane, lo, hi);
            return;                                        // (mode 2: xout is the INPUT, nothing to write)
        }
        if (writer && rvalid && p.xout) {
#pragma unroll
            for (int t = 0; t < 8; ++t) st4(p.xout + grow * ELG_E + 16 * t + 4 * hi, x[t]);
        }
        return;
    }
    float4 s[8];
    if (act) {
        {
            const TileAddr ta = tile_addr(b, N1, wave, lane, ELG_E);
            // (the residual x1 and the FFN2 bias are already in partial 0: its producer, the slice c = 0 of enc_f2, has x1 in
            //  registers -- one tile less to fetch here, in every workgroup of the instance)
            float4 v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = zero4();
            tile_add_partials(v, p.P, p.pstride, p.np, ta);
            to_frag(v, s, sT, lane, lo, hi);
        }
        if (!rvalid) {
#pragma unroll
            for (int t = 0; t < 8; ++t) s[t] = zero4();
        }
        colsum_put(s, stat, wave, lo, hi);
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)N1;
    float4 mean[8];
    if (act) {
        colsum_get(mean, stat, nact, hi);
        float4 sq[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            mean[t] = make_float4(mean[t].x * inv_n, mean[t].y * inv_n, mean[t].z * inv_n, mean[t].w * inv_n);
            const float dx = s[t].x - mean[t].x, dy = s[t].y - mean[t].y, dz = s[t].z - mean[t].z, dw = s[t].w - mean[t].w;
            sq[t] = rvalid ? make_float4(dx * dx, dy * dy, dz * dz, dw * dw) : zero4();
        }
        colsum_put(sq, stat + 8 * ELG_E, wave, lo, hi);
    }
    __syncthreads();
    if (!act) return;
    float4 var[8];
    colsum_get(var, stat + 8 * ELG_E, nact, hi);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float4 rs = make_float4(rsq(var[t].x * inv_n + p.eps), rsq(var[t].y * inv_n + p.eps), rsq(var[t].z * inv_n + p.eps),
                                      rsq(var[t].w * inv_n + p.eps));
        const float4 xh = make_float4((s[t].x - mean[t].x) * rs.x, (s[t].y - mean[t].y) * rs.y, (s[t].z - mean[t].z) * rs.z,
                                      (s[t].w - mean[t].w) * rs.w);
        const float4 ga = ld4(p.gamma + 16 * t + 4 * hi), be = ld4(p.beta + 16 * t + 4 * hi);
        x[t] = make_float4(fmaf(xh.x, ga.x, be.x), fmaf(xh.y, ga.y, be.y), fmaf(xh.z, ga.z, be.z), fmaf(xh.w, ga.w, be.w));
        if (writer) {
            if (rvalid) {
                if (p.xout) st4(p.xout + grow * ELG_E + 16 * t + 4 * hi, x[t]);
                if (p.xhat) st4(p.xhat + grow * ELG_E + 16 * t + 4 * hi, xh);
            }
            if (wave == 0 && lo == 0 && p.rstd) st4(p.rstd + (size_t)b * ELG_E + 16 * t + 4 * hi, rs);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// F1: [previous add & norm | embedding] -> Q | K | V of one head pair -> softmax(Q K^T / 4) V.   grid B x 4, 8 waves.
struct EncF1 {
    EncPro pro;
    const float *Wq, *Wk, *Wv;
    float *QKV, *O, *LSE;               // QKV (R,384) and LSE (B,8,N1) only when the backward will run (else NULL)
    int B, N1;
};

template <int NT, bool BF>
__global__ __launch_bounds__(512) void enc_f1_kernel(const EncF1 g) {
    constexpr int ROWS = NT * 16, KP = 20, PT = ROWS + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stat = lds;                                   // 2 x 8 x 128
    float* sW = stat + 2 * 8 * ELG_E;                    // 96 x WP128: Wq | Wk | Wv rows of the head pair
    float* sK = sW + 96 * WP128;                         // [2][ROWS][KP]
    float* sVT = sK + 2 * ROWS * KP;                     // [2][16][PT]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sT = sK + wave * TBUF;                        // the prologue's transpose buffers share the K / V region (a barrier apart)
    const int lo = lane & 15, hi = lane >> 4;
    int b, c;
    map_block(blockIdx.x, g.B, 4, b, c);
    const int N1 = g.N1, nact = (N1 + 15) >> 4;
    const bool act = wave < nact;
    const int row = 16 * wave + lo;
    const bool rvalid = row < N1;
    const size_t grow = (size_t)b * N1 + min(row, N1 - 1);
    // the head pair's 3 x 32 weight rows: requested before the prologue, landed in LDS behind it
    float4 wr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int idx = tid + 512 * i, r = idx >> 5, which = r >> 5;               // r = 32 which + row of the pair
        const float* W = which == 0 ? g.Wq : (which == 1 ? g.Wk : g.Wv);
        wr[i] = ld4(W + (size_t)(32 * c + (r & 31)) * ELG_E + 4 * (idx & 31));
    }
    float4 x[8];
    STAMP(0, 0)
    enc_prologue(g.pro, x, stat, sT, b, N1, c == 0, wave, lane, act, nact);
    STAMP(0, 1)
    stage_commit<BF, 6>(wr, sW, tid);
    __syncthreads();
    STAMP(0, 2)
    f32x4 qkv[6];                                        // q0 q1 k0 k1 v0 v1
    if (act) {
        mma_lds<BF, 6, 8, WP128>(sW, x, qkv, lo, hi);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (g.QKV && rvalid) {
                float* o = g.QKV + grow * (3 * ELG_E) + 32 * c + 16 * u + 4 * hi;
                st4(o, f4(qkv[u]));
                st4(o + ELG_E, f4(qkv[2 + u]));
                st4(o + 2 * ELG_E, f4(qkv[4 + u]));
            }
            // keys past N1 in the tile are finite duplicates of the last row (clamped loads), masked in the softmax
            st4(sK + (u * ROWS + row) * KP + 4 * hi, f4(qkv[2 + u]));
#pragma unroll
            for (int i = 0; i < 4; ++i) sVT[(u * 16 + 4 * hi + i) * PT + row] = qkv[4 + u][i];
        }
    } else if (wave < NT) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            st4(sK + (u * ROWS + row) * KP + 4 * hi, zero4());
#pragma unroll
            for (int i = 0; i < 4; ++i) sVT[(u * 16 + 4 * hi + i) * PT + row] = 0.f;
        }
    }
    STAMP(0, 3)
    __syncthreads();
    STAMP(0, 4)
    if (!act) return;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        f32x4 S[NT];
        float mx = ELG_NEG_INF;
        const u32x4 qb = {pk_bf16(qkv[u][0], qkv[u][1]), pk_bf16(qkv[u][2], qkv[u][3]), 0u, 0u};     // (BF) k-slots (hi, 4..7) empty
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const float4 ka = ld4(sK + (u * ROWS + 16 * kt + lo) * KP + 4 * hi);
            f32x4 sc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
                sc = mfma_bf(u32x4{pk_bf16(ka.x, ka.y), pk_bf16(ka.z, ka.w), 0u, 0u}, qb, sc);
            } else {
                sc = mfma4(ka.x, qkv[u][0], sc);
                sc = mfma4(ka.y, qkv[u][1], sc);
                sc = mfma4(ka.z, qkv[u][2], sc);
                sc = mfma4(ka.w, qkv[u][3], sc);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = (16 * kt + 4 * hi + i < N1) ? sc[i] * 0.25f : ELG_NEG_INF;      // key 16 kt + 4 hi + i, query row lo
                mx = fmaxf(mx, sc[i]);
            }
            S[kt] = sc;
        }
        mx = quarters_max(mx);
        float l = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
            // two key tiles per instruction: k-slot (hi, j) = key 16 (2 p) + 4 hi + j, (hi, 4 + j) = key 16 (2 p + 1) + 4 hi + j;
            // the numerators go in rounded to bf16, the normaliser is the sum of the unrounded ones
#pragma unroll
            for (int pr = 0; pr < (NT + 1) / 2; ++pr) {
                const int t0 = 2 * pr, t1 = 2 * pr + 1;
                const float4 v0 = ld4(sVT + (u * 16 + lo) * PT + 16 * t0 + 4 * hi);
                const float4 v1 = t1 < NT ? ld4(sVT + (u * 16 + lo) * PT + 16 * t1 + 4 * hi) : zero4();
                float e0[4], e1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e0[i] = __expf(S[t0][i] - mx);
                    e1[i] = t1 < NT ? __expf(S[t1 < NT ? t1 : t0][i] - mx) : 0.f;
                    l += e0[i] + e1[i];
                }
                o = mfma_bf(pack8(v0, v1), u32x4{pk_bf16(e0[0], e0[1]), pk_bf16(e0[2], e0[3]), pk_bf16(e1[0], e1[1]), pk_bf16(e1[2], e1[3])}, o);
            }
        } else {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const float4 vt = ld4(sVT + (u * 16 + lo) * PT + 16 * kt + 4 * hi);
                const float vv[4] = {vt.x, vt.y, vt.z, vt.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __expf(S[kt][i] - mx);
                    l += p;
                    o = mfma4(vv[i], p, o);
                }
            }
        }
        l = quarters_sum(l);
        const float inv = 1.0f / l;
        if (rvalid) {
            st4(g.O + grow * ELG_E + 32 * c + 16 * u + 4 * hi, make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv));
            if (hi == 0 && g.LSE) g.LSE[((size_t)b * 8 + 2 * c + u) * N1 + row] = mx + __logf(l);
        }
    }
    STAMP(0, 5)
}

// ------------------------------------------------------------------------------------------------------------------
// F2: x1 = InstanceNorm(x + combine(att)) ; partial[c] = relu(x1 W1_c^T + b1_c) W2_c^T   (hidden slice c).  grid B x (ff / 128).
struct EncF2 {
    const float *O, *Xin, *Wc, *bc, *g1, *b1, *W1, *bf1, *W2, *bf2;
    float *X1, *XH1, *RS1, *H, *P;      // X1 always (the next prologue's residual); XH1 / RS1 / H when saving (else NULL)
    long pstride;
    int B, N1, FF;
    float eps;
};

template <bool BF>
__global__ __launch_bounds__(512) void enc_f2_kernel(const EncF2 g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stat = lds;                                   // 2 x 8 x 128
    float* sA = stat + 2 * 8 * ELG_E;                    // two weight stages of 64 rows
    float* sB = sA + 64 * WP128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sT = sB + 64 * WP128 + wave * TBUF;
    const int lo = lane & 15, hi = lane >> 4;
    int b, c;
    map_block(blockIdx.x, g.B, g.FF >> 7, b, c);
    const int N1 = g.N1, nact = (N1 + 15) >> 4;
    const bool act = wave < nact;
    const int row = 16 * wave + lo;
    const bool rvalid = row < N1;
    const size_t grow = (size_t)b * N1 + min(row, N1 - 1);
    const bool writer = c == 0;
    const float* W1c = g.W1 + (size_t)(128 * c) * ELG_E;         // rows of the hidden slice
    const float* W2c = g.W2 + 128 * c;                           // columns of the hidden slice
    float4 wr[4];
    STAMP(1, 0)
    stage_fetch<4>(wr, g.Wc, ELG_E, tid);
    float4 o[8], s[8];
    if (act) {
        const TileAddr ta = tile_addr(b, N1, wave, lane, ELG_E);
        float4 v[8], w[8];
        tile_ld(v, g.O, ta);
        tile_ld(w, g.Xin, ta);                           // the residual
        to_frag(v, o, sT, lane, lo, hi);
        to_frag(w, s, sT, lane, lo, hi);
    }
    stage_commit<BF, 4>(wr, sA, tid);
    __syncthreads();
    STAMP(1, 1)
    f32x4 a[8];
    stage_fetch<4>(wr, g.Wc + 64 * ELG_E, ELG_E, tid);
    if (act) mma_lds<BF, 4, 8, WP128>(sA, o, a, lo, hi);
    STAMP(1, 2)
    stage_commit<BF, 4>(wr, sB, tid);
    __syncthreads();
    STAMP(1, 3)
    stage_fetch<4>(wr, W1c, ELG_E, tid);
    if (act) {
        mma_lds<BF, 4, 8, WP128>(sB, o, a + 4, lo, hi);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            s[t] = add4(add4(f4(a[t]), ld4(g.bc + 16 * t + 4 * hi)), s[t]);
            if (!rvalid) s[t] = zero4();
        }
        colsum_put(s, stat, wave, lo, hi);
    }
    STAMP(1, 4)
    stage_commit<BF, 4>(wr, sA, tid);
    __syncthreads();
    STAMP(1, 5)
    const float inv_n = 1.0f / (float)N1;
    float4 mean[8];
    if (act) {
        colsum_get(mean, stat, nact, hi);
        float4 sq[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            mean[t] = make_float4(mean[t].x * inv_n, mean[t].y * inv_n, mean[t].z * inv_n, mean[t].w * inv_n);
            const float dx = s[t].x - mean[t].x, dy = s[t].y - mean[t].y, dz = s[t].z - mean[t].z, dw = s[t].w - mean[t].w;
            sq[t] = rvalid ? make_float4(dx * dx, dy * dy, dz * dz, dw * dw) : zero4();
        }
        colsum_put(sq, stat + 8 * ELG_E, wave, lo, hi);
    }
    __syncthreads();
    STAMP(1, 6)
    float4 x1[8];
    if (act) {
        float4 var[8];
        colsum_get(var, stat + 8 * ELG_E, nact, hi);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 rs = make_float4(rsq(var[t].x * inv_n + g.eps), rsq(var[t].y * inv_n + g.eps), rsq(var[t].z * inv_n + g.eps),
                                          rsq(var[t].w * inv_n + g.eps));
            const float4 xh = make_float4((s[t].x - mean[t].x) * rs.x, (s[t].y - mean[t].y) * rs.y, (s[t].z - mean[t].z) * rs.z,
                                          (s[t].w - mean[t].w) * rs.w);
            const float4 ga = ld4(g.g1 + 16 * t + 4 * hi), be = ld4(g.b1 + 16 * t + 4 * hi);
            x1[t] = make_float4(fmaf(xh.x, ga.x, be.x), fmaf(xh.y, ga.y, be.y), fmaf(xh.z, ga.z, be.z), fmaf(xh.w, ga.w, be.w));
            if (writer) {
                if (rvalid) {
                    st4(g.X1 + grow * ELG_E + 16 * t + 4 * hi, x1[t]);
                    if (g.XH1) st4(g.XH1 + grow * ELG_E + 16 * t + 4 * hi, xh);
                }
                if (wave == 0 && lo == 0 && g.RS1) st4(g.RS1 + (size_t)b * ELG_E + 16 * t + 4 * hi, rs);
            }
        }
    }
    // ---- h = relu(x1 W1_c^T + b1_c)
    STAMP(1, 7)
    stage_fetch<4>(wr, W1c + 64 * ELG_E, ELG_E, tid);
    if (act) mma_lds<BF, 4, 8, WP128>(sA, x1, a, lo, hi);
    STAMP(1, 8)
    stage_commit<BF, 4>(wr, sB, tid);
    __syncthreads();
    STAMP(1, 9)
    stage_fetch<4>(wr, W2c, g.FF, tid);
    float4 h[8];
    if (act) {
        mma_lds<BF, 4, 8, WP128>(sB, x1, a + 4, lo, hi);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 bb = ld4(g.bf1 + 128 * c + 16 * t + 4 * hi);
            h[t] = make_float4(fmaxf(a[t][0] + bb.x, 0.f), fmaxf(a[t][1] + bb.y, 0.f), fmaxf(a[t][2] + bb.z, 0.f), fmaxf(a[t][3] + bb.w, 0.f));
            if (g.H && rvalid) st4(g.H + grow * g.FF + 128 * c + 16 * t + 4 * hi, h[t]);
        }
    }
    STAMP(1, 10)
    stage_commit<BF, 4>(wr, sA, tid);
    __syncthreads();
    STAMP(1, 11)
    // ---- partial[c] = h W2_c^T
    stage_fetch<4>(wr, W2c + (size_t)64 * g.FF, g.FF, tid);
    if (act) mma_lds<BF, 4, 8, WP128>(sA, h, a, lo, hi);
    stage_commit<BF, 4>(wr, sB, tid);
    __syncthreads();
    STAMP(1, 12)
    if (act) {
        mma_lds<BF, 4, 8, WP128>(sB, h, a + 4, lo, hi);
        STAMP(1, 13)
        if (rvalid) {
            float* op = g.P + c * g.pstride + grow * ELG_E + 4 * hi;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float4 v = f4(a[t]);
                if (writer) v = add4(add4(v, ld4(g.bf2 + 16 * t + 4 * hi)), x1[t]);       // partial 0 carries the bias and the residual
                st4(op + 16 * t, v);
            }
        }
    }
    STAMP(1, 14)
}

// ------------------------------------------------------------------------------------------------------------------
// F3: the last add & norm -> encoded nodes ; table c = enc W_c (K, V, Q1, PK (+ pb), TSP: Q2)   (models.py:300-308 set_kv and the
// folds of engine.fold_decoder_tables).  grid B x max(ntab, 1).
struct EncF3 {
    EncPro pro;
    const float* W[5]; int sm[5], sk[5]; float alpha[5];
    float* out[5];
    int pb_tab;                          // the table whose workgroup also writes pb (-1: none)
    const float* bc; float* pb; float pb_scale;
    const float* wl_src; float* wl;      // CVRP: wl = Wq_last[:, 128]
    int ntab, B, N1;
};

template <typename T>
__device__ __forceinline__ T pick5(const T (&p)[5], int i) {
    T r = p[0];
    r = i == 1 ? p[1] : r;
    r = i == 2 ? p[2] : r;
    r = i == 3 ? p[3] : r;
    r = i == 4 ? p[4] : r;
    return r;
}

template <bool BF>
__global__ __launch_bounds__(512) void enc_f3_kernel(const EncF3 g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stat = lds;
    float* sA = stat + 2 * 8 * ELG_E;
    float* sB = sA + 64 * WP128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sT = sB + 64 * WP128 + wave * TBUF;
    const int lo = lane & 15, hi = lane >> 4;
    int b, c;
    map_block(blockIdx.x, g.B, max(g.ntab, 1), b, c);
    const int N1 = g.N1, nact = (N1 + 15) >> 4;
    const bool act = wave < nact;
    const int row = 16 * wave + lo;
    const bool rvalid = row < N1;
    const size_t grow = (size_t)b * N1 + min(row, N1 - 1);
    if (blockIdx.x == 0 && g.wl && tid < ELG_E) g.wl[tid] = g.wl_src[tid * (ELG_E + 1) + ELG_E];
    const float* W = pick5(g.W, c);
    const int sm = pick5(g.sm, c), sk = pick5(g.sk, c);
    float4 wr[4];
    if (g.ntab) stage_fetch_any(wr, W, sm, sk, tid);
    float4 x[8];
    enc_prologue(g.pro, x, stat, sT, b, N1, c == 0, wave, lane, act, nact);
    if (g.ntab == 0) return;
    stage_commit_any<BF>(wr, sA, sk, tid);
    __syncthreads();
    stage_fetch_any(wr, W + (size_t)64 * sm, sm, sk, tid);
    f32x4 a[8];
    if (act) mma_lds<BF, 4, 8, WP128>(sA, x, a, lo, hi);
    stage_commit_any<BF>(wr, sB, sk, tid);
    __syncthreads();
    if (!act) return;
    mma_lds<BF, 4, 8, WP128>(sB, x, a + 4, lo, hi);
    const float alpha = pick5(g.alpha, c);
    float* out = pick5(g.out, c);
    if (rvalid) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            st4(out + grow * ELG_E + 16 * t + 4 * hi, make_float4(a[t][0] * alpha, a[t][1] * alpha, a[t][2] * alpha, a[t][3] * alpha));
    }
    if (c == g.pb_tab) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = dot4(x[t], ld4(g.bc + 16 * t + 4 * hi), acc);
        acc = quarters_sum(acc);
        if (rvalid && hi == 0) g.pb[grow] = acc * g.pb_scale;
    }
}

// ==================================================================================================================
// backward

// W^T copies of the layer weights (once per step): the backward products dY W need W's columns as the contraction-contiguous
// operand rows.  dst[c][r] = src[r][c]; 32 x 32 tiles through LDS.
struct WtJob { const float* src; float* dst; int rows, cols, tile0; };
constexpr int WT_MAX_JOBS = 48;
struct WtBatch { WtJob job[WT_MAX_JOBS]; int njobs; };

__global__ __launch_bounds__(256) void enc_wt_kernel(const WtBatch bt) {
    __shared__ float tile[32][33];
    int j = 0;
    for (int t = 1; t < bt.njobs; ++t) j = (int)blockIdx.x >= bt.job[t].tile0 ? t : j;
    const float* src = nullptr; float* dst = nullptr; int rows = 0, cols = 0, tile0 = 0;
#pragma unroll
    for (int t = 0; t < WT_MAX_JOBS; ++t)
        if (t == j) { src = bt.job[t].src; dst = bt.job[t].dst; rows = bt.job[t].rows; cols = bt.job[t].cols; tile0 = bt.job[t].tile0; }
    const int tl = blockIdx.x - tile0, tc = cols >> 5;
    const int r0 = (tl / tc) * 32, c0 = (tl % tc) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = src[(size_t)(r0 + ty + 8 * i) * cols + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[(size_t)(c0 + ty + 8 * i) * rows + r0 + tx] = tile[tx][ty + 8 * i];
}

// ---- prologue of the backward kernels: d = base + sum of np partials is the cotangent of an add & instance norm output
// y = xhat gamma + beta:  ds = gamma rstd (d - mean(d) - xhat mean(d xhat)),  dbeta += sum d,  dgamma += sum d xhat
// (true divisions for the two means: the bias gradients upstream are sums of ds that cancel to zero).
struct EncBPro {
    const float* base; const float* P; long pstride; int np;
    const float *xhat, *rstd, *gamma;
    float *dgamma, *dbeta;
    float* dout;                         // ds, written by the workgroup c = 0 (the weight-gradient launch reads it)
};

__device__ __forceinline__ void enc_bprologue(const EncBPro& p, float4 (&ds)[8], float* stat, float* sT, const int b, const int N1,
                                              const bool writer, const int wave, const int lane, const bool act, const int nact) {
    const int lo = lane & 15, hi = lane >> 4;
    const int row = 16 * wave + lo;
    const bool rvalid = row < N1;
    const size_t grow = (size_t)b * N1 + min(row, N1 - 1);
    float4 d[8], xh[8];
    if (act) {
        {
            const TileAddr ta = tile_addr(b, N1, wave, lane, ELG_E);
            float4 v[8], w[8];
            tile_ld(w, p.xhat, ta);
            if (p.base) tile_ld(v, p.base, ta);
            else {
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = zero4();
            }
            tile_add_partials(v, p.P, p.pstride, p.np, ta);
            to_frag(w, xh, sT, lane, lo, hi);
            to_frag(v, d, sT, lane, lo, hi);
        }
        float4 dx[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (!rvalid) { d[t] = zero4(); xh[t] = zero4(); }
            dx[t] = make_float4(d[t].x * xh[t].x, d[t].y * xh[t].y, d[t].z * xh[t].z, d[t].w * xh[t].w);
        }
        colsum_put(d, stat, wave, lo, hi);
        colsum_put(dx, stat + 8 * ELG_E, wave, lo, hi);
    }
    __syncthreads();
    if (!act) return;
    float4 s1[8], s2[8];
    colsum_get(s1, stat, nact, hi);
    colsum_get(s2, stat + 8 * ELG_E, nact, hi);
    const float fn = (float)N1, inv_fn = 1.0f / fn;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (writer && wave == 0 && lo == 0) {
            float* gb = p.dbeta + 16 * t + 4 * hi;
            float* gg = p.dgamma + 16 * t + 4 * hi;
            atomicAdd(gb, s1[t].x); atomicAdd(gb + 1, s1[t].y); atomicAdd(gb + 2, s1[t].z); atomicAdd(gb + 3, s1[t].w);
            atomicAdd(gg, s2[t].x); atomicAdd(gg + 1, s2[t].y); atomicAdd(gg + 2, s2[t].z); atomicAdd(gg + 3, s2[t].w);
        }
        const float4 ga = ld4(p.gamma + 16 * t + 4 * hi), rs = ld4(p.rstd + (size_t)b * ELG_E + 16 * t + 4 * hi);
        const float4 k = make_float4(ga.x * rs.x, ga.y * rs.y, ga.z * rs.z, ga.w * rs.w);
        const float4 m1 = make_float4(div_n(s1[t].x, fn, inv_fn), div_n(s1[t].y, fn, inv_fn), div_n(s1[t].z, fn, inv_fn), div_n(s1[t].w, fn, inv_fn));
        const float4 m2 = make_float4(div_n(s2[t].x, fn, inv_fn), div_n(s2[t].y, fn, inv_fn), div_n(s2[t].z, fn, inv_fn), div_n(s2[t].w, fn, inv_fn));
        ds[t] = make_float4(k.x * (d[t].x - m1.x - xh[t].x * m2.x), k.y * (d[t].y - m1.y - xh[t].y * m2.y),
                            k.z * (d[t].z - m1.z - xh[t].z * m2.z), k.w * (d[t].w - m1.w - xh[t].w * m2.w));
        if (!rvalid) ds[t] = zero4();
        if (writer && rvalid) st4(p.dout + grow * ELG_E + 16 * t + 4 * hi, ds[t]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// B0: the tables' backward: partial[c] = alpha_c g_c W_c (+ gpb bc^T alpha on the PK partial).  grid B x ntab.
struct EncB0 {
    const float* G[5]; const float* W[5]; int sm[5], sk[5]; float alpha[5];
    int pb_tab; const float* gpb; const float* bc;
    const float* g_enc;                  // d loss / d enc itself (or NULL): added to partial 0
    float* PX; long pstride;
    int ntab, B, N1;
};

template <bool BF>
__global__ __launch_bounds__(512) void enc_b0_kernel(const EncB0 g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sA = lds;
    float* sB = sA + 64 * WP128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sT = sB + 64 * WP128 + wave * TBUF;
    const int lo = lane & 15, hi = lane >> 4;
    int b, c;
    map_block(blockIdx.x, g.B, g.ntab, b, c);
    const int N1 = g.N1, nact = (N1 + 15) >> 4;
    const bool act = wave < nact;
    const int row = 16 * wave + lo;
    const bool rvalid = row < N1;
    const size_t grow = (size_t)b * N1 + min(row, N1 - 1);
    const float* G = pick5(g.G, c);
    const float* W = pick5(g.W, c);
    const int sm = pick5(g.sm, c), sk = pick5(g.sk, c);
    const float alpha = pick5(g.alpha, c);
    float4 wr[4];
    stage_fetch_any(wr, W, sm, sk, tid);
    float4 x[8];
    if (act) {
        float4 v[8];
        tile_ld(v, G, tile_addr(b, N1, wave, lane, ELG_E));
        to_frag(v, x, sT, lane, lo, hi);
    }
    stage_commit_any<BF>(wr, sA, sk, tid);
    __syncthreads();
    stage_fetch_any(wr, W + (size_t)64 * sm, sm, sk, tid);
    f32x4 a[8];
    if (act) mma_lds<BF, 4, 8, WP128>(sA, x, a, lo, hi);
    stage_commit_any<BF>(wr, sB, sk, tid);
    __syncthreads();
    if (!act) return;
    mma_lds<BF, 4, 8, WP128>(sB, x, a + 4, lo, hi);
    if (!rvalid) return;
    const float gp = (c == g.pb_tab && g.gpb) ? g.gpb[grow] * alpha : 0.f;
    float* o = g.PX + c * g.pstride + grow * ELG_E + 4 * hi;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float4 v = make_float4(a[t][0] * alpha, a[t][1] * alpha, a[t][2] * alpha, a[t][3] * alpha);
        if (c == g.pb_tab && g.gpb) {
            const float4 bb = ld4(g.bc + 16 * t + 4 * hi);
            v.x = fmaf(gp, bb.x, v.x); v.y = fmaf(gp, bb.y, v.y); v.z = fmaf(gp, bb.z, v.z); v.w = fmaf(gp, bb.w, v.w);
        }
        if (c == 0 && g.g_enc) v = add4(v, ld4(g.g_enc + grow * ELG_E + 16 * t + 4 * hi));
        st4(o + 16 * t, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// B1: second add & norm backward -> dS2 ; dH_c = (dS2 W2_c) [h_c > 0] ; partial[c] = dH_c W1_c.   grid B x (ff / 128).
struct EncB1 {
    EncBPro pro;
    const float *W2T, *W1T, *H;
    float *gH, *P1;
    long pstride;
    int B, N1, FF;
};

template <bool BF>
__global__ __launch_bounds__(512) void enc_b1_kernel(const EncB1 g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stat = lds;
    float* sA = stat + 2 * 8 * ELG_E;
    float* sB = sA + 64 * WP128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sT = sB + 64 * WP128 + wave * TBUF;
    const int lo = lane & 15, hi = lane >> 4;
    int b, c;
    map_block(blockIdx.x, g.B, g.FF >> 7, b, c);
    const int N1 = g.N1, nact = (N1 + 15) >> 4;
    const bool act = wave < nact;
    const int row = 16 * wave + lo;
    const bool rvalid = row < N1;
    const size_t grow = (size_t)b * N1 + min(row, N1 - 1);
    const float* W2Tc = g.W2T + (size_t)(128 * c) * ELG_E;       // W2T (ff,128): rows of the hidden slice
    const float* W1Tc = g.W1T + 128 * c;                         // W1T (128,ff): columns of the hidden slice
    float4 wr[4];
    stage_fetch<4>(wr, W2Tc, ELG_E, tid);
    float4 ds[8];
    STAMP(3, 0)
    enc_bprologue(g.pro, ds, stat, sT, b, N1, c == 0, wave, lane, act, nact);
    STAMP(3, 1)
    float4 hm[8];
    if (act) {
        float4 v[8];
        tile_ld(v, g.H + 128 * c, tile_addr(b, N1, wave, lane, g.FF));
        to_frag(v, hm, sT, lane, lo, hi);
    }
    stage_commit<BF, 4>(wr, sA, tid);
    __syncthreads();
    STAMP(3, 2)
    f32x4 a[8];
    stage_fetch<4>(wr, W2Tc + 64 * ELG_E, ELG_E, tid);
    if (act) mma_lds<BF, 4, 8, WP128>(sA, ds, a, lo, hi);
    STAMP(3, 3)
    stage_commit<BF, 4>(wr, sB, tid);
    __syncthreads();
    STAMP(3, 4)
    stage_fetch<4>(wr, W1Tc, g.FF, tid);
    float4 dh[8];
    if (act) {
        mma_lds<BF, 4, 8, WP128>(sB, ds, a + 4, lo, hi);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            dh[t] = make_float4(hm[t].x > 0.f ? a[t][0] : 0.f, hm[t].y > 0.f ? a[t][1] : 0.f, hm[t].z > 0.f ? a[t][2] : 0.f,
                                hm[t].w > 0.f ? a[t][3] : 0.f);
            if (rvalid) st4(g.gH + grow * g.FF + 128 * c + 16 * t + 4 * hi, dh[t]);
        }
    }
    STAMP(3, 5)
    stage_commit<BF, 4>(wr, sA, tid);
    __syncthreads();
    STAMP(3, 6)
    stage_fetch<4>(wr, W1Tc + (size_t)64 * g.FF, g.FF, tid);
    if (act) mma_lds<BF, 4, 8, WP128>(sA, dh, a, lo, hi);
    stage_commit<BF, 4>(wr, sB, tid);
    __syncthreads();
    STAMP(3, 7)
    if (act) {
        mma_lds<BF, 4, 8, WP128>(sB, dh, a + 4, lo, hi);
        STAMP(3, 8)
        if (rvalid) {
            float* o = g.P1 + c * g.pstride + grow * ELG_E + 4 * hi;
#pragma unroll
            for (int t = 0; t < 8; ++t) st4(o + 16 * t, c == 0 ? add4(f4(a[t]), ds[t]) : f4(a[t]));     // partial 0 carries the residual path dS2
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// B2: first add & norm backward -> dY ; d att (head pair c) = dY Wc[:, pair] ; self-attention backward of the two heads ;
// partial[c] = dQ Wq_pair + dK Wk_pair + dV Wv_pair.   grid B x 4, 8 waves, dynamic LDS.
// The score tile is formed in both orientations (rows on lanes -> dQ of the wave's row tile; keys on lanes -> dK, dV of the
// wave's key tile), so every gradient accumulates in registers of the wave that owns the rows: no atomics, no transposes, and
// the three results are the B-operand chunks of the input-gradient product.
struct EncB2 {
    EncBPro pro;
    const float *WcT, *WqT, *WkT, *WvT;
    const float *QKV, *O, *LSE;
    float *dQKV, *P2;
    long pstride;
    int B, N1;
};

template <int NT>
constexpr int b2_attn_floats() {
    constexpr int ROWS = NT * 16, PT = ROWS + 4;
    constexpr int attn = 4 * 2 * ROWS * 20 + 3 * 2 * 16 * PT + 2 * 2 * ROWS;
    constexpr int other = 8 * TBUF > 2 * 64 * WP96 ? 8 * TBUF : 2 * 64 * WP96;
    return attn > other ? attn : other;
}

template <int NT, bool BF>
__global__ __launch_bounds__(512) void enc_b2_kernel(const EncB2 g) {
    constexpr int ROWS = NT * 16, KP = 20, PT = ROWS + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stat = lds;                                   // 2 x 8 x 128
    float* sWc = stat + 2 * 8 * ELG_E;                   // 32 x WP128: the head pair's rows of Wc^T
    float* sQ = sWc + 32 * WP128;                        // [2][ROWS][KP] each
    float* sK = sQ + 2 * ROWS * KP;
    float* sV = sK + 2 * ROWS * KP;
    float* sD = sV + 2 * ROWS * KP;
    float* sQT = sD + 2 * ROWS * KP;                     // [2][16][PT] each
    float* sKT = sQT + 2 * 16 * PT;
    float* sDT = sKT + 2 * 16 * PT;
    float* sL = sDT + 2 * 16 * PT;                       // [2][ROWS]
    float* sDel = sL + 2 * ROWS;
    float* sA = sQ;                                      // after the attention: two stages of the input-gradient weights (64 x WP96)
    float* sB = sA + 64 * WP96;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* sT = sQ + wave * TBUF;                        // before it: the prologue's transpose buffers
    const int lo = lane & 15, hi = lane >> 4;
    int b, c;
    map_block(blockIdx.x, g.B, 4, b, c);
    const int N1 = g.N1, nact = (N1 + 15) >> 4;
    const bool act = wave < nact;
    const int row = 16 * wave + lo;
    const bool rvalid = row < N1;
    const size_t grow = (size_t)b * N1 + min(row, N1 - 1);
    float4 wc[2];
    stage_fetch<2>(wc, g.WcT + (size_t)(32 * c) * ELG_E, ELG_E, tid);
    float4 dy[8];
    STAMP(4, 0)
    enc_bprologue(g.pro, dy, stat, sT, b, N1, c == 0, wave, lane, act, nact);
    STAMP(4, 1)
    stage_commit<BF, 2>(wc, sWc, tid);
    __syncthreads();
    STAMP(4, 2)
    float4 q4[2], k4[2], v4[2], d4[2];
    float lr[2], del[2];
    if (act) {
        f32x4 dO[2];
        mma_lds<BF, 2, 8, WP128>(sWc, dy, dO, lo, hi);
        const float mk = rvalid ? 1.f : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float* qp = g.QKV + grow * (3 * ELG_E) + 32 * c + 16 * u + 4 * hi;
            float4 q = ld4(qp);
            const float4 k = ld4(qp + ELG_E), v = ld4(qp + 2 * ELG_E);
            const float4 o = ld4(g.O + grow * ELG_E + 32 * c + 16 * u + 4 * hi);
            float4 d = f4(dO[u]);
            q.x *= mk; q.y *= mk; q.z *= mk; q.w *= mk;
            d.x *= mk; d.y *= mk; d.z *= mk; d.w *= mk;
            const float dl = quarters_sum(dot4(d, o, 0.f));
            const float ls = rvalid ? g.LSE[((size_t)b * 8 + 2 * c + u) * N1 + row] : __builtin_huge_valf();     // exp(s - inf) = 0
            q4[u] = q; k4[u] = k; v4[u] = v; d4[u] = d; lr[u] = ls; del[u] = dl;
            st4(sQ + (u * ROWS + row) * KP + 4 * hi, q);
            st4(sK + (u * ROWS + row) * KP + 4 * hi, k);
            st4(sV + (u * ROWS + row) * KP + 4 * hi, v);
            st4(sD + (u * ROWS + row) * KP + 4 * hi, d);
            const float qv[4] = {q.x, q.y, q.z, q.w}, kv[4] = {k.x, k.y, k.z, k.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sQT[(u * 16 + 4 * hi + i) * PT + row] = qv[i];
                sKT[(u * 16 + 4 * hi + i) * PT + row] = kv[i];
                sDT[(u * 16 + 4 * hi + i) * PT + row] = dv[i];
            }
            if (hi == 0) { sL[u * ROWS + row] = ls; sDel[u * ROWS + row] = dl; }
        }
    } else if (wave < NT) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            st4(sQ + (u * ROWS + row) * KP + 4 * hi, zero4());
            st4(sK + (u * ROWS + row) * KP + 4 * hi, zero4());
            st4(sV + (u * ROWS + row) * KP + 4 * hi, zero4());
            st4(sD + (u * ROWS + row) * KP + 4 * hi, zero4());
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sQT[(u * 16 + 4 * hi + i) * PT + row] = 0.f;
                sKT[(u * 16 + 4 * hi + i) * PT + row] = 0.f;
                sDT[(u * 16 + 4 * hi + i) * PT + row] = 0.f;
            }
            if (hi == 0) { sL[u * ROWS + row] = __builtin_huge_valf(); sDel[u * ROWS + row] = 0.f; }
        }
    }
    STAMP(4, 3)
    __syncthreads();
    STAMP(4, 4)
    // the input-gradient weights W(m = channel, k = (q | k | v, head of the pair, 16)) = W{q,k,v}^T[m][32 c + ..]: 128 x 96 in two
    // stages of 64 rows, 3 float4 per thread; the first is requested now, under the attention
    float4 wr[3];
    auto fetch3 = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 512 * i, r = idx / 24, pc = idx - r * 24, which = pc >> 3;
            const float* W = which == 0 ? g.WqT : (which == 1 ? g.WkT : g.WvT);
            wr[i] = ld4(W + (size_t)(m0 + r) * ELG_E + 32 * c + 4 * (pc & 7));
        }
    };
    auto commit3 = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = tid + 512 * i, r = idx / 24, pc = idx - r * 24;
            image_put<BF, WP96>(dst, r, pc, wr[i]);
        }
    };
    fetch3(0);
    float4 gin[6];                                       // dQ (2 heads) | dK | dV of the wave's rows
    if (act) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
                // bf16 operands: the 16-channel contractions fill k-slots (hi, 0..3) only; the contractions over keys / rows take two
                // tiles per instruction (slots (hi, j) = tile 2 p, (hi, 4 + j) = tile 2 p + 1).  p is recomputed from the scores of
                // bf16(q) . bf16(k), as the forward formed them.
                auto half = [](const float4 v) { return u32x4{pk_bf16(v.x, v.y), pk_bf16(v.z, v.w), 0u, 0u}; };
                const u32x4 qb = half(q4[u]), db = half(d4[u]), kb = half(k4[u]), vb = half(v4[u]);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                // ---- rows on lanes: dQ of the wave's row tile
#pragma unroll
                for (int pr = 0; pr < (NT + 1) / 2; ++pr) {
                    float dsv[2][4];
                    float4 kT[2];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int kt = 2 * pr + h2;
                        if (kt < NT) {
                            const f32x4 S = mfma_bf(half(ld4(sK + (u * ROWS + 16 * kt + lo) * KP + 4 * hi)), qb, z);
                            const f32x4 dP = mfma_bf(half(ld4(sV + (u * ROWS + 16 * kt + lo) * KP + 4 * hi)), db, z);
                            kT[h2] = ld4(sKT + (u * 16 + lo) * PT + 16 * kt + 4 * hi);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int key = 16 * kt + 4 * hi + i;
                                const float p = key < N1 ? __expf(S[i] * 0.25f - lr[u]) : 0.f;
                                dsv[h2][i] = p * (dP[i] - del[u]) * 0.25f;
                            }
                        } else {
                            kT[h2] = zero4();
#pragma unroll
                            for (int i = 0; i < 4; ++i) dsv[h2][i] = 0.f;
                        }
                    }
                    dq = mfma_bf(pack8(kT[0], kT[1]), u32x4{pk_bf16(dsv[0][0], dsv[0][1]), pk_bf16(dsv[0][2], dsv[0][3]),
                                                            pk_bf16(dsv[1][0], dsv[1][1]), pk_bf16(dsv[1][2], dsv[1][3])}, dq);
                }
                // ---- keys on lanes: dK, dV of the wave's key tile
#pragma unroll
                for (int pr = 0; pr < (NT + 1) / 2; ++pr) {
                    float pv[2][4], dsv[2][4];
                    float4 dT[2], qT[2];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int rt = 2 * pr + h2;
                        if (rt < NT) {
                            const f32x4 S = mfma_bf(half(ld4(sQ + (u * ROWS + 16 * rt + lo) * KP + 4 * hi)), kb, z);
                            const f32x4 dP = mfma_bf(half(ld4(sD + (u * ROWS + 16 * rt + lo) * KP + 4 * hi)), vb, z);
                            const float4 l4 = ld4(sL + u * ROWS + 16 * rt + 4 * hi);
                            const float4 e4 = ld4(sDel + u * ROWS + 16 * rt + 4 * hi);
                            dT[h2] = ld4(sDT + (u * 16 + lo) * PT + 16 * rt + 4 * hi);
                            qT[h2] = ld4(sQT + (u * 16 + lo) * PT + 16 * rt + 4 * hi);
                            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                pv[h2][i] = rvalid ? __expf(S[i] * 0.25f - lv[i]) : 0.f;
                                dsv[h2][i] = pv[h2][i] * (dP[i] - ev[i]) * 0.25f;
                            }
                        } else {
                            dT[h2] = zero4(); qT[h2] = zero4();
#pragma unroll
                            for (int i = 0; i < 4; ++i) { pv[h2][i] = 0.f; dsv[h2][i] = 0.f; }
                        }
                    }
                    dv = mfma_bf(pack8(dT[0], dT[1]), u32x4{pk_bf16(pv[0][0], pv[0][1]), pk_bf16(pv[0][2], pv[0][3]),
                                                            pk_bf16(pv[1][0], pv[1][1]), pk_bf16(pv[1][2], pv[1][3])}, dv);
                    dk = mfma_bf(pack8(qT[0], qT[1]), u32x4{pk_bf16(dsv[0][0], dsv[0][1]), pk_bf16(dsv[0][2], dsv[0][3]),
                                                            pk_bf16(dsv[1][0], dsv[1][1]), pk_bf16(dsv[1][2], dsv[1][3])}, dk);
                }
            } else {
            // ---- rows on lanes: dQ of the wave's row tile
            const float qB[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w}, dB[4] = {d4[u].x, d4[u].y, d4[u].z, d4[u].w};
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const float4 kA = ld4(sK + (u * ROWS + 16 * kt + lo) * KP + 4 * hi);
                const float4 vA = ld4(sV + (u * ROWS + 16 * kt + lo) * KP + 4 * hi);
                f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
                S = mfma4(kA.x, qB[0], S);
                dP = mfma4(vA.x, dB[0], dP);
                S = mfma4(kA.y, qB[1], S);
                dP = mfma4(vA.y, dB[1], dP);
                S = mfma4(kA.z, qB[2], S);
                dP = mfma4(vA.z, dB[2], dP);
                S = mfma4(kA.w, qB[3], S);
                dP = mfma4(vA.w, dB[3], dP);
                const float4 kT = ld4(sKT + (u * 16 + lo) * PT + 16 * kt + 4 * hi);
                const float kTv[4] = {kT.x, kT.y, kT.z, kT.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = 16 * kt + 4 * hi + i;
                    const float p = key < N1 ? __expf(S[i] * 0.25f - lr[u]) : 0.f;
                    const float dsv = p * (dP[i] - del[u]) * 0.25f;
                    dq = mfma4(kTv[i], dsv, dq);
                }
            }
            // ---- keys on lanes: dK, dV of the wave's key tile
            const float kB[4] = {k4[u].x, k4[u].y, k4[u].z, k4[u].w}, vB[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) {
                const float4 qA = ld4(sQ + (u * ROWS + 16 * rt + lo) * KP + 4 * hi);
                const float4 dA = ld4(sD + (u * ROWS + 16 * rt + lo) * KP + 4 * hi);
                f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
                S = mfma4(qA.x, kB[0], S);
                dP = mfma4(dA.x, vB[0], dP);
                S = mfma4(qA.y, kB[1], S);
                dP = mfma4(dA.y, vB[1], dP);
                S = mfma4(qA.z, kB[2], S);
                dP = mfma4(dA.z, vB[2], dP);
                S = mfma4(qA.w, kB[3], S);
                dP = mfma4(dA.w, vB[3], dP);
                const float4 l4 = ld4(sL + u * ROWS + 16 * rt + 4 * hi);
                const float4 e4 = ld4(sDel + u * ROWS + 16 * rt + 4 * hi);
                const float4 dT = ld4(sDT + (u * 16 + lo) * PT + 16 * rt + 4 * hi);
                const float4 qT = ld4(sQT + (u * 16 + lo) * PT + 16 * rt + 4 * hi);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
                const float dTv[4] = {dT.x, dT.y, dT.z, dT.w}, qTv[4] = {qT.x, qT.y, qT.z, qT.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = rvalid ? __expf(S[i] * 0.25f - lv[i]) : 0.f;
                    const float dsv = p * (dP[i] - ev[i]) * 0.25f;
                    dv = mfma4(dTv[i], p, dv);
                    dk = mfma4(qTv[i], dsv, dk);
                }
            }
            }
            gin[u] = f4(dq); gin[2 + u] = f4(dk); gin[4 + u] = f4(dv);
            if (rvalid) {
                float* o = g.dQKV + grow * (3 * ELG_E) + 32 * c + 16 * u + 4 * hi;
                st4(o, gin[u]);
                st4(o + ELG_E, gin[2 + u]);
                st4(o + 2 * ELG_E, gin[4 + u]);
            }
        }
    }
    STAMP(4, 5)
    __syncthreads();                                     // every wave is done with the attention operands: the region becomes sA | sB
    STAMP(4, 6)
    commit3(sA);
    __syncthreads();
    STAMP(4, 7)
    fetch3(64);
    f32x4 a[8];
    if (act) mma_lds<BF, 4, 6, WP96>(sA, gin, a, lo, hi);
    commit3(sB);
    __syncthreads();
    STAMP(4, 8)
    if (act) {
        mma_lds<BF, 4, 6, WP96>(sB, gin, a + 4, lo, hi);
        STAMP(4, 9)
        if (rvalid) {
            float* o = g.P2 + c * g.pstride + grow * ELG_E + 4 * hi;
#pragma unroll
            for (int t = 0; t < 8; ++t) st4(o + 16 * t, c == 0 ? add4(f4(a[t]), dy[t]) : f4(a[t]));     // partial 0 carries the residual path dY
        }
    }
}

// d embedding weights from d x0 = base + sum of np partials: grid over row chunks, 512 threads = 128 channels x 4 row phases
__global__ __launch_bounds__(512) void enc_embed_bwd2_kernel(const float* __restrict__ xy, const float* __restrict__ demand,
                                                             const float* __restrict__ base, const float* __restrict__ P, long pstride,
                                                             int np, float* gWd, float* gbd, float* gWn, float* gbn, int N1, long rows,
                                                             int rows_per_block) {
    __shared__ float part[3][7][ELG_E];
    const int c = threadIdx.x & (ELG_E - 1), q = threadIdx.x >> 7;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float v[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // node w0 w1 w2 b | depot w0 w1 b
    for (long r = r0 + q; r < r1; r += 4) {
        float gx = base ? base[r * ELG_E + c] : 0.f;
        for (int k = 0; k < np; ++k) gx += P[k * pstride + r * ELG_E + c];
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

// d bc[e] += sum_r enc[r][e] gpb[r] * scale ; d Wq_last[:, 128] += gwl     (elg_enc.hip)
int launch_fold_small_bwd(const float* enc, const float* gpb, const float* gwl, float* gbc, float* gWq_last, long rows, float scale,
                          hipStream_t s);

// ==================================================================================================================
// host

bool enc_fused_ok(const elg_encoder_args* a) {
    static const bool on = [] {
        const char* e = std::getenv("ELG_ENC_FUSED");
        return !(e && e[0] == '0');
    }();
    return on && a->N1 <= 128 && a->N1 >= 4 && a->ff_hidden >= 128 && (a->ff_hidden % 128) == 0 && a->ff_hidden <= 1024;
}

#define ENCF_TRY(x)                    \
    {                                  \
        const int rc_ = (x);           \
        if (rc_ != ELG_OK) return rc_; \
    }

// every fused kernel takes its LDS dynamically (most images exceed the 64 KB static limit)
template <typename K, typename A>
static int launch1(K kern, DynLds& optin, const char* what, int grid, size_t lds, hipStream_t s, const A& args) {
    if (!optin.opt_in(reinterpret_cast<const void*>(kern), lds)) return fail(ELG_ELAUNCH, std::string(what) + ": hipFuncSetAttribute failed");
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, args);
    return launch_status(what);
}
constexpr size_t LDS_STAT = sizeof(float) * 2 * 8 * ELG_E;
constexpr size_t LDS_2STAGE = sizeof(float) * 2 * 64 * WP128;
constexpr size_t LDS_TBUF = sizeof(float) * 8 * TBUF;

template <int NT, bool BF>
static int launch_f1_t(const EncF1& g, hipStream_t s) {
    constexpr int ROWS = NT * 16, PT = ROWS + 4;
    constexpr int kv = 2 * ROWS * 20 + 2 * 16 * PT;
    constexpr size_t lds = LDS_STAT + sizeof(float) * (96 * WP128 + (kv > 8 * TBUF ? kv : 8 * TBUF));
    static DynLds optin;
    return launch1(enc_f1_kernel<NT, BF>, optin, "enc_f1", g.B * 4, lds, s, g);
}
template <bool BF>
static int launch_f1_b(const EncF1& g, hipStream_t s) {
    const int nt = (g.N1 + 15) / 16;
    if (nt <= 2) return launch_f1_t<2, BF>(g, s);
    if (nt <= 4) return launch_f1_t<4, BF>(g, s);
    if (nt <= 7) return launch_f1_t<7, BF>(g, s);
    return launch_f1_t<8, BF>(g, s);
}
static int launch_f1(const EncF1& g, bool bf, hipStream_t s) { return bf ? launch_f1_b<true>(g, s) : launch_f1_b<false>(g, s); }
template <bool BF>
static int launch_f2_b(const EncF2& g, hipStream_t s) {
    static DynLds optin;
    return launch1(enc_f2_kernel<BF>, optin, "enc_f2", g.B * (g.FF >> 7), LDS_STAT + LDS_2STAGE + LDS_TBUF, s, g);
}
static int launch_f2(const EncF2& g, bool bf, hipStream_t s) { return bf ? launch_f2_b<true>(g, s) : launch_f2_b<false>(g, s); }
template <bool BF>
static int launch_f3_b(const EncF3& g, hipStream_t s) {
    static DynLds optin;
    return launch1(enc_f3_kernel<BF>, optin, "enc_f3", g.B * (g.ntab > 0 ? g.ntab : 1), LDS_STAT + LDS_2STAGE + LDS_TBUF, s, g);
}
static int launch_f3(const EncF3& g, bool bf, hipStream_t s) { return bf ? launch_f3_b<true>(g, s) : launch_f3_b<false>(g, s); }
template <bool BF>
static int launch_b0_b(const EncB0& g, hipStream_t s) {
    static DynLds optin;
    return launch1(enc_b0_kernel<BF>, optin, "enc_b0", g.B * g.ntab, LDS_2STAGE + LDS_TBUF, s, g);
}
static int launch_b0(const EncB0& g, bool bf, hipStream_t s) { return bf ? launch_b0_b<true>(g, s) : launch_b0_b<false>(g, s); }
template <bool BF>
static int launch_b1_b(const EncB1& g, hipStream_t s) {
    static DynLds optin;
    return launch1(enc_b1_kernel<BF>, optin, "enc_b1", g.B * (g.FF >> 7), LDS_STAT + LDS_2STAGE + LDS_TBUF, s, g);
}
static int launch_b1(const EncB1& g, bool bf, hipStream_t s) { return bf ? launch_b1_b<true>(g, s) : launch_b1_b<false>(g, s); }
template <int NT, bool BF>
static int launch_b2_t(const EncB2& g, hipStream_t s) {
    constexpr size_t lds = LDS_STAT + sizeof(float) * (32 * WP128 + b2_attn_floats<NT>());
    static_assert(lds <= 160 * 1024, "enc_b2: LDS image over 160 KB");
    static DynLds optin;
    return launch1(enc_b2_kernel<NT, BF>, optin, "enc_b2", g.B * 4, lds, s, g);
}
template <bool BF>
static int launch_b2_b(const EncB2& g, hipStream_t s) {
    const int nt = (g.N1 + 15) / 16;
    if (nt <= 2) return launch_b2_t<2, BF>(g, s);
    if (nt <= 4) return launch_b2_t<4, BF>(g, s);
    if (nt <= 7) return launch_b2_t<7, BF>(g, s);
    return launch_b2_t<8, BF>(g, s);
}
static int launch_b2(const EncB2& g, bool bf, hipStream_t s) { return bf ? launch_b2_b<true>(g, s) : launch_b2_b<false>(g, s); }

int enc_fused_fwd(const elg_encoder_args* a, hipStream_t s) {
    const int B = a->B, N1 = a->N1, FF = a->ff_hidden, NS = FF >> 7, NL = a->n_layers;
    const bool tsp = a->problem == ELG_PROBLEM_TSP, bf = a->precision == 1;
    const EncWs w = enc_ws(B, N1, NL, FF, a->save);
    const long R = w.R;
    float* ws = a->ws;
    auto lay = [&](int l) { return ws + w.layer0 + w.layer_stride * l; };
    auto xin_of = [&](int l) -> float* { return (l == 0 || !a->save) ? ws + w.X0 : lay(l - 1) + w.oXout; };
    auto norm_pro = [&](int l, float* xout) {          // the add & norm that closes layer l
        EncPro p{};
        const elg_enc_layer& L = a->W.layer[l];
        p.mode = 1; p.np = NS; p.P = ws + w.P; p.pstride = R * ELG_E;
        p.res = nullptr; p.bias = nullptr; p.gamma = L.g2; p.beta = L.b2;        // (x1 + bias ride in partial 0)
        p.xhat = a->save ? lay(l) + w.oXH2 : nullptr; p.rstd = a->save ? lay(l) + w.oRS2 : nullptr; p.xout = xout;
        p.eps = a->eps;
        return p;
    };
    for (int l = 0; l < NL; ++l) {
        const elg_enc_layer& L = a->W.layer[l];
        float* lb = lay(l);
        EncF1 f1{};
        if (l == 0) {
            f1.pro.mode = 0; f1.pro.xy = a->xy; f1.pro.demand = tsp ? nullptr : a->demand;
            f1.pro.Wd = tsp ? nullptr : a->W.emb_depot_w; f1.pro.bd = tsp ? nullptr : a->W.emb_depot_b;
            f1.pro.Wn = a->W.emb_w; f1.pro.bn = a->W.emb_b; f1.pro.xout = xin_of(0);
        } else {
            f1.pro = norm_pro(l - 1, xin_of(l));
        }
        f1.Wq = L.Wq; f1.Wk = L.Wk; f1.Wv = L.Wv;
        f1.QKV = a->save ? lb + w.oQKV : nullptr; f1.O = lb + w.oO; f1.LSE = a->save ? lb + w.oLSE : nullptr;
        f1.B = B; f1.N1 = N1;
        ENCF_TRY(launch_f1(f1, bf, s))
        EncF2 f2{};
        f2.O = lb + w.oO; f2.Xin = xin_of(l); f2.Wc = L.Wc; f2.bc = L.bc; f2.g1 = L.g1; f2.b1 = L.b1; f2.W1 = L.W1; f2.bf1 = L.bf1;
        f2.W2 = L.W2; f2.bf2 = L.bf2; f2.X1 = lb + w.oX1; f2.XH1 = a->save ? lb + w.oXH1 : nullptr; f2.RS1 = a->save ? lb + w.oRS1 : nullptr;
        f2.H = a->save ? lb + w.oH : nullptr; f2.P = ws + w.P; f2.pstride = R * ELG_E; f2.B = B; f2.N1 = N1; f2.FF = FF; f2.eps = a->eps;
        ENCF_TRY(launch_f2(f2, bf, s))
    }
    EncF3 f3{};
    if (NL > 0) f3.pro = norm_pro(NL - 1, a->enc);
    else { f3.pro.mode = 2; f3.pro.xout = a->enc; }
    f3.B = B; f3.N1 = N1; f3.pb_tab = -1;
    if (a->K) {
        if (!a->V || !a->PK || !a->pb || !a->Q1 || !a->W.dec_Wk || !a->W.dec_Wv || !a->W.dec_Wc || !a->W.dec_bc || !a->W.dec_Wq_last)
            return fail(ELG_EINVAL, "encoder: decoder tables requested but a buffer / weight is null");
        if (tsp && (!a->Q2 || !a->W.dec_Wq_first)) return fail(ELG_EINVAL, "encoder: TSP needs Q2 / Wq_first");
        const float inv_sqrt_e = 0.08838834764831845f;
        int n = 0;
        auto tab = [&](const float* W, int sm, int sk, float alpha, float* out) {
            f3.W[n] = W; f3.sm[n] = sm; f3.sk[n] = sk; f3.alpha[n] = alpha; f3.out[n] = out; ++n;
        };
        tab(a->W.dec_Wk, ELG_E, 1, 1.f, a->K);
        tab(a->W.dec_Wv, ELG_E, 1, 1.f, a->V);
        tab(a->W.dec_Wq_last, tsp ? ELG_E : ELG_E + 1, 1, 1.f, a->Q1);
        f3.pb_tab = n;
        tab(a->W.dec_Wc, 1, ELG_E, inv_sqrt_e, a->PK);            // PK = enc Wc / sqrt(E): W(m = n, k) = Wc[k][n]
        if (tsp) tab(a->W.dec_Wq_first, ELG_E, 1, 1.f, a->Q2);
        f3.ntab = n; f3.bc = a->W.dec_bc; f3.pb = a->pb; f3.pb_scale = inv_sqrt_e;
        if (!tsp) { f3.wl_src = a->W.dec_Wq_last; f3.wl = a->wl; }
    } else if (NL == 0) return ELG_OK;
    return launch_f3(f3, bf, s);
}

int enc_fused_bwd(const elg_encoder_bwd_args* ba, DwList& dw, hipStream_t s) {
    const elg_encoder_args* a = &ba->fwd;
    const int B = a->B, N1 = a->N1, FF = a->ff_hidden, NS = FF >> 7, NL = a->n_layers;
    const bool tsp = a->problem == ELG_PROBLEM_TSP, bf = a->precision == 1;
    const EncWs w = enc_ws(B, N1, NL, FF, 1);
    const EncWs2 w2 = enc_ws2(B, N1, NL, FF);
    const long R = w.R;
    const elg_enc_weights& G = ba->G;
    float* ws = a->ws;
    float* s2 = ba->ws2;
    const float inv_sqrt_e = 0.08838834764831845f;
    auto need = [&](const float* p, const char* what) -> int { return p ? ELG_OK : fail(ELG_EINVAL, std::string("encoder bwd: null ") + what); };
    // ---- transposed copies of the layer weights
    {
        WtBatch bt{};
        int tiles = 0;
        auto add = [&](const float* src, float* dst, int rows, int cols) {
            WtJob& j = bt.job[bt.njobs++];
            j.src = src; j.dst = dst; j.rows = rows; j.cols = cols; j.tile0 = tiles;
            tiles += (rows >> 5) * (cols >> 5);
        };
        for (int l = 0; l < NL; ++l) {
            const elg_enc_layer& L = a->W.layer[l];
            float* t = s2 + w2.WT + w2.wt_stride * l;
            add(L.Wq, t, ELG_E, ELG_E);
            add(L.Wk, t + ELG_E * ELG_E, ELG_E, ELG_E);
            add(L.Wv, t + 2 * ELG_E * ELG_E, ELG_E, ELG_E);
            add(L.Wc, t + 3 * ELG_E * ELG_E, ELG_E, ELG_E);
            add(L.W1, t + 4 * ELG_E * ELG_E, FF, ELG_E);                       // W1 (ff,128) -> W1T (128,ff)
            add(L.W2, t + 4 * ELG_E * ELG_E + ELG_E * FF, ELG_E, FF);         // W2 (128,ff) -> W2T (ff,128)
        }
        (void)hipGetLastError();
        hipLaunchKernelGGL(enc_wt_kernel, dim3(tiles), dim3(256), 0, s, bt);
        ENCF_TRY(launch_status("enc_wt"))
    }
    // ---- d enc from the decoder tables (autograd of set_kv / fold_decoder_tables): one partial per cotangent
    EncB0 b0{};
    {
        int n = 0;
        auto tab = [&](const float* Gc, const float* W, int sm, int sk, float alpha) {
            b0.G[n] = Gc; b0.W[n] = W; b0.sm[n] = sm; b0.sk[n] = sk; b0.alpha[n] = alpha; ++n;
        };
        b0.pb_tab = -1;
        if (ba->gK) {       // d enc[r][k] += sum_n gK[r][n] Wk[n][k]: W(m = k, kk = n) = Wk[n][k]
            ENCF_TRY(need(G.dec_Wk, "d Wk"))
            tab(ba->gK, a->W.dec_Wk, 1, ELG_E, 1.f);
            ENCF_TRY(dw.add(ba->gK, ELG_E, a->enc, ELG_E, (float*)G.dec_Wk, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        }
        if (ba->gV) {
            ENCF_TRY(need(G.dec_Wv, "d Wv"))
            tab(ba->gV, a->W.dec_Wv, 1, ELG_E, 1.f);
            ENCF_TRY(dw.add(ba->gV, ELG_E, a->enc, ELG_E, (float*)G.dec_Wv, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        }
        if (ba->gQ1) {
            ENCF_TRY(need(G.dec_Wq_last, "d Wq_last"))
            const int ldq = tsp ? ELG_E : ELG_E + 1;
            tab(ba->gQ1, a->W.dec_Wq_last, 1, ldq, 1.f);
            ENCF_TRY(dw.add(ba->gQ1, ELG_E, a->enc, ELG_E, (float*)G.dec_Wq_last, ldq, ELG_E, ELG_E, nullptr, 1.f))
        }
        if (tsp && ba->gQ2) {
            ENCF_TRY(need(G.dec_Wq_first, "d Wq_first"))
            tab(ba->gQ2, a->W.dec_Wq_first, 1, ELG_E, 1.f);
            ENCF_TRY(dw.add(ba->gQ2, ELG_E, a->enc, ELG_E, (float*)G.dec_Wq_first, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        }
        if (ba->gPK) {      // PK = enc Wc / sqrt(E): d enc[r][k] += sum_n gPK[r][n] Wc[k][n] / sqrt(E) ; d Wc = enc^T gPK / sqrt(E)
            ENCF_TRY(need(G.dec_Wc, "d Wc"))
            b0.pb_tab = n; b0.gpb = ba->gpb; b0.bc = a->W.dec_bc;
            tab(ba->gPK, a->W.dec_Wc, ELG_E, 1, inv_sqrt_e);
            ENCF_TRY(dw.add(a->enc, ELG_E, ba->gPK, ELG_E, (float*)G.dec_Wc, ELG_E, ELG_E, ELG_E, nullptr, inv_sqrt_e))
        } else if (ba->gpb) return fail(ELG_EINVAL, "encoder bwd: gpb without gPK");
        b0.ntab = n; b0.PX = s2 + w2.PX; b0.pstride = R * ELG_E; b0.B = B; b0.N1 = N1; b0.g_enc = ba->g_enc;
        if (n == 0 && !ba->g_enc) return fail(ELG_EINVAL, "encoder bwd: no cotangent given");
        if (n > 0) ENCF_TRY(launch_b0(b0, bf, s))
    }
    if (ba->gpb || (ba->gwl && !tsp)) {
        if (ba->gpb) ENCF_TRY(need(G.dec_bc, "d bc"))
        ENCF_TRY(launch_fold_small_bwd(a->enc, ba->gpb, tsp ? nullptr : ba->gwl, (float*)G.dec_bc, (float*)G.dec_Wq_last, R, inv_sqrt_e, s))
    }
    // ---- layers, last to first
    const float* base = b0.ntab > 0 ? nullptr : ba->g_enc;      // (with table cotangents, d enc itself rides in partial 0)
    int np = b0.ntab;
    for (int l = NL - 1; l >= 0; --l) {
        const elg_enc_layer& GL = G.layer[l];
        const elg_enc_layer& L = a->W.layer[l];
        float* gS = s2 + w2.lay0 + w2.lay_stride * l;
        float* gH = gS + R * ELG_E;
        float* gY = gH + R * FF;
        float* dQKV = gY + R * ELG_E;
        float* lb = ws + w.layer0 + w.layer_stride * l;
        const float* Xin = (l == 0) ? ws + w.X0 : ws + w.layer0 + w.layer_stride * (l - 1) + w.oXout;
        const float* wt = s2 + w2.WT + w2.wt_stride * l;
        EncB1 b1{};
        b1.pro.base = base; b1.pro.P = s2 + w2.PX; b1.pro.pstride = R * ELG_E; b1.pro.np = np;
        b1.pro.xhat = lb + w.oXH2; b1.pro.rstd = lb + w.oRS2; b1.pro.gamma = L.g2; b1.pro.dgamma = (float*)GL.g2; b1.pro.dbeta = (float*)GL.b2;
        b1.pro.dout = gS;
        b1.W2T = wt + 4 * ELG_E * ELG_E + ELG_E * FF; b1.W1T = wt + 4 * ELG_E * ELG_E; b1.H = lb + w.oH; b1.gH = gH;
        b1.P1 = s2 + w2.P1; b1.pstride = R * ELG_E; b1.B = B; b1.N1 = N1; b1.FF = FF;
        ENCF_TRY(launch_b1(b1, bf, s))
        ENCF_TRY(dw.add(gS, ELG_E, lb + w.oH, FF, (float*)GL.W2, FF, ELG_E, FF, (float*)GL.bf2, 1.f))
        ENCF_TRY(dw.add(gH, FF, lb + w.oX1, ELG_E, (float*)GL.W1, ELG_E, FF, ELG_E, (float*)GL.bf1, 1.f))
        EncB2 b2{};
        b2.pro.base = nullptr;                                   // (dS2 rides in partial 0)
        b2.pro.P = s2 + w2.P1; b2.pro.pstride = R * ELG_E; b2.pro.np = NS;
        b2.pro.xhat = lb + w.oXH1; b2.pro.rstd = lb + w.oRS1; b2.pro.gamma = L.g1; b2.pro.dgamma = (float*)GL.g1; b2.pro.dbeta = (float*)GL.b1;
        b2.pro.dout = gY;
        b2.WqT = wt; b2.WkT = wt + ELG_E * ELG_E; b2.WvT = wt + 2 * ELG_E * ELG_E; b2.WcT = wt + 3 * ELG_E * ELG_E;
        b2.QKV = lb + w.oQKV; b2.O = lb + w.oO; b2.LSE = lb + w.oLSE; b2.dQKV = dQKV; b2.P2 = s2 + w2.PX; b2.pstride = R * ELG_E;
        b2.B = B; b2.N1 = N1;
        ENCF_TRY(launch_b2(b2, bf, s))
        ENCF_TRY(dw.add(gY, ELG_E, lb + w.oO, ELG_E, (float*)GL.Wc, ELG_E, ELG_E, ELG_E, (float*)GL.bc, 1.f))
        ENCF_TRY(dw.add(dQKV, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wq, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        ENCF_TRY(dw.add(dQKV + ELG_E, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wk, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        ENCF_TRY(dw.add(dQKV + 2 * ELG_E, 3 * ELG_E, Xin, ELG_E, (float*)GL.Wv, ELG_E, ELG_E, ELG_E, nullptr, 1.f))
        base = nullptr;                                        // (dY rides in partial 0)
        np = 4;
    }
    // ---- input embeddings: d x0 = dY(layer 0) + the four partials
    {
        const int rpb = 32;
        (void)hipGetLastError();
        hipLaunchKernelGGL(enc_embed_bwd2_kernel, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(512), 0, s, a->xy,
                           tsp ? nullptr : a->demand, base, s2 + w2.PX, R * ELG_E, np, tsp ? nullptr : (float*)G.emb_depot_w,
                           tsp ? nullptr : (float*)G.emb_depot_b, (float*)G.emb_w, (float*)G.emb_b, N1, R, rpb);
        ENCF_TRY(launch_status("enc_embed_bwd"))
    }
    return dw.launch();          // every weight gradient of the call in one grouped launch
}

}  // namespace elg

#ifdef ELG_STAMPS
extern "C" int elg_enc_debug_stamps(void* p) {
    unsigned long long* v = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(elg::g_enc_stamps), &v, sizeof(v)) == hipSuccess ? 0 : 1;
}
#endif
