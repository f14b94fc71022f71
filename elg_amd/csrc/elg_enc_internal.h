// Internals shared by the two translation units of the encoder (elg_enc.hip: the per-GEMM kernels of every size and the
// grouped weight-gradient launch; elg_enc_fused.hip: the per-instance fused layer kernels for N1 <= 128).  Not part of the ABI.
#pragma once
#include "elg_common.h"
#include "../../include/elg_hip.h"
#include <string>

namespace elg {
int fail(int code, const std::string& msg);
int launch_status(const char* what);

using f32x4 = __attribute__((ext_vector_type(4))) float;

// workspace layout of elg_encoder_fwd (floats)
struct EncWs {
    long R, X0, tmp, P, layer0, layer_stride;
    long oQKV, oO, oLSE, oXH1, oRS1, oX1, oH, oXH2, oRS2, oXout, total;
};
EncWs enc_ws(int B, int N1, int n_layers, int ff, int save);

// scratch layout of elg_encoder_bwd (floats)
struct EncWs2 {
    long R, gX, gO, gT, lay0, lay_stride, delta, PX, P1, WT, wt_stride, DW, dw_floats, total;
    // per layer (offsets from lay0 + l * lay_stride): gS (R,128) | gH (R,ff) | gY (R,128) | dQKV (R,384)
    // per layer of WT (offsets from WT + l * wt_stride): WqT | WkT | WvT | WcT (128,128 each) | W1T (128,ff) | W2T (ff,128)
};
EncWs2 enc_ws2(int B, int N1, int n_layers, int ff);
constexpr int ENC_PX = 5;          // partial-sum buffers of the tables' backward / of the attention block's input gradient

// one job of the grouped weight-gradient launch: dW[M,N] += alpha dY^T X over the rows (db += column sums of dY)
struct DwJob {
    const float* dY; const float* X; float* dW; float* db;
    int ldy, ldx, ldw, M, N, tile0; float alpha;
};
constexpr int DW_MAX_JOBS = 48;
struct DwBatch {
    DwJob job[DW_MAX_JOBS];
    int njobs, ntiles, rows, rows_per_split;
    float* scratch;                   // [split][tile][128][128] partial tiles
    int bf16;                         // 1: operands rounded to bf16, v_mfma_f32_32x32x16_bf16 (the encoder's bf16 mode)
};
struct DwList {
    DwBatch bt;
    long rows;
    hipStream_t s;
    float* scratch; long scratch_floats;
    DwList(long rows_, hipStream_t s_, float* scratch_, long scratch_floats_) : rows(rows_), s(s_), scratch(scratch_), scratch_floats(scratch_floats_) {
        bt.njobs = 0; bt.ntiles = 0; bt.scratch = nullptr; bt.bf16 = 0;
    }
    int add(const float* dY, int ldy, const float* X, int ldx, float* dW, int ldw, int M, int N, float* db, float alpha);
    int launch();
};

// the fused path (elg_enc_fused.hip): N1 <= 128, ff_hidden a multiple of 128, <= 1024
bool enc_fused_ok(const elg_encoder_args* a);
int enc_fused_fwd(const elg_encoder_args* a, hipStream_t s);
int enc_fused_bwd(const elg_encoder_bwd_args* ba, DwList& dw, hipStream_t s);
}  // namespace elg
