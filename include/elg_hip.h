/* elg_hip.h -- C ABI of libelg_hip.so, the MI355X (gfx950) ELG-POMO rollout engine.
 *
 * Drop-in boundary for the reference's construction hot path (gaocrr/ELG).  The reference has no
 * FFI: its "plugin interface" is the duck-typed Python protocol CVRPEnv / CVRPModel / rollout
 * (SURVEY.md section 8b).  These entry points are what a binding for that protocol calls; the
 * Python classes in elg_amd/{CVRP,TSP}/ are that binding (ctypes, raw device pointers).
 *
 * Conventions: every pointer is a DEVICE pointer into caller-owned memory (no ownership transfer),
 * row-major, fp32 / int32 unless stated; `stream` is a hipStream_t passed as void*; every function
 * only enqueues work on `stream` (no allocation, no synchronisation -- graph-capturable) and
 * returns 0 or a negative ELG_E* code; elg_last_error() gives the message for the calling thread.
 */
#ifndef ELG_HIP_H
#define ELG_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELG_OK 0
#define ELG_EINVAL (-1)     /* bad argument / unsupported shape   (Python: ValueError)        */
#define ELG_ELAUNCH (-2)    /* HIP launch failure                 (Python: RuntimeError)      */
#define ELG_ENOTIMPL (-3)   /* configuration not built            (Python: NotImplementedError) */

#define ELG_PROBLEM_CVRP 0
#define ELG_PROBLEM_TSP 1
#define ELG_MODE_GREEDY 0   /* argmax, ties -> lowest node index  (CVRPModel.py:70-73)         */
#define ELG_MODE_SAMPLE 1   /* categorical sample                 (CVRPModel.py:59-68)         */
#define ELG_MODE_FORCED 2   /* teacher forcing: actions given, probabilities returned          */

/* layout of the folded local-policy table buffer (floats); rows padded to ELG_LOC_ROWS slots */
#define ELG_LOC_ROWS 64
#define ELG_LOC_LA 0        /* [4][3]   q_h^T Wk We / sqrt(8)          (pad to 16 floats)      */
#define ELG_LOC_LT 16       /* [64][4]  q_h . Wk(be + PE[j]) / sqrt(8)                        */
#define ELG_LOC_LAV 272     /* [32][3]  Wv We                                                 */
#define ELG_LOC_LCV 368     /* [64][32] Wv (be + PE[j])                                       */
#define ELG_LOC_LWC 2416    /* [32][32] multi_head_combine.weight                             */
#define ELG_LOC_LBC 3440    /* [32]     multi_head_combine.bias                               */
#define ELG_LOC_LWE 3472    /* [32][3]  We / sqrt(32)                                         */
#define ELG_LOC_LPE 3568    /* [64][32] (be + PE[j]) / sqrt(32)                               */
#define ELG_LOC_SIZE 5616
#define ELG_MAX_ENS 4       /* model_params['ensemble_size'] <= 4 local policies (models.py:296-298)   */

const char* elg_version(void);
const char* elg_last_error(void);

/* utils.augment_xy_data_by_8_fold (CVRP/utils.py:69-87): in (B,N,2) -> out (8B,N,2). */
int elg_aug8(const float* xy_in, float* xy_out, int B, int N, void* stream);

/* Distance matrix (CVRPEnv.py:148, TSPEnv.py:65): xy (B,N,2) -> dist (B,N,N). */
int elg_dist_matrix(const float* xy, float* dist, int B, int N, void* stream);

/* Per-instance neighbour tables: for every node c the list of all nodes sorted by
 * (dist(c,n), n) ascending, with the distance and polar angle atan2(y_n-y_c, x_n-x_c).
 * Replaces the per-step take_along_dim / atan2 / topk of CVRPEnv.get_cur_feature
 * (CVRPEnv.py:291-318) and models.py:55-120,355-403.  Outputs are (B,N,N).  N <= 8192. */
int elg_nbr_tables(const float* xy, int32_t* nbr_idx, float* nbr_dist, float* nbr_theta,
                   int B, int N, void* stream);

/* Closed-tour length (CVRPEnv._get_reward / compute_unscaled_reward, CVRPEnv.py:251-288;
 * TSPEnv.py:158-184).  xy (Bxy,N,2) with instance b using row b % Bxy... no: Bxy == B.
 * tour (B,M,T) int64, out (B,M) = +length.  rounding != 0 rounds every segment (half-even). */
int elg_route_length(const float* xy, const int64_t* tour, float* out, int B, int M, int T, int N,
                     int rounding, void* stream);

/* Arguments of the construction kernels (three kernels behind one entry point: `variant`, `lds_stage`). */
typedef struct elg_rollout_args {
    int32_t problem;        /* ELG_PROBLEM_*                                                    */
    int32_t B, M, N1;       /* instances, POMO trajectories per instance, nodes (CVRP: + depot) */
    int32_t K;              /* model_params['local_size'][0]                                    */
    int32_t Tmax;           /* time capacity of actions / probs                                 */
    int32_t mode;           /* ELG_MODE_*                                                       */
    int32_t Tforced;        /* time extent of `forced`                                          */
    int32_t has_local;      /* decoder.local (add_local_policy called) and model_params.ensemble */
    int32_t has_penalty;    /* model_params.distance_penalty                                    */
    int32_t max_steps;      /* <=0: run every trajectory to completion; >0: at most that many   */
    int32_t do_decode;      /* 0: env update only with forced actions (CVRPEnv.step)            */
    int32_t do_update;      /* 0: decode only (CVRPModel.one_step_rollout)                      */
    int32_t use_state;      /* 1: load/store the st_* arrays (step-wise protocol)               */
    int32_t waves;          /* wavefronts per workgroup: 8                                      */
    int32_t tiles;          /* workgroups per instance                                          */
    int32_t lds_stage;      /* 1: keep the instance's tables on chip (N1 <= 112): fused rollouts run the
                               cooperative lockstep MFMA kernel, step-wise calls the LDS-staged kernel   */
    int32_t dump_T;         /* time extent of full_probs (0 = no dump)                          */
    float xi;               /* model_params.xi                                                  */
    float clip;             /* model_params.logit_clipping                                      */
    float inv_ens;          /* 1 / ensemble_size                                                */
    int32_t variant;        /* 0: pick the kernel by shape (cooperative MFMA kernel for N1 <= 112, node-streaming MFMA kernel
                               for 128 < N1 <= 1024); 1: the one-wavefront-per-trajectory kernel for any N1 (what the step-wise
                               protocol and 112 < N1 <= 128 always use; the A/B reference of the parity tests);
                               2: the one-wavefront-per-trajectory N1 > 1024 kernel (runtime node loops, needs `scratch`) for any
                               N1: the tests' reference at Vrp-Set-XXL sizes; 3: the matrix-core N1 > 1024 kernel (what variant 0
                               picks for 1024 < N1 <= 8192: 16 lockstep trajectories per workgroup, K / V / PK streamed in
                               MFMA-fragment order, score rows in `scratch`) for any N1; 4: the split-group form of the
                               cooperative kernel (N1 <= 112: two independent 4-wave groups per workgroup, LDS-counter barriers;
                               bit-identical results, measured 9 % slower: DESIGN 4.1); 5: the split-group form at four waves per
                               SIMD (two groups of eight waves, 128 registers; bit-identical, 14 % slower: DESIGN 7)      */
    int32_t dump_logits;    /* what full_probs receives: 0 probabilities, 1 the clipped + masked logits
                               clip * tanh(s) (-inf at closed nodes), 2 the scores s before the clip            */
    int32_t euclidean;      /* model_params.euclidean: local-policy slot features (x, y) / norm relative to the current node
                               instead of (dist / norm, theta)   (models.py:95-125, TSP/models.py:67-75)            */
    int32_t ens;            /* model_params.ensemble_size (0 or 1: one local policy with local_size K).  > 1 (CVRP,
                               models.py:296-298,409-413): member i has its own folded tables loc + i * ELG_LOC_SIZE and its own
                               local_size Kens[i] (Kens[0] == K, which also stays the distance penalty's k); the members'
                               slot scores are summed and scaled by inv_ens.  Runs the one-wavefront-per-trajectory kernel
                               (N1 <= 1024); training goes through elg_rollout_bwd's replay (N1 <= 1024)                */
    int32_t Kens[ELG_MAX_ENS];
    int32_t precision;      /* 0: f32 (the parity mode: every product exact f32).  1: bf16 throughput mode (BASELINE configs[1]): the
                               glimpse score / output and pointer products take bf16 operands on v_mfma_f32_16x16x32_bf16 with f32
                               accumulation; softmax, masks, local policy, environment stay f32.  Honoured by the cooperative kernel
                               (N1 <= 112; training and evaluation), the streaming kernel (128 < N1 <= 1024, evaluation) and the
                               N1 > 1024 kernel (evaluation); the one-wavefront kernels and the N1 > 128 training forward compute in
                               f32 whatever it says.                                                                     */
    uint64_t seed;          /* sampling seed (Philox key)                                       */
    const float* Kmat;      /* (B,N1,128) decoder.Wk enc                                        */
    const float* Vmat;      /* (B,N1,128) decoder.Wv enc                                        */
    const float* PK;        /* (B,N1,128) enc Wc / sqrt(128)  (pointer keys folded with combine) */
    const float* pb;        /* (B,N1)     enc . bc / sqrt(128)                                  */
    const float* Q1;        /* (B,N1,128) CVRP: Wq_last[:, :128] enc ; TSP: Wq_last enc         */
    const float* Q2;        /* (B,N1,128) TSP: Wq_first enc ; CVRP: NULL                        */
    const float* wl;        /* (128)      CVRP: Wq_last[:, 128] (load column)                   */
    const float* xy;        /* (B,N1,2)                                                         */
    const float* demand;    /* (B,N1) CVRP, demand[:,0] = 0                                     */
    const int32_t* nbr_idx; /* (B,N1,N1) from elg_nbr_tables                                    */
    const float* nbr_dist;
    const float* nbr_theta;
    const float* loc;       /* (max(ens,1) x ELG_LOC_SIZE) folded local-policy tables, NULL if !has_local */
    const int32_t* starts;  /* (M) POMO start nodes (CVRPModel.py:46-51 / TSPModel.py:30-34)    */
    const int32_t* forced;  /* (B,M,Tforced) or NULL                                            */
    const float* uniforms;  /* (B,M,Tmax) externally drawn U[0,1) or NULL (Philox)              */
    int32_t* st_cur;        /* (B,M)   state, step-wise protocol                                */
    int32_t* st_cnt;        /* (B,M)                                                            */
    int32_t* st_fin;        /* (B,M)                                                            */
    int32_t* st_first;      /* (B,M)   TSP first node                                           */
    float* st_load;         /* (B,M)                                                            */
    float* st_len;          /* (B,M)                                                            */
    uint64_t* st_vis;       /* (B,M,ceil(N1/64)) visited bitmask                                */
    int32_t* actions;       /* (B,M,Tmax) out                                                   */
    float* probs;           /* (B,Tmax,M) out, probability of the chosen node; NULL: not wanted -- a greedy construction
                               with N1 > 128 then forms no softmax normaliser (the reference's greedy rollout returns no
                               probabilities: CVRPModel.py:70-73, utils.py:24-25)                                    */
    float* reward;          /* (B,M) out, -tour length on `xy`                                  */
    int32_t* tlen;          /* (B,M) out, number of steps taken                                 */
    float* full_probs;      /* (B,M,dump_T,N1) out or NULL: whole probability rows (tests)      */
    /* training rows (all NULL for inference): saved per decode step so that the backward needs no
     * replay of the glimpse.  Row index r = t*M + m (time-major), Rcap = Tmax*M rows per instance;
     * rows of steps that are not decoded are left untouched (caller zero-fills). */
    float* trA;             /* (B,8,Rcap,N1)  glimpse attention weights a_h[n]                  */
    float* trPC;            /* (B,Rcap,N1)    p[n] * clip * (1 - tanh^2)  (softmax x clip Jacobian) */
    float* trCsel;          /* (B,Rcap)       clip * (1 - tanh^2) at the chosen node            */
    float* trQ;             /* (B,Rcap,128)   glimpse query                                     */
    float* trO;             /* (B,Rcap,128)   glimpse output                                    */
    float* trLoad;          /* (B,Rcap)       load at the step (CVRP)                           */
    int32_t* trSlot;        /* (B,Rcap,48)    node of every k-NN slot (-1: none, -2: depot slot, masked) */
    float* trF;             /* (B,Rcap,3,48)  local-policy features of every slot (NULL: not saved)   */
    uint64_t* trMask;       /* (B,Rcap,2)     feasibility mask words of the row (bit n = node n closed); with it the
                               cooperative kernel (N1 <= 112) may skip trA: the backward recomputes a_h from q, K   */
    float* scratch;         /* elg_rollout_scratch_floats() floats of workspace (fused rollouts with N1 > 128), else NULL */
    float* trLse;           /* (B,Rcap,8)     with trMask: log2 of the glimpse softmax denominator per head, in the units of
                               s log2(e) / 4, so that a_h[n] = exp2(q_h.K_h[n] log2(e) / 4 - trLse) (NULL: not saved) */
} elg_rollout_args;

/* POMO construction: CVRPEnv.reset/step + CVRPModel.one_step_rollout + utils.rollout fused into one
 * persistent launch (CVRP/utils.py:7-29), or single steps of it (use_state / max_steps). */
int elg_rollout_fwd(const elg_rollout_args* args, void* stream);
/* Which construction kernel the calling thread's last elg_rollout_fwd launched (0: none yet): the tests assert that a
   `variant` they name is the kernel that ran. */
#define ELG_KERNEL_WAVE 1        /* rollout_fwd_kernel: one wavefront per trajectory (step-wise protocol, variant 1, ensembles)  */
#define ELG_KERNEL_COOP 2        /* rollout_fwd_coop_kernel: N1 <= 112, lockstep trajectories on the matrix cores              */
#define ELG_KERNEL_COOP_SPLIT 3  /* rollout_fwd_coop2_kernel: the same with two independent 4-wave groups (variant 4)          */
#define ELG_KERNEL_STREAM 4      /* rollout_fwd_mt_kernel: 128 < N1 <= 1024, operands streamed from L2                         */
#define ELG_KERNEL_XL 5          /* rollout_fwd_xl_kernel: one wavefront per trajectory, runtime node loops (variant 2)        */
#define ELG_KERNEL_XM 6          /* rollout_fwd_xm_kernel: N1 > 1024 on the matrix cores (variant 3)                           */
#define ELG_KERNEL_COOP_WIDE 7   /* rollout_fwd_coop3_kernel: two groups of eight waves, four waves per SIMD (variant 5)        */
int elg_rollout_last_kernel(void);
/* Floats of elg_rollout_args.scratch a fused rollout of this shape needs (0: none): the fragment-major K / V / PK copies the
   128 < N1 <= 1024 kernel streams its matrix-core operands from; for N1 > 1024 (or variant 3) the same copies over
   64 ceil(N1 / 64) padded rows plus one score row per trajectory slot (16 per workgroup); variant 2: B M N1 score rows. */
int64_t elg_rollout_scratch_floats(int32_t B, int32_t M, int32_t N1, int32_t variant);

/* Backward of the chosen-node probabilities w.r.t. the per-instance tables and the folded local
 * tables (replaces autograd's tape over CVRP/utils.py:14-21 + models.py:322-423; train.py:112-125).
 * The recorded actions are replayed (no tape): every step is recomputed inside one wavefront.
 * Launch 1 (rollout_bwd_kernel) differentiates the softmax / clipping and emits, per decode row
 * r = m*T + t, what the dense part of the backward needs:
 *     rowDL[r,n] = d s[n] (pointer scores)      rowA[h,r,n] = a_h[n] (glimpse attention weights)
 *     rowQ[r], rowO[r] = glimpse query / output  rowLoad[r] = load seen by the query
 *     rowDU[r,j] = d u_slot[j] (local policy output)
 * The glimpse / pointer backward itself is batched dense algebra over the R = M*T rows of an instance
 *     dO = rowDL PK,  dA_h = dO_h V_h^T,  dS = a (dA - <dO_h, O_h>) / 4,  dQ_h = dS_h K_h,
 *     dK_h = dS_h^T Q_h,  dV_h = a_h^T dO_h,  dPK = rowDL^T rowO,  dpb = sum_r rowDL
 * and is run by the host as library GEMMs on the matrix cores (engine.py).
 * Launch 2 (local_bwd_kernel) replays the k-NN/local policy only and reduces the gradient of the
 * folded local tables on chip (register accumulators, one flush per workgroup) into gloc. */
typedef struct elg_bwd_args {
    elg_rollout_args fwd;   /* same tables; fwd.forced = recorded actions (B,M,T), fwd.Tforced = T  */
    int32_t T;              /* steps to replay                                                      */
    int32_t member;         /* internal (set by elg_rollout_bwd): ensemble member of a local-policy replay launch */
    const float* gprob;     /* (B,T,M) dJ/d prob[b,t,m]                                             */
    float* rowA;            /* (B,8,R,N1)  glimpse attention weights a_h[n],  R = M*T               */
    float* rowDL;           /* (B,R,N1)    d pointer score s[n]                                     */
    float* rowQ;            /* (B,R,128)   glimpse query q                                          */
    float* rowO;            /* (B,R,128)   glimpse output o                                         */
    float* rowLoad;         /* (B,R)       load at the step (CVRP), may be NULL                     */
    float* rowDU;           /* (B,R,48)    d u_slot (already includes 1/ensemble_size)              */
    float* gloc;            /* (max(fwd.ens,1) x ELG_LOC_SIZE) accumulated d loc, caller zeroes      */
    int32_t time_major;     /* 0: r = m*T + t (replay rows); 1: r = t*M + m (rows saved by the forward) */
    int32_t local_only;     /* 1: skip launch 1 (rows came from the forward), run the local replay only */
    int64_t row_stride;     /* rows per instance in rowDU (R, or Rcap for forward-saved rows)        */
} elg_bwd_args;
int elg_rollout_bwd(const elg_bwd_args* args, void* stream);

/* Local-policy backward over independent decode rows on the matrix cores (autograd of models.py:133-166 w.r.t.
 * the folded tables): 16 rows per wavefront tile, every contraction with a shared table as v_mfma_f32_16x16x4_f32.
 * loc (ELG_LOC_SIZE); trF (B,Rcap,3,48) / trSlot (B,Rcap,48) as saved by a training forward; rowDU (B,R,48) from
 * elg_rows_prep; gloc (ELG_LOC_SIZE) accumulated (caller zeroes).  n_slots = local_size (+1 for the CVRP depot).
 * T_dev (or NULL) / M: as elg_decoder_bwd_args.T_dev -- R = min(R, T_dev[0] * M) is taken on the device.
 * max_workgroups: 0 = one workgroup per CU (the kernel holds 464 registers per lane: it shares a CU with nothing); > 0 caps the
 * persistent grid, so that a caller that runs the kernel on a side stream leaves the other CUs to the work it overlaps with
 * (the training step: next to the encoder's latency-bound backward chain, on half the chip). */
int elg_local_bwd_rows(const float* loc, const float* trF, const int32_t* trSlot, const float* rowDU, float* gloc,
                       int B, int R, int64_t Rcap, int n_slots, const int32_t* T_dev, int M, int max_workgroups, void* stream);

/* Row-wise part of the glimpse backward (softmax backward of models.py:478-500 on the saved weights):
 * per decode row r and head h:  dS = a (dO_h V_h^T - <dO_h, O_h>) / 4,  dQ_h = dS K_h.
 * rowA, dS (B,8,R,N1); dO, rowO, dQ (B,R,128); Kmat, Vmat (B,N1,128).  dS may alias nothing else. */
int elg_glimpse_rows_bwd(const float* rowA, const float* dO, const float* rowO, const float* Kmat,
                         const float* Vmat, float* dS, float* dQ, int B, int R, int N1,
                         int64_t rowA_rows, int64_t rowO_rows, void* stream);
/* rowA_rows / rowO_rows: rows per (instance[, head]) in the rowA / rowO buffers (>= R; R for dense buffers). */

/* Whole glimpse backward of an instance batch in one launch on the matrix cores (v_mfma_f32_16x16x4_f32; the
 * autograd of models.py:478-500 + the K/V projections' inputs): per decode row r and head h
 *     dS = a (dO_h V_h^T - <dO_h, O_h>) / 4 (kept in registers),  dQ_h = dS K_h,
 *     dK_h = sum_r dS^T Q_h,  dV_h = sum_r a^T dO_h.
 * rowA (B,8,rowA_rows,N1); dO, dQ (B,R,128); rowO (B,rowO_rows,128); rowQ (B,rowQ_rows,128); Kmat, Vmat (B,N1,128);
 * dK_part, dV_part (splits,B,N1,128): partial sums over `splits` contiguous row ranges (the caller adds them).
 * rowMask (B,rowQ_rows,2) non-NULL: the weights a are not read but recomputed, a_h = softmax(q_h K_h^T / 4 + mask)
 * (rowA may then be NULL; N1 <= 128 nodes = two 64-bit mask words per row).  N1 <= 128. */
int elg_glimpse_bwd_fused(const float* rowA, const uint64_t* rowMask, const float* dO, const float* rowO,
                          const float* rowQ, const float* Kmat, const float* Vmat, float* dQ, float* dK_part,
                          float* dV_part, int B, int R, int N1, int64_t rowA_rows, int64_t rowO_rows,
                          int64_t rowQ_rows, int splits, void* stream);

/* ---- training-step glue (one launch each instead of chains of framework kernels) ----------------------------
 * REINFORCE / POMO loss with the shared baseline (reference CVRP/train.py:112-121, TSP/train.py:107-118):
 *   adv = reward - mean_m reward;  J[b,m] = -adv * sum_t log probs[b,t,m];  scaled: J / max_m adv.
 * probs is (B,T,M) with element strides (probs_bstride, probs_tstride, 1).  Outputs per instance: J_raw[b] and
 * J_scaled[b] = sum_m J, adv_max[b], and coef_*[b,m] = d(sum J)/d(sum_t log p[b,.,m]).  The caller picks the variant
 * and divides by B*M (the reference's .mean()). */
int elg_pomo_loss(const float* probs, const float* reward, int B, int T, int M, int64_t probs_bstride,
                  int64_t probs_tstride, float* J_raw, float* J_scaled, float* adv_max, float* coef_raw,
                  float* coef_scaled, void* stream);

/* The scaled variant of that loss together with its gradient w.r.t. the chosen probabilities, for the CVRP training step
 * (train.py:112-121 with scale_norm; the +1e-6 of CVRPModel.py:67-68 on the steps flagged in zero_steps (T int32, or NULL)):
 *   J_terms[b] = inv_count * sum_m (-adv / max_m adv) sum_t log p',  gprob[b,t,m] = inv_count * (-adv / max adv) / p',
 * p' = probs + 1e-6 zero_steps[t];  J = sum_b J_terms with inv_count = 1 / (B M) is the reference's .mean(); gprob is (B,T,M)
 * contiguous.  One launch instead of the loss kernel + five element-wise framework kernels of its autograd chain.
 * T_dev (device int32, or NULL): the rollout's longest trajectory; the steps from there on hold probability 1 by contract and are
 * not read (their gprob is the coefficient itself, as the division by 1 gives).  J_total (device float, or NULL) receives
 * sum_b J_terms[b], added in index order by the last workgroup to finish; `ticket` is the int32 word that elects it -- zero before
 * the first call, left zero by every call, required exactly when J_total is given. */
int elg_pomo_loss_grad(const float* probs, const float* reward, const int32_t* zero_steps, int B, int T, int M,
                       int64_t probs_bstride, int64_t probs_tstride, float inv_count, float* J_terms, float* gprob,
                       const int32_t* T_dev, float* J_total, int32_t* ticket, void* stream);

/* torch.optim.Adam update (L2 weight decay added to the gradient, bias correction; reference train.py:101) over n
 * floats in one launch.  grad / exp_avg / exp_avg_sq are flat.  Parameters: either flat (`param`), or left in place
 * and addressed through `param_table[k]` (device array of n_tensors device pointers) with element i of the flat
 * index space belonging to tensor k iff offsets[k] <= i < offsets[k+1] (device array, n_tensors+1 entries).
 * `step` counts from 1; grad is multiplied by grad_scale first (1/world for summed data-parallel gradients). */
int elg_adam_step(float* param, float* const* param_table, const int64_t* offsets, int n_tensors, const float* grad,
                  float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int64_t step, float grad_scale, void* stream);

/* Residual add + instance normalisation over the node axis (reference CVRP/models.py:506-527:
 * nn.InstanceNorm1d(C, affine=True, eps) on (a + b) viewed as (B, C, N)).  a, b, out, xhat, dout, ds: (B,N,C) f32,
 * b may be NULL; rstd: (B,C); gamma, beta, dgamma, dbeta: (C).  Forward saves xhat / rstd for the backward;
 * backward returns ds = d(a + b) and ACCUMULATES dgamma / dbeta (caller zeroes).  C multiple of 32. */
int elg_add_instnorm_fwd(const float* a, const float* b, const float* gamma, const float* beta, float* out,
                         float* xhat, float* rstd, int B, int N, int C, float eps, void* stream);
int elg_add_instnorm_bwd(const float* dout, const float* xhat, const float* rstd, const float* gamma, float* ds,
                         float* dgamma, float* dbeta, int B, int N, int C, void* stream);

/* fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32, exact f32) for the encoder's nn.Linear layers and
 * their backward (reference CVRP/models.py:240-269,550-561):
 *   C[M,N] (+)= op(A)[M,K] op(B)[K,N] (+ bias[N]) (ReLU);  transA: A stored KxM;  transB: B stored NxK.
 * split_k > 1 accumulates with f32 atomics into a caller-zeroed C (weight gradients, K = batch*nodes).
 * a_rowsum (M floats, caller-zeroed, may be NULL): += sum_k op(A)[m][k] -- the bias gradient when C = dY^T X. */
int elg_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                 int lda, int ldb, int ldc, int transA, int transB, int relu, int split_k, float* a_rowsum,
                 void* stream);
/* The same product for a two-level batch in one launch: C(o, i) = alpha op(A(o, i)) op(B(o, i)), o < n_outer, i < n_inner,
 * X(o, i) = X + o sX_outer + i sX_inner (element strides; an inner stride may be a column offset of a wider matrix, e.g.
 * head i of a (rows, 128) buffer: sX_inner = 16 with ldx = 128).  The row contractions of the decoder's replay backward for
 * N + 1 > 128 nodes (autograd of CVRP/models.py:330-352 over the R = M T decode rows of an instance):
 * dO = dS PK, dA_h = dO_h V_h^T, dQ_h = dS_h K_h, dK_h = dS_h^T Q_h, dV_h = a_h^T dO_h, dPK = dS^T O. */
int elg_gemm_f32_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                         int transA, int transB, int n_outer, int n_inner, int64_t sA_outer, int64_t sA_inner,
                         int64_t sB_outer, int64_t sB_inner, int64_t sC_outer, int64_t sC_inner, float alpha, void* stream);


/* ---- attention encoder + decoder tables (CVRPModel.pre_forward: CVRP/CVRPModel.py:21-34 -> CVRP_Encoder.forward,
 * CVRP/models.py:211-229; EncoderLayer :249-269; multi_head_attention :455-503; AddAndInstanceNormalization :506-527;
 * FeedForward :550-561; CVRP_Decoder.set_kv :300-308.  TSP: TSP/models.py:134-194, 231-243), forward and backward,
 * entirely on hand-written gfx950 kernels (f32 MFMA).  Weight pointers are the reference's parameters where they lie
 * (nn.Linear layout (out, in), row-major; state_dict names in the comments).  embedding 128, 8 heads x 16. */
#define ELG_ENC_MAX_LAYERS 8
typedef struct elg_enc_layer {          /* encoder.layers.{l}.                                                     */
    const float *Wq, *Wk, *Wv;          /* Wq / Wk / Wv .weight (128,128), no bias                                 */
    const float *Wc, *bc;               /* multi_head_combine.weight (128,128), .bias (128)                        */
    const float *g1, *b1;               /* add_n_normalization_1.norm.weight / .bias (TSP: addAndNormalization1)   */
    const float *W1, *bf1;              /* feed_forward.W1.weight (ff,128), .bias (ff)   (TSP: feedForward)        */
    const float *W2, *bf2;              /* feed_forward.W2.weight (128,ff), .bias (128)                            */
    const float *g2, *b2;               /* add_n_normalization_2.norm.weight / .bias                               */
} elg_enc_layer;
typedef struct elg_enc_weights {
    const float *emb_depot_w, *emb_depot_b;   /* CVRP encoder.embedding_depot (128,2), (128); TSP: NULL            */
    const float *emb_w, *emb_b;               /* CVRP encoder.embedding_node (128,3); TSP encoder.embedding (128,2) */
    elg_enc_layer layer[ELG_ENC_MAX_LAYERS];
    const float *dec_Wq_first;                /* TSP decoder.Wq_first.weight (128,128); CVRP: NULL                 */
    const float *dec_Wq_last;                 /* decoder.Wq_last.weight: CVRP (128,129) (last column = load), TSP (128,128) */
    const float *dec_Wk, *dec_Wv;             /* decoder.Wk / Wv .weight (128,128)                                 */
    const float *dec_Wc, *dec_bc;             /* decoder.multi_head_combine.weight (128,128), .bias (128)          */
} elg_enc_weights;

typedef struct elg_encoder_args {
    int32_t problem;        /* ELG_PROBLEM_*                                                                       */
    int32_t B, N1;          /* instances, nodes per instance (CVRP: depot = node 0)                                */
    int32_t n_layers;       /* model_params.encoder_layer_num (<= ELG_ENC_MAX_LAYERS)                              */
    int32_t ff_hidden;      /* model_params.ff_hidden_dim (multiple of 64)                                         */
    int32_t save;           /* 1: keep every layer's activations in `ws` for elg_encoder_bwd (training)            */
    float eps;              /* InstanceNorm1d eps (1e-5)                                                           */
    int32_t precision;      /* 0: f32 (the parity mode).  1: bf16 throughput mode (BASELINE configs[1]), N1 <= 128 only (larger
                               instances compute in f32 whatever it says): every GEMM of the encoder, of the tables and of
                               their backward takes bf16 operands on v_mfma_f32_16x16x32_bf16 with f32 accumulation; bias,
                               residual, instance norm, softmax, the saved activations and the bias gradients stay f32 (the weight
                               gradients are bf16 products of the saved f32 activations with f32 accumulation).  The
                               backward must be given the forward's value.                                                */
    const float* xy;        /* (B,N1,2)                                                                            */
    const float* demand;    /* (B,N1) CVRP (depot entry unused); TSP: NULL                                         */
    elg_enc_weights W;
    float* enc;             /* (B,N1,128) out: encoded nodes                                                       */
    /* decoder tables, all-or-none (K == NULL: encoder only); same meaning as in elg_rollout_args               */
    float *K, *V, *PK;      /* (B,N1,128)                                                                          */
    float* pb;              /* (B,N1)                                                                              */
    float* Q1;              /* (B,N1,128)                                                                          */
    float* Q2;              /* (B,N1,128) TSP                                                                      */
    float* wl;              /* (128) CVRP: contiguous copy of Wq_last[:, 128]                                      */
    float* ws;              /* workspace, elg_encoder_ws_floats() floats, 16-byte aligned; with save = 1 it must stay
                               untouched until elg_encoder_bwd has run                                             */
    int64_t ws_floats;
} elg_encoder_args;
int64_t elg_encoder_ws_floats(int B, int N1, int n_layers, int ff_hidden, int save);
int elg_encoder_fwd(const elg_encoder_args* args, void* stream);

/* Backward of elg_encoder_fwd (what autograd records for CVRPModel.pre_forward in train.py:105-125): cotangents of
 * the outputs in, parameter gradients out.  N1 <= 128 (training sizes). */
typedef struct elg_encoder_bwd_args {
    elg_encoder_args fwd;   /* exactly the forward's arguments (same ws, save = 1)                                 */
    const float* g_enc;     /* (B,N1,128) d loss / d enc, or NULL                                                  */
    const float *gK, *gV, *gPK;   /* (B,N1,128) or NULL (= zero)                                                   */
    const float* gpb;       /* (B,N1) or NULL (needs gPK)                                                          */
    const float *gQ1, *gQ2; /* (B,N1,128) or NULL                                                                  */
    const float* gwl;       /* (128) or NULL                                                                       */
    elg_enc_weights G;      /* gradient destinations, same shapes as the weights; written through (float*), every
                               gradient is ACCUMULATED (+=): the caller zero-fills                                 */
    float* ws2;             /* scratch, elg_encoder_bwd_ws_floats() floats                                         */
    int64_t ws2_floats;
} elg_encoder_bwd_args;
int64_t elg_encoder_bwd_ws_floats(int B, int N1, int n_layers, int ff_hidden);
int elg_encoder_bwd(const elg_encoder_bwd_args* args, void* stream);

/* ---- local policy: parameters -> slot tables (ELG_LOC_* above).  local_policy_att's parameters (CVRP/models.py:8-36,
 * TSP/models.py:12-33; state_dict decoder.local_policies.0.* / decoder.local_policy_0.*): init_emb (32,nfeat) + (32),
 * cur_token_emb (32), Wq / Wk / Wv (32,32), multi_head_combine (32,32) + (32).  nfeat = 3 (CVRP) or 2 (TSP); n_slots =
 * local_size (+1 for the CVRP depot slot); positional = model_params['positional'] (sinusoid table added to the
 * slot embedding, models.py:142-143).  The backward WRITES the eight gradients (grads: same struct, float*). */
typedef struct elg_local_weights {
    const float *init_emb_w, *init_emb_b, *cur_token_emb, *Wq, *Wk, *Wv, *combine_w, *combine_b;
} elg_local_weights;
int elg_local_fold_fwd(const elg_local_weights* w, int nfeat, int n_slots, int positional, float* loc, void* stream);
int elg_local_fold_bwd(const elg_local_weights* w, int nfeat, int n_slots, int positional, const float* gloc,
                       const elg_local_weights* grads, void* stream);

/* utils.check_feasible (CVRP/utils.py:90-119; TSP/utils.py:72-78) for the M tours of ONE instance: pi (M rows of T node
 * ids, int64, row stride m_stride elements), demand (N) of the customers (NULL: TSP, nodes 0..N-1 exactly once).
 * flags (2 x int32, caller zeroes): flags[0] = 1 "Invalid tour", flags[1] = 1 "Used more than capacity" (the
 * reference's sequential fp32 scan, threshold 1 + 1e-4). */
int elg_check_feasible(const int64_t* pi, int64_t m_stride, const float* demand, int M, int T, int N, int32_t* flags,
                       void* stream);

/* After elg_rollout_fwd: stats[0] = max over tlen (the T of utils.rollout's outputs), stats[1] = 1 if a chosen
 * probability of a decoded step is exactly 0 (CVRPModel.py:67-68 then adds 1e-6 to that step).  stats: 2 x int32,
 * caller zeroes.  tlen (B,M), probs (B,Tcap,M) or NULL (stats[1] and zero_steps stay 0).  zero_steps (Tcap x int32, caller zeroes) or NULL: [t] = 1 for the
 * steps in which that happened. */
int elg_rollout_stats(const int32_t* tlen, const float* probs, int B, int M, int Tcap, int32_t* stats,
                      int32_t* zero_steps, void* stream);

/* ---- decoder backward over the rows saved by a training forward (elg_rollout_args.tr*, time-major r = t*M + m) ----
 * Autograd of the chosen-node probabilities (CVRP/models.py:322-423 under train.py:112-125) w.r.t. the decoder tables:
 *   w_r = gprob * pval * [first_decode_step <= t < tlen[b,m]]
 *   dl[r,n] = w_r (Csel_r [n == action_r] - PC[r,n])          d loss / d (pre-clip score), kept on chip
 *   dO = dl PK,  dPK += dl^T O,  dpb += sum_r dl,  rowDU[r,j] = dl[r, Slot[r,j]] * inv_ens
 *   glimpse attention backward (dK, dV) from dO, and the query-gather backward
 *   dQ1[n] += sum_{r: prev node = n} dQ_r,  dQ2[n] += sum_{r: first node = n} dQ_r (TSP),  dwl += sum_r load_r dQ_r (CVRP)
 * Every d* output is ACCUMULATED with atomics: the caller zero-fills.  N1 <= 128. */
typedef struct elg_decoder_bwd_args {
    int32_t problem, B, M, N1;
    int32_t T;                  /* decode steps covered (R = T*M rows per instance)                               */
    int32_t Tcap_actions;       /* time extent of `actions`                                                       */
    int32_t first_decode_step;  /* 2 (CVRP: depot, POMO start) or 1 (TSP)                                         */
    float inv_ens;              /* 1 / ensemble_size                                                              */
    int64_t Rcap;               /* rows per instance in the tr* buffers (>= R)                                    */
    const float* gprob;         /* (B,T,M) d loss / d prob                                                        */
    const float* pval;          /* (B,T,M) the chosen probabilities                                               */
    const int32_t* tlen;        /* (B,M)                                                                          */
    const int32_t* actions;     /* (B,M,Tcap_actions)                                                             */
    const float* trPC;          /* (B,Rcap,N1)   as saved by elg_rollout_fwd                                      */
    const float* trCsel;        /* (B,Rcap)                                                                       */
    const float* trQ;           /* (B,Rcap,128)                                                                   */
    const float* trO;           /* (B,Rcap,128)                                                                   */
    const float* trLoad;        /* (B,Rcap) CVRP                                                                  */
    const int32_t* trSlot;      /* (B,Rcap,48) or NULL (no local policy)                                          */
    const float* trA;           /* (B,8,Rcap,N1) glimpse weights, or NULL if trMask is given                      */
    const uint64_t* trMask;     /* (B,Rcap,W) mask words: the weights are recomputed from trQ, Kmat.  W = 2 for N1 <= 128;
                                 * for the rows of the streaming kernel W = mask_words = 4 / 8 / 16 (N1 <= 256 / 512 / 1024)  */
    const float* trLse;         /* (B,Rcap,8) with trMask: the forward's log2-sum-exp per head (NULL: recompute)  */
    const float *Kmat, *Vmat, *PK;          /* (B,N1,128) decoder tables                                          */
    float *dK, *dV, *dPK;       /* (B,N1,128) out (+=)                                                            */
    float* dpb;                 /* (B,N1) out (+=)                                                                */
    float* dQ1;                 /* (B,N1,128) out (+=)                                                            */
    float* dQ2;                 /* (B,N1,128) out (+=), TSP                                                       */
    float* dwl;                 /* (128) out (+=), CVRP                                                           */
    float* rowDU;               /* (B,R,48) out (written) for elg_local_bwd_rows, or NULL                         */
    float* dO;                  /* (B,R,128) scratch                                                              */
    int32_t* idx_prev;          /* (B,R) scratch                                                                  */
    int32_t* idx_first;         /* (B,R) scratch, TSP                                                             */
    float* rowW;                /* (B,R,4) scratch, 16-byte aligned                                               */
    const int32_t* T_dev;       /* NULL, or the device-resident step count (elg_rollout_stats' stats[0]) when the   *
                                 * host has not read it yet: `T` is then only an upper bound, the kernels use       *
                                 * min(T, T_dev[0]) (same value in elg_local_bwd_rows), scratch strides follow it    */
    int32_t gprob_T;            /* with T_dev: time extent of gprob / pval (>= T); ignored otherwise                */
    int32_t tables_frozen;      /* 1: the decoder tables carry no gradient (`training: only_local`, CVRPModel.py:78-131: they
                                 * are zeros): only the softmax / pointer pass runs (it produces rowDU for the local policy);
                                 * dK, dV, dQ1, dQ2, dwl are left untouched                                            */
    /* 128 < N1 <= 1024 (rows saved by the streaming rollout kernel; T_dev must be NULL): the glimpse weights and their
     * cotangents are formed by two tile kernels (f32 MFMA) into the scratch below, the remaining row contractions run as batched
     * f32 MFMA GEMMs (the reductions over the rows accumulate with f32 atomics: dK, dV, dPK must be zeroed, as for N1 <= 128);
     * trPC is overwritten with d s (in place). */
    int32_t mask_words;         /* W of trMask for N1 > 128 (ignored otherwise)                                        */
    int32_t mfma_mode;          /* N1 <= 128, mask-row mode: arithmetic of the glimpse backward's five products.  0: f32 MFMAs
                                 * (v_mfma_f32_16x16x4_f32, exact f32; 6.5e-7 of the largest gradient entry against float64; the query-gather
                                 * scatter is a one-hot product of exact 0 / 1 factors with the three bf16 terms of dq, f32 accumulation).
                                 * Split-bf16 (v_mfma_f32_16x16x32_bf16 on bf16 terms of the f32 operands, f32 accumulation):
                                 * 1 = 2 terms per operand (16 significand bits; 3e-5), 2 = 3-term score product q K^T +
                                 * 2-term linear products (7.5e-6); 3 = the backward of a bf16 rollout (elg_rollout_args.precision
                                 * = 1): the scores are recomputed as there -- bf16(q) . bf16(K), one term, scaled after the f32
                                 * sum -- so that exp2(s - lse) reproduces the forward's weights; linear products as in 1    */
    float* ws;                  /* scratch for N1 > 128 (NULL otherwise): k * elg_decoder_bwd_ws_floats(1, R, N1) floats,
                                 * 1 <= k <= B -- the batch is walked in chunks of k instances                          */
    int64_t ws_floats;
} elg_decoder_bwd_args;
int elg_decoder_bwd(const elg_decoder_bwd_args* args, void* stream);
/* Scratch of elg_decoder_bwd for N1 > 128: two (B,8,R,N1) buffers (glimpse weights / their cotangents) + the (B,R,128) query
 * cotangent rows; 0 for N1 <= 128. */
int64_t elg_decoder_bwd_ws_floats(int32_t B, int64_t R, int32_t N1);

#ifdef __cplusplus
}
#endif
#endif
