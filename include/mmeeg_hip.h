/*
 * mmeeg_hip.h — C ABI of libmmeeg_hip.so, the MI355X (gfx950) kernel library
 * behind multimodal_eeg_fmri_amd.
 *
 * The reference (bacon205/Multimodal_eeg_fmri) has no FFI / operator plug-in
 * interface: its only boundary is the Python nn.Module surface, whose leaf
 * arithmetic is torch.nn (ATen).  Each entry point below therefore cites the
 * reference call site(s) whose arithmetic it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Contract (all entry points)
 *   - plain pointers and sizes only; device pointers are BORROWED (no
 *     allocation, free or host sync inside the library; workspaces come in);
 *   - asynchronous on `stream`; safe to capture in a hipGraph;
 *   - return 0 on success, <0 on error (-1 bad argument, -2 launch failure,
 *     -3 unsupported shape); mm_last_error() gives the thread-local message;
 *   - `void*` tensors are bf16, `float*` fp32; channels-last layouts:
 *     1-D activations [B][T][C], tokens [M][D], volumes [B][D][H][W][C];
 *   - results are BIT-REPRODUCIBLE: no floating-point atomics anywhere.  Per-channel accumulators
 *     written by many workgroups (`stats`, `sums_out`, `dbias`, dgamma/dbeta scratch, `tapsum`, ...) are
 *     ACCUMULATOR WORKSPACES: the caller allocates and ZEROES 32 x n fp32-sized elements (written
 *     "[32][...]" below) and hands them to the consumer untouched; their content is opaque - 16 replicas
 *     of n 64-bit fixed-point sums (integer atomics are order-free; a workgroup adds rint(v * 2^k) into
 *     replica blockIdx % 16; k = 28 for activation statistics, 40 for gradient sums: csrc/common.h).
 *     mm_bn_finalize, the *_bwd_apply passes, mm_conv3d_l1_bwd and mm_transpose_add read them directly;
 *     everything else goes through mm_acc_reduce / mm_reduce_many (-> fp32).  Weight gradients use
 *     per-workgroup SLOTS (one writer per element) summed in slot order by mm_wgrad_scatter;
 *   - activation codes: 0 none, 1 GELU(erf), 2 ReLU, 3 tanh, 4 sigmoid;
 *   - dropout: keep iff hash(seed, element index) >= p * 2^32, scaled 1/(1-p);
 *     the backward entry points recompute the same mask from (p, seed).
 *     `seed_epoch` (nullable device word) is mixed into the seed on the device, so
 *     a launch recorded in a hipGraph draws a new mask on every replay.
 */
#ifndef MMEEG_HIP_H
#define MMEEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

const char* mm_last_error(void);
int mm_abi_version(void);

/* ---- layout packers --------------------------------------------------------
 * (B,C,T) fp32 -> (B,T,Cp) bf16, zero-padded channels.  Replaces the implicit
 * NCT layout of every nn.Conv1d input (enhanced_models_v4.py:129, 211-223). */
int mm_pack_nct_bf16(const float* x, void* y, int B, int C, int T, int Cp, hipStream_t stream);
/* inverse for the input gradient: (B,T,Cp) bf16 -> (B,C,T) fp32 */
int mm_unpack_ntc_f32(const void* g, float* dx, int B, int C, int T, int Cp, hipStream_t stream);
/* weight (Cout,Cin,k) fp32 -> forward image [Cout][k][Cinp] bf16 and (optional)
 * data-gradient image [Cinp][k flipped][Coutp] bf16 */
int mm_prep_conv_weight(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int k,
                        int Cinp, int Coutp, hipStream_t stream);

/* ---- 1-D implicit GEMM (bf16 MFMA, fp32 accumulate) -----------------------
 * Y[b,t,n] = epilogue( sum_{tap,c} X[b,t+tap-pad,c] * W[n,tap,c] )
 * epilogue: v = acc*scale[n] + shift[n]; stats[0][n]+=v, stats[1][n]+=v*v;
 *           out_pre = v; v = act(v); v *= dropout; v += residual; v += pe[t][n];
 *           max over t pairs if pool == 2; store fp32 and/or bf16.
 * backward fusion: with gradz != NULL, v *= act'(gradz[idx]) before the dropout mask, so
 * the data-gradient GEMM of a Linear directly yields d(pre-activation) of the layer below.
 * Replaces nn.Conv1d (+ folded eval BatchNorm1d + GELU + MaxPool1d)
 * (enhanced_models_v4.py:128-144, 210-234; crossmodal_v4_enhancements.py:822-877)
 * and, with taps == 1, every nn.Linear / MHA projection on the path
 * (enhanced_models_v4.py:71-81, 164; bridge_utils.py:34-66). */
int mm_conv1d_fwd(const void* x, const void* w, int B, int T, int Cin, int Cout, int taps, int pad,
                  const float* scale, const float* shift, int act, const float* residual,
                  const float* pe, int pool, float* stats, float* out_f32, void* out_bf16,
                  void* out_pre, float drop_p, uint32_t drop_seed, const uint32_t* seed_epoch,
                  const void* gradz, int gradz_act, hipStream_t stream);
/* mm_conv1d_fwd with the reduction split over input-channel slices (few output tiles, long reduction: the merged
 * 192-channel k = 7 convolution of EnhancedPowerEncoder over the ~6 000 STFT channels of config #5 is 96 tiles of 98
 * chunks on 256 CUs): nsplit x the workgroups, each slice's raw fp32 tile to ws, a second launch adds the slices in order
 * (same bits every run) and runs the epilogue.  mm_conv1d_fwd_splitk_plan says when it pays (nsplit_host = 1: call
 * mm_conv1d_fwd) and how many floats ws needs; k > 1 convolutions with Cin % 64 == 0 only. */
int mm_conv1d_fwd_splitk_plan(int B, int T, int Cin, int Cout, int taps, int* nsplit_host, int64_t* ws_floats_host,
                              hipStream_t stream);
int mm_conv1d_fwd_splitk(const void* x, const void* w, int B, int T, int Cin, int Cout, int taps, int pad,
                  const float* scale, const float* shift, int act, const float* residual,
                  const float* pe, int pool, float* stats, float* out_f32, void* out_bf16,
                  void* out_pre, float drop_p, uint32_t drop_seed, const uint32_t* seed_epoch,
                  const void* gradz, int gradz_act, float* ws, int nsplit, hipStream_t stream);
/* dW[n][c][tap] (fp32, strides sn/sc/stap in elements) += sum_{b,t} dY[b,t,n]*X[b,t+tap-pad,c];
 * optional dbias[n] += sum dY.  Replaces the weight/bias gradients autograd
 * derives for the layers above (loss.backward(), run_training_lite.py:486).
 * slot_mode must be 1 (0, fp32 atomics into replicas, is gone: -1): dw is a workspace of nrep >= mm_conv1d_wgrad_slots(...) slots of rep_stride floats;
 * workgroup row-chunk x stores (no atomics, no zeroing needed) its partial dW into slot x; the caller
 * sums the slots (mm_wgrad_scatter / mm_scatter_many with nrep = slots).  dbias (nullable) is an accumulator
 * workspace [32][Cout]. */
int mm_conv1d_wgrad(const void* dy, const void* x, float* dw, float* dbias, int B, int T, int Cin,
                    int Cout, int taps, int pad, int Cin_real, int64_t sn, int64_t sc, int64_t stap,
                    int nrep, int64_t rep_stride, int slot_mode, hipStream_t stream);
/* *slots_host (HOST int) = number of slots a slot-mode launch with these dimensions writes */
int mm_conv1d_wgrad_slots(int B, int T, int Cin, int Cout, int taps, int* slots_host, hipStream_t stream);
/* several Linear weight gradients (taps 1, slot mode) in one launch: desc_host = n x 64 bytes
 * {dy, x, workspace [nslots][Cout][Cin], dbias replicas (nullable)} pointers + int B, T, Cin, Cout, Cin_real,
 * nslots (= mm_conv1d_wgrad_many_slots), 0, 0.  Same arithmetic as n mm_conv1d_wgrad(..., slot_mode = 1) calls. */
int mm_conv1d_wgrad_many(const void* desc_host, int n, hipStream_t stream);
/* slot count of one problem of such a grouped launch (fewer, longer workgroups per problem than a stand-alone
 * mm_conv1d_wgrad: the group supplies the parallelism) */
int mm_conv1d_wgrad_many_slots(int B, int T, int Cin, int Cout, int* slots_host, hipStream_t stream);
/* dw[n][c][tap] += sum_slot ws[slot][n][tap][c] (nrep = slots, summed in slot order): the wgrad kernels write
 * channel-contiguous slots; this moves the sum to the parameter layout once. */
int mm_wgrad_scatter(const float* ws, float* dw, int Cout, int Cin, int taps, int Cinp, int nrep,
                     hipStream_t stream);
/* PositionalEncoding.forward as a stand-alone op (enhanced_models_v4.py:44-55 =
 * crossmodal_v4_enhancements.py:40-50): out[b][l][d] = dropout(x[b][l][d] + pe[l][d]) on the fp32
 * stream, x (B,L,D), pe (L,D) = rows of the sinusoid buffer; fp32 and/or bf16 output.  pe == NULL is
 * the op's backward: dx = dout * the same dropout mask (same p, seed).  D % 4 == 0.
 * (Inside the encoders the add rides in the conv-3 / fusion-conv epilogue: mm_bn_act_fwd `pe`.) */
int mm_add_pe(const float* x, const float* pe, float* out_f32, void* out_bf16, int B, int L, int D,
              float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t stream);

/* debugging: buf[idx] = 100 MHz wall clock, written in stream order (tools/ and bench --stamps) */
int mm_debug_stamp(void* buf, int idx, hipStream_t stream);
/* mm_prep_conv_weight for ndesc tensors in one launch per 64 descriptors; desc_host = HOST array
 * of {const float* w; void* w_fwd; void* w_dgrad /*nullable*/; int32 Cout, Cin, k, Cinp, Coutp, 0}
 * (48 bytes each), copied into the kernel arguments (capturable in a hipGraph) */
int mm_prep_many(const void* desc_host, int ndesc, hipStream_t stream);
/* the same, and the launch also zeroes `nzero` floats at `zero` (16-byte aligned, nzero % 4 == 0): a training step's
 * accumulator workspaces, which would otherwise be one more fill node in front of the first kernel */
int mm_prep_many_zero(const void* desc_host, int ndesc, float* zero, int64_t nzero, hipStream_t stream);
/* mm_wgrad_scatter for ndesc workspaces in one launch per 64 descriptors; desc_host = HOST array of
 * {const float* ws; float* dw; int32 Cout, Cin, taps, Cinp, nrep, cout_all} (40 bytes each).  A descriptor may take a BLOCK
 * of a wider workspace (the three branches of EnhancedPowerEncoder's merged convolution, mm_power_merge): taps = the
 * gradient tensor's taps | first workspace tap << 8 | workspace taps << 16 (upper bits 0 = the whole kernel), ws already
 * advanced to the block's first output channel, cout_all = the workspace's output channels (0 = Cout). */
int mm_scatter_many(const void* desc_host, int ndesc, hipStream_t stream);
/* ndesc independent reductions into parameter gradients in one launch per 64 descriptors; desc_host =
 * HOST array of {const void* src; float* dst; int64 K, nrep, rep_stride} (40 bytes each), copied into the
 * kernel arguments (capturable in a hipGraph).  nrep = 16: src = a gradient accumulator workspace (8-byte
 * aligned; element e of it is at byte offset 8 e; rep_stride in 64-bit elements), dst[k] += its sum in fp32;
 * nrep = 1: src = compact fp32 vector, dst[k] += src[k]. */
int mm_reduce_many(const void* desc_host, int ndesc, hipStream_t stream);
/* mm_scatter_many(scatter_desc_host, nscatter) and mm_reduce_many(reduce_desc_host, nreduce) in ONE launch (per 48
 * descriptors of each kind): the two are independent, and a graph node costs ~5 us on the stream that flushes */
int mm_flush_many(const void* scatter_desc_host, int nscatter, const void* reduce_desc_host, int nreduce,
                  hipStream_t stream);
/* dst[k] += fp32(sum over the 16 replicas of acc[rep * rep_stride + k]),  k < K: gradient accumulator workspace
 * (64-bit elements: offset e is byte offset 8 e) -> fp32 */
int mm_acc_reduce(const float* acc, float* dst, int K, int64_t rep_stride, hipStream_t stream);
/* fp32: dst[k] += sum_rep src[rep * rep_stride + k],  k < K  (replicas summed in order) */
int mm_reduce_replicas(const float* src, float* dst, int K, int nrep, int64_t rep_stride, hipStream_t stream);

/* ---- BatchNorm / activation / pool ----------------------------------------
 * mode 0 (train): stats{sum,sumsq}/count -> out4 = {scale, shift, mean, rstd},
 * running stats updated (momentum, unbiased var);  mode 1 (eval): fold running
 * stats (+conv bias).  nn.BatchNorm1d/3d (enhanced_models_v4.py:130,135,141).
 * batches_tracked (int64 device scalar, nullable) is incremented in train mode, as
 * nn.BatchNorm's num_batches_tracked buffer is. */
int mm_bn_finalize(const float* stats, const float* gamma, const float* beta, float* run_mean,
                   float* run_var, const float* conv_bias, float* out4, int N, float count,
                   float momentum, float eps, int mode, void* batches_tracked, hipStream_t stream);
/* The train-mode finalize as the PROLOGUE of its consumer (one launch and one graph node less per BatchNorm layer): the
 * `*_fin` forms of the apply passes take, instead of scale / shift (or out4), a HOST pointer to this descriptor (read at
 * call time, like the tables of mm_prep_many); every workgroup forms scale / shift of the N <= 256 channels from the
 * statistics workspace with mm_bn_finalize's arithmetic (the same device function: same bits), workgroup 0 also writes
 * out4 [4][N] = {scale, shift, mean, rstd} (the backward reads it) and updates the running statistics / batches_tracked.
 * GELU only (the encoders' BatchNorm layers). */
typedef struct {
    const float* stats;      /* accumulator workspace [32][2][N] {sum, sumsq} */
    const float* gamma; const float* beta;
    float* run_mean; float* run_var;
    float* out4;             /* [4][N], written */
    void* batches_tracked;   /* int64 device scalar, nullable */
    float count, momentum, eps;
    int reserved;
} mm_bn_fin_t;
int mm_bn_act_fwd_fin(const float* y, const void* bn_fin_host, const float* pe, void* out_bf16, float* out_f32,
                      int R, int S, int N, int act, int pool, int drop_first, float drop_p, uint32_t seed,
                      float drop2_p, uint32_t seed2, const uint32_t* seed_epoch, hipStream_t stream);
int mm_bn_act_fwd_ln_fin(const float* y, const void* bn_fin_host, const float* pe, float* out_f32, int R, int S,
                         int act, float drop_p, uint32_t seed, float drop2_p, uint32_t seed2,
                         const uint32_t* seed_epoch, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         void* ln_out_bf16, float* ln_stat, hipStream_t stream);
int mm_pool3d_bn_act_fwd_fin(const void* y, const void* bn_fin_host, void* out_bf16, void* ysel, void* arg, int B,
                             int D, int H, int W, int N, int act, float drop_p, uint32_t seed,
                             const uint32_t* seed_epoch, hipStream_t stream);
int mm_conv3d_l1_fwd_fin(const float* x, const void* wimg, const float* bias, const void* bn_fin_host, void* out,
                         int B, int D, int H, int W, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                         hipStream_t stream);
/* y fp32 [R][S][N] -> act(y*scale+shift) [-> maxpool2 over S] [-> dropout] [+pe[s][n]]
 * (BN -> GELU -> MaxPool1d -> Dropout -> PositionalEncoding add,
 *  enhanced_models_v4.py:130-143, 49-54) */
int mm_bn_act_fwd(const float* y, const float* scale, const float* shift, const float* pe,
                  void* out_bf16, float* out_f32, int R, int S, int N, int act, int pool,
                  int drop_first, float drop_p, uint32_t seed, float drop2_p, uint32_t seed2,
                  const uint32_t* seed_epoch, hipStream_t stream);
/* the same pass for the last conv block of EnhancedERPEncoder (N = 128, no pooling) with the first
 * TemporalTransformerBlock's norm1 (enhanced_models_v4.py:97-99, LayerNorm(128)) of every finished row fused in:
 * out_f32 = the transformer input, ln_out_bf16 = norm1(out) (the QKV projection's operand), ln_stat [R*S][2] =
 * mean, rstd (nullable; the backward's).  Replaces mm_bn_act_fwd + mm_layernorm_fwd. */
int mm_bn_act_fwd_ln(const float* y, const float* scale, const float* shift, const float* pe, float* out_f32,
                     int R, int S, int act, float drop_p, uint32_t seed, float drop2_p, uint32_t seed2,
                     const uint32_t* seed_epoch, const float* ln_gamma, const float* ln_beta, float ln_eps,
                     void* ln_out_bf16, float* ln_stat, hipStream_t stream);
/* drop2 = the PositionalEncoding dropout applied AFTER the table add (:55)  * mm_bn_act_bwd_apply: sums = the accumulator workspace [32][2][N] as written by mm_bn_act_bwd_reduce
 * (sums_nrep = 32: the kernel adds the replicas up itself) or a compact fp32 [2][N] (sums_nrep = 1). */
int mm_bn_act_bwd_reduce(const float* y, const float* out4, const void* dout_bf16,
                         const float* dout_f32, float* sums_out, int R, int S, int N, int act,
                         int pool, int drop_first, float drop_p, uint32_t seed, float drop2_p,
                         uint32_t seed2, const uint32_t* seed_epoch, hipStream_t stream);
int mm_bn_act_bwd_apply(const float* y, const float* out4, const void* dout_bf16,
                        const float* dout_f32, const float* sums, void* dy, float* dy_f32, int R,
                        int S, int N, int act, int pool, int drop_first, float drop_p, uint32_t seed,
                        float drop2_p, uint32_t seed2, const uint32_t* seed_epoch, int train, int sums_nrep, hipStream_t stream);

/* ---- LayerNorm (nn.LayerNorm, enhanced_models_v4.py:80-81; bridge_utils.py:36,42,62) */
int mm_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* out_bf16,
                     float* out_f32, float* stat, int M, int D, float eps, hipStream_t stream);
/* dgb_repl = zeroed accumulator workspace [32][2][D]: {dgamma, dbeta} partial sums;
 * dx_bf16 (optional) = bf16(dx * dropout_mask(drop_p, seed)): the masked GEMM operand of the
 * residual branch feeding this LayerNorm's input */
int mm_layernorm_bwd(const void* dy_bf16, const float* dy_f32, const float* x, const float* stat,
                     const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgb_repl,
                     int M, int D, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                     hipStream_t stream);
/* TemporalTransformerBlock backward (enhanced_models_v4.py:86-105, autograd of norm1/norm2 feeding
 * self_attn.in_proj / linear1): data gradient of that Linear and the LayerNorm-128 backward in one launch.
 * dy (M, K) bf16; w = the Linear's data-gradient weight image (mm_prep_conv_weight's w_dgrad: 128 x K);
 * x / stat / gamma / dres / dx / dx_bf16 / dgb_repl / dropout arguments exactly as mm_layernorm_bwd. */
int mm_linear_dgrad_ln_bwd(const void* dy, const void* w, int M, int K, const float* x, const float* stat,
                           const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgb_repl,
                           float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t stream);

/* EnhancedERPEncoder.conv_layers backward (enhanced_models_v4.py:99-105, autograd of Conv1d -> BatchNorm1d -> GELU
 * [-> MaxPool1d(2)] -> Dropout stacked twice): the data gradient of one conv block (dy (B, T, Cin) bf16, w_dgrad = that
 * block's data-gradient weight image, dx_bf16 (B, T, Cout) bf16 = d(out) of the block BELOW) with the BatchNorm-backward
 * reduce pass of the block below as its epilogue.  y_below (B, T * pool, Cout) fp32, out4_below, sums_below (zeroed
 * [32][2][Cout] workspace) and the activation / pool / dropout arguments are those of
 * mm_bn_act_bwd_reduce(y_below, out4_below, dx_bf16, NULL, sums_below, B, T * pool, Cout, ...), whose launch this replaces
 * (drop2_p = 0).  Needs taps > 1 or Cout <= 64 (the 64 x 64 tile). */
int mm_conv1d_dgrad_bn_reduce(const void* dy, const void* w_dgrad, int B, int T, int Cin, int Cout, int taps, int pad,
                              void* dx_bf16, const float* y_below, const float* out4_below, float* sums_below, int act,
                              int pool, int drop_first, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                              hipStream_t stream);

/* mm_linear_fwd_ln (the last Linear of a TemporalTransformerBlock with the NEXT block's norm1 fused,
 * enhanced_models_v4.py:97-107) followed, inside the launch, by that next block's self_attn in_proj on the LayerNorm rows:
 * out2_bf16 (M, n2) = ln_out_bf16 @ w2^T + bias2, w2 = in_proj's forward weight image (n2 rows of 128, n2 % 128 == 0).
 * Bit-identical to mm_conv1d_fwd(ln_out_bf16, w2, 1, M, 128, n2, 1, 0, NULL, bias2, ..., out_bf16 = out2_bf16). */
int mm_linear_fwd_ln_gemm2(const void* x, const void* w, int M, int K, const float* bias, const float* residual,
                           float* out_f32, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                           const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out_bf16, float* ln_stat,
                           const void* w2, const float* bias2, int n2, void* out2_bf16, hipStream_t stream);
/* ... and with an epilogue on the second GEMM: the attention out-projection (+ residual, dropout, norm2 fused) followed by
 * the first FFN Linear on norm2's rows - out2_bf16 (M, n2) = dropout(act2(ln_out @ w2^T + bias2)), pre2_bf16 (nullable) =
 * its bf16 pre-activation (enhanced_models_v4.py:99-105).  Bit-identical to mm_conv1d_fwd(ln_out_bf16, w2, 1, M, 128, n2,
 * 1, 0, NULL, bias2, act2, ..., out_bf16 = out2_bf16, out_pre = pre2_bf16, drop2_p, seed2, seed_epoch, ...). */
int mm_linear_fwd_ln_gemm2_act(const void* x, const void* w, int M, int K, const float* bias, const float* residual,
                               float* out_f32, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                               const float* ln_gamma, const float* ln_beta, float ln_eps, void* ln_out_bf16,
                               float* ln_stat, const void* w2, const float* bias2, int n2, void* out2_bf16,
                               void* pre2_bf16, int act2, float drop2_p, uint32_t seed2, hipStream_t stream);
/* mm_linear_dgrad_ln_bwd of norm2 / linear1 with the attention out-projection's data gradient as a second GEMM in the
 * same launch (TemporalTransformerBlock backward, enhanced_models_v4.py:99-103: x1 = x0 + dropout(out_proj(attn)),
 * norm2(x1)): do_bf16 (M, 128) = dx_bf16 @ w2, w2 = out_proj's data-gradient weight image (128 x 128), dx_bf16 = the
 * rows masked with out_proj's dropout (drop_p, seed).  Bit-identical to mm_conv1d_fwd(dx_bf16, w2, 1, M, 128, 128, 1, 0,
 * ..., out_bf16 = do_bf16) after mm_linear_dgrad_ln_bwd.  dres_rows_per_sample > 0: dres holds ONE row per that many
 * consecutive rows (mm_pooled_head_bwd_rows), 0: a row per row. */
int mm_linear_dgrad_ln_bwd_gemm2(const void* dy, const void* w, int M, int K, const float* x, const float* stat,
                                 const float* gamma, const float* dres, float* dx, void* dx_bf16, float* dgb_repl,
                                 float drop_p, uint32_t seed, const uint32_t* seed_epoch, const void* w2, void* do_bf16,
                                 int dres_rows_per_sample, hipStream_t stream);
/* mm_linear_dgrad_ln_bwd for the FIRST transformer block of EnhancedERPEncoder, whose LayerNorm input is the last conv
 * block's output (enhanced_models_v4.py:143-147: conv_layers[-1] -> pos_encoder -> transformer_layers[0].norm1): the rows dx
 * (fp32) are that block's d(out), so its BatchNorm-backward reduce pass (mm_bn_act_bwd_reduce(y_below, out4_below, NULL, dx,
 * sums_below, 1, M, 128, act, 1, 1, bn_drop_p, bn_seed, bn_drop2_p, bn_seed2, ...)) rides in the same launch.
 * y_below (M, 128) fp32; sums_below = zeroed [32][2][128] workspace; drop2 = the PositionalEncoding dropout. */
int mm_linear_dgrad_ln_bwd_bn_reduce(const void* dy, const void* w, int M, int K, const float* x, const float* stat,
                                     const float* gamma, const float* dres, float* dx, float* dgb_repl,
                                     const uint32_t* seed_epoch, const float* y_below, const float* out4_below,
                                     float* sums_below, int act, float bn_drop_p, uint32_t bn_seed, float bn_drop2_p,
                                     uint32_t bn_seed2, hipStream_t stream);

/* The two BatchNorm-backward passes for a block whose d(out) is the backward of a mean over its S positions (the voxel
 * encoder's last conv block under AdaptiveAvgPool3d(1) -> Linear): dout_rows (R, N) fp32 = ONE row per sample, every
 * position's d(out) = dout_rows[r] * scale (scale = 1 / S).  Same results as mm_bn_act_bwd_reduce / _apply on the
 * broadcast (R, S, N) tensor, which is never written or read (pool 1, no second dropout). */
int mm_bn_act_bwd_reduce_bcast(const float* y, const float* out4, const float* dout_rows, float scale, float* sums_out,
                               int R, int S, int N, int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                               hipStream_t stream);
int mm_bn_act_bwd_apply_bcast(const float* y, const float* out4, const float* dout_rows, float scale, const float* sums,
                              void* dy, int R, int S, int N, int act, float drop_p, uint32_t seed,
                              const uint32_t* seed_epoch, int train, int sums_nrep, hipStream_t stream);

/* mm_pooled_head_bwd (the backward of x.mean(dim=1) -> output_proj, enhanced_models_v4.py:161-167) without the fp32
 * (B, L, D) token gradients: rows_out (B, D) = the one row every token of a sample receives; dx_bf16 (B, L, D) = that row
 * under the consumer's dropout mask, as mm_pooled_head_bwd writes it. */
int mm_pooled_head_bwd_rows(const float* dout, const void* z_pre_bf16, const float* W, void* dz_bf16, float* rows_out,
                            void* dx_bf16, int B, int L, int D, int N, int act, float drop_p, uint32_t seed,
                            float emit_drop_p, uint32_t emit_seed, const uint32_t* seed_epoch, hipStream_t stream);

/* ---- multi-head self-attention, head_dim 32 (nn.MultiheadAttention,
 * enhanced_models_v4.py:71-73, 99).  qkv [B][L][3E] bf16 -> out [B][L][E] bf16,
 * lse [B][H][L] fp32.  drop_p = attention-probability dropout (train mode).  The
 * head-averaged weights the reference computes and discards (:99) are not produced.
 * attn_mask (nullable): the `mask` of TemporalTransformerBlock.forward(x, mask)
 * (enhanced_models_v4.py:88-98 -> self_attn(..., attn_mask=mask)) as an ADDITIVE fp32 (L, L)
 * matrix shared by every batch element and head (a boolean mask = 0 / -inf), or - attn_mask_per_head != 0 -
 * nn.MultiheadAttention's 3-D form: (B*H, L, L), matrix b*H + h for head h of batch element b.
 * A fully masked row yields NaN, as in PyTorch. */
int mm_attn_fwd(const void* qkv, void* out, float* lse, int B, int L, int H, int head_dim,
                float scale, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                const float* attn_mask, int attn_mask_per_head, hipStream_t stream);
int mm_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                float* delta_ws, int B, int L, int H, int head_dim, float scale, float drop_p,
                uint32_t seed, const uint32_t* seed_epoch, const float* attn_mask, int attn_mask_per_head,
                hipStream_t stream);

/* ---- small reductions / elementwise --------------------------------------- */
int mm_colsum(const void* a_bf16, const float* a_f32, float* out, int M, int N, hipStream_t stream);
/* AdaptiveAvgPool1d(1)+Flatten (enhanced_models_v4.py:162-163) on fp32 tokens */
int mm_meanpool_fwd(const float* x, float* out_f32, void* out_bf16, int B, int L, int D, hipStream_t stream);
int mm_meanpool_bwd(const float* g, float* dx, int B, int L, int D, hipStream_t stream);
int mm_cast_bf16(const float* x, void* y, int64_t n, hipStream_t stream);
/* A training step's inputs into the static buffers a captured step reads, ONE launch: mm_pack_nct_bf16(eeg ->
 * eeg_packed_bf16) [+ an fp32 copy of the EEG batch into eeg_copy, nullable] + an fp32 copy of the fMRI batch
 * (fmri_n floats, a multiple of 4, 16-byte aligned).  The captured step then starts at the first convolution. */
int mm_stage_inputs(const float* eeg, void* eeg_packed_bf16, float* eeg_copy, int B, int C, int T, int Cp,
                    float* fmri_dst, const float* fmri_src, int64_t fmri_n, hipStream_t stream);
int mm_cast_f32(const void* x, float* y, int64_t n, hipStream_t stream);
/* out = bf16( g * dropout_mask * act'(z) ) */
int mm_act_bwd(const float* g_f32, const void* g_bf16, const void* z, void* out, int64_t n, int act,
               float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t stream);

/* ---- 3-D voxel convolution, k=3 pad=1 (north-star extension: the reference has
 * no volume code, SURVEY.md section 0; semantics = torch.nn.Conv3d / BatchNorm3d /
 * MaxPool3d(2)).  Volumes are channels-last [B][D][H][W][C] bf16. */
/* (B,1,D,H,W) fp32 -> [B][D][H][W][Cp] bf16, channel 0 = voxel value, rest 0 */
int mm_pack_volume_bf16(const float* x, void* y, int64_t nvox, int Cp, hipStream_t stream);
/* Y = X (*) W + shift; W image [Cout][27][Cin] (mm_prep_conv_weight with k=27);
 * optional accumulator workspace stats[32][2][Cout] (sum, sumsq of the fp32 results) for training BatchNorm; fp32 and/or
 * bf16 output.  Generic path: LDS-staged (TD+2)x10x10 halo block -> 27 tap-shifted A fragments -> bf16 MFMA.
 * Cin = 32, Cout = 64 with bf16 output only and >= 64 tiles of 4x8x8 voxels (layer 2 of the voxel encoder)
 * runs the weight-resident persistent kernel of csrc/conv3d_wres.hip. */
int mm_conv3d_fwd(const void* x, const void* w, int B, int D, int H, int W, int Cin, int Cout,
                  const float* shift, float* stats, float* out_f32, void* out_bf16, hipStream_t stream);
int mm_conv3d_wgrad(const void* dy, const void* x, float* dw, float* dbias, int B, int D, int H, int W,
                    int Cin, int Cout, int Cin_real, int64_t sn, int64_t sc, int64_t stap, int nrep,
                    int64_t rep_stride, int slot_mode, hipStream_t stream);
/* slot_mode as in mm_conv1d_wgrad; *slots_host (HOST int) = slots a slot-mode launch writes */
int mm_conv3d_wgrad_slots(int B, int D, int H, int W, int Cin, int Cout, int* slots_host, hipStream_t stream);
/* y bf16 [B][D][H][W][N] (the convolution's pre-BatchNorm output) -> act(BN(y)) -> MaxPool3d(2) -> dropout ->
 * bf16 [B][D/2][H/2][W/2][N].  N % 8 == 0.  Training also keeps, per pooled element, the winner's pre-BN
 * value (ysel bf16) and its index in the 2x2x2 window (arg, one byte: 4 d + 2 h + w); both null in eval.
 * The reduction of the BatchNorm gradient sums then reads only pooled data (gradients vanish off the
 * winners) and the apply pass (dy bf16, full volume) takes the argmax from `arg`; its `sums` is the reduce pass's
 * accumulator workspace (sums_nrep = 32) or a compact fp32 [2][N] (sums_nrep = 1), as for mm_bn_act_bwd_apply. */
int mm_pool3d_bn_act_fwd(const void* y, const float* out4, void* out_bf16, void* ysel, void* arg, int B, int D,
                         int H, int W, int N, int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                         hipStream_t stream);
int mm_pool3d_bn_act_bwd_reduce(const void* ysel, const float* out4, const void* dout_bf16, float* sums_out,
                                int B, int D, int H, int W, int N, int act, float drop_p, uint32_t seed,
                                const uint32_t* seed_epoch, hipStream_t stream);
int mm_pool3d_bn_act_bwd_apply(const void* y, const void* arg, const float* out4, const void* dout_bf16,
                               const float* sums, void* dy, int B, int D, int H, int W, int N, int act,
                               float drop_p, uint32_t seed, const uint32_t* seed_epoch, int train, int sums_nrep,
                               hipStream_t stream);

/* Fused first voxel layer Conv3d(1->32,k3,p1)+BatchNorm3d+GELU+MaxPool3d(2)[+Dropout]
 * on fp32 [B][D][H][W] volumes; the 32x larger pre-BN tensor is recomputed, never
 * stored.  mode 0: stats[2][32] += {sum, sumsq} of conv+bias; mode 1: forward
 * (out bf16 [B][D/2][H/2][W/2][32]); mode 2: stats += {sum dz, sum dz*xhat};
 * mode 3: dw_tapmajor[27][32] += x^T dy, dbias += sum dy.  wimg = bf16 [32][32]
 * (n, tap) from mm_prep_conv_weight(w as (32,27,1)). */
int mm_conv3d_l1(int mode, const float* x, const void* wimg, const float* bias, const float* out4,
                 const void* dout, const float* sums, float* stats, void* out, float* dw_tapmajor,
                 float* dbias, int B, int D, int H, int W, int train, float drop_p, uint32_t seed,
                 const uint32_t* seed_epoch, hipStream_t stream);
/* mode 1 (the forward) with one more output: arg u8 [B][D/2][H/2][W/2][32] = which member of each 2x2x2 pooling
 * window won, j = (dd << 2) | (hh << 1) | ww (the convention of mm_pool3d_bn_act_fwd's `arg`).  The training path
 * recomputes the winners in its backward and never stores them; this entry point lets a caller inspect the routing
 * (the parity tests evaluate the fp32 oracle with the HIP path's routing: an arg-max flip between two near-equal
 * window members is then not counted as a gradient error). */
int mm_conv3d_l1_fwd_winners(const float* x, const void* wimg, const float* bias, const float* out4, void* out,
                             void* arg, int B, int D, int H, int W, int train, float drop_p, uint32_t seed,
                             const uint32_t* seed_epoch, hipStream_t stream);
/* Training forward of the same layer without a statistics pass over the convolution: the Gram matrix of the im2col
 * matrix, G[t][t'] = sum over output voxels of xcol[v][t] * xcol[v][t'] for the 27 taps plus a column of ones (bf16-
 * rounded, zero-padded volume - what the convolution sees), holds everything the layer needs from the input:
 *   sum_v y_n = w_n . S + M b_n,  sum_v y_n^2 = w_n^T G w_n + 2 b_n w_n . S + M b_n^2   (S[t] = G[t][27], M = G[27][27])
 * and, in the backward, A3[t][n] = sum_v xcol[v][t] xhat[v][n] = rstd_n ((G w_n)[t] + (b_n - mean_n) S[t]).
 * mm_conv3d_l1_gram: gram = ZEROED accumulator workspace [32][32][32] (activation-statistics scale; G is symmetric and
 * only its upper triangle t <= t' is accumulated; kept for mm_conv3d_l1_bwd), stats = ZEROED accumulator workspace
 * [32][2][32] receiving {sum y, sum y^2} of conv + bias - every workgroup adds the sums of its own voxels (they are
 * linear in G) - i.e. the input of mm_bn_finalize, as after any other convolution.
 * Training backward in ONE recompute pass (replaces modes 2 + 3): BatchNorm's backward is linear in the two sums
 * S1 = sum dz, S2 = sum dz * xhat, so
 *   dW = scale * (A1 - (S1/M) * S - (S2/M) * A3),  A1 = x^T dz, S and A3 from the Gram workspace.
 * Zeroed accumulator workspaces: sums_out [32][2][32] (also the BatchNorm parameter gradients: dbeta = S1,
 * dgamma = S2), a1 [32][27][32].  dw (PyTorch layout [32][1][3][3][3]) and dbias are ADDED to (dbias only when
 * train == 0; it is identically 0 otherwise).  gram may be NULL when train == 0 (frozen BatchNorm: the two correction
 * terms vanish).  mm_conv3d_l1_tapsum (S alone, by row sums) remains for callers that want it. */
int mm_conv3d_l1_gram(const float* x, const void* wimg, const float* bias, float* gram, float* stats, int B, int D,
                      int H, int W, hipStream_t stream);
int mm_conv3d_l1_tapsum(const float* x, float* tapsum, int B, int D, int H, int W, hipStream_t stream);
int mm_conv3d_l1_bwd(const float* x, const void* wimg, const float* bias, const float* out4, const void* dout,
                     float* sums_out, float* a1, const float* gram, float* dw, float* dbias, int B, int D,
                     int H, int W, int train, float drop_p, uint32_t seed, const uint32_t* seed_epoch,
                     hipStream_t stream);
/* dst[c][r] += fp32(sum over replicas of src[rep][r][c]); src = gradient accumulator workspace [32][R][C], nrep = 16 */
int mm_transpose_add(const float* src, float* dst, int R, int C, int nrep, hipStream_t stream);

/* ---- small fp32 row kernels (projection bridge, tabular fMRI/conn MLPs) ------
 * y = dropout(act((x W^T + b) * scale + shift)) (scale/shift = folded eval
 * BatchNorm1d, may be NULL), optional pre-activation copy.  nn.Linear on
 * (B, K) feature rows: bridge_utils.py:34-45,60-66; fmri_utils.py:26-35,44-53;
 * crossmodal_v4_enhancements.py:695-723. */
int mm_small_linear_fwd(const float* x, const float* W, const float* bias, const float* scale,
                        const float* shift, float* y, float* pre, int B, int K, int N, int act,
                        float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t stream);
/* dx = dy W ; dW += dy^T x ; db += colsum(dy)   (dy already through act') */
int mm_small_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db,
                        int B, int K, int N, hipStream_t stream);
int mm_act_f32(const float* z, float* y, int64_t n, int act, float drop_p, uint32_t seed, const uint32_t* seed_epoch, hipStream_t stream);
int mm_act_bwd_f32(const float* g, const float* z, float* out, int64_t n, int act, float drop_p,
                   uint32_t seed, const uint32_t* seed_epoch, hipStream_t stream);
/* sum_b x and sum_b x^2 (BatchNorm1d over (B, N)) into replica 0 of the ZEROED statistics accumulator workspace
 * stats [32][2][N] (the input of mm_bn_finalize) */
int mm_colstats(const float* x, float* stats, int B, int N, hipStream_t stream);
/* Both projection heads of the contrastive bridge (bridge_utils.py:34-45 eeg_proj / fmri_proj:
 * Linear(K -> N) -> LayerNorm -> GELU -> Dropout) followed by F.normalize, one launch each way.
 * x_* fp32 [B][K_*]; W_* [N][K_*]; z packed [B][2N] = [ze | zf]; nrm [2][B].  z1 / hn / stat
 * ([2][B][N], [2][B][N], [2][B][2]; all three or none) are what the backward needs.  The
 * backward ADDS parameter gradients (any of them may be null; one writer per element, rows summed in order) and
 * writes dx. */
int mm_proj_heads_fwd(const float* x_e, const float* W_e, const float* b_e, const float* g_e, const float* be_e,
                      int K_e, const float* x_f, const float* W_f, const float* b_f, const float* g_f,
                      const float* be_f, int K_f, float* z1, float* hn, float* stat, float* z, float* nrm, int B,
                      int N, float eps, float drop_p, uint32_t seed_e, uint32_t seed_f,
                      const uint32_t* seed_epoch, hipStream_t stream);
int mm_proj_heads_bwd(const float* dz, const float* z, const float* nrm, const float* hn, const float* z1,
                      const float* stat, const float* x_e, const float* W_e, const float* g_e, int K_e,
                      const float* x_f, const float* W_f, const float* g_f, int K_f, float* dx_e, float* dW_e,
                      float* db_e, float* dg_e, float* dbe_e, float* dx_f, float* dW_f, float* db_f, float* dg_f,
                      float* dbe_f, int B, int N, float drop_p, uint32_t seed_e, uint32_t seed_f,
                      const uint32_t* seed_epoch, hipStream_t stream);
/* F.normalize(h, dim=1): z = h / max(||h||, 1e-12)  (extension a-X2) */
int mm_l2norm_fwd(const float* h, float* z, float* nrm, int B, int N, int ldz, hipStream_t stream);
int mm_l2norm_bwd(const float* dz, const float* z, const float* nrm, float* dh, int B, int N, int ldz,
                  hipStream_t stream);
/* batch-pairwise cosine-similarity matrix + symmetric InfoNCE, bit-reproducible (no float atomics).
 * Embeddings are packed rows [ze (N) | zf (N)]: z_all [Bg][2N] = the all-gathered global batch, this rank's
 * pairs at rows [row0, row0+B).  C[r][j] = ze_r . zf_j; e->f = row softmax of exp(logit_scale) C, f->e =
 * column softmax.  scal4 = {mean loss, top1 e->f, top1 f->e, d loss / d logit_scale} over the rank's own rows
 * (plain stores: no zeroing needed); dz_local [B][2N] (nullable, plain stores) = d (SUM over ranks of their
 * losses) / d z_local - exactly the block a reduce-scatter-sum of every rank's d loss / d z_all would
 * deliver, so the step needs no reduce-scatter (every rank evaluates all Bg rows of the gathered batch:
 * 2 Bg^2 N MACs).  ws = scratch of mm_clip_loss_ws_floats(B, Bg) floats (log-sum-exp of every row and
 * column + per-row scalars), no initialisation needed.  N % 4 == 0.  Two launches on `stream`.
 * Extension a-X2: the reference trains a CE classifier (_test_bridge.py:858). */
int mm_clip_loss_own_rows(const float* z_all, const float* logit_scale, float* scal4, float* dz_local, float* ws,
                          int B, int Bg, int N, int row0, hipStream_t stream);
int mm_clip_loss_ws_floats(int B, int Bg, int* floats_host, hipStream_t stream);

/* ---- EnhancedPowerEncoder: its three Conv1d(C -> 64, k = 3 | 5 | 7) + BatchNorm1d(64) branches
 * (enhanced_models_v4.py:210-234, forward :258-266: torch.cat of the three) as ONE Conv1d(C -> 192, k = 7, p = 3) +
 * BatchNorm1d(192).  desc_host = HOST pointer to this descriptor (read at call time, like the tables of mm_prep_many):
 *   mode 0  parts -> merged: W[o][c][t] = w_i[o % 64][c][t - (7 - k_i) / 2] inside branch i = o / 64's taps, else 0;
 *           bias / gamma / beta / running mean / running var concatenated
 *   mode 1  merged running mean / var -> the parts' (after a train-mode forward); batches_tracked[i] += 1 (nullable)
 *   mode 2  gradients: the parts' sinks (w, b, gamma, beta; null = frozen) += their slices of the merged gradients
 *           (W, B, Gamma, Beta; null = absent)
 *   mode 3  as mode 0, but W receives the merged weight as mm_conv1d_fwd's bf16 image [192][7][cinp] (what
 *           mm_prep_conv_weight makes of the fp32 merged weight, which is then never written) */
typedef struct {
    float* w[3]; float* b[3]; float* gamma[3]; float* beta[3]; float* run_mean[3]; float* run_var[3];
    void* batches_tracked[3];            /* int64 device scalars */
    float* W; float* B; float* Gamma; float* Beta; float* Run_mean; float* Run_var;
    int cin; int k[3]; int cinp; int reserved;
} mm_power_merge_t;
int mm_power_merge(const void* desc_host, int mode, hipStream_t stream);

/* ---- fused tails of the small models (forward) ------------------------------ */
/* fMRIFusionNet weighted concat (fmri_utils.py:93-96) */
int mm_softmax2_concat(const float* a, const float* c, const float* pa, const float* pc, float* out, int B,
                       int Ha, int Hc, hipStream_t stream);
/* LearnedFusionModule combine (enhanced_models_v4.py:468-484); dyn = gate_net output [B][M] */
int mm_learned_fusion(const float* f0, const float* f1, const float* f2, const float* dyn,
                      const float* logits, const float* temperature, float* fused, float* weights, int B,
                      int H, int M, hipStream_t stream);
/* nn.MultiheadAttention core for ONE query token and K <= 4 key/value tokens per sample: the
 * modality-level cross attention of the V4 classifiers (crossmodal_v4_enhancements.py:366-372 K = 3,
 * :448-456 K = 2).  p_j = in_proj(token_j) fp32 [B][3E] = [q | k | v]; the query is token 0's q.
 * backward == 0: ctx [B][E] (+ head-averaged attw [B][K], nullable);  backward == 1: dp_j [B][3E] from
 * dctx.  Attention-probability dropout by counter hash, recomputed in backward. */
int mm_attn_1xk(const float* p0, const float* p1, const float* p2, const float* p3, int K, const float* dctx,
                float* ctx, float* attw, float* dp0, float* dp1, float* dp2, float* dp3, int B, int E,
                int nhead, float drop_p, uint32_t seed, const uint32_t* seed_epoch, int backward,
                hipStream_t stream);
int mm_add_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t stream);
/* bridge cross-attention core, 1 query x 2 keys (bridge_utils.py:75-82) */
int mm_attn_1x2(const float* proj_e, const float* proj_f, float* ctx, float* attw, int B, int E, int nhead,
                hipStream_t stream);
/* trainable form of the 1x2 cross-attention core (with attention-probability dropout):
 * backward == 0: ctx/attw from proj_e/proj_f;  backward == 1: dproj_e/dproj_f [B][3E] from dctx */
int mm_attn_1x2_train(const float* proj_e, const float* proj_f, const float* dctx, float* ctx, float* attw,
                      float* dproj_e, float* dproj_f, int B, int E, int nhead, float drop_p, uint32_t seed,
                      const uint32_t* seed_epoch, int backward, hipStream_t stream);
/* backward of mm_learned_fusion: df_m, ddyn, and (ADDED; rows summed in a fixed order) dlogits[M], dtemp[1] */
int mm_learned_fusion_bwd(const float* f0, const float* f1, const float* f2, const float* dyn,
                          const float* logits, const float* temperature, const float* dfused, float* df0,
                          float* df1, float* df2, float* ddyn, float* dlogits, float* dtemp, int B, int H,
                          int M, hipStream_t stream);
int mm_softmax2_concat_bwd(const float* dout, const float* a, const float* c, const float* pa,
                           const float* pc, float* da, float* dc, float* dpa, float* dpc, int B, int Ha,
                           int Hc, hipStream_t stream);
/* nn.CrossEntropyLoss(weight=class_weight) (_test_bridge.py:858; run_fmri_v11.py): loss_out[0] += loss */
int mm_weighted_ce(const float* logits, const void* target_i64, const float* class_weight, float* loss_out,
                   float* dlogits, int B, int C, hipStream_t stream);
/* FocalLoss (CrossModal_EEG_scr.ipynb cell 20; FlexibleTrainer(use_focal_loss=True), cell 23):
 * fl_b = alpha (1 - exp(-ce_b))^gamma ce_b.  loss_out[0] += scale * sum_b fl_b (zeroed by the caller),
 * per_sample[b] = fl_b (nullable), dlogits = d fl_b / d logits without the reduction's factor (nullable) */
int mm_focal_loss(const float* logits, const void* target_i64, float* loss_out, float* per_sample,
                  float* dlogits, int B, int C, float alpha, float gamma, float scale, hipStream_t stream);
/* HybridFusionModule gate + mix + conn boost (crossmodal_v4_enhancements.py:787-797) */
int mm_gate2_mix(const float* g, const float* erp, const float* pw, const float* conn, float* comb,
                 float* gate, int B, int H, float boost, hipStream_t stream);
int mm_gate2_mix_bwd(const float* dcomb, const float* g, const float* erp, const float* pw, float* derp,
                     float* dpw, float* dconn, float* dg, int B, int H, float boost, hipStream_t stream);
/* LabelSmoothingCrossEntropy (crossmodal_v4_enhancements.py:665-677): loss_out[0] += loss (zeroed by
 * the caller), dlogits = d loss / d logits; target is int64 class indices */
int mm_smoothed_ce(const float* logits, const void* target_i64, float* loss_out, float* dlogits, int B,
                   int C, float smoothing, hipStream_t stream);
/* Multi-scale STFT power front-end (extension a-X3; semantics = torch.stft(center=True,
 * pad_mode="reflect", periodic Hann) then |.|^2): x (B,C,T) fp32 -> channels-last
 * out[b][frame][ch_off + c*F + f], F = nfft/2+1, frames = T/hop+1, row width ch_total
 * (several scales write side by side into one activation tensor). */
int mm_stft_power(const float* x, void* out_bf16, float* out_f32, int B, int C, int T, int nfft, int hop,
                  int ch_off, int ch_total, hipStream_t stream);
/* normalize_modality (run_training_lite.py:48-51; applied per sample to the power features at :162):
 * out[b] = bf16((x[b] - mean(x[b])) / (std(x[b]) + eps)), population std (numpy ddof = 0), over all rows x ch_valid elements of
 * sample b; x / out [B][rows][ch_total] channels-last, channels >= ch_valid are written as zeros.
 * ws (nullable; MM_ZSCORE_WS_DOUBLES doubles per sample, 8-byte aligned): with it, big unpadded samples (ch_valid ==
 * ch_total, >= 65 536 elements) are dealt out over 32 workgroups each - per-chunk sums in double, added in chunk order by
 * every workgroup of the second launch (same bits every run); without it one workgroup per sample does it all. */
#define MM_ZSCORE_WS_DOUBLES 64
int mm_sample_zscore_bf16(const float* x, void* out_bf16, double* ws, int B, int rows, int ch_valid, int ch_total,
                          float eps, hipStream_t stream);
/* Backward of the STFT power front-end and of its z-score (gradient w.r.t. the RAW EEG: saliency / integrated
 * gradients on the config-#5 model, the protocol of bridge_utils.py:158-229 / eeg_xai_analysis.py on an end-to-end
 * path).  mm_sample_zscore_bwd: x = the fp32 spectra the forward z-scored, g_bf16 = gradient w.r.t. the z-scored
 * bf16 tensor (same [B][rows][ch_total] layout) -> dx fp32 (padding channels 0).  mm_stft_power_bwd: g_power fp32
 * [B][frames][ch_total] -> dx fp32 [B][C][T] is ADDED to (one launch per scale; zero it first).  Gathers, no atomics. */
int mm_sample_zscore_bwd(const float* x, const void* g_bf16, float* dx, int B, int rows, int ch_valid, int ch_total,
                         float eps, hipStream_t stream);
int mm_stft_power_bwd(const float* x, const float* g_power, float* dx, int B, int C, int T, int nfft, int hop,
                      int ch_off, int ch_total, hipStream_t stream);
int mm_mul_f32(const float* a, const float* b, float* out, int64_t n, hipStream_t stream);
/* Encoder tail (enhanced_models_v4.py:161-167, 186-191: mean over time -> output_proj = Linear -> GELU ->
 * Dropout).  mm_linear_fwd_meanpool is the last transformer block's linear2 (+ dropout + residual, fp32 rows
 * out_f32 (M, 128)) that also accumulates the mean over each group of rows_per_group rows (one EEG epoch's
 * tokens) into the ZEROED pool_out, an accumulator of (M / rows_per_group) x 128 64-bit elements (ONE replica:
 * 2 x that many floats).  mm_pooled_head_fwd applies the head to the pooled rows - fp32 `pooled`, or that
 * accumulator as `pooled_acc` (exactly one non-null) - in fp32 (W is the nn.Linear weight (N, D)); z_pre_bf16 / pooled_bf16 (nullable) are what the
 * backward and the weight-gradient GEMM need.  mm_pooled_head_bwd: dz = dout * dropout mask * act'(z) (bf16 copy
 * dz_bf16 for the weight gradient), d pooled = dz W, and dx[b][l][:] = d pooled / L for every token; dx_bf16
 * (nullable) = the same rows times the consumer's dropout mask (emit_drop_p, emit_seed; element index as in
 * mm_act_bwd). */
int mm_linear_fwd_meanpool(const void* x, const void* w, int M, int K, const float* bias, const float* residual,
                           float* out_f32, float drop_p, uint32_t seed, const uint32_t* seed_epoch, float* pool_out,
                           int rows_per_group, hipStream_t stream);
/* TemporalTransformerBlock forward (enhanced_models_v4.py:86-105): a sub-layer's closing Linear (attention
 * out_proj or linear2; width 128) + dropout + residual, fp32 rows out_f32 (M, 128), with the NEXT sub-layer's
 * pre-norm fused: ln_out_bf16 = LayerNorm(out rows; gamma, beta, eps) and ln_stat [M][2] = mean, rstd (nullable),
 * i.e. mm_conv1d_fwd followed by mm_layernorm_fwd without re-reading the rows. */
int mm_linear_fwd_ln(const void* x, const void* w, int M, int K, const float* bias, const float* residual,
                     float* out_f32, float drop_p, uint32_t seed, const uint32_t* seed_epoch, const float* ln_gamma,
                     const float* ln_beta, float ln_eps, void* ln_out_bf16, float* ln_stat, hipStream_t stream);
int mm_pooled_head_fwd(const float* pooled, const float* pooled_acc, const float* W, const float* bias, float* out, void* z_pre_bf16,
                       void* pooled_bf16, int B, int D, int N, int act, float drop_p, uint32_t seed,
                       const uint32_t* seed_epoch, hipStream_t stream);
int mm_pooled_head_bwd(const float* dout, const void* z_pre_bf16, const float* W, void* dz_bf16, float* dx,
                       void* dx_bf16, int B, int L, int D, int N, int act, float drop_p, uint32_t seed,
                       float emit_drop_p, uint32_t emit_seed, const uint32_t* seed_epoch, hipStream_t stream);
/* drop_path / DropPath (crossmodal_v4_enhancements.py:639-658): out[b][...] = x[b][...] * keep_b / (1 - p),
 * keep_b from the counter hash of (seed, b); calling it on the upstream gradient is the backward */
int mm_drop_path(const float* x, float* out, int64_t B, int64_t inner, float drop_p, uint32_t seed,
                 const uint32_t* seed_epoch, hipStream_t stream);
/* AdaptiveAvgPool1d(1) of the Lite encoders on bf16 [R][S][N] */
int mm_meanpool_bf16(const void* x, float* out, int R, int S, int N, hipStream_t stream);

/* ---- optimizer: clip_grad_norm_(max_norm) + AdamW.step() on one flat bucket
 * (run_training_lite.py:487-488; _test_bridge.py:784-786).  state (device, MM_OPT_STATE_FLOATS
 * = 8 + 1024 floats): [0] step count, [1] sum of squared grads of the last step, [2] lr,
 * [3] clip coefficient of the last step, [4] grad norm of the last step, [8..) per-block
 * partial sums written by mm_sumsq and added in a fixed order by mm_adamw_clip (no float
 * atomics: ranks holding the same all-reduced gradient stay bit-identical).
 * zero_grad != 0: g is cleared after use (the next step's zero_grad()); seed_epoch (nullable): the
 * device dropout-epoch word is incremented for the next step. */
#define MM_OPT_STATE_FLOATS 1032
int mm_sumsq(const float* g, float* state, int64_t n, hipStream_t stream);
int mm_adamw_clip(float* p, float* g, float* m, float* v, float* state, int64_t n, float beta1,
                  float beta2, float eps, float weight_decay, float max_norm, float grad_scale,
                  int zero_grad, uint32_t* seed_epoch, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMEEG_HIP_H */
