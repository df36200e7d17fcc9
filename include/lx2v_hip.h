/*
 * lx2v_hip.h -- C ABI of liblx2v_hip.so: gfx950 (MI355X / CDNA4) kernels for the Wan DiT
 * transformer-block forward.  Plain pointers and sizes only; every pointer is DEVICE memory
 * unless noted.  Each entry point enqueues work on `stream` (a hipStream_t passed as void*;
 * NULL = the legacy default stream), never synchronises, never allocates (unless stated), and
 * returns 0 on success or a negative LX2V_E* code (lx2v_last_error() gives the text).
 *
 * bf16 tensors are raw 16-bit storage (uint16_t); scale/shift/gate/bias/norm-weight vectors
 * and the RoPE table are fp32.  "ld*" arguments are row strides in ELEMENTS.
 *
 * The reference has no C ABI on this path: its boundary is a set of Python plugin classes.
 * Each function below names the reference interface it stands behind (paths relative to the
 * reference repository root).
 */
#ifndef LX2V_HIP_H
#define LX2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LX2V_OK 0
#define LX2V_EINVAL (-1)   /* shape / alignment / null-pointer precondition violated */
#define LX2V_EUNSUP (-2)   /* valid request outside what the kernels are built for */
#define LX2V_EHIP (-3)     /* a HIP runtime call failed (launch error) */

/* ABI version of this header (bumped when a signature changes or an entry point is added). */
int lx2v_abi_version(void);
/* Thread-local text of the last non-zero return. Host pointer, valid until the next call. */
const char* lx2v_last_error(void);

/* ---- kernel-selection switches -------------------------------------------------------------
 * Named integer options, initialised once from the environment (LX2V_ATTN_KERNEL, LX2V_ATTN_QB,
 * LX2V_ATTN_TAIL, LX2V_ATTN_ROUND_SYNC, LX2V_ATTN_VARIANT, LX2V_GEMM_KERNEL)
 * and changeable at run time; launches read them, never the environment.  Names: "attn_kernel"
 * (16 | 4 | 8), "attn_qb" (0 | 4 | 5), "attn_tail", "attn_round_sync", "attn_variant" (0 | 1),
 * "gemm_kernel" (0 default | 1 one barrier per K-tile | 2 ping-pong), "attn_fp8_kernel" (4 | 8).
 * Every setting of these computes the same function; they exist for A/B timing and for tests.
 * One option is an arithmetic choice instead: "attn_fp8_p" (LX2V_ATTN_FP8_P=linear), 0 (default) =
 * the fp8 attention takes P = 2^x by v_exp_f32 and rounds it to e4m3, 1 = it takes the e4m3 code
 * round(8 x + 56) directly, i.e. 2^x with a linear mantissa, and its row sums over the same codes
 * (no exps: ~11 % faster; cosine against the fp32 softmax 0.9983 instead of 0.9986).  Not to be
 * changed while another thread launches.  Stands where the reference selects a backend by registry name
 * (lightx2v/utils/registry_factory.py:1-56): these choose among kernels behind ONE name. */
int lx2v_set_option(const char* name, int value);
int lx2v_get_option(const char* name, int* value);

/* ---- LayerNorm (+ optional affine) (+ optional AdaLN modulate), one pass over HBM ----------
 * out[r,:] = (LN(x[r,:]) * ln_w + ln_b) * (1 + scale) + shift       (fp32 statistics)
 * ln_w/ln_b (both or neither) and scale/shift (both or neither) may be NULL.
 * Replaces LNWeight.apply  lightx2v/common/ops/norm/layer_norm_weight.py:157-169
 * fused with modulate      lightx2v/models/networks/wan/infer/transformer_infer.py:12-13
 * (and fuse_scale_shift_kernel  lightx2v/models/networks/wan/infer/triton_ops.py:120-229).
 * dim % 8 == 0, dim <= 8192. */
int lx2v_ln_modulate_bf16(const uint16_t* x, int64_t ldx, const float* ln_w, const float* ln_b,
                          const float* scale, const float* shift, uint16_t* out, int64_t ldo,
                          int64_t rows, int dim, float eps, void* stream);

/* Same with a per-token part of the modulation: scale_rows / shift_rows are bf16 [rows, dim] (row
 * stride ld_rows) and are added to the scale / shift vectors (both required) before use.
 * The reference reaches this by broadcasting (modulation [1,6,D] + embed0 [S,6,D]) when the timestep is
 * a per-token vector (Wan2.2 ti2v)
 *                 lightx2v/models/networks/wan/infer/transformer_infer.py:130-137, 157
 *                 lightx2v/models/schedulers/wan/scheduler.py:388-392. */
int lx2v_ln_modulate_rows_bf16(const uint16_t* x, int64_t ldx, const float* ln_w, const float* ln_b,
                               const float* scale, const float* shift, const uint16_t* scale_rows,
                               const uint16_t* shift_rows, int64_t ld_rows, uint16_t* out, int64_t ldo,
                               int64_t rows, int dim, float eps, void* stream);

/* The same LayerNorm + modulation (scale_rows / shift_rows may be NULL: vectors only) whose result
 * leaves as OCP e4m3 rows `out` (row stride ldo bytes) with one fp32 scale per row, row_scale[r] =
 * max|row r| / 448: the bytes and scales lx2v_quant_fp8_rowwise gives for the bf16 result, without
 * that result's trip through memory.  Replaces, for a quantised Linear fed by a LayerNorm, the
 * reference's separate activation quantisation
 *                 lightx2v/common/ops/mm/mm_weight.py:492-501 (act_quant_fp8_perchannel_sym_*). */
int lx2v_ln_modulate_quant_fp8(const uint16_t* x, int64_t ldx, const float* ln_w, const float* ln_b,
                               const float* scale, const float* shift, const uint16_t* scale_rows,
                               const uint16_t* shift_rows, int64_t ld_rows, uint8_t* out, int64_t ldo, float* row_scale,
                               int64_t rows, int dim, float eps, void* stream);

/* ---- RMSNorm over the whole row (+ optional per-head interleaved-pair RoPE) ----------------
 * y = x * rsqrt(mean(x^2) + eps) * w ; then for each head of `head_dim` columns and each pair
 * i: (y[2i], y[2i+1]) *= (cos[r,i] + j sin[r,i]).  rope_cos_sin is fp32 [rows, head_dim]
 * laid out cos(head_dim/2) || sin(head_dim/2) per row (the reference's flashinfer layout,
 * lightx2v/models/schedulers/wan/scheduler.py:90-95), or NULL for no rotation.
 * `do_norm`: 0 = no normalisation (rotation only), 1 = statistics over the whole row with
 * w[dim] (Wan: all heads jointly), 2 = statistics per head with w[head_dim] shared by every head
 * (HunyuanVideo-1.5: the norm is applied to the [L, H, d] view; head_dim a power of two 8..512).
 * `w` may be NULL (no weight).  In-place allowed (out == x).
 * Replaces RMSWeightSgl.apply  lightx2v/common/ops/norm/rms_norm_weight.py:131-146
 * (called on [S, D] at wan/infer/transformer_infer.py:164-165 and on [L, H, d] at
 * hunyuan_video/infer/transformer_infer.py:171-172, 204-205)
 * and apply_wan_rope_with_torch / _flashinfer
 *                              lightx2v/models/networks/wan/infer/utils.py:12-28,64-87,
 * apply_hunyuan_rope_with_torch / _flashinfer
 *                              lightx2v/models/networks/hunyuan_video/infer/transformer_infer.py:41-95. */
int lx2v_rmsnorm_rope_bf16(const uint16_t* x, int64_t ldx, const float* w, int do_norm,
                           const float* rope_cos_sin, int head_dim, uint16_t* out, int64_t ldo,
                           int64_t rows, int dim, float eps, void* stream);

/* ---- x += y * gate (gate fp32 [dim] or NULL => x += y), in place on x ----------------------
 * Replaces transformer_infer.py:209-213, 279, 312-316. */
int lx2v_gate_residual_bf16(uint16_t* x, int64_t ldx, const uint16_t* y, int64_t ldy,
                            const float* gate, int64_t rows, int dim, void* stream);

/* x += y * (gate + gate_rows[r, :]) with a per-token part gate_rows (bf16 [rows, dim], row stride
 * ld_rows): the per-token form of the same lines (see lx2v_ln_modulate_rows_bf16). */
int lx2v_gate_residual_rows_bf16(uint16_t* x, int64_t ldx, const uint16_t* y, int64_t ldy,
                                 const float* gate, const uint16_t* gate_rows, int64_t ld_rows,
                                 int64_t rows, int dim, void* stream);

/* ---- the residual stream kept in fp32 --------------------------------------------------------
 * The reference's SENSITIVE_LAYER_DTYPE=FP32 mode holds x in fp32 between the Linears: the three
 * residual sums upcast (`x.to(fp32) + y.to(fp32) * gate`), the norms read fp32 and their modulated
 * result is cast back to the compute dtype for the next Linear
 *                 lightx2v/models/networks/wan/infer/transformer_infer.py:86-90, 150-160, 209-213, 289-299, 312-316
 *                 lightx2v/utils/envs.py:39-43.
 * lx2v_ln_modulate_f32in: lx2v_ln_modulate_rows_bf16 with fp32 rows x (scale_rows / shift_rows may be
 * NULL: vector modulation only); bf16 out.  lx2v_gate_residual_f32: x (fp32) += y (bf16) * (gate +
 * gate_rows[r, :]), gate / gate_rows optional as in the bf16 entry points. */
int lx2v_ln_modulate_f32in(const float* x, int64_t ldx, const float* ln_w, const float* ln_b,
                           const float* scale, const float* shift, const uint16_t* scale_rows,
                           const uint16_t* shift_rows, int64_t ld_rows, uint16_t* out, int64_t ldo,
                           int64_t rows, int dim, float eps, void* stream);
int lx2v_gate_residual_f32(float* x, int64_t ldx, const uint16_t* y, int64_t ldy, const float* gate,
                           const uint16_t* gate_rows, int64_t ld_rows, int64_t rows, int dim, void* stream);

/* ---- GELU (tanh approximation), elementwise -------------------------------------------------
 * Replaces torch.nn.functional.gelu(approximate="tanh")  transformer_infer.py:305. */
int lx2v_gelu_tanh_bf16(const uint16_t* x, uint16_t* out, int64_t n, void* stream);

/* ---- exact (erf) GELU, bf16, out may alias x; n % 8 == 0 -----------------------------------
 * Replaces torch.nn.functional.gelu(approximate="none") of the CLIP image-token MLP
 *                 lightx2v/models/networks/wan/infer/pre_infer.py:106. */
int lx2v_gelu_erf_bf16(const uint16_t* x, uint16_t* out, int64_t n, void* stream);

/* ---- Linear: C = epilogue(A @ W^T + bias) on MFMA ------------------------------------------
 * A [M,K] (row stride lda), W [N,K] (the checkpoint's [out,in] layout, row stride ldw),
 * bias fp32 [N] or NULL, C [M,N] (row stride ldc), fp32 accumulation.
 *   epilogue LX2V_EPI_NONE      : C = acc + bias
 *            LX2V_EPI_GELU_TANH : C = gelu_tanh(acc + bias)
 *            LX2V_EPI_RESIDUAL  : C = resid + (acc + bias) * gate   (gate fp32 [N] or NULL => 1;
 *                                  resid [M,N] row stride ldr; resid may alias C)
 * A's columns may be stored as segments (`kseg` != NULL): `seg` columns each, segment s of row m
 * starting at A + s*stride + m*lda; from K column `split` on (0 = no second region) the segments are
 * `seg2` columns wide and segment t of that region starts at A + base2 + t*stride2 + m*lda2.  Pass
 * kseg = NULL for a plain matrix.  This lets the output projection read the sequence-parallel
 * attention result ([source rank][S/N][heads*d] per head chunk, chunks of two widths) without a
 * gather copy.  The struct is read before the call returns.
 * Requires K % 64 == 0, N % 8 == 0, 16-byte aligned rows.
 * Replaces MMWeight.apply (torch.addmm)  lightx2v/common/ops/mm/mm_weight.py:163-170, with the
 * residual/GELU steps of transformer_infer.py:209-213,279,305,312-316 fused as epilogues. */
typedef struct lx2v_kseg {
  int64_t seg, stride;          /* region 1: columns per segment, elements between segment starts */
  int64_t split;                /* first K column of region 2; 0 = one region */
  int64_t seg2, stride2, base2; /* region 2: segment width, stride, element offset of its first segment */
  int64_t lda2;                 /* region 2: row stride in elements */
} lx2v_kseg;
#define LX2V_EPI_NONE 0
#define LX2V_EPI_GELU_TANH 1
#define LX2V_EPI_RESIDUAL 2
int lx2v_gemm_bf16(const uint16_t* A, int64_t lda, const lx2v_kseg* kseg,
                   const uint16_t* W, int64_t ldw, const float* bias,
                   uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                   const uint16_t* resid, int64_t ldr, const float* gate, void* stream);

/* The same with a caller-owned workspace (no counterpart in the reference).  Output tiles are computed in rounds of
 * one 256x256 tile per CU; when the last round would fill at most a quarter of the CUs (5920 tiles on 256 CUs:
 * 23 rounds + 32 tiles) and K >= 4096, those tiles are split along K over all CUs instead: raw fp32 partial sums go
 * to the workspace, a last small launch adds them in a fixed order and applies the epilogue.  The tail tiles'
 * outputs then differ from lx2v_gemm_bf16's by fp32 summation order (deterministic from call to call); every other
 * output is bit-identical.  lx2v_gemm_bf16_workspace_bytes: what a call of that shape can use (0: it would not
 * split); a NULL, misaligned (16 bytes) or smaller workspace only switches the split off.  Calls that share a
 * workspace must be ordered (one stream).  Not used with a segmented A. */
int64_t lx2v_gemm_bf16_workspace_bytes(int64_t M, int64_t N, int64_t K);
int lx2v_gemm_bf16_ws(const uint16_t* A, int64_t lda, const lx2v_kseg* kseg,
                      const uint16_t* W, int64_t ldw, const float* bias,
                      uint16_t* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int epilogue,
                      const uint16_t* resid, int64_t ldr, const float* gate,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- Attention forward: out = softmax(q k^T * scale) v, non-causal, no mask, batch 1 --------
 * q [sq, heads, head_dim], k/v [skv, heads, head_dim], out [sq, heads*head_dim]; token strides
 * ld* in elements, heads are contiguous blocks of head_dim inside a token row.  head_dim must
 * be 128.  Online softmax with fp32 statistics and fp32 accumulation on MFMA.
 * If rope_cos_sin != NULL (fp32 [max(sq,skv), 128], cos||sin per row) the interleaved-pair
 * rotation of lightx2v/models/networks/wan/infer/utils.py:12-28 is applied to q and k as they
 * are loaded (requires sq == skv); pass NULL when q/k are already rotated.
 * Replaces AttnWeightTemplate.apply: TorchSDPAWeight lightx2v/common/ops/attn/torch_sdpa.py:14-39,
 * FlashAttn2Weight/FlashAttn3Weight lightx2v/common/ops/attn/flash_attn.py:29-96. */
int lx2v_attn_fwd_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                       const uint16_t* v, int64_t ldv, uint16_t* out, int64_t ldo,
                       int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                       const float* rope_cos_sin, void* stream);

/* The same call with room for partial results.  A launch of FEW heads over a long sequence (a head chunk of a
 * sequence-parallel rank) is only a few rounds of one workgroup per CU long, and its last, partly filled round is a
 * large share of it; with a workspace the launcher may split the KEYS of that round's query rows into ranges
 * (shorter workgroups fill the round), leaving unnormalised partial outputs and (max, sum) pairs in `workspace`,
 * and merge them in a second small kernel.  Same result up to the fp32 summation order of the merged rows.
 * lx2v_attn_bf16_workspace_bytes gives the most a call of that shape uses (0: no split would be planned);
 * `workspace` 16-byte aligned; a smaller or NULL workspace only limits the plan.  The library allocates nothing.
 * No counterpart in the reference (flash-attn's num_splits heuristic lives inside the wheel). */
int64_t lx2v_attn_bf16_workspace_bytes(int64_t sq, int64_t skv, int heads);
int lx2v_attn_fwd_bf16_ws(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                          const uint16_t* v, int64_t ldv, uint16_t* out, int64_t ldo,
                          int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                          const float* rope_cos_sin, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- Ulysses send-side pack: q,k,v [rows, heads*head_dim] -> [world][rows][3][heads/world*head_dim]
 * so ONE all-to-all moves q, k and v and the receiver sees a uniform token stride.
 * Replaces the three transpose().contiguous() packs of all2all_seq2head
 * lightx2v/common/ops/attn/utils/all2all.py:29-33. */
int lx2v_ulysses_pack_qkv_bf16(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                               const uint16_t* v, int64_t ldv, uint16_t* out, int64_t rows,
                               int dim, int world, void* stream);

/* ==== Denoise-step glue (around the blocks; all small) ====================================== */

/* ---- latents -> rows of the patch-embedding GEMM ------------------------------------------
 * latents fp32 [C, F, H, W] -> out bf16 [F*(H/2)*(W/2), 4*C] (row stride ldo >= 4*C; columns
 * past 4*C are not written), column (c, ph, pw): a conv3d with kernel = stride = (1,2,2) is then
 * lx2v_gemm_bf16 against weight.reshape(D, 4*C), both zero-padded to a K the GEMM accepts.
 * round_latents != 0 also overwrites `latents` with their bf16-rounded values (the reference
 * casts its stored latents to bf16 in BaseScheduler.step_pre, lightx2v/models/schedulers/scheduler.py:16-19).
 * Replaces weights.patch_embedding.apply + flatten/transpose
 *                 lightx2v/models/networks/wan/infer/pre_infer.py:56,63-64
 * (Conv3dWeight.apply  lightx2v/common/ops/conv/conv3d.py). */
int lx2v_patchify_bf16(float* latents, uint16_t* out, int64_t ldo, int channels, int frames,
                       int height, int width, int round_latents, void* stream);

/* ---- head output -> latent-shaped prediction ----------------------------------------------
 * y bf16 [S, 4*C] (column (ph, pw, c), row stride ldy) -> out fp32 [C, F, H, W].
 * Replaces WanPostInfer.unpatchify + .float()
 *                 lightx2v/models/networks/wan/infer/post_infer.py:17-31. */
int lx2v_unpatchify_f32(const uint16_t* y, int64_t ldy, float* out, int channels, int frames,
                        int height, int width, void* stream);

/* ---- channel planes <-> token rows (patch size (1,1,1): HunyuanVideo-1.5) --------------------
 * src fp32 [channels, tokens] -> out bf16 [tokens, kpad] (row stride ldo; columns channels..kpad-1
 * zeroed): a conv3d with kernel = stride = (1,1,1) is then lx2v_gemm_bf16 against
 * weight.reshape(D, channels) zero-padded to kpad.
 * Replaces weights.img_in.apply + flatten(2).transpose(1, 2)
 *                 lightx2v/models/networks/hunyuan_video/infer/pre_infer.py:104-109. */
int lx2v_channels_to_rows_bf16(const float* src, uint16_t* out, int64_t ldo, int channels,
                               int64_t tokens, int kpad, void* stream);

/* y bf16 [tokens, channels] (row stride ldy) -> out fp32 [channels, tokens].
 * Replaces HunyuanVideo15PostInfer.unpatchify (patch (1,1,1)) + .float()
 *                 lightx2v/models/networks/hunyuan_video/infer/post_infer.py:21-39. */
int lx2v_rows_to_channels_f32(const uint16_t* y, int64_t ldy, float* out, int channels,
                              int64_t tokens, void* stream);

/* ---- out bf16 [dim] = sum_l weights[l] x[l, :] / sum_l weights[l] (fp32 accumulation) --------
 * x bf16 [rows, dim] (row stride ldx), weights fp32 [rows] (a 0/1 mask gives the masked mean).
 * Replaces (txt * mask).sum(1) / mask.sum(1)
 *                 lightx2v/models/networks/hunyuan_video/infer/pre_infer.py:141-143. */
int lx2v_weighted_mean_rows_bf16(const uint16_t* x, int64_t ldx, const float* weights, int rows,
                                 int dim, uint16_t* out, void* stream);

/* ---- out[r, :] = x[r, :] + vec (bf16, one rounding); out may alias x ------------------------
 * Replaces txt + cond_type_embedding(...)
 *                 lightx2v/models/networks/hunyuan_video/infer/pre_infer.py:154-158. */
int lx2v_add_rowvec_bf16(const uint16_t* x, int64_t ldx, const uint16_t* vec, uint16_t* out,
                         int64_t ldo, int64_t rows, int dim, void* stream);

/* ---- sinusoidal timestep embedding: out bf16 [dim] = cos(t*w_i) (dim/2) || sin(t*w_i) -------
 * w_i = 10000^(-i/(dim/2)).  `t` is a HOST scalar.
 * Replaces sinusoidal_embedding_1d  lightx2v/models/networks/wan/infer/utils.py:205-215. */
int lx2v_timestep_embedding_bf16(float t, int dim, uint16_t* out, void* stream);

/* ---- SiLU, bf16, out may alias x.  Replaces torch.nn.functional.silu at pre_infer.py:79,81. */
int lx2v_silu_bf16(const uint16_t* x, uint16_t* out, int64_t n, void* stream);

/* ---- out = sum_{k<terms} coef[k] * x[k], fp32, 1 <= terms <= 4 -----------------------------
 * `x` (HOST array of device pointers) and `coef` (HOST array) are read before the call returns;
 * out may alias any input; n % 4 == 0, 16-byte aligned tensors.
 * The tensor arithmetic of the UniPC update (convert_model_output, multistep_uni_c_bh_update,
 * multistep_uni_p_bh_update)  lightx2v/models/schedulers/wan/scheduler.py:172-188,200-386. */
int lx2v_lincomb_f32(float* out, int64_t n, int terms, const float* const* x, const float* coef,
                     void* stream);

/* ---- classifier-free guidance: out = uncond + g * (cond - uncond), fp32 ---------------------
 * Replaces lightx2v/models/networks/wan/model.py:433. */
int lx2v_cfg_combine_f32(float* out, const float* cond, const float* uncond, float guide_scale,
                         int64_t n, void* stream);

/* ==== fp8 attention on the CDNA4 block-scaled fp8 MFMA ===================================== */

/* ---- softmax(q k^T * scale) v with both contractions in e4m3 --------------------------------
 * Same tensors and result layout as lx2v_attn_fwd_bf16 (bf16 in, bf16 out, no mask, head_dim 128).
 * A quantisation pre-pass (part of this call) writes into `workspace`: q*scale*log2(e) and k as
 * e4m3 with one power-of-two (E8M0) scale per (token, head, 32 channels); v transposed per 64-key
 * tile with one E8M0 scale per (channel, 32 keys).  P is rounded to e4m3 in registers; all
 * accumulation is fp32.  `workspace` must hold lx2v_attn_fp8_workspace_bytes(sq, skv, heads) bytes,
 * 256-byte aligned; the library allocates nothing.
 * Stands where the reference selects a quantised attention backend (SageAttn2Weight,
 * lightx2v/common/ops/attn/sage_attn.py:36-64: int8 QK^T + fp8 PV from the sageattention wheel);
 * the quantisation scheme here is CDNA4's MX block format, not that wheel's, and the reference
 * holds no fixture for either -- results are checked against the fp32 softmax and an emulation of
 * this scheme. */
int64_t lx2v_attn_fp8_workspace_bytes(int64_t sq, int64_t skv, int heads);
int lx2v_attn_fwd_fp8(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                      const uint16_t* v, int64_t ldv, uint16_t* out, int64_t ldo,
                      int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* ---- lx2v_attn_fwd_fp8 for callers that cannot allocate ---------------------------------------
 * No counterpart in the reference.  It is for callers that cannot call hipMalloc or hand over a
 * workspace: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 * Same computation and results as lx2v_attn_fwd_fp8, with the quantised q / K / V images in a fixed
 * array of LX2V_ATTN_FP8_STATIC_BYTES that every GPU the library runs on holds until the process
 * ends.  Shapes with lx2v_attn_fp8_workspace_bytes(sq, skv, heads) above that size are rejected
 * with LX2V_EUNSUP (e.g. 12 heads x 4 680 tokens need 22.6 MB; use lx2v_attn_fwd_fp8 beyond).
 * Calls no allocating routine.  All calls on one GPU share that GPU's array: calls that overlap on
 * one GPU (two streams, two threads) race; calls on one stream are ordered.  Nothing in the array
 * has to start at zero (the pre-pass of a call writes every byte its kernels read).
 * Under stream capture: the FIRST call on a device resolves that device's copy of the array
 * (hipGetSymbolAddress), so make one call per device outside capture before capturing one; a
 * captured call leaves the long-sequence round alignment off (a replay cannot advance the host's
 * view of the alignment counters), which changes no result. */
#define LX2V_ATTN_FP8_STATIC_BYTES (32 << 20)
int64_t lx2v_attn_fp8_static_bytes(void);
int lx2v_attn_fwd_fp8_static(const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
                             const uint16_t* v, int64_t ldv, uint16_t* out, int64_t ldo,
                             int64_t sq, int64_t skv, int heads, int head_dim, float scale,
                             void* stream);

/* ==== fp8 (OCP e4m3) W8A8 Linear on the CDNA4 fp8 MFMA ===================================== */

/* ---- row-wise dynamic quantisation: scale[r] = max|x[r,:]| / 448 ; out[r,:] = e4m3(x[r,:] / scale[r]) ----
 * (round to nearest even, saturating).  Per-token for activations, per-output-channel for weights.
 * Replaces act_quant_fp8_perchannel_sym_sgl (sgl_per_token_quant_fp8) / _vllm (scaled_fp8_quant,
 * use_per_token_if_dynamic)  lightx2v/common/ops/mm/mm_weight.py:492-501, and the per-channel
 * FloatQuantizer("e4m3") weight path  mm_weight.py:312-320, lightx2v/utils/quant_utils.py:132-161. */
int lx2v_quant_fp8_rowwise(const uint16_t* x, int64_t ldx, uint8_t* out, int64_t ldo, float* scale,
                           int64_t rows, int dim, void* stream);
/* The same from a source whose columns are stored as segments (`kseg` as in lx2v_gemm_bf16; NULL = plain
 * rows): the sequence-parallel attention result in its receive layout is quantised straight into the
 * plain [rows, dim] operand of lx2v_gemm_fp8, so the W8A8 output projection needs no gather copy either
 * (what the reference assembles with .contiguous() copies, lightx2v/common/ops/attn/utils/all2all.py:70-87,
 * before mm_weight.py:492-501 quantises it). */
/* The same from fp32 rows: a checkpoint weight kept in fp32 (or widened from fp16) is quantised from its own values,
 * as the reference does (`weight.to(torch.float32)` in front of the quantiser, mm_weight.py:312-320), not from a bf16 rounding of them. */
int lx2v_quant_fp8_rowwise_f32(const float* x, int64_t ldx, uint8_t* out, int64_t ldo, float* scale,
                               int64_t rows, int dim, void* stream);
int lx2v_quant_fp8_rowwise_kseg(const uint16_t* x, int64_t ldx, const lx2v_kseg* kseg, uint8_t* out, int64_t ldo,
                                float* scale, int64_t rows, int dim, void* stream);

/* ---- inverse for the fp8 Ulysses payload: out[r,:] = bf16(e4m3(q[r,:]) * bf16(scale[r])) --------
 * (the value of `q.to(bf16) * scale.to(bf16)`).  With lx2v_quant_fp8_rowwise on rows of head_dim
 * values it replaces quant_fp8_vllm / dequant_fp8_vllm  lightx2v/utils/quant_utils.py:180-186 as
 * used around the all-to-alls at lightx2v/common/ops/attn/ulysses_attn.py:59-73,125-131. */
int lx2v_dequant_fp8_rowwise(const uint8_t* q, int64_t ldq, const float* scale, uint16_t* out, int64_t ldo,
                             int64_t rows, int dim, void* stream);

/* ---- C[M,N] = epilogue((A8 @ W8^T) * a_scale[m] * w_scale[n] + bias), bf16 out ---------------
 * A8 [M,K] and W8 [N,K] are e4m3 bytes (W in the checkpoint's [out,in] layout), a_scale fp32 [M],
 * w_scale fp32 [N]; epilogues as lx2v_gemm_bf16.  K % 128 == 0, N % 8 == 0, fp8 row strides % 16.
 * Replaces MMWeightWfp8channelAfp8channeldynamicSgl.apply ("fp8-sgl": sgl_kernel.fp8_scaled_mm)
 * and its vllm / q8f siblings  lightx2v/common/ops/mm/mm_weight.py:572-605,806-834,901-928. */
int lx2v_gemm_fp8(const uint8_t* A, int64_t lda, const float* a_scale, const uint8_t* W, int64_t ldw,
                  const float* w_scale, const float* bias, uint16_t* C, int64_t ldc, int64_t M,
                  int64_t N, int64_t K, int epilogue, const uint16_t* resid, int64_t ldr,
                  const float* gate, void* stream);

/* ==== int8 W8A8 Linear on the int8 MFMA ===================================================== */

/* ---- row-wise symmetric int8 quantisation (round half to even, saturating to [-128, 127]) ------
 * weight_mode 0, activations per token: scale[r] = max|x[r,:]| / 127 ; out = rint(x * (127 / max|x[r,:]|)).
 *   Replaces act_quant_int8_perchannel_sym_vllm (vllm ops.scaled_int8_quant(x, scale=None, azp=None,
 *   symmetric=True), dynamic per-token)  lightx2v/common/ops/mm/mm_weight.py:503-505.
 * weight_mode 1, weights per output channel: scale[r] = max(max|x[r,:]|, 1e-5) / 127 ; out = rint(x / scale[r]).
 *   Replaces IntegerQuantizer(8, True, "per_channel").real_quant_tensor as load_int8_perchannel_sym uses it
 *   lightx2v/common/ops/mm/mm_weight.py:322-330, lightx2v/utils/quant_utils.py:39-49,85-92,111-113. */
int lx2v_quant_int8_rowwise(const uint16_t* x, int64_t ldx, int8_t* out, int64_t ldo, float* scale,
                            int64_t rows, int dim, int weight_mode, void* stream);
/* From fp32 rows (mm_weight.py:324-327 quantises `weight.to(torch.float32)`). */
int lx2v_quant_int8_rowwise_f32(const float* x, int64_t ldx, int8_t* out, int64_t ldo, float* scale,
                                int64_t rows, int dim, int weight_mode, void* stream);
/* Activation form (weight_mode 0) from a segmented source, as lx2v_quant_fp8_rowwise_kseg. */
int lx2v_quant_int8_rowwise_kseg(const uint16_t* x, int64_t ldx, const lx2v_kseg* kseg, int8_t* out, int64_t ldo,
                                 float* scale, int64_t rows, int dim, void* stream);

/* ---- C[M,N] = epilogue((A8 @ W8^T) * a_scale[m] * w_scale[n] + bias), bf16 out ---------------
 * lx2v_gemm_fp8 with int8 operands: the products are accumulated exactly in int32, converted to fp32
 * once, then scaled.  K % 128 == 0, K <= 131072, N % 8 == 0, row strides % 16.
 * Replaces MMWeightWint8channelAint8channeldynamicVllm.apply ("int8-vllm": cutlass_scaled_mm) and its
 * sgl / q8f / torchao siblings  lightx2v/common/ops/mm/mm_weight.py:607-641,836-866,931-990. */
int lx2v_gemm_int8(const int8_t* A, int64_t lda, const float* a_scale, const int8_t* W, int64_t ldw,
                   const float* w_scale, const float* bias, uint16_t* C, int64_t ldc, int64_t M,
                   int64_t N, int64_t K, int epilogue, const uint16_t* resid, int64_t ldr,
                   const float* gate, void* stream);

/* lx2v_gemm_fp8 / lx2v_gemm_int8 with a caller-owned workspace (no counterpart in the reference): as
 * lx2v_gemm_bf16_ws, a last round of output tiles that would fill at most a quarter of the CUs (K >= 4096) is split
 * along K over all CUs, raw partial sums in the workspace, a last small launch adding them in a fixed order.  int8:
 * the partial sums are exact int32, so every output equals the plain call's bit for bit; fp8: the tail tiles'
 * outputs differ by fp32 summation order.  lx2v_gemm_w8a8_workspace_bytes: what a call of that shape can use (0: no
 * split); a NULL, misaligned or smaller workspace only switches the split off. */
int64_t lx2v_gemm_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K);
int lx2v_gemm_fp8_ws(const uint8_t* A, int64_t lda, const float* a_scale, const uint8_t* W, int64_t ldw,
                     const float* w_scale, const float* bias, uint16_t* C, int64_t ldc, int64_t M,
                     int64_t N, int64_t K, int epilogue, const uint16_t* resid, int64_t ldr,
                     const float* gate, void* workspace, int64_t workspace_bytes, void* stream);
int lx2v_gemm_int8_ws(const int8_t* A, int64_t lda, const float* a_scale, const int8_t* W, int64_t ldw,
                      const float* w_scale, const float* bias, uint16_t* C, int64_t ldc, int64_t M,
                      int64_t N, int64_t K, int epilogue, const uint16_t* resid, int64_t ldr,
                      const float* gate, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LX2V_HIP_H */
