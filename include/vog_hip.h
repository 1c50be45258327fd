/*
 * vog_hip.h — C ABI of libvog_hip.so: the MI355X (gfx950) engine underneath the
 * VOGNet forward path.
 *
 * The reference (TheShadow29/vognet-pytorch) has no FFI: its plugin boundary is
 * the Python surface `get_mdl_loss_eval(cfg)` (code/mdl_selector.py:26-69),
 * `cls(cfg, comm)` (code/mdl_base.py:11-22) and `forward(dict) -> dict`
 * (code/mdl_conc_single.py:68-127, code/mdl_conc_sep.py:131-217). The build
 * keeps that surface in `vognet-pytorch_amd/` and binds THIS library under it
 * with ctypes. Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - every entry returns int: 0 = ok, <0 = error (message: vog_last_error()).
 *   - tensors are raw DEVICE pointers + explicit sizes; the caller owns every
 *     buffer (inputs, outputs, workspace). No allocation, no hidden sync and no
 *     global mutable state on the launch path => re-entrant across streams.
 *   - `stream` is a hipStream_t passed as void* (torch: current_stream().cuda_stream).
 *   - 16-bit tensors ("t16") are raw bf16 or IEEE f16 bit patterns, selected by
 *     a vog_dtype argument; accumulation is always fp32.
 *   - weights are HOST fp32 pointers handed over once (vog_ctx_set_weight) under
 *     the reference's state_dict key names (utils/trn_utils.py:534-593 loads
 *     exactly these keys), uploaded / converted / padded by vog_ctx_finalize.
 */
#ifndef VOG_HIP_H
#define VOG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOG_ABI_VERSION 1

typedef enum { VOG_BF16 = 0, VOG_F16 = 1 } vog_dtype;
typedef enum { VOG_MDL_IGRND = 0, VOG_MDL_VGRND = 1, VOG_MDL_VOG = 2 } vog_mdl_kind;
/* svsq is SEP with ncmp = 1 */
typedef enum { VOG_CONC_SEP = 0, VOG_CONC_TEMP = 1, VOG_CONC_SPAT = 2 } vog_conc_type;

int vog_version(void);
const char* vog_last_error(void);

/* ------------------------------------------------------------------------- *
 * Operator level (one per hot-path kernel; used by the parity tests and by
 * vog_forward internally)
 * ------------------------------------------------------------------------- */

/* C = act(A * W^T + bias) [+ residual]   — replaces nn.Linear everywhere on the
 * path (transformer_code.py:58-61,77-81,169-172; mdl_vog.py:182-188,202-207,
 * 224-230). A: [M,K] fp32 (a_is_f32=1) or t16, row pitch lda; optional row
 * gather a_rows[M] (int32) — the embedding lookup of mdl_srl_utils.py:128.
 * W: [N,K] t16 row pitch ldw (K % 8 == 0). Outputs (either may be NULL):
 * c32 fp32 / c16 t16 (type c16_dtype), row pitch ldc; output row = m*rep + j for j < rep
 * (rep > 1 broadcasts a frame's segment feature onto its proposals,
 * mdl_conc_single.py:51-66). */
struct vog_vislang_args;
typedef struct vog_gemm_args {
  const void* a; int a_is_f32; int64_t lda; const int32_t* a_rows;
  const void* w; int64_t ldw;
  const float* bias;           /* [N] or NULL */
  const float* residual;       /* [M, ldr] fp32 or NULL (rep must be 1) */
  int64_t ldr;
  float* c32; void* c16; int64_t ldc; int64_t ldc16;
  int M, N, K; int relu; int rep; vog_dtype dtype;
  int c16_dtype;               /* vog_dtype of c16, or -1 = same as dtype */
  /* optional output-row scatter: column n belongs to segment n / out_rows_ncol and
   * row m of that segment is written to row out_rows[seg*M + m] (< 0: dropped).
   * Used to emit the LSTM input projections directly in (direction, step) order. */
  const int32_t* out_rows; int out_rows_ncol;
  /* optional implicit residual: the residual row of output row m is the vis||lang
   * token m of this layout (never materialised); `residual` must then be NULL. */
  const struct vog_vislang_args* res_vislang;
  /* split-K (> 1): the K range is cut into `splitk` slices, slice s writes its raw
   * partial product to c32 + s*M*ldc (fp32 slabs, plain stores); bias / relu /
   * residual / c16 / rep must be unset — apply them with vog_splitk_finish. For
   * GEMMs with fewer output tiles than CUs and a long K (the two feature encoders). */
  int splitk;
  /* w_frag = 1: `w` is in MFMA-fragment order (vog_pack_w_frag: [N/16][K/32][64 lanes][8],
   * one contiguous KiB per fragment) — only for the M <= 64 weight-streaming kernel
   * (K % 32 == 0, N % 16 == 0), where the row-major layout makes every wave load touch
   * 16 half-used cache lines. */
  int w_frag;
  /* a_frag = 1 (M <= 64 kernel only, 16-bit A): `a` is in fragment order
   * [m/16][K/32][lane = ((k%32)/8)*16 + m%16][k%8] — written that way by vog_bilstm_step
   * (out_frag) so that the LSTM -> projection hand-off needs no strided fragment loads. */
  int a_frag;
  /* round 6, optional (M <= 64 kernel, a_is_f32, w_frag): w_lo = the fragment-ordered 16-bit remainder t16(w - t16(w)) of the fp32
   * weights; the fp32 rows of `a` are split the same way in the kernel and the product is a.w + a_lo.w + a.w_lo (three MFMAs). */
  const void* w_lo;
} vog_gemm_args;
/* host: fp32 [N, ld] (first K columns) -> 16-bit fragment order, N*K halfwords. */
int vog_pack_w_frag(const float* w, int64_t ld, int N, int K, void* dst_host, vog_dtype dtype);
int vog_gemm_bias_act(const vog_gemm_args* g, void* stream);

/* out[m*rep + j, n] = act(sum_s slab[s][m][n] + bias[n]) for up to two problems in
 * one launch (prop_encoder + seg_encoder), fp32 and/or 16-bit outputs. */
typedef struct vog_splitk_prob {
  const float* slabs; int splits; int M, N; const float* bias; int relu; int rep;
  float* c32; void* c16; int64_t ldc, ldc16; int c16_dtype;
} vog_splitk_prob;
int vog_splitk_finish(const vog_splitk_prob* p0, const vog_splitk_prob* p1, void* stream);

/* Fused QKV projection for one (Rel)MultiHead (transformer_code.py:64-67,
 * 180-183): x[S*N, K] * Wqkv_pad^T where Wqkv_pad is [3*H*dp, K] (heads padded
 * to dp columns with zero rows, dp % 32 == 0). Writes q, k, v per (sequence, head)
 * in MFMA-FRAGMENT ORDER, npad*dp halfwords each, npad = N rounded up to 32:
 *   q/k : [token/32][dd/16][lane = ((dd/8)&1)*32 + token%32][dd%8]
 *   v   : [token/32][dd/32][(token%32)/16][lane = hi*32 + dd%32][j],
 *         token%16 = 8*(j>>2) + 4*hi + (j&3)
 * (csrc/common.h frag_qk / frag_v). Pad tokens are never written: zero the
 * buffers once (vog_workspace_init does). */
typedef struct vog_qkv_args {
  const void* x16; int64_t ldx; const void* wqkv; int64_t ldw;
  void* q; void* k; void* vt;
  int S, N, H, dp, npad, K; vog_dtype dtype;
  /* structured layer 0 of mul_tx (pl != NULL): x16 holds the n_vid*nfrm*nppf VISUAL rows
   * only ([rows, K = d_vis]), wqkv its first K columns (row pitch ldw); pl = lang Wqkv[:, d_vis:]^T
   * ([n_lang*nsrl, 3*H*dp] fp32). The epilogue emits, for every visual row and every one of
   * the nsrl arguments, token (arg*nppf + p) = projection + pl[lang row of that arg]:
   * S = n_vid*nfrm sequences of N = nsrl*nppf tokens, no [tokens, d] matrix and no fp32
   * intermediate in HBM (see vog_qkv_combine for the unfused form). */
  const float* pl; int nsrl, nppf, nfrm, lang_per_vid, nc_v;
  /* kv_visual_only = 1 (with pl): q is fanned out as above, but k and vt are written for the nppf
   * VISUAL tokens of every sequence only ([S,H,npad_kv*dp], no language part): the operands of
   * vog_rel_attention_struct_fwd. */
  int kv_visual_only, npad_kv;
  /* wqkv_p32 != NULL: row-block form for MANY rows (csrc/qkvrb_dev.h; p100): the same [3*H*dp, K] weights in vog_pack_w_frag32
   * order; one workgroup per 64 rows stages its rows in LDS once and walks all output columns, streaming each weight fragment
   * once. Needs vog_qkv_rowblock_supported(3*H*dp, K); wqkv / ldw are not read. */
  const void* wqkv_p32;
  /* round 6, hi + lo operands (optional; all four or none; plain form, pl == NULL): x16_lo = t16(x - t16(x)) of the fp32
   * activations ([rows, ldx] like x16), wqkv_lo the same of the fp32 weights ([3*H*dp, ldw] like wqkv); the projection is
   * x.w + x_lo.w + x.w_lo (three MFMAs) and Q / K are written as q + q_lo, k + k_lo (V^T as one 16-bit image). */
  const void* x16_lo; const void* wqkv_lo; void* q_lo; void* k_lo;
} vog_qkv_args;
int vog_qkv_proj(const vog_qkv_args* a, void* stream);
int vog_qkv_rowblock_supported(int n_out, int K);

/* Layer-0 QKV of mul_tx through the token structure: every token is
 * [vis[v, f*nppf+p] || lang[l, a]], so x Wqkv^T = PV[vis row] + PL[lang row] with
 * PV = vis Wqkv[:, :dv]^T ([n_vid*NP, 3*H*dp] fp32) and PL = lang Wqkv[:, dv:]^T
 * ([n_lang*nsrl, 3*H*dp] fp32): 5x fewer projection FLOPs than the dense
 * [tokens, d] GEMM of transformer_code.py:180 and no token matrix in HBM. This
 * entry adds the two parts (one rounding to 16 bit) and emits q, k, v in the
 * fragment order of vog_qkv_proj. */
typedef struct vog_qkvcomb_args {
  const float* pv; const float* pl; void* q; void* k; void* vt;
  int n_vid, nfrm, nppf, nsrl, H, dp, npad; int lang_per_vid, nc_v; vog_dtype dtype;
} vog_qkvcomb_args;
int vog_qkv_combine(const vog_qkvcomb_args* a, void* stream);

/* softmax((q k^T + bias)/scale) v per (sequence, head), flash-style, with the
 * relative-position bias computed on the fly: bias[i,j] = relu(u[i]-u[j]+pe_b[h])
 * where u[token,h] = W_pe[h,:].box_norm[token,:]  (RelAttention.forward
 * transformer_code.py:136-160 + compute_pe mdl_vog.py:456-490 + do_cross
 * mdl_srl_utils.py:30-69 + Linear(5,H)+ReLU mdl_vog.py:446-451,580-585; the
 * [S,N,N,H] tensor is never materialised). use_rel=0 gives Attention.forward
 * (transformer_code.py:42-50). q, k, vt: fragment order of vog_qkv_proj (npad = N up
 * to 32). u: [n_vid, NP, H] fp32; token j of sequence s uses row
 * (s / seq_per_vid)*NP + (s % seq_per_vid)*n_box + (j % n_box).
 * out16: [S*N, H*dp] t16 row-major (heads concatenated, padded). */
#define VOG_LOGIT_WORDS 32     /* words of one logit_max report ... */
#define VOG_LOGIT_STRIDE 32    /* ... this many 32-bit words apart (one 128-byte line each) */
typedef struct vog_attn_args {
  const void* q; const void* k; const void* vt; void* out16;
  const float* u; const float* pe_b;
  int S, N, H, dp, npad; int use_rel; int n_box, seq_per_vid, NP;
  float inv_scale; vog_dtype dtype;
  /* optional: 4 device bytes owned by the caller. With it, bf16 sequences of >= 1024 tokens and head
   * dims <= 192 run attn_tile2_kernel (64 queries per wave, softmax against a fixed per-row reference
   * overlapped with the MFMAs; csrc/attn_tile2_dev.h); the flag is cleared, raised by the kernel if a
   * row left the safe range of that formulation, and a second (normally empty) launch of the
   * running-maximum kernel redoes the call when it was raised. NULL: running-maximum kernel only.
   * guard_precleared != 0: the caller guarantees *guard_flag == 0 on entry (vog_forward keeps the flags in the workspace
   * region its prologue zero-fills), so the clearing launch is skipped. */
  int* guard_flag;
  int guard_precleared;
  /* round 6, hi + lo operands (optional; both or neither): q_lo / k_lo = the 16-bit remainders t16(x - t16(x)) of the fp32 Q / K
   * projections, same fragment order as q / k (vog_qkv_args.q_lo / k_lo). With them Q.K^T = q.k + q_lo.k + q.k_lo (three MFMAs,
   * fp32 accumulate): the logits carry ~2^-21 relative operand error instead of 2^-11 (f16) - what a checkpoint with sharp
   * attention needs. Sequences of <= 256 tokens. out16_lo (optional): remainder of out16, same layout (read
   * by a hi + lo vog_tx_tail_fwd). logit_max (optional): VOG_LOGIT_WORDS device words VOG_LOGIT_STRIDE words apart (4 KiB),
   * which the launch only ever RAISES: zero them for a fresh measurement; the largest of them after the launch is the largest
   * |logit| (after bias and scale, in nats) seen since, as the bits of a non-negative float (the workgroups spread over the
   * words; a wave issues an atomic only when it would raise its word). */
  const void* q_lo; const void* k_lo; void* out16_lo; unsigned int* logit_max;
} vog_attn_args;
int vog_rel_attention_fwd(const vog_attn_args* a, void* stream);

/* Attention of mul_tx layer 0 through the token structure (exact): token (arg a, proposal p) =
 * [vis[p] || lang[a]], so k = Kv[p] + Kl[a], v = Vv[p] + Vl[a] and the bias depends on (p, p') only.
 * The logit of key (a', p') is X[p'] + Y[a'] with X = q.Kv[p'] + bias, Y = q.Kl[a']: the softmax over
 * the nsrl*nppf keys FACTORISES into softmax_p'(X) x softmax_a'(Y), and the output is
 *     softmax_p'(X) . Vv  +  softmax_a'(Y) . Vl
 * - nppf + nsrl keys per query instead of nsrl*nppf (25 instead of 100 at gt5, 405 instead of 2000
 * at p100), and no fanned-out K / V in HBM. q: [S,H,npad_q*dp] fragment order, N_q = nsrl*nppf
 * tokens (token = a*nppf + p); kv, vv: [S,H,npad_kv*dp] fragment order, nppf tokens; pl: the
 * language projection [n_lang*nsrl, 3*H*dp] fp32 of vog_qkv_args (K block at column H*dp, V block at
 * 2*H*dp); u / pe_b / seq_per_vid / NP as in vog_attn_args with n_box = nppf. out16: [S*N_q, H*dp]. */
typedef struct vog_attn_struct_args {
  const void* q; const void* kv; const void* vv; const float* pl; void* out16;
  const float* u; const float* pe_b;
  int S, H, dp, nsrl, nppf, npad_q, npad_kv, nfrm, lang_per_vid, nc_v;
  int use_rel, seq_per_vid, NP; float inv_scale; vog_dtype dtype;
  /* q_visual = 1: q holds only the nppf VISUAL query parts Qv ([S,H,npad_kv*dp], like kv) and the
   * kernel forms q(a, p) = Qv[p] + Ql[a] itself (Ql = pl columns [0, H*dp)): nothing is fanned out
   * in HBM (plain vog_qkv_proj over the visual rows, pl = NULL). npad_q is ignored. */
  int q_visual;
  /* round 5, optional: one int, zero at launch. With it, q_visual launches over several key blocks (p100) use the E x F
   * factorisation (141 instead of 330 us at cfg 4): the kernel sets it when a row may leave the safe range of its 16-bit
   * fragments and the same call then re-runs the per-row kernel (an empty launch otherwise). NULL: per-row kernels only. */
  int* guard_flag;
  /* round 6, hi + lo operands (optional; both or neither; q_visual form with nppf <= 32): the 16-bit remainders of Qv / Kv, same
   * fragment order; the language parts are split in the kernel from the fp32 `pl`. out16_lo / logit_max: as in vog_attn_args
   * (the logit bound is max|x| + max|y| of the separable parts). */
  const void* q_lo; const void* kv_lo; void* out16_lo; unsigned int* logit_max;
} vog_attn_struct_args;
int vog_rel_attention_struct_fwd(const vog_attn_struct_args* a, void* stream);

/* y = LayerNorm(x) * gamma + beta, eps 1e-5 (ResidualBlock.forward
 * transformer_code.py:30-31; the residual add is fused into the producing
 * GEMM's epilogue). x,y32: [rows,d] fp32; y16 optional t16 copy. */
int vog_residual_layernorm(const float* x, const float* gamma, const float* beta,
                           float* y32, void* y16, int rows, int d, vog_dtype dtype,
                           void* stream);

/* ---- fused encoder-layer tail ---------------------------------------------------------------
 * Everything of one (Rel)EncoderLayer after the attention is row-local, so ONE launch does
 *     x1 = LayerNorm(x + attn Wo^T)                    (RelMultiHead.forward tail transformer_code.py:184-186
 *                                                       + ResidualBlock.forward :30-31)
 *     y  = LayerNorm(x1 + W2 relu(W1 x1 + b1) + b2)    (FeedForward.forward :80-81 + ResidualBlock)
 * and, for the last mul_tx layer (score != NULL), the score head on top of it
 *     logit = lin2.2 . relu(lin2.0 y + b) + b          (mdl_vog.py:224-230,675-677)
 *     mdl_outs / mdl_outs_eval = inverse regroup + sigmoid * masks   (as vog_score_head)
 * for 64 token rows per workgroup: replaces vog_gemm_bias_act x 3, vog_residual_layernorm x 2
 * (+ the lin2 GEMM and vog_score_head) and all their intermediates in HBM.
 *   attn16  [M, kwo] t16 row-major = out16 of vog_rel_attention(_struct)_fwd (kwo = H*dp, kwo % 64 == 0)
 *   wo_p / w1_p / w2_p / wl_p: Wo_pad [d, kwo], linear1 [dh, d], linear2 [d, dh], lin2.0 [256, d] in
 *           the 32x16 fragment order of vog_pack_w_frag32 (types: dtype, dtype, dtype, head_dtype)
 *   residual [M, ldr] fp32, or res_vislang = the implicit vis||lang token matrix (exactly one of them)
 *   y32 / y16 (type y16_dtype, -1 = dtype): [M, d] outputs, either may be NULL
 *   x1_scratch: vog_tx_tail_scratch_bytes(M, d) bytes (0 for d = 512), workgroup-private spill slab
 * Shapes: d in {512, 768}, dh = d/2 (vog_tx_tail_supported); other shapes use the unfused entries. */
struct vog_score_args;
typedef struct vog_tx_tail_args {
  const void* attn16; int kwo;
  const void* wo_p; const void* w1_p; const void* w2_p;
  const float* residual; int64_t ldr;
  const struct vog_vislang_args* res_vislang;
  const float* ln1g; const float* ln1b; const float* b1; const float* b2; const float* ln2g; const float* ln2b;
  float* y32; void* y16; int y16_dtype;
  const void* wl_p; const float* bl;            /* score head (with `score`): lin2.0 packed / bias */
  const struct vog_score_args* score;            /* h1 is ignored; w2 / b2 = lin2.2 */
  int head_dtype;                                /* vog_dtype of wl_p (VOG_F16) */
  float* x1_scratch;
  int M, d, dh; vog_dtype dtype;
  /* round 6, hi + lo operands (optional; the first four together, not with `score`): attn16_lo = out16_lo of the attention,
   * w*_p_lo = the 32x16 fragment-ordered 16-bit remainders t16(w - t16(w)) of the fp32 weights. Every GEMM stage is then
   * W.X + W_lo.X + W.X_lo (three MFMAs, 32 rows per workgroup) and y16_lo (optional) receives the remainder of y16: the tail of a
   * layer whose OUTPUT feeds another attention layer of a checkpoint with sharp logits. */
  const void* attn16_lo; const void* wo_p_lo; const void* w1_p_lo; const void* w2_p_lo; void* y16_lo;
} vog_tx_tail_args;
int vog_tx_tail_supported(int d, int dh, int kwo);
int64_t vog_tx_tail_scratch_bytes(int M, int d);
int vog_tx_tail_fwd(const vog_tx_tail_args* a, void* stream);
/* One whole (Rel)EncoderLayer.forward (transformer_code.py:189-203 = RelMultiHead :176-186 + the two
 * ResidualBlocks :21-31 + FeedForward :73-81) as ONE entry - SURVEY.md 8(b)'s `vog_encoder_layer_fwd`:
 * QKV projection -> RelAttention -> tail, three launches, nothing but Q/K/V^T fragments and the 16-bit
 * attention rows in HBM between them. The three argument blocks are the ones of the stand-alone entries
 * and must be consistent (qkv.q/k/vt = attn.q/k/vt, attn.out16 = tail.attn16, tail.M = qkv.S * qkv.N,
 * tail.kwo = qkv.H * qkv.dp); the call checks that. */
typedef struct vog_encoder_layer_args {
  vog_qkv_args qkv;
  vog_attn_args attn;
  vog_tx_tail_args tail;
} vog_encoder_layer_args;
int vog_encoder_layer_fwd(const vog_encoder_layer_args* a, void* stream);

/* host: fp32 [N, ld] (first K columns) -> 16-bit 32x16 fragment order, N*K halfwords:
 * [N/32][K/16][lane = ((k%16)/8)*32 + n%32][k%8]  (N % 32 == 0, K % 16 == 0). */
int vog_pack_w_frag32(const float* w, int64_t ld, int N, int K, void* dst_host, vog_dtype dtype);

/* Proposal + segment encoders + their concat in ONE launch (prop_feats_encode / seg_feats_encode
 * mdl_vog.py:291-314, Linear+ReLU :202-207; concat_prop_seg_feats mdl_conc_single.py:51-66,156-174,
 * mdl_conc_sep.py:44-62): out[r, :prop_enc] = relu(W_p prop[r] + b_p), out[r, prop_enc:] =
 * relu(W_s seg[r / nppf0] + b_s). prop: [n_prop_rows, prop_dim] fp32, seg: [n_prop_rows/nppf0, seg_dim]
 * fp32 (read once, rounded to `dtype` in registers); w_*_f: weights in the fragment order of
 * vog_pack_w_frag, type `dtype`; c32 / c16 (type c16_dtype): [n_prop_rows, ldc], either may be NULL.
 * Needs feature dims % 256 == 0, encode sizes % 32 == 0 and <= 256 (vog_vis_encode_supported);
 * other shapes use vog_cast_f32_to_t16 + vog_gemm_bias_act (+ vog_splitk_finish). */
typedef struct vog_visenc_args {
  const float* prop; const float* seg;
  const void* w_prop_f; const void* w_seg_f; const float* b_prop; const float* b_seg;
  float* c32; void* c16; int64_t ldc; int c16_dtype;
  int n_prop_rows, nppf0, prop_dim, seg_dim, prop_enc, seg_enc; vog_dtype dtype;
  /* lean = 1: 64-row x 128-column workgroups (32 of them at cfg 2) that read the fp32 rows once per column
   * half and stage them in LDS: ~2x the latency of the wide form but ~5x less busy-CU time - the form used
   * when the encoders share the launch of a persistent BiLSTM layer (csrc/pair.hip). Same results up to
   * fp32 summation order. */
  int lean;
  /* Replication of the segment rows over the nppf0 proposals of their frame (lean form only). 0: done by
   * vog_vis_encode - inside the encoder kernel for nppf0 <= 16, otherwise by a copy kernel it launches
   * behind it (at 100 proposals per frame the 25 MB of replica stores took the 6 workgroups owning the
   * segment rows 100 us longer than everybody else). 1: vog_vis_encode writes replica 0 only and the caller
   * runs vog_seg_replicate (the forward does, so that the encoder kernel stays ONE launch and can share
   * the launch of a BiLSTM layer). */
  int defer_replicas;
  /* round 6, hi + lo operands (optional; all three or none; lean = 1, nppf0 <= 16): w_*_f_lo = the fragment-ordered 16-bit
   * remainders t16(w - t16(w)) of the fp32 weights; the fp32 feature rows are split the same way in the kernel and a k-step is
   * x.w + x_lo.w + x.w_lo (three MFMAs). c16_lo: the remainder of the output rows, laid out like c16 (read by a hi + lo
   * vog_qkv_proj). */
  const void* w_prop_f_lo; const void* w_seg_f_lo; void* c16_lo;
} vog_visenc_args;
int vog_vis_encode_supported(int prop_dim, int seg_dim, int prop_enc, int seg_enc);
int vog_vis_encode(const vog_visenc_args* a, void* stream);
/* rows r*nppf0 + j (j = 1 .. nppf0-1), columns [prop_enc, prop_enc + seg_enc) of c32 / c16 := row r*nppf0 */
int vog_seg_replicate(const vog_visenc_args* a, void* stream);

/* dst[i] = (t16) src[i] for two arrays in one launch (raw proposal / segment
 * features -> the encoders' MFMA operand type; replaces the implicit fp32 read
 * of nn.Linear in prop_feats_encode / seg_feats_encode mdl_vog.py:291-314).
 * n0, n1 multiples of 4; src1 may be NULL. */
int vog_cast_f32_to_t16(const float* src0, void* dst0, int64_t n0, const float* src1, void* dst1,
                        int64_t n1, vog_dtype dtype, void* stream);

/* u[v, r, h] = sum_c W_pe[h,c] * norm(box[v,r,c]), norm = x/vid_w, y/vid_h,
 * x/vid_w, y/vid_h, frame/nfrm_div (compute_pe mdl_vog.py:456-463).
 * props: [n_rows, 7] fp32 (pad_proposals). */
int vog_box_u(const float* props, const float* w_pe, float* u, int n_rows, int H,
              float vid_w, float vid_h, float nfrm_div, void* stream);

/* Token re-index (get_srl_arg_seq_to_sent_seq mdl_vog.py:67-95):
 * tok[b*T+t] = words[b, mask[b,t]] if mask[b,t] >= 0 else vocab_size. */
int vog_srl_gather(const int64_t* words_ind, const int64_t* word_mask, int32_t* tok,
                   int Bn, int T, int nsrl, int seq_len, int vocab_size, void* stream);
/* Packed-sequence schedule of the BiLSTM, as GEMM out_rows (pitch 4R, segment
 * width 4R): rows[dir*Bn*T + b*T + t] = dir*(T*Bn - 1) + step*Bn + b with step = t
 * (dir 0) or len_b-1-t (dir 1), so that row*4R + col lands in gxs[dir][step][b][:];
 * -1 for t >= len_b. */
int vog_lstm_schedule(const int64_t* lens, int32_t* rows, int Bn, int T, void* stream);

/* Fused prologue of the language path (one launch instead of memset + vog_srl_gather +
 * vog_lstm_schedule): zero `zero_bytes` at `zero` (multiple of 16), fill the `ones_bytes` (multiple of
 * 16) that follow them with 0xff (the "not written yet" pattern of the persistent BiLSTM's hand-off
 * slots), token re-index and the packed-sequence schedule. */
int vog_lang_prep(void* zero, int64_t zero_bytes, int64_t ones_bytes, const int64_t* words_ind, const int64_t* word_mask,
                  const int64_t* lens, int32_t* tok, int32_t* rows, int Bn, int T, int nsrl,
                  int seq_len, int vocab_size,
                  /* optional (a0_frag != NULL): also gather the tokens' 16-bit embedding rows
                   * [vocab+1, emb_dim] into a0_frag in the A-fragment order of vog_gemm_args.a_frag
                   * (rows beyond Bn*T of the last 16-row tile must already be zero) */
                  const void* emb16, void* a0_frag, int emb_dim, void* stream);

/* Fused prologue of the visual path (one launch instead of vog_cast_f32_to_t16 + up to two
 * vog_box_u): u0/u1 = bias precursors for obj_tx / mul_tx (either w_pe may be NULL). */
typedef struct vog_visprep_args {
  const float* src0; void* dst0; int64_t n0; const float* src1; void* dst1; int64_t n1; vog_dtype dtype;
  const float* props; int n_rows; float vid_w, vid_h;
  const float* w_pe0; float* u0; int H0; float nfrm_div0;
  const float* w_pe1; float* u1; int H1; float nfrm_div1;
} vog_visprep_args;
int vog_vis_prep(const vog_visprep_args* a, void* stream);
/* vog_lang_prep + vog_vis_prep in ONE launch (the two prologues are independent of each other;
 * one launch less on the forward's dependent chain). Arguments as for the two entries. */
int vog_prep_fused(void* zero, int64_t zero_bytes, int64_t ones_bytes, const int64_t* words_ind, const int64_t* word_mask,
                   const int64_t* lens, int32_t* tok, int32_t* rows, int Bn, int T, int nsrl,
                   int seq_len, int vocab_size, const void* emb16, void* a0_frag, int emb_dim,
                   const vog_visprep_args* vis, void* stream);

/* One time step of one BiLSTM layer, both directions, packed-sequence
 * semantics (LSTMEncoder.forward mdl_srl_utils.py:134-148; nn.LSTM gate order
 * i,f,g,o). gxs: [2][T][Bn][4R] fp32 = x W_ih^T + b_ih + b_hh in (direction, step)
 * order (vog_lstm_schedule + the GEMM's out_rows scatter); whh: the 16-bit
 * recurrent weights in MFMA-fragment order as packed by vog_lstm_pack_whh;
 * h_in/h_out: [Bn16, 2R] t16 ping-pong state; c: [Bn16,2R] fp32; out16:
 * [Bn*T, 2R] t16 (zero where t >= len). */
typedef struct vog_lstm_step_args {
  const float* gx; const void* whh; const void* h_in; void* h_out; float* c;
  void* out16; const int64_t* lens; int Bn, T, R, step; vog_dtype dtype;
  /* out_frag = 1: out16 is written in the A-fragment order of vog_gemm_args.a_frag
   * (K = 2R, rows m = b*T + pos) and every active step also writes h into row
   * final_row0 + b (the final hidden state ends up there). 0: row-major [.., 2R]. */
  int out_frag; int final_row0;
} vog_lstm_step_args;
int vog_bilstm_step(const vog_lstm_step_args* a, void* stream);
/* ALL T steps of one BiLSTM layer in ONE launch (persistent workgroups): 2 x R/32 workgroups of 8 waves,
 * every wave keeps its 16 rows of W_hh in registers for the whole sequence (W_hh is read once per
 * layer instead of once per step). Between steps the hidden state goes through `hx` =
 * [T slots][2 dirs][Bn][R] 16-bit values, one slot per step, which must hold 0xffff in every halfword
 * at launch (vog_lang_prep / vog_prep_fused `ones_bytes` arm it; vog_bilstm_hx_bytes gives the size):
 * every value validates itself - 0xffff is a NaN pattern no h can take - so a consumer needs no tag,
 * flag or fence, just one (re-tried) 16-byte L1-bypassing load per thread and step; the workgroup
 * stages the vector in LDS (double buffered) and every wave reads its MFMA B fragments from there.
 * Publishing: if all workgroups of a direction run on ONE XCD (they report their XCC id in `sync` at
 * start; small layers) every lane stores its value with a store that stays in that XCD's L2; otherwise
 * the workgroup's 64 bytes per sentence are written through by one wave (0.75 / 1.33 us per exchange,
 * scratch/ubench/handoff_v3.hip). Measured (cfg 2, Bn = 4, R = 1024): 2.5 us per step (3.0 in round 2).
 * Requires Bn <= 16 and R/32 in {1,2,4,32} (vog_bilstm_layer_supported); sync (256 x u32) must be zero
 * at launch (vog_lang_prep does it); every wait is bounded (~1 s): on timeout sync[2] is set, the
 * kernel drains and writes NaN into ALL its output rows so that a stalled run cannot pass for a result.
 * CO-RESIDENCY: all 2 x R/32 workgroups (one per CU) must be resident at once; launch at most 4
 * instances concurrently on a 256-CU part (HIP's 4 hardware queues guarantee that for streams). */
typedef struct vog_lstm_layer_args {
  const float* gxs; const void* whh; void* hx; uint32_t* sync; void* out16;
  const int64_t* lens; int Bn, T, R; vog_dtype dtype;
  int out_frag;   /* 1: out16 in the A-fragment order of vog_gemm_args.a_frag (as vog_bilstm_step) */
  /* Fused input projection (wih != NULL; gxs is then ignored): the kernel computes x W_ih^T + bias for
   * its own gate rows in a prologue (W_ih streamed once through registers, the layer input staged in
   * LDS) - no separate GEMM launch and no [2][T][Bn][4R] fp32 round trip. wih: vog_lstm_pack_w of
   * (weight_ih, weight_ih_reverse) [2][4R][K]; xa: layer input [Bn*T (row b*T + t), K] t16 in the
   * A-fragment order of vog_gemm_args.a_frag (pad rows of the last 16-row tile readable); bias: [2][4R]
   * fp32 = b_ih + b_hh per direction. Needs Bn*T <= vog_bilstm_fused_cols() (80: a bs = 4 batch with 20-word sentences) and
   * K % 256 == 0. */
  const void* wih; const void* xa; const float* bias; int K;
  /* round 5: sticky fault counter (optional; device memory or device-visible pinned host memory): +1 per launch whose
   * hand-off timed out. The library never clears it (sync[2] is re-zeroed by the next forward's prologue): the host reads
   * it when it looks at the results - a stalled forward is an ERROR at the API, not NaN scores with rc 0 (the reference's
   * LSTM, utils/mdl_srl_utils.py:114-169, cannot fail by scheduling). inject_stall: test hook, 1 = the launch behaves as
   * if its hand-off had timed out at once, 2 = only the workgroups of direction 1 do. Whichever workgroup ends dead first
   * reports (sync[3], re-armed by the prologue): exactly +1 per stalled launch. */
  uint32_t* fault; int inject_stall;
  /* round 6, gate table (gx_table != NULL; gxs and wih are then ignored): the layer's input is an embedding row, so its
   * projection depends on the TOKEN alone. gx_table = [vocab + 1][2][R][4] fp32, row v = emb[v] W_ih^T + b_ih + b_hh for
   * (direction, unit, gate i f g o) - built once per checkpoint for the whole vocabulary (vog_ctx_finalize; 164 MB at vocab 5000,
   * R = 1024); tok = [Bn * T] token ids (vog_lang_prep). The kernel reads its gate inputs of step s + 1 from the rows of
   * that step's tokens while step s runs: no input projection for this layer, neither as a GEMM launch nor in the kernel's
   * prologue, for any Bn * T (LSTMEncoder.forward, utils/mdl_srl_utils.py:134-148). */
  const float* gx_table; const int32_t* tok;
} vog_lstm_layer_args;
int vog_bilstm_layer_supported(int Bn, int R);
int vog_bilstm_fused_cols(void);                      /* largest Bn*T the in-kernel input projection (wih) takes: 80 */
int64_t vog_bilstm_hx_bytes(int Bn, int T, int R);   /* size of vog_lstm_layer_args.hx */
int vog_bilstm_layer(const vog_lstm_layer_args* a, void* stream);

/* host: [2][4R][R] fp32 (weight_hh_l*, weight_hh_l*_reverse) -> fragment order, 16 bit.
 * dst holds 2*4R*R halfwords: [dir][unit/4][k/32][lane 64][8]. */
int vog_lstm_pack_whh(const float* whh_fwd, const float* whh_bwd, void* dst_host, int R, vog_dtype dtype);
/* same packing for a [4R][K] pair (weight_ih): [dir][unit/4][K/32][lane 64][8], K % 32 == 0 */
int vog_lstm_pack_w(const float* w_fwd, const float* w_bwd, void* dst_host, int R, int K, vog_dtype dtype);

/* lang[b,a,:] = relu(W [full[b,cap0] || full[b,cap1]] + bias) * msk
 * (retrieve_srl_arg_from_lang_encode mdl_vog.py:97-140). full: [Bn*T, L] fp32. */
int vog_srl_argvec(const float* full, const int64_t* capture, const int64_t* inds_msk,
                   const float* w, const float* bias, float* lang,
                   int Bn, int T, int nsrl, int L, void* stream);

/* Build the mul_tx token matrix (concate_vis_lang_feats mdl_vog.py:316-344 +
 * the regroup of conc_encode2 mdl_vog.py:693-699): row (s=(v,f), j=a*nppf+p) =
 * [vis[v, f*nppf+p, :dv] || lang[lv(v), a, :dl]]; writes fp32 and t16 copies. */
typedef struct vog_vislang_args {
  const float* vis; const float* lang; float* x32; void* x16;
  int n_vid, nfrm, nppf, nsrl, dv, dl; int lang_per_vid; /* 1: lang row = v, 0: v / nc_v */
  int nc_v; vog_dtype dtype;
} vog_vislang_args;
int vog_vislang_layout(const vog_vislang_args* a, void* stream);

/* lin2 second layer + inverse regroup + masks (mdl_vog.py:675-677, 724-737;
 * mdl_conc_single.py:118-122): logit = w2.h1[row] + b2 scattered to
 * mdl_outs[v, a, f*nppf+p]; eval = sigmoid * arg_msk * cmp_msk. */
typedef struct vog_score_args {
  const float* h1; const float* w2; const float* b2;
  const int64_t* arg_msk; const int64_t* cmp_msk;
  float* outs; float* outs_eval;
  int n_vid, nfrm, nppf, nsrl, dh; int conc_type; int ncmp, nc_v, nvl, nfrm0, nppf0;
} vog_score_args;
int vog_score_head(const vog_score_args* a, void* stream);

/* pred_cmp head of SEP (get_seg_verb_feats_to_process / compute_seg_verb_feats_out
 * mdl_vog.py:365-397, compute_fin_scores mdl_conc_sep.py:64-129). */
typedef struct vog_predcmp_args {
  const float* final_hidden;   /* [B*nvl, L] fp32 */
  const float* prop_seg;       /* [B*ncmp, NP, dps] fp32; seg part = cols [dp0, dps) */
  const float* w0; const float* b0; const float* w2; const float* b2;
  const float* outs;           /* mdl_outs [B, ncmp, nsrl, NP] */
  const int64_t* arg_msk; const int64_t* cmp_msk; const int64_t* verb_ind;
  float* vidf_outs; float* fin_scores_loss; float* fin_scores;
  int B, ncmp, nvl, nsrl, NP, nfrm0, nppf0, L, dp0, dps;
} vog_predcmp_args;
int vog_pred_cmp_head(const vog_predcmp_args* a, void* stream);

/* Evaluator*.get_out_results_boxes (eval_vsrl_corr.py:162-220,289-345,357-424):
 * per (query,arg,video,frame) max/argmax over the frame's proposals, gather the
 * 7-d proposal row, pred_cmp index. Output is ONE packed record per query (the
 * unit of the cross-rank all-gather that replaces the pickle-file gather of
 * eval_vsrl_corr.py:125-140):
 *   float boxes[nsrl][ncmp][nfrm0][7]; float scores[nsrl][ncmp][nfrm0];
 *   int64 indexs[nsrl][nfrm0]  (temp: zeros; the reference returns float zeros) */
typedef struct vog_pred_args {
  const float* outs_eval; const float* props; const float* fin_scores; void* rec;
  int B, ncmp, nsrl, nfrm0, nppf0; int conc_type;
  /* round 6, optional: logit_max = the forward's [2 stacks][4 layers] reports of vog_attn_args.logit_max (4 KiB each); the head (the last
   * kernel of a forward) folds them into stats[0] (obj_tx) / stats[1] (mul_tx) with a system-scope atomic max - pinned host
   * memory, read by the host without a device synchronisation (vog_batch.stats). */
  const unsigned int* logit_max; unsigned int* stats;
  unsigned int* published;     /* 2 device words, zero at first use: what this caller's forwards last folded into `stats` (the host
                                * word is only touched when the value rises) */
} vog_pred_args;
int64_t vog_pred_record_bytes(int ncmp, int nsrl, int nfrm0);
int vog_pred_head(const vog_pred_args* a, void* stream);

/* SPAT / TEMP batch assembly on the device (verb_item_getter_SPAT / _TEMP,
 * code/dat_loader_simple.py:1046-1338): per-video items of B queries ([B, ncmp, ...], what
 * AV_CS.itemcollector stacks) -> the tensors the forward and the loss read, written straight into the
 * caller's (slot's) input buffers. Forward part (always): props [B,ncmp,NPv,7] -> [B,ncmp*NPv,7]
 * (spat: x1,x2 += vid_w*video, (frame,video,prop) order; temp: frame += nfrm0*video), region features
 * [.., prop_dim] and pnt mask (bytes, optional) re-ordered the same way, seg_feature_for_frms
 * [B,ncmp,nfrm0,seg_dim] -> [B,ncmp*nfrm0,seg_dim] (spat: (frame,video) order). Loss part (gt_in != NULL):
 * gt boxes [B,ncmp,G,5] shifted the same way, first num_box[b,v] of every video concatenated and zero
 * padded -> [B,G,5]; srl_boxes += boxes in front of target_cmp where srl_boxes_lens > 0; frm_mask
 * [B,ncmp*NPv,G] bytes = frame(prop) != frame(gt) for g < total boxes else 1; num_box_out [B] int64.
 * Bit-exact with the reference (fp32 adds, copies). */
typedef struct vog_assemble_args {
  const float* props_in; float* props_out; const float* region_in; float* region_out;
  const float* seg_in; float* seg_out; const unsigned char* pnt_in; unsigned char* pnt_out;
  const float* gt_in; float* gt_out; const int64_t* num_box; int64_t* num_box_out; const int64_t* target_cmp;
  const int64_t* srl_boxes_in; int64_t* srl_boxes_out; const int64_t* srl_boxes_lens; unsigned char* frm_out;
  int B, ncmp, nfrm0, nppf0, prop_dim, seg_dim, G, nv, nsrl, nbox, conc_type; float vid_w;
} vog_assemble_args;
int vog_assemble_batch(const vog_assemble_args* a, void* stream);

/* Byte ranges src -> dst in ONE launch (16-byte aligned pointers, any length, <= VOG_MAX_COPY_SEGS ranges). The sources may be
 * pinned host memory (hipHostMalloc / torch pin_memory: mapped into the device's address space): the kernel then reads over the
 * host link - the reference's `batch[k].to(device)` of the small per-batch arrays (code/utils/trn_utils.py:478, :562) without
 * one DMA transfer, and one transfer's fixed latency, per array. vog_assemble_batch accepts pinned-host *_in pointers the same way. */
#define VOG_MAX_COPY_SEGS 24
typedef struct vog_copy_seg { const void* src; void* dst; size_t bytes; } vog_copy_seg;
int vog_copy_segments(const vog_copy_seg* segs, int n, void* stream);

/* Loss of one batch on the device: LossB_TEMP / LossB_SPAT
 * (code/mdl_conc_single.py:180-433) and LossB_SEP (code/mdl_conc_sep.py:220-447) with the IoU targets of
 * utils/box_utils.py:61-118: target[b,v,a,r] = max_k(IoU(prop r, gt box srl_boxes[b,v,a,k]) * mask *
 * [video(r) == target_cmp[b]] * srl_boxes_lens[b,v,a,k]) > 0.5; masked mean of BCE-with-logits over
 * mdl_outs, times the number of proposals, times loss_lambda. sep also returns the verb loss.
 *   mdl_outs [B, (ncmp if sep else 1), nsrl, NP]; NP = proposals per row block (temp / spat: all ncmp videos)
 *   pad_proposals [B,(ncmp,)NP,7], pad_gt_bboxs [B,(ncmp,)G,5] fp32; pad_frm_mask [B,(ncmp,)NP,G],
 *   pad_pnt_mask [B,(ncmp,)NP] bytes (nonzero = the IoU counts); srl_boxes / srl_boxes_lens
 *   [B,nv,nsrl,nbox], srl_arg_boxes_mask [B,nv,nsrl], target_cmp [B], num_cmp_msk [B,ncmp],
 *   verb_cmp [B,ncmp], verb_cross_cmp_msk [B,ncmp,ncmp] int64.
 *   out: 6 floats = loss, mdl_out_loss, verb_loss (0 unless sep), number of elements in the mean, 1 if the
 *   mean is the masked one (0: no argument has boxes -> plain mean), rows in the verb-loss mean.
 *   scratch: vog_loss_scratch_bytes(). Deterministic (fixed-order reduction, no atomics).
 * vog_loss_bwd: d loss / d mdl_outs [same shape] = (sigmoid(x) - target) [* video mask for sep] * NP * lambda / n
 *   for the elements in the mean, 0 elsewhere, and (optional, sep) d verb_loss / d vidf_outs [B, ncmp]; call after
 *   vog_loss_fwd with the same args (it reads out[3..5]). First link of the training path. */
typedef struct vog_loss_args {
  const float* mdl_outs; const float* vidf_outs;
  const float* pad_proposals; const float* pad_gt_bboxs;
  const unsigned char* pad_frm_mask; const unsigned char* pad_pnt_mask;
  const int64_t* srl_boxes; const int64_t* srl_boxes_lens; const int64_t* srl_arg_boxes_mask;
  const int64_t* target_cmp; const int64_t* num_cmp_msk; const int64_t* verb_cmp; const int64_t* verb_cross_cmp_msk;
  float* out; float* scratch;
  int B, ncmp, nv, nsrl, nbox, NP, G, nppf0, conc_type; float loss_lambda;
} vog_loss_args;
int64_t vog_loss_scratch_bytes(const vog_loss_args* a);
int vog_loss_fwd(const vog_loss_args* a, void* stream);
int vog_loss_bwd(const vog_loss_args* a, float* grad_mdl_outs, float* grad_vidf_outs, void* stream);

/* SURVEY.md 8(f)-4, first slice of the backward: from d loss / d mdl_outs (vog_loss_bwd) through the score
 * head (lin2.0 -> ReLU -> lin2.2, code/mdl_vog.py:224-230, 675-677) and the tail of the LAST mul_tx encoder
 * layer (Wo + residual + LayerNorm + FFN + residual + LayerNorm, code/transformer_code.py:21-31, 73-81,
 * 189-203) to the gradients of every parameter on that path and of the tail's two inputs. fp32 on the
 * fp32 matrix pipe; the forward tail keeps nothing, so the activations are recomputed from `attn` and `x`.
 * All pointers are device fp32. attn: [M, d] the concatenated heads (input of Wo); x: [M, d] the layer
 * input (residual); rows m = (sequence s = (video, frame), token j = arg*nppf + p), M = n_vid*nfrm*nsrl*nppf;
 * d_mdl_outs: [n_vid, nsrl, nfrm*nppf] as vog_loss_bwd writes it. Weights in the reference's own shapes
 * (wo [d,d], w1 [dh,d], w2 [d,dh], wl = lin2.0.weight [dhead,d], wl2 = lin2.2.weight [dhead]); g_*: the
 * gradients, same shapes (g_bl2: 1 value); d_attn / d_x: [M, d] or NULL. Pinned against autograd through
 * the reference modules (tests/golden/bwd__*.npz, oracle/make_golden_bwd.py). */
typedef struct vog_tail_bwd_args {
  const float* attn; const float* x; const float* d_mdl_outs;
  float* d_attn; float* d_x;
  const float *wo, *ln1g, *ln1b, *w1, *b1, *w2, *b2, *ln2g, *ln2b, *wl, *bl, *wl2;
  float *g_wo, *g_ln1g, *g_ln1b, *g_w1, *g_b1, *g_w2, *g_b2, *g_ln2g, *g_ln2b, *g_wl, *g_bl, *g_wl2, *g_bl2;
  void* scratch; size_t scratch_bytes;
  int M, d, dh, dhead, n_vid, nfrm, nppf, nsrl;
  /* round 3, second slice: the same tail for a layer that is NOT followed by the score head (obj_tx, inner mul_tx
   * layers): no_head = 1, d_y [M, d] = gradient of the layer's output (wl / bl / wl2 / d_mdl_outs / dhead unused).
   * y_out (optional, any mode) receives the layer's fp32 output; no_head with d_y == NULL recomputes the forward
   * only (y_out required, no gradient is written). */
  int no_head; const float* d_y; float* y_out;
  /* train-mode dropout of the two sub-layer outputs (ResidualBlock: x + dropout(layer(x)), transformer_code.py:31):
   * probability drop_p (0 = off), masks from the counter-based generator of csrc/backward.hip (seed, sites drop_site + 1
   * and drop_site + 2, element = row * d + column); forward recomputation and backward use the same masks. */
  float drop_p; unsigned long long drop_seed; int drop_site;
} vog_tail_bwd_args;
int64_t vog_mul_tail_bwd_scratch_bytes(int M, int d, int dh, int dhead);
int vog_mul_tail_bwd(const vog_tail_bwd_args* a, void* stream);

/* Attention + Q / K / V projections of one (Rel)EncoderLayer in fp32 (code/transformer_code.py:136-162, 21-31;
 * box bias code/mdl_vog.py:446-451, 456-490): the other half of the layer's backward (vog_mul_tail_bwd is the
 * tail). x [S*N, d] = layer input, rows (sequence s, token i); token i = arg*n + p, the bias of (p, q) tiled over
 * the N/n argument blocks; heads = torch.chunk(d, n_heads) (unequal and odd sizes allowed); scale sqrt(d).
 *   forward  (d_cat == NULL): cat_out [S*N, d] = concatenated heads softmax((Q K^T + bias) / scale) V.
 *   backward (d_cat != NULL): g_wq / g_wk / g_wv [d, d], g_pe_w [H, 5], g_pe_b [H] (written), d_x [S*N, d] =
 *   (accumulate_dx ? d_x : 0) + dQ Wq + dK Wk + dV Wv; cat_out optional.
 * props [S*n, prop_stride >= 5] = (x1, y1, x2, y2, frame) per proposal, normalised by (vid_w, vid_h, vid_w, vid_h,
 * nfrm_div) as compute_pe does; props == NULL: no bias (use_rel off). Activations are recomputed (Q, K, V, the
 * probabilities of one head at a time) in `scratch`. */
typedef struct vog_attn_f32_args {
  const float* x; const float* d_cat;
  const float *wq, *wk, *wv;
  const float* props; int prop_stride; float vid_w, vid_h, nfrm_div;
  const float *pe_w, *pe_b;
  float* cat_out;
  float *g_wq, *g_wk, *g_wv, *g_pe_w, *g_pe_b;
  float* d_x; int accumulate_dx;
  void* scratch; size_t scratch_bytes;
  int S, N, n, d, n_heads;
  /* train-mode dropout on the attention probabilities (transformer_code.py:50, 153): element ((s*H + h)*N + i)*N + j of
   * site drop_site; 0 = off */
  float drop_p; unsigned long long drop_seed; int drop_site;
} vog_attn_f32_args;
int64_t vog_attn_f32_scratch_bytes(int S, int N, int n, int d);
int vog_attn_f32(const vog_attn_f32_args* a, void* stream);

/* Backward of concate_vis_lang_feats + the (frame, argument) regroup in front of mul_tx (code/mdl_vog.py:316-344,
 * 681-744). d_x [(q, v, f), (arg, p), dobj + dlang] = gradient of mul_tx's input ->
 *   d_ps   [(q, v), f*nppf + p, dobj]            summed over the arguments,
 *   d_lang [(q, v | 1), arg, dlang] (optional)   summed over the proposals (and the videos when the argument vectors
 *   are shared by them: lang_per_vid == 0), zero where inds_msk [(q, v | 1), arg] (int64, optional) is 0
 *   (retrieve_srl_arg_from_lang_encode's mask, code/mdl_vog.py:138-140). Fixed summation order. */
int vog_conc_f32_bwd(const float* d_x, float* d_ps, float* d_lang, const int64_t* inds_msk, int n_q, int nc_v, int nfrm,
                     int nppf, int nsrl, int dobj, int dlang, int lang_per_vid, void* stream);

/* Linear (+ ReLU) in fp32, forward recomputation and backward: y = act(x W^T + b), x [M, K] (row stride ldx, 0 = K),
 * W [N, K]. dy == NULL: forward only (y required). Otherwise dy [M*rep, ldy] is the gradient of the output rows, each
 * output row having been replicated `rep` times downstream (segment rows over the proposals of their frame,
 * concat_prop_seg_feats code/mdl_conc_single.py:51-66; rep = 1 otherwise; ldy = 0 means N; dy may point into a wider
 * matrix): g_w [N, K], g_b [N] (optional), d_x [M, ldx] (optional; accumulate_dx adds). The prop / segment encoders
 * (code/mdl_vog.py:291-314), lstm_out_feat_proj and srl_arg_words_out_enc (:250-283, 97-140). */
typedef struct vog_linear_f32_args {
  const float* x; int64_t ldx; const float *w, *b; int relu;
  float* y;
  const float* dy; int64_t ldy; int rep;
  float *g_w, *g_b, *d_x; int accumulate_dx;
  void* scratch; size_t scratch_bytes;
  int M, N, K;
} vog_linear_f32_args;
int64_t vog_linear_f32_scratch_bytes(int M, int N);
int vog_linear_f32(const vog_linear_f32_args* a, void* stream);

/* Language side in fp32, forward recomputation and backward: token re-index (get_srl_arg_seq_to_sent_seq
 * code/mdl_vog.py:67-95) -> embedding -> packed multi-layer BiLSTM (LSTMEncoder utils/mdl_srl_utils.py:114-169) ->
 * lstm_out_feat_proj on every step (code/mdl_vog.py:250-283) -> argument vectors relu(W [h(start) | h(end)] + b)
 * (retrieve_srl_arg_from_lang_encode :97-140, before the argument mask).
 *   words_ind [Bn, words_len], word_mask [Bn, mask_len], lens [Bn], capture [Bn, nsrl, 2]: int64, device, as vog_batch;
 *   T = longest sentence of the batch; emb [vocab_size + 1, E]; w_ih[l][dir] [4R, E | 2R], w_hh [4R, R], b_* [4R]
 *   (dir 0 = forward, 1 = reverse; gate order i, f, g, o); w_proj [D, 2R]; w_arg [L, 2D].
 *   d_lang_enc == NULL: forward only (lang_enc_out [Bn*nsrl, L] and / or full_out [Bn*T, D]).
 *   Otherwise d_lang_enc [Bn*nsrl, L] = gradient of the argument vectors (already masked) and every g_* is written:
 *   back-propagation through time with packed-sequence semantics (a sentence's state is frozen past its length, its
 *   outputs there are zero). The projection of final_hidden feeds only the sep head and gets no gradient here. */
typedef struct vog_lang_f32_args {
  const int64_t *words_ind, *word_mask, *lens, *capture;
  int Bn, nsrl, words_len, mask_len, T, vocab_size, E, R, layers, D, L;
  const float* emb;
  const float* w_ih[4][2]; const float* w_hh[4][2]; const float* b_ih[4][2]; const float* b_hh[4][2];
  const float *w_proj, *b_proj, *w_arg, *b_arg;
  const float* d_lang_enc;
  float *lang_enc_out, *full_out;
  float* g_emb;
  float* g_w_ih[4][2]; float* g_w_hh[4][2]; float* g_b_ih[4][2]; float* g_b_hh[4][2];
  float *g_w_proj, *g_b_proj, *g_w_arg, *g_b_arg;
  void* scratch; size_t scratch_bytes;
  float* hid_out;   /* optional [Bn, D]: lstm_out_feat_proj(final_hidden[-1]) - the verb feature of the sep head */
  /* train-mode dropout of LSTMEncoder (utils/mdl_srl_utils.py:104, 128, 150): drop_in on the embedded tokens (site 1),
   * drop_out behind every BiLSTM layer (site 2 + l between layers, 10 on the output); 0 = off */
  float drop_in, drop_out; unsigned long long drop_seed;
  /* 1: `scratch` still holds the forward of an earlier call with the same inputs, weights and dropout seed (no other call used
   * the buffer in between): the backward starts from those activations instead of recomputing them */
  int reuse_forward;
} vog_lang_f32_args;
int64_t vog_lang_f32_scratch_bytes(int Bn, int T, int nsrl, int E, int R, int layers, int D, int L);
int vog_lang_f32(const vog_lang_f32_args* a, void* stream);

/* The remaining forward pieces of the fp32 training path (the forward vog_*_bwd differentiate, kept in fp32 with
 * its activations; the 16-bit inference forward above keeps none):
 *   vog_concat_rows_f32: out[m, :Na] = a[m / rep_a], out[m, Na:] = b[m / rep_b] (concat_prop_seg_feats: rep_b = nppf0);
 *   vog_conc_f32_fwd:    mul_tx's input from obj_tx's output and the (masked) argument vectors - the forward of
 *                        vog_conc_f32_bwd, same argument meaning;
 *   vog_score_head_f32:  mdl_outs [n_vid, nsrl, nfrm*nppf] = lin2(y) regrouped (code/mdl_vog.py:224-230, 724-737),
 *                        scratch >= M * dhead * 4 bytes.
 * vog_adam_f32: one torch.optim.Adam step (no weight decay / amsgrad; the reference uses betas (0.9, 0.99),
 * code/main_dist.py:55) on one parameter tensor; step counts from 1. */
int vog_concat_rows_f32(const float* a, int Na, int rep_a, const float* b, int Nb, int rep_b, float* out, int M, void* stream);
int vog_conc_f32_fwd(const float* ps, const float* lang, const int64_t* inds_msk, float* out, int n_q, int nc_v, int nfrm, int nppf,
                     int nsrl, int dobj, int dlang, int lang_per_vid, void* stream);
int vog_score_head_f32(const float* y, const float* wl, const float* bl, const float* wl2, const float* bl2, float* mdl_outs,
                       void* scratch, size_t scratch_bytes, int M, int d, int dhead, int n_vid, int nfrm, int nppf, int nsrl,
                       void* stream);
int vog_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
                 void* stream);
/* Backward of vog_score_head_f32 alone (ImgGrnd / VidGrnd: lin2 reads the [vis | lang] token matrix directly,
 * code/mdl_vog.py:224-230, 286-344): g_wl [dhead, d], g_bl [dhead], g_wl2 [dhead], g_bl2 [1], d_x [M, d] (optional). */
int64_t vog_score_head_f32_bwd_scratch_bytes(int M, int d, int dhead);
int vog_score_head_f32_bwd(const float* x, const float* d_mdl_outs, const float* wl, const float* bl, const float* wl2,
                           float* g_wl, float* g_bl, float* g_wl2, float* g_bl2, float* d_x, void* scratch, size_t scratch_bytes,
                           int M, int d, int dhead, int n_vid, int nfrm, int nppf, int nsrl, void* stream);
/* Switches of the training path. "bf16_gemm" = 1: the tile GEMMs of vog_*_f32 / vog_*_bwd round their operands to bf16 and
 * use the 16-bit matrix instruction with fp32 accumulation (mixed precision: faster, gradients within ~1e-2 of the fp32
 * ones); 0 (default): fp32 operands, the path pinned against autograd through the reference.
 * The switch belongs to the CALLING THREAD (thread-local; no process-wide state): it selects the kernels of the vog_*_f32 /
 * vog_*_bwd calls that thread issues afterwards. A trainer sets it at the top of every step (train.FP32Trainer), so two
 * trainers with different settings - in one thread or in two - never see each other's choice. */
int vog_train_set_int(const char* name, int value);
int vog_train_get_int(const char* name, int* value);
/* out[g, n] = mean over f of x[g, f, n] (the segment mean of the sep verb head, code/mdl_conc_sep.py:64-129) */
int vog_row_mean_f32(const float* x, float* out, int G, int F, int N, void* stream);

/* ------------------------------------------------------------------------- *
 * Whole forward (replaces Conc{TEMP,SPAT,SEP}.forward + the evaluator head)
 * ------------------------------------------------------------------------- */
typedef struct vog_model_desc {
  int mdl_kind, conc_type;
  int vocab_size, emb_dim, rnn_size, rnn_layers;
  int prop_dim, seg_dim, prop_enc, seg_enc, lang_enc;
  int obj_layers, obj_heads, obj_use_rel, obj_one_frm, obj_to_use;
  int mul_layers, mul_heads, mul_use_rel;
  int nfrm0, nppf0, nsrl, seq_len;
  float vid_w, vid_h;
  int tx_dtype;      /* vog_dtype of the two transformers */
  int enc_dtype;     /* vog_dtype of LSTM / encoders / score head (default f16) */
} vog_model_desc;

typedef struct vog_ctx vog_ctx;
int vog_ctx_create(const vog_model_desc* d, vog_ctx** out);
int vog_ctx_set_weight(vog_ctx* c, const char* name, const float* host, int64_t numel);
int vog_ctx_finalize(vog_ctx* c);          /* synchronous; uploads + converts */
int vog_ctx_destroy(vog_ctx* c);
int vog_ctx_num_weights(const vog_ctx* c); /* names the model expects */
const char* vog_ctx_weight_name(const vog_ctx* c, int i);
int64_t vog_ctx_weight_numel(const vog_ctx* c, int i);

typedef struct vog_batch {
  int B, ncmp, T;                      /* T = max sentence length of the batch (host) */
  const int64_t* srl_arg_words_ind;    /* [B,nvl,nsrl,seq_len] */
  const int64_t* srl_arg_word_mask;    /* [B,nvl,seq_len]  (not modified) */
  const int64_t* srl_arg_word_mask_len;/* [B,nvl] */
  const int64_t* srl_arg_words_capture;/* [B,nvl,nsrl,2] */
  const int64_t* srl_arg_inds_msk;     /* [B,nvl,nsrl] */
  const int64_t* num_cmp_msk;          /* [B,ncmp] */
  const int64_t* verb_ind_in_srl;      /* [B,ncmp] (sep) or NULL */
  const float* pad_region_feature;     /* [B,(ncmp,)NP,prop_dim] */
  const float* seg_feature_for_frms;   /* [B,(ncmp,)F,seg_dim] */
  const float* pad_proposals;          /* [B,(ncmp,)NP,7] */
  /* outputs (device, caller-owned) */
  float* mdl_outs; float* mdl_outs_eval;           /* [B,nc_v,nsrl,NP] */
  float* vidf_outs; float* fin_scores_loss; float* fin_scores;  /* sep only */
  void* pred_rec;                                  /* [B] packed records or NULL */
  /* Optional: argument vectors computed elsewhere (vog_lang_forward over a GROUP of batches).
   * shared_lang != NULL: this batch's rows [B*nvl*nsrl, lang_enc] of the group encoder's output;
   * the language chain is skipped and the four word-level inputs above may be NULL.
   * shared_final_hidden: its rows [B*nvl, lang_enc] of the final hidden projection (sep only). */
  const float* shared_lang;
  const float* shared_final_hidden;
  /* Optional (round 5): sticky stall counter of this batch's forwards, see vog_lstm_layer_args.fault. Pinned host memory
   * lets the host poll it without synchronising the device. */
  uint32_t* fault;
  /* Optional (round 6): two words of pinned host memory, the largest |attention logit| (nats; bits of a non-negative float)
   * the forwards of this batch have seen in obj_tx / mul_tx. Sticky maximum, owned by the host (it may reset it): the
   * run-time check behind the per-checkpoint precision plan (engine.py: a logit scale outside the envelope of the operand
   * precision in use is reported and the plan is raised). Needs pred_rec (the prediction head publishes it). */
  uint32_t* stats;
} vog_batch;

int64_t vog_workspace_bytes(const vog_ctx* c, int B, int ncmp, int T);
/* zero the parts of a fresh workspace that kernels rely on (V^T key padding,
 * LSTM state padding). Call once per (workspace, shape). */
int vog_workspace_init(const vog_ctx* c, int B, int ncmp, int T, void* ws, size_t ws_bytes,
                       void* stream);
int vog_forward(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes, void* stream);

/* named intermediate inside the workspace (parity tests): returns offset/bytes */
int vog_workspace_stage(const vog_ctx* c, int B, int ncmp, int T, const char* stage,
                        int64_t* offset, int64_t* bytes);

/* hipGraph capture of one forward with fixed pointers (launch-bound regime:
 * ~50 launches per batch). */
typedef struct vog_graph vog_graph;
int vog_graph_capture(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes,
                      void* stream, vog_graph** out);
int vog_graph_launch(vog_graph* g, void* stream);
/* The same graph with the batch's way onto the device in front of the forward ("fed" graph): one host -> device DMA transfer
 * (memcpy node) of dma->bytes if dma != NULL, then vog_assemble_batch(asm_args) if asm_args != NULL, then
 * vog_copy_segments(segs, nseg) if nseg > 0, then the forward - so that a step of a host-fed loop is ONE hipGraphLaunch: the
 * loader writes the next batch into pinned host memory at fixed addresses. Two forms: zero copy (dma == NULL; asm_args' *_in
 * pointers and the segments' sources ARE the pinned host buffer, the kernels read it over the host link: no transfer set-up
 * latency, ~40 GB/s) and DMA (the packed host buffer goes to a device staging buffer first, the kernels read that: the copy
 * engine's ~55 GB/s on large batches). The caller must not rewrite the host buffer before the launch that reads it has
 * completed. */
int vog_graph_capture_fed(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes, const vog_copy_seg* dma,
                          const vog_assemble_args* asm_args, const vog_copy_seg* segs, int nseg, void* stream, vog_graph** out);
/* Integer options of a context: eight switches and the precision plan (round 6 removed chain_obj_qkv, pair_attn, fused_argvec,
 * fused_pred, qkv_lean and graph_dag with the measured-negative paths behind them: scratch/negatives/r6_pruned/).
 * "tx_split" (default 0; set by engine.py from the checkpoint): hi + lo 16-bit operands, see vog_ctx_split_supported below.
 * "lstm_persistent" (default 1; env VOG_LSTM_PERSISTENT presets it): use vog_bilstm_layer instead
 * of T step launches where vog_bilstm_layer_supported. Its co-residency limit (4 instances) is met
 * automatically on HIP streams (4 hardware queues execute at most 4 kernels at once); set it to 0
 * with GPU_MAX_HW_QUEUES > 4. "lstm_inject_stall": test hook (vog_lstm_layer_args.inject_stall).
 * "fused_tail" (default 1): run everything after the attention of an encoder layer (and, for the
 * last mul_tx layer, lin2 + the score head) as ONE vog_tx_tail_fwd launch where
 * vog_tx_tail_supported; 0 = the separate GEMM / LayerNorm / score launches (always used for other shapes).
 * "fused_enc" (default 1): vog_vis_encode instead of cast + two split-K GEMMs + finish where supported.
 * "enc_lean" (default -1 = the 64-row stream form exactly when the encoders will share a BiLSTM layer's launch; 0 / 1 force).
 * "pair_launches" (default 1): step i of the language chain (input projection / BiLSTM layer / out-projection)
 * and step i of the visual chain (encoders / obj_tx QKV, attention, tail / mul_tx QKV) - independent until
 * mul_tx's attention - share ONE launch (csrc/pair.hip: blocks [0, nA) run one kernel body, the rest the
 * other) wherever a pair kernel exists for the two shapes; 0 = every step its own launch. Same kernel
 * bodies either way: results are bit-identical (tests/test_gpu_forward.py). "pair_mask" (default 15): which of the
 * pairs are formed (1 BiLSTM layer 0 + encoders, 2 layer 1 + obj_tx tail, 4 out-projection + mul_tx QKV, 8 (round 6) layer-1 input
 * projection + obj_tx QKV where that projection is a GEMM launch: more than 80 (sentence, position) columns).
 * "fused_ih" (default 1): the BiLSTM input projections run inside the persistent layer kernel
 * (vog_lstm_layer_args.wih) instead of as GEMM launches, where Bn*T <= 80 and K % 256 == 0; round 6: where that prologue
 * does not reach (more than 80 columns: cfg 3, cfg 5, grouped requests) layer 0 reads its gate inputs from the checkpoint's
 * gate table (vog_lstm_layer_args.gx_table, built by vog_ctx_finalize; VOG_GX_TABLE_MAX_MB, default 2048, bounds its size)
 * instead of running a GEMM launch (0: GEMM launches only, 2: layer 0 only, 3: layers >= 1 only, 4: as 1 without the table,
 * 5: the table wherever the checkpoint has one). Measured on MI355X (cfg 2): two launches fewer, W_ih streamed
 * by the layer's 64 CUs (+7 / +17 us per layer against 6.4 / 9.7 us for the whole-chip GEMMs):
 * 43.1 k vs 40.9 k queries/s with 4 batches in flight, 10 us more single-batch latency. */
int vog_ctx_set_int(vog_ctx* c, const char* name, int value);
/* round 6: 1 if the option "tx_split" (hi + lo 16-bit operands: three MFMAs per product for everything that feeds attention
 * logits - encoders, QKV projections, Q.K^T, the tails whose output is another layer's input) has kernels for this model at
 * `ncmp` videos per query (gt5-sized sequences). Set the option BEFORE vog_ctx_finalize (the remainder weights are made there). */
int vog_ctx_split_supported(const vog_ctx* c, int ncmp);
int vog_graph_destroy(vog_graph* g);

/* ---- language encoder over a group of batches ------------------------------------------------
 * The BiLSTM re-streams W_hh (16.8 MB per layer) in every one of its 2T dependent step launches:
 * at bs = 4 that is more than half of a forward's HBM traffic, for 4 of the 16 rows of the MFMA
 * tile. vog_lang_forward runs the language chain (mdl_vog.py:250-283 + :97-140) once for the
 * concatenated sentences of several in-flight batches (`lb`: B = sum of the members' B, T = max
 * of their T; the five word-level arrays are the members' arrays back to back, which is how
 * VogEngine.make_group allocates them). Rows never interact (the recurrence, the projections and
 * the argument gather are row-wise), so each member's argument vectors are what its own forward
 * computes up to fp32 summation order. Members then run vog_forward / graphs / AQL programs with
 * vog_batch.shared_lang pointing at their rows of vog_lang_outputs(). */
int64_t vog_lang_workspace_bytes(const vog_ctx* c, int B_total, int ncmp, int T);
int vog_lang_workspace_init(const vog_ctx* c, int B_total, int ncmp, int T, void* lang_ws, size_t bytes, void* stream);
/* lb: only B, ncmp, T and the five srl_* word-level pointers are read */
int vog_lang_forward(vog_ctx* c, const vog_batch* lb, void* lang_ws, size_t bytes, void* stream);
/* LSTMEncoder.forward on its own (replaces /root/reference utils/mdl_srl_utils.py:114-169; SURVEY.md 8(b) `vog_bilstm_fwd`):
 * token re-index (code/mdl_vog.py:67-95) + embedding + the packed 2-layer BiLSTM - both layers, both directions - on a
 * language workspace (vog_lang_workspace_bytes / _init). `lb` as for vog_lang_forward. Outputs, plain fp32 rows:
 * x_out [Bn, T, 2R] (zero past each sentence's length, like pad_packed_sequence) and final_hidden [Bn, 2R] =
 * [h_fwd(last valid step) || h_bwd(first step)] of the top layer (what lang_encode feeds lstm_out_feat_proj,
 * code/mdl_vog.py:262-283). The forward itself never calls this: it keeps the layer output in MFMA operand order for the
 * projection that follows (vog_bilstm_layer x 2 inside vog_forward / vog_lang_forward). */
int vog_bilstm_fwd(vog_ctx* c, const vog_batch* lb, void* lang_ws, size_t bytes, float* x_out, float* final_hidden,
                   void* stream);
/* the conversion at its end: 16-bit layer output (frag = 1: A-fragment order of the M <= 64 GEMM) -> plain fp32 rows */
int vog_lstm_out_to_f32(const void* out16, int frag, int rows_x, int rows_f, int W, vog_dtype dtype, float* x, float* fin,
                        void* stream);
/* device pointers inside lang_ws: argument vectors [B_total*nvl*nsrl, lang_enc] and the final
 * hidden projection [B_total*nvl, lang_enc] */
int vog_lang_outputs(const vog_ctx* c, int B_total, int ncmp, int T, void* lang_ws, float** lang,
                     float** final_hidden);
/* One hipGraph / one AQL program for a whole group: the group's language chain, then every
 * member's remaining forward (members[i].shared_lang must already point into lang_ws).
 * AQL rows: language rows first, then row r of every member behind one barrier packet. */
int vog_group_forward(vog_ctx* c, const vog_batch* lb, void* lang_ws, size_t lang_bytes,
                      const vog_batch* const* members, void* const* workspaces, const size_t* ws_bytes,
                      int n_members, void* stream);
int vog_group_graph_capture(vog_ctx* c, const vog_batch* lb, void* lang_ws, size_t lang_bytes,
                            const vog_batch* const* members, void* const* workspaces,
                            const size_t* ws_bytes, int n_members, void* stream, vog_graph** out);


/* HIP-event timing of `iters` back-to-back launches of ONE hot kernel of the
 * forward on `stream` (bench.py roofline leg). kernel: "mul_qkv", "mul_attn",
 * "mul_wo", "mul_ffn1", "mul_ffn2", "lin2", "obj_qkv", "obj_attn", "prop_enc".
 * Returns average microseconds per launch in *usec. */
int vog_time_kernel(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes,
                    const char* kernel, int iters, void* stream, float* usec);

#ifdef __cplusplus
}
#endif
#endif /* VOG_HIP_H */
