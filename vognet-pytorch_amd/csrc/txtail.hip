// Row-local tail of one (Rel)EncoderLayer as ONE kernel (vog_tx_tail_fwd):
//
//   x1  = LayerNorm(x + concat_h(attn_h) Wo^T)                 RelMultiHead tail + ResidualBlock
//   y   = LayerNorm(x1 + W2 relu(W1 x1 + b1) + b2)             FeedForward + ResidualBlock
//   [s  = w2 . relu(Wl y + bl) + b2 -> masks -> mdl_outs]      lin2 + score head (last mul_tx layer)
//
// (transformer_code.py:176-203, 21-31, 73-81; mdl_vog.py:224-230,675-677; mdl_conc_single.py:118-122).
// After the attention every one of these ops is row-local, so a workgroup that owns 64 token rows
// can run the whole chain without another workgroup: seven launches of the unfused path (Wo GEMM,
// LayerNorm, FFN1, FFN2, LayerNorm, lin2, score) and their fp32 / 16-bit round trips through HBM
// (tmp, x1, x1_16, ffn16, out, out16, h1: ~75 MB per cfg-2 forward) collapse into one.
//
// Mapping (gfx950, 512 threads = 8 waves, 1 workgroup per CU):
//   * the row block's 16-bit activations live in LDS ([64][K] + 8 halfwords of padding per row, so
//     the 16 lanes of one ds_read_b128 phase hit 16 different bank groups) and are the MFMA *B*
//     operand; weights are the *A* operand, streamed straight from L2 into registers in
//     fragment order (vog_pack_w_frag32: one contiguous KiB per 32x16 fragment, 4 k-steps of
//     prefetch): an operand that each wave reads exactly once must not round-trip through LDS.
//     Wave w owns output columns [w*D/8, (w+1)*D/8) for all 64 rows, so every weight fragment is
//     fetched once per workgroup.
//   * products are "swapped" (D[n][m] = W[n][:] . X[m][:]): a lane holds ONE row m and 4-column
//     strips of n, so the LayerNorm statistics are in-register sums + one cross-half shuffle + one
//     8-way LDS exchange, and the next stage's LDS operand is written with 8-byte stores.
//   * the fp32 residual stream stays in registers for both widths (d = 512, 768): the residual is the
//     initial accumulator of the Wo chain, x1 (+ b2) the initial accumulator of FFN2.
// Bound: the workgroup streams 2.75 MB (mul_tx incl. lin2) / 1.05 MB (obj_tx) of weights through
// one CU's L2 port (~64 B/clk) and issues 5376 / 1536 32x32x16 MFMAs: both ~18 us at cfg 2 -> the
// kernel sits at the CU's own MFMA/ingest balance point; what it removes is six dependent launches,
// their ramps, and every intermediate tensor.
#include "txtail_dev.h"
#include "pair_ids.h"

namespace vog {

template <typename T16, int NB, bool SCORE, int DBG = 0>
static int launch_tail(const TailParams& p, hipStream_t st) {
  constexpr int D = NB * 256, DH = D / 2;
  const size_t lds = TxTailBody<T16, F16, NB, SCORE, DBG>::lds_bytes(p.KWO);
  auto kern = tx_tail_kernel<T16, F16, NB, SCORE, DBG>;
  static bool attr_set = false;
  if (!attr_set) {
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_set = true;
  }
  if (lds > 160 * 1024) VOG_FAIL(-1, "fused encoder tail: %zu bytes of LDS needed", lds);
  const int nblk = ceil_div(p.M, 64);
  ::vog::launch(kern, dim3(p.xcds > 0 ? ceil_div(nblk, p.xcds) * 8 : nblk), dim3(512), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}

// 32-row workgroups, two per CU (TxTailBody<..., RB = 1>)
template <typename T16, int NB, bool SCORE>
static int launch_tail32(const TailParams& p, hipStream_t st) {
  using Body = TxTailBody<T16, F16, NB, SCORE, 0, 1>;
  const size_t lds = Body::lds_bytes(p.KWO);
  auto kern = tx_tail32_kernel<T16, F16, NB, SCORE>;
  static bool attr_set = false;
  if (!attr_set) {
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (lds > 160 * 1024) VOG_FAIL(-1, "fused encoder tail (32 rows): %zu bytes of LDS needed", lds);
  ::vog::launch(kern, dim3(ceil_div(p.M, 32)), dim3(512), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}

// hi + lo operands (round 6): TxTailBody<..., RB = 1, SPLIT>
template <typename T16, int NB>
static int launch_tail_split(const TailParams& p, hipStream_t st) {
  using Body = TxTailBody<T16, F16, NB, false, 0, 1, true>;
  const size_t lds = Body::lds_bytes(p.KWO);
  auto kern = tx_tail_split_kernel<T16, F16, NB>;
  static bool attr_set = false;
  if (!attr_set) {
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (lds > 160 * 1024) VOG_FAIL(-1, "fused encoder tail (hi + lo operands): %zu bytes of LDS needed", lds);
  ::vog::launch(kern, dim3(ceil_div(p.M, 32)), dim3(512), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}
const void* kid_tx_tail_split_512_f16() { return reinterpret_cast<const void*>(tx_tail_split_kernel<F16, F16, 2>); }

const void* kid_tx_tail_512(int dtype) {
  return dtype == VOG_BF16 ? reinterpret_cast<const void*>(tx_tail_kernel<BF16, F16, 2, false, 0>)
                           : reinterpret_cast<const void*>(tx_tail_kernel<F16, F16, 2, false, 0>);
}

int tx_tail_supported(int d, int dh, int kwo) {
  // (d = 768: the Wo stage walks kwo / 16 k-steps three at a time, VOG_TAIL_PFA3 - every head layout of a 768-wide model gives 768)
  return (d == 512 || d == 768) && dh == d / 2 && kwo > 0 && (kwo % 64) == 0 && kwo <= 768 && (d != 768 || (kwo % 192) == 0);
}

int64_t tx_tail_scratch_bytes(int M, int d) {
  (void)M; (void)d;
  return 0;          // the fp32 residual stream stays in registers for both supported widths
}

int tx_tail_run(const vog_tx_tail_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->attn16 && a->wo_p && a->w1_p && a->w2_p && a->ln1g && a->ln1b && a->b1 && a->b2 &&
                a->ln2g && a->ln2b && a->M > 0);
  if (!tx_tail_supported(a->d, a->dh, a->kwo))
    VOG_FAIL(-1, "fused encoder tail: unsupported shape d=%d dh=%d kwo=%d (d in {512,768}, dh = d/2, kwo %% 64 == 0)",
             a->d, a->dh, a->kwo);
  VOG_CHECK_ARG((a->residual != nullptr) != (a->res_vislang != nullptr));
  VOG_CHECK_ARG(a->y32 || a->y16 || a->score);
  TailParams p{};
  p.attn16 = (const unsigned short*)a->attn16; p.KWO = a->kwo;
  p.wo_p = (const unsigned short*)a->wo_p; p.w1_p = (const unsigned short*)a->w1_p;
  p.w2_p = (const unsigned short*)a->w2_p;
  p.residual = a->residual; p.ldr = a->ldr;
  if (a->res_vislang) {
    const vog_vislang_args* r = a->res_vislang;
    VOG_CHECK_ARG(r->vis && r->lang && r->dv + r->dl == a->d && (r->dv % 32) == 0 && (r->dl % 32) == 0);
    p.res_vis = r->vis; p.res_lang = r->lang; p.rv_nfrm = r->nfrm; p.rv_nppf = r->nppf; p.rv_nsrl = r->nsrl;
    p.rv_dv = r->dv; p.rv_dl = r->dl; p.rv_lpv = r->lang_per_vid; p.rv_ncv = r->nc_v;
  } else {
    VOG_CHECK_ARG((a->ldr % 4) == 0);
  }
  p.ln1g = a->ln1g; p.ln1b = a->ln1b; p.b1 = a->b1; p.b2 = a->b2; p.ln2g = a->ln2g; p.ln2b = a->ln2b;
  p.y32 = a->y32; p.y16 = (unsigned short*)a->y16;
  p.y16_bf16 = (a->y16_dtype < 0 ? (int)a->dtype : a->y16_dtype) == VOG_BF16;
  p.M = a->M;
  {
    // many row blocks per XCD (>= 4 rounds of workgroups): stream the activation rows past L2's retention
    // (VOG_TAIL_NT = 0 / 1 forces it off / on, perf experiments)
    static const int nt_env = perf_env("VOG_TAIL_NT") ? atoi(perf_env("VOG_TAIL_NT")) : -1;
    p.nt_rows = nt_env >= 0 ? nt_env : (a->M >= 4 * 256 * 64 ? 1 : 0);
  }
  const bool score = a->score != nullptr;
  {
    static const int xc_env = perf_env("VOG_TAIL_XCDS") ? atoi(perf_env("VOG_TAIL_XCDS")) : -1;
    p.xcds = (xc_env >= 0 && !g_pair_capture) ? xc_env : 0;
  }
  if (score) {
    VOG_CHECK_ARG(a->wl_p && a->bl && a->score->w2 && a->score->b2 && a->score->arg_msk && a->score->cmp_msk &&
                  a->score->outs && a->score->outs_eval && a->score->dh == 256 && a->head_dtype == VOG_F16);
    VOG_CHECK_ARG((int64_t)a->score->n_vid * a->score->nfrm * a->score->nsrl * a->score->nppf == a->M);
    p.wl_p = (const unsigned short*)a->wl_p; p.bl = a->bl; p.wl2 = a->score->w2; p.bl2 = a->score->b2;
    p.sc = *a->score;
  }
  if (a->attn16_lo || a->wo_p_lo || a->w1_p_lo || a->w2_p_lo || a->y16_lo) {
    // hi + lo operands (round 6): every operand of the three GEMM stages with its 16-bit remainder
    VOG_CHECK_ARG(a->attn16_lo && a->wo_p_lo && a->w1_p_lo && a->w2_p_lo && !score);
    p.attn16_lo = (const unsigned short*)a->attn16_lo; p.wo_p_lo = (const unsigned short*)a->wo_p_lo;
    p.w1_p_lo = (const unsigned short*)a->w1_p_lo; p.w2_p_lo = (const unsigned short*)a->w2_p_lo;
    p.y16_lo = (unsigned short*)a->y16_lo; p.nt_rows = 0; p.xcds = 0;
    if (a->d == 512) { VOG_DISPATCH_DTYPE(a->dtype, return (launch_tail_split<T16, 2>(p, st))); }
    else { VOG_DISPATCH_DTYPE(a->dtype, return (launch_tail_split<T16, 3>(p, st))); }
  }
#define VOG_TAIL(NBV)                                                                        \
  do {                                                                                       \
    if (score) { VOG_DISPATCH_DTYPE(a->dtype, return (launch_tail<T16, NBV, true>(p, st))); } \
    else { VOG_DISPATCH_DTYPE(a->dtype, return (launch_tail<T16, NBV, false>(p, st))); }      \
  } while (0)
  // VOG_TAIL_DEBUG (perf experiments only; results are WRONG): ablations of the bf16 kernels
  static const int dbg = perf_env("VOG_TAIL_DEBUG") ? atoi(perf_env("VOG_TAIL_DEBUG")) : 0;
  static const int dbgf = perf_env("VOG_TAIL_DEBUG_F") ? atoi(perf_env("VOG_TAIL_DEBUG_F")) : 0;
  p.dbgf = dbgf;
  if (dbg && a->dtype == VOG_BF16) {
    if (a->d == 512 && !score) {
      if (dbg == 1) return launch_tail<BF16, 2, false, 1>(p, st);
      if (dbg == 2) return launch_tail<BF16, 2, false, 2>(p, st);
      if (dbg == 3) return launch_tail<BF16, 2, false, 3>(p, st);
      if (dbg == 4) return launch_tail<BF16, 2, false, 4>(p, st);
    }
    if (a->d == 768 && score) {
      if (dbg == 1) return launch_tail<BF16, 3, true, 1>(p, st);
      if (dbg == 2) return launch_tail<BF16, 3, true, 2>(p, st);
      if (dbg == 3) return launch_tail<BF16, 3, true, 3>(p, st);
      if (dbg == 4) return launch_tail<BF16, 3, true, 4>(p, st);
    }
  }
  // VOG_TAIL_ROWS32=1 (perf experiments): the mul_tx (d = 768) tail as 32-row workgroups, two per CU. Measured in round 4
  // (profiles/round4_tail_32rows.md): 327 -> 434 us at cfg 4, cfg 2 59.0 -> 50.7 k queries/s - the kernel is bound by the
  // weight bytes a CU ingests per row, which this form doubles. Off.
  static const int rows32 = perf_env("VOG_TAIL_ROWS32") ? atoi(perf_env("VOG_TAIL_ROWS32")) : 0;
  if (rows32 && !g_pair_capture && a->d == 768) {
    if (score) { VOG_DISPATCH_DTYPE(a->dtype, return (launch_tail32<T16, 3, true>(p, st))); }
    else { VOG_DISPATCH_DTYPE(a->dtype, return (launch_tail32<T16, 3, false>(p, st))); }
  }
  if (a->d == 512) VOG_TAIL(2);
  else VOG_TAIL(3);
#undef VOG_TAIL
  return 0;
}

}  // namespace vog

extern "C" int vog_tx_tail_supported(int d, int dh, int kwo) { return vog::tx_tail_supported(d, dh, kwo); }
extern "C" int64_t vog_tx_tail_scratch_bytes(int M, int d) { return vog::tx_tail_scratch_bytes(M, d); }
extern "C" int vog_tx_tail_fwd(const vog_tx_tail_args* a, void* stream) {
  return vog::tx_tail_run(a, (hipStream_t)stream);
}

extern "C" int vog_encoder_layer_fwd(const vog_encoder_layer_args* a, void* stream) {
  VOG_CHECK_ARG(a != nullptr);
  VOG_CHECK_ARG(a->qkv.q == a->attn.q && a->qkv.k == a->attn.k && a->qkv.vt == a->attn.vt);
  VOG_CHECK_ARG(a->attn.out16 == a->tail.attn16 && a->tail.M == a->qkv.S * a->qkv.N &&
                a->tail.kwo == a->qkv.H * a->qkv.dp && a->attn.S == a->qkv.S && a->attn.N == a->qkv.N &&
                a->attn.H == a->qkv.H && a->attn.dp == a->qkv.dp && a->attn.npad == a->qkv.npad &&
                a->qkv.pl == nullptr);
  int rc = vog_qkv_proj(&a->qkv, stream);
  if (rc != 0) return rc;
  rc = vog_rel_attention_fwd(&a->attn, stream);
  if (rc != 0) return rc;
  return vog_tx_tail_fwd(&a->tail, stream);
}

extern "C" int vog_pack_w_frag32(const float* w, int64_t ld, int N, int K, void* dst_host, vog_dtype dtype) {
  VOG_CHECK_ARG(w && dst_host && N > 0 && K > 0 && (N % 32) == 0 && (K % 16) == 0 && ld >= K);
  unsigned short* dst = (unsigned short*)dst_host;
  const int ks_n = K / 16;
  for (int nb = 0; nb < N / 32; ++nb)
    for (int ks = 0; ks < ks_n; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const float* src = w + (int64_t)(nb * 32 + (lane & 31)) * ld + ks * 16 + (lane >> 5) * 8;
        unsigned short* d = dst + (((int64_t)nb * ks_n + ks) * 64 + lane) * 8;
        for (int j = 0; j < 8; ++j) {
          if (dtype == VOG_BF16) {
            unsigned int u; memcpy(&u, &src[j], 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) { d[j] = (unsigned short)((u >> 16) | 0x40); continue; }
            u += 0x7fffu + ((u >> 16) & 1u);
            d[j] = (unsigned short)(u >> 16);
          } else {
            _Float16 h = (_Float16)src[j];
            memcpy(&d[j], &h, 2);
          }
        }
      }
  return 0;
}
