// Proposal / segment feature encoders + concat_prop_seg_feats as ONE kernel (vog_vis_encode):
//
//   prop_seg[r, :Np]  = relu(W_p . pad_region_feature[r] + b_p)                 (prop_feats_encode mdl_vog.py:291-301)
//   prop_seg[r, Np:]  = relu(W_s . seg_feature_for_frms[r / nppf0] + b_s)       (seg_feats_encode :303-314 +
//                                                                                concat_prop_seg_feats
//                                                                                mdl_conc_single.py:51-66)
//
// The unfused path needed four launches (fp32 -> 16-bit cast of the 8.7 MB of raw features, two
// split-K GEMMs writing 16 fp32 slabs, a finishing pass) and moved the features three times. Here the
// raw fp32 rows are read ONCE, rounded to the MFMA operand type in registers, and the result is
// written straight into the [rows, d_obj] matrix (fp32 + the 16-bit copy obj_tx / mul_tx consume).
//
// Work item = (16-row tile, 32 output columns); its 8 waves split K (v_mfma_f32_16x16x32, activations
// = A operand straight from the fp32 rows, weights = B operand in the fragment order of
// vog_pack_w_frag: one contiguous KiB per load), partial sums meet in LDS. The eight column slices
// of a row tile run on the same XCD (block b is observed on XCD b % 8), so the 128-192 KB of fp32
// features of the tile come from HBM once and from that L2 three times. HBM-bound by construction:
// 8.7 MB of features + 2.6 MB of weights per cfg-2 forward, 240 workgroups.
#include "visenc_dev.h"
#include "pair_ids.h"

namespace vog {

// Segment rows are replicated over the nppf0 proposals of their frame (concat_prop_seg_feats,
// mdl_conc_single.py:51-66). With 100 proposals per frame the replication is 25 MB of stores that the lean
// form left to the 6 workgroups owning the segment rows, one 2- or 4-byte store per lane and replica:
// ~100 us after every other workgroup had finished (157 us for the whole kernel at p100). Those
// workgroups now write replica 0 only and this kernel copies it to the other nppf0 - 1 rows, 16 bytes per
// lane, on the whole chip.
__global__ __launch_bounds__(256) void seg_replicate_kernel(float* c32, unsigned short* c16, int64_t ldc, int col0, int ncol,
                                                            int rows, int rep) {
  const int per_row = ncol >> 2;                              // 4 columns per thread
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)rows * (rep - 1) * per_row;
  if (i >= total) return;
  const int c4 = (int)(i % per_row);
  const int64_t rj = i / per_row;
  const int j = (int)(rj % (rep - 1)) + 1, r = (int)(rj / (rep - 1));
  const int64_t src = (int64_t)r * rep * ldc + col0 + c4 * 4, dst = ((int64_t)r * rep + j) * ldc + col0 + c4 * 4;
  if (c32) *reinterpret_cast<float4*>(c32 + dst) = *reinterpret_cast<const float4*>(c32 + src);
  if (c16) *reinterpret_cast<u16x4*>(c16 + dst) = *reinterpret_cast<const u16x4*>(c16 + src);
}

const void* kid_vis_enc_f16() { return reinterpret_cast<const void*>(vis_enc_kernel<F16>); }
const void* kid_vis_enc_stream_f16() { return reinterpret_cast<const void*>(vis_enc_stream_kernel<F16>); }
const void* kid_vis_enc_stream_split_f16() { return reinterpret_cast<const void*>(vis_enc_stream_kernel<F16, true>); }

int vis_encode_supported(int prop_dim, int seg_dim, int prop_enc, int seg_enc) {
  return (prop_dim % 256) == 0 && (seg_dim % 256) == 0 && (prop_enc % 32) == 0 && (seg_enc % 32) == 0 &&
         prop_enc <= 256 && seg_enc <= 256;
}

int vis_encode_run(const vog_visenc_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->prop && a->seg && a->w_prop_f && a->w_seg_f && a->b_prop && a->b_seg && (a->c32 || a->c16));
  VOG_CHECK_ARG(a->n_prop_rows > 0 && a->nppf0 > 0 && (a->n_prop_rows % a->nppf0) == 0);
  if (!vis_encode_supported(a->prop_dim, a->seg_dim, a->prop_enc, a->seg_enc))
    VOG_FAIL(-1, "fused encoders: unsupported dims (feature dims %% 256, encode sizes %% 32 and <= 256)");
  VisEncParams p{};
  p.p[0] = VisEncProb{a->prop, (const unsigned short*)a->w_prop_f, a->b_prop, a->n_prop_rows, a->prop_enc,
                      a->prop_dim, 1, 0, (const unsigned short*)a->w_prop_f_lo};
  p.p[1] = VisEncProb{a->seg, (const unsigned short*)a->w_seg_f, a->b_seg, a->n_prop_rows / a->nppf0, a->seg_enc,
                      a->seg_dim, a->nppf0, a->prop_enc, (const unsigned short*)a->w_seg_f_lo};
  const bool split = a->w_prop_f_lo || a->w_seg_f_lo || a->c16_lo;     // hi + lo operands (round 6): the stream form only
  if (split) VOG_CHECK_ARG(a->w_prop_f_lo && a->w_seg_f_lo && a->c16_lo && a->c16 && a->lean && !a->defer_replicas);
  p.c16_lo = (unsigned short*)a->c16_lo;
  p.tiles0 = ceil_div(p.p[0].M, 16);
  p.tiles_all = p.tiles0 + ceil_div(p.p[1].M, 16);
  p.c32 = a->c32; p.c16 = (unsigned short*)a->c16; p.ldc = a->ldc;
  p.c16_bf16 = a->c16_dtype == VOG_BF16;
  if (a->lean) {
    const int nb = ceil_div(p.tiles0, 4) + ceil_div(p.tiles_all - p.tiles0, 4);
    // many replicas per segment row (p100: 100): the encoder kernel writes replica 0, a copy kernel the rest
    const bool can_copy = (p.p[1].N % 4) == 0 && (p.p[1].col0 % 4) == 0 && (p.ldc % 4) == 0;
    if (a->defer_replicas && !can_copy) VOG_FAIL(-1, "vog_vis_encode: defer_replicas needs encode sizes and ldc %% 4 == 0");
    const bool split_rep = !a->defer_replicas && p.p[1].rep > 16 && can_copy;
    p.rep_first_only = (split_rep || a->defer_replicas) ? 1 : 0;
    // the stream form (round 5; visenc_dev.h): 64 rows x 128 columns per workgroup, K chunks of 128, two register sets of fp32 row
    // pieces + weight fragments in flight. cfg 2 pair launch with BiLSTM layer 0: 36.0 -> 32.4 us; p100: 57.6 -> 45 us alone.
    // (Round 4's lean form and the 128-row x 256-column wide form - measured 112 us against 41.5 at p100 - were removed in round 6:
    // scratch/negatives/r6_pruned/.)
    if (split) {
      if (split_rep) VOG_FAIL(-1, "vog_vis_encode with hi + lo operands: more than 16 replicas per segment row are not supported");
      auto launch_split = [&](auto tag) {
        using T16 = decltype(tag);
        auto kern = vis_enc_stream_kernel<T16, true>;
        static bool attr = false;
        if (!attr) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)VisEncStreamBody<T16, VOG_VS_DEPTH, true>::LDS);
          attr = true;
        }
        ::vog::launch(kern, dim3(ceil_div(nb, 8) * 16), dim3(512), VisEncStreamBody<T16, VOG_VS_DEPTH, true>::LDS, st, p);
      };
      if (a->dtype == VOG_BF16) launch_split(BF16{}); else launch_split(F16{});
      VOG_LAUNCH_CHECK();
      return 0;
    }
    VOG_DISPATCH_DTYPE(a->dtype, ::vog::launch((vis_enc_stream_kernel<T16>), dim3(ceil_div(nb, 8) * 16), dim3(512),
                                               VisEncStreamBody<T16>::LDS, st, p));
    VOG_LAUNCH_CHECK();
    if (split_rep) {
      const int64_t total = (int64_t)p.p[1].M * (p.p[1].rep - 1) * (p.p[1].N / 4);
      ::vog::launch(seg_replicate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.c32, p.c16, p.ldc,
                    p.p[1].col0, p.p[1].N, p.p[1].M, p.p[1].rep);
      VOG_LAUNCH_CHECK();
    }
    return 0;
  }
  const int groups = ceil_div(p.tiles_all, 8);
  VOG_DISPATCH_DTYPE(a->dtype, ::vog::launch((vis_enc_kernel<T16>), dim3(groups * 8 * 8), dim3(512), VisEncBody<T16>::LDS, st, p));
  VOG_LAUNCH_CHECK();
  return 0;
}

}  // namespace vog

extern "C" int vog_vis_encode_supported(int prop_dim, int seg_dim, int prop_enc, int seg_enc) {
  return vog::vis_encode_supported(prop_dim, seg_dim, prop_enc, seg_enc);
}
extern "C" int vog_seg_replicate(const vog_visenc_args* a, void* stream) {
  VOG_CHECK_ARG(a && (a->c32 || a->c16) && a->n_prop_rows > 0 && a->nppf0 > 0 && (a->n_prop_rows % a->nppf0) == 0);
  VOG_CHECK_ARG((a->seg_enc % 4) == 0 && (a->prop_enc % 4) == 0 && (a->ldc % 4) == 0);
  if (a->nppf0 == 1) return 0;
  const int rows = a->n_prop_rows / a->nppf0;
  const int64_t total = (int64_t)rows * (a->nppf0 - 1) * (a->seg_enc / 4);
  ::vog::launch(vog::seg_replicate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a->c32,
                (unsigned short*)a->c16, a->ldc, a->prop_enc, a->seg_enc, rows, a->nppf0);
  VOG_LAUNCH_CHECK();
  return 0;
}
extern "C" int vog_vis_encode(const vog_visenc_args* a, void* stream) {
  return vog::vis_encode_run(a, (hipStream_t)stream);
}
