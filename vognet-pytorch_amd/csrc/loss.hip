// Validation / training loss on the device: LossB_TEMP / LossB_SPAT
// (code/mdl_conc_single.py:180-433), LossB_SEP (code/mdl_conc_sep.py:220-447) with the IoU targets of
// utils/box_utils.py:61-118, as two small launches on the batch's own device tensors.
//
// The reference materialises overlaps[B, NP, 100], gathers the <= 4 ground-truth boxes of every
// (query, argument) out of it and thresholds; only those gathered entries are ever used, so here one
// thread owns one (query, video-slot, argument, proposal) element, computes the <= nbox IoUs it
// needs on the fly (proposal row, gt rows, the frame / padding mask bytes), the target, the BCE term
// and the selection mask. Block partials are written in block order and summed by ONE block in a fixed
// order (no atomics: the result is bit-reproducible). fp32 throughout; HBM/latency bound (KB-MB).
#include "common.h"

namespace vog {

struct LossParams {
  vog_loss_args a;
  int nvo, NPo;          // mdl_outs is [B, nvo, nsrl, NPo]
  int64_t total;         // elements
  int blocks;
};

__device__ __forceinline__ float bce_logits(float x, float t) {
  return fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
}

// IoU * mask with the reference's conventions (+1 pixel, 0 for an empty gt box, -1 for an empty proposal)
__device__ __forceinline__ float iou_masked(const float* p, const float* g, float mask) {
  const float ax = p[2] - p[0] + 1.f, ay = p[3] - p[1] + 1.f;
  const float gx = g[2] - g[0] + 1.f, gy = g[3] - g[1] + 1.f;
  float iw = fminf(p[2], g[2]) - fmaxf(p[0], g[0]) + 1.f;
  float ih = fminf(p[3], g[3]) - fmaxf(p[1], g[1]) + 1.f;
  iw = iw < 0.f ? 0.f : iw;
  ih = ih < 0.f ? 0.f : ih;
  float ov = iw * ih / (ax * ay + gx * gy - iw * ih);
  ov *= mask;
  if (gx == 1.f && gy == 1.f) ov = 0.f;
  if (ax == 1.f && ay == 1.f) ov = -1.f;
  return ov;
}

// One element (query b, output video slot vo, argument arg, proposal r) of mdl_outs: its logit, its IoU
// target, the selection mask bm (an element enters the masked mean iff bm != 0) and the video mask cm
// (sep multiplies the BCE term by it).
struct LossElem { float x, tgt, bm, cm; bool sep; };
__device__ __forceinline__ LossElem loss_elem(const LossParams& q, int64_t i) {
  const vog_loss_args& a = q.a;
  LossElem e;
  e.sep = a.conc_type == VOG_CONC_SEP;
  const bool sep = e.sep;
  const int r = (int)(i % q.NPo);
  int64_t t = i / q.NPo;
  const int arg = (int)(t % a.nsrl); t /= a.nsrl;
  const int vo = (int)(t % q.nvo);
  const int b = (int)(t / q.nvo);
  // video slot of the proposal, rows of this (query[, video])
  int vid;
  if (sep) vid = vo;
  else if (a.conc_type == VOG_CONC_TEMP) vid = r / (q.NPo / a.ncmp);
  else vid = (r / a.nppf0) % a.ncmp;
  const int64_t pv = sep ? (int64_t)b * a.ncmp + vo : b;          // index of the [NP, ...] block
  const float* prop = a.pad_proposals + (pv * q.NPo + r) * 7;
  const float* gts = a.pad_gt_bboxs + pv * a.G * 5;
  const unsigned char* fm = a.pad_frm_mask + (pv * q.NPo + r) * (int64_t)a.G;
  const unsigned char pm = a.pad_pnt_mask[pv * q.NPo + r];
  const int lv = a.nv > 1 ? vo : 0;                                // language / annotation copy
  const int64_t sb = (((int64_t)b * a.nv + lv) * a.nsrl + arg) * a.nbox;
  const bool on_target = vid == (int)a.target_cmp[b];
  float best = -3.0e38f;
  for (int k = 0; k < a.nbox; ++k) {
    const int64_t gi = a.srl_boxes[sb + k];
    float ov = iou_masked(prop, gts + gi * 5, (float)((fm[gi] | pm) != 0));
    ov *= on_target ? 1.f : 0.f;
    ov *= (float)a.srl_boxes_lens[sb + k];
    best = fmaxf(best, ov);
  }
  e.tgt = best > 0.5f ? 1.f : 0.f;
  e.x = a.mdl_outs[i];
  e.cm = (float)a.num_cmp_msk[(int64_t)b * a.ncmp + vid];
  e.bm = sep ? e.cm : (float)a.srl_arg_boxes_mask[((int64_t)b * a.nv + lv) * a.nsrl + arg] * e.cm;
  return e;
}

__global__ __launch_bounds__(256) void loss_partial_kernel(LossParams q) {
  __shared__ float red[3][4];
  const vog_loss_args& a = q.a;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float s_sel = 0.f, c_sel = 0.f, s_all = 0.f;
  if (i < q.total) {
    const LossElem e = loss_elem(q, i);
    float l = bce_logits(e.x, e.tgt);
    if (e.sep) l *= e.bm;
    s_all = l;
    if (e.bm != 0.f) { s_sel = l; c_sel = 1.f; }
  }
  // block reduction in a fixed order
  float v[3] = {s_sel, c_sel, s_all};
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_xor(v[j], o);
    if (lane == 0) red[j][wid] = v[j];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int j = threadIdx.x;
    a.scratch[(int64_t)blockIdx.x * 3 + j] = (red[j][0] + red[j][1]) + (red[j][2] + red[j][3]);
  }
}

__global__ __launch_bounds__(256) void loss_finish_kernel(LossParams q) {
  __shared__ float red[6][4];
  const vog_loss_args& a = q.a;
  const int tid = threadIdx.x;
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // s_sel, c_sel, s_all, abm_max, verb_sum, verb_cnt
  for (int k = tid; k < q.blocks; k += 256) {
    v[0] += a.scratch[(int64_t)k * 3]; v[1] += a.scratch[(int64_t)k * 3 + 1]; v[2] += a.scratch[(int64_t)k * 3 + 2];
  }
  const int nabm = a.B * a.nv * a.nsrl;
  for (int k = tid; k < nabm; k += 256) v[3] = fmaxf(v[3], (float)a.srl_arg_boxes_mask[k]);
  if (a.conc_type == VOG_CONC_SEP && a.vidf_outs) {
    for (int k = tid; k < a.B * a.ncmp; k += 256) {
      float m = 0.f;
      for (int j = 0; j < a.ncmp; ++j) m += (float)a.verb_cross_cmp_msk[(int64_t)k * a.ncmp + j];
      if (m > 0.f) { v[4] += bce_logits(a.vidf_outs[k], (float)a.verb_cmp[k]); v[5] += 1.f; }
    }
  }
  const int lane = tid & 63, wid = tid >> 6;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float t = __shfl_xor(v[j], o);
      v[j] = j == 3 ? fmaxf(v[j], t) : v[j] + t;
    }
    if (lane == 0) red[j][wid] = v[j];
  }
  __syncthreads();
  if (tid == 0) {
    float r[6];
    for (int j = 0; j < 6; ++j)
      r[j] = j == 3 ? fmaxf(fmaxf(red[j][0], red[j][1]), fmaxf(red[j][2], red[j][3]))
                    : (red[j][0] + red[j][1]) + (red[j][2] + red[j][3]);
    // masked_select(...).mean() when any argument has boxes, else the plain mean (reference quirk kept)
    const float mean = r[3] > 0.f ? r[0] / r[1] : r[2] / (float)q.total;
    const float mdl = mean * (float)q.NPo * a.loss_lambda;
    a.out[0] = mdl;
    a.out[1] = mdl;
    a.out[2] = (a.conc_type == VOG_CONC_SEP && a.vidf_outs) ? r[4] / r[5] * a.loss_lambda : 0.f;
    a.out[3] = r[3] > 0.f ? r[1] : (float)q.total;      // elements in the mean (vog_loss_bwd divides by it)
    a.out[4] = r[3] > 0.f ? 1.f : 0.f;                   // 1: masked mean, 0: plain mean over everything
    a.out[5] = r[5];                                     // rows in the verb-loss mean
  }
}

// d loss / d mdl_outs (and d verb_loss / d vidf_outs): the masked mean of BCE-with-logits has the gradient
// (sigmoid(x) - target) [* cm for sep] * NP * lambda / n for the elements in the mean, 0 elsewhere - the
// first link of the training path; reads n from the out[] of vog_loss_fwd.
__global__ __launch_bounds__(256) void loss_bwd_kernel(LossParams q, float* g_outs, float* g_vidf) {
  const vog_loss_args& a = q.a;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < q.total) {
    const LossElem e = loss_elem(q, i);
    const bool masked = a.out[4] != 0.f;
    const bool in_mean = masked ? e.bm != 0.f : true;
    const float sg = 1.0f / (1.0f + __expf(-e.x));
    const float scale = (float)q.NPo * a.loss_lambda / a.out[3];
    g_outs[i] = in_mean ? (sg - e.tgt) * (e.sep ? e.bm : 1.f) * scale : 0.f;
  }
  if (g_vidf && a.conc_type == VOG_CONC_SEP && a.vidf_outs && i < (int64_t)a.B * a.ncmp) {
    float m = 0.f;
    for (int j = 0; j < a.ncmp; ++j) m += (float)a.verb_cross_cmp_msk[i * a.ncmp + j];
    const float sg = 1.0f / (1.0f + __expf(-a.vidf_outs[i]));
    g_vidf[i] = m > 0.f ? (sg - (float)a.verb_cmp[i]) * a.loss_lambda / a.out[5] : 0.f;
  }
}

}  // namespace vog

extern "C" int64_t vog_loss_scratch_bytes(const vog_loss_args* a) {
  if (!a) return -1;
  const bool sep = a->conc_type == VOG_CONC_SEP;
  const int64_t total = (int64_t)a->B * (sep ? a->ncmp : 1) * a->nsrl * a->NP;
  return ((total + 255) / 256) * 3 * (int64_t)sizeof(float);
}

extern "C" int vog_loss_fwd(const vog_loss_args* a, void* stream) {
  using namespace vog;
  VOG_CHECK_ARG(a && a->mdl_outs && a->pad_proposals && a->pad_gt_bboxs && a->pad_frm_mask && a->pad_pnt_mask &&
                a->srl_boxes && a->srl_boxes_lens && a->srl_arg_boxes_mask && a->target_cmp && a->num_cmp_msk &&
                a->out && a->scratch);
  VOG_CHECK_ARG(a->B > 0 && a->ncmp > 0 && a->nv > 0 && a->nsrl > 0 && a->nbox > 0 && a->NP > 0 && a->G > 0 && a->nppf0 > 0);
  const bool sep = a->conc_type == VOG_CONC_SEP;
  VOG_CHECK_ARG(!sep || !a->vidf_outs || (a->verb_cmp && a->verb_cross_cmp_msk));
  VOG_CHECK_ARG(sep || (a->nv == 1 && (a->NP % a->ncmp) == 0));
  VOG_CHECK_ARG(!sep || a->nv == 1 || a->nv == a->ncmp);
  LossParams q{};
  q.a = *a;
  q.nvo = sep ? a->ncmp : 1;
  q.NPo = a->NP;
  q.total = (int64_t)a->B * q.nvo * a->nsrl * a->NP;
  q.blocks = (int)((q.total + 255) / 256);
  ::vog::launch(loss_partial_kernel, dim3(q.blocks), dim3(256), 0, (hipStream_t)stream, q);
  VOG_LAUNCH_CHECK();
  ::vog::launch(loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_loss_bwd(const vog_loss_args* a, float* grad_mdl_outs, float* grad_vidf_outs, void* stream) {
  using namespace vog;
  VOG_CHECK_ARG(a && a->mdl_outs && a->out && grad_mdl_outs && a->pad_proposals && a->pad_gt_bboxs &&
                a->pad_frm_mask && a->pad_pnt_mask && a->srl_boxes && a->srl_boxes_lens && a->srl_arg_boxes_mask &&
                a->target_cmp && a->num_cmp_msk);
  const bool sep = a->conc_type == VOG_CONC_SEP;
  VOG_CHECK_ARG(!grad_vidf_outs || (sep && a->vidf_outs && a->verb_cmp && a->verb_cross_cmp_msk));
  LossParams q{};
  q.a = *a;
  q.nvo = sep ? a->ncmp : 1;
  q.NPo = a->NP;
  q.total = (int64_t)a->B * q.nvo * a->nsrl * a->NP;
  q.blocks = (int)((q.total + 255) / 256);
  ::vog::launch(loss_bwd_kernel, dim3(q.blocks), dim3(256), 0, (hipStream_t)stream, q, grad_mdl_outs, grad_vidf_outs);
  VOG_LAUNCH_CHECK();
  return 0;
}
