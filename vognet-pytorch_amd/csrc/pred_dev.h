// Per-frame arg-max + box gather of the prediction head (Evaluator*.get_out_results_boxes,
// code/eval_vsrl_corr.py:162-424) as a device function of the work-item index: the stand-alone kernel
// (elementwise.hip) and the tail of the last mul_tx encoder-layer tail (txtail_dev.h: the workgroup that
// finishes last runs the head for the whole batch) share it, so both forms are bit-identical.
#pragma once
#include "common.h"

namespace vog {

// COHERENT: outs_eval was written by other workgroups of the SAME launch (write-through stores): L1/L2-bypassing loads
template <bool COHERENT>
__device__ __forceinline__ float pred_ld(const float* base, int64_t i) {
  if (!COHERENT) return base[i];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(i * 4), 0, 16 /* sc1 */));
}

template <bool COHERENT>
__device__ __forceinline__ void pred_item(const vog_pred_args& a, int64_t rec_bytes, int i) {

  // one thread per (query, arg, frame, video): arg-max over the proposals of that video's frame
  // and the box gather; the thread of video 0 also does the pred_cmp arg-max over the videos
  // (it re-reads the ncmp x nppf0 scores: all addresses are known up front, so the kernel is two
  // dependent memory levels deep instead of 2 x ncmp)
  const int per_q = a.nsrl * a.nfrm0 * a.ncmp;
  if (i >= a.B * per_q) return;
  const int b = i / per_q, r = i % per_q;
  const int arg = r / (a.nfrm0 * a.ncmp), f = (r / a.ncmp) % a.nfrm0, c = r % a.ncmp;
  const int npv = a.nfrm0 * a.nppf0;
  unsigned char* rec = reinterpret_cast<unsigned char*>(a.rec) + (int64_t)b * rec_bytes;
  float* boxes = reinterpret_cast<float*>(rec);
  float* scores = boxes + (int64_t)a.nsrl * a.ncmp * a.nfrm0 * 7;
  int64_t* idx = reinterpret_cast<int64_t*>(rec + (int64_t)a.nsrl * a.ncmp * a.nfrm0 * 8 * 4);
  auto first_prop = [&](int cc, int64_t* e0, int64_t* p0) {   // in outs_eval / in props
    if (a.conc_type == VOG_CONC_SPAT) {
      const int r0 = (f * a.ncmp + cc) * a.nppf0;
      *e0 = ((int64_t)b * a.nsrl + arg) * ((int64_t)a.ncmp * npv) + r0;
      *p0 = (int64_t)b * a.ncmp * npv + r0;
    } else if (a.conc_type == VOG_CONC_TEMP) {
      const int r0 = (cc * a.nfrm0 + f) * a.nppf0;
      *e0 = ((int64_t)b * a.nsrl + arg) * ((int64_t)a.ncmp * npv) + r0;
      *p0 = (int64_t)b * a.ncmp * npv + r0;
    } else {
      *e0 = (((int64_t)b * a.ncmp + cc) * a.nsrl + arg) * npv + (int64_t)f * a.nppf0;
      *p0 = ((int64_t)b * a.ncmp + cc) * npv + (int64_t)f * a.nppf0;
    }
  };
  int64_t e0, p0;
  first_prop(c, &e0, &p0);
  float best = pred_ld<COHERENT>(a.outs_eval, e0);
  int bi = 0;
  for (int k = 1; k < a.nppf0; ++k) {
    const float v = pred_ld<COHERENT>(a.outs_eval, e0 + k);
    if (v > best) { best = v; bi = k; }          // first maximum wins (torch.max on CPU)
  }
  const int64_t o = ((int64_t)arg * a.ncmp + c) * a.nfrm0 + f;
  const float* pr = a.props + (p0 + bi) * 7;
#pragma unroll
  for (int k = 0; k < 7; ++k) boxes[o * 7 + k] = pr[k];
  scores[o] = best;
  if (c != 0) return;
  int64_t out = 0;
  if (a.conc_type == VOG_CONC_SPAT) {
    float best_c = best;                         // video 0; first maximum over the videos
    for (int cc = 1; cc < a.ncmp; ++cc) {
      int64_t e1, p1;
      first_prop(cc, &e1, &p1);
      float bc = pred_ld<COHERENT>(a.outs_eval, e1);
      for (int k = 1; k < a.nppf0; ++k) bc = fmaxf(bc, pred_ld<COHERENT>(a.outs_eval, e1 + k));
      if (bc > best_c) { best_c = bc; out = cc; }
    }
  } else if (a.conc_type == VOG_CONC_SEP) {
    float bf = a.fin_scores[(int64_t)b * a.ncmp];
    for (int cc = 1; cc < a.ncmp; ++cc) {
      const float v = a.fin_scores[(int64_t)b * a.ncmp + cc];
      if (v > bf) { bf = v; out = cc; }
    }
  }
  idx[(int64_t)arg * a.nfrm0 + f] = out;
}

}  // namespace vog
