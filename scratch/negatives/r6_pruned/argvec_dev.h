// Argument vectors (retrieve_srl_arg_from_lang_encode mdl_vog.py:97-140): lang[b,a] =
// relu(W [full[b,cap0] || full[b,cap1]] + bias) * msk, fp32. Shared by the stand-alone kernel
// (elementwise.hip) and the tail of the language out-projection (gemm_dev.h, GemmSkinnyBody): same per-lane
// summation order and wave reduction in both, so the two forms are bit-identical.
#pragma once
#include "common.h"

namespace vog {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave: outputs o0 .. o0+3 of NR rows `ba[r]` (ba[r] < 0: skipped). 2L <= 1024, L % 4 == 0: 16 bytes
// per lane per access, every load of the wave requested before the first FMA.
// COHERENT: `full` was written by other workgroups of the SAME launch (write-through stores): its rows are
// read with L1-bypassing (sc1) loads.
typedef __attribute__((ext_vector_type(4))) float av_f32x4;
__device__ __forceinline__ float4 av_load_row(const float* row, int i, bool coherent) {
  if (!coherent) return reinterpret_cast<const float4*>(row)[i];
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, 0x7fffffff, 0x00020000);
  const av_f32x4 v = __builtin_bit_cast(av_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, i * 16, 0, 16 /* sc1 */));
  return make_float4(v[0], v[1], v[2], v[3]);
}

template <int NR, bool COHERENT = false>
__device__ __forceinline__ void argvec_rows(const float* __restrict__ full, const int64_t* __restrict__ capture,
                                            const int64_t* __restrict__ msk, const float* __restrict__ w,
                                            const float* __restrict__ bias, float* __restrict__ lang, int T, int nsrl,
                                            int L, int o0, const int (&ba)[NR], int lane, bool poison) {
  constexpr int MAXQ = 4;
  const int nq = (2 * L) >> 2, lq = L >> 2;
  float4 wv[4][MAXQ], xv[NR][MAXQ];
  float mk[NR];
  const float* x0[NR]; const float* x1[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int row = ba[r] < 0 ? 0 : ba[r];
    const int b = row / nsrl;
    int64_t c0 = capture[(int64_t)row * 2], c1 = capture[(int64_t)row * 2 + 1];
    c0 = c0 < 0 ? 0 : (c0 >= T ? T - 1 : c0);
    c1 = c1 < 0 ? 0 : (c1 >= T ? T - 1 : c1);
    x0[r] = full + ((int64_t)b * T + c0) * L;
    x1[r] = full + ((int64_t)b * T + c1) * L;
    mk[r] = (float)msk[row];
  }
#pragma unroll
  for (int it = 0; it < MAXQ; ++it) {
    const int i = lane + it * 64;
    const bool ok = i < nq;
#pragma unroll
    for (int r = 0; r < NR; ++r)
      xv[r][it] = ok ? (i < lq ? av_load_row(x0[r], i, COHERENT) : av_load_row(x1[r], i - lq, COHERENT))
                     : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      wv[k][it] = (ok && o0 + k < L) ? reinterpret_cast<const float4*>(w + (int64_t)(o0 + k) * 2 * L)[i]
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < MAXQ; ++it)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[k] += (wv[k][it].x * xv[r][it].x + wv[k][it].y * xv[r][it].y) + (wv[k][it].z * xv[r][it].z + wv[k][it].w * xv[r][it].w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = wave_sum(acc[k]);
      if (lane == 0 && o0 + k < L && ba[r] >= 0)
        lang[(int64_t)ba[r] * L + o0 + k] = poison ? __builtin_nanf("") : relu_nan(v + bias[o0 + k]) * mk[r];
    }
  }
}

}  // namespace vog
