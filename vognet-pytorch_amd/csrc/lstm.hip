// K5: one time step of a bidirectional LSTM layer with packed-sequence
// semantics (LSTMEncoder.forward utils/mdl_srl_utils.py:114-169: embedding ->
// pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence; gate
// order i,f,g,o; zero initial state).
//
// The input projections x W_ih^T + b_ih + b_hh of ALL steps are one GEMM
// (gemm.hip); this kernel is the serial part: gates = gx[pos] + h W_hh^T,
// c = sig(f) c + sig(i) tanh(g), h = sig(o) tanh(c).
//
// Regime: B*nv <= 64 sentences => a weight-streaming GEMV-like op (8 MB of
// 16-bit W_hh per direction per step, L2-resident across steps because the
// blockIdx -> rows mapping is fixed). One workgroup = 4 hidden units x 4 gates =
// one 16-row MFMA tile of W_hh for one direction; its 4 waves split K and keep
// their W fragments in registers across the batch tiles; partial sums meet in
// LDS; wave 0 applies the pointwise LSTM update. With the 16x16x32 C layout
// (row = 4*(lane>>4) + reg, col = lane&15) and rows ordered unit-major /
// gate-minor, each lane ends up with exactly (i,f,g,o) of one (unit, sentence).
//
// Packed semantics: direction 0 visits position t = step, direction 1 visits
// t = len-1-step; a sentence is active while step < len; inactive sentences keep
// (h, c) and write nothing (out16 is pre-zeroed => zeros past each length).
//
// The step is a ~2 us latency chain, so nothing on it may wait for a second
// dependent load: the input projections arrive already in (direction, step)
// order (the GEMM scatters its rows by the schedule of vog_lstm_schedule), and
// W_hh is stored in MFMA-fragment order so each wave load is one contiguous KiB.
#include <stdlib.h>
#include "lstm_dev.h"
#include "pair_ids.h"

namespace vog {

__global__ void lstm_schedule_kernel(const int64_t* __restrict__ lens, int32_t* __restrict__ rows,
                                     int Bn, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bn * T) return;
  const int b = i / T, t = i % T;
  const int len = (int)lens[b];
  rows[i] = t < len ? t * Bn + b : -1;
  // direction 1 occupies columns [4R, 8R) of a 4R-pitch buffer: the GEMM addresses
  // row*4R + col, so plane 1 (row offset T*Bn) is reached with row index T*Bn - 1 + r
  rows[Bn * T + i] = t < len ? T * Bn - 1 + (len - 1 - t) * Bn + b : -1;
}

const void* kid_lstm_layer_f16() { return reinterpret_cast<const void*>(lstm_layer_kernel<F16, 32>); }

int lstm_step_run(const vog_lstm_step_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->gx && a->whh && a->h_in && a->h_out && a->c && a->out16 && a->lens);
  VOG_CHECK_ARG(a->Bn > 0 && a->T > 0 && a->R > 0 && (a->R % 32) == 0 && a->step >= 0 && a->step < a->T);
  VOG_CHECK_ARG(a->h_in != a->h_out);
  LstmParams p{a->gx, (const unsigned short*)a->whh, (const unsigned short*)a->h_in,
               (unsigned short*)a->h_out, a->c, (unsigned short*)a->out16, a->lens,
               a->Bn, a->T, a->R, a->step, a->out_frag, a->final_row0, 0};
  static const int dbg = perf_env("VOG_LSTM_DEBUG") ? atoi(perf_env("VOG_LSTM_DEBUG")) : 0;
  p.debug = dbg;
  dim3 grid(ceil_div(a->R, 4), 2);
  VOG_DISPATCH_DTYPE(a->dtype, ::vog::launch((lstm_step_kernel<T16>), grid, dim3(256), 0, st, p));
  VOG_LAUNCH_CHECK();
  return 0;
}

}  // namespace vog

extern "C" int vog_bilstm_step(const vog_lstm_step_args* a, void* stream) {
  return vog::lstm_step_run(a, (hipStream_t)stream);
}

extern "C" int vog_bilstm_layer_supported(int Bn, int R) {
  const int ks = R / 32;
  return Bn >= 1 && Bn <= 16 && R % 32 == 0 && R / 32 <= 64 && (ks == 1 || ks == 2 || ks == 4 || ks == 32);
}

// (sentence, position) columns the layer kernel's own input projection takes (vog_lstm_layer_args.wih)
extern "C" int vog_bilstm_fused_cols(void) { return 16 * vog::LstmLayerBody<vog::F16, 32>::NCT_MAX; }

extern "C" int64_t vog_bilstm_hx_bytes(int Bn, int T, int R) {
  if (Bn <= 0 || T <= 0 || R <= 0) return -1;
  return (int64_t)T * 2 * Bn * R * 2;
}

extern "C" int vog_bilstm_layer(const vog_lstm_layer_args* a, void* stream) {
  VOG_CHECK_ARG(a && (a->gxs || a->wih || a->gx_table) && a->whh && a->hx && a->sync && a->out16 && a->lens && a->T > 0);
  VOG_CHECK_ARG(!a->gx_table || (a->tok && a->Bn * a->T <= vog::LstmLayerBody<vog::F16, 32>::TOK_MAX));
  if (!vog_bilstm_layer_supported(a->Bn, a->R))
    VOG_FAIL(-1, "persistent BiLSTM layer: unsupported Bn=%d R=%d (use vog_bilstm_step)", a->Bn, a->R);
  vog::LstmLayerParams p{a->gxs, (const unsigned short*)a->whh, (unsigned short*)a->hx, a->sync,
                         (unsigned short*)a->out16, a->lens, a->Bn, a->T, a->R, a->out_frag,
                         (const unsigned short*)a->wih, (const unsigned short*)a->xa, a->bias, a->K,
                         a->fault, a->inject_stall, 0, a->gx_table, a->tok};
  if (const char* e = vog::perf_env("VOG_LSTM_HALF_PROJ")) p.half_proj = atoi(e);
  const bool fused = a->wih != nullptr && !a->gx_table;
  if (a->gx_table) p.wih = nullptr;
  if (fused) VOG_CHECK_ARG(a->xa && a->bias && a->K > 0 && (a->K % 256) == 0 && a->Bn * a->T <= vog_bilstm_fused_cols());
  dim3 grid(a->R / 32, 2);
  hipStream_t st = (hipStream_t)stream;
  // perf experiments only: a larger LDS claim per workgroup (prices what the footprint costs the other streams' kernels)
  size_t lds_extra = 0;
  if (const char* e = vog::perf_env("VOG_LSTM_LDS_EXTRA")) lds_extra = (size_t)atoi(e);
#define VOG_LAUNCH_LAYER(KS)                                                                          \
  VOG_DISPATCH_DTYPE(a->dtype, {                                                                      \
    auto kern = vog::lstm_layer_kernel<T16, KS>;                                                      \
    using Body = vog::LstmLayerBody<T16, KS>;                                                         \
    static bool attr_set = false;                                                                     \
    if (!attr_set) {                                                                                  \
      VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));          \
      attr_set = true;                                                                                \
    }                                                                                                 \
    ::vog::launch(kern, grid, dim3(512),                                                              \
                  (fused ? Body::lds_fused(a->Bn, a->Bn * a->T) : Body::lds_plain(a->Bn)) + lds_extra, st, p); \
  })
  switch (a->R / 32) {
    case 1: VOG_LAUNCH_LAYER(1); break;
    case 2: VOG_LAUNCH_LAYER(2); break;
    case 4: VOG_LAUNCH_LAYER(4); break;
    default: VOG_LAUNCH_LAYER(32); break;
  }
#undef VOG_LAUNCH_LAYER
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_lstm_schedule(const int64_t* lens, int32_t* rows, int Bn, int T, void* stream) {
  VOG_CHECK_ARG(lens && rows && Bn > 0 && T > 0);
  ::vog::launch(vog::lstm_schedule_kernel, dim3(vog::ceil_div(Bn * T, 128)), dim3(128), 0,
                     (hipStream_t)stream, lens, rows, Bn, T);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_lstm_pack_whh(const float* whh_fwd, const float* whh_bwd, void* dst_host, int R,
                                 vog_dtype dtype) {
  return vog_lstm_pack_w(whh_fwd, whh_bwd, dst_host, R, R, dtype);
}

extern "C" int vog_lstm_pack_w(const float* whh_fwd, const float* whh_bwd, void* dst_host, int R, int K,
                               vog_dtype dtype) {
  VOG_CHECK_ARG(whh_fwd && whh_bwd && dst_host && R > 0 && (R % 4) == 0 && K > 0 && (K % 32) == 0);
  unsigned short* dst = (unsigned short*)dst_host;
  const int ksteps = K / 32;
  for (int dir = 0; dir < 2; ++dir) {
    const float* w = dir == 0 ? whh_fwd : whh_bwd;
    for (int tile = 0; tile < R / 4; ++tile)
      for (int ks = 0; ks < ksteps; ++ks)
        for (int lane = 0; lane < 64; ++lane) {
          const int rr = lane & 15;                           // tile row = unit_local*4 + gate
          const int64_t grow = (int64_t)(rr & 3) * R + tile * 4 + (rr >> 2);
          const float* src = w + grow * K + ks * 32 + (lane >> 4) * 8;
          unsigned short* d = dst + ((((int64_t)dir * (R / 4) + tile) * ksteps + ks) * 64 + lane) * 8;
          for (int j = 0; j < 8; ++j) {
            if (dtype == VOG_BF16) {
              unsigned int u; memcpy(&u, &src[j], 4);
              u += 0x7fffu + ((u >> 16) & 1u);
              d[j] = (unsigned short)(u >> 16);
            } else {
              _Float16 h = (_Float16)src[j];
              memcpy(&d[j], &h, 2);
            }
          }
        }
  }
  return 0;
}
