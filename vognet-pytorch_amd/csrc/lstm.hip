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
#include "common.h"

namespace vog {

struct LstmParams {
  const float* gxs; const unsigned short* whh; const unsigned short* h_in; unsigned short* h_out;
  float* c; unsigned short* out16; const int64_t* lens;
  int Bn, T, R, step;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) {
  // tanh via exp of -2|x| : accurate to ~1e-7 rel, no overflow
  const float a = fabsf(x);
  const float e = __expf(-2.0f * a);
  const float t = (1.0f - e) / (1.0f + e);
  return x < 0.f ? -t : t;
}

constexpr int LS_CH = 8;      // k-steps per wave kept in registers per chunk

template <typename T16>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmParams p) {
  __shared__ float red[4][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int dir = blockIdx.y;
  const int tile = blockIdx.x;
  const int u0 = tile * 4;
  const int R = p.R;
  const int kg = (lane >> 4) * 8;
  const int ksteps = R / 32;
  // fragment-ordered weights: [dir][tile][kstep][lane][8] -> every load is one contiguous KiB
  const unsigned short* wp = p.whh + (((int64_t)dir * (R / 4) + tile) * ksteps) * 512 + lane * 8;
  const int nbt = (p.Bn + 15) / 16;
  const int unit = u0 + (lane >> 4);

  for (int bt = 0; bt < nbt; ++bt) {
    const int b = bt * 16 + (lane & 15);
    // ---- wave 0: everything the pointwise update needs is requested BEFORE the
    // matrix part, so its latency hides under the W / h loads (none of these
    // addresses depends on another load: gxs is already in step order)
    float g_in[4] = {0.f, 0.f, 0.f, 0.f};
    float c_prev = 0.f;
    unsigned short h_prev = 0;
    int len = 0;
    const bool mine = wid == 0 && unit < R && b < p.Bn;
    const int64_t st = (int64_t)b * 2 * R + (int64_t)dir * R + unit;
    if (mine) {
      len = (int)p.lens[b];
      const float* g = p.gxs + (((int64_t)dir * p.T + p.step) * p.Bn + b) * 4 * R + unit;
#pragma unroll
      for (int r = 0; r < 4; ++r) g_in[r] = g[(int64_t)r * R];
      c_prev = p.c[st];
      h_prev = p.h_in[st];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* hp = p.h_in + (int64_t)b * 2 * R + (int64_t)dir * R;   // rows < Bn16 exist
    for (int base = wid; base < ksteps; base += 4 * LS_CH) {
      u16x8 fw[LS_CH], fh[LS_CH];
#pragma unroll
      for (int c = 0; c < LS_CH; ++c) {
        const int ks = base + c * 4;
        u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        fw[c] = (ks < ksteps) ? *reinterpret_cast<const u16x8*>(wp + (int64_t)ks * 512) : z;
        fh[c] = (ks < ksteps) ? *reinterpret_cast<const u16x8*>(hp + ks * 32 + kg) : z;
      }
#pragma unroll
      for (int c = 0; c < LS_CH; ++c) acc = mfma16<T16>(fw[c], fh[c], acc);
    }
    __syncthreads();                              // red[] free (previous batch tile consumed)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][lane][r] = acc[r];
    __syncthreads();
    if (mine) {
      if (p.step < len) {
        const int pos = dir == 0 ? p.step : len - 1 - p.step;
        float gate[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          gate[r] = red[0][lane][r] + red[1][lane][r] + red[2][lane][r] + red[3][lane][r] + g_in[r];
        const float cn = sigm(gate[1]) * c_prev + sigm(gate[0]) * tanh_(gate[2]);
        const float hn = sigm(gate[3]) * tanh_(cn);
        p.c[st] = cn;
        const unsigned short h16 = to16<T16>(hn);
        p.h_out[st] = h16;
        p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
      } else {
        p.h_out[st] = h_prev;
      }
    }
  }
}

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

int lstm_step_run(const vog_lstm_step_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->gx && a->whh && a->h_in && a->h_out && a->c && a->out16 && a->lens);
  VOG_CHECK_ARG(a->Bn > 0 && a->T > 0 && a->R > 0 && (a->R % 32) == 0 && a->step >= 0 && a->step < a->T);
  VOG_CHECK_ARG(a->h_in != a->h_out);
  LstmParams p{a->gx, (const unsigned short*)a->whh, (const unsigned short*)a->h_in,
               (unsigned short*)a->h_out, a->c, (unsigned short*)a->out16, a->lens,
               a->Bn, a->T, a->R, a->step};
  dim3 grid(ceil_div(a->R, 4), 2);
  VOG_DISPATCH_DTYPE(a->dtype, hipLaunchKernelGGL((lstm_step_kernel<T16>), grid, dim3(256), 0, st, p));
  VOG_LAUNCH_CHECK();
  return 0;
}

}  // namespace vog

extern "C" int vog_bilstm_step(const vog_lstm_step_args* a, void* stream) {
  return vog::lstm_step_run(a, (hipStream_t)stream);
}

extern "C" int vog_lstm_schedule(const int64_t* lens, int32_t* rows, int Bn, int T, void* stream) {
  VOG_CHECK_ARG(lens && rows && Bn > 0 && T > 0);
  hipLaunchKernelGGL(vog::lstm_schedule_kernel, dim3(vog::ceil_div(Bn * T, 128)), dim3(128), 0,
                     (hipStream_t)stream, lens, rows, Bn, T);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_lstm_pack_whh(const float* whh_fwd, const float* whh_bwd, void* dst_host, int R,
                                 vog_dtype dtype) {
  VOG_CHECK_ARG(whh_fwd && whh_bwd && dst_host && R > 0 && (R % 32) == 0 && (R % 4) == 0);
  unsigned short* dst = (unsigned short*)dst_host;
  const int ksteps = R / 32;
  for (int dir = 0; dir < 2; ++dir) {
    const float* w = dir == 0 ? whh_fwd : whh_bwd;
    for (int tile = 0; tile < R / 4; ++tile)
      for (int ks = 0; ks < ksteps; ++ks)
        for (int lane = 0; lane < 64; ++lane) {
          const int rr = lane & 15;                           // tile row = unit_local*4 + gate
          const int64_t grow = (int64_t)(rr & 3) * R + tile * 4 + (rr >> 2);
          const float* src = w + grow * R + ks * 32 + (lane >> 4) * 8;
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
