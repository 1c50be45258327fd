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
#include "common.h"

namespace vog {

struct LstmParams {
  const float* gxs; const unsigned short* whh; const unsigned short* h_in; unsigned short* h_out;
  float* c; unsigned short* out16; const int64_t* lens;
  int Bn, T, R, step; int out_frag, final_row0; int debug;
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

#ifdef VOG_TS_DEBUG   // scratch/ts_lstm.hip: per-wave wall-clock stamps (100 MHz) to split launch gap / in-kernel latency
__device__ unsigned long long g_ts[64][2048][4];
__device__ int g_ts_launch;
#define VOG_TS(slot) do { if (lane == 0) g_ts[p.step][(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wid][slot] = wall_clock64(); } while (0)
#else
#define VOG_TS(slot) do { } while (0)
#endif

template <typename T16>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmParams p) {
  __shared__ float red[4][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  VOG_TS(0);
  const int dir = blockIdx.y;
  const int tile = blockIdx.x;
  const int u0 = tile * 4;
  const int R = p.R;
  const int kg = (lane >> 4) * 8;
  const int ksteps = R / 32;
  // fragment-ordered weights: [dir][tile][kstep][lane][8] -> every load is one contiguous KiB
  // debug & 1 (perf experiments only): every workgroup reads tile 0's weights
  const unsigned short* wp = p.whh + (((int64_t)dir * (R / 4) + ((p.debug & 1) ? 0 : tile)) * ksteps) * 512 + lane * 8;
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
#ifdef VOG_TS_DEBUG
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      VOG_TS(3);
#endif
#pragma unroll
      for (int c = 0; c < LS_CH; ++c) acc = mfma16<T16>(fw[c], fh[c], acc);
    }
    VOG_TS(1);
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
        if (p.out_frag) {
          p.out16[frag_a(b * p.T + pos, dir * R + unit, 2 * R)] = h16;
          p.out16[frag_a(p.final_row0 + b, dir * R + unit, 2 * R)] = h16;   // last active step wins
        } else {
          p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
        }
      } else {
        p.h_out[st] = h_prev;
      }
    }
    VOG_TS(2);
  }
}

// ----------------------------------------------------------------------------
// persistent layer kernel: all T steps, both directions, one launch
// ----------------------------------------------------------------------------
struct LstmLayerParams {
  const float* gxs; const unsigned short* whh; unsigned long long* hx; unsigned int* sync;
  unsigned short* out16; const int64_t* lens; int Bn, T, R; int out_frag;
};

#define VOG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <typename T16, int KSTEPS>
__global__ __launch_bounds__(256) void lstm_layer_kernel(LstmLayerParams p) {
  constexpr int RW = KSTEPS * 32, HS_LD = RW + 8;          // +8 halfwords: rows land on different banks
  __shared__ __attribute__((aligned(16))) unsigned short hs[16 * HS_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.y, G = gridDim.x;
  const int R = p.R;
  const int tile0 = blockIdx.x * 8 + wid * 2;            // two 16-row tiles (8 units) per wave
  const int b = lane & 15, ul = lane >> 4, kg = (lane >> 4) * 8;
  const bool valid_b = b < p.Bn;
  const int len = valid_b ? (int)p.lens[b] : 0;

  // this wave's 32 rows of W_hh: registers for the whole sequence
  u16x8 wf[2][KSTEPS];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      wf[t][ks] = *reinterpret_cast<const u16x8*>(
          p.whh + ((((int64_t)dir * (R / 4) + tile0 + t) * KSTEPS + ks) * 64 + lane) * 8);

  float c[2] = {0.f, 0.f}, h_own[2] = {0.f, 0.f};
  // hand-off buffer, u64 words: [parity][dir][16 sentences][R/2]; a word = two 16-bit h values +
  // the 32-bit number of the step that produced them. The tag makes every word self-validating:
  // a consumer needs no arrival flag and no acknowledgement wait, just one (re-tried) load.
  const int64_t hx_dir = (int64_t)dir * 16 * (R / 2);
  const int64_t hx_par = (int64_t)2 * 16 * (R / 2);
  bool dead = false;

#ifdef VOG_TS_DEBUG
#define VOG_TSL(slot) do { if (tid == 0 && blockIdx.x == 3) g_ts[s][dir][slot & 3] = wall_clock64(); } while (0)
#define VOG_TSL2(slot) do { if (tid == 0 && blockIdx.x == 3) g_ts[s][2 + dir][slot & 3] = wall_clock64(); } while (0)
#else
#define VOG_TSL(slot) do { } while (0)
#define VOG_TSL2(slot) do { } while (0)
#endif
  for (int s = 0; s < p.T; ++s) {
    VOG_TSL(0);
    // input projections of this step (address-independent of everything else)
    float gin[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        gin[t][r] = valid_b ? p.gxs[(((int64_t)dir * p.T + s) * p.Bn + b) * 4 * R + (int64_t)r * R + (tile0 + t) * 4 + ul]
                            : 0.f;
    // h_{s-1} of ALL units: written by the other workgroups with write-through atomics, read
    // with L1-bypassing atomics (agent scope on both sides: no fences needed). The four waves need
    // the same Bn x R vector: the workgroup fetches it ONCE, 8 bytes per thread per sentence, into
    // LDS (measured: per-lane fragment loads, 64 dependent-ish 8-byte atomics per lane, were 7.4 of
    // a 9.6 us step) and every wave reads its MFMA B fragments from there.
    {
      const unsigned long long* hsrc = p.hx + (s & 1) * hx_par + hx_dir;
      const int items = p.Bn * (RW / 2);                   // words to fetch: sentence-major
      for (int base = tid; base < items; base += 256 * 8) {
        unsigned long long v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int it = base + j * 256;
          v[j] = it < items ? __hip_atomic_load(hsrc + (int64_t)(it / (RW / 2)) * (R / 2) + it % (RW / 2), VOG_RLX_AGENT)
                            : ((unsigned long long)(unsigned)s << 32);
        }
        // re-fetch, as ONE batch per round, the words whose producer had not stored yet (a
        // word-at-a-time retry chain cost up to 8 sequential fabric round trips per step)
        unsigned int spins = 0;
        for (;;) {
          bool stale = false;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            stale |= (base + j * 256 < items) && (unsigned int)(v[j] >> 32) != (unsigned int)s;
          if (!stale || dead) break;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int it = base + j * 256;
            if (it < items && (unsigned int)(v[j] >> 32) != (unsigned int)s)
              v[j] = __hip_atomic_load(hsrc + (int64_t)(it / (RW / 2)) * (R / 2) + it % (RW / 2), VOG_RLX_AGENT);
          }
          if ((++spins & 255u) == 0 &&
              (spins > (1u << 20) || __hip_atomic_load(p.sync + 2, VOG_RLX_AGENT) != 0)) {   // ~1 s: give up
            __hip_atomic_store(p.sync + 2, 1u, VOG_RLX_AGENT);
            dead = true;
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int it = base + j * 256;
          if (it < items)
            *reinterpret_cast<unsigned int*>(&hs[(it / (RW / 2)) * HS_LD + (it % (RW / 2)) * 2]) = (unsigned int)v[j];
        }
      }
    }
    dead = __syncthreads_or(dead ? 1 : 0) != 0;
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      u16x8 fh = {0, 0, 0, 0, 0, 0, 0, 0};
      if (valid_b) fh = *reinterpret_cast<const u16x8*>(&hs[b * HS_LD + ks * 32 + kg]);
      acc[0] = mfma16<T16>(wf[0][ks], fh, acc[0]);
      acc[1] = mfma16<T16>(wf[1][ks], fh, acc[1]);
    }
    VOG_TSL(1);
    const bool active = s < len;
    const int pos = dir == 0 ? s : len - 1 - s;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int unit = (tile0 + t) * 4 + ul;
      if (active) {
        const float gi = acc[t][0] + gin[t][0], gf = acc[t][1] + gin[t][1];
        const float gg = acc[t][2] + gin[t][2], go = acc[t][3] + gin[t][3];
        c[t] = sigm(gf) * c[t] + sigm(gi) * tanh_(gg);
        const float hn = sigm(go) * tanh_(c[t]);
        const unsigned short h16 = to16<T16>(hn);
        h_own[t] = from16<T16>(h16);
        if (p.out_frag) p.out16[frag_a(b * p.T + pos, dir * R + unit, 2 * R)] = h16;
        else p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
      }
      // publish h_s of this tile: 4 units of one sentence = one 8-byte write-through store
      const unsigned int x0 = to16<T16>(h_own[t]);
      const unsigned int x1 = __shfl(x0, b + 16), x2 = __shfl(x0, b + 32), x3 = __shfl(x0, b + 48);
      if (lane < 16 && valid_b) {
        const unsigned long long tag = (unsigned long long)(unsigned int)(s + 1) << 32;
        unsigned long long* dst = p.hx + ((s + 1) & 1) * hx_par + hx_dir + (int64_t)b * (R / 2) + (tile0 + t) * 2;
        __hip_atomic_store(dst, tag | x0 | ((unsigned long long)x1 << 16), VOG_RLX_AGENT);
        __hip_atomic_store(dst + 1, tag | x2 | ((unsigned long long)x3 << 16), VOG_RLX_AGENT);
      }
    }
    if (s + 1 == p.T) break;                             // nothing reads h_T through hx
    VOG_TSL(2);
    __syncthreads();                                     // hs is rewritten at the top of the next step
    VOG_TSL2(1);
  }
  // final hidden state rows (h of the last ACTIVE step of every sentence). A stalled hand-off
  // (a producer workgroup never became resident: more of these kernels in flight than the chip
  // holds, see vog_hip.h) must not pass for a result: poison the rows with NaN.
  if (valid_b) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int unit = (tile0 + t) * 4 + ul;
      const unsigned short v = dead ? (unsigned short)0x7fff : to16<T16>(h_own[t]);
      if (p.out_frag) p.out16[frag_a(p.Bn * p.T + b, dir * R + unit, 2 * R)] = v;
      else p.out16[((int64_t)p.Bn * p.T + b) * 2 * R + (int64_t)dir * R + unit] = v;
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

extern "C" int vog_bilstm_layer(const vog_lstm_layer_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->gxs && a->whh && a->hx && a->sync && a->out16 && a->lens && a->T > 0);
  if (!vog_bilstm_layer_supported(a->Bn, a->R))
    VOG_FAIL(-1, "persistent BiLSTM layer: unsupported Bn=%d R=%d (use vog_bilstm_step)", a->Bn, a->R);
  vog::LstmLayerParams p{a->gxs, (const unsigned short*)a->whh, (unsigned long long*)a->hx, a->sync,
                         (unsigned short*)a->out16, a->lens, a->Bn, a->T, a->R, a->out_frag};
  dim3 grid(a->R / 32, 2);
  hipStream_t st = (hipStream_t)stream;
#define VOG_LAUNCH_LAYER(KS)                                                                     \
  VOG_DISPATCH_DTYPE(a->dtype, ::vog::launch((vog::lstm_layer_kernel<T16, KS>), grid, dim3(256), 0, st, p))
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
