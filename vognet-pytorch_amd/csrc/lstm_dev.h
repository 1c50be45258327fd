// Device side of lstm.hip (kernel bodies; also included by pair.hip, which fuses two bodies into one launch).
#pragma once
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
// persistent layer kernel: shader-clock stamps of wave 0 of every workgroup, [wg][step (T = kernel-level row)][slot]
__device__ unsigned long long g_tsl[64][24][8];
#define VOG_TSL(step, slot) do { if (tid == 0) g_tsl[cx.by * cx.gx + cx.bx][step][slot] = wall_clock64(); } while (0)
#else
#define VOG_TS(slot) do { } while (0)
#define VOG_TSL(step, slot) do { } while (0)
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
  // fused input projection (wih != nullptr; gxs unused): x W_ih^T + bias for THIS workgroup's gate
  // rows is computed in the prologue. wih: [dir][unit/4][K/32][lane][8] (vog_lstm_pack_w), xa: the
  // layer input [Bn*T rows (b*T + t), K] in A-fragment order, bias: [2][4R] (b_ih + b_hh).
  const unsigned short* wih; const unsigned short* xa; const float* bias; int K;
};

#define VOG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Body form (common.h): 512 threads = 8 waves, ONE 16-row tile (4 units x 4 gates) of W_hh per wave,
// i.e. 128 registers of weights per lane - the kernel fits the 256-register budget of a 512-thread
// workgroup and can therefore share a launch with the other 512-thread bodies (pair.hip). (The first
// form, 4 waves x 2 tiles, held 256 registers of weights per lane: nothing else could live beside it.)
template <typename T16, int KSTEPS>
struct LstmLayerBody {
  using Params = LstmLayerParams;
  static constexpr int THREADS = 512;
  static constexpr int RW = KSTEPS * 32, HS_LD = RW + 8;   // +8 halfwords: rows land on different banks
  static constexpr size_t LDS = (size_t)16 * HS_LD * 2;
  // fused input projection: + gates of every (sentence, position) for the 128 gate rows of the
  // workgroup ([8 waves][<= 64 columns][4 units][4 gates] fp32, 80-byte column pitch) + one K chunk
  // (256) of the layer input in fragment order ([<= 4 column tiles][8 k-steps][64 lanes][16 B]), double buffered
  static constexpr int GX_PITCH = 20;                      // floats per column (16 + 4: spreads the banks)
  static constexpr size_t LDS_HS = ((size_t)16 * HS_LD * 2 + 15) / 16 * 16;
  static constexpr size_t LDS_FUSED = LDS_HS + (size_t)8 * 64 * GX_PITCH * 4 + (size_t)2 * 4 * 8 * 1024;   // upper bound (4 column tiles)
  static constexpr size_t lds_fused(int ncols) { return LDS_HS + (size_t)8 * 64 * GX_PITCH * 4 + (size_t)2 * ((ncols + 15) / 16) * 8 * 1024; }

  static __device__ __forceinline__ void run(const LstmLayerParams& p, const BlockCtx& cx, unsigned char* smem) {
    unsigned short* hs = reinterpret_cast<unsigned short*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = cx.by;
    const int R = p.R;
    const int tile0 = cx.bx * 8 + wid;                     // one 16-row tile (4 units) per wave
    const int b = lane & 15, ul = lane >> 4, kg = (lane >> 4) * 8;
    const bool valid_b = b < p.Bn;
    const int len = valid_b ? (int)p.lens[b] : 0;

    float c = 0.f, h_own = 0.f;
    VOG_TSL(p.T, 0);
    // ---- fused input projection (replaces the separate x W_ih^T GEMM launch and its [2][T][Bn][4R]
    // fp32 round trip): every wave computes the gates of ITS 16 rows for all Bn*T (sentence, position)
    // columns: W_ih rows stream once through registers (K in chunks of 256), the layer input is
    // staged per chunk in LDS and shared by the 8 waves; the result stays in LDS for the recurrence.
    const bool fused = p.wih != nullptr;
    float* gxl = reinterpret_cast<float*>(smem + LDS_HS) + (size_t)wid * 64 * GX_PITCH;
    if (fused) {
      unsigned char* xch = smem + LDS_HS + (size_t)8 * 64 * GX_PITCH * 4;
      const int ncols = p.Bn * p.T, nct = (ncols + 15) >> 4;          // <= 4 column tiles
      const int ksteps = p.K >> 5, nchunk = ksteps >> 3;               // K % 256 == 0
      f32x4 ga[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) ga[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      const u16x8* wsrc = reinterpret_cast<const u16x8*>(p.wih) + (((int64_t)dir * (R / 4) + tile0) * ksteps) * 64 + lane;
      const uint4* xsrc = reinterpret_cast<const uint4*>(p.xa);
      // chunk c of column tile ct: k-steps [8c, 8c+8) = 8 KiB contiguous at ((ct*ksteps + 8c)*64) uint4.
      // Pipeline: the layer input goes global -> LDS by LDS-DMA (global_load_lds: no VGPR round trip, so
      // the compiler cannot serialise it behind the weight loads; first form, through registers: every
      // chunk waited vmcnt(0), 4 us per chunk), double buffered; the weight fragments of the NEXT
      // chunk are in flight in a second register set. Two chunks = 80 KB per workgroup are always
      // outstanding; per chunk: one counted wait, barrier, 8 x nct MFMAs, barrier.
      // (two images of nct x 8 KiB: the launcher sizes the LDS for the column tiles that exist, so that
      // 38 KB of the CU stay free at cfg 2 - enough for a workgroup of the LDS-DMA GEMM to share the CU
      // with this one, which spends most of its life waiting for hand-offs)
      unsigned char* xbuf[2] = {xch, xch + nct * 8 * 1024};
      auto load_w = [&](u16x8 (&wq)[8], int c) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wq[ks] = wsrc[(c * 8 + ks) * 64];
      };
      auto dma_x = [&](int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int piece = j * 8 + wid;                                  // 1 KiB = 64 lanes x 16 B
          const int ct = piece >> 3, r = ((piece & 7) << 6) + lane;       // 8 pieces per (column tile, chunk)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)(xsrc + ((int64_t)(ct < nct ? ct : 0) * ksteps + c * 8) * 64 + r),
              // (tiles past nct re-write tile 0's piece with tile 0's data: no LDS of their own)
              (__attribute__((address_space(3))) void*)(xbuf[c & 1] + (ct < nct ? piece : (piece & 7)) * 1024), 16, 0, 0);
        }
      };
      auto mfmas = [&](const u16x8 (&wq)[8], int c) {
        const unsigned char* xb = xbuf[c & 1];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          if (ct < nct) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
              const u16x8 xf = *reinterpret_cast<const u16x8*>(xb + ((ct * 8 + ks) * 64 + lane) * 16);
              ga[ct] = mfma16<T16>(wq[ks], xf, ga[ct]);
            }
          }
      };
      u16x8 w0[8], w1[8];
      dma_x(0); load_w(w0, 0);
      if (nchunk > 1) { dma_x(1); load_w(w1, 1); }
      for (int c = 0; c < nchunk; c += 2) {
        // chunk c (even): everything issued before the 12 most recent VMEM ops has landed
        if (c + 1 < nchunk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mfmas(w0, c);
        __syncthreads();                                                  // xbuf[0] free again
        if (c + 2 < nchunk) { dma_x(c + 2); load_w(w0, c + 2); }
        if (c + 1 < nchunk) {
          if (c + 2 < nchunk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          mfmas(w1, c + 1);
          __syncthreads();
          if (c + 3 < nchunk) { dma_x(c + 3); load_w(w1, c + 3); }
        }
      }
      // D[row = 4*ul + gate][col = lane & 15]: lane (col, ul) holds the 4 gates of unit ul: + bias, park
      float bs[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[r] = p.bias[(int64_t)dir * 4 * R + (int64_t)r * R + tile0 * 4 + ul];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        if (ct < nct)
          *reinterpret_cast<float4*>(gxl + (ct * 16 + b) * GX_PITCH + ul * 4) =
              make_float4(ga[ct][0] + bs[0], ga[ct][1] + bs[1], ga[ct][2] + bs[2], ga[ct][3] + bs[3]);
      // (gxl is wave-private: no barrier needed before this wave reads it back below)
    }
    VOG_TSL(p.T, 1);
    // this wave's 16 rows of W_hh: registers for the whole sequence
    u16x8 wf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      wf[ks] = *reinterpret_cast<const u16x8*>(
          p.whh + ((((int64_t)dir * (R / 4) + tile0) * KSTEPS + ks) * 64 + lane) * 8);
    // hand-off buffer, u64 words: [parity][dir][16 sentences][R/2]; a word = two 16-bit h values +
    // the 32-bit number of the step that produced them. The tag makes every word self-validating:
    // a consumer needs no arrival flag and no acknowledgement wait, just one (re-tried) load.
    const int64_t hx_dir = (int64_t)dir * 16 * (R / 2);
    const int64_t hx_par = (int64_t)2 * 16 * (R / 2);
    bool dead = false;

    for (int s = 0; s < p.T; ++s) {
      VOG_TSL(s, 0);
      // input projections of this step (address-independent of everything else)
      float gin[4] = {0.f, 0.f, 0.f, 0.f};
      if (fused) {
        if (valid_b && s < len) {
          const int pos_in = dir == 0 ? s : len - 1 - s;
          const float4 g4 = *reinterpret_cast<const float4*>(gxl + (b * p.T + pos_in) * GX_PITCH + ul * 4);
          gin[0] = g4.x; gin[1] = g4.y; gin[2] = g4.z; gin[3] = g4.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          gin[r] = valid_b ? p.gxs[(((int64_t)dir * p.T + s) * p.Bn + b) * 4 * R + (int64_t)r * R + tile0 * 4 + ul] : 0.f;
      }
      // h_{s-1} of ALL units: written by the other workgroups with write-through atomics, read
      // with L1-bypassing atomics (agent scope on both sides: no fences needed). The eight waves need
      // the same Bn x R vector: the workgroup fetches it ONCE, 8 bytes per thread per round, into
      // LDS and every wave reads its MFMA B fragments from there.
      {
        const unsigned long long* hsrc = p.hx + (s & 1) * hx_par + hx_dir;
        const int items = p.Bn * (RW / 2);                   // words to fetch: sentence-major
        for (int base = tid; base < items; base += THREADS * 4) {
          unsigned long long v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int it = base + j * THREADS;
            v[j] = it < items ? __hip_atomic_load(hsrc + (int64_t)(it / (RW / 2)) * (R / 2) + it % (RW / 2), VOG_RLX_AGENT)
                              : ((unsigned long long)(unsigned)s << 32);
          }
          // re-fetch, as ONE batch per round, the words whose producer had not stored yet
          unsigned int spins = 0;
          for (;;) {
            bool stale = false;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              stale |= (base + j * THREADS < items) && (unsigned int)(v[j] >> 32) != (unsigned int)s;
            if (!stale || dead) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int it = base + j * THREADS;
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
          for (int j = 0; j < 4; ++j) {
            const int it = base + j * THREADS;
            if (it < items)
              *reinterpret_cast<unsigned int*>(&hs[(it / (RW / 2)) * HS_LD + (it % (RW / 2)) * 2]) = (unsigned int)v[j];
          }
        }
      }
      VOG_TSL(s, 1);
      dead = __syncthreads_or(dead ? 1 : 0) != 0;
      VOG_TSL(s, 2);
      // two accumulation chains (even / odd k-steps): a dependent MFMA issues every ~2x its issue slot
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks += 2) {
        u16x8 fh0 = {0, 0, 0, 0, 0, 0, 0, 0}, fh1 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (valid_b) {
          fh0 = *reinterpret_cast<const u16x8*>(&hs[b * HS_LD + ks * 32 + kg]);
          if (ks + 1 < KSTEPS) fh1 = *reinterpret_cast<const u16x8*>(&hs[b * HS_LD + (ks + 1) * 32 + kg]);
        }
        acc0 = mfma16<T16>(wf[ks], fh0, acc0);
        if (ks + 1 < KSTEPS) acc1 = mfma16<T16>(wf[ks + 1], fh1, acc1);
      }
#ifdef VOG_TS_DEBUG
      asm volatile("" :: "v"(acc0), "v"(acc1));
#endif
      VOG_TSL(s, 3);
      const bool active = s < len;
      const int pos = dir == 0 ? s : len - 1 - s;
      const int unit = tile0 * 4 + ul;
      if (active) {
        const float gi = acc0[0] + acc1[0] + gin[0], gf = acc0[1] + acc1[1] + gin[1];
        const float gg = acc0[2] + acc1[2] + gin[2], go = acc0[3] + acc1[3] + gin[3];
        c = sigm(gf) * c + sigm(gi) * tanh_(gg);
        const float hn = sigm(go) * tanh_(c);
        const unsigned short h16 = to16<T16>(hn);
        h_own = from16<T16>(h16);
        if (p.out_frag) p.out16[frag_a(b * p.T + pos, dir * R + unit, 2 * R)] = h16;
        else p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
      }
      // publish h_s of this tile: 4 units of one sentence = two 8-byte write-through stores
      {
        const unsigned int x0 = to16<T16>(h_own);
        const unsigned int x1 = __shfl(x0, b + 16), x2 = __shfl(x0, b + 32), x3 = __shfl(x0, b + 48);
        if (lane < 16 && valid_b) {
          const unsigned long long tag = (unsigned long long)(unsigned int)(s + 1) << 32;
          unsigned long long* dst = p.hx + ((s + 1) & 1) * hx_par + hx_dir + (int64_t)b * (R / 2) + tile0 * 2;
          __hip_atomic_store(dst, tag | x0 | ((unsigned long long)x1 << 16), VOG_RLX_AGENT);
          __hip_atomic_store(dst + 1, tag | x2 | ((unsigned long long)x3 << 16), VOG_RLX_AGENT);
        }
      }
      VOG_TSL(s, 4);
      if (s + 1 == p.T) break;                             // nothing reads h_T through hx
      __syncthreads();                                     // hs is rewritten at the top of the next step
    }
    // final hidden state rows (h of the last ACTIVE step of every sentence). A stalled hand-off
    // (a producer workgroup never became resident: more of these kernels in flight than the chip
    // holds, see vog_hip.h) must not pass for a result: the WHOLE output of the layer is poisoned
    // with NaN (every consumer - next layer, projection, argument vectors, both heads - propagates it
    // to mdl_outs) and sync[2] stays set for the host (vog_lstm_status).
    if (valid_b) {
      const int unit = tile0 * 4 + ul;
      const unsigned short v = dead ? (unsigned short)0x7fff : to16<T16>(h_own);
      if (p.out_frag) p.out16[frag_a(p.Bn * p.T + b, dir * R + unit, 2 * R)] = v;
      else p.out16[((int64_t)p.Bn * p.T + b) * 2 * R + (int64_t)dir * R + unit] = v;
      if (dead)
        for (int t = 0; t < p.T; ++t) {
          if (p.out_frag) p.out16[frag_a(b * p.T + t, dir * R + unit, 2 * R)] = (unsigned short)0x7fff;
          else p.out16[((int64_t)b * p.T + t) * 2 * R + (int64_t)dir * R + unit] = (unsigned short)0x7fff;
        }
    }
  }
};

template <typename T16, int KSTEPS>
__global__ __launch_bounds__(512) void lstm_layer_kernel(LstmLayerParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lstm_smem[];
  LstmLayerBody<T16, KSTEPS>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, lstm_smem);
}

}  // namespace vog
