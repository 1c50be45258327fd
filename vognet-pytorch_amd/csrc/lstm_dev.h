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
// the persistent layer kernel's forms: v_rcp_f32 (1 ulp) instead of the IEEE division sequence (12
// instructions per quotient, 5 quotients on the dependent chain of every step)
__device__ __forceinline__ float sigm_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
  const float a = fabsf(x);
  const float e = __expf(-2.0f * a);
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return x < 0.f ? -t : t;
}
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
  const float* gxs; const unsigned short* whh; unsigned short* hx; unsigned int* sync;
  unsigned short* out16; const int64_t* lens; int Bn, T, R; int out_frag;
  // fused input projection (wih != nullptr; gxs unused): x W_ih^T + bias for THIS workgroup's gate
  // rows is computed in the prologue. wih: [dir][unit/4][K/32][lane][8] (vog_lstm_pack_w), xa: the
  // layer input [Bn*T rows (b*T + t), K] in A-fragment order, bias: [2][4R] (b_ih + b_hh).
  const unsigned short* wih; const unsigned short* xa; const float* bias; int K;
  // round 5: `fault` (optional; device or device-visible pinned host memory) is incremented once per launch whose hand-off
  // timed out - unlike sync[2], which the next forward's prologue re-zeroes, it is never cleared by the library: the HOST
  // owns it (engine.Slot reads it without a device synchronisation). inject_stall: test hook, every workgroup behaves as
  // if its first wait had timed out.
  unsigned int* fault; int inject_stall;
  int half_proj;   // perf experiments only (WRONG results): layers with K >= 2048 run half of their input projection - bounds what
                   // moving half of it to helper workgroups could give (profiles/round5_lstm_proj_bound.md)
  // round 6, gate table (gxtab != nullptr; gxs / wih unused): the layer input is an embedding row, so x W_ih^T + bias is a
  // function of the TOKEN: gxtab = [vocab + 1][2][R][4 gates] fp32 holds it for every token (built once per checkpoint), tok = [Bn * T]
  // token ids. The gate inputs of step s + 1 are requested right behind step s's hand-off barrier - a whole step before they are
  // used - so the layer has no input projection and no prologue.
  const float* gxtab; const int32_t* tok;
};

#define VOG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define VOG_RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP


// 16 bytes straight from the L2 (sc1: the CU's own L1 is bypassed - it is never refreshed by another
// CU's stores). A compiler-visible buffer load (not inline asm): hipcc keeps its own vmcnt books.
__device__ __forceinline__ u32x4 load16_l2(const void* base, unsigned byte_off, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16 /* sc1 */);
}
// any of the 8 halfwords == 0xffff (the "not written yet" pattern; as f16 / bf16 it is a NaN no h can be)
__device__ __forceinline__ bool has_unwritten(u32x4 v) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const unsigned y = ~v[i]; m |= (y - 0x00010001u) & v[i] & 0x80008000u; }
  return m != 0;
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}

// Body form (common.h): 512 threads = 8 waves, ONE 16-row tile (4 units x 4 gates) of W_hh per wave,
// i.e. 128 registers of weights per lane - the kernel fits the 256-register budget of a 512-thread
// workgroup and can therefore share a launch with the other 512-thread bodies (pair.hip).
//
// Hand-off (round 3). h_s of ALL units travels through `hx` = [step][dir][sentence][R] 16-bit values,
// one slot per step (nothing is ever rewritten inside a forward), armed with 0xffff by the forward's
// prologue (vog_lang_prep `ones_bytes`). Every VALUE validates itself: a consumer needs no tag, no flag
// and no ordering between words - a 16-byte load whose 8 halfwords hold no 0xffff is complete, whatever
// the granularity at which the stores became visible. Per step a workgroup fetches the Bn x R vector
// ONCE (one 16-byte L1-bypassing load per thread at Bn = 4), re-fetches chunks that still hold the
// pattern, stages it in LDS (two buffers: ONE barrier per step) and every wave reads its MFMA B
// fragments from there.
// Where the workgroups of a direction sit decides the store flavour: every workgroup reports its XCC id
// at kernel start, and only if all of a direction report the same one (small layers: one or two
// workgroups per direction) the direction uses stores that stay in that XCD's L2 - correctness never
// depends on the placement. (A forced placement - the 32 workgroups of a direction on ONE XCD, the other
// blocks of the launch on the remaining six - was built and measured in round 3: 1.9 instead of 2.5 us per
// step, but W_ih / W_hh then enter through one XCD's fabric port (+7 us per layer) and every other kernel's
// blocks dealt to the two occupied XCDs wait for the layer: 31 k instead of 56 k queries/s with 4 forwards
// in flight. Removed; scratch/ubench/{xcd_allgather,handoff_v3}.hip keep the exchange measurements.)
template <typename T16, int KSTEPS>
struct LstmLayerBody {
  using Params = LstmLayerParams;
  static constexpr int THREADS = 512;
  static constexpr int RW = KSTEPS * 32, HS_LD = RW + 32;  // +32 halfwords: sentence rows 16 banks apart
  // LDS: [gates of the fused input projection, kept for the whole recurrence]
  //      [union: layer-input chunks of the projection (2 x nct x 8 KiB) | h staging (2 x Bn x HS_LD x 2 B)] [16 B flags] [publish buffer]
  static constexpr int GX_PITCH = 20;                      // floats per column (16 + 4: spreads the banks)
  // gates of the fused projection: [8 waves][column tiles x 16 columns][GX_PITCH] fp32, sized by the columns the layer
  // really has (Bn*T <= 80, round 6: 64 before - a bs = 4 batch with a sentence of 17-20 words used to fall back to the GEMM
  // launches and the gx round trip): 30 KB instead of 40 at cfg 2, which puts the whole workgroup under half a CU's LDS
  static constexpr int NCT_MAX = 5;                        // 16-column tiles of the fused projection
  static constexpr size_t gx_bytes(int ncols) { return (size_t)8 * ((ncols + 15) / 16 * 16) * GX_PITCH * 4; }
  static constexpr size_t GX_BYTES = gx_bytes(16 * NCT_MAX);
  static constexpr size_t hs_bytes(int Bn) { return ((size_t)2 * Bn * HS_LD * 2 + 15) / 16 * 16; }
  static constexpr size_t xbuf_bytes(int ncols) { return (size_t)2 * ((ncols + 15) / 16) * 8 * 1024; }
  static constexpr int TOK_MAX = 16 * 24;                   // token ids staged for the gate table (Bn * T)
  static constexpr size_t TAIL = 16 + 16 * 32 * 2 + TOK_MAX * 4;   // flags + the publish buffer [<= 16 sentences][32 units] + token ids
  static constexpr size_t lds_plain(int Bn) { return hs_bytes(Bn) + TAIL; }
  static constexpr size_t lds_fused(int Bn, int ncols) {
    return gx_bytes(ncols) + (hs_bytes(Bn) > xbuf_bytes(ncols) ? hs_bytes(Bn) : xbuf_bytes(ncols)) + TAIL;
  }
  static constexpr size_t LDS_MAX = GX_BYTES + ((size_t)2 * 16 * HS_LD * 2 > (size_t)2 * NCT_MAX * 8 * 1024 ? (size_t)2 * 16 * HS_LD * 2 : (size_t)2 * NCT_MAX * 8 * 1024) + TAIL;

  static __device__ __forceinline__ void run(const LstmLayerParams& p, const BlockCtx& cx, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = cx.by;
    const int R = p.R;
    const int tile0 = cx.bx * 8 + wid;                     // one 16-row tile (4 units) per wave
    const int b = lane & 15, ul = lane >> 4, kg = (lane >> 4) * 8;
    const bool valid_b = b < p.Bn;
    const int len = valid_b ? (int)p.lens[b] : 0;
    const bool fused = p.wih != nullptr;
    const int ncols = p.Bn * p.T;
    const int gx_cols = (ncols + 15) / 16 * 16;
    unsigned char* uni = smem + (fused ? gx_bytes(ncols) : 0);
    unsigned short* hs = reinterpret_cast<unsigned short*>(uni);
    const size_t uni_bytes = fused ? (hs_bytes(p.Bn) > xbuf_bytes(ncols) ? hs_bytes(p.Bn) : xbuf_bytes(ncols)) : hs_bytes(p.Bn);
    unsigned int* flags = reinterpret_cast<unsigned int*>(uni + uni_bytes);   // [0] timeout seen, [1] direction shares an XCD
    unsigned short* pub = reinterpret_cast<unsigned short*>(uni + uni_bytes + 16);
    int* tokl = reinterpret_cast<int*>(uni + uni_bytes + 16 + 16 * 32 * 2);
    const bool tab = p.gxtab != nullptr;
    const int unit0 = tile0 * 4 + ul;
    // gate inputs of step s2 from the table row of that step's token (zeros past the sentence's end: never used)
    auto load_tab = [&](int s2, float (&g)[4], const int* tk) __attribute__((always_inline)) {
      int row = 0;
      const bool on = valid_b && s2 < len && s2 < p.T;
      if (on) row = tk[b * p.T + (dir == 0 ? s2 : len - 1 - s2)];
      // (the four gates of a unit are adjacent in the table: one 16-byte load per lane, 64 contiguous bytes per sentence and
      // wave; unconditional: lanes that are off read row 0 and never use it)
      const float4 v = *reinterpret_cast<const float4*>(p.gxtab + (((int64_t)row * 2 + dir) * R + unit0) * 4);
      g[0] = v.x; g[1] = v.y; g[2] = v.z; g[3] = v.w;
    };
    float gtA[4] = {0.f, 0.f, 0.f, 0.f}, gtB[4] = {0.f, 0.f, 0.f, 0.f}, gtC[4] = {0.f, 0.f, 0.f, 0.f};
    if (tab) {
      // steps 0 - 2 straight from the global token ids (two dependent loads from kernel start, beside the W_hh loads); the
      // ids of the later steps come from LDS
      load_tab(0, gtA, p.tok);
      load_tab(1, gtB, p.tok);
      load_tab(2, gtC, p.tok);
      for (int i = tid; i < p.Bn * p.T; i += THREADS) tokl[i] = p.tok[i];
    }

    // where am I: one report per workgroup, read back before the first publish (after the prologue)
    if (tid == 0) {
      flags[0] = 0;
      // ONE atomic per workgroup: a counter per (direction, XCC id). (A mask word + an arrival counter
      // would be two relaxed atomics whose order another workgroup need not observe.)
      __hip_atomic_fetch_add(p.sync + 16 + dir * 16 + xcc_id(), 1u, VOG_RLX_AGENT);
    }

    float c = 0.f, h_own = 0.f;
    VOG_TSL(p.T, 0);
    // ---- fused input projection (replaces the separate x W_ih^T GEMM launch and its [2][T][Bn][4R]
    // fp32 round trip): every wave computes the gates of ITS 16 rows for all Bn*T (sentence, position)
    // columns: W_ih rows stream once through registers (K in chunks of 256), the layer input is
    // staged per chunk in LDS and shared by the 8 waves; the result stays in LDS for the recurrence.
    float* gxl = reinterpret_cast<float*>(smem) + (size_t)wid * gx_cols * GX_PITCH;
    if (fused) {
      unsigned char* xch = uni;
      const int nct = (ncols + 15) >> 4;                                // <= NCT_MAX column tiles
      const int ksteps = p.K >> 5;                                     // K % 256 == 0
      const int nchunk = (p.half_proj && p.K >= 2048) ? (ksteps >> 4) : (ksteps >> 3);
      f32x4 ga[NCT_MAX];
#pragma unroll
      for (int ct = 0; ct < NCT_MAX; ++ct) ga[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      const u16x8* wsrc = reinterpret_cast<const u16x8*>(p.wih) + (((int64_t)dir * (R / 4) + tile0) * ksteps) * 64 + lane;
      const uint4* xsrc = reinterpret_cast<const uint4*>(p.xa);
      // chunk c of column tile ct: k-steps [8c, 8c+8) = 8 KiB contiguous at ((ct*ksteps + 8c)*64) uint4.
      // Pipeline: the layer input goes global -> LDS by LDS-DMA (global_load_lds: no VGPR round trip, so
      // the compiler cannot serialise it behind the weight loads; first form, through registers: every
      // chunk waited vmcnt(0), 4 us per chunk), double buffered; the weight fragments of the NEXT
      // chunk are in flight in a second register set. Two chunks = 80 KB per workgroup are always
      // outstanding; per chunk: one counted wait, barrier, 8 x nct MFMAs, barrier.
      unsigned char* xbuf[2] = {xch, xch + nct * 8 * 1024};
      auto load_w = [&](u16x8 (&wq)[8], int c) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) wq[ks] = wsrc[(c * 8 + ks) * 64];
      };
      auto dma_x = [&](int c) {
#pragma unroll
        for (int j = 0; j < NCT_MAX; ++j) {
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
        for (int ct = 0; ct < NCT_MAX; ++ct)
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
        // chunk c (even): everything issued before the NCT_MAX + 8 most recent VMEM ops (the next chunk's) has landed
        if (c + 1 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCT_MAX + 8) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mfmas(w0, c);
        __syncthreads();                                                  // xbuf[0] free again
        if (c + 2 < nchunk) { dma_x(c + 2); load_w(w0, c + 2); }
        if (c + 1 < nchunk) {
          if (c + 2 < nchunk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCT_MAX + 8) : "memory");
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
      for (int ct = 0; ct < NCT_MAX; ++ct)
        if (ct < nct)
          *reinterpret_cast<float4*>(gxl + (ct * 16 + b) * GX_PITCH + ul * 4) =
              make_float4(ga[ct][0] + bs[0], ga[ct][1] + bs[1], ga[ct][2] + bs[2], ga[ct][3] + bs[3]);
      // (gxl is wave-private: no barrier needed before this wave reads it back below; the layer-input
      // chunks share their LDS with the h staging buffers, which are first written in step 1 - behind
      // the barrier below)
    }
    VOG_TSL(p.T, 1);
    // this wave's 16 rows of W_hh: registers for the whole sequence
    u16x8 wf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      wf[ks] = *reinterpret_cast<const u16x8*>(
          p.whh + ((((int64_t)dir * (R / 4) + tile0) * KSTEPS + ks) * 64 + lane) * 8);
    bool dead = false;
    // do all workgroups of this direction share one XCD? (every one has reported by now, or will)
    if (tid == 0) {
      unsigned spins = 0, nz = 0;
      for (;;) {
        unsigned sum = 0;
        nz = 0;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          const unsigned v = __hip_atomic_load(p.sync + 16 + dir * 16 + x, VOG_RLX_AGENT);
          sum += v;
          nz += v != 0 ? 1u : 0u;
        }
        if (sum >= cx.gx) break;                           // counters only grow: every report is in
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 20) || __hip_atomic_load(p.sync + 2, VOG_RLX_AGENT) != 0) {   // ~1 s: give up
          __hip_atomic_store(p.sync + 2, 1u, VOG_RLX_AGENT);
          flags[0] = 1;
          nz = 2;
          break;
        }
      }
      flags[1] = nz == 1 ? 1u : 0u;
    }
    __syncthreads();
    const bool same_xcd = flags[1] != 0;
    // inject_stall (test hook): 1 = every workgroup behaves as if its first wait had timed out; 2 = only the workgroups of
    // direction 1 do (the case a late / non-resident workgroup of ONE direction produces: direction 0 ends clean)
    dead = flags[0] != 0 || p.inject_stall == 1 || (p.inject_stall == 2 && dir == 1);

    const unsigned hx_bytes = (unsigned)((size_t)p.T * 2 * p.Bn * R * 2);
    const int cps = RW / 8;                                // 16-byte chunks per sentence
    const int nchunks = p.Bn * cps;
    const int hs_buf = p.Bn * HS_LD;                       // halfwords per staging buffer

    for (int s = 0; s < p.T; ++s) {
      VOG_TSL(s, 0);
      // input projections of this step (address-independent of everything else)
      float gin[4] = {0.f, 0.f, 0.f, 0.f};
      if (tab) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { gin[r] = gtA[r]; gtA[r] = gtB[r]; gtB[r] = gtC[r]; }
        if (s == 0) load_tab(3, gtC, tokl);                  // (no hand-off in step 0: requested here)
      } else
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
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (s > 0) {                                         // h_{-1} = 0: step 0 has no matrix part
        // h_{s-1} of ALL units: slot s-1. One 16-byte chunk (8 units of one sentence) per thread and round.
        unsigned short* hsb = hs + (s & 1) * hs_buf;
        const unsigned slot_off = (unsigned)(((size_t)(s - 1) * 2 + dir) * p.Bn * R * 2);
        // Round 6: ALL of a thread's chunks are requested before the first one is looked at (Bn = 16: four per thread; the loop
        // used to wait out one L2 round trip per chunk - a step took 4.4 us at Bn = 16 against 2.5 at Bn = 4, with the same
        // MFMA and gate work); chunks that still hold the pattern are re-fetched one by one as before.
        constexpr int MAXR = 4;                            // Bn <= 16: <= 4 chunks per thread
#ifndef VOG_LSTM_INFL
#define VOG_LSTM_INFL 4      // measured on the GPU (scratch/r6_c.sh): see DESIGN.md
#endif
        constexpr int INFL = VOG_LSTM_INFL;                // chunks in flight per thread (registers: 4 each)
        u32x4 vq[INFL];
        // (cps = RW / 8 is a power of two at every supported width: chunk r of a thread is sentence sb0 + r * (THREADS / cps))
        const int sb0 = tid / cps, j0 = tid - sb0 * cps;
        const unsigned off0 = slot_off + (unsigned)(sb0 * R + j0 * 8) * 2;
        const unsigned off_step = (unsigned)((THREADS / cps) * R) * 2;
        const int lds0 = sb0 * HS_LD + j0 * 8, lds_step = (THREADS / cps) * HS_LD;
        static_assert(THREADS % (RW / 8) == 0 || (RW / 8) % THREADS == 0, "hand-off fetch: chunk stride");
#pragma unroll
        for (int r0 = 0; r0 < MAXR; r0 += INFL) {
#pragma unroll
        for (int q = 0; q < INFL; ++q)
          if (tid + (r0 + q) * THREADS < nchunks) vq[q] = load16_l2(p.hx, off0 + (r0 + q) * off_step, hx_bytes);
#pragma unroll
        for (int q = 0; q < INFL; ++q) {
          const int r = r0 + q;
          if (tid + r * THREADS < nchunks) {
            u32x4 v = vq[q];
            unsigned int spins = 0;
            const unsigned offr = off0 + r * off_step;
            while (has_unwritten(v) && !dead) {
              asm volatile("" ::: "memory");
              v = load16_l2(p.hx, offr, hx_bytes);
              if ((++spins & 255u) == 0 &&
                  (spins > (1u << 20) || __hip_atomic_load(p.sync + 2, VOG_RLX_AGENT) != 0)) {   // ~1 s: give up
                __hip_atomic_store(p.sync + 2, 1u, VOG_RLX_AGENT);
                dead = true;
              }
            }
            *reinterpret_cast<u32x4*>(&hsb[lds0 + r * lds_step]) = v;
          }
        }
        }
        static_assert(16 * (RW / 8) <= MAXR * THREADS, "hand-off fetch: more chunks per thread than MAXR");
        if (dead) flags[0] = 1;
        VOG_TSL(s, 1);
        __syncthreads();
        VOG_TSL(s, 2);
        // the table rows of step s + 3 (gtA / gtB hold steps s + 1 / s + 2 by now), two steps ahead of their use - one step
        // was measured short: +0.5 us per step at Bn = 4, +0.7 at Bn = 16 (random 64-byte reads of a 164 MB table). Issued
        // behind this step's hand-off, so that the wait for the NEXT hand-off - loads return in order - is the first one that
        // can see them.
        if (tab) load_tab(s + 3, gtC, tokl);
        const unsigned int dead_wg = flags[0];
        // lanes of the unused MFMA columns read sentence 0 (their results are never looked at)
        const unsigned short* hrow = hsb + (valid_b ? b : 0) * HS_LD + kg;
        // two accumulation chains (even / odd k-steps). The B fragments come from LDS 8 at a time, one
        // group AHEAD of the MFMAs that use them (two register sets; the scheduling fences keep hipcc from
        // folding the groups back into read-2 / wait / MFMA-2, which exposes the LDS latency 16 times per
        // step: 0.95 us of a 3 us step in round 2)
        // Group size 2 (round 4; 8 before): the layer kernel allocates 190 registers instead of 238, i.e. 2 x 192 of a SIMD lane's
        // 512 - what is left holds one wave of a <= 128-register workgroup, so the lean kernels of OTHER forwards share the 64 CUs a
        // layer holds for 35-45 us (+8 % at 128 CUs, +2 % single stream, neutral at 256 CUs / 4 streams; same accumulation order:
        // even k-steps -> acc0, odd -> acc1, bit-identical). profiles/round4_cu_bound_experiments.md, section 7
#ifndef VOG_LSTM_G
#define VOG_LSTM_G 2
#endif
        constexpr int G = KSTEPS < VOG_LSTM_G ? KSTEPS : VOG_LSTM_G;
        u16x8 fa[G], fb[G];
#pragma unroll
        for (int j = 0; j < G; ++j) fa[j] = *reinterpret_cast<const u16x8*>(hrow + j * 32);
#pragma unroll
        for (int k0 = 0; k0 < KSTEPS; k0 += 2 * G) {
          __builtin_amdgcn_sched_barrier(0);
          if (k0 + G < KSTEPS) {
#pragma unroll
            for (int j = 0; j < G; ++j) fb[j] = *reinterpret_cast<const u16x8*>(hrow + (k0 + G + j) * 32);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < G; ++j) {
            if (j & 1) acc1 = mfma16<T16>(wf[k0 + j], fa[j], acc1);
            else acc0 = mfma16<T16>(wf[k0 + j], fa[j], acc0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (k0 + 2 * G < KSTEPS) {
#pragma unroll
            for (int j = 0; j < G; ++j) fa[j] = *reinterpret_cast<const u16x8*>(hrow + (k0 + 2 * G + j) * 32);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (k0 + G < KSTEPS) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
              if (j & 1) acc1 = mfma16<T16>(wf[k0 + G + j], fb[j], acc1);
              else acc0 = mfma16<T16>(wf[k0 + G + j], fb[j], acc0);
            }
          }
        }
        dead = dead || dead_wg != 0;
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
        c = sigm_fast(gf) * c + sigm_fast(gi) * tanh_fast(gg);
        const float hn = sigm_fast(go) * tanh_fast(c);
        unsigned short h16 = to16<T16>(hn);
        if (h16 == 0xffffu) h16 = 0xfe00u;                 // a NaN, but never the "not written yet" pattern
        h_own = from16<T16>(h16);
        if (p.out_frag) p.out16[frag_a(b * p.T + pos, dir * R + unit, 2 * R)] = h16;
        else p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
      }
      // publish h_s of this tile into slot s (nothing reads h_{T-1} through hx).
      // One XCD: every lane stores its own 16-bit value with a store that stays in the shared L2, where
      // the readers' sc1 loads are served (0.75 us per exchange in scratch/ubench/handoff_v3.hip).
      // Several XCDs: the stores are written through to memory, and there the NUMBER of write
      // transactions decides (1.75-2.0 us with a store per wave and sentence, 1.33 us with 16-byte
      // stores of one wave): the 8 waves park their values in LDS and 4 lanes per sentence store the
      // workgroup's 64 bytes - one more barrier per step, half a microsecond less on the fabric.
      // (No cross-lane packing with ds_bpermute: hipcc puts s_waitcnt vmcnt(0) in front of it here, i.e.
      // the publish would wait for the write acknowledge of the previous step's stores.)
      if (s + 1 < p.T) {
        unsigned short x0 = to16<T16>(h_own);
        if (x0 == 0xffffu) x0 = 0xfe00u;
        unsigned short* slot = p.hx + ((size_t)s * 2 + dir) * p.Bn * R;
        if (same_xcd) {
          if (valid_b) __hip_atomic_store(slot + (size_t)b * R + tile0 * 4 + ul, x0, VOG_RLX_WG);
        } else {
          if (valid_b) pub[b * 32 + wid * 4 + ul] = x0;
          __syncthreads();
          if (tid < p.Bn * 4) {
            const int sb = tid >> 2, q = tid & 3;
            const u32x4 v = *reinterpret_cast<const u32x4*>(pub + sb * 32 + q * 8);
            u32x4* dst = reinterpret_cast<u32x4*>(slot + (size_t)sb * R + cx.bx * 32 + q * 8);
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
          }
        }
      }
      VOG_TSL(s, 4);
    }
    // final hidden state rows (h of the last ACTIVE step of every sentence). A stalled hand-off
    // (a producer workgroup never became resident: more of these kernels in flight than the chip
    // holds, see vog_hip.h) must not pass for a result: the WHOLE output of the layer is poisoned
    // with NaN (every consumer - next layer, projection, argument vectors, both heads - propagates it
    // to mdl_outs) and sync[2] stays set for the host (vog_lstm_status).
    // once per launch, reported by WHICHEVER workgroup ends dead first (the hand-off slots and the residency wait
    // are per direction and sync[2] is only polled while spinning, so a stall confined to one direction - or one whose only
    // late workgroup is (0, 0) itself - never reaches workgroup (0, 0)): sync[3] is zeroed by the forward's prologue with the
    // rest of the sync block, the first dead workgroup to swap it to 1 bumps the host's sticky counter.
    if (dead && p.fault && tid == 0 &&
        __hip_atomic_exchange(p.sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
      __hip_atomic_fetch_add(p.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

#ifndef VOG_LSTM_KATTR
#define VOG_LSTM_KATTR
#endif
template <typename T16, int KSTEPS>
__global__ __launch_bounds__(512) VOG_LSTM_KATTR void lstm_layer_kernel(LstmLayerParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lstm_smem[];
  LstmLayerBody<T16, KSTEPS>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, lstm_smem);
}

}  // namespace vog
