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
//   * the fp32 residual stream stays in registers (d = 512) or in a workgroup-private, thread-major
//     scratch slab (d = 768: 96 more registers would not fit beside the accumulators).
// Bound: the workgroup streams 2.75 MB (mul_tx incl. lin2) / 1.05 MB (obj_tx) of weights through
// one CU's L2 port (~64 B/clk) and issues 5376 / 1536 32x32x16 MFMAs: both ~18 us at cfg 2 -> the
// kernel sits at the CU's own MFMA/ingest balance point; what it removes is six dependent launches,
// their ramps, and every intermediate tensor.
#include "common.h"

namespace vog {

struct TailParams {
  const unsigned short* attn16; int KWO;
  const unsigned short *wo_p, *w1_p, *w2_p, *wl_p;
  const float* residual; int64_t ldr;
  const float *res_vis, *res_lang; int rv_nfrm, rv_nppf, rv_nsrl, rv_dv, rv_dl, rv_lpv, rv_ncv;
  const float *ln1g, *ln1b, *b1, *b2, *ln2g, *ln2b;
  float* y32; unsigned short* y16; int y16_bf16;
  const float *bl, *wl2, *bl2;
  vog_score_args sc;
  int M;
  int dbgf;      // perf experiments only, read by the DBG & 4 instantiation: 1 no residual, 2 no attention staging, 4 no LayerNorm, 8 no outputs
};

// One GEMM stage of the chain: acc[i][rb] (32 columns n x 32 rows m, swapped) += W_blk(i) . X^T over
// KS k-steps of 16. Weight fragments of n-block b start at wp + b*KS*512 halfwords.
//  * Straight-line software pipeline, PF k-steps of weight prefetch, NO branch around a load (a
//    conditional prefetch makes hipcc fall back to s_waitcnt vmcnt(0) in front of every k-step), and a
//    sched_barrier behind every refill (left alone, the scheduler sinks all PF refills to the end of
//    the unrolled body and the prefetch distance collapses to one k-step).
//  * The k-steps are visited in ROTATED order, starting at `rot` (a function of the workgroup's
//    position on its XCD): the ~8 workgroups that share an L2 then stream 8 different parts of the
//    weight matrix at any moment, so a line is fetched from the Infinity Cache by ONE of them and
//    found in L2 by the other seven (in lock-step every workgroup took the ~2 us fabric miss on
//    every line: kernel boundaries leave the XCD L2s cold). The fp32 summation order depends on the
//    row block only, so results stay bit-reproducible.
template <typename TT, int NBW, int PF, bool ZERO, int DBG = 0>
__device__ __forceinline__ void tail_gemm(f32x16 (&acc)[NBW][2], const unsigned short* __restrict__ wp,
                                          int blk0, int blk_step, int KS, int rot,
                                          const unsigned char* xl, int pitch, int lane) {
  constexpr int KST = (DBG & 1) ? 0 : 64;     // DBG 1 (perf experiments): every weight load hits the block's first KiB
  const int ml = lane & 31, hi = lane >> 5;
  const u16x8* wb[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    wb[i] = reinterpret_cast<const u16x8*>(wp + ((int64_t)(blk0 + i * blk_step) * KS) * 512) + lane;
    if constexpr (ZERO) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][rb][r] = 0.f;
    }
  }
  if constexpr (DBG & 4) return;              // DBG 4: no GEMM stage at all (skeleton: loads, LayerNorms, stores)
  auto kk = [&](int t) { const int k = t + rot; return k >= KS ? k - KS : k; };   // t < 2*KS - rot
  u16x8 wq[PF][NBW];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int k = kk(j);
#pragma unroll
    for (int i = 0; i < NBW; ++i) wq[j][i] = wb[i][k * KST];
  }
  const unsigned char* x0 = xl + ml * pitch + hi * 16;
  const unsigned char* x1 = x0 + 32 * pitch;
  u16x8 xf0 = *reinterpret_cast<const u16x8*>(x0 + kk(0) * 32), xf1 = *reinterpret_cast<const u16x8*>(x1 + kk(0) * 32);
  auto step = [&](int j, int t, bool refill) {
    const int kn = kk(t + 1);                  // next k-step's activations (t + 1 == KS wraps to `rot`: in bounds)
    const u16x8 n0 = *reinterpret_cast<const u16x8*>(x0 + kn * 32);
    const u16x8 n1 = *reinterpret_cast<const u16x8*>(x1 + kn * 32);
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      if constexpr (DBG & 2) {                 // DBG 2: no matrix work (keeps the operands live)
        acc[i][0][0] += __builtin_bit_cast(float, (unsigned)wq[j][i][0] | ((unsigned)xf0[0] << 16));
        acc[i][1][0] += __builtin_bit_cast(float, (unsigned)wq[j][i][1] | ((unsigned)xf1[0] << 16));
      } else {
        acc[i][0] = mfma32<TT>(wq[j][i], xf0, acc[i][0]);
        acc[i][1] = mfma32<TT>(wq[j][i], xf1, acc[i][1]);
      }
    }
    if (refill) {
      const int kl = kk(t + PF);
#pragma unroll
      for (int i = 0; i < NBW; ++i) wq[j][i] = wb[i][kl * KST];
    }
    xf0 = n0; xf1 = n1;
    __builtin_amdgcn_sched_barrier(0);
  };
  int t = 0;
#pragma unroll 1
  for (; t < KS - PF; t += PF) {               // KS % PF == 0, KS >= 2 * PF (not unrolled further: with a
                                               // compile-time KS hipcc unrolls all of K and spills the addresses)
#pragma unroll
    for (int j = 0; j < PF; ++j) step(j, t + j, true);
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) step(j, t + j, false);
}

// LayerNorm over n of the swapped accumulator tile of the whole workgroup (D columns spread over the
// 8 waves): two-pass statistics as layernorm_kernel (mean, then sum of squared deviations); gamma / beta
// come from LDS (prefetched at kernel start: no dependent global round trip in the epilogue).
template <int NB>
__device__ __forceinline__ void tail_ln(f32x16 (&acc)[NB][2], const float* gamma_l, const float* beta_l,
                                        float* red, int w, int lane, int nblk0) {
  constexpr int D = NB * 256;
  int ml = lane & 31, hi = lane >> 5;
  // opaque copies: keeps hipcc from sharing the 16 exchange addresses between the two LayerNorms of
  // the kernel (it kept them live - spilled - across two GEMM stages instead of re-deriving them)
  asm volatile("" : "+v"(ml), "+v"(hi));
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[rb] += acc[i][rb][r];
  s[0] += __shfl_xor(s[0], 32); s[1] += __shfl_xor(s[1], 32);
  if (hi == 0) { red[w * 64 + ml] = s[0]; red[w * 64 + 32 + ml] = s[1]; }
  __syncthreads();
  float mean[2], rstd[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) t += red[ww * 64 + rb * 32 + ml];
    mean[rb] = t / (float)D;
  }
  float q[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float dd = acc[i][rb][r] - mean[rb]; q[rb] += dd * dd; }
  q[0] += __shfl_xor(q[0], 32); q[1] += __shfl_xor(q[1], 32);
  __syncthreads();                              // every wave has read the sums
  if (hi == 0) { red[w * 64 + ml] = q[0]; red[w * 64 + 32 + ml] = q[1]; }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) t += red[ww * 64 + rb * 32 + ml];
    rstd[rb] = 1.0f / sqrtf(t / (float)D + 1e-5f);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = (nblk0 + i) * 32 + 8 * g + 4 * hi;
      const float4 gm = *reinterpret_cast<const float4*>(gamma_l + n);
      const float4 bt = *reinterpret_cast<const float4*>(beta_l + n);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        acc[i][rb][4 * g + 0] = (acc[i][rb][4 * g + 0] - mean[rb]) * rstd[rb] * gm.x + bt.x;
        acc[i][rb][4 * g + 1] = (acc[i][rb][4 * g + 1] - mean[rb]) * rstd[rb] * gm.y + bt.y;
        acc[i][rb][4 * g + 2] = (acc[i][rb][4 * g + 2] - mean[rb]) * rstd[rb] * gm.z + bt.z;
        acc[i][rb][4 * g + 3] = (acc[i][rb][4 * g + 3] - mean[rb]) * rstd[rb] * gm.w + bt.w;
      }
    }
}

// opaque copy of a lane index: address arithmetic derived from it cannot be shared (and kept live in
// registers, i.e. spilled) across the stages of the chain
__device__ __forceinline__ int fresh(int v) { asm volatile("" : "+v"(v)); return v; }

template <typename TT>
__device__ __forceinline__ u16x4 cvt4(float a, float b, float c, float d) {
  return u16x4{to16<TT>(a), to16<TT>(b), to16<TT>(c), to16<TT>(d)};
}

__device__ __forceinline__ float tail_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <typename T16, typename TH, int NB, bool SCORE, int DBG = 0>
__global__ __launch_bounds__(512) void tx_tail_kernel(TailParams p) {
  constexpr int D = NB * 256, DH = D / 2;
  constexpr int NB1 = NB == 3 ? 2 : 1;              // FFN1 n-blocks per wave (DH/32 = 8 or 12 over 8 waves)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, ml = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 64;
  const int xcols = p.KWO > D ? p.KWO : D;
  unsigned char* X = smem;
  unsigned char* Y = X + 64 * (xcols + 8) * 2;
  float* red = reinterpret_cast<float*>(Y + 64 * (DH + 8) * 2);      // [8 waves][64 rows]
  float* vec2 = red + 512;                          // ln2 gamma, ln2 beta, b2 [D each], b1 [DH]
  float* vec1 = reinterpret_cast<float*>(Y);        // ln1 gamma, ln1 beta [D each]: parked in Y until FFN1 writes it
  const int p1 = (p.KWO + 8) * 2, pD = (D + 8) * 2, pH = (DH + 8) * 2;
  // position of this workgroup among the ones that share its XCD's L2 (block b runs on XCD b % 8;
  // speed only): staggers the k order of the weight streams
  const int xpos = (blockIdx.x >> 3) & 7;

  // ---- stage 0: attention rows + the epilogue vectors -> LDS; residual rows -> accumulators.
  // Everything the chain will need from memory besides the weight streams is requested here, in one
  // round trip: no epilogue below waits for a global load.
  {
    const int cpr = ((DBG & 4) && (p.dbgf & 2)) ? 0 : (p.KWO >> 3);
    for (int idx = tid; idx < 64 * cpr; idx += 512) {
      const int r = idx / cpr, c = idx - r * cpr;
      int m = m0 + r;
      m = m < p.M ? m : p.M - 1;
      const uint4 v = *reinterpret_cast<const uint4*>(p.attn16 + (int64_t)m * p.KWO + c * 8);
      *reinterpret_cast<uint4*>(X + r * p1 + c * 16) = v;
    }
    for (int i = tid; i < D / 4; i += 512) {
      reinterpret_cast<float4*>(vec1)[i] = reinterpret_cast<const float4*>(p.ln1g)[i];
      reinterpret_cast<float4*>(vec1 + D)[i] = reinterpret_cast<const float4*>(p.ln1b)[i];
      reinterpret_cast<float4*>(vec2)[i] = reinterpret_cast<const float4*>(p.ln2g)[i];
      reinterpret_cast<float4*>(vec2 + D)[i] = reinterpret_cast<const float4*>(p.ln2b)[i];
      reinterpret_cast<float4*>(vec2 + 2 * D)[i] = reinterpret_cast<const float4*>(p.b2)[i];
    }
    for (int i = tid; i < DH / 4; i += 512)
      reinterpret_cast<float4*>(vec2 + 3 * D)[i] = reinterpret_cast<const float4*>(p.b1)[i];
  }
  int mrow[2];
  f32x16 acc[NB][2];
  {
    const float* rp[2]; const float* lp[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      mrow[rb] = m0 + rb * 32 + ml;
      const int mc = mrow[rb] < p.M ? mrow[rb] : p.M - 1;
      if (p.res_vis) {
        const int N = p.rv_nsrl * p.rv_nppf;
        const int s = mc / N, j = mc - s * N;
        const int a = j / p.rv_nppf, pp = j - a * p.rv_nppf;
        const int v = s / p.rv_nfrm;
        const int lv = p.rv_lpv ? v : v / p.rv_ncv;
        rp[rb] = p.res_vis + ((int64_t)s * p.rv_nppf + pp) * p.rv_dv;
        lp[rb] = p.res_lang + ((int64_t)lv * p.rv_nsrl + a) * p.rv_dl;
      } else {
        rp[rb] = p.residual + (int64_t)mc * p.ldr;
        lp[rb] = rp[rb];
      }
    }
    // the residual IS the initial accumulator: x + attn Wo^T is one MFMA chain, no epilogue add
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (w * NB + i) * 32 + 8 * g + 4 * hi;
        const bool in_lang = p.res_vis && n >= p.rv_dv;       // wave-uniform per (i): dv % 32 == 0
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
          float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
          if (!((DBG & 4) && (p.dbgf & 1))) r = *reinterpret_cast<const float4*>(in_lang ? lp[rb] + (n - p.rv_dv) : rp[rb] + n);
          acc[i][rb][4 * g + 0] = r.x; acc[i][rb][4 * g + 1] = r.y;
          acc[i][rb][4 * g + 2] = r.z; acc[i][rb][4 * g + 3] = r.w;
        }
      }
  }
  __syncthreads();

  // ---- stage 1: x + attn Wo^T, LayerNorm
  tail_gemm<T16, NB, NB == 3 ? 4 : 6, false, DBG>(acc, p.wo_p, w * NB, 1, p.KWO >> 4, (xpos * (p.KWO >> 4)) >> 3, X, p1, lane);
  if (!((DBG & 4) && (p.dbgf & 4))) tail_ln<NB>(acc, vec1, vec1 + D, red, w, lane, w * NB);
  // x1 stays in the accumulator registers through FFN1 (which accumulates elsewhere) and becomes,
  // with b2 added, the initial accumulator of FFN2: the fp32 residual stream never leaves registers.
  // Its 16-bit copy = FFN1 operand (X is free: every wave is past stage 1, LayerNorm took barriers).
  {
  const int mlx = fresh(ml), hix = fresh(hi);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = (w * NB + i) * 32 + 8 * g + 4 * hix;
      const float4 b = *reinterpret_cast<const float4*>(vec2 + 2 * D + n);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        *reinterpret_cast<u16x4*>(X + (rb * 32 + mlx) * pD + n * 2) =
            cvt4<T16>(acc[i][rb][4 * g], acc[i][rb][4 * g + 1], acc[i][rb][4 * g + 2], acc[i][rb][4 * g + 3]);
        acc[i][rb][4 * g + 0] += b.x; acc[i][rb][4 * g + 1] += b.y;
        acc[i][rb][4 * g + 2] += b.z; acc[i][rb][4 * g + 3] += b.w;
      }
    }
  }
  __syncthreads();

  // ---- stage 2: FFN1 + bias + ReLU -> Y (16 bit). DH/32 = 8 or 12 n-blocks over 8 waves: at d = 768
  // waves 0-3 own two blocks (w, w + 8), waves 4-7 one - a wave-uniform choice of instantiation
  // (no per-block conditions inside the pipelined loop)
  {
    const float* b1l = vec2 + 3 * D;
    auto ffn1_epi = [&](const f32x16 (&h)[2], int blk) {
      const int mlx = fresh(ml), hix = fresh(hi);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = blk * 32 + 8 * g + 4 * hix;
        const float4 b = *reinterpret_cast<const float4*>(b1l + n);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          *reinterpret_cast<u16x4*>(Y + (rb * 32 + mlx) * pH + n * 2) =
              cvt4<T16>(fmaxf(h[rb][4 * g] + b.x, 0.f), fmaxf(h[rb][4 * g + 1] + b.y, 0.f),
                        fmaxf(h[rb][4 * g + 2] + b.z, 0.f), fmaxf(h[rb][4 * g + 3] + b.w, 0.f));
      }
    };
    const int rot = (xpos * (D >> 4)) >> 3;
    // (two blocks = two passes over K with one accumulator pair: 64 fewer live registers than one
    // pass with two pairs, which spilled x1; the extra LDS operand reads are free here)
    f32x16 hacc[1][2];
    tail_gemm<T16, 1, 4, true, DBG>(hacc, p.w1_p, w, 8, D >> 4, rot, X, pD, lane);
    ffn1_epi(hacc[0], w);
    if (NB1 == 2 && w < 4) {
      tail_gemm<T16, 1, 4, true, DBG>(hacc, p.w1_p, w + 8, 8, D >> 4, rot, X, pD, lane);
      ffn1_epi(hacc[0], w + 8);
    }
  }
  __syncthreads();

  // ---- stage 3: (x1 + b2) + W2 hidden, LayerNorm
  tail_gemm<T16, NB, NB == 3 ? 4 : 8, false, DBG>(acc, p.w2_p, w * NB, 1, DH >> 4, (xpos * (DH >> 4)) >> 3, Y, pH, lane);
  if (!((DBG & 4) && (p.dbgf & 4))) tail_ln<NB>(acc, vec2, vec2 + D, red, w, lane, w * NB);
  const int mlo = fresh(ml), hio = fresh(hi);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (w * NB + i) * 32 + 8 * g + 4 * hio;
        const float a0 = acc[i][rb][4 * g], a1 = acc[i][rb][4 * g + 1], a2 = acc[i][rb][4 * g + 2],
                    a3 = acc[i][rb][4 * g + 3];
        if (mrow[rb] < p.M && !((DBG & 4) && (p.dbgf & 8))) {
          if (p.y32) *reinterpret_cast<float4*>(p.y32 + (int64_t)mrow[rb] * D + n) = make_float4(a0, a1, a2, a3);
          if (p.y16)
            *reinterpret_cast<u16x4*>(p.y16 + (int64_t)mrow[rb] * D + n) =
                p.y16_bf16 ? cvt4<BF16>(a0, a1, a2, a3) : cvt4<F16>(a0, a1, a2, a3);
        }
        if constexpr (SCORE)       // lin2 operand, in the head's own 16-bit type (X was last read in stage 2)
          *reinterpret_cast<u16x4*>(X + (rb * 32 + mlo) * pD + n * 2) = cvt4<TH>(a0, a1, a2, a3);
      }
  // score-head operands that are not weight streams: requested before the lin2 GEMM, used after it
  float4 hb[4], hw[4];
  float am = 0.f, cm = 0.f;
  int64_t o_idx = 0;
  if constexpr (SCORE) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = w * 32 + 8 * g + 4 * hi;
      hb[g] = *reinterpret_cast<const float4*>(p.bl + n);
      hw[g] = *reinterpret_cast<const float4*>(p.wl2 + n);
    }
    if (tid < 64) {
      const int64_t row0 = (int64_t)m0 + tid;
      const int64_t row = row0 < p.M ? row0 : p.M - 1;
      const vog_score_args& a = p.sc;      // same index arithmetic as score_kernel (elementwise.hip)
      const int N = a.nsrl * a.nppf;
      const int s = (int)(row / N), j = (int)(row % N);
      const int v = s / a.nfrm, f = s % a.nfrm;
      const int arg = j / a.nppf, pp = j % a.nppf;
      const int NP = a.nfrm * a.nppf;
      const int r = f * a.nppf + pp;
      o_idx = ((int64_t)v * a.nsrl + arg) * NP + r;
      const int b = v / a.nc_v, c = v % a.nc_v;
      int cmp;
      if (a.conc_type == VOG_CONC_TEMP) cmp = r / (a.nfrm0 * a.nppf0);
      else if (a.conc_type == VOG_CONC_SPAT) cmp = (r / a.nppf0) % a.ncmp;
      else cmp = c;
      const int lrow = a.nvl > 1 ? (b * a.nvl + c) : b;
      am = (float)a.arg_msk[(int64_t)lrow * a.nsrl + arg];
      cm = (float)a.cmp_msk[(int64_t)b * a.ncmp + cmp];
    }
  }
  if constexpr (SCORE) {
    __syncthreads();
    // ---- stage 4: lin2.0 + ReLU, lin2.2 as a row dot product, inverse regroup + masks
    f32x16 sacc[1][2];
    tail_gemm<TH, 1, 8, true, DBG>(sacc, p.wl_p, w, 1, D >> 4, (xpos * (D >> 4)) >> 3, X, pD, lane);
    float part[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = hb[g], ww = hw[g];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        part[rb] += fmaxf(sacc[0][rb][4 * g] + b.x, 0.f) * ww.x + fmaxf(sacc[0][rb][4 * g + 1] + b.y, 0.f) * ww.y +
                    fmaxf(sacc[0][rb][4 * g + 2] + b.z, 0.f) * ww.z + fmaxf(sacc[0][rb][4 * g + 3] + b.w, 0.f) * ww.w;
    }
    part[0] += __shfl_xor(part[0], 32); part[1] += __shfl_xor(part[1], 32);
    if (hi == 0) { red[w * 64 + ml] = part[0]; red[w * 64 + 32 + ml] = part[1]; }
    __syncthreads();
    if (tid < 64 && (int64_t)m0 + tid < p.M) {
      float logit = p.bl2[0];
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) logit += red[ww * 64 + tid];
      p.sc.outs[o_idx] = logit;
      p.sc.outs_eval[o_idx] = tail_sigmoid(logit) * am * cm;
    }
  }
}

template <typename T16, int NB, bool SCORE, int DBG = 0>
static int launch_tail(const TailParams& p, hipStream_t st) {
  constexpr int D = NB * 256, DH = D / 2;
  const int xcols = p.KWO > D ? p.KWO : D;
  const size_t lds = (size_t)64 * (xcols + 8) * 2 + (size_t)64 * (DH + 8) * 2 + (512 + 3 * D + DH) * sizeof(float);
  auto kern = tx_tail_kernel<T16, F16, NB, SCORE, DBG>;
  static bool attr_set = false;
  if (!attr_set) {
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_set = true;
  }
  if (lds > 160 * 1024) VOG_FAIL(-1, "fused encoder tail: %zu bytes of LDS needed", lds);
  ::vog::launch(kern, dim3(ceil_div(p.M, 64)), dim3(512), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}

int tx_tail_supported(int d, int dh, int kwo) {
  return (d == 512 || d == 768) && dh == d / 2 && kwo > 0 && (kwo % 64) == 0 && kwo <= 768;
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
  const bool score = a->score != nullptr;
  if (score) {
    VOG_CHECK_ARG(a->wl_p && a->bl && a->score->w2 && a->score->b2 && a->score->arg_msk && a->score->cmp_msk &&
                  a->score->outs && a->score->outs_eval && a->score->dh == 256 && a->head_dtype == VOG_F16);
    VOG_CHECK_ARG((int64_t)a->score->n_vid * a->score->nfrm * a->score->nsrl * a->score->nppf == a->M);
    p.wl_p = (const unsigned short*)a->wl_p; p.bl = a->bl; p.wl2 = a->score->w2; p.bl2 = a->score->b2;
    p.sc = *a->score;
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
