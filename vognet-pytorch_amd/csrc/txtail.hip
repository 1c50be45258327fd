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
  float* x1_scratch;
  int M;
};

// One GEMM stage of the chain: acc[i][rb] (32 columns n x 32 rows m, swapped) += W_blk(i) . X^T over
// KS k-steps of 16. Weight fragments of n-block b start at wp + b*KS*512 halfwords.
template <typename TT, int NBW>
__device__ __forceinline__ void tail_gemm(f32x16 (&acc)[NBW][2], const unsigned short* __restrict__ wp,
                                          int blk0, int blk_step, int cnt, int KS,
                                          const unsigned char* xl, int pitch, int lane) {
  constexpr int PF = 4;
  const int ml = lane & 31, hi = lane >> 5;
  const u16x8* wb[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    wb[i] = reinterpret_cast<const u16x8*>(wp + ((int64_t)(blk0 + i * blk_step) * KS) * 512) + lane;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][rb][r] = 0.f;
  }
  u16x8 wq[PF][NBW];
#pragma unroll
  for (int j = 0; j < PF; ++j)
#pragma unroll
    for (int i = 0; i < NBW; ++i)
      if (i < cnt) wq[j][i] = wb[i][j * 64];
  const unsigned char* x0 = xl + ml * pitch + hi * 16;
  const unsigned char* x1 = x0 + 32 * pitch;
  u16x8 xf0 = *reinterpret_cast<const u16x8*>(x0), xf1 = *reinterpret_cast<const u16x8*>(x1);
  for (int ks = 0; ks < KS; ks += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int kn = ks + j + 1;
      u16x8 n0 = xf0, n1 = xf1;
      if (kn < KS) {
        n0 = *reinterpret_cast<const u16x8*>(x0 + kn * 32);
        n1 = *reinterpret_cast<const u16x8*>(x1 + kn * 32);
      }
#pragma unroll
      for (int i = 0; i < NBW; ++i)
        if (i < cnt) {
          acc[i][0] = mfma32<TT>(wq[j][i], xf0, acc[i][0]);
          acc[i][1] = mfma32<TT>(wq[j][i], xf1, acc[i][1]);
        }
      const int kl = ks + j + PF;
      if (kl < KS) {
#pragma unroll
        for (int i = 0; i < NBW; ++i)
          if (i < cnt) wq[j][i] = wb[i][kl * 64];
      }
      xf0 = n0; xf1 = n1;
    }
  }
}

// LayerNorm over n of the swapped accumulator tile of the whole workgroup (D columns spread over the
// 8 waves): two-pass statistics as layernorm_kernel (mean, then sum of squared deviations).
template <int NB>
__device__ __forceinline__ void tail_ln(f32x16 (&acc)[NB][2], const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float* red0, float* red1,
                                        int w, int lane, int nblk0) {
  constexpr int D = NB * 256;
  const int ml = lane & 31, hi = lane >> 5;
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[rb] += acc[i][rb][r];
  s[0] += __shfl_xor(s[0], 32); s[1] += __shfl_xor(s[1], 32);
  if (hi == 0) { red0[w * 64 + ml] = s[0]; red0[w * 64 + 32 + ml] = s[1]; }
  __syncthreads();
  float mean[2], rstd[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) t += red0[ww * 64 + rb * 32 + ml];
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
  if (hi == 0) { red1[w * 64 + ml] = q[0]; red1[w * 64 + 32 + ml] = q[1]; }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) t += red1[ww * 64 + rb * 32 + ml];
    rstd[rb] = 1.0f / sqrtf(t / (float)D + 1e-5f);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = (nblk0 + i) * 32 + 8 * g + 4 * hi;
      const float4 gm = *reinterpret_cast<const float4*>(gamma + n);
      const float4 bt = *reinterpret_cast<const float4*>(beta + n);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        acc[i][rb][4 * g + 0] = (acc[i][rb][4 * g + 0] - mean[rb]) * rstd[rb] * gm.x + bt.x;
        acc[i][rb][4 * g + 1] = (acc[i][rb][4 * g + 1] - mean[rb]) * rstd[rb] * gm.y + bt.y;
        acc[i][rb][4 * g + 2] = (acc[i][rb][4 * g + 2] - mean[rb]) * rstd[rb] * gm.z + bt.z;
        acc[i][rb][4 * g + 3] = (acc[i][rb][4 * g + 3] - mean[rb]) * rstd[rb] * gm.w + bt.w;
      }
      if (g == 3) asm volatile("" ::: "memory");
    }
}

template <typename TT>
__device__ __forceinline__ u16x4 cvt4(float a, float b, float c, float d) {
  return u16x4{to16<TT>(a), to16<TT>(b), to16<TT>(c), to16<TT>(d)};
}

__device__ __forceinline__ float tail_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <typename T16, typename TH, int NB, bool SCORE>
__global__ __launch_bounds__(512) void tx_tail_kernel(TailParams p) {
  constexpr int D = NB * 256, DH = D / 2;
  constexpr int NB1 = NB == 3 ? 2 : 1;              // FFN1 n-blocks per wave (DH/32 = 8 or 12 over 8 waves)
  constexpr bool X1_REGS = NB <= 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, ml = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 64;
  const int xcols = p.KWO > D ? p.KWO : D;
  unsigned char* X = smem;
  unsigned char* Y = X + 64 * (xcols + 8) * 2;
  float* red0 = reinterpret_cast<float*>(Y + 64 * (DH + 8) * 2);     // [8 waves][64 rows]
  float* red1 = red0 + 512;
  const int p1 = (p.KWO + 8) * 2, pD = (D + 8) * 2, pH = (DH + 8) * 2;

  // ---- stage 0: the attention output rows of this block -> LDS
  {
    const int cpr = p.KWO >> 3;
    for (int idx = tid; idx < 64 * cpr; idx += 512) {
      const int r = idx / cpr, c = idx - r * cpr;
      int m = m0 + r;
      m = m < p.M ? m : p.M - 1;
      const uint4 v = *reinterpret_cast<const uint4*>(p.attn16 + (int64_t)m * p.KWO + c * 8);
      *reinterpret_cast<uint4*>(X + r * p1 + c * 16) = v;
    }
  }
  // residual row pointers (independent of the GEMM: issued now, used in the epilogue)
  int mrow[2], mcl[2];
  const float* rp[2]; const float* lp[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    mrow[rb] = m0 + rb * 32 + ml;
    mcl[rb] = mrow[rb] < p.M ? mrow[rb] : p.M - 1;
    if (p.res_vis) {
      const int N = p.rv_nsrl * p.rv_nppf;
      const int s = mcl[rb] / N, j = mcl[rb] - s * N;
      const int a = j / p.rv_nppf, pp = j - a * p.rv_nppf;
      const int v = s / p.rv_nfrm;
      const int lv = p.rv_lpv ? v : v / p.rv_ncv;
      rp[rb] = p.res_vis + ((int64_t)s * p.rv_nppf + pp) * p.rv_dv;
      lp[rb] = p.res_lang + ((int64_t)lv * p.rv_nsrl + a) * p.rv_dl;
    } else {
      rp[rb] = p.residual + (int64_t)mcl[rb] * p.ldr;
      lp[rb] = rp[rb];
    }
  }
  __syncthreads();

  // ---- stage 1: Wo, + residual, LayerNorm
  f32x16 acc[NB][2];
  tail_gemm<T16, NB>(acc, p.wo_p, w * NB, 1, NB, p.KWO >> 4, X, p1, lane);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = (w * NB + i) * 32 + 8 * g + 4 * hi;
      const bool in_lang = p.res_vis && n >= p.rv_dv;       // wave-uniform per (i): dv % 32 == 0
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const float4 r = *reinterpret_cast<const float4*>(in_lang ? lp[rb] + (n - p.rv_dv) : rp[rb] + n);
        acc[i][rb][4 * g + 0] += r.x; acc[i][rb][4 * g + 1] += r.y;
        acc[i][rb][4 * g + 2] += r.z; acc[i][rb][4 * g + 3] += r.w;
      }
      if (g == 3) asm volatile("" ::: "memory");     // keep at most one n-block of loads in flight (registers)
    }
  tail_ln<NB>(acc, p.ln1g, p.ln1b, red0, red1, w, lane, w * NB);
  // x1: fp32 copy for the second residual, 16-bit copy = FFN1 operand (X is free: every wave is
  // past stage 1, the LayerNorm exchanged through two barriers)
  f32x16 x1r[X1_REGS ? NB : 1][2];
  float* xs = p.x1_scratch + (int64_t)blockIdx.x * (NB * 2 * 16) * 512;   // uniform base + lane offset: scalar-base addressing
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      if constexpr (X1_REGS) x1r[i][rb] = acc[i][rb];
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) (xs + ((i * 2 + rb) * 16 + r) * 512)[(unsigned)tid] = acc[i][rb][r];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (w * NB + i) * 32 + 8 * g + 4 * hi;
        *reinterpret_cast<u16x4*>(X + (rb * 32 + ml) * pD + n * 2) =
            cvt4<T16>(acc[i][rb][4 * g], acc[i][rb][4 * g + 1], acc[i][rb][4 * g + 2], acc[i][rb][4 * g + 3]);
      }
    }
  __syncthreads();

  // ---- stage 2: FFN1 + bias + ReLU -> Y (16 bit)
  {
    f32x16 hacc[NB1][2];
    const int cnt = NB == 3 ? (w < 4 ? 2 : 1) : 1;
    tail_gemm<T16, NB1>(hacc, p.w1_p, w, 8, cnt, D >> 4, X, pD, lane);
#pragma unroll
    for (int i = 0; i < NB1; ++i)
      if (i < cnt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = (w + 8 * i) * 32 + 8 * g + 4 * hi;
          const float4 b = *reinterpret_cast<const float4*>(p.b1 + n);
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
            *reinterpret_cast<u16x4*>(Y + (rb * 32 + ml) * pH + n * 2) =
                cvt4<T16>(fmaxf(hacc[i][rb][4 * g] + b.x, 0.f), fmaxf(hacc[i][rb][4 * g + 1] + b.y, 0.f),
                          fmaxf(hacc[i][rb][4 * g + 2] + b.z, 0.f), fmaxf(hacc[i][rb][4 * g + 3] + b.w, 0.f));
        }
      }
  }
  __syncthreads();

  // ---- stage 3: FFN2 + bias + x1, LayerNorm
  tail_gemm<T16, NB>(acc, p.w2_p, w * NB, 1, NB, DH >> 4, Y, pH, lane);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = (w * NB + i) * 32 + 8 * g + 4 * hi;
      const float4 b = *reinterpret_cast<const float4*>(p.b2 + n);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float r0, r1, r2, r3;
        if constexpr (X1_REGS) {
          r0 = x1r[i][rb][4 * g]; r1 = x1r[i][rb][4 * g + 1]; r2 = x1r[i][rb][4 * g + 2]; r3 = x1r[i][rb][4 * g + 3];
        } else {
          const float* q = xs + ((i * 2 + rb) * 16 + 4 * g) * 512;
          r0 = q[(unsigned)tid]; r1 = (q + 512)[(unsigned)tid]; r2 = (q + 1024)[(unsigned)tid]; r3 = (q + 1536)[(unsigned)tid];
        }
        acc[i][rb][4 * g + 0] += b.x + r0; acc[i][rb][4 * g + 1] += b.y + r1;
        acc[i][rb][4 * g + 2] += b.z + r2; acc[i][rb][4 * g + 3] += b.w + r3;
      }
      if (g == 3) asm volatile("" ::: "memory");
    }
  tail_ln<NB>(acc, p.ln2g, p.ln2b, red0, red1, w, lane, w * NB);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (w * NB + i) * 32 + 8 * g + 4 * hi;
        const float a0 = acc[i][rb][4 * g], a1 = acc[i][rb][4 * g + 1], a2 = acc[i][rb][4 * g + 2],
                    a3 = acc[i][rb][4 * g + 3];
        if (mrow[rb] < p.M) {
          if (p.y32) *reinterpret_cast<float4*>(p.y32 + (int64_t)mrow[rb] * D + n) = make_float4(a0, a1, a2, a3);
          if (p.y16)
            *reinterpret_cast<u16x4*>(p.y16 + (int64_t)mrow[rb] * D + n) =
                p.y16_bf16 ? cvt4<BF16>(a0, a1, a2, a3) : cvt4<F16>(a0, a1, a2, a3);
        }
        if constexpr (SCORE)       // lin2 operand, in the head's own 16-bit type (X was last read in stage 2)
          *reinterpret_cast<u16x4*>(X + (rb * 32 + ml) * pD + n * 2) = cvt4<TH>(a0, a1, a2, a3);
      }
  if constexpr (SCORE) {
    __syncthreads();
    // ---- stage 4: lin2.0 + ReLU, lin2.2 as a row dot product, inverse regroup + masks
    f32x16 sacc[1][2];
    tail_gemm<TH, 1>(sacc, p.wl_p, w, 1, 1, D >> 4, X, pD, lane);
    float part[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = w * 32 + 8 * g + 4 * hi;
      const float4 b = *reinterpret_cast<const float4*>(p.bl + n);
      const float4 ww = *reinterpret_cast<const float4*>(p.wl2 + n);
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        part[rb] += fmaxf(sacc[0][rb][4 * g] + b.x, 0.f) * ww.x + fmaxf(sacc[0][rb][4 * g + 1] + b.y, 0.f) * ww.y +
                    fmaxf(sacc[0][rb][4 * g + 2] + b.z, 0.f) * ww.z + fmaxf(sacc[0][rb][4 * g + 3] + b.w, 0.f) * ww.w;
    }
    part[0] += __shfl_xor(part[0], 32); part[1] += __shfl_xor(part[1], 32);
    if (hi == 0) { red0[w * 64 + ml] = part[0]; red0[w * 64 + 32 + ml] = part[1]; }
    __syncthreads();
    if (tid < 64) {
      const int64_t row = (int64_t)m0 + tid;
      if (row < p.M) {
        float logit = p.bl2[0];
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) logit += red0[ww * 64 + tid];
        const vog_score_args& a = p.sc;      // same index arithmetic as score_kernel (elementwise.hip)
        const int N = a.nsrl * a.nppf;
        const int s = (int)(row / N), j = (int)(row % N);
        const int v = s / a.nfrm, f = s % a.nfrm;
        const int arg = j / a.nppf, pp = j % a.nppf;
        const int NP = a.nfrm * a.nppf;
        const int r = f * a.nppf + pp;
        const int64_t o = ((int64_t)v * a.nsrl + arg) * NP + r;
        const int b = v / a.nc_v, c = v % a.nc_v;
        int cmp;
        if (a.conc_type == VOG_CONC_TEMP) cmp = r / (a.nfrm0 * a.nppf0);
        else if (a.conc_type == VOG_CONC_SPAT) cmp = (r / a.nppf0) % a.ncmp;
        else cmp = c;
        const int lrow = a.nvl > 1 ? (b * a.nvl + c) : b;
        const float am = (float)a.arg_msk[(int64_t)lrow * a.nsrl + arg];
        const float cm = (float)a.cmp_msk[(int64_t)b * a.ncmp + cmp];
        a.outs[o] = logit;
        a.outs_eval[o] = tail_sigmoid(logit) * am * cm;
      }
    }
  }
}

template <typename T16, int NB, bool SCORE>
static int launch_tail(const TailParams& p, hipStream_t st) {
  constexpr int D = NB * 256, DH = D / 2;
  const int xcols = p.KWO > D ? p.KWO : D;
  const size_t lds = (size_t)64 * (xcols + 8) * 2 + (size_t)64 * (DH + 8) * 2 + 2 * 512 * sizeof(float);
  auto kern = tx_tail_kernel<T16, F16, NB, SCORE>;
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
  return d > 512 ? (int64_t)ceil_div(M, 64) * (d / 256 * 2 * 16) * 512 * 4 : 0;
}

int tx_tail_run(const vog_tx_tail_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->attn16 && a->wo_p && a->w1_p && a->w2_p && a->ln1g && a->ln1b && a->b1 && a->b2 &&
                a->ln2g && a->ln2b && a->M > 0);
  if (!tx_tail_supported(a->d, a->dh, a->kwo))
    VOG_FAIL(-1, "fused encoder tail: unsupported shape d=%d dh=%d kwo=%d (d in {512,768}, dh = d/2, kwo %% 64 == 0)",
             a->d, a->dh, a->kwo);
  VOG_CHECK_ARG((a->residual != nullptr) != (a->res_vislang != nullptr));
  VOG_CHECK_ARG(a->y32 || a->y16 || a->score);
  VOG_CHECK_ARG(a->d <= 512 || a->x1_scratch);
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
  p.x1_scratch = a->x1_scratch; p.M = a->M;
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
