// Device side of txtail.hip (kernel bodies; also included by pair.hip, which fuses two bodies into one launch).
#pragma once
#include "common.h"

#ifndef VOG_TAIL_PF1
#define VOG_TAIL_PF1 4      // k-steps of weight prefetch in the FFN1 stage (one 32-column block per wave); 8 measured: 84 more bytes of scratch in the mul tail, 0.5-1 % slower at cfg 2 and cfg 4 (scratch/r4_pf.sh)
#endif

#ifndef VOG_TAIL_PRIME
#define VOG_TAIL_PRIME 0    // 1 (round 5): the first PF k-steps of a stage's weights are requested BEFORE the LayerNorm / epilogue /
#endif                      // barrier in front of it (a stage boundary no longer restarts the weight stream from an empty pipe)

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
  int nt_rows;   // non-temporal loads of the attention rows (large M)
  int xcds;      // > 0 (round 5): the row blocks run on the first `xcds` XCDs only (grid = ceil(blocks / xcds) * 8; block b sits on
                 // XCD b % 8): every XCD's L2 fetches the whole weight set from the fabric once, so 63 workgroups on 4 XCDs move
                 // half the fabric bytes of 63 workgroups on 8

  int dbgf;      // perf experiments only, read by the DBG & 4 instantiation: 1 no residual, 2 no attention staging, 4 no LayerNorm, 8 no outputs
  // round 6, hi + lo operands (SPLIT bodies): 16-bit remainders of the attention rows and of the three weight matrices (same
  // layouts), and of the output rows (y16_lo, optional)
  const unsigned short *attn16_lo, *wo_p_lo, *w1_p_lo, *w2_p_lo; unsigned short* y16_lo;
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
// the first PF k-steps of a stage's weight stream (what tail_gemm starts with), as a call of its own: issued ahead of the
// LayerNorm / epilogue / barrier in front of the stage (TailParams::prime)
template <int NBW, int PF, int DBG = 0>
__device__ __forceinline__ void tail_prime(u16x8 (&wq)[PF][NBW], const unsigned short* __restrict__ wp,
                                           int blk0, int blk_step, int KS, int rot, int lane) {
  constexpr int KST = (DBG & 1) ? 0 : 64;
  if constexpr (DBG & 4) return;
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    int k = j + rot; k = k >= KS ? k - KS : k;
#pragma unroll
    for (int i = 0; i < NBW; ++i)
      wq[j][i] = (reinterpret_cast<const u16x8*>(wp + ((int64_t)(blk0 + i * blk_step) * KS) * 512) + lane)[k * KST];
  }
}

template <typename TT, int NBW, int PF, bool ZERO, int DBG = 0, int RB = 2, bool PRIMED = false>
__device__ __forceinline__ void tail_gemm(f32x16 (&acc)[NBW][RB], const unsigned short* __restrict__ wp,
                                          int blk0, int blk_step, int KS, int rot,
                                          const unsigned char* xl, int pitch, int lane, u16x8 (*wq_in)[NBW] = nullptr) {
  constexpr int KST = (DBG & 1) ? 0 : 64;     // DBG 1 (perf experiments): every weight load hits the block's first KiB
  const int ml = lane & 31, hi = lane >> 5;
  const u16x8* wb[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    wb[i] = reinterpret_cast<const u16x8*>(wp + ((int64_t)(blk0 + i * blk_step) * KS) * 512) + lane;
    if constexpr (ZERO) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
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
    for (int i = 0; i < NBW; ++i) {
      if constexpr (PRIMED) wq[j][i] = wq_in[j][i];
      else wq[j][i] = wb[i][k * KST];
    }
  }
  const unsigned char* x0 = xl + ml * pitch + hi * 16;
  u16x8 xf[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) xf[rb] = *reinterpret_cast<const u16x8*>(x0 + rb * 32 * pitch + kk(0) * 32);
  auto step = [&](int j, int t, bool refill) {
    const int kn = kk(t + 1);                  // next k-step's activations (t + 1 == KS wraps to `rot`: in bounds)
    u16x8 nx[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) nx[rb] = *reinterpret_cast<const u16x8*>(x0 + rb * 32 * pitch + kn * 32);
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if constexpr (DBG & 2) {               // DBG 2: no matrix work (keeps the operands live)
          acc[i][rb][0] += __builtin_bit_cast(float, (unsigned)wq[j][i][rb] | ((unsigned)xf[rb][0] << 16));
        } else {
          acc[i][rb] = mfma32<TT>(wq[j][i], xf[rb], acc[i][rb]);
        }
      }
    }
    if (refill) {
      const int kl = kk(t + PF);
#pragma unroll
      for (int i = 0; i < NBW; ++i) wq[j][i] = wb[i][kl * KST];
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xf[rb] = nx[rb];
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

// The same stage with hi + lo operands (round 6): weights W + W_lo (two fragment streams), activations X + X_lo (two LDS images,
// same pitch): acc += W.X + W_lo.X + W.X_lo per k-step - fp32-grade operand precision at three MFMAs per product. For the tails
// whose OUTPUT feeds another attention layer of a checkpoint with sharp logits; same rotated k order.
template <typename TT, int NBW, int PF, bool ZERO, int RB>
__device__ __forceinline__ void tail_gemm_split(f32x16 (&acc)[NBW][RB], const unsigned short* __restrict__ wp,
                                                const unsigned short* __restrict__ wpl, int blk0, int blk_step, int KS, int rot,
                                                const unsigned char* xl, const unsigned char* xll, int pitch, int lane) {
  const int ml = lane & 31, hi = lane >> 5;
  const u16x8* wb[NBW]; const u16x8* wbl[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    wb[i] = reinterpret_cast<const u16x8*>(wp + ((int64_t)(blk0 + i * blk_step) * KS) * 512) + lane;
    wbl[i] = reinterpret_cast<const u16x8*>(wpl + ((int64_t)(blk0 + i * blk_step) * KS) * 512) + lane;
    if constexpr (ZERO) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][rb][r] = 0.f;
    }
  }
  auto kk = [&](int t) { const int k = t + rot; return k >= KS ? k - KS : k; };   // t < 2*KS - rot
  u16x8 wq[PF][NBW], wql[PF][NBW];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int k = kk(j);
#pragma unroll
    for (int i = 0; i < NBW; ++i) { wq[j][i] = wb[i][k * 64]; wql[j][i] = wbl[i][k * 64]; }
  }
  const unsigned char* x0 = xl + ml * pitch + hi * 16;
  const unsigned char* x0l = xll + ml * pitch + hi * 16;
  auto step = [&](int j, int t, bool refill) {
    const int kc = kk(t);
    u16x8 xf[RB], xfl[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      xf[rb] = *reinterpret_cast<const u16x8*>(x0 + rb * 32 * pitch + kc * 32);
      xfl[rb] = *reinterpret_cast<const u16x8*>(x0l + rb * 32 * pitch + kc * 32);
    }
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        acc[i][rb] = mfma32<TT>(wq[j][i], xf[rb], acc[i][rb]);
        acc[i][rb] = mfma32<TT>(wql[j][i], xf[rb], acc[i][rb]);
        acc[i][rb] = mfma32<TT>(wq[j][i], xfl[rb], acc[i][rb]);
      }
    }
    if (refill) {
      const int kl = kk(t + PF);
#pragma unroll
      for (int i = 0; i < NBW; ++i) { wq[j][i] = wb[i][kl * 64]; wql[j][i] = wbl[i][kl * 64]; }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  int t = 0;
#pragma unroll 1
  for (; t < KS - PF; t += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) step(j, t + j, true);
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) step(j, t + j, false);
}

// LayerNorm over n of the swapped accumulator tile of the whole workgroup (D columns spread over the
// 8 waves): two-pass statistics as layernorm_kernel (mean, then sum of squared deviations); gamma / beta
// come from LDS (prefetched at kernel start: no dependent global round trip in the epilogue).
template <int NB, int RB = 2>
__device__ __forceinline__ void tail_ln(f32x16 (&acc)[NB][RB], const float* gamma_l, const float* beta_l,
                                        float* red, int w, int lane, int nblk0) {
  constexpr int RW = 32 * RB;                  // rows of the workgroup (red: [8 waves][RW])
  constexpr int D = NB * 256;
  int ml = lane & 31, hi = lane >> 5;
  // opaque copies: keeps hipcc from sharing the 16 exchange addresses between the two LayerNorms of
  // the kernel (it kept them live - spilled - across two GEMM stages instead of re-deriving them)
  asm volatile("" : "+v"(ml), "+v"(hi));
  float s[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) s[rb] = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[rb] += acc[i][rb][r];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) s[rb] += __shfl_xor(s[rb], 32);
  if (hi == 0) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) red[w * RW + rb * 32 + ml] = s[rb];
  }
  __syncthreads();
  float mean[RB], rstd[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) t += red[ww * RW + rb * 32 + ml];
    mean[rb] = t / (float)D;
  }
  float q[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) q[rb] = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float dd = acc[i][rb][r] - mean[rb]; q[rb] += dd * dd; }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) q[rb] += __shfl_xor(q[rb], 32);
  __syncthreads();                              // every wave has read the sums
  if (hi == 0) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) red[w * RW + rb * 32 + ml] = q[rb];
  }
  __syncthreads();
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float t = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) t += red[ww * RW + rb * 32 + ml];
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
      for (int rb = 0; rb < RB; ++rb) {
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

// RB = row blocks of 32 per workgroup. 2 (default): 64 rows, one workgroup per CU (162 KB of LDS at d = 768). 1 (round 4, the
// "<= 80 KB / <= 128 VGPR" form round 3's verdict asked to be measured): 32 rows, the LayerNorm / bias vectors read from
// memory instead of LDS (76 KB), so that two workgroups share a CU and overlap each other's stage boundaries - at twice the
// weight bytes per row through the CU's vector memory path.
// SPLIT (round 6; RB = 1): hi + lo operands in the three GEMM stages (tail_gemm_split), second LDS images of the activations behind
// the first ones, outputs as y16 + y16_lo.
template <typename T16, typename TH, int NB, bool SCORE, int DBG = 0, int RB = 2, bool SPLIT = false>
struct TxTailBody {
  using Params = TailParams;
  static constexpr int THREADS = 512;
  static constexpr int ROWS = 32 * RB;
  static constexpr size_t lds_base(int kwo) {
    const int D_ = NB * 256, DH_ = D_ / 2, xcols = kwo > D_ ? kwo : D_;
    return (size_t)ROWS * (xcols + 8) * 2 + (size_t)ROWS * (DH_ + 8) * 2 + (size_t)8 * ROWS * sizeof(float) +
           (RB == 2 ? (size_t)(3 * D_ + DH_) * sizeof(float) : 0);
  }
  static constexpr size_t lds_bytes(int kwo) {
    const int D_ = NB * 256, DH_ = D_ / 2, xcols = kwo > D_ ? kwo : D_;
    return lds_base(kwo) + (SPLIT ? (size_t)ROWS * (xcols + 8) * 2 + (size_t)ROWS * (DH_ + 8) * 2 : 0);
  }
  static __device__ __forceinline__ void run(const TailParams& p, const BlockCtx& cx, unsigned char* smem) {
  constexpr int D = NB * 256, DH = D / 2;
  constexpr int NB1 = NB == 3 ? 2 : 1;              // FFN1 n-blocks per wave (DH/32 = 8 or 12 over 8 waves)
  const int tid = threadIdx.x, lane = tid & 63, ml = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int rblk = cx.bx;
  if (p.xcds > 0) {                                   // row blocks on the first `xcds` XCDs only (block b sits on XCD b % 8)
    const int x = cx.bx & 7;
    if (x >= p.xcds) return;
    rblk = (cx.bx >> 3) * p.xcds + x;
    if (rblk * ROWS >= p.M) return;
  }
  const int m0 = rblk * ROWS;
  const int xcols = p.KWO > D ? p.KWO : D;
  unsigned char* X = smem;
  unsigned char* Y = X + ROWS * (xcols + 8) * 2;
  float* red = reinterpret_cast<float*>(Y + ROWS * (DH + 8) * 2);    // [8 waves][ROWS rows]
  float* vec2 = red + 8 * ROWS;                     // ln2 gamma, ln2 beta, b2 [D each], b1 [DH]  (RB == 2 only)
  float* vec1 = reinterpret_cast<float*>(Y);        // ln1 gamma, ln1 beta [D each]: parked in Y until FFN1 writes it (RB == 2 only)
  unsigned char* XL = smem + lds_base(p.KWO);        // SPLIT: the remainder images of X and Y
  unsigned char* YL = XL + ROWS * (xcols + 8) * 2;
  // the epilogue vectors: staged in LDS (RB == 2) or read where they are (RB == 1: L1 / L2 hits, 10 KB less LDS)
  const float* g1p = RB == 2 ? vec1 : p.ln1g;
  const float* b1np = RB == 2 ? vec1 + D : p.ln1b;
  const float* g2p = RB == 2 ? vec2 : p.ln2g;
  const float* b2np = RB == 2 ? vec2 + D : p.ln2b;
  const float* fb2p = RB == 2 ? vec2 + 2 * D : p.b2;
  const float* fb1p = RB == 2 ? vec2 + 3 * D : p.b1;
  const int p1 = (p.KWO + 8) * 2, pD = (D + 8) * 2, pH = (DH + 8) * 2;
  // position of this workgroup among the ones that share its XCD's L2 (block b runs on XCD b % 8;
  // speed only): staggers the k order of the weight streams
  const int xpos = (cx.bx >> 3) & 7;
  constexpr bool PRIME = VOG_TAIL_PRIME != 0;
#ifndef VOG_TAIL_PFA3
#define VOG_TAIL_PFA3 3      // k-steps of weight prefetch in the Wo / FFN2 stages at d = 768. Measured (scratch/r6_ah.sh, r6_ai.sh): 2: 49.6 /
                             // 266 us (cfg 2 / cfg 4 mul tail), 3: 49.2 / 268, 4 (rounds 2-6): 50.3 / 273, 6: 52.0 / 285 - the deeper sets
                             // cost spills (48 B of scratch at 4); cfg 4 5535 -> 5585 queries/s with 3, cfg 2 equal. Same k order:
                             // bit-identical. Needs (kwo / 16) % 3 == 0 (tx_tail_supported: kwo % 192 == 0 at d = 768)
#endif
  constexpr int PF_WO = NB == 3 ? VOG_TAIL_PFA3 : 6, PF_W2 = NB == 3 ? VOG_TAIL_PFA3 : 8;
  u16x8 wq_a[PF_WO][NB];                              // primed weights of the NB-block stages (Wo, then W2)
  u16x8 wq_b[PF_W2][NB];
  u16x8 wq_1[VOG_TAIL_PF1][1];                        // ... of FFN1's pass(es)
  u16x8 wq_s[8][1];                                   // ... of lin2
  if constexpr (PRIME) tail_prime<NB, PF_WO, DBG>(wq_a, p.wo_p, w * NB, 1, p.KWO >> 4, (xpos * (p.KWO >> 4)) >> 3, lane);

  // ---- stage 0: attention rows + the epilogue vectors -> LDS; residual rows -> accumulators.
  // Everything the chain will need from memory besides the weight streams is requested here, in one
  // round trip: no epilogue below waits for a global load.
  {
    const int cpr = ((DBG & 4) && (p.dbgf & 2)) ? 0 : (p.KWO >> 3);
    // Round 6: ALL of a thread's row chunks (<= 12 x 16 bytes at 64 rows x 768 columns) are requested before the first one is
    // stored. The loop this replaces (runtime trip count: load, wait, LDS store, next) paid one memory round trip per chunk -
    // most of the ~17 us of "skeleton" the ablations of round 4 found around the GEMM stages (profiles/round4_tail_ablation_and_
    // ingest.md). Indices past the end are clamped (no conditional load), their stores skipped.
    constexpr int SIT = ROWS * (768 / 8) / 512;            // kwo <= 768
    const int nchk = ROWS * cpr;
    if (cpr > 0) {
      u32x4 stg[SIT];
      u32x4 stgl[SPLIT ? SIT : 1];
      int ldsoff[SIT];
#pragma unroll
      for (int it = 0; it < SIT; ++it) {
        int idx = tid + it * 512;
        idx = idx < nchk ? idx : nchk - 1;
        const int r = idx / cpr, c = idx - r * cpr;
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;
        // (read once: with p.nt_rows the load is non-temporal, so that at many row blocks per XCD - 156 at cfg 4 - the streamed
        // activation rows do not displace the 2.75 MB of weights every workgroup of the XCD re-reads from its L2)
        const u32x4* src = reinterpret_cast<const u32x4*>(p.attn16 + (int64_t)m * p.KWO + c * 8);
        stg[it] = p.nt_rows ? __builtin_nontemporal_load(src) : *src;
        if constexpr (SPLIT) stgl[it] = *reinterpret_cast<const u32x4*>(p.attn16_lo + (int64_t)m * p.KWO + c * 8);
        ldsoff[it] = r * p1 + c * 16;
      }
#pragma unroll
      for (int it = 0; it < SIT; ++it) {
        if (tid + it * 512 < nchk) {
          *reinterpret_cast<u32x4*>(X + ldsoff[it]) = stg[it];
          if constexpr (SPLIT) *reinterpret_cast<u32x4*>(XL + ldsoff[it]) = stgl[it];
        }
      }
    }
    if constexpr (RB == 2) {
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
  }
  int mrow[RB];
  f32x16 acc[NB][RB];
  {
    const float* rp[RB]; const float* lp[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
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
        for (int rb = 0; rb < RB; ++rb) {
          f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
          if (!((DBG & 4) && (p.dbgf & 1))) {
            const f32x4* src = reinterpret_cast<const f32x4*>(in_lang ? lp[rb] + (n - p.rv_dv) : rp[rb] + n);
            r = p.nt_rows ? __builtin_nontemporal_load(src) : *src;      // (streamed once at large M, like the attention rows)
          }
          acc[i][rb][4 * g + 0] = r[0]; acc[i][rb][4 * g + 1] = r[1];
          acc[i][rb][4 * g + 2] = r[2]; acc[i][rb][4 * g + 3] = r[3];
        }
      }
  }
  __syncthreads();

  // ---- stage 1: x + attn Wo^T, LayerNorm
  if constexpr (SPLIT) tail_gemm_split<T16, NB, 4, false, RB>(acc, p.wo_p, p.wo_p_lo, w * NB, 1, p.KWO >> 4, (xpos * (p.KWO >> 4)) >> 3, X, XL, p1, lane);
  else
  tail_gemm<T16, NB, PF_WO, false, DBG, RB, PRIME>(acc, p.wo_p, w * NB, 1, p.KWO >> 4, (xpos * (p.KWO >> 4)) >> 3, X, p1, lane, wq_a);
  if constexpr (PRIME) tail_prime<1, VOG_TAIL_PF1, DBG>(wq_1, p.w1_p, w, 8, D >> 4, (xpos * (D >> 4)) >> 3, lane);
  if (!((DBG & 4) && (p.dbgf & 4))) tail_ln<NB, RB>(acc, g1p, b1np, red, w, lane, w * NB);
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
      const float4 b = *reinterpret_cast<const float4*>(fb2p + n);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const u16x4 xh = cvt4<T16>(acc[i][rb][4 * g], acc[i][rb][4 * g + 1], acc[i][rb][4 * g + 2], acc[i][rb][4 * g + 3]);
        *reinterpret_cast<u16x4*>(X + (rb * 32 + mlx) * pD + n * 2) = xh;
        if constexpr (SPLIT)
          *reinterpret_cast<u16x4*>(XL + (rb * 32 + mlx) * pD + n * 2) =
              cvt4<T16>(acc[i][rb][4 * g] - from16<T16>(xh[0]), acc[i][rb][4 * g + 1] - from16<T16>(xh[1]),
                        acc[i][rb][4 * g + 2] - from16<T16>(xh[2]), acc[i][rb][4 * g + 3] - from16<T16>(xh[3]));
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
    const float* b1l = fb1p;
    auto ffn1_epi = [&](const f32x16 (&h)[RB], int blk) {
      const int mlx = fresh(ml), hix = fresh(hi);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = blk * 32 + 8 * g + 4 * hix;
        const float4 b = *reinterpret_cast<const float4*>(b1l + n);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const float h0 = relu_nan(h[rb][4 * g] + b.x), h1 = relu_nan(h[rb][4 * g + 1] + b.y),
                      h2 = relu_nan(h[rb][4 * g + 2] + b.z), h3 = relu_nan(h[rb][4 * g + 3] + b.w);
          const u16x4 yh = cvt4<T16>(h0, h1, h2, h3);
          *reinterpret_cast<u16x4*>(Y + (rb * 32 + mlx) * pH + n * 2) = yh;
          if constexpr (SPLIT)
            *reinterpret_cast<u16x4*>(YL + (rb * 32 + mlx) * pH + n * 2) =
                cvt4<T16>(h0 - from16<T16>(yh[0]), h1 - from16<T16>(yh[1]), h2 - from16<T16>(yh[2]), h3 - from16<T16>(yh[3]));
        }
      }
    };
    const int rot = (xpos * (D >> 4)) >> 3;
    // (two blocks = two passes over K with one accumulator pair: 64 fewer live registers than one
    // pass with two pairs, which spilled x1; the extra LDS operand reads are free here)
    f32x16 hacc[1][RB];
    if constexpr (SPLIT) tail_gemm_split<T16, 1, 4, true, RB>(hacc, p.w1_p, p.w1_p_lo, w, 8, D >> 4, rot, X, XL, pD, lane);
    else
    tail_gemm<T16, 1, VOG_TAIL_PF1, true, DBG, RB, PRIME>(hacc, p.w1_p, w, 8, D >> 4, rot, X, pD, lane, wq_1);
    if (NB1 == 2 && w < 4) {
      if constexpr (PRIME) tail_prime<1, VOG_TAIL_PF1, DBG>(wq_1, p.w1_p, w + 8, 8, D >> 4, rot, lane);
      ffn1_epi(hacc[0], w);
      if constexpr (SPLIT) tail_gemm_split<T16, 1, 4, true, RB>(hacc, p.w1_p, p.w1_p_lo, w + 8, 8, D >> 4, rot, X, XL, pD, lane);
      else
      tail_gemm<T16, 1, VOG_TAIL_PF1, true, DBG, RB, PRIME>(hacc, p.w1_p, w + 8, 8, D >> 4, rot, X, pD, lane, wq_1);
      if constexpr (PRIME) tail_prime<NB, PF_W2, DBG>(wq_b, p.w2_p, w * NB, 1, DH >> 4, (xpos * (DH >> 4)) >> 3, lane);
      ffn1_epi(hacc[0], w + 8);
    } else {
      if constexpr (PRIME) tail_prime<NB, PF_W2, DBG>(wq_b, p.w2_p, w * NB, 1, DH >> 4, (xpos * (DH >> 4)) >> 3, lane);
      ffn1_epi(hacc[0], w);
    }
  }
  __syncthreads();

  // ---- stage 3: (x1 + b2) + W2 hidden, LayerNorm
  if constexpr (SPLIT) tail_gemm_split<T16, NB, 4, false, RB>(acc, p.w2_p, p.w2_p_lo, w * NB, 1, DH >> 4, (xpos * (DH >> 4)) >> 3, Y, YL, pH, lane);
  else
  tail_gemm<T16, NB, PF_W2, false, DBG, RB, PRIME>(acc, p.w2_p, w * NB, 1, DH >> 4, (xpos * (DH >> 4)) >> 3, Y, pH, lane, wq_b);
  if constexpr (PRIME && SCORE) tail_prime<1, 8, DBG>(wq_s, p.wl_p, w, 1, D >> 4, (xpos * (D >> 4)) >> 3, lane);
  if (!((DBG & 4) && (p.dbgf & 4))) tail_ln<NB, RB>(acc, g2p, b2np, red, w, lane, w * NB);
  const int mlo = fresh(ml), hio = fresh(hi);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (w * NB + i) * 32 + 8 * g + 4 * hio;
        const float a0 = acc[i][rb][4 * g], a1 = acc[i][rb][4 * g + 1], a2 = acc[i][rb][4 * g + 2],
                    a3 = acc[i][rb][4 * g + 3];
        if (mrow[rb] < p.M && !((DBG & 4) && (p.dbgf & 8))) {
          if (p.y32) *reinterpret_cast<float4*>(p.y32 + (int64_t)mrow[rb] * D + n) = make_float4(a0, a1, a2, a3);
          if (p.y16) {
            const u16x4 yh = p.y16_bf16 ? cvt4<BF16>(a0, a1, a2, a3) : cvt4<F16>(a0, a1, a2, a3);
            *reinterpret_cast<u16x4*>(p.y16 + (int64_t)mrow[rb] * D + n) = yh;
            if constexpr (SPLIT) {
              if (p.y16_lo)
                *reinterpret_cast<u16x4*>(p.y16_lo + (int64_t)mrow[rb] * D + n) =
                    p.y16_bf16 ? cvt4<BF16>(a0 - from16<BF16>(yh[0]), a1 - from16<BF16>(yh[1]), a2 - from16<BF16>(yh[2]), a3 - from16<BF16>(yh[3]))
                               : cvt4<F16>(a0 - from16<F16>(yh[0]), a1 - from16<F16>(yh[1]), a2 - from16<F16>(yh[2]), a3 - from16<F16>(yh[3]));
            }
          }
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
    if (tid < ROWS) {
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
    f32x16 sacc[1][RB];
    tail_gemm<TH, 1, 8, true, DBG, RB, PRIME>(sacc, p.wl_p, w, 1, D >> 4, (xpos * (D >> 4)) >> 3, X, pD, lane, wq_s);
    float part[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) part[rb] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = hb[g], ww = hw[g];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        part[rb] += relu_nan(sacc[0][rb][4 * g] + b.x) * ww.x + relu_nan(sacc[0][rb][4 * g + 1] + b.y) * ww.y +
                    relu_nan(sacc[0][rb][4 * g + 2] + b.z) * ww.z + relu_nan(sacc[0][rb][4 * g + 3] + b.w) * ww.w;
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) part[rb] += __shfl_xor(part[rb], 32);
    if (hi == 0) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) red[w * ROWS + rb * 32 + ml] = part[rb];
    }
    __syncthreads();
    if (tid < ROWS && (int64_t)m0 + tid < p.M) {
      float logit = p.bl2[0];
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) logit += red[ww * ROWS + tid];
      p.sc.outs[o_idx] = logit;
      const float ev = tail_sigmoid(logit) * am * cm;
      p.sc.outs_eval[o_idx] = ev;
    }
  }
}
};

template <typename T16, typename TH, int NB, bool SCORE, int DBG = 0>
__global__ __launch_bounds__(512) void tx_tail_kernel(TailParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tt_smem[];
  TxTailBody<T16, TH, NB, SCORE, DBG>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, tt_smem);
}

// hi + lo operands (round 6): 32 rows per workgroup (the second LDS images take the room of the other 32)
template <typename T16, typename TH, int NB>
__global__ __launch_bounds__(512) void tx_tail_split_kernel(TailParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tts_smem[];
  TxTailBody<T16, TH, NB, false, 0, 1, true>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, tts_smem);
}

// the 32-row form: <= 128 registers (4 waves per SIMD), two workgroups per CU
template <typename T16, typename TH, int NB, bool SCORE>
__global__ __launch_bounds__(512, 4) void tx_tail32_kernel(TailParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tt32_smem[];
  TxTailBody<T16, TH, NB, SCORE, 0, 1>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, tt32_smem);
}

}  // namespace vog
