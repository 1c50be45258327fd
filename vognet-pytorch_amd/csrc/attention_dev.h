// Device side of attention.hip (kernel bodies; also included by pair.hip, which fuses two bodies into one launch).
#pragma once
#include <stdlib.h>
#include "common.h"

namespace vog {

struct AttnParams {
  const unsigned short* q; const unsigned short* k; const unsigned short* vt;
  unsigned short* out;
  const float* u; const float* pe_b;
  int S, N, H, dp, npad, use_rel, n_box, seq_per_vid, NP;
  float inv_scale;
  int* guard;       // attn_tile2_kernel: set to 1 when its fixed-reference softmax left its safe range;
                    // attn_tile_kernel: when non-null, run only if *guard != 0 (fallback pass)
  int guard_precleared;   // host side only: *guard is already 0 (no clearing launch)
  // round 6, hi + lo operands (SPLIT kernels): q / k = q + q_lo, k + k_lo with q_lo = t16(x - t16(x)) in the same fragment
  // order; Q.K^T = q.k + q_lo.k + q.k_lo (three MFMAs, the lo x lo term is below fp32 resolution). out_lo (optional): the
  // 16-bit remainder of the output rows (the Wo GEMM of a split tail reads both).
  const unsigned short* q_lo; const unsigned short* k_lo; unsigned short* out_lo;
  // optional: 8 words (zeroed by the caller); their maximum = the largest |scaled logit| this launch saw, as the bits of a
  // non-negative float (publish_logit_max)
  unsigned int* logit_max;
};

// S^T += K_blk . Q^T for one k-step; SPLIT adds the two cross terms of the hi + lo operands
template <typename T16, bool SPLIT>
__device__ __forceinline__ f32x16 qk_mfma(u16x8 kh, u16x8 qh, u16x8 kl, u16x8 ql, f32x16 s) {
  s = mfma32<T16>(kh, qh, s);
  if constexpr (SPLIT) {
    s = mfma32<T16>(kl, qh, s);
    s = mfma32<T16>(kh, ql, s);
  }
  return s;
}
// hi + lo of an fp32 vector piece
template <typename T16>
__device__ __forceinline__ void split16(float v, unsigned short& hi, unsigned short& lo) {
  hi = to16<T16>(v);
  lo = to16<T16>(v - from16<T16>(hi));
}
// one wave's contribution to the running max |logit| (amax >= 0 in every lane). `dst` is VOG_LOGIT_WORDS words, 128 bytes apart
// (one L2 line each), that only ever RISE: a workgroup reads its word when it starts (logit_prev: the load is in flight behind
// the whole kernel) and a wave issues a no-return atomic max only if it would raise it - in steady state no atomic at all.
// Measured forms (cfg 2, 4 forwards in flight): every wave an atomicMax on one word cost mul_tx's attention 5 of its 12 us (480
// serialised read-modify-writes on one L2 line); 8 adjacent words zeroed per forward with the read in front of the atomic, i.e.
// in the middle of the kernel: 52.4 k queries/s against 57.2 k without any report (scratch/r6_i.sh); this form: see DESIGN.md.
#ifndef VOG_LOGIT_SAMPLE
#define VOG_LOGIT_SAMPLE 1        // 0: nobody reports (perf experiments)
#endif
static constexpr int kLogitWords = 32, kLogitStride = 32;     // = VOG_LOGIT_WORDS / VOG_LOGIT_STRIDE of vog_hip.h
__device__ __forceinline__ unsigned int logit_prev(const unsigned int* dst) {
  if (!dst || VOG_LOGIT_SAMPLE == 0) return 0xffffffffu;
  return __hip_atomic_load(dst + (blockIdx.x & (kLogitWords - 1)) * kLogitStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_logit_max(unsigned int* dst, unsigned int prev, float amax, int lane) {
  if (!dst || VOG_LOGIT_SAMPLE == 0) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  const unsigned int bits = __float_as_uint(amax);
  if (lane == 0 && bits > prev) atomicMax(dst + (blockIdx.x & (kLogitWords - 1)) * kLogitStride, bits);
}
// output rows: 4 consecutive head columns as 16-bit (+ their 16-bit remainder when the consumer is a split tail)
template <typename T16>
__device__ __forceinline__ void store_out4(unsigned short* orow, unsigned short* orow_lo, float a, float b, float c, float d) {
  u16x4 v = {to16<T16>(a), to16<T16>(b), to16<T16>(c), to16<T16>(d)};
  *reinterpret_cast<u16x4*>(orow) = v;
  if (orow_lo) {
    u16x4 l = {to16<T16>(a - from16<T16>(v[0])), to16<T16>(b - from16<T16>(v[1])), to16<T16>(c - from16<T16>(v[2])),
               to16<T16>(d - from16<T16>(v[3]))};
    *reinterpret_cast<u16x4*>(orow_lo) = l;
  }
}

template <typename T16, int NDB>
struct AttnFragBody {
  using Params = AttnParams;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const AttnParams& p, const BlockCtx& cx, unsigned char* smem_raw) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  constexpr int OSLOT = NDB * 16 * 64;               // floats of one wave's O^T partial
  float* smem = reinterpret_cast<float*>(smem_raw);
  float* obuf = smem;                                // [2][OSLOT]
  float* mlbuf = smem + 2 * OSLOT;                   // [4][2][64]  (m, l) per wave
  float* us = mlbuf + 4 * 2 * 64;                    // [npad] bias precursor of every key
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  // 1-D grid. Workgroup b runs on XCD b % 8 (observed; speed only): give the 8 XCDs 8
  // different (sequence, head) pairs and keep all query blocks of a pair on ONE XCD, so
  // its K/V fragments are fetched into one L2 instead of up to eight.
  const int nqb = (p.N + 31) >> 5;
  const int npair = p.S * p.H;
  int pair, qb;
  {
    const int b = cx.bx;
    const int full = (npair / 8) * 8;                 // pairs that form complete groups of 8
    const int grp = b / (8 * nqb);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); qb = (b >> 3) % nqb; }
    else { const int r = b - full * nqb; pair = full + r / nqb; qb = r % nqb; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int qi = qb * 32 + ql;
  const bool q_ok = qi < p.N;
  const int nkb = nqb;
  const int64_t base = ((int64_t)s * p.H + h) * (int64_t)p.npad * DP;
  const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + base) + (int64_t)qb * KS * 64 + lane;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.k + base) + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vt + base) + lane;

  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.n_box;
    peb = p.pe_b[h];
    stage_batched<4, 256, float>(p.npad, tid,
        [&](int key) { return p.u[(u_base + ((key < p.N ? key : 0) % p.n_box)) * p.H + h]; },
        [&](int key, float v) { us[key] = key < p.N ? v : 0.f; });
    if (q_ok) uq = p.u[(u_base + (qi % p.n_box)) * p.H + h];
  }
  __syncthreads();

  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // ---- the 4 waves of the workgroup split the KEY blocks of this query block (wave w
  // takes kb = w, w+4, ...): at N = 100 every wave has exactly one block, so the whole
  // attention is one round of loads + 32 MFMAs per wave, then a merge. No wave waits on
  // another until the merge. (Variants measured and rejected on MI355X: one wave per
  // query block walking all key blocks 23.8 us; + software-prefetched next K block 30 us
  // — the extra 64 registers spill; K/V register-resident across 2 query blocks with Q
  // shared through LDS 25-31 us — spills again. This form: 21.9 us mul, 10.5 us obj.)
  for (int kb = wid; kb < nkb; kb += 4) {
    u16x8 kf[KS], qf[KS], vf[NDB * 2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[ks] = Kf[((int64_t)kb * KS + ks) * 64];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = Qf[ks * 64];
#pragma unroll
    for (int i = 0; i < NDB * 2; ++i) vf[i] = Vf[((int64_t)kb * NDB * 2 + i) * 64];
    // ---- S^T block [32 keys x 32 queries]; two chains halve the dependent-MFMA latency
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      s0 = mfma32<T16>(kf[ks], qf[ks], s0);
      if (ks + 1 < KS) s1 = mfma32<T16>(kf[ks + 1], qf[ks + 1], s1);
    }
    // ---- bias, scale, mask, block max
    f32x16 sacc;
    float mloc = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + c32_row(r, lane);
      float x = s0[r] + s1[r];
      if (p.use_rel) x += fmaxf(uq - us[key] + peb, 0.f);
      x *= p.inv_scale;
      x = key < p.N ? x : -1e30f;
      sacc[r] = x;
      mloc = fmaxf(mloc, x);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __expf(m_run - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(sacc[r] - m_new);
      sacc[r] = e;
      lsum += e;
    }
    lsum += __shfl_xor(lsum, 32);
    l_run = l_run * alpha + lsum;
    m_run = m_new;
    if (kb >= 4 && !__all(alpha == 1.0f)) {
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    // ---- P^T fragments straight from the accumulator registers
    u16x8 pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[ks][j] = to16<T16>(sacc[ks * 8 + j]);
    // pad keys of the last block carry p = 0 but their V fragment entries are
    // whatever the (zero-initialised, never written) buffer holds: finite by contract
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) o[db] = mfma32<T16>(vf[db * 2 + ks], pf[ks], o[db]);
  }

  // ---- merge the 4 partial (m, l, O^T) with a two-level tree through LDS
  auto publish = [&](int slot) {
    float* ob = obuf + slot * OSLOT;
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) ob[(i * 16 + r) * 64 + lane] = o[i][r];
    mlbuf[(wid * 2 + 0) * 64 + lane] = m_run;
    mlbuf[(wid * 2 + 1) * 64 + lane] = l_run;
  };
  auto absorb = [&](int slot, int other) {
    const float* ob = obuf + slot * OSLOT;
    const float mb = mlbuf[(other * 2 + 0) * 64 + lane], lb = mlbuf[(other * 2 + 1) * 64 + lane];
    const float m = fmaxf(m_run, mb);
    const float fa = __expf(m_run - m), fb = __expf(mb - m);
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = o[i][r] * fa + ob[(i * 16 + r) * 64 + lane] * fb;
    l_run = l_run * fa + lb * fb;
    m_run = m;
  };
  if (nkb > 2) {                      // waves 2,3 hold something only then
    if (wid >= 2) publish(wid - 2);
    __syncthreads();
    if (wid < 2) absorb(wid, wid + 2);
    __syncthreads();
  }
  if (nkb > 1) {
    if (wid == 1) publish(0);
    __syncthreads();
    if (wid == 0) absorb(0, 1);
  }
  // ---- normalise and store: O^T[d][q] -> out[(s*N+q), h*DP + d], 4 consecutive d per store
  if (wid == 0 && q_ok) {
    const float inv_l = 1.0f / l_run;
    unsigned short* orow = p.out + ((int64_t)s * p.N + qi) * ((int64_t)p.H * DP) + (int64_t)h * DP;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = to16<T16>(o[db][g * 4 + e] * inv_l);
        *reinterpret_cast<u16x4*>(orow + db * 32 + g * 8 + hi * 4) = v;
      }
  }
}
};

// ---- attn_frag under a quarter of a CU's registers (round 4; <= 8 key blocks, N <= 256: obj_tx at gt5). AttnFragBody keeps
// K, Q, V^T of a key block and all NDB output accumulators per wave (368 registers at dp = 192: ONE 256-thread workgroup per
// CU, 84 CUs for 12 us per forward at 5 % MFMA utilisation, and the 4-forward loop is bound by held CU time). Three phases,
// two barriers: (1) the 4 waves split the KEY blocks: S^T tiles stay in registers (contraction walked two k-steps at a time),
// block maxima to LDS; (2) softmax against the row maximum over ALL blocks (no running maximum, no accumulator rescale, no
// merge of partial outputs): P^T fragments (16 bit, MFMA B-operand order) and block sums to LDS; (3) the waves split the
// OUTPUT d-blocks: one accumulator per d-block, V^T fragments streamed from L2, P^T fragments from LDS. <= 128 registers:
// four workgroups per CU.
template <typename T16, int NDB, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? 2 : 4) void attn_frag_lean_kernel(AttnParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  const unsigned int lprev = logit_prev(p.logit_max);   // (in flight behind the kernel: publish_logit_max)
  constexpr int MAXKB = 8;
  static_assert(KS % 2 == 0, "two k-steps per round");
  extern __shared__ __attribute__((aligned(16))) unsigned char afl_smem[];
  u16x8* Pl = reinterpret_cast<u16x8*>(afl_smem);                               // [MAXKB * 2][64] fragments
  float* mloc = reinterpret_cast<float*>(afl_smem + MAXKB * 2 * 64 * 16);      // [MAXKB][32] block maxima
  float* lloc = mloc + MAXKB * 32;                                              // [MAXKB][32] block sums
  float* us = lloc + MAXKB * 32;                                                // [npad] bias precursor of every key
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int nqb = (p.N + 31) >> 5;
  const int npair = p.S * p.H;
  int pair, qb;
  {   // as AttnFragBody: all query blocks of a (sequence, head) on ONE XCD (block b runs on XCD b % 8; speed only)
    const int b = blockIdx.x;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * nqb);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); qb = (b >> 3) % nqb; }
    else { const int r = b - full * nqb; pair = full + r / nqb; qb = r % nqb; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int qi = qb * 32 + ql;
  const bool q_ok = qi < p.N;
  const int nkb = nqb;
  const int64_t base = ((int64_t)s * p.H + h) * (int64_t)p.npad * DP;
  const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + base) + (int64_t)qb * KS * 64 + lane;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.k + base) + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vt + base) + lane;
  // (SPLIT: the 16-bit remainders of Q and K, same fragment order)
  const u16x8* Qlf = SPLIT ? reinterpret_cast<const u16x8*>(p.q_lo + base) + (int64_t)qb * KS * 64 + lane : Qf;
  const u16x8* Klf = SPLIT ? reinterpret_cast<const u16x8*>(p.k_lo + base) + lane : Kf;
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.n_box;
    peb = p.pe_b[h];
    stage_batched<4, 256, float>(p.npad, tid,
        [&](int key) { return p.u[(u_base + ((key < p.N ? key : 0) % p.n_box)) * p.H + h]; },
        [&](int key, float v) { us[key] = key < p.N ? v : 0.f; });
    if (q_ok) uq = p.u[(u_base + (qi % p.n_box)) * p.H + h];
  }
  __syncthreads();
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;

  // ---- phase 1: S^T tiles of this wave's key blocks (kb = wid, wid + 4), scaled logits kept in registers
  f32x16 sacc[2];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kb = wid + 4 * i;
    if (kb < nkb) {
      const u16x8* Kb = Kf + (int64_t)kb * KS * 64;
      const u16x8* Klb = Klf + (int64_t)kb * KS * 64;
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      u16x8 nq0 = Qf[0], nq1 = Qf[64], nk0 = Kb[0], nk1 = Kb[64];
      u16x8 nql0 = nq0, nql1 = nq1, nkl0 = nk0, nkl1 = nk1;
      if constexpr (SPLIT) { nql0 = Qlf[0]; nql1 = Qlf[64]; nkl0 = Klb[0]; nkl1 = Klb[64]; }
#pragma unroll 1
      for (int ks = 0; ks < KS; ks += 2) {
        const u16x8 q0 = nq0, q1 = nq1, k0 = nk0, k1 = nk1;
        const u16x8 ql0 = nql0, ql1 = nql1, kl0 = nkl0, kl1 = nkl1;
        if (ks + 2 < KS) {
          nq0 = Qf[(ks + 2) * 64]; nq1 = Qf[(ks + 3) * 64]; nk0 = Kb[(ks + 2) * 64]; nk1 = Kb[(ks + 3) * 64];
          if constexpr (SPLIT) { nql0 = Qlf[(ks + 2) * 64]; nql1 = Qlf[(ks + 3) * 64]; nkl0 = Klb[(ks + 2) * 64]; nkl1 = Klb[(ks + 3) * 64]; }
        }
        s0 = qk_mfma<T16, SPLIT>(k0, q0, kl0, ql0, s0);
        s1 = qk_mfma<T16, SPLIT>(k1, q1, kl1, ql1, s1);
      }
      float mblk = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + c32_row(r, lane);
        float x = s0[r] + s1[r];
        if (p.use_rel) x += fmaxf(uqp - us[key], 0.f);
        x = key < p.N ? x * c2 : -1e30f;
        sacc[i][r] = x;
        mblk = fmaxf(mblk, x);
        amax = fmaxf(amax, key < p.N ? fabsf(x) : 0.f);
      }
      mblk = fmaxf(mblk, __shfl_xor(mblk, 32));
      if (hi == 0) mloc[kb * 32 + ql] = mblk;
    }
  }
  publish_logit_max(p.logit_max, lprev, q_ok ? amax * 0.69314718056f : 0.f, lane);      // (log2 units -> nats)
  __syncthreads();
  // ---- phase 2: probabilities against the row maximum over all key blocks; P^T fragments straight from the registers
  float m = -1e30f;
  for (int kb = 0; kb < nkb; ++kb) m = fmaxf(m, mloc[kb * 32 + ql]);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kb = wid + 4 * i;
    if (kb < nkb) {
      float lsum = 0.f;
      u16x8 pf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sacc[i][r] - m);
        lsum += e;
        pf[r >> 3][r & 7] = to16<T16>(e);
      }
      lsum += __shfl_xor(lsum, 32);
      if (hi == 0) lloc[kb * 32 + ql] = lsum;
      Pl[(kb * 2 + 0) * 64 + lane] = pf[0];
      Pl[(kb * 2 + 1) * 64 + lane] = pf[1];
    }
  }
  __syncthreads();
  // ---- phase 3: O^T d-blocks (db = wid, wid + 4, ...): V^T fragments from L2, P^T from LDS
  float l = 0.f;
  for (int kb = 0; kb < nkb; ++kb) l += lloc[kb * 32 + ql];
  const float inv_l = 1.0f / l;
  for (int db = wid; db < NDB; db += 4) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    u16x8 v0 = Vf[(int64_t)(db * 2) * 64], v1 = Vf[(int64_t)(db * 2 + 1) * 64];
    for (int kb = 0; kb < nkb; ++kb) {
      const u16x8 a0 = v0, a1 = v1;
      if (kb + 1 < nkb) {
        v0 = Vf[((int64_t)(kb + 1) * NDB * 2 + db * 2) * 64];
        v1 = Vf[((int64_t)(kb + 1) * NDB * 2 + db * 2 + 1) * 64];
      }
      o = mfma32<T16>(a0, Pl[(kb * 2 + 0) * 64 + lane], o);
      o = mfma32<T16>(a1, Pl[(kb * 2 + 1) * 64 + lane], o);
    }
    if (q_ok) {
      const int64_t oo = ((int64_t)s * p.N + qi) * ((int64_t)p.H * DP) + (int64_t)h * DP + db * 32 + hi * 4;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        store_out4<T16>(p.out + oo + g * 8, p.out_lo ? p.out_lo + oo + g * 8 : nullptr, o[g * 4] * inv_l, o[g * 4 + 1] * inv_l,
                        o[g * 4 + 2] * inv_l, o[g * 4 + 3] * inv_l);
    }
  }
}

// ---- attn_frag with every operand requested in ONE round trip (round 6; <= 8 key blocks: obj_tx at gt5). The lean form above walks
// Q.K^T in KS / 2 rounds and P.V in one round per key block, each round one L2 latency behind the previous one (fragments of the
// next round only): 84 workgroups at cfg 2, one per CU, nothing to hide ~13 dependent round trips behind - 10.5 us for 0.25 us of
// MFMA work. Here 8 waves: wave w owns KEY block w in phases 1-2 and OUTPUT d-block w in phase 3, so a wave's whole K block (KS
// fragments) and its whole V^T column (2 fragments per key block) are register-resident and requested at kernel start together
// with the query block (shared: staged in LDS once per workgroup) and the bias precursors; the V^T fragments land while Q.K^T and
// the softmax run. The output rows leave through an LDS tile (the query image, free after phase 1): a store instruction writes
// whole 16-byte pieces of contiguous head rows instead of scattering 8-byte pieces over 32 rows.
// Same arithmetic as the lean form except the P.V sum, which runs as two chains (even / odd fragments) added at the end.
template <typename T16, int NDB>
__global__ __launch_bounds__(512, 2) void attn_frag8_kernel(AttnParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16, MAXKB = 8, QF = (KS + 7) / 8;
  static_assert(KS % 2 == 0, "two accumulation chains");
  const unsigned int lprev = logit_prev(p.logit_max);   // (in flight behind the kernel: publish_logit_max)
  extern __shared__ __attribute__((aligned(16))) unsigned char af8_smem[];
  u16x8* Qimg = reinterpret_cast<u16x8*>(af8_smem);                             // [KS][64] fragments; later the output tile
  u16x8* Pl = Qimg + KS * 64;                                                   // [MAXKB * 2][64] fragments
  float* mloc = reinterpret_cast<float*>(Pl + MAXKB * 2 * 64);                  // [MAXKB][32] block maxima
  float* lloc = mloc + MAXKB * 32;                                              // [MAXKB][32] block sums
  float* us = lloc + MAXKB * 32;                                                // [npad] bias precursor of every key
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int nqb = (p.N + 31) >> 5;
  const int npair = p.S * p.H;
  int pair, qb;
  {   // as the lean form: all query blocks of a (sequence, head) on ONE XCD (block b runs on XCD b % 8; speed only)
    const int b = blockIdx.x;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * nqb);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); qb = (b >> 3) % nqb; }
    else { const int r = b - full * nqb; pair = full + r / nqb; qb = r % nqb; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int qi = qb * 32 + ql;
  const bool q_ok = qi < p.N;
  const int nkb = nqb;
  const bool kb_ok = wid < nkb, db_ok = wid < NDB;
  const int kb = kb_ok ? wid : 0, db = db_ok ? wid : 0;
  const int64_t base = ((int64_t)s * p.H + h) * (int64_t)p.npad * DP;
  const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + base) + (int64_t)qb * KS * 64 + lane;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.k + base) + (int64_t)kb * KS * 64 + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vt + base) + (int64_t)(db * 2) * 64 + lane;
  // ---- every request of the kernel (no load below this block)
  u16x8 qf[QF];
#pragma unroll
  for (int i = 0; i < QF; ++i) {
    const int f = wid + 8 * i;
    qf[i] = Qf[(f < KS ? f : KS - 1) * 64];
  }
  float uk = 0.f, uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.n_box;
    peb = p.pe_b[h];
    uk = p.u[(u_base + ((tid < p.N ? tid : 0) % p.n_box)) * p.H + h];
    uq = p.u[(u_base + ((q_ok ? qi : 0) % p.n_box)) * p.H + h];
  }
  u16x8 kf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kf[ks] = Kf[ks * 64];
  u16x8 vf[MAXKB * 2];
#pragma unroll
  for (int j = 0; j < MAXKB; ++j) {
    const int kbc = j < nkb ? j : nkb - 1;           // (blocks past the end repeat the last one: no conditional load)
    vf[2 * j] = Vf[(int64_t)kbc * NDB * 2 * 64];
    vf[2 * j + 1] = Vf[((int64_t)kbc * NDB * 2 + 1) * 64];
  }
#pragma unroll
  for (int i = 0; i < QF; ++i) {
    const int f = wid + 8 * i;
    if (f < KS) Qimg[f * 64 + lane] = qf[i];
  }
  if (tid < p.npad) us[tid] = (p.use_rel && tid < p.N) ? uk : 0.f;
  __syncthreads();
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;

  // ---- phase 1: the S^T tile of this wave's key block, scaled logits kept in registers
  f32x16 sacc;
  float amax = 0.f;
  if (kb_ok) {
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    float4 ub[4];                                      // bias precursors of this lane's 16 keys (rows 8g + 4hi + 0..3): read ahead of
#pragma unroll                                         // the MFMAs, not one by one behind a branch after them
    for (int g = 0; g < 4; ++g) ub[g] = *reinterpret_cast<const float4*>(us + kb * 32 + g * 8 + hi * 4);
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      const u16x8 q0 = Qimg[ks * 64 + lane], q1 = Qimg[(ks + 1) * 64 + lane];
      s0 = mfma32<T16>(kf[ks], q0, s0);
      s1 = mfma32<T16>(kf[ks + 1], q1, s1);
    }
    float mblk = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + c32_row(r, lane);
      float x = s0[r] + s1[r];
      const float ukey = (r & 3) == 0 ? ub[r >> 2].x : ((r & 3) == 1 ? ub[r >> 2].y : ((r & 3) == 2 ? ub[r >> 2].z : ub[r >> 2].w));
      if (p.use_rel) x += fmaxf(uqp - ukey, 0.f);
      x = key < p.N ? x * c2 : -1e30f;
      sacc[r] = x;
      mblk = fmaxf(mblk, x);
      amax = fmaxf(amax, key < p.N ? fabsf(x) : 0.f);
    }
    mblk = fmaxf(mblk, __shfl_xor(mblk, 32));
    if (hi == 0) mloc[kb * 32 + ql] = mblk;
  }
  publish_logit_max(p.logit_max, lprev, (kb_ok && q_ok) ? amax * 0.69314718056f : 0.f, lane);      // (log2 units -> nats)
  __syncthreads();
  // ---- phase 2: probabilities against the row maximum over all key blocks; P^T fragments straight from the registers
  if (kb_ok) {
    float m = -1e30f;
    for (int j = 0; j < nkb; ++j) m = fmaxf(m, mloc[j * 32 + ql]);
    float lsum = 0.f;
    u16x8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sacc[r] - m);
      lsum += e;
      pf[r >> 3][r & 7] = to16<T16>(e);
    }
    lsum += __shfl_xor(lsum, 32);
    if (hi == 0) lloc[kb * 32 + ql] = lsum;
    Pl[(kb * 2 + 0) * 64 + lane] = pf[0];
    Pl[(kb * 2 + 1) * 64 + lane] = pf[1];
  }
  __syncthreads();
  // ---- phase 3: the O^T d-block of this wave, V^T fragments from registers, P^T from LDS; rows parked in the output tile
  unsigned char* tile = af8_smem;                    // [32 rows][NDB * 64 B], 16-byte chunks XOR-swizzled by the row (low 3 bits)
  constexpr int SWZ = NDB * 4 >= 8 ? 7 : NDB * 4 - 1;
  if (db_ok) {
    float l = 0.f;
    for (int j = 0; j < nkb; ++j) l += lloc[j * 32 + ql];
    const float inv_l = 1.0f / l;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < MAXKB; ++j)
      if (j < nkb) {
        o0 = mfma32<T16>(vf[2 * j], Pl[(j * 2 + 0) * 64 + lane], o0);
        o1 = mfma32<T16>(vf[2 * j + 1], Pl[(j * 2 + 1) * 64 + lane], o1);
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const u16x4 v = {to16<T16>((o0[g * 4] + o1[g * 4]) * inv_l), to16<T16>((o0[g * 4 + 1] + o1[g * 4 + 1]) * inv_l),
                       to16<T16>((o0[g * 4 + 2] + o1[g * 4 + 2]) * inv_l), to16<T16>((o0[g * 4 + 3] + o1[g * 4 + 3]) * inv_l)};
      *reinterpret_cast<u16x4*>(tile + ql * (NDB * 64) + (((db * 4 + g) ^ (ql & SWZ)) << 4) + hi * 8) = v;
    }
  }
  __syncthreads();
  {
    const int64_t ldo = (int64_t)p.H * DP;
    unsigned short* o0p = p.out + ((int64_t)s * p.N + qb * 32) * ldo + (int64_t)h * DP;
#pragma unroll
    for (int it = 0; it < (32 * NDB * 4 + 511) / 512; ++it) {
      const int c = tid + it * 512;
      const int row = c / (NDB * 4), ch = c - row * (NDB * 4);
      if (c < 32 * NDB * 4 && qb * 32 + row < p.N)
        *reinterpret_cast<u16x8*>(o0p + (int64_t)row * ldo + ch * 8) =
            *reinterpret_cast<const u16x8*>(tile + row * (NDB * 64) + ((ch ^ (row & SWZ)) << 4));
    }
  }
}

template <typename T16, int NDB>
__global__ __launch_bounds__(256) void attn_frag_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char af_smem[];
  AttnFragBody<T16, NDB>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, af_smem);
}

// ----------------------------------------------------------------------------
// Single-pass variant for N <= 128 (<= 4 key blocks: every mul_tx shape of gt5, where this
// kernel is the largest single item of the forward). Wave w owns key block w, so nothing
// has to persist across key blocks, and the kernel is shaped around OCCUPANCY: the general
// kernel needs 462 registers per wave = one workgroup per CU, i.e. 480 workgroups run as
// two serial rounds of ~8.5 us that are each >50 % load wait (measured: SQ_WAIT_ANY 53 %).
// Here the softmax is made global BEFORE the PV product (the 4 waves exchange (max, sum)
// through LDS and fold exp(m_w - m*) into P), so the partial O^T of the waves are plain
// summands; PV runs in two head-dim halves that reuse the same 32 + 64 registers, and the
// 4 waves reduce the halves in parallel (wave w sums and stores d-block w). <= 256
// registers => two workgroups per CU overlap each other's load round trip.
// ----------------------------------------------------------------------------
template <typename T16, int NDB, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void attn_sb_kernel(AttnParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  const unsigned int lprev = logit_prev(p.logit_max);   // (in flight behind the kernel: publish_logit_max)
  constexpr int HB = (NDB + 1) / 2;                  // d-blocks per half
  constexpr int SLOT = HB * 16 * 64;                 // floats of one wave's half partial
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* slots = smem;                               // [4][SLOT]
  float* mlbuf = smem + 4 * SLOT;                    // [4][2][64]
  float* us = mlbuf + 4 * 2 * 64;                    // [npad]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int nqb = (p.N + 31) >> 5;                   // = number of key blocks <= 4
  const int npair = p.S * p.H;
  int pair, qb;
  {
    const int b = blockIdx.x;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * nqb);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); qb = (b >> 3) % nqb; }
    else { const int r = b - full * nqb; pair = full + r / nqb; qb = r % nqb; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int qi = qb * 32 + ql;
  const bool q_ok = qi < p.N;
  const bool active = wid < nqb;                     // wave-uniform: has a key block
  const int64_t base = ((int64_t)s * p.H + h) * (int64_t)p.npad * DP;
  const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + base) + (int64_t)qb * KS * 64 + lane;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.k + base) + (int64_t)wid * KS * 64 + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vt + base) + (int64_t)wid * NDB * 2 * 64 + lane;
  // (SPLIT: the 16-bit remainders of Q and K; streamed through the contraction loop, two k-steps at a time)
  const u16x8* Qlf = SPLIT ? reinterpret_cast<const u16x8*>(p.q_lo + base) + (int64_t)qb * KS * 64 + lane : Qf;
  const u16x8* Klf = SPLIT ? reinterpret_cast<const u16x8*>(p.k_lo + base) + (int64_t)wid * KS * 64 + lane : Kf;

  // requests first: K and Q (needed at once), then the first V half
  u16x8 kf[KS], qf[KS], vf[HB * 2];
  if (active) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kf[ks] = Kf[ks * 64]; qf[ks] = Qf[ks * 64]; }
#pragma unroll
    for (int i = 0; i < HB * 2; ++i) vf[i] = Vf[i * 64];
  }
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.n_box;
    peb = p.pe_b[h];
    stage_batched<4, 256, float>(p.npad, tid,
        [&](int key) { return p.u[(u_base + ((key < p.N ? key : 0) % p.n_box)) * p.H + h]; },
        [&](int key, float v) { us[key] = key < p.N ? v : 0.f; });
    if (q_ok) uq = p.u[(u_base + (qi % p.n_box)) * p.H + h];
  }
  __syncthreads();

  float m_w = -1e30f, l_w = 0.f;
  f32x16 sacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
  if (active) {
    f32x16 s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      if constexpr (SPLIT) {
        const u16x8 kl0 = Klf[ks * 64], ql0 = Qlf[ks * 64];
        sacc = qk_mfma<T16, true>(kf[ks], qf[ks], kl0, ql0, sacc);
        if (ks + 1 < KS) {
          const u16x8 kl1 = Klf[(ks + 1) * 64], ql1 = Qlf[(ks + 1) * 64];
          s1 = qk_mfma<T16, true>(kf[ks + 1], qf[ks + 1], kl1, ql1, s1);
        }
      } else {
        sacc = mfma32<T16>(kf[ks], qf[ks], sacc);
        if (ks + 1 < KS) s1 = mfma32<T16>(kf[ks + 1], qf[ks + 1], s1);
      }
    }
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = wid * 32 + c32_row(r, lane);
      float x = sacc[r] + s1[r];
      if (p.use_rel) x += fmaxf(uq - us[key] + peb, 0.f);
      x *= p.inv_scale;
      amax = fmaxf(amax, (key < p.N && q_ok) ? fabsf(x) : 0.f);
      x = key < p.N ? x : -1e30f;
      sacc[r] = x;
      m_w = fmaxf(m_w, x);
    }
    publish_logit_max(p.logit_max, lprev, amax, lane);
    m_w = fmaxf(m_w, __shfl_xor(m_w, 32));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(sacc[r] - m_w);
      sacc[r] = e;
      l_w += e;
    }
    l_w += __shfl_xor(l_w, 32);
  }
  mlbuf[(wid * 2 + 0) * 64 + lane] = m_w;
  mlbuf[(wid * 2 + 1) * 64 + lane] = l_w;
  __syncthreads();
  // global softmax statistics of this query (same in every wave)
  float m_all = -1e30f;
#pragma unroll
  for (int w = 0; w < 4; ++w) m_all = fmaxf(m_all, mlbuf[(w * 2) * 64 + lane]);
  float l_all = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) l_all += mlbuf[(w * 2 + 1) * 64 + lane] * __expf(mlbuf[(w * 2) * 64 + lane] - m_all);
  const float f_w = __expf(m_w - m_all);
  const float inv_l = 1.0f / l_all;
  // P^T fragments, already on the global scale
  u16x8 pf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[ks][j] = to16<T16>(sacc[ks * 8 + j] * f_w);

#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int nb = half == 0 ? HB : NDB - HB;        // d-blocks in this half
    if (nb <= 0) break;
    if (half == 1) {
      __syncthreads();                               // slots consumed by the previous half
      if (active) {
#pragma unroll
        for (int i = 0; i < (NDB - HB) * 2; ++i) vf[i] = Vf[(HB * 2 + i) * 64];
      }
    }
    float* mine = slots + wid * SLOT;
    if (active) {
#pragma unroll
      for (int db = 0; db < HB; ++db) {
        if (db < nb) {
          f32x16 o;
#pragma unroll
          for (int r = 0; r < 16; ++r) o[r] = 0.f;
          o = mfma32<T16>(vf[db * 2], pf[0], o);
          o = mfma32<T16>(vf[db * 2 + 1], pf[1], o);
#pragma unroll
          for (int r = 0; r < 16; ++r) mine[(db * 16 + r) * 64 + lane] = o[r];
        }
      }
    }
    __syncthreads();
    // wave w reduces and stores d-block w (w + 4, ...) of this half
    for (int db = wid; db < nb; db += 4) {
      float acc[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int w = 0; w < nqb; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += slots[w * SLOT + (db * 16 + r) * 64 + lane];
      if (q_ok) {
        const int64_t oo = ((int64_t)s * p.N + qi) * ((int64_t)p.H * DP) + (int64_t)h * DP + (half * HB + db) * 32 + hi * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          store_out4<T16>(p.out + oo + g * 8, p.out_lo ? p.out_lo + oo + g * 8 : nullptr, acc[g * 4] * inv_l,
                          acc[g * 4 + 1] * inv_l, acc[g * 4 + 2] * inv_l, acc[g * 4 + 3] * inv_l);
      }
    }
  }
}

// ----------------------------------------------------------------------------
// Shared-tile variant for long sequences (p100: N = 2000 / 4000). The kernels above give every
// 32-query block its own pass over K and V (fine up to a few hundred tokens: everything is
// L2-resident and parallelism matters more); at N = 2000 that is 1 MiB of K/V per 32 queries and the
// kernel becomes L2-bandwidth bound (measured: 345 TFLOP/s = 14 % of the MFMA peak). Here a workgroup
// owns 128 queries (one 32-query block per wave, Q fragments in registers for the whole pass) and the
// 4 waves walk the key blocks TOGETHER: each 32-key block of K and V^T fragments is brought into LDS
// once per workgroup by LDS-DMA (the fragment order is lane-linear, i.e. exactly what
// global_load_lds writes), double buffered, one barrier per block; every wave reads its MFMA A
// operands from LDS (conflict-free 16-byte lane-linear reads). No merge at the end: a wave owns
// its queries' whole softmax row. K/V traffic per query drops 4x, Q is read once.
// ----------------------------------------------------------------------------
template <typename T16, int NDB>
__global__ __launch_bounds__(256) void attn_tile_kernel(AttnParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  constexpr int NF = KS + 2 * NDB;                   // KiB fragments per key block (K then V^T)
  extern __shared__ __attribute__((aligned(1024))) unsigned char tsm[];
  unsigned char* kv = tsm;                           // [2][NF][1024]
  float* us = reinterpret_cast<float*>(tsm + 2 * NF * 1024);   // [npad] bias precursor of every key
  if (p.guard && *reinterpret_cast<volatile const int*>(p.guard) == 0) return;   // fallback pass of attn_tile2_kernel: not needed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int nkb = (p.N + 31) >> 5;
  const int nqg = (p.N + 127) >> 7;                  // 128-query groups per (sequence, head)
  const int npair = p.S * p.H;
  int pair, qg;
  {   // XCD-aware: the query groups of one (sequence, head) stay on one XCD (its K/V in one L2)
    const int b = blockIdx.x;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * nqg);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); qg = (b >> 3) % nqg; }
    else { const int r = b - full * nqg; pair = full + r / nqg; qg = r % nqg; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int qb = qg * 4 + wid;                       // this wave's 32-query block
  const bool wave_ok = qb < nkb;                     // a wave past the end still helps with the DMA
  const int qi = qb * 32 + ql;
  const bool q_ok = qi < p.N;
  const int64_t base = ((int64_t)s * p.H + h) * (int64_t)p.npad * DP;
  const unsigned short* Kg = p.k + base;
  const unsigned short* Vg = p.vt + base;

  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.n_box;
    peb = p.pe_b[h];
    stage_batched<4, 256, float>(p.npad, tid,
        [&](int key) { return p.u[(u_base + ((key < p.N ? key : 0) % p.n_box)) * p.H + h]; },
        [&](int key, float v) { us[key] = key < p.N ? v : 0.f; });
    if (q_ok) uq = p.u[(u_base + (qi % p.n_box)) * p.H + h];
  }
  // Q fragments of this wave's block: registers for the whole pass
  u16x8 qf[KS];
  {
    const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + base) + (int64_t)(wave_ok ? qb : 0) * KS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = Qf[ks * 64];
  }
  // fragment f of key block kb: f < KS -> K fragment, else V^T fragment f - KS
  auto issue = [&](int kb, int buf) {
    for (int f = wid; f < NF; f += 4) {
      const unsigned short* src = f < KS ? Kg + ((int64_t)kb * KS + f) * 512
                                         : Vg + ((int64_t)kb * NDB * 2 + (f - KS)) * 512;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + lane * 8),
          (__attribute__((address_space(3))) void*)(kv + (buf * NF + f) * 1024), 16, 0, 0);
    }
  };

  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;
  // (a 3-buffer software pipeline that issues S^T of block kb+1 before the softmax of block kb was
  // measured equal: mul 1151 vs 1166 us, obj 497 vs 474 us at p100 - kept simple)

  issue(0, 0);
  for (int kb = 0; kb < nkb; ++kb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's share of block kb has landed
    __syncthreads();                                      // ... everybody's; and block kb-1 is consumed
    if (kb + 1 < nkb) issue(kb + 1, (kb + 1) & 1);
    if (!wave_ok) continue;
    const unsigned char* blk = kv + ((kb & 1) * NF) * 1024 + lane * 16;
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      s0 = mfma32<T16>(*reinterpret_cast<const u16x8*>(blk + ks * 1024), qf[ks], s0);
      if (ks + 1 < KS) s1 = mfma32<T16>(*reinterpret_cast<const u16x8*>(blk + (ks + 1) * 1024), qf[ks + 1], s1);
    }
    // softmax in the log2 domain: x2 = (s + bias) * (inv_scale * log2 e), p = 2^(x2 - m2). The row
    // of register r is (r&3) + 8*(r>>2) + 4*hi: the 4 bias precursors of a register quad are one
    // 16-byte LDS read; keys >= N exist only in the last block (uniform branch).
    f32x16 sacc;
    float mloc = -1e30f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 ub = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.use_rel) ub = *reinterpret_cast<const float4*>(&us[kb * 32 + 8 * g + 4 * hi]);
      const float ubv[4] = {ub.x, ub.y, ub.z, ub.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = g * 4 + e;
        float x = s0[r] + s1[r];
        if (p.use_rel) x += fmaxf(uqp - ubv[e], 0.f);
        sacc[r] = x * c2;
      }
    }
    if (kb == nkb - 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + c32_row(r, lane) >= p.N) sacc[r] = -1e30f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, sacc[r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sacc[r] - m_new);
      sacc[r] = e;
      lsum += e;
    }
    lsum += __shfl_xor(lsum, 32);
    l_run = l_run * alpha + lsum;
    m_run = m_new;
    if (kb > 0 && !__all(alpha == 1.0f)) {
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    u16x8 pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[ks][j] = to16<T16>(sacc[ks * 8 + j]);
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        o[db] = mfma32<T16>(*reinterpret_cast<const u16x8*>(blk + (KS + db * 2 + ks) * 1024), pf[ks], o[db]);
  }
  if (wave_ok && q_ok) {
    const float inv_l = 1.0f / l_run;
    unsigned short* orow = p.out + ((int64_t)s * p.N + qi) * ((int64_t)p.H * DP) + (int64_t)h * DP;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = to16<T16>(o[db][g * 4 + e] * inv_l);
        *reinterpret_cast<u16x4*>(orow + db * 32 + g * 8 + hi * 4) = v;
      }
  }
}

// ----------------------------------------------------------------------------
// Separable attention of mul_tx layer 0 (include/vog_hip.h, vog_attn_struct_args). Token (a, p) has
// k = Kv[p] + Kl[a], v = Vv[p] + Vl[a] and a bias that depends on (p, p') only, so the softmax over
// the nsrl*nppf keys is the product of a softmax over the nppf visual keys and one over the nsrl
// language keys, and   out = softmax_p'(q.Kv + bias).Vv + softmax_a'(q.Kl).Vl   (exact).
// One wave = one 32-query block, nothing shared between waves: the visual part is the flash loop of
// the kernels above over ceil(nppf/32) key blocks (ONE at gt5), the language part one masked block
// whose K / V fragments are assembled from the fp32 language projection (5 rows) in registers. The
// language probabilities are normalised BEFORE their P.V product so that it accumulates into the
// (already normalised) visual output registers.
// ----------------------------------------------------------------------------
struct AttnStructParams {
  const unsigned short* q; const unsigned short* kv; const unsigned short* vv; const float* pl;
  unsigned short* out; const float* u; const float* pe_b;
  int S, H, dp, nsrl, nppf, npad_q, npad_kv, nfrm, lpv, ncv, use_rel, seq_per_vid, NP;
  float inv_scale; int q_visual;
  int dbg;          // perf experiments only (VOG_ATTN_STRUCT_DEBUG; wrong results): 1 no output stores, 2 no language block,
                    // 4 no visual key blocks
  // round 5: attn_struct_ef_kernel sets *guard = 1 when a row of its shift mA[p] + mB[a] may sit too far above the row's true
  // maximum for the 16-bit E / P fragments (attn_struct_ef_dev.h); attn_struct_lds_kernel with guard_gate = 1 runs only then
  int* guard; int guard_gate;
  // round 6 (SPLIT kernel; q_visual form): 16-bit remainders of the visual query / key parts, same fragment order as q / kv;
  // the language parts come from the fp32 `pl` and are split in the kernel. out_lo / logit_max: as in AttnParams.
  const unsigned short* q_lo; const unsigned short* kv_lo; unsigned short* out_lo; unsigned int* logit_max;
};

#ifdef VOG_TS_ATTN   // scratch/ts_attn.hip: wall-clock stamps (100 MHz) per wave
__device__ unsigned long long g_ats[4096][8];
#define VOG_ATS(slot) do { if (lane == 0) g_ats[(blockIdx.x * 4 + wid) & 4095][slot] = wall_clock64(); } while (0)
#else
#define VOG_ATS(slot) do { } while (0)
#endif

// shared pieces of the two struct kernels ----------------------------------------------------------
template <typename T16, int KS>
__device__ __forceinline__ void struct_load_q(const AttnStructParams& p, u16x8 (&qf)[KS], int s, int h, int qbs,
                                              int lane, const float* plr, int ldp, int64_t kvbase) {
  constexpr int DP = KS * 16;
  const int hi = lane >> 5, ql = lane & 31;
  if (p.q_visual) {
    // query (a, p) = Qv[p] + Ql[a]: the visual part is this token's 16-byte chunk of the fragment-
    // ordered Qv (tokens of a block are consecutive p: mostly one contiguous run), the language part
    // 8 floats of the projection row of argument a (a handful of distinct rows per wave)
    const int t = qbs * 32 + ql;
    int a = t / p.nppf;
    const int pp = t - a * p.nppf;
    a = a < p.nsrl ? a : p.nsrl - 1;                  // tokens past the end are never stored
    const unsigned short* qv = p.q + kvbase;
    const float* qlr = plr + (int64_t)a * ldp + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u16x8 v = *reinterpret_cast<const u16x8*>(qv + frag_qk(pp, ks * 16 + hi * 8, DP));
      const float4 l0 = *reinterpret_cast<const float4*>(qlr + ks * 16);
      const float4 l1 = *reinterpret_cast<const float4*>(qlr + ks * 16 + 4);
      qf[ks] = u16x8{to16<T16>(from16<T16>(v[0]) + l0.x), to16<T16>(from16<T16>(v[1]) + l0.y),
                     to16<T16>(from16<T16>(v[2]) + l0.z), to16<T16>(from16<T16>(v[3]) + l0.w),
                     to16<T16>(from16<T16>(v[4]) + l1.x), to16<T16>(from16<T16>(v[5]) + l1.y),
                     to16<T16>(from16<T16>(v[6]) + l1.z), to16<T16>(from16<T16>(v[7]) + l1.w)};
    }
  } else {
    const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + ((int64_t)s * p.H + h) * (int64_t)p.npad_q * DP) +
                      (int64_t)qbs * KS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = Qf[ks * 64];
  }
}

// language K fragments: lane = key a (rows >= nsrl are zero), 8 consecutive head columns
template <typename T16, int KS>
__device__ __forceinline__ void struct_load_kl(const AttnStructParams& p, u16x8 (&klf)[KS], int lane,
                                               const float* plr, int hd, int ldp) {
  const int hi = lane >> 5, ql = lane & 31;
  const bool a_ok = ql < p.nsrl;
  const float* kr = plr + hd + (int64_t)ql * ldp + hi * 8;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
    if (a_ok) { x0 = *reinterpret_cast<const float4*>(kr + ks * 16); x1 = *reinterpret_cast<const float4*>(kr + ks * 16 + 4); }
    klf[ks] = u16x8{to16<T16>(x0.x), to16<T16>(x0.y), to16<T16>(x0.z), to16<T16>(x0.w),
                    to16<T16>(x1.x), to16<T16>(x1.y), to16<T16>(x1.z), to16<T16>(x1.w)};
  }
}

// language V fragment of d-block db, k-step ks: lane = (hi, head column), register j = key
// 16*ks + 8*(j>>2) + 4*hi + (j&3)
template <typename T16>
__device__ __forceinline__ u16x8 struct_load_vl(const AttnStructParams& p, int db, int ks, int lane,
                                                const float* plr, int hd, int ldp) {
  const int hi = lane >> 5, ql = lane & 31;
  const float* vr = plr + 2 * hd + db * 32 + ql;
  u16x8 vl;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int key = ks * 16 + 8 * (j >> 2) + 4 * hi + (j & 3);
    vl[j] = key < p.nsrl ? to16<T16>(vr[(int64_t)key * ldp]) : (unsigned short)0;
  }
  return vl;
}

template <typename T16>
__device__ __forceinline__ void struct_store(const AttnStructParams& p, const f32x16& o, int db, int64_t row,
                                             int h, int DP, int hi) {
  const int64_t oo = row * ((int64_t)p.H * DP) + (int64_t)h * DP + db * 32 + hi * 4;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    store_out4<T16>(p.out + oo + g * 8, p.out_lo ? p.out_lo + oo + g * 8 : nullptr, o[g * 4], o[g * 4 + 1], o[g * 4 + 2],
                    o[g * 4 + 3]);
}

// ---- ONE visual key block (nppf <= 32: every gt5 shape). Both softmaxes are complete before any P.V product, so the output is
// produced d-block by d-block with ONE accumulator. The contraction over the head dimension is walked two k-steps at a time (the
// next pair's global fragments in flight), <= 128 registers: four workgroups share a CU and their load latencies hide each other
// (round 4; the form that held Q, K and K_lang of the whole head dimension - 272 registers, one workgroup per CU, 120 CUs for
// 13 us at 6 % MFMA utilisation - was removed in round 6: scratch/negatives/r6_pruned/).
template <typename T16, int NDB, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? 2 : 4) void attn_struct1_lean_kernel(AttnStructParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  const unsigned int lprev = logit_prev(p.logit_max);   // (in flight behind the kernel: publish_logit_max)
  static_assert(KS % 2 == 0, "two k-steps per round");
  extern __shared__ __attribute__((aligned(16))) float ssm[];
  float* us = ssm;                                   // [32] bias precursor of the visual keys
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int Nq = p.nsrl * p.nppf;
  const int nqb = (Nq + 31) >> 5, nqg = (nqb + 3) >> 2;
  const int pair = blockIdx.x / nqg, qg = blockIdx.x - pair * nqg;
  const int s = pair / p.H, h = pair - s * p.H;
  const int qb = qg * 4 + wid;
  const bool wave_ok = qb < nqb;
  const int qi = qb * 32 + ql;
  const bool q_ok = wave_ok && qi < Nq;
  const int hd = p.H * DP, ldp = 3 * hd;
  const int vid = s / p.nfrm;
  const int lv = p.lpv ? vid : vid / p.ncv;
  const float* plr = p.pl + (int64_t)lv * p.nsrl * ldp + h * DP;     // + hd: K block, + 2*hd: V block
  const int64_t kvbase = ((int64_t)s * p.H + h) * (int64_t)p.npad_kv * DP;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.kv + kvbase) + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vv + kvbase) + lane;
  float* pls = ssm + 32;                             // [nsrl][3][DP]
  {
    const int per_row = 3 * DP / 4;                  // float4 per argument
    stage_batched<4, 256, float4>(p.nsrl * per_row, tid,
        [&](int i) {
          const int a = i / per_row, c = i - a * per_row;
          const int part = c / (DP / 4), dd4 = c - part * (DP / 4);
          return *reinterpret_cast<const float4*>(plr + (int64_t)a * ldp + part * hd + dd4 * 4);
        },
        [&](int i, const float4& v) {
          const int a = i / per_row, c = i - a * per_row;
          const int part = c / (DP / 4), dd4 = c - part * (DP / 4);
          *reinterpret_cast<float4*>(&pls[(a * 3 + part) * DP + dd4 * 4]) = v;
        });
  }
  const int qbs = wave_ok ? qb : 0;
  int qa = 0, qp = 0;
  const unsigned short* qv = p.q + kvbase;
  const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + ((int64_t)s * p.H + h) * (int64_t)p.npad_q * DP) + (int64_t)qbs * KS * 64 + lane;
  if (p.q_visual) {
    const int t = qbs * 32 + ql;
    qa = t / p.nppf;
    qp = t - qa * p.nppf;
    qa = qa < p.nsrl ? qa : p.nsrl - 1;              // tokens past the end are never stored
  }
  auto load_q = [&](int ks) -> u16x8 {
    return p.q_visual ? *reinterpret_cast<const u16x8*>(qv + frag_qk(qp, ks * 16 + hi * 8, DP)) : Qf[ks * 64];
  };
  // (SPLIT, q_visual form only: 16-bit remainders of Qv and Kv)
  const unsigned short* qvl = SPLIT ? p.q_lo + kvbase : qv;
  const u16x8* Klof = SPLIT ? reinterpret_cast<const u16x8*>(p.kv_lo + kvbase) + lane : Kf;
  auto load_ql = [&](int ks) -> u16x8 { return *reinterpret_cast<const u16x8*>(qvl + frag_qk(qp, ks * 16 + hi * 8, DP)); };
  u16x8 nq0 = load_q(0), nq1 = load_q(1), nk0 = Kf[0], nk1 = Kf[64];
  u16x8 nql0 = nq0, nql1 = nq1, nkl0 = nk0, nkl1 = nk1;
  if constexpr (SPLIT) { nql0 = load_ql(0); nql1 = load_ql(1); nkl0 = Klof[0]; nkl1 = Klof[64]; }
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.nppf;
    peb = p.pe_b[h];
    if (tid < 32) us[tid] = tid < p.nppf ? p.u[(u_base + tid) * p.H + h] : 0.f;
    if (q_ok) uq = p.u[(u_base + (qi % p.nppf)) * p.H + h];
  }
  __syncthreads();
  if (!wave_ok) return;
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;
  f32x16 sv, sl;
  {
    f32x16 s1, l1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] = 0.f; s1[r] = 0.f; sl[r] = 0.f; l1[r] = 0.f; }
    const float* qlr = pls + (qa * 3 + 0) * DP + hi * 8;
    const bool a_ok = ql < p.nsrl;
    const float* kr = pls + ((a_ok ? ql : 0) * 3 + 1) * DP + hi * 8;
    auto add_ql = [&](u16x8 v, int ks) -> u16x8 {       // q(a, p) = Qv[p] + Ql[a]
      if (!p.q_visual) return v;
      const float4 l0 = *reinterpret_cast<const float4*>(qlr + ks * 16);
      const float4 l1_ = *reinterpret_cast<const float4*>(qlr + ks * 16 + 4);
      return u16x8{to16<T16>(from16<T16>(v[0]) + l0.x), to16<T16>(from16<T16>(v[1]) + l0.y),
                   to16<T16>(from16<T16>(v[2]) + l0.z), to16<T16>(from16<T16>(v[3]) + l0.w),
                   to16<T16>(from16<T16>(v[4]) + l1_.x), to16<T16>(from16<T16>(v[5]) + l1_.y),
                   to16<T16>(from16<T16>(v[6]) + l1_.z), to16<T16>(from16<T16>(v[7]) + l1_.w)};
    };
    auto load_kl = [&](int ks) -> u16x8 {               // language K fragment: lane = key a
      float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
      if (a_ok) { x0 = *reinterpret_cast<const float4*>(kr + ks * 16); x1 = *reinterpret_cast<const float4*>(kr + ks * 16 + 4); }
      return u16x8{to16<T16>(x0.x), to16<T16>(x0.y), to16<T16>(x0.z), to16<T16>(x0.w),
                   to16<T16>(x1.x), to16<T16>(x1.y), to16<T16>(x1.z), to16<T16>(x1.w)};
    };
    // SPLIT: q(a, p) = (Qv + Qv_lo)[p] + Ql[a] formed in fp32 and split again; the language keys split from their fp32 rows
    auto make_q = [&](u16x8 v, u16x8 vl, int ks, u16x8& qh, u16x8& qlo) {
      const float4 l0 = *reinterpret_cast<const float4*>(qlr + ks * 16);
      const float4 l1_ = *reinterpret_cast<const float4*>(qlr + ks * 16 + 4);
      const float lq[8] = {l0.x, l0.y, l0.z, l0.w, l1_.x, l1_.y, l1_.z, l1_.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned short h_, l_;
        split16<T16>(from16<T16>(v[j]) + from16<T16>(vl[j]) + lq[j], h_, l_);
        qh[j] = h_; qlo[j] = l_;
      }
    };
    auto make_kl = [&](int ks, u16x8& kh, u16x8& klo) {
      float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
      if (a_ok) { x0 = *reinterpret_cast<const float4*>(kr + ks * 16); x1 = *reinterpret_cast<const float4*>(kr + ks * 16 + 4); }
      const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned short h_, l_;
        split16<T16>(xs[j], h_, l_);
        kh[j] = h_; klo[j] = l_;
      }
    };
#pragma unroll 1
    for (int ks = 0; ks < KS; ks += 2) {
      u16x8 q0, q1, ql0, ql1, kl0, kl1, kll0, kll1;
      const u16x8 k0 = nk0, k1 = nk1, kv0 = nkl0, kv1 = nkl1;
      if constexpr (SPLIT) {
        make_q(nq0, nql0, ks, q0, ql0); make_q(nq1, nql1, ks + 1, q1, ql1);
      } else {
        q0 = add_ql(nq0, ks); q1 = add_ql(nq1, ks + 1); ql0 = q0; ql1 = q1;
      }
      if (ks + 2 < KS) {                                  // the next round's global fragments
        nq0 = load_q(ks + 2); nq1 = load_q(ks + 3);
        nk0 = Kf[(ks + 2) * 64]; nk1 = Kf[(ks + 3) * 64];
        if constexpr (SPLIT) { nql0 = load_ql(ks + 2); nql1 = load_ql(ks + 3); nkl0 = Klof[(ks + 2) * 64]; nkl1 = Klof[(ks + 3) * 64]; }
      }
      if constexpr (SPLIT) { make_kl(ks, kl0, kll0); make_kl(ks + 1, kl1, kll1); }
      else { kl0 = load_kl(ks); kl1 = load_kl(ks + 1); kll0 = kl0; kll1 = kl1; }
      sv = qk_mfma<T16, SPLIT>(k0, q0, kv0, ql0, sv);
      sl = qk_mfma<T16, SPLIT>(kl0, q0, kll0, ql0, sl);
      s1 = qk_mfma<T16, SPLIT>(k1, q1, kv1, ql1, s1);
      l1 = qk_mfma<T16, SPLIT>(kl1, q1, kll1, ql1, l1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] += s1[r]; sl[r] += l1[r]; }
  }
  // ---- two independent softmaxes, probabilities normalised before P.V
  u16x8 pv_[2], pl_[2];
  {
    float mv = -1e30f, ml = -1e30f, av = 0.f, al = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = c32_row(r, lane);
      float x = sv[r];
      if (p.use_rel) x += fmaxf(uqp - us[key], 0.f);
      x = key < p.nppf ? x * c2 : -1e30f;
      const float y = key < p.nsrl ? sl[r] * c2 : -1e30f;
      sv[r] = x; sl[r] = y;
      mv = fmaxf(mv, x); ml = fmaxf(ml, y);
      av = fmaxf(av, key < p.nppf ? fabsf(x) : 0.f); al = fmaxf(al, key < p.nsrl ? fabsf(y) : 0.f);
    }
    // (the logit of key (a', p') is x[p'] + y[a']: the largest magnitude of the pair bounds it; log2 units -> nats)
    publish_logit_max(p.logit_max, lprev, q_ok ? (av + al) * 0.69314718056f : 0.f, lane);
    mv = fmaxf(mv, __shfl_xor(mv, 32));
    ml = fmaxf(ml, __shfl_xor(ml, 32));
    float lv_ = 0.f, ll = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sv[r] = __builtin_amdgcn_exp2f(sv[r] - mv); lv_ += sv[r];
      sl[r] = __builtin_amdgcn_exp2f(sl[r] - ml); ll += sl[r];
    }
    lv_ += __shfl_xor(lv_, 32);
    ll += __shfl_xor(ll, 32);
    const float iv = 1.0f / lv_, il = 1.0f / ll;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pv_[ks][j] = to16<T16>(sv[ks * 8 + j] * iv);
        pl_[ks][j] = to16<T16>(sl[ks * 8 + j] * il);
      }
  }
  // ---- output, one d-block at a time
  const int nksl = p.nsrl > 16 ? 2 : 1;
#pragma unroll 2
  for (int db = 0; db < NDB; ++db) {
    const u16x8 v0 = Vf[(db * 2) * 64], v1 = Vf[(db * 2 + 1) * 64];
    u16x8 w0;                                        // language V fragment from the staged rows
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = 8 * (j >> 2) + 4 * hi + (j & 3);
      w0[j] = key < p.nsrl ? to16<T16>(pls[(key * 3 + 2) * DP + db * 32 + ql]) : (unsigned short)0;
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    o = mfma32<T16>(v0, pv_[0], o);
    o = mfma32<T16>(v1, pv_[1], o);
    o = mfma32<T16>(w0, pl_[0], o);
    if (nksl > 1) o = mfma32<T16>(struct_load_vl<T16>(p, db, 1, lane, plr, hd, ldp), pl_[1], o);
    if (q_ok) struct_store<T16>(p, o, db, (int64_t)s * Nq + qi, h, DP, hi);
  }
}

// ---- ONE visual key block, queries formed in the kernel, every operand in LDS after ONE round trip (round 6). The lean form
// above walks the head dimension in KS / 2 rounds of "fragments of the next round in flight"; at cfg 2 the launch is 120
// workgroups - one per CU, nothing to hide those rounds behind - and per wave it spent (wall-clock stamps, scratch/ts_attn6.hip)
// 2.2 us in Q.K^T (per k-step: fp32 adds and conversions to form q = Qv + Ql and the language keys, in every wave again),
// 4.1 us in P.V (8 conditional LDS reads per d-block, each behind its own branch and wait) and 1.2 us in the softmax. Here:
//  * the workgroup's K, Qv and V^T blocks (KS + KS + 2 NDB fragments of 1 KiB: 48 KiB at head dim 256 - the four waves share
//    them) go global -> LDS by LDS-DMA, all requested before anything is waited for;
//  * the language rows are converted ONCE per workgroup on their way into LDS: Ql and Kl as 16-bit rows (MFMA operands read
//    with one 16-byte LDS load), Vl directly as the P.V product's A fragments;
//  * q.k = (Qv + Ql).(k) is two MFMAs on 16-bit operands instead of one MFMA behind 24 VALU instructions (linearity; each part is
//    rounded on its own, which is no worse than rounding the sum);
//  * head dim 256: the output rows leave through LDS tiles (the K / Qv images are free by then), 4 rows x 256 B per store.
// nsrl <= 16 (one language key step). The hi + lo plan and other shapes keep the lean kernel.
#ifndef VOG_DMA_TSTORE
#define VOG_DMA_TSTORE 1
#endif
template <typename T16, int NDB>
__global__ __launch_bounds__(256, 2) void attn_struct1_dma_kernel(AttnStructParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  constexpr int NFRAG = 2 * KS + 2 * NDB;              // K, Qv, V^T
  const unsigned int lprev = logit_prev(p.logit_max);   // (in flight behind the kernel: publish_logit_max)
  extern __shared__ __attribute__((aligned(1024))) unsigned char dsm[];
  unsigned char* Kl = dsm;                             // [KS][1024]
  unsigned char* Ql = Kl + KS * 1024;                  // [KS][1024]  Qv of the frame's proposals (fragment order)
  unsigned char* Vl = Ql + KS * 1024;                  // [2 NDB][1024]
  unsigned char* Wl = Vl + 2 * NDB * 1024;             // [NDB][1024]  language V as A fragments of the P.V product
  unsigned short* QL16 = reinterpret_cast<unsigned short*>(Wl + NDB * 1024);   // [nsrl + 1][DP] 16-bit Ql rows (+ a zero row)
  unsigned short* KL16 = QL16 + (p.nsrl + 1) * DP;                             // [nsrl + 1][DP] 16-bit Kl rows (+ a zero row)
  float* us = reinterpret_cast<float*>(KL16 + (p.nsrl + 1) * DP);              // [32] bias precursor of the visual keys
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int Nq = p.nsrl * p.nppf;
  const int nqb = (Nq + 31) >> 5, nqg = (nqb + 3) >> 2;
  const int pair = blockIdx.x / nqg, qg = blockIdx.x - pair * nqg;
  const int s = pair / p.H, h = pair - s * p.H;
  const int qb = qg * 4 + wid;
  const bool wave_ok = qb < nqb;
  const int qi = qb * 32 + ql;
  const bool q_ok = wave_ok && qi < Nq;
  const int hd = p.H * DP, ldp = 3 * hd;
  const int vid = s / p.nfrm;
  const int lv = p.lpv ? vid : vid / p.ncv;
  const float* plr = p.pl + (int64_t)lv * p.nsrl * ldp + h * DP;     // + hd: K block, + 2*hd: V block
  VOG_ATS(0);
  const int64_t kvbase = ((int64_t)s * p.H + h) * (int64_t)32 * DP;  // (npad_kv == 32)
  {
    const unsigned short* src[3] = {p.kv + kvbase, p.q + kvbase, p.vv + kvbase};
#pragma unroll
    for (int i = 0; i < (NFRAG + 3) / 4; ++i) {
      int f = wid + 4 * i;                             // fragment of the [K | Qv | V^T] image
      f = f < NFRAG ? f : NFRAG - 1;                   // (tail waves repeat the last one)
      const int which = f < KS ? 0 : (f < 2 * KS ? 1 : 2);
      const int fi = f - (which == 0 ? 0 : (which == 1 ? KS : 2 * KS));
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src[which] + (int64_t)fi * 512 + lane * 8),
          (__attribute__((address_space(3))) void*)(dsm + (size_t)f * 1024), 16, 0, 0);
    }
  }
  {
    // Ql / Kl rows -> 16 bit (row nsrl of each = zeros: what the key lanes past nsrl read)
    const int per_part = DP / 4, per_row = 2 * per_part;            // float4 per (argument, part)
    stage_batched<3, 256, float4>(p.nsrl * per_row, tid,
        [&](int i) {
          const int a = i / per_row, c = i - a * per_row;
          const int part = c / per_part, dd4 = c - part * per_part;
          return *reinterpret_cast<const float4*>(plr + (int64_t)a * ldp + part * hd + dd4 * 4);
        },
        [&](int i, const float4& v) {
          const int a = i / per_row, c = i - a * per_row;
          const int part = c / per_part, dd4 = c - part * per_part;
          *reinterpret_cast<u16x4*>((part ? KL16 : QL16) + a * DP + dd4 * 4) =
              u16x4{to16<T16>(v.x), to16<T16>(v.y), to16<T16>(v.z), to16<T16>(v.w)};
        });
    for (int i = tid; i < DP / 4; i += 256) {
      *reinterpret_cast<u16x4*>(QL16 + p.nsrl * DP + i * 4) = u16x4{0, 0, 0, 0};
      *reinterpret_cast<u16x4*>(KL16 + p.nsrl * DP + i * 4) = u16x4{0, 0, 0, 0};
    }
    // Vl -> the A fragments of the language P.V product: fragment db, lane (d = db * 32 + (lane & 31)), element j = key
    // 8 (j >> 2) + 4 (lane >> 5) + (j & 3); one (d-block, lane) per thread and pass, its 8 row reads in flight together
    const float* vrow = plr + 2 * hd;
#pragma unroll
    for (int it = 0; it < (NDB * 64 + 255) / 256; ++it) {
      const int e = tid + it * 256;
      const int db = (e >> 6) < NDB ? (e >> 6) : NDB - 1, ln = e & 63;
      const int hh = ln >> 5, dcol = db * 32 + (ln & 31);
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = 8 * (j >> 2) + 4 * hh + (j & 3);
        wv[j] = vrow[(int64_t)(key < p.nsrl ? key : 0) * ldp + dcol];
      }
      u16x8 w;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = 8 * (j >> 2) + 4 * hh + (j & 3);
        w[j] = key < p.nsrl ? to16<T16>(wv[j]) : (unsigned short)0;
      }
      if (e < NDB * 64) *reinterpret_cast<u16x8*>(Wl + (size_t)e * 16) = w;
    }
  }
  const int t = (wave_ok ? qb : 0) * 32 + ql;
  int qa = t / p.nppf;
  const int qp = t - qa * p.nppf;
  qa = qa < p.nsrl ? qa : p.nsrl - 1;                // tokens past the end are never stored
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.nppf;
    peb = p.pe_b[h];
    if (tid < 32) us[tid] = p.u[(u_base + (tid < p.nppf ? tid : 0)) * p.H + h];
    if (q_ok) uq = p.u[(u_base + (qi % p.nppf)) * p.H + h];
  }
  VOG_ATS(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the DMA image
  __syncthreads();                                   // ... everybody's, and the staged rows
  VOG_ATS(2);
  if (!wave_ok) return;
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;
  f32x16 sv, sl;
  {
    f32x16 s1, l1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] = 0.f; s1[r] = 0.f; sl[r] = 0.f; l1[r] = 0.f; }
    const unsigned char* qsrc = Ql + frag_qk(qp, hi * 8, DP) * 2;        // + ks * 1024: Qv[p], this lane's 8 columns of k-step ks
    const unsigned short* qlsrc = QL16 + qa * DP + hi * 8;               // + ks * 16:   Ql[a], the same columns
    const unsigned char* ksrc = Kl + lane * 16;
    const unsigned short* klsrc = KL16 + (ql < p.nsrl ? ql : p.nsrl) * DP + hi * 8;   // language key a = lane (zero row past nsrl)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u16x8 qv = *reinterpret_cast<const u16x8*>(qsrc + ks * 1024);
      const u16x8 qlg = *reinterpret_cast<const u16x8*>(qlsrc + ks * 16);
      const u16x8 k = *reinterpret_cast<const u16x8*>(ksrc + ks * 1024);
      const u16x8 kl = *reinterpret_cast<const u16x8*>(klsrc + ks * 16);
      sv = mfma32<T16>(k, qv, sv);                   // q.k = (Qv + Ql).k: two products on 16-bit operands
      s1 = mfma32<T16>(k, qlg, s1);
      sl = mfma32<T16>(kl, qv, sl);
      l1 = mfma32<T16>(kl, qlg, l1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] += s1[r]; sl[r] += l1[r]; }
  }
  VOG_ATS(3);
  // ---- two independent softmaxes, probabilities normalised before P.V
  u16x8 pv_[2], pl_;
  {
    float mv = -1e30f, ml = -1e30f, av = 0.f, al = 0.f;
    float ub[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ub[r] = us[c32_row(r, lane)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = c32_row(r, lane);
      float x = sv[r];
      if (p.use_rel) x += fmaxf(uqp - ub[r], 0.f);
      x = key < p.nppf ? x * c2 : -1e30f;
      const float y = key < p.nsrl ? sl[r] * c2 : -1e30f;
      sv[r] = x; sl[r] = y;
      mv = fmaxf(mv, x); ml = fmaxf(ml, y);
      av = fmaxf(av, key < p.nppf ? fabsf(x) : 0.f); al = fmaxf(al, key < p.nsrl ? fabsf(y) : 0.f);
    }
    // (the logit of key (a', p') is x[p'] + y[a']: the largest magnitude of the pair bounds it; log2 units -> nats)
    publish_logit_max(p.logit_max, lprev, q_ok ? (av + al) * 0.69314718056f : 0.f, lane);
    mv = fmaxf(mv, __shfl_xor(mv, 32));
    ml = fmaxf(ml, __shfl_xor(ml, 32));
    float lv_ = 0.f, ll = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sv[r] = __builtin_amdgcn_exp2f(sv[r] - mv); lv_ += sv[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {                    // (nsrl <= 16: the language keys are the first k-step's rows)
      sl[r] = __builtin_amdgcn_exp2f(sl[r] - ml); ll += sl[r];
    }
    lv_ += __shfl_xor(lv_, 32);
    ll += __shfl_xor(ll, 32);
    const float iv = 1.0f / lv_, il = 1.0f / ll;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pv_[0][j] = to16<T16>(sv[j] * iv);
      pv_[1][j] = to16<T16>(sv[8 + j] * iv);
      pl_[j] = to16<T16>(sl[j] * il);
    }
  }
  VOG_ATS(4);
  // ---- output, one d-block at a time. Head dim 256 (VOG_DMA_TSTORE): the rows leave through LDS, four d-blocks (256 bytes of
  // a row) at a time, so that a store instruction writes 4 rows x 256 contiguous bytes; straight from the accumulator layout a
  // lane owns 4 columns of its row and an instruction scatters 16-byte pieces over 32 rows. The tile of wave w is the w-th
  // quarter of the K / Qv images, which nobody reads after the Q.K^T phase (barrier below).
  const unsigned char* vsrc = Vl + lane * 16;
  const unsigned char* wsrc = Wl + lane * 16;
  constexpr bool TSTORE = NDB == 8 && VOG_DMA_TSTORE;  // (4 tiles of 8 KiB = the K + Qv images at head dim 256)
  unsigned char* tile = dsm + wid * 8192;             // [32 rows][256 B], 16-byte chunks XOR-swizzled by the row
  if constexpr (TSTORE) __syncthreads();              // (waves without a query block have left; the barrier counts live waves)
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    const u16x8 v0 = *reinterpret_cast<const u16x8*>(vsrc + (db * 2) * 1024), v1 = *reinterpret_cast<const u16x8*>(vsrc + (db * 2 + 1) * 1024);
    const u16x8 w0 = *reinterpret_cast<const u16x8*>(wsrc + db * 1024);
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    o = mfma32<T16>(v0, pv_[0], o);
    o = mfma32<T16>(v1, pv_[1], o);
    o = mfma32<T16>(w0, pl_, o);
    if constexpr (!TSTORE) {
      if (q_ok) struct_store<T16>(p, o, db, (int64_t)s * Nq + qi, h, DP, hi);
    } else {
      const int j4 = db & 3;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u16x4 v = {to16<T16>(o[g * 4]), to16<T16>(o[g * 4 + 1]), to16<T16>(o[g * 4 + 2]), to16<T16>(o[g * 4 + 3])};
        *reinterpret_cast<u16x4*>(tile + ql * 256 + (((j4 * 4 + g) ^ (ql & 15)) << 4) + hi * 8) = v;
      }
      if (j4 == 3) {                                 // four d-blocks parked: 32 rows x 256 bytes, 8 instructions of 4 rows each
        const int64_t o0 = ((int64_t)s * Nq + qb * 32) * ((int64_t)p.H * DP) + (int64_t)h * DP + (db - 3) * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int row = i * 4 + (lane >> 4), ch = lane & 15;
          const u16x8 v = *reinterpret_cast<const u16x8*>(tile + row * 256 + ((ch ^ (row & 15)) << 4));
          if (qb * 32 + row < Nq) *reinterpret_cast<u16x8*>(p.out + o0 + (int64_t)row * ((int64_t)p.H * DP) + ch * 8) = v;
        }
      }
    }
  }
  VOG_ATS(5);
}

// ---- general form: flash loop over the visual key blocks (p100), then the language block
template <typename T16, int NDB>
__global__ __launch_bounds__(256, (NDB <= 4 ? 2 : 1)) void attn_struct_kernel(AttnStructParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  extern __shared__ __attribute__((aligned(16))) float ssm[];
  float* us = ssm;                                   // [npad_kv] bias precursor of the visual keys
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int Nq = p.nsrl * p.nppf;
  const int nqb = (Nq + 31) >> 5, nqg = (nqb + 3) >> 2;
  int pair, qg;
  {   // XCD-aware (block b runs on XCD b % 8): the query groups of one (sequence, head) share an XCD, so
      // its K / V fragments come from HBM once and from that L2 for the other groups (in plain block
      // order the nqg groups of a pair sat on nqg different XCDs: 539 MB fetched at p100 for 63 MB of K/V)
    const int b = blockIdx.x, npair = p.S * p.H;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * nqg);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); qg = (b >> 3) % nqg; }
    else { const int r = b - full * nqg; pair = full + r / nqg; qg = r % nqg; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int qb = qg * 4 + wid;
  const bool wave_ok = qb < nqb;
  const int qi = qb * 32 + ql;
  const bool q_ok = wave_ok && qi < Nq;
  const int nkb = p.npad_kv >> 5;
  const int hd = p.H * DP, ldp = 3 * hd;
  const int vid = s / p.nfrm;
  const int lv = p.lpv ? vid : vid / p.ncv;
  const float* plr = p.pl + (int64_t)lv * p.nsrl * ldp + h * DP;     // + hd: K block, + 2*hd: V block
  const int64_t kvbase = ((int64_t)s * p.H + h) * (int64_t)p.npad_kv * DP;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.kv + kvbase) + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vv + kvbase) + lane;
  u16x8 qf[KS];
  struct_load_q<T16, KS>(p, qf, s, h, wave_ok ? qb : 0, lane, plr, ldp, kvbase);
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.nppf;
    peb = p.pe_b[h];
    stage_batched<2, 256, float>(p.npad_kv, tid,
        [&](int key) { return p.u[(u_base + (key < p.nppf ? key : 0)) * p.H + h]; },
        [&](int key, float v) { us[key] = key < p.nppf ? v : 0.f; });
    if (q_ok) uq = p.u[(u_base + (qi % p.nppf)) * p.H + h];
  }
  __syncthreads();
  if (!wave_ok) return;
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;

  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  for (int kb = 0; kb < nkb; ++kb) {
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    {
      u16x8 kf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[ks] = Kf[((int64_t)kb * KS + ks) * 64];
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        s0 = mfma32<T16>(kf[ks], qf[ks], s0);
        if (ks + 1 < KS) s1 = mfma32<T16>(kf[ks + 1], qf[ks + 1], s1);
      }
    }
    u16x8 vf[NDB * 2];
#pragma unroll
    for (int i = 0; i < NDB * 2; ++i) vf[i] = Vf[((int64_t)kb * NDB * 2 + i) * 64];
    f32x16 sacc;
    float mloc = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + c32_row(r, lane);
      float x = s0[r] + s1[r];
      if (p.use_rel) x += fmaxf(uqp - us[key], 0.f);
      x = key < p.nppf ? x * c2 : -1e30f;
      sacc[r] = x;
      mloc = fmaxf(mloc, x);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sacc[r] - m_new);
      sacc[r] = e;
      lsum += e;
    }
    lsum += __shfl_xor(lsum, 32);
    l_run = l_run * alpha + lsum;
    m_run = m_new;
    if (kb > 0 && !__all(alpha == 1.0f)) {
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    }
    u16x8 pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[ks][j] = to16<T16>(sacc[ks * 8 + j]);
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) o[db] = mfma32<T16>(vf[db * 2 + ks], pf[ks], o[db]);
  }
  {
    const float inv_l = 1.0f / l_run;                // normalise the visual part in place
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= inv_l;
  }
  // ---- language keys: one masked block, its own softmax, probabilities normalised before P.V so
  // that it accumulates into the normalised visual output
  {
    u16x8 klf[KS];
    struct_load_kl<T16, KS>(p, klf, lane, plr, hd, ldp);
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      s0 = mfma32<T16>(klf[ks], qf[ks], s0);
      if (ks + 1 < KS) s1 = mfma32<T16>(klf[ks + 1], qf[ks + 1], s1);
    }
    f32x16 sacc;
    float m2 = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = c32_row(r, lane);
      const float x = key < p.nsrl ? (s0[r] + s1[r]) * c2 : -1e30f;
      sacc[r] = x;
      m2 = fmaxf(m2, x);
    }
    m2 = fmaxf(m2, __shfl_xor(m2, 32));
    float l2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sacc[r] - m2);
      sacc[r] = e;
      l2 += e;
    }
    l2 += __shfl_xor(l2, 32);
    const float inv_l2 = 1.0f / l2;
    u16x8 pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[ks][j] = to16<T16>(sacc[ks * 8 + j] * inv_l2);
    const int nksl = p.nsrl > 16 ? 2 : 1;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      o[db] = mfma32<T16>(struct_load_vl<T16>(p, db, 0, lane, plr, hd, ldp), pf[0], o[db]);
      if (nksl > 1) o[db] = mfma32<T16>(struct_load_vl<T16>(p, db, 1, lane, plr, hd, ldp), pf[1], o[db]);
    }
  }
  if (q_ok) {
#pragma unroll
    for (int db = 0; db < NDB; ++db)
      struct_store<T16>(p, o[db], db, (int64_t)s * Nq + qi, h, DP, hi);
  }
}

}  // namespace vog
