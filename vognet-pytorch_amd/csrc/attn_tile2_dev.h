// Long-sequence RelAttention (p100: 2000 / 4000 tokens per sequence), second form.
//
// attn_tile_kernel (attention_dev.h) runs 4 waves per workgroup = ONE wave per SIMD with the classic
// online softmax. Measured on gfx950 (scratch/ubench/mfma_rate.hip, lds_bw.hip):
//   * back-to-back v_mfma_f32_32x32x16_bf16 sustain one per ~46 nominal cycles per SIMD (1.7-1.9
//     PFLOP/s whole chip: the clock gives way under full matrix load), not the 32 of the data sheet;
//   * VALU work of the SAME or the OTHER wave of a SIMD overlaps an MFMA stream only by about half
//     (MFMA + 12 fma per MFMA: 65-70 cycles per slot, against 47 alone and 51 for the fmas alone) -
//     the ~700 VALU cycles of a key block's softmax are mostly ADDED to its 24 MFMAs;
//   * ds_read_b128 delivers > 200 B/clk per CU with 8 waves, but a single outstanding read costs
//     ~130 cycles: fragment reads have to be requested several MFMAs ahead;
//   * hipcc puts s_waitcnt vmcnt(0) in front of the first LDS read after every LDS-DMA issue (it
//     cannot tell which bytes the DMA will write): a "prefetched" key block is waited for at once.
// Result: 11 % of the nominal MFMA peak on obj_tx at p100 (471 us).
// This form:
//   * 8 waves per workgroup = TWO waves per SIMD (32 queries each, 256 per workgroup, <= 256
//     registers): while one wave of a SIMD does its softmax the other one owns the matrix pipe.
//     s_setprio raises a wave for its MFMA bursts so they are not queued behind the neighbour's VALU.
//   * NO running maximum: exponentials are taken against a FIXED per-query reference m_ref = the row
//     maximum over key block 0 (O and the row sum l carry the same factor 2^-m_ref; only O / l leaves
//     the kernel). That removes the max / rescale work (a logit costs sub, max, fma, exp, add) and
//     the O *= alpha pass. P may exceed 1 when a later block holds a larger logit; a row whose block
//     sum passes 2^40 (a logit 28 nats above everything in block 0 - or anything non-finite) raises
//     p.guard and the host-side sequence (attention.hip) re-runs the call with attn_tile_kernel,
//     whose running maximum is safe for any input. f16 (round 5): the same with the trip point at a block sum of 2^15.
//   * K / V^T of NBUF key blocks live in an LDS ring filled by LDS-DMA two blocks ahead; one barrier
//     per key block, counted vmcnt; every LDS read in the loop goes through lds_read128 / lds_wait
//     (common.h) so the DMA really stays in flight, with three fragments requested ahead.
// 217 us for the same launch (605 TFLOP/s on the 171 real head columns, 680 on the padded 192). Two
// other forms measured the same: 64 queries per wave with the softmax of one 32-query half software-
// pipelined between the MFMAs of the other (460 registers), with and without fine interleaving -
// the kernel sits on the MFMA-plus-half-the-VALU issue bound above, not on LDS or L2.
// The bias precursors are pre-multiplied by inv_scale * log2(e) (positive):
//   p = 2^((s + relu(uq - uk)) * c - m_ref) = 2^fma(s, c, max(uq*c - m_ref - uk*c, -m_ref)).
#pragma once
#include "attention_dev.h"

namespace vog {

// Body form (common.h): at p100 its 192 workgroups (4 sequences x 3 heads x 16 query groups) and the 64 of a persistent
// BiLSTM layer fill the chip exactly - the layer rides inside this launch (pair.hip, round 5).
template <typename T16, int NDB>
struct AttnTile2Body {
  using Params = AttnParams;
  static constexpr int THREADS = 512;
  static __device__ __forceinline__ void run(const AttnParams& p, const BlockCtx& cx, unsigned char* t2sm) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  const unsigned int lprev = logit_prev(p.logit_max);   // (in flight behind the kernel: publish_logit_max)
  constexpr int NF = KS + 2 * NDB;                   // KiB fragments per key block (K then V^T)
  constexpr int NBUF = 4, DIST = 2;                  // ring depth; blocks requested ahead
  constexpr int FPW = (NF + 7) / 8;                  // DMA instructions per wave per block (tail waves repeat the last fragment)
  constexpr int PF = 3;                              // LDS fragments requested ahead of their MFMA
  unsigned char* kv = t2sm;                          // [NBUF][NF][1024]
  float* us = reinterpret_cast<float*>(t2sm + NBUF * NF * 1024);   // [npad] key bias precursors * c2
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int nkb = (p.N + 31) >> 5;
  const int nqg = (p.N + 255) >> 8;                  // 256-query groups per (sequence, head)
  const int npair = p.S * p.H;
  int pair, qg;
  {   // XCD-aware: the query groups of one (sequence, head) stay on one XCD (its K/V in one L2)
    const int b = cx.bx;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * nqg);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); qg = (b >> 3) % nqg; }
    else { const int r = b - full * nqg; pair = full + r / nqg; qg = r % nqg; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int qb = qg * 8 + wid;                       // this wave's 32-query block
  const bool wave_ok = qb < nkb;                     // a wave past the end still helps with the DMA
  const int qi = qb * 32 + ql;
  const int64_t base = ((int64_t)s * p.H + h) * (int64_t)p.npad * DP;
  const unsigned short* Kg = p.k + base;
  const unsigned short* Vg = p.vt + base;
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)

  float uq = 0.f;
  {
    float peb = 0.f;
    int64_t u_base = 0;
    if (p.use_rel) {
      u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.n_box;
      peb = p.pe_b[h];
    }
    if (p.use_rel)
      stage_batched<8, 512, float>(p.npad, tid,
          [&](int key) { return p.u[(u_base + ((key < p.N ? key : 0) % p.n_box)) * p.H + h]; },
          [&](int key, float v) { us[key] = key < p.N ? v * c2 : 0.f; });
    else
      for (int key = tid; key < p.npad; key += 512) us[key] = 0.f;
    if (p.use_rel && qi < p.N) uq = (p.u[(u_base + (qi % p.n_box)) * p.H + h] + peb) * c2;
  }
  // Q fragments of this wave's block: registers for the whole pass
  u16x8 qf[KS];
  {
    const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + base) + (int64_t)(wave_ok ? qb : 0) * KS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = Qf[ks * 64];
  }
  // fragment f of key block kb: f < KS -> K fragment, else V^T fragment f - KS. Blocks past the end
  // re-load the last one (keeps the vmcnt bookkeeping uniform; never consumed).
  auto issue = [&](int kb) __attribute__((always_inline)) {
    const int buf = kb % NBUF;
    const int src_kb = kb < nkb ? kb : nkb - 1;
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      int f = wid * FPW + i;
      f = f < NF ? f : NF - 1;
      const unsigned short* src = f < KS ? Kg + ((int64_t)src_kb * KS + f) * 512
                                         : Vg + ((int64_t)src_kb * NDB * 2 + (f - KS)) * 512;
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
  float l_run = 0.f, nm = 0.f, uqm = 0.f;
  bool trip = false;

  // S^T(kb) = K(kb) Q^T for this wave's 32 queries: KS MFMAs, fragments PF ahead
  auto s_phase = [&](int kb, f32x16& sv) __attribute__((always_inline)) {
    const unsigned char* kblk = kv + ((kb % NBUF) * NF) * 1024 + lane * 16;
    u16x8 fr[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) fr[j] = lds_read128(kblk + (j < KS ? j : KS - 1) * 1024);
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = 0.f;
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // reads complete in order: fragment ks has landed when at most the newer ones are outstanding
      if (ks + PF <= KS) lds_wait<PF - 1>(fr[ks % PF]);
      else if (ks + 2 == KS) lds_wait<1>(fr[ks % PF]);
      else lds_wait<0>(fr[ks % PF]);
      const u16x8 kf = fr[ks % PF];
      if (ks + PF < KS) fr[ks % PF] = lds_read128(kblk + (ks + PF) * 1024);
      sv = mfma32<T16>(kf, qf[ks], sv);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  __syncthreads();                                   // us[] is in place
  issue(0); issue(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FPW) : "memory");
  __syncthreads();                                   // block 0 is in LDS
  {   // m_ref = row maximum over key block 0 (keys >= N only if the sequence has one block)
    f32x16 s0;
    s_phase(0, s0);
    float mx = -3.0e38f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 ub = lds_read_f4(&us[8 * g + 4 * hi]);
      lds_wait<0>(ub);
      const float ubv[4] = {ub[0], ub[1], ub[2], ub[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = 8 * g + 4 * hi + e < p.N;
        mx = fmaxf(mx, ok ? fmaf(s0[4 * g + e], c2, fmaxf(uq - ubv[e], 0.f)) : -3.0e38f);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // (run-time logit-scale report, AttnParams::logit_max: this kernel's softmax never looks at single logits again, so what it
    // reports is the magnitude of each row's reference - the maximum over its first 32 keys - a LOWER bound of the row's largest
    // |logit| (typically within 2x: 2 sigma against ~4 sigma over 4000 keys); one value per wave, log2 units -> nats)
    publish_logit_max(p.logit_max, lprev, (wave_ok && qi < p.N) ? fabsf(mx) * 0.69314718056f : 0.f, lane);
    // f16: the reference sits 4 binary orders ABOVE block 0's maximum - every P, row sum and accumulator carries 2^-4, which
    // cancels in O / l - so that the headroom to f16's 65504 is 19 binary orders (13 nats above block 0) instead of 15; block
    // 0's own maximum becomes 2^-4 and P stays a normal f16 number down to 2^-10 of it (below: absolute error 2^-25 per key)
    if constexpr (!std::is_same<T16, BF16>::value) mx += 4.0f;
    nm = -mx;
    uqm = uq - mx;
  }

  for (int kb = 0; kb < nkb; ++kb) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FPW * (DIST - 1)) : "memory");   // block kb landed (kb+1 may be in flight)
    __builtin_amdgcn_s_barrier();                                              // ... everybody's share; block kb-1 is consumed
    asm volatile("" ::: "memory");
    issue(kb + DIST);
    if (!wave_ok) continue;
    f32x16 sv;
    s_phase(kb, sv);
    // bias quads of this block (landed by the first lds_wait below; LDS returns in order)
    f32x4 ub[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ub[g] = lds_read_f4(&us[kb * 32 + 8 * g + 4 * hi]);
    // the V^T fragments are requested before the softmax: its VALU work covers their latency
    const unsigned char* vblk = kv + ((kb % NBUF) * NF + KS) * 1024 + lane * 16;
    u16x8 fr[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) fr[j] = lds_read128(vblk + (j < 2 * NDB ? j : 2 * NDB - 1) * 1024);
    lds_wait<PF>(ub[0], ub[1]);
    lds_wait<PF>(ub[2], ub[3]);
    if (kb == nkb - 1) {                             // keys >= N exist only in the last block
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + c32_row(r, lane) >= p.N) sv[r] = -3.0e38f;   // * c2 stays finite, 2^x = 0
    }
    u16x8 pf[2];
    float lsum = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float ubv[4] = {ub[g][0], ub[g][1], ub[g][2], ub[g][3]};
      float ev[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ev[e] = __builtin_amdgcn_exp2f(fmaf(sv[4 * g + e], c2, fmaxf(uqm - ubv[e], nm)));
        lsum += ev[e];
        pf[g >> 1][(g & 1) * 4 + e] = to16<T16>(ev[e]);
      }
    }
    lsum += __shfl_xor(lsum, 32);
    // bf16 P keeps fp32's exponent range: 2^40. f16 (round 5: the package default) holds 65504: a block sum under 2^15
    // bounds every P of the block (a logit 10.4 nats above everything in block 0 sends the call to the fallback); values
    // far UNDER the reference lose nothing that matters - the row's true maximum is >= the reference (block 0 is part of
    // the row), so a P below f16's normal range (2^-14) weighs < 2^-14 of the row. Also catches inf / NaN.
    constexpr float TRIP = std::is_same<T16, BF16>::value ? 1.0995e12f : 32768.0f;
    trip = trip || !(lsum < TRIP);
    l_run += lsum;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int j = 0; j < 2 * NDB; ++j) {
      if (j + PF <= 2 * NDB) lds_wait<PF - 1>(fr[j % PF]);
      else if (j + 2 == 2 * NDB) lds_wait<1>(fr[j % PF]);
      else lds_wait<0>(fr[j % PF]);
      const u16x8 vf = fr[j % PF];
      if (j + PF < 2 * NDB) fr[j % PF] = lds_read128(vblk + (j + PF) * 1024);
      o[j >> 1] = mfma32<T16>(vf, pf[j & 1], o[j >> 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the wave
  if (!wave_ok) return;
  if (__any(trip)) { if (lane == 0) atomicOr(p.guard, 1); }   // the call is re-run by attn_tile_kernel
  if (qi < p.N) {
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

template <typename T16, int NDB>
__global__ __launch_bounds__(512) void attn_tile2_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char t2sm_k[];
  AttnTile2Body<T16, NDB>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, t2sm_k);
}

}  // namespace vog
