// Separable mul_tx layer-0 attention, SECOND factorisation (round 4): several visual key blocks (p100).
//
// Token (a, p) of mul_tx's layer 0 is [vis[p] || lang[a]], so q(a, p) = Qv[p] + Ql[a], k(a', p') = Kv[p'] + Kl[a'] and
// the bias depends on (p, p') only. attention_dev.h already splits the softmax over the nsrl * nppf keys into one over
// the nppf visual keys and one over the nsrl language keys (exact). The visual logits split once more:
//
//     x((a, p), p') = [Qv[p].Kv[p'] + bias(p, p')] + Ql[a].Kv[p'] = A[p, p'] + B[a, p']
//     exp(x - mA[p] - mB[a]) = E[p, p'] * F[a, p'],   E = exp(A - mA[p]),  F = exp(B - mB[a])
//
// so the nsrl queries that share a proposal p share ONE row of Qv.Kv^T and ONE row of exponentials:
//     out_v(a, p) = sum_p' E[p, p'] F[a, p'] Vv[p'] / sum_p' E[p, p'] F[a, p']
// (any per-row shift cancels in the quotient; E, F <= 1). Per 32 proposals and 400 visual keys: 1/5 of the Q.K^T MFMAs
// and 1/5 of the exponentials of the per-(a, p) kernels (attn_struct_lds_kernel: 330 us at cfg 4, 16 % MFMA busy);
// the P.V product keeps its size (P differs per (a, p)) and becomes the kernel. The language logits split the same way:
//     y((a, p), a') = Qv[p].Kl[a'] + Ql[a].Kl[a'] = C[p, a'] + D[a, a'].
//
// One workgroup = one (sequence, head, block of 32 proposals) = 32 * 5 queries; 4 waves.
//   phase 1  the waves split the KEY blocks (kb = wave + 4 i): A (kept in registers) and B tiles with K fragments straight from
//            L2 (every K fragment is read by exactly one wave); block maxima, B, C, D to LDS
//   phase 2  E = exp2(A - max over all blocks) -> LDS (f16, P^T fragment order); F = exp2(B - max) in place in LDS
//   phase 3  in rounds of 2 key blocks: wave w turns one 16-key half of a block (4 halves per round) into the 5 P^T
//            fragments P_a = E * F_a (fp32 product, exact row sums, one 16-bit rounding) and parks them in LDS; then the
//            waves split the OUTPUT d-blocks: V^T fragments of a key block straight from L2 (every one read by exactly
//            one wave) x the 10 parked fragments of that block -> 5 x DPW accumulators. The probabilities are computed
//            ONCE per workgroup (the first form of this kernel formed them in every wave: 18 VALU instructions per MFMA,
//            205 us at cfg 4). Outputs leave through LDS: 16-byte stores, 8 lanes per 128 contiguous bytes of a row.
// WPE = 2: <= 256 registers, ~79 KB of LDS: two workgroups per CU, whose phases overlap.
#pragma once
#include "attention_dev.h"

namespace vog {

constexpr int EF_MAXA = 5;          // arguments per query set (cfg.misc.srl_arg_length)


// E is parked as f16 and P = E * F is rounded to the MFMA operand type, both relative to the shift mA[p] + mB[a], which can sit far
// above a row's true maximum max_p'(A + B) (a key that is weak in A but dominant through B: sharp trained attention, the -13.8 of
// use_rel's log(clamp(., 1e-6))). Unscaled, E flushed to zero 24 binary orders (16.6 nats) under mA and P lost precision from 14
// under the shift (f16 subnormals) - silently, there is no fallback on this path. E is therefore stored as
// exp2(A - mA + EF_ESHIFT): every P, every row sum and every output accumulator carries the same factor 2^12, which cancels in
// out = (P V) / sum P. Largest value 4096 (exact in f16 / bf16); full precision now reaches 26 binary orders (18 nats) under the
// shift, flush-to-zero 36 (25 nats). Rows that may lie further down raise the guard flag (behind phase 2) and the launch is
// redone by the per-row kernel.
constexpr float EF_ESHIFT = 12.0f;
constexpr float EF_GUARD_F = 5.9604644775390625e-08f;   // 2^-24: see the guard behind phase 2
constexpr float EF_GUARD_E = 0.000244140625f;           // 2^-12 = 2^(EF_ESHIFT - 24)

constexpr int EF_PLS_PAD = 16;      // floats between the language rows of two arguments (the 5 rows a lane group reads at once
                                    // sat 3 * DP floats apart = on the same banks)

template <int NDB>
static inline size_t attn_struct_ef_lds(int npad_kv) {
  const int nkb = npad_kv >> 5;
  return (size_t)EF_MAXA * (3 * NDB * 32 + EF_PLS_PAD) * 4   // pls
         + (size_t)npad_kv * 4                    // us
         + (size_t)EF_MAXA * npad_kv * 4          // Bl / F
         + (size_t)nkb * 32 * 4                   // block maxima
         + (size_t)16 * 64 * 4                    // C tile (accumulator layout)
         + (size_t)EF_MAXA * 8 * 4                // D
         + (size_t)(32 + 8) * 4                   // kA / kB: a key at which row p of A / row a of B takes its maximum (guard)
         + (size_t)(nkb < 3 ? 3 : nkb) * 2 * 64 * 16   // E fragments (f16); later the row-sum partials [a][wave][lane]
         + (size_t)2 * 2 * EF_MAXA * 64 * 16;     // P^T fragments of one round: [2 blocks][2][a][64 lanes] x 16 B; later the output staging
}

// WPE = waves per SIMD the register allocation aims at: 1 -> up to 512 registers, one workgroup per CU (no spills; 352
// registers at head dim 256); 2 -> 256 registers, two workgroups per CU (30 registers spilled at head dim 256)
template <typename T16, int NDB, int WPE>
__global__ __launch_bounds__(256, WPE) void attn_struct_ef_kernel(AttnStructParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  const unsigned int lprev = logit_prev(p.logit_max);   // (in flight behind the kernel: publish_logit_max)
  constexpr int DPW = NDB / 4;                       // output d-blocks per wave
  constexpr int NA = EF_MAXA;                        // arguments (the launcher checks p.nsrl == NA)
  constexpr int PLS = 3 * DP + EF_PLS_PAD;           // floats per argument in the staged language rows
  static_assert(NDB % 4 == 0 && KS % 2 == 0, "head dim 128 or 256");
  extern __shared__ __attribute__((aligned(16))) unsigned char efsm[];
  const int nkb = p.npad_kv >> 5;
  float* pls = reinterpret_cast<float*>(efsm);                       // [NA][3][DP] (+ pad) language Q / K / V rows
  float* us = pls + NA * PLS;                                        // [npad_kv] bias precursors
  float* Bl = us + p.npad_kv;                                        // [NA][npad_kv]: B, then F
  float* mblk = Bl + NA * p.npad_kv;                                 // [nkb][32]
  float* Cl = mblk + nkb * 32;                                       // [16][64]
  float* Dl = Cl + 16 * 64;                                          // [NA][8]
  int* kAl = reinterpret_cast<int*>(Dl + NA * 8);                    // [32]
  int* kBl = kAl + 32;                                               // [8]
  u16x8* El = reinterpret_cast<u16x8*>(kBl + 8);                     // [nkb * 2][64] f16 fragments of E
  float* Ls = reinterpret_cast<float*>(El);                          // [NA][4 waves][64] row-sum partials (after the rounds)
  u16x8* Pl = El + (nkb < 3 ? 3 : nkb) * 2 * 64;                     // [2 blocks][2][NA][64] P^T fragments of one round
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int npb = (p.nppf + 31) >> 5;
  int pair, pb;
  {   // XCD-aware (block b runs on XCD b % 8): the proposal blocks of one (sequence, head) share an L2
    const int b = blockIdx.x, npair = p.S * p.H;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * npb);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); pb = (b >> 3) % npb; }
    else { const int r = b - full * npb; pair = full + r / npb; pb = r % npb; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int pi = pb * 32 + ql;                       // this lane's proposal
  const bool p_ok = pi < p.nppf;
  const int hd = p.H * DP, ldp = 3 * hd;
  const int vid = s / p.nfrm;
  const int lv = p.lpv ? vid : vid / p.ncv;
  const float* plr = p.pl + (int64_t)lv * NA * ldp + h * DP;         // + hd: K block, + 2*hd: V block
  const int64_t kvbase = ((int64_t)s * p.H + h) * (int64_t)p.npad_kv * DP;
  const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + kvbase) + (int64_t)pb * KS * 64 + lane;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.kv + kvbase) + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vv + kvbase) + lane;

  VOG_ATS(0);
  for (int i = tid; i < NA * 3 * (DP / 4); i += 256) {
    const int a = i / (3 * (DP / 4)), r = i - a * 3 * (DP / 4);
    const int which = r / (DP / 4), c = r - which * (DP / 4);
    *reinterpret_cast<float4*>(pls + a * PLS + which * DP + c * 4) =
        *reinterpret_cast<const float4*>(plr + (int64_t)a * ldp + which * hd + c * 4);
  }
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.nppf;
    peb = p.pe_b[h];
    stage_batched<2, 256, float>(p.npad_kv, tid,
        [&](int key) { return p.u[(u_base + (key < p.nppf ? key : 0)) * p.H + h]; },
        [&](int key, float v) { us[key] = key < p.nppf ? v : 0.f; });
    if (p_ok) uq = p.u[(u_base + pi) * p.H + h];
  } else {
    for (int key = tid; key < p.npad_kv; key += 256) us[key] = 0.f;
  }
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;
  // language fragment of k-step ks (lane = argument a, zero past NA), from the staged fp32 rows
  const bool a_ok = ql < NA;
  const float* qlrow = pls + (a_ok ? ql : 0) * PLS + 0 * DP + hi * 8;
  const float* klrow = pls + (a_ok ? ql : 0) * PLS + 1 * DP + hi * 8;
  auto lang_frag = [&](const float* row, int ks) -> u16x8 {
    float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
    if (a_ok) { x0 = *reinterpret_cast<const float4*>(row + ks * 16); x1 = *reinterpret_cast<const float4*>(row + ks * 16 + 4); }
    return u16x8{to16<T16>(x0.x), to16<T16>(x0.y), to16<T16>(x0.z), to16<T16>(x0.w),
                 to16<T16>(x1.x), to16<T16>(x1.y), to16<T16>(x1.z), to16<T16>(x1.w)};
  };

  // ---- phase 1: A and B tiles of this wave's key blocks (kb = wid + 4 i; at most 4: npad_kv <= 512); the scaled logits A stay in `keep`
  f32x16 keep[4];
  {
    // the visual query fragments of the 32 proposals (MFMA B operand: lane = proposal) are staged ONCE per workgroup in
    // LDS (the area of the E / P^T fragments, unused in this phase) instead of once per wave: 48 KB less through the CU's
    // vector memory path per workgroup - this phase runs at the per-CU ingest rate (K: 13 x 16 KB per workgroup)
    u16x8* qf = El;
    for (int i = tid; i < KS * 64; i += 256) qf[i] = Qf[i - lane];      // (Qf already carries + lane)
    __syncthreads();                                 // pls / us / Q in place
    VOG_ATS(1);
    u16x8 lqf[WPE == 1 ? KS : 1];
    if constexpr (WPE == 1) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) lqf[ks] = lang_frag(qlrow, ks);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kb = wid + 4 * i;
      if (kb < nkb) {
        const u16x8* Kb = Kf + (int64_t)kb * KS * 64;
        f32x16 s0, s1, bt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; bt[r] = 0.f; }
        if constexpr (WPE == 1) {
          // registers to spare: the whole K block of 16 fragments is requested at once (one L2 / fabric round trip per block),
          // the language query fragments were formed once per wave (lqf)
          u16x8 kf[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) kf[ks] = Kb[ks * 64];
          f32x16 b1;
#pragma unroll
          for (int r = 0; r < 16; ++r) b1[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ks += 2) {
            s0 = mfma32<T16>(kf[ks], qf[ks * 64 + lane], s0);
            bt = mfma32<T16>(kf[ks], lqf[ks], bt);
            s1 = mfma32<T16>(kf[ks + 1], qf[(ks + 1) * 64 + lane], s1);
            b1 = mfma32<T16>(kf[ks + 1], lqf[ks + 1], b1);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) bt[r] += b1[r];
        } else {
          u16x8 nk0 = Kb[0], nk1 = Kb[64];
#pragma unroll
          for (int ks = 0; ks < KS; ks += 2) {
            const u16x8 k0 = nk0, k1 = nk1;
            if (ks + 2 < KS) { nk0 = Kb[(ks + 2) * 64]; nk1 = Kb[(ks + 3) * 64]; }
            s0 = mfma32<T16>(k0, qf[ks * 64 + lane], s0);
            bt = mfma32<T16>(k0, lang_frag(qlrow, ks), bt);
            s1 = mfma32<T16>(k1, qf[(ks + 1) * 64 + lane], s1);
            bt = mfma32<T16>(k1, lang_frag(qlrow, ks + 1), bt);
          }
        }
        float mb = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + c32_row(r, lane);
          float x = s0[r] + s1[r];
          if (p.use_rel) x += fmaxf(uqp - us[key], 0.f);
          x = key < p.nppf ? x * c2 : -1e30f;
          keep[i][r] = x;
          mb = fmaxf(mb, x);
          if (a_ok) Bl[ql * p.npad_kv + key] = key < p.nppf ? bt[r] * c2 : -1e30f;
        }
        mb = fmaxf(mb, __shfl_xor(mb, 32));
        if (hi == 0) mblk[kb * 32 + ql] = mb;
      }
    }
    if (wid == (nkb & 3)) {
      // the language key block: C^T[a', p] = Kl[a'].Qv[p] and D^T[a', a] = Kl[a'].Ql[a] (rows a' >= NA are zero)
      f32x16 ct, dt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { ct[r] = 0.f; dt[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u16x8 kl = lang_frag(klrow, ks);
        ct = mfma32<T16>(kl, qf[ks * 64 + lane], ct);
        if constexpr (WPE == 1) dt = mfma32<T16>(kl, lqf[ks], dt);
        else dt = mfma32<T16>(kl, lang_frag(qlrow, ks), dt);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Cl[r * 64 + lane] = ct[r];
        const int ak = c32_row(r, lane);
        if (a_ok && ak < 8) Dl[ql * 8 + ak] = dt[r];
      }
    }
  }
  VOG_ATS(2);
  __syncthreads();
  VOG_ATS(3);
  // ---- phase 2: E = exp2(A - row maximum over all key blocks), packed f16, in this wave's registers; F = exp2(B - max) in place
  {
    float m = -1e30f;
    for (int kb = 0; kb < nkb; ++kb) m = fmaxf(m, mblk[kb * 32 + ql]);
    // (run-time logit-scale report, AttnStructParams::logit_max: |row maximum of the proposal part A| - with the argument part's
    // |mB| published below, the larger of the two is a lower bound of the row's largest |A + B| within 2x; log2 units -> nats)
    publish_logit_max(p.logit_max, lprev, p_ok ? fabsf(m) * 0.69314718056f : 0.f, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kb = wid + 4 * i;
      if (kb < nkb) {
        // guard: remember a key at which this proposal's row of A takes its maximum (exact equality: m IS one of these values;
        // several lanes may write - any of them is a witness)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (keep[i][r] == m) kAl[ql] = kb * 32 + c32_row(r, lane);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u16x8 ef;
#pragma unroll
          for (int j = 0; j < 8; ++j) ef[j] = to16<F16>(__builtin_amdgcn_exp2f(keep[i][ks * 8 + j] - m + EF_ESHIFT));
          El[(kb * 2 + ks) * 64 + lane] = ef;
        }
      }
    }
    if (tid < NA * 32) {
      const int a = tid >> 5, sub = tid & 31;
      float* brow = Bl + a * p.npad_kv;
      float mB = -1e30f;
      int kB = 0;
      for (int key = sub; key < p.npad_kv; key += 32)
        if (brow[key] > mB) { mB = brow[key]; kB = key; }
#pragma unroll
      for (int o_ = 16; o_ >= 1; o_ >>= 1) {
        const float om = __shfl_xor(mB, o_);
        const int ok = __shfl_xor(kB, o_);
        if (om > mB || (om == mB && ok < kB)) { mB = om; kB = ok; }
      }
      if (sub == 0) {
        kBl[a] = kB;
        if (p.logit_max) {                       // |mB| of this argument row (raise-only, as publish_logit_max)
          const unsigned int bits = __float_as_uint(fabsf(mB) * 0.69314718056f);
          if (bits > lprev) atomicMax(p.logit_max + (blockIdx.x & (kLogitWords - 1)) * kLogitStride, bits);
        }
      }
      for (int key = sub; key < p.npad_kv; key += 32) brow[key] = __builtin_amdgcn_exp2f(brow[key] - mB);
    }
  }
  __syncthreads();
  VOG_ATS(4);
  // ---- guard. x((a, p), kA[p]) = mA[p] + B[a, kA[p]] is a logit of the row, so its true maximum is at least that and the shift
  // mA[p] + mB[a] sits at most mB[a] - B[a, kA[p]] = -log2 F[a, kA[p]] above it. With E scaled by 2^EF_ESHIFT the fragments keep
  // full precision to 26 binary orders; past EF_GUARD (2 orders of margin) the launch is redone by the per-row kernel
  // (attn_struct_lds_kernel, gated on this flag: normally an empty launch). A sufficient bound, not the exact gap: it can send
  // a launch to the slow kernel needlessly, never the other way round.
  // The same with the roles swapped - x((a, p), kB[a]) = A[p, kB[a]] + mB[a], read back from the parked E = 2^12 exp2(A - mA) -
  // gives a second bound; the row is safe when EITHER witness lies within reach.
  if (p.guard && tid < NA * 32) {
    const int a = tid >> 5, pp = tid & 31;
    if (pb * 32 + pp < p.nppf) {
      const int kb_ = kBl[a], r16 = kb_ & 15;
      const int jj = ((r16 >> 3) << 2) + (r16 & 3), hh = (r16 >> 2) & 1;
      const unsigned short eh = reinterpret_cast<const unsigned short*>(El + ((kb_ >> 5) * 2 + ((kb_ & 31) >> 4)) * 64 + hh * 32 + pp)[jj];
      if (Bl[a * p.npad_kv + kAl[pp]] < EF_GUARD_F && from16<F16>(eh) < EF_GUARD_E)
        __hip_atomic_store(p.guard, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- phase 3: rounds of 2 key blocks
  f32x16 o[NA][DPW];
  float lsum[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    lsum[a] = 0.f;
#pragma unroll
    for (int t = 0; t < DPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[a][t][r] = 0.f;
  }
  const int db0 = wid * DPW;
  // V^T fragments of this wave's d-blocks, PFD key blocks ahead of their use (slot = kb % PFD)
  constexpr int PFD = WPE == 1 ? 4 : 2;
  u16x8 nv[PFD][DPW][2];
#pragma unroll
  for (int d = 0; d < PFD; ++d)
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
      const int kbp = d < nkb ? d : nkb - 1;
      nv[d][t][0] = Vf[((int64_t)kbp * NDB * 2 + (db0 + t) * 2) * 64];
      nv[d][t][1] = Vf[((int64_t)kbp * NDB * 2 + (db0 + t) * 2 + 1) * 64];
    }
  const int nrounds = (nkb + 1) >> 1;
#pragma unroll 2
  for (int rd = 0; rd < nrounds; ++rd) {
    // (a) wave w: the 16-key half (kbl = w >> 1, ks = w & 1) of the round -> NA fragments P_a = E * F_a, row sums in fp32
    {
      const int kbl = wid >> 1, ks = wid & 1;
      const int kbo = 2 * rd + kbl;
      if (kbo < nkb) {
        const u16x8 eh = El[(kbo * 2 + ks) * 64 + lane];
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = from16<F16>(eh[j]);
        // keys of register j: kb*32 + 16*ks + 8*(j>>2) + 4*hi + (j&3) (the P^T fragment order, common.h frag_v)
        const int k0 = kbo * 32 + 16 * ks + 4 * hi;
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const float4 f0 = *reinterpret_cast<const float4*>(Bl + a * p.npad_kv + k0);
          const float4 f1 = *reinterpret_cast<const float4*>(Bl + a * p.npad_kv + k0 + 8);
          const float pr[8] = {e[0] * f0.x, e[1] * f0.y, e[2] * f0.z, e[3] * f0.w, e[4] * f1.x, e[5] * f1.y, e[6] * f1.z, e[7] * f1.w};
          lsum[a] += ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7]));
          Pl[(wid * NA + a) * 64 + lane] =
              u16x8{to16<T16>(pr[0]), to16<T16>(pr[1]), to16<T16>(pr[2]), to16<T16>(pr[3]),
                    to16<T16>(pr[4]), to16<T16>(pr[5]), to16<T16>(pr[6]), to16<T16>(pr[7])};
        }
      }
    }
    __syncthreads();
    // (b) every wave: its d-blocks x the parked fragments of the round's 4 halves
    {
      const int nst = 2 * ((nkb - 2 * rd) < 2 ? (nkb - 2 * rd) : 2);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        if (st >= nst) break;
        const int kbl = st >> 1, ks = st & 1;
        u16x8 pf[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) pf[a] = Pl[(st * NA + a) * 64 + lane];
        // slot of block kb = 2 rd + kbl: kb % PFD (PFD even: the parity of the slot is kbl)
        const int slot = (2 * rd + kbl) % PFD;
#pragma unroll
        for (int sl = 0; sl < PFD; ++sl) {
          if (sl == slot) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
              for (int t = 0; t < DPW; ++t) o[a][t] = mfma32<T16>(nv[sl][t][ks], pf[a], o[a][t]);
            if (ks == 1) {                               // this key block's V^T fragments are consumed: request block kb + PFD
              const int kb = 2 * rd + kbl;
              const int kbn = kb + PFD < nkb ? kb + PFD : nkb - 1;
#pragma unroll
              for (int t = 0; t < DPW; ++t) {
                nv[sl][t][0] = Vf[((int64_t)kbn * NDB * 2 + (db0 + t) * 2) * 64];
                nv[sl][t][1] = Vf[((int64_t)kbn * NDB * 2 + (db0 + t) * 2 + 1) * 64];
              }
            }
          }
        }
      }
    }
    __syncthreads();                                 // the next round overwrites the fragments
  }
  VOG_ATS(5);
  // ---- row sums over all blocks: every wave holds the partial of ITS blocks
#pragma unroll
  for (int a = 0; a < NA; ++a) Ls[(a * 4 + wid) * 64 + lane] = lsum[a];
  __syncthreads();
  // ---- normalise the visual part, add the language keys (their own softmax, normalised before P.V), store
  const int Nq = NA * p.nppf;
  float cacc[16];                                    // C^T[a', p] of this lane (the same for every argument a)
#pragma unroll
  for (int r = 0; r < 16; ++r) cacc[r] = Cl[r * 64 + lane];
  u16x8 vlf[DPW];                                    // language V fragments of this wave's d-blocks: lane = (hi, head column), j = key a'
#pragma unroll
  for (int t = 0; t < DPW; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int key = 8 * (j >> 2) + 4 * hi + (j & 3);
      vlf[t][j] = key < NA ? to16<T16>(pls[key * PLS + 2 * DP + (db0 + t) * 32 + ql]) : (unsigned short)0;
    }
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    float l = (Ls[(a * 4 + 0) * 64 + lane] + Ls[(a * 4 + 1) * 64 + lane]) + (Ls[(a * 4 + 2) * 64 + lane] + Ls[(a * 4 + 3) * 64 + lane]);
    l += __shfl_xor(l, 32);
    const float inv_l = 1.0f / l;
    float y[16];
    float m2 = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ak = c32_row(r, lane);
      y[r] = ak < NA ? (cacc[r] + Dl[a * 8 + (ak & 7)]) * c2 : -1e30f;
      m2 = fmaxf(m2, y[r]);
    }
    m2 = fmaxf(m2, __shfl_xor(m2, 32));
    float l2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { y[r] = __builtin_amdgcn_exp2f(y[r] - m2); l2 += y[r]; }
    l2 += __shfl_xor(l2, 32);
    const float inv_l2 = 1.0f / l2;
    u16x8 plf;
#pragma unroll
    for (int j = 0; j < 8; ++j) plf[j] = to16<T16>(y[j] * inv_l2);
#pragma unroll
    for (int t = 0; t < DPW; ++t) {
      const int db = db0 + t;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[a][t][r] *= inv_l;
      o[a][t] = mfma32<T16>(vlf[t], plf, o[a][t]);
      (void)db;
    }
    // store through LDS (the fragment area is free now; one private slice per wave): a lane of the accumulator layout owns
    // 4 consecutive head columns of ONE query row - written straight, every store instruction scatters 64 pieces of 8
    // bytes over 64 rows 1.5 KB apart (123 MB of output at cfg 4 = 15 M write requests). Transposed through LDS, 8 lanes
    // write the 128 contiguous bytes this wave owns of a row with one 16-byte store each.
    constexpr int OLD = DPW * 32 + 4;                  // u16 per staged row (+ 8 bytes: rows 16 banks apart... 34 dwords)
    unsigned short* ost = reinterpret_cast<unsigned short*>(Pl) + (size_t)wid * 32 * OLD;
#pragma unroll
    for (int t = 0; t < DPW; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<u16x4*>(ost + ql * OLD + t * 32 + g * 8 + hi * 4) =
            u16x4{to16<T16>(o[a][t][g * 4 + 0]), to16<T16>(o[a][t][g * 4 + 1]), to16<T16>(o[a][t][g * 4 + 2]), to16<T16>(o[a][t][g * 4 + 3])};
    constexpr int CPR = DPW * 32 / 8;                  // 16-byte chunks per row
#pragma unroll
    for (int it = 0; it < (32 * CPR) / 64; ++it) {
      const int idx = it * 64 + lane, q = idx / CPR, ch = idx - q * CPR;
      const u16x4 lo = *reinterpret_cast<const u16x4*>(ost + q * OLD + ch * 8);
      const u16x4 hi4 = *reinterpret_cast<const u16x4*>(ost + q * OLD + ch * 8 + 4);
      const int pq = pb * 32 + q;
      if (pq < p.nppf) {
        unsigned short* orow = p.out + ((int64_t)s * Nq + (int64_t)a * p.nppf + pq) * ((int64_t)p.H * DP) + (int64_t)h * DP + db0 * 32 + ch * 8;
        // (non-temporal: 123 MB of output at cfg 4 that the next kernel reads from memory anyway - it must not displace
        // the K / V^T fragments the 13 proposal blocks of a (sequence, head) re-read from L2)
        __builtin_nontemporal_store(u16x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]}, reinterpret_cast<u16x8*>(orow));
      }
    }
  }
  VOG_ATS(6);
}

}  // namespace vog
