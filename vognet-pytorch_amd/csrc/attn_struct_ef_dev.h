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
// One workgroup = one (sequence, head, block of 32 proposals) = 32 * nsrl queries; 4 waves.
//   phase 1  the waves split the KEY blocks: A (kept in registers) and B tiles with K fragments straight from L2
//            (every K fragment is read by exactly one wave), block maxima / B / C / D to LDS
//   phase 2  E = exp2(A - max over all blocks) -> LDS, f16, in MFMA B-operand (P^T) order; F = exp2(B - max) in place
//   phase 3  the waves split the OUTPUT d-blocks: per key block the V^T fragments of the wave's d-blocks come straight
//            from L2 ONCE (every V^T fragment is read by exactly one wave) and serve all nsrl arguments:
//            P_a = E * F_a (fp32, row sums exact) -> 16 bit -> nsrl x DPW accumulators
// <= 256 registers, ~60 KB of LDS: two workgroups per CU.
#pragma once
#include "attention_dev.h"

namespace vog {

constexpr int EF_MAXA = 5;          // arguments per query set (cfg.misc.srl_arg_length)

template <int NDB>
static inline size_t attn_struct_ef_lds(int nsrl, int npad_kv) {
  const int nkb = npad_kv >> 5;
  return (size_t)nsrl * 3 * NDB * 32 * 4          // pls
         + (size_t)npad_kv * 4                    // us
         + (size_t)EF_MAXA * npad_kv * 4          // Bl / F
         + (size_t)nkb * 32 * 4                   // block maxima
         + (size_t)16 * 64 * 4                    // C tile (accumulator layout)
         + (size_t)EF_MAXA * 8 * 4                // D
         + (size_t)nkb * 2 * 64 * 16;             // E fragments (f16)
}

template <typename T16, int NDB>
__global__ __launch_bounds__(256, 2) void attn_struct_ef_kernel(AttnStructParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  constexpr int DPW = NDB / 4;                       // output d-blocks per wave
  static_assert(NDB % 4 == 0 && KS % 2 == 0, "head dim 128 or 256");
  extern __shared__ __attribute__((aligned(16))) unsigned char efsm[];
  const int nkb = p.npad_kv >> 5;
  float* pls = reinterpret_cast<float*>(efsm);                       // [nsrl][3][DP] language Q / K / V rows
  float* us = pls + p.nsrl * 3 * DP;                                 // [npad_kv] bias precursors
  float* Bl = us + p.npad_kv;                                        // [EF_MAXA][npad_kv]: B, then F
  float* mblk = Bl + EF_MAXA * p.npad_kv;                            // [nkb][32]
  float* Cl = mblk + nkb * 32;                                       // [16][64]
  float* Dl = Cl + 16 * 64;                                          // [EF_MAXA][8]
  u16x8* El = reinterpret_cast<u16x8*>(Dl + EF_MAXA * 8);            // [nkb * 2][64] f16 fragments
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
  const float* plr = p.pl + (int64_t)lv * p.nsrl * ldp + h * DP;     // + hd: K block, + 2*hd: V block
  const int64_t kvbase = ((int64_t)s * p.H + h) * (int64_t)p.npad_kv * DP;
  const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + kvbase) + (int64_t)pb * KS * 64 + lane;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.kv + kvbase) + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vv + kvbase) + lane;

  for (int i = tid; i < p.nsrl * 3 * (DP / 4); i += 256) {
    const int a = i / (3 * (DP / 4)), r = i - a * 3 * (DP / 4);
    const int which = r / (DP / 4), c = r - which * (DP / 4);
    reinterpret_cast<float4*>(pls)[i] = *reinterpret_cast<const float4*>(plr + (int64_t)a * ldp + which * hd + c * 4);
  }
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.nppf;
    peb = p.pe_b[h];
    for (int key = tid; key < p.npad_kv; key += 256)
      us[key] = key < p.nppf ? p.u[(u_base + key) * p.H + h] : 0.f;
    if (p_ok) uq = p.u[(u_base + pi) * p.H + h];
  } else {
    for (int key = tid; key < p.npad_kv; key += 256) us[key] = 0.f;
  }
  // the visual query fragments of the 32 proposals (MFMA B operand: lane = proposal)
  u16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = Qf[ks * 64];
  __syncthreads();                                   // pls / us in place
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;
  // language query fragment of k-step ks (B operand: lane = argument a, zero past nsrl), from the staged fp32 rows
  const bool a_ok = ql < p.nsrl;
  const float* qlrow = pls + ((a_ok ? ql : 0) * 3 + 0) * DP + hi * 8;
  const float* klrow = pls + ((a_ok ? ql : 0) * 3 + 1) * DP + hi * 8;
  auto lang_frag = [&](const float* row, int ks) -> u16x8 {
    float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
    if (a_ok) { x0 = *reinterpret_cast<const float4*>(row + ks * 16); x1 = *reinterpret_cast<const float4*>(row + ks * 16 + 4); }
    return u16x8{to16<T16>(x0.x), to16<T16>(x0.y), to16<T16>(x0.z), to16<T16>(x0.w),
                 to16<T16>(x1.x), to16<T16>(x1.y), to16<T16>(x1.z), to16<T16>(x1.w)};
  };

  // ---- phase 1: A and B tiles of this wave's key blocks (kb = wid, wid + 4, ...; at most 4: npad_kv <= 512)
  f32x16 keep[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kb = wid + 4 * i;
    if (kb < nkb) {
      const u16x8* Kb = Kf + (int64_t)kb * KS * 64;
      f32x16 s0, s1, bt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; bt[r] = 0.f; }
      u16x8 nk0 = Kb[0], nk1 = Kb[64];
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        const u16x8 k0 = nk0, k1 = nk1;
        if (ks + 2 < KS) { nk0 = Kb[(ks + 2) * 64]; nk1 = Kb[(ks + 3) * 64]; }
        s0 = mfma32<T16>(k0, qf[ks], s0);
        bt = mfma32<T16>(k0, lang_frag(qlrow, ks), bt);
        s1 = mfma32<T16>(k1, qf[ks + 1], s1);
        bt = mfma32<T16>(k1, lang_frag(qlrow, ks + 1), bt);
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
    // the language key block: C^T[a', p] = Kl[a'].Qv[p] and D^T[a', a] = Kl[a'].Ql[a] (rows a' >= nsrl are zero)
    f32x16 ct, dt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ct[r] = 0.f; dt[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u16x8 kl = lang_frag(klrow, ks);
      ct = mfma32<T16>(kl, qf[ks], ct);
      dt = mfma32<T16>(kl, lang_frag(qlrow, ks), dt);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Cl[r * 64 + lane] = ct[r];
      const int ak = c32_row(r, lane);
      if (a_ok && ak < 8) Dl[ql * 8 + ak] = dt[r];
    }
  }
  __syncthreads();
  // ---- phase 2: E = exp2(A - row maximum over all key blocks) -> LDS (f16, P^T fragment order); F = exp2(B - max) in place
  {
    float m = -1e30f;
    for (int kb = 0; kb < nkb; ++kb) m = fmaxf(m, mblk[kb * 32 + ql]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kb = wid + 4 * i;
      if (kb < nkb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u16x8 ef;
#pragma unroll
          for (int j = 0; j < 8; ++j) ef[j] = to16<F16>(__builtin_amdgcn_exp2f(keep[i][ks * 8 + j] - m));
          El[(kb * 2 + ks) * 64 + lane] = ef;
        }
      }
    }
    if (tid < p.nsrl * 32) {
      const int a = tid >> 5, sub = tid & 31;
      float* brow = Bl + a * p.npad_kv;
      float mB = -1e30f;
      for (int key = sub; key < p.npad_kv; key += 32) mB = fmaxf(mB, brow[key]);
#pragma unroll
      for (int o_ = 16; o_ >= 1; o_ >>= 1) mB = fmaxf(mB, __shfl_xor(mB, o_));
      for (int key = sub; key < p.npad_kv; key += 32) brow[key] = __builtin_amdgcn_exp2f(brow[key] - mB);
    }
  }
  __syncthreads();
  // ---- phase 3: output d-blocks of this wave (db = wid * DPW + t), all arguments
  f32x16 o[EF_MAXA][DPW];
  float lsum[EF_MAXA];
#pragma unroll
  for (int a = 0; a < EF_MAXA; ++a) {
    lsum[a] = 0.f;
#pragma unroll
    for (int t = 0; t < DPW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[a][t][r] = 0.f;
  }
  const int db0 = wid * DPW;
  u16x8 nv[DPW][2];
#pragma unroll
  for (int t = 0; t < DPW; ++t) { nv[t][0] = Vf[(int64_t)((db0 + t) * 2) * 64]; nv[t][1] = Vf[(int64_t)((db0 + t) * 2 + 1) * 64]; }
  for (int kb = 0; kb < nkb; ++kb) {
    u16x8 vf[DPW][2];
#pragma unroll
    for (int t = 0; t < DPW; ++t) { vf[t][0] = nv[t][0]; vf[t][1] = nv[t][1]; }
    if (kb + 1 < nkb) {
#pragma unroll
      for (int t = 0; t < DPW; ++t) {
        nv[t][0] = Vf[((int64_t)(kb + 1) * NDB * 2 + (db0 + t) * 2) * 64];
        nv[t][1] = Vf[((int64_t)(kb + 1) * NDB * 2 + (db0 + t) * 2 + 1) * 64];
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u16x8 eh = El[(kb * 2 + ks) * 64 + lane];
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = from16<F16>(eh[j]);
      // keys of register j: kb*32 + 16*ks + 8*(j>>2) + 4*hi + (j&3) (the P^T fragment order, common.h frag_v)
      const int k0 = kb * 32 + 16 * ks + 4 * hi;
#pragma unroll
      for (int a = 0; a < EF_MAXA; ++a) {
        if (a < p.nsrl) {
          const float4 f0 = *reinterpret_cast<const float4*>(Bl + a * p.npad_kv + k0);
          const float4 f1 = *reinterpret_cast<const float4*>(Bl + a * p.npad_kv + k0 + 8);
          const float pr[8] = {e[0] * f0.x, e[1] * f0.y, e[2] * f0.z, e[3] * f0.w, e[4] * f1.x, e[5] * f1.y, e[6] * f1.z, e[7] * f1.w};
          lsum[a] += ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7]));
          const u16x8 pf = u16x8{to16<T16>(pr[0]), to16<T16>(pr[1]), to16<T16>(pr[2]), to16<T16>(pr[3]),
                                 to16<T16>(pr[4]), to16<T16>(pr[5]), to16<T16>(pr[6]), to16<T16>(pr[7])};
#pragma unroll
          for (int t = 0; t < DPW; ++t) o[a][t] = mfma32<T16>(vf[t][ks], pf, o[a][t]);
        }
      }
    }
  }
  // ---- normalise the visual part, add the language keys (their own softmax, normalised before P.V), store
  const int Nq = p.nsrl * p.nppf;
#pragma unroll
  for (int a = 0; a < EF_MAXA; ++a) {
    if (a < p.nsrl) {
      const float l = lsum[a] + __shfl_xor(lsum[a], 32);
      const float inv_l = 1.0f / l;
      float y[16];
      float m2 = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ak = c32_row(r, lane);
        y[r] = ak < p.nsrl ? (Cl[r * 64 + lane] + Dl[a * 8 + (ak & 7)]) * c2 : -1e30f;
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
        u16x8 vl;                                    // language V fragment: lane = (hi, head column), register j = key a'
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = 8 * (j >> 2) + 4 * hi + (j & 3);
          vl[j] = key < p.nsrl ? to16<T16>(pls[(key * 3 + 2) * DP + db * 32 + ql]) : (unsigned short)0;
        }
        o[a][t] = mfma32<T16>(vl, plf, o[a][t]);
        if (p_ok) struct_store<T16>(p, o[a][t], db, (int64_t)s * Nq + (int64_t)a * p.nppf + pi, h, DP, hi);
      }
    }
  }
}

}  // namespace vog
