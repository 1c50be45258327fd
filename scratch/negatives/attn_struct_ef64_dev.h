// E x F separable attention (attn_struct_ef_dev.h) with 64 proposals per workgroup: head dim 256, 8 waves.
//
// attn_struct_ef_kernel runs at the per-CU ingest rate: every workgroup (32 proposals) pulls all K and V^T fragments of its
// (sequence, head) through its CU's vector memory path - 13 x 16 KB each at p100, 575 KB per workgroup with Q and the outputs.
// With TWO query tiles per workgroup the same K / V^T stream serves 64 proposals: a K fragment feeds two S^T tiles, a V^T
// fragment ten P.V products (5 arguments x 2 tiles) instead of five - half the bytes per query. 8 waves:
//   phase 1  the waves split the key blocks (kb = wave, wave + 8): A tiles of both query tiles + the B tile per block
//   phase 2  E -> LDS (f16), F in place
//   phase 3  rounds of 2 key blocks = 8 (block, 16-key half, query tile) pieces, one per wave: P_a = E * F_a -> LDS; then
//            wave w owns output d-block w: one V^T fragment x the 10 parked fragments of its (block, half)
// One workgroup per CU (130 KB of LDS), two waves per SIMD, <= 256 registers.
#pragma once
#include "attn_struct_ef_dev.h"

namespace vog {

static inline size_t attn_struct_ef64_lds(int npad_kv) {
  const int nkb = npad_kv >> 5;
  return (size_t)EF_MAXA * (3 * 256 + EF_PLS_PAD) * 4   // pls
         + (size_t)npad_kv * 4                          // us
         + (size_t)EF_MAXA * npad_kv * 4                // Bl / F
         + (size_t)nkb * 64 * 4                         // block maxima [kb][tile][32]
         + (size_t)2 * 16 * 64 * 4                      // C tiles
         + (size_t)EF_MAXA * 8 * 4                      // D
         + (size_t)(nkb < 4 ? 4 : nkb) * 4 * 64 * 16    // E fragments [kb][tile][ks][64] (first: the Q staging; last: row sums)
         + (size_t)8 * EF_MAXA * 64 * 16;               // P^T fragments of one round [8 pieces][a][64]; last: output staging
}

template <typename T16>
__global__ __launch_bounds__(512, 2) void attn_struct_ef64_kernel(AttnStructParams p) {
  constexpr int NDB = 8, DP = 256, KS = 16, NA = EF_MAXA, PLS = 3 * DP + EF_PLS_PAD;
  extern __shared__ __attribute__((aligned(16))) unsigned char ef64sm[];
  const int nkb = p.npad_kv >> 5;
  float* pls = reinterpret_cast<float*>(ef64sm);                     // [NA][3][DP] (+ pad)
  float* us = pls + NA * PLS;                                        // [npad_kv]
  float* Bl = us + p.npad_kv;                                        // [NA][npad_kv]: B, then F
  float* mblk = Bl + NA * p.npad_kv;                                 // [nkb][2][32]
  float* Cl = mblk + nkb * 64;                                       // [2][16][64]
  float* Dl = Cl + 2 * 16 * 64;                                      // [NA][8]
  u16x8* El = reinterpret_cast<u16x8*>(Dl + NA * 8);                 // [nkb][2 tiles][2 ks][64] f16 fragments of E
  float* Ls = reinterpret_cast<float*>(El);                          // [NA][8 waves][64] row-sum partials (after the rounds)
  u16x8* Pl = El + (nkb < 4 ? 4 : nkb) * 4 * 64;                     // [8 pieces][NA][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int npb = (p.nppf + 63) >> 6;
  int pair, pb;
  {   // XCD-aware (block b runs on XCD b % 8): the proposal blocks of one (sequence, head) share an L2
    const int b = blockIdx.x, npair = p.S * p.H;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * npb);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); pb = (b >> 3) % npb; }
    else { const int r = b - full * npb; pair = full + r / npb; pb = r % npb; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const bool t1_ok = pb * 2 + 1 < nkb;               // the second query tile exists (its Q fragments lie inside the padded buffer)
  const int hd = p.H * DP, ldp = 3 * hd;
  const int vid = s / p.nfrm;
  const int lv = p.lpv ? vid : vid / p.ncv;
  const float* plr = p.pl + (int64_t)lv * NA * ldp + h * DP;
  const int64_t kvbase = ((int64_t)s * p.H + h) * (int64_t)p.npad_kv * DP;
  const u16x8* Qg = reinterpret_cast<const u16x8*>(p.q + kvbase) + (int64_t)pb * 2 * KS * 64;
  const u16x8* Kf = reinterpret_cast<const u16x8*>(p.kv + kvbase) + lane;
  const u16x8* Vf = reinterpret_cast<const u16x8*>(p.vv + kvbase) + lane;

  for (int i = tid; i < NA * 3 * (DP / 4); i += 512) {
    const int a = i / (3 * (DP / 4)), r = i - a * 3 * (DP / 4);
    const int which = r / (DP / 4), c = r - which * (DP / 4);
    *reinterpret_cast<float4*>(pls + a * PLS + which * DP + c * 4) =
        *reinterpret_cast<const float4*>(plr + (int64_t)a * ldp + which * hd + c * 4);
  }
  float uq[2] = {0.f, 0.f}, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.nppf;
    peb = p.pe_b[h];
    for (int key = tid; key < p.npad_kv; key += 512)
      us[key] = key < p.nppf ? p.u[(u_base + key) * p.H + h] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int pi = pb * 64 + t * 32 + ql;
      if (pi < p.nppf) uq[t] = p.u[(u_base + pi) * p.H + h];
    }
  } else {
    for (int key = tid; key < p.npad_kv; key += 512) us[key] = 0.f;
  }
  // visual query fragments of both tiles, staged once per workgroup (the E area is unused until phase 2)
  u16x8* qs = El;                                                    // [2][KS][64]
  for (int i = tid; i < 2 * KS * 64; i += 512) {
    u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    qs[i] = (i < KS * 64 || t1_ok) ? Qg[i] : z;
  }
  const float c2 = p.inv_scale * 1.44269504088896340736f;
  const bool a_ok = ql < NA;
  const float* qlrow = pls + (a_ok ? ql : 0) * PLS + 0 * DP + hi * 8;
  const float* klrow = pls + (a_ok ? ql : 0) * PLS + 1 * DP + hi * 8;
  auto lang_frag = [&](const float* row, int ks) -> u16x8 {
    float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
    if (a_ok) { x0 = *reinterpret_cast<const float4*>(row + ks * 16); x1 = *reinterpret_cast<const float4*>(row + ks * 16 + 4); }
    return u16x8{to16<T16>(x0.x), to16<T16>(x0.y), to16<T16>(x0.z), to16<T16>(x0.w),
                 to16<T16>(x1.x), to16<T16>(x1.y), to16<T16>(x1.z), to16<T16>(x1.w)};
  };
  __syncthreads();                                   // pls / us / Q in place

  // ---- phase 1: this wave's key blocks (kb = wid, wid + 8): A tiles of both query tiles (kept), B tile, block maxima
  f32x16 keep[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kb = wid + 8 * i;
    if (kb < nkb) {
      const u16x8* Kb = Kf + (int64_t)kb * KS * 64;
      f32x16 s0[2], s1[2], bt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[0][r] = 0.f; s0[1][r] = 0.f; s1[0][r] = 0.f; s1[1][r] = 0.f; bt[r] = 0.f; }
      u16x8 nk0 = Kb[0], nk1 = Kb[64];
#pragma unroll
      for (int ks = 0; ks < KS; ks += 2) {
        const u16x8 k0 = nk0, k1 = nk1;
        if (ks + 2 < KS) { nk0 = Kb[(ks + 2) * 64]; nk1 = Kb[(ks + 3) * 64]; }
        s0[0] = mfma32<T16>(k0, qs[(0 * KS + ks) * 64 + lane], s0[0]);
        s0[1] = mfma32<T16>(k0, qs[(1 * KS + ks) * 64 + lane], s0[1]);
        bt = mfma32<T16>(k0, lang_frag(qlrow, ks), bt);
        s1[0] = mfma32<T16>(k1, qs[(0 * KS + ks + 1) * 64 + lane], s1[0]);
        s1[1] = mfma32<T16>(k1, qs[(1 * KS + ks + 1) * 64 + lane], s1[1]);
        bt = mfma32<T16>(k1, lang_frag(qlrow, ks + 1), bt);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float uqp = uq[t] + peb;
        float mb = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + c32_row(r, lane);
          float x = s0[t][r] + s1[t][r];
          if (p.use_rel) x += fmaxf(uqp - us[key], 0.f);
          x = key < p.nppf ? x * c2 : -1e30f;
          keep[i][t][r] = x;
          mb = fmaxf(mb, x);
        }
        mb = fmaxf(mb, __shfl_xor(mb, 32));
        if (hi == 0) mblk[(kb * 2 + t) * 32 + ql] = mb;
      }
      if (a_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + c32_row(r, lane);
          Bl[ql * p.npad_kv + key] = key < p.nppf ? bt[r] * c2 : -1e30f;
        }
      }
    }
  }
  if (wid == (nkb & 7)) {
    // the language key block: C^T[a', p] of both tiles and D^T[a', a]
    f32x16 ct[2], dt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ct[0][r] = 0.f; ct[1][r] = 0.f; dt[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u16x8 kl = lang_frag(klrow, ks);
      ct[0] = mfma32<T16>(kl, qs[(0 * KS + ks) * 64 + lane], ct[0]);
      ct[1] = mfma32<T16>(kl, qs[(1 * KS + ks) * 64 + lane], ct[1]);
      dt = mfma32<T16>(kl, lang_frag(qlrow, ks), dt);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Cl[(0 * 16 + r) * 64 + lane] = ct[0][r];
      Cl[(1 * 16 + r) * 64 + lane] = ct[1][r];
      const int ak = c32_row(r, lane);
      if (a_ok && ak < 8) Dl[ql * 8 + ak] = dt[r];
    }
  }
  __syncthreads();
  // ---- phase 2: E = exp2(A - row maximum over all key blocks) -> LDS (over the Q staging: every wave is past phase 1); F in place
  {
    float m[2] = {-1e30f, -1e30f};
    for (int kb = 0; kb < nkb; ++kb) { m[0] = fmaxf(m[0], mblk[(kb * 2 + 0) * 32 + ql]); m[1] = fmaxf(m[1], mblk[(kb * 2 + 1) * 32 + ql]); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kb = wid + 8 * i;
      if (kb < nkb) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            u16x8 ef;
#pragma unroll
            for (int j = 0; j < 8; ++j) ef[j] = to16<F16>(__builtin_amdgcn_exp2f(keep[i][t][ks * 8 + j] - m[t]));
            El[((kb * 2 + t) * 2 + ks) * 64 + lane] = ef;
          }
      }
    }
    if (tid < NA * 32) {
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
  // ---- phase 3: rounds of 2 key blocks; this wave produces piece wid = (kbl, ks, tile) and owns output d-block wid
  f32x16 o[NA][2];
  float lsum[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    lsum[a] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[a][t][r] = 0.f;
  }
  const int db = wid;
  const int pt = wid & 1, pks = (wid >> 1) & 1, pkbl = wid >> 2;     // the piece this wave produces in every round
  u16x8 nv[2][2];                                                    // V^T fragments of d-block db: [slot = kb & 1][ks]
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int kbp = d < nkb ? d : nkb - 1;
    nv[d][0] = Vf[((int64_t)kbp * NDB * 2 + db * 2) * 64];
    nv[d][1] = Vf[((int64_t)kbp * NDB * 2 + db * 2 + 1) * 64];
  }
  const int nrounds = (nkb + 1) >> 1;
  for (int rd = 0; rd < nrounds; ++rd) {
    {
      const int kbo = 2 * rd + pkbl;
      if (kbo < nkb) {
        const u16x8 eh = El[((kbo * 2 + pt) * 2 + pks) * 64 + lane];
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = from16<F16>(eh[j]);
        const int k0 = kbo * 32 + 16 * pks + 4 * hi;
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
#pragma unroll
    for (int kbl = 0; kbl < 2; ++kbl) {
      const int kb = 2 * rd + kbl;
      if (kb < nkb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            u16x8 pf[NA];
#pragma unroll
            for (int a = 0; a < NA; ++a) pf[a] = Pl[(((kbl * 2 + ks) * 2 + t) * NA + a) * 64 + lane];
#pragma unroll
            for (int a = 0; a < NA; ++a) o[a][t] = mfma32<T16>(nv[kbl][ks], pf[a], o[a][t]);
          }
        }
        const int kbn = kb + 2 < nkb ? kb + 2 : nkb - 1;             // this block's fragments are consumed: request block kb + 2
        nv[kbl][0] = Vf[((int64_t)kbn * NDB * 2 + db * 2) * 64];
        nv[kbl][1] = Vf[((int64_t)kbn * NDB * 2 + db * 2 + 1) * 64];
      }
    }
    __syncthreads();
  }
  // ---- row sums: wave w holds the partial of the pieces it produced (tile w & 1)
#pragma unroll
  for (int a = 0; a < NA; ++a) Ls[(a * 8 + wid) * 64 + lane] = lsum[a];
  __syncthreads();
  const int Nq = NA * p.nppf;
  u16x8 vlf;                                         // language V fragment of d-block db
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int key = 8 * (j >> 2) + 4 * hi + (j & 3);
    vlf[j] = key < NA ? to16<T16>(pls[key * PLS + 2 * DP + db * 32 + ql]) : (unsigned short)0;
  }
  constexpr int OLD = 32 + 4;                        // u16 per staged row
  unsigned short* ost = reinterpret_cast<unsigned short*>(Pl) + (size_t)wid * 64 * OLD;    // [2 tiles x 32 rows][OLD]
#pragma unroll
  for (int a = 0; a < NA; ++a) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float l = (Ls[(a * 8 + t) * 64 + lane] + Ls[(a * 8 + 2 + t) * 64 + lane]) + (Ls[(a * 8 + 4 + t) * 64 + lane] + Ls[(a * 8 + 6 + t) * 64 + lane]);
      l += __shfl_xor(l, 32);
      const float inv_l = 1.0f / l;
      float y[16];
      float m2 = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ak = c32_row(r, lane);
        y[r] = ak < NA ? (Cl[(t * 16 + r) * 64 + lane] + Dl[a * 8 + (ak & 7)]) * c2 : -1e30f;
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
      for (int r = 0; r < 16; ++r) o[a][t][r] *= inv_l;
      o[a][t] = mfma32<T16>(vlf, plf, o[a][t]);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<u16x4*>(ost + (t * 32 + ql) * OLD + g * 8 + hi * 4) =
            u16x4{to16<T16>(o[a][t][g * 4 + 0]), to16<T16>(o[a][t][g * 4 + 1]), to16<T16>(o[a][t][g * 4 + 2]), to16<T16>(o[a][t][g * 4 + 3])};
    }
    // 64 rows x 64 bytes of this wave's d-block: 4 lanes write one row's 64 contiguous bytes with 16-byte stores
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 64 + lane, q = idx >> 2, ch = idx & 3;
      const u16x4 lo = *reinterpret_cast<const u16x4*>(ost + q * OLD + ch * 8);
      const u16x4 hi4 = *reinterpret_cast<const u16x4*>(ost + q * OLD + ch * 8 + 4);
      const int pq = pb * 64 + q;
      if (pq < p.nppf) {
        unsigned short* orow = p.out + ((int64_t)s * Nq + (int64_t)a * p.nppf + pq) * ((int64_t)p.H * DP) + (int64_t)h * DP + db * 32 + ch * 8;
        __builtin_nontemporal_store(u16x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]}, reinterpret_cast<u16x8*>(orow));
      }
    }
  }
}

}  // namespace vog
