// Separable mul_tx layer-0 attention, general form, with the visual K / V^T fragments of the
// (sequence, head) going through an LDS RING shared by the workgroup (p100: 100 visual keys per frame
// for temp / sep, 400 for spat = 4 / 13 key blocks of 32 KB at head dim 256).
//
// attn_struct_kernel (attention_dev.h) lets every wave load the K and V^T fragments of every key block
// straight from L2: at cfg 4 that is 13 x 32 KB per wave, 1.6 MB per workgroup through one CU's vector
// memory path (~45 GB/s in practice) plus two exposed L2 round trips per key block - 456 us for the
// launch with ~100 GFLOP of real matrix work. Here the 4 waves of a workgroup (128 queries) bring each
// key block into LDS ONCE by LDS-DMA, two blocks ahead in a ring of four (counted vmcnt waits + one
// barrier per block), and read their MFMA A operands from there (lds_read128: invisible to hipcc's
// waitcnt pass, which would otherwise wait for ALL the DMA in front of the first LDS read).
// Everything else - queries formed from Qv[p] + Ql[a], online softmax over the visual blocks, the
// language block with its own softmax, the output layout - is attn_struct_kernel's (same helpers);
// the two share the tests.
#pragma once
#include "attention_dev.h"

namespace vog {

template <typename T16, int NDB>
__global__ __launch_bounds__(256, 1) void attn_struct_lds_kernel(AttnStructParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  constexpr int NF = KS + 2 * NDB;                   // KiB fragments per key block (K then V^T)
  constexpr int FPW = NF / 4;                        // DMA instructions per wave per block
  constexpr int PF = 3;                              // LDS fragments requested ahead of their MFMA
  static_assert(NF % 4 == 0, "fragments per key block must divide over the 4 waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char slsm[];
  if (p.guard_gate && *reinterpret_cast<volatile const int*>(p.guard) == 0) return;   // fallback pass of attn_struct_ef_kernel: not needed
  constexpr int NBUF = 4, DIST = 2;                  // ring depth; blocks requested ahead
  const int nkb = p.npad_kv >> 5;
  unsigned char* kv = slsm;                          // [NBUF][NF][1024]
  float* us = reinterpret_cast<float*>(slsm + (size_t)NBUF * NF * 1024);   // [npad_kv] bias precursors
  float* pls = us + p.npad_kv;                       // [nsrl][3][DP] language Q / K / V rows of this (video, head)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  const int Nq = p.nsrl * p.nppf;
  const int nqb = (Nq + 31) >> 5, nqg = (nqb + 3) >> 2;
  int pair, qg;
  {   // XCD-aware (block b runs on XCD b % 8): the query groups of one (sequence, head) share an L2
    const int b = blockIdx.x, npair = p.S * p.H;
    const int full = (npair / 8) * 8;
    const int grp = b / (8 * nqg);
    if (grp * 8 < full) { pair = grp * 8 + (b & 7); qg = (b >> 3) % nqg; }
    else { const int r = b - full * nqg; pair = full + r / nqg; qg = r % nqg; }
  }
  const int s = pair / p.H, h = pair - s * p.H;
  const int qb = qg * 4 + wid;
  const bool wave_ok = qb < nqb;                     // a wave past the end still does its share of the DMA
  const int qi = qb * 32 + ql;
  const bool q_ok = wave_ok && qi < Nq;
  const int hd = p.H * DP, ldp = 3 * hd;
  const int vid = s / p.nfrm;
  const int lv = p.lpv ? vid : vid / p.ncv;
  const float* plr = p.pl + (int64_t)lv * p.nsrl * ldp + h * DP;     // + hd: K block, + 2*hd: V block
  const int64_t kvbase = ((int64_t)s * p.H + h) * (int64_t)p.npad_kv * DP;

  // the language rows of this (video, head) - Ql / Kl / Vl, nsrl x 3 x DP floats - are staged in LDS once
  // per workgroup: formed per wave from global memory, the query fragments and the language key block
  // cost ~100 small dependent loads per wave (95 + ~30 of the 360 us of this launch at cfg 4)
  stage_batched<4, 256, float4>(p.nsrl * 3 * (DP / 4), tid,
      [&](int i) {
        const int a = i / (3 * (DP / 4)), r = i - a * 3 * (DP / 4);
        const int which = r / (DP / 4), c = r - which * (DP / 4);
        return *reinterpret_cast<const float4*>(plr + (int64_t)a * ldp + which * hd + c * 4);
      },
      [&](int i, const float4& v) { reinterpret_cast<float4*>(pls)[i] = v; });
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.nppf;
    peb = p.pe_b[h];
    stage_batched<2, 256, float>(p.npad_kv, tid,
        [&](int key) { return p.u[(u_base + (key < p.nppf ? key : 0)) * p.H + h]; },
        [&](int key, float v) { us[key] = key < p.nppf ? v : 0.f; });
    if (q_ok) uq = p.u[(u_base + (qi % p.nppf)) * p.H + h];
  } else {
    for (int key = tid; key < p.npad_kv; key += 256) us[key] = 0.f;
  }
  __syncthreads();                                   // pls / us in place
  // queries: q(a, p) = Qv[p] + Ql[a] (visual fragment chunk from global memory, language row from LDS)
  u16x8 qf[KS];
  if (p.q_visual) {
    const int t = (wave_ok ? qb : 0) * 32 + ql;
    int a = t / p.nppf;
    const int pp = t - a * p.nppf;
    a = a < p.nsrl ? a : p.nsrl - 1;                 // tokens past the end are never stored
    const unsigned short* qv = p.q + kvbase;
    const float* qlr = pls + (a * 3 + 0) * DP + hi * 8;
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
    struct_load_q<T16, KS>(p, qf, s, h, wave_ok ? qb : 0, lane, plr, ldp, kvbase);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // fragment f of key block kb (f < KS: K, else V^T) -> ring slot kb % NBUF; FPW per wave per block. Blocks
  // past the end re-load the last one (keeps the vmcnt bookkeeping uniform; never consumed).
  const unsigned short* Kg = p.kv + kvbase;
  const unsigned short* Vg = p.vv + kvbase;
  auto issue = [&](int kb) __attribute__((always_inline)) {
    const int buf = kb % NBUF;
    const int src_kb = kb < nkb ? kb : nkb - 1;
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      const int f = wid * FPW + i;
      const unsigned short* src = f < KS ? Kg + ((int64_t)src_kb * KS + f) * 512
                                         : Vg + ((int64_t)src_kb * NDB * 2 + (f - KS)) * 512;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src + lane * 8),
          (__attribute__((address_space(3))) void*)(kv + ((size_t)buf * NF + f) * 1024), 16, 0, 0);
    }
  };
  issue(0); issue(1);
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;

  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  for (int kb = 0; kb < ((p.dbg & 4) ? 0 : nkb); ++kb) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FPW * (DIST - 1)) : "memory");   // block kb landed (kb+1 may be in flight)
    __builtin_amdgcn_s_barrier();                    // ... everybody's share of it (and us[] for kb = 0); block kb-1 is consumed
    asm volatile("" ::: "memory");
    issue(kb + DIST);
    if (!wave_ok) continue;
    const unsigned char* kblk = kv + ((size_t)(kb % NBUF) * NF) * 1024 + lane * 16;
    f32x16 s0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = 0.f;
    {
      u16x8 fr[PF];
#pragma unroll
      for (int j = 0; j < PF; ++j) fr[j] = lds_read128(kblk + (j < KS ? j : KS - 1) * 1024);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + PF <= KS) lds_wait<PF - 1>(fr[ks % PF]);
        else if (ks + 2 == KS) lds_wait<1>(fr[ks % PF]);
        else lds_wait<0>(fr[ks % PF]);
        const u16x8 kf = fr[ks % PF];
        if (ks + PF < KS) fr[ks % PF] = lds_read128(kblk + (ks + PF) * 1024);
        s0 = mfma32<T16>(kf, qf[ks], s0);
      }
    }
    // bias precursors of the block + the first V^T fragments: requested before the softmax
    f32x4 ub[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ub[g] = lds_read_f4(&us[kb * 32 + 8 * g + 4 * hi]);
    const unsigned char* vblk = kblk + KS * 1024;
    u16x8 fr[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) fr[j] = lds_read128(vblk + (j < 2 * NDB ? j : 2 * NDB - 1) * 1024);
    lds_wait<PF>(ub[0], ub[1]);
    lds_wait<PF>(ub[2], ub[3]);
    f32x16 sacc;
    float mloc = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = s0[r];
      if (p.use_rel) x += fmaxf(uqp - ub[r >> 2][r & 3], 0.f);
      sacc[r] = x * c2;
    }
    if (kb == nkb - 1) {                             // keys >= nppf exist only in the last block
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + c32_row(r, lane) >= p.nppf) sacc[r] = -1e30f;
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
    for (int j = 0; j < 2 * NDB; ++j) {
      if (j + PF <= 2 * NDB) lds_wait<PF - 1>(fr[j % PF]);
      else if (j + 2 == 2 * NDB) lds_wait<1>(fr[j % PF]);
      else lds_wait<0>(fr[j % PF]);
      const u16x8 vf = fr[j % PF];
      if (j + PF < 2 * NDB) fr[j % PF] = lds_read128(vblk + (j + PF) * 1024);
      o[j >> 1] = mfma32<T16>(vf, pf[j & 1], o[j >> 1]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no LDS-DMA may outlive the wave)
  if (!wave_ok) return;
  {
    const float inv_l = 1.0f / l_run;                // normalise the visual part in place
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= inv_l;
  }
  // ---- language keys: one masked block, its own softmax, probabilities normalised before P.V so
  // that it accumulates into the normalised visual output (as attn_struct_kernel)
  if (!(p.dbg & 2)) {
    u16x8 klf[KS];
    {   // language K fragments: lane = key a (rows >= nsrl are zero), 8 consecutive head columns, from LDS
      const bool a_ok = ql < p.nsrl;
      const float* kr = pls + ((a_ok ? ql : 0) * 3 + 1) * DP + hi * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
        if (a_ok) { x0 = *reinterpret_cast<const float4*>(kr + ks * 16); x1 = *reinterpret_cast<const float4*>(kr + ks * 16 + 4); }
        klf[ks] = u16x8{to16<T16>(x0.x), to16<T16>(x0.y), to16<T16>(x0.z), to16<T16>(x0.w),
                        to16<T16>(x1.x), to16<T16>(x1.y), to16<T16>(x1.z), to16<T16>(x1.w)};
      }
    }
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
      // language V fragment of d-block db, k-step ks: lane = (hi, head column), register j = key
      // 16*ks + 8*(j>>2) + 4*hi + (j&3), from LDS
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks >= nksl) break;
        u16x8 vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = ks * 16 + 8 * (j >> 2) + 4 * hi + (j & 3);
          vl[j] = key < p.nsrl ? to16<T16>(pls[(key * 3 + 2) * DP + db * 32 + ql]) : (unsigned short)0;
        }
        o[db] = mfma32<T16>(vl, pf[ks], o[db]);
      }
    }
  }
  if (q_ok && !((p.dbg & 1) && o[0][0] != 123.456f)) {
#pragma unroll
    for (int db = 0; db < NDB; ++db)
      struct_store<T16>(p, o[db], db, (int64_t)s * Nq + qi, h, DP, hi);
  }
}

}  // namespace vog
