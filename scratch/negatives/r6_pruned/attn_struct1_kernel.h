// attn_struct1_kernel (272 registers; replaced by attn_struct1_lean_kernel in round 4, bit-identical) as it stood at the start of round 6
// ---- ONE visual key block (nppf <= 32: every gt5 shape). Both softmaxes are complete before any
// P.V product, so the output is produced d-block by d-block with ONE accumulator: the kernel holds
// Q, K, K_lang (3 x KS fragments) in its first phase and V, V_lang + 16 accumulator registers in the
// second, instead of the flash kernel's NDB accumulators alive through everything (dp = 256: 512
// registers and 104 spills, 23 us).
template <typename T16, int NDB>
__global__ __launch_bounds__(256) void attn_struct1_kernel(AttnStructParams p) {
  constexpr int DP = NDB * 32, KS = DP / 16;
  extern __shared__ __attribute__((aligned(16))) float ssm[];
  float* us = ssm;                                   // [32] bias precursor of the visual keys
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ql = lane & 31;
  VOG_ATS(0);
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

  // The language projection rows of this (video, head) - nsrl x {Q, K, V} x DP floats - are staged in
  // LDS once per workgroup (a few 16-byte loads per thread) instead of ~70 small global loads per
  // wave; Qv and K fragments go straight to registers meanwhile.
  float* pls = ssm + 32;                             // [nsrl][3][DP]
  {
    const int per_row = 3 * DP / 4;                  // float4 per argument
    for (int i = tid; i < p.nsrl * per_row; i += 256) {
      const int a = i / per_row, c = i - a * per_row;
      const int part = c / (DP / 4), dd4 = c - part * (DP / 4);
      *reinterpret_cast<float4*>(&pls[(a * 3 + part) * DP + dd4 * 4]) =
          *reinterpret_cast<const float4*>(plr + (int64_t)a * ldp + part * hd + dd4 * 4);
    }
  }
  u16x8 qf[KS], kf[KS], klf[KS];
  const int qbs = wave_ok ? qb : 0;
  int qa = 0, qp = 0;
  if (p.q_visual) {
    const int t = qbs * 32 + ql;
    qa = t / p.nppf;
    qp = t - qa * p.nppf;
    qa = qa < p.nsrl ? qa : p.nsrl - 1;              // tokens past the end are never stored
    const unsigned short* qv = p.q + kvbase;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const u16x8*>(qv + frag_qk(qp, ks * 16 + hi * 8, DP));
  } else {
    const u16x8* Qf = reinterpret_cast<const u16x8*>(p.q + ((int64_t)s * p.H + h) * (int64_t)p.npad_q * DP) +
                      (int64_t)qbs * KS * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = Qf[ks * 64];
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kf[ks] = Kf[ks * 64];
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    const int64_t u_base = (int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.nppf;
    peb = p.pe_b[h];
    if (tid < 32) us[tid] = tid < p.nppf ? p.u[(u_base + tid) * p.H + h] : 0.f;
    if (q_ok) uq = p.u[(u_base + (qi % p.nppf)) * p.H + h];
  }
  VOG_ATS(1);
  __syncthreads();
  if (!wave_ok) return;
  if (p.q_visual) {                                  // q(a, p) = Qv[p] + Ql[a]
    const float* qlr = pls + (qa * 3 + 0) * DP + hi * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u16x8 v = qf[ks];
      const float4 l0 = *reinterpret_cast<const float4*>(qlr + ks * 16);
      const float4 l1 = *reinterpret_cast<const float4*>(qlr + ks * 16 + 4);
      qf[ks] = u16x8{to16<T16>(from16<T16>(v[0]) + l0.x), to16<T16>(from16<T16>(v[1]) + l0.y),
                     to16<T16>(from16<T16>(v[2]) + l0.z), to16<T16>(from16<T16>(v[3]) + l0.w),
                     to16<T16>(from16<T16>(v[4]) + l1.x), to16<T16>(from16<T16>(v[5]) + l1.y),
                     to16<T16>(from16<T16>(v[6]) + l1.z), to16<T16>(from16<T16>(v[7]) + l1.w)};
    }
  }
  {                                                  // language K fragments: lane = key a
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
  const float c2 = p.inv_scale * 1.44269504088896340736f;   // exp(x * inv_scale) = 2^(x * c2)
  const float uqp = uq + peb;
  // ---- both logit blocks
  f32x16 sv, sl;
  {
    f32x16 s1, l1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] = 0.f; s1[r] = 0.f; sl[r] = 0.f; l1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      sv = mfma32<T16>(kf[ks], qf[ks], sv);
      sl = mfma32<T16>(klf[ks], qf[ks], sl);
      if (ks + 1 < KS) {
        s1 = mfma32<T16>(kf[ks + 1], qf[ks + 1], s1);
        l1 = mfma32<T16>(klf[ks + 1], qf[ks + 1], l1);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] += s1[r]; sl[r] += l1[r]; }
  }
  VOG_ATS(2);
  // ---- two independent softmaxes, probabilities normalised before P.V
  u16x8 pv_[2], pl_[2];
  {
    float mv = -1e30f, ml = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = c32_row(r, lane);
      float x = sv[r];
      if (p.use_rel) x += fmaxf(uqp - us[key], 0.f);
      x = key < p.nppf ? x * c2 : -1e30f;
      const float y = key < p.nsrl ? sl[r] * c2 : -1e30f;
      sv[r] = x; sl[r] = y;
      mv = fmaxf(mv, x); ml = fmaxf(ml, y);
    }
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
  VOG_ATS(3);
  // ---- output, one d-block at a time
  const int nksl = p.nsrl > 16 ? 2 : 1;
#pragma unroll
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
  VOG_ATS(4);
#ifdef VOG_TS_ATTN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  VOG_ATS(5);
#endif
}

