// Device side of gemm.hip (kernel bodies; also included by pair.hip, which fuses two bodies into one launch).
#pragma once
#include "common.h"

namespace vog {

enum { EPI_PLAIN = 0, EPI_QKV = 1 };
#ifndef VOG_GEMM_COLMAJOR
#define VOG_GEMM_COLMAJOR 1      // 0 (perf experiments): row-major tile order for every shape
#endif
struct GemmParams;
static bool pipe_ok(const GemmParams& p, bool a_f32);

// VOG_GEMM_DEBUG (ablation, perf experiments only): 1 = no DMA, 2 = no MFMA, 4 = no epilogue
static int gemm_debug_flags() {
  static int v = -1;
  if (v < 0) { const char* e = perf_env("VOG_GEMM_DEBUG"); v = e ? atoi(e) : 0; }
  return v;
}

struct GemmParams {
  const void* a; const int32_t* a_rows; int64_t lda;
  const unsigned short* w; int64_t ldw;
  const unsigned short* w_p32;         // row-block QKV (qkvrb_dev.h): weights in vog_pack_w_frag32 order
  const float* bias; const float* residual; int64_t ldr;
  float* c32; unsigned short* c16; int64_t ldc; int64_t ldc16;
  int M, N, K; int relu; int rep; int c16_bf16; int debug;
  const int32_t* out_rows; int out_rows_ncol;
  int splitk; int w_frag; int a_frag;
  // implicit vis||lang residual (res_vis != nullptr)
  const float* res_vis; const float* res_lang; int rv_nfrm, rv_nppf, rv_nsrl, rv_dv, rv_dl, rv_lpv, rv_ncv;
  // division of a row index by loop-invariant counts: q = (umulhi(n, mul) + n) >> shr (n < 2^31)
  unsigned fdN_mul, fdN_shr, fdP_mul, fdP_shr, fdF_mul, fdF_shr, fdC_mul, fdC_shr;
  unsigned fdT_mul, fdT_shr;           // QKV epilogue: row / tokens-per-sequence (plain fragment writers)
  // QKV epilogue
  unsigned short* q; unsigned short* k; unsigned short* vt;
  int ntok, H, dp, npad;
  // structured QKV (pl != nullptr): rows are visual rows, fan out over nsrl arguments
  const float* pl; int st_nsrl, st_nppf, st_nfrm, st_lpv, st_ncv;
  // st_kv_vis: K and V fragments only for the VISUAL rows (ntok = nppf per sequence, npad_kv), no
  // language part added: the separable attention (attention.hip, attn_struct_kernel) adds it itself
  int st_kv_vis, npad_kv;
  // round 6, hi + lo operands: a_lo / w_lo = t16(x - t16(x)) of the fp32 activations / weights in the layout of a / w (w_lo in
  // fragment order for the M <= 64 kernel); product = a.w + a_lo.w + a.w_lo (three MFMAs, fp32 accumulate). QKV epilogue:
  // q_lo / k_lo = the remainders of the Q / K fragments (the attention kernels' hi + lo contraction reads them).
  const void* a_lo; const unsigned short* w_lo; unsigned short* q_lo; unsigned short* k_lo;
};

template <typename T16, bool A_F32>
__device__ __forceinline__ u16x8 load_a_chunk(const void* a, int64_t row_off, int col, bool ok) {
  u16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!ok) return r;
  if constexpr (A_F32) {
    const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a) + row_off + col);
    float4 x = p[0], y = p[1];
    r[0] = to16<T16>(x.x); r[1] = to16<T16>(x.y); r[2] = to16<T16>(x.z); r[3] = to16<T16>(x.w);
    r[4] = to16<T16>(y.x); r[5] = to16<T16>(y.y); r[6] = to16<T16>(y.z); r[7] = to16<T16>(y.w);
  } else {
    r = *reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(a) + row_off + col);
  }
  return r;
}

// residual pointer of token row m, column n, for the implicit vis||lang token matrix
// (row m = (s=(v,f), j=a*nppf+p); a 4-column chunk never straddles dv since dv % 4 == 0)
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned shr) {
  return (int)(((unsigned long long)__umulhi((unsigned)n, mul) + (unsigned)n) >> shr);
}
// (three hardware integer divisions per 16-byte chunk here were ~70 of the 250 us of the p100 Wo GEMM)
__device__ __forceinline__ const float* vislang_res_ptr(const GemmParams& p, int m, int n) {
  const int N = p.rv_nsrl * p.rv_nppf;
  const int s = fast_div(m, p.fdN_mul, p.fdN_shr), j = m - s * N;
  const int a = fast_div(j, p.fdP_mul, p.fdP_shr), pp = j - a * p.rv_nppf;
  if (n < p.rv_dv)       // visual row of sequence s = (v, f): v*nfrm*nppf + f*nppf + pp = s*nppf + pp
    return p.res_vis + ((int64_t)s * p.rv_nppf + pp) * p.rv_dv + n;
  const int v = fast_div(s, p.fdF_mul, p.fdF_shr);
  const int lv = p.rv_lpv ? v : fast_div(v, p.fdC_mul, p.fdC_shr);
  return p.res_lang + ((int64_t)lv * p.rv_nsrl + a) * p.rv_dl + (n - p.rv_dv);
}

template <typename T16>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int row, int col, float v) {
  if (row >= p.M || col >= p.N) return;
  {
    if (p.bias) v += p.bias[col];
    if (p.residual) v += p.residual[(int64_t)row * p.ldr + col];
    if (p.res_vis) v += *vislang_res_ptr(p, row, col);
    if (p.relu) v = relu_nan(v);
    if (p.out_rows) {
      const int orow = p.out_rows[(int64_t)(col / p.out_rows_ncol) * p.M + row];
      if (orow < 0) return;
      if (p.c32) p.c32[(int64_t)orow * p.ldc + col] = v;
      if (p.c16) p.c16[(int64_t)orow * p.ldc16 + col] = p.c16_bf16 ? to16<BF16>(v) : to16<F16>(v);
      return;
    }
    for (int j = 0; j < p.rep; ++j) {
      int64_t orow = (int64_t)row * p.rep + j;
      if (p.c32) p.c32[orow * p.ldc + col] = v;
      if (p.c16) p.c16[orow * p.ldc16 + col] = p.c16_bf16 ? to16<BF16>(v) : to16<F16>(v);
    }
  }
}

// QKV epilogue for one 32x32 accumulator fragment. dp % 32 == 0 and fragment
// column bases are multiples of 32, so (which, head) is WAVE-UNIFORM: derive it
// from the fragment base through readfirstlane and branch on scalars. (A
// per-lane 3-way pointer select here was miscompiled by hipcc 7.2: the V^T
// stores went through the K base pointer.)
template <typename T16>
__device__ __forceinline__ void qkv_store_frag(const GemmParams& p, int row0, int col0, int lane,
                                               const f32x16& acc) {
  col0 = __builtin_amdgcn_readfirstlane(col0);
  if (col0 >= p.N) return;
  const int hd = p.H * p.dp;
  const int which = col0 / hd;
  const int h = (col0 - which * hd) / p.dp;
  const int dd = (col0 % p.dp) + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + c32_row(r, lane);
    if (row >= p.M) continue;
    const int s = row / p.ntok;
    const int i = row - s * p.ntok;
    const int64_t base = ((int64_t)s * p.H + h) * p.npad * p.dp;
    const unsigned short o = to16<T16>(acc[r]);
    if (which == 0) {
      p.q[base + frag_qk(i, dd, p.dp)] = o;
    } else if (which == 1) {
      p.k[base + frag_qk(i, dd, p.dp)] = o;
    } else {
      p.vt[base + frag_v(i, dd, p.dp)] = o;
    }
  }
}

// ----------------------------------------------------------------------------
// tiled kernel
// ----------------------------------------------------------------------------
constexpr int BK = 64;
constexpr int LDS_LD = BK + 8;   // 144 B rows: conflict-free ds_read_b128 (36 dwords stride)

template <typename T16, int BM, int BN, bool A_F32, int EPI>
__global__ __launch_bounds__(256) void gemm_tiled(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned short As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[BN * LDS_LD];
  constexpr int FM = BM / 64, FN = BN / 64;        // 32x32 fragments per wave
  constexpr int CA = BM / 32, CB = BN / 32;        // 16-byte chunks per thread per tile
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  // XCD-aware tile order: consecutive linear ids that share an A panel stay on one XCD
  // (block b is observed on XCD b % 8; speed only, never correctness).
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = bid / nbn, bn = bid % nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-thread chunk coordinates (fixed across K tiles)
  int64_t a_off[CA]; bool a_ok[CA]; int a_lds[CA];
  int64_t b_off[CB]; bool b_ok[CB]; int b_lds[CB];
  const int ccol = (tid & 7) * 8;
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int r = (tid >> 3) + i * 32, m = m0 + r;
    a_ok[i] = m < p.M;
    const int64_t src = a_ok[i] ? (p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m) : 0;
    a_off[i] = src * p.lda;
    a_lds[i] = r * LDS_LD + ccol;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    const int r = (tid >> 3) + i * 32, n = n0 + r;
    b_ok[i] = n < p.N;
    b_off[i] = (int64_t)(b_ok[i] ? n : 0) * p.ldw;
    b_lds[i] = r * LDS_LD + ccol;
  }
  u16x8 ra[CA], rb[CB];
  auto gload = [&](int k0) {
    const bool kok = (k0 + ccol) < p.K;     // K % 8 == 0: a chunk is all-in or all-out
#pragma unroll
    for (int i = 0; i < CA; ++i) ra[i] = load_a_chunk<T16, A_F32>(p.a, a_off[i], k0 + ccol, a_ok[i] && kok);
#pragma unroll
    for (int i = 0; i < CB; ++i) rb[i] = load_a_chunk<T16, false>(p.w, b_off[i], k0 + ccol, b_ok[i] && kok);
  };

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int i = 0; i < CA; ++i) *reinterpret_cast<u16x8*>(&As[a_lds[i]]) = ra[i];
#pragma unroll
    for (int i = 0; i < CB; ++i) *reinterpret_cast<u16x8*>(&Bs[b_lds[i]]) = rb[i];
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      u16x8 fa[FM], fb[FN];
      const int kk = ks * 16 + (lane >> 5) * 8;
#pragma unroll
      for (int i = 0; i < FM; ++i)
        fa[i] = *reinterpret_cast<const u16x8*>(&As[(wm * (BM / 2) + i * 32 + (lane & 31)) * LDS_LD + kk]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        fb[j] = *reinterpret_cast<const u16x8*>(&Bs[(wn * (BN / 2) + j * 32 + (lane & 31)) * LDS_LD + kk]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma32<T16>(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      if constexpr (EPI == EPI_QKV) {
        qkv_store_frag<T16>(p, m0 + wm * (BM / 2) + i * 32, n0 + wn * (BN / 2) + j * 32, lane, acc[i][j]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 32 + c32_row(r, lane);
          epilogue_store<T16>(p, row, col, acc[i][j][r]);
        }
      }
    }
}

// QKV epilogue of one wave tile parked in LDS (ep[row][EP_LD], WTM rows x WTN columns, origin (mw, nw)):
// writes the Q / K / V^T MFMA-fragment images the attention kernels read. Shared by the LDS-DMA GEMM
// below and the row-block kernel of qkvrb_dev.h.
template <typename T16, int WTM, int WTN>
__device__ __forceinline__ void qkv_epilogue_tile(const GemmParams& p, const float* ep, int mw, int nw, int lane) {
  constexpr int EP_LD = WTN + 4;
  // QKV: handle the wave tile in 32-column groups; (which, head) is uniform per group
  const int hd = p.H * p.dp;
#pragma unroll
  for (int cg = 0; cg < WTN / 32; ++cg) {
    const int nb = __builtin_amdgcn_readfirstlane(nw + cg * 32);
    if (nb >= p.N) continue;
    const int which = nb / hd;
    const int h = (nb - which * hd) / p.dp;
    const int dd0 = nb % p.dp;
    // (token count per sequence, padded count) of the plain fragment writers below
    const bool kv_vis = p.pl && p.st_kv_vis && which >= 1;
    const int ntok_w = kv_vis ? p.st_nppf : p.ntok, npad_w = kv_vis ? p.npad_kv : p.npad;
    if (p.pl && !kv_vis) {
      // structured layer 0: row m = visual row (v, f, p'); token(arg) = arg*nppf + p'
      const int ldp = 3 * hd;
      if (which < 2) {
        // one lane = 8 consecutive head columns of one visual row = ONE 16-byte fragment chunk per
        // argument (vs two 8-byte halves from adjacent lanes: same 13.9 us at cfg 2 - the epilogue's
        // 8.5 us are the 18 MB of fan-out writes themselves, not their granularity)
        unsigned short* base = which == 0 ? p.q : p.k;
        const int c = lane & 3, rsub = lane >> 2;
#pragma unroll 2
        for (int ps = 0; ps < WTM / 16; ++ps) {
          const int rl = ps * 16 + rsub;
          const int m = mw + rl;
          if (m >= p.M) continue;
          const float4 v0 = *reinterpret_cast<const float4*>(&ep[rl * EP_LD + cg * 32 + 8 * c]);
          const float4 v1 = *reinterpret_cast<const float4*>(&ep[rl * EP_LD + cg * 32 + 8 * c + 4]);
          const int sq = m / p.st_nppf, pp = m - sq * p.st_nppf;
          const int vid = sq / p.st_nfrm;
          const int lv = p.st_lpv ? vid : vid / p.st_ncv;
          const float* plr = p.pl + (int64_t)lv * p.st_nsrl * ldp + nb + 8 * c;
          unsigned short* dst = base + ((int64_t)sq * p.H + h) * p.npad * p.dp;
          for (int ar = 0; ar < p.st_nsrl; ++ar) {
            const float4 l0 = *reinterpret_cast<const float4*>(plr + (int64_t)ar * ldp);
            const float4 l1 = *reinterpret_cast<const float4*>(plr + (int64_t)ar * ldp + 4);
            const u16x8 o = {to16<T16>(v0.x + l0.x), to16<T16>(v0.y + l0.y), to16<T16>(v0.z + l0.z), to16<T16>(v0.w + l0.w),
                             to16<T16>(v1.x + l1.x), to16<T16>(v1.y + l1.y), to16<T16>(v1.z + l1.z), to16<T16>(v1.w + l1.w)};
            *reinterpret_cast<u16x8*>(dst + frag_qk(ar * p.st_nppf + pp, dd0 + 8 * c, p.dp)) = o;
          }
        }
      } else if ((p.st_nppf & 3) == 0) {
        // V fragments, vector form: an aligned group of 4 visual rows = 4 consecutive tokens of
        // every argument = 4 consecutive j of one fragment lane -> one 8-byte store. Lanes run
        // along dd: conflict-free LDS column reads, 16-byte-strided global stores.
        const int dd = lane & 31, gsub = lane >> 5;
        for (int rg = gsub; rg < WTM / 4; rg += 2) {
          const int rl = rg * 4;
          const int m = mw + rl;
          if (m >= p.M) continue;                    // M % 4 == 0 here (nppf % 4 == 0)
          const int sq = m / p.st_nppf, pp = m - sq * p.st_nppf;
          const int vid = sq / p.st_nfrm;
          const int lv = p.st_lpv ? vid : vid / p.st_ncv;
          const float* plr = p.pl + (int64_t)lv * p.st_nsrl * ldp + nb + dd;
          unsigned short* dst = p.vt + ((int64_t)sq * p.H + h) * p.npad * p.dp;
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = ep[(rl + e) * EP_LD + cg * 32 + dd];
          for (int ar = 0; ar < p.st_nsrl; ++ar) {
            const float l = plr[(int64_t)ar * ldp];
            const u16x4 o = {to16<T16>(x[0] + l), to16<T16>(x[1] + l), to16<T16>(x[2] + l), to16<T16>(x[3] + l)};
            *reinterpret_cast<u16x4*>(dst + frag_v(ar * p.st_nppf + pp, dd0 + dd, p.dp)) = o;
          }
        }
      } else {
#pragma unroll
        for (int th = 0; th < WTM / 64 + (WTM % 64 ? 1 : 0); ++th) {
          const int rl = th * 64 + lane;
          const int m = mw + rl;
          if (rl < WTM && m < p.M) {
            const int sq = m / p.st_nppf, pp = m - sq * p.st_nppf;
            const int vid = sq / p.st_nfrm;
            const int lv = p.st_lpv ? vid : vid / p.st_ncv;
            const float* plr = p.pl + (int64_t)lv * p.st_nsrl * ldp + nb;
            unsigned short* dst = p.vt + ((int64_t)sq * p.H + h) * p.npad * p.dp;
            for (int ar = 0; ar < p.st_nsrl; ++ar) {
              unsigned short* d2 = dst + frag_v(ar * p.st_nppf + pp, dd0, p.dp);
              const float* l = plr + (int64_t)ar * ldp;
#pragma unroll 8
              for (int dd = 0; dd < 32; ++dd)
                d2[dd * 8] = to16<T16>(ep[rl * EP_LD + cg * 32 + dd] + l[dd]);
            }
          }
        }
      }
    } else if (which < 2) {
      unsigned short* base = which == 0 ? p.q : p.k;
      unsigned short* base_lo = which == 0 ? p.q_lo : p.k_lo;   // (hi + lo operands: the remainder fragments, or null)
      const int c = lane & 3, rsub = lane >> 2;     // 4 chunks of 8 columns (one 16-byte fragment piece) per row, 16 rows per pass
#pragma unroll
      for (int ps = 0; ps < WTM / 16; ++ps) {
        const int rl = ps * 16 + rsub;
        const int m = mw + rl;
        if (m >= p.M) continue;
        const float4 v = *reinterpret_cast<const float4*>(&ep[rl * EP_LD + cg * 32 + 8 * c]);
        const float4 w = *reinterpret_cast<const float4*>(&ep[rl * EP_LD + cg * 32 + 8 * c + 4]);
        const int sq = fast_div(m, p.fdT_mul, p.fdT_shr), tok = m - sq * ntok_w;
        const u16x8 o = {to16<T16>(v.x), to16<T16>(v.y), to16<T16>(v.z), to16<T16>(v.w),
                         to16<T16>(w.x), to16<T16>(w.y), to16<T16>(w.z), to16<T16>(w.w)};
        const int64_t off = ((int64_t)sq * p.H + h) * npad_w * p.dp + frag_qk(tok, dd0 + 8 * c, p.dp);
        *reinterpret_cast<u16x8*>(base + off) = o;
        if (base_lo) {
          const u16x8 l = {to16<T16>(v.x - from16<T16>(o[0])), to16<T16>(v.y - from16<T16>(o[1])),
                           to16<T16>(v.z - from16<T16>(o[2])), to16<T16>(v.w - from16<T16>(o[3])),
                           to16<T16>(w.x - from16<T16>(o[4])), to16<T16>(w.y - from16<T16>(o[5])),
                           to16<T16>(w.z - from16<T16>(o[6])), to16<T16>(w.w - from16<T16>(o[7]))};
          *reinterpret_cast<u16x8*>(base_lo + off) = l;
        }
      }
    } else if ((ntok_w & 3) == 0 && (mw & 3) == 0) {
      // V fragments, vector form (round 6; as the structured writer above): an aligned group of 4 token rows = 4 consecutive j of
      // one fragment lane -> one 8-byte store; lanes run along dd (conflict-free LDS column reads). The token-per-lane form below
      // issues 32 two-byte stores per lane: most of the 1.8 us this epilogue took of a 7.4 us QKV launch at cfg 2.
      const int dd = lane & 31, gsub = lane >> 5;
#pragma unroll
      for (int rg = gsub; rg < WTM / 4; rg += 2) {
        const int rl = rg * 4;
        const int m = mw + rl;
        if (m >= p.M) continue;                      // (M % 4 == 0: ntok % 4 == 0)
        const int sq = fast_div(m, p.fdT_mul, p.fdT_shr), tok = m - sq * ntok_w;
        const u16x4 o = {to16<T16>(ep[rl * EP_LD + cg * 32 + dd]), to16<T16>(ep[(rl + 1) * EP_LD + cg * 32 + dd]),
                         to16<T16>(ep[(rl + 2) * EP_LD + cg * 32 + dd]), to16<T16>(ep[(rl + 3) * EP_LD + cg * 32 + dd])};
        *reinterpret_cast<u16x4*>(p.vt + ((int64_t)sq * p.H + h) * npad_w * p.dp + frag_v(tok, dd0 + dd, p.dp)) = o;
      }
    } else {
      // V fragments: lane = token; 2-byte stores inside this token's fragment block
#pragma unroll
      for (int th = 0; th < WTM / 64 + (WTM % 64 ? 1 : 0); ++th) {
        const int rl = th * 64 + lane;
        const int m = mw + rl;
        if (rl < WTM && m < p.M) {
          const int sq = fast_div(m, p.fdT_mul, p.fdT_shr), tok = m - sq * ntok_w;
          // dd0 % 32 == 0: the 32 columns of this group are one d-block of the V fragment
          unsigned short* dst = p.vt + ((int64_t)sq * p.H + h) * npad_w * p.dp + frag_v(tok, dd0, p.dp);
#pragma unroll 8
          for (int dd = 0; dd < 32; ++dd)
            dst[dd * 8] = to16<T16>(ep[rl * EP_LD + cg * 32 + dd]);
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------
// pipelined kernel: K % 64 == 0, 16-bit A. Global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass), STAGES-deep
// ring, ONE raw s_barrier per K tile, counted vmcnt so the next tile's DMA stays
// in flight across the barrier (the tiled kernel above exposes a full L2/HBM
// round trip per 64-deep step; at M = 4000 that, not MFMA issue, set its time).
//
// LDS image: rows of 128 B (64 halfwords), lane-linear per DMA instruction
// (1 KiB = 8 rows). Bank-conflict-free ds_read_b128 needs 16 consecutive rows on
// 16 distinct 16-B slots of the 256-B bank row: slot = (row&1)*8 + (chunk ^
// ((row>>1)&7)). The DMA destination cannot be permuted, so the permutation is
// applied to the per-lane SOURCE chunk and, identically, to the read address
// (same involution on both sides).
// ----------------------------------------------------------------------------
// SPLIT (round 6): hi + lo operands. A stage holds two images, [A | W] and [A_lo | W_lo] (same rows, same swizzle); a k-step is
// three MFMAs (a.w + a.w_lo + a_lo.w). Twice the DMA instructions and LDS bytes per tile, fp32-grade products at 1/3 of the 16-bit
// MFMA rate - the QKV projections of a checkpoint whose attention is too sharp for 16-bit logits.
template <typename T16, int BM, int BN, int STAGES, int EPI, bool SPLIT = false>
struct GemmPipeBody {
  using Params = GemmParams;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(GemmParams p, const BlockCtx& cx, unsigned char* smem) {
  constexpr int ROWS = BM + BN;
  constexpr int HALF_BYTES = ROWS * 128;
  constexpr int STAGE_BYTES = HALF_BYTES * (SPLIT ? 2 : 1);
  constexpr int LPH = ROWS / 32;                   // DMA instructions per wave per tile and image
  constexpr int LPT = LPH * (SPLIT ? 2 : 1);       // ... per tile
  constexpr int FM = BM / 64, FN = BN / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = cx.bx;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // Few row blocks over many column blocks (the BiLSTM input projections of cfg 3 / cfg 5: M = 96 / 192, N = 8192, K = 2048): the
  // row blocks of ONE column panel are neighbours (round 6), i.e. they run on the same XCD at the same time and the 256 KB weight
  // panel is fetched from the fabric by one of them and found in that L2 by the others. In row-major tile order the 2-3 readers
  // of a panel sit on different XCDs and the 33.5 MB of W_ih cross the fabric 2-3 times.
  const bool col_major = nbm <= 4 && nbn >= 32 && VOG_GEMM_COLMAJOR;
  const int bm = col_major ? bid % nbm : bid / nbn, bn = col_major ? bid / nbm : bid % nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // per-lane source pointers of this wave's DMA instructions (advance by 64 halfwords per tile)
  const unsigned short* gsrc[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int ih = i % LPH;                                  // (SPLIT: i >= LPH addresses the remainder images)
    const bool lo = i >= LPH;
    const int rr = (wid * LPH + ih) * 8 + (lane >> 3);       // row in the combined [A | W] tile
    const int c = (lane & 7) ^ ((rr >> 1) & 7);              // source chunk for LDS chunk lane&7
    if (rr < BM) {
      int m = m0 + rr;
      m = m < p.M ? m : p.M - 1;                             // clamp: rows >= M are discarded later
      const int64_t src = p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m;
      gsrc[i] = reinterpret_cast<const unsigned short*>(lo ? p.a_lo : p.a) + src * p.lda + c * 8;
    } else {
      int n = n0 + rr - BM;
      n = n < p.N ? n : p.N - 1;
      gsrc[i] = (lo ? p.w_lo : p.w) + (int64_t)n * p.ldw + c * 8;
    }
  }
  auto issue = [&](int kt, int stage) {
    if (p.debug & 1) return;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(gsrc[i] + (int64_t)kt * 64),
          (__attribute__((address_space(3))) void*)(smem + stage * STAGE_BYTES + (i >= LPH ? HALF_BYTES : 0) +
                                                    (wid * LPH + (i % LPH)) * 1024),
          16, 0, 0);
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int a_row[FM], b_row[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) a_row[i] = wm * (BM / 2) + i * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < FN; ++j) b_row[j] = BM + wn * (BN / 2) + j * 32 + (lane & 31);
  const int hi = lane >> 5;

  // split-K: grid row y owns k tiles [kbeg, kbeg + nk) and its own fp32 output slab
  int nk = p.K / 64;
  int kbeg = 0;
  if (p.splitk > 1) {
    const int per = (nk + p.splitk - 1) / p.splitk;
    kbeg = cx.by * per;
    nk = nk - kbeg < per ? nk - kbeg : per;
    if (nk < 0) nk = 0;
    p.c32 += (int64_t)cx.by * p.M * p.ldc;
#pragma unroll
    for (int i = 0; i < LPT; ++i) gsrc[i] += (int64_t)kbeg * 64;
  }
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    // tiles in flight now: kt .. min(kt+STAGES-2, nk-1). Retire tile kt only.
    // (counted exactly for every ring depth, round 6: the 2-tile cap of the first form made rings deeper than 4 stages pointless)
    const int ahead = (nk - 1 - kt) < (STAGES - 2) ? (nk - 1 - kt) : (STAGES - 2);
    static_assert((STAGES - 2) * LPT <= 63 && STAGES <= 9, "vmcnt is a 6-bit counter");
    switch (ahead) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT <= 63 ? 3 * LPT : 63) : "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPT <= 63 ? 4 * LPT : 63) : "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * LPT <= 63 ? 5 * LPT : 63) : "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * LPT <= 63 ? 6 * LPT : 63) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(7 * LPT <= 63 ? 7 * LPT : 63) : "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    const unsigned char* st = smem + (kt % STAGES) * STAGE_BYTES;
    if (p.debug & 2) continue;
    // all fragments of the tile are requested before its first MFMA (the waits then step down with the
    // arrivals): read-then-multiply per k-step exposed one LDS round trip per MFMA group
    u16x8 fa[4][FM], fb[4][FN];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int g = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < FM; ++i)
        fa[ks][i] = *reinterpret_cast<const u16x8*>(st + a_row[i] * 128 + ((g ^ ((a_row[i] >> 1) & 7)) << 4));
#pragma unroll
      for (int j = 0; j < FN; ++j)
        fb[ks][j] = *reinterpret_cast<const u16x8*>(st + b_row[j] * 128 + ((g ^ ((b_row[j] >> 1) & 7)) << 4));
    }
    if constexpr (SPLIT) {
      // the remainder images one k-step at a time (8 more fragment registers instead of 32): a.w + a.w_lo + a_lo.w
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int g = ks * 2 + hi;
        u16x8 la[FM], lb[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
          la[i] = *reinterpret_cast<const u16x8*>(st + HALF_BYTES + a_row[i] * 128 + ((g ^ ((a_row[i] >> 1) & 7)) << 4));
#pragma unroll
        for (int j = 0; j < FN; ++j)
          lb[j] = *reinterpret_cast<const u16x8*>(st + HALF_BYTES + b_row[j] * 128 + ((g ^ ((b_row[j] >> 1) & 7)) << 4));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            acc[i][j] = mfma32<T16>(fb[ks][j], fa[ks][i], acc[i][j]);
            acc[i][j] = mfma32<T16>(lb[j], fa[ks][i], acc[i][j]);
            acc[i][j] = mfma32<T16>(fb[ks][j], la[i], acc[i][j]);
          }
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma32<T16>(fb[ks][j], fa[ks][i], acc[i][j]);   // C^T: D[n][m]
    }
  }
  if (p.debug & 4) { if (acc[0][0][0] != 123.456f) return; }
  // ---- epilogue through LDS -------------------------------------------------------
  // The MFMA C layout gives a lane 4-element column strips of many rows; stores
  // straight from it touch 32-64 distinct cache lines per instruction (measured:
  // 26 us of a 62 us QKV launch). Each wave parks its (BM/2 x BN/2) fp32 tile in
  // the (now idle) stage buffers and re-reads it row-wise, so every global
  // load/store instruction covers whole 128-256 B row segments.
  constexpr int WTM = BM / 2, WTN = BN / 2, EP_LD = WTN + 4;
  __builtin_amdgcn_s_barrier();                       // all waves done with the last stage
  asm volatile("" ::: "memory");
  float* ep = reinterpret_cast<float*>(smem) + wid * (WTM * EP_LD);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(&ep[(i * 32 + (lane & 31)) * EP_LD + j * 32 + 8 * g + 4 * hi]) =
            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
  const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;   // wave tile origin
  if constexpr (EPI == EPI_PLAIN) {
    constexpr int CPR = WTN / 4, RPP = 64 / CPR;      // 16-B chunks per row, rows per pass
    const int c = lane % CPR, rsub = lane / CPR;
    const int n = nw + 4 * c;
    const bool vec = (p.N & 3) == 0;
    if (n < p.N) {
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && vec) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll 4
      for (int ps = 0; ps < WTM / RPP; ++ps) {
        const int rl = ps * RPP + rsub;
        const int m = mw + rl;
        if (m >= p.M) continue;
        float4 v = *reinterpret_cast<const float4*>(&ep[rl * EP_LD + 4 * c]);
        if (vec) {
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          if (p.residual) {
            const float4 r = *reinterpret_cast<const float4*>(p.residual + (int64_t)m * p.ldr + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          if (p.res_vis) {
            const float4 r = *reinterpret_cast<const float4*>(vislang_res_ptr(p, m, n));
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          if (p.relu) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }
          u16x4 h;
          if (p.c16) {
            if (p.c16_bf16) h = u16x4{to16<BF16>(v.x), to16<BF16>(v.y), to16<BF16>(v.z), to16<BF16>(v.w)};
            else h = u16x4{to16<F16>(v.x), to16<F16>(v.y), to16<F16>(v.z), to16<F16>(v.w)};
          }
          if (p.out_rows) {
            const int orow = p.out_rows[(int64_t)(n / p.out_rows_ncol) * p.M + m];
            if (orow >= 0) {
              if (p.c32) *reinterpret_cast<float4*>(p.c32 + (int64_t)orow * p.ldc + n) = v;
              if (p.c16) *reinterpret_cast<u16x4*>(p.c16 + (int64_t)orow * p.ldc16 + n) = h;
            }
            continue;
          }
          for (int j = 0; j < p.rep; ++j) {
            const int64_t orow = (int64_t)m * p.rep + j;
            if (p.c32) *reinterpret_cast<float4*>(p.c32 + orow * p.ldc + n) = v;
            if (p.c16) *reinterpret_cast<u16x4*>(p.c16 + orow * p.ldc16 + n) = h;
          }
        } else {
          epilogue_store<T16>(p, m, n, v.x); epilogue_store<T16>(p, m, n + 1, v.y);
          epilogue_store<T16>(p, m, n + 2, v.z); epilogue_store<T16>(p, m, n + 3, v.w);
        }
      }
    }
  } else {
    qkv_epilogue_tile<T16, WTM, WTN>(p, ep, mw, nw, lane);
  }
}
};

template <typename T16, int BM, int BN, int STAGES, int EPI, bool SPLIT = false>
__global__ __launch_bounds__(256) void gemm_pipe(GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  GemmPipeBody<T16, BM, BN, STAGES, EPI, SPLIT>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, smem);
}

// ----------------------------------------------------------------------------
// skinny kernel (M <= 64, K % 32 == 0)
// ----------------------------------------------------------------------------
// SK_CH = k-steps (of 32) per register chunk. The W panel is the HBM-bound stream of this
// kernel: with K = 2048 a wave owns 16 k-steps, and all 16 of its weight fragments are
// requested before anything else (one round trip instead of two).
// SPLIT (round 6; fp32 A, fragment-ordered W and W_lo): hi + lo operands, three MFMAs per k-step - the language half of
// mul_tx's layer-0 QKV for checkpoints whose attention is too sharp for 16-bit logits.
template <typename T16, bool A_F32, int SK_CH, int NT, int KW = 4, bool SPLIT = false>
struct GemmSkinnyBody {
  using Params = GemmParams;
  static constexpr int THREADS = KW * 64;
  static __device__ __forceinline__ void run(const GemmParams& p, const BlockCtx& cx, unsigned char* smem) {
  // NT 16-column tiles per workgroup: every A fragment a wave loads feeds NT MFMAs, so the
  // L2 traffic for A (re-read by every workgroup) drops by NT. KW waves split K. (NT, KW) = (1, 4)
  // is the general form; (2, 8) keeps the wave count of (1, 4) with half the workgroups, i.e. half
  // the re-reads of the activation panel, for the N >= 8192 LSTM input projections whose L2->CU
  // traffic was 80 % activation re-reads.
  float (*red)[NT][4][64][4] = reinterpret_cast<float (*)[NT][4][64][4]>(smem);   // [wave][ntile][mtile][lane][reg]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ct0 = cx.bx * NT;            // first 16-column tile
  const int kg = (lane >> 4) * 8;
  const int mt_all = (p.M + 15) / 16;         // <= 4
  // grid.y > 1: one 16-row tile of A per workgroup (few output columns: parallelism
  // matters more than re-reading the small W panel)
  const int mt_lo = cx.gy > 1 ? (int)cx.by : 0;
  const int mt_n = cx.gy > 1 ? mt_lo + 1 : mt_all;
  const int ksteps = p.K / 32;
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  int64_t a_off[4]; bool a_ok[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = mt * 16 + (lane & 15);
    a_ok[mt] = (mt >= mt_lo) && (mt < mt_n) && (m < p.M);
    const int64_t src = a_ok[mt] ? (p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m) : 0;
    a_off[mt] = src * p.lda;
  }
  int64_t w_off[NT]; bool n_ok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = (ct0 + t) * 16 + (lane & 15);
    n_ok[t] = n < p.N;
    w_off[t] = (int64_t)(n_ok[t] ? n : 0) * p.ldw;
  }

  // wave `wid` owns k-steps wid, wid+4, ... ; processed SK_CH at a time
  for (int base = wid; base < ksteps; base += KW * SK_CH) {
    u16x8 fw[NT][SK_CH];
    u16x8 fwl[SPLIT ? NT : 1][SPLIT ? SK_CH : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < SK_CH; ++c) {
        const int ks = base + c * KW;
        if (p.w_frag) {   // one contiguous KiB per (column tile, k-step)
          u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          fw[t][c] = (ks < ksteps && n_ok[t]) ? *reinterpret_cast<const u16x8*>(
                                    p.w + (((int64_t)(ct0 + t) * ksteps + ks) * 64 + lane) * 8) : z;
          if constexpr (SPLIT)
            fwl[t][c] = (ks < ksteps && n_ok[t]) ? *reinterpret_cast<const u16x8*>(
                                       p.w_lo + (((int64_t)(ct0 + t) * ksteps + ks) * 64 + lane) * 8) : z;
        } else {
          fw[t][c] = load_a_chunk<T16, false>(p.w, w_off[t], ks * 32 + kg, n_ok[t] && ks < ksteps);
        }
      }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt >= mt_lo && mt < mt_n) {
        u16x8 fa[SK_CH];
#pragma unroll
        for (int c = 0; c < SK_CH; ++c) {
          const int ks = base + c * KW;
          if (!A_F32 && p.a_frag) {   // contiguous KiB per (row tile, k-step); pad rows are zero-filled
            u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            fa[c] = ks < ksteps ? *reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(p.a) +
                                      (((int64_t)mt * ksteps + ks) * 64 + lane) * 8) : z;
          } else {
            fa[c] = load_a_chunk<T16, A_F32>(p.a, a_off[mt], ks * 32 + kg, a_ok[mt] && ks < ksteps);
          }
        }
        if constexpr (SPLIT) {
          // fp32 rows again for the remainder: lo = t16(x - t16(x)) (A_F32 only)
          u16x8 fal[SK_CH];
#pragma unroll
          for (int c = 0; c < SK_CH; ++c) {
            const int ks = base + c * KW;
            u16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
            if (a_ok[mt] && ks < ksteps) {
              const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.a) + a_off[mt] + ks * 32 + kg);
              const float4 x = src[0], y = src[1];
              const float xs[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
              for (int j = 0; j < 8; ++j) r[j] = to16<T16>(xs[j] - from16<T16>(fa[c][j]));
            }
            fal[c] = r;
          }
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < SK_CH; ++c) {
              acc[t][mt] = mfma16<T16>(fa[c], fw[t][c], acc[t][mt]);
              acc[t][mt] = mfma16<T16>(fal[c], fw[t][c], acc[t][mt]);
              acc[t][mt] = mfma16<T16>(fa[c], fwl[t][c], acc[t][mt]);
            }
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int c = 0; c < SK_CH; ++c) acc[t][mt] = mfma16<T16>(fa[c], fw[t][c], acc[t][mt]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wid][t][mt][lane][r] = acc[t][mt][r];
  __syncthreads();
  // wave w finishes m-tile w % 4 of column tile(s) w / 4, w / 4 + KW / 4, ...
  const int mt = wid & 3;
  if (mt >= mt_lo && mt < mt_n) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if ((t % (KW / 4)) != (wid >> 2)) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KW; ++w) v += red[w][t][mt][lane][r];
        const int row = mt * 16 + (lane >> 4) * 4 + r;
        epilogue_store<T16>(p, row, (ct0 + t) * 16 + (lane & 15), v);
      }
    }
  }
}
};

template <typename T16, bool A_F32, int SK_CH, int NT, int KW = 4, bool SPLIT = false>
__global__ __launch_bounds__(KW * 64) void gemm_skinny(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
  GemmSkinnyBody<T16, A_F32, SK_CH, NT, KW, SPLIT>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, sk_smem);
}

}  // namespace vog
