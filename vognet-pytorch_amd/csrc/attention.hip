// K1: rel_attention_fwd — softmax((Q K^T + bias)/sqrt(d_model)) V per
// (sequence, head), flash-style on gfx950, with the reference's relative
// position bias recomputed per tile instead of materialised.
//
// Replaces RelAttention.forward / Attention.forward (transformer_code.py:136-160,
// 42-50) and the whole of compute_pe (mdl_vog.py:456-490): the reference builds
// pe[S,N,N,H] = relu(Linear(5,H)(box_i - box_j)) (5.9 GB of temporaries at p100);
// the Linear is affine in the box, so pe[i,j,h] = relu(u[i,h] - u[j,h] + b[h])
// with u = W_pe . box — three VALU ops per (i,j), nothing stored.
//
// Mapping (wave64, v_mfma_f32_32x32x16):
//   * one wave owns 32 query rows; a workgroup of NW waves shares LDS K / V^T tiles
//     of 64 keys.
//   * "swapped" products so the softmax row lives in one lane:
//       S^T[key][q] = K_tile . Q^T        (A = K rows from LDS, B = Q from registers)
//       O^T[d][q]  += V^T_tile . P^T      (A = V^T rows from LDS, B = P from registers)
//     C/D layout has col = lane&31 = query for both, so the running max / sum /
//     rescale are per-lane scalars, and P goes from the S accumulator registers
//     straight into the B operand: the MFMA k index is a free summation index,
//     so register r of half hi is declared to be k = hi*8 + (r&7) of k-step r>>3
//     and the V^T fragment is read with the same key permutation
//     (key = 16*ks + 8*(j>>2) + 4*hi + (j&3)). No cross-lane movement, no LDS
//     round trip for P.
//   * V arrives already transposed ([S,H,dp,npad], written by the QKV GEMM
//     epilogue), so both LDS tiles are filled with 16-byte row chunks.
#include "common.h"

namespace vog {

struct AttnParams {
  const unsigned short* q; const unsigned short* k; const unsigned short* vt;
  unsigned short* out;
  const float* u; const float* pe_b;
  int S, N, H, dp, npad, use_rel, n_box, seq_per_vid, NP;
  float inv_scale;
};

constexpr int KT = 64;               // keys per LDS tile

template <typename T16, int NDB, int NW>
__global__ __launch_bounds__(NW * 64) void attn_kernel(AttnParams p) {
  constexpr int DP = NDB * 32;
  constexpr int KLD = DP + 8;        // K tile row pitch (halfwords): 16B-aligned, conflict-free b128
  constexpr int VLD = KT + 4;        // V^T tile row pitch: 8B-aligned, conflict-free b64
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* Ks = reinterpret_cast<unsigned short*>(smem);                 // [KT][KLD]
  unsigned short* Vs = Ks + KT * KLD;                                           // [DP][VLD]
  float* us = reinterpret_cast<float*>(Vs + DP * VLD);                          // [KT]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int hi = lane >> 5, ql = lane & 31;
  const int s = blockIdx.z, h = blockIdx.y;
  const int q0 = (blockIdx.x * NW + wid) * 32;
  const int qi = q0 + ql;
  const bool q_ok = qi < p.N;
  const int64_t sh = (int64_t)s * p.H + h;
  const unsigned short* Qg = p.q + sh * p.N * DP;
  const unsigned short* Kg = p.k + sh * p.N * DP;
  const unsigned short* Vg = p.vt + sh * DP * (int64_t)p.npad;

  // Q fragments (B operand of S^T): lane (q, hi) holds Q[q][ks*16 + hi*8 .. +8]
  u16x8 qf[DP / 16];
#pragma unroll
  for (int ks = 0; ks < DP / 16; ++ks) {
    u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    qf[ks] = q_ok ? *reinterpret_cast<const u16x8*>(Qg + (int64_t)qi * DP + ks * 16 + hi * 8) : z;
  }
  const int64_t u_base = p.use_rel
      ? ((int64_t)(s / p.seq_per_vid) * p.NP + (int64_t)(s % p.seq_per_vid) * p.n_box) : 0;
  float uq = 0.f, peb = 0.f;
  if (p.use_rel) {
    peb = p.pe_b[h];
    if (q_ok) uq = p.u[(u_base + (qi % p.n_box)) * p.H + h];
  }

  f32x16 o[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int ntiles = (p.N + KT - 1) / KT;
  for (int t = 0; t < ntiles; ++t) {
    const int kt0 = t * KT;
    __syncthreads();                      // previous tile fully consumed
    // ---- stage K tile: KT rows x DP halfwords, 16-byte chunks
    for (int c = tid; c < KT * (DP / 8); c += NW * 64) {
      const int row = c / (DP / 8), cc = c % (DP / 8);
      u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (kt0 + row < p.N) v = *reinterpret_cast<const u16x8*>(Kg + (int64_t)(kt0 + row) * DP + cc * 8);
      *reinterpret_cast<u16x8*>(&Ks[row * KLD + cc * 8]) = v;
    }
    // ---- stage V^T tile: DP rows x KT keys; keys >= N forced to zero
    for (int c = tid; c < DP * (KT / 8); c += NW * 64) {
      const int row = c / (KT / 8), cc = c % (KT / 8);
      const int key = kt0 + cc * 8;
      u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (key < p.N) {
        v = *reinterpret_cast<const u16x8*>(Vg + (int64_t)row * p.npad + key);
        if (key + 8 > p.N) {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (key + e >= p.N) v[e] = 0;
        }
      }
      u16x4 lo = {v[0], v[1], v[2], v[3]}, hi4 = {v[4], v[5], v[6], v[7]};
      *reinterpret_cast<u16x4*>(&Vs[row * VLD + cc * 8]) = lo;
      *reinterpret_cast<u16x4*>(&Vs[row * VLD + cc * 8 + 4]) = hi4;
    }
    if (p.use_rel) {
      for (int c = tid; c < KT; c += NW * 64) {
        const int key = kt0 + c;
        us[c] = key < p.N ? p.u[(u_base + (key % p.n_box)) * p.H + h] : 0.f;
      }
    }
    __syncthreads();

#pragma unroll
    for (int kb = 0; kb < KT / 32; ++kb) {
      if (kt0 + kb * 32 >= p.N) break;                 // wave-uniform
      // ---- S^T block [32 keys x 32 queries]
      f32x16 sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < DP / 16; ++ks) {
        const u16x8 kf = *reinterpret_cast<const u16x8*>(&Ks[(kb * 32 + ql) * KLD + ks * 16 + hi * 8]);
        sacc = mfma32<T16>(kf, qf[ks], sacc);
      }
      // ---- bias, scale, mask, block max
      float mloc = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = kb * 32 + c32_row(r, lane);     // key within tile
        float x = sacc[r];
        if (p.use_rel) x += fmaxf(uq - us[kl] + peb, 0.f);
        x *= p.inv_scale;
        x = (kt0 + kl < p.N) ? x : -1e30f;
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
      if (!__all(alpha == 1.0f)) {
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
      // ---- O^T += V^T_tile . P^T
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned short* vr = &Vs[(db * 32 + ql) * VLD + kb * 32 + ks * 16 + hi * 4];
          const u16x4 v0 = *reinterpret_cast<const u16x4*>(vr);
          const u16x4 v1 = *reinterpret_cast<const u16x4*>(vr + 8);
          const u16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          o[db] = mfma32<T16>(vf, pf[ks], o[db]);
        }
      }
    }
  }
  // ---- normalise and store: O^T[d][q] -> out[(s*N+q), h*DP + d], 4 consecutive d per store
  if (q_ok) {
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

template <typename T16, int NDB, int NW>
static int launch_attn(const AttnParams& p, hipStream_t st) {
  constexpr int DP = NDB * 32;
  const size_t lds = (size_t)KT * (DP + 8) * 2 + (size_t)DP * (KT + 4) * 2 + KT * 4;
  auto kern = attn_kernel<T16, NDB, NW>;
  static bool attr_set = false;       // benign race: idempotent
  if (!attr_set && lds > 48 * 1024) {
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(ceil_div(p.N, 32 * NW), p.H, p.S);
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}

template <typename T16, int NDB>
static int launch_attn_nw(const AttnParams& p, hipStream_t st) {
  if (p.N <= 32) return launch_attn<T16, NDB, 1>(p, st);
  if (p.N <= 64) return launch_attn<T16, NDB, 2>(p, st);
  return launch_attn<T16, NDB, 4>(p, st);
}

template <typename T16>
static int attn_dispatch(const AttnParams& p, hipStream_t st) {
  switch (p.dp) {
    case 32: return launch_attn_nw<T16, 1>(p, st);
    case 64: return launch_attn_nw<T16, 2>(p, st);
    case 128: return launch_attn_nw<T16, 4>(p, st);
    case 192: return launch_attn_nw<T16, 6>(p, st);
    case 256: return launch_attn_nw<T16, 8>(p, st);
    default: VOG_FAIL(-1, "rel_attention: unsupported padded head dim %d (32/64/128/192/256)", p.dp);
  }
}

int attn_head_pad(int dh) {
  const int opts[5] = {32, 64, 128, 192, 256};
  for (int i = 0; i < 5; ++i) if (dh <= opts[i]) return opts[i];
  return -1;
}

int attn_run(const vog_attn_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->q && a->k && a->vt && a->out16);
  VOG_CHECK_ARG(a->S > 0 && a->N > 0 && a->H > 0 && a->npad >= a->N && (a->npad % KT) == 0);
  VOG_CHECK_ARG(!a->use_rel || (a->u && a->pe_b && a->n_box > 0 && a->seq_per_vid > 0));
  AttnParams p{};
  p.q = (const unsigned short*)a->q; p.k = (const unsigned short*)a->k;
  p.vt = (const unsigned short*)a->vt; p.out = (unsigned short*)a->out16;
  p.u = a->u; p.pe_b = a->pe_b;
  p.S = a->S; p.N = a->N; p.H = a->H; p.dp = a->dp; p.npad = a->npad; p.use_rel = a->use_rel;
  p.n_box = a->n_box; p.seq_per_vid = a->seq_per_vid; p.NP = a->NP; p.inv_scale = a->inv_scale;
  VOG_DISPATCH_DTYPE(a->dtype, return attn_dispatch<T16>(p, st));
  return 0;
}

}  // namespace vog

extern "C" int vog_rel_attention_fwd(const vog_attn_args* a, void* stream) {
  return vog::attn_run(a, (hipStream_t)stream);
}
