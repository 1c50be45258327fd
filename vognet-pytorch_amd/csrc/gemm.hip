// K2: C = act(A * W^T + bias) (+residual)  — the nn.Linear of the reference
// (transformer_code.py:58-61,77-81,169-172; mdl_vog.py:182-188,202-207,224-230)
// as MFMA kernels for gfx950. Two shapes of kernel:
//
//  * gemm_tiled:  BMxBNx64 LDS-staged tiles, 4 waves (2x2), v_mfma_f32_32x32x16
//    with fp32 accumulators, register prefetch of the next K tile. Used when
//    M is large (token matrices: 800..80000 rows).
//  * gemm_skinny: M <= 64 (language path: B*T rows). Weight-streaming regime:
//    one workgroup owns 16 output columns, its 4 waves split K, every wave
//    keeps its slice of the W panel in registers (deep load queue, no LDS
//    round trip for an operand that is read exactly once) and loops over the
//    16-row tiles of A with v_mfma_f32_16x16x32; partial sums meet in LDS.
//
// Both read W as [N,K] row-major 16-bit (K contiguous), which is exactly the
// MFMA B-operand fragment order (8 consecutive k per lane): no transposes.
#include "common.h"

namespace vog {

enum { EPI_PLAIN = 0, EPI_QKV = 1 };

struct GemmParams {
  const void* a; const int32_t* a_rows; int64_t lda;
  const unsigned short* w; int64_t ldw;
  const float* bias; const float* residual; int64_t ldr;
  float* c32; unsigned short* c16; int64_t ldc; int64_t ldc16;
  int M, N, K; int relu; int rep; int c16_bf16;
  // QKV epilogue
  unsigned short* q; unsigned short* k; unsigned short* vt;
  int ntok, H, dp, npad;
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

template <typename T16>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, int row, int col, float v) {
  if (row >= p.M || col >= p.N) return;
  {
    if (p.bias) v += p.bias[col];
    if (p.residual) v += p.residual[(int64_t)row * p.ldr + col];
    if (p.relu) v = fmaxf(v, 0.f);
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
    const int64_t sh = (int64_t)s * p.H + h;
    const unsigned short o = to16<T16>(acc[r]);
    if (which == 0) {
      p.q[(sh * p.ntok + i) * p.dp + dd] = o;
    } else if (which == 1) {
      p.k[(sh * p.ntok + i) * p.dp + dd] = o;
    } else {
      p.vt[(sh * p.dp + dd) * p.npad + i] = o;
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

// ----------------------------------------------------------------------------
// skinny kernel (M <= 64, K % 32 == 0)
// ----------------------------------------------------------------------------
constexpr int SK_CH = 8;   // k-steps (of 32) per register chunk

template <typename T16, bool A_F32>
__global__ __launch_bounds__(256) void gemm_skinny(GemmParams p) {
  __shared__ float red[4][4][64][4];          // [wave][mtile][lane][reg]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n0 = blockIdx.x * 16;
  const int n = n0 + (lane & 15);
  const bool n_ok = n < p.N;
  const int kg = (lane >> 4) * 8;
  const int mt_n = (p.M + 15) / 16;           // <= 4
  const int ksteps = p.K / 32;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  int64_t a_off[4]; bool a_ok[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = mt * 16 + (lane & 15);
    a_ok[mt] = (mt < mt_n) && (m < p.M);
    const int64_t src = a_ok[mt] ? (p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m) : 0;
    a_off[mt] = src * p.lda;
  }
  const int64_t w_off = (int64_t)(n_ok ? n : 0) * p.ldw;

  // wave `wid` owns k-steps wid, wid+4, ... ; processed SK_CH at a time
  for (int base = wid; base < ksteps; base += 4 * SK_CH) {
    u16x8 fw[SK_CH];
#pragma unroll
    for (int c = 0; c < SK_CH; ++c) {
      const int ks = base + c * 4;
      fw[c] = load_a_chunk<T16, false>(p.w, w_off, ks * 32 + kg, n_ok && ks < ksteps);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt < mt_n) {
        u16x8 fa[SK_CH];
#pragma unroll
        for (int c = 0; c < SK_CH; ++c) {
          const int ks = base + c * 4;
          fa[c] = load_a_chunk<T16, A_F32>(p.a, a_off[mt], ks * 32 + kg, a_ok[mt] && ks < ksteps);
        }
#pragma unroll
        for (int c = 0; c < SK_CH; ++c) acc[mt] = mfma16<T16>(fa[c], fw[c], acc[mt]);
      }
    }
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][mt][lane][r] = acc[mt][r];
  __syncthreads();
  // wave w finishes m-tile w
  const int mt = wid;
  if (mt < mt_n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = red[0][mt][lane][r] + red[1][mt][lane][r] + red[2][mt][lane][r] + red[3][mt][lane][r];
      const int row = mt * 16 + (lane >> 4) * 4 + r;
      epilogue_store<T16>(p, row, n, v);
    }
  }
}

// ----------------------------------------------------------------------------
// host launchers
// ----------------------------------------------------------------------------
template <typename T16, bool A_F32, int EPI>
static int launch_tiled(const GemmParams& p, hipStream_t st) {
  const int64_t t128 = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 128);
  if (p.M > 64 && t128 >= 192) {
    dim3 grid(ceil_div(p.M, 128) * ceil_div(p.N, 128));
    hipLaunchKernelGGL((gemm_tiled<T16, 128, 128, A_F32, EPI>), grid, dim3(256), 0, st, p);
  } else {
    dim3 grid(ceil_div(p.M, 64) * ceil_div(p.N, 64));
    hipLaunchKernelGGL((gemm_tiled<T16, 64, 64, A_F32, EPI>), grid, dim3(256), 0, st, p);
  }
  VOG_LAUNCH_CHECK();
  return 0;
}

template <typename T16>
static int gemm_dispatch(const vog_gemm_args* g, hipStream_t st) {
  GemmParams p{};
  p.a = g->a; p.a_rows = g->a_rows; p.lda = g->lda;
  p.w = (const unsigned short*)g->w; p.ldw = g->ldw;
  p.bias = g->bias; p.residual = g->residual; p.ldr = g->ldr;
  p.c32 = g->c32; p.c16 = (unsigned short*)g->c16; p.ldc = g->ldc; p.ldc16 = g->ldc16;
  p.M = g->M; p.N = g->N; p.K = g->K; p.relu = g->relu; p.rep = g->rep < 1 ? 1 : g->rep;
  p.c16_bf16 = (g->c16_dtype < 0 ? (int)g->dtype : g->c16_dtype) == VOG_BF16;
  if (p.M <= 64 && (p.K % 32) == 0) {
    dim3 grid(ceil_div(p.N, 16));
    if (g->a_is_f32) hipLaunchKernelGGL((gemm_skinny<T16, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_skinny<T16, false>), grid, dim3(256), 0, st, p);
    VOG_LAUNCH_CHECK();
    return 0;
  }
  if (g->a_is_f32) return launch_tiled<T16, true, EPI_PLAIN>(p, st);
  return launch_tiled<T16, false, EPI_PLAIN>(p, st);
}

int gemm_run(const vog_gemm_args* g, hipStream_t st) {
  VOG_CHECK_ARG(g && g->a && g->w && (g->c32 || g->c16));
  VOG_CHECK_ARG(g->M > 0 && g->N > 0 && g->K > 0 && (g->K % 8) == 0);
  VOG_CHECK_ARG((g->lda % (g->a_is_f32 ? 4 : 8)) == 0 && (g->ldw % 8) == 0);
  VOG_CHECK_ARG(!(g->residual && g->rep > 1));
  VOG_DISPATCH_DTYPE(g->dtype, return gemm_dispatch<T16>(g, st));
  return 0;
}

int qkv_run(const vog_qkv_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->x16 && a->wqkv && a->q && a->k && a->vt);
  VOG_CHECK_ARG(a->K % 8 == 0 && a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->npad >= a->N);
  GemmParams p{};
  p.a = a->x16; p.lda = a->ldx; p.w = (const unsigned short*)a->wqkv; p.ldw = a->ldw;
  p.M = a->S * a->N; p.N = 3 * a->H * a->dp; p.K = a->K; p.rep = 1;
  p.q = (unsigned short*)a->q; p.k = (unsigned short*)a->k; p.vt = (unsigned short*)a->vt;
  p.ntok = a->N; p.H = a->H; p.dp = a->dp; p.npad = a->npad;
  VOG_DISPATCH_DTYPE(a->dtype, return (launch_tiled<T16, false, EPI_QKV>(p, st)));
  return 0;
}

}  // namespace vog

extern "C" int vog_gemm_bias_act(const vog_gemm_args* g, void* stream) {
  return vog::gemm_run(g, (hipStream_t)stream);
}
extern "C" int vog_qkv_proj(const vog_qkv_args* a, void* stream) {
  return vog::qkv_run(a, (hipStream_t)stream);
}
