// SURVEY.md 8(f)-4, first slice of the backward: from d loss / d mdl_outs (vog_loss_bwd) through the score head
// (lin2: code/mdl_vog.py:224-230) and the tail of the LAST mul_tx encoder layer - Wo, residual, LayerNorm,
// FFN, residual, LayerNorm (code/transformer_code.py:21-31, 73-81, 189-203) - to the gradients of every
// parameter on that path and of the tail's two inputs (the concatenated attention heads and the layer
// input). What the reference gets from autograd in Learner.train_epoch (utils/trn_utils.py:485-532).
//
// fp32 throughout, on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 products and sums, 1/16 of
// the 16-bit MFMA rate): this slice pins the MATH against autograd through the reference modules
// (tests/golden/bwd__*.npz); the 16-bit operand path of the forward kernels is the next step. The forward
// tail keeps its fp32 stream in registers and writes nothing back, so the backward recomputes the
// activations it needs from the tail's inputs (activation recomputation, fp32).
//
//   t = a Wo^T + x            x1 = LN1(t)
//   f = relu(x1 W1^T + b1)    u = f W2^T + b2 + x1      y = LN2(u)
//   h = relu(y Wl^T + bl)     logit = h . w2 + b2'      mdl_outs[v, arg, frame*nppf + p] = logit(row)
#include <map>
#include <mutex>
#include <string>
#include "common.h"

namespace vog {

// ---- C[M,N] = A . B (+ bias[n]) (relu), generic strides: A(m,k) = a[m*am + k*ak], B(k,n) = b[k*bk + n*bn] --------
// One of (am, ak) and one of (bk, bn) is 1 (row- or column-major operands; the launcher checks it), every
// dimension is a multiple of 4 and the pointers are 16-byte aligned: tiles are fetched with 16-byte loads
// along the unit stride, one chunk ahead of the MFMAs (registers), and parked in LDS as [k][m] / [k][n].
// 128 x 64 tile per workgroup, 4 waves of 64 x 32 (4 x 2 MFMA tiles of 16 x 16), K in chunks of 16.
struct GemmF32 {
  const float* a; int64_t am, ak; const float* b; int64_t bk, bn; float* c; int64_t ldc;
  const float* bias; int relu; int M, N, K;
  int64_t sa, sb, sc;       // batch strides (blockIdx.z)
  int accum;                // C += ...
  int scalar;               // any dimensions / alignments: element loads with per-element bounds checks
  int kchunk;               // split K: block z handles k in [z * kchunk, ...) (sa / sb advance a / b by a chunk, sc = one partial C)
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32 p) {
  constexpr int TM = 128, TN = 64, TK = 16;
  __shared__ __attribute__((aligned(16))) float As[TK][TM + 4];      // [k][m]
  __shared__ __attribute__((aligned(16))) float Bs[TK][TN + 4];      // [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 32;
  const float* pa = p.a + (int64_t)blockIdx.z * p.sa;
  const float* pb = p.b + (int64_t)blockIdx.z * p.sb;
  float* pc = p.c + (int64_t)blockIdx.z * p.sc;
  const int KL = p.kchunk ? (p.K - (int)blockIdx.z * p.kchunk < p.kchunk ? p.K - (int)blockIdx.z * p.kchunk : p.kchunk) : p.K;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool a_kfast = p.ak == 1, b_nfast = p.bn == 1;
  float4 ra[2], rb;
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * 256;
      int m, k;
      if (a_kfast) { m = idx >> 2; k = (idx & 3) * 4; } else { k = idx >> 5; m = (idx & 31) * 4; }
      const int gm = m0 + m, gk = k0 + k;
      if (!p.scalar) {
        ra[e] = (gm < p.M && gk < KL) ? *reinterpret_cast<const float4*>(pa + (int64_t)gm * p.am + (int64_t)gk * p.ak)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mm = a_kfast ? gm : gm + q, kk = a_kfast ? gk + q : gk;
          v[q] = (mm < p.M && kk < KL) ? pa[(int64_t)mm * p.am + (int64_t)kk * p.ak] : 0.f;
        }
        ra[e] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    int n, k;
    if (b_nfast) { k = tid >> 4; n = (tid & 15) * 4; } else { n = tid >> 2; k = (tid & 3) * 4; }
    const int gn = n0 + n, gk = k0 + k;
    if (!p.scalar) {
      rb = (gn < p.N && gk < KL) ? *reinterpret_cast<const float4*>(pb + (int64_t)gk * p.bk + (int64_t)gn * p.bn)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nn = b_nfast ? gn + q : gn, kk = b_nfast ? gk : gk + q;
        v[q] = (nn < p.N && kk < KL) ? pb[(int64_t)kk * p.bk + (int64_t)nn * p.bn] : 0.f;
      }
      rb = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * 256;
      if (a_kfast) {
        const int m = idx >> 2, k = (idx & 3) * 4;
        As[k][m] = ra[e].x; As[k + 1][m] = ra[e].y; As[k + 2][m] = ra[e].z; As[k + 3][m] = ra[e].w;
      } else {
        const int k = idx >> 5, m = (idx & 31) * 4;
        *reinterpret_cast<float4*>(&As[k][m]) = ra[e];
      }
    }
    if (b_nfast) {
      const int k = tid >> 4, n = (tid & 15) * 4;
      *reinterpret_cast<float4*>(&Bs[k][n]) = rb;
    } else {
      const int n = tid >> 2, k = (tid & 3) * 4;
      Bs[k][n] = rb.x; Bs[k + 1][n] = rb.y; Bs[k + 2][n] = rb.z; Bs[k + 3][n] = rb.w;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < KL; k0 += TK) {
    park();
    __syncthreads();
    if (k0 + TK < KL) fetch(k0 + TK);
#pragma unroll
    for (int ks = 0; ks < TK; ks += 4) {
      const int kq = ks + (lane >> 4);
      float fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = As[kq][wm + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = Bs[kq][wn + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + (lane >> 4) * 4 + r, n = n0 + wn + j * 16 + (lane & 15);
        if (m < p.M && n < p.N) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[n];
          if (p.relu) v = v < 0.f ? 0.f : v;
          float* dst = pc + (int64_t)m * p.ldc + n;
          *dst = p.accum ? *dst + v : v;
        }
      }
}

// ---- the same product with 16-bit operands (option `train_bf16`, off by default): A and B are rounded to bf16 on their way
// into LDS and multiplied with v_mfma_f32_16x16x32_bf16 (fp32 accumulation, 16 x the rate of the fp32 matrix instruction).
// Same tile, strides, batching, split-K and epilogue as gemm_f32_kernel (vector path only); K in chunks of 32.
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmF32 p) {
  constexpr int TM = 128, TN = 64, TK = 32, LP = TK + 8;              // LP: row pitch in halfwords (80 B: 16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) unsigned short As[TM][LP];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[TN][LP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int wm = (wid >> 1) * 64, wn = (wid & 1) * 32;
  const float* pa = p.a + (int64_t)blockIdx.z * p.sa;
  const float* pb = p.b + (int64_t)blockIdx.z * p.sb;
  float* pc = p.c + (int64_t)blockIdx.z * p.sc;
  const int KL = p.kchunk ? (p.K - (int)blockIdx.z * p.kchunk < p.kchunk ? p.K - (int)blockIdx.z * p.kchunk : p.kchunk) : p.K;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool a_kfast = p.ak == 1, b_nfast = p.bn == 1;
  float4 ra[4], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      int m, k;
      if (a_kfast) { m = idx >> 3; k = (idx & 7) * 4; } else { k = idx >> 5; m = (idx & 31) * 4; }
      const int gm = m0 + m, gk = k0 + k;
      ra[e] = (gm < p.M && gk < KL) ? *reinterpret_cast<const float4*>(pa + (int64_t)gm * p.am + (int64_t)gk * p.ak)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * 256;
      int n, k;
      if (b_nfast) { k = idx >> 4; n = (idx & 15) * 4; } else { n = idx >> 3; k = (idx & 7) * 4; }
      const int gn = n0 + n, gk = k0 + k;
      rb[e] = (gn < p.N && gk < KL) ? *reinterpret_cast<const float4*>(pb + (int64_t)gk * p.bk + (int64_t)gn * p.bn)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      const unsigned short h0 = to16<BF16>(ra[e].x), h1 = to16<BF16>(ra[e].y), h2 = to16<BF16>(ra[e].z), h3 = to16<BF16>(ra[e].w);
      if (a_kfast) {
        const int m = idx >> 3, k = (idx & 7) * 4;
        *reinterpret_cast<u16x4*>(&As[m][k]) = u16x4{h0, h1, h2, h3};
      } else {
        const int k = idx >> 5, m = (idx & 31) * 4;
        As[m][k] = h0; As[m + 1][k] = h1; As[m + 2][k] = h2; As[m + 3][k] = h3;
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * 256;
      const unsigned short h0 = to16<BF16>(rb[e].x), h1 = to16<BF16>(rb[e].y), h2 = to16<BF16>(rb[e].z), h3 = to16<BF16>(rb[e].w);
      if (b_nfast) {
        const int k = idx >> 4, n = (idx & 15) * 4;
        Bs[n][k] = h0; Bs[n + 1][k] = h1; Bs[n + 2][k] = h2; Bs[n + 3][k] = h3;
      } else {
        const int n = idx >> 3, k = (idx & 7) * 4;
        *reinterpret_cast<u16x4*>(&Bs[n][k]) = u16x4{h0, h1, h2, h3};
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < KL; k0 += TK) {
    park();
    __syncthreads();
    if (k0 + TK < KL) fetch(k0 + TK);
    const int r = lane & 15, g8 = (lane >> 4) * 8;
    u16x8 fa[4], fb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const u16x8*>(&As[wm + i * 16 + r][g8]);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const u16x8*>(&Bs[wn + j * 16 + r][g8]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<BF16>(fa[i], fb[j], acc[i][j]);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + (lane >> 4) * 4 + r, n = n0 + wn + j * 16 + (lane & 15);
        if (m < p.M && n < p.N) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[n];
          if (p.relu) v = v < 0.f ? 0.f : v;
          float* dst = pc + (int64_t)m * p.ldc + n;
          *dst = p.accum ? *dst + v : v;
        }
      }
}

static thread_local int g_train_bf16 = 0;   // vog_train_set_int("bf16_gemm", 1), per calling thread: 16-bit operands for the tile GEMMs of the training path

static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// ---- C[M <= 16, N] = A[M, K] . W[N, K]^T (+ bias, relu, accumulate): the recurrent products of the BiLSTM (M = number of
// sentences). A weight-stream kernel: the 128 x 64 tile GEMM above launches N / 64 workgroups that each walk all of K
// (262 us for dh = dG W_hh at cfg 2); here one WAVE owns CPW output columns, reads their weight rows with 16-byte
// loads (1 KiB per wave and instruction, CPW * K / 256 of them in flight) and the M activation rows from LDS.
template <int MT, int CPW>
__global__ __launch_bounds__(256) void skinny_f32_kernel(GemmF32 p) {
  constexpr int KC = 1024;
  __shared__ __attribute__((aligned(16))) float As[MT][KC];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n0 = (blockIdx.x * 4 + wid) * CPW;
  float acc[CPW][MT];
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[c][m] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += KC) {
    const int kc = p.K - k0 < KC ? p.K - k0 : KC;
    __syncthreads();
    for (int i = tid * 4; i < MT * KC; i += 256 * 4) {
      const int m = i / KC, k = i % KC;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < p.M && k < kc) v = *reinterpret_cast<const float4*>(p.a + (int64_t)m * p.am + k0 + k);
      *reinterpret_cast<float4*>(&As[m][k]) = v;
    }
    __syncthreads();
    for (int kk = lane * 4; kk < kc; kk += 256) {
      float4 w[CPW];
#pragma unroll
      for (int c = 0; c < CPW; ++c)
        w[c] = (n0 + c < p.N) ? *reinterpret_cast<const float4*>(p.b + (int64_t)(n0 + c) * p.bn + k0 + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 a = *reinterpret_cast<const float4*>(&As[m][kk]);
#pragma unroll
        for (int c = 0; c < CPW; ++c) acc[c][m] += a.x * w[c].x + a.y * w[c].y + a.z * w[c].z + a.w * w[c].w;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPW; ++c)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v = acc[c][m];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0 && m < p.M && n0 + c < p.N) {
        if (p.bias) v += p.bias[n0 + c];
        if (p.relu) v = v < 0.f ? 0.f : v;
        float* dst = p.c + (int64_t)m * p.ldc + n0 + c;
        *dst = p.accum ? *dst + v : v;
      }
    }
}

// out[n, k] = in[k, n]  (W_hh^T for the backward recurrence, once per layer, direction and step() call)
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* in, float* out, int rows, int cols) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)
    if (by + j < rows && bx + tx < cols) t[j][tx] = in[(int64_t)(by + j) * cols + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < cols && by + tx < rows) out[(int64_t)(bx + j) * rows + by + tx] = t[tx][j];
}

// out = (accum ? out : 0) + sum_z part[z] (+ bias, relu), z in order: the second half of a split-K product
__global__ void splitk_reduce_kernel(const float* part, int S, float* c, int64_t ldc, const float* bias, int relu, int accum, int M, int N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  const int n = (int)(i % N);
  const int64_t m = i / N;
  float v = 0.f;
  for (int z = 0; z < S; ++z) v += part[(int64_t)z * M * N + i];
  if (bias) v += bias[n];
  if (relu) v = v < 0.f ? 0.f : v;
  float* dst = c + m * ldc + n;
  *dst = accum ? *dst + v : v;
}

// partial products of split-K launches: one buffer per stream, grown on demand (training path only; never under graph capture)
static float* splitk_scratch(hipStream_t st, size_t bytes) {
  static std::mutex mu;
  static std::map<hipStream_t, std::pair<float*, size_t>> bufs;
  std::lock_guard<std::mutex> g(mu);
  auto& b = bufs[st];
  if (b.second < bytes) {
    if (b.first) { (void)hipStreamSynchronize(st); (void)hipFree(b.first); }
    b.first = nullptr; b.second = 0;
    const size_t want = bytes < ((size_t)16 << 20) ? ((size_t)16 << 20) : bytes;
    if (hipMalloc(&b.first, want) != hipSuccess) return nullptr;
    b.second = want;
  }
  return b.first;
}

// batched form; the vector path is chosen when every dimension, stride and pointer allows 16-byte loads
static int gemm_f32_b(const float* a, int64_t am, int64_t ak, int64_t sa, const float* b, int64_t bk, int64_t bn, int64_t sb,
                      float* c, int64_t ldc, int64_t sc, const float* bias, int relu, int accum, int M, int N, int K, int batch,
                      hipStream_t st) {
  VOG_CHECK_ARG((am == 1 || ak == 1) && (bk == 1 || bn == 1) && M > 0 && N > 0 && K > 0 && batch > 0);
  auto m4 = [](int64_t v) { return (v & 3) == 0; };
  const bool vec = m4(M) && m4(N) && m4(K) && (m4(am) || am == 1) && (m4(ak) || ak == 1) && (m4(bk) || bk == 1) &&
                   (m4(bn) || bn == 1) && m4(sa) && m4(sb) && al16(a) && al16(b);
  GemmF32 p{a, am, ak, b, bk, bn, c, ldc, bias, relu, M, N, K, sa, sb, sc, accum, vec ? 0 : 1, 0};
  if (batch == 1 && M <= 16 && ak == 1 && bk == 1 && m4(K) && m4(am) && m4(bn) && al16(a) && al16(b) && N >= 256) {
    // few rows against a K-contiguous weight matrix: weight-stream kernel (one wave per 1 / 4 output columns)
    const bool wide = N >= 2048;
    const int cols_per_wg = 4 * (wide ? 4 : 1);
    const dim3 grid(ceil_div(N, cols_per_wg));
    if (M <= 4) { if (wide) ::vog::launch((skinny_f32_kernel<4, 4>), grid, dim3(256), 0, st, p); else ::vog::launch((skinny_f32_kernel<4, 1>), grid, dim3(256), 0, st, p); }
    else if (M <= 8) { if (wide) ::vog::launch((skinny_f32_kernel<8, 4>), grid, dim3(256), 0, st, p); else ::vog::launch((skinny_f32_kernel<8, 1>), grid, dim3(256), 0, st, p); }
    else { if (wide) ::vog::launch((skinny_f32_kernel<16, 2>), dim3(ceil_div(N, 8)), dim3(256), 0, st, p); else ::vog::launch((skinny_f32_kernel<16, 1>), grid, dim3(256), 0, st, p); }
    VOG_LAUNCH_CHECK();
    return 0;
  }
  const int tiles = ceil_div(N, 64) * ceil_div(M, 128);
  if (batch == 1 && tiles <= 96 && K >= 1024) {
    // few output tiles, long K (the weight gradients: K = rows of the batch): split K over the chip, fixed-order reduction
    int S = 256 / tiles; if (S > K / 256) S = K / 256; if (S > 32) S = 32;
    if (S >= 2) {
      const int kchunk = ceil_div(ceil_div(K, S), 16) * 16;
      S = ceil_div(K, kchunk);
      float* part = splitk_scratch(st, (size_t)S * M * N * 4);
      if (!part) VOG_FAIL(-3, "gemm_f32: no memory for %d split-K partials", S);
      GemmF32 q = p;
      q.c = part; q.ldc = N; q.bias = nullptr; q.relu = 0; q.accum = 0; q.kchunk = kchunk;
      q.sa = (int64_t)kchunk * ak; q.sb = (int64_t)kchunk * bk; q.sc = (int64_t)M * N;
      if (!(m4(q.sa) && m4(q.sb))) q.scalar = 1;
      if (g_train_bf16 && !q.scalar) ::vog::launch(gemm_bf16_kernel, dim3(ceil_div(N, 64), ceil_div(M, 128), S), dim3(256), 0, st, q);
      else ::vog::launch(gemm_f32_kernel, dim3(ceil_div(N, 64), ceil_div(M, 128), S), dim3(256), 0, st, q);
      ::vog::launch(splitk_reduce_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, st, (const float*)part, S, c, ldc,
                    bias, relu, accum, M, N);
      VOG_LAUNCH_CHECK();
      return 0;
    }
  }
  if (g_train_bf16 && !p.scalar) ::vog::launch(gemm_bf16_kernel, dim3(ceil_div(N, 64), ceil_div(M, 128), batch), dim3(256), 0, st, p);
  else ::vog::launch(gemm_f32_kernel, dim3(ceil_div(N, 64), ceil_div(M, 128), batch), dim3(256), 0, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}
static int gemm_f32(const float* a, int64_t am, int64_t ak, const float* b, int64_t bk, int64_t bn, float* c, int64_t ldc,
                    const float* bias, int relu, int M, int N, int K, hipStream_t st, int accum = 0) {
  return gemm_f32_b(a, am, ak, 0, b, bk, bn, 0, c, ldc, 0, bias, relu, accum, M, N, K, 1, st);
}

// ---- train-mode dropout: counter-based masks, recomputed wherever they are needed (forward, backward): element idx of
// site `site` is kept iff the top 32 bits of splitmix64(seed * 0x9E3779B97F4A7C15 + site * 0xBF58476D1CE4E5B9 + idx)
// are >= p * 2^32; kept elements are scaled by 1 / (1 - p). The CPU oracle restates it (oracle/vog_oracle.py::drop_mask).
struct Drop { unsigned long long seed; unsigned int site; unsigned int thr; float inv_keep; };   // thr == 0: off
__device__ __forceinline__ float drop_scale(const Drop& d, unsigned long long idx) {
  if (d.thr == 0) return 1.0f;
  unsigned long long z = d.seed * 0x9E3779B97F4A7C15ull + (unsigned long long)d.site * 0xBF58476D1CE4E5B9ull + idx;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned int)(z >> 32) >= d.thr ? d.inv_keep : 0.0f;
}
static Drop make_drop(float p, unsigned long long seed, int site) {
  Drop d{seed, (unsigned int)site, 0u, 1.0f};
  if (p > 0.f) {
    const double pd = (double)p;
    d.thr = (unsigned int)(unsigned long long)(pd * 4294967296.0);
    d.inv_keep = (float)(1.0 / (1.0 - pd));
  }
  return d;
}
// out[i] = in[i] * mask(i)
__global__ void mask_mul_kernel(const float* in, float* out, Drop d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * drop_scale(d, (unsigned long long)i);
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- row-wise pieces -----------------------------------------------------------------------------
// t = t + x (optional), stats[m] = (mean, rstd), y = (t - mean) * rstd * g + b      one wave per row
__global__ __launch_bounds__(256) void ln_fwd_kernel(float* t, const float* x, const float* g, const float* b, float* y,
                                                     float2* stats, int M, int d, Drop dr) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float* tr = t + (int64_t)row * d;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) {        // t = dropout(t) + x  (ResidualBlock: x + dropout(layer(x)))
    float v = tr[i];
    if (x) { v = v * drop_scale(dr, (unsigned long long)row * d + i) + x[(int64_t)row * d + i]; tr[i] = v; }
    s += v;
  }
  const float mean = wsum(s) / (float)d;
  float q = 0.f;
  for (int i = lane; i < d; i += 64) { const float c = tr[i] - mean; q += c * c; }
  const float rstd = 1.0f / sqrtf(wsum(q) / (float)d + 1e-5f);
  for (int i = lane; i < d; i += 64) y[(int64_t)row * d + i] = (tr[i] - mean) * rstd * g[i] + b[i];
  if (lane == 0) stats[row] = make_float2(mean, rstd);
}

// dt = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)), dxh = dy * g, xh = (t - mean) * rstd; also writes
// dyxh = dy * xh (the summand of d gamma). `add` (optional) is added to dy first (a second gradient path).
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const float* add, const float* t, const float2* stats,
                                                     const float* g, float* dt, float* dyxh, float* dysum, int M, int d) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float2 st = stats[row];
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < d; i += 64) {
    const int64_t o = (int64_t)row * d + i;
    const float dv = dy[o] + (add ? add[o] : 0.f);
    const float xh = (t[o] - st.x) * st.y, dxh = dv * g[i];
    s1 += dxh; s2 += dxh * xh;
    dyxh[o] = dv * xh;
    if (dysum) dysum[o] = dv;
  }
  s1 = wsum(s1) / (float)d; s2 = wsum(s2) / (float)d;
  for (int i = lane; i < d; i += 64) {
    const int64_t o = (int64_t)row * d + i;
    const float dv = dy[o] + (add ? add[o] : 0.f);
    const float xh = (t[o] - st.x) * st.y;
    dt[o] = st.y * (dv * g[i] - s1 - xh * s2);
  }
}

// out[n] = sum_m in[m, n] in two fixed-order levels (reproducible): CS_CHUNKS row chunks x 64-column blocks write
// partial sums, then one thread per column adds the chunks in order.
constexpr int CS_CHUNKS = 64;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* in, float* part, int M, int d) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n = blockIdx.x * 64 + lane;
  const int rows = (M + CS_CHUNKS - 1) / CS_CHUNKS, r0 = blockIdx.y * rows, r1 = min(M, r0 + rows);
  float s = 0.f;
  if (n < d) for (int m = r0 + wid; m < r1; m += 4) s += in[(int64_t)m * d + n];
  red[wid][lane] = s;
  __syncthreads();
  if (wid == 0 && n < d) part[(int64_t)blockIdx.y * d + n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ void colsum_final_kernel(const float* part, float* out, int d) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= d) return;
  float s = 0.f;
  for (int c = 0; c < CS_CHUNKS; ++c) s += part[(int64_t)c * d + n];
  out[n] = s;
}
static int colsum(const float* in, float* out, float* part, int M, int d, hipStream_t st) {
  ::vog::launch(colsum_partial_kernel, dim3(ceil_div(d, 64), CS_CHUNKS), dim3(256), 0, st, in, part, M, d);
  ::vog::launch(colsum_final_kernel, dim3(ceil_div(d, 256)), dim3(256), 0, st, (const float*)part, out, d);
  VOG_LAUNCH_CHECK();
  return 0;
}

// y = dy where pre > 0 else 0
__global__ void relu_bwd_kernel(const float* dy, const float* act, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = act[i] > 0.f ? dy[i] : 0.f;
}

// score head: dlog[m] gathered from d mdl_outs [n_vid, nsrl, nfrm*nppf] (row m = (s = (v, f), j = arg*nppf + p),
// the inverse regroup of code/mdl_vog.py:724-737), dh[m, :] = dlog * w2 (h > 0), hw[m, :] = dlog * h (summand of d w2)
struct ScoreBwd { const float* d_outs; const float* h; const float* w2; float* dh; float* hw; float* dlog; int M, nfrm, nppf, nsrl, dh_dim; };
__global__ __launch_bounds__(256) void score_bwd_kernel(ScoreBwd a) {
  const int row = blockIdx.x, N = a.nsrl * a.nppf;
  const int s = row / N, j = row % N, v = s / a.nfrm, f = s % a.nfrm, arg = j / a.nppf, pp = j % a.nppf;
  const float dl = a.d_outs[((int64_t)v * a.nsrl + arg) * ((int64_t)a.nfrm * a.nppf) + (int64_t)f * a.nppf + pp];
  if (threadIdx.x == 0) a.dlog[row] = dl;
  for (int i = threadIdx.x; i < a.dh_dim; i += blockDim.x) {
    const float hv = a.h[(int64_t)row * a.dh_dim + i];
    a.dh[(int64_t)row * a.dh_dim + i] = hv > 0.f ? dl * a.w2[i] : 0.f;
    a.hw[(int64_t)row * a.dh_dim + i] = dl * hv;
  }
}

// ---- attention (fp32) ---------------------------------------------------------------------------------
// normalised boxes (compute_pe code/mdl_vog.py:456-463): bx[r] = (x1/w, y1/h, x2/w, y2/h, frame/nfrm_div), 8 floats per row
__global__ void norm_boxes_kernel(const float* props, int stride, float vw, float vh, float fdiv, float* bx, int rows) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* q = props + (int64_t)r * stride;
  float* o = bx + (int64_t)r * 8;
  o[0] = q[0] / vw; o[1] = q[1] / vh; o[2] = q[2] / vw; o[3] = q[3] / vh; o[4] = q[4] / fdiv; o[5] = o[6] = o[7] = 0.f;
}

struct AttnRow {
  float* P;             // [S, N, N] logits in, probabilities out (softmax) | probabilities (ds)
  float* D;             // [S, N, N] dP in, d logits (before the 1/scale of Q K^T + bias) out (ds)
  const float* bx;      // [S*n, 8] or null (no relative-position bias)
  const float* w; const float* b;  // this head's Linear(5, H) row and bias entry (code/mdl_vog.py:446-451), device
  float* part;          // [S*N, 8] per-row sums of d bias . (box difference, 1) (ds)
  int S, N, n; float inv_scale;
  Drop dr; int h, H;    // dropout on the probabilities: element ((s*H + h)*N + i)*N + j of the layer's site
};

__device__ __forceinline__ float box_z(const AttnRow& a, const float* bi, const float* bj) {
  return a.w[0] * (bi[0] - bj[0]) + a.w[1] * (bi[1] - bj[1]) + a.w[2] * (bi[2] - bj[2]) + a.w[3] * (bi[3] - bj[3]) +
         a.w[4] * (bi[4] - bj[4]) + a.b[0];
}

// P[row] = softmax((P[row] + relu(w . (box_i - box_j) + b)) / scale): one wave per row (transformer_code.py:136-162)
__global__ __launch_bounds__(256) void attn_softmax_kernel(AttnRow a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.S * a.N) return;
  const int s = row / a.N, i = row % a.N;
  float* pr = a.P + (int64_t)row * a.N;
  const float* bi = a.bx ? a.bx + ((int64_t)s * a.n + i % a.n) * 8 : nullptr;
  float mx = -3.0e38f;
  for (int j = lane; j < a.N; j += 64) {
    float v = pr[j];
    if (bi) v += fmaxf(box_z(a, bi, a.bx + ((int64_t)s * a.n + j % a.n) * 8), 0.f);
    v *= a.inv_scale;
    pr[j] = v;
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int j = lane; j < a.N; j += 64) { const float e = expf(pr[j] - mx); pr[j] = e; sum += e; }
  sum = wsum(sum);
  const float inv = 1.0f / sum;
  float* dr_ = a.D + (int64_t)row * a.N;
  const unsigned long long base = (((unsigned long long)s * a.H + a.h) * a.N + i) * a.N;
  for (int j = lane; j < a.N; j += 64) {
    const float pv = pr[j] * inv;
    pr[j] = pv;
    if (a.dr.thr) dr_[j] = pv * drop_scale(a.dr, base + j);            // D := dropout(P), the operand of P V and of dV
  }
}

// D[row] = P (dP - sum_j P dP) / scale  (= d (Q K^T) = d bias); part[row] = sum_j [z > 0] D (box_i - box_j, 1)
__global__ __launch_bounds__(256) void attn_ds_kernel(AttnRow a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.S * a.N) return;
  const int s = row / a.N, i = row % a.N;
  const float* pr = a.P + (int64_t)row * a.N;
  float* dr = a.D + (int64_t)row * a.N;
  float rs = 0.f;
  const unsigned long long base = (((unsigned long long)s * a.H + a.h) * a.N + i) * a.N;
  if (a.dr.thr) for (int j = lane; j < a.N; j += 64) dr[j] *= drop_scale(a.dr, base + j);   // d P = d dropout(P) * mask (own elements)
  for (int j = lane; j < a.N; j += 64) rs += pr[j] * dr[j];
  rs = wsum(rs);
  const float* bi = a.bx ? a.bx + ((int64_t)s * a.n + i % a.n) * 8 : nullptr;
  float g[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = lane; j < a.N; j += 64) {
    const float ds = pr[j] * (dr[j] - rs) * a.inv_scale;
    dr[j] = ds;
    if (bi) {
      const float* bj = a.bx + ((int64_t)s * a.n + j % a.n) * 8;
      if (box_z(a, bi, bj) > 0.f) {
#pragma unroll
        for (int c = 0; c < 5; ++c) g[c] += ds * (bi[c] - bj[c]);
        g[5] += ds;
      }
    }
  }
  if (a.part) {
#pragma unroll
    for (int c = 0; c < 6; ++c) g[c] = wsum(g[c]);
    if (lane < 8) a.part[(int64_t)row * 8 + lane] = lane == 0 ? g[0] : lane == 1 ? g[1] : lane == 2 ? g[2] : lane == 3 ? g[3]
                                                    : lane == 4 ? g[4] : lane == 5 ? g[5] : 0.f;
  }
}

__global__ void pe_grad_store_kernel(const float* sum8, float* g_w, float* g_b, int h) {
  if (threadIdx.x < 5) g_w[h * 5 + threadIdx.x] = sum8[threadIdx.x];
  if (threadIdx.x == 5) g_b[h] = sum8[5];
}

// ---- seam between mul_tx's input and (obj_tx output, argument vectors): the backward of concate_vis_lang_feats +
// the (frame, argument) regroup (code/mdl_vog.py:316-344, 681-744). d_x rows ((bv, f), (arg, p)) x [dobj | dlang].
//   d_ps[(bv, f*nppf + p), c]  = sum_arg d_x[.., c]                         (every argument saw the same visual row)
//   d_lang[(b, v | 0, arg), c] = mask * sum_{(v,) f, p} d_x[.., dobj + c]   (every proposal saw the same argument vector)
struct ConcBwd { const float* dx; float* d_ps; float* d_lang; const int64_t* mask; int n_q, nc_v, nfrm, nppf, nsrl, dobj, dlang, lang_per_vid; };
__global__ void conc_bwd_ps_kernel(ConcBwd a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.n_q * a.nc_v * a.nfrm * a.nppf * a.dobj;
  if (i >= total) return;
  const int c = (int)(i % a.dobj);
  const int64_t r = i / a.dobj;                              // (bv, f, p)
  const int pp = (int)(r % a.nppf);
  const int64_t sf = r / a.nppf;                             // sequence (bv, f)
  const int vld = a.dobj + a.dlang;
  float s = 0.f;
  for (int ar = 0; ar < a.nsrl; ++ar) s += a.dx[((sf * a.nsrl + ar) * a.nppf + pp) * vld + c];
  a.d_ps[i] = s;
}
__global__ void conc_bwd_lang_kernel(ConcBwd a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nvl = a.lang_per_vid ? a.nc_v : 1;
  const int64_t total = (int64_t)a.n_q * nvl * a.nsrl * a.dlang;
  if (i >= total) return;
  const int c = (int)(i % a.dlang);
  const int64_t r = i / a.dlang;
  const int ar = (int)(r % a.nsrl);
  const int64_t bv = r / a.nsrl;                             // (b, v) or b
  const int vld = a.dobj + a.dlang;
  float s = 0.f;
  const int v0 = a.lang_per_vid ? 0 : 0, v1 = a.lang_per_vid ? 1 : a.nc_v;
  for (int v = v0; v < v1; ++v) {
    const int64_t q = a.lang_per_vid ? bv : bv * a.nc_v + v;
    for (int f = 0; f < a.nfrm; ++f)
      for (int pp = 0; pp < a.nppf; ++pp)
        s += a.dx[(((q * a.nfrm + f) * a.nsrl + ar) * a.nppf + pp) * vld + a.dobj + c];
  }
  a.d_lang[i] = a.mask && a.mask[r] == 0 ? 0.f : s;
}

// ---- Linear (+ ReLU): y = act(x W^T + b), rows optionally replicated downstream (segment rows: concat_prop_seg_feats)
// dpre[m, n] = [y > 0] * sum_{j < rep} dy[(m*rep + j) * ldy + n]
__global__ void lin_dpre_kernel(const float* dy, int64_t ldy, int rep, const float* y, int relu, float* dpre, int M, int N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  const int n = (int)(i % N);
  const int64_t m = i / N;
  float s = 0.f;
  for (int j = 0; j < rep; ++j) s += dy[(m * rep + j) * ldy + n];
  dpre[i] = (!relu || y[i] > 0.f) ? s : 0.f;
}

__global__ void add_kernel(const float* a, const float* b, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

}  // namespace vog

using namespace vog;

extern "C" int64_t vog_mul_tail_bwd_scratch_bytes(int M, int d, int dh, int dhead) {
  if (M <= 0 || d <= 0 || dh <= 0 || dhead < 0) return -1;
  // t, x1, u, y, dy, du, dx1, dt, tmp_d (9 x [M,d]); pre1, f?, dpre1 (3 x [M,dh]); h, dh_, hw (3 x [M,dhead]); stats, dlog
  const int64_t wmax = d > dhead ? d : dhead;
  return ((int64_t)9 * M * d + (int64_t)3 * M * dh + (int64_t)3 * M * dhead + (int64_t)5 * M + CS_CHUNKS * wmax + 1024) * 4;
}

extern "C" int vog_mul_tail_bwd(const vog_tail_bwd_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->attn && a->x && a->scratch && a->M > 0 && a->d > 0 && a->dh > 0);
  VOG_CHECK_ARG(a->wo && a->ln1g && a->ln1b && a->w1 && a->b1 && a->w2 && a->b2 && a->ln2g && a->ln2b);
  // no_head: a layer that is not followed by the score head - its output gradient d_y comes from the caller; without
  // d_y only the forward is recomputed (y_out = the layer's output)
  const bool head = !a->no_head;
  const bool fwd_only = a->no_head && !a->d_y;
  if (fwd_only) VOG_CHECK_ARG(a->y_out);
  if (head) {
    VOG_CHECK_ARG(a->d_mdl_outs && a->dhead > 0 && a->wl && a->bl && a->wl2);
    VOG_CHECK_ARG(a->M == a->n_vid * a->nfrm * a->nsrl * a->nppf);
  }
  if ((int64_t)a->scratch_bytes < vog_mul_tail_bwd_scratch_bytes(a->M, a->d, a->dh, head ? a->dhead : 0))
    VOG_FAIL(-2, "vog_mul_tail_bwd: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const int M = a->M, d = a->d, H1 = a->dh, HD = head ? a->dhead : 0;
  float* s = (float*)a->scratch;
  auto take = [&](int64_t n) { float* r = s; s += n; return r; };
  float *t = take((int64_t)M * d), *x1 = take((int64_t)M * d), *u = take((int64_t)M * d), *y = take((int64_t)M * d);
  float *dy = take((int64_t)M * d), *du = take((int64_t)M * d), *dx1 = take((int64_t)M * d), *dt = take((int64_t)M * d);
  float* tmp = take((int64_t)M * d);
  float *pre1 = take((int64_t)M * H1), *dpre1 = take((int64_t)M * H1), *df = take((int64_t)M * H1);
  float *h = take((int64_t)M * HD), *dh = take((int64_t)M * HD), *hw = take((int64_t)M * HD);
  float2* st1 = (float2*)take((int64_t)2 * M); float2* st2 = (float2*)take((int64_t)2 * M);
  float* dlog = take((M + 3) / 4 * 4);
  float* part = take((int64_t)CS_CHUNKS * (d > HD ? d : HD));
  const int64_t nd = (int64_t)M * d, n1 = (int64_t)M * H1;
  auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  // ---- recompute the forward in fp32
  VOG_TRY(gemm_f32(a->attn, d, 1, a->wo, 1, d, t, d, nullptr, 0, M, d, d, st));                    // a Wo^T   (Wo [d_out, d_in])
  const Drop dr1 = make_drop(a->drop_p, a->drop_seed, a->drop_site + 1), dr2 = make_drop(a->drop_p, a->drop_seed, a->drop_site + 2);
  ::vog::launch(ln_fwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, t, a->x, a->ln1g, a->ln1b, x1, st1, M, d, dr1);   // t = drop(t) + x
  VOG_TRY(gemm_f32(x1, d, 1, a->w1, 1, d, pre1, H1, a->b1, 0, M, H1, d, st));                       // pre1 (relu applied on use)
  ::vog::launch(relu_bwd_kernel, blocks(n1), dim3(256), 0, st, pre1, pre1, df, n1);                  // df = relu(pre1) (reused as f)
  VOG_TRY(gemm_f32(df, H1, 1, a->w2, 1, H1, u, d, a->b2, 0, M, d, H1, st));                          // f W2^T + b2
  ::vog::launch(ln_fwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, u, (const float*)x1, a->ln2g, a->ln2b, y, st2, M, d, dr2);   // u = drop(u) + x1
  if (a->y_out) VOG_HIP(hipMemcpyAsync(a->y_out, y, (size_t)nd * 4, hipMemcpyDeviceToDevice, st));   // the layer's output
  if (fwd_only) { VOG_LAUNCH_CHECK(); return 0; }
  const float* dyp = a->d_y;
  if (head) {
    VOG_TRY(gemm_f32(y, d, 1, a->wl, 1, d, h, HD, a->bl, 1, M, HD, d, st));                          // h = relu(y Wl^T + bl)
    // ---- score head
    ScoreBwd sb{a->d_mdl_outs, h, a->wl2, dh, hw, dlog, M, a->nfrm, a->nppf, a->nsrl, HD};
    ::vog::launch(score_bwd_kernel, dim3(M), dim3(256), 0, st, sb);
    VOG_TRY(colsum(hw, a->g_wl2, part, M, HD, st));   // d lin2.2.weight
    VOG_TRY(colsum(dlog, a->g_bl2, part, M, 1, st));                 // d lin2.2.bias
    VOG_TRY(gemm_f32(dh, 1, HD, y, d, 1, a->g_wl, d, nullptr, 0, HD, d, M, st));                     // d lin2.0.weight = dh^T y
    VOG_TRY(colsum(dh, a->g_bl, part, M, HD, st));
    VOG_TRY(gemm_f32(dh, HD, 1, a->wl, d, 1, dy, d, nullptr, 0, M, d, HD, st));                      // dy = dh Wl
    dyp = dy;
  }
  // ---- LayerNorm 2
  ::vog::launch(ln_bwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, dyp, (const float*)nullptr, (const float*)u,
                (const float2*)st2, a->ln2g, du, tmp, (float*)nullptr, M, d);
  VOG_TRY(colsum(tmp, a->g_ln2g, part, M, d, st));
  VOG_TRY(colsum(dyp, a->g_ln2b, part, M, d, st));
  // ---- FFN (dum = the gradient behind the feed-forward sub-layer's dropout; du itself is the residual path)
  const float* dum = du;
  if (dr2.thr) { ::vog::launch(mask_mul_kernel, blocks(nd), dim3(256), 0, st, (const float*)du, dy, dr2, nd); dum = dy; }
  VOG_TRY(gemm_f32(dum, 1, d, df, H1, 1, a->g_w2, H1, nullptr, 0, d, H1, M, st));                    // d W2 = du^T f
  VOG_TRY(colsum(dum, a->g_b2, part, M, d, st));
  VOG_TRY(gemm_f32(dum, d, 1, a->w2, H1, 1, dpre1, H1, nullptr, 0, M, H1, d, st));                   // df = du W2
  ::vog::launch(relu_bwd_kernel, blocks(n1), dim3(256), 0, st, (const float*)dpre1, (const float*)pre1, dpre1, n1);
  VOG_TRY(gemm_f32(dpre1, 1, H1, x1, d, 1, a->g_w1, d, nullptr, 0, H1, d, M, st));                   // d W1 = dpre1^T x1
  VOG_TRY(colsum(dpre1, a->g_b1, part, M, H1, st));
  VOG_TRY(gemm_f32(dpre1, H1, 1, a->w1, d, 1, dx1, d, nullptr, 0, M, d, H1, st));                    // dpre1 W1
  // ---- LayerNorm 1 (dx1 = du + dpre1 W1)
  ::vog::launch(ln_bwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, (const float*)dx1, (const float*)du, (const float*)t,
                (const float2*)st1, a->ln1g, dt, tmp, dy, M, d);                                       // dy := dx1 + du (summand of d beta)
  VOG_TRY(colsum(tmp, a->g_ln1g, part, M, d, st));
  VOG_TRY(colsum(dy, a->g_ln1b, part, M, d, st));
  // ---- Wo and the two inputs (dtm = the gradient behind the attention sub-layer's dropout)
  const float* dtm = dt;
  if (dr1.thr) { ::vog::launch(mask_mul_kernel, blocks(nd), dim3(256), 0, st, (const float*)dt, tmp, dr1, nd); dtm = tmp; }
  VOG_TRY(gemm_f32(dtm, 1, d, a->attn, d, 1, a->g_wo, d, nullptr, 0, d, d, M, st));                  // d Wo = dt^T a
  if (a->d_attn) VOG_TRY(gemm_f32(dtm, d, 1, a->wo, d, 1, a->d_attn, d, nullptr, 0, M, d, d, st));   // da = dt Wo
  if (a->d_x) VOG_HIP(hipMemcpyAsync(a->d_x, dt, (size_t)nd * 4, hipMemcpyDeviceToDevice, st));      // dx = dt
  VOG_LAUNCH_CHECK();
  return 0;
}

// ---- attention + Q/K/V projections of one (Rel)EncoderLayer in fp32: forward (concatenated heads) and backward ----
extern "C" int64_t vog_attn_f32_scratch_bytes(int S, int N, int n, int d) {
  if (S <= 0 || N <= 0 || n <= 0 || d <= 0) return -1;
  const int64_t M = (int64_t)S * N;
  return (6 * M * d + 2 * M * N + (int64_t)S * n * 8 + M * 8 + CS_CHUNKS * 8 + 64) * 4 + 256;
}

extern "C" int vog_attn_f32(const vog_attn_f32_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->x && a->wq && a->wk && a->wv && a->scratch && a->S > 0 && a->N > 0 && a->n > 0 && a->d > 0 && a->n_heads > 0);
  VOG_CHECK_ARG((a->N % a->n) == 0 && a->n_heads <= a->d);
  const bool bwd = a->d_cat != nullptr, rel = a->props != nullptr;
  if (rel) VOG_CHECK_ARG(a->pe_w && a->pe_b && a->prop_stride >= 5 && a->vid_w > 0.f && a->vid_h > 0.f && a->nfrm_div > 0.f);
  if (bwd) VOG_CHECK_ARG(a->g_wq && a->g_wk && a->g_wv && a->d_x && (!rel || (a->g_pe_w && a->g_pe_b)));
  if (!bwd) VOG_CHECK_ARG(a->cat_out);
  if ((int64_t)a->scratch_bytes < vog_attn_f32_scratch_bytes(a->S, a->N, a->n, a->d)) VOG_FAIL(-2, "vog_attn_f32: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const int S = a->S, N = a->N, n = a->n, d = a->d, H = a->n_heads, M = S * N;
  float* s = (float*)(((uintptr_t)a->scratch + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t cnt) { float* r = s; s += (cnt + 3) / 4 * 4; return r; };
  const int64_t md = (int64_t)M * d, nn = (int64_t)N * N;
  float *q = take(md), *k = take(md), *v = take(md), *dq = take(md), *dk = take(md), *dv = take(md);
  float *P = take((int64_t)S * nn), *D = take((int64_t)S * nn);
  float* bx = take((int64_t)S * n * 8);
  float* part = take((int64_t)M * 8);
  float* cpart = take((int64_t)CS_CHUNKS * 8);
  float* sum8 = take(8);
  // q, k, v = x W^T  (transformer_code.py:136-150; no bias)
  VOG_TRY(gemm_f32(a->x, d, 1, a->wq, 1, d, q, d, nullptr, 0, M, d, d, st));
  VOG_TRY(gemm_f32(a->x, d, 1, a->wk, 1, d, k, d, nullptr, 0, M, d, d, st));
  VOG_TRY(gemm_f32(a->x, d, 1, a->wv, 1, d, v, d, nullptr, 0, M, d, d, st));
  if (rel)
    ::vog::launch(norm_boxes_kernel, dim3(ceil_div(S * n, 256)), dim3(256), 0, st, a->props, a->prop_stride, a->vid_w, a->vid_h,
                  a->nfrm_div, bx, S * n);
  const Drop drp = make_drop(a->drop_p, a->drop_seed, a->drop_site);
  const int chunk = (d + H - 1) / H;                       // torch.chunk split sizes (transformer_code.py:66-67)
  const float inv_scale = 1.0f / sqrtf((float)d);          // scale = sqrt(d_model)
  const int64_t sx = (int64_t)N * d;
  for (int h = 0; h < H; ++h) {
    const int off = h * chunk, dh = (d - off) < chunk ? (d - off) : chunk;
    if (dh <= 0) VOG_FAIL(-1, "vog_attn_f32: %d heads do not split %d features", H, d);
    // logits = Q_h K_h^T, then the softmax with the box bias
    VOG_TRY(gemm_f32_b(q + off, d, 1, sx, k + off, 1, d, sx, P, N, nn, nullptr, 0, 0, N, N, dh, S, st));
    AttnRow ar{P, D, rel ? bx : nullptr, rel ? a->pe_w + h * 5 : nullptr, rel ? a->pe_b + h : nullptr, bwd && rel ? part : nullptr,
               S, N, n, inv_scale, drp, h, H};
    ::vog::launch(attn_softmax_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, ar);
    const float* Pd = drp.thr ? D : P;                                   // dropout(P) (D holds it until dP overwrites it)
    if (a->cat_out)
      VOG_TRY(gemm_f32_b(Pd, N, 1, nn, v + off, d, 1, sx, a->cat_out + off, d, sx, nullptr, 0, 0, N, dh, N, S, st));  // O_h = drop(P) V_h
    if (!bwd) continue;
    VOG_TRY(gemm_f32_b(Pd, 1, N, nn, a->d_cat + off, d, 1, sx, dv + off, d, sx, nullptr, 0, 0, N, dh, N, S, st));     // dV_h = drop(P)^T dO_h
    VOG_TRY(gemm_f32_b(a->d_cat + off, d, 1, sx, v + off, 1, d, sx, D, N, nn, nullptr, 0, 0, N, N, dh, S, st));       // d drop(P) = dO_h V_h^T
    ::vog::launch(attn_ds_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, ar);
    VOG_TRY(gemm_f32_b(D, N, 1, nn, k + off, d, 1, sx, dq + off, d, sx, nullptr, 0, 0, N, dh, N, S, st));             // dQ_h = dS K_h
    VOG_TRY(gemm_f32_b(D, 1, N, nn, q + off, d, 1, sx, dk + off, d, sx, nullptr, 0, 0, N, dh, N, S, st));             // dK_h = dS^T Q_h
    if (rel) {
      VOG_TRY(colsum(part, sum8, cpart, M, 8, st));
      ::vog::launch(pe_grad_store_kernel, dim3(1), dim3(64), 0, st, (const float*)sum8, a->g_pe_w, a->g_pe_b, h);
    }
  }
  if (bwd) {
    VOG_TRY(gemm_f32(dq, 1, d, a->x, d, 1, a->g_wq, d, nullptr, 0, d, d, M, st));                   // d Wq = dQ^T x
    VOG_TRY(gemm_f32(dk, 1, d, a->x, d, 1, a->g_wk, d, nullptr, 0, d, d, M, st));
    VOG_TRY(gemm_f32(dv, 1, d, a->x, d, 1, a->g_wv, d, nullptr, 0, d, d, M, st));
    VOG_TRY(gemm_f32(dq, d, 1, a->wq, d, 1, a->d_x, d, nullptr, 0, M, d, d, st, a->accumulate_dx ? 1 : 0));   // dx (+)= dQ Wq + dK Wk + dV Wv
    VOG_TRY(gemm_f32(dk, d, 1, a->wk, d, 1, a->d_x, d, nullptr, 0, M, d, d, st, 1));
    VOG_TRY(gemm_f32(dv, d, 1, a->wv, d, 1, a->d_x, d, nullptr, 0, M, d, d, st, 1));
  }
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_conc_f32_bwd(const float* d_x, float* d_ps, float* d_lang, const int64_t* inds_msk, int n_q, int nc_v, int nfrm,
                                int nppf, int nsrl, int dobj, int dlang, int lang_per_vid, void* stream) {
  VOG_CHECK_ARG(d_x && d_ps && n_q > 0 && nc_v > 0 && nfrm > 0 && nppf > 0 && nsrl > 0 && dobj > 0 && dlang >= 0);
  ConcBwd a{d_x, d_ps, d_lang, inds_msk, n_q, nc_v, nfrm, nppf, nsrl, dobj, dlang, lang_per_vid};
  const int64_t t1 = (int64_t)n_q * nc_v * nfrm * nppf * dobj;
  ::vog::launch(conc_bwd_ps_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  if (d_lang && dlang > 0) {
    const int64_t t2 = (int64_t)n_q * (lang_per_vid ? nc_v : 1) * nsrl * dlang;
    ::vog::launch(conc_bwd_lang_kernel, dim3((unsigned)((t2 + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  }
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t vog_linear_f32_scratch_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return -1;
  return ((int64_t)2 * M * N + (int64_t)CS_CHUNKS * N + 64) * 4;
}

extern "C" int vog_linear_f32(const vog_linear_f32_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->x && a->w && a->M > 0 && a->N > 0 && a->K > 0 && a->scratch);
  if ((int64_t)a->scratch_bytes < vog_linear_f32_scratch_bytes(a->M, a->N)) VOG_FAIL(-2, "vog_linear_f32: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const int M = a->M, N = a->N, K = a->K;
  float* y = a->y ? a->y : (float*)a->scratch;
  float* dpre = (float*)a->scratch + (int64_t)M * N;
  float* part = dpre + (int64_t)M * N;
  const int64_t ldx = a->ldx > 0 ? a->ldx : K;
  VOG_TRY(gemm_f32(a->x, ldx, 1, a->w, 1, K, y, N, a->b, a->relu, M, N, K, st));                     // y = act(x W^T + b)
  if (!a->dy) { VOG_LAUNCH_CHECK(); return 0; }
  VOG_CHECK_ARG(a->g_w && a->rep >= 1);
  const int64_t ldy = a->ldy > 0 ? a->ldy : N;
  ::vog::launch(lin_dpre_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, st, a->dy, ldy, a->rep,
                (const float*)y, a->relu, dpre, M, N);
  VOG_TRY(gemm_f32(dpre, 1, N, a->x, ldx, 1, a->g_w, K, nullptr, 0, N, K, M, st));                   // d W = dpre^T x
  if (a->g_b) VOG_TRY(colsum(dpre, a->g_b, part, M, N, st));
  if (a->d_x) VOG_TRY(gemm_f32(dpre, N, 1, a->w, K, 1, a->d_x, a->ldx > 0 ? a->ldx : K, nullptr, 0, M, K, N, st, a->accumulate_dx ? 1 : 0));
  VOG_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================================
// Language side in fp32: token re-index -> embedding -> packed multi-layer BiLSTM -> lstm_out_feat_proj -> argument
// vectors (code/mdl_vog.py:67-140, 250-283; utils/mdl_srl_utils.py:114-169), forward recomputation and backward
// (back-propagation through time with packed-sequence semantics: a sentence's state is frozen past its length).
// =====================================================================================================================
namespace vog {

__global__ void lang_tokens_kernel(const int64_t* words, const int64_t* mask, int64_t* tok, int Bn, int T, int wlen, int mlen, int V) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bn * T) return;
  const int bn = i / T, t = i % T;
  const int64_t m = mask[(int64_t)bn * mlen + t];
  tok[i] = m >= 0 ? words[(int64_t)bn * wlen + m] : V;
}
__global__ void embed_gather_kernel(const float* emb, const int64_t* tok, float* x, int rows, int E, Drop dr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * E) return;
  x[i] = emb[tok[i / E] * E + i % E] * drop_scale(dr, (unsigned long long)i);      // F.dropout(x, dropout_in) (mdl_srl_utils.py:128)
}
// g_emb[v, c] = sum over the token positions holding v (fixed order)
__global__ void embed_scatter_kernel(const float* dx, const int64_t* tok, float* g, int rows, int E, int nv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)nv * E) return;
  const int64_t v = i / E; const int c = (int)(i % E);
  float s = 0.f;
  for (int r = 0; r < rows; ++r) if (tok[r] == v) s += dx[(int64_t)r * E + c];
  g[i] = s;
}
__global__ void vec_add_kernel(const float* a, const float* b, float* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}
__global__ void argvec_gather_kernel(const float* full, const int64_t* cap, float* enc, int Bn, int nsrl, int T, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Bn * nsrl * 2 * D) return;
  const int c = (int)(i % (2 * D));
  const int64_t r = i / (2 * D);
  const int64_t bn = r / nsrl;
  const int64_t t = cap[r * 2 + (c >= D ? 1 : 0)];
  enc[i] = full[(bn * T + t) * D + (c >= D ? c - D : c)];
}
__global__ void argvec_scatter_kernel(const float* denc, const int64_t* cap, float* dfull, int Bn, int nsrl, int T, int D) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Bn * T * D) return;
  const int c = (int)(i % D);
  const int64_t r = i / D;
  const int64_t bn = r / T; const int t = (int)(r % T);
  float s = 0.f;
  for (int a = 0; a < nsrl; ++a) {
    const int64_t q = bn * nsrl + a;
    if (cap[q * 2] == t) s += denc[q * 2 * D + c];
    if (cap[q * 2 + 1] == t) s += denc[q * 2 * D + D + c];
  }
  dfull[i] = s;
}

struct LstmStep {
  const float* gpre;      // [Bn, 4R] h_{s-1} W_hh^T
  const float* xg;        // [Bn*T, 4R] x W_ih^T + b_ih + b_hh
  const int64_t* lens;
  float* gates;           // [T, Bn, 4R] activated (i, f, g, o)
  float* c;               // [T+1, Bn, R] slot s+1 = state after step s, slot 0 = 0
  float* h;               // [T+1, Bn, R]
  float* out;             // [Bn, T, 2R]
  int Bn, T, R, s, reverse;
  // backward
  const float* d_out;     // [Bn, T, 2R]
  const float* dh_cur; float* dh_nxt; float* dc;   // [Bn, R]
  float* dGs;             // [T, Bn, 4R] by step
  float* dGp;             // [Bn, T, 4R] by position (zero where a sentence has ended)
};
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void lstm_cell_fwd_kernel(LstmStep a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.Bn * a.R) return;
  const int bn = i / a.R, r = i % a.R, R = a.R;
  const int len = (int)a.lens[bn];
  const bool active = a.s < len;
  const int pos = a.reverse ? (len - 1 - a.s > 0 ? len - 1 - a.s : 0) : a.s;
  const int64_t st = ((int64_t)a.s * a.Bn + bn), st1 = ((int64_t)(a.s + 1) * a.Bn + bn);
  float* gt = a.gates + st * 4 * R;
  if (!active) {
    a.c[st1 * R + r] = a.c[st * R + r];
    a.h[st1 * R + r] = a.h[st * R + r];
    gt[r] = gt[R + r] = gt[2 * R + r] = gt[3 * R + r] = 0.f;
    return;
  }
  const float* gp = a.gpre + (int64_t)bn * 4 * R;
  const float* xg = a.xg + ((int64_t)bn * a.T + pos) * 4 * R;
  const float gi = sigm(gp[r] + xg[r]), gf = sigm(gp[R + r] + xg[R + r]);
  const float gg = tanhf(gp[2 * R + r] + xg[2 * R + r]), go = sigm(gp[3 * R + r] + xg[3 * R + r]);
  const float cn = gf * a.c[st * R + r] + gi * gg;
  const float hn = go * tanhf(cn);
  gt[r] = gi; gt[R + r] = gf; gt[2 * R + r] = gg; gt[3 * R + r] = go;
  a.c[st1 * R + r] = cn;
  a.h[st1 * R + r] = hn;
  a.out[((int64_t)bn * a.T + pos) * 2 * R + (a.reverse ? R : 0) + r] = hn;
}

__global__ void lstm_cell_bwd_kernel(LstmStep a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.Bn * a.R) return;
  const int bn = i / a.R, r = i % a.R, R = a.R;
  const int len = (int)a.lens[bn];
  const bool active = a.s < len;
  const int pos = a.reverse ? (len - 1 - a.s > 0 ? len - 1 - a.s : 0) : a.s;
  const int64_t st = ((int64_t)a.s * a.Bn + bn), st1 = ((int64_t)(a.s + 1) * a.Bn + bn);
  float* gs = a.dGs + st * 4 * R;
  if (!active) {
    a.dh_nxt[i] = a.dh_cur[i];                       // frozen state: the gradient passes through
    gs[r] = gs[R + r] = gs[2 * R + r] = gs[3 * R + r] = 0.f;
    return;
  }
  const float* gt = a.gates + st * 4 * R;
  const float gi = gt[r], gf = gt[R + r], gg = gt[2 * R + r], go = gt[3 * R + r];
  const float tc = tanhf(a.c[st1 * R + r]);
  const float dh = a.d_out[((int64_t)bn * a.T + pos) * 2 * R + (a.reverse ? R : 0) + r] + a.dh_cur[i];
  const float dc = a.dc[i] + dh * go * (1.f - tc * tc);
  const float d_i = dc * gg * gi * (1.f - gi), d_f = dc * a.c[st * R + r] * gf * (1.f - gf);
  const float d_g = dc * gi * (1.f - gg * gg), d_o = dh * tc * go * (1.f - go);
  a.dc[i] = dc * gf;
  a.dh_nxt[i] = 0.f;                                 // + dGs W_hh (accumulating GEMM behind this kernel)
  gs[r] = d_i; gs[R + r] = d_f; gs[2 * R + r] = d_g; gs[3 * R + r] = d_o;
  float* gp = a.dGp + ((int64_t)bn * a.T + pos) * 4 * R;
  gp[r] = d_i; gp[R + r] = d_f; gp[2 * R + r] = d_g; gp[3 * R + r] = d_o;
}

}  // namespace vog

namespace vog { __global__ void concat_rows_kernel(const float* a, int Na, int rep_a, const float* b, int Nb, int rep_b, float* out, int M); }

static int64_t lang_scratch_floats(int Bn, int T, int nsrl, int E, int R, int layers, int D, int L) {
  const int64_t BT = (int64_t)Bn * T;
  const int64_t kin_max = E > 2 * R ? E : 2 * R;
  int64_t n = 0;
  n += BT * 2 + 16;                                    // tokens (int64)
  n += BT * E;                                         // x0
  n += (int64_t)layers * BT * 2 * R;                   // layer outputs
  n += (int64_t)layers * 2 * (BT * 4 * R               // xg
                              + BT * 4 * R             // gates
                              + 2 * (int64_t)(T + 1) * Bn * R);   // c, h
  n += (int64_t)Bn * 4 * R;                            // gpre
  n += 2 * BT * 4 * R;                                 // dGs, dGp
  n += 3 * (int64_t)Bn * R;                            // dh x2, dc
  n += 4 * (int64_t)R;                                 // bias sum
  n += 2 * BT * kin_max;                               // d_x of a layer (two buffers: current layer's d_out / next)
  n += BT * D * 2;                                     // full, d_full
  n += (int64_t)Bn * nsrl * (2 * D) * 2;               // enc, d_enc
  n += (int64_t)Bn * nsrl * L * 2 + (int64_t)CS_CHUNKS * (4 * R > L ? 4 * R : L);
  n += BT * D * 2 + BT * 2 * R;                        // linear scratch (y, dpre) for the projection
  n += (int64_t)4 * R * R;                             // W_hh^T of the direction in flight
  return n + 4096;
}

extern "C" int64_t vog_lang_f32_scratch_bytes(int Bn, int T, int nsrl, int E, int R, int layers, int D, int L) {
  if (Bn <= 0 || T <= 0 || nsrl <= 0 || E <= 0 || R <= 0 || layers <= 0 || layers > 4 || D <= 0 || L <= 0) return -1;
  return lang_scratch_floats(Bn, T, nsrl, E, R, layers, D, L) * 4;
}

extern "C" int vog_lang_f32(const vog_lang_f32_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->words_ind && a->word_mask && a->lens && a->capture && a->emb && a->w_proj && a->b_proj && a->w_arg && a->b_arg);
  VOG_CHECK_ARG(a->Bn > 0 && a->T > 0 && a->nsrl > 0 && a->E > 0 && a->R > 0 && a->layers > 0 && a->layers <= 4 && a->D > 0 && a->L > 0);
  VOG_CHECK_ARG(a->T <= a->mask_len && a->scratch);
  if ((int64_t)a->scratch_bytes < vog_lang_f32_scratch_bytes(a->Bn, a->T, a->nsrl, a->E, a->R, a->layers, a->D, a->L))
    VOG_FAIL(-2, "vog_lang_f32: scratch too small");
  for (int l = 0; l < a->layers; ++l)
    for (int dr = 0; dr < 2; ++dr) VOG_CHECK_ARG(a->w_ih[l][dr] && a->w_hh[l][dr] && a->b_ih[l][dr] && a->b_hh[l][dr]);
  const bool bwd = a->d_lang_enc != nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int Bn = a->Bn, T = a->T, nsrl = a->nsrl, E = a->E, R = a->R, NL = a->layers, D = a->D, L = a->L;
  const int BT = Bn * T, G = 4 * R;
  float* s = (float*)(((uintptr_t)a->scratch + 255) & ~(uintptr_t)255);
  auto take = [&](int64_t cnt) { float* r = s; s += (cnt + 63) / 64 * 64; return r; };
  auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  int64_t* tok = (int64_t*)take((int64_t)BT * 2 + 16);
  float* x0 = take((int64_t)BT * E);
  float* lout[4]; float *xg[4][2], *gates[4][2], *cst[4][2], *hst[4][2];
  for (int l = 0; l < NL; ++l) {
    lout[l] = take((int64_t)BT * 2 * R);
    for (int dr = 0; dr < 2; ++dr) {
      xg[l][dr] = take((int64_t)BT * G); gates[l][dr] = take((int64_t)BT * G);
      cst[l][dr] = take((int64_t)(T + 1) * Bn * R); hst[l][dr] = take((int64_t)(T + 1) * Bn * R);
    }
  }
  float* gpre = take((int64_t)Bn * G);
  float *dGs = take((int64_t)BT * G), *dGp = take((int64_t)BT * G);
  float *dh0 = take((int64_t)Bn * R), *dh1 = take((int64_t)Bn * R), *dc = take((int64_t)Bn * R);
  float* bsum = take(G);
  const int kin_max = E > 2 * R ? E : 2 * R;
  float *dxa = take((int64_t)BT * kin_max), *dxb = take((int64_t)BT * kin_max);
  float *full = take((int64_t)BT * D), *dfull = take((int64_t)BT * D);
  float *enc = take((int64_t)Bn * nsrl * 2 * D), *denc = take((int64_t)Bn * nsrl * 2 * D);
  float *lenc = take((int64_t)Bn * nsrl * L), *dpre = take((int64_t)Bn * nsrl * L);
  float* part = take((int64_t)CS_CHUNKS * (G > L ? G : L));
  float* dpre2 = take((int64_t)BT * D);
  float* whh_t = take((int64_t)G * R);
  const Drop drop_emb = make_drop(a->drop_in, a->drop_seed, 1);
  // reuse_forward: `scratch` still holds the forward of an earlier call with the same inputs, weights and dropout seed
  // (the trainer's forward pass): the backward starts from it instead of recomputing 2 T recurrent products per layer
  if (!a->reuse_forward) {
  // ---- forward (re)computation
  ::vog::launch(lang_tokens_kernel, blocks(BT), dim3(256), 0, st, a->words_ind, a->word_mask, tok, Bn, T, a->words_len, a->mask_len,
                a->vocab_size);
  ::vog::launch(embed_gather_kernel, blocks((int64_t)BT * E), dim3(256), 0, st, a->emb, (const int64_t*)tok, x0, BT, E, drop_emb);
  for (int l = 0; l < NL; ++l) {
    const float* xin = l == 0 ? x0 : lout[l - 1];
    const int K = l == 0 ? E : 2 * R;
    VOG_HIP(hipMemsetAsync(lout[l], 0, (size_t)BT * 2 * R * 4, st));             // zero past each sentence's length
    for (int dr = 0; dr < 2; ++dr) {
      ::vog::launch(vec_add_kernel, blocks(G), dim3(256), 0, st, a->b_ih[l][dr], a->b_hh[l][dr], bsum, G);
      VOG_TRY(gemm_f32(xin, K, 1, a->w_ih[l][dr], 1, K, xg[l][dr], G, bsum, 0, BT, G, K, st));
      VOG_HIP(hipMemsetAsync(cst[l][dr], 0, (size_t)Bn * R * 4, st));
      VOG_HIP(hipMemsetAsync(hst[l][dr], 0, (size_t)Bn * R * 4, st));
      for (int t = 0; t < T; ++t) {
        VOG_TRY(gemm_f32(hst[l][dr] + (int64_t)t * Bn * R, R, 1, a->w_hh[l][dr], 1, R, gpre, G, nullptr, 0, Bn, G, R, st));
        LstmStep ls{}; ls.gpre = gpre; ls.xg = xg[l][dr]; ls.lens = a->lens; ls.gates = gates[l][dr]; ls.c = cst[l][dr];
        ls.h = hst[l][dr]; ls.out = lout[l]; ls.Bn = Bn; ls.T = T; ls.R = R; ls.s = t; ls.reverse = dr;
        ::vog::launch(lstm_cell_fwd_kernel, blocks((int64_t)Bn * R), dim3(256), 0, st, ls);
      }
    }
    // nn.LSTM's dropout between the layers / F.dropout on the encoder output (mdl_srl_utils.py:104, 150): in place - the
    // recurrence reads its own state buffers, only the next layer / the projection read this tensor
    const Drop dl = make_drop(a->drop_out, a->drop_seed, l < NL - 1 ? 2 + l : 10);
    if (dl.thr) ::vog::launch(mask_mul_kernel, blocks((int64_t)BT * 2 * R), dim3(256), 0, st, (const float*)lout[l], lout[l], dl, (int64_t)BT * 2 * R);
  }
  VOG_TRY(gemm_f32(lout[NL - 1], 2 * R, 1, a->w_proj, 1, 2 * R, full, D, a->b_proj, 1, BT, D, 2 * R, st));   // relu(x W^T + b), every step
  ::vog::launch(argvec_gather_kernel, blocks((int64_t)Bn * nsrl * 2 * D), dim3(256), 0, st, (const float*)full, a->capture, enc, Bn,
                nsrl, T, D);
  VOG_TRY(gemm_f32(enc, 2 * D, 1, a->w_arg, 1, 2 * D, lenc, L, a->b_arg, 1, Bn * nsrl, L, 2 * D, st));
  if (a->hid_out) {
    // final_hidden = [h_fwd after the last valid step | h_bwd after its last step (position 0)] of the top layer: the state
    // slot T (frozen past each sentence's length), then the same projection (code/mdl_vog.py:270-283)
    float* fin = dpre2;                                                      // [Bn, 2R] (free until the backward)
    VOG_CHECK_ARG((int64_t)Bn * 2 * R <= (int64_t)BT * D + (int64_t)G * R);   // dpre2 and whh_t are adjacent
    ::vog::launch(concat_rows_kernel, blocks((int64_t)Bn * 2 * R), dim3(256), 0, st, (const float*)(hst[NL - 1][0] + (int64_t)T * Bn * R), R, 1,
                  (const float*)(hst[NL - 1][1] + (int64_t)T * Bn * R), R, 1, fin, Bn);
    VOG_TRY(gemm_f32(fin, 2 * R, 1, a->w_proj, 1, 2 * R, a->hid_out, D, a->b_proj, 1, Bn, D, 2 * R, st));
  }
  if (a->lang_enc_out) VOG_HIP(hipMemcpyAsync(a->lang_enc_out, lenc, (size_t)Bn * nsrl * L * 4, hipMemcpyDeviceToDevice, st));
  if (a->full_out) VOG_HIP(hipMemcpyAsync(a->full_out, full, (size_t)BT * D * 4, hipMemcpyDeviceToDevice, st));
  }
  if (!bwd) { VOG_LAUNCH_CHECK(); return 0; }
  VOG_CHECK_ARG(a->g_emb && a->g_w_proj && a->g_b_proj && a->g_w_arg && a->g_b_arg);
  // ---- srl_arg_words_out_enc
  const int MA = Bn * nsrl;
  ::vog::launch(lin_dpre_kernel, blocks((int64_t)MA * L), dim3(256), 0, st, a->d_lang_enc, (int64_t)L, 1, (const float*)lenc, 1, dpre, MA, L);
  VOG_TRY(gemm_f32(dpre, 1, L, enc, 2 * D, 1, a->g_w_arg, 2 * D, nullptr, 0, L, 2 * D, MA, st));
  VOG_TRY(colsum(dpre, a->g_b_arg, part, MA, L, st));
  VOG_TRY(gemm_f32(dpre, L, 1, a->w_arg, 2 * D, 1, denc, 2 * D, nullptr, 0, MA, 2 * D, L, st));
  ::vog::launch(argvec_scatter_kernel, blocks((int64_t)BT * D), dim3(256), 0, st, (const float*)denc, a->capture, dfull, Bn, nsrl, T, D);
  // ---- lstm_out_feat_proj (the per-step call; the call on final_hidden feeds only the sep head)
  ::vog::launch(lin_dpre_kernel, blocks((int64_t)BT * D), dim3(256), 0, st, (const float*)dfull, (int64_t)D, 1, (const float*)full, 1, dpre2, BT, D);
  VOG_TRY(gemm_f32(dpre2, 1, D, lout[NL - 1], 2 * R, 1, a->g_w_proj, 2 * R, nullptr, 0, D, 2 * R, BT, st));
  VOG_TRY(colsum(dpre2, a->g_b_proj, part, BT, D, st));
  float* d_out = dxa; float* d_in = dxb;
  VOG_TRY(gemm_f32(dpre2, D, 1, a->w_proj, 2 * R, 1, d_out, 2 * R, nullptr, 0, BT, 2 * R, D, st));
  // ---- BiLSTM, top layer first
  for (int l = NL - 1; l >= 0; --l) {
    const float* xin = l == 0 ? x0 : lout[l - 1];
    const int K = l == 0 ? E : 2 * R;
    const Drop dl = make_drop(a->drop_out, a->drop_seed, l < NL - 1 ? 2 + l : 10);
    if (dl.thr) ::vog::launch(mask_mul_kernel, blocks((int64_t)BT * 2 * R), dim3(256), 0, st, (const float*)d_out, d_out, dl, (int64_t)BT * 2 * R);
    for (int dr = 0; dr < 2; ++dr) {
      VOG_CHECK_ARG(a->g_w_ih[l][dr] && a->g_w_hh[l][dr] && a->g_b_ih[l][dr] && a->g_b_hh[l][dr]);
      VOG_HIP(hipMemsetAsync(dGp, 0, (size_t)BT * G * 4, st));
      VOG_HIP(hipMemsetAsync(dh0, 0, (size_t)Bn * R * 4, st));
      VOG_HIP(hipMemsetAsync(dc, 0, (size_t)Bn * R * 4, st));
      float *cur = dh0, *nxt = dh1;
      ::vog::launch(transpose_f32_kernel, dim3(ceil_div(R, 32), ceil_div(G, 32)), dim3(256), 0, st, a->w_hh[l][dr], whh_t, G, R);   // [R, 4R]
      for (int t = T - 1; t >= 0; --t) {
        LstmStep ls{}; ls.lens = a->lens; ls.gates = gates[l][dr]; ls.c = cst[l][dr]; ls.h = hst[l][dr]; ls.Bn = Bn; ls.T = T; ls.R = R;
        ls.s = t; ls.reverse = dr; ls.d_out = d_out; ls.dh_cur = cur; ls.dh_nxt = nxt; ls.dc = dc; ls.dGs = dGs; ls.dGp = dGp;
        ::vog::launch(lstm_cell_bwd_kernel, blocks((int64_t)Bn * R), dim3(256), 0, st, ls);
        // dh_{s-1} += dG_s W_hh
        VOG_TRY(gemm_f32(dGs + (int64_t)t * Bn * G, G, 1, whh_t, 1, G, nxt, R, nullptr, 0, Bn, R, G, st, 1));
        float* sw = cur; cur = nxt; nxt = sw;
      }
      VOG_TRY(gemm_f32(dGs, 1, G, hst[l][dr], R, 1, a->g_w_hh[l][dr], R, nullptr, 0, G, R, BT, st));          // sum_s dG_s^T h_{s-1}
      VOG_TRY(gemm_f32(dGp, 1, G, xin, K, 1, a->g_w_ih[l][dr], K, nullptr, 0, G, K, BT, st));                   // dG^T x
      VOG_TRY(colsum(dGp, a->g_b_ih[l][dr], part, BT, G, st));
      VOG_HIP(hipMemcpyAsync(a->g_b_hh[l][dr], a->g_b_ih[l][dr], (size_t)G * 4, hipMemcpyDeviceToDevice, st));
      VOG_TRY(gemm_f32(dGp, G, 1, a->w_ih[l][dr], K, 1, d_in, K, nullptr, 0, BT, K, G, st, dr));               // d x (both directions)
    }
    float* sw = d_out; d_out = d_in; d_in = sw;
  }
  if (drop_emb.thr) ::vog::launch(mask_mul_kernel, blocks((int64_t)BT * E), dim3(256), 0, st, (const float*)d_out, d_out, drop_emb, (int64_t)BT * E);
  const int nv = a->vocab_size + 1;
  ::vog::launch(embed_scatter_kernel, blocks((int64_t)nv * E), dim3(256), 0, st, (const float*)d_out, (const int64_t*)tok, a->g_emb, BT, E, nv);
  VOG_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================================
// The remaining forward pieces of the fp32 training path (concatenations, score head) and the optimizer step
// =====================================================================================================================
namespace vog {

// out[m, :Na] = a[m / rep_a], out[m, Na:] = b[m / rep_b]        (concat_prop_seg_feats code/mdl_conc_single.py:51-66)
__global__ void concat_rows_kernel(const float* a, int Na, int rep_a, const float* b, int Nb, int rep_b, float* out, int M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = Na + Nb;
  if (i >= (int64_t)M * W) return;
  const int c = (int)(i % W);
  const int64_t m = i / W;
  out[i] = c < Na ? a[(m / rep_a) * Na + c] : b[(m / rep_b) * Nb + (c - Na)];
}

// x_mul[((q, v), f), (arg, p), :] = [ps[(q, v), f*nppf + p, :] | mask * lang[(q, v | 0), arg, :]]   (code/mdl_vog.py:316-344, 681-700)
__global__ void conc_fwd_kernel(const float* ps, const float* lang, const int64_t* mask, float* out, int n_q, int nc_v, int nfrm,
                                int nppf, int nsrl, int dobj, int dlang, int lang_per_vid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int vld = dobj + dlang;
  const int64_t total = (int64_t)n_q * nc_v * nfrm * nsrl * nppf * vld;
  if (i >= total) return;
  const int c = (int)(i % vld);
  int64_t r = i / vld;
  const int pp = (int)(r % nppf); r /= nppf;
  const int ar = (int)(r % nsrl); r /= nsrl;
  const int f = (int)(r % nfrm);
  const int64_t qv = r / nfrm;
  if (c < dobj) { out[i] = ps[((qv * nfrm + f) * nppf + pp) * dobj + c]; return; }
  const int64_t lr = (lang_per_vid ? qv : qv / nc_v) * nsrl + ar;
  out[i] = (mask && mask[lr] == 0) ? 0.f : lang[lr * dlang + (c - dobj)];
}

// mdl_outs[v, arg, f*nppf + p] = h[row] . w2 + b2, row = ((v, f), (arg, p))      (code/mdl_vog.py:224-230, 724-737)
__global__ __launch_bounds__(256) void score_fwd_kernel(const float* h, const float* w2, const float* b2, float* outs, int M, int HD,
                                                        int nfrm, int nppf, int nsrl) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float s = 0.f;
  for (int i = lane; i < HD; i += 64) s += h[(int64_t)row * HD + i] * w2[i];
  s = wsum(s);
  if (lane == 0) {
    const int N = nsrl * nppf, sq = row / N, j = row % N, v = sq / nfrm, f = sq % nfrm, ar = j / nppf, pp = j % nppf;
    outs[((int64_t)v * nsrl + ar) * ((int64_t)nfrm * nppf) + (int64_t)f * nppf + pp] = s + b2[0];
  }
}

// out[g, n] = mean_f x[g, f, n]   (seg_feats.mean(dim=-2) of the sep verb head, code/mdl_conc_sep.py:64-129)
__global__ void row_mean_kernel(const float* x, float* out, int G, int F, int N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)G * N) return;
  const int n = (int)(i % N);
  const int64_t g = i / N;
  float s = 0.f;
  for (int f = 0; f < F; ++f) s += x[(g * F + f) * N + n];
  out[i] = s / (float)F;
}

// torch.optim.Adam (no weight decay, no amsgrad): the reference's optimizer, betas (0.9, 0.99) (code/main_dist.py:55)
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                            float bc1, float bc2_sqrt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

}  // namespace vog

extern "C" int vog_concat_rows_f32(const float* a, int Na, int rep_a, const float* b, int Nb, int rep_b, float* out, int M, void* stream) {
  VOG_CHECK_ARG(a && b && out && Na > 0 && Nb > 0 && rep_a >= 1 && rep_b >= 1 && M > 0);
  ::vog::launch(concat_rows_kernel, dim3((unsigned)(((int64_t)M * (Na + Nb) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, Na,
                rep_a, b, Nb, rep_b, out, M);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_conc_f32_fwd(const float* ps, const float* lang, const int64_t* inds_msk, float* out, int n_q, int nc_v, int nfrm,
                                int nppf, int nsrl, int dobj, int dlang, int lang_per_vid, void* stream) {
  VOG_CHECK_ARG(ps && lang && out && n_q > 0 && nc_v > 0 && nfrm > 0 && nppf > 0 && nsrl > 0 && dobj > 0 && dlang > 0);
  const int64_t total = (int64_t)n_q * nc_v * nfrm * nsrl * nppf * (dobj + dlang);
  ::vog::launch(conc_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ps, lang, inds_msk, out, n_q,
                nc_v, nfrm, nppf, nsrl, dobj, dlang, lang_per_vid);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_score_head_f32(const float* y, const float* wl, const float* bl, const float* wl2, const float* bl2, float* mdl_outs,
                                  void* scratch, size_t scratch_bytes, int M, int d, int dhead, int n_vid, int nfrm, int nppf, int nsrl,
                                  void* stream) {
  VOG_CHECK_ARG(y && wl && bl && wl2 && bl2 && mdl_outs && scratch && M > 0 && d > 0 && dhead > 0 && M == n_vid * nfrm * nppf * nsrl);
  if (scratch_bytes < (size_t)M * dhead * 4) VOG_FAIL(-2, "vog_score_head_f32: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  float* h = (float*)scratch;
  VOG_TRY(gemm_f32(y, d, 1, wl, 1, d, h, dhead, bl, 1, M, dhead, d, st));
  ::vog::launch(score_fwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, st, (const float*)h, wl2, bl2, mdl_outs, M, dhead, nfrm, nppf, nsrl);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            int step, void* stream) {
  VOG_CHECK_ARG(p && g && m && v && n > 0 && step >= 1 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f);
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  ::vog::launch(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                bc1, sqrtf(bc2));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_row_mean_f32(const float* x, float* out, int G, int F, int N, void* stream) {
  VOG_CHECK_ARG(x && out && G > 0 && F > 0 && N > 0);
  ::vog::launch(row_mean_kernel, dim3((unsigned)(((int64_t)G * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, G, F, N);
  VOG_LAUNCH_CHECK();
  return 0;
}

// lin2 alone (ImgGrnd / VidGrnd: the score head reads the [vis | lang] token matrix directly, code/mdl_vog.py:224-230,
// 286-344): gradients of lin2.{0,2} and of its input. x [M, d], rows ((video, frame), (arg, p)) as vog_score_head_f32
// (frame-free models: nfrm = 1, nppf = NP). scratch >= 4 * M * dhead * 4 + 64 * max(d, dhead) * 4 + 4 * M bytes.
extern "C" int64_t vog_score_head_f32_bwd_scratch_bytes(int M, int d, int dhead) {
  if (M <= 0 || d <= 0 || dhead <= 0) return -1;
  return ((int64_t)3 * M * dhead + (int64_t)CS_CHUNKS * (d > dhead ? d : dhead) + M + 1024) * 4;
}
extern "C" int vog_score_head_f32_bwd(const float* x, const float* d_mdl_outs, const float* wl, const float* bl, const float* wl2,
                                      float* g_wl, float* g_bl, float* g_wl2, float* g_bl2, float* d_x, void* scratch,
                                      size_t scratch_bytes, int M, int d, int dhead, int n_vid, int nfrm, int nppf, int nsrl, void* stream) {
  VOG_CHECK_ARG(x && d_mdl_outs && wl && bl && wl2 && g_wl && g_bl && g_wl2 && g_bl2 && scratch && M == n_vid * nfrm * nppf * nsrl);
  if ((int64_t)scratch_bytes < vog_score_head_f32_bwd_scratch_bytes(M, d, dhead)) VOG_FAIL(-2, "vog_score_head_f32_bwd: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  float* s = (float*)scratch;
  float *h = s, *dh = h + (int64_t)M * dhead, *hw = dh + (int64_t)M * dhead, *dlog = hw + (int64_t)M * dhead;
  float* part = dlog + (M + 3) / 4 * 4;
  VOG_TRY(gemm_f32(x, d, 1, wl, 1, d, h, dhead, bl, 1, M, dhead, d, st));
  ScoreBwd sb{d_mdl_outs, h, wl2, dh, hw, dlog, M, nfrm, nppf, nsrl, dhead};
  ::vog::launch(score_bwd_kernel, dim3(M), dim3(256), 0, st, sb);
  VOG_TRY(colsum(hw, g_wl2, part, M, dhead, st));
  VOG_TRY(colsum(dlog, g_bl2, part, M, 1, st));
  VOG_TRY(gemm_f32(dh, 1, dhead, x, d, 1, g_wl, d, nullptr, 0, dhead, d, M, st));
  VOG_TRY(colsum(dh, g_bl, part, M, dhead, st));
  if (d_x) VOG_TRY(gemm_f32(dh, dhead, 1, wl, d, 1, d_x, d, nullptr, 0, M, d, dhead, st));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_train_set_int(const char* name, int value) {
  VOG_CHECK_ARG(name);
  if (strcmp(name, "bf16_gemm") == 0) { g_train_bf16 = value ? 1 : 0; return 0; }
  VOG_FAIL(-1, "vog_train_set_int: unknown option %s", name);
}

extern "C" int vog_train_get_int(const char* name, int* value) {
  VOG_CHECK_ARG(name && value);
  if (strcmp(name, "bf16_gemm") == 0) { *value = g_train_bf16; return 0; }
  VOG_FAIL(-1, "vog_train_get_int: unknown option %s", name);
}
