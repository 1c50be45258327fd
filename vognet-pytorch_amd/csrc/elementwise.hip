// HBM/latency-bound kernels of the forward path: layernorm, box-bias precursor,
// token re-index, argument vectors, vis||lang token layout, score head,
// pred_cmp head, prediction head. fp32 arithmetic throughout (these are exact
// restatements; only the MFMA contractions run in 16 bit).
#include "common.h"
#include "pred_dev.h"

namespace vog {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------
// K3 layernorm: one wave per row, row held in registers (d <= 1024)
// (ResidualBlock.forward transformer_code.py:30-31, nn.LayerNorm eps 1e-5)
// ---------------------------------------------------------------------------
template <typename T16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        float* __restrict__ y32,
                                                        unsigned short* __restrict__ y16,
                                                        int rows, int d) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  // d % 4 == 0 (checked by the launcher): 16 bytes per lane per access, a wave covers 1 KiB of the row
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * d);
  const int nq = d >> 2;
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    v[i] = c < nq ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    if (c < nq) {
      const float t0 = v[i].x - mean, t1 = v[i].y - mean, t2 = v[i].z - mean, t3 = v[i].w - mean;
      q += (t0 * t0 + t1 * t1) + (t2 * t2 + t3 * t3);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + i * 64;
    if (c < nq) {
      const float4 g = g4[c], bb = b4[c];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + bb.x; o.y = (v[i].y - mean) * rstd * g.y + bb.y;
      o.z = (v[i].z - mean) * rstd * g.z + bb.z; o.w = (v[i].w - mean) * rstd * g.w + bb.w;
      if (y32) reinterpret_cast<float4*>(y32 + (int64_t)row * d)[c] = o;
      if (y16) {
        const u16x4 h = {to16<T16>(o.x), to16<T16>(o.y), to16<T16>(o.z), to16<T16>(o.w)};
        reinterpret_cast<u16x4*>(y16 + (int64_t)row * d)[c] = h;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// fp32 -> 16-bit cast of the raw feature blocks (HBM-bound: 16 B in, 8 B out per lane)
// ---------------------------------------------------------------------------
template <typename T16>
__global__ __launch_bounds__(256) void cast2_kernel(const float4* __restrict__ s0, u16x4* __restrict__ d0,
                                                    int64_t n0, const float4* __restrict__ s1,
                                                    u16x4* __restrict__ d1, int64_t n1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += stride) {
    const bool first = i < n0;
    const float4 v = first ? s0[i] : s1[i - n0];
    u16x4 o = {to16<T16>(v.x), to16<T16>(v.y), to16<T16>(v.z), to16<T16>(v.w)};
    if (first) d0[i] = o; else d1[i - n0] = o;
  }
}

// ---------------------------------------------------------------------------
// split-K finish: sum the slabs, bias, ReLU, row replication, fp32 + 16-bit copies
// ---------------------------------------------------------------------------
struct SplitkProbs { vog_splitk_prob p[2]; int blocks0; };

__global__ __launch_bounds__(256) void splitk_finish_kernel(SplitkProbs a) {
  const bool second = (int)blockIdx.x >= a.blocks0;
  const vog_splitk_prob& q = a.p[second ? 1 : 0];
  const int64_t i4 = (int64_t)(blockIdx.x - (second ? a.blocks0 : 0)) * 256 + threadIdx.x;   // float4 index
  const int n4 = q.N / 4;
  if (i4 >= (int64_t)q.M * n4) return;
  const int m = (int)(i4 / n4), n = (int)(i4 - (int64_t)m * n4) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < q.splits; ++s) {
    const float4 x = *reinterpret_cast<const float4*>(q.slabs + ((int64_t)s * q.M + m) * q.N + n);
    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
  }
  if (q.bias) {
    const float4 b = *reinterpret_cast<const float4*>(q.bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (q.relu) { v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w); }
  u16x4 h;
  if (q.c16) {
    if (q.c16_dtype == VOG_BF16) h = u16x4{to16<BF16>(v.x), to16<BF16>(v.y), to16<BF16>(v.z), to16<BF16>(v.w)};
    else h = u16x4{to16<F16>(v.x), to16<F16>(v.y), to16<F16>(v.z), to16<F16>(v.w)};
  }
  for (int j = 0; j < q.rep; ++j) {
    const int64_t orow = (int64_t)m * q.rep + j;
    if (q.c32) *reinterpret_cast<float4*>(q.c32 + orow * q.ldc + n) = v;
    if (q.c16) *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(q.c16) + orow * q.ldc16 + n) = h;
  }
}

// ---------------------------------------------------------------------------
// u[row,h] = W_pe[h,:] . norm(box[row,:5])   (compute_pe mdl_vog.py:456-463)
// ---------------------------------------------------------------------------
__global__ void box_u_kernel(const float* __restrict__ props, const float* __restrict__ w,
                             float* __restrict__ u, int n_rows, int H, float vid_w, float vid_h,
                             float nfrm_div) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * H) return;
  const int r = i / H, h = i % H;
  const float* b = props + (int64_t)r * 7;
  // true divisions, as the reference normalises (not multiplies by a reciprocal)
  const float b0 = b[0] / vid_w, b1 = b[1] / vid_h, b2 = b[2] / vid_w, b3 = b[3] / vid_h,
              b4 = b[4] / nfrm_div;
  const float* wh = w + h * 5;
  u[i] = wh[0] * b0 + wh[1] * b1 + wh[2] * b2 + wh[3] * b3 + wh[4] * b4;
}

// ---------------------------------------------------------------------------
// K6 token re-index (get_srl_arg_seq_to_sent_seq mdl_vog.py:67-95)
// ---------------------------------------------------------------------------
__global__ void srl_gather_kernel(const int64_t* __restrict__ words, const int64_t* __restrict__ mask,
                                  int32_t* __restrict__ tok, int Bn, int T, int nsrl, int seq_len,
                                  int vocab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bn * T) return;
  const int b = i / T, t = i % T;
  const int64_t m = mask[(int64_t)b * seq_len + t];
  int64_t v = vocab;
  if (m >= 0 && m < (int64_t)nsrl * seq_len) v = words[(int64_t)b * nsrl * seq_len + m];
  tok[i] = (int32_t)v;
}

// ---------------------------------------------------------------------------
// argument vectors (retrieve_srl_arg_from_lang_encode mdl_vog.py:97-140)
// one workgroup per (sentence, arg); wave-per-output dot products, fp32 exact
// ---------------------------------------------------------------------------
constexpr int AV_ROWS = 20;      // (sentence, argument) rows per workgroup
constexpr int AV_MAXSL = 64;     // 2L / 16 <= 64 elements of a dot product per thread (L <= 512)
// CSL > 0: the slice length 2L / 16 is this compile-time constant (L = 256: 32) - no guards, no index arithmetic in the loops
// (the guarded form is ~200 instructions per row at one wave per SIMD: 18 us for 5 MFLOP); CSL = 0: any L % 8 == 0.
template <int CSL>
__global__ __launch_bounds__(256) void argvec_kernel(const float* __restrict__ full,
                                                     const int64_t* __restrict__ capture,
                                                     const int64_t* __restrict__ msk,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ lang, int T, int nsrl, int L, int nrows) {
  // grid (ceil(rows / 20), L/16): a workgroup owns 16 output columns for up to 20 (sentence, argument) rows. Round 6: 16
  // workgroups at cfg 2 instead of 320 (one per row and column block: 1137 CU-us for 5 MFLOP, each re-fetching its 32 KB
  // weight slice; profiles/round5_busy_cu_cfg2.md). Thread (o = tid & 15, s = tid >> 4) holds slice s (1/16 of the 2L-long
  // dot product) of output column o's weight row in registers; the rows' [full[cap0] || full[cap1]] vectors are staged in
  // LDS with ONE round of loads (capture positions and masks first, so nothing inside a loop waits for memory); partial
  // sums meet in LDS. No wave reductions: the first 16-workgroup form kept the old kernel's wave-per-4-outputs mapping and
  // spent 28 us in 80 dependent 64-lane shuffle reductions per wave (6 LDS-pipe round trips each).
  extern __shared__ __attribute__((aligned(16))) float av_x[];          // [AV_ROWS][2L]
  __shared__ int av_src[AV_ROWS * 2];
  __shared__ float av_msk[AV_ROWS];
  __shared__ float av_part[AV_ROWS][16][17];
  const int tid = threadIdx.x;
  const int r0 = (int)blockIdx.x * AV_ROWS;
  const int nr = min(AV_ROWS, nrows - r0);
  const int K = CSL > 0 ? CSL * 16 : 2 * L, nq = K >> 2, lq = nq >> 1;      // (CSL > 0: compile-time - the index divisions fold)
  if (tid < nr * 2) {
    const int row = r0 + (tid >> 1), b = row / nsrl;
    int64_t c = capture[(int64_t)row * 2 + (tid & 1)];
    c = c < 0 ? 0 : (c >= T ? T - 1 : c);
    av_src[tid] = b * T + (int)c;
  } else if (tid >= 64 && tid < 64 + nr) {
    av_msk[tid - 64] = (float)msk[r0 + tid - 64];
  }
  // this thread's slice of its weight row (requested before the barrier: in flight while the rows are staged)
  const int o = tid & 15, sl = tid >> 4;
  const int col = (int)blockIdx.y * 16 + o;
  const int SL = CSL > 0 ? CSL : ((K + 15) >> 4), j0 = sl * SL;
  const bool vec = CSL > 0 || (SL & 3) == 0;                          // (L = 256: 32 elements per slice)
  constexpr int NW = CSL > 0 ? CSL : AV_MAXSL;
  float wr[NW];
  if (vec) {
#pragma unroll
    for (int q = 0; q < NW / 4; ++q) {
      const bool ok = col < L && (CSL > 0 || (q * 4 < SL && j0 + q * 4 < K));
      const float4 v = ok ? *reinterpret_cast<const float4*>(w + (int64_t)col * K + j0 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      wr[q * 4] = v.x; wr[q * 4 + 1] = v.y; wr[q * 4 + 2] = v.z; wr[q * 4 + 3] = v.w;
    }
  } else if constexpr (CSL == 0) {
#pragma unroll
    for (int q = 0; q < NW; ++q) wr[q] = (col < L && q < SL && j0 + q < K) ? w[(int64_t)col * K + j0 + q] : 0.f;
  }
  __syncthreads();
  constexpr int AV_IT = AV_ROWS;                     // nq <= 256 float4 per row: <= AV_ROWS chunks per thread
  float4 stg[AV_IT];
#pragma unroll
  for (int it = 0; it < AV_IT; ++it) {
    int idx = tid + it * 256;
    idx = idx < nr * nq ? idx : nr * nq - 1;         // (clamped, not skipped: every element of stg is assigned - registers, no scratch)
    const int r = idx / nq, i = idx - r * nq;
    stg[it] = reinterpret_cast<const float4*>(full + (int64_t)av_src[r * 2 + (i < lq ? 0 : 1)] * L)[i < lq ? i : i - lq];
  }
  // CSL > 0: a thread's slice of a row starts SLP = CSL + 4 floats after the previous one's (not CSL): the 16 slices of a row are
  // read at the same moment by the 16 threads of an output column, and at a stride of 32 floats all of them sit on the same
  // four LDS banks (16-way conflict on every one of the 160 float4 reads of the product loop)
  constexpr int SLP = CSL > 0 ? CSL + 4 : 0;
  const int KP = CSL > 0 ? 16 * SLP : K;               // row pitch in LDS
#pragma unroll
  for (int it = 0; it < AV_IT; ++it) {
    const int idx = tid + it * 256;
    if (idx < nr * nq) {
      if constexpr (CSL > 0) {
        const int r = idx / nq, i = idx - r * nq;      // float4 i of row r: slice i / (CSL / 4), piece i % (CSL / 4)
        reinterpret_cast<float4*>(av_x)[r * (KP / 4) + (i / (CSL / 4)) * (SLP / 4) + (i % (CSL / 4))] = stg[it];
      } else {
        reinterpret_cast<float4*>(av_x)[idx] = stg[it];
      }
    }
  }
  __syncthreads();
  for (int r = 0; r < nr; ++r) {
    const float* xr = av_x + r * KP + (CSL > 0 ? sl * SLP : j0);
    float acc = 0.f;
    if (vec) {
#pragma unroll
      for (int q = 0; q < NW / 4; ++q)
        if (CSL > 0 || (q * 4 < SL && j0 + q * 4 < K)) {
          const float4 x = *reinterpret_cast<const float4*>(xr + q * 4);
          acc += (wr[q * 4] * x.x + wr[q * 4 + 1] * x.y) + (wr[q * 4 + 2] * x.z + wr[q * 4 + 3] * x.w);
        }
    } else if constexpr (CSL == 0) {
#pragma unroll
      for (int q = 0; q < NW; ++q)
        if (q < SL && j0 + q < K) acc += wr[q] * xr[q];
    }
    av_part[r][o][sl] = acc;
  }
  __syncthreads();
  for (int idx = tid; idx < nr * 16; idx += 256) {
    const int r = idx >> 4, oo = idx & 15, c2 = (int)blockIdx.y * 16 + oo;
    if (c2 >= L) continue;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += av_part[r][oo][q];
    lang[(int64_t)(r0 + r) * L + c2] = relu_nan(v + bias[c2]) * av_msk[r];
  }
}

// ---------------------------------------------------------------------------
// K4 vis||lang token layout (concate_vis_lang_feats mdl_vog.py:316-344 + the
// regroup of conc_encode2 :693-699). Row (s=(v,f), j=a*nppf+p).
// ---------------------------------------------------------------------------
template <typename T16>
__global__ __launch_bounds__(256) void vislang_kernel(vog_vislang_args a) {
  const int64_t row = blockIdx.x;
  const int N = a.nsrl * a.nppf;
  const int s = (int)(row / N), j = (int)(row % N);
  const int v = s / a.nfrm, f = s % a.nfrm;
  const int arg = j / a.nppf, pp = j % a.nppf;
  const float* vis = a.vis + ((int64_t)v * a.nfrm * a.nppf + (int64_t)f * a.nppf + pp) * a.dv;
  const int lv = a.lang_per_vid ? v : v / a.nc_v;
  const float* lang = a.lang + ((int64_t)lv * a.nsrl + arg) * a.dl;
  const int d = a.dv + a.dl;
  float* x32 = a.x32 ? a.x32 + row * d : nullptr;
  unsigned short* x16 = a.x16 ? reinterpret_cast<unsigned short*>(a.x16) + row * d : nullptr;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    const float val = c < a.dv ? vis[c] : lang[c - a.dv];
    if (x32) x32[c] = val;
    if (x16) x16[c] = to16<T16>(val);
  }
}

// ---------------------------------------------------------------------------
// structured layer-0 QKV of mul_tx: q/k/v[token(a,p)] = PV[vis row p] + PL[lang row a]
// (see vog_qkvcomb_args). grid (sequence, head, {q,k,v}); each PV element is read
// once and fanned out to the nsrl tokens that share it.
// ---------------------------------------------------------------------------
template <typename T16>
__global__ __launch_bounds__(256) void qkv_combine_kernel(vog_qkvcomb_args a, int chunks) {
  // grid.x = sequence * chunks: one work item per thread so that every load of the
  // launch is in flight at once (this is a pure L2 -> HBM streaming pass)
  const int s = blockIdx.x / chunks, chunk = blockIdx.x - s * chunks;
  const int h = blockIdx.y, which = blockIdx.z;
  const int v = s / a.nfrm, f = s - v * a.nfrm;
  const int ldp = 3 * a.H * a.dp;
  const int col0 = (which * a.H + h) * a.dp;
  const int lv = a.lang_per_vid ? v : v / a.nc_v;
  const float* pv = a.pv + ((int64_t)v * a.nfrm * a.nppf + (int64_t)f * a.nppf) * ldp + col0;
  const float* pl = a.pl + (int64_t)lv * a.nsrl * ldp + col0;
  const int64_t sh = (int64_t)s * a.H + h;
  const int it = chunk * 256 + threadIdx.x;
  if (which < 2) {
    unsigned short* dst = reinterpret_cast<unsigned short*>(which == 0 ? a.q : a.k) + sh * a.npad * a.dp;
    const int cpr = a.dp / 8;                       // 8-column chunks per row
    if (it >= a.nppf * cpr) return;
    const int pp = it / cpr, c = it - pp * cpr;
    const float4 x0 = *reinterpret_cast<const float4*>(pv + (int64_t)pp * ldp + c * 8);
    const float4 x1 = *reinterpret_cast<const float4*>(pv + (int64_t)pp * ldp + c * 8 + 4);
#pragma unroll 5
    for (int ar = 0; ar < a.nsrl; ++ar) {
      const float4 l0 = *reinterpret_cast<const float4*>(pl + (int64_t)ar * ldp + c * 8);
      const float4 l1 = *reinterpret_cast<const float4*>(pl + (int64_t)ar * ldp + c * 8 + 4);
      u16x8 o = {to16<T16>(x0.x + l0.x), to16<T16>(x0.y + l0.y), to16<T16>(x0.z + l0.z), to16<T16>(x0.w + l0.w),
                 to16<T16>(x1.x + l1.x), to16<T16>(x1.y + l1.y), to16<T16>(x1.z + l1.z), to16<T16>(x1.w + l1.w)};
      *reinterpret_cast<u16x8*>(dst + frag_qk(ar * a.nppf + pp, c * 8, a.dp)) = o;
    }
  } else {
    // V fragments: thread = (dd, group of 4 proposals); lanes run along dd so the PV
    // reads are coalesced; each thread emits nsrl 8-byte stores
    unsigned short* dst = reinterpret_cast<unsigned short*>(a.vt) + sh * a.npad * a.dp;
    const int ng = (a.nppf + 3) / 4;
    const bool vec = (a.nppf & 3) == 0;
    if (it >= a.dp * ng) return;
    const int g = it / a.dp, dd = it - g * a.dp;
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pp = g * 4 + e;
      x[e] = pp < a.nppf ? pv[(int64_t)pp * ldp + dd] : 0.f;
    }
#pragma unroll 5
    for (int ar = 0; ar < a.nsrl; ++ar) {
      const float l = pl[(int64_t)ar * ldp + dd];
      const int tok = ar * a.nppf + g * 4;
      if (vec) {   // 4 consecutive tokens, tok % 4 == 0 -> 4 consecutive j of one fragment lane
        u16x4 o = {to16<T16>(x[0] + l), to16<T16>(x[1] + l), to16<T16>(x[2] + l), to16<T16>(x[3] + l)};
        *reinterpret_cast<u16x4*>(dst + frag_v(tok, dd, a.dp)) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (g * 4 + e < a.nppf) dst[frag_v(tok + e, dd, a.dp)] = to16<T16>(x[e] + l);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K7 score head tail: lin2.2 + inverse regroup + masks
// (mdl_vog.py:675-677,724-737; mdl_conc_single.py:39-49,118-122,144-154;
//  mdl_conc_sep.py:32-42,205-210). One wave per token row.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_kernel(vog_score_args a) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int N = a.nsrl * a.nppf;
  const int64_t nrows = (int64_t)a.n_vid * a.nfrm * N;
  if (row >= nrows) return;
  const float* h = a.h1 + row * a.dh;
  float acc = 0.f;
  for (int i = lane; i < a.dh; i += 64) acc += h[i] * a.w2[i];
  acc = wave_sum(acc);
  if (lane != 0) return;
  const float logit = acc + a.b2[0];
  const int s = (int)(row / N), j = (int)(row % N);
  const int v = s / a.nfrm, f = s % a.nfrm;
  const int arg = j / a.nppf, pp = j % a.nppf;
  const int NP = a.nfrm * a.nppf;
  const int r = f * a.nppf + pp;                     // proposal row inside the model video
  const int64_t o = ((int64_t)v * a.nsrl + arg) * NP + r;
  const int b = v / a.nc_v, c = v % a.nc_v;
  int cmp;
  if (a.conc_type == VOG_CONC_TEMP) cmp = r / (a.nfrm0 * a.nppf0);
  else if (a.conc_type == VOG_CONC_SPAT) cmp = (r / a.nppf0) % a.ncmp;
  else cmp = c;
  const int lrow = a.nvl > 1 ? (b * a.nvl + c) : b;  // language copy of this video
  const float am = (float)a.arg_msk[(int64_t)lrow * a.nsrl + arg];
  const float cm = (float)a.cmp_msk[(int64_t)b * a.ncmp + cmp];
  a.outs[o] = logit;
  a.outs_eval[o] = sigmoidf_(logit) * am * cm;
}

// ---------------------------------------------------------------------------
// K8 pred_cmp head (mdl_vog.py:365-397; mdl_conc_sep.py:64-129). One workgroup
// per (query, video).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void predcmp_kernel(vog_predcmp_args a) {
  extern __shared__ float sm[];                 // [L + dseg] input, [256] hidden, [nsrl] s
  const int bc = blockIdx.x, b = bc / a.ncmp, c = bc % a.ncmp;
  const int dseg = a.dps - a.dp0;
  const int din = a.L + dseg;
  float* xin = sm;
  float* hid = sm + din;
  float* sarg = hid + 256;
  const int lrow = a.nvl > 1 ? (b * a.nvl + c) : b;
  const int Fv = a.NP / a.nppf0;
  for (int i = threadIdx.x; i < din; i += blockDim.x) {
    if (i < a.L) xin[i] = a.final_hidden[(int64_t)lrow * a.L + i];
    else {
      float s = 0.f;                            // mean over frames of the segment encoding
      for (int f = 0; f < Fv; ++f)
        s += a.prop_seg[((int64_t)bc * a.NP + (int64_t)f * a.nppf0) * a.dps + a.dp0 + (i - a.L)];
      xin[i] = s / (float)Fv;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int o = wid; o < 256; o += nw) {
    const float* wr = a.w0 + (int64_t)o * din;
    float acc = 0.f;
    for (int i = lane; i < din; i += 64) acc += wr[i] * xin[i];
    acc = wave_sum(acc);
    if (lane == 0) hid[o] = relu_nan(acc + a.b0[o]);
  }
  __syncthreads();
  float vid = 0.f;
  if (wid == 0) {
    float acc = 0.f;
    for (int i = lane; i < 256; i += 64) acc += a.w2[i] * hid[i];
    vid = wave_sum(acc) + a.b2[0];
    if (lane == 0) a.vidf_outs[bc] = vid;
  }
  // per-arg max over proposals of sigmoid(logit) = sigmoid(max logit)
  for (int arg = wid; arg < a.nsrl; arg += nw) {
    const float* o = a.outs + ((int64_t)bc * a.nsrl + arg) * a.NP;
    float m = -3.0e38f;
    for (int i = lane; i < a.NP; i += 64) m = fmaxf(m, sigmoidf_(o[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) sarg[arg] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int vslot = (int)a.verb_ind[bc];
    const float cm = (float)a.cmp_msk[bc];
    float num = 0.f, den = 0.f;
    for (int arg = 0; arg < a.nsrl; ++arg) {
      float s = (arg == vslot) ? sigmoidf_(vid) : sarg[arg];
      const float am = (float)a.arg_msk[(int64_t)lrow * a.nsrl + arg];
      s *= am;
      num += s; den += am;
      a.fin_scores_loss[(int64_t)bc * a.nsrl + arg] = s * cm;
    }
    a.fin_scores[bc] = num / den * cm;
  }
}

// ---------------------------------------------------------------------------
// prediction head (eval_vsrl_corr.py:162-220, 289-345, 357-424): one thread per
// (query, arg, video, frame); packed record per query.
// ---------------------------------------------------------------------------
// Many proposals per frame (p100: 100): one WAVE per (query, arg, frame, video): the lanes split the
// proposals, the arg-max is a wave reduction that keeps torch.max's rule (first maximum wins: larger value,
// then smaller index). The thread-per-item form below walks the proposals in a dependent loop: 33.7 us at
// nppf0 = 100 for 832 threads.
__device__ __forceinline__ void wave_argmax_first(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(i, o);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

// the forward's largest |attention logit| per stack -> the host's sticky maximum (vog_batch.stats): threads 0 / 1 of block 0
// The head folds the two stacks' logit reports (vog_attn_args.logit_max: [2][4 layers][VOG_LOGIT_WORDS words, 128 B apart]) into the
// pinned host words. Wave 0 of workgroup 0: lane (stack, word) loads its 4 layers' words when the kernel STARTS (stats_begin: in
// flight behind the head's own loads) and folds them when it ends (stats_end). The host word is touched only when this workspace's
// forwards have seen a larger value than they last published - a.published, zeroed with the workspace only: a system-scope atomic
// on pinned host memory is a PCIe round trip the kernel would otherwise wait out in every forward.
struct StatsProbe { unsigned int m, last; };
__device__ __forceinline__ StatsProbe stats_begin(const vog_pred_args& a) {
  StatsProbe s{0u, 0xffffffffu};
  if (!a.stats || !a.logit_max || !a.published || blockIdx.x != 0 || threadIdx.x >= 64) return s;
  static_assert(VOG_LOGIT_WORDS == 32, "one lane per (stack, word)");
  const int stack = threadIdx.x >> 5, word = threadIdx.x & 31;
  const unsigned int* w = a.logit_max + (stack * 4) * (VOG_LOGIT_WORDS * VOG_LOGIT_STRIDE) + word * VOG_LOGIT_STRIDE;
  unsigned int v[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) v[l] = __hip_atomic_load(w + l * (VOG_LOGIT_WORDS * VOG_LOGIT_STRIDE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s.last = a.published[stack];
#pragma unroll
  for (int l = 0; l < 4; ++l) s.m = v[l] > s.m ? v[l] : s.m;
  return s;
}
__device__ __forceinline__ void stats_end(const vog_pred_args& a, StatsProbe s) {
  if (!a.stats || !a.logit_max || !a.published || blockIdx.x != 0 || threadIdx.x >= 64) return;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { const unsigned int t = __shfl_xor(s.m, o); s.m = t > s.m ? t : s.m; }
  const int stack = threadIdx.x >> 5;
  if ((threadIdx.x & 31) == 0 && s.m > s.last) {
    a.published[stack] = s.m;
    __hip_atomic_fetch_max(a.stats + stack, s.m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__device__ __forceinline__ void pred_wave_item(const vog_pred_args& a, int64_t rec_bytes) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int per_q = a.nsrl * a.nfrm0 * a.ncmp;
  if (i >= a.B * per_q) return;
  const int b = i / per_q, r = i % per_q;
  const int arg = r / (a.nfrm0 * a.ncmp), f = (r / a.ncmp) % a.nfrm0, c = r % a.ncmp;
  const int npv = a.nfrm0 * a.nppf0;
  unsigned char* rec = reinterpret_cast<unsigned char*>(a.rec) + (int64_t)b * rec_bytes;
  float* boxes = reinterpret_cast<float*>(rec);
  float* scores = boxes + (int64_t)a.nsrl * a.ncmp * a.nfrm0 * 7;
  int64_t* idx = reinterpret_cast<int64_t*>(rec + (int64_t)a.nsrl * a.ncmp * a.nfrm0 * 8 * 4);
  auto first_prop = [&](int cc, int64_t* e0, int64_t* p0) {   // in outs_eval / in props
    if (a.conc_type == VOG_CONC_SPAT) {
      const int r0 = (f * a.ncmp + cc) * a.nppf0;
      *e0 = ((int64_t)b * a.nsrl + arg) * ((int64_t)a.ncmp * npv) + r0;
      *p0 = (int64_t)b * a.ncmp * npv + r0;
    } else if (a.conc_type == VOG_CONC_TEMP) {
      const int r0 = (cc * a.nfrm0 + f) * a.nppf0;
      *e0 = ((int64_t)b * a.nsrl + arg) * ((int64_t)a.ncmp * npv) + r0;
      *p0 = (int64_t)b * a.ncmp * npv + r0;
    } else {
      *e0 = (((int64_t)b * a.ncmp + cc) * a.nsrl + arg) * npv + (int64_t)f * a.nppf0;
      *p0 = ((int64_t)b * a.ncmp + cc) * npv + (int64_t)f * a.nppf0;
    }
  };
  auto frame_max = [&](int cc, float* best, int* bi) {      // every lane returns the frame's (max, first index)
    int64_t e0, p0;
    first_prop(cc, &e0, &p0);
    float v = -INFINITY; int k0 = 0x7fffffff;
    for (int k = lane; k < a.nppf0; k += 64) {
      const float x = a.outs_eval[e0 + k];
      if (x > v) { v = x; k0 = k; }                          // ascending k per lane: the first maximum stays
    }
    wave_argmax_first(v, k0);
    *best = v; *bi = k0;
  };
  float best; int bi;
  frame_max(c, &best, &bi);
  int64_t e0, p0;
  first_prop(c, &e0, &p0);
  const int64_t o = ((int64_t)arg * a.ncmp + c) * a.nfrm0 + f;
  if (lane < 7) boxes[o * 7 + lane] = a.props[(p0 + bi) * 7 + lane];
  if (lane == 0) scores[o] = best;
  if (c != 0) return;
  int64_t out = 0;
  if (a.conc_type == VOG_CONC_SPAT) {
    float best_c = best;                           // video 0; first maximum over the videos
    for (int cc = 1; cc < a.ncmp; ++cc) {
      float bc; int dummy;
      frame_max(cc, &bc, &dummy);
      if (bc > best_c) { best_c = bc; out = cc; }
    }
  } else if (a.conc_type == VOG_CONC_SEP) {
    float bf = a.fin_scores[(int64_t)b * a.ncmp];
    for (int cc = 1; cc < a.ncmp; ++cc) {
      const float v = a.fin_scores[(int64_t)b * a.ncmp + cc];
      if (v > bf) { bf = v; out = cc; }
    }
  }
  if (lane == 0) idx[(int64_t)arg * a.nfrm0 + f] = out;
}
__global__ __launch_bounds__(256) void pred_wave_kernel(vog_pred_args a, int64_t rec_bytes) {
  const StatsProbe sp = stats_begin(a);
  pred_wave_item(a, rec_bytes);
  stats_end(a, sp);
}

__global__ void pred_kernel(vog_pred_args a, int64_t rec_bytes) {
  const StatsProbe sp = stats_begin(a);
  pred_item<false>(a, rec_bytes, blockIdx.x * blockDim.x + threadIdx.x);
  stats_end(a, sp);
}

// ---------------------------------------------------------------------------
// fused prologues (one graph node each instead of 3)
// ---------------------------------------------------------------------------
struct LangPrepArgs {
  uint4* zero; int64_t zero16; int64_t ones16; const int64_t* words; const int64_t* mask; const int64_t* lens;
  int32_t* tok; int32_t* rows; int Bn, T, nsrl, seq_len, vocab;
  const unsigned short* emb16; unsigned short* a0; int E;
};

__device__ __forceinline__ void lang_prep_body(const LangPrepArgs& a, int bid, int nblocks) {
  uint4* __restrict__ zero = a.zero; const int64_t zero16 = a.zero16;
  const int64_t* __restrict__ words = a.words; const int64_t* __restrict__ mask = a.mask;
  const int64_t* __restrict__ lens = a.lens; int32_t* __restrict__ tok = a.tok; int32_t* __restrict__ rows = a.rows;
  const int Bn = a.Bn, T = a.T, nsrl = a.nsrl, seq_len = a.seq_len, vocab = a.vocab;
  const unsigned short* __restrict__ emb16 = a.emb16; unsigned short* __restrict__ a0 = a.a0; const int E = a.E;
  const int64_t gid = (int64_t)bid * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)nblocks * blockDim.x;
  // [zero16 x 16 B of zeros][ones16 x 16 B of 0xff: the hand-off slots of the persistent BiLSTM, lstm_dev.h]
  const int64_t fill16 = zero16 + a.ones16;
  for (int64_t i = gid; i < fill16; i += stride) {
    const unsigned v = i < zero16 ? 0u : 0xffffffffu;
    zero[i] = make_uint4(v, v, v, v);
  }
  if (a0) {
    // embedding rows of the Bn*T tokens, 16 bit, in the A-fragment order of the M <= 64 GEMM:
    // one 16-byte chunk (8 consecutive k of one token) per thread
    const int cpr = E / 8;
    for (int64_t j = gid; j < (int64_t)Bn * T * cpr; j += stride) {
      const int i = (int)(j / cpr), ch = (int)(j % cpr), b = i / T, t = i % T;
      const int64_t m = mask[(int64_t)b * seq_len + t];
      int64_t v = vocab;
      if (m >= 0 && m < (int64_t)nsrl * seq_len) v = words[(int64_t)b * nsrl * seq_len + m];
      *reinterpret_cast<uint4*>(a0 + frag_a(i, ch * 8, E)) =
          *reinterpret_cast<const uint4*>(emb16 + v * E + ch * 8);
    }
  }
  if (gid < (int64_t)Bn * T) {
    const int i = (int)gid, b = i / T, t = i % T;
    const int64_t m = mask[(int64_t)b * seq_len + t];
    int64_t v = vocab;
    if (m >= 0 && m < (int64_t)nsrl * seq_len) v = words[(int64_t)b * nsrl * seq_len + m];
    tok[i] = (int32_t)v;
    const int len = (int)lens[b];
    rows[i] = t < len ? t * Bn + b : -1;
    rows[Bn * T + i] = t < len ? T * Bn - 1 + (len - 1 - t) * Bn + b : -1;
  }
}

__global__ __launch_bounds__(256) void lang_prep_kernel(LangPrepArgs a) { lang_prep_body(a, blockIdx.x, gridDim.x); }

template <typename T16>
__device__ __forceinline__ void vis_prep_body(const vog_visprep_args& a, int cast_blocks, int bid) {
  if (bid < cast_blocks) {
    const int64_t q0 = a.n0 / 4, q1 = a.n1 / 4;
    const int64_t stride = (int64_t)cast_blocks * blockDim.x;
    const float4* s0 = reinterpret_cast<const float4*>(a.src0);
    const float4* s1 = reinterpret_cast<const float4*>(a.src1);
    u16x4* d0 = reinterpret_cast<u16x4*>(a.dst0);
    u16x4* d1 = reinterpret_cast<u16x4*>(a.dst1);
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < q0 + q1; i += stride) {
      const bool first = i < q0;
      const float4 v = first ? s0[i] : s1[i - q0];
      u16x4 o = {to16<T16>(v.x), to16<T16>(v.y), to16<T16>(v.z), to16<T16>(v.w)};
      if (first) d0[i] = o; else d1[i - q0] = o;
    }
    return;
  }
  const int i = (bid - cast_blocks) * blockDim.x + threadIdx.x;
  const int n0 = a.w_pe0 ? a.n_rows * a.H0 : 0, n1 = a.w_pe1 ? a.n_rows * a.H1 : 0;
  if (i >= n0 + n1) return;
  const bool second = i >= n0;
  const int k = second ? i - n0 : i;
  const int H = second ? a.H1 : a.H0;
  const float fd = second ? a.nfrm_div1 : a.nfrm_div0;
  const float* w = (second ? a.w_pe1 : a.w_pe0) + (k % H) * 5;
  const float* b = a.props + (int64_t)(k / H) * 7;
  const float v = w[0] * (b[0] / a.vid_w) + w[1] * (b[1] / a.vid_h) + w[2] * (b[2] / a.vid_w) +
                  w[3] * (b[3] / a.vid_h) + w[4] * (b[4] / fd);
  (second ? a.u1 : a.u0)[k] = v;
}

template <typename T16>
__global__ __launch_bounds__(256) void vis_prep_kernel(vog_visprep_args a, int cast_blocks) {
  vis_prep_body<T16>(a, cast_blocks, blockIdx.x);
}

// both prologues of a forward in one launch (they are independent of each other; one launch less on
// the dependent chain): blocks [0, lang_blocks) run the language part, the rest the visual part
template <typename T16>
__global__ __launch_bounds__(256) void prep_fused_kernel(LangPrepArgs la, int lang_blocks, vog_visprep_args va,
                                                         int cast_blocks) {
  if ((int)blockIdx.x < lang_blocks) lang_prep_body(la, blockIdx.x, lang_blocks);
  else vis_prep_body<T16>(va, cast_blocks, (int)blockIdx.x - lang_blocks);
}

static int lang_prep_setup(void* zero, int64_t zero_bytes, int64_t ones_bytes, const int64_t* words_ind, const int64_t* word_mask,
                           const int64_t* lens, int32_t* tok, int32_t* rows, int Bn, int T, int nsrl,
                           int seq_len, int vocab_size, const void* emb16, void* a0_frag, int emb_dim,
                           LangPrepArgs* la, int* blocks_out) {
  VOG_CHECK_ARG(words_ind && word_mask && lens && tok && rows && Bn > 0 && T > 0 && T <= seq_len);
  VOG_CHECK_ARG(!a0_frag || (emb16 && emb_dim > 0 && (emb_dim % 32) == 0));
  VOG_CHECK_ARG(zero_bytes >= 0 && (zero_bytes % 16) == 0 && (zero_bytes == 0 || zero));
  VOG_CHECK_ARG(ones_bytes >= 0 && (ones_bytes % 16) == 0 && (ones_bytes == 0 || zero));
  const int64_t z16 = zero_bytes / 16, o16 = ones_bytes / 16;
  // 16 stores of 16 bytes per thread (round 6; one per thread before: 405 workgroups = 502 CU-us at cfg 2 for 1.6 MB of fill)
  static const int fill_per_thread = perf_env("VOG_PREP_FILL") ? atoi(perf_env("VOG_PREP_FILL")) : 16;
  int64_t blocks = (z16 + o16 + 256 * fill_per_thread - 1) / (256 * fill_per_thread);
  if (blocks > 1024) blocks = 1024;
  int64_t need = ((int64_t)Bn * T + 255) / 256;
  if (a0_frag) { const int64_t n2 = ((int64_t)Bn * T * (emb_dim / 8) + 255) / 256; need = n2 > need ? n2 : need; }
  if (need > 1024) need = 1024;
  if (blocks < need) blocks = need;
  *la = LangPrepArgs{(uint4*)zero, z16, o16, words_ind, word_mask, lens, tok, rows, Bn, T, nsrl, seq_len, vocab_size,
                     (const unsigned short*)emb16, (unsigned short*)a0_frag, emb_dim};
  *blocks_out = (int)blocks;
  return 0;
}

static int vis_prep_setup(const vog_visprep_args* a, int* cast_blocks, int* u_blocks) {
  VOG_CHECK_ARG(a && (a->n0 % 4) == 0 && (a->n1 % 4) == 0 && (a->n0 == 0 || (a->src0 && a->dst0)) &&
                (a->n1 == 0 || (a->src1 && a->dst1)));
  VOG_CHECK_ARG((!a->w_pe0 && !a->w_pe1) || (a->props && a->n_rows > 0));
  VOG_CHECK_ARG((!a->w_pe0 || (a->u0 && a->H0 > 0)) && (!a->w_pe1 || (a->u1 && a->H1 > 0)));
  const int64_t q = (a->n0 + a->n1) / 4;
  *cast_blocks = (int)((q + 255) / 256 < 2048 ? (q + 255) / 256 : 2048);
  const int nu = (a->w_pe0 ? a->n_rows * a->H0 : 0) + (a->w_pe1 ? a->n_rows * a->H1 : 0);
  *u_blocks = ceil_div(nu, 256);
  return 0;
}

}  // namespace vog

using namespace vog;

namespace vog {
// top BiLSTM layer's 16-bit output (A-fragment order or plain rows) -> plain fp32 rows: [rows_x, W] into x and the rows behind
// them (the final states) into fin (vog_bilstm_fwd: the LSTMEncoder's own return values, utils/mdl_srl_utils.py:152-169)
template <typename T16>
__global__ __launch_bounds__(256) void lstm_out_f32_kernel(const unsigned short* __restrict__ o16, int frag, int rows_x, int rows_f,
                                                           int W, float* __restrict__ x, float* __restrict__ fin) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)(rows_x + rows_f) * W) return;
  const int r = (int)(i / W), k = (int)(i - (int64_t)r * W);
  const float v = from16<T16>(o16[frag ? frag_a(r, k, W) : (int64_t)r * W + k]);
  if (r < rows_x) x[(int64_t)r * W + k] = v;
  else fin[(int64_t)(r - rows_x) * W + k] = v;
}
}  // namespace vog

extern "C" int vog_lstm_out_to_f32(const void* out16, int frag, int rows_x, int rows_f, int W, vog_dtype dtype, float* x,
                                   float* fin, void* stream) {
  VOG_CHECK_ARG(out16 && x && fin && rows_x > 0 && rows_f > 0 && W > 0 && (!frag || (W % 32) == 0));
  const int64_t n = (int64_t)(rows_x + rows_f) * W;
  VOG_DISPATCH_DTYPE(dtype, ::vog::launch(vog::lstm_out_f32_kernel<T16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                                          (hipStream_t)stream, (const unsigned short*)out16, frag, rows_x, rows_f, W, x, fin));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_residual_layernorm(const float* x, const float* gamma, const float* beta,
                                      float* y32, void* y16, int rows, int d, vog_dtype dtype,
                                      void* stream) {
  VOG_CHECK_ARG(x && gamma && beta && (y32 || y16) && rows > 0 && d > 0 && d <= 1024 && (d % 4) == 0);
  VOG_DISPATCH_DTYPE(dtype, ::vog::launch((layernorm_kernel<T16>), dim3(ceil_div(rows, 4)),
                     dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y32, (unsigned short*)y16, rows, d));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_cast_f32_to_t16(const float* src0, void* dst0, int64_t n0, const float* src1,
                                   void* dst1, int64_t n1, vog_dtype dtype, void* stream) {
  VOG_CHECK_ARG(src0 && dst0 && n0 > 0 && (n0 % 4) == 0 && n1 >= 0 && (n1 % 4) == 0 && (n1 == 0 || (src1 && dst1)));
  const int64_t q = (n0 + n1) / 4;
  const int grid = (int)((q + 255) / 256 < 2048 ? (q + 255) / 256 : 2048);
  VOG_DISPATCH_DTYPE(dtype, ::vog::launch((cast2_kernel<T16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)src0, (u16x4*)dst0, n0 / 4, (const float4*)src1, (u16x4*)dst1, n1 / 4));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_lang_prep(void* zero, int64_t zero_bytes, int64_t ones_bytes, const int64_t* words_ind, const int64_t* word_mask,
                             const int64_t* lens, int32_t* tok, int32_t* rows, int Bn, int T, int nsrl,
                             int seq_len, int vocab_size, const void* emb16, void* a0_frag, int emb_dim,
                             void* stream) {
  LangPrepArgs la; int blocks = 0;
  VOG_TRY(lang_prep_setup(zero, zero_bytes, ones_bytes, words_ind, word_mask, lens, tok, rows, Bn, T, nsrl, seq_len,
                          vocab_size, emb16, a0_frag, emb_dim, &la, &blocks));
  ::vog::launch(lang_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, la);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_prep_fused(void* zero, int64_t zero_bytes, int64_t ones_bytes, const int64_t* words_ind, const int64_t* word_mask,
                              const int64_t* lens, int32_t* tok, int32_t* rows, int Bn, int T, int nsrl,
                              int seq_len, int vocab_size, const void* emb16, void* a0_frag, int emb_dim,
                              const vog_visprep_args* vis, void* stream) {
  LangPrepArgs la; int lblocks = 0, cast_blocks = 0, u_blocks = 0;
  VOG_TRY(lang_prep_setup(zero, zero_bytes, ones_bytes, words_ind, word_mask, lens, tok, rows, Bn, T, nsrl, seq_len,
                          vocab_size, emb16, a0_frag, emb_dim, &la, &lblocks));
  VOG_TRY(vis_prep_setup(vis, &cast_blocks, &u_blocks));
  VOG_DISPATCH_DTYPE(vis->dtype, ::vog::launch((prep_fused_kernel<T16>), dim3(lblocks + cast_blocks + u_blocks),
                     dim3(256), 0, (hipStream_t)stream, la, lblocks, *vis, cast_blocks));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_vis_prep(const vog_visprep_args* a, void* stream) {
  int cast_blocks = 0, u_blocks = 0;
  VOG_TRY(vis_prep_setup(a, &cast_blocks, &u_blocks));
  if (cast_blocks + u_blocks == 0) return 0;
  VOG_DISPATCH_DTYPE(a->dtype, ::vog::launch((vis_prep_kernel<T16>), dim3(cast_blocks + u_blocks), dim3(256), 0,
                     (hipStream_t)stream, *a, cast_blocks));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_splitk_finish(const vog_splitk_prob* p0, const vog_splitk_prob* p1, void* stream) {
  VOG_CHECK_ARG(p0 && p0->slabs && p0->splits > 0 && (p0->N % 4) == 0 && (p0->c32 || p0->c16) && p0->rep >= 1);
  VOG_CHECK_ARG(!p1 || (p1->slabs && p1->splits > 0 && (p1->N % 4) == 0 && (p1->c32 || p1->c16) && p1->rep >= 1));
  SplitkProbs a{};
  a.p[0] = *p0;
  a.blocks0 = (int)(((int64_t)p0->M * (p0->N / 4) + 255) / 256);
  int blocks1 = 0;
  if (p1) { a.p[1] = *p1; blocks1 = (int)(((int64_t)p1->M * (p1->N / 4) + 255) / 256); }
  ::vog::launch(splitk_finish_kernel, dim3(a.blocks0 + blocks1), dim3(256), 0, (hipStream_t)stream, a);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_box_u(const float* props, const float* w_pe, float* u, int n_rows, int H,
                         float vid_w, float vid_h, float nfrm_div, void* stream) {
  VOG_CHECK_ARG(props && w_pe && u && n_rows > 0 && H > 0);
  const int n = n_rows * H;
  ::vog::launch(box_u_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     props, w_pe, u, n_rows, H, vid_w, vid_h, nfrm_div);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_srl_gather(const int64_t* words_ind, const int64_t* word_mask, int32_t* tok,
                              int Bn, int T, int nsrl, int seq_len, int vocab_size, void* stream) {
  VOG_CHECK_ARG(words_ind && word_mask && tok && Bn > 0 && T > 0 && T <= seq_len);
  ::vog::launch(srl_gather_kernel, dim3(ceil_div(Bn * T, 256)), dim3(256), 0,
                     (hipStream_t)stream, words_ind, word_mask, tok, Bn, T, nsrl, seq_len, vocab_size);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_srl_argvec(const float* full, const int64_t* capture, const int64_t* inds_msk,
                              const float* w, const float* bias, float* lang,
                              int Bn, int T, int nsrl, int L, void* stream) {
  VOG_CHECK_ARG(full && capture && inds_msk && w && bias && lang && Bn > 0 && L > 0 && L <= 512 && (L % 8) == 0);
  const dim3 grid(ceil_div(Bn * nsrl, AV_ROWS), ceil_div(L, 16));
  const size_t lds = (size_t)AV_ROWS * 2 * L * sizeof(float);
  if (L == 256) {      // lang_encode_size of the reference configuration
    const size_t lds32 = (size_t)AV_ROWS * 16 * (32 + 4) * sizeof(float);      // (slices 36 floats apart: bank spread)
    static bool av32_attr = false;
    if (!av32_attr) {
      VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(argvec_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      av32_attr = true;
    }
    ::vog::launch(argvec_kernel<32>, grid, dim3(256), lds32, (hipStream_t)stream, full, capture, inds_msk, w, bias, lang, T, nsrl, L, Bn * nsrl);
  } else {
    static bool av_attr = false;
    if (!av_attr && lds > 40 * 1024) {
      VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(argvec_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      av_attr = true;
    }
    ::vog::launch(argvec_kernel<0>, grid, dim3(256), lds, (hipStream_t)stream, full, capture, inds_msk, w, bias, lang, T, nsrl, L, Bn * nsrl);
  }
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_vislang_layout(const vog_vislang_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->vis && a->lang && (a->x32 || a->x16));
  const int64_t rows = (int64_t)a->n_vid * a->nfrm * a->nsrl * a->nppf;
  VOG_DISPATCH_DTYPE(a->dtype, ::vog::launch((vislang_kernel<T16>), dim3((unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, *a));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_qkv_combine(const vog_qkvcomb_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->pv && a->pl && a->q && a->k && a->vt);
  VOG_CHECK_ARG((a->dp % 32) == 0 && a->npad >= a->nsrl * a->nppf && (a->npad % 32) == 0);
  const int items_qk = a->nppf * (a->dp / 8), items_v = a->dp * ((a->nppf + 3) / 4);
  const int chunks = ceil_div(items_qk > items_v ? items_qk : items_v, 256);
  dim3 grid(a->n_vid * a->nfrm * chunks, a->H, 3);
  VOG_DISPATCH_DTYPE(a->dtype, ::vog::launch((qkv_combine_kernel<T16>), grid, dim3(256), 0,
                     (hipStream_t)stream, *a, chunks));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_score_head(const vog_score_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->h1 && a->w2 && a->b2 && a->arg_msk && a->cmp_msk && a->outs && a->outs_eval);
  const int64_t rows = (int64_t)a->n_vid * a->nfrm * a->nsrl * a->nppf;
  ::vog::launch(score_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_pred_cmp_head(const vog_predcmp_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->final_hidden && a->prop_seg && a->outs && a->vidf_outs && a->fin_scores &&
                a->fin_scores_loss && a->verb_ind);
  const size_t sm = (size_t)(a->L + (a->dps - a->dp0) + 256 + a->nsrl) * sizeof(float);
  ::vog::launch(predcmp_kernel, dim3(a->B * a->ncmp), dim3(256), sm, (hipStream_t)stream, *a);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t vog_pred_record_bytes(int ncmp, int nsrl, int nfrm0) {
  return (int64_t)nsrl * ncmp * nfrm0 * 8 * 4 + (int64_t)nsrl * nfrm0 * 8;
}

extern "C" int vog_pred_head(const vog_pred_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->outs_eval && a->props && a->rec && a->B > 0);
  VOG_CHECK_ARG(a->conc_type != VOG_CONC_SEP || a->fin_scores);
  const int64_t rb = vog_pred_record_bytes(a->ncmp, a->nsrl, a->nfrm0);
  const int n1 = a->B * a->nsrl * a->nfrm0 * a->ncmp;
  if (a->nppf0 >= 32) ::vog::launch(pred_wave_kernel, dim3(ceil_div(n1, 4)), dim3(256), 0, (hipStream_t)stream, *a, rb);
  else ::vog::launch(pred_kernel, dim3(ceil_div(n1, 64)), dim3(64), 0, (hipStream_t)stream, *a, rb);
  VOG_LAUNCH_CHECK();
  return 0;
}
