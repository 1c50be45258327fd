// Device side of visenc.hip (kernel bodies; also included by pair.hip, which fuses two bodies into one launch).
#pragma once
#include "common.h"

namespace vog {

struct VisEncProb {
  const float* x; const unsigned short* w; const float* bias;
  int M, N, K, rep, col0;        // output rows m*rep + j, columns [col0, col0 + N)
};
struct VisEncParams {
  VisEncProb p[2];
  int tiles0, tiles_all;         // 16-row tiles of problem 0, of both
  float* c32; unsigned short* c16; int64_t ldc; int c16_bf16;
};

template <typename T16>
struct VisEncBody {
  using Params = VisEncParams;
  static constexpr int THREADS = 512;
  static constexpr size_t LDS = (size_t)8 * 2 * 64 * 4 * sizeof(float);
  static __device__ __forceinline__ void run(const VisEncParams& a, const BlockCtx& cx, unsigned char* smem) {
  // 8 waves split K, 2 column tiles (32 columns) per workgroup: with K/8 = 256 (8 k-steps) a wave's
  // whole slice - 16 fp32 row pieces + 16 weight fragments - is requested in ONE round trip.
  // (First form: 4 waves x 64 columns, 4 rounds of 4 k-steps, one workgroup per CU: 17 us, every
  // round's HBM latency exposed.)
  float (*red)[2][64][4] = reinterpret_cast<float (*)[2][64][4]>(smem);   // [wave][col tile][lane][reg]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware item order: the 8 column slices of a row tile share an XCD (speed only)
  const int xcd = cx.bx & 7, idx = cx.bx >> 3;
  const int tile = (idx >> 3) * 8 + xcd, slice = idx & 7;
  if (tile >= a.tiles_all) return;
  const bool second = tile >= a.tiles0;
  // (explicit selects: indexing the by-value array with a runtime value would put it in scratch)
  const float* qx = second ? a.p[1].x : a.p[0].x;
  const unsigned short* qw = second ? a.p[1].w : a.p[0].w;
  const float* qb = second ? a.p[1].bias : a.p[0].bias;
  const int qM = second ? a.p[1].M : a.p[0].M, qN = second ? a.p[1].N : a.p[0].N;
  const int qK = second ? a.p[1].K : a.p[0].K, qrep = second ? a.p[1].rep : a.p[0].rep;
  const int qcol0 = second ? a.p[1].col0 : a.p[0].col0;
  const int m0 = (second ? tile - a.tiles0 : tile) * 16;
  const int n0 = slice * 32;
  if (n0 >= qN) return;
  const int ksteps = qK >> 5;                  // K % 256 == 0
  const int kw = ksteps >> 3;                  // k-steps per wave (contiguous slice)
  const int ml = lane & 15, kg = lane >> 4;
  int m = m0 + ml;
  m = m < qM ? m : qM - 1;
  const float* xr = qx + (int64_t)m * qK + (w * kw) * 32 + kg * 8;
  const u16x8* wf = reinterpret_cast<const u16x8*>(qw) + ((int64_t)(n0 >> 4) * ksteps + w * kw) * 64 + lane;
  f32x4 acc[2];
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int UN = 8;
  for (int ks = 0; ks < kw; ks += UN) {
    float4 xa[UN][2];
    u16x8 wq[UN][2];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const bool ok = ks + j < kw;
      xa[j][0] = ok ? *reinterpret_cast<const float4*>(xr + (ks + j) * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
      xa[j][1] = ok ? *reinterpret_cast<const float4*>(xr + (ks + j) * 32 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        wq[j][ct] = ok ? wf[((int64_t)ct * ksteps + ks + j) * 64] : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const u16x8 af = {to16<T16>(xa[j][0].x), to16<T16>(xa[j][0].y), to16<T16>(xa[j][0].z), to16<T16>(xa[j][0].w),
                        to16<T16>(xa[j][1].x), to16<T16>(xa[j][1].y), to16<T16>(xa[j][1].z), to16<T16>(xa[j][1].w)};
      acc[0] = mfma16<T16>(af, wq[j][0], acc[0]);
      acc[1] = mfma16<T16>(af, wq[j][1], acc[1]);
    }
  }
  *reinterpret_cast<f32x4*>(&red[w][0][lane][0]) = acc[0];
  *reinterpret_cast<f32x4*>(&red[w][1][lane][0]) = acc[1];
  __syncthreads();
  // waves 0/1 finish column tile 0/1: lane = (row group, column), 4 rows per lane
  if (w >= 2) return;
  const int col = n0 + w * 16 + ml;
  if (col >= qN) return;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ww = 0; ww < 8; ++ww) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(&red[ww][w][lane][0]);
    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
  }
  const float b = qb[col];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + kg * 4 + r;
    if (row >= qM) continue;
    const float o = fmaxf(v[r] + b, 0.f);
    const unsigned short h = a.c16_bf16 ? to16<BF16>(o) : to16<F16>(o);
    for (int j = 0; j < qrep; ++j) {
      const int64_t off = ((int64_t)row * qrep + j) * a.ldc + qcol0 + col;
      if (a.c32) a.c32[off] = o;
      if (a.c16) a.c16[off] = h;
    }
  }
}
};

template <typename T16>
__global__ __launch_bounds__(512) void vis_enc_kernel(VisEncParams a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ve_smem[];
  VisEncBody<T16>::run(a, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, ve_smem);
}

}  // namespace vog
