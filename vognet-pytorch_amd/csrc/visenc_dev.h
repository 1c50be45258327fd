// Device side of visenc.hip (kernel bodies; also included by pair.hip, which fuses two bodies into one launch).
#pragma once
#include "common.h"

namespace vog {

struct VisEncProb {
  const float* x; const unsigned short* w; const float* bias;
  int M, N, K, rep, col0;        // output rows m*rep + j, columns [col0, col0 + N)
  const unsigned short* w_lo;    // round 6 (stream form, SPLIT): 16-bit remainder of the fp32 weights, same fragment order
};
struct VisEncParams {
  VisEncProb p[2];
  int tiles0, tiles_all;         // 16-row tiles of problem 0, of both
  float* c32; unsigned short* c16; int64_t ldc; int c16_bf16;
  int rep_first_only;            // 1: only replica j = 0 of every row is written here (seg_replicate_kernel writes the rest)
  unsigned short* c16_lo;        // round 6 (SPLIT): 16-bit remainder of the output rows (c16 + c16_lo = the fp32 value to ~2^-22)
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

// ---------------------------------------------------------------------------------------------------
// "Stream" form (round 5) for many rows (p100: 16 000 proposal rows x 2048 fp32 = 131 MB per forward): the lean form's
// chunk loop keeps ONE chunk of fp32 rows in flight per workgroup (64 KB per CU) and waits out the HBM latency once per
// chunk - 66.7 us for 133.7 MB = 2.0 TB/s (profiles/round4_pmc_cfg4.md). Same tiling (64 rows x 128 columns per
// workgroup, 8 waves x 16 columns, A converted to the MFMA operand type on the way into LDS), but K chunks of 128 and
// DEPTH of them requested ahead through DEPTH register sets (fp32 rows and weight fragments alike): the wait for chunk
// c + DEPTH - 1 overlaps DEPTH - 1 chunks of conversion + MFMA. Measured: what pays is OCCUPANCY, not depth - DEPTH = 2
// fits two workgroups on a CU (45 us = 2.9 TB/s), 3 / 4 sets of registers leave room for one (52 / 55 us). No load is
// conditional (chunk indices past the end are clamped: hipcc answers a conditional prefetch with s_waitcnt vmcnt(0)).
// Same k order per output column as the lean form within a chunk; chunks are 128 instead of 256 deep, so the fp32
// summation order is that of the lean form with VOG_VE_KC = 128.
// ---------------------------------------------------------------------------------------------------
#ifndef VOG_VS_DEPTH
#define VOG_VS_DEPTH 2      // measured at p100 (scratch/r5_ve.sh): 2: 45.0 us (~110 registers: two workgroups per CU), 3: 52.2, 4: 54.6; lean form 57.6
#endif
// SPLIT (round 6): hi + lo operands - the fp32 rows are split into t16(x) and t16(x - t16(x)) on the way into LDS (two images per
// chunk), the weights arrive as two fragment streams, a k-step is three MFMAs (x.w + x_lo.w + x.w_lo): encoder outputs with
// fp32-grade operand precision for checkpoints whose attention logits amplify a 2^-11 input error past the 1e-3 bound. The kernel
// stays bound by the fp32 feature stream; the outputs carry a 16-bit remainder copy (c16_lo) for the hi + lo QKV projection.
template <typename T16, int DEPTH = VOG_VS_DEPTH, bool SPLIT = false>
struct VisEncStreamBody {
  using Params = VisEncParams;
  static constexpr int THREADS = 512;
  static constexpr int RB = 64, KC = 128, KSC = KC / 32;    // 4 k-steps per chunk
  static constexpr int PIECES = KC / 8, RPP = THREADS / PIECES, NPASS = RB / RPP;   // 16 pieces per row, 32 rows per pass, 2 passes
  static constexpr int IMG = (RB / 16) * KSC * 1024;        // one A-chunk image (fragment order): 16 KB
  static constexpr size_t LDS = (size_t)2 * IMG * (SPLIT ? 2 : 1);   // (SPLIT: the remainder images behind the two hi images)

  static __device__ __forceinline__ void run(const VisEncParams& a, const BlockCtx& cx, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb0 = (a.tiles0 + 3) >> 2, nb_all = nb0 + ((a.tiles_all - a.tiles0 + 3) >> 2);
    // the two column halves of a row block sit 8 block ids apart = on the same XCD (second read of the rows: that L2)
    const int grp = cx.bx >> 4, pos = cx.bx & 15;
    const int blk = grp * 8 + (pos & 7), half0 = pos >> 3;
    if (blk >= nb_all) return;
    const bool second = blk >= nb0;
    const float* qx = second ? a.p[1].x : a.p[0].x;
    const unsigned short* qw = second ? a.p[1].w : a.p[0].w;
    const float* qb = second ? a.p[1].bias : a.p[0].bias;
    const int qM = second ? a.p[1].M : a.p[0].M, qN = second ? a.p[1].N : a.p[0].N;
    const int qK = second ? a.p[1].K : a.p[0].K, qrep = second ? a.p[1].rep : a.p[0].rep;
    const int qcol0 = second ? a.p[1].col0 : a.p[0].col0;
    const int m0 = (second ? blk - nb0 : blk) * RB;
    if (half0 * 128 >= qN) return;
    const int ksteps = qK >> 5, nchunk = qK / KC;            // K % 256 == 0
    const int n0 = half0 * 128 + w * 16;
    const bool n_ok = n0 < qN;
    const u16x8* wf = reinterpret_cast<const u16x8*>(qw) + ((int64_t)((n_ok ? n0 : 0) >> 4) * ksteps) * 64 + lane;
    const unsigned short* qwl = second ? a.p[1].w_lo : a.p[0].w_lo;
    const u16x8* wfl = SPLIT ? reinterpret_cast<const u16x8*>(qwl) + ((int64_t)((n_ok ? n0 : 0) >> 4) * ksteps) * 64 + lane : wf;
    const int pr = tid / PIECES, pc = tid % PIECES;
    const float* xrow[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      int m = m0 + ps * RPP + pr;
      m = m < qM ? m : qM - 1;
      xrow[ps] = qx + (int64_t)m * qK + pc * 8;
    }
    f32x4 acc[RB / 16];
#pragma unroll
    for (int mt = 0; mt < RB / 16; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 xa[DEPTH][NPASS][2];
    u16x8 wq[DEPTH][KSC];
    u16x8 wql[SPLIT ? DEPTH : 1][SPLIT ? KSC : 1];
    auto request = [&](int set, int c) {                     // fp32 row pieces + weight fragments of chunk c (clamped)
      const int cc = c < nchunk ? c : nchunk - 1;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const float4* src = reinterpret_cast<const float4*>(xrow[ps] + cc * KC);
        xa[set][ps][0] = src[0];                             // (plain loads: the other column half of the row block, on the same
        xa[set][ps][1] = src[1];                             // XCD, reads the same rows from that L2)
      }
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) {
        wq[set][ks] = wf[(cc * KSC + ks) * 64];
        if constexpr (SPLIT) wql[set][ks] = wfl[(cc * KSC + ks) * 64];
      }
    };
    auto store_a = [&](int set, int c) {
      unsigned char* img = smem + (size_t)(c & 1) * IMG;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const u16x8 h = {to16<T16>(xa[set][ps][0].x), to16<T16>(xa[set][ps][0].y), to16<T16>(xa[set][ps][0].z), to16<T16>(xa[set][ps][0].w),
                         to16<T16>(xa[set][ps][1].x), to16<T16>(xa[set][ps][1].y), to16<T16>(xa[set][ps][1].z), to16<T16>(xa[set][ps][1].w)};
        const int ks = pc >> 2, kgp = pc & 3;
        const int rl = ps * RPP + pr;
        *reinterpret_cast<u16x8*>(img + (((rl >> 4) * KSC + ks) * 64 + kgp * 16 + (rl & 15)) * 16) = h;
        if constexpr (SPLIT) {
          const float xs[8] = {xa[set][ps][0].x, xa[set][ps][0].y, xa[set][ps][0].z, xa[set][ps][0].w,
                               xa[set][ps][1].x, xa[set][ps][1].y, xa[set][ps][1].z, xa[set][ps][1].w};
          u16x8 l;
#pragma unroll
          for (int j = 0; j < 8; ++j) l[j] = to16<T16>(xs[j] - from16<T16>(h[j]));
          *reinterpret_cast<u16x8*>(img + 2 * IMG + (((rl >> 4) * KSC + ks) * 64 + kgp * 16 + (rl & 15)) * 16) = l;
        }
      }
    };
    auto mfmas = [&](int set, int c) {
      const unsigned char* img = smem + (size_t)(c & 1) * IMG;
      if constexpr (SPLIT) {
        // x.w + x_lo.w + x.w_lo per k-step (the stream of fp32 rows, not the matrix pipe, bounds this kernel)
#pragma unroll
        for (int mt = 0; mt < RB / 16; ++mt) {
          u16x8 fh[KSC], fl[KSC];
#pragma unroll
          for (int j = 0; j < KSC; ++j) {
            fh[j] = *reinterpret_cast<const u16x8*>(img + ((mt * KSC + j) * 64 + lane) * 16);
            fl[j] = *reinterpret_cast<const u16x8*>(img + 2 * IMG + ((mt * KSC + j) * 64 + lane) * 16);
          }
#pragma unroll
          for (int j = 0; j < KSC; ++j) {
            acc[mt] = mfma16<T16>(fh[j], wq[set][j], acc[mt]);
            acc[mt] = mfma16<T16>(fl[j], wq[set][j], acc[mt]);
            acc[mt] = mfma16<T16>(fh[j], wql[set][j], acc[mt]);
          }
        }
        return;
      }
      u16x8 fa[KSC], fb[KSC];
      auto rd = [&](u16x8 (&f)[KSC], int mt) {
#pragma unroll
        for (int j = 0; j < KSC; ++j) f[j] = *reinterpret_cast<const u16x8*>(img + ((mt * KSC + j) * 64 + lane) * 16);
      };
      rd(fa, 0);
#pragma unroll
      for (int mt = 0; mt < RB / 16; mt += 2) {
        __builtin_amdgcn_sched_barrier(0);
        rd(fb, mt + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KSC; ++j) acc[mt] = mfma16<T16>(fa[j], wq[set][j], acc[mt]);
        __builtin_amdgcn_sched_barrier(0);
        if (mt + 2 < RB / 16) rd(fa, mt + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KSC; ++j) acc[mt + 1] = mfma16<T16>(fb[j], wq[set][j], acc[mt + 1]);
      }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) request(d, d);
    int c0 = 0;
    for (; c0 + DEPTH <= nchunk; c0 += DEPTH) {              // (no condition around a load in the steady state)
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        const int c = c0 + j;
        store_a(j, c);                                       // image (c & 1) was last read two chunks ago (a barrier since)
        request((j + DEPTH - 1) % DEPTH, c + DEPTH - 1);     // the set chunk c - 1 has just released
        lds_barrier();
        mfmas(j, c);
      }
    }
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j) {                    // nchunk % DEPTH left-over chunks (their requests are out already)
      const int c = c0 + j;
      if (c < nchunk) {
        store_a(j, c);
        lds_barrier();
        mfmas(j, c);
      }
    }
    // D[row = 4*(lane>>4) + reg][col = lane & 15]
    const int col = n0 + (lane & 15);
    if (n_ok && col < qN) {
      const float b = qb[col];
      const int nrep = a.rep_first_only ? 1 : qrep;
#pragma unroll
      for (int mt = 0; mt < RB / 16; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + mt * 16 + (lane >> 4) * 4 + r;
          if (row >= qM) continue;
          const float o = fmaxf(acc[mt][r] + b, 0.f);
          const unsigned short hv = a.c16_bf16 ? to16<BF16>(o) : to16<F16>(o);
          const unsigned short lv = a.c16_bf16 ? to16<BF16>(o - from16<BF16>(hv)) : to16<F16>(o - from16<F16>(hv));
          for (int j = 0; j < nrep; ++j) {
            const int64_t off = ((int64_t)row * qrep + j) * a.ldc + qcol0 + col;
            if (a.c32) a.c32[off] = o;
            if (a.c16) a.c16[off] = hv;
            if (SPLIT && a.c16_lo) a.c16_lo[off] = lv;
          }
        }
    }
  }
};

template <typename T16, bool SPLIT = false>
__global__ __launch_bounds__(512) void vis_enc_stream_kernel(VisEncParams a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vs_smem[];
  VisEncStreamBody<T16, VOG_VS_DEPTH, SPLIT>::run(a, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, vs_smem);
}

}  // namespace vog
