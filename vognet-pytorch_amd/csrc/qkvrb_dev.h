// Row-block form of the fused QKV projection for MANY rows (vog_qkv_proj with wqkv_p32 set; p100: M = 16 000): one workgroup
// per 64 rows walks all 3*H*dp output columns (QkvRowAllBody below). The 64-row x 512-column form of rounds 2-5 (a third of the
// tiled GEMM's busy-CU time at twice its latency; measured -3 % at cfg 2 in round 6, 50.7 vs 52.2 k queries/s) was removed in
// round 6: scratch/negatives/r6_pruned/.
#pragma once
#include "gemm_dev.h"
#include "txtail_dev.h"

namespace vog {

// ---------------------------------------------------------------------------------------------------
// "All columns" form (round 5) for many rows (p100: M = 16 000): the row-block form above re-stages the 64 activation rows
// once per 256-column group (7-9 x per row block) and spends most of a workgroup's life in staging and epilogue - its
// 64 MFMAs per wave are < 1 us (profiles/round5_qkv_p100.md: the tiled GEMM's DMA, MFMA and epilogue phases ADD UP,
// 27 + 15 + 14 us). Here ONE workgroup per row block stages the rows once and walks ALL 3*H*dp output columns: wave w
// owns the 32-column blocks w, w + 8, ...; per block: GEMM over K with the weights streamed through registers (the
// first k-steps of the NEXT block requested before this block's epilogue: tail_prime), accumulators parked in a
// wave-private LDS tile, shared Q / K / V^T fragment writer. No workgroup barrier after the staging: the waves never
// meet again, so one wave's epilogue stores overlap the other waves' MFMAs. 250 workgroups at p100 = one per CU.
// 
// ---------------------------------------------------------------------------------------------------
template <typename T16>
struct QkvRowAllBody {
  using Params = GemmParams;
  static constexpr int THREADS = 512;
  static constexpr int EP_W = 64 * 36 * 4;                       // one 64 x 32 fp32 tile (+4 pad) per wave
  static __host__ __device__ constexpr int lds_bytes(int K) { return 64 * (K + 8) * 2 + 8 * EP_W; }
  static __device__ __forceinline__ void run(const GemmParams& p, const BlockCtx& cx, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = cx.bx * 64;
    if (m0 >= p.M) return;
    const int pitch = (p.K + 8) * 2;
    {
      const int cpr = p.K >> 3;
      const unsigned short* a = reinterpret_cast<const unsigned short*>(p.a);
      stage_batched<8, 512, uint4>(64 * cpr, tid,
          [&](int idx) {
            const int r = idx / cpr, c = idx - r * cpr;
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            const int64_t src = p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m;
            return *reinterpret_cast<const uint4*>(a + src * p.lda + c * 8);
          },
          [&](int idx, const uint4& v) {
            const int r = idx / cpr, c = idx - r * cpr;
            *reinterpret_cast<uint4*>(smem + r * pitch + c * 16) = v;
          });
    }
    const int KS = p.K >> 4, nblk = p.N >> 5;
    const int rot = (((cx.bx >> 3) & 7) * KS) >> 3;                // workgroups of one XCD start at different k
    u16x8 wq[8][1];
    if (w < nblk) tail_prime<1, 8>(wq, p.w_p32, w, 1, KS, rot, lane);
    __syncthreads();                                               // the rows are staged; from here on the waves are independent
    float* ep = reinterpret_cast<float*>(smem + 64 * pitch) + w * (64 * 36);
    for (int blk = w; blk < nblk; blk += 8) {
      f32x16 acc[1][2];
      tail_gemm<T16, 1, 8, true, 0, 2, true>(acc, p.w_p32, blk, 1, KS, rot, smem, pitch, lane, wq);
      const int nxt = blk + 8 < nblk ? blk + 8 : blk;              // (unconditional: the last prime re-reads its own block)
      tail_prime<1, 8>(wq, p.w_p32, nxt, 1, KS, rot, lane);
      // ---- epilogue. Plain (non-structured) Q / K blocks go from the accumulator registers straight to their fragment slots:
      // in the swapped product a lane holds ONE token and 4-column strips, the two half-waves hold the two 8-byte halves of the
      // same 16-byte fragment chunk, so one store instruction covers 32 whole chunks = 512 contiguous bytes - no LDS round
      // trip. V^T blocks need the transpose (a fragment lane = one head column, 8 tokens): parked in the wave's LDS tile and,
      // when a 16-token group never straddles a sequence (ntok % 16 == 0: p100), written as whole 16-byte fragment chunks -
      // 4 store instructions per block instead of 32 two-byte stores per lane. Everything else: the shared writer.
      const int nb = blk * 32, hd = p.H * p.dp;
      const int which = nb / hd, hh = (nb - which * hd) / p.dp, dd0 = nb % p.dp;
      const bool kv_vis = p.pl && p.st_kv_vis && which >= 1;
      const bool plain = !p.pl || kv_vis;
      const int ntok_w = kv_vis ? p.st_nppf : p.ntok, npad_w = kv_vis ? p.npad_kv : p.npad;
      if (plain && which < 2) {
        unsigned short* base = which == 0 ? p.q : p.k;
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk) {
          const int m = m0 + rbk * 32 + (lane & 31);
          if (m >= p.M) continue;
          const int sq = fast_div(m, p.fdT_mul, p.fdT_shr), tok = m - sq * ntok_w;
          unsigned short* dst = base + ((int64_t)sq * p.H + hh) * npad_w * p.dp;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<u16x4*>(dst + frag_qk(tok, dd0 + 8 * g + 4 * hi, p.dp)) =
                u16x4{to16<T16>(acc[0][rbk][4 * g]), to16<T16>(acc[0][rbk][4 * g + 1]), to16<T16>(acc[0][rbk][4 * g + 2]),
                      to16<T16>(acc[0][rbk][4 * g + 3])};
        }
        continue;
      }
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(&ep[(rbk * 32 + (lane & 31)) * 36 + 8 * g + 4 * hi]) =
              make_float4(acc[0][rbk][4 * g], acc[0][rbk][4 * g + 1], acc[0][rbk][4 * g + 2], acc[0][rbk][4 * g + 3]);
      if (plain && which == 2 && (ntok_w & 15) == 0 && (p.M & 15) == 0) {
        const int ddl = lane & 31;
#pragma unroll
        for (int tg = 0; tg < 4; ++tg) {
          const int mg = m0 + tg * 16;                             // 16 consecutive tokens of ONE sequence, 16-aligned in it
          if (mg >= p.M) continue;
          const int sq = fast_div(mg, p.fdT_mul, p.fdT_shr), t0 = mg - sq * ntok_w;
          unsigned short v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = to16<T16>(ep[(tg * 16 + 8 * (j >> 2) + 4 * hi + (j & 3)) * 36 + ddl]);
          unsigned short* dst = p.vt + ((int64_t)sq * p.H + hh) * npad_w * p.dp +
                                (((((int64_t)(t0 >> 5) * (p.dp >> 5) + (dd0 >> 5)) * 2 + ((t0 & 31) >> 4)) * 64 + (hi << 5) + ddl) << 3);
          *reinterpret_cast<u16x8*>(dst) = u16x8{v8[0], v8[1], v8[2], v8[3], v8[4], v8[5], v8[6], v8[7]};
        }
        continue;
      }
      qkv_epilogue_tile<T16, 64, 32>(p, ep, m0, blk * 32, lane);
    }
  }
};

template <typename T16>
__global__ __launch_bounds__(512) void qkv_rowall_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char qkvra_smem[];
  QkvRowAllBody<T16>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, qkvra_smem);
}

}  // namespace vog
