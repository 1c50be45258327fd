// QkvRowBlockBody (64 rows x 512 columns per workgroup; option qkv_lean, chain_obj_qkv consumer) as it stood at the start of round 6
// Row-block form of the fused QKV projection (vog_qkv_proj with wqkv_p32 set).
//
// The LDS-DMA GEMM covers x[M, K] Wqkv^T with 64x64 tiles: at the M = 800 of a cfg-2 forward that is
// ~350-470 short workgroups which each stage 128 KB of operands for 2 MFMA tiles per wave and spend
// most of their life in prologue/epilogue: ~2000 busy-CU-microseconds per launch. With several
// forwards in flight the chip is busy-CU-time bound, so this form trades latency for occupancy:
// a workgroup owns 64 rows x 512 output columns; the 64 activation rows are staged in LDS ONCE
// (padded rows, the MFMA B operand of "swapped" products as in txtail_dev.h), each of the 8 waves
// streams the weight fragments of its 64 columns straight into registers (vog_pack_w_frag32 order,
// one contiguous KiB per load, read once per workgroup), and the accumulators go through the shared
// Q/K/V^T fragment writer of gemm_dev.h. 52 (obj_tx) / 65 (mul_tx) workgroups per launch.
template <typename T16, int NBW_>
struct QkvRowBlockBody {
  using Params = GemmParams;
  static constexpr int THREADS = 512;
  static constexpr int NBW = NBW_;                    // 32-column blocks per wave (2: 512 columns per workgroup, 1: 256)
  static constexpr int WG_COLS = 8 * NBW * 32;
  static constexpr int EP_BYTES = 8 * 64 * 36 * 4;    // one 64 x 32 fp32 tile (+4 pad) per wave
  static __host__ __device__ constexpr int lds_bytes(int K) {
    const int x = 64 * (K + 8) * 2;
    return x > EP_BYTES ? x : EP_BYTES;
  }
  static __device__ __forceinline__ void run(const GemmParams& p, const BlockCtx& cx, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nrb = (p.M + 63) >> 6, ncg = (p.N + WG_COLS - 1) / WG_COLS;
    const int total = nrb * ncg, per = (total + 7) >> 3;
    // block b runs on XCD b % 8: give every XCD a contiguous range of (column group, row block) ids,
    // column group major, so the workgroups that share an L2 mostly stream the same weights
    const int xcd = cx.bx & 7, xpos = cx.bx >> 3;
    const int v = xcd * per + xpos;
    if (xpos >= per || v >= total) return;
    const int cg = v / nrb, rb = v - cg * nrb;
    const int m0 = rb * 64;
    const int pitch = (p.K + 8) * 2;
    if (p.dep_flags) {
      // the rows of this block come from encoder workgroups of the SAME launch (dispatched before this one): wait for the
      // 2 + (2 or 4) of them that write rows [m0, m0 + 64); bounded (a stuck producer shows as garbage, not as a hang)
      if (tid == 0) {
        const int mlast = (m0 + 63 < p.M ? m0 + 63 : p.M - 1);
        const int s0 = (m0 / p.dep_rep) >> 6, s1 = (mlast / p.dep_rep) >> 6;
        unsigned spins = 0;
        auto wait = [&](int idx) {
          while (__hip_atomic_load(p.dep_flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) break;
          }
        };
        for (int h = 0; h < p.dep_nh0; ++h) wait(rb * 2 + h);
        for (int sb = s0; sb <= s1; ++sb)
          for (int h = 0; h < p.dep_nh1; ++h) wait((p.dep_nb0 + sb) * 2 + h);
      }
      __syncthreads();
    }
    {
      const int cpr = p.K >> 3;
      const unsigned short* a = reinterpret_cast<const unsigned short*>(p.a);
      for (int idx = tid; idx < 64 * cpr; idx += 512) {
        const int r = idx / cpr, c = idx - r * cpr;
        int m = m0 + r;
        m = m < p.M ? m : p.M - 1;
        const int64_t src = p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m;
        *reinterpret_cast<uint4*>(smem + r * pitch + c * 16) = *reinterpret_cast<const uint4*>(a + src * p.lda + c * 8);
      }
    }
    __syncthreads();
    const int blk0 = (cg * 8 + w) * NBW;
    const bool live = blk0 * 32 < p.N;                 // (N / 32) % NBW == 0: a wave's blocks are all in or all out
    f32x16 acc[NBW][2];
    const int KS = p.K >> 4;
    if (live) tail_gemm<T16, NBW, 8, true>(acc, p.w_p32, blk0, 1, KS, ((xpos & 7) * KS) >> 3, smem, pitch, lane);
    __syncthreads();                                   // every wave is done with the activation rows
    if (!live) return;
    float* ep = reinterpret_cast<float*>(smem) + w * (64 * 36);
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
#pragma unroll
      for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(&ep[(rbk * 32 + (lane & 31)) * 36 + 8 * g + 4 * hi]) =
              make_float4(acc[i][rbk][4 * g], acc[i][rbk][4 * g + 1], acc[i][rbk][4 * g + 2], acc[i][rbk][4 * g + 3]);
      qkv_epilogue_tile<T16, 64, 32>(p, ep, m0, (blk0 + i) * 32, lane);
    }
  }
};

template <typename T16, int NBW>
__global__ __launch_bounds__(512) void qkv_rowblock_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char qkvrb_smem[];
  QkvRowBlockBody<T16, NBW>::run(p, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, qkvrb_smem);
}
