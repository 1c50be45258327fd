p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
marker = "// ----------------------------------------------------------------------------\n// skinny kernel (M <= 64, K % 32 == 0)"
pipe = r'''// ----------------------------------------------------------------------------
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
template <typename T16, int BM, int BN, int STAGES, int EPI>
__global__ __launch_bounds__(256) void gemm_pipe(GemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int ROWS = BM + BN;
  constexpr int STAGE_BYTES = ROWS * 128;
  constexpr int LPT = ROWS / 32;                   // DMA instructions per wave per tile
  constexpr int FM = BM / 64, FN = BN / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = bid / nbn, bn = bid % nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // per-lane source pointers of this wave's DMA instructions (advance by 64 halfwords per tile)
  const unsigned short* gsrc[LPT];
#pragma unroll
  for (int i = 0; i < LPT; ++i) {
    const int rr = (wid * LPT + i) * 8 + (lane >> 3);        // row in the combined [A | W] tile
    const int c = (lane & 7) ^ ((rr >> 1) & 7);              // source chunk for LDS chunk lane&7
    if (rr < BM) {
      int m = m0 + rr;
      m = m < p.M ? m : p.M - 1;                             // clamp: rows >= M are discarded later
      const int64_t src = p.a_rows ? (int64_t)p.a_rows[m] : (int64_t)m;
      gsrc[i] = reinterpret_cast<const unsigned short*>(p.a) + src * p.lda + c * 8;
    } else {
      int n = n0 + rr - BM;
      n = n < p.N ? n : p.N - 1;
      gsrc[i] = p.w + (int64_t)n * p.ldw + c * 8;
    }
  }
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(gsrc[i] + (int64_t)kt * 64),
          (__attribute__((address_space(3))) void*)(smem + stage * STAGE_BYTES + (wid * LPT + i) * 1024),
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

  const int nk = p.K / 64;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);
  for (int kt = 0; kt < nk; ++kt) {
    // tiles in flight now: kt .. min(kt+STAGES-2, nk-1). Retire tile kt only.
    const int ahead = (nk - 1 - kt) < (STAGES - 2) ? (nk - 1 - kt) : (STAGES - 2);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
    const unsigned char* st = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u16x8 fa[FM], fb[FN];
      const int g = ks * 2 + hi;
#pragma unroll
      for (int i = 0; i < FM; ++i)
        fa[i] = *reinterpret_cast<const u16x8*>(st + a_row[i] * 128 + ((g ^ ((a_row[i] >> 1) & 7)) << 4));
#pragma unroll
      for (int j = 0; j < FN; ++j)
        fb[j] = *reinterpret_cast<const u16x8*>(st + b_row[j] * 128 + ((g ^ ((b_row[j] >> 1) & 7)) << 4));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = mfma32<T16>(fa[i], fb[j], acc[i][j]);
    }
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

'''
s = s.replace(marker, pipe + marker)
old = '''template <typename T16, bool A_F32, int EPI>
static int launch_tiled(const GemmParams& p, hipStream_t st) {'''
new = '''template <typename T16, int BM, int BN, int STAGES, int EPI>
static int launch_pipe_cfg(const GemmParams& p, hipStream_t st) {
  constexpr size_t lds = (size_t)STAGES * (BM + BN) * 128;
  auto kern = gemm_pipe<T16, BM, BN, STAGES, EPI>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}

template <typename T16, int EPI>
static int launch_pipe(const GemmParams& p, hipStream_t st) {
  const int64_t t128 = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 128);
  if (t128 >= 400) return launch_pipe_cfg<T16, 128, 128, 3, EPI>(p, st);
  return launch_pipe_cfg<T16, 64, 64, 4, EPI>(p, st);
}

static bool pipe_ok(const GemmParams& p, bool a_f32) {
  return !a_f32 && p.M > 64 && (p.K % 64) == 0 && (p.lda % 8) == 0 && (p.ldw % 8) == 0 &&
         ((uintptr_t)p.a % 16) == 0 && ((uintptr_t)p.w % 16) == 0;
}

template <typename T16, bool A_F32, int EPI>
static int launch_tiled(const GemmParams& p, hipStream_t st) {
  if (pipe_ok(p, A_F32)) return launch_pipe<T16, EPI>(p, st);'''
assert old in s
s = s.replace(old, new)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/elementwise.hip'
s=open(p).read()
old = s[s.index("__global__ __launch_bounds__(256) void argvec_kernel("):s.index("// ---------------------------------------------------------------------------\n// K4 vis||lang token layout")]
new = '''__global__ __launch_bounds__(256) void argvec_kernel(const float* __restrict__ full,
                                                     const int64_t* __restrict__ capture,
                                                     const int64_t* __restrict__ msk,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ lang, int T, int nsrl, int L) {
  // grid (sentence*arg, L/16): each wave owns 4 outputs with 4 independent
  // accumulators, so all of its weight-row loads are in flight together
  const int ba = blockIdx.x;                   // b*nsrl + a
  const int b = ba / nsrl;
  int64_t c0 = capture[(int64_t)ba * 2], c1 = capture[(int64_t)ba * 2 + 1];
  c0 = c0 < 0 ? 0 : (c0 >= T ? T - 1 : c0);
  c1 = c1 < 0 ? 0 : (c1 >= T ? T - 1 : c1);
  const float* x0 = full + ((int64_t)b * T + c0) * L;
  const float* x1 = full + ((int64_t)b * T + c1) * L;
  const float mk = (float)msk[ba];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int o0 = blockIdx.y * 16 + wid * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < 2 * L; i += 64) {
    const float xv = i < L ? x0[i] : x1[i - L];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (o0 + k < L) acc[k] += w[(int64_t)(o0 + k) * 2 * L + i] * xv;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0 && o0 + k < L) lang[(int64_t)ba * L + o0 + k] = fmaxf(v + bias[o0 + k], 0.f) * mk;
  }
}

'''
s = s.replace(old, new)
s = s.replace("hipLaunchKernelGGL(argvec_kernel, dim3(Bn * nsrl), dim3(256), 2 * L * sizeof(float),","hipLaunchKernelGGL(argvec_kernel, dim3(Bn * nsrl, ceil_div(L, 16)), dim3(256), 0,")
open(p,'w').write(s)
