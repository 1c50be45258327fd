p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
a=s.index("template <typename T16, bool A_F32, int SK_CH>\n__global__ __launch_bounds__(256) void gemm_skinny")
b=s.index("}\n", s.index("epilogue_store<T16>(p, row, n, v);", a))
# find the end of the kernel: the closing of function after that
end=s.index("\n}\n", b)+3
new_kernel='''template <typename T16, bool A_F32, int SK_CH, int NT>
__global__ __launch_bounds__(256) void gemm_skinny(GemmParams p) {
  // NT 16-column tiles per workgroup: every A fragment a wave loads feeds NT MFMAs, so the
  // L2 traffic for A (re-read by every workgroup) drops by NT; used when there are enough
  // column tiles to still fill the chip.
  __shared__ float red[4][NT][4][64][4];      // [wave][ntile][mtile][lane][reg]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ct0 = blockIdx.x * NT;            // first 16-column tile
  const int kg = (lane >> 4) * 8;
  const int mt_all = (p.M + 15) / 16;         // <= 4
  // grid.y > 1: one 16-row tile of A per workgroup (few output columns: parallelism
  // matters more than re-reading the small W panel)
  const int mt_lo = gridDim.y > 1 ? blockIdx.y : 0;
  const int mt_n = gridDim.y > 1 ? mt_lo + 1 : mt_all;
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
  for (int base = wid; base < ksteps; base += 4 * SK_CH) {
    u16x8 fw[NT][SK_CH];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < SK_CH; ++c) {
        const int ks = base + c * 4;
        if (p.w_frag) {   // one contiguous KiB per (column tile, k-step)
          u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          fw[t][c] = (ks < ksteps && n_ok[t]) ? *reinterpret_cast<const u16x8*>(
                                    p.w + (((int64_t)(ct0 + t) * ksteps + ks) * 64 + lane) * 8) : z;
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
          const int ks = base + c * 4;
          if (!A_F32 && p.a_frag) {   // contiguous KiB per (row tile, k-step); pad rows are zero-filled
            u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            fa[c] = ks < ksteps ? *reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(p.a) +
                                      (((int64_t)mt * ksteps + ks) * 64 + lane) * 8) : z;
          } else {
            fa[c] = load_a_chunk<T16, A_F32>(p.a, a_off[mt], ks * 32 + kg, a_ok[mt] && ks < ksteps);
          }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int c = 0; c < SK_CH; ++c) acc[t][mt] = mfma16<T16>(fa[c], fw[t][c], acc[t][mt]);
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
  // wave w finishes m-tile w
  const int mt = wid;
  if (mt >= mt_lo && mt < mt_n) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = red[0][t][mt][lane][r] + red[1][t][mt][lane][r] + red[2][t][mt][lane][r] + red[3][t][mt][lane][r];
        const int row = mt * 16 + (lane >> 4) * 4 + r;
        epilogue_store<T16>(p, row, (ct0 + t) * 16 + (lane & 15), v);
      }
  }
}
'''
s=s[:a]+new_kernel+s[end:]
old=s[s.index("    const int ncol = ceil_div(p.N, 16);\n    dim3 grid(ncol, ncol < 128"):s.index("    VOG_LAUNCH_CHECK();\n    return 0;\n  }\n  if (g->a_is_f32) return launch_tiled<T16, true, EPI_PLAIN>")]
new='''    const int ncol = ceil_div(p.N, 16);
    // (a 16-deep weight prefetch measured SLOWER: 26 vs 18.6 us at M=48,N=8192,K=2048, 230 VGPRs)
    static const int nt_env = getenv("VOG_SKINNY_NT") ? atoi(getenv("VOG_SKINNY_NT")) : 0;
    const int nt = nt_env ? nt_env : (ncol >= 512 ? 2 : 1);
    if (nt == 2) {
      dim3 grid(ceil_div(ncol, 2), 1);
      if (g->a_is_f32) hipLaunchKernelGGL((gemm_skinny<T16, true, 8, 2>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((gemm_skinny<T16, false, 8, 2>), grid, dim3(256), 0, st, p);
    } else {
      dim3 grid(ncol, ncol < 128 ? ceil_div(p.M, 16) : 1);
      if (g->a_is_f32) hipLaunchKernelGGL((gemm_skinny<T16, true, 8, 1>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((gemm_skinny<T16, false, 8, 1>), grid, dim3(256), 0, st, p);
    }
'''
s=s.replace(old,new,1)
open(p,'w').write(s)
