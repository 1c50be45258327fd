p='vognet-pytorch_amd/csrc/elementwise.hip'
s=open(p).read()
# ---- argvec: all loads up front (template on chunks of 2L/64)
old=s[s.index("  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;\n  const int o0 = blockIdx.y * 16 + wid * 4;"):s.index("// ---------------------------------------------------------------------------\n// K4 vis||lang token layout")]
new='''  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int o0 = blockIdx.y * 16 + wid * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // 2L <= 1024: up to 16 strided elements per lane, all loads issued before the FMAs
  constexpr int MAXI = 16;
  float xv[MAXI], wv[4][MAXI];
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int i = lane + it * 64;
    const bool ok = i < 2 * L;
    xv[it] = ok ? (i < L ? x0[i] : x1[i - L]) : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      wv[k][it] = (ok && o0 + k < L) ? w[(int64_t)(o0 + k) * 2 * L + i] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < MAXI; ++it)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += wv[k][it] * xv[it];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0 && o0 + k < L) lang[(int64_t)ba * L + o0 + k] = fmaxf(v + bias[o0 + k], 0.f) * mk;
  }
}

'''
s=s.replace(old,new)
s=s.replace("VOG_CHECK_ARG(full && capture && inds_msk && w && bias && lang && Bn > 0 && L > 0);","VOG_CHECK_ARG(full && capture && inds_msk && w && bias && lang && Bn > 0 && L > 0 && L <= 512);")
# ---- combine: one item per thread, grid covers items
old=s[s.index("template <typename T16>\n__global__ __launch_bounds__(256) void qkv_combine_kernel(vog_qkvcomb_args a) {"):s.index("// ---------------------------------------------------------------------------\n// K7 score head tail")]
new=r'''template <typename T16>
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

'''
s=s.replace(old,new)
old='''  dim3 grid(a->n_vid * a->nfrm, a->H, 3);
  VOG_DISPATCH_DTYPE(a->dtype, hipLaunchKernelGGL((qkv_combine_kernel<T16>), grid, dim3(256), 0,
                     (hipStream_t)stream, *a));'''
new='''  const int items_qk = a->nppf * (a->dp / 8), items_v = a->dp * ((a->nppf + 3) / 4);
  const int chunks = ceil_div(items_qk > items_v ? items_qk : items_v, 256);
  dim3 grid(a->n_vid * a->nfrm * chunks, a->H, 3);
  VOG_DISPATCH_DTYPE(a->dtype, hipLaunchKernelGGL((qkv_combine_kernel<T16>), grid, dim3(256), 0,
                     (hipStream_t)stream, *a, chunks));'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

# ---- gemm tile heuristic: few tiles -> deep ring (nothing else on the CU to hide latency)
p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
old='''  if (ntiles(128, 64) >= 512) return launch_pipe_cfg<T16, 128, 64, 2, EPI>(p, st);
  return launch_pipe_cfg<T16, 64, 64, 2, EPI>(p, st);'''
new='''  if (ntiles(128, 64) >= 512) return launch_pipe_cfg<T16, 128, 64, 2, EPI>(p, st);
  if (ntiles(64, 64) < 256) return launch_pipe_cfg<T16, 64, 64, 4, EPI>(p, st);   // <1 tile per CU: go deep
  return launch_pipe_cfg<T16, 64, 64, 2, EPI>(p, st);'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)
