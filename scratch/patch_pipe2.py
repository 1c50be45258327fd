p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
# 1) swapped MFMA in the pipe kernel main loop
old='''        for (int j = 0; j < FN; ++j) acc[i][j] = mfma32<T16>(fa[i], fb[j], acc[i][j]);
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
new='''        for (int j = 0; j < FN; ++j) acc[i][j] = mfma32<T16>(fb[j], fa[i], acc[i][j]);   // C^T: D[n][m]
    }
  }
  // Swapped operands => each lane owns ONE output row m (= lane&31 of the fragment)
  // and, per register quad, 4 CONSECUTIVE output columns n: 16-byte fp32 / 8-byte
  // 16-bit row-major stores, one row-pointer computation per fragment row.
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nb = n0 + wn * (BN / 2) + j * 32;                 // wave-uniform fragment column base
      if constexpr (EPI == EPI_QKV) qkv_store_swapped<T16>(p, m, nb, hi, acc[i][j]);
      else plain_store_swapped<T16>(p, m, nb, hi, acc[i][j]);
    }
  }
}
'''
assert old in s
s=s.replace(old,new)
# 2) swapped epilogue helpers, placed before the pipe kernel
marker="// ----------------------------------------------------------------------------\n// pipelined kernel: K % 64 == 0"
helpers=r'''// ---- epilogues for the swapped (C^T) accumulator layout -------------------------
// acc[4*g + e] = C[m][nb + 8*g + 4*hi + e]
template <typename T16>
__device__ __forceinline__ void plain_store_swapped(const GemmParams& p, int m, int nb, int hi,
                                                    const f32x16& acc) {
  if (m >= p.M) return;
  const bool vec = (p.N & 3) == 0;
  const float* res = p.residual ? p.residual + (int64_t)m * p.ldr : nullptr;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = nb + 8 * g + 4 * hi;
    if (n >= p.N) continue;
    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    if (vec) {
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (res) {
        const float4 r = *reinterpret_cast<const float4*>(res + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
      if (p.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      u16x4 h;
      if (p.c16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = p.c16_bf16 ? to16<BF16>(v[e]) : to16<F16>(v[e]);
      }
      for (int j = 0; j < p.rep; ++j) {
        const int64_t orow = (int64_t)m * p.rep + j;
        if (p.c32) *reinterpret_cast<float4*>(p.c32 + orow * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (p.c16) *reinterpret_cast<u16x4*>(p.c16 + orow * p.ldc16 + n) = h;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) epilogue_store<T16>(p, m, n + e, v[e]);
    }
  }
}

template <typename T16>
__device__ __forceinline__ void qkv_store_swapped(const GemmParams& p, int m, int nb, int hi,
                                                  const f32x16& acc) {
  nb = __builtin_amdgcn_readfirstlane(nb);
  if (nb >= p.N || m >= p.M) return;
  const int hd = p.H * p.dp;
  const int which = nb / hd;                          // wave-uniform (dp % 32 == 0)
  const int h = (nb - which * hd) / p.dp;
  const int dd0 = nb % p.dp;
  const int s = m / p.ntok;
  const int i = m - s * p.ntok;
  const int64_t sh = (int64_t)s * p.H + h;
  if (which < 2) {
    unsigned short* dst = (which == 0 ? p.q : p.k) + (sh * p.ntok + i) * p.dp + dd0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = to16<T16>(acc[4 * g + e]);
      *reinterpret_cast<u16x4*>(dst + 8 * g + 4 * hi) = v;
    }
  } else {
    // V^T[dd][token]: for a fixed dd the 32 lanes of a half-wave hold consecutive tokens
    unsigned short* dst = p.vt + (sh * p.dp + dd0) * p.npad + i;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        dst[(int64_t)(8 * g + 4 * hi + e) * p.npad] = to16<T16>(acc[4 * g + e]);
  }
}

'''
assert marker in s
s=s.replace(marker, helpers+marker)
# 3) tile selection
old=s[s.index("template <typename T16, int EPI>\nstatic int launch_pipe("):s.index("static bool pipe_ok(")]
new='''template <typename T16, int EPI>
static int launch_pipe(const GemmParams& p, hipStream_t st) {
  // largest tile that still gives every CU a workgroup (256 CUs)
  auto ntiles = [&](int bm, int bn) { return (int64_t)ceil_div(p.M, bm) * ceil_div(p.N, bn); };
  if (ntiles(128, 128) >= 256) return launch_pipe_cfg<T16, 128, 128, 3, EPI>(p, st);
  if (ntiles(128, 64) >= 256) return launch_pipe_cfg<T16, 128, 64, 3, EPI>(p, st);
  return launch_pipe_cfg<T16, 64, 64, 4, EPI>(p, st);
}

'''
s=s.replace(old,new)
# QKV requires N%32 fragments: fine. pipe requires ldc alignment for vector stores
old='''static bool pipe_ok(const GemmParams& p, bool a_f32) {
  return !a_f32 && p.M > 64 && (p.K % 64) == 0 && (p.lda % 8) == 0 && (p.ldw % 8) == 0 &&
         ((uintptr_t)p.a % 16) == 0 && ((uintptr_t)p.w % 16) == 0;
}'''
new='''static bool pipe_ok(const GemmParams& p, bool a_f32) {
  const bool out_ok = (!p.c32 || ((p.ldc % 4) == 0 && ((uintptr_t)p.c32 % 16) == 0)) &&
                      (!p.c16 || ((p.ldc16 % 4) == 0 && ((uintptr_t)p.c16 % 8) == 0)) &&
                      (!p.residual || ((p.ldr % 4) == 0 && ((uintptr_t)p.residual % 16) == 0)) &&
                      (!p.bias || ((uintptr_t)p.bias % 16) == 0);
  return !a_f32 && p.M > 64 && (p.K % 64) == 0 && (p.lda % 8) == 0 && (p.ldw % 8) == 0 &&
         ((uintptr_t)p.a % 16) == 0 && ((uintptr_t)p.w % 16) == 0 && out_ok;
}'''
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
