import re
# ---------------- header
p='include/vog_hip.h'
s=open(p).read()
old='''  int M, N, K; int relu; int rep; vog_dtype dtype;
  int c16_dtype;               /* vog_dtype of c16, or -1 = same as dtype */
} vog_gemm_args;'''
new='''  int M, N, K; int relu; int rep; vog_dtype dtype;
  int c16_dtype;               /* vog_dtype of c16, or -1 = same as dtype */
  /* optional output-row scatter: column n belongs to segment n / out_rows_ncol and
   * row m of that segment is written to row out_rows[seg*M + m] (< 0: dropped).
   * Used to emit the LSTM input projections directly in (direction, step) order. */
  const int32_t* out_rows; int out_rows_ncol;
} vog_gemm_args;'''
assert old in s; s=s.replace(old,new)
old='''int vog_srl_gather(const int64_t* words_ind, const int64_t* word_mask, int32_t* tok,
                   int Bn, int T, int nsrl, int seq_len, int vocab_size, void* stream);'''
new='''int vog_srl_gather(const int64_t* words_ind, const int64_t* word_mask, int32_t* tok,
                   int Bn, int T, int nsrl, int seq_len, int vocab_size, void* stream);
/* Packed-sequence schedule of the BiLSTM: rows[dir*Bn*T + b*T + t] = step*Bn + b
 * where step = t (dir 0) or len_b-1-t (dir 1), or -1 for t >= len_b. */
int vog_lstm_schedule(const int64_t* lens, int32_t* rows, int Bn, int T, void* stream);'''
assert old in s; s=s.replace(old,new)
old=''' * i,f,g,o). gx: [Bn*T, 8R] fp32 = x W_ih^T + b_ih + b_hh for (dir,gate,unit);
 * whh: [2][4R][R] t16; h_in/h_out: [Bn16, 2R] t16 ping-pong state; c: [Bn16,2R]
 * fp32; out16: [Bn*T, 2R] t16 (zero where t >= len). */'''
new=''' * i,f,g,o). gxs: [2][T][Bn][4R] fp32 = x W_ih^T + b_ih + b_hh in (direction, step)
 * order (vog_lstm_schedule + the GEMM's out_rows scatter); whh: the 16-bit
 * recurrent weights in MFMA-fragment order as packed by vog_lstm_pack_whh;
 * h_in/h_out: [Bn16, 2R] t16 ping-pong state; c: [Bn16,2R] fp32; out16:
 * [Bn*T, 2R] t16 (zero where t >= len). */'''
assert old in s; s=s.replace(old,new)
old='''int vog_bilstm_step(const vog_lstm_step_args* a, void* stream);'''
new='''int vog_bilstm_step(const vog_lstm_step_args* a, void* stream);
/* host: [2][4R][R] fp32 (weight_hh_l*, weight_hh_l*_reverse) -> fragment order, 16 bit.
 * dst holds 2*4R*R halfwords: [dir][unit/4][k/32][lane 64][8]. */
int vog_lstm_pack_whh(const float* whh_fwd, const float* whh_bwd, void* dst_host, int R, vog_dtype dtype);'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

# ---------------- lib.py
p='vognet-pytorch_amd/lib.py'
s=open(p).read()
s=s.replace('("dtype", c_i32), ("c16_dtype", c_i32)]\n\n    def __init__','("dtype", c_i32), ("c16_dtype", c_i32), ("out_rows", c_vp), ("out_rows_ncol", c_i32)]\n\n    def __init__')
s=s.replace('''    "vog_bilstm_step":''','''    "vog_lstm_schedule": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp]),
    "vog_lstm_pack_whh": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32]),
    "vog_bilstm_step":''')
open(p,'w').write(s)

# ---------------- gemm.hip: out_rows in params + epilogues
p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
s=s.replace("  int M, N, K; int relu; int rep; int c16_bf16; int debug;\n","  int M, N, K; int relu; int rep; int c16_bf16; int debug;\n  const int32_t* out_rows; int out_rows_ncol;\n")
# scalar epilogue_store
old='''    if (p.relu) v = fmaxf(v, 0.f);
    for (int j = 0; j < p.rep; ++j) {
      int64_t orow = (int64_t)row * p.rep + j;
      if (p.c32) p.c32[orow * p.ldc + col] = v;
      if (p.c16) p.c16[orow * p.ldc16 + col] = p.c16_bf16 ? to16<BF16>(v) : to16<F16>(v);
    }'''
new='''    if (p.relu) v = fmaxf(v, 0.f);
    if (p.out_rows) {
      const int orow = p.out_rows[(int64_t)(col / p.out_rows_ncol) * p.M + row];
      if (orow < 0) return;
      if (p.c32) p.c32[(int64_t)orow * p.ldc + col] = v;
      if (p.c16) p.c16[(int64_t)orow * p.ldc16 + col] = p.c16_bf16 ? to16<BF16>(v) : to16<F16>(v);
      return;
    }
    for (int j = 0; j < p.rep; ++j) {
      int64_t orow = (int64_t)row * p.rep + j;
      if (p.c32) p.c32[orow * p.ldc + col] = v;
      if (p.c16) p.c16[orow * p.ldc16 + col] = p.c16_bf16 ? to16<BF16>(v) : to16<F16>(v);
    }'''
assert old in s; s=s.replace(old,new)
# pipe epilogue vector path
old='''          for (int j = 0; j < p.rep; ++j) {
            const int64_t orow = (int64_t)m * p.rep + j;
            if (p.c32) *reinterpret_cast<float4*>(p.c32 + orow * p.ldc + n) = v;
            if (p.c16) *reinterpret_cast<u16x4*>(p.c16 + orow * p.ldc16 + n) = h;
          }
        } else {
          epilogue_store<T16>(p, m, n, v.x); epilogue_store<T16>(p, m, n + 1, v.y);'''
new='''          if (p.out_rows) {
            const int orow = p.out_rows[(int64_t)(n / p.out_rows_ncol) * p.M + m];
            if (orow >= 0) {
              if (p.c32) *reinterpret_cast<float4*>(p.c32 + (int64_t)orow * p.ldc + n) = v;
              if (p.c16) *reinterpret_cast<u16x4*>(p.c16 + (int64_t)orow * p.ldc16 + n) = h;
            }
            continue;
          }
          for (int j = 0; j < p.rep; ++j) {
            const int64_t orow = (int64_t)m * p.rep + j;
            if (p.c32) *reinterpret_cast<float4*>(p.c32 + orow * p.ldc + n) = v;
            if (p.c16) *reinterpret_cast<u16x4*>(p.c16 + orow * p.ldc16 + n) = h;
          }
        } else {
          epilogue_store<T16>(p, m, n, v.x); epilogue_store<T16>(p, m, n + 1, v.y);'''
assert old in s; s=s.replace(old,new)
old='''  p.debug = gemm_debug_flags();
  if (p.M <= 64'''
new='''  p.debug = gemm_debug_flags();
  p.out_rows = g->out_rows; p.out_rows_ncol = g->out_rows_ncol;
  if (p.M <= 64'''
assert old in s; s=s.replace(old,new)
old='''  VOG_CHECK_ARG(!(g->residual && g->rep > 1));'''
new='''  VOG_CHECK_ARG(!(g->residual && g->rep > 1));
  VOG_CHECK_ARG(!g->out_rows || (g->rep <= 1 && g->out_rows_ncol > 0 && (g->out_rows_ncol % 4) == 0));'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

# ---------------- lstm.hip rewrite
p='vognet-pytorch_amd/csrc/lstm.hip'
s=open(p).read()
a=s.index("struct LstmParams {")
new_tail=r'''struct LstmParams {
  const float* gxs; const unsigned short* whh; const unsigned short* h_in; unsigned short* h_out;
  float* c; unsigned short* out16; const int64_t* lens;
  int Bn, T, R, step;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) {
  // tanh via exp of -2|x| : accurate to ~1e-7 rel, no overflow
  const float a = fabsf(x);
  const float e = __expf(-2.0f * a);
  const float t = (1.0f - e) / (1.0f + e);
  return x < 0.f ? -t : t;
}

constexpr int LS_CH = 8;      // k-steps per wave kept in registers per chunk

template <typename T16>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmParams p) {
  __shared__ float red[4][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int dir = blockIdx.y;
  const int tile = blockIdx.x;
  const int u0 = tile * 4;
  const int R = p.R;
  const int kg = (lane >> 4) * 8;
  const int ksteps = R / 32;
  // fragment-ordered weights: [dir][tile][kstep][lane][8] -> every load is one contiguous KiB
  const unsigned short* wp = p.whh + (((int64_t)dir * (R / 4) + tile) * ksteps) * 512 + lane * 8;
  const int nbt = (p.Bn + 15) / 16;
  const int unit = u0 + (lane >> 4);

  for (int bt = 0; bt < nbt; ++bt) {
    const int b = bt * 16 + (lane & 15);
    // ---- wave 0: everything the pointwise update needs is requested BEFORE the
    // matrix part, so its latency hides under the W / h loads (none of these
    // addresses depends on another load: gxs is already in step order)
    float g_in[4] = {0.f, 0.f, 0.f, 0.f};
    float c_prev = 0.f;
    unsigned short h_prev = 0;
    int len = 0;
    const bool mine = wid == 0 && unit < R && b < p.Bn;
    const int64_t st = (int64_t)b * 2 * R + (int64_t)dir * R + unit;
    if (mine) {
      len = (int)p.lens[b];
      const float* g = p.gxs + (((int64_t)dir * p.T + p.step) * p.Bn + b) * 4 * R + unit;
#pragma unroll
      for (int r = 0; r < 4; ++r) g_in[r] = g[(int64_t)r * R];
      c_prev = p.c[st];
      h_prev = p.h_in[st];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* hp = p.h_in + (int64_t)b * 2 * R + (int64_t)dir * R;   // rows < Bn16 exist
    for (int base = wid; base < ksteps; base += 4 * LS_CH) {
      u16x8 fw[LS_CH], fh[LS_CH];
#pragma unroll
      for (int c = 0; c < LS_CH; ++c) {
        const int ks = base + c * 4;
        u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        fw[c] = (ks < ksteps) ? *reinterpret_cast<const u16x8*>(wp + (int64_t)ks * 512) : z;
        fh[c] = (ks < ksteps) ? *reinterpret_cast<const u16x8*>(hp + ks * 32 + kg) : z;
      }
#pragma unroll
      for (int c = 0; c < LS_CH; ++c) acc = mfma16<T16>(fw[c], fh[c], acc);
    }
    __syncthreads();                              // red[] free (previous batch tile consumed)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][lane][r] = acc[r];
    __syncthreads();
    if (mine) {
      if (p.step < len) {
        const int pos = dir == 0 ? p.step : len - 1 - p.step;
        float gate[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
          gate[r] = red[0][lane][r] + red[1][lane][r] + red[2][lane][r] + red[3][lane][r] + g_in[r];
        const float cn = sigm(gate[1]) * c_prev + sigm(gate[0]) * tanh_(gate[2]);
        const float hn = sigm(gate[3]) * tanh_(cn);
        p.c[st] = cn;
        const unsigned short h16 = to16<T16>(hn);
        p.h_out[st] = h16;
        p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
      } else {
        p.h_out[st] = h_prev;
      }
    }
  }
}

__global__ void lstm_schedule_kernel(const int64_t* __restrict__ lens, int32_t* __restrict__ rows,
                                     int Bn, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bn * T) return;
  const int b = i / T, t = i % T;
  const int len = (int)lens[b];
  rows[i] = t < len ? t * Bn + b : -1;
  rows[Bn * T + i] = t < len ? (len - 1 - t) * Bn + b : -1;
}

int lstm_step_run(const vog_lstm_step_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->gx && a->whh && a->h_in && a->h_out && a->c && a->out16 && a->lens);
  VOG_CHECK_ARG(a->Bn > 0 && a->T > 0 && a->R > 0 && (a->R % 32) == 0 && a->step >= 0 && a->step < a->T);
  VOG_CHECK_ARG(a->h_in != a->h_out);
  LstmParams p{a->gx, (const unsigned short*)a->whh, (const unsigned short*)a->h_in,
               (unsigned short*)a->h_out, a->c, (unsigned short*)a->out16, a->lens,
               a->Bn, a->T, a->R, a->step};
  dim3 grid(ceil_div(a->R, 4), 2);
  VOG_DISPATCH_DTYPE(a->dtype, hipLaunchKernelGGL((lstm_step_kernel<T16>), grid, dim3(256), 0, st, p));
  VOG_LAUNCH_CHECK();
  return 0;
}

}  // namespace vog

extern "C" int vog_bilstm_step(const vog_lstm_step_args* a, void* stream) {
  return vog::lstm_step_run(a, (hipStream_t)stream);
}

extern "C" int vog_lstm_schedule(const int64_t* lens, int32_t* rows, int Bn, int T, void* stream) {
  VOG_CHECK_ARG(lens && rows && Bn > 0 && T > 0);
  hipLaunchKernelGGL(vog::lstm_schedule_kernel, dim3(vog::ceil_div(Bn * T, 128)), dim3(128), 0,
                     (hipStream_t)stream, lens, rows, Bn, T);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_lstm_pack_whh(const float* whh_fwd, const float* whh_bwd, void* dst_host, int R,
                                 vog_dtype dtype) {
  VOG_CHECK_ARG(whh_fwd && whh_bwd && dst_host && R > 0 && (R % 32) == 0 && (R % 4) == 0);
  unsigned short* dst = (unsigned short*)dst_host;
  const int ksteps = R / 32;
  for (int dir = 0; dir < 2; ++dir) {
    const float* w = dir == 0 ? whh_fwd : whh_bwd;
    for (int tile = 0; tile < R / 4; ++tile)
      for (int ks = 0; ks < ksteps; ++ks)
        for (int lane = 0; lane < 64; ++lane) {
          const int rr = lane & 15;                           // tile row = unit_local*4 + gate
          const int64_t grow = (int64_t)(rr & 3) * R + tile * 4 + (rr >> 2);
          const float* src = w + grow * R + ks * 32 + (lane >> 4) * 8;
          unsigned short* d = dst + ((((int64_t)dir * (R / 4) + tile) * ksteps + ks) * 64 + lane) * 8;
          for (int j = 0; j < 8; ++j) {
            if (dtype == VOG_BF16) {
              unsigned int u; memcpy(&u, &src[j], 4);
              u += 0x7fffu + ((u >> 16) & 1u);
              d[j] = (unsigned short)(u >> 16);
            } else {
              _Float16 h = (_Float16)src[j];
              memcpy(&d[j], &h, 2);
            }
          }
        }
  }
  return 0;
}
'''
s=s[:a]+new_tail
s=s.replace('''// Packed semantics: direction 0 visits position t = step, direction 1 visits
// t = len-1-step; a sentence is active while step < len; inactive sentences keep
// (h, c) and write nothing (out16 is pre-zeroed => zeros past each length).''','''// Packed semantics: direction 0 visits position t = step, direction 1 visits
// t = len-1-step; a sentence is active while step < len; inactive sentences keep
// (h, c) and write nothing (out16 is pre-zeroed => zeros past each length).
//
// The step is a ~2 us latency chain, so nothing on it may wait for a second
// dependent load: the input projections arrive already in (direction, step)
// order (the GEMM scatters its rows by the schedule of vog_lstm_schedule), and
// W_hh is stored in MFMA-fragment order so each wave load is one contiguous KiB.''')
open(p,'w').write(s)

# ---------------- forward.hip
p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
old='''      for (size_t i = 0; i < h.size(); ++i) whh[(size_t)dir * 4 * R * R + i] = h_to16(h[i], et);'''
new='''      (void)h;'''
assert old in s; s=s.replace(old,new)
old='''    unsigned short *pw, *ph; float* pb;'''
new='''    {
      std::string s0 = "_l" + std::to_string(l), s1 = s0 + "_reverse";
      VOG_TRY(vog_lstm_pack_whh(W(c, "lstm_encoder.lstm.weight_hh" + s0).data(),
                                W(c, "lstm_encoder.lstm.weight_hh" + s1).data(), whh.data(), R, (vog_dtype)et));
    }
    unsigned short *pw, *ph; float* pb;'''
assert old in s; s=s.replace(old,new)
old='''  p.add("tok", (int64_t)g.Bn * g.T * 4);'''
new='''  p.add("tok", (int64_t)g.Bn * g.T * 4);
  p.add("lstm_rows", (int64_t)2 * g.Bn * g.T * 4);'''
assert old in s; s=s.replace(old,new)
old='''    float* gx = ws.at<float>("gx");'''
new='''    float* gx = ws.at<float>("gx");
    int32_t* lrows = ws.at<int32_t>("lstm_rows");
    {
      const int64_t* lens = b->srl_arg_word_mask_len;
      steps.push_back({"lstm_schedule", [=](hipStream_t st) { return vog_lstm_schedule(lens, lrows, Bn, T, st); }});
    }'''
assert old in s; s=s.replace(old,new)
old='''      ga.w = c->wih[l]; ga.ldw = ga.K; ga.bias = c->bsum[l]; ga.c32 = gx; ga.ldc = 8 * R;
      ga.M = Bn * T; ga.N = 8 * R; ga.rep = 1; ga.dtype = et;'''
new='''      // output rows land in (direction, step) order: gxs[dir][step][b][4R]
      ga.w = c->wih[l]; ga.ldw = ga.K; ga.bias = c->bsum[l]; ga.c32 = gx; ga.ldc = 4 * R;
      ga.M = Bn * T; ga.N = 8 * R; ga.rep = 1; ga.dtype = et;
      ga.out_rows = lrows; ga.out_rows_ncol = 4 * R;'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)
