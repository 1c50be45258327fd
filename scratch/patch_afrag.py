p='include/vog_hip.h'
s=open(p).read()
old='''  int w_frag;
} vog_gemm_args;'''
new='''  int w_frag;
  /* a_frag = 1 (M <= 64 kernel only, 16-bit A): `a` is in fragment order
   * [m/16][K/32][lane = ((k%32)/8)*16 + m%16][k%8] — written that way by vog_bilstm_step
   * (out_frag) so that the LSTM -> projection hand-off needs no strided fragment loads. */
  int a_frag;
} vog_gemm_args;'''
assert old in s; s=s.replace(old,new)
old='''  void* out16; const int64_t* lens; int Bn, T, R, step; vog_dtype dtype;
} vog_lstm_step_args;'''
new='''  void* out16; const int64_t* lens; int Bn, T, R, step; vog_dtype dtype;
  /* out_frag = 1: out16 is written in the A-fragment order of vog_gemm_args.a_frag
   * (K = 2R, rows m = b*T + pos) and every active step also writes h into row
   * final_row0 + b (the final hidden state ends up there). 0: row-major [.., 2R]. */
  int out_frag; int final_row0;
} vog_lstm_step_args;'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

p='vognet-pytorch_amd/lib.py'
s=open(p).read()
s=s.replace('("splitk", c_i32), ("w_frag", c_i32)]','("splitk", c_i32), ("w_frag", c_i32), ("a_frag", c_i32)]')
s=s.replace('''                ("out16", c_vp), ("lens", c_vp), ("Bn", c_i32), ("T", c_i32), ("R", c_i32),
                ("step", c_i32), ("dtype", c_i32)]''','''                ("out16", c_vp), ("lens", c_vp), ("Bn", c_i32), ("T", c_i32), ("R", c_i32),
                ("step", c_i32), ("dtype", c_i32), ("out_frag", c_i32), ("final_row0", c_i32)]''')
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/common.h'
s=open(p).read()
old="static inline int ceil_div(int a, int b)"
new='''// A operand of the M <= 64 GEMM in fragment order (K = number of columns)
__host__ __device__ __forceinline__ int64_t frag_a(int m, int k, int K) {
  return ((((int64_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64) + (((k >> 3) & 3) << 4) + (m & 15)) * 8 + (k & 7);
}

static inline int ceil_div(int a, int b)'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/lstm.hip'
s=open(p).read()
old='''  float* c; unsigned short* out16; const int64_t* lens;
  int Bn, T, R, step;
};

__device__ __forceinline__ float sigm'''
new='''  float* c; unsigned short* out16; const int64_t* lens;
  int Bn, T, R, step; int out_frag, final_row0;
};

__device__ __forceinline__ float sigm'''
assert old in s; s=s.replace(old,new,1)
old='''        p.h_out[st] = h16;
        p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
      } else {
        p.h_out[st] = h_prev;
      }'''
new='''        p.h_out[st] = h16;
        if (p.out_frag) {
          p.out16[frag_a(b * p.T + pos, dir * R + unit, 2 * R)] = h16;
          p.out16[frag_a(p.final_row0 + b, dir * R + unit, 2 * R)] = h16;   // last active step wins
        } else {
          p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
        }
      } else {
        p.h_out[st] = h_prev;
      }'''
assert old in s; s=s.replace(old,new,1)
old='''               a->Bn, a->T, a->R, a->step};
  dim3 grid(ceil_div(a->R, 4), 2);'''
new='''               a->Bn, a->T, a->R, a->step, a->out_frag, a->final_row0};
  dim3 grid(ceil_div(a->R, 4), 2);'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
s=s.replace("  int splitk; int w_frag;\n","  int splitk; int w_frag; int a_frag;\n",1)
old='''        u16x8 fa[SK_CH];
#pragma unroll
        for (int c = 0; c < SK_CH; ++c) {
          const int ks = base + c * 4;
          fa[c] = load_a_chunk<T16, A_F32>(p.a, a_off[mt], ks * 32 + kg, a_ok[mt] && ks < ksteps);
        }'''
new='''        u16x8 fa[SK_CH];
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
        }'''
assert old in s; s=s.replace(old,new,1)
old='''  p.w_frag = g->w_frag;'''
new='''  p.w_frag = g->w_frag; p.a_frag = g->a_frag;
  if (p.a_frag && !(p.M <= 64 && (p.K % 32) == 0 && !g->a_is_f32 && !g->a_rows))
    VOG_FAIL(-1, "a_frag activations are only valid for the M <= 64 kernel with a 16-bit A (M=%d K=%d)", p.M, p.K);'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
old='''    p.add("lstm_hB_" + std::to_string(l), (int64_t)g.Bn16 * 2 * g.R * 2);'''
new='''    p.add("lstm_hB_" + std::to_string(l), (int64_t)g.Bn16 * 2 * g.R * 2);
    p.add("lstm_hA2_" + std::to_string(l), (int64_t)g.Bn16 * 2 * g.R * 2);   // ping buffer when out16 is fragment-ordered'''
assert old in s; s=s.replace(old,new,1)
# out16 buffer needs room for whole 16-row tiles in frag mode: (ceil((Bn*T+Bn)/16)*16) rows
old='''    p.add("lstm_out16_" + std::to_string(l), (int64_t)(g.Bn * g.T + g.Bn16) * 2 * g.R * 2);'''
new='''    p.add("lstm_out16_" + std::to_string(l), (int64_t)(round_up64(g.Bn * g.T + g.Bn, 16) + g.Bn16) * 2 * g.R * 2);'''
assert old in s; s=s.replace(old,new,1)
old='''      if (l == 0) { ga.a = c->emb; ga.a_is_f32 = 1; ga.lda = g.E; ga.a_rows = tok; ga.K = g.E; }
      else { ga.a = ws.at<void>("lstm_out16_" + std::to_string(l - 1)); ga.lda = 2 * R; ga.K = 2 * R; }'''
new='''      // LSTM outputs feed M <= 64 GEMMs (next layer's input projection, final projection): then the
      // step kernel writes them in A-fragment order and those GEMMs load contiguous fragments
      const bool ofrag = (Bn * T + Bn) <= 64 && !(c->lstm_persistent && vog_bilstm_layer_supported(Bn, R));
      if (l == 0) { ga.a = c->emb; ga.a_is_f32 = 1; ga.lda = g.E; ga.a_rows = tok; ga.K = g.E; }
      else {
        ga.a = ws.at<void>("lstm_out16_" + std::to_string(l - 1)); ga.lda = 2 * R; ga.K = 2 * R;
        ga.a_frag = ofrag ? 1 : 0;
      }'''
assert old in s; s=s.replace(old,new,1)
old='''      void* hA = ws.at<unsigned short>("lstm_out16_" + std::to_string(l)) + (int64_t)Bn * T * 2 * R;'''
new='''      void* hA = ofrag ? ws.at<void>("lstm_hA2_" + std::to_string(l))
                       : (void*)(ws.at<unsigned short>("lstm_out16_" + std::to_string(l)) + (int64_t)Bn * T * 2 * R);'''
assert old in s; s=s.replace(old,new,1)
old='''        la.lens = b->srl_arg_word_mask_len; la.Bn = Bn; la.T = T; la.R = R; la.step = s; la.dtype = et;'''
new='''        la.lens = b->srl_arg_word_mask_len; la.Bn = Bn; la.T = T; la.R = R; la.step = s; la.dtype = et;
        la.out_frag = ofrag ? 1 : 0; la.final_row0 = Bn * T;'''
assert old in s; s=s.replace(old,new,1)
old='''    if (po.M <= 64 && c->w_outproj_f) { po.w = c->w_outproj_f; po.w_frag = 1; }'''
new='''    if (po.M <= 64 && c->w_outproj_f) { po.w = c->w_outproj_f; po.w_frag = 1; }
    po.a_frag = ((Bn * T + Bn) <= 64 && !(c->lstm_persistent && vog_bilstm_layer_supported(Bn, R))) ? 1 : 0;'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)
