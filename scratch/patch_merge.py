p='include/vog_hip.h'
s=open(p).read()
old='''/* One time step of one BiLSTM layer, both directions, packed-sequence'''
new='''/* Fused prologue of the language path (one launch instead of memset + vog_srl_gather +
 * vog_lstm_schedule): zero `zero_bytes` at `zero` (multiple of 16), token re-index and the
 * packed-sequence schedule. */
int vog_lang_prep(void* zero, int64_t zero_bytes, const int64_t* words_ind, const int64_t* word_mask,
                  const int64_t* lens, int32_t* tok, int32_t* rows, int Bn, int T, int nsrl,
                  int seq_len, int vocab_size, void* stream);

/* Fused prologue of the visual path (one launch instead of vog_cast_f32_to_t16 + up to two
 * vog_box_u): u0/u1 = bias precursors for obj_tx / mul_tx (either w_pe may be NULL). */
typedef struct vog_visprep_args {
  const float* src0; void* dst0; int64_t n0; const float* src1; void* dst1; int64_t n1; vog_dtype dtype;
  const float* props; int n_rows; float vid_w, vid_h;
  const float* w_pe0; float* u0; int H0; float nfrm_div0;
  const float* w_pe1; float* u1; int H1; float nfrm_div1;
} vog_visprep_args;
int vog_vis_prep(const vog_visprep_args* a, void* stream);

/* One time step of one BiLSTM layer, both directions, packed-sequence'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/lib.py'
s=open(p).read()
s=s.replace('''class QkvArgs(C.Structure):''','''class VisprepArgs(C.Structure):
    _fields_ = [("src0", c_vp), ("dst0", c_vp), ("n0", c_i64), ("src1", c_vp), ("dst1", c_vp), ("n1", c_i64),
                ("dtype", c_i32), ("props", c_vp), ("n_rows", c_i32), ("vid_w", c_f32), ("vid_h", c_f32),
                ("w_pe0", c_vp), ("u0", c_vp), ("H0", c_i32), ("nfrm_div0", c_f32),
                ("w_pe1", c_vp), ("u1", c_vp), ("H1", c_i32), ("nfrm_div1", c_f32)]


class QkvArgs(C.Structure):''',1)
s=s.replace('''    "vog_bilstm_step":''','''    "vog_lang_prep": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vog_vis_prep": (c_i32, [C.POINTER(VisprepArgs), c_vp]),
    "vog_bilstm_step":''')
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/elementwise.hip'
s=open(p).read()
# ---- merged pred kernel
a=s.index("__global__ void pred_kernel(vog_pred_args a, int64_t rec_bytes) {")
b=s.index("}  // namespace vog\n\nusing namespace vog;")
new=r'''__global__ void pred_kernel(vog_pred_args a, int64_t rec_bytes) {
  // one thread per (query, arg, frame): the ncmp videos of that frame, then pred_cmp
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int per_q = a.nsrl * a.nfrm0;
  if (i >= a.B * per_q) return;
  const int b = i / per_q, arg = (i % per_q) / a.nfrm0, f = i % a.nfrm0;
  const int npv = a.nfrm0 * a.nppf0;
  unsigned char* rec = reinterpret_cast<unsigned char*>(a.rec) + (int64_t)b * rec_bytes;
  float* boxes = reinterpret_cast<float*>(rec);
  float* scores = boxes + (int64_t)a.nsrl * a.ncmp * a.nfrm0 * 7;
  int64_t* idx = reinterpret_cast<int64_t*>(rec + (int64_t)a.nsrl * a.ncmp * a.nfrm0 * 8 * 4);
  float best_c = -3.0e38f;
  int64_t arg_c = 0;
  for (int c = 0; c < a.ncmp; ++c) {
    int64_t e0, p0;   // first proposal of this (video, frame): in outs_eval / in props
    if (a.conc_type == VOG_CONC_SPAT) {
      const int r0 = (f * a.ncmp + c) * a.nppf0;
      e0 = ((int64_t)b * a.nsrl + arg) * ((int64_t)a.ncmp * npv) + r0;
      p0 = (int64_t)b * a.ncmp * npv + r0;
    } else if (a.conc_type == VOG_CONC_TEMP) {
      const int r0 = (c * a.nfrm0 + f) * a.nppf0;
      e0 = ((int64_t)b * a.nsrl + arg) * ((int64_t)a.ncmp * npv) + r0;
      p0 = (int64_t)b * a.ncmp * npv + r0;
    } else {
      e0 = (((int64_t)b * a.ncmp + c) * a.nsrl + arg) * npv + (int64_t)f * a.nppf0;
      p0 = ((int64_t)b * a.ncmp + c) * npv + (int64_t)f * a.nppf0;
    }
    float best = a.outs_eval[e0];
    int bi = 0;
    for (int k = 1; k < a.nppf0; ++k) {
      const float v = a.outs_eval[e0 + k];
      if (v > best) { best = v; bi = k; }          // first maximum wins (torch.max on CPU)
    }
    const int64_t o = ((int64_t)arg * a.ncmp + c) * a.nfrm0 + f;
    const float* pr = a.props + (p0 + bi) * 7;
#pragma unroll
    for (int k = 0; k < 7; ++k) boxes[o * 7 + k] = pr[k];
    scores[o] = best;
    if (best > best_c) { best_c = best; arg_c = c; }   // first maximum over the videos
  }
  int64_t out = 0;
  if (a.conc_type == VOG_CONC_SPAT) out = arg_c;
  else if (a.conc_type == VOG_CONC_SEP) {
    float bf = a.fin_scores[(int64_t)b * a.ncmp];
    for (int c = 1; c < a.ncmp; ++c) {
      const float v = a.fin_scores[(int64_t)b * a.ncmp + c];
      if (v > bf) { bf = v; out = c; }
    }
  }
  idx[(int64_t)arg * a.nfrm0 + f] = out;
}

// ---------------------------------------------------------------------------
// fused prologues (one graph node each instead of 3)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lang_prep_kernel(uint4* __restrict__ zero, int64_t zero16,
                                                        const int64_t* __restrict__ words,
                                                        const int64_t* __restrict__ mask,
                                                        const int64_t* __restrict__ lens,
                                                        int32_t* __restrict__ tok, int32_t* __restrict__ rows,
                                                        int Bn, int T, int nsrl, int seq_len, int vocab) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = gid; i < zero16; i += stride) zero[i] = make_uint4(0, 0, 0, 0);
  if (gid < (int64_t)Bn * T) {
    const int i = (int)gid, b = i / T, t = i % T;
    const int64_t m = mask[(int64_t)b * seq_len + t];
    int64_t v = vocab;
    if (m >= 0 && m < (int64_t)nsrl * seq_len) v = words[(int64_t)b * nsrl * seq_len + m];
    tok[i] = (int32_t)v;
    const int len = (int)lens[b];
    rows[i] = t < len ? t * Bn + b : -1;
    rows[Bn * T + i] = t < len ? T * Bn - 1 + (len - 1 - t) * Bn + b : -1;
  }
}

template <typename T16>
__global__ __launch_bounds__(256) void vis_prep_kernel(vog_visprep_args a, int cast_blocks) {
  if ((int)blockIdx.x < cast_blocks) {
    const int64_t q0 = a.n0 / 4, q1 = a.n1 / 4;
    const int64_t stride = (int64_t)cast_blocks * blockDim.x;
    const float4* s0 = reinterpret_cast<const float4*>(a.src0);
    const float4* s1 = reinterpret_cast<const float4*>(a.src1);
    u16x4* d0 = reinterpret_cast<u16x4*>(a.dst0);
    u16x4* d1 = reinterpret_cast<u16x4*>(a.dst1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q0 + q1; i += stride) {
      const bool first = i < q0;
      const float4 v = first ? s0[i] : s1[i - q0];
      u16x4 o = {to16<T16>(v.x), to16<T16>(v.y), to16<T16>(v.z), to16<T16>(v.w)};
      if (first) d0[i] = o; else d1[i - q0] = o;
    }
    return;
  }
  const int i = ((int)blockIdx.x - cast_blocks) * blockDim.x + threadIdx.x;
  const int n0 = a.w_pe0 ? a.n_rows * a.H0 : 0, n1 = a.w_pe1 ? a.n_rows * a.H1 : 0;
  if (i >= n0 + n1) return;
  const bool second = i >= n0;
  const int k = second ? i - n0 : i;
  const int H = second ? a.H1 : a.H0;
  const float fd = second ? a.nfrm_div1 : a.nfrm_div0;
  const float* w = (second ? a.w_pe1 : a.w_pe0) + (k % H) * 5;
  const float* b = a.props + (int64_t)(k / H) * 7;
  const float v = w[0] * (b[0] / a.vid_w) + w[1] * (b[1] / a.vid_h) + w[2] * (b[2] / a.vid_w) +
                  w[3] * (b[3] / a.vid_h) + w[4] * (b[4] / fd);
  (second ? a.u1 : a.u0)[k] = v;
}

'''
s=s[:a]+new+s[b:]
old=s[s.index('  const int n1 = a->B * a->nsrl * a->ncmp * a->nfrm0;'):s.index('  VOG_LAUNCH_CHECK();\n  return 0;\n}', s.index('  const int n1 = a->B * a->nsrl * a->ncmp * a->nfrm0;'))]
new='''  const int n1 = a->B * a->nsrl * a->nfrm0;
  hipLaunchKernelGGL(pred_kernel, dim3(ceil_div(n1, 64)), dim3(64), 0, (hipStream_t)stream, *a, rb);
'''
s=s.replace(old,new)
s=s.replace('''extern "C" int vog_splitk_finish(''','''extern "C" int vog_lang_prep(void* zero, int64_t zero_bytes, const int64_t* words_ind, const int64_t* word_mask,
                             const int64_t* lens, int32_t* tok, int32_t* rows, int Bn, int T, int nsrl,
                             int seq_len, int vocab_size, void* stream) {
  VOG_CHECK_ARG(words_ind && word_mask && lens && tok && rows && Bn > 0 && T > 0 && T <= seq_len);
  VOG_CHECK_ARG(zero_bytes >= 0 && (zero_bytes % 16) == 0 && (zero_bytes == 0 || zero));
  const int64_t z16 = zero_bytes / 16;
  int64_t blocks = (z16 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  const int64_t need = ((int64_t)Bn * T + 255) / 256;
  if (blocks < need) blocks = need;
  hipLaunchKernelGGL(lang_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (uint4*)zero, z16, words_ind, word_mask, lens, tok, rows, Bn, T, nsrl, seq_len, vocab_size);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_vis_prep(const vog_visprep_args* a, void* stream) {
  VOG_CHECK_ARG(a && (a->n0 % 4) == 0 && (a->n1 % 4) == 0 && (a->n0 == 0 || (a->src0 && a->dst0)) &&
                (a->n1 == 0 || (a->src1 && a->dst1)));
  VOG_CHECK_ARG((!a->w_pe0 && !a->w_pe1) || (a->props && a->n_rows > 0));
  VOG_CHECK_ARG((!a->w_pe0 || (a->u0 && a->H0 > 0)) && (!a->w_pe1 || (a->u1 && a->H1 > 0)));
  const int64_t q = (a->n0 + a->n1) / 4;
  int cast_blocks = (int)((q + 255) / 256 < 2048 ? (q + 255) / 256 : 2048);
  const int nu = (a->w_pe0 ? a->n_rows * a->H0 : 0) + (a->w_pe1 ? a->n_rows * a->H1 : 0);
  const int u_blocks = ceil_div(nu, 256);
  if (cast_blocks + u_blocks == 0) return 0;
  VOG_DISPATCH_DTYPE(a->dtype, hipLaunchKernelGGL((vis_prep_kernel<T16>), dim3(cast_blocks + u_blocks), dim3(256), 0,
                     (hipStream_t)stream, *a, cast_blocks));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_splitk_finish(''',1)
open(p,'w').write(s)

# ---- forward.hip: use the fused prologues
p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
old=s[s.index('    steps.push_back({"zero", [=](hipStream_t st) { VOG_HIP(hipMemsetAsync(z, 0, zb, st)); return 0; }});'):s.index('    for (int l = 0; l < d.rnn_layers; ++l) {')]
new='''    int32_t* tok = ws.at<int32_t>("tok");
    const int64_t *wi = b->srl_arg_words_ind, *wm = b->srl_arg_word_mask;
    const int nsrl = d.nsrl, sl = d.seq_len, V = d.vocab_size;
    float* gx = ws.at<float>("gx");
    int32_t* lrows = ws.at<int32_t>("lstm_rows");
    {
      const int64_t* lens = b->srl_arg_word_mask_len;
      steps.push_back({"lang_prep", [=](hipStream_t st) {
        return vog_lang_prep(z, zb, wi, wm, lens, tok, lrows, Bn, T, nsrl, sl, V, st); }});
    }
'''
s=s.replace(old,new)
# box_u steps: removed from tx_steps (done in vis_prep)
old='''  if (tw.use_rel) {
    const float* props = b->pad_proposals;
    const int n_rows = (int)g.rows_obj, H = tw.H;
    const float* pw = tw.pe_w;
    const float vw = d.vid_w, vh = d.vid_h;
    steps.push_back({n + "_box_u", [=](hipStream_t st) {
      return vog_box_u(props, pw, u, n_rows, H, vw, vh, fdiv, st); }});
  }'''
new='''  (void)fdiv;   // the bias precursors u are produced by the fused visual prologue (vis_prep)'''
assert old in s; s=s.replace(old,new)
old=s[s.index('    // raw features -> encoder operand type once (then both encoders run on the\n'):s.index('    // the two encoders have 52 / 12 output tiles')]
new='''    // fused visual prologue: raw features -> encoder operand type (the LDS-DMA GEMM cannot
    // convert in flight) + the box-bias precursors of both transformers
    {
      vog_visprep_args vp{};
      vp.src0 = b->pad_region_feature; vp.dst0 = ws.at<void>("prop16"); vp.n0 = g.rows_obj * d.prop_dim;
      vp.src1 = b->seg_feature_for_frms; vp.dst1 = ws.at<void>("seg16"); vp.n1 = (int64_t)g.n_vid * g.Fv * d.seg_dim;
      vp.dtype = et; vp.props = b->pad_proposals; vp.n_rows = (int)g.rows_obj; vp.vid_w = d.vid_w; vp.vid_h = d.vid_h;
      if (has_obj(d) && c->obj.use_rel) {
        vp.w_pe0 = c->obj.pe_w; vp.u0 = ws.at<float>("obj_u"); vp.H0 = c->obj.H; vp.nfrm_div0 = g.fdiv_obj;
      }
      if (has_mul(d) && c->mul.use_rel) {
        vp.w_pe1 = c->mul.pe_w; vp.u1 = ws.at<float>("mul_u"); vp.H1 = c->mul.H; vp.nfrm_div1 = (float)g.nfrm;
      }
      steps.push_back({"vis_prep", [=](hipStream_t st) { return vog_vis_prep(&vp, st); }});
    }
'''
s=s.replace(old,new)
open(p,'w').write(s)
