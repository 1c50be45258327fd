p='include/vog_hip.h'
s=open(p).read()
old='''  const struct vog_vislang_args* res_vislang;
} vog_gemm_args;
int vog_gemm_bias_act(const vog_gemm_args* g, void* stream);'''
new='''  const struct vog_vislang_args* res_vislang;
  /* split-K (> 1): the K range is cut into `splitk` slices, slice s writes its raw
   * partial product to c32 + s*M*ldc (fp32 slabs, plain stores); bias / relu /
   * residual / c16 / rep must be unset — apply them with vog_splitk_finish. For
   * GEMMs with fewer output tiles than CUs and a long K (the two feature encoders). */
  int splitk;
} vog_gemm_args;
int vog_gemm_bias_act(const vog_gemm_args* g, void* stream);

/* out[m*rep + j, n] = act(sum_s slab[s][m][n] + bias[n]) for up to two problems in
 * one launch (prop_encoder + seg_encoder), fp32 and/or 16-bit outputs. */
typedef struct vog_splitk_prob {
  const float* slabs; int splits; int M, N; const float* bias; int relu; int rep;
  float* c32; void* c16; int64_t ldc, ldc16; int c16_dtype;
} vog_splitk_prob;
int vog_splitk_finish(const vog_splitk_prob* p0, const vog_splitk_prob* p1, void* stream);'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

p='vognet-pytorch_amd/lib.py'
s=open(p).read()
s=s.replace('("res_vislang", c_vp)]','("res_vislang", c_vp), ("splitk", c_i32)]')
s=s.replace('''class QkvArgs(C.Structure):''','''class SplitkProb(C.Structure):
    _fields_ = [("slabs", c_vp), ("splits", c_i32), ("M", c_i32), ("N", c_i32), ("bias", c_vp),
                ("relu", c_i32), ("rep", c_i32), ("c32", c_vp), ("c16", c_vp), ("ldc", c_i64),
                ("ldc16", c_i64), ("c16_dtype", c_i32)]


class QkvArgs(C.Structure):''',1)
s=s.replace('''    "vog_qkv_proj":''','''    "vog_splitk_finish": (c_i32, [C.POINTER(SplitkProb), C.POINTER(SplitkProb), c_vp]),
    "vog_qkv_proj":''')
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
s=s.replace("  const int32_t* out_rows; int out_rows_ncol;\n  // implicit vis||lang residual","  const int32_t* out_rows; int out_rows_ncol;\n  int splitk;\n  // implicit vis||lang residual",1)
# pipe kernel: k range per split
old='''  const int nk = p.K / 64;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);
  for (int kt = 0; kt < nk; ++kt) {'''
new='''  // split-K: blockIdx.y owns k tiles [kbeg, kbeg + nk) and its own fp32 output slab
  int nk = p.K / 64;
  int kbeg = 0;
  if (p.splitk > 1) {
    const int per = (nk + p.splitk - 1) / p.splitk;
    kbeg = blockIdx.y * per;
    nk = nk - kbeg < per ? nk - kbeg : per;
    if (nk < 0) nk = 0;
    p.c32 += (int64_t)blockIdx.y * p.M * p.ldc;
#pragma unroll
    for (int i = 0; i < LPT; ++i) gsrc[i] += (int64_t)kbeg * 64;
  }
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);
  for (int kt = 0; kt < nk; ++kt) {'''
assert old in s; s=s.replace(old,new)
old='''  dim3 grid(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);'''
new='''  dim3 grid(ceil_div(p.M, BM) * ceil_div(p.N, BN), p.splitk > 1 ? p.splitk : 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);'''
assert old in s; s=s.replace(old,new)
old='''  if (ntiles(64, 64) < 256) return launch_pipe_cfg<T16, 64, 64, 4, EPI>(p, st);   // <1 tile per CU: go deep'''
new='''  if (ntiles(64, 64) * (p.splitk > 1 ? p.splitk : 1) < 256)
    return launch_pipe_cfg<T16, 64, 64, 4, EPI>(p, st);                            // <1 tile per CU: go deep'''
assert old in s; s=s.replace(old,new)
old='''  if (g->res_vislang) {'''
new='''  p.splitk = g->splitk;
  if (g->res_vislang) {'''
assert old in s; s=s.replace(old,new,1)
old='''  if (p.M <= 64 && (p.K % 32) == 0) {'''
new='''  if (p.splitk > 1) {
    if (!pipe_ok(p, g->a_is_f32 != 0) || g->bias || g->residual || g->relu || g->c16 || p.rep != 1 ||
        g->out_rows || g->res_vislang || !g->c32 || g->splitk > p.K / 64)
      VOG_FAIL(-1, "split-K GEMM needs the LDS-DMA path (16-bit A, K %% 64 == 0, M > 64) and a bare fp32 output");
    return launch_pipe<T16, EPI_PLAIN>(p, st);
  }
  if (p.M <= 64 && (p.K % 32) == 0) {'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/elementwise.hip'
s=open(p).read()
old="// ---------------------------------------------------------------------------\n// u[row,h] = W_pe[h,:] . norm(box[row,:5])"
new='''// ---------------------------------------------------------------------------
// split-K finish: sum the slabs, bias, ReLU, row replication, fp32 + 16-bit copies
// ---------------------------------------------------------------------------
struct SplitkProbs { vog_splitk_prob p[2]; int blocks0; };

__global__ __launch_bounds__(256) void splitk_finish_kernel(SplitkProbs a) {
  const bool second = (int)blockIdx.x >= a.blocks0;
  const vog_splitk_prob& q = a.p[second ? 1 : 0];
  const int64_t i4 = (int64_t)(blockIdx.x - (second ? a.blocks0 : 0)) * 256 + threadIdx.x;   // float4 index
  const int n4 = q.N / 4;
  if (i4 >= (int64_t)q.M * n4) return;
  const int m = (int)(i4 / n4), n = (int)(i4 - (int64_t)m * n4) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < q.splits; ++s) {
    const float4 x = *reinterpret_cast<const float4*>(q.slabs + ((int64_t)s * q.M + m) * q.N + n);
    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
  }
  if (q.bias) {
    const float4 b = *reinterpret_cast<const float4*>(q.bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (q.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  u16x4 h;
  if (q.c16) {
    if (q.c16_dtype == VOG_BF16) h = u16x4{to16<BF16>(v.x), to16<BF16>(v.y), to16<BF16>(v.z), to16<BF16>(v.w)};
    else h = u16x4{to16<F16>(v.x), to16<F16>(v.y), to16<F16>(v.z), to16<F16>(v.w)};
  }
  for (int j = 0; j < q.rep; ++j) {
    const int64_t orow = (int64_t)m * q.rep + j;
    if (q.c32) *reinterpret_cast<float4*>(q.c32 + orow * q.ldc + n) = v;
    if (q.c16) *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(q.c16) + orow * q.ldc16 + n) = h;
  }
}

'''+old
assert old in s; s=s.replace(old,new,1)
old='''extern "C" int vog_box_u('''
new='''extern "C" int vog_splitk_finish(const vog_splitk_prob* p0, const vog_splitk_prob* p1, void* stream) {
  VOG_CHECK_ARG(p0 && p0->slabs && p0->splits > 0 && (p0->N % 4) == 0 && (p0->c32 || p0->c16) && p0->rep >= 1);
  VOG_CHECK_ARG(!p1 || (p1->slabs && p1->splits > 0 && (p1->N % 4) == 0 && (p1->c32 || p1->c16) && p1->rep >= 1));
  SplitkProbs a{};
  a.p[0] = *p0;
  a.blocks0 = (int)(((int64_t)p0->M * (p0->N / 4) + 255) / 256);
  int blocks1 = 0;
  if (p1) { a.p[1] = *p1; blocks1 = (int)(((int64_t)p1->M * (p1->N / 4) + 255) / 256); }
  hipLaunchKernelGGL(splitk_finish_kernel, dim3(a.blocks0 + blocks1), dim3(256), 0, (hipStream_t)stream, a);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_box_u('''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
old='''  p.add("prop_seg", g.rows_obj * g.d_obj * 4);'''
new='''  p.add("enc_slabs", (int64_t)16 * (g.rows_obj * d.prop_enc + (int64_t)g.n_vid * g.Fv * d.seg_enc) * 4);
  p.add("prop_seg", g.rows_obj * g.d_obj * 4);'''
assert old in s; s=s.replace(old,new,1)
a=s.index("    vog_gemm_args pe{}; pe.c16_dtype = d.tx_dtype;")
b=s.index("  // ---- object transformer (a7, a8)")
new='''    // the two encoders have 52 / 12 output tiles and K = 2048 / 3072: split K so every CU
    // gets a slice, partial products go to fp32 slabs, one finishing pass for both
    auto pick_split = [](int M, int N, int K) {
      const int64_t tiles = (int64_t)ceil_div(M, 64) * ceil_div(N, 64);
      int sp = (int)((511 + tiles) / tiles);
      const int nk = K / 64;
      if (sp > nk / 4) sp = nk / 4;
      if (sp > 16) sp = 16;
      return sp < 1 ? 1 : sp;
    };
    const int Mp = (int)g.rows_obj, Ms = g.n_vid * g.Fv;
    const bool can_split = (d.prop_dim % 64) == 0 && (d.seg_dim % 64) == 0 && Mp > 64 && Ms > 64 &&
                           (d.prop_enc % 4) == 0 && (d.seg_enc % 4) == 0;
    const int sp_p = can_split ? pick_split(Mp, d.prop_enc, d.prop_dim) : 1;
    const int sp_s = can_split ? pick_split(Ms, d.seg_enc, d.seg_dim) : 1;
    float* slab_p = ws.at<float>("enc_slabs");
    float* slab_s = slab_p + (int64_t)16 * Mp * d.prop_enc;
    vog_gemm_args pe{}; pe.c16_dtype = d.tx_dtype;
    pe.a = ws.at<void>("prop16"); pe.a_is_f32 = 0; pe.lda = d.prop_dim; pe.w = c->w_prop; pe.ldw = d.prop_dim;
    pe.M = Mp; pe.N = d.prop_enc; pe.K = d.prop_dim; pe.rep = 1; pe.dtype = et;
    vog_gemm_args se{}; se.c16_dtype = d.tx_dtype;
    se.a = ws.at<void>("seg16"); se.a_is_f32 = 0; se.lda = d.seg_dim; se.w = c->w_seg; se.ldw = d.seg_dim;
    se.M = Ms; se.N = d.seg_enc; se.K = d.seg_dim; se.dtype = et;
    if (can_split && sp_p > 1 && sp_s > 1) {
      pe.c32 = slab_p; pe.ldc = d.prop_enc; pe.splitk = sp_p;
      se.c32 = slab_s; se.ldc = d.seg_enc; se.splitk = sp_s; se.rep = 1;
      steps.push_back({"prop_enc", [=](hipStream_t st) { return vog_gemm_bias_act(&pe, st); }});
      steps.push_back({"seg_enc", [=](hipStream_t st) { return vog_gemm_bias_act(&se, st); }});
      vog_splitk_prob f0{}, f1{};
      f0.slabs = slab_p; f0.splits = sp_p; f0.M = Mp; f0.N = d.prop_enc; f0.bias = c->b_prop; f0.relu = 1; f0.rep = 1;
      f0.c32 = ps32; f0.c16 = ps16; f0.ldc = g.d_obj; f0.ldc16 = g.d_obj; f0.c16_dtype = d.tx_dtype;
      f1.slabs = slab_s; f1.splits = sp_s; f1.M = Ms; f1.N = d.seg_enc; f1.bias = c->b_seg; f1.relu = 1; f1.rep = d.nppf0;
      f1.c32 = ps32 + d.prop_enc; f1.c16 = (unsigned short*)ps16 + d.prop_enc; f1.ldc = g.d_obj; f1.ldc16 = g.d_obj;
      f1.c16_dtype = d.tx_dtype;
      steps.push_back({"enc_finish", [=](hipStream_t st) { return vog_splitk_finish(&f0, &f1, st); }});
    } else {
      pe.bias = c->b_prop; pe.relu = 1; pe.c32 = ps32; pe.c16 = ps16; pe.ldc = g.d_obj; pe.ldc16 = g.d_obj;
      steps.push_back({"prop_enc", [=](hipStream_t st) { return vog_gemm_bias_act(&pe, st); }});
      se.bias = c->b_seg; se.relu = 1; se.c32 = ps32 + d.prop_enc;
      se.c16 = (unsigned short*)ps16 + d.prop_enc; se.ldc = g.d_obj; se.ldc16 = g.d_obj; se.rep = d.nppf0;
      steps.push_back({"seg_enc", [=](hipStream_t st) { return vog_gemm_bias_act(&se, st); }});
    }
  }
'''
s=s[:a]+new+s[b:]
open(p,'w').write(s)
