# ---------------- header
p='include/vog_hip.h'
s=open(p).read()
old='''  const int32_t* out_rows; int out_rows_ncol;
} vog_gemm_args;'''
new='''  const int32_t* out_rows; int out_rows_ncol;
  /* optional implicit residual: the residual row of output row m is the vis||lang
   * token m of this layout (never materialised); `residual` must then be NULL. */
  const struct vog_vislang_args* res_vislang;
} vog_gemm_args;'''
assert old in s; s=s.replace(old,new)
# move vislang struct decl before gemm args: add forward declaration
s=s.replace("typedef struct vog_gemm_args {","struct vog_vislang_args;\ntypedef struct vog_gemm_args {",1)
old='''/* softmax((q k^T + bias)/scale) v per (sequence, head), flash-style, with the'''
new='''/* Layer-0 QKV of mul_tx through the token structure: every token is
 * [vis[v, f*nppf+p] || lang[l, a]], so x Wqkv^T = PV[vis row] + PL[lang row] with
 * PV = vis Wqkv[:, :dv]^T ([n_vid*NP, 3*H*dp] fp32) and PL = lang Wqkv[:, dv:]^T
 * ([n_lang*nsrl, 3*H*dp] fp32): 5x fewer projection FLOPs than the dense
 * [tokens, d] GEMM of transformer_code.py:180 and no token matrix in HBM. This
 * entry adds the two parts (one rounding to 16 bit) and emits q,k [S,H,N,dp] and
 * v^T [S,H,dp,npad] exactly as vog_qkv_proj does. */
typedef struct vog_qkvcomb_args {
  const float* pv; const float* pl; void* q; void* k; void* vt;
  int n_vid, nfrm, nppf, nsrl, H, dp, npad; int lang_per_vid, nc_v; vog_dtype dtype;
} vog_qkvcomb_args;
int vog_qkv_combine(const vog_qkvcomb_args* a, void* stream);

/* softmax((q k^T + bias)/scale) v per (sequence, head), flash-style, with the'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

# ---------------- lib.py
p='vognet-pytorch_amd/lib.py'
s=open(p).read()
s=s.replace('("out_rows", c_vp), ("out_rows_ncol", c_i32)]','("out_rows", c_vp), ("out_rows_ncol", c_i32), ("res_vislang", c_vp)]')
s=s.replace('''class AttnArgs(C.Structure):''','''class QkvCombArgs(C.Structure):
    _fields_ = [("pv", c_vp), ("pl", c_vp), ("q", c_vp), ("k", c_vp), ("vt", c_vp),
                ("n_vid", c_i32), ("nfrm", c_i32), ("nppf", c_i32), ("nsrl", c_i32), ("H", c_i32),
                ("dp", c_i32), ("npad", c_i32), ("lang_per_vid", c_i32), ("nc_v", c_i32),
                ("dtype", c_i32)]


class AttnArgs(C.Structure):''')
s=s.replace('''    "vog_rel_attention_fwd":''','''    "vog_qkv_combine": (c_i32, [C.POINTER(QkvCombArgs), c_vp]),
    "vog_rel_attention_fwd":''')
open(p,'w').write(s)

# ---------------- gemm.hip: implicit vislang residual in pipe epilogue + scalar epilogue
p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
s=s.replace("  const int32_t* out_rows; int out_rows_ncol;\n","  const int32_t* out_rows; int out_rows_ncol;\n  // implicit vis||lang residual (res_vis != nullptr)\n  const float* res_vis; const float* res_lang; int rv_nfrm, rv_nppf, rv_nsrl, rv_dv, rv_dl, rv_lpv, rv_ncv;\n",1)
helper='''// residual pointer of token row m, column n, for the implicit vis||lang token matrix
// (row m = (s=(v,f), j=a*nppf+p); a 4-column chunk never straddles dv since dv % 4 == 0)
__device__ __forceinline__ const float* vislang_res_ptr(const GemmParams& p, int m, int n) {
  const int N = p.rv_nsrl * p.rv_nppf;
  const int s = m / N, j = m - s * N;
  const int v = s / p.rv_nfrm, f = s - v * p.rv_nfrm;
  const int a = j / p.rv_nppf, pp = j - a * p.rv_nppf;
  if (n < p.rv_dv)
    return p.res_vis + ((int64_t)v * p.rv_nfrm * p.rv_nppf + (int64_t)f * p.rv_nppf + pp) * p.rv_dv + n;
  const int lv = p.rv_lpv ? v : v / p.rv_ncv;
  return p.res_lang + ((int64_t)lv * p.rv_nsrl + a) * p.rv_dl + (n - p.rv_dv);
}

template <typename T16>
__device__ __forceinline__ void epilogue_store('''
s=s.replace("template <typename T16>\n__device__ __forceinline__ void epilogue_store(",helper,1)
s=s.replace("    if (p.residual) v += p.residual[(int64_t)row * p.ldr + col];\n    if (p.relu) v = fmaxf(v, 0.f);\n    if (p.out_rows) {","    if (p.residual) v += p.residual[(int64_t)row * p.ldr + col];\n    if (p.res_vis) v += *vislang_res_ptr(p, row, col);\n    if (p.relu) v = fmaxf(v, 0.f);\n    if (p.out_rows) {")
old='''          if (p.residual) {
            const float4 r = *reinterpret_cast<const float4*>(p.residual + (int64_t)m * p.ldr + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          if (p.relu) { v.x = fmaxf(v.x, 0.f);'''
new='''          if (p.residual) {
            const float4 r = *reinterpret_cast<const float4*>(p.residual + (int64_t)m * p.ldr + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          if (p.res_vis) {
            const float4 r = *reinterpret_cast<const float4*>(vislang_res_ptr(p, m, n));
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          if (p.relu) { v.x = fmaxf(v.x, 0.f);'''
assert old in s; s=s.replace(old,new)
old='''  p.out_rows = g->out_rows; p.out_rows_ncol = g->out_rows_ncol;'''
new='''  p.out_rows = g->out_rows; p.out_rows_ncol = g->out_rows_ncol;
  if (g->res_vislang) {
    const vog_vislang_args* r = g->res_vislang;
    p.res_vis = r->vis; p.res_lang = r->lang; p.rv_nfrm = r->nfrm; p.rv_nppf = r->nppf; p.rv_nsrl = r->nsrl;
    p.rv_dv = r->dv; p.rv_dl = r->dl; p.rv_lpv = r->lang_per_vid; p.rv_ncv = r->nc_v;
  }'''
assert old in s; s=s.replace(old,new)
old='''  VOG_CHECK_ARG(!g->out_rows ||'''
new='''  VOG_CHECK_ARG(!g->res_vislang || (!g->residual && g->res_vislang->vis && g->res_vislang->lang &&
                                    (g->res_vislang->dv % 4) == 0 && (g->res_vislang->dl % 4) == 0 &&
                                    g->res_vislang->dv + g->res_vislang->dl == g->N));
  VOG_CHECK_ARG(!g->out_rows ||'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

# ---------------- elementwise.hip: combine kernel
p='vognet-pytorch_amd/csrc/elementwise.hip'
s=open(p).read()
old="// ---------------------------------------------------------------------------\n// K7 score head tail"
new=r'''// ---------------------------------------------------------------------------
// structured layer-0 QKV of mul_tx: q/k/v[token(a,p)] = PV[vis row p] + PL[lang row a]
// (see vog_qkvcomb_args). grid (sequence, head, {q,k,v}); each PV element is read
// once and fanned out to the nsrl tokens that share it.
// ---------------------------------------------------------------------------
template <typename T16>
__global__ __launch_bounds__(256) void qkv_combine_kernel(vog_qkvcomb_args a) {
  const int s = blockIdx.x, h = blockIdx.y, which = blockIdx.z;
  const int v = s / a.nfrm, f = s - v * a.nfrm;
  const int ldp = 3 * a.H * a.dp;
  const int col0 = (which * a.H + h) * a.dp;
  const int lv = a.lang_per_vid ? v : v / a.nc_v;
  const float* pv = a.pv + ((int64_t)v * a.nfrm * a.nppf + (int64_t)f * a.nppf) * ldp + col0;
  const float* pl = a.pl + (int64_t)lv * a.nsrl * ldp + col0;
  const int N = a.nsrl * a.nppf;
  const int64_t sh = (int64_t)s * a.H + h;
  if (which < 2) {
    unsigned short* dst = reinterpret_cast<unsigned short*>(which == 0 ? a.q : a.k) + sh * N * a.dp;
    const int cpr = a.dp / 8;                       // 8-column chunks per row
    for (int it = threadIdx.x; it < a.nppf * cpr; it += blockDim.x) {
      const int pp = it / cpr, c = it - pp * cpr;
      const float4 x0 = *reinterpret_cast<const float4*>(pv + (int64_t)pp * ldp + c * 8);
      const float4 x1 = *reinterpret_cast<const float4*>(pv + (int64_t)pp * ldp + c * 8 + 4);
      for (int ar = 0; ar < a.nsrl; ++ar) {
        const float4 l0 = *reinterpret_cast<const float4*>(pl + (int64_t)ar * ldp + c * 8);
        const float4 l1 = *reinterpret_cast<const float4*>(pl + (int64_t)ar * ldp + c * 8 + 4);
        u16x8 o = {to16<T16>(x0.x + l0.x), to16<T16>(x0.y + l0.y), to16<T16>(x0.z + l0.z), to16<T16>(x0.w + l0.w),
                   to16<T16>(x1.x + l1.x), to16<T16>(x1.y + l1.y), to16<T16>(x1.z + l1.z), to16<T16>(x1.w + l1.w)};
        *reinterpret_cast<u16x8*>(dst + ((int64_t)ar * a.nppf + pp) * a.dp + c * 8) = o;
      }
    }
  } else {
    // V^T[dd][token]: thread = (dd, group of 4 proposals); lanes run along dd so the PV
    // reads are coalesced; each thread emits nsrl 8-byte stores
    unsigned short* dst = reinterpret_cast<unsigned short*>(a.vt) + sh * a.dp * a.npad;
    const int ng = (a.nppf + 3) / 4;
    const bool vec = (a.nppf & 3) == 0;
    for (int it = threadIdx.x; it < a.dp * ng; it += blockDim.x) {
      const int g = it / a.dp, dd = it - g * a.dp;
      float x[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int pp = g * 4 + e;
        x[e] = pp < a.nppf ? pv[(int64_t)pp * ldp + dd] : 0.f;
      }
      for (int ar = 0; ar < a.nsrl; ++ar) {
        const float l = pl[(int64_t)ar * ldp + dd];
        unsigned short* d = dst + (int64_t)dd * a.npad + ar * a.nppf + g * 4;
        if (vec) {
          u16x4 o = {to16<T16>(x[0] + l), to16<T16>(x[1] + l), to16<T16>(x[2] + l), to16<T16>(x[3] + l)};
          *reinterpret_cast<u16x4*>(d) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (g * 4 + e < a.nppf) d[e] = to16<T16>(x[e] + l);
        }
      }
    }
  }
}

'''+old
assert old in s; s=s.replace(old,new,1)
old='''extern "C" int vog_score_head('''
new='''extern "C" int vog_qkv_combine(const vog_qkvcomb_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->pv && a->pl && a->q && a->k && a->vt);
  VOG_CHECK_ARG((a->dp % 8) == 0 && a->npad >= a->nsrl * a->nppf && (a->npad % 4) == 0);
  dim3 grid(a->n_vid * a->nfrm, a->H, 3);
  VOG_DISPATCH_DTYPE(a->dtype, hipLaunchKernelGGL((qkv_combine_kernel<T16>), grid, dim3(256), 0,
                     (hipStream_t)stream, *a));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_score_head('''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

# ---------------- forward.hip
p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
# plan: PV / PL buffers for mul
old='''  if (has_mul(d)) tx("mul", c->mul, g.rows_mul, g.S_mul, g.npad_mul);'''
new='''  if (has_mul(d)) {
    tx("mul", c->mul, g.rows_mul, g.S_mul, g.npad_mul);
    p.add("mul_pv", g.rows_obj * 3 * c->mul.H * c->mul.dp * 4);
    p.add("mul_pl", (int64_t)g.Bn * d.nsrl * 3 * c->mul.H * c->mul.dp * 4);
  }'''
assert old in s; s=s.replace(old,new)
# tx_steps signature: structured input
old='''                     int npad, int spv, int n_box, float fdiv, int last_dt, std::vector<Step>& steps,
                     const float** out32, const void** out16) {'''
new='''                     int npad, int spv, int n_box, float fdiv, int last_dt, std::vector<Step>& steps,
                     const float** out32, const void** out16,
                     const vog_vislang_args* structured = nullptr, const void* vis16 = nullptr) {'''
assert old in s; s=s.replace(old,new)
old='''    vog_qkv_args qa{};
    qa.x16 = cur16; qa.ldx = tw.d; qa.wqkv = L.wqkv; qa.ldw = tw.d;
    qa.q = ws.at<void>(n + "_q"); qa.k = ws.at<void>(n + "_k"); qa.vt = ws.at<void>(n + "_vt");
    qa.S = S; qa.N = N; qa.H = tw.H; qa.dp = tw.dp; qa.npad = npad; qa.K = tw.d; qa.dtype = dt;
    steps.push_back({n + "_qkv", [=](hipStream_t st) { return vog_qkv_proj(&qa, st); }});'''
new='''    vog_qkv_args qa{};
    qa.x16 = cur16; qa.ldx = tw.d; qa.wqkv = L.wqkv; qa.ldw = tw.d;
    qa.q = ws.at<void>(n + "_q"); qa.k = ws.at<void>(n + "_k"); qa.vt = ws.at<void>(n + "_vt");
    qa.S = S; qa.N = N; qa.H = tw.H; qa.dp = tw.dp; qa.npad = npad; qa.K = tw.d; qa.dtype = dt;
    const bool fact = structured && l == 0;
    if (fact) {
      // layer 0 of mul_tx: tokens are [vis[p] || lang[a]] -> project the two parts once each
      const vog_vislang_args sv = *structured;
      const int ncol = 3 * tw.H * tw.dp;
      vog_gemm_args gv{}; gv.c16_dtype = -1;
      gv.a = vis16; gv.lda = sv.dv; gv.w = L.wqkv; gv.ldw = tw.d; gv.c32 = ws.at<float>(n + "_pv"); gv.ldc = ncol;
      gv.M = (int)g.rows_obj; gv.N = ncol; gv.K = sv.dv; gv.rep = 1; gv.dtype = dt;
      steps.push_back({n + "_pv", [=](hipStream_t st) { return vog_gemm_bias_act(&gv, st); }});
      vog_gemm_args gl{}; gl.c16_dtype = -1;
      gl.a = sv.lang; gl.a_is_f32 = 1; gl.lda = sv.dl; gl.w = L.wqkv + sv.dv; gl.ldw = tw.d;
      gl.c32 = ws.at<float>(n + "_pl"); gl.ldc = ncol; gl.M = g.Bn * sv.nsrl; gl.N = ncol; gl.K = sv.dl;
      gl.rep = 1; gl.dtype = dt;
      steps.push_back({n + "_pl", [=](hipStream_t st) { return vog_gemm_bias_act(&gl, st); }});
      vog_qkvcomb_args ca{};
      ca.pv = gv.c32; ca.pl = gl.c32; ca.q = qa.q; ca.k = qa.k; ca.vt = qa.vt;
      ca.n_vid = sv.n_vid; ca.nfrm = sv.nfrm; ca.nppf = sv.nppf; ca.nsrl = sv.nsrl; ca.H = tw.H; ca.dp = tw.dp;
      ca.npad = npad; ca.lang_per_vid = sv.lang_per_vid; ca.nc_v = sv.nc_v; ca.dtype = dt;
      steps.push_back({n + "_combine", [=](hipStream_t st) { return vog_qkv_combine(&ca, st); }});
    } else {
      steps.push_back({n + "_qkv", [=](hipStream_t st) { return vog_qkv_proj(&qa, st); }});
    }'''
assert old in s; s=s.replace(old,new)
old='''    wo.residual = cur32; wo.ldr = tw.d; wo.c32 = ws.at<float>(n + "_tmp"); wo.ldc = tw.d;
    wo.M = (int)rows; wo.N = tw.d; wo.K = tw.H * tw.dp; wo.rep = 1; wo.dtype = dt;
    steps.push_back({n + "_wo", [=](hipStream_t st) { return vog_gemm_bias_act(&wo, st); }});'''
new='''    wo.residual = cur32; wo.ldr = tw.d; wo.c32 = ws.at<float>(n + "_tmp"); wo.ldc = tw.d;
    wo.M = (int)rows; wo.N = tw.d; wo.K = tw.H * tw.dp; wo.rep = 1; wo.dtype = dt;
    if (fact) {
      const vog_vislang_args sv = *structured;       // residual = the (unmaterialised) token matrix
      wo.residual = nullptr;
      steps.push_back({n + "_wo", [=](hipStream_t st) {
        vog_gemm_args w2 = wo; w2.res_vislang = &sv; return vog_gemm_bias_act(&w2, st); }});
    } else {
      steps.push_back({n + "_wo", [=](hipStream_t st) { return vog_gemm_bias_act(&wo, st); }});
    }'''
assert old in s; s=s.replace(old,new)
# obj: write tx-typed 16-bit copy of the last layer when mul consumes it
old='''             g.fdiv_obj, -1, steps, &vis32, &vis16);'''
new='''             g.fdiv_obj, has_mul(d) ? d.tx_dtype : -1, steps, &vis32, &vis16);'''
assert old in s; s=s.replace(old,new)
# vislang: only when mul_tx is absent; otherwise structured
old='''  {
    vog_vislang_args va{};
    va.vis = vis32; va.lang = ws.at<float>("lang"); va.x32 = ws.at<float>("xmul"); va.x16 = ws.at<void>("xmul16");
    va.n_vid = g.n_vid; va.nfrm = g.nfrm; va.nppf = g.nppf; va.nsrl = d.nsrl; va.dv = g.d_obj; va.dl = g.L;
    va.lang_per_vid = g.nvl > 1 ? 1 : 0; va.nc_v = g.nc_v; va.dtype = (vog_dtype)(has_mul(d) ? d.tx_dtype : d.enc_dtype);
    steps.push_back({"vislang", [=](hipStream_t st) { return vog_vislang_layout(&va, st); }});
  }'''
new='''  vog_vislang_args va{};
  va.vis = vis32; va.lang = ws.at<float>("lang"); va.x32 = ws.at<float>("xmul"); va.x16 = ws.at<void>("xmul16");
  va.n_vid = g.n_vid; va.nfrm = g.nfrm; va.nppf = g.nppf; va.nsrl = d.nsrl; va.dv = g.d_obj; va.dl = g.L;
  va.lang_per_vid = g.nvl > 1 ? 1 : 0; va.nc_v = g.nc_v; va.dtype = (vog_dtype)(has_mul(d) ? d.tx_dtype : d.enc_dtype);
  // mul_tx consumes the token structure directly (layer-0 QKV and its residual), so the
  // token matrix is only materialised for ImgGrnd / VidGrnd, whose lin2 reads it
  const bool structured = has_mul(d) && (g.d_obj % 64) == 0 && (g.L % 32) == 0;
  if (!structured)
    steps.push_back({"vislang", [=](hipStream_t st) { return vog_vislang_layout(&va, st); }});'''
assert old in s; s=s.replace(old,new)
old='''             (float)g.nfrm, d.enc_dtype, steps, &x32, &x16);'''
new='''             (float)g.nfrm, d.enc_dtype, steps, &x32, &x16, structured ? &va : nullptr, vis16);'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)
