p='include/vog_hip.h'
s=open(p).read()
old='''  int splitk;
} vog_gemm_args;'''
new='''  int splitk;
  /* w_frag = 1: `w` is in MFMA-fragment order (vog_pack_w_frag: [N/16][K/32][64 lanes][8],
   * one contiguous KiB per fragment) — only for the M <= 64 weight-streaming kernel
   * (K % 32 == 0, N % 16 == 0), where the row-major layout makes every wave load touch
   * 16 half-used cache lines. */
  int w_frag;
} vog_gemm_args;
/* host: fp32 [N, ld] (first K columns) -> 16-bit fragment order, N*K halfwords. */
int vog_pack_w_frag(const float* w, int64_t ld, int N, int K, void* dst_host, vog_dtype dtype);'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

p='vognet-pytorch_amd/lib.py'
s=open(p).read()
s=s.replace('("res_vislang", c_vp), ("splitk", c_i32)]','("res_vislang", c_vp), ("splitk", c_i32), ("w_frag", c_i32)]')
s=s.replace('''    "vog_splitk_finish":''','''    "vog_pack_w_frag": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i32]),
    "vog_splitk_finish":''')
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
s=s.replace("  int splitk;\n  // implicit vis||lang residual","  int splitk; int w_frag;\n  // implicit vis||lang residual",1)
# skinny kernel: fragment-ordered W + optional m-tile split over grid.y
old='''  const int mt_n = (p.M + 15) / 16;           // <= 4
  const int ksteps = p.K / 32;'''
new='''  const int mt_all = (p.M + 15) / 16;         // <= 4
  // grid.y > 1: one 16-row tile of A per workgroup (few output columns: parallelism
  // matters more than re-reading the small W panel)
  const int mt_lo = gridDim.y > 1 ? blockIdx.y : 0;
  const int mt_n = gridDim.y > 1 ? mt_lo + 1 : mt_all;
  const int ksteps = p.K / 32;'''
assert old in s; s=s.replace(old,new)
old='''#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = mt * 16 + (lane & 15);
    a_ok[mt] = (mt < mt_n) && (m < p.M);'''
new='''#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = mt * 16 + (lane & 15);
    a_ok[mt] = (mt >= mt_lo) && (mt < mt_n) && (m < p.M);'''
assert old in s; s=s.replace(old,new)
old='''      fw[c] = load_a_chunk<T16, false>(p.w, w_off, ks * 32 + kg, n_ok && ks < ksteps);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt < mt_n) {'''
new='''      if (p.w_frag) {   // one contiguous KiB per (column tile, k-step)
        u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        fw[c] = ks < ksteps ? *reinterpret_cast<const u16x8*>(
                                  p.w + (((int64_t)blockIdx.x * ksteps + ks) * 64 + lane) * 8) : z;
      } else {
        fw[c] = load_a_chunk<T16, false>(p.w, w_off, ks * 32 + kg, n_ok && ks < ksteps);
      }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mt >= mt_lo && mt < mt_n) {'''
assert old in s; s=s.replace(old,new)
old='''  // wave w finishes m-tile w
  const int mt = wid;
  if (mt < mt_n) {'''
new='''  // wave w finishes m-tile w
  const int mt = wid;
  if (mt >= mt_lo && mt < mt_n) {'''
assert old in s; s=s.replace(old,new)
old='''  if (p.M <= 64 && (p.K % 32) == 0) {
    dim3 grid(ceil_div(p.N, 16));'''
new='''  p.w_frag = g->w_frag;
  if (p.w_frag && !(p.M <= 64 && (p.K % 32) == 0 && (p.N % 16) == 0))
    VOG_FAIL(-1, "w_frag weights are only valid for the M <= 64 kernel (M=%d N=%d K=%d)", p.M, p.N, p.K);
  if (p.M <= 64 && (p.K % 32) == 0) {
    const int ncol = ceil_div(p.N, 16);
    dim3 grid(ncol, ncol < 128 ? ceil_div(p.M, 16) : 1);'''
assert old in s; s=s.replace(old,new)
old='''extern "C" int vog_gemm_bias_act('''
new='''extern "C" int vog_pack_w_frag(const float* w, int64_t ld, int N, int K, void* dst_host, vog_dtype dtype) {
  VOG_CHECK_ARG(w && dst_host && N > 0 && K > 0 && (N % 16) == 0 && (K % 32) == 0 && ld >= K);
  unsigned short* dst = (unsigned short*)dst_host;
  const int ksteps = K / 32;
  for (int nt = 0; nt < N / 16; ++nt)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const float* src = w + (int64_t)(nt * 16 + (lane & 15)) * ld + ks * 32 + (lane >> 4) * 8;
        unsigned short* d = dst + (((int64_t)nt * ksteps + ks) * 64 + lane) * 8;
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
  return 0;
}

extern "C" int vog_gemm_bias_act('''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
# ctx members
s=s.replace("  std::vector<unsigned short*> wih;                     // [layer] [8R, in]","  std::vector<unsigned short*> wih;                     // [layer] [8R, in]\n  std::vector<unsigned short*> wih_f;                   // same, fragment order (M <= 64 kernel)\n  unsigned short* w_outproj_f = nullptr;")
s=s.replace("struct TxLayer {\n  unsigned short *wqkv, *wo, *w1, *w2;     // 16-bit, padded","struct TxLayer {\n  unsigned short *wqkv, *wo, *w1, *w2;     // 16-bit, padded\n  unsigned short* wqkv_lang_f;             // Wqkv[:, d_vis:] in fragment order (structured layer 0)")
s=s.replace("  c->wih.clear(); c->whh.clear(); c->bsum.clear();","  c->wih.clear(); c->wih_f.clear(); c->whh.clear(); c->bsum.clear();")
# pack wih fragment copies: build fp32 concat then pack
old='''    unsigned short *pw, *ph; float* pb;
    VOG_TRY(upload<unsigned short>(c, wih, &pw));'''
new='''    unsigned short *pw, *ph; float* pb;
    {
      std::vector<float> cat((size_t)8 * R * in);
      int dd = 0;
      for (const char* sfx : {"", "_reverse"}) {
        const auto& a = W(c, "lstm_encoder.lstm.weight_ih_l" + std::to_string(l) + sfx);
        memcpy(&cat[(size_t)dd * 4 * R * in], a.data(), a.size() * sizeof(float));
        ++dd;
      }
      std::vector<unsigned short> wf((size_t)8 * R * in);
      unsigned short* pf = nullptr;
      if (in % 32 == 0) {
        VOG_TRY(vog_pack_w_frag(cat.data(), in, 8 * R, in, wf.data(), (vog_dtype)et));
        VOG_TRY(upload<unsigned short>(c, wf, &pf));
      }
      c->wih_f.push_back(pf);
    }
    VOG_TRY(upload<unsigned short>(c, wih, &pw));'''
assert old in s; s=s.replace(old,new)
old='''  VOG_TRY(up16(c, "lstm_out_feat_proj.0.weight", et, &c->w_outproj));'''
new='''  VOG_TRY(up16(c, "lstm_out_feat_proj.0.weight", et, &c->w_outproj));
  c->w_outproj_f = nullptr;
  if (d.lang_enc % 16 == 0) {
    std::vector<unsigned short> wf((size_t)d.lang_enc * 2 * R);
    VOG_TRY(vog_pack_w_frag(W(c, "lstm_out_feat_proj.0.weight").data(), 2 * R, d.lang_enc, 2 * R, wf.data(), (vog_dtype)et));
    VOG_TRY(upload<unsigned short>(c, wf, &c->w_outproj_f));
  }'''
assert old in s; s=s.replace(old,new)
# tx layer: lang part of wqkv in fragment order (only meaningful for mul; built for layer 0 when dims allow)
old='''    VOG_TRY(upload<unsigned short>(c, wqkv, &L.wqkv));'''
new='''    VOG_TRY(upload<unsigned short>(c, wqkv, &L.wqkv));
    L.wqkv_lang_f = nullptr;
    {
      const int dl = c->d.lang_enc, dv = d - dl;
      if (l == 0 && std::string(prefix) == "mult_txf" && dv > 0 && dl % 32 == 0) {
        // fp32 image of the padded lang columns, then fragment order
        std::vector<float> wl((size_t)3 * H * dp * dl, 0.f);
        for (int which = 0; which < 3; ++which) {
          const auto& w = W(c, p + ".selfattn.layer." + nm[which] + ".weight");
          for (int h = 0; h < H; ++h)
            for (int dd = 0; dd < tw->head_dim[h]; ++dd)
              memcpy(&wl[((size_t)(which * H + h) * dp + dd) * dl], &w[(size_t)(tw->head_off[h] + dd) * d + dv],
                     dl * sizeof(float));
        }
        std::vector<unsigned short> wf(wl.size());
        VOG_TRY(vog_pack_w_frag(wl.data(), dl, 3 * H * dp, dl, wf.data(), (vog_dtype)dt));
        VOG_TRY(upload<unsigned short>(c, wf, &L.wqkv_lang_f));
      }
    }'''
assert old in s; s=s.replace(old,new)
# use in steps: gx GEMMs
old='''      ga.out_rows = lrows; ga.out_rows_ncol = 4 * R;'''
new='''      ga.out_rows = lrows; ga.out_rows_ncol = 4 * R;
      if (ga.M <= 64 && c->wih_f[l]) { ga.w = c->wih_f[l]; ga.w_frag = 1; }'''
assert old in s; s=s.replace(old,new)
old='''    po.rep = 1; po.dtype = et;'''
new='''    po.rep = 1; po.dtype = et;
    if (po.M <= 64 && c->w_outproj_f) { po.w = c->w_outproj_f; po.w_frag = 1; }'''
assert old in s; s=s.replace(old,new)
old='''      gl.rep = 1; gl.dtype = dt;'''
new='''      gl.rep = 1; gl.dtype = dt;
      if (gl.M <= 64 && L.wqkv_lang_f) { gl.w = L.wqkv_lang_f; gl.ldw = sv.dl; gl.w_frag = 1; }'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)
