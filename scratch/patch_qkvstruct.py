p='include/vog_hip.h'
s=open(p).read()
old='''  void* q; void* k; void* vt;
  int S, N, H, dp, npad, K; vog_dtype dtype;
} vog_qkv_args;'''
new='''  void* q; void* k; void* vt;
  int S, N, H, dp, npad, K; vog_dtype dtype;
  /* structured layer 0 of mul_tx (pl != NULL): x16 holds the n_vid*nfrm*nppf VISUAL rows
   * only ([rows, K = d_vis]), wqkv its first K columns (row pitch ldw); pl = lang Wqkv[:, d_vis:]^T
   * ([n_lang*nsrl, 3*H*dp] fp32). The epilogue emits, for every visual row and every one of
   * the nsrl arguments, token (arg*nppf + p) = projection + pl[lang row of that arg]:
   * S = n_vid*nfrm sequences of N = nsrl*nppf tokens, no [tokens, d] matrix and no fp32
   * intermediate in HBM (see vog_qkv_combine for the unfused form). */
  const float* pl; int nsrl, nppf, nfrm, lang_per_vid, nc_v;
} vog_qkv_args;'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

p='vognet-pytorch_amd/lib.py'
s=open(p).read()
old='''                ("S", c_i32), ("N", c_i32), ("H", c_i32), ("dp", c_i32), ("npad", c_i32),
                ("K", c_i32), ("dtype", c_i32)]


class QkvCombArgs'''
new='''                ("S", c_i32), ("N", c_i32), ("H", c_i32), ("dp", c_i32), ("npad", c_i32),
                ("K", c_i32), ("dtype", c_i32),
                ("pl", c_vp), ("nsrl", c_i32), ("nppf", c_i32), ("nfrm", c_i32), ("lang_per_vid", c_i32),
                ("nc_v", c_i32)]


class QkvCombArgs'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
s=s.replace("  int ntok, H, dp, npad;\n};","  int ntok, H, dp, npad;\n  // structured QKV (pl != nullptr): rows are visual rows, fan out over nsrl arguments\n  const float* pl; int st_nsrl, st_nppf, st_nfrm, st_lpv, st_ncv;\n};",1)
# pipe epilogue QKV branch: structured path
old='''      if (which < 2) {
        unsigned short* base = which == 0 ? p.q : p.k;
        const int c = lane & 7, rsub = lane >> 3;     // 8 chunks of 4 columns per row, 8 rows per pass'''
new='''      if (p.pl) {
        // structured layer 0: row m = visual row (v, f, p'); token(arg) = arg*nppf + p'
        const int ldp = 3 * hd;
        if (which < 2) {
          unsigned short* base = which == 0 ? p.q : p.k;
          const int c = lane & 7, rsub = lane >> 3;
#pragma unroll 2
          for (int ps = 0; ps < WTM / 8; ++ps) {
            const int rl = ps * 8 + rsub;
            const int m = mw + rl;
            if (m >= p.M) continue;
            const float4 v = *reinterpret_cast<const float4*>(&ep[rl * EP_LD + cg * 32 + 4 * c]);
            const int sq = m / p.st_nppf, pp = m - sq * p.st_nppf;
            const int vid = sq / p.st_nfrm;
            const int lv = p.st_lpv ? vid : vid / p.st_ncv;
            const float* plr = p.pl + (int64_t)lv * p.st_nsrl * ldp + nb + 4 * c;
            unsigned short* dst = base + ((int64_t)sq * p.H + h) * p.npad * p.dp;
            for (int ar = 0; ar < p.st_nsrl; ++ar) {
              const float4 l = *reinterpret_cast<const float4*>(plr + (int64_t)ar * ldp);
              const u16x4 o = {to16<T16>(v.x + l.x), to16<T16>(v.y + l.y), to16<T16>(v.z + l.z), to16<T16>(v.w + l.w)};
              *reinterpret_cast<u16x4*>(dst + frag_qk(ar * p.st_nppf + pp, dd0 + 4 * c, p.dp)) = o;
            }
          }
        } else {
#pragma unroll
          for (int th = 0; th < WTM / 64 + (WTM % 64 ? 1 : 0); ++th) {
            const int rl = th * 64 + lane;
            const int m = mw + rl;
            if (rl < WTM && m < p.M) {
              const int sq = m / p.st_nppf, pp = m - sq * p.st_nppf;
              const int vid = sq / p.st_nfrm;
              const int lv = p.st_lpv ? vid : vid / p.st_ncv;
              const float* plr = p.pl + (int64_t)lv * p.st_nsrl * ldp + nb;
              unsigned short* dst = p.vt + ((int64_t)sq * p.H + h) * p.npad * p.dp;
              for (int ar = 0; ar < p.st_nsrl; ++ar) {
                unsigned short* d2 = dst + frag_v(ar * p.st_nppf + pp, dd0, p.dp);
                const float* l = plr + (int64_t)ar * ldp;
#pragma unroll 8
                for (int dd = 0; dd < 32; ++dd)
                  d2[dd * 8] = to16<T16>(ep[rl * EP_LD + cg * 32 + dd] + l[dd]);
              }
            }
          }
        }
      } else if (which < 2) {
        unsigned short* base = which == 0 ? p.q : p.k;
        const int c = lane & 7, rsub = lane >> 3;     // 8 chunks of 4 columns per row, 8 rows per pass'''
assert old in s; s=s.replace(old,new)
# qkv_run: structured params
old='''  p.M = a->S * a->N; p.N = 3 * a->H * a->dp; p.K = a->K; p.rep = 1;'''
new='''  p.M = a->S * a->N; p.N = 3 * a->H * a->dp; p.K = a->K; p.rep = 1;
  if (a->pl) {
    VOG_CHECK_ARG(a->nsrl > 0 && a->nppf > 0 && a->nfrm > 0 && a->nc_v > 0 && a->N == a->nsrl * a->nppf &&
                  (a->S % a->nfrm) == 0);
    p.pl = a->pl; p.st_nsrl = a->nsrl; p.st_nppf = a->nppf; p.st_nfrm = a->nfrm; p.st_lpv = a->lang_per_vid;
    p.st_ncv = a->nc_v;
    p.M = a->S * a->nppf;                        // visual rows
    if (!pipe_ok(p, false) || p.M <= 64)
      VOG_FAIL(-1, "structured QKV needs the LDS-DMA GEMM (K %% 64 == 0, > 64 visual rows)");
  }'''
assert old in s; s=s.replace(old,new)
# pipe_ok is defined after qkv_run? ensure declaration order: add forward declaration near top
s=s.replace("enum { EPI_PLAIN = 0, EPI_QKV = 1 };","enum { EPI_PLAIN = 0, EPI_QKV = 1 };\nstruct GemmParams;\nstatic bool pipe_ok(const GemmParams& p, bool a_f32);",1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
old=s[s.index('      const vog_vislang_args sv = *structured;\n      const int ncol = 3 * tw.H * tw.dp;'):s.index('    } else {\n      steps.push_back({n + "_qkv"')]
new='''      const vog_vislang_args sv = *structured;
      vog_qkv_args qs = qa;
      qs.x16 = vis16; qs.ldx = sv.dv; qs.K = sv.dv;          // visual rows x first d_vis weight columns
      qs.pl = ws.at<float>(n + "_pl"); qs.nsrl = sv.nsrl; qs.nppf = sv.nppf; qs.nfrm = sv.nfrm;
      qs.lang_per_vid = sv.lang_per_vid; qs.nc_v = sv.nc_v;
      steps.push_back({n + "_pv", [=](hipStream_t st) { return vog_qkv_proj(&qs, st); }});
'''
s=s.replace(old,new)
# structured needs > 64 visual rows
s=s.replace("  const bool structured = has_mul(d) && (g.d_obj % 64) == 0 && (g.L % 32) == 0;","  const bool structured = has_mul(d) && (g.d_obj % 64) == 0 && (g.L % 32) == 0 && g.rows_obj > 64;")
open(p,'w').write(s)
