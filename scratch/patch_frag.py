import re
# ---------------- common.h: fragment-order index helpers
p='vognet-pytorch_amd/csrc/common.h'
s=open(p).read()
old="static inline int ceil_div(int a, int b)"
new='''// ---- fragment-ordered attention operands ---------------------------------------------
// q, k and v of one (sequence, head) are stored in the order the attention kernel's MFMA
// operands consume them, npad = N rounded up to 32 tokens, npad*dp halfwords each:
//   q/k : [token/32][dd/16][lane = ((dd/8)&1)*32 + token%32][dd%8]
//         (= A/B fragment of v_mfma_f32_32x32x16: row token%32, k = dd%16)
//   v   : [token/32][dd/32][ks = (token%32)/16][lane = hi*32 + dd%32][j]
//         with token%16 = 8*(j>>2) + 4*hi + (j&3)  (the key permutation under which the
//         S^T accumulator registers are directly the P^T operand, attention.hip)
// Every fragment is one contiguous KiB: wave loads and LDS-DMA need no swizzle.
__host__ __device__ __forceinline__ int64_t frag_qk(int i, int dd, int dp) {
  return ((int64_t)(i >> 5) * (dp >> 4) + (dd >> 4)) * 512 + ((((dd >> 3) & 1) << 5) + (i & 31)) * 8 + (dd & 7);
}
__host__ __device__ __forceinline__ int64_t frag_v(int i, int dd, int dp) {
  const int kl = i & 31, r = kl & 15;
  const int j = ((r >> 3) << 2) + (r & 3), hi = (r >> 2) & 1;
  return ((((int64_t)(i >> 5) * (dp >> 5) + (dd >> 5)) * 2 + (kl >> 4)) * 64 + (hi << 5) + (dd & 31)) * 8 + j;
}

static inline int ceil_div(int a, int b)'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

# ---------------- gemm.hip
p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
# (1) tiled-kernel per-fragment epilogue
old='''    const int64_t sh = (int64_t)s * p.H + h;
    const unsigned short o = to16<T16>(acc[r]);
    if (which == 0) {
      p.q[(sh * p.ntok + i) * p.dp + dd] = o;
    } else if (which == 1) {
      p.k[(sh * p.ntok + i) * p.dp + dd] = o;
    } else {
      p.vt[(sh * p.dp + dd) * p.npad + i] = o;
    }'''
new='''    const int64_t base = ((int64_t)s * p.H + h) * p.npad * p.dp;
    const unsigned short o = to16<T16>(acc[r]);
    if (which == 0) {
      p.q[base + frag_qk(i, dd, p.dp)] = o;
    } else if (which == 1) {
      p.k[base + frag_qk(i, dd, p.dp)] = o;
    } else {
      p.vt[base + frag_v(i, dd, p.dp)] = o;
    }'''
assert old in s; s=s.replace(old,new)
# (2) drop dead swapped helpers
a=s.index("// ---- epilogues for the swapped (C^T) accumulator layout")
b=s.index("// ----------------------------------------------------------------------------\n// pipelined kernel: K % 64 == 0")
s=s[:a]+s[b:]
# (3) pipe LDS epilogue QKV part
old='''          const int sq = m / p.ntok, tok = m - sq * p.ntok;
          const u16x4 o = {to16<T16>(v.x), to16<T16>(v.y), to16<T16>(v.z), to16<T16>(v.w)};
          *reinterpret_cast<u16x4*>(base + (((int64_t)sq * p.H + h) * p.ntok + tok) * p.dp + dd0 + 4 * c) = o;'''
new='''          const int sq = m / p.ntok, tok = m - sq * p.ntok;
          const u16x4 o = {to16<T16>(v.x), to16<T16>(v.y), to16<T16>(v.z), to16<T16>(v.w)};
          *reinterpret_cast<u16x4*>(base + ((int64_t)sq * p.H + h) * p.npad * p.dp +
                                    frag_qk(tok, dd0 + 4 * c, p.dp)) = o;'''
assert old in s; s=s.replace(old,new)
old='''            unsigned short* dst = p.vt + (((int64_t)sq * p.H + h) * p.dp + dd0) * p.npad + tok;
#pragma unroll 8
            for (int dd = 0; dd < 32; ++dd)
              dst[(int64_t)dd * p.npad] = to16<T16>(ep[rl * EP_LD + cg * 32 + dd]);'''
new='''            // dd0 % 32 == 0: the 32 columns of this group are one d-block of the V fragment
            unsigned short* dst = p.vt + ((int64_t)sq * p.H + h) * p.npad * p.dp + frag_v(tok, dd0, p.dp);
#pragma unroll 8
            for (int dd = 0; dd < 32; ++dd)
              dst[dd * 8] = to16<T16>(ep[rl * EP_LD + cg * 32 + dd]);'''
assert old in s; s=s.replace(old,new)
s=s.replace("  VOG_CHECK_ARG(a->K % 8 == 0 && a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->npad >= a->N);","  VOG_CHECK_ARG(a->K % 8 == 0 && a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->npad >= a->N && (a->npad % 32) == 0 && (a->dp % 32) == 0);")
s=s.replace("        // V^T[dd][token]: lane = token, so each store instruction writes a token-contiguous run","        // V fragments: lane = token; 2-byte stores inside this token's fragment block")
open(p,'w').write(s)

# ---------------- elementwise.hip combine kernel
p='vognet-pytorch_amd/csrc/elementwise.hip'
s=open(p).read()
old='''  const int N = a.nsrl * a.nppf;
  const int64_t sh = (int64_t)s * a.H + h;
  if (which < 2) {
    unsigned short* dst = reinterpret_cast<unsigned short*>(which == 0 ? a.q : a.k) + sh * N * a.dp;'''
new='''  const int64_t sh = (int64_t)s * a.H + h;
  if (which < 2) {
    unsigned short* dst = reinterpret_cast<unsigned short*>(which == 0 ? a.q : a.k) + sh * a.npad * a.dp;'''
assert old in s; s=s.replace(old,new)
old='''        *reinterpret_cast<u16x8*>(dst + ((int64_t)ar * a.nppf + pp) * a.dp + c * 8) = o;'''
new='''        *reinterpret_cast<u16x8*>(dst + frag_qk(ar * a.nppf + pp, c * 8, a.dp)) = o;'''
assert old in s; s=s.replace(old,new)
old='''    unsigned short* dst = reinterpret_cast<unsigned short*>(a.vt) + sh * a.dp * a.npad;'''
new='''    unsigned short* dst = reinterpret_cast<unsigned short*>(a.vt) + sh * a.npad * a.dp;'''
assert old in s; s=s.replace(old,new)
old='''        unsigned short* d = dst + (int64_t)dd * a.npad + ar * a.nppf + g * 4;
        if (vec) {
          u16x4 o = {to16<T16>(x[0] + l), to16<T16>(x[1] + l), to16<T16>(x[2] + l), to16<T16>(x[3] + l)};
          *reinterpret_cast<u16x4*>(d) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (g * 4 + e < a.nppf) d[e] = to16<T16>(x[e] + l);
        }'''
new='''        const int tok = ar * a.nppf + g * 4;
        if (vec) {   // 4 consecutive tokens, tok % 4 == 0 -> 4 consecutive j of one fragment lane
          u16x4 o = {to16<T16>(x[0] + l), to16<T16>(x[1] + l), to16<T16>(x[2] + l), to16<T16>(x[3] + l)};
          *reinterpret_cast<u16x4*>(dst + frag_v(tok, dd, a.dp)) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (g * 4 + e < a.nppf) dst[frag_v(tok + e, dd, a.dp)] = to16<T16>(x[e] + l);
        }'''
assert old in s; s=s.replace(old,new)
s=s.replace("  VOG_CHECK_ARG((a->dp % 8) == 0 && a->npad >= a->nsrl * a->nppf && (a->npad % 4) == 0);","  VOG_CHECK_ARG((a->dp % 32) == 0 && a->npad >= a->nsrl * a->nppf && (a->npad % 32) == 0);")
open(p,'w').write(s)

# ---------------- forward.hip: npad = round_up(N,32), buffer sizes
p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
s=s.replace("g.npad_obj = (int)round_up64(g.N_obj, 64);","g.npad_obj = (int)round_up64(g.N_obj, 32);")
s=s.replace("g.npad_mul = (int)round_up64(g.N_mul, 64);","g.npad_mul = (int)round_up64(g.N_mul, 32);")
old='''    p.add(n + "_q", rows * tw.H * tw.dp * 2);
    p.add(n + "_k", rows * tw.H * tw.dp * 2);
    p.add(n + "_vt", (int64_t)S * tw.H * tw.dp * npad * 2);'''
new='''    p.add(n + "_q", (int64_t)S * tw.H * tw.dp * npad * 2);      // fragment order, npad = N up to 32
    p.add(n + "_k", (int64_t)S * tw.H * tw.dp * npad * 2);
    p.add(n + "_vt", (int64_t)S * tw.H * tw.dp * npad * 2);'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)

# ---------------- header docs
p='include/vog_hip.h'
s=open(p).read()
old=''' * to dp columns with zero rows). Writes q,k as [S,H,N,dp] and v TRANSPOSED as
 * vt [S,H,dp,npad] (npad = N rounded up to 64; pad region must be zeroed once). */'''
new=''' * to dp columns with zero rows, dp % 32 == 0). Writes q, k, v per (sequence, head)
 * in MFMA-FRAGMENT ORDER, npad*dp halfwords each, npad = N rounded up to 32:
 *   q/k : [token/32][dd/16][lane = ((dd/8)&1)*32 + token%32][dd%8]
 *   v   : [token/32][dd/32][(token%32)/16][lane = hi*32 + dd%32][j],
 *         token%16 = 8*(j>>2) + 4*hi + (j&3)
 * (csrc/common.h frag_qk / frag_v). Pad tokens are never written: zero the
 * buffers once (vog_workspace_init does). */'''
assert old in s; s=s.replace(old,new)
old=''' * entry adds the two parts (one rounding to 16 bit) and emits q,k [S,H,N,dp] and
 * v^T [S,H,dp,npad] exactly as vog_qkv_proj does. */'''
new=''' * entry adds the two parts (one rounding to 16 bit) and emits q, k, v in the
 * fragment order of vog_qkv_proj. */'''
assert old in s; s=s.replace(old,new)
old=''' * (transformer_code.py:42-50). u: [n_vid, NP, H] fp32; token j of sequence s
 * uses row (s / seq_per_vid)*NP + (s % seq_per_vid)*n_box + (j % n_box).
 * out16: [S*N, H*dp] t16 (heads concatenated, padded). */'''
new=''' * (transformer_code.py:42-50). q, k, vt: fragment order of vog_qkv_proj (npad = N up
 * to 32). u: [n_vid, NP, H] fp32; token j of sequence s uses row
 * (s / seq_per_vid)*NP + (s % seq_per_vid)*n_box + (j % n_box).
 * out16: [S*N, H*dp] t16 row-major (heads concatenated, padded). */'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)
