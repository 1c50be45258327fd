p='include/vog_hip.h'
s=open(p).read()
old="/* u[v, r, h] = sum_c W_pe[h,c]"
new='''/* dst[i] = (t16) src[i] for two arrays in one launch (raw proposal / segment
 * features -> the encoders' MFMA operand type; replaces the implicit fp32 read
 * of nn.Linear in prop_feats_encode / seg_feats_encode mdl_vog.py:291-314).
 * n0, n1 multiples of 4; src1 may be NULL. */
int vog_cast_f32_to_t16(const float* src0, void* dst0, int64_t n0, const float* src1, void* dst1,
                        int64_t n1, vog_dtype dtype, void* stream);

/* u[v, r, h] = sum_c W_pe[h,c]'''
assert old in s
s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/elementwise.hip'
s=open(p).read()
old="// ---------------------------------------------------------------------------\n// u[row,h] = W_pe[h,:] . norm(box[row,:5])"
new='''// ---------------------------------------------------------------------------
// fp32 -> 16-bit cast of the raw feature blocks (HBM-bound: 16 B in, 8 B out per lane)
// ---------------------------------------------------------------------------
template <typename T16>
__global__ __launch_bounds__(256) void cast2_kernel(const float4* __restrict__ s0, u16x4* __restrict__ d0,
                                                    int64_t n0, const float4* __restrict__ s1,
                                                    u16x4* __restrict__ d1, int64_t n1) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += stride) {
    const bool first = i < n0;
    const float4 v = first ? s0[i] : s1[i - n0];
    u16x4 o = {to16<T16>(v.x), to16<T16>(v.y), to16<T16>(v.z), to16<T16>(v.w)};
    if (first) d0[i] = o; else d1[i - n0] = o;
  }
}

'''+old
assert old in s
s=s.replace(old,new,1)
s=s.replace('''extern "C" int vog_box_u(''','''extern "C" int vog_cast_f32_to_t16(const float* src0, void* dst0, int64_t n0, const float* src1,
                                   void* dst1, int64_t n1, vog_dtype dtype, void* stream) {
  VOG_CHECK_ARG(src0 && dst0 && n0 > 0 && (n0 % 4) == 0 && n1 >= 0 && (n1 % 4) == 0 && (n1 == 0 || (src1 && dst1)));
  const int64_t q = (n0 + n1) / 4;
  const int grid = (int)((q + 255) / 256 < 2048 ? (q + 255) / 256 : 2048);
  VOG_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((cast2_kernel<T16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)src0, (u16x4*)dst0, n0 / 4, (const float4*)src1, (u16x4*)dst1, n1 / 4));
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_box_u(''',1)
open(p,'w').write(s)

p='vognet-pytorch_amd/lib.py'
s=open(p).read()
s=s.replace('''    "vog_box_u":''','''    "vog_cast_f32_to_t16": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "vog_box_u":''')
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
old='''  p.add("prop_seg", g.rows_obj * g.d_obj * 4);'''
new='''  p.add("prop16", g.rows_obj * d.prop_dim * 2);
  p.add("seg16", (int64_t)g.n_vid * g.Fv * d.seg_dim * 2);
  p.add("prop_seg", g.rows_obj * g.d_obj * 4);'''
assert old in s
s=s.replace(old,new)
old='''    vog_gemm_args pe{}; pe.c16_dtype = d.tx_dtype;
    pe.a = b->pad_region_feature; pe.a_is_f32 = 1; pe.lda = d.prop_dim;'''
new='''    // raw features -> encoder operand type once (then both encoders run on the
    // LDS-DMA GEMM, which cannot convert in flight)
    {
      const float *s0 = b->pad_region_feature, *s1 = b->seg_feature_for_frms;
      void *d0 = ws.at<void>("prop16"), *d1 = ws.at<void>("seg16");
      const int64_t n0 = g.rows_obj * d.prop_dim, n1 = (int64_t)g.n_vid * g.Fv * d.seg_dim;
      steps.push_back({"cast_feats", [=](hipStream_t st) {
        return vog_cast_f32_to_t16(s0, d0, n0, s1, d1, n1, et, st); }});
    }
    vog_gemm_args pe{}; pe.c16_dtype = d.tx_dtype;
    pe.a = ws.at<void>("prop16"); pe.a_is_f32 = 0; pe.lda = d.prop_dim;'''
assert old in s
s=s.replace(old,new)
old='''    se.a = b->seg_feature_for_frms; se.a_is_f32 = 1; se.lda = d.seg_dim;'''
new='''    se.a = ws.at<void>("seg16"); se.a_is_f32 = 0; se.lda = d.seg_dim;'''
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
