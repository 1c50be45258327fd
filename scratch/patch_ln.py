p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
old='''                     int npad, int spv, int n_box, float fdiv, std::vector<Step>& steps,
                     const float** out32, const void** out16) {'''
new='''                     int npad, int spv, int n_box, float fdiv, int last_dt, std::vector<Step>& steps,
                     const float** out32, const void** out16) {'''
assert old in s; s=s.replace(old,new)
old='''    steps.push_back({n + "_ln2", [=](hipStream_t st) {
      return vog_residual_layernorm(tmp, L.ln2g, L.ln2b, o32, o16, (int)rows, d_, dt, st); }});'''
new='''    // 16-bit copy of the LAST layer's output: typed for its consumer (none for obj_tx,
    // the f16 score head for mul_tx)
    const bool last = l == tw.n_layers - 1;
    void* o16w = (last && last_dt < 0) ? nullptr : o16;
    const vog_dtype odt = last && last_dt >= 0 ? (vog_dtype)last_dt : dt;
    steps.push_back({n + "_ln2", [=](hipStream_t st) {
      return vog_residual_layernorm(tmp, L.ln2g, L.ln2b, o32, o16w, (int)rows, d_, odt, st); }});'''
assert old in s; s=s.replace(old,new)
old='''             g.fdiv_obj, steps, &vis32, &vis16);'''
new='''             g.fdiv_obj, -1, steps, &vis32, &vis16);'''
assert old in s; s=s.replace(old,new)
old='''             (float)g.nfrm, steps, &x32, &x16);'''
new='''             (float)g.nfrm, d.enc_dtype, steps, &x32, &x16);'''
assert old in s; s=s.replace(old,new)
old='''  int head_dt = has_mul(d) ? d.tx_dtype : d.enc_dtype;   // dtype of the 16-bit copy feeding lin2'''
new='''  int head_dt = d.enc_dtype;   // the 16-bit copy feeding lin2 is always written in the head's type'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)
