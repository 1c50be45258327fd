p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
# Step gets a branch tag
old='''struct Step {
  std::string name;
  std::function<int(hipStream_t)> fn;
};'''
new='''// branch 0 = visual + joint path (caller's stream), 1 = language path (captured as a
// parallel branch of the graph), -1 = join marker: everything after it needs both.
struct Step {
  std::string name;
  std::function<int(hipStream_t)> fn;
  int branch = 0;
};'''
assert old in s; s=s.replace(old,new)
# ctx: side stream + events for capture
old='''  bool finalized = false;
  // device weights'''
new='''  bool finalized = false;
  hipStream_t side = nullptr;           // language branch during graph capture
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // device weights'''
assert old in s; s=s.replace(old,new)
# mark language steps: wrap the language block
old='''  // ---- language path (a14-a16)
  {'''
new='''  // ---- language path (a14-a16): branch 1
  const size_t lang_begin = steps.size();
  const bool structured = has_mul(d) && (g.d_obj % 64) == 0 && (g.L % 32) == 0;
  {'''
assert old in s; s=s.replace(old,new)
old='''    steps.push_back({"argvec", [=](hipStream_t st) {
      return vog_srl_argvec(full, cap, im, wa, ba, lang, Bn, T, nsrl, L, st); }});
  }'''
new='''    steps.push_back({"argvec", [=](hipStream_t st) {
      return vog_srl_argvec(full, cap, im, wa, ba, lang, Bn, T, nsrl, L, st); }});
    if (structured) {
      // language half of mul_tx's layer-0 QKV: depends on `lang` only, so it rides on this branch
      const TxWeights& tw = c->mul;
      const TxLayer& L0 = tw.layers[0];
      const int ncol = 3 * tw.H * tw.dp;
      vog_gemm_args gl{}; gl.c16_dtype = -1;
      gl.a = lang; gl.a_is_f32 = 1; gl.lda = g.L; gl.w = L0.wqkv + g.d_obj; gl.ldw = tw.d;
      gl.c32 = ws.at<float>("mul_pl"); gl.ldc = ncol; gl.M = g.Bn * d.nsrl; gl.N = ncol; gl.K = g.L;
      gl.rep = 1; gl.dtype = (vog_dtype)d.tx_dtype;
      if (gl.M <= 64 && L0.wqkv_lang_f) { gl.w = L0.wqkv_lang_f; gl.ldw = g.L; gl.w_frag = 1; }
      steps.push_back({"mul_pl", [=](hipStream_t st) { return vog_gemm_bias_act(&gl, st); }});
    }
  }
  for (size_t i = lang_begin; i < steps.size(); ++i) steps[i].branch = 1;'''
assert old in s; s=s.replace(old,new)
# tx_steps: do not emit _pl (already on the language branch)
old=s[s.index('      vog_gemm_args gl{}; gl.c16_dtype = -1;\n      gl.a = sv.lang;'):s.index('      vog_qkvcomb_args ca{};')]
s=s.replace(old,'')
s=s.replace("      ca.pv = gv.c32; ca.pl = gl.c32;","      ca.pv = gv.c32; ca.pl = ws.at<float>(n + \"_pl\");")
# join marker before first consumer of lang
old='''  // mul_tx consumes the token structure directly (layer-0 QKV and its residual), so the
  // token matrix is only materialised for ImgGrnd / VidGrnd, whose lin2 reads it
  const bool structured = has_mul(d) && (g.d_obj % 64) == 0 && (g.L % 32) == 0;
  if (!structured)'''
new='''  // mul_tx consumes the token structure directly (layer-0 QKV and its residual), so the
  // token matrix is only materialised for ImgGrnd / VidGrnd, whose lin2 reads it
  { Step j; j.name = "join"; j.branch = -1; steps.push_back(j); }
  if (!structured)'''
assert old in s; s=s.replace(old,new)
# runners
old='''extern "C" int vog_forward(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes, void* stream) {
  VOG_CHECK_ARG(c && b && ws);
  Plan plan;
  std::vector<Step> steps;
  VOG_TRY(build_steps(c, b, ws, ws_bytes, plan, steps));
  for (auto& s : steps) {
    const int r = s.fn((hipStream_t)stream);
    if (r != 0) return r;
  }
  return 0;
}'''
new='''extern "C" int vog_forward(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes, void* stream) {
  VOG_CHECK_ARG(c && b && ws);
  Plan plan;
  std::vector<Step> steps;
  VOG_TRY(build_steps(c, b, ws, ws_bytes, plan, steps));
  for (auto& s : steps) {               // eager: one stream, program order (re-entrant)
    if (s.branch < 0) continue;
    const int r = s.fn((hipStream_t)stream);
    if (r != 0) return r;
  }
  return 0;
}'''
assert old in s; s=s.replace(old,new)
old='''  hipStream_t st = (hipStream_t)stream;
  VOG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (auto& s : steps) {
    rc = s.fn(st);
    if (rc != 0) break;
  }
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(st, &g);'''
new='''  hipStream_t st = (hipStream_t)stream;
  if (!c->side) {
    VOG_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    VOG_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    VOG_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  // The graph is a DAG: the language chain (prep -> 2 x (input GEMM + T steps) -> projections)
  // is captured on a side stream forked from `st` and joined before the first kernel that
  // needs the argument vectors, so it runs beside the encoders + obj_tx instead of ahead of them.
  VOG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  hipError_t fe = hipEventRecord(c->ev_fork, st);
  if (fe == hipSuccess) fe = hipStreamWaitEvent(c->side, c->ev_fork, 0);
  bool joined = false;
  if (fe != hipSuccess) rc = -(int)fe - 1000;
  for (auto& s : steps) {
    if (rc != 0) break;
    if (s.branch < 0) {
      if (!joined) {
        fe = hipEventRecord(c->ev_join, c->side);
        if (fe == hipSuccess) fe = hipStreamWaitEvent(st, c->ev_join, 0);
        if (fe != hipSuccess) rc = -(int)fe - 1000;
        joined = true;
      }
      continue;
    }
    rc = s.fn(s.branch == 1 && !joined ? c->side : st);
  }
  if (rc == 0 && !joined) {             // no join marker (cannot happen today): join at the end
    fe = hipEventRecord(c->ev_join, c->side);
    if (fe == hipSuccess) fe = hipStreamWaitEvent(st, c->ev_join, 0);
    if (fe != hipSuccess) rc = -(int)fe - 1000;
  }
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(st, &g);'''
assert old in s; s=s.replace(old,new)
old='''extern "C" int vog_ctx_destroy(vog_ctx* c) {
  if (!c) return 0;'''
new='''extern "C" int vog_ctx_destroy(vog_ctx* c) {
  if (!c) return 0;
  if (c->side) { (void)hipStreamDestroy(c->side); (void)hipEventDestroy(c->ev_fork); (void)hipEventDestroy(c->ev_join); }'''
assert old in s; s=s.replace(old,new)
# time_kernel: skip join marker (fn empty)
s=s.replace("  for (auto& x : steps) if (x.name == kernel) { s = &x; break; }","  for (auto& x : steps) if (x.name == kernel && x.branch >= 0) { s = &x; break; }")
open(p,'w').write(s)
