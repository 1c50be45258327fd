p='vognet-pytorch_amd/csrc/gemm.hip'
s=open(p).read()
s=s.replace("  int M, N, K; int relu; int rep; int c16_bf16;\n","  int M, N, K; int relu; int rep; int c16_bf16; int debug;\n")
# read env once
s=s.replace("enum { EPI_PLAIN = 0, EPI_QKV = 1 };","enum { EPI_PLAIN = 0, EPI_QKV = 1 };\n\n// VOG_GEMM_DEBUG (ablation, perf experiments only): 1 = no DMA, 2 = no MFMA, 4 = no epilogue\nstatic int gemm_debug_flags() {\n  static int v = -1;\n  if (v < 0) { const char* e = getenv(\"VOG_GEMM_DEBUG\"); v = e ? atoi(e) : 0; }\n  return v;\n}")
s=s.replace('#include "common.h"\n\nnamespace vog {\n\nenum { EPI_PLAIN','#include <stdlib.h>\n#include "common.h"\n\nnamespace vog {\n\nenum { EPI_PLAIN',1)
# in pipe kernel: guard issue / mfma / epilogue
s=s.replace("  auto issue = [&](int kt, int stage) {\n#pragma unroll\n    for (int i = 0; i < LPT; ++i) {\n      __builtin_amdgcn_global_load_lds(","  auto issue = [&](int kt, int stage) {\n    if (p.debug & 1) return;\n#pragma unroll\n    for (int i = 0; i < LPT; ++i) {\n      __builtin_amdgcn_global_load_lds(")
s=s.replace("    const unsigned char* st = smem + (kt % STAGES) * STAGE_BYTES;\n#pragma unroll\n    for (int ks = 0; ks < 4; ++ks) {\n      u16x8 fa[FM], fb[FN];\n      const int g = ks * 2 + hi;","    const unsigned char* st = smem + (kt % STAGES) * STAGE_BYTES;\n    if (p.debug & 2) continue;\n#pragma unroll\n    for (int ks = 0; ks < 4; ++ks) {\n      u16x8 fa[FM], fb[FN];\n      const int g = ks * 2 + hi;")
s=s.replace("  // Swapped operands => each lane owns ONE output row m","  if (p.debug & 4) { if (acc[0][0][0] != 123.456f) return; }\n  // Swapped operands => each lane owns ONE output row m")
s=s.replace("  p.c16_bf16 = (g->c16_dtype < 0 ? (int)g->dtype : g->c16_dtype) == VOG_BF16;","  p.c16_bf16 = (g->c16_dtype < 0 ? (int)g->dtype : g->c16_dtype) == VOG_BF16;\n  p.debug = gemm_debug_flags();")
s=s.replace("  p.ntok = a->N; p.H = a->H; p.dp = a->dp; p.npad = a->npad;","  p.ntok = a->N; p.H = a->H; p.dp = a->dp; p.npad = a->npad;\n  p.debug = gemm_debug_flags();")
open(p,'w').write(s)
