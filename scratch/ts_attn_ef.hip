// Phase timing of the E x F separable attention kernel at the cfg-4 shape (wall-clock stamps per wave, 100 MHz).
#define VOG_TS_ATTN 1
#include "../vognet-pytorch_amd/csrc/attention.hip"
#include <vector>
#include <stdlib.h>
#include <algorithm>
namespace vog { thread_local LaunchRecorder* g_recorder = nullptr; thread_local std::vector<LaunchRecord>* g_pair_capture = nullptr; void set_error(const char*, ...) {} }
int main() {
  const int S = 40, H = 3, dp = 256, nsrl = 5, nppf = 400, nfrm = 10, npad_kv = 416;
  const size_t kvn = (size_t)S * H * npad_kv * dp;
  unsigned short *q, *k, *v, *out; float *pl, *u, *peb;
  hipMalloc(&q, kvn * 2); hipMalloc(&k, kvn * 2); hipMalloc(&v, kvn * 2);
  hipMemset(q, 0, kvn * 2); hipMemset(k, 0, kvn * 2); hipMemset(v, 0, kvn * 2);
  hipMalloc(&out, (size_t)S * nsrl * nppf * H * dp * 2);
  hipMalloc(&pl, (size_t)4 * nsrl * 3 * H * dp * 4); hipMemset(pl, 0, (size_t)4 * nsrl * 3 * H * dp * 4);
  hipMalloc(&u, (size_t)4 * nfrm * nppf * H * 4); hipMemset(u, 0, (size_t)4 * nfrm * nppf * H * 4);
  hipMalloc(&peb, H * 4); hipMemset(peb, 0, H * 4);
  hipStream_t st; hipStreamCreate(&st);
  vog_attn_struct_args a{};
  a.q = q; a.kv = k; a.vv = v; a.pl = pl; a.out16 = out; a.u = u; a.pe_b = peb;
  a.S = S; a.H = H; a.dp = dp; a.nsrl = nsrl; a.nppf = nppf; a.npad_q = 2016; a.npad_kv = npad_kv; a.nfrm = nfrm;
  a.lang_per_vid = 0; a.nc_v = 1; a.use_rel = 1; a.seq_per_vid = nfrm; a.NP = nfrm * nppf; a.inv_scale = 0.036f;
  a.dtype = VOG_BF16; a.q_visual = 1;
  for (int i = 0; i < 5; ++i) if (vog_rel_attention_struct_fwd(&a, st) != 0) { printf("launch failed: %s\n", "x"); return 1; }
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  hipEventRecord(e0, st);
  for (int i = 0; i < 20; ++i) vog_rel_attention_struct_fwd(&a, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("back-to-back launches: %.2f us per launch\n", ms * 1000 / 20);
  static unsigned long long ts[4096][8];
  hipMemcpyFromSymbol(ts, HIP_SYMBOL(vog::g_ats), sizeof(ts));
  const int nw = 4096;
  unsigned long long t0 = ~0ull, t6 = 0; double d[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < nw; ++w) {
    t0 = std::min(t0, ts[w][0]); t6 = std::max(t6, ts[w][6]);
    for (int j = 1; j < 7; ++j) d[j] += (double)(ts[w][j] - ts[w][j - 1]);
  }
  printf("first start .. last end (of the stamped waves) %.2f us\n", (t6 - t0) / 100.0);
  printf("mean per wave: staging + Q loads %.2f | phase 1 (QK tiles) %.2f | barrier %.2f | phase 2 + barrier %.2f | phase 3 rounds %.2f | epilogue %.2f us\n",
         d[1] / nw / 100, d[2] / nw / 100, d[3] / nw / 100, d[4] / nw / 100, d[5] / nw / 100, d[6] / nw / 100);
  for (int w = 0; w < 8; ++w) { printf("wave %d:", w); for (int j = 0; j < 7; ++j) printf(" %.2f", (ts[w][j] - ts[w][0]) / 100.0); printf("\n"); }
  return 0;
}
