// Phase timing of the one-key-block separable attention (attn_struct1_dma_kernel / lean) at the cfg-2 shape: wall-clock stamps
// (100 MHz) per wave. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include scratch/ts_attn6.hip -o scratch/ts_attn6
#define VOG_TS_ATTN 1
#include "../vognet-pytorch_amd/csrc/attention.hip"
#include <vector>
#include <stdlib.h>
#include <algorithm>
namespace vog { thread_local std::vector<LaunchRecord>* g_pair_capture = nullptr; void set_error(const char*, ...) {} }
int main() {
  const int S = 40, H = 3, dp = 256, nsrl = 5, nppf = 20, nfrm = 10, npad_kv = 32;
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
  a.S = S; a.H = H; a.dp = dp; a.nsrl = nsrl; a.nppf = nppf; a.npad_q = 128; a.npad_kv = npad_kv; a.nfrm = nfrm;
  a.lang_per_vid = 1; a.nc_v = 1; a.use_rel = 1; a.seq_per_vid = nfrm; a.NP = nfrm * nppf; a.inv_scale = 0.036f;
  a.dtype = VOG_BF16; a.q_visual = 1;
  for (int i = 0; i < 20; ++i) if (vog_rel_attention_struct_fwd(&a, st) != 0) { printf("launch failed\n"); return 1; }
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  hipEventRecord(e0, st);
  for (int i = 0; i < 200; ++i) vog_rel_attention_struct_fwd(&a, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("back-to-back launches: %.2f us per launch\n", ms * 1000 / 200);
  static unsigned long long ts[4096][8];
  hipMemcpyFromSymbol(ts, HIP_SYMBOL(vog::g_ats), sizeof(ts));
  const int nw = S * H * 4;
  unsigned long long t0 = ~0ull, t5 = 0, s_max = 0; double d[6] = {0, 0, 0, 0, 0, 0}; int live = 0;
  for (int w = 0; w < nw; ++w) {
    t0 = std::min(t0, ts[w][0]); s_max = std::max(s_max, ts[w][0]);
    if (ts[w][5] < ts[w][0]) continue;                 // (wave without a query block: stale end stamp)
    ++live; t5 = std::max(t5, ts[w][5]);
    for (int j = 1; j < 6; ++j) d[j] += (double)(ts[w][j] - ts[w][j - 1]);
  }
  printf("first wave start .. last wave start %.2f us; first start .. last end %.2f us (%d live waves)\n", (s_max - t0) / 100.0, (t5 - t0) / 100.0, live);
  printf("mean per wave: issue DMA + stage rows %.2f | wait + barrier %.2f | Q.K^T %.2f | softmax %.2f | P.V + stores %.2f us\n",
         d[1] / live / 100, (d[2]) / live / 100, d[3] / live / 100, d[4] / live / 100, d[5] / live / 100);
  return 0;
}
