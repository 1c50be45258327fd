// Phase timing of the persistent BiLSTM layer kernel (shader-clock stamps of wave 0 of every workgroup).
//   FUSED=K: input projection in the kernel with K input columns (512 = layer 0, 2048 = layer 1)
#define VOG_TS_DEBUG 1
#include "../vognet-pytorch_amd/csrc/lstm.hip"
#include <vector>
#include <stdlib.h>
namespace vog { thread_local LaunchRecorder* g_recorder = nullptr; thread_local std::vector<LaunchRecord>* g_pair_capture = nullptr; void set_error(const char*, ...) {} }
int main() {
  const int Bn = 4, T = 12, R = 1024;
  const int K = getenv("FUSED") ? atoi(getenv("FUSED")) : 0;
  float* gx; unsigned short *whh, *out16, *wih = nullptr, *xa = nullptr; float* bias = nullptr; int64_t* lens; void* hx; unsigned int* sync;
  hipMalloc(&gx, (size_t)2 * T * Bn * 4 * R * 4); hipMemset(gx, 0, (size_t)2 * T * Bn * 4 * R * 4);
  hipMalloc(&whh, (size_t)2 * 4 * R * R * 2); hipMemset(whh, 0, (size_t)2 * 4 * R * R * 2);
  hipMalloc(&out16, (size_t)(Bn * T + 64) * 2 * R * 2);
  if (K) {
    hipMalloc(&wih, (size_t)2 * 4 * R * K * 2); hipMemset(wih, 0, (size_t)2 * 4 * R * K * 2);
    hipMalloc(&xa, (size_t)64 * K * 2); hipMemset(xa, 0, (size_t)64 * K * 2);
    hipMalloc(&bias, (size_t)8 * R * 4); hipMemset(bias, 0, (size_t)8 * R * 4);
  }
  const size_t hxb = (size_t)vog_bilstm_hx_bytes(Bn, T, R); hipMalloc(&hx, hxb); hipMalloc(&sync, 1024);
  std::vector<int64_t> hl(Bn, T); if (getenv("RAGGED")) { hl[1] = 7; hl[2] = 3; hl[3] = 9; } hipMalloc(&lens, Bn * 8); hipMemcpy(lens, hl.data(), Bn * 8, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemsetAsync(hx, 0xff, hxb, st); hipMemsetAsync(sync, 0, 1024, st);
    vog_lstm_layer_args a{}; a.gxs = gx; a.whh = whh; a.hx = hx; a.sync = sync; a.out16 = out16; a.lens = lens;
    a.Bn = Bn; a.T = T; a.R = R; a.dtype = VOG_F16; a.out_frag = 1;
    if (K) { a.wih = wih; a.xa = xa; a.bias = bias; a.K = K; }
    hipEventRecord(e0, st);
    if (vog_bilstm_layer(&a, st) != 0) { printf("launch failed: %s\n", "x"); return 1; }
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    hipEventElapsedTime(&ms, e0, e1);
  }
  printf("kernel %.2f us (event)\n", ms * 1e3);
  static unsigned long long ts[64][24][8];
  hipMemcpyFromSymbol(ts, HIP_SYMBOL(vog::g_tsl), sizeof(ts));
  // shader clock: calibrate against the whole kernel of wg 0
  unsigned long long t_first = ~0ull, t_last = 0;
  for (int w = 0; w < 64; ++w) { if (ts[w][T][0] < t_first) t_first = ts[w][T][0]; if (ts[w][T - 1][4] > t_last) t_last = ts[w][T - 1][4]; }
  const double cyc_per_us = 100.0;   // s_memrealtime: 100 MHz, one counter for the chip
  printf("stamps span %.2f us\n", (t_last - t_first) / cyc_per_us);
    const double f = 1.0 / 100.0;   // s_memtime ticks at 100 MHz on gfx950? printed both ways below
  for (int w : {0, 17, 32, 63}) {
    auto* t = ts[w];
    printf("wg %2d: start +%6.2f | prologue %6.2f | W_hh load -> step0 %6.2f  [x %.0f ticks/us]\n", w, (t[T][0] - t_first) / cyc_per_us,
           (t[T][1] - t[T][0]) / cyc_per_us, (t[0][0] - t[T][1]) / cyc_per_us, cyc_per_us);
  }
  double acc[5] = {0, 0, 0, 0, 0};
  for (int s = 1; s < T - 1; ++s) {
    double ph[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 64; ++w) {
      auto* a = ts[w][s]; auto* n = ts[w][s + 1];
      ph[0] += (a[1] - a[0]) / cyc_per_us / 64; ph[1] += (a[2] - a[1]) / cyc_per_us / 64; ph[2] += (a[3] - a[2]) / cyc_per_us / 64;
      ph[3] += (a[4] - a[3]) / cyc_per_us / 64; ph[4] += (n[0] - a[4]) / cyc_per_us / 64;
    }
    if (getenv("VERBOSE")) printf("step %2d: fetch(+retries) %5.2f | barrier %5.2f | lds+mfma %5.2f | gates+publish %5.2f | barrier->next %5.2f | total %5.2f us\n",
           s, ph[0], ph[1], ph[2], ph[3], ph[4], ph[0] + ph[1] + ph[2] + ph[3] + ph[4]);
    for (int i = 0; i < 5; ++i) acc[i] += ph[i] / (T - 2);
  }
  printf("mean   : fetch(+retries) %5.2f | barrier %5.2f | lds+mfma %5.2f | gates+publish %5.2f | barrier->next %5.2f | total %5.2f us\n",
         acc[0], acc[1], acc[2], acc[3], acc[4], acc[0] + acc[1] + acc[2] + acc[3] + acc[4]);
  (void)f;
  return 0;
}
