// Phase timing of the persistent BiLSTM layer kernel (wall-clock stamps, workgroup 3 of each direction).
#define VOG_TS_DEBUG 1
#include "../vognet-pytorch_amd/csrc/lstm.hip"
#include <vector>
#include <stdlib.h>
namespace vog { thread_local LaunchRecorder* g_recorder = nullptr; void set_error(const char*, ...) {} }
int main() {
  const int Bn = 4, T = 12, R = 1024;
  float* gx; unsigned short *whh, *out16; int64_t* lens; void* hx; unsigned int* sync;
  hipMalloc(&gx, (size_t)2 * T * Bn * 4 * R * 4); hipMemset(gx, 0, (size_t)2 * T * Bn * 4 * R * 4);
  hipMalloc(&whh, (size_t)2 * 4 * R * R * 2); hipMemset(whh, 0, (size_t)2 * 4 * R * R * 2);
  hipMalloc(&out16, (size_t)(Bn * T + 64) * 2 * R * 2);
  hipMalloc(&hx, (size_t)2 * 2 * 16 * (R / 2) * 8); hipMalloc(&sync, 1024);
  std::vector<int64_t> hl(Bn, T); if (getenv("RAGGED")) { hl[1] = 7; hl[2] = 3; hl[3] = 9; } hipMalloc(&lens, Bn * 8); hipMemcpy(lens, hl.data(), Bn * 8, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemsetAsync(hx, 0, (size_t)2 * 2 * 16 * (R / 2) * 8, st); hipMemsetAsync(sync, 0, 1024, st);
    vog_lstm_layer_args a{}; a.gxs = gx; a.whh = whh; a.hx = hx; a.sync = sync; a.out16 = out16; a.lens = lens;
    a.Bn = Bn; a.T = T; a.R = R; a.dtype = VOG_F16; a.out_frag = getenv("FRAG") ? 1 : 0;
    if (vog_bilstm_layer(&a, st) != 0) { printf("launch failed\n"); return 1; }
  }
  hipStreamSynchronize(st);
  static unsigned long long ts[64][2048][4];
  hipMemcpyFromSymbol(ts, HIP_SYMBOL(vog::g_ts), sizeof(ts));
  for (int s = 0; s < T - 1; ++s) {
    unsigned long long* a = ts[s][0]; unsigned long long* b = ts[s][2]; unsigned long long* n = ts[s + 1][0];
    printf("step %2d dir0: gx + tagged h fetch + LDS + MFMA %5.2f | gates+publish %5.2f | block sync %5.2f | to next step %5.2f  (total %5.2f us)\n",
           s, (a[1] - a[0]) / 100.0, (a[2] - a[1]) / 100.0, (b[1] - a[2]) / 100.0, (n[0] - b[1]) / 100.0, (n[0] - a[0]) / 100.0);
  }
  return 0;
}
