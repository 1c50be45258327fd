// Where do the ~4.7 us of one dependent LSTM step go? Per-wave wall-clock stamps (100 MHz).
#define VOG_TS_DEBUG 1
#include "../vognet-pytorch_amd/csrc/lstm.hip"
#include <algorithm>
#include <vector>
namespace vog { thread_local LaunchRecorder* g_recorder = nullptr; void set_error(const char*, ...) {} }
int main() {
  const int Bn = 4, T = 12, R = 1024;
  float *gx, *c; unsigned short *whh, *hA, *hB, *out16; int64_t* lens;
  hipMalloc(&gx, (size_t)2 * T * Bn * 4 * R * 4); hipMemset(gx, 0, (size_t)2 * T * Bn * 4 * R * 4);
  hipMalloc(&c, 32 * 2 * R * 4); hipMemset(c, 0, 32 * 2 * R * 4);
  hipMalloc(&whh, (size_t)2 * 4 * R * R * 2); hipMemset(whh, 0, (size_t)2 * 4 * R * R * 2);
  hipMalloc(&hA, 32 * 2 * R * 2); hipMemset(hA, 0, 32 * 2 * R * 2);
  hipMalloc(&hB, 32 * 2 * R * 2); hipMemset(hB, 0, 32 * 2 * R * 2);
  hipMalloc(&out16, (size_t)(Bn * T + 64) * 2 * R * 2);
  std::vector<int64_t> hl(Bn, T); hipMalloc(&lens, Bn * 8); hipMemcpy(lens, hl.data(), Bn * 8, hipMemcpyHostToDevice);
  hipStream_t st; hipStreamCreate(&st);
  for (int rep = 0; rep < 3; ++rep)
    for (int s = 0; s < T; ++s) {
      vog_lstm_step_args a{}; a.gx = gx; a.whh = whh; a.h_in = (s & 1) ? hB : hA; a.h_out = (s & 1) ? hA : hB; a.c = c;
      a.out16 = out16; a.lens = lens; a.Bn = Bn; a.T = T; a.R = R; a.step = s; a.dtype = VOG_F16;
      if (vog_bilstm_step(&a, st) != 0) { printf("launch failed\n"); return 1; }
    }
  hipStreamSynchronize(st);
  static unsigned long long ts[64][2048][4];
  hipMemcpyFromSymbol(ts, HIP_SYMBOL(vog::g_ts), sizeof(ts));
  unsigned long long prev_end = 0;
  for (int s = 0; s < T; ++s) {
    unsigned long long s_min = ~0ull, s_max = 0, m_max = 0, e_min = ~0ull, e_max = 0; double life = 0, ld = 0, mm = 0;
    for (int w = 0; w < 2048; ++w) { ld += (double)(ts[s][w][3] - ts[s][w][0]); mm += (double)(ts[s][w][1] - ts[s][w][3]); }
    printf("   mean per wave: entry -> operands loaded %5.2f us, MFMA part %5.2f us\n", ld / 2048 / 100.0, mm / 2048 / 100.0);
    for (int w = 0; w < 2048; ++w) {
      s_min = std::min(s_min, ts[s][w][0]); s_max = std::max(s_max, ts[s][w][0]); m_max = std::max(m_max, ts[s][w][1]);
      if (w % 4 == 0) { e_min = std::min(e_min, ts[s][w][2]); e_max = std::max(e_max, ts[s][w][2]); life += (double)(ts[s][w][2] - ts[s][w][0]); }
    }
    printf("step %2d: gap since prev last-wave-end %6.2f us | first..last wave start %5.2f us | last matrix-part done +%5.2f us | "
           "last wave end +%5.2f us (mean wave-0 lifetime %5.2f us)\n", s, prev_end ? (s_min - prev_end) / 100.0 : 0.0,
           (s_max - s_min) / 100.0, (m_max - s_min) / 100.0, (e_max - s_min) / 100.0, life / 512 / 100.0);
    prev_end = e_max;
  }
  return 0;
}
