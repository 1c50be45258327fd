// MFMA issue rate and MFMA/VALU overlap inside one wave on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 0: 4 independent MFMA chains; 1: one dependent chain; 2: MFMA + NV independent VALU (fma) between; 3: VALU only
template <int MODE, int NV>
__global__ void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)(float)(lane - j); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = lane * 0.001f + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0 || MODE == 2) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
          for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], 1.0001f, 0.5f);
          __builtin_amdgcn_sched_barrier(0);
        }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
          for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], 1.0001f, 0.5f);
          __builtin_amdgcn_sched_barrier(0);
        }
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
          for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], 1.0001f, 0.5f);
          __builtin_amdgcn_sched_barrier(0);
        }
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
          for (int j = 0; j < NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], 1.0001f, 0.5f);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (MODE == 1) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 4 * NV; ++j) v[j % 8] = __builtin_fmaf(v[j % 8], 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
  for (int j = 0; j < 8; ++j) s += v[j];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int NV>
void run(int waves, const char* name) {
  float* out; hipMalloc(&out, 4096 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NV><<<256, waves * 64>>>(out, 10);
  hipEventRecord(e0);
  k<MODE, NV><<<256, waves * 64>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double slots = (double)iters * 16;           // MFMAs (or VALU groups) per wave
  const double wps = waves / 4.0;                    // waves per SIMD
  printf("%-34s waves/CU=%d: %8.1f us  %.1f cycles per MFMA-slot per SIMD (2.4 GHz)%s\n", name, waves, ms * 1000,
         ms * 1e-3 * 2.4e9 / (slots * (wps < 1 ? 1 : wps)),
         MODE != 3 ? "" : " [VALU only]");
  if (MODE != 3) printf("      -> %.0f TFLOP/s whole chip\n", slots * waves * 256 * 32768.0 / (ms * 1e-3) / 1e12);
}
int main() {
  run<0, 0>(4, "4 independent chains");
  run<0, 0>(8, "4 independent chains");
  run<1, 0>(4, "1 dependent chain");
  run<1, 0>(8, "1 dependent chain");
  run<3, 6>(4, "VALU only, 6 fma per slot");
  run<2, 2>(4, "MFMA + 2 fma");
  run<2, 4>(4, "MFMA + 4 fma");
  run<2, 6>(4, "MFMA + 6 fma");
  run<2, 8>(4, "MFMA + 8 fma");
  run<2, 12>(4, "MFMA + 12 fma");
  run<2, 6>(8, "MFMA + 6 fma");
  run<2, 12>(8, "MFMA + 12 fma");
  return 0;
}
