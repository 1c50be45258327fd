// Per-CU ingest rate vs memory-level parallelism: one workgroup per CU streams 16-byte-per-lane loads (1 KiB per wave
// instruction) with U independent loads in flight per wave, from (a) a 2 MB region every workgroup shares (L2 hits),
// (b) a 512 KB region per workgroup, 128 MB in total, read repeatedly (Infinity Cache), (c) 32 MB per workgroup read once (HBM).
// hipcc --offload-arch=gfx950 -O3 -o cu_ingest cu_ingest.hip && ./cu_ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ base, size_t wg_stride16, size_t region16, int passes,
                                                      unsigned int* sink) {
  const u32x4* p = base + (size_t)blockIdx.x * wg_stride16;
  const int nthr = blockDim.x;
  u32x4 acc = {0, 0, 0, 0};
  for (int ps = 0; ps < passes; ++ps) {
    for (size_t i = threadIdx.x; i + (size_t)(U - 1) * nthr < region16; i += (size_t)U * nthr) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[i + (size_t)u * nthr];
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if (acc[0] == 0x12345678u && acc[1] == 77u) sink[0] = acc[2] ^ acc[3];
}

template <int U>
static float run(const u32x4* buf, size_t wg_stride16, size_t region16, int passes, int nwg, int threads, unsigned int* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(stream_kernel<U>, dim3(nwg), dim3(threads), 0, 0, buf, wg_stride16, region16, 1, sink);   // warm
  hipEventRecord(e0);
  hipLaunchKernelGGL(stream_kernel<U>, dim3(nwg), dim3(threads), 0, 0, buf, wg_stride16, region16, passes, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const size_t total = (size_t)8 << 30;                        // 8 GB
  u32x4* buf; unsigned int* sink;
  CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 1, total));
  const int nwg = 256;
  struct Mode { const char* name; size_t stride, region; int passes; } modes[] = {
    {"L2   (2 MB shared by all)", 0, (size_t)2 << 20, 64},
    {"MALL (512 KB per WG, 128 MB)", (size_t)512 << 10, (size_t)512 << 10, 64},
    {"HBM  (32 MB per WG, once)", (size_t)32 << 20, (size_t)32 << 20, 1},
  };
  printf("GB/s per CU (256 workgroups, one per CU); columns: loads in flight per wave\n");
  for (auto& m : modes) {
    printf("%s\n  waves   U=1     U=2     U=4     U=8     U=16\n", m.name);
    for (int waves : {4, 8, 16}) {
      printf("  %5d", waves);
      const int thr = waves * 64;
      const size_t s16 = m.stride / 16, r16 = m.region / 16;
      float ms[5] = {run<1>(buf, s16, r16, m.passes, nwg, thr, sink), run<2>(buf, s16, r16, m.passes, nwg, thr, sink),
                     run<4>(buf, s16, r16, m.passes, nwg, thr, sink), run<8>(buf, s16, r16, m.passes, nwg, thr, sink),
                     run<16>(buf, s16, r16, m.passes, nwg, thr, sink)};
      for (float t : ms) printf(" %7.1f", (double)m.region * m.passes / (t * 1e-3) / 1e9);
      printf("\n");
    }
  }
  // fewer CUs active: does a CU get more when the others are idle?
  printf("MALL mode, 8 waves, U=8, by number of workgroups (one per CU):\n");
  for (int n : {8, 32, 64, 128, 256}) {
    float t = run<8>(buf, ((size_t)512 << 10) / 16, ((size_t)512 << 10) / 16, 64, n, 512, sink);
    printf("  %3d WGs: %7.1f GB/s per CU, %8.1f GB/s total\n", n, (double)(512 << 10) * 64 / (t * 1e-3) / 1e9, (double)(512 << 10) * 64 * n / (t * 1e-3) / 1e9);
  }
  return 0;
}
