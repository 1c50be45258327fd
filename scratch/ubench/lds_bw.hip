// LDS read throughput per CU on gfx950: ds_read_b128 / ds_read_b64, lane-linear, W waves per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 lds_bw.hip -o lds_bw ; run: ./lds_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int MODE>
__global__ void k(unsigned* out, int iters, int stride) {
  extern __shared__ unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(sm)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  const unsigned base = (unsigned)(uintptr_t)sm + w * 8192 + lane * stride;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      u32x4 a, b, c, d;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                   : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(base));
      acc += a[0] ^ b[1] ^ c[2] ^ d[3];
    } else {
      u32x2 a, b, c, d;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)"
                   : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(base));
      acc += a[0] ^ b[1] ^ c[0] ^ d[1];
    }
  }
  if (acc == 0x12345) out[tid] = acc;
}

template <int MODE>
void run(int waves, int stride, const char* name) {
  unsigned* out; hipMalloc(&out, 4096 * 4);
  const int iters = 20000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, waves * 64, 65536>>>(out, 100, stride);
  hipEventRecord(e0);
  k<MODE><<<256, waves * 64, 65536>>>(out, iters, stride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)iters * 4 * waves * 64 * (MODE == 0 ? 16 : 8);
  printf("%s waves=%d stride=%d: %.1f us, %.1f B/clk/CU at 2.4 GHz, %.1f cycles per wave-instruction-slot\n", name, waves, stride,
         ms * 1000, bytes_per_cu / (ms * 1e-3) / 2.4e9, ms * 1e-3 * 2.4e9 / (iters * 4.0 * waves));
}
int main() {
  for (int w : {1, 2, 4, 8}) run<0>(w, 16, "b128 lane-linear");
  for (int w : {4, 8}) run<1>(w, 8, "b64 lane-linear");
  return 0;
}
