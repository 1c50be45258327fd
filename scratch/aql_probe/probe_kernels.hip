#include <hip/hip_runtime.h>
extern "C" __global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
// each block copies `per_block` float4 (strided by block) -> exec time knob; uses gridDim (hidden args)
extern "C" __global__ __launch_bounds__(256) void stream_kernel(const float4* in, float4* out, int per_block) {
  const size_t base = (size_t)blockIdx.x * per_block;
  for (int i = threadIdx.x; i < per_block; i += blockDim.x) {
    float4 v = in[base + i]; v.x += (float)gridDim.x; out[base + i] = v;
  }
}
// dependent chain check: out[i] = in[i] + 1
extern "C" __global__ void inc_kernel(const int* in, int* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1;
}

// minimal real work: one load + one store per thread, no hidden arguments, no dispatch-packet reads
extern "C" __global__ __launch_bounds__(256) void touch_kernel(const float* in, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  out[i] = in[i] + 1.0f;
}
// two dependent loads (pointer chase through a small table) + store
extern "C" __global__ __launch_bounds__(256) void chase_kernel(const int* idx, const float* in, float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  out[i] = in[idx[i]] + 1.0f;
}
