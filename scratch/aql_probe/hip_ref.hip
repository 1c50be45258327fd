// same kernels through the HIP runtime, for comparison with the raw AQL numbers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include "probe_kernels.hip"
int main() {
  float4 *in, *out; hipMalloc(&in, 64 << 20); hipMalloc(&out, 64 << 20);
  hipStream_t st; hipStreamCreate(&st);
  for (int per : {256, 1024, 4096}) {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(stream_kernel, dim3(512), dim3(256), 0, st, in, out, per);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    const int N = 400;
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(stream_kernel, dim3(512), dim3(256), 0, st, in, out, per);
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("HIP stream launches per=%d: %.2f us/kernel\n", per, us / N);
  }
  return 0;
}
