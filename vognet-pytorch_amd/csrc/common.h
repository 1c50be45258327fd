// Shared device/host helpers for libvog_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/vog_hip.h"

namespace vog {

// ---- error plumbing --------------------------------------------------------
void set_error(const char* fmt, ...);
#define VOG_FAIL(code, ...) do { ::vog::set_error(__VA_ARGS__); return (code); } while (0)
#define VOG_CHECK_ARG(cond) do { if (!(cond)) VOG_FAIL(-1, "%s:%d: bad argument: %s", __FILE__, __LINE__, #cond); } while (0)
#define VOG_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) VOG_FAIL(-(int)e_ - 1000, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); } while (0)
#define VOG_LAUNCH_CHECK() VOG_HIP(hipGetLastError())
#define VOG_TRY(expr) do { int r_ = (expr); if (r_ != 0) return r_; } while (0)

// ---- vector types ------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct BF16 {};   // tags for the 16-bit storage/MFMA type
struct F16 {};

// fp32 -> 16-bit (round to nearest even) and back; raw bit patterns in memory.
template <typename T> __device__ __forceinline__ unsigned short to16(float f);
template <> __device__ __forceinline__ unsigned short to16<BF16>(float f) {
  // gfx950 has a hardware RNE convert (v_cvt_pk_bf16_f32); the cast lowers to it.
  // (A bit-twiddled RNE costs ~6 VALU per element: measured 4.2k VALU per attention wave.)
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
template <> __device__ __forceinline__ unsigned short to16<F16>(float f) {
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(unsigned short, h);
}
template <typename T> __device__ __forceinline__ float from16(unsigned short v);
template <> __device__ __forceinline__ float from16<BF16>(unsigned short v) {
  return __uint_as_float(((unsigned int)v) << 16);
}
template <> __device__ __forceinline__ float from16<F16>(unsigned short v) {
  return (float)__builtin_bit_cast(_Float16, v);
}

// ---- LDS reads the compiler does not see -------------------------------------------
// hipcc's waitcnt pass cannot tell which LDS bytes an in-flight LDS-DMA (global_load_lds) will write,
// so it puts s_waitcnt vmcnt(0) in front of EVERY LDS load that follows one: a prefetch ring fed by
// LDS-DMA degenerates into issue -> wait -> compute (measured: one full L2 round trip per key block /
// K tile). Kernels that keep DMA in flight across LDS reads therefore read through these wrappers
// (the asm hides the address space) and do the bookkeeping themselves: lds_wait<N>() before the first
// use of a fragment (in/out operands tie the consumer to the wait).
__device__ __forceinline__ u16x8 lds_read128(const void* p) {
  u16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)p));
  return v;
}
template <int OFF>
__device__ __forceinline__ u16x8 lds_read128(const void* p) {
  u16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(uintptr_t)p), "n"(OFF));
  return v;
}
__device__ __forceinline__ f32x4 lds_read_f4(const void* p) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)p));
  return v;
}
template <int N, typename A>
__device__ __forceinline__ void lds_wait(A& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N, typename A, typename B>
__device__ __forceinline__ void lds_wait(A& a, B& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }

// Workgroup barrier WITHOUT the release / acquire fences of __syncthreads(): hipcc lowers those to
// s_waitcnt vmcnt(0) lgkmcnt(0), i.e. every barrier also waits for ALL outstanding global loads (a prefetch
// issued before the barrier is drained at it) and for the write acknowledge of every store issued so far.
// The caller's own LDS writes are waited for here; anything else the barrier has to cover is the caller's.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// ---- MFMA wrappers (fp32 accumulate) ----------------------------------------------
// 32x32x16: A lane l holds row (l&31), k = (l>>5)*8+j; B lane l holds col (l&31),
// same k; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
template <typename T> __device__ __forceinline__ f32x16 mfma32(u16x8 a, u16x8 b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma32<BF16>(u16x8 a, u16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma32<F16>(u16x8 a, u16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// 16x16x32: A lane l holds row (l&15), k = (l>>4)*8+j; B lane l holds col (l&15);
// C/D: col = l&15, row = (l>>4)*4 + reg.
template <typename T> __device__ __forceinline__ f32x4 mfma16(u16x8 a, u16x8 b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mfma16<BF16>(u16x8 a, u16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                 __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mfma16<F16>(u16x8 a, u16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a),
                                                __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ReLU that PROPAGATES NaN (fmaxf(NaN, 0) is 0): the persistent BiLSTM poisons its output with NaN
// when a hand-off times out, and that poison has to reach mdl_outs through every projection.
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ int c32_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

// ---- fragment-ordered attention operands ---------------------------------------------
// q, k and v of one (sequence, head) are stored in the order the attention kernel's MFMA
// operands consume them, npad = N rounded up to 32 tokens, npad*dp halfwords each:
//   q/k : [token/32][dd/16][lane = ((dd/8)&1)*32 + token%32][dd%8]
//         (= A/B fragment of v_mfma_f32_32x32x16: row token%32, k = dd%16)
//   v   : [token/32][dd/32][ks = (token%32)/16][lane = hi*32 + dd%32][j]
//         with token%16 = 8*(j>>2) + 4*hi + (j&3)  (the key permutation under which the
//         S^T accumulator registers are directly the P^T operand, attention.hip)
// Every fragment is one contiguous KiB: wave loads and LDS-DMA need no swizzle.
__host__ __device__ __forceinline__ int64_t frag_qk(int i, int dd, int dp) {
  return ((int64_t)(i >> 5) * (dp >> 4) + (dd >> 4)) * 512 + ((((dd >> 3) & 1) << 5) + (i & 31)) * 8 + (dd & 7);
}
__host__ __device__ __forceinline__ int64_t frag_v(int i, int dd, int dp) {
  const int kl = i & 31, r = kl & 15;
  const int j = ((r >> 3) << 2) + (r & 3), hi = (r >> 2) & 1;
  return ((((int64_t)(i >> 5) * (dp >> 5) + (dd >> 5)) * 2 + (kl >> 4)) * 64 + (hi << 5) + (dd & 31)) * 8 + j;
}

// A operand of the M <= 64 GEMM in fragment order (K = number of columns)
__host__ __device__ __forceinline__ int64_t frag_a(int m, int k, int K) {
  return ((((int64_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64) + (((k >> 3) & 3) << 4) + (m & 15)) * 8 + (k & 7);
}

// ---- kernel bodies ---------------------------------------------------------------------------
// The kernels that can share a launch with another one are written as BODIES: a struct with
// Params, THREADS and a static __device__ run(params, ctx, smem) that takes its block index and
// grid size from `ctx` and all of its LDS from `smem`. Their own __global__ kernel calls the body
// with the hardware indices; pair_kernel (pair.hip) calls two different bodies from one grid.
struct BlockCtx { unsigned bx, by, gx, gy; };

// Cooperative staging with B loads in flight per thread. A "load, wait, store to LDS" loop with a run-time trip count pays one
// memory round trip per iteration (the compiler does not move the next load over the LDS store); here a batch of B loads is issued
// with clamped indices before the first store of the batch (round 6: the same change took argvec from 28 to 8.8 us).
template <int B, int NT, typename T, typename Load, typename Store>
__device__ __forceinline__ void stage_batched(int n, int tid, Load load, Store store) {
  for (int i0 = tid; i0 < n; i0 += B * NT) {
    T v[B];
#pragma unroll
    for (int q = 0; q < B; ++q) { const int i = i0 + q * NT; v[q] = load(i < n ? i : n - 1); }
#pragma unroll
    for (int q = 0; q < B; ++q) { const int i = i0 + q * NT; if (i < n) store(i, v[q]); }
  }
}

// ---- kernel launch ---------------------------------------------------------------------------
// Every kernel of the library is launched through vog::launch: hipLaunchKernelGGL on the caller's stream, or - while pair.hip
// has its capture open - a record of the kernel's host stub, geometry and packed kernarg bytes, from which two independent steps
// of the forward are issued as ONE grid (horizontal fusion). (Round 2-5 also recorded whole forwards into raw AQL packets on the
// library's own HSA queues; measured no faster than hipGraph replay - 54.9 vs 57.1 k queries/s - and removed in round 6:
// scratch/negatives/r6_pruned/.)
struct LaunchRecord {
  const void* host_fn;
  unsigned grid[3], block[3];      // grid in workgroups
  unsigned dyn_lds;
  unsigned arg_bytes;              // explicit kernarg bytes (natural alignment packing)
  unsigned char args[1024];
};
extern thread_local std::vector<LaunchRecord>* g_pair_capture;

template <typename T>
inline void pack_arg(LaunchRecord& r, const T& v) {
  unsigned off = (r.arg_bytes + alignof(T) - 1) / alignof(T) * alignof(T);
  static_assert(sizeof(T) <= sizeof(r.args), "kernel argument too large");
  if (off + sizeof(T) > sizeof(r.args)) { r.arg_bytes = 0xffffffffu; return; }
  memcpy(r.args + off, &v, sizeof(T));
  r.arg_bytes = off + (unsigned)sizeof(T);
}

template <typename... KArgs, typename... Args>
inline void launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "argument count mismatch");
  if (g_pair_capture) {
    LaunchRecord r;
    r.host_fn = reinterpret_cast<const void*>(kern);
    r.grid[0] = grid.x; r.grid[1] = grid.y; r.grid[2] = grid.z;
    r.block[0] = block.x; r.block[1] = block.y; r.block[2] = block.z;
    r.dyn_lds = (unsigned)lds; r.arg_bytes = 0;
    (pack_arg<KArgs>(r, static_cast<KArgs>(args)), ...);
    g_pair_capture->push_back(r);
    return;
  }
  hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<KArgs>(args)...);
}

// Perf-experiment knobs (tile forcing, ablations that produce WRONG results, fence scopes) are read
// from the environment only when VOG_PERF_EXPERIMENTS=1 is also set, so that a stray variable can
// never change what the product path computes.
static inline const char* perf_env(const char* name) {
  const char* on = getenv("VOG_PERF_EXPERIMENTS");
  if (!on || on[0] != '1') return nullptr;
  return getenv(name);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t round_up64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// dispatch a templated launcher on the runtime 16-bit type
#define VOG_DISPATCH_DTYPE(dt, ...)                          \
  do {                                                       \
    if ((dt) == VOG_BF16) { using T16 = ::vog::BF16; __VA_ARGS__; } \
    else if ((dt) == VOG_F16) { using T16 = ::vog::F16; __VA_ARGS__; } \
    else VOG_FAIL(-1, "unknown vog_dtype %d", (int)(dt));    \
  } while (0)

}  // namespace vog
