// All-gather of tagged 8-byte granules among 32 workgroups (the BiLSTM hand-off, lstm_dev.h) as a
// function of WHERE the 32 workgroups sit and of the store / load flavour:
//   place 0: blocks 0..31 of a 256-block grid (4 per XCD, the layout of the layer kernel in round 2)
//   place 1: the 32 blocks with blockIdx % 8 == X (one XCD: its L2 is the exchange point)
//   store : sc1 (agent write-through, drops the line from L2) | plain (stays in the XCD's L2) | sc0 | sc0 sc1
//   load  : sc1 (bypasses L1, L2 served) | sc0 sc1
// Every word is checked (tag = step, payload = f(step, index)); a spin is bounded.
// build: hipcc --offload-arch=gfx950 -O3 xcd_allgather.hip -o xcd_allgather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;

template <int ST> __device__ __forceinline__ void store8(u64* p, u64 v) {
  if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  else if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int LD> __device__ __forceinline__ void load8x4(const u64* p0, const u64* p1, const u64* p2, const u64* p3,
                                                          u64& v0, u64& v1, u64& v2, u64& v3) {
  if (LD == 0)
    asm volatile("global_load_dwordx2 %0, %4, off sc1\n global_load_dwordx2 %1, %5, off sc1\n"
                 "global_load_dwordx2 %2, %6, off sc1\n global_load_dwordx2 %3, %7, off sc1\n s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
  else
    asm volatile("global_load_dwordx2 %0, %4, off sc0 sc1\n global_load_dwordx2 %1, %5, off sc0 sc1\n"
                 "global_load_dwordx2 %2, %6, off sc0 sc1\n global_load_dwordx2 %3, %7, off sc0 sc1\n s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}

__device__ __forceinline__ unsigned payload(unsigned s, unsigned it) { return s * 2654435761u + it * 40503u; }

struct Res { u64 t0, t1; unsigned bad, spins, xcc, dead; };

// NW words per sentence (= R/2), NS sentences; every participant owns NW/32 consecutive words of each sentence
template <int ST, int LD>
__global__ __launch_bounds__(512) void k(u64* hx, Res* res, unsigned* done, const uint4* bg, size_t bg_n, int place, int xsel,
                                         int steps, int warm, int NS, int NW, int load_bg, int fresh) {
  extern __shared__ unsigned lds[];
  const int tid = threadIdx.x;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15u;
  const int b = blockIdx.x;
  int pi = -1;
  if (place == 0) { if (b < 32) pi = b; }
  else { if ((b & 7) == xsel) pi = b >> 3; }
  if (pi < 0) {
    if (!load_bg) return;
    // background: stream a large buffer until the participants are done
    uint4 acc = {0, 0, 0, 0};
    size_t i = ((size_t)b * 512 + tid) % bg_n;
    for (int it = 0; it < 4000000; ++it) {
      uint4 v = bg[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
      i += 256 * 512; if (i >= bg_n) i -= bg_n;
      if ((it & 63) == 0 && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc.x == 0x12345 && acc.y == 77) lds[tid] = acc.z;
    return;
  }
  const int per = NW / 32;                       // words of a sentence owned by this participant
  const int items = NS * NW;
  const size_t par = (size_t)NS * NW;
  unsigned bad = 0, spins = 0; bool dead = false;
  u64 t0 = 0;
  for (int s = 0; s < steps && !dead; ++s) {
    if (s == warm && tid == 0) t0 = wall_clock64();
    // publish step s+1 tags into parity (s+1)&1 ... like the layer kernel: step s reads parity s&1 (tag s), writes (s+1)&1 (tag s+1)
    // gather tag s (step 0: zero-initialised buffer = tag 0 payload 0)
    for (int base = tid; base < items; base += 512 * 4) {
      const u64* src = hx + (size_t)(fresh ? s : (s & 1)) * par;
      u64 v[4];
      int it[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { it[j] = base + j * 512; if (it[j] >= items) it[j] = base; }
      load8x4<LD>(src + it[0], src + it[1], src + it[2], src + it[3], v[0], v[1], v[2], v[3]);
      unsigned sp = 0;
      for (;;) {
        bool stale = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) stale |= (unsigned)(v[j] >> 32) != (unsigned)s && !(fresh && s == 0);
        if (!stale) break;
        load8x4<LD>(src + it[0], src + it[1], src + it[2], src + it[3], v[0], v[1], v[2], v[3]);
        if (++sp > (1u << 18)) { dead = true; break; }
      }
      spins += sp;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (s > 0 && (unsigned)v[j] != payload(s, it[j])) ++bad;
        lds[it[j]] = (unsigned)v[j];
      }
    }
    dead = __syncthreads_or(dead ? 1 : 0) != 0;
    // "compute": touch LDS a little, then publish own words of step s+1
    if (tid < NS * per) {
      const int sen = tid / per, j = tid % per;
      const int itw = sen * NW + pi * per + j;
      const u64 val = ((u64)(unsigned)(s + 1) << 32) | payload(s + 1, itw);
      store8<ST>(hx + (size_t)(fresh ? s + 1 : ((s + 1) & 1)) * par + itw, val);
    }
    __syncthreads();
  }
  if (tid == 0) {
    res[pi].t0 = t0; res[pi].t1 = wall_clock64(); res[pi].xcc = xcc; res[pi].dead = dead;
  }
  atomicAdd(&res[pi].bad, bad);
  atomicAdd(&res[pi].spins, spins);
  if (tid == 0 && pi == 0) __hip_atomic_store(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int ST, int LD>
static void run(const char* name, int place, int xsel, int load_bg, int fresh, u64* hx, Res* res, unsigned* done, uint4* bg, size_t bg_n,
                int NS, int NW) {
  const int steps = 400, warm = 50;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(hx, 0, (size_t)402 * NS * NW * 8));
    CHECK(hipMemset(res, 0, sizeof(Res) * 32));
    CHECK(hipMemset(done, 0, 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<ST, LD>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipLaunchKernelGGL((k<ST, LD>), dim3(256), dim3(512), 100 * 1024, 0, hx, res, done, bg, bg_n, place, xsel, steps, warm, NS, NW, load_bg, fresh);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<Res> h(32);
  CHECK(hipMemcpy(h.data(), res, sizeof(Res) * 32, hipMemcpyDeviceToHost));
  double mn = 1e30, mx = 0, av = 0; unsigned bad = 0, dead = 0; unsigned long long spins = 0;
  unsigned xmask = 0;
  for (auto& r : h) {
    const double us = (double)(r.t1 - r.t0) / 100.0 / (steps - warm);
    mn = std::min(mn, us); mx = std::max(mx, us); av += us / 32; bad += r.bad; dead += r.dead; spins += r.spins; xmask |= 1u << r.xcc;
  }
  printf("%-34s fresh %d place %d bg %d NS %d NW %4d: %.3f us/step (min %.3f max %.3f)  bad %u dead %u  retries/step/WG %.1f  xcc mask 0x%02x\n",
         name, fresh, place, load_bg, NS, NW, av, mn, mx, bad, dead, (double)spins / 32 / steps, xmask);
}

int main() {
  u64* hx; Res* res; unsigned* done; uint4* bg;
  const size_t bg_n = (size_t)512 * 1024 * 1024 / 16;
  CHECK(hipMalloc(&hx, (size_t)402 * 16 * 512 * 8));
  CHECK(hipMalloc(&res, sizeof(Res) * 32));
  CHECK(hipMalloc(&done, 4));
  CHECK(hipMalloc(&bg, bg_n * 16));
  CHECK(hipMemset(bg, 1, bg_n * 16));
  for (int fresh = 0; fresh < 2; ++fresh) {
    run<0, 0>("store sc1      load sc1", 0, 0, 0, fresh, hx, res, done, bg, bg_n, 4, 512);
    run<0, 0>("store sc1      load sc1", 1, 3, 0, fresh, hx, res, done, bg, bg_n, 4, 512);
    run<1, 0>("store plain    load sc1", 1, 3, 0, fresh, hx, res, done, bg, bg_n, 4, 512);
  }
  return 0;
}
