// Round-3 hand-off protocol in isolation (sentinel-validated 16-bit values, one slot per step), spread
// over the XCDs (blocks 0..31 of a 256-block grid) or inside one XCD (blocks b % 8 == 3), as a function of
//   ST   : producer store width 2 / 8 bytes
//   PITCH: bytes of a sentence row owned by one workgroup: 64 (two workgroups share a 128-byte line) or
//          128 (a line is written by ONE workgroup; half of every line is padding)
//   LDW  : consumer load width 16 (buffer_load_dwordx4 sc1) or 8 (two global_load_dwordx2 sc1)
// build: hipcc --offload-arch=gfx950 -O3 handoff_v3.hip -o handoff_v3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
struct Res { u64 t0, t1; unsigned bad, spins, xcc, dead; };

__device__ __forceinline__ u32x4 load16(const void* base, unsigned off, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16);
}
__device__ __forceinline__ bool unwritten(u32x4 v) {
  unsigned m = 0;
  for (int i = 0; i < 4; ++i) { const unsigned y = ~v[i]; m |= (y - 0x00010001u) & v[i] & 0x80008000u; }
  return m != 0;
}
__device__ __forceinline__ unsigned short val(unsigned s, unsigned unit, unsigned sen) { return (unsigned short)((s * 131u + unit * 7u + sen * 3u) & 0x7fffu); }

// NS sentences x 1024 units; 32 participants, each owns 32 units (8 waves x 4)
template <int ST, int PITCH, int LDW>
__global__ __launch_bounds__(512) void k(unsigned short* hx, Res* res, int place, int steps, int warm, int NS, int plain) {
  extern __shared__ unsigned lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 15u;
  const int b = blockIdx.x;
  int pi = -1;
  if (place == 0) { if (b < 32) pi = b; } else { if ((b & 7) == 3) pi = b >> 3; }
  if (pi < 0) return;
  constexpr int ROW = 32 * PITCH;                       // bytes of one sentence row
  const unsigned slot_bytes = (unsigned)NS * ROW;
  const unsigned total = (unsigned)steps * slot_bytes;
  const int cps = 128;                                  // 16-byte chunks of real data per sentence
  unsigned bad = 0, spins = 0; bool dead = false; u64 t0 = 0;
  for (int s = 0; s < steps && !dead; ++s) {
    if (s == warm && tid == 0) t0 = wall_clock64();
    if (s > 0) {
      for (int base = tid; base < NS * cps; base += 512) {
        const int sen = base / cps, j = base % cps;       // chunk j = units [8j, 8j+8) -> owner j/4, 16-byte piece j%4
        const unsigned off = (unsigned)(s - 1) * slot_bytes + sen * ROW + (j >> 2) * PITCH + (j & 3) * 16;
        u32x4 v;
        unsigned sp = 0;
        for (;;) {
          if (LDW == 16) v = load16(hx, off, total);
          else {
            const u64 a = __hip_atomic_load((const u64*)((const char*)hx + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 c = __hip_atomic_load((const u64*)((const char*)hx + off + 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[0] = (unsigned)a; v[1] = (unsigned)(a >> 32); v[2] = (unsigned)c; v[3] = (unsigned)(c >> 32);
          }
          if (!unwritten(v)) break;
          asm volatile("" ::: "memory");
          if (++sp > (1u << 18)) { dead = true; break; }
        }
        spins += sp;
        for (int q = 0; q < 4; ++q) {
          const unsigned e0 = val(s - 1, j * 8 + q * 2, sen), e1 = val(s - 1, j * 8 + q * 2 + 1, sen);
          if (!dead && v[q] != (e0 | (e1 << 16))) ++bad;
        }
        *reinterpret_cast<u32x4*>(&lds[(sen * cps + j) * 4]) = v;
      }
    }
    __syncthreads();
    if (ST == 16) {
      unsigned short* pub = reinterpret_cast<unsigned short*>(lds + 16384);      // [sen][32 units]
      const int sen = lane & 15, ul = lane >> 4;
      if (sen < NS) pub[sen * 32 + wid * 4 + ul] = val(s, pi * 32 + wid * 4 + ul, sen);
      __syncthreads();
      if (tid < NS * 4) {
        const int sn = tid >> 2, q = tid & 3;
        const u32x4 x = *reinterpret_cast<const u32x4*>(pub + sn * 32 + q * 8);
        char* dst = (char*)hx + (size_t)s * slot_bytes + sn * ROW + pi * PITCH + q * 16;
        if (plain) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(dst), "v"(x) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(x) : "memory");
      }
    }
    // publish slot s: wave w, lanes (sen = lane & 15 < NS, ul = lane >> 4): unit = pi*32 + w*4 + ul
    {
      const int sen = lane & 15, ul = lane >> 4;
      if (sen < NS) {
        char* rowp = (char*)hx + (size_t)s * slot_bytes + sen * ROW + pi * PITCH + wid * 8;
        if (ST == 2) {
          const unsigned short x = val(s, pi * 32 + wid * 4 + ul, sen);
          if (plain) __hip_atomic_store((unsigned short*)(rowp + ul * 2), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else __hip_atomic_store((unsigned short*)(rowp + ul * 2), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (ST == 16) {
          // handled below (after the block barrier)
        } else if (ul == 0) {
          u64 x = 0;
          for (int q = 0; q < 4; ++q) x |= (u64)val(s, pi * 32 + wid * 4 + q, sen) << (16 * q);
          if (plain) __hip_atomic_store((u64*)rowp, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else __hip_atomic_store((u64*)rowp, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  if (tid == 0) { res[pi].t0 = t0; res[pi].t1 = wall_clock64(); res[pi].xcc = xcc; res[pi].dead = dead; }
  atomicAdd(&res[pi].bad, bad); atomicAdd(&res[pi].spins, spins);
}

template <int ST, int PITCH, int LDW>
static void run(int place, int plain, unsigned short* hx, size_t hx_bytes, Res* res, int NS) {
  const int steps = 200, warm = 20;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(hx, 0xff, hx_bytes)); CHECK(hipMemset(res, 0, sizeof(Res) * 32));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<ST, PITCH, LDW>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipLaunchKernelGGL((k<ST, PITCH, LDW>), dim3(256), dim3(512), 100 * 1024, 0, hx, res, place, steps, warm, NS, plain);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<Res> h(32);
  CHECK(hipMemcpy(h.data(), res, sizeof(Res) * 32, hipMemcpyDeviceToHost));
  double av = 0; unsigned bad = 0, dead = 0; unsigned long long spins = 0; unsigned xm = 0;
  for (auto& r : h) { av += (double)(r.t1 - r.t0) / 100.0 / (steps - warm) / 32; bad += r.bad; dead += r.dead; spins += r.spins; xm |= 1u << r.xcc; }
  printf("place %d %s store %dB pitch %3d load %2dB NS %2d: %.3f us/step  bad %u dead %u retries/step/WG %.1f xcc 0x%02x\n",
         place, plain ? "L2   " : "write", ST, PITCH, LDW, NS, av, bad, dead, (double)spins / 32 / steps, xm);
}

int main() {
  const size_t hx_bytes = (size_t)200 * 16 * 32 * 128;
  unsigned short* hx; Res* res;
  CHECK(hipMalloc(&hx, hx_bytes)); CHECK(hipMalloc(&res, sizeof(Res) * 32));
  for (int NS : {4}) {
    run<8, 64, 16>(0, 0, hx, hx_bytes, res, NS);
    run<8, 128, 16>(0, 0, hx, hx_bytes, res, NS);
    run<2, 64, 16>(0, 0, hx, hx_bytes, res, NS);
    run<2, 128, 16>(0, 0, hx, hx_bytes, res, NS);
    run<8, 64, 8>(0, 0, hx, hx_bytes, res, NS);
    run<8, 128, 8>(0, 0, hx, hx_bytes, res, NS);
    run<8, 64, 16>(1, 1, hx, hx_bytes, res, NS);
    run<2, 64, 16>(1, 1, hx, hx_bytes, res, NS);
    run<8, 128, 16>(1, 1, hx, hx_bytes, res, NS);
    run<8, 64, 16>(1, 0, hx, hx_bytes, res, NS);
    run<16, 64, 16>(0, 0, hx, hx_bytes, res, NS);
    run<16, 128, 16>(0, 0, hx, hx_bytes, res, NS);
    run<16, 64, 16>(1, 1, hx, hx_bytes, res, NS);
  }
  return 0;
}
