// VisEncLeanBody (round 4) and VisEncWideBody (round 5) as they stood at the start of round 6

// ---------------------------------------------------------------------------------------------------
// "Lean" form for the launch it shares with the persistent BiLSTM layer (pair.hip): there the encoders
// have ~40 us to finish and only the CUs the BiLSTM leaves free, and what they cost the OTHER batches in
// flight is busy-CU time. The wide form above keeps 480 workgroups busy (~4600 CU*us per cfg-2 forward,
// most of it re-reading the fp32 rows 8x and the weights 60x out of L2); this one gives a workgroup 64
// rows x 128 columns: the fp32 rows are read ONCE per column half, converted and staged in LDS (fragment
// order) per K chunk of 256, the 8 waves own 16 columns each and stream their weight fragments once.
// 32 workgroups, ~1 MB through each CU: ~25 us (proposals) / ~37 us (segments), ~900 CU*us.
// ---------------------------------------------------------------------------------------------------
#ifndef VOG_VE_G
#define VOG_VE_G 4
#endif
#ifndef VOG_VE_KC
#define VOG_VE_KC 256
#endif
#ifndef VOG_VE_BOTH
#define VOG_VE_BOTH 0
#endif
// BOTH (round 4, measured and left OFF): ONE workgroup computes both 128-column halves of its 64 rows from the same staged A
// images - the fp32 rows cross the CU's vector memory path once instead of twice (1.5 instead of 2 MB per row block); the blocks
// that used to own the second halves exit at once. Half as many workgroups, each a longer dependent chain: cfg 4 72 -> 94 us,
// a cfg-2 forward alone 201 -> 218 us, 57.5 -> 55.5 k queries/s (scratch/r4_ve.sh): the kernel is bound by the latency of a
// workgroup's chunk loop and by how many of them run side by side, not by ingest bytes. Per chunk: half 0 with the weight set loaded
// during the previous unit, half 1 with the set loaded during half 0; accumulation order per column unchanged (bit-identical).
template <typename T16, bool BOTH = (VOG_VE_BOTH != 0)>
struct VisEncLeanBody {
  using Params = VisEncParams;
  static constexpr int THREADS = 512;
  static constexpr int RB = 64, KC = VOG_VE_KC;
  static constexpr int KSC = KC / 32;                       // k-steps per chunk
  static constexpr int PIECES = KC / 8, RPP = THREADS / PIECES, NPASS = RB / RPP;   // A staging: 8-float pieces per row, rows per pass
  static constexpr int NH = BOTH ? 2 : 1;
  static constexpr size_t LDS = (size_t)2 * (RB / 16) * (KC / 32) * 1024 > (size_t)RB * (128 + 8) * 2
                                    ? (size_t)2 * (RB / 16) * (KC / 32) * 1024 : (size_t)RB * (128 + 8) * 2;   // two A-chunk images (fragment order) | the chained epilogue's tile

  static __device__ __forceinline__ void run(const VisEncParams& a, const BlockCtx& cx, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // item = (row block of 64, column half); tiles0 / tiles_all count 16-row tiles: 4 per block
    const int nb0 = (a.tiles0 + 3) >> 2, nb_all = nb0 + ((a.tiles_all - a.tiles0 + 3) >> 2);
    // the two column halves of a row block read the same 64 fp32 feature rows (512 KB): they sit 8 block ids apart, i.e.
    // on the same XCD (block b is dealt to XCD b % 8), so the second read is served by that XCD's L2
    // (2.1 x over-fetch at p100 with the halves on neighbouring XCDs)
    const int grp = cx.bx >> 4, pos = cx.bx & 15;
    const int blk = grp * 8 + (pos & 7), half0 = pos >> 3;
    if (blk >= nb_all) return;
    if (BOTH && half0 != 0) return;
    const bool second = blk >= nb0;
    const float* qx = second ? a.p[1].x : a.p[0].x;
    const unsigned short* qw = second ? a.p[1].w : a.p[0].w;
    const float* qb = second ? a.p[1].bias : a.p[0].bias;
    const int qM = second ? a.p[1].M : a.p[0].M, qN = second ? a.p[1].N : a.p[0].N;
    const int qK = second ? a.p[1].K : a.p[0].K, qrep = second ? a.p[1].rep : a.p[0].rep;
    const int qcol0 = second ? a.p[1].col0 : a.p[0].col0;
    const int m0 = (second ? blk - nb0 : blk) * RB;
    if (half0 * 128 >= qN) return;
    const int nh = BOTH ? (qN > 128 ? 2 : 1) : 1;            // halves this workgroup computes
    const int ksteps = qK >> 5, nchunk = qK / KC;            // K % 256 == 0
    int n0[NH]; bool n_ok[NH]; const u16x8* wf[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      n0[h] = (half0 + h) * 128 + w * 16;                    // this wave's 16 columns of half h
      n_ok[h] = n0[h] < qN;
      wf[h] = reinterpret_cast<const u16x8*>(qw) + ((int64_t)((n_ok[h] ? n0[h] : 0) >> 4) * ksteps) * 64 + lane;
    }
    // A staging: thread -> (row, 8-column piece): PIECES pieces per row and chunk, RPP rows per pass, NPASS passes
    const int pr = tid / PIECES, pc = tid % PIECES;
    f32x4 acc[NH][RB / 16];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int mt = 0; mt < RB / 16; ++mt) acc[h][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 xa[NPASS][2];
    u16x8 wq[2][KSC];
    auto load_a = [&](int c) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        int m = m0 + ps * RPP + pr;
        m = m < qM ? m : qM - 1;
        const float* src = qx + (int64_t)m * qK + c * KC + pc * 8;
        xa[ps][0] = *reinterpret_cast<const float4*>(src);
        xa[ps][1] = *reinterpret_cast<const float4*>(src + 4);
      }
    };
    auto store_a = [&](int c) {      // -> fragment order [m tile][k-step][lane = kgroup*16 + m%16][8 halfwords]
      unsigned char* img = smem + (size_t)(c & 1) * (RB / 16) * KSC * 1024;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const u16x8 h = {to16<T16>(xa[ps][0].x), to16<T16>(xa[ps][0].y), to16<T16>(xa[ps][0].z), to16<T16>(xa[ps][0].w),
                         to16<T16>(xa[ps][1].x), to16<T16>(xa[ps][1].y), to16<T16>(xa[ps][1].z), to16<T16>(xa[ps][1].w)};
        const int ks = pc >> 2, kgp = pc & 3;
        const int rl = ps * RPP + pr;                          // row of the block: tile rl / 16, row rl % 16 of the tile
        *reinterpret_cast<u16x8*>(img + (((rl >> 4) * KSC + ks) * 64 + kgp * 16 + (rl & 15)) * 16) = h;
      }
    };
    auto load_w = [&](u16x8 (&q)[KSC], int c, int h) {
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) q[ks] = wf[h][(c * KSC + ks) * 64];
    };
    auto mfmas = [&](const u16x8 (&q)[KSC], int c, f32x4 (&ac)[RB / 16]) {
      // A fragments come from LDS in units of VG k-steps, one unit AHEAD of the MFMAs that use them (the scheduling fences
      // keep hipcc from folding this into read-2 / wait / MFMA-2, which exposes the LDS latency 16 times per chunk).
      // unit u = (row tile u / UPT, k-steps (u % UPT) * VG ...); VG = 4 keeps the whole body under 192 registers, which is
      // what lets a <= 128-register workgroup of another stream share the CU when this body rides in the BiLSTM layer's
      // launch (the pair kernel allocates the maximum of its two bodies)
      const unsigned char* img = smem + (size_t)(c & 1) * (RB / 16) * KSC * 1024;
      constexpr int VG = VOG_VE_G < KSC ? VOG_VE_G : KSC, UPT = KSC / VG, NU = (RB / 16) * UPT;
      u16x8 fa[VG], fb[VG];
      auto rd = [&](u16x8 (&f)[VG], int u) {
#pragma unroll
        for (int j = 0; j < VG; ++j)
          f[j] = *reinterpret_cast<const u16x8*>(img + (((u / UPT) * KSC + (u % UPT) * VG + j) * 64 + lane) * 16);
      };
      rd(fa, 0);
#pragma unroll
      for (int u = 0; u < NU; u += 2) {
        __builtin_amdgcn_sched_barrier(0);
        rd(fb, u + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < VG; ++j) ac[u / UPT] = mfma16<T16>(fa[j], q[(u % UPT) * VG + j], ac[u / UPT]);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < NU) rd(fa, u + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < VG; ++j) ac[(u + 1) / UPT] = mfma16<T16>(fb[j], q[((u + 1) % UPT) * VG + j], ac[(u + 1) / UPT]);
      }
    };
    // (fence-free barriers: __syncthreads() would wait for the next chunk's loads, issued just above it)
    if constexpr (!BOTH) {
      load_a(0); load_w(wq[0], 0, 0);
      for (int c = 0; c < nchunk; c += 2) {
        store_a(c);                                            // image (c & 1) was last read two chunks ago
        if (c + 1 < nchunk) { load_a(c + 1); load_w(wq[1], c + 1, 0); }
        lds_barrier();
        mfmas(wq[0], c, acc[0]);
        if (c + 1 < nchunk) {
          store_a(c + 1);
          if (c + 2 < nchunk) { load_a(c + 2); load_w(wq[0], c + 2, 0); }
          lds_barrier();
          mfmas(wq[1], c + 1, acc[0]);
        }
      }
    } else {
      // units (chunk c, half h): weight set u & 1; the set of unit u + 1 is in flight while unit u computes
      load_a(0); load_w(wq[0], 0, 0);
      for (int c = 0; c < nchunk; ++c) {
        store_a(c);                                            // image (c & 1) was last read two chunks ago
        if (c + 1 < nchunk) load_a(c + 1);
        if (nh == 2) load_w(wq[1], c, NH - 1);
        lds_barrier();
        mfmas(wq[0], c, acc[0]);
        if (c + 1 < nchunk) load_w(wq[0], c + 1, 0);
        if (nh == 2) mfmas(wq[1], c, acc[NH - 1]);
      }
    }
    // D[row = 4*(lane>>4) + reg][col = lane & 15]
    const bool chained = a.done_flags != nullptr && a.c16 != nullptr && (a.ldc & 7) == 0 && (qcol0 & 7) == 0;
    // chained form (consumers in the same launch): the 16-bit tile is parked in LDS and written THROUGH as 16-byte chunks
    // (2-byte write-through stores, one per lane and element, cost more than the launch the chaining saves)
    constexpr int TP = 128 + 8;                                // tile pitch in halfwords (272 B: 16-byte aligned rows)
    unsigned short* tile = reinterpret_cast<unsigned short*>(smem);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (h >= nh) break;
      const int half = half0 + h;
      const int col = n0[h] + (lane & 15);
      if (chained) __syncthreads();                            // every wave is done reading the A images / the previous half's tile
      if (n_ok[h] && col < qN) {
        const float b = qb[col];
#pragma unroll
        for (int mt = 0; mt < RB / 16; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = mt * 16 + (lane >> 4) * 4 + r;
            const int row = m0 + rl;
            if (row >= qM) continue;
            const float o = fmaxf(acc[h][mt][r] + b, 0.f);
            const unsigned short hv = a.c16_bf16 ? to16<BF16>(o) : to16<F16>(o);
            if (chained) tile[rl * TP + w * 16 + (lane & 15)] = hv;
            const int nrep = a.rep_first_only ? 1 : qrep;
            for (int j = 0; j < nrep; ++j) {
              const int64_t off = ((int64_t)row * qrep + j) * a.ldc + qcol0 + col;
              if (a.c32) a.c32[off] = o;
              if (a.c16 && !chained) {
                if (a.done_flags) __hip_atomic_store(&a.c16[off], hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write through
                else a.c16[off] = hv;
              }
            }
          }
      }
      if (chained) {
        __syncthreads();
        const int ncol_wg = (qN - half * 128) < 128 ? (qN - half * 128) : 128;     // columns of this half (multiple of 16)
        const int cpr = ncol_wg >> 3;                                               // 16-byte chunks per row
        const int nrep = a.rep_first_only ? 1 : qrep;
        const int rows_wg = (qM - m0) < RB ? (qM - m0) : RB;
        const int total = rows_wg * nrep * cpr;
        for (int id = tid; id < total; id += THREADS) {
          const int ch = id % cpr, rj = id / cpr, j = rj % nrep, rl = rj / nrep;
          const u32x4 v = *reinterpret_cast<const u32x4*>(tile + rl * TP + ch * 8);
          u32x4* dst = reinterpret_cast<u32x4*>(a.c16 + ((int64_t)(m0 + rl) * qrep + j) * a.ldc + qcol0 + half * 128 + ch * 8);
          asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
        }
      }
    }
    if (a.done_flags) {                                       // every thread of the workgroup gets here
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        for (int h = 0; h < nh; ++h)
          __hip_atomic_store(&a.done_flags[blk * 2 + half0 + h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
};

template <typename T16>
__global__ __launch_bounds__(512) void vis_enc_lean_kernel(VisEncParams a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vl_smem[];
  VisEncLeanBody<T16>::run(a, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, vl_smem);
}

// ---------------------------------------------------------------------------------------------------
// "Wide" stream form (round 5) for p100-sized launches (>= 8192 proposal rows): 128 rows x ALL 256 columns per workgroup.
// The stream form above moves every fp32 row through the CUs twice (once per column half) and every weight fragment once per
// 64 rows: 500 workgroups x 1 MB = 0.5 GB of L2 -> CU traffic for 134 MB of features (11 TB/s at 45 us). Here a row is read
// ONCE and a weight fragment once per 128 rows: a quarter of the weight bytes, half of the row bytes. 8 waves x 32 columns
// (two 16-column tiles) x 8 row tiles: 64 accumulator registers, two register sets of (32 row-piece + 32 weight) registers,
// ~240 in all - one workgroup per CU, which is what the launch gets anyway when it rides with the BiLSTM layer (the pair
// kernel allocates the layer's 190 registers for every block). Same k order per output as the other stream form.
// ---------------------------------------------------------------------------------------------------
template <typename T16>
struct VisEncWideBody {
  using Params = VisEncParams;
  static constexpr int THREADS = 512;
  static constexpr int RB = 128, KC = 128, KSC = KC / 32;   // 4 k-steps per chunk
  static constexpr int PIECES = KC / 8, RPP = THREADS / PIECES, NPASS = RB / RPP;   // 16 pieces per row, 32 rows per pass, 4 passes
  static constexpr int IMG = (RB / 16) * KSC * 1024;        // one A-chunk image: 32 KB
  static constexpr size_t LDS = (size_t)2 * IMG;
  static constexpr int DEPTH = 2;

  static __device__ __forceinline__ void run(const VisEncParams& a, const BlockCtx& cx, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb0 = (a.p[0].M + RB - 1) / RB, nb_all = nb0 + (a.p[1].M + RB - 1) / RB;
    // the FEW long segment blocks (K = 3072: 24 chunks) first, so that they do not start behind a round of proposal blocks
    const int nb1 = nb_all - nb0;
    if ((int)cx.bx >= nb_all) return;
    const bool second = (int)cx.bx < nb1;
    const int blk = second ? (int)cx.bx : (int)cx.bx - nb1;
    const float* qx = second ? a.p[1].x : a.p[0].x;
    const unsigned short* qw = second ? a.p[1].w : a.p[0].w;
    const float* qb = second ? a.p[1].bias : a.p[0].bias;
    const int qM = second ? a.p[1].M : a.p[0].M, qN = second ? a.p[1].N : a.p[0].N;
    const int qK = second ? a.p[1].K : a.p[0].K, qrep = second ? a.p[1].rep : a.p[0].rep;
    const int qcol0 = second ? a.p[1].col0 : a.p[0].col0;
    const int m0 = blk * RB;
    const int ksteps = qK >> 5, nchunk = qK / KC;
    const int n0 = w * 32;                                   // this wave's 32 columns (two 16-column tiles)
    const bool n_ok0 = n0 < qN, n_ok1 = n0 + 16 < qN;
    const u16x8* wf0 = reinterpret_cast<const u16x8*>(qw) + ((int64_t)((n_ok0 ? n0 : 0) >> 4) * ksteps) * 64 + lane;
    const u16x8* wf1 = reinterpret_cast<const u16x8*>(qw) + ((int64_t)((n_ok1 ? n0 + 16 : 0) >> 4) * ksteps) * 64 + lane;
    const int pr = tid / PIECES, pc = tid % PIECES;
    const float* xrow[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      int m = m0 + ps * RPP + pr;
      m = m < qM ? m : qM - 1;
      xrow[ps] = qx + (int64_t)m * qK + pc * 8;
    }
    f32x4 acc[RB / 16][2];
#pragma unroll
    for (int mt = 0; mt < RB / 16; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float4 xa[DEPTH][NPASS][2];
    u16x8 wq[DEPTH][2][KSC];
    auto request = [&](int set, int c) {
      const int cc = c < nchunk ? c : nchunk - 1;            // (clamped: no conditional load)
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const f32x4* src = reinterpret_cast<const f32x4*>(xrow[ps] + cc * KC);
        const f32x4 v0 = __builtin_nontemporal_load(src);    // read exactly once in this launch
        const f32x4 v1 = __builtin_nontemporal_load(src + 1);
        xa[set][ps][0] = make_float4(v0[0], v0[1], v0[2], v0[3]);
        xa[set][ps][1] = make_float4(v1[0], v1[1], v1[2], v1[3]);
      }
#pragma unroll
      for (int ks = 0; ks < KSC; ++ks) { wq[set][0][ks] = wf0[(cc * KSC + ks) * 64]; wq[set][1][ks] = wf1[(cc * KSC + ks) * 64]; }
    };
    auto store_a = [&](int set, int c) {
      unsigned char* img = smem + (size_t)(c & 1) * IMG;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const u16x8 h = {to16<T16>(xa[set][ps][0].x), to16<T16>(xa[set][ps][0].y), to16<T16>(xa[set][ps][0].z), to16<T16>(xa[set][ps][0].w),
                         to16<T16>(xa[set][ps][1].x), to16<T16>(xa[set][ps][1].y), to16<T16>(xa[set][ps][1].z), to16<T16>(xa[set][ps][1].w)};
        const int ks = pc >> 2, kgp = pc & 3;
        const int rl = ps * RPP + pr;
        *reinterpret_cast<u16x8*>(img + (((rl >> 4) * KSC + ks) * 64 + kgp * 16 + (rl & 15)) * 16) = h;
      }
    };
    auto mfmas = [&](int set, int c) {
      const unsigned char* img = smem + (size_t)(c & 1) * IMG;
      u16x8 fa[KSC], fb[KSC];
      auto rd = [&](u16x8 (&f)[KSC], int mt) {
#pragma unroll
        for (int j = 0; j < KSC; ++j) f[j] = *reinterpret_cast<const u16x8*>(img + ((mt * KSC + j) * 64 + lane) * 16);
      };
      rd(fa, 0);
#pragma unroll
      for (int mt = 0; mt < RB / 16; mt += 2) {
        __builtin_amdgcn_sched_barrier(0);
        rd(fb, mt + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KSC; ++j) {
          acc[mt][0] = mfma16<T16>(fa[j], wq[set][0][j], acc[mt][0]);
          acc[mt][1] = mfma16<T16>(fa[j], wq[set][1][j], acc[mt][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mt + 2 < RB / 16) rd(fa, mt + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KSC; ++j) {
          acc[mt + 1][0] = mfma16<T16>(fb[j], wq[set][0][j], acc[mt + 1][0]);
          acc[mt + 1][1] = mfma16<T16>(fb[j], wq[set][1][j], acc[mt + 1][1]);
        }
      }
    };
    request(0, 0);
    int c0 = 0;
    for (; c0 + DEPTH <= nchunk; c0 += DEPTH) {              // nchunk is even (K % 256 == 0)
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        const int c = c0 + j;
        store_a(j, c);
        request((j + DEPTH - 1) % DEPTH, c + DEPTH - 1);
        lds_barrier();
        mfmas(j, c);
      }
    }
    // D[row = 4*(lane>>4) + reg][col = lane & 15]
    const int nrep = a.rep_first_only ? 1 : qrep;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int col = n0 + ct * 16 + (lane & 15);
      if (col >= qN) continue;
      const float b = qb[col];
#pragma unroll
      for (int mt = 0; mt < RB / 16; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + mt * 16 + (lane >> 4) * 4 + r;
          if (row >= qM) continue;
          const float o = fmaxf(acc[mt][ct][r] + b, 0.f);
          const unsigned short hv = a.c16_bf16 ? to16<BF16>(o) : to16<F16>(o);
          for (int j = 0; j < nrep; ++j) {
            const int64_t off = ((int64_t)row * qrep + j) * a.ldc + qcol0 + col;
            if (a.c32) a.c32[off] = o;
            if (a.c16) a.c16[off] = hv;
          }
        }
    }
  }
};

template <typename T16>
__global__ __launch_bounds__(512) void vis_enc_wide_kernel(VisEncParams a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vw_smem[];
  VisEncWideBody<T16>::run(a, BlockCtx{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}, vw_smem);
}

}  // namespace vog
