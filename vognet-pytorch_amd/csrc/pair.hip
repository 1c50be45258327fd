// Horizontal fusion: two INDEPENDENT steps of one forward issued as ONE grid.
//
// A forward is a chain of ~17 short kernels; most of them fill a fraction of the chip (the
// persistent BiLSTM layer holds 64 CUs for ~46 us, the obj_tx tail 13 workgroups, ...), and the
// hardware runs at most 4 queues, i.e. 4 such kernels, at a time - more streams or queues only
// time-slice (measured: 8 hardware queues 131 us per batch instead of 101; two HIP streams per
// forward 125 us; AQL packets without the barrier bit do not overlap inside a queue at all). The
// concurrency therefore has to live INSIDE a launch: the language chain (input projection ->
// BiLSTM layer -> input projection -> BiLSTM layer -> projection) and the visual chain (encoders ->
// obj_tx QKV -> attention -> tail -> mul_tx QKV) do not depend on each other until mul_tx's
// attention, so step i of one is paired with step i of the other:
//
//     pair_kernel<A, B>: blocks [0, nA) run body A with its own virtual grid, the rest run body B.
//
// Bodies (common.h) take block index, grid size and LDS base as arguments, so the same code is the
// stand-alone kernel and one half of a pair. A pair has max(threads), max(LDS) and max(registers)
// of its halves; a 256-thread body inside a 512-thread pair lets waves 4-7 exit at once (s_barrier
// only counts live waves). A's blocks are dispatched first: the persistent BiLSTM (whose
// workgroups need a CU each and talk to each other) is always A.
//
// Host side: vog::launch has a capture mode (g_pair_capture); pair_launch runs the two steps under
// capture, looks the two kernel identities up in the registry below and issues the fused grid
// through vog::launch (so hipGraph capture and the AQL recorder see one ordinary kernel). Unknown
// combinations (other shapes, other tile choices) simply launch the two steps one after the other.
#include <functional>
#include <map>
#include <mutex>
#include <utility>
#include "gemm_dev.h"
#include "lstm_dev.h"
#include "txtail_dev.h"
#include "visenc_dev.h"
#include "qkvrb_dev.h"
#include <tuple>
#include "pair_ids.h"

#ifndef VOG_VS_PAIR_DEPTH
#define VOG_VS_PAIR_DEPTH 2      // measured (scratch/r5_ve2.sh): 2: 32.4 us for the cfg-2 pair launch, 4: 33.2, 6: 33.9; lean form 36.0
#endif

namespace vog {

thread_local std::vector<LaunchRecord>* g_pair_capture = nullptr;

template <typename A, typename B>
__global__ __launch_bounds__((A::THREADS > B::THREADS ? A::THREADS : B::THREADS))
void pair_kernel(typename A::Params a, typename B::Params b, unsigned nA, unsigned gax, unsigned gay,
                 unsigned gbx, unsigned gby, unsigned delay_b) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char pair_smem[];
  constexpr int MAXT = A::THREADS > B::THREADS ? A::THREADS : B::THREADS;
  if (blockIdx.x < nA) {
    if (A::THREADS < MAXT && (int)threadIdx.x >= A::THREADS) return;
    A::run(a, BlockCtx{blockIdx.x % gax, blockIdx.x / gax, gax, gay}, pair_smem);
  } else {
    const unsigned id = blockIdx.x - nA;
    if (B::THREADS < MAXT && (int)threadIdx.x >= B::THREADS) return;
    // perf experiments (VOG_PAIR_DELAY, units of ~0.5 us): the partner starts late, so that A's prologue has the fabric to itself
    for (unsigned i = 0; i < delay_b; ++i) __builtin_amdgcn_s_sleep(16);
    B::run(b, BlockCtx{id % gbx, id / gbx, gbx, gby}, pair_smem);
  }
}

typedef int (*PairFn)(const LaunchRecord&, const LaunchRecord&, hipStream_t);
template <typename A, typename B>
static int launch_pair(const LaunchRecord& ra, const LaunchRecord& rb, hipStream_t st) {
  constexpr int MAXT = A::THREADS > B::THREADS ? A::THREADS : B::THREADS;
  if ((int)ra.block[0] != A::THREADS || (int)rb.block[0] != B::THREADS || ra.grid[2] != 1 || rb.grid[2] != 1 ||
      ra.arg_bytes != sizeof(typename A::Params) || rb.arg_bytes != sizeof(typename B::Params))
    VOG_FAIL(-1, "pair launch: recorded launches do not match the registered bodies");
  typename A::Params pa; typename B::Params pb;
  memcpy(&pa, ra.args, sizeof(pa));
  memcpy(&pb, rb.args, sizeof(pb));
  const unsigned nA = ra.grid[0] * ra.grid[1], nB = rb.grid[0] * rb.grid[1];
  const size_t lds = ra.dyn_lds > rb.dyn_lds ? ra.dyn_lds : rb.dyn_lds;
  if (lds > 156 * 1024) VOG_FAIL(-1, "pair launch: %zu bytes of LDS", lds);
  auto kern = pair_kernel<A, B>;
  static bool attr_set = false;
  if (!attr_set) {
    // (a pair that contains __syncthreads_or carries 1 KiB of static LDS: leave room for it)
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                156 * 1024));
    attr_set = true;
  }
  static const unsigned delay_b = perf_env("VOG_PAIR_DELAY") ? (unsigned)atoi(perf_env("VOG_PAIR_DELAY")) : 0u;
  ::vog::launch(kern, dim3(nA + nB), dim3(MAXT), lds, st, pa, pb, nA, ra.grid[0], ra.grid[1], rb.grid[0], rb.grid[1], delay_b);
  VOG_LAUNCH_CHECK();
  return 0;
}

static constexpr int kPairDeclined = 1;
static std::map<std::pair<const void*, const void*>, PairFn>& registry() {
  static std::map<std::pair<const void*, const void*>, PairFn> r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the out-projection next to the QKV GEMM: 4-deep register chunk = 128 registers, so that the pair keeps the QKV workgroups'
    // 4 per CU (the stand-alone kernel's 8-deep form has 200; same k order: bit-identical)
    using SkinnyIh = GemmSkinnyBody<F16, false, 4, 1, 4>;
    using Lstm = LstmLayerBody<F16, 32>;
    r[{kid_lstm_layer_f16(), kid_vis_enc_f16()}] = &launch_pair<Lstm, VisEncBody<F16>>;
    r[{kid_lstm_layer_f16(), kid_vis_enc_stream_f16()}] = &launch_pair<Lstm, VisEncStreamBody<F16, VOG_VS_PAIR_DEPTH>>;   // (one workgroup per CU inside the pair: depth instead of occupancy)
    r[{kid_lstm_layer_f16(), kid_gemm_pipe_qkv(VOG_BF16)}] = &launch_pair<Lstm, GemmPipeBody<BF16, 64, 64, 2, EPI_QKV>>;
    r[{kid_lstm_layer_f16(), kid_gemm_pipe_qkv(VOG_F16)}] = &launch_pair<Lstm, GemmPipeBody<F16, 64, 64, 2, EPI_QKV>>;
    r[{kid_lstm_layer_f16(), kid_tx_tail_512(VOG_BF16)}] = &launch_pair<Lstm, TxTailBody<BF16, F16, 2, false, 0>>;
    r[{kid_lstm_layer_f16(), kid_tx_tail_512(VOG_F16)}] = &launch_pair<Lstm, TxTailBody<F16, F16, 2, false, 0>>;
    // hi + lo operand forms (round 6): the same three pairs for a checkpoint on the tx_split plan
    r[{kid_lstm_layer_f16(), kid_vis_enc_stream_split_f16()}] = &launch_pair<Lstm, VisEncStreamBody<F16, VOG_VS_DEPTH, true>>;
    r[{kid_lstm_layer_f16(), kid_tx_tail_split_512_f16()}] = &launch_pair<Lstm, TxTailBody<F16, F16, 2, false, 0, 1, true>>;
    r[{kid_gemm_skinny_f16(), kid_gemm_pipe_qkv_split_f16()}] = &launch_pair<SkinnyIh, GemmPipeBody<F16, 64, 64, 2, EPI_QKV, true>>;
    r[{kid_gemm_skinny_f16(), kid_gemm_pipe_qkv(VOG_BF16)}] = &launch_pair<SkinnyIh, GemmPipeBody<BF16, 64, 64, 2, EPI_QKV>>;
    r[{kid_gemm_skinny_f16(), kid_gemm_pipe_qkv(VOG_F16)}] = &launch_pair<SkinnyIh, GemmPipeBody<F16, 64, 64, 2, EPI_QKV>>;
    // the layer-1 input projection where it is a GEMM launch (more than 80 columns: cfg 3, cfg 5) next to obj_tx's QKV projection:
    // two LDS-DMA GEMM bodies of the same shape (round 6; round 2's attempt paired the 512-thread skinny form: 22.5 vs 9.5 + 7.4 us)
    r[{kid_gemm_pipe_plain3_f16(), kid_gemm_pipe_qkv(VOG_BF16)}] = &launch_pair<GemmPipeBody<F16, 64, 64, 3, EPI_PLAIN>, GemmPipeBody<BF16, 64, 64, 2, EPI_QKV>>;
    r[{kid_gemm_pipe_plain3_f16(), kid_gemm_pipe_qkv(VOG_F16)}] = &launch_pair<GemmPipeBody<F16, 64, 64, 3, EPI_PLAIN>, GemmPipeBody<F16, 64, 64, 2, EPI_QKV>>;
  });
  return r;
}

// Run two independent steps as one launch when their kernels form a registered pair; otherwise
// (or when a step is more than one launch) run them back to back. *fused (optional) reports which.
int pair_launch(const std::function<int(hipStream_t)>& fa, const std::function<int(hipStream_t)>& fb,
                hipStream_t st, bool* fused) {
  if (fused) *fused = false;
  std::vector<LaunchRecord> recs;
  if (g_pair_capture) { VOG_TRY(fa(st)); return fb(st); }      // nested: no pairing
  g_pair_capture = &recs;
  int rc = fa(st);
  const size_t na = recs.size();
  if (rc == 0) rc = fb(st);
  g_pair_capture = nullptr;
  if (rc != 0) return rc;
  if (na == 1 && recs.size() == 2) {
    auto it = registry().find({recs[0].host_fn, recs[1].host_fn});
    if (it != registry().end()) {
      const int prc = it->second(recs[0], recs[1], st);
      if (prc != kPairDeclined) {
        if (fused) *fused = prc == 0;
        return prc;
      }
    }
  }
  VOG_TRY(fa(st));
  return fb(st);
}

}  // namespace vog
