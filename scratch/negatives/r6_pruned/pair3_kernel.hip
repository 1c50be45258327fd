// pair3_kernel / pair_launch3 (chain_obj_qkv) as they stood at the start of round 6
// three bodies in one grid: A first (the persistent BiLSTM), then B, then C (C may depend on B through flags in memory:
// its blocks are dispatched after B's, so a waiting C block can never keep a B block from becoming resident)
template <typename A, typename B, typename Cc>
__global__ __launch_bounds__((A::THREADS > B::THREADS ? (A::THREADS > Cc::THREADS ? A::THREADS : Cc::THREADS)
                                                      : (B::THREADS > Cc::THREADS ? B::THREADS : Cc::THREADS)))
void pair3_kernel(typename A::Params a, typename B::Params b, typename Cc::Params c, unsigned nA, unsigned nB, unsigned gax,
                  unsigned gay, unsigned gbx, unsigned gby, unsigned gcx, unsigned gcy) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char pair3_smem[];
  constexpr int MAXT = A::THREADS > B::THREADS ? (A::THREADS > Cc::THREADS ? A::THREADS : Cc::THREADS)
                                               : (B::THREADS > Cc::THREADS ? B::THREADS : Cc::THREADS);
  if (blockIdx.x < nA) {
    if (A::THREADS < MAXT && (int)threadIdx.x >= A::THREADS) return;
    A::run(a, BlockCtx{blockIdx.x % gax, blockIdx.x / gax, gax, gay}, pair3_smem);
  } else if (blockIdx.x < nA + nB) {
    const unsigned id = blockIdx.x - nA;
    if (B::THREADS < MAXT && (int)threadIdx.x >= B::THREADS) return;
    B::run(b, BlockCtx{id % gbx, id / gbx, gbx, gby}, pair3_smem);
  } else {
    const unsigned id = blockIdx.x - nA - nB;
    if (Cc::THREADS < MAXT && (int)threadIdx.x >= Cc::THREADS) return;
    Cc::run(c, BlockCtx{id % gcx, id / gcx, gcx, gcy}, pair3_smem);
  }
}

// Three steps as one launch when a triple kernel is registered for them (fc may depend on fb: see pair3_kernel); otherwise
// fa + fb as a pair, then fc.
int pair_launch3(const std::function<int(hipStream_t)>& fa, const std::function<int(hipStream_t)>& fb,
                 const std::function<int(hipStream_t)>& fc, hipStream_t st, bool* fused) {
  if (fused) *fused = false;
  if (g_pair_capture) { VOG_TRY(fa(st)); VOG_TRY(fb(st)); return fc(st); }
  std::vector<LaunchRecord> recs;
  g_pair_capture = &recs;
  int rc = fa(st);
  const size_t na = recs.size();
  if (rc == 0) rc = fb(st);
  const size_t nb = recs.size();
  if (rc == 0) rc = fc(st);
  g_pair_capture = nullptr;
  if (rc != 0) return rc;
  if (na == 1 && nb == 2 && recs.size() == 3) {
    auto it = registry3().find({recs[0].host_fn, recs[1].host_fn, recs[2].host_fn});
    if (it != registry3().end()) {
      if (fused) *fused = true;
      return it->second(recs[0], recs[1], recs[2], st);
    }
  }
  VOG_TRY(pair_launch(fa, fb, st, nullptr));
  return fc(st);
}

