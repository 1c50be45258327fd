p='include/vog_hip.h'
s=open(p).read()
old='''/* host: [2][4R][R] fp32 (weight_hh_l*, weight_hh_l*_reverse) -> fragment order, 16 bit.'''
new='''/* ALL T steps of one BiLSTM layer in ONE launch (persistent workgroups): 2 x R/32
 * workgroups, each keeps its 128 rows of W_hh in registers for the whole sequence; the
 * hidden state is exchanged between steps through `hx` with 8-byte agent-scope
 * (write-through / L1-bypassing) atomics and a per-direction arrival counter in `sync`.
 * Replaces T launches of vog_bilstm_step: a dependent graph node costs ~4.4 us on MI355X
 * and the GPU retires at most ~0.7 kernels/us over 4 queues, so the 2T step launches were
 * the largest single cost of the forward. Requires Bn <= 16 and R/32 in {1,2,4,32}
 * (vog_bilstm_layer_supported); hx ([2][2][16][R] t16) and sync (16 x u32) must be zero at
 * launch; every wait is bounded (on timeout sync[2] is set and the kernel drains).
 * Launch at most 4 instances concurrently (64 workgroups x 1 per CU each). */
typedef struct vog_lstm_layer_args {
  const float* gxs; const void* whh; void* hx; uint32_t* sync; void* out16;
  const int64_t* lens; int Bn, T, R; vog_dtype dtype;
} vog_lstm_layer_args;
int vog_bilstm_layer_supported(int Bn, int R);
int vog_bilstm_layer(const vog_lstm_layer_args* a, void* stream);

/* host: [2][4R][R] fp32 (weight_hh_l*, weight_hh_l*_reverse) -> fragment order, 16 bit.'''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/lib.py'
s=open(p).read()
s=s.replace('''class VislangArgs(C.Structure):''','''class LstmLayerArgs(C.Structure):
    _fields_ = [("gxs", c_vp), ("whh", c_vp), ("hx", c_vp), ("sync", c_vp), ("out16", c_vp),
                ("lens", c_vp), ("Bn", c_i32), ("T", c_i32), ("R", c_i32), ("dtype", c_i32)]


class VislangArgs(C.Structure):''',1)
s=s.replace('''    "vog_lstm_schedule":''','''    "vog_bilstm_layer_supported": (c_i32, [c_i32, c_i32]),
    "vog_bilstm_layer": (c_i32, [C.POINTER(LstmLayerArgs), c_vp]),
    "vog_lstm_schedule":''')
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/lstm.hip'
s=open(p).read()
old='''__global__ void lstm_schedule_kernel('''
new=r'''// ----------------------------------------------------------------------------
// persistent layer kernel: all T steps, both directions, one launch
// ----------------------------------------------------------------------------
struct LstmLayerParams {
  const float* gxs; const unsigned short* whh; unsigned long long* hx; unsigned int* sync;
  unsigned short* out16; const int64_t* lens; int Bn, T, R;
};

#define VOG_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <typename T16, int KSTEPS>
__global__ __launch_bounds__(256) void lstm_layer_kernel(LstmLayerParams p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.y, G = gridDim.x;
  const int R = p.R;
  const int tile0 = blockIdx.x * 8 + wid * 2;            // two 16-row tiles (8 units) per wave
  const int b = lane & 15, ul = lane >> 4, kg = (lane >> 4) * 8;
  const bool valid_b = b < p.Bn;
  const int len = valid_b ? (int)p.lens[b] : 0;

  // this wave's 32 rows of W_hh: registers for the whole sequence
  u16x8 wf[2][KSTEPS];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
      wf[t][ks] = *reinterpret_cast<const u16x8*>(
          p.whh + ((((int64_t)dir * (R / 4) + tile0 + t) * KSTEPS + ks) * 64 + lane) * 8);

  float c[2] = {0.f, 0.f}, h_own[2] = {0.f, 0.f};
  const int64_t hx_dir = (int64_t)dir * 16 * R / 4;        // u64 units; [parity][dir][16][R]
  const int64_t hx_par = (int64_t)2 * 16 * R / 4;
  bool dead = false;

  for (int s = 0; s < p.T; ++s) {
    // input projections of this step (address-independent of everything else)
    float gin[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        gin[t][r] = valid_b ? p.gxs[(((int64_t)dir * p.T + s) * p.Bn + b) * 4 * R + (int64_t)r * R + (tile0 + t) * 4 + ul]
                            : 0.f;
    // h_{s-1} of ALL units: written by the other workgroups with write-through atomics,
    // read here with L1-bypassing atomics (agent scope on both sides: no fences needed)
    const unsigned long long* hp = p.hx + (s & 1) * hx_par + hx_dir + (int64_t)b * R / 4;
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      unsigned long long lo = 0, hi = 0;
      if (valid_b) {
        lo = __hip_atomic_load(hp + (ks * 32 + kg) / 4, VOG_RLX_AGENT);
        hi = __hip_atomic_load(hp + (ks * 32 + kg) / 4 + 1, VOG_RLX_AGENT);
      }
      u16x8 fh;
      fh[0] = (unsigned short)lo; fh[1] = (unsigned short)(lo >> 16); fh[2] = (unsigned short)(lo >> 32);
      fh[3] = (unsigned short)(lo >> 48);
      fh[4] = (unsigned short)hi; fh[5] = (unsigned short)(hi >> 16); fh[6] = (unsigned short)(hi >> 32);
      fh[7] = (unsigned short)(hi >> 48);
      acc[0] = mfma16<T16>(wf[0][ks], fh, acc[0]);
      acc[1] = mfma16<T16>(wf[1][ks], fh, acc[1]);
    }
    const bool active = s < len;
    const int pos = dir == 0 ? s : len - 1 - s;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int unit = (tile0 + t) * 4 + ul;
      if (active) {
        const float gi = acc[t][0] + gin[t][0], gf = acc[t][1] + gin[t][1];
        const float gg = acc[t][2] + gin[t][2], go = acc[t][3] + gin[t][3];
        c[t] = sigm(gf) * c[t] + sigm(gi) * tanh_(gg);
        const float hn = sigm(go) * tanh_(c[t]);
        const unsigned short h16 = to16<T16>(hn);
        h_own[t] = from16<T16>(h16);
        p.out16[((int64_t)b * p.T + pos) * 2 * R + (int64_t)dir * R + unit] = h16;
      }
      // publish h_s of this tile: 4 units of one sentence = one 8-byte write-through store
      const unsigned int x0 = to16<T16>(h_own[t]);
      const unsigned int x1 = __shfl(x0, b + 16), x2 = __shfl(x0, b + 32), x3 = __shfl(x0, b + 48);
      if (lane < 16 && valid_b) {
        const unsigned long long v = (unsigned long long)x0 | ((unsigned long long)x1 << 16) |
                                     ((unsigned long long)x2 << 32) | ((unsigned long long)x3 << 48);
        __hip_atomic_store(p.hx + ((s + 1) & 1) * hx_par + hx_dir + ((int64_t)b * R + (tile0 + t) * 4) / 4, v,
                           VOG_RLX_AGENT);
      }
    }
    if (s + 1 == p.T) break;                             // nothing reads h_T through hx
    // arrive (all stores of this workgroup acknowledged first), then wait for the other G-1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(p.sync + dir, 1u, VOG_RLX_AGENT);
      const unsigned int want = (unsigned int)G * (unsigned int)(s + 1);
      if (!dead) {
        unsigned int spins = 0;
        while (__hip_atomic_load(p.sync + dir, VOG_RLX_AGENT) < want) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 21)) {                    // ~1 s: give up, flag, drain
            __hip_atomic_store(p.sync + 2, 1u, VOG_RLX_AGENT);
            break;
          }
        }
      }
    }
    __syncthreads();
    if (!dead && __hip_atomic_load(p.sync + 2, VOG_RLX_AGENT) != 0) dead = true;   // uniform enough: only skips waits
  }
  // final hidden state rows (h of the last ACTIVE step of every sentence)
  if (valid_b) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      p.out16[((int64_t)p.Bn * p.T + b) * 2 * R + (int64_t)dir * R + (tile0 + t) * 4 + ul] = to16<T16>(h_own[t]);
  }
}

__global__ void lstm_schedule_kernel('''
assert old in s; s=s.replace(old,new,1)
old='''extern "C" int vog_lstm_schedule('''
new='''extern "C" int vog_bilstm_layer_supported(int Bn, int R) {
  const int ks = R / 32;
  return Bn >= 1 && Bn <= 16 && R % 32 == 0 && (ks == 1 || ks == 2 || ks == 4 || ks == 32);
}

extern "C" int vog_bilstm_layer(const vog_lstm_layer_args* a, void* stream) {
  VOG_CHECK_ARG(a && a->gxs && a->whh && a->hx && a->sync && a->out16 && a->lens && a->T > 0);
  if (!vog_bilstm_layer_supported(a->Bn, a->R))
    VOG_FAIL(-1, "persistent BiLSTM layer: unsupported Bn=%d R=%d (use vog_bilstm_step)", a->Bn, a->R);
  vog::LstmLayerParams p{a->gxs, (const unsigned short*)a->whh, (unsigned long long*)a->hx, a->sync,
                         (unsigned short*)a->out16, a->lens, a->Bn, a->T, a->R};
  dim3 grid(a->R / 32, 2);
  hipStream_t st = (hipStream_t)stream;
#define VOG_LAUNCH_LAYER(KS)                                                                     \\
  VOG_DISPATCH_DTYPE(a->dtype, hipLaunchKernelGGL((vog::lstm_layer_kernel<T16, KS>), grid, dim3(256), 0, st, p))
  switch (a->R / 32) {
    case 1: VOG_LAUNCH_LAYER(1); break;
    case 2: VOG_LAUNCH_LAYER(2); break;
    case 4: VOG_LAUNCH_LAYER(4); break;
    default: VOG_LAUNCH_LAYER(32); break;
  }
#undef VOG_LAUNCH_LAYER
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_lstm_schedule('''
assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)

p='vognet-pytorch_amd/csrc/forward.hip'
s=open(p).read()
s=s.replace("  int graph_dag = 0;                    // capture the language chain as a parallel branch","  int graph_dag = 0;                    // capture the language chain as a parallel branch\n  int lstm_persistent = 1;              // one launch per BiLSTM layer when the shape allows")
old='''    p.add("lstm_c_" + std::to_string(l), (int64_t)g.Bn16 * 2 * g.R * 4);'''
new='''    p.add("lstm_c_" + std::to_string(l), (int64_t)g.Bn16 * 2 * g.R * 4);
    p.add("lstm_hx_" + std::to_string(l), (int64_t)2 * 2 * 16 * g.R * 2);
    p.add("lstm_sync_" + std::to_string(l), 64);'''
assert old in s; s=s.replace(old,new)
old='''      for (int s = 0; s < T; ++s) {
        vog_lstm_step_args la{};'''
new='''      if (c->lstm_persistent && vog_bilstm_layer_supported(Bn, R)) {
        vog_lstm_layer_args pa{};
        pa.gxs = gx; pa.whh = c->whh[l]; pa.hx = ws.at<void>("lstm_hx_" + std::to_string(l));
        pa.sync = ws.at<uint32_t>("lstm_sync_" + std::to_string(l));
        pa.out16 = ws.at<void>("lstm_out16_" + std::to_string(l));
        pa.lens = b->srl_arg_word_mask_len; pa.Bn = Bn; pa.T = T; pa.R = R; pa.dtype = et;
        steps.push_back({"lstm_layer", [=](hipStream_t st) { return vog_bilstm_layer(&pa, st); }});
        continue;
      }
      for (int s = 0; s < T; ++s) {
        vog_lstm_step_args la{};'''
assert old in s; s=s.replace(old,new)
s=s.replace('''  if (strcmp(name, "graph_dag") == 0) { c->graph_dag = value ? 1 : 0; return 0; }''','''  if (strcmp(name, "graph_dag") == 0) { c->graph_dag = value ? 1 : 0; return 0; }
  if (strcmp(name, "lstm_persistent") == 0) { c->lstm_persistent = value ? 1 : 0; return 0; }''')
open(p,'w').write(s)
