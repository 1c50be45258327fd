// K1: rel_attention_fwd — softmax((Q K^T + bias)/sqrt(d_model)) V per
// (sequence, head), flash-style on gfx950, with the reference's relative
// position bias recomputed per block instead of materialised.
//
// Replaces RelAttention.forward / Attention.forward (transformer_code.py:136-160,
// 42-50) and the whole of compute_pe (mdl_vog.py:456-490): the reference builds
// pe[S,N,N,H] = relu(Linear(5,H)(box_i - box_j)) (5.9 GB of temporaries at p100);
// the Linear is affine in the box, so pe[i,j,h] = relu(u[i,h] - u[j,h] + b[h])
// with u = W_pe . box — three VALU ops per (i,j), nothing stored.
//
// Five kernels share the math below and differ in how K / V reach the MFMA (csrc/attention_dev.h,
// attn_tile2_dev.h):
// attn_sb (N <= 128: one workgroup per (sequence, head), keys split over the 4 waves, global softmax
// statistics exchanged through LDS before P.V), attn_frag (128 < N < 512: per-wave running softmax,
// LDS tree merge), attn_tile (N >= 512: K / V^T blocks staged once per workgroup by LDS-DMA, double
// buffered), attn_tile2 (bf16, N >= 1024, head dim <= 192: two waves per SIMD, fixed-reference softmax
// with a guard flag + attn_tile as the fallback pass) and attn_struct / attn_struct1 (mul_tx layer 0:
// separable softmax over visual + language keys).
// Common mapping (wave64, v_mfma_f32_32x32x16): a wave owns 32 queries;
//   * "swapped" products so a query's softmax row lives in one lane:
//       S^T[key][q] = K_blk . Q^T        (A = K fragment, B = Q fragment)
//       O^T[d][q]  += V^T_blk . P^T      (A = V fragment, B = P from registers)
//     C/D has col = lane&31 = query for both, so running max / sum / rescale are
//     per-lane scalars, and P goes from the S accumulator registers straight into
//     the B operand: the MFMA k index is a free summation index, so register r
//     of half hi is declared to be k = hi*8 + (r&7) of k-step r>>3 and V is stored
//     with the matching key permutation (common.h frag_v). No cross-lane
//     movement, no LDS round trip for P.
//   * q, k, v arrive in MFMA-fragment order (written that way by the QKV GEMM
//     epilogue / vog_qkv_combine): every operand fragment is one contiguous KiB,
//     loaded straight into registers (or, for attn_tile, into LDS by global_load_lds) with
//     16-byte-per-lane coalesced accesses: no transposes, no swizzles.
#include <type_traits>
#include "attention_dev.h"
#include "attn_tile2_dev.h"
#include "attn_struct_lds_dev.h"
#include "attn_struct_ef_dev.h"

namespace vog {

template <typename T16, int NDB>
static int launch_attn_struct(const AttnStructParams& p, hipStream_t st) {
  const int nqb = (p.nsrl * p.nppf + 31) / 32;
  const size_t lds = p.npad_kv == 32 ? (size_t)(32 + p.nsrl * 3 * NDB * 32) * sizeof(float)
                                     : (size_t)p.npad_kv * sizeof(float);
  if (lds > 64 * 1024) VOG_FAIL(-1, "struct attention: %d visual keys exceed the LDS budget", p.nppf);
  dim3 grid(p.S * p.H * ((nqb + 3) / 4));
  if (p.q_lo && !(p.npad_kv == 32 && p.q_visual && p.kv_lo))
    VOG_FAIL(-1, "struct attention with hi + lo operands: needs q_visual and one visual key block (nppf <= 32)");
  // several visual key blocks: K / V^T through an LDS ring shared by the workgroup (attn_struct_lds_dev.h)
  static int lds_form = -2;           // VOG_ATTN_STRUCT_LDS=0 (perf experiments): per-wave L2 loads instead
  if (lds_form == -2) { const char* e = perf_env("VOG_ATTN_STRUCT_LDS"); lds_form = e ? atoi(e) : 1; }
  constexpr int NF = (NDB * 32) / 16 + 2 * NDB;
  const size_t lds_res = (size_t)4 * NF * 1024 + ((size_t)p.npad_kv + (size_t)p.nsrl * 3 * NDB * 32) * sizeof(float);
  // several visual key blocks, queries formed in the kernel: the E x F factorisation (attn_struct_ef_dev.h; head dim 128 / 256)
  static int ef_form = -2;            // VOG_ATTN_STRUCT_EF=0 (perf experiments): the per-(a, p) kernels below
  if (ef_form == -2) { const char* e = perf_env("VOG_ATTN_STRUCT_EF"); ef_form = e ? atoi(e) : 1; }
  if constexpr (NDB % 4 == 0) {
    // (needs the guard word and its fallback, the LDS-ring kernel below: without them the per-row kernels run)
    if (ef_form && p.q_visual && p.npad_kv > 32 && p.npad_kv <= 512 && p.nsrl == EF_MAXA && !(p.dbg) &&
        (p.guard != nullptr || ef_form == 2) && lds_form && lds_res <= 150 * 1024) {
      const size_t lds_ef = attn_struct_ef_lds<NDB>(p.npad_kv);
      if (lds_ef <= 150 * 1024) {
        const dim3 grid_ef(p.S * p.H * ((p.nppf + 31) / 32));
        if (ef_form != 3) {           // 256 registers, two workgroups per CU whose phases overlap (141 vs 244 us at cfg 4)
          auto kern = attn_struct_ef_kernel<T16, NDB, 2>;
          static bool attr_ef2 = false;
          if (!attr_ef2) {
            VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_ef2 = true;
          }
          ::vog::launch(kern, grid_ef, dim3(256), lds_ef, st, p);
        } else {                      // (VOG_ATTN_STRUCT_EF=3, perf experiments: up to 512 registers, one workgroup per CU)
          auto kern = attn_struct_ef_kernel<T16, NDB, 1>;
          static bool attr_ef = false;
          if (!attr_ef) {
            VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_ef = true;
          }
          ::vog::launch(kern, grid_ef, dim3(256), lds_ef, st, p);
        }
        VOG_LAUNCH_CHECK();
        if (p.guard) {                // fallback pass: runs only if a row left the factorisation's safe range
          AttnStructParams pf = p;
          pf.guard_gate = 1;
          auto kern = attn_struct_lds_kernel<T16, NDB>;
          static bool attr_slf = false;
          if (!attr_slf) {
            VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_slf = true;
          }
          ::vog::launch(kern, grid, dim3(256), lds_res, st, pf);
          VOG_LAUNCH_CHECK();
        }
        return 0;
      }
    }
  }
  if (p.npad_kv > 32 && lds_form && lds_res <= 150 * 1024) {
    auto kern = attn_struct_lds_kernel<T16, NDB>;
    static bool attr_sl = false;
    if (!attr_sl) {
      VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_sl = true;
    }
    ::vog::launch(kern, grid, dim3(256), lds_res, st, p);
    VOG_LAUNCH_CHECK();
    return 0;
  }
  // hi + lo operands (round 6): one visual key block, queries formed in the kernel - the gt5 shapes
  if (p.q_lo) {
    if constexpr (((NDB * 32) / 16) % 2 == 0) {
      if (p.npad_kv == 32 && p.q_visual && p.kv_lo) {
        ::vog::launch((attn_struct1_lean_kernel<T16, NDB, true>), grid, dim3(256), lds, st, p);
        VOG_LAUNCH_CHECK();
        return 0;
      }
    }
    VOG_FAIL(-1, "struct attention with hi + lo operands: needs q_visual, one visual key block (nppf <= 32) and a head dim of 64 / 128 / 192 / 256");
  }
  // one visual key block, queries formed in the kernel: every operand through LDS after one round trip (round 6)
  static int dma_form = -2;           // VOG_ATTN_STRUCT_DMA=0 (perf experiments): the lean form below
  if (dma_form == -2) { const char* e = perf_env("VOG_ATTN_STRUCT_DMA"); dma_form = e ? atoi(e) : 1; }
  if constexpr (((NDB * 32) / 16) % 2 == 0) {
    const size_t lds_dma = (size_t)(7 * NDB) * 1024 + (size_t)2 * (p.nsrl + 1) * NDB * 32 * sizeof(unsigned short) + 32 * sizeof(float);
    if (dma_form && p.npad_kv == 32 && p.q_visual && !p.out_lo && p.nsrl <= 16 && lds_dma <= 72 * 1024) {
      auto kern = attn_struct1_dma_kernel<T16, NDB>;
      static bool attr_dma = false;
      if (!attr_dma) {
        VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        attr_dma = true;
      }
      ::vog::launch(kern, grid, dim3(256), lds_dma, st, p);
      VOG_LAUNCH_CHECK();
      return 0;
    }
  }
  // one visual key block: the lean form (<= 128 registers, four workgroups per CU)
  if constexpr (((NDB * 32) / 16) % 2 == 0) {
    if (p.npad_kv == 32) {
      ::vog::launch((attn_struct1_lean_kernel<T16, NDB>), grid, dim3(256), lds, st, p);
      VOG_LAUNCH_CHECK();
      return 0;
    }
  }
  ::vog::launch((attn_struct_kernel<T16, NDB>), grid, dim3(256), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}

template <typename T16>
static int attn_struct_dispatch(const AttnStructParams& p, hipStream_t st) {
  switch (p.dp) {
    case 32: return launch_attn_struct<T16, 1>(p, st);
    case 64: return launch_attn_struct<T16, 2>(p, st);
    case 128: return launch_attn_struct<T16, 4>(p, st);
    case 192: return launch_attn_struct<T16, 6>(p, st);
    case 256: return launch_attn_struct<T16, 8>(p, st);
    default: VOG_FAIL(-1, "struct attention: unsupported padded head dim %d (32/64/128/192/256)", p.dp);
  }
}

int attn_struct_run(const vog_attn_struct_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->q && a->kv && a->vv && a->pl && a->out16);
  VOG_CHECK_ARG(a->S > 0 && a->H > 0 && a->nsrl > 0 && a->nsrl <= 32 && a->nppf > 0 && a->nfrm > 0 && a->nc_v > 0);
  VOG_CHECK_ARG((a->npad_kv % 32) == 0 && a->npad_kv >= a->nppf);
  VOG_CHECK_ARG(a->q_visual || ((a->npad_q % 32) == 0 && a->npad_q >= a->nsrl * a->nppf));
  VOG_CHECK_ARG(!a->use_rel || (a->u && a->pe_b && a->seq_per_vid > 0));
  AttnStructParams p{(const unsigned short*)a->q, (const unsigned short*)a->kv, (const unsigned short*)a->vv, a->pl,
                     (unsigned short*)a->out16, a->u, a->pe_b, a->S, a->H, a->dp, a->nsrl, a->nppf, a->npad_q,
                     a->npad_kv, a->nfrm, a->lang_per_vid, a->nc_v, a->use_rel, a->seq_per_vid, a->NP, a->inv_scale,
                     a->q_visual ? 1 : 0, 0, a->guard_flag, 0,
                     (const unsigned short*)a->q_lo, (const unsigned short*)a->kv_lo, (unsigned short*)a->out16_lo, a->logit_max};
  VOG_CHECK_ARG((a->q_lo == nullptr) == (a->kv_lo == nullptr));
  { static const int dbg = perf_env("VOG_ATTN_STRUCT_DEBUG") ? atoi(perf_env("VOG_ATTN_STRUCT_DEBUG")) : 0; p.dbg = dbg; }
  VOG_DISPATCH_DTYPE(a->dtype, return (attn_struct_dispatch<T16>(p, st)));
  return 0;
}

__global__ void attn_guard_clear_kernel(int* g) { if (threadIdx.x == 0) *g = 0; }

template <typename T16, int NDB>
static int launch_attn(const AttnParams& p, hipStream_t st) {
  static int force_general = -2;      // VOG_ATTN_GENERAL=1: perf experiments only
  if (force_general == -2) { const char* e = perf_env("VOG_ATTN_GENERAL"); force_general = e ? atoi(e) : 0; }
  if (p.q_lo) {
    // hi + lo Q / K (round 6): the two kernels of the gt5 shapes carry the three-MFMA contraction
    if (p.N <= 128) {
      constexpr int HB = (NDB + 1) / 2;
      const size_t lds = ((size_t)4 * HB * 16 * 64 + 4 * 2 * 64 + p.npad) * sizeof(float);
      auto kern = attn_sb_kernel<T16, NDB, true>;
      static bool attr_sbs = false;
      if (!attr_sbs && lds > 48 * 1024) {
        VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        attr_sbs = true;
      }
      ::vog::launch(kern, dim3(ceil_div(p.N, 32) * p.H * p.S), dim3(256), lds, st, p);
      VOG_LAUNCH_CHECK();
      return 0;
    }
    if constexpr (((NDB * 32) / 16) % 2 == 0) {
      if (p.npad <= 256) {
        const size_t ldsl = (size_t)8 * 2 * 64 * 16 + (size_t)2 * 8 * 32 * sizeof(float) + (size_t)p.npad * sizeof(float);
        ::vog::launch((attn_frag_lean_kernel<T16, NDB, true>), dim3(ceil_div(p.N, 32) * p.H * p.S), dim3(256), ldsl, st, p);
        VOG_LAUNCH_CHECK();
        return 0;
      }
    }
    VOG_FAIL(-1, "rel_attention with hi + lo operands: sequences of more than 256 tokens are not supported (N = %d)", p.N);
  }
  if (p.N <= 128 && !force_general) {
    constexpr int HB = (NDB + 1) / 2;
    const size_t lds = ((size_t)4 * HB * 16 * 64 + 4 * 2 * 64 + p.npad) * sizeof(float);
    auto kern = attn_sb_kernel<T16, NDB>;
    static bool attr_sb = false;
    if (!attr_sb && lds > 48 * 1024) {
      VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      attr_sb = true;
    }
    dim3 grid(ceil_div(p.N, 32) * p.H * p.S);
    ::vog::launch(kern, grid, dim3(256), lds, st, p);
    VOG_LAUNCH_CHECK();
    return 0;
  }
  // long sequences: enough 128-query groups to fill the chip -> shared K/V tiles through LDS
  static int tile_min = -2;           // VOG_ATTN_TILE_MIN (perf experiments): N threshold, 0 = never
  if (tile_min == -2) { const char* e = perf_env("VOG_ATTN_TILE_MIN"); tile_min = e ? atoi(e) : 512; }
  // very long sequences: two waves per SIMD, fixed-reference softmax (attn_tile2_dev.h)
  static int tile2_min = -2;          // VOG_ATTN_TILE2_MIN (perf experiments): N threshold, 0 = never
  if (tile2_min == -2) { const char* e = perf_env("VOG_ATTN_TILE2_MIN"); tile2_min = e ? atoi(e) : 1024; }
  bool fallback_pass = false;         // tile2 ran: attn_tile_kernel below runs only if the guard was raised
  if constexpr (NDB <= 6) {           // (head dim 256 would spill at 2 waves per SIMD)
    constexpr int NF2 = (NDB * 32) / 16 + 2 * NDB;
    const size_t lds2 = (size_t)4 * NF2 * 1024 + (size_t)p.npad * sizeof(float);
    if (tile2_min > 0 && p.N >= tile2_min && p.guard && !force_general && lds2 <= 160 * 1024) {
      auto kern = attn_tile2_kernel<T16, NDB>;
      static bool attr_t2 = false;
      if (!attr_t2) {
        VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_t2 = true;
      }
      if (!p.guard_precleared)
        ::vog::launch(attn_guard_clear_kernel, dim3(1), dim3(64), 0, st, p.guard);   // (a kernel, not a memset: graph-capturable on any stream)
      dim3 grid(ceil_div(p.N, 256) * p.H * p.S);
      ::vog::launch(kern, grid, dim3(512), lds2, st, p);
      VOG_LAUNCH_CHECK();
      fallback_pass = true;
    }
  }
  AttnParams pt = p;
  if (!fallback_pass) pt.guard = nullptr;
  // (after attn_tile2 the running-maximum tile kernel is ALWAYS the second pass, whatever the experiment
  // knobs say: it is the only kernel that honours the guard flag - anything else would redo the whole
  // attention unconditionally)
  if (fallback_pass || (tile_min > 0 && p.N >= tile_min && !force_general)) {
    constexpr int NF = (NDB * 32) / 16 + 2 * NDB;
    const size_t lds = (size_t)2 * NF * 1024 + (size_t)p.npad * sizeof(float);
    if (lds > 150 * 1024) VOG_FAIL(-1, "rel_attention: sequence of %d tokens exceeds the LDS budget", p.N);
    auto kern = attn_tile_kernel<T16, NDB>;
    static bool attr_tile = false;
    if (!attr_tile && lds > 48 * 1024) {
      VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_tile = true;
    }
    dim3 grid(ceil_div(p.N, 128) * p.H * p.S);
    ::vog::launch(kern, grid, dim3(256), lds, st, pt);
    VOG_LAUNCH_CHECK();
    return 0;
  }
  // <= 8 key blocks: the lean form (<= 128 registers, four workgroups per CU; VOG_ATTN_FRAG_LEAN=0 (perf experiments): the old one)
  static int lean_f = -2;
  if (lean_f == -2) { const char* e = perf_env("VOG_ATTN_FRAG_LEAN"); lean_f = e ? atoi(e) : 1; }
  // ... with every operand requested in one round trip (8 waves; VOG_ATTN_FRAG8=0 (perf experiments): the lean form)
  static int frag8_f = -2;
  if (frag8_f == -2) { const char* e = perf_env("VOG_ATTN_FRAG8"); frag8_f = e ? atoi(e) : 1; }
  if constexpr (((NDB * 32) / 16) % 2 == 0) {
    if (frag8_f && lean_f && p.npad <= 256 && !p.out_lo) {
      constexpr int KS = (NDB * 32) / 16;
      const size_t lds8 = (size_t)(KS + 16) * 1024 + (size_t)2 * 8 * 32 * sizeof(float) + (size_t)p.npad * sizeof(float);
      dim3 grid(ceil_div(p.N, 32) * p.H * p.S);
      auto kern8 = attn_frag8_kernel<T16, NDB>;
      static bool attr_f8 = false;
      if (!attr_f8 && lds8 > 48 * 1024) {
        VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern8), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_f8 = true;
      }
      ::vog::launch(kern8, grid, dim3(512), lds8, st, p);
      VOG_LAUNCH_CHECK();
      return 0;
    }
  }
  if constexpr (((NDB * 32) / 16) % 2 == 0) {
    if (lean_f && p.npad <= 256) {
      const size_t ldsl = (size_t)8 * 2 * 64 * 16 + (size_t)2 * 8 * 32 * sizeof(float) + (size_t)p.npad * sizeof(float);
      dim3 grid(ceil_div(p.N, 32) * p.H * p.S);
      ::vog::launch((attn_frag_lean_kernel<T16, NDB>), grid, dim3(256), ldsl, st, p);
      VOG_LAUNCH_CHECK();
      return 0;
    }
  }
  const size_t lds = ((size_t)2 * NDB * 16 * 64 + 4 * 2 * 64 + p.npad) * sizeof(float);
  if (lds > 150 * 1024) VOG_FAIL(-1, "rel_attention: sequence of %d tokens exceeds the LDS budget", p.N);
  auto kern = attn_frag_kernel<T16, NDB>;
  static bool attr_set = false;       // benign race: idempotent
  if (!attr_set && lds > 48 * 1024) {
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_set = true;
  }
  dim3 grid(ceil_div(p.N, 32) * p.H * p.S);
  ::vog::launch(kern, grid, dim3(256), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}

template <typename T16>
static int attn_dispatch(const AttnParams& p, hipStream_t st) {
  switch (p.dp) {
    case 32: return launch_attn<T16, 1>(p, st);
    case 64: return launch_attn<T16, 2>(p, st);
    case 128: return launch_attn<T16, 4>(p, st);
    case 192: return launch_attn<T16, 6>(p, st);
    case 256: return launch_attn<T16, 8>(p, st);
    default: VOG_FAIL(-1, "rel_attention: unsupported padded head dim %d (32/64/128/192/256)", p.dp);
  }
}

int attn_head_pad(int dh) {
  const int opts[5] = {32, 64, 128, 192, 256};
  for (int i = 0; i < 5; ++i) if (dh <= opts[i]) return opts[i];
  return -1;
}

int attn_run(const vog_attn_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->q && a->k && a->vt && a->out16);
  VOG_CHECK_ARG(a->S > 0 && a->N > 0 && a->H > 0 && a->npad >= a->N && (a->npad % 32) == 0);
  VOG_CHECK_ARG(!a->use_rel || (a->u && a->pe_b && a->n_box > 0 && a->seq_per_vid > 0));
  AttnParams p{};
  p.q = (const unsigned short*)a->q; p.k = (const unsigned short*)a->k;
  p.vt = (const unsigned short*)a->vt; p.out = (unsigned short*)a->out16;
  p.u = a->u; p.pe_b = a->pe_b;
  p.S = a->S; p.N = a->N; p.H = a->H; p.dp = a->dp; p.npad = a->npad; p.use_rel = a->use_rel;
  p.n_box = a->n_box; p.seq_per_vid = a->seq_per_vid; p.NP = a->NP; p.inv_scale = a->inv_scale;
  p.guard = a->guard_flag; p.guard_precleared = a->guard_precleared;
  VOG_CHECK_ARG((a->q_lo == nullptr) == (a->k_lo == nullptr));
  p.q_lo = (const unsigned short*)a->q_lo; p.k_lo = (const unsigned short*)a->k_lo; p.out_lo = (unsigned short*)a->out16_lo;
  p.logit_max = a->logit_max;
  VOG_DISPATCH_DTYPE(a->dtype, return attn_dispatch<T16>(p, st));
  return 0;
}

}  // namespace vog

extern "C" int vog_rel_attention_struct_fwd(const vog_attn_struct_args* a, void* stream) {
  return vog::attn_struct_run(a, (hipStream_t)stream);
}

extern "C" int vog_rel_attention_fwd(const vog_attn_args* a, void* stream) {
  return vog::attn_run(a, (hipStream_t)stream);
}
