// K2: C = act(A * W^T + bias) (+residual)  — the nn.Linear of the reference
// (transformer_code.py:58-61,77-81,169-172; mdl_vog.py:182-188,202-207,224-230)
// as MFMA kernels for gfx950 (device code: gemm_dev.h, qkvrb_dev.h). Since the encoder-layer tails
// (txtail.hip) and the feature encoders (visenc.hip) have their own fused kernels, what runs here in a
// forward are the QKV projections and the M <= 64 language-path GEMMs:
//
//  * gemm_pipe:   BMxBNx64 tiles, 4 waves (2x2), v_mfma_f32_32x32x16, operands by LDS-DMA
//    (global_load_lds) through a 2-4 stage ring, one raw s_barrier per K tile, epilogue through LDS
//    (plain rows, or the Q / K / V^T MFMA-fragment images the attention kernels read). K % 64 == 0.
//  * qkv_rowall (vog_qkv_args.wqkv_p32; p100): one workgroup per 64 rows walks all output columns, rows in LDS once,
//    weights streamed in fragment order (the tail kernel's machinery).
//  * gemm_skinny: M <= 64. Weight-streaming regime: one workgroup owns 16 output columns, its 4
//    waves split K, weights and activations in fragment order go straight to registers (no LDS
//    round trip for an operand that is read exactly once), v_mfma_f32_16x16x32, partial sums meet
//    in LDS.
//  * gemm_tiled:  register-staged fallback for shapes the DMA kernel does not take (K % 64 != 0,
//    fp32 A operand with M > 64).
//
// W is [N,K] row-major 16-bit (K contiguous) = the MFMA B-operand fragment order (8 consecutive k
// per lane), or pre-packed per 16x32 / 32x16 fragment (vog_pack_w_frag / vog_pack_w_frag32).
#include <stdlib.h>
#include "gemm_dev.h"
#include "qkvrb_dev.h"
#include "pair_ids.h"

namespace vog {

// ----------------------------------------------------------------------------
// host launchers
// ----------------------------------------------------------------------------
template <typename T16, int BM, int BN, int STAGES, int EPI, bool SPLIT = false>
static int launch_pipe_cfg(const GemmParams& p, hipStream_t st) {
  constexpr size_t ring = (size_t)STAGES * (BM + BN) * 128 * (SPLIT ? 2 : 1);
  constexpr size_t epi = (size_t)4 * (BM / 2) * (BN / 2 + 4) * 4;   // LDS-staged epilogue tile
  constexpr size_t lds = ring > epi ? ring : epi;
  auto kern = gemm_pipe<T16, BM, BN, STAGES, EPI, SPLIT>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(ceil_div(p.M, BM) * ceil_div(p.N, BN), p.splitk > 1 ? p.splitk : 1);
  ::vog::launch(kern, grid, dim3(256), lds, st, p);
  VOG_LAUNCH_CHECK();
  return 0;
}

template <typename T16, int EPI>
static int launch_pipe(const GemmParams& p, hipStream_t st) {
  // VOG_GEMM_TILE (perf experiments only) forces a tile configuration
  static int force = -2;
  if (force == -2) { const char* e = perf_env("VOG_GEMM_TILE"); force = e ? atoi(e) : -1; }
  switch (force) {
    case 0: return launch_pipe_cfg<T16, 128, 128, 3, EPI>(p, st);
    case 1: return launch_pipe_cfg<T16, 128, 64, 3, EPI>(p, st);
    case 2: return launch_pipe_cfg<T16, 64, 64, 4, EPI>(p, st);
    case 3: return launch_pipe_cfg<T16, 128, 128, 2, EPI>(p, st);
    case 4: return launch_pipe_cfg<T16, 128, 64, 2, EPI>(p, st);
    case 5: return launch_pipe_cfg<T16, 64, 64, 2, EPI>(p, st);
    case 6: return launch_pipe_cfg<T16, 64, 128, 3, EPI>(p, st);
    case 7: return launch_pipe_cfg<T16, 64, 64, 3, EPI>(p, st);
    default: break;
  }
  // Measured on MI355X (scratch/mb_gemm.py, M = 800..8192; again in round 6 with exactly counted waits for 5 / 6 / 8 stages on
  // the cfg 3 / cfg 5 input projections: 17.2 / 18.3-34 / 18.1-34 us against 16.0-19.0, scratch/r6_x.sh): occupancy beats pipeline
  // depth — 2-stage rings (2-5 workgroups per CU hide each other's barrier and
  // LDS-latency stalls) are faster than 3-4-stage rings at one workgroup per CU
  // for every shape of this model. 128x128 only when there are >= 8 tiles per CU.
  auto ntiles = [&](int bm, int bn) { return (int64_t)ceil_div(p.M, bm) * ceil_div(p.N, bn); };
  if (ntiles(128, 128) >= 2048) return launch_pipe_cfg<T16, 128, 128, 2, EPI>(p, st);
  if (ntiles(128, 64) >= 512) return launch_pipe_cfg<T16, 128, 64, 2, EPI>(p, st);
  if (ntiles(64, 64) * (p.splitk > 1 ? p.splitk : 1) < 256)
    return launch_pipe_cfg<T16, 64, 64, 4, EPI>(p, st);                            // <1 tile per CU: go deep
  // long K, 1-3 tiles per CU (the BiLSTM input projections of cfg 3 / cfg 5: M = 96 / 192, N = 8192, K = 2048): a workgroup
  // walks 32 k-tiles and there are too few of them per CU to hide each other's round trips - one more stage in flight:
  // 21.5 -> 17.6 us (cfg 3), 23.1 -> 22.3 (cfg 5); scratch/r5_ih.sh
  if (p.K >= 2048 && p.splitk <= 1 && ntiles(64, 64) <= 768) return launch_pipe_cfg<T16, 64, 64, 3, EPI>(p, st);
  return launch_pipe_cfg<T16, 64, 64, 2, EPI>(p, st);
}

static bool pipe_ok(const GemmParams& p, bool a_f32) {
  const bool out_ok = (!p.c32 || ((p.ldc % 4) == 0 && ((uintptr_t)p.c32 % 16) == 0)) &&
                      (!p.c16 || ((p.ldc16 % 4) == 0 && ((uintptr_t)p.c16 % 8) == 0)) &&
                      (!p.residual || ((p.ldr % 4) == 0 && ((uintptr_t)p.residual % 16) == 0)) &&
                      (!p.bias || ((uintptr_t)p.bias % 16) == 0);
  return !a_f32 && p.M > 64 && (p.K % 64) == 0 && (p.lda % 8) == 0 && (p.ldw % 8) == 0 &&
         ((uintptr_t)p.a % 16) == 0 && ((uintptr_t)p.w % 16) == 0 && out_ok;
}

template <typename T16, bool A_F32, int EPI>
static int launch_tiled(const GemmParams& p, hipStream_t st) {
  if (pipe_ok(p, A_F32)) return launch_pipe<T16, EPI>(p, st);
  const int64_t t128 = (int64_t)ceil_div(p.M, 128) * ceil_div(p.N, 128);
  if (p.M > 64 && t128 >= 192) {
    dim3 grid(ceil_div(p.M, 128) * ceil_div(p.N, 128));
    ::vog::launch((gemm_tiled<T16, 128, 128, A_F32, EPI>), grid, dim3(256), 0, st, p);
  } else {
    dim3 grid(ceil_div(p.M, 64) * ceil_div(p.N, 64));
    ::vog::launch((gemm_tiled<T16, 64, 64, A_F32, EPI>), grid, dim3(256), 0, st, p);
  }
  VOG_LAUNCH_CHECK();
  return 0;
}

template <typename T16>
static int gemm_dispatch(const vog_gemm_args* g, hipStream_t st) {
  GemmParams p{};
  p.a = g->a; p.a_rows = g->a_rows; p.lda = g->lda;
  p.w = (const unsigned short*)g->w; p.ldw = g->ldw;
  p.bias = g->bias; p.residual = g->residual; p.ldr = g->ldr;
  p.c32 = g->c32; p.c16 = (unsigned short*)g->c16; p.ldc = g->ldc; p.ldc16 = g->ldc16;
  p.M = g->M; p.N = g->N; p.K = g->K; p.relu = g->relu; p.rep = g->rep < 1 ? 1 : g->rep;
  p.c16_bf16 = (g->c16_dtype < 0 ? (int)g->dtype : g->c16_dtype) == VOG_BF16;
  p.debug = gemm_debug_flags();
  p.out_rows = g->out_rows; p.out_rows_ncol = g->out_rows_ncol;
  p.splitk = g->splitk;
  if (g->res_vislang) {
    const vog_vislang_args* r = g->res_vislang;
    auto mk = [](int d, unsigned* mul, unsigned* shr) {
      unsigned sh = 0;
      while ((1ull << sh) < (unsigned long long)d) ++sh;
      *shr = sh;
      *mul = d <= 1 ? 0u : (unsigned)((((1ull << 32) * ((1ull << sh) - (unsigned long long)d)) / (unsigned long long)d) + 1ull);
    };
    mk(r->nsrl * r->nppf, &p.fdN_mul, &p.fdN_shr); mk(r->nppf, &p.fdP_mul, &p.fdP_shr);
    mk(r->nfrm, &p.fdF_mul, &p.fdF_shr); mk(r->nc_v, &p.fdC_mul, &p.fdC_shr);
    p.res_vis = r->vis; p.res_lang = r->lang; p.rv_nfrm = r->nfrm; p.rv_nppf = r->nppf; p.rv_nsrl = r->nsrl;
    p.rv_dv = r->dv; p.rv_dl = r->dl; p.rv_lpv = r->lang_per_vid; p.rv_ncv = r->nc_v;
  }
  if (p.splitk > 1) {
    if (!pipe_ok(p, g->a_is_f32 != 0) || g->bias || g->residual || g->relu || g->c16 || p.rep != 1 ||
        g->out_rows || g->res_vislang || !g->c32 || g->splitk > p.K / 64)
      VOG_FAIL(-1, "split-K GEMM needs the LDS-DMA path (16-bit A, K %% 64 == 0, M > 64) and a bare fp32 output");
    return launch_pipe<T16, EPI_PLAIN>(p, st);
  }
  p.w_frag = g->w_frag; p.a_frag = g->a_frag;
  if (g->w_lo && !(p.M <= 64 && (p.K % 32) == 0 && p.w_frag))
    VOG_FAIL(-1, "vog_gemm_args.w_lo: only the M <= 64 kernel with fragment-ordered weights takes hi + lo operands");
  if (p.a_frag && !(p.M <= 64 && (p.K % 32) == 0 && !g->a_is_f32 && !g->a_rows))
    VOG_FAIL(-1, "a_frag activations are only valid for the M <= 64 kernel with a 16-bit A (M=%d K=%d)", p.M, p.K);
  if (p.w_frag && !(p.M <= 64 && (p.K % 32) == 0 && (p.N % 16) == 0))
    VOG_FAIL(-1, "w_frag weights are only valid for the M <= 64 kernel (M=%d N=%d K=%d)", p.M, p.N, p.K);
  if (p.M <= 64 && (p.K % 32) == 0) {
    const int ncol = ceil_div(p.N, 16);
    // (a 16-deep weight prefetch measured SLOWER: 26 vs 18.6 us at M=48,N=8192,K=2048, 230 VGPRs)
    static const int nt_env = perf_env("VOG_SKINNY_NT") ? atoi(perf_env("VOG_SKINNY_NT")) : 0;
    // NT = 2 halves the L2 re-reads of A but also the workgroup count: measured SLOWER at
    // M=48,N=8192,K=2048 (12.7 vs 11.1 us) and much slower at small N -> opt-in for experiments
    const int nt = nt_env == 2 ? 2 : 1;
    const size_t lds1 = (size_t)4 * 1 * 4 * 64 * 4 * sizeof(float), lds2 = (size_t)4 * 2 * 4 * 64 * 4 * sizeof(float);
    // wide projections (LSTM input GEMMs, N = 8192): 8 waves split K, 2 column tiles per workgroup
    const bool wide = ncol >= 512 && (ncol % 2) == 0 && p.K / 32 >= 64 && nt_env != 1;
    if (wide) {
      dim3 grid(ncol / 2, 1);
      const size_t lds8 = (size_t)8 * 2 * 4 * 64 * 4 * sizeof(float);          // 64 KiB
      if (g->a_is_f32) {
        auto kern = gemm_skinny<T16, true, 8, 2, 8>;
        static bool attr = false;
        if (!attr) { VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8)); attr = true; }
        ::vog::launch(kern, grid, dim3(512), lds8, st, p);
      } else {
        auto kern = gemm_skinny<T16, false, 8, 2, 8>;
        static bool attr = false;
        if (!attr) { VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8)); attr = true; }
        ::vog::launch(kern, grid, dim3(512), lds8, st, p);
      }
    } else if (nt == 2) {
      dim3 grid(ceil_div(ncol, 2), 1);
      if (g->a_is_f32) ::vog::launch((gemm_skinny<T16, true, 8, 2>), grid, dim3(256), lds2, st, p);
      else ::vog::launch((gemm_skinny<T16, false, 8, 2>), grid, dim3(256), lds2, st, p);
    } else {
      dim3 grid(ncol, ncol < 128 ? ceil_div(p.M, 16) : 1);
      // short K (a wave owns <= 2 k-steps: the language half of mul_tx's layer-0 QKV, K = 256): the 2-deep register chunk
      // - 84 instead of 192 registers (the 8-deep form loads six zero fragments per operand), i.e. 4 instead of 2
      // workgroups per CU for the 144 workgroups of that launch. Same k order per wave: bit-identical.
      const bool shortk = p.K / 32 <= 8;
      if (g->w_lo) {
        // hi + lo operands (round 6): fp32 rows split in the kernel, fragment-ordered W and W_lo
        if (!(g->a_is_f32 && p.w_frag))
          VOG_FAIL(-1, "hi + lo M <= 64 GEMM: needs an fp32 A operand and fragment-ordered weights (w_frag) with their remainder (w_lo)");
        p.w_lo = (const unsigned short*)g->w_lo;
        ::vog::launch((gemm_skinny<T16, true, 2, 1, 4, true>), grid, dim3(256), lds1, st, p);
      } else if (shortk) {
        if (g->a_is_f32) ::vog::launch((gemm_skinny<T16, true, 2, 1, 4>), grid, dim3(256), lds1, st, p);
        else ::vog::launch((gemm_skinny<T16, false, 2, 1, 4>), grid, dim3(256), lds1, st, p);
      } else if (g->a_is_f32) ::vog::launch((gemm_skinny<T16, true, 8, 1>), grid, dim3(256), lds1, st, p);
      else ::vog::launch((gemm_skinny<T16, false, 8, 1>), grid, dim3(256), lds1, st, p);
    }
    VOG_LAUNCH_CHECK();
    return 0;
  }
  if (g->a_is_f32) return launch_tiled<T16, true, EPI_PLAIN>(p, st);
  return launch_tiled<T16, false, EPI_PLAIN>(p, st);
}

const void* kid_gemm_skinny_f16() { return reinterpret_cast<const void*>(gemm_skinny<F16, false, 8, 1, 4>); }
const void* kid_gemm_skinny_wide_f16() { return reinterpret_cast<const void*>(gemm_skinny<F16, false, 8, 2, 8>); }
const void* kid_gemm_pipe_qkv_split_f16() { return reinterpret_cast<const void*>(gemm_pipe<F16, 64, 64, 2, EPI_QKV, true>); }
const void* kid_gemm_pipe_plain3_f16() { return reinterpret_cast<const void*>(gemm_pipe<F16, 64, 64, 3, EPI_PLAIN>); }
const void* kid_gemm_pipe_qkv(int dtype) {
  return dtype == VOG_BF16 ? reinterpret_cast<const void*>(gemm_pipe<BF16, 64, 64, 2, EPI_QKV>)
                           : reinterpret_cast<const void*>(gemm_pipe<F16, 64, 64, 2, EPI_QKV>);
}


int gemm_run(const vog_gemm_args* g, hipStream_t st) {
  VOG_CHECK_ARG(g && g->a && g->w && (g->c32 || g->c16));
  VOG_CHECK_ARG(g->M > 0 && g->N > 0 && g->K > 0 && (g->K % 8) == 0);
  VOG_CHECK_ARG((g->lda % (g->a_is_f32 ? 4 : 8)) == 0 && (g->ldw % 8) == 0);
  VOG_CHECK_ARG(!(g->residual && g->rep > 1));
  VOG_CHECK_ARG(!g->res_vislang || (!g->residual && g->res_vislang->vis && g->res_vislang->lang &&
                                    (g->res_vislang->dv % 4) == 0 && (g->res_vislang->dl % 4) == 0 &&
                                    g->res_vislang->dv + g->res_vislang->dl == g->N));
  VOG_CHECK_ARG(!g->out_rows || (g->rep <= 1 && g->out_rows_ncol > 0 && (g->out_rows_ncol % 4) == 0));
  VOG_DISPATCH_DTYPE(g->dtype, return gemm_dispatch<T16>(g, st));
  return 0;
}

int qkv_rowblock_supported(int n_out, int K);

int qkv_run(const vog_qkv_args* a, hipStream_t st) {
  VOG_CHECK_ARG(a && a->x16 && a->wqkv && a->q && a->k && a->vt);
  VOG_CHECK_ARG(a->K % 8 == 0 && a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->npad >= a->N && (a->npad % 32) == 0 && (a->dp % 32) == 0);
  GemmParams p{};
  p.a = a->x16; p.lda = a->ldx; p.w = (const unsigned short*)a->wqkv; p.ldw = a->ldw;
  p.M = a->S * a->N; p.N = 3 * a->H * a->dp; p.K = a->K; p.rep = 1;
  if (a->pl) {
    VOG_CHECK_ARG(a->nsrl > 0 && a->nppf > 0 && a->nfrm > 0 && a->nc_v > 0 && a->N == a->nsrl * a->nppf &&
                  (a->S % a->nfrm) == 0);
    p.pl = a->pl; p.st_nsrl = a->nsrl; p.st_nppf = a->nppf; p.st_nfrm = a->nfrm; p.st_lpv = a->lang_per_vid;
    p.st_ncv = a->nc_v;
    p.M = a->S * a->nppf;                        // visual rows
    p.st_kv_vis = a->kv_visual_only ? 1 : 0; p.npad_kv = a->npad_kv;
    VOG_CHECK_ARG(!a->kv_visual_only || (a->npad_kv >= a->nppf && (a->npad_kv % 32) == 0));
    if (!pipe_ok(p, false) || p.M <= 64)
      VOG_FAIL(-1, "structured QKV needs the LDS-DMA GEMM (K %% 64 == 0, > 64 visual rows)");
  }
  p.q = (unsigned short*)a->q; p.k = (unsigned short*)a->k; p.vt = (unsigned short*)a->vt;
  p.ntok = a->N; p.H = a->H; p.dp = a->dp; p.npad = a->npad;
  {   // tokens per sequence seen by the plain fragment writers (visual-only K/V use nppf)
    const int d = (a->pl && a->kv_visual_only) ? a->nppf : a->N;
    unsigned sh = 0;
    while ((1ull << sh) < (unsigned long long)d) ++sh;
    p.fdT_shr = sh;
    p.fdT_mul = d <= 1 ? 0u : (unsigned)((((1ull << 32) * ((1ull << sh) - (unsigned long long)d)) / (unsigned long long)d) + 1ull);
  }
  p.debug = gemm_debug_flags();
  if (a->x16_lo || a->wqkv_lo || a->q_lo || a->k_lo) {
    // hi + lo operands (round 6): the LDS-DMA GEMM with two images per stage, Q / K written as hi + lo fragments
    VOG_CHECK_ARG(a->x16_lo && a->wqkv_lo && a->q_lo && a->k_lo && !a->pl);
    p.a_lo = a->x16_lo; p.w_lo = (const unsigned short*)a->wqkv_lo; p.q_lo = (unsigned short*)a->q_lo; p.k_lo = (unsigned short*)a->k_lo;
    if (!pipe_ok(p, false) || ((uintptr_t)p.a_lo % 16) != 0 || ((uintptr_t)p.w_lo % 16) != 0)
      VOG_FAIL(-1, "hi + lo QKV projection needs the LDS-DMA GEMM (K %% 64 == 0, more than 64 rows, 16-byte aligned operands)");
    VOG_DISPATCH_DTYPE(a->dtype, return (launch_pipe_cfg<T16, 64, 64, 2, EPI_QKV, true>(p, st)));
  }
  // one workgroup per row block walks ALL output columns (QkvRowAllBody, qkvrb_dev.h); shapes it does not take (K = 768: the staged
  // rows + 8 epilogue tiles pass 160 KB of LDS) run the tiled GEMM on `wqkv`
  if (a->wqkv_p32 && qkv_rowblock_supported(p.N, p.K) && (p.N % 32) == 0 && QkvRowAllBody<F16>::lds_bytes(p.K) <= 160 * 1024) {
    p.w_p32 = (const unsigned short*)a->wqkv_p32;
    const int nrb = ceil_div(p.M, 64);
    const size_t lds_a = QkvRowAllBody<F16>::lds_bytes(p.K);
    VOG_DISPATCH_DTYPE(a->dtype, {
      auto kern = qkv_rowall_kernel<T16>;
      static bool attr_set = false;
      if (!attr_set) {
        VOG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
      }
      ::vog::launch(kern, dim3(nrb), dim3(512), lds_a, st, p);
    });
    VOG_LAUNCH_CHECK();
    return 0;
  }
  VOG_DISPATCH_DTYPE(a->dtype, return (launch_tiled<T16, false, EPI_QKV>(p, st)));
  return 0;
}

int qkv_rowblock_supported(int n_out, int K) {
  return (K % 128) == 0 && K >= 256 && K <= 1024 && (n_out % 64) == 0;
}

}  // namespace vog

extern "C" int vog_pack_w_frag(const float* w, int64_t ld, int N, int K, void* dst_host, vog_dtype dtype) {
  VOG_CHECK_ARG(w && dst_host && N > 0 && K > 0 && (N % 16) == 0 && (K % 32) == 0 && ld >= K);
  unsigned short* dst = (unsigned short*)dst_host;
  const int ksteps = K / 32;
  for (int nt = 0; nt < N / 16; ++nt)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const float* src = w + (int64_t)(nt * 16 + (lane & 15)) * ld + ks * 32 + (lane >> 4) * 8;
        unsigned short* d = dst + (((int64_t)nt * ksteps + ks) * 64 + lane) * 8;
        for (int j = 0; j < 8; ++j) {
          if (dtype == VOG_BF16) {
            unsigned int u; memcpy(&u, &src[j], 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            d[j] = (unsigned short)(u >> 16);
          } else {
            _Float16 h = (_Float16)src[j];
            memcpy(&d[j], &h, 2);
          }
        }
      }
  return 0;
}

extern "C" int vog_gemm_bias_act(const vog_gemm_args* g, void* stream) {
  return vog::gemm_run(g, (hipStream_t)stream);
}
extern "C" int vog_qkv_rowblock_supported(int n_out, int K) { return vog::qkv_rowblock_supported(n_out, K); }
extern "C" int vog_qkv_proj(const vog_qkv_args* a, void* stream) {
  return vog::qkv_run(a, (hipStream_t)stream);
}
