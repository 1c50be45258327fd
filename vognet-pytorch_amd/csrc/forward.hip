// Whole-forward engine: weight registry, workspace plan, launch sequence,
// hipGraph capture. Replaces Conc{TEMP,SPAT,SEP}.forward
// (mdl_conc_single.py:68-127, mdl_conc_sep.py:131-217) and
// Evaluator*.get_out_results_boxes (eval_vsrl_corr.py:162-424) with ~50 kernel
// launches on caller-owned buffers; no allocation and no sync on the launch path.
#include <stdarg.h>
#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include <stdlib.h>
#include "common.h"

namespace vog {

// ---- error string -------------------------------------------------------------
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int attn_head_pad(int dh);
int tx_tail_supported(int d, int dh, int kwo);
int64_t tx_tail_scratch_bytes(int M, int d);
int vis_encode_supported(int prop_dim, int seg_dim, int prop_enc, int seg_enc);
int pair_launch(const std::function<int(hipStream_t)>& fa, const std::function<int(hipStream_t)>& fb,
                hipStream_t st, bool* fused);

// ---- host fp32 -> 16 bit ------------------------------------------------------
static unsigned short h_to16(float f, int dt) {
  if (dt == VOG_BF16) {
    unsigned int u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  }
  _Float16 h = (_Float16)f;
  unsigned short r;
  memcpy(&r, &h, 2);
  return r;
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct TxLayer {
  unsigned short *wqkv, *wo, *w1, *w2;     // 16-bit, padded
  unsigned short* wqkv_lang_f;             // Wqkv[:, d_vis:] in fragment order (structured layer 0)
  unsigned short *wo_p, *w1_p, *w2_p;      // 32x16 fragment order (fused encoder tail, txtail.hip), or null
  unsigned short *wqkv_p, *wqkv_pv;        // padded Wqkv in the same order (all-columns QKV of p100, qkvrb_dev.h): all d columns / the first d_vis
  float *b1, *b2, *ln1g, *ln1b, *ln2g, *ln2b;
  // hi + lo operands (tx_split): 16-bit remainders t16(w - t16(w)) in the layouts of wqkv, wqkv_lang_f, wo_p, w1_p, w2_p
  unsigned short *wqkv_lo, *wqkv_lang_f_lo, *wo_p_lo, *w1_p_lo, *w2_p_lo;
};
struct TxWeights {
  int d = 0, H = 0, dp = 0, dh = 0, n_layers = 0, use_rel = 0;
  std::vector<int> head_off, head_dim;
  std::vector<TxLayer> layers;
  float *pe_w = nullptr, *pe_b = nullptr;
};

}  // namespace vog

using namespace vog;

struct vog_ctx {
  vog_model_desc d;
  std::vector<std::string> names;                       // required weights
  std::map<std::string, int64_t> numel;
  std::map<std::string, std::vector<float>> host;
  std::vector<void*> allocs;
  bool finalized = false;
  int tx_split = 0;                     // round 6: hi + lo 16-bit operands (three MFMAs per product) for everything that feeds attention
                                        // logits - encoders, QKV projections, Q.K^T, and the tails whose output is another layer's
                                        // input: the plan for checkpoints whose attention is too sharp for 16-bit logits but does
                                        // not need the fp32 path (engine.py picks it from the weights; set before vog_ctx_finalize)
  unsigned short *w_prop_f_lo = nullptr, *w_seg_f_lo = nullptr;
  int lstm_inject_stall = 0;            // test hook: persistent layer launches behave as if their hand-off had timed out
  int lstm_persistent = 1;              // one launch per BiLSTM layer where supported (W_hh resident on chip; vog_hip.h)
  int fused_tail = 1;                   // Wo..LN2 (+ lin2 + score) of an encoder layer as one launch where supported
  // device weights
  float* emb = nullptr;
  unsigned short* emb16 = nullptr;                      // 16-bit copy: A operand when the LSTM input GEMM has M > 64
  std::vector<unsigned short*> wih;                     // [layer] [8R, in]
  std::vector<unsigned short*> wih_f;                   // same, fragment order (M <= 64 kernel)
  unsigned short* w_outproj_f = nullptr;
  std::vector<unsigned short*> whh;                     // [layer] [2][4R][R]
  std::vector<unsigned short*> wih_p;                   // [layer] W_ih in the W_hh tile order (fused input projection), or null
  int fused_ih = 1;                     // LSTM input projections inside the persistent layer kernel where supported (no GEMM
                                        // launches, no gx round trip; W_ih streams through the layer's 64 CUs: +7 / +17 us per
                                        // layer against 6.4 / 9.7 us whole-chip launches -> 5 % more throughput with 4 batches
                                        // in flight, 10 us more single-batch latency)
  std::vector<float*> bsum;                             // [layer] [8R]
  float* gx0_tab = nullptr;             // round 6: [vocab + 1][2][R][4 gates] fp32 = emb . W_ih_l0^T + b_ih + b_hh for every token and both directions
                                        // (vog_lstm_layer_args.gx_table): layer 0's gate inputs are a look-up, built once per checkpoint
  unsigned short *w_outproj = nullptr, *w_prop = nullptr, *w_seg = nullptr, *w_lin2 = nullptr;
  unsigned short* w_lin2_p = nullptr;                   // lin2.0 in 32x16 fragment order (fused score head)
  unsigned short *w_prop_f = nullptr, *w_seg_f = nullptr;   // encoder weights in 16x32 fragment order (visenc.hip)
  int fused_enc = 1;                    // both feature encoders + concat as one launch where supported
  int enc_lean = -1;                    // -1: lean form exactly when the encoders share a BiLSTM layer's launch
  int pair_launches = 1;                // step i of the language chain shares a launch with step i of the visual chain (pair.hip)
  int pair_mask = 15;                   // which pairs are formed: 1 BiLSTM layer 0 + encoders, 2 layer 1 + obj tail, 4 out-projection + mul QKV,
                                        // 8 layer-1 input projection (where it is a GEMM launch) + obj QKV
  float *b_outproj = nullptr, *b_prop = nullptr, *b_seg = nullptr, *b_lin2 = nullptr;
  float *w_arg = nullptr, *b_arg = nullptr, *w_lin2b = nullptr, *b_lin2b = nullptr;
  float *w_sv0 = nullptr, *b_sv0 = nullptr, *w_sv2 = nullptr, *b_sv2 = nullptr;
  TxWeights obj, mul;
};

namespace vog {

static bool has_obj(const vog_model_desc& d) {
  return d.mdl_kind == VOG_MDL_VGRND || (d.mdl_kind == VOG_MDL_VOG && d.obj_to_use);
}
static bool has_obj_weights(const vog_model_desc& d) { return d.mdl_kind != VOG_MDL_IGRND; }
static bool has_mul(const vog_model_desc& d) { return d.mdl_kind == VOG_MDL_VOG; }

static void add_w(vog_ctx* c, const std::string& n, int64_t numel) {
  c->names.push_back(n);
  c->numel[n] = numel;
}

static void declare_tx(vog_ctx* c, const char* prefix, int d, int n_layers) {
  const int dh = d / 2;
  for (int l = 0; l < n_layers; ++l) {
    std::string p = std::string(prefix) + ".encoder.layers." + std::to_string(l);
    for (const char* w : {"wq", "wk", "wv", "wo"}) add_w(c, p + ".selfattn.layer." + w + ".weight", (int64_t)d * d);
    add_w(c, p + ".selfattn.layernorm.weight", d);
    add_w(c, p + ".selfattn.layernorm.bias", d);
    add_w(c, p + ".feedforward.layer.linear1.weight", (int64_t)dh * d);
    add_w(c, p + ".feedforward.layer.linear1.bias", dh);
    add_w(c, p + ".feedforward.layer.linear2.weight", (int64_t)d * dh);
    add_w(c, p + ".feedforward.layer.linear2.bias", d);
    add_w(c, p + ".feedforward.layernorm.weight", d);
    add_w(c, p + ".feedforward.layernorm.bias", d);
  }
}

template <typename T>
static int upload(vog_ctx* c, const std::vector<T>& h, T** out) {
  void* p = nullptr;
  VOG_HIP(hipMalloc(&p, h.size() * sizeof(T) + 64));
  c->allocs.push_back(p);
  VOG_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)p;
  return 0;
}

static const std::vector<float>& W(vog_ctx* c, const std::string& n) { return c->host.at(n); }

// w - t16(w): what the second operand of a hi + lo product carries (exact in fp32)
static std::vector<float> remainder16(const float* w, size_t n, int dt) {
  std::vector<float> r(n);
  for (size_t i = 0; i < n; ++i) {
    const unsigned short h = h_to16(w[i], dt);
    float back;
    if (dt == VOG_BF16) { unsigned int u = (unsigned int)h << 16; memcpy(&back, &u, 4); }
    else { _Float16 f; memcpy(&f, &h, 2); back = (float)f; }
    r[i] = w[i] - back;
  }
  return r;
}

static int up32(vog_ctx* c, const std::string& n, float** out) { return upload<float>(c, W(c, n), out); }

static int up16(vog_ctx* c, const std::string& n, int dt, unsigned short** out) {
  const auto& w = W(c, n);
  std::vector<unsigned short> h(w.size());
  for (size_t i = 0; i < w.size(); ++i) h[i] = h_to16(w[i], dt);
  return upload<unsigned short>(c, h, out);
}

// fp32 [N, ld] host matrix -> 32x16 fragment order on the device (vog_pack_w_frag32)
static int up_frag32(vog_ctx* c, const float* w, int64_t ld, int N, int K, int dt, unsigned short** out) {
  std::vector<unsigned short> h((size_t)N * K);
  VOG_TRY(vog_pack_w_frag32(w, ld, N, K, h.data(), (vog_dtype)dt));
  return upload<unsigned short>(c, h, out);
}

static int finalize_tx(vog_ctx* c, const char* prefix, const char* pe_name, int d, int H,
                       int n_layers, int use_rel, TxWeights* tw) {
  const int dt = c->d.tx_dtype;
  tw->d = d; tw->H = H; tw->n_layers = n_layers; tw->use_rel = use_rel;
  const int chunk = (d + H - 1) / H;                     // torch.chunk: ceil(d/H), last shorter
  int off = 0;
  for (int h = 0; h < H; ++h) {
    const int sz = std::min(chunk, d - off);
    if (sz <= 0) VOG_FAIL(-1, "d_model %d cannot be chunked into %d heads", d, H);
    tw->head_off.push_back(off);
    tw->head_dim.push_back(sz);
    off += sz;
  }
  tw->dp = attn_head_pad(chunk);
  if (tw->dp < 0) VOG_FAIL(-1, "head dim %d > 256 unsupported", chunk);
  const int dp = tw->dp, dh = d / 2;
  tw->dh = dh;
  for (int l = 0; l < n_layers; ++l) {
    std::string p = std::string(prefix) + ".encoder.layers." + std::to_string(l);
    TxLayer L{};
    std::vector<unsigned short> wqkv((size_t)3 * H * dp * d, 0);
    const char* nm[3] = {"wq", "wk", "wv"};
    for (int which = 0; which < 3; ++which) {
      const auto& w = W(c, p + ".selfattn.layer." + nm[which] + ".weight");
      for (int h = 0; h < H; ++h)
        for (int dd = 0; dd < tw->head_dim[h]; ++dd) {
          const float* src = &w[(size_t)(tw->head_off[h] + dd) * d];
          unsigned short* dst = &wqkv[((size_t)(which * H + h) * dp + dd) * d];
          for (int k = 0; k < d; ++k) dst[k] = h_to16(src[k], dt);
        }
    }
    VOG_TRY(upload<unsigned short>(c, wqkv, &L.wqkv));
    L.wqkv_lang_f = nullptr;
    L.wqkv_p = L.wqkv_pv = nullptr;
    L.wqkv_lo = L.wqkv_lang_f_lo = L.wo_p_lo = L.w1_p_lo = L.w2_p_lo = nullptr;
    {
      std::vector<float> wqf((size_t)3 * H * dp * d, 0.f);
      for (int which = 0; which < 3; ++which) {
        const auto& w = W(c, p + ".selfattn.layer." + nm[which] + ".weight");
        for (int h = 0; h < H; ++h)
          for (int dd = 0; dd < tw->head_dim[h]; ++dd)
            memcpy(&wqf[((size_t)(which * H + h) * dp + dd) * d], &w[(size_t)(tw->head_off[h] + dd) * d], d * sizeof(float));
      }
      if (c->tx_split) {     // remainder of the padded [3*H*dp, d] weights, same plain layout
        const std::vector<float> r = remainder16(wqf.data(), wqf.size(), dt);
        std::vector<unsigned short> lo(r.size());
        for (size_t i = 0; i < r.size(); ++i) lo[i] = h_to16(r[i], dt);
        VOG_TRY(upload<unsigned short>(c, lo, &L.wqkv_lo));
      }
      if (vog_qkv_rowblock_supported(3 * H * dp, d)) VOG_TRY(up_frag32(c, wqf.data(), d, 3 * H * dp, d, dt, &L.wqkv_p));
      const int dv = d - c->d.lang_enc;
      if (l == 0 && std::string(prefix) == "mult_txf" && dv > 0 && vog_qkv_rowblock_supported(3 * H * dp, dv))
        VOG_TRY(up_frag32(c, wqf.data(), d, 3 * H * dp, dv, dt, &L.wqkv_pv));
    }
    {
      const int dl = c->d.lang_enc, dv = d - dl;
      if (l == 0 && std::string(prefix) == "mult_txf" && dv > 0 && dl % 32 == 0) {
        // fp32 image of the padded lang columns, then fragment order
        std::vector<float> wl((size_t)3 * H * dp * dl, 0.f);
        for (int which = 0; which < 3; ++which) {
          const auto& w = W(c, p + ".selfattn.layer." + nm[which] + ".weight");
          for (int h = 0; h < H; ++h)
            for (int dd = 0; dd < tw->head_dim[h]; ++dd)
              memcpy(&wl[((size_t)(which * H + h) * dp + dd) * dl], &w[(size_t)(tw->head_off[h] + dd) * d + dv],
                     dl * sizeof(float));
        }
        std::vector<unsigned short> wf(wl.size());
        VOG_TRY(vog_pack_w_frag(wl.data(), dl, 3 * H * dp, dl, wf.data(), (vog_dtype)dt));
        VOG_TRY(upload<unsigned short>(c, wf, &L.wqkv_lang_f));
        if (c->tx_split) {
          const std::vector<float> r = remainder16(wl.data(), wl.size(), dt);
          VOG_TRY(vog_pack_w_frag(r.data(), dl, 3 * H * dp, dl, wf.data(), (vog_dtype)dt));
          VOG_TRY(upload<unsigned short>(c, wf, &L.wqkv_lang_f_lo));
        }
      }
    }
    std::vector<unsigned short> wo((size_t)d * H * dp, 0);
    {
      const auto& w = W(c, p + ".selfattn.layer.wo.weight");
      for (int o = 0; o < d; ++o)
        for (int h = 0; h < H; ++h)
          for (int dd = 0; dd < tw->head_dim[h]; ++dd)
            wo[(size_t)o * H * dp + (size_t)h * dp + dd] = h_to16(w[(size_t)o * d + tw->head_off[h] + dd], dt);
    }
    VOG_TRY(upload<unsigned short>(c, wo, &L.wo));
    L.wo_p = L.w1_p = L.w2_p = nullptr;
    if (tx_tail_supported(d, dh, H * dp)) {
      // the padded Wo again as fp32 (zero columns for the head padding), then fragment order
      std::vector<float> wof((size_t)d * H * dp, 0.f);
      const auto& w = W(c, p + ".selfattn.layer.wo.weight");
      for (int o = 0; o < d; ++o)
        for (int h = 0; h < H; ++h)
          for (int dd = 0; dd < tw->head_dim[h]; ++dd)
            wof[(size_t)o * H * dp + (size_t)h * dp + dd] = w[(size_t)o * d + tw->head_off[h] + dd];
      VOG_TRY(up_frag32(c, wof.data(), (int64_t)H * dp, d, H * dp, dt, &L.wo_p));
      VOG_TRY(up_frag32(c, W(c, p + ".feedforward.layer.linear1.weight").data(), d, dh, d, dt, &L.w1_p));
      VOG_TRY(up_frag32(c, W(c, p + ".feedforward.layer.linear2.weight").data(), dh, d, dh, dt, &L.w2_p));
      if (c->tx_split) {
        const auto& w1 = W(c, p + ".feedforward.layer.linear1.weight");
        const auto& w2 = W(c, p + ".feedforward.layer.linear2.weight");
        VOG_TRY(up_frag32(c, remainder16(wof.data(), wof.size(), dt).data(), (int64_t)H * dp, d, H * dp, dt, &L.wo_p_lo));
        VOG_TRY(up_frag32(c, remainder16(w1.data(), w1.size(), dt).data(), d, dh, d, dt, &L.w1_p_lo));
        VOG_TRY(up_frag32(c, remainder16(w2.data(), w2.size(), dt).data(), dh, d, dh, dt, &L.w2_p_lo));
      }
    }
    VOG_TRY(up16(c, p + ".feedforward.layer.linear1.weight", dt, &L.w1));
    VOG_TRY(up16(c, p + ".feedforward.layer.linear2.weight", dt, &L.w2));
    VOG_TRY(up32(c, p + ".feedforward.layer.linear1.bias", &L.b1));
    VOG_TRY(up32(c, p + ".feedforward.layer.linear2.bias", &L.b2));
    VOG_TRY(up32(c, p + ".selfattn.layernorm.weight", &L.ln1g));
    VOG_TRY(up32(c, p + ".selfattn.layernorm.bias", &L.ln1b));
    VOG_TRY(up32(c, p + ".feedforward.layernorm.weight", &L.ln2g));
    VOG_TRY(up32(c, p + ".feedforward.layernorm.bias", &L.ln2b));
    tw->layers.push_back(L);
  }
  VOG_TRY(up32(c, std::string(pe_name) + ".weight", &tw->pe_w));
  VOG_TRY(up32(c, std::string(pe_name) + ".bias", &tw->pe_b));
  return 0;
}

// ---- geometry + workspace plan ---------------------------------------------------
struct Geo {
  int B, ncmp, T, sep, nvl, nc_v, nfrm, nppf, NP, Fv, n_vid, Bn, Bn16, R, E, L, d_obj, d_mul;
  int S_obj, N_obj, spv_obj, npad_obj; float fdiv_obj;
  int S_mul, N_mul, npad_mul;
  int64_t rows_obj, rows_mul;
};

static Geo make_geo(const vog_model_desc& d, int B, int ncmp, int T) {
  Geo g{};
  g.B = B; g.ncmp = ncmp; g.T = T;
  g.sep = d.conc_type == VOG_CONC_SEP;
  g.nvl = g.sep ? ncmp : 1;
  g.nc_v = g.sep ? ncmp : 1;
  g.nfrm = d.conc_type == VOG_CONC_TEMP ? ncmp * d.nfrm0 : d.nfrm0;
  g.nppf = d.conc_type == VOG_CONC_SPAT ? ncmp * d.nppf0 : d.nppf0;
  g.NP = g.nfrm * g.nppf;
  g.Fv = g.NP / d.nppf0;
  g.n_vid = B * g.nc_v;
  g.Bn = B * g.nvl;
  g.Bn16 = (int)round_up64(g.Bn, 16);
  g.R = d.rnn_size; g.E = d.emb_dim; g.L = d.lang_enc;
  g.d_obj = d.prop_enc + d.seg_enc;
  g.d_mul = g.d_obj + d.lang_enc;
  g.S_obj = d.obj_one_frm ? g.n_vid * g.nfrm : g.n_vid;
  g.N_obj = d.obj_one_frm ? g.nppf : g.NP;
  g.spv_obj = d.obj_one_frm ? g.nfrm : 1;
  g.fdiv_obj = d.obj_one_frm ? (float)g.nfrm : 1.0f;
  g.npad_obj = (int)round_up64(g.N_obj, 32);
  g.S_mul = g.n_vid * g.nfrm;
  g.N_mul = d.nsrl * g.nppf;
  g.npad_mul = (int)round_up64(g.N_mul, 32);
  g.rows_obj = (int64_t)g.n_vid * g.NP;
  g.rows_mul = (int64_t)g.S_mul * g.N_mul;
  return g;
}

struct Plan {
  std::map<std::string, std::pair<int64_t, int64_t>> buf;   // name -> (offset, bytes)
  int64_t total = 0;
  int64_t zero_off = 0, zero_bytes = 0, ones_off = 0, ones_bytes = 0;
  int64_t add(const std::string& n, int64_t bytes) {
    const int64_t off = total;
    buf[n] = {off, bytes};
    total += round_up64(bytes, 256);
    return off;
  }
};

static Plan make_plan(const vog_ctx* c, const Geo& g, bool lang_only = false) {
  const vog_model_desc& d = c->d;
  Plan p;
  const int nl = d.rnn_layers;
  p.add("logit_pub", 256);                            // the logit maxima this workspace's forwards last published to the host
                                                      // (vog_pred_args.published; zeroed with the workspace, not per forward)
  p.add("logit_max", 2 * 4 * VOG_LOGIT_WORDS * VOG_LOGIT_STRIDE * 4);   // [2 stacks][4 layers][words, 128 B apart]: the running largest
                                                      // |attention logit| of this workspace's forwards (float bits; raise-only, zeroed
                                                      // with the workspace)
  // ---- zero-initialised region (one memset per forward)
  p.zero_off = p.total;
  for (int l = 0; l < nl; ++l) {
    // out16 of layer l is followed directly by the h buffer that holds the FINAL
    // state, so [out16 ; h_final] is one contiguous A operand for the out-proj GEMM
    // (hA lives inside the same allocation: rows [Bn*T, Bn*T + Bn16) )
    p.add("lstm_out16_" + std::to_string(l), (int64_t)(round_up64(g.Bn * g.T + g.Bn, 16) + g.Bn16) * 2 * g.R * 2);
    p.add("lstm_hB_" + std::to_string(l), (int64_t)g.Bn16 * 2 * g.R * 2);
    p.add("lstm_hA2_" + std::to_string(l), (int64_t)g.Bn16 * 2 * g.R * 2);   // ping buffer when out16 is fragment-ordered
    p.add("lstm_c_" + std::to_string(l), (int64_t)g.Bn16 * 2 * g.R * 4);
    p.add("lstm_sync_" + std::to_string(l), 1024);   // [2] timeout, [16 + 16 dir + xcc] workgroups arrived per XCC id
  }
  p.add("obj_guard", 256);                            // vog_attn_args.guard_flag of the two stacks (long-sequence attention): zeroed
  p.add("mul_guard", 256);                            // with the rest of this region, so the attention needs no clearing launch
  p.add("mul_ef_guard", 256);                         // vog_attn_struct_args.guard_flag (E x F attention of mul_tx layer 0, p100)
  p.zero_bytes = p.total - p.zero_off;
  // ---- 0xff-initialised region, directly behind the zeros (same fill loop of the prologue): the
  // hand-off slots of the persistent BiLSTM, [T][2][Bn][R] 16-bit per layer (lstm_dev.h)
  p.ones_off = p.total;
  for (int l = 0; l < nl; ++l) p.add("lstm_hx_" + std::to_string(l), vog_bilstm_hx_bytes(g.Bn, g.T, g.R));
  p.ones_bytes = p.total - p.ones_off;
  p.add("emb_a0", round_up64(g.Bn * g.T, 16) * g.E * 2);   // layer-0 A operand in fragment order (M <= 64)
  p.add("tok", (int64_t)g.Bn * g.T * 4);
  p.add("lstm_rows", (int64_t)2 * g.Bn * g.T * 4);
  p.add("gx", (int64_t)g.Bn * g.T * 8 * g.R * 4);
  p.add("full", (int64_t)(g.Bn * g.T + g.Bn16) * g.L * 4);
  p.add("lang", (int64_t)g.Bn * d.nsrl * g.L * 4);
  p.add("outproj_slabs", (int64_t)8 * (g.Bn * g.T + g.Bn) * g.L * 4);
  if (lang_only) return p;
  p.add("prop16", g.rows_obj * d.prop_dim * 2);
  p.add("seg16", (int64_t)g.n_vid * g.Fv * d.seg_dim * 2);
  p.add("enc_slabs", (int64_t)16 * (g.rows_obj * d.prop_enc + (int64_t)g.n_vid * g.Fv * d.seg_enc) * 4);
  p.add("prop_seg", g.rows_obj * g.d_obj * 4);
  p.add("prop_seg16", g.rows_obj * g.d_obj * 2);
  if (c->tx_split) p.add("prop_seg16_lo", g.rows_obj * g.d_obj * 2);
  auto tx = [&](const char* nm, const TxWeights& tw, int64_t rows, int S, int npad) {
    const std::string n(nm);
    if (c->tx_split) {     // 16-bit remainders of Q / K, of the attention rows and of the layer outputs (hi + lo operands)
      p.add(n + "_q_lo", (int64_t)S * tw.H * tw.dp * npad * 2);
      p.add(n + "_k_lo", (int64_t)S * tw.H * tw.dp * npad * 2);
      p.add(n + "_attn16_lo", rows * tw.H * tw.dp * 2);
      p.add(n + "_outA16_lo", rows * tw.d * 2);
      if (tw.n_layers > 1) p.add(n + "_outB16_lo", rows * tw.d * 2);
    }
    p.add(n + "_u", g.rows_obj * tw.H * 4);
    p.add(n + "_q", (int64_t)S * tw.H * tw.dp * npad * 2);      // fragment order, npad = N up to 32
    p.add(n + "_k", (int64_t)S * tw.H * tw.dp * npad * 2);
    p.add(n + "_vt", (int64_t)S * tw.H * tw.dp * npad * 2);
    p.add(n + "_attn16", rows * tw.H * tw.dp * 2);
    p.add(n + "_tmp", rows * tw.d * 4);
    p.add(n + "_x1", rows * tw.d * 4);
    p.add(n + "_x1_16", rows * tw.d * 2);
    p.add(n + "_ffn16", rows * tw.dh * 2);
    p.add(n + "_outA", rows * tw.d * 4);
    p.add(n + "_outA16", rows * tw.d * 2);
    p.add(n + "_x1s", tx_tail_scratch_bytes((int)rows, tw.d) + 16);
    if (tw.n_layers > 1) {
      p.add(n + "_outB", rows * tw.d * 4);
      p.add(n + "_outB16", rows * tw.d * 2);
    }
  };
  if (has_obj(d)) tx("obj", c->obj, g.rows_obj, g.S_obj, g.npad_obj);
  p.add("xmul", g.rows_mul * g.d_mul * 4);
  p.add("xmul16", g.rows_mul * g.d_mul * 2);
  if (has_mul(d)) {
    tx("mul", c->mul, g.rows_mul, g.S_mul, g.npad_mul);
    p.add("mul_pv", g.rows_obj * 3 * c->mul.H * c->mul.dp * 4);
    p.add("mul_pl", (int64_t)g.Bn * d.nsrl * 3 * c->mul.H * c->mul.dp * 4);
  }
  p.add("h1", g.rows_mul * 256 * 4);
  return p;
}

// branch 0 = visual + joint path (caller's stream), 1 = language path (captured as a
// parallel branch of the graph), -1 = join marker: everything after it needs both.
struct Step {
  std::string name;
  std::function<int(hipStream_t)> fn;
  int branch = 0;
};

struct WS {
  char* base; const Plan* plan;
  template <typename T> T* at(const std::string& n) const {
    return reinterpret_cast<T*>(base + plan->buf.at(n).first);
  }
};

static void tx_steps(const vog_ctx* c, const TxWeights& tw, const char* nm, const Geo& g, const WS& ws,
                     const vog_batch* b, const float* x_in32, const void* x_in16, int S, int N,
                     int npad, int spv, int n_box, float fdiv, int last_dt, std::vector<Step>& steps,
                     const float** out32, const void** out16,
                     const vog_vislang_args* structured = nullptr, const void* vis16 = nullptr,
                     bool last_needs_f32 = true, const vog_score_args* score = nullptr,
                     // hi + lo operands (c->tx_split): remainders of the stack's input rows / of the visual rows (structured
                     // layer 0); out_feeds_attn: the LAST layer's output is another attention layer's input (obj_tx under mul_tx);
                     // *out16_lo: remainder of the 16-bit output copy
                     const void* x_in16_lo = nullptr, const void* vis16_lo = nullptr, bool out_feeds_attn = false,
                     const void** out16_lo = nullptr, int* err = nullptr) {
  const std::string n(nm);
  const vog_model_desc& d = c->d;
  const vog_dtype dt = (vog_dtype)d.tx_dtype;
  const int64_t rows = (int64_t)S * N;
  float* u = ws.at<float>(n + "_u");
  (void)fdiv;   // the bias precursors u are produced by the fused visual prologue (vis_prep)
  const float* cur32 = x_in32;
  const void* cur16 = x_in16;
  const bool split = c->tx_split != 0;
  const void* cur16_lo = x_in16_lo;
  constexpr int kLogitGroup = VOG_LOGIT_WORDS * VOG_LOGIT_STRIDE;                // words of one layer's report
  unsigned int* lmax = ws.at<unsigned int>("logit_max") + (n == "mul" ? 4 * kLogitGroup : 0);
  for (int l = 0; l < tw.n_layers; ++l) {
    const TxLayer& L = tw.layers[l];
    const bool toA = (l % 2) == 0;
    float* o32 = ws.at<float>(n + (toA ? "_outA" : "_outB"));
    void* o16 = ws.at<void>(n + (toA ? "_outA16" : "_outB16"));
    vog_qkv_args qa{};
    qa.x16 = cur16; qa.ldx = tw.d; qa.wqkv = L.wqkv; qa.ldw = tw.d;
    qa.q = ws.at<void>(n + "_q"); qa.k = ws.at<void>(n + "_k"); qa.vt = ws.at<void>(n + "_vt");
    qa.S = S; qa.N = N; qa.H = tw.H; qa.dp = tw.dp; qa.npad = npad; qa.K = tw.d; qa.dtype = dt;
    // many rows (p100): one workgroup per 64 rows walks all output columns (qkvrb_dev.h: +1.5 % at cfg 4); the tiled LDS-DMA GEMM
    // otherwise (a 64-row x 512-column row-block form measured -3 % at cfg 2 and was removed in round 6)
    const bool qkv_rb = g.rows_obj >= 8192;
    qa.wqkv_p32 = qkv_rb ? L.wqkv_p : nullptr;
    const bool fact = structured && l == 0;
    const bool last_l = l == tw.n_layers - 1;
    // hi + lo tail: this layer's output is read by another attention layer (a later layer of the stack, or mul_tx behind obj_tx)
    const bool tail_split = split && (!last_l || out_feeds_attn);
    void* attn16_lo = split ? ws.at<void>(n + "_attn16_lo") : nullptr;
    void* o16_lo = split ? ws.at<void>(n + (toA ? "_outA16_lo" : "_outB16_lo")) : nullptr;
    if (split) {
      qa.x16_lo = cur16_lo; qa.wqkv_lo = L.wqkv_lo; qa.q_lo = ws.at<void>(n + "_q_lo"); qa.k_lo = ws.at<void>(n + "_k_lo");
      qa.wqkv_p32 = nullptr;
      if (!qa.x16_lo || !qa.wqkv_lo) { if (err) *err = -1; set_error("hi + lo operands: layer %d of %s has no remainder rows", l, nm); return; }
    }
    if (fact) {
      // layer 0 of mul_tx: tokens are [vis[p] || lang[a]] -> project the two parts once each
      const vog_vislang_args sv = *structured;
      vog_qkv_args qs = qa;
      qs.x16 = vis16; qs.ldx = sv.dv; qs.K = sv.dv;          // visual rows x first d_vis weight columns
      // nothing is fanned out: q, k, v fragments of the nppf VISUAL tokens of every sequence (a plain
      // QKV projection of the visual rows); the separable attention below adds the language parts
      const float* plang = ws.at<float>(n + "_pl");
      const int npad_kv = (int)round_up64(sv.nppf, 32);
      qs.N = sv.nppf; qs.npad = npad_kv;
      qs.wqkv_p32 = (qkv_rb && !split) ? L.wqkv_pv : nullptr;
      if (split) qs.x16_lo = vis16_lo;
      steps.push_back({n + "_pv", [=](hipStream_t st) { return vog_qkv_proj(&qs, st); }});
      vog_attn_struct_args sa{};
      sa.q_visual = 1;
      sa.q = qa.q; sa.kv = qa.k; sa.vv = qa.vt; sa.pl = plang; sa.out16 = ws.at<void>(n + "_attn16");
      sa.u = u; sa.pe_b = tw.pe_b; sa.S = S; sa.H = tw.H; sa.dp = tw.dp; sa.nsrl = sv.nsrl; sa.nppf = sv.nppf;
      sa.npad_q = npad; sa.npad_kv = npad_kv; sa.nfrm = sv.nfrm; sa.lang_per_vid = sv.lang_per_vid;
      sa.nc_v = sv.nc_v; sa.use_rel = tw.use_rel; sa.seq_per_vid = spv; sa.NP = g.NP;
      sa.inv_scale = 1.0f / sqrtf((float)tw.d); sa.dtype = dt;
      sa.guard_flag = (n == "mul") ? ws.at<int>("mul_ef_guard") : nullptr;
      sa.logit_max = lmax + (l < 3 ? l : 3) * kLogitGroup;
      if (split) { sa.q_lo = qa.q_lo; sa.kv_lo = qa.k_lo; sa.out16_lo = tail_split ? attn16_lo : nullptr; }
      steps.push_back({n + "_attn", [=](hipStream_t st) { return vog_rel_attention_struct_fwd(&sa, st); }});
    } else {
      steps.push_back({n + "_qkv", [=](hipStream_t st) { return vog_qkv_proj(&qa, st); }});
    }
    vog_attn_args aa{};
    aa.q = qa.q; aa.k = qa.k; aa.vt = qa.vt; aa.out16 = ws.at<void>(n + "_attn16");
    aa.u = u; aa.pe_b = tw.pe_b; aa.S = S; aa.N = N; aa.H = tw.H; aa.dp = tw.dp; aa.npad = npad;
    aa.use_rel = tw.use_rel; aa.n_box = n_box; aa.seq_per_vid = spv; aa.NP = g.NP;
    aa.inv_scale = 1.0f / sqrtf((float)tw.d); aa.dtype = dt;
    // one guard word PER LAYER inside the stack's zero-filled 256 bytes (with one word per stack a flag raised by
    // layer 0 stayed up - the prologue clears once per forward - and every later layer re-ran the running-maximum fallback)
    aa.guard_flag = ws.at<int>(n + "_guard") + (l < 63 ? l : 63);
    aa.guard_precleared = l < 63 ? 1 : 0;
    aa.logit_max = lmax + (l < 3 ? l : 3) * kLogitGroup;   // (layers past the 4th share the last group)
    if (split) { aa.q_lo = qa.q_lo; aa.k_lo = qa.k_lo; aa.out16_lo = tail_split ? attn16_lo : nullptr; }
    if (!fact) steps.push_back({n + "_attn", [=](hipStream_t st) { return vog_rel_attention_fwd(&aa, st); }});
    const bool last = l == tw.n_layers - 1;
    // 16-bit copy of the LAST layer's output: typed for its consumer (none for obj_tx,
    // the f16 score head for mul_tx)
    void* o16w = (last && last_dt < 0) ? nullptr : o16;
    const vog_dtype odt = last && last_dt >= 0 ? (vog_dtype)last_dt : dt;
    // the last layer's fp32 output is written only if somebody reads it (mul_tx: the score
    // head takes the 16-bit copy -> 12 MB less HBM traffic per forward at cfg 2)
    float* o32w = (last && !last_needs_f32 && o16w) ? nullptr : o32;
    if (c->fused_tail && L.wo_p && tx_tail_supported(tw.d, tw.dh, tw.H * tw.dp)) {
      // everything after the attention is row-local: one launch (txtail.hip); with `score` the
      // last layer also runs lin2 + the score head and its output never leaves the chip
      vog_tx_tail_args ta{};
      ta.attn16 = aa.out16; ta.kwo = tw.H * tw.dp; ta.wo_p = L.wo_p; ta.w1_p = L.w1_p; ta.w2_p = L.w2_p;
      ta.residual = fact ? nullptr : cur32; ta.ldr = tw.d;
      ta.ln1g = L.ln1g; ta.ln1b = L.ln1b; ta.b1 = L.b1; ta.b2 = L.b2; ta.ln2g = L.ln2g; ta.ln2b = L.ln2b;
      ta.y32 = o32w; ta.y16 = o16w; ta.y16_dtype = (int)odt;
      ta.x1_scratch = ws.at<float>(n + "_x1s");
      ta.M = (int)rows; ta.d = tw.d; ta.dh = tw.dh; ta.dtype = dt; ta.head_dtype = d.enc_dtype;
      if (tail_split) {
        ta.attn16_lo = attn16_lo; ta.wo_p_lo = L.wo_p_lo; ta.w1_p_lo = L.w1_p_lo; ta.w2_p_lo = L.w2_p_lo;
        ta.y16_lo = o16w ? o16_lo : nullptr;
      }
      const bool with_score = last && score && c->w_lin2_p;
      vog_score_args sc{};
      if (with_score) { sc = *score; ta.wl_p = c->w_lin2_p; ta.bl = c->b_lin2; ta.y32 = nullptr; ta.y16 = nullptr; }
      vog_vislang_args sv{};
      if (fact) sv = *structured;
      steps.push_back({n + "_tail", [=](hipStream_t st) {
        vog_tx_tail_args t2 = ta;
        if (fact) t2.res_vislang = &sv;
        if (with_score) t2.score = &sc;
        return vog_tx_tail_fwd(&t2, st); }});
      if (with_score) *out16 = nullptr;            // tells the caller that lin2 + score already ran
      cur32 = o32;
      cur16 = o16;
      cur16_lo = tail_split ? o16_lo : nullptr;
      if (last) { *out32 = cur32; if (!with_score) *out16 = cur16; if (out16_lo) *out16_lo = cur16_lo; return; }
      continue;
    }
    if (split) { if (err) *err = -1; set_error("hi + lo operands need the fused encoder-layer tail (d = 512 / 768; stack %s)", nm); return; }
    vog_gemm_args wo{}; wo.c16_dtype = -1;
    wo.a = aa.out16; wo.a_is_f32 = 0; wo.lda = (int64_t)tw.H * tw.dp; wo.w = L.wo; wo.ldw = (int64_t)tw.H * tw.dp;
    wo.residual = cur32; wo.ldr = tw.d; wo.c32 = ws.at<float>(n + "_tmp"); wo.ldc = tw.d;
    wo.M = (int)rows; wo.N = tw.d; wo.K = tw.H * tw.dp; wo.rep = 1; wo.dtype = dt;
    if (fact) {
      const vog_vislang_args sv = *structured;       // residual = the (unmaterialised) token matrix
      wo.residual = nullptr;
      steps.push_back({n + "_wo", [=](hipStream_t st) {
        vog_gemm_args w2 = wo; w2.res_vislang = &sv; return vog_gemm_bias_act(&w2, st); }});
    } else {
      steps.push_back({n + "_wo", [=](hipStream_t st) { return vog_gemm_bias_act(&wo, st); }});
    }
    float* x1 = ws.at<float>(n + "_x1");
    void* x1_16 = ws.at<void>(n + "_x1_16");
    float* tmp = wo.c32;
    const int d_ = tw.d;
    steps.push_back({n + "_ln1", [=](hipStream_t st) {
      return vog_residual_layernorm(tmp, L.ln1g, L.ln1b, x1, x1_16, (int)rows, d_, dt, st); }});
    vog_gemm_args f1{}; f1.c16_dtype = -1;
    f1.a = x1_16; f1.lda = tw.d; f1.w = L.w1; f1.ldw = tw.d; f1.bias = L.b1; f1.relu = 1;
    f1.c16 = ws.at<void>(n + "_ffn16"); f1.ldc16 = tw.dh; f1.M = (int)rows; f1.N = tw.dh; f1.K = tw.d;
    f1.rep = 1; f1.dtype = dt;
    steps.push_back({n + "_ffn1", [=](hipStream_t st) { return vog_gemm_bias_act(&f1, st); }});
    vog_gemm_args f2{}; f2.c16_dtype = -1;
    f2.a = f1.c16; f2.lda = tw.dh; f2.w = L.w2; f2.ldw = tw.dh; f2.bias = L.b2; f2.residual = x1;
    f2.ldr = tw.d; f2.c32 = tmp; f2.ldc = tw.d; f2.M = (int)rows; f2.N = tw.d; f2.K = tw.dh; f2.rep = 1;
    f2.dtype = dt;
    steps.push_back({n + "_ffn2", [=](hipStream_t st) { return vog_gemm_bias_act(&f2, st); }});
    steps.push_back({n + "_ln2", [=](hipStream_t st) {
      return vog_residual_layernorm(tmp, L.ln2g, L.ln2b, o32w, o16w, (int)rows, d_, odt, st); }});
    cur32 = o32;
    cur16 = o16;
  }
  *out32 = cur32;
  *out16 = cur16;
}

// lang_only: just the language chain of `b` (group encoder, vog_lang_forward): only the language
// inputs of `b` are read and the workspace is the language-only plan.
// b->shared_lang != NULL: the language chain is NOT run; argument vectors (and the final hidden
// states for the sep head) come from a group encoder's workspace.
static int build_steps(const vog_ctx* c, const vog_batch* b, void* wsp, size_t ws_bytes,
                       Plan& plan, std::vector<Step>& steps, bool lang_only = false, bool allow_pairs = true) {
  const vog_model_desc& d = c->d;
  VOG_CHECK_ARG(c->finalized);
  VOG_CHECK_ARG(b && b->B > 0 && b->ncmp > 0 && b->T > 0 && b->T <= d.seq_len);
  const bool shared = !lang_only && b->shared_lang != nullptr;
  VOG_CHECK_ARG(b->srl_arg_inds_msk != nullptr);
  VOG_CHECK_ARG(shared || (b->srl_arg_words_ind && b->srl_arg_word_mask && b->srl_arg_word_mask_len &&
                           b->srl_arg_words_capture));
  VOG_CHECK_ARG(lang_only || (b->num_cmp_msk && b->pad_region_feature && b->seg_feature_for_frms &&
                              b->pad_proposals && b->mdl_outs && b->mdl_outs_eval));
  const Geo g = make_geo(d, b->B, b->ncmp, b->T);
  VOG_CHECK_ARG(lang_only || !g.sep || (b->verb_ind_in_srl && b->vidf_outs && b->fin_scores && b->fin_scores_loss));
  VOG_CHECK_ARG(!shared || !g.sep || b->shared_final_hidden);
  plan = make_plan(c, g, lang_only);
  if ((int64_t)ws_bytes < plan.total) VOG_FAIL(-2, "workspace too small: %zu < %lld", ws_bytes, (long long)plan.total);
  WS ws{(char*)wsp, &plan};
  const vog_dtype et = (vog_dtype)d.enc_dtype;
  const int R = g.R, T = g.T, Bn = g.Bn;

  // ---- language path (a14-a16): branch 1
  const size_t lang_begin = steps.size();
  const bool structured = has_mul(d) && (g.d_obj % 64) == 0 && (g.L % 32) == 0 && g.rows_obj > 64;
  float* const lang_vec = shared ? const_cast<float*>(b->shared_lang) : ws.at<float>("lang");
  // the language and the visual prologue are independent: one launch for both (not when the language
  // chain is captured as its own graph branch, and not in the group forms, where they live in
  // different programs)
  const bool fuse_prep = !shared && !lang_only;
  // one launch for both encoders + the concat, straight from the fp32 features (visenc.hip)
  const bool enc_fused = c->fused_enc && c->w_prop_f && c->w_seg_f && !lang_only;
  auto make_visprep = [&]() {
    vog_visprep_args vp{};
    if (!enc_fused) {
      vp.src0 = b->pad_region_feature; vp.dst0 = ws.at<void>("prop16"); vp.n0 = g.rows_obj * d.prop_dim;
      vp.src1 = b->seg_feature_for_frms; vp.dst1 = ws.at<void>("seg16"); vp.n1 = (int64_t)g.n_vid * g.Fv * d.seg_dim;
    }
    vp.dtype = et; vp.props = b->pad_proposals; vp.n_rows = (int)g.rows_obj; vp.vid_w = d.vid_w; vp.vid_h = d.vid_h;
    if (has_obj(d) && c->obj.use_rel) {
      vp.w_pe0 = c->obj.pe_w; vp.u0 = ws.at<float>("obj_u"); vp.H0 = c->obj.H; vp.nfrm_div0 = g.fdiv_obj;
    }
    if (has_mul(d) && c->mul.use_rel) {
      vp.w_pe1 = c->mul.pe_w; vp.u1 = ws.at<float>("mul_u"); vp.H1 = c->mul.H; vp.nfrm_div1 = (float)g.nfrm;
    }
    return vp;
  };
  if (!shared) {
    char* z = ws.base + plan.zero_off;
    const int64_t zb = plan.zero_bytes, ob = plan.ones_bytes;   // (the 0xff region starts where the zeros end)
    int32_t* tok = ws.at<int32_t>("tok");
    const int64_t *wi = b->srl_arg_words_ind, *wm = b->srl_arg_word_mask;
    const int nsrl = d.nsrl, sl = d.seq_len, V = d.vocab_size;
    float* gx = ws.at<float>("gx");
    int32_t* lrows = ws.at<int32_t>("lstm_rows");
    // layer-0 gate inputs read from the checkpoint's gate table by the layer kernel itself (no input projection for layer 0 at all)
    // Used where the in-kernel projection does not reach (Bn x T > 80 columns: cfg 3, cfg 5, grouped / batched requests): there it
    // replaces a GEMM launch and the gate round trip (cfg 3 +2.3 %, cfg 5 +4.7 ... 7 %). Where the layer kernel can project in its
    // prologue (cfg 2) the two forms measured EQUAL (63.3 vs 63.5 k queries/s: the table reads are not quite hidden, 29.8 vs 30.7 us
    // for the layer) and the prologue, whose operands are always cache-resident, stays. fused_ih = 5: the table wherever it exists.
    const bool fuse0_ok = c->lstm_persistent && vog_bilstm_layer_supported(Bn, R) && !c->wih_p.empty() && c->wih_p[0] &&
                          (g.E % 256) == 0 && Bn * T <= vog_bilstm_fused_cols() && c->emb16 && (g.E % 32) == 0;
    const bool use_tab = c->gx0_tab && ((c->fused_ih == 1 && !fuse0_ok) || c->fused_ih == 5) && c->lstm_persistent &&
                         vog_bilstm_layer_supported(Bn, R) && Bn * T <= 16 * 24;
    {
      const int64_t* lens = b->srl_arg_word_mask_len;
      const bool a0f = !use_tab && Bn * T <= vog_bilstm_fused_cols() && c->emb16 && (g.E % 32) == 0;
      const void* e16 = c->emb16;
      void* a0 = a0f ? ws.at<void>("emb_a0") : nullptr;
      const int E = g.E;
      if (fuse_prep) {
        const vog_visprep_args vp = make_visprep();
        steps.push_back({"prep", [=](hipStream_t st) {
          return vog_prep_fused(z, zb, ob, wi, wm, lens, tok, lrows, Bn, T, nsrl, sl, V, e16, a0, E, &vp, st); }});
      } else {
        steps.push_back({"lang_prep", [=](hipStream_t st) {
          return vog_lang_prep(z, zb, ob, wi, wm, lens, tok, lrows, Bn, T, nsrl, sl, V, e16, a0, E, st); }});
      }
    }
    // input projection inside the persistent layer kernel (no GEMM launch, no gx round trip): needs the layer input in fragment
    // order (embedding rows from lang_prep / the previous layer's out16). Round 6: up to 80 (sentence, position) columns (64
    // before: a bs = 4 batch whose longest sentence has 17-20 words fell back to the GEMM launches).
    // fused_ih: 1 = every layer in the kernel where its prologue reaches, layer 0 by table look-up where it does not; 4 = the same
    // without the table (GEMM launches beyond 80 columns: the default before round 6); 5 = the table wherever the checkpoint has
    // one; 2 = layer 0 only, 3 = layers >= 1 only (experiments), 0 = GEMM launches
    auto can_fuse_ih = [&](int l) {
      const int Kin = l == 0 ? g.E : 2 * R;
      if (l == 0 && use_tab) return false;
      return l < d.rnn_layers && (c->fused_ih == 1 || c->fused_ih == 4 || c->fused_ih == 5 || (c->fused_ih == 2 && l == 0) || (c->fused_ih == 3 && l > 0)) && c->lstm_persistent &&
             vog_bilstm_layer_supported(Bn, R) && c->wih_p[l] && (Kin % 256) == 0 && Bn * T <= vog_bilstm_fused_cols() &&
             (l > 0 || (c->emb16 && (g.E % 32) == 0));
    };
    // LSTM outputs that feed M <= 64 GEMMs (next layer's input projection, final projection) or the next layer's in-kernel
    // projection are written in A-fragment order: those consumers load contiguous fragments
    auto out_frag_of = [&](int l) {
      const bool small = (Bn * T + Bn) <= 64;
      return l == d.rnn_layers - 1 ? small : (small || can_fuse_ih(l + 1));
    };
    for (int l = 0; l < d.rnn_layers; ++l) {
      vog_gemm_args ga{}; ga.c16_dtype = -1;
      const bool ofrag = out_frag_of(l);
      const int Kin = l == 0 ? g.E : 2 * R;
      const bool ih_fused = can_fuse_ih(l) && (l == 0 || out_frag_of(l - 1));
      if (l == 0) {
        ga.a = c->emb; ga.a_is_f32 = 1; ga.lda = g.E; ga.a_rows = tok; ga.K = g.E;
        // M > 64 runs on the LDS-DMA kernel, which cannot convert in flight: same values, pre-rounded
        if (Bn * T > 64 && c->emb16 && (g.E % 64) == 0) { ga.a = c->emb16; ga.a_is_f32 = 0; }
        // M <= 64: lang_prep already gathered the rows, 16 bit, in fragment order
        if (Bn * T <= 64 && c->emb16 && (g.E % 32) == 0) {
          ga.a = ws.at<void>("emb_a0"); ga.a_is_f32 = 0; ga.a_rows = nullptr; ga.a_frag = 1;
        }
      }
      else {
        ga.a = ws.at<void>("lstm_out16_" + std::to_string(l - 1)); ga.lda = 2 * R; ga.K = 2 * R;
        ga.a_frag = out_frag_of(l - 1) ? 1 : 0;      // (only read when this layer's projection is NOT in its kernel: then M <= 64)
      }
      // output rows land in (direction, step) order: gxs[dir][step][b][4R]
      ga.w = c->wih[l]; ga.ldw = ga.K; ga.bias = c->bsum[l]; ga.c32 = gx; ga.ldc = 4 * R;
      ga.M = Bn * T; ga.N = 8 * R; ga.rep = 1; ga.dtype = et;
      ga.out_rows = lrows; ga.out_rows_ncol = 4 * R;
      if (ga.M <= 64 && c->wih_f[l]) { ga.w = c->wih_f[l]; ga.w_frag = 1; }
      if (!ih_fused && !(l == 0 && use_tab))
        steps.push_back({"lstm_ih" + std::to_string(l), [=](hipStream_t st) { return vog_gemm_bias_act(&ga, st); }});
      // final state must land in hA (adjacent to out16): after T steps it is in buf[T % 2]
      void* hA = ofrag ? ws.at<void>("lstm_hA2_" + std::to_string(l))
                       : (void*)(ws.at<unsigned short>("lstm_out16_" + std::to_string(l)) + (int64_t)Bn * T * 2 * R);
      void* hB = ws.at<void>("lstm_hB_" + std::to_string(l));
      void* hb[2] = {(T % 2) == 0 ? hA : hB, (T % 2) == 0 ? hB : hA};
      if (c->lstm_persistent && vog_bilstm_layer_supported(Bn, R)) {
        vog_lstm_layer_args pa{};
        pa.gxs = gx; pa.whh = c->whh[l]; pa.hx = ws.at<void>("lstm_hx_" + std::to_string(l));
        pa.sync = ws.at<uint32_t>("lstm_sync_" + std::to_string(l));
        pa.out16 = ws.at<void>("lstm_out16_" + std::to_string(l));
        pa.lens = b->srl_arg_word_mask_len; pa.Bn = Bn; pa.T = T; pa.R = R; pa.dtype = et; pa.out_frag = ofrag ? 1 : 0;
        pa.fault = b->fault; pa.inject_stall = c->lstm_inject_stall;
        if (l == 0 && use_tab) { pa.gx_table = c->gx0_tab; pa.tok = tok; }
        if (ih_fused) {
          pa.wih = c->wih_p[l]; pa.bias = c->bsum[l]; pa.K = Kin;
          pa.xa = l == 0 ? ws.at<void>("emb_a0") : ws.at<void>("lstm_out16_" + std::to_string(l - 1));
        }
        steps.push_back({"lstm_layer", [=](hipStream_t st) { return vog_bilstm_layer(&pa, st); }});
        continue;
      }
      for (int s = 0; s < T; ++s) {
        vog_lstm_step_args la{};
        la.gx = gx; la.whh = c->whh[l]; la.h_in = hb[s % 2]; la.h_out = hb[(s + 1) % 2];
        la.c = ws.at<float>("lstm_c_" + std::to_string(l));
        la.out16 = ws.at<void>("lstm_out16_" + std::to_string(l));
        la.lens = b->srl_arg_word_mask_len; la.Bn = Bn; la.T = T; la.R = R; la.step = s; la.dtype = et;
        la.out_frag = ofrag ? 1 : 0; la.final_row0 = Bn * T;
        steps.push_back({"lstm_step", [=](hipStream_t st) { return vog_bilstm_step(&la, st); }});
      }
    }
    vog_gemm_args po{}; po.c16_dtype = -1;
    po.a = ws.at<void>("lstm_out16_" + std::to_string(d.rnn_layers - 1)); po.lda = 2 * R;
    po.w = c->w_outproj; po.ldw = 2 * R; po.bias = c->b_outproj; po.relu = 1;
    po.c32 = ws.at<float>("full"); po.ldc = g.L; po.M = Bn * T + Bn; po.N = g.L; po.K = 2 * R;
    po.rep = 1; po.dtype = et;
    if (po.M <= 64 && c->w_outproj_f) { po.w = c->w_outproj_f; po.w_frag = 1; }
    po.a_frag = (Bn * T + Bn) <= 64 ? 1 : 0;
    if (po.M > 64 && (po.K % 512) == 0 && (g.L % 4) == 0) {
      // few output tiles (L = 256 columns), long K: split K over 8 slabs, bias + ReLU in the finish
      float* full32 = po.c32;
      vog_gemm_args ps = po; ps.bias = nullptr; ps.relu = 0; ps.c32 = ws.at<float>("outproj_slabs"); ps.splitk = 8;
      ps.w_frag = 0; ps.a_frag = 0; ps.w = c->w_outproj; ps.ldw = 2 * R;
      steps.push_back({"lstm_outproj", [=](hipStream_t st) { return vog_gemm_bias_act(&ps, st); }});
      vog_splitk_prob f0{};
      f0.slabs = ps.c32; f0.splits = 8; f0.M = po.M; f0.N = g.L; f0.bias = c->b_outproj; f0.relu = 1; f0.rep = 1;
      f0.c32 = full32; f0.ldc = g.L;
      steps.push_back({"lstm_outproj_finish", [=](hipStream_t st) { return vog_splitk_finish(&f0, nullptr, st); }});
    } else {
      steps.push_back({"lstm_outproj", [=](hipStream_t st) { return vog_gemm_bias_act(&po, st); }});
    }
    const float* full = po.c32;
    float* lang = lang_vec;
    const int64_t *cap = b->srl_arg_words_capture, *im = b->srl_arg_inds_msk;
    const float *wa = c->w_arg, *ba = c->b_arg;
    const int L = g.L;
    steps.push_back({"argvec", [=](hipStream_t st) {
      return vog_srl_argvec(full, cap, im, wa, ba, lang, Bn, T, nsrl, L, st); }});
  }
  if (structured && !lang_only) {
    {
      // language half of mul_tx's layer-0 QKV: depends on `lang` only, so it rides on this branch
      float* lang = lang_vec;
      const TxWeights& tw = c->mul;
      const TxLayer& L0 = tw.layers[0];
      const int ncol = 3 * tw.H * tw.dp;
      vog_gemm_args gl{}; gl.c16_dtype = -1;
      gl.a = lang; gl.a_is_f32 = 1; gl.lda = g.L; gl.w = L0.wqkv + g.d_obj; gl.ldw = tw.d;
      gl.c32 = ws.at<float>("mul_pl"); gl.ldc = ncol; gl.M = g.Bn * d.nsrl; gl.N = ncol; gl.K = g.L;
      gl.rep = 1; gl.dtype = (vog_dtype)d.tx_dtype;
      if (gl.M <= 64 && L0.wqkv_lang_f) { gl.w = L0.wqkv_lang_f; gl.ldw = g.L; gl.w_frag = 1; }
      if (c->tx_split) {
        // hi + lo operands: the M <= 64 kernel carries them; more than 64 (sentence, argument) rows go through it 64 at a time
        if (!L0.wqkv_lang_f || !L0.wqkv_lang_f_lo) VOG_FAIL(-5, "tx_split: lang_enc %% 32 != 0");
        gl.w = L0.wqkv_lang_f; gl.ldw = g.L; gl.w_frag = 1; gl.w_lo = L0.wqkv_lang_f_lo;
        const int Mall = gl.M;
        steps.push_back({"mul_pl", [=](hipStream_t st) {
          for (int m0 = 0; m0 < Mall; m0 += 64) {
            vog_gemm_args g2 = gl;
            g2.a = (const float*)gl.a + (int64_t)m0 * gl.lda; g2.c32 = gl.c32 + (int64_t)m0 * gl.ldc;
            g2.M = Mall - m0 < 64 ? Mall - m0 : 64;
            VOG_TRY(vog_gemm_bias_act(&g2, st));
          }
          return 0; }});
      } else
      steps.push_back({"mul_pl", [=](hipStream_t st) { return vog_gemm_bias_act(&gl, st); }});
    }
  }
  for (size_t i = lang_begin; i < steps.size(); ++i) steps[i].branch = steps[i].name == "prep" ? 2 : 1;   // 2: before both chains
  if (lang_only) return 0;
  // ---- visual encoders (a12, a13)
  float* ps32 = ws.at<float>("prop_seg");
  void* ps16 = ws.at<void>("prop_seg16");
  {
    // fused visual prologue: raw features -> encoder operand type (the LDS-DMA GEMM cannot
    // convert in flight) + the box-bias precursors of both transformers
    if (!fuse_prep) {
      const vog_visprep_args vp = make_visprep();
      steps.push_back({"vis_prep", [=](hipStream_t st) { return vog_vis_prep(&vp, st); }});
    }
    // the two encoders have 52 / 12 output tiles and K = 2048 / 3072: split K so every CU
    // gets a slice, partial products go to fp32 slabs, one finishing pass for both
    auto pick_split = [](int M, int N, int K) {
      const int64_t tiles = (int64_t)ceil_div(M, 64) * ceil_div(N, 64);
      int sp = (int)((511 + tiles) / tiles);
      const int nk = K / 64;
      if (sp > nk / 4) sp = nk / 4;
      if (sp > 16) sp = 16;
      return sp < 1 ? 1 : sp;
    };
    const int Mp = (int)g.rows_obj, Ms = g.n_vid * g.Fv;
    if (enc_fused) {
      vog_visenc_args ve{};
      ve.prop = b->pad_region_feature; ve.seg = b->seg_feature_for_frms;
      ve.w_prop_f = c->w_prop_f; ve.w_seg_f = c->w_seg_f; ve.b_prop = c->b_prop; ve.b_seg = c->b_seg;
      ve.c32 = ps32; ve.c16 = ps16; ve.ldc = g.d_obj; ve.c16_dtype = d.tx_dtype;
      ve.n_prop_rows = Mp; ve.nppf0 = d.nppf0; ve.prop_dim = d.prop_dim; ve.seg_dim = d.seg_dim;
      ve.prop_enc = d.prop_enc; ve.seg_enc = d.seg_enc; ve.dtype = et;
      const bool will_pair = c->pair_launches && !shared && c->lstm_persistent &&
                             vog_bilstm_layer_supported(Bn, R) && allow_pairs;
      // lean form (64-row x 128-column workgroups, every fp32 row read once per column half): when the
      // encoders share the launch of a BiLSTM layer (busy-CU time matters, latency is hidden), and for
      // the p100 shapes where the wide form's 8 column slices per row tile re-stream the features
      ve.lean = c->enc_lean < 0 ? ((will_pair || Mp >= 4096) ? 1 : 0) : c->enc_lean;
      if (c->tx_split) {     // hi + lo operands: the stream form carries them (visenc_dev.h)
        ve.lean = 1;
        ve.w_prop_f_lo = c->w_prop_f_lo; ve.w_seg_f_lo = c->w_seg_f_lo; ve.c16_lo = ws.at<void>("prop_seg16_lo");
      }
      // many proposals per frame: the replication of the segment rows is its own (chip-wide) copy launch,
      // so the encoder kernel stays one launch and can still share the BiLSTM layer's
      const bool rep_step = ve.lean && d.nppf0 > 16 && (d.seg_enc % 4) == 0 && (d.prop_enc % 4) == 0 && (g.d_obj % 4) == 0;
      if (c->tx_split && rep_step) VOG_FAIL(-5, "tx_split: more than 16 proposals per frame are not supported (use the fp32 path)");
      ve.defer_replicas = rep_step ? 1 : 0;
      steps.push_back({"vis_enc", [=](hipStream_t st) { return vog_vis_encode(&ve, st); }});
      if (rep_step) steps.push_back({"seg_rep", [=](hipStream_t st) { return vog_seg_replicate(&ve, st); }});
    }
    const bool can_split = !enc_fused && (d.prop_dim % 64) == 0 && (d.seg_dim % 64) == 0 && Mp > 64 && Ms > 64 &&
                           (d.prop_enc % 4) == 0 && (d.seg_enc % 4) == 0;
    const int sp_p = can_split ? pick_split(Mp, d.prop_enc, d.prop_dim) : 1;
    const int sp_s = can_split ? pick_split(Ms, d.seg_enc, d.seg_dim) : 1;
    float* slab_p = ws.at<float>("enc_slabs");
    float* slab_s = slab_p + (int64_t)16 * Mp * d.prop_enc;
    vog_gemm_args pe{}; pe.c16_dtype = d.tx_dtype;
    pe.a = ws.at<void>("prop16"); pe.a_is_f32 = 0; pe.lda = d.prop_dim; pe.w = c->w_prop; pe.ldw = d.prop_dim;
    pe.M = Mp; pe.N = d.prop_enc; pe.K = d.prop_dim; pe.rep = 1; pe.dtype = et;
    vog_gemm_args se{}; se.c16_dtype = d.tx_dtype;
    se.a = ws.at<void>("seg16"); se.a_is_f32 = 0; se.lda = d.seg_dim; se.w = c->w_seg; se.ldw = d.seg_dim;
    se.M = Ms; se.N = d.seg_enc; se.K = d.seg_dim; se.dtype = et;
    if (can_split && sp_p > 1 && sp_s > 1) {
      pe.c32 = slab_p; pe.ldc = d.prop_enc; pe.splitk = sp_p;
      se.c32 = slab_s; se.ldc = d.seg_enc; se.splitk = sp_s; se.rep = 1;
      steps.push_back({"prop_enc", [=](hipStream_t st) { return vog_gemm_bias_act(&pe, st); }});
      steps.push_back({"seg_enc", [=](hipStream_t st) { return vog_gemm_bias_act(&se, st); }});
      vog_splitk_prob f0{}, f1{};
      f0.slabs = slab_p; f0.splits = sp_p; f0.M = Mp; f0.N = d.prop_enc; f0.bias = c->b_prop; f0.relu = 1; f0.rep = 1;
      f0.c32 = ps32; f0.c16 = ps16; f0.ldc = g.d_obj; f0.ldc16 = g.d_obj; f0.c16_dtype = d.tx_dtype;
      f1.slabs = slab_s; f1.splits = sp_s; f1.M = Ms; f1.N = d.seg_enc; f1.bias = c->b_seg; f1.relu = 1; f1.rep = d.nppf0;
      f1.c32 = ps32 + d.prop_enc; f1.c16 = (unsigned short*)ps16 + d.prop_enc; f1.ldc = g.d_obj; f1.ldc16 = g.d_obj;
      f1.c16_dtype = d.tx_dtype;
      steps.push_back({"enc_finish", [=](hipStream_t st) { return vog_splitk_finish(&f0, &f1, st); }});
    } else if (!enc_fused) {
      pe.bias = c->b_prop; pe.relu = 1; pe.c32 = ps32; pe.c16 = ps16; pe.ldc = g.d_obj; pe.ldc16 = g.d_obj;
      steps.push_back({"prop_enc", [=](hipStream_t st) { return vog_gemm_bias_act(&pe, st); }});
      se.bias = c->b_seg; se.relu = 1; se.c32 = ps32 + d.prop_enc;
      se.c16 = (unsigned short*)ps16 + d.prop_enc; se.ldc = g.d_obj; se.ldc16 = g.d_obj; se.rep = d.nppf0;
      steps.push_back({"seg_enc", [=](hipStream_t st) { return vog_gemm_bias_act(&se, st); }});
    }
  }
  // ---- object transformer (a7, a8)
  const float* vis32 = ps32;
  const void* vis16 = ps16;
  const void* vis16_lo = c->tx_split ? ws.at<void>("prop_seg16_lo") : nullptr;
  int tx_err = 0;
  if (c->tx_split && !(c->fused_enc && c->w_prop_f_lo))
    VOG_FAIL(-5, "tx_split needs the fused feature encoders (feature dims %% 256, encode sizes %% 32 and <= 256)");
  if (has_obj(d)) {
    const void* in_lo = vis16_lo;
    tx_steps(c, c->obj, "obj", g, ws, b, ps32, ps16, g.S_obj, g.N_obj, g.npad_obj, g.spv_obj, g.N_obj,
             g.fdiv_obj, has_mul(d) ? d.tx_dtype : -1, steps, &vis32, &vis16, nullptr, nullptr, true, nullptr,
             in_lo, nullptr, /*out_feeds_attn=*/has_mul(d), &vis16_lo, &tx_err);
    if (tx_err) return tx_err;
  }
  // ---- vis || lang tokens in mul_tx order (a10, a11)
  vog_vislang_args va{};
  va.vis = vis32; va.lang = lang_vec; va.x32 = ws.at<float>("xmul"); va.x16 = ws.at<void>("xmul16");
  va.n_vid = g.n_vid; va.nfrm = g.nfrm; va.nppf = g.nppf; va.nsrl = d.nsrl; va.dv = g.d_obj; va.dl = g.L;
  va.lang_per_vid = g.nvl > 1 ? 1 : 0; va.nc_v = g.nc_v; va.dtype = (vog_dtype)(has_mul(d) ? d.tx_dtype : d.enc_dtype);
  // mul_tx consumes the token structure directly (layer-0 QKV and its residual), so the
  // token matrix is only materialised for ImgGrnd / VidGrnd, whose lin2 reads it
  { Step j; j.name = "join"; j.branch = -1; steps.push_back(j); }
  if (!structured)
    steps.push_back({"vislang", [=](hipStream_t st) { return vog_vislang_layout(&va, st); }});
  const float* x32 = ws.at<float>("xmul");
  const void* x16 = ws.at<void>("xmul16");
  int head_dt = d.enc_dtype;   // the 16-bit copy feeding lin2 is always written in the head's type
  vog_score_args sa{};
  sa.w2 = c->w_lin2b; sa.b2 = c->b_lin2b; sa.arg_msk = b->srl_arg_inds_msk;
  sa.cmp_msk = b->num_cmp_msk; sa.outs = b->mdl_outs; sa.outs_eval = b->mdl_outs_eval;
  sa.n_vid = g.n_vid; sa.nfrm = g.nfrm; sa.nppf = g.nppf; sa.nsrl = d.nsrl; sa.dh = 256;
  sa.conc_type = d.conc_type; sa.ncmp = g.ncmp; sa.nc_v = g.nc_v; sa.nvl = g.nvl;
  sa.nfrm0 = d.nfrm0; sa.nppf0 = d.nppf0;
  vog_pred_args pr{};
  pr.outs_eval = b->mdl_outs_eval; pr.props = b->pad_proposals; pr.fin_scores = b->fin_scores;
  pr.rec = b->pred_rec; pr.B = g.B; pr.ncmp = g.ncmp; pr.nsrl = d.nsrl; pr.nfrm0 = d.nfrm0;
  pr.nppf0 = d.nppf0; pr.conc_type = d.conc_type;
  pr.logit_max = ws.at<unsigned int>("logit_max"); pr.stats = b->stats; pr.published = ws.at<unsigned int>("logit_pub");
  if (has_mul(d))
    tx_steps(c, c->mul, "mul", g, ws, b, x32, x16, g.S_mul, g.N_mul, g.npad_mul, g.nfrm, g.nppf,
             (float)g.nfrm, d.enc_dtype, steps, &x32, &x16, structured ? &va : nullptr, vis16,
             /*last_needs_f32=*/false, d.enc_dtype == VOG_F16 ? &sa : nullptr,
             // (not structured: mul_tx would read the materialised token matrix, which has no remainder copy)
             structured ? vis16_lo : nullptr, structured ? vis16_lo : nullptr, false, nullptr, &tx_err);
  if (tx_err) return tx_err;
  if (c->tx_split && has_mul(d) && !structured)
    VOG_FAIL(-5, "tx_split needs the structured mul_tx layer 0 (d_obj %% 64 == 0, lang_enc %% 32 == 0, more than 64 visual rows)");
  // ---- score head (a9 tail / a20 / a17); x16 == NULL: the fused mul_tx tail already ran it
  if (x16 != nullptr) {
    vog_gemm_args l2{}; l2.c16_dtype = -1;
    if (head_dt == d.enc_dtype) { l2.a = x16; l2.a_is_f32 = 0; }
    else { l2.a = x32; l2.a_is_f32 = 1; }               // re-round from fp32 in the head's own type
    l2.lda = g.d_mul; l2.w = c->w_lin2; l2.ldw = g.d_mul; l2.bias = c->b_lin2; l2.relu = 1;
    l2.c32 = ws.at<float>("h1"); l2.ldc = 256; l2.M = (int)g.rows_mul; l2.N = 256; l2.K = g.d_mul;
    l2.rep = 1; l2.dtype = et;
    steps.push_back({"lin2", [=](hipStream_t st) { return vog_gemm_bias_act(&l2, st); }});
    sa.h1 = l2.c32;
    steps.push_back({"score", [=](hipStream_t st) { return vog_score_head(&sa, st); }});
  }
  if (g.sep) {
    vog_predcmp_args pa{};
    pa.final_hidden = shared ? b->shared_final_hidden : ws.at<float>("full") + (int64_t)g.Bn * g.T * g.L;
    pa.prop_seg = ps32; pa.w0 = c->w_sv0; pa.b0 = c->b_sv0; pa.w2 = c->w_sv2; pa.b2 = c->b_sv2;
    pa.outs = b->mdl_outs; pa.arg_msk = b->srl_arg_inds_msk; pa.cmp_msk = b->num_cmp_msk;
    pa.verb_ind = b->verb_ind_in_srl; pa.vidf_outs = b->vidf_outs; pa.fin_scores_loss = b->fin_scores_loss;
    pa.fin_scores = b->fin_scores; pa.B = g.B; pa.ncmp = g.ncmp; pa.nvl = g.nvl; pa.nsrl = d.nsrl;
    pa.NP = g.NP; pa.nfrm0 = d.nfrm0; pa.nppf0 = d.nppf0; pa.L = g.L; pa.dp0 = d.prop_enc; pa.dps = g.d_obj;
    steps.push_back({"pred_cmp", [=](hipStream_t st) { return vog_pred_cmp_head(&pa, st); }});
  }
  if (b->pred_rec)
    steps.push_back({"pred_head", [=](hipStream_t st) { return vog_pred_head(&pr, st); }});
  // ---- horizontal fusion (pair.hip): the language chain and the visual chain are independent until
  // mul_tx's attention, and neither fills the chip (the persistent BiLSTM layer holds 64 CUs for
  // ~46 us): step i of one shares a launch with step i of the other. The visual step moves up to the
  // language step's position (its own inputs are produced by earlier pairs); combinations without a
  // registered pair kernel fall back to two launches inside pair_launch.
  if (allow_pairs && c->pair_launches && !shared && c->lstm_persistent) {
    auto find = [&](const char* nm, int occurrence) {
      for (size_t i = 0; i < steps.size(); ++i)
        if (steps[i].name == nm && steps[i].branch >= 0 && occurrence-- == 0) return (int)i;
      return -1;
    };
    // language step -> visual step that shares its launch (+ a visual step that follows on its own).
    // Pairs are chosen by shape: a 512-thread body next to a 512-thread body, 256 next to 256 (a
    // register-heavy 512-thread partner would cut the occupancy of a small-block streaming kernel:
    // input projection + encoders in one grid measured 29.8 us against 6.4 + 16.0 apart).
    // The BiLSTM layers (64 CUs for ~38 us each) take the encoders and the obj_tx tail as partners; the
    // obj_tx QKV projection and attention follow the first pair on their own (paired with the layer-1
    // input projection they measured SLOWER than apart: 22.5 vs 9.5 + 7.4 us).
    struct Want { const char* lang; int occ; const char* vis; const char* then[3]; };
    const bool has_rep = find("seg_rep", 0) >= 0;
    // (round 6) where layer 1's input projection is a GEMM launch (more than 80 columns) obj_tx's QKV projection shares THAT launch
    // (pair_mask bit 8) and the attention follows it, instead of both following the first pair on their own
    // Measured (scratch/r6_ae.sh, three interleaved runs each): cfg 3 (Bn x T = 96: the projection is 256 tiles, one per CU) 68.2 vs
    // 66.9 k queries/s; cfg 5 (192 columns: 384 tiles) 111.5 vs 113.1 k - the pair only where the projection leaves room on the chip.
    const bool ih1_pair = find("lstm_ih1", 0) >= 0 && find("obj_qkv", 0) >= 0 && ((c->pair_mask >> 3) & 1) && Bn * T <= 128;
    std::vector<Want> want;
    if (!ih1_pair) {
      want.push_back(has_rep ? Want{"lstm_layer", 0, "vis_enc", {"seg_rep", "obj_qkv", "obj_attn"}}
                             : Want{"lstm_layer", 0, "vis_enc", {"obj_qkv", "obj_attn", nullptr}});
    } else {
      want.push_back(has_rep ? Want{"lstm_layer", 0, "vis_enc", {"seg_rep", nullptr, nullptr}}
                             : Want{"lstm_layer", 0, "vis_enc", {nullptr, nullptr, nullptr}});
    }
    want.push_back({"lstm_layer", 1, "obj_tail", {nullptr, nullptr, nullptr}});
    want.push_back({"lstm_outproj", 0, "mul_pv", {nullptr, nullptr, nullptr}});
    if (ih1_pair) want.insert(want.begin() + 1, Want{"lstm_ih1", 0, "obj_qkv", {"obj_attn", nullptr, nullptr}});
    struct Plan2 { int ia, ib; int it[3]; };
    std::vector<Plan2> plans;
    bool ok = true;
    int prev_vis = -1;
    int wi = -1;
    for (auto& w : want) {
      ++wi;
      const int bit = std::string(w.lang) == "lstm_ih1" ? 3 : (std::string(w.lang) == "lstm_outproj" ? 2 : w.occ);
      if (!((c->pair_mask >> bit) & 1)) continue;      // this pair stays two launches (its visual step keeps its place)
      Plan2 q{find(w.lang, w.occ), find(w.vis, 0), {-1, -1, -1}};
      // every visual step only moves EARLIER (its producers sit in earlier pairs) and the visual chain
      // keeps its own order; any missing piece (other model variants / shapes) leaves the rest unpaired
      if (q.ia < 0 || q.ib < 0 || q.ib < q.ia || q.ib < prev_vis) { ok = false; break; }
      int last = q.ib;
      for (int k = 0; k < 3 && w.then[k]; ++k) {
        q.it[k] = find(w.then[k], 0);
        if (q.it[k] < last) { ok = false; break; }
        last = q.it[k];
      }
      if (!ok) break;
      prev_vis = last;
      plans.push_back(q);
    }
    if (plans.empty()) ok = false;
    if (ok) {
      // nothing else of the visual chain may sit between the moved steps (it would be overtaken)
      std::vector<int> moved;
      for (auto& q : plans) { moved.push_back(q.ib); for (int k = 0; k < 3; ++k) if (q.it[k] >= 0) moved.push_back(q.it[k]); }
      for (int i = moved.front(); i <= moved.back() && ok; ++i)
        if (steps[i].branch == 0 && std::find(moved.begin(), moved.end(), i) == moved.end()) ok = false;
    }
    if (ok) {
      std::vector<int> role(steps.size(), 0);           // 1 = removed from its old position
      std::vector<Step> out;
      for (auto& q : plans) { role[q.ib] = 1; for (int k = 0; k < 3; ++k) if (q.it[k] >= 0) role[q.it[k]] = 1; }
      for (size_t i = 0; i < steps.size(); ++i) {
        if (role[i]) continue;
        const Plan2* q = nullptr;
        for (auto& x : plans) if (x.ia == (int)i) q = &x;
        if (!q) { out.push_back(steps[i]); continue; }
        Step m = steps[i];
        auto fa = steps[q->ia].fn, fb = steps[q->ib].fn;
        m.name = steps[q->ia].name + "+" + steps[q->ib].name;
        m.fn = [fa, fb](hipStream_t st) { return pair_launch(fa, fb, st, nullptr); };
        out.push_back(m);
        for (int k = 0; k < 3; ++k)
          if (q->it[k] >= 0) { Step t = steps[q->it[k]]; t.branch = m.branch; out.push_back(t); }
      }
      steps.swap(out);
    }
  }
  // VOG_SKIP_STEPS=name,name,... (perf experiments only; results are WRONG): drop steps whose
  // name starts with one of the entries, to measure their marginal cost in the throughput regime
  if (const char* skip = perf_env("VOG_SKIP_STEPS")) {
    std::vector<std::string> pre;
    std::string cur;
    for (const char* q = skip;; ++q) {
      if (*q == ',' || *q == 0) { if (!cur.empty()) pre.push_back(cur); cur.clear(); if (!*q) break; }
      else cur.push_back(*q);
    }
    std::vector<Step> kept;
    for (auto& s : steps) {
      bool drop = false;
      for (auto& pfx : pre) drop |= s.name.rfind(pfx, 0) == 0;
      if (!drop) kept.push_back(s);
    }
    steps.swap(kept);
  }
  return 0;
}

}  // namespace vog

// =============================================================================
// C ABI
// =============================================================================
extern "C" int vog_version(void) { return VOG_ABI_VERSION; }
extern "C" const char* vog_last_error(void) { return vog::g_err; }

extern "C" int vog_ctx_create(const vog_model_desc* d, vog_ctx** out) {
  VOG_CHECK_ARG(d && out);
  VOG_CHECK_ARG(d->mdl_kind >= 0 && d->mdl_kind <= 2 && d->conc_type >= 0 && d->conc_type <= 2);
  VOG_CHECK_ARG(d->rnn_size % 32 == 0 && d->emb_dim % 8 == 0 && d->prop_dim % 8 == 0 && d->seg_dim % 8 == 0);
  VOG_CHECK_ARG(d->prop_enc % 8 == 0 && d->seg_enc % 8 == 0 && d->lang_enc % 8 == 0);
  VOG_CHECK_ARG(((d->prop_enc + d->seg_enc) / 2) % 8 == 0 && ((d->prop_enc + d->seg_enc + d->lang_enc) / 2) % 8 == 0);
  VOG_CHECK_ARG(d->nsrl > 0 && d->seq_len > 0 && d->nfrm0 > 0 && d->nppf0 > 0 && d->rnn_layers > 0);
  vog_ctx* c = new vog_ctx();
  c->d = *d;
  // VOG_LSTM_PERSISTENT=0/1 presets the option (test sweeps); vog_ctx_set_int overrides it
  // HIP streams map onto GPU_MAX_HW_QUEUES hardware queues (default 4 = the co-residency limit of the
  // persistent BiLSTM layer kernel: 4 instances x 64 CUs): with more queues it is off by default
  if (const char* e = getenv("GPU_MAX_HW_QUEUES")) { if (atoi(e) > 4) c->lstm_persistent = 0; }
  if (const char* e = getenv("VOG_LSTM_PERSISTENT")) c->lstm_persistent = atoi(e) ? 1 : 0;
  if (const char* e = perf_env("VOG_FUSED_IH")) c->fused_ih = atoi(e);
  const int R = d->rnn_size, E = d->emb_dim, L = d->lang_enc;
  add_w(c, "lstm_encoder.embed_tokens.weight", (int64_t)(d->vocab_size + 1) * E);
  for (int l = 0; l < d->rnn_layers; ++l)
    for (const char* sfx : {"", "_reverse"}) {
      const int in = l == 0 ? E : 2 * R;
      std::string s = "_l" + std::to_string(l) + sfx;
      add_w(c, "lstm_encoder.lstm.weight_ih" + s, (int64_t)4 * R * in);
      add_w(c, "lstm_encoder.lstm.weight_hh" + s, (int64_t)4 * R * R);
      add_w(c, "lstm_encoder.lstm.bias_ih" + s, 4 * R);
      add_w(c, "lstm_encoder.lstm.bias_hh" + s, 4 * R);
    }
  add_w(c, "lstm_out_feat_proj.0.weight", (int64_t)L * 2 * R);
  add_w(c, "lstm_out_feat_proj.0.bias", L);
  add_w(c, "srl_arg_words_out_enc.0.weight", (int64_t)L * 2 * L);
  add_w(c, "srl_arg_words_out_enc.0.bias", L);
  add_w(c, "prop_encoder.0.weight", (int64_t)d->prop_enc * d->prop_dim);
  add_w(c, "prop_encoder.0.bias", d->prop_enc);
  add_w(c, "seg_encoder.0.weight", (int64_t)d->seg_enc * d->seg_dim);
  add_w(c, "seg_encoder.0.bias", d->seg_enc);
  add_w(c, "seg_verb_classf.0.weight", (int64_t)256 * (d->seg_enc + L));
  add_w(c, "seg_verb_classf.0.bias", 256);
  add_w(c, "seg_verb_classf.2.weight", 256);
  add_w(c, "seg_verb_classf.2.bias", 1);
  const int d_obj = d->prop_enc + d->seg_enc, d_mul = d_obj + L;
  add_w(c, "lin2.0.weight", (int64_t)256 * d_mul);
  add_w(c, "lin2.0.bias", 256);
  add_w(c, "lin2.2.weight", 256);
  add_w(c, "lin2.2.bias", 1);
  if (has_obj_weights(*d)) {
    declare_tx(c, "obj_txf", d_obj, d->obj_layers);
    add_w(c, "pe_obj_sub_enc.0.weight", (int64_t)d->obj_heads * 5);
    add_w(c, "pe_obj_sub_enc.0.bias", d->obj_heads);
  }
  if (has_mul(*d)) {
    declare_tx(c, "mult_txf", d_mul, d->mul_layers);
    add_w(c, "pe_mul_sub_enc.0.weight", (int64_t)d->mul_heads * 5);
    add_w(c, "pe_mul_sub_enc.0.bias", d->mul_heads);
  }
  *out = c;
  return 0;
}

extern "C" int vog_ctx_num_weights(const vog_ctx* c) { return c ? (int)c->names.size() : -1; }
extern "C" const char* vog_ctx_weight_name(const vog_ctx* c, int i) {
  return (c && i >= 0 && i < (int)c->names.size()) ? c->names[i].c_str() : nullptr;
}
extern "C" int64_t vog_ctx_weight_numel(const vog_ctx* c, int i) {
  return (c && i >= 0 && i < (int)c->names.size()) ? c->numel.at(c->names[i]) : -1;
}

extern "C" int vog_ctx_set_weight(vog_ctx* c, const char* name, const float* host, int64_t numel) {
  VOG_CHECK_ARG(c && name && host);
  std::string n(name);
  if (n.rfind("module.", 0) == 0) n = n.substr(7);          // DDP-wrapped checkpoints (trn_utils.py:536-592)
  // transformers trained with mdl.{obj,mul}_tx.use_ddp=True keep an inner `module.` (mdl_vog.py:441-445,577-578)
  for (const char* pre : {"mult_txf.", "obj_txf."}) {
    const std::string pm = std::string(pre) + "module.";
    if (n.rfind(pm, 0) == 0) n = std::string(pre) + n.substr(pm.size());
  }
  // legacy LayerNorm parameter names (trn_utils.py:560-565)
  for (const char* pr : {".gamma", ".beta"}) {
    const std::string suf(pr);
    if (n.size() > suf.size() && n.compare(n.size() - suf.size(), suf.size(), suf) == 0 &&
        n.find("layernorm") != std::string::npos)
      n = n.substr(0, n.size() - suf.size()) + (suf == ".gamma" ? ".weight" : ".bias");
  }
  auto it = c->numel.find(n);
  if (it == c->numel.end()) {
    // parameters that exist in reference checkpoints but are not read by forward: srl_simple_lin and
    // lin_tmp always; a transformer's weights only when THIS model variant does not declare that
    // transformer at all (a VidGrnd / VOGNet checkpoint loaded into a smaller variant). A key of a
    // declared transformer that does not match is an error, never silently dropped.
    const bool obj_decl = has_obj_weights(c->d), mul_decl = has_mul(c->d);
    if (n.rfind("srl_simple_lin", 0) == 0 || n.rfind("lin_tmp", 0) == 0 ||
        (!obj_decl && (n.rfind("obj_txf", 0) == 0 || n.rfind("pe_obj_sub_enc", 0) == 0)) ||
        (!mul_decl && (n.rfind("mult_txf", 0) == 0 || n.rfind("pe_mul_sub_enc", 0) == 0)))
      return 0;
    VOG_FAIL(-3, "unexpected weight '%s'", name);
  }
  if (it->second != numel) VOG_FAIL(-3, "weight '%s': numel %lld, expected %lld", name, (long long)numel, (long long)it->second);
  c->host[n].assign(host, host + numel);
  c->finalized = false;
  return 0;
}

extern "C" int vog_ctx_finalize(vog_ctx* c) {
  VOG_CHECK_ARG(c);
  for (auto& n : c->names)
    if (!c->host.count(n)) VOG_FAIL(-3, "missing weight '%s'", n.c_str());
  for (void* p : c->allocs) (void)hipFree(p);
  c->allocs.clear();
  c->wih.clear(); c->wih_f.clear(); c->whh.clear(); c->bsum.clear(); c->wih_p.clear();
  c->obj = TxWeights(); c->mul = TxWeights();
  const vog_model_desc& d = c->d;
  const int R = d.rnn_size, et = d.enc_dtype;
  VOG_TRY(up32(c, "lstm_encoder.embed_tokens.weight", &c->emb));
  VOG_TRY(up16(c, "lstm_encoder.embed_tokens.weight", d.enc_dtype, &c->emb16));
  for (int l = 0; l < d.rnn_layers; ++l) {
    const int in = l == 0 ? d.emb_dim : 2 * R;
    std::vector<unsigned short> wih((size_t)8 * R * in), whh((size_t)8 * R * R);
    std::vector<float> bs((size_t)8 * R);
    int dir = 0;
    for (const char* sfx : {"", "_reverse"}) {
      std::string s = "_l" + std::to_string(l) + sfx;
      const auto& a = W(c, "lstm_encoder.lstm.weight_ih" + s);
      const auto& h = W(c, "lstm_encoder.lstm.weight_hh" + s);
      const auto& bi = W(c, "lstm_encoder.lstm.bias_ih" + s);
      const auto& bh = W(c, "lstm_encoder.lstm.bias_hh" + s);
      for (size_t i = 0; i < a.size(); ++i) wih[(size_t)dir * 4 * R * in + i] = h_to16(a[i], et);
      (void)h;
      for (int i = 0; i < 4 * R; ++i) bs[(size_t)dir * 4 * R + i] = bi[i] + bh[i];
      ++dir;
    }
    {
      std::string s0 = "_l" + std::to_string(l), s1 = s0 + "_reverse";
      VOG_TRY(vog_lstm_pack_whh(W(c, "lstm_encoder.lstm.weight_hh" + s0).data(),
                                W(c, "lstm_encoder.lstm.weight_hh" + s1).data(), whh.data(), R, (vog_dtype)et));
    }
    {
      unsigned short* pp = nullptr;
      if (in % 256 == 0 && R % 32 == 0) {
        std::vector<unsigned short> wp((size_t)8 * R * in);
        std::string s0 = "_l" + std::to_string(l);
        VOG_TRY(vog_lstm_pack_w(W(c, "lstm_encoder.lstm.weight_ih" + s0).data(),
                                W(c, "lstm_encoder.lstm.weight_ih" + s0 + "_reverse").data(), wp.data(), R, in, (vog_dtype)et));
        VOG_TRY(upload<unsigned short>(c, wp, &pp));
      }
      c->wih_p.push_back(pp);
    }
    unsigned short *pw, *ph; float* pb;
    {
      std::vector<float> cat((size_t)8 * R * in);
      int dd = 0;
      for (const char* sfx : {"", "_reverse"}) {
        const auto& a = W(c, "lstm_encoder.lstm.weight_ih_l" + std::to_string(l) + sfx);
        memcpy(&cat[(size_t)dd * 4 * R * in], a.data(), a.size() * sizeof(float));
        ++dd;
      }
      std::vector<unsigned short> wf((size_t)8 * R * in);
      unsigned short* pf = nullptr;
      if (in % 32 == 0) {
        VOG_TRY(vog_pack_w_frag(cat.data(), in, 8 * R, in, wf.data(), (vog_dtype)et));
        VOG_TRY(upload<unsigned short>(c, wf, &pf));
      }
      c->wih_f.push_back(pf);
    }
    VOG_TRY(upload<unsigned short>(c, wih, &pw));
    VOG_TRY(upload<unsigned short>(c, whh, &ph));
    VOG_TRY(upload<float>(c, bs, &pb));
    c->wih.push_back(pw); c->whh.push_back(ph); c->bsum.push_back(pb);
  }
  // Gate table of layer 0 (round 6): G[v] = emb[v] . W_ih_l0^T + b_ih + b_hh for every token v and both directions - the input
  // projection of layer 0 depends on the token alone, so it is computed ONCE per checkpoint for the whole vocabulary (the same
  // 16-bit operands and fp32 accumulation the per-batch projection used) and a forward reads Bn x T rows of it (vog_lstm_layer_args.gx_table).
  // 164 MB at vocab 5000, R = 1024: sized for 288 GB of HBM. VOG_GX_TABLE_MAX_MB bounds it (default 2048; 0 = never).
  c->gx0_tab = nullptr;
  {
    const int64_t rows = (int64_t)d.vocab_size + 1, cols = (int64_t)8 * R;
    static const int64_t max_mb = getenv("VOG_GX_TABLE_MAX_MB") ? atoll(getenv("VOG_GX_TABLE_MAX_MB")) : 2048;
    if (c->emb16 && rows > 64 && (d.emb_dim % 64) == 0 && rows * cols * 4 <= max_mb * (1ll << 20) && vog_bilstm_layer_supported(1, R)) {
      float* tab = nullptr;
      VOG_HIP(hipMalloc(&tab, (size_t)(rows * cols * 4)));
      c->allocs.push_back(tab);
      // table column (dir, unit, gate): the four gates of a unit adjacent (one 16-byte read per lane of the layer kernel) -
      // W_ih rows and biases permuted accordingly for the one GEMM that builds it
      const int E0 = d.emb_dim;
      std::vector<unsigned short> wperm((size_t)8 * R * E0);
      std::vector<float> bperm((size_t)8 * R);
      {
        int dd = 0;
        for (const char* sfx : {"", "_reverse"}) {
          const auto& a = W(c, std::string("lstm_encoder.lstm.weight_ih_l0") + sfx);
          const auto& bi = W(c, std::string("lstm_encoder.lstm.bias_ih_l0") + sfx);
          const auto& bh = W(c, std::string("lstm_encoder.lstm.bias_hh_l0") + sfx);
          for (int r = 0; r < 4; ++r)
            for (int u = 0; u < R; ++u) {
              const size_t dst = (size_t)dd * 4 * R + (size_t)u * 4 + r, src = (size_t)r * R + u;
              bperm[dst] = bi[src] + bh[src];
              for (int k = 0; k < E0; ++k) wperm[dst * E0 + k] = h_to16(a[src * E0 + k], et);
            }
          ++dd;
        }
      }
      unsigned short* wp_d = nullptr; float* bp_d = nullptr;
      VOG_HIP(hipMalloc(&wp_d, wperm.size() * 2));
      VOG_HIP(hipMalloc(&bp_d, bperm.size() * 4));
      VOG_HIP(hipMemcpy(wp_d, wperm.data(), wperm.size() * 2, hipMemcpyHostToDevice));
      VOG_HIP(hipMemcpy(bp_d, bperm.data(), bperm.size() * 4, hipMemcpyHostToDevice));
      vog_gemm_args ga{}; ga.c16_dtype = -1;
      ga.a = c->emb16; ga.a_is_f32 = 0; ga.lda = E0; ga.K = E0;
      ga.w = wp_d; ga.ldw = E0; ga.bias = bp_d; ga.c32 = tab; ga.ldc = cols;
      ga.M = (int)rows; ga.N = (int)cols; ga.rep = 1; ga.dtype = (vog_dtype)et;
      const int grc = vog_gemm_bias_act(&ga, nullptr);
      VOG_HIP(hipStreamSynchronize(nullptr));
      (void)hipFree(wp_d); (void)hipFree(bp_d);
      if (grc != 0) return grc;
      c->gx0_tab = tab;
    }
  }
  VOG_TRY(up16(c, "lstm_out_feat_proj.0.weight", et, &c->w_outproj));
  c->w_outproj_f = nullptr;
  if (d.lang_enc % 16 == 0) {
    std::vector<unsigned short> wf((size_t)d.lang_enc * 2 * R);
    VOG_TRY(vog_pack_w_frag(W(c, "lstm_out_feat_proj.0.weight").data(), 2 * R, d.lang_enc, 2 * R, wf.data(), (vog_dtype)et));
    VOG_TRY(upload<unsigned short>(c, wf, &c->w_outproj_f));
  }
  VOG_TRY(up32(c, "lstm_out_feat_proj.0.bias", &c->b_outproj));
  VOG_TRY(up16(c, "prop_encoder.0.weight", et, &c->w_prop));
  VOG_TRY(up32(c, "prop_encoder.0.bias", &c->b_prop));
  VOG_TRY(up16(c, "seg_encoder.0.weight", et, &c->w_seg));
  c->w_prop_f = c->w_seg_f = nullptr;
  if (vis_encode_supported(d.prop_dim, d.seg_dim, d.prop_enc, d.seg_enc)) {
    std::vector<unsigned short> wf((size_t)d.prop_enc * d.prop_dim), ws((size_t)d.seg_enc * d.seg_dim);
    VOG_TRY(vog_pack_w_frag(W(c, "prop_encoder.0.weight").data(), d.prop_dim, d.prop_enc, d.prop_dim, wf.data(), (vog_dtype)et));
    VOG_TRY(vog_pack_w_frag(W(c, "seg_encoder.0.weight").data(), d.seg_dim, d.seg_enc, d.seg_dim, ws.data(), (vog_dtype)et));
    VOG_TRY(upload<unsigned short>(c, wf, &c->w_prop_f));
    VOG_TRY(upload<unsigned short>(c, ws, &c->w_seg_f));
    c->w_prop_f_lo = c->w_seg_f_lo = nullptr;
    if (c->tx_split) {
      const auto& wp = W(c, "prop_encoder.0.weight");
      const auto& wsg = W(c, "seg_encoder.0.weight");
      VOG_TRY(vog_pack_w_frag(remainder16(wp.data(), wp.size(), et).data(), d.prop_dim, d.prop_enc, d.prop_dim, wf.data(), (vog_dtype)et));
      VOG_TRY(vog_pack_w_frag(remainder16(wsg.data(), wsg.size(), et).data(), d.seg_dim, d.seg_enc, d.seg_dim, ws.data(), (vog_dtype)et));
      VOG_TRY(upload<unsigned short>(c, wf, &c->w_prop_f_lo));
      VOG_TRY(upload<unsigned short>(c, ws, &c->w_seg_f_lo));
    }
  }
  VOG_TRY(up32(c, "seg_encoder.0.bias", &c->b_seg));
  VOG_TRY(up16(c, "lin2.0.weight", et, &c->w_lin2));
  VOG_TRY(up32(c, "lin2.0.bias", &c->b_lin2));
  c->w_lin2_p = nullptr;
  {
    const int dm = d.prop_enc + d.seg_enc + d.lang_enc;
    if (has_mul(d) && tx_tail_supported(dm, dm / 2, 192))    // (any kwo both widths take: only d / dh are in question here)
      VOG_TRY(up_frag32(c, W(c, "lin2.0.weight").data(), dm, 256, dm, et, &c->w_lin2_p));
  }
  VOG_TRY(up32(c, "lin2.2.weight", &c->w_lin2b));
  VOG_TRY(up32(c, "lin2.2.bias", &c->b_lin2b));
  VOG_TRY(up32(c, "srl_arg_words_out_enc.0.weight", &c->w_arg));
  VOG_TRY(up32(c, "srl_arg_words_out_enc.0.bias", &c->b_arg));
  VOG_TRY(up32(c, "seg_verb_classf.0.weight", &c->w_sv0));
  VOG_TRY(up32(c, "seg_verb_classf.0.bias", &c->b_sv0));
  VOG_TRY(up32(c, "seg_verb_classf.2.weight", &c->w_sv2));
  VOG_TRY(up32(c, "seg_verb_classf.2.bias", &c->b_sv2));
  const int d_obj = d.prop_enc + d.seg_enc, d_mul = d_obj + d.lang_enc;
  if (has_obj(d))
    VOG_TRY(finalize_tx(c, "obj_txf", "pe_obj_sub_enc.0", d_obj, d.obj_heads, d.obj_layers, d.obj_use_rel, &c->obj));
  if (has_mul(d))
    VOG_TRY(finalize_tx(c, "mult_txf", "pe_mul_sub_enc.0", d_mul, d.mul_heads, d.mul_layers, d.mul_use_rel, &c->mul));
  VOG_HIP(hipDeviceSynchronize());
  c->finalized = true;
  return 0;
}

extern "C" int vog_ctx_destroy(vog_ctx* c) {
  if (!c) return 0;
  for (void* p : c->allocs) (void)hipFree(p);
  delete c;
  return 0;
}

extern "C" int64_t vog_workspace_bytes(const vog_ctx* c, int B, int ncmp, int T) {
  if (!c || !c->finalized || B <= 0 || ncmp <= 0 || T <= 0) return -1;
  return make_plan(c, make_geo(c->d, B, ncmp, T)).total;
}

// Can this model run with hi + lo operands (option tx_split) at `ncmp` videos per query? The kernels that carry the three-MFMA
// products cover the gt5-sized shapes: fused feature encoders (stream form, <= 16 proposals per frame), the LDS-DMA QKV GEMM,
// attention over <= 256 tokens (plain) / one visual key block (structured mul_tx layer 0), the fused encoder-layer tails.
extern "C" int vog_ctx_split_supported(const vog_ctx* c, int ncmp) {
  if (!c || ncmp <= 0) return 0;
  const vog_model_desc& d = c->d;
  if (!has_obj_weights(d)) return 0;                                   // ImgGrnd: no attention at all
  if (!vis_encode_supported(d.prop_dim, d.seg_dim, d.prop_enc, d.seg_enc) || d.nppf0 > 16) return 0;
  const Geo g = make_geo(d, 4, ncmp, 1);
  const int d_obj = d.prop_enc + d.seg_enc, d_mul = d_obj + d.lang_enc;
  auto tx_ok = [&](int dm, int H, int N) {
    const int chunk = (dm + H - 1) / H, dp = attn_head_pad(chunk);
    return dp > 0 && tx_tail_supported(dm, dm / 2, H * dp) && (dm % 64) == 0 && N <= 256;
  };
  if (has_obj(d) && !tx_ok(d_obj, d.obj_heads, g.N_obj)) return 0;
  if (has_mul(d)) {
    if ((d_obj % 64) != 0 || (d.lang_enc % 32) != 0 || g.nppf > 32) return 0;     // structured layer 0, one visual key block
    if (!tx_ok(d_mul, d.mul_heads, d.mul_layers > 1 ? g.N_mul : 1)) return 0;
    if (d.mul_layers > 1 && g.N_mul > 256) return 0;
  }
  return 1;
}

extern "C" int vog_workspace_init(const vog_ctx* c, int B, int ncmp, int T, void* ws, size_t ws_bytes,
                                  void* stream) {
  VOG_CHECK_ARG(c && ws);
  const int64_t need = vog_workspace_bytes(c, B, ncmp, T);
  if (need < 0 || (int64_t)ws_bytes < need) VOG_FAIL(-2, "workspace too small");
  VOG_HIP(hipMemsetAsync(ws, 0, (size_t)need, (hipStream_t)stream));
  return 0;
}

extern "C" int vog_workspace_stage(const vog_ctx* c, int B, int ncmp, int T, const char* stage,
                                   int64_t* offset, int64_t* bytes) {
  VOG_CHECK_ARG(c && c->finalized && stage && offset && bytes);
  Plan p = make_plan(c, make_geo(c->d, B, ncmp, T));
  auto it = p.buf.find(stage);
  if (it == p.buf.end()) VOG_FAIL(-4, "no stage '%s'", stage);
  *offset = it->second.first;
  *bytes = it->second.second;
  return 0;
}

// ---- group language encoder -------------------------------------------------------------------
extern "C" int64_t vog_lang_workspace_bytes(const vog_ctx* c, int B, int ncmp, int T) {
  if (!c || !c->finalized || B <= 0 || ncmp <= 0 || T <= 0) return -1;
  return make_plan(c, make_geo(c->d, B, ncmp, T), true).total;
}

extern "C" int vog_lang_workspace_init(const vog_ctx* c, int B, int ncmp, int T, void* ws, size_t ws_bytes,
                                       void* stream) {
  VOG_CHECK_ARG(c && ws);
  const int64_t need = vog_lang_workspace_bytes(c, B, ncmp, T);
  if (need < 0 || (int64_t)ws_bytes < need) VOG_FAIL(-2, "language workspace too small");
  VOG_HIP(hipMemsetAsync(ws, 0, (size_t)need, (hipStream_t)stream));
  return 0;
}

extern "C" int vog_lang_outputs(const vog_ctx* c, int B, int ncmp, int T, void* ws, float** lang,
                                float** final_hidden) {
  VOG_CHECK_ARG(c && c->finalized && ws && B > 0 && ncmp > 0 && T > 0);
  const Geo g = make_geo(c->d, B, ncmp, T);
  Plan p = make_plan(c, g, true);
  WS w{(char*)ws, &p};
  if (lang) *lang = w.at<float>("lang");
  if (final_hidden) *final_hidden = w.at<float>("full") + (int64_t)g.Bn * g.T * g.L;
  return 0;
}

extern "C" int vog_lang_forward(vog_ctx* c, const vog_batch* lb, void* ws, size_t ws_bytes, void* stream) {
  VOG_CHECK_ARG(c && lb && ws);
  Plan plan;
  std::vector<Step> steps;
  VOG_TRY(build_steps(c, lb, ws, ws_bytes, plan, steps, true));
  for (auto& s : steps) VOG_TRY(s.fn((hipStream_t)stream));
  return 0;
}

// LSTMEncoder.forward (utils/mdl_srl_utils.py:114-169) on its own: token re-index + embedding + the packed 2-layer BiLSTM,
// both layers, both directions - the language chain of the forward up to (not including) lstm_out_feat_proj.
extern "C" int vog_bilstm_fwd(vog_ctx* c, const vog_batch* lb, void* ws, size_t ws_bytes, float* x_out, float* final_hidden,
                              void* stream) {
  VOG_CHECK_ARG(c && lb && ws && x_out && final_hidden);
  Plan plan;
  std::vector<Step> steps;
  VOG_TRY(build_steps(c, lb, ws, ws_bytes, plan, steps, true));
  for (auto& s : steps) {
    const bool lstm_part = s.name == "prep" || s.name == "lang_prep" || s.name.rfind("lstm_ih", 0) == 0 ||
                           s.name == "lstm_layer" || s.name == "lstm_step";
    if (lstm_part) VOG_TRY(s.fn((hipStream_t)stream));
  }
  const Geo g = make_geo(c->d, lb->B, lb->ncmp, lb->T);
  WS w{(char*)ws, &plan};
  const int top = c->d.rnn_layers - 1;
  const bool ofrag = (g.Bn * g.T + g.Bn) <= 64;
  return vog_lstm_out_to_f32(w.at<void>("lstm_out16_" + std::to_string(top)), ofrag ? 1 : 0, g.Bn * g.T, g.Bn, 2 * g.R,
                             (vog_dtype)c->d.enc_dtype, x_out, final_hidden, stream);
}

namespace vog {
// steps of a whole group: [language chain] + members' forwards; member_of[i] = -1 for language steps
static int build_group(vog_ctx* c, const vog_batch* lb, void* lws, size_t lbytes, const vog_batch* const* members,
                       void* const* wss, const size_t* wbytes, int n, std::vector<Step>& steps,
                       std::vector<int>& member_of) {
  VOG_CHECK_ARG(c && lb && lws && members && wss && wbytes && n >= 1 && n <= 16);
  Plan lp;
  VOG_TRY(build_steps(c, lb, lws, lbytes, lp, steps, true));
  member_of.assign(steps.size(), -1);
  int b_sum = 0;
  for (int m = 0; m < n; ++m) {
    VOG_CHECK_ARG(members[m] && wss[m] && members[m]->shared_lang && members[m]->ncmp == lb->ncmp);
    b_sum += members[m]->B;
    Plan mp;
    std::vector<Step> ms;
    VOG_TRY(build_steps(c, members[m], wss[m], wbytes[m], mp, ms));
    for (auto& s : ms) {
      if (s.branch < 0) continue;
      steps.push_back(s);
      member_of.push_back(m);
    }
  }
  if (b_sum != lb->B) VOG_FAIL(-1, "group: members hold %d queries, the language batch %d", b_sum, lb->B);
  return 0;
}
}  // namespace vog

extern "C" int vog_group_forward(vog_ctx* c, const vog_batch* lb, void* lws, size_t lbytes,
                                 const vog_batch* const* members, void* const* wss, const size_t* wbytes,
                                 int n, void* stream) {
  std::vector<Step> steps;
  std::vector<int> mo;
  VOG_TRY(vog::build_group(c, lb, lws, lbytes, members, wss, wbytes, n, steps, mo));
  for (auto& s : steps) VOG_TRY(s.fn((hipStream_t)stream));
  return 0;
}

extern "C" int vog_forward(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes, void* stream) {
  VOG_CHECK_ARG(c && b && ws);
  Plan plan;
  std::vector<Step> steps;
  VOG_TRY(build_steps(c, b, ws, ws_bytes, plan, steps));
  for (auto& s : steps) {               // eager: one stream, program order (re-entrant)
    if (s.branch < 0) continue;
    const int r = s.fn((hipStream_t)stream);
    if (r != 0) return r;
  }
  return 0;
}

struct vog_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

static int graph_capture_impl(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes, const vog_copy_seg* dma,
                              const vog_assemble_args* asm_args, const vog_copy_seg* segs, int nseg, void* stream, vog_graph** out) {
  VOG_CHECK_ARG(c && b && ws && out && stream);
  VOG_CHECK_ARG(nseg >= 0 && nseg <= VOG_MAX_COPY_SEGS && (nseg == 0 || segs));
  Plan plan;
  std::vector<Step> steps;
  VOG_TRY(build_steps(c, b, ws, ws_bytes, plan, steps));
  hipStream_t st = (hipStream_t)stream;
  // One linear chain. (A DAG form - the language chain captured as a parallel branch - was measured and
  // removed: graph branches of 4 forwards in flight oversubscribe the 4 hardware queues, 33 k instead of
  // 48 k queries/s, and replaying two such graphs on 2 streams crashed inside hipGraphLaunch on ROCm 7.2.
  // Steps that can run side by side share a launch instead: csrc/pair.hip.)
  VOG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  // fed graph: the batch's way onto the device first (kernel nodes reading pinned host memory at fixed addresses)
  if (dma && dma->bytes) {
    hipError_t de = hipMemcpyAsync(dma->dst, dma->src, dma->bytes, hipMemcpyHostToDevice, st);
    if (de != hipSuccess) { rc = -(int)de - 1000; vog::set_error("hipMemcpyAsync (fed graph): %s", hipGetErrorString(de)); }
  }
  if (rc == 0 && asm_args) rc = vog_assemble_batch(asm_args, st);
  if (rc == 0 && nseg > 0) rc = vog_copy_segments(segs, nseg, st);
  for (auto& s : steps) {
    if (rc != 0) break;
    if (s.branch < 0) continue;          // join marker
    rc = s.fn(st);
    if (rc != 0) break;
  }
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(st, &g);
  if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) VOG_FAIL(-(int)e - 1000, "hipStreamEndCapture: %s", hipGetErrorString(e));
  vog_graph* vg = new vog_graph();
  vg->graph = g;
  e = hipGraphInstantiate(&vg->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); delete vg; VOG_FAIL(-(int)e - 1000, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
  *out = vg;
  return 0;
}

extern "C" int vog_graph_capture(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes,
                                 void* stream, vog_graph** out) {
  return graph_capture_impl(c, b, ws, ws_bytes, nullptr, nullptr, nullptr, 0, stream, out);
}

extern "C" int vog_graph_capture_fed(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes, const vog_copy_seg* dma,
                                     const vog_assemble_args* asm_args, const vog_copy_seg* segs, int nseg, void* stream,
                                     vog_graph** out) {
  return graph_capture_impl(c, b, ws, ws_bytes, dma, asm_args, segs, nseg, stream, out);
}

extern "C" int vog_group_graph_capture(vog_ctx* c, const vog_batch* lb, void* lws, size_t lbytes,
                                       const vog_batch* const* members, void* const* wss,
                                       const size_t* wbytes, int n, void* stream, vog_graph** out) {
  VOG_CHECK_ARG(out && stream);
  std::vector<Step> steps;
  std::vector<int> mo;
  VOG_TRY(vog::build_group(c, lb, lws, lbytes, members, wss, wbytes, n, steps, mo));
  hipStream_t st = (hipStream_t)stream;
  VOG_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (auto& s : steps) { rc = s.fn(st); if (rc != 0) break; }
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(st, &g);
  if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) VOG_FAIL(-(int)e - 1000, "hipStreamEndCapture: %s", hipGetErrorString(e));
  vog_graph* vg = new vog_graph();
  vg->graph = g;
  e = hipGraphInstantiate(&vg->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); delete vg; VOG_FAIL(-(int)e - 1000, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
  *out = vg;
  return 0;
}

extern "C" int vog_ctx_set_int(vog_ctx* c, const char* name, int value) {
  VOG_CHECK_ARG(c && name);
  if (strcmp(name, "tx_split") == 0) {        // takes effect at the next vog_ctx_finalize (the remainder weights are made there)
    if ((value != 0) != (c->tx_split != 0)) c->finalized = false;
    c->tx_split = value ? 1 : 0;
    return 0;
  }
  if (strcmp(name, "lstm_persistent") == 0) { c->lstm_persistent = value ? 1 : 0; return 0; }
  if (strcmp(name, "lstm_inject_stall") == 0) { c->lstm_inject_stall = value == 2 ? 2 : (value ? 1 : 0); return 0; }   // 2: direction 1 only
  if (strcmp(name, "fused_tail") == 0) { c->fused_tail = value ? 1 : 0; return 0; }
  if (strcmp(name, "fused_enc") == 0) { c->fused_enc = value ? 1 : 0; return 0; }
  if (strcmp(name, "pair_launches") == 0) { c->pair_launches = value ? 1 : 0; return 0; }
  if (strcmp(name, "pair_mask") == 0) { c->pair_mask = value & 15; return 0; }
  if (strcmp(name, "fused_ih") == 0) { c->fused_ih = value; return 0; }
  if (strcmp(name, "enc_lean") == 0) { c->enc_lean = value; return 0; }
  VOG_FAIL(-4, "unknown option '%s'", name);
}

extern "C" int vog_graph_launch(vog_graph* g, void* stream) {
  VOG_CHECK_ARG(g && g->exec);
  VOG_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return 0;
}

extern "C" int vog_graph_destroy(vog_graph* g) {
  if (!g) return 0;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return 0;
}

extern "C" int vog_time_kernel(vog_ctx* c, const vog_batch* b, void* ws, size_t ws_bytes,
                               const char* kernel, int iters, void* stream, float* usec) {
  VOG_CHECK_ARG(c && b && ws && kernel && iters > 0 && usec);
  Plan plan;
  std::vector<Step> steps;
  VOG_TRY(build_steps(c, b, ws, ws_bytes, plan, steps));
  // "name#k": the k-th step of that name (the two BiLSTM layers share one)
  std::string want(kernel);
  int occ = 0;
  { const size_t h = want.find('#'); if (h != std::string::npos) { occ = atoi(want.c_str() + h + 1); want.resize(h); } }
  const Step* s = nullptr;
  auto pick = [&]() { int k = occ; for (auto& x : steps) if (x.name == want && x.branch >= 0 && k-- == 0) { s = &x; break; } };
  pick();
  if (!s) {     // a step that runs paired in the forward can still be timed on its own
    steps.clear();
    VOG_TRY(build_steps(c, b, ws, ws_bytes, plan, steps, false, false));
    pick();
  }
  if (!s) VOG_FAIL(-4, "no kernel step '%s'", kernel);
  hipStream_t st = (hipStream_t)stream;
  // the persistent layer kernel consumes per-forward state (hand-off tags zeroed by lang_prep):
  // time (lang_prep + layer) pairs and subtract lang_prep timed alone the same way
  const Step* reset = nullptr;
  if (s->name.rfind("lstm_layer", 0) == 0)
    for (auto& x : steps) if (x.name == "lang_prep" || x.name == "prep") { reset = &x; break; }
  hipEvent_t e0, e1;
  VOG_HIP(hipEventCreate(&e0));
  VOG_HIP(hipEventCreate(&e1));
  auto timed = [&](const Step* a, const Step* b2, float* out_ms) -> int {
    for (int i = 0; i < 3; ++i) { if (a) VOG_TRY(a->fn(st)); if (b2) VOG_TRY(b2->fn(st)); }
    VOG_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) { if (a) VOG_TRY(a->fn(st)); if (b2) VOG_TRY(b2->fn(st)); }
    VOG_HIP(hipEventRecord(e1, st));
    VOG_HIP(hipEventSynchronize(e1));
    VOG_HIP(hipEventElapsedTime(out_ms, e0, e1));
    return 0;
  };
  float ms = 0.f, ms_reset = 0.f;
  VOG_TRY(timed(reset, s, &ms));
  if (reset) VOG_TRY(timed(reset, nullptr, &ms_reset));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *usec = (ms - ms_reset) * 1000.0f / (float)iters;
  return 0;
}
