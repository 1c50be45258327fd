// Device-side SPAT / TEMP batch assembly: verb_item_getter_SPAT / _TEMP
// (code/dat_loader_simple.py:1046-1338) for a whole batch of queries, writing straight into the
// forward's / loss's input buffers. Pure layout work on the 2 MB/query feature block:
//   SPAT: proposals x1, x2 += 720 * video, rows re-ordered (video, frame, prop) -> (frame, video, prop);
//         region features and the proposal padding mask re-ordered the same way; per-frame segment
//         features (video, frame) -> (frame, video)
//   TEMP: proposal frame index += 10 * video; everything else is the plain concatenation
//   both: ground-truth boxes shifted the same way, the first num_box[v] of every video concatenated
//         and zero padded (the reference's gt[0, 0] fallback when there are none); srl_boxes shifted
//         by the boxes in front of the target video where srl_boxes_lens > 0; frm_mask[r, g] =
//         (frame(proposal r) != frame(gt g)) for g < total boxes, else 1.
// HBM bound: every byte is read once and written once with 16-byte accesses (feature rows) by
// `assemble_rows_kernel`; `assemble_gt_kernel` (one workgroup per query) does the KB-sized part.
// Bit-exact vs the reference (fp32 adds of small integers * 720 / * 10, otherwise copies).
#include "common.h"

namespace vog {

__global__ __launch_bounds__(256) void assemble_rows_kernel(vog_assemble_args a) {
  const int NPv = a.nfrm0 * a.nppf0;
  const int64_t n_prop_rows = (int64_t)a.B * a.ncmp * NPv;
  const int64_t n_seg_rows = (int64_t)a.B * a.ncmp * a.nfrm0;
  const int64_t row = blockIdx.x;
  const bool spat = a.conc_type == VOG_CONC_SPAT;
  if (row < n_prop_rows) {
    // source row (b, v, f, p)
    const int b = (int)(row / ((int64_t)a.ncmp * NPv));
    int r = (int)(row - (int64_t)b * a.ncmp * NPv);
    const int v = r / NPv; r -= v * NPv;
    const int f = r / a.nppf0, p = r - f * a.nppf0;
    const int64_t dst = (int64_t)b * a.ncmp * NPv +
                        (spat ? ((int64_t)f * a.ncmp + v) * a.nppf0 + p : (int64_t)v * NPv + f * a.nppf0 + p);
    const float4* s4 = reinterpret_cast<const float4*>(a.region_in + row * a.prop_dim);
    float4* d4 = reinterpret_cast<float4*>(a.region_out + dst * a.prop_dim);
    for (int i = threadIdx.x; i < a.prop_dim / 4; i += 256) d4[i] = s4[i];
    if (threadIdx.x < 7) {
      float x = a.props_in[row * 7 + threadIdx.x];
      const int c = threadIdx.x;
      if (spat && (c == 0 || c == 2)) x = x + (float)v * a.vid_w;
      if (!spat && c == 4) x = x + (float)v * (float)a.nfrm0;
      a.props_out[dst * 7 + c] = x;
    }
    if (threadIdx.x == 7 && a.pnt_in) a.pnt_out[dst] = a.pnt_in[row];
    return;
  }
  const int64_t srow = row - n_prop_rows;
  if (srow >= n_seg_rows) return;
  const int b = (int)(srow / ((int64_t)a.ncmp * a.nfrm0));
  const int r = (int)(srow - (int64_t)b * a.ncmp * a.nfrm0);
  const int v = r / a.nfrm0, f = r - v * a.nfrm0;
  const int64_t dst = (int64_t)b * a.ncmp * a.nfrm0 + (spat ? (int64_t)f * a.ncmp + v : (int64_t)r);
  const float4* s4 = reinterpret_cast<const float4*>(a.seg_in + srow * a.seg_dim);
  float4* d4 = reinterpret_cast<float4*>(a.seg_out + dst * a.seg_dim);
  for (int i = threadIdx.x; i < a.seg_dim / 4; i += 256) d4[i] = s4[i];
}

// one workgroup per query: gt boxes, srl_boxes, frame mask (reads props_out of THIS query: launched after
// assemble_rows_kernel on the same stream)
__global__ __launch_bounds__(256) void assemble_gt_kernel(vog_assemble_args a) {
  __shared__ int cum[65];
  __shared__ float gfrm[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool spat = a.conc_type == VOG_CONC_SPAT;
  const int NPt = a.ncmp * a.nfrm0 * a.nppf0;
  if (tid == 0) {
    cum[0] = 0;
    for (int v = 0; v < a.ncmp; ++v) cum[v + 1] = cum[v] + (int)a.num_box[(int64_t)b * a.ncmp + v];
  }
  __syncthreads();
  const int total = cum[a.ncmp];
  const int n_rows = total > 0 ? total : 1;                 // the reference's gt[0, 0] fallback
  float* gout = a.gt_out + (int64_t)b * a.G * 5;
  for (int i = tid; i < a.G * 5; i += 256) {
    const int g = i / 5, c = i - g * 5;
    float x = 0.f;
    if (g < n_rows && g < a.G) {
      int v = 0, k = 0;
      if (total > 0) { while (g >= cum[v + 1]) ++v; k = g - cum[v]; }
      x = a.gt_in[(((int64_t)b * a.ncmp + v) * a.G + k) * 5 + c];
      if (spat && (c == 0 || c == 2)) x = x + (float)v * a.vid_w;
      if (!spat && c == 4) x = x + (float)v * (float)a.nfrm0;
    }
    gout[i] = x;
    if (c == 4 && g < 1024) gfrm[g] = x;
  }
  if (tid == 0) a.num_box_out[b] = total;
  const int shift = cum[(int)a.target_cmp[b]];
  const int nsb = a.nv * a.nsrl * a.nbox;
  for (int i = tid; i < nsb; i += 256) {
    const int64_t j = (int64_t)b * nsb + i;
    a.srl_boxes_out[j] = a.srl_boxes_in[j] + (a.srl_boxes_lens[j] > 0 ? shift : 0);
  }
  __syncthreads();
  unsigned char* fm = a.frm_out + (int64_t)b * NPt * a.G;
  const float* pout = a.props_out + (int64_t)b * NPt * 7;
  for (int i = tid; i < NPt * a.G; i += 256) {
    const int r = i / a.G, g = i - r * a.G;
    fm[i] = g < total ? (unsigned char)(pout[(int64_t)r * 7 + 4] != gfrm[g]) : (unsigned char)1;
  }
}

// byte ranges src -> dst, one launch: blockIdx.y = segment, the blocks of a segment stride over its 16-byte chunks.
// The sources may be PINNED HOST memory (mapped into the device's address space): the loads then cross the host link,
// which wants many 16-byte requests in flight and nothing else - no staging copy, no DMA set-up latency.
struct CopySegs { vog_copy_seg s[VOG_MAX_COPY_SEGS]; };
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs cs) {
  const vog_copy_seg sg = cs.s[blockIdx.y];
  const size_t n16 = sg.bytes >> 4;
  const u32x4* s4 = reinterpret_cast<const u32x4*>(sg.src);
  u32x4* d4 = reinterpret_cast<u32x4*>(sg.dst);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d4[i] = s4[i];
  if (blockIdx.x == 0) {
    const unsigned char* sb = reinterpret_cast<const unsigned char*>(sg.src);
    unsigned char* db = reinterpret_cast<unsigned char*>(sg.dst);
    for (size_t i = (n16 << 4) + threadIdx.x; i < sg.bytes; i += 256) db[i] = sb[i];
  }
}

}  // namespace vog

extern "C" int vog_copy_segments(const vog_copy_seg* segs, int n, void* stream) {
  using namespace vog;
  VOG_CHECK_ARG(n >= 0 && n <= VOG_MAX_COPY_SEGS && (n == 0 || segs));
  if (n == 0) return 0;
  CopySegs cs;
  size_t mx = 0;
  for (int i = 0; i < n; ++i) {
    VOG_CHECK_ARG(segs[i].src && segs[i].dst && ((uintptr_t)segs[i].src & 15) == 0 && ((uintptr_t)segs[i].dst & 15) == 0);
    cs.s[i] = segs[i];
    mx = segs[i].bytes > mx ? segs[i].bytes : mx;
  }
  size_t gx = (mx + 4095) / 4096;                          // one 16-byte chunk per thread for the largest segment ...
  gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);               // ... up to 1024 blocks (4 MB in flight per pass)
  ::vog::launch(copy_segments_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, cs);
  VOG_LAUNCH_CHECK();
  return 0;
}

extern "C" int vog_assemble_batch(const vog_assemble_args* a, void* stream) {
  using namespace vog;
  VOG_CHECK_ARG(a && a->props_in && a->props_out && a->region_in && a->region_out && a->seg_in && a->seg_out);
  VOG_CHECK_ARG(a->conc_type == VOG_CONC_SPAT || a->conc_type == VOG_CONC_TEMP);
  VOG_CHECK_ARG(a->B > 0 && a->ncmp > 0 && a->ncmp <= 64 && a->nfrm0 > 0 && a->nppf0 > 0 &&
                (a->prop_dim % 4) == 0 && (a->seg_dim % 4) == 0);
  VOG_CHECK_ARG((a->pnt_in == nullptr) == (a->pnt_out == nullptr));
  const int64_t rows = (int64_t)a->B * a->ncmp * a->nfrm0 * (a->nppf0 + 1);
  ::vog::launch(assemble_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, *a);
  VOG_LAUNCH_CHECK();
  if (a->gt_in) {
    VOG_CHECK_ARG(a->gt_out && a->num_box && a->num_box_out && a->target_cmp && a->srl_boxes_in && a->srl_boxes_out &&
                  a->srl_boxes_lens && a->frm_out && a->G > 0 && a->G <= 1024 && a->nv > 0 && a->nsrl > 0 && a->nbox > 0);
    ::vog::launch(assemble_gt_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a);
    VOG_LAUNCH_CHECK();
  }
  return 0;
}
