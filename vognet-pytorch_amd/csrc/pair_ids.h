// Host-side identities (kernel stub addresses) of the kernels that pair.hip can fuse pairwise into
// one launch. Each function is defined in the translation unit that owns the kernel, so that
// pair.hip does not instantiate the stand-alone kernels a second time.
#pragma once
namespace vog {
const void* kid_gemm_skinny_f16();          // gemm_skinny<F16, false, 8, 1, 4>   (M <= 64 projections)
const void* kid_gemm_skinny_wide_f16();     // gemm_skinny<F16, false, 8, 2, 8>   (N >= 8192)
const void* kid_gemm_pipe_qkv(int dtype);   // gemm_pipe<T16, 64, 64, 2, EPI_QKV>
const void* kid_gemm_pipe_plain3_f16();     // gemm_pipe<F16, 64, 64, 3, EPI_PLAIN>  (BiLSTM input projections beyond 80 columns)
const void* kid_lstm_layer_f16();           // lstm_layer_kernel<F16, 32>
const void* kid_tx_tail_512(int dtype);     // tx_tail_kernel<T16, F16, 2, false, 0>
const void* kid_vis_enc_f16();              // vis_enc_kernel<F16>
const void* kid_vis_enc_stream_f16();       // vis_enc_stream_kernel<F16>
// hi + lo operand forms (round 6, f16)
const void* kid_gemm_pipe_qkv_split_f16();  // gemm_pipe<F16, 64, 64, 2, EPI_QKV, true>
const void* kid_tx_tail_split_512_f16();    // tx_tail_split_kernel<F16, F16, 2>
const void* kid_vis_enc_stream_split_f16(); // vis_enc_stream_kernel<F16, true>
}  // namespace vog
