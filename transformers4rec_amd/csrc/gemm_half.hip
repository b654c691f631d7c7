// bf16 / fp16 matrix-core instantiations of the GEMM kernel (gemm_kernel.h, PREC = 1, 2, 3), in their own
// translation unit so that they compile next to the fp32 ones.  Selected by gemm_f32.hip: launch_layout.
//   PREC 1: fp32-accurate products by the exact three-way bf16 split (six v_mfma_f32_32x32x16_bf16 per K = 16)
//   PREC 4: fp32-class products by the two-way fp16 split (three v_mfma_f32_32x32x16_f16 per K = 16), operands positioned by
//           powers of two from their maxima (the weight gradients of the fused XLNet layer)
//   PREC 2 / 3: mixed precision (bf16 / fp16 operands, fp32 accumulation) -- the reference's AMP mode,
//               transformers4rec/torch/trainer.py:363-367, model/prediction_task.py:430
// Tiles: 64 x 64 x 32 for every variant (three-plane images: 52 KB of LDS, three workgroups per CU),
// 128 x 128 x 32 for the one-plane precisions on plain launches.
#include "gemm_kernel.h"

template <int BM, int BN, bool TA, bool TB, int PREC>
static int half_cfg(const GemmParams& p, int batch, hipStream_t stream) {
    if (p.sg_lse) {
        if constexpr (BM == 64 && BN == 64 && !TB) return launch_vec<64, 64, 32, TA, false, 1, true, PREC>(p, batch, stream);
        if constexpr (BM == 128 && BN == 64 && !TB && PREC == 1) return launch_vec<128, 64, 32, TA, false, 1, true, 1>(p, batch, stream);
        t4r_set_error("gemm (half): softmax-grad operand needs the 64x64 (or, split form, 128x64) tile and transB = 0");
        return -1;
    }
    if (p.drop.p > 0.f && (p.epilogue == EPI_BIAS_GELU || p.epilogue == EPI_BIAS_RESID)) {
        if constexpr (BM == 64 && BN == 64) return launch_vec<64, 64, 32, TA, TB, 2, true, PREC>(p, batch, stream);
        t4r_set_error("gemm (half): epilogue dropout needs the 64x64 tile");
        return -1;
    }
    return launch_vec<BM, BN, 32, TA, TB, 0, true, PREC>(p, batch, stream);
}

template <bool TA, bool TB>
static int half_layout(const GemmParams& p, int batch, int big, int prec, hipStream_t stream) {
    if (prec == 1) {
        if (big == 2) return half_cfg<128, 64, TA, TB, 1>(p, batch, stream);
        return big ? half_cfg<128, 128, TA, TB, 1>(p, batch, stream) : half_cfg<64, 64, TA, TB, 1>(p, batch, stream);
    }
    if (prec == 2) return big ? half_cfg<128, 128, TA, TB, 2>(p, batch, stream) : half_cfg<64, 64, TA, TB, 2>(p, batch, stream);
    if (prec == 3) return big ? half_cfg<128, 128, TA, TB, 3>(p, batch, stream) : half_cfg<64, 64, TA, TB, 3>(p, batch, stream);
    if (prec == 4) {        // two-way fp16 split with operand scales (plain launches only: the layer's weight gradients)
        if (p.sg_lse || p.drop.p > 0.f || !p.amaxA || !p.amaxB) { t4r_set_error("gemm (fp16 split): plain products with operand maxima only"); return -1; }
        return launch_vec<64, 64, 32, TA, TB, 0, true, 4>(p, batch, stream);
    }
    t4r_set_error("gemm (half): unknown precision");
    return -1;
}

int t4r_gemm_half_dispatch(const GemmParams& p, int batch, int ta, int tb, int big, int prec, hipStream_t stream) {
    if (ta) return tb ? half_layout<true, true>(p, batch, big, prec, stream) : half_layout<true, false>(p, batch, big, prec, stream);
    return tb ? half_layout<false, true>(p, batch, big, prec, stream) : half_layout<false, false>(p, batch, big, prec, stream);
}
