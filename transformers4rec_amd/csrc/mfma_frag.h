// Fragment helpers for single-wave attention kernels on the fp32 matrix cores
// (xlnet_attn_mfma.hip, mha_mfma.hip).  MFMA operand conventions (lane l: c = l & 31, kh = l >> 5):
//   D[i][j] += sum_ks A[i][ks] * B[ks][j]; the lane supplies A[i = c][ks = kh], B[ks = kh][j = c]
//   and holds D[(r & 3) + 8 * (r >> 2) + 4 * kh][c] in accumulator register r.
//   The k-slots are permuted: step s of a contraction of length K takes k = kh * K/2 + s, so a
//   "row fragment" (k along a row of a row-major matrix) is a contiguous run per lane, a matrix held
//   one row per lane pair ("row layout": lane (i, kh) owns columns kh*16 .. kh*16+15) IS the A
//   operand of a contraction over its columns, and a "column fragment" (k = row index) is a
//   128-byte coalesced scalar load per step.
#pragma once
#include "t4r_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define XM_SP 33      // pitch of the [32][32] exchange buffer (odd: conflict free by rows and by columns)
#define XM_RP 66      // pitch of the [32][64] raw / d raw buffer (RP - 1 odd: the shifted gather is conflict free)

__device__ __forceinline__ int xm_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

template <int N>
struct Frag { float v[N]; };

// Addresses are a wave-uniform base pointer (session start) plus a 32-bit per-lane element offset, so
// the loads use the scalar-base form and an offset costs one VGPR.
// row fragment: base[off + s], s < DH/2 (k along the head dimension; off = row*D + hc + kh*DH/2)
template <int DH>
__device__ __forceinline__ Frag<DH / 2> row_frag(const float* base, int off) {
    Frag<DH / 2> f;
    const float* p = base + off;
#pragma unroll
    for (int s = 0; s < DH / 2; s += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + s);
        f.v[s] = t.x; f.v[s + 1] = t.y; f.v[s + 2] = t.z; f.v[s + 3] = t.w;
    }
    return f;
}
// column fragment: base[min(kh*KH + s, nrows-1)*D + col], s < KH (rows >= nrows clamped: the other
// operand is zero there)
template <int KH>
__device__ __forceinline__ Frag<KH> col_frag(const float* base, int col, int D, int nrows, int kh, float add) {
    Frag<KH> f;
#pragma unroll
    for (int s = 0; s < KH; ++s) f.v[s] = base[min(kh * KH + s, nrows - 1) * D + col] + add;
    return f;
}
template <int N>
__device__ __forceinline__ void mfma_chain(f32x16& acc, const Frag<N>& a, const Frag<N>& b) {
#pragma unroll
    for (int s = 0; s < N; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

