// EmbeddingBag branch of the input block: a bag of ids per row -> ONE combined row.
//
// Reference behaviour restated:
//   EmbeddingFeatures.forward, 2-D / (values, offsets) / 1-D inputs   transformers4rec/torch/features/embedding.py:226-240
//   EmbeddingFeatures.table_to_embedding_module -> EmbeddingBagWrapper(mode = TableConfig.combiner)   :86-93, 260-273
//   TableConfig.combiner in {"mean", "sum", "sqrtn"}                    :416-460
// i.e. torch.nn.EmbeddingBag WITHOUT a padding index: id 0 is an ordinary row, an empty bag gives a zero row,
//   sum: out[b] = sum_i table[id_i]      mean: / n_b      sqrtn: / sqrt(n_b)   (n_b = bag size)
// Bags arrive either as a fixed-width id matrix [B, K] (K = 1 for the 1-D form) or ragged as
// (values [n], offsets [B]) where bag b = values[offsets[b] : offsets[b+1]) and the last bag runs to n.
//
// HBM-bound gather: one lane group per bag, 16-byte row segments, four member rows requested before the
// first is consumed.  Algorithmic bytes: n*(8 id + 4*dim row) + B*4*dim out.
// The backward is the transpose: every member lookup receives scale_b * dout[b]; that [n, dim] expansion
// (`embedding_bag_bwd_rows_kernel`) feeds the deterministic sorted scatter of embedding_sorted.hip (padding_idx -1),
// so bag tables get the same bit-reproducible gradient as the sequence tables.
#include "t4r_common.h"

enum { BAG_SUM = 0, BAG_MEAN = 1, BAG_SQRTN = 2 };

__device__ __forceinline__ float bag_scale(int combiner, long n) {
    if (combiner == BAG_SUM || n <= 0) return 1.f;
    return combiner == BAG_MEAN ? 1.f / (float)n : rsqrtf((float)n);
}

// GROUP lanes per bag; lane gl owns the float4 columns gl*4 + j*GROUP*4 (j < NV)
template <int GROUP, int NV>
__global__ __launch_bounds__(256) void embedding_bag_fwd_kernel(const float* __restrict__ table, long rows, int dim,
                                                                 const long* __restrict__ values,
                                                                 const long* __restrict__ offsets, long n_bags,
                                                                 long n_values, int fixed_k, int combiner,
                                                                 float* __restrict__ out, long ld_out, int col,
                                                                 int* __restrict__ err) {
    const long bag = ((long)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
    const int gl = threadIdx.x & (GROUP - 1);
    if (bag >= n_bags) return;
    long p0, p1;
    if (offsets) {
        p0 = offsets[bag];
        p1 = bag + 1 < n_bags ? offsets[bag + 1] : n_values;
    } else {
        p0 = bag * fixed_k;
        p1 = p0 + fixed_k;
    }
    float4 acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    bool bad = false;
    for (long p = p0; p < p1; p += 4) {
        long id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) id[u] = p + u < p1 ? values[p + u] : -1;
        float4 r[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = p + u < p1;
            const bool ok = in && id[u] >= 0 && id[u] < rows;
            bad |= in && !ok;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = gl * 4 + j * GROUP * 4;
                r[u][j] = (ok && c < dim) ? *reinterpret_cast<const float4*>(table + id[u] * (long)dim + c)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // fixed order p, p+1, ...: the sum does not depend on the launch geometry
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                acc[j].x += r[u][j].x; acc[j].y += r[u][j].y; acc[j].z += r[u][j].z; acc[j].w += r[u][j].w;
            }
    }
    if (bad && err) *err = 1;
    const float s = bag_scale(combiner, p1 - p0);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = gl * 4 + j * GROUP * 4;
        if (c < dim)
            *reinterpret_cast<float4*>(out + bag * ld_out + col + c) =
                make_float4(acc[j].x * s, acc[j].y * s, acc[j].z * s, acc[j].w * s);
    }
}

// scalar-column variant for dims that are not multiples of 4 (or unaligned column offsets)
__global__ __launch_bounds__(256) void embedding_bag_fwd_scalar_kernel(const float* __restrict__ table, long rows, int dim,
                                                                        const long* __restrict__ values,
                                                                        const long* __restrict__ offsets, long n_bags,
                                                                        long n_values, int fixed_k, int combiner,
                                                                        float* __restrict__ out, long ld_out, int col,
                                                                        int* __restrict__ err) {
    const long bag = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (bag >= n_bags) return;
    long p0, p1;
    if (offsets) {
        p0 = offsets[bag];
        p1 = bag + 1 < n_bags ? offsets[bag + 1] : n_values;
    } else {
        p0 = bag * fixed_k;
        p1 = p0 + fixed_k;
    }
    const float s = bag_scale(combiner, p1 - p0);
    bool bad = false;
    for (int c = lane; c < dim; c += 64) {
        float a = 0.f;
        for (long p = p0; p < p1; ++p) {
            const long id = values[p];
            const bool ok = id >= 0 && id < rows;
            bad |= !ok;
            if (ok) a += table[id * (long)dim + c];
        }
        out[bag * ld_out + col + c] = a * s;
    }
    if (bad && err) *err = 1;
}

// backward expansion: rows_out[i, :] = scale(bag of i) * dout[bag of i, col : col + dim]   (i = member lookup)
__global__ __launch_bounds__(256) void embedding_bag_bwd_rows_kernel(const float* __restrict__ dout, long ld, int col,
                                                                      int dim, const long* __restrict__ offsets,
                                                                      long n_bags, long n_values, int fixed_k,
                                                                      int combiner, float* __restrict__ rows_out) {
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n_values) return;
    long bag, n;
    if (offsets) {
        // last bag with offsets[bag] <= i (empty bags share an offset with their successor: skip to the last of them)
        long lo = 0, hi = n_bags - 1;
        while (lo < hi) {
            const long mid = (lo + hi + 1) >> 1;
            if (offsets[mid] <= i) lo = mid; else hi = mid - 1;
        }
        bag = lo;
        n = (bag + 1 < n_bags ? offsets[bag + 1] : n_values) - offsets[bag];
    } else {
        bag = i / fixed_k;
        n = fixed_k;
    }
    const float s = bag_scale(combiner, n);
    for (int c = lane; c < dim; c += 64) rows_out[i * (long)dim + c] = s * dout[bag * ld + col + c];
}

// values: int64 [n_values] (ragged form) or [n_bags * fixed_k] (matrix form, offsets == NULL); out row pitch ld_out floats,
// the bag rows land in columns [col, col + dim).  err (int*, may be NULL) is set to 1 on an id outside [0, rows).
extern "C" int t4r_embedding_bag_fwd(void* stream, const float* table, long rows, int dim, const long* values,
                                     const long* offsets, long n_bags, long n_values, int fixed_k, int combiner,
                                     float* out, long ld_out, int col, int* err) {
    if (n_bags <= 0) return 0;
    T4R_CHECK_ARG(table && values && out, "embedding_bag_fwd: null pointer");
    T4R_CHECK_ARG(combiner >= BAG_SUM && combiner <= BAG_SQRTN, "embedding_bag_fwd: combiner must be 0 sum, 1 mean, 2 sqrtn");
    T4R_CHECK_ARG(offsets || (fixed_k > 0 && n_values == n_bags * (long)fixed_k), "embedding_bag_fwd: matrix form needs n_values == n_bags * fixed_k");
    T4R_CHECK_ARG(dim > 0 && rows > 0 && ld_out >= col + dim, "embedding_bag_fwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = dim % 4 == 0 && col % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)table % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                     dim <= 1024;
    if (!vec) {
        hipLaunchKernelGGL(embedding_bag_fwd_scalar_kernel, dim3((unsigned)((n_bags + 3) / 4)), dim3(256), 0, st, table, rows,
                           dim, values, offsets, n_bags, n_values, fixed_k, combiner, out, ld_out, col, err);
        T4R_LAUNCH_CHECK();
        return 0;
    }
    const int v4 = dim / 4;
#define BAG_LAUNCH(G, NV)                                                                                          \
    hipLaunchKernelGGL((embedding_bag_fwd_kernel<G, NV>), dim3((unsigned)((n_bags * G + 255) / 256)), dim3(256), 0, st, table, \
                       rows, dim, values, offsets, n_bags, n_values, fixed_k, combiner, out, ld_out, col, err)
    if (v4 <= 4) BAG_LAUNCH(4, 1);
    else if (v4 <= 8) BAG_LAUNCH(8, 1);
    else if (v4 <= 16) BAG_LAUNCH(16, 1);
    else if (v4 <= 32) BAG_LAUNCH(32, 1);
    else if (v4 <= 64) BAG_LAUNCH(64, 1);
    else if (v4 <= 128) BAG_LAUNCH(64, 2);
    else BAG_LAUNCH(64, 4);
#undef BAG_LAUNCH
    T4R_LAUNCH_CHECK();
    return 0;
}

// rows_out [n_values, dim]: the gradient row of every member lookup, in lookup order
extern "C" int t4r_embedding_bag_bwd_rows(void* stream, const float* dout, long ld, int col, int dim,
                                          const long* offsets, long n_bags, long n_values, int fixed_k, int combiner,
                                          float* rows_out) {
    if (n_values <= 0) return 0;
    T4R_CHECK_ARG(dout && rows_out, "embedding_bag_bwd_rows: null pointer");
    T4R_CHECK_ARG(combiner >= BAG_SUM && combiner <= BAG_SQRTN, "embedding_bag_bwd_rows: combiner must be 0 sum, 1 mean, 2 sqrtn");
    T4R_CHECK_ARG(offsets || fixed_k > 0, "embedding_bag_bwd_rows: matrix form needs fixed_k");
    hipLaunchKernelGGL(embedding_bag_bwd_rows_kernel, dim3((unsigned)((n_values + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dout,
                       ld, col, dim, offsets, n_bags, n_values, fixed_k, combiner, rows_out);
    T4R_LAUNCH_CHECK();
    return 0;
}
