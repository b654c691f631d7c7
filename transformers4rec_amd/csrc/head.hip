// Next-item head: softmax cross-entropy over the item vocabulary (forward + backward) on the
// materialised logits [N, V] (the reference returns them as "predictions"), the mean
// reduction, the sampled-softmax logit assembly, and top-k for the inference/eval path.
//
// Reference behaviour restated (transformers4rec/torch/model/prediction_task.py):
//   logits = X @ W^T ; torch.div(logits, T)      :664-669   (GEMM, alpha = 1/T: gemm_f32.hip)
//   loss = torch.nn.CrossEntropyLoss()(logits,y) :347,446   mean over the N label rows
//   label smoothing variant                      transformers4rec/torch/losses.py:4-20
//   sampled softmax logits                       :673-696
//   top-k at inference                           :466-470
// HBM-bound: the [N, V] matrix is read exactly once per pass with 16-byte loads; one
// workgroup per row, online (max, sum-exp) per thread then a workgroup reduction.
#include "t4r_common.h"

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { m = mn; s = 0.f; return; }
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

#ifndef T4R_CE_UNROLL
#define T4R_CE_UNROLL 4
#endif
// loss_row[i] = lse_i - (1-eps)*logit[i,y_i] - eps/V * sum_j logit[i,j]   (eps = label smoothing)
__global__ __launch_bounds__(256) void softmax_ce_fwd_kernel(const float* __restrict__ logits,
                                                              const long* __restrict__ labels,
                                                              float* __restrict__ loss_row,
                                                              float* __restrict__ lse_out, int N,
                                                              int V, long ld, float smoothing) {
    const int row = blockIdx.x;
    const float* x = logits + (long)row * ld;
    float m = -INFINITY, s = 0.f, tot = 0.f;
    const bool vec = (ld % 4 == 0) && ((uintptr_t)logits % 16 == 0);
    if (vec) {
        const int v4 = V / 4;
        int i = threadIdx.x;
#if T4R_CE_UNROLL == 4
        for (; i + 768 < v4; i += 1024) {     // four 16-byte loads in flight per thread, one rescale per 16 elements
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(x + 4 * (i + 256 * u));
            float mx = m;
#pragma unroll
            for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fmaxf(t[u].x, t[u].y), fmaxf(t[u].z, t[u].w)));
            float e = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                e += (__expf(t[u].x - mx) + __expf(t[u].y - mx)) + (__expf(t[u].z - mx) + __expf(t[u].w - mx));
                tot += (t[u].x + t[u].y) + (t[u].z + t[u].w);
            }
            s = s * __expf(m - mx) + e;
            m = mx;
        }
#endif
        for (; i + 256 < v4; i += 512) {      // two 16-byte loads in flight per thread
            const float4 t = *reinterpret_cast<const float4*>(x + 4 * i);
            const float4 u = *reinterpret_cast<const float4*>(x + 4 * (i + 256));
            const float mx = fmaxf(fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)), fmaxf(fmaxf(u.x, u.y), fmaxf(u.z, u.w)));
            const float mn = fmaxf(m, mx);
            s = s * __expf(m - mn) + (__expf(t.x - mn) + __expf(t.y - mn) + __expf(t.z - mn) + __expf(t.w - mn)) +
                (__expf(u.x - mn) + __expf(u.y - mn) + __expf(u.z - mn) + __expf(u.w - mn));
            m = mn;
            tot += (t.x + t.y + t.z + t.w) + (u.x + u.y + u.z + u.w);
        }
        for (; i < v4; i += 256) {
            const float4 t = *reinterpret_cast<const float4*>(x + 4 * i);
            const float mx = fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w));
            const float mn = fmaxf(m, mx);
            s = s * __expf(m - mn) + __expf(t.x - mn) + __expf(t.y - mn) + __expf(t.z - mn) + __expf(t.w - mn);
            m = mn;
            tot += t.x + t.y + t.z + t.w;
        }
        for (int i = v4 * 4 + threadIdx.x; i < V; i += 256) {
            const float t = x[i];
            const float mn = fmaxf(m, t);
            s = s * __expf(m - mn) + __expf(t - mn);
            m = mn;
            tot += t;
        }
    } else {
        for (int i = threadIdx.x; i < V; i += 256) {
            const float t = x[i];
            const float mn = fmaxf(m, t);
            s = s * __expf(m - mn) + __expf(t - mn);
            m = mn;
            tot += t;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
        tot += __shfl_xor(tot, o, 64);
    }
    __shared__ float sm[4], ss[4], st[4];
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; ss[threadIdx.x >> 6] = s; st[threadIdx.x >> 6] = tot; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { online_merge(m, s, sm[w], ss[w]); tot += st[w]; }
        const float lse = m + __logf(s);
        const long y = labels[row];
        float loss = lse - x[y];
        if (smoothing > 0.f) loss = (1.f - smoothing) * loss + smoothing * (lse - tot / V);
        loss_row[row] = loss;
        lse_out[row] = lse;
    }
}

// deterministic mean of n floats (single workgroup)
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ x, int n,
                                                     float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = wave_sum(s);
    __shared__ float sm[16];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += sm[w];
        *out = n > 0 ? t / n : 0.f;
    }
}

int t4r_mean_launch(hipStream_t stream, const float* x, int n, float* out) {      // used by head_split.hip
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, stream, x, n, out);
    T4R_LAUNCH_CHECK();
    return 0;
}

extern "C" int t4r_softmax_ce_fwd(void* stream, const float* logits, const long* labels,
                                  float* loss_rows, float* lse, float* loss_mean, int N, int V, long ld,
                                  float label_smoothing) {
    hipStream_t st = (hipStream_t)stream;
    if (N > 0) {
        hipLaunchKernelGGL(softmax_ce_fwd_kernel, dim3(N), dim3(256), 0, st, logits, labels, loss_rows,
                           lse, N, V, ld, label_smoothing);
    }
    if (loss_mean) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, st, loss_rows, N, loss_mean);
    T4R_LAUNCH_CHECK();
    return 0;
}

// dlogits[i,j] = gscale * (softmax_ij - (1-eps)*[j==y_i] - eps/V),  gscale = *gout / N (mean)
// columns V..ld-1 (padding of the leading dimension) are written as 0.
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float* __restrict__ logits,
                                                              const long* __restrict__ labels,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ gout,
                                                              float* __restrict__ dlogits, int N,
                                                              int V, long ld, float smoothing) {
    const int row = blockIdx.y;
    const float* x = logits + (long)row * ld;
    float* dx = dlogits + (long)row * ld;
    const float g = (gout ? *gout : 1.f) / N;
    const float l = lse[row];
    const int y = (int)labels[row];
    const float sub = smoothing / V;
    const bool vec = (ld % 4 == 0) && ((uintptr_t)logits % 16 == 0) && ((uintptr_t)dlogits % 16 == 0);
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= ld) return;
    if (vec && i0 + 4 <= V) {
        float4 t = *reinterpret_cast<const float4*>(x + i0);
        t.x = g * (__expf(t.x - l) - sub); t.y = g * (__expf(t.y - l) - sub);
        t.z = g * (__expf(t.z - l) - sub); t.w = g * (__expf(t.w - l) - sub);
        const float hit = g * (1.f - smoothing);
        if (y == i0) t.x -= hit; else if (y == i0 + 1) t.y -= hit;
        else if (y == i0 + 2) t.z -= hit; else if (y == i0 + 3) t.w -= hit;
        *reinterpret_cast<float4*>(dx + i0) = t;
    } else {
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e;
            if (i >= ld) break;
            float v = 0.f;
            if (i < V) {
                v = g * (__expf(x[i] - l) - sub);
                if (i == y) v -= g * (1.f - smoothing);
            }
            dx[i] = v;
        }
    }
}

extern "C" int t4r_softmax_ce_bwd(void* stream, const float* logits, const long* labels,
                                  const float* lse, const float* grad_out, float* dlogits, int N, int V,
                                  long ld, float label_smoothing) {
    if (N == 0) return 0;
    dim3 grid((unsigned)((ld + 1023) / 1024), N);
    hipLaunchKernelGGL(softmax_ce_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, labels,
                       lse, grad_out, dlogits, N, V, ld, label_smoothing);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// sampled softmax (prediction_task.py:673-696): logits[N, 1+S]
//   col 0   : sum(x * W[y]) - log(q[y] + 1e-16)
//   col 1+s : x . W[neg_s]  - log(q[neg_s] + 1e-16), or finfo(fp16).min/100 on accidental hits
// then / T.  One wave per (row): the S+1 dot products of length D.
__global__ __launch_bounds__(256) void sampled_logits_kernel(
    const float* __restrict__ x, const long* __restrict__ y, const float* __restrict__ W,
    const long* __restrict__ neg, const float* __restrict__ qdist, float* __restrict__ out, int N,
    int D, int S, float inv_t) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* xr = x + (long)row * D;
    const long yi = y[row];
    for (int c = 0; c <= S; ++c) {
        const long id = c == 0 ? yi : neg[c - 1];
        const float* wr = W + id * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += xr[d] * wr[d];
        s = wave_sum(s);
        if (lane == 0) {
            float v = s - __logf(qdist[id] + 1e-16f);
            if (c > 0 && id == yi) v = -65504.0f / 100.0f;
            out[(long)row * (S + 1) + c] = v * inv_t;
        }
    }
}

// GEMM form (ws given): the S negatives are shared by all rows, so their scores are ONE dense contraction
// X[N,D] @ W_neg[S,D]^T on the matrix cores (W_neg gathered into ws) instead of N*S row dot products re-reading the
// S rows of W for every row (C4: 27 k rows x 100 negatives x 1 KB = 2.8 GB through L2, 545 us); the fix-up kernel
// adds the positive column (one row of W per label), the log-q corrections, the accidental-hit constant and 1/T.
__global__ __launch_bounds__(256) void sampled_fix_kernel(const float* __restrict__ x, const long* __restrict__ y,
                                                           const float* __restrict__ W, const long* __restrict__ neg,
                                                           const float* __restrict__ qdist, float* __restrict__ out, int N,
                                                           int D, int S, float inv_t) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const long yi = y[row];
    const float* xr = x + (long)row * D;
    const float* wr = W + yi * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += xr[d] * wr[d];
    s = wave_sum(s);
    float* orow = out + (long)row * (S + 1);
    if (lane == 0) orow[0] = (s - __logf(qdist[yi] + 1e-16f)) * inv_t;
    for (int c = lane; c < S; c += 64) {
        const long id = neg[c];
        float v = orow[1 + c] - __logf(qdist[id] + 1e-16f);
        if (id == yi) v = -65504.0f / 100.0f;
        orow[1 + c] = v * inv_t;
    }
}
__global__ __launch_bounds__(256) void sampled_rows_kernel(const float* __restrict__ W, const long* __restrict__ ids,
                                                            float* __restrict__ out, float* __restrict__ dW,
                                                            const float* __restrict__ add, int n, int D);
int t4r_gemm_launch(hipStream_t stream, int transA, int transB, int M, int N, int K, float alpha,
                    const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                    const float* bias, int epilogue, float* aux, long ldaux, int splitk,
                    int accumulate, int batch, long sA, long sB, long sC, const DropCfg* drop);

// ws: n_neg * D floats of scratch (NULL: the row-wise kernel)
extern "C" int t4r_sampled_logits_fwd(void* stream, const float* x, const long* labels, const float* W,
                                      const long* neg_samples, const float* sampling_dist, float* out,
                                      int N, int D, int n_neg, float temperature, float* ws) {
    if (N == 0) return 0;
    const float inv_t = temperature != 0.f ? 1.f / temperature : 1.f;
    hipStream_t st = (hipStream_t)stream;
    if (ws && n_neg >= 4) {
        const long sd = (long)n_neg * D;
        hipLaunchKernelGGL(sampled_rows_kernel, dim3((unsigned)((sd + 255) / 256)), dim3(256), 0, st, W, neg_samples, ws,
                           nullptr, nullptr, n_neg, D);
        T4R_LAUNCH_CHECK();
        const int rc = t4r_gemm_launch(st, 0, 1, N, n_neg, D, 1.f, x, D, ws, D, out + 1, n_neg + 1, nullptr, 0, nullptr,
                                       0, 1, 0, 1, 0, 0, 0, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL(sampled_fix_kernel, dim3((N + 3) / 4), dim3(256), 0, st, x, labels, W, neg_samples,
                           sampling_dist, out, N, D, n_neg, inv_t);
        T4R_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(sampled_logits_kernel, dim3((N + 3) / 4), dim3(256), 0, st, x,
                       labels, W, neg_samples, sampling_dist, out, N, D, n_neg, inv_t);
    T4R_LAUNCH_CHECK();
    return 0;
}

// Negative sampling of the sampled-softmax head: n draws (with replacement) from the log-uniform distribution
// of LogUniformSampler (prediction_task.py:766-786): P(min_id + j) = (ln(j + 2) - ln(j + 1)) / ln(R), j = 0 .. R - 2,
// R = max_id - min_id + 1.  The reference materialises that distribution and calls torch.multinomial (a
// renormalisation + prefix sum over all V categories per call: 0.5 ms per step at 1 M items); the CDF is
// ln(j + 2) / ln R, so the draw is its inverse in closed form: j = floor(R^u) - 1, u ~ U[0, 1) from Philox.
__global__ void log_uniform_sample_kernel(long* __restrict__ out, int n, long min_id, long R, unsigned long long seed,
                                          unsigned long long ctr_hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Philox rng(seed);
    const uint4 r = rng((unsigned long long)i, ctr_hi);
    const double u = ((double)r.x * 4294967296.0 + (double)r.y) * (1.0 / 18446744073709551616.0);   // 64 random bits
    long j = (long)floor(exp(u * log((double)R))) - 1;
    j = j < 0 ? 0 : (j > R - 2 ? R - 2 : j);
    out[i] = min_id + j;
}
extern "C" int t4r_log_uniform_sample(void* stream, long* out, int n, long min_id, long max_id,
                                      unsigned long long seed, unsigned long long ctr_hi) {
    if (n <= 0) return 0;
    T4R_CHECK_ARG(out && max_id - min_id + 1 >= 2, "log_uniform_sample: need at least two ids");
    hipLaunchKernelGGL(log_uniform_sample_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, n,
                       min_id, max_id - min_id + 1, seed, ctr_hi);
    T4R_LAUNCH_CHECK();
    return 0;
}

// backward of sampled logits:  dx[row] = sum_c g[row,c]/T * W[id_c] ; dW[id_c] += g[row,c]/T * x[row]
// (accidental hits carry no gradient: their value is a constant).
// The S negatives are SHARED by all rows, so their part is two small dense contractions on the
// matrix cores instead of N*S*D atomics onto S*D addresses (3.2 ms at N = 25k, S = 100, D = 256):
//   d x      = (1/T) G_neg[N,S] @ W_neg[S,D]          W_neg = gathered rows of W
//   d W_neg  = (1/T) G_neg^T[S,N] @ x[N,D]            then added to the S rows of dW
// after the accidental-hit entries of G were zeroed in place; the positive column (one row of W per
// label) stays a row-wise kernel with atomics on dW[y] (labels rarely collide).
int t4r_gemm_launch(hipStream_t stream, int transA, int transB, int M, int N, int K, float alpha,
                    const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                    const float* bias, int epilogue, float* aux, long ldaux, int splitk,
                    int accumulate, int batch, long sA, long sB, long sC, const DropCfg* drop);

__global__ __launch_bounds__(256) void sampled_mask_hits_kernel(float* __restrict__ g, const long* __restrict__ y,
                                                                 const long* __restrict__ neg, int N, int S) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * S) return;
    const int row = (int)(i / S), s = (int)(i % S);
    if (neg[s] == y[row]) g[(long)row * (S + 1) + 1 + s] = 0.f;
}
__global__ __launch_bounds__(256) void sampled_rows_kernel(const float* __restrict__ W, const long* __restrict__ ids,
                                                            float* __restrict__ out, float* __restrict__ dW,
                                                            const float* __restrict__ add, int n, int D) {
    // gather (add == null): out[i,:] = W[ids[i],:]   |   scatter-add: dW[ids[i],:] += add[i,:]  (ids unique)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * D) return;
    const long r = i / D, c = i % D;
    if (add) dW[ids[r] * D + c] += add[i]; else out[i] = W[ids[r] * D + c];
}
// positive column: dx[row,:] += g0/T * W[y,:] ; dW[y,:] += g0/T * x[row,:]   (one wave per row)
__global__ __launch_bounds__(256) void sampled_pos_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                               const long* __restrict__ y, const float* __restrict__ W,
                                                               float* __restrict__ dx, float* __restrict__ dW,
                                                               float* __restrict__ rows_out, int N, int D, int S,
                                                               float inv_t) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const long yi = y[row];
    const float gv = g[(long)row * (S + 1)] * inv_t;
    for (int d = lane; d < D; d += 64) {
        dx[(long)row * D + d] += gv * W[yi * D + d];
        if (rows_out) rows_out[(long)row * D + d] = gv * x[(long)row * D + d];     // row-sparse form: d W[y_row] contribution
        else atomicAdd(dW + yi * D + d, gv * x[(long)row * D + d]);
    }
}

// dlogits is MODIFIED in place (accidental-hit entries zeroed).  ws: 2 * n_neg * D floats of scratch.
// rows_out == null: the weight gradient is accumulated into the dense dW[V, D] (atomics on the label rows).
// rows_out != null (dW ignored): ROW-SPARSE form -- rows_out[(N + n_neg), D] receives the gradient rows of
// the ids (labels[0..N) ++ neg_samples[0..n_neg)); the caller sums them into the table with the
// deterministic sorted scatter (t4r_embedding_bwd_sorted) or exchanges them between data-parallel ranks
// (a 1 M x 256 table gradient is 1 GB dense, ~50 MB as rows).
static int sampled_logits_bwd_impl(void* stream, float* dlogits, const float* x, const long* labels,
                                   const float* W, const long* neg_samples, float* dx, float* dW,
                                   float* rows_out, float* ws, int N, int D, int n_neg, float temperature) {
    if (N == 0) return 0;
    T4R_CHECK_ARG(ws != nullptr, "sampled_logits_bwd: workspace (2 * n_neg * D floats) required");
    hipStream_t st = (hipStream_t)stream;
    const int S = n_neg;
    const float inv_t = temperature != 0.f ? 1.f / temperature : 1.f;
    float* w_neg = ws;
    float* dw_neg = rows_out ? rows_out + (long)N * D : ws + (long)S * D;
    const long ns = (long)N * S, sd = (long)S * D;
    hipLaunchKernelGGL(sampled_mask_hits_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, st, dlogits,
                       labels, neg_samples, N, S);
    hipLaunchKernelGGL(sampled_rows_kernel, dim3((unsigned)((sd + 255) / 256)), dim3(256), 0, st, W, neg_samples,
                       w_neg, nullptr, nullptr, S, D);
    T4R_LAUNCH_CHECK();
    // d x = (1/T) G_neg @ W_neg
    int rc = t4r_gemm_launch(st, 0, 0, N, D, S, inv_t, dlogits + 1, S + 1, w_neg, D, dx, D, nullptr, 0, nullptr, 0,
                             1, 0, 1, 0, 0, 0, nullptr);
    if (rc) return rc;
    // d W_neg = (1/T) G_neg^T @ x   (split-K over the rows)
    rc = t4r_gemm_launch(st, 1, 0, S, D, N, inv_t, dlogits + 1, S + 1, x, D, dw_neg, D, nullptr, 0, nullptr, 0, -1,
                         0, 1, 0, 0, 0, nullptr);
    if (rc) return rc;
    if (!rows_out)
        hipLaunchKernelGGL(sampled_rows_kernel, dim3((unsigned)((sd + 255) / 256)), dim3(256), 0, st, W, neg_samples,
                           nullptr, dW, dw_neg, S, D);
    hipLaunchKernelGGL(sampled_pos_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, st, dlogits, x, labels, W, dx, dW,
                       rows_out, N, D, S, inv_t);
    T4R_LAUNCH_CHECK();
    return 0;
}

extern "C" int t4r_sampled_logits_bwd(void* stream, float* dlogits, const float* x, const long* labels,
                                      const float* W, const long* neg_samples, float* dx, float* dW,
                                      float* ws, int N, int D, int n_neg, float temperature) {
    T4R_CHECK_ARG(dW != nullptr, "sampled_logits_bwd: dW required");
    return sampled_logits_bwd_impl(stream, dlogits, x, labels, W, neg_samples, dx, dW, nullptr, ws, N, D, n_neg,
                                   temperature);
}

extern "C" int t4r_sampled_logits_bwd_rows(void* stream, float* dlogits, const float* x, const long* labels,
                                           const float* W, const long* neg_samples, float* dx, float* rows_out,
                                           float* ws, int N, int D, int n_neg, float temperature) {
    T4R_CHECK_ARG(rows_out != nullptr, "sampled_logits_bwd_rows: rows_out required");
    return sampled_logits_bwd_impl(stream, dlogits, x, labels, W, neg_samples, dx, nullptr, rows_out, ws, N, D,
                                   n_neg, temperature);
}

// ------------------------------------------------------------------------------------------
// top-k per row (k <= 64) for inference (prediction_task.py:466-470) and Recall/NDCG@k:
// Exact top-k of every row (value descending, ties to the lower index), one workgroup per row:
//   A. every thread takes the maximum of its strided slice; t0 = the k-th largest of the 256 slice
//      maxima is a lower bound of the row's k-th largest value (k elements >= t0 exist);
//   B. second pass: the few elements >= t0 (typically k .. 3k) are appended to an LDS candidate list;
//   C. each candidate counts the candidates that beat it: rank < k -> output slot `rank`.
// Both passes are plain 16-byte streaming reads; no per-element LDS traffic (the former per-thread
// sorted lists in LDS diverged on every insertion: 2.1 ms at 1024 x 100001, this form ~0.3 ms).
// If the candidate list overflows (adversarial rows with > TOPK_CAP values >= t0, e.g. constant
// rows) the workgroup falls back to the sorted-list algorithm below.
#define TOPK_MAX 256         // the threshold of step A is the k-th largest of 256 slice maxima
#define TOPK_LISTS_MAX 64    // the sorted-list fallback keeps [256][k] values + indices in LDS (128 KB at k = 64)
#define TOPK_CAP 2048
__device__ void topk_lists_fallback(const float* x, int V, int k, float* sh, float* out_val, long* out_idx, long row);
__device__ void topk_select_fallback(const float* x, int V, int k, float* out_val, long* out_idx, long row);

__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ scores, int V, long ld,
                                                    int k, float* __restrict__ out_val,
                                                    long* __restrict__ out_idx) {
    extern __shared__ float sh[];            // fallback: [256][k] values + [256][k] indices
    __shared__ float cand_v[TOPK_CAP];
    __shared__ int cand_i[TOPK_CAP];
    __shared__ float tmax[256];
    __shared__ float t0s;
    __shared__ int cnt;
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = scores + (long)row * ld;
    const bool vec = (ld % 4 == 0) && ((uintptr_t)scores % 16 == 0);
    const int v4 = vec ? V / 4 : 0;
    // A: slice maxima
    float m = -INFINITY;
    for (int q = tid; q < v4; q += 256) {
        const float4 t = *reinterpret_cast<const float4*>(x + 4 * q);
        m = fmaxf(m, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
    }
    for (int i = v4 * 4 + tid; i < V; i += 256) m = fmaxf(m, x[i]);
    tmax[tid] = m;
    if (tid == 0) cnt = 0;
    __syncthreads();
    {
        int rank = 0;
        for (int o = 0; o < 256; ++o) {
            const float v = tmax[o];
            rank += (v > m || (v == m && o < tid)) ? 1 : 0;
        }
        if (rank == k - 1) t0s = m;          // k <= 256 ranks 0 .. 255: exactly one thread has this rank
    }
    __syncthreads();
    const float t0 = t0s;
    // B: candidates
    auto offer = [&](float v, int i) {
        if (v >= t0) {
            const int slot = atomicAdd(&cnt, 1);
            if (slot < TOPK_CAP) { cand_v[slot] = v; cand_i[slot] = i; }
        }
    };
    for (int q = tid; q < v4; q += 256) {
        const float4 t = *reinterpret_cast<const float4*>(x + 4 * q);
        offer(t.x, 4 * q); offer(t.y, 4 * q + 1); offer(t.z, 4 * q + 2); offer(t.w, 4 * q + 3);
    }
    for (int i = v4 * 4 + tid; i < V; i += 256) offer(x[i], i);
    __syncthreads();
    const int C = cnt;
    if (C > TOPK_CAP) {                      // workgroup-uniform
        if (k <= TOPK_LISTS_MAX) topk_lists_fallback(x, V, k, sh, out_val, out_idx, row);
        else topk_select_fallback(x, V, k, out_val, out_idx, row);
        return;
    }
    // C: rank among candidates
    for (int c = tid; c < C; c += 256) {
        const float v = cand_v[c];
        const int i = cand_i[c];
        int rank = 0;
        for (int o = 0; o < C; ++o) {
            const float v2 = cand_v[o];
            rank += (v2 > v || (v2 == v && cand_i[o] < i)) ? 1 : 0;
        }
        if (rank < k) {
            out_val[(long)row * k + rank] = v;
            out_idx[(long)row * k + rank] = i;
        }
    }
}

// sorted per-thread lists in LDS + tournament merge (fallback path; ties resolve to the lower index)
__device__ void topk_lists_fallback(const float* x, int V, int k, float* sh, float* out_val, long* out_idx, long row) {
    float* cv = sh;                         // [256][k]
    int* ci = (int*)(sh + 256 * k);         // [256][k]
    float* myv = cv + threadIdx.x * k;
    int* myi = ci + threadIdx.x * k;
    for (int j = 0; j < k; ++j) { myv[j] = -INFINITY; myi[j] = 0x7fffffff; }
    float thr = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) {
        const float v = x[i];
        if (v > thr) {
            int j = k - 1;
            while (j > 0 && (myv[j - 1] < v)) {
                myv[j] = myv[j - 1]; myi[j] = myi[j - 1]; --j;
            }
            myv[j] = v; myi[j] = i;
            thr = myv[k - 1];
        }
    }
    __syncthreads();
    __shared__ int head[256];
    __shared__ float bv[4];
    __shared__ int bi[4], bt[4];
    head[threadIdx.x] = 0;
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        const int hd = head[threadIdx.x];
        float v = hd < k ? myv[hd] : -INFINITY;
        int idx = hd < k ? myi[hd] : 0x7fffffff;
        int t = threadIdx.x;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(v, o, 64);
            const int i2 = __shfl_xor(idx, o, 64);
            const int t2 = __shfl_xor(t, o, 64);
            if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; t = t2; }
        }
        if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = v; bi[threadIdx.x >> 6] = idx; bt[threadIdx.x >> 6] = t; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (bv[w] > v || (bv[w] == v && bi[w] < idx)) { v = bv[w]; idx = bi[w]; t = bt[w]; }
            out_val[row * k + r] = v;
            out_idx[row * k + r] = idx;
            head[t] += 1;
        }
        __syncthreads();
    }
}

// fallback of the fallback for 64 < k <= 256 (no LDS lists of that size): k rounds of a workgroup-wide arg-max over the elements
// that come after the previous pick in (value descending, index ascending) order.  k V reads per row: adversarial rows only
// (more than TOPK_CAP values at or above the k-th largest slice maximum, e.g. constant rows).
__device__ void topk_select_fallback(const float* x, int V, int k, float* out_val, long* out_idx, long row) {
    __shared__ float wv[4];
    __shared__ int wi[4];
    __shared__ float pv_s;
    __shared__ int pi_s;
    float prev_v = INFINITY;
    int prev_i = -1;
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < V; i += 256) {
            const float v = x[i];
            const bool after = v < prev_v || (v == prev_v && i > prev_i);
            if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
        }
        if ((threadIdx.x & 63) == 0) { wv[threadIdx.x >> 6] = bv; wi[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (wv[w] > bv || (wv[w] == bv && wi[w] < bi)) { bv = wv[w]; bi = wi[w]; }
            out_val[row * k + r] = bv;
            out_idx[row * k + r] = bi;
            pv_s = bv; pi_s = bi;
        }
        __syncthreads();
        prev_v = pv_s; prev_i = pi_s;
        __syncthreads();
    }
}

extern "C" int t4r_topk(void* stream, const float* scores, int N, int V, long ld, int k, float* out_val,
                        long* out_idx) {
    if (N == 0) return 0;
    T4R_CHECK_ARG(k >= 1 && k <= TOPK_MAX && k <= V, "topk: 1 <= k <= min(256, V)");
    const size_t smem = k <= TOPK_LISTS_MAX ? (size_t)256 * k * 8 : 0;
    static T4rLdsAttr attr;
    t4r_ensure_dynamic_lds((const void*)topk_kernel, smem, attr);
    hipLaunchKernelGGL(topk_kernel, dim3(N), dim3(256), smem, (hipStream_t)stream, scores, V, ld, k,
                       out_val, out_idx);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Non-materialising head: linear (tied / untied output projection) + softmax cross-entropy without
// an [N, V] logits tensor.  Replaces prediction_task.py:664-669 (X @ W^T, / T) + :446 (CrossEntropyLoss)
// and their autograd when `predictions` are not asked for -- the only form that can run C5
// (15 k label rows x 10 M items = 600 GB of logits).
//
// The vocabulary is streamed in chunks of `chunk_cols` columns through ONE [N, chunk] buffer sized to
// stay in the 256 MB Infinity Cache: the chunk's logits are produced by the same fp32 MFMA GEMM as the
// materialised head (bit-identical values), consumed by an online (max, sum-exp) update, and dropped.
// The backward recomputes each chunk (one extra GEMM: 4 instead of 3 vocabulary-wide products) and feeds
// it to the two gradient contractions whose A operand forms the softmax gradient on the fly
// (t4r_gemm_softmax_grad_launch): neither the logits nor their gradient ever exist at [N, V].
//   stats: m[N] | s[N] | tot[N] | tgt[N]   (running max, sum exp(x - m), sum x, logit of the label)
int t4r_gemm_launch(hipStream_t stream, int transA, int transB, int M, int N, int K, float alpha,
                    const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                    const float* bias, int epilogue, float* aux, long ldaux, int splitk,
                    int accumulate, int batch, long sA, long sB, long sC, const DropCfg* drop);
int t4r_gemm_softmax_grad_launch(hipStream_t stream, int transA, int n_rows, int Vc, int V, int yoff, int N,
                                 float alpha, const float* logits, long ld_logits, const float* lse,
                                 const long* labels, const float* grad_out, float label_smoothing, const float* B,
                                 long ldb, float* C, long ldc, int splitk, int accumulate);

// one workgroup per row: online softmax statistics of the chunk's Vc columns merged into the running ones
__global__ __launch_bounds__(256) void ce_chunk_stats_kernel(const float* __restrict__ chunk, long ld, int Vc,
                                                              int v0, const long* __restrict__ labels,
                                                              float* __restrict__ stats, int N, int first) {
    const int row = blockIdx.x;
    const float* x = chunk + (long)row * ld;
    float m = -INFINITY, s = 0.f, tot = 0.f;
    const int v4 = Vc / 4;                      // ld % 4 == 0 and the buffer is 16-byte aligned (host checks)
    int i = threadIdx.x;
    for (; i + 768 < v4; i += 1024) {
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(x + 4 * (i + 256 * u));
        float mx = m;
#pragma unroll
        for (int u = 0; u < 4; ++u) mx = fmaxf(mx, fmaxf(fmaxf(t[u].x, t[u].y), fmaxf(t[u].z, t[u].w)));
        float e = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            e += (__expf(t[u].x - mx) + __expf(t[u].y - mx)) + (__expf(t[u].z - mx) + __expf(t[u].w - mx));
            tot += (t[u].x + t[u].y) + (t[u].z + t[u].w);
        }
        s = s * __expf(m - mx) + e;
        m = mx;
    }
    for (; i < v4; i += 256) {
        const float4 t = *reinterpret_cast<const float4*>(x + 4 * i);
        const float mx = fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w));
        const float mn = fmaxf(m, mx);
        s = s * __expf(m - mn) + __expf(t.x - mn) + __expf(t.y - mn) + __expf(t.z - mn) + __expf(t.w - mn);
        m = mn;
        tot += t.x + t.y + t.z + t.w;
    }
    for (int k = v4 * 4 + threadIdx.x; k < Vc; k += 256) {
        const float t = x[k];
        const float mn = fmaxf(m, t);
        s = s * __expf(m - mn) + __expf(t - mn);
        m = mn;
        tot += t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
        tot += __shfl_xor(tot, o, 64);
    }
    __shared__ float sm[4], ss[4], st[4];
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; ss[threadIdx.x >> 6] = s; st[threadIdx.x >> 6] = tot; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { online_merge(m, s, sm[w], ss[w]); tot += st[w]; }
        float* pm = stats + row; float* ps = stats + N + row; float* pt = stats + 2L * N + row; float* pg = stats + 3L * N + row;
        if (!first) { online_merge(m, s, *pm, *ps); tot += *pt; }
        *pm = m; *ps = s; *pt = tot;
        const long y = labels[row] - v0;
        if (y >= 0 && y < Vc) *pg = x[y];
    }
}

__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ stats, float* __restrict__ loss_row,
                                                           float* __restrict__ lse_out, int N, int V, float smoothing) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const float lse = stats[row] + __logf(stats[N + row]);
    float loss = lse - stats[3L * N + row];
    if (smoothing > 0.f) loss = (1.f - smoothing) * loss + smoothing * (lse - stats[2L * N + row] / V);
    loss_row[row] = loss;
    lse_out[row] = lse;
}

static long chunk_ld(int c) { return ((long)c + 63) / 64 * 64; }     // rows on 256-byte boundaries (as ops.pad_ld)

// floats of the [N, chunk] logits buffer both passes stream the vocabulary through
extern "C" long t4r_linear_softmax_ce_chunk_floats(int N, int chunk_cols) {
    return (long)N * chunk_ld(chunk_cols);
}

// loss_rows[N], lse[N], *loss_mean = mean CE of softmax(alpha * X @ W^T) against labels.
// chunk_buf: t4r_linear_softmax_ce_chunk_floats(N, chunk_cols) floats, 16-byte aligned; stats: 4*N floats.
// Every label must be in [0, V).
extern "C" int t4r_linear_softmax_ce_fwd(void* stream, const float* X, long ldx, const float* W, long ldw,
                                         const long* labels, int N, int V, int D, float alpha,
                                         float label_smoothing, int chunk_cols, float* chunk_buf, float* stats,
                                         float* loss_rows, float* lse, float* loss_mean) {
    if (N <= 0) return 0;
    T4R_CHECK_ARG(X && W && labels && chunk_buf && stats && loss_rows && lse, "linear_softmax_ce_fwd: null pointer");
    T4R_CHECK_ARG(chunk_cols >= 4 && chunk_cols % 4 == 0, "linear_softmax_ce_fwd: chunk_cols must be a multiple of 4");
    T4R_CHECK_ARG((uintptr_t)chunk_buf % 16 == 0, "linear_softmax_ce_fwd: chunk buffer must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const long ld = chunk_ld(chunk_cols);
    for (int v0 = 0; v0 < V; v0 += chunk_cols) {
        const int vc = V - v0 < chunk_cols ? V - v0 : chunk_cols;
        const int rc = t4r_gemm_launch(st, 0, 1, N, vc, D, alpha, X, ldx, W + (long)v0 * ldw, ldw, chunk_buf, ld,
                                       nullptr, 0, nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr);
        if (rc != 0) return rc;
        hipLaunchKernelGGL(ce_chunk_stats_kernel, dim3(N), dim3(256), 0, st, chunk_buf, ld, vc, v0, labels, stats, N,
                           v0 == 0 ? 1 : 0);
    }
    hipLaunchKernelGGL(ce_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, st, stats, loss_rows, lse, N, V,
                       label_smoothing);
    if (loss_mean) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, st, loss_rows, N, loss_mean);
    T4R_LAUNCH_CHECK();
    return 0;
}

// backward of the above for a scalar upstream gradient *grad_out (null: 1):
//   dX[N, D]  = alpha * dlogits @ W          (overwritten)
//   dW[V, D] += alpha * dlogits^T @ X        (accumulated; null: skipped)
// with dlogits = (*grad_out / N) * (softmax - (1-eps) onehot - eps/V) recomputed chunk by chunk from `lse`.
extern "C" int t4r_linear_softmax_ce_bwd(void* stream, const float* X, long ldx, const float* W, long ldw,
                                         const long* labels, const float* lse, const float* grad_out, int N, int V,
                                         int D, float alpha, float label_smoothing, int chunk_cols,
                                         float* chunk_buf, float* dX, long lddx, float* dW, long lddw) {
    if (N <= 0) return 0;
    T4R_CHECK_ARG(X && W && labels && lse && chunk_buf && dX, "linear_softmax_ce_bwd: null pointer");
    T4R_CHECK_ARG(chunk_cols >= 4 && chunk_cols % 4 == 0, "linear_softmax_ce_bwd: chunk_cols must be a multiple of 4");
    hipStream_t st = (hipStream_t)stream;
    const long ld = chunk_ld(chunk_cols);
    for (int v0 = 0; v0 < V; v0 += chunk_cols) {
        const int vc = V - v0 < chunk_cols ? V - v0 : chunk_cols;
        int rc = t4r_gemm_launch(st, 0, 1, N, vc, D, alpha, X, ldx, W + (long)v0 * ldw, ldw, chunk_buf, ld,
                                 nullptr, 0, nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr);
        if (rc != 0) return rc;
        // d X (+)= dlogits_chunk @ W[v0 : v0 + vc]      (split-K over the chunk's columns)
        rc = t4r_gemm_softmax_grad_launch(st, 0, N, vc, V, v0, D, alpha, chunk_buf, ld, lse, labels, grad_out,
                                          label_smoothing, W + (long)v0 * ldw, ldw, dX, lddx, -1, v0 == 0 ? 0 : 1);
        if (rc != 0) return rc;
        if (dW) {   // d W[v0 : v0 + vc] += dlogits_chunk^T @ X
            rc = t4r_gemm_softmax_grad_launch(st, 1, N, vc, V, v0, D, alpha, chunk_buf, ld, lse, labels, grad_out,
                                              label_smoothing, X, ldx, dW + (long)v0 * lddw, lddw, 1, 1);
            if (rc != 0) return rc;
        }
    }
    return 0;
}
