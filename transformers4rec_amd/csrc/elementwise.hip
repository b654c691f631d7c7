// Row-wise / element-wise HBM-bound kernels of the hot path (gfx950, wave = 64).
//   residual + LayerNorm fwd/bwd  : XLNet post-LN  (HF modeling_xlnet.py:142-152, 297-305; eps 0.03)
//                                   TabularLayerNorm (transformers4rec/torch/tabular/transformations.py:128-132)
//   GELU(erf) backward            : XLNetFeedForward activation (HF modeling_xlnet.py:300)
//   ReLU backward                 : projection MLP (transformers4rec/torch/block/mlp.py:133-135)
//   column sums                   : bias gradients
//   fused Adam                    : torch.optim.Adam semantics (reference Model.fit, torch/model/base.py:669-718)
// All are one pass over their operands with 16-byte accesses where the row width allows.
#include "t4r_common.h"

// ---------------------------------------------------------------- residual + LayerNorm fwd
template <int VEC>
struct alignas(4 * VEC) FV {
    float v[VEC];
};

// y = LN(a + b) * gamma + beta ; one wave per row.  Lane l owns columns (c*64 + l)*VEC.. for
// c < NC (statically unrolled so the row stays in registers).  b may be null.
// Saves mean/rstd per row for the backward.
template <int VEC, int NC>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* ar = a + (long)row * D;
    const float* br = b ? b + (long)row * D : nullptr;
    FV<VEC> v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c0 = (c * 64 + lane) * VEC;
        if (c0 < D) {
            v[c] = *reinterpret_cast<const FV<VEC>*>(ar + c0);
            if (br) {
                const FV<VEC> t = *reinterpret_cast<const FV<VEC>*>(br + c0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[c].v[e] += t.v[e];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += v[c].v[e];
        }
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c0 = (c * 64 + lane) * VEC;
        if (c0 < D) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float d = v[c].v[e] - mu;
                q += d * d;
            }
        }
    }
    const float var = wave_sum(q) / D;
    const float rs = rsqrtf(var + eps);
    float* yr = y + (long)row * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c0 = (c * 64 + lane) * VEC;
        if (c0 < D) {
            const FV<VEC> g = *reinterpret_cast<const FV<VEC>*>(gamma + c0);
            const FV<VEC> bb = *reinterpret_cast<const FV<VEC>*>(beta + c0);
            FV<VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.v[e] = (v[c].v[e] - mu) * rs * g.v[e] + bb.v[e];
            *reinterpret_cast<FV<VEC>*>(yr + c0) = o;
        }
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

static bool ln_pick(int D, int* vec, int* nc) {
    int v = (D % 256 == 0) ? 4 : (D % 128 == 0) ? 2 : 1;
    int chunks = (D + 64 * v - 1) / (64 * v);
    if (chunks > 8 && D % 4 == 0) { v = 4; chunks = (D + 255) / 256; }
    if (chunks > 8) return false;
    int n = 1;
    while (n < chunks) n <<= 1;
    *vec = v; *nc = n;
    return true;
}

#define LN_DISPATCH(KERNEL, ...)                                                            \
    do {                                                                                    \
        if (vec == 4 && nc == 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, st, __VA_ARGS__); \
        else if (vec == 4 && nc == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, st, __VA_ARGS__); \
        else if (vec == 4 && nc == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, 0, st, __VA_ARGS__); \
        else if (vec == 4 && nc == 8) hipLaunchKernelGGL((KERNEL<4, 8>), grid, block, 0, st, __VA_ARGS__); \
        else if (vec == 2 && nc == 1) hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, st, __VA_ARGS__); \
        else if (vec == 2) hipLaunchKernelGGL((KERNEL<2, 8>), grid, block, 0, st, __VA_ARGS__); \
        else if (nc == 1) hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, st, __VA_ARGS__); \
        else if (nc == 2) hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<1, 8>), grid, block, 0, st, __VA_ARGS__);           \
    } while (0)

extern "C" int t4r_add_layernorm_fwd(void* stream, const float* a, const float* b,
                                     const float* gamma, const float* beta, float* y, float* mean,
                                     float* rstd, int rows, int D, float eps) {
    if (rows <= 0) return 0;
    int vec, nc;
    T4R_CHECK_ARG(D > 0 && ln_pick(D, &vec, &nc), "layernorm: D out of range");
    hipStream_t st = (hipStream_t)stream;
    const int wpb = 4;
    dim3 grid((rows + wpb - 1) / wpb), block(64 * wpb);
    LN_DISPATCH(add_layernorm_fwd_kernel, a, b, gamma, beta, y, mean, rstd, rows, D, eps);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- residual + LayerNorm bwd
// x = a + b (recomputed), xhat = (x - mean) * rstd
// dx = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat)),  g = dy * gamma
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy     (atomics, one per column per wave)
// Each wave walks rows_per_wave rows keeping its column partials in registers.
template <int VEC, int NC>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dy,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int D,
    int rows_per_wave, int accumulate_dx) {
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    FV<VEC> pg[NC], pb[NC], gam[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c0 = (c * 64 + lane) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) pg[c].v[e] = pb[c].v[e] = gam[c].v[e] = 0.f;
        if (c0 < D) gam[c] = *reinterpret_cast<const FV<VEC>*>(gamma + c0);
    }
    const int r0 = wave_global * rows_per_wave;
    const int r1 = min(rows, r0 + rows_per_wave);
    for (int row = r0; row < r1; ++row) {
        const float* ar = a + (long)row * D;
        const float* br = b ? b + (long)row * D : nullptr;
        const float* dyr = dy + (long)row * D;
        const float mu = mean[row], rs = rstd[row];
        FV<VEC> xh[NC], g[NC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c0 = (c * 64 + lane) * VEC;
            if (c0 < D) {
                FV<VEC> x = *reinterpret_cast<const FV<VEC>*>(ar + c0);
                if (br) {
                    const FV<VEC> t = *reinterpret_cast<const FV<VEC>*>(br + c0);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) x.v[e] += t.v[e];
                }
                const FV<VEC> d = *reinterpret_cast<const FV<VEC>*>(dyr + c0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float h = (x.v[e] - mu) * rs;
                    const float gg = d.v[e] * gam[c].v[e];
                    xh[c].v[e] = h;
                    g[c].v[e] = gg;
                    s1 += gg;
                    s2 += gg * h;
                    pg[c].v[e] += d.v[e] * h;
                    pb[c].v[e] += d.v[e];
                }
            }
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
        float* dxr = dx + (long)row * D;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c0 = (c * 64 + lane) * VEC;
            if (c0 < D) {
                FV<VEC> o;
                if (accumulate_dx) o = *reinterpret_cast<const FV<VEC>*>(dxr + c0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float v = rs * (g[c].v[e] - s1 - xh[c].v[e] * s2);
                    o.v[e] = accumulate_dx ? o.v[e] + v : v;
                }
                *reinterpret_cast<FV<VEC>*>(dxr + c0) = o;
            }
        }
    }
    if (r0 >= r1) return;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c0 = (c * 64 + lane) * VEC;
        if (c0 < D) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (dgamma) atomicAdd(dgamma + c0 + e, pg[c].v[e]);
                if (dbeta) atomicAdd(dbeta + c0 + e, pb[c].v[e]);
            }
        }
    }
}

extern "C" int t4r_add_layernorm_bwd(void* stream, const float* a, const float* b,
                                     const float* gamma, const float* mean, const float* rstd,
                                     const float* dy, float* dx, float* dgamma, float* dbeta,
                                     int rows, int D, int accumulate_dx) {
    if (rows <= 0) return 0;
    int vec, nc;
    T4R_CHECK_ARG(D > 0 && ln_pick(D, &vec, &nc), "layernorm: D out of range");
    hipStream_t st = (hipStream_t)stream;
    const int wpb = 4;
    int rpw = (rows + 2048 * wpb - 1) / (2048 * wpb);  // ~2048 blocks at most
    if (rpw < 4) rpw = 4;
    const int waves = (rows + rpw - 1) / rpw;
    dim3 grid((waves + wpb - 1) / wpb), block(64 * wpb);
    LN_DISPATCH(add_layernorm_bwd_kernel, a, b, gamma, mean, rstd, dy, dx, dgamma, dbeta, rows, D,
                rpw, accumulate_dx);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- activation backward + bias grad
// mode 0: GELU(erf):  dpre = dact * gelu'(pre)      (pre = saved pre-activation)
// mode 1: ReLU     :  dpre = dact * (out > 0)       (pre = saved OUTPUT of relu)
// dbias[N] += column sums of dpre.  Block = 256 threads handles a [rows_per_block, N] slab;
// thread t owns columns t, t+256, ... so the column partial stays in a register.
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(
    const float* __restrict__ dact, const float* __restrict__ pre, float* __restrict__ dpre,
    float* __restrict__ dbias, long rows, int N, int rows_per_block, int mode) {
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < N; c += 256) {
        float acc = 0.f;
        for (long r = r0; r < r1; ++r) {
            const long i = r * N + c;
            const float p = pre[i];
            float g = dact[i];
            g *= (mode == 0) ? gelu_erf_grad(p) : (p > 0.f ? 1.f : 0.f);
            dpre[i] = g;
            acc += g;
        }
        if (dbias) atomicAdd(dbias + c, acc);
    }
}

extern "C" int t4r_act_bwd_bias(void* stream, const float* dact, const float* pre, float* dpre,
                                float* dbias, long rows, int N, int mode) {
    if (rows <= 0) return 0;
    const int rpb = 16;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(act_bwd_bias_kernel, grid, dim3(256), 0, (hipStream_t)stream, dact, pre, dpre,
                       dbias, rows, N, rpb, mode);
    T4R_LAUNCH_CHECK();
    return 0;
}

// column sums: out[N] += sum_rows x[rows, N]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x,
                                                      float* __restrict__ out, long rows, int N,
                                                      long ld, int rows_per_block) {
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    for (int c = threadIdx.x; c < N; c += 256) {
        float acc = 0.f;
        for (long r = r0; r < r1; ++r) acc += x[r * ld + c];
        atomicAdd(out + c, acc);
    }
}

extern "C" int t4r_colsum(void* stream, const float* x, float* out, long rows, int N, long ld) {
    if (rows <= 0) return 0;
    const int rpb = 32;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, rows, N, ld, rpb);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- fused Adam over a flat buffer
// torch.optim.Adam (amsgrad=False, maximize=False): with step t (1-based)
//   g = grad (+ wd * p) ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// grad_scale multiplies grad first (1/world_size for the DP mean).  Optionally zeroes the grad.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt,
                                                    float grad_scale, int zero_grad) {
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4*>(p + i);
            float4 gg = *reinterpret_cast<float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float gr = G[e] * grad_scale + wd * P[e];
                M[e] = b1 * M[e] + (1.f - b1) * gr;
                V[e] = b2 * V[e] + (1.f - b2) * gr * gr;
                const float denom = sqrtf(V[e]) / bc2_sqrt + eps;
                P[e] -= (lr / bc1) * (M[e] / denom);
            }
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
            if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (long j = i; j < n; ++j) {
                float gr = g[j] * grad_scale + wd * p[j];
                m[j] = b1 * m[j] + (1.f - b1) * gr;
                v[j] = b2 * v[j] + (1.f - b2) * gr * gr;
                const float denom = sqrtf(v[j]) / bc2_sqrt + eps;
                p[j] -= (lr / bc1) * (m[j] / denom);
                if (zero_grad) g[j] = 0.f;
            }
        }
    }
}

extern "C" int t4r_adam_step(void* stream, float* param, float* grad, float* exp_avg,
                             float* exp_avg_sq, long n, int step, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float grad_scale, int zero_grad) {
    if (n <= 0) return 0;
    T4R_CHECK_ARG(step >= 1, "adam: step is 1-based");
    T4R_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                  "adam: buffers must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param,
                       grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                       grad_scale, zero_grad);
    T4R_LAUNCH_CHECK();
    return 0;
}
