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
    float* __restrict__ rstd, int rows, int D, float eps, DropCfg drop) {
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
            if (drop.p > 0.f) {
                float msk[VEC];
                drop_scale_vec<VEC>(drop, (unsigned long long)row * D + c0, (D & 3) == 0, msk);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[c].v[e] *= msk[e];
            }
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
                                     float* rstd, int rows, int D, float eps, float drop_p,
                                     unsigned long long seed, unsigned long long ctr_hi) {
    if (rows <= 0) return 0;
    int vec, nc;
    T4R_CHECK_ARG(D > 0 && ln_pick(D, &vec, &nc), "layernorm: D out of range");
    T4R_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "layernorm: dropout p in [0, 1)");
    hipStream_t st = (hipStream_t)stream;
    const int wpb = 4;
    dim3 grid((rows + wpb - 1) / wpb), block(64 * wpb);
    const DropCfg drop = make_drop(drop_p, seed, ctr_hi);
    LN_DISPATCH(add_layernorm_fwd_kernel, a, b, gamma, beta, y, mean, rstd, rows, D, eps, drop);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- deterministic column reductions
// Every batch-reduced gradient (LayerNorm gamma/beta, biases, attention biases / k_r) is produced
// in two stages: each workgroup writes its partial sums to a workspace row, then this kernel
// sums the rows.  No atomics: 20k token rows hitting a few hundred addresses serialise in L2
// (measured: 254 us for a LayerNorm backward whose data pass takes ~10 us).
//   out_s[i] (+)= sum_b part[b*n + off_s + i]   for up to three output segments s.
struct ReduceSeg { float* out; int len; int accumulate; };
// Workgroup = 16 output columns x 32 row groups: n / 16 workgroups (the reduced widths are only a
// few hundred columns, so 64-column workgroups left the launch at 2-8 workgroups and ~15 us of
// pure load latency).
__global__ __launch_bounds__(512) void reduce_partials_kernel(const float* __restrict__ part,
                                                               int nblocks, int n, ReduceSeg s0,
                                                               ReduceSeg s1, ReduceSeg s2) {
    __shared__ float sm[32][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + c;
    float acc = 0.f;
    if (i < n) {
        int b = g;
        for (; b + 96 < nblocks; b += 128)
            acc += (part[(long)b * n + i] + part[(long)(b + 32) * n + i]) +
                   (part[(long)(b + 64) * n + i] + part[(long)(b + 96) * n + i]);
        for (; b < nblocks; b += 32) acc += part[(long)b * n + i];
    }
    sm[g][c] = acc;
    __syncthreads();
    if (g == 0 && i < n) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) v += sm[r][c];
        int j = i;
        ReduceSeg seg = s0;
        if (j >= s0.len) { j -= s0.len; seg = s1; if (j >= s1.len) { j -= s1.len; seg = s2; } }
        if (seg.out) seg.out[j] = seg.accumulate ? seg.out[j] + v : v;
    }
}

// Second-stage redirect (installed by the XLNet layer backward for the duration of one call): the
// second stages feed parameter gradients only, so they can leave the critical chain and run on the
// weight-gradient stream.  Each redirected launch is ordered after the first stage by its own event;
// the installer owns the partial buffers (one per site) and joins the streams before they are reused.
static thread_local hipStream_t g_red_side = nullptr;
static thread_local hipEvent_t* g_red_events = nullptr;
static thread_local int g_red_n = 0, g_red_used = 0;
void t4r_reduce_redirect(hipStream_t side, hipEvent_t* events, int n_events) {
    g_red_side = side; g_red_events = events; g_red_n = side ? n_events : 0; g_red_used = 0;
}

bool t4r_splitk_sink_add_reduce(const float* part, int nblocks, int n, float* const* outs, const int* lens, const int* accs,
                                int n_seg);     // gemm_f32.hip
int t4r_reduce_partials_launch(hipStream_t st, const float* part, int nblocks, float* o0, int n0, int a0,
                               float* o1, int n1, int a1, float* o2, int n2, int a2) {
    const int n = n0 + n1 + n2;
    if (n <= 0 || nblocks <= 0) return 0;
    {   // inside a layer backward the sums join the layer's one reduction launch (gemm_f32.hip: split-K sink)
        float* const outs[3] = {o0, o1, o2};
        const int lens[3] = {n0, n1, n2}, accs[3] = {a0, a1, a2};
        if (t4r_splitk_sink_add_reduce(part, nblocks, n, outs, lens, accs, 3)) return 0;
    }
    ReduceSeg s0{o0, n0, a0}, s1{o1, n1, a1}, s2{o2, n2, a2};
    if (g_red_side && g_red_used < g_red_n) {
        hipEvent_t ev = g_red_events[g_red_used++];
        (void)hipEventRecord(ev, st);
        (void)hipStreamWaitEvent(g_red_side, ev, 0);
        st = g_red_side;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 15) / 16), dim3(512), 0, st, part, nblocks, n,
                       s0, s1, s2);
    T4R_LAUNCH_CHECK();
    return 0;
}

#define T4R_COLRED_ROWS 16
// workspace floats needed by the column-reducing kernels below for `rows` rows and `ncols`
// reduced columns in total (LayerNorm backward: 2*D ; act_bwd_bias / colsum: N)
extern "C" long t4r_colreduce_ws_floats(long rows, int ncols) {
    return ((rows + T4R_COLRED_ROWS - 1) / T4R_COLRED_ROWS) * (long)ncols;
}

// ---------------------------------------------------------------- residual + LayerNorm bwd
// x = a + b (recomputed), xhat = (x - mean) * rstd
// dx = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat)),  g = dy * gamma
// dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy   (two-stage, see above)
// One workgroup = 4 waves x 16 rows; column partials stay in registers, are combined across the
// 4 waves through LDS and written to part[block][0:D | D:2D].
template <int VEC, int NC>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dy,
    float* __restrict__ dx, float* __restrict__ dxa, float* __restrict__ part, int rows, int D,
    int accumulate_dx, DropCfg drop) {
    extern __shared__ float sm[];   // [4][2*D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    FV<VEC> pg[NC], pb[NC], gam[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c0 = (c * 64 + lane) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) pg[c].v[e] = pb[c].v[e] = gam[c].v[e] = 0.f;
        if (c0 < D) gam[c] = *reinterpret_cast<const FV<VEC>*>(gamma + c0);
    }
    const int r0 = blockIdx.x * T4R_COLRED_ROWS;
    const int r1 = min(rows, r0 + T4R_COLRED_ROWS);
    if constexpr (NC == 1) {
        // narrow rows (D <= 64 * VEC): the wave's four rows are processed together -- branch-free loads
        // from clamped row indices, so all twelve row segments are in flight before the first reduction
        // (one row at a time the kernel was latency-bound: 26 us for 52 MB at D = 128)
        constexpr int RPW = T4R_COLRED_ROWS / 4;
        const int c0 = lane * VEC;
        const bool cok = c0 < D;
        const int cc = cok ? c0 : 0;
        FV<VEC> xh[RPW], g[RPW], ds[RPW];
        float s1[RPW], s2[RPW], rsv[RPW];
        FV<VEC> xa[RPW], xb[RPW], dd[RPW];
        float muv[RPW];
#pragma unroll
        for (int it = 0; it < RPW; ++it) {
            const int rc = min(r0 + wave + 4 * it, rows - 1);
            xa[it] = *reinterpret_cast<const FV<VEC>*>(a + (long)rc * D + cc);
            if (b) xb[it] = *reinterpret_cast<const FV<VEC>*>(b + (long)rc * D + cc);
            dd[it] = *reinterpret_cast<const FV<VEC>*>(dy + (long)rc * D + cc);
            muv[it] = mean[rc]; rsv[it] = rstd[rc];
        }
#pragma unroll
        for (int it = 0; it < RPW; ++it) {
            const int row = r0 + wave + 4 * it;
            const float live = (row < r1 && cok) ? 1.f : 0.f;
            FV<VEC> x = xa[it];
            if (drop.p > 0.f) {
                drop_scale_vec<VEC>(drop, (unsigned long long)min(row, rows - 1) * D + cc, (D & 3) == 0, ds[it].v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) x.v[e] *= ds[it].v[e];
            }
            if (b) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) x.v[e] += xb[it].v[e];
            }
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float h = (x.v[e] - muv[it]) * rsv[it];
                const float dv = dd[it].v[e] * live;
                const float gg = dv * gam[0].v[e];
                xh[it].v[e] = h;
                g[it].v[e] = gg;
                t1 += gg;
                t2 += gg * h;
                pg[0].v[e] += dv * h;
                pb[0].v[e] += dv;
            }
            s1[it] = t1; s2[it] = t2;
        }
#pragma unroll
        for (int it = 0; it < RPW; ++it) { s1[it] = wave_sum(s1[it]) / D; s2[it] = wave_sum(s2[it]) / D; }
#pragma unroll
        for (int it = 0; it < RPW; ++it) {
            const int row = r0 + wave + 4 * it;
            if (row < r1 && cok) {
                float* dxr = dx + (long)row * D;
                FV<VEC> o;
                if (accumulate_dx) o = *reinterpret_cast<const FV<VEC>*>(dxr + c0);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float v = rsv[it] * (g[it].v[e] - s1[it] - xh[it].v[e] * s2[it]);
                    o.v[e] = accumulate_dx ? o.v[e] + v : v;
                }
                *reinterpret_cast<FV<VEC>*>(dxr + c0) = o;
                if (dxa) {   // gradient of the dropped operand `a`
                    FV<VEC> oa;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float v = rsv[it] * (g[it].v[e] - s1[it] - xh[it].v[e] * s2[it]);
                        oa.v[e] = drop.p > 0.f ? v * ds[it].v[e] : v;
                    }
                    *reinterpret_cast<FV<VEC>*>(dxa + (long)row * D + c0) = oa;
                }
            }
        }
    } else
    for (int row = r0 + wave; row < r1; row += 4) {
        const float* ar = a + (long)row * D;
        const float* br = b ? b + (long)row * D : nullptr;
        const float* dyr = dy + (long)row * D;
        const float mu = mean[row], rs = rstd[row];
        FV<VEC> xh[NC], g[NC], ds[NC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c0 = (c * 64 + lane) * VEC;
            if (c0 < D) {
                FV<VEC> x = *reinterpret_cast<const FV<VEC>*>(ar + c0);
                if (drop.p > 0.f) {
                    drop_scale_vec<VEC>(drop, (unsigned long long)row * D + c0, (D & 3) == 0, ds[c].v);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) x.v[e] *= ds[c].v[e];
                }
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
                if (dxa) {   // gradient of the dropped operand `a`
                    FV<VEC> oa;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float v = rs * (g[c].v[e] - s1 - xh[c].v[e] * s2);
                        oa.v[e] = drop.p > 0.f ? v * ds[c].v[e] : v;
                    }
                    *reinterpret_cast<FV<VEC>*>(dxa + (long)row * D + c0) = oa;
                }
            }
        }
    }
    float* mine = sm + wave * 2 * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int c0 = (c * 64 + lane) * VEC;
        if (c0 < D) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { mine[c0 + e] = pg[c].v[e]; mine[D + c0 + e] = pb[c].v[e]; }
        }
    }
    __syncthreads();
    float* prow = part + (long)blockIdx.x * 2 * D;
    for (int i = threadIdx.x; i < 2 * D; i += 256)
        prow[i] = sm[i] + sm[2 * D + i] + sm[4 * D + i] + sm[6 * D + i];
}

// ws: t4r_colreduce_ws_floats(rows, 2*D) floats.  With dropout (x = drop(a) + b): dx = d loss/d b
// (the residual operand) and dxa = d loss/d a = dx * mask/(1-p); dxa may be NULL when p == 0.
extern "C" int t4r_add_layernorm_bwd(void* stream, const float* a, const float* b,
                                     const float* gamma, const float* mean, const float* rstd,
                                     const float* dy, float* dx, float* dxa, float* dgamma,
                                     float* dbeta, float* ws, int rows, int D, int accumulate_dx,
                                     float drop_p, unsigned long long seed, unsigned long long ctr_hi) {
    if (rows <= 0) return 0;
    int vec, nc;
    T4R_CHECK_ARG(D > 0 && ln_pick(D, &vec, &nc), "layernorm: D out of range");
    T4R_CHECK_ARG(ws != nullptr, "layernorm_bwd: workspace required");
    hipStream_t st = (hipStream_t)stream;
    const int nblocks = (rows + T4R_COLRED_ROWS - 1) / T4R_COLRED_ROWS;
    dim3 grid(nblocks), block(256);
    const size_t smem = (size_t)8 * D * sizeof(float);
    T4R_CHECK_ARG(drop_p == 0.f || dxa, "layernorm_bwd: dxa required with dropout");
    const DropCfg drop = make_drop(drop_p, seed, ctr_hi);
#define LN_BWD(V, N) hipLaunchKernelGGL((add_layernorm_bwd_kernel<V, N>), grid, block, smem, st, a, b, gamma, mean, rstd, dy, dx, dxa, ws, rows, D, accumulate_dx, drop)
    if (vec == 4 && nc == 1) LN_BWD(4, 1);
    else if (vec == 4 && nc == 2) LN_BWD(4, 2);
    else if (vec == 4 && nc == 4) LN_BWD(4, 4);
    else if (vec == 4 && nc == 8) LN_BWD(4, 8);
    else if (vec == 2 && nc == 1) LN_BWD(2, 1);
    else if (vec == 2) LN_BWD(2, 8);
    else if (nc == 1) LN_BWD(1, 1);
    else if (nc == 2) LN_BWD(1, 2);
    else LN_BWD(1, 8);
#undef LN_BWD
    T4R_LAUNCH_CHECK();
    return t4r_reduce_partials_launch(st, ws, nblocks, dgamma, D, 1, dbeta, D, 1, nullptr, 0, 0);
}

// ---------------------------------------------------------------- activation backward + bias grad
// mode 0: GELU(erf):  dpre = dact * gelu'(pre)      (pre = saved pre-activation)
// mode 1: ReLU     :  dpre = dact * (out > 0)       (pre = saved OUTPUT of relu)
// mode 2: plain column sum of dact (no dpre)
// dbias[N] += column sums (two-stage).  A workgroup owns 64 rows; threads are laid out as
// (row group g, float4 column cq) so that every global access is a 16-byte coalesced access.
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(
    const float* __restrict__ dact, const float* __restrict__ pre, float* __restrict__ dpre,
    float* __restrict__ part, long rows, int N, long ld, int mode, int cqp, DropCfg drop) {
    __shared__ float4 sm[256];
    const int NQ = N >> 2;
    const int ng = 256 / cqp;
    const int tc = threadIdx.x % cqp, g = threadIdx.x / cqp;
    const long r0 = (long)blockIdx.x * T4R_COLRED_ROWS;
    const long r1 = min(rows, r0 + T4R_COLRED_ROWS);
    for (int cq0 = 0; cq0 < NQ; cq0 += cqp) {
        const int cq = cq0 + tc;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cq < NQ) {
            for (long r = r0 + g; r < r1; r += ng) {
                const long i = r * ld + 4 * cq;
                float4 d = *reinterpret_cast<const float4*>(dact + i);
                if (drop.p > 0.f) {   // act_out = drop(act(pre)): d act = d out * mask/(1-p)
                    const float4 m = drop_scale4(drop, (unsigned long long)r * N + 4 * cq);
                    d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
                }
                if (mode != 2) {
                    const float4 p = *reinterpret_cast<const float4*>(pre + i);
                    if (mode == 0) {
                        d.x *= gelu_erf_grad(p.x); d.y *= gelu_erf_grad(p.y);
                        d.z *= gelu_erf_grad(p.z); d.w *= gelu_erf_grad(p.w);
                    } else {
                        d.x = p.x > 0.f ? d.x : 0.f; d.y = p.y > 0.f ? d.y : 0.f;
                        d.z = p.z > 0.f ? d.z : 0.f; d.w = p.w > 0.f ? d.w : 0.f;
                    }
                    *reinterpret_cast<float4*>(dpre + i) = d;
                }
                acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
            }
        }
        if (part) {
            sm[threadIdx.x] = acc;
            __syncthreads();
            if (g == 0 && cq < NQ) {
                for (int k = 1; k < ng; ++k) {
                    const float4 o = sm[k * cqp + tc];
                    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
                }
                *reinterpret_cast<float4*>(part + (long)blockIdx.x * N + 4 * cq) = acc;
            }
            __syncthreads();
        }
    }
}

static int colred_launch(hipStream_t st, const float* dact, const float* pre, float* dpre, float* dbias,
                         float* ws, long rows, int N, long ld, int mode, DropCfg drop) {
    if (rows <= 0) return 0;
    T4R_CHECK_ARG(N % 4 == 0 && ld % 4 == 0, "act_bwd_bias/colsum: N and ld must be multiples of 4");
    T4R_CHECK_ARG(!dbias || ws, "act_bwd_bias/colsum: workspace required");
    int cqp = 1;
    while (cqp * 2 <= (N >> 2) && cqp < 256) cqp <<= 1;
    const int nblocks = (int)((rows + T4R_COLRED_ROWS - 1) / T4R_COLRED_ROWS);
    hipLaunchKernelGGL(act_bwd_bias_kernel, dim3(nblocks), dim3(256), 0, st, dact, pre, dpre,
                       dbias ? ws : nullptr, rows, N, ld, mode, cqp, drop);
    T4R_LAUNCH_CHECK();
    if (dbias) return t4r_reduce_partials_launch(st, ws, nblocks, dbias, N, 1, nullptr, 0, 0, nullptr, 0, 0);
    return 0;
}

// ws: t4r_colreduce_ws_floats(rows, N) floats (only needed when dbias != NULL)
extern "C" int t4r_act_bwd_bias(void* stream, const float* dact, const float* pre, float* dpre,
                                float* dbias, float* ws, long rows, int N, int mode, float drop_p,
                                unsigned long long seed, unsigned long long ctr_hi) {
    T4R_CHECK_ARG(mode == 0 || mode == 1, "act_bwd_bias: mode 0 (gelu) or 1 (relu)");
    return colred_launch((hipStream_t)stream, dact, pre, dpre, dbias, ws, rows, N, N, mode,
                         make_drop(drop_p, seed, ctr_hi));
}

// out[N] += sum_rows x[rows, N]   (bias gradients).  ws: t4r_colreduce_ws_floats(rows, N)
extern "C" int t4r_colsum(void* stream, const float* x, float* out, float* ws, long rows, int N, long ld) {
    return colred_launch((hipStream_t)stream, x, nullptr, nullptr, out, ws, rows, N, ld, 2, make_drop(0.f, 0, 0));
}

// ---------------------------------------------------------------- element-wise dropout
// out = x * mask/(1-p)   (HF XLNetModel: dropout on inputs_embeds :1116, pos_emb :1143, final :1177).
// The same kernel is its own backward (apply to the incoming gradient).  `rep`: x is broadcast
// `rep` times (pos_emb [2L,D] -> [B,2L,D]: one independent mask per batch row, as the reference's
// batch-expanded pos_emb gets).  mask_out (uint8, may be NULL) exports the keep mask for tests.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                       unsigned char* __restrict__ mask_out, long n,
                                                       long n_src, DropCfg drop) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 m = drop_scale4(drop, (unsigned long long)i);
    const float mm[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i + e < n) {
            if (out) out[i + e] = x[(i + e) % n_src] * mm[e];
            if (mask_out) mask_out[i + e] = mm[e] != 0.f;
        }
    }
}

extern "C" int t4r_dropout(void* stream, const float* x, float* out, unsigned char* mask_out, long n,
                           long n_src, float p, unsigned long long seed, unsigned long long ctr_hi) {
    if (n <= 0) return 0;
    T4R_CHECK_ARG(p >= 0.f && p < 1.f && n_src > 0, "dropout: p in [0,1), n_src > 0");
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, out, mask_out, n, n_src, make_drop(p, seed, ctr_hi));
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- learned position (+ token type) embeddings
// out[t, :] = x[t, :] + pos[t % L, :] (+ tt[:])      GPT-2: inputs_embeds + wpe (HF gpt2 :576-577)
//                                                     BERT : + position + token_type(0) (HF bert embeddings)
__global__ __launch_bounds__(256) void add_pos_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                           const float* __restrict__ tt, float* __restrict__ out,
                                                           long ntok, int L, int D) {
    const int dq = D >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntok * dq) return;
    const long t = i / dq;
    const int c = (int)(i % dq) * 4;
    const int l = (int)(t % L);
    float4 v = *reinterpret_cast<const float4*>(x + t * D + c);
    const float4 p = *reinterpret_cast<const float4*>(pos + (long)l * D + c);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    if (tt) {
        const float4 q = *reinterpret_cast<const float4*>(tt + c);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *reinterpret_cast<float4*>(out + t * D + c) = v;
}
// d pos[l, :] += sum_b dy[b, l, :]   (16 batch chunks, one atomic per (chunk, l, column))
__global__ __launch_bounds__(256) void add_pos_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dpos,
                                                           int B, int L, int D) {
    const int l = blockIdx.x, chunk = blockIdx.y, nchunk = gridDim.y;
    for (int c = threadIdx.x; c < D; c += 256) {
        float acc = 0.f;
        for (int b = chunk; b < B; b += nchunk) acc += dy[((long)b * L + l) * D + c];
        atomicAdd(dpos + (long)l * D + c, acc);
    }
}
extern "C" int t4r_add_pos_fwd(void* stream, const float* x, const float* pos, const float* token_type,
                               float* out, int B, int L, int D) {
    const long ntok = (long)B * L;
    if (ntok == 0) return 0;
    T4R_CHECK_ARG(D % 4 == 0, "add_pos: D must be a multiple of 4");
    const long n = ntok * (D / 4);
    hipLaunchKernelGGL(add_pos_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       pos, token_type, out, ntok, L, D);
    T4R_LAUNCH_CHECK();
    return 0;
}
extern "C" int t4r_add_pos_bwd(void* stream, const float* dy, float* d_pos, int B, int L, int D) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(add_pos_bwd_kernel, dim3(L, B < 16 ? B : 16), dim3(256), 0, (hipStream_t)stream, dy, d_pos, B,
                       L, D);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- fused Adam over a flat buffer
// torch.optim.Adam (amsgrad=False, maximize=False): with step t (1-based)
//   g = grad (+ wd * p) ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// grad_scale multiplies grad first (1/world_size for the DP mean).  Optionally zeroes the grad.
// AMAX (round 6): the launch also leaves, per workgroup, the largest |p| AFTER the update among the elements [amax_lo, amax_hi)
// in amax_part[blockIdx.x] -- the tied item table's maximum, which the next step's head needs to position its fp16 images
// (csrc/head_split.hip: split_w_images_kernel reduces the <= 1024 partials) and used to get from a memset + a 21 us pass over
// the 51 MB the optimizer has just streamed.  Plain stores, one slot per workgroup (same-address atomics serialise).
template <bool AMAX>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt,
                                                    float grad_scale, int zero_grad, long amax_lo, long amax_hi,
                                                    float* __restrict__ amax_part) {
    float mx = 0.f;
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
                if (AMAX && i + e >= amax_lo && i + e < amax_hi) mx = fmaxf(mx, fabsf(P[e]));
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
                if (AMAX && j >= amax_lo && j < amax_hi) mx = fmaxf(mx, fabsf(p[j]));
                if (zero_grad) g[j] = 0.f;
            }
        }
    }
    if (AMAX) {
        __shared__ float sh[4];
        mx = wave_max(mx);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) amax_part[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
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
    hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param,
                       grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                       grad_scale, zero_grad, 0L, 0L, nullptr);
    T4R_LAUNCH_CHECK();
    return 0;
}
// the same step; also amax_part[b] = max |param[i]| after the update over i in [amax_lo, amax_hi) seen by workgroup b, for
// b < the returned number of workgroups (<= 1024 = the capacity the caller provides); < 0: error
extern "C" int t4r_adam_step_amax(void* stream, float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, int step,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                  int zero_grad, long amax_lo, long amax_hi, float* amax_part) {
    if (n <= 0) return 0;
    if (!(step >= 1)) { t4r_set_error("adam: step is 1-based"); return -1; }
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 != 0 || !amax_part ||
        amax_lo < 0 || amax_hi > n || amax_lo >= amax_hi) {
        t4r_set_error("adam_step_amax: buffers must be 16-byte aligned, the range inside the buffer, amax_part non-null");
        return -1;
    }
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 512) blocks = 512;      // (the consumer reduces the partials in every workgroup: 512 x 4 bytes)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, zero_grad, amax_lo, amax_hi,
                       amax_part);
    if (hipGetLastError() != hipSuccess) { t4r_set_error("adam_step_amax: launch failed"); return -1; }
    return (int)blocks;
}
