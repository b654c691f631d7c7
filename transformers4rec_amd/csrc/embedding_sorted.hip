// Deterministic embedding-table gradient: sort the lookups by row id, then a segmented sum.
//
// Replaces the backward of the reference's per-feature `nn.Embedding` lookup
// (transformers4rec/torch/features/embedding.py:226-249 -> ATen embedding_dense_backward: a dense
// V x D zero fill + an index_add) and is also the local "apply" step of the row-sparse gradient
// exchange between data-parallel ranks (distributed.py).
//
// Why not atomics (embedding.hip: embedding_bwd_kernel): fp32 row atomics sum in arrival order, so
// two ranks -- or two runs -- round differently, and hot rows serialise on one L2 line
// (measured r01: 18 % of the HBM roofline at 1.3 M tokens on a 100 k-row table).  Here every table
// row has exactly ONE owner wave and a fixed summation order (ascending token index):
//   1. keys[t] = id (padding / out-of-range -> sentinel `rows`), stable radix sort of (key, t)
//      [hipCUB, only the significant bits]; depends on the ids only, so the host runs it in the
//      FORWARD pass and the backward starts from the sorted order;
//   2. pass A: one wave per "super-chunk" of 16*S consecutive sorted positions; all 16 gradient rows
//      of a chunk are requested before the first is consumed (16 x dim/64 loads in flight per lane),
//      then runs of equal keys are summed in registers.  A run that lies inside the super-chunk is
//      the row's only run: plain read-modify-write of the table row (no atomic).  A run cut by a
//      super-chunk border is parked as a partial sum;
//   3. pass B: the wave whose super-chunk STARTS a cut run walks the following partials in order and
//      adds the total to the table row.
// Algorithmic bytes: n*(4 key + 4 perm) + n*dim*4 gradient rows + 2*U*dim*4 table RMW (U unique ids).
#include "t4r_common.h"
#include <hipcub/hipcub.hpp>

static size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

__global__ __launch_bounds__(256) void emb_keys_kernel(const long* __restrict__ ids, long n, long rows,
                                                        int padding_idx, int* __restrict__ keys,
                                                        int* __restrict__ idx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long id = ids[i];
    keys[i] = (id == padding_idx || id < 0 || id >= rows) ? (int)rows : (int)id;
    idx[i] = (int)i;
}

// (Round 6 built a one-launch LDS radix sort for workgroup-sized problems -- key << IB | index words in LDS, two stable 9-bit
// counting passes, equal digits matched by wave ballots, per-wave counter rows, no atomics: bit-identical to the library sort
// in tests -- and measured it at 70 us per launch against the 38 us of the library's eight launches: ONE workgroup is one CU,
// and 20 480 words x 4 walks x ~45 instructions per 64 words is ~190 k issue cycles on its four SIMDs whatever the LDS does.
// Removed again; a multi-workgroup form needs the grid-wide passes the library already has.  DESIGN.md section 4.3.)
struct SortLayout { size_t keys_in, idx_in, tmp, tmp_bytes, total; };
static SortLayout sort_layout(long n) {
    SortLayout l;
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += al256(b); return at; };
    l.keys_in = take((size_t)n * 4);
    l.idx_in = take((size_t)n * 4);
    size_t sb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sb, (int*)nullptr, (int*)nullptr, (int*)nullptr,
                                             (int*)nullptr, (int)n);
    l.tmp_bytes = sb;
    l.tmp = take(sb);
    l.total = o;
    return l;
}

extern "C" long t4r_sort_ids_ws_bytes(long n) { return n <= 0 ? 0 : (long)sort_layout(n).total; }

// keys_sorted[i] (ascending; invalid lookups carry `rows` and come last), perm[i] = index of the lookup.
// Stable: equal ids keep ascending lookup order -- that IS the summation order of the gradient.
extern "C" int t4r_sort_ids(void* stream, const long* ids, long n, long rows, int padding_idx,
                            int* keys_sorted, int* perm, void* ws, long ws_bytes) {
    if (n <= 0) return 0;
    T4R_CHECK_ARG(ids && keys_sorted && perm && ws, "sort_ids: null pointer");
    T4R_CHECK_ARG(n < (1L << 31) && rows > 0 && rows < (1L << 31) - 1, "sort_ids: sizes must fit 31 bits");
    const SortLayout l = sort_layout(n);
    T4R_CHECK_ARG(ws_bytes >= (long)l.total, "sort_ids: workspace too small (t4r_sort_ids_ws_bytes)");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    int* keys_in = (int*)(w + l.keys_in);
    int* idx_in = (int*)(w + l.idx_in);
    hipLaunchKernelGGL(emb_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, n, rows,
                       padding_idx, keys_in, idx_in);
    int end_bit = 1;
    while (end_bit < 31 && (1L << end_bit) <= rows) ++end_bit;     // bits of the largest key (= rows)
    size_t tb = l.tmp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(w + l.tmp, tb, keys_in, keys_sorted, idx_in, perm, (int)n, 0, end_bit,
                                           st) != hipSuccess) {
        t4r_set_error("sort_ids: device radix sort failed");
        return -1;
    }
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------- several tables, ONE sort
// A multi-feature input block (BASELINE configs[2]: item id + three more categoricals) sorts the lookups of every table in
// the forward pass: F sorts of n pairs are ~10 small launches EACH on the caller's stream (hipCUB picks its merge sort at
// n = 20 480).  The lookups of all F tables sorted as ONE array of F n pairs -- feature f's keys moved into its own range
// [off_f, off_f + rows_f], off_f = sum_{g<f} (rows_g + 1) -- leave every feature's n pairs in positions [f n, (f + 1) n) of the
// result, in the order its own sort would have given (the sort is stable, the ranges are disjoint): a third of the launches.
constexpr int kSortMaxFeatures = 16;
struct SortMulti { const long* ids[kSortMaxFeatures]; long rows[kSortMaxFeatures]; long off[kSortMaxFeatures]; int pad[kSortMaxFeatures]; int F; long n; };

__global__ __launch_bounds__(256) void emb_keys_multi_kernel(SortMulti p, int* __restrict__ keys, int* __restrict__ idx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n * p.F) return;
    const int f = (int)(i / p.n);
    const long id = p.ids[f][i - f * p.n];
    const long rows = p.rows[f];
    keys[i] = (int)(p.off[f] + ((id == p.pad[f] || id < 0 || id >= rows) ? rows : id));
    idx[i] = (int)i;
}
// back to each feature's own row ids and lookup indices
__global__ __launch_bounds__(256) void emb_keys_multi_fix_kernel(SortMulti p, int* __restrict__ keys, int* __restrict__ perm) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n * p.F) return;
    const int f = (int)(i / p.n);
    keys[i] -= (int)p.off[f];
    perm[i] -= (int)(f * p.n);
}

extern "C" long t4r_sort_ids_multi_ws_bytes(long n, int F) { return (n <= 0 || F <= 0) ? 0 : (long)sort_layout(n * F).total; }

// ids: HOST array of F device pointers (n int64 lookups each); rows / padding_idx: host arrays [F].
// keys_sorted / perm: [F * n]; feature f's result is the slice [f n, (f + 1) n) -- what t4r_sort_ids gives for it alone.
extern "C" int t4r_sort_ids_multi(void* stream, const long* const* ids, int F, long n, const long* rows, const int* padding_idx,
                                  int* keys_sorted, int* perm, void* ws, long ws_bytes) {
    if (n <= 0 || F <= 0) return 0;
    T4R_CHECK_ARG(ids && rows && padding_idx && keys_sorted && perm && ws, "sort_ids_multi: null pointer");
    T4R_CHECK_ARG(F <= kSortMaxFeatures, "sort_ids_multi: at most 16 tables per call");
    SortMulti p;
    p.F = F; p.n = n;
    long off = 0;
    for (int f = 0; f < F; ++f) {
        T4R_CHECK_ARG(ids[f] && rows[f] > 0, "sort_ids_multi: null ids or empty table");
        p.ids[f] = ids[f]; p.rows[f] = rows[f]; p.pad[f] = padding_idx[f]; p.off[f] = off;
        off += rows[f] + 1;
    }
    T4R_CHECK_ARG(n * F < (1L << 31) && off < (1L << 31) - 1, "sort_ids_multi: sizes must fit 31 bits");
    const SortLayout l = sort_layout(n * F);
    T4R_CHECK_ARG(ws_bytes >= (long)l.total, "sort_ids_multi: workspace too small (t4r_sort_ids_multi_ws_bytes)");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    int* keys_in = (int*)(w + l.keys_in);
    int* idx_in = (int*)(w + l.idx_in);
    const unsigned grid = (unsigned)((n * F + 255) / 256);
    hipLaunchKernelGGL(emb_keys_multi_kernel, dim3(grid), dim3(256), 0, st, p, keys_in, idx_in);
    int end_bit = 1;
    while (end_bit < 31 && (1L << end_bit) < off) ++end_bit;       // bits of the largest key (= off - 1)
    size_t tb = l.tmp_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(w + l.tmp, tb, keys_in, keys_sorted, idx_in, perm, (int)(n * F), 0, end_bit, st) !=
        hipSuccess) {
        t4r_set_error("sort_ids_multi: device radix sort failed");
        return -1;
    }
    hipLaunchKernelGGL(emb_keys_multi_fix_kernel, dim3(grid), dim3(256), 0, st, p, keys_sorted, perm);
    T4R_LAUNCH_CHECK();
    return 0;
}

// per-lane slice of a gradient row: NV vectors of VEC floats, vector j at column cb + (lane + 64*j)*VEC
template <int VEC> struct FV { float v[VEC]; };
template <int VEC>
__device__ __forceinline__ FV<VEC> ldv(const float* p) {
    FV<VEC> r;
    if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1 % VEC] = t.y; r.v[2 % VEC] = t.z; r.v[3 % VEC] = t.w; }
    else if (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1 % VEC] = t.y; }
    else r.v[0] = *p;
    return r;
}

// pass A.  grid.x * 4 waves >= number of super-chunks (CH*S sorted positions each); grid.y = column blocks
// of 64*VEC*NV columns.  VEC > 1 needs dim, W, col multiples of VEC and a 4*VEC-byte aligned dout.
template <int VEC, int NV, int CH>
__global__ __launch_bounds__(256) void emb_seg_sum_kernel(const float* __restrict__ dout,
                                                           const int* __restrict__ keys,
                                                           const int* __restrict__ perm,
                                                           float* __restrict__ dtable,
                                                           float* __restrict__ partial, long n, int W, int col,
                                                           int dim, int rows, int ids_div, int S) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long span = (long)CH * S;
    const long p0 = w * span;
    if (p0 >= n) return;
    const long p1 = min(n, p0 + span);
    const int cb = blockIdx.y * 64 * VEC * NV;              // first column of this column block
    const int kp = p0 > 0 ? keys[p0 - 1] : -1, kn = p1 < n ? keys[p1] : -2;
    int cur = keys[p0];
    const bool left_open = kp == cur;
    bool first = true;
    float acc[NV][VEC];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[j][e] = 0.f;

    auto flush = [&](int k, bool is_first, bool is_last) __attribute__((always_inline)) {
        const bool lo = is_first && left_open, ro = is_last && kn == k;
        float* dst = nullptr;
        bool add = false;
        if (lo) dst = partial + (w * 2 + 0) * (long)dim;
        else if (ro) dst = partial + (w * 2 + 1) * (long)dim;
        else if (k < rows) { dst = dtable + (long)k * dim; add = true; }
        if (dst) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = cb + (lane + 64 * j) * VEC;
                if (c < dim) {          // dim % VEC == 0: a vector is inside or outside as a whole
#pragma unroll
                    for (int e = 0; e < VEC; ++e) dst[c + e] = add ? dst[c + e] + acc[j][e] : acc[j][e];
                }
            }
        }
    };

    for (long c0 = p0; c0 < p1; c0 += CH) {
        const int cnt = (int)min((long)CH, p1 - c0);
        int myk = 0, myp = 0;
        if (lane < cnt) { myk = keys[c0 + lane]; myp = perm[c0 + lane]; }
        FV<VEC> v[CH][NV];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int pr = __builtin_amdgcn_readlane(myp, i);
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[i][j].v[e] = 0.f;
            if (i < cnt) {
                const float* src = dout + ((long)pr * ids_div) * W + col + cb + lane * VEC;
                if (ids_div == 1) {
#pragma unroll
                    for (int j = 0; j < NV; ++j)
                        if (cb + (lane + 64 * j) * VEC < dim) v[i][j] = ldv<VEC>(src + 64 * j * VEC);
                } else {
                    for (int l = 0; l < ids_div; ++l) {
#pragma unroll
                        for (int j = 0; j < NV; ++j)
                            if (cb + (lane + 64 * j) * VEC < dim) {
                                const FV<VEC> t = ldv<VEC>(src + (long)l * W + 64 * j * VEC);
#pragma unroll
                                for (int e = 0; e < VEC; ++e) v[i][j].v[e] += t.v[e];
                            }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (i < cnt) {
                const int k = __builtin_amdgcn_readlane(myk, i);
                if (k != cur) {
                    flush(cur, first, false);
                    first = false;
                    cur = k;
#pragma unroll
                    for (int j = 0; j < NV; ++j)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[j][e] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[j][e] += v[i][j].v[e];
            }
        }
    }
    flush(cur, first, true);
}

// pass B: runs cut by super-chunk borders.  One 512-thread workgroup per super-chunk; only the one whose
// super-chunk holds the START of a cut run works.  The run's end is found by binary search in the sorted
// keys; its partial rows (right partial of this super-chunk, left partials of the following ones) are
// summed by 8 waves over contiguous row ranges and combined in a fixed order (deterministic tree).
#define EFIX_WAVES 8
__global__ __launch_bounds__(64 * EFIX_WAVES) void emb_seg_fix_kernel(const int* __restrict__ keys,
                                                                       float* __restrict__ dtable,
                                                                       const float* __restrict__ partial, long n,
                                                                       int dim, int rows, long span) {
    __shared__ float red[EFIX_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long w = blockIdx.x;
    const long p0 = w * span;
    if (p0 >= n) return;
    const long p1 = min(n, p0 + span);
    const int kf = keys[p0], kl = keys[p1 - 1];
    const int kp = p0 > 0 ? keys[p0 - 1] : -1, kn = p1 < n ? keys[p1] : -2;
    if (kn != kl) return;                     // the last run ends here
    if (kf == kl && kp == kf) return;         // the whole super-chunk continues a run started earlier
    if (kl >= rows) return;                   // padding / out-of-range lookups carry no gradient
    long lo = p1, hi = n;                     // first position >= p1 whose key differs from kl
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (keys[mid] == kl) lo = mid + 1; else hi = mid;
    }
    const long w_last = (lo - 1) / span;      // last super-chunk of the run
    const long M = w_last - w;                // left partials of super-chunks w+1 .. w_last
    const long per = (M + EFIX_WAVES - 1) / EFIX_WAVES;
    const long r0 = min(M, wave * per), r1 = min(M, r0 + per);
    for (int cb = 0; cb < dim; cb += 256) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (long r = r0; r < r1; ++r) {
            const float* src = partial + ((w + 1 + r) * 2 + 0) * (long)dim + cb + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (cb + lane + 64 * j < dim) a[j] += src[64 * j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][lane + 64 * j] = a[j];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = cb + lane + 64 * j;
                if (c < dim) {
                    float t = partial[(w * 2 + 1) * (long)dim + c];
#pragma unroll
                    for (int v = 0; v < EFIX_WAVES; ++v) t += red[v][lane + 64 * j];
                    dtable[(long)kl * dim + c] += t;
                }
            }
        }
        __syncthreads();
    }
}

// super-chunk geometry: CH rows are in flight per wave at a time (8 when a lane holds 8 floats of a row)
static int seg_CH(int dim) { return dim > 256 ? 8 : 16; }
static int seg_S(long n, int dim) {
    long s = n / ((long)seg_CH(dim) * 8192);
    return (int)(s < 1 ? 1 : (s > 8 ? 8 : s));
}

extern "C" long t4r_embedding_bwd_sorted_ws_floats(long n, int dim) {
    if (n <= 0) return 0;
    const long span = (long)seg_CH(dim) * seg_S(n, dim);
    return ((n + span - 1) / span) * 2 * (long)dim;
}

// dtable[key] += sum over the lookups with that key of the gradient row of lookup `perm`:
// gradient row of lookup p = sum_{l < ids_div} dout[(p*ids_div + l) * W + col : +dim]
// (ids_div = 1 for a sequence feature, L for a per-session feature broadcast over the sequence).
extern "C" int t4r_embedding_bwd_sorted(void* stream, const float* dout, const int* keys_sorted,
                                        const int* perm, float* dtable, long n, int W, int col, int dim,
                                        long rows, int ids_div, float* ws) {
    if (n <= 0) return 0;
    T4R_CHECK_ARG(dout && keys_sorted && perm && dtable && ws, "embedding_bwd_sorted: null pointer");
    T4R_CHECK_ARG(ids_div >= 1 && dim >= 1 && col >= 0 && col + dim <= W, "embedding_bwd_sorted: bad slice");
    T4R_CHECK_ARG(rows < (1L << 31) - 1, "embedding_bwd_sorted: rows must fit 31 bits");
    hipStream_t st = (hipStream_t)stream;
    const int CH = seg_CH(dim), S = seg_S(n, dim);
    const long span = (long)CH * S;
    const long nw = (n + span - 1) / span;
    const unsigned gx = (unsigned)((nw + 3) / 4);
    auto vec_ok = [&](int v) { return dim % v == 0 && W % v == 0 && col % v == 0 && (uintptr_t)dout % (4 * v) == 0; };
#define ESEG_LAUNCH(VEC, NV, CHN)                                                                             \
    hipLaunchKernelGGL((emb_seg_sum_kernel<VEC, NV, CHN>),                                                    \
                       dim3(gx, (unsigned)((dim + 64 * VEC * NV - 1) / (64 * VEC * NV))), dim3(256), 0, st, dout, \
                       keys_sorted, perm, dtable, ws, n, W, col, dim, (int)rows, ids_div, S)
    if (dim > 256) { if (vec_ok(4)) ESEG_LAUNCH(4, 2, 8); else ESEG_LAUNCH(1, 8, 8); }
    else if (dim > 128) { if (vec_ok(4)) ESEG_LAUNCH(4, 1, 16); else ESEG_LAUNCH(1, 4, 16); }
    else if (dim > 64) { if (vec_ok(2)) ESEG_LAUNCH(2, 1, 16); else ESEG_LAUNCH(1, 2, 16); }
    else ESEG_LAUNCH(1, 1, 16);
#undef ESEG_LAUNCH
    T4R_LAUNCH_CHECK();
    hipLaunchKernelGGL(emb_seg_fix_kernel, dim3((unsigned)nw), dim3(64 * EFIX_WAVES), 0, st, keys_sorted, dtable, ws,
                       n, dim, (int)rows, span);
    T4R_LAUNCH_CHECK();
    return 0;
}
