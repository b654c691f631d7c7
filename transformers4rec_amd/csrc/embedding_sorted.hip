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

// ---------------------------------------------------------------------------------------------- one-launch LDS sort (round 6)
// The (row id, lookup index) pairs of ONE workgroup-sized problem -- n <= 32 768 lookups, bits(rows) + bits(n - 1) <= 32: the
// item table of BASELINE configs[1] (20 480 lookups of a 100 001-row table) -- sorted by ONE launch instead of hipCUB's eight
// (key build + merge-sort passes, ~38 us on the caller's stream in front of every step's body): a pair is one 32-bit word
// key << IB | index, kept in LDS; two stable counting passes over 9-bit digits of the key.
//   * 16 waves, wave w owns the CONTIGUOUS chunk w of the current order (stable: the global order is wave-major), its words in
//     registers (one batch of loads, no dependent chain);
//   * counting: a wave walks its chunk 64 words at a time; lanes with the same digit find each other by nine wave ballots (one
//     per digit bit), the lowest of them adds their number to the wave's OWN counter row cnt[w][digit] (no atomics);
//   * one thread per digit turns the 16 rows into exclusive prefixes over the waves, one wave scans the 512 digit totals;
//   * scatter: the same walk; position = digit base + this wave's running prefix + rank among the equal lanes of the step.
// Pass 1 reads the ids from memory and scatters the words into LDS, pass 2 reads them in order from LDS and scatters
// keys_sorted / perm to memory.  Same result as the stable library sort, bit for bit (both are THE stable order).
#define T4R_LSORT_MAXN 32768
#define T4R_LSORT_STEPS (T4R_LSORT_MAXN / 1024)          // 64-word steps of a wave's chunk at the largest n
__device__ __forceinline__ unsigned long long lsort_match(unsigned d, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}
// one counting pass.  fetch(s) = this lane's word of step s of its wave's chunk (pass 1: built from the id in memory, pass 2:
// read from the LDS buffer).  Both walks are ROLLED loops over groups of four steps with the next group's words requested
// before the current group is processed (the fully unrolled form -- 32 steps x 2 walks x 2 passes with a 64-bit ballot mask per
// digit bit -- spilled 15 k registers and ran 1.9 ms).
template <bool TO_GLOBAL, int G, class Fetch>
__device__ __forceinline__ void lsort_pass(Fetch fetch, int steps, int begin, int n, int shift, int ib, unsigned short* cnt,
                                           unsigned* tot, unsigned* buf_out, int* keys_out, int* perm_out, int w, int lane, int tid) {
    unsigned short* my = cnt + w * 512;
    // zero this wave's counter row (512 x 2 bytes = 64 lanes x 16 bytes)
    reinterpret_cast<uint4*>(my)[lane] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    unsigned cur[G], nxt[G];
#pragma unroll
    for (int u = 0; u < G; ++u) cur[u] = fetch(min(u, steps - 1));
#pragma unroll 1
    for (int g0 = 0; g0 < steps; g0 += G) {
#pragma unroll
        for (int u = 0; u < G; ++u) nxt[u] = fetch(min(g0 + G + u, steps - 1));
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int st = g0 + u;
            if (st < steps) {                                       // wave-uniform
                const bool valid = begin + st * 64 + lane < n;
                const unsigned d = ((cur[u] >> ib) >> shift) & 511u;
                const unsigned long long peers = lsort_match(d, valid);
                if (valid && (peers & ((1ull << lane) - 1ull)) == 0ull) my[d] = (unsigned short)(my[d] + __popcll(peers));
            }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) cur[u] = nxt[u];
    }
    __syncthreads();
    if (tid < 512) {                                                // exclusive prefix over the waves, per digit
        unsigned run = 0;
#pragma unroll
        for (int ww = 0; ww < 16; ++ww) {
            const unsigned c = cnt[ww * 512 + tid];
            cnt[ww * 512 + tid] = (unsigned short)run;
            run += c;
        }
        tot[tid] = run;
    }
    __syncthreads();
    if (w == 0) {                                                   // exclusive scan of the 512 totals: 8 per lane + wave scan
        unsigned loc[8], sum = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { loc[e] = tot[lane * 8 + e]; sum += loc[e]; }
        unsigned inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        unsigned run = inc - sum;
#pragma unroll
        for (int e = 0; e < 8; ++e) { tot[lane * 8 + e] = run; run += loc[e]; }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < G; ++u) cur[u] = fetch(min(u, steps - 1));
#pragma unroll 1
    for (int g0 = 0; g0 < steps; g0 += G) {
#pragma unroll
        for (int u = 0; u < G; ++u) nxt[u] = fetch(min(g0 + G + u, steps - 1));
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int st = g0 + u;
            if (st < steps) {
                const bool valid = begin + st * 64 + lane < n;
                const unsigned wd = cur[u];
                const unsigned d = ((wd >> ib) >> shift) & 511u;
                const unsigned long long peers = lsort_match(d, valid);
                const unsigned below = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
                const unsigned pos = valid ? tot[d] + my[d] + below : 0u;
                if (valid) {
                    if (TO_GLOBAL) { keys_out[pos] = (int)(wd >> ib); perm_out[pos] = (int)(wd & ((1u << ib) - 1u)); }
                    else buf_out[pos] = wd;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);                 // every lane has read my[d] before its leader moves it on
                __builtin_amdgcn_wave_barrier();
                if (valid && below == 0u) my[d] = (unsigned short)(my[d] + __popcll(peers));
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) cur[u] = nxt[u];
    }
    __syncthreads();
}
__device__ __forceinline__ void lsort_body(const long* __restrict__ ids, int n, long rows, int padding_idx, int ib,
                                           int* __restrict__ keys_sorted, int* __restrict__ perm) {
    extern __shared__ unsigned lsort_smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int chunk = (((n + 15) / 16) + 63) / 64 * 64, steps = chunk / 64, begin = w * chunk;
    unsigned* buf = lsort_smem;                                           // [16 * chunk] words
    unsigned* tot = buf + 16 * chunk;                                     // [512]
    unsigned short* cnt = reinterpret_cast<unsigned short*>(tot + 512);   // [16][512]
    auto from_ids = [&](int st) -> unsigned {
        const int i = min(begin + st * 64 + lane, n - 1);
        const long id = ids[i];
        const unsigned key = (id == padding_idx || id < 0 || id >= rows) ? (unsigned)rows : (unsigned)id;
        return (key << ib) | (unsigned)i;
    };
    lsort_pass<false, 8>(from_ids, steps, begin, n, 0, ib, cnt, tot, buf, nullptr, nullptr, w, lane, tid);
    auto from_lds = [&](int st) -> unsigned { return buf[min(begin + st * 64 + lane, n - 1)]; };
    lsort_pass<true, 4>(from_lds, steps, begin, n, 9, ib, cnt, tot, nullptr, keys_sorted, perm, w, lane, tid);
}
__global__ __launch_bounds__(1024) void sort_ids_lds_kernel(const long* __restrict__ ids, int n, long rows, int padding_idx, int ib,
                                                             int* __restrict__ keys_sorted, int* __restrict__ perm) {
    lsort_body(ids, n, rows, padding_idx, ib, keys_sorted, perm);
}
// does the one-launch sort take this problem?  (key bits <= 18: two 9-bit passes; a word holds key and index)
static int lsort_index_bits(long n, long rows) {
    if (n < 1 || n > T4R_LSORT_MAXN || rows < 1 || rows >= (1L << 18)) return 0;
    int ib = 1, kb = 1;
    while ((1L << ib) < n) ++ib;
    while ((1L << kb) <= rows) ++kb;
    return ib + kb <= 32 ? ib : 0;
}

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
    if (const int ib = lsort_index_bits(n, rows)) {        // one launch, words in LDS (no workspace used)
        const int chunk = (int)((((n + 15) / 16) + 63) / 64 * 64);
        const size_t smem = (size_t)16 * chunk * 4 + 512 * 4 + 16 * 512 * 2;
        static T4rLdsAttr attr;
        t4r_ensure_dynamic_lds((const void*)sort_ids_lds_kernel, smem, attr);
        hipLaunchKernelGGL(sort_ids_lds_kernel, dim3(1), dim3(1024), smem, st, ids, (int)n, rows, padding_idx, ib, keys_sorted, perm);
        T4R_LAUNCH_CHECK();
        return 0;
    }
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

// one workgroup per table: the F sorts of an input block run side by side in ONE launch (round 6; every table one-workgroup sized)
__global__ __launch_bounds__(1024) void sort_ids_lds_multi_kernel(SortMulti p, int ib, int* __restrict__ keys, int* __restrict__ perm) {
    const int f = blockIdx.x;
    lsort_body(p.ids[f], (int)p.n, p.rows[f], p.pad[f], ib, keys + (long)f * p.n, perm + (long)f * p.n);
}
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
    {
        int ib = lsort_index_bits(n, rows[0]);
        for (int f = 1; f < F && ib; ++f)
            if (lsort_index_bits(n, rows[f]) != ib) ib = 0;        // (the index bits depend on n only; 0 = a table does not fit)
        if (ib) {       // every table is a one-workgroup problem: F workgroups, one launch, no workspace
            const int chunk = (int)((((n + 15) / 16) + 63) / 64 * 64);
            const size_t smem = (size_t)16 * chunk * 4 + 512 * 4 + 16 * 512 * 2;
            static T4rLdsAttr attr;
            t4r_ensure_dynamic_lds((const void*)sort_ids_lds_multi_kernel, smem, attr);
            hipLaunchKernelGGL(sort_ids_lds_multi_kernel, dim3(F), dim3(1024), smem, st, p, ib, keys_sorted, perm);
            T4R_LAUNCH_CHECK();
            return 0;
        }
    }
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
