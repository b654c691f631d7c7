// StochasticSwapNoise: train-time replacement of sequence-feature values by values sampled
// (without replacement) from the other non-padded positions of the same feature in the batch.
//
// Reference behaviour restated (transformers4rec/torch/tabular/transformations.py:29-93):
//   mask      = item_id != pad_token                       (config/schema.py:59-66; [:, 0] for a
//                                                           per-session feature, :63-65)
//   replace   = bernoulli(replacement_prob) & mask
//   masked    = masked_select(x, mask)                      row-major order of the non-pad values
//   out[pos_k] = masked[randperm(len(masked))[k]]           pos_k = k-th replaced position, row-major
// Integer / byte work, HBM-bound: two prefix sums (hipCUB device scan), one key sort, one gather.
// The draws are either injected by the caller (`bern`, `perm`: parity tests replay the reference's
// own torch.bernoulli / torch.randperm results) or produced on the device: Philox4x32-10 uniforms
// for the Bernoulli trial and a uniformly random permutation obtained by radix-sorting the
// positions by 64-bit Philox keys (pad positions carry the maximal key, so the first nnz sorted
// entries are a random permutation of the non-pad positions).
#include "t4r_common.h"
#include <hipcub/hipcub.hpp>

__global__ __launch_bounds__(256) void ssn_flags_kernel(
    const long* __restrict__ item_ids, long pad_token, long mask_stride, const unsigned char* __restrict__ bern,
    float p, unsigned long long seed, unsigned long long ctr_hi, long n, int* __restrict__ valid,
    int* __restrict__ rep, unsigned long long* __restrict__ keys, int* __restrict__ idx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool v = item_ids[i * mask_stride] != pad_token;
    const Philox rng(seed);
    const uint4 r = rng((unsigned long long)i, ctr_hi);
    const bool b = bern ? bern[i] != 0 : u32_to_unit(r.x) < p;
    valid[i] = v ? 1 : 0;
    rep[i] = (v && b) ? 1 : 0;
    if (keys) {
        keys[i] = v ? (((unsigned long long)r.y << 32) | r.z) >> 1 : ~0ull;   // valid keys < 2^63
        idx[i] = (int)i;
    }
}

// explicit-permutation mode: cidx[rank among valid] = position
__global__ __launch_bounds__(256) void ssn_compact_kernel(const int* __restrict__ valid,
                                                           const int* __restrict__ vrank, long n,
                                                           int* __restrict__ cidx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && valid[i]) cidx[vrank[i]] = (int)i;
}

template <typename T>
__global__ __launch_bounds__(256) void ssn_apply_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                         const int* __restrict__ rep,
                                                         const int* __restrict__ rrank,
                                                         const int* __restrict__ src_pos,   // sorted idx | cidx
                                                         const long* __restrict__ perm, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T v = x[i];
    if (rep[i]) {
        const int k = rrank[i];
        v = x[perm ? src_pos[perm[k]] : src_pos[k]];
    }
    out[i] = v;
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct SsnLayout {
    size_t valid, rep, vrank, rrank, idx_in, idx_out, keys_in, keys_out, tmp, tmp_bytes, total;
};
static SsnLayout ssn_layout(long n) {
    SsnLayout l;
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += align256(b); return at; };
    l.valid = take(n * 4); l.rep = take(n * 4); l.vrank = take(n * 4); l.rrank = take(n * 4);
    l.idx_in = take(n * 4); l.idx_out = take(n * 4); l.keys_in = take(n * 8); l.keys_out = take(n * 8);
    size_t scan_b = 0, sort_b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_b, (int*)nullptr, (int*)nullptr, (int)n);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_b, (unsigned long long*)nullptr,
                                             (unsigned long long*)nullptr, (int*)nullptr, (int*)nullptr, (int)n);
    l.tmp_bytes = scan_b > sort_b ? scan_b : sort_b;
    l.tmp = take(l.tmp_bytes);
    l.total = o;
    return l;
}

extern "C" long t4r_swap_noise_ws_bytes(long n) { return n <= 0 ? 0 : (long)ssn_layout(n).total; }

extern "C" int t4r_swap_noise(void* stream, const void* x, void* out, int elem_bytes, long n,
                              const long* item_ids, long pad_token, long mask_stride, float p,
                              const unsigned char* bern, const long* perm, unsigned long long seed,
                              unsigned long long ctr_hi, void* ws, long ws_bytes) {
    if (n <= 0) return 0;
    T4R_CHECK_ARG(elem_bytes == 4 || elem_bytes == 8, "swap_noise: elements are int64 ids or fp32 values");
    T4R_CHECK_ARG(x && out && item_ids && ws, "swap_noise: null pointer");
    T4R_CHECK_ARG(n < (1L << 31), "swap_noise: at most 2^31 positions");
    T4R_CHECK_ARG(p >= 0.f && p <= 1.f, "swap_noise: replacement_prob in [0, 1]");
    const SsnLayout l = ssn_layout(n);
    T4R_CHECK_ARG(ws_bytes >= (long)l.total, "swap_noise: workspace too small (t4r_swap_noise_ws_bytes)");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    int* valid = (int*)(w + l.valid); int* rep = (int*)(w + l.rep);
    int* vrank = (int*)(w + l.vrank); int* rrank = (int*)(w + l.rrank);
    int* idx_in = (int*)(w + l.idx_in); int* idx_out = (int*)(w + l.idx_out);
    unsigned long long* keys_in = (unsigned long long*)(w + l.keys_in);
    unsigned long long* keys_out = (unsigned long long*)(w + l.keys_out);
    void* tmp = w + l.tmp;
    size_t tmp_b = l.tmp_bytes;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(ssn_flags_kernel, grid, block, 0, st, item_ids, pad_token, mask_stride, bern, p, seed, ctr_hi, n,
                       valid, rep, perm ? nullptr : keys_in, idx_in);
    if (hipcub::DeviceScan::ExclusiveSum(tmp, tmp_b, rep, rrank, (int)n, st) != hipSuccess) {
        t4r_set_error("swap_noise: device scan failed");
        return -1;
    }
    const int* src_pos;
    if (perm) {
        tmp_b = l.tmp_bytes;
        if (hipcub::DeviceScan::ExclusiveSum(tmp, tmp_b, valid, vrank, (int)n, st) != hipSuccess) {
            t4r_set_error("swap_noise: device scan failed");
            return -1;
        }
        hipLaunchKernelGGL(ssn_compact_kernel, grid, block, 0, st, valid, vrank, n, idx_out);
        src_pos = idx_out;
    } else {
        tmp_b = l.tmp_bytes;
        if (hipcub::DeviceRadixSort::SortPairs(tmp, tmp_b, keys_in, keys_out, idx_in, idx_out, (int)n, 0, 64,
                                               st) != hipSuccess) {
            t4r_set_error("swap_noise: device sort failed");
            return -1;
        }
        src_pos = idx_out;
    }
    if (elem_bytes == 8)
        hipLaunchKernelGGL(ssn_apply_kernel<long>, grid, block, 0, st, (const long*)x, (long*)out, rep, rrank,
                           src_pos, perm, n);
    else
        hipLaunchKernelGGL(ssn_apply_kernel<float>, grid, block, 0, st, (const float*)x, (float*)out, rep,
                           rrank, src_pos, perm, n);
    T4R_LAUNCH_CHECK();
    return 0;
}

// strided column-block copy:  dst[r, 0:dim] = src[r, col:col+dim]   (dir 0)
//                             src[r, col:col+dim] = dst[r, 0:dim]   (dir 1)
// used to hand one feature's slice of a concatenated row to the per-feature LayerNorm backward
__global__ __launch_bounds__(256) void copy_cols_kernel(float* __restrict__ wide, long ldw, int col,
                                                         float* __restrict__ narrow, int dim, long rows,
                                                         int dir) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * dim) return;
    const long r = i / dim;
    const int c = (int)(i % dim);
    if (dir == 0) narrow[i] = wide[r * ldw + col + c];
    else wide[r * ldw + col + c] = narrow[i];
}

extern "C" int t4r_copy_cols(void* stream, float* wide, long ldw, int col, float* narrow, int dim,
                             long rows, int dir) {
    if (rows <= 0 || dim <= 0) return 0;
    T4R_CHECK_ARG(wide && narrow && col >= 0 && col + dim <= ldw, "copy_cols: bad slice");
    const long n = rows * dim;
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, wide, ldw, col, narrow, dim, rows, dir);
    T4R_LAUNCH_CHECK();
    return 0;
}

// gradient of the broadcast of a per-session (context) feature over the sequence:
//   out[b, 0:dim] = sum_l wide[(b*L + l), col:col+dim]
__global__ __launch_bounds__(256) void seq_sum_cols_kernel(const float* __restrict__ wide, long ldw,
                                                            int col, float* __restrict__ out, int dim,
                                                            long B, int L) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const long b = i / dim;
    const int c = (int)(i % dim);
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += wide[(b * L + l) * ldw + col + c];
    out[i] = acc;
}

extern "C" int t4r_seq_sum_cols(void* stream, const float* wide, long ldw, int col, float* out, int dim,
                                long B, int L) {
    if (B <= 0 || dim <= 0) return 0;
    T4R_CHECK_ARG(wide && out && col >= 0 && col + dim <= ldw && L >= 1, "seq_sum_cols: bad slice");
    const long n = B * dim;
    hipLaunchKernelGGL(seq_sum_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, wide, ldw, col, out, dim, B, L);
    T4R_LAUNCH_CHECK();
    return 0;
}
