/* TEST INFRASTRUCTURE ONLY -- the arithmetic of oracle/device_rng.py in C (gcc, OpenMP) for full-size tensors.
 *
 * Philox4x32-10 (Salmon et al., SC'11; Random123 kat_vectors pin it in tests/test_device_rng_oracle.py), the dropout-site
 * keep rule and the MLM training draws of the HIP path, restated from their documented contract
 * (transformers4rec_amd/csrc/t4r_common.h "Dropout masks", csrc/masking.hip header).  The reference itself draws from
 * torch's generator (transformers4rec/torch/masking.py:425-459; nn.Dropout sites of HF modeling_xlnet.py:132,147,301,303,
 * 1116,1143,1177); this restatement lets the CPU oracle take the decisions the device took.
 *
 *   gcc -O2 -fopenmp -shared -fPIC oracle/device_rng.c -o oracle/_build/libt4r_oracle_rng.so      (oracle/build_c.py)
 */
#include <stdint.h>

static void philox(uint64_t seed, uint64_t lo, uint64_t hi, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)lo, c1 = (uint32_t)(lo >> 32), c2 = (uint32_t)hi, c3 = (uint32_t)(hi >> 32);
    uint32_t a = (uint32_t)seed, b = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void t4r_oracle_philox(uint64_t seed, uint64_t lo, uint64_t hi, uint32_t* out4) { philox(seed, lo, hi, out4); }

/* keep[idx] = word (idx & 3) of block (idx >> 2, ctr_hi) >= thr */
void t4r_oracle_dropout_keep(uint64_t seed, uint64_t ctr_hi, uint64_t n, uint32_t thr, uint8_t* keep) {
    const int64_t nblk = (int64_t)((n + 3) / 4);
#pragma omp parallel for schedule(static)
    for (int64_t blk = 0; blk < nblk; ++blk) {
        uint32_t w[4];
        philox(seed, (uint64_t)blk, ctr_hi, w);
        for (int k = 0; k < 4; ++k) {
            const uint64_t idx = (uint64_t)blk * 4 + k;
            if (idx < n) keep[idx] = w[k] >= thr;
        }
    }
}

static float unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

/* bern[b, l] = unit(word x of block (offset + b L + l, 0)) < p ; (u1, u2)[b] = unit(words x, y of block (offset + b, 1)) */
void t4r_oracle_mlm_draws(uint64_t seed, uint64_t offset, int64_t B, int64_t L, float p, uint8_t* bern, float* u1, float* u2) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < B * L; ++i) {
        uint32_t w[4];
        philox(seed, offset + (uint64_t)i, 0, w);
        bern[i] = unit(w[0]) < p;
    }
    for (int64_t b = 0; b < B; ++b) {
        uint32_t w[4];
        philox(seed, offset + (uint64_t)b, 1, w);
        u1[b] = unit(w[0]);
        u2[b] = unit(w[1]);
    }
}
