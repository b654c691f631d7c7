// probe of gfx950's transposing LDS read (ds_read_b64_tr_b16): fills LDS with its own element indices and prints what every lane receives
// for three address patterns.  hipcc --offload-arch=gfx950 -O2 -o tools/bin/tr_probe tools/ds_read_tr_probe.hip ; tools/bin/tr_probe {0,1,2}
// Result (MI355X): in each group of 16 lanes, output lane c receives, for j = 0..3, element c % 4 of what lane 4 j + c / 4 addressed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main(int argc, char** argv) {
    int h[64]; unsigned short o[256];
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    for (int l = 0; l < 64; ++l) {
        if (mode == 0) h[l] = 4 * l;                       // linear: lane l reads elements 4l..4l+3
        else if (mode == 1) h[l] = (l & 15) * 64 + (l >> 4) * 4;   // 16 rows of pitch 64, 4 columns per group
        else h[l] = (l >> 4) * 64 + ((l & 3) * 16 + ((l & 15) >> 2) * 4);
    }
    int* da; unsigned short* dout;
    hipMalloc(&da, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(da, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %4d %4d %4d %4d\n", l, h[l], o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
    return 0;
}
