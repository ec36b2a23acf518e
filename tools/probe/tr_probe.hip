// dev probe: semantics of ds_read_b64_tr_b16 (gfx950) for a row-strided [k][col] bf16 image
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(s4* out, int stride) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    // group g: column block (g&1)*16, k rows 8*(g>>1) + (i>>2); lane i supplies row i/4, cols 4*(i%4)
    const unsigned short* p = lds + (8 * (g >> 1) + (i >> 2)) * stride + (g & 1) * 16 + (i & 3) * 4;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    out[l] = v;
}
int main() {
    s4* d; hipMalloc(&d, 64 * sizeof(s4));
    for (int stride : {32, 64}) {
        k<<<1, 64>>>(d, stride);
        s4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            int g = l >> 4, i = l & 15;
            for (int j = 0; j < 4; ++j) {
                int expect = (8 * (g >> 1) + j) * stride + (g & 1) * 16 + i;      // row j of the block, column i
                if ((unsigned short)h[l][j] != expect) { if (bad < 8) printf("stride %d lane %d j %d got %d expect %d\n", stride, l, j, (unsigned short)h[l][j], expect); ++bad; }
            }
        }
        printf("stride %d: %s (%d mismatches)\n", stride, bad ? "MISMATCH" : "OK: lane (g,i) gets rows k0..k0+3 of column i", bad);
    }
    return 0;
}
