// dev probe: issue rate of v_mfma_f32_32x32x16_bf16 chains on ONE accumulator vs round-robin over 2 / 4 accumulators (one wave per SIMD, 4 waves per CU)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(1.0f + j * 0.01f); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % NACC], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int WAVES>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 64 * WAVES * 4 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 2000;
    k<NACC, WAVES><<<256, 64 * WAVES>>>(out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NACC, WAVES><<<256, 64 * WAVES>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 24;
    printf("%-28s waves/CU %d  acc %d: %.1f ns per MFMA per wave, %.2f PF/s chip (%.0f readcyclecounter ticks per MFMA)\n", name, WAVES, NACC, ms * 1e6 / n,
           256.0 * WAVES * n * 32768.0 / (ms * 1e-3) / 1e15, (double)c / n);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 4>("same accumulator"); run<2, 4>("2 accumulators"); run<4, 4>("4 accumulators");
    run<1, 8>("same accumulator"); run<2, 8>("2 accumulators"); run<4, 8>("4 accumulators");
    return 0;
}
