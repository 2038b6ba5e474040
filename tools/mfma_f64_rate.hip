// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950 (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a, double b) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a, double b) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F>
double timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; hipMalloc(&out, sizeof(double) * 256 * 4096);
  const int iters = 20000;
  for (int wpc : {4, 8, 16}) {   // waves per CU
    int blocks = 256 * wpc / 4;
    double ms = timeit([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
    double flops = (double)blocks * 4 * iters * 8 * 2048.0;
    printf("mfma_f64_16x16x4 waves/CU=%2d: %.3f ms  %.1f TFLOP/s  cycles/MFMA/SIMD@2.4GHz=%.1f\n", wpc, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wpc / 4));
    ms = timeit([&] { hipLaunchKernelGGL(k_fma<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    flops = (double)blocks * 256 * iters * 16 * 2.0;
    printf("v_fma_f64        waves/CU=%2d: %.3f ms  %.1f TFLOP/s\n", wpc, ms, flops / ms / 1e9);
  }
  return 0;
}
