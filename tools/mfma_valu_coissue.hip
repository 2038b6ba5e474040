// Microbenchmark: do v_mfma_f64_16x16x4_f64 and v_fma_f64 overlap on gfx950?
//  mode 0: MFMA only    mode 1: FMA only    mode 2: both interleaved in every wave
//  mode 3: even waves MFMA, odd waves FMA (separate waves on the same SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
  d4 acc[4];
  double f[16];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) f[i] = i + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_fma = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (do_fma) {
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = fma(f[i], a, b);
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(double* out, int wpc, const char* name) {
  const int iters = 20000, blocks = 256 * wpc / 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  double mf = 0, ff = 0;
  if (MODE == 0 || MODE == 2) mf = waves * iters * 4 * 2048.0;
  if (MODE == 3) mf = waves / 2 * iters * 4 * 2048.0;
  if (MODE == 1 || MODE == 2) ff = waves * 64 * iters * 16 * 2.0;
  if (MODE == 3) ff = waves / 2 * 64 * iters * 16 * 2.0;
  printf("%-28s waves/CU=%2d  %.3f ms  mfma %.1f TF/s + fma %.1f TF/s = %.1f TF/s\n", name, wpc, ms,
         mf / ms / 1e9, ff / ms / 1e9, (mf + ff) / ms / 1e9);
}
int main() {
  double* out; (void)hipMalloc(&out, sizeof(double) * 256 * 8192);
  for (int wpc : {8, 16}) {
    run<0>(out, wpc, "mfma only");
    run<1>(out, wpc, "fma only");
    run<2>(out, wpc, "both in every wave");
    run<3>(out, wpc, "mfma waves + fma waves");
  }
  return 0;
}
