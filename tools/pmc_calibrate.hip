// pmc_calibrate.hip -- what FETCH_SIZE / WRITE_SIZE report for the access shapes this engine uses
// (MI355X_MICROARCH.md, HBM: 16 B/lane streaming reads are tallied at 1/2; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel reads a
// 1-GiB buffer exactly once; run under  rocprofv3 --pmc FETCH_SIZE --kernel-trace  and compare.
//   cal_read16    : 16 B per lane, 1 KiB contiguous per wave      (stencil, BLAS-1, k_ell)
//   cal_read8     : 8 B per lane, 512 B contiguous per wave
//   cal_read8_seg : 8 B per lane, four 128-B segments in four rows (X operand of k_bsr_mfma)
//   cal_write8_seg: 8 B per lane stores in the same shape          (Y of k_bsr_mfma)
// build: hipcc -O3 --offload-arch=gfx950 -o pmc_calibrate pmc_calibrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_BYTES (1ull << 30)

__global__ void cal_read16(const double2* __restrict__ p, double* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (; i < n; i += stride) { double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 12345.678) out[0] = acc;
}
__global__ void cal_read8(const double* __restrict__ p, double* out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (; i < n; i += stride) acc += p[i];
  if (acc == 12345.678) out[0] = acc;
}
// rows of `ld` doubles; a wave reads rows r..r+3, 16 doubles (128 B) at column c0 of each
__global__ void cal_read8_seg(const double* __restrict__ p, double* out, size_t nrows, int ld) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int segs = ld / 16;
  double acc = 0.0;
  for (size_t item = wave; item < (nrows / 4) * segs; item += nwaves) {
    const size_t rg = item / segs, sg = item % segs;
    acc += p[(rg * 4 + (lane >> 4)) * ld + sg * 16 + (lane & 15)];
  }
  if (acc == 12345.678) out[0] = acc;
}
__global__ void cal_write8_seg(double* __restrict__ p, size_t nrows, int ld) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int segs = ld / 16;
  for (size_t item = wave; item < (nrows / 4) * segs; item += nwaves) {
    const size_t rg = item / segs, sg = item % segs;
    p[(rg * 4 + (lane >> 4)) * ld + sg * 16 + (lane & 15)] = (double)lane;
  }
}

int main() {
  void *buf, *out;
  if (hipMalloc(&buf, N_BYTES) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  hipMemset(buf, 1, N_BYTES);
  const int ld = 512;   // doubles per row (= 2 * nbp of a 256-probe batch)
  const size_t nrows = N_BYTES / 8 / ld;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(cal_read16, dim3(4096), dim3(256), 0, 0, (const double2*)buf, (double*)out, N_BYTES / 16);
    hipLaunchKernelGGL(cal_read8, dim3(4096), dim3(256), 0, 0, (const double*)buf, (double*)out, N_BYTES / 8);
    hipLaunchKernelGGL(cal_read8_seg, dim3(4096), dim3(256), 0, 0, (const double*)buf, (double*)out, nrows, ld);
    hipLaunchKernelGGL(cal_write8_seg, dim3(4096), dim3(256), 0, 0, (double*)buf, nrows, ld);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  std::printf("pmc_calibrate: each kernel touched %llu bytes once per launch\n", (unsigned long long)N_BYTES);
  return 0;
}
