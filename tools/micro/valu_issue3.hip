// Microbenchmark 3: does a wavefront whose upper 32 lanes are inactive issue faster?  (3-VGPR-operand FMAs, 8 chains,
// one and two wavefronts per SIMD.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int K>
__global__ __launch_bounds__(512) void fma3(float* out, unsigned long long* cyc, int iters, int active_lanes) {
  if ((int)(threadIdx.x & 63) >= active_lanes) return;
  float x[K], y[K], z[K];
#pragma unroll
  for (int k = 0; k < K; k++) { x[k] = out[threadIdx.x + k]; y[k] = out[threadIdx.x + 64 + k] + 0.999f; z[k] = out[threadIdx.x + 128 + k] + 0.001f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / K; r++)
#pragma unroll
      for (int k = 0; k < K; k++) x[k] = fmaf(x[k], y[k], z[k]);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; k++) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
static double run(int threads, int active) {
  const int blocks = 256, iters = 2000;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * blocks * 512 + 4096); (void)hipMemset(out, 0, sizeof(float) * blocks * 512 + 4096);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 8); (void)hipMemset(cyc, 0, sizeof(unsigned long long) * blocks * 8);
  hipLaunchKernelGGL(fma3<8>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, active); (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8); (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; int n = 0; for (auto v : h) if (v) { s += (double)v; n++; }
  (void)hipFree(out); (void)hipFree(cyc);
  return s / n / (iters * 64.0);
}
int main() {
  for (int threads : {256, 512})
    for (int active : {64, 32, 16})
      printf("%d wavefront(s) per SIMD, %2d active lanes: %.2f cycles per instruction per wavefront\n", threads / 256, active, run(threads, active));
  return 0;
}
