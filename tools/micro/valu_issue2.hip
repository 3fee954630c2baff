// Microbenchmark 2: one wavefront per SIMD, realistic operand patterns -- three-VGPR FMAs and chained 3x3 matrix
// products (the shape of the IK's forward kinematics).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int K>
__global__ __launch_bounds__(256) void fma3(float* out, unsigned long long* cyc, int iters) {
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
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(256) void mat3chain(float* out, unsigned long long* cyc, int iters) {
  float R[9], C[9];
#pragma unroll
  for (int k = 0; k < 9; k++) { R[k] = out[threadIdx.x + k] + (k % 4 == 0 ? 1.f : 0.f); C[k] = out[threadIdx.x + 64 + k] + (k % 4 == 0 ? 1.f : 0.01f); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
      float N[9];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) N[3 * i + j] = R[3 * i] * C[j] + R[3 * i + 1] * C[3 + j] + R[3 * i + 2] * C[6 + j];
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = N[k];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 9; k++) s += R[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
static double run(void (*launch)(float*, unsigned long long*, int), double ops_per_iter) {
  const int blocks = 256, iters = 2000;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * blocks * 256 + 4096); (void)hipMemset(out, 0, sizeof(float) * blocks * 256 + 4096);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
  launch(out, cyc, iters); (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4); (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  (void)hipFree(out); (void)hipFree(cyc);
  return s / h.size() / (iters * ops_per_iter);
}
int main() {
  printf("v_fma 3 VGPR operands, 1 chain : %.2f cycles per instruction\n", run([](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL(fma3<1>, dim3(256), dim3(256), 0, 0, o, c, n); }, 64));
  printf("v_fma 3 VGPR operands, 4 chains: %.2f cycles per instruction\n", run([](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL(fma3<4>, dim3(256), dim3(256), 0, 0, o, c, n); }, 64));
  printf("v_fma 3 VGPR operands, 8 chains: %.2f cycles per instruction\n", run([](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL(fma3<8>, dim3(256), dim3(256), 0, 0, o, c, n); }, 64));
  printf("chained 3x3 products (27 mul/fma each, + copies): %.2f cycles per product\n", run([](float* o, unsigned long long* c, int n) { hipLaunchKernelGGL(mat3chain, dim3(256), dim3(256), 0, 0, o, c, n); }, 4));
  return 0;
}
