// Microbenchmark 4: issue rate of PACKED fp32 FMAs (v_pk_fma_f32: two fp32 FMAs per lane per instruction) against scalar FMAs,
// 3-VGPR-operand form, 8 independent chains, one and two wavefronts per SIMD.  If a lone wavefront issues a v_pk_fma_f32 at the
// cadence of a v_fma_f32, hand-packed 3x3 / 6x6 algebra halves the length of the instruction streams that bound step_kernel_par.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
template <int K, bool PK>
__global__ __launch_bounds__(512) void fma3(float* out, unsigned long long* cyc, int iters) {
  v2 x[K], y[K], z[K];
#pragma unroll
  for (int k = 0; k < K; k++) { x[k] = v2{out[threadIdx.x + k], out[threadIdx.x + k + 1]}; y[k] = v2{out[threadIdx.x + 64 + k] + 0.999f, out[threadIdx.x + 65 + k] + 0.998f}; z[k] = v2{out[threadIdx.x + 128 + k] + 0.001f, out[threadIdx.x + 129 + k] + 0.002f}; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / K; r++)
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (PK) x[k] = __builtin_elementwise_fma(x[k], y[k], z[k]);
        else x[k].x = fmaf(x[k].x, y[k].x, z[k].x);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; k++) s += x[k].x + x[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <bool PK>
static double run(int threads) {
  const int blocks = 256, iters = 2000;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * blocks * 512 + 4096); (void)hipMemset(out, 0, sizeof(float) * blocks * 512 + 4096);
  (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 8); (void)hipMemset(cyc, 0, sizeof(unsigned long long) * blocks * 8);
  hipLaunchKernelGGL((fma3<8, PK>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8); (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; int n = 0; for (auto v : h) if (v) { s += (double)v; n++; }
  (void)hipFree(out); (void)hipFree(cyc);
  return s / n / (iters * 64.0);
}
int main() {
  for (int threads : {256, 512}) {
    printf("%d wavefront(s) per SIMD: v_fma_f32 %.2f, v_pk_fma_f32 %.2f cycles per instruction per wavefront\n", threads / 256, run<false>(threads), run<true>(threads));
  }
  return 0;
}
