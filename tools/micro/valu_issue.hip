// Microbenchmark: cycles per VALU instruction for ONE wavefront per SIMD on gfx950, as a function of the
// instruction-level parallelism of the stream (dependent chain vs K independent chains) -- the regime the
// step kernel's main wave lives in.  hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int K, int TRANS>
__global__ __launch_bounds__(256) void chains(float* out, unsigned long long* cyc, int iters, float a, float b) {
  float x[K];
#pragma unroll
  for (int k = 0; k < K; k++) x[k] = out[threadIdx.x + k];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / K; r++)
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (TRANS) x[k] = __builtin_amdgcn_rcpf(x[k]) + a; else x[k] = fmaf(x[k], a, b);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; k++) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int K, int TRANS>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 2000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * 256 + 4096); hipMemset(out, 0, sizeof(float) * blocks * 256 + 4096);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
  hipLaunchKernelGGL((chains<K, TRANS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.999f, 0.001f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 4); hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  const double per = s / h.size() / ((double)iters * (64 / K) * K * (TRANS ? 2 : 1));
  printf("%-28s chains=%2d waves/SIMD=%d: %.2f cycles per instruction\n", name, K, waves_per_simd, per);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w = 1; w <= 2; w++) {
    run<1, 0>("v_fma dependent", w); run<2, 0>("v_fma", w); run<4, 0>("v_fma", w); run<8, 0>("v_fma", w); run<16, 0>("v_fma", w);
    run<1, 1>("v_rcp+v_add dependent", w); run<4, 1>("v_rcp+v_add", w);
  }
  return 0;
}
