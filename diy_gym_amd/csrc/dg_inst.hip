// dg_inst.hip -- instantiates the kernels of one envs-per-wavefront mode.  Compiled several times:
//   -DDG_LANES={64,32,16,8,4,1,0}  -DDG_PART=0  step kernels (+ stamped build for 64, 16 and 8)
//                            -DDG_PART=1  reset / observe / frame / pose kernels and the mode's launch table
//   -DDG_LANES=64            -DDG_PART=2  helper-wave step kernels
//   -DDG_LANES=-16 -DDG_TAG=g16           the global-workspace mode with 16 envs per wavefront
#include <hip/hip_runtime.h>
#include "dg_launch.h"
#include "dg_entry.h"

#define DG_CAT_(a, b) a##b
#define DG_CAT(a, b) DG_CAT_(a, b)
#ifndef DG_TAG
#define DG_TAG DG_LANES
#endif
#define DGL(name) DG_CAT(DG_CAT(name, _), DG_TAG)

namespace dg {

constexpr int L = DG_LANES;
constexpr bool HAS_PROF = (DG_LANES == 64 || DG_LANES == 32 || DG_LANES == 16 || DG_LANES == 8 || DG_LANES == 4 || DG_LANES == 1 || DG_LANES == -16);

void DGL(l_step)(dim3 grid, int lds, hipStream_t st, bool prof, DG_STEP_PARAMS, float* gws);
hipError_t DGL(l_prepare_step)(int lds);
#if DG_LANES == 64
void l_step_par_64(dim3 grid, int lds, hipStream_t st, bool prof, DG_STEP_PARAMS, const uint8_t* reset_mask, int reset_mode);
hipError_t l_prepare_par_64(int lds);
#endif

#if DG_PART == 0
void DGL(l_step)(dim3 grid, int lds, hipStream_t st, bool prof, DG_STEP_PARAMS, float* gws) {
  if constexpr (HAS_PROF) { if (prof) { hipLaunchKernelGGL((step_kernel<L, true>), grid, dim3(64), lds, st, sc, mt, state, actions, mask, obs, rew, term, rew_sum, term_flag, diag, cycles, gws); return; } }
  hipLaunchKernelGGL((step_kernel<L, false>), grid, dim3(64), lds, st, sc, mt, state, actions, mask, obs, rew, term, rew_sum, term_flag, diag, (unsigned long long*)nullptr, gws);
}
hipError_t DGL(l_prepare_step)(int lds) {
  hipError_t e = hipFuncSetAttribute((const void*)step_kernel<L, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if constexpr (HAS_PROF) { if (e == hipSuccess) e = hipFuncSetAttribute((const void*)step_kernel<L, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
  return e;
}
#elif DG_PART == 2
void l_step_par_64(dim3 grid, int lds, hipStream_t st, bool prof, DG_STEP_PARAMS, const uint8_t* reset_mask, int reset_mode) {
  if (reset_mode) hipLaunchKernelGGL(reset_kernel_par<1>, grid, dim3(256), lds, st, sc, mt, state, reset_mask, obs);
  else if (prof) hipLaunchKernelGGL(step_kernel_par<true>, grid, dim3(256), lds, st, sc, mt, state, actions, mask, obs, rew, term, rew_sum, term_flag, diag, cycles);
  else hipLaunchKernelGGL(step_kernel_par<false>, grid, dim3(256), lds, st, sc, mt, state, actions, mask, obs, rew, term, rew_sum, term_flag, diag, (unsigned long long*)nullptr);
}
hipError_t l_prepare_par_64(int lds) {
  hipError_t e = hipFuncSetAttribute((const void*)step_kernel_par<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)step_kernel_par<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)reset_kernel_par<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  return e;
}
#else  // DG_PART == 1
static void l_reset(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, const uint8_t* mask, float* obs, float* gws) {
  hipLaunchKernelGGL(reset_kernel<L>, grid, dim3(64), lds, st, sc, mt, state, mask, obs, gws);
}
static void l_observe(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, float* obs, float* rew, uint8_t* term, float* rew_sum, uint8_t* term_flag, float* gws) {
  hipLaunchKernelGGL(observe_kernel<L>, grid, dim3(64), lds, st, sc, mt, state, obs, rew, term, rew_sum, term_flag, gws);
}
static void l_frame(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, int body, int frame, int com, float* out, float* gws) {
  hipLaunchKernelGGL(frame_kernel<L>, grid, dim3(64), lds, st, sc, mt, state, body, frame, com, out, gws);
}
static void l_wrench(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, int body, int frame, int link_frame, const float* force, const float* pos, const float* torque, float* gws) {
  hipLaunchKernelGGL(wrench_kernel<L>, grid, dim3(64), lds, st, sc, mt, state, body, frame, link_frame, force, pos, torque, gws);
}
static void l_pose(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, int ncam, cip CI, cfp CF, float* table, float* gws) {
  hipLaunchKernelGGL(pose_kernel<L>, grid, dim3(64), lds, st, sc, mt, state, ncam, CI, CF, table, gws);
}
static hipError_t l_prepare(int lds) {
  if (L == 0) return hipSuccess;
  if (L < 0) {  // only the step kernel uses LDS (the sliced sweeps' accumulated impulses)
    return DGL(l_prepare_step)(lds);
  }
  hipError_t e = DGL(l_prepare_step)(lds);
#define DG_ATTR(K) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, lds)
  DG_ATTR(reset_kernel<L>); DG_ATTR(observe_kernel<L>); DG_ATTR(frame_kernel<L>); DG_ATTR(wrench_kernel<L>); DG_ATTR(pose_kernel<L>);
#undef DG_ATTR
#if DG_LANES == 64
  if (e == hipSuccess) e = l_prepare_par_64(lds);
#endif
  return e;
}
#ifndef __HIP_DEVICE_COMPILE__  // a host-side table of host function pointers
extern const LaunchTable DGL(g_launch_table) = {
    HAS_PROF, l_prepare, DGL(l_step),
#if DG_LANES == 64
    l_step_par_64,
#else
    nullptr,
#endif
    l_reset, l_observe, l_frame, l_wrench, l_pose};
#endif
#endif

}  // namespace dg
