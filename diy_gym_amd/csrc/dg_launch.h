// dg_launch.h -- host-side launch table: one set of wrappers per envs-per-wavefront mode, each defined in its own
// translation unit (dg_inst.hip compiled with -DDG_LANES=... -DDG_PART=...), so the library builds in parallel.
#pragma once
#include "dg_kernels.h"

namespace dg {

#define DG_STEP_PARAMS DevScene sc, MotorTable mt, float* state, const float* actions, uint64_t mask, float* obs, float* rew, uint8_t* term, \
                       float* rew_sum, uint8_t* term_flag, int32_t* diag, unsigned long long* cycles
struct LaunchTable {
  bool has_prof;
  hipError_t (*prepare)(int lds_bytes);
  void (*step)(dim3 grid, int lds, hipStream_t st, bool prof, DG_STEP_PARAMS, float* gws);
  void (*step_par)(dim3 grid, int lds, hipStream_t st, bool prof, DG_STEP_PARAMS, const uint8_t* reset_mask, int reset_mode);  // reset_mode 1: masked reset + one hot-start step (mask NULL = every env)
  void (*reset)(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, const uint8_t* mask, float* obs, float* gws);
  void (*observe)(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, float* obs, float* rew, uint8_t* term, float* rew_sum, uint8_t* term_flag, float* gws);
  void (*frame)(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, int body, int frame, int com, float* out, float* gws);
  void (*wrench)(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, int body, int frame, int link_frame, const float* force, const float* pos, const float* torque, float* gws);
  void (*pose)(dim3 grid, int lds, hipStream_t st, DevScene sc, MotorTable mt, float* state, int ncam, cip CI, cfp CF, float* table, float* gws);
};
const LaunchTable& launch_table(int lanes);  // lanes in {64, 32, 16, 8, 4, 0, -16}

}  // namespace dg
