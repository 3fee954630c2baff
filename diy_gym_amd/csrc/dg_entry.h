// dg_entry.h -- the __global__ entry points of the step path (templates on the envs-per-wavefront mode).
// Instantiated in dg_inst.hip, one translation unit per (mode, part) so that the library builds in parallel.
#pragma once
#include "dg_solver.h"
#include "dg_render.h"

namespace dg {

template <int LANES, bool PROF>
__global__ __launch_bounds__(64) void step_kernel(DevScene sc, MotorTable mt, float* state, const float* actions, uint64_t mask,
                                                   float* obs, float* rew, uint8_t* term, float* rew_sum, uint8_t* term_flag, int32_t* diag,
                                                   unsigned long long* cycles, float* gws) {
  extern __shared__ float smem[];
  constexpr int ACTIVE = envs_per_wave(LANES);
  // 16 / 32 envs per wavefront: the spare lanes stay alive and join the dense Gauss-Seidel sweeps (pgs_dense_sliced)
  constexpr bool SLICED = LANES == 32 || LANES == 16 || LANES == 8 || LANES == 4 || LANES == 1 || LANES == -16;
  const int lane = threadIdx.x; if (!SLICED && lane >= ACTIVE) return;
  const bool primary = lane < ACTIVE;
  const int env = blockIdx.x * ACTIVE + lane; const bool valid = primary && env < sc.num_envs; const int e = env < sc.num_envs ? env : sc.num_envs - 1;
  Lane<LANES> ln(sc, mt, workspace_of<LANES>(sc, smem, gws, lane), state + e, e, valid);
  Prof<PROF> prof; prof.start();
  if (primary) {
    for (int b = 0; b < sc.nba; b++) ln.kinematics(b);
    prof.stamp(PS_KIN);
    if (actions) run_update_ops(ln, actions + (size_t)e * sc.act_dim, mask, -1, -1, diag);
    ln.Sset(DG_ST_STEP, ln.S(DG_ST_STEP) + 1.0f);
    prof.stamp(PS_UPDATE);
  }
  sim_step<LANES, PROF, false, SLICED, LANES == 64 || LANES == 0>(ln, diag, prof, smem, gws);
  if (!primary) return;
  for (int b = 0; b < sc.nba; b++) ln.kinematics(b);
  prof.stamp(PS_KIN);
  run_output_ops(ln, (valid && obs) ? obs + (size_t)e * sc.obs_dim : nullptr, (valid && rew) ? rew + (size_t)e * sc.rew_dim : nullptr,
                 (valid && term) ? term + (size_t)e * sc.term_dim : nullptr, (valid && rew_sum) ? rew_sum + e : nullptr,
                 (valid && term_flag) ? term_flag + e : nullptr);
  prof.stamp(PS_OUTPUT);
  if constexpr (PROF) { if (lane == 0) for (int k = 0; k < PS_COUNT; k++) cycles[(size_t)blockIdx.x * PS_ROW + k] = prof.acc[k]; }
}

// First substep's dynamics of the moving bodies number `parity`, `parity` + 2, ... (two wavefronts share them while the
// other two run the update ops; sc.coll_wave guarantees every moving body is a register-resident chain).
template <int LANES>
DGD void early_dynamics(const Lane<LANES>& ln, int parity) {
  const DevScene& sc = ln.sc; Prof<false> none; int m = 0;
  for (int b = 0; b < sc.nba; b++) {
    if (ln.fixed(b) && ln.bi(b)[DG_BI_N_LINKS] == 0) continue;
    if ((m++ & 1) != parity) continue;
    if (sc.substeps == 1) save_prev_velocities(ln, b);  // the early substep is the step's last one
    ln.template dynamics_chain<6>(b, none);
    const int dvo = ln.plb(b)[PLB_DV], nv = ln.plb(b)[PLB_NV]; for (int k = 0; k < nv; k++) ln.L(dvo + k) = 0.f;
  }
}

// Four wavefronts per workgroup, same 64 envs, same LDS workspace (wave 3 takes half of the narrow phase in substeps
// where it is on the critical path): wave 1 (the helper) runs the inverse kinematics
// and the register-resident dynamics of sc.helper_body, wave 2 the narrow phase (when sc.coll_wave), wave 0 everything else.  Every global / LDS
// hand-off between the two is separated by a __syncthreads (workgroup-scope release / acquire).
// Diagnostic build only: wavefronts 1..3 record when they reach the workgroup's hand-over points (shader cycles since
// their own start) behind the main wave's section stamps: cycles[workgroup][PS_COUNT + 4 (wave - 1) + {B0, B0', B4, end}].
#define DG_WAVE_STAMP(k) do { if constexpr (PROF) { if (lane == 0) cycles[(size_t)blockIdx.x * PS_ROW + PS_COUNT + 4 * (wave - 1) + (k)] = __builtin_amdgcn_s_memtime() - t_start; } } while (0)
template <bool PROF>
__global__ __launch_bounds__(256) void step_kernel_par(DevScene sc, MotorTable mt, float* state, const float* actions, uint64_t mask,
                                                        float* obs, float* rew, uint8_t* term, float* rew_sum, uint8_t* term_flag, int32_t* diag,
                                                        unsigned long long* cycles) {
  constexpr const uint8_t* reset_mask = nullptr;
#define DG_PAR_RESET 0
#include "dg_step_par_body.inc"
#undef DG_PAR_RESET
}
// dg_world_reset of a four-wavefront scene with one hot-start step: the same body in reset mode, under its own name so that
// profiles keep the step's launches and the (mostly empty) reset launches apart
template <int HOT_START_STEPS>  // (a template so that the header can be included by every translation unit; only <1> exists)
__global__ __launch_bounds__(256) void reset_kernel_par(DevScene sc, MotorTable mt, float* state, const uint8_t* reset_mask, float* obs) {
  constexpr bool PROF = HOT_START_STEPS < 0;  // false (dependent on the template parameter so that the stamped branches are discarded)
  const float* actions = nullptr; const uint64_t mask = 0ull; float* rew = nullptr; uint8_t* term = nullptr; float* rew_sum = nullptr; uint8_t* term_flag = nullptr;
  int32_t* diag = nullptr; unsigned long long* cycles = nullptr;
#define DG_PAR_RESET 1
#include "dg_step_par_body.inc"
#undef DG_PAR_RESET
}

template <int LANES>
__global__ __launch_bounds__(64) void reset_kernel(DevScene sc, MotorTable mt, float* state, const uint8_t* mask, float* obs, float* gws) {
  extern __shared__ float smem[];
  constexpr int ACTIVE = envs_per_wave(LANES);
  // LDS modes with fewer than 64 envs per wavefront: the hot-start steps run the lane-sliced step (narrow phase, row
  // construction and sweeps shared by the lanes of an env's group) exactly as step_kernel does -- one lane per env
  // through the generic solver cost from_the_readme 16 ms per masked reset against 3.4 ms per step.  Envs of the
  // wavefront that are NOT being reset take part with `valid` off: they compute in their (scratch) workspace and store
  // nothing.
  constexpr bool SLICED = LANES == 32 || LANES == 16 || LANES == 8 || LANES == 4 || LANES == 1;
  const int lane = threadIdx.x; if (!SLICED && lane >= ACTIVE) return;
  const bool primary = lane < ACTIVE;
  const int env = blockIdx.x * ACTIVE + lane; const bool valid = primary && env < sc.num_envs; const int e = env < sc.num_envs ? env : sc.num_envs - 1;
  const bool doit = valid && (mask == nullptr || mask[e] != 0);
  Lane<LANES> ln(sc, mt, workspace_of<LANES>(sc, smem, gws, lane), state + e, e, SLICED ? doit : valid);
  if constexpr (SLICED) {
    if (sc.no_sliced_reset) {  // DG_NO_SLICED_RESET: one lane per env through the generic solver (ablation / tests)
      if (!primary) return;
      if (doit) { ln.Sset(DG_ST_STEP, 0.0f); run_reset_ops(ln); Prof<false> prof; for (int k = 0; k < sc.hot_start; k++) sim_step(ln, nullptr, prof); }
    } else {
    if (doit) { ln.Sset(DG_ST_STEP, 0.0f); run_reset_ops(ln); }
    if (__any(doit)) { Prof<false> prof; for (int k = 0; k < sc.hot_start; k++) sim_step<LANES, false, false, true, false>(ln, nullptr, prof, smem, gws); }
    if (!primary) return;
    }
  } else if (doit) {
    ln.Sset(DG_ST_STEP, 0.0f);
    run_reset_ops(ln);
    Prof<false> prof;
    for (int k = 0; k < sc.hot_start; k++) sim_step(ln, nullptr, prof);
  }
  // observations of the envs that were reset; a wavefront without one has nothing to refresh (its rows are current)
  if (obs && (mask == nullptr || __any(doit))) {
    for (int b = 0; b < sc.nba; b++) ln.kinematics(b);
    run_output_ops(ln, valid ? obs + (size_t)e * sc.obs_dim : nullptr, nullptr, nullptr, nullptr, nullptr, OUT_ALL, -1,
                   doit ? (sc.hot_start > 0 ? 0 : 1) : 2);
  }
}

template <int LANES>
__global__ __launch_bounds__(64) void observe_kernel(DevScene sc, MotorTable mt, float* state, float* obs, float* rew, uint8_t* term,
                                                      float* rew_sum, uint8_t* term_flag, float* gws) {
  extern __shared__ float smem[];
  constexpr int ACTIVE = envs_per_wave(LANES);
  const int lane = threadIdx.x; if (lane >= ACTIVE) return;
  const int env = blockIdx.x * ACTIVE + lane; const bool valid = env < sc.num_envs; const int e = valid ? env : sc.num_envs - 1;
  Lane<LANES> ln(sc, mt, workspace_of<LANES>(sc, smem, gws, lane), state + e, e, false);  // never stores state
  for (int b = 0; b < sc.nba; b++) ln.kinematics(b);
  run_output_ops(ln, (valid && obs) ? obs + (size_t)e * sc.obs_dim : nullptr, (valid && rew) ? rew + (size_t)e * sc.rew_dim : nullptr,
                 (valid && term) ? term + (size_t)e * sc.term_dim : nullptr, (valid && rew_sum) ? rew_sum + e : nullptr,
                 (valid && term_flag) ? term_flag + e : nullptr, OUT_ALL, -1, 1 /* no contact list outside a step */);
}

template <int LANES>
__global__ __launch_bounds__(64) void frame_kernel(DevScene sc, MotorTable mt, float* state, int body, int frame, int com, float* out, float* gws) {
  extern __shared__ float smem[];
  constexpr int ACTIVE = envs_per_wave(LANES);
  const int lane = threadIdx.x; if (lane >= ACTIVE) return;
  const int env = blockIdx.x * ACTIVE + lane; if (env >= sc.num_envs) return;
  Lane<LANES> ln(sc, mt, workspace_of<LANES>(sc, smem, gws, lane), state + env, env, false);
  ln.kinematics(body);
  V3 p, v, w; Q4 q; ln.frame_state(body, frame, com != 0, p, q, v, w, true);
  float* o = out + (size_t)env * 13;
  o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w; o[7] = v.x; o[8] = v.y; o[9] = v.z; o[10] = w.x; o[11] = w.y; o[12] = w.z;
}

// dg_world_apply_wrench: p.applyExternalForce / p.applyExternalTorque for user addons written in Python (reference
// examples/drone_pilot/drone_pilot.py:34-37, diy_gym/addons/controllers/external_force.py:24), every env at once;
// force / pos / torque are [num_envs][3] or null (= zero).  Consumed by the next dg_world_step.
template <int LANES>
__global__ __launch_bounds__(64) void wrench_kernel(DevScene sc, MotorTable mt, float* state, int body, int frame, int link_frame,
                                                     const float* force, const float* pos, const float* torque, float* gws) {
  extern __shared__ float smem[];
  constexpr int ACTIVE = envs_per_wave(LANES);
  const int lane = threadIdx.x; if (lane >= ACTIVE) return;
  const int env = blockIdx.x * ACTIVE + lane; if (env >= sc.num_envs) return;
  Lane<LANES> ln(sc, mt, workspace_of<LANES>(sc, smem, gws, lane), state + env, env, true);
  ln.kinematics(body);
  auto row = [&](const float* p) { return p ? v3(p[3 * (size_t)env], p[3 * (size_t)env + 1], p[3 * (size_t)env + 2]) : v3(0.f, 0.f, 0.f); };
  apply_frame_wrench(ln, body, frame, row(force), row(pos), row(torque), link_frame != 0);
}

}  // namespace dg
