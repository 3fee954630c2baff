/* diygym_hip.h -- C-ABI of the MI355X batched simulation backend.
 *
 * This is the boundary that replaces the `pybullet` module API on DIYGym's
 * step path.  The reference is Python and binds pybullet through its CPython
 * extension; the calls it makes are enumerated in SURVEY.md 8(b).  Each entry
 * point below names the reference call sites it replaces.  All buffer
 * arguments are DEVICE pointers owned by the caller (torch tensors'
 * data_ptr()); `stream` is a hipStream_t (NULL = default stream).  No entry
 * point allocates, frees or synchronises after dg_world_create, so every call
 * can be captured into a hipGraph.  Return value: 0 on success, a negative
 * DG_ERR_* otherwise, with dg_last_error() giving the message.  One world per
 * GPU, one host thread per world (same rule as a pybullet client).
 *
 * Batched buffers:
 *   state    float [state_dim][env_stride]  struct-of-arrays over envs (coalesced)
 *   actions  float [num_envs][act_dim]      row per env, columns in the order of
 *                                           flatten(action_space) (reference utils.py:46-60)
 *   obs      float [num_envs][obs_dim]      ... of flatten(observe())
 *   rew      float [num_envs][rew_dim]      one column per reward addon
 *   term     uint8 [num_envs][term_dim]     one column per terminal addon
 *   rew_sum  float [num_envs]               collapsed reward  (sum_rewards, diy_gym.py:94,168)
 *   term_flag uint8 [num_envs]              collapsed terminal (terminal_if_any/all, diy_gym.py:95-96,185)
 */
#ifndef DIYGYM_HIP_H
#define DIYGYM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DG_OK 0
#define DG_ERR_BAD_SCENE (-1)
#define DG_ERR_HIP (-2)
#define DG_ERR_UNSUPPORTED (-3)
#define DG_ERR_ARG (-4)

typedef struct dg_world dg_world;

/* library identification: (major << 16) | minor */
int32_t dg_version(void);
const char* dg_last_error(void);

/* Replaces p.connect / p.resetSimulation / p.setPhysicsEngineParameter /
 * p.setGravity / p.loadURDF / p.changeDynamics (reference diy_gym.py:68-82,
 * model.py:65-83): builds the device-side constant tables for `num_envs`
 * copies of the scene blob (include/diygym_scene.h) on HIP device `device`.
 * `env_stride` (>= num_envs, multiple of 64) is the row pitch of `state`.
 * `seed` and `env_index_base` key the per-env respawn RNG streams. */
int32_t dg_world_create(const int32_t* idata, int64_t n_i, const double* fdata, int64_t n_f, int32_t num_envs,
                        int32_t env_stride, int32_t device, uint64_t seed, int64_t env_index_base, dg_world** out);

/* Replaces p.disconnect (reference diy_gym.py:225). */
void dg_world_destroy(dg_world* w);

/* dims[0..7] = state_dim, act_dim, obs_dim, rew_dim, term_dim, n_links,
 * lds_bytes_per_workgroup, workspace mode.  Mode: 64 / 32 / 16 / 8 / 4 = that many envs per wavefront with the per-env
 * scratch in LDS; 0 = 64 envs per wavefront, scratch in a device buffer the world owns; -16 = 16 envs per
 * wavefront, scratch in that buffer (chosen by dg_world_create from the scene's scratch footprint). */
int32_t dg_world_dims(const dg_world* w, int32_t dims[8]);

/* Name of the step kernel dg_world_step launches for this world (for benchmark / profile labels):
 * "step_kernel<lanes>" or "step_kernel_par (...)".  The pointer is valid until the calling thread's next call. */
const char* dg_world_kernel_name(const dg_world* w);

/* Host-side view of the motor table (p.setJointMotorControlArray gains/forces,
 * uniform over envs): cfg[n_links][3] = kp, kd, max_force (<0: raw impulse). */
int32_t dg_world_get_motor_cfg(const dg_world* w, double* cfg);
int32_t dg_world_set_motor_cfg(dg_world* w, const double* cfg);

/* Writes the load-time state (model poses from the config, joints at zero)
 * into `state` for every env.  Replaces the p.resetBasePositionAndOrientation
 * done while models are constructed (reference model.py:68). */
int32_t dg_world_init_state(dg_world* w, float* state, void* stream);

/* Replaces DIYGym.reset()'s addon.reset() + hot_start x p.stepSimulation +
 * observe (reference diy_gym.py:130-148; respawn.py:31-35,
 * joint_controller.py:36-38) for the envs whose `mask` byte is non-zero
 * (mask == NULL: all envs).  obs (nullable) is written for the envs that were reset -- with mask == NULL that is every
 * env; with a mask, the rows of the other envs are left as the last step / reset / observe wrote them (in units of
 * one wavefront: rows sharing a wavefront with a reset env are recomputed, to the same values). */
int32_t dg_world_reset(dg_world* w, float* state, const uint8_t* mask, float* obs, void* stream);

/* Replaces one DIYGym.step(): addon.update() for the controller addons whose
 * bit is set in `update_mask` (p.setJointMotorControlArray,
 * p.calculateInverseKinematics, p.applyExternalForce/Torque), step_counter += 1,
 * p.stepSimulation, then observe / reward / is_terminal (p.getJointStates,
 * p.getLinkState, p.getBasePositionAndOrientation, p.getBaseVelocity)
 * (reference diy_gym.py:187-209).  Any output pointer may be NULL. */
int32_t dg_world_step(dg_world* w, float* state, const float* actions, uint64_t update_mask, float* obs, float* rew,
                      uint8_t* term, float* rew_sum, uint8_t* term_flag, void* stream);

/* Outputs for the current state without stepping (reference diy_gym.py:150-185). */
int32_t dg_world_observe(dg_world* w, const float* state, float* obs, float* rew, uint8_t* term, float* rew_sum,
                         uint8_t* term_flag, void* stream);

/* Debug getter replacing p.getLinkState / p.getBasePositionAndOrientation /
 * p.getBaseVelocity: out[num_envs][13] = pos3 quat4 linvel3 angvel3 of `frame`
 * (pybullet joint index, -1 = base) of body `body`; com != 0 selects the
 * inertial frame (items 0,1,6,7), else the URDF link frame (items 4,5). */
int32_t dg_world_frame_state(dg_world* w, const float* state, int32_t body, int32_t frame, int32_t com, float* out,
                             void* stream);

/* Replaces p.applyExternalForce(uid, linkIndex, force, pos, flags) + p.applyExternalTorque(uid, linkIndex, torque, flags)
 * for addons that are NOT compiled into the step kernel -- a user's Python addon acting on the world from its update()
 * hook (reference examples/drone_pilot/drone_pilot.py:34-37, diy_gym/addons/addon.py:80-81 registry, :91-186 hooks;
 * diy_gym/addons/controllers/external_force.py:24) -- every env at once: force, pos, torque are device arrays
 * [num_envs][3] (any may be NULL = zero).  `frame` is the pybullet joint index of the link (-1 = base).  flags as in
 * pybullet: DG_WRENCH_LINK_FRAME -- force / torque along the axes of the link's INERTIAL frame (URDF <inertial> origin:
 * centre of mass and its rpy), pos relative to that origin -- the frame pybullet resolves LINK_FRAME against [R], NOT the
 * joint frame; for a link without an <inertial><origin> the two coincide;
 * DG_WRENCH_WORLD_FRAME -- all three in world coordinates.  The wrench acts during the NEXT dg_world_step only
 * (pybullet clears external forces after every stepSimulation) and adds to what compiled ops apply.  For a frame on a
 * movable link the joints between it and the base receive J^T of the wrench.  The compiled external_force / propellor
 * ops run the same device function, so a Python addon built on this entry reproduces them bit for bit. */
enum { DG_WRENCH_LINK_FRAME = 1, DG_WRENCH_WORLD_FRAME = 2 };
int32_t dg_world_apply_wrench(dg_world* w, float* state, int32_t body, int32_t frame, int32_t flags, const float* force,
                              const float* pos, const float* torque, void* stream);

/* Replaces Camera.observe -> p.computeProjectionMatrixFOV / p.getCameraImage (reference
 * diy_gym/addons/sensors/camera.py:44,58-92) for camera `camera` of the scene, all envs:
 *   rgb   float [num_envs][h*w*3]  flat shaded body colour (NOT a parity output: pybullet renders visual meshes)
 *   depth float [num_envs][h*w]    eye-space z exactly as camera.py:82-85 computes it (negative; -far = nothing hit)
 *   seg   int32 [num_envs][h*w]    uid + ((link + 1) << 24), -1 = background
 * flat pixel index = row * width + col, row 0 at the top; any pointer may be NULL. */
int32_t dg_world_render(dg_world* w, const float* state, int32_t camera, float* rgb, float* depth, int32_t* seg, void* stream);

/* Diagnostic switches of dg_world_render (bit 0: test every shape for every pixel group -- the brute-force picture the
 * culled one must equal bit for bit; 2: no intersections, background only; 4: hulls skipped; 128: no strip / tile level culling;
 * 512: one band per picture whatever the batch size; 16: stage counters of a -DDG_RENDER_COUNTERS build, tools/gpu_cam_bench.py).
 * The environment variables DG_RENDER_NO_CULL / DG_RENDER_DIAG give the initial value, read once at dg_world_create. */
int32_t dg_world_set_render_diag(dg_world* w, int32_t flags);

/* Per-env diagnostics of the last step: diag[num_envs][DG_DIAG_STRIDE] (int32), columns DG_DIAG_*: contact count and
 * Gauss-Seidel iterations of the final substep, the same two of the first substep, and the iterations each of the
 * scene's first DG_DIAG_N_IK inverse-kinematics ops ran for that env.  Optional; pass NULL to disable (default).  The
 * buffer must stay valid until changed. */
enum { DG_DIAG_CONTACTS = 0, DG_DIAG_PGS_ITERS = 1, DG_DIAG_PGS_ITERS_FIRST = 2, DG_DIAG_CONTACTS_FIRST = 3, DG_DIAG_IK_ITERS = 4,
       DG_DIAG_N_IK = 4, DG_DIAG_STRIDE = 8 };
int32_t dg_world_set_diag_buffer(dg_world* w, int32_t* diag);

/* Diagnostic build of the step kernel with in-kernel cycle stamps (s_memtime): when `cycles` is
 * non-NULL, dg_world_step launches the stamped instantiation and lane 0 of every wavefront writes
 * cycles[workgroup][24]: entries 0..11 = shader cycles the (main) wavefront spent in {update ops (IK), kinematics, narrow phase, ABA passes,
 * M^-1 columns, row setup, PGS (rest), position update, output ops, PGS motor rows, PGS limit rows, PGS contact rows}; entries 12..23 (helper-wave kernel only) = cycles after which wavefronts
 * 1, 2, 3 reached {pose hand-over, end of the update phase, final hand-over, their end}.  Never used for timing quotes: the
 * stamps serialise the instruction stream.  Pass NULL (default) for the production kernel. */
int32_t dg_world_set_profile_buffer(dg_world* w, uint64_t* cycles);

#ifdef __cplusplus
}
#endif
#endif
