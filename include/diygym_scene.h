/* diygym_scene.h -- flat scene description ("scene blob") shared by every
 * implementation of the DIYGym batched step path.
 *
 * This header defines a DATA FORMAT only: two arrays, `int32_t I[]` and
 * `double F[]`, that describe one environment (bodies, 1-DoF links, named
 * frames, collision shapes, the addon program and solver parameters).  The
 * host (diy_gym_amd/scene.py) emits it from the YAML config + URDF files; the
 * HIP library (diy_gym_amd/csrc) and the CPU oracle (oracle/) both consume it.
 * It is the batched replacement for the sequence of pybullet "model" calls the
 * reference issues while constructing an environment (reference:
 * diy_gym/diy_gym.py:74-91, diy_gym/model.py:53-83; SURVEY.md 8(b) groups
 * "world" and "model").
 *
 * All indices are 0-based.  Quaternions are xyzw.  Rotation matrices are
 * row-major 3x3 with x_parent = R * x_child + p.
 */
#ifndef DIYGYM_SCENE_H
#define DIYGYM_SCENE_H

#define DG_MAGIC 0x44475953 /* 'DGYS' */
#define DG_VERSION 13

/* ---- header ints ---------------------------------------------------- */
enum {
  DG_H_MAGIC = 0,
  DG_H_VERSION,
  DG_H_N_BODIES,
  DG_H_N_LINKS,      /* moving (1-DoF) links over all bodies == total DoF */
  DG_H_N_FRAMES,
  DG_H_N_SHAPES,
  DG_H_N_POINTS,     /* convex hull points over all SHAPE_POINTS shapes   */
  DG_H_N_PLANES,     /* convex hull face planes (ray casting only)         */
  DG_H_N_CAMERAS,
  DG_H_N_PAIRS,      /* shape pairs that may collide                       */
  DG_H_N_GROUPS,     /* runs of consecutive pairs between the same two bodies (broad phase) */
  DG_H_N_OPS,        /* addon program length                               */
  DG_H_N_ILIST,      /* length of the op int-list pool                     */
  DG_H_N_FLIST,      /* length of the op float-list pool                   */
  DG_H_ACT_DIM,
  DG_H_OBS_DIM,
  DG_H_REW_DIM,
  DG_H_TERM_DIM,
  DG_H_SUBSTEPS,     /* numSubSteps (reference diy_gym.py:77)              */
  DG_H_SOLVER_ITERS, /* numSolverIterations (reference diy_gym.py:78)      */
  DG_H_MAX_EPISODE_STEPS, /* -1 = disabled (reference diy_gym.py:57)        */
  DG_H_HOT_START,    /* sim steps after reset (reference diy_gym.py:58,145) */
  DG_H_IK_ITERS,
  DG_H_STATE_DIM,    /* floats of persistent state per env                 */
  DG_H_ADDON_STATE_OFF, /* offset of addon state inside the env state      */
  DG_H_N_ADDON_STATE,
  DG_H_MAX_CONTACTS,
  DG_H_REW_MODE,     /* DG_COLLAPSE_NONE | DG_COLLAPSE_SUM (reference diy_gym.py:94)      */
  DG_H_TERM_MODE,    /* DG_COLLAPSE_NONE | _ANY | _ALL (reference diy_gym.py:95-96)      */
  DG_H_N_TERM_GROUPS,/* receptors that own terminal columns (for _ALL)                   */
  DG_H_OFF_BODY_I,
  DG_H_OFF_LINK_I,
  DG_H_OFF_FRAME_I,
  DG_H_OFF_SHAPE_I,
  DG_H_OFF_PAIR_I,
  DG_H_OFF_GROUP_I,
  DG_H_OFF_OP_I,
  DG_H_OFF_ILIST,
  DG_H_OFF_BODY_F,
  DG_H_OFF_LINK_F,
  DG_H_OFF_FRAME_F,
  DG_H_OFF_SHAPE_F,
  DG_H_OFF_POINT_F,
  DG_H_OFF_PLANE_F,  /* nx ny nz d per face, n.x + d <= 0 inside, same frame as the points */
  DG_H_OFF_CAMERA_I,
  DG_H_OFF_CAMERA_F,
  DG_H_OFF_OP_F,
  DG_H_OFF_FLIST,
  DG_H_WARM_OFF,     /* state offset of the contact impulse cache (DG_WS_*), or -1: no warm starting */
  DG_H_N_CONSTRAINTS,/* fixed constraints between two bodies (child models attached with `attach: constraint`) */
  DG_H_OFF_CONS_I,
  DG_H_OFF_CONS_F,
  DG_H_INT_COUNT /* header length in I[] */
};

/* ---- header floats -------------------------------------------------- */
enum {
  DG_HF_DT = 0,        /* substep length = timestep / substeps            */
  DG_HF_GRAV_X, DG_HF_GRAV_Y, DG_HF_GRAV_Z,
  DG_HF_RESIDUAL_THRESHOLD, /* PGS early-out on squared velocity residual */
  DG_HF_CONTACT_ERP,
  DG_HF_LIMIT_ERP,
  DG_HF_LINEAR_SLOP,
  DG_HF_LIN_DAMPING,   /* btMultiBody linear damping k (K1 = K2 = k)      */
  DG_HF_ANG_DAMPING,
  DG_HF_MAX_COORD_VEL, /* joint velocity clamp                            */
  DG_HF_DEFAULT_MOTOR_IMPULSE, /* velocity motor every joint gets at load  */
  DG_HF_IK_LAMBDA_SQ,  /* DLS damping (task space, null-space variant)    */
  DG_HF_IK_JOINT_DAMPING, /* DLS2 diagonal (joint space variant)          */
  DG_HF_IK_RESIDUAL,   /* position residual threshold                     */
  DG_HF_IK_MAX_ANGLE,  /* per-iteration joint step clamp (rad)            */
  DG_HF_IK_NULL_REST_GAIN,
  DG_HF_IK_NULL_LIMIT_GAIN,
  DG_HF_CONTACT_MARGIN, /* speculative contact distance                   */
  DG_HF_WARMSTART,      /* a contact that persists (same pair, same feature) starts its NORMAL row from this factor x the
                           impulse it ended the previous substep with (Bullet: m_warmstartingFactor 0.85 [R]); 0 = off */
  DG_HF_WARMSTART_FRICTION, /* the same for its two friction rows (Bullet starts friction rows from zero [R])          */
  DG_HF_MOTOR_GUESS,    /* > 0: the motor rows of a body start from the clamped solution of the body's unclamped motor system
                           (M^-1 restricted to the motorised joints) lambda = b instead of from zero.  By the body's joint
                           count n (what the factorisation costs on the device decides the cut-offs):
                             n <= DG_MOTOR_GUESS_REFINE: if a row of the solution exceeds its bound, a primal-dual active set of at
                                  most DG_MOTOR_GUESS_ROUNDS rounds -- the rows beyond their bounds held there, the others solved
                                  again, the sets re-read from x + residual -- then the clamp (round 3: one round);
                             n <= DG_MOTOR_GUESS_MAX:    if a row exceeds its bound the body starts from zero (as without);
                             beyond:                     no starting guess (zero) */
  DG_HF_LIMIT_GUESS,    /* > 0 (with DG_HF_MOTOR_GUESS, bodies of n <= DG_MOTOR_GUESS_REFINE joints): a joint whose motor target
                           lies BEYOND an active joint-limit row (target velocity b_m above what the upper-limit row allows, or
                           below what the lower-limit row demands) enters the guess as ONE unknown -- the joint's total impulse
                           with the limit row's velocity as its right-hand side -- and starts with its motor saturated into the
                           limit and the limit row holding the rest (acc_limit = max force x h - |total|), unless the motor alone
                           is too weak to reach the limit velocity (then: held at its bound, limit row at zero).  From a motor-
                           only guess (or from zero) such a joint ramps its two rows up against each other by
                           (b_m - b_limit) / diag per sweep until the motor saturates: hundreds of sweeps for a target a
                           hair beyond the limit.  Same fixed point; 0 = motor rows only, as in round 3 */
  DG_HF_MOTOR_IMPULSE_SCALE, /* impulse bound of a motor row = max force x substep x this.  1: the substep is the time base
                           (default).  Set to the number of substeps for the other reading of pybullet -- maxAppliedImpulse =
                           force x fixedTimeStep, the FULL step, while the solver runs at fixedTimeStep / numSubSteps [R] --
                           engine parameter motor_impulse_timebase = 'step' */
  DG_HF_HULL_CONTACTS,  /* > 0: two convex hulls (URDF collision meshes, boxes on moving bodies) collide as HULLS -- GJK closest
                           points, an expanding polytope for the depth once they overlap -- one contact per pair, as Bullet's
                           btConvexConvexAlgorithm finds per call [R]; 0: through the capsule fitted to each hull (rounds 1-3) */
  DG_HF_HULL_MARGIN,    /* collision margin of a hull shape: the hull is inflated by this radius, i.e. the distance of two hulls
                           is the GJK distance minus twice this (gUrdfDefaultCollisionMargin = 0.001 [R]) */
  DG_HF_FLOAT_COUNT
};

/* ---- contact impulse cache (per-env state at DG_H_WARM_OFF, only when DG_HF_WARMSTART* > 0 and the scene has pairs) ----
 * [count] then max_contacts entries [key, normal, t1, t2]: the contacts of the env's most recent substep and the impulses
 * their rows ended with.  key = candidate pair index * 256 + feature (sphere / fitted capsule: 0; capsule against a box:
 * which end; hull against a box: the hull vertex index).  A reset clears the count. */
enum { DG_WS_KEY = 0, DG_WS_NORMAL, DG_WS_T1, DG_WS_T2, DG_WS_STRIDE };
#define DG_CONTACT_KEY(pair, feature) ((pair) * 256 + ((feature) & 255))  /* exact in fp32 up to 65 536 candidate pairs; hulls of up to 256 points (DIYGym's max_hull_points is capped there) keep one key per vertex */

/* ---- fixed constraints (reference model.py:74-75: p.createConstraint(parent, parent_frame, child, child_frame,
 * JOINT_FIXED, ...)) as SOLVER ROWS: three linear rows along the world axes at the pivot (pulling the pivot on side B onto
 * the pivot on side A) and three angular rows, bilateral, each bounded by MAX_FORCE x substep; position and orientation
 * errors are fed back with DG_HF_CONTACT_ERP.  Swept after the joint-limit rows and before the contact rows.  Pivots are
 * given in the LINK frame of their side (link < 0: the body's base link frame). */
enum { DG_KI_BODY_A = 0, DG_KI_LINK_A /* global link index or -1 */, DG_KI_BODY_B, DG_KI_LINK_B, DG_KI_STRIDE };
enum { DG_KF_POS_A = 0, DG_KF_QUAT_A = 3, DG_KF_POS_B = 7, DG_KF_QUAT_B = 10, DG_KF_MAX_FORCE = 14, DG_KF_STRIDE = 16 };
#define DG_MAX_CONSTRAINTS 4

#define DG_MOTOR_GUESS_REFINE 8
#ifndef DG_MOTOR_GUESS_ROUNDS
#define DG_MOTOR_GUESS_ROUNDS 4   /* rounds of the primal-dual active set in the motor guess (bodies of <= DG_MOTOR_GUESS_REFINE joints) */
#endif
#define DG_MOTOR_GUESS_MAX 10

/* ---- per-env state prefix --------------------------------------------- */
enum { DG_ST_STEP = 0 /* step_counter (reference diy_gym.py:139,206) */, DG_ST_EPISODE /* resets so far (RNG stream) */,
       DG_ST_PREFIX };

/* ---- body table ------------------------------------------------------ */
#define DG_BODY_FIXED 1  /* base does not move (use_fixed_base / massless root) */
#define DG_BODY_FROZEN 2 /* fixed, no joints and never respawned: its shapes are stored in WORLD coordinates,
                            it has no per-env state (DG_BI_STATE_OFF = -1) and its pose is DG_BF_INIT_* */
enum { DG_BI_FLAGS = 0, DG_BI_FIRST_LINK, DG_BI_N_LINKS, DG_BI_STATE_OFF,
       DG_BI_DYN_OFF /* state offset of the body's per-env angular damping (dynamics_randomizer), or -1: DG_HF_ANG_DAMPING */,
       DG_BI_PREV_OFF /* state offset where the body's generalised velocity at the START of a step's last substep is kept
                         (joint rates in link order, then base linvel3 angvel3 if floating) for force_torque_sensor, or -1 */,
       DG_BI_COLOR_OFF /* state offset of the body's per-env texture (visual_randomizer, DG_TX_*), or -1: the shapes' own colours */,
       DG_BI_STRIDE };
/* per-env state of a body at STATE_OFF: pos[3] quat[4] (base link frame, world);
 * then, for a floating base only, linvel[3] (of the base-frame origin, world)
 * angvel[3] (world); then ext_force[3] ext_torque[3] (world, about the base
 * origin... see DG_EXT_*); then per link see DG_LS_*.                        */
enum {
  DG_BF_MASS = 0, DG_BF_COM = 1,        /* lumped base: mass, com[3]          */
  DG_BF_INERTIA = 4,                    /* xx xy xz yy yz zz about the com    */
  DG_BF_INIT_POS = 10, DG_BF_INIT_QUAT = 13, /* base link frame at load      */
  DG_BF_REPORT_POS = 17, DG_BF_REPORT_QUAT = 20, /* root inertial frame in the
                                          base link frame: what pybullet's
                                          getBasePositionAndOrientation reports */
  DG_BF_COLOR = 24,                     /* rgba, visual only                  */
  DG_BF_BOUND = 28,                     /* radius around the base origin that contains every collision shape in any joint configuration */
  DG_BF_STRIDE = 32
};

/* ---- link table (one row per DoF, bodies contiguous, parents first) -- */
enum { DG_LI_PARENT = 0 /* global link index, -1 = base of own body */, DG_LI_TYPE /* 0 revolute 1 prismatic */,
       DG_LI_BODY, DG_LI_STATE_OFF,
       DG_LI_MASS_SCALE /* state offset of the link's per-env mass (and inertia) scale (dynamics_randomizer), or -1: 1 */,
       DG_LI_STRIDE };
enum {
  DG_LF_POS = 0,   /* joint frame origin in the parent reference frame      */
  DG_LF_ROT = 3,   /* 3x3                                                    */
  DG_LF_AXIS = 12, /* in the link frame                                      */
  DG_LF_MASS = 15, DG_LF_COM = 16, DG_LF_INERTIA = 19,
  DG_LF_DAMPING = 25, DG_LF_LOWER = 26, DG_LF_UPPER = 27, DG_LF_MAX_FORCE = 28, DG_LF_MAX_VEL = 29,
  DG_LF_STRIDE = 32
};
/* per-env state of a link at DG_LI_STATE_OFF */
enum { DG_LS_Q = 0, DG_LS_QD, DG_LS_TARGET_POS, DG_LS_TARGET_VEL, DG_LS_TORQUE /* joint torque for the next step */,
       DG_LS_APPLIED /* motor torque applied during the last substep */, DG_LS_STRIDE };

/* offsets inside the body state block */
enum { DG_BS_POS = 0, DG_BS_QUAT = 3, DG_BS_FIXED_END = 7, DG_BS_LINVEL = 7, DG_BS_ANGVEL = 10, DG_BS_FLOAT_END = 13 };
/* after FIXED_END / FLOAT_END: external wrench for the next step, world frame,
 * torque taken about the base-frame origin of the body */
enum { DG_EXT_FORCE = 0, DG_EXT_TORQUE = 3, DG_EXT_STRIDE = 6 };

/* ---- motor configuration (uniform over envs, owned by the world) ----- */
enum { DG_MC_KP = 0, DG_MC_KD, DG_MC_MAX_IMPULSE_SCALE /* max force; impulse = force*dt, or <0: raw impulse */,
       DG_MC_STRIDE };

/* ---- frame table ------------------------------------------------------ */
enum { DG_FI_BODY = 0, DG_FI_LINK /* global link index, -1 = base */, DG_FI_STRIDE };
enum { DG_FF_POS = 0, DG_FF_QUAT = 3, DG_FF_COM_POS = 7, DG_FF_COM_QUAT = 10, DG_FF_STRIDE = 14 };

/* ---- shapes ----------------------------------------------------------- */
enum { DG_SHAPE_SPHERE = 0, DG_SHAPE_BOX = 1, DG_SHAPE_CAPSULE = 2, DG_SHAPE_POINTS = 3 };
enum { DG_SI_TYPE = 0, DG_SI_BODY, DG_SI_LINK, DG_SI_POINT_OFF, DG_SI_N_POINTS, DG_SI_PLANE_OFF, DG_SI_N_PLANES, DG_SI_FLAGS, DG_SI_STRIDE };
#define DG_SHAPE_WORLD 1 /* transform (and points / planes) are already in world coordinates (frozen body) */
#define DG_SHAPE_NO_COLLIDE 2 /* visual only: seen by cameras, ignored by the narrow phase */
/* flags bits 8..23: (pybullet link index of the owning URDF link) + 1, 0 = base; used by segmentation masks */
enum { DG_SF_POS = 0, DG_SF_ROT = 3, DG_SF_PARAMS = 12 /* sphere r | box half[3] | capsule r, half_len (axis = local z) */,
       DG_SF_FRICTION = 15,
       DG_SF_COLOR = 16 /* rgb of the shape in camera images: the URDF <material><color> of its link's first <visual>, the
                           YAML `color` for shapes of the base link (reference model.py:82-83: changeVisualShape(uid, -1, rgbaColor)),
                           else grey 0.8 */,
       DG_SF_HULL_HALF = 19 /* DG_SHAPE_POINTS: half length of the capsule of radius PARAMS[0] along the fitted capsule's axis that
                           CONTAINS every hull point (the fitted one, PARAMS[1], lets points near its caps stick out); PARAMS[2] =
                           radius of the sphere around the capsule's centre that contains them.  Culling data of the hull-hull
                           narrow phase (DG_HF_HULL_CONTACTS) */,
       DG_SF_STRIDE = 20 };
/* ---- procedural textures (visual_randomizer) ------------------------------------------------------------------------
 * The per-env addon state of a visual_randomizer op: colour A rgb, colour B rgb, frequency [cells per metre], kind.
 * With p = the hit point in the shape's reference frame (a hull: its link frame; other shapes: the shape frame) and
 * u = floor(p * frequency) (three integers):
 *   DG_TEX_FLAT     colour A
 *   DG_TEX_CHECKER  (ux + uy + uz) odd ? B : A
 *   DG_TEX_STRIPES  ux odd ? B : A
 *   DG_TEX_CELLS    A + (B - A) t,  t = dg_tex_hash(ux, uy, uz) / 2^24  (every cell its own blend)
 * They stand in for the "describable textures" images the reference swaps in (visual_randomizer.py:33-46; its 600 MB
 * download :48-77 is out of scope). */
enum { DG_TX_A = 0, DG_TX_B = 3, DG_TX_FREQ = 6, DG_TX_KIND = 7, DG_TX_STRIDE = 8 };
enum { DG_TEX_FLAT = 0, DG_TEX_CHECKER = 1, DG_TEX_STRIPES = 2, DG_TEX_CELLS = 3 };
/* 24-bit hash of a cell (uint32 wrap-around arithmetic; the same in every implementation) */
#define DG_TEX_HASH(ux, uy, uz, h) do { uint32_t h_ = (uint32_t)(ux) * 73856093u ^ (uint32_t)(uy) * 19349663u ^ (uint32_t)(uz) * 83492791u; \
    h_ ^= h_ >> 15; h_ *= 0x2C1B3C6Du; h_ ^= h_ >> 12; h_ *= 0x297A2D39u; h_ ^= h_ >> 15; (h) = h_ >> 8; } while (0)
enum { DG_PI_A = 0, DG_PI_B, DG_PI_STRIDE };
/* pair groups: the pair list is ordered so that all pairs between one moving body and one shape of the static world
 * (or between two moving bodies) are consecutive; a group is culled as a whole with bounding spheres */
enum { DG_GI_FIRST = 0, DG_GI_COUNT, DG_GI_BODY_A /* moving */, DG_GI_BODY_B /* moving, or -1 */, DG_GI_STATIC_SHAPE /* or -1 */, DG_GI_STRIDE };

/* ---- cameras (reference diy_gym/addons/sensors/camera.py:26-98) ------- */
/* A camera is rendered by its own launch (dg_world_render), not by the step kernel. */
enum { DG_CI_BODY = 0 /* -1: fixed in the world */, DG_CI_FRAME /* global frame index or -1 = base */, DG_CI_WIDTH, DG_CI_HEIGHT,
       DG_CI_FLAGS, DG_CI_STRIDE };
enum { DG_CF_POS = 0, DG_CF_QUAT = 3 /* T_parent_cam */, DG_CF_FOV = 7 /* vertical, degrees */, DG_CF_NEAR = 8, DG_CF_FAR = 9,
       DG_CF_TAN_HALF_FOV = 10 /* tan(fov / 2), so that no kernel evaluates a tangent */, DG_CF_STRIDE = 12 };
#define DG_CAM_DEPTH 1
#define DG_CAM_SEGMENTATION 2
/* per-body flat colour for the (non parity) rgb output lives in the body float table */

/* ---- addon program ---------------------------------------------------- */
/* phases: an op runs in exactly one phase */
enum {
  DG_OP_NOP = 0,
  /* update phase (reference Addon.update, diy_gym.py:202-204) */
  DG_OP_JOINT_CONTROL = 1,   /* joint_controller.py:40-58 */
  DG_OP_IK_CONTROL = 2,      /* ik_controller.py:51-80    */
  DG_OP_EXTERNAL_FORCE = 3,  /* external_force.py:21-24   */
  DG_OP_PROPELLOR = 4,       /* examples/drone_pilot/drone_pilot.py:31-37 */
  DG_OP_ADMITTANCE = 5,      /* admittance_controller.py:36-55: J^T wrench + gravity compensation + joint PD -> torques */
  /* reset phase (reference Addon.reset, diy_gym.py:141-143) */
  DG_OP_RESPAWN = 16,        /* respawn.py:31-39 */
  DG_OP_RESET_JOINTS = 17,   /* joint_controller.py:36-38, ik_controller.py:47-49 */
  DG_OP_RANDOMIZE_DYNAMICS = 18, /* dynamics_randomizer.py:24-32: per joint k (ILIST, FLIST = URDF joint damping):
                                  mass_k <- log(U(f0, f1)) * mass_k (compounding, as the reference reads the current mass back),
                                  angularDamping <- log(U(f2, f3)) * jointDamping_k (body-wide: the last joint's draw stays).
                                  Addon state: N mass scales, then the angular damping.  Guards (the reference formula goes
                                  negative for U < 1): |log U| for the mass, scale clamped to [f4, f5]; damping >= 0.  Drawn
                                  twice at an env's first reset (the reference draws at construction and again in reset()) */
  DG_OP_RANDOMIZE_COLOR = 19,    /* visual_randomizer.py:40-46 with procedural textures instead of its image data set: per env
                                    and episode colours A, B ~ U(0,1)^3, frequency 2 + 14 U, kind 1 + floor(3 U) from the counter
                                    RNG (components 0..7), kept in DG_TX_STRIDE floats of addon state (camera rgb only) */
  /* observe phase */
  DG_OP_OBS_JOINT_STATE = 32,  /* joint_state_sensor.py:47-57 */
  DG_OP_OBS_OBJECT_STATE = 33, /* object_state_sensor.py:49-75 */
  DG_OP_OBS_ADDON_STATE = 34,  /* drone_pilot.py:39-40 */
  DG_OP_OBS_FT = 35,           /* force_torque_sensor.py:14-23: reaction wrench across joint FRAME, 6 columns (force, torque).
                                  ILIST = [n_moving, moving links on the child side..., n_shapes, shapes on the child side...],
                                  FLIST = mass com[3] inertia[6] of the rigid cluster on the child side (anchor link frame),
                                  FLAGS & DG_FT_WHOLE_LINK: the joint is the anchor link's own (movable) joint */
  /* reward phase */
  DG_OP_REW_REACH = 48,        /* reach_target.py:32-33 */
  DG_OP_REW_ELECTRICITY = 49,  /* electricity_cost.py:15-18 */
  DG_OP_REW_CONST = 50,        /* time_penalty.py:11-12 */
  /* terminal phase */
  DG_OP_TERM_REACH = 64,       /* reach_target.py:35-36 */
  DG_OP_TERM_TILT = 65,        /* drone_pilot.py:53-55 */
  DG_OP_TERM_TIMER = 66        /* diy_gym.py:180-183 */
};
enum {
  DG_OI_CODE = 0,
  DG_OI_BODY,      /* primary body                                          */
  DG_OI_FRAME,     /* frame index (global) or -1 = base                      */
  DG_OI_BODY2,     /* secondary body (source) or -1                          */
  DG_OI_FRAME2,
  DG_OI_FLAGS,
  DG_OI_N,         /* list length                                            */
  DG_OI_ILIST,     /* offset into the int-list pool (global link indices)    */
  DG_OI_FLIST,     /* offset into the float-list pool                        */
  DG_OI_IO_OFF,    /* column in the action / obs / reward / terminal buffer  */
  DG_OI_STATE_OFF, /* addon state offset (relative to ADDON_STATE_OFF)       */
  DG_OI_SLOT,      /* update ops: bit index in the per-step update mask      */
  DG_OI_STRIDE
};
enum { DG_OF_STRIDE = 16 };

/* DG_OP_JOINT_CONTROL flags */
enum { DG_JC_POSITION = 0, DG_JC_VELOCITY = 1, DG_JC_TORQUE = 2 };
/* DG_OP_IK_CONTROL flags */
#define DG_IK_USE_ORIENTATION 1
#define DG_IK_NULLSPACE 2
/* DG_OP_OBS_JOINT_STATE flags */
#define DG_JS_VELOCITY 1
#define DG_JS_EFFORT 2
/* DG_OP_OBS_OBJECT_STATE flags */
#define DG_OS_ROTATION 1
#define DG_OS_VELOCITY 2
/* DG_OP_OBS_FT flags */
#define DG_FT_WHOLE_LINK 1
/* DG_OP_RESPAWN flags */
#define DG_RS_ONCE 1

/* collapse modes for the reward / terminal outputs */
enum { DG_COLLAPSE_NONE = 0, DG_COLLAPSE_SUM = 1, DG_COLLAPSE_ANY = 2, DG_COLLAPSE_ALL = 3 };

#endif /* DIYGYM_SCENE_H */
