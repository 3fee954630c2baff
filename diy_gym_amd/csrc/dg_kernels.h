// dg_kernels.h -- the batched DIYGym step path as hand-written HIP for gfx950.
//
// One environment per lane.  step_kernel runs a workgroup of ONE wavefront (64 threads); LANES of its lanes own
// an environment each (64 normally; 32 / 16 / 8 / 4 when a scene's per-env scratch would not fit 160 KiB of LDS
// at 64 -- the spare lanes then join the Gauss-Seidel sweeps) and all cross-lane traffic is wave ballots / DPP.
// step_kernel_par (dg_entry.h) runs FOUR wavefronts per workgroup on the same 64 envs and the same LDS workspace
// and hands work between them through __syncthreads.
//
// What the kernels replace (reference call sites):
//   step_kernel   : DIYGym.step (diy_gym/diy_gym.py:187-209) = addon.update for
//                   every controller + p.stepSimulation + observe/reward/terminal
//   reset_kernel  : DIYGym.reset (diy_gym.py:130-148), per-env masked
//   observe_kernel: DIYGym.observe/reward/is_terminal (diy_gym.py:150-185)
//
// Algorithm per substep (same mathematics as the fp64 CPU checker under oracle/, different
// organisation): world kinematics -> narrow-phase contacts -> per body:
// articulated-body algorithm in link coordinates (block-form articulated
// inertias) + velocity update + the body's inverse mass matrix M^-1 from ABA
// impulse responses -> velocity-level rows (motors, joint limits, contact
// normal + 2 friction) with world-frame Jacobians and M^-1 J^T responses ->
// projected Gauss-Seidel with a per-env residual early-out and a wave-level
// "everyone converged" exit -> semi-implicit position update.
#pragma once
#include "dg_device.h"
#include "../../include/diygym_scene.h"
#include "../../include/diygym_hip.h"

namespace dg {

#define DG_MAX_LINKS 64   // total 1-DoF links in a scene (motor table travels in kernarg)
#define DG_MAX_BODIES 192

// Scene tables are immutable for the lifetime of a world.  Their pointers are typed as CONSTANT
// address space (AS4) so that every wave-uniform read is provably invariant and becomes an s_load
// through the scalar data cache; with plain global pointers the compiler must assume the state stores
// may alias the tables and falls back to 64-lane vector loads of one address.
#define DG_CONSTANT __attribute__((address_space(4)))
typedef const int32_t DG_CONSTANT* cip;
typedef const float DG_CONSTANT* cfp;

// Everything wave-uniform the kernels need.  Passed by value (kernarg -> SGPRs / scalar loads).
struct DevScene {
  cip BI, LI, FI, SI, PI, GI, OI, IL;
  cfp BF, LF, FF, SF, PF, OF, FL, HF;
  cip PLB;  // per body: [R0_off, minv_off, dv_off, nv]
  int split_pgs;  // helper-wave kernel: the two register-chain bodies are (main wave: reg_body[0], helper: helper_body) and no other body has joints
  int early_dyn;  // with coll_wave: the first substep's narrow phase and dynamics run on that wavefront DURING the update ops (no op writes torques / forces)
  int coll_split, cont2_off;  // with coll_wave: a fourth wavefront tests the second half of the pair table into its own contact list at cont2_off
  int coll_wave;  // helper-wave step kernel: a third wavefront runs the narrow phase (every moving body has register-resident dynamics)
  cfp GD;   // per pair group, device-only: [x y z reach] of a frozen static partner (reach < 0: none), see dg_world_create
  cip AM;   // per link, device-only: bit i set = link first + i of the same body is this link or one of its ancestors (minv_sliced)
  cip SD;   // per shape, device-only: [pose slot | base position state offset or -1 | first hull point | hull points], see dg_world_create
  cip PD;   // per candidate pair, device-only: first shape | second << 12 | types << 24 | swapped << 28 (canonical order)
  cip PLL;  // per link: [pose_off, mrow_off, iaacc_off]
  int32_t nba;  // 1 + the last body that is not frozen in the world: per-body loops of the step stop here (a maze is one robot + 120 frozen walls)
  int32_t no_sliced_reset;  // DG_NO_SLICED_RESET
  int32_t no_minv_slices;  // DG_NO_MINV_SLICES: the M^-1 columns stay with one lane per env (ablation / tests)
  int32_t nsha; // 1 + the last shape that is not an analytic box (the narrow phase caches a segment per round shape)
  int32_t debug_keep_ext;  // DG_DEBUG_KEEP_EXT: external wrenches and joint torques are NOT cleared at the end of a step (diagnostic: lets a test read what the update ops applied)
  int32_t no_chain_rows;  // DG_NO_CHAIN_ROWS: contacts between two register-chain bodies build their rows pair by pair (ablation / tests)
  int32_t warm_off;  // state offset of the contact impulse cache (DG_WS_*), -1: no warm starting
  float* hull_ws;  // hull-hull narrow phase (dg_hull.h): HH_WS_SLOTS x 64 floats per wavefront of the step grid, or null (no pair of two hulls)
  int32_t ncons; cip KI; cfp KF;  // fixed constraints between two bodies (DG_KI_*, DG_KF_*): six solver rows each, generic sweeps only
  int32_t nb, nl, nfr, nsh, npairs, ngroups, nops, act_dim, obs_dim, rew_dim, term_dim, substeps, iters, hot_start, ik_iters;
  int32_t state_dim, addon_off, max_contacts, term_mode, n_term_groups;
  int32_t tr_off, tr_slots, cont_off, nv_max, total_slots, ab_stride;  // LDS plan
  int32_t nt, dv_base, dense;  // total DoF over moving bodies; LDS offset of the first velocity block; dense rows (nt <= 32)
  int32_t crow_tail;    // contact row: [JA nv_max][RA nv_max]([JB][RB] only if some pair has two moving bodies)[b][acc][diag]
  int32_t helper_body;  // fixed-base chain body whose IK and dynamics a second wavefront of the workgroup runs (-1: none)
  int32_t reg_body[2];  // up to two fixed-base bodies with <= 6 joints whose solver rows live in registers (-1: none)
  int32_t num_envs, stride;
  uint64_t seed; int64_t env_base;
  float h, gx, gy, gz;
  float hm;  // time base of a motor row's impulse bound: h x DG_HF_MOTOR_IMPULSE_SCALE (the substep, or the full step)
};
struct MotorTable { float v[DG_MAX_LINKS * 3]; };  // kp, kd, max_force (<0 raw impulse)

enum { PLB_R0 = 0, PLB_MINV, PLB_DV, PLB_NV, PLB_CHAIN /* fixed-base serial chain of <= 6 joints: register-resident dynamics */, PLB_STRIDE };
enum { PLL_POSE = 0, PLL_MROW, PLL_IAACC /* LDS accumulator for children that are not link+1, or -1 */, PLL_STRIDE };
// transient ABA workspace per link.  Articulated inertias are NOT stored per link: along a chain
// (parent == link - 1) the child's contribution is carried in registers; only links with a child that
// is not their immediate successor get a 21-slot LDS accumulator (PLL_IAACC).
enum { AW_E = 0, AW_R = 9, AW_V = 12, AW_PA = 18, AW_U = 24, AW_D = 30, AW_UU = 31, AW_STRIDE = 32 };
// transient base block (before the per-link blocks): V, A always; IA, PA, L only when a floating base exists
enum { AB_V = 0, AB_A = 6, AB_FIXED_STRIDE = 12, AB_IA = 12, AB_PA = 33, AB_L = 39, AB_FLOAT_STRIDE = 60 };
// contact list entry (body info is per lane: the pair id differs between lanes)
// (CL_KEY: DG_CONTACT_KEY(pair, feature), the contact's identity from substep to substep -- warm starting)
enum { CL_PAIR = 0, CL_P = 1, CL_N = 4, CL_DIST = 7, CL_KEY = 8, CL_DVA = 9, CL_NVA = 10, CL_DVB = 11, CL_NVB = 12, CL_MU = 13, CL_STRIDE = 14 };
// motor / limit row block per link: b_motor acc_motor b_lo acc_lo b_hi acc_hi
enum { MR_B = 0, MR_ACC, MR_LO_B, MR_LO_ACC, MR_HI_B, MR_HI_ACC, MR_STRIDE };

// Workspace modes (the LANES template argument everywhere):
//   64 / 32 / 16 / 8 / 4  that many envs per wavefront, per-env scratch in LDS as ws[slot][lane]; below 64 the
//                 spare lanes join the dense Gauss-Seidel sweeps (8 and 4 only exist for scenes that can use them)
//   0             64 envs per wavefront, scratch in a global buffer [workgroup][slot][lane] (scene too big for LDS)
//   -16           16 envs per wavefront, global scratch; the 48 spare lanes join the dense Gauss-Seidel sweeps
constexpr int envs_per_wave(int lanes) { return lanes > 0 ? lanes : (lanes < 0 ? -lanes : 64); }

// this lane's workspace column: LDS, or its workgroup's block of the global scratch buffer
template <int LANES>
DGD float* workspace_of(const DevScene& sc, float* smem, float* gws, int lane) {
  if constexpr (LANES > 0) return smem + lane;
  else return gws + (size_t)blockIdx.x * (size_t)sc.total_slots * envs_per_wave(LANES) + lane;
}

template <int LANES>
struct Lane {
  const DevScene& sc;
  const MotorTable& mt;
  float* lds;   // workspace base for this lane (already offset by workgroup and lane): LDS, or -- LANES <= 0 -- this
                // workgroup's block of a global scratch buffer [workgroup][slot][lane] for scenes too big for 160 KiB of LDS
  float* st;    // state base for this env (already offset by env)
  int env;      // clamped env index
  bool valid;   // this lane owns a real env (stores allowed)

  DGD Lane(const DevScene& s, const MotorTable& m, float* l, float* state, int e, bool v) : sc(s), mt(m), lds(l), st(state), env(e), valid(v) {}

  // every mode: consecutive slots are envs_per_wave floats apart (LDS: ws[slot][lane]; global: [workgroup][slot][lane])
  DGD float& L(int slot) const { return lds[slot * envs_per_wave(LANES)]; }
  DGD float S(int k) const { return st[(size_t)k * sc.stride]; }
  DGD void Sset(int k, float v) const { if (valid) st[(size_t)k * sc.stride] = v; }

  DGD V3 L3(int o) const { return v3(L(o), L(o + 1), L(o + 2)); }
  DGD void L3set(int o, V3 v) const { L(o) = v.x; L(o + 1) = v.y; L(o + 2) = v.z; }
  DGD M3 LM(int o) const { M3 A; _Pragma("unroll") for (int k = 0; k < 9; k++) A.m[k] = L(o + k); return A; }
  DGD void LMset(int o, const M3& A) const { _Pragma("unroll") for (int k = 0; k < 9; k++) L(o + k) = A.m[k]; }
  // POSE entries: first two columns of R (6 slots), then the position (3 slots)
  DGD M3 LR(int o) const {
    V3 c0 = v3(L(o), L(o + 1), L(o + 2)), c1 = v3(L(o + 3), L(o + 4), L(o + 5)), c2 = cross(c0, c1);
    M3 R = {{c0.x, c1.x, c2.x, c0.y, c1.y, c2.y, c0.z, c1.z, c2.z}}; return R;
  }
  DGD void LRset(int o, const M3& R) const { L(o) = R.m[0]; L(o + 1) = R.m[3]; L(o + 2) = R.m[6]; L(o + 3) = R.m[1]; L(o + 4) = R.m[4]; L(o + 5) = R.m[7]; }
  DGD S6 L6(int o) const { S6 s = {L3(o), L3(o + 3)}; return s; }
  DGD void L6set(int o, const S6& s) const { L3set(o, s.a); L3set(o + 3, s.l); }
  DGD void L6add(int o, const S6& s) const { L(o) += s.a.x; L(o + 1) += s.a.y; L(o + 2) += s.a.z; L(o + 3) += s.l.x; L(o + 4) += s.l.y; L(o + 5) += s.l.z; }
  DGD AI LAI(int o) const {
    AI A; A.I.xx = L(o); A.I.xy = L(o + 1); A.I.xz = L(o + 2); A.I.yy = L(o + 3); A.I.yz = L(o + 4); A.I.zz = L(o + 5);
    _Pragma("unroll") for (int k = 0; k < 9; k++) A.H.m[k] = L(o + 6 + k);
    A.M.xx = L(o + 15); A.M.xy = L(o + 16); A.M.xz = L(o + 17); A.M.yy = L(o + 18); A.M.yz = L(o + 19); A.M.zz = L(o + 20);
    return A;
  }
  DGD void LAIset(int o, const AI& A) const {
    L(o) = A.I.xx; L(o + 1) = A.I.xy; L(o + 2) = A.I.xz; L(o + 3) = A.I.yy; L(o + 4) = A.I.yz; L(o + 5) = A.I.zz;
    _Pragma("unroll") for (int k = 0; k < 9; k++) L(o + 6 + k) = A.H.m[k];
    L(o + 15) = A.M.xx; L(o + 16) = A.M.xy; L(o + 17) = A.M.xz; L(o + 18) = A.M.yy; L(o + 19) = A.M.yz; L(o + 20) = A.M.zz;
  }
  DGD void LAIadd(int o, const AI& A) const {
    L(o) += A.I.xx; L(o + 1) += A.I.xy; L(o + 2) += A.I.xz; L(o + 3) += A.I.yy; L(o + 4) += A.I.yz; L(o + 5) += A.I.zz;
    _Pragma("unroll") for (int k = 0; k < 9; k++) L(o + 6 + k) += A.H.m[k];
    L(o + 15) += A.M.xx; L(o + 16) += A.M.xy; L(o + 17) += A.M.xz; L(o + 18) += A.M.yy; L(o + 19) += A.M.yz; L(o + 20) += A.M.zz;
  }

  // ---- scene table accessors (wave-uniform indices) ----
  DGD cip bi(int b) const { return sc.BI + b * DG_BI_STRIDE; }
  DGD cfp bf(int b) const { return sc.BF + b * DG_BF_STRIDE; }
  DGD cip li(int l) const { return sc.LI + l * DG_LI_STRIDE; }
  DGD cfp lf(int l) const { return sc.LF + l * DG_LF_STRIDE; }
  DGD cip plb(int b) const { return sc.PLB + b * PLB_STRIDE; }
  DGD cip pll(int l) const { return sc.PLL + l * PLL_STRIDE; }
  DGD bool fixed(int b) const { return bi(b)[DG_BI_FLAGS] & DG_BODY_FIXED; }
  DGD bool frozen(int b) const { return bi(b)[DG_BI_FLAGS] & DG_BODY_FROZEN; }
  DGD int ext_off(int b) const { return bi(b)[DG_BI_STATE_OFF] + (fixed(b) ? DG_BS_FIXED_END : DG_BS_FLOAT_END); }
  // per-env dynamics parameters written by the dynamics_randomizer's reset op (wave-uniform offsets, -1 = none)
  DGD float mass_scale(int gl) const { const int o = li(gl)[DG_LI_MASS_SCALE]; return o >= 0 ? S(o) : 1.0f; }
  DGD float ang_damping(int b) const { const int o = bi(b)[DG_BI_DYN_OFF]; return o >= 0 ? S(o) : sc.HF[DG_HF_ANG_DAMPING]; }
  // rigid inertia of link gl about its frame origin with the per-env mass scale applied to mass and inertia tensor
  DGD void link_inertia(int gl, float& m, V3& c, Sym3& Ic) const {
    cfp f = lf(gl); m = f[DG_LF_MASS]; c = v3(f[DG_LF_COM], f[DG_LF_COM + 1], f[DG_LF_COM + 2]); Ic = sym6(f + DG_LF_INERTIA);
    if (li(gl)[DG_LI_MASS_SCALE] >= 0) { const float s = mass_scale(gl); m *= s; Ic.xx *= s; Ic.xy *= s; Ic.xz *= s; Ic.yy *= s; Ic.yz *= s; Ic.zz *= s; }
  }

  DGD V3 base_pos(int b) const {
    if (frozen(b)) { cfp f = bf(b) + DG_BF_INIT_POS; return v3(f[0], f[1], f[2]); }
    int o = bi(b)[DG_BI_STATE_OFF]; return v3(S(o), S(o + 1), S(o + 2));
  }
  DGD Q4 base_quat(int b) const {
    if (frozen(b)) { cfp f = bf(b) + DG_BF_INIT_QUAT; Q4 q = {f[0], f[1], f[2], f[3]}; return q; }
    int o = bi(b)[DG_BI_STATE_OFF] + DG_BS_QUAT; Q4 q = {S(o), S(o + 1), S(o + 2), S(o + 3)}; return q;
  }

  // link (global index, -1 = base of body b) world frame from the POSE region
  DGD void link_world(int b, int gl, M3& R, V3& p) const {
    if (gl < 0) { R = frozen(b) ? qmat(base_quat(b)) : LR(plb(b)[PLB_R0]); p = base_pos(b); }
    else { int o = pll(gl)[PLL_POSE]; R = LR(o); p = L3(o + 6); }
  }

  // parent->child rotation (Rpc) and offset r for link gl at joint value q
  DGD void joint_xform(int gl, float q, M3& Rpc, V3& r) const {
    cfp f = lf(gl);
    M3 RT; _Pragma("unroll") for (int k = 0; k < 9; k++) RT.m[k] = f[DG_LF_ROT + k];
    V3 pT = v3(f[DG_LF_POS], f[DG_LF_POS + 1], f[DG_LF_POS + 2]);
    V3 ax = v3(f[DG_LF_AXIS], f[DG_LF_AXIS + 1], f[DG_LF_AXIS + 2]);
    if (li(gl)[DG_LI_TYPE] == 0) { Rpc = mul(RT, rot_axis(ax, q)); r = pT; }
    else { Rpc = RT; r = pT + mul(RT, ax * q); }
  }

  // ------------------------------------------------------------ kinematics
  // world pose of every link of body b into the POSE region (q from state, or from LDS at qoff when qoff >= 0)
  // The same for a fixed-base serial chain of at most six joints (PLB_CHAIN: every 6-axis arm): the parent's pose is carried in
  // registers instead of being read back from the POSE slots it was just written to -- a lone wavefront pays that LDS round
  // trip per link, and the whole call was ~10 k cycles for ~130 instructions of arithmetic per link (three calls per step: 14 %
  // of step_kernel_par).  The carried rotation is what LR() would read back -- the two stored columns and their cross product
  // -- so the poses are the same bits as the general loop's.
  DGD void kinematics_chain(int b) const {
    constexpr int N = 6;
    cip B = bi(b); const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS];
    M3 R = qmat(base_quat(b)); V3 p = base_pos(b);
    LRset(plb(b)[PLB_R0], R);
    float qv[N];
#pragma unroll
    for (int j = 0; j < N; j++) qv[j] = S(li(first + min(j, n - 1))[DG_LI_STATE_OFF] + DG_LS_Q);  // every load before the first transform
#pragma unroll
    for (int j = 0; j < N; j++) {
      if (j < n) {
        const int gl = first + j; M3 Rpc; V3 r; joint_xform(gl, qv[j], Rpc, r);
        const M3 Rn = mul(R, Rpc); p = p + mul(R, r);
        const int o = pll(gl)[PLL_POSE];
        LRset(o, Rn); L3set(o + 6, p);
        const V3 c0 = v3(Rn.m[0], Rn.m[3], Rn.m[6]), c1 = v3(Rn.m[1], Rn.m[4], Rn.m[7]), c2 = cross(c0, c1);
        const M3 Rb = {{c0.x, c1.x, c2.x, c0.y, c1.y, c2.y, c0.z, c1.z, c2.z}}; R = Rb;
      }
    }
  }
  DGD void kinematics(int b, int qoff = -1) const {
    cip B = bi(b);
    if (B[DG_BI_FLAGS] & DG_BODY_FROZEN) return;  // no per-env pose storage: its shapes are in world coordinates
    if (qoff < 0 && plb(b)[PLB_CHAIN]) { kinematics_chain(b); return; }
    int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS];
    M3 R0 = qmat(base_quat(b)); V3 p0 = base_pos(b);
    LRset(plb(b)[PLB_R0], R0);
    // joint angles eight links at a time, all loads issued before the first transform: a link at a time pays one
    // scalar-table round trip plus one global round trip per link (cold at the start of a step: ~1.4 k cycles each)
    constexpr int KCH = 8;
    for (int i0 = 0; i0 < n; i0 += KCH) {
      float qv[KCH];
#pragma unroll
      for (int j = 0; j < KCH; j++) { const int i = min(i0 + j, n - 1); qv[j] = qoff >= 0 ? L(qoff + i) : S(li(first + i)[DG_LI_STATE_OFF] + DG_LS_Q); }
#pragma unroll
      for (int j = 0; j < KCH; j++) {
        const int i = i0 + j; if (i >= n) break;
        int gl = first + i, par = li(gl)[DG_LI_PARENT];
        M3 Rpc; V3 r; joint_xform(gl, qv[j], Rpc, r);
        M3 Rp; V3 pp;
        if (par < 0) { Rp = R0; pp = p0; } else { int o = pll(par)[PLL_POSE]; Rp = LR(o); pp = L3(o + 6); }
        int o = pll(gl)[PLL_POSE];
        LRset(o, mul(Rp, Rpc)); L3set(o + 6, pp + mul(Rp, r));
      }
    }
  }

  // world pose + velocity of frame fr (global frame index, -1 base) of body b; POSE must be current
  DGD void frame_state(int b, int fr, bool com, V3& p, Q4& q, V3& v, V3& w, bool want_vel) const {
    V3 off; Q4 qo; int gl;
    if (fr < 0) {
      gl = -1;
      if (com) { cfp f = bf(b); off = v3(f[DG_BF_REPORT_POS], f[DG_BF_REPORT_POS + 1], f[DG_BF_REPORT_POS + 2]);
        Q4 t = {f[DG_BF_REPORT_QUAT], f[DG_BF_REPORT_QUAT + 1], f[DG_BF_REPORT_QUAT + 2], f[DG_BF_REPORT_QUAT + 3]}; qo = t; }
      else { off = v3(0, 0, 0); Q4 t = {0, 0, 0, 1}; qo = t; }
    } else {
      gl = sc.FI[fr * DG_FI_STRIDE + DG_FI_LINK];
      cfp f = sc.FF + fr * DG_FF_STRIDE + (com ? DG_FF_COM_POS : DG_FF_POS);
      off = v3(f[0], f[1], f[2]); Q4 t = {f[3], f[4], f[5], f[6]}; qo = t;
    }
    M3 R; V3 o; link_world(b, gl, R, o);
    Q4 ql = gl < 0 ? base_quat(b) : qfrom_mat(R);
    p = o + mul(R, off); q = qnormalize(qmul(ql, qo));
    if (!want_vel) return;
    // velocity of the link origin: walk the chain root -> link
    V3 wl = v3(0, 0, 0), vl = v3(0, 0, 0), po = base_pos(b);
    if (!fixed(b)) { int so = bi(b)[DG_BI_STATE_OFF]; vl = v3(S(so + DG_BS_LINVEL), S(so + DG_BS_LINVEL + 1), S(so + DG_BS_LINVEL + 2));
      wl = v3(S(so + DG_BS_ANGVEL), S(so + DG_BS_ANGVEL + 1), S(so + DG_BS_ANGVEL + 2)); }
    if (gl >= 0) {
      // ancestors have smaller indices: accumulate along the path by scanning the body's links
      int first = bi(b)[DG_BI_FIRST_LINK];
      // path marking: walk up from gl, then process in increasing order
      unsigned long long path = 0ull; for (int k = gl; k >= 0; k = li(k)[DG_LI_PARENT]) path |= 1ull << (k - first);
      for (int k = first; k <= gl; k++) {
        if (!((path >> (k - first)) & 1ull)) continue;
        int po_off = pll(k)[PLL_POSE]; M3 Rk = LR(po_off); V3 pk = L3(po_off + 6);
        cfp f = lf(k); V3 axw = mul(Rk, v3(f[DG_LF_AXIS], f[DG_LF_AXIS + 1], f[DG_LF_AXIS + 2]));
        float qd = S(li(k)[DG_LI_STATE_OFF] + DG_LS_QD);
        vl = vl + cross(wl, pk - po);  // move the reference point to this link's origin (parent's angular velocity)
        if (li(k)[DG_LI_TYPE] == 0) wl = wl + axw * qd; else vl = vl + axw * qd;
        po = pk;
      }
    }
    w = wl; v = vl + cross(wl, p - po);
  }

  // ------------------------------------------------- articulated-body pass
  DGD int aw(int i) const { return sc.tr_off + sc.ab_stride + i * AW_STRIDE; }
  DGD int ab() const { return sc.tr_off; }

  DGD S6 damping_force(float m, V3 c, const Sym3& Ic, const S6& v, float ka) const {
    float kl = sc.HF[DG_HF_LIN_DAMPING];
    V3 vc = v.l + cross(v.a, c);
    V3 f = vc * (-m * (kl + kl * norm(vc)));
    V3 n = mul(Ic, v.a) * (-(ka + ka * norm(v.a)));
    S6 o = {n + cross(c, f), f};
    return o;
  }
  DGD Sym3 sym6(cfp p) const { Sym3 s = {p[0], p[1], p[2], p[3], p[4], p[5]}; return s; }
  DGD S6 subspace(int gl) const {
    cfp f = lf(gl); V3 ax = v3(f[DG_LF_AXIS], f[DG_LF_AXIS + 1], f[DG_LF_AXIS + 2]);
    S6 s; if (li(gl)[DG_LI_TYPE] == 0) { s.a = ax; s.l = v3(0, 0, 0); } else { s.a = v3(0, 0, 0); s.l = ax; }
    return s;
  }

  // forward dynamics of body b, velocity update, and M^-1 (packed symmetric) into the MINV region
  template <class PROF_T>
  DGD void dynamics(int b, PROF_T& prof, bool skip_minv = false) const {
    cip B = bi(b); cfp Bf = bf(b);
    const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS], so = B[DG_BI_STATE_OFF];
    const bool fx = fixed(b); const float h = sc.h;
    const int nv = plb(b)[PLB_NV], mo = plb(b)[PLB_MINV], nb6 = fx ? 0 : 6;
    const float ka = ang_damping(b);
    M3 R0 = LR(plb(b)[PLB_R0]);
    // ---- pass 1
    S6 v0 = {v3(0, 0, 0), v3(0, 0, 0)};
    if (!fx) {
      V3 vw = v3(S(so + DG_BS_LINVEL), S(so + DG_BS_LINVEL + 1), S(so + DG_BS_LINVEL + 2));
      V3 ww = v3(S(so + DG_BS_ANGVEL), S(so + DG_BS_ANGVEL + 1), S(so + DG_BS_ANGVEL + 2));
      v0.a = tmul(R0, ww); v0.l = tmul(R0, vw);
      V3 c = v3(Bf[DG_BF_COM], Bf[DG_BF_COM + 1], Bf[DG_BF_COM + 2]); Sym3 Ic = sym6(Bf + DG_BF_INERTIA);
      AI I0 = rigid_inertia(Bf[DG_BF_MASS], c, Ic);
      S6 p0 = crf(v0, mul(I0, v0)) - damping_force(Bf[DG_BF_MASS], c, Ic, v0, ka);
      int eo = ext_off(b);
      S6 fx6; fx6.l = tmul(R0, v3(S(eo), S(eo + 1), S(eo + 2))); fx6.a = tmul(R0, v3(S(eo + 3), S(eo + 4), S(eo + 5)));
      LAIset(ab() + AB_IA, I0); L6set(ab() + AB_PA, p0 - fx6);
    }
    L6set(ab() + AB_V, v0);
    for (int i = 0; i < n; i++) {
      int gl = first + i, par = li(gl)[DG_LI_PARENT]; cfp f = lf(gl);
      int lo = li(gl)[DG_LI_STATE_OFF]; float q = S(lo + DG_LS_Q), qd = S(lo + DG_LS_QD);
      M3 Rpc; V3 r; joint_xform(gl, q, Rpc, r); M3 E = transpose(Rpc);
      S6 vp = par < 0 ? v0 : L6(aw(par - first) + AW_V);
      S6 Sx = subspace(gl);
      S6 v = xmotion(E, r, vp) + Sx * qd;
      float lm; V3 c; Sym3 Ic; link_inertia(gl, lm, c, Ic);
      AI I = rigid_inertia(lm, c, Ic);
      S6 pA = crf(v, mul(I, v)) - damping_force(lm, c, Ic, v, ka);
      int o = aw(i);
      LMset(o + AW_E, E); L3set(o + AW_R, r); L6set(o + AW_V, v); L6set(o + AW_PA, pA);
      const int acc = pll(gl)[PLL_IAACC];
      if (acc >= 0) { for (int k = 0; k < 21; k++) L(acc + k) = 0.f; }
    }
    AI carry; int carry_parent = -2;  // contribution of link i+1 to link i, kept in registers along chains
    // ---- pass 2
    for (int i = n - 1; i >= 0; i--) {
      int gl = first + i, par = li(gl)[DG_LI_PARENT]; cfp f = lf(gl); int o = aw(i);
      int lo = li(gl)[DG_LI_STATE_OFF]; float qd = S(lo + DG_LS_QD);
      float tau = S(lo + DG_LS_TORQUE) - f[DG_LF_DAMPING] * qd;
      AI IA; { float lm; V3 lc; Sym3 lI; link_inertia(gl, lm, lc, lI); IA = rigid_inertia(lm, lc, lI); }
      if (carry_parent == gl) {
        IA.I.xx += carry.I.xx; IA.I.xy += carry.I.xy; IA.I.xz += carry.I.xz; IA.I.yy += carry.I.yy; IA.I.yz += carry.I.yz; IA.I.zz += carry.I.zz;
        IA.M.xx += carry.M.xx; IA.M.xy += carry.M.xy; IA.M.xz += carry.M.xz; IA.M.yy += carry.M.yy; IA.M.yz += carry.M.yz; IA.M.zz += carry.M.zz;
#pragma unroll
        for (int k = 0; k < 9; k++) IA.H.m[k] += carry.H.m[k];
      }
      { const int acc = pll(gl)[PLL_IAACC];
        if (acc >= 0) { AI t = LAI(acc);
          IA.I.xx += t.I.xx; IA.I.xy += t.I.xy; IA.I.xz += t.I.xz; IA.I.yy += t.I.yy; IA.I.yz += t.I.yz; IA.I.zz += t.I.zz;
          IA.M.xx += t.M.xx; IA.M.xy += t.M.xy; IA.M.xz += t.M.xz; IA.M.yy += t.M.yy; IA.M.yz += t.M.yz; IA.M.zz += t.M.zz;
#pragma unroll
          for (int k = 0; k < 9; k++) IA.H.m[k] += t.H.m[k]; } }
      S6 Sx = subspace(gl); S6 pA = L6(o + AW_PA); S6 v = L6(o + AW_V);
      S6 U = mul(IA, Sx); float d = dot(Sx, U); float u = tau - dot(Sx, pA); float dinv = 1.0f / d;
      L6set(o + AW_U, U); L(o + AW_D) = d; L(o + AW_UU) = u;
      if (par >= 0 || !fx) {
        // Ia = IA - U U^T / d
        AI Ia = IA;
        Ia.I.xx -= U.a.x * U.a.x * dinv; Ia.I.xy -= U.a.x * U.a.y * dinv; Ia.I.xz -= U.a.x * U.a.z * dinv;
        Ia.I.yy -= U.a.y * U.a.y * dinv; Ia.I.yz -= U.a.y * U.a.z * dinv; Ia.I.zz -= U.a.z * U.a.z * dinv;
        Ia.M.xx -= U.l.x * U.l.x * dinv; Ia.M.xy -= U.l.x * U.l.y * dinv; Ia.M.xz -= U.l.x * U.l.z * dinv;
        Ia.M.yy -= U.l.y * U.l.y * dinv; Ia.M.yz -= U.l.y * U.l.z * dinv; Ia.M.zz -= U.l.z * U.l.z * dinv;
        float ua[3] = {U.a.x, U.a.y, U.a.z}, ul[3] = {U.l.x, U.l.y, U.l.z};
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int c2 = 0; c2 < 3; c2++) Ia.H.m[3 * a + c2] -= ua[a] * ul[c2] * dinv;
        S6 c = crm(v, Sx * qd);
        S6 pa = pA + mul(Ia, c) + U * (u * dinv);
        M3 E = LM(o + AW_E); V3 r = L3(o + AW_R);
        AI Ip = to_parent(Ia, E, r); S6 pf = xforce_to_parent(E, r, pa);
        if (par < 0) { LAIadd(ab() + AB_IA, Ip); L6add(ab() + AB_PA, pf); }
        else {
          L6add(aw(par - first) + AW_PA, pf);
          if (par == gl - 1) { carry = Ip; carry_parent = par; } else LAIadd(pll(par)[PLL_IAACC], Ip);
        }
      }
    }
    // ---- base
    V3 gb = tmul(R0, v3(sc.gx, sc.gy, sc.gz));
    S6 a0; float Lb[21];
    if (fx) { a0.a = v3(0, 0, 0); a0.l = -gb; }
    else {
      AI IA0 = LAI(ab() + AB_IA); ai_to_packed(IA0, Lb); chol6(Lb);
#pragma unroll
      for (int k = 0; k < 21; k++) L(ab() + AB_L + k) = Lb[k];
      S6 p0 = L6(ab() + AB_PA);
      float rhs[6] = {-p0.a.x, -p0.a.y, -p0.a.z, -p0.l.x, -p0.l.y, -p0.l.z}, x[6];
      chol6_solve(Lb, rhs, x);
      a0.a = v3(x[0], x[1], x[2]); a0.l = v3(x[3], x[4], x[5]);
    }
    L6set(ab() + AB_A, a0);
    // ---- pass 3 (+ joint velocity update)
    for (int i = 0; i < n; i++) {
      int gl = first + i, par = li(gl)[DG_LI_PARENT]; int o = aw(i); int lo = li(gl)[DG_LI_STATE_OFF];
      float qd = S(lo + DG_LS_QD);
      M3 E = LM(o + AW_E); V3 r = L3(o + AW_R); S6 Sx = subspace(gl); S6 v = L6(o + AW_V);
      S6 ap = par < 0 ? a0 : L6(aw(par - first) + AW_V);  // parents already hold their acceleration
      S6 a1 = xmotion(E, r, ap) + crm(v, Sx * qd);
      S6 U = L6(o + AW_U); float qdd = (L(o + AW_UU) - dot(U, a1)) / L(o + AW_D);
      L6set(o + AW_V, a1 + Sx * qdd);
      Sset(lo + DG_LS_QD, qd + h * qdd);
    }
    if (!fx) {
      V3 al = a0.l + gb, aa = a0.a;
      V3 acl = al + cross(v0.a, v0.l);
      V3 dvw = mul(R0, acl), dww = mul(R0, aa);
      Sset(so + DG_BS_LINVEL, S(so + DG_BS_LINVEL) + h * dvw.x); Sset(so + DG_BS_LINVEL + 1, S(so + DG_BS_LINVEL + 1) + h * dvw.y);
      Sset(so + DG_BS_LINVEL + 2, S(so + DG_BS_LINVEL + 2) + h * dvw.z);
      Sset(so + DG_BS_ANGVEL, S(so + DG_BS_ANGVEL) + h * dww.x); Sset(so + DG_BS_ANGVEL + 1, S(so + DG_BS_ANGVEL + 1) + h * dww.y);
      Sset(so + DG_BS_ANGVEL + 2, S(so + DG_BS_ANGVEL + 2) + h * dww.z);
    }
    prof.stamp(3 /* PS_ABA */);
    if (skip_minv) return;  // (lane-sliced modes: minv_sliced does the columns with every lane of the env's group)
    // ---- M^-1 by unit impulse responses; column col of generalised coordinates (base 6 first when floating).
    // p (bias) reuses AW_PA, link accelerations reuse AW_V.  M^-1 is symmetric, so only its lower triangle is computed:
    // the response to an impulse on joint j is propagated inward along j's ancestors only and outward over the links
    // 0..j only (every entry to the right of the diagonal is the mirror image of one computed by a later column), and
    // a base column needs nothing but its 6 x 6 block.  Roughly half the work of propagating every column through
    // every link.
    for (int col = 0; col < nv; col++) {
      const int jdof = col - nb6;  // joint index or negative for a base coordinate
      S6 p0 = {v3(0, 0, 0), v3(0, 0, 0)};
      if (jdof >= 0) {
        // inward from the driven joint along its ancestors; every other link carries no bias
        for (int i = 0; i <= jdof; i++) L(aw(i) + AW_UU) = 0.0f;
        for (int i = jdof; i >= 0; ) { const int par = li(first + i)[DG_LI_PARENT]; S6 z = {v3(0, 0, 0), v3(0, 0, 0)}; L6set(aw(i) + AW_PA, z); i = par < 0 ? -1 : par - first; }
        for (int i = jdof; i >= 0; ) {
          int gl = first + i, par = li(gl)[DG_LI_PARENT]; int o = aw(i);
          S6 p = L6(o + AW_PA); S6 Sx = subspace(gl);
          float u = (i == jdof ? 1.0f : 0.0f) - dot(Sx, p);
          L(o + AW_UU) = u;
          if (par >= 0 || !fx) {
            S6 pa = p + L6(o + AW_U) * (u / L(o + AW_D));
            S6 pf = xforce_to_parent(LM(o + AW_E), L3(o + AW_R), pa);
            if (par < 0) p0 = p0 + pf; else L6add(aw(par - first) + AW_PA, pf);
          }
          i = par < 0 ? -1 : par - first;
        }
      }
      S6 a0c = {v3(0, 0, 0), v3(0, 0, 0)};
      if (!fx) {
        float rhs[6] = {-p0.a.x, -p0.a.y, -p0.a.z, -p0.l.x, -p0.l.y, -p0.l.z}, x[6];
        if (jdof < 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) rhs[k] = (k == col) ? 1.0f : 0.0f;
        }
        chol6_solve(Lb, rhs, x);
        a0c.a = v3(x[0], x[1], x[2]); a0c.l = v3(x[3], x[4], x[5]);
#pragma unroll
        for (int k = 0; k < 6; k++) L(mo + col * nv + k) = x[k];
      }
      for (int i = 0; i <= jdof; i++) {  // (none for a base column)
        int gl = first + i, par = li(gl)[DG_LI_PARENT]; int o = aw(i);
        S6 ap = par < 0 ? a0c : L6(aw(par - first) + AW_V);
        S6 a1 = xmotion(LM(o + AW_E), L3(o + AW_R), ap);
        float qdd = (L(o + AW_UU) - dot(L6(o + AW_U), a1)) / L(o + AW_D);
        L6set(o + AW_V, a1 + subspace(gl) * qdd);
        L(mo + col * nv + nb6 + i) = qdd;
      }
    }
    for (int r = 0; r < nv; r++) for (int c = r + 1; c < nv; c++) L(mo + r * nv + c) = L(mo + c * nv + r);  // mirror the lower triangle
    prof.stamp(4 /* PS_MINV */);
  }

  // Register-resident forward dynamics + M^-1 for a fixed-base serial chain of at most N joints (every 6-axis
  // arm): link transforms, velocities, bias forces, U = I^A S and the articulated inertia being propagated all
  // stay in VGPRs; the only LDS traffic is the finished M^-1.  Same recursion as dynamics().
  template <int N, class PROF_T>
  DGD void dynamics_chain(int b, PROF_T& prof) const {
    cip B = bi(b); const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS], mo = plb(b)[PLB_MINV]; const float h = sc.h;
    const M3 R0 = LR(plb(b)[PLB_R0]); const V3 gb = tmul(R0, v3(sc.gx, sc.gy, sc.gz));
    const float ka = ang_damping(b);
    const S6 zero6 = {v3(0.f, 0.f, 0.f), v3(0.f, 0.f, 0.f)};
    M3 E[N]; V3 r[N]; S6 v[N], pA[N], U[N], Sx[N]; float dinv[N], u[N], qd[N];
    S6 vp = zero6;
#pragma unroll
    for (int i = 0; i < N; i++) {
      E[i] = R0; r[i] = v3(0.f, 0.f, 0.f); v[i] = zero6; pA[i] = zero6; U[i] = zero6; Sx[i] = zero6; dinv[i] = 0.f; u[i] = 0.f; qd[i] = 0.f;
      if (i < n) {
        const int gl = first + i, lo = li(gl)[DG_LI_STATE_OFF]; cfp f = lf(gl);
        const float q = S(lo + DG_LS_Q); qd[i] = S(lo + DG_LS_QD);
        M3 Rpc; joint_xform(gl, q, Rpc, r[i]); E[i] = transpose(Rpc); Sx[i] = subspace(gl);
        v[i] = xmotion(E[i], r[i], vp) + Sx[i] * qd[i]; vp = v[i];
        float lm; V3 c; Sym3 Ic; link_inertia(gl, lm, c, Ic); (void)f;
        const AI I = rigid_inertia(lm, c, Ic);
        pA[i] = crf(v[i], mul(I, v[i])) - damping_force(lm, c, Ic, v[i], ka);
      }
    }
    AI carry; bool have = false;
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
      if (i < n) {
        const int gl = first + i, lo = li(gl)[DG_LI_STATE_OFF]; cfp f = lf(gl);
        const float tau = S(lo + DG_LS_TORQUE) - f[DG_LF_DAMPING] * qd[i];
        AI IA; { float lm; V3 lc; Sym3 lI; link_inertia(gl, lm, lc, lI); IA = rigid_inertia(lm, lc, lI); }
        if (have) {
          IA.I.xx += carry.I.xx; IA.I.xy += carry.I.xy; IA.I.xz += carry.I.xz; IA.I.yy += carry.I.yy; IA.I.yz += carry.I.yz; IA.I.zz += carry.I.zz;
          IA.M.xx += carry.M.xx; IA.M.xy += carry.M.xy; IA.M.xz += carry.M.xz; IA.M.yy += carry.M.yy; IA.M.yz += carry.M.yz; IA.M.zz += carry.M.zz;
#pragma unroll
          for (int k = 0; k < 9; k++) IA.H.m[k] += carry.H.m[k];
        }
        U[i] = mul(IA, Sx[i]); const float d = dot(Sx[i], U[i]); dinv[i] = 1.0f / d; u[i] = tau - dot(Sx[i], pA[i]);
        if (i > 0) {
          const float di = dinv[i]; const S6 Ui = U[i];
          AI Ia = IA;
          Ia.I.xx -= Ui.a.x * Ui.a.x * di; Ia.I.xy -= Ui.a.x * Ui.a.y * di; Ia.I.xz -= Ui.a.x * Ui.a.z * di;
          Ia.I.yy -= Ui.a.y * Ui.a.y * di; Ia.I.yz -= Ui.a.y * Ui.a.z * di; Ia.I.zz -= Ui.a.z * Ui.a.z * di;
          Ia.M.xx -= Ui.l.x * Ui.l.x * di; Ia.M.xy -= Ui.l.x * Ui.l.y * di; Ia.M.xz -= Ui.l.x * Ui.l.z * di;
          Ia.M.yy -= Ui.l.y * Ui.l.y * di; Ia.M.yz -= Ui.l.y * Ui.l.z * di; Ia.M.zz -= Ui.l.z * Ui.l.z * di;
          const float ua[3] = {Ui.a.x, Ui.a.y, Ui.a.z}, ul[3] = {Ui.l.x, Ui.l.y, Ui.l.z};
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) Ia.H.m[3 * a + c2] -= ua[a] * ul[c2] * di;
          const S6 c = crm(v[i], Sx[i] * qd[i]);
          const S6 pa = pA[i] + mul(Ia, c) + Ui * (u[i] * di);
          carry = to_parent(Ia, E[i], r[i]); have = true;
          pA[i - 1] = pA[i - 1] + xforce_to_parent(E[i], r[i], pa);
        }
      }
    }
    S6 ap; ap.a = v3(0.f, 0.f, 0.f); ap.l = -gb;
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (i < n) {
        const S6 a1 = xmotion(E[i], r[i], ap) + crm(v[i], Sx[i] * qd[i]);
        const float qdd = (u[i] - dot(U[i], a1)) * dinv[i];
        ap = a1 + Sx[i] * qdd;
        Sset(li(first + i)[DG_LI_STATE_OFF] + DG_LS_QD, qd[i] + h * qdd);
      }
    }
    prof.stamp(3 /* PS_ABA */);
    // M^-1: response to a unit impulse on joint `col`
#pragma unroll
    for (int col = 0; col < N; col++) {
      if (col < n) {
        S6 p[N]; float uu[N];
#pragma unroll
        for (int i = 0; i < N; i++) { p[i] = zero6; uu[i] = 0.f; }
#pragma unroll
        for (int i = col; i >= 0; i--) {
          uu[i] = (i == col ? 1.0f : 0.0f) - dot(Sx[i], p[i]);
          if (i > 0) p[i - 1] = p[i - 1] + xforce_to_parent(E[i], r[i], p[i] + U[i] * (uu[i] * dinv[i]));
        }
        S6 a = zero6;
#pragma unroll
        for (int i = 0; i < N; i++) {
          if (i < n) {
            const S6 a1 = xmotion(E[i], r[i], a);
            const float qdd = (uu[i] - dot(U[i], a1)) * dinv[i];
            a = a1 + Sx[i] * qdd;
            L(mo + col * n + i) = qdd;
          }
        }
      }
    }
    prof.stamp(4 /* PS_MINV */);
  }

  // ------------------------------------------------------------ solver rows
  // generalized velocity of body b dotted with a Jacobian stored at LDS offset jo (length nv)
  // generalised velocity of body b (base twist in base coordinates, then joint rates) into LDS at vo
  DGD void gen_vel_store(int b, int vo) const {
    cip B = bi(b); const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS], so = B[DG_BI_STATE_OFF]; int k = 0;
    if (!fixed(b)) {
      M3 R0 = LR(plb(b)[PLB_R0]);
      L3set(vo, tmul(R0, v3(S(so + DG_BS_ANGVEL), S(so + DG_BS_ANGVEL + 1), S(so + DG_BS_ANGVEL + 2))));
      L3set(vo + 3, tmul(R0, v3(S(so + DG_BS_LINVEL), S(so + DG_BS_LINVEL + 1), S(so + DG_BS_LINVEL + 2)))); k = 6;
    }
    for (int i = 0; i < n; i++) L(vo + k + i) = S(li(first + i)[DG_LI_STATE_OFF] + DG_LS_QD);
  }
  // J . v with v stored by gen_vel_store
  DGD float gen_vel_dot_lds(int jo, int vo, int nv) const {
    float r = 0.f;
    for (int c0 = 0; c0 < nv; c0 += 8) {
      float x[8], y[8];
      _Pragma("unroll") for (int t = 0; t < 8; t++) { x[t] = L(jo + c0 + t); y[t] = L(vo + c0 + t); }
      _Pragma("unroll") for (int t = 0; t < 8; t++) r += (c0 + t < nv) ? x[t] * y[t] : 0.f;
    }
    return r;
  }
  DGD float gen_vel_dot(int b, int jo) const {
    cip B = bi(b); const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS], so = B[DG_BI_STATE_OFF];
    float r = 0.f; int k = 0;
    if (!fixed(b)) {
      M3 R0 = LR(plb(b)[PLB_R0]);
      V3 wb = tmul(R0, v3(S(so + DG_BS_ANGVEL), S(so + DG_BS_ANGVEL + 1), S(so + DG_BS_ANGVEL + 2)));
      V3 vb = tmul(R0, v3(S(so + DG_BS_LINVEL), S(so + DG_BS_LINVEL + 1), S(so + DG_BS_LINVEL + 2)));
      r = L(jo) * wb.x + L(jo + 1) * wb.y + L(jo + 2) * wb.z + L(jo + 3) * vb.x + L(jo + 4) * vb.y + L(jo + 5) * vb.z; k = 6;
    }
    for (int i = 0; i < n; i++) r += L(jo + k + i) * S(li(first + i)[DG_LI_STATE_OFF] + DG_LS_QD);
    return r;
  }
  // Whether minv_sliced can take body b: the free tail of the transient region (behind the body's articulated-body
  // workspace; the contact rows that live there are built after the dynamics) must hold the private scratch of at
  // least two lanes -- bias force, joint impulse and acceleration per link, 13 slots.  Returns the number of slices.
  DGD int minv_slices(int b, int max_slices) const {
    const int n = bi(b)[DG_BI_N_LINKS]; if (n < 1 || n > 32) return 0;
    const int ns = min((sc.tr_slots - sc.ab_stride - n * AW_STRIDE) / (13 * n), max_slices);
    return ns >= 2 ? ns : 0;
  }
  // M^-1 of body b by unit impulse responses (the column loop of dynamics()), the columns shared by the `ns` lanes of the
  // env's group: lane `sl` takes columns sl, sl + ns, ...  The loops over links are wave-uniform (table lookups stay
  // scalar); what a lane's column needs of them is a predicate -- the ancestor mask of its joint for the inward pass,
  // i <= joint for the outward pass -- and each lane keeps bias forces, joint impulses and accelerations in its own
  // scratch.  Same arithmetic per column as the serial loop.  Called by every lane of the wavefront with the Lane of
  // ITS env (the grouping of the sweeps); the articulated inertias (AW_E, AW_R, AW_U, AW_D, AB_L) are read-only here.
  DGD void minv_sliced(int b, int sl, int ns, int group) const {
    cip B = bi(b); const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS];
    const bool fx = fixed(b); const int nv = plb(b)[PLB_NV], mo = plb(b)[PLB_MINV], nb6 = fx ? 0 : 6;
    const int sb = sc.tr_off + sc.ab_stride + n * AW_STRIDE + (sl < ns ? sl : 0) * 13 * n;
    auto sx = [&](int i) { return sb + 13 * i; };  // [bias force 6][joint impulse][acceleration 6] of link i, this lane's
    float Lb[21];
#pragma unroll
    for (int k = 0; k < 21; k++) Lb[k] = fx ? 0.f : L(ab() + AB_L + k);
    for (int c0 = 0; c0 < nv; c0 += ns) {
      const int col = c0 + sl; const bool act = sl < ns && col < nv; const int jdof = act ? col - nb6 : -1;
      const unsigned anc = jdof >= 0 ? (unsigned)sc.AM[first + jdof] : 0u;  // ancestors-or-self of the driven joint, body-local bits
      S6 p0 = {v3(0, 0, 0), v3(0, 0, 0)};
      if (act) for (int i = 0; i < n; i++) { S6 z = {v3(0, 0, 0), v3(0, 0, 0)}; L6set(sx(i), z); L(sx(i) + 6) = 0.0f; }
      for (int i = n - 1; i >= 0; i--) {  // inward along each lane's own path
        const bool on = (anc >> i) & 1u;
        if (!__any(on)) continue;
        const int gl = first + i, par = li(gl)[DG_LI_PARENT], o = aw(i);
        if (on) {
          const S6 p = L6(sx(i)); const S6 Sx = subspace(gl);
          const float u = (i == jdof ? 1.0f : 0.0f) - dot(Sx, p);
          L(sx(i) + 6) = u;
          if (par >= 0 || !fx) {
            const S6 pa = p + L6(o + AW_U) * (u / L(o + AW_D));
            const S6 pf = xforce_to_parent(LM(o + AW_E), L3(o + AW_R), pa);
            if (par < 0) p0 = p0 + pf; else L6add(sx(par - first), pf);
          }
        }
      }
      S6 a0c = {v3(0, 0, 0), v3(0, 0, 0)};
      if (!fx) {
        float rhs[6] = {-p0.a.x, -p0.a.y, -p0.a.z, -p0.l.x, -p0.l.y, -p0.l.z}, x[6];
        if (jdof < 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) rhs[k] = (k == col) ? 1.0f : 0.0f;
        }
        chol6_solve(Lb, rhs, x);
        a0c.a = v3(x[0], x[1], x[2]); a0c.l = v3(x[3], x[4], x[5]);
        if (act) {
#pragma unroll
          for (int k = 0; k < 6; k++) L(mo + col * nv + k) = x[k];
        }
      }
      for (int i = 0; i < n; i++) {  // outward over the links up to each lane's joint (none for a base column)
        const bool on = act && i <= jdof;
        if (!__any(on)) break;
        const int gl = first + i, par = li(gl)[DG_LI_PARENT], o = aw(i);
        if (on) {
          const S6 ap = par < 0 ? a0c : L6(sx(par - first) + 7);
          const S6 a1 = xmotion(LM(o + AW_E), L3(o + AW_R), ap);
          const float qdd = (L(sx(i) + 6) - dot(L6(o + AW_U), a1)) / L(o + AW_D);
          L6set(sx(i) + 7, a1 + subspace(gl) * qdd);
          L(mo + col * nv + nb6 + i) = qdd;
        }
      }
    }
    for (int r = sl; r < nv; r += group) for (int c = r + 1; c < nv; c++) L(mo + r * nv + c) = L(mo + c * nv + r);  // mirror the lower triangle
  }

  // Jacobian (into jo) and response M^-1 J^T (into ro) of body b for a unit force along world direction
  // dir at world point p on link gl (-1 base) -- TORQUE: for a unit torque about dir on that link.  Returns J M^-1 J^T.
  template <bool TORQUE = false>
  DGD float point_row(int b, int gl, V3 p, V3 dir, int jo, int ro) const {
    cip B = bi(b); const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS];
    const int nv = plb(b)[PLB_NV], mo = plb(b)[PLB_MINV]; int k0 = 0;
    for (int k = 0; k < nv; k++) L(jo + k) = 0.f;
    if (!fixed(b)) {
      M3 R0 = LR(plb(b)[PLB_R0]); V3 ja = TORQUE ? tmul(R0, dir) : tmul(R0, cross(p - base_pos(b), dir)), jl = TORQUE ? v3(0.f, 0.f, 0.f) : tmul(R0, dir);
      L3set(jo, ja); L3set(jo + 3, jl); k0 = 6;
    }
    for (int k = gl; k >= 0; k = li(k)[DG_LI_PARENT]) {
      int po = pll(k)[PLL_POSE]; M3 Rk = LR(po); V3 pk = L3(po + 6); cfp f = lf(k);
      V3 axw = mul(Rk, v3(f[DG_LF_AXIS], f[DG_LF_AXIS + 1], f[DG_LF_AXIS + 2]));
      if (TORQUE) L(jo + k0 + (k - first)) = li(k)[DG_LI_TYPE] == 0 ? dot(axw, dir) : 0.f;
      else L(jo + k0 + (k - first)) = li(k)[DG_LI_TYPE] == 0 ? dot(axw, cross(p - pk, dir)) : dot(axw, dir);
    }
    (void)n;
    // Response M^-1 J^T, eight entries at a time: J is sparse (base + the chain above link gl) and M^-1 symmetric,
    // so the response is a short sum of (contiguous) M^-1 rows scaled by the non-zero Jacobian entries.  Reads run
    // <= 7 slots past a row; those are allocated and masked out at the store.
    for (int c0 = 0; c0 < nv; c0 += 8) {
      float acc[8];
      _Pragma("unroll") for (int t = 0; t < 8; t++) acc[t] = 0.f;
      if (!fixed(b)) {
        float Jb[6], Mb[6][8];
        _Pragma("unroll") for (int j = 0; j < 6; j++) { Jb[j] = L(jo + j); _Pragma("unroll") for (int t = 0; t < 8; t++) Mb[j][t] = L(mo + j * nv + c0 + t); }
        _Pragma("unroll") for (int j = 0; j < 6; j++) _Pragma("unroll") for (int t = 0; t < 8; t++) acc[t] += Mb[j][t] * Jb[j];
      }
      for (int k = gl; k >= 0; k = li(k)[DG_LI_PARENT]) {
        const int j = k0 + (k - first); const float Jj = L(jo + j); float Mr[8];
        _Pragma("unroll") for (int t = 0; t < 8; t++) Mr[t] = L(mo + j * nv + c0 + t);
        _Pragma("unroll") for (int t = 0; t < 8; t++) acc[t] += Mr[t] * Jj;
      }
      _Pragma("unroll") for (int t = 0; t < 8; t++) if (c0 + t < nv) L(ro + c0 + t) = acc[t];
    }
    float diag = 0.f;
    for (int c0 = 0; c0 < nv; c0 += 8) {
      float x[8], y[8];
      _Pragma("unroll") for (int t = 0; t < 8; t++) { x[t] = L(ro + c0 + t); y[t] = L(jo + c0 + t); }
      _Pragma("unroll") for (int t = 0; t < 8; t++) diag += (c0 + t < nv) ? x[t] * y[t] : 0.f;
    }
    return diag;
  }
};

}  // namespace dg
