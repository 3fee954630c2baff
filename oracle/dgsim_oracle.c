/* dgsim_oracle.c -- CPU oracle for the DIYGym batched step path (plain C; fp64 unless -DDGO_REAL=float).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker for the HIP path.  The product
 * (diy_gym_amd/) never links, loads or calls it.
 *
 * PARITY UNPINNED.  The arithmetic of the reference's step path lives in the
 * third-party `pybullet` wheel (reference requirements.txt:3, unpinned, source
 * not vendored, not installable here).  This file restates (a) the reference's
 * own Python semantics -- every addon hook and the DIYGym.step/reset sequence,
 * each function citing the reference file:line it follows -- and (b) the
 * published algorithms pybullet's multibody path is built on (Featherstone's
 * articulated-body algorithm; sequential-impulse / projected Gauss-Seidel on
 * velocity-level rows; damped-least-squares IK), with Bullet's parameter
 * values taken FROM RECOLLECTION (tagged [R] below, listed in DESIGN.md).
 * There are no golden vectors in the reference, so what pins
 * this oracle is tests/test_oracle_kat.py: analytic known-answer tests
 * (free fall, resting contact, pendulum period and energy, mass-matrix
 * symmetry, ABA vs. Lagrangian real pendulum, FK of the UR5 at the
 * reference's rest pose computed independently in numpy, IK fixed point) and
 * the reference's one behavioural test (tests/test_environment.py:23-40).
 *
 * One env at a time, arrays-of-structs, straightforward dense 6x6 spatial
 * algebra: deliberately NOT the layout or the code of the HIP kernels.
 */
#include "dgsim_oracle.h"

#include <math.h>
#include <tgmath.h> /* sqrt / sin / atan2 ... of a `real` pick the float versions in the fp32 build */
#undef I /* (complex.h's imaginary unit, pulled in by tgmath.h) */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/diygym_scene.h"

#define MAXL 24     /* moving links per body */
#define MAXV (6 + MAXL)
#define MAXC 32     /* contacts per env */
#define MAXROWS (3 * MAXL * 8 + 3 * MAXC)

#define HUGE_R ((real)(sizeof(real) == 8 ? 1e300 : 3.0e38))   /* "no bound" */
#define TINY_R ((real)(sizeof(real) == 8 ? 1e-300 : 1.0e-37)) /* "parallel" */
int32_t dgo_real_bytes(void) { return (int32_t)sizeof(real); }

static char g_err[512];
static void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
const char* dgo_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ math */
typedef struct { real x, y, z; } v3;
typedef struct { real m[3][3]; } m3;
typedef struct { real x, y, z, w; } qt;

static v3 V(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 vscale(v3 a, real s) { return V(a.x * s, a.y * s, a.z * s); }
static real vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static real vnorm(v3 a) { return sqrt(vdot(a, a)); }
static v3 mv(const m3* A, v3 b) {
  return V(A->m[0][0] * b.x + A->m[0][1] * b.y + A->m[0][2] * b.z, A->m[1][0] * b.x + A->m[1][1] * b.y + A->m[1][2] * b.z,
           A->m[2][0] * b.x + A->m[2][1] * b.y + A->m[2][2] * b.z);
}
static v3 mtv(const m3* A, v3 b) { /* A^T b */
  return V(A->m[0][0] * b.x + A->m[1][0] * b.y + A->m[2][0] * b.z, A->m[0][1] * b.x + A->m[1][1] * b.y + A->m[2][1] * b.z,
           A->m[0][2] * b.x + A->m[1][2] * b.y + A->m[2][2] * b.z);
}
static m3 mmul(const m3* A, const m3* B) {
  m3 C;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) C.m[i][j] = A->m[i][0] * B->m[0][j] + A->m[i][1] * B->m[1][j] + A->m[i][2] * B->m[2][j];
  return C;
}
static m3 mident(void) { m3 I = {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; return I; }
static m3 mfrom9(const real* p) {
  m3 A;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A.m[i][j] = p[3 * i + j];
  return A;
}
static m3 msym6(const real* p) { /* xx xy xz yy yz zz */
  m3 A = {{{p[0], p[1], p[2]}, {p[1], p[3], p[4]}, {p[2], p[4], p[5]}}};
  return A;
}
static qt qmul(qt a, qt b) {
  qt r = {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
  return r;
}
static qt qconj(qt a) { qt r = {-a.x, -a.y, -a.z, a.w}; return r; }
static qt qnormalize(qt a) {
  real n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  qt r = {a.x / n, a.y / n, a.z / n, a.w / n};
  return r;
}
static m3 qmat(qt q) {
  real n = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = n > 0 ? 2.0 / n : 0.0;
  real xs = q.x * s, ys = q.y * s, zs = q.z * s;
  real wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys,
         yz = q.y * zs, zz = q.z * zs;
  m3 R = {{{1 - (yy + zz), xy - wz, xz + wy}, {xy + wz, 1 - (xx + zz), yz - wx}, {xz - wy, yz + wx, 1 - (xx + yy)}}};
  return R;
}
static qt qfrom_mat(const m3* R) {
  real tr = R->m[0][0] + R->m[1][1] + R->m[2][2];
  qt q;
  if (tr > 0) {
    real s = sqrt(tr + 1.0) * 2.0;
    q.x = (R->m[2][1] - R->m[1][2]) / s; q.y = (R->m[0][2] - R->m[2][0]) / s; q.z = (R->m[1][0] - R->m[0][1]) / s; q.w = 0.25 * s;
  } else if (R->m[0][0] > R->m[1][1] && R->m[0][0] > R->m[2][2]) {
    real s = sqrt(1.0 + R->m[0][0] - R->m[1][1] - R->m[2][2]) * 2.0;
    q.x = 0.25 * s; q.y = (R->m[0][1] + R->m[1][0]) / s; q.z = (R->m[0][2] + R->m[2][0]) / s; q.w = (R->m[2][1] - R->m[1][2]) / s;
  } else if (R->m[1][1] > R->m[2][2]) {
    real s = sqrt(1.0 + R->m[1][1] - R->m[0][0] - R->m[2][2]) * 2.0;
    q.x = (R->m[0][1] + R->m[1][0]) / s; q.y = 0.25 * s; q.z = (R->m[1][2] + R->m[2][1]) / s; q.w = (R->m[0][2] - R->m[2][0]) / s;
  } else {
    real s = sqrt(1.0 + R->m[2][2] - R->m[0][0] - R->m[1][1]) * 2.0;
    q.x = (R->m[0][2] + R->m[2][0]) / s; q.y = (R->m[1][2] + R->m[2][1]) / s; q.z = 0.25 * s; q.w = (R->m[1][0] - R->m[0][1]) / s;
  }
  return qnormalize(q);
}
/* pybullet getQuaternionFromEuler: fixed-axis XYZ = Rz(yaw) Ry(pitch) Rx(roll) [R] */
static qt qfrom_euler(real r, real p, real y) {
  real cr = cos(r * 0.5), sr = sin(r * 0.5), cp = cos(p * 0.5), sp = sin(p * 0.5), cy = cos(y * 0.5), sy = sin(y * 0.5);
  qt q = {sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
  return q;
}
/* pybullet getEulerFromQuaternion (btQuaternion::getEulerZYX branches) [R] */
static v3 euler_from_q(qt q) {
  real sarg = -2.0 * (q.x * q.z - q.w * q.y);
  if (sarg <= -0.99999) return V(0.0, -0.5 * M_PI, 2.0 * atan2(q.x, -q.y));
  if (sarg >= 0.99999) return V(0.0, 0.5 * M_PI, 2.0 * atan2(-q.x, q.y));
  real sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, sqw = q.w * q.w;
  return V(atan2(2.0 * (q.y * q.z + q.w * q.x), sqw - sqx - sqy + sqz), asin(sarg),
           atan2(2.0 * (q.x * q.y + q.w * q.z), sqw + sqx - sqy - sqz));
}
static m3 rot_axis(v3 a, real th) { /* Rodrigues, |a| = 1 */
  real c = cos(th), s = sin(th), t = 1 - c;
  m3 R = {{{t * a.x * a.x + c, t * a.x * a.y - s * a.z, t * a.x * a.z + s * a.y},
           {t * a.x * a.y + s * a.z, t * a.y * a.y + c, t * a.y * a.z - s * a.x},
           {t * a.x * a.z - s * a.y, t * a.y * a.z + s * a.x, t * a.z * a.z + c}}};
  return R;
}

/* spatial vectors: [angular(3); linear(3)] */
typedef struct { real v[6]; } s6;
typedef struct { real m[6][6]; } m6;
static v3 ang(const s6* a) { return V(a->v[0], a->v[1], a->v[2]); }
static v3 lin(const s6* a) { return V(a->v[3], a->v[4], a->v[5]); }
static s6 mk6(v3 a, v3 l) { s6 r = {{a.x, a.y, a.z, l.x, l.y, l.z}}; return r; }
static s6 s6add(s6 a, s6 b) { for (int i = 0; i < 6; i++) a.v[i] += b.v[i]; return a; }
static s6 s6scale(s6 a, real s) { for (int i = 0; i < 6; i++) a.v[i] *= s; return a; }
static real s6dot(const s6* a, const s6* b) { real s = 0; for (int i = 0; i < 6; i++) s += a->v[i] * b->v[i]; return s; }
static s6 m6v(const m6* A, const s6* x) {
  s6 r;
  for (int i = 0; i < 6; i++) { real s = 0; for (int j = 0; j < 6; j++) s += A->m[i][j] * x->v[j]; r.v[i] = s; }
  return r;
}
/* motion transform parent -> child:  E = rotation parent->child coords, r = child origin in parent coords */
static s6 xmotion(const m3* E, v3 r, const s6* v) {
  v3 w = ang(v), l = lin(v);
  return mk6(mv(E, w), mv(E, vsub(l, vcross(r, w))));
}
/* force transform child -> parent (transpose of the motion transform) */
static s6 xforce_to_parent(const m3* E, v3 r, const s6* f) {
  v3 n = mtv(E, ang(f)), l = mtv(E, lin(f));
  return mk6(vadd(n, vcross(r, l)), l);
}
static m6 xmat(const m3* E, v3 r) { /* 6x6 of xmotion */
  m6 X; memset(&X, 0, sizeof X);
  m3 rx = {{{0, -r.z, r.y}, {r.z, 0, -r.x}, {-r.y, r.x, 0}}};
  m3 Erx = mmul(E, &rx);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { X.m[i][j] = E->m[i][j]; X.m[3 + i][3 + j] = E->m[i][j]; X.m[3 + i][j] = -Erx.m[i][j]; }
  return X;
}
static s6 crm(const s6* v, const s6* s) { /* v x s (motion) */
  v3 w = ang(v), l = lin(v), sw = ang(s), sl = lin(s);
  return mk6(vcross(w, sw), vadd(vcross(w, sl), vcross(l, sw)));
}
static s6 crf(const s6* v, const s6* f) { /* v x* f (force) */
  v3 w = ang(v), l = lin(v), n = ang(f), fl = lin(f);
  return mk6(vadd(vcross(w, n), vcross(l, fl)), vcross(w, fl));
}
static m6 rigid_inertia(real m, v3 c, const m3* Ic) {
  m6 I; memset(&I, 0, sizeof I);
  m3 cx = {{{0, -c.z, c.y}, {c.z, 0, -c.x}, {-c.y, c.x, 0}}};
  real cc = vdot(c, c);
  real cv[3] = {c.x, c.y, c.z};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      I.m[i][j] = Ic->m[i][j] + m * ((i == j ? cc : 0.0) - cv[i] * cv[j]);
      I.m[i][3 + j] = m * cx.m[i][j];
      I.m[3 + i][j] = -m * cx.m[i][j];
      I.m[3 + i][3 + j] = (i == j) ? m : 0.0;
    }
  return I;
}
/* solve A x = b for SPD 6x6 (Cholesky); returns 0 on failure */
static int spd_solve(int n, const real* A, const real* b, real* x) {
  real L[MAXV * MAXV];
  if (n > MAXV) return 0;
  memset(L, 0, sizeof L);
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      real s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      if (i == j) { if (s <= 0) return 0; L[i * n + i] = sqrt(s); } else L[i * n + j] = s / L[j * n + j];
    }
  real y[MAXV];
  for (int i = 0; i < n; i++) { real s = b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { real s = y[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
  return 1;
}

/* counter-based RNG shared (as a spec) with the device: splitmix64 finaliser,
 * 24-bit mantissa so the value is exact in fp32 and fp64 alike */
static uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31;
  return z;
}
static real rng_uniform(uint64_t seed, uint64_t env, uint64_t episode, uint64_t op, uint64_t comp) {
  uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (env + 1));
  z = mix64(z ^ (episode * 0xD1342543DE82EF95ULL + op * 0x2545F4914F6CDD1DULL + comp + 1));
  return (real)(z >> 40) * (1.0 / 16777216.0);
}

/* ------------------------------------------------------------- the world */
typedef struct {
  int32_t* I; real* F; int64_t ni, nf;
  int nb, nl, nfr, nsh, npairs, nops;
  int act_dim, obs_dim, rew_dim, term_dim, substeps, iters, max_steps, hot_start, ik_iters, state_dim;
  int addon_off, max_contacts, rew_mode, term_mode, n_term_groups, warm_off, ncons;
  const int32_t* KI; const real* KF; /* fixed constraints (DG_KI_*, DG_KF_*) */
  const int32_t *BI, *LI, *FI, *SI, *PI, *OI, *IL;
  const real *BF, *LF, *FF, *SF, *PF, *OF, *FL;
  real h; v3 g;
} Scene;

typedef struct {
  int body_a, link_a, body_b, link_b; /* link = global link index or -1 (base) */
  v3 p; /* world contact point (midway) */
  v3 n; /* world normal, from B towards A */
  real dist; /* signed distance (negative = penetration) */
  real mu;
  int shape_a, shape_b;      /* global shape indices */
  int key;                   /* DG_CONTACT_KEY(candidate pair, feature): identity of the contact from substep to substep */
  v3 t1, t2; real imp[3];  /* friction directions and the solved impulses (normal, t1, t2), filled in after the sweeps */
} Contact;

struct dgo_world {
  Scene sc;
  int B;
  uint64_t seed; int64_t env_base;
  real* state; /* [B][state_dim] */
  real* mcfg;  /* [nl][DG_MC_STRIDE] */
  int* last_contacts; int* last_iters;
  Contact* last_cs; /* [B][MAXC]: contacts and impulses of each env's most recent substep (force_torque_sensor) */
};

/* per-body workspace for one env */
typedef struct {
  int n; int first; int fixed;
  v3 p0; qt q0; m3 R0; /* base link frame in world */
  /* per link */
  m3 E[MAXL]; v3 r[MAXL];         /* parent->child motion transform */
  m3 Rw[MAXL]; v3 pw[MAXL];       /* link frame in world */
  s6 S[MAXL], v[MAXL], c[MAXL], pA[MAXL], U[MAXL], a[MAXL];
  m6 IA[MAXL]; real d[MAXL], u[MAXL], qdd[MAXL];
  s6 v0, pA0, a0; m6 IA0;
  int parent[MAXL]; /* local index, -1 base */
  real dv[MAXV]; /* solver velocity change: base(6, base coords) + joints */
} BodyWS;

/* per-thread scratch that outlives the call (the solver rows of one substep are ~0.7 MB: a malloc of that size is an mmap /
 * munmap pair plus page faults per substep, serialised between the OpenMP threads by the kernel).  Never freed. */
static void* scratch(int slot, size_t bytes) {
  static _Thread_local void* buf[4]; static _Thread_local size_t cap[4];
  if (cap[slot] < bytes) { free(buf[slot]); buf[slot] = malloc(bytes); cap[slot] = bytes; }
  return buf[slot];
}
static real* env_state(dgo_world* w, int e) { return w->state + (size_t)e * w->sc.state_dim; }
static const int32_t* body_i(const Scene* s, int b) { return s->BI + b * DG_BI_STRIDE; }
static const real* body_f(const Scene* s, int b) { return s->BF + b * DG_BF_STRIDE; }
static const int32_t* link_i(const Scene* s, int l) { return s->LI + l * DG_LI_STRIDE; }
static const real* link_f(const Scene* s, int l) { return s->LF + l * DG_LF_STRIDE; }
static int body_fixed(const Scene* s, int b) { return body_i(s, b)[DG_BI_FLAGS] & DG_BODY_FIXED; }
static real* body_ext(const Scene* s, real* st, int b) {
  return st + body_i(s, b)[DG_BI_STATE_OFF] + (body_fixed(s, b) ? DG_BS_FIXED_END : DG_BS_FLOAT_END);
}

static int parse_scene(Scene* s, const int32_t* I, int64_t ni, const double* F64, int64_t nf) {
  if (ni < DG_H_INT_COUNT || I[DG_H_MAGIC] != DG_MAGIC) { set_err("bad scene magic"); return 0; }
  if (I[DG_H_VERSION] != DG_VERSION) { set_err("scene version %d, expected %d", I[DG_H_VERSION], DG_VERSION); return 0; }
  s->I = (int32_t*)malloc(sizeof(int32_t) * (size_t)ni); memcpy(s->I, I, sizeof(int32_t) * (size_t)ni);
  s->F = (real*)malloc(sizeof(real) * (size_t)nf); for (int64_t k = 0; k < nf; k++) s->F[k] = (real)F64[k];
  s->ni = ni; s->nf = nf; I = s->I; const real* F = s->F;
  s->nb = I[DG_H_N_BODIES]; s->nl = I[DG_H_N_LINKS]; s->nfr = I[DG_H_N_FRAMES]; s->nsh = I[DG_H_N_SHAPES];
  s->npairs = I[DG_H_N_PAIRS]; s->nops = I[DG_H_N_OPS];
  s->act_dim = I[DG_H_ACT_DIM]; s->obs_dim = I[DG_H_OBS_DIM]; s->rew_dim = I[DG_H_REW_DIM]; s->term_dim = I[DG_H_TERM_DIM];
  s->substeps = I[DG_H_SUBSTEPS]; s->iters = I[DG_H_SOLVER_ITERS]; s->max_steps = I[DG_H_MAX_EPISODE_STEPS];
  s->hot_start = I[DG_H_HOT_START]; s->ik_iters = I[DG_H_IK_ITERS]; s->state_dim = I[DG_H_STATE_DIM];
  s->addon_off = I[DG_H_ADDON_STATE_OFF]; s->max_contacts = I[DG_H_MAX_CONTACTS];
  s->warm_off = I[DG_H_WARM_OFF];
  s->ncons = I[DG_H_N_CONSTRAINTS]; s->KI = I + I[DG_H_OFF_CONS_I]; s->KF = F + I[DG_H_OFF_CONS_F];
  if (s->ncons > DG_MAX_CONSTRAINTS) { set_err("%d fixed constraints > %d", s->ncons, DG_MAX_CONSTRAINTS); return 0; }
  s->rew_mode = I[DG_H_REW_MODE]; s->term_mode = I[DG_H_TERM_MODE]; s->n_term_groups = I[DG_H_N_TERM_GROUPS];
  s->BI = I + I[DG_H_OFF_BODY_I]; s->LI = I + I[DG_H_OFF_LINK_I]; s->FI = I + I[DG_H_OFF_FRAME_I];
  s->SI = I + I[DG_H_OFF_SHAPE_I]; s->PI = I + I[DG_H_OFF_PAIR_I]; s->OI = I + I[DG_H_OFF_OP_I]; s->IL = I + I[DG_H_OFF_ILIST];
  s->BF = F + I[DG_H_OFF_BODY_F]; s->LF = F + I[DG_H_OFF_LINK_F]; s->FF = F + I[DG_H_OFF_FRAME_F];
  s->SF = F + I[DG_H_OFF_SHAPE_F]; s->PF = F + I[DG_H_OFF_POINT_F]; s->OF = F + I[DG_H_OFF_OP_F]; s->FL = F + I[DG_H_OFF_FLIST];
  s->h = F[DG_HF_DT]; s->g = V(F[DG_HF_GRAV_X], F[DG_HF_GRAV_Y], F[DG_HF_GRAV_Z]);
  if (s->max_contacts > MAXC) { set_err("max_contacts %d > %d", s->max_contacts, MAXC); return 0; }
  for (int b = 0; b < s->nb; b++)
    if (body_i(s, b)[DG_BI_N_LINKS] > MAXL) { set_err("body %d has more than %d links", b, MAXL); return 0; }
  return 1;
}

dgo_world* dgo_create(const int32_t* idata, int64_t n_i, const double* fdata, int64_t n_f, int32_t num_envs, uint64_t seed,
                      int64_t env_index_base) {
  dgo_world* w = (dgo_world*)calloc(1, sizeof *w);
  if (!parse_scene(&w->sc, idata, n_i, fdata, n_f)) { free(w); return NULL; }
  w->B = num_envs; w->seed = seed; w->env_base = env_index_base;
  w->state = (real*)calloc((size_t)num_envs * w->sc.state_dim, sizeof(real));
  w->mcfg = (real*)calloc((size_t)(w->sc.nl > 0 ? w->sc.nl : 1) * DG_MC_STRIDE, sizeof(real));
  w->last_contacts = (int*)calloc(num_envs, sizeof(int));
  w->last_iters = (int*)calloc(num_envs, sizeof(int));
  w->last_cs = (Contact*)calloc((size_t)num_envs * MAXC, sizeof(Contact));
  /* every movable joint starts with pybullet's default velocity motor: target 0,
   * kd 1, kp 0, fixed impulse budget per substep [R] */
  for (int l = 0; l < w->sc.nl; l++) {
    w->mcfg[l * DG_MC_STRIDE + DG_MC_KP] = 0.0;
    w->mcfg[l * DG_MC_STRIDE + DG_MC_KD] = 1.0;
    w->mcfg[l * DG_MC_STRIDE + DG_MC_MAX_IMPULSE_SCALE] = -w->sc.F[DG_HF_DEFAULT_MOTOR_IMPULSE];
  }
  for (int op = 0; op < w->sc.nops; op++) { /* admittance_controller.py:34: VELOCITY_CONTROL with forces = 0 at construction */
    const int32_t* oi = w->sc.OI + op * DG_OI_STRIDE;
    if (oi[DG_OI_CODE] == DG_OP_ADMITTANCE) for (int k = 0; k < oi[DG_OI_N]; k++) w->mcfg[w->sc.IL[oi[DG_OI_ILIST] + k] * DG_MC_STRIDE + DG_MC_MAX_IMPULSE_SCALE] = 0.0;
  }
  /* initial state = load pose (reference model.py:68), joints at zero */
  for (int e = 0; e < num_envs; e++) {
    real* st = env_state(w, e);
    for (int b = 0; b < w->sc.nb; b++) {
      if (body_i(&w->sc, b)[DG_BI_FLAGS] & DG_BODY_FROZEN) continue;
      real* bs = st + body_i(&w->sc, b)[DG_BI_STATE_OFF];
      const real* bf = body_f(&w->sc, b);
      for (int k = 0; k < 3; k++) bs[DG_BS_POS + k] = bf[DG_BF_INIT_POS + k];
      for (int k = 0; k < 4; k++) bs[DG_BS_QUAT + k] = bf[DG_BF_INIT_QUAT + k];
    }
    for (int op = 0; op < w->sc.nops; op++) { /* dynamics_randomizer state before its first draw: URDF masses, default damping */
      const int32_t* oi = w->sc.OI + op * DG_OI_STRIDE;
      if (oi[DG_OI_CODE] == DG_OP_RANDOMIZE_COLOR) { /* the configured colour, flat, until the first draw */
        real* tx = st + w->sc.addon_off + oi[DG_OI_STATE_OFF];
        for (int k = 0; k < 3; k++) tx[DG_TX_A + k] = tx[DG_TX_B + k] = body_f(&w->sc, oi[DG_OI_BODY])[DG_BF_COLOR + k];
        tx[DG_TX_FREQ] = 1.0; tx[DG_TX_KIND] = DG_TEX_FLAT;
        continue;
      }
      if (oi[DG_OI_CODE] != DG_OP_RANDOMIZE_DYNAMICS) continue;
      real* ps = st + w->sc.addon_off + oi[DG_OI_STATE_OFF];
      for (int k = 0; k < oi[DG_OI_N]; k++) ps[k] = 1.0;
      ps[oi[DG_OI_N]] = w->sc.F[DG_HF_ANG_DAMPING];
    }
  }
  return w;
}
void dgo_destroy(dgo_world* w) {
  if (!w) return;
  free(w->sc.I); free(w->sc.F); free(w->state); free(w->mcfg); free(w->last_contacts); free(w->last_iters); free(w->last_cs); free(w);
}
int32_t dgo_state_dim(const dgo_world* w) { return w->sc.state_dim; }
real* dgo_state(dgo_world* w) { return w->state; }
real* dgo_motor_cfg(dgo_world* w) { return w->mcfg; }
int32_t dgo_last_contact_count(const dgo_world* w, int32_t env) { return w->last_contacts[env]; }
int32_t dgo_last_iterations(const dgo_world* w, int32_t env) { return w->last_iters[env]; }
/* contact k of env's most recent substep: [point 3, normal 3 (from B towards A), signed distance, normal impulse] */
int32_t dgo_last_contact(const dgo_world* w, int32_t env, int32_t k, real* out8) {
  if (env < 0 || env >= w->B || k < 0 || k >= w->last_contacts[env]) return 0;
  const Contact* c = &w->last_cs[(size_t)env * MAXC + k];
  out8[0] = c->p.x; out8[1] = c->p.y; out8[2] = c->p.z; out8[3] = c->n.x; out8[4] = c->n.y; out8[5] = c->n.z; out8[6] = c->dist; out8[7] = c->imp[0];
  return 1;
}

/* --------------------------------------------------------- kinematics */
static void body_kinematics(const Scene* s, const real* st, int b, BodyWS* ws, const real* q_override) {
  const int32_t* bi = body_i(s, b);
  const real* bs = (bi[DG_BI_FLAGS] & DG_BODY_FROZEN) ? body_f(s, b) + DG_BF_INIT_POS : st + bi[DG_BI_STATE_OFF]; /* pos3 quat4 either way */
  ws->n = bi[DG_BI_N_LINKS]; ws->first = bi[DG_BI_FIRST_LINK]; ws->fixed = bi[DG_BI_FLAGS] & DG_BODY_FIXED;
  ws->p0 = V(bs[0], bs[1], bs[2]);
  qt q0 = {bs[3], bs[4], bs[5], bs[6]}; ws->q0 = q0; ws->R0 = qmat(q0);
  for (int i = 0; i < ws->n; i++) {
    int gl = ws->first + i;
    const int32_t* li = link_i(s, gl); const real* lf = link_f(s, gl);
    int par = li[DG_LI_PARENT]; ws->parent[i] = par < 0 ? -1 : par - ws->first;
    real q = q_override ? q_override[i] : st[li[DG_LI_STATE_OFF] + DG_LS_Q];
    m3 RT = mfrom9(lf + DG_LF_ROT); v3 pT = V(lf[DG_LF_POS], lf[DG_LF_POS + 1], lf[DG_LF_POS + 2]);
    v3 ax = V(lf[DG_LF_AXIS], lf[DG_LF_AXIS + 1], lf[DG_LF_AXIS + 2]);
    m3 Rpc; v3 r;
    if (li[DG_LI_TYPE] == 0) { m3 Rq = rot_axis(ax, q); Rpc = mmul(&RT, &Rq); r = pT; ws->S[i] = mk6(ax, V(0, 0, 0)); }
    else { Rpc = RT; r = vadd(pT, mv(&RT, vscale(ax, q))); ws->S[i] = mk6(V(0, 0, 0), ax); }
    /* E = Rpc^T */
    for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) ws->E[i].m[a][c] = Rpc.m[c][a];
    ws->r[i] = r;
    const m3* Rp = ws->parent[i] < 0 ? &ws->R0 : &ws->Rw[ws->parent[i]];
    v3 pp = ws->parent[i] < 0 ? ws->p0 : ws->pw[ws->parent[i]];
    ws->Rw[i] = mmul(Rp, &Rpc);
    ws->pw[i] = vadd(pp, mv(Rp, r));
  }
}
/* link (-1 = base) frame in world */
static void link_world(const BodyWS* ws, int local_link, m3* R, v3* p) {
  if (local_link < 0) { *R = ws->R0; *p = ws->p0; } else { *R = ws->Rw[local_link]; *p = ws->pw[local_link]; }
}

/* velocities (spatial, link coords) from state */
static void body_velocities(const Scene* s, const real* st, int b, BodyWS* ws) {
  const real* bs = st + (ws->fixed ? 0 : body_i(s, b)[DG_BI_STATE_OFF]);
  if (ws->fixed) memset(&ws->v0, 0, sizeof ws->v0);
  else {
    v3 vw = V(bs[DG_BS_LINVEL], bs[DG_BS_LINVEL + 1], bs[DG_BS_LINVEL + 2]);
    v3 ww = V(bs[DG_BS_ANGVEL], bs[DG_BS_ANGVEL + 1], bs[DG_BS_ANGVEL + 2]);
    ws->v0 = mk6(mtv(&ws->R0, ww), mtv(&ws->R0, vw));
  }
  for (int i = 0; i < ws->n; i++) {
    real qd = st[link_i(s, ws->first + i)[DG_LI_STATE_OFF] + DG_LS_QD];
    const s6* vp = ws->parent[i] < 0 ? &ws->v0 : &ws->v[ws->parent[i]];
    s6 vJ = s6scale(ws->S[i], qd);
    ws->v[i] = s6add(xmotion(&ws->E[i], ws->r[i], vp), vJ);
    ws->c[i] = crm(&ws->v[i], &vJ);
  }
}

/* Bullet's per-link "global" damping [R]: force -m v_com (k + k|v_com|), torque
 * -I_c w (k + k|w|), returned as a spatial force about the link origin */
static s6 damping_force(real m, v3 c, const m3* Ic, const s6* v, real kl, real ka) {
  v3 w = ang(v), vo = lin(v);
  v3 vc = vadd(vo, vcross(w, c));
  v3 f = vscale(vc, -m * (kl + kl * vnorm(vc)));
  v3 n = vscale(mv(Ic, w), -(ka + ka * vnorm(w)));
  return mk6(vadd(n, vcross(c, f)), f);
}

/* articulated-body algorithm, passes 1-3 (Featherstone, RBDA ch. 7).  tau = joint
 * torques (damping, torque-control).  Leaves IA, U, d in ws for impulse responses. */
/* per-env link mass: the URDF mass times the dynamics_randomizer's scale for this link, if it has one (the inertia
 * tensor is scaled with the mass) */
static real link_mass_scale(const Scene* s, const real* st, int gl) {
  int o = link_i(s, gl)[DG_LI_MASS_SCALE]; return o >= 0 ? st[o] : 1.0;
}
static void body_aba(const Scene* s, const real* st, int b, BodyWS* ws) {
  const real* bf = body_f(s, b);
  real kl = s->F[DG_HF_LIN_DAMPING], ka = s->F[DG_HF_ANG_DAMPING];
  if (body_i(s, b)[DG_BI_DYN_OFF] >= 0) ka = st[body_i(s, b)[DG_BI_DYN_OFF]]; /* changeDynamics(angularDamping=) per env */
  /* pass 1: inertias and bias forces */
  for (int i = 0; i < ws->n; i++) {
    const real* lf = link_f(s, ws->first + i); const real ms = link_mass_scale(s, st, ws->first + i);
    v3 c = V(lf[DG_LF_COM], lf[DG_LF_COM + 1], lf[DG_LF_COM + 2]); m3 Ic = msym6(lf + DG_LF_INERTIA);
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) Ic.m[r][cc] *= ms;
    ws->IA[i] = rigid_inertia(lf[DG_LF_MASS] * ms, c, &Ic);
    s6 Iv = m6v(&ws->IA[i], &ws->v[i]);
    ws->pA[i] = crf(&ws->v[i], &Iv);
    s6 fd = damping_force(lf[DG_LF_MASS] * ms, c, &Ic, &ws->v[i], kl, ka);
    for (int k = 0; k < 6; k++) ws->pA[i].v[k] -= fd.v[k];
  }
  if (!ws->fixed) {
    v3 c = V(bf[DG_BF_COM], bf[DG_BF_COM + 1], bf[DG_BF_COM + 2]); m3 Ic = msym6(bf + DG_BF_INERTIA);
    ws->IA0 = rigid_inertia(bf[DG_BF_MASS], c, &Ic);
    s6 Iv = m6v(&ws->IA0, &ws->v0);
    ws->pA0 = crf(&ws->v0, &Iv);
    s6 fd = damping_force(bf[DG_BF_MASS], c, &Ic, &ws->v0, kl, ka);
    /* external wrench (world, about the base origin) -> base coords */
    const real* ex = body_ext(s, (real*)st, b);
    v3 fe = mtv(&ws->R0, V(ex[0], ex[1], ex[2])), ne = mtv(&ws->R0, V(ex[3], ex[4], ex[5]));
    s6 fx = mk6(ne, fe);
    for (int k = 0; k < 6; k++) ws->pA0.v[k] -= fd.v[k] + fx.v[k];
  }
  /* pass 2: articulated inertias, leaves to root */
  for (int i = ws->n - 1; i >= 0; i--) {
    const int32_t* li = link_i(s, ws->first + i); const real* lf = link_f(s, ws->first + i);
    real qd = st[li[DG_LI_STATE_OFF] + DG_LS_QD];
    real tau = st[li[DG_LI_STATE_OFF] + DG_LS_TORQUE] - lf[DG_LF_DAMPING] * qd; /* Bullet joint damping [R] */
    ws->U[i] = m6v(&ws->IA[i], &ws->S[i]);
    ws->d[i] = s6dot(&ws->S[i], &ws->U[i]);
    ws->u[i] = tau - s6dot(&ws->S[i], &ws->pA[i]);
    m6 Ia = ws->IA[i];
    for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) Ia.m[a][c] -= ws->U[i].v[a] * ws->U[i].v[c] / ws->d[i];
    s6 Iac = m6v(&Ia, &ws->c[i]);
    s6 pa;
    for (int k = 0; k < 6; k++) pa.v[k] = ws->pA[i].v[k] + Iac.v[k] + ws->U[i].v[k] * ws->u[i] / ws->d[i];
    m6 X = xmat(&ws->E[i], ws->r[i]);
    /* parent += X^T Ia X ; X^T pa */
    m6 T; for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) { real t = 0; for (int k = 0; k < 6; k++) t += Ia.m[a][k] * X.m[k][c]; T.m[a][c] = t; }
    m6* Ip = ws->parent[i] < 0 ? &ws->IA0 : &ws->IA[ws->parent[i]];
    s6* pp = ws->parent[i] < 0 ? &ws->pA0 : &ws->pA[ws->parent[i]];
    if (ws->parent[i] >= 0 || !ws->fixed) {
      for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) { real t = 0; for (int k = 0; k < 6; k++) t += X.m[k][a] * T.m[k][c]; Ip->m[a][c] += t; }
      s6 pf = xforce_to_parent(&ws->E[i], ws->r[i], &pa);
      for (int k = 0; k < 6; k++) pp->v[k] += pf.v[k];
    }
  }
  /* pass 3: accelerations.  Gravity enters as a base acceleration of -g (fixed
   * base) or is added afterwards (floating base, RBDA 9.4). */
  v3 gb = mtv(&ws->R0, s->g);
  if (ws->fixed) ws->a0 = mk6(V(0, 0, 0), vscale(gb, -1.0));
  else {
    real rhs[6], x[6];
    for (int k = 0; k < 6; k++) rhs[k] = -ws->pA0.v[k];
    if (!spd_solve(6, &ws->IA0.m[0][0], rhs, x)) memset(x, 0, sizeof x);
    for (int k = 0; k < 6; k++) ws->a0.v[k] = x[k];
  }
  for (int i = 0; i < ws->n; i++) {
    const s6* ap = ws->parent[i] < 0 ? &ws->a0 : &ws->a[ws->parent[i]];
    s6 a1 = s6add(xmotion(&ws->E[i], ws->r[i], ap), ws->c[i]);
    ws->qdd[i] = (ws->u[i] - s6dot(&ws->U[i], &a1)) / ws->d[i];
    ws->a[i] = s6add(a1, s6scale(ws->S[i], ws->qdd[i]));
  }
  if (!ws->fixed) { ws->a0.v[3] += gb.x; ws->a0.v[4] += gb.y; ws->a0.v[5] += gb.z; }
}

/* impulse response: generalized velocity change caused by a unit of (spatial force
 * f at local link `lk` (-1 = base), expressed in that link's coords) and/or a
 * joint impulse on dof `dof` (-1 none).  Same recursion as ABA with zero
 * velocity and zero gravity (Featherstone RBDA 7.3 / Bullet's
 * calcAccelerationDeltasMultiDof).  Also returns the row Jacobian J with
 * J . gen_velocity == f . v_link  (+ qd[dof]). */
static void body_response(const BodyWS* ws, int lk, const s6* f, int dof, real* J, real* dv) {
  s6 p[MAXL], p0; real u[MAXL];
  memset(p, 0, sizeof(s6) * (size_t)(ws->n > 0 ? ws->n : 1)); memset(&p0, 0, sizeof p0);
  for (int k = 0; k < 6 + ws->n; k++) { J[k] = 0; dv[k] = 0; }
  s6 fj[MAXL], fj0; memset(fj, 0, sizeof(s6) * (size_t)(ws->n > 0 ? ws->n : 1)); memset(&fj0, 0, sizeof fj0);
  if (f) { if (lk < 0) { fj0 = *f; for (int k = 0; k < 6; k++) p0.v[k] = -f->v[k]; } else { fj[lk] = *f; for (int k = 0; k < 6; k++) p[lk].v[k] = -f->v[k]; } }
  for (int i = ws->n - 1; i >= 0; i--) {
    /* Jacobian: pure force propagation */
    J[6 + i] = s6dot(&ws->S[i], &fj[i]) + (i == dof ? 1.0 : 0.0);
    s6 fjp = xforce_to_parent(&ws->E[i], ws->r[i], &fj[i]);
    if (ws->parent[i] < 0) fj0 = s6add(fj0, fjp); else fj[ws->parent[i]] = s6add(fj[ws->parent[i]], fjp);
    /* response */
    u[i] = (i == dof ? 1.0 : 0.0) - s6dot(&ws->S[i], &p[i]);
    s6 pa; for (int k = 0; k < 6; k++) pa.v[k] = p[i].v[k] + ws->U[i].v[k] * u[i] / ws->d[i];
    s6 pf = xforce_to_parent(&ws->E[i], ws->r[i], &pa);
    if (ws->parent[i] < 0) p0 = s6add(p0, pf); else p[ws->parent[i]] = s6add(p[ws->parent[i]], pf);
  }
  s6 a[MAXL], a0; memset(&a0, 0, sizeof a0);
  if (!ws->fixed) {
    for (int k = 0; k < 6; k++) J[k] = fj0.v[k];
    real rhs[6], x[6]; for (int k = 0; k < 6; k++) rhs[k] = -p0.v[k];
    if (!spd_solve(6, &ws->IA0.m[0][0], rhs, x)) memset(x, 0, sizeof x);
    for (int k = 0; k < 6; k++) { a0.v[k] = x[k]; dv[k] = x[k]; }
  }
  for (int i = 0; i < ws->n; i++) {
    const s6* ap = ws->parent[i] < 0 ? &a0 : &a[ws->parent[i]];
    s6 a1 = xmotion(&ws->E[i], ws->r[i], ap);
    real qdd = (u[i] - s6dot(&ws->U[i], &a1)) / ws->d[i];
    a[i] = s6add(a1, s6scale(ws->S[i], qdd));
    dv[6 + i] = qdd;
  }
}

/* ----------------------------------------------------------- collision */
typedef struct { int id, type, body, llink /* local */, glink; m3 R; v3 p; m3 Rl; v3 pl; /* frame the hull points live in */ const real* prm; real mu; int poff, npts; } WShape;

static void shape_world(const Scene* s, const BodyWS* wsb, int sh, WShape* o) {
  const int32_t* si = s->SI + sh * DG_SI_STRIDE; const real* sf = s->SF + sh * DG_SF_STRIDE;
  o->id = sh; o->type = si[DG_SI_TYPE]; o->body = si[DG_SI_BODY]; o->glink = si[DG_SI_LINK];
  const BodyWS* ws = &wsb[o->body];
  o->llink = o->glink < 0 ? -1 : o->glink - ws->first;
  m3 Rl; v3 pl; link_world(ws, o->llink, &Rl, &pl);
  if (si[DG_SI_FLAGS] & DG_SHAPE_WORLD) { Rl = mident(); pl = V(0, 0, 0); } /* frozen body: stored in world coordinates */
  o->Rl = Rl; o->pl = pl;
  m3 Rs = mfrom9(sf + DG_SF_ROT);
  o->R = mmul(&Rl, &Rs); o->p = vadd(pl, mv(&Rl, V(sf[DG_SF_POS], sf[DG_SF_POS + 1], sf[DG_SF_POS + 2])));
  o->prm = sf + DG_SF_PARAMS; o->mu = sf[DG_SF_FRICTION]; o->poff = si[DG_SI_POINT_OFF]; o->npts = si[DG_SI_N_POINTS];
}
static void add_contact(Contact* cs, int* nc, int maxc, const WShape* a, const WShape* b, v3 pa, v3 pb, v3 n, real dist, int key) {
  if (*nc >= maxc) return;
  Contact* c = &cs[(*nc)++]; c->key = key;
  c->body_a = a->body; c->link_a = a->llink; c->body_b = b->body; c->link_b = b->llink; c->shape_a = a->id; c->shape_b = b->id;
  c->imp[0] = c->imp[1] = c->imp[2] = 0.0;
  c->p = vscale(vadd(pa, pb), 0.5); c->n = n; c->dist = dist; c->mu = a->mu * b->mu; /* Bullet combines friction by product [R] */
}
/* sphere (centre c, radius r) against box shape bx: returns 1 and contact data when dist < margin */
static int sphere_box(v3 c, real r, const WShape* bx, real margin, v3* pa, v3* pb, v3* n, real* dist) {
  v3 lc = mtv(&bx->R, vsub(c, bx->p));
  real h[3] = {bx->prm[0], bx->prm[1], bx->prm[2]}, l[3] = {lc.x, lc.y, lc.z}, cl[3];
  int inside = 1;
  for (int k = 0; k < 3; k++) { cl[k] = l[k] < -h[k] ? -h[k] : (l[k] > h[k] ? h[k] : l[k]); if (cl[k] != l[k]) inside = 0; }
  v3 nl; real d;
  if (!inside) {
    v3 df = V(l[0] - cl[0], l[1] - cl[1], l[2] - cl[2]); d = vnorm(df); nl = vscale(df, 1.0 / d);
  } else { /* centre inside: leave through the nearest face */
    int best = 0; real bd = HUGE_R; real sg = 1;
    for (int k = 0; k < 3; k++) { real dp = h[k] - l[k], dm = l[k] + h[k]; if (dp < bd) { bd = dp; best = k; sg = 1; } if (dm < bd) { bd = dm; best = k; sg = -1; } }
    real nn[3] = {0, 0, 0}; nn[best] = sg; nl = V(nn[0], nn[1], nn[2]); d = -bd; cl[best] = sg * h[best];
  }
  if (d - r >= margin) return 0;
  *n = mv(&bx->R, nl);
  *pb = vadd(bx->p, mv(&bx->R, V(cl[0], cl[1], cl[2])));
  *pa = vsub(c, vscale(*n, r));
  *dist = d - r;
  return 1;
}
static void seg_ends(const WShape* c, v3* e0, v3* e1) {
  v3 ax = V(c->R.m[0][2], c->R.m[1][2], c->R.m[2][2]);
  *e0 = vsub(c->p, vscale(ax, c->prm[1])); *e1 = vadd(c->p, vscale(ax, c->prm[1]));
}
static v3 closest_on_seg(v3 a, v3 b, v3 p) {
  v3 ab = vsub(b, a); real den = vdot(ab, ab);
  real t = den > 0 ? vdot(vsub(p, a), ab) / den : 0.0; t = t < 0 ? 0 : (t > 1 ? 1 : t);
  return vadd(a, vscale(ab, t));
}
/* closest points between segments p1-q1, p2-q2 (Ericson, Real-Time Collision Detection 5.1.9) */
static void seg_seg(v3 p1, v3 q1, v3 p2, v3 q2, v3* c1, v3* c2) {
  v3 d1 = vsub(q1, p1), d2 = vsub(q2, p2), r = vsub(p1, p2);
  real a = vdot(d1, d1), e = vdot(d2, d2), f = vdot(d2, r), sN, tN; const real eps = 1e-12;
  if (a <= eps && e <= eps) { *c1 = p1; *c2 = p2; return; }
  if (a <= eps) { sN = 0; tN = f / e; tN = tN < 0 ? 0 : (tN > 1 ? 1 : tN); }
  else {
    real c = vdot(d1, r);
    if (e <= eps) { tN = 0; sN = -c / a; sN = sN < 0 ? 0 : (sN > 1 ? 1 : sN); }
    else {
      real b = vdot(d1, d2), den = a * e - b * b;
      sN = den > eps ? (b * f - c * e) / den : 0.0; sN = sN < 0 ? 0 : (sN > 1 ? 1 : sN);
      tN = (b * sN + f) / e;
      if (tN < 0) { tN = 0; sN = -c / a; sN = sN < 0 ? 0 : (sN > 1 ? 1 : sN); }
      else if (tN > 1) { tN = 1; sN = (b - c) / a; sN = sN < 0 ? 0 : (sN > 1 ? 1 : sN); }
    }
  }
  *c1 = vadd(p1, vscale(d1, sN)); *c2 = vadd(p2, vscale(d2, tN));
}
static int sphere_sphere(v3 ca, real ra, v3 cb, real rb, real margin, v3* pa, v3* pb, v3* n, real* dist) {
  v3 d = vsub(ca, cb); real len = vnorm(d);
  if (len - ra - rb >= margin) return 0;
  *n = len > 1e-12 ? vscale(d, 1.0 / len) : V(0, 0, 1);
  *pa = vsub(ca, vscale(*n, ra)); *pb = vadd(cb, vscale(*n, rb)); *dist = len - ra - rb;
  return 1;
}


/* ---- convex hull against convex hull (DG_HF_HULL_CONTACTS) ----------------------------------------------------------------
 * What pybullet does with two URDF collision meshes (reference model.py:65 loadURDF; e.g. data/ur5/ur5_robot.urdf collision
 * <mesh> elements): Bullet collides their convex hulls -- btConvexConvexAlgorithm, GJK closest points on the hulls WITHOUT their
 * collision margin, the margins (gUrdfDefaultCollisionMargin = 0.001 per shape [R]) subtracted from the distance afterwards, and
 * an expanding-polytope search for the penetration depth once the margin-free hulls themselves overlap [R].  Restated here on
 * the (thinned) hull points of the scene tables:
 *   C = A - B (Minkowski difference), support s_C(d) = s_A(d) - s_B(-d);
 *   GJK: closest point v of C to the origin, simplex of <= 4 support points with the Voronoi-region tests of Ericson, Real-Time
 *        Collision Detection 5.1.5 / 5.1.6; distance |v|, normal v / |v| (from B to A), witness points from the barycentric
 *        weights; gives up early once v . w / |v| (a lower bound of the distance) exceeds `max_dist`;
 *   EPA: when the origin is inside C or nearer than HH_SWITCH to it -- an expanding polytope inside C, started from a
 *        tetrahedron of four support points (the one GJK ended with, if it ended inside one; else two opposite support points,
 *        the one farthest from their line and the one farthest from their plane), its faces kept with SIGNED plane
 *        distances of the origin (so a start polytope that does not yet hold the origin is fine): the face with the smallest
 *        one is pushed out to its support point until it is a face of C; depth = that distance, normal = minus its normal.
 * Coordinates are relative to A's frame origin (fp32 build: world coordinates of ~1 m would cost three digits).
 * The HIP kernels run the same steps (dg_solver.h hull_hull); both are checked against a brute-force construction of C with
 * scipy's ConvexHull in tests/test_oracle_kat.py. */
#define HH_GJK_ITERS 32
#define HH_EPA_ITERS 24
#define HH_EPA_MAXV (4 + HH_EPA_ITERS)
#define HH_EPA_MAXF (2 * HH_EPA_MAXV)
#define HH_EPA_MAXE 96  /* horizon edges held at a time (the device keeps them in per-lane scratch) */
#define HH_FACE_PTS 8  /* coplanar support points gathered for the witness points of a polytope face */
#define HH_SWITCH ((real)1e-4)  /* the origin nearer to C than this: the polytope search decides (v / |v| is noise there) */
typedef struct { const real *pa, *pb; int na, nb; m3 RA, RB; v3 tBA; } HullPair;
typedef struct { v3 w, a, b; int ia, ib; } HV;
static int hh_argmax(const real* p, int n, v3 d) {
  int bi = 0; real best = -HUGE_R;
  for (int k = 0; k < n; k++) { real sd = p[3 * k] * d.x + p[3 * k + 1] * d.y + p[3 * k + 2] * d.z; if (sd > best) { best = sd; bi = k; } }
  return bi;
}
static HV hh_support(const HullPair* h, v3 d) {
  HV o; o.ia = hh_argmax(h->pa, h->na, mtv(&h->RA, d)); o.ib = hh_argmax(h->pb, h->nb, mtv(&h->RB, vscale(d, -1.0)));
  o.a = mv(&h->RA, V(h->pa[3 * o.ia], h->pa[3 * o.ia + 1], h->pa[3 * o.ia + 2]));
  o.b = vadd(mv(&h->RB, V(h->pb[3 * o.ib], h->pb[3 * o.ib + 1], h->pb[3 * o.ib + 2])), h->tBA);
  o.w = vsub(o.a, o.b); return o;
}
/* barycentric weights of the point of triangle (a, b, c) closest to the origin (Ericson 5.1.5) */
static void hh_closest_tri(v3 a, v3 b, v3 c, real* l) {
  v3 ab = vsub(b, a), ac = vsub(c, a);
  real d1 = -vdot(ab, a), d2 = -vdot(ac, a);
  l[0] = l[1] = l[2] = 0;
  if (d1 <= 0 && d2 <= 0) { l[0] = 1; return; }
  real d3 = -vdot(ab, b), d4 = -vdot(ac, b);
  if (d3 >= 0 && d4 <= d3) { l[1] = 1; return; }
  real vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { real t = d1 / (d1 - d3); l[0] = 1 - t; l[1] = t; return; }
  real d5 = -vdot(ab, c), d6 = -vdot(ac, c);
  if (d6 >= 0 && d5 <= d6) { l[2] = 1; return; }
  real vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { real t = d2 / (d2 - d6); l[0] = 1 - t; l[2] = t; return; }
  real va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { real t = (d4 - d3) / ((d4 - d3) + (d5 - d6)); l[1] = 1 - t; l[2] = t; return; }
  real den = 1.0 / (va + vb + vc); l[1] = vb * den; l[2] = vc * den; l[0] = 1 - l[1] - l[2];
}
/* point of the simplex (n vertices) closest to the origin: weights l[0..n); returns 1 when the origin is inside a tetrahedron */
static int hh_closest_simplex(const HV* sx, int n, real* l) {
  for (int k = 0; k < 4; k++) l[k] = 0;
  if (n == 1) { l[0] = 1; return 0; }
  if (n == 2) {
    v3 ab = vsub(sx[1].w, sx[0].w); real den = vdot(ab, ab), t = den > 0 ? -vdot(sx[0].w, ab) / den : 0.0;
    t = t < 0 ? 0 : (t > 1 ? 1 : t); l[0] = 1 - t; l[1] = t; return 0;
  }
  if (n == 3) { hh_closest_tri(sx[0].w, sx[1].w, sx[2].w, l); return 0; }
  /* tetrahedron (Ericson 5.1.6): the faces that have the origin on their outer side (a flat tetrahedron: every face) */
  static const int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
  real best = HUGE_R; int inside = 1;
  for (int f = 0; f < 4; f++) {
    v3 a = sx[F[f][0]].w, b = sx[F[f][1]].w, c = sx[F[f][2]].w, d = sx[F[f][3]].w;
    v3 nn = vcross(vsub(b, a), vsub(c, a));
    real sp = -vdot(a, nn), sd = vdot(vsub(d, a), nn), scale = vdot(nn, nn) * vdot(vsub(d, a), vsub(d, a));
    int flat = sd * sd <= (real)1e-10 * scale, outside = flat || sp * sd < 0;
    if (!outside) continue;
    inside = 0;
    real lt[3]; hh_closest_tri(a, b, c, lt);
    v3 q = vadd(vadd(vscale(a, lt[0]), vscale(b, lt[1])), vscale(c, lt[2])); real qq = vdot(q, q);
    if (qq < best) { best = qq; for (int k = 0; k < 4; k++) l[k] = 0; l[F[f][0]] = lt[0]; l[F[f][1]] = lt[1]; l[F[f][2]] = lt[2]; }
  }
  return inside;
}
typedef struct { int v[3]; v3 n; real d; int alive; } HF;
static void hh_face(HF* f, const HV* vs, int i, int j, int k, v3 g) {
  f->v[0] = i; f->v[1] = j; f->v[2] = k; f->alive = 1;
  v3 nn = vcross(vsub(vs[j].w, vs[i].w), vsub(vs[k].w, vs[i].w)); real len = vnorm(nn);
  if (!(len > (real)1e-12)) { f->n = V(0, 0, 1); f->d = HUGE_R; return; }  /* a sliver: kept for the topology, never the closest */
  nn = vscale(nn, 1.0 / len);
  if (vdot(nn, vsub(vs[i].w, g)) < 0) { nn = vscale(nn, -1.0); f->v[1] = k; f->v[2] = j; }  /* outward: away from the interior point g */
  f->n = nn; f->d = vdot(nn, vs[i].w);
}
/* signed distance of the origin from the boundary of C along its nearest face (> 0: inside, the penetration depth), that
 * face's outward normal and the witness points of the origin's projection onto it */
static real hh_epa(const HullPair* h, v3 seed, const HV* start, v3* n_out, v3* pa, v3* pb, int* iters_out) {
  HV vs[HH_EPA_MAXV]; HF fs[HH_EPA_MAXF]; int nv = 0, nf = 0;
  /* start: the tetrahedron GJK ended with when it found the origin inside one (it already has a face near the origin: a shallow
   * overlap then takes 4 further support points at the median instead of 6); else one built here */
  if (start) { for (int k = 0; k < 4; k++) vs[k] = start[k]; } else {
  /* a tetrahedron of C: two opposite support points, the one farthest from their line, the one farthest from their plane */
  v3 d0 = vdot(seed, seed) > (real)1e-12 ? vscale(seed, 1.0 / vnorm(seed)) : V(1, 0, 0);
  vs[0] = hh_support(h, d0); vs[1] = hh_support(h, vscale(d0, -1.0));
  v3 e = vsub(vs[1].w, vs[0].w);
  v3 ax = fabs(e.x) <= fabs(e.y) && fabs(e.x) <= fabs(e.z) ? V(1, 0, 0) : (fabs(e.y) <= fabs(e.z) ? V(0, 1, 0) : V(0, 0, 1));
  v3 d1 = vcross(e, ax); d1 = vscale(d1, 1.0 / (vnorm(d1) + TINY_R));
  HV c1 = hh_support(h, d1), c2 = hh_support(h, vscale(d1, -1.0));
  vs[2] = fabs(vdot(vsub(c1.w, vs[0].w), d1)) >= fabs(vdot(vsub(c2.w, vs[0].w), d1)) ? c1 : c2;
  v3 nn = vcross(e, vsub(vs[2].w, vs[0].w)); nn = vscale(nn, 1.0 / (vnorm(nn) + TINY_R));
  c1 = hh_support(h, nn); c2 = hh_support(h, vscale(nn, -1.0));
  vs[3] = fabs(vdot(vsub(c1.w, vs[0].w), nn)) >= fabs(vdot(vsub(c2.w, vs[0].w), nn)) ? c1 : c2;
  }
  nv = 4;
  v3 g = vscale(vadd(vadd(vs[0].w, vs[1].w), vadd(vs[2].w, vs[3].w)), 0.25);
  hh_face(&fs[0], vs, 0, 1, 2, g); hh_face(&fs[1], vs, 0, 1, 3, g); hh_face(&fs[2], vs, 0, 2, 3, g); hh_face(&fs[3], vs, 1, 2, 3, g); nf = 4;
  int best = 0;
  for (int it = 0; it < HH_EPA_ITERS; it++) {
    best = -1; real bd = HUGE_R;
    for (int f = 0; f < nf; f++) if (fs[f].alive && fs[f].d < bd) { bd = fs[f].d; best = f; }
    if (best < 0) { best = 0; break; }
    HV w = hh_support(h, fs[best].n);
    if (vdot(w.w, fs[best].n) - fs[best].d <= (real)1e-6 || nv >= HH_EPA_MAXV) break;  /* the face lies on the boundary of C */
    int dup = 0; for (int k = 0; k < nv; k++) if (vs[k].ia == w.ia && vs[k].ib == w.ib) dup = 1;
    if (dup) break;
    vs[nv] = w;
    /* faces that see the new point go; the edges that belonged to exactly one of them are the horizon */
    int ed[HH_EPA_MAXE], ne = 0, full = 0;  /* an edge = its two vertex indices, lower one first: i | j << 8 */
    for (int f = 0; f < nf; f++) {
      if (!fs[f].alive || fs[f].d >= HUGE_R) continue;
      if (!(vdot(fs[f].n, w.w) - fs[f].d > (real)1e-9) && f != best) continue;
      fs[f].alive = 0;
      for (int q = 0; q < 3; q++) {
        int i = fs[f].v[q], j = fs[f].v[(q + 1) % 3], key = i < j ? (i | (j << 8)) : (j | (i << 8)), found = -1;
        for (int t = 0; t < ne; t++) if (ed[t] == key) { found = t; break; }
        if (found >= 0) { ed[found] = ed[ne - 1]; ne--; } else if (ne < HH_EPA_MAXE) ed[ne++] = key; else full = 1;
      }
    }
    /* (a sliver is never removed: its edges never enter the horizon and the surface stays closed) */
    for (int t = 0; t < ne; t++) {
      int slot = -1; for (int f = 0; f < nf; f++) if (!fs[f].alive) { slot = f; break; }
      if (slot < 0) { if (nf >= HH_EPA_MAXF) break; slot = nf++; }
      hh_face(&fs[slot], vs, ed[t] & 255, ed[t] >> 8, nv, g);
    }
    if (full) { nv++; break; }
    nv++;
  }
  if (iters_out) *iters_out = nv - 4;
  const HF* f = &fs[best];
  /* Witness points: the origin's projection p onto the face's plane, as a combination of support points lying IN that plane.
   * The triangle found is only part of C's face there -- a parallelogram when two edges cross, a polygon when a face of one hull
   * rests on the other -- and p may lie in another part of it: while p is outside every triangle of the coplanar points found so
   * far, the support point of a direction tilted from the normal towards p (1e-3 rad) is the face's corner on that side. */
  const v3 p = vscale(f->n, f->d);
  HV pl[HH_FACE_PTS]; int np = 3; pl[0] = vs[f->v[0]]; pl[1] = vs[f->v[1]]; pl[2] = vs[f->v[2]];
  int bi = 0, bj = 1, bk = 2; real bl[3] = {1, 0, 0};
  for (int round = 0; ; round++) {
    real bq = HUGE_R; v3 qbest = p;
    for (int i = 0; i < np; i++) for (int j = i + 1; j < np; j++) for (int k = j + 1; k < np; k++) {
      real l[3]; hh_closest_tri(vsub(pl[i].w, p), vsub(pl[j].w, p), vsub(pl[k].w, p), l);
      v3 q = vadd(vadd(vscale(vsub(pl[i].w, p), l[0]), vscale(vsub(pl[j].w, p), l[1])), vscale(vsub(pl[k].w, p), l[2])); real qq = vdot(q, q);
      if (qq < bq) { bq = qq; qbest = q; bi = i; bj = j; bk = k; bl[0] = l[0]; bl[1] = l[1]; bl[2] = l[2]; }
    }
    if (bq <= (real)1e-12 || np >= HH_FACE_PTS || round >= HH_FACE_PTS) break;
    HV w = hh_support(h, vadd(f->n, vscale(qbest, -(real)1e-3 / sqrt(bq))));  /* (qbest = nearest point - p: towards p is -qbest) */
    if (f->d - vdot(w.w, f->n) > (real)1e-4) break;  /* not in the plane (0.1 mm): the face ends before p */
    int dup = 0; for (int k = 0; k < np; k++) if (pl[k].ia == w.ia && pl[k].ib == w.ib) dup = 1;
    if (dup) break;
    pl[np++] = w;
  }
  *pa = vadd(vadd(vscale(pl[bi].a, bl[0]), vscale(pl[bj].a, bl[1])), vscale(pl[bk].a, bl[2]));
  *pb = vadd(vadd(vscale(pl[bi].b, bl[0]), vscale(pl[bj].b, bl[1])), vscale(pl[bk].b, bl[2]));
  *n_out = f->n; return f->d;
}
/* signed distance of hull A from hull B (< 0: they overlap by that much), unit normal from B towards A, witness points (world,
 * relative to A's frame origin).  Returns 0 -- nothing else set -- once the distance is known to exceed max_dist. */
static long long g_hh_calls, g_hh_far, g_hh_epa, g_hh_gjk_iters;  /* diagnostic tallies (dgo_hull_tallies; not thread-safe: serial builds only) */
static int hull_hull(const HullPair* h, v3 seed, real max_dist, v3* pa, v3* pb, v3* n, real* dist, int* stats) {
  g_hh_calls++;
  HV sx[4]; int ns = 0; real l[4] = {0, 0, 0, 0};
  v3 v = vdot(seed, seed) > (real)1e-12 ? seed : V(1, 0, 0); real vv = HUGE_R; int inside = 0, it;
  for (it = 0; it < HH_GJK_ITERS; it++) {
    HV w = hh_support(h, vscale(v, -1.0));
    if (ns > 0) {
      real vw = vdot(v, w.w), vn = sqrt(vv);
      if (vw > max_dist * vn) { if (stats) stats[0] = it + 1; g_hh_far++; g_hh_gjk_iters += it + 1; return 0; }  /* a separating plane farther than anyone asks */
      int dup = 0; for (int k = 0; k < ns; k++) if (sx[k].ia == w.ia && sx[k].ib == w.ib) dup = 1;
      if (dup || vv - vw <= (real)1e-6 * vv + (real)1e-7 * vn) break;  /* no support point nearer along v: v is the closest point */
    }
    sx[ns] = w;
    real ln[4]; int in = hh_closest_simplex(sx, ns + 1, ln);
    if (in) { inside = 1; break; }
    v3 nv = V(0, 0, 0); for (int k = 0; k <= ns; k++) nv = vadd(nv, vscale(sx[k].w, ln[k]));
    real nvv = vdot(nv, nv);
    if (ns > 0 && !(nvv < vv)) break;  /* (rounding: no progress -- keep the previous simplex) */
    int m = 0; for (int k = 0; k <= ns; k++) if (ln[k] > 0) { sx[m] = sx[k]; l[m] = ln[k]; m++; }
    ns = m; v = nv; vv = nvv;
    if (vv <= HH_SWITCH * HH_SWITCH) break;
  }
  if (stats) { stats[0] = it + 1; stats[1] = inside || vv <= HH_SWITCH * HH_SWITCH; }
  g_hh_gjk_iters += it + 1;
  if (inside || vv <= HH_SWITCH * HH_SWITCH) {
    g_hh_epa++;
    v3 nf; real d = hh_epa(h, seed, inside ? sx : NULL, &nf, pa, pb, stats ? &stats[2] : NULL);
    *n = vscale(nf, -1.0); *dist = -d; return 1;
  }
  real vn = sqrt(vv); *n = vscale(v, 1.0 / vn); *dist = vn;
  *pa = V(0, 0, 0); *pb = V(0, 0, 0);
  for (int k = 0; k < ns; k++) { *pa = vadd(*pa, vscale(sx[k].a, l[k])); *pb = vadd(*pb, vscale(sx[k].b, l[k])); }
  return 1;
}
/* diagnostic: [calls of the hull-hull routine, of them left early as too far apart, decided by the polytope search, GJK iterations in all] since the last reset */
void dgo_hull_tallies(int64_t* out4, int32_t reset) {
  out4[0] = g_hh_calls; out4[1] = g_hh_far; out4[2] = g_hh_epa; out4[3] = g_hh_gjk_iters;
  if (reset) g_hh_calls = g_hh_far = g_hh_epa = g_hh_gjk_iters = 0;
}
/* test entry (tests/test_oracle_kat.py): two point sets with poses [R 9 row-major | t 3]; out = [pa 3, pb 3, n 3, dist]; stats
 * = [GJK iterations, 1 if the polytope search decided, support points it added]; returns 0 when the hulls are farther apart than max_dist */
int32_t dgo_hull_hull(const real* pts_a, int32_t na, const real* pose_a, const real* pts_b, int32_t nb, const real* pose_b, real max_dist, real* out10, int32_t* stats3) {
  HullPair h; h.pa = pts_a; h.pb = pts_b; h.na = na; h.nb = nb; h.RA = mfrom9(pose_a); h.RB = mfrom9(pose_b);
  v3 ta = V(pose_a[9], pose_a[10], pose_a[11]), tb = V(pose_b[9], pose_b[10], pose_b[11]); h.tBA = vsub(tb, ta);
  v3 ca = V(0, 0, 0), cb = V(0, 0, 0);
  for (int k = 0; k < na; k++) ca = vadd(ca, V(pts_a[3 * k], pts_a[3 * k + 1], pts_a[3 * k + 2]));
  for (int k = 0; k < nb; k++) cb = vadd(cb, V(pts_b[3 * k], pts_b[3 * k + 1], pts_b[3 * k + 2]));
  v3 seed = vsub(mv(&h.RA, vscale(ca, 1.0 / na)), vadd(mv(&h.RB, vscale(cb, 1.0 / nb)), h.tBA));
  v3 pa, pb, n; real dist; int st[3] = {0, 0, 0};
  int hit = hull_hull(&h, seed, max_dist, &pa, &pb, &n, &dist, st);
  if (stats3) { stats3[0] = st[0]; stats3[1] = st[1]; stats3[2] = st[2]; }
  if (!hit) return 0;
  pa = vadd(pa, ta); pb = vadd(pb, ta);
  out10[0] = pa.x; out10[1] = pa.y; out10[2] = pa.z; out10[3] = pb.x; out10[4] = pb.y; out10[5] = pb.z; out10[6] = n.x; out10[7] = n.y; out10[8] = n.z; out10[9] = dist;
  return 1;
}

static int collide(const Scene* s, const BodyWS* wsb, Contact* cs) {
  int nc = 0; real margin = s->F[DG_HF_CONTACT_MARGIN];
  for (int pi = 0; pi < s->npairs; pi++) {
    WShape A, Bs; shape_world(s, wsb, s->PI[pi * DG_PI_STRIDE + DG_PI_A], &A); shape_world(s, wsb, s->PI[pi * DG_PI_STRIDE + DG_PI_B], &Bs);
    const WShape *a = &A, *b = &Bs; real flip = 1.0;
    /* canonical order: lower type id first, except that a box is always `b` */
    if (a->type == DG_SHAPE_BOX || (b->type != DG_SHAPE_BOX && a->type > b->type)) { const WShape* t = a; a = b; b = t; flip = -1.0; }
    v3 pa, pb, n; real dist;
    #define EMIT_F(feature) do { if (flip > 0) add_contact(cs, &nc, s->max_contacts, a, b, pa, pb, n, dist, DG_CONTACT_KEY(pi, feature)); \
                        else add_contact(cs, &nc, s->max_contacts, b, a, pb, pa, vscale(n, -1.0), dist, DG_CONTACT_KEY(pi, feature)); } while (0)
    #define EMIT() EMIT_F(0)
    if (a->type == DG_SHAPE_SPHERE && b->type == DG_SHAPE_SPHERE) {
      if (sphere_sphere(a->p, a->prm[0], b->p, b->prm[0], margin, &pa, &pb, &n, &dist)) EMIT();
    } else if (a->type == DG_SHAPE_SPHERE && b->type == DG_SHAPE_BOX) {
      if (sphere_box(a->p, a->prm[0], b, margin, &pa, &pb, &n, &dist)) EMIT();
    } else if (a->type == DG_SHAPE_SPHERE && (b->type == DG_SHAPE_CAPSULE || b->type == DG_SHAPE_POINTS)) {
      v3 e0, e1; seg_ends(b, &e0, &e1); v3 cb = closest_on_seg(e0, e1, a->p);
      if (sphere_sphere(a->p, a->prm[0], cb, b->prm[0], margin, &pa, &pb, &n, &dist)) EMIT();
    } else if (a->type == DG_SHAPE_POINTS && b->type == DG_SHAPE_POINTS && s->F[DG_HF_HULL_CONTACTS] > 0) {
      /* hull against hull (DG_HF_HULL_CONTACTS): bounding spheres around the fitted capsules' centres first (prm[2]) */
      const real hm = s->F[DG_HF_HULL_MARGIN], reach = margin + 2 * hm;
      if (vnorm(vsub(a->p, b->p)) - a->prm[2] - b->prm[2] < reach) {
        HullPair h; h.pa = s->PF + 3 * a->poff; h.na = a->npts; h.pb = s->PF + 3 * b->poff; h.nb = b->npts; h.RA = a->Rl; h.RB = b->Rl; h.tBA = vsub(b->pl, a->pl);
        if (hull_hull(&h, vsub(a->p, b->p), reach, &pa, &pb, &n, &dist, NULL) && dist - 2 * hm < margin) {
          dist -= 2 * hm; pa = vsub(vadd(pa, a->pl), vscale(n, hm)); pb = vadd(vadd(pb, a->pl), vscale(n, hm)); EMIT();
        }
      }
    } else if ((a->type == DG_SHAPE_CAPSULE || a->type == DG_SHAPE_POINTS) && (b->type == DG_SHAPE_CAPSULE || b->type == DG_SHAPE_POINTS)) {
      /* hull against capsule, and hull against hull with DG_HF_HULL_CONTACTS off: the capsule fitted to each hull (documented approximation) */
      v3 a0, a1, b0, b1, ca, cb; seg_ends(a, &a0, &a1); seg_ends(b, &b0, &b1); seg_seg(a0, a1, b0, b1, &ca, &cb);
      if (sphere_sphere(ca, a->prm[0], cb, b->prm[0], margin, &pa, &pb, &n, &dist)) EMIT();
    } else if (a->type == DG_SHAPE_CAPSULE && b->type == DG_SHAPE_BOX) {
      v3 e0, e1; seg_ends(a, &e0, &e1);
      if (sphere_box(e0, a->prm[0], b, margin, &pa, &pb, &n, &dist)) EMIT();
      if (a->prm[1] > 0 && sphere_box(e1, a->prm[0], b, margin, &pa, &pb, &n, &dist)) EMIT_F(1);
    } else if (a->type == DG_SHAPE_POINTS && b->type == DG_SHAPE_BOX) {
      /* hull vertices against the box: keep the 4 deepest (ties -> lower index) */
      int bi[4] = {-1, -1, -1, -1}; real bd[4] = {HUGE_R, HUGE_R, HUGE_R, HUGE_R};
      m3 Rl = a->Rl; v3 pl = a->pl;
      for (int k = 0; k < a->npts; k++) {
        const real* pp = s->PF + 3 * (a->poff + k);
        v3 pwk = vadd(pl, mv(&Rl, V(pp[0], pp[1], pp[2])));
        v3 qa, qb, qn; real qd;
        if (!sphere_box(pwk, 0.0, b, margin, &qa, &qb, &qn, &qd)) continue;
        for (int j = 0; j < 4; j++) if (qd < bd[j]) { for (int m = 3; m > j; m--) { bd[m] = bd[m - 1]; bi[m] = bi[m - 1]; } bd[j] = qd; bi[j] = k; break; }
      }
      for (int j = 0; j < 4; j++) if (bi[j] >= 0) {
        const real* pp = s->PF + 3 * (a->poff + bi[j]);
        v3 pwk = vadd(pl, mv(&Rl, V(pp[0], pp[1], pp[2])));
        if (sphere_box(pwk, 0.0, b, margin, &pa, &pb, &n, &dist)) EMIT_F(bi[j]);
      }
    }
    #undef EMIT
    #undef EMIT_F
  }
  return nc;
}

/* ------------------------------------------------------------- solver */
typedef struct {
  int body_a, body_b; /* body_b = -1: single-body row */
  real JA[MAXV], RA[MAXV], JB[MAXV], RB[MAXV];
  real b, lo, hi, acc, diag; /* diag = J M^-1 J^T */
  int normal_row; real mu;   /* friction rows: index of the normal row */
  int motor_link;              /* global link index for motor rows else -1 */
  int limit_dof; real limit_sign; /* joint-limit rows: local joint index (else -1) and +1 (lower) / -1 (upper) */
} Row;

static real row_jv(const Row* r, BodyWS* wsb) {
  real s = 0; const BodyWS* a = &wsb[r->body_a];
  for (int k = 0; k < 6 + a->n; k++) s += r->JA[k] * a->dv[k];
  if (r->body_b >= 0) { const BodyWS* b = &wsb[r->body_b]; for (int k = 0; k < 6 + b->n; k++) s += r->JB[k] * b->dv[k]; }
  return s;
}
static real gen_vel_dot(const Scene* s, const real* st, const BodyWS* ws, const real* J) {
  real r = 0;
  if (!ws->fixed) for (int k = 0; k < 6; k++) r += J[k] * ws->v0.v[k];
  for (int i = 0; i < ws->n; i++) r += J[6 + i] * st[link_i(s, ws->first + i)[DG_LI_STATE_OFF] + DG_LS_QD];
  return r;
}
/* spatial unit force at world point p along world direction n, in the coords of local link lk */
static s6 point_force(const BodyWS* ws, int lk, v3 p, v3 n) {
  m3 R; v3 o; link_world(ws, lk, &R, &o);
  v3 rl = mtv(&R, vsub(p, o)), nl = mtv(&R, n);
  return mk6(vcross(rl, nl), nl);
}
static void tangent_basis(v3 n, v3* t1, v3* t2) { /* btPlaneSpace1 [R] */
  if (fabs(n.z) > 0.7071067811865475244) {
    real a = n.y * n.y + n.z * n.z, k = 1.0 / sqrt(a);
    *t1 = V(0, -n.z * k, n.y * k); *t2 = V(a * k, -n.x * t1->z, n.x * t1->y);
  } else {
    real a = n.x * n.x + n.y * n.y, k = 1.0 / sqrt(a);
    *t1 = V(-n.y * k, n.x * k, 0); *t2 = V(-n.z * t1->y, n.z * t1->x, a * k);
  }
}
static int make_contact_row(const Scene* s, const real* st, BodyWS* wsb, const Contact* c, v3 dir, Row* r) {
  memset(r, 0, sizeof *r); r->motor_link = -1; r->normal_row = -1; r->limit_dof = -1;
  BodyWS *A = &wsb[c->body_a], *Bw = &wsb[c->body_b];
  int a_dyn = !(A->fixed && A->n == 0), b_dyn = !(Bw->fixed && Bw->n == 0);
  if (!a_dyn && !b_dyn) return 0;
  real diag = 0, jv = 0;
  if (a_dyn) {
    s6 f = point_force(A, c->link_a, c->p, dir);
    r->body_a = c->body_a; body_response(A, c->link_a, &f, -1, r->JA, r->RA);
    for (int k = 0; k < 6 + A->n; k++) diag += r->JA[k] * r->RA[k];
    jv += gen_vel_dot(s, st, A, r->JA);
    if (b_dyn) {
      s6 fb = point_force(Bw, c->link_b, c->p, vscale(dir, -1.0));
      r->body_b = c->body_b; body_response(Bw, c->link_b, &fb, -1, r->JB, r->RB);
      for (int k = 0; k < 6 + Bw->n; k++) diag += r->JB[k] * r->RB[k];
      jv += gen_vel_dot(s, st, Bw, r->JB);
    } else r->body_b = -1;
  } else {
    s6 fb = point_force(Bw, c->link_b, c->p, vscale(dir, -1.0));
    r->body_a = c->body_b; r->body_b = -1; body_response(Bw, c->link_b, &fb, -1, r->JA, r->RA);
    for (int k = 0; k < 6 + Bw->n; k++) diag += r->JA[k] * r->RA[k];
    jv += gen_vel_dot(s, st, Bw, r->JA);
  }
  r->diag = diag; r->b = -jv;
  return diag > 1e-18;
}

/* One row of a fixed constraint (reference model.py:74-75, createConstraint(JOINT_FIXED); Bullet btMultiBodyFixedConstraint
 * [R]): a unit force along world direction `dir` at world point pa on side A and the opposite one at pb on side B
 * (torque = 0), or a unit torque about `dir` on A and the opposite on B (torque = 1).  Like make_contact_row, the row's
 * first side is the dynamic one of (A, B). */
static int make_constraint_row(const Scene* s, const real* st, BodyWS* wsb, const int32_t* ki, v3 pa, v3 pb, v3 dir, int torque, Row* r) {
  memset(r, 0, sizeof *r); r->motor_link = -1; r->normal_row = -1; r->limit_dof = -1; r->body_b = -1;
  const int ba = ki[DG_KI_BODY_A], bb = ki[DG_KI_BODY_B];
  BodyWS *A = &wsb[ba], *Bw = &wsb[bb];
  const int la = ki[DG_KI_LINK_A] < 0 ? -1 : ki[DG_KI_LINK_A] - A->first, lb = ki[DG_KI_LINK_B] < 0 ? -1 : ki[DG_KI_LINK_B] - Bw->first;
  int a_dyn = !(A->fixed && A->n == 0), b_dyn = !(Bw->fixed && Bw->n == 0);
  if (!a_dyn && !b_dyn) return 0;
  real diag = 0, jv = 0; int side = 0;
  for (int k = 0; k < 2; k++) {
    BodyWS* W = k == 0 ? A : Bw; if (!(k == 0 ? a_dyn : b_dyn)) continue;
    const int lk = k == 0 ? la : lb; const v3 d = k == 0 ? dir : vscale(dir, -1.0);
    s6 f;
    if (torque) { m3 R; v3 o; link_world(W, lk, &R, &o); f = mk6(mtv(&R, d), V(0, 0, 0)); }
    else f = point_force(W, lk, k == 0 ? pa : pb, d);
    real* J = side == 0 ? r->JA : r->JB; real* Rr = side == 0 ? r->RA : r->RB;
    if (side == 0) r->body_a = k == 0 ? ba : bb; else r->body_b = bb;
    body_response(W, lk, &f, -1, J, Rr);
    for (int q = 0; q < 6 + W->n; q++) diag += J[q] * Rr[q];
    jv += gen_vel_dot(s, st, W, J);
    side++;
  }
  r->diag = diag; r->b = -jv;
  return diag > 1e-18;
}

/* one substep of length h for env state st (Bullet btMultiBodyDynamicsWorld::
 * internalSingleStepSimulation order [R]: collide at the current poses, forward
 * dynamics, velocity update, constraint solve, position update) */
static void substep(dgo_world* w, int env, int last) {
  Scene* s = &w->sc; real* st = env_state(w, env); real h = s->h;
  if (last) /* force_torque_sensor: generalised velocities at the start of the step's last substep */
    for (int b = 0; b < s->nb; b++) {
      const int32_t* bi = body_i(s, b); int po = bi[DG_BI_PREV_OFF]; if (po < 0) continue;
      for (int i = 0; i < bi[DG_BI_N_LINKS]; i++) st[po + i] = st[link_i(s, bi[DG_BI_FIRST_LINK] + i)[DG_LI_STATE_OFF] + DG_LS_QD];
      if (!body_fixed(s, b)) for (int k = 0; k < 6; k++) st[po + bi[DG_BI_N_LINKS] + k] = st[bi[DG_BI_STATE_OFF] + DG_BS_LINVEL + k];
    }
  BodyWS* wsb = (BodyWS*)scratch(0, sizeof(BodyWS) * (size_t)s->nb);
  Row* rows = (Row*)scratch(1, sizeof(Row) * MAXROWS); int nr = 0;
  Contact cs[MAXC];
  for (int b = 0; b < s->nb; b++) { body_kinematics(s, st, b, &wsb[b], NULL); body_velocities(s, st, b, &wsb[b]); }
  int nc = collide(s, wsb, cs); w->last_contacts[env] = nc;
  /* forward dynamics + velocity update */
  for (int b = 0; b < s->nb; b++) {
    BodyWS* ws = &wsb[b];
    if (ws->fixed && ws->n == 0) continue;
    body_aba(s, st, b, ws);
    real* bs = st + body_i(s, b)[DG_BI_STATE_OFF];
    if (!ws->fixed) {
      v3 al = lin(&ws->a0), aa = ang(&ws->a0), wb = ang(&ws->v0), vb = lin(&ws->v0);
      v3 acl = vadd(al, vcross(wb, vb)); /* classical acceleration of the base origin */
      v3 dvw = mv(&ws->R0, acl), dww = mv(&ws->R0, aa);
      bs[DG_BS_LINVEL] += h * dvw.x; bs[DG_BS_LINVEL + 1] += h * dvw.y; bs[DG_BS_LINVEL + 2] += h * dvw.z;
      bs[DG_BS_ANGVEL] += h * dww.x; bs[DG_BS_ANGVEL + 1] += h * dww.y; bs[DG_BS_ANGVEL + 2] += h * dww.z;
    }
    for (int i = 0; i < ws->n; i++) st[link_i(s, ws->first + i)[DG_LI_STATE_OFF] + DG_LS_QD] += h * ws->qdd[i];
    body_velocities(s, st, b, ws); /* rows see the updated velocities */
    memset(ws->dv, 0, sizeof ws->dv);
  }
  /* rows: motors, then joint limits (btMultiBodyJointMotor / JointLimitConstraint [R]) */
  real erp = s->F[DG_HF_LIMIT_ERP];
  for (int b = 0; b < s->nb; b++) {
    BodyWS* ws = &wsb[b];
    for (int i = 0; i < ws->n; i++) {
      int gl = ws->first + i; real* ls = st + link_i(s, gl)[DG_LI_STATE_OFF]; const real* mc = w->mcfg + gl * DG_MC_STRIDE;
      ls[DG_LS_APPLIED] = 0.0;
      real maxf = mc[DG_MC_MAX_IMPULSE_SCALE], maximp = maxf < 0 ? -maxf : maxf * h * s->F[DG_HF_MOTOR_IMPULSE_SCALE]; /* time base of the bound: substep or full step [R] */
      if (maximp > 0) {
        Row* r = &rows[nr++]; memset(r, 0, sizeof *r); r->body_a = b; r->body_b = -1; r->normal_row = -1; r->motor_link = gl; r->limit_dof = -1;
        body_response(ws, -1, NULL, i, r->JA, r->RA);
        r->diag = r->RA[6 + i];
        /* rhs = kp*(q*-q)/dt + qd + kd*(qd*-qd) as a velocity target; error = target - qd */
        r->b = mc[DG_MC_KP] * (ls[DG_LS_TARGET_POS] - ls[DG_LS_Q]) / h + mc[DG_MC_KD] * (ls[DG_LS_TARGET_VEL] - ls[DG_LS_QD]);
        r->lo = -maximp; r->hi = maximp;
      }
    }
  }
  for (int b = 0; b < s->nb; b++) {
    BodyWS* ws = &wsb[b];
    for (int i = 0; i < ws->n; i++) {
      int gl = ws->first + i; const real* lf = link_f(s, gl); real* ls = st + link_i(s, gl)[DG_LI_STATE_OFF];
      if (lf[DG_LF_LOWER] > lf[DG_LF_UPPER]) continue;
      for (int side = 0; side < 2; side++) {
        real sg = side == 0 ? 1.0 : -1.0;
        real dist = side == 0 ? ls[DG_LS_Q] - lf[DG_LF_LOWER] : lf[DG_LF_UPPER] - ls[DG_LS_Q];
        if (dist >= 0.25) continue; /* rows that cannot become active within one substep are skipped */
        Row* r = &rows[nr++]; memset(r, 0, sizeof *r); r->body_a = b; r->body_b = -1; r->normal_row = -1; r->motor_link = -1; r->limit_dof = i; r->limit_sign = sg;
        body_response(ws, -1, NULL, i, r->JA, r->RA);
        for (int k = 0; k < 6 + ws->n; k++) { r->JA[k] *= sg; r->RA[k] *= sg; }
        r->diag = r->RA[6 + i] * sg;
        real relv = sg * ls[DG_LS_QD];
        r->b = -relv + (dist > 0 ? -dist / h : -dist * erp / h);
        r->lo = 0; r->hi = HUGE_R;
      }
    }
  }
  /* Starting impulses of the motor rows (DG_HF_MOTOR_GUESS): without the clamps a body's motor rows are the linear system
   * (M^-1 restricted to the motorised joints) lambda = b, solved directly here and clamped to the rows' bounds -- the
   * sweeps then start next to their fixed point instead of at zero (Bullet starts at zero [R]; the fixed point is the
   * same, the residual early-out fires after ~6 sweeps instead of ~35 for a position-controlled arm).
   * DG_HF_LIMIT_GUESS: a joint whose motor target lies beyond an ACTIVE limit row of the same joint (the two rows share their
   * Jacobian) is "pinned": it enters the system as one unknown, the joint's total impulse t, with the limit row's velocity as
   * right-hand side; it starts with the motor saturated into the limit (dir x bound) and the limit row holding the balance,
   * bound - dir x t >= 0 -- unless the motor alone cannot reach the limit velocity (dir x t > bound: held at its bound like any
   * other row that leaves its bounds, limit row at zero).  Left to the sweeps, such a pair of rows ramps up against each other
   * by (b_motor - b_limit) / diag per sweep until the motor saturates. */
  if (s->F[DG_HF_MOTOR_GUESS] > 0) {
    const int pinning = s->F[DG_HF_LIMIT_GUESS] > 0;
    /* (only a target that lies beyond the limit by more than the sweeps' own early-out: closer than that the two rows are
     * converged as they stand -- a joint RESTING on its limit under a zero-velocity motor sits exactly there, +-rounding) */
    const real ptol = sqrt(s->F[DG_HF_RESIDUAL_THRESHOLD]);
    for (int b = 0; b < s->nb; b++) {
      BodyWS* ws = &wsb[b]; int idx[MAXL], dof[MAXL], k = 0;
      for (int r = 0; r < nr; r++) if (rows[r].body_a == b && rows[r].motor_link >= 0) { idx[k] = r; dof[k] = rows[r].motor_link - ws->first; k++; }
      if (k == 0 || ws->n > DG_MOTOR_GUESS_MAX) continue;
      /* (symmetrically scaled to a unit diagonal first: finger joints and shoulder joints differ by 1e5 in M^-1, and the
       * device solves this in fp32) */
      real A[MAXL * MAXL], bb[MAXL], x[MAXL], sc_[MAXL]; int pin[MAXL], lrow[MAXL];
      for (int j = 0; j < k; j++) sc_[j] = 1.0 / sqrt(rows[idx[j]].RA[6 + dof[j]]);
      for (int j = 0; j < k; j++) {
        real target = rows[idx[j]].b; pin[j] = 0; lrow[j] = -1;
        if (pinning && ws->n <= DG_MOTOR_GUESS_REFINE)
          for (int r = 0; r < nr; r++) { /* the joint's active limit rows: JA = +-e_dof, b = the velocity the row demands along JA */
            const Row* q = &rows[r]; if (q->body_a != b || q->motor_link >= 0 || q->limit_dof != dof[j]) continue;
            const real sg = q->limit_sign, vlim = sg * q->b; /* lower (sg = +1): dv >= vlim; upper (sg = -1): dv <= vlim */
            if (sg > 0 ? rows[idx[j]].b < vlim - ptol : rows[idx[j]].b > vlim + ptol) { pin[j] = sg > 0 ? -1 : 1; lrow[j] = r; target = vlim; }
          }
        bb[j] = target * sc_[j]; for (int l = 0; l < k; l++) A[l * k + j] = rows[idx[j]].RA[6 + dof[l]] * sc_[j] * sc_[l];
      }
      if (!spd_solve(k, A, bb, x)) continue;
      /* Bounds of the unknowns in the scaled system (unit diagonal).  A pinned joint's unknown is its TOTAL impulse, bounded on
       * one side only: dir x t <= the motor's bound (beyond it the motor is too weak to reach the limit velocity). */
      real blo[MAXL], bhi[MAXL]; int held[MAXL], up[MAXL], dn[MAXL], any = 0;
      for (int j = 0; j < k; j++) {
        const Row* r = &rows[idx[j]];
        blo[j] = pin[j] > 0 ? -HUGE_R : r->lo / sc_[j]; bhi[j] = pin[j] < 0 ? HUGE_R : r->hi / sc_[j];
        up[j] = x[j] > bhi[j]; dn[j] = x[j] < blo[j]; held[j] = up[j] || dn[j]; any |= held[j];
      }
      if (any && ws->n > DG_MOTOR_GUESS_REFINE) continue; /* a bigger body whose solution does not fit its bounds: zero start */
      /* Primal-dual active set (Hintermueller, Ito, Kunisch 2002 for box-constrained problems with an M-matrix-like operator):
       * rows beyond their bounds are held there and the others solved again; then the sets are re-read from x + residual --
       * a held row whose residual pulls it back inside is released, a free row that left its bounds is held -- until the sets
       * repeat, at most DG_MOTOR_GUESS_ROUNDS times.  The fixed sets ARE the solution of the clamped system (2 000 systems of
       * an ur_high_5 rollout: 1.25 rounds on average, 4 at most, and the sweeps confirm it in ONE iteration; a single round
       * left residuals of 1e-2 .. 1e-1 rad/s behind whenever a row saturated: 8 sweeps at the 90th percentile, 26 at the 99th). */
      for (int round = 0; any && round < DG_MOTOR_GUESS_ROUNDS; round++) {
        real A2[MAXL * MAXL], b2[MAXL], val[MAXL];
        for (int j = 0; j < k; j++) val[j] = up[j] ? bhi[j] : blo[j];
        for (int j = 0; j < k; j++) { b2[j] = bb[j]; for (int l = 0; l < k; l++) A2[l * k + j] = A[l * k + j]; }
        for (int j = 0; j < k; j++) if (held[j]) for (int l = 0; l < k; l++) if (!held[l]) b2[l] -= A[l * k + j] * val[j];
        for (int j = 0; j < k; j++) if (held[j]) { for (int l = 0; l < k; l++) { A2[l * k + j] = 0; A2[j * k + l] = 0; } A2[j * k + j] = 1; b2[j] = val[j]; }
        if (!spd_solve(k, A2, b2, x)) break;
        int changed = 0;
        for (int j = 0; j < k; j++) { /* x + residual (the diagonal is 1): a free row's residual is zero, a held row's says which way it wants to go */
          real y = x[j] + bb[j]; for (int l = 0; l < k; l++) y -= A[j * k + l] * x[l];
          const int nu = y > bhi[j], nd = y < blo[j]; changed |= nu != up[j] || nd != dn[j]; up[j] = nu; dn[j] = nd; held[j] = nu || nd;
        }
        if (!changed) break;
      }
      for (int j = 0; j < k; j++) {
        Row* r = &rows[idx[j]]; real t = x[j] * sc_[j], imp, lim = 0.0;
        if (pin[j] && !(pin[j] * t > r->hi)) { /* motor saturated into the limit, the limit row holds the balance (never negative) */
          imp = pin[j] * r->hi; lim = r->hi - pin[j] * t; if (lim < 0) lim = 0;
          Row* q = &rows[lrow[j]]; q->acc = lim; t = imp - pin[j] * lim; /* (the limit row pushes along -dir) */
        } else { imp = t < r->lo ? r->lo : (t > r->hi ? r->hi : t); t = imp; }
        r->acc = imp; for (int q = 0; q < 6 + ws->n; q++) ws->dv[q] += r->RA[q] * t;
      }
    }
  }
  /* fixed constraints: behind the motor / limit rows (the non-contact rows of btMultiBodyConstraintSolver [R]), before the
   * contacts.  Errors: the pivot of side B minus the pivot of side A, and the rotation vector that takes A's pivot frame
   * onto B's (2 x the vector part of qB qA^-1, shorter arc); a fraction DG_HF_CONTACT_ERP of each is corrected per
   * substep.  Impulses bounded by max_force x h.  Started from zero every substep. */
  for (int q = 0; q < s->ncons; q++) {
    const int32_t* ki = s->KI + q * DG_KI_STRIDE; const real* kf = s->KF + q * DG_KF_STRIDE;
    const real kerp = s->F[DG_HF_CONTACT_ERP], maximp = kf[DG_KF_MAX_FORCE] * h;
    v3 P[2]; qt Q[2];
    for (int k = 0; k < 2; k++) {
      const BodyWS* W = &wsb[ki[k == 0 ? DG_KI_BODY_A : DG_KI_BODY_B]]; const int gl = ki[k == 0 ? DG_KI_LINK_A : DG_KI_LINK_B];
      m3 R; v3 o; link_world(W, gl < 0 ? -1 : gl - W->first, &R, &o);
      const real* pp = kf + (k == 0 ? DG_KF_POS_A : DG_KF_POS_B); const real* qq = kf + (k == 0 ? DG_KF_QUAT_A : DG_KF_QUAT_B);
      P[k] = vadd(o, mv(&R, V(pp[0], pp[1], pp[2])));
      qt ql = gl < 0 ? W->q0 : qfrom_mat(&R); qt qo = {qq[0], qq[1], qq[2], qq[3]}; Q[k] = qnormalize(qmul(ql, qo));
    }
    const v3 perr = vsub(P[1], P[0]);
    qt qe = qmul(Q[1], qconj(Q[0])); const real sg = qe.w < 0 ? -2.0 : 2.0; const v3 aerr = V(sg * qe.x, sg * qe.y, sg * qe.z);
    for (int d = 0; d < 6; d++) {
      const v3 dir = V(d % 3 == 0, d % 3 == 1, d % 3 == 2); Row* r = &rows[nr];
      if (!make_constraint_row(s, st, wsb, ki, P[0], P[1], dir, d >= 3, r)) continue;
      r->b += kerp * vdot(d < 3 ? perr : aerr, dir) / h; r->lo = -maximp; r->hi = maximp; nr++;
    }
  }
  /* contacts: all normal rows first, then the friction rows (btMultiBodyConstraintSolver order [R]) */
  int first_normal = nr; int crow[MAXC];
  real cerp = s->F[DG_HF_CONTACT_ERP], slop = s->F[DG_HF_LINEAR_SLOP];
  for (int k = 0; k < nc; k++) {
    Row* r = &rows[nr]; crow[k] = -1;
    if (!make_contact_row(s, st, wsb, &cs[k], cs[k].n, r)) continue;
    real pen = cs[k].dist + slop;
    r->b += pen > 0 ? -pen / h : -pen * cerp / h;
    r->lo = 0; r->hi = HUGE_R; crow[k] = nr++;
  }
  (void)first_normal;
  int frow[MAXC][2];
  for (int k = 0; k < nc; k++) {
    frow[k][0] = frow[k][1] = -1; tangent_basis(cs[k].n, &cs[k].t1, &cs[k].t2);
    if (crow[k] < 0 || cs[k].mu <= 0) continue;
    for (int d = 0; d < 2; d++) {
      Row* r = &rows[nr];
      if (!make_contact_row(s, st, wsb, &cs[k], d == 0 ? cs[k].t1 : cs[k].t2, r)) continue;
      r->normal_row = crow[k]; r->mu = cs[k].mu; frow[k][d] = nr++;
    }
  }
  /* warm starting (btSequentialImpulseConstraintSolver: m_appliedImpulse = cp.m_appliedImpulse * m_warmstartingFactor [R]):
   * a contact that was there in the previous substep -- same candidate pair, same feature -- starts from a fraction of the
   * impulses its rows ended with, applied to the velocity change before the first sweep */
  real* warm = s->warm_off >= 0 ? st + s->warm_off : NULL;
  if (warm) {
    const real wfac[3] = {s->F[DG_HF_WARMSTART], s->F[DG_HF_WARMSTART_FRICTION], s->F[DG_HF_WARMSTART_FRICTION]};
    const int np = (int)warm[0];
    for (int k = 0; k < nc; k++) {
      const real* e = NULL;
      for (int j = 0; j < np && !e; j++) if ((int)warm[1 + j * DG_WS_STRIDE + DG_WS_KEY] == cs[k].key) e = warm + 1 + j * DG_WS_STRIDE;
      if (!e) continue;
      const int ri[3] = {crow[k], frow[k][0], frow[k][1]};
      for (int d = 0; d < 3; d++) {
        if (ri[d] < 0 || wfac[d] <= 0) continue;
        Row* r = &rows[ri[d]]; const real imp = wfac[d] * e[DG_WS_NORMAL + d];
        r->acc = imp;
        BodyWS* A = &wsb[r->body_a]; for (int j = 0; j < 6 + A->n; j++) A->dv[j] += r->RA[j] * imp;
        if (r->body_b >= 0) { BodyWS* Bw = &wsb[r->body_b]; for (int j = 0; j < 6 + Bw->n; j++) Bw->dv[j] += r->RB[j] * imp; }
      }
    }
  }
  /* projected Gauss-Seidel with the residual early-out (pybullet solverResidualThreshold [R]) */
  real thr = s->F[DG_HF_RESIDUAL_THRESHOLD]; int it;
  for (it = 0; it < s->iters; it++) {
    real maxres = 0;
    for (int k = 0; k < nr; k++) {
      Row* r = &rows[k];
      real lo = r->lo, hi = r->hi;
      if (r->normal_row >= 0) { hi = r->mu * rows[r->normal_row].acc; lo = -hi; }
      real delta = (r->b - row_jv(r, wsb)) / r->diag;
      real nacc = r->acc + delta; nacc = nacc < lo ? lo : (nacc > hi ? hi : nacc);
      delta = nacc - r->acc; r->acc = nacc;
      BodyWS* A = &wsb[r->body_a]; for (int j = 0; j < 6 + A->n; j++) A->dv[j] += r->RA[j] * delta;
      if (r->body_b >= 0) { BodyWS* Bw = &wsb[r->body_b]; for (int j = 0; j < 6 + Bw->n; j++) Bw->dv[j] += r->RB[j] * delta; }
      real res = delta * r->diag; if (res * res > maxres) maxres = res * res;
    }
    if (maxres <= thr) { it++; break; }
  }
  w->last_iters[env] = it;
  for (int k = 0; k < nc; k++) { /* solved impulses of this substep's contacts, for the force/torque sensor */
    cs[k].imp[0] = crow[k] >= 0 ? rows[crow[k]].acc : 0.0;
    cs[k].imp[1] = frow[k][0] >= 0 ? rows[frow[k][0]].acc : 0.0; cs[k].imp[2] = frow[k][1] >= 0 ? rows[frow[k][1]].acc : 0.0;
    w->last_cs[(size_t)env * MAXC + k] = cs[k];
  }
  if (warm) { /* this substep's contacts and impulses, for the next one */
    warm[0] = (real)nc;
    for (int k = 0; k < nc; k++) {
      real* e = warm + 1 + k * DG_WS_STRIDE;
      e[DG_WS_KEY] = (real)cs[k].key; e[DG_WS_NORMAL] = cs[k].imp[0]; e[DG_WS_T1] = cs[k].imp[1]; e[DG_WS_T2] = cs[k].imp[2];
    }
  }
  for (int k = 0; k < nr; k++) if (rows[k].motor_link >= 0) st[link_i(s, rows[k].motor_link)[DG_LI_STATE_OFF] + DG_LS_APPLIED] = rows[k].acc / h;
  /* apply velocity changes, integrate positions (btMultiBody::stepPositionsMultiDof [R]) */
  real vmax = s->F[DG_HF_MAX_COORD_VEL];
  for (int b = 0; b < s->nb; b++) {
    BodyWS* ws = &wsb[b];
    if (ws->fixed && ws->n == 0) continue;
    real* bs = st + body_i(s, b)[DG_BI_STATE_OFF];
    if (!ws->fixed) {
      v3 dw = mv(&ws->R0, V(ws->dv[0], ws->dv[1], ws->dv[2])), dl = mv(&ws->R0, V(ws->dv[3], ws->dv[4], ws->dv[5]));
      bs[DG_BS_ANGVEL] += dw.x; bs[DG_BS_ANGVEL + 1] += dw.y; bs[DG_BS_ANGVEL + 2] += dw.z;
      bs[DG_BS_LINVEL] += dl.x; bs[DG_BS_LINVEL + 1] += dl.y; bs[DG_BS_LINVEL + 2] += dl.z;
      for (int k = 0; k < 3; k++) bs[DG_BS_POS + k] += h * bs[DG_BS_LINVEL + k];
      v3 wv = V(bs[DG_BS_ANGVEL], bs[DG_BS_ANGVEL + 1], bs[DG_BS_ANGVEL + 2]);
      real wn = vnorm(wv), th = wn * h; qt dq;
      if (th > 1e-12) { real sn = sin(0.5 * th) / wn; dq.x = wv.x * sn; dq.y = wv.y * sn; dq.z = wv.z * sn; dq.w = cos(0.5 * th); }
      else { dq.x = 0.5 * h * wv.x; dq.y = 0.5 * h * wv.y; dq.z = 0.5 * h * wv.z; dq.w = 1.0; }
      qt q0 = {bs[3], bs[4], bs[5], bs[6]}; qt qn = qnormalize(qmul(dq, q0));
      bs[3] = qn.x; bs[4] = qn.y; bs[5] = qn.z; bs[6] = qn.w;
    }
    for (int i = 0; i < ws->n; i++) {
      real* ls = st + link_i(s, ws->first + i)[DG_LI_STATE_OFF];
      real qd = ls[DG_LS_QD] + ws->dv[6 + i]; qd = qd > vmax ? vmax : (qd < -vmax ? -vmax : qd);
      ls[DG_LS_QD] = qd; ls[DG_LS_Q] += h * qd;
    }
  }
}

/* -------------------------------------------------- frames and queries */
typedef struct { v3 p; qt q; v3 v; v3 w; } FrameState;
/* world pose/velocity of frame fr (-1 = base) of body b.  com selects the inertial frame. */
static void frame_state(const Scene* s, const real* st, int b, int fr, int com, const real* q_override, FrameState* o) {
  BodyWS* ws = (BodyWS*)malloc(sizeof(BodyWS));
  body_kinematics(s, st, b, ws, q_override); body_velocities(s, st, b, ws);
  int lk; v3 off; qt qoff;
  if (fr < 0) {
    const real* bf = body_f(s, b); lk = -1;
    if (com) { off = V(bf[DG_BF_REPORT_POS], bf[DG_BF_REPORT_POS + 1], bf[DG_BF_REPORT_POS + 2]); qt t = {bf[DG_BF_REPORT_QUAT], bf[DG_BF_REPORT_QUAT + 1], bf[DG_BF_REPORT_QUAT + 2], bf[DG_BF_REPORT_QUAT + 3]}; qoff = t; }
    else { off = V(0, 0, 0); qt t = {0, 0, 0, 1}; qoff = t; }
  } else {
    const int32_t* fi = s->FI + fr * DG_FI_STRIDE; const real* ff = s->FF + fr * DG_FF_STRIDE;
    lk = fi[DG_FI_LINK] < 0 ? -1 : fi[DG_FI_LINK] - ws->first;
    const real* pp = ff + (com ? DG_FF_COM_POS : DG_FF_POS); const real* qq = ff + (com ? DG_FF_COM_QUAT : DG_FF_QUAT);
    off = V(pp[0], pp[1], pp[2]); qt t = {qq[0], qq[1], qq[2], qq[3]}; qoff = t;
  }
  m3 R; v3 p; link_world(ws, lk, &R, &p);
  qt ql = qfrom_mat(&R);
  if (lk < 0) ql = ws->q0;
  o->p = vadd(p, mv(&R, off)); o->q = qnormalize(qmul(ql, qoff));
  const s6* v = lk < 0 ? &ws->v0 : &ws->v[lk];
  v3 wl = ang(v), vl = lin(v);
  o->w = mv(&R, wl); o->v = mv(&R, vadd(vl, vcross(wl, off)));
  free(ws);
}
int dgo_frame_state(dgo_world* w, int32_t env, int32_t body, int32_t frame, int32_t com, real* out) {
  FrameState f; frame_state(&w->sc, env_state(w, env), body, frame, com, NULL, &f);
  out[0] = f.p.x; out[1] = f.p.y; out[2] = f.p.z; out[3] = f.q.x; out[4] = f.q.y; out[5] = f.q.z; out[6] = f.q.w;
  out[7] = f.v.x; out[8] = f.v.y; out[9] = f.v.z; out[10] = f.w.x; out[11] = f.w.y; out[12] = f.w.z;
  return 0;
}

/* ------------------------------------------------ inverse kinematics */
/* Restates what p.calculateInverseKinematics does for the reference's call
 * (ik_controller.py:61-69) as recollected [R]: up to IK_ITERS damped
 * least-squares steps from the current joint angles, each with a fresh
 * Jacobian; with the four null-space lists of DoF length (UR5) the task-space
 * DLS + null-space projection variant, otherwise (Jaco) the joint-space DLS
 * variant; steps scaled so no joint moves more than IK_MAX_ANGLE; stop when
 * the position error is below IK_RESIDUAL. */
static void ik_jacobian(const BodyWS* ws, int lk, v3 pe, real* Jm /* [6][n] */) {
  int n = ws->n; memset(Jm, 0, sizeof(real) * 6 * (size_t)n);
  for (int i = lk; i >= 0; i = ws->parent[i]) {
    v3 sw = ang(&ws->S[i]), sl = lin(&ws->S[i]);
    v3 aw = mv(&ws->Rw[i], sw), al = mv(&ws->Rw[i], sl);
    v3 jl = vadd(vcross(aw, vsub(pe, ws->pw[i])), al);
    Jm[0 * n + i] = jl.x; Jm[1 * n + i] = jl.y; Jm[2 * n + i] = jl.z; Jm[3 * n + i] = aw.x; Jm[4 * n + i] = aw.y; Jm[5 * n + i] = aw.z;
  }
}
static void run_ik(dgo_world* w, const real* st, int op, const real* act, real* q /* [n] out */) {
  const Scene* s = &w->sc; const int32_t* oi = s->OI + op * DG_OI_STRIDE; const real* of = s->OF + op * DG_OF_STRIDE;
  int b = oi[DG_OI_BODY], fr = oi[DG_OI_FRAME], flags = oi[DG_OI_FLAGS];
  int use_orn = flags & DG_IK_USE_ORIENTATION, nullsp = flags & DG_IK_NULLSPACE, m = use_orn ? 6 : 3;
  BodyWS* ws = (BodyWS*)malloc(sizeof(BodyWS));
  body_kinematics(s, st, b, ws, NULL);
  int n = ws->n;
  for (int i = 0; i < n; i++) q[i] = st[link_i(s, ws->first + i)[DG_LI_STATE_OFF] + DG_LS_Q];
  FrameState cur; frame_state(s, st, b, fr, 1, NULL, &cur);
  v3 tp = vadd(cur.p, V(act[0], act[1], act[2]));
  qt tq = cur.q;
  if (use_orn) tq = qmul(cur.q, qfrom_euler(act[3], act[4], act[5])); /* ik_controller.py:56-59 */
  const int32_t* fi = s->FI + fr * DG_FI_STRIDE; int lk = fi[DG_FI_LINK] < 0 ? -1 : fi[DG_FI_LINK] - ws->first;
  const real* rest = s->FL + oi[DG_OI_FLIST]; /* rest[n], lower[n], upper[n], range[n] */
  real lam2 = s->F[DG_HF_IK_LAMBDA_SQ], jd = s->F[DG_HF_IK_JOINT_DAMPING], maxang = s->F[DG_HF_IK_MAX_ANGLE];
  real g0 = s->F[DG_HF_IK_NULL_REST_GAIN], g1 = s->F[DG_HF_IK_NULL_LIMIT_GAIN];
  (void)of;
  for (int it = 0; it < s->ik_iters; it++) {
    FrameState f; frame_state(s, st, b, fr, 1, q, &f);
    v3 ep = vsub(tp, f.p);
    if (vnorm(ep) < s->F[DG_HF_IK_RESIDUAL] && it > 0) break;
    real dS[6] = {ep.x, ep.y, ep.z, 0, 0, 0};
    if (use_orn) {
      qt dq = qmul(tq, qconj(f.q));
      if (dq.w < 0) { dq.x = -dq.x; dq.y = -dq.y; dq.z = -dq.z; dq.w = -dq.w; }
      real sn = sqrt(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z), an = 2.0 * atan2(sn, dq.w);
      real k = sn > 1e-12 ? an / sn : 2.0;
      dS[3] = dq.x * k; dS[4] = dq.y * k; dS[5] = dq.z * k;
    }
    body_kinematics(s, st, b, ws, q);
    real J6[6 * MAXL], J[6 * MAXL]; ik_jacobian(ws, lk, f.p, J6);
    for (int r = 0; r < m; r++) for (int c = 0; c < n; c++) J[r * n + c] = J6[r * n + c];
    real dth[MAXL];
    if (nullsp) {
      real U[36], y[6];
      for (int r = 0; r < m; r++) for (int c = 0; c < m; c++) { real t = 0; for (int k = 0; k < n; k++) t += J[r * n + k] * J[c * n + k]; U[r * m + c] = t + (r == c ? lam2 : 0.0); }
      spd_solve(m, U, dS, y);
      for (int k = 0; k < n; k++) { real t = 0; for (int r = 0; r < m; r++) t += J[r * n + k] * y[r]; dth[k] = t; }
      /* null-space velocity: towards the rest pose, away from violated limits */
      real v0[MAXL], Jv[6], z[6];
      for (int k = 0; k < n; k++) {
        v0[k] = g0 * (rest[k] - q[k]);
        real lo = rest[n + k], hi = rest[2 * n + k], rg = rest[3 * n + k];
        if (q[k] > hi) v0[k] += g1 * (hi - q[k]) / rg;
        if (q[k] < lo) v0[k] += g1 * (lo - q[k]) / rg;
      }
      for (int r = 0; r < m; r++) { real t = 0; for (int k = 0; k < n; k++) t += J[r * n + k] * v0[k]; Jv[r] = t; }
      spd_solve(m, U, Jv, z);
      for (int k = 0; k < n; k++) { real t = 0; for (int r = 0; r < m; r++) t += J[r * n + k] * z[r]; dth[k] += v0[k] - t; }
    } else {
      real A[12 * 12], rhs[12];
      if (n > 12) n = 12;
      for (int r = 0; r < n; r++) { for (int c = 0; c < n; c++) { real t = 0; for (int k = 0; k < m; k++) t += J[k * ws->n + r] * J[k * ws->n + c]; A[r * n + c] = t + (r == c ? jd : 0.0); }
        real t = 0; for (int k = 0; k < m; k++) t += J[k * ws->n + r] * dS[k]; rhs[r] = t; }
      spd_solve(n, A, rhs, dth); n = ws->n;
    }
    real mx = 0; for (int k = 0; k < n; k++) if (fabs(dth[k]) > mx) mx = fabs(dth[k]);
    real sc = mx > maxang ? maxang / mx : 1.0;
    for (int k = 0; k < n; k++) q[k] += sc * dth[k];
  }
  free(ws);
}
int dgo_ik(dgo_world* w, int32_t env, int32_t op, const real* action, real* q_out) {
  run_ik(w, env_state(w, env), op, action, q_out); return 0;
}

/* ------------------------------------------------------ addon program */
static void set_motor(dgo_world* w, int gl, real kp, real kd, real maxforce) {
  real* mc = w->mcfg + gl * DG_MC_STRIDE; mc[DG_MC_KP] = kp; mc[DG_MC_KD] = kd; mc[DG_MC_MAX_IMPULSE_SCALE] = maxforce;
}
static void run_update_ops(dgo_world* w, int env, const real* act, uint64_t mask) {
  Scene* s = &w->sc; real* st = env_state(w, env);
  for (int op = 0; op < s->nops; op++) {
    const int32_t* oi = s->OI + op * DG_OI_STRIDE; const real* of = s->OF + op * DG_OF_STRIDE;
    int code = oi[DG_OI_CODE];
    if (code < DG_OP_JOINT_CONTROL || code > DG_OP_ADMITTANCE) continue;
    if (!((mask >> oi[DG_OI_SLOT]) & 1ULL)) continue;
    const real* a = act + oi[DG_OI_IO_OFF]; const int32_t* il = s->IL + oi[DG_OI_ILIST]; int n = oi[DG_OI_N];
    if (code == DG_OP_JOINT_CONTROL) { /* joint_controller.py:40-58 */
      int mode = oi[DG_OI_FLAGS];
      for (int k = 0; k < n; k++) {
        int gl = il[k]; real* ls = st + link_i(s, gl)[DG_LI_STATE_OFF]; real maxf = link_f(s, gl)[DG_LF_MAX_FORCE];
        if (mode == DG_JC_POSITION) { ls[DG_LS_TARGET_POS] = a[k]; ls[DG_LS_TARGET_VEL] = 0.0; set_motor(w, gl, of[0], of[1], maxf); }
        else if (mode == DG_JC_VELOCITY) { ls[DG_LS_TARGET_VEL] = a[k]; ls[DG_LS_TARGET_POS] = 0.0; set_motor(w, gl, 0.0, of[1], maxf); }
        else ls[DG_LS_TORQUE] = a[k]; /* TORQUE_CONTROL: the velocity motor is left as it was */
      }
    } else if (code == DG_OP_IK_CONTROL) { /* ik_controller.py:51-80 */
      real q[MAXL]; run_ik(w, st, op, a, q);
      int first = body_i(s, oi[DG_OI_BODY])[DG_BI_FIRST_LINK];
      for (int k = 0; k < n; k++) {
        int gl = il[k]; real* ls = st + link_i(s, gl)[DG_LI_STATE_OFF];
        (void)first;
        ls[DG_LS_TARGET_POS] = q[k]; /* joint_cmds[k] pairs with joint_ids[k] (ik_controller.py:69-74) */
        ls[DG_LS_TARGET_VEL] = 0.0; set_motor(w, gl, of[0], of[1], link_f(s, gl)[DG_LF_MAX_FORCE]);
      }
    } else if (code == DG_OP_ADMITTANCE) { /* admittance_controller.py:36-55 */
      /* torque_j = force . J_lin[:,j] + torque . J_ang[:,j]   (p.calculateJacobian at the end-effector link's
       *            inertial frame + offset, :39-46, :50)
       *          + gravity torque needed to hold the pose (p.calculateInverseDynamics with zero velocity and
       *            acceleration, :49)  + kp (target - q) - kd qd  (:52-53), applied in TORQUE_CONTROL (:55) */
      int b = oi[DG_OI_BODY]; BodyWS* ws = (BodyWS*)malloc(sizeof(BodyWS)); body_kinematics(s, st, b, ws, NULL);
      FrameState f; frame_state(s, st, b, oi[DG_OI_FRAME], 1, NULL, &f);
      m3 Rf = qmat(f.q); v3 pw = vadd(f.p, mv(&Rf, V(of[0], of[1], of[2])));
      const int32_t* fi = s->FI + oi[DG_OI_FRAME] * DG_FI_STRIDE; int lk = fi[DG_FI_LINK] < 0 ? -1 : fi[DG_FI_LINK] - ws->first;
      v3 F = V(a[0], a[1], a[2]), T = V(a[3], a[4], a[5]); const real* tgt = s->FL + oi[DG_OI_FLIST];
      for (int k = 0; k < n; k++) {
        int gl = il[k], j = gl - ws->first; real* ls = st + link_i(s, gl)[DG_LI_STATE_OFF];
        v3 aw = mv(&ws->Rw[j], ang(&ws->S[j])), lw = mv(&ws->Rw[j], lin(&ws->S[j])); int rev = link_i(s, gl)[DG_LI_TYPE] == 0;
        /* is joint j an ancestor of (or equal to) the end-effector link? */
        int anc = 0; for (int i = lk; i >= 0; i = ws->parent[i]) if (i == j) anc = 1;
        real tau = 0;
        if (anc) tau += rev ? vdot(F, vcross(aw, vsub(pw, ws->pw[j]))) + vdot(T, aw) : vdot(F, lw);
        /* gravity: every link in the subtree of j */
        for (int i = 0; i < ws->n; i++) {
          int sub = 0; for (int q = i; q >= 0; q = ws->parent[q]) if (q == j) sub = 1;
          if (!sub) continue;
          const real* lf = link_f(s, ws->first + i);
          v3 cw = vadd(ws->pw[i], mv(&ws->Rw[i], V(lf[DG_LF_COM], lf[DG_LF_COM + 1], lf[DG_LF_COM + 2])));
          v3 w8 = vscale(s->g, lf[DG_LF_MASS] * link_mass_scale(s, st, ws->first + i));
          tau -= rev ? vdot(w8, vcross(aw, vsub(cw, ws->pw[j]))) : vdot(w8, lw);
        }
        tau += of[3] * (tgt[k] - ls[DG_LS_Q]) - of[4] * ls[DG_LS_QD];
        ls[DG_LS_TORQUE] += tau;
      }
      free(ws);
    } else if (code == DG_OP_EXTERNAL_FORCE) { /* external_force.py:21-24: WORLD_FRAME force at a world position */
      int b = oi[DG_OI_BODY]; if (body_fixed(s, b)) continue;
      real* bs = st + body_i(s, b)[DG_BI_STATE_OFF]; real* ex = body_ext(s, st, b);
      v3 f = V(a[0], a[1], a[2]); v3 rel = vsub(V(of[0], of[1], of[2]), V(bs[0], bs[1], bs[2])); v3 t = vcross(rel, f);
      ex[0] += f.x; ex[1] += f.y; ex[2] += f.z; ex[3] += t.x; ex[4] += t.y; ex[5] += t.z;
    } else if (code == DG_OP_PROPELLOR) { /* drone_pilot.py:31-37 */
      int b = oi[DG_OI_BODY]; real* as = st + s->addon_off + oi[DG_OI_STATE_OFF];
      as[0] = as[0] + (a[0] - as[0]) * of[2];
      if (body_fixed(s, b)) continue;
      FrameState f; frame_state(s, st, b, oi[DG_OI_FRAME], 1, NULL, &f); /* LINK_FRAME = the link's INERTIAL frame (btMultiBody's m_cachedWorldTransform [R]) */
      m3 R = qmat(f.q); real* bs = st + body_i(s, b)[DG_BI_STATE_OFF]; real* ex = body_ext(s, st, b);
      v3 fw = mv(&R, V(0, 0, of[0] * as[0])), tw = mv(&R, V(0, 0, of[1] * as[0]));
      v3 t = vadd(vcross(vsub(f.p, V(bs[0], bs[1], bs[2])), fw), tw);
      ex[0] += fw.x; ex[1] += fw.y; ex[2] += fw.z; ex[3] += t.x; ex[4] += t.y; ex[5] += t.z;
    }
  }
}
/* p.applyExternalForce / p.applyExternalTorque on frame `fr` (global frame index, -1 base) of body b for one env, as a user
 * addon written in Python issues them (drone_pilot.py:34-37): base wrench + J^T on the joints between the link and the base.
 * LINK_FRAME: force / torque along the axes of the link's INERTIAL frame, pos relative to its origin, the centre of mass
 * (PhysicsServerCommandProcessor: mb->getLink(i).m_cachedWorldTransform / getBaseWorldTransform() [R]) */
int dgo_apply_wrench(dgo_world* w, int32_t body, int32_t frame, int32_t link_frame, const real* force, const real* pos, const real* torque) {
  Scene* s = &w->sc;
  if (body < 0 || body >= s->nb || (body_i(s, body)[DG_BI_FLAGS] & DG_BODY_FROZEN)) { set_err("bad body %d", body); return 1; }
  for (int e = 0; e < w->B; e++) {
    real* st = env_state(w, e);
    v3 F = force ? V(force[3 * e], force[3 * e + 1], force[3 * e + 2]) : V(0, 0, 0), T = torque ? V(torque[3 * e], torque[3 * e + 1], torque[3 * e + 2]) : V(0, 0, 0);
    v3 P = pos ? V(pos[3 * e], pos[3 * e + 1], pos[3 * e + 2]) : V(0, 0, 0);
    if (link_frame) { FrameState f; frame_state(s, st, body, frame, 1, NULL, &f); /* the link's inertial frame, as pybullet's LINK_FRAME [R] */ m3 R = qmat(f.q); F = mv(&R, F); T = mv(&R, T); P = vadd(f.p, mv(&R, P)); }
    if (!body_fixed(s, body)) {
      real* bs = st + body_i(s, body)[DG_BI_STATE_OFF]; real* ex = body_ext(s, st, body);
      v3 t = vadd(vcross(vsub(P, V(bs[0], bs[1], bs[2])), F), T);
      ex[0] += F.x; ex[1] += F.y; ex[2] += F.z; ex[3] += t.x; ex[4] += t.y; ex[5] += t.z;
    }
    if (frame >= 0) {
      BodyWS* ws = (BodyWS*)malloc(sizeof(BodyWS)); body_kinematics(s, st, body, ws, NULL);
      for (int k = s->FI[frame * DG_FI_STRIDE + DG_FI_LINK]; k >= 0; k = link_i(s, k)[DG_LI_PARENT]) {
        const real* lf = link_f(s, k); int lk = k - ws->first;
        v3 axw = mv(&ws->Rw[lk], V(lf[DG_LF_AXIS], lf[DG_LF_AXIS + 1], lf[DG_LF_AXIS + 2]));
        real tau = link_i(s, k)[DG_LI_TYPE] == 0 ? vdot(axw, vadd(vcross(vsub(P, ws->pw[lk]), F), T)) : vdot(axw, F);
        st[link_i(s, k)[DG_LI_STATE_OFF] + DG_LS_TORQUE] += tau;
      }
      free(ws);
    }
  }
  return 0;
}
static void set_base_com_pose(const Scene* s, real* st, int b, v3 pc, qt qc) {
  /* p.resetBasePositionAndOrientation takes the pose of the root inertial frame and zeroes the velocity [R] */
  const real* bf = body_f(s, b); real* bs = st + body_i(s, b)[DG_BI_STATE_OFF];
  qt qr = {bf[DG_BF_REPORT_QUAT], bf[DG_BF_REPORT_QUAT + 1], bf[DG_BF_REPORT_QUAT + 2], bf[DG_BF_REPORT_QUAT + 3]};
  qt ql = qnormalize(qmul(qc, qconj(qr))); m3 R = qmat(ql);
  v3 pl = vsub(pc, mv(&R, V(bf[DG_BF_REPORT_POS], bf[DG_BF_REPORT_POS + 1], bf[DG_BF_REPORT_POS + 2])));
  bs[0] = pl.x; bs[1] = pl.y; bs[2] = pl.z; bs[3] = ql.x; bs[4] = ql.y; bs[5] = ql.z; bs[6] = ql.w;
  if (!body_fixed(s, b)) for (int k = 0; k < 6; k++) bs[DG_BS_LINVEL + k] = 0.0;
}
static void run_reset_ops(dgo_world* w, int env) {
  Scene* s = &w->sc; real* st = env_state(w, env);
  uint64_t episode = (uint64_t)st[DG_ST_EPISODE];
  for (int op = 0; op < s->nops; op++) {
    const int32_t* oi = s->OI + op * DG_OI_STRIDE; const real* of = s->OF + op * DG_OF_STRIDE; int code = oi[DG_OI_CODE];
    if (code == DG_OP_RESPAWN) { /* respawn.py:31-39 */
      uint64_t ep = (oi[DG_OI_FLAGS] & DG_RS_ONCE) ? 0 : episode + 1, ge = (uint64_t)(w->env_base + env);
      real u[6]; for (int k = 0; k < 6; k++) u[k] = rng_uniform(w->seed, ge, ep, (uint64_t)op, (uint64_t)k) - 0.5;
      v3 p = V(of[0] + u[0] * of[7], of[1] + u[1] * of[8], of[2] + u[2] * of[9]);
      qt q0 = {of[3], of[4], of[5], of[6]};
      qt q = qmul(q0, qfrom_euler(u[3] * of[10], u[4] * of[11], u[5] * of[12]));
      set_base_com_pose(s, st, oi[DG_OI_BODY], p, q);
    } else if (code == DG_OP_RESET_JOINTS) { /* joint_controller.py:36-38 */
      const int32_t* il = s->IL + oi[DG_OI_ILIST]; const real* fl = s->FL + oi[DG_OI_FLIST];
      for (int k = 0; k < oi[DG_OI_N]; k++) { real* ls = st + link_i(s, il[k])[DG_LI_STATE_OFF]; ls[DG_LS_Q] = fl[k]; ls[DG_LS_QD] = 0.0; }
    } else if (code == DG_OP_RANDOMIZE_COLOR) { /* visual_randomizer.py:40-46 (a procedural texture instead of an image: DG_TX_*) */
      real* ps = st + s->addon_off + oi[DG_OI_STATE_OFF]; const uint64_t ge = (uint64_t)(w->env_base + env);
      real u[DG_TX_STRIDE]; for (int k = 0; k < DG_TX_STRIDE; k++) u[k] = rng_uniform(w->seed, ge, episode + 1, (uint64_t)op, (uint64_t)k);
      for (int k = 0; k < 6; k++) ps[k] = u[k];
      ps[DG_TX_FREQ] = 2.0 + 14.0 * u[6]; ps[DG_TX_KIND] = (real)(1 + (int)(3.0 * u[7]));
    } else if (code == DG_OP_RANDOMIZE_DYNAMICS) { /* dynamics_randomizer.py:24-32 */
      const real* fl = s->FL + oi[DG_OI_FLIST]; const int n = oi[DG_OI_N]; real* ps = st + s->addon_off + oi[DG_OI_STATE_OFF];
      const uint64_t ge = (uint64_t)(w->env_base + env);
      /* the reference draws in __init__ and again in the constructor's reset(): two rounds at an env's first reset */
      for (int round = (episode == 0 ? 0 : 1); round < 2; round++) {
        const uint64_t ep = round == 0 ? 0 : episode + 1;
        for (int k = 0; k < n; k++) {
          const real um = of[0] + (of[1] - of[0]) * rng_uniform(w->seed, ge, ep, (uint64_t)op, (uint64_t)(2 * k));
          const real ud = of[2] + (of[3] - of[2]) * rng_uniform(w->seed, ge, ep, (uint64_t)op, (uint64_t)(2 * k + 1));
          /* new mass = log(U) * CURRENT mass (guards: |log U|, accumulated scale clamped to [of[4], of[5]]) */
          real sc_ = ps[k] * fabs(log(um)); sc_ = sc_ < of[4] ? of[4] : (sc_ > of[5] ? of[5] : sc_); ps[k] = sc_;
          /* angularDamping = log(U) * URDF joint damping: body-wide, the last joint's value stays (guard: >= 0) */
          real da = log(ud) * fl[k]; ps[n] = da > 0.0 ? da : 0.0;
        }
      }
    }
  }
  for (int b = 0; b < s->nb; b++) { /* force_torque_sensor: no acceleration across a reset */
    const int32_t* bi = body_i(s, b); int po = bi[DG_BI_PREV_OFF]; if (po < 0) continue;
    for (int i = 0; i < bi[DG_BI_N_LINKS]; i++) st[po + i] = st[link_i(s, bi[DG_BI_FIRST_LINK] + i)[DG_LI_STATE_OFF] + DG_LS_QD];
    if (!body_fixed(s, b)) for (int k = 0; k < 6; k++) st[po + bi[DG_BI_N_LINKS] + k] = st[bi[DG_BI_STATE_OFF] + DG_BS_LINVEL + k];
  }
  if (s->warm_off >= 0) st[s->warm_off] = 0.0; /* a reset teleports bodies: no contact persists across it */
  st[DG_ST_EPISODE] = (real)(episode + 1);
}
static real reach_dist(const Scene* s, const real* st, const int32_t* oi) { /* reach_target.py:21-30 */
  FrameState a, b;
  /* link frames use getLinkState item 4 (URDF frame); bases use the reported (inertial) position */
  frame_state(s, st, oi[DG_OI_BODY2], oi[DG_OI_FRAME2], oi[DG_OI_FRAME2] < 0, NULL, &a);
  frame_state(s, st, oi[DG_OI_BODY], oi[DG_OI_FRAME], oi[DG_OI_FRAME] < 0, NULL, &b);
  return vnorm(vsub(b.p, a.p));
}
/* ---- force_torque_sensor.py:14-23 -------------------------------------------------------------------------------
 * Reaction wrench across the joint of frame `fr`: what the parent side exerts on the child side, by Newton-Euler in the
 * world frame over the child side with the accelerations of the last substep, (v_end - v_start) / h, minus gravity and
 * that substep's contact forces on the child side; reported in the child link's inertial frame, torque about its origin
 * [R: Bullet's joint feedback is I^A a + Z^A of the child link in its own (inertial) frame]. */
typedef struct { m3 R; v3 p, w, v, al, a; } LinkMotion;
static void link_motion(const Scene* s, const real* st, int b, const BodyWS* ws, int lk /* local, -1 base */, LinkMotion* o) {
  const int32_t* bi = body_i(s, b); int po = bi[DG_BI_PREV_OFF]; real h = s->h;
  o->R = ws->R0; o->p = ws->p0; o->w = o->v = o->al = o->a = V(0, 0, 0);
  if (!ws->fixed) {
    const real* bs = st + bi[DG_BI_STATE_OFF]; const real* pv = st + po + ws->n;
    o->v = V(bs[DG_BS_LINVEL], bs[DG_BS_LINVEL + 1], bs[DG_BS_LINVEL + 2]); o->w = V(bs[DG_BS_ANGVEL], bs[DG_BS_ANGVEL + 1], bs[DG_BS_ANGVEL + 2]);
    o->a = vscale(vsub(o->v, V(pv[0], pv[1], pv[2])), 1.0 / h); o->al = vscale(vsub(o->w, V(pv[3], pv[4], pv[5])), 1.0 / h);
  }
  int path[MAXL], np = 0; for (int k = lk; k >= 0; k = ws->parent[k]) path[np++] = k;
  for (int t = np - 1; t >= 0; t--) {
    int j = path[t]; int gl = ws->first + j; const real* lf = link_f(s, gl);
    real qd = st[link_i(s, gl)[DG_LI_STATE_OFF] + DG_LS_QD], qdd = (qd - st[po + j]) / h;
    v3 ax = mv(&ws->Rw[j], V(lf[DG_LF_AXIS], lf[DG_LF_AXIS + 1], lf[DG_LF_AXIS + 2])), r = vsub(ws->pw[j], o->p);
    v3 a1 = vadd(o->a, vadd(vcross(o->al, r), vcross(o->w, vcross(o->w, r)))), v1 = vadd(o->v, vcross(o->w, r));
    if (link_i(s, gl)[DG_LI_TYPE] == 0) { o->al = vadd(o->al, vadd(vscale(ax, qdd), vcross(o->w, vscale(ax, qd)))); o->a = a1; o->v = v1; o->w = vadd(o->w, vscale(ax, qd)); }
    else { o->a = vadd(a1, vadd(vscale(ax, qdd), vscale(vcross(o->w, vscale(ax, qd)), 2.0))); o->v = vadd(v1, vscale(ax, qd)); }
    o->R = ws->Rw[j]; o->p = ws->pw[j];
  }
}
static void ft_add_part(const Scene* s, const LinkMotion* m, real mass, v3 c, const m3* Ic, v3 ps, v3* F, v3* T) {
  v3 rc = mv(&m->R, c), pc = vadd(m->p, rc);
  v3 ac = vadd(m->a, vadd(vcross(m->al, rc), vcross(m->w, vcross(m->w, rc))));
  v3 f = vscale(vsub(ac, s->g), mass);
  m3 Rt; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt.m[i][j] = m->R.m[j][i];
  m3 t = mmul(&m->R, Ic), Iw = mmul(&t, &Rt);
  v3 nt = vadd(mv(&Iw, m->al), vcross(m->w, mv(&Iw, m->w)));
  *F = vadd(*F, f); *T = vadd(*T, vadd(nt, vcross(vsub(pc, ps), f)));
}
static void ft_wrench(dgo_world* w, int env, const int32_t* oi, int with_contacts, real* out6) {
  Scene* s = &w->sc; const real* st = env_state(w, env); int b = oi[DG_OI_BODY], fr = oi[DG_OI_FRAME];
  BodyWS* ws = (BodyWS*)malloc(sizeof(BodyWS)); body_kinematics(s, st, b, ws, NULL);
  const int32_t* il = s->IL + oi[DG_OI_ILIST]; const real* fl = s->FL + oi[DG_OI_FLIST]; const real* ff = s->FF + fr * DG_FF_STRIDE;
  int ga = s->FI[fr * DG_FI_STRIDE + DG_FI_LINK], la = ga < 0 ? -1 : ga - ws->first;
  LinkMotion ma; link_motion(s, st, b, ws, la, &ma);
  qt qo = {ff[DG_FF_COM_QUAT], ff[DG_FF_COM_QUAT + 1], ff[DG_FF_COM_QUAT + 2], ff[DG_FF_COM_QUAT + 3]}; m3 Ro = qmat(qo);
  m3 Rs = mmul(&ma.R, &Ro); v3 ps = vadd(ma.p, mv(&ma.R, V(ff[DG_FF_COM_POS], ff[DG_FF_COM_POS + 1], ff[DG_FF_COM_POS + 2])));
  v3 F = V(0, 0, 0), T = V(0, 0, 0);
  if (oi[DG_OI_FLAGS] & DG_FT_WHOLE_LINK) {
    const real* lf = link_f(s, ga); real ms = link_mass_scale(s, st, ga); m3 Ic = msym6(lf + DG_LF_INERTIA);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ic.m[r][c] *= ms;
    ft_add_part(s, &ma, lf[DG_LF_MASS] * ms, V(lf[DG_LF_COM], lf[DG_LF_COM + 1], lf[DG_LF_COM + 2]), &Ic, ps, &F, &T);
  } else if (fl[0] > 0.0) {
    real ms = ga >= 0 ? link_mass_scale(s, st, ga) : 1.0; m3 Ic = msym6(fl + 4);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ic.m[r][c] *= ms;
    ft_add_part(s, &ma, fl[0] * ms, V(fl[1], fl[2], fl[3]), &Ic, ps, &F, &T);
  }
  int nm = il[0];
  for (int k = 0; k < nm; k++) {
    int gl = il[1 + k]; const real* lf = link_f(s, gl); real ms = link_mass_scale(s, st, gl); m3 Ic = msym6(lf + DG_LF_INERTIA);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ic.m[r][c] *= ms;
    LinkMotion mk; link_motion(s, st, b, ws, gl - ws->first, &mk);
    ft_add_part(s, &mk, lf[DG_LF_MASS] * ms, V(lf[DG_LF_COM], lf[DG_LF_COM + 1], lf[DG_LF_COM + 2]), &Ic, ps, &F, &T);
  }
  if (with_contacts) {
    int nsh = il[1 + nm]; const int32_t* shp = il + 2 + nm;
    for (int k = 0; k < w->last_contacts[env]; k++) {
      const Contact* c = &w->last_cs[(size_t)env * MAXC + k]; int inA = 0, inB = 0;
      for (int q = 0; q < nsh; q++) { if (shp[q] == c->shape_a) inA = 1; if (shp[q] == c->shape_b) inB = 1; }
      if (inA == inB) continue;
      /* the rows push side A along +dir and side B along -dir */
      v3 f = vscale(vadd(vscale(c->n, c->imp[0]), vadd(vscale(c->t1, c->imp[1]), vscale(c->t2, c->imp[2]))), (inA ? 1.0 : -1.0) / s->h);
      F = vsub(F, f); T = vsub(T, vcross(vsub(c->p, ps), f));
    }
  }
  v3 Fl = mtv(&Rs, F), Tl = mtv(&Rs, T);
  out6[0] = Fl.x; out6[1] = Fl.y; out6[2] = Fl.z; out6[3] = Tl.x; out6[4] = Tl.y; out6[5] = Tl.z;
  free(ws);
}
/* ft_mode: 0 = after a step (the last substep's contacts count), 1 = plain observe (no contact term), 2 = leave the
 * force/torque columns as they are (envs a masked reset did not touch) */
static void run_output_ops(dgo_world* w, int env, real* obs, real* rew, uint8_t* term, real* rew_sum, uint8_t* term_flag, int ft_mode) {
  Scene* s = &w->sc; const real* st = env_state(w, env);
  real rsum = 0; int gany[64]; memset(gany, 0, sizeof gany); int any = 0;
  for (int op = 0; op < s->nops; op++) {
    const int32_t* oi = s->OI + op * DG_OI_STRIDE; const real* of = s->OF + op * DG_OF_STRIDE;
    int code = oi[DG_OI_CODE], io = oi[DG_OI_IO_OFF]; const int32_t* il = s->IL + oi[DG_OI_ILIST];
    if (code == DG_OP_OBS_JOINT_STATE) { /* joint_state_sensor.py:47-57 */
      int n = oi[DG_OI_N], k2 = n;
      for (int k = 0; k < n; k++) if (obs) obs[io + k] = st[link_i(s, il[k])[DG_LI_STATE_OFF] + DG_LS_Q];
      if (oi[DG_OI_FLAGS] & DG_JS_VELOCITY) { for (int k = 0; k < n; k++) if (obs) obs[io + k2 + k] = st[link_i(s, il[k])[DG_LI_STATE_OFF] + DG_LS_QD]; k2 += n; }
      if (oi[DG_OI_FLAGS] & DG_JS_EFFORT) for (int k = 0; k < n; k++) if (obs) obs[io + k2 + k] = st[link_i(s, il[k])[DG_LI_STATE_OFF] + DG_LS_APPLIED];
    } else if (code == DG_OP_OBS_OBJECT_STATE) { /* object_state_sensor.py:32-75: inertial-frame states (items 0,1,6,7) */
      FrameState t; frame_state(s, st, oi[DG_OI_BODY], oi[DG_OI_FRAME], 1, NULL, &t);
      if (oi[DG_OI_BODY2] >= 0) {
        FrameState sr; frame_state(s, st, oi[DG_OI_BODY2], oi[DG_OI_FRAME2], 1, NULL, &sr);
        t.p = vsub(t.p, sr.p); t.v = vsub(t.v, sr.v); t.q = qmul(sr.q, t.q); t.w = vsub(t.w, sr.w);
      }
      if (obs) {
        int k = io; obs[k++] = t.p.x; obs[k++] = t.p.y; obs[k++] = t.p.z;
        /* output order = insertion order of the dict built in observe(), object_state_sensor.py:64-73:
         * position, velocity, rotation, angular_velocity */
        if (oi[DG_OI_FLAGS] & DG_OS_VELOCITY) { obs[k++] = t.v.x; obs[k++] = t.v.y; obs[k++] = t.v.z; }
        if (oi[DG_OI_FLAGS] & DG_OS_ROTATION) { v3 e = euler_from_q(t.q); obs[k++] = e.x; obs[k++] = e.y; obs[k++] = e.z; }
        if ((oi[DG_OI_FLAGS] & DG_OS_ROTATION) && (oi[DG_OI_FLAGS] & DG_OS_VELOCITY)) { obs[k++] = t.w.x; obs[k++] = t.w.y; obs[k++] = t.w.z; }
      }
    } else if (code == DG_OP_OBS_ADDON_STATE) {
      for (int k = 0; k < oi[DG_OI_N]; k++) if (obs) obs[io + k] = st[s->addon_off + oi[DG_OI_STATE_OFF] + k];
    } else if (code == DG_OP_OBS_FT) {
      if (obs && ft_mode != 2) ft_wrench(w, env, oi, ft_mode == 0, obs + io);
    } else if (code == DG_OP_REW_REACH) { real r = -reach_dist(s, st, oi) * of[0]; if (rew) rew[io] = r; rsum += r; }
    else if (code == DG_OP_REW_ELECTRICITY) { /* electricity_cost.py:15-18 */
      const int32_t* bi = body_i(s, oi[DG_OI_BODY]); real acc = 0;
      for (int i = 0; i < bi[DG_BI_N_LINKS]; i++) { const real* ls = st + link_i(s, bi[DG_BI_FIRST_LINK] + i)[DG_LI_STATE_OFF]; acc += fabs(ls[DG_LS_APPLIED] * ls[DG_LS_QD]); }
      real r = -acc * of[0]; if (rew) rew[io] = r; rsum += r;
    } else if (code == DG_OP_REW_CONST) { if (rew) rew[io] = of[0]; rsum += of[0]; }
    else if (code == DG_OP_TERM_REACH || code == DG_OP_TERM_TILT || code == DG_OP_TERM_TIMER) {
      int t = 0;
      if (code == DG_OP_TERM_REACH) t = reach_dist(s, st, oi) < of[1];
      else if (code == DG_OP_TERM_TILT) { /* drone_pilot.py:53-55 */
        FrameState f; frame_state(s, st, oi[DG_OI_BODY], -1, 1, NULL, &f);
        t = 2.0 * atan2(sqrt(f.q.x * f.q.x + f.q.y * f.q.y + f.q.z * f.q.z), fabs(f.q.w)) > of[0];
      } else t = st[DG_ST_STEP] >= of[0]; /* diy_gym.py:180-183 */
      if (term) term[io] = (uint8_t)t;
      if (t) { any = 1; gany[oi[DG_OI_SLOT] & 63] = 1; }
    }
  }
  if (rew_sum) *rew_sum = rsum;
  if (term_flag) {
    if (s->term_mode == DG_COLLAPSE_ALL) { int all = s->n_term_groups > 0; for (int g = 0; g < s->n_term_groups; g++) all = all && gany[g]; *term_flag = (uint8_t)all; }
    else *term_flag = (uint8_t)any;
  }
}

static void sim_step(dgo_world* w, int env) {
  Scene* s = &w->sc; real* st = env_state(w, env);
  for (int k = 0; k < s->substeps; k++) substep(w, env, k == s->substeps - 1);
  /* external wrenches and joint torques last for one stepSimulation [R] */
  for (int b = 0; b < s->nb; b++) { if (body_i(s, b)[DG_BI_FLAGS] & DG_BODY_FROZEN) continue; real* ex = body_ext(s, st, b); for (int k = 0; k < 6; k++) ex[k] = 0.0; }
  for (int l = 0; l < s->nl; l++) st[link_i(s, l)[DG_LI_STATE_OFF] + DG_LS_TORQUE] = 0.0;
}

static int observe_all(dgo_world* w, real* obs, real* rew, uint8_t* term, real* rew_sum, uint8_t* term_flag, int ft_mode, const uint8_t* fresh) {
  Scene* s = &w->sc;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int e = 0; e < w->B; e++)
    run_output_ops(w, e, obs ? obs + (size_t)e * s->obs_dim : NULL, rew ? rew + (size_t)e * s->rew_dim : NULL,
                   term ? term + (size_t)e * s->term_dim : NULL, rew_sum ? rew_sum + e : NULL, term_flag ? term_flag + e : NULL,
                   (fresh && !fresh[e]) ? 2 : ft_mode);
  return 0;
}
int dgo_observe(dgo_world* w, real* obs, real* rew, uint8_t* term, real* rew_sum, uint8_t* term_flag) {
  return observe_all(w, obs, rew, term, rew_sum, term_flag, 1, NULL);
}
/* reference diy_gym.py:130-148 */
int dgo_reset(dgo_world* w, const uint8_t* mask, real* obs) {
  Scene* s = &w->sc;
  for (int e = 0; e < w->B; e++) {
    if (mask && !mask[e]) continue;
    real* st = env_state(w, e);
    st[DG_ST_STEP] = 0.0;
    run_reset_ops(w, e);
    for (int k = 0; k < s->hot_start; k++) sim_step(w, e);
  }
  if (obs) observe_all(w, obs, NULL, NULL, NULL, NULL, s->hot_start > 0 ? 0 : 1, mask);
  return 0;
}
/* reference diy_gym.py:187-209 */
int dgo_step(dgo_world* w, const real* actions, uint64_t update_mask, real* obs, real* rew, uint8_t* term, real* rew_sum,
             uint8_t* term_flag) {
  Scene* s = &w->sc;
  /* envs are independent; the motor table is uniform over envs and every env writes the same values
   * into it, so it is brought up to date by env 0 first and the remaining envs can run in parallel */
  int e0 = 0;
  if (w->B > 0) {
    real* st = env_state(w, 0);
    if (actions) run_update_ops(w, 0, actions, update_mask);
    st[DG_ST_STEP] += 1.0; sim_step(w, 0); e0 = 1;
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int e = e0; e < w->B; e++) {
    real* st = env_state(w, e);
    if (actions) run_update_ops(w, e, actions + (size_t)e * s->act_dim, update_mask);
    st[DG_ST_STEP] += 1.0;
    sim_step(w, e);
  }
  return observe_all(w, obs, rew, term, rew_sum, term_flag, 0, NULL);
}

/* --------------------------------------------------------------- camera */
/* Restates Camera.observe (camera.py:58-92): view = inv(T_world_parent T_parent_cam), OpenGL eye frame
 * (looks down -z, +y up), gluPerspective(fov, aspect = res[0]/res[1], near, far); the reference's depth
 * formula (:82-85) returns the eye-space z of the nearest surface (negative; -far where nothing is hit).
 * pybullet's DIRECT-mode renderer draws the VISUAL meshes with lighting; this restatement ray-casts the
 * collision geometry, so rgb is not a parity quantity. */
/* Near plane: a surface the ray ENTERS nearer than `tmin` (= the camera's near distance; t is eye-space depth, the rays
 * have z = -1) neither shows nor hides anything -- what clipping at the near plane does in a rasteriser.  In particular
 * the link a camera is mounted on (its eye sits ON a face of that link's hull, at t = +-1 ulp) cannot blank the picture. */
typedef struct { real t; v3 n; int shape; real tmin; } RayHit;
static void ray_sphere(v3 o, v3 d, v3 c, real r, RayHit* h, int sh) {
  v3 oc = vsub(o, c); real a = vdot(d, d), b = vdot(oc, d), cc = vdot(oc, oc) - r * r, disc = b * b - a * cc;
  if (disc < 0) return;
  real t = (-b - sqrt(disc)) / a;
  if (t >= h->tmin && t < h->t) { h->t = t; h->n = vscale(vsub(vadd(o, vscale(d, t)), c), 1.0 / r); h->shape = sh; }
}
static void ray_slabs(v3 o, v3 d, const real* hx, real* tn, real* tf, int* axis, real* sgn) {
  real oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}; *tn = -HUGE_R; *tf = HUGE_R; *axis = 0; *sgn = 1;
  for (int k = 0; k < 3; k++) {
    if (fabs(dd[k]) < TINY_R) { if (fabs(oo[k]) > hx[k]) { *tn = HUGE_R; *tf = -HUGE_R; } continue; }
    real t1 = (-hx[k] - oo[k]) / dd[k], t2 = (hx[k] - oo[k]) / dd[k], s = -1;
    if (t1 > t2) { real tt = t1; t1 = t2; t2 = tt; s = 1; }
    if (t1 > *tn) { *tn = t1; *axis = k; *sgn = s; }
    if (t2 < *tf) *tf = t2;
  }
}
static void ray_box(v3 o, v3 d, const m3* R, v3 p, const real* hx, RayHit* h, int sh) {
  v3 ol = mtv(R, vsub(o, p)), dl = mtv(R, d); real tn, tf, sg; int ax;
  ray_slabs(ol, dl, hx, &tn, &tf, &ax, &sg);
  if (tn > tf || tn < h->tmin || tn >= h->t) return;
  v3 nl = V(ax == 0 ? sg : 0, ax == 1 ? sg : 0, ax == 2 ? sg : 0);
  h->t = tn; h->n = mv(R, nl); h->shape = sh;
}
static void ray_capsule(v3 o, v3 d, v3 e0, v3 e1, real r, RayHit* h, int sh) {
  v3 ax = vsub(e1, e0); real L2 = vdot(ax, ax);
  if (L2 > 1e-24) {
    v3 oc = vsub(o, e0); real dax = vdot(d, ax), oax = vdot(oc, ax);
    real a = vdot(d, d) - dax * dax / L2, b = vdot(oc, d) - oax * dax / L2, c = vdot(oc, oc) - oax * oax / L2 - r * r, disc = b * b - a * c;
    if (a > 1e-24 && disc >= 0) {
      real t = (-b - sqrt(disc)) / a, s = (oax + t * dax) / L2;
      if (t >= h->tmin && t < h->t && s >= 0 && s <= 1) {
        v3 pt = vadd(o, vscale(d, t)); v3 q = vadd(e0, vscale(ax, s));
        h->t = t; h->n = vscale(vsub(pt, q), 1.0 / r); h->shape = sh;
      }
    }
  }
  ray_sphere(o, d, e0, r, h, sh); ray_sphere(o, d, e1, r, h, sh);
}
static void ray_hull(v3 o, v3 d, const m3* Rl, v3 pl, const real* planes, int np, RayHit* h, int sh) {
  v3 ol = mtv(Rl, vsub(o, pl)), dl = mtv(Rl, d); real tn = -HUGE_R, tf = HUGE_R; v3 nn = V(0, 0, 1);
  for (int k = 0; k < np; k++) {
    const real* pp = planes + 4 * k; v3 n = V(pp[0], pp[1], pp[2]);
    real den = vdot(n, dl), dist = vdot(n, ol) + pp[3];
    if (fabs(den) < TINY_R) { if (dist > 0) return; continue; }
    real t = -dist / den;
    if (den < 0) { if (t > tn) { tn = t; nn = n; } } else if (t < tf) tf = t;
  }
  if (np == 0 || tn > tf || tn < h->tmin || tn >= h->t) return;
  h->t = tn; h->n = mv(Rl, nn); h->shape = sh;
}
int dgo_render(dgo_world* w, int32_t camera, real* rgb, real* depth, int32_t* seg) {
  Scene* s = &w->sc; const int32_t* I = s->I; const real* F = s->F;
  if (camera < 0 || camera >= I[DG_H_N_CAMERAS]) { set_err("camera %d out of range", camera); return -1; }
  const int32_t* ci = I + I[DG_H_OFF_CAMERA_I] + camera * DG_CI_STRIDE; const real* cf = F + I[DG_H_OFF_CAMERA_F] + camera * DG_CF_STRIDE;
  const real* PLN = F + I[DG_H_OFF_PLANE_F];
  const int W = ci[DG_CI_WIDTH], Hh = ci[DG_CI_HEIGHT]; const real fov = cf[DG_CF_FOV], zn = cf[DG_CF_NEAR], zf = cf[DG_CF_FAR];
  const real tanh2 = tan(0.5 * fov * M_PI / 180.0), aspect = (real)W / (real)Hh;
  const v3 light = V(0.30151134457776363, 0.30151134457776363, 0.9045340337332909);
  for (int e = 0; e < w->B; e++) {
    const real* st = env_state(w, e);
    BodyWS* wsb = (BodyWS*)malloc(sizeof(BodyWS) * (size_t)s->nb);
    for (int b = 0; b < s->nb; b++) body_kinematics(s, st, b, &wsb[b], NULL);
    /* camera pose */
    m3 Rp = mident(); v3 pp = V(0, 0, 0);
    if (ci[DG_CI_BODY] >= 0) {
      FrameState f; frame_state(s, st, ci[DG_CI_BODY], ci[DG_CI_FRAME], ci[DG_CI_FRAME] < 0, NULL, &f); /* link: item 4,5; base: reported pose */
      Rp = qmat(f.q); pp = f.p;
    }
    qt qc = {cf[DG_CF_QUAT], cf[DG_CF_QUAT + 1], cf[DG_CF_QUAT + 2], cf[DG_CF_QUAT + 3]}; m3 Rc0 = qmat(qc);
    m3 Rc = mmul(&Rp, &Rc0); v3 pc = vadd(pp, mv(&Rp, V(cf[DG_CF_POS], cf[DG_CF_POS + 1], cf[DG_CF_POS + 2])));
    WShape* shp = (WShape*)malloc(sizeof(WShape) * (size_t)(s->nsh > 0 ? s->nsh : 1));
    for (int k = 0; k < s->nsh; k++) shape_world(s, wsb, k, &shp[k]);
    for (int row = 0; row < Hh; row++) for (int col = 0; col < W; col++) {
      const real xn = ((col + 0.5) / W) * 2.0 - 1.0, yn = 1.0 - ((row + 0.5) / Hh) * 2.0;
      v3 d = mv(&Rc, V(xn * tanh2 * aspect, yn * tanh2, -1.0));
      RayHit h; h.t = zf; h.shape = -1; h.n = V(0, 0, 1); h.tmin = zn;
      for (int k = 0; k < s->nsh; k++) {
        const WShape* a = &shp[k]; const int32_t* si = s->SI + k * DG_SI_STRIDE;
        if (a->type == DG_SHAPE_SPHERE) ray_sphere(pc, d, a->p, a->prm[0], &h, k);
        else if (a->type == DG_SHAPE_BOX) ray_box(pc, d, &a->R, a->p, a->prm, &h, k);
        else if (a->type == DG_SHAPE_CAPSULE) { v3 e0, e1; seg_ends(a, &e0, &e1); ray_capsule(pc, d, e0, e1, a->prm[0], &h, k); }
        else ray_hull(pc, d, &a->Rl, a->pl, PLN + 4 * si[DG_SI_PLANE_OFF], si[DG_SI_N_PLANES], &h, k);
      }
      const int hit = h.shape >= 0; const size_t px = (size_t)e * W * Hh + (size_t)row * W + col;
      if (depth) depth[px] = hit ? -h.t : -zf;
      if (seg) {
        if (!hit) seg[px] = -1;
        else { const int32_t* si = s->SI + h.shape * DG_SI_STRIDE; seg[px] = si[DG_SI_BODY] + ((((si[DG_SI_FLAGS] >> 8) & 0xFFFF)) << 24); }
      }
      if (rgb) {
        real c[3] = {0.75, 0.75, 0.75};
        if (hit) {
          const int hb_ = s->SI[h.shape * DG_SI_STRIDE + DG_SI_BODY]; const int co_ = body_i(s, hb_)[DG_BI_COLOR_OFF];
          real col3[3]; for (int k = 0; k < 3; k++) col3[k] = s->SF[h.shape * DG_SF_STRIDE + DG_SF_COLOR + k]; /* URDF material / YAML colour */
          if (co_ >= 0) { /* per-env texture of a visual_randomizer, evaluated in the shape's reference frame */
            const real* tx = st + co_; const int kind = (int)tx[DG_TX_KIND]; real t = 0;
            if (kind != DG_TEX_FLAT) {
              const WShape* a = &shp[h.shape]; const m3* Rr = a->type == DG_SHAPE_POINTS ? &a->Rl : &a->R; v3 pr = a->type == DG_SHAPE_POINTS ? a->pl : a->p;
              v3 pl_ = mtv(Rr, vsub(vadd(pc, vscale(d, h.t)), pr));
              const int32_t ux = (int32_t)floor(pl_.x * tx[DG_TX_FREQ]), uy = (int32_t)floor(pl_.y * tx[DG_TX_FREQ]), uz = (int32_t)floor(pl_.z * tx[DG_TX_FREQ]);
              if (kind == DG_TEX_CHECKER) t = ((ux + uy + uz) & 1) ? 1 : 0;
              else if (kind == DG_TEX_STRIPES) t = (ux & 1) ? 1 : 0;
              else { uint32_t hh; DG_TEX_HASH(ux, uy, uz, hh); t = (real)hh * (1.0 / 16777216.0); }
            }
            for (int k = 0; k < 3; k++) col3[k] = tx[DG_TX_A + k] + (tx[DG_TX_B + k] - tx[DG_TX_A + k]) * t;
          }
          real nl = vdot(h.n, light); real sh = 0.4 + 0.6 * (nl > 0 ? nl : 0);
          for (int k = 0; k < 3; k++) c[k] = col3[k] * sh;
        }
        for (int k = 0; k < 3; k++) rgb[3 * px + k] = c[k];
      }
    }
    free(shp); free(wsb);
  }
  return 0;
}

int dgo_forward_dynamics(dgo_world* w, int32_t env, int32_t body, real* qdd_out, real* base_acc6_out) {
  Scene* s = &w->sc; real* st = env_state(w, env);
  BodyWS* ws = (BodyWS*)malloc(sizeof(BodyWS));
  body_kinematics(s, st, body, ws, NULL); body_velocities(s, st, body, ws); body_aba(s, st, body, ws);
  for (int i = 0; i < ws->n; i++) qdd_out[i] = ws->qdd[i];
  if (base_acc6_out) for (int k = 0; k < 6; k++) base_acc6_out[k] = ws->a0.v[k];
  free(ws); return 0;
}
int dgo_unit_response(dgo_world* w, int32_t env, int32_t body, int32_t dof, real* dv_out) {
  Scene* s = &w->sc; real* st = env_state(w, env);
  BodyWS* ws = (BodyWS*)malloc(sizeof(BodyWS)); real J[MAXV];
  body_kinematics(s, st, body, ws, NULL); body_velocities(s, st, body, ws); body_aba(s, st, body, ws);
  body_response(ws, -1, NULL, dof, J, dv_out);
  free(ws); return 0;
}
