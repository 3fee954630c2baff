// dg_api.hip -- kernels' entry points and the C-ABI (include/diygym_hip.h).
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared dg_api.hip -o libdiygym_hip.so
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/diygym_hip.h"
#include "dg_launch.h"
#define DG_DEFINE_RENDER_KERNEL
#include "dg_render.h"

using namespace dg;

namespace dg {
extern const LaunchTable g_launch_table_64, g_launch_table_32, g_launch_table_16, g_launch_table_8, g_launch_table_4, g_launch_table_1, g_launch_table_0, g_launch_table_g16;
const LaunchTable& launch_table(int lanes) { return lanes == 64 ? g_launch_table_64 : lanes == 32 ? g_launch_table_32 : lanes == 16 ? g_launch_table_16 : lanes == 8 ? g_launch_table_8 : lanes == 4 ? g_launch_table_4 : lanes == 1 ? g_launch_table_1 : lanes == -16 ? g_launch_table_g16 : g_launch_table_0; }
}  // namespace dg

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf; return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(DG_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

// Every entry point runs with the world's device current and puts the caller's device back on the way out (a torch
// process may have another device current; launches and frees must not land there).
struct DeviceGuard {
  int prev = -1; bool switched = false; hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define DG_ON_DEVICE(dev) DeviceGuard guard_(dev); if (guard_.err != hipSuccess) return fail(DG_ERR_HIP, "hipSetDevice(%d): %s", (dev), hipGetErrorString(guard_.err))

__global__ void init_state_kernel(const float* init, float* state, int state_dim, int stride) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x; if (e >= stride) return;
  for (int k = 0; k < state_dim; k++) state[(size_t)k * stride + e] = init[k];
}

// ------------------------------------------------------------------ world
struct dg_world {
  DevScene sc; MotorTable mt;
  std::vector<int32_t> I; std::vector<double> F;
  int device = 0, lanes = 64, lds_bytes = 0, num_envs = 0, stride = 0;
  void* d_blob_i = nullptr; void* d_blob_f = nullptr; void* d_plan = nullptr; float* d_init = nullptr;
  int32_t* diag = nullptr;
  unsigned long long* profile_cycles = nullptr;
  bool par = false;  // step runs as two wavefronts per workgroup (helper wave)
  bool no_par_reset = false;  // DG_NO_PAR_RESET: masked resets through reset_kernel<64> (one wavefront, generic solver)
  float* d_gws = nullptr;  // global scratch when the scene does not fit LDS (lanes == 0)
  float* d_hull_ws = nullptr;  // polytope workspace of the hull-hull narrow phase (dg_hull.h), one block per wavefront of the step grid
  int cu_count = 1;     // multiProcessorCount of `device`, read once in dg_world_create
  int render_diag = 0;  // DG_RENDER_NO_CULL / DG_RENDER_DIAG at creation (diagnostics), dg_world_set_render_diag later
  int render_wpe = 2;   // wavefronts per SIMD of the render kernel's build (DG_RENDER_WPE=3: the spilling build)
  int ncam = 0; float* d_render_table = nullptr; cip d_CI = nullptr; cfp d_CF = nullptr, d_PLN = nullptr;
  ~dg_world() {  // also the clean-up of a dg_world_create that failed half way
    DeviceGuard g(device);
    (void)hipFree(d_gws); (void)hipFree(d_hull_ws); (void)hipFree(d_render_table); (void)hipFree(d_blob_i); (void)hipFree(d_blob_f); (void)hipFree(d_plan); (void)hipFree(d_init);
  }
};

// Bounds of the blob's tables against the array lengths the caller passed: a malformed blob must fail here, not read
// out of bounds on the host or the device.
static const char* check_blob(const int32_t* I, int64_t n_i, int64_t n_f) {
  struct T { int off, count_idx, stride; bool is_f; const char* name; };
  const T tables[] = {
    {DG_H_OFF_BODY_I, DG_H_N_BODIES, DG_BI_STRIDE, false, "body ints"}, {DG_H_OFF_LINK_I, DG_H_N_LINKS, DG_LI_STRIDE, false, "link ints"},
    {DG_H_OFF_FRAME_I, DG_H_N_FRAMES, DG_FI_STRIDE, false, "frame ints"}, {DG_H_OFF_SHAPE_I, DG_H_N_SHAPES, DG_SI_STRIDE, false, "shape ints"},
    {DG_H_OFF_PAIR_I, DG_H_N_PAIRS, DG_PI_STRIDE, false, "pairs"}, {DG_H_OFF_GROUP_I, DG_H_N_GROUPS, DG_GI_STRIDE, false, "pair groups"},
    {DG_H_OFF_CAMERA_I, DG_H_N_CAMERAS, DG_CI_STRIDE, false, "camera ints"}, {DG_H_OFF_OP_I, DG_H_N_OPS, DG_OI_STRIDE, false, "op ints"},
    {DG_H_OFF_ILIST, DG_H_N_ILIST, 1, false, "int list"},
    {DG_H_OFF_BODY_F, DG_H_N_BODIES, DG_BF_STRIDE, true, "body floats"}, {DG_H_OFF_LINK_F, DG_H_N_LINKS, DG_LF_STRIDE, true, "link floats"},
    {DG_H_OFF_FRAME_F, DG_H_N_FRAMES, DG_FF_STRIDE, true, "frame floats"}, {DG_H_OFF_SHAPE_F, DG_H_N_SHAPES, DG_SF_STRIDE, true, "shape floats"},
    {DG_H_OFF_POINT_F, DG_H_N_POINTS, 3, true, "hull points"}, {DG_H_OFF_PLANE_F, DG_H_N_PLANES, 4, true, "hull planes"},
    {DG_H_OFF_CAMERA_F, DG_H_N_CAMERAS, DG_CF_STRIDE, true, "camera floats"}, {DG_H_OFF_OP_F, DG_H_N_OPS, DG_OF_STRIDE, true, "op floats"},
    {DG_H_OFF_FLIST, DG_H_N_FLIST, 1, true, "float list"},
    {DG_H_OFF_CONS_I, DG_H_N_CONSTRAINTS, DG_KI_STRIDE, false, "constraint ints"}, {DG_H_OFF_CONS_F, DG_H_N_CONSTRAINTS, DG_KF_STRIDE, true, "constraint floats"}};
  for (const T& t : tables) {
    const int64_t off = I[t.off], cnt = I[t.count_idx], lim = t.is_f ? n_f : n_i;
    if (cnt < 0 || off < (t.is_f ? DG_HF_FLOAT_COUNT : DG_H_INT_COUNT) || off + cnt * t.stride > lim) return t.name;
  }
  return nullptr;
}

extern "C" {

int32_t dg_version(void) { return (0 << 16) | 5; }
const char* dg_last_error(void) { return g_err.c_str(); }

int32_t dg_world_create(const int32_t* I, int64_t n_i, const double* F, int64_t n_f, int32_t num_envs, int32_t env_stride,
                        int32_t device, uint64_t seed, int64_t env_index_base, dg_world** out) {
  if (!I || !F || !out || n_i < DG_H_INT_COUNT) return fail(DG_ERR_ARG, "null or short scene arrays");
  if (I[DG_H_MAGIC] != DG_MAGIC || I[DG_H_VERSION] != DG_VERSION) return fail(DG_ERR_BAD_SCENE, "bad scene magic/version (%x, %d)", I[DG_H_MAGIC], I[DG_H_VERSION]);
  if (num_envs <= 0 || env_stride < num_envs) return fail(DG_ERR_ARG, "num_envs=%d env_stride=%d", num_envs, env_stride);
  const int nb = I[DG_H_N_BODIES], nl = I[DG_H_N_LINKS];
  if (nl > DG_MAX_LINKS) return fail(DG_ERR_UNSUPPORTED, "%d links > %d supported", nl, DG_MAX_LINKS);
  if (nb > DG_MAX_BODIES) return fail(DG_ERR_UNSUPPORTED, "%d bodies > %d supported", nb, DG_MAX_BODIES);
  if (n_f < DG_HF_FLOAT_COUNT) return fail(DG_ERR_BAD_SCENE, "float array shorter than its header");
  if (const char* bad = check_blob(I, n_i, n_f)) return fail(DG_ERR_BAD_SCENE, "scene table '%s' does not fit the arrays passed (n_i=%lld, n_f=%lld)", bad, (long long)n_i, (long long)n_f);
  if (I[DG_H_N_SHAPES] > 4096) return fail(DG_ERR_UNSUPPORTED, "%d shapes > 4096 supported", I[DG_H_N_SHAPES]);
  if (I[DG_H_N_CONSTRAINTS] > DG_MAX_CONSTRAINTS) return fail(DG_ERR_UNSUPPORTED, "%d fixed constraints > %d supported", I[DG_H_N_CONSTRAINTS], DG_MAX_CONSTRAINTS);
  for (int q = 0; q < I[DG_H_N_CONSTRAINTS]; q++) {
    const int32_t* ki = I + I[DG_H_OFF_CONS_I] + q * DG_KI_STRIDE;
    for (int k = 0; k < 2; k++) {
      const int b = ki[k == 0 ? DG_KI_BODY_A : DG_KI_BODY_B], gl = ki[k == 0 ? DG_KI_LINK_A : DG_KI_LINK_B];
      if (b < 0 || b >= nb) return fail(DG_ERR_BAD_SCENE, "constraint %d: body %d out of range", q, b);
      const int32_t* B = I + I[DG_H_OFF_BODY_I] + b * DG_BI_STRIDE;
      if (gl >= 0 && (gl < B[DG_BI_FIRST_LINK] || gl >= B[DG_BI_FIRST_LINK] + B[DG_BI_N_LINKS])) return fail(DG_ERR_BAD_SCENE, "constraint %d: link %d is not a link of body %d", q, gl, b);
    }
  }
  if (I[DG_H_N_TERM_GROUPS] > 64) return fail(DG_ERR_UNSUPPORTED, "%d receptors with terminal addons > 64 supported", I[DG_H_N_TERM_GROUPS]);
  { int ndev = 0; HIP_TRY(hipGetDeviceCount(&ndev)); if (device < 0 || device >= ndev) return fail(DG_ERR_ARG, "device %d out of range (%d visible)", device, ndev); }
  DG_ON_DEVICE(device);
  std::unique_ptr<dg_world> holder(new dg_world());  // every early return below frees what was allocated so far
  dg_world* w = holder.get();
  w->I.assign(I, I + n_i); w->F.assign(F, F + n_f); w->device = device; w->num_envs = num_envs; w->stride = env_stride;
  const int32_t* BI = I + I[DG_H_OFF_BODY_I]; const int32_t* LI = I + I[DG_H_OFF_LINK_I]; const int32_t* OI = I + I[DG_H_OFF_OP_I];
  // ---- LDS plan (slots per lane)
  std::vector<int32_t> plan((size_t)nb * PLB_STRIDE + (size_t)nl * PLL_STRIDE);
  plan.reserve(plan.size() + (size_t)I[DG_H_N_PAIRS] + 4 * (size_t)I[DG_H_N_GROUPS] + 1);  // pair descriptors are appended below; PLB / PLL must stay valid
  int32_t* PLB = plan.data(); int32_t* PLL = plan.data() + (size_t)nb * PLB_STRIDE;
  int slot = 0, nvmax = 0, nmax = 0; bool any_float = false;
  for (int b = 0; b < nb; b++) {
    const int32_t* B = BI + b * DG_BI_STRIDE; const bool fx = B[DG_BI_FLAGS] & DG_BODY_FIXED; const int n = B[DG_BI_N_LINKS];
    const int nv = (fx ? 0 : 6) + n;
    if (!fx) any_float = true;
    if (B[DG_BI_FLAGS] & DG_BODY_FROZEN) PLB[b * PLB_STRIDE + PLB_R0] = -1; else { PLB[b * PLB_STRIDE + PLB_R0] = slot; slot += 6; }
    PLB[b * PLB_STRIDE + PLB_MINV] = slot; slot += nv * nv;
    PLB[b * PLB_STRIDE + PLB_NV] = nv;
    { bool chain = fx && n >= 1 && n <= 6;
      for (int i = 0; i < n && chain; i++) chain = LI[(B[DG_BI_FIRST_LINK] + i) * DG_LI_STRIDE + DG_LI_PARENT] == (i == 0 ? -1 : B[DG_BI_FIRST_LINK] + i - 1);
      PLB[b * PLB_STRIDE + PLB_CHAIN] = chain ? 1 : 0; }
    nvmax = std::max(nvmax, nv); nmax = std::max(nmax, n);
  }
  // velocity-change blocks of all bodies back to back, then nv_max slots of padding (branch-free contact sweeps)
  for (int b = 0; b < nb; b++) { PLB[b * PLB_STRIDE + PLB_DV] = slot; slot += PLB[b * PLB_STRIDE + PLB_NV]; }
  slot += nvmax + 8;  // chunked helpers read up to 7 slots past a vector
  for (int l = 0; l < nl; l++) { PLL[l * PLL_STRIDE + PLL_POSE] = slot; slot += 9; PLL[l * PLL_STRIDE + PLL_IAACC] = -1; }
  for (int l = 0; l < nl; l++) { PLL[l * PLL_STRIDE + PLL_MROW] = slot; slot += MR_STRIDE; }  // contiguous: pgs_rows_small strides through them
  const int maxc = I[DG_H_MAX_CONTACTS], ncons = I[DG_H_N_CONSTRAINTS];
  // (a fixed constraint keeps two pseudo contact slots behind the real ones: its linear and its angular rows, build_constraint_rows)
  const int cont_off = slot; slot += 1 + (maxc + 2 * ncons) * CL_STRIDE;
  const int ab_stride = any_float ? AB_FLOAT_STRIDE : AB_FIXED_STRIDE;
  // transient region: ABA workspace (+ inertia accumulators for links with a child that is not link+1),
  // contact rows, IK scratch -- never live at the same time
  const int tr_off = slot;
  int tr = 0;
  for (int b = 0; b < nb; b++) {
    const int32_t* B = BI + b * DG_BI_STRIDE; const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS];
    int need = ab_stride + n * AW_STRIDE;
    for (int i = 0; i < n; i++) {
      const int par = LI[(first + i) * DG_LI_STRIDE + DG_LI_PARENT];
      if (par >= 0 && par != first + i - 1 && PLL[par * PLL_STRIDE + PLL_IAACC] < 0) { PLL[par * PLL_STRIDE + PLL_IAACC] = tr_off + need; need += 21; }
    }
    tr = std::max(tr, need);
    if (n > 6) tr = std::max(tr, n * (n + 1) / 2 + 2 * n + 24);  // motor_guess_lds: packed factor + y + scaling, padded
  }
  // contact rows carry a second body's Jacobian / response only if some candidate pair has two moving bodies
  bool two_sided = ncons > 0;
  { const int32_t* PIh = I + I[DG_H_OFF_PAIR_I]; const int32_t* SIh = I + I[DG_H_OFF_SHAPE_I];
    auto moving = [&](int sh) { const int32_t* B = BI + SIh[sh * DG_SI_STRIDE + DG_SI_BODY] * DG_BI_STRIDE; return !((B[DG_BI_FLAGS] & DG_BODY_FIXED) && B[DG_BI_N_LINKS] == 0); };
    for (int p = 0; p < I[DG_H_N_PAIRS]; p++) if (moving(PIh[p * DG_PI_STRIDE + DG_PI_A]) && moving(PIh[p * DG_PI_STRIDE + DG_PI_B])) two_sided = true; }
  int nt = 0; for (int b = 0; b < nb; b++) nt += PLB[b * PLB_STRIDE + PLB_NV];
  const bool dense = nt <= 32 && ncons == 0;  // contact rows indexed by global DoF, swept with the velocity change in registers (fixed-constraint rows: generic sweeps only)
  const int crow_tail = dense ? 2 * nt : (two_sided ? 4 : 2) * nvmax;
  tr = std::max(tr, 3 * (maxc + 2 * ncons) * (crow_tail + 3));
  if (I[DG_H_N_PAIRS] > 0) tr = std::max(tr, (int)SC_STRIDE * I[DG_H_N_SHAPES]);  // narrow-phase shape cache
  for (int op = 0; op < I[DG_H_N_OPS]; op++)
    if (OI[op * DG_OI_STRIDE + DG_OI_CODE] == DG_OP_IK_CONTROL) {
      const int n = BI[OI[op * DG_OI_STRIDE + DG_OI_BODY] * DG_BI_STRIDE + DG_BI_N_LINKS];
      tr = std::max(tr, 9 * n);
    }
  slot += tr + 8;  // + padding for the chunked vector helpers
  const int total = slot;
  { hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, device)); w->cu_count = std::max(prop.multiProcessorCount, 1); }  // (the one query)
  int lanes = 64; const int LDS_MAX = 160 * 1024;
  if (const char* ml = getenv("DG_MAX_LANES")) { const int v = atoi(ml); if (v == 32 || v == 16 || v == 8 || v == 4 || v == 1) lanes = v; }
  // all-dense scenes (every row indexed by global DoF, no register-chain body) can put spare lanes to work in the
  // Gauss-Seidel sweeps, so for them 8 and 4 envs per wavefront are worth having; other scenes stop at 16
  bool has_reg = false;
  for (int b = 0; b < nb; b++) { const int32_t* B = BI + b * DG_BI_STRIDE; if ((B[DG_BI_FLAGS] & DG_BODY_FIXED) && B[DG_BI_N_LINKS] >= 1 && B[DG_BI_N_LINKS] <= 6) has_reg = true; }
  const bool sliceable = dense && nt >= 1 && !has_reg;
  int min_lanes = (sliceable && !getenv("DG_NO_NARROW_MODES")) ? 4 : 16;
  // One env per wavefront: a scene whose rows do not fit the register budget of the 4-envs-per-wavefront sweeps (more than
  // 16 links, or a contact budget above 12) at a batch that gives every SIMD at most one such wavefront -- every row of the
  // scene then sits in registers (pgs_wave_env) and four times as many SIMDs work.  (from_the_readme at 1 024 envs: 5.6 -> 3.8 ms.)
  if (sliceable && lanes > 1 && !getenv("DG_MAX_LANES") && !getenv("DG_NO_NARROW_MODES") && !getenv("DG_NO_WAVE_ENV") && nt <= 32 && nl <= 32 && maxc <= 32 && (nl > 16 || maxc > 12) && total * 4 <= LDS_MAX) {
    if (num_envs <= 4 * w->cu_count) lanes = 1;
  }
  if (lanes == 1 && !(sliceable && nt <= 32 && nl <= 32 && maxc <= 32)) lanes = 4;  // (DG_MAX_LANES=1 on a scene the mode does not hold)
  if (sliceable && lanes == 1) min_lanes = 1;  // (asked for with DG_MAX_LANES=1, or picked above)
  while (lanes >= min_lanes && total * lanes * 4 > LDS_MAX) lanes >>= 1;
  // Latency: a big batch of a sliceable scene that still leaves SIMDs without a wavefront (fewer than FOUR one-wavefront
  // workgroups per CU) is cut into smaller workgroups -- the sweeps get more lanes per env, the rest loses nothing, and a scene
  // whose workspace lets only one or two workgroups of 32 envs share a CU's LDS gets three to eight of 16.  (Round 4: the target
  // was two per CU; at 16 384 envs one wavefront on EVERY SIMD measured drone_pilot 0.195 -> 0.168 ms per step and the 12-joint
  // UR5 + gripper tree 1.16 -> 0.84, marbles unchanged; two per SIMD -- 8 envs per wavefront -- is slower again for drone_pilot:
  // profiles/r4_workspace_modes_16384.txt.)
  if (sliceable && lanes >= min_lanes && num_envs >= 2048 && !getenv("DG_MAX_LANES") && !getenv("DG_NO_NARROW_MODES")) {
    while (lanes > 8 && (num_envs + lanes - 1) / lanes < 4 * w->cu_count) lanes >>= 1;
  }
  if (lanes < min_lanes) {
    // too big for LDS even at 16 envs per wavefront: per-env scratch moves to a global buffer
    // [workgroup][slot][lane] (coalesced, L2-resident); same kernels, Lane<0>
    lanes = 0;
    // all-dense scenes (every row indexed by global DoF, no register-chain bodies) run 16 envs per wavefront
    // instead, so that the other 48 lanes can share each env's solver rows; LDS then only holds the
    // accumulated impulses of those rows
    const int acc_rows = 3 * maxc + 3 * nl;
    if (dense && nt >= 1 && !has_reg && acc_rows * 16 * 4 <= 64 * 1024 && !getenv("DG_NO_SLICED_GLOBAL")) lanes = -16;
    { const int per = envs_per_wave(lanes); const size_t blocks = ((size_t)num_envs + per - 1) / per;  // [workgroup][slot][lane]
      HIP_TRY(hipMalloc((void**)&w->d_gws, sizeof(float) * blocks * (size_t)total * (size_t)per)); }
  }
  w->render_diag = (getenv("DG_RENDER_NO_CULL") ? 1 : 0) | (getenv("DG_RENDER_DIAG") ? atoi(getenv("DG_RENDER_DIAG")) : 0);
  if (const char* e = getenv("DG_RENDER_WPE")) { const int v = atoi(e); w->render_wpe = (v == 3 || v == 1) ? v : 2; }
  w->lanes = lanes; w->lds_bytes = lanes > 0 ? total * lanes * 4 : (lanes < 0 ? (3 * maxc + 3 * nl) * 16 * 4 : 0);
  // ---- device tables (floats converted once)
  std::vector<float> Ff((size_t)n_f); for (int64_t k = 0; k < n_f; k++) Ff[(size_t)k] = (float)F[k];
  // device copy of the int tables, with device-only hints: IK ops on serial chains of <= 6 joints take the
  // register-resident solver
  std::vector<int32_t> Idev(I, I + n_i);
  for (int op = 0; op < I[DG_H_N_OPS]; op++) {
    int32_t* oi = Idev.data() + I[DG_H_OFF_OP_I] + op * DG_OI_STRIDE;
    if (oi[DG_OI_CODE] != DG_OP_IK_CONTROL) continue;
    const int32_t* B = BI + oi[DG_OI_BODY] * DG_BI_STRIDE; const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS];
    bool chain = n >= 1 && n <= 6;
    for (int i = 0; i < n && chain; i++) chain = LI[(first + i) * DG_LI_STRIDE + DG_LI_PARENT] == (i == 0 ? -1 : first + i - 1);
    if (chain) oi[DG_OI_FLAGS] |= DG_IK_DEV_CHAIN;
    // six revolute joints with the end-effector frame on the last link: the fully specialised solve
    bool full = chain && n == 6 && I[I[DG_H_OFF_FRAME_I] + oi[DG_OI_FRAME] * DG_FI_STRIDE + DG_FI_LINK] == first + 5;
    for (int i = 0; i < n && full; i++) full = LI[(first + i) * DG_LI_STRIDE + DG_LI_TYPE] == 0;
    if (full && !getenv("DG_NO_FULL_IK")) oi[DG_OI_FLAGS] |= DG_IK_DEV_FULL;
  }
  HIP_TRY(hipMalloc(&w->d_blob_i, sizeof(int32_t) * (size_t)n_i)); HIP_TRY(hipMemcpy(w->d_blob_i, Idev.data(), sizeof(int32_t) * (size_t)n_i, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&w->d_blob_f, sizeof(float) * (size_t)n_f)); HIP_TRY(hipMemcpy(w->d_blob_f, Ff.data(), sizeof(float) * (size_t)n_f, hipMemcpyHostToDevice));
  // pair descriptors in canonical order (lower shape type first, a box always second), one word per pair
  const size_t pd_off = plan.size();
  { const int32_t* PIh = I + I[DG_H_OFF_PAIR_I]; const int32_t* SIh = I + I[DG_H_OFF_SHAPE_I];
    for (int p = 0; p < I[DG_H_N_PAIRS]; p++) {
      const int sA = PIh[p * DG_PI_STRIDE + DG_PI_A], sB = PIh[p * DG_PI_STRIDE + DG_PI_B];
      const int tA = SIh[sA * DG_SI_STRIDE + DG_SI_TYPE], tB = SIh[sB * DG_SI_STRIDE + DG_SI_TYPE];
      const bool swap = tA == DG_SHAPE_BOX || (tB != DG_SHAPE_BOX && tA > tB);
      const int sa = swap ? sB : sA, sb = swap ? sA : sB, ta = swap ? tB : tA, tb = swap ? tA : tB;
      plan.push_back(sa | (sb << 12) | (ta << 24) | (tb << 26) | ((swap ? 1 : 0) << 28));
    } }
  // group descriptors (broad phase), device-only: centre and reach of the group's static shape when that shape is frozen in
  // the world -- [x y z reach], reach = bound of the moving body + margin + extent of the shape; reach < 0: the narrow
  // phase works the group's bounds out from the tables (a moving partner)
  const size_t gd_off = plan.size();
  { const int32_t* GIh = I + I[DG_H_OFF_GROUP_I]; const int32_t* SIh = I + I[DG_H_OFF_SHAPE_I]; const double* SFh = F + I[DG_H_OFF_SHAPE_F]; const double* BFh = F + I[DG_H_OFF_BODY_F];
    for (int g = 0; g < I[DG_H_N_GROUPS]; g++) {
      const int32_t* gi = GIh + g * DG_GI_STRIDE; const int ss = gi[DG_GI_STATIC_SHAPE]; float d[4] = {0.f, 0.f, 0.f, -1.f};
      if (ss >= 0 && (SIh[ss * DG_SI_STRIDE + DG_SI_FLAGS] & DG_SHAPE_WORLD)) {
        const double* sf = SFh + ss * DG_SF_STRIDE; const int st = SIh[ss * DG_SI_STRIDE + DG_SI_TYPE];
        const float p0 = (float)sf[DG_SF_PARAMS], p1 = (float)sf[DG_SF_PARAMS + 1], p2 = (float)sf[DG_SF_PARAMS + 2];
        const float ext = st == DG_SHAPE_SPHERE ? p0 : st == DG_SHAPE_BOX ? sqrtf(p0 * p0 + p1 * p1 + p2 * p2) : p0 + p1;
        d[0] = (float)sf[DG_SF_POS]; d[1] = (float)sf[DG_SF_POS + 1]; d[2] = (float)sf[DG_SF_POS + 2];
        d[3] = (float)BFh[gi[DG_GI_BODY_A] * DG_BF_STRIDE + DG_BF_BOUND] + (float)F[DG_HF_CONTACT_MARGIN] + ext;
      }
      for (int k = 0; k < 4; k++) { int32_t bits; memcpy(&bits, &d[k], 4); plan.push_back(bits); }
    } }
  // shape frame descriptors, device-only, for narrow-phase lanes that each test a different shape: [LDS slot of the pose
  // of the shape's link (rotation columns, position at + 6) or of its body's base rotation | state offset of the base
  // position (base shapes) or -1 | first hull point | hull points]; slot -1: frozen in the world
  const size_t sd_off = plan.size();
  { const int32_t* SIh = I + I[DG_H_OFF_SHAPE_I];
    for (int s = 0; s < I[DG_H_N_SHAPES]; s++) {
      const int32_t* si = SIh + s * DG_SI_STRIDE; const int b = si[DG_SI_BODY], gl = si[DG_SI_LINK];
      const bool world = (si[DG_SI_FLAGS] & DG_SHAPE_WORLD) != 0;
      const int32_t rslot = world ? -1 : gl >= 0 ? plan[(size_t)nb * PLB_STRIDE + (size_t)gl * PLL_STRIDE + PLL_POSE] : plan[(size_t)b * PLB_STRIDE + PLB_R0];
      plan.push_back(rslot);
      plan.push_back((world || gl >= 0) ? -1 : BI[b * DG_BI_STRIDE + DG_BI_STATE_OFF]);
      plan.push_back(si[DG_SI_POINT_OFF]); plan.push_back(si[DG_SI_N_POINTS]);
    } }
  // ancestor masks per link (body-local bits; bodies with more than 32 links get zeros and are never sliced)
  const size_t am_off = plan.size();
  for (int b = 0; b < nb; b++) {
    const int32_t* B = BI + b * DG_BI_STRIDE; const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS];
    for (int i = 0; i < n; i++) {
      uint32_t m = 0u;
      if (n <= 32) { m = 1u << i; const int par = LI[(first + i) * DG_LI_STRIDE + DG_LI_PARENT]; if (par >= 0) m |= (uint32_t)plan[am_off + (size_t)par]; }
      plan.push_back((int32_t)m);
    }
  }
  PLB = plan.data(); PLL = plan.data() + (size_t)nb * PLB_STRIDE;  // (the appends above may have moved the vector)
  HIP_TRY(hipMalloc(&w->d_plan, sizeof(int32_t) * std::max<size_t>(plan.size(), 1))); HIP_TRY(hipMemcpy(w->d_plan, plan.data(), sizeof(int32_t) * plan.size(), hipMemcpyHostToDevice));
  // global -> constant address space: a no-op on the hardware, a promise of immutability to the compiler
  cip dI = (cip)w->d_blob_i; cfp dF = (cfp)w->d_blob_f;
  DevScene& sc = w->sc; memset(&sc, 0, sizeof sc);
  sc.BI = dI + I[DG_H_OFF_BODY_I]; sc.LI = dI + I[DG_H_OFF_LINK_I]; sc.FI = dI + I[DG_H_OFF_FRAME_I]; sc.SI = dI + I[DG_H_OFF_SHAPE_I];
  sc.PI = dI + I[DG_H_OFF_PAIR_I]; sc.GI = dI + I[DG_H_OFF_GROUP_I]; sc.OI = dI + I[DG_H_OFF_OP_I]; sc.IL = dI + I[DG_H_OFF_ILIST];
  sc.BF = dF + I[DG_H_OFF_BODY_F]; sc.LF = dF + I[DG_H_OFF_LINK_F]; sc.FF = dF + I[DG_H_OFF_FRAME_F]; sc.SF = dF + I[DG_H_OFF_SHAPE_F];
  sc.PF = dF + I[DG_H_OFF_POINT_F]; sc.OF = dF + I[DG_H_OFF_OP_F]; sc.FL = dF + I[DG_H_OFF_FLIST]; sc.HF = dF;
  sc.PLB = (cip)w->d_plan; sc.PLL = sc.PLB + (size_t)nb * PLB_STRIDE; sc.PD = sc.PLB + pd_off; sc.GD = (cfp)(sc.PLB + gd_off); sc.SD = sc.PLB + sd_off; sc.AM = sc.PLB + am_off;
  sc.nba = 0; for (int b = 0; b < nb; b++) if (!(BI[b * DG_BI_STRIDE + DG_BI_FLAGS] & DG_BODY_FROZEN)) sc.nba = b + 1;
  sc.nsha = 0; for (int s = 0; s < I[DG_H_N_SHAPES]; s++) if (I[I[DG_H_OFF_SHAPE_I] + s * DG_SI_STRIDE + DG_SI_TYPE] != DG_SHAPE_BOX) sc.nsha = s + 1;
  sc.no_minv_slices = getenv("DG_NO_MINV_SLICES") ? 1 : 0; sc.no_chain_rows = getenv("DG_NO_CHAIN_ROWS") ? 1 : 0; sc.no_sliced_reset = getenv("DG_NO_SLICED_RESET") ? 1 : 0;
  sc.nb = nb; sc.nl = nl; sc.nfr = I[DG_H_N_FRAMES]; sc.nsh = I[DG_H_N_SHAPES]; sc.npairs = I[DG_H_N_PAIRS]; sc.ngroups = I[DG_H_N_GROUPS]; sc.nops = I[DG_H_N_OPS];
  sc.act_dim = I[DG_H_ACT_DIM]; sc.obs_dim = I[DG_H_OBS_DIM]; sc.rew_dim = I[DG_H_REW_DIM]; sc.term_dim = I[DG_H_TERM_DIM];
  sc.substeps = I[DG_H_SUBSTEPS]; sc.iters = I[DG_H_SOLVER_ITERS]; sc.hot_start = I[DG_H_HOT_START]; sc.ik_iters = I[DG_H_IK_ITERS];
  sc.state_dim = I[DG_H_STATE_DIM]; sc.addon_off = I[DG_H_ADDON_STATE_OFF]; sc.max_contacts = maxc; sc.warm_off = I[DG_H_WARM_OFF]; sc.ncons = ncons; sc.KI = dI + I[DG_H_OFF_CONS_I]; sc.KF = dF + I[DG_H_OFF_CONS_F]; sc.debug_keep_ext = getenv("DG_DEBUG_KEEP_EXT") ? 1 : 0; sc.term_mode = I[DG_H_TERM_MODE]; sc.n_term_groups = I[DG_H_N_TERM_GROUPS];
  sc.tr_off = tr_off; sc.tr_slots = tr; sc.cont_off = cont_off; sc.nv_max = nvmax; sc.total_slots = total; sc.ab_stride = ab_stride; sc.crow_tail = crow_tail; sc.nt = nt; sc.dense = dense ? 1 : 0; sc.dv_base = nb > 0 ? PLB[PLB_DV] : 0;
  sc.num_envs = num_envs; sc.stride = env_stride; sc.seed = seed; sc.env_base = env_index_base;
  // bodies whose solver rows are held in registers by the step kernel
  sc.reg_body[0] = sc.reg_body[1] = -1;
  for (int b = 0, k = 0; b < nb && k < 2; b++) {
    const int32_t* B = BI + b * DG_BI_STRIDE;
    if ((B[DG_BI_FLAGS] & DG_BODY_FIXED) && B[DG_BI_N_LINKS] >= 1 && B[DG_BI_N_LINKS] <= 6) sc.reg_body[k++] = b;
  }
  // helper wave: the LAST fixed-base chain body (so that wave 0 keeps the first arm), provided the scene has other
  // work to overlap with and every inverse-kinematics op on that body has the register-resident form
  sc.helper_body = -1;
  if (lanes == 64 && ncons == 0 && !getenv("DG_NO_HELPER_WAVE")) {
    int n_dyn = 0; for (int b = 0; b < nb; b++) { const int32_t* B = BI + b * DG_BI_STRIDE; if (!((B[DG_BI_FLAGS] & DG_BODY_FIXED) && B[DG_BI_N_LINKS] == 0)) n_dyn++; }
    for (int b = nb - 1; b >= 0 && n_dyn >= 2; b--) {
      if (!PLB[b * PLB_STRIDE + PLB_CHAIN]) continue;
      bool ok = true;
      for (int op = 0; op < I[DG_H_N_OPS]; op++) {
        const int32_t* oi = Idev.data() + I[DG_H_OFF_OP_I] + op * DG_OI_STRIDE;
        if (oi[DG_OI_BODY] == b && oi[DG_OI_CODE] == DG_OP_IK_CONTROL && !(oi[DG_OI_FLAGS] & DG_IK_DEV_CHAIN)) ok = false;
      }
      if (ok) { sc.helper_body = b; break; }
    }
  }
  w->par = sc.helper_body >= 0; w->no_par_reset = getenv("DG_NO_PAR_RESET") != nullptr;
  // polytope workspace of the hull-hull narrow phase: only worlds that collide two hulls
  { bool hull_pairs = false; const int32_t* PIh = I + I[DG_H_OFF_PAIR_I]; const int32_t* SIh = I + I[DG_H_OFF_SHAPE_I];
    for (int p = 0; p < I[DG_H_N_PAIRS] && !hull_pairs; p++)
      hull_pairs = SIh[PIh[p * DG_PI_STRIDE + DG_PI_A] * DG_SI_STRIDE + DG_SI_TYPE] == DG_SHAPE_POINTS && SIh[PIh[p * DG_PI_STRIDE + DG_PI_B] * DG_SI_STRIDE + DG_SI_TYPE] == DG_SHAPE_POINTS;
    if (hull_pairs && F[DG_HF_HULL_CONTACTS] > 0) {
      const int per = envs_per_wave(lanes); const size_t waves = (size_t)((num_envs + per - 1) / per) * (w->par ? 4 : 1);
      HIP_TRY(hipMalloc((void**)&w->d_hull_ws, sizeof(float) * waves * (size_t)HH_WS_SLOTS * 64));
    }
    sc.hull_ws = w->d_hull_ws; }

  // third wavefront for the narrow phase: it uses the transient region as its shape cache while the other two run
  // dynamics, so every moving body must have the register-resident (transient-free) dynamics
  sc.coll_wave = 0;
  if (w->par && I[DG_H_N_PAIRS] > 0 && !getenv("DG_NO_COLLIDE_WAVE")) {
    bool ok = true;
    for (int b = 0; b < nb; b++) { const int32_t* B = BI + b * DG_BI_STRIDE; const bool stat = (B[DG_BI_FLAGS] & DG_BODY_FIXED) && B[DG_BI_N_LINKS] == 0; if (!stat && !PLB[b * PLB_STRIDE + PLB_CHAIN]) ok = false; }
    sc.coll_wave = ok ? 1 : 0;
  }
  // sweeps split between the main and the helper wave: exactly two register-chain bodies, the second is the helper's,
  // and no other body carries joints (their rows would have to run on the main wave between the exchanges)
  sc.split_pgs = 0;
  if (w->par && !getenv("DG_NO_SPLIT_SWEEPS")) {
    int jointed = 0; for (int b = 0; b < nb; b++) if (BI[b * DG_BI_STRIDE + DG_BI_N_LINKS] > 0) jointed++;
    if (jointed == 2 && sc.reg_body[0] >= 0 && sc.reg_body[1] == sc.helper_body && sc.reg_body[0] != sc.helper_body) {
      sc.split_pgs = 1;
      // contact rows too when the dense DoF vector holds nothing but the two arms (no free body a contact could involve)
      if (dense && nt == PLB[sc.reg_body[0] * PLB_STRIDE + PLB_NV] + PLB[sc.reg_body[1] * PLB_STRIDE + PLB_NV] && !getenv("DG_NO_SPLIT_CONTACTS")) sc.split_pgs = 2;
    }
  }
  if (!w->par && getenv("DG_NO_REG_ROWS")) sc.split_pgs = -1;  // ablation: the sliced sweeps keep their rows in LDS
  // a fourth wavefront for the second half of the pair table, if its contact list still fits LDS
  sc.coll_split = 0; sc.cont2_off = 0;
  if (sc.coll_wave && lanes == 64 && I[DG_H_N_PAIRS] >= 8 && !getenv("DG_NO_COLLIDE_SPLIT")) {
    const int extra = 1 + maxc * CL_STRIDE;
    if ((sc.total_slots + extra) * 64 * 4 <= 160 * 1024) {
      sc.cont2_off = sc.total_slots; sc.total_slots += extra; sc.coll_split = 1;
      w->lds_bytes = sc.total_slots * 64 * 4;
    }
  }
  // The update ops of such a scene (inverse kinematics above all) only write motor targets unless one of them is a
  // torque / force op; then the first substep's dynamics do not depend on them and can run alongside.
  sc.early_dyn = 0;
  if (sc.coll_wave && !getenv("DG_NO_EARLY_DYNAMICS")) {
    bool ok = true, long_update = false;  // worth it only when the update phase is long: an inverse-kinematics solve
    for (int op = 0; op < I[DG_H_N_OPS]; op++) {
      const int32_t* oi = OI + op * DG_OI_STRIDE; const int code = oi[DG_OI_CODE];
      if (code == DG_OP_IK_CONTROL) long_update = true;
      if (code == DG_OP_EXTERNAL_FORCE || code == DG_OP_PROPELLOR || code == DG_OP_ADMITTANCE || (code == DG_OP_JOINT_CONTROL && oi[DG_OI_FLAGS] == DG_JC_TORQUE)) ok = false;
    }
    sc.early_dyn = (ok && long_update) ? 1 : 0;
  }
  sc.h = (float)F[DG_HF_DT]; sc.hm = (float)(F[DG_HF_DT] * F[DG_HF_MOTOR_IMPULSE_SCALE]); sc.gx = (float)F[DG_HF_GRAV_X]; sc.gy = (float)F[DG_HF_GRAV_Y]; sc.gz = (float)F[DG_HF_GRAV_Z];
  // ---- default velocity motors on every joint
  memset(&w->mt, 0, sizeof w->mt);
  for (int l = 0; l < nl; l++) { w->mt.v[3 * l] = 0.f; w->mt.v[3 * l + 1] = 1.f; w->mt.v[3 * l + 2] = -(float)F[DG_HF_DEFAULT_MOTOR_IMPULSE]; }
  for (int op = 0; op < I[DG_H_N_OPS]; op++) {  // admittance_controller.py:34: its joints' velocity motors are switched off at construction
    const int32_t* oi = OI + op * DG_OI_STRIDE;
    if (oi[DG_OI_CODE] == DG_OP_ADMITTANCE) for (int k = 0; k < oi[DG_OI_N]; k++) w->mt.v[3 * (I[I[DG_H_OFF_ILIST] + oi[DG_OI_ILIST] + k]) + 2] = 0.f;
  }
  // ---- load-time state vector
  std::vector<float> init((size_t)sc.state_dim, 0.f);
  const double* BF = F + I[DG_H_OFF_BODY_F];
  for (int b = 0; b < nb; b++) {
    const int so = BI[b * DG_BI_STRIDE + DG_BI_STATE_OFF]; if (so < 0) continue;  // frozen: no state
    for (int k = 0; k < 3; k++) init[so + DG_BS_POS + k] = (float)BF[b * DG_BF_STRIDE + DG_BF_INIT_POS + k];
    for (int k = 0; k < 4; k++) init[so + DG_BS_QUAT + k] = (float)BF[b * DG_BF_STRIDE + DG_BF_INIT_QUAT + k];
  }
  for (int op = 0; op < I[DG_H_N_OPS]; op++) {  // dynamics_randomizer state before its first draw: URDF masses, default damping
    const int32_t* oi = OI + op * DG_OI_STRIDE;
    if (oi[DG_OI_CODE] == DG_OP_RANDOMIZE_COLOR) {  // visual_randomizer: the configured colour, flat, until the first draw
      float* tx = init.data() + sc.addon_off + oi[DG_OI_STATE_OFF];
      for (int k = 0; k < 3; k++) tx[DG_TX_A + k] = tx[DG_TX_B + k] = (float)F[I[DG_H_OFF_BODY_F] + oi[DG_OI_BODY] * DG_BF_STRIDE + DG_BF_COLOR + k];
      tx[DG_TX_FREQ] = 1.f; tx[DG_TX_KIND] = (float)DG_TEX_FLAT;
      continue;
    }
    if (oi[DG_OI_CODE] != DG_OP_RANDOMIZE_DYNAMICS) continue;
    const int so = sc.addon_off + oi[DG_OI_STATE_OFF];
    for (int k = 0; k < oi[DG_OI_N]; k++) init[so + k] = 1.f;
    init[so + oi[DG_OI_N]] = (float)F[DG_HF_ANG_DAMPING];
  }
  HIP_TRY(hipMalloc((void**)&w->d_init, sizeof(float) * init.size())); HIP_TRY(hipMemcpy(w->d_init, init.data(), sizeof(float) * init.size(), hipMemcpyHostToDevice));
  // cameras: per-env shape/camera pose table written by pose_kernel, read by render_kernel
  w->ncam = I[DG_H_N_CAMERAS]; w->d_CI = dI + I[DG_H_OFF_CAMERA_I]; w->d_CF = dF + I[DG_H_OFF_CAMERA_F]; w->d_PLN = dF + I[DG_H_OFF_PLANE_F];
  if (w->ncam > 0) HIP_TRY(hipMalloc((void**)&w->d_render_table, sizeof(float) * (size_t)num_envs * (size_t)(sc.nsh * RS_STRIDE + w->ncam * RC_STRIDE)));
  // allow > 64 KiB of dynamic LDS for this mode's kernels
  HIP_TRY(launch_table(lanes).prepare(w->lds_bytes));
  *out = holder.release();
  return DG_OK;
}

void dg_world_destroy(dg_world* w) { delete w; }  // ~dg_world frees the device allocations on the world's device

const char* dg_world_kernel_name(const dg_world* w) {
  if (!w) return "";
  static thread_local char buf[96];
  if (w->par) snprintf(buf, sizeof buf, "step_kernel_par (4 wavefronts per 64 envs)");
  else snprintf(buf, sizeof buf, "step_kernel<%d>", w->lanes);
  return buf;
}

int32_t dg_world_dims(const dg_world* w, int32_t dims[8]) {
  if (!w || !dims) return fail(DG_ERR_ARG, "null argument");
  dims[0] = w->sc.state_dim; dims[1] = w->sc.act_dim; dims[2] = w->sc.obs_dim; dims[3] = w->sc.rew_dim; dims[4] = w->sc.term_dim;
  dims[5] = w->sc.nl; dims[6] = w->lds_bytes; dims[7] = w->lanes;
  return DG_OK;
}
int32_t dg_world_get_motor_cfg(const dg_world* w, double* cfg) {
  if (!w || !cfg) return fail(DG_ERR_ARG, "null argument");
  for (int k = 0; k < 3 * w->sc.nl; k++) cfg[k] = w->mt.v[k];
  return DG_OK;
}
int32_t dg_world_set_motor_cfg(dg_world* w, const double* cfg) {
  if (!w || !cfg) return fail(DG_ERR_ARG, "null argument");
  for (int k = 0; k < 3 * w->sc.nl; k++) w->mt.v[k] = (float)cfg[k];
  return DG_OK;
}
int32_t dg_world_set_diag_buffer(dg_world* w, int32_t* diag) { if (!w) return fail(DG_ERR_ARG, "null world"); w->diag = diag; return DG_OK; }

int32_t dg_world_init_state(dg_world* w, float* state, void* stream) {
  if (!w || !state) return fail(DG_ERR_ARG, "null argument");
  DG_ON_DEVICE(w->device);
  hipLaunchKernelGGL(init_state_kernel, dim3((w->stride + 255) / 256), dim3(256), 0, (hipStream_t)stream, w->d_init, state, w->sc.state_dim, w->stride);
  HIP_TRY(hipGetLastError());
  return DG_OK;
}

static dim3 grid_of(const dg_world* w) { const int per = envs_per_wave(w->lanes); return dim3((w->num_envs + per - 1) / per); }

int32_t dg_world_reset(dg_world* w, float* state, const uint8_t* mask, float* obs, void* stream) {
  if (!w || !state) return fail(DG_ERR_ARG, "null argument");
  DG_ON_DEVICE(w->device);
  // four-wavefront scenes with the usual single hot-start step: the reset ops and that step run in the step kernel itself
  // (reset mode), the envs the mask does not name computing in their scratch workspace with stores off
  if (w->par && w->sc.hot_start == 1 && !w->no_par_reset)
    launch_table(w->lanes).step_par(grid_of(w), w->lds_bytes, (hipStream_t)stream, false, w->sc, w->mt, state, nullptr, 0ull, obs, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, mask, 1);
  else
    launch_table(w->lanes).reset(grid_of(w), w->lds_bytes, (hipStream_t)stream, w->sc, w->mt, state, mask, obs, w->d_gws);
  HIP_TRY(hipGetLastError());
  return DG_OK;
}

int32_t dg_world_step(dg_world* w, float* state, const float* actions, uint64_t update_mask, float* obs, float* rew, uint8_t* term,
                      float* rew_sum, uint8_t* term_flag, void* stream) {
  if (!w || !state) return fail(DG_ERR_ARG, "null argument");
  if (w->sc.act_dim > 0 && update_mask != 0 && !actions) return fail(DG_ERR_ARG, "update_mask selects controller addons but actions is NULL");
  DG_ON_DEVICE(w->device);
  if (actions) {
    // motor gains / force limits are uniform over envs: the controller ops selected by the mask set them here
    // (p.setJointMotorControlArray's positionGains / velocityGains / forces; joint_controller.py:53-58, ik_controller.py:71-80)
    const int32_t* I = w->I.data(); const double* F = w->F.data();
    const int32_t* OI = I + I[DG_H_OFF_OP_I]; const int32_t* IL = I + I[DG_H_OFF_ILIST]; const double* OF = F + I[DG_H_OFF_OP_F]; const double* LF = F + I[DG_H_OFF_LINK_F];
    for (int op = 0; op < w->sc.nops; op++) {
      const int32_t* oi = OI + op * DG_OI_STRIDE; const double* of = OF + op * DG_OF_STRIDE; const int code = oi[DG_OI_CODE];
      if (code != DG_OP_JOINT_CONTROL && code != DG_OP_IK_CONTROL) continue;
      if (!((update_mask >> oi[DG_OI_SLOT]) & 1ull)) continue;
      if (code == DG_OP_JOINT_CONTROL && oi[DG_OI_FLAGS] == DG_JC_TORQUE) continue;
      const bool vel = code == DG_OP_JOINT_CONTROL && oi[DG_OI_FLAGS] == DG_JC_VELOCITY;
      for (int k = 0; k < oi[DG_OI_N]; k++) {
        const int gl = IL[oi[DG_OI_ILIST] + k];
        w->mt.v[3 * gl] = vel ? 0.f : (float)of[0]; w->mt.v[3 * gl + 1] = (float)of[1]; w->mt.v[3 * gl + 2] = (float)LF[gl * DG_LF_STRIDE + DG_LF_MAX_FORCE];
      }
    }
  }
  {
    const LaunchTable& lt = launch_table(w->lanes); const bool prof = w->profile_cycles != nullptr;
    if (prof && !lt.has_prof) return fail(DG_ERR_UNSUPPORTED, "in-kernel stamps are not built for this workspace mode");
    if (w->par) lt.step_par(grid_of(w), w->lds_bytes, (hipStream_t)stream, prof, w->sc, w->mt, state, actions, update_mask, obs, rew, term, rew_sum, term_flag, w->diag, w->profile_cycles, nullptr, 0);
    else lt.step(grid_of(w), w->lds_bytes, (hipStream_t)stream, prof, w->sc, w->mt, state, actions, update_mask, obs, rew, term, rew_sum, term_flag, w->diag, w->profile_cycles, w->d_gws);
    HIP_TRY(hipGetLastError());
  }
  return DG_OK;
}

int32_t dg_world_render(dg_world* w, const float* state, int32_t camera, float* rgb, float* depth, int32_t* seg, void* stream) {
  if (!w || !state) return fail(DG_ERR_ARG, "null argument");
  if (camera < 0 || camera >= w->ncam) return fail(DG_ERR_ARG, "camera %d out of range (scene has %d)", camera, w->ncam);
  DG_ON_DEVICE(w->device);
  launch_table(w->lanes).pose(grid_of(w), w->lds_bytes, (hipStream_t)stream, w->sc, w->mt, const_cast<float*>(state), w->ncam, w->d_CI, w->d_CF, w->d_render_table, w->d_gws);
  HIP_TRY(hipGetLastError());
  const int32_t* I = w->I.data(); const int32_t* ci = I + I[DG_H_OFF_CAMERA_I] + camera * DG_CI_STRIDE;
  // Rows per workgroup: a multiple of 8 (the tile height of render_kernel's wavefronts; 200-wide images: whole cache lines).
  // Every workgroup first builds its list of shapes, faces and vertices (phase A, a few microseconds of dependent loads),
  // so a workgroup should render as many rows as the machine's occupancy allows: about four workgroups per CU over the
  // whole launch -- with >= 1024 envs one workgroup renders a whole image.
  const int W = ci[DG_CI_WIDTH], H = ci[DG_CI_HEIGHT];
  if (W > 16 * 512) return fail(DG_ERR_UNSUPPORTED, "render: pictures wider than %d pixels are not supported (%d)", 16 * 512, W);  // DG_RTILE_CAP tiles per strip
  int band_rows;
  { const int want_blocks = 4 * std::max(w->cu_count, 1), bands_per_env = std::max(1, (want_blocks + w->num_envs - 1) / w->num_envs);
    band_rows = std::max(8, ((H + bands_per_env - 1) / bands_per_env + 7) / 8 * 8); }
  if (w->render_diag & 512) band_rows = H;  // (diag 512: one band per picture whatever the batch -- what a batch >= 4 x CUs gets; for tests)
  band_rows = std::min(band_rows, H);
  const int nbands = (H + band_rows - 1) / band_rows;
  const long long blocks = (long long)nbands * w->num_envs;  // the env index is folded into grid.x (grid.y stops at 65535)
  if (blocks > 0x7fffffffLL) return fail(DG_ERR_UNSUPPORTED, "render: %lld workgroups exceed the grid limit", blocks);
  if (w->render_wpe == 1) {
    hipLaunchKernelGGL(render_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w->sc, w->d_CI, w->d_CF, w->d_PLN, camera, w->ncam,
                     (cfp)w->d_render_table, rgb, depth, seg, band_rows, nbands, w->render_diag);
  } else if (w->render_wpe == 3) {
    hipLaunchKernelGGL(render_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w->sc, w->d_CI, w->d_CF, w->d_PLN, camera, w->ncam,
                     (cfp)w->d_render_table, rgb, depth, seg, band_rows, nbands, w->render_diag);
  } else {
    hipLaunchKernelGGL(render_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w->sc, w->d_CI, w->d_CF, w->d_PLN, camera, w->ncam,
                     (cfp)w->d_render_table, rgb, depth, seg, band_rows, nbands, w->render_diag);
  }
  HIP_TRY(hipGetLastError());
  return DG_OK;
}

// diagnostics: candidate counters of the render kernel (DG_RENDER_DIAG & 16); not part of the public header
int32_t dg_debug_render_counters(uint64_t* out16, int32_t reset) {
  unsigned long long h[16];
  HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_render_count), sizeof h));
  for (int k = 0; k < 16; k++) out16[k] = h[k];
  if (reset) { memset(h, 0, sizeof h); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_render_count), h, sizeof h)); }
  return DG_OK;
}

// diagnostics: the hull-against-hull narrow phase (dg_hull.h) on its own, one pair of poses per lane -- tests/test_hull_contacts.py
// compares it with the CPU checker and a brute-force Minkowski difference; host pointers; not part of the public header.
// poses [n][24] = A: rotation (9, row-major), position (3); B: the same.  out [n][12] = witness on A, witness on B, normal from B
// towards A, signed distance, 1 if the pair is nearer than max_dist (else the rest is undefined), GJK iterations the lane needed.
__global__ __launch_bounds__(64) void hull_pair_kernel(cfp pts, int na, int nb, const float* poses, int n, float max_dist, float* out, float* ws) {
  const int i = blockIdx.x * 64 + threadIdx.x; const bool have = i < n; const float* p = poses + 24 * (size_t)(have ? i : n - 1);
  HullPairD h; h.pa = pts; h.na = na; h.pb = pts + 3 * na; h.nb = nb;
#pragma unroll
  for (int k = 0; k < 9; k++) { h.RA.m[k] = p[k]; h.RB.m[k] = p[12 + k]; }
  const V3 ta = v3(p[9], p[10], p[11]), tb = v3(p[21], p[22], p[23]); h.tBA = tb - ta;
  V3 ca = v3(0.f, 0.f, 0.f), cb = ca;  // seed: the difference of the centroids, as the checker's test entry
  for (int k = 0; k < na; k++) ca = ca + v3(h.pa[3 * k], h.pa[3 * k + 1], h.pa[3 * k + 2]);
  for (int k = 0; k < nb; k++) cb = cb + v3(h.pb[3 * k], h.pb[3 * k + 1], h.pb[3 * k + 2]);
  const V3 seed = mul(h.RA, ca * (1.0f / (float)na)) - (mul(h.RB, cb * (1.0f / (float)nb)) + h.tBA);
  h.ew = hull_ws_of(ws); hull_tables(h, na <= 64 && nb <= 64);  // (every lane of the wavefront is here)
  HullHit r; hull_hull(h, seed, max_dist, have, r);
  if (have) { float* o = out + 12 * (size_t)i; o[11] = (float)r.iters; const V3 pa = r.pa + ta, pb = r.pb + ta;
    o[0] = pa.x; o[1] = pa.y; o[2] = pa.z; o[3] = pb.x; o[4] = pb.y; o[5] = pb.z; o[6] = r.n.x; o[7] = r.n.y; o[8] = r.n.z; o[9] = r.dist; o[10] = r.hit ? 1.f : 0.f; }
}
int32_t dg_debug_hull_hull(const float* pts_a, int32_t na, const float* pts_b, int32_t nb, const float* poses, int32_t n, float max_dist, float* out11 /* [n][12] */) {
  if (!pts_a || !pts_b || !poses || !out11 || na < 1 || nb < 1 || na > 256 || nb > 256 || n < 1) return fail(DG_ERR_ARG, "dg_debug_hull_hull: bad argument");
  struct Bufs { float *pts = nullptr, *poses = nullptr, *out = nullptr, *ws = nullptr;
                ~Bufs() { (void)hipFree(pts); (void)hipFree(poses); (void)hipFree(out); (void)hipFree(ws); } } b;  // (freed on every way out)
  float *&d_pts = b.pts, *&d_poses = b.poses, *&d_out = b.out, *&d_ws = b.ws; const size_t blocks = (size_t)((n + 63) / 64);
  HIP_TRY(hipMalloc(&d_ws, sizeof(float) * blocks * (size_t)HH_WS_SLOTS * 64));
  HIP_TRY(hipMalloc(&d_pts, sizeof(float) * 3 * (size_t)(na + nb))); HIP_TRY(hipMalloc(&d_poses, sizeof(float) * 24 * (size_t)n)); HIP_TRY(hipMalloc(&d_out, sizeof(float) * 12 * (size_t)n));
  HIP_TRY(hipMemcpy(d_pts, pts_a, sizeof(float) * 3 * (size_t)na, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(d_pts + 3 * na, pts_b, sizeof(float) * 3 * (size_t)nb, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_poses, poses, sizeof(float) * 24 * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(hull_pair_kernel, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)0, (cfp)d_pts, na, nb, (const float*)d_poses, n, max_dist, d_out, d_ws);
  HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out11, d_out, sizeof(float) * 12 * (size_t)n, hipMemcpyDeviceToHost));
  return DG_OK;
}

int32_t dg_world_set_profile_buffer(dg_world* w, uint64_t* cycles) { if (!w) return fail(DG_ERR_ARG, "null world"); w->profile_cycles = (unsigned long long*)cycles; return DG_OK; }

int32_t dg_world_observe(dg_world* w, const float* state, float* obs, float* rew, uint8_t* term, float* rew_sum, uint8_t* term_flag, void* stream) {
  if (!w || !state) return fail(DG_ERR_ARG, "null argument");
  DG_ON_DEVICE(w->device);
  launch_table(w->lanes).observe(grid_of(w), w->lds_bytes, (hipStream_t)stream, w->sc, w->mt, const_cast<float*>(state), obs, rew, term, rew_sum, term_flag, w->d_gws);
  HIP_TRY(hipGetLastError());
  return DG_OK;
}

int32_t dg_world_frame_state(dg_world* w, const float* state, int32_t body, int32_t frame, int32_t com, float* out, void* stream) {
  if (!w || !state || !out) return fail(DG_ERR_ARG, "null argument");
  if (body < 0 || body >= w->sc.nb) return fail(DG_ERR_ARG, "body %d out of range", body);
  // `frame` is the body-local pybullet joint index; the kernels use the global frame index
  int gf = -1;
  if (frame >= 0) {
    const int32_t* I = w->I.data(); const int32_t* FI = I + I[DG_H_OFF_FRAME_I]; int seen = 0; bool found = false;
    for (int f = 0; f < w->sc.nfr; f++) if (FI[f * DG_FI_STRIDE + DG_FI_BODY] == body) { if (seen == frame) { gf = f; found = true; break; } seen++; }
    if (!found) return fail(DG_ERR_ARG, "body %d has no frame %d", body, frame);
  }
  DG_ON_DEVICE(w->device);
  launch_table(w->lanes).frame(grid_of(w), w->lds_bytes, (hipStream_t)stream, w->sc, w->mt, const_cast<float*>(state), body, gf, com, out, w->d_gws);
  HIP_TRY(hipGetLastError());
  return DG_OK;
}

int32_t dg_world_set_render_diag(dg_world* w, int32_t flags) {
  if (!w) return fail(DG_ERR_ARG, "null argument");
  w->render_diag = flags;
  return DG_OK;
}

int32_t dg_world_apply_wrench(dg_world* w, float* state, int32_t body, int32_t frame, int32_t flags, const float* force, const float* pos,
                              const float* torque, void* stream) {
  if (!w || !state) return fail(DG_ERR_ARG, "null argument");
  if (body < 0 || body >= w->sc.nb) return fail(DG_ERR_ARG, "body %d out of range", body);
  const int32_t* I = w->I.data();
  if (I[I[DG_H_OFF_BODY_I] + body * DG_BI_STRIDE + DG_BI_FLAGS] & DG_BODY_FROZEN) return fail(DG_ERR_ARG, "body %d is part of the frozen static world: it has no state to push on", body);
  if (flags != DG_WRENCH_WORLD_FRAME && flags != DG_WRENCH_LINK_FRAME) return fail(DG_ERR_ARG, "flags must be DG_WRENCH_LINK_FRAME (1) or DG_WRENCH_WORLD_FRAME (2)");
  int gf = -1;  // `frame` is the body-local pybullet joint index; the kernels use the global frame index
  if (frame >= 0) {
    const int32_t* FI = I + I[DG_H_OFF_FRAME_I]; int seen = 0; bool found = false;
    for (int f = 0; f < w->sc.nfr; f++) if (FI[f * DG_FI_STRIDE + DG_FI_BODY] == body) { if (seen == frame) { gf = f; found = true; break; } seen++; }
    if (!found) return fail(DG_ERR_ARG, "body %d has no frame %d", body, frame);
  }
  if (!force && !torque) return DG_OK;
  DG_ON_DEVICE(w->device);
  launch_table(w->lanes).wrench(grid_of(w), w->lds_bytes, (hipStream_t)stream, w->sc, w->mt, state, body, gf, flags == DG_WRENCH_LINK_FRAME ? 1 : 0, force, pos, torque, w->d_gws);
  HIP_TRY(hipGetLastError());
  return DG_OK;
}

}  // extern "C"
