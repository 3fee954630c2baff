// dg_render.h -- batched camera addon: depth / segmentation / flat-shaded rgb by ray casting the
// collision geometry (reference: diy_gym/addons/sensors/camera.py:26-98, p.getCameraImage).
//
// Two launches per camera:
//   pose_kernel   one env per lane (same LDS workspace as the step kernels): world frame + bounding
//                 sphere of every shape and the camera pose -> a small per-env table in HBM;
//   render_kernel one pixel per thread, one env per blockIdx.y: the env's table is read through
//                 wave-uniform (scalar) loads, every shape is culled by its bounding sphere and then
//                 intersected analytically (sphere, box slabs, capsule, convex hull face planes).
// The image writes are the dominant HBM traffic of a camera scene (16 B per pixel for rgb + depth):
// this is the one kernel of the path that is bound by HBM write bandwidth, not by instruction issue.
#pragma once
#include "dg_solver.h"

namespace dg {

enum { RS_R = 0, RS_P = 9, RS_C = 12, RS_BOUND = 15, RS_COLOR = 16 /* this env's texture of the shape (DG_TX_*): colour A, colour B, frequency, kind */, RS_STRIDE = 24, RC_STRIDE = 12 };

template <int LANES>
__global__ __launch_bounds__(64) void pose_kernel(DevScene sc, MotorTable mt, float* state, int ncam, cip CI, cfp CF, float* table, float* gws) {
  extern __shared__ float smem[];
  constexpr int ACTIVE = envs_per_wave(LANES);
  // the 64 / ACTIVE lanes that a narrow mode leaves idle per env share the env's shapes: each of them runs the (cheap, serial)
  // kinematics of the env into the same workspace -- same values, same addresses -- and then takes every (64 / ACTIVE)-th shape
  constexpr int GROUP = 64 / ACTIVE;
  const int lane = threadIdx.x % ACTIVE, sub = threadIdx.x / ACTIVE;
  const int env = blockIdx.x * ACTIVE + lane; if (env >= sc.num_envs) return;
  Lane<LANES> ln(sc, mt, workspace_of<LANES>(sc, smem, gws, lane), state + env, env, false);
  for (int b = 0; b < sc.nba; b++) ln.kinematics(b);
  float* out = table + (size_t)env * (sc.nsh * RS_STRIDE + ncam * RC_STRIDE);
  for (int sh = sub; sh < sc.nsh; sh += GROUP) {
    WShape w; shape_world(ln, sh, w); cip si = sc.SI + sh * DG_SI_STRIDE; float* o = out + sh * RS_STRIDE;
    M3 R = w.R; V3 p = w.p; float bound;
    if (w.type == DG_SHAPE_POINTS) {  // hull planes live in the link frame (or the world for a frozen body)
      if (si[DG_SI_FLAGS] & DG_SHAPE_WORLD) { M3 Id = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}; R = Id; p = v3(0.f, 0.f, 0.f); }
      else ln.link_world(w.body, w.glink, R, p);
      bound = w.prm0 + w.prm1;
    } else if (w.type == DG_SHAPE_SPHERE) bound = w.prm0;
    else if (w.type == DG_SHAPE_BOX) bound = sqrtf(w.prm0 * w.prm0 + w.prm1 * w.prm1 + w.prm2 * w.prm2);
    else bound = w.prm0 + w.prm1;
#pragma unroll
    for (int k = 0; k < 9; k++) o[RS_R + k] = R.m[k];
    o[RS_P] = p.x; o[RS_P + 1] = p.y; o[RS_P + 2] = p.z; o[RS_C] = w.p.x; o[RS_C + 1] = w.p.y; o[RS_C + 2] = w.p.z; o[RS_BOUND] = bound;
    { const int co = ln.bi(w.body)[DG_BI_COLOR_OFF]; cfp sc3 = sc.SF + sh * DG_SF_STRIDE + DG_SF_COLOR;  // per-env texture of a visual_randomizer, else the shape's own colour, flat
      _Pragma("unroll") for (int k = 0; k < DG_TX_STRIDE; k++) o[RS_COLOR + k] = co >= 0 ? ln.S(co + k) : (k < 6 ? sc3[k % 3] : (k == DG_TX_FREQ ? 1.f : (float)DG_TEX_FLAT)); }
  }
  for (int c = sub; c < ncam; c += GROUP) {
    cip ci = CI + c * DG_CI_STRIDE; cfp cf = CF + c * DG_CF_STRIDE;
    M3 Rp = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}; V3 pp = v3(0.f, 0.f, 0.f);
    if (ci[DG_CI_BODY] >= 0) { V3 v, w; Q4 q; ln.frame_state(ci[DG_CI_BODY], ci[DG_CI_FRAME], ci[DG_CI_FRAME] < 0, pp, q, v, w, false); Rp = qmat(q); }
    Q4 qc = {cf[DG_CF_QUAT], cf[DG_CF_QUAT + 1], cf[DG_CF_QUAT + 2], cf[DG_CF_QUAT + 3]};
    M3 Rc = mul(Rp, qmat(qc)); V3 pc = pp + mul(Rp, v3(cf[DG_CF_POS], cf[DG_CF_POS + 1], cf[DG_CF_POS + 2]));
    float* o = out + sc.nsh * RS_STRIDE + c * RC_STRIDE;
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = Rc.m[k];
    o[9] = pc.x; o[10] = pc.y; o[11] = pc.z;
  }
}

// Near plane: a surface the ray ENTERS nearer than tmin (= the camera's near distance; t is eye-space depth, the rays have
// z = -1) neither shows nor hides anything -- what clipping at the near plane does in a rasteriser.  In particular the link a
// camera is mounted on (its eye sits ON a face of that link's hull, at t = +-1 ulp) cannot blank the picture, whichever
// way the last bit falls.
struct RayHit { float t; V3 n; int shape; float tmin; };

DGD void ray_sphere(V3 o, V3 d, V3 c, float r, RayHit& h, int sh) {
  const V3 oc = o - c; const float a = dot(d, d), b = dot(oc, d), cc = dot(oc, oc) - r * r, disc = b * b - a * cc;
  if (disc < 0.f) return;
  const float t = fdiv(-b - sqrtf(disc), a);
  if (t >= h.tmin && t < h.t) { h.t = t; h.n = ((o + d * t) - c) * __frcp_rn(r); h.shape = sh; }
}
DGD void ray_box(V3 o, V3 d, const M3& R, V3 p, float hx, float hy, float hz, RayHit& h, int sh) {
  const V3 ol = tmul(R, o - p), dl = tmul(R, d);
  const float oo[3] = {ol.x, ol.y, ol.z}, dd[3] = {dl.x, dl.y, dl.z}, hh[3] = {hx, hy, hz};
  float tn = -3.0e38f, tf = 3.0e38f, sg = 1.f; int ax = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (fabsf(dd[k]) < 1e-30f) { if (fabsf(oo[k]) > hh[k]) { tn = 3.0e38f; tf = -3.0e38f; } }
    else {
      const float inv = __frcp_rn(dd[k]); float t1 = (-hh[k] - oo[k]) * inv, t2 = (hh[k] - oo[k]) * inv, s = -1.f;
      if (t1 > t2) { const float tt = t1; t1 = t2; t2 = tt; s = 1.f; }
      if (t1 > tn) { tn = t1; ax = k; sg = s; }
      tf = fminf(tf, t2);
    }
  }
  if (tn > tf || tn < h.tmin || tn >= h.t) return;
  h.t = tn; h.n = mul(R, v3(ax == 0 ? sg : 0.f, ax == 1 ? sg : 0.f, ax == 2 ? sg : 0.f)); h.shape = sh;
}
DGD void ray_capsule(V3 o, V3 d, V3 e0, V3 e1, float r, RayHit& h, int sh) {
  const V3 ax = e1 - e0; const float L2 = dot(ax, ax);
  if (L2 > 1e-24f) {
    const V3 oc = o - e0; const float dax = dot(d, ax), oax = dot(oc, ax);
    const float iL2 = __frcp_rn(L2); const float a = dot(d, d) - dax * dax * iL2, b = dot(oc, d) - oax * dax * iL2, c = dot(oc, oc) - oax * oax * iL2 - r * r, disc = b * b - a * c;
    if (a > 1e-24f && disc >= 0.f) {
      const float t = fdiv(-b - sqrtf(disc), a), s = (oax + t * dax) * iL2;
      if (t >= h.tmin && t < h.t && s >= 0.f && s <= 1.f) { h.t = t; h.n = ((o + d * t) - (e0 + ax * s)) * __frcp_rn(r); h.shape = sh; }
    }
  }
  ray_sphere(o, d, e0, r, h, sh); ray_sphere(o, d, e1, r, h, sh);
}
DGD void ray_hull(V3 o, V3 d, const M3& Rl, V3 pl, cfp planes, int np, RayHit& h, int sh) {
  const V3 ol = tmul(Rl, o - pl), dl = tmul(Rl, d); float tn = -3.0e38f, tf = 3.0e38f; V3 nn = v3(0.f, 0.f, 1.f); bool miss = np == 0;
  for (int k = 0; k < np; k++) {
    cfp pp = planes + 4 * k; const V3 n = v3(pp[0], pp[1], pp[2]);
    const float den = dot(n, dl), dist = dot(n, ol) + pp[3];
    if (fabsf(den) < 1e-30f) { if (dist > 0.f) miss = true; }
    else { const float t = -dist * __frcp_rn(den); if (den < 0.f) { if (t > tn) { tn = t; nn = n; } } else tf = fminf(tf, t); }
  }
  if (miss || tn > tf || tn < h.tmin || tn >= h.t) return;
  h.t = tn; h.n = mul(Rl, nn); h.shape = sh;
}

#ifdef DG_DEFINE_RENDER_KERNEL  // defined in exactly one translation unit (dg_api.hip)
// One workgroup renders a BAND of full image rows of one env: the band is one contiguous, cache-line aligned piece of
// every output image (200 x 200: 8 rows = 50 lines of depth, 150 of rgb), so each image line is written whole, by one
// workgroup, through one L2.
//   Phase A (once per band, 256 threads): thread t tests shape t against the band's viewing cone; survivors are
//     compacted IN SHAPE ORDER (ties between coincident surfaces resolve exactly as in a brute-force loop) into an LDS
//     list that holds everything phase B needs -- culling data, pose, parameters -- and the face planes of the
//     surviving hulls are rewritten as (world normal, signed distance of the EYE): a ray then costs 3 FMAs and a
//     reciprocal per face, with one broadcast LDS read.  A hull that contains the eye can never be ENTERED by a ray and
//     is dropped.  Nothing in phase B touches the scene tables in global memory: with a handful of wavefronts per SIMD
//     the dependent scalar loads of the previous version (two round trips to L2 per candidate, one per hull face) were
//     what the kernel waited for -- 3.3 ms against 0.35 ms with the intersections switched off.
//   Phase B: a wavefront takes a 16 x 8 pixel tile of the band at a time (two pixels per lane; the band is 8 rows
//     high, so its 64-byte row pieces are completed into whole lines by the neighbouring tiles of the same workgroup),
//     culls the band's list against the tile's cone (lane l tests entry l, one ballot), rejects every box / hull that
//     has a SEPARATING FACE for the whole tile -- the eye on its outer side and no ray of the cone heading back towards
//     it: lane l tests face l, one ballot -- and intersects what is left.  The separating-face test is what makes an
//     eye sitting next to (or inside the bounding sphere of) big or nearby geometry cheap: the ground under a camera
//     that looks up, the gripper the camera is mounted on.
// Rays are affine in the pixel coordinates (dir = A + col B + row C): no division anywhere in the pixel path.
__device__ unsigned long long g_render_count[16];  // diagnostic build-in counters (DG_RENDER_DIAG & 16), read by dg_debug_render_counters
#ifdef DG_RENDER_COUNTERS  /* make CXXFLAGS+=-DDG_RENDER_COUNTERS: per-stage candidate counts for tools/gpu_cam_bench.py */
#define DG_RCOUNT(k) do { if ((diag & 16) && lane == 0) atomicAdd(&g_render_count[k], 1ull); } while (0)
#define DG_RTIME(k, t0) do { if ((diag & 16) && lane == 0) atomicAdd(&g_render_count[k], (unsigned long long)(__builtin_amdgcn_s_memtime() - (t0))); } while (0)
#define DG_RNOW() __builtin_amdgcn_s_memtime()
#else
#define DG_RCOUNT(k) do { } while (0)
#define DG_RTIME(k, t0) do { } while (0)
#define DG_RNOW() 0ull
#endif
#define DG_RL_CAP 96     /* list entries per band */
#define DG_RP_CAP 1024   /* hull faces per band   */
#define DG_RT_CAP 1024   /* hull points / box corners per band */
#define DG_RS_CAP 64     /* strips of eight rows per chunk */
#define DG_RTILE_CAP 512  /* 16 x 8 tiles per chunk of strips */
enum { RL_V = 0 /* centre - eye */, RL_BOUND = 3, RL_R = 4, RL_P = 13, RL_PRM = 16, RL_TEX = 20 /* DG_TX_*: the shape's colours / texture in this env */, RL_STRIDE = 28 };  // floats per entry; ints alongside
enum { RLI_TYPE = 0 /* -1: dropped */, RLI_SHAPE, RLI_PLANE_OFF, RLI_NP, RLI_PT_OFF, RLI_NPT, RLI_NOUT /* faces with the eye on their outer side: stored first */, RLI_SEG /* the shape's segmentation value */, RLI_STRIDE };
// Two rays (the lane's two pixels) against one convex hull whose faces are (world normal n, s = signed distance of the
// eye), the `nout` faces with the eye on their OUTER side (s > 0) first.  The ray eye + t d crosses a face at
// t = -s / (n . d).  A ray can only ENTER through an outer-side face it approaches (n . d < 0); an outer-side face it
// does not approach is never crossed inwards -- a miss.  So the first pass (outer-side faces) settles the entry point
// and most misses, and the second pass (the other faces: exit = the earliest crossing with n . d > 0) is skipped by
// the whole wavefront when no ray is still in play.  Same arithmetic and the same entry face as a single loop over
// the faces in their original order (the partition is stable).
// The first pass ends as soon as every ray of the wavefront has missed.  (Seeding `miss` with the per-ray bounding-sphere test
// was tried: the face planes of a thinned hull reach a hair beyond the sphere around its points, and single silhouette pixels
// then differ from the brute-force picture.)
DGD void ray_hull_world2(const V3 (&d)[2], const float (*pl)[4], int nout, int np, RayHit (&h)[2], int sh) {
  // Crossing depths are kept as FRACTIONS while the faces are scanned -- the latest entry as sn / qn (s = the eye's distance from
  // the face > 0, q = -(n . d) > 0), the earliest exit as af / df -- and compared by cross-multiplication: the loop then holds no
  // reciprocal (a quarter-rate instruction, ~4 issue slots of the ~14 a face cost per ray); the two depths a ray needs are
  // divided out once, after the loops, and are the bits the division per face gave (-s rcp(den) = s rcp(-den)).
  float sn[2] = {-1.f, -1.f}, qn[2] = {1.f, 1.f}, af[2] = {3.0e38f, 3.0e38f}, df[2] = {1.f, 1.f}; int kn[2] = {0, 0}; bool miss[2] = {np == 0, np == 0};
  // (four faces per round, their LDS reads issued together: one face at a time the loop waits a whole LDS round trip per face)
  for (int k0 = 0; k0 < nout; k0 += 4) {
    float f[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int k = min(k0 + j, nout - 1); const float4 v = *reinterpret_cast<const float4*>(pl[k]); f[j][0] = v.x; f[j][1] = v.y; f[j][2] = v.z; f[j][3] = v.w; }
    // (a round's spare slots repeat the last face: the same crossing again changes neither the latest entry -- the comparison is
    // strict -- nor the verdict)
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const float q = -(f[j][0] * d[u].x + f[j][1] * d[u].y + f[j][2] * d[u].z);
        const bool front = q >= 1e-30f;  // approached from outside, and not (numerically) parallel
        miss[u] = miss[u] || !front;
        const bool later = front && f[j][3] * qn[u] > sn[u] * q;  // s / q > sn / qn, both denominators positive
        sn[u] = later ? f[j][3] : sn[u]; qn[u] = later ? q : qn[u]; kn[u] = later ? k0 + j : kn[u];  // (a repeated face never is `later`: kn < nout)
      }
    }
    if (!__any(!miss[0] || !miss[1])) return;
  }
  bool alive[2]; float tn[2];
#pragma unroll
  for (int u = 0; u < 2; u++) { tn[u] = -sn[u] * __frcp_rn(-qn[u]); alive[u] = !miss[u] && tn[u] >= h[u].tmin && tn[u] < h[u].t; }
  if (!__any(alive[0] || alive[1])) return;
  for (int k0 = nout; k0 < np; k0 += 4) {
    float f[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int k = min(k0 + j, np - 1); const float4 v = *reinterpret_cast<const float4*>(pl[k]); f[j][0] = v.x; f[j][1] = v.y; f[j][2] = v.z; f[j][3] = v.w; }
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const float den = f[j][0] * d[u].x + f[j][1] * d[u].y + f[j][2] * d[u].z, a = -f[j][3];
        const bool earlier = den >= 1e-30f && a * df[u] < af[u] * den;  // a / den < af / df
        af[u] = earlier ? a : af[u]; df[u] = earlier ? den : df[u];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const float tf = af[u] * __frcp_rn(df[u]);
    if (alive[u] && !(tn[u] > tf)) { h[u].t = tn[u]; h[u].n = v3(pl[kn[u]][0], pl[kn[u]][1], pl[kn[u]][2]); h[u].shape = sh; }
  }
}
// colour of a pixel whose ray (direction d) hit at h: Lambert factor x the shape's colour / procedural texture; e = rotation (9),
// position (3) and texture (DG_TX_*) of the shape that was hit
DGD void shade_pixel(V3 pc, const RayHit& h, V3 d, const float* e, float (&col)[3]) {
  const float* tx = e + 12;
  const float nl = h.n.x * 0.30151134457776363f + h.n.y * 0.30151134457776363f + h.n.z * 0.9045340337332909f, shd = 0.4f + 0.6f * fmaxf(nl, 0.f);
  float t = 0.f; const int kind = (int)tx[DG_TX_KIND];
  if (kind != DG_TEX_FLAT) {  // procedural texture in the shape's reference frame (DG_TX_* in diygym_scene.h)
    M3 R; _Pragma("unroll") for (int q = 0; q < 9; q++) R.m[q] = e[q];
    const V3 pl = tmul(R, (pc + d * h.t) - v3(e[9], e[10], e[11])); const float fr = tx[DG_TX_FREQ];
    const int ux = (int)floorf(pl.x * fr), uy = (int)floorf(pl.y * fr), uz = (int)floorf(pl.z * fr);
    if (kind == DG_TEX_CHECKER) t = ((ux + uy + uz) & 1) ? 1.f : 0.f;
    else if (kind == DG_TEX_STRIPES) t = (ux & 1) ? 1.f : 0.f;
    else { uint32_t hh; DG_TEX_HASH(ux, uy, uz, hh); t = (float)hh * (1.0f / 16777216.0f); }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) col[k] = (tx[DG_TX_A + k] + (tx[DG_TX_B + k] - tx[DG_TX_A + k]) * t) * shd;
}

// More survivors than the band's list holds (a maze seen from above: 130 shapes in view): every pixel of the band against
// every shape, straight from the tables -- as slow as it is general.  A function of its own, not inlined: inside the kernel its
// registers counted against the tile loop (19 spilled registers there, whose scratch reloads cost a third of the kernel's time).
__device__ __noinline__ void render_band_slow(cip SI, cfp SF, int nsh, cfp tb, cfp PLN, V3 pc, V3 rayA, V3 rayB, V3 rayC, int W, int H, int r0, int npx, float zn, float zf,
                                              int env, int diag, int tid, float* rgb, float* depth, int32_t* seg) {
  const float inv_w = 1.0f / (float)W;
  for (int p = tid; p < npx; p += 256) {
    int row = (int)(((float)p + 0.5f) * inv_w), col = p - row * W; if (col < 0) { row--; col += W; } else if (col >= W) { row++; col -= W; } row += r0;
    const V3 d = rayA + rayB * (col + 0.5f) + rayC * (row + 0.5f);
    RayHit h; h.t = zf; h.shape = -1; h.n = v3(0.f, 0.f, 1.f); h.tmin = (diag & 32) ? 1e-30f : zn;
    if (!(diag & 2)) for (int k = 0; k < nsh; k++) {
      cfp s = tb + k * RS_STRIDE; cip si = SI + k * DG_SI_STRIDE; cfp sf = SF + k * DG_SF_STRIDE; const int type = si[DG_SI_TYPE];
      M3 R; _Pragma("unroll") for (int q = 0; q < 9; q++) R.m[q] = s[RS_R + q];
      const V3 pp = v3(s[RS_P], s[RS_P + 1], s[RS_P + 2]);
      if (type == DG_SHAPE_SPHERE) ray_sphere(pc, d, pp, sf[DG_SF_PARAMS], h, k);
      else if (type == DG_SHAPE_BOX) ray_box(pc, d, R, pp, sf[DG_SF_PARAMS], sf[DG_SF_PARAMS + 1], sf[DG_SF_PARAMS + 2], h, k);
      else if (type == DG_SHAPE_CAPSULE) { const V3 ax = v3(R.m[2], R.m[5], R.m[8]) * sf[DG_SF_PARAMS + 1]; ray_capsule(pc, d, pp - ax, pp + ax, sf[DG_SF_PARAMS], h, k); }
      else if (!(diag & 4)) ray_hull(pc, d, R, pp, PLN + 4 * si[DG_SI_PLANE_OFF], si[DG_SI_N_PLANES], h, k);
    }
    const bool hit = h.shape >= 0; const size_t o = (size_t)env * W * H + (size_t)row * W + col;
    if (depth) depth[o] = hit ? -h.t : -zf;
    if (seg) { int v = -1; if (hit) { cip si = SI + h.shape * DG_SI_STRIDE; v = si[DG_SI_BODY] + (((si[DG_SI_FLAGS] >> 8) & 0xFFFF) << 24); } seg[o] = v; }
    if (rgb) {
      float colr[3] = {0.75f, 0.75f, 0.75f};
      if (hit) { float eb[12 + DG_TX_STRIDE]; cfp e = tb + h.shape * RS_STRIDE; _Pragma("unroll") for (int q = 0; q < 12; q++) eb[q] = e[RS_R + q]; _Pragma("unroll") for (int q = 0; q < DG_TX_STRIDE; q++) eb[12 + q] = e[RS_COLOR + q]; shade_pixel(pc, h, d, eb, colr); }
      rgb[3 * o] = colr[0]; rgb[3 * o + 1] = colr[1]; rgb[3 * o + 2] = colr[2];
    }
  }
}

// WPE: wavefronts per SIMD the register allocation is held to (216 registers fit 2; 3 spills 41 of them -- measured)
template <int WPE>
__global__ __launch_bounds__(256, WPE) void render_kernel(DevScene sc, cip CI, cfp CF, cfp PLN, int cam, int ncam, cfp table, float* rgb, float* depth, int32_t* seg,
                                                      int band_rows, int nbands, int diag) {
  const int no_cull = diag & 1;  // diagnostics (DG_RENDER_DIAG): 1 test every shape for every pixel group, 2 skip every intersection, 4 skip hulls
  __shared__ float s_f[DG_RL_CAP][RL_STRIDE]; __shared__ int s_i[DG_RL_CAP][RLI_STRIDE]; __shared__ __align__(16) float s_pl[DG_RP_CAP][4] /* 16-byte aligned: a face is one ds_read_b128 */; __shared__ float s_pt[DG_RT_CAP][3];
  __shared__ float s_bb[DG_RL_CAP][4];  // image-space bounds of each entry, in pixel coordinates: [c min, c max, r min, r max]
  __shared__ int s_wave_count[4]; __shared__ int s_scan[4], s_scan2[4];
  const int env = blockIdx.x / nbands, band = blockIdx.x - env * nbands, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  cip ci = CI + cam * DG_CI_STRIDE; cfp cf = CF + cam * DG_CF_STRIDE;
  const int W = ci[DG_CI_WIDTH], H = ci[DG_CI_HEIGHT];
  const int r0 = band * band_rows, nrows = min(band_rows, H - r0), npx = nrows * W;
  cfp tb = table + (size_t)env * (sc.nsh * RS_STRIDE + ncam * RC_STRIDE); cfp cp = tb + sc.nsh * RS_STRIDE + cam * RC_STRIDE;
  M3 Rc; _Pragma("unroll") for (int k = 0; k < 9; k++) Rc.m[k] = cp[k];
  const V3 pc = v3(cp[9], cp[10], cp[11]);
  const float zn = cf[DG_CF_NEAR], zf = cf[DG_CF_FAR], th = cf[DG_CF_TAN_HALF_FOV], aspect = (float)W / (float)H;
  // pixel (col + 0.5, row + 0.5) -> xn = 2 c / W - 1, yn = 1 - 2 r / H, dir = Rc (xn th aspect, yn th, -1) = A + c B + r C
  const float kx = th * aspect, sx = 2.0f * kx / (float)W, sy = -2.0f * th / (float)H;
  const V3 rc0 = v3(Rc.m[0], Rc.m[3], Rc.m[6]), rc1 = v3(Rc.m[1], Rc.m[4], Rc.m[7]), rc2 = v3(Rc.m[2], Rc.m[5], Rc.m[8]);
  const V3 rayA = rc1 * th - rc0 * kx - rc2, rayB = rc0 * sx, rayC = rc1 * sy;
  const float lenB = sqrtf(dot(rayB, rayB)), lenC = sqrtf(dot(rayC, rayC));
  auto ray = [&](float c, float r) { return rayA + rayB * c + rayC * r; };
  // conservative cone around the rectangle [c0, c1] x [q0, q1] of pixel coordinates: axis through its centre, and
  // every ray of the rectangle is the centre ray plus at most rho, so sin(angle) <= rho / |centre ray|
  auto cone_of = [&](float c0, float c1, float q0, float q1, V3& axis, float& cos_t, float& sin_t) {
    const V3 a = ray(0.5f * (c0 + c1), 0.5f * (q0 + q1)); const float ia = rsqrtf(dot(a, a));
    axis = a * ia; sin_t = fminf((0.5f * (c1 - c0) * lenB + 0.5f * (q1 - q0) * lenC) * ia, 1.0f); cos_t = sqrtf(fmaxf(1.0f - sin_t * sin_t, 0.f));
  };
  auto cone_pass = [&](V3 v, float R, V3 axis, float cos_t, float sin_t) {
    const float d2 = dot(v, v);
    if (d2 <= R * R) return true;
    const float inv = rsqrtf(d2), cos_a = dot(v, axis) * inv, sin_b = R * inv, cos_b = sqrtf(fmaxf(1.0f - sin_b * sin_b, 0.f));
    const float cos_sum = cos_t * cos_b - sin_t * sin_b, sin_sum = sin_t * cos_b + cos_t * sin_b;  // cos / sin (theta + beta)
    return sin_sum < 0.f || cos_sum <= -1.0f || cos_a >= cos_sum - 1e-4f;
  };
  const float inv_w = 1.0f / (float)W;
  auto row_col = [&](int p, int& row, int& col) { row = (int)(((float)p + 0.5f) * inv_w); col = p - row * W; if (col < 0) { row--; col += W; } else if (col >= W) { row++; col -= W; } };
  const unsigned long long t_start = DG_RNOW(); (void)t_start;
  // ---------------- phase A: the band's list
  int total = 0, total_planes = 0, total_points = 0;
  {
    V3 baxis; float bcos, bsin; cone_of(0.f, (float)W, (float)r0, (float)(r0 + nrows), baxis, bcos, bsin);
    for (int chunk = 0; chunk < sc.nsh; chunk += 256) {
      const int sh = chunk + tid; bool pass = false; V3 v = v3(0.f, 0.f, 0.f); float Rb = 0.f;
      if (sh < sc.nsh) {
        cfp s = tb + sh * RS_STRIDE; v = v3(s[RS_C], s[RS_C + 1], s[RS_C + 2]) - pc; Rb = s[RS_BOUND];
        // (a shape wholly nearer than the near plane -- behind the camera, or around it like the link it is mounted on -- is only
        // ever entered at a depth below the near distance: it neither shows nor hides anything, see RayHit)
        pass = no_cull || (cone_pass(v, Rb, baxis, bcos, bsin) && -dot(rc2, v) + Rb >= zn);
        if ((diag & 256) && sc.SI[sh * DG_SI_STRIDE + DG_SI_BODY] == ci[DG_CI_BODY]) pass = false;  // (experiment: the camera's own body left out)
      }
      const unsigned long long m = __ballot(pass);
      if (lane == 0) s_wave_count[wv] = __popcll(m);
      __syncthreads();
      int off = total; for (int k = 0; k < wv; k++) off += s_wave_count[k];
      if (pass) {
        const int idx = off + __popcll(m & ((1ull << lane) - 1ull));
        if (idx < DG_RL_CAP) {
          cfp s = tb + sh * RS_STRIDE; cip si = sc.SI + sh * DG_SI_STRIDE; cfp sf = sc.SF + sh * DG_SF_STRIDE; float* o = s_f[idx];
          o[RL_V] = v.x; o[RL_V + 1] = v.y; o[RL_V + 2] = v.z; o[RL_BOUND] = Rb;
          _Pragma("unroll") for (int q = 0; q < 9; q++) o[RL_R + q] = s[RS_R + q];
          o[RL_P] = s[RS_P]; o[RL_P + 1] = s[RS_P + 1]; o[RL_P + 2] = s[RS_P + 2];
          o[RL_PRM] = sf[DG_SF_PARAMS]; o[RL_PRM + 1] = sf[DG_SF_PARAMS + 1]; o[RL_PRM + 2] = sf[DG_SF_PARAMS + 2];
          _Pragma("unroll") for (int q = 0; q < DG_TX_STRIDE; q++) o[RL_TEX + q] = s[RS_COLOR + q];
          s_i[idx][RLI_SEG] = si[DG_SI_BODY] + (((si[DG_SI_FLAGS] >> 8) & 0xFFFF) << 24);
          s_i[idx][RLI_TYPE] = si[DG_SI_TYPE]; s_i[idx][RLI_SHAPE] = sh; s_i[idx][RLI_NP] = si[DG_SI_TYPE] == DG_SHAPE_POINTS ? si[DG_SI_N_PLANES] : (si[DG_SI_TYPE] == DG_SHAPE_BOX ? 6 : 0); s_i[idx][RLI_PLANE_OFF] = 0;
          s_i[idx][RLI_NPT] = si[DG_SI_TYPE] == DG_SHAPE_POINTS ? si[DG_SI_N_POINTS] : (si[DG_SI_TYPE] == DG_SHAPE_BOX ? 8 : 0); s_i[idx][RLI_PT_OFF] = 0;
        }
      }
      total += s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
      __syncthreads();
    }
    // where each surviving hull's faces go: exclusive scan of the face counts over the list (<= 96 entries: two wavefronts)
    if (total <= DG_RL_CAP) {
      const int np = tid < total ? s_i[tid][RLI_NP] : 0, npt = tid < total ? s_i[tid][RLI_NPT] : 0; int incl = np, incl2 = npt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(incl, o), t3 = __shfl_up(incl2, o); if (lane >= o) { incl += t2; incl2 += t3; } }
      if (lane == 63) { s_scan[wv] = incl; s_scan2[wv] = incl2; }
      __syncthreads();
      int base = 0, base2 = 0; for (int k = 0; k < wv; k++) { base += s_scan[k]; base2 += s_scan2[k]; }
      if (tid < total) { s_i[tid][RLI_PLANE_OFF] = base + incl - np; s_i[tid][RLI_PT_OFF] = base2 + incl2 - npt; }
      total_planes = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3]; total_points = s_scan2[0] + s_scan2[1] + s_scan2[2] + s_scan2[3];
      __syncthreads();
    }
  }
  const bool overflow = total > DG_RL_CAP || total_planes > DG_RP_CAP || total_points > DG_RT_CAP;  // (then: the slow path below, straight from the tables)
  if (overflow) { render_band_slow(sc.SI, sc.SF, sc.nsh, tb, PLN, pc, rayA, rayB, rayC, W, H, r0, npx, zn, zf, env, diag, tid, rgb, depth, seg); return; }
  {
    // faces of the surviving hulls and boxes -> (world normal, signed distance of the eye); one wavefront per entry at a time
    // A point w (relative to the eye) at depth z = -rc2 . w > 0 is seen by the ray of pixel coordinates
    // c = (x / z + kx) / sx, r = (y / z - th) / sy; a convex shape wholly in front of the eye projects inside the box of
    // its projected vertices.  Anything that reaches behind the eye keeps an unbounded box (the tests of phase B deal
    // with it).  Half a pixel of padding covers the rounding of the projection.
    const float isx = 1.0f / sx, isy = 1.0f / sy, BIG = 3.0e38f;
    // the band's four side planes through the eye, outward normals: a hull / box with every vertex outside one of them is not in
    // the band at all and leaves the list here, once, instead of failing the same test strip by strip
    V3 bnL, bnR, bnT, bnB;
    { const V3 eL = rayA, eR = rayA + rayB * (float)W, eT = rayA + rayC * (float)r0, eB = rayA + rayC * (float)(r0 + nrows);
      bnL = cross(eL, rayC); bnR = cross(eR, rayC); bnT = cross(eT, rayB); bnB = cross(eB, rayB);
      if (dot(bnL, rayB) > 0.f) bnL = -bnL;
      if (dot(bnR, rayB) < 0.f) bnR = -bnR;
      if (dot(bnT, rayC) > 0.f) bnT = -bnT;
      if (dot(bnB, rayC) < 0.f) bnB = -bnB; }
    auto to_c = [&](float tx) { return (tx + kx) * isx; };
    auto to_r = [&](float ty) { return (ty - th) * isy; };  // (sy < 0: r decreases with ty)
    for (int e = wv; e < total; e += 4) {
      const int np = s_i[e][RLI_NP];
      if (np == 0) {  // sphere / capsule: the camera-aligned cube around its bounding sphere
        if (lane == 0) {
          const V3 v = v3(s_f[e][RL_V], s_f[e][RL_V + 1], s_f[e][RL_V + 2]); const float Rb = s_f[e][RL_BOUND];
          const float x = dot(rc0, v), y = dot(rc1, v), z = -dot(rc2, v);
          float b0 = -BIG, b1 = BIG, b2 = -BIG, b3 = BIG;
          if (z - Rb > 1e-5f) {
            const float in = 1.0f / (z - Rb), ifar = 1.0f / (z + Rb);
            const float xh = x + Rb, xl = x - Rb, yh = y + Rb, yl = y - Rb;
            const float txh = xh >= 0.f ? xh * in : xh * ifar, txl = xl >= 0.f ? xl * ifar : xl * in;
            const float tyh = yh >= 0.f ? yh * in : yh * ifar, tyl = yl >= 0.f ? yl * ifar : yl * in;
            b0 = to_c(txl) - 0.5f; b1 = to_c(txh) + 0.5f; b2 = to_r(tyh) - 0.5f; b3 = to_r(tyl) + 0.5f;
          }
          s_bb[e][0] = b0; s_bb[e][1] = b1; s_bb[e][2] = b2; s_bb[e][3] = b3;
        }
        continue;
      }
      const int sh = s_i[e][RLI_SHAPE], po = s_i[e][RLI_PLANE_OFF]; const bool box = s_i[e][RLI_TYPE] == DG_SHAPE_BOX;
      cfp planes = PLN + 4 * sc.SI[sh * DG_SI_STRIDE + DG_SI_PLANE_OFF];
      M3 Rl; _Pragma("unroll") for (int q = 0; q < 9; q++) Rl.m[q] = s_f[e][RL_R + q];
      const V3 ol = tmul(Rl, pc - v3(s_f[e][RL_P], s_f[e][RL_P + 1], s_f[e][RL_P + 2]));
      bool outside = false;
      auto face = [&](int k, V3& n, float& dist) {
        float d0;
        if (box) { const int ax = k >> 1; const float sg = (k & 1) ? -1.f : 1.f; n = v3(ax == 0 ? sg : 0.f, ax == 1 ? sg : 0.f, ax == 2 ? sg : 0.f); d0 = -s_f[e][RL_PRM + ax]; }  // face +-axis: n . x - half <= 0
        else { n = v3(planes[4 * k], planes[4 * k + 1], planes[4 * k + 2]); d0 = planes[4 * k + 3]; }
        dist = dot(n, ol) + d0;
      };
      // stable partition: the faces that have the eye on their outer side first, the others behind them.  "Outer side" with
      // the margin of the eye-inside test below: a camera mounted ON a face of its own link (from_the_readme's gripper
      // camera) has a signed distance of +-1 ulp there, and which sign comes out must not decide whether that link hides
      // the whole picture -- within a micrometre of a face the eye counts as behind it (fp64, the oracle, gets -0).
      int nout = 0;
      for (int k0 = 0; k0 < np; k0 += 64) { V3 n; float dist = -1.f; if (k0 + lane < np) face(k0 + lane, n, dist); nout += __popcll(__ballot(dist > 1e-6f)); }
      int at_out = 0, at_in = nout;
      for (int k0 = 0; k0 < np; k0 += 64) {
        const bool have = k0 + lane < np; V3 n = v3(0.f, 0.f, 1.f); float dist = -1.f; if (have) face(k0 + lane, n, dist);
        const bool out = have && dist > 1e-6f, in = have && !out;
        const unsigned long long mo = __ballot(out), mi = __ballot(in), below = (1ull << lane) - 1ull;
        if (have) {
          const int at = out ? at_out + __popcll(mo & below) : at_in + __popcll(mi & below); const V3 nw = mul(Rl, n);
          s_pl[po + at][0] = nw.x; s_pl[po + at][1] = nw.y; s_pl[po + at][2] = nw.z; s_pl[po + at][3] = dist;
          outside = outside || !(dist < -1e-6f);
        }
        at_out += __popcll(mo); at_in += __popcll(mi);
      }
      if (lane == 0) s_i[e][RLI_NOUT] = nout;
      // the eye inside a hull: no ray can ENTER it (ray_hull wants tn > 0); a box is entered from inside never either (ray_box: tn <= 0)
      if (!__any(outside) && !no_cull && lane == 0) s_i[e][RLI_TYPE] = -1;
      // vertices relative to the eye, for the tile-frustum test of phase B (hull points; the eight corners of a box)
      { const int npt = s_i[e][RLI_NPT], pto = s_i[e][RLI_PT_OFF]; cfp pts = sc.PF + 3 * sc.SI[sh * DG_SI_STRIDE + DG_SI_POINT_OFF];
        const V3 pl = v3(s_f[e][RL_P], s_f[e][RL_P + 1], s_f[e][RL_P + 2]) - pc;
        float txl = BIG, txh = -BIG, tyl = BIG, tyh = -BIG, zmax = -BIG; bool behind = false, inL = false, inR = false, inT = false, inB = false;
        for (int k = lane; k < npt; k += 64) {
          V3 q;
          if (box) q = v3((k & 1) ? s_f[e][RL_PRM] : -s_f[e][RL_PRM], (k & 2) ? s_f[e][RL_PRM + 1] : -s_f[e][RL_PRM + 1], (k & 4) ? s_f[e][RL_PRM + 2] : -s_f[e][RL_PRM + 2]);
          else q = v3(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]);
          const V3 w = pl + mul(Rl, q); s_pt[pto + k][0] = w.x; s_pt[pto + k][1] = w.y; s_pt[pto + k][2] = w.z;
          inL = inL || dot(bnL, w) <= 0.f; inR = inR || dot(bnR, w) <= 0.f; inT = inT || dot(bnT, w) <= 0.f; inB = inB || dot(bnB, w) <= 0.f;
          const float z = -dot(rc2, w); behind = behind || !(z > 1e-5f); zmax = fmaxf(zmax, z);
          const float iz = 1.0f / fmaxf(z, 1e-5f), tx = dot(rc0, w) * iz, ty = dot(rc1, w) * iz;
          txl = fminf(txl, tx); txh = fmaxf(txh, tx); tyl = fminf(tyl, ty); tyh = fmaxf(tyh, ty);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { txl = fminf(txl, __shfl_xor(txl, o)); txh = fmaxf(txh, __shfl_xor(txh, o)); tyl = fminf(tyl, __shfl_xor(tyl, o)); tyh = fmaxf(tyh, __shfl_xor(tyh, o)); zmax = fmaxf(zmax, __shfl_xor(zmax, o)); }
        if (npt > 0 && zmax < zn && !no_cull && lane == 0) s_i[e][RLI_TYPE] = -1;  // every vertex nearer than the near plane: so is the whole convex shape
        if (npt > 0 && !no_cull && (!__any(inL) || !__any(inR) || !__any(inT) || !__any(inB)) && lane == 0) s_i[e][RLI_TYPE] = -1;  // wholly outside the band's frustum
        const bool unbounded = __any(behind) || npt == 0;
        if (lane == 0) { s_bb[e][0] = unbounded ? -BIG : to_c(txl) - 0.5f; s_bb[e][1] = unbounded ? BIG : to_c(txh) + 0.5f; s_bb[e][2] = unbounded ? -BIG : to_r(tyh) - 0.5f; s_bb[e][3] = unbounded ? BIG : to_r(tyl) + 0.5f; }
      }
    }
    __syncthreads();
  }
  if (wv == 0) DG_RTIME(4, t_start);  // [4] phase A, wavefront 0
  struct Px { V3 d; float idd; RayHit h; int row, col; bool inside; };
  Px px[2];
  const int n_entries = total;
  // ---------------- phase B: a STRIP of eight full image rows per wavefront at a time, 16 x 8 tiles inside it (lane: column
  // lane & 15, rows lane >> 4 and (lane >> 4) + 4).  The strip comes first: its list is culled once against the strip's
  // cone, frustum and separating faces (the tests a tile makes, 13 times less often), the tiles then only look at the
  // strip's survivors, and a strip without any -- most of a picture of the sky -- is one contiguous piece of each output
  // image and is filled with 16-byte stores.
  const bool wide = (W & 3) == 0 && !(diag & 64);  // every 4-pixel piece of a row is 16-byte aligned (diag 64: scalar stores, for tests)
  const int ntx = (W + 15) >> 4, nstrips = (nrows + 7) >> 3;
  const size_t img = (size_t)env * W * H;
  // The strips are taken DG_RTILE_CAP tiles at a time (whole strips; a 200 x 200 picture is one such chunk).  Per chunk:
  // B1a: a wavefront per strip culls the band's list against the strip (cone, image-space boxes, separating faces, frustum);
  // B1b: a THREAD per 16 x 8 tile narrows its strip's mask with the tile's own cone and rectangle -- the tile's candidates;
  // B1c: a wavefront per strip writes the background of every tile that has none: a strip without any -- most of a picture of
  //      the sky -- is one contiguous piece of each output image; in the others the empty tiles' pieces of each row, as
  //      16-byte stores either way;
  // B2:  the tiles that do have candidates go through a queue (an LDS counter): the wavefront that is free takes the next one,
  //      so a picture whose content sits in a few tiles -- a hand in a corner of the sky -- is spread over all four.
  __shared__ unsigned long long s_smask[DG_RS_CAP][2]; __shared__ unsigned long long s_tmask[DG_RTILE_CAP][2];
  __shared__ int s_queue[DG_RTILE_CAP]; __shared__ int s_nq, s_next_tile;
  const int spc = max(1, min((int)DG_RS_CAP, (int)DG_RTILE_CAP / ntx));  // strips per chunk (dg_world_render refuses pictures wider than 16 DG_RTILE_CAP)
  const bool tile_cull = !no_cull && !(diag & 128);  // (diag 128: no strip / tile level culling, every tile looks at the whole list)
  const float4 bg_d = make_float4(-zf, -zf, -zf, -zf), bg_c = make_float4(0.75f, 0.75f, 0.75f, 0.75f); const int4 bg_s = make_int4(-1, -1, -1, -1);
  for (int s0 = 0; s0 < nstrips; s0 += spc) {
  const int ns = min(spc, nstrips - s0), ntl = ns * ntx;
  if (tid == 0) { s_nq = 0; s_next_tile = 0; }
  // B1a: a THREAD per (strip, entry) pair -- the strip's cone and rows against the entry's bounding sphere and image box, then for
  // hulls and boxes the separating-face test (a face with the eye outside that no ray of the strip's cone approaches) and the
  // strip's four side planes against the vertices.  (A wavefront per strip with the lanes over the faces of one candidate at a
  // time made the same decisions at 7.6 k instructions per wavefront instead of ~1.5 k.)
  for (int i = tid; i < 2 * ns; i += 256) s_smask[i >> 1][i & 1] = tile_cull ? 0ull : ~0ull;
  __syncthreads();
  if (tile_cull) for (int pidx = tid; pidx < ns * total; pidx += 256) {
    const int ls = pidx / total, j = pidx - ls * total, type = s_i[j][RLI_TYPE];
    if (type < 0) continue;
    const int q0 = r0 + 8 * (s0 + ls), qn = min(8, r0 + nrows - q0);  // rows [q0, q0 + qn)
    if (!(s_bb[j][3] >= (float)q0 && s_bb[j][2] <= (float)(q0 + qn))) continue;
    V3 saxis; float scos, ssin; cone_of(0.f, (float)W, (float)q0, (float)(q0 + qn), saxis, scos, ssin);
    if (!cone_pass(v3(s_f[j][RL_V], s_f[j][RL_V + 1], s_f[j][RL_V + 2]), s_f[j][RL_BOUND], saxis, scos, ssin)) continue;
    bool keep = true;
    if (type == DG_SHAPE_BOX || type == DG_SHAPE_POINTS) {
      const int po = s_i[j][RLI_PLANE_OFF], nout = s_i[j][RLI_NOUT];
      for (int f = 0; f < nout && keep; f++) keep = !(s_pl[po + f][3] > 0.f && s_pl[po + f][0] * saxis.x + s_pl[po + f][1] * saxis.y + s_pl[po + f][2] * saxis.z >= ssin + 1e-5f);
      const int pto = s_i[j][RLI_PT_OFF], npt = s_i[j][RLI_NPT];
      if (keep && npt > 0) {
        const V3 eL = rayA, eR = rayA + rayB * (float)W, eT = rayA + rayC * (float)q0, eB = rayA + rayC * (float)(q0 + qn);
        V3 nL = cross(eL, rayC), nR = cross(eR, rayC), nT = cross(eT, rayB), nB = cross(eB, rayB);
        if (dot(nL, rayB) > 0.f) nL = -nL;
        if (dot(nR, rayB) < 0.f) nR = -nR;
        if (dot(nT, rayC) > 0.f) nT = -nT;
        if (dot(nB, rayC) < 0.f) nB = -nB;
        bool inL = false, inR = false, inT = false, inB = false;
        for (int f = 0; f < npt; f++) {
          const V3 w = v3(s_pt[pto + f][0], s_pt[pto + f][1], s_pt[pto + f][2]);
          inL = inL || dot(nL, w) <= 0.f; inR = inR || dot(nR, w) <= 0.f; inT = inT || dot(nT, w) <= 0.f; inB = inB || dot(nB, w) <= 0.f;
        }
        keep = inL && inR && inT && inB;
      }
    }
    if (keep) atomicOr(&s_smask[ls][j >> 6], 1ull << (j & 63));
  }
  __syncthreads();
  for (int t = tid; t < ntl; t += 256) {  // B1b
    const int ls = t / ntx, txi = t - ls * ntx, q0 = r0 + 8 * (s0 + ls), qn = min(8, r0 + nrows - q0), c0 = txi << 4, c1 = min(c0 + 16, W);
    unsigned long long m[2] = {s_smask[ls][0], s_smask[ls][1]};
    if (tile_cull && (m[0] | m[1])) {
      V3 axis; float cos_t, sin_t; cone_of((float)c0, (float)c1, (float)q0, (float)(q0 + qn), axis, cos_t, sin_t);
#pragma unroll
      for (int hf = 0; hf < 2; hf++) for (unsigned long long mm = m[hf]; mm; mm &= mm - 1) {
        const int bit = __ffsll((long long)mm) - 1, j = 64 * hf + bit;
        const bool boxed = s_bb[j][1] >= (float)c0 && s_bb[j][0] <= (float)c1 && s_bb[j][3] >= (float)q0 && s_bb[j][2] <= (float)(q0 + qn);
        if (!(boxed && cone_pass(v3(s_f[j][RL_V], s_f[j][RL_V + 1], s_f[j][RL_V + 2]), s_f[j][RL_BOUND], axis, cos_t, sin_t))) m[hf] &= ~(1ull << bit);
      }
    }
    if (diag & 2) m[0] = m[1] = 0ull;  // (diag 2: no intersections at all, every tile is background)
    s_tmask[t][0] = m[0]; s_tmask[t][1] = m[1];
    if (m[0] | m[1]) s_queue[atomicAdd(&s_nq, 1)] = t;
  }
  __syncthreads();
  for (int ls = wv; ls < ns; ls += 4) {  // B1c
    const int q0 = r0 + 8 * (s0 + ls), qn = min(8, r0 + nrows - q0); const size_t o0 = img + (size_t)q0 * W;
    bool some = false; for (int x = lane; x < ntx; x += 64) some = some || (s_tmask[ls * ntx + x][0] | s_tmask[ls * ntx + x][1]) != 0ull;
    if (!__any(some)) {  // nothing in sight: the strip's rows are one contiguous piece of every image
      const int count = qn * W;
      if (wide) {
        if (depth) for (int i = lane; i < count / 4; i += 64) reinterpret_cast<float4*>(depth + o0)[i] = bg_d;
        if (seg) for (int i = lane; i < count / 4; i += 64) reinterpret_cast<int4*>(seg + o0)[i] = bg_s;
        if (rgb) for (int i = lane; i < 3 * count / 4; i += 64) reinterpret_cast<float4*>(rgb + 3 * o0)[i] = bg_c;
      } else {
        if (depth) for (int i = lane; i < count; i += 64) depth[o0 + i] = -zf;
        if (seg) for (int i = lane; i < count; i += 64) seg[o0 + i] = -1;
        if (rgb) for (int i = lane; i < 3 * count; i += 64) rgb[3 * o0 + i] = 0.75f;
      }
    } else if (wide) {  // the empty tiles' pieces of each row (a tile is 16 pixels wide: 4 pieces of a depth row, 12 of an rgb row)
      for (int pc4 = lane; pc4 < W / 4; pc4 += 64) {
        const int x = pc4 >> 2; if (s_tmask[ls * ntx + x][0] | s_tmask[ls * ntx + x][1]) continue;
        for (int r = 0; r < qn; r++) {
          if (depth) reinterpret_cast<float4*>(depth + o0 + (size_t)r * W)[pc4] = bg_d;
          if (seg) reinterpret_cast<int4*>(seg + o0 + (size_t)r * W)[pc4] = bg_s;
        }
      }
      if (rgb) for (int pc4 = lane; pc4 < 3 * W / 4; pc4 += 64) {
        const int x = (pc4 * 43691) >> 19;  // pc4 / 12 (exact below 10 922)
        if (s_tmask[ls * ntx + x][0] | s_tmask[ls * ntx + x][1]) continue;
        for (int r = 0; r < qn; r++) reinterpret_cast<float4*>(rgb + 3 * (o0 + (size_t)r * W))[pc4] = bg_c;
      }
    } else {
      for (int c = lane; c < W; c += 64) {
        const int x = c >> 4; if (s_tmask[ls * ntx + x][0] | s_tmask[ls * ntx + x][1]) continue;
        for (int r = 0; r < qn; r++) {
          const size_t o = o0 + (size_t)r * W + c;
          if (depth) depth[o] = -zf;
          if (seg) seg[o] = -1;
          if (rgb) { rgb[3 * o] = 0.75f; rgb[3 * o + 1] = 0.75f; rgb[3 * o + 2] = 0.75f; }
        }
      }
    }
  }
  const int nq = s_nq;
  if (wv == 0) { DG_RTIME(6, t_start); if ((diag & 16) && lane == 0) atomicAdd(&g_render_count[8], (unsigned long long)nq); }  // [6] up to the end of B1 (cumulative), [8] queued tiles
  const unsigned long long t_b2 = DG_RNOW(); (void)t_b2;
  for (;;) {
    int qi = 0; if (lane == 0) qi = atomicAdd(&s_next_tile, 1);
    qi = __builtin_amdgcn_readfirstlane(qi);
    if (qi >= nq) break;
    const int tile = s_queue[qi], ls = tile / ntx, txi = tile - ls * ntx;
    const int q0 = r0 + 8 * (s0 + ls), qn = min(8, r0 + nrows - q0);
    const unsigned long long tm[2] = {s_tmask[tile][0], s_tmask[tile][1]};
  {
    const int c0 = txi << 4;  // tile: columns [c0, c0 + 16) of the strip
    V3 axis; float cos_t, sin_t;
    cone_of((float)c0, (float)min(c0 + 16, W), (float)q0, (float)(q0 + qn), axis, cos_t, sin_t);
    // the tile's four side planes through the eye, outward normals (rays are A + c B + r C, so the plane of a column
    // edge is spanned by its ray at row 0 and C, that of a row edge by its ray at column 0 and B)
    const V3 eL = rayA + rayB * (float)c0, eR = rayA + rayB * (float)min(c0 + 16, W), eT = rayA + rayC * (float)q0, eB = rayA + rayC * (float)(q0 + qn);
    V3 nL = cross(eL, rayC), nR = cross(eR, rayC), nT = cross(eT, rayB), nB = cross(eB, rayB);
    if (dot(nL, rayB) > 0.f) nL = -nL;
    if (dot(nR, rayB) < 0.f) nR = -nR;
    if (dot(nT, rayC) > 0.f) nT = -nT;
    if (dot(nB, rayC) < 0.f) nB = -nB;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int prow = (lane >> 4) + 4 * u, pcol = c0 + (lane & 15);
      px[u].inside = prow < qn && pcol < W;
      px[u].row = q0 + min(prow, qn - 1); px[u].col = min(pcol, W - 1);
      px[u].d = ray(px[u].col + 0.5f, px[u].row + 0.5f); px[u].idd = __frcp_rn(dot(px[u].d, px[u].d));
      px[u].h.t = zf; px[u].h.shape = -1; px[u].h.n = v3(0.f, 0.f, 1.f); px[u].h.tmin = (diag & 32) ? 1e-30f : zn;
    }
    for (int base = 0; base < n_entries; base += 64) {
      const int j = min(base + lane, n_entries - 1);
      const bool cand = base + lane < n_entries && ((tm[(base >> 6) & 1] >> lane) & 1ull) && s_i[j][RLI_TYPE] >= 0;  // the tile's candidates: its mask of B1b
      for (unsigned long long mm = __ballot(cand); mm; mm &= mm - 1) {
        const int jj = base + __ffsll((long long)mm) - 1;  // wave-uniform
        {
          const float* e = s_f[jj]; const int type = s_i[jj][RLI_TYPE], k = jj;  // (a hit remembers the LIST entry: what the pixel's colour needs is in LDS too)
          DG_RCOUNT(type);  // 0..3: candidates after the sphere-cone test, by type
          const V3 oc = v3(e[RL_V], e[RL_V + 1], e[RL_V + 2]); const float bound = e[RL_BOUND];
          if ((type == DG_SHAPE_BOX || type == DG_SHAPE_POINTS) && !no_cull) {
            // separating face for the whole tile: the eye on its outer side (s > 0) and every ray of the cone moving
            // away from it or along it (n . d >= 0 for all d within the cone <=> n . axis >= sin_t for unit n)
            const int po = s_i[jj][RLI_PLANE_OFF], nout = s_i[jj][RLI_NOUT]; bool sep = false;  // (only a face with the eye outside can separate)
            for (int f0 = 0; f0 < nout; f0 += 64) { const int f = min(f0 + lane, nout - 1); sep = sep || (s_pl[po + f][3] > 0.f && s_pl[po + f][0] * axis.x + s_pl[po + f][1] * axis.y + s_pl[po + f][2] * axis.z >= sin_t + 1e-5f); }
            if (__any(sep)) continue;
            DG_RCOUNT(4 + type);  // after the separating-face test
            // every vertex outside one of the tile's side planes: the whole convex shape is (lane l tests vertex l)
            const int pto = s_i[jj][RLI_PT_OFF], npt = s_i[jj][RLI_NPT]; bool inL = false, inR = false, inT = false, inB = false;
            for (int f0 = 0; f0 < npt; f0 += 64) {
              const bool have = f0 + lane < npt; const int f = min(f0 + lane, npt - 1); const V3 w = v3(s_pt[pto + f][0], s_pt[pto + f][1], s_pt[pto + f][2]);
              inL = inL || (have && dot(nL, w) <= 0.f); inR = inR || (have && dot(nR, w) <= 0.f); inT = inT || (have && dot(nT, w) <= 0.f); inB = inB || (have && dot(nB, w) <= 0.f);
            }
            if (npt > 0 && (!__any(inL) || !__any(inR) || !__any(inT) || !__any(inB))) continue;
            DG_RCOUNT(8 + type);  // after the frustum test
          }
          bool far[2] = {false, false};  // the ray passes the entry's bounding sphere by
          if (type != DG_SHAPE_BOX && !no_cull) {  // per-pixel bounding-sphere test before anything else
#pragma unroll
            for (int u = 0; u < 2; u++) { const float tc = dot(oc, px[u].d) * px[u].idd; const V3 qv = oc - px[u].d * tc; far[u] = !(dot(qv, qv) <= bound * bound); }
            if (!__any(!far[0] || !far[1])) continue;
          }
          DG_RCOUNT(12 + type);  // intersected
          if (type == DG_SHAPE_POINTS) {
            if (!(diag & 4)) { const int po = s_i[jj][RLI_PLANE_OFF], np = s_i[jj][RLI_NP], nout = s_i[jj][RLI_NOUT];
              const V3 dd[2] = {px[0].d, px[1].d}; RayHit hh[2] = {px[0].h, px[1].h};
              ray_hull_world2(dd, &s_pl[po], nout, np, hh, k); if (!(diag & 1024)) { px[0].h = hh[0]; px[1].h = hh[1]; } }  // (diag 1024: intersected, result dropped -- timing)
            continue;
          }
          M3 R; _Pragma("unroll") for (int q = 0; q < 9; q++) R.m[q] = e[RL_R + q];
          const V3 pp = v3(e[RL_P], e[RL_P + 1], e[RL_P + 2]); const float p0 = e[RL_PRM], p1 = e[RL_PRM + 1], p2 = e[RL_PRM + 2];
#pragma unroll
          for (int u = 0; u < 2; u++) {
            if (type == DG_SHAPE_SPHERE) ray_sphere(pc, px[u].d, pp, p0, px[u].h, k);
            else if (type == DG_SHAPE_BOX) ray_box(pc, px[u].d, R, pp, p0, p1, p2, px[u].h, k);
            else { const V3 ax = v3(R.m[2], R.m[5], R.m[8]) * p1; ray_capsule(pc, px[u].d, pp - ax, pp + ax, p0, px[u].h, k); }
          }
        }
      }
    }
    if (wide && !__any(px[0].h.shape >= 0 || px[1].h.shape >= 0)) {  // every ray missed after all: the tile's background as 16-byte stores
      const int np4 = (min(c0 + 16, W) - c0) >> 2;
      { const int r = lane >> 2, pcx = lane & 3;
        if (r < qn && pcx < np4) {
          const size_t o = img + (size_t)(q0 + r) * W + c0;
          if (depth) reinterpret_cast<float4*>(depth + o)[pcx] = bg_d;
          if (seg) reinterpret_cast<int4*>(seg + o)[pcx] = bg_s;
        } }
      if (rgb) { const int sr = (lane * 43691) >> 19, pcx = lane - 12 * sr;  // lane / 12: four rows at a time, 12 pieces each
        if (sr < 4 && pcx < 3 * np4) for (int r = sr; r < qn; r += 4) reinterpret_cast<float4*>(rgb + 3 * (img + (size_t)(q0 + r) * W + c0))[pcx] = bg_c; }
      continue;
    }
    float vdepth[2]; int vsegm[2]; float vcol[2][3];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const RayHit& h = px[u].h;
      const bool hit = h.shape >= 0;
      vdepth[u] = hit ? -h.t : -zf; vsegm[u] = -1; vcol[u][0] = vcol[u][1] = vcol[u][2] = 0.75f;
      // (no global LOAD in this phase -- the first one would wait for every background store the wavefront issued in B1c: the
      // colours of the shape that was hit come from its list entry)
      if (seg && hit) vsegm[u] = s_i[h.shape][RLI_SEG];
      if (rgb && hit) {
        float eb[12 + DG_TX_STRIDE]; const float* e = s_f[h.shape];
        _Pragma("unroll") for (int q = 0; q < 12; q++) eb[q] = e[RL_R + q];
        _Pragma("unroll") for (int q = 0; q < DG_TX_STRIDE; q++) eb[12 + q] = e[RL_TEX + q];
        shade_pixel(pc, h, px[u].d, eb, vcol[u]);
      }
    }
    {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        if (!px[u].inside) continue;
        const size_t o = img + (size_t)px[u].row * W + px[u].col;
        if (depth) depth[o] = vdepth[u];
        if (seg) seg[o] = vsegm[u];
        if (rgb) { rgb[3 * o] = vcol[u][0]; rgb[3 * o + 1] = vcol[u][1]; rgb[3 * o + 2] = vcol[u][2]; }
      }
    }
  }
  }
  DG_RTIME(10, t_b2);  // [10] B2, summed over the four wavefronts
  __syncthreads();  // (the next chunk of strips reuses the masks and the queue)
  if (wv == 0) DG_RTIME(5, t_b2);  // [5] B2 until the slowest wavefront is done
  }
}

#endif  // DG_DEFINE_RENDER_KERNEL

}  // namespace dg
