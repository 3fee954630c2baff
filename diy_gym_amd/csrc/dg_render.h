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

enum { RS_R = 0, RS_P = 9, RS_C = 12, RS_BOUND = 15, RS_STRIDE = 16, RC_STRIDE = 12 };

template <int LANES>
__global__ __launch_bounds__(64) void pose_kernel(DevScene sc, MotorTable mt, float* state, int ncam, cip CI, cfp CF, float* table, float* gws) {
  extern __shared__ float smem[];
  constexpr int ACTIVE = envs_per_wave(LANES);
  const int lane = threadIdx.x; if (lane >= ACTIVE) return;
  const int env = blockIdx.x * ACTIVE + lane; if (env >= sc.num_envs) return;
  Lane<LANES> ln(sc, mt, workspace_of<LANES>(sc, smem, gws, lane), state + env, env, false);
  for (int b = 0; b < sc.nb; b++) ln.kinematics(b);
  float* out = table + (size_t)env * (sc.nsh * RS_STRIDE + ncam * RC_STRIDE);
  for (int sh = 0; sh < sc.nsh; sh++) {
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
  }
  for (int c = 0; c < ncam; c++) {
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

struct RayHit { float t; V3 n; int shape; };

DGD void ray_sphere(V3 o, V3 d, V3 c, float r, RayHit& h, int sh) {
  const V3 oc = o - c; const float a = dot(d, d), b = dot(oc, d), cc = dot(oc, oc) - r * r, disc = b * b - a * cc;
  if (disc < 0.f) return;
  const float t = fdiv(-b - sqrtf(disc), a);
  if (t > 0.f && t < h.t) { h.t = t; h.n = ((o + d * t) - c) * __frcp_rn(r); h.shape = sh; }
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
  if (tn > tf || tn <= 0.f || tn >= h.t) return;
  h.t = tn; h.n = mul(R, v3(ax == 0 ? sg : 0.f, ax == 1 ? sg : 0.f, ax == 2 ? sg : 0.f)); h.shape = sh;
}
DGD void ray_capsule(V3 o, V3 d, V3 e0, V3 e1, float r, RayHit& h, int sh) {
  const V3 ax = e1 - e0; const float L2 = dot(ax, ax);
  if (L2 > 1e-24f) {
    const V3 oc = o - e0; const float dax = dot(d, ax), oax = dot(oc, ax);
    const float iL2 = __frcp_rn(L2); const float a = dot(d, d) - dax * dax * iL2, b = dot(oc, d) - oax * dax * iL2, c = dot(oc, oc) - oax * oax * iL2 - r * r, disc = b * b - a * c;
    if (a > 1e-24f && disc >= 0.f) {
      const float t = fdiv(-b - sqrtf(disc), a), s = (oax + t * dax) * iL2;
      if (t > 0.f && t < h.t && s >= 0.f && s <= 1.f) { h.t = t; h.n = ((o + d * t) - (e0 + ax * s)) * __frcp_rn(r); h.shape = sh; }
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
  if (miss || tn > tf || tn <= 0.f || tn >= h.t) return;
  h.t = tn; h.n = mul(Rl, nn); h.shape = sh;
}

#ifdef DG_DEFINE_RENDER_KERNEL  // defined in exactly one translation unit (dg_api.hip)
// One 16 x 16 pixel tile per workgroup.  Phase 1: the 256 threads cull the env's shapes against the tile's viewing
// cone (bounding spheres) and compact the survivors IN SHAPE ORDER into an LDS list, so ties between coincident
// surfaces resolve exactly as in a brute-force loop.  Phase 2: every pixel intersects only the listed shapes; the
// shape index is wave-uniform, so poses and parameters come through scalar loads.
#define DG_TILE 32  /* pixels per tile side; 256 threads x 4 pixels each */
__global__ __launch_bounds__(256) void render_kernel(DevScene sc, cip CI, cfp CF, cfp PLN, int cam, int ncam, cfp table, float* rgb, float* depth, int32_t* seg) {
  __shared__ int s_list[1024]; __shared__ int s_wave_count[4]; __shared__ int s_total;
  const int env = blockIdx.y, tid = threadIdx.x; cip ci = CI + cam * DG_CI_STRIDE; cfp cf = CF + cam * DG_CF_STRIDE;
  const int W = ci[DG_CI_WIDTH], H = ci[DG_CI_HEIGHT]; const int tiles_x = (W + DG_TILE - 1) / DG_TILE;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  cfp tb = table + (size_t)env * (sc.nsh * RS_STRIDE + ncam * RC_STRIDE); cfp cp = tb + sc.nsh * RS_STRIDE + cam * RC_STRIDE;
  M3 Rc; _Pragma("unroll") for (int k = 0; k < 9; k++) Rc.m[k] = cp[k];
  const V3 pc = v3(cp[9], cp[10], cp[11]);
  const float zn = cf[DG_CF_NEAR], zf = cf[DG_CF_FAR], th = tanf(0.5f * cf[DG_CF_FOV] * 0.017453292519943295f), aspect = (float)W / (float)H;
  auto ray = [&](float c, float r) { const float xn = (c / W) * 2.0f - 1.0f, yn = 1.0f - (r / H) * 2.0f; return mul(Rc, v3(xn * th * aspect, yn * th, -1.0f)); };
  // ---- phase 1: tile cone = axis through the tile centre, half angle to the farthest corner
  const float c0 = tx * DG_TILE, c1 = fminf((float)(tx + 1) * DG_TILE, (float)W), r0 = ty * DG_TILE, r1 = fminf((float)(ty + 1) * DG_TILE, (float)H);
  V3 axis = ray(0.5f * (c0 + c1), 0.5f * (r0 + r1)); axis = axis * rsqrtf(dot(axis, axis));
  float cos_t = 1.0f;
  { const V3 k0 = ray(c0, r0), k1 = ray(c1, r0), k2 = ray(c0, r1), k3 = ray(c1, r1);
    cos_t = fminf(fminf(dot(k0, axis) * rsqrtf(dot(k0, k0)), dot(k1, axis) * rsqrtf(dot(k1, k1))), fminf(dot(k2, axis) * rsqrtf(dot(k2, k2)), dot(k3, axis) * rsqrtf(dot(k3, k3)))); }
  const float sin_t = sqrtf(fmaxf(1.0f - cos_t * cos_t, 0.f));
  int total = 0;
  for (int base = 0; base < sc.nsh; base += 256) {
    const int sh = base + tid; bool pass = false;
    if (sh < sc.nsh) {
      cfp s = tb + sh * RS_STRIDE; const V3 v = v3(s[RS_C], s[RS_C + 1], s[RS_C + 2]) - pc; const float R = s[RS_BOUND], d2 = dot(v, v);
      if (d2 <= R * R) pass = true;
      else {
        const float inv = rsqrtf(d2), cos_a = dot(v, axis) * inv, sin_b = R * inv, cos_b = sqrtf(fmaxf(1.0f - sin_b * sin_b, 0.f));
        const float cos_sum = cos_t * cos_b - sin_t * sin_b, sin_sum = sin_t * cos_b + cos_t * sin_b;  // cos / sin (theta + beta)
        pass = sin_sum < 0.f || cos_sum <= -1.0f || cos_a >= cos_sum - 1e-4f;
      }
    }
    const unsigned long long m = __ballot(pass); const int wv = tid >> 6, ln_ = tid & 63;
    if (ln_ == 0) s_wave_count[wv] = __popcll(m);
    __syncthreads();
    int off = total; for (int k = 0; k < wv; k++) off += s_wave_count[k];
    if (pass) { const int idx = off + __popcll(m & ((1ull << ln_) - 1ull)); if (idx < 1024) s_list[idx] = sh; }
    total += s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
    __syncthreads();
  }
  total = min(total, 1024);
  (void)s_total;
  // ---- phase 2: four pixels per thread (16 x 16 sub-tiles)
  for (int sub = 0; sub < 4; sub++) {
  const int col = tx * DG_TILE + (tid & 15) + 16 * (sub & 1), row = ty * DG_TILE + (tid >> 4) + 16 * (sub >> 1);
  if (col >= W || row >= H) continue;
  const V3 d = ray(col + 0.5f, row + 0.5f); const float idd = __frcp_rn(dot(d, d));
  RayHit h; h.t = zf; h.shape = -1; h.n = v3(0.f, 0.f, 1.f);
  for (int q = 0; q < total; q++) {
    const int k = __builtin_amdgcn_readfirstlane(s_list[q]);
    cfp s = tb + k * RS_STRIDE; cip si = sc.SI + k * DG_SI_STRIDE; cfp sf = sc.SF + k * DG_SF_STRIDE;
    const int type = si[DG_SI_TYPE];
    if (type == DG_SHAPE_POINTS || type == DG_SHAPE_CAPSULE) {
      // per-pixel bounding-sphere test before the expensive ones (a hull has dozens of face planes)
      const V3 oc = v3(s[RS_C], s[RS_C + 1], s[RS_C + 2]) - pc; const float tc = dot(oc, d) * idd; const V3 qv = oc - d * tc; const float bound = s[RS_BOUND];
      if (!__any(dot(qv, qv) <= bound * bound)) continue;
    }
    M3 R; _Pragma("unroll") for (int j = 0; j < 9; j++) R.m[j] = s[RS_R + j];
    const V3 p = v3(s[RS_P], s[RS_P + 1], s[RS_P + 2]);
    if (type == DG_SHAPE_SPHERE) ray_sphere(pc, d, p, sf[DG_SF_PARAMS], h, k);
    else if (type == DG_SHAPE_BOX) ray_box(pc, d, R, p, sf[DG_SF_PARAMS], sf[DG_SF_PARAMS + 1], sf[DG_SF_PARAMS + 2], h, k);
    else if (type == DG_SHAPE_CAPSULE) { const V3 ax = v3(R.m[2], R.m[5], R.m[8]) * sf[DG_SF_PARAMS + 1]; ray_capsule(pc, d, p - ax, p + ax, sf[DG_SF_PARAMS], h, k); }
    else ray_hull(pc, d, R, p, PLN + 4 * si[DG_SI_PLANE_OFF], si[DG_SI_N_PLANES], h, k);
  }
  const bool hit = h.shape >= 0 && h.t >= zn; const size_t px = (size_t)env * W * H + (size_t)row * W + col;
  if (depth) depth[px] = hit ? -h.t : -zf;
  if (seg) {
    int v = -1;
    if (hit) { cip si = sc.SI + h.shape * DG_SI_STRIDE; v = si[DG_SI_BODY] + (((si[DG_SI_FLAGS] >> 8) & 0xFFFF) << 24); }
    seg[px] = v;
  }
  if (rgb) {
    float c0r = 0.75f, c1r = 0.75f, c2r = 0.75f;
    if (hit) {
      cfp colr = sc.BF + sc.SI[h.shape * DG_SI_STRIDE + DG_SI_BODY] * DG_BF_STRIDE + DG_BF_COLOR;  // per-lane index: vector loads
      const float nl = h.n.x * 0.30151134457776363f + h.n.y * 0.30151134457776363f + h.n.z * 0.9045340337332909f, shd = 0.4f + 0.6f * fmaxf(nl, 0.f);
      c0r = colr[0] * shd; c1r = colr[1] * shd; c2r = colr[2] * shd;
    }
    rgb[3 * px] = c0r; rgb[3 * px + 1] = c1r; rgb[3 * px + 2] = c2r;
  }
  }
}

#endif  // DG_DEFINE_RENDER_KERNEL

}  // namespace dg
