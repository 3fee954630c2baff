// dg_hull.h -- convex hull against convex hull (DG_HF_HULL_CONTACTS): what pybullet does with two URDF collision meshes
// (reference model.py:65 loadURDF -> Bullet's btConvexConvexAlgorithm on the meshes' convex hulls [R]).
//
//   C = A - B (Minkowski difference), support s_C(d) = s_A(d) - s_B(-d), the hull points read through wave-uniform addresses
//   (both hulls of a candidate pair are the same in every lane; the poses differ);
//   GJK: closest point v of C to the origin over a simplex of <= 4 support points (Voronoi-region tests of Ericson, Real-Time
//        Collision Detection 5.1.5 / 5.1.6): distance |v|, normal v / |v| (from B towards A), witness points from the
//        barycentric weights; a lane leaves as soon as v . w / |v| -- a lower bound of its distance -- exceeds what it asks for;
//   EPA: origin inside C or nearer than HH_SWITCH: an expanding polytope inside C with SIGNED plane distances of the origin, started
//        from a tetrahedron of four support points of its own; the face with the smallest distance is pushed out to its support point
//        until it is a face of C.  Its vertices, faces and horizon edges are arrays with per-lane sizes and indices in a device
//        buffer of the world: the rare, slow part -- only lanes whose hulls interpenetrate beyond their 2 x 1 mm of margin get here.
// Same steps, same constants, same tie-breaks (strict comparisons, lowest index first) as the CPU checker's hull_hull (the tests
// compare the two, and both with a brute-force Minkowski difference); coordinates are relative to A's frame origin.  The simplex of the GJK part lives in registers: every index into it
// is a compile-time constant, a run-time position is a chain of selects.
#pragma once
#include "dg_device.h"

namespace dg {

// The two routines are inlined into the narrow phase: as functions of their own (one copy per kernel instead of one per call site)
// they made every step kernel a kernel WITH CALLS -- stack pointer and scratch descriptor reserved, a different register
// allocation throughout -- and the contact-free headline scene, which never enters them, ran 25 % slower in every section.
#ifndef HH_FN
#define HH_FN DGD
#endif
#define HH_GJK_ITERS 32
#define HH_EPA_ITERS 24
#define HH_EPA_MAXV (4 + HH_EPA_ITERS)
#define HH_EPA_MAXF (2 * HH_EPA_MAXV)
#define HH_EPA_MAXE 96
#define HH_FACE_PTS 8
#define HH_SWITCH 1e-4f

// compile-time loop: every index into the simplex arrays below must be a constant or the arrays land in scratch memory (a
// `#pragma unroll` the optimizer declines leaves run-time indices behind -- it did, inside the GJK loop: 800 bytes of scratch per
// lane and, worse, a different spilling strategy for the whole kernel)
template <int I, int N, class F> DGD void hh_for(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); hh_for<I + 1, N>(f); } }
typedef const float __attribute__((address_space(4)))* hh_cfp;
// tabled: lane k of the wavefront holds hull point k of A in (tax, tay, taz) and of B in (tbx, tby, tbz) -- loaded once per pair with
// every lane active (hull_tables); the support loops then read a point with three v_readlane instead of a scalar load whose
// round trip to the scalar cache a lone wavefront cannot hide (32 + 32 dependent loads per support: 25 k cycles, measured as
// 11 x the step time of arms in contact).  Not tabled (the reset kernel steps under a per-env mask; hulls of more points than
// active lanes): the points come through wave-uniform scalar loads.
struct HullPairD { hh_cfp pa, pb; int na, nb; M3 RA, RB; V3 tBA; float tax, tay, taz, tbx, tby, tbz; bool tabled; float* ew /* this lane's column of its wavefront's polytope workspace (hull_ws_of) */; };
DGD void hull_tables(HullPairD& h, bool tabled) {
  const int lane = threadIdx.x & 63; h.tabled = tabled; h.tax = h.tay = h.taz = h.tbx = h.tby = h.tbz = 0.f;
  if (tabled) {
    const int ka = min(lane, h.na - 1), kb = min(lane, h.nb - 1);
    h.tax = h.pa[3 * ka]; h.tay = h.pa[3 * ka + 1]; h.taz = h.pa[3 * ka + 2]; h.tbx = h.pb[3 * kb]; h.tby = h.pb[3 * kb + 1]; h.tbz = h.pb[3 * kb + 2];
  }
}
struct HullHit { V3 pa, pb, n; float dist; bool hit; int iters /* GJK iterations this lane needed (diagnostics) */; };
struct HV { V3 w, a, b; int id; };
// component-wise selects: `c ? a : b` on two structs selects between their ADDRESSES, which keeps both (and every array one of them
// is an element of) in scratch memory -- the simplex arrays of the GJK loop were, 84 scratch round trips per iteration: 28 k cycles
// an iteration instead of 7 k
DGD V3 hh_sel(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
DGD HV hh_sel(bool c, const HV& a, const HV& b) { HV r; r.w = hh_sel(c, a.w, b.w); r.a = hh_sel(c, a.a, b.a); r.b = hh_sel(c, a.b, b.b); r.id = c ? a.id : b.id; return r; }

// (four points per round, their loads in flight together)
DGD int hh_argmax(hh_cfp p, int n, V3 d, V3& pt) {
  int bi = 0; float best = -3.0e38f; V3 bp = v3(0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < n; k0 += 4) {
    float q[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++) { const int k = min(k0 + j, n - 1); q[j][0] = p[3 * k]; q[j][1] = p[3 * k + 1]; q[j][2] = p[3 * k + 2]; }
#pragma unroll
    for (int j = 0; j < 4; j++) {  // (a round's spare slots repeat the last point: never strictly better)
      const float sd = q[j][0] * d.x + q[j][1] * d.y + q[j][2] * d.z; const bool g = sd > best;
      best = g ? sd : best; bi = g ? min(k0 + j, n - 1) : bi; bp.x = g ? q[j][0] : bp.x; bp.y = g ? q[j][1] : bp.y; bp.z = g ? q[j][2] : bp.z;
    }
  }
  pt = bp; return bi;
}
DGD int hh_argmax_tab(float tx, float ty, float tz, int n, V3 d, V3& pt) {
  // (the loop carries the best index only; the point itself is fetched afterwards with three ds_bpermute from the lane tables --
  // every lane of the wavefront is active here, the GJK loop is wave-uniform)
  int bi = 0; float best = -3.0e38f;
  auto rl = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
  for (int k = 0; k < n; k++) {
    const float x = rl(tx, k), y = rl(ty, k), z = rl(tz, k), sd = x * d.x + y * d.y + z * d.z; const bool g = sd > best;
    best = g ? sd : best; bi = g ? k : bi;
  }
  pt = v3(__shfl(tx, bi), __shfl(ty, bi), __shfl(tz, bi)); return bi;
}
template <bool TAB = true>
DGD HV hh_support(const HullPairD& h, V3 d) {
  HV o; V3 qa, qb; int ia, ib;
  if (TAB && h.tabled) { ia = hh_argmax_tab(h.tax, h.tay, h.taz, h.na, tmul(h.RA, d), qa); ib = hh_argmax_tab(h.tbx, h.tby, h.tbz, h.nb, tmul(h.RB, -d), qb); }
  else { ia = hh_argmax(h.pa, h.na, tmul(h.RA, d), qa); ib = hh_argmax(h.pb, h.nb, tmul(h.RB, -d), qb); }
  o.a = mul(h.RA, qa); o.b = mul(h.RB, qb) + h.tBA; o.w = o.a - o.b; o.id = (ia << 8) | ib; return o;
}
// barycentric weights of the point of triangle (a, b, c) closest to the origin (Ericson 5.1.5)
DGD void hh_closest_tri(V3 a, V3 b, V3 c, float (&l)[3]) {
  const V3 ab = b - a, ac = c - a;
  const float d1 = -dot(ab, a), d2 = -dot(ac, a);
  l[0] = l[1] = l[2] = 0.f;
  if (d1 <= 0.f && d2 <= 0.f) { l[0] = 1.f; return; }
  const float d3 = -dot(ab, b), d4 = -dot(ac, b);
  if (d3 >= 0.f && d4 <= d3) { l[1] = 1.f; return; }
  const float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { const float t = fdiv(d1, d1 - d3); l[0] = 1.f - t; l[1] = t; return; }
  const float d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.f && d5 <= d6) { l[2] = 1.f; return; }
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { const float t = fdiv(d2, d2 - d6); l[0] = 1.f - t; l[2] = t; return; }
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) { const float t = fdiv(d4 - d3, (d4 - d3) + (d5 - d6)); l[1] = 1.f - t; l[2] = t; return; }
  const float den = frcp(va + vb + vc); l[1] = vb * den; l[2] = vc * den; l[0] = 1.f - l[1] - l[2];
}
// point of the simplex (n vertices, in slots 0 .. n - 1) closest to the origin: weights l; true: the origin is inside a tetrahedron
DGD bool hh_closest_simplex(const V3 (&w)[4], int n, float (&l)[4]) {
  l[0] = l[1] = l[2] = l[3] = 0.f;
  if (n == 1) { l[0] = 1.f; return false; }
  if (n == 2) {
    const V3 ab = w[1] - w[0]; const float den = dot(ab, ab); float t = den > 0.f ? fdiv(-dot(w[0], ab), den) : 0.f;
    t = fminf(fmaxf(t, 0.f), 1.f); l[0] = 1.f - t; l[1] = t; return false;
  }
  if (n == 3) { float lt[3]; hh_closest_tri(w[0], w[1], w[2], lt); l[0] = lt[0]; l[1] = lt[1]; l[2] = lt[2]; return false; }
  // tetrahedron (Ericson 5.1.6): the faces that have the origin on their outer side (a flat tetrahedron: every face)
  float best = 3.0e38f; bool inside = true;
  hh_for<0, 4>([&](auto FI) {
    constexpr int f = decltype(FI)::value;
    constexpr int F[4][4] = {{0, 1, 2, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {1, 3, 2, 0}};
    const V3 a = w[F[f][0]], b = w[F[f][1]], c = w[F[f][2]], d = w[F[f][3]];
    const V3 nn = cross(b - a, c - a);
    const float sp = -dot(a, nn), sd = dot(d - a, nn), scale = dot(nn, nn) * dot(d - a, d - a);
    const bool flat = sd * sd <= 1e-10f * scale, outside = flat || sp * sd < 0.f;
    if (outside) {
      inside = false;
      float lt[3]; hh_closest_tri(a, b, c, lt);
      const V3 q = a * lt[0] + b * lt[1] + c * lt[2]; const float qq = dot(q, q);
      if (qq < best) { best = qq; l[0] = l[1] = l[2] = l[3] = 0.f; l[F[f][0]] = lt[0]; l[F[f][1]] = lt[1]; l[F[f][2]] = lt[2]; }
    }
  });
  return inside;
}

// ---- the expanding polytope.  Its vertices, faces and horizon edges are arrays with per-lane sizes and indices: they live in a
// device buffer the world owns (DevScene::hull_ws), one block of HH_WS_SLOTS x 64 floats per wavefront of the step grid laid out
// [slot][lane] -- as private arrays they made every step kernel carry 3 KB of scratch per lane, and the launch of the contact-free
// headline scene 25 % slower for a routine none of its lanes ever entered.  Called under divergence (only the lanes that
// need it): the support points come through scalar-free loads here, not the lane tables (an inactive lane's table
// registers are not restored around the call).
enum { HW_VW = 0 /* vertex w: 3 x MAXV */, HW_VID = 3 * HH_EPA_MAXV, HW_FV = HW_VID + HH_EPA_MAXV /* face: vertex indices, 8 bits each, bit 24 = alive */,
       HW_FN = HW_FV + HH_EPA_MAXF /* normal: 3 x MAXF */, HW_FD = HW_FN + 3 * HH_EPA_MAXF, HW_ED = HW_FD + HH_EPA_MAXF, HW_PW = HW_ED + HH_EPA_MAXE /* coplanar points: 3 x FACE_PTS */,
       HW_PID = HW_PW + 3 * HH_FACE_PTS, HH_WS_SLOTS = HW_PID + HH_FACE_PTS };
// this lane's column of its wavefront's block
DGD float* hull_ws_of(float* ws) { return ws ? ws + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (size_t)(HH_WS_SLOTS * 64) + (threadIdx.x & 63) : nullptr; }
struct HEpa {
  float* e;
  DGD float& F(int slot) const { return e[slot * 64]; }
  DGD int& I(int slot) const { return reinterpret_cast<int*>(e)[slot * 64]; }
  DGD V3 W(int k) const { return v3(F(HW_VW + 3 * k), F(HW_VW + 3 * k + 1), F(HW_VW + 3 * k + 2)); }
  DGD void put(int k, const HV& s) const { F(HW_VW + 3 * k) = s.w.x; F(HW_VW + 3 * k + 1) = s.w.y; F(HW_VW + 3 * k + 2) = s.w.z; I(HW_VID + k) = s.id; }
  DGD V3 N(int f) const { return v3(F(HW_FN + 3 * f), F(HW_FN + 3 * f + 1), F(HW_FN + 3 * f + 2)); }
  DGD void face(int f, int i, int j, int k, V3 g) const {
    const V3 a = W(i), b = W(j), c = W(k); V3 nn = cross(b - a, c - a); const float len = norm(nn);
    if (!(len > 1e-12f)) { I(HW_FV + f) = i | (j << 8) | (k << 16) | (1 << 24); F(HW_FN + 3 * f) = 0.f; F(HW_FN + 3 * f + 1) = 0.f; F(HW_FN + 3 * f + 2) = 1.f; F(HW_FD + f) = 3.0e38f; return; }  // a sliver: kept for the topology, never the closest
    nn = nn * frcp(len);
    const bool flip = dot(nn, a - g) < 0.f;  // outward: away from the interior point g
    if (flip) nn = -nn;
    I(HW_FV + f) = i | ((flip ? k : j) << 8) | ((flip ? j : k) << 16) | (1 << 24); F(HW_FN + 3 * f) = nn.x; F(HW_FN + 3 * f + 1) = nn.y; F(HW_FN + 3 * f + 2) = nn.z; F(HW_FD + f) = dot(nn, a);
  }
};
// signed distance of the origin from the boundary of C along its nearest face (> 0: inside, the penetration depth), that face's
// outward normal and the witness points of the origin's projection onto it
HH_FN float hh_epa(const HullPairD& h, V3 seed, bool have_start, const V3 (&sw)[4], const int (&sid)[4], V3& n_out, V3& pa, V3& pb) {
  const HEpa E = {h.ew}; int nv = 0, nf = 0;
  // start: the tetrahedron GJK ended with when it found the origin inside one (it already has a face near the origin); else a
  // tetrahedron of C built here: two opposite support points, the one farthest from their line, the one farthest from their plane
  if (have_start) { hh_for<0, 4>([&](auto K) { constexpr int k = decltype(K)::value; HV s; s.w = sw[k]; s.id = sid[k]; s.a = s.w; s.b = s.w; E.put(k, s); }); }
  else {
  const V3 d0 = dot(seed, seed) > 1e-12f ? seed * frsq(dot(seed, seed)) : v3(1.f, 0.f, 0.f);
  const HV s0 = hh_support<false>(h, d0), s1 = hh_support<false>(h, -d0); E.put(0, s0); E.put(1, s1);
  const V3 e = s1.w - s0.w;
  const V3 ax = (fabsf(e.x) <= fabsf(e.y) && fabsf(e.x) <= fabsf(e.z)) ? v3(1.f, 0.f, 0.f) : (fabsf(e.y) <= fabsf(e.z) ? v3(0.f, 1.f, 0.f) : v3(0.f, 0.f, 1.f));
  V3 d1 = cross(e, ax); d1 = d1 * frcp(norm(d1) + 1e-37f);
  { const HV c1 = hh_support<false>(h, d1), c2 = hh_support<false>(h, -d1); E.put(2, hh_sel(fabsf(dot(c1.w - s0.w, d1)) >= fabsf(dot(c2.w - s0.w, d1)), c1, c2)); }
  V3 nn = cross(e, E.W(2) - s0.w); nn = nn * frcp(norm(nn) + 1e-37f);
  { const HV c1 = hh_support<false>(h, nn), c2 = hh_support<false>(h, -nn); E.put(3, hh_sel(fabsf(dot(c1.w - s0.w, nn)) >= fabsf(dot(c2.w - s0.w, nn)), c1, c2)); }
  }
  nv = 4;
  const V3 g = ((E.W(0) + E.W(1)) + (E.W(2) + E.W(3))) * 0.25f;
  E.face(0, 0, 1, 2, g); E.face(1, 0, 1, 3, g); E.face(2, 0, 2, 3, g); E.face(3, 1, 2, 3, g); nf = 4;
  int best = 0;
  for (int it = 0; it < HH_EPA_ITERS; it++) {
    best = -1; float bd = 3.0e38f;
    for (int f0 = 0; f0 < nf; f0 += 4) {  // (four faces per round, their loads issued together: a load at a time is a memory round trip per face)
      float fd4[4]; int fv4[4];
#pragma unroll
      for (int j = 0; j < 4; j++) { const int f = min(f0 + j, nf - 1); fd4[j] = E.F(HW_FD + f); fv4[j] = E.I(HW_FV + f); }
#pragma unroll
      for (int j = 0; j < 4; j++) if (f0 + j < nf && (fv4[j] >> 24) && fd4[j] < bd) { bd = fd4[j]; best = f0 + j; }
    }
    if (best < 0) { best = 0; break; }
    const V3 bn = E.N(best);
    const HV w = hh_support<false>(h, bn);
    if (dot(w.w, bn) - bd <= 1e-6f || nv >= HH_EPA_MAXV) break;  // the face lies on the boundary of C
    bool dup = false; for (int k = 0; k < nv; k++) dup = dup || E.I(HW_VID + k) == w.id;
    if (dup) break;
    E.put(nv, w);
    // faces that see the new point go; the edges that belonged to exactly one of them are the horizon
    int ne = 0; bool full = false;
    for (int f = 0; f < nf; f++) {
      const int fv = E.I(HW_FV + f); const float fd = E.F(HW_FD + f);
      if (!(fv >> 24) || fd >= 3.0e38f) continue;
      if (!(dot(E.N(f), w.w) - fd > 1e-9f) && f != best) continue;
      E.I(HW_FV + f) = fv & 0xFFFFFF;
      for (int q = 0; q < 3; q++) {
        const int i = (fv >> (8 * q)) & 255, j = (fv >> (8 * ((q + 1) % 3))) & 255, key = i < j ? (i | (j << 8)) : (j | (i << 8));
        int found = -1;
        for (int t = 0; t < ne; t++) if (E.I(HW_ED + t) == key) { found = t; break; }
        if (found >= 0) { E.I(HW_ED + found) = E.I(HW_ED + ne - 1); ne--; } else if (ne < HH_EPA_MAXE) { E.I(HW_ED + ne) = key; ne++; } else full = true;
      }
    }
    for (int t = 0; t < ne; t++) {
      int slot = -1; for (int f = 0; f < nf; f++) if (!(E.I(HW_FV + f) >> 24)) { slot = f; break; }
      if (slot < 0) { if (nf >= HH_EPA_MAXF) break; slot = nf++; }
      const int key = E.I(HW_ED + t); E.face(slot, key & 255, key >> 8, nv, g);
    }
    nv++;
    if (full) break;
  }
  // Witness points: the origin's projection p onto the face's plane as a combination of support points lying IN that plane.  The
  // triangle found is only part of C's face there (a parallelogram when two edges cross, a polygon when a face of one hull rests on
  // the other) and p may lie in another part of it: while p is outside every triangle of the coplanar points found so far, the
  // support point of a direction tilted from the normal towards p (1e-3 rad) is the face's corner on that side.
  const V3 fn = E.N(best); const float fd = E.F(HW_FD + best); const V3 p = fn * fd; int np = 3;
  for (int q = 0; q < 3; q++) { const int k = (E.I(HW_FV + best) >> (8 * q)) & 255; const V3 wk = E.W(k); E.F(HW_PW + 3 * q) = wk.x; E.F(HW_PW + 3 * q + 1) = wk.y; E.F(HW_PW + 3 * q + 2) = wk.z; E.I(HW_PID + q) = E.I(HW_VID + k); }
  auto P = [&](int k) { return v3(E.F(HW_PW + 3 * k), E.F(HW_PW + 3 * k + 1), E.F(HW_PW + 3 * k + 2)) - p; };
  int bi = 0, bj = 1, bk = 2; float bl[3] = {1.f, 0.f, 0.f};
  for (int round = 0; ; round++) {
    float bq = 3.0e38f; V3 qbest = p;
    for (int i = 0; i < np; i++) for (int j = i + 1; j < np; j++) for (int k = j + 1; k < np; k++) {
      float l[3]; hh_closest_tri(P(i), P(j), P(k), l);
      const V3 q = P(i) * l[0] + P(j) * l[1] + P(k) * l[2]; const float qq = dot(q, q);
      if (qq < bq) { bq = qq; qbest = q; bi = i; bj = j; bk = k; bl[0] = l[0]; bl[1] = l[1]; bl[2] = l[2]; }
    }
    if (bq <= 1e-12f || np >= HH_FACE_PTS || round >= HH_FACE_PTS) break;
    const HV w = hh_support<false>(h, fn + qbest * (-1e-3f * frsq(bq)));  // (qbest = nearest point - p: towards p is -qbest)
    if (fd - dot(w.w, fn) > 1e-4f) break;  // not in the plane (0.1 mm): the face ends before p
    bool dup = false; for (int k = 0; k < np; k++) dup = dup || E.I(HW_PID + k) == w.id;
    if (dup) break;
    E.F(HW_PW + 3 * np) = w.w.x; E.F(HW_PW + 3 * np + 1) = w.w.y; E.F(HW_PW + 3 * np + 2) = w.w.z; E.I(HW_PID + np) = w.id; np++;
  }
  auto wit = [&](int k, V3& a, V3& b) {  // the two hull points behind support point k (per-lane index: a vector load from the point table)
    const int id = E.I(HW_PID + k), ia = id >> 8, ib = id & 255;
    a = mul(h.RA, v3(h.pa[3 * ia], h.pa[3 * ia + 1], h.pa[3 * ia + 2])); b = mul(h.RB, v3(h.pb[3 * ib], h.pb[3 * ib + 1], h.pb[3 * ib + 2])) + h.tBA;
  };
  V3 a0, b0, a1, b1, a2, b2; wit(bi, a0, b0); wit(bj, a1, b1); wit(bk, a2, b2);
  pa = a0 * bl[0] + a1 * bl[1] + a2 * bl[2]; pb = b0 * bl[0] + b1 * bl[1] + b2 * bl[2];
  n_out = fn; return fd;
}

// signed distance of hull A from hull B (< 0: they overlap by that much), unit normal from B towards A, witness points (relative
// to A's frame origin).  hit = false once the distance is known to exceed max_dist, or for a lane that did not ask (active false).
HH_FN void hull_hull(const HullPairD& h, V3 seed, float max_dist, bool active, HullHit& out) {
  V3 sw[4], sa[4], sb[4]; int sid[4]; float l[4] = {0.f, 0.f, 0.f, 0.f}; int ns = 0;
  hh_for<0, 4>([&](auto K) { constexpr int k = decltype(K)::value; sw[k] = v3(0.f, 0.f, 0.f); sa[k] = sw[k]; sb[k] = sw[k]; sid[k] = -1; });
  V3 v = dot(seed, seed) > 1e-12f ? seed : v3(1.f, 0.f, 0.f); float vv = 3.0e38f; bool inside = false, far = false, done = !active; int my_iters = 0;
  for (int it = 0; it < HH_GJK_ITERS; it++) {
    if (!__any(!done)) break;
    my_iters += done ? 0 : 1;
    const HV w = hh_support(h, -v);
    if (!done && ns > 0) {
      const float vw = dot(v, w.w), vn = fsqrt(vv);
      if (vw > max_dist * vn) { far = true; done = true; }  // a separating plane farther than anyone asks
      bool dup = false;
      hh_for<0, 4>([&](auto K) { constexpr int k = decltype(K)::value; dup = dup || (k < ns && sid[k] == w.id); });
      if (dup || vv - vw <= 1e-6f * vv + 1e-7f * vn) done = true;  // no support point nearer along v: v is the closest point
    }
    if (!done) {
      V3 tw[4], ta[4], tb[4]; int ti[4];
      hh_for<0, 4>([&](auto K) { constexpr int k = decltype(K)::value; const bool nw = k == ns; tw[k] = hh_sel(nw, w.w, sw[k]); ta[k] = hh_sel(nw, w.a, sa[k]); tb[k] = hh_sel(nw, w.b, sb[k]); ti[k] = nw ? w.id : sid[k]; });
      float ln[4]; const bool in = hh_closest_simplex(tw, ns + 1, ln);
      const V3 nv = tw[0] * ln[0] + tw[1] * ln[1] + tw[2] * ln[2] + tw[3] * ln[3]; const float nvv = dot(nv, nv);
      hh_for<0, 4>([&](auto K) { constexpr int k = decltype(K)::value; sw[k] = hh_sel(in, tw[k], sw[k]); sid[k] = in ? ti[k] : sid[k]; });  // (inside: the tetrahedron the polytope search starts from)
      if (in) { inside = true; done = true; }
      else if (ns > 0 && !(nvv < vv)) done = true;  // (rounding: no progress -- keep the previous simplex)
      else {
        // keep the vertices that carry weight, in their order (slot m takes the m-th of them)
        int m = 0;
        hh_for<0, 4>([&](auto K) {
          constexpr int k = decltype(K)::value; const bool keep = k <= ns && ln[k] > 0.f;
          hh_for<0, k + 1>([&](auto J) { constexpr int j = decltype(J)::value; const bool here = keep && m == j;
            sw[j] = hh_sel(here, tw[k], sw[j]); sa[j] = hh_sel(here, ta[k], sa[j]); sb[j] = hh_sel(here, tb[k], sb[j]); sid[j] = here ? ti[k] : sid[j]; l[j] = here ? ln[k] : l[j]; });
          m += keep ? 1 : 0;
        });
        hh_for<0, 4>([&](auto J) { constexpr int j = decltype(J)::value; l[j] = j < m ? l[j] : 0.f; });  // (slots beyond the simplex keep stale vertices: no weight)
        ns = m; v = nv; vv = nvv;
        if (vv <= HH_SWITCH * HH_SWITCH) done = true;
      }
    }
  }
  const bool deep = active && !far && (inside || vv <= HH_SWITCH * HH_SWITCH);
  out.hit = active && !far; out.iters = my_iters;
  if (__any(deep)) {
    if (deep && h.ew) { V3 nf, pa, pb; const float d = hh_epa(h, seed, inside, sw, sid, nf, pa, pb); out.n = -nf; out.dist = -d; out.pa = pa; out.pb = pb; }
  }
  if (!deep || !h.ew) {  // (no polytope workspace -- a world without hull pairs never gets here: the GJK answer, ~0 along the last direction)
    const float vn = fsqrt(vv), iv = vn > 0.f ? frcp(vn) : 0.f; out.n = v * iv; out.dist = vn;
    out.pa = sa[0] * l[0] + sa[1] * l[1] + sa[2] * l[2] + sa[3] * l[3]; out.pb = sb[0] * l[0] + sb[1] * l[1] + sb[2] * l[2] + sb[3] * l[3];
  }
}

}  // namespace dg
