// dg_solver.h -- narrow phase, constraint rows, projected Gauss-Seidel, position
// update, inverse kinematics and the addon program, all per-lane on top of the
// Lane<> workspace of dg_kernels.h.
#pragma once
#include "dg_kernels.h"
#include "dg_hull.h"

namespace dg {

// ---------------------------------------------------------------- in-kernel stamps
// Diagnostic builds only (template PROF = true): wave-uniform cycle counters per section of the step.
// The production instantiation (PROF = false) contains no stamp at all.
enum { PS_UPDATE = 0, PS_KIN, PS_COLLIDE, PS_ABA, PS_MINV, PS_ROWS, PS_PGS, PS_INTEGRATE, PS_OUTPUT, PS_PGS_MOTOR, PS_PGS_LIMIT, PS_PGS_CONTACT, PS_COUNT,
       PS_ROW = PS_COUNT + 12 /* row pitch of the stamp buffer: + {B0, B0', B4, end} arrival of wavefronts 1..3 (helper-wave kernel) */ };
template <bool PROF> struct Prof;
template <> struct Prof<false> { DGD void start() {} DGD void stamp(int) {} };
template <> struct Prof<true> {
  unsigned long long last, acc[PS_COUNT];
  DGD void start() { _Pragma("unroll") for (int k = 0; k < PS_COUNT; k++) acc[k] = 0ull; last = __builtin_amdgcn_s_memtime(); }
  DGD void stamp(int id) {
    unsigned long long t = __builtin_amdgcn_s_memtime();
    _Pragma("unroll") for (int k = 0; k < PS_COUNT; k++) if (k == id) acc[k] += t - last;
    last = t;
  }
};

// ---------------------------------------------------------------- narrow phase
struct WShape { int type, body, glink; M3 R; V3 p; float prm0, prm1, prm2, mu; int poff, npts; };

template <int LANES>
DGD void shape_world(const Lane<LANES>& ln, int sh, WShape& o) {
  cip si = ln.sc.SI + sh * DG_SI_STRIDE; cfp sf = ln.sc.SF + sh * DG_SF_STRIDE;
  o.type = si[DG_SI_TYPE]; o.body = si[DG_SI_BODY]; o.glink = si[DG_SI_LINK]; o.poff = si[DG_SI_POINT_OFF]; o.npts = si[DG_SI_N_POINTS];
  M3 Rs; _Pragma("unroll") for (int k = 0; k < 9; k++) Rs.m[k] = sf[DG_SF_ROT + k];
  const V3 ps = v3(sf[DG_SF_POS], sf[DG_SF_POS + 1], sf[DG_SF_POS + 2]);
  if (si[DG_SI_FLAGS] & DG_SHAPE_WORLD) { o.R = Rs; o.p = ps; }  // frozen body: table holds world coordinates
  else { M3 Rl; V3 pl; ln.link_world(o.body, o.glink, Rl, pl); o.R = mul(Rl, Rs); o.p = pl + mul(Rl, ps); }
  o.prm0 = sf[DG_SF_PARAMS]; o.prm1 = sf[DG_SF_PARAMS + 1]; o.prm2 = sf[DG_SF_PARAMS + 2]; o.mu = sf[DG_SF_FRICTION];
}

struct Hit { bool hit; V3 pa, pb, n; float dist; };

DGD Hit sphere_box(V3 c, float r, const WShape& bx, float margin) {
  Hit h; V3 lc = tmul(bx.R, c - bx.p);
  float hx[3] = {bx.prm0, bx.prm1, bx.prm2}, l[3] = {lc.x, lc.y, lc.z}, cl[3]; bool inside = true;
#pragma unroll
  for (int k = 0; k < 3; k++) { cl[k] = fminf(fmaxf(l[k], -hx[k]), hx[k]); inside = inside && (cl[k] == l[k]); }
  V3 nl; float d;
  if (!inside) { V3 df = v3(l[0] - cl[0], l[1] - cl[1], l[2] - cl[2]); d = norm(df); nl = df * (1.0f / d); }
  else {
    int best = 0; float bd = 3.0e38f, sg = 1.f;
#pragma unroll
    for (int k = 0; k < 3; k++) { float dp = hx[k] - l[k], dm = l[k] + hx[k]; if (dp < bd) { bd = dp; best = k; sg = 1.f; } if (dm < bd) { bd = dm; best = k; sg = -1.f; } }
    nl = v3(best == 0 ? sg : 0.f, best == 1 ? sg : 0.f, best == 2 ? sg : 0.f); d = -bd;
#pragma unroll
    for (int k = 0; k < 3; k++) if (k == best) cl[k] = sg * hx[k];
  }
  h.hit = (d - r) < margin; h.n = mul(bx.R, nl); h.pb = bx.p + mul(bx.R, v3(cl[0], cl[1], cl[2])); h.pa = c - h.n * r; h.dist = d - r;
  return h;
}
DGD Hit sphere_sphere(V3 ca, float ra, V3 cb, float rb, float margin) {
  Hit h; V3 d = ca - cb; float len = norm(d);
  h.hit = (len - ra - rb) < margin; h.n = len > 1e-12f ? d * __frcp_rn(len) : v3(0, 0, 1);
  h.pa = ca - h.n * ra; h.pb = cb + h.n * rb; h.dist = len - ra - rb;
  return h;
}
DGD void seg_ends(const WShape& c, V3& e0, V3& e1) {
  V3 ax = v3(c.R.m[2], c.R.m[5], c.R.m[8]); e0 = c.p - ax * c.prm1; e1 = c.p + ax * c.prm1;
}
DGD float clamp01(float t) { return fminf(fmaxf(t, 0.f), 1.f); }
DGD V3 closest_on_seg(V3 a, V3 b, V3 p) {
  // (den > 1e-30, not > 0: v_rcp_f32 of a denormal is +inf)
  V3 ab = b - a; float den = dot(ab, ab); float t = den > 1e-30f ? clamp01(fdiv(dot(p - a, ab), den)) : 0.f; return a + ab * t;
}
DGD void seg_seg(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1, V3& c2) {
  V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2; float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), sN, tN; const float eps = 1e-12f;
  if (a <= eps && e <= eps) { c1 = p1; c2 = p2; return; }
  if (a <= eps) { sN = 0.f; tN = clamp01(fdiv(f, e)); }
  else {
    float c = dot(d1, r);
    if (e <= eps) { tN = 0.f; sN = clamp01(fdiv(-c, a)); }
    else {
      float b = dot(d1, d2), den = a * e - b * b;
      sN = den > eps ? clamp01(fdiv(b * f - c * e, den)) : 0.f;
      tN = fdiv(b * sN + f, e);
      if (tN < 0.f) { tN = 0.f; sN = clamp01(fdiv(-c, a)); } else if (tN > 1.f) { tN = 1.f; sN = clamp01(fdiv(b - c, a)); }
    }
  }
  c1 = p1 + d1 * sN; c2 = p2 + d2 * tN;
}

// contact list lives at sc.cont_off: slot 0 = count, then max_contacts entries of CL_STRIDE
template <int LANES>
DGD void emit_contact(const Lane<LANES>& ln, int list, int& cnt, int pair, const Hit& h, float flip, int feature = 0) {
  if (!h.hit || cnt >= ln.sc.max_contacts) return;
  int o = list + 1 + cnt * CL_STRIDE;
  ln.L(o + CL_PAIR) = (float)pair; ln.L(o + CL_KEY) = (float)DG_CONTACT_KEY(pair, feature);
  ln.L3set(o + CL_P, (h.pa + h.pb) * 0.5f);
  ln.L3set(o + CL_N, h.n * flip);
  ln.L(o + CL_DIST) = h.dist;
  cnt++;
}

// Round shapes (sphere / capsule / convex mesh via its fitted capsule) are reduced once per substep to a
// world-space segment + radius, cached in the transient LDS region (free until the dynamics pass):
//   [e0 3][e1 3][r][bounding radius] per shape.  Pairs whose bounding spheres are apart in every lane of the
// wave are skipped with one wave-uniform branch.
// narrow-phase cache per round shape: segment centre, half-axis (end points = centre -/+ half), radius, bounding radius
enum { SC_C = 0, SC_H = 3, SC_R = 6, SC_BOUND = 7, SC_STRIDE = 8 };

// TBL > 0: at least the first TBL lanes of the wavefront are active at the call (step kernels: 64, or the envs per
// wavefront of the sliced modes) -- pair descriptors are then fetched TBL at a time with one vector load and read
// back with v_readlane, instead of a chain of dependent scalar loads per pair.  TBL == 0 (reset kernel, which runs
// under a per-env mask): one scalar load per pair.
// Only the candidate pairs [pair_lo, pair_hi) are tested and their contacts go to the list at `list` (the helper-wave
// kernel cuts the pair table in two for two wavefronts; pair order, hence contact order, is kept by appending the
// second list to the first).
// SLC > 1 (lane-sliced step kernels): the call is made by all 64 lanes, `ln` is the lane's ENV column in the sweeps'
// grouping (lane = env * SLC + sl) -- the SLC lanes of an env do everything identically (same values to the same LDS
// slots) except the pairs of a group against a box frozen in the world (the ground plane, every wall), which they
// test SLC at a time, one pair per lane, and then append to the env's list in pair order.
template <int LANES, int TBL, int SLC = 1>
DGD int collide(const Lane<LANES>& ln, int pair_lo = 0, int pair_hi = 0x7fffffff, int list = -1, int sl = 0) {
  const DevScene& sc = ln.sc; int cnt = 0; const float margin = sc.HF[DG_HF_CONTACT_MARGIN];
  const bool hull_mode = sc.HF[DG_HF_HULL_CONTACTS] > 0.f; const float hmg = sc.HF[DG_HF_HULL_MARGIN];
  if (list < 0) list = sc.cont_off;
  if (sc.npairs == 0) { ln.L(list) = 0.f; return 0; }
  for (int sh = 0; sh < sc.nsha; sh++) {
    const int type = sc.SI[sh * DG_SI_STRIDE + DG_SI_TYPE]; if (type == DG_SHAPE_BOX) continue;
    WShape w; shape_world(ln, sh, w); V3 e0 = w.p, e1 = w.p;
    if (type != DG_SHAPE_SPHERE) seg_ends(w, e0, e1);
    const int o = sc.tr_off + sh * SC_STRIDE;
    ln.L3set(o + SC_C, (e0 + e1) * 0.5f); ln.L3set(o + SC_H, (e1 - e0) * 0.5f); ln.L(o + SC_R) = w.prm0;
    // (a hull's fitted capsule lets hull points near its caps stick out: with hull contacts the bound is the sphere that holds them all)
    ln.L(o + SC_BOUND) = (hull_mode && type == DG_SHAPE_POINTS) ? fmaxf(w.prm0 + w.prm1, w.prm2) : w.prm0 + (type == DG_SHAPE_SPHERE ? 0.f : w.prm1);
  }
  // broad phase: a group = all pairs between one moving body and one shape of the static world (or another
  // moving body); skipped as a whole when the bounding spheres are apart in every lane of the wave
  int cached_body = -1; V3 cpos = v3(0.f, 0.f, 0.f);
  const int lane = threadIdx.x & 63;
  // Group descriptors TBL at a time in lane tables (one vector load per table, v_readlane per group): with one
  // wavefront per SIMD a chain of dependent scalar loads per group -- group -> shape -> parameters -- was a third of a
  // maze step's narrow phase (120 groups, almost all of them culled).
  float gx = 0.f, gy = 0.f, gz = 0.f, gr = -1.f; int gfirst = 0, gcount = 0, gba = 0;
  auto rlf = [](float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); };
  for (int g = 0; g < sc.ngroups; g++) {
    cip gi = sc.GI + g * DG_GI_STRIDE;
    int g_first, g_count, ba; float ox, oy, oz, orr;
    if constexpr (TBL > 0) {
      if ((g & (TBL - 1)) == 0) {  // (only the first TBL lanes are guaranteed active)
        const int gg = min(g + lane, sc.ngroups - 1); cfp gd = sc.GD + 4 * gg; cip gj = sc.GI + gg * DG_GI_STRIDE;
        gx = gd[0]; gy = gd[1]; gz = gd[2]; gr = gd[3]; gfirst = gj[DG_GI_FIRST]; gcount = gj[DG_GI_COUNT]; gba = gj[DG_GI_BODY_A];
      }
      const int l = g & (TBL - 1);
      g_first = __builtin_amdgcn_readlane(gfirst, l); g_count = __builtin_amdgcn_readlane(gcount, l); ba = __builtin_amdgcn_readlane(gba, l);
      ox = rlf(gx, l); oy = rlf(gy, l); oz = rlf(gz, l); orr = rlf(gr, l);
    } else { g_first = gi[DG_GI_FIRST]; g_count = gi[DG_GI_COUNT]; ba = gi[DG_GI_BODY_A]; cfp gd = sc.GD + 4 * g; ox = gd[0]; oy = gd[1]; oz = gd[2]; orr = gd[3]; }
    if (g_first >= pair_hi || g_first + g_count <= pair_lo) continue;
    if (ba != cached_body) { cpos = ln.base_pos(ba); cached_body = ba; }
    if (orr >= 0.f) {  // frozen static partner: everything needed is in the descriptor
      const V3 dc = cpos - v3(ox, oy, oz); if (!__any(dot(dc, dc) < orr * orr)) continue;
    } else {
      const int bb = gi[DG_GI_BODY_B], ss = gi[DG_GI_STATIC_SHAPE];
      V3 other; float reach = sc.BF[ba * DG_BF_STRIDE + DG_BF_BOUND] + margin;
      if (ss >= 0) {
        cip si = sc.SI + ss * DG_SI_STRIDE; cfp sf = sc.SF + ss * DG_SF_STRIDE; const int st = si[DG_SI_TYPE];
        reach += st == DG_SHAPE_SPHERE ? sf[DG_SF_PARAMS] : st == DG_SHAPE_BOX ? sqrtf(sf[DG_SF_PARAMS] * sf[DG_SF_PARAMS] + sf[DG_SF_PARAMS + 1] * sf[DG_SF_PARAMS + 1] + sf[DG_SF_PARAMS + 2] * sf[DG_SF_PARAMS + 2]) : sf[DG_SF_PARAMS] + sf[DG_SF_PARAMS + 1];
        if (si[DG_SI_FLAGS] & DG_SHAPE_WORLD) other = v3(sf[DG_SF_POS], sf[DG_SF_POS + 1], sf[DG_SF_POS + 2]);
        else { WShape w; shape_world(ln, ss, w); other = w.p; }
      } else { other = ln.base_pos(bb); reach += sc.BF[bb * DG_BF_STRIDE + DG_BF_BOUND]; }
      { const V3 dc = cpos - other; if (!__any(dot(dc, dc) < reach * reach)) continue; }
    }
  const int first = max(g_first, pair_lo), count = min(g_first + g_count, pair_hi) - first;  // this wave's share
  if constexpr (SLC > 1) {
    const int ss = gi[DG_GI_STATIC_SHAPE];
    if (orr >= 0.f && sc.SI[ss * DG_SI_STRIDE + DG_SI_TYPE] == DG_SHAPE_BOX) {
      WShape b; shape_world(ln, ss, b);  // (the same box for every pair of the group)
      for (int c0 = 0; c0 < count; c0 += SLC) {
        const bool act = c0 + sl < count; const int pi = first + (act ? c0 + sl : 0), d = sc.PD[pi];
        const int sa = d & 4095, ta = (d >> 24) & 3; const float flip = (d >> 28) & 1 ? -1.f : 1.f;
        const int oa = sc.tr_off + sa * SC_STRIDE;
        const V3 ca = ln.L3(oa + SC_C);
        // up to four contacts per pair, in fixed slots (sphere: 0; capsule: its two ends; hull: its four deepest corners)
        float hp[4][3], hn[4][3], hd[4]; unsigned hv = 0u; int hf[4] = {0, 1, 0, 0};  // (feature of slot j: capsule end j; a hull's vertex index, below)
        auto keep = [&](int j, const Hit& h, bool ok) {
          if (h.hit && ok) hv |= 1u << j;
          const V3 m = (h.pa + h.pb) * 0.5f, nn = h.n * flip;
          hp[j][0] = m.x; hp[j][1] = m.y; hp[j][2] = m.z; hn[j][0] = nn.x; hn[j][1] = nn.y; hn[j][2] = nn.z; hd[j] = h.dist;
        };
        if (act && sphere_box(ca, ln.L(oa + SC_BOUND), b, margin).hit) {  // bounding sphere of the round shape against the box
          const V3 ha = ln.L3(oa + SC_H); const V3 a0 = ca - ha, a1 = ca + ha; const float ra = ln.L(oa + SC_R);
          if (ta == DG_SHAPE_SPHERE) keep(0, sphere_box(a0, ra, b, margin), true);
          else if (ta == DG_SHAPE_CAPSULE) {
            keep(0, sphere_box(a0, ra, b, margin), true);
            if (sc.SF[sa * DG_SF_STRIDE + DG_SF_PARAMS + 1] > 0.f) keep(1, sphere_box(a1, ra, b, margin), true);
          } else if (ta == DG_SHAPE_POINTS) {
            cip sd = sc.SD + 4 * sa; const int rslot = sd[0], soff = sd[1], poff = sd[2], npts = sd[3];
            M3 Rl; V3 pl;
            if (rslot < 0) { M3 Id = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}; Rl = Id; pl = v3(0.f, 0.f, 0.f); }
            else { Rl = ln.LR(rslot); pl = soff >= 0 ? v3(ln.S(soff), ln.S(soff + 1), ln.S(soff + 2)) : ln.L3(rslot + 6); }
            int bi4[4] = {-1, -1, -1, -1}; float bd4[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
            for (int k2 = 0; k2 < npts; k2++) {
              cfp pp = sc.PF + 3 * (poff + k2);
              Hit h = sphere_box(pl + mul(Rl, v3(pp[0], pp[1], pp[2])), 0.f, b, margin);
              if (!h.hit) continue;
              bool placed = false;
#pragma unroll
              for (int j = 0; j < 4; j++) {
                if (!placed && h.dist < bd4[j]) {
#pragma unroll
                  for (int m = 3; m > j; m--) { bd4[m] = bd4[m - 1]; bi4[m] = bi4[m - 1]; }
                  bd4[j] = h.dist; bi4[j] = k2; placed = true;
                }
              }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
              cfp pp = sc.PF + 3 * (poff + (bi4[j] < 0 ? 0 : bi4[j])); hf[j] = bi4[j] < 0 ? 0 : bi4[j];
              keep(j, sphere_box(pl + mul(Rl, v3(pp[0], pp[1], pp[2])), 0.f, b, margin), bi4[j] >= 0);
            }
          }
        }
        if (!__any(hv != 0u)) continue;
        // append in pair order: slice j's contacts before slice j + 1's; every lane of the env keeps the same count
        for (int j = 0; j < SLC; j++) {
          const unsigned fl = (unsigned)__shfl((int)hv, (int)((threadIdx.x & 63 & ~(SLC - 1)) | j));
          if (!__any(fl != 0u)) continue;
#pragma unroll
          for (int t = 0; t < 4; t++) {
            if (((fl >> t) & 1u) && cnt < sc.max_contacts) {
              if (sl == j) {
                const int o = list + 1 + cnt * CL_STRIDE;
                ln.L(o + CL_PAIR) = (float)pi; ln.L(o + CL_KEY) = (float)DG_CONTACT_KEY(pi, hf[t]); ln.L3set(o + CL_P, v3(hp[t][0], hp[t][1], hp[t][2])); ln.L3set(o + CL_N, v3(hn[t][0], hn[t][1], hn[t][2])); ln.L(o + CL_DIST) = hd[t];
              }
              cnt++;
            }
          }
        }
      }
      continue;
    }
  }
  constexpr int CH = TBL > 0 ? TBL : 1;
  for (int c0 = 0; c0 < count; c0 += CH) {
    const int n = min(CH, count - c0);
    int mydesc = 0;
    if constexpr (TBL > 0) { if (lane < n) mydesc = sc.PD[first + c0 + lane]; }
    auto desc_of = [&](int k) -> int { if constexpr (TBL > 0) return __builtin_amdgcn_readlane(mydesc, k); else return sc.PD[first + c0 + k]; };
    // the bounding data of pair k + 1 is read from LDS while pair k is tested
    struct Cull { V3 ca, cb; float reach; int desc; };
    auto fetch = [&](int k) { Cull c; c.desc = desc_of(k); const int oa = sc.tr_off + (c.desc & 4095) * SC_STRIDE, ob = sc.tr_off + ((c.desc >> 12) & 4095) * SC_STRIDE;
      c.ca = ln.L3(oa + SC_C); c.cb = ln.L3(ob + SC_C); c.reach = ln.L(oa + SC_BOUND) + ln.L(ob + SC_BOUND) + margin; return c; };
    Cull nx = fetch(0);
  for (int k = 0; k < n; k++) {
    const Cull cu = nx; if (k + 1 < n) nx = fetch(k + 1);
    const int pi = first + c0 + k, d = cu.desc;
    const int sa = d & 4095, sb = (d >> 12) & 4095, ta = (d >> 24) & 3, tb = (d >> 26) & 3; const float flip = (d >> 28) & 1 ? -1.f : 1.f;
    const int oa = sc.tr_off + sa * SC_STRIDE, ob = sc.tr_off + sb * SC_STRIDE;
#ifndef DG_NO_HULL_CODE
    if (hull_mode && ta == DG_SHAPE_POINTS && tb == DG_SHAPE_POINTS) {
      // hull against hull (DG_HF_HULL_CONTACTS, dg_hull.h): the lanes whose bounding spheres reach each other run GJK on the two
      // point sets in their link frames; one contact per pair
      const float reach = cu.reach + 2.f * hmg; const V3 dc = cu.ca - cu.cb; const bool near = dot(dc, dc) < reach * reach;
      if (!__any(near)) continue;
      cip da = sc.SD + 4 * sa, db = sc.SD + 4 * sb;
      auto frame_of = [&](cip sd, M3& Rl, V3& pl) {
        if (sd[0] < 0) { M3 Id = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}; Rl = Id; pl = v3(0.f, 0.f, 0.f); }
        else { Rl = ln.LR(sd[0]); pl = sd[1] >= 0 ? v3(ln.S(sd[1]), ln.S(sd[1] + 1), ln.S(sd[1] + 2)) : ln.L3(sd[0] + 6); }
      };
      HullPairD hp; V3 pla, plb; frame_of(da, hp.RA, pla); frame_of(db, hp.RB, plb);
      // second cull: the capsules that CONTAIN the hulls (the fitted axis and radius, half length DG_SF_HULL_HALF) -- their distance
      // is a lower bound of the hulls'.  Two arms working next to each other pass the sphere test all the time and this one rarely.
      bool close = near;
      { cfp fa = sc.SF + sa * DG_SF_STRIDE, fb = sc.SF + sb * DG_SF_STRIDE;
        M3 Rsa, Rsb; _Pragma("unroll") for (int q = 0; q < 9; q++) { Rsa.m[q] = fa[DG_SF_ROT + q]; Rsb.m[q] = fb[DG_SF_ROT + q]; }
        const V3 axa = mul(hp.RA, v3(Rsa.m[2], Rsa.m[5], Rsa.m[8])) * fa[DG_SF_HULL_HALF], axb = mul(hp.RB, v3(Rsb.m[2], Rsb.m[5], Rsb.m[8])) * fb[DG_SF_HULL_HALF];
        V3 qa, qb; seg_seg(cu.ca - axa, cu.ca + axa, cu.cb - axb, cu.cb + axb, qa, qb);
        const float lim = ln.L(oa + SC_R) + ln.L(ob + SC_R) + margin + 2.f * hmg; const V3 dq = qa - qb;
        close = near && dot(dq, dq) < lim * lim; }
      if (!__any(close)) continue;
      hp.pa = sc.PF + 3 * da[2]; hp.na = da[3]; hp.pb = sc.PF + 3 * db[2]; hp.nb = db[3]; hp.tBA = plb - pla;
      hp.ew = hull_ws_of(sc.hull_ws);
      hull_tables(hp, TBL > 0 && hp.na <= (TBL > 0 ? TBL : 1) && hp.nb <= (TBL > 0 ? TBL : 1));
      HullHit hh; hull_hull(hp, dc, margin + 2.f * hmg, close, hh);
      Hit h; h.hit = hh.hit && hh.dist - 2.f * hmg < margin; h.n = hh.n; h.dist = hh.dist - 2.f * hmg;
      h.pa = (hh.pa + pla) - hh.n * hmg; h.pb = (hh.pb + pla) + hh.n * hmg;
      emit_contact(ln, list, cnt, pi, h, flip);
    } else
#endif
    if (tb != DG_SHAPE_BOX) {
      // round vs round: closest points of the two segments, then sphere-sphere
      { const V3 dc = cu.ca - cu.cb; if (!__any(dot(dc, dc) < cu.reach * cu.reach)) continue; }
      const V3 ha = ln.L3(oa + SC_H), hb = ln.L3(ob + SC_H); const float ra = ln.L(oa + SC_R), rb = ln.L(ob + SC_R);
      const V3 a0 = cu.ca - ha, a1 = cu.ca + ha, b0 = cu.cb - hb, b1 = cu.cb + hb;
      V3 ca = a0, cb = b0;
      if (ta == DG_SHAPE_SPHERE && tb != DG_SHAPE_SPHERE) cb = closest_on_seg(b0, b1, a0);
      else if (ta != DG_SHAPE_SPHERE) seg_seg(a0, a1, b0, b1, ca, cb);
      emit_contact(ln, list, cnt, pi, sphere_sphere(ca, ra, cb, rb, margin), flip);
    } else {
      WShape b; shape_world(ln, sb, b);
      const V3 ha = ln.L3(oa + SC_H); const V3 a0 = cu.ca - ha, a1 = cu.ca + ha; const float ra = ln.L(oa + SC_R);
      // cull with the bounding sphere of the round shape against the box
      { Hit hb = sphere_box(cu.ca, ln.L(oa + SC_BOUND), b, margin); if (!__any(hb.hit)) continue; }
      if (ta == DG_SHAPE_SPHERE) emit_contact(ln, list, cnt, pi, sphere_box(a0, ra, b, margin), flip);
      else if (ta == DG_SHAPE_CAPSULE) {
        emit_contact(ln, list, cnt, pi, sphere_box(a0, ra, b, margin), flip);
        if (sc.SF[sa * DG_SF_STRIDE + DG_SF_PARAMS + 1] > 0.f) emit_contact(ln, list, cnt, pi, sphere_box(a1, ra, b, margin), flip, 1);
      } else if (ta == DG_SHAPE_POINTS) {
        const int abody = sc.SI[sa * DG_SI_STRIDE + DG_SI_BODY], alink = sc.SI[sa * DG_SI_STRIDE + DG_SI_LINK];
        const int poff = sc.SI[sa * DG_SI_STRIDE + DG_SI_POINT_OFF], npts = sc.SI[sa * DG_SI_STRIDE + DG_SI_N_POINTS];
        int bi4[4] = {-1, -1, -1, -1}; float bd4[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
        M3 Rl; V3 pl;
        if (sc.SI[sa * DG_SI_STRIDE + DG_SI_FLAGS] & DG_SHAPE_WORLD) { M3 Id = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}; Rl = Id; pl = v3(0.f, 0.f, 0.f); }
        else ln.link_world(abody, alink, Rl, pl);
        for (int k2 = 0; k2 < npts; k2++) {
          cfp pp = sc.PF + 3 * (poff + k2);
          Hit h = sphere_box(pl + mul(Rl, v3(pp[0], pp[1], pp[2])), 0.f, b, margin);
          if (!h.hit) continue;
          bool placed = false;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (!placed && h.dist < bd4[j]) {
#pragma unroll
              for (int m = 3; m > j; m--) { bd4[m] = bd4[m - 1]; bi4[m] = bi4[m - 1]; }
              bd4[j] = h.dist; bi4[j] = k2; placed = true;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          int k2 = bi4[j] < 0 ? 0 : bi4[j];
          cfp pp = sc.PF + 3 * (poff + k2);  // per-lane index: vector load
          Hit h = sphere_box(pl + mul(Rl, v3(pp[0], pp[1], pp[2])), 0.f, b, margin);
          h.hit = h.hit && bi4[j] >= 0;
          emit_contact(ln, list, cnt, pi, h, flip, k2);
        }
      }
    }
  }
  }
  }
  ln.L(list) = (float)cnt;
  return cnt;
}

// ------------------------------------------------------------------- rows
// contact row r (3 per contact: normal, t1, t2) in the transient region:
//   [JA nvmax][RA nvmax][JB nvmax][RB nvmax][b][acc][diag]     (dv offsets / lengths live in the contact list)
DGD int crow_stride(int tail) { return tail + 3; }

DGD void tangent_basis(V3 n, V3& t1, V3& t2) {
  if (fabsf(n.z) > 0.70710678118654752f) { float a = n.y * n.y + n.z * n.z, k = 1.0f / sqrtf(a); t1 = v3(0.f, -n.z * k, n.y * k); t2 = v3(a * k, -n.x * t1.z, n.x * t1.y); }
  else { float a = n.x * n.x + n.y * n.y, k = 1.0f / sqrtf(a); t1 = v3(-n.y * k, n.x * k, 0.f); t2 = v3(-n.z * t1.y, n.z * t1.x, a * k); }
}

// ---- warm starting ---------------------------------------------------------------------------------------------------
// The contact impulse cache of an env (state, DG_WS_*): [count][key normal t1 t2] x max_contacts, written at the end of
// every substep (store_warm_cache) and cleared by a reset.  A row of a contact whose key was there in the previous substep
// starts from DG_HF_WARMSTART (normal) / DG_HF_WARMSTART_FRICTION (tangents) x the impulse it ended with; the sweeps add
// the velocity change those starting impulses amount to before their first iteration.
// Index of the cached entry with this key, -1 if there is none (or no cache).  Called under divergence: the loop bound
// is uniform over the ACTIVE lanes, the key loads are independent of each other.
// (The cache lives in the state, i.e. in global memory, and a lone wavefront pays a whole round trip per DEPENDENT load: the keys
// are read eight at a time with the count, their loads issued together, and a contact's three impulses together -- two round trips
// per contact instead of count + 4.  With one load per entry behind an early exit this search was ~40 % of the row construction of
// every scene with resting contacts: marbles 105 k of 335 k cycles per step in `rows`, from_the_readme (25 contacts: 25 x 25
// dependent loads per substep) 421 k of 2.36 M.)
template <int LANES>
DGD int warm_find(const Lane<LANES>& ln, float key) {
  const DevScene& sc = ln.sc; if (sc.warm_off < 0) return -1;
  constexpr int CH = 8;
  int found = -1, np = 0;
  for (int j0 = 0; j0 < sc.max_contacts; j0 += CH) {
    if (j0 > 0 && !__any(j0 < np)) break;
    float kj[CH]; const float npf = ln.S(sc.warm_off);
#pragma unroll
    for (int t = 0; t < CH; t++) kj[t] = ln.S(sc.warm_off + 1 + min(j0 + t, sc.max_contacts - 1) * DG_WS_STRIDE + DG_WS_KEY);
    np = (int)npf;
#pragma unroll
    for (int t = 0; t < CH; t++) if (j0 + t < np && kj[t] == key && found < 0) found = j0 + t;
  }
  return found;
}
// the starting impulses of a contact's three rows (normal, t1, t2): factor x what the cached entry ended with, zero without one
template <int LANES>
DGD void warm_impulses(const Lane<LANES>& ln, int found, float (&out)[3]) {
  const DevScene& sc = ln.sc; out[0] = out[1] = out[2] = 0.f;
  if (sc.warm_off < 0) return;
  const int e = sc.warm_off + 1 + max(found, 0) * DG_WS_STRIDE + DG_WS_NORMAL;
  const float v0 = ln.S(e), v1 = ln.S(e + 1), v2 = ln.S(e + 2), fn = sc.HF[DG_HF_WARMSTART], ft = sc.HF[DG_HF_WARMSTART_FRICTION];
  out[0] = (found >= 0 && fn > 0.f) ? fn * v0 : 0.f; out[1] = (found >= 0 && ft > 0.f) ? ft * v1 : 0.f; out[2] = (found >= 0 && ft > 0.f) ? ft * v2 : 0.f;
}
// this substep's contacts and the impulses their rows ended with (every sweep form leaves them in the rows' `acc` slots)
template <int LANES>
DGD void store_warm_cache(const Lane<LANES>& ln, int ncont, int wave_max_cont) {
  const DevScene& sc = ln.sc; if (sc.warm_off < 0) return;
  const int rs = sc.crow_tail + 3;
  ln.Sset(sc.warm_off, (float)ncont);
  for (int c = 0; c < wave_max_cont; c++) {
    if (c >= ncont) continue;
    const int e = sc.warm_off + 1 + c * DG_WS_STRIDE, ro = sc.tr_off + 3 * c * rs + sc.crow_tail + 1;
    const float key = ln.L(sc.cont_off + 1 + c * CL_STRIDE + CL_KEY), i0 = ln.L(ro), i1 = ln.L(ro + rs), i2 = ln.L(ro + 2 * rs);
    ln.Sset(e + DG_WS_KEY, key); ln.Sset(e + DG_WS_NORMAL, i0); ln.Sset(e + DG_WS_T1, i1); ln.Sset(e + DG_WS_T2, i2);
  }
}

// builds the three rows of contact slot c for the lanes whose contact belongs to (uniform) pair `pair`
template <int LANES>
DGD void build_contact_rows(const Lane<LANES>& ln, int c, int pair, bool mine, bool vel_in_lds, int d_lo = 0, int d_hi = 3) {
  const DevScene& sc = ln.sc; const int nvm = sc.nv_max, tl = sc.crow_tail, rs = crow_stride(tl);
  cip sa = sc.SI + sc.PI[pair * DG_PI_STRIDE + DG_PI_A] * DG_SI_STRIDE; cip sb = sc.SI + sc.PI[pair * DG_PI_STRIDE + DG_PI_B] * DG_SI_STRIDE;
  const int ba = sa[DG_SI_BODY], la = sa[DG_SI_LINK], bb = sb[DG_SI_BODY], lb = sb[DG_SI_LINK];
  const bool a_dyn = !(ln.fixed(ba) && ln.bi(ba)[DG_BI_N_LINKS] == 0), b_dyn = !(ln.fixed(bb) && ln.bi(bb)[DG_BI_N_LINKS] == 0);
  if (!mine) return;
  const int co = sc.cont_off + 1 + c * CL_STRIDE;
  V3 p = ln.L3(co + CL_P), n = ln.L3(co + CL_N); float dist = ln.L(co + CL_DIST);
  V3 t1, t2; tangent_basis(n, t1, t2);
  const float h = sc.h, cerp = sc.HF[DG_HF_CONTACT_ERP], slop = sc.HF[DG_HF_LINEAR_SLOP];
  ln.L(co + CL_MU) = sc.SF[sc.PI[pair * DG_PI_STRIDE + DG_PI_A] * DG_SF_STRIDE + DG_SF_FRICTION] * sc.SF[sc.PI[pair * DG_PI_STRIDE + DG_PI_B] * DG_SF_STRIDE + DG_SF_FRICTION];
  const int wfound = warm_find(ln, ln.L(co + CL_KEY)); float wimp[3]; warm_impulses(ln, wfound, wimp);
  for (int d = d_lo; d < d_hi; d++) {  // (lane-sliced callers give each lane of an env's group one direction)
    V3 dir = d == 0 ? n : (d == 1 ? t1 : t2);
    int ro = sc.tr_off + (3 * c + d) * rs; float diag = 0.f, jv = 0.f;
    for (int k = 0; k < tl; k++) ln.L(ro + k) = 0.f;  // rows are swept branch-free: pad with zeros
    // first side = the dynamic one of (A, B); the oracle makes the same choice
    int b1 = a_dyn ? ba : bb, l1 = a_dyn ? la : lb; V3 d1 = a_dyn ? dir : -dir;
    // dense rows (total DoF <= 32): Jacobian and response are indexed by GLOBAL DoF, [J nt][R nt]; otherwise
    // per-body blocks [JA nv_max][RA nv_max]([JB][RB])
    const int g1 = sc.dense ? ln.plb(b1)[PLB_DV] - sc.dv_base : 0, g2 = sc.dense ? ln.plb(bb)[PLB_DV] - sc.dv_base : 0;
    const int j1 = ro + g1, r1 = ro + (sc.dense ? sc.nt : nvm) + g1;
    diag += ln.point_row(b1, l1, p, d1, j1, r1); jv += vel_in_lds ? ln.gen_vel_dot_lds(j1, ln.plb(b1)[PLB_DV], ln.plb(b1)[PLB_NV]) : ln.gen_vel_dot(b1, j1);
    ln.L(co + CL_DVA) = (float)ln.plb(b1)[PLB_DV]; ln.L(co + CL_NVA) = (float)ln.plb(b1)[PLB_NV];
    if (a_dyn && b_dyn) {
      const int j2 = sc.dense ? ro + g2 : ro + 2 * nvm, r2 = sc.dense ? ro + sc.nt + g2 : ro + 3 * nvm;
      diag += ln.point_row(bb, lb, p, -dir, j2, r2); jv += vel_in_lds ? ln.gen_vel_dot_lds(j2, ln.plb(bb)[PLB_DV], ln.plb(bb)[PLB_NV]) : ln.gen_vel_dot(bb, j2);
      ln.L(co + CL_DVB) = (float)ln.plb(bb)[PLB_DV]; ln.L(co + CL_NVB) = (float)ln.plb(bb)[PLB_NV];
    } else { ln.L(co + CL_DVB) = 0.f; ln.L(co + CL_NVB) = 0.f; }
    float b = -jv;
    if (d == 0) { float pen = dist + slop; b += pen > 0.f ? -pen / h : -pen * cerp / h; }
    ln.L(ro + tl) = b; ln.L(ro + tl + 1) = diag > 1e-18f ? (d == 0 ? wimp[0] : d == 1 ? wimp[1] : wimp[2]) : 0.f; ln.L(ro + tl + 2) = diag;
  }
}

// ---- fixed constraints (DG_KI_* / DG_KF_*; reference model.py:74-75 createConstraint(JOINT_FIXED)) as solver rows ------
// Constraint q keeps two pseudo contact slots behind the real ones -- max_contacts + 2 q (linear rows x y z) and + 1 (angular
// rows) -- so that solve_crow sweeps its rows like any two-sided contact row.  Scenes with constraints take the generic
// sweeps only (dg_world_create: not dense, no helper wavefronts, rows two-sided).  Every table read is wave-uniform.
template <int LANES>
DGD void build_constraint_rows(const Lane<LANES>& ln) {
  const DevScene& sc = ln.sc; const int nvm = sc.nv_max, tl = sc.crow_tail, rs = crow_stride(tl); const float h = sc.h, erp = sc.HF[DG_HF_CONTACT_ERP];
  for (int q = 0; q < sc.ncons; q++) {
    cip ki = sc.KI + q * DG_KI_STRIDE; cfp kf = sc.KF + q * DG_KF_STRIDE;
    const int ba = ki[DG_KI_BODY_A], la = ki[DG_KI_LINK_A], bb = ki[DG_KI_BODY_B], lb = ki[DG_KI_LINK_B];
    const bool a_dyn = !(ln.fixed(ba) && ln.bi(ba)[DG_BI_N_LINKS] == 0), b_dyn = !(ln.fixed(bb) && ln.bi(bb)[DG_BI_N_LINKS] == 0);
    V3 P[2]; Q4 Q[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int b = k == 0 ? ba : bb, gl = k == 0 ? la : lb; M3 R; V3 o; ln.link_world(b, gl, R, o);
      cfp pp = kf + (k == 0 ? DG_KF_POS_A : DG_KF_POS_B); cfp qq = kf + (k == 0 ? DG_KF_QUAT_A : DG_KF_QUAT_B);
      P[k] = o + mul(R, v3(pp[0], pp[1], pp[2]));
      const Q4 ql = gl < 0 ? ln.base_quat(b) : qfrom_mat(R); const Q4 qo = {qq[0], qq[1], qq[2], qq[3]}; Q[k] = qnormalize(qmul(ql, qo));
    }
    const V3 perr = P[1] - P[0];
    const Q4 qe = qmul(Q[1], qconj(Q[0])); const float sg = qe.w < 0.f ? -2.f : 2.f; const V3 aerr = v3(sg * qe.x, sg * qe.y, sg * qe.z);
    const int b1 = a_dyn ? ba : bb, l1 = a_dyn ? la : lb;
#pragma unroll 1
    for (int d = 0; d < 6; d++) {
      const bool tq = d >= 3; const int c = sc.max_contacts + 2 * q + (tq ? 1 : 0), co = sc.cont_off + 1 + c * CL_STRIDE, ro = sc.tr_off + (3 * c + d % 3) * rs;
      const V3 dir = v3(d % 3 == 0 ? 1.f : 0.f, d % 3 == 1 ? 1.f : 0.f, d % 3 == 2 ? 1.f : 0.f);
      for (int k = 0; k < tl; k++) ln.L(ro + k) = 0.f;
      float diag = 0.f, jv = 0.f;
      if (a_dyn || b_dyn) {
        const V3 d1 = a_dyn ? dir : -dir, p1 = a_dyn ? P[0] : P[1];
        diag += tq ? ln.template point_row<true>(b1, l1, p1, d1, ro, ro + nvm) : ln.point_row(b1, l1, p1, d1, ro, ro + nvm);
        jv += ln.gen_vel_dot(b1, ro);
        ln.L(co + CL_DVA) = (float)ln.plb(b1)[PLB_DV]; ln.L(co + CL_NVA) = (float)ln.plb(b1)[PLB_NV];
        if (a_dyn && b_dyn) {
          diag += tq ? ln.template point_row<true>(bb, lb, P[1], -dir, ro + 2 * nvm, ro + 3 * nvm) : ln.point_row(bb, lb, P[1], -dir, ro + 2 * nvm, ro + 3 * nvm);
          jv += ln.gen_vel_dot(bb, ro + 2 * nvm);
          ln.L(co + CL_DVB) = (float)ln.plb(bb)[PLB_DV]; ln.L(co + CL_NVB) = (float)ln.plb(bb)[PLB_NV];
        } else { ln.L(co + CL_DVB) = 0.f; ln.L(co + CL_NVB) = 0.f; }
      }
      ln.L(ro + tl) = -jv + erp * dot(tq ? aerr : perr, dir) / h; ln.L(ro + tl + 1) = 0.f; ln.L(ro + tl + 2) = diag;
    }
  }
}

// The same rows for the lanes whose contact joins two BASE shapes (no link on either side: marbles, a drone on the
// ground, a free body on a table) in an all-dense scene, without the pair-by-pair serialisation: every table entry is
// fetched per lane, so ONE pass serves all lanes whatever pairs they hold.  Returns whether this lane's contact was
// such a contact (the others go through build_contact_rows, pair by pair).  Same arithmetic per row as point_row.
template <int LANES>
DGD bool build_contact_rows_base(const Lane<LANES>& ln, int c, bool has, bool vel_in_lds, int d_lo = 0, int d_hi = 3) {
  const DevScene& sc = ln.sc;
  if (!sc.dense) return false;
  const int tl = sc.crow_tail, rs = crow_stride(tl), nt = sc.nt;
  const int cc = has ? c : 0, co = sc.cont_off + 1 + cc * CL_STRIDE;
  const int pair = has ? (int)ln.L(co + CL_PAIR) : 0;
  const int pa = sc.PI[pair * DG_PI_STRIDE + DG_PI_A], pb = sc.PI[pair * DG_PI_STRIDE + DG_PI_B];  // (per-lane indices: vector loads)
  const int ba = sc.SI[pa * DG_SI_STRIDE + DG_SI_BODY], la = sc.SI[pa * DG_SI_STRIDE + DG_SI_LINK], bb = sc.SI[pb * DG_SI_STRIDE + DG_SI_BODY], lb = sc.SI[pb * DG_SI_STRIDE + DG_SI_LINK];
  const bool mine = has && la < 0 && lb < 0;
  if (!__any(mine)) return false;
  if (mine) {
    const int fa = sc.BI[ba * DG_BI_STRIDE + DG_BI_FLAGS], fb = sc.BI[bb * DG_BI_STRIDE + DG_BI_FLAGS];
    const bool a_dyn = !((fa & DG_BODY_FIXED) && sc.BI[ba * DG_BI_STRIDE + DG_BI_N_LINKS] == 0), b_dyn = !((fb & DG_BODY_FIXED) && sc.BI[bb * DG_BI_STRIDE + DG_BI_N_LINKS] == 0);
    const V3 p = ln.L3(co + CL_P), n = ln.L3(co + CL_N); const float dist = ln.L(co + CL_DIST);
    V3 t1, t2; tangent_basis(n, t1, t2);
    const float h = sc.h, cerp = sc.HF[DG_HF_CONTACT_ERP], slop = sc.HF[DG_HF_LINEAR_SLOP];
    ln.L(co + CL_MU) = sc.SF[pa * DG_SF_STRIDE + DG_SF_FRICTION] * sc.SF[pb * DG_SF_STRIDE + DG_SF_FRICTION];
    const int wfound = warm_find(ln, ln.L(co + CL_KEY)); float wimp[3]; warm_impulses(ln, wfound, wimp);
    // one side: Jacobian of the body's base coordinates, response through the first six rows of its M^-1, J . v
    auto side = [&](int b, int flags, V3 d, int ro, float& diag, float& jv) {
      cip P = sc.PLB + b * PLB_STRIDE; const int nv = P[PLB_NV], mo = P[PLB_MINV], dvo = P[PLB_DV], g = dvo - sc.dv_base;
      if (flags & DG_BODY_FIXED) return;  // a base that does not move: no entries (the row stays zero there)
      const int so = sc.BI[b * DG_BI_STRIDE + DG_BI_STATE_OFF];
      const M3 R0 = ln.LR(P[PLB_R0]); const V3 pos = v3(ln.S(so), ln.S(so + 1), ln.S(so + 2));
      const V3 ja = tmul(R0, cross(p - pos, d)), jl = tmul(R0, d);
      const float J[6] = {ja.x, ja.y, ja.z, jl.x, jl.y, jl.z};
      ln.L3set(ro + g, ja); ln.L3set(ro + g + 3, jl);
      for (int k = 0; k < nv; k++) {
        float r = 0.f;
#pragma unroll
        for (int j = 0; j < 6; j++) r += ln.L(mo + j * nv + k) * J[j];
        ln.L(ro + nt + g + k) = r;
        if (k < 6) diag += r * J[k];
      }
      if (vel_in_lds) {
#pragma unroll
        for (int k = 0; k < 6; k++) jv += J[k] * ln.L(dvo + k);
      } else {
        const V3 wb = tmul(R0, v3(ln.S(so + DG_BS_ANGVEL), ln.S(so + DG_BS_ANGVEL + 1), ln.S(so + DG_BS_ANGVEL + 2)));
        const V3 vb = tmul(R0, v3(ln.S(so + DG_BS_LINVEL), ln.S(so + DG_BS_LINVEL + 1), ln.S(so + DG_BS_LINVEL + 2)));
        jv += J[0] * wb.x + J[1] * wb.y + J[2] * wb.z + J[3] * vb.x + J[4] * vb.y + J[5] * vb.z;
      }
    };
    const int b1 = a_dyn ? ba : bb, f1 = a_dyn ? fa : fb;
    for (int d = d_lo; d < d_hi; d++) {
      const V3 dir = d == 0 ? n : (d == 1 ? t1 : t2);
      const int ro = sc.tr_off + (3 * c + d) * rs; float diag = 0.f, jv = 0.f;
      for (int k = 0; k < tl; k++) ln.L(ro + k) = 0.f;
      side(b1, f1, a_dyn ? dir : -dir, ro, diag, jv);
      ln.L(co + CL_DVA) = (float)sc.PLB[b1 * PLB_STRIDE + PLB_DV]; ln.L(co + CL_NVA) = (float)sc.PLB[b1 * PLB_STRIDE + PLB_NV];
      if (a_dyn && b_dyn) {
        side(bb, fb, -dir, ro, diag, jv);
        ln.L(co + CL_DVB) = (float)sc.PLB[bb * PLB_STRIDE + PLB_DV]; ln.L(co + CL_NVB) = (float)sc.PLB[bb * PLB_STRIDE + PLB_NV];
      } else { ln.L(co + CL_DVB) = 0.f; ln.L(co + CL_NVB) = 0.f; }
      float b = -jv;
      if (d == 0) { const float pen = dist + slop; b += pen > 0.f ? -pen / h : -pen * cerp / h; }
      ln.L(ro + tl) = b; ln.L(ro + tl + 1) = diag > 1e-18f ? (d == 0 ? wimp[0] : d == 1 ? wimp[1] : wimp[2]) : 0.f; ln.L(ro + tl + 2) = diag;
    }
  }
  return mine;
}

// The same rows for the lanes whose contact joins shapes on two FIXED-BASE SERIAL CHAINS of at most six joints (two arms touching:
// the goal state of ur_high_5), in an all-dense scene, again without the pair-by-pair serialisation.  build_contact_rows keeps every
// table lookup wave-uniform by handling the lanes pair by pair; at 16 384 envs in as many different poses the 64 envs of a
// wavefront hold up to ~20 different (link, link) pairs per contact slot, and the row construction was 75 % of an in-contact step
// (1.69 M of 2.26 M cycles per wavefront, profiles/r4_touching_ik_16384_inkernel_stamps_before.txt).  Here the pair's links are
// per-lane values: the loops run over the six possible joints of a chain with a per-lane mask (joint k is above the contact's
// link), poses and M^-1 come from per-lane LDS slots, axes from per-lane table loads -- ONE pass per contact slot.  Same
// arithmetic per row as point_row (the Jacobian entry of a revolute joint is dir . (axis x (p - origin)): the cross product is
// shared by the three directions).  Returns whether this lane's contact was such a contact.
template <int LANES>
DGD bool build_contact_rows_chains(const Lane<LANES>& ln, int c, bool has) {
  const DevScene& sc = ln.sc;
  if (!sc.dense) return false;
  constexpr int W = envs_per_wave(LANES);
  const int tl = sc.crow_tail, rs = crow_stride(tl), nt = sc.nt;
  const int cc = has ? c : 0, co = sc.cont_off + 1 + cc * CL_STRIDE;
  const int pair = has ? (int)ln.L(co + CL_PAIR) : 0;
  const int pa = sc.PI[pair * DG_PI_STRIDE + DG_PI_A], pb = sc.PI[pair * DG_PI_STRIDE + DG_PI_B];  // (per-lane indices: vector loads)
  const int bod[2] = {sc.SI[pa * DG_SI_STRIDE + DG_SI_BODY], sc.SI[pb * DG_SI_STRIDE + DG_SI_BODY]};
  const int lnk[2] = {sc.SI[pa * DG_SI_STRIDE + DG_SI_LINK], sc.SI[pb * DG_SI_STRIDE + DG_SI_LINK]};
  const bool mine = has && sc.PLB[bod[0] * PLB_STRIDE + PLB_CHAIN] != 0 && sc.PLB[bod[1] * PLB_STRIDE + PLB_CHAIN] != 0;
  if (!__any(mine)) return false;
  if (mine) {
    const V3 p = ln.L3(co + CL_P), n = ln.L3(co + CL_N); const float dist = ln.L(co + CL_DIST);
    V3 t1, t2; tangent_basis(n, t1, t2);
    const float h = sc.h, cerp = sc.HF[DG_HF_CONTACT_ERP], slop = sc.HF[DG_HF_LINEAR_SLOP];
    ln.L(co + CL_MU) = sc.SF[pa * DG_SF_STRIDE + DG_SF_FRICTION] * sc.SF[pb * DG_SF_STRIDE + DG_SF_FRICTION];
    const int wfound = warm_find(ln, ln.L(co + CL_KEY)); float wimp[3]; warm_impulses(ln, wfound, wimp);
    const V3 dirs[3] = {n, t1, t2};
    float diag[3] = {0.f, 0.f, 0.f}, jv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 3; d++) { const int ro = sc.tr_off + (3 * c + d) * rs; for (int k = 0; k < tl; k++) ln.L(ro + k) = 0.f; }  // rows are swept branch-free: pad with zeros
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const int b = bod[side]; const float sg = side == 0 ? 1.f : -1.f;  // (both sides can move: the first is A, pushed along +dir)
      const int first = sc.BI[b * DG_BI_STRIDE + DG_BI_FIRST_LINK], nl = sc.BI[b * DG_BI_STRIDE + DG_BI_N_LINKS];
      const int mo = sc.PLB[b * PLB_STRIDE + PLB_MINV], g = sc.PLB[b * PLB_STRIDE + PLB_DV] - sc.dv_base, top = lnk[side] < 0 ? -1 : lnk[side] - first;
      V3 wk[6]; float qd[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {  // w_k: the Jacobian entry of joint k along a direction is dir . w_k (zero for a joint not above the contact's link)
        const bool on = k < nl && k <= top; const int gl = first + (on ? k : 0);
        const float* pz = ln.lds + (size_t)sc.PLL[gl * PLL_STRIDE + PLL_POSE] * W;  // per-lane POSE slot of the link
        const V3 c0 = v3(pz[0], pz[W], pz[2 * W]), c1 = v3(pz[3 * W], pz[4 * W], pz[5 * W]), c2 = cross(c0, c1), pk = v3(pz[6 * W], pz[7 * W], pz[8 * W]);
        cfp f = sc.LF + gl * DG_LF_STRIDE; const V3 ax = v3(f[DG_LF_AXIS], f[DG_LF_AXIS + 1], f[DG_LF_AXIS + 2]);
        const V3 axw = c0 * ax.x + c1 * ax.y + c2 * ax.z;  // R (axis): the columns of R are c0, c1, c0 x c1
        const V3 w = sc.LI[gl * DG_LI_STRIDE + DG_LI_TYPE] == 0 ? cross(axw, p - pk) : axw;
        wk[k] = on ? w * sg : v3(0.f, 0.f, 0.f);
        qd[k] = on ? ln.S(sc.LI[gl * DG_LI_STRIDE + DG_LI_STATE_OFF] + DG_LS_QD) : 0.f;
      }
      float Mi[36];  // the body's M^-1 (n x n, row-major from a per-lane slot), zero-padded to 6 x 6
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int q = 0; q < 6; q++) { const bool in = r < nl && q < nl; const float m = ln.lds[(size_t)(mo + (in ? r * nl + q : 0)) * W]; Mi[6 * r + q] = in ? m : 0.f; }
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int ro = sc.tr_off + (3 * c + d) * rs; float J[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { J[k] = dot(dirs[d], wk[k]); jv[d] += J[k] * qd[k]; }
#pragma unroll
        for (int q = 0; q < 6; q++) {
          float r = 0.f;
#pragma unroll
          for (int k = 0; k < 6; k++) r += Mi[6 * k + q] * J[k];  // (M^-1 is symmetric: column q = row q)
          diag[d] += r * J[q];
          if (q < nl) { ln.lds[(size_t)(ro + g + q) * W] = J[q]; ln.lds[(size_t)(ro + nt + g + q) * W] = r; }
        }
      }
      if (side == 0) { ln.L(co + CL_DVA) = (float)sc.PLB[b * PLB_STRIDE + PLB_DV]; ln.L(co + CL_NVA) = (float)sc.PLB[b * PLB_STRIDE + PLB_NV]; }
      else { ln.L(co + CL_DVB) = (float)sc.PLB[b * PLB_STRIDE + PLB_DV]; ln.L(co + CL_NVB) = (float)sc.PLB[b * PLB_STRIDE + PLB_NV]; }
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int ro = sc.tr_off + (3 * c + d) * rs; float b = -jv[d];
      if (d == 0) { const float pen = dist + slop; b += pen > 0.f ? -pen / h : -pen * cerp / h; }
      ln.L(ro + tl) = b; ln.L(ro + tl + 1) = diag[d] > 1e-18f ? wimp[d] : 0.f; ln.L(ro + tl + 2) = diag[d];
    }
  }
  return mine;
}

// ---- batched LDS vector helpers -------------------------------------------------------------------------
// A single wave pays a full LDS round trip for every dependent access, so vectors of run-time length n are moved in
// chunks of 8 independent accesses (reads past n stay inside the padded regions and are masked out).
template <int LANES>
DGD float lds_dot(const Lane<LANES>& ln, int a, int b, int n) {
  float s = 0.f;
  for (int k0 = 0; k0 < n; k0 += 8) {
    float x[8], y[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { x[j] = ln.L(a + k0 + j); y[j] = ln.L(b + k0 + j); }
#pragma unroll
    for (int j = 0; j < 8; j++) s += (k0 + j < n) ? x[j] * y[j] : 0.f;
  }
  return s;
}
template <int LANES>
DGD void lds_axpy(const Lane<LANES>& ln, int y, int x, float alpha, int n) {  // y[0..n) += alpha * x[0..n)
  for (int k0 = 0; k0 < n; k0 += 8) {
    float xv[8], yv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { xv[j] = ln.L(x + k0 + j); yv[j] = ln.L(y + k0 + j); }
#pragma unroll
    for (int j = 0; j < 8; j++) if (k0 + j < n) ln.L(y + k0 + j) = yv[j] + alpha * xv[j];
  }
}

// ---- starting impulses of the motor rows (DG_HF_MOTOR_GUESS) -----------------------------------------------------------
// Without the clamps the motor rows of one body are the linear system  A lambda = b,  A = M^-1 restricted to the motorised
// joints (symmetric positive definite), b = the rows' velocity targets.  The sweeps start from its solution clamped to the
// rows' impulse bounds instead of from zero: the same fixed point, reached in ~6 sweeps instead of ~35 for a
// position-controlled arm (the oracle does the same; Bullet starts from zero [R]).
// Register form, up to six joints: M row-major 6 x 6 (zero-padded), smax[i] = 0 for a joint without a motor.  acc = the
// starting impulses, dv += M acc.
// packed Cholesky (lower triangle, diagonal stored INVERTED as in chol6: multiply-only) of the unit-diagonal matrix whose
// off-diagonal entry (i, j) is off(i, j), with a pivot floor, and the solve P y = rhs
template <int N, class OFF>
DGD void chol_unit_solve(OFF off, const float* rhs, float* x) {
  float P[N * (N + 1) / 2], y[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float t = i == j ? 1.f : off(i, j);
#pragma unroll
      for (int k = 0; k < j; k++) t -= P[i * (i + 1) / 2 + k] * P[j * (j + 1) / 2 + k];
      P[i * (i + 1) / 2 + j] = i == j ? __frsqrt_rn(fmaxf(t, 1e-6f)) : t * P[j * (j + 1) / 2 + j];
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    float t = rhs[i];
#pragma unroll
    for (int k = 0; k < i; k++) t -= P[i * (i + 1) / 2 + k] * y[k];
    y[i] = t * P[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    float t = y[i];
#pragma unroll
    for (int k = i + 1; k < N; k++) t -= P[k * (k + 1) / 2 + i] * x[k];
    x[i] = t * P[i * (i + 1) / 2 + i];
  }
}
// DG_HF_LIMIT_GUESS (pinning): lb0 / la0 and lb1 / la1 are the right-hand sides and impulses of the joints' lower / upper limit
// rows (la < 0: the row is not active).  A joint whose motor target lies beyond an active limit row enters the system as ONE
// unknown -- the joint's total impulse, with the limit row's velocity as right-hand side -- and starts with the motor saturated
// into the limit and the limit row holding the balance (written to la0 / la1); if the motor alone is too weak to reach the limit
// velocity it is held at its bound like any row that leaves its bounds.  dv += M x (total impulse per joint).  Same steps as
// the CPU checker (its comment "Starting impulses of the motor rows").
template <int N>
DGD void chain_motor_guess_n(const float* M, const float* b, const float* smax, float* acc, float* dv, float ptol, const float* lb0, float* la0, const float* lb1, float* la1) {
  const bool pinning = ptol >= 0.f;  // ptol: how far beyond the limit row's velocity the motor target must lie (the sweeps' own early-out, limit_ptol()); < 0: off
  // (the system is scaled symmetrically to a unit diagonal first: wrist and shoulder joints differ by orders of magnitude
  // in M^-1 and this is an fp32 factorisation; a pivot is floored at 1e-6 of its diagonal)
  float rhs[N], x[N], sd[N], pin[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    sd[i] = smax[i] > 0.f ? __frsqrt_rn(fmaxf(M[i * N + i], 1e-30f)) : 0.f;
    const bool plo = pinning && smax[i] > 0.f && la0[i] >= 0.f && b[i] < lb0[i] - ptol, phi = pinning && smax[i] > 0.f && la1[i] >= 0.f && b[i] > -lb1[i] + ptol;
    pin[i] = plo ? -1.f : (phi ? 1.f : 0.f);
    rhs[i] = (plo ? lb0[i] : (phi ? -lb1[i] : b[i])) * sd[i];
  }
  // the scaled matrix (unit diagonal), strictly-lower triangle packed: every round of the active set below works on it
  float A[N * (N - 1) / 2 + 1];
#pragma unroll
  for (int i = 1; i < N; i++)
#pragma unroll
    for (int j = 0; j < i; j++) A[i * (i - 1) / 2 + j] = M[i * N + j] * sd[i] * sd[j];
  auto a_of = [&](int i, int j) { return i > j ? A[i * (i - 1) / 2 + j] : A[j * (j - 1) / 2 + i]; };  // (i != j)
  chol_unit_solve<N>([&](int i, int j) { return A[i * (i - 1) / 2 + j]; }, rhs, x);
  // Primal-dual active set, at most DG_MOTOR_GUESS_ROUNDS rounds: the rows beyond their bounds are held there and the others
  // solved again; the sets are then re-read from x + residual (the diagonal of the scaled system is 1) -- a held row whose
  // residual pulls it back inside is released, a free row that left its bounds is held -- until no lane's sets change.  The
  // fixed sets are the solution of the clamped system: the sweeps then only confirm it (one iteration for ur_high_5; a single
  // round left 1e-2 .. 1e-1 rad/s behind whenever a row saturated: 8 sweeps at the 90th percentile, 18-24 for the slowest env
  // of a wavefront).  Bounds in scaled units; a pinned joint's unknown (its total impulse) is bounded on one side only.
  float blo[N], bhi[N]; bool up[N], dn[N], any = false;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const float bs = smax[i] > 0.f ? smax[i] * frcp(sd[i]) : 3.0e38f;
    blo[i] = pin[i] > 0.f ? -3.0e38f : -bs; bhi[i] = pin[i] < 0.f ? 3.0e38f : bs;
    up[i] = x[i] > bhi[i]; dn[i] = x[i] < blo[i]; any = any || up[i] || dn[i];
  }
  if (__any(any)) {
#pragma unroll 1
    for (int round = 0; round < DG_MOTOR_GUESS_ROUNDS; round++) {
      float r2[N], x2[N], val[N]; bool held[N];
#pragma unroll
      for (int i = 0; i < N; i++) { held[i] = up[i] || dn[i]; val[i] = up[i] ? bhi[i] : (dn[i] ? blo[i] : 0.f); }
#pragma unroll
      for (int i = 0; i < N; i++) {
        float t = held[i] ? val[i] : rhs[i];
#pragma unroll
        for (int j = 0; j < N; j++) if (j != i) t -= (!held[i] && held[j]) ? a_of(i, j) * val[j] : 0.f;
        r2[i] = t;
      }
      chol_unit_solve<N>([&](int i, int j) { return (held[i] || held[j]) ? 0.f : A[i * (i - 1) / 2 + j]; }, r2, x2);
      bool changed = false;
#pragma unroll
      for (int i = 0; i < N; i++) x[i] = any ? x2[i] : x[i];   // (a lane without a held row keeps its first solution, bit for bit)
#pragma unroll
      for (int i = 0; i < N; i++) {
        float y = rhs[i];  // x_i + residual_i = rhs_i - sum_{j != i} A_ij x_j  (unit diagonal)
#pragma unroll
        for (int j = 0; j < N; j++) if (j != i) y -= a_of(i, j) * x[j];
        const bool nu = any && y > bhi[i], nd = any && y < blo[i]; changed = changed || nu != up[i] || nd != dn[i]; up[i] = nu; dn[i] = nd;
      }
      if (!__any(changed)) break;
    }
  }
  float tot[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    const float t = x[i] * sd[i], clamped = __builtin_amdgcn_fmed3f(t, -smax[i], smax[i]);
    const bool pinned = pin[i] != 0.f && !(pin[i] * t > smax[i]);  // (beyond the bound: the motor alone is too weak to reach the limit velocity -- an ordinary saturated row)
    const float lim = pinned ? fmaxf(smax[i] - pin[i] * t, 0.f) : 0.f;  // what the limit row holds (it pushes along -pin)
    acc[i] = pinned ? pin[i] * smax[i] : clamped; tot[i] = acc[i] - pin[i] * lim;
    if (pinning) { if (pinned && pin[i] < 0.f) la0[i] = lim; if (pinned && pin[i] > 0.f) la1[i] = lim; }
  }
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int c = 0; c < N; c++) dv[c] += M[i * N + c] * tot[i];
}
DGD void chain_motor_guess(const float* M, const float* b, const float* smax, float* acc, float* dv, float ptol, const float* lb0, float* la0, const float* lb1, float* la1) { chain_motor_guess_n<6>(M, b, smax, acc, dv, ptol, lb0, la0, lb1, la1); }
// DG_HF_LIMIT_GUESS as the tolerance chain_motor_guess_n takes: the sweeps' early-out on a row's velocity residual, or -1 (off)
DGD float limit_ptol(const DevScene& sc) { return sc.HF[DG_HF_LIMIT_GUESS] > 0.f ? sqrtf(sc.HF[DG_HF_RESIDUAL_THRESHOLD]) : -1.f; }
// LDS form for a body with at most N joints (fixed or floating base): reads the joint block of M^-1 and the rows'
// right-hand sides into registers, writes the starting impulses into the rows' MR_ACC slots (every sweep form picks them
// up there and adds the velocity change they amount to before its first iteration).
template <int LANES, int N>
DGD void motor_guess_small(const Lane<LANES>& ln, int b) {
  const DevScene& sc = ln.sc; const float h = sc.h;
  const int first = ln.bi(b)[DG_BI_FIRST_LINK], n = ln.bi(b)[DG_BI_N_LINKS], k0 = ln.fixed(b) ? 0 : 6;
  const int nv = ln.plb(b)[PLB_NV], mvo = ln.plb(b)[PLB_MINV], mo0 = ln.pll(first)[PLL_MROW];
  float M[N * N], bb[N], smax[N], acc[N], dv[N], lb0[N], la0[N], lb1[N], la1[N];
  const float ptol = limit_ptol(sc); const bool pinning = ptol >= 0.f;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const bool has = i < n; const int ic = has ? i : 0;
    const float maxf = ln.mt.v[3 * (first + ic) + 2]; smax[i] = has ? (maxf < 0.f ? -maxf : maxf * sc.hm) : 0.f;
    bb[i] = has ? ln.L(mo0 + ic * MR_STRIDE + MR_B) : 0.f; dv[i] = 0.f;
    lb0[i] = ln.L(mo0 + ic * MR_STRIDE + MR_LO_B); la0[i] = has ? ln.L(mo0 + ic * MR_STRIDE + MR_LO_ACC) : -1.f;
    lb1[i] = ln.L(mo0 + ic * MR_STRIDE + MR_HI_B); la1[i] = has ? ln.L(mo0 + ic * MR_STRIDE + MR_HI_ACC) : -1.f;
#pragma unroll
    for (int c = 0; c < N; c++) { const float m = ln.L(mvo + (k0 + ic) * nv + k0 + (c < n ? c : 0)); M[i * N + c] = (has && c < n) ? m : 0.f; }
  }
  chain_motor_guess_n<N>(M, bb, smax, acc, dv, ptol, lb0, la0, lb1, la1);
#pragma unroll
  for (int i = 0; i < N; i++) if (i < n) {
    ln.L(mo0 + i * MR_STRIDE + MR_ACC) = acc[i];
    if (pinning) { ln.L(mo0 + i * MR_STRIDE + MR_LO_ACC) = la0[i]; ln.L(mo0 + i * MR_STRIDE + MR_HI_ACC) = la1[i]; }  // (starting impulses of pinned limit rows; every sweep form adds their velocity change)
  }
}
// LDS form for a body with more than six joints: packed Cholesky of the motorised block in the (free) transient region --
// call it after the dynamics and before the contact rows are built there.  The motor table is uniform over the envs, so
// every loop bound and every slot index is wave-uniform.
template <int LANES>
DGD void motor_guess_lds(const Lane<LANES>& ln, int b) {
  const DevScene& sc = ln.sc; const float h = sc.h;
  const int first = ln.bi(b)[DG_BI_FIRST_LINK], n = ln.bi(b)[DG_BI_N_LINKS], k0 = ln.fixed(b) ? 0 : 6;
  const int nv = ln.plb(b)[PLB_NV], mvo = ln.plb(b)[PLB_MINV], mo0 = ln.pll(first)[PLL_MROW];
  uint32_t motors = 0u; int k = 0;
  for (int i = 0; i < n && i < 32; i++) { const float maxf = ln.mt.v[3 * (first + i) + 2]; if ((maxf < 0.f ? -maxf : maxf * sc.hm) > 0.f) { motors |= 1u << i; k++; } }
  if (k == 0) return;
  const bool all = k == n;  // (every joint has a motor: the usual case -- pybullet gives every joint one at load)
  auto nth = [&](int a) { if (all) return a; uint32_t m = motors; for (int t = 0; t < a; t++) m &= m - 1; return __ffs((int)m) - 1; };  // a-th motorised joint
  // workspace in the transient region: packed lower triangle (diagonal inverted, as chol6), then y / x, then the scaling
  const int A = sc.tr_off, Y = A + k * (k + 1) / 2 + 8, S = Y + k + 8;
  auto at = [&](int a, int c) { return A + a * (a + 1) / 2 + c; };
  // symmetric scaling to a unit diagonal (finger and shoulder joints differ by 1e5 in M^-1; this is an fp32 factorisation)
  for (int a = 0; a < k; a++) { const int ia = nth(a); ln.L(S + a) = __frsqrt_rn(fmaxf(ln.L(mvo + (k0 + ia) * nv + k0 + ia), 1e-30f)); }
  for (int a = 0; a < k; a++) { const int ia = nth(a); const float sa = ln.L(S + a); for (int c = 0; c < a; c++) ln.L(at(a, c)) = ln.L(mvo + (k0 + ia) * nv + k0 + nth(c)) * sa * ln.L(S + c); ln.L(at(a, a)) = 1.f; }
  for (int a = 0; a < k; a++) {
    for (int c = 0; c <= a; c++) {
      const float s = ln.L(at(a, c)) - lds_dot(ln, at(a, 0), at(c, 0), c);
      ln.L(at(a, c)) = a == c ? __frsqrt_rn(fmaxf(s, 1e-6f)) : s * ln.L(at(c, c));  // (pivot floored at 1e-6 of its diagonal)
    }
  }
  for (int a = 0; a < k; a++) ln.L(Y + a) = (ln.L(mo0 + nth(a) * MR_STRIDE + MR_B) * ln.L(S + a) - lds_dot(ln, at(a, 0), Y, a)) * ln.L(at(a, a));
  for (int a = k - 1; a >= 0; a--) {
    float s = ln.L(Y + a);
    for (int t = a + 1; t < k; t++) s -= ln.L(at(t, a)) * ln.L(Y + t);
    ln.L(Y + a) = s * ln.L(at(a, a));
  }
  bool fits = true;  // (a body of this size whose solution does not fit its bounds starts from zero, as without the guess)
  for (int a = 0; a < k; a++) {
    const int i = nth(a); const float maxf = ln.mt.v[3 * (first + i) + 2], lim = maxf < 0.f ? -maxf : maxf * sc.hm;
    fits = fits && fabsf(ln.L(Y + a) * ln.L(S + a)) <= lim;
  }
  for (int a = 0; a < k; a++) ln.L(mo0 + nth(a) * MR_STRIDE + MR_ACC) = fits ? ln.L(Y + a) * ln.L(S + a) : 0.f;
}
template <int LANES>
DGD void motor_guess(const Lane<LANES>& ln, int b) {
  if (!(ln.sc.HF[DG_HF_MOTOR_GUESS] > 0.f)) return;
  const int n = ln.bi(b)[DG_BI_N_LINKS]; if (n == 0) return;
  // (registers up to eight joints, with one active-set round; up to ten in LDS, all-or-nothing; none beyond: on a 12-joint
  // tree under saturating position control the LDS factorisation cost 17 % of the step and the clamped guess two more sweeps)
  if (n <= 6) motor_guess_small<LANES, 6>(ln, b); else if (n <= DG_MOTOR_GUESS_REFINE) motor_guess_small<LANES, DG_MOTOR_GUESS_REFINE>(ln, b); else if (n <= DG_MOTOR_GUESS_MAX) motor_guess_lds(ln, b);
}

// one PGS update of contact row at ro; returns the squared velocity residual
template <int LANES>
DGD float solve_crow(const Lane<LANES>& ln, int ro, int co, float lo, float hi, bool live, bool has) {
  if (!has) return 0.f;  // lanes without this contact slot hold no row data at all
  // Branch-free over nv_max entries: Jacobians / responses are zero-padded, and reading a few slots past a body's
  // velocity block only ever multiplies them by those zeros (the DV region ends with nv_max slots of padding).
  const int nvm = ln.sc.nv_max, tl = ln.sc.crow_tail; const bool two = tl > 2 * nvm;
  const int dA = (int)ln.L(co + CL_DVA), dB = (int)ln.L(co + CL_DVB), nB = (int)ln.L(co + CL_NVB);
  float jv = lds_dot(ln, ro, dA, nvm);
  if (two && nB > 0) jv += lds_dot(ln, ro + 2 * nvm, dB, nvm);
  const float diag = ln.L(ro + tl + 2), acc = ln.L(ro + tl + 1);
  float delta = (ln.L(ro + tl) - jv) / diag;
  const float nacc = fminf(fmaxf(acc + delta, lo), hi);
  delta = live && diag > 1e-18f ? nacc - acc : 0.f;
  ln.L(ro + tl + 1) = acc + delta;
  lds_axpy(ln, dA, ro + nvm, delta, nvm);
  if (two && nB > 0) lds_axpy(ln, dB, ro + 3 * nvm, delta, nvm);
  const float res = delta * diag; return res * res;
}

// ---- motor / joint-limit rows of one body, one Gauss-Seidel sweep ------------------------------------
// generic version: any body, everything through LDS
template <int LANES, bool LIMITS>
DGD float pgs_rows_generic(const Lane<LANES>& ln, int b, bool live) {
  const DevScene& sc = ln.sc; const float h = sc.h; float maxres = 0.f;
  const int first = ln.bi(b)[DG_BI_FIRST_LINK], n = ln.bi(b)[DG_BI_N_LINKS], k0 = ln.fixed(b) ? 0 : 6;
  const int nv = ln.plb(b)[PLB_NV], dvo = ln.plb(b)[PLB_DV], mvo = ln.plb(b)[PLB_MINV];
  for (int i = 0; i < n; i++) {
    const int gl = first + i, j = k0 + i, mo = ln.pll(gl)[PLL_MROW], col = mvo + j * nv;
    const float diag = ln.L(col + j);
    if (!LIMITS) {
      const float maxf = ln.mt.v[3 * gl + 2]; const float maximp = maxf < 0.f ? -maxf : maxf * sc.hm;
      if (!(maximp > 0.f)) continue;
      const float acc = ln.L(mo + MR_ACC);
      float delta = (ln.L(mo + MR_B) - ln.L(dvo + j)) / diag;
      float nacc = fminf(fmaxf(acc + delta, -maximp), maximp);
      delta = live ? nacc - acc : 0.f; ln.L(mo + MR_ACC) = acc + delta;
      lds_axpy(ln, dvo, col, delta, nv);
      float res = delta * diag; maxres = fmaxf(maxres, res * res);
    } else {
      cfp f = ln.lf(gl); if (!(f[DG_LF_LOWER] <= f[DG_LF_UPPER])) continue;
#pragma unroll
      for (int side = 0; side < 2; side++) {
        const float sg = side == 0 ? 1.f : -1.f; const int bo = mo + (side == 0 ? MR_LO_B : MR_HI_B);
        const float acc = ln.L(bo + 1); const bool act = acc >= 0.f;
        if (!__any(act)) continue;
        float delta = (ln.L(bo) - sg * ln.L(dvo + j)) / diag;
        float nacc = fmaxf(acc + delta, 0.f);
        delta = (live && act) ? nacc - acc : 0.f; if (act) ln.L(bo + 1) = acc + delta;
        lds_axpy(ln, dvo, col, sg * delta, nv);
        float res = delta * diag; maxres = fmaxf(maxres, res * res);
      }
    }
  }
  return maxres;
}
// fixed-base body with at most MAXN joints (every arm): the body's velocity change stays in registers for the
// whole sweep and the loops are unrolled, so a row costs 2 LDS reads + one M^-1 column instead of 3 n accesses
template <int LANES, int MAXN, bool LIMITS>
DGD float pgs_rows_small(const Lane<LANES>& ln, int b, bool live) {
  const DevScene& sc = ln.sc; const float h = sc.h; float maxres = 0.f;
  const int first = ln.bi(b)[DG_BI_FIRST_LINK], n = ln.bi(b)[DG_BI_N_LINKS];
  const int dvo = ln.plb(b)[PLB_DV], mvo = ln.plb(b)[PLB_MINV], mo0 = ln.pll(first)[PLL_MROW];
  float dv[MAXN];
#pragma unroll
  for (int k = 0; k < MAXN; k++) dv[k] = k < n ? ln.L(dvo + k) : 0.f;
#pragma unroll
  for (int i = 0; i < MAXN; i++) {
    if (i < n) {
      const int gl = first + i, mo = mo0 + i * MR_STRIDE /* MROW blocks of a body are contiguous */, col = mvo + i * n;
      const float diag = ln.L(col + i);
      if (!LIMITS) {
        const float maxf = ln.mt.v[3 * gl + 2]; const float maximp = maxf < 0.f ? -maxf : maxf * sc.hm;
        if (maximp > 0.f) {
          const float acc = ln.L(mo + MR_ACC);
          float delta = (ln.L(mo + MR_B) - dv[i]) / diag;
          float nacc = fminf(fmaxf(acc + delta, -maximp), maximp);
          delta = live ? nacc - acc : 0.f; ln.L(mo + MR_ACC) = acc + delta;
#pragma unroll
          for (int k = 0; k < MAXN; k++) if (k < n) dv[k] += ln.L(col + k) * delta;
          float res = delta * diag; maxres = fmaxf(maxres, res * res);
        }
      } else {
        cfp f = ln.lf(gl);
        if (f[DG_LF_LOWER] <= f[DG_LF_UPPER]) {
#pragma unroll
          for (int side = 0; side < 2; side++) {
            const float sg = side == 0 ? 1.f : -1.f; const int bo = mo + (side == 0 ? MR_LO_B : MR_HI_B);
            const float acc = ln.L(bo + 1); const bool act = acc >= 0.f;
            if (__any(act)) {
              float delta = (ln.L(bo) - sg * dv[i]) / diag;
              float nacc = fmaxf(acc + delta, 0.f);
              delta = (live && act) ? nacc - acc : 0.f; if (act) ln.L(bo + 1) = acc + delta;
#pragma unroll
              for (int k = 0; k < MAXN; k++) if (k < n) dv[k] += sg * ln.L(col + k) * delta;
              float res = delta * diag; maxres = fmaxf(maxres, res * res);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXN; k++) if (k < n) ln.L(dvo + k) = dv[k];
  return maxres;
}

// ---- every row of the scene with the velocity change in registers (total DoF <= NTB <= 32, no register-chain
// bodies).  The whole Gauss-Seidel loop runs here: the velocity change never leaves the registers between
// iterations, motor / joint-limit rows read their M^-1 column straight from LDS into a register vector, contact rows
// are streamed [J | R | b acc diag] with the NEXT row's loads issued before the current row is solved (an LDS store
// would otherwise fence them), and vectors are padded to NTB in registers by clamped-address loads times a 0/1
// mask -- no branches inside a row.
// Per-link row descriptors of the dense sweeps, kept in VGPRs as 64-entry wave-uniform tables (lane gl holds link
// gl; v_readlane fetches an entry in a few cycles).  Looking them up through the scene tables instead costs a chain
// of dependent scalar loads per row per iteration.  FULL: all 64 lanes of the wave are ACTIVE at the call (lane gl
// must exist and execute; not so in the reset kernel, which steps under a per-env mask, nor in the early-exit
// 16 / 32-lane kernels).
template <int LANES, bool FULL>
struct LinkRows {
  int col, mo, j, base, nv; float lim; uint64_t motors;
  const Lane<LANES>& ln;
  DGD LinkRows(const Lane<LANES>& l) : ln(l) {
    col = mo = j = base = nv = 0; lim = 0.f; motors = 0ull;
    if constexpr (FULL) {
      const int gl = threadIdx.x & 63;
      if (gl < ln.sc.nl) fill(gl, col, mo, j, base, nv, lim);
      motors = __ballot(lim > 0.f);
    } else {
      for (int gl = 0; gl < ln.sc.nl && gl < 64; gl++) { const float maxf = ln.mt.v[3 * gl + 2]; if ((maxf < 0.f ? -maxf : maxf * ln.sc.hm) > 0.f) motors |= 1ull << gl; }
    }
  }
  DGD void fill(int gl, int& c, int& m, int& jj, int& bs, int& n, float& lm) const {
    const DevScene& sc = ln.sc;
    const int b = sc.LI[gl * DG_LI_STRIDE + DG_LI_BODY]; cip B = sc.BI + b * DG_BI_STRIDE; cip P = sc.PLB + b * PLB_STRIDE;
    n = P[PLB_NV]; bs = P[PLB_DV] - sc.dv_base;
    const int jb = ((B[DG_BI_FLAGS] & DG_BODY_FIXED) ? 0 : 6) + gl - B[DG_BI_FIRST_LINK];
    c = P[PLB_MINV] + jb * n; jj = bs + jb; m = sc.PLL[gl * PLL_STRIDE + PLL_MROW];
    const float maxf = ln.mt.v[3 * gl + 2]; lm = maxf < 0.f ? -maxf : maxf * sc.hm;
  }
  DGD void get(int gl, int& c, int& m, int& jj, int& bs, int& n, float& lm) const {
    if constexpr (FULL) {
      c = __builtin_amdgcn_readlane(col, gl); m = __builtin_amdgcn_readlane(mo, gl); jj = __builtin_amdgcn_readlane(j, gl);
      bs = __builtin_amdgcn_readlane(base, gl); n = __builtin_amdgcn_readlane(nv, gl);
      lm = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lim), gl));
    } else fill(gl, c, m, jj, bs, n, lm);
  }
};

template <int NTB> struct DenseRow { float J[NTB], R[NTB], b, acc, diag; };
template <int NTB> struct DenseCol { float R[NTB], b, acc, diag, lim; int mo, j; };

template <int LANES, int NTB, bool PROF, bool FULLWAVE>
DGD int pgs_dense(const Lane<LANES>& ln, int ncont, int wave_max_cont, uint64_t limit_rows, Prof<PROF>& prof) {
  const DevScene& sc = ln.sc; const int nt = sc.nt, rs = sc.crow_tail + 3; const float h = sc.h;
  const float thr = sc.HF[DG_HF_RESIDUAL_THRESHOLD];
  constexpr int T0 = NTB - 8;  // elements below T0 always exist (bucket choice)
  float dv[NTB];
#pragma unroll
  for (int k = 0; k < NTB; k++) dv[k] = 0.f;
  // v[k] = L(o + k) for k < nt, 0 beyond: every element is loaded (the <= 7 slots past the vector are allocated
  // workspace whose contents do not matter), then the tail is cleared by a select on the uniform bound
  auto load_vec = [&](float (&v)[NTB], int o) {
#pragma unroll
    for (int k = 0; k < NTB; k++) v[k] = ln.L(o + k);
#pragma unroll
    for (int k = T0; k < NTB; k++) v[k] = k < nt ? v[k] : 0.f;
  };
  auto load_row = [&](DenseRow<NTB>& r, int ro) { load_vec(r.J, ro); load_vec(r.R, ro + nt); r.b = ln.L(ro + 2 * nt); r.acc = ln.L(ro + 2 * nt + 1); r.diag = ln.L(ro + 2 * nt + 2); };
  float maxres = 0.f; bool live = ln.valid;
  auto solve_row = [&](const DenseRow<NTB>& r, int ro, float lo, float hi) {
    float jv = 0.f;
#pragma unroll
    for (int k = 0; k < NTB; k++) jv += r.J[k] * dv[k];
    float delta = fdiv(r.b - jv, r.diag);
    const float nacc = fminf(fmaxf(r.acc + delta, lo), hi);
    delta = live && r.diag > 1e-18f ? nacc - r.acc : 0.f;
    ln.L(ro + 2 * nt + 1) = r.acc + delta;
#pragma unroll
    for (int k = 0; k < NTB; k++) dv[k] += r.R[k] * delta;
    const float res = delta * r.diag; maxres = fmaxf(maxres, res * res);
  };
  // a motor or limit row on global DoF j of a body whose DoFs are [base, base + nv): R = column of the body's M^-1
  auto load_col = [&](float (&v)[NTB], int col, int base, int nv) {
    // the body's DoFs are [base, base + nv) of the global vector: read the column as if it started `base` slots
    // earlier (those slots are allocated workspace -- M^-1 blocks come after the pose slots) and clear the rest
#pragma unroll
    for (int k = 0; k < NTB; k++) v[k] = ln.L(col - base + k);
#pragma unroll
    for (int k = 0; k < NTB; k++) v[k] = (k >= base && k < base + nv) ? v[k] : 0.f;
  };
  const LinkRows<LANES, FULLWAVE> rows(ln);
  auto load_motor = [&](DenseCol<NTB>& r, int gl) {
    int col, base, nv; rows.get(gl, col, r.mo, r.j, base, nv, r.lim);
    load_col(r.R, col, base, nv); r.b = ln.L(r.mo + MR_B); r.acc = ln.L(r.mo + MR_ACC); r.diag = ln.L(col + r.j - base);
  };
  auto solve_motor = [&](const DenseCol<NTB>& r) {
    float delta = fdiv(r.b - dv[r.j], r.diag);
    const float nacc = fminf(fmaxf(r.acc + delta, -r.lim), r.lim);
    delta = live ? nacc - r.acc : 0.f; ln.L(r.mo + MR_ACC) = r.acc + delta;
#pragma unroll
    for (int k = 0; k < NTB; k++) dv[k] += r.R[k] * delta;
    const float res = delta * r.diag; maxres = fmaxf(maxres, res * res);
  };
  if (wave_max_cont > 0 && sc.warm_off >= 0) {  // warm start: the velocity change the rows' starting impulses amount to
    for (int r = 0; r < 3 * wave_max_cont; r++) {
      DenseRow<NTB> W; load_row(W, sc.tr_off + r * rs); const bool on = r < 3 * ncont;  // (a slot beyond THIS env's contacts holds whatever was there: select, never multiply by zero)
#pragma unroll
      for (int k = 0; k < NTB; k++) dv[k] += on ? W.R[k] * W.acc : 0.f;
    }
  }
  for (uint64_t m = rows.motors; m; m &= m - 1) {  // ... and the motor rows' (motor_guess; limit rows of pinned joints: limit_guess)
    DenseCol<NTB> W; load_motor(W, __ffsll((long long)m) - 1);
    const float tot = W.acc + fmaxf(ln.L(W.mo + MR_LO_ACC), 0.f) - fmaxf(ln.L(W.mo + MR_HI_ACC), 0.f);
#pragma unroll
    for (int k = 0; k < NTB; k++) dv[k] += W.R[k] * tot;
  }
  int iters_done = 0;
  for (int it = 0; it < sc.iters; it++) {
    maxres = 0.f;
    // ---- motor rows (oracle order: link by link), loads one row ahead of the solve
    {
      DenseCol<NTB> A, B;
      uint64_t m = rows.motors;
      if (m) load_motor(A, __ffsll((long long)m) - 1);
      while (m) {
        m &= m - 1; if (m) load_motor(B, __ffsll((long long)m) - 1);
        solve_motor(A);
        if (!m) break;
        m &= m - 1; if (m) load_motor(A, __ffsll((long long)m) - 1);
        solve_motor(B);
      }
    }
    prof.stamp(PS_PGS_MOTOR);
    // ---- joint-limit rows: only those some lane has active (the flag cannot change during the sweeps)
    for (uint64_t m = limit_rows; m; m &= m - 1) {
      const int bit = __ffsll((long long)m) - 1, gl = bit >> 1, side = bit & 1;
      int col, mo, jg, base, nv; float lim_unused; rows.get(gl, col, mo, jg, base, nv, lim_unused);
      const int jb = jg - base, bo = mo + (side == 0 ? MR_LO_B : MR_HI_B); const float sg = side == 0 ? 1.f : -1.f;
      float R[NTB]; load_col(R, col, base, nv);
      const float diag = ln.L(col + jb), acc = ln.L(bo + 1), bb = ln.L(bo); const bool act = acc >= 0.f;
      float delta = fdiv(bb - sg * dv[base + jb], diag);
      const float nacc = fmaxf(acc + delta, 0.f);
      delta = (live && act) ? nacc - acc : 0.f; if (act) ln.L(bo + 1) = acc + delta;
      const float sd = sg * delta;
#pragma unroll
      for (int k = 0; k < NTB; k++) dv[k] += R[k] * sd;
      const float res = delta * diag; maxres = fmaxf(maxres, res * res);
    }
    prof.stamp(PS_PGS_LIMIT);
    // ---- contact normals, then friction pairs; two row buffers ping-pong so that loads run one row ahead
    if (wave_max_cont > 0) {
      DenseRow<NTB> A, B;
      const int r0 = sc.tr_off, cl = wave_max_cont - 1;
      if (0 < ncont) load_row(A, r0);
      for (int c = 0; c < wave_max_cont; c += 2) {
        if (c + 1 < ncont) load_row(B, r0 + 3 * (c + 1) * rs);
        if (c < ncont) solve_row(A, r0 + 3 * c * rs, 0.f, 3.0e38f);
        if (c + 2 < ncont) load_row(A, r0 + 3 * (c + 2) * rs);
        if (c + 1 < ncont) solve_row(B, r0 + 3 * (c + 1) * rs, 0.f, 3.0e38f);
      }
      {  // friction pairs: the two rows, the friction coefficient and the normal impulse of contact c + 1 are read
         // (unconditionally: past the end they re-read the last contact) while contact c is solved
        DenseRow<NTB> A2, B2; float muA, muB, naA, naB;
        auto fetch = [&](DenseRow<NTB>& r1, DenseRow<NTB>& r2, float& mu, float& na, int c) {
          const int cc = min(c, cl);
          load_row(r1, r0 + (3 * cc + 1) * rs); load_row(r2, r0 + (3 * cc + 2) * rs);
          mu = ln.L(sc.cont_off + 1 + cc * CL_STRIDE + CL_MU); na = ln.L(r0 + 3 * cc * rs + 2 * nt + 1);
        };
        auto pair = [&](const DenseRow<NTB>& r1, const DenseRow<NTB>& r2, float mu, float na, int c) {
          if (c < ncont && mu > 0.f) { const float lim = mu * na; solve_row(r1, r0 + (3 * c + 1) * rs, -lim, lim); solve_row(r2, r0 + (3 * c + 2) * rs, -lim, lim); }
        };
        fetch(A, A2, muA, naA, 0);
        for (int c = 0; c < wave_max_cont; c += 2) {
          fetch(B, B2, muB, naB, c + 1);
          pair(A, A2, muA, naA, c);
          fetch(A, A2, muA, naA, c + 2);
          if (c + 1 < wave_max_cont) pair(B, B2, muB, naB, c + 1);
        }
      }
    }
    prof.stamp(PS_PGS_CONTACT);
    if (live) iters_done = it + 1;
    live = live && !(maxres <= thr);
    if (!__any(live)) break;
  }
#pragma unroll
  for (int k = 0; k < NTB; k++) if (k < nt) ln.L(sc.dv_base + k) = dv[k];
  return iters_done;
}

// ---- the same sweeps with SL = 64 / LANES lanes per environment ------------------------------------------------
// Modes with fewer than 64 envs per wavefront (LDS-heavy scenes) leave 3/4 or 1/2 of every VALU instruction idle.
// During the Gauss-Seidel loop -- the bulk of such a step -- the idle lanes are put to work: lane l serves env
// (l / SL) and owns the DoFs k with k % SL == l % SL.  A row costs each lane NTB / SL loads and FMAs per vector
// plus a DPP reduction of the partial J.dv over the group (quad permutes, then row_half_mirror / row_mirror for
// groups of 8 / 16); every lane of an env's group computes the same impulse.
DGD float group_sum2(float x) { return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); }  // quad_perm [1,0,3,2]
DGD float group_sum4(float x) { x = group_sum2(x); return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)); }  // then [2,3,0,1]
DGD float group_sum8(float x) { x = group_sum4(x); return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true)); }   // then row_half_mirror
DGD float group_sum16(float x) { x = group_sum8(x); return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true)); }  // then row_mirror

template <int LANES, int NTB, bool PROF>
DGD int pgs_dense_sliced(const Lane<LANES>& ln, int ncont_primary, int wave_max_cont, uint64_t limit_rows, Prof<PROF>& prof) {
  static_assert(LANES == 32 || LANES == 16 || LANES == 8 || LANES == 4, "sliced sweeps are for the modes with fewer than 64 envs per wavefront");
  constexpr int SL = 64 / LANES, LOG = SL == 16 ? 4 : SL == 8 ? 3 : SL == 4 ? 2 : 1, NS = NTB / SL;
  static_assert(NTB % SL == 0 && NS >= 1, "the padded DoF count must be a multiple of the lanes per env");
  const DevScene& sc = ln.sc; const int nt = sc.nt, rs = sc.crow_tail + 3; const float h = sc.h;
  const float thr = sc.HF[DG_HF_RESIDUAL_THRESHOLD];
  const int lane = threadIdx.x, sl = lane & (SL - 1), q = lane >> LOG;
  const int envq = blockIdx.x * LANES + q; const bool validq = envq < sc.num_envs; const int eq = validq ? envq : sc.num_envs - 1;
  const Lane<LANES> lq(sc, ln.mt, ln.lds - lane + q, ln.st - ln.env + eq, eq, validq);
  const int ncont = __shfl(ncont_primary, q);
  float* const ls = lq.lds + sl * LANES;  // this lane's slice: slot (o + i SL) through ls is DoF i SL + sl of vector o
  auto group_sum = [&](float x) { return SL == 16 ? group_sum16(x) : SL == 8 ? group_sum8(x) : SL == 4 ? group_sum4(x) : group_sum2(x); };
  float dv[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) dv[i] = 0.f;
  auto load_vec = [&](float (&v)[NS], int o) {
#pragma unroll
    for (int i = 0; i < NS; i++) v[i] = ls[(o + i * SL) * LANES];
#pragma unroll
    for (int i = (NTB - 8) / SL; i < NS; i++) v[i] = (i * SL + sl) < nt ? v[i] : 0.f;
  };
  auto load_col = [&](float (&v)[NS], int col, int base, int nv) {
    // the body's DoFs are [base, base + nv) of the global vector: read the column as if it started `base` slots
    // earlier (allocated workspace) and clear what is not the body's
#pragma unroll
    for (int i = 0; i < NS; i++) v[i] = ls[(col - base + i * SL) * LANES];
#pragma unroll
    for (int i = 0; i < NS; i++) { const int k = i * SL + sl; v[i] = (k >= base && k < base + nv) ? v[i] : 0.f; }
  };
  // DoF j of the velocity change, known to every lane of the group
  auto dv_at = [&](int j) { const float mine = dv[j >> LOG]; return group_sum((j & (SL - 1)) == sl ? mine : 0.f); };
  struct Row { float J[NS], R[NS], b, acc, diag; };
  struct Col { float R[NS], b, acc, diag, lim; int mo, j; };
  auto load_row = [&](Row& r, int ro) { load_vec(r.J, ro); load_vec(r.R, ro + nt); r.b = lq.L(ro + 2 * nt); r.acc = lq.L(ro + 2 * nt + 1); r.diag = lq.L(ro + 2 * nt + 2); };
  float maxres = 0.f; bool live = validq;
  auto solve_row = [&](const Row& r, int ro, float lo, float hi) {
    float jp = 0.f;
#pragma unroll
    for (int i = 0; i < NS; i++) jp += r.J[i] * dv[i];
    const float jv = group_sum(jp);
    float delta = fdiv(r.b - jv, r.diag);
    const float nacc = fminf(fmaxf(r.acc + delta, lo), hi);
    delta = live && r.diag > 1e-18f ? nacc - r.acc : 0.f;
    lq.L(ro + 2 * nt + 1) = r.acc + delta;  // every lane of the group stores the same value
#pragma unroll
    for (int i = 0; i < NS; i++) dv[i] += r.R[i] * delta;
    const float res = delta * r.diag; maxres = fmaxf(maxres, res * res);
  };
  const LinkRows<LANES, true> rows(lq);
  auto load_motor = [&](Col& r, int gl) {
    int col, base, nv; rows.get(gl, col, r.mo, r.j, base, nv, r.lim);
    load_col(r.R, col, base, nv); r.b = lq.L(r.mo + MR_B); r.acc = lq.L(r.mo + MR_ACC); r.diag = lq.L(col + r.j - base);
  };
  auto solve_motor = [&](const Col& r) {
    float delta = fdiv(r.b - dv_at(r.j), r.diag);
    const float nacc = fminf(fmaxf(r.acc + delta, -r.lim), r.lim);
    delta = live ? nacc - r.acc : 0.f; lq.L(r.mo + MR_ACC) = r.acc + delta;
#pragma unroll
    for (int i = 0; i < NS; i++) dv[i] += r.R[i] * delta;
    const float res = delta * r.diag; maxres = fmaxf(maxres, res * res);
  };
  if (wave_max_cont > 0 && sc.warm_off >= 0) {  // warm start: the velocity change the rows' starting impulses amount to
    for (int r = 0; r < 3 * wave_max_cont; r++) {
      Row W; load_row(W, sc.tr_off + r * rs); const bool on = r < 3 * ncont;  // (a slot beyond THIS env's contacts holds whatever was there -- NaN patterns included: select, never multiply by zero)
#pragma unroll
      for (int i = 0; i < NS; i++) dv[i] += on ? W.R[i] * W.acc : 0.f;
    }
  }
  for (uint64_t m = rows.motors; m; m &= m - 1) {  // ... and the motor rows' (motor_guess; limit rows of pinned joints: limit_guess)
    Col W; load_motor(W, __ffsll((long long)m) - 1);
    const float tot = W.acc + fmaxf(lq.L(W.mo + MR_LO_ACC), 0.f) - fmaxf(lq.L(W.mo + MR_HI_ACC), 0.f);
#pragma unroll
    for (int i = 0; i < NS; i++) dv[i] += W.R[i] * tot;
  }
  int iters_done = 0;
  for (int it = 0; it < sc.iters; it++) {
    maxres = 0.f;
    if (rows.motors) {  // loads are unconditional (a row past the end re-reads the last one) so that the
      Col A, B;         // compiler can count the loads in flight instead of draining them at every branch
      uint64_t m = rows.motors; int g = __ffsll((long long)m) - 1;
      load_motor(A, g);
      while (true) {
        m &= m - 1; const bool more1 = m != 0; g = more1 ? __ffsll((long long)m) - 1 : g; load_motor(B, g);
        solve_motor(A);
        if (!more1) break;
        m &= m - 1; const bool more2 = m != 0; g = more2 ? __ffsll((long long)m) - 1 : g; load_motor(A, g);
        solve_motor(B);
        if (!more2) break;
      }
    }
    prof.stamp(PS_PGS_MOTOR);
    for (uint64_t m = limit_rows; m; m &= m - 1) {
      const int bit = __ffsll((long long)m) - 1, gl = bit >> 1, side = bit & 1;
      int col, mo, jg, base, nv; float lim_unused; rows.get(gl, col, mo, jg, base, nv, lim_unused);
      const int jb = jg - base, bo = mo + (side == 0 ? MR_LO_B : MR_HI_B); const float sg = side == 0 ? 1.f : -1.f;
      float R[NS]; load_col(R, col, base, nv);
      const float diag = lq.L(col + jb), acc = lq.L(bo + 1), bb = lq.L(bo); const bool act = acc >= 0.f;
      float delta = fdiv(bb - sg * dv_at(base + jb), diag);
      const float nacc = fmaxf(acc + delta, 0.f);
      delta = (live && act) ? nacc - acc : 0.f; if (act) lq.L(bo + 1) = acc + delta;
      const float sd = sg * delta;
#pragma unroll
      for (int i = 0; i < NS; i++) dv[i] += R[i] * sd;
      const float res = delta * diag; maxres = fmaxf(maxres, res * res);
    }
    prof.stamp(PS_PGS_LIMIT);
    if (wave_max_cont > 0) {
      Row A, B;
      const int r0 = sc.tr_off;
      const int cl = wave_max_cont - 1;  // loads are unconditional: past the end they re-read the last row
      load_row(A, r0);
      for (int c = 0; c < wave_max_cont; c += 2) {
        load_row(B, r0 + 3 * min(c + 1, cl) * rs);
        if (c < ncont) solve_row(A, r0 + 3 * c * rs, 0.f, 3.0e38f);
        load_row(A, r0 + 3 * min(c + 2, cl) * rs);
        if (c + 1 < ncont) solve_row(B, r0 + 3 * (c + 1) * rs, 0.f, 3.0e38f);
      }
      {  // friction pairs: the two rows, the friction coefficient and the normal impulse of contact c + 1 are read
         // (unconditionally: past the end they re-read the last contact) while contact c is solved
        Row A2, B2; float muA, muB, naA, naB;
        auto fetch = [&](Row& r1, Row& r2, float& mu, float& na, int c) {
          const int cc = min(c, cl);
          load_row(r1, r0 + (3 * cc + 1) * rs); load_row(r2, r0 + (3 * cc + 2) * rs);
          mu = lq.L(sc.cont_off + 1 + cc * CL_STRIDE + CL_MU); na = lq.L(r0 + 3 * cc * rs + 2 * nt + 1);
        };
        auto pair = [&](const Row& r1, const Row& r2, float mu, float na, int c) {
          if (c < ncont && mu > 0.f) { const float lim = mu * na; solve_row(r1, r0 + (3 * c + 1) * rs, -lim, lim); solve_row(r2, r0 + (3 * c + 2) * rs, -lim, lim); }
        };
        fetch(A, A2, muA, naA, 0);
        for (int c = 0; c < wave_max_cont; c += 2) {
          fetch(B, B2, muB, naB, c + 1);
          pair(A, A2, muA, naA, c);
          fetch(A, A2, muA, naA, c + 2);
          if (c + 1 < wave_max_cont) pair(B, B2, muB, naB, c + 1);
        }
      }
    }
    prof.stamp(PS_PGS_CONTACT);
    if (live) iters_done = it + 1;
    live = live && !(maxres <= thr);
    if (!__any(live)) break;
  }
#pragma unroll
  for (int i = 0; i < NS; i++) if (i * SL + sl < nt) ls[(sc.dv_base + i * SL) * LANES] = dv[i];
  return __shfl(iters_done, (lane << LOG) & 63);  // primary lane e reads the count of env e's group
}

// ---- the same sliced sweeps with EVERY row in registers ------------------------------------------------------------
// A row of the LDS version costs ~340-470 cycles (seven LDS reads, address arithmetic, an impulse store that the next
// reads queue behind, per-row branches) for ~25 instructions of arithmetic, and a step of a contact scene is 10 000+
// strictly sequential rows.  Here the rows a lane needs -- its slice of every M^-1 column, of every contact row's J and R,
// right-hand sides, reciprocal diagonals, limits -- are loaded ONCE per substep, the accumulated impulses live in
// registers, and a sweep is straight-line code over NLR links and CB contacts (absent rows have zero limits / zero data,
// so they are no-ops without a branch; only the joint-limit block, rarely active, is guarded per row).  Preconditions
// (checked by the caller): NTB / SL <= 2 entries per lane, at most NLR links, at most CB contacts in any env of the wave.
// (A hybrid for bigger scenes -- impulses in registers, contact rows re-read from LDS -- was tried for from_the_readme,
// 18 links / 25 contacts: 450+ live registers, slower than the LDS sweeps; not kept.)
template <int LANES, int NTB, int NLR, int CB, bool PROF>
DGD int pgs_dense_sliced_regs(const Lane<LANES>& ln, int ncont_primary, uint64_t limit_rows, Prof<PROF>& prof) {
  constexpr int SL = 64 / LANES, LOG = SL == 16 ? 4 : SL == 8 ? 3 : SL == 4 ? 2 : 1, NS = NTB / SL;
  static_assert(NTB % SL == 0 && NS >= 1 && NS <= 2, "register-resident sweeps hold at most two entries of a vector per lane");
  const DevScene& sc = ln.sc; const int nt = sc.nt, rs = sc.crow_tail + 3;
  const float thr = sc.HF[DG_HF_RESIDUAL_THRESHOLD];
  const int lane = threadIdx.x, sl = lane & (SL - 1), q = lane >> LOG;
  const int envq = blockIdx.x * LANES + q; const bool validq = envq < sc.num_envs; const int eq = validq ? envq : sc.num_envs - 1;
  const Lane<LANES> lq(sc, ln.mt, ln.lds - lane + q, ln.st - ln.env + eq, eq, validq);
  const int ncont = __shfl(ncont_primary, q);
  float* const ls = lq.lds + sl * LANES;  // this lane's slice: slot (o + i SL) through ls is DoF i SL + sl of vector o
  auto group_sum = [&](float x) { return SL == 16 ? group_sum16(x) : SL == 8 ? group_sum8(x) : SL == 4 ? group_sum4(x) : group_sum2(x); };
  float dv[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) dv[i] = 0.f;
  // ---- per link: slice of the M^-1 column, diagonal, motor and limit rows
  const LinkRows<LANES, true> rows(lq);
  float lR[NLR][NS], ldg[NLR], lrd[NLR], mb[NLR], mlim[NLR], macc[NLR], lb[2][NLR], la[2][NLR]; int lji[NLR]; bool lmine[NLR];
#pragma unroll
  for (int gl = 0; gl < NLR; gl++) {
    const bool have = gl < sc.nl; int col, mo, j, base, nv; float lim; rows.get(have ? gl : 0, col, mo, j, base, nv, lim);
#pragma unroll
    for (int i = 0; i < NS; i++) { const int k = i * SL + sl; const float v = ls[(col - base + i * SL) * LANES]; lR[gl][i] = (have && k >= base && k < base + nv) ? v : 0.f; }
    const float dg = lq.L(col + j - base);
    ldg[gl] = have ? dg : 1.f; lrd[gl] = have ? frcp(dg) : 0.f;
    mb[gl] = have ? lq.L(mo + MR_B) : 0.f; mlim[gl] = (have && ((rows.motors >> gl) & 1ull)) ? lim : 0.f;
    macc[gl] = (have && ((rows.motors >> gl) & 1ull)) ? lq.L(mo + MR_ACC) : 0.f;  // (starting impulse: motor_guess)
#pragma unroll
    for (int i = 0; i < NS; i++) dv[i] += lR[gl][i] * macc[gl];
    lb[0][gl] = lq.L(mo + MR_LO_B); la[0][gl] = have ? lq.L(mo + MR_LO_ACC) : -1.f; lb[1][gl] = lq.L(mo + MR_HI_B); la[1][gl] = have ? lq.L(mo + MR_HI_ACC) : -1.f;
    { const float tl = fmaxf(la[0][gl], 0.f) - fmaxf(la[1][gl], 0.f);  // (starting impulses of pinned limit rows: limit_guess)
      _Pragma("unroll") for (int i = 0; i < NS; i++) dv[i] += lR[gl][i] * tl; }
    lji[gl] = j >> LOG; lmine[gl] = (j & (SL - 1)) == sl;
  }
  // DoF j of the velocity change, known to every lane of the group
  auto dv_of = [&](int gl) { const float mine = (NS == 1 || lji[gl] == 0) ? dv[0] : dv[NS - 1]; return group_sum(lmine[gl] ? mine : 0.f); };
  // ---- contact rows: J pre-divided by the diagonal (delta = b' - J' . dv), R, right-hand side, diagonal, impulse
  float cJ[3 * CB][NS], cR[3 * CB][NS], cb[3 * CB], cdg[3 * CB], cacc[3 * CB], cmu[CB];
#pragma unroll
  for (int c = 0; c < CB; c++) {
    const bool has = c < ncont; const int cc = has ? c : 0;
    cmu[c] = has ? lq.L(sc.cont_off + 1 + cc * CL_STRIDE + CL_MU) : 0.f;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int r = 3 * c + d, ro = sc.tr_off + (3 * cc + d) * rs;
      const float dg = lq.L(ro + 2 * nt + 2), rd = (has && dg > 1e-18f) ? frcp(dg) : 0.f;
#pragma unroll
      for (int i = 0; i < NS; i++) {
        const bool in = has && (i * SL + sl) < nt;
        const float jj = ls[(ro + i * SL) * LANES], rr = ls[(ro + nt + i * SL) * LANES];
        cJ[r][i] = in ? jj * rd : 0.f; cR[r][i] = in ? rr : 0.f;
      }
      cb[r] = has ? lq.L(ro + 2 * nt) * rd : 0.f; cdg[r] = has ? dg : 0.f; cacc[r] = has ? lq.L(ro + 2 * nt + 1) : 0.f;  // (starting impulse: warm start)
#pragma unroll
      for (int i = 0; i < NS; i++) dv[i] += cR[r][i] * cacc[r];
    }
  }
  float maxres = 0.f; bool live = validq; int iters_done = 0;
  auto contact_row = [&](int r, float lo, float hi, float lv) {
    float jp = 0.f;
#pragma unroll
    for (int i = 0; i < NS; i++) jp += cJ[r][i] * dv[i];
    const float nacc = fminf(fmaxf(cacc[r] + (cb[r] - group_sum(jp)), lo), hi);
    const float delta = (nacc - cacc[r]) * lv; cacc[r] += delta;
#pragma unroll
    for (int i = 0; i < NS; i++) dv[i] += cR[r][i] * delta;
    const float res = delta * cdg[r]; maxres = fmaxf(maxres, res * res);
  };
  for (int it = 0; it < sc.iters; it++) {
    maxres = 0.f; const float lv = live ? 1.f : 0.f;
#pragma unroll
    for (int gl = 0; gl < NLR; gl++) {  // motor rows, link by link (oracle order); a link without a motor has limit 0
      const float nacc = __builtin_amdgcn_fmed3f(macc[gl] + (mb[gl] - dv_of(gl)) * lrd[gl], -mlim[gl], mlim[gl]);
      const float delta = (nacc - macc[gl]) * lv; macc[gl] += delta;
#pragma unroll
      for (int i = 0; i < NS; i++) dv[i] += lR[gl][i] * delta;
      const float res = delta * ldg[gl]; maxres = fmaxf(maxres, res * res);
    }
    if (limit_rows) {  // joint-limit rows: only those some lane has active (the flag cannot change during the sweeps)
#pragma unroll
      for (int gl = 0; gl < NLR; gl++) {
#pragma unroll
        for (int side = 0; side < 2; side++) {
          if (!((limit_rows >> (2 * gl + side)) & 1ull)) continue;
          const float sg = side == 0 ? 1.f : -1.f; const bool act = la[side][gl] >= 0.f;
          const float nacc = fmaxf(la[side][gl] + (lb[side][gl] - sg * dv_of(gl)) * lrd[gl], 0.f);
          const float delta = act ? (nacc - la[side][gl]) * lv : 0.f; la[side][gl] += delta;
          const float sd = sg * delta;
#pragma unroll
          for (int i = 0; i < NS; i++) dv[i] += lR[gl][i] * sd;
          const float res = delta * ldg[gl]; maxres = fmaxf(maxres, res * res);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CB; c++) contact_row(3 * c, 0.f, 3.0e38f, lv);  // contact normals, then the friction pairs
#pragma unroll
    for (int c = 0; c < CB; c++) { const float lim = cmu[c] * cacc[3 * c]; contact_row(3 * c + 1, -lim, lim, lv); contact_row(3 * c + 2, -lim, lim, lv); }
    if (live) iters_done = it + 1;
    live = live && !(maxres <= thr);
    if (!__any(live)) break;
  }
  prof.stamp(PS_PGS_CONTACT);
#pragma unroll
  for (int i = 0; i < NS; i++) if (i * SL + sl < nt) ls[(sc.dv_base + i * SL) * LANES] = dv[i];
  if (sl == 0) {  // one lane per env: motor impulses for the applied-torque readout, contact impulses for the force/torque sensor
#pragma unroll
    for (int gl = 0; gl < NLR; gl++) if (gl < sc.nl && ((rows.motors >> gl) & 1ull)) { int col, mo, j, base, nv; float lim; rows.get(gl, col, mo, j, base, nv, lim); lq.L(mo + MR_ACC) = macc[gl]; }
#pragma unroll
    for (int c = 0; c < CB; c++) if (c < ncont) { _Pragma("unroll") for (int d = 0; d < 3; d++) lq.L(sc.tr_off + (3 * c + d) * rs + 2 * nt + 1) = cacc[3 * c + d]; }
  }
  return __shfl(iters_done, (lane << LOG) & 63);  // primary lane e reads the count of env e's group
}
// picks the register-resident instantiation that covers the scene's links and this wavefront's contacts; false: none does
template <int LANES, int NTB, bool PROF>
DGD bool pgs_sliced_regs_dispatch(const Lane<LANES>& ln, int ncont, int wave_max_cont, uint64_t limit_rows, Prof<PROF>& prof, int& iters_done) {
  constexpr int SL = 64 / LANES;
  if constexpr (NTB % SL != 0 || NTB / SL > 2 || NTB / SL < 1) return false;
  else {
    const int nl = ln.sc.nl;
    if (nl > 16 || wave_max_cont > 12 || (limit_rows >> 32) != 0ull) return false;
#define DG_REGS(NLR, CB) do { iters_done = pgs_dense_sliced_regs<LANES, NTB, NLR, CB, PROF>(ln, ncont, limit_rows, prof); return true; } while (0)
    if (nl <= 8) { if (wave_max_cont <= 4) DG_REGS(8, 4); if (wave_max_cont <= 8) DG_REGS(8, 8); DG_REGS(8, 12); }
    if (wave_max_cont <= 4) DG_REGS(16, 4); if (wave_max_cont <= 8) DG_REGS(16, 8); DG_REGS(16, 12);
#undef DG_REGS
  }
}

// ---- sliced sweeps for the global-workspace mode (LANES == -16) ---------------------------------------------------
// Rows stream from the L2-resident scratch buffer, whose latency is ~10x that of LDS: loads run D rows ahead of the
// solves, which requires the loop to be free of global stores (vmcnt orders loads behind them) -- so the accumulated
// impulses, the only thing a sweep writes, live in LDS: acc[row id][env of the group].  Row ids: contact (c, d) ->
// 3 c + d, motor of link gl -> 3 maxc + gl, limit (gl, side) -> 3 maxc + nl + 2 gl + side.
// ---- one env per wavefront (LANES == 1): every row of the scene in registers, however many contacts ------------
// Big scenes at a modest batch (from_the_readme: 24 DoF, 18 links, up to 32 contacts, 1 024 envs) leave most SIMDs
// without a wavefront in the 4-envs-per-wavefront mode, and their rows do not fit that mode's register budget (every
// lane of an env's 16-lane group carries the row scalars).  With the whole wavefront on ONE env the row scalars are
// wave-uniform: lane k holds DoF k of every vector (two registers per contact row: J / diag and R), row r's scalars
// (right-hand side, impulse, diagonal) live in lane 16 + r % 16 of six registers, a link's in lane gl; a row update does
// its arithmetic in the owner lane's registers and v_readlane hands the impulse change to the other lanes.  No masks:
// `live`, the contact count and the active joint limits are uniform, so converged envs leave the loop, absent
// contacts are skipped by scalar branches.  Same row order and the same pre-scaled arithmetic as pgs_dense_sliced_regs.
DGD float row_pair_sum(float x) {  // x[l] + x[l ^ 16] in every lane (v_permlane16_swap: odd rows of a <-> even rows of b; see half_swap_sum)
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
DGD float rdl(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
// v_writelane_b32 with a compile-time lane (this compiler has no builtin for it): lane L of v becomes the uniform value s
template <int L> DGD float wrl(float v, float s) {
  int si = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s));  // (already uniform: pins it in an SGPR)
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(si), "n"(L));
  return v;
}
template <int I, int N, class F> DGD void static_for(F&& f) { if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); } }

template <bool PROF>
DGD int pgs_wave_env(const Lane<1>& lq, int ncont, uint64_t limit_rows, Prof<PROF>& prof) {
  constexpr int NLM = 32, CM = 32;  // links / contacts with register rows (dense scenes: <= 32 DoF; max_contacts <= 32)
  const DevScene& sc = lq.sc; const int nt = sc.nt, rs = sc.crow_tail + 3, nl = sc.nl;
  const float thr = sc.HF[DG_HF_RESIDUAL_THRESHOLD];
  const int lane = threadIdx.x & 63;
  // total over lanes 0..31, delivered to lanes 16..31 only (where the contact rows' owner lanes are): four DPP steps within
  // each row of 16, then row 1 adds lane 15 of row 0 (row_bcast:15, rows 1 and 3 written) -- no cross-row swap needed
  auto sum32 = [&](float x) {
    const float y = group_sum16(x);
    return y + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0x142, 0xA, 0xF, false));
  };
  float dv = 0.f;
  const LinkRows<1, true> rows(lq);  // lane gl: column offset, motor row, DoF, body base / size of link gl
  // ---- links, keyed by their DoF: the scalars of the joint that owns DoF j live in lane j -- the lane that holds
  // dv[j], so the motor / limit residual needs no broadcast of dv -- and its M^-1 column (lane k: entry of DoF k) is
  // register j of lRd.  DoFs ascend with the links (bodies and their links are laid out in order), so sweeping the DoFs in
  // order is sweeping the links in order.
  int mylink = -1;
  for (int gl = 0; gl < nl; gl++) { if (lane == __builtin_amdgcn_readlane(rows.j, gl)) mylink = gl; }
  const bool isj = mylink >= 0; const int ml = isj ? mylink : 0;
  const int mcol = __shfl(rows.col, ml), mmo = __shfl(rows.mo, ml), mbase = __shfl(rows.base, ml); const float mlimv = __shfl(rows.lim, ml);
  const bool has_motor = isj && ((rows.motors >> ml) & 1ull);
  float ldg = 1.f, lrd = 0.f, mb = 0.f, mlim = 0.f, macc = 0.f, lb0 = 0.f, la0 = -1.f, lb1 = 0.f, la1 = -1.f;
  if (isj) {
    ldg = lq.L(mcol + lane - mbase); lrd = frcp(ldg); mb = lq.L(mmo + MR_B); mlim = has_motor ? mlimv : 0.f; macc = has_motor ? lq.L(mmo + MR_ACC) : 0.f;  // (starting impulse: motor_guess)
    lb0 = lq.L(mmo + MR_LO_B); la0 = lq.L(mmo + MR_LO_ACC); lb1 = lq.L(mmo + MR_HI_B); la1 = lq.L(mmo + MR_HI_ACC);
  }
  const uint64_t jmask = __ballot(isj), jmotor = __ballot(has_motor);
  const uint64_t jlo = __ballot(isj && ((limit_rows >> (2 * ml)) & 1ull) && la0 >= 0.f), jhi = __ballot(isj && ((limit_rows >> (2 * ml + 1)) & 1ull) && la1 >= 0.f);
  float lRd[NLM];
  static_for<0, NLM>([&](auto jc) {
    constexpr int j = decltype(jc)::value; lRd[j] = 0.f;
    if ((jmask >> j) & 1ull) {
      const int g = __builtin_amdgcn_readlane(mylink, j); int col, mo, jj, base, nv; float lim; rows.get(g, col, mo, jj, base, nv, lim);
      const float v = lq.L(col - base + min(lane, nt - 1)); lRd[j] = (lane >= base && lane < base + nv) ? v : 0.f;
    }
  });
  // ---- contact rows: J / diag and R by DoF; scalars of row r = 3 c + d in lane r & 31 of slot r >> 5
  constexpr int NSL = (3 * CM + 15) / 16;  // scalar slots: row r lives in lane 16 + (r & 15) of slot r >> 4
  float cJ[3 * CM], cR[3 * CM], cbv[NSL], caccv[NSL], cdgv[NSL], cmuv[NSL];
#pragma unroll
  for (int s = 0; s < NSL; s++) {
    const int r = (lane - 16) + 16 * s; const bool has = lane >= 16 && lane < 32 && r < 3 * ncont; const int rr = has ? r : 0, ro = sc.tr_off + rr * rs;
    const float dg = lq.L(ro + 2 * nt + 2), rd = (has && dg > 1e-18f) ? frcp(dg) : 0.f;
    cbv[s] = has ? lq.L(ro + 2 * nt) * rd : 0.f; cdgv[s] = has ? dg : 0.f; caccv[s] = has ? lq.L(ro + 2 * nt + 1) : 0.f;  // (starting impulse: warm start)
    cmuv[s] = (has && rr % 3 == 0) ? lq.L(sc.cont_off + 1 + (rr / 3) * CL_STRIDE + CL_MU) : 0.f;  // friction coefficient: with the contact's normal row
  }
#pragma unroll
  for (int c = 0; c < CM; c++) {
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const int r = 3 * c + d; cJ[r] = 0.f; cR[r] = 0.f;
      if (c < ncont) {
        const int ro = sc.tr_off + r * rs; const float dg = lq.L(ro + 2 * nt + 2), rd = dg > 1e-18f ? frcp(dg) : 0.f;
        const bool in = lane < nt; const int k = min(lane, nt - 1);
        const float jj = lq.L(ro + k), rv = lq.L(ro + nt + k);
        cJ[r] = in ? jj * rd : 0.f; cR[r] = in ? rv : 0.f;
      }
    }
  }
  static_for<0, NLM>([&](auto jc) {  // the velocity change the motor rows' starting impulses amount to
    constexpr int j = decltype(jc)::value;
    if ((jmotor >> j) & 1ull) dv += lRd[j] * rdl(macc + fmaxf(la0, 0.f) - fmaxf(la1, 0.f), j);  // (limit rows of pinned joints: limit_guess)
  });
  if (sc.warm_off >= 0) static_for<0, 3 * CM>([&](auto rc) {  // warm start: the velocity change the rows' starting impulses amount to
    constexpr int R = decltype(rc)::value, o = 16 + (R & 15), sl_ = R >> 4;
    if (R < 3 * ncont) dv += cR[R] * rdl(caccv[sl_], o);
  });
  float maxres = 0.f; int iters_done = 0;
  // the owner lane's registers hold the row's scalars; v_readlane broadcasts what the other lanes need
  auto contact_row = [&](auto rc, float lim, bool friction) {  // row R (compile time: its registers, its owner lane); lim: friction bound
    constexpr int R = decltype(rc)::value, o = 16 + (R & 15), s = R >> 4;
    const float jv = sum32(cJ[R] * dv);
    const float want = caccv[s] + (cbv[s] - jv);
    const float nacc = friction ? __builtin_amdgcn_fmed3f(want, -lim, lim) : fmaxf(want, 0.f);
    const float dl = nacc - caccv[s];
    const float delta = rdl(dl, o), res = rdl(dl * cdgv[s], o);
    caccv[s] = wrl<o>(caccv[s], rdl(nacc, o));
    dv += cR[R] * delta;
    maxres = fmaxf(maxres, res * res);
  };
  for (int it = 0; it < sc.iters; it++) {
    maxres = 0.f;
    static_for<0, NLM>([&](auto jc) {  // motor rows, DoF by DoF = link by link (oracle order); lane j does the arithmetic
      constexpr int j = decltype(jc)::value;
      if ((jmotor >> j) & 1ull) {
        const float nacc = __builtin_amdgcn_fmed3f(macc + (mb - dv) * lrd, -mlim, mlim), dl = nacc - macc;
        const float delta = rdl(dl, j), res = rdl(dl * ldg, j);
        macc = wrl<j>(macc, rdl(nacc, j));
        dv += lRd[j] * delta;
        maxres = fmaxf(maxres, res * res);
      }
    });
    prof.stamp(PS_PGS_MOTOR);
    // joint-limit rows this env has active (the flags cannot change during the sweeps): lower, then upper, joint by joint
    if (jlo | jhi) static_for<0, NLM>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (((jlo | jhi) >> j) & 1ull) {
        if ((jlo >> j) & 1ull) {
          const float nacc = fmaxf(la0 + (lb0 - dv) * lrd, 0.f), dl = nacc - la0;
          const float delta = rdl(dl, j), res = rdl(dl * ldg, j);
          la0 = wrl<j>(la0, rdl(nacc, j)); dv += lRd[j] * delta; maxres = fmaxf(maxres, res * res);
        }
        if ((jhi >> j) & 1ull) {
          const float nacc = fmaxf(la1 + (lb1 + dv) * lrd, 0.f), dl = nacc - la1;
          const float delta = rdl(dl, j), res = rdl(dl * ldg, j);
          la1 = wrl<j>(la1, rdl(nacc, j)); dv -= lRd[j] * delta; maxres = fmaxf(maxres, res * res);
        }
      }
    });
    prof.stamp(PS_PGS_LIMIT);
    // contact normals, then the friction pairs (compile-time rows: they are registers)
    static_for<0, CM>([&](auto cc) { constexpr int C = decltype(cc)::value; if (C < ncont) contact_row(std::integral_constant<int, 3 * C>{}, 0.f, false); });
    static_for<0, CM>([&](auto cc) {
      constexpr int C = decltype(cc)::value, on = 16 + ((3 * C) & 15), sn = (3 * C) >> 4;
      if (C < ncont) {
        const float lim = rdl(cmuv[sn] * caccv[sn], on);  // mu x the normal impulse just solved
        contact_row(std::integral_constant<int, 3 * C + 1>{}, lim, true); contact_row(std::integral_constant<int, 3 * C + 2>{}, lim, true);
      }
    });
    prof.stamp(PS_PGS_CONTACT);
    iters_done = it + 1;
    if (maxres <= thr) break;
  }
  if (lane < nt) lq.L(sc.dv_base + lane) = dv;
  if (has_motor) lq.L(mmo + MR_ACC) = macc;  // motor impulses for the applied-torque readout
#pragma unroll
  for (int s = 0; s < NSL; s++) { const int r = (lane - 16) + 16 * s; if (lane >= 16 && lane < 32 && r < 3 * ncont) lq.L(sc.tr_off + r * rs + 2 * nt + 1) = caccv[s]; }  // contact impulses (force/torque sensor)
  return iters_done;
}

template <int LANES, int NTB, bool PROF>
DGD int pgs_dense_sliced_global(const Lane<LANES>& ln, float* accl, float* gws, int ncont_primary, int wave_max_cont, uint64_t limit_rows, Prof<PROF>& prof) {
  static_assert(LANES == -16, "global-workspace sliced sweeps run 16 envs per wavefront");
  constexpr int EPW = 16, SL = 4, LOG = 2, NS = NTB / SL, D = 4;
  const DevScene& sc = ln.sc; const int nt = sc.nt, rs = sc.crow_tail + 3, maxc = sc.max_contacts; const float h = sc.h;
  const float thr = sc.HF[DG_HF_RESIDUAL_THRESHOLD]; constexpr unsigned W = EPW;  // slot stride of the [workgroup][slot][lane] workspace
  const int lane = threadIdx.x, sl = lane & (SL - 1), q = lane >> LOG;
  const int envq = blockIdx.x * EPW + q; const bool validq = envq < sc.num_envs; const int eq = validq ? envq : sc.num_envs - 1;
  // Workspace addressing: raw buffer loads -- resource (this workgroup's block) in SGPRs, uniform slot offset in the
  // scalar offset operand, per-lane byte offset in one VGPR: no per-load address arithmetic on the vector ALU.
  float* const blk = gws + (size_t)blockIdx.x * (size_t)sc.total_slots * EPW;  // this workgroup's block (uniform)
  const unsigned colq = (unsigned)q;                                // env q's column inside the block
  const unsigned lane_off = ((unsigned)sl * W + colq) * 4u;         // byte offset of this lane's slice of env q's vectors
  const unsigned col_off = colq * 4u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(blk, 0, (int)((unsigned)sc.total_slots * W * 4u), 0x00020000);
  auto BL = [&](unsigned voff, int slot) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (unsigned)slot * W * 4u, 0)); };
  auto G = [&](int slot) -> float { return BL(col_off, slot); };
  const Lane<LANES> lq(sc, ln.mt, ln.lds, ln.st - ln.env + eq, eq, validq);  // tables only
  float* const acc = accl + q;  // acc[id * EPW]
  const int n_acc = 3 * maxc + 3 * sc.nl;
  const int ncont = __shfl(ncont_primary, q), rsw = sc.crow_tail + 3;
  // (contact rows start from the impulse their builder left in the row -- warm start -- the others from zero)
  // (per-lane slots: the whole offset goes through the vector operand)
  for (int id = sl; id < n_acc; id += SL) acc[id * EPW] = (sc.warm_off >= 0 && id < 3 * ncont) ? BL(col_off + (unsigned)(sc.tr_off + id * rsw + 2 * sc.nt + 1) * W * 4u, 0) : 0.f;
  for (int gl = sl; gl < sc.nl; gl += SL) {
    const unsigned mo = (unsigned)sc.PLL[gl * PLL_STRIDE + PLL_MROW];
    acc[(3 * maxc + gl) * EPW] = BL(col_off + (mo + MR_ACC) * W * 4u, 0);  // motor rows: motor_guess (zero without it)
    acc[(3 * maxc + sc.nl + 2 * gl) * EPW] = fmaxf(BL(col_off + (mo + MR_LO_ACC) * W * 4u, 0), 0.f);  // limit rows of pinned joints: limit_guess
    acc[(3 * maxc + sc.nl + 2 * gl + 1) * EPW] = fmaxf(BL(col_off + (mo + MR_HI_ACC) * W * 4u, 0), 0.f);
  }
  float dv[NS];
#pragma unroll
  for (int i = 0; i < NS; i++) dv[i] = 0.f;
  auto load_vec = [&](float (&v)[NS], int o) {
#pragma unroll
    for (int i = 0; i < NS; i++) v[i] = BL(lane_off + (unsigned)(i * SL) * W * 4u, o);  // constant part folds into the instruction offset
#pragma unroll
    for (int i = (NTB - 8) / SL; i < NS; i++) v[i] = (i * SL + sl) < nt ? v[i] : 0.f;
  };
  auto load_col = [&](float (&v)[NS], int col, int base, int nv) {
    // the body's DoFs are [base, base + nv) of the global vector: read the column as if it started `base` slots
    // earlier (allocated workspace) and clear what is not the body's
#pragma unroll
    for (int i = 0; i < NS; i++) v[i] = BL(lane_off + (unsigned)(i * SL) * W * 4u, col - base);
#pragma unroll
    for (int i = 0; i < NS; i++) { const int k = i * SL + sl; v[i] = (k >= base && k < base + nv) ? v[i] : 0.f; }
  };
  auto dv_at = [&](int j) { const float mine = dv[j >> LOG]; return group_sum4((j & (SL - 1)) == sl ? mine : 0.f); };
  struct Row { float J[NS], R[NS], b, diag, mu, acc, nacc; };
  struct Col { float R[NS], b, diag, lim, acc; int id, j; };
  auto load_row = [&](Row& r, int ro, int co, int id) { r.acc = acc[id * EPW]; r.nacc = acc[(id - id % 3) * EPW] /* its contact's normal impulse */; load_vec(r.J, ro); load_vec(r.R, ro + nt); r.b = G(ro + 2 * nt); r.diag = G(ro + 2 * nt + 2); r.mu = G(co + CL_MU); };
  float maxres = 0.f; bool live = validq;
  auto solve_row = [&](const Row& r, int id, float lo, float hi) {
    float jp = 0.f;
#pragma unroll
    for (int i = 0; i < NS; i++) jp += r.J[i] * dv[i];
    const float jv = group_sum4(jp), a0 = r.acc;
    float delta = fdiv(r.b - jv, r.diag);
    const float nacc = fminf(fmaxf(a0 + delta, lo), hi);
    delta = live && r.diag > 1e-18f ? nacc - a0 : 0.f;
    acc[id * EPW] = a0 + delta;  // every lane of the group stores the same value
#pragma unroll
    for (int i = 0; i < NS; i++) dv[i] += r.R[i] * delta;
    const float res = delta * r.diag; maxres = fmaxf(maxres, res * res);
  };
  const LinkRows<LANES, true> rows(lq);
  auto load_motor = [&](Col& r, int gl) {
    int col, base, nv, mo; rows.get(gl, col, mo, r.j, base, nv, r.lim); r.id = 3 * maxc + gl;
    r.acc = acc[r.id * EPW]; load_col(r.R, col, base, nv); r.b = G(mo + MR_B); r.diag = G(col + r.j - base);
  };
  auto solve_motor = [&](const Col& r) {
    const float a0 = r.acc;
    float delta = fdiv(r.b - dv_at(r.j), r.diag);
    const float nacc = fminf(fmaxf(a0 + delta, -r.lim), r.lim);
    delta = live ? nacc - a0 : 0.f; acc[r.id * EPW] = a0 + delta;
#pragma unroll
    for (int i = 0; i < NS; i++) dv[i] += r.R[i] * delta;
    const float res = delta * r.diag; maxres = fmaxf(maxres, res * res);
  };
  const int r0 = sc.tr_off, c0 = sc.cont_off + 1;
  if (wave_max_cont > 0 && sc.warm_off >= 0) {  // warm start: the velocity change the rows' starting impulses amount to
    for (int r = 0; r < 3 * wave_max_cont; r++) {
      Row W; load_row(W, r0 + r * rs, c0 + (r / 3) * CL_STRIDE, r); const bool on = r < 3 * ncont;
#pragma unroll
      for (int i = 0; i < NS; i++) dv[i] += on ? W.R[i] * W.acc : 0.f;
    }
  }
  for (uint64_t m = rows.motors; m; m &= m - 1) {  // ... and the motor rows' (motor_guess; limit rows of pinned joints: limit_guess)
    const int gl = __ffsll((long long)m) - 1; Col W; load_motor(W, gl);
    const float tot = W.acc + acc[(3 * maxc + sc.nl + 2 * gl) * EPW] - acc[(3 * maxc + sc.nl + 2 * gl + 1) * EPW];
#pragma unroll
    for (int i = 0; i < NS; i++) dv[i] += W.R[i] * tot;
  }
  int iters_done = 0;
  for (int it = 0; it < sc.iters; it++) {
    maxres = 0.f;
    if (rows.motors) {  // loads are unconditional (a row past the end re-reads the last one) so that the
      Col A, B;         // compiler can count the loads in flight instead of draining them at every branch
      uint64_t m = rows.motors; int g = __ffsll((long long)m) - 1;
      load_motor(A, g);
      while (true) {
        m &= m - 1; const bool more1 = m != 0; g = more1 ? __ffsll((long long)m) - 1 : g; load_motor(B, g);
        solve_motor(A);
        if (!more1) break;
        m &= m - 1; const bool more2 = m != 0; g = more2 ? __ffsll((long long)m) - 1 : g; load_motor(A, g);
        solve_motor(B);
        if (!more2) break;
      }
    }
    prof.stamp(PS_PGS_MOTOR);
    for (uint64_t m = limit_rows; m; m &= m - 1) {
      const int bit = __ffsll((long long)m) - 1, gl = bit >> 1, side = bit & 1, id = 3 * maxc + sc.nl + bit;
      int col, mo, jg, base, nv; float lim_unused; rows.get(gl, col, mo, jg, base, nv, lim_unused);
      const int jb = jg - base, bo = mo + (side == 0 ? MR_LO_B : MR_HI_B); const float sg = side == 0 ? 1.f : -1.f;
      float R[NS]; load_col(R, col, base, nv);
      const float diag = G(col + jb), bb = G(bo); const bool act = G(bo + 1) >= 0.f;  // the flag is not rewritten here
      const float a0 = acc[id * EPW];
      float delta = fdiv(bb - sg * dv_at(base + jb), diag);
      const float nacc = fmaxf(a0 + delta, 0.f);
      delta = (live && act) ? nacc - a0 : 0.f; acc[id * EPW] = a0 + delta;
      const float sd = sg * delta;
#pragma unroll
      for (int i = 0; i < NS; i++) dv[i] += R[i] * sd;
      const float res = delta * diag; maxres = fmaxf(maxres, res * res);
    }
    prof.stamp(PS_PGS_LIMIT);
    if (wave_max_cont > 0) {
      Row buf[D];
      {  // normals: row t = contact t
        const int n = wave_max_cont;
#pragma unroll
        for (int d = 0; d < D - 1; d++) { const int tc = min(d, n - 1); load_row(buf[d], r0 + 3 * tc * rs, c0 + tc * CL_STRIDE, 3 * tc); }
        for (int t0 = 0; t0 < n; t0 += D) {
#pragma unroll
          for (int d = 0; d < D; d++) {
            const int t = t0 + d, tp = t + D - 1;
            { const int tc = min(tp, n - 1); load_row(buf[(d + D - 1) % D], r0 + 3 * tc * rs, c0 + tc * CL_STRIDE, 3 * tc); }
            if (t < n && t < ncont) solve_row(buf[d], 3 * t, 0.f, 3.0e38f);
          }
        }
      }
      {  // friction: row t = (contact t / 2, direction 1 + t % 2)
        const int n = 2 * wave_max_cont;
#pragma unroll
        for (int d = 0; d < D - 1; d++) { const int tc = min(d, n - 1); load_row(buf[d], r0 + (3 * (tc >> 1) + 1 + (tc & 1)) * rs, c0 + (tc >> 1) * CL_STRIDE, 3 * (tc >> 1) + 1 + (tc & 1)); }
        for (int t0 = 0; t0 < n; t0 += D) {
#pragma unroll
          for (int d = 0; d < D; d++) {
            const int t = t0 + d, tp = t + D - 1;
            { const int tc = min(tp, n - 1); load_row(buf[(d + D - 1) % D], r0 + (3 * (tc >> 1) + 1 + (tc & 1)) * rs, c0 + (tc >> 1) * CL_STRIDE, 3 * (tc >> 1) + 1 + (tc & 1)); }
            if (t < n && (t >> 1) < ncont) {
              const float mu = buf[d].mu;
              if (mu > 0.f) { const float lim = mu * buf[d].nacc; solve_row(buf[d], 3 * (t >> 1) + 1 + (t & 1), -lim, lim); }
            }
          }
        }
      }
    }
    prof.stamp(PS_PGS_CONTACT);
    if (live) iters_done = it + 1;
    live = live && !(maxres <= thr);
    if (!__any(live)) break;
  }
#pragma unroll
  for (int i = 0; i < NS; i++) if (i * SL + sl < nt) blk[(unsigned)(sc.dv_base + i * SL) * W + (lane_off >> 2)] = dv[i];  // lane_off is in bytes
  // contact impulses back into their rows (force/torque sensor), motor impulses for the applied-torque readout
  if (sl == 0) for (int c = 0; c < ncont; c++) { _Pragma("unroll") for (int d = 0; d < 3; d++) blk[(unsigned)(r0 + (3 * c + d) * rs + 2 * nt + 1) * W + colq] = acc[(3 * c + d) * EPW]; }
  for (uint64_t m = rows.motors; m; m &= m - 1) {
    const int gl = __ffsll((long long)m) - 1; int col, mo, jg, base, nv; float lm; rows.get(gl, col, mo, jg, base, nv, lm);
    blk[(unsigned)(mo + MR_ACC) * W + colq] = acc[(3 * maxc + gl) * EPW];
  }
  return __shfl(iters_done, (lane << LOG) & 63);  // primary lane e reads the count of env e's group
}

// ---- motor and joint-limit rows of the links [g0, g1) ---------------------------------------------------------------
// Links in chunks: all the state loads of a chunk are issued before the first LDS store (a link at a time would pay
// one global round trip per link).
template <int LANES>
DGD void setup_link_rows(const Lane<LANES>& ln, int g0, int g1, uint64_t& limit_mask, uint64_t& limit_rows) {
  const DevScene& sc = ln.sc; const float h = sc.h, lerp = sc.HF[DG_HF_LIMIT_ERP];
  constexpr int LCH = 6;
  for (int gc = g0; gc < g1; gc += LCH) {
    float q_[LCH], qd_[LCH], tp_[LCH], tv_[LCH];
#pragma unroll
    for (int j = 0; j < LCH; j++) {
      const int gl = min(gc + j, g1 - 1), lo = ln.li(gl)[DG_LI_STATE_OFF];
      q_[j] = ln.S(lo + DG_LS_Q); qd_[j] = ln.S(lo + DG_LS_QD); tp_[j] = ln.S(lo + DG_LS_TARGET_POS); tv_[j] = ln.S(lo + DG_LS_TARGET_VEL);
    }
#pragma unroll
    for (int j = 0; j < LCH; j++) {
      const int gl = gc + j; if (gl >= g1) break;
      const int mo = ln.pll(gl)[PLL_MROW]; cfp f = ln.lf(gl);
      const float q = q_[j], qd = qd_[j];
      const float kp = ln.mt.v[3 * gl], kd = ln.mt.v[3 * gl + 1];
      ln.L(mo + MR_B) = kp * (tp_[j] - q) / h + kd * (tv_[j] - qd);
      ln.L(mo + MR_ACC) = 0.f;
      const bool limited = f[DG_LF_LOWER] <= f[DG_LF_UPPER];
      const float dlo = q - f[DG_LF_LOWER], dhi = f[DG_LF_UPPER] - q;
      // acc < 0 marks an inactive limit row
      ln.L(mo + MR_LO_B) = -qd + (dlo > 0.f ? -dlo / h : -dlo * lerp / h); ln.L(mo + MR_LO_ACC) = (limited && dlo < 0.25f) ? 0.f : -1.f;
      ln.L(mo + MR_HI_B) = qd + (dhi > 0.f ? -dhi / h : -dhi * lerp / h); ln.L(mo + MR_HI_ACC) = (limited && dhi < 0.25f) ? 0.f : -1.f;
      if (__any(limited && (dlo < 0.25f || dhi < 0.25f))) limit_mask |= 1ull << (ln.li(gl)[DG_LI_BODY] & 63);
      if (gl < 32) { if (__any(limited && dlo < 0.25f)) limit_rows |= 1ull << (2 * gl); if (__any(limited && dhi < 0.25f)) limit_rows |= 2ull << (2 * gl); }
    }
  }
}

// ---- apply the velocity change of body b and integrate its positions ---------------------------------------------------
template <int LANES>
DGD void integrate_body(const Lane<LANES>& ln, int b) {
  const DevScene& sc = ln.sc; const float h = sc.h, vmax = sc.HF[DG_HF_MAX_COORD_VEL];
  constexpr int LCH = 6;
  cip B = ln.bi(b); const int n = B[DG_BI_N_LINKS], first = B[DG_BI_FIRST_LINK], so = B[DG_BI_STATE_OFF];
  const bool fx = ln.fixed(b); if (fx && n == 0) return;
  const int dvo = ln.plb(b)[PLB_DV];
  if (!fx) {
    M3 R0 = ln.LR(ln.plb(b)[PLB_R0]);
    V3 dw = mul(R0, ln.L3(dvo)), dl = mul(R0, ln.L3(dvo + 3));
    V3 w = v3(ln.S(so + DG_BS_ANGVEL), ln.S(so + DG_BS_ANGVEL + 1), ln.S(so + DG_BS_ANGVEL + 2)) + dw;
    V3 v = v3(ln.S(so + DG_BS_LINVEL), ln.S(so + DG_BS_LINVEL + 1), ln.S(so + DG_BS_LINVEL + 2)) + dl;
    ln.Sset(so + DG_BS_ANGVEL, w.x); ln.Sset(so + DG_BS_ANGVEL + 1, w.y); ln.Sset(so + DG_BS_ANGVEL + 2, w.z);
    ln.Sset(so + DG_BS_LINVEL, v.x); ln.Sset(so + DG_BS_LINVEL + 1, v.y); ln.Sset(so + DG_BS_LINVEL + 2, v.z);
    ln.Sset(so, ln.S(so) + h * v.x); ln.Sset(so + 1, ln.S(so + 1) + h * v.y); ln.Sset(so + 2, ln.S(so + 2) + h * v.z);
    float wn = norm(w), th = wn * h; Q4 dq;
    if (th > 1e-12f) { float sn, cs; sincosf(0.5f * th, &sn, &cs); sn /= wn; dq.x = w.x * sn; dq.y = w.y * sn; dq.z = w.z * sn; dq.w = cs; }
    else { dq.x = 0.5f * h * w.x; dq.y = 0.5f * h * w.y; dq.z = 0.5f * h * w.z; dq.w = 1.f; }
    Q4 qn = qnormalize(qmul(dq, ln.base_quat(b)));
    ln.Sset(so + 3, qn.x); ln.Sset(so + 4, qn.y); ln.Sset(so + 5, qn.z); ln.Sset(so + 6, qn.w);
  }
  const int k0 = fx ? 0 : 6;
  for (int i0 = 0; i0 < n; i0 += LCH) {  // loads of a chunk first, then its stores (state loads cannot pass state stores)
    float q_[LCH], qd_[LCH], dv_[LCH], ac_[LCH]; int lo_[LCH];
#pragma unroll
    for (int j = 0; j < LCH; j++) {
      const int i = min(i0 + j, n - 1); lo_[j] = ln.li(first + i)[DG_LI_STATE_OFF];
      q_[j] = ln.S(lo_[j] + DG_LS_Q); qd_[j] = ln.S(lo_[j] + DG_LS_QD); dv_[j] = ln.L(dvo + k0 + i); ac_[j] = ln.L(ln.pll(first + i)[PLL_MROW] + MR_ACC);
    }
#pragma unroll
    for (int j = 0; j < LCH; j++) {
      const int i = i0 + j; if (i >= n) break;
      const float maxf = ln.mt.v[3 * (first + i) + 2]; const float maximp = maxf < 0.f ? -maxf : maxf * sc.hm;
      ln.Sset(lo_[j] + DG_LS_APPLIED, maximp > 0.f ? ac_[j] / h : 0.f);
      const float qd = fminf(fmaxf(qd_[j] + dv_[j], -vmax), vmax);
      ln.Sset(lo_[j] + DG_LS_QD, qd); ln.Sset(lo_[j] + DG_LS_Q, q_[j] + h * qd);
    }
  }
}

// contacts per env the arm-per-half-wavefront sweeps can carry in registers (3 rows of 6 + 6 floats each per lane);
// substeps with more contacts in some env of the workgroup take the single-wave streamed sweeps
#define DG_SPLIT_MAX_CONTACTS 3
// Six workspace slots for the residual exchange (4) and the two decision flags: the tail of the padding behind the
// velocity-change blocks (nv_max + 8 slots; readers of that padding only ever multiply it by zero, and these values
// are finite).
DGD int split_slots(const DevScene& sc) { return sc.dv_base + sc.nt + sc.nv_max; }

// Per-substep decision of the helper-wave kernel: are this substep's sweeps split?  The main wave publishes "no
// contact and no active limit row on my bodies", the helper "no active limit row on my body" (it has just set up its
// own rows); after one barrier every wavefront of the workgroup reads both and reaches the same verdict.
template <int LANES>
DGD bool split_decide_main(const Lane<LANES>& ln, int wave_max_cont, uint64_t& limit_mask, uint64_t& limit_rows) {
  const DevScene& sc = ln.sc; const int xo = split_slots(sc);
  // split unless some env has more contacts than the register sweeps carry (limit rows are swept in registers too);
  // sc.split_pgs == 2: the dense DoF vector is exactly the two arms, so contact rows can be split by arm as well
  ln.L(xo + 4) = (sc.split_pgs && (wave_max_cont == 0 || (sc.split_pgs == 2 && wave_max_cont <= DG_SPLIT_MAX_CONTACTS))) ? 1.f : 0.f;
  __syncthreads();  // Bq
  // the helper's active limit rows (2 bits per joint of its body, as a small integer), for the single-wave sweeps
  // that run when contacts couple the bodies
  const unsigned hbits = (unsigned)ln.L(xo + 5);
  if (hbits) { limit_mask |= 1ull << (sc.helper_body & 63); limit_rows |= (uint64_t)hbits << (2 * ln.bi(sc.helper_body)[DG_BI_FIRST_LINK]); }
  return ln.L(xo + 4) != 0.f;
}
template <int LANES>
DGD bool split_decide_follow(const Lane<LANES>& ln, unsigned my_limit_bits, bool i_am_helper) {
  const DevScene& sc = ln.sc; const int xo = split_slots(sc);
  if (i_am_helper) ln.L(xo + 5) = (float)my_limit_bits;  // <= 12 bits: exact
  __syncthreads();  // Bq
  return ln.L(xo + 4) != 0.f;
}

// ---- register-chain sweeps of the helper-wave kernel, one (env, arm) per lane ------------------------------------
// When a substep has no contact, the rows of the two register-chain bodies share no unknown; the only coupling is
// the per-env residual that decides the early-out.  Wavefront w (0 = main, 1 = helper) sweeps BOTH arms of the envs
// [32 w, 32 w + 32) of the workgroup: lane l holds arm (l >> 5) of env 32 w + (l & 31), so the two residuals of an env
// sit in lanes l and l ^ 32 of the same wavefront and are combined with one v_permlane32_swap -- no LDS round trip,
// no workgroup barrier inside the loop (the previous form, one arm per wavefront, exchanged the residual through LDS
// and a __syncthreads every iteration: ~400 of its ~860 cycles).  Each wavefront leaves the loop as soon as ITS 32
// envs have converged.  M^-1, right-hand sides and limit flags are read from where the per-arm passes left them in
// LDS (per-lane slot offsets); velocity changes and accumulated impulses go back the same way, and the caller's
// __syncthreads hands them to the wavefront that integrates the arm.
// Bitwise the same impulses as the single-wave sweep: rows of different bodies were already independent chains.
DGD float half_swap_max(float x) {  // max over lanes l and l ^ 32, in every lane
  // v_permlane32_swap a, b exchanges lanes 32..63 of a with lanes 0..31 of b: with a = b = x on entry, a ends up as
  // x[l & 31] and b as x[32 + (l & 31)] in every lane.  Inline assembly because hipcc 7.2 folds the two results of
  // __builtin_amdgcn_permlane32_swap(x, x) into one register (it emits max(a, a)); the s_nop covers the VALU-write ->
  // permlane hazard the compiler pads for its own builtin.
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
DGD float half_swap_sum(float x) {  // x[l & 31] + x[32 + (l & 31)] in every lane (same operand order in both halves)
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
template <int LANES>
DGD void pgs_reg_halves(const Lane<LANES>& ln, int wave) {
  constexpr int RN = 6;
  const DevScene& sc = ln.sc; const float h = sc.h; const float thr_abs = sqrtf(sc.HF[DG_HF_RESIDUAL_THRESHOLD]);
  const int lane = threadIdx.x & 63, half = lane >> 5, el = (lane & 31) + 32 * wave;
  float* const col = ln.lds - lane + el;  // this lane's ENV column of the workspace
  const bool valid = (int)blockIdx.x * 64 + el < sc.num_envs;
  const int b0 = sc.reg_body[0], b1 = sc.helper_body;
  const int f0 = ln.bi(b0)[DG_BI_FIRST_LINK], f1 = ln.bi(b1)[DG_BI_FIRST_LINK], n0 = ln.bi(b0)[DG_BI_N_LINKS], n1 = ln.bi(b1)[DG_BI_N_LINKS];
  const int n = half ? n1 : n0, mvo = half ? ln.plb(b1)[PLB_MINV] : ln.plb(b0)[PLB_MINV], dvo = half ? ln.plb(b1)[PLB_DV] : ln.plb(b0)[PLB_DV];
  const int mo0 = half ? ln.pll(f1)[PLL_MROW] : ln.pll(f0)[PLL_MROW];
  auto W = [&](int slot) -> float& { return col[slot * 64]; };
  float rM[RN * RN], rdv[RN], rb[RN], racc[RN], rdi[RN], rdg[RN], smax[RN], lb[2][RN], la[2][RN]; bool any_limit = false;
  if (n0 == RN && n1 == RN) {
    // both arms have all six joints (every pair of 6-axis arms): no per-lane row count, so every slot is a compile-time offset
    // from two per-lane bases (the arm's M^-1 block, its first motor / limit row block) -- one ds_read per value, no address
    // arithmetic, no masks (the general form below spends ~560 instructions on these 72 values)
    const float* const cm = col + (size_t)mvo * 64; const float* const cr = col + (size_t)mo0 * 64;
#pragma unroll
    for (int i = 0; i < RN; i++) {
      const float mf0 = ln.mt.v[3 * (f0 + i) + 2], mf1 = ln.mt.v[3 * (f1 + i) + 2], maxf = half ? mf1 : mf0;
      smax[i] = maxf < 0.f ? -maxf : maxf * sc.hm;
#pragma unroll
      for (int c = 0; c < RN; c++) rM[i * RN + c] = cm[(i * RN + c) * 64];
      rb[i] = cr[(i * MR_STRIDE + MR_B) * 64]; rdg[i] = rM[i * RN + i]; rdi[i] = 1.0f / rdg[i];
      lb[0][i] = cr[(i * MR_STRIDE + MR_LO_B) * 64]; la[0][i] = cr[(i * MR_STRIDE + MR_LO_ACC) * 64];
      lb[1][i] = cr[(i * MR_STRIDE + MR_HI_B) * 64]; la[1][i] = cr[(i * MR_STRIDE + MR_HI_ACC) * 64];
      rdv[i] = 0.f; racc[i] = 0.f;
    }
  } else {
#pragma unroll
  for (int i = 0; i < RN; i++) {
    const bool has = i < n; const int ic = has ? i : 0;  // absent rows: clamped address, zeroed value
    const float mf0 = ln.mt.v[3 * (f0 + (i < n0 ? i : 0)) + 2], mf1 = ln.mt.v[3 * (f1 + (i < n1 ? i : 0)) + 2];
    const float maxf = half ? mf1 : mf0;
    smax[i] = has ? (maxf < 0.f ? -maxf : maxf * sc.hm) : 0.f;
    const int mo = mo0 + ic * MR_STRIDE;
    rb[i] = has ? W(mo + MR_B) : 0.f; rdg[i] = has ? W(mvo + ic * n + ic) : 1.f; rdi[i] = has ? 1.0f / rdg[i] : 0.f;
    lb[0][i] = W(mo + MR_LO_B); la[0][i] = has ? W(mo + MR_LO_ACC) : -1.f;
    lb[1][i] = W(mo + MR_HI_B); la[1][i] = has ? W(mo + MR_HI_ACC) : -1.f;
    any_limit = any_limit || la[0][i] >= 0.f || la[1][i] >= 0.f;
    rdv[i] = 0.f; racc[i] = 0.f;
#pragma unroll
    for (int c = 0; c < RN; c++) { const bool hc = has && c < n; const float m = W(mvo + ic * n + (c < n ? c : 0)); rM[i * RN + c] = hc ? m : 0.f; }
  }
  }
  if (sc.HF[DG_HF_MOTOR_GUESS] > 0.f) chain_motor_guess(rM, rb, smax, racc, rdv, limit_ptol(sc), lb[0], la[0], lb[1], la[1]);  // the sweeps start next to their fixed point (limit rows of pinned joints included)
  // limit rows some lane of the wavefront has active (the flags cannot change during the sweeps): bit 2 i + side.  Rows
  // nobody needs are skipped with a wave-uniform branch, so a sweep costs what the wavefront's active limits cost
  unsigned lim_rows = 0u;
#pragma unroll
  for (int i = 0; i < RN; i++) { if (__any(la[0][i] >= 0.f)) lim_rows |= 1u << (2 * i); if (__any(la[1][i] >= 0.f)) lim_rows |= 2u << (2 * i); }
  (void)any_limit;
  // Contact rows (arms touching each other): the dense rows the main wave built are [J nt][R nt][b acc diag] over the
  // global DoF vector = (first arm, second arm), so this lane's share of a row is the 6 + 6 entries of ITS arm; the
  // row's J . dv is the sum over the two halves (one v_permlane32_swap), and both lanes apply the same impulse to
  // their own arm.  Same row order as every other path: normals of all contacts, then the friction pairs.
  constexpr int CM = DG_SPLIT_MAX_CONTACTS;
  const int ncont = (int)W(sc.cont_off), g = (half ? ln.plb(b1)[PLB_DV] : ln.plb(b0)[PLB_DV]) - sc.dv_base;
  int cmax = 0;
#pragma unroll
  for (int c = 0; c < CM; c++) if (__any(c < ncont)) cmax = c + 1;
  float cJ[3 * CM][RN], cR[3 * CM][RN], cb[3 * CM], cdg[3 * CM], cdi[3 * CM], cacc[3 * CM], cmu[CM];
  if (cmax > 0) {
    const int nt = sc.nt, rs = sc.crow_tail + 3;
#pragma unroll
    for (int c = 0; c < CM; c++) {
      const bool has = c < ncont; const int cc = has ? c : 0;  // lanes without this contact read contact 0's slots and zero them
      cmu[c] = has ? W(sc.cont_off + 1 + cc * CL_STRIDE + CL_MU) : 0.f;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int r = 3 * c + d, ro = sc.tr_off + (3 * cc + d) * rs;
#pragma unroll
        for (int k = 0; k < RN; k++) { const bool hk = has && k < n; const int kk = k < n ? k : 0; const float j = W(ro + g + kk), rr = W(ro + nt + g + kk); cJ[r][k] = hk ? j : 0.f; cR[r][k] = hk ? rr : 0.f; }
        const float bb = W(ro + 2 * nt), dd = W(ro + 2 * nt + 2);
        cb[r] = has ? bb : 0.f; cdg[r] = has ? dd : 1.f; cdi[r] = (has && dd > 1e-18f) ? frcp(dd) : 0.f;
        const float a0 = W(ro + 2 * nt + 1); cacc[r] = (has && cdi[r] != 0.f) ? a0 : 0.f;  // (starting impulse: warm start)
#pragma unroll
        for (int k = 0; k < RN; k++) rdv[k] += cR[r][k] * cacc[r];
      }
    }
  }
  bool live = valid; int iters_done = 0;
  auto contact_row = [&](int r, float lo, float hi, float lv, float& maxabs) {
    float jp = 0.f;
#pragma unroll
    for (int k = 0; k < RN; k++) jp += cJ[r][k] * rdv[k];
    const float jv = half_swap_sum(jp);
    const float nacc = fminf(fmaxf(cacc[r] + (cb[r] - jv) * cdi[r], lo), hi);
    const float delta = cdi[r] != 0.f ? (nacc - cacc[r]) * lv : 0.f; cacc[r] += delta;
#pragma unroll
    for (int k = 0; k < RN; k++) rdv[k] += cR[r][k] * delta;
    maxabs = fmaxf(maxabs, fabsf(delta * cdg[r]));
  };
  // one sweep over the six motor rows (straight-line), then the limit rows the wavefront needs, then the contact rows
  auto sweep = [&](auto with_limits, auto with_contacts) {
    float maxabs = 0.f; const float lv = live ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < RN; i++) {
      const float want = racc[i] + (rb[i] - rdv[i]) * rdi[i];
      const float nacc = __builtin_amdgcn_fmed3f(want, -smax[i], smax[i]);
      const float delta = (nacc - racc[i]) * lv; racc[i] += delta;
#pragma unroll
      for (int c = 0; c < RN; c++) rdv[c] += rM[i * RN + c] * delta;
      maxabs = fmaxf(maxabs, fabsf(delta * rdg[i]));
    }
    if constexpr (decltype(with_limits)::value) {  // after the body's motor rows, as in the single-wave order
#pragma unroll
      for (int i = 0; i < RN; i++) {
#pragma unroll
        for (int side = 0; side < 2; side++) {
          if (!((lim_rows >> (2 * i + side)) & 1u)) continue;
          const float sg = side == 0 ? 1.f : -1.f; const bool act = la[side][i] >= 0.f;
          const float nacc = fmaxf(la[side][i] + (lb[side][i] - sg * rdv[i]) * rdi[i], 0.f);
          const float delta = act ? (nacc - la[side][i]) * lv : 0.f; la[side][i] += delta;
          const float sd = sg * delta;
#pragma unroll
          for (int c = 0; c < RN; c++) rdv[c] += rM[i * RN + c] * sd;
          maxabs = fmaxf(maxabs, fabsf(delta * rdg[i]));
        }
      }
    }
    if constexpr (decltype(with_contacts)::value) {
#pragma unroll
      for (int c = 0; c < CM; c++) if (c < cmax) contact_row(3 * c, 0.f, 3.0e38f, lv, maxabs);
#pragma unroll
      for (int c = 0; c < CM; c++) if (c < cmax) {
        const float lim = cmu[c] * cacc[3 * c];  // mu = 0 (or no such contact in this lane): the pair stays at zero
        contact_row(3 * c + 1, -lim, lim, lv, maxabs); contact_row(3 * c + 2, -lim, lim, lv, maxabs);
      }
    }
    const float m = half_swap_max(maxabs);  // the env's residual over both arms
    live = live && !(m <= thr_abs);
  };
  // three copies of the loop: the common one (no lane near a joint limit, no contact) carries neither kind of code
  if (cmax > 0) {
    for (int it = 0; it < sc.iters; it++) { if (live) iters_done = it + 1; sweep(std::true_type{}, std::true_type{}); if (!__any(live)) break; }
  } else if (lim_rows == 0u) {
    for (int it = 0; it < sc.iters; it++) { if (live) iters_done = it + 1; sweep(std::false_type{}, std::false_type{}); if (!__any(live)) break; }
  } else {
    for (int it = 0; it < sc.iters; it++) { if (live) iters_done = it + 1; sweep(std::true_type{}, std::false_type{}); if (!__any(live)) break; }
  }
#pragma unroll
  for (int i = 0; i < RN; i++) if (i < n) { W(dvo + i) = rdv[i]; W(mo0 + i * MR_STRIDE + MR_ACC) = racc[i]; }
  if (cmax > 0 && half == 0) {  // solved contact impulses back into their rows (the force/torque sensor reads them)
    const int rs = sc.crow_tail + 3;
#pragma unroll
    for (int c = 0; c < CM; c++) if (c < ncont) { _Pragma("unroll") for (int d = 0; d < 3; d++) W(sc.tr_off + (3 * c + d) * rs + sc.crow_tail + 1) = cacc[3 * c + d]; }
  }
  if (half == 0) W(split_slots(sc)) = (float)iters_done;  // for the diagnostics column, read by the main wave after the barrier
}

// ---------------------------------------------------------------- substep
// PAR: this wave is the MAIN wave of a two-wave workgroup; the helper wave (helper_substep below) owns body
// sc.helper_body -- its kinematics and its register-resident dynamics run concurrently with everything here up to
// the second barrier.  Both waves execute exactly three __syncthreads per substep.
// SLICED (16 / 32 envs per wavefront, step kernel only): all 64 lanes are alive; lanes >= LANES sit out everything
// except the dense Gauss-Seidel loop, where they take a share of each env's rows (pgs_dense_sliced).
// FULLWAVE: every lane of the wavefront is active at the call (step kernels of the 64-lane and global-workspace
// modes; not the reset kernel, which runs the step under a per-env mask).
template <int LANES, bool PROF, bool PAR = false, bool SLICED = false, bool FULLWAVE = false>
DGD void substep(const Lane<LANES>& ln, int32_t* diag_out, Prof<PROF>& prof, float* smem = nullptr, float* gws = nullptr, int index = 0) {
  const DevScene& sc = ln.sc; const float h = sc.h; const int hb = PAR ? sc.helper_body : -1;
  const bool primary = SLICED ? (int)threadIdx.x < envs_per_wave(LANES) : true;
  constexpr int LCH = 6;  // links per chunk in the per-link loops
  int ncont = 0, wave_max_cont = 0, iters_done = 0;
  uint64_t limit_mask = 0ull;  // bit (b & 63): some lane of this wave has an active limit row on body b
  uint64_t limit_rows = 0ull;  // bit (2 gl + side), links 0..31: some lane has that limit row active (dense sweeps)
  const float thr = sc.HF[DG_HF_RESIDUAL_THRESHOLD]; const int rs = crow_stride(sc.crow_tail);
  // early: the narrow-phase wavefront already ran this (first) substep's narrow phase and dynamics during the update
  // ops; positions have not changed since the poses of the step's start were computed
  const bool early = PAR && sc.early_dyn && index == 0;
  // lane-sliced modes: this lane's env in the grouping of the sweeps (lane = env * SL + slice), see pgs_dense_sliced
  constexpr int EPW = envs_per_wave(LANES), SLN = SLICED ? 64 / EPW : 1, SLOG = SLN == 64 ? 6 : SLN == 16 ? 4 : SLN == 8 ? 3 : SLN == 4 ? 2 : SLN == 2 ? 1 : 0;
  const int qlane = threadIdx.x & 63, qsl = qlane & (SLN - 1), qe = qlane >> SLOG;
  const int qenv = blockIdx.x * EPW + qe; const bool qvalid = qenv < sc.num_envs; const int qec = qvalid ? qenv : sc.num_envs - 1;
  const Lane<LANES> lq(sc, ln.mt, SLICED ? ln.lds - qlane + qe : ln.lds, SLICED ? ln.st - ln.env + qec : ln.st, SLICED ? qec : ln.env, SLICED ? qvalid : ln.valid);
  if (primary) {
  if (index == sc.substeps - 1 && !early) for (int b = 0; b < sc.nba; b++) if (b != hb) save_prev_velocities(ln, b);
  if (!early) for (int b = 0; b < sc.nba; b++) if (b != hb) ln.kinematics(b);
  }
  if (PAR) __syncthreads();  // B1: every pose is in LDS
  prof.stamp(PS_KIN);
  if constexpr (SLICED) collide<LANES, 64, SLN>(lq, 0, 0x7fffffff, -1, qsl);  // every lane: the env's group shares the pairs
  const bool own_collide = !(PAR && sc.coll_wave);  // else the third wavefront is doing it right now
  if (primary) {
  if constexpr (SLICED) ncont = (int)ln.L(sc.cont_off);
  else if (own_collide) ncont = collide<LANES, FULLWAVE ? 64 : (PAR ? 64 : 0)>(ln);
  prof.stamp(PS_COLLIDE);
  }  // primary
  if (!early) for (int b = 0; b < sc.nba; b++) {
    if (b == hb || (ln.fixed(b) && ln.bi(b)[DG_BI_N_LINKS] == 0)) continue;
    // lane-sliced modes: the M^-1 columns of a tree body are shared by the lanes of the env's group (minv_sliced)
    const int mslices = (SLICED && !ln.plb(b)[PLB_CHAIN] && !sc.no_minv_slices) ? ln.minv_slices(b, SLN) : 0;
    if (primary) { if (ln.plb(b)[PLB_CHAIN]) ln.template dynamics_chain<6>(b, prof); else ln.dynamics(b, prof, mslices > 0); }
    if constexpr (SLICED) { if (mslices > 0) { lq.minv_sliced(b, qsl, mslices, SLN); prof.stamp(4 /* PS_MINV */); } }
    if (primary) {
      const int dvo = ln.plb(b)[PLB_DV], nv = ln.plb(b)[PLB_NV];
      for (int k = 0; k < nv; k++) ln.L(dvo + k) = 0.f;
    }
  }
  if (primary) {
  if (PAR) __syncthreads();  // B2: the helper's joint velocities (state) and M^-1 (LDS) are in place
  if (!own_collide) {
    ncont = (int)ln.L(sc.cont_off);  // written by the narrow-phase wavefront(s) before B2
    if (sc.coll_split) {  // append the second wavefront's contacts (later pairs) behind the first's (the early first substep
                          // too: its narrow phase is cut in two like the others since round 4)
      const int nb2 = (int)ln.L(sc.cont2_off);
      if (__any(nb2 > 0)) {
        for (int j = 0; j < sc.max_contacts; j++) {
          if (!__any(j < nb2)) break;
          if (j < nb2 && ncont + j < sc.max_contacts) {
            float* dst = ln.lds + (sc.cont_off + 1 + (ncont + j) * CL_STRIDE) * envs_per_wave(LANES);  // per-lane destination entry
#pragma unroll
            for (int k = 0; k < CL_KEY + 1; k++) dst[k * envs_per_wave(LANES)] = ln.L(sc.cont2_off + 1 + j * CL_STRIDE + k);
          }
        }
        ncont = min(ncont + nb2, sc.max_contacts); ln.L(sc.cont_off) = (float)ncont;
      }
    }
  }
  // ---- motor and joint-limit rows (per link, uniform)
  // (the helper wave sets up the rows of its own body when the sweeps can be split)
  const bool helper_rows = PAR && sc.split_pgs;
  if (helper_rows) {
    const int hf = ln.bi(hb)[DG_BI_FIRST_LINK], hn = ln.bi(hb)[DG_BI_N_LINKS];
    setup_link_rows(ln, 0, hf, limit_mask, limit_rows); setup_link_rows(ln, hf + hn, sc.nl, limit_mask, limit_rows);
  } else setup_link_rows(ln, 0, sc.nl, limit_mask, limit_rows);
  // ---- starting impulses of the motor rows (motor_guess): bodies with more than six joints need the transient region,
  // which is free between the dynamics and the contact rows; register-chain bodies are done where their rows end up
  // (in registers, or below once it is known that this substep streams every row)
  for (int b = 0; b < sc.nba; b++) {
    if (ln.bi(b)[DG_BI_N_LINKS] == 0 || b == sc.reg_body[0] || b == sc.reg_body[1] || (helper_rows && b == hb)) continue;
    motor_guess(ln, b);
  }
  if (sc.ncons > 0) build_constraint_rows(ln);
  // ---- contact rows: lanes are grouped by pair id so that every table lookup stays wave-uniform
  // all-dense scenes: the sweeps start from a zero velocity change held in registers, so until they finish the LDS
  // velocity-change blocks are free -- park the generalised velocities there for the row right-hand sides
  const bool vel_dense = sc.dense && sc.nt >= 1 && sc.reg_body[0] < 0;
  wave_max_cont = [&] { int m = ncont; for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o)); return m; }();
  if (vel_dense && wave_max_cont > 0)
    for (int b = 0; b < sc.nba; b++) if (!(ln.fixed(b) && ln.bi(b)[DG_BI_N_LINKS] == 0)) ln.gen_vel_store(b, ln.plb(b)[PLB_DV]);
  if constexpr (!SLICED) {
  for (int c = 0; c < wave_max_cont; c++) {
    const bool has = c < ncont; const int mypair = has ? (int)ln.L(sc.cont_off + 1 + c * CL_STRIDE + CL_PAIR) : -1;
    bool todo = has && !build_contact_rows_base(ln, c, has, vel_dense);  // base-on-base contacts: one pass for every pair
    if (!sc.no_chain_rows) todo = todo && !build_contact_rows_chains(ln, c, todo);  // chain-on-chain contacts (two arms touching): likewise
    while (__any(todo)) {
      const int leader = __ffsll((long long)__ballot(todo)) - 1;
      const int pair = __shfl(mypair, leader);
      const bool mine = todo && mypair == pair;
      build_contact_rows(ln, c, pair, mine, vel_dense);
      todo = todo && !mine;
    }
  }
  prof.stamp(PS_ROWS);
  }
  }  // primary
  if constexpr (SLICED) {
    // Lane-sliced modes: the SL lanes of an env's group (the grouping of the sweeps: lane = env * SL + slice) share the
    // row construction -- four lanes per contact, one direction each (the fourth idles), SL / 4 contacts at a time;
    // two lanes per env: one contact each, all three directions.  Same arithmetic per row as the one-lane loop.
    constexpr int SL = SLN, CPI = SL >= 4 ? SL / 4 : SL;  // contacts per env per pass
    const int sl = qsl;
    const int ncq = __shfl(ncont, qe), wmc = __builtin_amdgcn_readfirstlane(wave_max_cont);
    const bool vd = sc.dense && sc.nt >= 1 && sc.reg_body[0] < 0;
    const int csub = SL >= 4 ? sl >> 2 : sl, d_lo = SL >= 4 ? (sl & 3) : 0, d_hi = SL >= 4 ? ((sl & 3) < 3 ? (sl & 3) + 1 : 0) : 3;
    for (int c0 = 0; c0 < wmc; c0 += CPI) {
      const int c = c0 + csub; const bool has = c < ncq && d_lo < d_hi;
      const int mypair = has ? (int)lq.L(sc.cont_off + 1 + c * CL_STRIDE + CL_PAIR) : -1;
      bool todo = has && !build_contact_rows_base(lq, c, has, vd, d_lo, d_hi);  // base-on-base contacts: one pass for every pair
      while (__any(todo)) {
        const int leader = __ffsll((long long)__ballot(todo)) - 1;
        const int pair = __shfl(mypair, leader);
        const bool mine = todo && mypair == pair;
        build_contact_rows(lq, c, pair, mine, vd, d_lo, d_hi);
        todo = todo && !mine;
      }
    }
    prof.stamp(PS_ROWS);
  }
  // ---- projected Gauss-Seidel
  bool split_now = false;
  if constexpr (PAR) split_now = split_decide_main(ln, wave_max_cont, limit_mask, limit_rows);
  const bool all_dense = sc.dense && sc.nt >= 1 && sc.reg_body[0] < 0;
  if constexpr (SLICED) {  // wave-uniform results of the primary lanes, for every lane
    wave_max_cont = __builtin_amdgcn_readfirstlane(wave_max_cont);
    limit_rows = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(limit_rows >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)limit_rows);
  }
  if (SLICED && all_dense) {
    const bool try_regs = sc.split_pgs != -1;  // (dg_world_create: DG_NO_REG_ROWS sets -1 for ablation / tests)
    if constexpr (SLICED && LANES == 1) {
      iters_done = pgs_wave_env<PROF>(lq, __builtin_amdgcn_readfirstlane(ncont), limit_rows, prof);
    } else if constexpr (SLICED && LANES == 4) {  // 16 lanes per env: the padded DoF count is 16 or 32
      if (sc.nt <= 16) { if (!(try_regs && pgs_sliced_regs_dispatch<LANES, 16, PROF>(ln, ncont, wave_max_cont, limit_rows, prof, iters_done))) iters_done = pgs_dense_sliced<LANES, 16, PROF>(ln, ncont, wave_max_cont, limit_rows, prof); }
      else if (!(try_regs && pgs_sliced_regs_dispatch<LANES, 32, PROF>(ln, ncont, wave_max_cont, limit_rows, prof, iters_done))) iters_done = pgs_dense_sliced<LANES, 32, PROF>(ln, ncont, wave_max_cont, limit_rows, prof);
    } else if constexpr (SLICED && LANES > 0) {
      if (sc.nt <= 8) { if (!(try_regs && pgs_sliced_regs_dispatch<LANES, 8, PROF>(ln, ncont, wave_max_cont, limit_rows, prof, iters_done))) iters_done = pgs_dense_sliced<LANES, 8, PROF>(ln, ncont, wave_max_cont, limit_rows, prof); }
      else if (sc.nt <= 16) { if (!(try_regs && pgs_sliced_regs_dispatch<LANES, 16, PROF>(ln, ncont, wave_max_cont, limit_rows, prof, iters_done))) iters_done = pgs_dense_sliced<LANES, 16, PROF>(ln, ncont, wave_max_cont, limit_rows, prof); }
      else if (sc.nt <= 24) iters_done = pgs_dense_sliced<LANES, 24, PROF>(ln, ncont, wave_max_cont, limit_rows, prof);
      else iters_done = pgs_dense_sliced<LANES, 32, PROF>(ln, ncont, wave_max_cont, limit_rows, prof);
    } else if constexpr (SLICED) {
      if (sc.nt <= 8) iters_done = pgs_dense_sliced_global<LANES, 8, PROF>(ln, smem, gws, ncont, wave_max_cont, limit_rows, prof);
      else if (sc.nt <= 16) iters_done = pgs_dense_sliced_global<LANES, 16, PROF>(ln, smem, gws, ncont, wave_max_cont, limit_rows, prof);
      else if (sc.nt <= 24) iters_done = pgs_dense_sliced_global<LANES, 24, PROF>(ln, smem, gws, ncont, wave_max_cont, limit_rows, prof);
      else iters_done = pgs_dense_sliced_global<LANES, 32, PROF>(ln, smem, gws, ncont, wave_max_cont, limit_rows, prof);
    }
  } else if (PAR && split_now) {
    prof.stamp(PS_PGS_MOTOR);    // (stamped build, split sweeps: [pgs_motor] = the wait at Bq for the helper's rows,
    pgs_reg_halves(ln, 0);
    prof.stamp(PS_PGS_LIMIT);    //  [pgs_limit] = loads + motor guess + sweeps of this wavefront,
    __syncthreads();  // Bs: both wavefronts' velocity changes and impulses are in LDS
    prof.stamp(PS_PGS_CONTACT);  //  [pgs_contact] = the wait at Bs for the other wavefront's sweeps)
    iters_done = (int)ln.L(split_slots(sc));
  } else if (primary) {
  bool live = ln.valid;
  // register-chain bodies in contact: their rows are coupled, the register sweeps no longer apply -- every row
  // (motors and limits included) goes through the dense streaming sweeps instead
  constexpr bool FW = FULLWAVE || PAR;
  if (all_dense || (sc.dense && (wave_max_cont > 0 || limit_mask != 0ull) && sc.nl <= 32)) {  // (active limit rows: streamed too -- the register sweeps of this path carry motor rows only)
    for (int k = 0; k < 2; k++) if (sc.reg_body[k] >= 0) motor_guess(ln, sc.reg_body[k]);  // (<= 6 joints: no workspace needed)
    if (sc.nt <= 8) iters_done = pgs_dense<LANES, 8, PROF, FW>(ln, ncont, wave_max_cont, limit_rows, prof);
    else if (sc.nt <= 16) iters_done = pgs_dense<LANES, 16, PROF, FW>(ln, ncont, wave_max_cont, limit_rows, prof);
    else if (sc.nt <= 24) iters_done = pgs_dense<LANES, 24, PROF, FW>(ln, ncont, wave_max_cont, limit_rows, prof);
    else iters_done = pgs_dense<LANES, 32, PROF, FW>(ln, ncont, wave_max_cont, limit_rows, prof);
  } else {
  // Register-resident rows: for up to NBR fixed-base bodies with <= RN joints (every 6-axis arm) M^-1, the
  // velocity change, the motor targets and the accumulated impulses are loaded once and the sweeps below touch
  // no LDS at all -- a dependent LDS round trip per row is what bounds the generic path with one wave per SIMD.
  constexpr int NBR = 2, RN = 6;
  float rM[NBR][RN * RN], rdv[NBR][RN], rb[NBR][RN], racc[NBR][RN], rdi[NBR][RN], rdg[NBR][RN]; float smax[NBR][RN]; int rn[NBR];
#pragma unroll
  for (int k = 0; k < NBR; k++) {
    const int b = sc.reg_body[k]; rn[k] = 0;
#pragma unroll
    for (int i = 0; i < RN; i++) { rdv[k][i] = 0.f; rb[k][i] = 0.f; racc[k][i] = 0.f; rdi[k][i] = 0.f; rdg[k][i] = 0.f; smax[k][i] = 0.f;
      _Pragma("unroll") for (int c = 0; c < RN; c++) rM[k][i * RN + c] = 0.f; }
    if (b >= 0) {
      const int n = ln.bi(b)[DG_BI_N_LINKS], first = ln.bi(b)[DG_BI_FIRST_LINK], mvo = ln.plb(b)[PLB_MINV], mo0 = ln.pll(first)[PLL_MROW];
      rn[k] = n;
#pragma unroll
      for (int i = 0; i < RN; i++) {
        rdv[k][i] = 0.f; rb[k][i] = 0.f; racc[k][i] = 0.f; rdi[k][i] = 0.f; rdg[k][i] = 0.f; smax[k][i] = 0.f;
        if (i < n) {
          const float maxf = ln.mt.v[3 * (first + i) + 2]; smax[k][i] = maxf < 0.f ? -maxf : maxf * sc.hm;
          rb[k][i] = ln.L(mo0 + i * MR_STRIDE + MR_B); rdg[k][i] = ln.L(mvo + i * n + i); rdi[k][i] = 1.0f / rdg[k][i];
        }
#pragma unroll
        for (int c = 0; c < RN; c++) rM[k][i * RN + c] = (i < n && c < n) ? ln.L(mvo + i * n + c) : 0.f;
      }
    }
  }
  auto regs_to_lds = [&](int k) { const int b = sc.reg_body[k]; if (b < 0) return; const int dvo = ln.plb(b)[PLB_DV];
    _Pragma("unroll") for (int i = 0; i < RN; i++) if (i < rn[k]) ln.L(dvo + i) = rdv[k][i]; };
  auto lds_to_regs = [&](int k) { const int b = sc.reg_body[k]; if (b < 0) return; const int dvo = ln.plb(b)[PLB_DV];
    _Pragma("unroll") for (int i = 0; i < RN; i++) if (i < rn[k]) rdv[k][i] = ln.L(dvo + i); };
  bool has_generic = false;
  for (int b = 0; b < sc.nba; b++) if (ln.bi(b)[DG_BI_N_LINKS] > 0 && b != sc.reg_body[0] && b != sc.reg_body[1]) has_generic = true;
  const float thr_abs = sqrtf(thr);  // the register rows track |residual|; same test as residual^2 <= thr
  if (sc.HF[DG_HF_MOTOR_GUESS] > 0.f) {  // the motor rows start next to their fixed point (motor_guess)
#pragma unroll
    for (int k = 0; k < NBR; k++) if (sc.reg_body[k] >= 0) {
      // (the limit rows of a register-chain body are swept through LDS: their right-hand sides and starting impulses live there)
      const int mo0 = ln.pll(ln.bi(sc.reg_body[k])[DG_BI_FIRST_LINK])[PLL_MROW]; const float ptol = limit_ptol(sc); const bool pinning = ptol >= 0.f;
      float lb0[RN], la0[RN], lb1[RN], la1[RN];
      _Pragma("unroll") for (int i = 0; i < RN; i++) { const bool has = i < rn[k]; const int mo = mo0 + (has ? i : 0) * MR_STRIDE;
        lb0[i] = ln.L(mo + MR_LO_B); la0[i] = has ? ln.L(mo + MR_LO_ACC) : -1.f; lb1[i] = ln.L(mo + MR_HI_B); la1[i] = has ? ln.L(mo + MR_HI_ACC) : -1.f; }
      chain_motor_guess(rM[k], rb[k], smax[k], racc[k], rdv[k], ptol, lb0, la0, lb1, la1);
      if (pinning) { _Pragma("unroll") for (int i = 0; i < RN; i++) if (i < rn[k]) { ln.L(mo0 + i * MR_STRIDE + MR_LO_ACC) = la0[i]; ln.L(mo0 + i * MR_STRIDE + MR_HI_ACC) = la1[i]; } }
    }
    if (has_generic) for (int b = 0; b < sc.nba; b++) {  // (their starting impulses are in the rows' MR_ACC / MR_LO_ACC / MR_HI_ACC slots)
      const int n = ln.bi(b)[DG_BI_N_LINKS]; if (n == 0 || b == sc.reg_body[0] || b == sc.reg_body[1]) continue;
      const int first = ln.bi(b)[DG_BI_FIRST_LINK], k0 = ln.fixed(b) ? 0 : 6, nv = ln.plb(b)[PLB_NV], dvo = ln.plb(b)[PLB_DV], mvo = ln.plb(b)[PLB_MINV];
      for (int i = 0; i < n; i++) { const int mo = ln.pll(first + i)[PLL_MROW]; const float a0 = ln.L(mo + MR_ACC) + fmaxf(ln.L(mo + MR_LO_ACC), 0.f) - fmaxf(ln.L(mo + MR_HI_ACC), 0.f); lds_axpy(ln, dvo, mvo + (k0 + i) * nv, a0, nv); }
    }
  }
  if (wave_max_cont > 0 && sc.warm_off >= 0) {  // warm start: the velocity change the rows' starting impulses amount to
    const int nvm = sc.nv_max; const bool two = sc.crow_tail > 2 * nvm;
    regs_to_lds(0); regs_to_lds(1);  // (the motor guess of a register-chain body is in registers only: without this the reload below dropped it)
    for (int c = 0; c < wave_max_cont; c++) {
      const bool has = c < ncont; const int co = sc.cont_off + 1 + c * CL_STRIDE;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const int ro = sc.tr_off + (3 * c + d) * rs; const float a0 = has ? ln.L(ro + sc.crow_tail + 1) : 0.f;
        if (a0 != 0.f) {
          lds_axpy(ln, (int)ln.L(co + CL_DVA), ro + nvm, a0, nvm);
          if (two && (int)ln.L(co + CL_NVB) > 0) lds_axpy(ln, (int)ln.L(co + CL_DVB), ro + 3 * nvm, a0, nvm);
        }
      }
    }
    lds_to_regs(0); lds_to_regs(1);
  }
  for (int it = 0; it < sc.iters; it++) {
    float maxres = 0.f, maxabs = 0.f;
    // motor rows of every body first, then joint-limit rows of every body (oracle order; rows of different
    // bodies share no unknowns, so only the order inside a body matters)
    // Straight-line code, no branches: an absent row has smax = 0, so its impulse stays 0 and its delta is 0.
    // The rows of the two bodies are independent chains the scheduler can interleave.
    {
      const float lv = live ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < RN; i++) {
#pragma unroll
        for (int k = 0; k < NBR; k++) {
          const float want = racc[k][i] + (rb[k][i] - rdv[k][i]) * rdi[k][i];
          const float nacc = __builtin_amdgcn_fmed3f(want, -smax[k][i], smax[k][i]);
          const float delta = (nacc - racc[k][i]) * lv; racc[k][i] += delta;
#pragma unroll
          for (int c = 0; c < RN; c++) rdv[k][c] += rM[k][i * RN + c] * delta;
          maxabs = fmaxf(maxabs, fabsf(delta * rdg[k][i]));
        }
      }
    }
    if (has_generic) for (int b = 0; b < sc.nba; b++) {
      const int n = ln.bi(b)[DG_BI_N_LINKS]; if (n == 0 || b == sc.reg_body[0] || b == sc.reg_body[1]) continue;
      if (ln.fixed(b) && n <= 8) maxres = fmaxf(maxres, pgs_rows_small<LANES, 8, false>(ln, b, live));
      else maxres = fmaxf(maxres, pgs_rows_generic<LANES, false>(ln, b, live));
    }
    prof.stamp(PS_PGS_MOTOR);
    if (limit_mask) {
      for (int b = 0; b < sc.nba; b++) {
        const int n = ln.bi(b)[DG_BI_N_LINKS]; if (n == 0 || !((limit_mask >> (b & 63)) & 1ull)) continue;
        if (b == sc.reg_body[0]) regs_to_lds(0);
        if (b == sc.reg_body[1]) regs_to_lds(1);
        if (ln.fixed(b) && n <= 8) maxres = fmaxf(maxres, pgs_rows_small<LANES, 8, true>(ln, b, live));
        else maxres = fmaxf(maxres, pgs_rows_generic<LANES, true>(ln, b, live));
        if (b == sc.reg_body[0]) lds_to_regs(0);
        if (b == sc.reg_body[1]) lds_to_regs(1);
      }
    }
    prof.stamp(PS_PGS_LIMIT);
    if (wave_max_cont > 0 || sc.ncons > 0) { regs_to_lds(0); regs_to_lds(1); }
    for (int q = 0; q < sc.ncons; q++) {  // fixed constraints: behind the limit rows, before the contacts (oracle order)
      const float lim = sc.KF[q * DG_KF_STRIDE + DG_KF_MAX_FORCE] * h;
      for (int d = 0; d < 6; d++) {
        const int c = sc.max_contacts + 2 * q + d / 3;
        maxres = fmaxf(maxres, solve_crow(ln, sc.tr_off + (3 * c + d % 3) * rs, sc.cont_off + 1 + c * CL_STRIDE, -lim, lim, live, true));
      }
    }
    {  // (dense scenes with contacts never get here: they take pgs_dense above)
    for (int c = 0; c < wave_max_cont; c++) {  // contact normals
      const bool has = c < ncont;
      float r = solve_crow(ln, sc.tr_off + (3 * c) * rs, sc.cont_off + 1 + c * CL_STRIDE, 0.f, 3.0e38f, live, has);
      if (has) maxres = fmaxf(maxres, r);
    }
    for (int c = 0; c < wave_max_cont; c++) {  // friction
      const bool has = c < ncont;
      const float mu = has ? ln.L(sc.cont_off + 1 + c * CL_STRIDE + CL_MU) : 0.f;
      const bool act = has && mu > 0.f;
      const float lim = act ? mu * ln.L(sc.tr_off + (3 * c) * rs + sc.crow_tail + 1) : 0.f;
#pragma unroll
      for (int d = 1; d < 3; d++) { float r = solve_crow(ln, sc.tr_off + (3 * c + d) * rs, sc.cont_off + 1 + c * CL_STRIDE, -lim, lim, live, act); if (act) maxres = fmaxf(maxres, r); }
    }
    }
    if (wave_max_cont > 0 || sc.ncons > 0) { lds_to_regs(0); lds_to_regs(1); }
    prof.stamp(PS_PGS_CONTACT);
    if (live) iters_done = it + 1;
    live = live && !(maxres <= thr && maxabs <= thr_abs);
    if (!__any(live)) break;
  }
#pragma unroll
  for (int k = 0; k < NBR; k++) {
    const int b = sc.reg_body[k];
    if (b >= 0) {
      regs_to_lds(k);
      const int mo0 = ln.pll(ln.bi(b)[DG_BI_FIRST_LINK])[PLL_MROW];
#pragma unroll
      for (int i = 0; i < RN; i++) if (i < rn[k]) ln.L(mo0 + i * MR_STRIDE + MR_ACC) = racc[k][i];
    }
  }
  }  // !all_dense
  }  // unsliced
  prof.stamp(PS_PGS);
  if (primary) {
  store_warm_cache(ln, ncont, wave_max_cont);
  if (diag_out && ln.valid) {  // include/diygym_hip.h: dg_world_set_diag_buffer
    int32_t* d = diag_out + (size_t)DG_DIAG_STRIDE * ln.env;
    d[DG_DIAG_CONTACTS] = ncont; d[DG_DIAG_PGS_ITERS] = iters_done;
    if (index == 0) { d[DG_DIAG_PGS_ITERS_FIRST] = iters_done; d[DG_DIAG_CONTACTS_FIRST] = ncont; }
  }
  // ---- apply velocity changes and integrate positions
  for (int b = 0; b < sc.nba; b++) if (!(split_now && b == hb)) integrate_body(ln, b);  // split sweeps: the helper integrates its own body
  }  // primary
  if (PAR) __syncthreads();  // B3: positions integrated; the helper may start the next substep
}

// the helper wave's side of one substep
template <int LANES>
DGD void helper_substep(const Lane<LANES>& ln, bool early, bool last) {
  const DevScene& sc = ln.sc; const int hb = sc.helper_body; Prof<false> none;
  if (last && !early) save_prev_velocities(ln, hb);
  if (!early) ln.kinematics(hb);
  __syncthreads();  // B1
  if (!early) {
    ln.template dynamics_chain<6>(hb, none);
    const int dvo = ln.plb(hb)[PLB_DV], nv = ln.plb(hb)[PLB_NV]; for (int k = 0; k < nv; k++) ln.L(dvo + k) = 0.f;
  }
  __syncthreads();  // B2
  uint64_t lm = 0ull, lr = 0ull;
  if (sc.split_pgs) { const int hf = ln.bi(hb)[DG_BI_FIRST_LINK]; setup_link_rows(ln, hf, hf + ln.bi(hb)[DG_BI_N_LINKS], lm, lr); }
  const int hf0 = ln.bi(hb)[DG_BI_FIRST_LINK];
  if (split_decide_follow(ln, hf0 + 6 <= 32 ? (unsigned)((lr >> (2 * hf0)) & 0xFFFull) : 0u, true)) {
    pgs_reg_halves(ln, 1);
    __syncthreads();  // Bs
    integrate_body(ln, hb);
  }
  __syncthreads();  // B3
}

template <int LANES, bool PROF, bool PAR = false, bool SLICED = false, bool FULLWAVE = false>
DGD void sim_step(const Lane<LANES>& ln, int32_t* diag_out, Prof<PROF>& prof, float* smem = nullptr, float* gws = nullptr) {
  const DevScene& sc = ln.sc;
  for (int k = 0; k < sc.substeps; k++) { substep<LANES, PROF, PAR, SLICED, FULLWAVE>(ln, diag_out, prof, smem, gws, k); prof.stamp(PS_INTEGRATE); }
  if (SLICED && (int)threadIdx.x >= envs_per_wave(LANES)) return;
  if (sc.debug_keep_ext) return;
  for (int b = 0; b < sc.nba; b++) { if (ln.frozen(b)) continue; const int eo = ln.ext_off(b); for (int k = 0; k < 6; k++) ln.Sset(eo + k, 0.f); }
  for (int gl = 0; gl < sc.nl; gl++) ln.Sset(ln.li(gl)[DG_LI_STATE_OFF] + DG_LS_TORQUE, 0.f);
}

// ------------------------------------------------------- inverse kinematics
// Damped-least-squares IK with null-space projection (UR5 path of ik_controller.py:61-69).
// Transient layout: [q n][J 6n][v0 n][dth n].  POSE of the body is overwritten by trial poses
// and must be refreshed by the caller afterwards.
template <int LANES>
DGD int run_ik(const Lane<LANES>& ln, int op, const float* act, bool live_lane) {
  const DevScene& sc = ln.sc; cip oi = sc.OI + op * DG_OI_STRIDE;
  const int b = oi[DG_OI_BODY], fr = oi[DG_OI_FRAME], flags = oi[DG_OI_FLAGS];
  const bool use_orn = flags & DG_IK_USE_ORIENTATION, nullsp = flags & DG_IK_NULLSPACE;
  const int first = ln.bi(b)[DG_BI_FIRST_LINK], n = ln.bi(b)[DG_BI_N_LINKS];
  const int qo = sc.tr_off, jo = qo + n, vo = jo + 6 * n, dto = vo + n;
  cfp rest = sc.FL + oi[DG_OI_FLIST];
  for (int i = 0; i < n; i++) ln.L(qo + i) = ln.S(ln.li(first + i)[DG_LI_STATE_OFF] + DG_LS_Q);
  V3 cp, cv, cw; Q4 cq; ln.frame_state(b, fr, true, cp, cq, cv, cw, false);
  V3 tp = cp + v3(act[0], act[1], act[2]); Q4 tq = cq;
  if (use_orn) tq = qmul(cq, qfrom_euler(act[3], act[4], act[5]));
  const int eel = sc.FI[fr * DG_FI_STRIDE + DG_FI_LINK];
  // without the null-space lists pybullet solves (J^T J + d I) dq = J^T e in joint space; by the push-through
  // identity that equals J^T (J J^T + d I)^-1 e, i.e. the same 6x6 solve with lambda^2 = d
  const float lam2 = nullsp ? sc.HF[DG_HF_IK_LAMBDA_SQ] : sc.HF[DG_HF_IK_JOINT_DAMPING], maxang = sc.HF[DG_HF_IK_MAX_ANGLE], g0 = sc.HF[DG_HF_IK_NULL_REST_GAIN], g1 = sc.HF[DG_HF_IK_NULL_LIMIT_GAIN];
  const float resid = sc.HF[DG_HF_IK_RESIDUAL];
  bool live = live_lane; int iters = 0;
  for (int it = 0; it < sc.ik_iters; it++) {
    ln.kinematics(b, qo);
    V3 fp, fv, fw; Q4 fq; ln.frame_state(b, fr, true, fp, fq, fv, fw, false);
    V3 ep = tp - fp;
    if (it > 0 && norm(ep) < resid) live = false;
    if (!__any(live)) break;
    iters += live ? 1 : 0;
    float dS[6] = {ep.x, ep.y, ep.z, 0.f, 0.f, 0.f};
    if (use_orn) {
      Q4 dq = qmul(tq, qconj(fq)); if (dq.w < 0.f) { dq.x = -dq.x; dq.y = -dq.y; dq.z = -dq.z; dq.w = -dq.w; }
      float sn = sqrtf(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z), an = 2.0f * atan2f(sn, dq.w), k = sn > 1e-12f ? an / sn : 2.0f;
      dS[3] = dq.x * k; dS[4] = dq.y * k; dS[5] = dq.z * k;
    }
    // Jacobian columns (world frame) for the chain root -> end-effector link, zero elsewhere
    for (int i = 0; i < 6 * n; i++) ln.L(jo + i) = 0.f;
    for (int k = eel; k >= 0; k = ln.li(k)[DG_LI_PARENT]) {
      int po = ln.pll(k)[PLL_POSE]; M3 Rk = ln.LR(po); V3 pk = ln.L3(po + 6); cfp f = ln.lf(k);
      V3 axw = mul(Rk, v3(f[DG_LF_AXIS], f[DG_LF_AXIS + 1], f[DG_LF_AXIS + 2])); int i = k - first;
      V3 jl, ja; if (ln.li(k)[DG_LI_TYPE] == 0) { jl = cross(axw, fp - pk); ja = axw; } else { jl = axw; ja = v3(0, 0, 0); }
      ln.L(jo + i) = jl.x; ln.L(jo + n + i) = jl.y; ln.L(jo + 2 * n + i) = jl.z;
      if (use_orn) { ln.L(jo + 3 * n + i) = ja.x; ln.L(jo + 4 * n + i) = ja.y; ln.L(jo + 5 * n + i) = ja.z; }
    }
    // U = J J^T + lambda^2 I (rows 3..5 are zero without orientation: block diagonal, same solution)
    float U[21];
#pragma unroll
    for (int k = 0; k < 21; k++) U[k] = 0.f;
    for (int k = 0; k < n; k++) {
      float col[6];
#pragma unroll
      for (int r = 0; r < 6; r++) col[r] = ln.L(jo + r * n + k);
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c <= r; c++) U[r * (r + 1) / 2 + c] += col[r] * col[c];
    }
#pragma unroll
    for (int r = 0; r < 6; r++) U[r * (r + 1) / 2 + r] += lam2;
    chol6(U);
    float y[6]; chol6_solve(U, dS, y);
    float Jv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < n; k++) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 6; r++) t += ln.L(jo + r * n + k) * y[r];
      float v0 = 0.f;
      if (nullsp) {
        float q = ln.L(qo + k), lo = rest[n + k], hi = rest[2 * n + k], rg = rest[3 * n + k];
        v0 = g0 * (rest[k] - q);
        if (q > hi) v0 += g1 * (hi - q) / rg;
        if (q < lo) v0 += g1 * (lo - q) / rg;
#pragma unroll
        for (int r = 0; r < 6; r++) Jv[r] += ln.L(jo + r * n + k) * v0;
      }
      ln.L(vo + k) = v0; ln.L(dto + k) = t;
    }
    float mx = 0.f;
    if (nullsp) {
      float z[6]; chol6_solve(U, Jv, z);
      for (int k = 0; k < n; k++) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 6; r++) t += ln.L(jo + r * n + k) * z[r];
        float d = ln.L(dto + k) + ln.L(vo + k) - t; ln.L(dto + k) = d; mx = fmaxf(mx, fabsf(d));
      }
    } else {
      for (int k = 0; k < n; k++) mx = fmaxf(mx, fabsf(ln.L(dto + k)));
    }
    const float scl = mx > maxang ? maxang / mx : 1.0f;
    for (int k = 0; k < n; k++) if (live) ln.L(qo + k) += scl * ln.L(dto + k);
  }
  return iters;
}

// Register-resident variant for serial chains of at most N joints on a fixed or floating base (every 6-axis arm):
// forward kinematics, Jacobian columns, the 6x6 normal matrix and both solves live in registers; nothing goes
// through LDS.  Same recursion as run_ik; selected per op at world creation (DG_IK_DEV_CHAIN).
#define DG_IK_DEV_CHAIN 256
// FULL (flag DG_IK_DEV_FULL, set at world creation): the chain has exactly N joints, all revolute, and the end-effector
// frame sits on the last link -- every 6-axis arm.  The per-link conditions are then compile-time constants and the
// whole solve is straight-line code (no selects, no branches, no copies at control-flow joins): ~1 000 instructions
// per iteration instead of ~1 600.  NS / ORN: null-space projection / orientation target, fixed per instantiation.
#define DG_IK_DEV_FULL 512
template <int LANES, int N, bool FULL = false, bool NS = false, bool ORN = false>
DGD int run_ik_chain(const Lane<LANES>& ln, int op, const float* act, bool live_lane, float* qout) {
  const DevScene& sc = ln.sc; cip oi = sc.OI + op * DG_OI_STRIDE;
  const int b = oi[DG_OI_BODY], fr = oi[DG_OI_FRAME], flags = oi[DG_OI_FLAGS];
  const bool use_orn = FULL ? ORN : (flags & DG_IK_USE_ORIENTATION) != 0, nullsp = FULL ? NS : (flags & DG_IK_NULLSPACE) != 0;
  const int first = ln.bi(b)[DG_BI_FIRST_LINK], n = FULL ? N : ln.bi(b)[DG_BI_N_LINKS];
  const int eel = FULL ? N - 1 : sc.FI[fr * DG_FI_STRIDE + DG_FI_LINK] - first;
  cfp rest = sc.FL + oi[DG_OI_FLIST]; cfp ff = sc.FF + fr * DG_FF_STRIDE;
  const V3 off = v3(ff[DG_FF_COM_POS], ff[DG_FF_COM_POS + 1], ff[DG_FF_COM_POS + 2]);
  const Q4 qoff = {ff[DG_FF_COM_QUAT], ff[DG_FF_COM_QUAT + 1], ff[DG_FF_COM_QUAT + 2], ff[DG_FF_COM_QUAT + 3]};
  const M3 R0 = ln.LR(ln.plb(b)[PLB_R0]); const V3 p0 = ln.base_pos(b);
  float q[N]; V3 ow[N], aw[N];
  // Chain constants, held in VGPRs for the whole solve (the loop is too long for them to stay in SGPRs, and
  // re-fetching them through the scalar cache every iteration costs a memory round trip per link).
  // Every link frame is re-parameterised by a constant rotation Q_i whose third column is the joint axis:
  // with R'_i = R_i Q_i the recursion becomes R'_i = R'_{i-1} (Q_{i-1}^T RT_i Q_i) Rz(q_i), so whatever the axis the
  // joint rotation only mixes the first two columns of a matrix product, the world axis is the third column for
  // free, and offsets turn into Q_{i-1}^T p_i.  The constants below are those primed quantities.
  float cR[N][9], cP[N][3]; bool rev[N];
  M3 Qprev = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}, Qee = Qprev;  // Q of the parent link; of the end-effector link
#pragma unroll
  for (int i = 0; i < N; i++) {
    q[i] = 0.f; rev[i] = true;
#pragma unroll
    for (int k = 0; k < 9; k++) cR[i][k] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) cP[i][k] = 0.f;
    if (FULL || i < n) {
      cfp f = ln.lf(first + i); q[i] = ln.S(ln.li(first + i)[DG_LI_STATE_OFF] + DG_LS_Q); rev[i] = FULL || ln.li(first + i)[DG_LI_TYPE] == 0;
      M3 RT; _Pragma("unroll") for (int k = 0; k < 9; k++) RT.m[k] = pin(f[DG_LF_ROT + k]);
      const V3 pT = v3(pin(f[DG_LF_POS]), pin(f[DG_LF_POS + 1]), pin(f[DG_LF_POS + 2]));
      const V3 ax = v3(pin(f[DG_LF_AXIS]), pin(f[DG_LF_AXIS + 1]), pin(f[DG_LF_AXIS + 2]));
      V3 u, v; tangent_basis(ax, u, v);  // right-handed (u, v, axis)
      const M3 Qi = {{u.x, v.x, ax.x, u.y, v.y, ax.y, u.z, v.z, ax.z}};
      const M3 RTp = mul(transpose(Qprev), mul(RT, Qi)); const V3 pTp = tmul(Qprev, pT);
#pragma unroll
      for (int k = 0; k < 9; k++) cR[i][k] = RTp.m[k];
      cP[i][0] = pTp.x; cP[i][1] = pTp.y; cP[i][2] = pTp.z;
      Qprev = Qi; if (FULL ? i == N - 1 : i == eel) Qee = Qi;
    }
  }
  const V3 offp = tmul(Qee, off);  // frame offset in the primed end-effector frame
  // null-space constants per joint, pinned like the chain constants (the loop would otherwise re-fetch 24 scalars
  // through the scalar cache every iteration)
  float nRest[N], nLo[N], nHi[N], nIrg[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    nRest[i] = 0.f; nLo[i] = -3.0e38f; nHi[i] = 3.0e38f; nIrg[i] = 0.f;
    if (nullsp && (FULL || i < n)) { nRest[i] = pin(rest[i]); nLo[i] = pin(rest[n + i]); nHi[i] = pin(rest[2 * n + i]); nIrg[i] = pin(frcp(rest[3 * n + i])); }
  }
  V3 pe; M3 Re;  // end-effector point and PRIMED link rotation R_ee Q_ee (Q and the frame offset are folded into the target)
  auto fk = [&]() {
    M3 R = R0; V3 p = p0; M3 Rl = R0; V3 pl = p0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      if (FULL || i < n) {
        M3 RT; _Pragma("unroll") for (int k = 0; k < 9; k++) RT.m[k] = cR[i][k];
        p = p + mul(R, v3(cP[i][0], cP[i][1], cP[i][2]));
        const M3 C = mul(R, RT);
        const V3 c0 = v3(C.m[0], C.m[3], C.m[6]), c1 = v3(C.m[1], C.m[4], C.m[7]), c2 = v3(C.m[2], C.m[5], C.m[8]);
        if (FULL || rev[i]) {
          const float sn = __sinf(q[i]), cs = __cosf(q[i]);
          const V3 n0 = c0 * cs + c1 * sn, n1 = c1 * cs - c0 * sn;
          const M3 Rn = {{n0.x, n1.x, c2.x, n0.y, n1.y, c2.y, n0.z, n1.z, c2.z}}; R = Rn;
        } else { R = C; p = p + c2 * q[i]; }
        ow[i] = p; aw[i] = c2;
        if (FULL ? i == N - 1 : i == eel) { Rl = R; pl = p; }
      }
    }
    pe = pl + mul(Rl, offp); Re = Rl;
  };
  fk();
  const V3 tp = pe + v3(act[0], act[1], act[2]);
  // Orientation target as a matrix, with the frame offset folded in: the error rotation is
  // R_target (R_link R_off)^T = (R_target R_off^T) R_link^T, so each iteration needs only the trace and the
  // antisymmetric part of Tm R_link^T (27 FMAs) instead of matrix -> quaternion -> product -> angle-axis.
  M3 Tm = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}};
  if (use_orn) {
    const Q4 qe0 = qnormalize(qmul(qfrom_mat(mul(Re, transpose(Qee))), qoff));  // the true link rotation is R' Q^T
    const Q4 tq = qmul(qe0, qfrom_euler(act[3], act[4], act[5]));
    Tm = mul(mul(qmat(tq), transpose(qmat(qoff))), Qee);  // error rotation = Tm_true (R' Q^T)^T = (Tm_true Q) R'^T
  }
  // without the null-space lists pybullet solves (J^T J + d I) dq = J^T e in joint space; by the push-through
  // identity that equals J^T (J J^T + d I)^-1 e, i.e. the same 6x6 solve with lambda^2 = d
  const float lam2 = nullsp ? sc.HF[DG_HF_IK_LAMBDA_SQ] : sc.HF[DG_HF_IK_JOINT_DAMPING], maxang = sc.HF[DG_HF_IK_MAX_ANGLE], g0 = sc.HF[DG_HF_IK_NULL_REST_GAIN], g1 = sc.HF[DG_HF_IK_NULL_LIMIT_GAIN];
  const float resid = sc.HF[DG_HF_IK_RESIDUAL];
  bool live = live_lane; int iters = 0;
  for (int it = 0; it < sc.ik_iters; it++) {
    if (it > 0) fk();
    const V3 ep = tp - pe;
    if (it > 0 && norm(ep) < resid) live = false;
    if (!__any(live)) break;
    iters += live ? 1 : 0;
    float dS[6] = {ep.x, ep.y, ep.z, 0.f, 0.f, 0.f};
    if (use_orn) {
      // M = Tm Re^T; rotation vector = axis * angle with axis sin = vee(M - M^T) / 2, cos = (tr M - 1) / 2
      const float* T = Tm.m; const float* R = Re.m;
      const float tr = T[0] * R[0] + T[1] * R[1] + T[2] * R[2] + T[3] * R[3] + T[4] * R[4] + T[5] * R[5] + T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
      const float m21 = T[3] * R[0] + T[4] * R[1] + T[5] * R[2], m12 = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
      const float m31 = T[6] * R[0] + T[7] * R[1] + T[8] * R[2], m13 = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
      const float m32 = T[6] * R[3] + T[7] * R[4] + T[8] * R[5], m23 = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
      const float sx = 0.5f * (m32 - m23), sy = 0.5f * (m13 - m31), sz = 0.5f * (m21 - m12);
      const float sn = fsqrt(sx * sx + sy * sy + sz * sz), an = atan2f(sn, 0.5f * (tr - 1.0f)), k = sn > 1e-12f ? fdiv(an, sn) : 1.0f;
      dS[3] = sx * k; dS[4] = sy * k; dS[5] = sz * k;
    }
    float U[21], Jv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v0[N], cols[N][6];
#pragma unroll
    for (int k = 0; k < 21; k++) U[k] = 0.f;
    auto column = [&](int i, float* col) {
      const V3 jl = (FULL || rev[i]) ? cross(aw[i], pe - ow[i]) : aw[i]; const V3 ja = ((FULL || rev[i]) && use_orn) ? aw[i] : v3(0.f, 0.f, 0.f);
      col[0] = jl.x; col[1] = jl.y; col[2] = jl.z; col[3] = ja.x; col[4] = ja.y; col[5] = ja.z;
    };
#pragma unroll
    for (int i = 0; i < N; i++) {
      v0[i] = 0.f;
      if (FULL || i < n) {
        if (nullsp) {
          v0[i] = g0 * (nRest[i] - q[i]);
          if (q[i] > nHi[i]) v0[i] += g1 * (nHi[i] - q[i]) * nIrg[i];
          if (q[i] < nLo[i]) v0[i] += g1 * (nLo[i] - q[i]) * nIrg[i];
        }
        if (FULL || i <= eel) {
          float* col = cols[i]; column(i, col);  // kept for the back-substitution below (36 registers instead of 54 instructions)
#pragma unroll
          for (int r = 0; r < 6; r++) {
            Jv[r] += col[r] * v0[i];
#pragma unroll
            for (int c = 0; c <= r; c++) U[r * (r + 1) / 2 + c] += col[r] * col[c];
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 6; r++) U[r * (r + 1) / 2 + r] += lam2;
    chol6(U);
    // dq = J^T A^-1 e + (I - J^T A^-1 J) v0 = J^T A^-1 (e - J v0) + v0: one solve (Jv is zero without the null space)
    float rhs[6], y[6];
#pragma unroll
    for (int r = 0; r < 6; r++) rhs[r] = dS[r] - Jv[r];
    chol6_solve(U, rhs, y);
    float dth[N], mx = 0.f;
#pragma unroll
    for (int i = 0; i < N; i++) {
      dth[i] = 0.f;
      if (FULL || i < n) {
        float t = 0.f;
        if (FULL || i <= eel) {
#pragma unroll
          for (int r = 0; r < 6; r++) t += cols[i][r] * y[r];
        }
        dth[i] = t + v0[i]; mx = fmaxf(mx, fabsf(dth[i]));
      }
    }
    const float scl = mx > maxang ? fdiv(maxang, mx) : 1.0f;
#pragma unroll
    for (int i = 0; i < N; i++) if ((FULL || i < n) && live) q[i] += scl * dth[i];
  }
#pragma unroll
  for (int i = 0; i < N; i++) qout[i] = q[i];
  return iters;
}

// ---- the same solve with PACKED fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: two fp32 operations per lane per instruction) -----
// A lone wavefront issues a v_pk_fma_f32 at the cadence of a v_fma_f32 (tools/micro/valu_issue4.hip: 5.01 against 5.08 cycles per
// instruction per wavefront), and with one workgroup per CU the step kernel's time IS the length of its longest wavefront's
// instruction stream -- 39 % of which was this loop.  The compiler's own pairing (SLP) lost to the register shuffling it needs
// (csrc/Makefile); here the DATA is laid out in pairs instead: a vector is (xy pair, z), a rotation its three columns, so a
// matrix-vector product is three packed and three scalar FMAs instead of nine, scalars enter through the instruction's
// op_sel broadcast, and the 6-vectors of the task space are ordered [lin.x lin.y | ang.x ang.y | lin.z ang.z] so that a Jacobian
// column is three pairs that come straight out of the kinematics (the normal matrix and the solve work in that order; the
// solution is the same, the rounding differs in the last bits).  Same recursion, same constants, same iteration control as
// run_ik_chain<.., FULL = true>; chosen for six-revolute-joint chains with the null-space lists (every 6-axis arm).
typedef float v2f __attribute__((ext_vector_type(2)));
DGD v2f sp2(float s) { v2f r = {s, s}; return r; }
DGD v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
struct P3 { v2f xy; float z; };
DGD P3 p3(V3 v) { P3 r; r.xy.x = v.x; r.xy.y = v.y; r.z = v.z; return r; }
// acc + c0 x + c1 y + c2 z  (the columns of a rotation times a vector)
DGD P3 colmul_acc(const P3& c0, const P3& c1, const P3& c2, float x, float y, float z, const P3& acc) {
  P3 r; r.xy = fma2(c2.xy, sp2(z), fma2(c1.xy, sp2(y), fma2(c0.xy, sp2(x), acc.xy))); r.z = fmaf(c2.z, z, fmaf(c1.z, y, fmaf(c0.z, x, acc.z))); return r;
}
DGD P3 colmul(const P3& c0, const P3& c1, const P3& c2, float x, float y, float z) {
  P3 r; r.xy = fma2(c2.xy, sp2(z), fma2(c1.xy, sp2(y), c0.xy * sp2(x))); r.z = fmaf(c2.z, z, fmaf(c1.z, y, c0.z * x)); return r;
}
template <int LANES, bool ORN>
DGD int run_ik_chain_pk(const Lane<LANES>& ln, int op, const float* act, bool live_lane, float* qout) {
  constexpr int N = 6;
  const DevScene& sc = ln.sc; cip oi = sc.OI + op * DG_OI_STRIDE;
  const int b = oi[DG_OI_BODY], fr = oi[DG_OI_FRAME];
  const int first = ln.bi(b)[DG_BI_FIRST_LINK];
  cfp rest = sc.FL + oi[DG_OI_FLIST]; cfp ff = sc.FF + fr * DG_FF_STRIDE;
  const V3 off = v3(ff[DG_FF_COM_POS], ff[DG_FF_COM_POS + 1], ff[DG_FF_COM_POS + 2]);
  const Q4 qoff = {ff[DG_FF_COM_QUAT], ff[DG_FF_COM_QUAT + 1], ff[DG_FF_COM_QUAT + 2], ff[DG_FF_COM_QUAT + 3]};
  const M3 R0 = ln.LR(ln.plb(b)[PLB_R0]); const V3 p0 = ln.base_pos(b);
  float q[N];
  // chain constants in the primed frames of run_ik_chain (every joint rotates about the z axis of its frame), pinned in VGPRs
  float cR[N][9], cP[N][3];
  M3 Qprev = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}}, Qee = Qprev;
#pragma unroll
  for (int i = 0; i < N; i++) {
    cfp f = ln.lf(first + i); q[i] = ln.S(ln.li(first + i)[DG_LI_STATE_OFF] + DG_LS_Q);
    M3 RT; _Pragma("unroll") for (int k = 0; k < 9; k++) RT.m[k] = pin(f[DG_LF_ROT + k]);
    const V3 pT = v3(pin(f[DG_LF_POS]), pin(f[DG_LF_POS + 1]), pin(f[DG_LF_POS + 2]));
    const V3 ax = v3(pin(f[DG_LF_AXIS]), pin(f[DG_LF_AXIS + 1]), pin(f[DG_LF_AXIS + 2]));
    V3 u, v; tangent_basis(ax, u, v);
    const M3 Qi = {{u.x, v.x, ax.x, u.y, v.y, ax.y, u.z, v.z, ax.z}};
    const M3 RTp = mul(transpose(Qprev), mul(RT, Qi)); const V3 pTp = tmul(Qprev, pT);
#pragma unroll
    for (int k = 0; k < 9; k++) cR[i][k] = RTp.m[k];
    cP[i][0] = pTp.x; cP[i][1] = pTp.y; cP[i][2] = pTp.z;
    Qprev = Qi; if (i == N - 1) Qee = Qi;
  }
  const V3 offp = tmul(Qee, off);
  float nRest[N], nLo[N], nHi[N], nIrg[N];
#pragma unroll
  for (int i = 0; i < N; i++) { nRest[i] = pin(rest[i]); nLo[i] = pin(rest[N + i]); nHi[i] = pin(rest[2 * N + i]); nIrg[i] = pin(frcp(rest[3 * N + i])); }
  const P3 b0 = p3(v3(R0.m[0], R0.m[3], R0.m[6])), b1 = p3(v3(R0.m[1], R0.m[4], R0.m[7])), b2 = p3(v3(R0.m[2], R0.m[5], R0.m[8])), bp = p3(p0);
  P3 ow[N], aw[N], pe, e0, e1, e2;  // link origins, world joint axes; end-effector point and the columns of its (primed) rotation
  auto fk = [&]() {
    P3 c0 = b0, c1 = b1, c2 = b2, p = bp;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const float* T = cR[i];
      p = colmul_acc(c0, c1, c2, cP[i][0], cP[i][1], cP[i][2], p);
      const P3 C0 = colmul(c0, c1, c2, T[0], T[3], T[6]), C1 = colmul(c0, c1, c2, T[1], T[4], T[7]), C2 = colmul(c0, c1, c2, T[2], T[5], T[8]);
      const float sn = __sinf(q[i]), cs = __cosf(q[i]);
      P3 n0, n1;
      n0.xy = fma2(C1.xy, sp2(sn), C0.xy * sp2(cs)); n0.z = fmaf(C1.z, sn, C0.z * cs);
      n1.xy = fma2(C0.xy, sp2(-sn), C1.xy * sp2(cs)); n1.z = fmaf(C0.z, -sn, C1.z * cs);
      c0 = n0; c1 = n1; c2 = C2; ow[i] = p; aw[i] = C2;
    }
    pe = colmul_acc(c0, c1, c2, offp.x, offp.y, offp.z, p); e0 = c0; e1 = c1; e2 = c2;
  };
  fk();
  const V3 pe0 = v3(pe.xy.x, pe.xy.y, pe.z);
  const V3 tp = pe0 + v3(act[0], act[1], act[2]);
  M3 Tm = {{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}};
  if (ORN) {
    const M3 Re = {{e0.xy.x, e1.xy.x, e2.xy.x, e0.xy.y, e1.xy.y, e2.xy.y, e0.z, e1.z, e2.z}};
    const Q4 qe0 = qnormalize(qmul(qfrom_mat(mul(Re, transpose(Qee))), qoff));
    const Q4 tq = qmul(qe0, qfrom_euler(act[3], act[4], act[5]));
    Tm = mul(mul(qmat(tq), transpose(qmat(qoff))), Qee);
  }
  const float lam2 = sc.HF[DG_HF_IK_LAMBDA_SQ], maxang = sc.HF[DG_HF_IK_MAX_ANGLE], g0 = sc.HF[DG_HF_IK_NULL_REST_GAIN], g1 = sc.HF[DG_HF_IK_NULL_LIMIT_GAIN];
  const float resid = sc.HF[DG_HF_IK_RESIDUAL];
  bool live = live_lane; int iters = 0;
  for (int it = 0; it < sc.ik_iters; it++) {
    if (it > 0) fk();
    const v2f epxy = v2f{tp.x, tp.y} - pe.xy; const float epz = tp.z - pe.z;
    if (it > 0 && fsqrt(epxy.x * epxy.x + epxy.y * epxy.y + epz * epz) < resid) live = false;
    if (!__any(live)) break;
    iters += live ? 1 : 0;
    // task-space vectors in the order [lin.x lin.y | ang.x ang.y | lin.z ang.z]
    v2f dA = epxy, dB = {0.f, 0.f}, dC = {epz, 0.f};
    if (ORN) {
      // M = Tm Re^T, row i of M = sum_k (column k of Re) Tm[i][k]; rotation vector from its trace and antisymmetric part
      const P3 M0 = colmul(e0, e1, e2, Tm.m[0], Tm.m[1], Tm.m[2]), M1 = colmul(e0, e1, e2, Tm.m[3], Tm.m[4], Tm.m[5]), M2 = colmul(e0, e1, e2, Tm.m[6], Tm.m[7], Tm.m[8]);
      const float tr = M0.xy.x + M1.xy.y + M2.z;
      const float sx = 0.5f * (M2.xy.y - M1.z), sy = 0.5f * (M0.z - M2.xy.x), sz = 0.5f * (M1.xy.x - M0.xy.y);
      const float sn = fsqrt(sx * sx + sy * sy + sz * sz), an = atan2f(sn, 0.5f * (tr - 1.0f)), k = sn > 1e-12f ? fdiv(an, sn) : 1.0f;
      dB = v2f{sx * k, sy * k}; dC.y = sz * k;
    }
    // normal matrix U = J J^T + lambda^2 I in packed rows (row r: its entries left of and on the diagonal), J v0, the columns kept
    float U00 = 0.f, U22 = 0.f, U44 = 0.f; v2f U1 = {0.f, 0.f}, U2a = U1, U3a = U1, U3b = U1, U4a = U1, U4b = U1, U5a = U1, U5b = U1, U5c = U1;
    v2f JvA = U1, JvB = U1, JvC = U1, cA[N], cB[N], cC[N]; float v0[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
      float v = g0 * (nRest[i] - q[i]);
      if (q[i] > nHi[i]) v += g1 * (nHi[i] - q[i]) * nIrg[i];
      if (q[i] < nLo[i]) v += g1 * (nLo[i] - q[i]) * nIrg[i];
      v0[i] = v;
      // column i: linear part axis x (pe - origin), angular part the axis
      const v2f dxy = pe.xy - ow[i].xy; const float dz = pe.z - ow[i].z; const P3& a = aw[i];
      const v2f pa = {a.xy.y, -a.xy.x}, pd = {dxy.y, -dxy.x};   // (v.y, -v.x)
      const v2f A = fma2(pa, sp2(dz), -(pd * sp2(a.z)));         // (a x d).xy = d.z perp(a.xy) - a.z perp(d.xy)
      const float lz = a.xy.x * dxy.y - a.xy.y * dxy.x;
      const v2f B = a.xy, C = {lz, a.z};
      cA[i] = A; cB[i] = B; cC[i] = C;
      JvA = fma2(A, sp2(v), JvA); JvB = fma2(B, sp2(v), JvB); JvC = fma2(C, sp2(v), JvC);
      U00 = fmaf(A.x, A.x, U00); U1 = fma2(A, sp2(A.y), U1);
      U2a = fma2(A, sp2(B.x), U2a); U22 = fmaf(B.x, B.x, U22);
      U3a = fma2(A, sp2(B.y), U3a); U3b = fma2(B, sp2(B.y), U3b);
      U4a = fma2(A, sp2(C.x), U4a); U4b = fma2(B, sp2(C.x), U4b); U44 = fmaf(C.x, C.x, U44);
      U5a = fma2(A, sp2(C.y), U5a); U5b = fma2(B, sp2(C.y), U5b); U5c = fma2(C, sp2(C.y), U5c);
    }
    float U[21] = {U00 + lam2, U1.x, U1.y + lam2, U2a.x, U2a.y, U22 + lam2, U3a.x, U3a.y, U3b.x, U3b.y + lam2,
                   U4a.x, U4a.y, U4b.x, U4b.y, U44 + lam2, U5a.x, U5a.y, U5b.x, U5b.y, U5c.x, U5c.y + lam2};
    chol6(U);
    const float rhs[6] = {dA.x - JvA.x, dA.y - JvA.y, dB.x - JvB.x, dB.y - JvB.y, dC.x - JvC.x, dC.y - JvC.y};
    float y[6]; chol6_solve(U, rhs, y);
    const v2f yA = {y[0], y[1]}, yB = {y[2], y[3]}, yC = {y[4], y[5]};
    float dth[N], mx = 0.f;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const v2f t = fma2(cC[i], yC, fma2(cB[i], yB, cA[i] * yA));
      dth[i] = (t.x + t.y) + v0[i]; mx = fmaxf(mx, fabsf(dth[i]));
    }
    const float scl = mx > maxang ? fdiv(maxang, mx) : 1.0f;
#pragma unroll
    for (int i = 0; i < N; i++) if (live) q[i] += scl * dth[i];
  }
#pragma unroll
  for (int i = 0; i < N; i++) qout[i] = q[i];
  return iters;
}

// p.applyExternalForce / p.applyExternalTorque on frame `fr` (global frame index, -1: base) of body b: a force `f` acting at
// `pos` and a torque `t`, all three in the frame's axes with `pos` relative to its origin (link_frame) or in world
// coordinates, added to what the body feels during the NEXT simulation step only: the base's external wrench (state:
// world frame, about the base origin) and -- for a frame on a movable link -- the torques of the joints between that link
// and the base (J^T of the wrench).  POSE of the body must be current.  The compiled ops (external_force, propellor) and
// the C-ABI entry dg_world_apply_wrench (user addons written in Python) share this function, bit for bit.
// (Context-free copies of the quaternion helpers: with contraction off inside them their results do not depend on what the
// compiler finds around the call once it is inlined -- see apply_frame_wrench.)
DGD Q4 qmul_x(Q4 a, Q4 b) {
#pragma clang fp contract(off)
  Q4 r = {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
  return r;
}
DGD Q4 qnormalize_x(Q4 a) {
#pragma clang fp contract(off)
  const float n = __frsqrt_rn(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  Q4 r = {a.x * n, a.y * n, a.z * n, a.w * n};
  return r;
}
DGD M3 qmat_x(Q4 q) {
#pragma clang fp contract(off)
  const float n = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = n > 0.f ? 2.0f * __frcp_rn(n) : 0.f;
  const float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  M3 R = {{1.f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.f - (xx + yy)}};
  return R;
}
DGD float spool_up(float w, float a, float k) {  // w + (a - w) k, every operation rounded on its own
#pragma clang fp contract(off)
  const float d = a - w; const float p = d * k; return w + p;
}
DGD float mul_sep(float a, float b) {  // a product that no later addition may absorb into an FMA
#pragma clang fp contract(off)
  const float p = a * b; return p;
}
DGD V3 mulx(const M3& R, V3 v) {
#pragma clang fp contract(off)
  return v3(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z, R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z);
}
template <int LANES>
DGD void apply_frame_wrench(const Lane<LANES>& ln, int b, int fr, V3 f, V3 pos, V3 t, bool link_frame) {
  // Every product and sum below is rounded on its own (no fused multiply-add), and the pose of a frame that is rigidly
  // attached to the base comes straight from the state through the context-free helpers above: inlined into the step
  // kernel the compiled ops pass literal zeros and sit among other arithmetic, and a compiler free to contract would fuse
  // e.g. `r x F + R t` differently there than in wrench_kernel, where the same vectors arrive from memory.  This way the
  // two are the same bits (tests/test_environment_gpu.py::test_python_hook_addon_equals_the_compiled_propellor_bit_for_bit).
#pragma clang fp contract(off)
  const DevScene& sc = ln.sc;
  const int gl = fr < 0 ? -1 : sc.FI[fr * DG_FI_STRIDE + DG_FI_LINK];
  V3 F = f, T = t, P = pos;
  if (link_frame) {
    V3 fp; M3 R;
    // LINK_FRAME is the link's INERTIAL frame -- origin at its centre of mass, axes of the URDF <inertial> -- as pybullet's
    // (btMultiBody's m_cachedWorldTransform / base world transform [R]), not the joint frame
    if (gl < 0) {  // on the base: pose from the state
      const Q4 qb = ln.base_quat(b); const V3 pb = ln.base_pos(b);
      cfp ff = fr >= 0 ? sc.FF + fr * DG_FF_STRIDE + DG_FF_COM_POS : ln.bf(b) + DG_BF_REPORT_POS;  // (both: pos3 then quat4)
      const V3 off = v3(ff[0], ff[1], ff[2]); const Q4 qo = {ff[3], ff[4], ff[5], ff[6]};
      const V3 ro = mulx(qmat_x(qb), off); fp = v3(pb.x + ro.x, pb.y + ro.y, pb.z + ro.z); R = qmat_x(qnormalize_x(qmul_x(qb, qo)));
    } else { V3 fv, fw; Q4 fq; ln.frame_state(b, fr, true, fp, fq, fv, fw, false); R = qmat_x(fq); }
    F = mulx(R, f); T = mulx(R, t);
    const V3 rp = mulx(R, pos); P = v3(fp.x + rp.x, fp.y + rp.y, fp.z + rp.z);
  }
  if (!ln.fixed(b)) {
    const V3 bp = ln.base_pos(b), r = v3(P.x - bp.x, P.y - bp.y, P.z - bp.z);
    const V3 tq = v3((r.y * F.z - r.z * F.y) + T.x, (r.z * F.x - r.x * F.z) + T.y, (r.x * F.y - r.y * F.x) + T.z); const int eo = ln.ext_off(b);
    ln.Sset(eo, ln.S(eo) + F.x); ln.Sset(eo + 1, ln.S(eo + 1) + F.y); ln.Sset(eo + 2, ln.S(eo + 2) + F.z);
    ln.Sset(eo + 3, ln.S(eo + 3) + tq.x); ln.Sset(eo + 4, ln.S(eo + 4) + tq.y); ln.Sset(eo + 5, ln.S(eo + 5) + tq.z);
  }
  for (int k = gl; k >= 0; k = ln.li(k)[DG_LI_PARENT]) {
    const int po = ln.pll(k)[PLL_POSE], lo = ln.li(k)[DG_LI_STATE_OFF]; cfp lf = ln.lf(k);
    const M3 Rk = ln.LR(po); const V3 pk = ln.L3(po + 6), a = v3(lf[DG_LF_AXIS], lf[DG_LF_AXIS + 1], lf[DG_LF_AXIS + 2]);
    const V3 axw = v3(Rk.m[0] * a.x + Rk.m[1] * a.y + Rk.m[2] * a.z, Rk.m[3] * a.x + Rk.m[4] * a.y + Rk.m[5] * a.z, Rk.m[6] * a.x + Rk.m[7] * a.y + Rk.m[8] * a.z);
    const V3 r = v3(P.x - pk.x, P.y - pk.y, P.z - pk.z), m = v3((r.y * F.z - r.z * F.y) + T.x, (r.z * F.x - r.x * F.z) + T.y, (r.x * F.y - r.y * F.x) + T.z);
    const float tau = ln.li(k)[DG_LI_TYPE] == 0 ? axw.x * m.x + axw.y * m.y + axw.z * m.z : axw.x * F.x + axw.y * F.y + axw.z * F.z;
    ln.Sset(lo + DG_LS_TORQUE, ln.S(lo + DG_LS_TORQUE) + tau);
  }
}

// ------------------------------------------------------------ addon program
template <int LANES>
DGD void run_update_ops(const Lane<LANES>& ln, const float* act_row, uint64_t mask, int only_body = -1, int skip_body = -1, int32_t* diag = nullptr) {
  const DevScene& sc = ln.sc; int ik_ord = -1;
  for (int op = 0; op < sc.nops; op++) {
    cip oi = sc.OI + op * DG_OI_STRIDE; cfp of = sc.OF + op * DG_OF_STRIDE; const int code = oi[DG_OI_CODE];
    if (code == DG_OP_IK_CONTROL) ik_ord++;  // ordinal among the scene's inverse-kinematics ops (diagnostics column)
    if (code < DG_OP_JOINT_CONTROL || code > DG_OP_ADMITTANCE) continue;
    if ((only_body >= 0 && oi[DG_OI_BODY] != only_body) || oi[DG_OI_BODY] == skip_body) continue;
    if (!((mask >> oi[DG_OI_SLOT]) & 1ull)) continue;
    const float* a = act_row + oi[DG_OI_IO_OFF]; cip il = sc.IL + oi[DG_OI_ILIST]; const int n = oi[DG_OI_N];
    if (code == DG_OP_JOINT_CONTROL) {
      const int mode = oi[DG_OI_FLAGS];
      for (int k = 0; k < n; k++) {
        const int lo = ln.li(il[k])[DG_LI_STATE_OFF];
        if (mode == DG_JC_POSITION) { ln.Sset(lo + DG_LS_TARGET_POS, a[k]); ln.Sset(lo + DG_LS_TARGET_VEL, 0.f); }
        else if (mode == DG_JC_VELOCITY) { ln.Sset(lo + DG_LS_TARGET_VEL, a[k]); ln.Sset(lo + DG_LS_TARGET_POS, 0.f); }
        else ln.Sset(lo + DG_LS_TORQUE, a[k]);
      }
    } else if (code == DG_OP_IK_CONTROL) {
      const int b = oi[DG_OI_BODY];
      float av[6] = {a[0], a[1], a[2], 0.f, 0.f, 0.f};
      if (oi[DG_OI_FLAGS] & DG_IK_USE_ORIENTATION) { av[3] = a[3]; av[4] = a[4]; av[5] = a[5]; }
      if (oi[DG_OI_FLAGS] & DG_IK_DEV_CHAIN) {
        float qs[6]; int ik_it;
        const int fl = oi[DG_OI_FLAGS];
        // (six revolute joints, null-space lists: the packed solve; DG_NO_FULL_IK at world creation keeps such an arm on the general
        // register-resident form below -- the alternative the tests compare it with)
        if ((fl & DG_IK_DEV_FULL) && (fl & DG_IK_NULLSPACE) && (fl & DG_IK_USE_ORIENTATION)) ik_it = run_ik_chain_pk<LANES, true>(ln, op, av, ln.valid, qs);
        else if ((fl & DG_IK_DEV_FULL) && (fl & DG_IK_NULLSPACE)) ik_it = run_ik_chain_pk<LANES, false>(ln, op, av, ln.valid, qs);
        else ik_it = run_ik_chain<LANES, 6>(ln, op, av, ln.valid, qs);
        if (diag && ln.valid && ik_ord < DG_DIAG_N_IK) diag[(size_t)DG_DIAG_STRIDE * ln.env + DG_DIAG_IK_ITERS + ik_ord] = ik_it;
        const int first = ln.bi(b)[DG_BI_FIRST_LINK];
        for (int k = 0; k < n; k++) {
          const int lo = ln.li(il[k])[DG_LI_STATE_OFF], j = il[k] - first; float v = qs[0];
#pragma unroll
          for (int c = 1; c < 6; c++) v = (j == c) ? qs[c] : v;
          ln.Sset(lo + DG_LS_TARGET_POS, v); ln.Sset(lo + DG_LS_TARGET_VEL, 0.f);
        }
        continue;  // POSE was not touched
      }
      { const int ik_it = run_ik(ln, op, av, ln.valid);
        if (diag && ln.valid && ik_ord < DG_DIAG_N_IK) diag[(size_t)DG_DIAG_STRIDE * ln.env + DG_DIAG_IK_ITERS + ik_ord] = ik_it; }
      for (int k = 0; k < n; k++) {
        const int lo = ln.li(il[k])[DG_LI_STATE_OFF];
        ln.Sset(lo + DG_LS_TARGET_POS, ln.L(sc.tr_off + (il[k] - ln.bi(b)[DG_BI_FIRST_LINK]))); ln.Sset(lo + DG_LS_TARGET_VEL, 0.f);
      }
      ln.kinematics(b);  // restore POSE to the state's joint angles
    } else if (code == DG_OP_ADMITTANCE) {
      // J^T wrench at the end-effector point + gravity compensation + joint PD, applied as joint torques
      // (admittance_controller.py:36-55).  World-frame Jacobian columns from the POSE region.
      const int b = oi[DG_OI_BODY], first = ln.bi(b)[DG_BI_FIRST_LINK], nl = ln.bi(b)[DG_BI_N_LINKS];
      V3 fp, fv, fw; Q4 fq; ln.frame_state(b, oi[DG_OI_FRAME], true, fp, fq, fv, fw, false);
      const V3 pw = fp + mul(qmat(fq), v3(of[0], of[1], of[2]));
      const int eel = sc.FI[oi[DG_OI_FRAME] * DG_FI_STRIDE + DG_FI_LINK];
      const V3 F = v3(a[0], a[1], a[2]), T = v3(a[3], a[4], a[5]), g = v3(sc.gx, sc.gy, sc.gz); cfp tgt = sc.FL + oi[DG_OI_FLIST];
      for (int k = 0; k < n; k++) {
        const int gl = il[k], lo = ln.li(gl)[DG_LI_STATE_OFF], po = ln.pll(gl)[PLL_POSE]; cfp f = ln.lf(gl);
        const M3 Rj = ln.LR(po); const V3 oj = ln.L3(po + 6), axw = mul(Rj, v3(f[DG_LF_AXIS], f[DG_LF_AXIS + 1], f[DG_LF_AXIS + 2]));
        const bool rev = ln.li(gl)[DG_LI_TYPE] == 0;
        bool anc = false; for (int i = eel; i >= 0; i = ln.li(i)[DG_LI_PARENT]) if (i == gl) anc = true;
        float tau = 0.f;
        if (anc) tau += rev ? dot(F, cross(axw, pw - oj)) + dot(T, axw) : dot(F, axw);
        for (int i = first; i < first + nl; i++) {
          bool sub = false; for (int q = i; q >= 0; q = ln.li(q)[DG_LI_PARENT]) if (q == gl) sub = true;
          if (!sub) continue;
          cfp fi2 = ln.lf(i); const int pi2 = ln.pll(i)[PLL_POSE];
          const V3 cw = ln.L3(pi2 + 6) + mul(ln.LR(pi2), v3(fi2[DG_LF_COM], fi2[DG_LF_COM + 1], fi2[DG_LF_COM + 2])); const V3 w8 = g * (fi2[DG_LF_MASS] * ln.mass_scale(i));
          tau -= rev ? dot(w8, cross(axw, cw - oj)) : dot(w8, axw);
        }
        tau += of[3] * (tgt[k] - ln.S(lo + DG_LS_Q)) - of[4] * ln.S(lo + DG_LS_QD);
        ln.Sset(lo + DG_LS_TORQUE, ln.S(lo + DG_LS_TORQUE) + tau);
      }
    } else if (code == DG_OP_EXTERNAL_FORCE) {  // external_force.py:24: p.applyExternalForce(uid, -1, action, xyz, WORLD_FRAME)
      const int b = oi[DG_OI_BODY]; if (ln.fixed(b)) continue;
      apply_frame_wrench(ln, b, -1, v3(a[0], a[1], a[2]), v3(of[0], of[1], of[2]), v3(0.f, 0.f, 0.f), false);
    } else if (code == DG_OP_PROPELLOR) {  // drone_pilot.py:31-37: thrust along and torque about the motor frame's z, LINK_FRAME
      const int b = oi[DG_OI_BODY], so = sc.addon_off + oi[DG_OI_STATE_OFF];
      // (separately rounded difference, product and sum -- spool_up() -- so that a user addon doing the same three steps with
      // torch gets the same bits; HIP's __fadd_rn / __fmul_rn are plain operators and contract into an FMA like any other)
      const float w = spool_up(ln.S(so), a[0], of[2]); ln.Sset(so, w);
      if (ln.fixed(b)) continue;
      apply_frame_wrench(ln, b, oi[DG_OI_FRAME], v3(0.f, 0.f, mul_sep(of[0], w)), v3(0.f, 0.f, 0.f), v3(0.f, 0.f, mul_sep(of[1], w)), true);
    }
  }
}

template <int LANES>
DGD void set_base_com_pose(const Lane<LANES>& ln, int b, V3 pc, Q4 qc) {
  cfp f = ln.bf(b); const int so = ln.bi(b)[DG_BI_STATE_OFF];
  Q4 qr = {f[DG_BF_REPORT_QUAT], f[DG_BF_REPORT_QUAT + 1], f[DG_BF_REPORT_QUAT + 2], f[DG_BF_REPORT_QUAT + 3]};
  Q4 ql = qnormalize(qmul(qc, qconj(qr))); M3 R = qmat(ql);
  V3 pl = pc - mul(R, v3(f[DG_BF_REPORT_POS], f[DG_BF_REPORT_POS + 1], f[DG_BF_REPORT_POS + 2]));
  ln.Sset(so, pl.x); ln.Sset(so + 1, pl.y); ln.Sset(so + 2, pl.z); ln.Sset(so + 3, ql.x); ln.Sset(so + 4, ql.y); ln.Sset(so + 5, ql.z); ln.Sset(so + 6, ql.w);
  if (!ln.fixed(b)) for (int k = 0; k < 6; k++) ln.Sset(so + DG_BS_LINVEL + k, 0.f);
}

template <int LANES>
DGD void run_reset_ops(const Lane<LANES>& ln) {
  const DevScene& sc = ln.sc; const uint64_t episode = (uint64_t)ln.S(DG_ST_EPISODE);
  for (int op = 0; op < sc.nops; op++) {
    cip oi = sc.OI + op * DG_OI_STRIDE; cfp of = sc.OF + op * DG_OF_STRIDE; const int code = oi[DG_OI_CODE];
    if (code == DG_OP_RESPAWN) {
      const uint64_t ep = (oi[DG_OI_FLAGS] & DG_RS_ONCE) ? 0ull : episode + 1ull, ge = (uint64_t)(sc.env_base + ln.env);
      float u[6];
#pragma unroll
      for (int k = 0; k < 6; k++) u[k] = rng_uniform(sc.seed, ge, ep, (uint64_t)op, (uint64_t)k) - 0.5f;
      V3 p = v3(of[0] + u[0] * of[7], of[1] + u[1] * of[8], of[2] + u[2] * of[9]);
      Q4 q0 = {of[3], of[4], of[5], of[6]};
      set_base_com_pose(ln, oi[DG_OI_BODY], p, qmul(q0, qfrom_euler(u[3] * of[10], u[4] * of[11], u[5] * of[12])));
    } else if (code == DG_OP_RESET_JOINTS) {
      cip il = sc.IL + oi[DG_OI_ILIST]; cfp fl = sc.FL + oi[DG_OI_FLIST];
      for (int k = 0; k < oi[DG_OI_N]; k++) { const int lo = ln.li(il[k])[DG_LI_STATE_OFF]; ln.Sset(lo + DG_LS_Q, fl[k]); ln.Sset(lo + DG_LS_QD, 0.f); }
    } else if (code == DG_OP_RANDOMIZE_COLOR) {  // visual_randomizer.py:40-46: a procedural texture per env and episode (DG_TX_*) instead of an image
      const int so = sc.addon_off + oi[DG_OI_STATE_OFF]; const uint64_t ge = (uint64_t)(sc.env_base + ln.env);
      float u[DG_TX_STRIDE];
#pragma unroll
      for (int k = 0; k < DG_TX_STRIDE; k++) u[k] = rng_uniform(sc.seed, ge, episode + 1ull, (uint64_t)op, (uint64_t)k);
#pragma unroll
      for (int k = 0; k < 6; k++) ln.Sset(so + k, u[k]);
      ln.Sset(so + DG_TX_FREQ, 2.0f + 14.0f * u[6]); ln.Sset(so + DG_TX_KIND, (float)(1 + (int)(3.0f * u[7])));
    } else if (code == DG_OP_RANDOMIZE_DYNAMICS) {
      // dynamics_randomizer.py:24-32 (see DG_OP_RANDOMIZE_DYNAMICS in diygym_scene.h): new mass = log(U) * current mass
      // per joint, angular damping = log(U) * URDF joint damping (body-wide, the last joint's stays); two rounds at an
      // env's first reset, because the reference draws at construction and again in its constructor's reset()
      cfp fl = sc.FL + oi[DG_OI_FLIST]; const int n = oi[DG_OI_N], so = sc.addon_off + oi[DG_OI_STATE_OFF];
      const uint64_t ge = (uint64_t)(sc.env_base + ln.env); float damp = ln.S(so + n);
      for (int k = 0; k < n; k++) {
        float s = ln.S(so + k);
        for (int round = 0; round < 2; round++) {
          if (round == 0 && episode != 0ull) continue;
          const uint64_t ep = round == 0 ? 0ull : episode + 1ull;
          const float um = of[0] + (of[1] - of[0]) * rng_uniform(sc.seed, ge, ep, (uint64_t)op, (uint64_t)(2 * k));
          s = fminf(fmaxf(s * fabsf(logf(um)), of[4]), of[5]);
        }
        ln.Sset(so + k, s);
      }
      for (int round = 0; round < 2; round++) {  // rounds outermost for the damping: the LAST draw of the LAST round stays
        if (round == 0 && episode != 0ull) continue;
        const uint64_t ep = round == 0 ? 0ull : episode + 1ull;
        for (int k = 0; k < n; k++) {
          const float ud = of[2] + (of[3] - of[2]) * rng_uniform(sc.seed, ge, ep, (uint64_t)op, (uint64_t)(2 * k + 1));
          damp = fmaxf(logf(ud) * fl[k], 0.f);
        }
      }
      ln.Sset(so + n, damp);
    }
  }
  for (int b = 0; b < sc.nba; b++) save_prev_velocities(ln, b);  // force/torque sensors: no acceleration across a reset
  if (sc.warm_off >= 0) ln.Sset(sc.warm_off, 0.f);  // a reset teleports bodies: no contact persists across it
  ln.Sset(DG_ST_EPISODE, (float)(episode + 1ull));
}

template <int LANES>
DGD float reach_dist(const Lane<LANES>& ln, cip oi) {
  V3 pa, pb, v, w; Q4 q;
  ln.frame_state(oi[DG_OI_BODY2], oi[DG_OI_FRAME2], oi[DG_OI_FRAME2] < 0, pa, q, v, w, false);
  ln.frame_state(oi[DG_OI_BODY], oi[DG_OI_FRAME], oi[DG_OI_FRAME] < 0, pb, q, v, w, false);
  return norm(pb - pa);
}

// ---- force_torque_sensor (force_torque_sensor.py:14-23) ----------------------------------------------------------
// The body's generalised velocity at the start of the step's last substep, kept in the addon state of the first sensor
// on the body (joint rates in link order, then base linvel3 angvel3 when floating).
template <int LANES>
DGD void save_prev_velocities(const Lane<LANES>& ln, int b) {
  cip B = ln.bi(b); const int po = B[DG_BI_PREV_OFF]; if (po < 0) return;
  const int first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS];
  for (int i = 0; i < n; i++) ln.Sset(po + i, ln.S(ln.li(first + i)[DG_LI_STATE_OFF] + DG_LS_QD));
  if (!ln.fixed(b)) for (int k = 0; k < 6; k++) ln.Sset(po + n + k, ln.S(B[DG_BI_STATE_OFF] + DG_BS_LINVEL + k));
}
struct LinkMotion { M3 R; V3 p, w, v, al, a; };
// world pose, velocity and (last substep's) acceleration of link gl (-1: base) of body b; POSE must be current
template <int LANES>
DGD void link_motion(const Lane<LANES>& ln, int b, int gl, LinkMotion& o) {
  const DevScene& sc = ln.sc; cip B = ln.bi(b); const int po = B[DG_BI_PREV_OFF], first = B[DG_BI_FIRST_LINK], n = B[DG_BI_N_LINKS]; const float ih = 1.0f / sc.h;
  ln.link_world(b, -1, o.R, o.p); o.w = o.v = o.al = o.a = v3(0.f, 0.f, 0.f);
  if (!ln.fixed(b)) {
    const int so = B[DG_BI_STATE_OFF];
    o.v = v3(ln.S(so + DG_BS_LINVEL), ln.S(so + DG_BS_LINVEL + 1), ln.S(so + DG_BS_LINVEL + 2)); o.w = v3(ln.S(so + DG_BS_ANGVEL), ln.S(so + DG_BS_ANGVEL + 1), ln.S(so + DG_BS_ANGVEL + 2));
    o.a = (o.v - v3(ln.S(po + n), ln.S(po + n + 1), ln.S(po + n + 2))) * ih; o.al = (o.w - v3(ln.S(po + n + 3), ln.S(po + n + 4), ln.S(po + n + 5))) * ih;
  }
  if (gl < 0) return;
  unsigned long long path = 0ull; for (int k = gl; k >= 0; k = ln.li(k)[DG_LI_PARENT]) path |= 1ull << (k - first);
  for (int k = first; k <= gl; k++) {  // ancestors have smaller indices: the path in increasing order
    if (!((path >> (k - first)) & 1ull)) continue;
    M3 Rk; V3 pk; ln.link_world(b, k, Rk, pk); cfp f = ln.lf(k);
    const float qd = ln.S(ln.li(k)[DG_LI_STATE_OFF] + DG_LS_QD), qdd = (qd - ln.S(po + (k - first))) * ih;
    const V3 ax = mul(Rk, v3(f[DG_LF_AXIS], f[DG_LF_AXIS + 1], f[DG_LF_AXIS + 2])), r = pk - o.p;
    const V3 a1 = o.a + cross(o.al, r) + cross(o.w, cross(o.w, r)), v1 = o.v + cross(o.w, r);
    if (ln.li(k)[DG_LI_TYPE] == 0) { o.al = o.al + ax * qdd + cross(o.w, ax * qd); o.a = a1; o.v = v1; o.w = o.w + ax * qd; }
    else { o.a = a1 + ax * qdd + cross(o.w, ax * qd) * 2.0f; o.v = v1 + ax * qd; }
    o.R = Rk; o.p = pk;
  }
}
// Newton-Euler for one rigid part moving with a link: force and torque (about ps) needed for its motion, gravity taken off
template <int LANES>
DGD void ft_add_part(const Lane<LANES>& ln, const LinkMotion& m, float mass, V3 c, const Sym3& Ic, V3 ps, V3& F, V3& T) {
  const V3 rc = mul(m.R, c), pc = m.p + rc;
  const V3 ac = m.a + cross(m.al, rc) + cross(m.w, cross(m.w, rc));
  const V3 f = (ac - v3(ln.sc.gx, ln.sc.gy, ln.sc.gz)) * mass;
  // R Ic R^T applied to a vector: R (Ic (R^T x))
  const V3 nt = mul(m.R, mul(Ic, tmul(m.R, m.al))) + cross(m.w, mul(m.R, mul(Ic, tmul(m.R, m.w))));
  F = F + f; T = T + nt + cross(pc - ps, f);
}
// Reaction wrench across the joint of frame OI_FRAME: what the parent side exerts on the child side, in the child
// link's inertial frame, torque about its origin.  with_contacts: the last substep's contact list and solved impulses
// are in the workspace (step kernels; the reset kernel for the envs it stepped).
template <int LANES>
DGD void ft_wrench(const Lane<LANES>& ln, cip oi, bool with_contacts, float* out6) {
  const DevScene& sc = ln.sc; const int b = oi[DG_OI_BODY], fr = oi[DG_OI_FRAME];
  cip il = sc.IL + oi[DG_OI_ILIST]; cfp fl = sc.FL + oi[DG_OI_FLIST]; cfp ff = sc.FF + fr * DG_FF_STRIDE;
  const int ga = sc.FI[fr * DG_FI_STRIDE + DG_FI_LINK];
  LinkMotion ma; link_motion(ln, b, ga, ma);
  const Q4 qo = {ff[DG_FF_COM_QUAT], ff[DG_FF_COM_QUAT + 1], ff[DG_FF_COM_QUAT + 2], ff[DG_FF_COM_QUAT + 3]};
  const M3 Rs = mul(ma.R, qmat(qo)); const V3 ps = ma.p + mul(ma.R, v3(ff[DG_FF_COM_POS], ff[DG_FF_COM_POS + 1], ff[DG_FF_COM_POS + 2]));
  V3 F = v3(0.f, 0.f, 0.f), T = v3(0.f, 0.f, 0.f);
  if (oi[DG_OI_FLAGS] & DG_FT_WHOLE_LINK) { float m; V3 c; Sym3 Ic; ln.link_inertia(ga, m, c, Ic); ft_add_part(ln, ma, m, c, Ic, ps, F, T); }
  else if (fl[0] > 0.f) {
    const float ms = ga >= 0 ? ln.mass_scale(ga) : 1.0f; const Sym3 Ic = {fl[4] * ms, fl[5] * ms, fl[6] * ms, fl[7] * ms, fl[8] * ms, fl[9] * ms};
    ft_add_part(ln, ma, fl[0] * ms, v3(fl[1], fl[2], fl[3]), Ic, ps, F, T);
  }
  const int nm = il[0];
  for (int k = 0; k < nm; k++) {
    const int gl = il[1 + k]; float m; V3 c; Sym3 Ic; ln.link_inertia(gl, m, c, Ic);
    LinkMotion mk; link_motion(ln, b, gl, mk); ft_add_part(ln, mk, m, c, Ic, ps, F, T);
  }
  if (with_contacts) {
    const int nsh = il[1 + nm]; cip shp = il + 2 + nm; const int ncont = (int)ln.L(sc.cont_off), rs = sc.crow_tail + 3; const float ih = 1.0f / sc.h;
    for (int c = 0; c < sc.max_contacts; c++) {
      if (!__any(c < ncont)) break;
      if (c >= ncont) continue;
      const int co = sc.cont_off + 1 + c * CL_STRIDE, pair = (int)ln.L(co + CL_PAIR);
      const int sa = sc.PI[pair * DG_PI_STRIDE + DG_PI_A], sb = sc.PI[pair * DG_PI_STRIDE + DG_PI_B];  // per-lane index: vector loads
      bool inA = false, inB = false;
      for (int q = 0; q < nsh; q++) { inA = inA || shp[q] == sa; inB = inB || shp[q] == sb; }
      if (inA == inB) continue;
      // the rows push the pair's first DYNAMIC side along +dir and the other along -dir (build_contact_rows): side A is
      // that first side unless it is static
      const int ba = sc.SI[sa * DG_SI_STRIDE + DG_SI_BODY]; const bool a_dyn = !(ln.fixed(ba) && ln.bi(ba)[DG_BI_N_LINKS] == 0);
      const V3 n = ln.L3(co + CL_N), p = ln.L3(co + CL_P); V3 t1, t2; tangent_basis(n, t1, t2);
      const int ro = sc.tr_off + 3 * c * rs + sc.crow_tail + 1;
      const V3 imp = n * ln.L(ro) + t1 * ln.L(ro + rs) + t2 * ln.L(ro + 2 * rs);
      (void)a_dyn;
      const V3 f = imp * ((inA ? 1.0f : -1.0f) * ih);
      F = F - f; T = T - cross(p - ps, f);
    }
  }
  const V3 Fl = tmul(Rs, F), Tl = tmul(Rs, T);
  out6[0] = Fl.x; out6[1] = Fl.y; out6[2] = Fl.z; out6[3] = Tl.x; out6[4] = Tl.y; out6[5] = Tl.z;
}

// observe / reward / terminal ops; POSE must be current for every body.
// `part` lets the four wavefronts of the helper-wave kernel share the output phase (they write disjoint columns):
//   OUT_ALL everything; OUT_JOINT_OF: the joint-state ops of body `pb` (they read nothing but the state);
//   OUT_JOINT_NOT_OF: the joint-state ops of every other body; OUT_OBS_REST: the remaining observe ops;
//   OUT_REW_TERM: reward and terminal ops and the collapsed reward / terminal.
enum { OUT_ALL = 0, OUT_JOINT_OF, OUT_JOINT_NOT_OF, OUT_OBS_REST, OUT_REW_TERM };
template <int LANES>
// ft_mode (force/torque sensor columns): 0 after a step -- the last substep's contacts count; 1 plain observe -- no
// contact term; 2 leave them as they are (lanes a masked reset did not touch).
DGD void run_output_ops(const Lane<LANES>& ln, float* obs, float* rew, uint8_t* term, float* rew_sum, uint8_t* term_flag,
                        int part = OUT_ALL, int pb = -1, int ft_mode = 0) {
  const DevScene& sc = ln.sc; float rsum = 0.f; uint64_t groups = 0ull; bool any = false;
  // a reach_target addon emits a reward op and a terminal op on the same pair of frames: the distance is computed once
  int rk_a = -2, rk_b = -2, rk_c = -2, rk_d = -2; float rk_dist = 0.f;
  auto reach = [&](cip oi) {
    if (oi[DG_OI_BODY] != rk_a || oi[DG_OI_FRAME] != rk_b || oi[DG_OI_BODY2] != rk_c || oi[DG_OI_FRAME2] != rk_d) {
      rk_dist = reach_dist(ln, oi); rk_a = oi[DG_OI_BODY]; rk_b = oi[DG_OI_FRAME]; rk_c = oi[DG_OI_BODY2]; rk_d = oi[DG_OI_FRAME2];
    }
    return rk_dist;
  };
  for (int op = 0; op < sc.nops; op++) {
    cip oi = sc.OI + op * DG_OI_STRIDE; cfp of = sc.OF + op * DG_OF_STRIDE;
    const int code = oi[DG_OI_CODE], io = oi[DG_OI_IO_OFF]; cip il = sc.IL + oi[DG_OI_ILIST];
    if (part != OUT_ALL) {
      const bool js = code == DG_OP_OBS_JOINT_STATE, ob = code >= DG_OP_OBS_JOINT_STATE && code < DG_OP_REW_REACH;
      const bool mine = part == OUT_JOINT_OF ? (js && oi[DG_OI_BODY] == pb) : part == OUT_JOINT_NOT_OF ? (js && oi[DG_OI_BODY] != pb)
                        : part == OUT_OBS_REST ? (ob && !js) : code >= DG_OP_REW_REACH;
      if (!mine) continue;
    }
    if (code == DG_OP_OBS_JOINT_STATE) {
      const int n = oi[DG_OI_N]; int k2 = n;
      if (obs) {
        // six joints at a time, every state load before the first store (the compiler must assume that a store to
        // `obs` may alias the state, so a load-store-load chain would pay one global round trip per value)
        const bool wv = oi[DG_OI_FLAGS] & DG_JS_VELOCITY, we = oi[DG_OI_FLAGS] & DG_JS_EFFORT;
        const int ov = n, oe = wv ? 2 * n : n; (void)k2;
        for (int k0 = 0; k0 < n; k0 += 6) {
          float q_[6], v_[6], e_[6];
#pragma unroll
          for (int j = 0; j < 6; j++) {
            const int lo = ln.li(il[min(k0 + j, n - 1)])[DG_LI_STATE_OFF];
            q_[j] = ln.S(lo + DG_LS_Q); v_[j] = wv ? ln.S(lo + DG_LS_QD) : 0.f; e_[j] = we ? ln.S(lo + DG_LS_APPLIED) : 0.f;
          }
#pragma unroll
          for (int j = 0; j < 6; j++) {
            const int k = k0 + j; if (k >= n) break;
            obs[io + k] = q_[j]; if (wv) obs[io + ov + k] = v_[j]; if (we) obs[io + oe + k] = e_[j];
          }
        }
      }
    } else if (code == DG_OP_OBS_OBJECT_STATE) {
      V3 p, v, w; Q4 q; const bool wv = oi[DG_OI_FLAGS] & DG_OS_VELOCITY;
      ln.frame_state(oi[DG_OI_BODY], oi[DG_OI_FRAME], true, p, q, v, w, wv);
      if (oi[DG_OI_BODY2] >= 0) {
        V3 sp, sv, sw; Q4 sq; ln.frame_state(oi[DG_OI_BODY2], oi[DG_OI_FRAME2], true, sp, sq, sv, sw, wv);
        p = p - sp; q = qmul(sq, q); if (wv) { v = v - sv; w = w - sw; }
      }
      if (obs) {
        int k = io; obs[k++] = p.x; obs[k++] = p.y; obs[k++] = p.z;
        if (wv) { obs[k++] = v.x; obs[k++] = v.y; obs[k++] = v.z; }
        if (oi[DG_OI_FLAGS] & DG_OS_ROTATION) { V3 e = euler_from_q(q); obs[k++] = e.x; obs[k++] = e.y; obs[k++] = e.z; }
        if (wv && (oi[DG_OI_FLAGS] & DG_OS_ROTATION)) { obs[k++] = w.x; obs[k++] = w.y; obs[k++] = w.z; }
      }
    } else if (code == DG_OP_OBS_ADDON_STATE) {
      if (obs) for (int k = 0; k < oi[DG_OI_N]; k++) obs[io + k] = ln.S(sc.addon_off + oi[DG_OI_STATE_OFF] + k);
    } else if (code == DG_OP_OBS_FT) {
      if (ft_mode != 2) { float w6[6]; ft_wrench(ln, oi, ft_mode == 0, w6); if (obs) { _Pragma("unroll") for (int k = 0; k < 6; k++) obs[io + k] = w6[k]; } }
    } else if (code == DG_OP_REW_REACH) { float r = -reach(oi) * of[0]; if (rew) rew[io] = r; rsum += r; }
    else if (code == DG_OP_REW_ELECTRICITY) {
      cip B = ln.bi(oi[DG_OI_BODY]); float acc = 0.f;
      for (int i = 0; i < B[DG_BI_N_LINKS]; i++) { const int lo = ln.li(B[DG_BI_FIRST_LINK] + i)[DG_LI_STATE_OFF]; acc += fabsf(ln.S(lo + DG_LS_APPLIED) * ln.S(lo + DG_LS_QD)); }
      float r = -acc * of[0]; if (rew) rew[io] = r; rsum += r;
    } else if (code == DG_OP_REW_CONST) { if (rew) rew[io] = of[0]; rsum += of[0]; }
    else if (code == DG_OP_TERM_REACH || code == DG_OP_TERM_TILT || code == DG_OP_TERM_TIMER) {
      bool t;
      if (code == DG_OP_TERM_REACH) t = reach(oi) < of[1];
      else if (code == DG_OP_TERM_TILT) {
        V3 p, v, w; Q4 q; ln.frame_state(oi[DG_OI_BODY], -1, true, p, q, v, w, false);
        t = 2.0f * atan2f(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z), fabsf(q.w)) > of[0];
      } else t = ln.S(DG_ST_STEP) >= of[0];
      if (term) term[io] = t ? 1 : 0;
      if (t) { any = true; groups |= 1ull << (oi[DG_OI_SLOT] & 63); }
    }
  }
  if (part != OUT_ALL && part != OUT_REW_TERM) return;
  if (rew_sum) *rew_sum = rsum;
  if (term_flag) {
    if (sc.term_mode == DG_COLLAPSE_ALL) { const uint64_t want = sc.n_term_groups >= 64 ? ~0ull : ((1ull << sc.n_term_groups) - 1ull); *term_flag = (sc.n_term_groups > 0 && (groups & want) == want) ? 1 : 0; }
    else *term_flag = any ? 1 : 0;
  }
}

}  // namespace dg
