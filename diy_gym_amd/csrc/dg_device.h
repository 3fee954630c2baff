// dg_device.h -- device-side math and per-lane LDS workspace helpers for the
// batched DIYGym step kernels (gfx950 / CDNA4 only).
//
// Execution model: one environment per lane, 64 lanes per workgroup (a single
// wavefront), every piece of per-env scratch lives in LDS as
// workspace[slot][lane] so that a wave-uniform slot index gives a conflict-free
// ds_read_b32/ds_write_b32 (bank = lane mod 32 inside each 32-lane group).
// Scene constants are read through wave-uniform addresses, which the compiler
// turns into scalar (SMEM) loads served by the scalar data cache.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace dg {

struct V3 { float x, y, z; };
struct Q4 { float x, y, z, w; };
struct M3 { float m[9]; };             // row-major
struct S6 { V3 a, l; };                // spatial vector: angular, linear
struct Sym3 { float xx, xy, xz, yy, yz, zz; };
struct AI { Sym3 I; M3 H; Sym3 M; };   // articulated inertia [[I,H],[H^T,M]]

#define DGD __device__ __forceinline__

DGD V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
DGD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DGD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DGD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
DGD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DGD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DGD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// Hardware reciprocal / reciprocal square root / square root with one Newton step where needed: a few VALU instead
// of the 10-12 of the IEEE-rounded library expansions; <= 1 ulp on the normal-range operands of this code.
DGD float frcp(float b) { float r = __builtin_amdgcn_rcpf(b); return fmaf(r, fmaf(-b, r, 1.0f), r); }
DGD float frsq(float x) { float r = __builtin_amdgcn_rsqf(x); return r * fmaf(-0.5f * x * r, r, 1.5f); }
// (v_rsq_f32 answers +inf for a DENORMAL argument and the Newton step then turns that into -inf: a squared length
// between 1e-45 and 1e-38 -- a joint creeping at 1e-20 rad/s after a reset -- made norm() return -inf and the damping
// term NaN.  Below 1e-30 the root is < 1e-15 and is reported as 0.)
DGD float fsqrt(float x) { return x > 1e-30f ? x * frsq(x) : 0.f; }
DGD float norm(V3 a) { return fsqrt(dot(a, a)); }

DGD V3 mul(const M3& A, V3 b) {
  return v3(A.m[0] * b.x + A.m[1] * b.y + A.m[2] * b.z, A.m[3] * b.x + A.m[4] * b.y + A.m[5] * b.z,
            A.m[6] * b.x + A.m[7] * b.y + A.m[8] * b.z);
}
DGD V3 tmul(const M3& A, V3 b) {  // A^T b
  return v3(A.m[0] * b.x + A.m[3] * b.y + A.m[6] * b.z, A.m[1] * b.x + A.m[4] * b.y + A.m[7] * b.z,
            A.m[2] * b.x + A.m[5] * b.y + A.m[8] * b.z);
}
DGD M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
DGD M3 tmulm(const M3& A, const M3& B) {  // A^T B
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[i] * B.m[j] + A.m[3 + i] * B.m[3 + j] + A.m[6 + i] * B.m[6 + j];
  return C;
}
DGD M3 transpose(const M3& A) { M3 T = {{A.m[0], A.m[3], A.m[6], A.m[1], A.m[4], A.m[7], A.m[2], A.m[5], A.m[8]}}; return T; }
DGD M3 skew(V3 r) { M3 S = {{0.f, -r.z, r.y, r.z, 0.f, -r.x, -r.y, r.x, 0.f}}; return S; }
DGD V3 mul(const Sym3& S, V3 b) {
  return v3(S.xx * b.x + S.xy * b.y + S.xz * b.z, S.xy * b.x + S.yy * b.y + S.yz * b.z, S.xz * b.x + S.yz * b.y + S.zz * b.z);
}
DGD M3 full(const Sym3& S) { M3 A = {{S.xx, S.xy, S.xz, S.xy, S.yy, S.yz, S.xz, S.yz, S.zz}}; return A; }
DGD Sym3 symmetrize(const M3& A) {
  Sym3 S = {A.m[0], 0.5f * (A.m[1] + A.m[3]), 0.5f * (A.m[2] + A.m[6]), A.m[4], 0.5f * (A.m[5] + A.m[7]), A.m[8]};
  return S;
}

DGD Q4 qmul(Q4 a, Q4 b) {
  Q4 r = {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
  return r;
}
DGD Q4 qconj(Q4 a) { Q4 r = {-a.x, -a.y, -a.z, a.w}; return r; }
DGD Q4 qnormalize(Q4 a) {
  float n = frsq(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  Q4 r = {a.x * n, a.y * n, a.z * n, a.w * n};
  return r;
}
DGD M3 qmat(Q4 q) {
  float n = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = n > 0.f ? 2.0f * frcp(n) : 0.f;
  float xs = q.x * s, ys = q.y * s, zs = q.z * s;
  float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys, yz = q.y * zs,
        zz = q.z * zs;
  M3 R = {{1.f - (yy + zz), xy - wz, xz + wy, xy + wz, 1.f - (xx + zz), yz - wx, xz - wy, yz + wx, 1.f - (xx + yy)}};
  return R;
}
DGD Q4 qfrom_mat(const M3& R) {
  float tr = R.m[0] + R.m[4] + R.m[8];
  Q4 q;
  if (tr > 0.f) {
    float s = fsqrt(tr + 1.0f) * 2.0f; const float is = frcp(s);
    q.x = (R.m[7] - R.m[5]) * is; q.y = (R.m[2] - R.m[6]) * is; q.z = (R.m[3] - R.m[1]) * is; q.w = 0.25f * s;
  } else if (R.m[0] > R.m[4] && R.m[0] > R.m[8]) {
    float s = fsqrt(1.0f + R.m[0] - R.m[4] - R.m[8]) * 2.0f; const float is = frcp(s);
    q.x = 0.25f * s; q.y = (R.m[1] + R.m[3]) * is; q.z = (R.m[2] + R.m[6]) * is; q.w = (R.m[7] - R.m[5]) * is;
  } else if (R.m[4] > R.m[8]) {
    float s = fsqrt(1.0f + R.m[4] - R.m[0] - R.m[8]) * 2.0f; const float is = frcp(s);
    q.x = (R.m[1] + R.m[3]) * is; q.y = 0.25f * s; q.z = (R.m[5] + R.m[7]) * is; q.w = (R.m[2] - R.m[6]) * is;
  } else {
    float s = fsqrt(1.0f + R.m[8] - R.m[0] - R.m[4]) * 2.0f; const float is = frcp(s);
    q.x = (R.m[2] + R.m[6]) * is; q.y = (R.m[5] + R.m[7]) * is; q.z = 0.25f * s; q.w = (R.m[3] - R.m[1]) * is;
  }
  return qnormalize(q);
}
// fixed-axis XYZ euler -> quaternion (pybullet getQuaternionFromEuler)
DGD Q4 qfrom_euler(float r, float p, float y) {
  float sr, cr, sp, cp, sy, cy;
  sincosf(r * 0.5f, &sr, &cr); sincosf(p * 0.5f, &sp, &cp); sincosf(y * 0.5f, &sy, &cy);
  Q4 q = {sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
  return q;
}
// pybullet getEulerFromQuaternion (btQuaternion::getEulerZYX branch structure)
DGD V3 euler_from_q(Q4 q) {
  const float PI_2 = 1.57079632679489662f;
  float sarg = -2.0f * (q.x * q.z - q.w * q.y);
  if (sarg <= -0.99999f) return v3(0.f, -PI_2, 2.0f * atan2f(q.x, -q.y));
  if (sarg >= 0.99999f) return v3(0.f, PI_2, 2.0f * atan2f(-q.x, q.y));
  float sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, sqw = q.w * q.w;
  return v3(atan2f(2.0f * (q.y * q.z + q.w * q.x), sqw - sqx - sqy + sqz), asinf(sarg),
            atan2f(2.0f * (q.x * q.y + q.w * q.z), sqw + sqx - sqy - sqz));
}
// Joint rotations use the hardware sin/cos (v_sin_f32 / v_cos_f32 on the angle in revolutions): abs error
// ~1e-6 over the joint range, two instructions each instead of a ~80-instruction libm expansion.
DGD M3 rot_axis(V3 a, float th) {
  const float s = __sinf(th), c = __cosf(th); float t = 1.f - c;
  M3 R = {{t * a.x * a.x + c, t * a.x * a.y - s * a.z, t * a.x * a.z + s * a.y, t * a.x * a.y + s * a.z, t * a.y * a.y + c,
           t * a.y * a.z - s * a.x, t * a.x * a.z - s * a.y, t * a.y * a.z + s * a.x, t * a.z * a.z + c}};
  return R;
}

// ---- spatial algebra -----------------------------------------------------
// motion transform parent -> child with E = rotation parent->child coords, r = child origin in parent coords
DGD S6 xmotion(const M3& E, V3 r, const S6& v) { S6 o; o.a = mul(E, v.a); o.l = mul(E, v.l - cross(r, v.a)); return o; }
DGD S6 xforce_to_parent(const M3& E, V3 r, const S6& f) { S6 o; o.l = tmul(E, f.l); o.a = tmul(E, f.a) + cross(r, o.l); return o; }
DGD S6 operator+(const S6& a, const S6& b) { S6 o = {a.a + b.a, a.l + b.l}; return o; }
DGD S6 operator-(const S6& a, const S6& b) { S6 o = {a.a - b.a, a.l - b.l}; return o; }
DGD S6 operator*(const S6& a, float s) { S6 o = {a.a * s, a.l * s}; return o; }
DGD float dot(const S6& a, const S6& b) { return dot(a.a, b.a) + dot(a.l, b.l); }
DGD S6 crm(const S6& v, const S6& s) { S6 o; o.a = cross(v.a, s.a); o.l = cross(v.a, s.l) + cross(v.l, s.a); return o; }
DGD S6 crf(const S6& v, const S6& f) { S6 o; o.a = cross(v.a, f.a) + cross(v.l, f.l); o.l = cross(v.a, f.l); return o; }
DGD S6 mul(const AI& A, const S6& v) { S6 o; o.a = mul(A.I, v.a) + mul(A.H, v.l); o.l = tmul(A.H, v.a) + mul(A.M, v.l); return o; }
// rigid-body inertia about the frame origin: mass m, com c, inertia Ic about the com
DGD AI rigid_inertia(float m, V3 c, const Sym3& Ic) {
  AI A; float cc = dot(c, c);
  A.I.xx = Ic.xx + m * (cc - c.x * c.x); A.I.xy = Ic.xy - m * c.x * c.y; A.I.xz = Ic.xz - m * c.x * c.z;
  A.I.yy = Ic.yy + m * (cc - c.y * c.y); A.I.yz = Ic.yz - m * c.y * c.z; A.I.zz = Ic.zz + m * (cc - c.z * c.z);
  A.H = skew(c * m);
  A.M.xx = m; A.M.yy = m; A.M.zz = m; A.M.xy = 0.f; A.M.xz = 0.f; A.M.yz = 0.f;
  return A;
}
// X^T A X for the motion transform X = (E, r): child-coordinate inertia expressed in parent coordinates
DGD AI to_parent(const AI& A, const M3& E, V3 r) {
  M3 Et = transpose(E);
  M3 I1 = mul(Et, mul(full(A.I), E)), H1 = mul(Et, mul(A.H, E)), M1 = mul(Et, mul(full(A.M), E));
  M3 rx = skew(r);
  M3 Hp = mul(rx, M1);
#pragma unroll
  for (int k = 0; k < 9; k++) Hp.m[k] += H1.m[k];
  M3 t1 = mul(rx, transpose(H1)), t2 = mul(Hp, rx);
  M3 Ip;
#pragma unroll
  for (int k = 0; k < 9; k++) Ip.m[k] = I1.m[k] + t1.m[k] - t2.m[k];
  AI o; o.I = symmetrize(Ip); o.H = Hp; o.M = symmetrize(M1);
  return o;
}

// 6x6 SPD factorisation / solve, fully unrolled so everything stays in registers.
// packed lower triangle: L[i*(i+1)/2 + j], j <= i
DGD void ai_to_packed(const AI& A, float* P) {
  // rows 0..2: I ; rows 3..5: [H^T, M]
  P[0] = A.I.xx; P[1] = A.I.xy; P[2] = A.I.yy; P[3] = A.I.xz; P[4] = A.I.yz; P[5] = A.I.zz;
  P[6] = A.H.m[0]; P[7] = A.H.m[3]; P[8] = A.H.m[6]; P[9] = A.M.xx;
  P[10] = A.H.m[1]; P[11] = A.H.m[4]; P[12] = A.H.m[7]; P[13] = A.M.xy; P[14] = A.M.yy;
  P[15] = A.H.m[2]; P[16] = A.H.m[5]; P[17] = A.H.m[8]; P[18] = A.M.xz; P[19] = A.M.yz; P[20] = A.M.zz;
}
// The diagonal of the factor is stored INVERTED (1 / L_ii, from v_rsq_f32), so the factorisation and both
// triangular solves are multiply-only.
DGD bool chol6(float* P) {  // in place, returns false when not positive definite
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j <= i; j++) {
      float s = P[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= P[i * (i + 1) / 2 + k] * P[j * (j + 1) / 2 + k];
      if (i == j) { ok = ok && (s > 0.f); P[i * (i + 1) / 2 + i] = __frsqrt_rn(fmaxf(s, 1e-30f)); }
      else P[i * (i + 1) / 2 + j] = s * P[j * (j + 1) / 2 + j];
    }
  }
  return ok;
}
DGD void chol6_solve(const float* L, const float* b, float* x) {
  float y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i * (i + 1) / 2 + k] * y[k];
    y[i] = s * L[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    float s = y[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= L[k * (k + 1) / 2 + i] * x[k];
    x[i] = s * L[i * (i + 1) / 2 + i];
  }
}
// division through v_rcp_f32 (1 ulp) where IEEE rounding of the quotient does not matter (closest-point parameters...)
// a / b through v_rcp_f32 and one Newton step (4 VALU instead of the 12 of an IEEE division; <= 1 ulp for the
// normal-range operands of this code: masses, inertias, solver diagonals)
DGD float fdiv(float a, float b) { return a * frcp(b); }
// pins a wave-uniform value in a VGPR so that a long loop does not re-fetch it through the scalar cache
DGD float pin(float x) { float y; asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "s"(x)); return y; }

// counter-based RNG, identical integer recipe to the oracle (24-bit mantissa)
DGD uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31;
  return z;
}
DGD float rng_uniform(uint64_t seed, uint64_t env, uint64_t episode, uint64_t op, uint64_t comp) {
  uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ULL * (env + 1));
  z = mix64(z ^ (episode * 0xD1342543DE82EF95ULL + op * 0x2545F4914F6CDD1DULL + comp + 1));
  return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}

}  // namespace dg
