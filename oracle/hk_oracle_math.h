// hk_oracle_math.h - numeric contract of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see README
// in this directory): nothing under oracle/ is linked into, imported by or shipped with the
// product library.
//
// WGSL leaves the accuracy of sin/cos/exp/exp2/log2/pow, FMA contraction, NaN behaviour of
// min/max and the result of normalize(0) to the implementation.  A path tracer with ReSTIR feeds
// quantised state back every frame and makes discrete choices on it, so two implementations that
// differ by one ulp diverge at isolated pixels after a few frames.  To make parity a bit-exact
// statement instead of a statistical one, this header FIXES every implementation-defined choice
// using only operations that IEEE-754 defines exactly (+ - * / sqrt fma floor, integer bit
// twiddling).  The HIP kernels implement the same contract independently
// (bevy-hikari_amd/csrc/hk_device_math.hpp); tests/test_math_contract.py compares the two on the
// GPU bit for bit and compares this file with libm within a few ulp.
//
// Contract
//  * scalar expressions are evaluated as written, one IEEE rounding per operation
//    (-ffp-contract=off); the only fused operations are the explicit fmaf chains below.
//  * dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))          (4-wide: .w term outermost)
//  * cross(a,b) = (fma(a.y,b.z, -(a.z*b.y)), fma(a.z,b.x, -(a.x*b.z)), fma(a.x,b.y, -(a.y*b.x)))
//  * M*v        = per component  fma(c3,v.w, fma(c2,v.z, fma(c1,v.y, c0*v.x)))  (columns c0..c3)
//  * normalize(v) = v * (1 / sqrt(dot(v,v)))        (normalize(0) = NaN, as 0*inf)
//  * length(v)  = sqrt(dot(v,v));  mix(a,b,t) = a*(1-t) + b*t;  fract(x) = x - floor(x)
//  * min/max    = IEEE-754 minNum/maxNum: a NaN operand is dropped, -0 < +0  (what v_min_f32 /
//    v_max_f32 do on gfx950); clamp(x,lo,hi) = min(max(x,lo),hi); saturate = clamp(x,0,1)
//  * sin, cos, exp, exp2, log2: the polynomial routines below (Cephes single-precision
//    coefficients, explicit fma Horner); pow(x,y) = exp2(y*log2(x)) with pow(0,y>0) = 0, which is
//    the accuracy class WGSL itself states for pow.  Every pow on this path has a CONSTANT exponent
//    (5 in bevy_pbr's F_Schlick, 16 in normal_weight, 2 in the variance, 0.25 in luminance_weight) and
//    a non-negative base; these are evaluated the way shader compilers strength-reduce them - by
//    multiplications ((x*x)*(x*x))*x, squarings, sqrt(sqrt(x)) - which is within pow's WGSL accuracy
//    (a few ulp, tighter than exp2(y*log2 x)) and an order of magnitude cheaper: pow5_, pow16_, pow2_, pow_quarter_.
//  * f32->f16 round-to-nearest-even with overflow to inf, f16 denormals preserved.
//  * pack4x8snorm / pack2x16unorm / unpack*: the WGSL formulas (floor(0.5 + s*clamp(x))).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace orc {

struct v2 { float x, y; };
struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };
struct m3 { v3 c0, c1, c2; };           // columns
struct m4 { v4 c0, c1, c2, c3; };       // columns

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// ---- min / max: IEEE minNum / maxNum with -0 < +0
static inline float fmin_(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return (f2u(a) >> 31) ? a : b;
  return a < b ? a : b;
}
static inline float fmax_(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return (f2u(a) >> 31) ? b : a;
  return a > b ? a : b;
}
static inline float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
static inline float saturate(float x) { return clamp_(x, 0.0f, 1.0f); }
static inline float fract(float x) { return x - floorf(x); }
static inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline float sign_(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// ---- vectors
static inline v2 V2(float x, float y) { return v2{x, y}; }
static inline v3 V3(float x, float y, float z) { return v3{x, y, z}; }
static inline v3 V3s(float s) { return v3{s, s, s}; }
static inline v4 V4(float x, float y, float z, float w) { return v4{x, y, z, w}; }
static inline v4 V4(v3 a, float w) { return v4{a.x, a.y, a.z, w}; }
static inline v3 xyz(v4 a) { return v3{a.x, a.y, a.z}; }

static inline v2 operator+(v2 a, v2 b) { return {a.x + b.x, a.y + b.y}; }
static inline v2 operator-(v2 a, v2 b) { return {a.x - b.x, a.y - b.y}; }
static inline v2 operator*(v2 a, v2 b) { return {a.x * b.x, a.y * b.y}; }
static inline v2 operator/(v2 a, v2 b) { return {a.x / b.x, a.y / b.y}; }
static inline v2 operator*(v2 a, float s) { return {a.x * s, a.y * s}; }
static inline v2 operator*(float s, v2 a) { return {s * a.x, s * a.y}; }
static inline v2 operator+(v2 a, float s) { return {a.x + s, a.y + s}; }
static inline v2 operator-(v2 a, float s) { return {a.x - s, a.y - s}; }

static inline v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline v3 operator/(v3 a, v3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline v3 operator*(float s, v3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline v3 operator/(float s, v3 a) { return {s / a.x, s / a.y, s / a.z}; }
static inline v3 operator+(v3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
static inline v3 operator-(v3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
static inline v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }

static inline v4 operator+(v4 a, v4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline v4 operator-(v4 a, v4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
static inline v4 operator*(v4 a, v4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline v4 operator*(v4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline v4 operator*(float s, v4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
static inline v4 operator+(v4 a, float s) { return {a.x + s, a.y + s, a.z + s, a.w + s}; }

static inline float dot(v2 a, v2 b) { return fmaf(a.y, b.y, a.x * b.x); }
static inline float dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline float dot(v4 a, v4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }
static inline v3 cross(v3 a, v3 b) {
  return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
static inline float length(v3 a) { return sqrtf(dot(a, a)); }
static inline float length(v2 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize(v3 a) { float s = 1.0f / sqrtf(dot(a, a)); return a * s; }
static inline v2 normalize(v2 a) { float s = 1.0f / sqrtf(dot(a, a)); return a * s; }
static inline v3 min3(v3 a, v3 b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
static inline v3 max3(v3 a, v3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
static inline v3 mix(v3 a, v3 b, float t) { return {mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)}; }
static inline v4 fract(v4 a) { return {fract(a.x), fract(a.y), fract(a.z), fract(a.w)}; }

static inline v4 mul(const m4& m, v4 v) {
  return {fmaf(m.c3.x, v.w, fmaf(m.c2.x, v.z, fmaf(m.c1.x, v.y, m.c0.x * v.x))),
          fmaf(m.c3.y, v.w, fmaf(m.c2.y, v.z, fmaf(m.c1.y, v.y, m.c0.y * v.x))),
          fmaf(m.c3.z, v.w, fmaf(m.c2.z, v.z, fmaf(m.c1.z, v.y, m.c0.z * v.x))),
          fmaf(m.c3.w, v.w, fmaf(m.c2.w, v.z, fmaf(m.c1.w, v.y, m.c0.w * v.x)))};
}
static inline v3 mul(const m3& m, v3 v) {
  return {fmaf(m.c2.x, v.z, fmaf(m.c1.x, v.y, m.c0.x * v.x)),
          fmaf(m.c2.y, v.z, fmaf(m.c1.y, v.y, m.c0.y * v.x)),
          fmaf(m.c2.z, v.z, fmaf(m.c1.z, v.y, m.c0.z * v.x))};
}
static inline m4 transpose(const m4& m) {
  return {{m.c0.x, m.c1.x, m.c2.x, m.c3.x}, {m.c0.y, m.c1.y, m.c2.y, m.c3.y},
          {m.c0.z, m.c1.z, m.c2.z, m.c3.z}, {m.c0.w, m.c1.w, m.c2.w, m.c3.w}};
}
static inline m4 load_m4(const float* p) {
  return {{p[0], p[1], p[2], p[3]}, {p[4], p[5], p[6], p[7]}, {p[8], p[9], p[10], p[11]}, {p[12], p[13], p[14], p[15]}};
}

// ---- transcendental functions (Cephes single-precision polynomials, explicit fma Horner)
static inline float pow2i(int n) {  // 2^n for n in [-126,127]
  return u2f((uint32_t)(n + 127) << 23);
}

static inline float sin_poly(float r) {
  float z = r * r;
  float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  p = fmaf(p, z, -1.6666654611e-1f);
  return fmaf(p * z, r, r);
}
static inline float cos_poly(float r) {
  float z = r * r;
  float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  p = fmaf(p, z, 4.166664568298827e-2f);
  return fmaf(p * z, z, fmaf(-0.5f, z, 1.0f));
}
static inline float reduce_pio2(float x, int* q) {
  float kf = floorf(fmaf(x, 0.63661977236758134308f, 0.5f));
  *q = (int)kf;
  float r = fmaf(kf, -1.57079637050628662109375f, x);
  r = fmaf(kf, 4.37113882867379e-8f, r);
  return r;
}
static inline float sin_(float x) {
  int q; float r = reduce_pio2(x, &q);
  float s = sin_poly(r), c = cos_poly(r);
  float v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}
static inline float cos_(float x) {
  int q; float r = reduce_pio2(x, &q);
  float s = sin_poly(r), c = cos_poly(r);
  float v = (q & 1) ? s : c;
  return ((q + 1) & 2) ? -v : v;
}

static inline float scale2(float y, int k) {  // y * 2^k, k in [-252, 254], two exact-ish steps
  int k1 = k / 2, k2 = k - k1;
  return (y * pow2i(k1)) * pow2i(k2);
}
static inline float exp2_(float x) {
  if (x != x) return x;
  if (x >= 128.0f) return INFINITY;
  if (x < -150.0f) return 0.0f;
  float k = floorf(x + 0.5f);
  float f = x - k;
  float p = fmaf(1.535336188319500e-4f, f, 1.339887440266574e-3f);
  p = fmaf(p, f, 9.618437357674640e-3f);
  p = fmaf(p, f, 5.550332471162809e-2f);
  p = fmaf(p, f, 2.402264791363012e-1f);
  p = fmaf(p, f, 6.931472028550421e-1f);
  float y = fmaf(p, f, 1.0f);
  return scale2(y, (int)k);
}
static inline float exp_(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return INFINITY;
  if (x < -103.972084045410f) return 0.0f;
  float z = floorf(fmaf(1.44269504088896341f, x, 0.5f));
  float r = fmaf(z, -0.693359375f, x);
  r = fmaf(z, 2.12194440e-4f, r);
  float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float y = fmaf(p, r * r, r) + 1.0f;
  return scale2(y, (int)z);
}
static inline float log2_(float x) {
  if (x != x) return x;
  if (x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (x == INFINITY) return x;
  int e = 0;
  uint32_t u = f2u(x);
  if ((u >> 23) == 0) {  // denormal: scale up by 2^23
    x = x * 8388608.0f;
    u = f2u(x);
    e = -23;
  }
  e += (int)(u >> 23) - 126;                       // x = m * 2^e, m in [0.5,1)
  float m = u2f((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
  p = fmaf(p, m, 1.1676998740e-1f);
  p = fmaf(p, m, -1.2420140846e-1f);
  p = fmaf(p, m, 1.4249322787e-1f);
  p = fmaf(p, m, -1.6668057665e-1f);
  p = fmaf(p, m, 2.0000714765e-1f);
  p = fmaf(p, m, -2.4999993993e-1f);
  p = fmaf(p, m, 3.3333331174e-1f);
  float y = p * m * z;
  y = fmaf(-0.5f, z, y);
  // log2(1+m) = (m + y) * log2(e), split log2(e) = 1 + 0.44269504088896340735992
  float r = y * 0.44269504088896340735992f;
  r = fmaf(m, 0.44269504088896340735992f, r);
  r = r + y;
  r = r + m;
  return r + (float)e;
}
static inline float pow_(float x, float y) {
  if (x == 0.0f) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : INFINITY);
  return exp2_(y * log2_(x));
}
// pow(x, c) for the constant exponents of the path, x >= 0 (see the header comment)
static inline float pow2_(float x) { return x * x; }
static inline float pow5_(float x) { float x2 = x * x; return (x2 * x2) * x; }
static inline float pow16_(float x) { float x2 = x * x; float x4 = x2 * x2; float x8 = x4 * x4; return x8 * x8; }
static inline float pow_quarter_(float x) { return sqrtf(sqrtf(x)); }

// ---- f16 (IEEE binary16), round-to-nearest-even
static inline uint16_t f32_to_f16(float f) {
  uint32_t x = f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) {  // inf / nan
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((ax >> 13) & 0x3ffu));
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // >= 65520 rounds to inf
  if (ax < 0x33000001u) return (uint16_t)sign;               // <= 2^-25 rounds to 0
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t base;
  if (e < -14) {  // result denormal
    shift = 13 + (-14 - e);
    base = 0;
  } else {
    shift = 13;
    base = (uint32_t)(e + 15) << 10;
    m &= 0x7fffffu;
  }
  uint32_t q = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  uint32_t r = base + q;
  if (rem > half || (rem == half && (r & 1u))) r += 1;
  return (uint16_t)(sign | r);
}
static inline float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return u2f(sign);
    float v = (float)m * 5.9604644775390625e-8f;  // m * 2^-24, exact
    return sign ? -v : v;
  }
  if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
  return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

// ---- WGSL pack / unpack builtins
static inline uint32_t pack2x16float(v2 v) { return (uint32_t)f32_to_f16(v.x) | ((uint32_t)f32_to_f16(v.y) << 16); }
static inline v2 unpack2x16float(uint32_t u) { return {f16_to_f32((uint16_t)(u & 0xffffu)), f16_to_f32((uint16_t)(u >> 16))}; }
static inline uint32_t unorm16(float x) { return (uint32_t)floorf(0.5f + 65535.0f * clamp_(x, 0.0f, 1.0f)); }
static inline uint32_t pack2x16unorm(v2 v) { return unorm16(v.x) | (unorm16(v.y) << 16); }
static inline v2 unpack2x16unorm(uint32_t u) { return {(float)(u & 0xffffu) / 65535.0f, (float)(u >> 16) / 65535.0f}; }
static inline uint32_t snorm8(float x) { return (uint32_t)(int32_t)floorf(0.5f + 127.0f * clamp_(x, -1.0f, 1.0f)) & 0xffu; }
static inline uint32_t pack4x8snorm(v4 v) { return snorm8(v.x) | (snorm8(v.y) << 8) | (snorm8(v.z) << 16) | (snorm8(v.w) << 24); }
static inline float unsnorm8(uint32_t b) { return fmax_((float)(int8_t)(uint8_t)b / 127.0f, -1.0f); }
static inline v4 unpack4x8snorm(uint32_t u) { return {unsnorm8(u & 0xffu), unsnorm8((u >> 8) & 0xffu), unsnorm8((u >> 16) & 0xffu), unsnorm8(u >> 24)}; }

}  // namespace orc
