// hk_device_math.hpp - device-side numeric contract for gfx950 (see DESIGN.md "Numeric contract").
//
// WGSL leaves transcendental accuracy, FMA contraction and NaN handling of min/max to the
// implementation.  This library pins them so a frame is a pure function of its inputs on any
// conforming build: everything is built from IEEE-exact operations (v_add/v_mul/v_fma_f32,
// correctly rounded v_div/v_sqrt sequences, v_floor_f32, integer ops); the file is compiled with
// -ffp-contract=off so only the fmaf() written below become v_fma_f32.
//   dot / cross / matrix*vector : explicit fma chains (full-rate v_fma_f32)
//   min / max                   : v_min_f32 / v_max_f32 (IEEE minNum/maxNum, -0 < +0)
//   sin cos exp exp2 log2 pow   : short fma-Horner polynomials (Cephes single-precision
//                                 coefficients); arguments on this path are small (angles in
//                                 [0, 2pi], exponents of filter weights) so a 2-term Cody-Waite
//                                 reduction is enough and the routines are branch-light.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hkd {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct mat3 { f3 c0, c1, c2; };

#define HKD __device__ __forceinline__

HKD uint32_t f2u(float f) { return __float_as_uint(f); }
HKD float u2f(uint32_t u) { return __uint_as_float(u); }

HKD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
HKD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
HKD float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
HKD float saturate(float x) { return clamp_(x, 0.0f, 1.0f); }
HKD float fract(float x) { return x - floorf(x); }
HKD float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
HKD float sign_(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

HKD f2 F2(float x, float y) { return f2{x, y}; }
HKD f3 F3(float x, float y, float z) { return f3{x, y, z}; }
HKD f3 F3s(float s) { return f3{s, s, s}; }
HKD f4 F4(float x, float y, float z, float w) { return f4{x, y, z, w}; }
HKD f4 F4(f3 a, float w) { return f4{a.x, a.y, a.z, w}; }
HKD f4 F4(float4 a) { return f4{a.x, a.y, a.z, a.w}; }
HKD f3 xyz(f4 a) { return f3{a.x, a.y, a.z}; }
HKD f3 xyz(float4 a) { return f3{a.x, a.y, a.z}; }
HKD float4 to_float4(f4 a) { return make_float4(a.x, a.y, a.z, a.w); }

HKD f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
HKD f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
HKD f2 operator*(f2 a, f2 b) { return {a.x * b.x, a.y * b.y}; }
HKD f2 operator/(f2 a, f2 b) { return {a.x / b.x, a.y / b.y}; }
HKD f2 operator*(f2 a, float s) { return {a.x * s, a.y * s}; }
HKD f2 operator*(float s, f2 a) { return {s * a.x, s * a.y}; }
HKD f2 operator+(f2 a, float s) { return {a.x + s, a.y + s}; }
HKD f2 operator-(f2 a, float s) { return {a.x - s, a.y - s}; }

HKD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
HKD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
HKD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
HKD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
HKD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
HKD f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
HKD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
HKD f3 operator/(float s, f3 a) { return {s / a.x, s / a.y, s / a.z}; }
HKD f3 operator+(f3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
HKD f3 operator-(f3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
HKD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }

HKD f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
HKD f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
HKD f4 operator*(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
HKD f4 operator*(float s, f4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
HKD f4 operator+(f4 a, float s) { return {a.x + s, a.y + s, a.z + s, a.w + s}; }

HKD float dot(f2 a, f2 b) { return fmaf(a.y, b.y, a.x * b.x); }
HKD float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
HKD float dot(f4 a, f4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }
HKD f3 cross(f3 a, f3 b) {
  return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
HKD float length(f3 a) { return sqrtf(dot(a, a)); }
HKD f3 normalize(f3 a) { float s = 1.0f / sqrtf(dot(a, a)); return a * s; }
HKD f2 normalize(f2 a) { float s = 1.0f / sqrtf(dot(a, a)); return a * s; }
HKD f3 min3(f3 a, f3 b) { return {fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)}; }
HKD f3 max3(f3 a, f3 b) { return {fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)}; }
HKD f3 mix(f3 a, f3 b, float t) { return {mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t)}; }
HKD f4 fract(f4 a) { return {fract(a.x), fract(a.y), fract(a.z), fract(a.w)}; }

// columns c0..c3 given as float4; per component fma(c3,v.w, fma(c2,v.z, fma(c1,v.y, c0*v.x)))
HKD f4 mul(float4 c0, float4 c1, float4 c2, float4 c3, f4 v) {
  return {fmaf(c3.x, v.w, fmaf(c2.x, v.z, fmaf(c1.x, v.y, c0.x * v.x))), fmaf(c3.y, v.w, fmaf(c2.y, v.z, fmaf(c1.y, v.y, c0.y * v.x))),
          fmaf(c3.z, v.w, fmaf(c2.z, v.z, fmaf(c1.z, v.y, c0.z * v.x))), fmaf(c3.w, v.w, fmaf(c2.w, v.z, fmaf(c1.w, v.y, c0.w * v.x)))};
}
HKD f3 mul(const mat3& m, f3 v) {
  return {fmaf(m.c2.x, v.z, fmaf(m.c1.x, v.y, m.c0.x * v.x)), fmaf(m.c2.y, v.z, fmaf(m.c1.y, v.y, m.c0.y * v.x)),
          fmaf(m.c2.z, v.z, fmaf(m.c1.z, v.y, m.c0.z * v.x))};
}

// ---- transcendental routines
HKD float pow2i(int n) { return u2f((uint32_t)(n + 127) << 23); }
HKD float sin_poly(float r) {
  float z = r * r;
  float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  p = fmaf(p, z, -1.6666654611e-1f);
  return fmaf(p * z, r, r);
}
HKD float cos_poly(float r) {
  float z = r * r;
  float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  p = fmaf(p, z, 4.166664568298827e-2f);
  return fmaf(p * z, z, fmaf(-0.5f, z, 1.0f));
}
HKD float reduce_pio2(float x, int* q) {
  float kf = floorf(fmaf(x, 0.63661977236758134308f, 0.5f));
  *q = (int)kf;
  float r = fmaf(kf, -1.57079637050628662109375f, x);
  r = fmaf(kf, 4.37113882867379e-8f, r);
  return r;
}
HKD float sin_(float x) {
  int q;
  float r = reduce_pio2(x, &q);
  float s = sin_poly(r), c = cos_poly(r);
  float v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}
HKD float cos_(float x) {
  int q;
  float r = reduce_pio2(x, &q);
  float s = sin_poly(r), c = cos_poly(r);
  float v = (q & 1) ? s : c;
  return ((q + 1) & 2) ? -v : v;
}
HKD void sincos_(float x, float* sn, float* cs) {  // same values as sin_ / cos_, one reduction
  int q;
  float r = reduce_pio2(x, &q);
  float s = sin_poly(r), c = cos_poly(r);
  float vs = (q & 1) ? c : s;
  float vc = (q & 1) ? s : c;
  *sn = (q & 2) ? -vs : vs;
  *cs = ((q + 1) & 2) ? -vc : vc;
}
// y * 2^k rounded once.  The contract writes it as (y * 2^(k/2)) * 2^(k - k/2): for the k that exp_ / exp2_ produce (-150..128) and
// their y in [0.7, 1.42] the first product is exact and normal, so the second is the only rounding (to a subnormal, or an overflow
// to infinity, included) - which is v_ldexp_f32, one instruction instead of eight (tests/test_math_contract.py sweeps the edges).
HKD float scale2(float y, int k) {
#ifdef HK_SCALE2_MUL
  int k1 = k / 2, k2 = k - k1;
  return (y * pow2i(k1)) * pow2i(k2);
#else
  return __builtin_ldexpf(y, k);
#endif
}
// (exp2_ / exp_ keep their early returns: turning them into selects after the polynomial removes ~250 scalar instructions from
// k_denoise but made k_spatial_reuse 4 % slower - measured, round 2)
HKD float exp2_(float x) {
  if (x != x) return x;
  if (x >= 128.0f) return __builtin_inff();
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
HKD float exp_(float x) {
  if (x != x) return x;
  if (x > 88.72283905206835f) return __builtin_inff();
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
// exp_ for arguments that are never positive (filter weights exp(-|a| / b), b > 0): the overflow test cannot fire and is left out
HKD float exp_nonpositive_(float x) {
  if (x != x) return x;
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
HKD float log2_(float x) {
  if (x != x) return x;
  if (x < 0.0f) return __builtin_nanf("");
  if (x == 0.0f) return -__builtin_inff();
  if (x == __builtin_inff()) return x;
  int e = 0;
  uint32_t u = f2u(x);
  if ((u >> 23) == 0) {
    x = x * 8388608.0f;
    u = f2u(x);
    e = -23;
  }
  e += (int)(u >> 23) - 126;
  float m = u2f((u & 0x007fffffu) | 0x3f000000u);
  if (m < 0.70710678118654752440f) {
    e -= 1;
    m = m + m - 1.0f;
  } else {
    m = m - 1.0f;
  }
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
  float r = y * 0.44269504088896340735992f;
  r = fmaf(m, 0.44269504088896340735992f, r);
  r = r + y;
  r = r + m;
  return r + (float)e;
}
// x / d for many x and one d: rd = 1.0 / (double)d once (IEEE f64 division), then RN32((double)x * rd) per quotient - three full-rate
// instructions instead of the ~13-slot v_div_scale / v_rcp / fma / v_div_fmas / v_div_fixup sequence.  The f64 product is within
// 2^-52 (relative) of x / d, and a quotient of two f32 numbers that is not a rounding boundary of the f32 grid is at least 2^-49
// away from one (x = X 2^a, d = D 2^b, a boundary m = M 2^c with M odd and < 2^25: |x/d - m| / m = |X 2^s - M D| / (M D) with a
// non-zero integer numerator), so the f32 rounding is the IEEE quotient's - EXCEPT where the quotient is subnormal, where a
// boundary can be hit exactly and tie the other way.  Use it only where such a quotient cannot matter: the filter weights feed it
// to exp_, which returns 1.0f for every |argument| < 2^-25.  Zero, infinite and NaN operands behave like the division
// (0 * inf = NaN = 0 / 0 ...).  Compared against the oracle's plain divisions by every denoise parity test.
HKD float quotient_by_reciprocal(float x, double rd) { return (float)((double)x * rd); }
HKD float pow_(float x, float y) {
  if (x == 0.0f) return y > 0.0f ? 0.0f : (y == 0.0f ? 1.0f : __builtin_inff());
  return exp2_(y * log2_(x));
}
// pow(x, c) for the constant exponents of the path, x >= 0: the multiplication / square-root chains the
// numeric contract prescribes (oracle/hk_oracle_math.h header)
HKD float pow2_(float x) { return x * x; }
HKD float pow5_(float x) { float x2 = x * x; return (x2 * x2) * x; }
HKD float pow16_(float x) { float x2 = x * x; float x4 = x2 * x2; float x8 = x4 * x4; return x8 * x8; }
HKD float pow_quarter_(float x) { return sqrtf(sqrtf(x)); }

// ---- f16 storage (v_cvt_f16_f32 / v_cvt_f32_f16: round-to-nearest-even, denormals kept)
// The empty asm keeps the f32 value opaque: without it the backend folds a preceding multiply into
// v_fma_mixlo_f16 (f16(a*b + 0), ONE rounding and -0 + 0 = +0), which is not what storing an f32
// result to an rgba16float texture does (two roundings) and breaks the numeric contract.
HKD uint16_t f32_to_f16(float f) {
  asm("" : "+v"(f));
  return __builtin_bit_cast(uint16_t, (_Float16)f);
}
HKD float f16_to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
HKD uint32_t pack2x16float(float x, float y) { return (uint32_t)f32_to_f16(x) | ((uint32_t)f32_to_f16(y) << 16); }
HKD f2 unpack2x16float(uint32_t u) { return {f16_to_f32((uint16_t)(u & 0xffffu)), f16_to_f32((uint16_t)(u >> 16))}; }
HKD uint32_t unorm16(float x) { return (uint32_t)floorf(0.5f + 65535.0f * clamp_(x, 0.0f, 1.0f)); }
HKD uint32_t pack2x16unorm(float x, float y) { return unorm16(x) | (unorm16(y) << 16); }
// x / B for the integer-valued x of the norm formats (|x| <= 65535): q = x * RN(1/B) followed by one correction
// with the exact residual, q + fma(-q, B, x) * RN(1/B).  For B = 65535, 127 and 255 this equals the correctly
// rounded IEEE quotient for EVERY input of the format (checked exhaustively on the host by
// tests/test_math_contract.py and on the device against the oracle's plain division), at 3 VALU
// instructions instead of the ~11 of the v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence.
template <int B>
HKD float div_norm(float x) {
  constexpr float b = (float)B, c = 1.0f / (float)B;
  const float q = x * c;
  return fmaf(fmaf(-q, b, x), c, q);
}
HKD f2 unpack2x16unorm(uint32_t u) { return {div_norm<65535>((float)(u & 0xffffu)), div_norm<65535>((float)(u >> 16))}; }
HKD uint32_t snorm8(float x) { return (uint32_t)(int32_t)floorf(0.5f + 127.0f * clamp_(x, -1.0f, 1.0f)) & 0xffu; }
HKD uint32_t pack4x8snorm(f4 v) { return snorm8(v.x) | (snorm8(v.y) << 8) | (snorm8(v.z) << 16) | (snorm8(v.w) << 24); }
HKD float unsnorm8(uint32_t b) { return fmax_(div_norm<127>((float)(int32_t)(int8_t)(uint8_t)b), -1.0f); }
HKD float unorm8(uint32_t b) { return div_norm<255>((float)(b & 0xffu)); }
HKD f4 unpack4x8snorm(uint32_t u) { return {unsnorm8(u & 0xffu), unsnorm8((u >> 8) & 0xffu), unsnorm8((u >> 16) & 0xffu), unsnorm8(u >> 24)}; }
HKD uint2 pack_f16x4(f4 v) { return make_uint2(pack2x16float(v.x, v.y), pack2x16float(v.z, v.w)); }
HKD f4 unpack_f16x4(uint2 u) {
  f2 a = unpack2x16float(u.x), b = unpack2x16float(u.y);
  return {a.x, a.y, b.x, b.y};
}

// WGSL u32(f32) / i32(f32): v_cvt_u32_f32 / v_cvt_i32_f32 saturate and map NaN to 0
HKD uint32_t f32_to_u32(float f) { return (uint32_t)f; }
HKD int f32_to_i32(float f) { return (int)f; }

}  // namespace hkd
