// kernels_aa.hip - the anti-aliasing / upscale tail of PostProcessNode::run (post_process.rs:1236-1272)
//
//   k_smaa_tu4x              smaa.wgsl:81-188    one thread per render pixel = one 2x2 output quad
//   k_smaa_tu4x_extrapolate  smaa.wgsl:239-271   the two off-diagonal pixels of every quad, in place
//   k_taa_jasmine            taa.wgsl:75-170     one thread per output pixel
//
// The reference samples textures through a nearest and a linear sampler (post_process.rs:679-690,
// address mode clamp-to-edge).  Here every plane is a row-major array and the samplers are the
// functions below; the numeric contract fixes what WGSL leaves open: texel
// centres at integer + 0.5, exact f32 bilinear fractions, blend = mix(mix(t00,t10,fx), mix(t01,t11,fx), fy),
// textureGather order (u_min,v_max), (u_max,v_max), (u_max,v_min), (u_min,v_min).
// All three kernels are HBM / L2 gather streams: 5 x (gather4 + nearest) taps of the previous
// G-buffer per pixel, no arithmetic worth the name.
#include <hip/hip_runtime.h>

#include "hk_device.hpp"
#include "hk_kernels.hpp"

namespace hkd {

struct Plane16 {  // rgba16f
  const uint2* __restrict__ p;
  int w, h;
};
struct Plane32 {  // rgba32f
  const float4* __restrict__ p;
  int w, h;
};
struct Footprint {
  int x0, x1, y0, y1;
  float fx, fy;
};

HKD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
HKD void nearest_coords(int w, int h, f2 uv, int* x, int* y) {
  *x = clampi((int)floorf(uv.x * (float)w), 0, w - 1);
  *y = clampi((int)floorf(uv.y * (float)h), 0, h - 1);
}
HKD Footprint footprint(int w, int h, f2 uv) {
  const float px = uv.x * (float)w - 0.5f, py = uv.y * (float)h - 0.5f;
  const float flx = floorf(px), fly = floorf(py);
  Footprint f;
  f.fx = px - flx;
  f.fy = py - fly;
  const int ix = (int)flx, iy = (int)fly;
  f.x0 = clampi(ix, 0, w - 1);
  f.x1 = clampi(ix + 1, 0, w - 1);
  f.y0 = clampi(iy, 0, h - 1);
  f.y1 = clampi(iy + 1, 0, h - 1);
  return f;
}
HKD f4 texel(const Plane16& t, int x, int y) { return unpack_f16x4(t.p[x + t.w * y]); }
HKD f4 texel(const Plane32& t, int x, int y) { return F4(t.p[x + t.w * y]); }
template <typename P>
HKD f4 sample_nearest(const P& t, f2 uv) {
  int x, y;
  nearest_coords(t.w, t.h, uv, &x, &y);
  return texel(t, x, y);
}
HKD f4 sample_linear(const Plane16& t, f2 uv) {
  const Footprint f = footprint(t.w, t.h, uv);
  const f4 t00 = texel(t, f.x0, f.y0), t10 = texel(t, f.x1, f.y0), t01 = texel(t, f.x0, f.y1), t11 = texel(t, f.x1, f.y1);
  return mix4(mix4(t00, t10, f.fx), mix4(t01, t11, f.fx), f.fy);
}
struct PlaneW {  // the w component of a position texture (clip depth) as its own f32 plane
  const float* __restrict__ p;
  int w, h;
};
HKD f4 gather_w(const PlaneW& t, f2 uv) {  // textureGather(3, position-like texture, ..)
  const Footprint f = footprint(t.w, t.h, uv);
  return F4(t.p[f.x0 + t.w * f.y1], t.p[f.x1 + t.w * f.y1], t.p[f.x1 + t.w * f.y0], t.p[f.x0 + t.w * f.y0]);
}
HKD float sample_nearest_w(const PlaneW& t, f2 uv) {
  int x, y;
  nearest_coords(t.w, t.h, uv, &x, &y);
  return t.p[x + t.w * y];
}
HKD void load_loose(const uint2* p, int w, int h, int x, int y, f4* out) {  // textureLoad: zeros out of bounds
  *out = (x >= 0 && y >= 0 && x < w && y < h) ? unpack_f16x4(p[x + w * y]) : F4(0.0f, 0.0f, 0.0f, 0.0f);
}
HKD void store_loose(uint2* p, int w, int h, int x, int y, f4 v) {
  if (x >= 0 && y >= 0 && x < w && y < h) p[x + w * y] = pack_f16x4(v);
}

HKD f3 rgb(f4 a) { return F3(a.x, a.y, a.z); }
HKD f3 clamp01(f3 c) { return F3(clamp_(c.x, 0.0f, 1.0f), clamp_(c.y, 0.0f, 1.0f), clamp_(c.z, 0.0f, 1.0f)); }
HKD f3 sqrt3(f3 a) { return F3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
HKD f3 abs3(f3 a) { return F3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
HKD float length2(f2 a) { return sqrtf(dot(a, a)); }
HKD f3 RGB_to_YCoCg(f3 c) {  // taa.wgsl:20-25, smaa.wgsl:23-28
  const float y = (c.x / 4.0f) + (c.y / 2.0f) + (c.z / 4.0f);
  const float co = (c.x / 2.0f) - (c.z / 2.0f);
  const float cg = (-c.x / 4.0f) + (c.y / 2.0f) - (c.z / 4.0f);
  return F3(y, co, cg);
}
HKD f3 YCoCg_to_RGB(f3 y) {  // taa.wgsl:27-32
  const float r = y.x + y.y - y.z;
  const float g = y.x + y.z;
  const float b = y.x - y.y - y.z;
  return clamp01(F3(r, g, b));
}
HKD f3 clip_towards_aabb_center(f3 previous_color, f3 aabb_min, f3 aabb_max) {  // taa.wgsl:34-42
  const f3 p_clip = 0.5f * (aabb_max + aabb_min);
  const f3 e_clip = 0.5f * (aabb_max - aabb_min);
  const f3 v_clip = previous_color - p_clip;
  const f3 v_unit = v_clip / e_clip;
  const f3 a_unit = abs3(v_unit);
  const float ma_unit = fmax_(a_unit.x, fmax_(a_unit.y, a_unit.z));
  return (ma_unit > 1.0f) ? p_clip + v_clip / ma_unit : previous_color;
}
HKD bool any_lt(f4 a, float s) { return a.x < s || a.y < s || a.z < s || a.w < s; }
HKD bool any_gt(f4 a, float s) { return a.x > s || a.y > s || a.z > s || a.w > s; }
HKD f4 depth_ratio4(float current, f4 previous) {  // select(current / previous, 1.0, previous == 0.0)
  return F4(previous.x == 0.0f ? 1.0f : current / previous.x, previous.y == 0.0f ? 1.0f : current / previous.y,
            previous.z == 0.0f ? 1.0f : current / previous.z, previous.w == 0.0f ? 1.0f : current / previous.w);
}
// taa.wgsl:54-73, smaa.wgsl:54-73
HKD f2 nearest_velocity(const PlaneW& position, const Plane32& velocity_uv, f2 uv, f2 texel_size) {
  f4 depths;
  depths.x = sample_nearest_w(position, uv + F2(texel_size.x, texel_size.y));
  depths.y = sample_nearest_w(position, uv + F2(-texel_size.x, texel_size.y));
  depths.z = sample_nearest_w(position, uv + F2(texel_size.x, -texel_size.y));
  depths.w = sample_nearest_w(position, uv + F2(-texel_size.x, -texel_size.y));
  const float max_depth = fmax_(fmax_(depths.x, depths.y), fmax_(depths.z, depths.w));
  const float depth = sample_nearest_w(position, uv);
  f2 offset = F2(0.0f, 0.0f);
  if (depth < max_depth) {
    const f4 eq = F4(depths.x == max_depth ? 1.0f : 0.0f, depths.y == max_depth ? 1.0f : 0.0f, depths.z == max_depth ? 1.0f : 0.0f,
                     depths.w == max_depth ? 1.0f : 0.0f);
    const float x = dot(F4(texel_size.x, texel_size.x, texel_size.x, texel_size.x), F4(eq.x, -eq.y, eq.z, -eq.w));
    const float y = dot(F4(texel_size.y, texel_size.y, texel_size.y, texel_size.y), F4(eq.x, eq.y, -eq.z, -eq.w));
    offset = F2(x, y);
  }
  const f4 v = sample_nearest(velocity_uv, uv + offset);
  return F2(v.x, v.y);
}

struct AaTargets {
  Plane32 position, velocity_uv, previous_position, previous_velocity_uv;
  PlaneW depth, previous_depth;  // position.w / previous_position.w
  const float2* __restrict__ instance_material;  // full size (position.w x position.h)
  Plane16 render, previous_render;
  uint2* output;
  int ow, oh;
};

__global__ __launch_bounds__(256) void k_taa_jasmine(AaTargets t, float blend, float4 clear_color, int row_begin, int row_end) {
  const Pixel px = pixel_of_thread(t.ow, row_begin, row_end);
  if (!px.valid) return;
  const int x = px.x, y = px.y;
  const f2 size = F2((float)t.ow, (float)t.oh);
  const f2 texel_size = F2(1.0f / size.x, 1.0f / size.y);
  const f2 render_texel = F2(1.0f / (float)t.render.w, 1.0f / (float)t.render.h);
  const f2 uv = F2(((float)x + 0.5f) / size.x, ((float)y + 0.5f) / size.y);
  const f4 original_color = sample_nearest(t.render, uv);
  const f3 current_color = rgb(original_color);
  const f2 velocity = nearest_velocity(t.depth, t.velocity_uv, uv, render_texel);
  const f2 previous_uv = uv - velocity;
  const bool boundary_miss = fabsf(previous_uv.x - 0.5f) > 0.5f || fabsf(previous_uv.y - 0.5f) > 0.5f;
  const f4 current_position_depth = sample_nearest(t.position, uv);
  bool has_content = current_position_depth.w > 0.0f;
  bool depth_miss = current_position_depth.w == 0.0f;
  bool position_miss = current_position_depth.w == 0.0f;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const f2 bias = i == 0 ? F2(0.0f, 0.0f) : F2((i & 1) ? 1.5f : -1.5f, i <= 2 ? 1.5f : -1.5f) * texel_size;
    const f4 previous_depths = gather_w(t.previous_depth, previous_uv + bias);
    const f4 depth_ratio = depth_ratio4(current_position_depth.w, previous_depths);
    has_content = has_content || any_gt(previous_depths, 0.0f);
    depth_miss = depth_miss || any_lt(depth_ratio, 0.95f);
    const f3 pp = xyz(sample_nearest(t.previous_position, previous_uv + bias));
    position_miss = position_miss || length(xyz(current_position_depth) - pp) > 0.5f;
  }
  if (!has_content) {
    t.output[x + t.ow * y] = pack_f16x4(F4(clear_color));
    return;
  }
  const f4 pv = sample_nearest(t.previous_velocity_uv, previous_uv);
  const bool velocity_miss = length2(velocity - F2(pv.x, pv.y)) > 0.00005f;
  // 5-tap Catmull-Rom reprojection, taa.wgsl:127-144
  const f2 sample_position = (uv - velocity) * size;
  const f2 texel_position_1 = F2(floorf(sample_position.x - 0.5f) + 0.5f, floorf(sample_position.y - 0.5f) + 0.5f);
  const f2 f = sample_position - texel_position_1;
  auto poly = [](float fx, float a, float b, float cc) { return a + fx * (b + cc * fx); };
  const f2 w0 = F2(f.x * poly(f.x, -0.5f, 1.0f, -0.5f), f.y * poly(f.y, -0.5f, 1.0f, -0.5f));
  const f2 w1 = F2(1.0f + f.x * f.x * (-2.5f + 1.5f * f.x), 1.0f + f.y * f.y * (-2.5f + 1.5f * f.y));
  const f2 w2 = F2(f.x * poly(f.x, 0.5f, 2.0f, -1.5f), f.y * poly(f.y, 0.5f, 2.0f, -1.5f));
  const f2 w3 = F2(f.x * f.x * (-0.5f + 0.5f * f.x), f.y * f.y * (-0.5f + 0.5f * f.y));
  const f2 w12 = w1 + w2;
  const f2 offset12 = w2 / (w1 + w2);
  const f2 tp0 = (texel_position_1 - 1.0f) * texel_size;
  const f2 tp3 = (texel_position_1 + 2.0f) * texel_size;
  const f2 tp12 = (texel_position_1 + offset12) * texel_size;
  auto prev = [&](float u, float v) { return clamp01(rgb(sample_linear(t.previous_render, F2(u, v)))); };
  f3 previous_color = F3(0.0f, 0.0f, 0.0f);
  previous_color = previous_color + prev(tp12.x, tp0.y) * w12.x * w0.y;
  previous_color = previous_color + prev(tp0.x, tp12.y) * w0.x * w12.y;
  previous_color = previous_color + prev(tp12.x, tp12.y) * w12.x * w12.y;
  previous_color = previous_color + prev(tp3.x, tp12.y) * w3.x * w12.y;
  previous_color = previous_color + prev(tp12.x, tp3.y) * w12.x * w3.y;
  if (boundary_miss || (position_miss && velocity_miss && depth_miss)) {  // 3x3 YCoCg variance clipping, taa.wgsl:146-164
    auto smp = [&](f2 p) { return RGB_to_YCoCg(clamp01(rgb(sample_nearest(t.render, p)))); };
    const f3 s_tl = smp(uv + F2(-texel_size.x, texel_size.y));
    const f3 s_tm = smp(uv + F2(0.0f, texel_size.y));
    const f3 s_tr = smp(uv + texel_size);
    const f3 s_ml = smp(uv - F2(texel_size.x, 0.0f));
    const f3 s_mm = RGB_to_YCoCg(current_color);
    const f3 s_mr = smp(uv + F2(texel_size.x, 0.0f));
    const f3 s_bl = smp(uv - texel_size);
    const f3 s_bm = smp(uv - F2(0.0f, texel_size.y));
    const f3 s_br = smp(uv + F2(texel_size.x, -texel_size.y));
    const f3 moment_1 = s_tl + s_tm + s_tr + s_ml + s_mm + s_mr + s_bl + s_bm + s_br;
    const f3 moment_2 = (s_tl * s_tl) + (s_tm * s_tm) + (s_tr * s_tr) + (s_ml * s_ml) + (s_mm * s_mm) + (s_mr * s_mr) + (s_bl * s_bl) + (s_bm * s_bm) +
                        (s_br * s_br);
    const f3 mean = moment_1 / 9.0f;
    const f3 variance = sqrt3((moment_2 / 9.0f) - (mean * mean));
    previous_color = RGB_to_YCoCg(previous_color);
    previous_color = clip_towards_aabb_center(previous_color, mean - variance, mean + variance);
    previous_color = YCoCg_to_RGB(previous_color);
  }
  const f3 out = mix(previous_color, current_color, blend);  // taa.wgsl:167
  t.output[x + t.ow * y] = pack_f16x4(F4(out, original_color.w));
}

__global__ __launch_bounds__(256) void k_smaa_tu4x(AaTargets t, uint32_t frame_number, int row_begin, int row_end) {
  const Pixel px = pixel_of_thread(t.render.w, row_begin, row_end);
  if (!px.valid) return;
  const int x = px.x, y = px.y;
  const f2 input_size = F2((float)t.render.w, (float)t.render.h), output_size = F2((float)t.ow, (float)t.oh);
  const f2 texel_size = F2(1.0f / output_size.x, 1.0f / output_size.y);
  const f2 deferred_texel = F2(1.0f / (float)t.position.w, 1.0f / (float)t.position.h);
  const int current_jitter = (frame_number & 1u) == 0u ? 0 : 1;   // smaa.wgsl:75-77
  const int previous_jitter = (frame_number & 1u) == 0u ? 1 : 0;  // smaa.wgsl:79-81
  const float TAU = 6.283185307f;
  const f2 uv = F2(((float)x + 0.5f) / input_size.x, ((float)y + 0.5f) / input_size.y);
  const int cox = 2 * x + current_jitter, coy = 2 * y + current_jitter;
  const f3 current_color = rgb(sample_nearest(t.render, uv));
  const int pox = 2 * x + previous_jitter, poy = 2 * y + previous_jitter;
  const f2 previous_output_uv = F2(((float)pox + 0.5f) / output_size.x, ((float)poy + 0.5f) / output_size.y);
  const f2 velocity = nearest_velocity(t.depth, t.velocity_uv, previous_output_uv, deferred_texel);
  const f2 previous_reprojected_uv = previous_output_uv - velocity;
  f3 previous_color = rgb(sample_nearest(t.previous_render, previous_reprojected_uv));
  const bool boundary_miss = fabsf(previous_reprojected_uv.x - 0.5f) > 0.5f || fabsf(previous_reprojected_uv.y - 0.5f) > 0.5f;
  auto instance_at = [&](f2 p) {
    int ix, iy;
    nearest_coords(t.position.w, t.position.h, p, &ix, &iy);
    return t.instance_material[ix + t.position.w * iy].x;
  };
  const float current_instance = instance_at(previous_output_uv);
  bool instance_miss = false;
  const float current_depth = sample_nearest_w(t.depth, previous_output_uv);
  bool depth_miss = current_depth == 0.0f;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const f2 bias = i == 0 ? F2(0.0f, 0.0f) : F2((i & 1) ? 2.5f : -2.5f, i <= 2 ? 2.5f : -2.5f) * texel_size;
    const f4 previous_depths = gather_w(t.previous_depth, previous_reprojected_uv + bias);
    const f4 depth_ratio = depth_ratio4(current_depth, previous_depths);
    const bool any_ratio = any_lt(depth_ratio, 0.95f);
    depth_miss = depth_miss || any_ratio;
    const float previous_instance = instance_at(previous_reprojected_uv + bias);  // the CURRENT instance texture, smaa.wgsl:149
    instance_miss = instance_miss || (any_ratio && fabsf(previous_instance - current_instance) > 1.0f);
  }
  const f4 pv = sample_nearest(t.previous_velocity_uv, previous_reprojected_uv);
  const bool velocity_miss = length2(velocity - F2(pv.x, pv.y)) > 0.0001f;
  if (boundary_miss || ((depth_miss || instance_miss) && velocity_miss)) {  // 2x2 YCoCg variance clipping, smaa.wgsl:156-184
    f2 uv_bias = F2(0.0f, 0.0f);
    float min_ds = 10.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const f2 bias = i == 0 ? F2(0.0f, 0.0f) : F2((i & 1) ? 2.5f : -2.5f, i <= 2 ? 2.5f : -2.5f) * texel_size;
      const f4 ds = gather_w(t.depth, previous_output_uv + bias);
      const f4 d = F4(current_depth - ds.x, current_depth - ds.y, current_depth - ds.z, current_depth - ds.w);
      const float dds = sqrtf(dot(d, d));
      if (dds < min_ds) uv_bias = bias;
      min_ds = fmin_(min_ds, dds);
    }
    const Footprint fp = footprint(t.render.w, t.render.h, previous_output_uv + uv_bias);
    const f4 g0 = texel(t.render, fp.x0, fp.y1), g1 = texel(t.render, fp.x1, fp.y1), g2 = texel(t.render, fp.x1, fp.y0), g3 = texel(t.render, fp.x0, fp.y0);
    const f3 s1 = RGB_to_YCoCg(rgb(g0));
    const f3 s2 = RGB_to_YCoCg(rgb(g1));
    const f3 s3 = RGB_to_YCoCg(rgb(g2));
    const f3 s4 = RGB_to_YCoCg(rgb(g3));
    const f3 moment_1 = s1 + s2 + s3 + s4;
    const f3 moment_2 = s1 * s1 + s2 * s2 + s3 * s3 + s4 * s4;
    const f3 mean = moment_1 / 4.0f;
    const f3 variance = sqrt3((moment_2 / 4.0f) - (mean * mean));
    previous_color = RGB_to_YCoCg(previous_color);
    previous_color = clip_towards_aabb_center(previous_color, mean - variance, mean + variance);
    previous_color = YCoCg_to_RGB(previous_color);
  }
  // sub-pixel velocity blend, smaa.wgsl:186-193
  const f2 sv = F2(fract(velocity.x / (2.0f * texel_size.x)), fract(velocity.y / (2.0f * texel_size.y)));
  float blend_factor = fmax_(sv.x, sv.y);
  blend_factor = clamp_(-cos_(blend_factor * TAU), 0.0f, 1.0f);
  const f3 remix_color = rgb(sample_linear(t.render, previous_output_uv));
  previous_color = mix(previous_color, remix_color, blend_factor);
  store_loose(t.output, t.ow, t.oh, cox, coy, F4(current_color, 1.0f));
  store_loose(t.output, t.ow, t.oh, pox, poy, F4(previous_color, 1.0f));
}

HKD f3 differential_blend_factor(f4 t, f4 b, f4 n, f4 e, f4 s, f4 w) {  // smaa.wgsl:198-222
  const f2 dh = F2(luminance(abs3(rgb(w) - rgb(b))), luminance(abs3(rgb(t) - rgb(e))));
  const f2 dv = F2(luminance(abs3(rgb(t) - rgb(s))), luminance(abs3(rgb(n) - rgb(b))));
  const f2 factor_xy = F2(fmax_(dv.x, 0.001f) * fmax_(dv.y, 0.001f), fmax_(dh.x, 0.001f) * fmax_(dh.y, 0.001f));
  const float factor_z = 1.0f / (factor_xy.x + factor_xy.y);
  return F3(factor_xy.x, factor_xy.y, factor_z);
}
HKD f4 differential_blend(f4 t, f4 b, f4 l, f4 r, f3 factor) {  // smaa.wgsl:224-235
  f4 color = F4(0.0f, 0.0f, 0.0f, 0.0f);
  color = color + (l + r) * factor.x;
  color = color + (t + b) * factor.y;
  return (0.5f * factor.z) * color;
}
// Reads only the diagonal pixels k_smaa_tu4x wrote, writes only the off-diagonal ones: in place, no hazard.
__global__ __launch_bounds__(256) void k_smaa_tu4x_extrapolate(uint2* output, int ow, int oh, int render_w, int row_begin, int row_end) {
  const Pixel px = pixel_of_thread(render_w, row_begin, row_end);
  if (!px.valid) return;
  const int bx = 2 * px.x, by = 2 * px.y;
  f4 t_color, b_color, n_color, e_color, s_color, w_color;
  load_loose(output, ow, oh, bx, by, &t_color);
  load_loose(output, ow, oh, bx + 1, by + 1, &b_color);
  load_loose(output, ow, oh, bx + 1, by - 1, &n_color);
  load_loose(output, ow, oh, bx + 2, by, &e_color);
  load_loose(output, ow, oh, bx, by + 2, &s_color);
  load_loose(output, ow, oh, bx - 1, by + 1, &w_color);
  const f3 factor = differential_blend_factor(t_color, b_color, n_color, e_color, s_color, w_color);
  const f4 x_color = differential_blend(t_color, s_color, w_color, b_color, factor);
  const f4 y_color = differential_blend(n_color, b_color, t_color, e_color, factor);
  store_loose(output, ow, oh, bx, by + 1, x_color);
  store_loose(output, ow, oh, bx + 1, by, y_color);
}

}  // namespace hkd

namespace hk {
using namespace hkd;

static AaTargets make_targets(const AaBuffers& b) {
  AaTargets t;
  t.position = Plane32{(const float4*)b.position, b.full_w, b.full_h};
  t.velocity_uv = Plane32{(const float4*)b.velocity_uv, b.full_w, b.full_h};
  t.previous_position = Plane32{(const float4*)b.previous_position, b.full_w, b.full_h};
  t.previous_velocity_uv = Plane32{(const float4*)b.previous_velocity_uv, b.full_w, b.full_h};
  t.depth = PlaneW{b.depth, b.full_w, b.full_h};
  t.previous_depth = PlaneW{b.previous_depth, b.full_w, b.full_h};
  t.instance_material = (const float2*)b.instance_material;
  t.render = Plane16{(const uint2*)b.render, b.render_w, b.render_h};
  t.previous_render = Plane16{(const uint2*)b.previous_render, b.previous_w, b.previous_h};
  t.output = (uint2*)b.output;
  t.ow = b.out_w;
  t.oh = b.out_h;
  return t;
}

void launch_smaa_tu4x(hipStream_t st, const AaBuffers& b, uint32_t frame_number, int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_smaa_tu4x, grid_for(b.render_w, y1 - y0), dim3(256), 0, st, make_targets(b), frame_number, y0, y1);
}
void launch_smaa_tu4x_extrapolate(hipStream_t st, void* output, int out_w, int out_h, int render_w, int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_smaa_tu4x_extrapolate, grid_for(render_w, y1 - y0), dim3(256), 0, st, (uint2*)output, out_w, out_h, render_w, y0, y1);
}
void launch_taa_jasmine(hipStream_t st, const AaBuffers& b, float blend, const float clear_color[4], int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_taa_jasmine, grid_for(b.out_w, y1 - y0), dim3(256), 0, st, make_targets(b), blend,
                     make_float4(clear_color[0], clear_color[1], clear_color[2], clear_color[3]), y0, y1);
}

}  // namespace hk
