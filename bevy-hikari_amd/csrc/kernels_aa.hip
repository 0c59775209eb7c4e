// kernels_aa.hip - the anti-aliasing / upscale tail of PostProcessNode::run (post_process.rs:1236-1272)
//
//   k_smaa_tu4x              smaa.wgsl:81-188    one thread per render pixel = one 2x2 output quad
//   k_smaa_tu4x_extrapolate  smaa.wgsl:239-271   the two off-diagonal pixels of every quad, in place
//   k_taa_jasmine            taa.wgsl:75-170     one thread per output pixel
//   k_fsr_easu / k_fsr_rcas  fsr/source.zip      FSR1 upscale + sharpen, one thread per window pixel
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

// Pixel -> wave mapping of the post-process kernels (hk_kernels.hpp pixel_of_thread_rows; results cannot depend on it, only who
// computes which pixel).  Measured on Cornell 1080p traced -> 3840x2160 SMAA Tu4x + TAA (tools/config_probe.py aa), 8 x 8 tiles
// in XCD bands -> 64 x 1 rows in bands -> 64 x 1 rows round-robin:  k_smaa_tu4x 0.079 -> 0.063 -> 0.061 ms,
// k_taa_jasmine 0.206 -> 0.201 -> 0.180 ms, k_smaa_tu4x_extrapolate 0.031 -> 0.030 -> 0.033 ms.  The two FSR kernels have not
// been measured and keep the tiles unless built with -DHK_FSR_ROWS_W=64 [-DHK_FSR_ROWS_XCD=0].
#ifdef HK_FSR_ROWS_W
#ifndef HK_FSR_ROWS_XCD
#define HK_FSR_ROWS_XCD 1
#endif
#define HK_FSR_PIXEL(width, row_begin, row_end) pixel_of_thread_rows<HK_FSR_ROWS_W, HK_FSR_ROWS_XCD != 0>(width, row_begin, row_end)
#define HK_FSR_GRID(width, rows) grid_for_rows(HK_FSR_ROWS_W, width, rows)
#else
#define HK_FSR_PIXEL(width, row_begin, row_end) pixel_of_thread(width, row_begin, row_end)
#define HK_FSR_GRID(width, rows) grid_for(width, rows)
#endif

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
  const Pixel px = pixel_of_thread_rows<64, false>(t.ow, row_begin, row_end);
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
  const Pixel px = pixel_of_thread_rows<64, false>(t.render.w, row_begin, row_end);
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
  const Pixel px = pixel_of_thread_rows<64, true>(render_w, row_begin, row_end);
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

// ------------------------------------------------------------------------------------------
// FidelityFX Super Resolution 1.0 (Upscale::Fsr1), the f32 path of src/shaders/fsr/source.zip:
// ffx_fsr1.h FsrEasuF :315-441 (k_fsr_easu, one thread per window pixel, 12 taps of the scaled image) and
// FsrRcasF :684-768 (k_fsr_rcas, 5-tap cross).  fakeTextureGather (texture_gather.glsl) reads the four texels
// around a texel corner through four bilinear samples at texel centres; the contract takes the texels.
// ------------------------------------------------------------------------------------------
HKD float fsr_prx_lo_rcp(float a) { return u2f(0x7ef07ebbu - f2u(a)); }                                           // ffx_a.h APrxLoRcpF1
HKD float fsr_prx_med_rcp(float a) { const float b = u2f(0x7ef19fffu - f2u(a)); return b * (-b * a + 2.0f); }    // APrxMedRcpF1
HKD float fsr_prx_lo_rsq(float a) { return u2f(0x5f347d74u - (f2u(a) >> 1)); }                                    // APrxLoRsqF1
HKD float fsr_min3(float x, float y, float z) { return fmin_(x, fmin_(y, z)); }
HKD float fsr_max3(float x, float y, float z) { return fmax_(x, fmax_(y, z)); }
struct FsrGather { f4 r, g, b; };
HKD FsrGather fsr_gather(const Plane16& t, f2 p, f2 ps) {
  const f4 s3 = sample_nearest(t, F2(p.x + ps.x, p.y + ps.y));
  const f4 s1 = sample_nearest(t, F2(p.x - ps.x, p.y - ps.y));
  const f4 s2 = sample_nearest(t, F2(p.x + ps.x, p.y + (-ps.y)));
  const f4 s4 = sample_nearest(t, F2(p.x + (-ps.x), p.y + ps.y));
  FsrGather o;
  o.r = F4(s4.x, s3.x, s2.x, s1.x);
  o.g = F4(s4.y, s3.y, s2.y, s1.y);
  o.b = F4(s4.z, s3.z, s2.z, s1.z);
  return o;
}
HKD void fsr_easu_tap(f3* aC, float* aW, f2 off, f2 dir, f2 len, float lob, float clp, f3 c) {  // ffx_fsr1.h:239-272
  f2 v;
  v.x = (off.x * (dir.x)) + (off.y * dir.y);
  v.y = (off.x * (-dir.y)) + (off.y * dir.x);
  v = v * len;
  float d2 = v.x * v.x + v.y * v.y;
  d2 = fmin_(d2, clp);
  float wB = (float)(2.0 / 5.0) * d2 + -1.0f;
  float wA = lob * d2 + -1.0f;
  wB *= wB;
  wA *= wA;
  wB = (float)(25.0 / 16.0) * wB + (float)(-(25.0 / 16.0 - 1.0));
  const float w = wB * wA;
  *aC = *aC + c * w;
  *aW += w;
}
HKD void fsr_easu_set(f2* dir, float* len, f2 pp, int corner, float lA, float lB, float lC, float lD, float lE) {  // ffx_fsr1.h:275-312
  float w = 0.0f;
  if (corner == 0) w = (1.0f - pp.x) * (1.0f - pp.y);
  if (corner == 1) w = pp.x * (1.0f - pp.y);
  if (corner == 2) w = (1.0f - pp.x) * pp.y;
  if (corner == 3) w = pp.x * pp.y;
  const float dc = lD - lC;
  const float cb = lC - lB;
  float lenX = fmax_(fabsf(dc), fabsf(cb));
  lenX = fsr_prx_lo_rcp(lenX);
  const float dirX = lD - lB;
  dir->x += dirX * w;
  lenX = clamp_(fabsf(dirX) * lenX, 0.0f, 1.0f);
  lenX *= lenX;
  *len += lenX * w;
  const float ec = lE - lC;
  const float ca = lC - lA;
  float lenY = fmax_(fabsf(ec), fabsf(ca));
  lenY = fsr_prx_lo_rcp(lenY);
  const float dirY = lE - lA;
  dir->y += dirY * w;
  lenY = clamp_(fabsf(dirY) * lenY, 0.0f, 1.0f);
  lenY *= lenY;
  *len += lenY * w;
}
HKD f4 fsr_luma(const FsrGather& t) { return t.b * 0.5f + (t.r * 0.5f + t.g); }
struct FsrEasuArgs {
  Plane16 input;
  uint2* output;
  int ow, oh;
  float con0[4], con1[4], con2[4], con3[2];  // FsrEasuCon, evaluated on the host in f32
  f2 half_texel;
};
__global__ __launch_bounds__(256) void k_fsr_easu(FsrEasuArgs a, int row_begin, int row_end) {
  const Pixel px = HK_FSR_PIXEL(a.ow, row_begin, row_end);
  if (!px.valid) return;
  f2 pp = F2((float)px.x * a.con0[0] + a.con0[2], (float)px.y * a.con0[1] + a.con0[3]);
  const f2 fp = F2(floorf(pp.x), floorf(pp.y));
  pp = pp - fp;
  const f2 p0 = F2(fp.x * a.con1[0] + a.con1[2], fp.y * a.con1[1] + a.con1[3]);
  const f2 p1 = F2(p0.x + a.con2[0], p0.y + a.con2[1]);
  const f2 p2 = F2(p0.x + a.con2[2], p0.y + a.con2[3]);
  const f2 p3 = F2(p0.x + a.con3[0], p0.y + a.con3[1]);
  const FsrGather bczz = fsr_gather(a.input, p0, a.half_texel), ijfe = fsr_gather(a.input, p1, a.half_texel),
                  klhg = fsr_gather(a.input, p2, a.half_texel), zzon = fsr_gather(a.input, p3, a.half_texel);
  const f4 bczzL = fsr_luma(bczz), ijfeL = fsr_luma(ijfe), klhgL = fsr_luma(klhg), zzonL = fsr_luma(zzon);
  const float bL = bczzL.x, cL = bczzL.y, iL = ijfeL.x, jL = ijfeL.y, fL = ijfeL.z, eL = ijfeL.w, kL = klhgL.x, lL = klhgL.y, hL = klhgL.z,
              gL = klhgL.w, oL = zzonL.z, nL = zzonL.w;
  f2 dir = F2(0.0f, 0.0f);
  float len = 0.0f;
  fsr_easu_set(&dir, &len, pp, 0, bL, eL, fL, gL, jL);
  fsr_easu_set(&dir, &len, pp, 1, cL, fL, gL, hL, kL);
  fsr_easu_set(&dir, &len, pp, 2, fL, iL, jL, kL, nL);
  fsr_easu_set(&dir, &len, pp, 3, gL, jL, kL, lL, oL);
  const f2 dir2 = dir * dir;
  float dirR = dir2.x + dir2.y;
  const bool zro = dirR < (float)(1.0 / 32768.0);
  dirR = fsr_prx_lo_rsq(dirR);
  dirR = zro ? 1.0f : dirR;
  dir.x = zro ? 1.0f : dir.x;
  dir = dir * F2(dirR, dirR);
  len = len * 0.5f;
  len *= len;
  const float stretch = (dir.x * dir.x + dir.y * dir.y) * fsr_prx_lo_rcp(fmax_(fabsf(dir.x), fabsf(dir.y)));
  const f2 len2 = F2(1.0f + (stretch - 1.0f) * len, 1.0f + -0.5f * len);
  const float lob = 0.5f + (float)((1.0 / 4.0 - 0.04) - 0.5) * len;
  const float clp = fsr_prx_lo_rcp(lob);
  const f3 fC = F3(ijfe.r.z, ijfe.g.z, ijfe.b.z), gC = F3(klhg.r.w, klhg.g.w, klhg.b.w), jC = F3(ijfe.r.y, ijfe.g.y, ijfe.b.y),
           kC = F3(klhg.r.x, klhg.g.x, klhg.b.x);
  const f3 min4 = min3(F3(fsr_min3(fC.x, gC.x, jC.x), fsr_min3(fC.y, gC.y, jC.y), fsr_min3(fC.z, gC.z, jC.z)), kC);
  const f3 max4 = max3(F3(fsr_max3(fC.x, gC.x, jC.x), fsr_max3(fC.y, gC.y, jC.y), fsr_max3(fC.z, gC.z, jC.z)), kC);
  f3 aC = F3(0.0f, 0.0f, 0.0f);
  float aW = 0.0f;
#define HK_EASU_TAP(ox, oy, G, ch) fsr_easu_tap(&aC, &aW, F2((ox) - pp.x, (oy) - pp.y), dir, len2, lob, clp, F3(G.r.ch, G.g.ch, G.b.ch))
  HK_EASU_TAP(0.0f, -1.0f, bczz, x);  // b
  HK_EASU_TAP(1.0f, -1.0f, bczz, y);  // c
  HK_EASU_TAP(-1.0f, 1.0f, ijfe, x);  // i
  HK_EASU_TAP(0.0f, 1.0f, ijfe, y);   // j
  HK_EASU_TAP(0.0f, 0.0f, ijfe, z);   // f
  HK_EASU_TAP(-1.0f, 0.0f, ijfe, w);  // e
  HK_EASU_TAP(1.0f, 1.0f, klhg, x);   // k
  HK_EASU_TAP(2.0f, 1.0f, klhg, y);   // l
  HK_EASU_TAP(2.0f, 0.0f, klhg, z);   // h
  HK_EASU_TAP(1.0f, 0.0f, klhg, w);   // g
  HK_EASU_TAP(1.0f, 2.0f, zzon, z);   // o
  HK_EASU_TAP(0.0f, 2.0f, zzon, w);   // n
#undef HK_EASU_TAP
  const f3 pix = min3(max4, max3(min4, aC * (1.0f / aW)));
  a.output[px.x + a.ow * px.y] = pack_f16x4(F4(pix.x, pix.y, pix.z, 1.0f));  // hdr == 0 (post_process.rs:531)
}
__global__ __launch_bounds__(256) void k_fsr_rcas(const uint2* __restrict__ input, uint2* __restrict__ output, int w, int h, float sharpness,
                                                  int row_begin, int row_end) {
  const Pixel px = HK_FSR_PIXEL(w, row_begin, row_end);
  if (!px.valid) return;
  const float sharp = exp2_(-sharpness);  // FsrRcasCon, ffx_fsr1.h:662-673
  f4 b, d, e, f, hh;
  load_loose(input, w, h, px.x, px.y - 1, &b);
  load_loose(input, w, h, px.x - 1, px.y, &d);
  load_loose(input, w, h, px.x, px.y, &e);
  load_loose(input, w, h, px.x + 1, px.y, &f);
  load_loose(input, w, h, px.x, px.y + 1, &hh);
  const float mn4R = fmin_(fsr_min3(b.x, d.x, f.x), hh.x), mn4G = fmin_(fsr_min3(b.y, d.y, f.y), hh.y), mn4B = fmin_(fsr_min3(b.z, d.z, f.z), hh.z);
  const float mx4R = fmax_(fsr_max3(b.x, d.x, f.x), hh.x), mx4G = fmax_(fsr_max3(b.y, d.y, f.y), hh.y), mx4B = fmax_(fsr_max3(b.z, d.z, f.z), hh.z);
  const float peakCx = 1.0f, peakCy = -1.0f * 4.0f;
  const float hitMinR = fmin_(mn4R, e.x) * (1.0f / (4.0f * mx4R)), hitMinG = fmin_(mn4G, e.y) * (1.0f / (4.0f * mx4G)),
              hitMinB = fmin_(mn4B, e.z) * (1.0f / (4.0f * mx4B));
  const float hitMaxR = (peakCx - fmax_(mx4R, e.x)) * (1.0f / (4.0f * mn4R + peakCy)),
              hitMaxG = (peakCx - fmax_(mx4G, e.y)) * (1.0f / (4.0f * mn4G + peakCy)),
              hitMaxB = (peakCx - fmax_(mx4B, e.z)) * (1.0f / (4.0f * mn4B + peakCy));
  const float lobeR = fmax_(-hitMinR, hitMaxR), lobeG = fmax_(-hitMinG, hitMaxG), lobeB = fmax_(-hitMinB, hitMaxB);
  const float lobe = fmax_(-(float)(0.25 - (1.0 / 16.0)), fmin_(fsr_max3(lobeR, lobeG, lobeB), 0.0f)) * sharp;
  const float rcpL = fsr_prx_med_rcp(4.0f * lobe + 1.0f);
  const float pixR = (lobe * b.x + lobe * d.x + lobe * hh.x + lobe * f.x + e.x) * rcpL;
  const float pixG = (lobe * b.y + lobe * d.y + lobe * hh.y + lobe * f.y + e.y) * rcpL;
  const float pixB = (lobe * b.z + lobe * d.z + lobe * hh.z + lobe * f.z + e.z) * rcpL;
  output[px.x + w * px.y] = pack_f16x4(F4(pixR, pixG, pixB, 1.0f));
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
  hipLaunchKernelGGL(k_smaa_tu4x, grid_for_rows(64, b.render_w, y1 - y0), dim3(256), 0, st, make_targets(b), frame_number, y0, y1);
}
void launch_smaa_tu4x_extrapolate(hipStream_t st, void* output, int out_w, int out_h, int render_w, int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_smaa_tu4x_extrapolate, grid_for_rows(64, render_w, y1 - y0), dim3(256), 0, st, (uint2*)output, out_w, out_h, render_w, y0, y1);
}
void launch_taa_jasmine(hipStream_t st, const AaBuffers& b, float blend, const float clear_color[4], int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_taa_jasmine, grid_for_rows(64, b.out_w, y1 - y0), dim3(256), 0, st, make_targets(b), blend,
                     make_float4(clear_color[0], clear_color[1], clear_color[2], clear_color[3]), y0, y1);
}

void launch_fsr_easu(hipStream_t st, const void* input, int in_w, int in_h, void* output, int out_w, int out_h, int y0, int y1) {
  if (y1 <= y0) return;
  FsrEasuArgs a;
  a.input = Plane16{(const uint2*)input, in_w, in_h};
  a.output = (uint2*)output;
  a.ow = out_w;
  a.oh = out_h;
  // FsrEasuCon (ffx_fsr1.h:156-203) with inputViewport = inputSize = the scaled size (post_process.rs:522-530)
  const float ivx = (float)in_w, ivy = (float)in_h, isx = ivx, isy = ivy, osx = (float)out_w, osy = (float)out_h;
  a.con0[0] = ivx * (1.0f / osx);
  a.con0[1] = ivy * (1.0f / osy);
  a.con0[2] = 0.5f * ivx * (1.0f / osx) - 0.5f;
  a.con0[3] = 0.5f * ivy * (1.0f / osy) - 0.5f;
  a.con1[0] = 1.0f / isx;
  a.con1[1] = 1.0f / isy;
  a.con1[2] = 1.0f * (1.0f / isx);
  a.con1[3] = -1.0f * (1.0f / isy);
  a.con2[0] = -1.0f * (1.0f / isx);
  a.con2[1] = 2.0f * (1.0f / isy);
  a.con2[2] = 1.0f * (1.0f / isx);
  a.con2[3] = 2.0f * (1.0f / isy);
  a.con3[0] = 0.0f * (1.0f / isx);
  a.con3[1] = 4.0f * (1.0f / isy);
  a.half_texel = f2{(1.0f / (float)in_w) / 2.0f, (1.0f / (float)in_h) / 2.0f};
  hipLaunchKernelGGL(k_fsr_easu, HK_FSR_GRID(out_w, y1 - y0), dim3(256), 0, st, a, y0, y1);
}
void launch_fsr_rcas(hipStream_t st, const void* input, void* output, int w, int h, float sharpness, int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_fsr_rcas, HK_FSR_GRID(w, y1 - y0), dim3(256), 0, st, (const uint2*)input, (uint2*)output, w, h, sharpness, y0, y1);
}

}  // namespace hk
