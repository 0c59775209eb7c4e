// kernels_denoise.hip - demodulation + 4-level a-trous filter (denoise.wgsl:135-319) for gfx950.
//
// The reference runs, per render channel, demodulation then denoise L0..L3 (post_process.rs:
// 1190-1224): 15 dispatches per frame, each re-reading the same G-buffer taps (normal, depth,
// instance) from three planes.  Here
//   * the per-tap geometry lives in one 16-B record per pixel (`dn_g` = the NORMALISED stored normal and
//     the instance id) plus the 4-B depth plane, derived once per frame (k_derive_planes / k_prepass): a
//     tap costs a dwordx4 + a dword instead of three strided loads, and the normalisation (IEEE sqrt +
//     divide, 29 instructions) is done once per pixel instead of once per tap per level per launch;
//   * all channels of one level run in ONE launch (template NCH): the geometric weights
//     w_normal * w_depth * w_instance of a tap are channel-independent and computed once, only the
//     luminance weight and the accumulation are per channel.  5 launches per frame instead of 15.
// Per channel the arithmetic and its order are exactly the reference's, so results are bit-identical
// to running the channels one after another (tests: nodes path == frame path, GPU == oracle).
#include <hip/hip_runtime.h>

#include "hk_device.hpp"
#include "hk_kernels.hpp"

#ifndef HK_DENOISE_W
#define HK_DENOISE_W 64  // wave shape of the a-trous levels: 64 x 1 pixels (pixel_of_thread_rows)
#endif

namespace hkd {

// depth plane (f32) for the spatial-reuse ray march + packed denoise geometry, from the G-buffer
__global__ __launch_bounds__(256) void k_derive_planes(GBuffer g, float* __restrict__ depth_plane, float4* __restrict__ dn_g, int width, int row_begin,
                                                       int row_end) {
  const Pixel px = pixel_of_thread(width, row_begin, row_end);
  if (!px.valid) return;
  const int idx = px.x + width * px.y;
  const float depth = g.position[idx].w;
  depth_plane[idx] = depth;
  dn_g[idx] = denoise_geometry(g.normal[idx], g.instance_material[idx].x);
}

// geometry pixels per row of the depth plane (hk_row_costs): one workgroup per row
__global__ __launch_bounds__(256) void k_count_geometry_rows(const float* __restrict__ depth, int width, uint32_t* __restrict__ out) {
  __shared__ uint32_t partial[4];
  const int y = (int)blockIdx.x;
  uint32_t n = 0;
  for (int x = (int)threadIdx.x; x < width; x += 256) n += !(depth[(size_t)y * width + x] < HK_F32_EPSILON) ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) out[y] = partial[0] + partial[1] + partial[2] + partial[3];
}

template <int NCH>
__global__ __launch_bounds__(256) void k_demodulation(DFrame fr, DemodTargets d, int row_begin, int row_end) {  // denoise.wgsl:135-162
  const Pixel px = pixel_of_thread_rows<64, true>(fr.rw, row_begin, row_end);
  if (!px.valid) return;
  const int x = px.x, y = px.y, index = x + fr.rw * y;
  const f2 uv = coords_to_uv(fr, x, y);
  const f2 deferred_uv = jittered_deferred_uv(fr, uv, 0.5f);
  int ax, ay, rx, ry;
  nearest_coords(deferred_uv, fr.dw, fr.dh, &ax, &ay);
  const f3 albedo = xyz(unpack_f16x4(d.albedo[ax + fr.dw * ay]));
  nearest_coords(uv, fr.rw, fr.rh, &rx, &ry);
  uint2 render_in[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) render_in[ch] = d.render[ch][rx + fr.rw * ry];
  // The 3 x 3 variance taps (denoise.wgsl:152-160).  All of a pixel's loads are issued before the first is used - a tap outside the
  // image reads the pixel's own (valid) address and is dropped afterwards - so that a wave waits for memory once, not once per tap;
  // the sum itself runs in the reference's order (x outer, y inner) with the reference's conditions.
  float tap_variance[9][NCH];
  bool tap_inside[9];
#pragma unroll
  for (int ox = -1; ox <= 1; ++ox) {
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy) {
      const int k = (ox + 1) * 3 + (oy + 1);
      const f2 sample_uv = uv + F2((float)ox * fr.inv_rw, (float)oy * fr.inv_rh);  // ox, oy in {-1, 0, 1}: +-RN(1/size) or 0, exactly the quotient
      tap_inside[k] = !(sample_uv.x < 0.0f || sample_uv.y < 0.0f || sample_uv.x > 1.0f || sample_uv.y > 1.0f);
      int sx, sy;
      nearest_coords(sample_uv, fr.rw, fr.rh, &sx, &sy);
      const int at = tap_inside[k] ? sx + fr.rw * sy : index;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) tap_variance[k][ch] = d.variance[ch][at];
    }
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    f3 irradiance = xyz(unpack_f16x4(render_in[ch]));
    const f3 qd = irradiance / albedo;
    irradiance = F3(albedo.x < 0.01f ? 0.0f : qd.x, albedo.y < 0.01f ? 0.0f : qd.y, albedo.z < 0.01f ? 0.0f : qd.z);
    d.output[ch][index] = pack_f16x4(F4(irradiance, 1.0f));  // internal_texture_0
  }
  float sum_variance[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) sum_variance[ch] = 0.0f;
#pragma unroll
  for (int ox = -1; ox <= 1; ++ox) {
#pragma unroll
    for (int oy = -1; oy <= 1; ++oy) {
      const int k = (ox + 1) * 3 + (oy + 1);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const float variance = tap_variance[k][ch];
        const float with_tap = sum_variance[ch] + fr.kernel[(oy + 1) * 3 + (ox + 1)] * fmax_(variance, 0.0f);
        sum_variance[ch] = (tap_inside[k] && !(variance > HK_F32_MAX)) ? with_tap : sum_variance[ch];
      }
    }
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) d.internal_variance[ch][index] = sum_variance[ch];
}

// FFMASK bit ch = FIREFLY_FILTERING for channel ch (post_process.rs:773-783,1193-1197)
// is_nan(c) || c > F32_MAX for some component c - "not (c <= F32_MAX)": one compare per component
__device__ __forceinline__ bool nan_or_above_max(f3 v) { return !(v.x <= HK_F32_MAX) || !(v.y <= HK_F32_MAX) || !(v.z <= HK_F32_MAX); }

#ifndef HK_DENOISE_WAVES
#define HK_DENOISE_WAVES 8  // waves per SIMD the a-trous kernel is compiled for: 62-64 VGPRs without spills (unconstrained: 68-70 = 7 waves; frame -0.5 %, three interleaved A/B runs)
#endif
template <int LEVEL, int NCH, int FFMASK>
__global__ __launch_bounds__(256, HK_DENOISE_WAVES) void k_denoise(DFrame fr, DenoiseTargets d, int row_begin, int row_end) {  // denoise.wgsl:164-319
#if defined(HK_DENOISE_TILES_RR)
  const Pixel px = pixel_of_thread<false>(fr.rw, row_begin, row_end);
#else
  const Pixel px = pixel_of_thread_rows<HK_DENOISE_W, false>(fr.rw, row_begin, row_end);
#endif
  if (!px.valid) return;
  constexpr int STEP = 8 >> LEVEL;
  const int x = px.x, y = px.y, index = x + fr.rw * y;
  // The stencil's three columns and three rows: the uv of a tap, whether it lies in the image and the G-buffer texel under it are
  // separable in x and y, so the chain coords -> uv -> jittered uv -> nearest texel is evaluated once per column and once per row
  // (on the diagonal (x + a, y + a): column a's x and row a's y in one call) instead of once per tap; the same expressions.
  int col_texel[3], row_texel[3];
  bool col_inside[3], row_inside[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const f2 tap_uv = coords_to_uv(fr, x + (a - 1) * STEP, y + (a - 1) * STEP);
    col_inside[a] = !(tap_uv.x < 0.0f || tap_uv.x > 1.0f);
    row_inside[a] = !(tap_uv.y < 0.0f || tap_uv.y > 1.0f);
    nearest_coords(jittered_deferred_uv(fr, tap_uv, 0.5f), fr.dw, fr.dh, &col_texel[a], &row_texel[a]);
  }
  const int dx = col_texel[1], dy = row_texel[1];
  const int didx = dx + fr.dw * dy;
  const float4 gc = d.dn_g[didx];
  const float depth = d.depth[didx];
  if (depth < HK_F32_EPSILON) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) d.output[ch][index] = make_uint2(0u, 0u);
    if (LEVEL == 3 && d.tone_mapped) d.tone_mapped[index] = pack_f16x4(F4(d.clear_color[0], d.clear_color[1], d.clear_color[2], d.clear_color[3]));  // sum has w = 0
    return;
  }
  const float2 dg = d.depth_gradient[didx];
  const f2 depth_gradient = F2(dg.x, dg.y);
  const f3 normal = F3(gc.x, gc.y, gc.z);
  const float instance = gc.w;

  // The eight taps' channel-independent part first - where the tap lies, whether it is inside the image, its geometric weight
  // w_normal * w_depth * w_instance - then channel by channel (the reference runs the channels as separate dispatches: per channel
  // the taps are visited in its order, so every sum is formed in its order).
  constexpr int OX[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
  constexpr int OY[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
  bool tap_inside[8];
  float w_geometry[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ox = OX[k], oy = OY[k];
    tap_inside[k] = col_inside[ox + 1] && row_inside[oy + 1];  // sample_uv inside [0, 1]^2 (denoise.wgsl:232-234)
    w_geometry[k] = 0.0f;
    if (tap_inside[k]) {
      const int gx = col_texel[ox + 1], gy = row_texel[oy + 1];
      const float4 gs = d.dn_g[gx + fr.dw * gy];
      const float sample_depth = d.depth[gx + fr.dw * gy];
      const f3 sample_normal = F3(gs.x, gs.y, gs.z);
      const float w_normal = pow16_(fmax_(0.0f, dot(normal, sample_normal)));
      const float w_depth = exp_nonpositive_((-fabsf(depth - sample_depth)) / (fabsf(dot(depth_gradient, F2((float)ox, (float)oy))) + 0.01f));
      const float w_instance = fmax_(0.0f, 1.0f - fabsf(instance - gs.w));
      w_geometry[k] = w_normal * w_depth * w_instance;
    }
  }
  f4 albedo = F4(0, 0, 0, 0), tone_sum = F4(0, 0, 0, 0);
  if (LEVEL == 3) albedo = unpack_f16x4(d.albedo[didx]);  // nearest texel under jittered_deferred_uv(uv, 0.5): the centre's
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const float variance = d.internal_variance[ch][index];
    const uint2 centre = d.input[ch][index];
    f3 irradiance = xyz(unpack_f16x4(centre));
    f3 sum_irradiance = irradiance * fr.kernel[4];
    float sum_w = fr.kernel[4];
    if (nan_or_above_max(irradiance)) {  // any_is_nan(irradiance) || any(irradiance > F32_MAX), denoise.wgsl:201-205
      irradiance = F3(0, 0, 0);
      sum_irradiance = F3(0, 0, 0);
      sum_w = 0.0f;
    }
    const float lum = luminance(irradiance);
    const float lum_denominator = 4.0f * pow_quarter_(variance) + 0.001f;  // luminance_weight, denoise.wgsl:56-61
#ifndef HK_DN_F32_DIV
    const double inv_lum_denominator = 1.0 / (double)lum_denominator;  // the eight taps of a channel divide by the same number: see quotient_by_reciprocal
#endif
    uint2 tap[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) tap[k] = tap_inside[k] ? d.input[ch][index + OX[k] * STEP + fr.rw * (OY[k] * STEP)] : make_uint2(0u, 0u);
    float ff_moment_1 = 0.0f, ff_moment_2 = 0.0f, ff_count = 0.0f;
    // A channel that is BLACK under the whole stencil of every pixel of the wave (the sun's channel of a scene without a sun, of
    // pixels in its shadow; the emitters' far from any) needs none of the per-tap luminance arithmetic: with the centre and the tap
    // at +0 the luminance weight is exp(-|0 - 0| / denominator) = exp(-0) = 1 exactly (exp_nonpositive_: z = 0, r = +0, y = 1) for
    // every denominator that is not NaN, the tap adds (+0 * w) to the sum of irradiance and w = clamp(w_geometry) * kernel to the sum
    // of weights, and the firefly test (0 > ...) never fires.  Those additions are still made - a NaN weight must poison the sums as
    // it does on the long way - the 45 instructions per tap in front of them are not.  The rgb halves of the packed texel are tested
    // as bits: +0 only.  Wave-uniform by ballot.
    bool black = (centre.x | (centre.y & 0xFFFFu)) == 0u && lum_denominator == lum_denominator;
#pragma unroll
    for (int k = 0; k < 8; ++k) black = black && (tap[k].x | (tap[k].y & 0xFFFFu)) == 0u;
#if defined(HK_DN_F32_DIV) || defined(HK_DN_NO_BLACK)
    black = false;
#endif
    if (__ballot(!black) == 0ull) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!tap_inside[k]) continue;
        const float w = clamp_(w_geometry[k] * 1.0f, 0.0f, 1.0f) * fr.kernel[(OY[k] + 1) * 3 + (OX[k] + 1)];
        const float zw = 0.0f * w;
        sum_irradiance = sum_irradiance + F3(zw, zw, zw);
        sum_w += w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!tap_inside[k]) continue;
        const f3 irr = xyz(unpack_f16x4(tap[k]));
        if (nan_or_above_max(irr)) continue;  // any_is_nan(irr) || any(irr > F32_MAX)
        const float sample_luminance = luminance(irr);
#ifdef HK_DN_F32_DIV
        const float w_luminance = exp_nonpositive_((-fabsf(lum - sample_luminance)) / lum_denominator);
#else
        const float w_luminance = exp_nonpositive_(quotient_by_reciprocal(-fabsf(lum - sample_luminance), inv_lum_denominator));
#endif
        const float w = clamp_(w_geometry[k] * w_luminance, 0.0f, 1.0f) * fr.kernel[(OY[k] + 1) * 3 + (OX[k] + 1)];
        sum_irradiance = sum_irradiance + irr * w;
        sum_w += w;
        if ((FFMASK >> ch) & 1) {
          ff_moment_1 += sample_luminance;
          ff_moment_2 += sample_luminance * sample_luminance;
          ff_count += 1.0f;
        }
      }
    }
    const f3 qd = sum_irradiance / sum_w;
    irradiance = (sum_w < 0.0001f) ? F3(0, 0, 0) : qd;
    if ((FFMASK >> ch) & 1) {  // (a black channel: ff_count = 0, the mean is NaN and the comparison false - as 0 > mean + 3 sigma is on the long way)
      const float ff_mean = ff_moment_1 / ff_count;
      const float ff_var = ff_moment_2 / ff_count - ff_mean * ff_mean;
      if (lum > ff_mean + 3.0f * sqrtf(ff_var)) irradiance = ff_mean / lum * irradiance;
    }
    f4 color = F4(irradiance, 1.0f);
    if (LEVEL == 3) color = color * albedo;
    const uint2 packed = pack_f16x4(color);
    d.output[ch][index] = packed;
    if (LEVEL == 3) tone_sum = (ch == 0) ? unpack_f16x4(packed) : tone_sum + unpack_f16x4(packed);  // what tone_mapping loads back
  }
  if (LEVEL == 3 && d.tone_mapped) {  // tone_mapping.wgsl:21-32 (k_tone_mapping), channels summed in its order
    const f3 c = F3(fmax_(tone_sum.x, 0.0039f), fmax_(tone_sum.y, 0.0039f), fmax_(tone_sum.z, 0.0039f));
    const float l_old = dot(c, F3(0.2126f, 0.7152f, 0.0722f));
    const float l_new = l_old / (1.0f + l_old);
    f4 o = F4(c * (l_new / l_old), tone_sum.w);
    if (!(tone_sum.w > 0.0f)) o = F4(d.clear_color[0], d.clear_color[1], d.clear_color[2], d.clear_color[3]);
    d.tone_mapped[index] = pack_f16x4(o);
  }
}

}  // namespace hkd

namespace hk {
using namespace hkd;

void launch_derive_planes(hipStream_t st, const GBuffer& g, float* depth_plane, void* dn_g, int width, int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_derive_planes, grid_for(width, y1 - y0), dim3(256), 0, st, g, depth_plane, (float4*)dn_g, width, y0, y1);
}

void launch_count_geometry_rows(hipStream_t st, const float* depth, int width, int height, uint32_t* out) {
  if (height <= 0) return;
  hipLaunchKernelGGL(k_count_geometry_rows, dim3((unsigned)height), dim3(256), 0, st, depth, width, out);
}

void launch_demodulation(hipStream_t st, int nch, const DFrame& fr, const DemodTargets& d, int y0, int y1) {
  if (y1 <= y0) return;
  dim3 grid = grid_for_rows(64, fr.rw, y1 - y0);
  switch (nch) {
    case 1: hipLaunchKernelGGL(k_demodulation<1>, grid, dim3(256), 0, st, fr, d, y0, y1); break;
    case 2: hipLaunchKernelGGL(k_demodulation<2>, grid, dim3(256), 0, st, fr, d, y0, y1); break;
    default: hipLaunchKernelGGL(k_demodulation<3>, grid, dim3(256), 0, st, fr, d, y0, y1); break;
  }
}

template <int LEVEL>
static void launch_denoise_level(hipStream_t st, int nch, int ffmask, const DFrame& fr, const DenoiseTargets& d, int y0, int y1) {
#if defined(HK_DENOISE_TILES_RR)
  dim3 grid = grid_for(fr.rw, y1 - y0);
#else
  dim3 grid = grid_for_rows(HK_DENOISE_W, fr.rw, y1 - y0);
#endif
  if (nch == 1 && ffmask == 0) hipLaunchKernelGGL((k_denoise<LEVEL, 1, 0>), grid, dim3(256), 0, st, fr, d, y0, y1);
  else if (nch == 1) hipLaunchKernelGGL((k_denoise<LEVEL, 1, 1>), grid, dim3(256), 0, st, fr, d, y0, y1);
  else if (nch == 2) hipLaunchKernelGGL((k_denoise<LEVEL, 2, 2>), grid, dim3(256), 0, st, fr, d, y0, y1);  // sun, emissive
  else hipLaunchKernelGGL((k_denoise<LEVEL, 3, 6>), grid, dim3(256), 0, st, fr, d, y0, y1);                  // sun, emissive, indirect
}
// nch == 1: single channel with firefly filtering iff ffmask != 0; nch 2/3: the reference's channel order
void launch_denoise(hipStream_t st, int level, int nch, int ffmask, const DFrame& fr, const DenoiseTargets& d, int y0, int y1) {
  if (y1 <= y0) return;
  switch (level) {
    case 0: launch_denoise_level<0>(st, nch, ffmask, fr, d, y0, y1); break;
    case 1: launch_denoise_level<1>(st, nch, ffmask, fr, d, y0, y1); break;
    case 2: launch_denoise_level<2>(st, nch, ffmask, fr, d, y0, y1); break;
    default: launch_denoise_level<3>(st, nch, ffmask, fr, d, y0, y1); break;
  }
}

}  // namespace hk
