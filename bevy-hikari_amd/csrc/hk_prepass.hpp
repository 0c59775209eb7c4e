// hk_prepass.hpp - the prepass by primary rays (kernels.hip k_prepass): the parameters, the primary ray of a pixel (prepass.wgsl:40-100
// semantics by ray casting), the near-plane clip of the rasteriser it replaces, and the G-buffer record of a pixel from its closest hit.
#pragma once
#include <hip/hip_runtime.h>

#include "hk_device.hpp"
#include "hk_kernels.hpp"

namespace hkd {

struct PrepassParams {
  float4 ivp0, ivp1, ivp2, ivp3;  // inverse_view_proj columns
  float4 vp0, vp1, vp2, vp3;      // view_proj columns
  float4 pvp0, pvp1, pvp2, pvp3;  // previous view_proj columns
  float jitter_x, jitter_y;       // NDC shift of the geometry (prepass.wgsl:52-54,71)
  const float4* prev_models;      // previous model matrix (4 columns) per instance, read where DInstance::moved
  WideTrees wide;                 // scenes in global memory, product default: the records of the wide walk (tlas == nullptr: the skip-link walk)
};
// the point of the NEAR PLANE under the pixel's centre (reverse-Z: NDC z = 1), minus the geometry's jitter shift
__device__ __forceinline__ f3 primary_near_point(const DFrame& fr, const PrepassParams& pp, float px, float py) {
  const float ux = fr.uv_fast ? div_by(px + 0.5f, (float)fr.dw, fr.inv_dw) : (px + 0.5f) / (float)fr.dw;
  const float uy = fr.uv_fast ? div_by(py + 0.5f, (float)fr.dh, fr.inv_dh) : (py + 0.5f) / (float)fr.dh;
  float ndc_x = ux * 2.0f - 1.0f - pp.jitter_x;
  float ndc_y = 1.0f - uy * 2.0f - pp.jitter_y;
  f4 pn = mul(pp.ivp0, pp.ivp1, pp.ivp2, pp.ivp3, F4(ndc_x, ndc_y, 1.0f, 1.0f));
  return xyz(pn) / pn.w;
}
// Near-plane clip (round 6; prepass.rs:242-266: the reference's raster pipeline has `unclipped_depth: false`, so geometry in front
// of the camera's near plane - 0.1 by default - never reaches the G-buffer).  A perspective primary ray starts at the EYE (every
// stored bit of a frame without such geometry depends on that origin); when its closest hit lies in front of the near plane the
// pixel is traced once more from the near plane itself: what the rasteriser would have drawn there - the nearest surface BEYOND the
// plane, a triangle that straddles it cut exactly at it.  Returns whether `ray` was moved (the caller traces again).  Orthographic
// rays start on the near plane already.
__device__ __forceinline__ bool clip_at_near_plane(const DFrame& fr, const PrepassParams& pp, float px, float py, Ray& ray, const Hit& hit) {
  if (fr.is_ortho || hit.instance_index == HK_U32_MAX) return false;
  const f3 near_point = primary_near_point(fr, pp, px, py);
  if (!(hit.distance < length(near_point - ray.origin))) return false;
  ray.origin = near_point;
  return true;
}
__device__ __forceinline__ Ray primary_ray(const DFrame& fr, const PrepassParams& pp, float px, float py) {
  const f3 near_point = primary_near_point(fr, pp, px, py);
  Ray ray;
  if (fr.is_ortho) {
    ray.origin = near_point;
    ray.direction = -normalize(F3(fr.ortho_x, fr.ortho_y, fr.ortho_z));
  } else {
    ray.origin = F3(fr.cam_x, fr.cam_y, fr.cam_z);
    ray.direction = normalize(near_point - ray.origin);
  }
  ray.inv_direction = 1.0f / ray.direction;
  return ray;
}


// the pixel's G-buffer record (and albedo, when the dispatch also fills it) from the closest hit of its primary ray
__device__ __forceinline__ void prepass_store(const DScene& sc, const DFrame& fr, const PrepassParams& pp, const GBuffer& g, int x, int y, const Ray& ray, const Hit& hit) {
  const int idx = x + fr.dw * y;
  if (hit.instance_index == HK_U32_MAX) {
    g.position[idx] = make_float4(0, 0, 0, 0);
    g.normal[idx] = 0u;
    g.depth_gradient[idx] = make_float2(0, 0);
    g.instance_material[idx] = make_float2(0, 0);
    g.velocity_uv[idx] = make_float4(0, 0, 0, 0);
    g.depth[idx] = 0.0f;
    g.dn_g[idx] = denoise_geometry(0u, 0.0f);
    if (g.albedo_out) g.albedo_out[idx] = make_uint2(0u, 0u);
  } else {
    const DInstance& in = sc.instances[hit.instance_index];
    const float4 q0 = sc.tri_v0[hit.primitive_index], q1 = sc.tri_v1[hit.primitive_index], q2 = sc.tri_v2[hit.primitive_index];
    const uint32_t i0 = in.vertex + f2u(q0.w), i1 = in.vertex + f2u(q1.w), i2 = in.vertex + f2u(q2.w);
    const f2 b = hit.uv;
    const f3 world_position = ray.origin + ray.direction * hit.distance;
    const f4 clip = mul(pp.vp0, pp.vp1, pp.vp2, pp.vp3, F4(world_position, 1.0f));
    const float depth = clip.z / clip.w;
    const f3 n0 = local_to_world_normal(in, xyz(sc.vtx_normal[i0]));
    const f3 n1 = local_to_world_normal(in, xyz(sc.vtx_normal[i1]));
    const f3 n2 = local_to_world_normal(in, xyz(sc.vtx_normal[i2]));
    const f3 wn = n0 + b.x * (n1 - n0) + b.y * (n2 - n0);
    const float2 t0 = sc.vtx_uv[i0], t1 = sc.vtx_uv[i1], t2 = sc.vtx_uv[i2];
    const f2 uv = F2(t0.x, t0.y) + b.x * (F2(t1.x, t1.y) - F2(t0.x, t0.y)) + b.y * (F2(t2.x, t2.y) - F2(t0.x, t0.y));
    const f3 p0 = local_to_world_position(in, xyz(q0));
    const f3 p1 = local_to_world_position(in, xyz(q1));
    const f3 p2 = local_to_world_position(in, xyz(q2));
    const f3 ng = cross(p1 - p0, p2 - p0);
    float grad[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      Ray rn = primary_ray(fr, pp, (float)x + (k == 0 ? 1.0f : 0.0f), (float)y + (k == 1 ? 1.0f : 0.0f));
      float tn = dot(ng, p0 - rn.origin) / dot(ng, rn.direction);
      f3 wp = rn.origin + rn.direction * tn;
      f4 cn = mul(pp.vp0, pp.vp1, pp.vp2, pp.vp3, F4(wp, 1.0f));
      grad[k] = cn.z / cn.w - depth;
    }
    // prepass.wgsl:50,96: previous_world_position = previous_mesh.model * vertex, interpolated over the triangle
    f4 previous_world = F4(world_position, 1.0f);
    if (in.moved) {
      const float4* pm = pp.prev_models + 4u * hit.instance_index;
      const f3 local = xyz(q0) + b.x * (xyz(q1) - xyz(q0)) + b.y * (xyz(q2) - xyz(q0));
      previous_world = mul(pm[0], pm[1], pm[2], pm[3], F4(local, 1.0f));
    }
    const f2 velocity = clip_to_uv(clip) - clip_to_uv(mul(pp.pvp0, pp.pvp1, pp.pvp2, pp.pvp3, previous_world));
    g.position[idx] = make_float4(world_position.x, world_position.y, world_position.z, depth);
    const uint32_t packed_normal = pack4x8snorm(F4(wn, 1.0f));
    g.normal[idx] = packed_normal;
    g.depth_gradient[idx] = make_float2(grad[0], grad[1]);
    g.instance_material[idx] = make_float2((float)hit.instance_index + 0.5f, (float)in.material + 0.5f);
    g.velocity_uv[idx] = make_float4(velocity.x, velocity.y, uv.x, uv.y);
    g.depth[idx] = depth;
    g.dn_g[idx] = denoise_geometry(packed_normal, (float)hit.instance_index + 0.5f);
    if (g.albedo_out) {  // full_screen_albedo on the values just stored (light.wgsl:1019-1042)
      uint2 a = make_uint2(0u, 0u);
      if (!(depth < HK_F32_EPSILON)) {
        Surface surface = retreive_surface(sc, f32_to_u32((float)in.material + 0.5f), uv);
        a = pack_f16x4(F4(env_brdf(calculate_view(fr, world_position), xyz(unpack4x8snorm(packed_normal)), surface), 1.0f));
      }
      g.albedo_out[idx] = a;
    }
  }
}

// host side: the parameters from the view's matrices (column-major float[16])
inline PrepassParams make_prepass_params(const float* inverse_view_proj, const float* view_proj, const float* prev_view_proj, const float4* prev_models, float jitter_x,
                                         float jitter_y, const WideTrees* wide) {
  PrepassParams pp;
  pp.prev_models = prev_models;
  pp.wide = wide ? *wide : WideTrees{};
  auto col = [](const float* m, int c) { return make_float4(m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]); };
  pp.ivp0 = col(inverse_view_proj, 0); pp.ivp1 = col(inverse_view_proj, 1); pp.ivp2 = col(inverse_view_proj, 2); pp.ivp3 = col(inverse_view_proj, 3);
  pp.vp0 = col(view_proj, 0); pp.vp1 = col(view_proj, 1); pp.vp2 = col(view_proj, 2); pp.vp3 = col(view_proj, 3);
  pp.pvp0 = col(prev_view_proj, 0); pp.pvp1 = col(prev_view_proj, 1); pp.pvp2 = col(prev_view_proj, 2); pp.pvp3 = col(prev_view_proj, 3);
  pp.jitter_x = jitter_x;
  pp.jitter_y = jitter_y;
  return pp;
}

}  // namespace hkd
