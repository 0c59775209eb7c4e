// hk_device.hpp - device-side data layout and the per-ray / per-sample routines of the path.
//
// Reference semantics implemented here (paths relative to cryscan/bevy-hikari v0.3.15):
//   reservoir codec + WRS            src/shaders/light.wgsl:35-223, 911-1004
//   AABB / triangle tests, traversal src/shaders/light.wgsl:344-486
//   hit attributes                   src/shaders/light.wgsl:488-533
//   samplers, light selection        src/shaders/light.wgsl:537-708, utils.wgsl:15-48
//   PBR shading                      src/shaders/light.wgsl:711-908 + bevy_pbr 0.9.1 pbr_lighting
//
// Device layout (what hk_upload_* converts the reference's std430 AoS buffers into):
//   BVH nodes  : two float4 planes  lo = (min.xyz, entry bits)  hi = (max.xyz, exit bits).
//                The reference stores EMPTY boxes in leaves and re-derives them per visit from
//                the three vertices (BLAS, light.wgsl:409-412), the instance record (TLAS,
//                light.wgsl:455-457) or the emitter sphere (light BVH, light.wgsl:634-636).  Here
//                the leaf box is filled in once at upload (bit-identical: min/max of the same
//                values), so a leaf visit costs the same two 16-B loads as an inner node and the
//                48-B triangle / 208-B instance record is only touched when its box is hit.
//   triangles  : three float4 planes (v0|i0, v1|i1, v2|i2) - one 16-B load per plane per lane.
//   vertices   : normal plane (float4) + uv plane (float2); positions live in the triangle planes.
//   instances  : 13 x 16 B record: inverse model (the transpose of inverse_transpose_model is
//                taken once at upload), model, normal matrix, mesh offsets.
#pragma once
#include "hk_device_math.hpp"

namespace hkd {

// ------------------------------------------------------------------ constants (light.wgsl:226-256)
#define HK_PI 3.141592653589793f
#define HK_TAU 6.283185307f
#define HK_INV_TAU 0.159154943f
#define HK_F32_EPSILON 1.1920929E-7f
#define HK_F32_MAX 3.402823466E+38f
#define HK_U32_MAX 0xFFFFFFFFu
#define HK_LEAF 0x80000000u
#define HK_RAY_BIAS 0.02f
#define HK_DISTANCE_MAX 65535.0f
#define HK_GOLDEN_RATIO 1.618033989f
#define HK_MAX_VARIANCE 10.0f
#define HK_DONT_EXCLUDE 0xFFFFFFFFu
#define HK_DONT_SAMPLE_EMISSIVE 0x80000000u

// ------------------------------------------------------------------ device layout
struct DInstance {
  float4 im0, im1, im2, im3;  // inverse model, columns
  float4 m0, m1, m2, m3;      // model, columns
  float4 n0, n1, n2;          // inverse_transpose_model columns (xyz) = normal matrix
  uint32_t material, vertex, primitive, node_offset;
  uint32_t node_count, moved /* previous model differs: see PrepassParams::prev_models */, pad1, pad2;
};
struct DEmissive {
  float4 position_radius;
  uint32_t instance, alias_offset, alias_count;
  float surface_area;
};
struct DScene {
  // TLAS nodes [0, tlas_count) followed by all BLAS nodes [blas_base, ...), 32 B each:
  // nodes[2i] = (min.xyz, entry bits), nodes[2i+1] = (max.xyz, exit bits) - one cache sector /
  // two adjacent ds_read_b128 per node step, one base register for both levels
  const float4* __restrict__ nodes;
  const DInstance* __restrict__ instances;
  const float4* __restrict__ tri_v0;
  const float4* __restrict__ tri_v1;
  const float4* __restrict__ tri_v2;
  const float4* __restrict__ vtx_normal;
  const float2* __restrict__ vtx_uv;
  const float4* __restrict__ materials;  // 4 x float4 per material: base_color, emissive, (perceptual_roughness, metallic, reflectance, 0),
                                         // texture ids as bits (base_color, emissive, metallic_roughness, occlusion)
  const float4* __restrict__ light_lo;
  const float4* __restrict__ light_hi;
  const DEmissive* __restrict__ emissives;
  const float2* __restrict__ alias;      // (prob, index bits)
  const uint32_t* __restrict__ noise;    // 16 x 64 x 64 RGBA8
  const uint32_t* __restrict__ tex_data; // all material textures, RGBA8 texels
  const uint4* __restrict__ tex_info;    // per texture: (texel offset, width, height, flags: bit0 sRGB, bit1 bilinear, bits 4-5 / 6-7 address u / v)
  const float* __restrict__ srgb_lut;    // 256-entry sRGB -> linear table
  uint32_t n_textures;
  uint32_t tlas_count, blas_base, light_count;
  // direction-threaded BVHs (hikari_hip.h HK_CTX_EXACT_TRAVERSAL): ordering `oct` of the TLAS starts at node oct * tlas_stride,
  // of BLAS node k at blas_base + oct * blas_stride + k.  Both strides are 0 when only the reference's order is stored.
  uint32_t tlas_stride, blas_stride;
  const float4* __restrict__ blob;       // all arrays above (except noise) live in [blob, blob + blob_f4)
  uint32_t blob_f4;
  // 1 when every instance has the same (bitwise) inverse model matrix - a single-asset scene under one root transform, like the
  // Cornell box: the world -> local ray is then the same for every instance a ray enters (traverse_top)
  uint32_t shared_xform;
  // One-level walk for scenes whose instances all share one transform (shared_xform) and that fit the LDS copy: ONE BVH over
  // every triangle of every instance, in the shared local space (traverse_flat).  `flat` = flat_count 32-B nodes per ordering
  // (lo = (min.xyz, entry), hi = (max.xyz, exit | instance << 16)), ordering k of flat_mask + 1 at node k * flat_count.
  // flat_mode = 1: traverse_top takes this walk (stage_scene makes it a compile-time constant per kernel instantiation).
  const float4* __restrict__ flat;
  uint32_t flat_count, flat_mask, flat_mode;
};
// per-frame constants, passed by value (lands in SGPRs / scalar cache)
struct DFrame {
  float kernel[9];  // kernel[c*3+r] = frame.kernel[c][r]
  uint32_t number;
  uint32_t direct_validate_interval, emissive_validate_interval, indirect_bounces, temporal_reuse;
  uint32_t max_temporal_reuse_count, max_spatial_reuse_count;
  float max_reservoir_lifetime, solar_angle, max_indirect_luminance, upscale_ratio;
  float random_float_number;  // random_float(frame.number), utils.wgsl:26-28
  float number_golden;        // f32(frame.number) * GOLDEN_RATIO
  float cam_x, cam_y, cam_z;  // view.world_position
  float ortho_x, ortho_y, ortho_z;  // (view_proj[0].z, view_proj[1].z, view_proj[2].z)
  uint32_t is_ortho;          // view.projection[3].w == 1.0
  float sun_r, sun_g, sun_b, sun_dx, sun_dy, sun_dz;
  float amb_r, amb_g, amb_b;
  float clear_r, clear_g, clear_b, clear_a;
  int dw, dh, rw, rh;         // deferred (full) size, scaled render size
  // 1/size as the host's IEEE division gives it, and whether (k + 0.5) / size may be taken through
  // div_by() for every pixel coordinate k in [-64, size + 64): hk_resize checks ALL of them against the
  // IEEE quotient (context.hip certify_uv_division), so the flag only selects a cheaper route to the same bits
  float inv_dw, inv_dh, inv_rw, inv_rh;
  uint32_t uv_fast;
  double rcp_rw, rcp_rh;      // 1.0 / (double)rw, 1.0 / (double)rh (IEEE f64): hk_device_math.hpp quotient_by_reciprocal
};
struct PackedReservoir {  // light.wgsl:35-43
  uint2 radiance;
  uint2 random;
  float4 visible_position;
  float4 sample_position;
  uint32_t visible_normal, sample_normal;
  uint2 reservoir;
};

struct Sample {
  f4 radiance;
  f4 random;
  f4 visible_position;
  f3 visible_normal;
  uint32_t visible_instance;
  f4 sample_position;
  f3 sample_normal;
};
struct Reservoir {
  Sample s;
  float count, lifetime, w, w_sum, w2_sum;
};
struct Ray { f3 origin, direction, inv_direction; };
struct Hit { f2 uv; float distance; uint32_t instance_index, primitive_index; };
struct Surface { f4 base_color, emissive; float reflectance, metallic, roughness, occlusion; };
struct HitInfo { f4 position; f3 normal; f2 uv; uint32_t instance_index, material_index; };
struct LightCandidate { f3 direction; float max_distance, min_distance; uint32_t emissive_instance; float p; };
// tlas / blas: walks started (SURVEY 8d's ray definition).  The other four price a walk in the terms of SURVEY 8d's algorithmic BVH
// bytes per ray - node steps x 32 B, triangle tests x 48 B, instance entries x 208 B, closest hits whose attributes are fetched
// (hit_info: three vertex records) x 96 B.  Only the COUNT instantiations of the kernels read them; everywhere else they are dead.
struct RayCounters { uint32_t tlas, blas, nodes = 0u, tris = 0u, entries = 0u, hits = 0u, top_nodes = 0u; };  // top_nodes: the node steps taken in the instance tree

HKD Sample zero_sample() {
  Sample s;
  s.radiance = F4(0, 0, 0, 0);
  s.random = F4(0, 0, 0, 0);
  s.visible_position = F4(0, 0, 0, 0);
  s.visible_normal = F3(0, 0, 0);
  s.visible_instance = 0u;
  s.sample_position = F4(0, 0, 0, 0);
  s.sample_normal = F3(0, 0, 0);
  return s;
}
HKD Reservoir zero_reservoir() {
  Reservoir r;
  r.s = zero_sample();
  r.count = 0.0f; r.lifetime = 0.0f; r.w = 0.0f; r.w_sum = 0.0f; r.w2_sum = 0.0f;
  return r;
}

// ------------------------------------------------------------------ utils.wgsl
HKD bool is_nan(float v) { return !(v < 0.0f || 0.0f < v || v == 0.0f); }
HKD bool any_is_nan(f3 v) { return is_nan(v.x) || is_nan(v.y) || is_nan(v.z); }
HKD f2 clip_to_uv(f4 clip) {
  f2 uv = F2(clip.x / clip.w, clip.y / clip.w);
  uv = (uv + 1.0f) * 0.5f;
  uv.y = 1.0f - uv.y;
  return uv;
}
// x / b through c = RN(1/b): q = x * c, then one correction with the exact residual.  Correctly rounded for the
// operands it is certified for (see DFrame::uv_fast); 3 VALU instead of the ~11 of the IEEE division sequence.
HKD float div_by(float x, float b, float c) {
  const float q = x * c;
  return fmaf(fmaf(-q, b, x), c, q);
}
HKD f2 coords_to_uv(int cx, int cy, int sx, int sy) { return F2(((float)cx + 0.5f) / (float)sx, ((float)cy + 0.5f) / (float)sy); }
HKD f2 coords_to_uv(const DFrame& fr, int cx, int cy) {  // utils.wgsl:36-38 on the scaled render size
  if (fr.uv_fast) return F2(div_by((float)cx + 0.5f, (float)fr.rw, fr.inv_rw), div_by((float)cy + 0.5f, (float)fr.rh, fr.inv_rh));
  return coords_to_uv(cx, cy, fr.rw, fr.rh);
}
HKD mat3 normal_basis(f3 n) {
  float s = fmin_(sign_(n.z) * 2.0f + 1.0f, 1.0f);
  float u = -1.0f / (s + n.z);
  float v = n.x * n.y * u;
  f3 t = F3(1.0f + s * n.x * n.x * u, s * v, -s * n.x);
  f3 b = F3(v, s + n.y * n.y * u, -n.y);
  return mat3{t, b, n};
}
HKD float luminance(f3 v) { return dot(v, F3(0.2126f, 0.7152f, 0.0722f)); }

// ------------------------------------------------------------------ bevy_pbr 0.9.1 pbr_lighting / utils
HKD float perceptualRoughnessToRoughness(float pr) {
  float c = clamp_(pr, 0.089f, 1.0f);
  return c * c;
}
HKD float D_GGX(float roughness, float NoH) {
  float oneMinusNoHSquared = 1.0f - NoH * NoH;
  float a = NoH * roughness;
  float k = roughness / (oneMinusNoHSquared + a * a);
  return k * k * (1.0f / HK_PI);
}
HKD float V_SmithGGXCorrelated(float roughness, float NoV, float NoL) {
  float a2 = roughness * roughness;
  float lambdaV = NoL * sqrtf((NoV - a2 * NoV) * NoV + a2);
  float lambdaL = NoV * sqrtf((NoL - a2 * NoL) * NoL + a2);
  return 0.5f / (lambdaV + lambdaL);
}
HKD float F_Schlick(float f0, float f90, float VoH) { return f0 + (f90 - f0) * pow5_(1.0f - VoH); }
HKD f3 fresnel(f3 f0, float LoH) {
  float f90 = saturate(dot(f0, F3s(50.0f * 0.33f)));
  float p = pow5_(1.0f - LoH);
  return f0 + (F3s(f90) - f0) * p;
}
HKD f3 specular(f3 f0, float roughness, float NoV, float NoL, float NoH, float LoH, float specularIntensity) {
  float D = D_GGX(roughness, NoH);
  float V = V_SmithGGXCorrelated(roughness, NoV, NoL);
  f3 F = fresnel(f0, LoH);
  return (specularIntensity * D * V) * F;
}
HKD float Fd_Burley(float roughness, float NoV, float NoL, float LoH) {
  float f90 = 0.5f + 2.0f * roughness * LoH * LoH;
  float lightScatter = F_Schlick(1.0f, f90, NoL);
  float viewScatter = F_Schlick(1.0f, f90, NoV);
  return lightScatter * viewScatter * (1.0f / HK_PI);
}
// exp2(-9.28 * NoV) is shared by the two EnvBRDFApprox calls of ambient()/env_brdf()
HKD f3 EnvBRDFApprox_e(f3 f0, float perceptual_roughness, float e) {
  const f4 c0 = F4(-1.0f, -0.0275f, -0.572f, 0.022f);
  const f4 c1 = F4(1.0f, 0.0425f, 1.04f, -0.04f);
  f4 r = perceptual_roughness * c0 + c1;
  float a004 = fmin_(r.x * r.x, e) * r.x + r.y;
  f2 AB = F2(-1.04f, 1.04f) * a004 + F2(r.z, r.w);
  return f0 * AB.x + AB.y;
}

// ------------------------------------------------------------------ reservoir codec + WRS
HKD Reservoir unpack_reservoir(const PackedReservoir& p) {  // light.wgsl:77-109
  Reservoir r;
  f2 t0 = unpack2x16float(p.reservoir.x), t1 = unpack2x16float(p.reservoir.y);
  r.count = t0.x; r.w = t0.y; r.w_sum = t1.x; r.w2_sum = t1.y;
  t0 = unpack2x16float(p.radiance.x); t1 = unpack2x16float(p.radiance.y);
  r.s.radiance = F4(t0.x, t0.y, t1.x, t1.y);
  t0 = unpack2x16unorm(p.random.x); t1 = unpack2x16unorm(p.random.y);
  r.s.random = F4(t0.x, t0.y, t1.x, t1.y);
  f4 t2 = unpack4x8snorm(p.visible_normal);
  r.s.visible_position = F4(p.visible_position);
  r.s.visible_normal = normalize(xyz(t2));
  r.lifetime = 127.0f * (1.0f + t2.w);
  t2 = unpack4x8snorm(p.sample_normal);
  r.s.sample_position = F4(p.sample_position.x, p.sample_position.y, p.sample_position.z, t2.w);
  r.s.sample_normal = normalize(xyz(t2));
  r.s.visible_instance = f32_to_u32(p.sample_position.w);
  return r;
}
HKD PackedReservoir pack_reservoir(const Reservoir& r) {  // light.wgsl:111-136
  PackedReservoir p;
  p.reservoir = make_uint2(pack2x16float(r.count, r.w), pack2x16float(r.w_sum, r.w2_sum));
  p.radiance = make_uint2(pack2x16float(r.s.radiance.x, r.s.radiance.y), pack2x16float(r.s.radiance.z, r.s.radiance.w));
  p.random = make_uint2(pack2x16unorm(r.s.random.x, r.s.random.y), pack2x16unorm(r.s.random.z, r.s.random.w));
  p.visible_position = to_float4(r.s.visible_position);
  p.sample_position = make_float4(r.s.sample_position.x, r.s.sample_position.y, r.s.sample_position.z, (float)r.s.visible_instance);
  p.visible_normal = pack4x8snorm(F4(r.s.visible_normal, r.lifetime / 127.0f - 1.0f));
  p.sample_normal = pack4x8snorm(F4(r.s.sample_normal, r.s.sample_position.w));
  return p;
}
// 64-B record moved as four 16-B accesses
HKD PackedReservoir load_packed(const PackedReservoir* __restrict__ buf, int index) {
  const uint4* q = reinterpret_cast<const uint4*>(buf + index);
  uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  PackedReservoir p;
  p.radiance = make_uint2(a.x, a.y);
  p.random = make_uint2(a.z, a.w);
  p.visible_position = make_float4(u2f(b.x), u2f(b.y), u2f(b.z), u2f(b.w));
  p.sample_position = make_float4(u2f(c.x), u2f(c.y), u2f(c.z), u2f(c.w));
  p.visible_normal = d.x;
  p.sample_normal = d.y;
  p.reservoir = make_uint2(d.z, d.w);
  return p;
}
HKD void store_packed(PackedReservoir* __restrict__ buf, int index, const PackedReservoir& p) {
  uint4* q = reinterpret_cast<uint4*>(buf + index);
  q[0] = make_uint4(p.radiance.x, p.radiance.y, p.random.x, p.random.y);
  q[1] = make_uint4(f2u(p.visible_position.x), f2u(p.visible_position.y), f2u(p.visible_position.z), f2u(p.visible_position.w));
  q[2] = make_uint4(f2u(p.sample_position.x), f2u(p.sample_position.y), f2u(p.sample_position.z), f2u(p.sample_position.w));
  q[3] = make_uint4(p.visible_normal, p.sample_normal, p.reservoir.x, p.reservoir.y);
}
HKD void set_reservoir(Reservoir& r, const Sample& s, float w_new) {  // light.wgsl:138-144
  r.count = 1.0f;
  r.lifetime = 0.0f;
  r.w_sum = w_new;
  r.w2_sum = w_new * w_new;
  r.s = s;
}
HKD void update_reservoir(Reservoir& r, const Sample& s, float w_new) {  // light.wgsl:146-173
  r.w_sum += w_new;
  r.w2_sum += w_new * w_new;
  r.count = r.count + 1.0f;
  float rand = fract(dot(s.random, F4(1.0f, 1.0f, 1.0f, 1.0f)));
  if (rand < w_new / r.w_sum) r.s = s;
}
HKD void merge_reservoir(Reservoir& r, const Reservoir& other, float p) {  // light.wgsl:175-179
  float count = r.count;
  update_reservoir(r, other.s, p * other.w * other.count);
  r.count = count + other.count;
}
HKD Reservoir load_reservoir_uv(const PackedReservoir* __restrict__ buf, f2 uv, int sx, int sy) {  // light.wgsl:181-190,201-210
  Reservoir r = zero_reservoir();
  if (fabsf(uv.x - 0.5f) < 0.5f && fabsf(uv.y - 0.5f) < 0.5f) {
    int cx = f32_to_i32(uv.x * (float)sx), cy = f32_to_i32(uv.y * (float)sy);
    r = unpack_reservoir(load_packed(buf, cx + sx * cy));
  }
  return r;
}
HKD bool check_previous_reservoir(Reservoir& r, const Sample& s) {  // light.wgsl:917-935
  float depth_ratio = r.s.visible_position.w / s.visible_position.w;
  depth_ratio = (depth_ratio < 1.0f) ? 1.0f / depth_ratio : depth_ratio;
  bool depth_miss = depth_ratio > 1.05f * (1.0f + 0.5f * s.random.x);
  bool instance_miss = r.s.visible_instance != s.visible_instance;
  bool normal_miss = dot(s.visible_normal, r.s.visible_normal) < 0.9f;
  if (depth_miss || normal_miss || instance_miss) {
    r = zero_reservoir();
    return false;
  }
  return true;
}
HKD void temporal_restir(Reservoir& r, const Sample& s, float w_new, uint32_t max_sample_count) {  // light.wgsl:937-952
  update_reservoir(r, s, w_new);
  float m = (float)max_sample_count;
  if (r.count > m) {
    r.w_sum *= m / r.count;
    r.w2_sum *= m / r.count;
    r.count = m;
  }
}
// compute_jacobian for a caller that already holds d = normalize(q.sample_position - r.visible_position) and the length of that
// vector (k_spatial_reuse forms both for its facing test).  The reference normalises the OPPOSITE vector, r.visible_position -
// q.sample_position: IEEE subtraction is antisymmetric, the squares under the root are the same numbers, and a dot product of
// negated operands is the negated dot product (round-to-nearest is symmetric) - so |dot| and the squared length come out bit for
// bit (a zero component may carry the other sign; fabsf and the squares remove it).  One sqrt, one division, two dots fewer.
HKD float compute_jacobian_shared(const Sample& q, f3 d, float d_length) {
  f3 normal = q.sample_normal;
  float cos_phi_1 = fabsf(dot(d, normal));
  float cos_phi_2 = fabsf(dot(normalize(xyz(q.visible_position) - xyz(q.sample_position)), normal));
  float term_1 = cos_phi_1 / fmax_(0.0001f, cos_phi_2);
  float num = length(xyz(q.visible_position) - xyz(q.sample_position));
  num *= num;
  float denom = d_length;
  denom *= denom;
  float term_2 = num / fmax_(denom, 0.0001f);
  return clamp_(term_1 * term_2, 1.0f, 50.0f);
}
HKD float compute_jacobian(const Sample& q, const Sample& r) {  // light.wgsl:985-1004
  f3 normal = q.sample_normal;
  float cos_phi_1 = fabsf(dot(normalize(xyz(r.visible_position) - xyz(q.sample_position)), normal));
  float cos_phi_2 = fabsf(dot(normalize(xyz(q.visible_position) - xyz(q.sample_position)), normal));
  float term_1 = cos_phi_1 / fmax_(0.0001f, cos_phi_2);
  float num = length(xyz(q.visible_position) - xyz(q.sample_position));
  num *= num;
  float denom = length(xyz(r.visible_position) - xyz(q.sample_position));
  denom *= denom;
  float term_2 = num / fmax_(denom, 0.0001f);
  return clamp_(term_1 * term_2, 1.0f, 50.0f);
}
HKD float reservoir_variance(const Reservoir& r) {  // light.wgsl:1224-1226,1488-1490,1672-1674
  float variance = r.w2_sum / r.count - pow2_(r.w_sum / r.count);
  variance = (r.count < 1.0f) ? variance : variance / r.count;
  return fmin_(variance, HK_MAX_VARIANCE);
}

// ------------------------------------------------------------------ ray / box / triangle
HKD float intersects_aabb(const Ray& ray, f3 bmin, f3 bmax) {  // light.wgsl:344-362
  f3 t1 = (bmin - ray.origin) * ray.inv_direction;
  f3 t2 = (bmax - ray.origin) * ray.inv_direction;
  float t_min = fmin_(t1.x, t2.x);
  float t_max = fmax_(t1.x, t2.x);
  t_min = fmax_(t_min, fmin_(t1.y, t2.y));
  t_max = fmin_(t_max, fmax_(t1.y, t2.y));
  t_min = fmax_(t_min, fmin_(t1.z, t2.z));
  t_max = fmin_(t_max, fmax_(t1.z, t2.z));
  float t = HK_F32_MAX;
  if (t_max >= t_min && t_max >= 0.0f) t = t_min;
  return t;
}
// light.wgsl:364-398; returns distance (F32_MAX on miss) and writes uv exactly as the reference does
HKD float intersects_triangle(const Ray& ray, f3 p0, f3 p1, f3 p2, f2* uv_out) {
  *uv_out = F2(0.0f, 0.0f);
  f3 ab = p1 - p0;
  f3 ac = p2 - p0;
  f3 u_vec = cross(ray.direction, ac);
  float det = dot(ab, u_vec);
  if (fabsf(det) < HK_F32_EPSILON) return HK_F32_MAX;
  float inv_det = 1.0f / det;
  f3 ao = ray.origin - p0;
  float u = dot(ao, u_vec) * inv_det;
  if (u < 0.0f || u > 1.0f) {
    *uv_out = F2(u, 0.0f);
    return HK_F32_MAX;
  }
  f3 v_vec = cross(ao, ab);
  float v = dot(ray.direction, v_vec) * inv_det;
  *uv_out = F2(u, v);
  if (v < 0.0f || u + v > 1.0f) return HK_F32_MAX;
  float distance = dot(ac, v_vec) * inv_det;
  return (distance > HK_F32_EPSILON) ? distance : HK_F32_MAX;
}

HKD f3 world_to_local_position(const DInstance& in, f3 p) {  // light.wgsl:306-310
  f4 q = mul(in.im0, in.im1, in.im2, in.im3, F4(p, 1.0f));
  return xyz(q) / q.w;
}
HKD f3 world_to_local_direction(const DInstance& in, f3 d) {  // light.wgsl:312-316
  return xyz(mul(in.im0, in.im1, in.im2, in.im3, F4(d, 0.0f)));
}
HKD f3 local_to_world_position(const DInstance& in, f3 p) {  // light.wgsl:318-322
  f4 q = mul(in.m0, in.m1, in.m2, in.m3, F4(p, 1.0f));
  return xyz(q) / q.w;
}
HKD f3 local_to_world_normal(const DInstance& in, f3 n) {  // light.wgsl:324-338
  mat3 m = {xyz(in.n0), xyz(in.n1), xyz(in.n2)};
  return normalize(mul(m, n));
}

// sign pattern of a ray direction: selects the flattening whose child order is the one this ray meets (strides 0: always 0)
HKD uint32_t ray_octant(f3 d) { return (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u); }

// stackless skip-link walk of one BLAS, light.wgsl:400-440.  Its one stand-alone caller is the closest hit on the sampled EMITTER's own
// mesh (select_light_candidate, light.wgsl:687): round 5 walks it in the reference's order (ordering 0) whatever the ray's octant -
// two triangles of the emitter that tie exactly (a sampled point on a shared edge) then resolve as in the reference, and with the
// wide walk's rank rule (hk_wide.hpp) and traverse_top<true> every hit of the product default is the reference's.
// HK_EMITTER_WALK_REF_ORDER=0 is the A/B (the octant's ordering: the visits front to back).
#ifndef HK_EMITTER_WALK_REF_ORDER
#define HK_EMITTER_WALK_REF_ORDER 1
#endif
HKD bool traverse_bottom(const DScene& sc, Hit& hit, const Ray& ray, uint32_t node_offset, uint32_t node_count, uint32_t primitive_offset,
                         float early_distance, RayCounters& rc) {
  bool intersected = false;
  uint32_t index = 0u;
  const uint32_t nbase = sc.blas_base + (HK_EMITTER_WALK_REF_ORDER ? 0u : ray_octant(ray.direction) * sc.blas_stride) + node_offset;
  while (index < node_count) {
    const float4* __restrict__ nd = sc.nodes + 2u * (nbase + index);
    const float4 lo = nd[0];
    const float4 hi = nd[1];
    rc.nodes++;
    const uint32_t entry = f2u(lo.w), exit_ = f2u(hi.w);
    const bool box_hit = intersects_aabb(ray, xyz(lo), xyz(hi)) < hit.distance;
    if (entry >= HK_LEAF) {
      if (box_hit) {
        const uint32_t primitive_index = primitive_offset + entry - HK_LEAF;
        rc.tris++;
        f2 uv;
        float d = intersects_triangle(ray, xyz(sc.tri_v0[primitive_index]), xyz(sc.tri_v1[primitive_index]), xyz(sc.tri_v2[primitive_index]), &uv);
        if (d < hit.distance) {
          hit.uv = uv;
          hit.distance = d;
          hit.primitive_index = primitive_index;
          intersected = true;
          if (d < early_distance) return intersected;
        }
      }
      index = exit_;
    } else {
      index = box_hit ? entry : exit_;
    }
  }
  return intersected;
}
// One-level walk (DScene::flat).  When every instance has the same transform - the Cornell box, any single asset under one
// root - all triangles live in ONE local space and every instance entry of the reference's two-level walk forms the SAME
// local ray.  A single BVH over all triangles, walked with that ray, therefore runs the reference's per-triangle arithmetic on
// the reference's operands: distance, barycentrics, primitive and instance of the closest hit are the reference's bit for bit.
// What can differ is only WHICH candidates are visited: the reference tests a triangle only if the world-space ray also
// passed the slab test of its instance's world AABB (a ray grazing that box within rounding), and two candidates at exactly
// the same distance are taken in visit order - both measure-zero events (tests: a handful of pixels per million), which is
// why HK_CTX_EXACT_TRAVERSAL keeps the two-level walk for the bit-exact suite and this is the product default under the
// north star's 1e-3 gate.
// What it buys: one event kind instead of three.  In the two-level walk a 64-lane wave pays the node step, the triangle test
// and the instance entry / BLAS exit bookkeeping in nearly every iteration because SOME lane needs each (lane utilisation
// 0.36).  Here a leaf whose box is hit only QUEUES its triangle (up to HK_FLAT_CAP per lane) and the lane walks on; the
// triangle-test block runs when some lane's queue is full or has nothing else to do, for every lane that has one queued - so
// it runs in a third of the iterations with three times the lanes.
#ifdef HK_PROFILE_SECTIONS
extern __device__ unsigned long long g_walk_events[8];  // wave-iterations: all, with any triangle test, any instance entry, any BLAS exit; lane-iterations
// exactly one lane - the lowest active one - counts each wave iteration, so early exits of other lanes lose nothing
#define HK_WALK_LEADER() ((unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(__builtin_amdgcn_ballot_w64(true) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)__builtin_amdgcn_ballot_w64(true), 0u)) == 0u)
#define HK_WALK_EVENT(k, cond) do { const bool any_ = __builtin_amdgcn_ballot_w64(cond) != 0ull; if (any_ && HK_WALK_LEADER()) wev_[k] += 1u; } while (0)
#else
#define HK_WALK_EVENT(k, cond) ((void)0)
#endif
#ifndef HK_FLAT_CAP
#define HK_FLAT_CAP 2
#endif
template <bool REF_ORDER = false>
HKD Hit traverse_flat(const DScene& sc, const Ray& ray, float max_distance, float early_distance, uint32_t exclude_instance, RayCounters& rc) {
  rc.tlas++;
  Hit hit;
  hit.uv = F2(0.0f, 0.0f);
  hit.distance = max_distance;
  hit.instance_index = HK_U32_MAX;
  hit.primitive_index = HK_U32_MAX;
  const DInstance& in0 = sc.instances[0];
  Ray lr;  // light.wgsl:459-461, the same for every instance
  lr.origin = world_to_local_position(in0, ray.origin);
  lr.direction = world_to_local_direction(in0, ray.direction);
  lr.inv_direction = 1.0f / lr.direction;
  const float4* __restrict__ nodes = sc.flat + 2u * (REF_ORDER ? 0u : (ray_octant(lr.direction) & sc.flat_mask) * sc.flat_count);
  const uint32_t count = sc.flat_count;
  uint32_t index = 0u, npend = 0u;
  uint32_t pend[HK_FLAT_CAP];  // queued candidates, oldest first: primitive index | instance << 16
#pragma unroll
  for (int k = 0; k < HK_FLAT_CAP; ++k) pend[k] = 0u;
#ifdef HK_PROFILE_SECTIONS
  // [0] wave iterations, [1] of which ran the triangle-test block, [4] lanes walking a node (summed), [5] lanes testing a triangle
  // (summed), [6] rays, [7] lanes inside the loop (summed)
  uint32_t wev_[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  struct WalkFlush {
    uint32_t* w;
    __device__ ~WalkFlush() {
      for (int k = 0; k < 8; ++k)
        if (w[k]) atomicAdd(&g_walk_events[k], (unsigned long long)w[k]);
    }
  } wflush_{wev_};
  {
    const uint32_t lanes_ = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(true));
    if (HK_WALK_LEADER()) wev_[6] += lanes_;
  }
#endif
  for (;;) {
#ifdef HK_PROFILE_SECTIONS
    {
      const uint32_t in_ = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(true));
      const uint32_t walking_ = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(index < count));
      if (HK_WALK_LEADER()) { wev_[0] += 1u; wev_[4] += walking_; wev_[7] += in_; }
    }
#endif
    if (index < count) {
      const float4 lo = nodes[2u * index];
      const float4 hi = nodes[2u * index + 1u];
      rc.nodes++;
      const uint32_t entry = f2u(lo.w), link = f2u(hi.w);
      const f3 t1 = (xyz(lo) - lr.origin) * lr.inv_direction;  // intersects_aabb, light.wgsl:344-362
      const f3 t2 = (xyz(hi) - lr.origin) * lr.inv_direction;
      float t_min = fmin_(t1.x, t2.x);
      float t_max = fmax_(t1.x, t2.x);
      t_min = fmax_(t_min, fmin_(t1.y, t2.y));
      t_max = fmin_(t_max, fmax_(t1.y, t2.y));
      t_min = fmax_(t_min, fmin_(t1.z, t2.z));
      t_max = fmin_(t_max, fmax_(t1.z, t2.z));
      const float t_box = (t_max >= t_min && t_max >= 0.0f) ? t_min : HK_F32_MAX;
      const bool box_hit = t_box < hit.distance;
      const bool leaf = entry >= HK_LEAF;
      index = (leaf || !box_hit) ? (link & 0xFFFFu) : index + 1u;  // depth-first layout: an inner node's first child follows it
      if (leaf && box_hit && (link >> 16) != exclude_instance) {
        const uint32_t cand = (entry & 0xFFFFu) | (link & 0xFFFF0000u);
#pragma unroll
        for (int k = 0; k < HK_FLAT_CAP; ++k)
          if (npend == (uint32_t)k) pend[k] = cand;
        npend += 1u;
      }
    }
    const bool want = npend == (uint32_t)HK_FLAT_CAP || (npend > 0u && index >= count);
    if (__builtin_amdgcn_ballot_w64(want) != 0ull) {  // wave-uniform: everybody with a queued candidate tests its oldest
#ifdef HK_PROFILE_SECTIONS
      {
        const uint32_t testing_ = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(npend > 0u));
        if (HK_WALK_LEADER()) { wev_[1] += 1u; wev_[5] += testing_; }
      }
#endif
      if (npend > 0u) {
        const uint32_t cand = pend[0];
#pragma unroll
        for (int k = 0; k + 1 < HK_FLAT_CAP; ++k) pend[k] = pend[k + 1];
        npend -= 1u;
        const uint32_t primitive_index = cand & 0xFFFFu;
        rc.tris++;
        f2 uv;
        const float d = intersects_triangle(lr, xyz(sc.tri_v0[primitive_index]), xyz(sc.tri_v1[primitive_index]), xyz(sc.tri_v2[primitive_index]), &uv);
        if (d < hit.distance) {
          hit.uv = uv;
          hit.distance = d;
          hit.primitive_index = primitive_index;
          hit.instance_index = cand >> 16;
          if (d < early_distance) return hit;  // light.wgsl:421-423, 466-469
        }
      }
    }
    if (index >= count && npend == 0u) break;
  }
  return hit;
}

// Two-level stackless walk, light.wgsl:442-486 (TLAS) with light.wgsl:400-440 (BLAS) inlined as ONE
// loop: every iteration each live lane takes exactly one node step - of the TLAS or of the BLAS it
// is currently inside - so lanes that sit in different instances (or still in the TLAS) execute the
// same load + slab-test stream instead of serialising nested loops under partial exec masks.  Only
// the two rare events diverge: entering an instance (ray transform) and a triangle test.  Each
// lane's own visit order, and therefore every result bit, is that of the reference's nested loops.
// REF_ORDER (round 5): the walk takes ordering 0 - the REFERENCE's own child order (scene_layout.hip thread_orderings) - on both
// levels, whatever the ray's octant.  For the any-hit rays whose OCCLUDER is kept: direct_lit stores the occluder's position as the
// reservoir's sample_position (light.wgsl:526-533, 1117-1130), and a later validation frame shoots its ray at that position
// (light.wgsl:1156-1170).  Which occluder an any-hit walk reports - the first it meets - depends on the order of the visits; in any
// other order than the reference's the validation rays, and through them the reservoirs, drift away from the reference's
// (profiles/r05_default_mode_sequence_*: the emissive channel 1.6e-2 off after 30 frames).  An unoccluded ray visits the same boxes
// in any order, so only rays that do find an occluder pay for not meeting it front to back.
template <bool REF_ORDER = false>
HKD Hit traverse_top(const DScene& sc, const Ray& ray, float max_distance, float early_distance, uint32_t exclude_instance, RayCounters& rc) {
  // (a ray whose occluder is kept does not take the one-level walk either: that is ANOTHER tree - whichever of its orderings is
  // walked, the first occluder it meets need not be the reference's; the two-level walk below in ordering 0 is the reference's walk.
  // HK_FLAT_KEPT_OCCLUDERS = 1 is the A/B: round 3-4's behaviour, the one-level walk for these rays too.)
#ifndef HK_FLAT_KEPT_OCCLUDERS
#define HK_FLAT_KEPT_OCCLUDERS 0
#endif
  if (sc.flat_mode && (!REF_ORDER || HK_FLAT_KEPT_OCCLUDERS)) return traverse_flat<REF_ORDER>(sc, ray, max_distance, early_distance, exclude_instance, rc);
  rc.tlas++;
#ifdef HK_PROFILE_SECTIONS
  uint32_t wev_[5] = {0u, 0u, 0u, 0u, 0u};
  struct WalkFlush {
    uint32_t* w;
    __device__ ~WalkFlush() {
      for (int k = 0; k < 5; ++k)
        if (w[k]) atomicAdd(&g_walk_events[k], (unsigned long long)w[k]);
    }
  } wflush_{wev_};
#endif
  Hit hit;
  hit.uv = F2(0.0f, 0.0f);
  hit.distance = max_distance;
  hit.instance_index = HK_U32_MAX;
  hit.primitive_index = HK_U32_MAX;
  // cursor of the level being walked: node = nodes[base + index], index < limit
  const uint32_t tlas_base = REF_ORDER ? 0u : ray_octant(ray.direction) * sc.tlas_stride;
  uint32_t index = 0u, limit = sc.tlas_count, base = tlas_base;
  uint32_t t_resume = 0u;  // TLAS index to continue with when the current BLAS is exhausted
  uint32_t prim_base = 0u, cur_instance = 0u;
  bool in_blas = false, intersected = false;
  f3 co = ray.origin, cinv = ray.inv_direction;  // origin / inverse direction of the level being walked
  f3 ld = ray.direction;                          // local direction while inside a BLAS
  // One inverse model for all instances: the local ray is formed ONCE, with every lane of the wave, instead of at each instance
  // entry with the one or two lanes that happen to enter in that iteration (an entry is ~100 instructions and some lane needs
  // one in a fifth of the iterations).  Same operands, same operations: same bits.
  f3 hco = ray.origin, hld = ray.direction, hcinv = ray.inv_direction;
  if (sc.shared_xform) {
    const DInstance& in0 = sc.instances[0];
    hco = world_to_local_position(in0, ray.origin);
    hld = world_to_local_direction(in0, ray.direction);
    hcinv = 1.0f / hld;
  }
  for (;;) {
    HK_WALK_EVENT(0, true);
    HK_WALK_EVENT(3, index >= limit && in_blas);
#ifdef HK_PROFILE_SECTIONS
    {
      const uint32_t lanes_ = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(true));
      if (HK_WALK_LEADER()) wev_[4] += lanes_;
    }
#endif
    if (index >= limit) {
      if (!in_blas) break;
      // traverse_bottom returned, light.wgsl:465-470
      if (intersected) {
        hit.instance_index = cur_instance;
        if (hit.distance < early_distance) return hit;
      }
      in_blas = false;
      index = t_resume;
      limit = sc.tlas_count;
      base = tlas_base;
      co = ray.origin;
      cinv = ray.inv_direction;
      continue;
    }
    const float4* __restrict__ nd = sc.nodes + 2u * (base + index);
    const float4 lo = nd[0];
    const float4 hi = nd[1];
    rc.nodes++;
    rc.top_nodes += in_blas ? 0u : 1u;
    const uint32_t entry = f2u(lo.w), exit_ = f2u(hi.w);
    // intersects_aabb, light.wgsl:344-362, on the current level's ray
    const f3 t1 = (xyz(lo) - co) * cinv;
    const f3 t2 = (xyz(hi) - co) * cinv;
    float t_min = fmin_(t1.x, t2.x);
    float t_max = fmax_(t1.x, t2.x);
    t_min = fmax_(t_min, fmin_(t1.y, t2.y));
    t_max = fmin_(t_max, fmax_(t1.y, t2.y));
    t_min = fmax_(t_min, fmin_(t1.z, t2.z));
    t_max = fmin_(t_max, fmax_(t1.z, t2.z));
    const float t_box = (t_max >= t_min && t_max >= 0.0f) ? t_min : HK_F32_MAX;
    const bool box_hit = t_box < hit.distance;
    const bool leaf = entry >= HK_LEAF;
    // inner node: descend on a hit, skip the subtree otherwise; leaf: always continue at its exit
    index = (leaf || !box_hit) ? exit_ : entry;
    HK_WALK_EVENT(1, leaf && box_hit && in_blas);
    HK_WALK_EVENT(2, leaf && box_hit && !in_blas);
    if (leaf && box_hit) {  // the two rare events
      if (in_blas) {
        const uint32_t primitive_index = prim_base + entry - HK_LEAF;
        rc.tris++;
        Ray lr;
        lr.origin = co;
        lr.direction = ld;
        lr.inv_direction = cinv;
        f2 uv;
        const float d = intersects_triangle(lr, xyz(sc.tri_v0[primitive_index]), xyz(sc.tri_v1[primitive_index]), xyz(sc.tri_v2[primitive_index]), &uv);
        if (d < hit.distance) {
          hit.uv = uv;
          hit.distance = d;
          hit.primitive_index = primitive_index;
          intersected = true;
          if (d < early_distance) {  // light.wgsl:421-423 then 466-469
            hit.instance_index = cur_instance;
            return hit;
          }
        }
      } else {
        const uint32_t instance_index = entry - HK_LEAF;
        if (instance_index != exclude_instance) {
          const DInstance& in = sc.instances[instance_index];
          rc.entries++;
          if (sc.shared_xform) {
            co = hco;
            ld = hld;
            cinv = hcinv;
          } else {
            co = world_to_local_position(in, ray.origin);
            ld = world_to_local_direction(in, ray.direction);
            cinv = 1.0f / ld;
          }
          t_resume = index;
          base = sc.blas_base + (REF_ORDER ? 0u : ray_octant(ld) * sc.blas_stride) + in.node_offset;
          index = 0u;
          limit = in.node_count;
          prim_base = in.primitive;
          cur_instance = instance_index;
          in_blas = true;
          intersected = false;
        }
      }
    }
  }
  return hit;
}

HKD HitInfo empty_hit_info(f3 position, f3 direction) {  // light.wgsl:488-494
  HitInfo info;
  info.instance_index = HK_U32_MAX;
  info.material_index = HK_U32_MAX;
  info.position = F4(position + direction * HK_DISTANCE_MAX, 0.0f);
  info.normal = F3(0, 0, 0);
  info.uv = F2(0, 0);
  return info;
}
HKD HitInfo hit_info(const DScene& sc, const Ray& ray, const Hit& hit) {  // light.wgsl:496-523
  HitInfo info;
  info.instance_index = hit.instance_index;
  info.material_index = HK_U32_MAX;
  info.normal = F3(0, 0, 0);
  info.uv = F2(0, 0);
  if (hit.instance_index != HK_U32_MAX) {
    const DInstance& in = sc.instances[hit.instance_index];
    const uint32_t i0 = in.vertex + f2u(sc.tri_v0[hit.primitive_index].w);
    const uint32_t i1 = in.vertex + f2u(sc.tri_v1[hit.primitive_index].w);
    const uint32_t i2 = in.vertex + f2u(sc.tri_v2[hit.primitive_index].w);
    const float2 t0 = sc.vtx_uv[i0], t1 = sc.vtx_uv[i1], t2 = sc.vtx_uv[i2];
    const f3 n0 = xyz(sc.vtx_normal[i0]), n1 = xyz(sc.vtx_normal[i1]), n2 = xyz(sc.vtx_normal[i2]);
    f2 uv0 = F2(t0.x, t0.y), uv1 = F2(t1.x, t1.y), uv2 = F2(t2.x, t2.y);
    f2 uv = hit.uv;
    info.uv = uv0 + uv.x * (uv1 - uv0) + uv.y * (uv2 - uv0);
    info.normal = n0 + uv.x * (n1 - n0) + uv.y * (n2 - n0);
    info.normal = local_to_world_normal(in, info.normal);
    info.position = F4(ray.origin + ray.direction * hit.distance, 1.0f);
    info.material_index = in.material;
  } else {
    info.position = F4(ray.origin + ray.direction * HK_DISTANCE_MAX, 0.0f);
  }
  return info;
}
HKD void occlude_hit_info(const Ray& ray, const Hit& hit, HitInfo& info) {  // light.wgsl:526-533
  if (hit.instance_index != HK_U32_MAX) {
    info.instance_index = hit.instance_index;
    info.material_index = HK_U32_MAX;
    info.position = F4(ray.origin + ray.direction * hit.distance, 1.0f);
    info.normal = F3(0, 0, 0);
  }
}

// ------------------------------------------------------------------ sampling
HKD f4 sample_cosine_hemisphere(f2 rand) {  // light.wgsl:537-549
  float r = sqrtf(rand.x);
  float theta = 2.0f * HK_PI * rand.y;
  float sn, cs;
  sincos_(theta, &sn, &cs);
  f2 t = F2(r * cs, r * sn);
  f3 direction = F3(t.x, t.y, sqrtf(1.0f - dot(t, t)));
  float pdf = 2.0f * HK_INV_TAU * direction.z;
  return F4(direction, pdf);
}
HKD f3 sample_uniform_cone(f2 rand, float cos_angle) {  // light.wgsl:552-559 (direction only; the pdf is never read)
  float z = 1.0f - (1.0f - cos_angle) * rand.x;
  float theta = HK_TAU * rand.y;
  float r = sqrtf(1.0f - z * z);
  float sn, cs;
  sincos_(theta, &sn, &cs);
  return F3(r * cs, r * sn, z);
}
HKD f3 compute_emissive_radiance(f4 emissive) { return 255.0f * emissive.w * xyz(emissive); }  // light.wgsl:594-596

HKD bool inside_aabb(f3 p, f3 mn, f3 mx) {  // light.wgsl:340-342
  return p.x > mn.x && p.y > mn.y && p.z > mn.z && p.x < mx.x && p.y < mx.y && p.z < mx.z;
}

HKD LightCandidate select_light_candidate(const DScene& sc, const DFrame& fr, f4 rand, f3 position, f3 normal, uint32_t instance, HitInfo& info,
                                          RayCounters& rc) {  // light.wgsl:599-708
  LightCandidate candidate;
  candidate.max_distance = HK_F32_MAX;
  candidate.min_distance = HK_DISTANCE_MAX;
  candidate.emissive_instance = HK_DONT_SAMPLE_EMISSIVE;

  const f3 cone_dir = F3(fr.sun_dx, fr.sun_dy, fr.sun_dz);
  const float cone_cos = cos_(fr.solar_angle);
  f3 rand_direction = mul(normal_basis(cone_dir), sample_uniform_cone(F2(rand.z, rand.w), cone_cos));
  candidate.direction = rand_direction;
  candidate.p = 1.0f;
  info = empty_hit_info(position, rand_direction);
  if (instance == HK_DONT_SAMPLE_EMISSIVE) return candidate;

  // light-BVH point query with streaming uniform pick
  uint32_t chosen = HK_U32_MAX;
  float count = 0.0f;
  uint32_t index = 0u;
  float rand_1d = rand.x;
  while (index < sc.light_count) {
    const float4 lo = sc.light_lo[index];
    const float4 hi = sc.light_hi[index];
    const uint32_t entry = f2u(lo.w), exit_ = f2u(hi.w);
    const bool inside = inside_aabb(position, xyz(lo), xyz(hi));
    if (entry >= HK_LEAF) {
      const uint32_t emissive_index = entry - HK_LEAF;
      const uint32_t em_instance = sc.emissives[emissive_index].instance;
      if (instance != em_instance && inside) {
        rand_1d = fract(rand_1d + HK_GOLDEN_RATIO);
        count += 1.0f;
        if (rand_1d < 1.0f / count) {
          candidate.emissive_instance = em_instance;
          chosen = emissive_index;
        }
      }
      index = exit_;
    } else {
      index = inside ? entry : exit_;
    }
  }

  if (candidate.emissive_instance != HK_DONT_SAMPLE_EMISSIVE) {
    const DEmissive em = sc.emissives[chosen];
    const uint32_t alias_index = min(f32_to_u32(rand.x * (float)em.alias_count), em.alias_count - 1u);
    const float2 alias_entry = sc.alias[em.alias_offset + alias_index];
    const uint32_t primitive_local = (rand.y < alias_entry.x) ? f2u(alias_entry.y) : alias_index;

    const DInstance& ein = sc.instances[candidate.emissive_instance];
    const uint32_t prim = ein.primitive + primitive_local;
    const f3 v0 = xyz(sc.tri_v0[prim]), v1 = xyz(sc.tri_v1[prim]), v2 = xyz(sc.tri_v2[prim]);
    const float srx = sqrtf(rand.z);  // sample_uniform_triangle_barycentric, light.wgsl:562-565
    const f2 b = F2(1.0f - srx, rand.w * srx);
    const f3 p = local_to_world_position(ein, b.x * v0 + b.y * v1 + (1.0f - b.x - b.y) * v2);

    Hit hit;
    hit.uv = F2(0, 0);
    hit.distance = HK_F32_MAX;
    hit.instance_index = HK_U32_MAX;
    hit.primitive_index = HK_U32_MAX;

    Ray ray;
    ray.origin = position + normal * HK_RAY_BIAS;
    ray.direction = normalize(p - position);
    ray.inv_direction = F3(0, 0, 0);

    Ray r;
    r.origin = world_to_local_position(ein, ray.origin);
    r.direction = world_to_local_direction(ein, ray.direction);
    r.inv_direction = 1.0f / r.direction;

    candidate.direction = ray.direction;
    const bool front = dot(candidate.direction, normal) > 0.0f;
    if (front) rc.blas++;
    if (front && traverse_bottom(sc, hit, r, ein.node_offset, ein.node_count, ein.primitive, 0.0f, rc)) {
      hit.instance_index = em.instance;
      rc.hits++;
      info = hit_info(sc, ray, hit);
      candidate.max_distance = hit.distance;
      candidate.min_distance = hit.distance - 0.1f;
      f3 delta = xyz(info.position) - position;
      candidate.p = dot(delta, delta) / (fabsf(dot(ray.direction, info.normal) * em.surface_area));
      candidate.p = candidate.p / count;
    } else {
      info = empty_hit_info(ray.origin, ray.direction);
      candidate.emissive_instance = HK_DONT_SAMPLE_EMISSIVE;
      candidate.direction = rand_direction;
      candidate.p = 1.0f;
    }
  }
  return candidate;
}

// ------------------------------------------------------------------ shading
HKD f3 calculate_view(const DFrame& fr, f3 world_position) {  // light.wgsl:714-727
  if (fr.is_ortho) return normalize(F3(fr.ortho_x, fr.ortho_y, fr.ortho_z));
  return normalize(F3(fr.cam_x, fr.cam_y, fr.cam_z) - world_position);
}
// textureSampleLevel(textures[id], samplers[id], uv, 0.0), light.wgsl:756-789: texel centres at
// (i + 0.5) / size, f32 bilinear weights, sRGB rgb decoded to linear per texel before filtering.
HKD int wrap_coord(int i, int n, uint32_t mode) {
  if (mode == 1u) { int m = i % n; return m < 0 ? m + n : m; }                                            // repeat
  if (mode == 2u) { int p = 2 * n; int m = i % p; if (m < 0) m += p; return m < n ? m : p - 1 - m; }      // mirror repeat
  return min(max(i, 0), n - 1);                                                                            // clamp to edge
}
HKD f4 texel(const DScene& sc, uint4 ti, int x, int y) {
  const uint32_t t = sc.tex_data[ti.x + (uint32_t)y * ti.y + (uint32_t)x];
  const uint32_t r = t & 0xffu, g = (t >> 8) & 0xffu, b = (t >> 16) & 0xffu, a = t >> 24;
  if (ti.w & 1u) return F4(sc.srgb_lut[r], sc.srgb_lut[g], sc.srgb_lut[b], unorm8(a));
  return F4(unorm8(r), unorm8(g), unorm8(b), unorm8(a));
}
HKD f4 mix4(f4 a, f4 b, float t) { return F4(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t), mix(a.w, b.w, t)); }
HKD f4 sample_texture(const DScene& sc, uint32_t id, f2 uv) {
  const uint4 ti = sc.tex_info[id];
  const int w = (int)ti.y, h = (int)ti.z;
  const uint32_t au = (ti.w >> 4) & 3u, av = (ti.w >> 6) & 3u;
  if (!(ti.w & 2u)) {
    const int x = wrap_coord(f32_to_i32(floorf(uv.x * (float)w)), w, au), y = wrap_coord(f32_to_i32(floorf(uv.y * (float)h)), h, av);
    return texel(sc, ti, x, y);
  }
  const float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
  const float x0 = floorf(x), y0 = floorf(y);
  const float fx = x - x0, fy = y - y0;
  const int ix = f32_to_i32(x0), iy = f32_to_i32(y0);
  const int xa = wrap_coord(ix, w, au), xb = wrap_coord(ix + 1, w, au), ya = wrap_coord(iy, h, av), yb = wrap_coord(iy + 1, h, av);
  const f4 top = mix4(texel(sc, ti, xa, ya), texel(sc, ti, xb, ya), fx);
  const f4 bot = mix4(texel(sc, ti, xa, yb), texel(sc, ti, xb, yb), fx);
  return mix4(top, bot, fy);
}
HKD Surface retreive_surface(const DScene& sc, uint32_t material_index, f2 uv) {  // light.wgsl:730-742 (NO_TEXTURE) / 749-781
  Surface s;
  const float4 bc = sc.materials[4 * material_index], em = sc.materials[4 * material_index + 1], pr = sc.materials[4 * material_index + 2];
  s.base_color = F4(bc);
  s.emissive = F4(em);
  s.metallic = pr.y;
  s.occlusion = 1.0f;
  if (sc.n_textures) {  // scene-uniform: the NO_TEXTURE pipelines pay nothing
    const float4 ids = sc.materials[4 * material_index + 3];
    uint32_t id = f2u(ids.x);
    if (id != HK_U32_MAX) s.base_color = s.base_color * sample_texture(sc, id, uv);
    id = f2u(ids.y);
    if (id != HK_U32_MAX) s.emissive = s.emissive * sample_texture(sc, id, uv);
    id = f2u(ids.z);
    if (id != HK_U32_MAX) s.metallic *= sample_texture(sc, id, uv).x;
    id = f2u(ids.w);
    if (id != HK_U32_MAX) s.occlusion = sample_texture(sc, id, uv).x;
  }
  s.roughness = perceptualRoughnessToRoughness(pr.x);
  s.reflectance = pr.z;
  return s;
}
HKD f4 retreive_emissive(const DScene& sc, uint32_t material_index, f2 uv) {  // light.wgsl:744-747 / 783-793
  f4 emissive = F4(sc.materials[4 * material_index + 1]);
  if (sc.n_textures) {
    const uint32_t id = f2u(sc.materials[4 * material_index + 3].y);
    if (id != HK_U32_MAX) emissive = emissive * sample_texture(sc, id, uv);
  }
  return emissive;
}
HKD f3 lit(f3 radiance, f3 diffuse_color, float roughness, f3 F0, f3 L, f3 N, f3 V) {  // light.wgsl:796-818
  f3 Hh = normalize(L + V);
  float NoL = saturate(dot(N, L));
  float NoH = saturate(dot(N, Hh));
  float LoH = saturate(dot(L, Hh));
  float NdotV = fmax_(dot(N, V), 0.0001f);
  f3 diffuse = diffuse_color * Fd_Burley(roughness, NdotV, NoL, LoH);
  f3 specular_light = specular(F0, roughness, NdotV, NoL, NoH, LoH, 1.0f);
  return (specular_light + diffuse) * radiance * NoL;
}
HKD f3 env_brdf_terms(f3 diffuse_color, float roughness, float occlusion, f3 F0, f3 N, f3 V) {  // light.wgsl:828-832 / 901-907
  float NdotV = fmax_(dot(N, V), 0.0001f);
  float e = exp2_(-9.28f * NdotV);
  f3 diffuse_ambient = EnvBRDFApprox_e(diffuse_color, 1.0f, e);
  f3 specular_ambient = EnvBRDFApprox_e(F0, roughness, e);
  return occlusion * (diffuse_ambient + specular_ambient);
}
HKD f4 input_radiance(const DScene& sc, const DFrame& fr, const Ray& ray, const HitInfo& info, bool sample_directional, uint32_t sample_emissive,
                      bool sample_ambient) {  // light.wgsl:835-867
  f3 radiance = F3(0, 0, 0);
  float ambient_ = 0.0f;
  if (info.instance_index == HK_U32_MAX) {
    bool hit_directional = dot(ray.direction, F3(fr.sun_dx, fr.sun_dy, fr.sun_dz)) >= cos_(fr.solar_angle);
    if (sample_directional && hit_directional) {
      radiance = F3(fr.sun_r, fr.sun_g, fr.sun_b);
      ambient_ = 0.0f;
    } else {
      radiance = sample_ambient ? F3(fr.amb_r, fr.amb_g, fr.amb_b) : F3(0, 0, 0);
      ambient_ = 1.0f;
    }
  } else if (sample_emissive == info.instance_index) {
    radiance = compute_emissive_radiance(retreive_emissive(sc, info.material_index, info.uv));
  }
  return F4(radiance, 1.0f - ambient_);
}
// shading() split into the part that depends only on (V, N, surface) and the part that depends on
// the light direction, so kernels that shade one site against many directions (spatial_reuse: up to
// 18 per pixel) evaluate the view/ambient terms once.  Same operations in the same order as
// lit() + ambient() + the mix of light.wgsl:869-888 - only evaluated once instead of per call.
struct ShadingSite {
  f3 V, N, F0, diffuse_color, ambient_radiance;
  float roughness, NdotV, view_pow5, fresnel_f90;
};
HKD ShadingSite make_site(const DFrame& fr, f3 V, f3 N, const Surface& surface) {
  ShadingSite st;
  f3 base_color = xyz(surface.base_color);
  float reflectance = surface.reflectance, metallic = surface.metallic, occlusion = surface.occlusion;
  st.V = V;
  st.N = N;
  st.roughness = surface.roughness;
  st.F0 = F3s(0.16f * reflectance * reflectance * (1.0f - metallic)) + base_color * metallic;
  st.diffuse_color = base_color * (1.0f - metallic);
  st.NdotV = fmax_(dot(N, V), 0.0001f);
  st.view_pow5 = pow5_(1.0f - st.NdotV);                    // F_Schlick(1, f90, NoV) of Fd_Burley
  st.fresnel_f90 = saturate(dot(st.F0, F3s(50.0f * 0.33f)));     // fresnel()
  st.ambient_radiance = env_brdf_terms(st.diffuse_color, st.roughness, occlusion, st.F0, N, V) * F3(fr.amb_r, fr.amb_g, fr.amb_b);
  return st;
}
HKD f3 shade(const ShadingSite& st, f3 L, f4 in_radiance) {
  f3 Hh = normalize(L + st.V);
  float NoL = saturate(dot(st.N, L));
  float NoH = saturate(dot(st.N, Hh));
  float LoH = saturate(dot(L, Hh));
  // Fd_Burley
  float f90 = 0.5f + 2.0f * st.roughness * LoH * LoH;
  float lightScatter = 1.0f + (f90 - 1.0f) * pow5_(1.0f - NoL);
  float viewScatter = 1.0f + (f90 - 1.0f) * st.view_pow5;
  f3 diffuse = st.diffuse_color * (lightScatter * viewScatter * (1.0f / HK_PI));
  // specular
  float D = D_GGX(st.roughness, NoH);
  float Vs = V_SmithGGXCorrelated(st.roughness, st.NdotV, NoL);
  float p = pow5_(1.0f - LoH);
  f3 F = st.F0 + (F3s(st.fresnel_f90) - st.F0) * p;
  f3 specular_light = (1.0f * D * Vs) * F;
  f3 lit_radiance = (specular_light + diffuse) * xyz(in_radiance) * NoL;
  return mix(lit_radiance, st.ambient_radiance, 1.0f - in_radiance.w);
}
HKD f3 shading(const DFrame& fr, f3 V, f3 N, f3 L, const Surface& surface, f4 in_radiance) {  // light.wgsl:869-888
  return shade(make_site(fr, V, N, surface), L, in_radiance);
}
HKD f3 env_brdf(f3 V, f3 N, const Surface& surface) {  // light.wgsl:890-908
  f3 base_color = xyz(surface.base_color);
  float reflectance = surface.reflectance, roughness = surface.roughness, metallic = surface.metallic, occlusion = surface.occlusion;
  f3 F0 = F3s(0.16f * reflectance * reflectance * (1.0f - metallic)) + base_color * metallic;
  f3 diffuse_color = base_color * (1.0f - metallic);
  return env_brdf_terms(diffuse_color, roughness, occlusion, F0, N, V);
}

// ------------------------------------------------------------------ deferred addressing
HKD f2 jittered_deferred_uv(const DFrame& fr, f2 uv, float amount) {  // light.wgsl:1007-1011 (0.25) / denoise.wgsl:37-41 (0.5)
  f2 texel_size = F2(fr.inv_dw, fr.inv_dh);
  float ratio = fr.upscale_ratio - 1.0f;
  float sgn = ((fr.number & 1u) == 0u) ? -amount : amount;
  return uv + sgn * texel_size * ratio;
}
HKD void jittered_deferred_coords(const DFrame& fr, f2 uv, int* cx, int* cy) {  // light.wgsl:1013-1017
  f2 duv = jittered_deferred_uv(fr, uv, 0.25f);
  *cx = f32_to_i32(duv.x * (float)fr.dw);
  *cy = f32_to_i32(duv.y * (float)fr.dh);
}
HKD bool in_bounds(int x, int y, int w, int h) { return x >= 0 && y >= 0 && x < w && y < h; }
HKD void nearest_coords(f2 uv, int w, int h, int* x, int* y) {  // nearest sampler, clamp-to-edge
  int cx = (int)floorf(uv.x * (float)w), cy = (int)floorf(uv.y * (float)h);
  *x = min(max(cx, 0), w - 1);
  *y = min(max(cy, 0), h - 1);
}
HKD f4 noise_fetch(const DScene& sc, int x, int y, uint32_t n) {  // light.wgsl:1075-1078
  uint32_t noise_id = n % 16u;
  float fx = ((float)x + (float)n + 0.5f) / 64.0f, fy = ((float)y + (float)n + 0.5f) / 64.0f;
  int tx = (int)floorf(fract(fx) * 64.0f) & 63, ty = (int)floorf(fract(fy) * 64.0f) & 63;
  uint32_t t = sc.noise[(noise_id * 64u + (uint32_t)ty) * 64u + (uint32_t)tx];
  return F4(unorm8(t), unorm8(t >> 8), unorm8(t >> 16), unorm8(t >> 24));
}

}  // namespace hkd
