// hk_oracle.cpp - CPU restatement of bevy-hikari v0.3.15's light + denoise compute path.
//
// TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load the library built from this file.  It is never linked into, imported by or shipped
// with the product (bevy-hikari_amd/).
//
// PARITY: PINNED TO THE REFERENCE'S SHADER SOURCE, NOT TO A RUN OF THE REFERENCE.  The reference has no tests, golden
// vectors or fixtures for this path and cannot be built or run in this environment (no Rust toolchain, wgpu or Vulkan
// ICD).  The oracle is a line-by-line restatement of the reference WGSL; every function cites the reference file:line it
// follows (paths relative to /root/reference).  tests/tools/wgsl_pin.py executes that WGSL itself (tests/tools/wgsl: a WGSL -> Python
// translation, one f32 rounding per operation) dispatch by dispatch next to this file and finds no differing byte
// (tests/test_wgsl_pin.py); the FSR1 passes likewise from the GLSL in src/shaders/fsr/source.zip.  NOT
// covered by that pin and still "unpinned" in the strict sense: the bevy_pbr functions below and the ray-cast G-buffer.  Third-party arithmetic the
// path imports but the checkout does not vendor (bevy_pbr 0.9.1 `lighting`/`utils` WGSL modules,
// bevy_core_pipeline 0.9.1 `tonemapping`) is restated from the published bevy 0.9.1 sources in
// the section "bevy_pbr 0.9.1" below and kept in one place so it can be corrected.
//
// Implementation-defined behaviour (transcendental accuracy, FMA contraction, NaN handling of
// min/max, out-of-bounds texture loads) is fixed by the numeric contract in hk_oracle_math.h.
//
// The C API (orc_*) deliberately mirrors include/hikari_hip.h one to one so the same Python
// driver can run the product and the oracle side by side on the same inputs.
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

#include "../include/hikari_hip.h"
#include "hk_oracle_math.h"

namespace orc {

// ------------------------------------------------------------------------------------------
// constants, light.wgsl:226-256 / denoise.wgsl:30-35
// ------------------------------------------------------------------------------------------
static const float PI = 3.141592653589793f;  // bevy_pbr::utils PI
static const float TAU = 6.283185307f;
static const float INV_TAU = 0.159154943f;
static const float F32_EPSILON = 1.1920929E-7f;
static const float F32_MAX = 3.402823466E+38f;
static const uint32_t U32_MAX = 0xFFFFFFFFu;
static const uint32_t BVH_LEAF_FLAG = 0x80000000u;
static const float RAY_BIAS = 0.02f;
static const float DISTANCE_MAX = 65535.0f;
static const uint32_t NOISE_TEXTURE_COUNT = 16u;
static const float GOLDEN_RATIO = 1.618033989f;
static const float MAX_VARIANCE = 10.0f;
static const uint32_t DONT_EXCLUDE = 0xFFFFFFFFu;
static const uint32_t DONT_SAMPLE_EMISSIVE = 0x80000000u;
static const uint32_t SPATIAL_REUSE_TAPS = 4u;
static const uint32_t DIRECT_VALIDATION_FRAME_SAMPLE_THRESHOLD = 4u;
static const uint32_t SPATIAL_VARIANCE_SAMPLE_THRESHOLD = 4u;

// ------------------------------------------------------------------------------------------
// context: the resources the reference binds in groups 0..6
// ------------------------------------------------------------------------------------------
struct PackedReservoir {  // light.wgsl:35-43, 64 B
  uint32_t radiance[2];
  uint32_t random[2];
  float visible_position[4];
  float sample_position[4];
  uint32_t visible_normal;
  uint32_t sample_normal;
  uint32_t reservoir[2];
};
static_assert(sizeof(PackedReservoir) == 64, "PackedReservoir must be 64 bytes");
static_assert(sizeof(HkInstance) == 176 && sizeof(HkMaterial) == 80 && sizeof(HkEmissive) == 64 &&
                  sizeof(HkFrame) == 256 && sizeof(HkView) == 416 && sizeof(HkNode) == 32 && sizeof(HkPrimitive) == 48,
              "std430 layouts");

struct Stats {
  std::atomic<uint64_t> rays_primary{0}, rays_tlas{0}, rays_blas{0}, frames{0};
};

struct Ctx {
  std::vector<HkVertex> vertices;
  std::vector<HkPrimitive> primitives;
  std::vector<HkNode> asset_nodes;
  std::vector<HkMaterial> materials;
  std::vector<HkInstance> instances;
  std::vector<HkNode> instance_nodes;
  std::vector<HkEmissive> emissives;
  std::vector<HkNode> emissive_nodes;
  std::vector<HkAliasEntry> alias_table;
  std::vector<float> prev_models;  // PreviousMeshUniform::transform per instance (optional)
  std::vector<uint8_t> noise;  // [16][64][64][4]
  struct Texture { std::vector<uint8_t> rgba; uint32_t w, h, is_srgb, au, av, linear; };
  std::vector<Texture> textures;
  float srgb_lut[256];

  int W = 0, H = 0;    // full (deferred / albedo / reservoir allocation) size
  int RW = 0, RH = 0;  // scaled render size
  int UW = 0, UH = 0;  // SMAA Tu4x output size, ceil(size * 2 / ratio)
  uint16_t* step_rec = nullptr;  // orc_debug_record_steps: [RH][RW][slots][3] u16, filled by pass_indirect
  uint32_t step_slots = 0;
  uint32_t mapped_parity = 0;  // frame parity whose planes the non-PREVIOUS ids of the double-buffered set name
  float ratio = 1.0f;
  std::vector<uint8_t> buf[HK_BUF_COUNT];

  HkFrame frame{};
  HkView view{};
  HkPreviousView pview{};
  HkLights lights{};
  bool have_frame = false;
  uint32_t taa = HK_TAA_JASMINE, upscale_kind = HK_UPSCALE_SMAA_TU4X;  // prepass jitter selection
  float upscale_sharpness = 0.0f;                                       // Upscale::sharpness(), FSR1 RCAS

  uint32_t band_index = 0, band_count = 1;
  uint32_t history_rows = 0;  // orc_set_history_rows: > 0 on a band = its scatter stores are parked and resolved across bands
  std::vector<uint32_t> band_bounds;  // orc_set_band_bounds: explicit split of the scaled render rows (band_count + 1 entries) or empty
  Stats stats;
  uint32_t flags = 0;
};

static thread_local std::string g_err;

static int buf_bpp(uint32_t b) {
  if (b == HK_BUF_POSITION || b == HK_BUF_VELOCITY_UV) return 16;
  if (b == HK_BUF_NORMAL) return 4;
  if (b == HK_BUF_DEPTH_GRADIENT || b == HK_BUF_INSTANCE_MATERIAL) return 8;
  if (b == HK_BUF_ALBEDO) return 8;
  if (b >= HK_BUF_VARIANCE0 && b < HK_BUF_VARIANCE0 + 3) return 4;
  if (b >= HK_BUF_RENDER0 && b < HK_BUF_RENDER0 + 3) return 8;
  if (b >= HK_BUF_RESERVOIR0 && b < HK_BUF_RESERVOIR0 + 10) return 64;
  if (b >= HK_BUF_DENOISE_INTERNAL0 && b < HK_BUF_DENOISE_INTERNAL0 + 4) return 8;
  if (b == HK_BUF_DENOISE_INTERNAL_VARIANCE) return 4;
  if (b >= HK_BUF_DENOISE_RENDER0 && b < HK_BUF_DENOISE_RENDER0 + 3) return 8;
  if (b == HK_BUF_TONE_MAPPED || b == HK_BUF_PREVIOUS_TONE_MAPPED) return 8;
  if (b == HK_BUF_PREVIOUS_POSITION || b == HK_BUF_PREVIOUS_VELOCITY_UV) return 16;
  if (b == HK_BUF_UPSCALE_OUTPUT || b == HK_BUF_TAA_OUTPUT || b == HK_BUF_PREVIOUS_TAA_OUTPUT || b == HK_BUF_UPSCALE_SHARPENED) return 8;
  if (b >= HK_BUF_PARKED_TO0 && b < HK_BUF_PARKED_TO0 + 3) return 4;        // parked scatter stores (bands under motion, below)
  if (b >= HK_BUF_PARKED_RECORD0 && b < HK_BUF_PARKED_RECORD0 + 3) return 64;
  return 0;
}
static bool buf_full_size(uint32_t b) {
  return b <= HK_BUF_ALBEDO || (b >= HK_BUF_RESERVOIR0 && b < HK_BUF_RESERVOIR0 + 10) || b == HK_BUF_PREVIOUS_POSITION ||
         b == HK_BUF_PREVIOUS_VELOCITY_UV || b == HK_BUF_UPSCALE_SHARPENED;
}
static bool buf_upscaled(uint32_t b) { return b == HK_BUF_UPSCALE_OUTPUT || b == HK_BUF_TAA_OUTPUT || b == HK_BUF_PREVIOUS_TAA_OUTPUT; }

// texture access helpers.  Out-of-bounds textureLoad returns zeros (wgpu robust access).
struct Tex {
  uint8_t* p;
  int w, h, bpp;
  bool in(int x, int y) const { return x >= 0 && y >= 0 && x < w && y < h; }
  v4 load_f32x4(int x, int y) const {
    if (!in(x, y)) return V4(0, 0, 0, 0);
    const float* f = (const float*)(p + ((size_t)y * w + x) * 16);
    return V4(f[0], f[1], f[2], f[3]);
  }
  v2 load_f32x2(int x, int y) const {
    if (!in(x, y)) return V2(0, 0);
    const float* f = (const float*)(p + ((size_t)y * w + x) * 8);
    return V2(f[0], f[1]);
  }
  float load_f32(int x, int y) const {
    if (!in(x, y)) return 0.0f;
    return *(const float*)(p + ((size_t)y * w + x) * 4);
  }
  v4 load_snorm8x4(int x, int y) const {
    if (!in(x, y)) return V4(0, 0, 0, 0);
    return unpack4x8snorm(*(const uint32_t*)(p + ((size_t)y * w + x) * 4));
  }
  v4 load_f16x4(int x, int y) const {
    if (!in(x, y)) return V4(0, 0, 0, 0);
    const uint16_t* h16 = (const uint16_t*)(p + ((size_t)y * w + x) * 8);
    return V4(f16_to_f32(h16[0]), f16_to_f32(h16[1]), f16_to_f32(h16[2]), f16_to_f32(h16[3]));
  }
  void store_f32x4(int x, int y, v4 v) const {
    if (!in(x, y)) return;
    float* f = (float*)(p + ((size_t)y * w + x) * 16);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  void store_f32x2(int x, int y, v2 v) const {
    if (!in(x, y)) return;
    float* f = (float*)(p + ((size_t)y * w + x) * 8);
    f[0] = v.x; f[1] = v.y;
  }
  void store_f32(int x, int y, float v) const {
    if (!in(x, y)) return;
    *(float*)(p + ((size_t)y * w + x) * 4) = v;
  }
  void store_u32(int x, int y, uint32_t v) const {
    if (!in(x, y)) return;
    *(uint32_t*)(p + ((size_t)y * w + x) * 4) = v;
  }
  void store_f16x4(int x, int y, v4 v) const {
    if (!in(x, y)) return;
    uint16_t* h16 = (uint16_t*)(p + ((size_t)y * w + x) * 8);
    h16[0] = f32_to_f16(v.x); h16[1] = f32_to_f16(v.y); h16[2] = f32_to_f16(v.z); h16[3] = f32_to_f16(v.w);
  }
  // textureSampleLevel(tex, nearest_sampler, uv, 0): nearest filter, clamp-to-edge
  // (post_process.rs sampler bind group; denoise.wgsl:5-8)
  void nearest_coords(v2 uv, int* x, int* y) const {
    int cx = (int)floorf(uv.x * (float)w), cy = (int)floorf(uv.y * (float)h);
    *x = std::min(std::max(cx, 0), w - 1);
    *y = std::min(std::max(cy, 0), h - 1);
  }
  // The 2x2 footprint of the linear sampler (post_process.rs:685-690: mag/min Linear, address mode
  // clamp-to-edge): texel centres at integer + 0.5.  The numeric contract fixes what WGSL leaves to the
  // implementation: weights are the exact f32 fractions, the blend is mix(mix(t00,t10,fx), mix(t01,t11,fx), fy).
  struct Footprint { int x0, x1, y0, y1; float fx, fy; };
  Footprint footprint(v2 uv) const {
    float px = uv.x * (float)w - 0.5f, py = uv.y * (float)h - 0.5f;
    float flx = floorf(px), fly = floorf(py);
    Footprint f;
    f.fx = px - flx;
    f.fy = py - fly;
    int ix = (int)flx, iy = (int)fly;
    f.x0 = std::min(std::max(ix, 0), w - 1);
    f.x1 = std::min(std::max(ix + 1, 0), w - 1);
    f.y0 = std::min(std::max(iy, 0), h - 1);
    f.y1 = std::min(std::max(iy + 1, 0), h - 1);
    return f;
  }
  v4 texel(int x, int y) const { return bpp == 16 ? load_f32x4(x, y) : load_f16x4(x, y); }
  v4 sample_nearest(v2 uv) const {
    int x, y;
    nearest_coords(uv, &x, &y);
    return texel(x, y);
  }
  v2 sample_nearest_f32x2(v2 uv) const {
    int x, y;
    nearest_coords(uv, &x, &y);
    return load_f32x2(x, y);
  }
  v4 sample_linear(v2 uv) const {
    Footprint f = footprint(uv);
    v4 t00 = texel(f.x0, f.y0), t10 = texel(f.x1, f.y0), t01 = texel(f.x0, f.y1), t11 = texel(f.x1, f.y1);
    auto mix4 = [](v4 a, v4 b, float t) { return V4(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t), mix(a.w, b.w, t)); };
    return mix4(mix4(t00, t10, f.fx), mix4(t01, t11, f.fx), f.fy);
  }
  // textureGather(component, ..): x = (u_min, v_max), y = (u_max, v_max), z = (u_max, v_min), w = (u_min, v_min)
  v4 gather(int component, v2 uv) const {
    Footprint f = footprint(uv);
    auto comp = [&](int x, int y) { v4 t = texel(x, y); return component == 0 ? t.x : component == 1 ? t.y : component == 2 ? t.z : t.w; };
    return V4(comp(f.x0, f.y1), comp(f.x1, f.y1), comp(f.x1, f.y0), comp(f.x0, f.y0));
  }
};
// Logical size of a buffer.  upscale_output is created at scale 2/ratio for SMAA Tu4x and taa_output at
// the scale in effect after the upscale match (post_process.rs:712-733): 2/ratio for SMAA Tu4x, 1/ratio for FSR1.
static void buf_dims(const Ctx* c, uint32_t b, int* w, int* h) {
  if (buf_full_size(b)) { *w = c->W; *h = c->H; return; }
  if (buf_upscaled(b) && c->upscale_kind == HK_UPSCALE_SMAA_TU4X) { *w = c->UW; *h = c->UH; return; }
  if (b == HK_BUF_UPSCALE_OUTPUT) { *w = c->W; *h = c->H; return; }  // FSR1: upscale_output is created at scale 1.0 (post_process.rs:723)
  *w = c->RW;
  *h = c->RH;
}
static Tex tex(Ctx* c, uint32_t b) {
  int w, h;
  buf_dims(c, b, &w, &h);
  return Tex{c->buf[b].data(), w, h, buf_bpp(b)};
}

// ------------------------------------------------------------------------------------------
// utils.wgsl
// ------------------------------------------------------------------------------------------
static inline bool is_nan(float v) { return !(v < 0.0f || 0.0f < v || v == 0.0f); }  // utils.wgsl:3-5
static inline bool any_is_nan_vec3(v3 v) { return is_nan(v.x) || is_nan(v.y) || is_nan(v.z); }
static inline uint32_t hash(uint32_t value) {  // utils.wgsl:15-24
  uint32_t state = value;
  state = state ^ 2747636419u;
  state = state * 2654435769u;
  state = state ^ (state >> 16u);
  state = state * 2654435769u;
  state = state ^ (state >> 16u);
  state = state * 2654435769u;
  return state;
}
static inline float random_float(uint32_t value) { return (float)hash(value) / 4294967295.0f; }  // utils.wgsl:26-28
static inline v2 clip_to_uv(v4 clip) {  // utils.wgsl:30-35
  v2 uv = V2(clip.x / clip.w, clip.y / clip.w);
  uv = (uv + 1.0f) * 0.5f;
  uv.y = 1.0f - uv.y;
  return uv;
}
static inline v2 coords_to_uv(int cx, int cy, int sx, int sy) {  // utils.wgsl:37-39
  return V2(((float)cx + 0.5f) / (float)sx, ((float)cy + 0.5f) / (float)sy);
}
static inline m3 normal_basis(v3 n) {  // utils.wgsl:41-48
  float s = fmin_(sign_(n.z) * 2.0f + 1.0f, 1.0f);
  float u = -1.0f / (s + n.z);
  float v = n.x * n.y * u;
  v3 t = V3(1.0f + s * n.x * n.x * u, s * v, -s * n.x);
  v3 b = V3(v, s + n.y * n.y * u, -n.y);
  return m3{t, b, n};
}
static inline float luminance(v3 v) { return dot(v, V3(0.2126f, 0.7152f, 0.0722f)); }  // utils.wgsl:63-65

// ------------------------------------------------------------------------------------------
// bevy_pbr 0.9.1 (crates/bevy_pbr/src/render/{utils,pbr_lighting}.wgsl) - un-vendored, restated
// from the published source.  Call sites: light.wgsl:738,777,805-817,828-832,905-906.
// ------------------------------------------------------------------------------------------
static inline float perceptualRoughnessToRoughness(float perceptualRoughness) {
  float c = clamp_(perceptualRoughness, 0.089f, 1.0f);
  return c * c;
}
static inline float D_GGX(float roughness, float NoH) {
  float oneMinusNoHSquared = 1.0f - NoH * NoH;
  float a = NoH * roughness;
  float k = roughness / (oneMinusNoHSquared + a * a);
  float d = k * k * (1.0f / PI);
  return d;
}
static inline float V_SmithGGXCorrelated(float roughness, float NoV, float NoL) {
  float a2 = roughness * roughness;
  float lambdaV = NoL * sqrtf((NoV - a2 * NoV) * NoV + a2);
  float lambdaL = NoV * sqrtf((NoL - a2 * NoL) * NoL + a2);
  float v = 0.5f / (lambdaV + lambdaL);
  return v;
}
static inline v3 F_Schlick_vec(v3 f0, float f90, float VoH) {
  float p = pow5_(1.0f - VoH);
  return f0 + (V3s(f90) - f0) * p;
}
static inline float F_Schlick(float f0, float f90, float VoH) { return f0 + (f90 - f0) * pow5_(1.0f - VoH); }
static inline v3 fresnel(v3 f0, float LoH) {
  float f90 = saturate(dot(f0, V3s(50.0f * 0.33f)));
  return F_Schlick_vec(f0, f90, LoH);
}
static inline v3 specular(v3 f0, float roughness, float NoV, float NoL, float NoH, float LoH, float specularIntensity) {
  float D = D_GGX(roughness, NoH);
  float V = V_SmithGGXCorrelated(roughness, NoV, NoL);
  v3 F = fresnel(f0, LoH);
  return (specularIntensity * D * V) * F;
}
static inline float Fd_Burley(float roughness, float NoV, float NoL, float LoH) {
  float f90 = 0.5f + 2.0f * roughness * LoH * LoH;
  float lightScatter = F_Schlick(1.0f, f90, NoL);
  float viewScatter = F_Schlick(1.0f, f90, NoV);
  return lightScatter * viewScatter * (1.0f / PI);
}
static inline v3 EnvBRDFApprox(v3 f0, float perceptual_roughness, float NoV) {
  const v4 c0 = V4(-1.0f, -0.0275f, -0.572f, 0.022f);
  const v4 c1 = V4(1.0f, 0.0425f, 1.04f, -0.04f);
  v4 r = perceptual_roughness * c0 + c1;
  float a004 = fmin_(r.x * r.x, exp2_(-9.28f * NoV)) * r.x + r.y;
  v2 AB = V2(-1.04f, 1.04f) * a004 + V2(r.z, r.w);
  return f0 * AB.x + AB.y;
}
// bevy_core_pipeline 0.9.1 tonemapping_shared.wgsl (call site tone_mapping.wgsl:29)
static inline v3 reinhard_luminance(v3 color) {
  float l_old = dot(color, V3(0.2126f, 0.7152f, 0.0722f));
  float l_new = l_old / (1.0f + l_old);
  return color * (l_new / l_old);
}

// ------------------------------------------------------------------------------------------
// reservoir, light.wgsl:49-223
// ------------------------------------------------------------------------------------------
struct Sample {
  v4 radiance;
  v4 random;
  v4 visible_position;
  v3 visible_normal;
  uint32_t visible_instance;
  v4 sample_position;
  v3 sample_normal;
};
struct Reservoir {
  Sample s;
  float count, lifetime, w, w_sum, w2_sum;
};
static inline uint32_t f32_to_u32(float f) {  // WGSL u32(f32): truncation, clamped
  if (!(f > 0.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
static inline int f32_to_i32(float f) {  // WGSL i32(f32): truncation, clamped
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}

static Reservoir unpack_reservoir(const PackedReservoir& packed) {  // light.wgsl:77-109
  Reservoir r{};
  v2 t0, t1;
  t0 = unpack2x16float(packed.reservoir[0]);
  t1 = unpack2x16float(packed.reservoir[1]);
  r.count = t0.x;
  r.w = t0.y;
  r.w_sum = t1.x;
  r.w2_sum = t1.y;

  t0 = unpack2x16float(packed.radiance[0]);
  t1 = unpack2x16float(packed.radiance[1]);
  r.s.radiance = V4(t0.x, t0.y, t1.x, t1.y);

  t0 = unpack2x16unorm(packed.random[0]);
  t1 = unpack2x16unorm(packed.random[1]);
  r.s.random = V4(t0.x, t0.y, t1.x, t1.y);

  v4 t2 = unpack4x8snorm(packed.visible_normal);
  r.s.visible_position = V4(packed.visible_position[0], packed.visible_position[1], packed.visible_position[2], packed.visible_position[3]);
  r.s.visible_normal = normalize(xyz(t2));
  r.lifetime = 127.0f * (1.0f + t2.w);

  t2 = unpack4x8snorm(packed.sample_normal);
  r.s.sample_position = V4(packed.sample_position[0], packed.sample_position[1], packed.sample_position[2], t2.w);
  r.s.sample_normal = normalize(xyz(t2));
  r.s.visible_instance = f32_to_u32(packed.sample_position[3]);
  return r;
}
static PackedReservoir pack_reservoir(const Reservoir& r) {  // light.wgsl:111-136
  PackedReservoir packed;
  packed.reservoir[0] = pack2x16float(V2(r.count, r.w));
  packed.reservoir[1] = pack2x16float(V2(r.w_sum, r.w2_sum));
  packed.radiance[0] = pack2x16float(V2(r.s.radiance.x, r.s.radiance.y));
  packed.radiance[1] = pack2x16float(V2(r.s.radiance.z, r.s.radiance.w));
  packed.random[0] = pack2x16unorm(V2(r.s.random.x, r.s.random.y));
  packed.random[1] = pack2x16unorm(V2(r.s.random.z, r.s.random.w));
  packed.visible_position[0] = r.s.visible_position.x;
  packed.visible_position[1] = r.s.visible_position.y;
  packed.visible_position[2] = r.s.visible_position.z;
  packed.visible_position[3] = r.s.visible_position.w;
  packed.sample_position[0] = r.s.sample_position.x;
  packed.sample_position[1] = r.s.sample_position.y;
  packed.sample_position[2] = r.s.sample_position.z;
  packed.sample_position[3] = (float)r.s.visible_instance;
  packed.visible_normal = pack4x8snorm(V4(r.s.visible_normal, r.lifetime / 127.0f - 1.0f));
  packed.sample_normal = pack4x8snorm(V4(r.s.sample_normal, r.s.sample_position.w));
  return packed;
}
static void set_reservoir(Reservoir* r, const Sample& s, float w_new) {  // light.wgsl:138-144
  r->count = 1.0f;
  r->lifetime = 0.0f;
  r->w_sum = w_new;
  r->w2_sum = w_new * w_new;
  r->s = s;
}
static void update_reservoir(Reservoir* r, const Sample& s, float w_new) {  // light.wgsl:146-173
  r->w_sum += w_new;
  r->w2_sum += w_new * w_new;
  r->count = r->count + 1.0f;
  float rand = fract(dot(s.random, V4(1.0f, 1.0f, 1.0f, 1.0f)));
  if (rand < w_new / r->w_sum) r->s = s;
}
static void merge_reservoir(Reservoir* r, const Reservoir& other, float p) {  // light.wgsl:175-179
  float count = r->count;
  update_reservoir(r, other.s, p * other.w * other.count);
  r->count = count + other.count;
}

// The four reservoir bindings of group 6 for one light channel (light.rs:518-546).
struct ReservoirSet {
  PackedReservoir* previous;          // binding 0, read
  PackedReservoir* current;           // binding 1, written
  PackedReservoir* previous_spatial;  // binding 2, read + scatter-written
  PackedReservoir* spatial;           // binding 3, written
};
static ReservoirSet reservoir_set(Ctx* c, int channel) {
  static const int T[3] = {0, 2, 6}, S[3] = {4, 4, 8};
  uint32_t cur = c->frame.number % 2u, prev = 1u - cur;  // light.rs:376 head = counter % 2
  ReservoirSet rs;
  rs.previous = (PackedReservoir*)c->buf[HK_BUF_RESERVOIR0 + cur + T[channel]].data();
  rs.current = (PackedReservoir*)c->buf[HK_BUF_RESERVOIR0 + prev + T[channel]].data();
  rs.previous_spatial = (PackedReservoir*)c->buf[HK_BUF_RESERVOIR0 + cur + S[channel]].data();
  rs.spatial = (PackedReservoir*)c->buf[HK_BUF_RESERVOIR0 + prev + S[channel]].data();
  return rs;
}
// light.wgsl:181-190 / 201-210: uv-addressed loads with the open test |uv-0.5| < 0.5
static Reservoir load_reservoir_uv(const PackedReservoir* buf, v2 uv, int sx, int sy) {
  Reservoir r{};
  if (fabsf(uv.x - 0.5f) < 0.5f && fabsf(uv.y - 0.5f) < 0.5f) {
    int cx = f32_to_i32(uv.x * (float)sx), cy = f32_to_i32(uv.y * (float)sy);
    int index = cx + sx * cy;
    r = unpack_reservoir(buf[index]);
  }
  return r;
}

// Deterministic resolution of the reference's benign write-write race on
// previous_spatial_reservoir_buffer (light.wgsl:1063,1092-1095,1199-1202,1456-1459; SURVEY 5
// "race detection"): the store of the thread with the highest linear index wins.
struct ScatterStore {
  int from, to;
  PackedReservoir value;
};

// ------------------------------------------------------------------------------------------
// tracing, light.wgsl:259-533
// ------------------------------------------------------------------------------------------
struct Ray { v3 origin, direction, inv_direction; };
struct Aabb { v3 min, max; };
struct Intersection { v2 uv; float distance; };
struct Hit { Intersection intersection; uint32_t instance_index, primitive_index; };
struct Surface { v4 base_color, emissive; float reflectance, metallic, roughness, occlusion; };
struct HitInfo { v4 position; v3 normal; v2 uv; uint32_t instance_index, material_index; };
struct LightCandidate { v3 direction; float max_distance, min_distance; uint32_t emissive_instance; float p; };

struct Scene {
  const HkVertex* vertex_buffer;
  const HkPrimitive* primitive_buffer;
  const HkNode* asset_node_buffer;
  const HkAliasEntry* alias_table_buffer;
  const HkInstance* instance_buffer;
  const HkNode* instance_node_buffer; uint32_t instance_node_count;
  const HkMaterial* material_buffer;
  const HkNode* emissive_node_buffer; uint32_t emissive_node_count;
  const HkEmissive* emissive_buffer;
  const Ctx::Texture* textures; uint32_t n_textures; const float* srgb_lut;
  const HkFrame* frame;
  const HkView* view;
  const HkLights* lights;
  uint64_t n_tlas = 0, n_blas = 0;  // per-thread ray counters
  // tests/tools/divergence_model.py: per-ray traversal work (inner-node visits, triangle tests, instance entries)
  // of the current pixel's rays, in call order; null unless orc_debug_record_steps armed it
  uint16_t* step_rec = nullptr;
  uint32_t step_slot = 0, step_slots = 0, step_base = 0;  // slot = 3 * bounce + {0 closest hit, 1 emitter BLAS ray, 2 shadow ray}
  uint32_t cur_nodes = 0, cur_tris = 0, cur_entries = 0;
  void step_flush() {
    if (step_rec && step_slot < step_slots) {
      step_rec[3 * step_slot] = (uint16_t)std::min(cur_nodes, 65535u);
      step_rec[3 * step_slot + 1] = (uint16_t)std::min(cur_tris, 65535u);
      step_rec[3 * step_slot + 2] = (uint16_t)std::min(cur_entries, 65535u);
    }
    step_slot++;
    cur_nodes = cur_tris = cur_entries = 0;
  }
};
static inline v3 P3(const float* p) { return V3(p[0], p[1], p[2]); }

static v3 instance_position_world_to_local(const HkInstance& instance, v3 p) {  // light.wgsl:306-310
  m4 inverse_model = transpose(load_m4(instance.inverse_transpose_model));
  v4 position = mul(inverse_model, V4(p, 1.0f));
  return xyz(position) / position.w;
}
static v3 instance_direction_world_to_local(const HkInstance& instance, v3 p) {  // light.wgsl:312-316
  m4 inverse_model = transpose(load_m4(instance.inverse_transpose_model));
  v4 direction = mul(inverse_model, V4(p, 0.0f));
  return xyz(direction);
}
static v3 instance_position_local_to_world(const HkInstance& instance, v3 p) {  // light.wgsl:318-322
  m4 model = load_m4(instance.model);
  v4 position = mul(model, V4(p, 1.0f));
  return xyz(position) / position.w;
}
static v3 instance_normal_local_to_world(const HkInstance& instance, v3 n) {  // light.wgsl:324-338
  const float* m = instance.inverse_transpose_model;
  m3 mm = {V3(m[0], m[1], m[2]), V3(m[4], m[5], m[6]), V3(m[8], m[9], m[10])};
  return normalize(mul(mm, n));
}
static bool inside_aabb(v3 p, Aabb aabb) {  // light.wgsl:340-342
  return p.x > aabb.min.x && p.y > aabb.min.y && p.z > aabb.min.z && p.x < aabb.max.x && p.y < aabb.max.y && p.z < aabb.max.z;
}
static float intersects_aabb(const Ray& ray, Aabb aabb) {  // light.wgsl:344-362
  v3 t1 = (aabb.min - ray.origin) * ray.inv_direction;
  v3 t2 = (aabb.max - ray.origin) * ray.inv_direction;
  float t_min = fmin_(t1.x, t2.x);
  float t_max = fmax_(t1.x, t2.x);
  t_min = fmax_(t_min, fmin_(t1.y, t2.y));
  t_max = fmin_(t_max, fmax_(t1.y, t2.y));
  t_min = fmax_(t_min, fmin_(t1.z, t2.z));
  t_max = fmin_(t_max, fmax_(t1.z, t2.z));
  float t = F32_MAX;
  if (t_max >= t_min && t_max >= 0.0f) t = t_min;
  return t;
}
static Intersection intersects_triangle(const Ray& ray, const HkPrimitiveVertex tri[3]) {  // light.wgsl:364-398
  Intersection result;
  result.uv = V2(0, 0);
  result.distance = F32_MAX;
  v3 ab = P3(tri[1].position) - P3(tri[0].position);
  v3 ac = P3(tri[2].position) - P3(tri[0].position);
  v3 u_vec = cross(ray.direction, ac);
  float det = dot(ab, u_vec);
  if (fabsf(det) < F32_EPSILON) return result;
  float inv_det = 1.0f / det;
  v3 ao = ray.origin - P3(tri[0].position);
  float u = dot(ao, u_vec) * inv_det;
  if (u < 0.0f || u > 1.0f) {
    result.uv = V2(u, 0.0f);
    return result;
  }
  v3 v_vec = cross(ao, ab);
  float v = dot(ray.direction, v_vec) * inv_det;
  result.uv = V2(u, v);
  if (v < 0.0f || u + v > 1.0f) return result;
  float distance = dot(ac, v_vec) * inv_det;
  if (distance > F32_EPSILON) result.distance = distance;
  return result;
}
static bool traverse_bottom(Scene& sc, Hit* hit, const Ray& ray, const HkMeshIndex& mesh, float early_distance) {  // light.wgsl:400-440
  bool intersected = false;
  uint32_t index = 0u;
  for (; index < mesh.node_count;) {
    uint32_t node_index = mesh.node_offset + index;
    const HkNode& node = sc.asset_node_buffer[node_index];
    Aabb aabb;
    if (node.entry_index >= BVH_LEAF_FLAG) {
      uint32_t primitive_index = mesh.primitive + node.entry_index - BVH_LEAF_FLAG;
      const HkPrimitiveVertex* vertices = sc.primitive_buffer[primitive_index].vertices;
      aabb.min = min3(P3(vertices[0].position), min3(P3(vertices[1].position), P3(vertices[2].position)));
      aabb.max = max3(P3(vertices[0].position), max3(P3(vertices[1].position), P3(vertices[2].position)));
      if (intersects_aabb(ray, aabb) < hit->intersection.distance) {
        sc.cur_tris++;
        Intersection intersection = intersects_triangle(ray, vertices);
        if (intersection.distance < hit->intersection.distance) {
          hit->intersection = intersection;
          hit->primitive_index = primitive_index;
          intersected = true;
          if (intersection.distance < early_distance) return intersected;
        }
      }
      index = node.exit_index;
    } else {
      aabb.min = P3(node.min);
      aabb.max = P3(node.max);
      sc.cur_nodes++;
      index = (intersects_aabb(ray, aabb) < hit->intersection.distance) ? node.entry_index : node.exit_index;
    }
  }
  return intersected;
}
static Hit traverse_top(Scene& sc, const Ray& ray, float max_distance, float early_distance, uint32_t exclude_instance) {  // light.wgsl:442-486
  sc.n_tlas++;
  Hit hit;
  hit.intersection.uv = V2(0, 0);
  hit.intersection.distance = max_distance;
  hit.instance_index = U32_MAX;
  hit.primitive_index = U32_MAX;
  uint32_t index = 0u;
  for (; index < sc.instance_node_count;) {
    const HkNode& node = sc.instance_node_buffer[index];
    Aabb aabb;
    if (node.entry_index >= BVH_LEAF_FLAG) {
      uint32_t instance_index = node.entry_index - BVH_LEAF_FLAG;
      const HkInstance& instance = sc.instance_buffer[instance_index];
      aabb.min = P3(instance.min);
      aabb.max = P3(instance.max);
      if (instance_index != exclude_instance && intersects_aabb(ray, aabb) < hit.intersection.distance) {
        Ray r;
        r.origin = instance_position_world_to_local(instance, ray.origin);
        r.direction = instance_direction_world_to_local(instance, ray.direction);
        r.inv_direction = 1.0f / r.direction;
        sc.cur_entries++;
        if (traverse_bottom(sc, &hit, r, instance.mesh, early_distance)) {
          hit.instance_index = instance_index;
          if (hit.intersection.distance < early_distance) { sc.step_flush(); return hit; }
        }
      }
      index = node.exit_index;
    } else {
      aabb.min = P3(node.min);
      aabb.max = P3(node.max);
      sc.cur_nodes++;
      index = (intersects_aabb(ray, aabb) < hit.intersection.distance) ? node.entry_index : node.exit_index;
    }
  }
  sc.step_flush();
  return hit;
}
static HitInfo empty_hit_info(v3 position, v3 direction) {  // light.wgsl:488-494
  HitInfo info{};
  info.instance_index = U32_MAX;
  info.material_index = U32_MAX;
  info.position = V4(position + direction * DISTANCE_MAX, 0.0f);
  return info;
}
static HitInfo hit_info(Scene& sc, const Ray& ray, const Hit& hit) {  // light.wgsl:496-523
  HitInfo info{};
  info.instance_index = hit.instance_index;
  info.material_index = U32_MAX;
  if (hit.instance_index != U32_MAX) {
    const HkInstance& instance = sc.instance_buffer[hit.instance_index];
    const HkPrimitiveVertex* vertices = sc.primitive_buffer[hit.primitive_index].vertices;
    const HkVertex& v0 = sc.vertex_buffer[instance.mesh.vertex + vertices[0].index];
    const HkVertex& v1 = sc.vertex_buffer[instance.mesh.vertex + vertices[1].index];
    const HkVertex& v2_ = sc.vertex_buffer[instance.mesh.vertex + vertices[2].index];
    v2 uv0 = V2(v0.u, v0.v), uv1 = V2(v1.u, v1.v), uv2 = V2(v2_.u, v2_.v);
    v2 uv = hit.intersection.uv;
    info.uv = uv0 + uv.x * (uv1 - uv0) + uv.y * (uv2 - uv0);
    info.normal = P3(v0.normal) + uv.x * (P3(v1.normal) - P3(v0.normal)) + uv.y * (P3(v2_.normal) - P3(v0.normal));
    info.normal = instance_normal_local_to_world(instance, info.normal);
    info.position = V4(ray.origin + ray.direction * hit.intersection.distance, 1.0f);
    info.material_index = instance.material;
  } else {
    info.position = V4(ray.origin + ray.direction * DISTANCE_MAX, 0.0f);
  }
  return info;
}
static void occlude_hit_info(const Ray& ray, const Hit& hit, HitInfo* info) {  // light.wgsl:526-533
  if (hit.instance_index != U32_MAX) {
    info->instance_index = hit.instance_index;
    info->material_index = U32_MAX;
    info->position = V4(ray.origin + ray.direction * hit.intersection.distance, 1.0f);
    info->normal = V3(0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------
// sampling, light.wgsl:537-708
// ------------------------------------------------------------------------------------------
static v2 sample_uniform_disk(v2 rand) {  // light.wgsl:537-541
  float r = sqrtf(rand.x);
  float theta = 2.0f * PI * rand.y;
  return V2(r * cos_(theta), r * sin_(theta));
}
static v4 sample_cosine_hemisphere(v2 rand) {  // light.wgsl:544-549
  v2 t = sample_uniform_disk(rand);
  v3 direction = V3(t.x, t.y, sqrtf(1.0f - dot(t, t)));
  float pdf = 2.0f * INV_TAU * direction.z;
  return V4(direction, pdf);
}
static v4 sample_uniform_cone(v2 rand, float cos_angle) {  // light.wgsl:552-559
  float z = 1.0f - (1.0f - cos_angle) * rand.x;
  float theta = TAU * rand.y;
  float r = sqrtf(1.0f - z * z);
  v3 direction = V3(r * cos_(theta), r * sin_(theta), z);
  float pdf = INV_TAU / (1.0f - cos_angle);
  return V4(direction, pdf);
}
static v2 sample_uniform_triangle_barycentric(v2 rand) {  // light.wgsl:562-565
  float srx = sqrtf(rand.x);
  return V2(1.0f - srx, rand.y * srx);
}
static v4 compute_directional_cone(const Scene& sc) {  // light.wgsl:571-573
  return V4(P3(sc.lights->direction_to_light), cos_(sc.frame->solar_angle));
}
static v3 compute_emissive_radiance(v4 emissive) {  // light.wgsl:594-596
  return 255.0f * emissive.w * xyz(emissive);
}
static LightCandidate select_light_candidate(Scene& sc, v4 rand, v3 position, v3 normal, uint32_t instance, HitInfo* info) {  // light.wgsl:599-708
  LightCandidate candidate;
  candidate.max_distance = F32_MAX;
  candidate.min_distance = DISTANCE_MAX;
  candidate.emissive_instance = DONT_SAMPLE_EMISSIVE;

  v4 cone = compute_directional_cone(sc);
  v3 rand_direction = mul(normal_basis(xyz(cone)), xyz(sample_uniform_cone(V2(rand.z, rand.w), cone.w)));
  candidate.direction = rand_direction;
  candidate.p = 1.0f;

  *info = empty_hit_info(position, rand_direction);
  if (instance == DONT_SAMPLE_EMISSIVE) return candidate;

  // Traverse the LBVH to pick one emissive within range
  HkEmissive emissive{};
  float count = 0.0f;
  uint32_t index = 0u;
  float rand_1d = rand.x;
  for (; index < sc.emissive_node_count;) {
    const HkNode& node = sc.emissive_node_buffer[index];
    Aabb aabb;
    if (node.entry_index >= BVH_LEAF_FLAG) {
      uint32_t emissive_index = node.entry_index - BVH_LEAF_FLAG;
      const HkEmissive& current_emissive = sc.emissive_buffer[emissive_index];
      aabb.min = P3(current_emissive.position) - current_emissive.radius;
      aabb.max = P3(current_emissive.position) + current_emissive.radius;
      if (instance != current_emissive.instance && inside_aabb(position, aabb)) {
        rand_1d = fract(rand_1d + GOLDEN_RATIO);
        count += 1.0f;
        if (rand_1d < 1.0f / count) {
          candidate.emissive_instance = current_emissive.instance;
          emissive = current_emissive;
        }
      }
      index = node.exit_index;
    } else {
      aabb.min = P3(node.min);
      aabb.max = P3(node.max);
      index = inside_aabb(position, aabb) ? node.entry_index : node.exit_index;
    }
  }

  if (candidate.emissive_instance != DONT_SAMPLE_EMISSIVE) {
    uint32_t alias_index = std::min(f32_to_u32(rand.x * (float)emissive.alias_table[1]), emissive.alias_table[1] - 1u);
    const HkAliasEntry& alias_entry = sc.alias_table_buffer[emissive.alias_table[0] + alias_index];
    uint32_t primitive_index = (rand.y < alias_entry.prob) ? alias_entry.index : alias_index;

    const HkInstance& emissive_instance = sc.instance_buffer[candidate.emissive_instance];
    const HkPrimitiveVertex* v = sc.primitive_buffer[emissive_instance.mesh.primitive + primitive_index].vertices;
    v2 b = sample_uniform_triangle_barycentric(V2(rand.z, rand.w));
    v3 p = instance_position_local_to_world(
        emissive_instance, b.x * P3(v[0].position) + b.y * P3(v[1].position) + (1.0f - b.x - b.y) * P3(v[2].position));

    Hit hit;
    hit.intersection.uv = V2(0, 0);
    hit.intersection.distance = F32_MAX;
    hit.instance_index = U32_MAX;
    hit.primitive_index = U32_MAX;

    Ray ray;
    ray.origin = position + normal * RAY_BIAS;
    ray.direction = normalize(p - position);
    ray.inv_direction = V3(0, 0, 0);  // never read (hit_info only uses origin/direction)

    Ray r;
    r.origin = instance_position_world_to_local(emissive_instance, ray.origin);
    r.direction = instance_direction_world_to_local(emissive_instance, ray.direction);
    r.inv_direction = 1.0f / r.direction;

    candidate.direction = ray.direction;
    bool front = dot(candidate.direction, normal) > 0.0f;
    if (front) sc.n_blas++;
    sc.step_slot = sc.step_base + 1u;
    const bool blas_hit = front && traverse_bottom(sc, &hit, r, emissive_instance.mesh, 0.0f);
    sc.step_flush();
    if (blas_hit) {
      hit.instance_index = emissive.instance;
      *info = hit_info(sc, ray, hit);
      candidate.max_distance = hit.intersection.distance;
      candidate.min_distance = hit.intersection.distance - 0.1f;
      v3 delta = xyz(info->position) - position;
      candidate.p = dot(delta, delta) / (fabsf(dot(ray.direction, info->normal) * emissive.surface_area));
      candidate.p = candidate.p / count;
    } else {
      *info = empty_hit_info(ray.origin, ray.direction);
      candidate.emissive_instance = DONT_SAMPLE_EMISSIVE;
      candidate.direction = rand_direction;
      candidate.p = 1.0f;
    }
  }
  return candidate;
}

// ------------------------------------------------------------------------------------------
// shading, light.wgsl:711-908
// ------------------------------------------------------------------------------------------
static v3 calculate_view(const Scene& sc, v4 world_position, bool is_orthographic) {  // light.wgsl:714-727
  if (is_orthographic) {
    const float* vp = sc.view->view_proj;
    return normalize(V3(vp[2], vp[6], vp[10]));  // view_proj[0].z, [1].z, [2].z
  }
  return normalize(P3(sc.view->world_position) - xyz(world_position));
}
static inline v4 P4(const float* p) { return V4(p[0], p[1], p[2], p[3]); }
// textureSampleLevel(textures[id], samplers[id], uv, 0.0) (light.wgsl:756-789).  WebGPU leaves the
// filtering arithmetic to the implementation; the contract here: texel centres at (i + 0.5) / size,
// f32 bilinear weights, sRGB rgb decoded to linear per texel before filtering (as texture units do),
// mix() of the numeric contract.
static int wrap_coord(int i, int n, uint32_t mode) {
  if (mode == HK_ADDRESS_REPEAT) { int m = i % n; return m < 0 ? m + n : m; }
  if (mode == HK_ADDRESS_MIRROR_REPEAT) { int p = 2 * n; int m = i % p; if (m < 0) m += p; return m < n ? m : p - 1 - m; }
  return std::min(std::max(i, 0), n - 1);
}
static v4 texel(const Scene& sc, const Ctx::Texture& t, int x, int y) {
  const uint8_t* p = t.rgba.data() + ((size_t)y * t.w + x) * 4;
  if (t.is_srgb) return V4(sc.srgb_lut[p[0]], sc.srgb_lut[p[1]], sc.srgb_lut[p[2]], (float)p[3] / 255.0f);
  return V4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
}
static inline v4 mix4(v4 a, v4 b, float t) { return V4(mix(a.x, b.x, t), mix(a.y, b.y, t), mix(a.z, b.z, t), mix(a.w, b.w, t)); }
static v4 sample_texture(const Scene& sc, uint32_t id, v2 uv) {
  const Ctx::Texture& t = sc.textures[id];
  const int w = (int)t.w, h = (int)t.h;
  if (!t.linear) {
    int x = wrap_coord(f32_to_i32(floorf(uv.x * (float)w)), w, t.au), y = wrap_coord(f32_to_i32(floorf(uv.y * (float)h)), h, t.av);
    return texel(sc, t, x, y);
  }
  float x = uv.x * (float)w - 0.5f, y = uv.y * (float)h - 0.5f;
  float x0 = floorf(x), y0 = floorf(y);
  float fx = x - x0, fy = y - y0;
  int ix = f32_to_i32(x0), iy = f32_to_i32(y0);
  int xa = wrap_coord(ix, w, t.au), xb = wrap_coord(ix + 1, w, t.au), ya = wrap_coord(iy, h, t.av), yb = wrap_coord(iy + 1, h, t.av);
  v4 top = mix4(texel(sc, t, xa, ya), texel(sc, t, xb, ya), fx);
  v4 bot = mix4(texel(sc, t, xa, yb), texel(sc, t, xb, yb), fx);
  return mix4(top, bot, fy);
}
static Surface retreive_surface(const Scene& sc, uint32_t material_index, v2 uv) {  // light.wgsl:730-742 (NO_TEXTURE) / 749-781
  Surface surface;
  const HkMaterial& material = sc.material_buffer[material_index];
  surface.base_color = P4(material.base_color);
  surface.emissive = P4(material.emissive);
  surface.metallic = material.metallic;
  surface.occlusion = 1.0f;
  if (sc.n_textures) {
    uint32_t id = material.base_color_texture;
    if (id != U32_MAX) surface.base_color = surface.base_color * sample_texture(sc, id, uv);
    id = material.emissive_texture;
    if (id != U32_MAX) surface.emissive = surface.emissive * sample_texture(sc, id, uv);
    id = material.metallic_roughness_texture;
    if (id != U32_MAX) surface.metallic *= sample_texture(sc, id, uv).x;
    id = material.occlusion_texture;
    if (id != U32_MAX) surface.occlusion = sample_texture(sc, id, uv).x;
  }
  surface.roughness = perceptualRoughnessToRoughness(material.perceptual_roughness);
  surface.reflectance = material.reflectance;
  return surface;
}
static v4 retreive_emissive(const Scene& sc, uint32_t material_index, v2 uv) {  // light.wgsl:744-747 / 783-793
  const HkMaterial& material = sc.material_buffer[material_index];
  v4 emissive = P4(material.emissive);
  if (sc.n_textures && material.emissive_texture != U32_MAX) emissive = emissive * sample_texture(sc, material.emissive_texture, uv);
  return emissive;
}
static v3 lit(v3 radiance, v3 diffuse_color, float roughness, v3 F0, v3 L, v3 N, v3 V) {  // light.wgsl:796-818
  v3 Hh = normalize(L + V);
  float NoL = saturate(dot(N, L));
  float NoH = saturate(dot(N, Hh));
  float LoH = saturate(dot(L, Hh));
  float NdotV = fmax_(dot(N, V), 0.0001f);
  v3 diffuse = diffuse_color * Fd_Burley(roughness, NdotV, NoL, LoH);
  float specular_intensity = 1.0f;
  v3 specular_light = specular(F0, roughness, NdotV, NoL, NoH, LoH, specular_intensity);
  return (specular_light + diffuse) * radiance * NoL;
}
static v3 ambient(const Scene& sc, v3 diffuse_color, float roughness, float occlusion, v3 F0, v3 N, v3 V) {  // light.wgsl:820-833
  float NdotV = fmax_(dot(N, V), 0.0001f);
  v3 diffuse_ambient = EnvBRDFApprox(diffuse_color, 1.0f, NdotV);
  v3 specular_ambient = EnvBRDFApprox(F0, roughness, NdotV);
  return occlusion * (diffuse_ambient + specular_ambient) * V3(sc.lights->ambient_color[0], sc.lights->ambient_color[1], sc.lights->ambient_color[2]);
}
static v4 input_radiance(const Scene& sc, const Ray& ray, const HitInfo& info, bool sample_directional, uint32_t sample_emissive, bool sample_ambient) {  // light.wgsl:835-867
  v3 radiance = V3(0, 0, 0);
  float ambient_ = 0.0f;
  if (info.instance_index == U32_MAX) {
    v4 cone = compute_directional_cone(sc);
    bool hit_directional = dot(ray.direction, xyz(cone)) >= cone.w;
    if (sample_directional && hit_directional) {
      radiance = V3(sc.lights->directional_color[0], sc.lights->directional_color[1], sc.lights->directional_color[2]);
      ambient_ = 0.0f;
    } else {
      radiance = sample_ambient ? V3(sc.lights->ambient_color[0], sc.lights->ambient_color[1], sc.lights->ambient_color[2]) : V3(0, 0, 0);
      ambient_ = 1.0f;
    }
  } else {
    if (sample_emissive == info.instance_index) {
      v4 emissive = retreive_emissive(sc, info.material_index, info.uv);
      radiance = compute_emissive_radiance(emissive);
    }
  }
  return V4(radiance, 1.0f - ambient_);
}
static v3 shading(const Scene& sc, v3 V, v3 N, v3 L, const Surface& surface, v4 in_radiance) {  // light.wgsl:869-888
  v3 base_color = xyz(surface.base_color);
  float reflectance = surface.reflectance;
  float roughness = surface.roughness;
  float metallic = surface.metallic;
  float occlusion = surface.occlusion;
  v3 F0 = V3s(0.16f * reflectance * reflectance * (1.0f - metallic)) + base_color * metallic;
  v3 diffuse_color = base_color * (1.0f - metallic);
  v3 lit_radiance = lit(xyz(in_radiance), diffuse_color, roughness, F0, L, N, V);
  v3 ambient_radiance = ambient(sc, diffuse_color, roughness, occlusion, F0, N, V);
  return mix(lit_radiance, ambient_radiance, 1.0f - in_radiance.w);
}
static v3 env_brdf(v3 V, v3 N, const Surface& surface) {  // light.wgsl:890-908
  v3 base_color = xyz(surface.base_color);
  float reflectance = surface.reflectance;
  float roughness = surface.roughness;
  float metallic = surface.metallic;
  float occlusion = surface.occlusion;
  float NdotV = fmax_(dot(N, V), 0.0001f);
  v3 F0 = V3s(0.16f * reflectance * reflectance * (1.0f - metallic)) + base_color * metallic;
  v3 diffuse_color = base_color * (1.0f - metallic);
  v3 diffuse_ambient = EnvBRDFApprox(diffuse_color, 1.0f, NdotV);
  v3 specular_ambient = EnvBRDFApprox(F0, roughness, NdotV);
  return occlusion * (diffuse_ambient + specular_ambient);
}

// ------------------------------------------------------------------------------------------
// ReSTIR helpers, light.wgsl:911-1017
// ------------------------------------------------------------------------------------------
static float reservoir_lifetime(const Scene& sc) {  // light.wgsl:913-915
  return (sc.frame->max_reservoir_lifetime <= 1.0f) ? F32_MAX : sc.frame->max_reservoir_lifetime;
}
static bool check_previous_reservoir(Reservoir* r, const Sample& s) {  // light.wgsl:917-935
  float depth_ratio = r->s.visible_position.w / s.visible_position.w;
  depth_ratio = (depth_ratio < 1.0f) ? 1.0f / depth_ratio : depth_ratio;
  bool depth_miss = depth_ratio > 1.05f * (1.0f + 0.5f * s.random.x);
  bool instance_miss = r->s.visible_instance != s.visible_instance;
  bool normal_miss = dot(s.visible_normal, r->s.visible_normal) < 0.9f;
  if (depth_miss || normal_miss || instance_miss) {
    *r = Reservoir{};
    return false;
  }
  return true;
}
static void temporal_restir(Reservoir* r, const Sample& s, float w_new, uint32_t max_sample_count) {  // light.wgsl:937-952
  update_reservoir(r, s, w_new);
  float m = (float)max_sample_count;
  if (r->count > m) {
    r->w_sum *= m / r->count;
    r->w2_sum *= m / r->count;
    r->count = m;
  }
}
static float compute_jacobian(const Sample& q, const Sample& r) {  // light.wgsl:985-1004
  v3 normal = q.sample_normal;
  float cos_phi_1 = fabsf(dot(normalize(xyz(r.visible_position) - xyz(q.sample_position)), normal));
  float cos_phi_2 = fabsf(dot(normalize(xyz(q.visible_position) - xyz(q.sample_position)), normal));
  float term_1 = cos_phi_1 / fmax_(0.0001f, cos_phi_2);
  float num = length(xyz(q.visible_position) - xyz(q.sample_position));
  num *= num;
  float denom = length(xyz(r.visible_position) - xyz(q.sample_position));
  denom *= denom;
  float term_2 = num / fmax_(denom, 0.0001f);
  float jacobian = term_1 * term_2;
  return clamp_(jacobian, 1.0f, 50.0f);
}
struct Sizes { int dw, dh, rw, rh; };
static v2 jittered_deferred_uv(const Scene& sc, const Sizes& sz, v2 uv, float amount) {  // light.wgsl:1007-1011 (0.25), denoise.wgsl:37-41 (0.5)
  v2 texel_size = V2(1.0f / (float)sz.dw, 1.0f / (float)sz.dh);
  float ratio = sc.frame->upscale_ratio - 1.0f;
  float sgn = ((sc.frame->number & 1u) == 0u) ? -amount : amount;
  return uv + sgn * texel_size * ratio;
}
static void jittered_deferred_coords(const Scene& sc, const Sizes& sz, v2 uv, int* cx, int* cy) {  // light.wgsl:1013-1017
  v2 duv = jittered_deferred_uv(sc, sz, uv, 0.25f);
  *cx = f32_to_i32(duv.x * (float)sz.dw);
  *cy = f32_to_i32(duv.y * (float)sz.dh);
}

static Scene make_scene(Ctx* c) {
  Scene sc{};
  sc.vertex_buffer = c->vertices.data();
  sc.primitive_buffer = c->primitives.data();
  sc.asset_node_buffer = c->asset_nodes.data();
  sc.alias_table_buffer = c->alias_table.data();
  sc.instance_buffer = c->instances.data();
  sc.instance_node_buffer = c->instance_nodes.data();
  sc.instance_node_count = (uint32_t)c->instance_nodes.size();
  sc.material_buffer = c->materials.data();
  sc.emissive_node_buffer = c->emissive_nodes.data();
  sc.emissive_node_count = (uint32_t)c->emissive_nodes.size();
  sc.emissive_buffer = c->emissives.data();
  sc.textures = c->textures.data();
  sc.n_textures = (uint32_t)c->textures.size();
  sc.srgb_lut = c->srgb_lut;
  sc.frame = &c->frame;
  sc.view = &c->view;
  sc.lights = &c->lights;
  return sc;
}
static v4 noise_fetch(Ctx* c, int x, int y, uint32_t n) {  // light.wgsl:1075-1078: nearest + repeat sampler
  uint32_t noise_id = n % NOISE_TEXTURE_COUNT;
  // noise_uv = (coords + f32(n) + 0.5) / 64 -> texel ((x+n) mod 64, (y+n) mod 64), exact in f32
  float fx = ((float)x + (float)n + 0.5f) / 64.0f, fy = ((float)y + (float)n + 0.5f) / 64.0f;
  int tx = (int)floorf(fract(fx) * 64.0f) & 63, ty = (int)floorf(fract(fy) * 64.0f) & 63;
  const uint8_t* t = c->noise.data() + (((size_t)noise_id * 64 + ty) * 64 + tx) * 4;
  return V4((float)t[0] / 255.0f, (float)t[1] / 255.0f, (float)t[2] / 255.0f, (float)t[3] / 255.0f);
}

// A band of a sharded frame with a history halo (hk_frame_stage in hikari_hip.h, HK_STAGE_SPATIAL_WITH_HISTORY): the stores are
// not applied but PARKED in the HK_BUF_PARKED_* planes of the channel - pixel `from` leaves its slot and its record, a background
// pixel its own slot - so that the bands can hand each other the rows near their borders; resolve_parked applies them at the
// start of stage SPATIAL, the highest pixel index winning as in apply_scatter.
static bool parks_across_bands(const Ctx* c) { return c->band_count > 1 && c->history_rows > 0; }
static void park_scatter(Ctx* c, int channel, std::vector<std::vector<ScatterStore>>& rows, const PackedReservoir* dst, const std::vector<uint8_t>& own_written,
                         int y0, int y1) {
  int32_t* to = (int32_t*)c->buf[HK_BUF_PARKED_TO0 + channel].data();
  PackedReservoir* rec = (PackedReservoir*)c->buf[HK_BUF_PARKED_RECORD0 + channel].data();
  const int rw = c->RW;
  for (int i = y0 * rw; i < y1 * rw; ++i) {
    to[i] = own_written[i] ? i : -1;
    if (own_written[i]) rec[i] = dst[i];
  }
  for (auto& row : rows)
    for (auto& st : row) {  // (a pixel that stores twice, light.wgsl:1092-1095 then 1199-1202, keeps the later record: same slot)
      to[st.from] = st.to;
      rec[st.from] = st.value;
    }
}
static void resolve_parked(Ctx* c, int channel, PackedReservoir* dst, int y0, int y1) {
  const int32_t* to = (const int32_t*)c->buf[HK_BUF_PARKED_TO0 + channel].data();
  const PackedReservoir* rec = (const PackedReservoir*)c->buf[HK_BUF_PARKED_RECORD0 + channel].data();
  for (int i = y0 * c->RW; i < y1 * c->RW; ++i)
    if (to[i] >= 0) dst[to[i]] = rec[i];  // increasing pixel index: the highest one that stores to a slot stays
}
static void apply_scatter(std::vector<std::vector<ScatterStore>>& rows, PackedReservoir* dst, const std::vector<uint8_t>& own_written) {
  for (auto& row : rows)
    for (auto& st : row) {
      if (own_written[st.to] && st.to > st.from) continue;  // the slot's own (higher-index) thread wins
      dst[st.to] = st.value;
    }
}

// ------------------------------------------------------------------------------------------
// G-buffer by primary rays.  The reference rasterises (prepass.wgsl:40-100, prepass.rs:43-47);
// there is no reference arithmetic to restate for coverage, so this function DEFINES the
// ray-cast equivalent of the raster contract: pixel-centre sampling, geometry
// shifted by the TAA jitter (prepass.wgsl:30-38,53,71), nearest surface wins, interpolated
// per-vertex world normal (normalised per vertex as bevy_pbr::mesh_functions does, not
// re-normalised after interpolation) stored as rgba8snorm, depth = clip z/w, analytic per-pixel
// depth gradient on the hit triangle's plane, ids + 0.5, velocity from un-jittered reprojection
// with the camera's previous view-projection and, for instances that moved, their previous model.
// ------------------------------------------------------------------------------------------
static v2 frame_jitter(Ctx* c) {  // prepass.wgsl:30-38
  uint32_t index = (c->upscale_kind == HK_UPSCALE_SMAA_TU4X) ? ((c->frame.number >> 1u) & 15u) : (c->frame.number & 15u);
  const float* h = c->frame.halton[index >> 1u];
  return ((index & 1u) == 0u) ? V2(h[0], h[1]) : V2(h[2], h[3]);
}
static v3 primary_near_point(Ctx* c, float px, float py, v2 jitter_ndc) {
  float ndc_x = (px + 0.5f) / (float)c->W * 2.0f - 1.0f - jitter_ndc.x;
  float ndc_y = 1.0f - (py + 0.5f) / (float)c->H * 2.0f - jitter_ndc.y;
  v4 pn = mul(load_m4(c->view.inverse_view_proj), V4(ndc_x, ndc_y, 1.0f, 1.0f));  // reverse-Z: z = 1 is the near plane
  return xyz(pn) / pn.w;
}
static Ray primary_ray(Ctx* c, float px, float py, v2 jitter_ndc) {
  // pixel centre (px+0.5, py+0.5) in NDC, minus the geometry shift
  v3 near_point = primary_near_point(c, px, py, jitter_ndc);
  Ray ray;
  if (c->view.projection[15] == 1.0f) {  // orthographic (light.wgsl:1040 test)
    const float* vp = c->view.view_proj;
    ray.origin = near_point;
    ray.direction = -normalize(V3(vp[2], vp[6], vp[10]));
  } else {
    ray.origin = P3(c->view.world_position);
    ray.direction = normalize(near_point - ray.origin);
  }
  ray.inv_direction = 1.0f / ray.direction;
  return ray;
}
static void pass_prepass(Ctx* c, int y0, int y1) {
  Tex position = tex(c, HK_BUF_POSITION), normal = tex(c, HK_BUF_NORMAL), dgrad = tex(c, HK_BUF_DEPTH_GRADIENT),
      im = tex(c, HK_BUF_INSTANCE_MATERIAL), vuv = tex(c, HK_BUF_VELOCITY_UV);
  v2 jitter_ndc = V2(0, 0);
  if (c->taa != HK_TAA_NONE) {  // TEMPORAL_ANTI_ALIASING, prepass.rs:489, prepass.wgsl:52-54,71
    v2 j = frame_jitter(c);
    jitter_ndc = V2(2.0f * j.x * (1.0f / c->view.viewport[2]), -(2.0f * j.y * (1.0f / c->view.viewport[3])));
  }
  m4 view_proj = load_m4(c->view.view_proj), prev_view_proj = load_m4(c->pview.view_proj);
  uint64_t n_primary = 0, n_tlas = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : n_primary, n_tlas)
  for (int y = y0; y < y1; ++y) {
    Scene sc = make_scene(c);
    for (int x = 0; x < c->W; ++x) {
      Ray ray = primary_ray(c, (float)x, (float)y, jitter_ndc);
      Hit hit = traverse_top(sc, ray, F32_MAX, 0.0f, DONT_EXCLUDE);
      // Near-plane clip (prepass.rs:242-266: the raster pipeline the primary rays stand in for has `unclipped_depth: false` - geometry
      // in front of the near plane never reaches the G-buffer).  A perspective ray starts at the eye; if its closest hit lies in
      // front of the near plane the pixel is traced again from the near plane: the nearest surface BEYOND it, a triangle that
      // straddles it cut at it.  (Orthographic rays start on the near plane.)
      if (c->view.projection[15] != 1.0f && hit.instance_index != U32_MAX) {
        const v3 near_point = primary_near_point(c, (float)x, (float)y, jitter_ndc);
        if (hit.intersection.distance < length(near_point - ray.origin)) {
          ray.origin = near_point;
          hit = traverse_top(sc, ray, F32_MAX, 0.0f, DONT_EXCLUDE);
        }
      }
      n_primary++;
      if (hit.instance_index == U32_MAX) {  // LoadOp::Clear(Color::NONE), prepass.rs:792
        position.store_f32x4(x, y, V4(0, 0, 0, 0));
        normal.store_u32(x, y, 0u);
        dgrad.store_f32x2(x, y, V2(0, 0));
        im.store_f32x2(x, y, V2(0, 0));
        vuv.store_f32x4(x, y, V4(0, 0, 0, 0));
        continue;
      }
      const HkInstance& instance = sc.instance_buffer[hit.instance_index];
      const HkPrimitiveVertex* pv = sc.primitive_buffer[hit.primitive_index].vertices;
      const HkVertex& v0 = sc.vertex_buffer[instance.mesh.vertex + pv[0].index];
      const HkVertex& v1 = sc.vertex_buffer[instance.mesh.vertex + pv[1].index];
      const HkVertex& v2_ = sc.vertex_buffer[instance.mesh.vertex + pv[2].index];
      v2 b = hit.intersection.uv;
      v3 world_position = ray.origin + ray.direction * hit.intersection.distance;
      v4 clip = mul(view_proj, V4(world_position, 1.0f));
      float depth = clip.z / clip.w;
      // world normal: per-vertex normalize(mat3(inverse_transpose_model) * n), then interpolate
      v3 n0 = instance_normal_local_to_world(instance, P3(v0.normal));
      v3 n1 = instance_normal_local_to_world(instance, P3(v1.normal));
      v3 n2 = instance_normal_local_to_world(instance, P3(v2_.normal));
      v3 wn = n0 + b.x * (n1 - n0) + b.y * (n2 - n0);
      v2 uv = V2(v0.u, v0.v) + b.x * (V2(v1.u, v1.v) - V2(v0.u, v0.v)) + b.y * (V2(v2_.u, v2_.v) - V2(v0.u, v0.v));
      // analytic depth gradient: neighbouring pixel rays against the hit triangle's world plane
      v3 p0 = instance_position_local_to_world(instance, P3(pv[0].position));
      v3 p1 = instance_position_local_to_world(instance, P3(pv[1].position));
      v3 p2 = instance_position_local_to_world(instance, P3(pv[2].position));
      v3 ng = cross(p1 - p0, p2 - p0);
      float grad[2];
      for (int k = 0; k < 2; ++k) {
        Ray rn = primary_ray(c, (float)x + (k == 0 ? 1.0f : 0.0f), (float)y + (k == 1 ? 1.0f : 0.0f), jitter_ndc);
        float tn = dot(ng, p0 - rn.origin) / dot(ng, rn.direction);
        v3 wp = rn.origin + rn.direction * tn;
        v4 cn = mul(view_proj, V4(wp, 1.0f));
        grad[k] = cn.z / cn.w - depth;
      }
      // prepass.wgsl:50,96: previous_world_position = previous_mesh.model * vertex, interpolated over the
      // triangle; only evaluated for instances whose previous model differs (else it IS world_position)
      v4 previous_world = V4(world_position, 1.0f);
      if (c->prev_models.size() == 16 * c->instances.size() &&
          memcmp(&c->prev_models[16 * (size_t)hit.instance_index], instance.model, 64) != 0) {
        m4 pm = load_m4(&c->prev_models[16 * (size_t)hit.instance_index]);
        v3 local = P3(pv[0].position) + b.x * (P3(pv[1].position) - P3(pv[0].position)) + b.y * (P3(pv[2].position) - P3(pv[0].position));
        previous_world = mul(pm, V4(local, 1.0f));
      }
      v2 velocity = clip_to_uv(clip) - clip_to_uv(mul(prev_view_proj, previous_world));
      position.store_f32x4(x, y, V4(world_position, depth));
      normal.store_u32(x, y, pack4x8snorm(V4(wn, 1.0f)));
      dgrad.store_f32x2(x, y, V2(grad[0], grad[1]));
      im.store_f32x2(x, y, V2((float)hit.instance_index + 0.5f, (float)instance.material + 0.5f));
      vuv.store_f32x4(x, y, V4(velocity.x, velocity.y, uv.x, uv.y));
    }
    n_tlas += sc.n_tlas;
  }
  (void)n_tlas;  // primary rays are counted as primary, not as TLAS rays
  c->stats.rays_primary += n_primary;
}

// ------------------------------------------------------------------------------------------
// full_screen_albedo, light.wgsl:1019-1042 (grid = full size, light.rs:651)
// ------------------------------------------------------------------------------------------
static void pass_full_screen_albedo(Ctx* c, int y0, int y1) {
  Tex position_texture = tex(c, HK_BUF_POSITION), normal_texture = tex(c, HK_BUF_NORMAL),
      instance_material_texture = tex(c, HK_BUF_INSTANCE_MATERIAL), velocity_uv_texture = tex(c, HK_BUF_VELOCITY_UV),
      albedo_texture = tex(c, HK_BUF_ALBEDO);
#pragma omp parallel for schedule(static)
  for (int y = y0; y < y1; ++y) {
    Scene sc = make_scene(c);
    for (int x = 0; x < c->W; ++x) {
      v4 position_depth = position_texture.load_f32x4(x, y);
      v4 position = V4(xyz(position_depth), 1.0f);
      float depth = position_depth.w;
      if (depth < F32_EPSILON) {
        albedo_texture.store_f16x4(x, y, V4(0, 0, 0, 0));
        continue;
      }
      v3 normal = xyz(normal_texture.load_snorm8x4(x, y));
      v2 imf = instance_material_texture.load_f32x2(x, y);
      uint32_t material = f32_to_u32(imf.y);
      v4 velocity_uv = velocity_uv_texture.load_f32x4(x, y);
      Surface surface = retreive_surface(sc, material, V2(velocity_uv.z, velocity_uv.w));
      v3 view_direction = calculate_view(sc, position, c->view.projection[15] == 1.0f);
      albedo_texture.store_f16x4(x, y, V4(env_brdf(view_direction, normal, surface), 1.0f));
    }
  }
}

// ------------------------------------------------------------------------------------------
// direct_lit, light.wgsl:1044-1261.  emissive_lit = EMISSIVE_LIT variant (direct_emissive
// pipeline, light.rs:415-420); otherwise RENDER_EMISSIVE (direct_lit pipeline, light.rs:409-414)
// ------------------------------------------------------------------------------------------
static void pass_direct_lit(Ctx* c, bool emissive_lit, int y0, int y1) {
  const int channel = emissive_lit ? 1 : 0;
  Tex position_texture = tex(c, HK_BUF_POSITION), normal_texture = tex(c, HK_BUF_NORMAL),
      instance_material_texture = tex(c, HK_BUF_INSTANCE_MATERIAL), velocity_uv_texture = tex(c, HK_BUF_VELOCITY_UV),
      variance_texture = tex(c, HK_BUF_VARIANCE0 + channel), render_texture = tex(c, HK_BUF_RENDER0 + channel);
  ReservoirSet rs = reservoir_set(c, channel);
  Sizes sz{c->W, c->H, c->RW, c->RH};
  const int rw = c->RW;
  std::vector<std::vector<ScatterStore>> scatter(c->RH);
  std::vector<uint8_t> own_written((size_t)c->RW * c->RH, 0);
  uint64_t n_tlas = 0, n_blas = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : n_tlas, n_blas)
  for (int y = y0; y < y1; ++y) {
    Scene sc = make_scene(c);
    const HkFrame& frame = c->frame;
    for (int x = 0; x < rw; ++x) {
      const int index = x + rw * y;
      v2 uv = coords_to_uv(x, y, sz.rw, sz.rh);
      Sample s{};
      int dcx, dcy;
      jittered_deferred_coords(sc, sz, uv, &dcx, &dcy);
      v4 position_depth = position_texture.load_f32x4(dcx, dcy);
      v4 position = V4(xyz(position_depth), 1.0f);
      float depth = position_depth.w;

      if (depth < F32_EPSILON) {  // light.wgsl:1058-1069
        Reservoir r{};
        set_reservoir(&r, s, 0.0f);
        PackedReservoir pr = pack_reservoir(r);
        rs.current[index] = pr;
        rs.spatial[index] = pr;
        rs.previous_spatial[index] = pr;
        own_written[index] = 1;
        variance_texture.store_f32(x, y, 0.0f);
        render_texture.store_f16x4(x, y, V4(0, 0, 0, 0));
        continue;
      }

      v3 normal = xyz(normal_texture.load_snorm8x4(dcx, dcy));
      v2 imf = instance_material_texture.load_f32x2(dcx, dcy);
      uint32_t im_x = f32_to_u32(imf.x), im_y = f32_to_u32(imf.y);
      v4 velocity_uv = velocity_uv_texture.load_f32x4(dcx, dcy);

      s.random = noise_fetch(c, x, y, frame.number);
      s.random = fract(s.random + (float)frame.number * GOLDEN_RATIO);

      s.visible_position = V4(xyz(position), depth);
      s.visible_normal = normal;
      s.visible_instance = im_x;

      Ray ray{};
      Hit hit{};
      HitInfo info{};

      v2 previous_uv = jittered_deferred_uv(sc, sz, uv, 0.25f) - V2(velocity_uv.x, velocity_uv.y);
      Reservoir r = load_reservoir_uv(rs.previous, previous_uv, sz.rw, sz.rh);
      const bool prev_on_screen = fabsf(previous_uv.x - 0.5f) <= 0.5f && fabsf(previous_uv.y - 0.5f) <= 0.5f;
      const int previous_index = f32_to_i32(previous_uv.x * (float)sz.rw) + rw * f32_to_i32(previous_uv.y * (float)sz.rh);

      if (!check_previous_reservoir(&r, s) && prev_on_screen) {  // light.wgsl:1092-1095
        scatter[y].push_back({index, previous_index, pack_reservoir(r)});
      }

      const uint32_t validate_interval = emissive_lit ? frame.emissive_validate_interval : frame.direct_validate_interval;
      const uint32_t select_light_instance = emissive_lit ? im_x : DONT_SAMPLE_EMISSIVE;

      // Non-validation frame, or sample count too low, light.wgsl:1108-1153
      if (frame.number % validate_interval != 0u || r.count < (float)DIRECT_VALIDATION_FRAME_SAMPLE_THRESHOLD) {
        LightCandidate candidate = select_light_candidate(sc, s.random, xyz(s.visible_position), s.visible_normal, select_light_instance, &info);
        ray.origin = xyz(position) + normal * RAY_BIAS;
        ray.direction = candidate.direction;
        ray.inv_direction = 1.0f / ray.direction;

        bool trace_condition = dot(candidate.direction, normal) > 0.0f;
        trace_condition = trace_condition && candidate.p > 0.0f;
        if (emissive_lit) trace_condition = trace_condition && candidate.emissive_instance != DONT_SAMPLE_EMISSIVE;

        if (trace_condition) {
          hit = traverse_top(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance);
          occlude_hit_info(ray, hit, &info);
          if (emissive_lit)
            s.radiance = input_radiance(sc, ray, info, false, candidate.emissive_instance, false);
          else
            s.radiance = input_radiance(sc, ray, info, true, DONT_SAMPLE_EMISSIVE, false);
        }
        s.sample_position = info.position;
        s.sample_normal = info.normal;
        float w_new = (candidate.p > 0.0f) ? luminance(xyz(s.radiance)) / candidate.p : 0.0f;
        temporal_restir(&r, s, w_new, frame.max_temporal_reuse_count);
      }

      // Validation frame, light.wgsl:1156-1214
      if (frame.number % validate_interval == 0u) {
        LightCandidate candidate = select_light_candidate(sc, r.s.random, xyz(r.s.visible_position), r.s.visible_normal, select_light_instance, &info);
        ray.origin = xyz(s.visible_position) + s.visible_normal * RAY_BIAS;
        ray.direction = normalize(xyz(r.s.sample_position) - xyz(s.visible_position));
        ray.inv_direction = 1.0f / ray.direction;

        v4 validate_radiance = V4(0, 0, 0, 0);
        bool trace_condition = dot(candidate.direction, r.s.visible_normal) > 0.0f;
        trace_condition = trace_condition && candidate.p > 0.0f;
        if (emissive_lit) trace_condition = trace_condition && candidate.emissive_instance != DONT_SAMPLE_EMISSIVE;

        if (trace_condition) {
          hit = traverse_top(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance);
          occlude_hit_info(ray, hit, &info);
          if (emissive_lit)
            validate_radiance = input_radiance(sc, ray, info, false, candidate.emissive_instance, false);
          else
            validate_radiance = input_radiance(sc, ray, info, true, DONT_SAMPLE_EMISSIVE, false);
        }

        if (r.count >= (float)DIRECT_VALIDATION_FRAME_SAMPLE_THRESHOLD) {
          s.random = r.s.random;
          s.sample_position = info.position;
          s.sample_normal = info.normal;
          s.radiance = validate_radiance;
        }

        float luminance_ratio = luminance(xyz(validate_radiance)) / fmax_(luminance(xyz(r.s.radiance)), 0.0001f);
        if (luminance_ratio > 1.25f || luminance_ratio < 0.8f) {
          if (prev_on_screen) scatter[y].push_back({index, previous_index, pack_reservoir(r)});
          float w_new = (candidate.p > 0.0f) ? luminance(xyz(s.radiance)) / candidate.p : 0.0f;
          set_reservoir(&r, s, w_new);
        }
      }

      float total_lum = r.count * luminance(xyz(r.s.radiance));
      r.w = (total_lum > 0.0f) ? r.w_sum / total_lum : 0.0f;

      r.s.visible_position = s.visible_position;
      r.s.visible_normal = s.visible_normal;
      r.lifetime += 1.0f;

      float variance = r.w2_sum / r.count - pow2_(r.w_sum / r.count);
      variance = (r.count < 1.0f) ? variance : variance / r.count;
      variance = fmin_(variance, MAX_VARIANCE);
      variance_texture.store_f32(x, y, variance);

      if (frame.temporal_reuse > 0u) rs.current[index] = pack_reservoir(r);

      Surface surface = retreive_surface(sc, im_y, V2(velocity_uv.z, velocity_uv.w));
      v3 view_direction = calculate_view(sc, position, c->view.projection[15] == 1.0f);
      v3 out_radiance = shading(sc, view_direction, r.s.visible_normal,
                                normalize(xyz(r.s.sample_position) - xyz(r.s.visible_position)), surface, r.s.radiance);
      out_radiance = out_radiance * r.w;
      v3 out_color = emissive_lit ? out_radiance : out_radiance + compute_emissive_radiance(surface.emissive);  // RENDER_EMISSIVE, light.wgsl:1237-1259
      render_texture.store_f16x4(x, y, V4(out_color, 1.0f));
    }
    n_tlas += sc.n_tlas;
    n_blas += sc.n_blas;
  }
  if (parks_across_bands(c)) park_scatter(c, channel, scatter, rs.previous_spatial, own_written, y0, y1);
  else apply_scatter(scatter, rs.previous_spatial, own_written);
  c->stats.rays_tlas += n_tlas;
  c->stats.rays_blas += n_blas;
}

// ------------------------------------------------------------------------------------------
// indirect_lit_ambient, light.wgsl:1263-1498.  MULTIPLE_BOUNCES iff indirect_bounces >= 2
// (light.rs:663-666).
// ------------------------------------------------------------------------------------------
static void pass_indirect(Ctx* c, int y0, int y1) {
  const int channel = 2;
  const bool multiple_bounces = c->frame.indirect_bounces >= 2u;
  Tex position_texture = tex(c, HK_BUF_POSITION), normal_texture = tex(c, HK_BUF_NORMAL),
      instance_material_texture = tex(c, HK_BUF_INSTANCE_MATERIAL), velocity_uv_texture = tex(c, HK_BUF_VELOCITY_UV),
      variance_texture = tex(c, HK_BUF_VARIANCE0 + channel), render_texture = tex(c, HK_BUF_RENDER0 + channel);
  ReservoirSet rs = reservoir_set(c, channel);
  Sizes sz{c->W, c->H, c->RW, c->RH};
  const int rw = c->RW;
  std::vector<std::vector<ScatterStore>> scatter(c->RH);
  std::vector<uint8_t> own_written((size_t)c->RW * c->RH, 0);
  uint64_t n_tlas = 0, n_blas = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : n_tlas, n_blas)
  for (int y = y0; y < y1; ++y) {
    Scene sc = make_scene(c);
    const HkFrame& frame = c->frame;
    for (int x = 0; x < rw; ++x) {
      const int index = x + rw * y;
      if (c->step_rec) {  // tests/tools/divergence_model.py
        sc.step_rec = c->step_rec + (size_t)3 * c->step_slots * index;
        sc.step_slots = c->step_slots;
        sc.step_slot = 0;
        sc.cur_nodes = sc.cur_tris = sc.cur_entries = 0;
        sc.step_base = 0;
      }
      v2 uv = coords_to_uv(x, y, sz.rw, sz.rh);
      int dcx, dcy;
      jittered_deferred_coords(sc, sz, uv, &dcx, &dcy);
      v4 position_depth = position_texture.load_f32x4(dcx, dcy);
      v4 position = V4(xyz(position_depth), 1.0f);
      float depth = position_depth.w;

      Sample s{};
      Reservoir r{};

      if (frame.indirect_bounces == 0u || depth < F32_EPSILON) {  // light.wgsl:1279-1287
        PackedReservoir pr = pack_reservoir(r);
        rs.current[index] = pr;
        rs.spatial[index] = pr;
        rs.previous_spatial[index] = pr;
        own_written[index] = 1;
        variance_texture.store_f32(x, y, 0.0f);
        render_texture.store_f16x4(x, y, V4(0, 0, 0, 0));
        continue;
      }

      v3 normal = normalize(xyz(normal_texture.load_snorm8x4(dcx, dcy)));
      v2 imf = instance_material_texture.load_f32x2(dcx, dcy);
      uint32_t im_x = f32_to_u32(imf.x), im_y = f32_to_u32(imf.y);
      v4 velocity_uv = velocity_uv_texture.load_f32x4(dcx, dcy);

      s.random = noise_fetch(c, x, y, frame.number);
      s.random = fract(s.random + (float)frame.number * GOLDEN_RATIO);
      s.visible_position = V4(xyz(position), depth);
      s.visible_normal = normal;
      s.visible_instance = im_x;

      Ray ray{};
      Hit hit{};
      HitInfo info{};
      float pdf = 0.0f;
      Surface surface{};

      if (multiple_bounces) {  // light.wgsl:1309-1394
        Sample bounce_sample = s;
        v3 color_transport = V3(1.0f, 1.0f, 1.0f);
        for (uint32_t n = 0u; n < frame.indirect_bounces && (color_transport.x > 0.01f || color_transport.y > 0.01f || color_transport.z > 0.01f); n += 1u) {
          sc.step_base = 3u * n;
          v4 rand_sample = sample_cosine_hemisphere(V2(bounce_sample.random.x, bounce_sample.random.y));
          ray.origin = xyz(bounce_sample.visible_position) + bounce_sample.visible_normal * RAY_BIAS;
          ray.direction = mul(normal_basis(bounce_sample.visible_normal), xyz(rand_sample));
          ray.inv_direction = 1.0f / ray.direction;

          sc.step_slot = sc.step_base;
          hit = traverse_top(sc, ray, F32_MAX, 0.0f, DONT_EXCLUDE);
          info = hit_info(sc, ray, hit);

          if (n == 0u) {
            s.sample_position = info.position;
            s.sample_normal = info.normal;
            pdf = rand_sample.w;
          }
          bounce_sample.sample_position = info.position;
          bounce_sample.sample_normal = info.normal;

          if (hit.instance_index != U32_MAX) {
            v3 out_radiance = V3(0, 0, 0);
            surface = retreive_surface(sc, info.material_index, info.uv);
            surface.roughness = 1.0f;

            const uint32_t info_instance = info.instance_index;  // read before the call overwrites *info
            LightCandidate candidate = select_light_candidate(sc, bounce_sample.random, xyz(bounce_sample.sample_position),
                                                              bounce_sample.sample_normal, info_instance, &info);
            bool sample_directional = (candidate.emissive_instance == DONT_SAMPLE_EMISSIVE);
            v3 bounce_view_direction = normalize(xyz(bounce_sample.visible_position) - xyz(bounce_sample.sample_position));

            if (dot(candidate.direction, bounce_sample.sample_normal) > 0.0f && candidate.p > 0.0f) {
              ray.origin = xyz(bounce_sample.sample_position) + bounce_sample.sample_normal * RAY_BIAS;
              ray.direction = candidate.direction;
              ray.inv_direction = 1.0f / ray.direction;

              sc.step_slot = sc.step_base + 2u;
              hit = traverse_top(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance);
              occlude_hit_info(ray, hit, &info);

              v4 in_radiance = input_radiance(sc, ray, info, sample_directional, candidate.emissive_instance, false);
              out_radiance = shading(sc, bounce_view_direction, bounce_sample.sample_normal, ray.direction, surface, in_radiance);
              out_radiance = out_radiance / candidate.p;
              if (n > 0u) out_radiance = (rand_sample.w < 0.01f) ? V3(0, 0, 0) : out_radiance / rand_sample.w;

              float out_luminance = luminance(out_radiance);
              if (out_luminance > frame.max_indirect_luminance) out_radiance = out_radiance * frame.max_indirect_luminance / out_luminance;

              s.radiance = s.radiance + V4(color_transport * out_radiance, 1.0f);
            }
            color_transport = color_transport * env_brdf(bounce_view_direction, bounce_sample.sample_normal, surface);
            bounce_sample.random = fract(bounce_sample.random + (float)frame.number * GOLDEN_RATIO);
            bounce_sample.visible_position = bounce_sample.sample_position;
            bounce_sample.visible_normal = bounce_sample.sample_normal;
          } else {
            v3 out_radiance = xyz(input_radiance(sc, ray, info, false, DONT_SAMPLE_EMISSIVE, true));
            s.radiance = s.radiance + V4(color_transport * out_radiance, 0.0f);
            break;
          }
        }
      } else {  // light.wgsl:1395-1450
        v4 rand_sample = sample_cosine_hemisphere(V2(s.random.x, s.random.y));
        ray.origin = xyz(s.visible_position) + s.visible_normal * RAY_BIAS;
        ray.direction = mul(normal_basis(s.visible_normal), xyz(rand_sample));
        ray.inv_direction = 1.0f / ray.direction;

        sc.step_slot = sc.step_base;
          hit = traverse_top(sc, ray, F32_MAX, 0.0f, DONT_EXCLUDE);
        info = hit_info(sc, ray, hit);

        s.sample_position = info.position;
        s.sample_normal = info.normal;
        pdf = rand_sample.w;

        if (hit.instance_index != U32_MAX) {
          v3 out_radiance = V3(0, 0, 0);
          surface = retreive_surface(sc, info.material_index, info.uv);
          surface.roughness = 1.0f;
          const uint32_t info_instance = info.instance_index;
          LightCandidate candidate = select_light_candidate(sc, s.random, xyz(s.sample_position), s.sample_normal, info_instance, &info);
          bool sample_directional = (candidate.emissive_instance == DONT_SAMPLE_EMISSIVE);
          if (dot(candidate.direction, s.sample_normal) > 0.0f && candidate.p > 0.0f) {
            ray.origin = xyz(s.sample_position) + s.sample_normal * RAY_BIAS;
            ray.direction = candidate.direction;
            ray.inv_direction = 1.0f / ray.direction;
            sc.step_slot = sc.step_base + 2u;
              hit = traverse_top(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance);
            occlude_hit_info(ray, hit, &info);
            v4 in_radiance = input_radiance(sc, ray, info, sample_directional, candidate.emissive_instance, false);
            out_radiance = shading(sc, normalize(xyz(s.visible_position) - xyz(s.sample_position)), s.sample_normal, ray.direction, surface, in_radiance);
            out_radiance = out_radiance / candidate.p;
            s.radiance = s.radiance + V4(out_radiance, 1.0f);
          }
        } else {
          v3 out_radiance = xyz(input_radiance(sc, ray, info, false, DONT_SAMPLE_EMISSIVE, true));
          s.radiance = s.radiance + V4(out_radiance, 0.0f);
        }
      }

      // ReSTIR: Temporal, light.wgsl:1452-1497
      v2 previous_uv = jittered_deferred_uv(sc, sz, uv, 0.25f) - V2(velocity_uv.x, velocity_uv.y);
      r = load_reservoir_uv(rs.previous, previous_uv, sz.rw, sz.rh);
      if (!check_previous_reservoir(&r, s) && fabsf(previous_uv.x - 0.5f) <= 0.5f && fabsf(previous_uv.y - 0.5f) <= 0.5f) {
        int previous_index = f32_to_i32(previous_uv.x * (float)sz.rw) + rw * f32_to_i32(previous_uv.y * (float)sz.rh);
        scatter[y].push_back({index, previous_index, pack_reservoir(r)});
      }

      surface = retreive_surface(sc, im_y, V2(velocity_uv.z, velocity_uv.w));
      v3 view_direction = calculate_view(sc, position, c->view.projection[15] == 1.0f);
      v3 sample_radiance = shading(sc, view_direction, s.visible_normal, normalize(xyz(s.sample_position) - xyz(s.visible_position)), surface, s.radiance);
      float w_new = (pdf > 0.0f) ? luminance(sample_radiance) / pdf : 0.0f;
      temporal_restir(&r, s, w_new, frame.max_temporal_reuse_count);

      v3 out_radiance = shading(sc, view_direction, r.s.visible_normal, normalize(xyz(r.s.sample_position) - xyz(r.s.visible_position)), surface, r.s.radiance);
      float total_lum = r.count * luminance(out_radiance);
      r.w = (total_lum > 0.0f) ? r.w_sum / total_lum : 0.0f;

      r.s.visible_position = s.visible_position;
      r.s.visible_normal = s.visible_normal;
      r.lifetime += 1.0f;

      float variance = r.w2_sum / r.count - pow2_(r.w_sum / r.count);
      variance = (r.count < 1.0f) ? variance : variance / r.count;
      variance = fmin_(variance, MAX_VARIANCE);
      variance_texture.store_f32(x, y, variance);

      if (frame.temporal_reuse > 0u) rs.current[index] = pack_reservoir(r);
      render_texture.store_f16x4(x, y, V4(out_radiance * r.w, 1.0f));
    }
    n_tlas += sc.n_tlas;
    n_blas += sc.n_blas;
  }
  if (parks_across_bands(c)) park_scatter(c, channel, scatter, rs.previous_spatial, own_written, y0, y1);
  else apply_scatter(scatter, rs.previous_spatial, own_written);
  c->stats.rays_tlas += n_tlas;
  c->stats.rays_blas += n_blas;
}

// ------------------------------------------------------------------------------------------
// spatial_reuse, light.wgsl:1500-1684.  The 8x8 workgroup cache (light.wgsl:1500-1501,1522-1524,
// 1584-1591) holds exactly what load_reservoir / the depth load return, so every neighbour is
// read from the buffers here.
// ------------------------------------------------------------------------------------------
static void pass_spatial_reuse(Ctx* c, bool emissive_lit, int y0, int y1) {
  const int channel = emissive_lit ? 1 : 2;
  const uint32_t SPATIAL_REUSE_COUNT = emissive_lit ? 8u : 16u;   // light.wgsl:246-252
  const float SPATIAL_REUSE_RANGE = emissive_lit ? 10.0f : 20.0f;
  Tex position_texture = tex(c, HK_BUF_POSITION), instance_material_texture = tex(c, HK_BUF_INSTANCE_MATERIAL),
      velocity_uv_texture = tex(c, HK_BUF_VELOCITY_UV), variance_texture = tex(c, HK_BUF_VARIANCE0 + channel),
      render_texture = tex(c, HK_BUF_RENDER0 + channel);
  ReservoirSet rs = reservoir_set(c, channel);
  Sizes sz{c->W, c->H, c->RW, c->RH};
  const int rw = c->RW;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y) {
    Scene sc = make_scene(c);
    const HkFrame& frame = c->frame;
    for (int x = 0; x < rw; ++x) {
      const int index = x + rw * y;
      v2 uv = coords_to_uv(x, y, sz.rw, sz.rh);
      int dcx, dcy;
      jittered_deferred_coords(sc, sz, uv, &dcx, &dcy);
      v4 position_depth = position_texture.load_f32x4(dcx, dcy);
      v4 position = V4(xyz(position_depth), 1.0f);
      float depth = position_depth.w;

      Reservoir r = unpack_reservoir(rs.current[index]);

      if (depth < F32_EPSILON) {
        rs.spatial[index] = pack_reservoir(r);
        render_texture.store_f16x4(x, y, V4(0, 0, 0, 0));
        continue;
      }

      v2 imf = instance_material_texture.load_f32x2(dcx, dcy);
      uint32_t im_y = f32_to_u32(imf.y);
      v4 velocity_uv = velocity_uv_texture.load_f32x4(dcx, dcy);
      Surface surface = retreive_surface(sc, im_y, V2(velocity_uv.z, velocity_uv.w));

      bool use_spatial_variance = r.count <= (float)SPATIAL_VARIANCE_SAMPLE_THRESHOLD;

      v2 previous_uv = jittered_deferred_uv(sc, sz, uv, 0.25f) - V2(velocity_uv.x, velocity_uv.y);

      Reservoir q = r;
      const Sample s = q.s;

      if (r.lifetime <= reservoir_lifetime(sc)) r = load_reservoir_uv(rs.previous_spatial, previous_uv, sz.rw, sz.rh);

      v3 view_direction = calculate_view(sc, position, c->view.projection[15] == 1.0f);
      if (emissive_lit) {
        merge_reservoir(&r, q, luminance(xyz(q.s.radiance)));
      } else {
        v3 out_radiance = shading(sc, view_direction, s.visible_normal, normalize(xyz(s.sample_position) - xyz(s.visible_position)), surface, s.radiance);
        merge_reservoir(&r, q, luminance(out_radiance));
      }

      r.s.visible_position = s.visible_position;
      r.s.visible_normal = s.visible_normal;

      for (uint32_t i = 1u; i <= SPATIAL_REUSE_COUNT; i += 1u) {
        v2 polar_offset = V2(TAU * fract((float)i * GOLDEN_RATIO + dot(s.random, V4(1.0f, 1.0f, 1.0f, 1.0f)) + random_float(frame.number)),
                             sqrtf((float)i / (float)SPATIAL_REUSE_COUNT) * SPATIAL_REUSE_RANGE);
        v2 offset = polar_offset.y * V2(cos_(polar_offset.x), sin_(polar_offset.x));

        int scx = f32_to_i32(offset.x + (float)x), scy = f32_to_i32(offset.y + (float)y);
        v2 sample_uv = coords_to_uv(scx, scy, sz.rw, sz.rh);
        int sdx, sdy;
        jittered_deferred_coords(sc, sz, sample_uv, &sdx, &sdy);
        if (sample_uv.x < 0.0f || sample_uv.y < 0.0f || sample_uv.x > 1.0f || sample_uv.y > 1.0f) continue;

        float sample_depth = position_texture.load_f32x4(sdx, sdy).w;
        q = unpack_reservoir(rs.current[scx + rw * scy]);

        float depth_ratio = depth / sample_depth;
        if (depth_ratio < 0.9f || depth_ratio > 1.1f) continue;

        bool normal_miss = dot(s.visible_normal, q.s.visible_normal) < 0.866f;
        if (q.count < F32_EPSILON || normal_miss) continue;

        v3 sample_direction = normalize(xyz(q.s.sample_position) - xyz(s.visible_position));
        if (dot(sample_direction, s.visible_normal) < 0.0f) continue;

        // screen-space ray-marching of the depth, light.wgsl:1608-1628
        float tap_interval = fmax_(1.0f, polar_offset.y / (float)(SPATIAL_REUSE_TAPS + 1u));
        uint32_t tap_count = f32_to_u32(polar_offset.y / tap_interval);
        bool occluded = false;
        for (uint32_t j = 1u; j <= tap_count; j += 1u) {
          float tap_dist = (float)j * tap_interval;
          v2 tap_offset = tap_dist * normalize(offset);
          v2 tap_uv = uv + tap_offset / V2((float)sz.rw, (float)sz.rh);
          int tdx, tdy;
          jittered_deferred_coords(sc, sz, tap_uv, &tdx, &tdy);
          float tap_depth = position_texture.load_f32x4(tdx, tdy).w;
          float ref_depth = mix(depth, sample_depth, (float)j / (float)(tap_count + 1u));
          if (tap_depth > ref_depth + 0.00001f) {
            occluded = true;
            break;
          }
        }
        if (occluded) continue;

        float jacobian = (q.s.sample_position.w > 0.5f) ? compute_jacobian(q.s, s) : 1.0f;
        if (emissive_lit) {
          merge_reservoir(&r, q, luminance(xyz(q.s.radiance)) / jacobian);
        } else {
          v3 out_radiance = shading(sc, view_direction, s.visible_normal, sample_direction, surface, q.s.radiance);
          merge_reservoir(&r, q, luminance(out_radiance) / jacobian);
        }
      }

      float m = (float)frame.max_spatial_reuse_count;
      if (r.count > m) {
        r.w_sum *= m / r.count;
        r.w2_sum *= m / r.count;
        r.count = m;
      }

      v3 out_radiance = shading(sc, view_direction, s.visible_normal, normalize(xyz(r.s.sample_position) - xyz(s.visible_position)), surface, r.s.radiance);
      float total_lum = emissive_lit ? r.count * luminance(xyz(r.s.radiance)) : r.count * luminance(out_radiance);
      r.w = (total_lum > 0.0f) ? r.w_sum / total_lum : 0.0f;
      r.lifetime += 1.0f;

      rs.spatial[index] = pack_reservoir(r);

      if (use_spatial_variance) {
        float variance = r.w2_sum / r.count - pow2_(r.w_sum / r.count);
        variance = (r.count < 1.0f) ? variance : variance / r.count;
        variance = fmin_(variance, MAX_VARIANCE);
        variance_texture.store_f32(x, y, variance);
      }
      // RENDER_EMISSIVE is never defined for spatial_reuse pipelines (light.rs:433-442)
      v3 out_color = r.w * out_radiance;
      render_texture.store_f16x4(x, y, V4(out_color, 1.0f));
    }
  }
}

// ------------------------------------------------------------------------------------------
// denoise.wgsl
// ------------------------------------------------------------------------------------------
static inline float kernel_at(const HkFrame& f, int col, int row) { return f.kernel[col][row]; }  // frame.kernel[c][r], column-major mat3

static void accumulate_variance(Ctx* c, const Scene& sc, const Tex& variance_texture, v2 uv, int ox, int oy, float* sum_variance) {  // denoise.wgsl:116-133
  (void)sc;
  v2 sample_uv = uv + V2((float)ox, (float)oy) / V2((float)c->RW, (float)c->RH);
  if (sample_uv.x < 0.0f || sample_uv.y < 0.0f || sample_uv.x > 1.0f || sample_uv.y > 1.0f) return;
  int sx, sy;
  variance_texture.nearest_coords(sample_uv, &sx, &sy);
  float variance = variance_texture.load_f32(sx, sy);
  if (variance > F32_MAX) return;
  *sum_variance += kernel_at(c->frame, oy + 1, ox + 1) * fmax_(variance, 0.0f);
}
static void pass_demodulation(Ctx* c, int channel, int y0, int y1) {  // denoise.wgsl:135-162
  Tex albedo_texture = tex(c, HK_BUF_ALBEDO), variance_texture = tex(c, HK_BUF_VARIANCE0 + channel),
      render_texture = tex(c, HK_BUF_RENDER0 + channel), internal_texture_0 = tex(c, HK_BUF_DENOISE_INTERNAL0),
      internal_variance = tex(c, HK_BUF_DENOISE_INTERNAL_VARIANCE);
  Sizes sz{c->W, c->H, c->RW, c->RH};
#pragma omp parallel for schedule(static)
  for (int y = y0; y < y1; ++y) {
    Scene sc = make_scene(c);
    for (int x = 0; x < c->RW; ++x) {
      v2 uv = coords_to_uv(x, y, sz.rw, sz.rh);
      v2 deferred_uv = jittered_deferred_uv(sc, sz, uv, 0.5f);
      int ax, ay, rx, ry;
      albedo_texture.nearest_coords(deferred_uv, &ax, &ay);
      v3 albedo = xyz(albedo_texture.load_f16x4(ax, ay));
      render_texture.nearest_coords(uv, &rx, &ry);
      v3 irradiance = xyz(render_texture.load_f16x4(rx, ry));
      v3 q = irradiance / albedo;
      irradiance = V3(albedo.x < 0.01f ? 0.0f : q.x, albedo.y < 0.01f ? 0.0f : q.y, albedo.z < 0.01f ? 0.0f : q.z);
      internal_texture_0.store_f16x4(x, y, V4(irradiance, 1.0f));

      float sum_variance = 0.0f;
      static const int order[9][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 0}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};
      for (int k = 0; k < 9; ++k) accumulate_variance(c, sc, variance_texture, uv, order[k][0], order[k][1], &sum_variance);
      internal_variance.store_f32(x, y, sum_variance);
    }
  }
}
static float normal_weight(v3 n0, v3 n1) { return pow16_(fmax_(0.0f, dot(n0, n1))); }  // denoise.wgsl:44-47
static float depth_weight(float d0, float d1, v2 gradient, v2 offset) {  // denoise.wgsl:50-53
  float eps = 0.01f;
  return exp_((-fabsf(d0 - d1)) / (fabsf(dot(gradient, offset)) + eps));
}
static float luminance_weight(float l0, float l1, float variance) {  // denoise.wgsl:56-61
  float strictness = 4.0f, eps = 0.001f;  // exponent 0.25: pow_quarter_
  return exp_((-fabsf(l0 - l1)) / (strictness * pow_quarter_(variance) + eps));
}
static float instance_weight(float i0, float i1) { return fmax_(0.0f, 1.0f - fabsf(i0 - i1)); }  // denoise.wgsl:63-65

static void pass_denoise(Ctx* c, int channel, int level, int y0, int y1) {  // denoise.wgsl:164-319
  const bool firefly = channel != 0;  // denoise_direct has no FIREFLY_FILTERING, post_process.rs:773-783,1193-1197
  const int step = 8 >> level;        // denoise.wgsl:101-114
  Tex input = tex(c, HK_BUF_DENOISE_INTERNAL0 + level);
  Tex output = level == 3 ? tex(c, HK_BUF_DENOISE_RENDER0 + channel) : tex(c, HK_BUF_DENOISE_INTERNAL0 + level + 1);
  Tex position_texture = tex(c, HK_BUF_POSITION), normal_texture = tex(c, HK_BUF_NORMAL),
      depth_gradient_texture = tex(c, HK_BUF_DEPTH_GRADIENT), instance_material_texture = tex(c, HK_BUF_INSTANCE_MATERIAL),
      internal_variance = tex(c, HK_BUF_DENOISE_INTERNAL_VARIANCE), albedo_texture = tex(c, HK_BUF_ALBEDO);
  Sizes sz{c->W, c->H, c->RW, c->RH};
#pragma omp parallel for schedule(static)
  for (int y = y0; y < y1; ++y) {
    Scene sc = make_scene(c);
    for (int x = 0; x < c->RW; ++x) {
      v2 uv = coords_to_uv(x, y, sz.rw, sz.rh);
      v2 deferred_uv = jittered_deferred_uv(sc, sz, uv, 0.5f);
      int dx, dy;
      position_texture.nearest_coords(deferred_uv, &dx, &dy);
      float depth = position_texture.load_f32x4(dx, dy).w;
      v2 depth_gradient = depth_gradient_texture.load_f32x2(dx, dy);
      v3 normal = normalize(xyz(normal_texture.load_snorm8x4(dx, dy)));
      float instance = instance_material_texture.load_f32x2(dx, dy).x;

      if (depth < F32_EPSILON) {
        output.store_f16x4(x, y, V4(0, 0, 0, 0));
        continue;
      }
      float variance = internal_variance.load_f32(x, y);
      v3 irradiance = xyz(input.load_f16x4(x, y));
      v3 sum_irradiance = irradiance * kernel_at(c->frame, 1, 1);
      float sum_w = kernel_at(c->frame, 1, 1);
      if (any_is_nan_vec3(irradiance) || irradiance.x > F32_MAX || irradiance.y > F32_MAX || irradiance.z > F32_MAX) {
        irradiance = V3(0, 0, 0);
        sum_irradiance = V3(0, 0, 0);
        sum_w = 0.0f;
      }
      float lum = luminance(irradiance);
      float ff_moment_1 = 0.0f, ff_moment_2 = 0.0f, ff_count = 0.0f;

      static const int order[8][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};
      for (int k = 0; k < 8; ++k) {  // accumulate_irradiance, denoise.wgsl:164-213
        int ox = order[k][0], oy = order[k][1];
        int sx = x + ox * step, sy = y + oy * step;
        v2 sample_uv = coords_to_uv(sx, sy, sz.rw, sz.rh);
        v2 sample_deferred_uv = jittered_deferred_uv(sc, sz, sample_uv, 0.5f);
        if (sample_uv.x < 0.0f || sample_uv.y < 0.0f || sample_uv.x > 1.0f || sample_uv.y > 1.0f) continue;
        v3 irr = xyz(input.load_f16x4(sx, sy));
        if (any_is_nan_vec3(irr) || irr.x > F32_MAX || irr.y > F32_MAX || irr.z > F32_MAX) continue;
        int gx, gy;
        position_texture.nearest_coords(sample_deferred_uv, &gx, &gy);
        v3 sample_normal = normalize(xyz(normal_texture.load_snorm8x4(gx, gy)));
        float sample_depth = position_texture.load_f32x4(gx, gy).w;
        float sample_instance = instance_material_texture.load_f32x2(gx, gy).x;
        float sample_luminance = luminance(irr);

        float w_normal = normal_weight(normal, sample_normal);
        float w_depth = depth_weight(depth, sample_depth, depth_gradient, V2((float)ox, (float)oy));
        float w_instance = instance_weight(instance, sample_instance);
        float w_luminance = luminance_weight(lum, sample_luminance, variance);
        float w = clamp_(w_normal * w_depth * w_instance * w_luminance, 0.0f, 1.0f) * kernel_at(c->frame, oy + 1, ox + 1);
        sum_irradiance = sum_irradiance + irr * w;
        sum_w += w;
        if (firefly) {
          ff_moment_1 += sample_luminance;
          ff_moment_2 += sample_luminance * sample_luminance;
          ff_count += 1.0f;
        }
      }
      v3 qd = sum_irradiance / sum_w;
      irradiance = (sum_w < 0.0001f) ? V3(0, 0, 0) : qd;
      if (firefly) {
        float ff_mean = ff_moment_1 / ff_count;
        float ff_var = ff_moment_2 / ff_count - ff_mean * ff_mean;
        if (lum > ff_mean + 3.0f * sqrtf(ff_var)) irradiance = ff_mean / lum * irradiance;
      }
      v4 color = V4(irradiance, 1.0f);
      if (level == 3) {  // denoise.wgsl:314-315
        int ax, ay;
        albedo_texture.nearest_coords(deferred_uv, &ax, &ay);
        color = color * albedo_texture.load_f16x4(ax, ay);
      }
      output.store_f16x4(x, y, color);
    }
  }
}

// tone_mapping.wgsl:21-32.  Inputs follow post_process.rs:941-954: denoise_render[*] when
// settings.denoise, light render[*] otherwise; the indirect input is an all-zero fallback when
// indirect_bounces == 0.
static void pass_tone_mapping(Ctx* c, bool denoise, int y0, int y1) {
  Tex d = tex(c, (denoise ? HK_BUF_DENOISE_RENDER0 : HK_BUF_RENDER0) + 0), e = tex(c, (denoise ? HK_BUF_DENOISE_RENDER0 : HK_BUF_RENDER0) + 1),
      i = tex(c, (denoise ? HK_BUF_DENOISE_RENDER0 : HK_BUF_RENDER0) + 2), out = tex(c, HK_BUF_TONE_MAPPED);
  const bool has_indirect = c->frame.indirect_bounces != 0u;
#pragma omp parallel for schedule(static)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < c->RW; ++x) {
      v4 color = d.load_f16x4(x, y);
      color = color + e.load_f16x4(x, y);
      if (has_indirect) color = color + i.load_f16x4(x, y);
      v3 rgb = reinhard_luminance(V3(fmax_(color.x, 0.0039f), fmax_(color.y, 0.0039f), fmax_(color.z, 0.0039f)));
      color = V4(rgb, color.w);
      if (!(color.w > 0.0f)) color = P4(c->frame.clear_color);
      out.store_f16x4(x, y, color);
    }
}

// ------------------------------------------------------------------------------------------
// Temporal anti-aliasing and SMAA Tu4x upscaling: taa.wgsl, smaa.wgsl.
// Run after tone mapping by PostProcessNode::run (post_process.rs:1236-1272); bindings
// post_process.rs:983-1035: SMAA reads tone_mapping_output[previous / current] and writes
// upscale_output[0]; TAA reads taa_output[previous] and upscale_output[0] (SMAA) or
// tone_mapping_output[current] (FSR1) and writes taa_output[current].
// ------------------------------------------------------------------------------------------
static inline v3 rgb(v4 a) { return V3(a.x, a.y, a.z); }
static inline v3 clamp01(v3 c) { return V3(clamp_(c.x, 0.0f, 1.0f), clamp_(c.y, 0.0f, 1.0f), clamp_(c.z, 0.0f, 1.0f)); }
static inline v3 sqrt3(v3 a) { return V3(sqrtf(a.x), sqrtf(a.y), sqrtf(a.z)); }
static inline v3 abs3(v3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline v3 RGB_to_YCoCg(v3 c) {  // taa.wgsl:20-25, smaa.wgsl:23-28
  float y = (c.x / 4.0f) + (c.y / 2.0f) + (c.z / 4.0f);
  float co = (c.x / 2.0f) - (c.z / 2.0f);
  float cg = (-c.x / 4.0f) + (c.y / 2.0f) - (c.z / 4.0f);
  return V3(y, co, cg);
}
static inline v3 YCoCg_to_RGB(v3 y) {  // taa.wgsl:27-32
  float r = y.x + y.y - y.z;
  float g = y.x + y.z;
  float b = y.x - y.y - y.z;
  return clamp01(V3(r, g, b));
}
static inline v3 clip_towards_aabb_center(v3 previous_color, v3 aabb_min, v3 aabb_max) {  // taa.wgsl:34-42 (current_color is unused there)
  v3 p_clip = 0.5f * (aabb_max + aabb_min);
  v3 e_clip = 0.5f * (aabb_max - aabb_min);
  v3 v_clip = previous_color - p_clip;
  v3 v_unit = v_clip / e_clip;
  v3 a_unit = abs3(v_unit);
  float ma_unit = fmax_(a_unit.x, fmax_(a_unit.y, a_unit.z));
  return (ma_unit > 1.0f) ? p_clip + v_clip / ma_unit : previous_color;
}
static inline float distance2(v2 a, v2 b) { return length(a - b); }
static inline float distance3(v3 a, v3 b) { return length(a - b); }
static inline bool any_lt(v4 a, float s) { return a.x < s || a.y < s || a.z < s || a.w < s; }
static inline bool any_gt(v4 a, float s) { return a.x > s || a.y > s || a.z > s || a.w > s; }
// select(current / previous, 1.0, previous == 0.0), taa.wgsl:111, smaa.wgsl:146
static inline v4 depth_ratio4(float current, v4 previous) {
  auto r = [&](float p) { return p == 0.0f ? 1.0f : current / p; };
  return V4(r(previous.x), r(previous.y), r(previous.z), r(previous.w));
}
// taa.wgsl:54-73, smaa.wgsl:54-73: velocity of the nearest (largest reverse-Z depth) of the 4 diagonal neighbours
static v2 nearest_velocity(const Tex& position, const Tex& velocity_uv, v2 uv, v2 texel_size) {
  v4 depths;
  depths.x = position.sample_nearest(uv + V2(texel_size.x, texel_size.y)).w;
  depths.y = position.sample_nearest(uv + V2(-texel_size.x, texel_size.y)).w;
  depths.z = position.sample_nearest(uv + V2(texel_size.x, -texel_size.y)).w;
  depths.w = position.sample_nearest(uv + V2(-texel_size.x, -texel_size.y)).w;
  float max_depth = fmax_(fmax_(depths.x, depths.y), fmax_(depths.z, depths.w));
  float depth = position.sample_nearest(uv).w;
  v2 offset = V2(0.0f, 0.0f);
  if (depth < max_depth) {
    v4 eq = V4(depths.x == max_depth ? 1.0f : 0.0f, depths.y == max_depth ? 1.0f : 0.0f, depths.z == max_depth ? 1.0f : 0.0f,
               depths.w == max_depth ? 1.0f : 0.0f);
    float x = dot(V4(texel_size.x, texel_size.x, texel_size.x, texel_size.x), V4(eq.x, -eq.y, eq.z, -eq.w));
    float y = dot(V4(texel_size.y, texel_size.y, texel_size.y, texel_size.y), V4(eq.x, eq.y, -eq.z, -eq.w));
    offset = V2(x, y);
  }
  v4 v = velocity_uv.sample_nearest(uv + offset);
  return V2(v.x, v.y);
}

static void pass_taa_jasmine(Ctx* c, int y0, int y1) {  // taa.wgsl:75-170
  Tex output = tex(c, HK_BUF_TAA_OUTPUT), previous_render = tex(c, HK_BUF_PREVIOUS_TAA_OUTPUT);
  Tex render = tex(c, c->upscale_kind == HK_UPSCALE_SMAA_TU4X ? HK_BUF_UPSCALE_OUTPUT : HK_BUF_TONE_MAPPED);  // post_process.rs:1010-1013
  Tex position = tex(c, HK_BUF_POSITION), velocity_uv = tex(c, HK_BUF_VELOCITY_UV);
  Tex previous_position = tex(c, HK_BUF_PREVIOUS_POSITION), previous_velocity_uv = tex(c, HK_BUF_PREVIOUS_VELOCITY_UV);
  const v2 size = V2((float)output.w, (float)output.h);
  const v2 texel_size = V2(1.0f / size.x, 1.0f / size.y);
  const v2 render_texel = V2(1.0f / (float)render.w, 1.0f / (float)render.h);
  const float blend = 0.1f / c->frame.upscale_ratio;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < output.w; ++x) {
      v2 uv = V2(((float)x + 0.5f) / size.x, ((float)y + 0.5f) / size.y);
      v4 original_color = render.sample_nearest(uv);
      v3 current_color = rgb(original_color);
      v2 velocity = nearest_velocity(position, velocity_uv, uv, render_texel);
      v2 previous_uv = uv - velocity;
      bool boundary_miss = fabsf(previous_uv.x - 0.5f) > 0.5f || fabsf(previous_uv.y - 0.5f) > 0.5f;
      v2 uv_biases[5] = {V2(0.0f, 0.0f), V2(1.5f, 1.5f) * texel_size, V2(-1.5f, 1.5f) * texel_size, V2(1.5f, -1.5f) * texel_size,
                         V2(-1.5f, -1.5f) * texel_size};
      v4 current_position_depth = position.sample_nearest(uv);
      bool has_content = current_position_depth.w > 0.0f;
      bool depth_miss = current_position_depth.w == 0.0f;
      bool position_miss = current_position_depth.w == 0.0f;
      for (int i = 0; i < 5; ++i) {
        v4 previous_depths = previous_position.gather(3, previous_uv + uv_biases[i]);
        v4 depth_ratio = depth_ratio4(current_position_depth.w, previous_depths);
        has_content = has_content || any_gt(previous_depths, 0.0f);
        depth_miss = depth_miss || any_lt(depth_ratio, 0.95f);
        v3 pp = xyz(previous_position.sample_nearest(previous_uv + uv_biases[i]));
        position_miss = position_miss || distance3(xyz(current_position_depth), pp) > 0.5f;
      }
      if (!has_content) {
        output.store_f16x4(x, y, P4(c->frame.clear_color));
        continue;
      }
      v4 pv = previous_velocity_uv.sample_nearest(previous_uv);
      bool velocity_miss = distance2(velocity, V2(pv.x, pv.y)) > 0.00005f;
      // 5-tap Catmull-Rom reprojection, taa.wgsl:127-144
      v2 sample_position = (uv - velocity) * size;
      v2 texel_position_1 = V2(floorf(sample_position.x - 0.5f) + 0.5f, floorf(sample_position.y - 0.5f) + 0.5f);
      v2 f = sample_position - texel_position_1;
      auto poly = [](float fx, float a, float b, float cc) { return a + fx * (b + cc * fx); };  // a + f*(b + c*f)
      v2 w0 = V2(f.x * poly(f.x, -0.5f, 1.0f, -0.5f), f.y * poly(f.y, -0.5f, 1.0f, -0.5f));
      v2 w1 = V2(1.0f + f.x * f.x * (-2.5f + 1.5f * f.x), 1.0f + f.y * f.y * (-2.5f + 1.5f * f.y));
      v2 w2 = V2(f.x * poly(f.x, 0.5f, 2.0f, -1.5f), f.y * poly(f.y, 0.5f, 2.0f, -1.5f));
      v2 w3 = V2(f.x * f.x * (-0.5f + 0.5f * f.x), f.y * f.y * (-0.5f + 0.5f * f.y));
      v2 w12 = w1 + w2;
      v2 offset12 = w2 / (w1 + w2);
      v2 tp0 = (texel_position_1 - 1.0f) * texel_size;
      v2 tp3 = (texel_position_1 + 2.0f) * texel_size;
      v2 tp12 = (texel_position_1 + offset12) * texel_size;
      auto prev = [&](float u, float v) { return clamp01(rgb(previous_render.sample_linear(V2(u, v)))); };  // taa.wgsl:44-47
      v3 previous_color = V3(0.0f, 0.0f, 0.0f);
      previous_color = previous_color + prev(tp12.x, tp0.y) * w12.x * w0.y;
      previous_color = previous_color + prev(tp0.x, tp12.y) * w0.x * w12.y;
      previous_color = previous_color + prev(tp12.x, tp12.y) * w12.x * w12.y;
      previous_color = previous_color + prev(tp3.x, tp12.y) * w3.x * w12.y;
      previous_color = previous_color + prev(tp12.x, tp3.y) * w12.x * w3.y;
      if (boundary_miss || (position_miss && velocity_miss && depth_miss)) {  // 3x3 YCoCg variance clipping, taa.wgsl:146-164
        auto smp = [&](v2 p) { return RGB_to_YCoCg(clamp01(rgb(render.sample_nearest(p)))); };  // taa.wgsl:49-52
        v3 s_tl = smp(uv + V2(-texel_size.x, texel_size.y));
        v3 s_tm = smp(uv + V2(0.0f, texel_size.y));
        v3 s_tr = smp(uv + texel_size);
        v3 s_ml = smp(uv - V2(texel_size.x, 0.0f));
        v3 s_mm = RGB_to_YCoCg(current_color);
        v3 s_mr = smp(uv + V2(texel_size.x, 0.0f));
        v3 s_bl = smp(uv - texel_size);
        v3 s_bm = smp(uv - V2(0.0f, texel_size.y));
        v3 s_br = smp(uv + V2(texel_size.x, -texel_size.y));
        v3 moment_1 = s_tl + s_tm + s_tr + s_ml + s_mm + s_mr + s_bl + s_bm + s_br;
        v3 moment_2 = (s_tl * s_tl) + (s_tm * s_tm) + (s_tr * s_tr) + (s_ml * s_ml) + (s_mm * s_mm) + (s_mr * s_mr) + (s_bl * s_bl) + (s_bm * s_bm) +
                      (s_br * s_br);
        v3 mean = moment_1 / 9.0f;
        v3 variance = sqrt3((moment_2 / 9.0f) - (mean * mean));
        previous_color = RGB_to_YCoCg(previous_color);
        previous_color = clip_towards_aabb_center(previous_color, mean - variance, mean + variance);
        previous_color = YCoCg_to_RGB(previous_color);
      }
      v3 out = mix(previous_color, current_color, blend);  // taa.wgsl:167
      output.store_f16x4(x, y, V4(out, original_color.w));
    }
}

static void pass_smaa_tu4x(Ctx* c, int y0, int y1) {  // smaa.wgsl:81-188; one thread per render pixel = one 2x2 output quad
  Tex output = tex(c, HK_BUF_UPSCALE_OUTPUT), render = tex(c, HK_BUF_TONE_MAPPED), previous_render = tex(c, HK_BUF_PREVIOUS_TONE_MAPPED);
  Tex position = tex(c, HK_BUF_POSITION), velocity_uv = tex(c, HK_BUF_VELOCITY_UV), instance_material = tex(c, HK_BUF_INSTANCE_MATERIAL);
  Tex previous_position = tex(c, HK_BUF_PREVIOUS_POSITION), previous_velocity_uv = tex(c, HK_BUF_PREVIOUS_VELOCITY_UV);
  const v2 input_size = V2((float)render.w, (float)render.h), output_size = V2((float)output.w, (float)output.h);
  const v2 texel_size = V2(1.0f / output_size.x, 1.0f / output_size.y);
  const v2 deferred_texel = V2(1.0f / (float)position.w, 1.0f / (float)position.h);  // smaa.wgsl:55
  const int current_jitter = (c->frame.number & 1u) == 0u ? 0 : 1;   // smaa.wgsl:75-77
  const int previous_jitter = (c->frame.number & 1u) == 0u ? 1 : 0;  // smaa.wgsl:79-81
  const float TAU = 6.283185307f;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < render.w; ++x) {
      v2 uv = V2(((float)x + 0.5f) / input_size.x, ((float)y + 0.5f) / input_size.y);
      v2 uv_biases[5] = {V2(0.0f, 0.0f), V2(2.5f, 2.5f) * texel_size, V2(-2.5f, 2.5f) * texel_size, V2(2.5f, -2.5f) * texel_size,
                         V2(-2.5f, -2.5f) * texel_size};
      const int cox = 2 * x + current_jitter, coy = 2 * y + current_jitter;
      v3 current_color = rgb(render.sample_nearest(uv));
      const int pox = 2 * x + previous_jitter, poy = 2 * y + previous_jitter;
      v2 previous_output_uv = V2(((float)pox + 0.5f) / output_size.x, ((float)poy + 0.5f) / output_size.y);
      v2 velocity = nearest_velocity(position, velocity_uv, previous_output_uv, deferred_texel);
      v2 previous_reprojected_uv = previous_output_uv - velocity;
      v3 previous_color = rgb(previous_render.sample_nearest(previous_reprojected_uv));
      bool boundary_miss = fabsf(previous_reprojected_uv.x - 0.5f) > 0.5f || fabsf(previous_reprojected_uv.y - 0.5f) > 0.5f;
      float current_instance = instance_material.sample_nearest_f32x2(previous_output_uv).x;
      bool instance_miss = false;
      float current_depth = position.sample_nearest(previous_output_uv).w;
      bool depth_miss = current_depth == 0.0f;
      for (int i = 0; i < 5; ++i) {
        v4 previous_depths = previous_position.gather(3, previous_reprojected_uv + uv_biases[i]);
        v4 depth_ratio = depth_ratio4(current_depth, previous_depths);
        bool any_ratio = any_lt(depth_ratio, 0.95f);
        depth_miss = depth_miss || any_ratio;
        // the reference samples the CURRENT instance texture here (smaa.wgsl:149)
        float previous_instance = instance_material.sample_nearest_f32x2(previous_reprojected_uv + uv_biases[i]).x;
        instance_miss = instance_miss || (any_ratio && fabsf(previous_instance - current_instance) > 1.0f);
      }
      v4 pv = previous_velocity_uv.sample_nearest(previous_reprojected_uv);
      bool velocity_miss = distance2(velocity, V2(pv.x, pv.y)) > 0.0001f;
      if (boundary_miss || ((depth_miss || instance_miss) && velocity_miss)) {  // 2x2 YCoCg variance clipping, smaa.wgsl:156-184
        v2 uv_bias = V2(0.0f, 0.0f);
        float min_ds = 10.0f;
        for (int i = 0; i < 5; ++i) {
          v4 ds = position.gather(3, previous_output_uv + uv_biases[i]);
          v4 d = V4(current_depth - ds.x, current_depth - ds.y, current_depth - ds.z, current_depth - ds.w);
          float dds = sqrtf(dot(d, d));
          if (dds < min_ds) uv_bias = uv_biases[i];
          min_ds = fmin_(min_ds, dds);
        }
        v4 cr = render.gather(0, previous_output_uv + uv_bias);
        v4 cg = render.gather(1, previous_output_uv + uv_bias);
        v4 cb = render.gather(2, previous_output_uv + uv_bias);
        v3 s1 = RGB_to_YCoCg(V3(cr.x, cg.x, cb.x));
        v3 s2 = RGB_to_YCoCg(V3(cr.y, cg.y, cb.y));
        v3 s3 = RGB_to_YCoCg(V3(cr.z, cg.z, cb.z));
        v3 s4 = RGB_to_YCoCg(V3(cr.w, cg.w, cb.w));
        v3 moment_1 = s1 + s2 + s3 + s4;
        v3 moment_2 = s1 * s1 + s2 * s2 + s3 * s3 + s4 * s4;
        v3 mean = moment_1 / 4.0f;
        v3 variance = sqrt3((moment_2 / 4.0f) - (mean * mean));
        previous_color = RGB_to_YCoCg(previous_color);
        previous_color = clip_towards_aabb_center(previous_color, mean - variance, mean + variance);
        previous_color = YCoCg_to_RGB(previous_color);
      }
      // sub-pixel velocity blend, smaa.wgsl:186-193
      v2 sv = V2(fract(velocity.x / (2.0f * texel_size.x)), fract(velocity.y / (2.0f * texel_size.y)));
      float blend_factor = fmax_(sv.x, sv.y);
      blend_factor = clamp_(-cos_(blend_factor * TAU), 0.0f, 1.0f);
      v3 remix_color = rgb(render.sample_linear(previous_output_uv));
      previous_color = mix(previous_color, remix_color, blend_factor);
      output.store_f16x4(cox, coy, V4(current_color, 1.0f));
      output.store_f16x4(pox, poy, V4(previous_color, 1.0f));
    }
}

// ------------------------------------------------------------------------------------------
// FidelityFX Super Resolution 1.0 (Upscale::Fsr1): EASU + RCAS, the 32-bit "slow fallback" path the reference
// compiles into fsr_pass_easu.spv / fsr_pass_rcas.spv.  Restated from the GLSL that ships inside
// src/shaders/fsr/source.zip: FSR_Pass.glsl (entry, constants computed per invocation, hdr = 0 from
// post_process.rs:522-533), ffx_fsr1.h (FsrEasuCon :156-203, FsrEasuTapF :239-272, FsrEasuSetF :275-312,
// FsrEasuF :315-441, FsrRcasCon :662-673, FsrRcasF :684-768), ffx_a.h (APrxLoRcpF1 / APrxMedRcpF1 / APrxLoRsqF1
// :1843-1845), texture_gather.glsl (fakeTextureGather: four bilinear samples half an input texel off a texel
// corner, i.e. exactly the four texels around it - the contract takes the texels).  Dispatch post_process.rs:1277-1308.
// ------------------------------------------------------------------------------------------
static inline float fsr_prx_lo_rcp(float a) { return u2f(0x7ef07ebbu - f2u(a)); }
static inline float fsr_prx_med_rcp(float a) { float b = u2f(0x7ef19fffu - f2u(a)); return b * (-b * a + 2.0f); }
static inline float fsr_prx_lo_rsq(float a) { return u2f(0x5f347d74u - (f2u(a) >> 1)); }
static inline float fsr_min3(float x, float y, float z) { return fmin_(x, fmin_(y, z)); }
static inline float fsr_max3(float x, float y, float z) { return fmax_(x, fmax_(y, z)); }
struct FsrEasuConst { float con0[4], con1[4], con2[4], con3[4]; };
static FsrEasuConst fsr_easu_con(float ivx, float ivy, float isx, float isy, float osx, float osy) {  // ffx_fsr1.h:156-203
  FsrEasuConst k;
  k.con0[0] = ivx * (1.0f / osx);
  k.con0[1] = ivy * (1.0f / osy);
  k.con0[2] = 0.5f * ivx * (1.0f / osx) - 0.5f;
  k.con0[3] = 0.5f * ivy * (1.0f / osy) - 0.5f;
  k.con1[0] = 1.0f / isx;
  k.con1[1] = 1.0f / isy;
  k.con1[2] = 1.0f * (1.0f / isx);
  k.con1[3] = -1.0f * (1.0f / isy);
  k.con2[0] = -1.0f * (1.0f / isx);
  k.con2[1] = 2.0f * (1.0f / isy);
  k.con2[2] = 1.0f * (1.0f / isx);
  k.con2[3] = 2.0f * (1.0f / isy);
  k.con3[0] = 0.0f * (1.0f / isx);
  k.con3[1] = 4.0f * (1.0f / isy);
  k.con3[2] = k.con3[3] = 0.0f;
  return k;
}
// texture_gather.glsl: (x, y, z, w) = texels at (u-,v+), (u+,v+), (u+,v-), (u-,v-) of the corner p, clamp-to-edge
struct FsrGather { v4 r, g, b; };
static FsrGather fsr_gather(const Tex& t, v2 p) {
  v2 ps = V2((1.0f / (float)t.w) / 2.0f, (1.0f / (float)t.h) / 2.0f);
  v4 s3 = t.sample_nearest(V2(p.x + ps.x, p.y + ps.y));
  v4 s1 = t.sample_nearest(V2(p.x - ps.x, p.y - ps.y));
  v4 s2 = t.sample_nearest(V2(p.x + ps.x, p.y + (-ps.y)));
  v4 s4 = t.sample_nearest(V2(p.x + (-ps.x), p.y + ps.y));
  FsrGather o;
  o.r = V4(s4.x, s3.x, s2.x, s1.x);
  o.g = V4(s4.y, s3.y, s2.y, s1.y);
  o.b = V4(s4.z, s3.z, s2.z, s1.z);
  return o;
}
static inline void fsr_easu_tap(v3* aC, float* aW, v2 off, v2 dir, v2 len, float lob, float clp, v3 c) {  // ffx_fsr1.h:239-272
  v2 v;
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
  float w = wB * wA;
  *aC = *aC + c * w;
  *aW += w;
}
static inline void fsr_easu_set(v2* dir, float* len, v2 pp, int corner, float lA, float lB, float lC, float lD, float lE) {  // ffx_fsr1.h:275-312
  float w = 0.0f;
  if (corner == 0) w = (1.0f - pp.x) * (1.0f - pp.y);
  if (corner == 1) w = pp.x * (1.0f - pp.y);
  if (corner == 2) w = (1.0f - pp.x) * pp.y;
  if (corner == 3) w = pp.x * pp.y;
  float dc = lD - lC;
  float cb = lC - lB;
  float lenX = fmax_(fabsf(dc), fabsf(cb));
  lenX = fsr_prx_lo_rcp(lenX);
  float dirX = lD - lB;
  dir->x += dirX * w;
  lenX = clamp_(fabsf(dirX) * lenX, 0.0f, 1.0f);
  lenX *= lenX;
  *len += lenX * w;
  float ec = lE - lC;
  float ca = lC - lA;
  float lenY = fmax_(fabsf(ec), fabsf(ca));
  lenY = fsr_prx_lo_rcp(lenY);
  float dirY = lE - lA;
  dir->y += dirY * w;
  lenY = clamp_(fabsf(dirY) * lenY, 0.0f, 1.0f);
  lenY *= lenY;
  *len += lenY * w;
}
static void pass_fsr_easu(Ctx* c, int y0, int y1) {  // FSR_Pass.glsl CurrFilter (SAMPLE_EASU) + ffx_fsr1.h:315-441
  Tex input = tex(c, c->taa == HK_TAA_JASMINE ? HK_BUF_TAA_OUTPUT : HK_BUF_TONE_MAPPED);  // post_process.rs:1037-1040
  Tex output = tex(c, HK_BUF_UPSCALE_OUTPUT);
  const FsrEasuConst k = fsr_easu_con((float)input.w, (float)input.h, (float)input.w, (float)input.h, (float)output.w, (float)output.h);
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < output.w; ++x) {
      v2 pp = V2((float)x * k.con0[0] + k.con0[2], (float)y * k.con0[1] + k.con0[3]);
      v2 fp = V2(floorf(pp.x), floorf(pp.y));
      pp = pp - fp;
      v2 p0 = V2(fp.x * k.con1[0] + k.con1[2], fp.y * k.con1[1] + k.con1[3]);
      v2 p1 = V2(p0.x + k.con2[0], p0.y + k.con2[1]);
      v2 p2 = V2(p0.x + k.con2[2], p0.y + k.con2[3]);
      v2 p3 = V2(p0.x + k.con3[0], p0.y + k.con3[1]);
      FsrGather bczz = fsr_gather(input, p0), ijfe = fsr_gather(input, p1), klhg = fsr_gather(input, p2), zzon = fsr_gather(input, p3);
      auto luma = [](const FsrGather& t) { return t.b * 0.5f + (t.r * 0.5f + t.g); };
      v4 bczzL = luma(bczz), ijfeL = luma(ijfe), klhgL = luma(klhg), zzonL = luma(zzon);
      float bL = bczzL.x, cL = bczzL.y, iL = ijfeL.x, jL = ijfeL.y, fL = ijfeL.z, eL = ijfeL.w, kL = klhgL.x, lL = klhgL.y, hL = klhgL.z,
            gL = klhgL.w, oL = zzonL.z, nL = zzonL.w;
      v2 dir = V2(0.0f, 0.0f);
      float len = 0.0f;
      fsr_easu_set(&dir, &len, pp, 0, bL, eL, fL, gL, jL);
      fsr_easu_set(&dir, &len, pp, 1, cL, fL, gL, hL, kL);
      fsr_easu_set(&dir, &len, pp, 2, fL, iL, jL, kL, nL);
      fsr_easu_set(&dir, &len, pp, 3, gL, jL, kL, lL, oL);
      v2 dir2 = dir * dir;
      float dirR = dir2.x + dir2.y;
      bool zro = dirR < (float)(1.0 / 32768.0);
      dirR = fsr_prx_lo_rsq(dirR);
      dirR = zro ? 1.0f : dirR;
      dir.x = zro ? 1.0f : dir.x;
      dir = dir * V2(dirR, dirR);
      len = len * 0.5f;
      len *= len;
      float stretch = (dir.x * dir.x + dir.y * dir.y) * fsr_prx_lo_rcp(fmax_(fabsf(dir.x), fabsf(dir.y)));
      v2 len2 = V2(1.0f + (stretch - 1.0f) * len, 1.0f + -0.5f * len);
      float lob = 0.5f + (float)((1.0 / 4.0 - 0.04) - 0.5) * len;
      float clp = fsr_prx_lo_rcp(lob);
      v3 fC = V3(ijfe.r.z, ijfe.g.z, ijfe.b.z), gC = V3(klhg.r.w, klhg.g.w, klhg.b.w), jC = V3(ijfe.r.y, ijfe.g.y, ijfe.b.y),
         kC = V3(klhg.r.x, klhg.g.x, klhg.b.x);
      v3 min4 = min3(V3(fsr_min3(fC.x, gC.x, jC.x), fsr_min3(fC.y, gC.y, jC.y), fsr_min3(fC.z, gC.z, jC.z)), kC);
      v3 max4 = max3(V3(fsr_max3(fC.x, gC.x, jC.x), fsr_max3(fC.y, gC.y, jC.y), fsr_max3(fC.z, gC.z, jC.z)), kC);
      v3 aC = V3(0.0f, 0.0f, 0.0f);
      float aW = 0.0f;
      auto tap = [&](float ox, float oy, float r, float g, float b) { fsr_easu_tap(&aC, &aW, V2(ox - pp.x, oy - pp.y), dir, len2, lob, clp, V3(r, g, b)); };
      tap(0.0f, -1.0f, bczz.r.x, bczz.g.x, bczz.b.x);  // b
      tap(1.0f, -1.0f, bczz.r.y, bczz.g.y, bczz.b.y);  // c
      tap(-1.0f, 1.0f, ijfe.r.x, ijfe.g.x, ijfe.b.x);  // i
      tap(0.0f, 1.0f, ijfe.r.y, ijfe.g.y, ijfe.b.y);   // j
      tap(0.0f, 0.0f, ijfe.r.z, ijfe.g.z, ijfe.b.z);   // f
      tap(-1.0f, 0.0f, ijfe.r.w, ijfe.g.w, ijfe.b.w);  // e
      tap(1.0f, 1.0f, klhg.r.x, klhg.g.x, klhg.b.x);   // k
      tap(2.0f, 1.0f, klhg.r.y, klhg.g.y, klhg.b.y);   // l
      tap(2.0f, 0.0f, klhg.r.z, klhg.g.z, klhg.b.z);   // h
      tap(1.0f, 0.0f, klhg.r.w, klhg.g.w, klhg.b.w);   // g
      tap(1.0f, 2.0f, zzon.r.z, zzon.g.z, zzon.b.z);   // o
      tap(0.0f, 2.0f, zzon.r.w, zzon.g.w, zzon.b.w);   // n
      v3 pix = min3(max4, max3(min4, aC * (1.0f / aW)));
      output.store_f16x4(x, y, V4(pix, 1.0f));          // hdr == 0: no c *= c
    }
}
static void pass_fsr_rcas(Ctx* c, int y0, int y1) {  // FSR_Pass.glsl CurrFilter (SAMPLE_RCAS) + ffx_fsr1.h:662-768
  Tex input = tex(c, HK_BUF_UPSCALE_OUTPUT), output = tex(c, HK_BUF_UPSCALE_SHARPENED);
  const float sharp = exp2_(-c->upscale_sharpness);  // FsrRcasCon
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < output.w; ++x) {
      v4 b = input.load_f16x4(x, y - 1), d = input.load_f16x4(x - 1, y), e = input.load_f16x4(x, y), f = input.load_f16x4(x + 1, y),
         h = input.load_f16x4(x, y + 1);  // texelFetch: zeros outside
      float bL = b.z * 0.5f + (b.x * 0.5f + b.y), dL = d.z * 0.5f + (d.x * 0.5f + d.y), eL = e.z * 0.5f + (e.x * 0.5f + e.y),
            fL = f.z * 0.5f + (f.x * 0.5f + f.y), hL = h.z * 0.5f + (h.x * 0.5f + h.y);
      float nz = 0.25f * bL + 0.25f * dL + 0.25f * fL + 0.25f * hL - eL;
      nz = clamp_(fabsf(nz) * fsr_prx_med_rcp(fsr_max3(fsr_max3(bL, dL, eL), fL, hL) - fsr_min3(fsr_min3(bL, dL, eL), fL, hL)), 0.0f, 1.0f);
      nz = -0.5f * nz + 1.0f;  // (computed as in the source; FSR_RCAS_DENOISE is not defined, so it is unused)
      (void)nz;
      float mn4R = fmin_(fsr_min3(b.x, d.x, f.x), h.x), mn4G = fmin_(fsr_min3(b.y, d.y, f.y), h.y), mn4B = fmin_(fsr_min3(b.z, d.z, f.z), h.z);
      float mx4R = fmax_(fsr_max3(b.x, d.x, f.x), h.x), mx4G = fmax_(fsr_max3(b.y, d.y, f.y), h.y), mx4B = fmax_(fsr_max3(b.z, d.z, f.z), h.z);
      const float peakCx = 1.0f, peakCy = -1.0f * 4.0f;
      float hitMinR = fmin_(mn4R, e.x) * (1.0f / (4.0f * mx4R)), hitMinG = fmin_(mn4G, e.y) * (1.0f / (4.0f * mx4G)),
            hitMinB = fmin_(mn4B, e.z) * (1.0f / (4.0f * mx4B));
      float hitMaxR = (peakCx - fmax_(mx4R, e.x)) * (1.0f / (4.0f * mn4R + peakCy)), hitMaxG = (peakCx - fmax_(mx4G, e.y)) * (1.0f / (4.0f * mn4G + peakCy)),
            hitMaxB = (peakCx - fmax_(mx4B, e.z)) * (1.0f / (4.0f * mn4B + peakCy));
      float lobeR = fmax_(-hitMinR, hitMaxR), lobeG = fmax_(-hitMinG, hitMaxG), lobeB = fmax_(-hitMinB, hitMaxB);
      float lobe = fmax_(-(float)(0.25 - (1.0 / 16.0)), fmin_(fsr_max3(lobeR, lobeG, lobeB), 0.0f)) * sharp;
      float rcpL = fsr_prx_med_rcp(4.0f * lobe + 1.0f);
      float pixR = (lobe * b.x + lobe * d.x + lobe * h.x + lobe * f.x + e.x) * rcpL;
      float pixG = (lobe * b.y + lobe * d.y + lobe * h.y + lobe * f.y + e.y) * rcpL;
      float pixB = (lobe * b.z + lobe * d.z + lobe * h.z + lobe * f.z + e.z) * rcpL;
      output.store_f16x4(x, y, V4(pixR, pixG, pixB, 1.0f));
    }
}

static inline v3 differential_blend_factor(v4 t, v4 b, v4 n, v4 e, v4 s, v4 w) {  // smaa.wgsl:198-222
  v2 dh = V2(luminance(abs3(rgb(w) - rgb(b))), luminance(abs3(rgb(t) - rgb(e))));
  v2 dv = V2(luminance(abs3(rgb(t) - rgb(s))), luminance(abs3(rgb(n) - rgb(b))));
  v2 factor_xy = V2(fmax_(dv.x, 0.001f) * fmax_(dv.y, 0.001f), fmax_(dh.x, 0.001f) * fmax_(dh.y, 0.001f));
  float factor_z = 1.0f / (factor_xy.x + factor_xy.y);
  return V3(factor_xy.x, factor_xy.y, factor_z);
}
static inline v4 differential_blend(v4 t, v4 b, v4 l, v4 r, v3 factor) {  // smaa.wgsl:224-235
  v4 color = V4(0.0f, 0.0f, 0.0f, 0.0f);
  color = color + (l + r) * factor.x;
  color = color + (t + b) * factor.y;
  return (0.5f * factor.z) * color;
}
static void pass_smaa_tu4x_extrapolate(Ctx* c, int y0, int y1) {  // smaa.wgsl:239-271; in place on upscale_output[0]
  Tex output = tex(c, HK_BUF_UPSCALE_OUTPUT);
  const int rw = c->RW;
  // every thread reads only pixels smaa_tu4x wrote (both diagonals' (0,0)/(1,1) slots) and writes only (0,1)/(1,0)
  // slots: no thread reads what another writes, so the in-place update is order-independent.
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < rw; ++x) {
      const int bx = 2 * x, by = 2 * y;
      v4 t_color = output.load_f16x4(bx, by);
      v4 b_color = output.load_f16x4(bx + 1, by + 1);
      v4 n_color = output.load_f16x4(bx + 1, by - 1);
      v4 e_color = output.load_f16x4(bx + 2, by);
      v4 s_color = output.load_f16x4(bx, by + 2);
      v4 w_color = output.load_f16x4(bx - 1, by + 1);
      v3 factor = differential_blend_factor(t_color, b_color, n_color, e_color, s_color, w_color);
      v4 x_color = differential_blend(t_color, s_color, w_color, b_color, factor);
      v4 y_color = differential_blend(n_color, b_color, t_color, e_color, factor);
      output.store_f16x4(bx, by + 1, x_color);
      output.store_f16x4(bx + 1, by, y_color);
    }
}

}  // namespace orc

// ==========================================================================================
// C API - mirrors include/hikari_hip.h with the prefix orc_
// ==========================================================================================
using namespace orc;
struct orc_ctx { Ctx c; };

#define ORC_CHECK(cond, code, msg) do { if (!(cond)) { g_err = msg; return code; } } while (0)

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }
uint32_t orc_abi_version(void) { return HK_ABI_VERSION; }

int orc_create(int device_id, uint32_t flags, orc_ctx** out) {
  (void)device_id;
  ORC_CHECK(out, HK_E_INVALID, "out is NULL");
  orc_ctx* p = new orc_ctx();
  p->c.flags = flags;
  *out = p;
  return HK_OK;
}
void orc_destroy(orc_ctx* ctx) { delete ctx; }
int orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }

int orc_upload_meshes(orc_ctx* ctx, const HkVertex* v, uint32_t nv, const HkPrimitive* p, uint32_t np, const HkNode* n, uint32_t nn) {
  ORC_CHECK(ctx && (v || !nv) && (p || !np) && (n || !nn), HK_E_INVALID, "null argument");
  ctx->c.vertices.assign(v, v + nv);
  ctx->c.primitives.assign(p, p + np);
  ctx->c.asset_nodes.assign(n, n + nn);
  return HK_OK;
}
int orc_upload_materials(orc_ctx* ctx, const HkMaterial* m, uint32_t n) {
  ORC_CHECK(ctx && (m || !n), HK_E_INVALID, "null argument");
  ctx->c.materials.assign(m, m + n);
  return HK_OK;
}
int orc_upload_instances(orc_ctx* ctx, const HkInstance* inst, uint32_t ni, const HkNode* inodes, uint32_t nin, const HkEmissive* em,
                         uint32_t ne, const HkNode* enodes, uint32_t nen, const HkAliasEntry* alias, uint32_t na) {
  ORC_CHECK(ctx && (inst || !ni) && (inodes || !nin) && (em || !ne) && (enodes || !nen) && (alias || !na), HK_E_INVALID, "null argument");
  ctx->c.instances.assign(inst, inst + ni);
  ctx->c.instance_nodes.assign(inodes, inodes + nin);
  ctx->c.emissives.assign(em, em + ne);
  ctx->c.emissive_nodes.assign(enodes, enodes + nen);
  ctx->c.alias_table.assign(alias, alias + na);
  ctx->c.prev_models.clear();
  return HK_OK;
}
int orc_upload_previous_transforms(orc_ctx* ctx, const float* models, uint32_t n) {
  ORC_CHECK(ctx && (models || !n), HK_E_INVALID, "null argument");
  ORC_CHECK(n == ctx->c.instances.size(), HK_E_INVALID, "previous transforms must match the uploaded instances");
  ctx->c.prev_models.assign(models, models + 16 * (size_t)n);
  return HK_OK;
}
int orc_upload_textures(orc_ctx* ctx, const HkImageDesc* images, uint32_t n) {
  ORC_CHECK(ctx && (images || !n), HK_E_INVALID, "null argument");
  ctx->c.textures.clear();
  for (uint32_t i = 0; i < n; ++i) {
    const HkImageDesc& d = images[i];
    ORC_CHECK(d.rgba8 && d.width && d.height && d.address_u <= 2 && d.address_v <= 2, HK_E_INVALID, "bad image");
    Ctx::Texture t;
    t.rgba.assign(d.rgba8, d.rgba8 + (size_t)d.width * d.height * 4);
    t.w = d.width; t.h = d.height; t.is_srgb = d.is_srgb; t.au = d.address_u; t.av = d.address_v; t.linear = d.filter_linear;
    ctx->c.textures.push_back(std::move(t));
  }
  for (int i = 0; i < 256; ++i) {  // sRGB EOTF, evaluated in double and rounded once
    double c = i / 255.0;
    ctx->c.srgb_lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
  }
  return HK_OK;
}
int orc_upload_noise(orc_ctx* ctx, const uint8_t* rgba, size_t bytes) {
  ORC_CHECK(ctx && rgba && bytes == 16u * 64u * 64u * 4u, HK_E_INVALID, "noise must be 16x64x64x4 bytes");
  ctx->c.noise.assign(rgba, rgba + bytes);
  return HK_OK;
}
int orc_resize(orc_ctx* ctx, uint32_t width, uint32_t height, float upscale_ratio) {
  ORC_CHECK(ctx && width && height, HK_E_INVALID, "bad size");
  Ctx& c = ctx->c;
  float ratio = clamp_(upscale_ratio, 1.0f, 2.0f);  // lib.rs:500-504
  c.W = (int)width;
  c.H = (int)height;
  c.ratio = ratio;
  float scale = 1.0f / ratio;                        // light.rs:318-319
  c.RW = (int)ceilf(scale * (float)width);
  c.RH = (int)ceilf(scale * (float)height);
  c.UW = (int)ceilf((float)width * (scale * 2.0f));  // post_process.rs:718-722
  c.UH = (int)ceilf((float)height * (scale * 2.0f));
  c.band_bounds.clear();  // (boundaries are render rows of the old size)
  for (uint32_t b = 0; b < HK_BUF_COUNT; ++b) {
    size_t n = buf_full_size(b) ? (size_t)c.W * c.H : (buf_upscaled(b) ? (size_t)c.UW * c.UH : (size_t)c.RW * c.RH);
    c.buf[b].assign(n * buf_bpp(b), (b >= HK_BUF_PARKED_TO0 && b < HK_BUF_PARKED_TO0 + 3) ? 0xFF : 0);  // (-1: nothing parked)
  }
  c.mapped_parity = 0;
  return HK_OK;
}
int orc_set_view_options(orc_ctx* ctx, uint32_t taa, uint32_t upscale_kind, float upscale_sharpness) {
  ORC_CHECK(ctx, HK_E_INVALID, "null ctx");
  ctx->c.upscale_sharpness = upscale_sharpness;
  ctx->c.taa = taa;
  ctx->c.upscale_kind = upscale_kind;
  return HK_OK;
}
int orc_frame_begin(orc_ctx* ctx, const HkFrame* f, const HkView* v, const HkPreviousView* pv, const HkLights* l) {
  ORC_CHECK(ctx && f && v && pv && l, HK_E_INVALID, "null argument");
  ctx->c.frame = *f;
  ctx->c.view = *v;
  ctx->c.pview = *pv;
  ctx->c.lights = *l;
  ctx->c.have_frame = true;
  // double-buffered planes follow frame.number % 2 (hikari_hip.h, HkBuffer): idempotent per frame number
  if ((f->number & 1u) != ctx->c.mapped_parity) {
    Ctx& c = ctx->c;
    c.buf[HK_BUF_POSITION].swap(c.buf[HK_BUF_PREVIOUS_POSITION]);
    c.buf[HK_BUF_VELOCITY_UV].swap(c.buf[HK_BUF_PREVIOUS_VELOCITY_UV]);
    c.buf[HK_BUF_TONE_MAPPED].swap(c.buf[HK_BUF_PREVIOUS_TONE_MAPPED]);
    c.buf[HK_BUF_TAA_OUTPUT].swap(c.buf[HK_BUF_PREVIOUS_TAA_OUTPUT]);
    c.mapped_parity = f->number & 1u;
  }
  return HK_OK;
}
static int ready(orc_ctx* ctx) {
  ORC_CHECK(ctx, HK_E_INVALID, "null ctx");
  Ctx& c = ctx->c;
  ORC_CHECK(c.W > 0 && c.have_frame && !c.noise.empty() && !c.instance_nodes.empty(), HK_E_NOT_READY, "scene / noise / size / uniforms missing");
  return HK_OK;
}
int orc_pass_run(orc_ctx* ctx, uint32_t pass, uint32_t arg, uint32_t row_begin, uint32_t row_end) {
  int rc = ready(ctx);
  if (rc) return rc;
  Ctx& c = ctx->c;
  const bool full_grid = pass == HK_PASS_PREPASS || pass == HK_PASS_FULL_SCREEN_ALBEDO;
  int rows = full_grid ? c.H : c.RH;
  if (pass == HK_PASS_TAA_JASMINE) { int w; buf_dims(&c, HK_BUF_TAA_OUTPUT, &w, &rows); }
  if (pass == HK_PASS_FSR_EASU || pass == HK_PASS_FSR_RCAS) {
    ORC_CHECK(c.upscale_kind == HK_UPSCALE_FSR1, HK_E_INVALID, "FSR passes need upscale_kind FSR1");
    rows = c.H;
  }
  int y0 = (int)row_begin, y1 = row_end == 0 ? rows : (int)row_end;
  ORC_CHECK(y0 >= 0 && y1 <= rows && y0 <= y1, HK_E_INVALID, "row range");
  switch (pass) {
    case HK_PASS_PREPASS: pass_prepass(&c, y0, y1); break;
    case HK_PASS_SMAA_TU4X: pass_smaa_tu4x(&c, y0, y1); break;
    case HK_PASS_SMAA_TU4X_EXTRAPOLATE: pass_smaa_tu4x_extrapolate(&c, y0, y1); break;
    case HK_PASS_TAA_JASMINE: pass_taa_jasmine(&c, y0, y1); break;
    case HK_PASS_FSR_EASU: pass_fsr_easu(&c, y0, y1); break;
    case HK_PASS_FSR_RCAS: pass_fsr_rcas(&c, y0, y1); break;
    case HK_PASS_FULL_SCREEN_ALBEDO: pass_full_screen_albedo(&c, y0, y1); break;
    case HK_PASS_DIRECT_LIT: pass_direct_lit(&c, false, y0, y1); break;
    case HK_PASS_DIRECT_EMISSIVE: pass_direct_lit(&c, true, y0, y1); break;
    case HK_PASS_INDIRECT: pass_indirect(&c, y0, y1); break;
    case HK_PASS_EMISSIVE_SPATIAL_REUSE: pass_spatial_reuse(&c, true, y0, y1); break;
    case HK_PASS_INDIRECT_SPATIAL_REUSE: pass_spatial_reuse(&c, false, y0, y1); break;
    case HK_PASS_DEMODULATION: ORC_CHECK(arg < 3, HK_E_INVALID, "channel"); pass_demodulation(&c, (int)arg, y0, y1); break;
    case HK_PASS_DENOISE_L0: case HK_PASS_DENOISE_L1: case HK_PASS_DENOISE_L2: case HK_PASS_DENOISE_L3:
      ORC_CHECK(arg < 3, HK_E_INVALID, "channel");
      pass_denoise(&c, (int)arg, (int)(pass - HK_PASS_DENOISE_L0), y0, y1);
      break;
    case HK_PASS_TONE_MAPPING: pass_tone_mapping(&c, arg != 0, y0, y1); break;
    default: ORC_CHECK(false, HK_E_INVALID, "unknown pass");
  }
  return HK_OK;
}

static void full_rows_for(Ctx& c, int ry0, int ry1, int* fy0, int* fy1) {
  // rows of the full-size image that scaled rows [ry0,ry1) read (jittered_deferred_coords)
  if (c.RH == c.H) { *fy0 = ry0; *fy1 = ry1; return; }
  *fy0 = std::max(0, (int)floorf((float)ry0 * (float)c.H / (float)c.RH) - 1);
  *fy1 = std::min(c.H, (int)ceilf((float)ry1 * (float)c.H / (float)c.RH) + 1);
}

// Node order of the reference for the rows of one band; see hk_frame_stage in hikari_hip.h.
// Apron sizes restate the kernel footprints: spatial reuse reads <= range px (+1 for the
// truncation / ray-march tap), the 4 a-trous levels 8+4+2+1 = 15 rows, the variance prefilter 1.
int orc_frame_stage_rows(orc_ctx* ctx, uint32_t stage, const HkSettings* st, uint32_t flags, uint32_t band_begin, uint32_t band_end) {
  int rc = ready(ctx);
  if (rc) return rc;
  ORC_CHECK(st, HK_E_INVALID, "null settings");
  Ctx& c = ctx->c;
  c.taa = st->taa;
  c.upscale_kind = st->upscale_kind;
  c.upscale_sharpness = st->upscale_sharpness;
  const int b0 = (int)band_begin, b1 = (int)band_end;
  auto clampr = [&](int v) { return std::min(std::max(v, 0), c.RH); };
  const int den = st->denoise ? 16 : 0;
  const int sp = st->indirect_spatial_reuse ? 21 : (st->emissive_spatial_reuse ? 11 : 0);  // the dispatch runs regardless of bounces, light.rs:676
  if (stage == HK_STAGE_TEMPORAL) {
    int g0 = clampr(b0 - std::max(den, 0) - sp), g1 = clampr(b1 + std::max(den, 0) + sp);
    int f0, f1;
    full_rows_for(c, g0, g1, &f0, &f1);
    if (!(flags & HK_FRAME_EXTERNAL_GBUFFER)) pass_prepass(&c, f0, f1);
    int a0, a1;
    full_rows_for(c, clampr(b0 - den), clampr(b1 + den), &a0, &a1);
    pass_full_screen_albedo(&c, a0, a1);
    pass_direct_lit(&c, false, b0, b1);
    pass_direct_lit(&c, true, b0, b1);
    pass_indirect(&c, b0, b1);
  } else if (stage == HK_STAGE_SPATIAL) {
    if (parks_across_bands(&c)) {
      // exchange A brought the parked stores of the pixels up to 2 x history rows outside the band (HK_STAGE_SPATIAL_WITH_HISTORY);
      // per channel in dispatch order - sun and emissive store into the same buffer.  A channel whose spatial pass is off has
      // no reader and no exchanged rows: its stores are resolved among the band's own pixels.
      const int reach = 2 * (int)c.history_rows;
      for (int channel = 0; channel < 3; ++channel) {
        const bool exchanged = channel == 2 ? st->indirect_spatial_reuse != 0 : st->emissive_spatial_reuse != 0;
        ReservoirSet rs = reservoir_set(&c, channel);
        resolve_parked(&c, channel, rs.previous_spatial, exchanged ? clampr(b0 - reach) : b0, exchanged ? clampr(b1 + reach) : b1);
      }
    }
    if (st->emissive_spatial_reuse) pass_spatial_reuse(&c, true, b0, b1);       // light.rs:675,689-697
    if (st->indirect_spatial_reuse) pass_spatial_reuse(&c, false, b0, b1);      // light.rs:676
  } else if (stage == HK_STAGE_POST_PROCESS) {
    if (st->denoise) {  // post_process.rs:1190-1224
      int nch = st->indirect_bounces == 0 ? 2 : 3;  // post_process.rs:949-954
      for (int ch = 0; ch < nch; ++ch) {
        pass_demodulation(&c, ch, clampr(b0 - 15), clampr(b1 + 15));
        pass_denoise(&c, ch, 0, clampr(b0 - 7), clampr(b1 + 7));
        pass_denoise(&c, ch, 1, clampr(b0 - 3), clampr(b1 + 3));
        pass_denoise(&c, ch, 2, clampr(b0 - 1), clampr(b1 + 1));
        pass_denoise(&c, ch, 3, b0, b1);
      }
    }
    pass_tone_mapping(&c, st->denoise != 0, b0, b1);
    c.stats.frames++;
  } else if (stage == HK_STAGE_ANTIALIAS) {  // post_process.rs:1236-1272
    const bool smaa = st->upscale_kind == HK_UPSCALE_SMAA_TU4X;
    if (smaa) {  // aprons: see hk_band_plan_for (exchange D)
      pass_smaa_tu4x(&c, clampr(b0 - 2), clampr(b1 + 2));
      pass_smaa_tu4x_extrapolate(&c, clampr(b0 - 1), clampr(b1 + 1));
    }
    if (st->taa == HK_TAA_JASMINE) {
      int w, h;
      buf_dims(&c, HK_BUF_TAA_OUTPUT, &w, &h);
      const int scale = smaa ? 2 : 1;
      pass_taa_jasmine(&c, std::min(h, scale * b0), b1 == c.RH ? h : std::min(h, scale * b1));
    }
  } else if (stage == HK_STAGE_UPSCALE) {  // post_process.rs:1277-1308
    if (st->upscale_kind == HK_UPSCALE_FSR1) {
      const int bi = (int)c.band_index, bn = (int)c.band_count, base = c.H / bn, rem = c.H % bn;  // hk_band_rows over the window height
      int w0 = bi * base + std::min(bi, rem), w1 = w0 + base + (bi < rem ? 1 : 0);
      if (c.band_bounds.size() == (size_t)bn + 1) {  // an explicit split of the render rows cuts the window rows where its boundaries fall in them
        auto cut = [&](int k) { return k == 0 ? 0 : (k == bn ? c.H : (int)(((uint64_t)c.band_bounds[k] * (uint64_t)c.H) / (uint64_t)c.RH)); };
        w0 = cut(bi);
        w1 = cut(bi + 1);
      }
      pass_fsr_easu(&c, std::max(w0 - 1, 0), std::min(w1 + 1, c.H));
      pass_fsr_rcas(&c, w0, w1);
    }
  } else {
    ORC_CHECK(false, HK_E_INVALID, "unknown stage");
  }
  return HK_OK;
}
int orc_set_band_bounds(orc_ctx* ctx, const uint32_t* bounds, uint32_t n_bounds) {  // hk_set_band_bounds
  ORC_CHECK(ctx, HK_E_INVALID, "null ctx");
  if (!bounds || n_bounds == 0) {
    ctx->c.band_bounds.clear();
    return HK_OK;
  }
  ORC_CHECK(n_bounds == ctx->c.band_count + 1 && bounds[0] == 0 && (int)bounds[n_bounds - 1] == ctx->c.RH, HK_E_INVALID, "band bounds");
  for (uint32_t k = 0; k + 1 < n_bounds; ++k) ORC_CHECK(bounds[k + 1] > bounds[k], HK_E_INVALID, "band bounds must increase");
  ctx->c.band_bounds.assign(bounds, bounds + n_bounds);
  return HK_OK;
}
int orc_row_costs(orc_ctx* ctx, uint32_t* out, uint32_t n_rows) {  // hk_row_costs: pixels of a row whose depth (position.w) is not < epsilon
  ORC_CHECK(ctx && out && (int)n_rows == ctx->c.H && ctx->c.H > 0, HK_E_INVALID, "one counter per full-size row");
  const float* p = reinterpret_cast<const float*>(ctx->c.buf[HK_BUF_POSITION].data());
  for (int y = 0; y < ctx->c.H; ++y) {
    uint32_t n = 0;
    for (int x = 0; x < ctx->c.W; ++x) n += !(p[4 * ((size_t)y * ctx->c.W + x) + 3] < 1.1920929e-7f) ? 1u : 0u;
    out[y] = n;
  }
  return HK_OK;
}
int orc_set_history_rows(orc_ctx* ctx, uint32_t rows) {  // hk_set_history_rows; the oracle takes counts only (no HK_HISTORY_AUTO: the bound is host logic of the product)
  ORC_CHECK(ctx && rows < HK_HISTORY_AUTO, HK_E_INVALID, "history rows: a count");
  ctx->c.history_rows = rows;
  return HK_OK;
}
int orc_history_rows(orc_ctx* ctx, uint32_t* rows) {
  ORC_CHECK(ctx && rows, HK_E_INVALID, "argument");
  *rows = ctx->c.band_count > 1 ? std::min(ctx->c.history_rows, (uint32_t)ctx->c.RH) : 0u;
  return HK_OK;
}
int orc_scene_bounds(orc_ctx* ctx, float mn[3], float mx[3]) {  // the union of the instances' boxes
  ORC_CHECK(ctx && mn && mx && !ctx->c.instances.empty(), HK_E_INVALID, "argument");
  for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
  for (const HkInstance& in : ctx->c.instances)
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(mn[k], in.min[k]);
      mx[k] = std::max(mx[k], in.max[k]);
    }
  return HK_OK;
}
int orc_set_band(orc_ctx* ctx, uint32_t band_index, uint32_t band_count) {
  ORC_CHECK(ctx && band_count > 0 && band_index < band_count, HK_E_INVALID, "band");
  if (band_count != ctx->c.band_count) ctx->c.band_bounds.clear();
  ctx->c.band_index = band_index;
  ctx->c.band_count = band_count;
  return HK_OK;
}
static void band_rows(uint32_t height, uint32_t i, uint32_t n, uint32_t* b0, uint32_t* b1) {
  uint32_t base = height / n, rem = height % n;
  *b0 = i * base + std::min(i, rem);
  *b1 = *b0 + base + (i < rem ? 1u : 0u);
}
int orc_frame_stage(orc_ctx* ctx, uint32_t stage, const HkSettings* st, uint32_t flags) {
  ORC_CHECK(ctx, HK_E_INVALID, "null ctx");
  uint32_t b0, b1;
  band_rows((uint32_t)ctx->c.RH, ctx->c.band_index, ctx->c.band_count, &b0, &b1);
  if (ctx->c.band_bounds.size() == (size_t)ctx->c.band_count + 1) {
    b0 = ctx->c.band_bounds[ctx->c.band_index];
    b1 = ctx->c.band_bounds[ctx->c.band_index + 1];
  }
  return orc_frame_stage_rows(ctx, stage, st, flags, b0, b1);
}
int orc_frame_render(orc_ctx* ctx, const HkFrame* f, const HkView* v, const HkPreviousView* pv, const HkLights* l, const HkSettings* st, uint32_t flags) {
  int rc = orc_frame_begin(ctx, f, v, pv, l);
  if (rc) return rc;
  for (uint32_t s = 0; s <= HK_STAGE_POST_PROCESS; ++s) {
    rc = orc_frame_stage(ctx, s, st, flags);
    if (rc) return rc;
  }
  if (flags & HK_FRAME_ANTIALIAS) {
    int rc = orc_frame_stage(ctx, HK_STAGE_ANTIALIAS, st, flags);
    return rc ? rc : orc_frame_stage(ctx, HK_STAGE_UPSCALE, st, flags);
  }
  return HK_OK;
}
int orc_frame_wait(orc_ctx* ctx) { (void)ctx; return HK_OK; }
// Measurement hook for tests/tools/divergence_model.py (not part of the mirrored API): while armed, pass_indirect
// records for every pixel the traversal work of its rays in call order - (inner-node visits, triangle
// tests, instance entries) per traverse_top call / stand-alone traverse_bottom call.  buf = [RH][RW][slots][3] u16.
int orc_debug_record_steps(orc_ctx* ctx, uint16_t* buf, uint32_t slots) {
  ORC_CHECK(ctx, HK_E_INVALID, "null ctx");
  ctx->c.step_rec = buf;
  ctx->c.step_slots = buf ? slots : 0;
  return HK_OK;
}
static size_t logical_bytes(const Ctx* c, uint32_t buffer) {
  int w, h;
  buf_dims(c, buffer, &w, &h);
  return (size_t)w * h * buf_bpp(buffer);
}

int orc_buffer_info(orc_ctx* ctx, uint32_t buffer, uint32_t* w, uint32_t* h, uint32_t* bpp) {
  ORC_CHECK(ctx && buffer < HK_BUF_COUNT && buf_bpp(buffer), HK_E_INVALID, "buffer id");
  int bw, bh;
  buf_dims(&ctx->c, buffer, &bw, &bh);
  if (w) *w = (uint32_t)bw;
  if (h) *h = (uint32_t)bh;
  if (bpp) *bpp = (uint32_t)buf_bpp(buffer);
  return HK_OK;
}
int orc_read_buffer(orc_ctx* ctx, uint32_t buffer, void* dst, size_t bytes) {
  ORC_CHECK(ctx && dst && buffer < HK_BUF_COUNT, HK_E_INVALID, "argument");
  ORC_CHECK(bytes == logical_bytes(&ctx->c, buffer), HK_E_INVALID, "size mismatch");
  memcpy(dst, ctx->c.buf[buffer].data(), bytes);
  return HK_OK;
}
int orc_write_buffer(orc_ctx* ctx, uint32_t buffer, const void* src, size_t bytes) {
  ORC_CHECK(ctx && src && buffer < HK_BUF_COUNT, HK_E_INVALID, "argument");
  ORC_CHECK(bytes == logical_bytes(&ctx->c, buffer), HK_E_INVALID, "size mismatch");
  memcpy(ctx->c.buf[buffer].data(), src, bytes);
  return HK_OK;
}
int orc_device_ptr(orc_ctx* ctx, uint32_t buffer, void** ptr, size_t* bytes) {  // host pointer (the oracle's "device" is the CPU)
  ORC_CHECK(ctx && ptr && buffer < HK_BUF_COUNT, HK_E_INVALID, "argument");
  *ptr = ctx->c.buf[buffer].data();
  if (bytes) *bytes = logical_bytes(&ctx->c, buffer);
  return HK_OK;
}
int orc_get_stats(orc_ctx* ctx, HkStats* out) {
  ORC_CHECK(ctx && out, HK_E_INVALID, "argument");
  memset(out, 0, sizeof(*out));
  out->rays_primary = ctx->c.stats.rays_primary;
  out->rays_tlas = ctx->c.stats.rays_tlas;
  out->rays_blas = ctx->c.stats.rays_blas;
  out->frames = ctx->c.stats.frames;
  return HK_OK;
}
int orc_reset_stats(orc_ctx* ctx) {
  ORC_CHECK(ctx, HK_E_INVALID, "argument");
  ctx->c.stats.rays_primary = 0;
  ctx->c.stats.rays_tlas = 0;
  ctx->c.stats.rays_blas = 0;
  ctx->c.stats.frames = 0;
  return HK_OK;
}
// elementwise math for tests/test_math_contract.py (same op codes as hk_debug_math)
int orc_debug_math(orc_ctx* ctx, uint32_t op, const float* x, const float* y, float* out, size_t n) {
  (void)ctx;
  for (size_t i = 0; i < n; ++i) {
    float a = (op >= 16 && op <= 19) ? 0.0f : x[i], b = y ? y[i] : 0.0f, r = 0.0f;
    switch (op) {
      case 0: r = sin_(a); break;
      case 1: r = cos_(a); break;
      case 2: r = exp_(a); break;
      case 3: r = exp2_(a); break;
      case 4: r = log2_(a); break;
      case 5: r = pow_(a, b); break;
      case 6: r = fmin_(a, b); break;
      case 7: r = fmax_(a, b); break;
      case 8: r = f16_to_f32(f32_to_f16(a)); break;
      case 9: r = a / b; break;
      case 10: r = sqrtf(a); break;
      case 11: r = saturate(a); break;
      case 12: r = clamp_(a, -1.0f, 1.0f); break;
      case 13: r = saturate(a * b); break;
      case 14: r = unpack2x16unorm((uint32_t)a).x; break;
      case 15: r = unsnorm8((uint32_t)a); break;
      case 20: r = (float)((uint32_t)a & 0xffu) / 255.0f; break;
      case 16: case 17: case 18: case 19: {
        const float* q = x + 16 * i;
        HkLights lights{};
        lights.ambient_color[0] = lights.ambient_color[1] = lights.ambient_color[2] = 0.05f;
        Scene sc{};
        sc.lights = &lights;
        Surface sf;
        sf.base_color = V4(q[9], q[10], q[11], 1.0f);
        sf.emissive = V4(0, 0, 0, 0);
        sf.reflectance = 0.5f; sf.metallic = 0.0f; sf.roughness = perceptualRoughnessToRoughness(b); sf.occlusion = 1.0f;
        v3 V = normalize(V3(q[0], q[1], q[2])), N = normalize(V3(q[3], q[4], q[5])), L = normalize(V3(q[6], q[7], q[8]));
        v3 o = (op == 19) ? env_brdf(V, N, sf) : shading(sc, V, N, L, sf, V4(q[12], q[13], q[14], q[15]));
        r = (op == 17) ? o.y : ((op == 18) ? o.z : o.x);
        break;
      }
      default: return HK_E_INVALID;
    }
    out[i] = r;
  }
  return HK_OK;
}
// single-function hooks for known-answer tests
int orc_kat_intersects_aabb(const float o[3], const float d[3], const float mn[3], const float mx[3], float* t) {
  Ray r{P3(o), P3(d), 1.0f / P3(d)};
  *t = intersects_aabb(r, Aabb{P3(mn), P3(mx)});
  return HK_OK;
}
int orc_kat_intersects_triangle(const float o[3], const float d[3], const float tri[9], float uvt[3]) {
  Ray r{P3(o), P3(d), 1.0f / P3(d)};
  HkPrimitiveVertex v[3];
  for (int i = 0; i < 3; ++i) { v[i].position[0] = tri[3 * i]; v[i].position[1] = tri[3 * i + 1]; v[i].position[2] = tri[3 * i + 2]; v[i].index = 0; }
  Intersection it = intersects_triangle(r, v);
  uvt[0] = it.uv.x; uvt[1] = it.uv.y; uvt[2] = it.distance;
  return HK_OK;
}
int orc_kat_reservoir_roundtrip(const void* packed_in, void* packed_out) {
  PackedReservoir p;
  memcpy(&p, packed_in, 64);
  PackedReservoir q = pack_reservoir(unpack_reservoir(p));
  memcpy(packed_out, &q, 64);
  return HK_OK;
}
int orc_kat_normal_basis(const float n[3], float out9[9]) {
  m3 b = normal_basis(P3(n));
  out9[0] = b.c0.x; out9[1] = b.c0.y; out9[2] = b.c0.z; out9[3] = b.c1.x; out9[4] = b.c1.y; out9[5] = b.c1.z; out9[6] = b.c2.x; out9[7] = b.c2.y; out9[8] = b.c2.z;
  return HK_OK;
}
int orc_kat_hash(uint32_t v, uint32_t* h, float* f) { *h = orc::hash(v); *f = random_float(v); return HK_OK; }
// closest-hit query against the uploaded scene (used to cross-check BVH builders by brute force)
int orc_kat_trace(orc_ctx* ctx, const float o[3], const float d[3], float max_distance, float early_distance, uint32_t exclude,
                  uint32_t* instance, uint32_t* primitive, float* t, float uv[2]) {
  ORC_CHECK(ctx, HK_E_INVALID, "ctx");
  Scene sc = make_scene(&ctx->c);
  Ray r{P3(o), P3(d), 1.0f / P3(d)};
  Hit h = traverse_top(sc, r, max_distance, early_distance, exclude);
  *instance = h.instance_index; *primitive = h.primitive_index; *t = h.intersection.distance;
  uv[0] = h.intersection.uv.x; uv[1] = h.intersection.uv.y;
  return HK_OK;
}

}  // extern "C"
