// hikari.hpp - C++17 host-side mirror of bevy-hikari's plugin interface on top of the C ABI
// (include/hikari_hip.h).  Header-only; link with -lhikari_hip.
//
// The reference's host code is Rust (compiled); this image has no Rust toolchain, so the compiled
// host layer above the C ABI is C++ and mirrors the reference's types and call order
// (cryscan/bevy-hikari v0.3.15):
//   graph::NAME / node names             src/lib.rs:43-51
//   WORKGROUP_SIZE, NOISE_TEXTURE_COUNT  src/lib.rs:53-54
//   HikariUniversalSettings              src/lib.rs:373-397
//   HikariSettings, Taa, Upscale         src/lib.rs:400-513
//   FrameCounter                         src/view.rs:75-103
//   PrepassNode / LightNode / PostProcessNode ::run   src/prepass.rs:769, src/light.rs:590, src/post_process.rs:1140
//   HikariPlugin                         src/lib.rs:95-370
// Errors: the C ABI's negative codes become hikari::Error; where the reference's nodes silently
// return Ok(()) for a missing resource (light.rs:606-617) run() returns false instead of throwing.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hikari_hip.h"

namespace hikari {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc, const char* fn) {
  if (rc != HK_OK) throw Error(rc, std::string(fn) + " failed with code " + std::to_string(rc) + ": " + hk_last_error());
}

namespace graph {  // lib.rs:43-51
constexpr const char* NAME = "hikari";
namespace node {
constexpr const char* PREPASS = "hikari_prepass";
constexpr const char* LIGHT = "hikari_light";
constexpr const char* POST_PROCESS = "hikari_post_process";
constexpr const char* OVERLAY = "hikari_overlay";
}  // namespace node
}  // namespace graph
constexpr uint32_t WORKGROUP_SIZE = 8;        // lib.rs:53
constexpr uint32_t NOISE_TEXTURE_COUNT = 16;  // lib.rs:54

enum class Taa : uint32_t { Jasmine = HK_TAA_JASMINE, None = HK_TAA_NONE };  // lib.rs:466-472

struct Upscale {  // lib.rs:474-513
  uint32_t kind = HK_UPSCALE_SMAA_TU4X;
  float ratio_ = 2.0f, sharpness_ = 0.0f;
  static Upscale Fsr1(float ratio, float sharpness) { return Upscale{HK_UPSCALE_FSR1, ratio, sharpness}; }
  static Upscale SmaaTu4x(float ratio) { return Upscale{HK_UPSCALE_SMAA_TU4X, ratio, 0.0f}; }
  static Upscale SMAA_TU_1_0() { return SmaaTu4x(1.0f); }
  static Upscale SMAA_TU_2_0() { return SmaaTu4x(2.0f); }
  float ratio() const { return ratio_ < 1.0f ? 1.0f : (ratio_ > 2.0f ? 2.0f : ratio_); }
  float sharpness() const { return kind == HK_UPSCALE_FSR1 ? sharpness_ : 0.0f; }
};

struct HikariUniversalSettings {  // lib.rs:373-389
  bool build_mesh_acceleration_structure = true;
  bool build_instance_acceleration_structure = true;
};

struct HikariSettings {  // lib.rs:400-455, field order and defaults
  size_t direct_validate_interval = 3;
  size_t emissive_validate_interval = 5;
  size_t max_temporal_reuse_count = 50;
  size_t max_spatial_reuse_count = 800;
  float max_reservoir_lifetime = 100.0f;
  float solar_angle = 0.046f;
  size_t indirect_bounces = 1;
  float max_indirect_luminance = 10.0f;
  std::array<float, 4> clear_color{0.4f, 0.4f, 0.4f, 1.0f};
  bool temporal_reuse = true;
  bool emissive_spatial_reuse = false;
  bool indirect_spatial_reuse = true;
  bool denoise = true;
  Taa taa = Taa::Jasmine;
  Upscale upscale = Upscale::SMAA_TU_2_0();

  HkSettings to_c() const {
    HkSettings s{};
    s.direct_validate_interval = (uint32_t)direct_validate_interval;
    s.emissive_validate_interval = (uint32_t)emissive_validate_interval;
    s.max_temporal_reuse_count = (uint32_t)max_temporal_reuse_count;
    s.max_spatial_reuse_count = (uint32_t)max_spatial_reuse_count;
    s.max_reservoir_lifetime = max_reservoir_lifetime;
    s.solar_angle = solar_angle;
    s.indirect_bounces = (uint32_t)indirect_bounces;
    s.max_indirect_luminance = max_indirect_luminance;
    std::memcpy(s.clear_color, clear_color.data(), 16);
    s.temporal_reuse = temporal_reuse;
    s.emissive_spatial_reuse = emissive_spatial_reuse;
    s.indirect_spatial_reuse = indirect_spatial_reuse;
    s.denoise = denoise;
    s.taa = (uint32_t)taa;
    s.upscale_kind = upscale.kind;
    s.upscale_ratio = upscale.ratio_;
    s.upscale_sharpness = upscale.sharpness_;
    return s;
  }
};

struct FrameCounter {  // view.rs:75-103: inserted as 0 on a new camera, +1 every frame
  size_t value = 0;
  size_t tick() { return ++value; }
};

// ------------------------------------------------------------------ 4x4 helpers (double, column-major), glam conventions
using Mat4d = std::array<double, 16>;  // m[c*4+r]
inline Mat4d mul(const Mat4d& a, const Mat4d& b) {
  Mat4d o{};
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += a[k * 4 + r] * b[c * 4 + k];
      o[c * 4 + r] = s;
    }
  return o;
}
inline void to_f32(const Mat4d& m, float out[16]) {
  for (int i = 0; i < 16; ++i) out[i] = (float)m[i];
}

// Camera3dBundle with bevy's default PerspectiveProjection (fov pi/4, near 0.1, infinite reverse-Z)
struct Camera {
  Mat4d transform{};  // camera-to-world
  uint32_t width = 0, height = 0;
  double fov = 0.78539816339744830962, near_ = 0.1;

  // Transform::from_translation(eye).looking_at(target, up)
  static Camera looking_at(std::array<double, 3> eye, std::array<double, 3> target, std::array<double, 3> up, uint32_t w, uint32_t h) {
    auto sub = [](auto a, auto b) { return std::array<double, 3>{a[0] - b[0], a[1] - b[1], a[2] - b[2]}; };
    auto norm = [](std::array<double, 3> v) {
      double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      return std::array<double, 3>{v[0] / l, v[1] / l, v[2] / l};
    };
    auto cross = [](auto a, auto b) { return std::array<double, 3>{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; };
    auto f = norm(sub(target, eye));
    auto r = norm(cross(f, up));
    auto u = cross(r, f);
    Camera c;
    c.transform = Mat4d{r[0], r[1], r[2], 0, u[0], u[1], u[2], 0, -f[0], -f[1], -f[2], 0, eye[0], eye[1], eye[2], 1};
    c.width = w;
    c.height = h;
    return c;
  }
  Mat4d projection() const {  // Mat4::perspective_infinite_reverse_rh
    double f = 1.0 / std::tan(0.5 * fov), aspect = (double)width / (double)height;
    Mat4d m{};
    m[0] = f / aspect;
    m[5] = f;
    m[11] = -1.0;
    m[14] = near_;
    return m;
  }
  Mat4d inverse_projection() const {
    Mat4d p = projection(), m{};
    m[0] = 1.0 / p[0];
    m[5] = 1.0 / p[5];
    m[11] = 1.0 / p[14];
    m[14] = -1.0;
    return m;
  }
  Mat4d inverse_view() const {  // rigid inverse: [R^T | -R^T t]
    const Mat4d& t = transform;
    Mat4d m{};
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) m[c * 4 + r] = t[r * 4 + c];
    for (int r = 0; r < 3; ++r) m[12 + r] = -(t[r * 4 + 0] * t[12] + t[r * 4 + 1] * t[13] + t[r * 4 + 2] * t[14]);
    m[15] = 1.0;
    return m;
  }
  HkView view_uniform() const {  // bevy_render 0.9.1 ViewUniform
    HkView v{};
    Mat4d proj = projection(), inv_view = inverse_view();
    to_f32(mul(proj, inv_view), v.view_proj);
    to_f32(mul(transform, inverse_projection()), v.inverse_view_proj);
    to_f32(transform, v.view);
    to_f32(inv_view, v.inverse_view);
    to_f32(proj, v.projection);
    to_f32(inverse_projection(), v.inverse_projection);
    v.world_position[0] = (float)transform[12];
    v.world_position[1] = (float)transform[13];
    v.world_position[2] = (float)transform[14];
    v.viewport[2] = (float)width;
    v.viewport[3] = (float)height;
    return v;
  }
  HkPreviousView previous_view_uniform() const {
    HkView v = view_uniform();
    HkPreviousView p{};
    std::memcpy(p.view_proj, v.view_proj, 64);
    std::memcpy(p.inverse_view_proj, v.inverse_view_proj, 64);
    return p;
  }
};

// bevy AmbientLight default (white, 0.05); no directional light = zero-filled entry 0
inline HkLights lights_uniform(double ambient_brightness = 0.05) {
  HkLights l{};
  for (int k = 0; k < 3; ++k) l.ambient_color[k] = (float)(1.0 * ambient_brightness);
  l.ambient_color[3] = (float)ambient_brightness;
  return l;
}

// StandardMaterial -> GpuStandardMaterial (material.rs:168-199); glTF factors pass through (bevy_gltf 0.9.1
// builds colours with Color::rgba, for which `Color -> Vec4` is the identity)
inline HkMaterial standard_material(const float base_color[4], const float emissive_rgb[3], float perceptual_roughness, float metallic, float reflectance = 0.5f) {
  HkMaterial m{};
  std::memcpy(m.base_color, base_color, 16);
  m.emissive[0] = emissive_rgb[0];
  m.emissive[1] = emissive_rgb[1];
  m.emissive[2] = emissive_rgb[2];
  m.emissive[3] = 1.0f;
  m.base_color_texture = m.emissive_texture = m.metallic_roughness_texture = m.normal_map_texture = m.occlusion_texture = HK_NO_TEXTURE;
  m.perceptual_roughness = perceptual_roughness;
  m.metallic = metallic;
  m.reflectance = reflectance;
  return m;
}

// ------------------------------------------------------------------ RAII over the C handles
class SceneBuilder {  // the Prepare-stage host work: mesh -> BLAS, TLAS, emissives, alias tables
 public:
  SceneBuilder() { check(hk_scene_builder_create(&h_), "hk_scene_builder_create"); }
  ~SceneBuilder() { hk_scene_builder_destroy(h_); }
  SceneBuilder(const SceneBuilder&) = delete;
  SceneBuilder& operator=(const SceneBuilder&) = delete;
  uint32_t add_mesh(const std::vector<float>& positions, const std::vector<float>& normals, const std::vector<float>& uvs,
                    const std::vector<uint32_t>& indices, uint32_t topology = HK_TOPOLOGY_TRIANGLE_LIST) {
    uint32_t id = 0;
    check(hk_scene_builder_add_mesh(h_, positions.data(), normals.data(), uvs.data(), (uint32_t)(positions.size() / 3),
                                    indices.empty() ? nullptr : indices.data(), (uint32_t)indices.size(), topology, &id),
          "hk_scene_builder_add_mesh");
    return id;
  }
  uint32_t add_material(const HkMaterial& m) {
    uint32_t id = 0;
    check(hk_scene_builder_add_material(h_, &m, &id), "hk_scene_builder_add_material");
    return id;
  }
  uint32_t add_instance(uint32_t mesh, uint32_t material, const float transform[16]) {
    uint32_t id = 0;
    check(hk_scene_builder_add_instance(h_, mesh, material, transform, &id), "hk_scene_builder_add_instance");
    return id;
  }
  // move an instance (GlobalTransform change -> InstanceEvent::Modified, instance.rs:137-176); the next
  // finish() redoes the instance-level work only and records the old transform as the previous one
  void set_instance_transform(uint32_t instance, const float transform[16]) {
    check(hk_scene_builder_set_instance_transform(h_, instance, transform), "hk_scene_builder_set_instance_transform");
  }
  void finish() { check(hk_scene_builder_finish(h_), "hk_scene_builder_finish"); }
  const hk_scene_builder* handle() const { return h_; }
  hk_scene_builder* handle() { return h_; }

 private:
  hk_scene_builder* h_ = nullptr;
};

class Context {
 public:
  explicit Context(int device = 0, uint32_t flags = 0) { check(hk_create(device, flags, &c_), "hk_create"); }
  ~Context() { hk_destroy(c_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  hk_ctx* get() const { return c_; }
  void pass_run(uint32_t pass, uint32_t arg = 0) { check(hk_pass_run(c_, pass, arg, 0, 0), "hk_pass_run"); }
  std::vector<uint8_t> read(uint32_t buffer) const {
    uint32_t w, h, bpp;
    check(hk_buffer_info(c_, buffer, &w, &h, &bpp), "hk_buffer_info");
    std::vector<uint8_t> out((size_t)w * h * bpp);
    check(hk_read_buffer(c_, buffer, out.data(), out.size()), "hk_read_buffer");
    return out;
  }

 private:
  hk_ctx* c_ = nullptr;
};

// ------------------------------------------------------------------ the reference's nodes
struct PrepassNode {  // prepass.rs:736-852
  Context& ctx;
  void run(const HikariSettings& s) {
    check(hk_set_view_options(ctx.get(), (uint32_t)s.taa, s.upscale.kind, s.upscale.sharpness()), "hk_set_view_options");
    ctx.pass_run(HK_PASS_PREPASS);
  }
};
struct LightNode {  // light.rs:557-703
  static constexpr const char* IN_VIEW = "view";
  Context& ctx;
  void run(const HikariSettings& s) {
    ctx.pass_run(HK_PASS_FULL_SCREEN_ALBEDO);                                  // light.rs:646-653
    ctx.pass_run(HK_PASS_DIRECT_LIT);                                          // render[0], reservoirs (0,4)
    ctx.pass_run(HK_PASS_DIRECT_EMISSIVE);                                     // render[1], reservoirs (2,4)
    if (s.emissive_spatial_reuse) ctx.pass_run(HK_PASS_EMISSIVE_SPATIAL_REUSE);  // light.rs:675,689-697
    ctx.pass_run(HK_PASS_INDIRECT);                                            // render[2], reservoirs (6,8)
    if (s.indirect_spatial_reuse) ctx.pass_run(HK_PASS_INDIRECT_SPATIAL_REUSE);  // light.rs:676
  }
};
struct PostProcessNode {  // post_process.rs:1107-1312
  Context& ctx;
  // antialias = also the SMAA Tu4x / TAA dispatches (post_process.rs:1236-1272); the north-star frame ends at tone mapping
  void run(const HikariSettings& s, bool antialias = false) {
    if (s.denoise) {                                                  // post_process.rs:1190-1224
      const uint32_t channels = s.indirect_bounces == 0 ? 2u : 3u;    // post_process.rs:949-954
      for (uint32_t ch = 0; ch < channels; ++ch) {
        ctx.pass_run(HK_PASS_DEMODULATION, ch);
        for (uint32_t level = 0; level < 4; ++level) ctx.pass_run(HK_PASS_DENOISE_L0 + level, ch);
      }
    }
    ctx.pass_run(HK_PASS_TONE_MAPPING, s.denoise ? 1u : 0u);          // post_process.rs:1226-1234
    if (!antialias) return;
    if (s.upscale.kind == HK_UPSCALE_SMAA_TU4X) {                     // post_process.rs:1236-1258
      ctx.pass_run(HK_PASS_SMAA_TU4X);
      ctx.pass_run(HK_PASS_SMAA_TU4X_EXTRAPOLATE);
    }
    if (s.taa == Taa::Jasmine) ctx.pass_run(HK_PASS_TAA_JASMINE);     // post_process.rs:1260-1275
    if (s.upscale.kind == HK_UPSCALE_FSR1) {                          // post_process.rs:1277-1308
      ctx.pass_run(HK_PASS_FSR_EASU);
      ctx.pass_run(HK_PASS_FSR_RCAS);
    }
  }
};

// App::add_plugin(HikariPlugin): owns the context, uploads the noise tiles at start-up (lib.rs:189-219)
// and renders one camera with the "hikari" sub-graph order PREPASS -> LIGHT -> POST_PROCESS.
class HikariPlugin {
 public:
  HikariUniversalSettings universal_settings;
  HikariPlugin(const std::vector<uint8_t>& noise_rgba8_16x64x64, int device = 0, uint32_t flags = 0)
      : ctx_(device, flags), prepass_{ctx_}, light_{ctx_}, post_process_{ctx_} {
    check(hk_upload_noise(ctx_.get(), noise_rgba8_16x64x64.data(), noise_rgba8_16x64x64.size()), "hk_upload_noise");
  }
  Context& context() { return ctx_; }
  void set_scene(const SceneBuilder& b) { check(hk_upload_scene(ctx_.get(), b.handle()), "hk_upload_scene"); }
  // after SceneBuilder::set_instance_transform + finish: rewrite the instance-level device arrays only
  void update_instances(const SceneBuilder& b) { check(hk_upload_scene_instances(ctx_.get(), b.handle()), "hk_upload_scene_instances"); }
  // instance motion on the device: the poses set on `b` since the last upload / refit; returns how many moved
  uint32_t refit_instances(SceneBuilder& b) {
    uint32_t moved = 0;
    check(hk_refit_scene_instances(ctx_.get(), b.handle(), &moved), "hk_refit_scene_instances");
    return moved;
  }
  void rebuild_trees(uint32_t mode = HK_TREE_SAH) { check(hk_rebuild_scene_trees(ctx_.get(), mode), "hk_rebuild_scene_trees"); }

  // one frame of the camera's render graph; by_nodes = dispatch by dispatch through the three nodes,
  // otherwise one hk_frame_render call.  Returns the frame number used.
  size_t render(const Camera& camera, const HikariSettings& settings, std::optional<size_t> frame_number = std::nullopt, bool by_nodes = false,
                const HkLights* lights = nullptr, bool antialias = false) {
    if (camera.width != width_ || camera.height != height_ || settings.upscale.ratio() != ratio_) {  // light.rs:342-363
      check(hk_resize(ctx_.get(), camera.width, camera.height, settings.upscale.ratio()), "hk_resize");
      width_ = camera.width;
      height_ = camera.height;
      ratio_ = settings.upscale.ratio();
    }
    const size_t n = frame_number ? *frame_number : counter_.tick();
    const HkSettings sc = settings.to_c();
    HkFrame frame;
    check(hk_frame_from_settings(&sc, (uint32_t)n, &frame), "hk_frame_from_settings");  // view.rs:141-193
    const HkView view = camera.view_uniform();
    const HkPreviousView pview = previous_ ? previous_->previous_view_uniform() : camera.previous_view_uniform();
    const HkLights l = lights ? *lights : lights_uniform();
    if (by_nodes) {
      check(hk_frame_begin(ctx_.get(), &frame, &view, &pview, &l), "hk_frame_begin");
      prepass_.run(settings);
      light_.run(settings);
      post_process_.run(settings, antialias);
    } else {
      check(hk_frame_render(ctx_.get(), &frame, &view, &pview, &l, &sc, antialias ? HK_FRAME_ANTIALIAS : 0u), "hk_frame_render");
    }
    previous_ = camera;
    return n;
  }
  void wait() { check(hk_frame_wait(ctx_.get()), "hk_frame_wait"); }
  // the image OverlayNode presents (overlay.rs:226-231)
  static uint32_t final_buffer(const HikariSettings& s, bool antialias) {
    if (!antialias) return HK_BUF_TONE_MAPPED;
    if (s.upscale.kind == HK_UPSCALE_FSR1) return HK_BUF_UPSCALE_SHARPENED;  // upscale_output[1]
    return s.taa == Taa::Jasmine ? HK_BUF_TAA_OUTPUT : HK_BUF_UPSCALE_OUTPUT;
  }

 private:
  Context ctx_;
  PrepassNode prepass_;
  LightNode light_;
  PostProcessNode post_process_;
  FrameCounter counter_;
  uint32_t width_ = 0, height_ = 0;
  float ratio_ = 0.0f;
  bool balance_next_ = false, gather_ = false;
  std::optional<Camera> previous_;
};

// The same plugin over several GPUs of one node, still ONE process and one render thread (Bevy's model): hk_multi_* cuts the
// frame into one band per device, replicates the scene and moves the halo rows between neighbouring bands itself
// (light.rs:689-697 is where the spatial dispatches need their neighbours' temporal reservoirs).
class HikariMultiGpuPlugin {
 public:
  HikariMultiGpuPlugin(const std::vector<uint8_t>& noise_rgba8_16x64x64, const std::vector<int>& devices, uint32_t flags = 0) {
    check(hk_multi_create((uint32_t)devices.size(), devices.data(), flags, &m_), "hk_multi_create");
    const int rc = hk_multi_upload_noise(m_, noise_rgba8_16x64x64.data(), noise_rgba8_16x64x64.size());
    if (rc) { hk_multi_destroy(m_); m_ = nullptr; check(rc, "hk_multi_upload_noise"); }
  }
  ~HikariMultiGpuPlugin() { hk_multi_destroy(m_); }
  HikariMultiGpuPlugin(const HikariMultiGpuPlugin&) = delete;
  HikariMultiGpuPlugin& operator=(const HikariMultiGpuPlugin&) = delete;
  void set_scene(const SceneBuilder& b) { check(hk_multi_upload_scene(m_, b.handle()), "hk_multi_upload_scene"); }
  void update_instances(const SceneBuilder& b) { check(hk_multi_upload_scene_instances(m_, b.handle()), "hk_multi_upload_scene_instances"); }
  uint32_t refit_instances(SceneBuilder& b) {
    uint32_t moved = 0;
    check(hk_multi_refit_scene_instances(m_, b.handle(), &moved), "hk_multi_refit_scene_instances");
    return moved;
  }
  void rebuild_trees(uint32_t mode = HK_TREE_SAH) { check(hk_multi_rebuild_scene_trees(m_, mode), "hk_multi_rebuild_scene_trees"); }
  // rows of last frame's reservoirs fetched across the band borders before reprojection (0 for a static camera)
  void set_history_rows(uint32_t rows) { check(hk_multi_set_history_rows(m_, rows), "hk_multi_set_history_rows"); }
  // bands of unequal height: explicit boundaries (scaled render rows, bands + 1 entries; empty = the equal split) ...
  void set_band_bounds(const std::vector<uint32_t>& bounds) {
    check(hk_multi_set_band_bounds(m_, bounds.empty() ? nullptr : bounds.data(), (uint32_t)bounds.size()), "hk_multi_set_band_bounds");
  }
  // ... or the split by cost, derived from the NEXT rendered frame's primary rays and kept from then on (HK_FRAME_BALANCE_BANDS):
  // for the first frame or after a cut - rows that change owner lose their reservoir history
  void balance_bands_on_next_frame() { balance_next_ = true; }
  // after every frame band 0's device collects the other bands' rows of the image the overlay presents
  // (HK_FRAME_GATHER: peer copies on its stream); read_gathered() then reads that one context instead of merging on the host
  void set_gather(bool on) { gather_ = on; }
  std::vector<uint8_t> read_gathered(uint32_t buffer, uint32_t* w = nullptr, uint32_t* h = nullptr) const {
    hk_ctx* c0 = nullptr;
    check(hk_multi_context(m_, 0, &c0), "hk_multi_context");
    uint32_t bw, bh, bpp;
    check(hk_buffer_info(c0, buffer, &bw, &bh, &bpp), "hk_buffer_info");
    std::vector<uint8_t> out((size_t)bw * bh * bpp);
    check(hk_read_buffer(c0, buffer, out.data(), out.size()), "hk_read_buffer");
    if (w) *w = bw;
    if (h) *h = bh;
    return out;
  }
  std::vector<uint32_t> band_bounds(uint32_t bands) const {
    hk_ctx* c0 = nullptr;
    check(hk_multi_context(m_, 0, &c0), "hk_multi_context");
    std::vector<uint32_t> b(bands + 1u);
    check(hk_get_band_bounds(c0, b.data(), (uint32_t)b.size()), "hk_get_band_bounds");
    return b;
  }
  size_t render(const Camera& camera, const HikariSettings& settings, std::optional<size_t> frame_number = std::nullopt, const HkLights* lights = nullptr,
                bool antialias = false) {
    if (camera.width != width_ || camera.height != height_ || settings.upscale.ratio() != ratio_) {
      check(hk_multi_resize(m_, camera.width, camera.height, settings.upscale.ratio()), "hk_multi_resize");
      width_ = camera.width;
      height_ = camera.height;
      ratio_ = settings.upscale.ratio();
    }
    const size_t n = frame_number ? *frame_number : counter_.tick();
    const HkSettings sc = settings.to_c();
    HkFrame frame;
    check(hk_frame_from_settings(&sc, (uint32_t)n, &frame), "hk_frame_from_settings");
    const HkView view = camera.view_uniform();
    const HkPreviousView pview = previous_ ? previous_->previous_view_uniform() : camera.previous_view_uniform();
    const HkLights l = lights ? *lights : lights_uniform();
    check(hk_multi_frame_render(m_, &frame, &view, &pview, &l, &sc, (antialias ? HK_FRAME_ANTIALIAS : 0u) | (balance_next_ ? HK_FRAME_BALANCE_BANDS : 0u) | (gather_ ? HK_FRAME_GATHER : 0u)),
          "hk_multi_frame_render");
    balance_next_ = false;
    previous_ = camera;
    return n;
  }
  void wait() { check(hk_multi_wait(m_), "hk_multi_wait"); }
  // the union of the bands (every band's own rows)
  std::vector<uint8_t> read(uint32_t buffer, uint32_t* w = nullptr, uint32_t* h = nullptr) const {
    hk_ctx* c0 = nullptr;
    check(hk_multi_context(m_, 0, &c0), "hk_multi_context");
    uint32_t bw, bh, bpp;
    check(hk_buffer_info(c0, buffer, &bw, &bh, &bpp), "hk_buffer_info");
    std::vector<uint8_t> out((size_t)bw * bh * bpp);
    check(hk_multi_read_buffer(m_, buffer, out.data(), out.size()), "hk_multi_read_buffer");
    if (w) *w = bw;
    if (h) *h = bh;
    return out;
  }

 private:
  hk_multi* m_ = nullptr;
  FrameCounter counter_;
  uint32_t width_ = 0, height_ = 0;
  float ratio_ = 0.0f;
  bool balance_next_ = false, gather_ = false;
  std::optional<Camera> previous_;
};

}  // namespace hikari
