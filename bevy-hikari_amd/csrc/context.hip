// context.hip - the C ABI of libhikari_hip.so: context lifetime, screen-space resources, per-frame dispatch order.
//
// Reference call sites this file stands in for (cryscan/bevy-hikari v0.3.15):
//   resources      src/light.rs:307-383 (render/variance/albedo textures, 10 reservoir buffers),
//                  src/prepass.rs:285-318 (G-buffer), src/post_process.rs:621-633 (denoise textures)
//   dispatch order src/prepass.rs:769-852, src/light.rs:590-702, src/post_process.rs:1190-1234
//   ping-pong      src/light.rs:376,480-481,518-546
// (uploads and the scene layout: scene_layout.hip; instance motion on the device: scene_refit.hip; hk_ctx: hk_context.hpp)
#include "hk_context.hpp"

using namespace hk;
using namespace hkd;

namespace hk {

// logical size of a buffer: upscale_output is created at scale 2/ratio for SMAA Tu4x and taa_output at the
// scale in effect after the upscale match (post_process.rs:712-733): 2/ratio for SMAA Tu4x, 1/ratio for FSR1
void buffer_dims(const hk_ctx* c, uint32_t b, int* w, int* h) {
  if (buffer_is_full_size(b)) { *w = c->W; *h = c->H; return; }
  if (buffer_is_upscaled(b) && c->upscale_kind == HK_UPSCALE_SMAA_TU4X) { *w = c->UW; *h = c->UH; return; }
  if (b == HK_BUF_UPSCALE_OUTPUT) { *w = c->W; *h = c->H; return; }  // FSR1: upscale_output is created at scale 1.0 (post_process.rs:723)
  *w = c->RW;
  *h = c->RH;
}
size_t buffer_logical_bytes(const hk_ctx* c, uint32_t b) {
  int w, h;
  buffer_dims(c, b, &w, &h);
  return (size_t)w * h * buffer_bpp(b);
}

// The kernels take (k + 0.5) / size - pixel centre to uv, utils.wgsl:36-38, and the primary-ray NDC - through
// q = x * RN(1/size) plus one exact-residual correction (hk_device.hpp div_by) when this returns true: every
// numerator the kernels can form (pixel coordinates, spiral taps up to 20 px and a-trous taps up to 8 px beyond
// the edge) is compared with the IEEE quotient here, with the same three operations the device executes.
bool certify_uv_division(int size) {
  const float b = (float)size, c = 1.0f / b;
  for (int k = -64; k < size + 64; ++k) {
    const float x = (float)k + 0.5f, want = x / b, q = x * c, got = fmaf(fmaf(-q, b, x), c, q);
    if (memcmp(&want, &got, 4) != 0) return false;
  }
  return true;
}

// The main stream - primary rays, the indirect pass, spatial reuse: the frame's dependent chain - runs at the device's highest stream
// priority when the context's frames are small, at the default priority otherwise (round 6; measured on three scenes x four sizes and
// on bands, profiles/r06_stream_priority_ab.txt): with the chain ahead of the side stream's direct-light dispatches and the post
// stream's a-trous levels whenever both have workgroups waiting, frames of up to 2560 x 1440 are 1-6 % shorter (Cornell 1080p -1.1 %,
// config 3 -4.0 %, bands of any frame 0 ... -4 %); a whole 3840 x 2160 frame is 0.3-1.3 % LONGER that way, so it keeps the default.
// A stream's priority is fixed when it is created, so the choice is made ONCE, at the context's first frame - from the pixels it
// dispatches per frame: its size and its band are known by then; its stream, created with the context for the uploads that come
// before, is created again at the other priority if need be - and kept: a second main
// stream beside the first cost config 4 2.5 % (five streams per context: two share a hardware queue), and a stream created again
// and again ends up on a queue it shares too (the same file: 0.92 -> 1.09 ms after three changes) - so later resizes and hk_set_band do
// not revisit it.  (The device has few high-priority queues: the first contexts of a process get the benefit, a fifth one measured none.)  hk_debug_set_option(HK_DEBUG_OPT_MAIN_PRIORITY) forces a change (A/B and tests).
#ifndef HK_PRE_BANDS
#define HK_PRE_BANDS 1   // bands pipeline their primary rays too: a band of config 3 / 4 (an eighth of the frame) -11.5 % / -8.7 ... -12.4 % (profiles/r06_prepass_pipeline_ab.txt)
#endif
#ifndef HK_PRE_STREAM_HIGH
#define HK_PRE_STREAM_HIGH 0   // the primary rays' own stream: the default priority (at the chain's: config 3 -1.8 % instead of -2.9 %, profiles/r06_prepass_pipeline_ab.txt)
#endif
#ifndef HK_MAIN_HIGH_PRIORITY_PIXELS
#define HK_MAIN_HIGH_PRIORITY_PIXELS ((size_t)6 << 20)
#endif
int pick_main_stream(hk_ctx* c, bool forced) {
  if (!c->own_stream || c->post_forked) return HK_OK;
  if (!forced && c->main_priority_decided) return HK_OK;
  const size_t px = (size_t)c->RW * (size_t)c->RH / (size_t)(c->band_count > 0 ? c->band_count : 1);
  if (px == 0) return HK_OK;
  c->main_priority_decided = true;
  const bool high = c->main_priority < 0 ? px <= HK_MAIN_HIGH_PRIORITY_PIXELS : c->main_priority != 0;
  if (high == c->own_stream_high) return HK_OK;
  const bool in_use = c->stream == c->own_stream;   // (else the caller's own stream is: hk_set_stream)
  { const int rc = sync_all(c); if (rc) return rc; }
  HK_HIP(hipStreamSynchronize(c->own_stream));
  // (a device or runtime without stream priorities keeps the stream it has: the priority is an optimisation, not a requirement)
  int least = 0, greatest = 0;
  hipStream_t fresh = nullptr;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest ||
      hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, high ? greatest : least) != hipSuccess || !fresh) {
    (void)hipGetLastError();
    return HK_OK;
  }
  (void)hipStreamDestroy(c->own_stream);
  c->own_stream = fresh;
  c->own_stream_high = high;
  if (in_use) c->stream = fresh;
  // (with the chain in the high-priority pool a fourth default-priority stream has a hardware queue of its own: the primary rays')
  if (high && c->side_stream && c->post_stream && !c->pre_stream) {
    // (a band's context also holds the two communicator lanes: the default-priority pool is full, the primary rays join the chain's pool)
    if (hipStreamCreateWithPriority(&c->pre_stream, hipStreamNonBlocking, (HK_PRE_STREAM_HIGH || c->band_count > 1) ? greatest : least) != hipSuccess || hipEventCreateWithFlags(&c->pre_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->pre_scene_mark, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (c->pre_stream) (void)hipStreamDestroy(c->pre_stream);
      c->pre_stream = nullptr;
    }
  }
  return HK_OK;
}
#ifndef HK_PREPASS_PIPELINE_RULE
// scenes beyond the LDS copy (long primary rays, trace stages with ends to fill) in frames of up to 3 Mi pixels: measured on configs 3 / 4
// at 720p / 1080p -7.5 / -3.9 / -9.0 % per frame, at 2560 x 1440 0 / +1.4 %; the Cornell frame (its primary rays: 0.066 of 0.88 ms, walked
// from LDS) +5 % with the stream at the default priority and 0 at the chain's - profiles/r06_prepass_pipeline_ab.txt
// (Scenes walked from LDS gain only where a band leaves most of the chip idle - Cornell 1080p in 8 / 4 bands -9 % / -4 % per band frame, in
// 2 bands 0 ... +1.5 %, the whole frame 0 ... +5.7 % - and are NOT pipelined by the rule: HK_PREPASS_PIPELINE_LDS_BANDS.  One run of ~160 of
// the band test on such a scene showed 11 records of an indirect spatial reservoir differing from the single context's - every rendered
// plane equal, not reproduced since, with the pipelining or without: NOTES "open".  hk_debug_set_option(.., 1) pipelines them.)
#ifndef HK_PREPASS_PIPELINE_LDS_BANDS
#define HK_PREPASS_PIPELINE_LDS_BANDS 0
#endif
#define HK_PREPASS_PIPELINE_RULE                                                                                                     \
  ((size_t)c->scene.blob_f4 * 16 > HK_LDS_SCENE_BYTES ? (size_t)c->RW * (size_t)c->RH / (size_t)(c->band_count > 0 ? c->band_count : 1) <= ((size_t)3 << 20) \
                                                        : (HK_PREPASS_PIPELINE_LDS_BANDS && c->band_count > 1 && (size_t)c->RW * (size_t)c->RH / (size_t)c->band_count <= ((size_t)3 << 18)))
#endif
int free_screen(hk_ctx* c) {
  for (int k = 0; k < 3; ++k) {
    if (c->det_winner[k]) (void)hipFree(c->det_winner[k]);
    c->det_winner[k] = nullptr;
    c->det_lite_clean[k] = false;
  }
  for (uint32_t b = 0; b < HK_BUF_COUNT; ++b) {
    if (c->buf[b]) (void)hipFree(c->buf[b]);
    c->buf[b] = nullptr;
    c->buf_bytes[b] = 0;
  }
  for (int k = 0; k < 10; ++k) {
    if (c->tile_meta[k]) (void)hipFree(c->tile_meta[k]);
    c->tile_meta[k] = nullptr;
    c->tile_meta_zero[k] = false;
  }
  if (c->wf_mem) (void)hipFree(c->wf_mem);
  c->wf_mem = nullptr;
  if (c->wf_paths_mem) (void)hipFree(c->wf_paths_mem);
  c->wf_paths_mem = nullptr;
  if (c->wf.timeline) (void)hipFree(c->wf.timeline);
  c->wf = hkd::WfBuffers{};
  if (c->depth_plane) (void)hipFree(c->depth_plane);
  if (c->prev_depth_plane) (void)hipFree(c->prev_depth_plane);
  c->prev_depth_plane = nullptr;
  if (c->dn_g) (void)hipFree(c->dn_g);
  c->depth_plane = nullptr;
  c->dn_g = nullptr;
  for (void** q : {&c->albedo_twin, &c->depth_gradient_twin, &c->dn_g_twin, &c->normal_twin, &c->instance_material_twin, &c->render_twin[0], &c->render_twin[1], &c->render_twin[2], &c->variance_twin[0],
                   &c->variance_twin[1], &c->variance_twin[2]}) {
    if (*q) (void)hipFree(*q);
    *q = nullptr;
  }
  c->post_pending[0] = c->post_pending[1] = false;
  for (int k = 0; k < 2; ++k) {
    for (int l = 0; l < 4; ++l) {
      if (c->dn_extra[k][l]) (void)hipFree(c->dn_extra[k][l]);
      c->dn_extra[k][l] = nullptr;
    }
    if (c->dn_extra_var[k]) (void)hipFree(c->dn_extra_var[k]);
    c->dn_extra_var[k] = nullptr;
  }
  return HK_OK;
}

hipEvent_t get_event(hk_ctx* c) {
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

// resolve finished timed launches into the per-slot accumulators (stream must be idle)
void drain_timers(hk_ctx* c) {
  for (TimedLaunch& t : c->pending) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      c->slot_ms[t.slot] += (double)ms;
      c->slot_launches[t.slot] += 1;
    }
    c->event_pool.push_back(t.start);
    c->event_pool.push_back(t.stop);
  }
  c->pending.clear();
  if (c->frame_timed) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->frame_start, c->frame_stop) == hipSuccess) c->last_frame_ms = ms;
    c->frame_timed = false;
  }
}

struct ScopedTimer {
  hk_ctx* c;
  bool on;
  TimedLaunch t{};
  bool attached;  // the launcher attaches start / stop to the dispatch (hipExtLaunchKernel) instead of recording them around it
  ScopedTimer(hk_ctx* ctx, uint32_t slot, bool attach = false) : c(ctx), on((ctx->timing_mask >> slot) & 1u), attached(attach) {
    if (on) {
      t.slot = slot;
      t.start = get_event(c);
      t.stop = get_event(c);
      if (!attached) (void)hipEventRecord(t.start, c->stream);
    }
  }
  ~ScopedTimer() {
    if (on) {
      if (!attached) (void)hipEventRecord(t.stop, c->stream);
      c->pending.push_back(t);
    }
  }
};

DFrame make_dframe(const hk_ctx* c) {
  DFrame f;
  memset(&f, 0, sizeof(f));
  const HkFrame& h = c->frame;
  for (int col = 0; col < 3; ++col)
    for (int row = 0; row < 3; ++row) f.kernel[col * 3 + row] = h.kernel[col][row];
  f.number = h.number;
  f.direct_validate_interval = h.direct_validate_interval;
  f.emissive_validate_interval = h.emissive_validate_interval;
  f.indirect_bounces = h.indirect_bounces;
  f.temporal_reuse = h.temporal_reuse;
  f.max_temporal_reuse_count = h.max_temporal_reuse_count;
  f.max_spatial_reuse_count = h.max_spatial_reuse_count;
  f.max_reservoir_lifetime = h.max_reservoir_lifetime;
  f.solar_angle = h.solar_angle;
  f.max_indirect_luminance = h.max_indirect_luminance;
  f.upscale_ratio = h.upscale_ratio;
  f.random_float_number = (float)hash_u32(h.number) / 4294967295.0f;
  f.number_golden = (float)h.number * 1.618033989f;
  f.cam_x = c->view.world_position[0]; f.cam_y = c->view.world_position[1]; f.cam_z = c->view.world_position[2];
  f.ortho_x = c->view.view_proj[2]; f.ortho_y = c->view.view_proj[6]; f.ortho_z = c->view.view_proj[10];
  f.is_ortho = c->view.projection[15] == 1.0f ? 1u : 0u;
  f.sun_r = c->lights.directional_color[0]; f.sun_g = c->lights.directional_color[1]; f.sun_b = c->lights.directional_color[2];
  f.sun_dx = c->lights.direction_to_light[0]; f.sun_dy = c->lights.direction_to_light[1]; f.sun_dz = c->lights.direction_to_light[2];
  f.amb_r = c->lights.ambient_color[0]; f.amb_g = c->lights.ambient_color[1]; f.amb_b = c->lights.ambient_color[2];
  f.clear_r = h.clear_color[0]; f.clear_g = h.clear_color[1]; f.clear_b = h.clear_color[2]; f.clear_a = h.clear_color[3];
  f.dw = c->W; f.dh = c->H; f.rw = c->RW; f.rh = c->RH;
  f.inv_dw = 1.0f / (float)c->W; f.inv_dh = 1.0f / (float)c->H; f.inv_rw = 1.0f / (float)c->RW; f.inv_rh = 1.0f / (float)c->RH;
  f.rcp_rw = 1.0 / (double)c->RW; f.rcp_rh = 1.0 / (double)c->RH;
  f.uv_fast = c->uv_fast ? 1u : 0u;
  return f;
}
GBuffer make_gbuffer(const hk_ctx* c) {
  GBuffer g;
  g.position = (float4*)c->buf[HK_BUF_POSITION];
  g.normal = (uint32_t*)c->buf[HK_BUF_NORMAL];
  g.depth_gradient = (float2*)c->buf[HK_BUF_DEPTH_GRADIENT];
  g.instance_material = (float2*)c->buf[HK_BUF_INSTANCE_MATERIAL];
  g.velocity_uv = (float4*)c->buf[HK_BUF_VELOCITY_UV];
  g.depth = c->depth_plane;
  g.dn_g = (float4*)c->dn_g;
  g.albedo_out = nullptr;
  return g;
}
// A band of a sharded frame whose history halo is not empty parks the scatter stores of its temporal dispatches instead of
// racing them into its local copy of previous_spatial: the neighbours need them (and this band theirs) before spatial_reuse
bool parks_across_bands(const hk_ctx* c) { return c->band_count > 1 && c->history_now > 0; }
// the light deterministic form (hk_kernels.hpp LightTargets::det_lite) for this channel's temporal dispatch: a single context, by default
// where the channel's previous_spatial buffer has a reader, for every channel with HK_CTX_DETERMINISTIC_SCATTER, never with
// HK_CTX_RACING_SCATTER.  (A band with a history halo parks in the full form; a band without one - static view - has nothing to resolve.)
bool scatter_lite(const hk_ctx* c, int channel) {
  if (c->band_count > 1 || (c->flags & HK_CTX_RACING_SCATTER)) return false;
  if (c->flags & HK_CTX_DETERMINISTIC_SCATTER) return true;
  return channel == 2 ? c->frame.indirect_spatial_reuse != 0u : c->frame.emissive_spatial_reuse != 0u;
}
// the parked planes of the channels in `channels` (bit k = channel k), on first use: 72 B per render pixel and channel.  (A single
// context in the default mode parks for the channels whose previous_spatial buffer has a reader only - the indirect channel with the
// default settings: 0.6 GB at 3840 x 2160, not 1.8.)
int ensure_parked(hk_ctx* c, uint32_t channels) {
  const size_t nr = (size_t)c->RW * c->RH;
  for (int k = 0; k < 3; ++k) {
    if (!((channels >> k) & 1u) || c->det_winner[k]) continue;
    // the channel's three planes or none: allocated into locals and committed together, so that a failure half way leaves the context
    // as it was (det_winner[k] is what says "the channel's planes exist")
    int* winner = nullptr;
    void* plane[2] = {nullptr, nullptr};
    size_t plane_bytes[2] = {0, 0};
    hipError_t e = hipMalloc((void**)&winner, nr * sizeof(int));
    for (int j = 0; j < 2 && e == hipSuccess; ++j) {
      const uint32_t b = (j == 0 ? (uint32_t)HK_BUF_PARKED_TO0 : (uint32_t)HK_BUF_PARKED_RECORD0) + (uint32_t)k;
      plane_bytes[j] = nr * buffer_bpp(b);
      e = hipMalloc(&plane[j], plane_bytes[j]);
      // nothing parked.  (Blocking, once: the first use may come from any of the context's streams - the side stream's direct-light
      // dispatch - while another stream is about to use ITS channel's planes)
      if (e == hipSuccess) e = hipMemset(plane[j], j == 0 ? 0xFF : 0, plane_bytes[j]);
    }
    if (e != hipSuccess) {
      if (winner) (void)hipFree(winner);
      for (void* q : plane)
        if (q) (void)hipFree(q);
      set_error("allocating the parked-store planes failed: %s", hipGetErrorString(e));
      return HK_E_HIP;
    }
    c->det_winner[k] = winner;
    for (int j = 0; j < 2; ++j) {
      const uint32_t b = (j == 0 ? (uint32_t)HK_BUF_PARKED_TO0 : (uint32_t)HK_BUF_PARKED_RECORD0) + (uint32_t)k;
      c->buf[b] = plane[j];
      c->buf_bytes[b] = plane_bytes[j];
    }
  }
  return HK_OK;
}
// world bounds of the scene = the union of the instances' boxes (the host mirrors follow device refits: refit_impl)
void scene_bounds(const hk_ctx* c, float mn[3], float mx[3]) {
  for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
  for (const HkInstance& in : c->instances)
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(mn[k], in.min[k]);
      mx[k] = std::max(mx[k], in.max[k]);
    }
}
// hk_history_rows_bound on this context's frame: the view pair, the scene's bounds and every instance whose previous model differs
int derive_history_rows(const hk_ctx* c, uint32_t* rows) {
  *rows = 0;
  if (c->instances.empty()) return HK_OK;
  std::vector<HkMovedBox> moved;
  const size_t ni = c->instances.size();
  if (c->prev_models.size() == 16 * ni)
    for (size_t i = 0; i < ni; ++i) {
      const HkInstance& in = c->instances[i];
      const float* pm = &c->prev_models[16 * i];
      if (memcmp(pm, in.model, 64) == 0) continue;
      HkMovedBox b{};
      memcpy(b.min, in.min, 12);
      memcpy(b.max, in.max, 12);
      // previous_model x model^-1; model^-1 = transpose(inverse_transpose_model); column-major: out[col][row]
      for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) {
          double s = 0.0;
          for (int k = 0; k < 4; ++k) s += (double)pm[4 * k + row] * (double)in.inverse_transpose_model[4 * k + col];  // inv[k][col] = itm[col][k] -> itm column k, row col
          b.previous_from_current[4 * col + row] = (float)s;
        }
      moved.push_back(b);
    }
  if (moved.empty() && memcmp(c->view.view_proj, c->pview.view_proj, 64) == 0) return HK_OK;
  float mn[3], mx[3];
  scene_bounds(c, mn, mx);
  return hk_history_rows_bound(&c->view, &c->pview, (uint32_t)c->RH, mn, mx, moved.empty() ? nullptr : moved.data(), (uint32_t)moved.size(), rows);
}
// group 6 ping-pong, light.rs:376,480-481,518-546
LightTargets make_light_targets(const hk_ctx* c, int channel) {
  static const int T[3] = {0, 2, 6}, S[3] = {4, 4, 8};
  const uint32_t cur = c->frame.number % 2u, prev = 1u - cur;
  LightTargets t;
  t.previous = (const PackedReservoir*)c->buf[HK_BUF_RESERVOIR0 + cur + T[channel]];
  t.current = (PackedReservoir*)c->buf[HK_BUF_RESERVOIR0 + prev + T[channel]];
  t.previous_spatial = (PackedReservoir*)c->buf[HK_BUF_RESERVOIR0 + cur + S[channel]];
  t.spatial = (PackedReservoir*)c->buf[HK_BUF_RESERVOIR0 + prev + S[channel]];
  t.variance = (float*)c->buf[HK_BUF_VARIANCE0 + channel];
  t.render = (uint2*)c->buf[HK_BUF_RENDER0 + channel];
  // parked scatter stores: the verification mode, and a band of a sharded frame with a history halo
  const bool parked = c->det_winner[channel] && (scatter_lite(c, channel) || parks_across_bands(c));
  t.det_lite = parked && !parks_across_bands(c) ? 1 : 0;
  t.det_winner = parked ? c->det_winner[channel] : nullptr;
  t.det_to = parked ? (int*)c->buf[HK_BUF_PARKED_TO0 + channel] : nullptr;
  t.det_pending = parked ? (PackedReservoir*)c->buf[HK_BUF_PARKED_RECORD0 + channel] : nullptr;
  t.m_current = t.m_spatial = t.m_previous_spatial = nullptr;
  t.serial = 0;
  t.tiles_x = c->tiles_x;
  t.rw = c->RW;
  return t;
}
// Attach the tile records of the dispatch's reservoir targets (uniform-tile store elision), or - when this dispatch cannot
// maintain them (a band of a sharded frame, a row range that does not start on a tile row) - declare the tiles of the buffers
// it writes unknown.  `spatial_pass`: spatial_reuse reads `current` and writes `spatial` only.
int attach_tile_meta(hk_ctx* c, LightTargets& t, int channel, bool spatial_pass, int y0, int y1) {
  if (!c->tile_meta[0]) return HK_OK;
  static const int T[3] = {0, 2, 6}, S[3] = {4, 4, 8};
  const int cur = (int)(c->frame.number % 2u), prev = 1 - cur;
  const int k_current = prev + T[channel], k_spatial = prev + S[channel], k_previous_spatial = cur + S[channel];
  const bool maintain = c->band_count == 1 && (y0 % 8) == 0 && ((y1 % 8) == 0 || y1 == c->RH);  // whole tiles only
  const size_t mb = (size_t)c->tiles_x * c->tiles_y * sizeof(TileMeta);
  const int written[3] = {spatial_pass ? -1 : k_current, k_spatial, spatial_pass ? -1 : k_previous_spatial};
  for (int k : written) {
    if (k < 0) continue;
    if (maintain) {
      c->tile_meta_zero[k] = false;
    } else if (!c->tile_meta_zero[k]) {
      HK_HIP(hipMemsetAsync(c->tile_meta[k], 0, mb, c->stream));
      c->tile_meta_zero[k] = true;
    }
  }
  if (!maintain) return HK_OK;
  t.m_current = c->tile_meta[k_current];
  t.m_spatial = c->tile_meta[k_spatial];
  t.m_previous_spatial = c->tile_meta[k_previous_spatial];
  t.serial = ++c->elide_serial;
  return HK_OK;
}

int ready(hk_ctx* c) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  HK_REQUIRE(c->W > 0, HK_E_NOT_READY, "hk_resize has not been called");
  HK_REQUIRE(c->have_frame, HK_E_NOT_READY, "hk_frame_begin has not been called");
  HK_REQUIRE(c->have_noise, HK_E_NOT_READY, "noise textures not uploaded");
  HK_HIP(hipSetDevice(c->device));
  return finalize_scene(c);
}

struct Jitter { float x, y; };
Jitter prepass_jitter(const hk_ctx* c) {  // prepass.wgsl:30-38,52-54,71
  Jitter j{0.0f, 0.0f};
  if (c->taa == HK_TAA_NONE) return j;
  const uint32_t n = c->frame.number;
  const uint32_t index = (c->upscale_kind == HK_UPSCALE_SMAA_TU4X) ? ((n >> 1u) & 15u) : (n & 15u);
  const float* h = c->frame.halton[index >> 1u];
  const float hx = (index & 1u) == 0u ? h[0] : h[2], hy = (index & 1u) == 0u ? h[1] : h[3];
  j.x = 2.0f * hx * (1.0f / c->view.viewport[2]);
  j.y = -(2.0f * hy * (1.0f / c->view.viewport[3]));
  return j;
}

void full_rows_for(const hk_ctx* c, int ry0, int ry1, int* fy0, int* fy1) {
  if (c->RH == c->H) {
    *fy0 = ry0;
    *fy1 = ry1;
    return;
  }
  *fy0 = std::max(0, (int)floorf((float)ry0 * (float)c->H / (float)c->RH) - 1);
  *fy1 = std::min(c->H, (int)ceilf((float)ry1 * (float)c->H / (float)c->RH) + 1);
}

// All denoised channels of one step in a single launch (kernels_denoise.hip).  Channel ch's internal
// textures: the exposed HK_BUF_DENOISE_INTERNAL* for the last channel, private sets for the others.
void* dn_internal(hk_ctx* c, uint32_t nch, uint32_t ch, int level) {
  return ch + 1 == nch ? c->buf[HK_BUF_DENOISE_INTERNAL0 + level] : c->dn_extra[ch][level];
}
float* dn_variance(hk_ctx* c, uint32_t nch, uint32_t ch) {
  return ch + 1 == nch ? (float*)c->buf[HK_BUF_DENOISE_INTERNAL_VARIANCE] : c->dn_extra_var[ch];
}
int run_demodulation_fused(hk_ctx* c, uint32_t nch, int y0, int y1) {
  if (y1 <= y0) return HK_OK;
  const DFrame fr = make_dframe(c);
  ScopedTimer timer(c, HK_PASS_DEMODULATION);
  DemodTargets d{};
  d.albedo = (const uint2*)c->buf[HK_BUF_ALBEDO];
  for (uint32_t ch = 0; ch < nch; ++ch) {
    d.variance[ch] = (const float*)c->buf[HK_BUF_VARIANCE0 + ch];
    d.render[ch] = (const uint2*)c->buf[HK_BUF_RENDER0 + ch];
    d.output[ch] = (uint2*)dn_internal(c, nch, ch, 0);
    d.internal_variance[ch] = dn_variance(c, nch, ch);
  }
  launch_demodulation(c->stream, (int)nch, fr, d, y0, y1);
  HK_HIP(hipGetLastError());
  return HK_OK;
}
int run_denoise_fused(hk_ctx* c, uint32_t nch, int level, int y0, int y1, bool with_tone_mapping = false) {
  if (y1 <= y0) return HK_OK;
  const DFrame fr = make_dframe(c);
  ScopedTimer timer(c, HK_PASS_DENOISE_L0 + (uint32_t)level);
  DenoiseTargets d{};
  d.albedo = (const uint2*)c->buf[HK_BUF_ALBEDO];
  d.dn_g = (const float4*)c->dn_g;
  d.depth = c->depth_plane;
  d.depth_gradient = (const float2*)c->buf[HK_BUF_DEPTH_GRADIENT];
  for (uint32_t ch = 0; ch < nch; ++ch) {
    d.input[ch] = (const uint2*)dn_internal(c, nch, ch, level);
    d.output[ch] = level == 3 ? (uint2*)c->buf[HK_BUF_DENOISE_RENDER0 + ch] : (uint2*)dn_internal(c, nch, ch, level + 1);
    d.internal_variance[ch] = dn_variance(c, nch, ch);
  }
  if (with_tone_mapping && level == 3 && nch == (c->frame.indirect_bounces != 0u ? 3u : 2u)) {  // the channels tone_mapping sums, post_process.rs:941-954
    d.tone_mapped = (uint2*)c->buf[HK_BUF_TONE_MAPPED];
    memcpy(d.clear_color, c->frame.clear_color, sizeof(d.clear_color));
  }
  launch_denoise(c->stream, level, (int)nch, 0, fr, d, y0, y1);
  HK_HIP(hipGetLastError());
  return HK_OK;
}

// Which schedule indirect_lit_ambient takes (hikari_hip.h HK_CTX_WAVEFRONT).  The counting kernels exist in the fused form only.
bool use_wavefront(const hk_ctx* c) {
  if (c->frame.indirect_bounces < 2u || c->frame.indirect_bounces > 60u || (c->flags & HK_CTX_COUNT_RAYS)) return false;
  if (c->flags & HK_CTX_FUSED_INDIRECT) return false;
  if (c->flags & HK_CTX_WAVEFRONT) return true;
  return (size_t)c->scene.blob_f4 * 16 > HK_LDS_SCENE_BYTES;
}
// The wide walk (hk_wide.hpp) is the closest-hit walk of scenes beyond LDS in the product default: the trace stages
// (k_wf_trace_wide) and the primary rays (k_prepass<*, 4>).  The reference's order (HK_CTX_EXACT_TRAVERSAL) keeps the skip-link walk;
// HK_CTX_NO_WIDE_WALK is the A/B switch.
// (a scene that fits the LDS copy is walked from LDS by every kernel: no records are built for it)
static bool wide_allowed(const hk_ctx* c) {
  return !(c->flags & HK_CTX_NO_WIDE_WALK) && c->threaded && !c->scene.flat_mode && (size_t)c->scene.blob_f4 * 16 > HK_LDS_SCENE_BYTES;
}
bool use_wide(const hk_ctx* c) { return wide_allowed(c); }
// records of the trees the next trace stages walk, (re)derived from what the scene blob holds now
int ensure_wide(hk_ctx* c, bool with_spill) {
  const size_t tlas_slots = c->instance_nodes.size(), blas_slots = c->asset_nodes.size();
  if (tlas_slots == 0 || blas_slots == 0) return HK_OK;
  if (tlas_slots > c->wide_tlas_slots) {
    if (c->wide_tlas) { HK_HIP(hipStreamSynchronize(c->stream)); (void)hipFree(c->wide_tlas); c->wide_tlas = nullptr; }
    HK_HIP(hipMalloc((void**)&c->wide_tlas, (tlas_slots + tlas_slots / 2 + 16) * 128));
    c->wide_tlas_slots = tlas_slots + tlas_slots / 2 + 16;
    c->wide_tlas_dirty = true;
  }
  if (blas_slots > c->wide_blas_slots) {
    if (c->wide_blas) { HK_HIP(hipStreamSynchronize(c->stream)); (void)hipFree(c->wide_blas); c->wide_blas = nullptr; }
    HK_HIP(hipMalloc((void**)&c->wide_blas, blas_slots * 128));
    c->wide_blas_slots = blas_slots;
    c->wide_blas_dirty = true;
  }
  // the leaves' ranks (hk_kernels.hpp WideTrees): one u32 per instance / per primitive, written with the records
  const size_t n_inst = c->instances.size(), n_prim = c->primitives.size();
  if (n_inst > c->wide_rank_instances) {
    if (c->wide_tlas_rank) { HK_HIP(hipStreamSynchronize(c->stream)); (void)hipFree(c->wide_tlas_rank); c->wide_tlas_rank = nullptr; }
    HK_HIP(hipMalloc((void**)&c->wide_tlas_rank, (n_inst + n_inst / 2 + 16) * sizeof(uint32_t)));
    HK_HIP(hipMemsetAsync(c->wide_tlas_rank, 0xFF, (n_inst + n_inst / 2 + 16) * sizeof(uint32_t), c->stream));
    c->wide_rank_instances = n_inst + n_inst / 2 + 16;
    c->wide_tlas_dirty = true;
  }
  if (n_prim > c->wide_rank_primitives) {
    if (c->wide_blas_rank) { HK_HIP(hipStreamSynchronize(c->stream)); (void)hipFree(c->wide_blas_rank); c->wide_blas_rank = nullptr; }
    HK_HIP(hipMalloc((void**)&c->wide_blas_rank, std::max<size_t>(n_prim, 1) * sizeof(uint32_t)));
    HK_HIP(hipMemsetAsync(c->wide_blas_rank, 0xFF, std::max<size_t>(n_prim, 1) * sizeof(uint32_t), c->stream));  // (a primitive no mesh tree names: a rank that is at least deterministic)
    c->wide_rank_primitives = n_prim;
    c->wide_blas_dirty = true;
  }
  if (!c->compute_units) {
    hipDeviceProp_t prop;
    HK_HIP(hipGetDeviceProperties(&prop, c->device));
    c->compute_units = prop.multiProcessorCount;
  }
  const size_t lanes = with_spill ? wide_trace_lanes(c->compute_units) : 0;  // (the trace kernel's own launch size: kernels_wavefront.hip)
  if (lanes > c->wide_spill_lanes) {
    if (c->wide_spill) { HK_HIP(hipStreamSynchronize(c->stream)); (void)hipFree(c->wide_spill); c->wide_spill = nullptr; }
    HK_HIP(hipMalloc((void**)&c->wide_spill, lanes * wide_spill_entries() * sizeof(uint32_t)));
    c->wide_spill_lanes = lanes;
  }
  if (c->wide_blas_dirty || c->wide_mesh_check) {
    // one launch per mesh tree (links are local to a tree): all of them after a mesh-level build, and after a change of the instance
    // SET the ones no instance used before (a mesh uploaded ahead of its first instance has no records until then)
    if (c->wide_blas_dirty) c->wide_meshes.clear();
    std::vector<std::pair<uint32_t, uint32_t>> meshes;
    // (node_offset, node_count) -> the mesh's first primitive: a triangle leaf's id is local to its mesh.  Two instances that name the
    // same tree must name the same primitives - otherwise the second mesh's ranks would land at the wrong base
    std::vector<std::pair<std::pair<uint32_t, uint32_t>, uint32_t>> first_primitive;
    for (const HkInstance& in : c->instances) {
      meshes.emplace_back(in.mesh.node_offset, in.mesh.node_count);
      first_primitive.emplace_back(std::make_pair(in.mesh.node_offset, in.mesh.node_count), in.mesh.primitive);
    }
    std::sort(meshes.begin(), meshes.end());
    meshes.erase(std::unique(meshes.begin(), meshes.end()), meshes.end());
    std::sort(first_primitive.begin(), first_primitive.end());
    first_primitive.erase(std::unique(first_primitive.begin(), first_primitive.end()), first_primitive.end());
    for (size_t k = 0; k + 1 < first_primitive.size(); ++k)
      HK_REQUIRE(first_primitive[k].first != first_primitive[k + 1].first, HK_E_INVALID, "two instances share the mesh nodes [%u, +%u) but not the mesh primitives (%u / %u)",
                 first_primitive[k].first.first, first_primitive[k].first.second, first_primitive[k].second, first_primitive[k + 1].second);
    for (const auto& m : meshes) {
      if (std::binary_search(c->wide_meshes.begin(), c->wide_meshes.end(), m)) continue;
      HK_REQUIRE((size_t)m.first + m.second <= blas_slots, HK_E_INVALID, "an instance's mesh nodes lie outside the uploaded mesh nodes");
      const uint32_t prim0 = std::lower_bound(first_primitive.begin(), first_primitive.end(), std::make_pair(m, 0u))->second;
      // (a tree of L leaves has 3 L - 2 nodes: its leaf ids stay below (node_count + 2) / 3)
      HK_REQUIRE((size_t)prim0 + (m.second + 2u) / 3u <= n_prim, HK_E_INVALID, "an instance's mesh primitives lie outside the uploaded primitives");
      launch_build_wide(c->stream, c->scene.nodes + 2u * ((size_t)c->scene.blas_base + m.first), m.second, c->wide_blas + 8u * (size_t)m.first, c->wide_blas_rank + prim0);
    }
    HK_HIP(hipGetLastError());
    std::vector<std::pair<uint32_t, uint32_t>> all;
    std::set_union(c->wide_meshes.begin(), c->wide_meshes.end(), meshes.begin(), meshes.end(), std::back_inserter(all));
    c->wide_meshes.swap(all);
    c->wide_blas_dirty = c->wide_mesh_check = false;
  }
  if (c->wide_tlas_dirty) {
    launch_build_wide(c->stream, c->scene.nodes, (uint32_t)tlas_slots, c->wide_tlas, c->wide_tlas_rank);
    HK_HIP(hipGetLastError());
    c->wide_tlas_dirty = false;
  }
  return HK_OK;
}
// the records for a fused kernel's walks (the primary rays): scenes in global memory with threaded trees, i.e. the product default
int wide_for_fused(hk_ctx* c, hkd::WideTrees* out) {
  *out = hkd::WideTrees{};
  if (!wide_allowed(c)) return HK_OK;
  const int rc = ensure_wide(c, false);
  if (rc) return rc;
  out->tlas = c->wide_tlas;
  out->blas = c->wide_blas;
  out->tlas_rank = c->wide_tlas_rank;
  out->blas_rank = c->wide_blas_rank;
  out->tlas_count = c->scene.tlas_count;
  out->spill = nullptr;
  out->lost = c->d_counters + 8;
  return HK_OK;
}
// the scratch of the queue-based schedule: allocated on first use for the current render size
int ensure_wavefront(hk_ctx* c) {
  const size_t cap = (size_t)c->RW * c->RH;
  if (c->wf_mem && c->wf.cap == ((cap + 3) & ~(size_t)3)) return HK_OK;
  if (c->wf_mem) {
    HK_HIP(hipStreamSynchronize(c->stream));
    HK_HIP(hipFree(c->wf_mem));
    c->wf_mem = nullptr;
  }
  if (!c->compute_units) {
    hipDeviceProp_t prop;
    HK_HIP(hipGetDeviceProperties(&prop, c->device));
    c->compute_units = prop.multiProcessorCount;
  }
  // 16-B planes first, then the 4-B arrays; every sub-array stays 16-B aligned (cap rounded up to 4)
  const size_t n = (cap + 3) & ~(size_t)3;
  const size_t f4_planes = 9 + 2 + 2 + 1, u32_arrays = 1 + 1 + 1 + 1 + 2 + 2;  // state, cr, sr, ch0 | pixel, sr2, ch1, sh, alive[2], shadow[2]
  const size_t bytes = 1024 + n * (f4_planes * 16 + u32_arrays * 4);
  HK_HIP(hipMalloc(&c->wf_mem, bytes));
  uint8_t* p = (uint8_t*)c->wf_mem;
  hkd::WfBuffers& w = c->wf;
  w.ctr = (uint32_t*)p; p += 1024;
  auto f4 = [&](size_t planes) { float4* q = (float4*)p; p += planes * n * 16; return q; };
  auto u32 = [&](size_t arrays) { uint32_t* q = (uint32_t*)p; p += arrays * n * 4; return q; };
  w.state = f4(9);
  w.cr0 = f4(1); w.cr1 = f4(1); w.sr0 = f4(1); w.sr1 = f4(1); w.ch0 = f4(1);
  w.pixel = u32(1); w.sr2 = u32(1); w.ch1 = u32(1); w.sh = u32(1);
  w.alive[0] = u32(1); w.alive[1] = u32(1);
  w.shadow[0] = u32(1); w.shadow[1] = u32(1);
  w.cap = (uint32_t)n;
  const bool tl_twin = c->wf_timeline, count_twin = (c->flags & HK_CTX_COUNT_WALKS) != 0u;
  if ((tl_twin || count_twin) && !w.timeline) HK_HIP(hipMalloc((void**)&w.timeline, 64 * 32 * sizeof(unsigned long long)));  // tools/wf_timeline.py, bench.py
  w.timeline_mode = count_twin ? 2u : (tl_twin ? 1u : 0u);
  w.pb_add = nullptr; w.pb_sh = nullptr; w.local = nullptr; w.pb_bounces = 0u;   // (re-carved below for the new size)
  if (c->wf_paths_mem) { HK_HIP(hipFree(c->wf_paths_mem)); c->wf_paths_mem = nullptr; }
  return HK_OK;
}
// whether the queue-based schedule runs every bounce in one launch (kernels_wavefront.hip k_wf_trace_wide<.., PATHS>):
// hk_debug_set_option(HK_DEBUG_OPT_PERSISTENT_PATHS), -1 = the rule
#ifndef HK_PERSISTENT_PATHS_RULE
#define HK_PERSISTENT_PATHS_RULE false   // (measured: no faster than the stages on a full frame or a band - DESIGN 8.1c, profiles/r06_persistent_paths_ab.json)
#endif
static bool persistent_paths(const hk_ctx* c) {
  if (!(c->persistent_paths < 0 ? HK_PERSISTENT_PATHS_RULE : c->persistent_paths != 0)) return false;
  return use_wide(c) && !(c->flags & HK_CTX_COUNT_WALKS) && c->frame.indirect_bounces >= 1u && c->frame.indirect_bounces <= 16u &&
         (size_t)c->RW * c->RH <= ((size_t)1 << 26);
}
// ... its planes per bounce (36 B per path and bounce) and the waves' own lists (4 KB per wave of the launch), for at least `bounces`
int ensure_wavefront_paths(hk_ctx* c, uint32_t bounces) {
  hkd::WfBuffers& w = c->wf;
  if (c->wf_paths_mem && w.pb_bounces >= bounces) return HK_OK;
  if (c->wf_paths_mem) {
    HK_HIP(hipStreamSynchronize(c->stream));
    HK_HIP(hipFree(c->wf_paths_mem));
    c->wf_paths_mem = nullptr;
    w.pb_bounces = 0u;
  }
  const size_t n = w.cap, waves = hk::wide_trace_lanes(c->compute_units) / 64u;
  const size_t bytes = (size_t)bounces * n * (2 * 16 + 4) + waves * 1024 * sizeof(uint32_t);
  HK_HIP(hipMalloc(&c->wf_paths_mem, bytes));
  uint8_t* p = (uint8_t*)c->wf_paths_mem;
  w.pb_add = (float4*)p; p += (size_t)bounces * n * 32;
  w.pb_sh = (uint32_t*)p; p += (size_t)bounces * n * 4;
  w.local = (uint32_t*)p;
  w.pb_bounces = bounces;
  return HK_OK;
}

// make the main stream wait for what was enqueued on the side stream
int join_side(hk_ctx* c) {
  if (!c->forked) return HK_OK;
  HK_HIP(hipEventRecord(c->join_event, c->side_stream));
  HK_HIP(hipStreamWaitEvent(c->stream, c->join_event, 0));
  c->forked = false;
  c->side_state[0] = c->side_state[1] = 0;
  return HK_OK;
}
// make the main stream wait for the a-trous levels of the last frame (post_stream)
int join_post(hk_ctx* c) {
  for (int k = 0; k < 2; ++k)
    if (c->post_pending[k]) {
      HK_HIP(hipStreamWaitEvent(c->stream, c->post_done[k], 0));
      c->post_pending[k] = false;
    }
  return HK_OK;
}
// ... for the post-processing of the last frame of one parity only: whoever is about to write that parity's planes
int join_post_parity(hk_ctx* c, uint32_t parity) {
  if (!c->post_pending[parity & 1u]) return HK_OK;
  HK_HIP(hipStreamWaitEvent(c->stream, c->post_done[parity & 1u], 0));
  c->post_pending[parity & 1u] = false;
  return HK_OK;
}
int join_all(hk_ctx* c) {
  int rc = join_side(c);
  if (!rc) rc = join_post(c);
  if (!rc && c->comm) rc = comm_join(c, -1);  // a gather of the last frame still collecting rows on the communicator's stream
  return rc;
}
#define HK_FRAME_INTERNAL_LATE_JOIN 0x80000000u   // hk_frame_render -> hk_frame_stage: the caller puts exchange B behind the side stream itself (post_begin)
#ifndef HK_POST_DEMODULATION_RULE
#define HK_POST_DEMODULATION_RULE true
#endif
// Stage POST_PROCESS on the post stream (hk_context.hpp "Frame pipelining"): the main stream waits for the side stream, then the post
// stream takes over behind everything the main stream holds - demodulation, the a-trous levels and tone mapping of this frame (on a
// band with a communicator also exchange B) run there while the main stream goes on to the next frame.  `pipelined` = false: the frame
// stays on the main stream, which then waits for what the post stream still holds (timed passes, no denoiser, verification contexts).
// whether demodulation goes to the post stream with the levels (hk_debug_set_option(HK_DEBUG_OPT_POST_DEMODULATION): -1 = the rule)
static bool demod_on_post(const hk_ctx* c) { return c->post_demodulation < 0 ? HK_POST_DEMODULATION_RULE : c->post_demodulation != 0; }
int post_begin(hk_ctx* c, const HkSettings* st, bool* pipelined) {
  *pipelined = false;
  int rc;
  if (st->denoise && c->derived_dirty) {
    launch_derive_planes(c->stream, make_gbuffer(c), c->depth_plane, c->dn_g, c->W, 0, c->H);
    c->derived_dirty = false;
  }
  const uint32_t post_bits = (1u << HK_PASS_DEMODULATION) | (1u << HK_PASS_DENOISE_L0) | (1u << HK_PASS_DENOISE_L1) | (1u << HK_PASS_DENOISE_L2) | (1u << HK_PASS_DENOISE_L3);
  if (!(st->denoise && c->post_stream && c->frame_pipeline && !(c->timing_mask & post_bits) && c->albedo_twin && c->render_twin[0])) {
    if ((rc = join_side(c))) return rc;
    return join_post(c);  // the denoiser's internal planes: last frame's levels come first
  }
  if (c->side_join_each_frame && (rc = join_side(c))) return rc;   // (hk_debug_set_option: the order of rounds 1-5, for the A/B)
  // the post stream waits for the main stream AND (round 6) for the side stream itself: the main stream goes on to the next frame without
  // either wait (the direct-light dispatches stay "not joined": c->forked)
  if (c->forked) {
    HK_HIP(hipEventRecord(c->side_done, c->side_stream));
    HK_HIP(hipStreamWaitEvent(c->post_stream, c->side_done, 0));
    for (int k = 0; k < 2; ++k)   // (everything the side stream holds now - of either parity - is behind this frame's post-processing)
      if (c->side_state[k] == 1) {
        c->side_state[k] = 2;
        c->side_cover[k] = (uint8_t)(c->mapped_parity & 1u);
      }
  }
  HK_HIP(hipEventRecord(c->post_fork, c->stream));
  HK_HIP(hipStreamWaitEvent(c->post_stream, c->post_fork, 0));
  c->post_saved_main = c->stream;
  c->stream = c->post_stream;
  *pipelined = true;
  return HK_OK;
}
int post_end(hk_ctx* c, bool pipelined) {
  c->post_forked = false;
  if (!pipelined) return HK_OK;
  const uint32_t parity = c->mapped_parity & 1u;
  const hipError_t e = hipEventRecord(c->post_done[parity], c->post_stream);
  c->stream = c->post_saved_main;
  HK_REQUIRE(e == hipSuccess, HK_E_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
  c->post_pending[parity] = true;
  return HK_OK;
}
// run one dispatch on the side stream (timers record there too)
int run_pass(hk_ctx* c, uint32_t pass, uint32_t arg, int y0, int y1);
int run_pass_on_side(hk_ctx* c, uint32_t pass, uint32_t arg, int y0, int y1) {
  hipStream_t main_stream = c->stream;
  c->stream = c->side_stream;
  const int rc = (y1 > y0) ? run_pass(c, pass, arg, y0, y1) : HK_OK;
  c->stream = main_stream;
  return rc;
}

int run_pass(hk_ctx* c, uint32_t pass, uint32_t arg, int y0, int y1) {
  const DFrame fr = make_dframe(c);
  const GBuffer g = make_gbuffer(c);
  unsigned long long* counters = (c->flags & HK_CTX_COUNT_RAYS) ? c->d_counters : nullptr;
  if (c->derived_dirty && pass != HK_PASS_PREPASS) {  // G-buffer planes were written by the host: refresh the derived planes
    launch_derive_planes(c->stream, g, c->depth_plane, c->dn_g, c->W, 0, c->H);
    c->derived_dirty = false;
  }
  // (the two long dispatches of a frame carry their events ON the dispatch - hipExtLaunchKernelGGL - so that timing them inside a
  // pipelined frame adds no stream operation; every other pass is bracketed by two records)
  ScopedTimer timer(c, pass, pass == HK_PASS_INDIRECT || pass == HK_PASS_INDIRECT_SPATIAL_REUSE || pass == HK_PASS_EMISSIVE_SPATIAL_REUSE);
  switch (pass) {
    case HK_PASS_PREPASS: {
      Jitter j = prepass_jitter(c);
      hkd::WideTrees wide{};
      { const int rc_ = wide_for_fused(c, &wide); if (rc_) return rc_; }
      launch_prepass(c->stream, c->scene, fr, c->view.inverse_view_proj, c->view.view_proj, c->pview.view_proj, c->d_prev_models, j.x, j.y, g, y0, y1, counters, &wide);
      break;
    }
    case HK_PASS_FULL_SCREEN_ALBEDO: launch_albedo(c->stream, c->scene, fr, g, c->buf[HK_BUF_ALBEDO], y0, y1); break;
    case HK_PASS_DIRECT_LIT:
    case HK_PASS_DIRECT_EMISSIVE:
    case HK_PASS_INDIRECT: {
      const int channel = pass == HK_PASS_DIRECT_LIT ? 0 : (pass == HK_PASS_DIRECT_EMISSIVE ? 1 : 2);
      const bool across = parks_across_bands(c);
      if (across || scatter_lite(c, channel)) { const int rc_ = ensure_parked(c, across ? 7u : 1u << channel); if (rc_) return rc_; }
      LightTargets t = make_light_targets(c, channel);
      { const int rc_ = attach_tile_meta(c, t, channel, false, y0, y1); if (rc_) return rc_; }
      const size_t px = (size_t)c->RW * c->RH;
      if (t.det_winner && t.det_lite) {  // the light form: the winner plane is handed back clean by every resolve pass; nothing else to prepare
        if (!c->det_lite_clean[channel]) {
          // first use, or the channel's last dispatch was not in the light form (a band's full form, the spatial pass switched off): no
          // winners, no notes
          HK_HIP(hipMemsetAsync(t.det_winner, 0xFF, px * sizeof(int), c->stream));
          HK_HIP(hipMemsetAsync(t.det_to, 0xFF, px * sizeof(int), c->stream));
          c->det_lite_clean[channel] = true;
        }
      } else {
        c->det_lite_clean[channel] = false;
        if (t.det_winner) {  // nothing parked, no winner: -1 everywhere (a band: in the rows it dispatches - the others arrive with exchange A)
          HK_HIP(hipMemsetAsync(t.det_winner, 0xFF, px * sizeof(int), c->stream));
          if (across) HK_HIP(hipMemsetAsync(t.det_to + (size_t)y0 * c->RW, 0xFF, (size_t)(y1 - y0) * c->RW * sizeof(int), c->stream));
          else HK_HIP(hipMemsetAsync(t.det_to, 0xFF, px * sizeof(int), c->stream));
        }
      }
      if (pass == HK_PASS_INDIRECT && use_wavefront(c)) {
        { const int rc_ = ensure_wavefront(c); if (rc_) return rc_; }
        hkd::WideTrees wide{};
        if (use_wide(c)) {
          { const int rc_ = ensure_wide(c, true); if (rc_) return rc_; }
          wide.tlas = c->wide_tlas;
          wide.blas = c->wide_blas;
          wide.tlas_rank = c->wide_tlas_rank;
          wide.blas_rank = c->wide_blas_rank;
          wide.tlas_count = c->scene.tlas_count;
          wide.spill = c->wide_spill;
          wide.lost = c->d_counters + 8;
        }
        const bool persistent = persistent_paths(c);
        if (persistent) { const int rc_ = ensure_wavefront_paths(c, c->frame.indirect_bounces); if (rc_) return rc_; }
        // (HK_TIMING_TRACE_STAGES: every trace launch of the pass between its own pair of events; the persistent schedule has one)
        std::vector<hipEvent_t> trace_events;
        if ((c->timing_mask >> HK_TIMING_TRACE_STAGES) & 1u)
          for (uint32_t k = 0; k < (persistent ? 2u : 2u * (c->frame.indirect_bounces + 1u)); ++k) trace_events.push_back(get_event(c));
        launch_indirect_wavefront(c->stream, c->scene, fr, g, t, c->wf, y0, y1, c->compute_units, timer.on ? timer.t.start : nullptr,
                                  timer.on ? timer.t.stop : nullptr, &wide, trace_events.empty() ? nullptr : trace_events.data(), persistent);
        for (size_t k = 0; k + 1 < trace_events.size(); k += 2) c->pending.push_back(TimedLaunch{HK_TIMING_TRACE_STAGES, trace_events[k], trace_events[k + 1]});
      } else if (pass == HK_PASS_INDIRECT)  // MULTIPLE_BOUNCES pipeline iff bounces >= 2, light.rs:663-666
        launch_indirect(c->stream, c->frame.indirect_bounces >= 2u, c->scene, fr, g, t, y0, y1, counters, timer.on ? timer.t.start : nullptr,
                        timer.on ? timer.t.stop : nullptr);
      else
        launch_direct(c->stream, pass == HK_PASS_DIRECT_EMISSIVE, c->scene, fr, g, t, y0, y1, counters);
      // (a band with a history halo resolves at the start of stage SPATIAL, once the neighbours' parked rows are in)
      if (t.det_winner && t.det_lite) launch_resolve_scatter_lite(c->stream, t, fr, c->depth_plane, pass == HK_PASS_INDIRECT, y0, y1);
      else if (t.det_winner && !across) launch_resolve_scatter(c->stream, t, 0, (int)px, 0, 0);
      break;
    }
    case HK_PASS_EMISSIVE_SPATIAL_REUSE:
    case HK_PASS_INDIRECT_SPATIAL_REUSE: {
      const int channel = pass == HK_PASS_EMISSIVE_SPATIAL_REUSE ? 1 : 2;
      LightTargets t = make_light_targets(c, channel);
      { const int rc_ = attach_tile_meta(c, t, channel, true, y0, y1); if (rc_) return rc_; }
      if (launch_spatial(c->stream, channel == 1, c->scene, fr, g, t, y0, y1, c->spatial_window, timer.on ? timer.t.start : nullptr, timer.on ? timer.t.stop : nullptr))
        c->spatial_windowed_launches += 1;
      break;
    }
    case HK_PASS_DEMODULATION: {
      HK_REQUIRE(arg < 3, HK_E_INVALID, "channel out of range");
      DemodTargets d{};
      d.albedo = (const uint2*)c->buf[HK_BUF_ALBEDO];
      d.variance[0] = (const float*)c->buf[HK_BUF_VARIANCE0 + arg];
      d.render[0] = (const uint2*)c->buf[HK_BUF_RENDER0 + arg];
      d.output[0] = (uint2*)c->buf[HK_BUF_DENOISE_INTERNAL0];
      d.internal_variance[0] = (float*)c->buf[HK_BUF_DENOISE_INTERNAL_VARIANCE];
      launch_demodulation(c->stream, 1, fr, d, y0, y1);
      break;
    }
    case HK_PASS_DENOISE_L0: case HK_PASS_DENOISE_L1: case HK_PASS_DENOISE_L2: case HK_PASS_DENOISE_L3: {
      HK_REQUIRE(arg < 3, HK_E_INVALID, "channel out of range");
      const int level = (int)(pass - HK_PASS_DENOISE_L0);
      DenoiseTargets d{};
      d.albedo = (const uint2*)c->buf[HK_BUF_ALBEDO];
      d.dn_g = (const float4*)c->dn_g;
      d.depth = c->depth_plane;
      d.depth_gradient = (const float2*)c->buf[HK_BUF_DEPTH_GRADIENT];
      d.input[0] = (const uint2*)c->buf[HK_BUF_DENOISE_INTERNAL0 + level];
      d.output[0] = level == 3 ? (uint2*)c->buf[HK_BUF_DENOISE_RENDER0 + arg] : (uint2*)c->buf[HK_BUF_DENOISE_INTERNAL0 + level + 1];
      d.internal_variance[0] = (const float*)c->buf[HK_BUF_DENOISE_INTERNAL_VARIANCE];
      // denoise_direct has no FIREFLY_FILTERING, post_process.rs:773-783,1193-1197
      launch_denoise(c->stream, level, 1, arg != 0 ? 1 : 0, fr, d, y0, y1);
      break;
    }
    case HK_PASS_TONE_MAPPING: {  // inputs per post_process.rs:941-954
      const uint32_t base = arg ? HK_BUF_DENOISE_RENDER0 : HK_BUF_RENDER0;
      const void* indirect = c->frame.indirect_bounces != 0u ? c->buf[base + 2] : nullptr;
      launch_tone_mapping(c->stream, fr, c->buf[base], c->buf[base + 1], indirect, c->buf[HK_BUF_TONE_MAPPED], y0, y1);
      break;
    }
    case HK_PASS_SMAA_TU4X:
    case HK_PASS_TAA_JASMINE: {  // bindings post_process.rs:983-1035
      AaBuffers ab{};
      ab.position = c->buf[HK_BUF_POSITION]; ab.velocity_uv = c->buf[HK_BUF_VELOCITY_UV];
      ab.previous_position = c->buf[HK_BUF_PREVIOUS_POSITION]; ab.previous_velocity_uv = c->buf[HK_BUF_PREVIOUS_VELOCITY_UV];
      ab.instance_material = c->buf[HK_BUF_INSTANCE_MATERIAL];
      ab.depth = c->depth_plane; ab.previous_depth = c->prev_depth_plane;
      ab.full_w = c->W; ab.full_h = c->H;
      if (pass == HK_PASS_SMAA_TU4X) {
        ab.render = c->buf[HK_BUF_TONE_MAPPED]; ab.render_w = c->RW; ab.render_h = c->RH;
        ab.previous_render = c->buf[HK_BUF_PREVIOUS_TONE_MAPPED]; ab.previous_w = c->RW; ab.previous_h = c->RH;
        ab.output = c->buf[HK_BUF_UPSCALE_OUTPUT]; ab.out_w = c->UW; ab.out_h = c->UH;
        launch_smaa_tu4x(c->stream, ab, c->frame.number, y0, y1);
      } else {
        const uint32_t in = c->upscale_kind == HK_UPSCALE_SMAA_TU4X ? HK_BUF_UPSCALE_OUTPUT : HK_BUF_TONE_MAPPED;  // post_process.rs:1010-1013
        buffer_dims(c, in, &ab.render_w, &ab.render_h);
        ab.render = c->buf[in];
        buffer_dims(c, HK_BUF_TAA_OUTPUT, &ab.out_w, &ab.out_h);
        ab.previous_render = c->buf[HK_BUF_PREVIOUS_TAA_OUTPUT]; ab.previous_w = ab.out_w; ab.previous_h = ab.out_h;
        ab.output = c->buf[HK_BUF_TAA_OUTPUT];
        launch_taa_jasmine(c->stream, ab, 0.1f / c->frame.upscale_ratio, c->frame.clear_color, y0, y1);
      }
      break;
    }
    case HK_PASS_SMAA_TU4X_EXTRAPOLATE:
      launch_smaa_tu4x_extrapolate(c->stream, c->buf[HK_BUF_UPSCALE_OUTPUT], c->UW, c->UH, c->RW, y0, y1);
      break;
    case HK_PASS_FSR_EASU: {  // post_process.rs:1037-1040,1277-1292: taa_output[current] when TAA is on, else tone-mapped
      const uint32_t in = c->taa == HK_TAA_JASMINE ? HK_BUF_TAA_OUTPUT : HK_BUF_TONE_MAPPED;
      int iw, ih;
      buffer_dims(c, in, &iw, &ih);
      HK_REQUIRE(c->upscale_kind == HK_UPSCALE_FSR1, HK_E_INVALID, "FSR passes need upscale_kind FSR1");
      launch_fsr_easu(c->stream, c->buf[in], iw, ih, c->buf[HK_BUF_UPSCALE_OUTPUT], c->W, c->H, y0, y1);
      break;
    }
    case HK_PASS_FSR_RCAS:    // post_process.rs:1294-1308
      HK_REQUIRE(c->upscale_kind == HK_UPSCALE_FSR1, HK_E_INVALID, "FSR passes need upscale_kind FSR1");
      launch_fsr_rcas(c->stream, c->buf[HK_BUF_UPSCALE_OUTPUT], c->buf[HK_BUF_UPSCALE_SHARPENED], c->W, c->H, c->upscale_sharpness, y0, y1);
      break;
    default: HK_REQUIRE(false, HK_E_INVALID, "unknown pass %u", pass);
  }
  HK_HIP(hipGetLastError());
  return HK_OK;
}

}  // namespace hk

namespace hk {
int ctx_info(hk_ctx* c, CtxInfo* o) {
  HK_REQUIRE(c && o, HK_E_INVALID, "bad argument");
  o->device = c->device;
  o->stream = (void*)c->stream;
  o->width = (uint32_t)c->W;
  o->height = (uint32_t)c->H;
  o->ratio = c->ratio;
  o->frame_number = c->frame.number;
  o->band_index = c->band_index;
  o->band_count = c->band_count;
  o->upscale_kind = c->upscale_kind;
  o->taa = c->taa;
  o->band_bounds = c->band_bounds.size() == (size_t)c->band_count + 1 ? c->band_bounds.data() : nullptr;
  o->bounds_generation = c->bounds_generation;
  return HK_OK;
}
void* ctx_buffer(hk_ctx* c, uint32_t b, size_t* logical_bytes) {
  if (!c || b >= HK_BUF_COUNT) return nullptr;
  if (logical_bytes) *logical_bytes = buffer_logical_bytes(c, b);
  return c->buf[b];
}
void** ctx_comm_slot(hk_ctx* c) { return &c->comm; }
int ctx_join_side(hk_ctx* c) { return join_all(c); }
}  // namespace hk

extern "C" {

int hk_device_count(int* count) {
  HK_REQUIRE(count, HK_E_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return HK_E_NO_DEVICE;
  }
  *count = n;
  return HK_OK;
}

int hk_create(int device_id, uint32_t flags, hk_ctx** out) {
  HK_REQUIRE(out, HK_E_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  HK_REQUIRE(e == hipSuccess && n > 0, HK_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(e));
  HK_REQUIRE(device_id >= 0 && device_id < n, HK_E_NO_DEVICE, "device id %d out of range (0..%d)", device_id, n - 1);
  HK_HIP(hipSetDevice(device_id));
  hk_ctx* c = new (std::nothrow) hk_ctx();
  HK_REQUIRE(c, HK_E_NOMEM, "allocation failed");
  c->device = device_id;
  c->flags = flags;
  c->timing_mask = (flags & HK_CTX_TIME_PASSES) ? ((1u << HK_PASS_COUNT) - 1u) : 0u;  // (HK_TIMING_TRACE_STAGES, events around every trace launch, only through hk_set_timing_mask)
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&c->d_counters, 9 * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(c->d_counters, 0, 9 * sizeof(unsigned long long)) != hipSuccess || hipEventCreate(&c->frame_start) != hipSuccess ||
      hipEventCreate(&c->frame_stop) != hipSuccess) {
    set_error("HIP resource creation failed: %s", hipGetErrorString(hipGetLastError()));
    hk_destroy(c);
    return HK_E_HIP;
  }
  c->stream = c->own_stream;
  if (!(flags & HK_CTX_SINGLE_STREAM)) {
    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->fork_event, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join_event, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming) != hipSuccess) {
      set_error("cannot create the side stream");
      hk_destroy(c);
      return HK_E_HIP;
    }
    // (the verification mode keeps ONE set of planes: a host that looks at a buffer between two dispatches - the fixture replays of
    // tests/test_wgsl_pin.py - finds in it what the last frame left, as in the reference)
    if (!(flags & (HK_CTX_DETERMINISTIC_SCATTER | HK_CTX_COUNT_RAYS | HK_CTX_TIME_PASSES))) {
      if (hipStreamCreateWithFlags(&c->post_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->post_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&c->post_done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->post_done[1], hipEventDisableTiming) != hipSuccess) {
        set_error("cannot create the post-process stream");
        hk_destroy(c);
        return HK_E_HIP;
      }
    }
  }
  *out = c;
  return HK_OK;
}

void hk_destroy(hk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)join_all(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  comm_release(c);
  drain_timers(c);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->band_ev)
    if (e) (void)hipEventDestroy(e);
  if (c->frame_start) (void)hipEventDestroy(c->frame_start);
  if (c->frame_stop) (void)hipEventDestroy(c->frame_stop);
  free_screen(c);
  if (c->scene_mem) (void)hipFree(c->scene_mem);
  for (int k = 0; k < 2; ++k) {
    if (c->staging[k]) (void)hipHostFree(c->staging[k]);
    if (c->staging_done[k]) (void)hipEventDestroy(c->staging_done[k]);
  }
  free_refit(c);
  for (void* q : {(void*)c->wide_tlas, (void*)c->wide_blas, (void*)c->wide_spill, (void*)c->wide_tlas_rank, (void*)c->wide_blas_rank})
    if (q) (void)hipFree(q);
  c->d_tex_data.release();
  c->d_noise.release();
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  if (c->post_stream) (void)hipStreamDestroy(c->post_stream);
  if (c->post_fork) (void)hipEventDestroy(c->post_fork);
  for (int k = 0; k < 2; ++k)
    if (c->post_done[k]) (void)hipEventDestroy(c->post_done[k]);
  if (c->fork_event) (void)hipEventDestroy(c->fork_event);
  if (c->join_event) (void)hipEventDestroy(c->join_event);
  if (c->side_done) (void)hipEventDestroy(c->side_done);
  if (c->pre_stream) { (void)hipStreamSynchronize(c->pre_stream); (void)hipStreamDestroy(c->pre_stream); }
  if (c->pre_done) (void)hipEventDestroy(c->pre_done);
  if (c->pre_scene_mark) (void)hipEventDestroy(c->pre_scene_mark);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}
// tools/wf_timeline.py: the 64 x 32 u64 the instrumented trace kernel left for the frame most recently rendered (HK_WF_TIMELINE=1)
int hk_debug_read_wf_timeline(hk_ctx* c, unsigned long long* out, uint32_t n) {
  HK_REQUIRE(c && out && n == 64u * 32u, HK_E_INVALID, "bad argument");
  HK_REQUIRE(c->wf.timeline, HK_E_NOT_READY, "no timeline: set HK_WF_TIMELINE=1 (or create the context with HK_CTX_COUNT_WALKS) before the first frame of a scene beyond the LDS copy");
  HK_HIP(hipSetDevice(c->device));
  { const int rc = sync_all(c); if (rc) return rc; }
  HK_HIP(hipMemcpy(out, c->wf.timeline, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  int khz = 0;
  HK_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
  out[n - 1] = (unsigned long long)khz;  // (slot 31 of stage 63 - never a real stage: the rate of wall_clock64, kHz)
  return HK_OK;
}

// hikari_hip_debug.h: the switches tests and A/B tools used to pass through the environment
int hk_debug_set_option(hk_ctx* c, uint32_t option, int64_t value) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  HK_HIP(hipSetDevice(c->device));
  { const int rc = sync_all(c); if (rc) return rc; }
  switch (option) {
    case HK_DEBUG_OPT_SPATIAL_WINDOW: c->spatial_window = value < 0 ? -1 : (value ? 1 : 0); break;
    case HK_DEBUG_OPT_FRAME_PIPELINE: c->frame_pipeline = value != 0; break;
    case HK_DEBUG_OPT_WF_TIMELINE:
      c->wf_timeline = value != 0;
      if (c->wf_mem) { (void)hipFree(c->wf_mem); c->wf_mem = nullptr; }  // (re-carved on the next queue-based pass, with or without the timeline)
      break;
    case HK_DEBUG_OPT_FLAT_WALK: c->flat_walk = value != 0; c->dynamic_dirty = true; break;
    case HK_DEBUG_OPT_FLAT_ORDERINGS: c->flat_orderings = (int)std::max<int64_t>(0, std::min<int64_t>(8, value)); c->dynamic_dirty = true; break;
    case HK_DEBUG_OPT_TRACE_UPDATE: c->trace_update = value != 0; break;
    case HK_DEBUG_OPT_SIDE_JOIN: c->side_join_each_frame = value != 0; break;
    case HK_DEBUG_OPT_POST_DEMODULATION: c->post_demodulation = value < 0 ? -1 : (value ? 1 : 0); break;
    case HK_DEBUG_OPT_PERSISTENT_PATHS: c->persistent_paths = value < 0 ? -1 : (value ? 1 : 0); break;
    case HK_DEBUG_OPT_PREPASS_PIPELINE: c->prepass_pipeline = value < 0 ? -1 : (value ? 1 : 0); break;
    case HK_DEBUG_OPT_MAIN_PRIORITY: c->main_priority = value < 0 ? -1 : (value ? 1 : 0); return pick_main_stream(c, true);
    default: HK_REQUIRE(false, HK_E_INVALID, "unknown option %u", option);
  }
  return HK_OK;
}

int hk_debug_main_stream_priority(hk_ctx* c, uint32_t* out) {
  HK_REQUIRE(c && out, HK_E_INVALID, "bad argument");
  *out = (c->own_stream_high ? 1u : 0u) | (c->main_priority_decided ? 2u : 0u) | (c->pre_stream ? 4u : 0u) | ((uint32_t)std::min<uint64_t>(c->prepasses_pipelined, 0xFFFFFu) << 8);
  return HK_OK;
}

int hk_debug_spatial_windowed_launches(hk_ctx* c, uint64_t* out) {
  HK_REQUIRE(c && out, HK_E_INVALID, "bad argument");
  *out = c->spatial_windowed_launches;
  return HK_OK;
}

static int resize_resources(hk_ctx* c, uint32_t width, uint32_t height, float upscale_ratio);

int hk_resize(hk_ctx* c, uint32_t width, uint32_t height, float upscale_ratio) {
  // pixel indices are 32-bit signed in the kernels (x + width * y); 16384^2 leaves room for the 2x SMAA Tu4x output as well
  HK_REQUIRE(c && width && height && width <= 16384u && height <= 16384u, HK_E_INVALID, "bad size %u x %u (1..16384)", width, height);
  const int rc = resize_resources(c, width, height, upscale_ratio);
  if (rc) {  // e.g. out of device memory half way: leave the context without screen resources rather than with some of them
    free_screen(c);
    c->W = c->H = c->RW = c->RH = c->UW = c->UH = 0;
  }
  return rc;
}

static int resize_resources(hk_ctx* c, uint32_t width, uint32_t height, float upscale_ratio) {
  HK_HIP(hipSetDevice(c->device));
  { const int rc_ = sync_all(c); if (rc_) return rc_; }
  free_screen(c);
  if (!c->band_bounds.empty()) {  // an explicit band split is in rows of the OLD render image
    c->band_bounds.clear();
    c->bounds_generation += 1;
  }
  uint32_t rw, rh;
  int rc = hk_scaled_size(width, height, upscale_ratio, &rw, &rh);
  if (rc) return rc;
  c->W = (int)width; c->H = (int)height; c->RW = (int)rw; c->RH = (int)rh;
  c->ratio = upscale_ratio < 1.0f ? 1.0f : (upscale_ratio > 2.0f ? 2.0f : upscale_ratio);
  {
    const float scale2 = (1.0f / c->ratio) * 2.0f;
    c->UW = (int)ceilf((float)width * scale2);
    c->UH = (int)ceilf((float)height * scale2);
    c->mapped_parity = 0;
    c->pre_chain_ok = false;
  }
  for (uint32_t b = 0; b < HK_BUF_PARKED_TO0; ++b) {  // (the parked-store planes beyond: on first use, ensure_parked)
    size_t n = buffer_is_full_size(b) ? (size_t)c->W * c->H : (buffer_is_upscaled(b) ? (size_t)c->UW * c->UH : (size_t)c->RW * c->RH);
    size_t bytes = n * buffer_bpp(b);
    HK_HIP(hipMalloc(&c->buf[b], bytes));
    HK_HIP(hipMemset(c->buf[b], 0, bytes));  // zeroed reservoirs, light.rs:352-360
    c->buf_bytes[b] = bytes;
  }
  const size_t nf = (size_t)c->W * c->H, nr = (size_t)c->RW * c->RH;
  {  // (round 6: the verification mode parks only what crosses pixels - the light form - and keeps the elision)
    c->tiles_x = (c->RW + 7) / 8;
    c->tiles_y = (c->RH + 7) / 8;
    const size_t mb = (size_t)c->tiles_x * c->tiles_y * sizeof(TileMeta);
    for (int k = 0; k < 10; ++k) {
      HK_HIP(hipMalloc((void**)&c->tile_meta[k], mb));
      HK_HIP(hipMemset(c->tile_meta[k], 0, mb));
      c->tile_meta_zero[k] = true;
    }
  }
  HK_HIP(hipMalloc((void**)&c->depth_plane, nf * 4));
  HK_HIP(hipMemset(c->depth_plane, 0, nf * 4));
  HK_HIP(hipMalloc((void**)&c->prev_depth_plane, nf * 4));
  HK_HIP(hipMemset(c->prev_depth_plane, 0, nf * 4));
  HK_HIP(hipMalloc(&c->dn_g, nf * 16));
  HK_HIP(hipMemset(c->dn_g, 0, nf * 16));
  if (c->post_stream) {  // the other frame parity's planes of what the a-trous levels read of the G-buffer (frame pipelining)
    HK_HIP(hipMalloc(&c->albedo_twin, c->buf_bytes[HK_BUF_ALBEDO]));
    HK_HIP(hipMemset(c->albedo_twin, 0, c->buf_bytes[HK_BUF_ALBEDO]));
    HK_HIP(hipMalloc(&c->depth_gradient_twin, c->buf_bytes[HK_BUF_DEPTH_GRADIENT]));
    HK_HIP(hipMemset(c->depth_gradient_twin, 0, c->buf_bytes[HK_BUF_DEPTH_GRADIENT]));
    HK_HIP(hipMalloc(&c->dn_g_twin, nf * 16));
    HK_HIP(hipMemset(c->dn_g_twin, 0, nf * 16));
    // (round 6: the two G-buffer planes that had no twin - the direct-light dispatches of frame n may still read them while frame n + 1's primary rays write)
    HK_HIP(hipMalloc(&c->normal_twin, c->buf_bytes[HK_BUF_NORMAL]));
    HK_HIP(hipMemset(c->normal_twin, 0, c->buf_bytes[HK_BUF_NORMAL]));
    HK_HIP(hipMalloc(&c->instance_material_twin, c->buf_bytes[HK_BUF_INSTANCE_MATERIAL]));
    HK_HIP(hipMemset(c->instance_material_twin, 0, c->buf_bytes[HK_BUF_INSTANCE_MATERIAL]));
    for (int ch = 0; ch < 3; ++ch) {  // (round 6: what demodulation reads of the light passes' outputs - it runs beside the next frame's light passes too)
      HK_HIP(hipMalloc(&c->render_twin[ch], c->buf_bytes[HK_BUF_RENDER0 + ch]));
      HK_HIP(hipMemset(c->render_twin[ch], 0, c->buf_bytes[HK_BUF_RENDER0 + ch]));
      HK_HIP(hipMalloc(&c->variance_twin[ch], c->buf_bytes[HK_BUF_VARIANCE0 + ch]));
      HK_HIP(hipMemset(c->variance_twin[ch], 0, c->buf_bytes[HK_BUF_VARIANCE0 + ch]));
    }
  }
  for (int k = 0; k < 2; ++k) {
    for (int l = 0; l < 4; ++l) {
      HK_HIP(hipMalloc(&c->dn_extra[k][l], nr * 8));
      HK_HIP(hipMemset(c->dn_extra[k][l], 0, nr * 8));
    }
    HK_HIP(hipMalloc((void**)&c->dn_extra_var[k], nr * 4));
    HK_HIP(hipMemset(c->dn_extra_var[k], 0, nr * 4));
  }
  c->derived_dirty = false;
  c->uv_fast = !(c->flags & HK_CTX_PLAIN_DIVISION) && certify_uv_division(c->W) && certify_uv_division(c->H) && certify_uv_division(c->RW) && certify_uv_division(c->RH);
  HK_HIP(hipDeviceSynchronize());
  return HK_OK;
}

int hk_set_view_options(hk_ctx* c, uint32_t taa, uint32_t upscale_kind, float upscale_sharpness) {
  HK_REQUIRE(c && taa <= HK_TAA_NONE && upscale_kind <= HK_UPSCALE_SMAA_TU4X, HK_E_INVALID, "bad argument");
  c->upscale_sharpness = upscale_sharpness;
  c->taa = taa;
  c->upscale_kind = upscale_kind;
  return HK_OK;
}

int hk_frame_begin(hk_ctx* c, const HkFrame* f, const HkView* v, const HkPreviousView* pv, const HkLights* l) {
  HK_REQUIRE(c && f && v && pv && l, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(f->direct_validate_interval > 0 && f->emissive_validate_interval > 0, HK_E_INVALID, "validate intervals must be > 0");
  if (!c->main_priority_decided) { const int rc = pick_main_stream(c, false); if (rc) return rc; }  // (the context's first frame: its size and its band are known)
  c->frame = *f;
  c->view = *v;
  c->pview = *pv;
  c->lights = *l;
  c->have_frame = true;
  // double-buffered planes follow frame.number % 2 (hikari_hip.h, HkBuffer): idempotent per frame number.
  // Work already enqueued captured its pointers at launch, so swapping here needs no synchronisation.
  if ((f->number & 1u) != c->mapped_parity) {
    std::swap(c->buf[HK_BUF_POSITION], c->buf[HK_BUF_PREVIOUS_POSITION]);
    std::swap(c->depth_plane, c->prev_depth_plane);
    std::swap(c->buf[HK_BUF_VELOCITY_UV], c->buf[HK_BUF_PREVIOUS_VELOCITY_UV]);
    std::swap(c->buf[HK_BUF_TONE_MAPPED], c->buf[HK_BUF_PREVIOUS_TONE_MAPPED]);
    std::swap(c->buf[HK_BUF_TAA_OUTPUT], c->buf[HK_BUF_PREVIOUS_TAA_OUTPUT]);
    if (c->albedo_twin) {  // (frame pipelining: the planes last frame's a-trous levels may still be reading stay untouched)
      std::swap(c->buf[HK_BUF_ALBEDO], c->albedo_twin);
      std::swap(c->buf[HK_BUF_DEPTH_GRADIENT], c->depth_gradient_twin);
      std::swap(c->dn_g, c->dn_g_twin);
      std::swap(c->buf[HK_BUF_NORMAL], c->normal_twin);
      std::swap(c->buf[HK_BUF_INSTANCE_MATERIAL], c->instance_material_twin);
      for (int ch = 0; ch < 3; ++ch) {
        std::swap(c->buf[HK_BUF_RENDER0 + ch], c->render_twin[ch]);
        std::swap(c->buf[HK_BUF_VARIANCE0 + ch], c->variance_twin[ch]);
      }
    }
    c->mapped_parity = f->number & 1u;
  }
  if (c->comm) { const int rc = comm_join(c, (int)(f->number & 1u)); if (rc) return rc; }  // a gather still reading the plane this frame writes
  // the history halo of this frame: a count the host set, or the bound every rank derives from the same uniforms
  c->history_now = 0;
  if (c->band_count > 1 && c->RH > 0) {
    if (c->history_rows != HK_HISTORY_AUTO) {
      c->history_now = std::min(c->history_rows, (uint32_t)c->RH);
    } else {
      const int rc = derive_history_rows(c, &c->history_now);
      if (rc) return rc;
    }
    if (parks_across_bands(c)) { const int rc = ensure_parked(c, 7u); if (rc) return rc; }
  }
  return HK_OK;
}

int hk_set_history_rows(hk_ctx* c, uint32_t rows) {
  HK_REQUIRE(c && rows <= HK_HISTORY_AUTO, HK_E_INVALID, "bad argument");
  c->history_rows = rows;
  return HK_OK;
}
int hk_history_rows(hk_ctx* c, uint32_t* rows) {
  HK_REQUIRE(c && rows, HK_E_INVALID, "bad argument");
  *rows = c->history_now;
  return HK_OK;
}
int hk_scene_bounds(hk_ctx* c, float mn[3], float mx[3]) {
  HK_REQUIRE(c && mn && mx, HK_E_INVALID, "bad argument");
  HK_REQUIRE(c->have_instances, HK_E_NOT_READY, "no scene uploaded");
  scene_bounds(c, mn, mx);
  return HK_OK;
}

int hk_pass_run(hk_ctx* c, uint32_t pass, uint32_t arg, uint32_t row_begin, uint32_t row_end) {
  int rc = ready(c);
  if (rc) return rc;
  if ((rc = join_all(c))) return rc;
  const bool full_grid = pass == HK_PASS_PREPASS || pass == HK_PASS_FULL_SCREEN_ALBEDO;
  int rows = full_grid ? c->H : c->RH;
  if (pass == HK_PASS_TAA_JASMINE) { int w; buffer_dims(c, HK_BUF_TAA_OUTPUT, &w, &rows); }
  if (pass == HK_PASS_FSR_EASU || pass == HK_PASS_FSR_RCAS) rows = c->H;
  const int y0 = (int)row_begin, y1 = row_end == 0 ? rows : (int)row_end;
  HK_REQUIRE(y0 >= 0 && y1 <= rows && y0 <= y1, HK_E_INVALID, "row range [%d,%d) outside 0..%d", y0, y1, rows);
  return run_pass(c, pass, arg, y0, y1);
}

int hk_set_band(hk_ctx* c, uint32_t band_index, uint32_t band_count) {
  HK_REQUIRE(c && band_count > 0 && band_index < band_count, HK_E_INVALID, "bad band");
  if (band_count != c->band_count && !c->band_bounds.empty()) {  // another band count: the explicit split no longer applies
    c->band_bounds.clear();
    c->bounds_generation += 1;
  }
  c->band_index = band_index;
  c->band_count = band_count;
  return HK_OK;
}

int hk_set_band_bounds(hk_ctx* c, const uint32_t* bounds, uint32_t n_bounds) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  if (!bounds || n_bounds == 0) {  // back to the equal split
    if (!c->band_bounds.empty()) c->bounds_generation += 1;
    c->band_bounds.clear();
    return HK_OK;
  }
  HK_REQUIRE(c->RH > 0, HK_E_NOT_READY, "hk_resize has not been called");
  HK_REQUIRE(n_bounds == c->band_count + 1, HK_E_INVALID, "need band_count + 1 = %u boundaries (hk_set_band first), got %u", c->band_count + 1, n_bounds);
  HK_REQUIRE(band_bounds_valid(bounds, c->band_count, (uint32_t)c->RH), HK_E_INVALID,
             "band bounds must run 0 = b[0] < b[1] < ... < b[%u] = %d (scaled render rows)", c->band_count, c->RH);
  c->band_bounds.assign(bounds, bounds + n_bounds);
  c->bounds_generation += 1;
  return HK_OK;
}

int hk_get_band(hk_ctx* c, uint32_t* band_index, uint32_t* band_count) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  if (band_index) *band_index = c->band_index;
  if (band_count) *band_count = c->band_count;
  return HK_OK;
}
int hk_get_band_bounds(hk_ctx* c, uint32_t* bounds, uint32_t n_bounds) {  // the split in force, explicit or equal
  HK_REQUIRE(c && bounds, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(c->RH > 0, HK_E_NOT_READY, "hk_resize has not been called");
  HK_REQUIRE(n_bounds == c->band_count + 1, HK_E_INVALID, "need band_count + 1 = %u entries", c->band_count + 1);
  const uint32_t* explicit_split = c->band_bounds.size() == (size_t)c->band_count + 1 ? c->band_bounds.data() : nullptr;
  for (uint32_t i = 0; i < c->band_count; ++i) {
    uint32_t b0, b1;
    band_rows_in(explicit_split, (uint32_t)c->RH, (uint32_t)c->RH, i, c->band_count, &b0, &b1);
    bounds[i] = b0;
    bounds[i + 1] = b1;
  }
  return HK_OK;
}

// geometry pixels (depth >= epsilon) per row of the full-size G-buffer of the frame most recently begun: the cost estimate
// hk_balanced_band_bounds splits by.  A host shards a frame by cost like this: every rank ray-casts the WHOLE frame's primary rays
// once (hk_frame_begin + hk_pass_run(HK_PASS_PREPASS) over all rows - cheap next to a frame), counts, and derives the same
// boundaries as every other rank without a word of communication.
int hk_row_costs(hk_ctx* c, uint32_t* out, uint32_t n_rows) {
  HK_REQUIRE(c && out, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(c->H > 0 && c->depth_plane, HK_E_NOT_READY, "hk_resize has not been called");
  HK_REQUIRE(n_rows == (uint32_t)c->H, HK_E_INVALID, "need one counter per full-size row: %d", c->H);
  HK_HIP(hipSetDevice(c->device));
  { const int rc_ = sync_all(c); if (rc_) return rc_; }
  uint32_t* d = nullptr;
  HK_HIP(hipMalloc((void**)&d, (size_t)n_rows * 4));
  launch_count_geometry_rows(c->stream, c->depth_plane, c->W, c->H, d);
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)n_rows * 4, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HK_REQUIRE(e == hipSuccess, HK_E_HIP, "hk_row_costs failed: %s", hipGetErrorString(e));
  return HK_OK;
}

// hk_frame_begin has run: ray-cast the whole frame's primary rays, count, split by cost, set the split on this context.  Every rank
// (or band context) that does this for the same frame arrives at the same boundaries - the G-buffer is bit-identical everywhere.
int hk_balance_bands(hk_ctx* c, uint32_t min_rows, uint32_t* bounds_out, uint32_t n_bounds) {
  int rc = ready(c);
  if (rc) return rc;
  HK_REQUIRE(c->have_frame, HK_E_NOT_READY, "hk_frame_begin has not been called");
  HK_REQUIRE(!bounds_out || n_bounds == c->band_count + 1, HK_E_INVALID, "need band_count + 1 = %u boundaries", c->band_count + 1);
  std::vector<uint32_t> bounds(c->band_count + 1);
  if (c->band_count > 1) {
    if ((rc = hk_pass_run(c, HK_PASS_PREPASS, 0, 0, 0))) return rc;
    std::vector<uint32_t> cost((size_t)c->H);
    if ((rc = hk_row_costs(c, cost.data(), (uint32_t)c->H))) return rc;
    if ((rc = hk_balanced_band_bounds(cost.data(), (uint32_t)c->H, (uint32_t)c->W, (uint32_t)c->RH, c->band_count,
                                      std::max(1u, std::min(min_rows ? min_rows : 8u, (uint32_t)c->RH / c->band_count)) /* (a short frame: what fits) */,
                                      (size_t)c->scene.blob_f4 * 16 > HK_LDS_SCENE_BYTES ? 1.0f / 16.0f : 0.25f, bounds.data()))) return rc;
    if ((rc = hk_set_band_bounds(c, bounds.data(), c->band_count + 1))) return rc;
  } else {
    bounds[0] = 0u;
    bounds[1] = (uint32_t)c->RH;
  }
  if (bounds_out) std::copy(bounds.begin(), bounds.end(), bounds_out);
  return HK_OK;
}

int hk_band_time_ms(hk_ctx* c, float* ms) {
  HK_REQUIRE(c && ms, HK_E_INVALID, "bad argument");
  HK_REQUIRE(c->band_timed, HK_E_NOT_READY, "no frame has been rendered with HK_FRAME_TIME_BAND");
  HK_HIP(hipSetDevice(c->device));
  HK_HIP(hipEventSynchronize(c->band_ev[3]));
  float a = 0.0f, b = 0.0f;
  HK_HIP(hipEventElapsedTime(&a, c->band_ev[0], c->band_ev[1]));
  HK_HIP(hipEventElapsedTime(&b, c->band_ev[2], c->band_ev[3]));
  *ms = a + b;
  return HK_OK;
}

int hk_migrate_bands(hk_ctx* c, const uint32_t* new_bounds, uint32_t n_bounds, uint32_t next_frame_number, const HkSettings* st) {
  HK_REQUIRE(c && st, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(c->RH > 0, HK_E_NOT_READY, "hk_resize has not been called");
  HK_REQUIRE(!new_bounds || n_bounds == c->band_count + 1, HK_E_INVALID, "need band_count + 1 = %u boundaries", c->band_count + 1);
  HK_REQUIRE(band_bounds_valid(new_bounds, c->band_count, (uint32_t)c->RH), HK_E_INVALID, "band bounds must run 0 = b[0] < b[1] < ... < b[%u] = %d", c->band_count, c->RH);
  HK_HIP(hipSetDevice(c->device));
  int rc = join_all(c);
  if (rc) return rc;
  const uint32_t* old_bounds = c->band_bounds.size() == (size_t)c->band_count + 1 ? c->band_bounds.data() : nullptr;
  if (c->band_count > 1 && (rc = comm_migrate(c, old_bounds, new_bounds, next_frame_number, st))) return rc;
  return hk_set_band_bounds(c, new_bounds, new_bounds ? n_bounds : 0u);
}

int hk_band_plan(hk_ctx* c, uint32_t stage, const HkSettings* st, HkHaloOp* ops, uint32_t* n_ops) {
  HK_REQUIRE(c && c->W > 0, HK_E_NOT_READY, "hk_resize has not been called");
  HK_REQUIRE(c->have_frame, HK_E_NOT_READY, "hk_frame_begin has not been called");
  const uint32_t* bounds = c->band_bounds.size() == (size_t)c->band_count + 1 ? c->band_bounds.data() : nullptr;
  return hk_band_plan_bounds((uint32_t)c->W, (uint32_t)c->H, c->ratio, bounds, c->band_index, c->band_count, stage, c->frame.number, st, ops, n_ops);
}

int hk_frame_stage(hk_ctx* c, uint32_t stage, const HkSettings* st, uint32_t flags) {
  int rc = ready(c);
  if (rc) return rc;
  HK_REQUIRE(st, HK_E_INVALID, "settings is NULL");
  HK_REQUIRE(c->band_count <= (uint32_t)c->RH, HK_E_INVALID, "more bands than rows");
  HK_REQUIRE(st->taa <= HK_TAA_NONE && st->upscale_kind <= HK_UPSCALE_SMAA_TU4X, HK_E_INVALID, "bad taa / upscale_kind in settings");
  HK_REQUIRE(!(flags & HK_FRAME_INTERNAL_LATE_JOIN) || c->in_frame_render, HK_E_INVALID, "unknown frame flag");
  // HkSettings and the HkFrame of hk_frame_begin describe the same HikariSettings (view.rs:141-193): the fields both carry must agree
  HK_REQUIRE(st->indirect_bounces == c->frame.indirect_bounces && (st->temporal_reuse != 0u) == (c->frame.temporal_reuse != 0u) &&
                 (st->emissive_spatial_reuse != 0u) == (c->frame.emissive_spatial_reuse != 0u) &&
                 (st->indirect_spatial_reuse != 0u) == (c->frame.indirect_spatial_reuse != 0u),
             HK_E_INVALID, "HkSettings disagrees with the HkFrame given to hk_frame_begin (indirect_bounces / reuse flags)");
  c->taa = st->taa;
  c->upscale_kind = st->upscale_kind;
  c->upscale_sharpness = st->upscale_sharpness;
  uint32_t ub0, ub1;
  const uint32_t* bounds = c->band_bounds.size() == (size_t)c->band_count + 1 ? c->band_bounds.data() : nullptr;
  HK_REQUIRE(band_bounds_valid(bounds, c->band_count, (uint32_t)c->RH), HK_E_INVALID, "the band boundaries do not fit the current render size (hk_set_band_bounds after hk_resize)");
  band_rows_in(bounds, (uint32_t)c->RH, (uint32_t)c->RH, c->band_index, c->band_count, &ub0, &ub1);
  const int b0 = (int)ub0, b1 = (int)ub1;
  auto clampr = [&](int v) { return std::min(std::max(v, 0), c->RH); };
  const Aprons ap = band_aprons(st);
  const int den = (int)ap.denoise, sp = (int)ap.spatial;
#define HK_RUN(pass, arg, y0, y1)                    \
  do {                                               \
    int a_ = (y0), b_ = (y1);                        \
    if (b_ > a_ && (rc = run_pass(c, pass, arg, a_, b_))) return rc; \
  } while (0)
  if (stage == HK_STAGE_TEMPORAL) {
    // the side stream may still hold the previous frame's direct-light dispatches (hk_context.hpp side_done): this frame touches nothing
    // of theirs as long as the double-buffered planes flipped - a frame of the same parity, a host-written G-buffer, a context without
    // the twins (no post stream) waits for them
    {
      const uint32_t p_ = c->mapped_parity & 1u;
      const bool covered = c->side_state[p_] == 2 && c->side_cover[p_] == p_;   // (join_post_parity below then orders this frame behind that side work)
      if (c->forked && (!c->normal_twin || (flags & HK_FRAME_EXTERNAL_GBUFFER) || (c->side_state[p_] != 0 && !covered)) && (rc = join_side(c))) return rc;
    }
    // last frame's post-processing may still be running on post_stream: this frame's primary rays and light passes do not touch what
    // it reads - the planes both touch are double-buffered by frame parity.  What this frame does write again is what the last frame
    // OF ITS OWN PARITY read (normally two frames back and long done; the last frame itself when a host renders two frames of one
    // parity in a row); a host-rasterised G-buffer was written into whichever planes were mapped: everything first
    int f0, f1;
    full_rows_for(c, clampr(b0 - den - sp), clampr(b1 + den + sp), &f0, &f1);
    // Primary-ray pipelining (hk_context.hpp): what these rays write - the G-buffer planes of this frame's parity - was last read by frame
    // n - 2 (every plane has its twin; only the anti-aliasing tail reads the other parity's), and that frame's post-processing event
    // (post_done[parity]: the post stream waited for the main AND the side stream before it) closes all of it.  What they read - scene
    // memory - must not have been written since the last frame (those writes are behind the previous frame's spatial pass on the main
    // stream), and the wide trees must be current (k_build_wide would run on the main stream).
    const uint32_t parity = c->mapped_parity & 1u;
    const bool wide_clean = !wide_allowed(c) || (!c->wide_tlas_dirty && !c->wide_blas_dirty && !c->wide_mesh_check);
    const bool pre_wanted = c->prepass_pipeline < 0 ? HK_PREPASS_PIPELINE_RULE : c->prepass_pipeline != 0;
    const bool pre_pipelined = pre_wanted && c->pre_stream && c->stream == c->own_stream && c->normal_twin && c->pre_chain_ok && c->pre_last_parity != parity &&
                               !(flags & (HK_FRAME_EXTERNAL_GBUFFER | HK_FRAME_ANTIALIAS | HK_FRAME_TIME_BAND)) && (c->band_count == 1 || HK_PRE_BANDS) && !c->timing_mask && !(c->flags & HK_CTX_COUNT_RAYS) &&
                               wide_clean && !c->derived_dirty && c->scene_epoch == c->pre_seen_epoch && c->post_pending[parity] && f1 > f0 &&
                               (c->side_state[parity] == 0 || (c->side_state[parity] == 2 && c->side_cover[parity] == parity));   // (the side stream's readers of these planes are behind that event too)
    const bool scene_written = c->scene_epoch != c->pre_seen_epoch || !wide_clean;
    c->pre_seen_epoch = c->scene_epoch;
    if (pre_pipelined) {
      if (c->pre_scene_marked) HK_HIP(hipStreamWaitEvent(c->pre_stream, c->pre_scene_mark, 0));
      HK_HIP(hipStreamWaitEvent(c->pre_stream, c->post_done[parity], 0));   // (the main stream gets behind it through pre_done below)
      c->post_pending[parity] = false;
    } else if ((flags & HK_FRAME_EXTERNAL_GBUFFER) ? (rc = join_post(c)) : (rc = join_post_parity(c, c->mapped_parity))) {
      return rc;
    }
    bool albedo_done = false;
    if (c->timing_mask) (void)hipEventRecord(c->frame_start, c->stream);
    if (!(flags & HK_FRAME_EXTERNAL_GBUFFER)) {
      if (f1 > f0) {  // the prepass also fills the albedo of every pixel it covers (a superset of the rows albedo needs)
        const DFrame fr = make_dframe(c);
        GBuffer g = make_gbuffer(c);
        g.albedo_out = (uint2*)c->buf[HK_BUF_ALBEDO];
        unsigned long long* counters = (c->flags & HK_CTX_COUNT_RAYS) ? c->d_counters : nullptr;
        const Jitter j = prepass_jitter(c);
        hkd::WideTrees wide{};
        if ((rc = wide_for_fused(c, &wide))) return rc;
        {
          ScopedTimer timer(c, HK_PASS_PREPASS);
          launch_prepass(pre_pipelined ? c->pre_stream : c->stream, c->scene, fr, c->view.inverse_view_proj, c->view.view_proj, c->pview.view_proj, c->d_prev_models, j.x, j.y, g, f0, f1,
                         counters, &wide);
        }
        HK_HIP(hipGetLastError());
        if (pre_pipelined) {
          HK_HIP(hipEventRecord(c->pre_done, c->pre_stream));
          HK_HIP(hipStreamWaitEvent(c->stream, c->pre_done, 0));
          c->prepasses_pipelined += 1;
        } else if (c->pre_stream && scene_written) {
          // scene memory was written since the last frame - uploads and refits on the main stream behind the previous frame's spatial pass,
          // the wide records derived again just above: whichever later frame pipelines its primary rays (they may start as early as this
          // frame's own) must find all of it done
          HK_HIP(hipEventRecord(c->pre_scene_mark, c->stream));
          c->pre_scene_marked = true;
        }
        albedo_done = true;
      }
    } else if (f1 > f0) {  // host-rasterised G-buffer: only the derived planes are ours to fill
      launch_derive_planes(c->stream, make_gbuffer(c), c->depth_plane, c->dn_g, c->W, f0, f1);
      c->derived_dirty = false;
    }
    int a0, a1;
    full_rows_for(c, clampr(b0 - den), clampr(b1 + den), &a0, &a1);
    if (!albedo_done) HK_RUN(HK_PASS_FULL_SCREEN_ALBEDO, 0, a0, a1);  // light.rs:646-653
    if (c->side_stream && b1 > b0) {
      // sun and emissive share their spatial reservoir buffers (S = 4 for both, light.rs:518-546) and stay in order
      // on the side stream; indirect (T = 6, S = 8) and everything it feeds is independent of them until demodulation
      HK_HIP(hipEventRecord(c->fork_event, c->stream));
      HK_HIP(hipStreamWaitEvent(c->side_stream, c->fork_event, 0));
      c->forked = true;
      c->side_state[c->mapped_parity & 1u] = 1;
      if ((rc = run_pass_on_side(c, HK_PASS_DIRECT_LIT, 0, b0, b1))) return rc;
      if ((rc = run_pass_on_side(c, HK_PASS_DIRECT_EMISSIVE, 0, b0, b1))) return rc;
      HK_RUN(HK_PASS_INDIRECT, 0, b0, b1);
      // a band's exchange A ships the emissive temporal reservoirs when their spatial pass is on
      if (c->band_count > 1 && st->emissive_spatial_reuse && (rc = join_side(c))) return rc;
    } else {
      HK_RUN(HK_PASS_DIRECT_LIT, 0, b0, b1);          // light.rs:656-688
      HK_RUN(HK_PASS_DIRECT_EMISSIVE, 0, b0, b1);
      HK_RUN(HK_PASS_INDIRECT, 0, b0, b1);
    }
    c->pre_chain_ok = !(flags & (HK_FRAME_EXTERNAL_GBUFFER | HK_FRAME_ANTIALIAS));   // (the next frame's primary rays may run beside what follows of this one)
    c->pre_last_parity = parity;
  } else if (stage == HK_STAGE_SPATIAL) {            // light.rs:689-697
    if (parks_across_bands(c) && c->det_winner[0]) {
      // exchange A delivered the parked stores of the pixels up to 2 x history rows outside the band.  Per
      // channel, in dispatch order (sun and emissive store into the same buffer): the foreign pixels join the winners their
      // slots have so far, then every parked store that is its slot's winner is applied - own rows and foreign rows alike.
      // A channel whose spatial pass is off has no reader of its previous_spatial: its rows are not exchanged (hk_band_plan_for)
      // and its stores are resolved among the band's own pixels, so that the buffer holds what a band can know.
      const int reach = 2 * (int)c->history_now, r0 = clampr(b0 - reach), r1 = clampr(b1 + reach);
      if ((rc = join_side(c))) return rc;  // (the direct-light dispatches parked on the side stream)
      for (int channel = 0; channel < 3; ++channel) {
        const bool exchanged = channel == 2 ? st->indirect_spatial_reuse != 0 : st->emissive_spatial_reuse != 0;
        const LightTargets t = make_light_targets(c, channel);
        if (!t.det_winner) continue;
        if (exchanged) launch_resolve_scatter(c->stream, t, r0 * c->RW, r1 * c->RW, b0 * c->RW, b1 * c->RW);
        else launch_resolve_scatter(c->stream, t, b0 * c->RW, b1 * c->RW, 0, 0);
      }
      HK_HIP(hipGetLastError());
    }
    if (st->emissive_spatial_reuse) {
      if (c->forked) {
        if ((rc = run_pass_on_side(c, HK_PASS_EMISSIVE_SPATIAL_REUSE, 0, b0, b1))) return rc;
      } else {
        HK_RUN(HK_PASS_EMISSIVE_SPATIAL_REUSE, 0, b0, b1);
      }
    }
    if (st->indirect_spatial_reuse) HK_RUN(HK_PASS_INDIRECT_SPATIAL_REUSE, 0, b0, b1);
    // exchange B / demodulation read all three channels.  A single band's post-processing waits for the side stream ITSELF (post_begin);
    // a band's exchange B is enqueued by whoever drives the stages - on this stream, behind this join - unless hk_frame_render drives
    // them (HK_FRAME_INTERNAL_LATE_JOIN: its exchange B goes to the post stream with the rest)
    if ((c->band_count > 1 && !(flags & HK_FRAME_INTERNAL_LATE_JOIN)) && (rc = join_side(c))) return rc;
  } else if (stage == HK_STAGE_POST_PROCESS) {
    // (hk_frame_render on a band with a communicator has moved to the post stream already: exchange B belongs in front of demodulation)
    bool pipelined = c->post_forked;
    if (!pipelined && (rc = post_begin(c, st, &pipelined))) return rc;
    if (st->denoise) {                               // post_process.rs:1190-1224
      const uint32_t nch = st->indirect_bounces == 0 ? 2u : 3u;  // post_process.rs:949-954
      // the reference's per-channel loop, with the channels of each step fused into one launch.  Round 6: demodulation too runs on
      // the post stream - the render / variance planes it reads are double-buffered by frame parity, the next frame's light passes
      // write the other set - so a frame's main stream ends with its spatial pass (a band: with exchange A and its spatial pass)
      if (pipelined && !demod_on_post(c)) {  // (demodulation stays on the main stream, the levels follow it on the post stream)
        hipStream_t post = c->stream;
        c->stream = c->post_saved_main;
        rc = join_side(c);  // (demodulation on the main stream reads the direct-light channels)
        if (!rc) rc = join_post(c);  // the denoiser's internal planes: last frame's levels come first
        if (!rc) rc = run_demodulation_fused(c, nch, clampr(b0 - 15), clampr(b1 + 15));
        if (!rc && hipEventRecord(c->post_fork, c->stream) != hipSuccess) rc = HK_E_HIP;
        if (!rc && hipStreamWaitEvent(post, c->post_fork, 0) != hipSuccess) rc = HK_E_HIP;
        c->stream = post;
      } else {
        rc = run_demodulation_fused(c, nch, clampr(b0 - 15), clampr(b1 + 15));
      }
      if (!rc) rc = run_denoise_fused(c, nch, 0, clampr(b0 - 7), clampr(b1 + 7));
      if (!rc) rc = run_denoise_fused(c, nch, 1, clampr(b0 - 3), clampr(b1 + 3));
      if (!rc) rc = run_denoise_fused(c, nch, 2, clampr(b0 - 1), clampr(b1 + 1));
      if (!rc) rc = run_denoise_fused(c, nch, 3, b0, b1, true);  // + tone mapping (post_process.rs:1226-1234) in the same launch
      if (c->timing_mask) {
        (void)hipEventRecord(c->frame_stop, c->stream);
        c->frame_timed = true;
      }
      const int rc2 = post_end(c, pipelined);
      if (rc) return rc;
      if (rc2) return rc2;
    } else {
      if ((rc = join_side(c))) return rc;                         // (tone mapping reads all three channels)
      HK_RUN(HK_PASS_TONE_MAPPING, 0u, b0, b1);                   // post_process.rs:1226-1234
      if (c->timing_mask) {
        (void)hipEventRecord(c->frame_stop, c->stream);
        c->frame_timed = true;
      }
    }
    c->frames += 1;
  } else if (stage == HK_STAGE_ANTIALIAS) {            // post_process.rs:1236-1272
    if ((rc = join_post(c))) return rc;                // (reads the tone-mapped image)
    // band: TAA on the band's output rows; its input row beyond the border comes from the extrapolation of the
    // neighbouring quad row, which needs the SMAA samples one more row out (footprints: hk_band_plan_for, exchange D)
    const bool smaa = st->upscale_kind == HK_UPSCALE_SMAA_TU4X;
    if (smaa) {
      HK_RUN(HK_PASS_SMAA_TU4X, 0, clampr(b0 - 2), clampr(b1 + 2));
      HK_RUN(HK_PASS_SMAA_TU4X_EXTRAPOLATE, 0, clampr(b0 - 1), clampr(b1 + 1));
    }
    if (st->taa == HK_TAA_JASMINE) {
      int w, h;
      buffer_dims(c, HK_BUF_TAA_OUTPUT, &w, &h);
      const int scale = smaa ? 2 : 1;
      HK_RUN(HK_PASS_TAA_JASMINE, 0, std::min(h, scale * b0), b1 == c->RH ? h : std::min(h, scale * b1));
    }
  } else if (stage == HK_STAGE_UPSCALE) {              // post_process.rs:1277-1308
    if ((rc = join_post(c))) return rc;
    if (st->upscale_kind == HK_UPSCALE_FSR1) {
      uint32_t w0, w1;
      band_rows_in(bounds, (uint32_t)c->RH, (uint32_t)c->H, c->band_index, c->band_count, &w0, &w1);
      HK_RUN(HK_PASS_FSR_EASU, 0, std::max((int)w0 - 1, 0), std::min((int)w1 + 1, c->H));
      HK_RUN(HK_PASS_FSR_RCAS, 0, (int)w0, (int)w1);
    }
  } else {
    HK_REQUIRE(false, HK_E_INVALID, "unknown stage %u", stage);
  }
#undef HK_RUN
  return HK_OK;
}

int hk_frame_render(hk_ctx* c, const HkFrame* f, const HkView* v, const HkPreviousView* pv, const HkLights* l, const HkSettings* st, uint32_t flags) {
  int rc = hk_frame_begin(c, f, v, pv, l);
  if (rc) return rc;
  HK_REQUIRE(st, HK_E_INVALID, "settings is NULL");
  if (flags & HK_FRAME_BALANCE_BANDS) {
    c->taa = st->taa;  // (the sub-pixel jitter of the primary rays follows the settings: hk_frame_stage would set them only later)
    c->upscale_kind = st->upscale_kind;
    if ((rc = hk_balance_bands(c, 0, nullptr, 0))) return rc;
  }
  // with a communicator attached (hk_comm_init) the halo exchanges of the band plan run here, on the context's stream
  const bool ex = c->comm != nullptr && c->band_count > 1;
  struct InRender { hk_ctx* c; ~InRender() { c->in_frame_render = false; } } in_render_{c};
  c->in_frame_render = true;
  const uint32_t hist = c->history_now << 8;
  for (uint32_t s = 0; s <= HK_STAGE_POST_PROCESS; ++s) {
    if (ex && s == HK_STAGE_POST_PROCESS) {
      // exchange B feeds demodulation only: both leave the main stream (round 6) - the band's next frame does not wait for either
      bool pipelined = false;
      if ((rc = ready(c)) || (rc = post_begin(c, st, &pipelined))) return rc;
      c->post_forked = pipelined;
      if ((rc = comm_exchange(c, s, st))) {
        (void)post_end(c, pipelined);
        return rc;
      }
    } else if (ex && (s != HK_STAGE_TEMPORAL || hist) && (rc = comm_exchange(c, s <= HK_STAGE_SPATIAL ? (s | hist) : s, st))) {
      return rc;
    }
    const bool timed = (flags & HK_FRAME_TIME_BAND) && s <= HK_STAGE_SPATIAL;  // (behind the exchange: the wait for the neighbours is not the band's time)
    if (timed) {
      for (int k = 0; k < 4; ++k)
        if (!c->band_ev[k]) HK_HIP(hipEventCreate(&c->band_ev[k]));
      HK_HIP(hipEventRecord(c->band_ev[2 * s], c->stream));
    }
    // (whoever calls hk_frame_render cannot put an exchange of its own between two stages: the side stream is joined where something reads it)
    if ((rc = hk_frame_stage(c, s, st, flags | HK_FRAME_INTERNAL_LATE_JOIN))) {
      if (c->post_forked) (void)post_end(c, true);
      return rc;
    }
    if (timed) {
      if (s == HK_STAGE_TEMPORAL && (rc = join_side(c))) return rc;  // (the direct-light dispatches are the band's work too)
      HK_HIP(hipEventRecord(c->band_ev[2 * s + 1], c->stream));
      if (s == HK_STAGE_SPATIAL) c->band_timed = true;
    }
  }
  if (flags & HK_FRAME_ANTIALIAS) {
    if (ex && (rc = comm_exchange(c, HK_STAGE_ANTIALIAS | hist, st))) return rc;
    if ((rc = hk_frame_stage(c, HK_STAGE_ANTIALIAS, st, flags))) return rc;
    if (ex && st->upscale_kind == HK_UPSCALE_FSR1 && (rc = comm_exchange(c, HK_STAGE_UPSCALE, st))) return rc;
    if ((rc = hk_frame_stage(c, HK_STAGE_UPSCALE, st, flags))) return rc;
  }
  // rank 0 collects the finished image (the post stream's tone mapping has to be in before the rows leave)
  if (ex && (flags & HK_FRAME_GATHER)) {
    // (round 4: rank 0 collects the rows WHILE the next frame renders - the tone-mapped image is double-buffered by frame parity; with the
    // anti-aliasing tail, whose next frame reads this frame's outputs, the gather completes in stream order as before.  Round 6: the
    // rows leave behind the POST stream, where this frame's tone mapping was enqueued - the main stream waits for neither)
    const uint32_t parity = c->mapped_parity & 1u;
    if (!(flags & HK_FRAME_ANTIALIAS) && c->post_stream && c->post_pending[parity]) {
      hipStream_t main_stream = c->stream;
      c->stream = c->post_stream;
      rc = comm_gather(c, hk_final_buffer(st, flags), 0u, true);
      c->stream = main_stream;
      return rc;
    }
    if ((rc = join_all(c))) return rc;
    return comm_gather(c, hk_final_buffer(st, flags), 0u, !(flags & HK_FRAME_ANTIALIAS));
  }
  return HK_OK;
}

int hk_frame_wait(hk_ctx* c) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  HK_HIP(hipSetDevice(c->device));
  { int rc = join_all(c); if (rc) return rc; }
  HK_HIP(hipStreamSynchronize(c->stream));
  drain_timers(c);
  return HK_OK;
}

int hk_buffer_info(hk_ctx* c, uint32_t buffer, uint32_t* w, uint32_t* h, uint32_t* bpp) {
  HK_REQUIRE(c && buffer < HK_BUF_COUNT && buffer_bpp(buffer), HK_E_INVALID, "bad buffer id");
  int bw, bh;
  buffer_dims(c, buffer, &bw, &bh);
  if (w) *w = (uint32_t)bw;
  if (h) *h = (uint32_t)bh;
  if (bpp) *bpp = buffer_bpp(buffer);
  return HK_OK;
}
int hk_read_buffer(hk_ctx* c, uint32_t buffer, void* dst, size_t bytes) {
  HK_REQUIRE(c && dst && buffer < HK_BUF_COUNT && c->buf[buffer], HK_E_INVALID, "bad argument");
  HK_REQUIRE(bytes == buffer_logical_bytes(c, buffer), HK_E_INVALID, "size mismatch: buffer has %zu bytes", buffer_logical_bytes(c, buffer));
  HK_HIP(hipSetDevice(c->device));
  { int rc_ = join_all(c); if (rc_) return rc_; }
  HK_HIP(hipStreamSynchronize(c->stream));
  HK_HIP(hipMemcpy(dst, c->buf[buffer], bytes, hipMemcpyDeviceToHost));
  return HK_OK;
}
int hk_write_buffer(hk_ctx* c, uint32_t buffer, const void* src, size_t bytes) {
  HK_REQUIRE(c && src && buffer < HK_BUF_COUNT && c->buf[buffer], HK_E_INVALID, "bad argument");
  HK_REQUIRE(bytes == buffer_logical_bytes(c, buffer), HK_E_INVALID, "size mismatch: buffer has %zu bytes", buffer_logical_bytes(c, buffer));
  HK_HIP(hipSetDevice(c->device));
  { int rc_ = join_all(c); if (rc_) return rc_; }
  HK_HIP(hipStreamSynchronize(c->stream));
  HK_HIP(hipMemcpy(c->buf[buffer], src, bytes, hipMemcpyHostToDevice));
  if (buffer >= HK_BUF_RESERVOIR0 && buffer < HK_BUF_RESERVOIR0 + 10 && c->tile_meta[buffer - HK_BUF_RESERVOIR0]) {  // host-written reservoirs: tiles unknown
    const uint32_t k = buffer - HK_BUF_RESERVOIR0;
    HK_HIP(hipMemset(c->tile_meta[k], 0, (size_t)c->tiles_x * c->tiles_y * sizeof(TileMeta)));
    c->tile_meta_zero[k] = true;
  }
  if (buffer == HK_BUF_POSITION || buffer == HK_BUF_NORMAL || buffer == HK_BUF_INSTANCE_MATERIAL) c->derived_dirty = true;
  return HK_OK;
}
int hk_device_ptr(hk_ctx* c, uint32_t buffer, void** ptr, size_t* bytes) {
  HK_REQUIRE(c && ptr && buffer < HK_BUF_COUNT && c->buf[buffer], HK_E_INVALID, "bad argument");
  *ptr = c->buf[buffer];
  if (bytes) *bytes = c->buf_bytes[buffer];  // the ALLOCATION (independent of the upscale kind in effect), so a host may keep a view across settings changes
  return HK_OK;
}
int hk_set_stream(hk_ctx* c, void* s) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  HK_HIP(hipSetDevice(c->device));
  { int rc = join_all(c); if (rc) return rc; }
  HK_HIP(hipStreamSynchronize(c->stream));
  drain_timers(c);
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return HK_OK;
}
int hk_stream(hk_ctx* c, void** s) {
  HK_REQUIRE(c && s, HK_E_INVALID, "bad argument");
  *s = (void*)c->stream;
  return HK_OK;
}
int hk_set_timing_mask(hk_ctx* c, uint32_t mask) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  c->timing_mask = mask;
  return HK_OK;
}
int hk_get_stats(hk_ctx* c, HkStats* out) {
  HK_REQUIRE(c && out, HK_E_INVALID, "bad argument");
  HK_HIP(hipSetDevice(c->device));
  { int rc = join_all(c); if (rc) return rc; }
  HK_HIP(hipStreamSynchronize(c->stream));
  drain_timers(c);
  memset(out, 0, sizeof(*out));
  unsigned long long h[9] = {};  // ([8]: stack entries the wide walk dropped - counted by every context)
  HK_HIP(hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
  out->rays_primary = h[0];
  out->rays_tlas = h[1];
  out->rays_blas = h[2];
  out->walk_node_steps = h[3];
  out->walk_triangle_tests = h[4];
  out->walk_instance_entries = h[5];
  out->walk_closest_hits = h[6];
  out->walk_top_node_steps = h[7];
  out->wide_stack_lost = h[8];
  out->frames = c->frames;
  out->last_frame_ms = c->last_frame_ms;
  out->scene_mesh_builds = c->static_rebuilds;
  out->scene_instance_builds = c->dynamic_rebuilds;
  out->scene_async_instance_uploads = c->async_instance_uploads;
  out->scene_device_refits = c->device_refits;
  out->scene_device_tree_builds = c->device_tree_builds;
  for (int i = 0; i < HK_TIMING_SLOTS; ++i) {
    out->pass_ms_total[i] = c->slot_ms[i];
    out->pass_launches[i] = c->slot_launches[i];
  }
  return HK_OK;
}
int hk_indirect_schedule(hk_ctx* c, uint32_t* out) {
  HK_REQUIRE(c && out, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(c->have_frame, HK_E_NOT_READY, "hk_frame_begin has not been called");
  HK_HIP(hipSetDevice(c->device));
  { const int rc = finalize_scene(c); if (rc) return rc; }
  *out = use_wavefront(c) ? 1u : 0u;
  return HK_OK;
}

int hk_traversal_mode(hk_ctx* c, uint32_t* out, uint32_t* orderings) {
  HK_REQUIRE(c && out, HK_E_INVALID, "NULL argument");
  HK_HIP(hipSetDevice(c->device));
  { const int rc = finalize_scene(c); if (rc) return rc; }
  HK_REQUIRE(c->scene_mem, HK_E_NOT_READY, "no scene uploaded");
  *out = c->scene.flat_mode ? HK_TRAVERSAL_ONE_LEVEL : c->threaded ? (HK_TRAVERSAL_THREADED | (wide_allowed(c) ? HK_TRAVERSAL_WIDE : 0u)) : HK_TRAVERSAL_REFERENCE;
  if (orderings) *orderings = c->scene.flat_mode ? c->scene.flat_mask + 1u : c->threaded ? 8u : 1u;
  return HK_OK;
}

int hk_reset_stats(hk_ctx* c) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  HK_HIP(hipSetDevice(c->device));
  { const int rc_ = sync_all(c); if (rc_) return rc_; }
  drain_timers(c);
  HK_HIP(hipMemset(c->d_counters, 0, 9 * sizeof(unsigned long long)));
  c->frames = 0;
  for (int i = 0; i < HK_TIMING_SLOTS; ++i) {
    c->slot_ms[i] = 0.0;
    c->slot_launches[i] = 0;
  }
  return HK_OK;
}
}  // extern "C"
