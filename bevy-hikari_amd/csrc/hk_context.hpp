// hk_context.hpp - the context behind the C ABI (private to libhikari_hip.so): `struct hk_ctx` and what the translation units that
// work on it share.
//   context.hip       lifetime, screen-space resources, the frame graph (hk_frame_*, hk_pass_run), bands, buffers, statistics
//   scene_layout.hip  uploads and the layout conversion: reference-layout scene arrays -> the device's scene blob (finalize_scene)
//   scene_refit.hip   instance motion and instance-set changes on the device (hk_refit_scene_instances, hk_rebuild_scene_trees, ...)
//   probes.hip        measurement hooks (hk_measure_*, hk_debug_math)
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <iterator>
#include <chrono>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "hk_internal.hpp"
#include "hk_kernels.hpp"

#define HK_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ::hk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return HK_E_HIP;                                                                   \
    }                                                                                    \
  } while (0)

using namespace hk;   // (a private header: every translation unit that includes it works inside these two namespaces)
using namespace hkd;

namespace hk {

template <typename T>
struct DevArray {
  T* p = nullptr;
  size_t n = 0;
  int upload(const std::vector<T>& h) {
    if (p) { (void)hipFree(p); p = nullptr; }
    n = h.size();
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HK_HIP(hipMalloc((void**)&p, bytes));
    if (n) HK_HIP(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return HK_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// One device allocation for all scene arrays, each 16-B aligned.
struct Blob {
  std::vector<uint8_t> bytes;
  template <typename T>
  size_t add(const std::vector<T>& v) {
    size_t off = (bytes.size() + 15) & ~(size_t)15;
    bytes.resize(off + std::max<size_t>(v.size(), 1) * sizeof(T), 0);
    if (!v.empty()) memcpy(bytes.data() + off, v.data(), v.size() * sizeof(T));
    return off;
  }
};

struct TimedLaunch {
  uint32_t slot;
  hipEvent_t start, stop;
};

// IEEE minNum / maxNum with -0 < +0 (the numeric contract of hk_device_math.hpp) on the host
inline float hmin(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return signbit(a) ? a : b;
  return a < b ? a : b;
}
inline float hmax(float a, float b) {
  if (a != a) return b;
  if (b != b) return a;
  if (a == b) return signbit(a) ? b : a;
  return a > b ? a : b;
}
inline uint32_t hash_u32(uint32_t value) {  // utils.wgsl:15-24
  uint32_t state = value;
  state = state ^ 2747636419u;
  state = state * 2654435769u;
  state = state ^ (state >> 16u);
  state = state * 2654435769u;
  state = state ^ (state >> 16u);
  state = state * 2654435769u;
  return state;
}
inline float as_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

}  // namespace hk


// byte offsets of the arrays inside the instance-level region of the scene allocation
struct DynOffsets { size_t tlas, instances, prev_models, light_lo, light_hi, emissives, alias, materials, tex_info, srgb_lut, flat; uint32_t flat_count, flat_orderings; };

struct hk_ctx {
  int device = 0;
  uint32_t flags = 0;
  hipStream_t stream = nullptr;      // stream all work is enqueued on (own_stream unless hk_set_stream)
  hipStream_t own_stream = nullptr;
  bool own_stream_high = false;      // ... created at the device's highest stream priority (context.hip pick_main_stream)
  bool main_priority_decided = false; // ... by the rule, at the first hk_resize
  // Primary-ray pipelining (round 6; context.hip stage TEMPORAL): frame n's primary rays on a stream of their own, ordered only behind
  // what last touched the G-buffer planes of ITS parity - frame n - 2, whose post-processing event covers all three of that frame's
  // streams - so that they run beside frame n - 1's spatial pass instead of behind it.  The stream exists only where the main stream
  // sits in the high-priority queue pool (a fourth stream of the default priority then has a hardware queue to itself).  scene_epoch
  // counts every write to scene memory (uploads, refits, rebuilds: scene_layout.hip / scene_refit.hip); a frame whose scene was
  // written since the last one takes the serial order - those writes sit behind the previous frame's spatial pass on the main stream.
  hipStream_t pre_stream = nullptr;
  hipEvent_t pre_done = nullptr, pre_scene_mark = nullptr;   // pre_scene_mark: the main stream behind the latest writes to scene memory
  bool pre_scene_marked = false;
  int prepass_pipeline = -1;          // HK_DEBUG_OPT_PREPASS_PIPELINE: -1 the rule, 0 never, 1 whenever the order allows
  uint64_t scene_epoch = 0, pre_seen_epoch = 0;
  bool pre_chain_ok = false;          // the previous frame went through stage TEMPORAL with the other parity, without the AA tail
  uint32_t pre_last_parity = 0;
  uint64_t prepasses_pipelined = 0;
  int main_priority = -1;            // HK_DEBUG_OPT_MAIN_PRIORITY: -1 the rule, 0 default priority, 1 highest
  hipStream_t side_stream = nullptr;   // the direct-light dispatches of the frame path run here (unless HK_CTX_SINGLE_STREAM)
  hipEvent_t fork_event = nullptr, join_event = nullptr;
  bool forked = false;                 // side_stream holds work the main stream has not waited for yet
  // Round 6: the main stream no longer waits for the side stream at the end of a frame.  What reads the direct-light dispatches'
  // outputs is the post-processing (post stream: it waits for side_done itself); the NEXT frame's primary rays and indirect pass touch
  // nothing the direct-light dispatches of this frame read or write - the G-buffer planes are double-buffered by frame parity (normal
  // and instance_material included, below), the sun / emissive reservoirs are the side stream's own - so the side stream may run up
  // to the post-processing behind.  side_parity = the frame parity of the work it was last given (a frame of the SAME parity joins).
  hipEvent_t side_done = nullptr;
  // per frame parity: 0 = the side stream holds nothing of that parity the main stream has not been ordered behind, 1 = it does,
  // 2 = it does, and the post-processing of parity side_cover[.] waits for it (so waiting for THAT post-processing is enough)
  uint8_t side_state[2] = {0, 0};
  uint8_t side_cover[2] = {0, 0};
  void* normal_twin = nullptr;
  void* instance_material_twin = nullptr;
  // Frame pipelining (round 3; round 6: the whole post-processing).  Demodulation, the a-trous levels and tone mapping of frame n - and,
  // on a band with a communicator, the halo exchange B in front of them - run on a third stream, so that the main stream goes straight
  // on to frame n + 1's primary rays and light passes.  What both touch is double-buffered by frame parity: albedo, depth gradient, the
  // derived planes (dn_g, depth) the a-trous taps read, and (round 6) the render / variance planes the light passes write and
  // demodulation reads.  Stream order keeps the denoiser's own planes safe (post-processing n + 1 follows post-processing n); the main
  // stream waits for the post-processing of the LAST FRAME OF THE SAME PARITY before it writes that parity's planes again.
  hipStream_t post_stream = nullptr;
  hipEvent_t post_fork = nullptr;
  hipEvent_t post_done[2] = {nullptr, nullptr};  // post stream: end of the post-processing of the last frame of that parity
  bool post_pending[2] = {false, false};        // ... which the main stream has not waited for yet
  bool post_forked = false;                     // hk_frame_render moved the context to the post stream ahead of stage POST_PROCESS (exchange B goes there too)
  hipStream_t post_saved_main = nullptr;
  void* albedo_twin = nullptr;         // the planes of the OTHER frame parity (swapped with buf[HK_BUF_ALBEDO] ... in hk_frame_begin)
  void* depth_gradient_twin = nullptr;
  void* dn_g_twin = nullptr;
  void* render_twin[3] = {nullptr, nullptr, nullptr};
  void* variance_twin[3] = {nullptr, nullptr, nullptr};
  // test / measurement switches (hikari_hip_debug.h hk_debug_set_option; the library reads no environment variable)
  int spatial_window = -1;               // which form of k_spatial_reuse a launch takes: -1 by its size, 0 plain, 1 windowed (kernels.hip launch_spatial)
  uint64_t spatial_windowed_launches = 0;
  bool frame_pipeline = true;            // the a-trous levels of frame n beside frame n + 1's light passes (post_stream)
  int persistent_paths = -1;             // HK_DEBUG_OPT_PERSISTENT_PATHS: every bounce of the queue-based indirect pass in one launch (-1: the rule)
  bool wf_timeline = false;              // the instrumented twin of the trace kernels (tools/wf_timeline.py)
  bool flat_walk = true;                 // the one-level tree for scenes under one transform (scene_layout.hip)
  int flat_orderings = 0;                // ... with this many direction orderings (0: as many as fit 4 KB)
  bool trace_update = false;             // timings of scene updates on stderr
  bool in_frame_render = false;          // hk_frame_render is driving the stages (its internal frame flag is only valid then)
  bool side_join_each_frame = false;     // the main stream waits for the side stream before the post-processing of every frame (rounds 1-5)
  int post_demodulation = -1;            // demodulation on the post stream with the levels: -1 by the rule (context.hip demod_on_post), 0 no, 1 yes

  // host copies of the reference-layout scene (kept for the layout conversion)
  std::vector<HkVertex> vertices;
  std::vector<HkPrimitive> primitives;
  std::vector<HkNode> asset_nodes;
  std::vector<HkMaterial> materials;
  std::vector<HkInstance> instances;
  std::vector<HkNode> instance_nodes;
  std::vector<HkEmissive> emissives;
  std::vector<HkNode> emissive_nodes;
  std::vector<HkAliasEntry> alias_table;
  struct HostTexture { std::vector<uint32_t> texels; uint32_t w, h, flags; };
  std::vector<HostTexture> textures;
  bool have_meshes = false, have_materials = false, have_instances = false, have_noise = false;
  // what finalize_scene has to redo: the mesh-level region, the instance-level region, the texel buffer
  bool mesh_dirty = true, dynamic_dirty = true, textures_dirty = true;
  std::vector<float> prev_models;          // PreviousMeshUniform::transform per instance (optional)
  std::vector<int64_t> node_prim_offset;   // primitive offset each BLAS node's leaves index (from the last mesh-level build)

  // device scene
  uint8_t* scene_mem = nullptr;  // every scene array in one allocation (so small scenes can be staged in LDS by one copy loop)
  size_t dyn_capacity = 0, static_bytes = 0;
  // Scenes too big for the LDS copy keep TWO slots of the instance-level region, [slot 0][slot 1][mesh region]: an
  // instance-only update goes through pinned staging into the slot the frames in flight do NOT read, in stream
  // order, so neither the host nor the GPU waits (animated scenes must not stall on the host).
  bool two_slots = false;
  int slot = 0;
  Blob dyn_blob;                // the instance-level region as last laid out: kept so that a per-frame update re-uses warm pages
  std::vector<float4> tlas_tmp;  // (20 MB of fresh allocations per update cost more in page faults than the layout itself)
  bool trees_pending_on_device = false;  // hk_update_scene_instances: the trees about to be uploaded are stand-ins the device overwrites in
                                         // stream order - no point threading their orderings on the host
  bool threaded = false;  // eight direction-ordered flattenings of every TLAS / BLAS are stored (hikari_hip.h HK_CTX_EXACT_TRAVERSAL)
  uint8_t* staging[2] = {nullptr, nullptr};
  size_t staging_bytes[2] = {0, 0};
  hipEvent_t staging_done[2] = {nullptr, nullptr};
  bool staging_pending[2] = {false, false};
  uint64_t async_instance_uploads = 0;
  size_t st_nodes = 0, st_v0 = 0, st_v1 = 0, st_v2 = 0, st_vn = 0, st_vuv = 0;  // offsets inside the mesh-level region
  uint64_t static_rebuilds = 0, dynamic_rebuilds = 0;
  // Parked previous_spatial stores (HK_CTX_DETERMINISTIC_SCATTER; every band of a sharded frame with a history halo: SURVEY 8e
  // step 6), one set per light channel.  `to` and the records are the HK_BUF_PARKED_* planes (c->buf: bands exchange their rows),
  // the winners are private.  Allocated on first use (ensure_parked).
  int* det_winner[3] = {nullptr, nullptr, nullptr};
  // Round 6: a single context resolves the reference's scatter race deterministically BY DEFAULT, in the light form (hk_kernels.hpp
  // LightTargets::det_lite) - for the channels whose previous_spatial buffer has a reader (the spatial pass that is on); with
  // HK_CTX_DETERMINISTIC_SCATTER for all three; HK_CTX_RACING_SCATTER restores the reference's race.  det_lite_clean[k]: channel k's
  // winner plane holds -1 everywhere (the light form hands it back that way; the full form of a band does not).
  bool det_lite_clean[3] = {false, false, false};
  // uniform-tile store elision (hk_kernels.hpp TileMeta): one record per 8x8 tile per reservoir buffer; tile_meta_zero[k] = the
  // device array of buffer k is known to be all zero ("contents unknown" everywhere)
  TileMeta* tile_meta[10] = {};
  bool tile_meta_zero[10] = {};
  int tiles_x = 0, tiles_y = 0;
  uint32_t elide_serial = 0;
  // ---- instance motion on the device (hk_refit_scene_instances, kernels_scene.hip)
  DynOffsets dyn_off{};                   // where the arrays of the instance-level region are (the slot in use)
  float4 *rf_inst_lo = nullptr, *rf_inst_hi = nullptr, *rf_prev_models = nullptr;  // world AABB / previous model per instance
  uint32_t* rf_emissive_of_instance = nullptr;
  float* rf_alias_scratch = nullptr;
  size_t rf_instances = 0, rf_alias = 0;  // sizes the side arrays were allocated for
  bool rf_ready = false;                  // side arrays describe the scene as uploaded (cleared by every host-side rebuild)
  hkd::RefitUpdate* rf_updates[2] = {nullptr, nullptr};  // pinned, read by the kernel over PCIe
  size_t rf_updates_cap[2] = {0, 0};
  hipEvent_t rf_done[2] = {nullptr, nullptr};
  bool rf_pending[2] = {false, false};
  int rf_k = 0;
  std::vector<uint32_t> rf_last_moved;    // instances whose `moved` flag is set on the device
  bool mirrors_stale = false;             // the host copies of emissives / tree boxes no longer describe the device scene
  uint64_t device_refits = 0, device_tree_builds = 0;
  void* lbvh_scratch = nullptr;           // hk_rebuild_scene_trees
  size_t lbvh_scratch_cap = 0;
  const float4* d_prev_models = nullptr;  // 4 columns per instance, valid where DInstance::moved
  DevArray<uint32_t> d_noise;
  DevArray<uint32_t> d_tex_data;
  DScene scene{};

  // screen-space resources
  int W = 0, H = 0, RW = 0, RH = 0;
  int UW = 0, UH = 0;           // SMAA Tu4x output size, ceil(size * 2 / ratio) (post_process.rs:718-722)
  bool uv_fast = false;         // (k + 0.5) / size certified for div_by() on all four sizes (certify_uv_division)
  uint32_t mapped_parity = 0;   // frame parity whose planes the non-PREVIOUS ids of the double-buffered set name
  float ratio = 1.0f;
  void* buf[HK_BUF_COUNT] = {};
  size_t buf_bytes[HK_BUF_COUNT] = {};
  // private planes (no HkBuffer id): derived G-buffer planes and the denoiser's per-channel sets used
  // when all channels of a level run in one launch (the exposed internals hold the LAST channel, which
  // is what they hold after the reference's channel-by-channel loop)
  float* depth_plane = nullptr;       // position.w of the current frame's G-buffer (4-B taps)
  float* prev_depth_plane = nullptr;  // ... of the previous frame's (follows the frame parity like HK_BUF_PREVIOUS_POSITION)
  void* dn_g = nullptr;
  void* dn_extra[2][4] = {};
  float* dn_extra_var[2] = {};
  bool derived_dirty = false;
  // scratch of the queue-based schedule of indirect_lit_ambient (hikari_hip.h HK_CTX_WAVEFRONT): ONE allocation, carved into
  // the planes of hkd::WfBuffers on first use and again after hk_resize
  void* wf_mem = nullptr;
  void* wf_paths_mem = nullptr;          // the persistent schedule's planes per bounce + the waves' own lists (context.hip ensure_wavefront_paths)
  hkd::WfBuffers wf{};
  // wide trees of the trace stages (hk_kernels.hpp WideTrees): records derived ON THE DEVICE from ordering 0 of the trees the scene
  // blob holds, lazily - the mesh trees when the mesh-level region was rebuilt, the instance tree whenever it was uploaded, refit or
  // rebuilt (wide_*_dirty), in stream order right before the pass that walks them
  float4* wide_tlas = nullptr;
  float4* wide_blas = nullptr;
  uint32_t* wide_spill = nullptr;
  uint32_t* wide_tlas_rank = nullptr;   // [instance] / [primitive]: the position of its leaf in ordering 0 (the reference's tie rule)
  uint32_t* wide_blas_rank = nullptr;
  size_t wide_rank_instances = 0, wide_rank_primitives = 0;
  size_t wide_tlas_slots = 0, wide_blas_slots = 0, wide_spill_lanes = 0;
  bool wide_tlas_dirty = true, wide_blas_dirty = true;
  bool wide_mesh_check = true;                                  // the instance SET changed: a mesh no instance used before may have none yet
  std::vector<std::pair<uint32_t, uint32_t>> wide_meshes;       // (node_offset, node_count) of the mesh trees whose records exist, sorted
  int compute_units = 0;

  // uniforms
  HkFrame frame{};
  HkView view{};
  HkPreviousView pview{};
  HkLights lights{};
  bool have_frame = false;
  uint32_t taa = HK_TAA_JASMINE, upscale_kind = HK_UPSCALE_SMAA_TU4X;
  float upscale_sharpness = 0.0f;

  uint32_t band_index = 0, band_count = 1;
  std::vector<uint32_t> band_bounds;   // explicit split of the scaled render rows (hk_set_band_bounds): band_count + 1 entries, or empty = equal split
  uint32_t bounds_generation = 0;
  void* comm = nullptr;        // RCCL communicator state, owned by comm.cpp (hk_comm_init)
  uint32_t history_rows = HK_HISTORY_AUTO;  // exchange C rows asked for (hk_set_history_rows): a count, or derived per frame
  uint32_t history_now = 0;    // ... in force for the frame most recently begun (0 for a single band)
  hipEvent_t band_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // HK_FRAME_TIME_BAND: around stage TEMPORAL, around stage SPATIAL (main stream)
  bool band_timed = false;

  // statistics
  unsigned long long* d_counters = nullptr;  // primary, tlas, blas, node steps, triangle tests, instance entries, closest hits (hk_light.hpp flush_counters)
  uint64_t frames = 0;
  uint32_t timing_mask = 0;
  std::vector<TimedLaunch> pending;
  std::vector<hipEvent_t> event_pool;
  double slot_ms[HK_TIMING_SLOTS] = {};
  uint64_t slot_launches[HK_TIMING_SLOTS] = {};
  hipEvent_t frame_start = nullptr, frame_stop = nullptr;
  bool frame_timed = false;
  float last_frame_ms = 0.0f;
};

namespace hk {
// ---- context.hip
int join_side(hk_ctx* c);   // the main stream waits for what was enqueued on the side stream (direct-light dispatches)
int join_post(hk_ctx* c);   // ... for the a-trous levels of the last frame (post_stream)
int join_all(hk_ctx* c);
// ---- scene_layout.hip
// wait for everything the context has enqueued, on ALL streams
int sync_all(hk_ctx* c);
// bring the device scene up to date with the host-side arrays (no-op when nothing is dirty)
int finalize_scene(hk_ctx* c);
void update_shared_transform(hk_ctx* c);
void point_scene_at_slot(hk_ctx* c);
// ---- scene_refit.hip
void free_refit(hk_ctx* c);
}  // namespace hk
