// context.hip - the C ABI of libhikari_hip.so: context, uploads, per-frame dispatch order.
//
// Reference call sites this file stands in for (cryscan/bevy-hikari v0.3.15):
//   uploads        src/mesh_material/mesh.rs:43-64, material.rs:201-202, instance.rs:82-108, src/lib.rs:189-219
//   resources      src/light.rs:307-383 (render/variance/albedo textures, 10 reservoir buffers),
//                  src/prepass.rs:285-318 (G-buffer), src/post_process.rs:621-633 (denoise textures)
//   dispatch order src/prepass.rs:769-852, src/light.rs:590-702, src/post_process.rs:1190-1234
//   ping-pong      src/light.rs:376,480-481,518-546
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#include "hk_internal.hpp"
#include "hk_kernels.hpp"

using namespace hk;
using namespace hkd;

#define HK_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ::hk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return HK_E_HIP;                                                                   \
    }                                                                                    \
  } while (0)

namespace {

__global__ void k_copy_u4(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

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

}  // namespace

// byte offsets of the arrays inside the instance-level region of the scene allocation
struct DynOffsets { size_t tlas, instances, prev_models, light_lo, light_hi, emissives, alias, materials, tex_info, srgb_lut, flat; uint32_t flat_count, flat_orderings; };

struct hk_ctx {
  int device = 0;
  uint32_t flags = 0;
  hipStream_t stream = nullptr;      // stream all work is enqueued on (own_stream unless hk_set_stream)
  hipStream_t own_stream = nullptr;
  hipStream_t side_stream = nullptr;   // the direct-light dispatches of the frame path run here (unless HK_CTX_SINGLE_STREAM)
  hipEvent_t fork_event = nullptr, join_event = nullptr;
  bool forked = false;                 // side_stream holds work the main stream has not waited for yet
  // Frame pipelining (round 3): the a-trous levels + tone mapping of frame n run on a third stream, so that the main stream goes
  // straight on to frame n + 1's primary rays and light passes (which read none of the denoiser's buffers).  What both touch is
  // double-buffered by frame parity: albedo, depth gradient and the derived planes (dn_g, depth) the a-trous taps read.
  hipStream_t post_stream = nullptr;
  hipEvent_t post_fork = nullptr, post_done = nullptr;
  bool post_pending = false;           // post_stream holds work the main stream has not waited for yet
  uint32_t post_parity = 0;            // mapped_parity of the frame whose a-trous levels are (were last) on post_stream
  void* albedo_twin = nullptr;         // the planes of the OTHER frame parity (swapped with buf[HK_BUF_ALBEDO] ... in hk_frame_begin)
  void* depth_gradient_twin = nullptr;
  void* dn_g_twin = nullptr;

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
  hkd::WfBuffers wf{};
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

namespace {

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

int free_screen(hk_ctx* c) {
  for (int k = 0; k < 3; ++k) {
    if (c->det_winner[k]) (void)hipFree(c->det_winner[k]);
    c->det_winner[k] = nullptr;
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
  if (c->wf.timeline) (void)hipFree(c->wf.timeline);
  c->wf = hkd::WfBuffers{};
  if (c->depth_plane) (void)hipFree(c->depth_plane);
  if (c->prev_depth_plane) (void)hipFree(c->prev_depth_plane);
  c->prev_depth_plane = nullptr;
  if (c->dn_g) (void)hipFree(c->dn_g);
  c->depth_plane = nullptr;
  c->dn_g = nullptr;
  for (void** q : {&c->albedo_twin, &c->depth_gradient_twin, &c->dn_g_twin}) {
    if (*q) (void)hipFree(*q);
    *q = nullptr;
  }
  c->post_pending = false;
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

// `bvh` 0.7.1's flat format puts a "navigator" node (child box, entry = next) in front of EVERY subtree,
// including single-leaf subtrees, and the reference then tests the leaf's own (re-derived) box again:
// two steps with the same box for every leaf reached.  With leaf boxes filled in at upload, a navigator
// whose subtree is one leaf with an equal box can take over the leaf's role (entry := leaf entry): the
// walk performs box test -> leaf action -> continue at the same exit index, i.e. exactly the outcomes
// of the two-step sequence, and the original leaf slot is simply never visited.  No index changes.
// entry/exit are LOCAL to [begin, begin + count).  Returns the number of folded navigators.
size_t fold_leaf_navigators(std::vector<float4>& lo, std::vector<float4>& hi, size_t begin, size_t count) {
  auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  size_t folded = 0;
  for (size_t k = 0; k + 1 < count; ++k) {
    float4& nlo = lo[begin + k];
    const float4& nhi = hi[begin + k];
    const uint32_t entry = bits(nlo.w), exit_ = bits(nhi.w);
    if (entry >= HK_BVH_LEAF_FLAG || entry != k + 1) continue;
    const float4& clo = lo[begin + k + 1];
    const float4& chi = hi[begin + k + 1];
    if (bits(clo.w) < HK_BVH_LEAF_FLAG || bits(chi.w) != exit_) continue;
    if (!(nlo.x == clo.x && nlo.y == clo.y && nlo.z == clo.z && nhi.x == chi.x && nhi.y == chi.y && nhi.z == chi.z)) continue;
    nlo.w = clo.w;
    ++folded;
  }
  return folded;
}

// Convert the reference-layout scene to the device layout (hk_device.hpp header comment).
//
// One device allocation, two regions:
//   [ instance-level region, `dyn_capacity` bytes ][ mesh-level region, `static_bytes` bytes ]
// The instance-level region (TLAS nodes first, instances, light BVH, emissives, alias tables,
// materials, texture descriptors) is what prepare_instances / prepare_material_assets rewrite when
// something moves (instance.rs:352-437); it is small (0.6 MB at 2 000 instances) and is the only part
// rebuilt and re-sent for an instance-only change.  The mesh-level region (BLAS nodes with their leaf
// boxes, triangle planes, vertex planes) changes only with the mesh assets (mesh.rs:106-166).
// Node indices are in 32-B units from the start of the allocation: TLAS node i is node i, BLAS node k
// of a mesh is node blas_base + node_offset + k with blas_base = dyn_capacity / 32.

// `orderings` flattenings of the reference-layout array `src` (ordering 0 = the reference's own order), each range
// [offset, offset + count) of `ranges` re-threaded on its own (hk_bvh_rethread); a malformed range keeps the reference order
void thread_orderings(const std::vector<HkNode>& src, const std::vector<std::pair<uint32_t, uint32_t>>& ranges, int orderings, std::vector<std::vector<HkNode>>& out) {
  out.assign((size_t)orderings, std::vector<HkNode>());
  out[0] = src;
  if (orderings <= 1) return;
  auto rethread = [&](int o) {
    for (const auto& r : ranges)
      if (r.second && !rethread_flat_bvh(src.data() + r.first, r.second, (uint32_t)o, out[o].data() + r.first))
        std::copy(src.begin() + r.first, src.begin() + r.first + r.second, out[o].begin() + r.first);
  };
  for (int o = 1; o < orderings; ++o) out[o] = src;
  // Small trees (the instance tree of an animated frame, a few thousand nodes) are re-threaded on the calling thread: spawning
  // seven threads costs more than the work and sits on the per-frame path.  Large mesh trees use worker threads; a thread that
  // cannot be created, or a worker that throws (bad_alloc), must not escape through the extern "C" boundary: the orderings it did
  // not produce are redone serially here.
  size_t total = 0;
  for (const auto& r : ranges) total += r.second;
  std::vector<uint8_t> done((size_t)orderings, 0);
  if (total >= 8192) {
    std::vector<std::thread> workers;
    try {
      for (int o = 1; o < orderings; ++o)
        workers.emplace_back([&, o]() {
          try {
            rethread(o);
            done[(size_t)o] = 1;
          } catch (...) {
          }
        });
    } catch (...) {
    }
    for (std::thread& w : workers) w.join();
  }
  for (int o = 1; o < orderings; ++o)
    if (!done[(size_t)o]) {
      out[o] = src;
      rethread(o);
    }
}

// mesh-level region; fills c->node_prim_offset.  Needs the instances' mesh records to know which
// primitive range a BLAS leaf indexes (GpuMeshIndex travels with the instance, mod.rs:147-156).
int build_static_region(hk_ctx* c, Blob& blob, size_t& off_nodes, size_t& off_v0, size_t& off_v1, size_t& off_v2, size_t& off_vn, size_t& off_vuv) {
  const size_t n_nodes = c->asset_nodes.size(), n_prims = c->primitives.size(), n_verts = c->vertices.size();
  std::vector<int64_t>& node_prim_offset = c->node_prim_offset;
  node_prim_offset.assign(n_nodes, -1);
  for (const HkInstance& in : c->instances)
    for (uint32_t k = 0; k < in.mesh.node_count; ++k) node_prim_offset[in.mesh.node_offset + k] = in.mesh.primitive;
  std::vector<std::pair<uint32_t, uint32_t>> ranges;  // distinct mesh ranges
  {
    std::vector<uint8_t> done(n_nodes + 1, 0);
    for (const HkInstance& in : c->instances) {
      if (in.mesh.node_count == 0 || done[in.mesh.node_offset]) continue;
      done[in.mesh.node_offset] = 1;
      ranges.emplace_back(in.mesh.node_offset, in.mesh.node_count);
    }
  }
  const int orderings = c->threaded ? 8 : 1;
  std::vector<std::vector<HkNode>> ordered;
  thread_orderings(c->asset_nodes, ranges, orderings, ordered);
  std::vector<float4> nodes;
  nodes.reserve(2 * n_nodes * (size_t)orderings);
  std::vector<float4> lo(n_nodes), hi(n_nodes);
  for (int o = 0; o < orderings; ++o) {
    const std::vector<HkNode>& src = ordered[o];
    for (size_t i = 0; i < n_nodes; ++i) {
      const HkNode& n = src[i];
      float mn[3] = {n.min[0], n.min[1], n.min[2]}, mx[3] = {n.max[0], n.max[1], n.max[2]};
      if (n.entry_index >= HK_BVH_LEAF_FLAG && node_prim_offset[i] >= 0) {  // light.wgsl:408-412
        size_t prim = (size_t)node_prim_offset[i] + (n.entry_index - HK_BVH_LEAF_FLAG);
        HK_REQUIRE(prim < n_prims, HK_E_INVALID, "BLAS leaf primitive out of bounds");
        const HkPrimitiveVertex* v = c->primitives[prim].vertices;
        for (int k = 0; k < 3; ++k) {
          mn[k] = hmin(v[0].position[k], hmin(v[1].position[k], v[2].position[k]));
          mx[k] = hmax(v[0].position[k], hmax(v[1].position[k], v[2].position[k]));
        }
      }
      lo[i] = make_float4(mn[0], mn[1], mn[2], as_f(n.entry_index));
      hi[i] = make_float4(mx[0], mx[1], mx[2], as_f(n.exit_index));
    }
    for (const auto& r : ranges) fold_leaf_navigators(lo, hi, r.first, r.second);  // fold single-leaf navigators, once per distinct mesh range
    for (size_t i = 0; i < n_nodes; ++i) { nodes.push_back(lo[i]); nodes.push_back(hi[i]); }
  }
  off_nodes = blob.add(nodes);  // offset 0: the region itself starts on a 32-B boundary

  std::vector<float4> v0(n_prims), v1(n_prims), v2(n_prims);
  for (size_t i = 0; i < n_prims; ++i) {
    const HkPrimitiveVertex* v = c->primitives[i].vertices;
    v0[i] = make_float4(v[0].position[0], v[0].position[1], v[0].position[2], as_f(v[0].index));
    v1[i] = make_float4(v[1].position[0], v[1].position[1], v[1].position[2], as_f(v[1].index));
    v2[i] = make_float4(v[2].position[0], v[2].position[1], v[2].position[2], as_f(v[2].index));
  }
  off_v0 = blob.add(v0);
  off_v1 = blob.add(v1);
  off_v2 = blob.add(v2);
  std::vector<float4> vn(n_verts);
  std::vector<float2> vuv(n_verts);
  for (size_t i = 0; i < n_verts; ++i) {
    vn[i] = make_float4(c->vertices[i].normal[0], c->vertices[i].normal[1], c->vertices[i].normal[2], 0.0f);
    vuv[i] = make_float2(c->vertices[i].u, c->vertices[i].v);
  }
  off_vn = blob.add(vn);
  off_vuv = blob.add(vuv);
  blob.bytes.resize((blob.bytes.size() + 15) & ~(size_t)15, 0);
  return HK_OK;
}

// ------------------------------------------------------------------ one-level BVH (DScene::flat, hk_device.hpp traverse_flat)
// Built on the host at every instance-level rebuild of a scene whose instances all share one transform and that fits the LDS
// copy: every triangle of every instance (local space = the one space they share), a top-down SAH build with an exact sweep
// along the three axes (the scenes are a few hundred triangles at most), one triangle per leaf, flattened depth-first with
// skip links once per ray-direction octant - each inner node's children in the order a ray of that octant meets them (axis
// of the larger centre separation), so a closest-hit walk finds its hit early and skips the rest by their boxes.
namespace flatbvh {
struct Tri { float lo[3], hi[3], c[3]; uint32_t prim, inst; };
struct Node { float lo[3], hi[3]; int left = -1, right = -1; uint32_t prim = 0, inst = 0; };
inline float half_area(const float* lo, const float* hi) {
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx * dy + dy * dz + dz * dx;
}
int build(std::vector<Node>& nodes, std::vector<Tri>& t, int b, int e) {
  const int id = (int)nodes.size();
  nodes.emplace_back();
  {
    Node& n = nodes[id];
    for (int k = 0; k < 3; ++k) { n.lo[k] = t[b].lo[k]; n.hi[k] = t[b].hi[k]; }
    for (int i = b + 1; i < e; ++i)
      for (int k = 0; k < 3; ++k) { n.lo[k] = std::min(n.lo[k], t[i].lo[k]); n.hi[k] = std::max(n.hi[k], t[i].hi[k]); }
  }
  if (e - b == 1) {
    nodes[id].prim = t[b].prim;
    nodes[id].inst = t[b].inst;
    return id;
  }
  const int n = e - b;
  double best = 1e300;
  int best_axis = 0, best_split = n / 2;
  std::vector<float> right_area((size_t)n);
  for (int axis = 0; axis < 3; ++axis) {
    std::stable_sort(t.begin() + b, t.begin() + e, [axis](const Tri& x, const Tri& y) { return x.c[axis] < y.c[axis]; });
    float lo[3], hi[3];
    for (int i = n - 1; i >= 1; --i) {  // right_area[i] = area of the box of t[b + i .. e)
      const Tri& q = t[b + i];
      for (int k = 0; k < 3; ++k) {
        lo[k] = i == n - 1 ? q.lo[k] : std::min(lo[k], q.lo[k]);
        hi[k] = i == n - 1 ? q.hi[k] : std::max(hi[k], q.hi[k]);
      }
      right_area[(size_t)i] = half_area(lo, hi);
    }
    for (int i = 1; i < n; ++i) {  // split: [b, b + i) | [b + i, e)
      const Tri& q = t[b + i - 1];
      for (int k = 0; k < 3; ++k) {
        lo[k] = i == 1 ? q.lo[k] : std::min(lo[k], q.lo[k]);
        hi[k] = i == 1 ? q.hi[k] : std::max(hi[k], q.hi[k]);
      }
      const double cost = (double)half_area(lo, hi) * i + (double)right_area[(size_t)i] * (n - i);
      if (cost < best) { best = cost; best_axis = axis; best_split = i; }
    }
  }
  std::stable_sort(t.begin() + b, t.begin() + e, [best_axis](const Tri& x, const Tri& y) { return x.c[best_axis] < y.c[best_axis]; });
  const int l = build(nodes, t, b, b + best_split);
  const int r = build(nodes, t, b + best_split, e);
  nodes[id].left = l;
  nodes[id].right = r;
  return id;
}
// depth-first flattening for direction octant `oct` (only the bits of `mask` are distinguished); returns the index after the subtree
uint32_t emit(const std::vector<Node>& nodes, int id, uint32_t oct, uint32_t mask, std::vector<float4>& out, uint32_t at) {
  const Node& n = nodes[id];
  if (n.left < 0) {
    out[2 * at] = make_float4(n.lo[0], n.lo[1], n.lo[2], as_f(HK_BVH_LEAF_FLAG | n.prim));
    out[2 * at + 1] = make_float4(n.hi[0], n.hi[1], n.hi[2], as_f((at + 1u) | (n.inst << 16)));
    return at + 1u;
  }
  const Node &a = nodes[n.left], &b = nodes[n.right];
  int axis = 0;
  float sep = -1.0f;
  for (int k = 0; k < 3; ++k) {
    const float d = std::fabs((b.lo[k] + b.hi[k]) - (a.lo[k] + a.hi[k]));
    if (d > sep) { sep = d; axis = k; }
  }
  const bool a_smaller = (a.lo[axis] + a.hi[axis]) <= (b.lo[axis] + b.hi[axis]);
  const bool negative = ((oct & mask) >> axis) & 1u;           // the ray travels towards smaller coordinates on this axis
  const bool a_first = negative ? !a_smaller : a_smaller;
  uint32_t next = emit(nodes, a_first ? n.left : n.right, oct, mask, out, at + 1u);
  next = emit(nodes, a_first ? n.right : n.left, oct, mask, out, next);
  out[2 * at] = make_float4(n.lo[0], n.lo[1], n.lo[2], as_f(at + 1u));
  out[2 * at + 1] = make_float4(n.hi[0], n.hi[1], n.hi[2], as_f(next));
  return next;
}
}  // namespace flatbvh

// Fills `out` with `orderings` flattenings of (2 T - 1) nodes each; returns false when the scene does not qualify.
bool build_flat_bvh(const hk_ctx* c, uint32_t orderings, std::vector<float4>& out, uint32_t& count) {
  using namespace flatbvh;
  std::vector<Tri> tris;
  for (size_t i = 0; i < c->instances.size(); ++i) {
    const HkInstance& in = c->instances[i];
    for (uint32_t k = 0; k < in.mesh.node_count; ++k) {
      const HkNode& nd = c->asset_nodes[in.mesh.node_offset + k];
      if (nd.entry_index < HK_BVH_LEAF_FLAG) continue;
      const size_t prim = (size_t)in.mesh.primitive + (nd.entry_index - HK_BVH_LEAF_FLAG);
      if (prim >= c->primitives.size() || prim > 0xFFFFu) return false;
      Tri t;
      const HkPrimitiveVertex* v = c->primitives[prim].vertices;
      for (int a = 0; a < 3; ++a) {
        t.lo[a] = hmin(v[0].position[a], hmin(v[1].position[a], v[2].position[a]));  // = the BLAS leaf box (light.wgsl:408-412)
        t.hi[a] = hmax(v[0].position[a], hmax(v[1].position[a], v[2].position[a]));
        t.c[a] = 0.5f * (t.lo[a] + t.hi[a]);
        if (!(t.lo[a] == t.lo[a]) || !(t.hi[a] == t.hi[a])) return false;  // NaN vertices: leave the scene to the reference walk
      }
      t.prim = (uint32_t)prim;
      t.inst = (uint32_t)i;
      tris.push_back(t);
    }
  }
  if (tris.empty() || tris.size() > 0x7FFFu) return false;
  std::vector<Node> nodes;
  nodes.reserve(2 * tris.size());
  build(nodes, tris, 0, (int)tris.size());
  count = (uint32_t)nodes.size();
  out.assign((size_t)orderings * count * 2, make_float4(0, 0, 0, 0));
  std::vector<float4> one((size_t)count * 2);
  for (uint32_t o = 0; o < orderings; ++o) {
    if (emit(nodes, 0, o, orderings - 1u, one, 0u) != count) return false;
    std::copy(one.begin(), one.end(), out.begin() + (size_t)o * count * 2);
  }
  return true;
}

int build_dynamic_region(hk_ctx* c, Blob& blob, DynOffsets& o, size_t static_bytes) {
  const size_t n_tlas = c->instance_nodes.size();
  const int orderings = c->threaded ? 8 : 1;
  std::vector<std::vector<HkNode>> ordered;
  // hk_update_scene_instances: the device is about to build every ordering of this tree in stream order - ordering 0 is laid
  // out (its leaf boxes are where the device build reads the instances' boxes from), the other seven slots stay zero
  const int host_orderings = c->trees_pending_on_device ? 1 : orderings;
  if (c->trees_pending_on_device) ordered.assign(1, c->instance_nodes);
  else thread_orderings(c->instance_nodes, {{0u, (uint32_t)n_tlas}}, orderings, ordered);
  std::vector<float4> tlo(n_tlas), thi(n_tlas);
  std::vector<float4>& tlas = c->tlas_tmp;
  tlas.clear();
  tlas.reserve(2 * n_tlas * (size_t)orderings);
  for (int ord = 0; ord < host_orderings; ++ord) {
    for (size_t i = 0; i < n_tlas; ++i) {
      const HkNode& n = ordered[ord][i];
      const float* mn = n.min;
      const float* mx = n.max;
      if (n.entry_index >= HK_BVH_LEAF_FLAG) {  // light.wgsl:454-457: the leaf box is the instance's world AABB
        uint32_t inst = n.entry_index - HK_BVH_LEAF_FLAG;
        HK_REQUIRE(inst < c->instances.size(), HK_E_INVALID, "TLAS leaf instance out of bounds");
        mn = c->instances[inst].min;
        mx = c->instances[inst].max;
      }
      tlo[i] = make_float4(mn[0], mn[1], mn[2], as_f(n.entry_index));
      thi[i] = make_float4(mx[0], mx[1], mx[2], as_f(n.exit_index));
    }
    fold_leaf_navigators(tlo, thi, 0, n_tlas);
    for (size_t i = 0; i < n_tlas; ++i) { tlas.push_back(tlo[i]); tlas.push_back(thi[i]); }
  }
  tlas.resize(2 * n_tlas * (size_t)orderings, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
  o.tlas = blob.add(tlas);  // offset 0: ordering `ord` starts at node ord * n_tlas

  const bool have_prev = c->prev_models.size() == 16 * c->instances.size();
  std::vector<DInstance> di(c->instances.size());
  std::vector<float4> pm;
  bool any_moved = false;
  for (size_t i = 0; i < di.size(); ++i) {
    const HkInstance& in = c->instances[i];
    const float* t = in.inverse_transpose_model;
    const float* m = in.model;
    DInstance& d = di[i];
    d.im0 = make_float4(t[0], t[4], t[8], t[12]);  // column j of transpose(itm) = row j of itm
    d.im1 = make_float4(t[1], t[5], t[9], t[13]);
    d.im2 = make_float4(t[2], t[6], t[10], t[14]);
    d.im3 = make_float4(t[3], t[7], t[11], t[15]);
    d.m0 = make_float4(m[0], m[1], m[2], m[3]);
    d.m1 = make_float4(m[4], m[5], m[6], m[7]);
    d.m2 = make_float4(m[8], m[9], m[10], m[11]);
    d.m3 = make_float4(m[12], m[13], m[14], m[15]);
    d.n0 = make_float4(t[0], t[1], t[2], 0.0f);
    d.n1 = make_float4(t[4], t[5], t[6], 0.0f);
    d.n2 = make_float4(t[8], t[9], t[10], 0.0f);
    d.material = in.material;
    d.vertex = in.mesh.vertex;
    d.primitive = in.mesh.primitive;
    d.node_offset = in.mesh.node_offset;
    d.node_count = in.mesh.node_count;
    d.moved = (have_prev && memcmp(&c->prev_models[16 * i], m, 64) != 0) ? 1u : 0u;
    any_moved = any_moved || d.moved;
    d.pad1 = d.pad2 = 0;
  }
  if (any_moved) {  // previous model matrices, 4 columns per instance (only when something moves)
    pm.resize(4 * di.size());
    for (size_t i = 0; i < di.size(); ++i)
      for (int col = 0; col < 4; ++col) {
        const float* q = &c->prev_models[16 * i + 4 * col];
        pm[4 * i + col] = make_float4(q[0], q[1], q[2], q[3]);
      }
  }
  o.instances = blob.add(di);
  o.prev_models = blob.add(pm);

  const size_t n_light = c->emissive_nodes.size();
  std::vector<float4> llo(n_light), lhi(n_light);
  for (size_t i = 0; i < n_light; ++i) {
    const HkNode& n = c->emissive_nodes[i];
    float mn[3] = {n.min[0], n.min[1], n.min[2]}, mx[3] = {n.max[0], n.max[1], n.max[2]};
    if (n.entry_index >= HK_BVH_LEAF_FLAG) {  // light.wgsl:633-636: position -/+ radius
      uint32_t e = n.entry_index - HK_BVH_LEAF_FLAG;
      HK_REQUIRE(e < c->emissives.size(), HK_E_INVALID, "light BVH leaf out of bounds");
      for (int k = 0; k < 3; ++k) {
        mn[k] = c->emissives[e].position[k] - c->emissives[e].radius;
        mx[k] = c->emissives[e].position[k] + c->emissives[e].radius;
      }
    }
    llo[i] = make_float4(mn[0], mn[1], mn[2], as_f(n.entry_index));
    lhi[i] = make_float4(mx[0], mx[1], mx[2], as_f(n.exit_index));
  }
  fold_leaf_navigators(llo, lhi, 0, n_light);
  o.light_lo = blob.add(llo);
  o.light_hi = blob.add(lhi);

  std::vector<DEmissive> de(c->emissives.size());
  for (size_t i = 0; i < de.size(); ++i) {
    const HkEmissive& e = c->emissives[i];
    HK_REQUIRE(e.instance < c->instances.size(), HK_E_INVALID, "emissive instance out of bounds");
    HK_REQUIRE((size_t)e.alias_table[0] + e.alias_table[1] <= c->alias_table.size() && e.alias_table[1] > 0, HK_E_INVALID, "emissive alias slice out of bounds");
    de[i].position_radius = make_float4(e.position[0], e.position[1], e.position[2], e.radius);
    de[i].instance = e.instance;
    de[i].alias_offset = e.alias_table[0];
    de[i].alias_count = e.alias_table[1];
    de[i].surface_area = e.surface_area;
  }
  o.emissives = blob.add(de);
  std::vector<float2> al(c->alias_table.size());
  for (size_t i = 0; i < al.size(); ++i) al[i] = make_float2(c->alias_table[i].prob, as_f(c->alias_table[i].index));
  o.alias = blob.add(al);

  const uint32_t n_tex = (uint32_t)c->textures.size();
  std::vector<float4> mats(4 * c->materials.size());
  for (size_t i = 0; i < c->materials.size(); ++i) {
    const HkMaterial& m = c->materials[i];
    const uint32_t ids[4] = {m.base_color_texture, m.emissive_texture, m.metallic_roughness_texture, m.occlusion_texture};
    for (uint32_t id : ids)  // MaterialTextures::id, material.rs:76-86: an index into the texture array or u32::MAX
      HK_REQUIRE(id == HK_NO_TEXTURE || id < n_tex, HK_E_INVALID, "material %zu references texture %u but only %u textures are uploaded", i, id, n_tex);
    mats[4 * i] = make_float4(m.base_color[0], m.base_color[1], m.base_color[2], m.base_color[3]);
    mats[4 * i + 1] = make_float4(m.emissive[0], m.emissive[1], m.emissive[2], m.emissive[3]);
    mats[4 * i + 2] = make_float4(m.perceptual_roughness, m.metallic, m.reflectance, 0.0f);
    mats[4 * i + 3] = make_float4(as_f(ids[0]), as_f(ids[1]), as_f(ids[2]), as_f(ids[3]));
  }
  o.materials = blob.add(mats);
  // material textures: a 16-B descriptor per texture + the sRGB decode table (texels live in their own buffer)
  std::vector<uint4> tex_info(n_tex);
  size_t texel_offset = 0;
  for (uint32_t i = 0; i < n_tex; ++i) {
    const hk_ctx::HostTexture& t = c->textures[i];
    tex_info[i] = make_uint4((uint32_t)texel_offset, t.w, t.h, t.flags);
    texel_offset += t.texels.size();
  }
  std::vector<float> srgb_lut(256);
  for (int i = 0; i < 256; ++i) {  // sRGB EOTF in double, rounded once
    double v = i / 255.0;
    srgb_lut[i] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4));
  }
  o.tex_info = blob.add(tex_info);
  o.srgb_lut = blob.add(srgb_lut);
  blob.bytes.resize((blob.bytes.size() + 31) & ~(size_t)31, 0);
  // the one-level BVH (traverse_flat): only for scenes that stay inside the LDS copy WITH it, whose instances share one
  // transform, outside the bit-exact verification mode; as many direction orderings (8, 4, 2, 1) as fit
  o.flat = 0;
  o.flat_count = o.flat_orderings = 0;
  if (!(c->flags & HK_CTX_EXACT_TRAVERSAL) && !c->threaded && !c->instances.empty() && c->instances.size() <= 0xFFFFu && !getenv("HK_FLAT_DISABLE")) {
    bool shared = true;
    for (const HkInstance& in : c->instances)
      if (memcmp(in.inverse_transpose_model, c->instances[0].inverse_transpose_model, 64) != 0) shared = false;
    // direction orderings: as many (8, 4, 2, 1) as keep the node array within 4 KB - every workgroup copies the blob into LDS and
    // the LDS a workgroup holds bounds the workgroups per CU; measured on the Cornell box (71 nodes, tools/ab_flat.sh): 1 / 2 / 4 / 8
    // orderings walk equally fast (0.27 ms k_indirect) and the direct-light kernels lose 13 % with the 18 KB of eight
    uint32_t want = 8, budget = 4096;
    if (const char* e = getenv("HK_FLAT_ORDERINGS")) { want = (uint32_t)std::max(1, std::min(8, atoi(e))); budget = HK_LDS_SCENE_BYTES; }
    while (want & (want - 1)) want &= want - 1;  // a power of two
    std::vector<float4> flat;
    uint32_t count = 0;
    if (shared && build_flat_bvh(c, 1u, flat, count)) {  // (a first build tells the node count: 2 T - 1)
      const size_t per_ordering = flat.size() * 16;
      uint32_t ord = want;
      while (ord > 1 && (per_ordering * ord > budget || blob.bytes.size() + per_ordering * ord + static_bytes > HK_LDS_SCENE_BYTES)) ord >>= 1;
      if (ord > 1 && !build_flat_bvh(c, ord, flat, count)) ord = 0;
      if (ord >= 1 && blob.bytes.size() + flat.size() * 16 + static_bytes <= HK_LDS_SCENE_BYTES && count <= 0xFFFFu) {
        o.flat = blob.add(flat);
        o.flat_count = count;
        o.flat_orderings = ord;
        blob.bytes.resize((blob.bytes.size() + 31) & ~(size_t)31, 0);
      }
    }
  }
  return HK_OK;
}

int join_side(hk_ctx* c);
int join_post(hk_ctx* c);
int join_all(hk_ctx* c);
// wait for everything the context has enqueued, on ALL streams (the direct-light dispatches of a frame may still be
// running on the side stream when a host uploads, resizes or reads statistics between two stages)
int sync_all(hk_ctx* c) {
  const int rc = join_all(c);
  if (rc) return rc;
  HK_HIP(hipStreamSynchronize(c->stream));
  return HK_OK;
}

// DScene::shared_xform: every instance has the same inverse model (bit for bit), so a traversal transforms its ray once instead of
// once per instance entry (hk_device.hpp traverse_top).  Derived from the host mirrors: whoever changes an instance's pose - an
// upload or a device refit - has to call this before the next frame is enqueued.
void update_shared_transform(hk_ctx* c) {
  c->scene.shared_xform = 1u;
  for (const HkInstance& in : c->instances)
    if (memcmp(in.inverse_transpose_model, c->instances[0].inverse_transpose_model, 64) != 0) c->scene.shared_xform = 0u;
  // the one-level BVH lives in the shared LOCAL space: it stays valid while the instances move together and is simply not
  // walked once one of them moves on its own
  c->scene.flat_mode = (c->dyn_off.flat_count && c->scene.shared_xform) ? 1u : 0u;
}

// point c->scene at the arrays of the slot in use
void point_scene_at_slot(hk_ctx* c) {
  const size_t slots = (c->two_slots ? 2 : 1) * c->dyn_capacity;
  const DynOffsets& o = c->dyn_off;
  const uint8_t* base = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
  const uint8_t* sbase = c->scene_mem + slots;
  DScene& s = c->scene;
  s.blob = (const float4*)base;  // (two slots: the scene is too big for the LDS copy, blob is not read)
  s.blob_f4 = (uint32_t)((slots + c->static_bytes) / 16);
  s.nodes = (const float4*)base;
  s.blas_base = (uint32_t)(((size_t)(sbase - base) + c->st_nodes) / 32);
  s.instances = (const DInstance*)(base + o.instances);
  c->d_prev_models = (const float4*)(base + o.prev_models);
  s.tri_v0 = (const float4*)(sbase + c->st_v0); s.tri_v1 = (const float4*)(sbase + c->st_v1); s.tri_v2 = (const float4*)(sbase + c->st_v2);
  s.vtx_normal = (const float4*)(sbase + c->st_vn); s.vtx_uv = (const float2*)(sbase + c->st_vuv);
  s.materials = (const float4*)(base + o.materials);
  s.tex_info = (const uint4*)(base + o.tex_info);
  s.srgb_lut = (const float*)(base + o.srgb_lut);
  s.tex_data = c->d_tex_data.p;
  s.n_textures = (uint32_t)c->textures.size();
  s.light_lo = (const float4*)(base + o.light_lo); s.light_hi = (const float4*)(base + o.light_hi);
  s.emissives = (const DEmissive*)(base + o.emissives); s.alias = (const float2*)(base + o.alias);
  s.noise = c->d_noise.p;
  s.tlas_count = (uint32_t)c->instance_nodes.size();
  s.tlas_stride = c->threaded ? (uint32_t)c->instance_nodes.size() : 0u;
  s.blas_stride = c->threaded ? (uint32_t)c->asset_nodes.size() : 0u;
  s.light_count = (uint32_t)c->emissive_nodes.size();
  s.flat = (const float4*)(base + o.flat);
  s.flat_count = o.flat_count;
  s.flat_mask = o.flat_orderings ? o.flat_orderings - 1u : 0u;
  update_shared_transform(c);
}

int finalize_scene(hk_ctx* c) {
  if (!c->mesh_dirty && !c->dynamic_dirty && !c->textures_dirty) return HK_OK;
  HK_REQUIRE(c->have_meshes && c->have_materials && c->have_instances, HK_E_NOT_READY, "meshes, materials and instances must be uploaded first");
  const size_t n_nodes = c->asset_nodes.size();
  bool need_static = c->mesh_dirty || !c->scene_mem || c->node_prim_offset.size() != n_nodes;
  {  // direction-threaded flattenings for everything that will not be traversed from the LDS copy (an estimate of the blob size decides;
     // a scene near the limit that ends up outside LDS without them merely walks in the reference's order)
    const size_t est = n_nodes * 32 + c->primitives.size() * 48 + c->vertices.size() * 24 + c->instance_nodes.size() * 32 + c->instances.size() * 208 +
                       c->materials.size() * 64 + c->emissive_nodes.size() * 32 + c->alias_table.size() * 8;
    const bool want = !(c->flags & HK_CTX_EXACT_TRAVERSAL) && est > HK_LDS_SCENE_BYTES;
    if (want != c->threaded) {
      c->threaded = want;
      need_static = true;
    }
  }
  for (const HkInstance& in : c->instances) {
    HK_REQUIRE((size_t)in.mesh.node_offset + in.mesh.node_count <= n_nodes, HK_E_INVALID, "instance mesh node range out of bounds");
    HK_REQUIRE(in.material < c->materials.size(), HK_E_INVALID, "instance material out of bounds");
    // a mesh range no earlier instance used: its leaf boxes have not been derived yet
    if (!need_static && in.mesh.node_count && (c->node_prim_offset[in.mesh.node_offset] != (int64_t)in.mesh.primitive ||
                                                c->node_prim_offset[in.mesh.node_offset + in.mesh.node_count - 1] != (int64_t)in.mesh.primitive))
      need_static = true;
  }
  int rc;
  if (c->textures_dirty) {
    std::vector<uint32_t> tex_data;
    for (const hk_ctx::HostTexture& t : c->textures) tex_data.insert(tex_data.end(), t.texels.begin(), t.texels.end());
    if ((rc = sync_all(c))) return rc;
    if ((rc = c->d_tex_data.upload(tex_data))) return rc;
    c->textures_dirty = false;
  }
  HK_REQUIRE(!(c->mirrors_stale && c->dynamic_dirty), HK_E_NOT_READY,
             "the instance-level arrays were last changed on the device (hk_refit_scene_instances): upload the instances again (hk_upload_scene_instances) "
             "before a change that rebuilds them on the host");
  Blob st;  // (the mesh-level region first: whether the one-level BVH still fits the LDS copy depends on its size)
  if (need_static && (rc = build_static_region(c, st, c->st_nodes, c->st_v0, c->st_v1, c->st_v2, c->st_vn, c->st_vuv))) return rc;
  Blob& dyn = c->dyn_blob;
  dyn.bytes.clear();
  DynOffsets o{};
  const double tb0_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if ((rc = build_dynamic_region(c, dyn, o, need_static ? st.bytes.size() : c->static_bytes))) return rc;
  if (getenv("HK_TRACE_UPDATE")) fprintf(stderr, "  build_dynamic_region %.2f ms (%zu bytes)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - tb0_, dyn.bytes.size());
  c->dyn_off = o;
  c->rf_ready = false;
  c->rf_last_moved.clear();
  const bool in_place = !need_static && dyn.bytes.size() <= c->dyn_capacity;
  if (!(in_place && c->two_slots) && (rc = sync_all(c))) return rc;  // frames in flight still read the arrays rewritten below
  if (need_static) {
    if (c->scene_mem) { (void)hipFree(c->scene_mem); c->scene_mem = nullptr; }
    c->dyn_capacity = dyn.bytes.size();  // exact: a small scene stays small enough for the LDS copy
    c->static_bytes = st.bytes.size();
    c->two_slots = c->dyn_capacity + c->static_bytes > HK_LDS_SCENE_BYTES;
    c->slot = 0;
    // room for the previous model matrix of every instance, so that the first moving frame already fits its slot
    if (c->two_slots) c->dyn_capacity = (c->dyn_capacity + 64 * c->instances.size() + 31) & ~(size_t)31;
    const size_t slots = (c->two_slots ? 2 : 1) * c->dyn_capacity;
    HK_HIP(hipMalloc((void**)&c->scene_mem, slots + c->static_bytes));
    HK_HIP(hipMemcpy(c->scene_mem + slots, st.bytes.data(), st.bytes.size(), hipMemcpyHostToDevice));
  } else if (!in_place) {  // instance count grew: move the mesh region behind larger slots, device to device
    const size_t cap = ((dyn.bytes.size() + dyn.bytes.size() / 2) + 31) & ~(size_t)31;
    const size_t old_slots = (c->two_slots ? 2 : 1) * c->dyn_capacity;
    c->two_slots = c->two_slots || cap + c->static_bytes > HK_LDS_SCENE_BYTES;
    c->slot = 0;
    const size_t slots = (c->two_slots ? 2 : 1) * cap;
    uint8_t* mem = nullptr;
    HK_HIP(hipMalloc((void**)&mem, slots + c->static_bytes));
    HK_HIP(hipMemcpy(mem + slots, c->scene_mem + old_slots, c->static_bytes, hipMemcpyDeviceToDevice));
    (void)hipFree(c->scene_mem);
    c->scene_mem = mem;
    c->dyn_capacity = cap;
  } else if (c->two_slots) {
    c->slot ^= 1;
  }
  uint8_t* const slot_mem = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
  if (in_place && c->two_slots) {
    const int k = c->slot;
    if (c->staging_pending[k]) {  // the copy that last read this staging buffer (two updates ago)
      HK_HIP(hipEventSynchronize(c->staging_done[k]));
      c->staging_pending[k] = false;
    }
    if (c->staging_bytes[k] < c->dyn_capacity) {
      if (c->staging[k]) (void)hipHostFree(c->staging[k]);
      c->staging[k] = nullptr;
      c->staging_bytes[k] = 0;
      HK_HIP(hipHostMalloc((void**)&c->staging[k], c->dyn_capacity, hipHostMallocDefault));
      c->staging_bytes[k] = c->dyn_capacity;
    }
    if (!c->staging_done[k]) HK_HIP(hipEventCreateWithFlags(&c->staging_done[k], hipEventDisableTiming));
    memcpy(c->staging[k], dyn.bytes.data(), dyn.bytes.size());
    memset(c->staging[k] + dyn.bytes.size(), 0, c->dyn_capacity - dyn.bytes.size());
    // a copy KERNEL reading the pinned buffer over PCIe: the update stays on the compute queue of the stream, between
    // the kernels of two frames, instead of a hand-off to an SDMA engine and back
    hipLaunchKernelGGL(k_copy_u4, dim3((unsigned)((c->dyn_capacity / 16 + 255) / 256)), dim3(256), 0, c->stream, (uint4*)slot_mem,
                       (const uint4*)c->staging[k], c->dyn_capacity / 16);
    HK_HIP(hipGetLastError());
    HK_HIP(hipEventRecord(c->staging_done[k], c->stream));
    c->staging_pending[k] = true;
    c->async_instance_uploads += 1;
  } else {
    HK_HIP(hipMemcpy(slot_mem, dyn.bytes.data(), dyn.bytes.size(), hipMemcpyHostToDevice));
    if (dyn.bytes.size() < c->dyn_capacity) HK_HIP(hipMemset(slot_mem + dyn.bytes.size(), 0, c->dyn_capacity - dyn.bytes.size()));
  }

  point_scene_at_slot(c);
  c->mesh_dirty = c->dynamic_dirty = false;
  c->static_rebuilds += need_static ? 1 : 0;
  c->dynamic_rebuilds += 1;
  return HK_OK;
}

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
// the parked planes, on first use (3 x 72 B per render pixel)
int ensure_parked(hk_ctx* c) {
  if (c->det_winner[0]) return HK_OK;
  const size_t nr = (size_t)c->RW * c->RH;
  for (int k = 0; k < 3; ++k) {
    HK_HIP(hipMalloc((void**)&c->det_winner[k], nr * sizeof(int)));
    for (uint32_t b : {(uint32_t)HK_BUF_PARKED_TO0 + k, (uint32_t)HK_BUF_PARKED_RECORD0 + k}) {
      const size_t bytes = nr * buffer_bpp(b);
      HK_HIP(hipMalloc(&c->buf[b], bytes));
      HK_HIP(hipMemsetAsync(c->buf[b], b < HK_BUF_PARKED_RECORD0 ? 0xFF : 0, bytes, c->stream));  // nothing parked
      c->buf_bytes[b] = bytes;
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
  const bool parked = c->det_winner[channel] && ((c->flags & HK_CTX_DETERMINISTIC_SCATTER) || parks_across_bands(c));
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
  if (getenv("HK_WF_TIMELINE") && !w.timeline) HK_HIP(hipMalloc((void**)&w.timeline, 64 * 32 * sizeof(unsigned long long)));  // tools/wf_timeline.py
  return HK_OK;
}

// make the main stream wait for what was enqueued on the side stream
int join_side(hk_ctx* c) {
  if (!c->forked) return HK_OK;
  HK_HIP(hipEventRecord(c->join_event, c->side_stream));
  HK_HIP(hipStreamWaitEvent(c->stream, c->join_event, 0));
  c->forked = false;
  return HK_OK;
}
// make the main stream wait for the a-trous levels of the last frame (post_stream)
int join_post(hk_ctx* c) {
  if (!c->post_pending) return HK_OK;
  HK_HIP(hipStreamWaitEvent(c->stream, c->post_done, 0));
  c->post_pending = false;
  return HK_OK;
}
int join_all(hk_ctx* c) {
  const int rc = join_side(c);
  return rc ? rc : join_post(c);
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
  ScopedTimer timer(c, pass, pass == HK_PASS_INDIRECT);
  switch (pass) {
    case HK_PASS_PREPASS: {
      Jitter j = prepass_jitter(c);
      launch_prepass(c->stream, c->scene, fr, c->view.inverse_view_proj, c->view.view_proj, c->pview.view_proj, c->d_prev_models, j.x, j.y, g, y0, y1, counters);
      break;
    }
    case HK_PASS_FULL_SCREEN_ALBEDO: launch_albedo(c->stream, c->scene, fr, g, c->buf[HK_BUF_ALBEDO], y0, y1); break;
    case HK_PASS_DIRECT_LIT:
    case HK_PASS_DIRECT_EMISSIVE:
    case HK_PASS_INDIRECT: {
      const int channel = pass == HK_PASS_DIRECT_LIT ? 0 : (pass == HK_PASS_DIRECT_EMISSIVE ? 1 : 2);
      const bool across = parks_across_bands(c);
      if (across || (c->flags & HK_CTX_DETERMINISTIC_SCATTER)) { const int rc_ = ensure_parked(c); if (rc_) return rc_; }
      LightTargets t = make_light_targets(c, channel);
      { const int rc_ = attach_tile_meta(c, t, channel, false, y0, y1); if (rc_) return rc_; }
      const size_t px = (size_t)c->RW * c->RH;
      if (t.det_winner) {  // nothing parked, no winner: -1 everywhere (a band: in the rows it dispatches - the others arrive with exchange A)
        HK_HIP(hipMemsetAsync(t.det_winner, 0xFF, px * sizeof(int), c->stream));
        if (across) HK_HIP(hipMemsetAsync(t.det_to + (size_t)y0 * c->RW, 0xFF, (size_t)(y1 - y0) * c->RW * sizeof(int), c->stream));
        else HK_HIP(hipMemsetAsync(t.det_to, 0xFF, px * sizeof(int), c->stream));
      }
      if (pass == HK_PASS_INDIRECT && use_wavefront(c)) {
        { const int rc_ = ensure_wavefront(c); if (rc_) return rc_; }
        launch_indirect_wavefront(c->stream, c->scene, fr, g, t, c->wf, y0, y1, c->compute_units, timer.on ? timer.t.start : nullptr,
                                  timer.on ? timer.t.stop : nullptr);
      } else if (pass == HK_PASS_INDIRECT)  // MULTIPLE_BOUNCES pipeline iff bounces >= 2, light.rs:663-666
        launch_indirect(c->stream, c->frame.indirect_bounces >= 2u, c->scene, fr, g, t, y0, y1, counters, timer.on ? timer.t.start : nullptr,
                        timer.on ? timer.t.stop : nullptr);
      else
        launch_direct(c->stream, pass == HK_PASS_DIRECT_EMISSIVE, c->scene, fr, g, t, y0, y1, counters);
      // (a band with a history halo resolves at the start of stage SPATIAL, once the neighbours' parked rows are in)
      if (t.det_winner && !across) launch_resolve_scatter(c->stream, t, 0, (int)px, 0, 0);
      break;
    }
    case HK_PASS_EMISSIVE_SPATIAL_REUSE:
    case HK_PASS_INDIRECT_SPATIAL_REUSE: {
      const int channel = pass == HK_PASS_EMISSIVE_SPATIAL_REUSE ? 1 : 2;
      LightTargets t = make_light_targets(c, channel);
      { const int rc_ = attach_tile_meta(c, t, channel, true, y0, y1); if (rc_) return rc_; }
      launch_spatial(c->stream, channel == 1, c->scene, fr, g, t, y0, y1);
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

}  // namespace

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
  c->timing_mask = (flags & HK_CTX_TIME_PASSES) ? 0xFFFFFFFFu : 0u;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&c->d_counters, 8 * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(c->d_counters, 0, 8 * sizeof(unsigned long long)) != hipSuccess || hipEventCreate(&c->frame_start) != hipSuccess ||
      hipEventCreate(&c->frame_stop) != hipSuccess) {
    set_error("HIP resource creation failed: %s", hipGetErrorString(hipGetLastError()));
    hk_destroy(c);
    return HK_E_HIP;
  }
  c->stream = c->own_stream;
  if (!(flags & HK_CTX_SINGLE_STREAM)) {
    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->fork_event, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join_event, hipEventDisableTiming) != hipSuccess) {
      set_error("cannot create the side stream");
      hk_destroy(c);
      return HK_E_HIP;
    }
    if (!getenv("HK_NO_FRAME_PIPELINE") && !(flags & (HK_CTX_DETERMINISTIC_SCATTER | HK_CTX_COUNT_RAYS | HK_CTX_TIME_PASSES))) {
      if (hipStreamCreateWithFlags(&c->post_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->post_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&c->post_done, hipEventDisableTiming) != hipSuccess) {
        set_error("cannot create the post-process stream");
        hk_destroy(c);
        return HK_E_HIP;
      }
    }
  }
  *out = c;
  return HK_OK;
}

namespace { void free_refit(hk_ctx* c); }
void hk_destroy(hk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)join_all(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  comm_release(c);
  drain_timers(c);
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->frame_start) (void)hipEventDestroy(c->frame_start);
  if (c->frame_stop) (void)hipEventDestroy(c->frame_stop);
  free_screen(c);
  if (c->scene_mem) (void)hipFree(c->scene_mem);
  for (int k = 0; k < 2; ++k) {
    if (c->staging[k]) (void)hipHostFree(c->staging[k]);
    if (c->staging_done[k]) (void)hipEventDestroy(c->staging_done[k]);
  }
  free_refit(c);
  c->d_tex_data.release();
  c->d_noise.release();
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
  if (c->post_stream) (void)hipStreamDestroy(c->post_stream);
  if (c->post_fork) (void)hipEventDestroy(c->post_fork);
  if (c->post_done) (void)hipEventDestroy(c->post_done);
  if (c->fork_event) (void)hipEventDestroy(c->fork_event);
  if (c->join_event) (void)hipEventDestroy(c->join_event);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int hk_upload_meshes(hk_ctx* c, const HkVertex* v, uint32_t nv, const HkPrimitive* p, uint32_t np, const HkNode* n, uint32_t nn) {
  HK_REQUIRE(c && v && p && n && nv && np && nn, HK_E_INVALID, "NULL or empty mesh buffers");
  c->vertices.assign(v, v + nv);
  c->primitives.assign(p, p + np);
  c->asset_nodes.assign(n, n + nn);
  c->have_meshes = true;
  c->mesh_dirty = true;
  return HK_OK;
}
int hk_upload_materials(hk_ctx* c, const HkMaterial* m, uint32_t n) {
  HK_REQUIRE(c && m && n, HK_E_INVALID, "NULL or empty material buffer");
  c->materials.assign(m, m + n);
  c->have_materials = true;
  c->dynamic_dirty = true;
  return HK_OK;
}
int hk_upload_instances(hk_ctx* c, const HkInstance* inst, uint32_t ni, const HkNode* inodes, uint32_t nin, const HkEmissive* em, uint32_t ne,
                        const HkNode* enodes, uint32_t nen, const HkAliasEntry* alias, uint32_t na) {
  HK_REQUIRE(c && inst && inodes && ni && nin, HK_E_INVALID, "NULL or empty instance buffers");
  HK_REQUIRE((em || !ne) && (enodes || !nen) && (alias || !na), HK_E_INVALID, "NULL emissive buffers");
  c->instances.assign(inst, inst + ni);
  c->instance_nodes.assign(inodes, inodes + nin);
  c->emissives.assign(em, em + ne);
  c->emissive_nodes.assign(enodes, enodes + nen);
  c->alias_table.assign(alias, alias + na);
  c->prev_models.clear();
  c->have_instances = true;
  c->dynamic_dirty = true;
  c->mirrors_stale = false;
  return HK_OK;
}
int hk_upload_previous_transforms(hk_ctx* c, const float* models, uint32_t n) {
  HK_REQUIRE(c && (models || !n), HK_E_INVALID, "NULL argument");
  HK_REQUIRE(c->have_instances && n == c->instances.size(), HK_E_INVALID, "previous transforms must match the %zu uploaded instances", c->instances.size());
  c->prev_models.assign(models, models + 16 * (size_t)n);
  c->dynamic_dirty = true;
  return HK_OK;
}
#define HK_NO_STANDINS(b)                                                                                                        \
  HK_REQUIRE(!builder_has_standin_trees(b), HK_E_NOT_READY,                                                                      \
             "the builder holds stand-in trees (hk_scene_builder_finish_instances): finish it with hk_scene_builder_finish, or use " \
             "hk_update_scene_instances, which builds the trees on the device")
int hk_upload_scene(hk_ctx* c, const hk_scene_builder* b) {
  HK_REQUIRE(c && b, HK_E_INVALID, "NULL argument");
  HK_NO_STANDINS(b);  // (frames from stand-in trees would differ silently in tie-breaks and visit order)
  const HkVertex* v; const HkPrimitive* p; const HkNode *an, *in_, *en; const HkMaterial* m; const HkInstance* inst; const HkEmissive* em; const HkAliasEntry* al;
  uint32_t nv, np, nan_, nm, ni, nin, ne, nen, nal;
  int rc;
  if ((rc = hk_scene_builder_vertices(b, &v, &nv))) return rc;
  if ((rc = hk_scene_builder_primitives(b, &p, &np))) return rc;
  if ((rc = hk_scene_builder_asset_nodes(b, &an, &nan_))) return rc;
  if ((rc = hk_scene_builder_materials(b, &m, &nm))) return rc;
  if ((rc = hk_scene_builder_instances(b, &inst, &ni))) return rc;
  if ((rc = hk_scene_builder_instance_nodes(b, &in_, &nin))) return rc;
  if ((rc = hk_scene_builder_emissives(b, &em, &ne))) return rc;
  if ((rc = hk_scene_builder_emissive_nodes(b, &en, &nen))) return rc;
  if ((rc = hk_scene_builder_alias_table(b, &al, &nal))) return rc;
  if ((rc = hk_upload_meshes(c, v, nv, p, np, an, nan_))) return rc;
  if ((rc = hk_upload_materials(c, m, nm))) return rc;
  if ((rc = hk_upload_instances(c, inst, ni, in_, nin, em, ne, en, nen, al, nal))) return rc;
  const float* pm; uint32_t npm;
  if ((rc = hk_scene_builder_previous_transforms(b, &pm, &npm))) return rc;
  return hk_upload_previous_transforms(c, pm, npm);
}
int hk_upload_scene_instances(hk_ctx* c, const hk_scene_builder* b) {
  HK_REQUIRE(c && b, HK_E_INVALID, "NULL argument");
  HK_NO_STANDINS(b);
  return hk::upload_scene_instances_unchecked(c, b);
}
}  // extern "C"
int hk::upload_scene_instances_unchecked(hk_ctx* c, const hk_scene_builder* b) {
  HK_REQUIRE(c && b, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(c->have_meshes && c->have_materials, HK_E_NOT_READY, "hk_upload_scene must come first");
  const HkNode *in_, *en; const HkInstance* inst; const HkEmissive* em; const HkAliasEntry* al; const float* pm;
  uint32_t ni, nin, ne, nen, nal, npm;
  int rc;
  if ((rc = hk_scene_builder_instances(b, &inst, &ni))) return rc;
  if ((rc = hk_scene_builder_instance_nodes(b, &in_, &nin))) return rc;
  if ((rc = hk_scene_builder_emissives(b, &em, &ne))) return rc;
  if ((rc = hk_scene_builder_emissive_nodes(b, &en, &nen))) return rc;
  if ((rc = hk_scene_builder_alias_table(b, &al, &nal))) return rc;
  if ((rc = hk_scene_builder_previous_transforms(b, &pm, &npm))) return rc;
  if ((rc = hk_upload_instances(c, inst, ni, in_, nin, em, ne, en, nen, al, nal))) return rc;
  return hk_upload_previous_transforms(c, pm, npm);
}
extern "C" {

// Instances added, removed or re-materialed (the reference re-runs prepare_instances for ANY instance change, instance.rs:352-437):
// the per-instance / per-emitter records are laid out on the host - O(instances), no tree build - and go to the spare slot through
// the asynchronous upload; both trees are then built on the device (hk_rebuild_scene_trees: HK_TREE_SAH = the reference's own
// tree, link for link).  What the host no longer does is the two `BVH::build` calls: 1.1 ms of 1.7 ms at 2 000 instances, 30 of
// 32 ms at 20 000.
int hk_update_scene_instances(hk_ctx* c, hk_scene_builder* b, uint32_t tree_mode) {
  HK_REQUIRE(c && b, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(tree_mode == HK_TREE_SAH || tree_mode == HK_TREE_LBVH, HK_E_INVALID, "unknown tree build mode %u", tree_mode);
  int rc;
  const bool trace = getenv("HK_TRACE_UPDATE") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  if ((rc = hk_scene_builder_finish_instances(b))) return rc;
  const double t1 = now();
  if ((rc = upload_scene_instances_unchecked(c, b))) return rc;
  const double t2 = now();
  uint32_t ni = 0;
  const HkInstance* inst = nullptr;
  if ((rc = hk_scene_builder_instances(b, &inst, &ni))) return rc;
  c->trees_pending_on_device = ni >= 2;
  rc = finalize_scene(c);
  c->trees_pending_on_device = false;
  const double t3 = now();
  if (!rc && ni >= 2) rc = hk_rebuild_scene_trees(c, tree_mode);  // (a tree of one leaf is what the host just laid out)
  // the region just uploaded carries stand-in trees with orderings 1..7 left zero (the device build was to overwrite
  // them in stream order).  If that build did not get enqueued, a threaded walk over zero nodes would never leave node 0: the
  // next use of the scene lays the region out again, from the host's (valid) stand-in trees, all orderings threaded.
  if (rc) c->dynamic_dirty = true;
  if (trace) fprintf(stderr, "hk_update_scene_instances: finish_instances %.2f ms, mirrors %.2f ms, layout + upload %.2f ms, device build enqueue %.2f ms\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
  return rc;
}

// ---- instance motion on the device (kernels_scene.hip) --------------------------------------------------
namespace {
void free_refit(hk_ctx* c) {
  for (void* q : {(void*)c->rf_inst_lo, (void*)c->rf_inst_hi, (void*)c->rf_prev_models, (void*)c->rf_emissive_of_instance, (void*)c->rf_alias_scratch})
    if (q) (void)hipFree(q);
  c->rf_inst_lo = c->rf_inst_hi = c->rf_prev_models = nullptr;
  c->rf_emissive_of_instance = nullptr;
  c->rf_alias_scratch = nullptr;
  c->rf_instances = c->rf_alias = 0;
  c->rf_ready = false;
  if (c->lbvh_scratch) (void)hipFree(c->lbvh_scratch);
  c->lbvh_scratch = nullptr;
  c->lbvh_scratch_cap = 0;
  for (int k = 0; k < 2; ++k) {
    if (c->rf_updates[k]) (void)hipHostFree(c->rf_updates[k]);
    if (c->rf_done[k]) (void)hipEventDestroy(c->rf_done[k]);
    c->rf_updates[k] = nullptr;
    c->rf_done[k] = nullptr;
    c->rf_updates_cap[k] = 0;
    c->rf_pending[k] = false;
  }
}
hkd::RefitScene refit_scene(hk_ctx* c) {
  uint8_t* base = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
  hkd::RefitScene r;
  r.instances = (DInstance*)(base + c->dyn_off.instances);
  r.prev_models = c->rf_prev_models;
  r.inst_lo = c->rf_inst_lo;
  r.inst_hi = c->rf_inst_hi;
  r.emissive_of_instance = c->rf_emissive_of_instance;
  r.emissives = (DEmissive*)(base + c->dyn_off.emissives);
  r.alias = (float2*)(base + c->dyn_off.alias);
  r.alias_scratch = c->rf_alias_scratch;
  r.materials = (const float4*)(base + c->dyn_off.materials);
  r.tri_v0 = c->scene.tri_v0;
  r.tri_v1 = c->scene.tri_v1;
  r.tri_v2 = c->scene.tri_v2;
  return r;
}
// side arrays of the refit, (re)filled from the scene as the host last laid it out: world AABB per instance from the TLAS
// leaves, emitter of an instance, previous models
int prepare_refit(hk_ctx* c) {
  if (c->rf_ready) return HK_OK;
  const size_t ni = c->instances.size(), na = c->alias_table.size();
  if (ni > c->rf_instances || na > c->rf_alias) {
    int rc = sync_all(c);
    if (rc) return rc;
    for (void* q : {(void*)c->rf_inst_lo, (void*)c->rf_inst_hi, (void*)c->rf_prev_models, (void*)c->rf_emissive_of_instance, (void*)c->rf_alias_scratch})
      if (q) (void)hipFree(q);
    c->rf_inst_lo = c->rf_inst_hi = c->rf_prev_models = nullptr;
    c->rf_emissive_of_instance = nullptr;
    c->rf_alias_scratch = nullptr;
    c->rf_instances = c->rf_alias = 0;  // (stays 0 if an allocation below fails: the next call starts over)
    const size_t cap_i = ni + ni / 4, cap_a = na + na / 4 + 4;
    HK_HIP(hipMalloc((void**)&c->rf_inst_lo, cap_i * 16));
    HK_HIP(hipMalloc((void**)&c->rf_inst_hi, cap_i * 16));
    HK_HIP(hipMalloc((void**)&c->rf_prev_models, cap_i * 64));
    HK_HIP(hipMalloc((void**)&c->rf_emissive_of_instance, cap_i * 4));
    HK_HIP(hipMalloc((void**)&c->rf_alias_scratch, cap_a * 5 * 4));
    c->rf_instances = cap_i;
    c->rf_alias = cap_a;
  }
  std::vector<uint32_t> eoi(ni, 0xFFFFFFFFu);
  for (size_t e = 0; e < c->emissives.size(); ++e) eoi[c->emissives[e].instance] = (uint32_t)e;
  HK_HIP(hipMemcpyAsync(c->rf_emissive_of_instance, eoi.data(), ni * 4, hipMemcpyHostToDevice, c->stream));
  HK_HIP(hipStreamSynchronize(c->stream));  // (eoi is a local; once per host-side rebuild)
  const hkd::RefitScene r = refit_scene(c);
  const uint8_t* base = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
  launch_gather_instance_boxes(c->stream, r, (const float4*)(base + c->dyn_off.tlas), (uint32_t)c->instance_nodes.size());
  // previous models of the instances the host marked as moved (the plane exists only then)
  if (c->prev_models.size() == 16 * ni && c->dyn_off.prev_models + ni * 64 <= c->dyn_capacity && c->d_prev_models && c->d_prev_models != c->rf_prev_models) {
    bool any = false;
    for (size_t i = 0; i < ni && !any; ++i) any = memcmp(&c->prev_models[16 * i], c->instances[i].model, 64) != 0;
    if (any) launch_copy_region(c->stream, c->rf_prev_models, base + c->dyn_off.prev_models, ni * 64);
  }
  HK_HIP(hipGetLastError());
  c->rf_last_moved.clear();
  for (size_t i = 0; i < ni; ++i)
    if (c->prev_models.size() == 16 * ni && memcmp(&c->prev_models[16 * i], c->instances[i].model, 64) != 0) c->rf_last_moved.push_back((uint32_t)i);
  c->rf_ready = true;
  return HK_OK;
}
}  // namespace

namespace {
// frames in flight keep reading the slot they were enqueued with: a device-side update works on a copy in the spare slot
// (two-slot scenes), or in place behind everything enqueued so far (scenes small enough for the LDS copy have one slot)
int begin_device_update(hk_ctx* c) {
  const int rc = join_side(c);
  if (rc) return rc;
  if (c->two_slots) {
    uint8_t* from = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
    c->slot ^= 1;
    uint8_t* to = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
    launch_copy_region(c->stream, to, from, c->dyn_capacity);
    const float4* prev = c->d_prev_models;
    point_scene_at_slot(c);
    if (prev == c->rf_prev_models) c->d_prev_models = prev;  // (the refit's own plane is not part of the slot)
  }
  return HK_OK;
}
}  // namespace

int hk_rebuild_scene_trees(hk_ctx* c, uint32_t mode) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  HK_REQUIRE(mode == HK_TREE_SAH || mode == HK_TREE_LBVH, HK_E_INVALID, "unknown tree build mode %u", mode);
  HK_REQUIRE(c->have_meshes && c->have_materials && c->have_instances, HK_E_NOT_READY, "hk_upload_scene must come first");
  HK_HIP(hipSetDevice(c->device));
  int rc;
  if ((rc = finalize_scene(c))) return rc;
  const uint32_t ni = (uint32_t)c->instances.size(), ne = (uint32_t)c->emissives.size();
  HK_REQUIRE(c->instance_nodes.size() == 3 * (size_t)ni - 2 && (ne == 0 || c->emissive_nodes.size() == 3 * (size_t)ne - 2), HK_E_UNSUPPORTED,
             "the uploaded trees are not in the flatten_custom layout of a binary tree (3n - 2 nodes): nothing to rebuild in place");
  if ((rc = prepare_refit(c))) return rc;
  const size_t need = std::max(lbvh_scratch_bytes(ni, nullptr), lbvh_scratch_bytes(std::max(ne, 1u), nullptr));
  if (need > c->lbvh_scratch_cap) {
    if ((rc = sync_all(c))) return rc;
    if (c->lbvh_scratch) (void)hipFree(c->lbvh_scratch);
    c->lbvh_scratch = nullptr;
    c->lbvh_scratch_cap = 0;
    HK_HIP(hipMalloc(&c->lbvh_scratch, need + need / 4));
    c->lbvh_scratch_cap = need + need / 4;
  }
  if ((rc = begin_device_update(c))) return rc;
  const hkd::RefitScene r = refit_scene(c);
  uint8_t* base = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
  float4* tlas = (float4*)(base + c->dyn_off.tlas);
  const int build = mode == HK_TREE_SAH ? 1 : 0;
  HK_REQUIRE(launch_tree_build(c->stream, build, false, r, ni, c->rf_inst_lo, c->rf_inst_hi, c->lbvh_scratch, tlas, tlas + 1, 2u, c->threaded ? 8u : 1u) == 0, HK_E_HIP,
             "device build of the instance tree failed: %s", hipGetErrorString(hipGetLastError()));
  if (ne)
    HK_REQUIRE(launch_tree_build(c->stream, build, true, r, ne, nullptr, nullptr, c->lbvh_scratch, (float4*)(base + c->dyn_off.light_lo), (float4*)(base + c->dyn_off.light_hi),
                                 1u, 1u) == 0, HK_E_HIP, "device build of the light tree failed: %s", hipGetErrorString(hipGetLastError()));
  c->mirrors_stale = true;
  c->device_tree_builds += 1;
  return HK_OK;
}

// tools/wf_timeline.py: the 64 x 32 u64 the instrumented trace kernel left for the frame most recently rendered (HK_WF_TIMELINE=1)
int hk_debug_read_wf_timeline(hk_ctx* c, unsigned long long* out, uint32_t n) {
  HK_REQUIRE(c && out && n == 64u * 32u, HK_E_INVALID, "bad argument");
  HK_REQUIRE(c->wf.timeline, HK_E_NOT_READY, "no timeline: set HK_WF_TIMELINE=1 before the first frame of a scene beyond the LDS copy");
  HK_HIP(hipSetDevice(c->device));
  { const int rc = sync_all(c); if (rc) return rc; }
  HK_HIP(hipMemcpy(out, c->wf.timeline, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  int khz = 0;
  HK_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
  out[n - 1] = (unsigned long long)khz;  // (slot 31 of stage 63 - never a real stage: the rate of wall_clock64, kHz)
  return HK_OK;
}

// Test hook: the instance tree (ordering 0) and the light tree as the device holds them, converted back to the reference layout
// (navigators that took over their single leaf's role - fold_leaf_navigators - point at the leaf again)
int hk_debug_read_trees(hk_ctx* c, HkNode* tlas, uint32_t tlas_cap, HkNode* light, uint32_t light_cap) {
  HK_REQUIRE(c && (tlas || !tlas_cap) && (light || !light_cap), HK_E_INVALID, "NULL argument");
  HK_HIP(hipSetDevice(c->device));
  int rc;
  if ((rc = finalize_scene(c))) return rc;
  if ((rc = sync_all(c))) return rc;
  const uint32_t nt = (uint32_t)c->instance_nodes.size(), nl = (uint32_t)c->emissive_nodes.size();
  HK_REQUIRE(tlas_cap >= nt && light_cap >= nl, HK_E_INVALID, "need room for %u + %u nodes", nt, nl);
  const uint8_t* base = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
  auto convert = [](const std::vector<float4>& lo, const std::vector<float4>& hi, HkNode* out) {
    const uint32_t n = (uint32_t)lo.size();
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    for (uint32_t k = 0; k < n; ++k) {
      out[k].min[0] = lo[k].x; out[k].min[1] = lo[k].y; out[k].min[2] = lo[k].z;
      out[k].max[0] = hi[k].x; out[k].max[1] = hi[k].y; out[k].max[2] = hi[k].z;
      out[k].entry_index = bits(lo[k].w);
      out[k].exit_index = bits(hi[k].w);
    }
    for (uint32_t k = 0; k + 1 < n; ++k)
      if (out[k].entry_index >= HK_BVH_LEAF_FLAG && out[k + 1].entry_index == out[k].entry_index && out[k + 1].exit_index == out[k].exit_index) out[k].entry_index = k + 1;
  };
  if (nt) {
    std::vector<float4> both(2 * (size_t)nt), lo(nt), hi(nt);
    HK_HIP(hipMemcpy(both.data(), base + c->dyn_off.tlas, both.size() * 16, hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < nt; ++k) { lo[k] = both[2 * k]; hi[k] = both[2 * k + 1]; }
    convert(lo, hi, tlas);
  }
  if (nl) {
    std::vector<float4> lo(nl), hi(nl);
    HK_HIP(hipMemcpy(lo.data(), base + c->dyn_off.light_lo, (size_t)nl * 16, hipMemcpyDeviceToHost));
    HK_HIP(hipMemcpy(hi.data(), base + c->dyn_off.light_hi, (size_t)nl * 16, hipMemcpyDeviceToHost));
    convert(lo, hi, light);
  }
  return HK_OK;
}

// `commit`: advance the builder's previous-transform bookkeeping (once per update, whichever context sees it last)
static int refit_impl(hk_ctx* c, hk_scene_builder* b, uint32_t* moved_out, bool commit) {
  HK_REQUIRE(c && b, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(c->have_meshes && c->have_materials && c->have_instances, HK_E_NOT_READY, "hk_upload_scene must come first");
  HK_HIP(hipSetDevice(c->device));
  int rc;
  if ((rc = finalize_scene(c))) return rc;
  const uint32_t ni = (uint32_t)c->instances.size();
  HK_REQUIRE(builder_instance_count(b) == ni, HK_E_INVALID, "the builder has %u instances, the uploaded scene %u: instances were added or removed (use hk_upload_scene_instances)",
             builder_instance_count(b), ni);
  // which instances moved since the pose the device holds; their new host-side records (the reference's per-instance work,
  // instance.rs:286-325, kept in step so that a later host-side rebuild starts from the right poses)
  std::vector<uint32_t> moved;
  std::vector<hkd::RefitUpdate> records;
  for (uint32_t i = 0; i < ni; ++i) {
    InstanceDecl d;
    HK_REQUIRE(builder_instance_decl(b, i, &d), HK_E_NOT_READY, "the builder has unfinished mesh changes (hk_scene_builder_finish + hk_upload_scene first)");
    HkInstance& in = c->instances[i];
    HK_REQUIRE(d.material == in.material && memcmp(&d.mesh, &in.mesh, sizeof(HkMeshIndex)) == 0, HK_E_INVALID,
               "instance %u changed its mesh or material (use hk_upload_scene_instances)", i);
    if (memcmp(d.transform, in.model, 64) == 0) continue;
    float mn[3], mx[3], itm[16];
    HK_REQUIRE(instance_world_record(d.transform, d.aabb_center, d.aabb_half, mn, mx, itm), HK_E_INVALID, "singular transform of instance %u", i);
    hkd::RefitUpdate u;
    u.instance = i;
    u.moved = 1u;
    memcpy(u.model, d.transform, 64);
    memcpy(u.aabb_center, d.aabb_center, 12);
    memcpy(u.aabb_half, d.aabb_half, 12);
    records.push_back(u);
    moved.push_back(i);
  }
  if (moved_out) *moved_out = (uint32_t)moved.size();
  if ((rc = prepare_refit(c))) return rc;
  {  // instances that moved in the previous update and rest now: their `moved` flag goes (previous model = model)
    std::vector<uint8_t> now(ni, 0);
    for (uint32_t i : moved) now[i] = 1;
    for (uint32_t i : c->rf_last_moved)
      if (!now[i]) {
        hkd::RefitUpdate u{};
        u.instance = i;
        u.moved = 0u;
        records.push_back(u);
      }
  }
  if (records.empty()) {
    if (commit) builder_commit_transforms(b);
    return HK_OK;
  }
  // the records of moved emitters first (k_refit_emitters runs one wave per such record and on no other)
  uint32_t n_emitter_updates = 0, emitter_triangles = 0;
  {
    std::vector<uint8_t> is_emitter(ni, 0);
    for (const HkEmissive& e : c->emissives)
      if (e.instance < ni) is_emitter[e.instance] = 1;
    auto mid = std::stable_partition(records.begin(), records.end(), [&](const hkd::RefitUpdate& u) { return u.moved && is_emitter[u.instance]; });
    n_emitter_updates = (uint32_t)(mid - records.begin());
    for (uint32_t k = 0; k < n_emitter_updates; ++k)
      emitter_triangles = std::max(emitter_triangles, (c->instances[records[k].instance].mesh.node_count + 2u) / 3u);  // a BLAS over n triangles: 3n - 2 nodes
  }
  // pinned update records, double-buffered against the kernel that reads them
  const int k = c->rf_k;
  c->rf_k ^= 1;
  if (c->rf_pending[k]) {
    HK_HIP(hipEventSynchronize(c->rf_done[k]));
    c->rf_pending[k] = false;
  }
  if (c->rf_updates_cap[k] < records.size()) {
    if (c->rf_updates[k]) (void)hipHostFree(c->rf_updates[k]);
    c->rf_updates[k] = nullptr;
    c->rf_updates_cap[k] = 0;
    const size_t cap = records.size() + records.size() / 2 + 16;
    HK_HIP(hipHostMalloc((void**)&c->rf_updates[k], cap * sizeof(hkd::RefitUpdate), hipHostMallocDefault));
    c->rf_updates_cap[k] = cap;
  }
  if (!c->rf_done[k]) HK_HIP(hipEventCreateWithFlags(&c->rf_done[k], hipEventDisableTiming));
  memcpy(c->rf_updates[k], records.data(), records.size() * sizeof(hkd::RefitUpdate));
  // frames in flight keep reading the slot they were enqueued with: refit a copy in the spare slot (two-slot scenes), or in
  // place behind everything enqueued so far (scenes small enough for the LDS copy have one slot)
  if ((rc = begin_device_update(c))) return rc;
  const hkd::RefitScene r = refit_scene(c);
  uint8_t* base = c->scene_mem + (size_t)c->slot * c->dyn_capacity;
  launch_refit(c->stream, r, c->rf_updates[k], (uint32_t)records.size(), n_emitter_updates, emitter_triangles, nullptr, (float4*)(base + c->dyn_off.tlas), (uint32_t)c->instance_nodes.size(),
               c->threaded ? 8u : 1u, (float4*)(base + c->dyn_off.light_lo), (float4*)(base + c->dyn_off.light_hi), (uint32_t)c->emissive_nodes.size());
  HK_HIP(hipGetLastError());
  HK_HIP(hipEventRecord(c->rf_done[k], c->stream));
  c->rf_pending[k] = true;
  // the update is enqueued: now the host mirrors of the moved instances follow (an error above leaves host and device agreeing)
  for (const hkd::RefitUpdate& u : records) {
    if (!u.moved) continue;
    HkInstance& in = c->instances[u.instance];
    if (c->prev_models.size() != 16 * (size_t)ni) {
      c->prev_models.resize(16 * (size_t)ni);
      for (uint32_t j = 0; j < ni; ++j) memcpy(&c->prev_models[16 * (size_t)j], c->instances[j].model, 64);
    }
    memcpy(&c->prev_models[16 * (size_t)u.instance], in.model, 64);
    memcpy(in.model, u.model, 64);
    (void)instance_world_record(u.model, u.aabb_center, u.aabb_half, in.min, in.max, in.inverse_transpose_model);
  }
  for (const hkd::RefitUpdate& u : records)
    if (!u.moved && c->prev_models.size() == 16 * (size_t)ni) memcpy(&c->prev_models[16 * (size_t)u.instance], c->instances[u.instance].model, 64);
  update_shared_transform(c);  // (a scene in one slot is refit in place: nothing else looks at the new poses before the next frame)
  c->d_prev_models = c->rf_prev_models;
  c->rf_last_moved = moved;
  c->mirrors_stale = true;
  c->device_refits += 1;
  if (commit) builder_commit_transforms(b);
  return HK_OK;
}
int hk_refit_scene_instances(hk_ctx* c, hk_scene_builder* b, uint32_t* moved_out) { return refit_impl(c, b, moved_out, true); }
int hk_upload_textures(hk_ctx* c, const HkImageDesc* images, uint32_t n) {
  HK_REQUIRE(c && (images || !n), HK_E_INVALID, "NULL argument");
  std::vector<hk_ctx::HostTexture> tex(n);
  for (uint32_t i = 0; i < n; ++i) {
    const HkImageDesc& d = images[i];
    HK_REQUIRE(d.rgba8 && d.width && d.height && d.width <= 16384 && d.height <= 16384, HK_E_INVALID, "image %u: bad pointer or size", i);
    HK_REQUIRE(d.address_u <= HK_ADDRESS_MIRROR_REPEAT && d.address_v <= HK_ADDRESS_MIRROR_REPEAT, HK_E_INVALID, "image %u: bad address mode", i);
    tex[i].w = d.width;
    tex[i].h = d.height;
    tex[i].flags = (d.is_srgb ? 1u : 0u) | (d.filter_linear ? 2u : 0u) | (d.address_u << 4) | (d.address_v << 6);
    tex[i].texels.resize((size_t)d.width * d.height);
    memcpy(tex[i].texels.data(), d.rgba8, tex[i].texels.size() * 4);
  }
  c->textures.swap(tex);
  c->textures_dirty = true;
  c->dynamic_dirty = true;
  return HK_OK;
}
int hk_upload_noise(hk_ctx* c, const uint8_t* rgba, size_t bytes) {
  HK_REQUIRE(c && rgba && bytes == 16u * 64u * 64u * 4u, HK_E_INVALID, "noise must be 16 tiles of 64x64 RGBA8 (262144 bytes)");
  HK_HIP(hipSetDevice(c->device));
  std::vector<uint32_t> words(16u * 64u * 64u);
  memcpy(words.data(), rgba, bytes);
  int rc = c->d_noise.upload(words);
  if (rc) return rc;
  c->scene.noise = c->d_noise.p;
  c->have_noise = true;
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
  }
  for (uint32_t b = 0; b < HK_BUF_PARKED_TO0; ++b) {  // (the parked-store planes beyond: on first use, ensure_parked)
    size_t n = buffer_is_full_size(b) ? (size_t)c->W * c->H : (buffer_is_upscaled(b) ? (size_t)c->UW * c->UH : (size_t)c->RW * c->RH);
    size_t bytes = n * buffer_bpp(b);
    HK_HIP(hipMalloc(&c->buf[b], bytes));
    HK_HIP(hipMemset(c->buf[b], 0, bytes));  // zeroed reservoirs, light.rs:352-360
    c->buf_bytes[b] = bytes;
  }
  const size_t nf = (size_t)c->W * c->H, nr = (size_t)c->RW * c->RH;
  if (c->flags & HK_CTX_DETERMINISTIC_SCATTER) { const int rc_ = ensure_parked(c); if (rc_) return rc_; }
  if (!(c->flags & HK_CTX_DETERMINISTIC_SCATTER)) {  // (the verification mode parks its scatter stores: no elision there)
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
    }
    c->mapped_parity = f->number & 1u;
  }
  // the history halo of this frame: a count the host set, or the bound every rank derives from the same uniforms
  c->history_now = 0;
  if (c->band_count > 1 && c->RH > 0) {
    if (c->history_rows != HK_HISTORY_AUTO) {
      c->history_now = std::min(c->history_rows, (uint32_t)c->RH);
    } else {
      const int rc = derive_history_rows(c, &c->history_now);
      if (rc) return rc;
    }
    if (parks_across_bands(c)) { const int rc = ensure_parked(c); if (rc) return rc; }
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
    if ((rc = join_side(c))) return rc;
    // last frame's a-trous levels may still be running on post_stream: this frame's primary rays and light passes do not touch
    // what they read - provided the double-buffered planes really flipped (a host that renders two frames of the same parity in a
    // row, or shards the frame into bands, or timed passes, gets the serial order)
    if (c->post_pending && (c->mapped_parity == c->post_parity || c->band_count > 1 || (flags & HK_FRAME_EXTERNAL_GBUFFER)) && (rc = join_post(c))) return rc;
    if (c->timing_mask) {
      (void)hipEventRecord(c->frame_start, c->stream);
    }
    int f0, f1;
    full_rows_for(c, clampr(b0 - den - sp), clampr(b1 + den + sp), &f0, &f1);
    bool albedo_done = false;
    if (!(flags & HK_FRAME_EXTERNAL_GBUFFER)) {
      if (f1 > f0) {  // the prepass also fills the albedo of every pixel it covers (a superset of the rows albedo needs)
        const DFrame fr = make_dframe(c);
        GBuffer g = make_gbuffer(c);
        g.albedo_out = (uint2*)c->buf[HK_BUF_ALBEDO];
        unsigned long long* counters = (c->flags & HK_CTX_COUNT_RAYS) ? c->d_counters : nullptr;
        const Jitter j = prepass_jitter(c);
        ScopedTimer timer(c, HK_PASS_PREPASS);
        launch_prepass(c->stream, c->scene, fr, c->view.inverse_view_proj, c->view.view_proj, c->pview.view_proj, c->d_prev_models, j.x, j.y, g, f0, f1, counters);
        HK_HIP(hipGetLastError());
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
    if ((rc = join_side(c))) return rc;              // exchange B / demodulation read all three channels
  } else if (stage == HK_STAGE_POST_PROCESS) {
    if ((rc = join_side(c))) return rc;
    if ((rc = join_post(c))) return rc;              // the denoiser's internal planes: last frame's levels come first
    if (st->denoise) {                               // post_process.rs:1190-1224
      const uint32_t nch = st->indirect_bounces == 0 ? 2u : 3u;  // post_process.rs:949-954
      if (c->derived_dirty) {
        launch_derive_planes(c->stream, make_gbuffer(c), c->depth_plane, c->dn_g, c->W, 0, c->H);
        c->derived_dirty = false;
      }
      // the reference's per-channel loop, with the channels of each step fused into one launch
      if ((rc = run_demodulation_fused(c, nch, clampr(b0 - 15), clampr(b1 + 15)))) return rc;
      // demodulation was the last reader of the light passes' render / variance planes: from here on nothing this frame still
      // does is touched by the next frame's light passes - the four levels go to post_stream (a single-band, untimed frame)
      const uint32_t level_bits = (1u << HK_PASS_DENOISE_L0) | (1u << HK_PASS_DENOISE_L1) | (1u << HK_PASS_DENOISE_L2) | (1u << HK_PASS_DENOISE_L3);
      const bool pipelined = c->post_stream && c->band_count == 1 && !(c->timing_mask & level_bits) && !c->comm && c->albedo_twin;
      hipStream_t main_stream = c->stream;
      if (pipelined) {
        HK_HIP(hipEventRecord(c->post_fork, c->stream));
        HK_HIP(hipStreamWaitEvent(c->post_stream, c->post_fork, 0));
        c->stream = c->post_stream;
      }
      rc = run_denoise_fused(c, nch, 0, clampr(b0 - 7), clampr(b1 + 7));
      if (!rc) rc = run_denoise_fused(c, nch, 1, clampr(b0 - 3), clampr(b1 + 3));
      if (!rc) rc = run_denoise_fused(c, nch, 2, clampr(b0 - 1), clampr(b1 + 1));
      if (!rc) rc = run_denoise_fused(c, nch, 3, b0, b1, true);  // + tone mapping (post_process.rs:1226-1234) in the same launch
      c->stream = main_stream;
      if (rc) return rc;
      if (pipelined) {
        HK_HIP(hipEventRecord(c->post_done, c->post_stream));
        c->post_pending = true;
        c->post_parity = c->mapped_parity;
      }
      if (c->timing_mask) {
        (void)hipEventRecord(c->frame_stop, pipelined ? c->post_stream : c->stream);
        c->frame_timed = true;
      }
    } else {
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
  const uint32_t hist = c->history_now << 8;
  for (uint32_t s = 0; s <= HK_STAGE_POST_PROCESS; ++s) {
    if (ex && (s != HK_STAGE_TEMPORAL || hist) && (rc = comm_exchange(c, s <= HK_STAGE_SPATIAL ? (s | hist) : s, st))) return rc;
    if ((rc = hk_frame_stage(c, s, st, flags))) return rc;
  }
  if (flags & HK_FRAME_ANTIALIAS) {
    if (ex && (rc = comm_exchange(c, HK_STAGE_ANTIALIAS | hist, st))) return rc;
    if ((rc = hk_frame_stage(c, HK_STAGE_ANTIALIAS, st, flags))) return rc;
    if (ex && st->upscale_kind == HK_UPSCALE_FSR1 && (rc = comm_exchange(c, HK_STAGE_UPSCALE, st))) return rc;
    if ((rc = hk_frame_stage(c, HK_STAGE_UPSCALE, st, flags))) return rc;
  }
  // rank 0 collects the finished image (the post stream's tone mapping has to be in before the rows leave)
  if (ex && (flags & HK_FRAME_GATHER)) {
    if ((rc = join_all(c))) return rc;
    return comm_gather(c, hk_final_buffer(st, flags), 0u);
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
  unsigned long long h[8] = {};
  HK_HIP(hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
  out->rays_primary = h[0];
  out->rays_tlas = h[1];
  out->rays_blas = h[2];
  out->walk_node_steps = h[3];
  out->walk_triangle_tests = h[4];
  out->walk_instance_entries = h[5];
  out->walk_closest_hits = h[6];
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
  *out = c->scene.flat_mode ? HK_TRAVERSAL_ONE_LEVEL : c->threaded ? HK_TRAVERSAL_THREADED : HK_TRAVERSAL_REFERENCE;
  if (orderings) *orderings = c->scene.flat_mode ? c->scene.flat_mask + 1u : c->threaded ? 8u : 1u;
  return HK_OK;
}

int hk_reset_stats(hk_ctx* c) {
  HK_REQUIRE(c, HK_E_INVALID, "ctx is NULL");
  HK_HIP(hipSetDevice(c->device));
  { const int rc_ = sync_all(c); if (rc_) return rc_; }
  drain_timers(c);
  HK_HIP(hipMemset(c->d_counters, 0, 8 * sizeof(unsigned long long)));
  c->frames = 0;
  for (int i = 0; i < HK_TIMING_SLOTS; ++i) {
    c->slot_ms[i] = 0.0;
    c->slot_launches[i] = 0;
  }
  return HK_OK;
}

int hk_debug_math(hk_ctx* c, uint32_t op, const float* x, const float* y, float* out, size_t n) {
  HK_REQUIRE(c && x && out && op <= 20, HK_E_INVALID, "bad argument");
  const size_t xin = (op >= 16 && op <= 19 ? 16 : 1) * n;
  if (n == 0) return HK_OK;
  HK_HIP(hipSetDevice(c->device));
  float *dx = nullptr, *dy = nullptr, *dout = nullptr;
  HK_HIP(hipMalloc((void**)&dx, xin * 4));
  HK_HIP(hipMalloc((void**)&dout, n * 4));
  HK_HIP(hipMemcpy(dx, x, xin * 4, hipMemcpyHostToDevice));
  if (y) {
    HK_HIP(hipMalloc((void**)&dy, n * 4));
    HK_HIP(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
  }
  launch_debug_math(c->stream, op, dx, dy, dout, n);
  HK_HIP(hipStreamSynchronize(c->stream));
  HK_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(dx);
  (void)hipFree(dout);
  if (dy) (void)hipFree(dy);
  return HK_OK;
}

}  // extern "C"

int hk::refit_instances_impl(hk_ctx* c, hk_scene_builder* b, uint32_t* moved, bool commit) { return refit_impl(c, b, moved, commit); }
