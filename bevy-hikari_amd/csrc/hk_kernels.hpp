// hk_kernels.hpp - kernel argument bundles and host launchers (implemented in kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "hk_device.hpp"

namespace hkd {

// group 1 of the reference pipelines (deferred_bindings.wgsl:3-12), row-major planes
struct GBuffer {
  float4* __restrict__ position;          // rgba32f: world xyz, clip depth
  uint32_t* __restrict__ normal;          // rgba8snorm
  float2* __restrict__ depth_gradient;    // rg32f
  float2* __restrict__ instance_material; // rg32f, id + 0.5
  float4* __restrict__ velocity_uv;       // rgba32f
  // derived planes (not part of the reference's bindings): written by k_prepass / k_derive_planes
  float* __restrict__ depth;              // position.w alone: 4-B taps for the spatial-reuse ray march
  float4* __restrict__ dn_g;              // (normalised stored normal xyz, instance id + 0.5): one 16-B tap for the denoiser
  // optional: k_prepass also evaluates full_screen_albedo (light.wgsl:1019-1042) for the pixels it covers and
  // writes it here (the frame path); null = the separate k_full_screen_albedo dispatch does it
  uint2* __restrict__ albedo_out;
};
// Uniform-tile store elision.  The reference stores a packed reservoir to up to three buffers for EVERY pixel of EVERY light
// dispatch, background included (light.wgsl:1058-1069,1279-1287,1527-1532) - on a frame that is 64 % sky that is ~1 GB of HBM
// writes per frame that rewrite the same constant record over itself.  Per reservoir buffer and 8x8 tile the library keeps a
// small record: `id` = hash of the record ALL 64 slots of the tile hold, `src` = the id of the input tile that record was derived
// from (spatial_reuse's background branch re-packs its input), `valid` = serial number of the dispatch that wrote the tile
// uniformly (0: contents unknown / not uniform), `poison` = highest serial of a dispatch that stored into the tile from OUTSIDE
// (the reprojected scatter stores to previous_spatial).  A wave whose 64 pixels are all background may skip its tile store iff
// the tile already holds exactly that record: valid > poison && id == the record's id.  Contents of every buffer are therefore
// bit for bit what they are without the elision; only redundant traffic disappears.
struct TileMeta {
  unsigned long long id, src;
  uint32_t valid, poison, pad0, pad1;
};
// groups 5 + 6 for one light channel (light.wgsl:26-31,68-75; ping-pong light.rs:518-546)
struct LightTargets {
  const PackedReservoir* __restrict__ previous;  // binding 0
  PackedReservoir* current;                      // binding 1
  PackedReservoir* previous_spatial;             // binding 2
  PackedReservoir* spatial;                      // binding 3
  float* variance;                               // r32f
  uint2* render;                                 // rgba16f
  // HK_CTX_DETERMINISTIC_SCATTER (verification mode, nullptr otherwise): stores to previous_spatial are parked here
  // and applied by k_resolve_scatter - the store of the highest thread index wins, the oracle's resolution of the
  // reference's write-write race (light.wgsl:1063,1092-1095,1199-1202,1456-1459)
  int* det_winner;               // per previous_spatial slot: highest pixel index that stores to it (-1: none)
  int* det_to;                   // per pixel: the slot its parked store goes to (-1: none)
  PackedReservoir* det_pending;  // per pixel: the parked value
  // Round 6, single contexts: the LIGHT form of the same rule (hk_light.hpp store_previous_spatial, kernels.hip k_resolve_scatter_lite).
  // A store to the pixel's OWN slot goes straight to previous_spatial - one pixel owns a slot, such stores do not race with each other -
  // and only a store to ANOTHER pixel's slot is parked and competes in det_winner.  The resolve pass applies the highest such store
  // unless the slot's owner stored to it too and has the higher index: the same winner as the full rule, without 68 B per pixel of
  // parked traffic, with the uniform-tile store elision and the frame pipelining left on.
  int det_lite;
  // uniform-tile store elision (TileMeta above); all null when it is off (verification mode, band-sharded or row-unaligned dispatches)
  TileMeta* m_current;
  TileMeta* m_spatial;
  TileMeta* m_previous_spatial;
  uint32_t serial;  // of this dispatch, > 0, increasing
  int tiles_x;      // 8x8 tiles per row of the render image
  int rw;           // render width (reservoir index = x + rw * y)
};
// The queue-based schedule of indirect_lit_ambient (kernels_wavefront.hip): what a path carries from stage to stage, in
// planes indexed by path slot, plus the ray queues and their counters.  `cap` = the most paths one dispatch can have
// (the pixels of the render image).
struct WfBuffers {
  uint32_t* ctr;       // 192 counters, zeroed per dispatch: [s] paths alive at bounce s, [64 + s] shadow rays / [128 + s] claimed rays of trace stage s
  uint32_t* pixel;     // [slot] x + rw * y
  float4* state;       // 9 planes of `cap`: random | position, pdf | normal, pending | transport | radiance | first hit position | first hit normal | radiance to add if the shadow ray is clear | if it is occluded
  float4 *cr0, *cr1;   // closest-hit ray of the bounce: (origin, -), (direction, pdf of the direction)
  float4 *sr0, *sr1;   // shadow ray of the bounce: (origin, max_distance), (direction, early-out distance)
  uint32_t* sr2;       // ... and the instance its walk excludes (the sampled emitter)
  float4* ch0;         // closest hit: (distance, u, v, primitive index bits)
  uint32_t* ch1;       // ... its instance index (U32_MAX: miss)
  uint32_t* sh;        // instance the shadow ray hit (U32_MAX: clear)
  uint32_t* alive[2];  // slots alive at a bounce (= the closest-hit rays of that bounce's trace stage), ping-pong
  uint32_t* shadow[2]; // slots whose shadow ray the trace stage walks, ping-pong
  uint32_t cap;
  // The persistent schedule (round 6; kernels_wavefront.hip k_wf_trace_wide<.., PATHS = true>): every bounce in ONE launch, a path stays
  // with the wave that claimed it.  Per bounce n: the two outcomes of its shadow ray (planes 2n, 2n + 1 of pb_add) and the instance that
  // ray hit (plane n of pb_sh) - a shadow ray is off the path's critical chain, nothing waits for it before k_wf_final.  `local`: 512
  // u32 per wave of the launch - a ring of 256 rays its own shading emitted and a ring of 256 paths whose closest hit waits for shading.
  float4* pb_add;
  uint32_t* pb_sh;
  uint32_t* local;
  uint32_t pb_bounces;  // bounces the per-bounce planes were allocated for (0: none - the staged schedule)
  // HK_WF_TIMELINE=1 (tools/wf_timeline.py; null otherwise, and then the instrumented instantiation of k_wf_trace never runs): 32
  // u64 per trace stage, zeroed per dispatch - [0] ~first wave start, [1] ~first time a wave found the queue exhausted, [2] last
  // wave exit (wall_clock64 ticks; ~x = UINT64_MAX - x so that atomicMax keeps the minimum), [3] sum of the waves' resident ticks,
  // [4] waves, [5] most node steps of one ray, [6] node steps, [7] rays, [8..23] rays by floor(log2(node steps + 1))
  // HK_CTX_COUNT_WALKS (round 5): the COUNTING twin of k_wf_trace_wide writes [0..4] as above and [8] records fetched, [9] of them in
  // the instance tree, [10] triangle tests, [11] instance entries, [12] rays claimed, [13] of them any-hit, [14] closest hits found,
  // [15] pieces of walks handed to idle lanes
  unsigned long long* timeline;
  uint32_t timeline_mode;  // 0: none (the product), 1: the timeline twin, 2: the counting twin
};
// Wide trees for the queue-based trace stage and the primary rays of scenes in global memory (round 4; kernels_wavefront.hip
// k_build_wide / k_wf_trace_wide, kernels.hip k_prepass<*, 4>, hk_wide.hpp; DESIGN 4 "Wide walk").
// One 128-B record per INNER node of a flatten_custom tree, at the node's own position in a parallel array: the boxes and links
// of its (up to four) grandchildren - four 32-B box-nodes, lo = (min.xyz, link), hi = (max.xyz, -).  link: HK_LEAF | id = a leaf
// (triangle of the mesh / instance), < HK_LEAF = position of an inner node (its record), 0xFFFFFFFF = no child.  The record of a
// tree's root (which flatten_custom does not store) sits in the tree's LAST slot (always a leaf's).  ONE ordering: the walk keeps
// a per-lane stack and takes the children nearest first, so the eight direction-threaded copies are not needed here.
// Ranks (round 5): the position of every leaf in the REFERENCE's own flattening (ordering 0) - the order in which the reference's
// stackless walk meets the leaves.  Of two candidates at exactly the same distance the reference keeps the one it meets first
// (light.wgsl:415-424: `<`), i.e. the one of smaller rank - instance leaves first, then triangle leaves inside the instance; the
// wide walk, which meets them nearest-box first, decides the tie by the ranks and so returns the reference's hit, bit for bit.
struct WideTrees {
  const float4* tlas;   // records of the instance tree: 8 float4 per slot, tlas_count slots
  const float4* blas;   // records of every mesh tree: slot node_offset + local position
  const uint32_t* tlas_rank;  // [instance] position of its leaf in the instance tree (ordering 0)
  const uint32_t* blas_rank;  // [primitive] position of its leaf in its mesh tree (ordering 0)
  uint32_t tlas_count;  // (its root: slot tlas_count - 1)
  unsigned long long* lost;  // counts stack entries that fit neither part (HkStats::wide_stack_lost)
  uint32_t* spill;      // stack entries beyond the LDS part: HK_WIDE_SPILL u32 per lane of the persistent launch (the trace stage;
                        // nullptr where only the prepass - whose lanes keep the rest in private memory - walks them)
};
// Instance motion on the device (kernels_scene.hip): the arrays of the instance-level region the refit kernels rewrite, plus
// the refit's own side arrays.
struct RefitUpdate {       // one record per instance whose transform changed (read from pinned host memory)
  uint32_t instance;
  uint32_t moved;          // 0: the instance moved in the previous update but not in this one (its `moved` flag is cleared)
  float model[16];         // new model matrix, column-major
  float aabb_center[3], aabb_half[3];  // the mesh's local box (bevy Aabb), instance.rs:286-296
};
struct RefitScene {
  DInstance* instances;
  float4* prev_models;                   // 4 columns per instance
  float4 *inst_lo, *inst_hi;             // world AABB per instance
  const uint32_t* emissive_of_instance;  // emitter index or U32_MAX
  DEmissive* emissives;
  float2* alias;
  float* alias_scratch;                  // 5 floats per alias entry
  const float4* materials;
  const float4 *tri_v0, *tri_v1, *tri_v2;
};
// groups 3 + 4 of the denoise pipeline (denoise.wgsl:10-28), for up to three channels per launch
struct DemodTargets {
  const uint2* __restrict__ albedo;     // rgba16f, full size
  const float* variance[3];             // light variance per channel
  const uint2* render[3];               // light render per channel
  uint2* output[3];                     // internal_texture_0 per channel
  float* internal_variance[3];
};
struct DenoiseTargets {
  const uint2* __restrict__ albedo;
  const float4* __restrict__ dn_g;              // (normalize(unpack4x8snorm(normal)), instance id + 0.5) per full-size pixel
  const float* __restrict__ depth;              // position.w per full-size pixel
  const float2* __restrict__ depth_gradient;
  const uint2* input[3];                        // internal_texture_<level> per channel
  uint2* output[3];                             // internal_texture_<level+1> / denoise_render[channel]
  const float* internal_variance[3];
  // optional, level 3 with all channels in the launch: also apply tone_mapping.wgsl:21-32 to the sum of the
  // (f16-rounded) channel outputs and write tone_mapping_output (the frame path); null = separate dispatch
  uint2* tone_mapped;
  float clear_color[4];
};

// what an a-trous tap needs of a G-buffer pixel besides its depth: the normal exactly as the filter derives it
// from the rgba8snorm texel (denoise.wgsl:226,262: normalize(textureLoad(normal_texture).xyz)) and the instance id
__device__ __forceinline__ float4 denoise_geometry(uint32_t packed_normal, float instance) {
  const f3 n = normalize(xyz(unpack4x8snorm(packed_normal)));
  return make_float4(n.x, n.y, n.z, instance);
}

struct Pixel { int x, y; bool valid; };

// ---- uniform-tile store elision helpers: every call is made by ALL live lanes of a wave with wave-uniform arguments
__device__ __forceinline__ unsigned long long record_id(const PackedReservoir& p) {
  const uint32_t w[16] = {p.radiance.x, p.radiance.y, p.random.x, p.random.y, f2u(p.visible_position.x), f2u(p.visible_position.y), f2u(p.visible_position.z),
                          f2u(p.visible_position.w), f2u(p.sample_position.x), f2u(p.sample_position.y), f2u(p.sample_position.z), f2u(p.sample_position.w),
                          p.visible_normal, p.sample_normal, p.reservoir.x, p.reservoir.y};
  unsigned long long h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    h ^= (unsigned long long)w[k] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
  }
  return h | 1ull;
}
__device__ __forceinline__ bool wave_leader() { return (int)__lane_id() == __ffsll((long long)__ballot(1)) - 1; }
// the 8x8 tile this wave covers (wave-uniform by construction of pixel_of_thread, valid lane or not)
__device__ __forceinline__ int wave_tile(const Pixel& px, int tiles_x) {
  const int lane = threadIdx.x & 63;
  return ((px.y - (lane >> 3)) >> 3) * tiles_x + ((px.x - (lane & 7)) >> 3);
}
__device__ __forceinline__ bool tile_holds(const TileMeta* m, int tile, unsigned long long id) {
  const TileMeta v = m[tile];
  return v.valid > v.poison && v.id == id;
}
__device__ __forceinline__ void tile_mark(TileMeta* m, int tile, unsigned long long id, unsigned long long src, uint32_t serial) {
  if (wave_leader()) {
    m[tile].id = id;
    m[tile].src = src;
    m[tile].valid = serial;
  }
}
__device__ __forceinline__ void tile_unknown(TileMeta* m, int tile) {
  if (wave_leader()) m[tile].valid = 0u;
}

// A wave's 64 reservoir records, written as full cache lines.  A lane holds its pixel's 64-B record in four 16-B
// chunks; storing them directly makes every store instruction touch 64 different lines, 16 B each, and the L2 has to
// stitch the lines together (nontemporal stores, which skip that, run these kernels 2x slower).  Here the wave
// transposes through LDS: instruction k writes chunks 64k..64k+63 of the tile in memory order, i.e. two rows of
// eight pixels = 2 x 512 contiguous bytes.  ALL lanes of the wave must call it (wave-uniform control flow);
// `write` says whether this lane's record is to be stored.  lds: 4 x 65 uint4 per wave (padded against bank conflicts).
constexpr int HK_TILE_LDS_UINT4 = 4 * 65;
__device__ __forceinline__ void store_packed_tile(uint4* lds, PackedReservoir* __restrict__ buf, int width, const Pixel& px, const PackedReservoir& p, bool write) {
  const int lane = threadIdx.x & 63;
  const unsigned long long wmask = __ballot(write);
  if (wmask == 0ull) return;
  lds[0 * 65 + lane] = make_uint4(p.radiance.x, p.radiance.y, p.random.x, p.random.y);
  lds[1 * 65 + lane] = make_uint4(f2u(p.visible_position.x), f2u(p.visible_position.y), f2u(p.visible_position.z), f2u(p.visible_position.w));
  lds[2 * 65 + lane] = make_uint4(f2u(p.sample_position.x), f2u(p.sample_position.y), f2u(p.sample_position.z), f2u(p.sample_position.w));
  lds[3 * 65 + lane] = make_uint4(p.visible_normal, p.sample_normal, p.reservoir.x, p.reservoir.y);
  __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order; this only pins the compiler's schedule
  // tile origin: this lane's pixel minus its position in the 8x8 tile (valid or not, the arithmetic is the same)
  const int tile_x0 = px.x - (lane & 7), tile_y0 = px.y - (lane >> 3);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int m = k * 64 + lane, record = m >> 2, part = m & 3;
    const uint4 v = lds[part * 65 + record];
    if ((wmask >> record) & 1ull) {
      const int x = tile_x0 + (record & 7), y = tile_y0 + (record >> 3);
      reinterpret_cast<uint4*>(buf + (x + width * y))[part] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();  // the next call reuses the buffer
}

// XCD_BANDS = true: linear workgroup ids are remapped so that each XCD (dispatcher: block b -> XCD b % 8) works on one
// contiguous eighth of the image and gather passes find their neighbours' data in that XCD's L2.  false: tiles go
// round-robin over the XCDs - for the ray kernels, whose cost per tile varies a lot and which gather nothing from
// neighbouring pixels, balance across the XCDs is worth more than locality.
template <bool XCD_BANDS = true>
__device__ __forceinline__ Pixel pixel_of_thread(int width, int row_begin, int row_end) {
  const int tiles_x = (width + 15) >> 4;
  const uint32_t nb = gridDim.x;
  uint32_t b = blockIdx.x;
  const uint32_t per = nb >> 3;
  // (a launch of a few workgroups per CU - one band of a multi-GPU split - has nothing to balance: keep the bands)
  if ((XCD_BANDS || nb < 2048u) && per > 0 && b < per * 8u) b = (b & 7u) * per + (b >> 3);
  const int tile_x = (int)(b % (uint32_t)tiles_x), tile_y = (int)(b / (uint32_t)tiles_x);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  Pixel p;
  p.x = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
  p.y = row_begin + tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
  p.valid = p.x < width && p.y < row_end;
  return p;
}

// The streaming / stencil kernels' mapping (demodulation, a-trous): a wave covers W x (64 / W) pixels and a workgroup four such
// strips stacked, so that a wave's load of a 4-byte plane is whole 128-B lines instead of eight 32-B pieces of eight lines (the
// 8 x 8 tiles above suit gathers and rays).  XCD_BANDS as above.  Measured (Cornell 1080p, round 2): demodulation 0.071 ms with
// 8 x 8 tiles -> 0.047 (32 x 2) -> 0.040 (64 x 1, bands); the a-trous levels 0.072 -> 0.064 by going round-robin, any shape.
template <int W, bool XCD_BANDS>
__device__ __forceinline__ Pixel pixel_of_thread_rows(int width, int row_begin, int row_end) {
  constexpr int H = 64 / W;  // of a wave
  const int tiles_x = (width + W - 1) / W;
  uint32_t b = blockIdx.x;
  const uint32_t nb = gridDim.x, per = nb >> 3;
  if (XCD_BANDS && per > 0 && b < per * 8u) b = (b & 7u) * per + (b >> 3);
  const int tile_x = (int)(b % (uint32_t)tiles_x), tile_y = (int)(b / (uint32_t)tiles_x);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  Pixel p;
  p.x = tile_x * W + (lane & (W - 1));
  p.y = row_begin + tile_y * (4 * H) + wave * H + (lane / W);
  p.valid = p.x < width && p.y < row_end;
  return p;
}

}  // namespace hkd

#define HK_LDS_SCENE_BYTES 32768u  // scenes up to this size are copied into LDS by the ray kernels (lds_bytes_for)

namespace hk {
static inline dim3 grid_for_rows(int wave_width, int width, int rows) {  // pixel_of_thread_rows<wave_width, *>
  int wg_rows = 4 * (64 / wave_width);
  int tiles_x = (width + wave_width - 1) / wave_width, tiles_y = (rows + wg_rows - 1) / wg_rows;
  return dim3((unsigned)(tiles_x * tiles_y), 1, 1);
}
static inline dim3 grid_for(int width, int rows) {
  int tiles_x = (width + 15) / 16, tiles_y = (rows + 15) / 16;
  return dim3((unsigned)(tiles_x * tiles_y), 1, 1);
}

void launch_prepass(hipStream_t st, const hkd::DScene& sc, const hkd::DFrame& fr, const float* inverse_view_proj, const float* view_proj,
                    const float* prev_view_proj, const float4* prev_models, float jitter_x, float jitter_y, const hkd::GBuffer& g, int y0, int y1,
                    unsigned long long* counters, const hkd::WideTrees* wide = nullptr);
void launch_albedo(hipStream_t st, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g, void* albedo, int y0, int y1);
void launch_direct(hipStream_t st, bool emissive_lit, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g, const hkd::LightTargets& t,
                   int y0, int y1, unsigned long long* counters);
void launch_indirect(hipStream_t st, bool multiple_bounces, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g,
                     const hkd::LightTargets& t, int y0, int y1, unsigned long long* counters, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// the same dispatch as launch_indirect(multiple_bounces = true), scheduled through ray queues (kernels_wavefront.hip)
// wide: records of the trees for the trace stages (nullptr members = the threaded walk)
void launch_build_wide(hipStream_t st, const float4* nodes, uint32_t count, float4* wide, uint32_t* rank);  // one flatten_custom tree (ordering 0) -> its records; rank[leaf id] = the leaf's position
// what WideTrees::spill must hold: wide_trace_lanes(compute_units) x wide_spill_entries() u32 (the lanes of the persistent trace launch
// x the stack entries a lane keeps beyond LDS) - asked of the file that launches the kernel, so that the two cannot disagree
size_t wide_trace_lanes(int compute_units);
size_t wide_spill_entries();
// trace_events: nullptr, or 2 x (bounces + 1) events - start / stop of every trace launch (HK_TIMING_TRACE_STAGES).  persistent: every
// bounce in ONE launch (k_wf_trace_wide<.., PATHS>; needs the wide trees and WfBuffers::pb_* for the frame's bounces, else the stages run)
void launch_indirect_wavefront(hipStream_t st, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g, const hkd::LightTargets& t,
                               const hkd::WfBuffers& w, int y0, int y1, int compute_units, hipEvent_t start = nullptr, hipEvent_t stop = nullptr,
                               const hkd::WideTrees* wide = nullptr, hipEvent_t* trace_events = nullptr, bool persistent = false);
void launch_copy_region(hipStream_t st, void* dst, const void* src, size_t bytes);
void launch_gather_instance_boxes(hipStream_t st, const hkd::RefitScene& s, const float4* tlas, uint32_t tlas_count);
// the first n_emitter_updates records are the moved emitters (the largest of their meshes has emitter_triangles triangles)
void launch_refit(hipStream_t st, const hkd::RefitScene& s, const hkd::RefitUpdate* updates, uint32_t n_updates, uint32_t n_emitter_updates, uint32_t emitter_triangles,
                  uint32_t* failed, float4* tlas,
                  uint32_t tlas_count, uint32_t orderings, float4* light_lo, float4* light_hi, uint32_t light_count);
// LBVH rebuild of a flat skip-link BVH over n shapes (kernels_scene.hip): scratch size, and the build into `lo` / `hi` (`stride`
// float4 between consecutive nodes: 2 for the interleaved TLAS, 1 for the two planes of the light BVH)
size_t lbvh_scratch_bytes(uint32_t n, size_t* sort_temp_bytes);
int launch_tree_build(hipStream_t st, int mode /* 0 LBVH, 1 the reference's binned SAH */, bool light, const hkd::RefitScene& s, uint32_t n, const float4* box_lo,
                      const float4* box_hi, void* scratch, float4* lo, float4* hi, uint32_t stride, uint32_t orderings);
// windowed: 0 the plain form, 1 the windowed form (depth window + tap lists in LDS: kernels.hip), -1 by the size of the launch; returns
// whether the windowed form was launched
bool launch_spatial(hipStream_t st, bool emissive_lit, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g, const hkd::LightTargets& t,
                    int y0, int y1, int windowed, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);  // (events attached to the dispatch: HkStats timing)
void launch_derive_planes(hipStream_t st, const hkd::GBuffer& g, float* depth_plane, void* dn_g, int width, int y0, int y1);
void launch_count_geometry_rows(hipStream_t st, const float* depth, int width, int height, uint32_t* out);
void launch_demodulation(hipStream_t st, int nch, const hkd::DFrame& fr, const hkd::DemodTargets& d, int y0, int y1);
void launch_denoise(hipStream_t st, int level, int nch, int ffmask, const hkd::DFrame& fr, const hkd::DenoiseTargets& d, int y0, int y1);
void launch_tone_mapping(hipStream_t st, const hkd::DFrame& fr, const void* direct, const void* emissive, const void* indirect, void* out, int y0, int y1);
// anti-aliasing tail (kernels_aa.hip): raw plane pointers + sizes of one dispatch's bindings
struct AaBuffers {
  const void *position, *velocity_uv, *previous_position, *previous_velocity_uv, *instance_material;  // full size
  const float *depth, *previous_depth;  // position.w / previous_position.w alone: the depth-only taps read 4 B instead of 16
  int full_w, full_h;
  const void* render;           // rgba16f: tone_mapping_output[current] (SMAA) / TAA input
  int render_w, render_h;
  const void* previous_render;  // rgba16f: tone_mapping_output[previous] (SMAA) / taa_output[previous] (TAA)
  int previous_w, previous_h;
  void* output;                 // rgba16f
  int out_w, out_h;
};
void launch_smaa_tu4x(hipStream_t st, const AaBuffers& b, uint32_t frame_number, int y0, int y1);
void launch_smaa_tu4x_extrapolate(hipStream_t st, void* output, int out_w, int out_h, int render_w, int y0, int y1);
void launch_taa_jasmine(hipStream_t st, const AaBuffers& b, float blend, const float clear_color[4], int y0, int y1);
void launch_fsr_easu(hipStream_t st, const void* input, int in_w, int in_h, void* output, int out_w, int out_h, int y0, int y1);
void launch_fsr_rcas(hipStream_t st, const void* input, void* output, int w, int h, float sharpness, int y0, int y1);
// apply the parked scatter stores of pixels [p0, p1); [own0, own1) = the pixels this context dispatched itself (their winners are
// in), empty = all of them
void launch_resolve_scatter(hipStream_t st, const hkd::LightTargets& t, int p0, int p1, int own0, int own1);
// the light form (LightTargets::det_lite): rows [y0, y1) of the pass that just ran; `indirect`: the pass was indirect_lit_ambient (whose
// pixels are background also when the frame has no bounces, light.wgsl:1279)
void launch_resolve_scatter_lite(hipStream_t st, const hkd::LightTargets& t, const hkd::DFrame& fr, const float* depth_plane, bool indirect, int y0, int y1);
void launch_debug_math(hipStream_t st, uint32_t op, const float* x, const float* y, float* out, size_t n);
}  // namespace hk
