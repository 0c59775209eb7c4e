/*
 * hikari_hip.h - C ABI of libhikari_hip.so, the MI355X-native (HIP / gfx950) replacement for the
 * compute path of cryscan/bevy-hikari v0.3.15 (reference paths below are relative to that repo).
 *
 * What this boundary replaces
 * ---------------------------
 * In the reference the path sits behind three Bevy render-graph nodes that record wgpu compute /
 * raster work:  PrepassNode::run (src/prepass.rs:769-852), LightNode::run (src/light.rs:590-702)
 * and PostProcessNode::run (src/post_process.rs:1140-1234, denoise + tone mapping part).  Their
 * inputs are the storage buffers written in the Prepare stage (src/mesh_material/mesh.rs:43-64,
 * material.rs:139-203, instance.rs:82-108), the noise images (src/lib.rs:189-219) and the four
 * dynamic uniforms (src/prepass.rs:546-553).  A Rust `HikariPlugin` keeps all of that host code
 * and calls the functions below instead of creating wgpu pipelines (see INTEGRATION.md for the
 * `extern "C"` block).
 *
 * Conventions
 * -----------
 *  - plain C, no C++/torch types; every function returns HK_OK (0) or a negative HK_E* code and
 *    never throws.  The reference's nodes silently return Ok(()) when a resource is missing
 *    (light.rs:606-617); here the same situation is reported as HK_E_NOT_READY and nothing runs.
 *  - the library owns all device memory.  Host arrays passed to hk_upload_* are copied before
 *    the call returns.
 *  - one hk_ctx is bound to one HIP device and is used from one host thread at a time (Bevy runs
 *    the render graph sequentially on the render thread).  All kernels are enqueued on the
 *    context's HIP stream; hk_frame_wait() is the only synchronisation point.
 *  - Hk* data structs are byte-for-byte the std430 / std140 layouts the reference writes with
 *    encase (src/mesh_material/mod.rs:60-299, src/view.rs:105-123) so the Rust side can hand over
 *    its existing buffers unchanged.  Conversion to the device layout happens inside hk_upload_*.
 */
#ifndef HIKARI_HIP_H
#define HIKARI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libhikari_hip.so is built with -fvisibility=hidden (tools/build_lib.py): the entry points declared between this push and the pop at
 * the end of the header - and the hooks of hikari_hip_debug.h - are ALL it exports; a host that links it sees no internal symbol. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define HK_ABI_VERSION 8

/* ------------------------------------------------------------------ error codes */
#define HK_OK 0
#define HK_E_INVALID (-1)    /* bad argument (NULL, size mismatch, out-of-range enum) */
#define HK_E_NO_DEVICE (-2)  /* no HIP device / device id out of range */
#define HK_E_HIP (-3)        /* a HIP runtime call failed; see hk_last_error() */
#define HK_E_NOT_READY (-4)  /* scene / noise / size / uniforms not uploaded yet */
#define HK_E_NOMEM (-5)
#define HK_E_UNSUPPORTED (-6)

/* ------------------------------------------------------------------ scene data (std430) */

/* mesh_material_types.wgsl:3-8, mod.rs:67-73 (GpuVertexCompact), 32 B */
typedef struct HkVertex {
  float position[3];
  float u;
  float normal[3];
  float v;
} HkVertex;

/* mesh_material_types.wgsl:10-17, mod.rs:115-145 (GpuPrimitiveCompact), 48 B */
typedef struct HkPrimitiveVertex {
  float position[3];
  uint32_t index;
} HkPrimitiveVertex;
typedef struct HkPrimitive {
  HkPrimitiveVertex vertices[3];
} HkPrimitive;

/* mesh_material_types.wgsl:35-40, mod.rs:177-201 (GpuNode), 32 B.
 * Skip-link flat BVH node.  entry_index >= 0x80000000 marks a leaf (low bits = shape index);
 * leaf boxes are EMPTY (min=+inf, max=-inf) exactly as `bvh` 0.7.1 flatten_custom emits them. */
typedef struct HkNode {
  float min[3];
  uint32_t entry_index;
  float max[3];
  uint32_t exit_index;
} HkNode;
#define HK_BVH_LEAF_FLAG 0x80000000u

/* mesh_material_types.wgsl:19-23, mod.rs:469-476 (GpuMeshIndex), 16 B */
typedef struct HkMeshIndex {
  uint32_t vertex;
  uint32_t primitive;
  uint32_t node_offset;
  uint32_t node_count;
} HkMeshIndex;

/* mesh_material_types.wgsl:25-33, mod.rs:147-156 (GpuInstance), 176 B. Matrices column-major. */
typedef struct HkInstance {
  float min[3];
  uint32_t material;
  float max[3];
  uint32_t node_index;
  float model[16];
  float inverse_transpose_model[16];
  HkMeshIndex mesh;
} HkInstance;

/* mesh_material_types.wgsl:42-56, mod.rs:203-218 (GpuStandardMaterial), 80 B */
typedef struct HkMaterial {
  float base_color[4];
  uint32_t base_color_texture;
  uint32_t _pad0[3];
  float emissive[4];
  uint32_t emissive_texture;
  float perceptual_roughness;
  float metallic;
  uint32_t metallic_roughness_texture;
  float reflectance;
  uint32_t normal_map_texture;
  uint32_t occlusion_texture;
  uint32_t _pad1;
} HkMaterial;
#define HK_NO_TEXTURE 0xFFFFFFFFu

/* mesh_material_types.wgsl:58-61, mod.rs:220-226, 8 B */
typedef struct HkAliasEntry {
  float prob;
  uint32_t index;
} HkAliasEntry;

/* mesh_material_types.wgsl:63-71, mod.rs:228-237 (GpuEmissive), 64 B */
typedef struct HkEmissive {
  float emissive[4];
  float position[3];
  float radius;
  uint32_t instance;
  uint32_t _pad0;
  uint32_t alias_table[2]; /* x = offset, y = count */
  float surface_area;
  uint32_t node_index;
  uint32_t _pad1[2];
} HkEmissive;

/* ------------------------------------------------------------------ uniforms */

/* mesh_view_types.wgsl:3-20, view.rs:105-123 (FrameUniform), std140: 244 B -> 256 B */
typedef struct HkFrame {
  float kernel[3][4]; /* mat3x3, each column padded to vec4 */
  float halton[8][4];
  float clear_color[4];
  uint32_t number;
  uint32_t direct_validate_interval;
  uint32_t emissive_validate_interval;
  uint32_t indirect_bounces;
  uint32_t temporal_reuse;
  uint32_t emissive_spatial_reuse;
  uint32_t indirect_spatial_reuse;
  uint32_t max_temporal_reuse_count;
  uint32_t max_spatial_reuse_count;
  float max_reservoir_lifetime;
  float solar_angle;
  float max_indirect_luminance;
  float upscale_ratio;
  uint32_t _pad[3];
} HkFrame;

/* bevy_pbr 0.9.1 `View` uniform (bevy_pbr::mesh_view_types; used at light.wgsl:721-724,1040 and
 * prepass.wgsl:45,71,96).  Matrices column-major. 416 B. */
typedef struct HkView {
  float view_proj[16];
  float inverse_view_proj[16];
  float view[16];
  float inverse_view[16];
  float projection[16];
  float inverse_projection[16];
  float world_position[3];
  float _pad0;
  float viewport[4]; /* x, y, width, height in physical pixels */
} HkView;

/* mesh_view_types.wgsl:22-25, view.rs:31-35 (PreviousViewUniform), 128 B */
typedef struct HkPreviousView {
  float view_proj[16];
  float inverse_view_proj[16];
} HkPreviousView;

/* The three fields of bevy_pbr 0.9.1 `Lights` the path reads (light.wgsl:611,832,847-855):
 * directional_lights[0].{color, direction_to_light} and ambient_color.  With no directional light
 * bevy zero-fills entry 0; pass n_directional_lights = 0 and zeros to reproduce that. */
typedef struct HkLights {
  float directional_color[4];
  float direction_to_light[3];
  uint32_t n_directional_lights;
  float ambient_color[4];
} HkLights;

/* HikariSettings (src/lib.rs:400-455), field for field.  hk_settings_default() fills the
 * reference defaults (lib.rs:435-455). */
typedef enum HkTaa { HK_TAA_JASMINE = 0, HK_TAA_NONE = 1 } HkTaa;           /* lib.rs:466-472 */
typedef enum HkUpscaleKind { HK_UPSCALE_FSR1 = 0, HK_UPSCALE_SMAA_TU4X = 1 } HkUpscaleKind; /* lib.rs:474-487 */
typedef struct HkSettings {
  uint32_t direct_validate_interval;
  uint32_t emissive_validate_interval;
  uint32_t max_temporal_reuse_count;
  uint32_t max_spatial_reuse_count;
  float max_reservoir_lifetime;
  float solar_angle;
  uint32_t indirect_bounces;
  float max_indirect_luminance;
  float clear_color[4];
  uint32_t temporal_reuse;
  uint32_t emissive_spatial_reuse;
  uint32_t indirect_spatial_reuse;
  uint32_t denoise;
  uint32_t taa;           /* HkTaa */
  uint32_t upscale_kind;  /* HkUpscaleKind */
  float upscale_ratio;    /* clamped to [1,2] like Upscale::ratio(), lib.rs:500-504 */
  float upscale_sharpness;
} HkSettings;

/* ------------------------------------------------------------------ buffers & passes */

/* Screen-space resources (group 1, 5, 6 of the light pipeline and group 3/4 of the denoise
 * pipeline: deferred_bindings.wgsl:3-20, light.wgsl:26-31,68-75, denoise.wgsl:10-28). Formats are
 * the reference's texture formats flattened row-major (prepass.rs:43-47, light.rs:29-31,
 * post_process.rs:29): rgba32f = 16 B, rgba8snorm = 4 B, rg32f = 8 B, rgba16f = 8 B, r32f = 4 B,
 * PackedReservoir = 64 B (light.wgsl:35-43). */
typedef enum HkBuffer {
  HK_BUF_POSITION = 0,          /* rgba32f, full size: world xyz, w = clip depth (0 = background) */
  HK_BUF_NORMAL = 1,            /* rgba8snorm, full */
  HK_BUF_DEPTH_GRADIENT = 2,    /* rg32f, full */
  HK_BUF_INSTANCE_MATERIAL = 3, /* rg32f (id + 0.5), full */
  HK_BUF_VELOCITY_UV = 4,       /* rgba32f, full */
  HK_BUF_ALBEDO = 5,            /* rgba16f, full */
  HK_BUF_VARIANCE0 = 6,         /* r32f, scaled; +channel (0 sun, 1 emissive, 2 indirect) */
  HK_BUF_RENDER0 = 9,           /* rgba16f, scaled; +channel */
  HK_BUF_RESERVOIR0 = 12,       /* 64 B x full W*H; +k, k in 0..9 (light.rs:342-363) */
  HK_BUF_DENOISE_INTERNAL0 = 22,/* rgba16f, scaled; +level 0..3 */
  HK_BUF_DENOISE_INTERNAL_VARIANCE = 26, /* r32f, scaled */
  HK_BUF_DENOISE_RENDER0 = 27,  /* rgba16f, scaled; +channel */
  HK_BUF_TONE_MAPPED = 30,      /* rgba16f, scaled (tone_mapping.wgsl:21-32): tone_mapping_output[current] */
  /* Temporal anti-aliasing / upscale inputs and outputs (post_process.rs:621-748, prepass.rs:285-318).
   * The reference double-buffers these by frame: position / velocity_uv swap every frame
   * (prepass.rs:308-317), tone_mapping_output and taa_output are indexed by frame.number % 2
   * (post_process.rs:737,866-867).  Here all four follow frame.number % 2 of hk_frame_begin: ids
   * without PREVIOUS name the plane frame n writes, PREVIOUS_* the plane frame n-1 wrote.
   * hk_device_ptr of these ids is therefore only valid until the next hk_frame_begin, and a host
   * that supplies its own G-buffer writes it AFTER hk_frame_begin of that frame.
   * Since ABI 6 the same holds for HK_BUF_ALBEDO and HK_BUF_DEPTH_GRADIENT (no PREVIOUS ids: nothing reads last frame's): the
   * a-trous levels of frame n may still be reading them on their own stream while frame n + 1's primary rays write the other
   * parity's planes (frame pipelining; contexts created with HK_CTX_SINGLE_STREAM / _DETERMINISTIC_SCATTER / _COUNT_RAYS /
   * _TIME_PASSES keep one plane).  hk_read_buffer / hk_write_buffer always address the current frame's plane. */
  HK_BUF_PREVIOUS_POSITION = 31,
  HK_BUF_PREVIOUS_VELOCITY_UV = 32,
  HK_BUF_PREVIOUS_TONE_MAPPED = 33,
  HK_BUF_UPSCALE_OUTPUT = 34,   /* rgba16f, ceil(size * 2 / ratio): upscale_output[0] of the SMAA Tu4x path */
  HK_BUF_TAA_OUTPUT = 35,       /* rgba16f: taa_output[current]; size of UPSCALE_OUTPUT (SMAA) or scaled (FSR1) */
  HK_BUF_PREVIOUS_TAA_OUTPUT = 36,
  /* FSR1 (Upscale::Fsr1): upscale_output[0] (EASU result) is HK_BUF_UPSCALE_OUTPUT at the window size, upscale_output[1]
   * (RCAS result, what OverlayNode presents, overlay.rs:228) is this one: rgba16f, window size */
  HK_BUF_UPSCALE_SHARPENED = 37,
  /* Parked scatter stores (ABI 7).  A temporal dispatch stores a rejected history reservoir at the REPROJECTED
   * pixel of previous_spatial (light.wgsl:1063,1092-1095,1199-1202,1456-1459) - a slot some other thread, possibly some other
   * BAND, owns.  Where those stores are parked instead of raced (HK_CTX_DETERMINISTIC_SCATTER, and every band of a sharded frame
   * whose history halo is not empty) pixel i of channel c (0 sun, 1 emissive, 2 indirect) leaves the slot it stores to in
   * PARKED_TO0 + c (i32 per render pixel, -1 = no store) and the 64-B record in PARKED_RECORD0 + c; the highest pixel index that
   * stores to a slot wins.  Rows of these planes are what bands hand each other with exchange A under motion
   * (HK_STAGE_SPATIAL_WITH_HISTORY).  Allocated on first use: hk_device_ptr / hk_read_buffer fail with HK_E_INVALID before. */
  HK_BUF_PARKED_TO0 = 38,       /* i32, scaled; +channel */
  HK_BUF_PARKED_RECORD0 = 41,   /* 64 B, scaled; +channel */
  HK_BUF_COUNT = 44
} HkBuffer;

/* One compute dispatch of the reference.  `arg` selects the render channel for
 * the denoise passes and the a-trous level is part of the pass id. */
typedef enum HkPass {
  HK_PASS_PREPASS = 0,             /* prepass.wgsl:40-100 semantics, produced by primary rays */
  HK_PASS_FULL_SCREEN_ALBEDO = 1,  /* light.wgsl:1019-1042 */
  HK_PASS_DIRECT_LIT = 2,          /* light.wgsl:1044-1261, RENDER_EMISSIVE (sun) */
  HK_PASS_DIRECT_EMISSIVE = 3,     /* light.wgsl:1044-1261, EMISSIVE_LIT */
  HK_PASS_INDIRECT = 4,            /* light.wgsl:1263-1498; MULTIPLE_BOUNCES iff bounces >= 2 (light.rs:663-666) */
  HK_PASS_EMISSIVE_SPATIAL_REUSE = 5, /* light.wgsl:1503-1684, EMISSIVE_LIT */
  HK_PASS_INDIRECT_SPATIAL_REUSE = 6, /* light.wgsl:1503-1684 */
  HK_PASS_DEMODULATION = 7,        /* denoise.wgsl:135-162; arg = channel */
  HK_PASS_DENOISE_L0 = 8,          /* denoise.wgsl:215-319; arg = channel; +level 0..3 */
  HK_PASS_DENOISE_L1 = 9,
  HK_PASS_DENOISE_L2 = 10,
  HK_PASS_DENOISE_L3 = 11,
  HK_PASS_TONE_MAPPING = 12,       /* tone_mapping.wgsl:21-32 */
  HK_PASS_SMAA_TU4X = 13,          /* smaa.wgsl:81-188: current + reprojected previous sample of each output quad */
  HK_PASS_SMAA_TU4X_EXTRAPOLATE = 14, /* smaa.wgsl:239-271: the other two pixels of each quad */
  HK_PASS_TAA_JASMINE = 15,        /* taa.wgsl:75-170 */
  HK_PASS_FSR_EASU = 16,           /* FidelityFX FSR 1.0 edge-adaptive spatial upsampling: src/shaders/fsr/source.zip,
                                      ffx_fsr1.h FsrEasuF through FSR_Pass.glsl (the blob fsr_pass_easu.spv), post_process.rs:1277-1293 */
  HK_PASS_FSR_RCAS = 17,           /* ... robust contrast-adaptive sharpening, FsrRcasF (fsr_pass_rcas.spv), post_process.rs:1295-1308 */
  HK_PASS_COUNT = 18
} HkPass;

/* Frame stages for band-sharded (multi-GPU) rendering: the host exchanges halo rows between
 * stages (hk_band_plan).  A single-GPU frame is the three stages back to back. */
typedef enum HkStage {
  HK_STAGE_TEMPORAL = 0,     /* prepass (+apron), albedo, direct_lit x2, indirect on the band */
  HK_STAGE_SPATIAL = 1,      /* spatial_reuse dispatches that are enabled */
  HK_STAGE_POST_PROCESS = 2, /* demodulation + a-trous x4 per channel, tone mapping */
  HK_STAGE_ANTIALIAS = 3,    /* the rest of PostProcessNode::run (post_process.rs:1236-1272): SMAA Tu4x (+extrapolate)
                                when upscale_kind is SMAA_TU4X, then TAA when taa is JASMINE, on the band (exchange D of
                                hk_band_plan_for: tone-mapped rows + last frame's TAA rows).  Not part of hk_frame_render
                                unless HK_FRAME_ANTIALIAS. */
  HK_STAGE_UPSCALE = 4,      /* upscale_kind FSR1 only (post_process.rs:1277-1308; a no-op for SMAA_TU4X): EASU + RCAS to the
                                window size.  A band owns the window rows hk_band_rows(height, ..) gives it: EASU runs on those
                                rows +-1 (RCAS's cross), RCAS on the rows themselves; exchange E of hk_band_plan_for brings the
                                3-4 rows of the EASU input (taa_output or tone-mapped) the 12 taps reach beyond the render band.
                                Runs after HK_STAGE_ANTIALIAS under HK_FRAME_ANTIALIAS. */
  HK_STAGE_COUNT = 5
} HkStage;

/* One halo transfer the host must perform BEFORE running `stage`: rows [row_begin,row_end) of
 * `buffer` (row = `row_bytes` bytes at device offset row*row_bytes) travel from the rank that owns
 * them to this rank.  peer = band index of the owner. */
typedef struct HkHaloOp {
  uint32_t buffer;     /* HkBuffer */
  uint32_t peer;       /* band index (= rank) that owns the rows */
  uint32_t row_begin;
  uint32_t row_end;
  uint64_t row_bytes;
} HkHaloOp;

#define HK_TIMING_SLOTS 24
/* slot 18 (beyond the HkPass ids): every TRACE launch of the queue-based indirect pass on its own (bounces + 1 launches per pass);
 * selected like a pass, by bit 18 of hk_set_timing_mask - and only so: HK_CTX_TIME_PASSES selects the HkPass slots 0 .. HK_PASS_COUNT - 1 */
#define HK_TIMING_TRACE_STAGES 18u
typedef struct HkStats {
  uint64_t rays_primary;      /* G-buffer rays */
  uint64_t rays_tlas;         /* traverse_top invocations (light.wgsl:442) */
  uint64_t rays_blas;         /* stand-alone traverse_bottom invocations (light.wgsl:687) */
  uint64_t frames;
  /* HIP-event timing on the context's stream.  Slot = HkPass id.  Only passes selected by
   * hk_set_timing_mask() (every HkPass, with HK_CTX_TIME_PASSES) are bracketed by events. */
  double pass_ms_total[HK_TIMING_SLOTS];
  uint64_t pass_launches[HK_TIMING_SLOTS];
  float last_frame_ms;        /* first dispatch of TEMPORAL .. last dispatch of POST_PROCESS */
  uint32_t _pad;
  /* how often the device scene was (re)built since hk_create: the mesh-level arrays (BLAS nodes,
   * triangles, vertices) and the instance-level arrays (TLAS, instances, lights, materials).  An
   * instance-only update must leave scene_mesh_builds unchanged. */
  uint64_t scene_mesh_builds;
  uint64_t scene_instance_builds;
  /* ... of which the instance-level arrays went through pinned staging into the spare slot in stream order, with no
   * host or device wait (scenes larger than the 32 KB LDS copy keep two slots of the instance-level region) */
  uint64_t scene_async_instance_uploads;
  /* instance updates that ran on the device (hk_refit_scene_instances): no host tree build, no scene buffer over PCIe */
  uint64_t scene_device_refits;
  uint64_t scene_device_tree_builds;  /* hk_rebuild_scene_trees */
  /* ABI 7, HK_CTX_COUNT_RAYS: what the counted walks did, in the terms of SURVEY 8d's algorithmic BVH bytes per ray - node steps
   * (32 B each: two 16-B loads), triangle tests (48 B), instance entries (a 208-B record; the survey's formula leaves them out),
   * closest hits whose attributes were fetched (three 32-B vertex records).  bench.py prices the trace kernels of the large-scene
   * configs with them (roofline of extra_configs 3 / 4). */
  uint64_t walk_node_steps;
  uint64_t walk_triangle_tests;
  uint64_t walk_instance_entries;
  uint64_t walk_closest_hits;
  uint64_t walk_top_node_steps;  /* ... of walk_node_steps, those taken in the instance tree (the rest walk mesh trees) */
  /* The wide walk (HK_TRAVERSAL_WIDE) keeps a lane's pending subtrees on a stack of 124 entries (28 in LDS, 96 behind them): enough
   * for trees 80 levels deep.  An entry that did not fit is DROPPED - geometry behind it is not seen by that ray - and counted here
   * (every context counts, not only HK_CTX_COUNT_RAYS): a host that loads degenerate trees checks this for 0 after its first
   * frames and creates the context with HK_CTX_NO_WIDE_WALK otherwise.  0 in every test and benchmark of this repository. */
  uint64_t wide_stack_lost;
} HkStats;

typedef struct hk_ctx hk_ctx;

/* ------------------------------------------------------------------ lifetime */
uint32_t hk_abi_version(void);
const char* hk_last_error(void); /* thread-local description of the last failure */
/* What this binary was built from (tools/build_lib.py): "sources <sha256[:16] of csrc/ + include/> | <hipcc, its version, every flag> |
 * built <UTC time>".  A host (and __graft_entry__.smoke()) can tell a stale library from the tree it sits in. */
const char* hk_build_info(void);
int hk_device_count(int* count);
/* Creates a context on HIP device `device_id`; flags: bit0 = count rays (HK_CTX_COUNT_RAYS),
 * bit1 = time every dispatch with HIP events (HK_CTX_TIME_PASSES), bit2 = take every (k + 0.5) / size through
 * the IEEE division sequence instead of the certified 3-instruction route (HK_CTX_PLAIN_DIVISION; the results
 * are identical bit for bit, the flag exists so that tests can show it). */
#define HK_CTX_COUNT_RAYS 1u
#define HK_CTX_TIME_PASSES 2u
#define HK_CTX_PLAIN_DIVISION 4u
/* By default hk_frame_stage / hk_frame_render run the two direct-light dispatches (sun, emissive: light.rs:656-688)
 * on a second HIP stream, concurrently with indirect_lit_ambient and its spatial pass - they touch disjoint
 * reservoir / render buffers (light.rs:518-546) - and join before anything reads their outputs (demodulation, halo
 * exchange B, hk_read_buffer, hk_frame_wait ...).  hk_pass_run never forks.  bit3 keeps everything on one stream:
 * same results, no overlap (used to time a kernel alone). */
#define HK_CTX_SINGLE_STREAM 8u
/* Verification mode.  The reference lets the stores to previous_spatial_reservoir_buffer race under camera / object
 * motion (light.wgsl:1063,1092-1095,1199-1202,1456-1459: a thread stores at the REPROJECTED pixel, which another thread
 * owns); by default so does this library - same kernels, whichever store arrives last stays.  bit4 parks those stores and
 * applies them after the dispatch so that the store of the highest thread index wins, which is how the CPU oracle
 * resolves the race: with it a moving camera and moving objects are bit-exact against the oracle too.  Costs three
 * extra launches and 72 B per pixel of scratch per light dispatch; not meant for production frames. */
#define HK_CTX_DETERMINISTIC_SCATTER 16u
/* Round 6.  The rule above costs little in its light form - a store to the pixel's own slot goes straight to the buffer, only a store
 * to another pixel's slot is parked and weighed against the slot's owner afterwards (one small launch per channel; the uniform-tile
 * store elision and the frame pipelining stay on) - so a single context now applies it BY DEFAULT to the channels whose
 * previous_spatial buffer has a reader (the indirect channel when indirect_spatial_reuse is on, sun + emissive when
 * emissive_spatial_reuse is on): what is rendered no longer depends on which store lands last, and equals the oracle's frames under
 * motion.  HK_CTX_DETERMINISTIC_SCATTER extends it to all three channels (every reservoir byte reproducible);
 * HK_CTX_RACING_SCATTER (bit10) restores the reference's own race - the A/B of bench.py --motion. */
#define HK_CTX_RACING_SCATTER 1024u
/* Traversal order.  The reference walks its flat BVHs in ONE fixed depth-first order (`bvh` 0.7.1 flatten_custom: left child
 * first, light.wgsl:400-486), which for a closest-hit ray coming "from the right" means visiting most of the tree before the
 * near hit that would have pruned it.  For scenes too large for the LDS copy the library therefore keeps EIGHT flattenings of
 * every TLAS / BLAS - one per sign pattern of the ray direction, children ordered the way such a ray meets them
 * (hk_bvh_rethread) - and each ray walks the one of its octant with the same stackless loop: same nodes, boxes, leaves and
 * per-candidate arithmetic, so the closest hit is the reference's except where two candidates tie exactly (which the order
 * breaks differently) - parity is then the north star's 1e-3 relative L2, not bit equality.  Occlusion (any-hit) results do
 * not depend on the order at all.  bit5 forces the reference's single order everywhere: the verification mode in which large
 * scenes are bit-exact against the oracle too.  Scenes that fit the LDS copy (Cornell) always use the reference order. */
#define HK_CTX_EXACT_TRAVERSAL 32u
/* Schedule of indirect_lit_ambient with two or more bounces (light.wgsl:1263-1498, the MULTIPLE_BOUNCES pipeline of
 * light.rs:663-666).  FUSED: one kernel, one pixel per lane from the G-buffer read to the reservoir store - the walks of a
 * wave last as long as its slowest ray.  WAVEFRONT: the same per-pixel arithmetic cut at the walks - a set-up dispatch gives
 * every non-background pixel a path slot, persistent trace waves pull rays from a queue (a lane whose ray has ended takes the
 * next one: wave ballot + prefix count, one atomic per 64 rays), a shade dispatch per bounce emits that bounce's shadow ray and
 * the next closest-hit ray into one queue and compacts the surviving paths; path state travels in 16-B planes indexed by slot
 * (~290 B per pixel of scratch, allocated on first use).  Both schedules produce the same bytes in every buffer.  Default:
 * wavefront for scenes beyond the LDS copy (long walks, where lane refill pays), fused for scenes that fit it (short walks,
 * where the ~0.4 KB per path and bounce of queue traffic costs more than the idle lanes).  bit6 / bit7 force one or the other. */
#define HK_CTX_WAVEFRONT 64u
#define HK_CTX_FUSED_INDIRECT 128u
/* Scenes beyond the LDS copy, product default (no HK_CTX_EXACT_TRAVERSAL): the CLOSEST-HIT walks - the primary rays of the prepass and
 * every ray of the wavefront schedule's trace stages - read 128-B records of an inner node's four grandchildren (derived on the
 * device from ordering 0 of the trees the scene holds) and take the children nearest first with a per-lane stack: two levels of the
 * tree per dependent fetch, on both levels of the scene.  Same candidates, same per-triangle arithmetic on the same operands, and of
 * two candidates at EXACTLY the same distance the one the reference's own walk meets first (decided from the leaves' positions in
 * the reference's flattening, kept next to the records): the closest hit is the reference's - up to box culls that depend on the
 * visit order (the reference tests a leaf's own box against the closest distance at the moment it VISITS the leaf, light.wgsl:412,
 * the wide walk when it processes the parent record: where a box is grazed within rounding one of them tests a triangle the other
 * skips; measured <= 1 primary hit per 8.3 M pixels and <= 6.3e-5 relative L2 over 32 frames against HK_CTX_EXACT_TRAVERSAL,
 * profiles/r05_default_mode_sequence_config4_4k.json) -, independent of the visit order, of
 * timing, and of how the trace stage splits a long walk among the idle lanes of its wave at the end of a stage; two runs of the
 * same frames are equal byte for byte.  Any-hit rays: whether a ray is occluded does not depend on the order; the rays whose
 * OCCLUDER is kept (the direct-light passes store its position in the reservoir) walk the reference's own order.  bit8 switches the
 * wide walk off (the closest-hit walks then take the direction-threaded skip-link walk, whose exact ties may fall differently): the
 * A/B the tests and `bench.py --no-wide-walk` use.  hk_traversal_mode reports HK_TRAVERSAL_WIDE when it is in use;
 * HkStats.wide_stack_lost counts pending subtrees a walk had to drop (0 for trees up to ~80 levels deep). */
#define HK_CTX_NO_WIDE_WALK 256u
/* Measurement (round 5): the frame takes exactly the schedule and walks it takes without the flag - unlike HK_CTX_COUNT_RAYS, whose
 * counting kernels exist in the fused form only and switch the queue-based schedule off - but the trace stages of the queue-based
 * indirect pass run the COUNTING twin of their kernel: per stage, records fetched (of them in the instance tree), triangle tests,
 * instance entries, rays, closest hits, pieces of long walks handed to idle lanes, and the moments the stage's queue ran dry and its
 * last wave left (hikari_hip_debug.h hk_debug_read_wf_timeline).  bench.py prices the trace kernel of configs 3 / 4 with these - the
 * walk that is TIMED, not a replay in another form. */
#define HK_CTX_COUNT_WALKS 512u
int hk_create(int device_id, uint32_t flags, hk_ctx** out);
void hk_destroy(hk_ctx* ctx);

/* ------------------------------------------------------------------ host-side mirrors of reference logic (no GPU needed) */
int hk_settings_default(HkSettings* out);                                        /* lib.rs:435-455 */
/* FrameUniform::extract_component, view.rs:141-193 (+ KERNEL / HALTON consts view.rs:125-139) */
int hk_frame_from_settings(const HkSettings* settings, uint32_t frame_number, HkFrame* out);
/* scaled render size = ceil(size / ratio), light.rs:318-319,623-624 */
int hk_scaled_size(uint32_t width, uint32_t height, float upscale_ratio, uint32_t* sw, uint32_t* sh);

/* Scene builder: the Prepare-stage host work of the reference, re-implemented in C++.
 *   add_mesh      -> TryFrom<Mesh> for GpuMesh, mod.rs:379-467 (triangle list / strip rules,
 *                    BLAS via `bvh` 0.7.1 BVH::build + flatten_custom(GpuNode::pack))
 *   add_material  -> material.rs:168-199 (values are passed through unchanged)
 *   add_instance  -> instance.rs:286-325 (world AABB from the 8 transformed half-extent corners)
 *   finish        -> mesh.rs:106-166 (concatenate + offsets), instance.rs:352-428 (TLAS, emissive
 *                    list, per-instance alias tables mod.rs:330-376, light BVH)
 * Arrays returned by the getters stay valid until the builder is destroyed. */
typedef struct hk_scene_builder hk_scene_builder;
#define HK_TOPOLOGY_TRIANGLE_LIST 0u
#define HK_TOPOLOGY_TRIANGLE_STRIP 1u
int hk_scene_builder_create(hk_scene_builder** out);
void hk_scene_builder_destroy(hk_scene_builder* b);
int hk_scene_builder_add_mesh(hk_scene_builder* b, const float* positions, const float* normals, const float* uvs,
                              uint32_t n_vertices, const uint32_t* indices, uint32_t n_indices, uint32_t topology,
                              uint32_t* mesh_id);
int hk_scene_builder_add_material(hk_scene_builder* b, const HkMaterial* material, uint32_t* material_id);
int hk_scene_builder_add_instance(hk_scene_builder* b, uint32_t mesh_id, uint32_t material_id,
                                  const float transform[16], uint32_t* instance_id);
int hk_scene_builder_finish(hk_scene_builder* b);
/* hk_scene_builder_finish WITHOUT its two `BVH::build` calls (instance.rs:365-371,422-428): every per-instance and per-emitter
 * record as above, the instance tree and the light tree as cheap valid stand-ins (index list halved recursively) of the final
 * size - for hosts that let the device build the trees (hk_update_scene_instances). */
int hk_scene_builder_finish_instances(hk_scene_builder* b);
/* Instance set edits (the reference's prepare_instances runs again on any of them, instance.rs:352-437).  Removing an instance
 * shifts the ids of the instances added after it down by one; its transform history goes with it. */
int hk_scene_builder_remove_instance(hk_scene_builder* b, uint32_t instance_id);
int hk_scene_builder_set_instance_material(hk_scene_builder* b, uint32_t instance_id, uint32_t material_id);
/* Dynamic scenes (instance.rs:352-437 re-runs whenever an instance changes): replace an instance's
 * transform after a finish; the next finish redoes only the instance-level work (world AABBs, TLAS,
 * emissive list, alias tables, light BVH) - meshes and their BLAS are kept.  The transform the
 * instance had at the previous finish becomes its "previous transform" (PreviousMeshUniform,
 * instance.rs:111-128), which the G-buffer's velocity output needs. */
int hk_scene_builder_set_instance_transform(hk_scene_builder* b, uint32_t instance_id, const float transform[16]);
/* n instances x 16 floats (column-major): each instance's transform at the finish before the last one */
int hk_scene_builder_previous_transforms(const hk_scene_builder* b, const float** p, uint32_t* n);
int hk_scene_builder_vertices(const hk_scene_builder* b, const HkVertex** p, uint32_t* n);
int hk_scene_builder_primitives(const hk_scene_builder* b, const HkPrimitive** p, uint32_t* n);
int hk_scene_builder_asset_nodes(const hk_scene_builder* b, const HkNode** p, uint32_t* n);
int hk_scene_builder_materials(const hk_scene_builder* b, const HkMaterial** p, uint32_t* n);
int hk_scene_builder_instances(const hk_scene_builder* b, const HkInstance** p, uint32_t* n);
int hk_scene_builder_instance_nodes(const hk_scene_builder* b, const HkNode** p, uint32_t* n);
int hk_scene_builder_emissives(const hk_scene_builder* b, const HkEmissive** p, uint32_t* n);
int hk_scene_builder_emissive_nodes(const hk_scene_builder* b, const HkNode** p, uint32_t* n);
int hk_scene_builder_alias_table(const hk_scene_builder* b, const HkAliasEntry** p, uint32_t* n);

/* The layout conversion behind HK_CTX_EXACT_TRAVERSAL's opposite (see there): `nodes[0..count)` is ONE flat BVH in the `bvh`
 * 0.7.1 flatten_custom layout the reference produces (mod.rs:185-201,458-459; entry / exit indices local to the array);
 * `out` receives the same tree flattened for ray-direction octant `octant` (bit k set = direction component k negative).
 * Pure host logic. */
int hk_bvh_rethread(const HkNode* nodes, uint32_t count, uint32_t octant, HkNode* out);

/* ------------------------------------------------------------------ uploads (Prepare stage) */
/* MeshRenderAssets::set + write_buffer, mesh.rs:43-64 (3 global buffers) */
int hk_upload_meshes(hk_ctx* ctx, const HkVertex* vertices, uint32_t n_vertices, const HkPrimitive* primitives,
                     uint32_t n_primitives, const HkNode* asset_nodes, uint32_t n_asset_nodes);
/* MaterialRenderAssets, material.rs:201-202.  The *_texture fields index the array given to
 * hk_upload_textures (HK_NO_TEXTURE = none), exactly like MaterialTextures::id (material.rs:76-86). */
int hk_upload_materials(hk_ctx* ctx, const HkMaterial* materials, uint32_t n_materials);
/* The `textures` / `samplers` binding arrays of group 3 (light.wgsl:15-18, mod.rs:760-782): one
 * RGBA8 image + its sampler per entry, sampled at LOD 0 (light.wgsl:749-793).  is_srgb = the image
 * format is Rgba8UnormSrgb (bevy loads base-colour / emissive images that way): rgb is decoded to
 * linear BEFORE filtering, alpha is linear.  n = 0 selects the NO_TEXTURE pipelines. */
typedef enum HkAddressMode { HK_ADDRESS_CLAMP_TO_EDGE = 0, HK_ADDRESS_REPEAT = 1, HK_ADDRESS_MIRROR_REPEAT = 2 } HkAddressMode;
typedef struct HkImageDesc {
  const uint8_t* rgba8; /* width * height * 4 bytes, row-major, row 0 = v = 0 */
  uint32_t width, height;
  uint32_t is_srgb;
  uint32_t address_u, address_v; /* HkAddressMode */
  uint32_t filter_linear;        /* 0 = nearest, 1 = bilinear (mag/min filter of the image's sampler) */
} HkImageDesc;
/* Instance motion on the DEVICE.  The reference re-runs prepare_instances on the CPU whenever an instance
 * moves (instance.rs:286-437): per-instance world AABB and inverse-transpose matrix, the emitter records that follow from them
 * (position, radius, surface area, alias table), a fresh `BVH::build` of the instance tree and of the light tree, and a re-upload
 * of all five buffers.  hk_upload_scene_instances is that path (host rebuild, asynchronous upload into the spare slot).  This
 * entry point instead diffs the builder's poses (hk_scene_builder_set_instance_transform) against the poses the device holds and
 * hands the moved instances - 96 B each, read by the kernel from pinned memory - to the GPU: one kernel redoes the per-instance and
 * per-emitter work with the host builder's exact arithmetic, two more REFIT the instance tree (in all of its direction-threaded
 * orderings) and the light tree: same topology, every inner box the union of the leaf boxes below it.  Stream-ordered, no host or
 * device wait; frames in flight keep the slot they were enqueued with.  What it does NOT do is change the shape of the trees: after
 * large displacements a host calls hk_rebuild_scene_trees (below) now and then to get the reference's SAH tree back (any-hit
 * identity and tie-breaks follow the tree, so a refit frame equals the reference frame for THAT tree, not for the rebuilt one).
 * Instances must be the ones uploaded (same count, meshes, materials); *moved (optional) = how many poses changed.  The builder's
 * previous-transform bookkeeping advances as it would in hk_scene_builder_finish.
 * After a device-side update the host copies of the trees and emitter records are stale: an upload that re-lays the instance-level
 * region out from them (hk_upload_materials, hk_upload_textures) is refused with HK_E_NOT_READY at the next frame until
 * hk_upload_scene_instances / hk_upload_instances brings the host's version of the scene back. */
int hk_refit_scene_instances(hk_ctx* ctx, hk_scene_builder* b, uint32_t* moved);
/* ... and the REBUILD on the device, for when refits have degraded a tree: new trees over the instances' and the emitters' current
 * boxes, written in place in the flatten_custom layout (all direction-threaded orderings of the instance tree; child order of
 * orderings 1-7 by the rule of hk_bvh_rethread).
 *   HK_TREE_SAH   the reference's OWN tree: `bvh` 0.7.1's binned-SAH build (BVH::build, instance.rs:365-371,422-428) level by level in
 *                 one workgroup - every reduction in it is a min, a max or a count and its re-ordering a stable sort by bucket, so
 *                 the parallel build makes the host's decisions and returns the host's tree shape (the tests compare the links).
 *                 A frame after it equals the frame after hk_upload_scene_instances.
 *   HK_TREE_LBVH  Morton codes of the box centres, one radix sort, Karras' parallel hierarchy: a few kernels whatever the depth of
 *                 the tree, but spatial-median splits - a worse tree (frames 17-23 % slower in the probe scenes); frames equal the
 *                 reference's up to exact ties between candidates, like any other valid tree over the same instances. */
#define HK_TREE_SAH 0u
#define HK_TREE_LBVH 1u
/* Instances ADDED, REMOVED or given another material (hk_scene_builder_add_instance / _remove_instance /
 * _set_instance_material, any number of pose changes with them): hk_scene_builder_finish_instances lays the instance-level
 * records out on the host (O(instances); no tree build), the asynchronous upload puts them into the spare slot and both trees
 * are built on the device in `tree_mode` - with HK_TREE_SAH the result is what hk_scene_builder_finish +
 * hk_upload_scene_instances gives (the reference's path), minus the two host-side SAH builds that dominate it. */
int hk_update_scene_instances(hk_ctx* ctx, hk_scene_builder* b, uint32_t tree_mode);
int hk_rebuild_scene_trees(hk_ctx* ctx, uint32_t mode);
int hk_upload_textures(hk_ctx* ctx, const HkImageDesc* images, uint32_t n_images);
/* InstanceRenderAssets::set + write_buffer, instance.rs:82-108 */
int hk_upload_instances(hk_ctx* ctx, const HkInstance* instances, uint32_t n_instances, const HkNode* instance_nodes,
                        uint32_t n_instance_nodes, const HkEmissive* emissives, uint32_t n_emissives,
                        const HkNode* emissive_nodes, uint32_t n_emissive_nodes, const HkAliasEntry* alias_table,
                        uint32_t n_alias);
/* PreviousMeshUniform::transform per instance (instance.rs:111-128; prepass.wgsl:50,96):
 * n_instances x 16 floats, column-major, the model matrices of the PREVIOUS frame.  Optional: without
 * it every instance is static (previous = current).  Applies to the instances of the last
 * hk_upload_instances call and is dropped by the next one, so a host with moving objects uploads both
 * every frame, as the reference extracts both every frame. */
int hk_upload_previous_transforms(hk_ctx* ctx, const float* models, uint32_t n_instances);
/* convenience: the three uploads above from a finished builder */
int hk_upload_scene(hk_ctx* ctx, const hk_scene_builder* b);
/* convenience: hk_upload_instances + hk_upload_previous_transforms from a re-finished builder whose
 * meshes and materials are unchanged.  Only the instance-level device arrays are rewritten. */
int hk_upload_scene_instances(hk_ctx* ctx, const hk_scene_builder* b);
/* NoiseTextures, lib.rs:189-219,515-598: 16 tiles of 64x64 RGBA8, tile-major */
int hk_upload_noise(hk_ctx* ctx, const uint8_t* rgba, size_t bytes);
/* prepare_light_textures / prepare_prepass_textures / post-process textures: (re)allocate all
 * screen-space resources and ZERO the 10 reservoir buffers (light.rs:307-383). */
int hk_resize(hk_ctx* ctx, uint32_t width, uint32_t height, float upscale_ratio);

/* ------------------------------------------------------------------ per-frame */
/* the four dynamic uniforms of bind group 0 (prepass.rs:546-553) */
/* TAA / upscale variant that selects the prepass sub-pixel jitter (prepass.rs:489-490,
 * prepass.wgsl:30-38), and Upscale::sharpness() (lib.rs:489-496) that HK_PASS_FSR_RCAS reads;
 * hk_frame_stage sets all three from HkSettings, hk_pass_run uses the last values. */
int hk_set_view_options(hk_ctx* ctx, uint32_t taa, uint32_t upscale_kind, float upscale_sharpness);
int hk_frame_begin(hk_ctx* ctx, const HkFrame* frame, const HkView* view, const HkPreviousView* previous_view,
                   const HkLights* lights);
/* One dispatch over rows [row_begin,row_end) of its grid (row_end = 0 means "all rows").  The
 * reservoir ping-pong (light.rs:376,480-481,518-546) follows frame.number of hk_frame_begin. */
int hk_pass_run(hk_ctx* ctx, uint32_t pass, uint32_t arg, uint32_t row_begin, uint32_t row_end);
/* The node order of the reference for the band of this context (whole image by default):
 *   TEMPORAL     = PrepassNode::run + LightNode::run up to and including the temporal dispatches
 *   SPATIAL      = the spatial_reuse dispatches of LightNode::run (light.rs:689-697)
 *   POST_PROCESS = PostProcessNode::run denoise loop + tone mapping (post_process.rs:1190-1234)
 * flags bit0: the G-buffer was supplied by the host (hk_write_buffer), skip the primary-ray prepass. */
#define HK_FRAME_EXTERNAL_GBUFFER 1u
/* flags bit1 (hk_frame_render): also run HK_STAGE_ANTIALIAS */
#define HK_FRAME_ANTIALIAS 2u
/* flags bit2 (hk_frame_render, hk_multi_frame_render): hk_balance_bands right after hk_frame_begin - split THIS frame's rows by
 * cost and keep the split from here on.  Rows that change owner find the new owner's (stale) reservoirs as history: use it on
 * the first frame, after a cut, or accept a few frames of re-convergence in the moved rows. */
#define HK_FRAME_BALANCE_BANDS 4u
/* flags bit3 (hk_frame_render with a communicator, hk_multi_frame_render): after the last stage, band 0 collects the other bands' rows of
 * the frame's final image (hk_final_buffer) - SURVEY 8e step 7, hk_comm_gather / hk_multi_gather.  With a communicator and without
 * HK_FRAME_ANTIALIAS the rows travel WHILE the next frame renders (ABI 7: every RCCL call runs on a stream of the communicator's own,
 * ordered against the context's stream by events; the tone-mapped image is double-buffered by frame parity): the image on band 0 is
 * complete once hk_frame_wait, any buffer access, or the hk_frame_begin of the frame after next has been passed. */
#define HK_FRAME_GATHER 8u
/* flags bit4 (hk_frame_render; round 6): take THIS frame's band time - HIP events on the context's stream around stage TEMPORAL and
 * around stage SPATIAL, i.e. the band's own work on the critical cycle of a sharded frame WITHOUT the waits for its neighbours' halos
 * in between (with a communicator every band's wall clock shows the slowest band's time; this does not) - for hk_band_time_ms. */
#define HK_FRAME_TIME_BAND 16u
int hk_frame_stage(hk_ctx* ctx, uint32_t stage, const HkSettings* settings, uint32_t flags);
/* hk_frame_begin + TEMPORAL, SPATIAL, POST_PROCESS (+ ANTIALIAS with HK_FRAME_ANTIALIAS): single GPU, no halo exchange */
int hk_frame_render(hk_ctx* ctx, const HkFrame* frame, const HkView* view, const HkPreviousView* previous_view,
                    const HkLights* lights, const HkSettings* settings, uint32_t flags);
int hk_frame_wait(hk_ctx* ctx);

/* ------------------------------------------------------------------ band sharding (multi-GPU) */
/* Restrict this context to band `band_index` of `band_count` horizontal bands of the render image
 * (bands are contiguous row ranges, remainder rows spread over the first bands). */
int hk_set_band(hk_ctx* ctx, uint32_t band_index, uint32_t band_count);
int hk_band_rows(uint32_t height, uint32_t band_index, uint32_t band_count, uint32_t* row_begin, uint32_t* row_end);
/* Bands of unequal height (round 3).  Equal row counts are equal WORK only when the rows are alike: a sky band of the city-class
 * 4K frame takes 0.3 ms, a band full of buildings 6.3 ms.  bounds[0] = 0 < bounds[1] < ... < bounds[band_count] = the scaled render
 * height: band i renders rows [bounds[i], bounds[i + 1]).  Every rank of a sharded frame must set the SAME boundaries (the halo
 * schedules are derived from them on each rank); NULL / 0 returns to the equal split; hk_set_band with another band count and
 * hk_resize drop them.
 *   hk_row_costs           geometry pixels (depth >= epsilon) per full-size row of the frame most recently begun.  A rank that
 *                          ray-casts the WHOLE frame's primary rays once (hk_frame_begin + hk_pass_run(HK_PASS_PREPASS, 0, 0, 0))
 *                          holds what every other rank holds, bit for bit, so all ranks derive the same split without talking.
 *   hk_balanced_band_bounds  pure host logic: boundaries that give every band about the same cost, cost(row) = its geometry
 *                          pixels + width x background_cost, at least min_rows rows per band; row_cost has cost_rows entries
 *                          (hk_row_costs: the full-size height).  hk_balance_bands uses 1/4 for scenes walked from LDS (cheap
 *                          geometry pixels: the Cornell box) and 1/16 beyond (profiles/r03_band_balance_probe.json).
 *   hk_balance_bands       the three steps in one call, after hk_frame_begin: full-frame primary rays, count, split, set on this
 *                          context (min_rows 0 = 8, never more than rows / bands); bounds_out (optional) receives the split.  Deterministic across ranks. */
int hk_set_band_bounds(hk_ctx* ctx, const uint32_t* bounds, uint32_t n_bounds /* band_count + 1 */);
int hk_get_band(hk_ctx* ctx, uint32_t* band_index, uint32_t* band_count); /* what hk_set_band set (either pointer may be NULL) */
int hk_get_band_bounds(hk_ctx* ctx, uint32_t* bounds, uint32_t n_bounds /* band_count + 1 */); /* the split in force (explicit or equal) */
int hk_balance_bands(hk_ctx* ctx, uint32_t min_rows, uint32_t* bounds_out, uint32_t n_bounds /* band_count + 1, or 0 with NULL */);
int hk_row_costs(hk_ctx* ctx, uint32_t* geometry_pixels_per_row, uint32_t n_rows /* = the full-size height */);
int hk_balanced_band_bounds(const uint32_t* row_cost, uint32_t cost_rows, uint32_t width, uint32_t render_rows, uint32_t band_count,
                            uint32_t min_rows, float background_cost /* of a background pixel, geometry pixel = 1; <= 0: 1/16 */, uint32_t* bounds);
/* Bands of equal MEASURED time (round 6).  The split by geometry pixels prices every geometry pixel alike; the rows of a city-class
 * frame differ three times in walk length, and part of a band's time is fixed cost.  So: every M frames each rank takes its band's
 * GPU time (hk_frame_render(.., HK_FRAME_TIME_BAND), hk_band_time_ms), the ranks all-gather these N floats over the host's channel,
 * and every rank evaluates the same pure function on them:
 *   hk_rebalanced_band_bounds  bounds[band_count + 1] in force + band_ms[band_count] -> out[band_count + 1]: the boundaries moved
 *                          `damping` (0..1] of the way towards equal times (a band's time spread over its rows evenly, or by the
 *                          optional prior row_weight[render_rows], e.g. geometry pixels per row), at most max_shift rows per call
 *                          (0: no limit), at least min_rows per band.  A band without a usable time keeps the split as it is.
 *   hk_band_migration_plan / _schedule  what must travel when the split changes between two frames: the rows of the history
 *                          reservoirs (the three temporal outputs + the spatial outputs whose pass is on: what frame
 *                          `next_frame_number` reads as history) that change owner, from the band that owned them under old_bounds to
 *                          the band that owns them under new_bounds (NULL = the equal split) - the same row plan / one global order
 *                          of sends and receives as hk_band_plan / hk_band_schedule.  After it every band holds, for the rows it now
 *                          owns, exactly what their previous owner held: the sharded frames that follow equal the single context bit
 *                          for bit as before.  (The anti-aliasing tail keeps more per-pixel state - previous tone-mapped / TAA /
 *                          G-buffer planes: with it keep the split, or treat a re-split as a cut.)
 *   hk_migrate_bands       with a communicator: the migration as one RCCL exchange on the communicator's stream, ordered behind
 *                          everything the context has enqueued, then hk_set_band_bounds(new_bounds).  Every rank calls it with
 *                          the same arguments, between two frames.  (A host with its own transport moves hk_band_migration_schedule's
 *                          rows and calls hk_set_band_bounds itself; hk_multi_migrate_bands is the one-process form.) */
int hk_rebalanced_band_bounds(const uint32_t* bounds, const float* band_ms, uint32_t band_count, uint32_t render_rows, const float* row_weight /* or NULL */,
                              uint32_t min_rows, uint32_t max_shift, float damping, uint32_t* out);
int hk_band_migration_plan(uint32_t width, uint32_t height, float upscale_ratio, const uint32_t* old_bounds, const uint32_t* new_bounds, uint32_t band_index,
                           uint32_t band_count, uint32_t next_frame_number, const HkSettings* settings, HkHaloOp* ops, uint32_t* n_ops);
int hk_migrate_bands(hk_ctx* ctx, const uint32_t* new_bounds, uint32_t n_bounds /* band_count + 1 */, uint32_t next_frame_number, const HkSettings* settings);
/* the band time of the last frame rendered with HK_FRAME_TIME_BAND, in ms (waits for that frame's spatial stage; HK_E_NOT_READY if none) */
int hk_band_time_ms(hk_ctx* ctx, float* ms);
/* Halo transfers that must complete before `stage` runs on this context.  Pure host logic.
 * ops may be NULL to query the count.
 * Moving cameras / objects: the temporal and spatial dispatches read LAST frame's reservoirs at reprojected
 * positions (light.wgsl:1091,1144,1404,1525), which can lie up to |velocity| rows inside a neighbouring band.
 * Passing HK_STAGE_TEMPORAL_WITH_HISTORY(rows) as the stage returns "exchange C": `rows` rows of the
 * reservoir buffers frame n reads as history (the temporal outputs of all three channels and the spatial
 * outputs that are enabled), to be received before stage TEMPORAL of frame n.  A reprojection that lands
 * beyond the halo reads the local, stale rows - the documented deviation; rows = 0 (static camera) is
 * plain HK_STAGE_TEMPORAL and has no transfers. */
#define HK_STAGE_TEMPORAL_WITH_HISTORY(rows) ((uint32_t)HK_STAGE_TEMPORAL | ((uint32_t)(rows) << 8))
/* ... and the scatter stores of frame n's temporal dispatches that cross a band border (ABI 7).  A pixel of
 * band B whose history reservoir was rejected stores it at the reprojected slot of previous_spatial, up to `rows` rows away - in
 * the rows of a neighbour, whose spatial pass reads that slot - and B's own spatial pass reads slots up to `rows` rows outside B
 * that pixels of OTHER bands store to.  With a history halo the temporal dispatches of a band therefore PARK their stores
 * (HK_BUF_PARKED_TO0 / _RECORD0 + channel) and HK_STAGE_SPATIAL_WITH_HISTORY(rows) adds to exchange A, for every channel whose
 * spatial pass is enabled (sun and emissive share their spatial buffers: both, in dispatch order), 2 x rows rows of the parked
 * planes per side: every store that can reach a slot the band reads comes from a pixel at most 2 x rows rows outside it.  Stage
 * SPATIAL then resolves them - the highest storing pixel index wins, the rule HK_CTX_DETERMINISTIC_SCATTER and the oracle apply
 * to the reference's write-write race - before spatial_reuse runs: the union of the bands equals the single context bit for bit
 * under camera and object motion. */
#define HK_STAGE_SPATIAL_WITH_HISTORY(rows) ((uint32_t)HK_STAGE_SPATIAL | ((uint32_t)(rows) << 8))
/* How many history rows does this frame need?  Pure host logic, the same number on every rank (all of them hold the same
 * uniforms and the same scene).  Returns in *rows an upper bound of |row of the reprojected pixel - row of the pixel| over every
 * surface point the frame can show: the reprojection is previous_uv = uv - velocity with velocity = clip_to_uv(view_proj x p) -
 * clip_to_uv(previous_view_proj x p_previous) (prepass.wgsl:94-95), i.e. in NDC a projective map M = previous_view_proj x T x
 * inverse_view_proj of the pixel's own (x, y, depth), T = identity for static geometry and previous_model x model^-1 for an
 * instance that moved.  The bound is taken over the NDC box of the scene's world bounds (static part) and of every moved box with
 * interval arithmetic on y x (M v).w - (M v).y in centred form over a grid of cells, divided by the smallest (M v).w of the cell;
 * cells whose reprojection is certainly off screen do not count (the temporal kernels neither load nor store there).  + 2 rows
 * for the sub-pixel jitter of the deferred texel and the truncation to a row index, rounded up to a multiple of 4 (schedules are
 * cached per row count).  0 when the view did not change and nothing moved.  A reprojection that cannot be bounded (the previous
 * camera's plane cuts through the visible volume) gives render_rows: every band then needs all of last frame's rows. */
typedef struct HkMovedBox {
  float min[3];
  float _pad0;
  float max[3];
  float _pad1;
  float previous_from_current[16]; /* previous_model x model^-1, column-major: where a point of the box was last frame */
} HkMovedBox;
int hk_history_rows_bound(const HkView* view, const HkPreviousView* previous_view, uint32_t render_rows, const float scene_min[3],
                          const float scene_max[3], const HkMovedBox* moved, uint32_t n_moved, uint32_t* rows);
int hk_band_plan(hk_ctx* ctx, uint32_t stage, const HkSettings* settings, HkHaloOp* ops, uint32_t* n_ops);
/* Same plan without a context (used by hosts that only schedule): */
int hk_band_plan_for(uint32_t width, uint32_t height, float upscale_ratio, uint32_t band_index, uint32_t band_count,
                     uint32_t stage, uint32_t frame_number, const HkSettings* settings, HkHaloOp* ops, uint32_t* n_ops);
/* ... with an explicit split (bounds: band_count + 1 scaled render rows, see hk_set_band_bounds; NULL = hk_band_plan_for) */
int hk_band_plan_bounds(uint32_t width, uint32_t height, float upscale_ratio, const uint32_t* bounds, uint32_t band_index,
                        uint32_t band_count, uint32_t stage, uint32_t frame_number, const HkSettings* settings, HkHaloOp* ops,
                        uint32_t* n_ops);

/* One side of a halo transfer, as the executor needs it: a byte range of a buffer (the same range on both sides - buffers are
 * allocated full-frame on every rank, so a halo row lands at the address it has on its owner) moving between this rank and
 * `peer`. */
typedef struct HkTransfer {
  uint32_t buffer;   /* HkBuffer */
  uint32_t peer;     /* rank (= band index) on the other side */
  uint32_t is_recv;  /* 1: this rank receives rows `peer` owns; 0: this rank sends rows it owns */
  uint32_t _pad;
  uint64_t offset;   /* bytes from the start of the buffer */
  uint64_t bytes;
} HkTransfer;
/* Every transfer rank `rank` of `n_ranks` takes part in before `stage` (hk_band_plan_for's stage argument, including
 * HK_STAGE_TEMPORAL_WITH_HISTORY), sends AND receives, in ONE global order all ranks derive identically (the receive plans of
 * rank 0, 1, ... in turn) - the order a transport that pairs sends with receives by issue order (RCCL) needs.  Pure host
 * logic; out may be NULL to query the count. */
int hk_band_schedule(uint32_t width, uint32_t height, float upscale_ratio, uint32_t rank, uint32_t n_ranks, uint32_t stage,
                     uint32_t frame_number, const HkSettings* settings, HkTransfer* out, uint32_t* n_out);
int hk_band_schedule_bounds(uint32_t width, uint32_t height, float upscale_ratio, const uint32_t* bounds, uint32_t rank,
                            uint32_t n_ranks, uint32_t stage, uint32_t frame_number, const HkSettings* settings, HkTransfer* out,
                            uint32_t* n_out);
/* the schedule of hk_band_migration_plan (above: "Bands of equal MEASURED time") */
int hk_band_migration_schedule(uint32_t width, uint32_t height, float upscale_ratio, const uint32_t* old_bounds, const uint32_t* new_bounds, uint32_t rank,
                               uint32_t n_ranks, uint32_t next_frame_number, const HkSettings* settings, HkTransfer* out, uint32_t* n_out);
/* SURVEY 8e step 7, the gather of a finished image: the transfers of `rank` when band `root` collects every band's rows of
 * `buffer` (the root: one receive per other band, in band order; band r: one send).  Which rows of a buffer a band owns follows
 * the buffer's kind: render rows, two rows per render row for the SMAA Tu4x outputs, window rows for the FSR1 outputs.
 * upscale_kind: HkUpscaleKind of the settings the frame was rendered with; bounds: NULL = the equal split. */
int hk_band_gather_schedule(uint32_t width, uint32_t height, float upscale_ratio, uint32_t upscale_kind, const uint32_t* bounds,
                            uint32_t rank, uint32_t n_ranks, uint32_t root, uint32_t buffer, HkTransfer* out, uint32_t* n_out);
/* the image the overlay presents (overlay.rs:226-231) after a frame rendered with these settings and frame flags: the tone-mapped
 * image, or with HK_FRAME_ANTIALIAS the TAA / SMAA Tu4x / sharpened FSR1 output */
uint32_t hk_final_buffer(const HkSettings* settings, uint32_t frame_flags);

/* ------------------------------------------------------------------ halo exchange inside the boundary: one process per GPU, RCCL over xGMI */
/* The reference's LightNode / PostProcessNode record every dispatch of a frame into one command encoder (light.rs:590-702);
 * a sharded frame needs a neighbour exchange between the temporal and the spatial dispatches (light.rs:689-697) and before
 * the denoiser.  With a communicator attached, hk_frame_render performs those exchanges itself, on the context's stream:
 *   [exchange C, history rows]  TEMPORAL  exchange A  SPATIAL  exchange B  POST_PROCESS  [exchange D  ANTIALIAS  exchange E  UPSCALE]
 * as ncclSend / ncclRecv pairs inside one ncclGroupStart / ncclGroupEnd per exchange (librccl is loaded on first use; a
 * single-GPU host never touches it).  Rank 0 obtains the id, ships its HK_COMM_ID_BYTES bytes to the other ranks by any host
 * channel (a Bevy app: its own launcher; bench.py: the torch.distributed store), every rank calls hk_comm_init - which also
 * does hk_set_band(rank, n_ranks) - and then simply renders frames. */
#define HK_COMM_ID_BYTES 128
int hk_comm_unique_id(uint8_t id[HK_COMM_ID_BYTES]);
/* Pre-flight for the rendezvous: HK_OK iff librccl loads with every entry point the exchange uses and the context's device
 * can be made current.  ncclCommInitRank blocks until all n_ranks have called it, so a launcher first lets every rank
 * report this over its host channel and only calls hk_comm_init when ALL ranks passed - a rank that cannot bring RCCL up
 * then fails the job instead of leaving the healthy ranks waiting inside the rendezvous. */
int hk_comm_available(hk_ctx* ctx);
int hk_comm_init(hk_ctx* ctx, uint32_t rank, uint32_t n_ranks, const uint8_t id[HK_COMM_ID_BYTES]);
int hk_comm_destroy(hk_ctx* ctx);
/* rows of last frame's reservoirs / AA history fetched from the neighbours before TEMPORAL / ANTIALIAS (exchange C).  Since ABI 7
 * the default is HK_HISTORY_AUTO: hk_frame_begin derives the count from the frame's own uniforms, the scene's bounds and the
 * instances that moved (hk_history_rows_bound; 0 for a static view, so a static frame exchanges nothing).  An explicit count
 * overrides it (tests; hosts that know better).  hk_history_rows returns the count in force for the frame most recently begun.
 * hk_comm_set_history_rows is the ABI 6 name of hk_set_history_rows. */
#define HK_HISTORY_AUTO 0xFFFFu
int hk_set_history_rows(hk_ctx* ctx, uint32_t rows);
int hk_history_rows(hk_ctx* ctx, uint32_t* rows);
int hk_comm_set_history_rows(hk_ctx* ctx, uint32_t rows);
/* world bounds of the uploaded scene (the union of the instances' boxes; what hk_history_rows_bound takes) */
int hk_scene_bounds(hk_ctx* ctx, float min[3], float max[3]);
/* one exchange by hand (hosts that drive hk_frame_stage themselves): everything hk_band_schedule lists for `stage` */
int hk_comm_exchange(hk_ctx* ctx, uint32_t stage, const HkSettings* settings);
/* rank `root` collects every other rank's rows of `buffer` (hk_band_gather_schedule as ncclSend / ncclRecv in one group, on the
 * context's stream); hk_frame_render(.., HK_FRAME_GATHER) does it for hk_final_buffer with root 0 */
int hk_comm_gather(hk_ctx* ctx, uint32_t buffer, uint32_t root);

/* ------------------------------------------------------------------ one process, several GPUs (hk_create(n_gpus, device_ids)) */
/* Bevy renders from ONE process and one render thread: hk_multi is the same band-sharded frame driven by a single host
 * thread - n contexts, one per device, scene replicated, each rendering its band; halo rows move with hipMemcpyPeerAsync on
 * the receiver's stream, ordered against the owner's stream with events (no host synchronisation inside a frame).
 * device_ids may repeat (several bands on one GPU: how the path is tested where only one GPU exists). */
typedef struct hk_multi hk_multi;
int hk_multi_create(uint32_t n, const int* device_ids, uint32_t flags, hk_multi** out);
void hk_multi_destroy(hk_multi* m);
int hk_multi_context(hk_multi* m, uint32_t i, hk_ctx** out); /* the i-th band's context (borrowed) */
int hk_multi_upload_scene(hk_multi* m, const hk_scene_builder* b);
int hk_multi_upload_scene_instances(hk_multi* m, const hk_scene_builder* b);
int hk_multi_refit_scene_instances(hk_multi* m, hk_scene_builder* b, uint32_t* moved); /* hk_refit_scene_instances on every band's replica */
int hk_multi_rebuild_scene_trees(hk_multi* m, uint32_t mode);
int hk_multi_set_band_bounds(hk_multi* m, const uint32_t* bounds, uint32_t n_bounds);
/* hk_migrate_bands for the one-process form: the rows that change owner travel as peer copies, then every band takes the new split */
int hk_multi_migrate_bands(hk_multi* m, const uint32_t* new_bounds, uint32_t n_bounds, uint32_t next_frame_number, const HkSettings* settings);
/* the root band's context collects every other band's rows of `buffer` on its own device (peer copies ordered by events, no host wait) */
int hk_multi_gather(hk_multi* m, uint32_t buffer, uint32_t root);  /* hk_set_band_bounds on every band's context */
int hk_multi_update_scene_instances(hk_multi* m, hk_scene_builder* b, uint32_t tree_mode);  /* hk_update_scene_instances on every band's replica */
int hk_multi_upload_textures(hk_multi* m, const HkImageDesc* images, uint32_t n_images);
int hk_multi_upload_noise(hk_multi* m, const uint8_t* rgba, size_t bytes);
int hk_multi_resize(hk_multi* m, uint32_t width, uint32_t height, float upscale_ratio);
int hk_multi_set_history_rows(hk_multi* m, uint32_t rows);
int hk_multi_frame_render(hk_multi* m, const HkFrame* frame, const HkView* view, const HkPreviousView* previous_view,
                          const HkLights* lights, const HkSettings* settings, uint32_t flags);
int hk_multi_wait(hk_multi* m);
/* the union of the bands: every band's own rows of `buffer` (a buffer whose rows are render rows, or full-size rows at
 * upscale ratio 1) gathered into one host image of the buffer's full size */
int hk_multi_read_buffer(hk_multi* m, uint32_t buffer, void* dst, size_t bytes);

/* ------------------------------------------------------------------ buffer access */
int hk_buffer_info(hk_ctx* ctx, uint32_t buffer, uint32_t* width, uint32_t* height, uint32_t* bytes_per_pixel);
/* synchronous copies (wait for the stream first) */
int hk_read_buffer(hk_ctx* ctx, uint32_t buffer, void* dst, size_t bytes);
int hk_write_buffer(hk_ctx* ctx, uint32_t buffer, const void* src, size_t bytes);
/* raw device pointer for zero-copy views (a host that composites or exchanges the buffers itself).  *bytes = the size of the
 * ALLOCATION, which does not depend on the upscale kind in effect (hk_buffer_info gives the logical size); the pointer is
 * valid until the next hk_resize (and, for the double-buffered ids, names another plane after the next hk_frame_begin). */
int hk_device_ptr(hk_ctx* ctx, uint32_t buffer, void** ptr, size_t* bytes);
/* the HIP stream the context enqueues its frames on.  Without hk_set_stream it is the context's own; the context's FIRST frame may
 * create it again at the device's highest priority (round 6: the frame's dependent chain ahead of the direct-light and a-trous
 * dispatches of the other streams when frames are small) - so ask after that frame: a handle taken before it may name a stream that
 * no longer exists. */
int hk_stream(hk_ctx* ctx, void** hip_stream);
/* Enqueue all subsequent work on a HIP stream owned by the host (e.g. the stream its RCCL calls
 * are ordered against) instead of the context's own stream; NULL restores the context's stream. */
int hk_set_stream(hk_ctx* ctx, void* hip_stream);
/* bit i set = bracket every dispatch of HkPass i with HIP events (see HkStats) */
int hk_set_timing_mask(hk_ctx* ctx, uint32_t pass_mask);
int hk_get_stats(hk_ctx* ctx, HkStats* out);
int hk_reset_stats(hk_ctx* ctx);
/* Which schedule the indirect_lit_ambient dispatch of the frame most recently begun takes (see HK_CTX_WAVEFRONT): 0 = fused
 * kernel, 1 = wavefront (ray queues).  Depends on the flags, the frame's indirect_bounces and the size of the uploaded scene. */
int hk_indirect_schedule(hk_ctx* ctx, uint32_t* out);
/* How rays walk the scene currently uploaded (light.wgsl:400-486 is the reference walk):
 *   HK_TRAVERSAL_REFERENCE  the reference's two-level walk in the reference's node order - bit-exact; always with
 *                           HK_CTX_EXACT_TRAVERSAL, and for LDS-resident scenes whose instances do not share one transform
 *   HK_TRAVERSAL_THREADED   the two-level walk over eight direction-ordered flattenings (scenes beyond the LDS copy)
 *   HK_TRAVERSAL_ONE_LEVEL  ONE BVH over all triangles in the instances' shared local space (LDS-resident scenes whose
 *                           instances all have the same transform, e.g. the Cornell box): the reference's per-triangle
 *                           arithmetic on the reference's operands, so the closest hit is the reference's except where a
 *                           ray grazes an instance's world box within rounding or two candidates tie exactly.
 * orderings (may be NULL): how many direction orderings of the trees are stored (1, 2, 4 or 8).
 * HK_TRAVERSAL_WIDE is OR-ed into HK_TRAVERSAL_THREADED when the closest-hit walks take the wide records (HK_CTX_NO_WIDE_WALK). */
#define HK_TRAVERSAL_REFERENCE 0u
#define HK_TRAVERSAL_THREADED 1u
#define HK_TRAVERSAL_ONE_LEVEL 2u
#define HK_TRAVERSAL_WIDE 0x100u
int hk_traversal_mode(hk_ctx* ctx, uint32_t* out, uint32_t* orderings);

/* Test and measurement hooks (hk_debug_*, hk_measure_*) are declared in hikari_hip_debug.h: a host that renders binds none of them. */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HIKARI_HIP_H */
