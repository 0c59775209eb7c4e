// hk_internal.hpp - shared host-side declarations of libhikari_hip.so (not part of the ABI).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/hikari_hip.h"
#include "../../include/hikari_hip_debug.h"

namespace hk {

void set_error(const char* fmt, ...);

#define HK_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      ::hk::set_error(__VA_ARGS__);  \
      return (code);                 \
    }                                \
  } while (0)

// flat skip-link BVH over n boxes (min xyz, max xyz) in the `bvh` 0.7.1 flatten_custom format
std::vector<HkNode> build_flat_bvh(const std::vector<float>& boxes_min_max);

// the same tree flattened for ray-direction octant `oct` (bit k set: direction component k negative); false if `nodes` is not a
// well-formed flatten_custom array.  Indices are local to the array.
bool rethread_flat_bvh(const HkNode* nodes, uint32_t count, uint32_t oct, HkNode* out);

// ---- what the device refit (context.hip hk_refit_scene_instances) needs from a scene builder (private to scene_builder.cpp)
struct InstanceDecl {
  HkMeshIndex mesh;
  uint32_t material;
  const float* transform;    // 16 floats, column-major: the pose set by add_instance / set_instance_transform
  const float* aabb_center;  // the mesh's local box (bevy Aabb): 3 + 3 floats
  const float* aabb_half;
};
uint32_t builder_instance_count(const hk_scene_builder* b);
// the builder was finished by hk_scene_builder_finish_instances: its two trees are valid STAND-INS (index list halved recursively)
// for a device-side build, not the reference's SAH trees - only hk_update_scene_instances may upload them
bool builder_has_standin_trees(const hk_scene_builder* b);
int upload_scene_instances_unchecked(hk_ctx* c, const hk_scene_builder* b);  // hk_upload_scene_instances without that check (context.hip)
bool builder_instance_decl(const hk_scene_builder* b, uint32_t i, InstanceDecl* out);
// what hk_scene_builder_finish does to the PreviousMeshUniform bookkeeping, without building anything
void builder_commit_transforms(hk_scene_builder* b);
// instance.rs:286-325 for one instance: world AABB (8 corners of the local box, seeded at zero) and inverse().transpose();
// false for a singular transform
bool instance_world_record(const float transform[16], const float aabb_center[3], const float aabb_half[3], float mn[3], float mx[3], float inverse_transpose_model[16]);

int refit_instances_impl(hk_ctx* c, hk_scene_builder* b, uint32_t* moved, bool commit);  // hk_refit_scene_instances

// bytes per pixel / full-size flag of an HkBuffer id (0 = invalid id)
uint32_t buffer_bpp(uint32_t buffer);
bool buffer_is_full_size(uint32_t buffer);
bool buffer_is_upscaled(uint32_t buffer);  // allocated at the SMAA Tu4x output size ceil(size * 2 / ratio)

// rows [b0,b1) of band i of n over `height` rows
void band_rows(uint32_t height, uint32_t band_index, uint32_t band_count, uint32_t* b0, uint32_t* b1);
// ... when the split of the `render_rows` scaled render rows is explicit (bounds[0..n], NULL = equal split); `rows` = the height of
// the plane being cut (render_rows, or the window rows FSR1 writes)
void band_rows_in(const uint32_t* bounds, uint32_t render_rows, uint32_t rows, uint32_t band_index, uint32_t band_count, uint32_t* b0, uint32_t* b1);
int band_buffer_rows(uint32_t width, uint32_t height, float ratio, uint32_t upscale_kind, const uint32_t* bounds, uint32_t buffer, uint32_t i, uint32_t n,
                     uint32_t* y0, uint32_t* y1, uint64_t* row_bytes);  // rows of `buffer` band i owns (host_logic.cpp)
bool band_bounds_valid(const uint32_t* bounds, uint32_t band_count, uint32_t render_rows);

// apron rows (in scaled render rows) each stage needs around a band, from the kernel footprints
struct Aprons {
  uint32_t spatial;   // rows of temporal reservoirs + G-buffer the spatial stage reads beyond the band
  uint32_t denoise;   // rows of render/variance the post-process stage reads beyond the band
};
Aprons band_aprons(const HkSettings* settings);


// ---- what comm.cpp needs from a context (hk_ctx is private to context.hip)
struct CtxInfo {
  int device;
  void* stream;            // hipStream_t the context enqueues on
  uint32_t width, height;  // window size
  float ratio;
  uint32_t frame_number, band_index, band_count, upscale_kind, taa;
  const uint32_t* band_bounds;   // explicit split of the scaled render rows (hk_set_band_bounds; band_count + 1 entries) or NULL
  uint32_t bounds_generation;    // changes whenever the split does (cached schedules compare it)
};
int ctx_info(hk_ctx* c, CtxInfo* out);
int comm_gather(hk_ctx* c, uint32_t buffer, uint32_t root, bool overlap);  // comm.cpp; overlap: see Comm
int comm_join(hk_ctx* c, int parity);  // the context's stream waits for a gather in flight (parity < 0: any)
void* ctx_buffer(hk_ctx* c, uint32_t buffer, size_t* logical_bytes);
void** ctx_comm_slot(hk_ctx* c);       // owned by comm.cpp (NULL = no communicator)
int ctx_join_side(hk_ctx* c);          // main stream waits for the side stream
// run before `stage` by hk_frame_render when a communicator is attached
int comm_exchange(hk_ctx* c, uint32_t stage_arg, const HkSettings* st);
int comm_migrate(hk_ctx* c, const uint32_t* old_bounds, const uint32_t* new_bounds, uint32_t next_frame_number, const HkSettings* st);  // hk_migrate_bands
void comm_release(hk_ctx* c);          // hk_destroy

}  // namespace hk
