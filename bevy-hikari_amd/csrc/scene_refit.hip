// scene_refit.hip - instance motion and instance-set changes on the device (kernels in kernels_scene.hip): the
// reference re-runs prepare_instances on the host for ANY instance change (instance.rs:352-437); here moved instances are refit,
// both trees rebuilt (the reference's SAH tree or an LBVH) and instance-set edits laid out without the host's two tree builds.
#include "hk_context.hpp"

using namespace hk;
using namespace hkd;

extern "C" {

// Instances added, removed or re-materialed (the reference re-runs prepare_instances for ANY instance change, instance.rs:352-437):
// the per-instance / per-emitter records are laid out on the host - O(instances), no tree build - and go to the spare slot through
// the asynchronous upload; both trees are then built on the device (hk_rebuild_scene_trees: HK_TREE_SAH = the reference's own
// tree, link for link).  What the host no longer does is the two `BVH::build` calls: 1.1 ms of 1.7 ms at 2 000 instances, 30 of
// 32 ms at 20 000.
int hk_update_scene_instances(hk_ctx* c, hk_scene_builder* b, uint32_t tree_mode) {
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
  HK_REQUIRE(c && b, HK_E_INVALID, "NULL argument");
  HK_REQUIRE(tree_mode == HK_TREE_SAH || tree_mode == HK_TREE_LBVH, HK_E_INVALID, "unknown tree build mode %u", tree_mode);
  int rc;
  const bool trace = c->trace_update;
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
}  // extern "C"
namespace hk {
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
}  // namespace hk
extern "C" {

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
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
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
  c->wide_tlas_dirty = true;
  c->device_tree_builds += 1;
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
  c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
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
  c->wide_tlas_dirty = true;
  c->device_refits += 1;
  if (commit) builder_commit_transforms(b);
  return HK_OK;
}
int hk_refit_scene_instances(hk_ctx* c, hk_scene_builder* b, uint32_t* moved_out) { return refit_impl(c, b, moved_out, true); }}  // extern "C"
int hk::refit_instances_impl(hk_ctx* c, hk_scene_builder* b, uint32_t* moved, bool commit) { return refit_impl(c, b, moved, commit); }
