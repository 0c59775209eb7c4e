// kernels_scene.hip - instance motion on the device (reference: prepare_instances, instance.rs:286-437).
//
// When an instance moves the reference recomputes, on the CPU, its world AABB and inverse-transpose matrix
// (instance.rs:286-325), the emitter record that follows from them (position, radius, surface area, alias table:
// instance.rs:380-420) and then REBUILDS both acceleration structures - `BVH::build` over all instances and over all
// emitters (instance.rs:365-371,422-428) - before re-uploading every buffer.  Here the per-instance work runs in one
// kernel over the moved instances, reading their new model matrices straight from pinned host memory, and the two
// trees are REFIT in place: same topology (in all eight direction-threaded orderings), every inner box recomputed as the
// union of the leaf boxes below it.  No tree is built and no scene buffer crosses PCIe; the host keeps re-building the
// reference's SAH tree at its leisure (hk_upload_scene_instances, asynchronous) when motion has degraded the refit one.
//
// Exactness: everything per instance is the host builder's arithmetic (scene_builder.cpp) operation for operation -
// the double-precision cofactor inverse rounded once, the 8-corner AABB seeded at zero, 0.5 * |cross| triangle areas
// summed in order, the alias-table construction of mod.rs:320-376 - so a refit scene equals what the host path would
// upload for the same tree shape, bit for bit (tests/test_device_refit.py feeds the oracle exactly that).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "hk_device.hpp"
#include "hk_kernels.hpp"

namespace hkd {

namespace {
__device__ __forceinline__ void mat_point(const float* m, const float* p, float* out) {  // glam Mat4::transform_point3
  for (int k = 0; k < 3; ++k) out[k] = m[k] * p[0] + m[4 + k] * p[1] + m[8 + k] * p[2] + m[12 + k];
}
__device__ __forceinline__ void mat_vector(const float* m, const float* p, float* out) {  // glam Mat4::transform_vector3
  for (int k = 0; k < 3; ++k) out[k] = m[k] * p[0] + m[4 + k] * p[1] + m[8 + k] * p[2];
}
// inverse().transpose() in double, rounded once: scene_builder.cpp inverse_transpose, term for term
__device__ bool inverse_transpose(const float* m, float* out) {
  double a[16], inv[16];
  for (int i = 0; i < 16; ++i) a[i] = m[i];
  inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
  inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
  inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
  inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
  inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
  inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
  inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
  inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
  inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
  inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
  inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
  inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
  inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
  inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
  inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
  inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
  const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
  if (det == 0.0) return false;
  const double id = 1.0 / det;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) out[c * 4 + r] = (float)(inv[r * 4 + c] * id);
  return true;
}
__device__ __forceinline__ float hmin(float a, float b) { return b < a ? b : a; }  // std::min / std::max of the host builder
__device__ __forceinline__ float hmax(float a, float b) { return a < b ? b : a; }
}  // namespace

// ------------------------------------------------------------------ per moved instance (instance.rs:286-325,380-420)
// One thread per update record.  `failed` receives 1 + instance id of a singular transform (the record is skipped).
__global__ __launch_bounds__(64) void k_refit_instances(RefitScene s, const RefitUpdate* __restrict__ updates, uint32_t n_updates, uint32_t* failed) {
  const uint32_t u = blockIdx.x * 64u + threadIdx.x;
  if (u >= n_updates) return;
  const RefitUpdate up = updates[u];
  const uint32_t id = up.instance;
  DInstance& d = s.instances[id];
  if (up.moved == 0u) {  // moved last time, at rest now: its previous model is its model (PreviousMeshUniform)
    d.moved = 0u;
    return;
  }
  float itm[16];
  if (!inverse_transpose(up.model, itm)) {
    if (failed) atomicMax(failed, id + 1u);  // (the host has checked every pose with the same arithmetic: it passes NULL)
    return;
  }
  // previous model = the model this instance was rendered with so far (prepass.wgsl:50,96)
  s.prev_models[4u * id + 0u] = d.m0;
  s.prev_models[4u * id + 1u] = d.m1;
  s.prev_models[4u * id + 2u] = d.m2;
  s.prev_models[4u * id + 3u] = d.m3;
  const float* m = up.model;
  d.im0 = make_float4(itm[0], itm[4], itm[8], itm[12]);  // context.hip build_dynamic_region
  d.im1 = make_float4(itm[1], itm[5], itm[9], itm[13]);
  d.im2 = make_float4(itm[2], itm[6], itm[10], itm[14]);
  d.im3 = make_float4(itm[3], itm[7], itm[11], itm[15]);
  d.m0 = make_float4(m[0], m[1], m[2], m[3]);
  d.m1 = make_float4(m[4], m[5], m[6], m[7]);
  d.m2 = make_float4(m[8], m[9], m[10], m[11]);
  d.m3 = make_float4(m[12], m[13], m[14], m[15]);
  d.n0 = make_float4(itm[0], itm[1], itm[2], 0.0f);
  d.n1 = make_float4(itm[4], itm[5], itm[6], 0.0f);
  d.n2 = make_float4(itm[8], itm[9], itm[10], 0.0f);
  d.moved = 1u;
  // world AABB: the mesh box's 8 corners as vectors, min / max seeded at zero, then the centre added (instance.rs:286-305)
  float center[3];
  mat_point(m, up.aabb_center, center);
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  for (int corner = 0; corner < 8; ++corner) {
    const float e[3] = {up.aabb_half[0] * (float)(2 * (corner & 1) - 1), up.aabb_half[1] * (float)(2 * ((corner >> 1) & 1) - 1),
                        up.aabb_half[2] * (float)(2 * ((corner >> 2) & 1) - 1)};
    float t[3];
    mat_vector(m, e, t);
    for (int k = 0; k < 3; ++k) {
      mn[k] = hmin(mn[k], t[k]);
      mx[k] = hmax(mx[k], t[k]);
    }
  }
  for (int k = 0; k < 3; ++k) {
    mn[k] = mn[k] + center[k];
    mx[k] = mx[k] + center[k];
  }
  s.inst_lo[id] = make_float4(mn[0], mn[1], mn[2], 0.0f);
  s.inst_hi[id] = make_float4(mx[0], mx[1], mx[2], 0.0f);
  const uint32_t e = s.emissive_of_instance[id];
  if (e == HK_U32_MAX) return;
  // the emitter record, instance.rs:380-420
  DEmissive& em = s.emissives[e];
  const float4 col = s.materials[4u * d.material + 1u];
  const float intensity = 255.0f * col.w * sqrtf(col.x * col.x + col.y * col.y + col.z * col.z);
  float pos[3], d2 = 0.0f;
  for (int k = 0; k < 3; ++k) {
    pos[k] = 0.5f * (mx[k] + mn[k]);
    const float dd = mx[k] - mn[k];
    d2 += dd * dd;
  }
  const float radius = 0.5f * sqrtf(d2) + sqrtf(intensity);
  em.position_radius = make_float4(pos[0], pos[1], pos[2], radius);
  // (its area sum and alias table: k_refit_emitters, one WAVE per emitter)
}

// GpuMesh::transformed_primitive_areas (mod.rs:307-318) and build_alias_table (mod.rs:320-376) of every emitter among the update
// records: one WAVE per record (one THREAD per emitter ran all of it until round 3 - O(n) dependent global loads for the
// 2 000-triangle emissive sphere of examples/scene.rs:231-235).  What has to be sequential stays sequential, so every bit is the
// host builder's:
//   1. the n triangle areas: lanes stride over the triangles (independent values, the sequential code's arithmetic each);
//   2. surface_area: ONE lane adds them in index order (f32 addition does not associate);
//   3. the two stacks of the alias construction hold the over- / under-full indices in increasing order: a wave-wide ordered
//      compaction, 64 triangles per step, writes exactly those sequences;
//   4. the pairing loop pops and pushes the stacks one element at a time: one lane - but on stacks that live in LDS whenever the
//      five work arrays fit (n <= 3 264), so its dependent loads cost ~100 cycles instead of a trip to L2 each.
constexpr uint32_t HK_EMITTER_LDS_TRIANGLES = 3264u;  // 5 arrays x 4 B x n <= 65 280 B of dynamic LDS
__global__ __launch_bounds__(64) void k_refit_emitters(RefitScene s, const RefitUpdate* __restrict__ updates, uint32_t n_updates, uint32_t lds_triangles) {
  extern __shared__ __attribute__((aligned(16))) float emitter_lds[];
  const uint32_t u = blockIdx.x, lane = threadIdx.x;
  if (u >= n_updates) return;
  if (updates[u].moved == 0u) return;
  const uint32_t id = updates[u].instance;
  const uint32_t e = s.emissive_of_instance[id];
  if (e == HK_U32_MAX) return;
  float m[16];  // (the records sit in pinned host memory: read once)
  for (int k = 0; k < 16; ++k) m[k] = updates[u].model[k];
  {  // a singular pose leaves the instance record alone (k_refit_instances): then the emitter record stays as well
    float itm[16];
    if (!inverse_transpose(m, itm)) return;
  }
  DEmissive& em = s.emissives[e];
  const uint32_t prim0 = s.instances[id].primitive;
  const uint32_t n = em.alias_count;  // = the mesh's triangle count
  float* work = n <= lds_triangles ? emitter_lds : s.alias_scratch + 5u * (size_t)em.alias_offset;  // (lds_triangles: what the launch reserved)
  float* areas = work;  // [n] areas, then the two stacks (id, prob) x 2
  uint32_t* over_id = reinterpret_cast<uint32_t*>(work + n);
  float* over_p = work + 2u * (size_t)n;
  uint32_t* under_id = reinterpret_cast<uint32_t*>(work + 3u * (size_t)n);
  float* under_p = work + 4u * (size_t)n;
  for (uint32_t i = lane; i < n; i += 64u) {
    const uint32_t prim = prim0 + i;
    const float4 q0 = s.tri_v0[prim], q1 = s.tri_v1[prim], q2 = s.tri_v2[prim];
    const float l0[3] = {q0.x, q0.y, q0.z}, l1[3] = {q1.x, q1.y, q1.z}, l2[3] = {q2.x, q2.y, q2.z};
    float v0[3], v1[3], v2[3];
    mat_point(m, l0, v0);
    mat_point(m, l1, v1);
    mat_point(m, l2, v2);
    const float a[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
    const float b[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
    const float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    areas[i] = 0.5f * fabsf(sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]));
  }
  __syncthreads();
  float surface_area = 0.0f;
  if (lane == 0u) {
    for (uint32_t i = 0; i < n; ++i) surface_area += areas[i];
    em.surface_area = surface_area;
  }
  surface_area = __shfl(surface_area, 0);
  const float mean_area = surface_area / (float)n;
  uint32_t n_over = 0u, n_under = 0u;  // wave-uniform
  float2* table = s.alias + em.alias_offset;
  for (uint32_t base = 0; base < n; base += 64u) {
    const uint32_t i = base + lane;
    const float p = i < n ? areas[i] / mean_area : 1.0f;
    const unsigned long long mo = __ballot(p > 1.0f), mu = __ballot(p < 1.0f);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (p > 1.0f) {
      const uint32_t at = n_over + (uint32_t)__popcll(mo & below);
      over_id[at] = i;
      over_p[at] = p;
    }
    if (p < 1.0f) {
      const uint32_t at = n_under + (uint32_t)__popcll(mu & below);
      under_id[at] = i;
      under_p[at] = p;
    }
    n_over += (uint32_t)__popcll(mo);
    n_under += (uint32_t)__popcll(mu);
    if (i < n) table[i] = make_float2(0.0f, u2f(i));
  }
  __syncthreads();
  if (lane != 0u) return;
  while (n_under != 0u && n_over != 0u) {
    --n_over;
    const uint32_t oid = over_id[n_over];
    float op = over_p[n_over];
    --n_under;
    const uint32_t uid = under_id[n_under];
    const float upb = under_p[n_under];
    const float delta = 1.0f - upb;
    op -= delta;
    if (op > 1.0f) { over_id[n_over] = oid; over_p[n_over] = op; ++n_over; }
    else if (op < 1.0f) { under_id[n_under] = oid; under_p[n_under] = op; ++n_under; }
    table[uid] = make_float2(delta, u2f(oid));
  }
}

// ------------------------------------------------------------------ refit of a flat skip-link BVH
// One WAVE per node.  A node whose entry carries the leaf flag (a leaf, or a navigator that took over its single leaf's role:
// context.hip fold_leaf_navigators) gets its shape's box; every other node is the navigator of the subtree stored in
// (i, exit): the union of the leaf boxes in that range - min / max are exact, so the order of the union does not matter.
// LIGHT = the light BVH (leaf box = emitter position -/+ radius, light.wgsl:633-636), else the TLAS (leaf box = the
// instance's world AABB, light.wgsl:454-457).  `count` nodes per ordering, `orderings` orderings back to back.
template <bool LIGHT>
__global__ __launch_bounds__(256) void k_refit_flat_bvh(RefitScene s, float4* lo, float4* hi, uint32_t stride /* float4 between lo of consecutive nodes */,
                                                        uint32_t count, uint32_t orderings) {
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (wave >= count * orderings) return;
  const uint32_t ord = wave / count, i = wave - ord * count;
  float4* nlo = lo + (size_t)ord * count * stride;
  float4* nhi = hi + (size_t)ord * count * stride;
  auto shape_box = [&](uint32_t shape, f3& mn, f3& mx) {
    if (LIGHT) {
      const float4 pr = s.emissives[shape].position_radius;
      mn = F3(pr.x - pr.w, pr.y - pr.w, pr.z - pr.w);
      mx = F3(pr.x + pr.w, pr.y + pr.w, pr.z + pr.w);
    } else {
      mn = xyz(F4(s.inst_lo[shape]));
      mx = xyz(F4(s.inst_hi[shape]));
    }
  };
  const uint32_t entry = f2u(nlo[(size_t)i * stride].w), exit_ = f2u(nhi[(size_t)i * stride].w);
  f3 mn = F3(INFINITY, INFINITY, INFINITY), mx = F3(-INFINITY, -INFINITY, -INFINITY);
  if (entry >= HK_LEAF) {
    shape_box(entry - HK_LEAF, mn, mx);
  } else {
    for (uint32_t k = i + 1u + lane; k < exit_ && k < count; k += 64u) {
      const uint32_t e = f2u(nlo[(size_t)k * stride].w);
      if (e < HK_LEAF) continue;
      f3 a, b;
      shape_box(e - HK_LEAF, a, b);
      mn = F3(hmin(mn.x, a.x), hmin(mn.y, a.y), hmin(mn.z, a.z));
      mx = F3(hmax(mx.x, b.x), hmax(mx.y, b.y), hmax(mx.z, b.z));
    }
    for (int off = 32; off > 0; off >>= 1) {
      mn = F3(hmin(mn.x, __shfl_xor(mn.x, off)), hmin(mn.y, __shfl_xor(mn.y, off)), hmin(mn.z, __shfl_xor(mn.z, off)));
      mx = F3(hmax(mx.x, __shfl_xor(mx.x, off)), hmax(mx.y, __shfl_xor(mx.y, off)), hmax(mx.z, __shfl_xor(mx.z, off)));
    }
  }
  if (lane == 0u) {
    nlo[(size_t)i * stride] = make_float4(mn.x, mn.y, mn.z, u2f(entry));
    nhi[(size_t)i * stride] = make_float4(mx.x, mx.y, mx.z, u2f(exit_));
  }
}

__global__ __launch_bounds__(256) void k_copy_region_u4(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
}
// world AABB of every instance, from the TLAS leaves of ordering 0 (after a host upload: the refit reads them from here)
__global__ __launch_bounds__(256) void k_gather_instance_boxes(RefitScene s, const float4* __restrict__ tlas, uint32_t count) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const float4 lo = tlas[2u * i], hi = tlas[2u * i + 1u];
  const uint32_t entry = f2u(lo.w);
  if (entry < HK_LEAF) return;
  s.inst_lo[entry - HK_LEAF] = make_float4(lo.x, lo.y, lo.z, 0.0f);
  s.inst_hi[entry - HK_LEAF] = make_float4(hi.x, hi.y, hi.z, 0.0f);
}


// ------------------------------------------------------------------ LBVH rebuild of a flat skip-link BVH (Lauterbach 2009 / Karras 2012)
// hk_rebuild_scene_trees: when motion has degraded a refit tree, a NEW tree over the current leaf boxes is built on the
// device - Morton codes of the box centres, one radix sort (rocPRIM), Karras' parallel hierarchy, boxes bottom-up - and
// written straight into the `bvh` 0.7.1 flatten_custom layout the walk expects: a subtree with L leaves occupies 3L - 2
// consecutive nodes, [navigator of the first child][its subtree][navigator of the second child][its subtree], so the position
// of every node follows from leaf counts alone (no traversal, no stack).  Any binary tree over n shapes has 3n - 2 nodes:
// the new tree fills the old one's storage exactly.  Child order per direction octant = the host's hk_bvh_rethread rule.
struct LbvhBuffers {
  uint32_t n;                   // shapes
  const float4 *box_lo, *box_hi;  // LIGHT: derived from emissives instead
  float* bounds;                // 6 floats: min / max of the box centres (x2)
  uint32_t *codes, *codes_sorted, *ids, *ids_sorted;
  // tree nodes: internal i in [0, n - 1), leaf j as n - 1 + j (j = position in the sorted order)
  uint32_t *parent, *left, *right, *first, *last;  // per internal node (first / last: sorted leaf range it covers)
  uint32_t* leaf_parent;        // per sorted leaf
  float4 *node_lo, *node_hi;    // per tree node (2n - 1)
  uint32_t* arrived;            // per internal node: bottom-up visit counter
  uint8_t* swap;                // per internal node: bit o set = the RIGHT child comes first in ordering o
  uint32_t keep_order0;         // 1: ordering 0 keeps left-before-right (the reference's own flattening of an SAH-built tree)
};
namespace {
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {  // 10 bits -> every third bit
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
template <bool LIGHT>
__device__ __forceinline__ void lbvh_shape_box(const RefitScene& s, const LbvhBuffers& b, uint32_t shape, f3& mn, f3& mx) {
  if (LIGHT) {
    const float4 pr = s.emissives[shape].position_radius;
    mn = F3(pr.x - pr.w, pr.y - pr.w, pr.z - pr.w);
    mx = F3(pr.x + pr.w, pr.y + pr.w, pr.z + pr.w);
  } else {
    mn = xyz(F4(b.box_lo[shape]));
    mx = xyz(F4(b.box_hi[shape]));
  }
}
// common prefix length of the (code, position) keys of sorted leaves i and j; -1 outside the array
__device__ __forceinline__ int lbvh_delta(const uint32_t* __restrict__ codes, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t a = codes[i], c = codes[j];
  if (a != c) return __clz((int)(a ^ c));
  return 32 + __clz(i ^ j);
}
}  // namespace

template <bool LIGHT>
__global__ __launch_bounds__(1024) void k_lbvh_bounds(RefitScene s, LbvhBuffers b) {  // one workgroup
  __shared__ float red[6][16];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = threadIdx.x; i < b.n; i += 1024u) {
    f3 lo, hi;
    lbvh_shape_box<LIGHT>(s, b, i, lo, hi);
    const float c[3] = {lo.x + hi.x, lo.y + hi.y, lo.z + hi.z};
    for (int k = 0; k < 3; ++k) {
      mn[k] = fminf(mn[k], c[k]);
      mx[k] = fmaxf(mx[k], c[k]);
    }
  }
  for (int k = 0; k < 3; ++k)
    for (int off = 32; off > 0; off >>= 1) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], off));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
    }
  if ((threadIdx.x & 63u) == 0u)
    for (int k = 0; k < 3; ++k) {
      red[k][threadIdx.x >> 6] = mn[k];
      red[3 + k][threadIdx.x >> 6] = mx[k];
    }
  __syncthreads();
  if (threadIdx.x < 6u) {
    float v = red[threadIdx.x][0];
    for (int w = 1; w < 16; ++w) v = threadIdx.x < 3u ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
    b.bounds[threadIdx.x] = v;
  }
}
template <bool LIGHT>
__global__ __launch_bounds__(256) void k_lbvh_codes(RefitScene s, LbvhBuffers b) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= b.n) return;
  f3 lo, hi;
  lbvh_shape_box<LIGHT>(s, b, i, lo, hi);
  const float c[3] = {lo.x + hi.x, lo.y + hi.y, lo.z + hi.z};
  uint32_t q[3];
  for (int k = 0; k < 3; ++k) {
    const float ext = b.bounds[3 + k] - b.bounds[k];
    const float t = ext > 0.0f ? (c[k] - b.bounds[k]) / ext : 0.0f;
    q[k] = (uint32_t)fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
  }
  b.codes[i] = (expand_bits(q[0]) << 2) | (expand_bits(q[1]) << 1) | expand_bits(q[2]);
  b.ids[i] = i;
}
// Karras 2012, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", section 4: one thread per internal node
__global__ __launch_bounds__(256) void k_lbvh_hierarchy(LbvhBuffers b) {
  const int n = (int)b.n, i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= n - 1) return;
  const uint32_t* codes = b.codes_sorted;
  const int d = (lbvh_delta(codes, n, i, i + 1) - lbvh_delta(codes, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = lbvh_delta(codes, n, i, i - d);
  int lmax = 2;
  while (lbvh_delta(codes, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (lbvh_delta(codes, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = lbvh_delta(codes, n, i, j);
  int sft = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (lbvh_delta(codes, n, i, i + (sft + t) * d) > dnode) sft += t;
    if (t <= 1) break;
  }
  const int gamma = i + sft * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const uint32_t lc = (lo == gamma) ? (uint32_t)(n - 1 + gamma) : (uint32_t)gamma;              // leaf gamma or internal gamma
  const uint32_t rc = (hi == gamma + 1) ? (uint32_t)(n - 1 + gamma + 1) : (uint32_t)(gamma + 1);
  b.left[i] = lc;
  b.right[i] = rc;
  b.first[i] = (uint32_t)lo;
  b.last[i] = (uint32_t)hi;
  if (lc >= (uint32_t)(n - 1)) b.leaf_parent[lc - (uint32_t)(n - 1)] = (uint32_t)i; else b.parent[lc] = (uint32_t)i;
  if (rc >= (uint32_t)(n - 1)) b.leaf_parent[rc - (uint32_t)(n - 1)] = (uint32_t)i; else b.parent[rc] = (uint32_t)i;
  if (i == 0) b.parent[0] = HK_U32_MAX;
}
// leaf boxes, then every internal node by the second of its children to arrive (min / max are exact: any order gives the same box)
template <bool LIGHT>
__global__ __launch_bounds__(256) void k_lbvh_boxes(RefitScene s, LbvhBuffers b) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= b.n) return;
  f3 mn, mx;
  lbvh_shape_box<LIGHT>(s, b, b.ids_sorted[j], mn, mx);
  b.node_lo[b.n - 1u + j] = make_float4(mn.x, mn.y, mn.z, 0.0f);
  b.node_hi[b.n - 1u + j] = make_float4(mx.x, mx.y, mx.z, 0.0f);
  if (b.n == 1u) return;
  uint32_t p = b.leaf_parent[j];
  for (;;) {
    __threadfence();
    if (atomicAdd(&b.arrived[p], 1u) == 0u) return;  // the sibling subtree is not finished: its thread continues from here
    __threadfence();
    const uint32_t l = b.left[p], r = b.right[p];
    const volatile float4* vlo = b.node_lo;
    const volatile float4* vhi = b.node_hi;
    const f3 al = F3(vlo[l].x, vlo[l].y, vlo[l].z), ah = F3(vhi[l].x, vhi[l].y, vhi[l].z);
    const f3 bl = F3(vlo[r].x, vlo[r].y, vlo[r].z), bh = F3(vhi[r].x, vhi[r].y, vhi[r].z);
    b.node_lo[p] = make_float4(hmin(al.x, bl.x), hmin(al.y, bl.y), hmin(al.z, bl.z), 0.0f);
    b.node_hi[p] = make_float4(hmax(ah.x, bh.x), hmax(ah.y, bh.y), hmax(ah.z, bh.z), 0.0f);
    // child order per direction octant: host_logic.cpp rethread_flat_bvh (the axis along which the two boxes are furthest apart)
    int axis = 0;
    float best = -1.0f, ca_axis = 0.0f, cb_axis = 0.0f;
    const float ca[3] = {al.x + ah.x, al.y + ah.y, al.z + ah.z}, cb[3] = {bl.x + bh.x, bl.y + bh.y, bl.z + bh.z};
    for (int k = 0; k < 3; ++k) {
      const float dd = fabsf(ca[k] - cb[k]);
      if (dd > best) { best = dd; axis = k; ca_axis = ca[k]; cb_axis = cb[k]; }
    }
    const bool a_lower = ca_axis <= cb_axis;
    uint32_t sw = 0u;
    for (uint32_t o = 0; o < 8u; ++o) {
      const bool negative = (o >> axis) & 1u;
      if (!(a_lower != negative)) sw |= 1u << o;
    }
    if (b.keep_order0) sw &= ~1u;
    b.swap[p] = (uint8_t)sw;
    p = b.parent[p];
    if (p == HK_U32_MAX) return;
  }
}
// every tree node except the root writes the navigator in front of its subtree; leaves also write their own slot
__global__ __launch_bounds__(256) void k_lbvh_emit(LbvhBuffers b, float4* lo, float4* hi, uint32_t stride, uint32_t orderings) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x, n = b.n, total = 2u * n - 1u;
  if (t >= total * orderings) return;
  const uint32_t o = t / total, v = t - o * total, count = 3u * n - 2u;
  float4* nlo = lo + (size_t)o * count * stride;
  float4* nhi = hi + (size_t)o * count * stride;
  const bool leaf = v >= n - 1u;
  const uint32_t shape = leaf ? b.ids_sorted[v - (n - 1u)] : 0u;
  const float4 blo = b.node_lo[v], bhi = b.node_hi[v];
  if (n == 1u) {  // flatten_custom of a single leaf: the leaf alone
    nlo[0] = make_float4(blo.x, blo.y, blo.z, u2f(HK_LEAF | shape));
    nhi[0] = make_float4(bhi.x, bhi.y, bhi.z, u2f(1u));
    return;
  }
  if (!leaf && v == 0u) return;  // the root has no navigator
  // start of this node's subtree: walk to the root; a first child starts one node after its parent's start (its navigator),
  // a second child after the whole first branch
  auto leaves_of = [&](uint32_t node) { return node >= n - 1u ? 1u : b.last[node] - b.first[node] + 1u; };
  uint32_t start = 0u, c = v, p = leaf ? b.leaf_parent[v - (n - 1u)] : b.parent[v];
  while (p != HK_U32_MAX) {
    const bool right_first = (b.swap[p] >> o) & 1u;
    const uint32_t first_child = right_first ? b.right[p] : b.left[p];
    start += (c == first_child) ? 1u : 2u + (3u * leaves_of(first_child) - 2u);
    c = p;
    p = b.parent[p];
  }
  const uint32_t size = 3u * leaves_of(v) - 2u;
  if (leaf) {  // folded navigator (context.hip fold_leaf_navigators) + the leaf slot
    nlo[(size_t)(start - 1u) * stride] = make_float4(blo.x, blo.y, blo.z, u2f(HK_LEAF | shape));
    nhi[(size_t)(start - 1u) * stride] = make_float4(bhi.x, bhi.y, bhi.z, u2f(start + 1u));
    nlo[(size_t)start * stride] = make_float4(blo.x, blo.y, blo.z, u2f(HK_LEAF | shape));
    nhi[(size_t)start * stride] = make_float4(bhi.x, bhi.y, bhi.z, u2f(start + 1u));
  } else {
    nlo[(size_t)(start - 1u) * stride] = make_float4(blo.x, blo.y, blo.z, u2f(start));
    nhi[(size_t)(start - 1u) * stride] = make_float4(bhi.x, bhi.y, bhi.z, u2f(start + size));
  }
}


// ------------------------------------------------------------------ the reference's own tree, built on the device
// `bvh` 0.7.1 BVHNode::build (scene_builder.cpp build_recursive; the crate the reference calls at instance.rs:365-371,422-428) is a
// top-down binned SAH: per node the bounds of the shapes and of their centres, the longest axis of the centre bounds, six buckets
// along it, the cheapest of the five splits, the shapes re-ordered bucket by bucket.  Every reduction in it is a min, a max or a
// count, and the re-ordering is a STABLE sort by bucket: nothing depends on the order in which a parallel machine visits the
// shapes.  So the same tree can be built level by level: one workgroup, all nodes of a level at once, the shapes in one array
// that is stably re-sorted per level (block scans of the six bucket flags, segmented at node boundaries, with running counters
// per node and bucket across the 1024-item chunks).  The costs are the host's float expressions term for term, so the splits -
// and with them the shape of the tree - are the host's: tests compare the entry / exit links with the host builder's arrays.
// (Zeros may come out with the other sign than std::min / std::max chains give - a box bound of -0 vs +0 changes no decision.)
struct SahBuffers {
  uint32_t* order[2];      // shape ids, segment by segment
  uint32_t* item_node[2];  // per position: the internal node whose segment it is in, or SAH_DONE once it is a leaf
  uint8_t* item_bucket;
  uint32_t* active[2];     // internal nodes split at this level / created for the next
  uint32_t* acc;           // per internal node: 54 words - bounds (6), centre bounds (6), 6 bucket boxes (36), 6 bucket counts
  float* split;            // per internal node: 4 words - centre-bounds min on the axis, axis size, (bits) axis, (bits) half-split flag
  uint32_t* offsets;       // per internal node: 14 words - first target position of each bucket (6), running counts (6), left count, split bucket
  uint32_t* node_level;    // per internal node: the level of the loop that splits it
  uint32_t* roots;         // nodes handed to k_sah_subtrees
  uint32_t* counters;      // [0] internal nodes allocated, [1] subtree roots, [2] the ping-pong side the top of the tree ended on
};
namespace {
constexpr uint32_t SAH_DONE = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t fkey(float f) {  // monotone map f32 -> u32 (for atomicMin / atomicMax)
  const uint32_t u = f2u(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) { return u2f((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
__device__ __forceinline__ float box_center(float mn, float mx) { return mn + (mx - mn) / 2.0f; }  // Box::center
__device__ __forceinline__ float box_area(const float* mn, const float* mx) {                      // Box::surface_area
  const float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
  return 2.0f * (x * y + x * z + y * z);
}
// min / max of a box into the accumulator words acc[0..5] (keys): when every lane of the wave that takes part adds to the SAME
// accumulator (the top levels of the tree: thousands of shapes per node) the wave reduces first - one hot address takes ~10 ns per
// atomic - otherwise every lane adds on its own.  Called by all lanes of the wave; `take` = this lane has a box for `acc`.
__device__ __forceinline__ void wave_box_accumulate(uint32_t* acc, bool take, const float* mn, const float* mx, uint32_t* count) {
  const unsigned long long m = __ballot(take);
  if (m == 0ull) return;
  const unsigned long long a = (unsigned long long)(size_t)acc;
  const int leader = __ffsll((long long)m) - 1;
  const unsigned long long a0 = __shfl(a, leader);
  const bool uniform = __ballot(take && a != a0) == 0ull;
  if (uniform) {
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
      lo[k] = take ? mn[k] : INFINITY;
      hi[k] = take ? mx[k] : -INFINITY;
      for (int off = 32; off > 0; off >>= 1) {
        lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
        hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
      }
    }
    if ((int)(threadIdx.x & 63u) == leader) {
      for (int k = 0; k < 3; ++k) {
        atomicMin(&acc[k], fkey(lo[k]));
        atomicMax(&acc[3 + k], fkey(hi[k]));
      }
      if (count) atomicAdd(count, (uint32_t)__popcll(m));
    }
  } else if (take) {
    for (int k = 0; k < 3; ++k) {
      atomicMin(&acc[k], fkey(mn[k]));
      atomicMax(&acc[3 + k], fkey(mx[k]));
    }
    if (count) atomicAdd(count, 1u);
  }
}
// the six bucket-flag prefix sums (packed two 16-bit counters per word: a chunk has 1024 items) and the run-head maximum of a chunk
// in ONE pass: three block barriers instead of twenty-one.  lds: 4 x 17 words
__device__ __forceinline__ void block_scan_buckets_and_heads(uint32_t bk, uint32_t head_value, uint32_t* lds, uint32_t pre[3], uint32_t& head) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t v[3] = {0u, 0u, 0u};
  if (bk < 6u) v[bk >> 1] = 1u << (16u * (bk & 1u));
  uint32_t inc[3] = {v[0], v[1], v[2]}, hmax = head_value;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t0 = __shfl_up(inc[0], off), t1 = __shfl_up(inc[1], off), t2 = __shfl_up(inc[2], off), th = __shfl_up(hmax, off);
    if ((int)lane >= off) {
      inc[0] += t0;
      inc[1] += t1;
      inc[2] += t2;
      hmax = max(hmax, th);
    }
  }
  __syncthreads();
  if (lane == 63u) {
    lds[wave] = inc[0];
    lds[17 + wave] = inc[1];
    lds[34 + wave] = inc[2];
    lds[51 + wave] = hmax;
  }
  __syncthreads();
  if (threadIdx.x < 4u) {
    uint32_t* a = lds + 17u * threadIdx.x;
    uint32_t run = 0u;
    for (int w = 0; w < 16; ++w) {
      const uint32_t t = a[w];
      a[w] = run;
      run = threadIdx.x == 3u ? max(run, t) : run + t;
    }
  }
  __syncthreads();
  for (int k = 0; k < 3; ++k) pre[k] = lds[17 * k + wave] + inc[k] - v[k];
  head = max(lds[51 + wave], hmax);
}
}  // namespace

// The level loop over one range [r0, r1) of item positions, run by ONE workgroup of 1024 threads: every node of a level at once.
// `first_level`: node_level value of the range's root (the nodes created below get first_level + 1, ...).  Nodes with at most
// `defer` shapes are not split here but appended to q.roots (their subtrees are built by k_sah_subtrees, one workgroup each, all
// at once).  Returns the ping-pong side that holds the range's final order.
template <bool LIGHT>
__device__ uint32_t sah_levels(const RefitScene& s, const LbvhBuffers& b, const SahBuffers& q, uint32_t r0, uint32_t r1, uint32_t root, uint32_t cur, uint32_t first_level,
                               uint32_t defer) {
  __shared__ uint32_t scan_lds[68];
  __shared__ uint16_t pre[6][1024];  // per chunk: exclusive prefix of each bucket's flags
  __shared__ uint32_t head_of[1024];
  __shared__ uint32_t n_active_lds[2];
  const uint32_t n = b.n, tid = threadIdx.x;
  if (tid == 0u) {
    q.active[cur][r0] = root;
    q.node_level[root] = first_level;
    n_active_lds[cur] = 1u;
    n_active_lds[cur ^ 1u] = 0u;
  }
  __syncthreads();
  for (uint32_t level = first_level;; ++level) {  // (a level with nothing to split ends the loop)
    const uint32_t n_active = n_active_lds[cur];
    if (n_active == 0u) break;
    const uint32_t* order = q.order[cur];
    const uint32_t* item_node = q.item_node[cur];
    uint32_t* order_out = q.order[cur ^ 1u];
    uint32_t* item_node_out = q.item_node[cur ^ 1u];
    const uint32_t* active = q.active[cur] + r0;
    uint32_t* active_out = q.active[cur ^ 1u] + r0;
    auto splitting = [&](uint32_t node) { return node != SAH_DONE && q.node_level[node] == level; };
    // A: accumulators
    for (uint32_t a = tid; a < n_active * 54u; a += 1024u) {
      const uint32_t node = active[a / 54u], w = a % 54u;
      uint32_t init = 0u;                                             // counts
      if (w < 48u) init = (w % 6u) < 3u ? fkey(INFINITY) : fkey(-INFINITY);  // a box: min xyz, max xyz
      q.acc[(size_t)node * 54u + w] = init;
    }
    __syncthreads();
    if (tid == 0u) n_active_lds[cur ^ 1u] = 0u;
    // B: bounds of the shapes and of their centres
    for (uint32_t p0 = r0; p0 < r1; p0 += 1024u) {  // (block-uniform trip count: the wave reductions want whole waves)
      const uint32_t p = p0 + tid;
      const uint32_t node = p < r1 ? item_node[p] : SAH_DONE;
      const bool take = splitting(node);
      float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, cc[3] = {0, 0, 0};
      if (take) {
        f3 lo, hi;
        lbvh_shape_box<LIGHT>(s, b, order[p], lo, hi);
        mn[0] = lo.x; mn[1] = lo.y; mn[2] = lo.z;
        mx[0] = hi.x; mx[1] = hi.y; mx[2] = hi.z;
        for (int k = 0; k < 3; ++k) cc[k] = box_center(mn[k], mx[k]);
      }
      uint32_t* acc = q.acc + (size_t)(take ? node : 0u) * 54u;
      wave_box_accumulate(acc, take, mn, mx, nullptr);
      wave_box_accumulate(acc + 6, take, cc, cc, nullptr);
    }
    __syncthreads();
    // C: split axis
    for (uint32_t a = tid; a < n_active; a += 1024u) {
      const uint32_t node = active[a];
      const uint32_t* acc = q.acc + (size_t)node * 54u;
      float cmn[3], cmx[3];
      for (int k = 0; k < 3; ++k) {
        cmn[k] = funkey(acc[6 + k]);
        cmx[k] = funkey(acc[9 + k]);
      }
      const float x = cmx[0] - cmn[0], y = cmx[1] - cmn[1], z = cmx[2] - cmn[2];
      const int axis = (x > y && x > z) ? 0 : (y > z ? 1 : 2);  // Box::largest_axis
      const float axis_size = cmx[axis] - cmn[axis];
      float* sp = q.split + (size_t)node * 4u;
      sp[0] = cmn[axis];
      sp[1] = axis_size;
      sp[2] = u2f((uint32_t)axis);
      sp[3] = u2f(axis_size < 0.00001f ? 1u : 0u);  // shapes too close together: the index list is cut in half
    }
    __syncthreads();
    // D: buckets
    for (uint32_t p0 = r0; p0 < r1; p0 += 1024u) {
      const uint32_t p = p0 + tid;
      const uint32_t node = p < r1 ? item_node[p] : SAH_DONE;
      bool take = splitting(node);
      float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
      int bk = 0;
      if (take) {
        const float* sp = q.split + (size_t)node * 4u;
        take = f2u(sp[3]) == 0u;
        if (take) {
          f3 lo, hi;
          lbvh_shape_box<LIGHT>(s, b, order[p], lo, hi);
          mn[0] = lo.x; mn[1] = lo.y; mn[2] = lo.z;
          mx[0] = hi.x; mx[1] = hi.y; mx[2] = hi.z;
          const int axis = (int)f2u(sp[2]);
          const float rel = (box_center(mn[axis], mx[axis]) - sp[0]) / sp[1];
          bk = (int)(rel * (6.0f - 0.01f));
          bk = min(max(bk, 0), 5);
          q.item_bucket[p] = (uint8_t)bk;
        }
      }
      for (int k6 = 0; k6 < 6; ++k6) {  // bucket by bucket: lanes of one node and one bucket share an accumulator
        const bool mine = take && bk == k6;
        uint32_t* acc = q.acc + (size_t)(mine ? node : 0u) * 54u;
        wave_box_accumulate(acc + 12 + 6 * k6, mine, mn, mx, acc + 48 + k6);
      }
    }
    __syncthreads();
    // E: the cheapest split, the children
    for (uint32_t a = tid; a < n_active; a += 1024u) {
      const uint32_t node = active[a];
      const uint32_t* acc = q.acc + (size_t)node * 54u;
      const float* sp = q.split + (size_t)node * 4u;
      uint32_t* off = q.offsets + (size_t)node * 14u;
      const uint32_t begin = b.first[node], count = b.last[node] - begin + 1u;
      uint32_t n_left = count / 2u, split_bucket = 6u;  // 6: cut the list in half
      if (f2u(sp[3]) == 0u) {
        float bounds_mn[3], bounds_mx[3];
        for (int k = 0; k < 3; ++k) {
          bounds_mn[k] = funkey(acc[k]);
          bounds_mx[k] = funkey(acc[3 + k]);
        }
        const float total_area = box_area(bounds_mn, bounds_mx);
        float min_cost = INFINITY;
        uint32_t min_bucket = 0u;
        for (uint32_t i = 0; i < 5u; ++i) {
          float lmn[3] = {INFINITY, INFINITY, INFINITY}, lmx[3] = {-INFINITY, -INFINITY, -INFINITY};
          float rmn[3] = {INFINITY, INFINITY, INFINITY}, rmx[3] = {-INFINITY, -INFINITY, -INFINITY};
          uint32_t ln = 0u, rn = 0u;
          for (uint32_t k6 = 0; k6 < 6u; ++k6) {
            const uint32_t* bb = acc + 12u + 6u * k6;
            float* tmn = k6 <= i ? lmn : rmn;
            float* tmx = k6 <= i ? lmx : rmx;
            for (int k = 0; k < 3; ++k) {
              tmn[k] = hmin(tmn[k], funkey(bb[k]));
              tmx[k] = hmax(tmx[k], funkey(bb[3 + k]));
            }
            if (k6 <= i) ln += acc[48u + k6]; else rn += acc[48u + k6];
          }
          const float cost = ((float)ln * box_area(lmn, lmx) + (float)rn * box_area(rmn, rmx)) / total_area;
          if (cost < min_cost) {
            min_bucket = i;
            min_cost = cost;
          }
        }
        uint32_t best_left = 0u;  // (all costs NaN: the host keeps bucket 0 as the split)
        for (uint32_t k6 = 0; k6 <= min_bucket; ++k6) best_left += acc[48u + k6];
        if (best_left != 0u && best_left != count) {
          n_left = best_left;
          split_bucket = min_bucket;
        }  // (an empty side - NaN costs - falls back to the half cut, like the host)
      }
      uint32_t run = begin;
      for (uint32_t k6 = 0; k6 < 6u; ++k6) {
        off[k6] = run;
        run += acc[48u + k6];
        off[6u + k6] = 0u;
      }
      off[12] = n_left;
      off[13] = split_bucket;
      // children: a side with one shape is a leaf at its position, a larger one an internal node - split at the next level, or
      // handed to a workgroup of its own when it is small enough
      const uint32_t n_right = count - n_left;
      uint32_t child[2];
      for (int side = 0; side < 2; ++side) {
        const uint32_t c_begin = side == 0 ? begin : begin + n_left, c_count = side == 0 ? n_left : n_right;
        if (c_count == 1u) {
          child[side] = (n - 1u) + c_begin;
          b.leaf_parent[c_begin] = node;
        } else {
          const uint32_t id = atomicAdd(&q.counters[0], 1u);
          child[side] = id;
          b.first[id] = c_begin;
          b.last[id] = c_begin + c_count - 1u;
          b.parent[id] = node;
          if (c_count <= defer) {
            q.node_level[id] = SAH_DONE;  // (not a level of this loop)
            q.roots[atomicAdd(&q.counters[1], 1u)] = id;
          } else {
            q.node_level[id] = level + 1u;
            active_out[atomicAdd(&n_active_lds[cur ^ 1u], 1u)] = id;
          }
        }
      }
      b.left[node] = child[0];
      b.right[node] = child[1];
    }
    __syncthreads();
    // F: stable re-order, bucket by bucket inside every segment; chunk after chunk so that the running counts stay in order
    for (uint32_t c0 = r0; c0 < r1; c0 += 1024u) {
      const uint32_t p = c0 + tid;
      const bool in = p < r1;
      const uint32_t node = in ? item_node[p] : SAH_DONE;
      const bool split_now = in && splitting(node);
      const bool moving = split_now && q.offsets[(size_t)node * 14u + 13u] != 6u;
      const uint32_t bk = moving ? q.item_bucket[p] : 7u;
      const bool head = tid == 0u || !in || item_node[p - 1u] != node;
      uint32_t packed[3], h;
      block_scan_buckets_and_heads(bk, head ? tid : 0u, scan_lds, packed, h);
      for (uint32_t k6 = 0; k6 < 6u; ++k6) pre[k6][tid] = (uint16_t)((packed[k6 >> 1] >> (16u * (k6 & 1u))) & 0xFFFFu);
      head_of[tid] = h;
      __syncthreads();
      if (split_now) {
        const uint32_t* off = q.offsets + (size_t)node * 14u;
        const uint32_t begin = b.first[node], n_left = off[12];
        uint32_t target = p;
        if (moving) target = off[bk] + off[6u + bk] + ((uint32_t)pre[bk][tid] - (uint32_t)pre[bk][h]);
        const bool goes_left = target < begin + n_left;
        const uint32_t child = goes_left ? b.left[node] : b.right[node];
        order_out[target] = order[p];
        item_node_out[target] = child >= n - 1u ? SAH_DONE : child;
      } else if (in) {  // a leaf already, or a node that waits for a workgroup of its own
        order_out[p] = order[p];
        item_node_out[p] = node;
      }
      __syncthreads();
      // the last item of a node's run in this chunk adds the chunk's counts to the node's running counts
      if (moving) {
        const bool last_of_run = tid == 1023u || p + 1u >= r1 || item_node[p + 1u] != node;
        if (last_of_run) {
          uint32_t* off = q.offsets + (size_t)node * 14u;
          for (uint32_t k6 = 0; k6 < 6u; ++k6) off[6u + k6] += ((uint32_t)pre[k6][tid] + (bk == k6 ? 1u : 0u)) - (uint32_t)pre[k6][h];
        }
      }
      __syncthreads();
    }
    cur ^= 1u;
  }
  __syncthreads();
  return cur;
}

// SAH_SUBTREE: nodes with at most this many shapes are built by a workgroup of their own (k_sah_subtrees), all of them at once
constexpr uint32_t SAH_SUBTREE = 1024u;
// the top of the tree: one workgroup over the whole array, down to nodes of at most SAH_SUBTREE shapes.  Outputs the LbvhBuffers
// topology (ids_sorted = leaf order, parent / left / right / first / last / leaf_parent) for k_lbvh_boxes + k_lbvh_emit
template <bool LIGHT>
__global__ __launch_bounds__(1024) void k_sah_build(RefitScene s, LbvhBuffers b, SahBuffers q) {
  const uint32_t n = b.n, tid = threadIdx.x;
  if (n == 1u) {
    if (tid == 0u) {
      b.ids_sorted[0] = 0u;
      q.counters[1] = 0u;
    }
    return;
  }
  for (uint32_t i = tid; i < n; i += 1024u) {
    q.order[0][i] = i;
    q.item_node[0][i] = 0u;
  }
  if (tid == 0u) {
    b.first[0] = 0u;
    b.last[0] = n - 1u;
    b.parent[0] = HK_U32_MAX;
    q.counters[0] = 1u;  // internal nodes allocated
    q.counters[1] = 0u;  // subtree roots
  }
  __syncthreads();
  const uint32_t cur = sah_levels<LIGHT>(s, b, q, 0u, n, 0u, 0u, 0u, n > SAH_SUBTREE ? SAH_SUBTREE : 0u);
  if (tid == 0u) q.counters[2] = cur;  // the side the subtree workgroups start from
  // leaves fixed at this stage are final; the ranges of the deferred nodes are copied by their own workgroups
  for (uint32_t i = tid; i < n; i += 1024u)
    if (q.item_node[cur][i] == SAH_DONE) b.ids_sorted[i] = q.order[cur][i];
}
// the subtrees below: one workgroup per deferred node, all at once
template <bool LIGHT>
__global__ __launch_bounds__(1024) void k_sah_subtrees(RefitScene s, LbvhBuffers b, SahBuffers q) {
  const uint32_t n_roots = q.counters[1], start = q.counters[2];
  for (uint32_t r = blockIdx.x; r < n_roots; r += gridDim.x) {
    const uint32_t root = q.roots[r], r0 = b.first[root], r1 = b.last[root] + 1u;
    const uint32_t cur = sah_levels<LIGHT>(s, b, q, r0, r1, root, start, 0x40000000u, 0u);
    for (uint32_t i = r0 + threadIdx.x; i < r1; i += 1024u) b.ids_sorted[i] = q.order[cur][i];
    __syncthreads();
  }
}

}  // namespace hkd

namespace hk {
using namespace hkd;

void launch_copy_region(hipStream_t st, void* dst, const void* src, size_t bytes) {
  const size_t n = bytes / 16;
  if (!n) return;
  hipLaunchKernelGGL(k_copy_region_u4, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, n);
}
void launch_gather_instance_boxes(hipStream_t st, const RefitScene& s, const float4* tlas, uint32_t tlas_count) {
  if (!tlas_count) return;
  hipLaunchKernelGGL(k_gather_instance_boxes, dim3((tlas_count + 255u) / 256u), dim3(256), 0, st, s, tlas, tlas_count);
}
void launch_refit(hipStream_t st, const RefitScene& s, const RefitUpdate* updates, uint32_t n_updates, uint32_t n_emitter_updates, uint32_t emitter_triangles,
                  uint32_t* failed, float4* tlas, uint32_t tlas_count,
                  uint32_t orderings, float4* light_lo, float4* light_hi, uint32_t light_count) {
  if (n_updates) {
    hipLaunchKernelGGL(k_refit_instances, dim3((n_updates + 63u) / 64u), dim3(64), 0, st, s, updates, n_updates, failed);
    // the host puts the records of moved EMITTERS first: one workgroup (= one wave) per such record only, with the LDS its largest
    // mesh needs (a 64 KB reservation on thousands of workgroups that exit at once capped everybody's occupancy)
    if (n_emitter_updates) {
      const uint32_t lds_triangles = std::min(emitter_triangles, HK_EMITTER_LDS_TRIANGLES);
      hipLaunchKernelGGL(k_refit_emitters, dim3(n_emitter_updates), dim3(64), lds_triangles * 5u * 4u, st, s, updates, n_emitter_updates, lds_triangles);
    }
  }
  // TLAS nodes are interleaved (lo, hi) pairs, the light BVH two planes
  if (tlas_count) {
    const uint32_t waves = tlas_count * orderings;
    hipLaunchKernelGGL((k_refit_flat_bvh<false>), dim3((waves + 3u) / 4u), dim3(256), 0, st, s, tlas, tlas + 1, 2u, tlas_count, orderings);
  }
  if (light_count) hipLaunchKernelGGL((k_refit_flat_bvh<true>), dim3((light_count + 3u) / 4u), dim3(256), 0, st, s, light_lo, light_hi, 1u, light_count, 1u);
}


// scratch of one tree build over n shapes: bytes, and the carving of a single allocation
size_t lbvh_scratch_bytes(uint32_t n, size_t* sort_temp_bytes) {
  size_t temp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, temp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 30);
  temp = (temp + 255) & ~(size_t)255;
  if (sort_temp_bytes) *sort_temp_bytes = temp;
  const size_t nn = ((size_t)n + 63) & ~(size_t)63;
  return 256 + temp + nn * (4 * 4 /* codes, ids x2 */ + 5 * 4 /* parent, left, right, first, last */ + 4 /* leaf_parent */ + 4 /* arrived */ + 4 /* swap (padded) */) +
         2 * nn * 2 * 16 /* node boxes, 2n - 1 */ + nn * (4 /* item_node[1] */ + 4 /* item_bucket, padded */ + 2 * 4 /* active lists */ + 54 * 4 + 4 * 4 + 14 * 4 + 2 * 4) /* SAH build */;
}
// mode 0: LBVH (Morton order), mode 1: the reference's binned SAH (`bvh` 0.7.1).  `lo` / `hi` = the node array to overwrite,
// `stride` float4 between consecutive nodes (2: interleaved pairs, 1: two planes)
int launch_tree_build(hipStream_t st, int mode, bool light, const RefitScene& s, uint32_t n, const float4* box_lo, const float4* box_hi, void* scratch, float4* lo,
                      float4* hi, uint32_t stride, uint32_t orderings) {
  if (n == 0) return 0;
  size_t temp = 0;
  (void)lbvh_scratch_bytes(n, &temp);
  const size_t nn = ((size_t)n + 63) & ~(size_t)63;
  uint8_t* p = (uint8_t*)scratch;
  LbvhBuffers b;
  b.n = n;
  b.box_lo = box_lo;
  b.box_hi = box_hi;
  b.keep_order0 = mode == 1 ? 1u : 0u;
  b.bounds = (float*)p; p += 256;
  void* sort_temp = p; p += temp;
  auto u32 = [&]() { uint32_t* q = (uint32_t*)p; p += nn * 4; return q; };
  b.codes = u32(); b.codes_sorted = u32(); b.ids = u32(); b.ids_sorted = u32();
  b.parent = u32(); b.left = u32(); b.right = u32(); b.first = u32(); b.last = u32(); b.leaf_parent = u32(); b.arrived = u32();
  b.swap = (uint8_t*)u32();
  b.node_lo = (float4*)p; p += 2 * nn * 16;
  b.node_hi = (float4*)p; p += 2 * nn * 16;
  (void)hipMemsetAsync(b.arrived, 0, nn * 4, st);
  (void)hipMemsetAsync(b.swap, 0, nn * 4, st);
  const dim3 per_shape((n + 255u) / 256u);
  if (mode == 1) {
    SahBuffers q;
    q.order[0] = b.codes; q.order[1] = b.codes_sorted;       // (the Morton arrays are free in this mode)
    q.item_node[0] = b.ids; q.item_node[1] = (uint32_t*)p; p += nn * 4;
    q.item_bucket = (uint8_t*)p; p += nn * 4;
    q.active[0] = (uint32_t*)p; p += nn * 4;
    q.active[1] = (uint32_t*)p; p += nn * 4;
    q.acc = (uint32_t*)p; p += nn * 54 * 4;
    q.split = (float*)p; p += nn * 4 * 4;
    q.offsets = (uint32_t*)p; p += nn * 14 * 4;
    q.node_level = (uint32_t*)p; p += nn * 4;
    q.roots = (uint32_t*)p; p += nn * 4;
    q.counters = (uint32_t*)b.bounds;                        // (256 B, unused by this mode)
    const dim3 subtrees((unsigned)std::min<size_t>(std::max<size_t>(n / 2, 1), 4096));
    if (light) {
      hipLaunchKernelGGL((k_sah_build<true>), dim3(1), dim3(1024), 0, st, s, b, q);
      if (n > SAH_SUBTREE) hipLaunchKernelGGL((k_sah_subtrees<true>), subtrees, dim3(1024), 0, st, s, b, q);
    } else {
      hipLaunchKernelGGL((k_sah_build<false>), dim3(1), dim3(1024), 0, st, s, b, q);
      if (n > SAH_SUBTREE) hipLaunchKernelGGL((k_sah_subtrees<false>), subtrees, dim3(1024), 0, st, s, b, q);
    }
  } else {
    if (light) {
      hipLaunchKernelGGL((k_lbvh_bounds<true>), dim3(1), dim3(1024), 0, st, s, b);
      hipLaunchKernelGGL((k_lbvh_codes<true>), per_shape, dim3(256), 0, st, s, b);
    } else {
      hipLaunchKernelGGL((k_lbvh_bounds<false>), dim3(1), dim3(1024), 0, st, s, b);
      hipLaunchKernelGGL((k_lbvh_codes<false>), per_shape, dim3(256), 0, st, s, b);
    }
    if (rocprim::radix_sort_pairs(sort_temp, temp, (const uint32_t*)b.codes, b.codes_sorted, (const uint32_t*)b.ids, b.ids_sorted, (size_t)n, 0, 30, st) != hipSuccess) return 1;
    if (n > 1) hipLaunchKernelGGL(k_lbvh_hierarchy, dim3((n + 254u) / 256u), dim3(256), 0, st, b);
  }
  if (light) hipLaunchKernelGGL((k_lbvh_boxes<true>), per_shape, dim3(256), 0, st, s, b);
  else hipLaunchKernelGGL((k_lbvh_boxes<false>), per_shape, dim3(256), 0, st, s, b);
  const uint32_t threads = (2u * n - 1u) * orderings;
  hipLaunchKernelGGL(k_lbvh_emit, dim3((threads + 255u) / 256u), dim3(256), 0, st, b, lo, hi, stride, orderings);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace hk
