// scene_layout.hip - uploads and the layout conversion behind them: the reference-layout scene arrays a host hands over
// (src/mesh_material/mesh.rs:43-64, material.rs:201-202, instance.rs:82-108, src/lib.rs:189-219) become the device's scene blob -
// SoA node / triangle / vertex planes, leaf boxes filled in, navigators folded, direction-threaded orderings, the one-level tree of
// scenes under one transform, two slots of the instance-level region with asynchronous updates.
#include "hk_context.hpp"

using namespace hk;
using namespace hkd;

namespace {
__global__ void k_copy_u4(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

}  // namespace

namespace hk {
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
  if (!(c->flags & HK_CTX_EXACT_TRAVERSAL) && !c->threaded && !c->instances.empty() && c->instances.size() <= 0xFFFFu && c->flat_walk) {
    bool shared = true;
    for (const HkInstance& in : c->instances)
      if (memcmp(in.inverse_transpose_model, c->instances[0].inverse_transpose_model, 64) != 0) shared = false;
    // direction orderings: as many (8, 4, 2, 1) as keep the node array within 4 KB - every workgroup copies the blob into LDS and
    // the LDS a workgroup holds bounds the workgroups per CU; measured on the Cornell box (71 nodes, tools/ab_flat.sh): 1 / 2 / 4 / 8
    // orderings walk equally fast (0.27 ms k_indirect) and the direct-light kernels lose 13 % with the 18 KB of eight
    uint32_t want = 8, budget = 4096;
    if (c->flat_orderings > 0) { want = (uint32_t)c->flat_orderings; budget = HK_LDS_SCENE_BYTES; }  // (hk_debug_set_option: the A/B of the orderings)
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
  c->scene_epoch += 1;   // (scene memory is about to be written on the main stream: hk_context.hpp, primary-ray pipelining)
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
  if (c->trace_update) fprintf(stderr, "  build_dynamic_region %.2f ms (%zu bytes)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - tb0_, dyn.bytes.size());
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
    // (the slot written now was read by the frame before last: its direct-light dispatches may still sit on the side stream - round 6,
    // hk_context.hpp side_done - so the copy goes behind them)
    if ((rc = join_side(c))) return rc;
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
  c->wide_tlas_dirty = true;                       // (the wide records follow the trees: context.hip ensure_wide)
  c->wide_mesh_check = true;                       // (... and the instance set may now use a mesh tree that has no records yet)
  if (need_static) c->wide_blas_dirty = true;
  return HK_OK;
}

}  // namespace hk

extern "C" {

int hk_upload_meshes(hk_ctx* c, const HkVertex* v, uint32_t nv, const HkPrimitive* p, uint32_t np, const HkNode* n, uint32_t nn) {
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
  HK_REQUIRE(c && v && p && n && nv && np && nn, HK_E_INVALID, "NULL or empty mesh buffers");
  c->vertices.assign(v, v + nv);
  c->primitives.assign(p, p + np);
  c->asset_nodes.assign(n, n + nn);
  c->have_meshes = true;
  c->mesh_dirty = true;
  return HK_OK;
}
int hk_upload_materials(hk_ctx* c, const HkMaterial* m, uint32_t n) {
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
  HK_REQUIRE(c && m && n, HK_E_INVALID, "NULL or empty material buffer");
  c->materials.assign(m, m + n);
  c->have_materials = true;
  c->dynamic_dirty = true;
  return HK_OK;
}
int hk_upload_instances(hk_ctx* c, const HkInstance* inst, uint32_t ni, const HkNode* inodes, uint32_t nin, const HkEmissive* em, uint32_t ne,
                        const HkNode* enodes, uint32_t nen, const HkAliasEntry* alias, uint32_t na) {
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
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
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
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
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
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
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
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
int hk_upload_textures(hk_ctx* c, const HkImageDesc* images, uint32_t n) {
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
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
  if (c) c->scene_epoch += 1;   // (scene memory is written: hk_context.hpp, primary-ray pipelining)
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
}  // extern "C"
