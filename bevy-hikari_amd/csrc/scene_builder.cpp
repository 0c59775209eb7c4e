// scene_builder.cpp - host-side Prepare-stage work of the path, re-implemented in C++.
//
// Mirrors (reference paths relative to cryscan/bevy-hikari v0.3.15):
//   mesh -> primitives + BLAS      src/mesh_material/mod.rs:379-467  (TryFrom<Mesh> for GpuMesh)
//   flat skip-link node format     src/mesh_material/mod.rs:177-201  (GpuNode::pack) on top of
//                                  the `bvh` crate =0.7.1 (Cargo.toml:21): BVH::build is a
//                                  recursive 6-bucket SAH split on the largest centroid axis,
//                                  flatten_custom a depth-first [navigator L, subtree L,
//                                  navigator R, subtree R] emission with EMPTY leaf boxes.  The
//                                  crate is not vendored in the reference checkout; the algorithm
//                                  is restated from its published source.
//   global buffers + offsets       src/mesh_material/mesh.rs:106-166
//   instance AABB / TLAS / emissive list / alias tables / light BVH
//                                  src/mesh_material/instance.rs:286-428, mod.rs:306-376
//
// Pure host code: no HIP calls, usable without a GPU.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <vector>

#include "hk_internal.hpp"

namespace hk {

namespace {

struct Box {
  float mn[3], mx[3];
  static Box empty() {
    const float inf = std::numeric_limits<float>::infinity();
    return Box{{inf, inf, inf}, {-inf, -inf, -inf}};
  }
  void grow(const float p[3]) {
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(mn[k], p[k]);
      mx[k] = std::max(mx[k], p[k]);
    }
  }
  void join(const Box& o) {
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(mn[k], o.mn[k]);
      mx[k] = std::max(mx[k], o.mx[k]);
    }
  }
  float size(int k) const { return mx[k] - mn[k]; }
  float center(int k) const { return mn[k] + size(k) / 2.0f; }
  float surface_area() const {
    float x = size(0), y = size(1), z = size(2);
    return 2.0f * (x * y + x * z + y * z);
  }
  int largest_axis() const {
    float x = size(0), y = size(1), z = size(2);
    if (x > y && x > z) return 0;
    if (y > z) return 1;
    return 2;
  }
};

// ---- `bvh` 0.7.1: BVHNode::build
struct TreeNode {
  bool leaf;
  uint32_t shape;       // leaf
  Box child_l, child_r; // inner
  int l, r;             // inner: indices into the tree vector
};

int build_recursive(const std::vector<Box>& shapes, const std::vector<uint32_t>& indices, std::vector<TreeNode>& nodes) {
  Box bounds = Box::empty(), centroids = Box::empty();
  for (uint32_t i : indices) {
    bounds.join(shapes[i]);
    float c[3] = {shapes[i].center(0), shapes[i].center(1), shapes[i].center(2)};
    centroids.grow(c);
  }
  if (indices.size() == 1) {
    nodes.push_back(TreeNode{true, indices[0], Box::empty(), Box::empty(), -1, -1});
    return (int)nodes.size() - 1;
  }
  const int node_index = (int)nodes.size();
  nodes.push_back(TreeNode{false, 0, Box::empty(), Box::empty(), -1, -1});

  const int axis = centroids.largest_axis();
  const float axis_size = centroids.mx[axis] - centroids.mn[axis];
  std::vector<uint32_t> left, right;
  Box lbox = Box::empty(), rbox = Box::empty();
  if (axis_size < 0.00001f) {
    // shapes too close together: split the index list in half
    size_t half = indices.size() / 2;
    left.assign(indices.begin(), indices.begin() + half);
    right.assign(indices.begin() + half, indices.end());
    for (uint32_t i : left) lbox.join(shapes[i]);
    for (uint32_t i : right) rbox.join(shapes[i]);
  } else {
    const int NUM_BUCKETS = 6;
    Box bucket_box[NUM_BUCKETS];
    uint32_t bucket_size[NUM_BUCKETS];
    std::vector<uint32_t> assign[NUM_BUCKETS];
    for (int b = 0; b < NUM_BUCKETS; ++b) {
      bucket_box[b] = Box::empty();
      bucket_size[b] = 0;
    }
    for (uint32_t i : indices) {
      float rel = (shapes[i].center(axis) - centroids.mn[axis]) / axis_size;
      int b = (int)(rel * ((float)NUM_BUCKETS - 0.01f));
      b = std::min(std::max(b, 0), NUM_BUCKETS - 1);
      bucket_box[b].join(shapes[i]);
      bucket_size[b] += 1;
      assign[b].push_back(i);
    }
    int min_bucket = 0;
    float min_cost = std::numeric_limits<float>::infinity();
    for (int i = 0; i < NUM_BUCKETS - 1; ++i) {
      Box l = Box::empty(), r = Box::empty();
      uint32_t ln = 0, rn = 0;
      for (int b = 0; b <= i; ++b) { l.join(bucket_box[b]); ln += bucket_size[b]; }
      for (int b = i + 1; b < NUM_BUCKETS; ++b) { r.join(bucket_box[b]); rn += bucket_size[b]; }
      float cost = ((float)ln * l.surface_area() + (float)rn * r.surface_area()) / bounds.surface_area();
      if (cost < min_cost) {
        min_bucket = i;
        min_cost = cost;
        lbox = l;
        rbox = r;
      }
    }
    for (int b = 0; b <= min_bucket; ++b) left.insert(left.end(), assign[b].begin(), assign[b].end());
    for (int b = min_bucket + 1; b < NUM_BUCKETS; ++b) right.insert(right.end(), assign[b].begin(), assign[b].end());
    if (left.empty() || right.empty()) {  // degenerate SAH (NaN costs): fall back to the half split
      left.clear(); right.clear();
      size_t half = indices.size() / 2;
      left.assign(indices.begin(), indices.begin() + half);
      right.assign(indices.begin() + half, indices.end());
      lbox = Box::empty(); rbox = Box::empty();
      for (uint32_t i : left) lbox.join(shapes[i]);
      for (uint32_t i : right) rbox.join(shapes[i]);
    }
  }
  int l = build_recursive(shapes, left, nodes);
  int r = build_recursive(shapes, right, nodes);
  nodes[node_index].child_l = lbox;
  nodes[node_index].child_r = rbox;
  nodes[node_index].l = l;
  nodes[node_index].r = r;
  return node_index;
}

// GpuNode::pack, mod.rs:185-201
HkNode pack_node(const Box& box, uint32_t entry_index, uint32_t exit_index, uint32_t shape_index) {
  HkNode n;
  if (entry_index == 0xFFFFFFFFu) entry_index = shape_index | HK_BVH_LEAF_FLAG;
  for (int k = 0; k < 3; ++k) {
    n.min[k] = box.mn[k];
    n.max[k] = box.mx[k];
  }
  n.entry_index = entry_index;
  n.exit_index = exit_index;
  return n;
}

// `bvh` 0.7.1: BVHNode::flatten_custom / create_flat_branch
uint32_t flatten(const std::vector<TreeNode>& nodes, int idx, std::vector<HkNode>& out, uint32_t next_free);
uint32_t flat_branch(const std::vector<TreeNode>& nodes, const Box& box, int child, std::vector<HkNode>& out, uint32_t next_free) {
  out.push_back(pack_node(Box::empty(), 0, 0, 0));  // dummy, replaced below
  uint32_t after = flatten(nodes, child, out, next_free + 1);
  out[next_free] = pack_node(box, next_free + 1, after, 0xFFFFFFFFu);
  return after;
}
uint32_t flatten(const std::vector<TreeNode>& nodes, int idx, std::vector<HkNode>& out, uint32_t next_free) {
  const TreeNode& n = nodes[idx];
  if (n.leaf) {
    out.push_back(pack_node(Box::empty(), 0xFFFFFFFFu, next_free + 1, n.shape));
    return next_free + 1;
  }
  uint32_t nf = flat_branch(nodes, n.child_l, n.l, out, next_free);
  nf = flat_branch(nodes, n.child_r, n.r, out, nf);
  return nf;
}

}  // namespace

std::vector<HkNode> build_flat_bvh(const std::vector<float>& boxes_min_max /* n x 6 */) {
  const size_t n = boxes_min_max.size() / 6;
  std::vector<HkNode> out;
  if (n == 0) return out;
  std::vector<Box> shapes(n);
  std::vector<uint32_t> indices(n);
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) {
      shapes[i].mn[k] = boxes_min_max[6 * i + k];
      shapes[i].mx[k] = boxes_min_max[6 * i + 3 + k];
    }
    indices[i] = (uint32_t)i;
  }
  std::vector<TreeNode> tree;
  tree.reserve(2 * n);
  build_recursive(shapes, indices, tree);
  out.reserve(3 * n);
  flatten(tree, 0, out, 0);
  return out;
}

// A VALID flat tree over the same shapes without the SAH work: the index list halved recursively, O(n log n) box unions.  It is
// what hk_scene_builder_finish_instances uploads when the device is about to build the real trees (hk_update_scene_instances):
// the right size, every leaf present, correct navigator boxes - a frame rendered from it would still be right, only slower.
static int build_halving(const std::vector<Box>& shapes, uint32_t begin, uint32_t end, std::vector<TreeNode>& nodes, Box* box_out) {
  if (end - begin == 1) {
    nodes.push_back(TreeNode{true, begin, Box::empty(), Box::empty(), -1, -1});
    *box_out = shapes[begin];
    return (int)nodes.size() - 1;
  }
  const int node_index = (int)nodes.size();
  nodes.push_back(TreeNode{false, 0, Box::empty(), Box::empty(), -1, -1});
  const uint32_t mid = begin + (end - begin) / 2;
  Box lb, rb;
  const int l = build_halving(shapes, begin, mid, nodes, &lb);
  const int r = build_halving(shapes, mid, end, nodes, &rb);
  nodes[node_index].child_l = lb;
  nodes[node_index].child_r = rb;
  nodes[node_index].l = l;
  nodes[node_index].r = r;
  *box_out = lb;
  box_out->join(rb);
  return node_index;
}
std::vector<HkNode> build_flat_placeholder(const std::vector<float>& boxes_min_max /* n x 6 */) {
  const size_t n = boxes_min_max.size() / 6;
  std::vector<HkNode> out;
  if (n == 0) return out;
  std::vector<Box> shapes(n);
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      shapes[i].mn[k] = boxes_min_max[6 * i + k];
      shapes[i].mx[k] = boxes_min_max[6 * i + 3 + k];
    }
  std::vector<TreeNode> tree;
  tree.reserve(2 * n);
  Box all;
  build_halving(shapes, 0u, (uint32_t)n, tree, &all);
  out.reserve(3 * n);
  flatten(tree, 0, out, 0);
  return out;
}

// ------------------------------------------------------------------------------------------
struct BuilderMesh {
  std::vector<HkVertex> vertices;
  std::vector<HkPrimitive> primitives;
  std::vector<HkNode> nodes;
  float aabb_center[3], aabb_half[3];
};
struct BuilderInstance {
  uint32_t mesh, material;
  float transform[16];
};

}  // namespace hk

struct hk_scene_builder {
  std::vector<hk::BuilderMesh> meshes;
  std::vector<HkMaterial> materials;
  std::vector<hk::BuilderInstance> instance_decl;
  bool finished = false;
  bool standin_trees = false;             // finished by hk_scene_builder_finish_instances: the two trees are valid stand-ins, not the reference's
  bool meshes_dirty = true;               // the concatenated mesh buffers must be rebuilt
  std::vector<float> finished_transforms;  // transforms at the last finish ...
  std::vector<float> previous_transforms;  // ... and at the one before (PreviousMeshUniform)
  // outputs
  std::vector<HkVertex> vertices;
  std::vector<HkPrimitive> primitives;
  std::vector<HkNode> asset_nodes;
  std::vector<HkMeshIndex> mesh_index;
  std::vector<HkInstance> instances;
  std::vector<HkNode> instance_nodes;
  std::vector<HkEmissive> emissives;
  std::vector<HkNode> emissive_nodes;
  std::vector<HkAliasEntry> alias_table;
};

namespace hk {
namespace {

void mat_point(const float m[16], const float p[3], float out[3]) {  // glam Mat4::transform_point3
  for (int k = 0; k < 3; ++k) out[k] = m[k] * p[0] + m[4 + k] * p[1] + m[8 + k] * p[2] + m[12 + k];
}
void mat_vector(const float m[16], const float p[3], float out[3]) {  // glam Mat4::transform_vector3
  for (int k = 0; k < 3; ++k) out[k] = m[k] * p[0] + m[4 + k] * p[1] + m[8 + k] * p[2];
}
// inverse().transpose() of an (arbitrary) 4x4, evaluated in double and rounded once
bool inverse_transpose(const float m[16], float out[16]) {
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
  double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
  if (det == 0.0) return false;
  double id = 1.0 / det;
  // out = transpose(inverse): out[col*4+row] = inv[row*4+col]
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) out[c * 4 + r] = (float)(inv[r * 4 + c] * id);
  return true;
}

// GpuMesh::transformed_primitive_areas, mod.rs:307-318
std::vector<float> primitive_areas(const BuilderMesh& mesh, const float transform[16]) {
  std::vector<float> areas;
  areas.reserve(mesh.primitives.size());
  for (const HkPrimitive& p : mesh.primitives) {
    float v[3][3];
    for (int k = 0; k < 3; ++k) mat_point(transform, mesh.vertices[p.vertices[k].index].position, v[k]);
    float a[3] = {v[1][0] - v[0][0], v[1][1] - v[0][1], v[1][2] - v[0][2]};
    float b[3] = {v[2][0] - v[0][0], v[2][1] - v[0][1], v[2][2] - v[0][2]};
    float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    areas.push_back(0.5f * fabsf(sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2])));
  }
  return areas;
}

// GpuMesh::build_alias_table, mod.rs:320-376
std::vector<HkAliasEntry> build_alias_table(const std::vector<float>& areas) {
  const size_t n = areas.size();
  std::vector<HkAliasEntry> table;
  if (n == 0) return table;
  float surface_area = 0.0f;
  for (float a : areas) surface_area += a;
  const float mean_area = surface_area / (float)n;
  struct Bucket { size_t id; float prob; };
  std::vector<Bucket> over, under;
  for (size_t i = 0; i < n; ++i) {
    float p = areas[i] / mean_area;
    if (p > 1.0f) over.push_back({i, p});
  }
  for (size_t i = 0; i < n; ++i) {
    float p = areas[i] / mean_area;
    if (p < 1.0f) under.push_back({i, p});
  }
  table.resize(n);
  for (size_t i = 0; i < n; ++i) table[i] = HkAliasEntry{0.0f, (uint32_t)i};
  while (!under.empty() && !over.empty()) {
    Bucket over_bucket = over.back();
    over.pop_back();
    Bucket under_bucket = under.back();
    under.pop_back();
    float delta = 1.0f - under_bucket.prob;
    over_bucket.prob -= delta;
    if (over_bucket.prob > 1.0f)
      over.push_back(over_bucket);
    else if (over_bucket.prob < 1.0f)
      under.push_back(over_bucket);
    table[under_bucket.id] = HkAliasEntry{delta, (uint32_t)over_bucket.id};
  }
  return table;
}

}  // namespace
}  // namespace hk

namespace hk {
bool instance_world_record(const float transform[16], const float aabb_center[3], const float aabb_half[3], float mn_out[3], float mx_out[3], float itm[16]) {
  float center[3];
  mat_point(transform, aabb_center, center);
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};  // seeded at zero, instance.rs:298-299
  for (int corner = 0; corner < 8; ++corner) {
    float e[3] = {aabb_half[0] * (float)(2 * (corner & 1) - 1), aabb_half[1] * (float)(2 * ((corner >> 1) & 1) - 1), aabb_half[2] * (float)(2 * ((corner >> 2) & 1) - 1)};
    float t[3];
    mat_vector(transform, e, t);
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(mn[k], t[k]);
      mx[k] = std::max(mx[k], t[k]);
    }
  }
  for (int k = 0; k < 3; ++k) {
    mn_out[k] = mn[k] + center[k];
    mx_out[k] = mx[k] + center[k];
  }
  return inverse_transpose(transform, itm);
}
bool builder_has_standin_trees(const hk_scene_builder* b) { return b && b->finished && b->standin_trees; }
uint32_t builder_instance_count(const hk_scene_builder* b) { return b ? (uint32_t)b->instance_decl.size() : 0u; }
bool builder_instance_decl(const hk_scene_builder* b, uint32_t i, InstanceDecl* out) {
  // (poses set since the last finish are exactly what the caller is after; the MESH buffers must be the finished ones)
  if (!b || b->meshes_dirty || b->mesh_index.size() != b->meshes.size() || i >= b->instance_decl.size()) return false;
  const BuilderInstance& d = b->instance_decl[i];
  out->mesh = b->mesh_index[d.mesh];
  out->material = d.material;
  out->transform = d.transform;
  out->aabb_center = b->meshes[d.mesh].aabb_center;
  out->aabb_half = b->meshes[d.mesh].aabb_half;
  return true;
}
void builder_commit_transforms(hk_scene_builder* b) {
  std::vector<float> now(b->instance_decl.size() * 16);
  for (size_t i = 0; i < b->instance_decl.size(); ++i) memcpy(&now[16 * i], b->instance_decl[i].transform, 64);
  b->previous_transforms = now;
  const size_t common = std::min(now.size(), b->finished_transforms.size());
  if (common) memcpy(b->previous_transforms.data(), b->finished_transforms.data(), common * sizeof(float));
  b->finished_transforms.swap(now);
}
}  // namespace hk

using namespace hk;

extern "C" {

int hk_scene_builder_create(hk_scene_builder** out) {
  HK_REQUIRE(out, HK_E_INVALID, "out is NULL");
  *out = new (std::nothrow) hk_scene_builder();
  HK_REQUIRE(*out, HK_E_NOMEM, "allocation failed");
  return HK_OK;
}
void hk_scene_builder_destroy(hk_scene_builder* b) { delete b; }

int hk_scene_builder_add_mesh(hk_scene_builder* b, const float* positions, const float* normals, const float* uvs, uint32_t n_vertices,
                              const uint32_t* indices, uint32_t n_indices, uint32_t topology, uint32_t* mesh_id) {
  HK_REQUIRE(b, HK_E_INVALID, "builder is NULL");
  // mod.rs:383-397: position, normal and uv0 are all required
  HK_REQUIRE(positions && normals && uvs && n_vertices > 0, HK_E_INVALID, "mesh needs position, normal and uv attributes");
  HK_REQUIRE(topology == HK_TOPOLOGY_TRIANGLE_LIST || topology == HK_TOPOLOGY_TRIANGLE_STRIP, HK_E_UNSUPPORTED, "incompatible primitive topology");
  BuilderMesh mesh;
  mesh.vertices.resize(n_vertices);
  float mn[3] = {positions[0], positions[1], positions[2]}, mx[3] = {positions[0], positions[1], positions[2]};
  for (uint32_t i = 0; i < n_vertices; ++i) {
    HkVertex& v = mesh.vertices[i];
    for (int k = 0; k < 3; ++k) {
      v.position[k] = positions[3 * i + k];
      v.normal[k] = normals[3 * i + k];
      mn[k] = std::min(mn[k], v.position[k]);
      mx[k] = std::max(mx[k], v.position[k]);
    }
    v.u = uvs[2 * i];
    v.v = uvs[2 * i + 1];
  }
  for (int k = 0; k < 3; ++k) {  // bevy Aabb::from_min_max
    mesh.aabb_center[k] = 0.5f * (mx[k] + mn[k]);
    mesh.aabb_half[k] = 0.5f * (mx[k] - mn[k]);
  }
  std::vector<uint32_t> idx;
  if (indices && n_indices) {
    idx.assign(indices, indices + n_indices);
  } else {  // mod.rs:408-411
    idx.resize(n_vertices);
    for (uint32_t i = 0; i < n_vertices; ++i) idx[i] = i;
  }
  for (uint32_t i : idx) HK_REQUIRE(i < n_vertices, HK_E_INVALID, "vertex index out of range");
  auto push = [&](uint32_t a, uint32_t c, uint32_t d) {
    HkPrimitive p;
    uint32_t id[3] = {a, c, d};
    for (int k = 0; k < 3; ++k) {
      memcpy(p.vertices[k].position, mesh.vertices[id[k]].position, 12);
      p.vertices[k].index = id[k];
    }
    mesh.primitives.push_back(p);
  };
  if (topology == HK_TOPOLOGY_TRIANGLE_LIST) {  // mod.rs:414-431
    HK_REQUIRE(idx.size() % 3 == 0, HK_E_UNSUPPORTED, "incompatible primitive topology (index count not a multiple of 3)");
    for (size_t i = 0; i + 2 < idx.size(); i += 3) push(idx[i], idx[i + 1], idx[i + 2]);
  } else {  // mod.rs:432-449: odd triangles flip winding
    for (size_t i = 0; i + 2 < idx.size(); ++i) {
      if ((i & 1) == 0)
        push(idx[i], idx[i + 1], idx[i + 2]);
      else
        push(idx[i + 1], idx[i], idx[i + 2]);
    }
  }
  HK_REQUIRE(!mesh.primitives.empty(), HK_E_INVALID, "mesh has no primitive");  // mod.rs:454-456
  std::vector<float> boxes;
  boxes.reserve(mesh.primitives.size() * 6);
  for (const HkPrimitive& p : mesh.primitives) {
    Box bx = Box::empty();
    for (int k = 0; k < 3; ++k) bx.grow(p.vertices[k].position);
    boxes.insert(boxes.end(), bx.mn, bx.mn + 3);
    boxes.insert(boxes.end(), bx.mx, bx.mx + 3);
  }
  mesh.nodes = build_flat_bvh(boxes);  // mod.rs:458-459
  b->meshes.push_back(std::move(mesh));
  b->meshes_dirty = true;
  b->finished = false;
  if (mesh_id) *mesh_id = (uint32_t)b->meshes.size() - 1;
  return HK_OK;
}

int hk_scene_builder_add_material(hk_scene_builder* b, const HkMaterial* material, uint32_t* material_id) {
  HK_REQUIRE(b && material, HK_E_INVALID, "bad argument");
  b->materials.push_back(*material);
  b->finished = false;
  if (material_id) *material_id = (uint32_t)b->materials.size() - 1;
  return HK_OK;
}

int hk_scene_builder_add_instance(hk_scene_builder* b, uint32_t mesh_id, uint32_t material_id, const float transform[16], uint32_t* instance_id) {
  HK_REQUIRE(b && transform, HK_E_INVALID, "bad argument");
  HK_REQUIRE(mesh_id < b->meshes.size() && material_id < b->materials.size(), HK_E_INVALID, "unknown mesh or material id");
  BuilderInstance inst;
  inst.mesh = mesh_id;
  inst.material = material_id;
  memcpy(inst.transform, transform, 64);
  b->instance_decl.push_back(inst);
  b->finished = false;
  if (instance_id) *instance_id = (uint32_t)b->instance_decl.size() - 1;
  return HK_OK;
}

int hk_scene_builder_set_instance_transform(hk_scene_builder* b, uint32_t instance_id, const float transform[16]) {
  HK_REQUIRE(b && transform, HK_E_INVALID, "bad argument");
  HK_REQUIRE(instance_id < b->instance_decl.size(), HK_E_INVALID, "unknown instance id");
  memcpy(b->instance_decl[instance_id].transform, transform, 64);
  b->finished = false;
  return HK_OK;
}

static int finish_impl(hk_scene_builder* b, bool build_trees);
int hk_scene_builder_finish(hk_scene_builder* b) { return finish_impl(b, true); }
int hk_scene_builder_finish_instances(hk_scene_builder* b) { return finish_impl(b, false); }

int hk_scene_builder_remove_instance(hk_scene_builder* b, uint32_t instance_id) {
  HK_REQUIRE(b, HK_E_INVALID, "builder is NULL");
  HK_REQUIRE(instance_id < b->instance_decl.size(), HK_E_INVALID, "unknown instance id");
  b->instance_decl.erase(b->instance_decl.begin() + instance_id);
  // the transform history is kept per instance: the rows of the removed one go with it
  for (std::vector<float>* v : {&b->finished_transforms, &b->previous_transforms})
    if (v->size() >= 16 * ((size_t)instance_id + 1)) v->erase(v->begin() + 16 * (size_t)instance_id, v->begin() + 16 * ((size_t)instance_id + 1));
  b->finished = false;
  return HK_OK;
}

int hk_scene_builder_set_instance_material(hk_scene_builder* b, uint32_t instance_id, uint32_t material_id) {
  HK_REQUIRE(b, HK_E_INVALID, "builder is NULL");
  HK_REQUIRE(instance_id < b->instance_decl.size() && material_id < b->materials.size(), HK_E_INVALID, "unknown instance or material id");
  b->instance_decl[instance_id].material = material_id;
  b->finished = false;
  return HK_OK;
}

static int finish_impl(hk_scene_builder* b, bool build_trees) {
  HK_REQUIRE(b, HK_E_INVALID, "builder is NULL");
  if (b->meshes_dirty) {  // mesh.rs:141-163: concatenate, remember offsets (only when a mesh was added)
    b->vertices.clear(); b->primitives.clear(); b->asset_nodes.clear(); b->mesh_index.clear();
    for (const BuilderMesh& m : b->meshes) {
      HkMeshIndex mi;
      mi.vertex = (uint32_t)b->vertices.size();
      mi.primitive = (uint32_t)b->primitives.size();
      mi.node_offset = (uint32_t)b->asset_nodes.size();
      mi.node_count = (uint32_t)m.nodes.size();
      b->vertices.insert(b->vertices.end(), m.vertices.begin(), m.vertices.end());
      b->primitives.insert(b->primitives.end(), m.primitives.begin(), m.primitives.end());
      b->asset_nodes.insert(b->asset_nodes.end(), m.nodes.begin(), m.nodes.end());
      b->mesh_index.push_back(mi);
    }
    b->meshes_dirty = false;
  }
  b->instances.clear(); b->instance_nodes.clear(); b->emissives.clear(); b->emissive_nodes.clear(); b->alias_table.clear();
  builder_commit_transforms(b);  // PreviousMeshUniform: what the transforms were at the last finish (new instances: their own)
  // instance.rs:286-325
  std::vector<float> boxes;
  for (const BuilderInstance& d : b->instance_decl) {
    const BuilderMesh& mesh = b->meshes[d.mesh];
    HkInstance inst;
    memset(&inst, 0, sizeof(inst));
    HK_REQUIRE(instance_world_record(d.transform, mesh.aabb_center, mesh.aabb_half, inst.min, inst.max, inst.inverse_transpose_model), HK_E_INVALID,
               "singular instance transform");
    inst.material = d.material;
    memcpy(inst.model, d.transform, 64);
    inst.mesh = b->mesh_index[d.mesh];
    b->instances.push_back(inst);
    boxes.insert(boxes.end(), inst.min, inst.min + 3);
    boxes.insert(boxes.end(), inst.max, inst.max + 3);
  }
  b->instance_nodes = build_trees ? build_flat_bvh(boxes) : build_flat_placeholder(boxes);  // instance.rs:365-371
  // `BHShape::set_bh_node_index` bookkeeping (unused by the shaders): position of the leaf node
  for (uint32_t n = 0; n < b->instance_nodes.size(); ++n)
    if (b->instance_nodes[n].entry_index >= HK_BVH_LEAF_FLAG) b->instances[b->instance_nodes[n].entry_index - HK_BVH_LEAF_FLAG].node_index = n;

  // instance.rs:380-420: emissive list, alias tables
  std::vector<float> eboxes;
  for (uint32_t id = 0; id < b->instances.size(); ++id) {
    const HkInstance& inst = b->instances[id];
    const HkMaterial& mat = b->materials[inst.material];
    const float* e = mat.emissive;
    float intensity = 255.0f * e[3] * sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (!(intensity > 0.0f)) continue;
    const BuilderMesh& mesh = b->meshes[b->instance_decl[id].mesh];
    std::vector<float> areas = primitive_areas(mesh, inst.model);
    std::vector<HkAliasEntry> table = build_alias_table(areas);
    HkEmissive em;
    memset(&em, 0, sizeof(em));
    memcpy(em.emissive, e, 16);
    float d2 = 0.0f;
    for (int k = 0; k < 3; ++k) {
      em.position[k] = 0.5f * (inst.max[k] + inst.min[k]);
      float d = inst.max[k] - inst.min[k];
      d2 += d * d;
    }
    em.radius = 0.5f * sqrtf(d2) + sqrtf(intensity);
    em.instance = id;
    em.alias_table[0] = (uint32_t)b->alias_table.size();
    em.alias_table[1] = (uint32_t)table.size();
    b->alias_table.insert(b->alias_table.end(), table.begin(), table.end());
    float area = 0.0f;
    for (float a : areas) area += a;
    em.surface_area = area;
    b->emissives.push_back(em);
    for (int k = 0; k < 3; ++k) eboxes.push_back(em.position[k] - em.radius);
    for (int k = 0; k < 3; ++k) eboxes.push_back(em.position[k] + em.radius);
  }
  b->emissive_nodes = build_trees ? build_flat_bvh(eboxes) : build_flat_placeholder(eboxes);  // instance.rs:422-428
  for (uint32_t n = 0; n < b->emissive_nodes.size(); ++n)
    if (b->emissive_nodes[n].entry_index >= HK_BVH_LEAF_FLAG) b->emissives[b->emissive_nodes[n].entry_index - HK_BVH_LEAF_FLAG].node_index = n;
  b->finished = true;
  b->standin_trees = !build_trees && (b->instances.size() > 1 || b->emissives.size() > 1);  // (a tree of one leaf has one shape)
  return HK_OK;
}

#define HK_BUILDER_GETTER(name, field, type)                                                   \
  int hk_scene_builder_##name(const hk_scene_builder* b, const type** p, uint32_t* n) {        \
    HK_REQUIRE(b && p && n, HK_E_INVALID, "bad argument");                                     \
    HK_REQUIRE(b->finished, HK_E_NOT_READY, "call hk_scene_builder_finish first");             \
    *p = b->field.data();                                                                      \
    *n = (uint32_t)b->field.size();                                                            \
    return HK_OK;                                                                              \
  }
HK_BUILDER_GETTER(vertices, vertices, HkVertex)
HK_BUILDER_GETTER(primitives, primitives, HkPrimitive)
HK_BUILDER_GETTER(asset_nodes, asset_nodes, HkNode)
HK_BUILDER_GETTER(materials, materials, HkMaterial)
HK_BUILDER_GETTER(instances, instances, HkInstance)
HK_BUILDER_GETTER(instance_nodes, instance_nodes, HkNode)
HK_BUILDER_GETTER(emissives, emissives, HkEmissive)
HK_BUILDER_GETTER(emissive_nodes, emissive_nodes, HkNode)
HK_BUILDER_GETTER(alias_table, alias_table, HkAliasEntry)

int hk_scene_builder_previous_transforms(const hk_scene_builder* b, const float** p, uint32_t* n) {
  HK_REQUIRE(b && p && n, HK_E_INVALID, "bad argument");
  HK_REQUIRE(b->finished, HK_E_NOT_READY, "call hk_scene_builder_finish first");
  *p = b->previous_transforms.data();
  *n = (uint32_t)(b->previous_transforms.size() / 16);
  return HK_OK;
}

}  // extern "C"
