// hk_wide.hpp - the wide walk (round 4; DESIGN 4 "Wide walk", 8.1): BVH traversal over 128-B records of an inner node's four
// grandchildren (hk_kernels.hpp WideTrees, built by kernels_wavefront.hip k_build_wide), nearest child first, with a per-lane stack.
// Shared by the queue-based trace stage (k_wf_trace_wide) and the fused kernels' walks of scenes in global memory
// (traverse_top_wide).  Same candidates and the same per-triangle arithmetic on the same operands as traverse_top (hk_device.hpp),
// and (round 5) the reference's own rule for two candidates at exactly the same distance (wide_tie_goes_to: the leaf the reference's
// walk meets first): the closest hit is the reference's except for box culls that depend on the visit order (a leaf box grazed within rounding is tested against an
// older, larger bound here than in the reference's walk: measured <= 1 primary hit per 8.3 M pixels, 6.3e-5 relative L2 over 32 frames); an any-hit ray's outcome - occluded or not - does not depend on the order
// (WHICH occluder it reports does: rays whose occluder is kept walk the reference's order, hk_device.hpp traverse_top<true>).
// The reference's order (HK_CTX_EXACT_TRAVERSAL) never comes here.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "hk_device.hpp"
#include "hk_kernels.hpp"

namespace hkd {

#ifndef HK_WIDE_LDS_STACK
#define HK_WIDE_LDS_STACK 28u  // entries of a lane's stack that live in LDS (28 KB per workgroup: five workgroups per CU next to the sharing tables)
#endif
#ifndef HK_WIDE_SPILL
#define HK_WIDE_SPILL 96u
#endif

constexpr uint32_t WIDE_NONE = 0xFFFFFFFFu;      // no child / nothing to visit
constexpr uint32_t WIDE_LEAVE = 0xFFFFFFFEu;     // stack marker: the mesh tree below this entry is done, back to the instance tree
enum : uint32_t { PH_IDLE = 0u, PH_NODE = 1u, PH_TRI = 2u, PH_ENTRY = 3u };  // what a lane of a walk waits for (kernels_wavefront.hip)

struct WideWalk {
  f3 origin, direction, inv_direction;  // the world-space ray
  float early_distance;
  uint32_t exclude_instance;
  Hit hit;
  float limit;         // a distance the closest hit is known not to exceed (another lane's piece of the same ray found one there: k_wf_trace_wide)
  uint32_t cur;        // record to visit next (a slot of the current level's array), or WIDE_NONE: pop
  uint32_t sp;         // the stack holds entries [base, sp)
  uint32_t base;       // (> 0 once idle lanes of the wave took pending subtrees off the BOTTOM of this stack: k_wf_trace_wide)
  uint32_t mark;       // inside a mesh tree: where its WIDE_LEAVE marker sits - the instance-tree entries are [base, mark)
  uint32_t blas_base;  // ... and the mesh tree's live entries are [blas_base, sp) (mark + 1 until entries were handed over)
  uint32_t mesh_base;  // slot of the current mesh tree's first node in WideTrees::blas
  uint32_t prim_base, cur_instance;
  bool in_blas, intersected;
  f3 co, cinv, ld;     // origin / inverse direction of the level being walked, local direction inside a mesh tree
};
// Where a lane's pending children wait.  The first HK_WIDE_LDS_STACK entries: LDS, entry-major (address = entry x 256 + thread: no bank
// conflicts).  Beyond: a global spill area indexed by the lane of a PERSISTENT launch (the trace kernel), or a small private array
// (the fused kernels: their grids are as large as the image).  Entries beyond both are dropped and COUNTED (HkStats::wide_stack_lost):
// 124 entries serve trees some eighty levels deep.
struct WideStackSpill {
  uint32_t* lds;
  uint32_t* spill;
  size_t stride, lane;  // spill[(entry - HK_WIDE_LDS_STACK) x stride + lane]
  unsigned long long* lost;
  __device__ __forceinline__ void put(uint32_t at, uint32_t e) {
    if (at < HK_WIDE_LDS_STACK) lds[at * 256u + threadIdx.x] = e;
    else if (at < HK_WIDE_LDS_STACK + HK_WIDE_SPILL) spill[(size_t)(at - HK_WIDE_LDS_STACK) * stride + lane] = e;
    else if (e != WIDE_NONE) atomicAdd(lost, 1ull);
  }
  __device__ __forceinline__ uint32_t get(uint32_t at) const {
    if (at < HK_WIDE_LDS_STACK) return lds[at * 256u + threadIdx.x];
    if (at < HK_WIDE_LDS_STACK + HK_WIDE_SPILL) return spill[(size_t)(at - HK_WIDE_LDS_STACK) * stride + lane];
    return WIDE_NONE;
  }
};
template <uint32_t LDS_ENTRIES, uint32_t PRIVATE_ENTRIES>
struct WideStackPrivate {
  uint32_t* lds;
  unsigned long long* lost;
  uint32_t priv[PRIVATE_ENTRIES];
  __device__ __forceinline__ void put(uint32_t at, uint32_t e) {
    if (at < LDS_ENTRIES) lds[at * 256u + threadIdx.x] = e;
    else if (at < LDS_ENTRIES + PRIVATE_ENTRIES) priv[at - LDS_ENTRIES] = e;
    else if (e != WIDE_NONE) atomicAdd(lost, 1ull);
  }
  __device__ __forceinline__ uint32_t get(uint32_t at) const {
    if (at < LDS_ENTRIES) return lds[at * 256u + threadIdx.x];
    if (at < LDS_ENTRIES + PRIVATE_ENTRIES) return priv[at - LDS_ENTRIES];
    return WIDE_NONE;
  }
};
template <class S>
__device__ __forceinline__ void wide_push(WideWalk& k, S& st, uint32_t e) {
  st.put(k.sp, e);
  k.sp += 1u;
}
template <class S>
__device__ __forceinline__ uint32_t wide_pop(WideWalk& k, S& st) {
  k.sp -= 1u;
  return st.get(k.sp);
}
__device__ __forceinline__ void wide_begin(WideWalk& k, const WideTrees& wt, f3 origin, f3 direction, float max_distance, float early_distance, uint32_t exclude) {
  k.origin = origin;
  k.direction = direction;
  k.inv_direction = 1.0f / direction;
  k.early_distance = early_distance;
  k.exclude_instance = exclude;
  k.hit.uv = F2(0.0f, 0.0f);
  k.hit.distance = max_distance;
  k.hit.instance_index = HK_U32_MAX;
  k.hit.primitive_index = HK_U32_MAX;
  k.limit = HK_F32_MAX;
  k.cur = wt.tlas_count - 1u;  // the root's record
  k.sp = 0u;
  k.base = 0u;
  k.mark = 0u;
  k.blas_base = 0u;
  k.mesh_base = 0u;
  k.prim_base = 0u;
  k.cur_instance = 0u;
  k.in_blas = false;
  k.intersected = false;
  k.co = origin;
  k.cinv = k.inv_direction;
  k.ld = direction;
}
// A NODE lane with nothing to visit takes the next entry off its stack.  Returns the lane's phase: PH_IDLE (the walk has ended),
// PH_TRI / PH_ENTRY (a leaf: parked at `pending`), PH_NODE - with k.cur set (a record to fetch) or still WIDE_NONE (the entry only
// changed the walk's state: the mesh tree's end marker, a tombstone, the excluded instance).
template <class S>
__device__ __forceinline__ uint32_t wide_pop_next(WideWalk& k, S& st, uint32_t& pending) {
  if (k.sp == k.base) return PH_IDLE;
  const uint32_t e = wide_pop(k, st);
  if (e == WIDE_LEAVE) {  // traverse_bottom returned, light.wgsl:465-470
    if (k.intersected) {
      k.hit.instance_index = k.cur_instance;
      if (k.hit.distance < k.early_distance) return PH_IDLE;
    }
    k.in_blas = false;
    k.co = k.origin;
    k.cinv = k.inv_direction;
    return PH_NODE;
  }
  if (e == WIDE_NONE) return PH_NODE;
  if (e >= HK_LEAF) {
    pending = e - HK_LEAF;
    if (k.in_blas) return PH_TRI;
    return pending != k.exclude_instance ? PH_ENTRY : PH_NODE;
  }
  k.cur = e;
  return PH_NODE;
}
__device__ __forceinline__ const float4* wide_record(const WideWalk& k, const WideTrees& wt) {  // the record k.cur names
  return k.in_blas ? wt.blas + 8u * (size_t)(k.mesh_base + k.cur) : wt.tlas + 8u * (size_t)k.cur;
}
// the record's four boxes against the ray: nearest child next, the others onto the stack (farthest first)
template <class S>
__device__ __forceinline__ uint32_t wide_node_test(WideWalk& k, const float4 (&r8)[8], S& st, uint32_t& pending) {
  float t[4];
  uint32_t link[4];
  const float bound = fmin_(k.hit.distance, k.limit);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 lo = r8[2 * c], hi = r8[2 * c + 1];
    const f3 t1 = (xyz(lo) - k.co) * k.cinv;  // intersects_aabb, light.wgsl:344-362
    const f3 t2 = (xyz(hi) - k.co) * k.cinv;
    float t_min = fmin_(t1.x, t2.x);
    float t_max = fmax_(t1.x, t2.x);
    t_min = fmax_(t_min, fmin_(t1.y, t2.y));
    t_max = fmin_(t_max, fmax_(t1.y, t2.y));
    t_min = fmax_(t_min, fmin_(t1.z, t2.z));
    t_max = fmin_(t_max, fmax_(t1.z, t2.z));
    const float t_box = (t_max >= t_min && t_max >= 0.0f) ? t_min : HK_F32_MAX;
    link[c] = f2u(lo.w);
    // (<=, not the reference's <: a candidate that TIES with the closest hit so far is still tested - wide_triangle decides ties by
    // a rule that does not depend on the order of the visits)
    t[c] = (link[c] != WIDE_NONE && t_box <= bound) ? t_box : HK_F32_MAX;
    if (t[c] == HK_F32_MAX) link[c] = WIDE_NONE;
  }
  // nearest first: a 5-comparator network on (t, link), then the three farther ones go to the stack, farthest first
#define HK_WIDE_CSWAP(a, b)                                  \
  if (t[b] < t[a]) {                                         \
    const float tt = t[a]; t[a] = t[b]; t[b] = tt;           \
    const uint32_t ll = link[a]; link[a] = link[b]; link[b] = ll; \
  }
  HK_WIDE_CSWAP(0, 1) HK_WIDE_CSWAP(2, 3) HK_WIDE_CSWAP(0, 2) HK_WIDE_CSWAP(1, 3) HK_WIDE_CSWAP(1, 2)
#undef HK_WIDE_CSWAP
#pragma unroll
  for (int c = 3; c >= 1; --c)
    if (link[c] != WIDE_NONE) wide_push(k, st, link[c]);
  k.cur = WIDE_NONE;
  if (link[0] == WIDE_NONE) return PH_NODE;  // nothing hit: pop next turn
  if (link[0] >= HK_LEAF) {
    pending = link[0] - HK_LEAF;
    if (k.in_blas) return PH_TRI;
    return pending != k.exclude_instance ? PH_ENTRY : PH_NODE;
  }
  k.cur = link[0];
  return PH_NODE;
}
// One step: pop (if there is nothing to visit) and / or visit one record.  Returns the lane's next phase; `pending` = the leaf a
// PH_TRI / PH_ENTRY lane is parked at.
// (COUNT: rc->nodes / rc->top_nodes count the records actually FETCHED - a turn that only pops fetches nothing)
template <class S, bool COUNT = false>
__device__ __forceinline__ uint32_t wide_node(WideWalk& k, const WideTrees& wt, S& st, uint32_t& pending, RayCounters* rc = nullptr) {
  if (k.cur == WIDE_NONE) {
    const uint32_t ph = wide_pop_next(k, st, pending);
    if (ph != PH_NODE || k.cur == WIDE_NONE) return ph;
  }
  const float4* __restrict__ rec = wide_record(k, wt);
  if (COUNT) {
    rc->nodes++;
    rc->top_nodes += k.in_blas ? 0u : 1u;
  }
  // The eight 16-B loads of the record are ISSUED TOGETHER - one round trip to the memory system per record, not four.  Left to
  // itself the scheduler sometimes sinks each pair next to its slab test to save registers (round 5: the build with the rank
  // loads of wide_tie_goes_to did, and the trace stages of configs 3 / 4 got 11 % / 18 % slower - profiles/r05_rank_rule_ab.txt);
  // the barrier keeps the batch whatever else changes in the kernel.
  float4 r8[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) r8[c] = rec[c];
  __builtin_amdgcn_sched_barrier(0);
  return wide_node_test(k, r8, st, pending);
}
// the reference's tie rule between two candidates at exactly the same distance: the one its stackless walk meets first wins, i.e. the
// leaf of smaller position in ordering 0 - instance leaves first, triangle leaves inside one instance (hk_kernels.hpp WideTrees ranks)
#ifndef HK_WIDE_TIE_BY_RANK
#define HK_WIDE_TIE_BY_RANK 1  // (0: round 4's rule - the smaller (instance, primitive); the A/B of what the rank loads cost)
#endif
__device__ __forceinline__ bool wide_tie_goes_to(const WideTrees& wt, uint32_t instance, uint32_t primitive, uint32_t best_instance, uint32_t best_primitive) {
#if HK_WIDE_TIE_BY_RANK
  if (instance != best_instance) return wt.tlas_rank[instance] < wt.tlas_rank[best_instance];
  return wt.blas_rank[primitive] < wt.blas_rank[best_primitive];
#else
  return instance < best_instance || (instance == best_instance && primitive < best_primitive);
#endif
}
__device__ __forceinline__ uint32_t wide_triangle_test(WideWalk& k, const WideTrees& wt, uint32_t primitive_index, f3 v0, f3 v1, f3 v2) {
  Ray lr;
  lr.origin = k.co;
  lr.direction = k.ld;
  lr.inv_direction = k.cinv;
  f2 uv;
  const float d = intersects_triangle(lr, v0, v1, v2, &uv);
  // closest hit; of two candidates at EXACTLY the same distance the reference keeps the first its walk meets (light.wgsl:415-424) -
  // the leaf of smaller rank, whichever this walk met first: the result is the reference's, and it depends neither on the order of
  // the visits, nor on how the walk was split among lanes, nor on timing
  bool closer = d < k.hit.distance;
  if (d == k.hit.distance && k.hit.primitive_index != HK_U32_MAX) {
    const uint32_t best_instance = k.intersected ? k.cur_instance : k.hit.instance_index;
    closer = wide_tie_goes_to(wt, k.cur_instance, primitive_index, best_instance, k.hit.primitive_index);
  }
  if (closer) {
    k.hit.uv = uv;
    k.hit.distance = d;
    k.hit.primitive_index = primitive_index;
    k.intersected = true;
    if (d < k.early_distance) {  // light.wgsl:421-423 then 466-469
      k.hit.instance_index = k.cur_instance;
      return PH_IDLE;
    }
  }
  return PH_NODE;
}
__device__ __forceinline__ uint32_t wide_triangle(WideWalk& k, const DScene& sc, const WideTrees& wt, uint32_t pending) {
  const uint32_t primitive_index = k.prim_base + pending;
  return wide_triangle_test(k, wt, primitive_index, xyz(sc.tri_v0[primitive_index]), xyz(sc.tri_v1[primitive_index]), xyz(sc.tri_v2[primitive_index]));
}
// entering an instance (light.wgsl:458-464) from the pieces of its record the walk needs: the inverse model's columns, the mesh's
// (material, vertex, primitive, node_offset) and its node count
template <class S>
__device__ __forceinline__ void wide_enter_apply(WideWalk& k, S& st, uint32_t instance_index, float4 im0, float4 im1, float4 im2, float4 im3, uint32_t primitive,
                                                  uint32_t node_offset, uint32_t node_count) {
  const f4 q = mul(im0, im1, im2, im3, F4(k.origin, 1.0f));  // world_to_local_position / _direction (hk_device.hpp), on the loaded columns
  k.co = xyz(q) / q.w;
  k.ld = xyz(mul(im0, im1, im2, im3, F4(k.direction, 0.0f)));
  k.cinv = 1.0f / k.ld;
  k.mark = k.sp;
  wide_push(k, st, WIDE_LEAVE);
  k.blas_base = k.sp;
  k.mesh_base = node_offset;
  k.cur = node_count - 1u;  // the mesh tree's root record
  k.prim_base = primitive;
  k.cur_instance = instance_index;
  k.in_blas = true;
  k.intersected = false;
}
template <class S>
__device__ __forceinline__ void wide_enter(WideWalk& k, const DScene& sc, S& st, uint32_t instance_index) {
  const DInstance& in = sc.instances[instance_index];
  wide_enter_apply(k, st, instance_index, in.im0, in.im1, in.im2, in.im3, in.primitive, in.node_offset, in.node_count);
}

// The whole walk for one ray per lane, in lock step (the fused kernels): every turn each live lane visits one record or pops; the
// triangle tests and instance entries of the lanes that reached a leaf follow in the same turn.
// (one memory round trip per turn - all three fetch kinds issued from one block - was built, bit-exact and slower:
// profiles/r05_overlapped_turns_ab.txt, profiles/r06_removed_wide_overlapped_turns.patch)
// (COUNT: the kernel's counting instantiation - the product's carries no counter at all; left to the optimiser, the counters of a
// RayCounters whose address is taken stayed in scratch memory, a load / add / store per record in the walk's loop)
template <bool COUNT, class S>
__device__ __forceinline__ Hit traverse_top_wide(const DScene& sc, const WideTrees& wt, const Ray& ray, float max_distance, float early_distance, uint32_t exclude_instance,
                                                  S& st, RayCounters& rc) {
  if (COUNT) rc.tlas++;
  WideWalk k;
  wide_begin(k, wt, ray.origin, ray.direction, max_distance, early_distance, exclude_instance);
  uint32_t phase = PH_NODE, pending = 0u;
  while (phase != PH_IDLE) {
    if (phase == PH_NODE) phase = wide_node<S, COUNT>(k, wt, st, pending, &rc);  // (rc.nodes: records fetched)
    if (phase == PH_TRI) {
      if (COUNT) rc.tris++;
      phase = wide_triangle(k, sc, wt, pending);
    } else if (phase == PH_ENTRY) {
      if (COUNT) rc.entries++;
      wide_enter(k, sc, st, pending);
      phase = PH_NODE;
    }
  }
  return k.hit;
}

}  // namespace hkd
