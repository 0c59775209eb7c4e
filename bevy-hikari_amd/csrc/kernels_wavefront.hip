// kernels_wavefront.hip - indirect_lit_ambient (light.wgsl:1263-1498, MULTIPLE_BOUNCES) as a queue-based schedule.
//
// The fused kernel (kernels.hip k_indirect) keeps one pixel in one lane from the G-buffer read to the reservoir store;
// its BVH walks then run until the slowest lane of the wave is done (Cornell: 28 of 64 lanes live per step; a scene
// of 256 k triangles: walks of hundreds of steps whose lengths differ by 10x inside one 8x8 tile).  Here the same
// arithmetic per pixel, per ray and per bounce is cut at the walks:
//
//   k_wf_setup   pixel-mapped like k_indirect: background pixels are finished on the spot (zero reservoirs, the
//                uniform-tile store elision); every other pixel gets a path SLOT (wave ballot + prefix count, one atomic
//                per wave), its path state and its first bounce ray
//   k_wf_trace   persistent waves; a lane whose ray has ended takes the next ray of the stage's queue (wave ballot of
//                the idle lanes + mbcnt prefix into a 64-entry block the wave reserved with ONE atomic), so a wave
//                keeps walking with full lanes until the queue is dry - the active-ray compaction across bounces
//                that BASELINE.json's north star names, applied to the whole frame instead of one tile
//   k_wf_shade   bounce n of every live path (hit attributes, light candidate, both outcomes of the shadow ray,
//                throughput, next bounce ray); emits the shadow ray of bounce n and the closest-hit ray of bounce n+1
//                - they are independent, so ONE trace launch walks both lists - and compacts the surviving paths
//   k_wf_final   adds the last shadow result and runs the temporal-reuse tail (hk_light.hpp indirect_temporal_tail)
//
// Launches for N bounces: setup, N x (trace, shade), trace, final.  What a path carries from stage to stage lives in
// 16-B planes indexed by slot (coalesced dwordx4 accesses): 9 planes of path state, 2 + 3 of rays, 2 + 1 of hits.
//
// Bit-exactness.  A shadow ray's outcome only selects between two radiance values: `occlude_hit_info` followed by
// `input_radiance` (light.wgsl:526-533,835-867) gives (0,0,0,1) for EVERY occluder (the sampled emitter is excluded from
// the walk, so the occluder never is the emitter) and leaves the candidate's own hit info untouched otherwise.  The
// shade stage therefore evaluates the reference's expression for both outcomes before the ray is traced and the next
// stage adds the one the walk selected - in bounce order, so every float addition happens in the reference's order.
// Everything else is the fused kernel's code on the same operands.  tests/test_parity_gpu.py compares the two schedules
// buffer by buffer, and each against the oracle.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

#include "hk_device.hpp"
#include "hk_kernels.hpp"
#include "hk_light.hpp"
#include "hk_prepass.hpp"
#include "hk_wide.hpp"

namespace hkd {

namespace {
constexpr uint32_t WF_SHADOW = 0x80000000u;
// counters (WfBuffers::ctr): per stage s
constexpr uint32_t WF_ALIVE = 0u;    // [s] paths alive at bounce s (s = 0: all slots)
constexpr uint32_t WF_SHADOWS = 64u;  // [s] shadow rays trace stage s walks (emitted by the shade stage of bounce s - 1)
constexpr uint32_t WF_QHEAD = 128u;   // [s] rays of trace stage s claimed so far (its queue = the alive list, then the shadow list)

#ifndef HK_WF_BLOCK_SMALL
#define HK_WF_BLOCK_SMALL 64u  // rays a wave reserves at a time once the queue is within four rounds of its end (256 before)
#endif
#ifndef HK_WF_REFILL_MIN
#define HK_WF_REFILL_MIN 8u   // a wave fetches new rays once this many lanes are idle (or all of them)
#endif
#ifndef HK_WF_TRACE_WAVES
#define HK_WF_TRACE_WAVES 7   // waves per SIMD the trace kernel is compiled for (69 VGPRs, no spills; 8 would spill 44 B per lane)
#endif
#ifndef HK_WF_TRACE_WG_PER_CU
#define HK_WF_TRACE_WG_PER_CU 8  // 256-thread workgroups of the persistent trace launch per CU (7 are resident at 69 VGPRs)
#endif
#ifndef HK_WF_STEPS
#define HK_WF_STEPS 4         // node steps per turn of the node phase
#endif
#ifndef HK_WF_NODE_WEIGHT
#define HK_WF_NODE_WEIGHT 1u  // the node phase runs while (lanes at nodes) x weight >= lanes parked at triangles / instance entries
#endif

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {  // set bits of `mask` below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// One atomic per WORKGROUP and list (a single hot counter takes ~10 ns per atomic on this chip: one per wave - 32 k per
// dispatch at 1080p - was a third of the shade stage).  Called by all 256 threads; lds: 6 words.
__device__ __forceinline__ uint32_t block_push(uint32_t* counter, bool want, uint32_t* lds) {
  const unsigned long long m = __ballot(want);
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0u) lds[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0u) {
    const uint32_t total = lds[0] + lds[1] + lds[2] + lds[3];
    lds[4] = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  uint32_t base = lds[4];
  for (uint32_t k = 0; k < wave; ++k) base += lds[k];
  __syncthreads();  // lds is reused by the next call
  return base + lane_rank(m);
}

// path state planes (WfBuffers::state + k * cap)
enum : uint32_t { PL_RANDOM = 0, PL_POSITION_PDF, PL_NORMAL_PENDING, PL_TRANSPORT, PL_RADIANCE, PL_FIRST_POSITION, PL_FIRST_NORMAL, PL_ADD_CLEAR, PL_ADD_OCCLUDED };
__device__ __forceinline__ float4* plane(const WfBuffers& w, uint32_t k) { return w.state + (size_t)k * w.cap; }

// the next bounce ray of a path: light.wgsl:1316-1321 (sample_cosine_hemisphere through the normal's basis, biased origin)
__device__ __forceinline__ void emit_bounce_ray(const WfBuffers& w, uint32_t slot, f4 random, f3 position, f3 normal) {
  const f4 rand_sample = sample_cosine_hemisphere(F2(random.x, random.y));
  const f3 origin = position + normal * HK_RAY_BIAS;
  const f3 direction = mul(normal_basis(normal), xyz(rand_sample));
  w.cr0[slot] = make_float4(origin.x, origin.y, origin.z, 0.0f);
  w.cr1[slot] = make_float4(direction.x, direction.y, direction.z, rand_sample.w);
}
}  // namespace

// ------------------------------------------------------------------ setup: background pixels, slots, first rays
__global__ __launch_bounds__(256) void k_wf_setup(DScene sc, DFrame fr, GBuffer g, LightTargets t, WfBuffers w, int row_begin, int row_end) {
  const Pixel px = pixel_of_thread<false>(fr.rw, row_begin, row_end);
  bool path = false;
  int index = 0;
  f4 random = F4(0, 0, 0, 0);
  f3 position = F3(0, 0, 0), normal = F3(0, 0, 0);
  if (px.valid) {  // the prologue of k_indirect, light.wgsl:1263-1308
    const int x = px.x, y = px.y;
    index = x + fr.rw * y;
    const f2 uv = coords_to_uv(fr, x, y);
    int dcx, dcy;
    jittered_deferred_coords(fr, uv, &dcx, &dcy);
    const bool din = in_bounds(dcx, dcy, fr.dw, fr.dh);
    const int didx = dcx + fr.dw * dcy;
    const float4 position_depth = din ? g.position[didx] : make_float4(0, 0, 0, 0);
    const float depth = position_depth.w;
    const bool background = fr.indirect_bounces == 0u || depth < HK_F32_EPSILON;
    const bool all_background = __ballot(!background) == 0ull;  // among the wave's valid pixels
    const int tile = wave_tile(px, t.tiles_x);
    if (background) {  // light.wgsl:1279-1287
      const PackedReservoir pr = pack_reservoir(zero_reservoir());
      bool skip_current = false, skip_spatial = false, skip_previous_spatial = false;
      if (t.m_current && all_background) {  // uniform-tile store elision (hk_kernels.hpp TileMeta)
        const unsigned long long id = record_id(pr);
        skip_current = tile_holds(t.m_current, tile, id);
        skip_spatial = tile_holds(t.m_spatial, tile, id);
        skip_previous_spatial = tile_holds(t.m_previous_spatial, tile, id);
        if (!skip_current) tile_mark(t.m_current, tile, id, 0ull, t.serial);
        if (!skip_spatial) tile_mark(t.m_spatial, tile, id, 0ull, t.serial);
        if (!skip_previous_spatial) tile_mark(t.m_previous_spatial, tile, id, 0ull, t.serial);
      }
      if (!skip_current) store_packed(t.current, index, pr);
      if (!skip_spatial) store_packed(t.spatial, index, pr);
      if (!skip_previous_spatial) store_previous_spatial(t, index, index, pr);
      t.variance[index] = 0.0f;
      t.render[index] = make_uint2(0u, 0u);
    } else {
      normal = normalize(xyz(unpack4x8snorm(g.normal[didx])));
      random = noise_fetch(sc, x, y, fr.number);
      random = fract(random + fr.number_golden);
      position = xyz(position_depth);
      path = true;
    }
    if (t.m_current && !all_background) {  // a tile with real pixels: no longer one record everywhere
      tile_unknown(t.m_current, tile);
      tile_unknown(t.m_previous_spatial, tile);  // (geometry pixels may store into their own slot of it: k_direct_lit has the long version)
      if (__ballot(background) != 0ull) tile_unknown(t.m_spatial, tile);
    }
  }
  __shared__ uint32_t push_lds[6];
  const uint32_t slot = block_push(&w.ctr[WF_ALIVE], path, push_lds);
  if (path) {
    w.pixel[slot] = (uint32_t)index;
    w.alive[0][slot] = slot;  // bounce 0: every slot is alive
    plane(w, PL_RANDOM)[slot] = to_float4(random);
    plane(w, PL_POSITION_PDF)[slot] = make_float4(position.x, position.y, position.z, 0.0f);
    plane(w, PL_NORMAL_PENDING)[slot] = make_float4(normal.x, normal.y, normal.z, 0.0f);
    // (transport = 1, radiance = 0: the shade stage of bounce 0 starts from these values without reading their planes)
    emit_bounce_ray(w, slot, random, position, normal);
  }
}

// ------------------------------------------------------------------ trace: persistent waves, lanes refill from the queue
// traverse_top (hk_device.hpp, light.wgsl:400-486) as a resumable state: begin() = its prologue, step() = one iteration
// of its loop; a lane's sequence of steps, and with it every result bit, is that of the fused kernels.
namespace {
struct Walk {
  f3 origin, direction, inv_direction;  // the world-space ray
  float early_distance;
  uint32_t exclude_instance;
  Hit hit;
  uint32_t tlas_base, index, limit, base, t_resume, prim_base, cur_instance;
  bool in_blas, intersected;
  f3 co, cinv, ld;  // origin / inverse direction of the level being walked, local direction inside a BLAS
};
__device__ __forceinline__ void walk_begin(Walk& k, const DScene& sc, f3 origin, f3 direction, float max_distance, float early_distance, uint32_t exclude) {
  k.origin = origin;
  k.direction = direction;
  k.inv_direction = 1.0f / direction;
  k.early_distance = early_distance;
  k.exclude_instance = exclude;
  k.hit.uv = F2(0.0f, 0.0f);
  k.hit.distance = max_distance;
  k.hit.instance_index = HK_U32_MAX;
  k.hit.primitive_index = HK_U32_MAX;
  k.tlas_base = ray_octant(direction) * sc.tlas_stride;
  k.index = 0u;
  k.limit = sc.tlas_count;
  k.base = k.tlas_base;
  k.t_resume = 0u;
  k.prim_base = 0u;
  k.cur_instance = 0u;
  k.in_blas = false;
  k.intersected = false;
  k.co = k.origin;
  k.cinv = k.inv_direction;
  k.ld = direction;
}
// One step of the walk is one of three things: a NODE step (two 16-B loads, slab test, skip-link select: ~35 instructions, every
// live lane takes one per iteration in the fused kernels), a TRIANGLE test (~60) or an instance ENTRY (ray transform, ~100).
// In lock step the last two run whenever ANY lane needs them - with 64 live lanes that is every iteration, at 2-5 live lanes -
// so refilled lanes alone leave the wave at ~16 % lane utilisation (measured, config 3).  Here a lane that reaches a hit leaf
// PARKS with the leaf in `pending`; the wave runs whichever of the three phases has the most lanes waiting, so triangle tests
// and entries execute with many lanes at once.  A lane's own sequence of steps - and with it every result bit - is unchanged.
// (the phases PH_IDLE / PH_NODE / PH_TRI / PH_ENTRY: hk_wide.hpp)
// NODE step; returns the lane's next phase (PH_IDLE: the walk has ended, k.hit is the result)
__device__ __forceinline__ uint32_t walk_node(Walk& k, const DScene& sc, uint32_t& pending) {
  if (k.index >= k.limit) {
    if (!k.in_blas) return PH_IDLE;
    if (k.intersected) {  // traverse_bottom returned, light.wgsl:465-470
      k.hit.instance_index = k.cur_instance;
      if (k.hit.distance < k.early_distance) return PH_IDLE;
    }
    k.in_blas = false;
    k.index = k.t_resume;
    k.limit = sc.tlas_count;
    k.base = k.tlas_base;
    k.co = k.origin;
    k.cinv = k.inv_direction;
    return PH_NODE;
  }
  const float4* __restrict__ nd = sc.nodes + 2u * (k.base + k.index);
  const float4 lo = nd[0];
  const float4 hi = nd[1];
  const uint32_t entry = f2u(lo.w), exit_ = f2u(hi.w);
  const f3 t1 = (xyz(lo) - k.co) * k.cinv;  // intersects_aabb, light.wgsl:344-362
  const f3 t2 = (xyz(hi) - k.co) * k.cinv;
  float t_min = fmin_(t1.x, t2.x);
  float t_max = fmax_(t1.x, t2.x);
  t_min = fmax_(t_min, fmin_(t1.y, t2.y));
  t_max = fmin_(t_max, fmax_(t1.y, t2.y));
  t_min = fmax_(t_min, fmin_(t1.z, t2.z));
  t_max = fmin_(t_max, fmax_(t1.z, t2.z));
  const float t_box = (t_max >= t_min && t_max >= 0.0f) ? t_min : HK_F32_MAX;
  const bool box_hit = t_box < k.hit.distance;
  const bool leaf = entry >= HK_LEAF;
  k.index = (leaf || !box_hit) ? exit_ : entry;
  if (leaf && box_hit) {
    pending = entry - HK_LEAF;
    if (k.in_blas) return PH_TRI;
    if (pending != k.exclude_instance) return PH_ENTRY;
  }
  return PH_NODE;
}
// TRIANGLE test of the parked leaf; PH_IDLE when the hit ends the walk (any-hit early out)
__device__ __forceinline__ uint32_t walk_triangle(Walk& k, const DScene& sc, uint32_t pending) {
  const uint32_t primitive_index = k.prim_base + pending;
  Ray lr;
  lr.origin = k.co;
  lr.direction = k.ld;
  lr.inv_direction = k.cinv;
  f2 uv;
  const float d = intersects_triangle(lr, xyz(sc.tri_v0[primitive_index]), xyz(sc.tri_v1[primitive_index]), xyz(sc.tri_v2[primitive_index]), &uv);
  if (d < k.hit.distance) {
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
// ENTRY into the parked instance (light.wgsl:458-464)
__device__ __forceinline__ void walk_enter(Walk& k, const DScene& sc, uint32_t instance_index) {
  const DInstance& in = sc.instances[instance_index];
  k.co = world_to_local_position(in, k.origin);
  k.ld = world_to_local_direction(in, k.direction);
  k.cinv = 1.0f / k.ld;
  k.t_resume = k.index;
  k.base = sc.blas_base + ray_octant(k.ld) * sc.blas_stride + in.node_offset;
  k.index = 0u;
  k.limit = in.node_count;
  k.prim_base = in.primitive;
  k.cur_instance = instance_index;
  k.in_blas = true;
  k.intersected = false;
}
}  // namespace

// Stage `stage` walks the closest-hit rays of the paths alive at bounce `stage` and the shadow rays the shade stage of bounce
// `stage - 1` emitted: queue entry i is alive[i] for i < n_alive, else shadow[i - n_alive].
// TL: the instrumented twin for tools/wf_timeline.py (WfBuffers::timeline) - when the queue ran dry, when the last wave left, how
// long the rays were; the product launches TL = false.
template <bool LDS, bool TL>
__global__ __launch_bounds__(256, HK_WF_TRACE_WAVES) void k_wf_trace(DScene gsc, WfBuffers w, uint32_t stage) {
  const DScene sc = stage_scene<LDS>(gsc);
  __shared__ uint32_t tl_hist[16];
  unsigned long long tl_start = 0ull;
  uint32_t tl_steps = 0u, tl_max = 0u, tl_sum = 0u, tl_rays = 0u, tl_claimed = 0u;  // tl_claimed: ticks after the wave's start at which the lane's ray was handed out
  bool tl_seen_dry = false;
  if (TL) {
    if (threadIdx.x < 16u) tl_hist[threadIdx.x] = 0u;
    __syncthreads();
    tl_start = wall_clock64();
    if ((threadIdx.x & 63u) == 0u) atomicMax(&w.timeline[32u * stage + 0u], ~tl_start);
  }
  const uint32_t n_alive = w.ctr[WF_ALIVE + stage], tail = n_alive + w.ctr[WF_SHADOWS + stage];
  const uint32_t* __restrict__ alive = w.alive[stage & 1u];
  const uint32_t* __restrict__ shadow = w.shadow[stage & 1u];
  uint32_t* head_ptr = &w.ctr[WF_QHEAD + stage];
  const uint32_t all_lanes = gridDim.x * 256u;
  // the wave's reserve: entries [res_base, res_base + res_count) of the queue are this wave's to hand to its lanes
  uint32_t res_base = 0u, res_count = 0u;
  bool exhausted = tail == 0u;  // nothing left to reserve from the queue
  uint32_t phase = PH_IDLE, pending = 0u;
  uint32_t entry_id = 0u;
  Walk k;
  walk_begin(k, sc, F3(0, 0, 0), F3(1, 1, 1), 0.0f, 0.0f, HK_DONT_EXCLUDE);
  auto finish = [&]() {  // the lane's walk has ended: its result goes to the slot, the lane is free
    if (TL) {
      tl_max = max(tl_max, tl_steps);
      tl_sum += tl_steps;
      tl_rays += 1u;
      atomicAdd(&tl_hist[min(15u, 31u - (uint32_t)__clz((int)(tl_steps + 1u)))], 1u);
      if (tl_steps >= 256u) {  // the long walks: how long they took, and when they were handed out
        const uint32_t now_rel = (uint32_t)(wall_clock64() - tl_start);
        unsigned long long* tl = w.timeline + 32u * stage;
        atomicAdd(&tl[24], (unsigned long long)(now_rel - tl_claimed));  // ticks in flight, summed
        atomicAdd(&tl[25], (unsigned long long)tl_steps);
        atomicAdd(&tl[26], 1ull);
        atomicMax(&tl[27], ((unsigned long long)(now_rel - tl_claimed) << 32) | tl_steps);  // the slowest of them: ticks | its steps
        atomicMax(&tl[28], ((unsigned long long)tl_claimed << 32) | tl_steps);              // the one handed out last: ticks after start | its steps
      }
      tl_steps = 0u;
    }
    const uint32_t slot = entry_id & ~WF_SHADOW;
    if (entry_id & WF_SHADOW) {
      w.sh[slot] = k.hit.instance_index;
    } else {
      w.ch0[slot] = make_float4(k.hit.distance, k.hit.uv.x, k.hit.uv.y, u2f(k.hit.primitive_index));
      w.ch1[slot] = k.hit.instance_index;
    }
  };
  for (;;) {
    const unsigned long long idle_mask = __ballot(phase == PH_IDLE);
    const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
    const bool dry = exhausted && res_count == 0u;
    if (TL && exhausted && !tl_seen_dry) {
      tl_seen_dry = true;
      if ((threadIdx.x & 63u) == 0u) atomicMax(&w.timeline[32u * stage + 1u], ~wall_clock64());
    }
    if (dry && n_idle == 64u) break;
    if (!dry && (n_idle >= HK_WF_REFILL_MIN || n_idle == 64u)) {
      const bool idle = phase == PH_IDLE;
      const uint32_t rank = lane_rank(idle_mask);
      uint32_t mine = HK_U32_MAX;
      uint32_t given = 0u;
      if (res_count < n_idle && !exhausted) {  // hand out what is left, then reserve the next block
        given = res_count;
        if (idle && rank < given) mine = res_base + rank;
        // one atomic per 256 rays while every lane of the launch can still be fed four more times from what is left, per 64
        // near the end of the queue (short blocks there keep the last waves from walking a long reserve alone).
        // (Round 4, tools/wf_timeline.py: when the queue runs dry up to 60 rays still sit in each wave's PRIVATE reserve, behind
        // lanes busy with long walks.  Guided self-scheduling - a wave takes 1/4 of an even share of what is left, at least its idle
        // lanes - hands the last ray out right when the queue empties, as intended, and makes the frames 2-8 % SLOWER: twice the
        // atomics on one hot counter, and the stage's end is set by its longest walks either way - see DESIGN 8.1.)
        const uint32_t block = (res_base + given + 4u * all_lanes < tail) ? 256u : HK_WF_BLOCK_SMALL;
        uint32_t b = 0u;
        if ((threadIdx.x & 63u) == 0u) b = atomicAdd(head_ptr, block);
        b = __builtin_amdgcn_readfirstlane(b);
        res_base = b;
        res_count = b < tail ? min(block, tail - b) : 0u;
        if (b + block >= tail) exhausted = true;
      }
      if (idle && rank >= given && rank - given < res_count) mine = res_base + (rank - given);
      const uint32_t used = min(n_idle - given, res_count);
      res_base += used;
      res_count -= used;
      if (mine != HK_U32_MAX) {
        entry_id = mine < n_alive ? alive[mine] : (shadow[mine - n_alive] | WF_SHADOW);
        const uint32_t slot = entry_id & ~WF_SHADOW;
        if (entry_id & WF_SHADOW) {
          const float4 a = w.sr0[slot], b4 = w.sr1[slot];
          walk_begin(k, sc, F3(a.x, a.y, a.z), F3(b4.x, b4.y, b4.z), a.w, b4.w, w.sr2[slot]);
        } else {
          const float4 a = w.cr0[slot], b4 = w.cr1[slot];
          walk_begin(k, sc, F3(a.x, a.y, a.z), F3(b4.x, b4.y, b4.z), HK_F32_MAX, 0.0f, HK_DONT_EXCLUDE);
        }
        phase = PH_NODE;
        if (TL) tl_claimed = (uint32_t)(wall_clock64() - tl_start);
      }
    }
    // the phase with the most lanes waiting runs (ties: nodes, then triangles).
    // (Round 4: serving EVERY parked lane every turn once the wave can no longer be refilled - latency instead of lane utilisation
    // for the walks that end the launch - changes no stage by more than 1 %: tools/wf_timeline.py)
    const uint32_t n_node = (uint32_t)__popcll(__ballot(phase == PH_NODE));
    const uint32_t n_tri = (uint32_t)__popcll(__ballot(phase == PH_TRI));
    const uint32_t n_entry = (uint32_t)__popcll(__ballot(phase == PH_ENTRY));
    if (n_node * HK_WF_NODE_WEIGHT >= n_tri && n_node * HK_WF_NODE_WEIGHT >= n_entry && n_node != 0u) {
#pragma unroll 1
      for (int s = 0; s < HK_WF_STEPS; ++s) {
        if (phase == PH_NODE) {
          if (TL) tl_steps += 1u;
          phase = walk_node(k, sc, pending);
          if (phase == PH_IDLE) finish();
        }
      }
    } else if (n_tri >= n_entry) {
      if (phase == PH_TRI) {
        phase = walk_triangle(k, sc, pending);
        if (phase == PH_IDLE) finish();
      }
    } else {
      if (phase == PH_ENTRY) {
        walk_enter(k, sc, pending);
        phase = PH_NODE;
      }
    }
  }
  if (TL) {
    const unsigned long long now = wall_clock64();
    for (int off = 32; off > 0; off >>= 1) {
      tl_max = max(tl_max, (uint32_t)__shfl_down(tl_max, off));
      tl_sum += __shfl_down(tl_sum, off);
      tl_rays += __shfl_down(tl_rays, off);
    }
    unsigned long long* tl = w.timeline + 32u * stage;
    if ((threadIdx.x & 63u) == 0u) {
      atomicMax(&tl[2], now);
      atomicAdd(&tl[3], now - tl_start);
      atomicAdd(&tl[4], 1ull);
      atomicMax(&tl[5], (unsigned long long)tl_max);
      atomicAdd(&tl[6], (unsigned long long)tl_sum);
      atomicAdd(&tl[7], (unsigned long long)tl_rays);
    }
    __syncthreads();
    if (threadIdx.x < 16u && tl_hist[threadIdx.x]) atomicAdd(&tl[8u + threadIdx.x], (unsigned long long)tl_hist[threadIdx.x]);
  }
}

// ------------------------------------------------------------------ wide walk (round 4)
// a trace stage is a bulk at the memory system's rate for DEPENDENT DIVERGENT FETCHES - a miss costs a 128-B line whatever
// the node's size - followed by a tail as long as the longest walk's chain.  The flatten_custom layout spends one dependent fetch
// per BOX (84 per ray, 34-42 of them in the instance tree).  Here a fetch is one 128-B record with the boxes of an inner node's FOUR
// grandchildren (hk_kernels.hpp WideTrees): two levels per dependent step, on both levels of the scene, from ONE copy of the trees
// (1/8 of the bytes the eight threaded orderings take: what the 4 MB L2s can hold goes up accordingly).  The order in which
// children are visited is not stored but decided per ray - nearest first - with a per-lane stack: 28 entries in LDS (entry-major:
// conflict-free), the rest in a global spill area.  Same candidates, same per-triangle arithmetic on the same operands as
// traverse_top: the closest hit is the reference's except where two candidates tie exactly (the product default's bar, like the
// threaded orderings WAS; since round 5 hk_wide.hpp wide_triangle decides ties by the REFERENCE's rule - the leaf its walk meets first,
// from the leaves' ranks in ordering 0 - so the closest hit is the reference's, whatever the order of the visits); an any-hit ray's
// outcome - occluded or not - does not depend on the order at all.
// one thread per slot of ONE tree (ordering 0; links local to the tree): the record of the inner node at that slot, and - in the
// last slot - the root's
__global__ __launch_bounds__(256) void k_build_wide(const float4* __restrict__ nodes, uint32_t count, float4* __restrict__ wide, uint32_t* __restrict__ rank) {
  const uint32_t x = blockIdx.x * 256u + threadIdx.x;
  if (x >= count) return;
  auto entry_of = [&](uint32_t i) { return f2u(nodes[2u * i].w); };
  auto exit_of = [&](uint32_t i) { return f2u(nodes[2u * i + 1u].w); };
  // the rank of a leaf = its position in this (the reference's) flattening: the order in which the reference's walk meets the
  // leaves (hk_kernels.hpp WideTrees).  A folded navigator (scene_layout.hip fold_leaf_navigators) carries its leaf's id one slot
  // ahead of the leaf's own, never visited slot: the navigator's position is the leaf's.
  if (entry_of(x) >= HK_LEAF && !(x > 0u && entry_of(x - 1u) == entry_of(x))) rank[entry_of(x) - HK_LEAF] = x;
  const bool is_root = x + 1u == count;
  if (!is_root && entry_of(x) >= HK_LEAF) return;  // a leaf (or a leaf's unused slot) has no record
  float4 rec[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rec[2 * k] = make_float4(HK_F32_MAX, HK_F32_MAX, HK_F32_MAX, u2f(WIDE_NONE));
    rec[2 * k + 1] = make_float4(-HK_F32_MAX, -HK_F32_MAX, -HK_F32_MAX, 0.0f);
  }
  int n = 0;
  auto add = [&](uint32_t g) {  // box-node g becomes a child of the record
    const float4 lo = nodes[2u * g], hi = nodes[2u * g + 1u];
    const uint32_t e = f2u(lo.w);
    const uint32_t link = e >= HK_LEAF ? e : g;
    for (int k = 0; k < 4; ++k)
      if (k == n) {
        rec[2 * k] = make_float4(lo.x, lo.y, lo.z, u2f(link));
        rec[2 * k + 1] = make_float4(hi.x, hi.y, hi.z, 0.0f);
      }
    n += 1;
  };
  // the children of this node: the box-nodes at `first` and at first's exit, inside [first, limit)
  uint32_t first, limit;
  if (is_root) {  // flatten_custom stores no node for the root: its children are the box-nodes at 0 and at 0's exit
    first = 0u;
    limit = count;
  } else {
    first = x + 1u;
    limit = exit_of(x);
  }
  if (is_root && count == 1u) {
    add(0u);  // a tree of one leaf
  } else {
    uint32_t c = first;
    for (int side = 0; side < 2 && c < limit; ++side) {
      if (entry_of(c) >= HK_LEAF) {
        add(c);
      } else {  // an inner child: its own children take its place
        const uint32_t cl = exit_of(c);
        uint32_t g = c + 1u;
        for (int gs = 0; gs < 2 && g < cl; ++gs) {
          add(g);
          g = exit_of(g);
        }
      }
      c = exit_of(c);
    }
  }
  // (the root of a tree whose last slot is an inner node does not occur: the last node of a depth-first flattening is a leaf's)
  float4* out = wide + 8u * (size_t)x;
#pragma unroll
  for (int k = 0; k < 8; ++k) out[k] = rec[k];
}

#ifndef HK_WIDE_STEPS
#define HK_WIDE_STEPS 2      // records per turn of the node phase
#endif
#ifndef HK_WF_WIDE_WAVES
#define HK_WF_WIDE_WAVES 5   // waves per SIMD the wide trace kernel is compiled for (<= 96 VGPRs): 5 workgroups x 31 KB of LDS per CU
                             // (4 x 35 KB with a 32-entry LDS stack: indirect pass +2.4 % / +6 %, profiles/r04_wide_share_ab.txt)
#endif
// k_wf_trace with the wide walk: the same queue, the same refill, the same three phases - a NODE step is one record.
// Work sharing inside a wave (round 4, HK_WF_WIDE_SHARE): tools/wf_timeline.py shows a trace stage ending with 0.5-1.1 ms in which
// the queue is dry and a few long walks finish - 200-500 records at 3-6 us each - while the other lanes of their waves idle.  What a
// walk still has to do sits on its stack as INDEPENDENT subtrees, and the bottom entry is the farthest (largest, last to be
// visited) of them.  Once the wave can no longer be refilled, every idle lane takes the bottom entry of a busy lane's stack and
// walks that subtree for it, starting from the owner's current closest distance; helpers can be helped in turn.  A helper's hit is
// merged into the ROOT lane's (the lane that claimed the ray) when the helper's stack is empty; the root writes the result when its
// own piece and all helpers are done.  The closest hit is the minimum over all pieces under wide_triangle's order-independent
// tie rule, and every piece prunes with the closest distance ANY piece of its ray has found (share_best, LDS): the result depends
// neither on who walked what nor on timing; an any-hit ray is occluded iff any piece found an occluder.  Handed over: the bottom
// of the lane's instance-tree entries (below its WIDE_LEAVE marker, or its whole stack outside a mesh tree), else the bottom entry
// of the mesh tree it is in (a tombstone stays); the walk's context - ray, local ray, closest distance, 22 dwords - travels through
// the TAKER's unused stack column.  A dry wave also serves every parked lane every turn (HK_WF_DRY_ALL_PHASES): with many lanes of
// a wave at work again, waiting a turn for one's phase is what makes the stage longer.  profiles/r04_wide_share_ab.txt.
#ifndef HK_WF_WIDE_SHARE
#define HK_WF_WIDE_SHARE 1
#endif
#ifndef HK_WF_DRY_ALL_PHASES
#define HK_WF_DRY_ALL_PHASES 1
#endif
// The wide trace kernel hands its queue out in a permuted order (round 5): what ends a stage is the waves whose blocks of 64 rays
// happened to be expensive, and a block of consecutive entries is one small region of the image.  Runs of 2^HK_WF_QUEUE_RUN
// consecutive entries stay together (neighbouring pixels: coherent rays).  Config 3 (4 stages of 0.2-1.4 M rays: four blocks per
// wave) indirect pass 3.67 -> 3.56 ms, frame 6.51 -> 6.43; config 4 (up to 5 M rays per stage: the law of large numbers already
// balances) unchanged with runs of 16, +1 % / +4 % with runs of 4 / 1 - coherence matters there (profiles/r05_interleave_ab.txt).
#ifndef HK_WF_QUEUE_INTERLEAVE
#define HK_WF_QUEUE_INTERLEAVE 1
#endif
#ifndef HK_WF_QUEUE_RUN
#define HK_WF_QUEUE_RUN 4      // log2 of the run of consecutive queue entries the permutation keeps together
#endif
#ifndef HK_WF_SHARE_MIN
#define HK_WF_SHARE_MIN 16u  // idle lanes a dry wave must have before its working lanes hand entries over
#endif
#ifndef HK_WF_SHARE_STEPS
#define HK_WF_SHARE_STEPS 8u  // ... and records a lane must have visited for its piece before it does: only the long walks end a stage
#endif
namespace {
template <bool PATHS>
__device__ __forceinline__ void shade_bounce(const DScene& sc, const DFrame& fr, const WfBuffers& w, uint32_t slot, uint32_t n, bool& want_shadow, bool& want_next);  // (below, with k_wf_shade)
}
enum : uint32_t { PH_WAIT = 4u, PH_HELPED = 5u, PH_READY = 6u };  // a root whose helpers are still out / a helper whose piece is done (merged at the next turn) / PATHS: a closest hit found, the path goes to the wave's shading list at the next turn
// PATHS: a queue entry = slot | bounce << 26 | WF_SHADOW
constexpr uint32_t WF_SLOT_BITS = 26u, WF_SLOT_MASK = (1u << WF_SLOT_BITS) - 1u, WF_BOUNCE_MASK = 31u;
#ifndef HK_WF_PATHS_SHADE_EARLY
#define HK_WF_PATHS_SHADE_EARLY 48u  // PATHS: paths waiting for shading from which on a wave shades them rather than take new paths from the global queue
#endif
#ifndef HK_WF_PATHS_BLOCK
#define HK_WF_PATHS_BLOCK 64u       // PATHS: paths a wave reserves at a time (the staged stages reserve 256 rays until near the end)
#endif
#ifndef HK_WF_PATHS_SHADE_MIN
#define HK_WF_PATHS_SHADE_MIN 16u  // PATHS, a dry wave: paths that must wait for shading before the wave shades them (or as many as half its working lanes)
#endif
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// TL: the instrumented twin, as for k_wf_trace - tools/wf_timeline.py.  COUNT (round 5; HK_CTX_COUNT_WALKS): the COUNTING twin of
// THIS walk - records fetched (of them in the instance tree), triangle tests, instance entries, rays claimed (of them any-hit),
// closest hits found, pieces handed to idle lanes - per stage into WfBuffers::timeline[32 stage + 8 ..], next to three stamps (first
// wave in, queue first seen dry, last wave out) from which bench.py takes the stage's tail fraction.  The product launches
// <false, false>; the three differ in bookkeeping only: a ray's walk and result are the same.
//
// PATHS (round 6): EVERY bounce of the dispatch in this one launch - `stage` is not read.  The global
// queue holds the paths themselves (bounce 0's closest-hit rays, k_wf_setup's slots); a wave that claims a path keeps it to its end:
// when the closest hit of bounce n is found the path goes to the wave's own SHADING list, the wave shades 64 of them at a time with all
// its lanes (shade_bounce<true>: the same code the staged schedule's k_wf_shade runs), and the rays that emits - the shadow ray of
// bounce n, the closest-hit ray of bounce n + 1 - go to the wave's own RAY list, which idle lanes are refilled from before they take
// new paths.  Both lists are the wave's private memory (WfBuffers::local), every plane of a path is written and read by ONE wave, in
// program order: no hand-off between workgroups, no flag, no fence, nothing to wait for - and so nothing that could hang.  What this
// buys: a stage's end (the queue dry, a few long walks left: 42-70 % of a stage's time on configs 3 / 4) exists once per
// dispatch instead of once per bounce, the walks of bounce n + 1 start while those of bounce n are still out, a shadow ray is off
// its path's chain (nothing waits for its outcome before k_wf_final), and the dispatch is three launches instead of 2 x bounces + 3.
// A ray's walk, a bounce's arithmetic and the order of a path's additions are the staged schedule's: the same bytes in every buffer
// (tests/test_parity_schedules_gpu.py).  The shadow ray's RECORD (sr0 / sr1 / sr2) stays one per path: the ray list is first in,
// first out and bounce n's shadow ray enters it before bounce n + 1's closest-hit ray, so its walk has begun (the record is in
// registers) before bounce n + 1 can be shaded; its RESULT has a plane per bounce.
#ifndef HK_WF_PATHS_WAVES
#define HK_WF_PATHS_WAVES 4  // waves per SIMD the PATHS instantiation is compiled for: the shading needs 116 VGPRs on its own, the walk 93 - at five waves
                             // (96) 113 are spilled and the dispatch is 1.33x the staged one; the shading as a CALL (the walk saved around it, the kernel's
                             // arguments read from its argument segment by the callee) 1.17-1.20x; four waves (128, 22 spilled) 1.04-1.10x (profiles/r06_persistent_paths_ab.json)
#endif
struct WideTraceArgs {
  DScene sc;
  DFrame fr;
  WfBuffers w;
  WideTrees wt;
  uint32_t stage;
};
template <bool TL, bool COUNT, bool PATHS>
__global__ __launch_bounds__(256, (PATHS ? HK_WF_PATHS_WAVES : HK_WF_WIDE_WAVES)) void k_wf_trace_wide(const WideTraceArgs args) {
  const DScene& sc = args.sc;
  const DFrame& fr = args.fr;
  const WfBuffers& w = args.w;
  const WideTrees& wt = args.wt;
  uint32_t stage = args.stage;
  __shared__ uint32_t stack_lds[HK_WIDE_LDS_STACK * 256u];
  __shared__ uint32_t tl_hist[16];
  unsigned long long tl_start = 0ull;
  uint32_t tl_steps = 0u, tl_max = 0u, tl_sum = 0u, tl_rays = 0u, tl_claimed = 0u;
  bool tl_seen_dry = false;
  RayCounters cn{0, 0};  // COUNT: tlas = rays claimed, blas = of them any-hit, nodes / top_nodes / tris / entries / hits; pieces below
  uint32_t cn_pieces = 0u;
  if (TL) {
    if (threadIdx.x < 16u) tl_hist[threadIdx.x] = 0u;
    __syncthreads();
  }
  if (TL || COUNT) {
    tl_start = wall_clock64();
    if ((threadIdx.x & 63u) == 0u) atomicMax(&w.timeline[32u * stage + 0u], ~tl_start);
  }
  WideStackSpill stack{stack_lds, wt.spill, (size_t)gridDim.x * 256u, (size_t)blockIdx.x * 256u + threadIdx.x, wt.lost};
  if (PATHS) stage = 0u;
  const uint32_t n_alive = w.ctr[WF_ALIVE + stage], q_count = PATHS ? n_alive : n_alive + w.ctr[WF_SHADOWS + stage];
  // PATHS: the wave's own lists, rings of 512 (positions count up, taken mod 512).  The ray list holds at most 256 (shading needs room
  // for 128), the shading list at most 63 + the 192 closest hits that can come in while the ray list drains from 256 to 128
  uint32_t* const rayq = PATHS ? w.local + ((size_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 1024u : nullptr;
  uint32_t* const shadeq = PATHS ? rayq + 512u : nullptr;
  uint32_t rq_head = 0u, rq_count = 0u, sq_head = 0u, sq_count = 0u;
#if HK_WF_QUEUE_INTERLEAVE
  // the queue is handed out in a PERMUTED order: runs of 8 consecutive entries (neighbouring pixels: coherent rays) from places a
  // large odd stride apart, so that a wave's block of 64 samples eight regions of the image instead of one - the cost of a block
  // depends on where it lies (sky edge, dense geometry), and what ends a stage is the waves with an expensive draw
  const uint32_t q_chunks = (q_count + ((1u << HK_WF_QUEUE_RUN) - 1u)) >> HK_WF_QUEUE_RUN, tail = q_chunks << HK_WF_QUEUE_RUN;
  const uint32_t q_stride = (q_chunks % 7919u) ? 7919u : 7907u;  // coprime with q_chunks: chunk j -> (j x stride) mod chunks is a bijection
#else
  const uint32_t tail = q_count;
#endif
  const uint32_t* __restrict__ alive = w.alive[stage & 1u];
  const uint32_t* __restrict__ shadow = w.shadow[stage & 1u];
  uint32_t* head_ptr = &w.ctr[WF_QHEAD + stage];
  const uint32_t all_lanes = gridDim.x * 256u;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t res_base = 0u, res_count = 0u;
  bool exhausted = tail == 0u;
  uint32_t phase = PH_IDLE, pending = 0u;
  uint32_t entry_id = 0u;
  uint32_t root = lane;  // the lane (of this wave) the ray belongs to
  __shared__ uint32_t share_lane[4][64];  // taker rank -> thread
  __shared__ uint32_t share_help[256];    // (per root) pieces of its ray other lanes still walk
  __shared__ uint32_t share_best[256];    // (per root) the closest distance any piece of its ray has found, as ordered bits
  share_help[threadIdx.x] = 0u;
  share_best[threadIdx.x] = f2u(HK_F32_MAX);
  uint32_t steps = 0u;  // records this lane has visited for its current piece
  WideWalk k;
  wide_begin(k, wt, F3(0, 0, 0), F3(1, 1, 1), 0.0f, 0.0f, HK_DONT_EXCLUDE);
  auto begin_ray = [&](uint32_t id, float bound) {  // the ray of queue entry `id`, from its planes
    const uint32_t slot = PATHS ? id & WF_SLOT_MASK : id & ~WF_SHADOW;
    if (id & WF_SHADOW) {
      const float4 a = w.sr0[slot], b4 = w.sr1[slot];
      wide_begin(k, wt, F3(a.x, a.y, a.z), F3(b4.x, b4.y, b4.z), fminf(a.w, bound), b4.w, w.sr2[slot]);
    } else {
      const float4 a = w.cr0[slot], b4 = w.cr1[slot];
      wide_begin(k, wt, F3(a.x, a.y, a.z), F3(b4.x, b4.y, b4.z), bound, 0.0f, HK_DONT_EXCLUDE);
    }
  };
  auto write_result = [&]() -> uint32_t {  // (root) the ray is done: its result goes to the slot; returns the lane's next phase
    const uint32_t slot = PATHS ? entry_id & WF_SLOT_MASK : entry_id & ~WF_SHADOW;
    if (entry_id & WF_SHADOW) {
      if (PATHS) w.pb_sh[(size_t)((entry_id >> WF_SLOT_BITS) & WF_BOUNCE_MASK) * w.cap + slot] = k.hit.instance_index;
      else w.sh[slot] = k.hit.instance_index;
    } else {
      w.ch0[slot] = make_float4(k.hit.distance, k.hit.uv.x, k.hit.uv.y, u2f(k.hit.primitive_index));
      w.ch1[slot] = k.hit.instance_index;
      if (COUNT) cn.hits += k.hit.instance_index != HK_U32_MAX ? 1u : 0u;
      if (PATHS) return PH_READY;
    }
    return PH_IDLE;
  };
  auto finish = [&]() {  // the lane's piece of a walk has ended
    if (TL) {
      tl_max = max(tl_max, tl_steps);
      tl_sum += tl_steps;
      tl_rays += 1u;
      atomicAdd(&tl_hist[min(15u, 31u - (uint32_t)__clz((int)(tl_steps + 1u)))], 1u);
      if (tl_steps >= 128u) {  // the long walks (a record is two levels: half the skip-link kernel's threshold)
        const uint32_t now_rel = (uint32_t)(wall_clock64() - tl_start);
        unsigned long long* tl = w.timeline + 32u * stage;
        atomicAdd(&tl[24], (unsigned long long)(now_rel - tl_claimed));
        atomicAdd(&tl[25], (unsigned long long)tl_steps);
        atomicAdd(&tl[26], 1ull);
        atomicMax(&tl[27], ((unsigned long long)(now_rel - tl_claimed) << 32) | tl_steps);
        atomicMax(&tl[28], ((unsigned long long)tl_claimed << 32) | tl_steps);
      }
      tl_steps = 0u;
    }
    if (root != lane) {
      phase = PH_HELPED;
    } else if (share_help[threadIdx.x] != 0u) {
      phase = PH_WAIT;
    } else {
      phase = write_result();
    }
  };
  for (;;) {
#if HK_WF_WIDE_SHARE
    // helpers whose piece ended since the last turn: their hits go to their roots (one at a time: the wave runs in lock step,
    // so a root's registers can be written from here)
    for (unsigned long long done = __ballot(phase == PH_HELPED); done != 0ull; done &= done - 1ull) {
      const int h = __ffsll((long long)done) - 1;
      const uint32_t to = lane_u32(root, h), h_prim = lane_u32(k.hit.primitive_index, h), h_inst = lane_u32(k.hit.instance_index, h);
      const float h_d = u2f(lane_u32(f2u(k.hit.distance), h)), h_u = u2f(lane_u32(f2u(k.hit.uv.x), h)), h_v = u2f(lane_u32(f2u(k.hit.uv.y), h));
      if (lane == to) {
        share_help[threadIdx.x] -= 1u;
        if (h_inst != HK_U32_MAX) {  // the helper found something below the distance it started from
          // (the root may be inside a mesh tree with a hit of its own there: that hit's instance is cur_instance until it leaves)
          const uint32_t mine_inst = k.intersected ? k.cur_instance : k.hit.instance_index;
          bool closer = h_d < k.hit.distance;
          if (h_d == k.hit.distance && k.hit.primitive_index != HK_U32_MAX) closer = wide_tie_goes_to(wt, h_inst, h_prim, mine_inst, k.hit.primitive_index);
          if (closer) {
            k.hit.distance = h_d;
            k.hit.uv = F2(h_u, h_v);
            k.hit.primitive_index = h_prim;
            k.hit.instance_index = h_inst;
            k.intersected = false;  // (what the current mesh tree contributed so far lost against it)
          }
        }
      }
    }
    if (phase == PH_HELPED) phase = PH_IDLE;
    if (phase == PH_WAIT && share_help[threadIdx.x] == 0u) phase = write_result();
#endif
    if (PATHS) {  // closest hits found since the last turn: their paths wait for shading
      const unsigned long long ready = __ballot(phase == PH_READY);
      if (ready != 0ull) {
        if (phase == PH_READY) {
          shadeq[(sq_head + sq_count + lane_rank(ready)) & 511u] = entry_id;
          phase = PH_IDLE;
        }
        sq_count += (uint32_t)__popcll(ready);
      }
      // Shade: 64 paths with all 64 lanes (a lane in the middle of a walk keeps its walk in its registers and takes a turn at shading
      // like the others) as soon as 64 wait and the ray list has room for what they may emit; a wave with no other source of rays left
      // shades what it has once that is worth stopping its working lanes for.
      const uint32_t n_working = 64u - (uint32_t)__popcll(__ballot(phase == PH_IDLE));
      const bool starving = exhausted && res_count == 0u && rq_count == 0u;
      // (... and BEFORE it takes new paths from the global queue for its idle lanes: a wave that hoards paths - 64 walking, 63 waiting for
      // shading, 128 rays listed - leaves nothing for the queue to balance: 5 120 waves x 250 paths is a whole 1080p frame)
      const uint32_t n_idle_now = 64u - n_working;
      const bool would_claim = n_idle_now >= HK_WF_REFILL_MIN && rq_count < n_idle_now;
      const bool shade_now = sq_count >= 64u ? rq_count <= 128u
                                             : ((would_claim && sq_count >= HK_WF_PATHS_SHADE_EARLY) ||
                                                (starving && sq_count != 0u && (sq_count >= HK_WF_PATHS_SHADE_MIN || n_working <= 2u * sq_count)));
      if (shade_now) {
        const uint32_t batch = min(sq_count, 64u);
        bool want_shadow = false, want_next = false;
        uint32_t e = 0u;
        if (lane < batch) {
          e = shadeq[(sq_head + lane) & 511u];
          shade_bounce<true>(sc, fr, w, e & WF_SLOT_MASK, (e >> WF_SLOT_BITS) & WF_BOUNCE_MASK, want_shadow, want_next);
        }
        sq_head += batch;
        sq_count -= batch;
        // (first in, first out, and a bounce's shadow ray ahead of the next bounce's closest-hit ray: see the header)
        const unsigned long long ms = __ballot(want_shadow), mn = __ballot(want_next);
        if (want_shadow) rayq[(rq_head + rq_count + lane_rank(ms)) & 511u] = e | WF_SHADOW;
        rq_count += (uint32_t)__popcll(ms);
        if (want_next) rayq[(rq_head + rq_count + lane_rank(mn)) & 511u] = e + (1u << WF_SLOT_BITS);
        rq_count += (uint32_t)__popcll(mn);
      }
    }
    const unsigned long long idle_mask = __ballot(phase == PH_IDLE);
    const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
    const bool dry = exhausted && res_count == 0u && (!PATHS || rq_count == 0u);
    if ((TL || COUNT) && exhausted && !tl_seen_dry) {
      tl_seen_dry = true;
      if ((threadIdx.x & 63u) == 0u) atomicMax(&w.timeline[32u * stage + 1u], ~wall_clock64());
    }
    if (dry && n_idle == 64u && (!PATHS || sq_count == 0u)) break;
    if (!dry && (n_idle >= HK_WF_REFILL_MIN || n_idle == 64u)) {
      const bool idle = phase == PH_IDLE;
      uint32_t rank = lane_rank(idle_mask);
      uint32_t mine = HK_U32_MAX;
      uint32_t given = 0u;
      uint32_t n_idle_q = n_idle;  // idle lanes the global queue is asked for
      bool local_ray = false;
      if (PATHS) {  // the wave's own rays first
        const uint32_t take = min(n_idle, rq_count);
        if (idle && rank < take) {
          entry_id = rayq[(rq_head + rank) & 511u];
          local_ray = true;
        }
        rq_head += take;
        rq_count -= take;
        n_idle_q = n_idle - take;
        rank -= take;  // (wraps for the lanes served above: they ask the global queue for nothing)
      }
      const bool ask = idle && !local_ray;
      if (res_count < n_idle_q && !exhausted) {
        given = res_count;
        if (ask && rank < given) mine = res_base + rank;
        const uint32_t block = PATHS ? HK_WF_PATHS_BLOCK : ((res_base + given + 4u * all_lanes < tail) ? 256u : HK_WF_BLOCK_SMALL);
        uint32_t b = 0u;
        if ((threadIdx.x & 63u) == 0u) b = atomicAdd(head_ptr, block);
        b = __builtin_amdgcn_readfirstlane(b);
        res_base = b;
        res_count = b < tail ? min(block, tail - b) : 0u;
        if (b + block >= tail) exhausted = true;
      }
      if (ask && rank >= given && rank - given < res_count) mine = res_base + (rank - given);
      const uint32_t used = min(n_idle_q - given, res_count);
      res_base += used;
      res_count -= used;
#if HK_WF_QUEUE_INTERLEAVE
      if (mine != HK_U32_MAX) {
        const uint32_t chunk = (uint32_t)(((unsigned long long)(mine >> HK_WF_QUEUE_RUN) * q_stride) % q_chunks);
        mine = (chunk << HK_WF_QUEUE_RUN) | (mine & ((1u << HK_WF_QUEUE_RUN) - 1u));
        if (mine >= q_count) mine = HK_U32_MAX;  // (the padding of the last run)
      }
#endif
      if (mine != HK_U32_MAX || local_ray) {
        if (!local_ray) entry_id = PATHS ? mine : (mine < n_alive ? alive[mine] : (shadow[mine - n_alive] | WF_SHADOW));  // (PATHS: bounce 0 of path `mine`, k_wf_setup's slots are their own list)
        begin_ray(entry_id, HK_F32_MAX);
        if (COUNT) {
          cn.tlas += 1u;
          cn.blas += (entry_id & WF_SHADOW) ? 1u : 0u;
        }
        root = lane;
        steps = 0u;
        share_best[threadIdx.x] = f2u(HK_F32_MAX);
        phase = PH_NODE;
        if (TL) tl_claimed = (uint32_t)(wall_clock64() - tl_start);
      }
    }
#if HK_WF_WIDE_SHARE
    if (dry && n_idle >= HK_WF_SHARE_MIN) {
      // who can give: a lane at work with a pending entry that is nobody's current business - the bottom of its instance-tree
      // entries (the farthest, largest subtree), else, inside a mesh tree, the bottom of that tree's entries
      const bool working = phase == PH_NODE || phase == PH_TRI || phase == PH_ENTRY;
      const uint32_t tlas_top = k.in_blas ? k.mark : k.sp;  // instance-tree entries: [base, tlas_top)
      uint32_t give_at = HK_U32_MAX;
      bool give_blas = false;
      if (working && steps >= HK_WF_SHARE_STEPS) {
        if (k.base < tlas_top) give_at = k.base;
        else if (k.in_blas && k.blas_base < k.sp) { give_at = k.blas_base; give_blas = true; }
      }
      uint32_t link = give_at != HK_U32_MAX ? stack.get(give_at) : WIDE_NONE;
      if (link == WIDE_NONE || link == WIDE_LEAVE) {  // (a tombstone: skip it for the next time)
        if (give_at != HK_U32_MAX) { if (give_blas) k.blas_base += 1u; else k.base += 1u; }
        give_at = HK_U32_MAX;
      }
      const unsigned long long givers = __ballot(give_at != HK_U32_MAX);
      const uint32_t n_givers = (uint32_t)__popcll(givers);
      if (n_givers != 0u) {
        const uint32_t wave = threadIdx.x >> 6;
        const bool idle = phase == PH_IDLE;
        const uint32_t t_rank = lane_rank(idle_mask), g_rank = lane_rank(givers);
        if (idle && t_rank < n_givers) share_lane[wave][t_rank] = threadIdx.x;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (give_at != HK_U32_MAX && g_rank < n_idle) {  // hand the entry over: the walk's context goes into the taker's (unused) stack column
          static_assert(HK_WIDE_LDS_STACK >= 22u, "the context a lane hands over travels through 22 entries of the taker's LDS stack column");
          uint32_t* col = stack_lds + share_lane[wave][g_rank];
          const bool in_blas = give_blas;
          const f3 co = in_blas ? k.co : k.origin, ld = in_blas ? k.ld : k.direction;  // (the reciprocals are taken again by the taker: the same divisions)
          const uint32_t ctx[22] = {link, entry_id, root, f2u(k.hit.distance), f2u(k.origin.x), f2u(k.origin.y), f2u(k.origin.z), f2u(k.direction.x), f2u(k.direction.y),
                                    f2u(k.direction.z), f2u(k.early_distance), k.exclude_instance, (in_blas ? 1u : 0u) | (k.hit.primitive_index != HK_U32_MAX ? 2u : 0u), k.mesh_base, k.prim_base, k.cur_instance,
                                    f2u(co.x), f2u(co.y), f2u(co.z), f2u(ld.x), f2u(ld.y), f2u(ld.z)};
#pragma unroll
          for (int e = 0; e < 22; ++e) col[e * 256] = ctx[e];
          atomicAdd(&share_help[(threadIdx.x & ~63u) + root], 1u);
          if (give_blas) {
            stack.put(give_at, WIDE_NONE);  // (a tombstone: popping it costs one turn)
            k.blas_base += 1u;
          } else {
            k.base += 1u;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (idle && t_rank < n_givers) {  // a walk of the same ray over that one subtree, limited by the giver's closest distance so far
          const uint32_t* col = stack_lds + threadIdx.x;
          uint32_t ctx[22];
#pragma unroll
          for (int e = 0; e < 22; ++e) ctx[e] = col[e * 256];
          entry_id = ctx[1];
          root = ctx[2];
          k.origin = F3(u2f(ctx[4]), u2f(ctx[5]), u2f(ctx[6]));
          k.direction = F3(u2f(ctx[7]), u2f(ctx[8]), u2f(ctx[9]));
          k.inv_direction = 1.0f / k.direction;   // (wide_begin's division)
          k.early_distance = u2f(ctx[10]);
          k.exclude_instance = ctx[11];
          k.hit.uv = F2(0.0f, 0.0f);
          // the giver's closest distance is this piece's limit.  If it is a HIT's distance, a candidate at exactly that distance must
          // still be accepted here - it may carry the smaller (instance, primitive) and win the tie at the merge - so the piece starts
          // from the next float above it (tests/test_wide_model.py::test_ties_do_not_depend_on_the_order is this case on the CPU)
          k.hit.distance = (ctx[12] & 2u) ? u2f(ctx[3] + 1u) : u2f(ctx[3]);
          k.limit = HK_F32_MAX;
          k.hit.instance_index = HK_U32_MAX;
          k.hit.primitive_index = HK_U32_MAX;
          k.in_blas = (ctx[12] & 1u) != 0u;
          k.mesh_base = ctx[13];
          k.prim_base = ctx[14];
          k.cur_instance = ctx[15];
          k.co = F3(u2f(ctx[16]), u2f(ctx[17]), u2f(ctx[18]));
          k.ld = F3(u2f(ctx[19]), u2f(ctx[20]), u2f(ctx[21]));
          k.cinv = k.in_blas ? 1.0f / k.ld : k.inv_direction;  // (wide_enter's division)
          k.intersected = false;
          k.cur = WIDE_NONE;
          k.sp = 0u;
          k.base = 0u;
          k.mark = 0u;
          k.blas_base = 0u;
          if (k.in_blas) {
            wide_push(k, stack, WIDE_LEAVE);
            k.blas_base = 1u;
          }
          wide_push(k, stack, ctx[0]);
          steps = 0u;
          phase = PH_NODE;
          if (COUNT) cn_pieces += 1u;
          if (TL) tl_claimed = (uint32_t)(wall_clock64() - tl_start);
        }
      }
    }
#endif
    const uint32_t n_node = (uint32_t)__popcll(__ballot(phase == PH_NODE));
    const uint32_t n_tri = (uint32_t)__popcll(__ballot(phase == PH_TRI));
    const uint32_t n_entry = (uint32_t)__popcll(__ballot(phase == PH_ENTRY));
    // While the queue lasts the phase with the most lanes waiting runs (lane utilisation: an idle lane is refilled).  Once the wave
    // is dry every parked lane is served every turn: what is left are the walks that end the stage, and (HK_WF_WIDE_SHARE) the
    // lanes that help them - a lane that waits a turn for its phase makes the stage a turn longer.
    const bool all_phases = HK_WF_DRY_ALL_PHASES && dry;
    if (all_phases ? n_node != 0u : (n_node >= n_tri && n_node >= n_entry && n_node != 0u)) {
#if HK_WF_WIDE_SHARE
      if (dry) k.limit = u2f(share_best[(threadIdx.x & ~63u) + root]);  // (what the other pieces of the ray have found meanwhile)
#endif
#pragma unroll 1
      for (int s = 0; s < HK_WIDE_STEPS; ++s) {
        if (phase == PH_NODE) {
          if (TL) tl_steps += 1u;
          steps += 1u;
          phase = wide_node<WideStackSpill, COUNT>(k, wt, stack, pending, &cn);
          if (phase == PH_IDLE) finish();
        }
      }
    }
    if (all_phases ? n_tri != 0u : (!(n_node >= n_tri && n_node >= n_entry && n_node != 0u) && n_tri >= n_entry)) {
      if (phase == PH_TRI) {
        const float before = k.hit.distance;
        if (COUNT) cn.tris += 1u;
        phase = wide_triangle(k, sc, wt, pending);
#if HK_WF_WIDE_SHARE
        if (dry && k.hit.distance < before) atomicMin(&share_best[(threadIdx.x & ~63u) + root], f2u(k.hit.distance));  // (distances are >= 0: their bits order like they do)
#endif
        if (phase == PH_IDLE) finish();
      }
    }
    if (all_phases ? n_entry != 0u : (!(n_node >= n_tri && n_node >= n_entry && n_node != 0u) && n_tri < n_entry)) {
      if (phase == PH_ENTRY) {
        if (COUNT) cn.entries += 1u;
        wide_enter(k, sc, stack, pending);
        phase = PH_NODE;
      }
    }
  }
  if (COUNT) {
    const unsigned long long now = wall_clock64();
    uint32_t v[8] = {cn.nodes, cn.top_nodes, cn.tris, cn.entries, cn.tlas, cn.blas, cn.hits, cn_pieces};
    for (int off = 32; off > 0; off >>= 1)
      for (int j = 0; j < 8; ++j) v[j] += __shfl_down(v[j], off);
    unsigned long long* tl = w.timeline + 32u * stage;
    if ((threadIdx.x & 63u) == 0u) {
      atomicMax(&tl[2], now);
      atomicAdd(&tl[3], now - tl_start);
      atomicAdd(&tl[4], 1ull);
      for (int j = 0; j < 8; ++j)
        if (v[j]) atomicAdd(&tl[8 + j], (unsigned long long)v[j]);
    }
  }
  if (TL) {
    const unsigned long long now = wall_clock64();
    for (int off = 32; off > 0; off >>= 1) {
      tl_max = max(tl_max, (uint32_t)__shfl_down(tl_max, off));
      tl_sum += __shfl_down(tl_sum, off);
      tl_rays += __shfl_down(tl_rays, off);
    }
    unsigned long long* tl = w.timeline + 32u * stage;
    if ((threadIdx.x & 63u) == 0u) {
      atomicMax(&tl[2], now);
      atomicAdd(&tl[3], now - tl_start);
      atomicAdd(&tl[4], 1ull);
      atomicMax(&tl[5], (unsigned long long)tl_max);
      atomicAdd(&tl[6], (unsigned long long)tl_sum);
      atomicAdd(&tl[7], (unsigned long long)tl_rays);
    }
    __syncthreads();
    if (threadIdx.x < 16u && tl_hist[threadIdx.x]) atomicAdd(&tl[8u + threadIdx.x], (unsigned long long)tl_hist[threadIdx.x]);
  }
}

// ------------------------------------------------------------------ shade: bounce n of every live path
#ifndef HK_WF_SHADE_WAVES
#define HK_WF_SHADE_WAVES 4
#endif
namespace {
// Bounce `n` of the path in `slot`, from its closest hit on: light.wgsl:1313-1394, one iteration (bounce_step of kernels.hip).
// PATHS = false, the staged schedule: the shadow ray of bounce n - 1 has been traced by the stage before, its outcome is added first.
// PATHS = true, k_wf_paths: nothing here waits for a shadow ray - the two outcomes of bounce n's go to the planes OF THAT BOUNCE
// (WfBuffers::pb_add), the bounce's bit to the path's pending mask, and k_wf_final<true> adds what the walks selected, in bounce order.
template <bool PATHS>
__device__ __forceinline__ void shade_bounce(const DScene& sc, const DFrame& fr, const WfBuffers& w, uint32_t slot, uint32_t n, bool& want_shadow, bool& want_next) {
  RayCounters rc{0, 0};
  f4 random = F4(plane(w, PL_RANDOM)[slot]);
  const float4 pp = plane(w, PL_POSITION_PDF)[slot];
  const float4 np = plane(w, PL_NORMAL_PENDING)[slot];
  f3 position = F3(pp.x, pp.y, pp.z), normal = F3(np.x, np.y, np.z);
  float pdf = pp.w;
  f3 transport = F3(1.0f, 1.0f, 1.0f);  // light.wgsl:1310-1311
  f4 radiance = F4(0.0f, 0.0f, 0.0f, 0.0f);
  if (n != 0u) {
    transport = xyz(F4(plane(w, PL_TRANSPORT)[slot]));
    if (!PATHS) radiance = F4(plane(w, PL_RADIANCE)[slot]);
  }
  uint32_t mask = PATHS ? f2u(np.w) : 0u;  // PATHS: bit k = bounce k has a shadow ray out, bit 31 = the path left the scene (its sky term: PL_RADIANCE)
  if (!PATHS && np.w != 0.0f) {  // the shadow ray of bounce n - 1 has been traced by now: add the outcome it selected
    const float4 add = (w.sh[slot] != HK_U32_MAX) ? plane(w, PL_ADD_OCCLUDED)[slot] : plane(w, PL_ADD_CLEAR)[slot];
    radiance = radiance + F4(add.x, add.y, add.z, 1.0f);
  }
  const float4 ro = w.cr0[slot], rd = w.cr1[slot], h0 = w.ch0[slot];
  Ray ray;
  ray.origin = F3(ro.x, ro.y, ro.z);
  ray.direction = F3(rd.x, rd.y, rd.z);
  ray.inv_direction = F3(0, 0, 0);  // not read below
  const float rand_sample_w = rd.w;
  Hit hit;
  hit.distance = h0.x;
  hit.uv = F2(h0.y, h0.z);
  hit.primitive_index = f2u(h0.w);
  hit.instance_index = w.ch1[slot];
  HitInfo info = hit_info(sc, ray, hit);
  if (n == 0u) {
    plane(w, PL_FIRST_POSITION)[slot] = to_float4(info.position);
    plane(w, PL_FIRST_NORMAL)[slot] = make_float4(info.normal.x, info.normal.y, info.normal.z, 0.0f);
    pdf = rand_sample_w;
  }
  const f3 sample_position = xyz(info.position);
  const f3 sample_normal = info.normal;
  float pending = 0.0f;
  if (hit.instance_index != HK_U32_MAX) {
    Surface surface = retreive_surface(sc, info.material_index, info.uv);
    surface.roughness = 1.0f;
    const uint32_t info_instance = info.instance_index;
    LightCandidate candidate = select_light_candidate(sc, fr, random, sample_position, sample_normal, info_instance, info, rc);
    const bool sample_directional = (candidate.emissive_instance == HK_DONT_SAMPLE_EMISSIVE);
    const f3 bounce_view_direction = normalize(position - sample_position);
    if (dot(candidate.direction, sample_normal) > 0.0f && candidate.p > 0.0f) {
      Ray sray;
      sray.origin = sample_position + sample_normal * HK_RAY_BIAS;
      sray.direction = candidate.direction;
      sray.inv_direction = F3(0, 0, 0);
      w.sr0[slot] = make_float4(sray.origin.x, sray.origin.y, sray.origin.z, candidate.max_distance);
      w.sr1[slot] = make_float4(sray.direction.x, sray.direction.y, sray.direction.z, candidate.min_distance);
      w.sr2[slot] = candidate.emissive_instance;
      // both outcomes of the walk (see the header): unoccluded = the candidate's own hit info, occluded = (0,0,0,1)
      const f4 in_clear = input_radiance(sc, fr, sray, info, sample_directional, candidate.emissive_instance, false);
      f3 add[2];
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const f4 in_radiance = o == 0 ? in_clear : F4(0.0f, 0.0f, 0.0f, 1.0f);
        f3 out_radiance = shading(fr, bounce_view_direction, sample_normal, sray.direction, surface, in_radiance);
        out_radiance = out_radiance / candidate.p;
        if (n > 0u) out_radiance = (rand_sample_w < 0.01f) ? F3(0, 0, 0) : out_radiance / rand_sample_w;
        const float out_luminance = luminance(out_radiance);
        if (out_luminance > fr.max_indirect_luminance) out_radiance = out_radiance * fr.max_indirect_luminance / out_luminance;
        add[o] = transport * out_radiance;
      }
      float4* add_clear = PATHS ? w.pb_add + (size_t)(2u * n) * w.cap : plane(w, PL_ADD_CLEAR);
      float4* add_occluded = PATHS ? w.pb_add + (size_t)(2u * n + 1u) * w.cap : plane(w, PL_ADD_OCCLUDED);
      add_clear[slot] = make_float4(add[0].x, add[0].y, add[0].z, 0.0f);
      add_occluded[slot] = make_float4(add[1].x, add[1].y, add[1].z, 0.0f);
      pending = 1.0f;
      mask |= 1u << n;
      want_shadow = true;
    }
    transport = transport * env_brdf(bounce_view_direction, sample_normal, surface);
    random = fract(random + fr.number_golden);
    position = sample_position;
    normal = sample_normal;
    // the loop condition of light.wgsl:1313 for bounce n + 1
    want_next = n + 1u < fr.indirect_bounces && (transport.x > 0.01f || transport.y > 0.01f || transport.z > 0.01f);
    if (want_next) emit_bounce_ray(w, slot, random, position, normal);
  } else {
    const f3 out_radiance = xyz(input_radiance(sc, fr, ray, info, false, HK_DONT_SAMPLE_EMISSIVE, true));
    if (PATHS) {  // (the last addition of the path: after every shadow ray's outcome - k_wf_final<true>)
      radiance = F4(transport * out_radiance, 0.0f);
      mask |= 0x80000000u;
    } else {
      radiance = radiance + F4(transport * out_radiance, 0.0f);
    }
  }
  plane(w, PL_RANDOM)[slot] = to_float4(random);
  plane(w, PL_POSITION_PDF)[slot] = make_float4(position.x, position.y, position.z, pdf);
  plane(w, PL_NORMAL_PENDING)[slot] = make_float4(normal.x, normal.y, normal.z, PATHS ? u2f(mask) : pending);
  plane(w, PL_TRANSPORT)[slot] = make_float4(transport.x, transport.y, transport.z, 0.0f);
  if (!PATHS || (mask & 0x80000000u)) plane(w, PL_RADIANCE)[slot] = to_float4(radiance);
}
}  // namespace
template <bool LDS>
__global__ __launch_bounds__(256, HK_WF_SHADE_WAVES) void k_wf_shade(DScene gsc, DFrame fr, WfBuffers w, uint32_t n) {
  const DScene sc = stage_scene<LDS>(gsc);
  const uint32_t count = w.ctr[WF_ALIVE + n];
  const uint32_t* __restrict__ alive_in = w.alive[n & 1u];
  uint32_t* alive_out = w.alive[(n + 1u) & 1u];
  uint32_t* shadow_out = w.shadow[(n + 1u) & 1u];
  __shared__ uint32_t push_lds[6];
  for (uint32_t first = blockIdx.x * 256u; first < count; first += gridDim.x * 256u) {  // workgroup-uniform trip count
    const uint32_t i = first + threadIdx.x;
    const bool valid = i < count;
    bool want_shadow = false, want_next = false;
    uint32_t slot = 0u;
    if (valid) {
      slot = alive_in[i];
      shade_bounce<false>(sc, fr, w, slot, n, want_shadow, want_next);
    }
    // survivors and shadow rays of the next trace stage: one atomic per workgroup and list
    const uint32_t a = block_push(&w.ctr[WF_ALIVE + n + 1u], want_next, push_lds);
    if (want_next) alive_out[a] = slot;
    const uint32_t q = block_push(&w.ctr[WF_SHADOWS + n + 1u], want_shadow, push_lds);
    if (want_shadow) shadow_out[q] = slot;
  }
}

// ------------------------------------------------------------------ final: last shadow result + the temporal-reuse tail
template <bool PATHS>
__global__ __launch_bounds__(256, 4) void k_wf_final(DScene sc, DFrame fr, GBuffer g, LightTargets t, WfBuffers w) {
  const uint32_t count = w.ctr[WF_ALIVE];
  for (uint32_t slot = blockIdx.x * 256u + threadIdx.x; slot < count; slot += gridDim.x * 256u) {
    const int index = (int)w.pixel[slot];
    const int y = index / fr.rw, x = index - y * fr.rw;
    // the G-buffer prologue of k_indirect again (light.wgsl:1263-1308): the same loads and operations, the same values
    const f2 uv = coords_to_uv(fr, x, y);
    int dcx, dcy;
    jittered_deferred_coords(fr, uv, &dcx, &dcy);
    const int didx = dcx + fr.dw * dcy;  // in bounds: the pixel was not background
    const float4 position_depth = g.position[didx];
    const f3 position = xyz(position_depth);
    const float2 imf = g.instance_material[didx];
    const uint32_t im_x = f32_to_u32(imf.x), im_y = f32_to_u32(imf.y);
    const float4 velocity_uv = g.velocity_uv[didx];
    Sample s = zero_sample();
    s.random = noise_fetch(sc, x, y, fr.number);
    s.random = fract(s.random + fr.number_golden);
    s.visible_position = F4(position, position_depth.w);
    s.visible_normal = normalize(xyz(unpack4x8snorm(g.normal[didx])));
    s.visible_instance = im_x;
    f4 radiance = F4(0.0f, 0.0f, 0.0f, 0.0f);
    if (PATHS) {  // the path's additions in bounce order (shade_bounce<true>): what each bounce's shadow ray selected, then the sky term if it left the scene
      const uint32_t mask = f2u(plane(w, PL_NORMAL_PENDING)[slot].w);
      for (uint32_t m = mask & 0x7FFFFFFFu; m != 0u; m &= m - 1u) {
        const uint32_t n = (uint32_t)__ffs((int)m) - 1u;
        const float4 add = w.pb_add[(size_t)(2u * n + (w.pb_sh[(size_t)n * w.cap + slot] != HK_U32_MAX ? 1u : 0u)) * w.cap + slot];
        radiance = radiance + F4(add.x, add.y, add.z, 1.0f);
      }
      if (mask & 0x80000000u) radiance = radiance + F4(plane(w, PL_RADIANCE)[slot]);
    } else {
      radiance = F4(plane(w, PL_RADIANCE)[slot]);
      if (plane(w, PL_NORMAL_PENDING)[slot].w != 0.0f) {
        const float4 add = (w.sh[slot] != HK_U32_MAX) ? plane(w, PL_ADD_OCCLUDED)[slot] : plane(w, PL_ADD_CLEAR)[slot];
        radiance = radiance + F4(add.x, add.y, add.z, 1.0f);
      }
    }
    s.radiance = radiance;
    s.sample_position = F4(plane(w, PL_FIRST_POSITION)[slot]);
    s.sample_normal = xyz(F4(plane(w, PL_FIRST_NORMAL)[slot]));
    const float pdf = plane(w, PL_POSITION_PDF)[slot].w;
    indirect_temporal_tail(sc, fr, t, index, uv, position, velocity_uv, im_y, s, pdf);
  }
}

}  // namespace hkd

// ------------------------------------------------------------------ host launcher
namespace hk {
using namespace hkd;

size_t wide_trace_lanes(int compute_units) { return (size_t)compute_units * HK_WF_WIDE_WAVES * 256u; }  // = the grid of k_wf_trace_wide below
size_t wide_spill_entries() { return HK_WIDE_SPILL; }
void launch_build_wide(hipStream_t st, const float4* nodes, uint32_t count, float4* wide, uint32_t* rank) {
  if (count) hipLaunchKernelGGL(k_build_wide, dim3((count + 255u) / 256u), dim3(256), 0, st, nodes, count, wide, rank);
}

void launch_indirect_wavefront(hipStream_t st, const DScene& sc, const DFrame& fr, const GBuffer& g, const LightTargets& t, const WfBuffers& w, int y0,
                               int y1, int compute_units, hipEvent_t start, hipEvent_t stop, const WideTrees* wide, hipEvent_t* trace_events, bool persistent) {
  if (y1 <= y0) return;
  (void)hipMemsetAsync(w.ctr, 0, 192 * sizeof(uint32_t), st);
  if (w.timeline) (void)hipMemsetAsync(w.timeline, 0, 64 * 32 * sizeof(unsigned long long), st);
  hipExtLaunchKernelGGL(k_wf_setup, grid_for(fr.rw, y1 - y0), dim3(256), 0, st, start, nullptr, 0, sc, fr, g, t, w, y0, y1);
  const size_t lds = (size_t)sc.blob_f4 * 16 <= HK_LDS_SCENE_BYTES ? (size_t)sc.blob_f4 * 16 : 0;
  const dim3 persistent_grid((unsigned)(compute_units * 8));  // 8 workgroups of 4 waves per CU: what 64 VGPRs leave resident
  // rays in flight = lanes of the trace launch.  Little's law: with R node steps per second served by the memory system, a step of
  // one ray takes (rays in flight) / R - every ray beyond what saturates R only makes all of them slower, and the launch ends with
  // its longest walk (tools/wf_timeline.py; -DHK_WF_TRACE_WG_PER_CU=n for the A/B)
  const dim3 tracers((unsigned)(compute_units * HK_WF_TRACE_WG_PER_CU));
  const dim3 wide_tracers((unsigned)(compute_units * HK_WF_WIDE_WAVES));
  const uint32_t bounces = fr.indirect_bounces;
  const bool use_wide = wide && wide->tlas && !lds;
  const uint32_t twin = w.timeline ? w.timeline_mode : 0u;
  // every bounce in one launch (k_wf_trace_wide<.., PATHS>): the wide walk's scenes, no instrumented twin, the per-bounce planes in place
  if (persistent && use_wide && twin <= 1u && w.local && bounces >= 1u && w.pb_bounces >= bounces && w.cap <= (1u << 26)) {
    hipEvent_t e0 = trace_events ? trace_events[0] : nullptr, e1 = trace_events ? trace_events[1] : nullptr;
    if (twin == 1u) hipExtLaunchKernelGGL((k_wf_trace_wide<true, false, true>), dim3((unsigned)(compute_units * HK_WF_PATHS_WAVES)), dim3(256), 0, st, e0, e1, 0, WideTraceArgs{sc, fr, w, *wide, 0u});
    else hipExtLaunchKernelGGL((k_wf_trace_wide<false, false, true>), dim3((unsigned)(compute_units * HK_WF_PATHS_WAVES)), dim3(256), 0, st, e0, e1, 0, WideTraceArgs{sc, fr, w, *wide, 0u});
    hipExtLaunchKernelGGL(k_wf_final<true>, persistent_grid, dim3(256), 0, st, nullptr, stop, 0, sc, fr, g, t, w);
    return;
  }
  for (uint32_t n = 0; n <= bounces; ++n) {
    hipEvent_t e0 = trace_events ? trace_events[2u * n] : nullptr, e1 = trace_events ? trace_events[2u * n + 1u] : nullptr;
    if (use_wide && twin == 1u) hipExtLaunchKernelGGL((k_wf_trace_wide<true, false, false>), wide_tracers, dim3(256), 0, st, e0, e1, 0, WideTraceArgs{sc, fr, w, *wide, n});
    else if (use_wide && twin == 2u) hipExtLaunchKernelGGL((k_wf_trace_wide<false, true, false>), wide_tracers, dim3(256), 0, st, e0, e1, 0, WideTraceArgs{sc, fr, w, *wide, n});
    else if (use_wide) hipExtLaunchKernelGGL((k_wf_trace_wide<false, false, false>), wide_tracers, dim3(256), 0, st, e0, e1, 0, WideTraceArgs{sc, fr, w, *wide, n});
    else if (twin == 1u && !lds) hipExtLaunchKernelGGL((k_wf_trace<false, true>), tracers, dim3(256), 0, st, e0, e1, 0, sc, w, n);
    else if (lds) hipExtLaunchKernelGGL((k_wf_trace<true, false>), tracers, dim3(256), lds, st, e0, e1, 0, sc, w, n);
    else hipExtLaunchKernelGGL((k_wf_trace<false, false>), tracers, dim3(256), 0, st, e0, e1, 0, sc, w, n);
    if (n == bounces) break;
    if (lds) hipLaunchKernelGGL((k_wf_shade<true>), persistent_grid, dim3(256), lds, st, sc, fr, w, n);
    else hipLaunchKernelGGL((k_wf_shade<false>), persistent_grid, dim3(256), 0, st, sc, fr, w, n);
  }
  hipExtLaunchKernelGGL(k_wf_final<false>, persistent_grid, dim3(256), 0, st, nullptr, stop, 0, sc, fr, g, t, w);
}

}  // namespace hk
