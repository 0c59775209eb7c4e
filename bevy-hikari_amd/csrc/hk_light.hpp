// hk_light.hpp - device helpers shared by the light kernels (kernels.hip: the fused dispatches; kernels_wavefront.hip: the
// queue-based schedule of indirect_lit_ambient): LDS staging of small scenes, the racing / parked store to
// previous_spatial, ray counters.
#pragma once
#include <hip/hip_runtime.h>

#include "hk_device.hpp"
#include "hk_kernels.hpp"

namespace hkd {

// Small scenes (the whole Cornell box is 9 KB) are copied into LDS once per workgroup and traversed
// from there: a node step is then a ds_read_b128 pair (~64-cycle latency, 128+ B/clk/CU) instead of
// an L1-hit global load (~120+ cycles) in the dependent load -> slab test -> next-index chain.
// (Skipping the copy in workgroups whose pixels are all background - 63 % of the Cornell frame - was measured and
// rejected: the vote needs the depth first, which puts an HBM round trip in front of the copy; 0.417 vs 0.38 ms.)
// MODE: 0 = global memory, two-level walk; 1 = LDS copy, two-level walk; 2 = LDS copy, one-level walk (DScene::flat);
// 3 = global memory, one-level walk (the ray-counting replay of a scene the timed kernels walk in mode 2: same visits, same
// bits).  (bool converts: true = 1, false = 0.)  The walk is a compile-time constant of the instantiation.
template <int MODE>
__device__ __forceinline__ DScene stage_scene(const DScene& sc) {
  if constexpr (MODE == 0 || MODE == 3 || MODE == 4) {  // (4: global memory, the wide walk - a prepass instantiation of its own, so that the skip-link one carries no stack)
    DScene g = sc;
    g.flat_mode = MODE == 3 ? 1u : 0u;
    return g;
  } else {
    extern __shared__ __attribute__((aligned(16))) float4 hk_smem[];
    for (uint32_t i = threadIdx.x; i < sc.blob_f4; i += 256u) hk_smem[i] = sc.blob[i];
    __syncthreads();
    DScene l = sc;
    l.tlas_stride = 0u;  // (a scene that fits the LDS copy keeps the reference's single order: compile-time zeros, the octant arithmetic folds away)
    l.blas_stride = 0u;
    l.flat_mode = MODE == 2 ? 1u : 0u;
    const char* gb = reinterpret_cast<const char*>(sc.blob);
    const char* lb = reinterpret_cast<const char*>(hk_smem);
#define HK_REBASE(field) l.field = reinterpret_cast<decltype(l.field)>(lb + (reinterpret_cast<const char*>(sc.field) - gb))
    HK_REBASE(nodes); HK_REBASE(instances);
    if (MODE == 2) HK_REBASE(flat);
    HK_REBASE(tri_v0); HK_REBASE(tri_v1); HK_REBASE(tri_v2); HK_REBASE(vtx_normal); HK_REBASE(vtx_uv);
    HK_REBASE(materials); HK_REBASE(tex_info); HK_REBASE(srgb_lut); HK_REBASE(light_lo); HK_REBASE(light_hi); HK_REBASE(emissives); HK_REBASE(alias);
#undef HK_REBASE
    return l;
  }
}

// every store to previous_spatial goes through here: the reference lets them race; the verification mode parks them
__device__ __forceinline__ void store_previous_spatial(const LightTargets& t, int from, int to, const PackedReservoir& v) {
  if (t.det_winner && (!t.det_lite || to != from)) {
    store_packed(t.det_pending, from, v);
    t.det_to[from] = to;
    atomicMax(&t.det_winner[to], from);
  } else if (t.det_lite) {  // the pixel's own slot: nobody else's own store goes there; k_resolve_scatter_lite weighs it against the parked ones
    store_packed(t.previous_spatial, to, v);
    t.det_to[from] = from;
  } else {
    store_packed(t.previous_spatial, to, v);
    // a store into a slot some other wave's tile owns: whatever that tile's record says, it no longer holds (TileMeta::poison)
    if (t.m_previous_spatial && to != from) {
      const int ty = to / t.rw, tx = to - ty * t.rw;
      atomicMax(&t.m_previous_spatial[(ty >> 3) * t.tiles_x + (tx >> 3)].poison, t.serial);
    }
  }
}
template <bool COUNT>
__device__ __forceinline__ void flush_counters(const RayCounters& rc, uint32_t primary, unsigned long long* counters) {
  if (!COUNT) return;
  // counters: [0] primary rays, [1] traverse_top walks, [2] stand-alone traverse_bottom walks, [3] node steps, [4] triangle tests,
  // [5] instance entries, [6] closest hits whose attributes were fetched, [7] node steps in the instance tree (HkStats)
  uint32_t v[8] = {primary, rc.tlas, rc.blas, rc.nodes, rc.tris, rc.entries, rc.hits, rc.top_nodes};
  for (int off = 32; off > 0; off >>= 1)
    for (int k = 0; k < 8; ++k) v[k] += __shfl_down(v[k], off);
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 8; ++k)
      if (v[k]) atomicAdd(&counters[k], (unsigned long long)v[k]);
}

// The temporal-reuse tail of indirect_lit_ambient, light.wgsl:1452-1497: reproject, reject, merge the new sample into last
// frame's reservoir, shade, store.  `s` is the pixel's new sample (radiance / sample_position / sample_normal filled by the
// bounces), `pdf` the pdf of its first bounce direction.
__device__ __forceinline__ void indirect_temporal_tail(const DScene& sc, const DFrame& fr, const LightTargets& t, int index, f2 uv, f3 position,
                                                       float4 velocity_uv, uint32_t im_y, const Sample& s, float pdf) {
  const f2 previous_uv = jittered_deferred_uv(fr, uv, 0.25f) - F2(velocity_uv.x, velocity_uv.y);
  Reservoir r = load_reservoir_uv(t.previous, previous_uv, fr.rw, fr.rh);
  if (t.det_lite) t.det_to[index] = -1;  // (every geometry pixel says each frame whether and where it stores: no memset of the plane)
  if (!check_previous_reservoir(r, s) && fabsf(previous_uv.x - 0.5f) <= 0.5f && fabsf(previous_uv.y - 0.5f) <= 0.5f) {
    const int previous_index = f32_to_i32(previous_uv.x * (float)fr.rw) + fr.rw * f32_to_i32(previous_uv.y * (float)fr.rh);
    store_previous_spatial(t, index, previous_index, pack_reservoir(r));
  }
  const Surface surface = retreive_surface(sc, im_y, F2(velocity_uv.z, velocity_uv.w));
  const f3 view_direction = calculate_view(fr, position);
  f3 sample_radiance = shading(fr, view_direction, s.visible_normal, normalize(xyz(s.sample_position) - xyz(s.visible_position)), surface, s.radiance);
  float w_new = (pdf > 0.0f) ? luminance(sample_radiance) / pdf : 0.0f;
  temporal_restir(r, s, w_new, fr.max_temporal_reuse_count);

  f3 out_radiance = shading(fr, view_direction, r.s.visible_normal, normalize(xyz(r.s.sample_position) - xyz(r.s.visible_position)), surface, r.s.radiance);
  float total_lum = r.count * luminance(out_radiance);
  r.w = (total_lum > 0.0f) ? r.w_sum / total_lum : 0.0f;
  r.s.visible_position = s.visible_position;
  r.s.visible_normal = s.visible_normal;
  r.lifetime += 1.0f;

  t.variance[index] = reservoir_variance(r);
  if (fr.temporal_reuse > 0u) store_packed(t.current, index, pack_reservoir(r));
  t.render[index] = pack_f16x4(F4(out_radiance * r.w, 1.0f));
}

}  // namespace hkd
