// kernels.hip - the compute dispatches of the path as HIP kernels for gfx950.
//
// One kernel per reference entry point; the shader-def specialisations of the
// reference (light.rs:139-152, post_process.rs:409-423) are template parameters here.
//   k_prepass            prepass.wgsl:40-100 semantics by primary rays (no rasteriser in HIP)
//   k_full_screen_albedo light.wgsl:1019-1042
//   k_direct_lit         light.wgsl:1044-1261   <EMISSIVE_LIT>
//   k_indirect           light.wgsl:1263-1498   <MULTIPLE_BOUNCES>
//   k_spatial_reuse      light.wgsl:1503-1684   <EMISSIVE_LIT>
//   k_demodulation       denoise.wgsl:135-162
//   k_denoise            denoise.wgsl:215-319   <LEVEL, FIREFLY_FILTERING>
//   k_tone_mapping       tone_mapping.wgsl:21-32
//
// Thread mapping: the reference dispatches 8x8 workgroups = one wave64 per tile.  Here a
// workgroup is 256 threads = four 8x8 wave tiles arranged 2x2 (16x16 pixels), so every wave still
// owns a compact 8x8 tile (coherent rays, coalesced 8-pixel row segments) while a CU gets four
// waves per workgroup slot.  The dispatcher hands block b to XCD b % 8.  The kernels of this file take their
// tiles in that order (pixel_of_thread<false>): the cost of a tile varies a lot (background vs geometry, trip
// counts), so spreading neighbouring tiles over all eight XCDs keeps them equally busy - measured against giving
// each XCD one contiguous eighth of the image (which the a-trous kernels do keep, for their neighbours' data in
// L2): k_indirect -17 %, k_spatial_reuse -13 %, k_prepass -12 %.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>

#include "hk_device.hpp"
#include "hk_kernels.hpp"
#include "hk_light.hpp"
#include "hk_prepass.hpp"
#include "hk_wide.hpp"

namespace hkd {

// Parked scatter stores (hk_light.hpp store_previous_spatial): pixel i left its slot in det_to[i], its record in det_pending[i]
// and - if this context dispatched it - its index in det_winner[slot] (atomicMax).  Pixels [p0, p1) outside [own0, own1) were
// dispatched by ANOTHER band (their rows of the two planes arrived with exchange A): k_join_winners lets them compete for the
// slots first, then k_resolve_scatter applies every store that is its slot's highest index - the oracle's resolution of the
// reference's write-write race (light.wgsl:1063,1092-1095,1199-1202,1456-1459), across band borders as within a band.
__global__ __launch_bounds__(256) void k_join_winners(LightTargets t, int p0, int p1, int own0, int own1) {
  const int i = p0 + (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= p1 || (i >= own0 && i < own1)) return;
  const int to = t.det_to[i];
  if (to >= 0) atomicMax(&t.det_winner[to], i);
}
__global__ __launch_bounds__(256) void k_resolve_scatter(LightTargets t, int p0, int p1) {
  const int i = p0 + (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= p1) return;
  const int to = t.det_to[i];
  if (to >= 0 && t.det_winner[to] == i) store_packed(t.previous_spatial, to, load_packed(t.det_pending, i));
}

// The light form (LightTargets::det_lite, single contexts).  Pixel p parked a store iff it went to ANOTHER pixel's slot: det_to[p] =
// that slot, det_winner[slot] = the highest such p.  The slot's owner may have stored to it as well - directly: a background pixel
// always does (light.wgsl:1058-1069, 1279-1287; the uniform-tile elision may have left the identical record in place) - the pass
// decides "background" from the depth plane exactly as the dispatch did -, a geometry pixel whose rejected history reprojects onto
// itself says so in det_to[owner] == owner.  Highest index wins (the oracle's apply_scatter): the parked store lands iff it is the
// highest parked one AND beats the owner.  Geometry pixels rewrite their det_to entry every frame and background pixels never park,
// so nothing is memset per frame; the winning thread hands the slot's det_winner back as -1.  (A note per background pixel instead of
// the depth test was tried: no faster, and wrong where a dispatch covers part of the rows.)
__device__ __forceinline__ bool lite_is_background(const DFrame& fr, const float* __restrict__ depth_plane, int pixel, bool indirect) {
  if (indirect && fr.indirect_bounces == 0u) return true;
  const int y = pixel / fr.rw, x = pixel - y * fr.rw;
  int dcx, dcy;
  jittered_deferred_coords(fr, coords_to_uv(fr, x, y), &dcx, &dcy);
  const float depth = in_bounds(dcx, dcy, fr.dw, fr.dh) ? depth_plane[dcx + fr.dw * dcy] : 0.0f;
  return depth < HK_F32_EPSILON;
}
__global__ __launch_bounds__(256) void k_resolve_scatter_lite(LightTargets t, DFrame fr, const float* __restrict__ depth_plane, int indirect, int p0, int p1) {
  const int p = p0 + (int)(blockIdx.x * 256u + threadIdx.x);
  if (p >= p1) return;
  const int to = t.det_to[p];   // (one 4-byte read is all that nearly every thread does)
  if (to < 0 || to == p || t.det_winner[to] != p) return;
  if (lite_is_background(fr, depth_plane, p, indirect != 0)) return;  // (background pixels store to their own slot only: this entry is a stale one)
  t.det_winner[to] = -1;
  const bool owner_stored = lite_is_background(fr, depth_plane, to, indirect != 0) || t.det_to[to] == to;
  if (owner_stored && to > p) return;
  store_packed(t.previous_spatial, to, load_packed(t.det_pending, p));
  if (t.m_previous_spatial) {  // (as the racing store into another tile's slot: whatever that tile's record says, it no longer holds)
    const int ty = to / t.rw, tx = to - ty * t.rw;
    atomicMax(&t.m_previous_spatial[(ty >> 3) * t.tiles_x + (tx >> 3)].poison, t.serial);
  }
}

// ------------------------------------------------------------------ prepass by primary rays (PrepassParams, primary_ray, prepass_store: hk_prepass.hpp)
// waves per SIMD the wide-walk prepass is compiled for.  Its walk is a chain of dependent fetches: one more resident wave hides more
// of them than the 8 VGPRs it spills cost - 5 waves (95 VGPRs; 5 x 28 KB of LDS stacks fit the CU) against the compiler's own 103
// VGPRs / 4 waves: primary rays of configs 3 / 4 0.905 -> 0.858 / 2.748 -> 2.544 ms (profiles/r05_sweep_ab.txt)
#ifndef HK_PREPASS_WIDE_WAVES
#define HK_PREPASS_WIDE_WAVES 5
#endif
#ifndef HK_PREPASS_FLAT_WAVES
#define HK_PREPASS_FLAT_WAVES 5   // (the one-level walk from the LDS copy, round 6: the compiler's own choice is 103 VGPRs = four waves per SIMD; capped at
                                  // 96 - two spilled - the Cornell primary rays take 0.069 -> 0.0655 ms and the frame -0.5 % over three interleaved rounds; six
                                  // waves (79 VGPRs, 20 spilled): 0.072 - profiles/r06_prepass_flat_waves_ab.txt)
#endif
template <bool COUNT, int LDS>
__global__ __launch_bounds__(256, (LDS == 4 ? HK_PREPASS_WIDE_WAVES : (LDS == 2 ? HK_PREPASS_FLAT_WAVES : 1))) void k_prepass(DScene gsc, DFrame fr, PrepassParams pp, GBuffer g, int row_begin, int row_end,
                                                  unsigned long long* counters) {
  const DScene sc = stage_scene<LDS>(gsc);
  const Pixel px = pixel_of_thread<false>(fr.dw, row_begin, row_end);
  RayCounters rc{0, 0};
  uint32_t primary = 0;
  __shared__ uint32_t wide_lds[LDS == 4 ? HK_WIDE_LDS_STACK * 256u : 1u];
  if (px.valid) {
    Ray ray = primary_ray(fr, pp, (float)px.x, (float)px.y);
    Hit hit;
#pragma unroll 1
    for (int attempt = 0;; ++attempt) {  // (one call site for the walk: a second trip only for a pixel whose nearest surface lies in front of the near plane)
      if (LDS == 4) {  // scenes in global memory, product default: the wide walk (hk_wide.hpp)
        WideStackPrivate<HK_WIDE_LDS_STACK, 96u> stack;
        stack.lds = wide_lds;
        stack.lost = pp.wide.lost;
        hit = traverse_top_wide<COUNT>(sc, pp.wide, ray, HK_F32_MAX, 0.0f, HK_DONT_EXCLUDE, stack, rc);
      } else {
        hit = traverse_top(sc, ray, HK_F32_MAX, 0.0f, HK_DONT_EXCLUDE, rc);
      }
      if (attempt != 0 || !clip_at_near_plane(fr, pp, (float)px.x, (float)px.y, ray, hit)) break;
    }
    rc.tlas = 0;  // counted as a primary ray
    primary = 1;
    rc.hits += hit.instance_index != HK_U32_MAX ? 1u : 0u;
    prepass_store(sc, fr, pp, g, px.x, px.y, ray, hit);
  }
  flush_counters<COUNT>(rc, primary, counters);
}

// ------------------------------------------------------------------ full_screen_albedo
__global__ __launch_bounds__(256) void k_full_screen_albedo(DScene sc, DFrame fr, GBuffer g, uint2* __restrict__ albedo, int row_begin, int row_end) {
  const Pixel px = pixel_of_thread(fr.dw, row_begin, row_end);
  if (!px.valid) return;
  const int idx = px.x + fr.dw * px.y;
  const float4 position_depth = g.position[idx];
  if (position_depth.w < HK_F32_EPSILON) {
    albedo[idx] = make_uint2(0u, 0u);
    return;
  }
  const f3 normal = xyz(unpack4x8snorm(g.normal[idx]));
  const uint32_t material = f32_to_u32(g.instance_material[idx].y);
  const float4 velocity_uv = g.velocity_uv[idx];
  Surface surface = retreive_surface(sc, material, F2(velocity_uv.z, velocity_uv.w));
  f3 view_direction = calculate_view(fr, xyz(position_depth));
  albedo[idx] = pack_f16x4(F4(env_brdf(view_direction, normal, surface), 1.0f));
}

// ------------------------------------------------------------------ direct_lit
// (scenes beyond LDS, LDS == 0: the walk waits on global loads 59 % of a wave's time - profiles/r03_compact16_ab.txt - and the
// compiler's free choice, 141 / 154 VGPRs, leaves 3 waves per SIMD to hide them.  4 waves (128 VGPRs, 9 / 54 spilled): direct passes of
// config 4 2.93 + 1.31 -> 2.34 + 1.17 ms, frame -4 %; 5 waves (96 VGPRs, 112 / 233 spilled) is slower again - profiles/r03_occupancy_ab.txt)
#ifndef HK_DIRECT_GLOBAL_WAVES
#define HK_DIRECT_GLOBAL_WAVES 4
#endif
// (LDS == 2, the one-level mode: since round 5 both walks of this kernel keep their occluder and take the reference's two-level walk from
// the LDS copy - hk_device.hpp traverse_top<true> - which wants 129 / 142 VGPRs where the one-level walk had 104 / 118.  Capped at
// 128 (4 waves per SIMD; 4 / 56 spilled) the Cornell frame is where it was - 0.981 / 0.984 ms against 0.986 / 0.983 with the
// one-level walk for these rays and 0.986 / 0.990 uncapped at 3 waves: profiles/r05_flat_kept_occluders_ab.txt)
#ifndef HK_DIRECT_FLAT_WAVES
#define HK_DIRECT_FLAT_WAVES 4
#endif
template <bool EMISSIVE_LIT, bool COUNT, int LDS>
__global__ __launch_bounds__(256, (LDS == 0 ? HK_DIRECT_GLOBAL_WAVES : (LDS == 2 ? HK_DIRECT_FLAT_WAVES : 1))) void k_direct_lit(DScene gsc, DFrame fr, GBuffer g, LightTargets t, int row_begin, int row_end,
                                                     unsigned long long* counters) {
  const DScene sc = stage_scene<LDS>(gsc);
  const Pixel px = pixel_of_thread<false>(fr.rw, row_begin, row_end);
  RayCounters rc{0, 0};
  __shared__ uint4 tile_lds[4][HK_TILE_LDS_UINT4];
  PackedReservoir out = pack_reservoir(zero_reservoir());  // what goes to t.current at the end (and, for background pixels, to both spatial buffers)
  bool write_current = false, background = false;
  if (px.valid) {
    const int x = px.x, y = px.y;
    const int index = x + fr.rw * y;
    const f2 uv = coords_to_uv(fr, x, y);
    Sample s = zero_sample();
    int dcx, dcy;
    jittered_deferred_coords(fr, uv, &dcx, &dcy);
    const bool din = in_bounds(dcx, dcy, fr.dw, fr.dh);
    const int didx = dcx + fr.dw * dcy;
    const float4 position_depth = din ? g.position[didx] : make_float4(0, 0, 0, 0);
    const f3 position = xyz(position_depth);
    const float depth = position_depth.w;

    if (depth < HK_F32_EPSILON) {  // light.wgsl:1058-1069
      Reservoir r = zero_reservoir();
      set_reservoir(r, s, 0.0f);
      out = pack_reservoir(r);
      write_current = true;
      background = true;
      t.variance[index] = 0.0f;
      t.render[index] = make_uint2(0u, 0u);
    } else {
      const f3 normal = xyz(unpack4x8snorm(g.normal[didx]));
      const float2 imf = g.instance_material[didx];
      const uint32_t im_x = f32_to_u32(imf.x), im_y = f32_to_u32(imf.y);
      const float4 velocity_uv = g.velocity_uv[didx];

      s.random = noise_fetch(sc, x, y, fr.number);
      s.random = fract(s.random + fr.number_golden);
      s.visible_position = F4(position, depth);
      s.visible_normal = normal;
      s.visible_instance = im_x;

      Ray ray;
      ray.origin = F3(0, 0, 0); ray.direction = F3(0, 0, 0); ray.inv_direction = F3(0, 0, 0);
      HitInfo info = empty_hit_info(F3(0, 0, 0), F3(0, 0, 0));
      info.position = F4(0, 0, 0, 0); info.instance_index = 0u; info.material_index = 0u;  // WGSL zero-init

      const f2 previous_uv = jittered_deferred_uv(fr, uv, 0.25f) - F2(velocity_uv.x, velocity_uv.y);
      Reservoir r = load_reservoir_uv(t.previous, previous_uv, fr.rw, fr.rh);
      if (t.det_lite) t.det_to[index] = -1;  // (every geometry pixel says each frame whether and where it stores: hk_light.hpp store_previous_spatial)
      const bool prev_on_screen = fabsf(previous_uv.x - 0.5f) <= 0.5f && fabsf(previous_uv.y - 0.5f) <= 0.5f;
      const int previous_index = f32_to_i32(previous_uv.x * (float)fr.rw) + fr.rw * f32_to_i32(previous_uv.y * (float)fr.rh);

      if (!check_previous_reservoir(r, s) && prev_on_screen) store_previous_spatial(t, index, previous_index, pack_reservoir(r));

      const uint32_t validate_interval = EMISSIVE_LIT ? fr.emissive_validate_interval : fr.direct_validate_interval;
      const uint32_t select_light_instance = EMISSIVE_LIT ? im_x : HK_DONT_SAMPLE_EMISSIVE;
      const bool validation_frame = (fr.number % validate_interval) == 0u;

      if (!validation_frame || r.count < 4.0f) {  // light.wgsl:1108-1153
        LightCandidate candidate = select_light_candidate(sc, fr, s.random, xyz(s.visible_position), s.visible_normal, select_light_instance, info, rc);
        ray.origin = position + normal * HK_RAY_BIAS;
        ray.direction = candidate.direction;
        ray.inv_direction = 1.0f / ray.direction;
        bool trace_condition = dot(candidate.direction, normal) > 0.0f;
        trace_condition = trace_condition && candidate.p > 0.0f;
        if (EMISSIVE_LIT) trace_condition = trace_condition && candidate.emissive_instance != HK_DONT_SAMPLE_EMISSIVE;
        if (trace_condition) {
          Hit hit = traverse_top<true>(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance, rc);  // (the occluder is KEPT: the reference's order, hk_device.hpp)
          occlude_hit_info(ray, hit, info);
          s.radiance = EMISSIVE_LIT ? input_radiance(sc, fr, ray, info, false, candidate.emissive_instance, false)
                                    : input_radiance(sc, fr, ray, info, true, HK_DONT_SAMPLE_EMISSIVE, false);
        }
        s.sample_position = info.position;
        s.sample_normal = info.normal;
        float w_new = (candidate.p > 0.0f) ? luminance(xyz(s.radiance)) / candidate.p : 0.0f;
        temporal_restir(r, s, w_new, fr.max_temporal_reuse_count);
      }

      if (validation_frame) {  // light.wgsl:1156-1214
        LightCandidate candidate = select_light_candidate(sc, fr, r.s.random, xyz(r.s.visible_position), r.s.visible_normal, select_light_instance, info, rc);
        ray.origin = xyz(s.visible_position) + s.visible_normal * HK_RAY_BIAS;
        ray.direction = normalize(xyz(r.s.sample_position) - xyz(s.visible_position));
        ray.inv_direction = 1.0f / ray.direction;
        f4 validate_radiance = F4(0, 0, 0, 0);
        bool trace_condition = dot(candidate.direction, r.s.visible_normal) > 0.0f;
        trace_condition = trace_condition && candidate.p > 0.0f;
        if (EMISSIVE_LIT) trace_condition = trace_condition && candidate.emissive_instance != HK_DONT_SAMPLE_EMISSIVE;
        if (trace_condition) {
          Hit hit = traverse_top<true>(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance, rc);  // (the occluder is KEPT: the reference's order, hk_device.hpp)
          occlude_hit_info(ray, hit, info);
          validate_radiance = EMISSIVE_LIT ? input_radiance(sc, fr, ray, info, false, candidate.emissive_instance, false)
                                           : input_radiance(sc, fr, ray, info, true, HK_DONT_SAMPLE_EMISSIVE, false);
        }
        if (r.count >= 4.0f) {
          s.random = r.s.random;
          s.sample_position = info.position;
          s.sample_normal = info.normal;
          s.radiance = validate_radiance;
        }
        float luminance_ratio = luminance(xyz(validate_radiance)) / fmax_(luminance(xyz(r.s.radiance)), 0.0001f);
        if (luminance_ratio > 1.25f || luminance_ratio < 0.8f) {
          if (prev_on_screen) store_previous_spatial(t, index, previous_index, pack_reservoir(r));
          float w_new = (candidate.p > 0.0f) ? luminance(xyz(s.radiance)) / candidate.p : 0.0f;
          set_reservoir(r, s, w_new);
        }
      }

      float total_lum = r.count * luminance(xyz(r.s.radiance));
      r.w = (total_lum > 0.0f) ? r.w_sum / total_lum : 0.0f;
      r.s.visible_position = s.visible_position;
      r.s.visible_normal = s.visible_normal;
      r.lifetime += 1.0f;

      t.variance[index] = reservoir_variance(r);
      if (fr.temporal_reuse > 0u) {
        out = pack_reservoir(r);
        write_current = true;
      }

      Surface surface = retreive_surface(sc, im_y, F2(velocity_uv.z, velocity_uv.w));
      f3 view_direction = calculate_view(fr, position);
      f3 out_radiance = shading(fr, view_direction, r.s.visible_normal, normalize(xyz(r.s.sample_position) - xyz(r.s.visible_position)), surface, r.s.radiance);
      out_radiance = out_radiance * r.w;
      f3 out_color = EMISSIVE_LIT ? out_radiance : out_radiance + compute_emissive_radiance(surface.emissive);
      t.render[index] = pack_f16x4(F4(out_color, 1.0f));
    }
  }
  // the per-pixel records leave as whole cache lines (store_packed_tile); every lane of the wave gets here
  uint4* lds = tile_lds[threadIdx.x >> 6];
  bool skip_current = false, skip_spatial = false, skip_previous_spatial = false;
  if (t.m_current && __ballot(px.valid) != 0ull) {  // uniform-tile store elision (hk_kernels.hpp TileMeta); a wave entirely beyond the image edge owns no tile
    const int tile = wave_tile(px, t.tiles_x);
    const bool all_background = __ballot(px.valid && !background) == 0ull;
    if (all_background) {
      // every VALID lane holds the same record: the reservoir of a background pixel (light.wgsl:1058-1069).  Lanes beyond the image
      // edge hold something else (their initial `out`), so the id comes from the record itself, not from each lane's copy: the
      // skip decisions below must be wave-uniform - store_packed_tile is a whole-wave operation.
      Reservoir bg = zero_reservoir();
      set_reservoir(bg, zero_sample(), 0.0f);
      const unsigned long long id = record_id(pack_reservoir(bg));
      skip_current = tile_holds(t.m_current, tile, id);
      skip_spatial = tile_holds(t.m_spatial, tile, id);
      skip_previous_spatial = tile_holds(t.m_previous_spatial, tile, id);
      if (!skip_current) tile_mark(t.m_current, tile, id, 0ull, t.serial);
      if (!skip_spatial) tile_mark(t.m_spatial, tile, id, 0ull, t.serial);
      if (!skip_previous_spatial) tile_mark(t.m_previous_spatial, tile, id, 0ull, t.serial);
    } else {
      tile_unknown(t.m_current, tile);
      // its geometry pixels may store a rejected history reservoir into their OWN slot of previous_spatial (store_previous_spatial with
      // to == from poisons nothing): the tile holds no single record there any more, background pixels or not.  (A tile can
      // alternate between all-background and all-geometry with the frame parity of the deferred-texel jitter: random case 5001.)
      tile_unknown(t.m_previous_spatial, tile);
      if (__ballot(background) != 0ull) tile_unknown(t.m_spatial, tile);  // a mixed tile: its background slots are rewritten below, the others keep older records
    }
  }
  if (!skip_current) store_packed_tile(lds, t.current, fr.rw, px, out, write_current);
  if (!skip_spatial) store_packed_tile(lds, t.spatial, fr.rw, px, out, background);
  // (background pixels store to their OWN slot: direct in the racing and in the light deterministic form, parked in the full one)
  const bool park_background = t.det_winner && !t.det_lite;
  if (!skip_previous_spatial) store_packed_tile(lds, t.previous_spatial, fr.rw, px, out, background && !park_background);
  if (park_background && background) store_previous_spatial(t, px.x + fr.rw * px.y, px.x + fr.rw * px.y, out);
  flush_counters<COUNT>(rc, 0, counters);
}

// HK_PROFILE_SECTIONS (tools/section_profile.py builds a variant with it): wave-level cycle shares of the sections of
// k_indirect, read with the scalar clock so that a section entered by any lane of the wave is charged once.
#ifdef HK_PROFILE_SECTIONS
__device__ unsigned long long g_sections[16];
__device__ unsigned long long g_walk_events[8];
struct SecTimer {
  unsigned long long t0, acc[12];
  int cur;
  __device__ void start() {
    for (int i = 0; i < 12; ++i) acc[i] = 0;
    cur = 0;
    t0 = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(int next) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    acc[cur] += t - t0;
    t0 = t;
    cur = next;
  }
  __device__ void flush() {
    mark(0);
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0)  // first active lane
      for (int i = 0; i < 12; ++i) atomicAdd(&g_sections[i], acc[i]);
  }
};
#define HK_SEC(tm, i) (tm).mark(i)
#else
struct SecTimer {
  __device__ void start() {}
  __device__ void flush() {}
};
#define HK_SEC(tm, i) ((void)0)
#endif

// HK_ABLATE_WALK_TWICE (tools/section_profile.py --walk-twice): every BVH walk of k_indirect is executed twice, the
// first result discarded behind an opaque register; the time the kernel gains is what the walks cost in issue slots.
#ifdef HK_ABLATE_WALK_TWICE
#define HK_ABLATE_WALK(sc, ray, maxd, mind, excl, rc)                 \
  do {                                                                \
    Ray r2_ = (ray);                                                  \
    asm volatile("" : "+v"(r2_.origin.x));                            \
    RayCounters rc2_{0, 0};                                           \
    const Hit h2_ = traverse_top(sc, r2_, maxd, mind, excl, rc2_);    \
    asm volatile("" ::"v"(h2_.distance), "v"(h2_.primitive_index));   \
  } while (0)
#else
#define HK_ABLATE_WALK(sc, ray, maxd, mind, excl, rc) ((void)0)
#endif

// One iteration of the MULTIPLE_BOUNCES loop (light.wgsl:1313-1394) on the state a path carries from
// bounce to bounce.  Returns false when the path ends at this bounce (miss -> ambient, `break`).
struct PathState {
  f4 random;                   // bounce_sample.random
  f3 position, normal;         // bounce_sample.visible_position.xyz / visible_normal
  f3 transport;                // color_transport
  f4 radiance;                 // s.radiance
  f4 first_position;           // s.sample_position (bounce 0)
  f3 first_normal;             // s.sample_normal   (bounce 0)
  float pdf;                   // rand_sample.w of bounce 0
};
__device__ __forceinline__ bool bounce_step(const DScene& sc, const DFrame& fr, uint32_t n, PathState& p, RayCounters& rc, SecTimer& tm) {
  HK_SEC(tm, 1);
  f4 rand_sample = sample_cosine_hemisphere(F2(p.random.x, p.random.y));
  Ray ray;
  ray.origin = p.position + p.normal * HK_RAY_BIAS;
  ray.direction = mul(normal_basis(p.normal), xyz(rand_sample));
  ray.inv_direction = 1.0f / ray.direction;

  HK_SEC(tm, 2);
  HK_ABLATE_WALK(sc, ray, HK_F32_MAX, 0.0f, HK_DONT_EXCLUDE, rc);
  Hit hit = traverse_top(sc, ray, HK_F32_MAX, 0.0f, HK_DONT_EXCLUDE, rc);
  HK_SEC(tm, 3);
  rc.hits += hit.instance_index != HK_U32_MAX ? 1u : 0u;
  HitInfo info = hit_info(sc, ray, hit);
  if (n == 0u) {
    p.first_position = info.position;
    p.first_normal = info.normal;
    p.pdf = rand_sample.w;
  }
  const f3 sample_position = xyz(info.position);
  const f3 sample_normal = info.normal;

  if (hit.instance_index != HK_U32_MAX) {
    f3 out_radiance = F3(0, 0, 0);
    Surface surface = retreive_surface(sc, info.material_index, info.uv);
    surface.roughness = 1.0f;
    const uint32_t info_instance = info.instance_index;
    HK_SEC(tm, 4);
    LightCandidate candidate = select_light_candidate(sc, fr, p.random, sample_position, sample_normal, info_instance, info, rc);
    HK_SEC(tm, 5);
    const bool sample_directional = (candidate.emissive_instance == HK_DONT_SAMPLE_EMISSIVE);
    const f3 bounce_view_direction = normalize(p.position - sample_position);

    // A shadow ray only selects between two radiance values (kernels_wavefront.hip, header): both are evaluated BEFORE the
    // walk, and so is the throughput update, so the walk runs with the path state and six floats live instead of the
    // surface, the candidate and the hit info.  Same operations on the same operands, same order of the additions.
    const f3 next_transport = p.transport * env_brdf(bounce_view_direction, sample_normal, surface);
    if (dot(candidate.direction, sample_normal) > 0.0f && candidate.p > 0.0f) {
      ray.origin = sample_position + sample_normal * HK_RAY_BIAS;
      ray.direction = candidate.direction;
      ray.inv_direction = 1.0f / ray.direction;
      const f4 in_clear = input_radiance(sc, fr, ray, info, sample_directional, candidate.emissive_instance, false);
      f3 add[2];
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const f4 in_radiance = o == 0 ? in_clear : F4(0.0f, 0.0f, 0.0f, 1.0f);
        out_radiance = shading(fr, bounce_view_direction, sample_normal, ray.direction, surface, in_radiance);
        out_radiance = out_radiance / candidate.p;
        if (n > 0u) out_radiance = (rand_sample.w < 0.01f) ? F3(0, 0, 0) : out_radiance / rand_sample.w;
        float out_luminance = luminance(out_radiance);
        if (out_luminance > fr.max_indirect_luminance) out_radiance = out_radiance * fr.max_indirect_luminance / out_luminance;
        add[o] = p.transport * out_radiance;
      }
      HK_SEC(tm, 6);
      HK_ABLATE_WALK(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance, rc);
      hit = traverse_top(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance, rc);
      HK_SEC(tm, 7);
      p.radiance = p.radiance + F4(hit.instance_index != HK_U32_MAX ? add[1] : add[0], 1.0f);
    }
    p.transport = next_transport;
    p.random = fract(p.random + fr.number_golden);
    p.position = sample_position;
    p.normal = sample_normal;
    return true;
  }
  f3 out_radiance = xyz(input_radiance(sc, fr, ray, info, false, HK_DONT_SAMPLE_EMISSIVE, true));
  p.radiance = p.radiance + F4(p.transport * out_radiance, 0.0f);
  return false;
}

// ------------------------------------------------------------------ indirect_lit_ambient
// Waves per SIMD of the headline's dominant kernel (LDS scene, one-level walk).  The kernel issues VALU instructions at 44 % lane
// utilisation and waits on LDS / memory in between: a FIFTH resident wave (96 VGPRs instead of 114, 35 spilled to scratch) hides
// more than the spills cost - Cornell 1080p: the kernel 0.408 -> 0.383 ms in the frame (0.271 -> 0.262 alone), the frame
// 0.974 / 0.977 -> 0.958 / 0.958 ms, every byte unchanged (round 5, profiles/r05_indirect_waves_ab.txt; round 3 had seen -0.5 % and left it).
// (no static LDS in this kernel: five workgroups x the largest scene copy, HK_LDS_SCENE_BYTES = 32 KB, are exactly the CU's 160 KB - the
// fifth wave is resident for every scene the one-level mode accepts; tests/test_kernel_resources.py holds both facts)
#ifndef HK_INDIRECT_FLAT_WAVES
#define HK_INDIRECT_FLAT_WAVES 5
#endif
template <bool MULTIPLE_BOUNCES, bool COUNT, int LDS>
__global__ __launch_bounds__(256, (LDS == 2 && MULTIPLE_BOUNCES ? HK_INDIRECT_FLAT_WAVES : 4)) void k_indirect(DScene gsc, DFrame fr, GBuffer g, LightTargets t, int row_begin, int row_end,
                                                   unsigned long long* counters) {
  const DScene sc = stage_scene<LDS>(gsc);
  const Pixel px = pixel_of_thread<false>(fr.rw, row_begin, row_end);
  RayCounters rc{0, 0};
  SecTimer tm;
  tm.start();
  if (px.valid) {
    const int x = px.x, y = px.y;
    const int index = x + fr.rw * y;
    const f2 uv = coords_to_uv(fr, x, y);
    int dcx, dcy;
    jittered_deferred_coords(fr, uv, &dcx, &dcy);
    const bool din = in_bounds(dcx, dcy, fr.dw, fr.dh);
    const int didx = dcx + fr.dw * dcy;
    const float4 position_depth = din ? g.position[didx] : make_float4(0, 0, 0, 0);
    const f3 position = xyz(position_depth);
    const float depth = position_depth.w;

    Sample s = zero_sample();
    Reservoir r = zero_reservoir();

    const bool background = fr.indirect_bounces == 0u || depth < HK_F32_EPSILON;
    const bool all_background = __ballot(!background) == 0ull;  // among the wave's valid pixels
    const int tile = wave_tile(px, t.tiles_x);
    if (background) {  // light.wgsl:1279-1287
      const PackedReservoir pr = pack_reservoir(r);
      bool skip_current = false, skip_spatial = false, skip_previous_spatial = false;
      if (t.m_current && all_background) {  // uniform-tile store elision (hk_kernels.hpp TileMeta): every lane stores the same record
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
      const f3 normal = normalize(xyz(unpack4x8snorm(g.normal[didx])));
      const float2 imf = g.instance_material[didx];
      const uint32_t im_x = f32_to_u32(imf.x), im_y = f32_to_u32(imf.y);
      const float4 velocity_uv = g.velocity_uv[didx];

      s.random = noise_fetch(sc, x, y, fr.number);
      s.random = fract(s.random + fr.number_golden);
      s.visible_position = F4(position, depth);
      s.visible_normal = normal;
      s.visible_instance = im_x;

      Ray ray;
      float pdf = 0.0f;
      Surface surface;

      if (MULTIPLE_BOUNCES) {  // light.wgsl:1309-1394
        PathState p;
        p.random = s.random;
        p.position = xyz(s.visible_position);
        p.normal = s.visible_normal;
        p.transport = F3(1.0f, 1.0f, 1.0f);
        p.radiance = s.radiance;
        p.first_position = s.sample_position;
        p.first_normal = s.sample_normal;
        p.pdf = 0.0f;
        for (uint32_t n = 0u; n < fr.indirect_bounces && (p.transport.x > 0.01f || p.transport.y > 0.01f || p.transport.z > 0.01f); n += 1u) {
          if (!bounce_step(sc, fr, n, p, rc, tm)) break;
        }
        s.radiance = p.radiance;
        s.sample_position = p.first_position;
        s.sample_normal = p.first_normal;
        pdf = p.pdf;
        // The pixel's G-buffer record and noise sample are read AGAIN for the temporal tail instead of being carried through
        // the bounce loop (~20 registers a lane does not have: the kernel sits at the 128-VGPR ceiling with spills): the same
        // loads and operations as in the prologue give the same values.  The opaque copy of the index keeps the compiler from
        // merging the two reads.
        {
          int didx2 = didx, x2 = x, y2 = y;
          asm volatile("" : "+v"(didx2), "+v"(x2), "+v"(y2));
          const float4 pd2 = g.position[didx2];
          const float2 imf2 = g.instance_material[didx2];
          Sample s2 = zero_sample();
          s2.random = noise_fetch(sc, x2, y2, fr.number);
          s2.random = fract(s2.random + fr.number_golden);
          s2.visible_position = F4(pd2);
          s2.visible_normal = normalize(xyz(unpack4x8snorm(g.normal[didx2])));
          s2.visible_instance = f32_to_u32(imf2.x);
          s2.radiance = s.radiance;
          s2.sample_position = s.sample_position;
          s2.sample_normal = s.sample_normal;
          HK_SEC(tm, 8);
          indirect_temporal_tail(sc, fr, t, x2 + fr.rw * y2, coords_to_uv(fr, x2, y2), xyz(F4(pd2)), g.velocity_uv[didx2], f32_to_u32(imf2.y), s2, pdf);
        }
      } else {  // light.wgsl:1395-1450
        f4 rand_sample = sample_cosine_hemisphere(F2(s.random.x, s.random.y));
        ray.origin = xyz(s.visible_position) + s.visible_normal * HK_RAY_BIAS;
        ray.direction = mul(normal_basis(s.visible_normal), xyz(rand_sample));
        ray.inv_direction = 1.0f / ray.direction;
        Hit hit = traverse_top(sc, ray, HK_F32_MAX, 0.0f, HK_DONT_EXCLUDE, rc);
        rc.hits += hit.instance_index != HK_U32_MAX ? 1u : 0u;
        HitInfo info = hit_info(sc, ray, hit);
        s.sample_position = info.position;
        s.sample_normal = info.normal;
        pdf = rand_sample.w;
        if (hit.instance_index != HK_U32_MAX) {
          f3 out_radiance = F3(0, 0, 0);
          surface = retreive_surface(sc, info.material_index, info.uv);
          surface.roughness = 1.0f;
          const uint32_t info_instance = info.instance_index;
          LightCandidate candidate = select_light_candidate(sc, fr, s.random, xyz(s.sample_position), s.sample_normal, info_instance, info, rc);
          const bool sample_directional = (candidate.emissive_instance == HK_DONT_SAMPLE_EMISSIVE);
          if (dot(candidate.direction, s.sample_normal) > 0.0f && candidate.p > 0.0f) {
            ray.origin = xyz(s.sample_position) + s.sample_normal * HK_RAY_BIAS;
            ray.direction = candidate.direction;
            ray.inv_direction = 1.0f / ray.direction;
            hit = traverse_top(sc, ray, candidate.max_distance, candidate.min_distance, candidate.emissive_instance, rc);
            occlude_hit_info(ray, hit, info);
            f4 in_radiance = input_radiance(sc, fr, ray, info, sample_directional, candidate.emissive_instance, false);
            out_radiance = shading(fr, normalize(xyz(s.visible_position) - xyz(s.sample_position)), s.sample_normal, ray.direction, surface, in_radiance);
            out_radiance = out_radiance / candidate.p;
            s.radiance = s.radiance + F4(out_radiance, 1.0f);
          }
        } else {
          f3 out_radiance = xyz(input_radiance(sc, fr, ray, info, false, HK_DONT_SAMPLE_EMISSIVE, true));
          s.radiance = s.radiance + F4(out_radiance, 0.0f);
        }
      }

      // ReSTIR: temporal, light.wgsl:1452-1497 (the MULTIPLE_BOUNCES pipeline ran it above, on the re-read pixel record)
      if (!MULTIPLE_BOUNCES) {
        HK_SEC(tm, 8);
        indirect_temporal_tail(sc, fr, t, index, uv, position, velocity_uv, im_y, s, pdf);
      }
    }
    if (t.m_current && !all_background) {  // a tile with real pixels: no longer one record everywhere
      tile_unknown(t.m_current, tile);
      tile_unknown(t.m_previous_spatial, tile);  // (geometry pixels may store into their own slot of it: k_direct_lit has the long version)
      if (__ballot(background) != 0ull) tile_unknown(t.m_spatial, tile);
    }
  }
  tm.flush();
  flush_counters<COUNT>(rc, 0, counters);
}

// ------------------------------------------------------------------ spatial_reuse
// The reference caches the workgroup's own 8x8 reservoirs + depths in workgroup memory
// (light.wgsl:1500-1501,1522-1524,1584-1591); the cached values are exactly what the buffer
// loads return, so reading neighbours from L2 is result-identical.
// per-tap constants of the Fibonacci spiral (light.wgsl:1568-1571,1609-1610): functions of the tap
// index only, evaluated once on the host with the same IEEE operations (f32 divide, sqrt, multiply)
struct SpatialTaps {
  float radius[16], tap_interval[16];
  uint32_t tap_count[16];
  float march_frac[16][6];  // f32(j) / f32(tap_count + 1), j = 1..tap_count (light.wgsl:1619)
};
// One neighbour of spatial_reuse's loop (light.wgsl:1565-1660) in two halves.  spatial_tap_reaches: everything that needs no record
// of the neighbour - where the tap lands, its depth against the pixel's, the screen-space depth march; 0 if the tap is still a
// candidate, else the HK_TAP statistic it ends in.  spatial_tap_merge: the neighbour's record, its own tests, the merge.
struct SpatialPixel {
  int x, y;
  f2 uv;
  float depth, rot;
};
// The WINDOWED form of the kernel (template parameter; launch_spatial picks it for frames of many rounds of workgroups):
//  * the depths a workgroup's taps can reach, in LDS.  A pixel reads ~85 depths around itself (16 neighbours and up to five march
//    steps towards each, within 20 pixels: light.wgsl:1570,1609).  The workgroup's 16 x 16 pixels + 22 around them are a 60 x 60
//    window of the depth plane (at upscale ratio 1; fewer texels otherwise): 14 KB, filled with coalesced row loads, read with
//    ds_read_b32.  A coordinate outside the window (never, at the reference's radii) falls through to the plane: the same value.
//  * the loop over the neighbours as two loops.  In the one loop a lane whose tap ended early (out of the image, another depth,
//    occluded: half of the taps of a Cornell frame) sits masked while the rest of its wave unpacks, shades and merges.  The first loop
//    only notes the taps that reach their record (its index in LDS, 16 x 256 words per workgroup), the second walks each lane's list:
//    a wave runs it as often as its BUSIEST lane has survivors.  Same taps, same order of merges: the same bits.
// Measured (profiles/r05_spatial_reuse_ab.txt): 15 % fewer VALU wave-instructions at lane utilisation 0.84 instead of 0.71 - worth
// -7 % / -18 % on the two spatial passes of a 4K Cornell frame (config 5: the frame 4.44 -> 4.25 ms), and NOTHING at 1080p, where the
// launch is three rounds of workgroups and ends in a tail as long as a wave lives: there the plain form stays.
constexpr int HK_SPATIAL_TILE_W = 60, HK_SPATIAL_TILE_REACH = 22;
template <bool WINDOWED>
struct DepthWindow {
  const float* __restrict__ plane;
  const float* tile;
  int x0, y0, dw, dh;
  HKD float at(int x, int y) const {  // in_bounds(x, y) ? depth[x + dw * y] : 0
    if (!in_bounds(x, y, dw, dh)) return 0.0f;
    if constexpr (WINDOWED) {
      const unsigned lx = (unsigned)(x - x0), ly = (unsigned)(y - y0);
      if (lx < (unsigned)HK_SPATIAL_TILE_W && ly < (unsigned)HK_SPATIAL_TILE_W) return tile[ly * (unsigned)HK_SPATIAL_TILE_W + lx];
    }
    return plane[x + dw * y];
  }
};
template <bool WINDOWED>
HKD uint32_t spatial_tap_reaches(const DFrame& fr, const DepthWindow<WINDOWED>& g, const SpatialTaps& taps, const SpatialPixel& me, uint32_t i, int* out_x, int* out_y) {
  const float angle = HK_TAU * fract((float)i * HK_GOLDEN_RATIO + me.rot + fr.random_float_number);
  const float radius = taps.radius[i - 1u];
  float sn, cs;
  sincos_(angle, &sn, &cs);
  const f2 offset = radius * F2(cs, sn);

  const int scx = f32_to_i32(offset.x + (float)me.x), scy = f32_to_i32(offset.y + (float)me.y);
  *out_x = scx;
  *out_y = scy;
  const f2 sample_uv = coords_to_uv(fr, scx, scy);
  if (sample_uv.x < 0.0f || sample_uv.y < 0.0f || sample_uv.x > 1.0f || sample_uv.y > 1.0f) return 1u;
  int sdx, sdy;
  jittered_deferred_coords(fr, sample_uv, &sdx, &sdy);
  const float sample_depth = g.at(sdx, sdy);

  const float depth_ratio = me.depth / sample_depth;
  if (depth_ratio < 0.9f || depth_ratio > 1.1f) return 2u;

  // The tap is merged iff it passes the neighbour's own tests (non-empty, normal within 30 degrees, sample in front of the pixel)
  // AND the screen-space depth march finds the segment unoccluded (light.wgsl:1593-1631).  All of them are pure predicates, so
  // their order is free: the march - which needs nothing of the neighbour's record, only depths - goes FIRST.  On the Cornell
  // frame it rejects 36 % of the taps (the record tests: 5 %), and those no longer fetch and unpack a 64-B record
  // (tools/section_profile.py "spatial_reuse_taps").
  const float tap_interval = taps.tap_interval[i - 1u];
  const uint32_t tap_count = taps.tap_count[i - 1u];
  bool occluded = false;
  const f2 dir = normalize(offset);
  // The march's depth taps (at most 6, a wave-uniform count): all loads first, then the comparisons.  `occluded` is the OR of
  // the steps' tests, so stopping at the first hit or looking at every step gives the same answer; the wave waits for memory
  // once per neighbour instead of once per step and the data-dependent loop (a branch and a wait per step) is gone:
  // 0.353 -> 0.318 ms.  Two more steps in the same direction changed nothing and are not done: the sixteen neighbours' own
  // depths fetched eight at a time (parked in LDS), and the neighbour's record requested together with its march taps.
  // tap_offset / (width, height) goes through the frame's f64 reciprocals (quotient_by_reciprocal: the IEEE quotient unless that
  // is subnormal, which it cannot be here - a non-zero component of `dir` is >= 1e-9: the polynomial of an angle that is at
  // least 2^-25 away from the multiples of pi/2, over a tap radius of a few hundred pixels at most).
  float march_depth[6];
#pragma unroll
  for (uint32_t j = 1u; j <= 6u; j += 1u) {
    march_depth[j - 1u] = 0.0f;
    if (j <= tap_count) {
      const float tap_dist = (float)j * tap_interval;
      const f2 tap_offset = tap_dist * dir;
      const f2 tap_uv = me.uv + F2(quotient_by_reciprocal(tap_offset.x, fr.rcp_rw), quotient_by_reciprocal(tap_offset.y, fr.rcp_rh));
      int tdx, tdy;
      jittered_deferred_coords(fr, tap_uv, &tdx, &tdy);
      march_depth[j - 1u] = g.at(tdx, tdy);
    }
  }
#pragma unroll
  for (uint32_t j = 1u; j <= 6u; j += 1u) {
    if (j <= tap_count) {
      const float ref_depth = mix(me.depth, sample_depth, taps.march_frac[i - 1u][j - 1u]);
      if (march_depth[j - 1u] > ref_depth + 0.00001f) occluded = true;
    }
  }
  return occluded ? 5u : 0u;
}
template <bool EMISSIVE_LIT>
HKD uint32_t spatial_tap_merge(const ShadingSite& site, const Sample& s, Reservoir& r, const PackedReservoir& record) {
  const Reservoir q = unpack_reservoir(record);
  const bool normal_miss = dot(s.visible_normal, q.s.visible_normal) < 0.866f;
  if (q.count < HK_F32_EPSILON || normal_miss) return 3u;

  // normalize(q.s.sample_position - s.visible_position), keeping the length for the Jacobian below (compute_jacobian_shared)
  const f3 to_sample = xyz(q.s.sample_position) - xyz(s.visible_position);
  const float to_sample_length = sqrtf(dot(to_sample, to_sample));
  const f3 sample_direction = to_sample * (1.0f / to_sample_length);
  if (dot(sample_direction, s.visible_normal) < 0.0f) return 4u;

  const float jacobian = (q.s.sample_position.w > 0.5f) ? compute_jacobian_shared(q.s, sample_direction, to_sample_length) : 1.0f;
  if (EMISSIVE_LIT) {
    merge_reservoir(r, q, luminance(xyz(q.s.radiance)) / jacobian);
  } else {
    f3 out_radiance = shade(site, sample_direction, q.s.radiance);
    merge_reservoir(r, q, luminance(out_radiance) / jacobian);
  }
  return 0u;
}
#ifndef HK_SPATIAL_WGS
#define HK_SPATIAL_WGS 4  // workgroups per CU = waves per SIMD: 128 VGPRs (5 -> 102 VGPRs spills: measured slower)
#endif
template <bool EMISSIVE_LIT, bool WINDOWED>
__global__ __launch_bounds__(256, HK_SPATIAL_WGS) void k_spatial_reuse(DScene sc, DFrame fr, GBuffer g, LightTargets t, SpatialTaps taps, int row_begin, int row_end) {
  const Pixel px = pixel_of_thread<false>(fr.rw, row_begin, row_end);
  if (!WINDOWED && !px.valid) return;  // (the windowed form has a workgroup barrier ahead: its invalid lanes leave after it)
  constexpr uint32_t SPATIAL_REUSE_COUNT = EMISSIVE_LIT ? 8u : 16u;
  const int x = px.x, y = px.y;
  const int index = x + fr.rw * y;
  const f2 uv = coords_to_uv(fr, x, y);
  int dcx, dcy;
  jittered_deferred_coords(fr, uv, &dcx, &dcy);
  const bool din = in_bounds(dcx, dcy, fr.dw, fr.dh) && px.valid;  // (px.valid: always, in the plain form)
  const int didx = dcx + fr.dw * dcy;
  const float4 position_depth = din ? g.position[didx] : make_float4(0, 0, 0, 0);
  const f3 position = xyz(position_depth);
  const float depth = position_depth.w;
  // the window of the depth plane this workgroup's taps reach (every thread arrives here; a workgroup of background pixels skips the fill)
  __shared__ float depth_tile[WINDOWED ? HK_SPATIAL_TILE_W * HK_SPATIAL_TILE_W : 1];
  DepthWindow<WINDOWED> window{g.depth, depth_tile, 0, 0, fr.dw, fr.dh};
  if constexpr (WINDOWED) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int block_x0 = x - (lane & 7) - (wave & 1) * 8, block_y0 = y - (lane >> 3) - (wave >> 1) * 8;
    window.x0 = (int)floorf((float)(block_x0 - HK_SPATIAL_TILE_REACH) * (float)fr.dw / (float)fr.rw);
    window.y0 = (int)floorf((float)(block_y0 - HK_SPATIAL_TILE_REACH) * (float)fr.dh / (float)fr.rh);
    if (__syncthreads_or(px.valid && !(depth < HK_F32_EPSILON))) {
      for (int k = (int)threadIdx.x; k < HK_SPATIAL_TILE_W * HK_SPATIAL_TILE_W; k += 256) {
        const int ly = k / HK_SPATIAL_TILE_W, lx = k - ly * HK_SPATIAL_TILE_W;
        const int gx = window.x0 + lx, gy = window.y0 + ly;
        depth_tile[k] = in_bounds(gx, gy, fr.dw, fr.dh) ? g.depth[gx + fr.dw * gy] : 0.0f;
      }
      __syncthreads();
    }
    if (!px.valid) return;
  }

  // background pixels re-pack their temporal reservoir into the spatial buffer (light.wgsl:1527-1532).  Uniform-tile store
  // elision (hk_kernels.hpp TileMeta): a wave of background pixels whose input tile holds ONE record everywhere, and whose
  // output tile already holds the re-pack of exactly that record, has nothing to load or store.
  const bool background = depth < HK_F32_EPSILON;
  const bool all_background = __ballot(!background) == 0ull;  // among the wave's valid pixels
  const int tile = wave_tile(px, t.tiles_x);
  unsigned long long input_id = 0ull;
  if (t.m_current && all_background) {
    const TileMeta in = t.m_current[tile];
    if (in.valid > in.poison) {
      input_id = in.id;
      const TileMeta have = t.m_spatial[tile];
      if (have.valid > have.poison && have.src == input_id) {
        t.render[index] = make_uint2(0u, 0u);
        return;
      }
    }
  }
  Reservoir r = unpack_reservoir(load_packed(t.current, index));
  if (background) {
    const PackedReservoir pr = pack_reservoir(r);
    store_packed(t.spatial, index, pr);
    if (t.m_spatial) {
      if (all_background && input_id != 0ull) tile_mark(t.m_spatial, tile, record_id(pr), input_id, t.serial);
      else tile_unknown(t.m_spatial, tile);
    }
    t.render[index] = make_uint2(0u, 0u);
    return;
  }
  const uint32_t im_y = f32_to_u32(g.instance_material[didx].y);
  const float4 velocity_uv = g.velocity_uv[didx];
  const Surface surface = retreive_surface(sc, im_y, F2(velocity_uv.z, velocity_uv.w));
  const bool use_spatial_variance = r.count <= 4.0f;
  const f2 previous_uv = jittered_deferred_uv(fr, uv, 0.25f) - F2(velocity_uv.x, velocity_uv.y);

  Reservoir q = r;
  const Sample s = q.s;
  const float max_lifetime = (fr.max_reservoir_lifetime <= 1.0f) ? HK_F32_MAX : fr.max_reservoir_lifetime;  // light.wgsl:913-915
  if (r.lifetime <= max_lifetime) r = load_reservoir_uv(t.previous_spatial, previous_uv, fr.rw, fr.rh);

  const f3 view_direction = calculate_view(fr, position);
  const ShadingSite site = make_site(fr, view_direction, s.visible_normal, surface);  // one site, up to 18 light directions
  if (EMISSIVE_LIT) {
    merge_reservoir(r, q, luminance(xyz(q.s.radiance)));
  } else {
    f3 out_radiance = shade(site, normalize(xyz(s.sample_position) - xyz(s.visible_position)), s.radiance);
    merge_reservoir(r, q, luminance(out_radiance));
  }
  r.s.visible_position = s.visible_position;
  r.s.visible_normal = s.visible_normal;

  const float rot = dot(s.random, F4(1.0f, 1.0f, 1.0f, 1.0f));
#ifdef HK_PROFILE_SECTIONS
  // tap statistics (tools/section_profile.py): [10] taps, rejected [11] outside the image, [12] by the depth ratio, [13] empty / normal,
  // [14] facing away, [15] occluded by the depth march; per lane, flushed by its destructor at whichever `continue` / exit it leaves through
  struct TapStats {
    uint32_t n[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    __device__ ~TapStats() {
      for (int k = 0; k < 6; ++k)
        if (n[k]) atomicAdd(&g_sections[10 + k], (unsigned long long)n[k]);
    }
  } taps_;
#define HK_TAP(k) (taps_.n[k] += 1u)
#else
#define HK_TAP(k) ((void)0)
#endif
  const SpatialPixel me{x, y, uv, depth, rot};
  __shared__ uint32_t kept[WINDOWED ? 16u * 256u : 1u];
  if constexpr (WINDOWED) {  // which taps reach their record, then the survivors (the comment above DepthWindow)
    uint32_t n_kept = 0u;
    for (uint32_t i = 1u; i <= SPATIAL_REUSE_COUNT; i += 1u) {
      HK_TAP(0);
      int scx, scy;
      const uint32_t why = spatial_tap_reaches(fr, window, taps, me, i, &scx, &scy);
      if (why) { HK_TAP(why); continue; }
      kept[n_kept * 256u + threadIdx.x] = (uint32_t)(scx + fr.rw * scy);
      n_kept += 1u;
    }
    for (uint32_t k = 0u; k < n_kept; k += 1u) {
      const uint32_t why = spatial_tap_merge<EMISSIVE_LIT>(site, s, r, load_packed(t.current, (int)kept[k * 256u + threadIdx.x]));
      if (why) HK_TAP(why);
    }
  } else {
    for (uint32_t i = 1u; i <= SPATIAL_REUSE_COUNT; i += 1u) {
      HK_TAP(0);
      int scx, scy;
      uint32_t why = spatial_tap_reaches(fr, window, taps, me, i, &scx, &scy);
      if (why) { HK_TAP(why); continue; }
      why = spatial_tap_merge<EMISSIVE_LIT>(site, s, r, load_packed(t.current, scx + fr.rw * scy));
      if (why) HK_TAP(why);
    }
  }

  const float m = (float)fr.max_spatial_reuse_count;
  if (r.count > m) {
    r.w_sum *= m / r.count;
    r.w2_sum *= m / r.count;
    r.count = m;
  }
  const f3 out_radiance = shade(site, normalize(xyz(r.s.sample_position) - xyz(s.visible_position)), r.s.radiance);
  const float total_lum = EMISSIVE_LIT ? r.count * luminance(xyz(r.s.radiance)) : r.count * luminance(out_radiance);
  r.w = (total_lum > 0.0f) ? r.w_sum / total_lum : 0.0f;
  r.lifetime += 1.0f;
  store_packed(t.spatial, index, pack_reservoir(r));
  if (t.m_spatial) tile_unknown(t.m_spatial, tile);
  if (use_spatial_variance) t.variance[index] = reservoir_variance(r);
  t.render[index] = pack_f16x4(F4(r.w * out_radiance, 1.0f));
}

// (demodulation + a-trous kernels live in kernels_denoise.hip)

__global__ __launch_bounds__(256) void k_tone_mapping(DFrame fr, const uint2* __restrict__ direct, const uint2* __restrict__ emissive,
                                                       const uint2* __restrict__ indirect, uint2* __restrict__ out, int row_begin, int row_end) {  // tone_mapping.wgsl:21-32
  const Pixel px = pixel_of_thread(fr.rw, row_begin, row_end);
  if (!px.valid) return;
  const int index = px.x + fr.rw * px.y;
  f4 color = unpack_f16x4(direct[index]);
  color = color + unpack_f16x4(emissive[index]);
  if (indirect) color = color + unpack_f16x4(indirect[index]);
  const f3 c = F3(fmax_(color.x, 0.0039f), fmax_(color.y, 0.0039f), fmax_(color.z, 0.0039f));
  const float l_old = dot(c, F3(0.2126f, 0.7152f, 0.0722f));  // bevy_core_pipeline 0.9.1 reinhard_luminance
  const float l_new = l_old / (1.0f + l_old);
  const f3 rgb = c * (l_new / l_old);
  f4 o = F4(rgb, color.w);
  if (!(color.w > 0.0f)) o = F4(fr.clear_r, fr.clear_g, fr.clear_b, fr.clear_a);
  out[index] = pack_f16x4(o);
}

__global__ void k_debug_math(uint32_t op, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = (op >= 16 && op <= 19) ? 0.0f : x[i], b = y ? y[i] : 0.0f, r = 0.0f;
  switch (op) {
    case 0: r = sin_(a); break;
    case 1: r = cos_(a); break;
    case 2: r = exp_(a); break;
    case 3: r = exp2_(a); break;
    case 4: r = log2_(a); break;
    case 5: r = pow_(a, b); break;
    case 6: r = fmin_(a, b); break;
    case 7: r = fmax_(a, b); break;
    case 8: r = f16_to_f32(f32_to_f16(a)); break;
    case 9: r = a / b; break;
    case 10: r = sqrtf(a); break;
    case 11: r = saturate(a); break;
    case 12: r = clamp_(a, -1.0f, 1.0f); break;
    case 13: r = saturate(a * b); break;
    case 14: r = unpack2x16unorm(f32_to_u32(a)).x; break;        // decode of the integer a in 0..65535
    case 15: r = unsnorm8(f32_to_u32(a)); break;                 // a = the byte 0..255
    case 20: r = unorm8(f32_to_u32(a)); break;                   // a = the byte 0..255
    case 16: case 17: case 18: case 19: {  // shading()/env_brdf() on 16 floats per item: V N L base_color radiance
      const float* q = x + 16 * i;
      DFrame fr;
      fr.amb_r = fr.amb_g = fr.amb_b = 0.05f;
      Surface sf;
      sf.base_color = F4(q[9], q[10], q[11], 1.0f);
      sf.emissive = F4(0, 0, 0, 0);
      sf.reflectance = 0.5f; sf.metallic = 0.0f; sf.roughness = perceptualRoughnessToRoughness(b); sf.occlusion = 1.0f;
      f3 V = normalize(F3(q[0], q[1], q[2])), N = normalize(F3(q[3], q[4], q[5])), L = normalize(F3(q[6], q[7], q[8]));
      f3 o = (op == 19) ? env_brdf(V, N, sf) : shading(fr, V, N, L, sf, F4(q[12], q[13], q[14], q[15]));
      r = (op == 17) ? o.y : ((op == 18) ? o.z : o.x);
      break;
    }
    default: break;
  }
  out[i] = r;
}

}  // namespace hkd

// ------------------------------------------------------------------ host launchers
namespace hk {
using namespace hkd;


// LDS staging is used when the whole scene blob fits comfortably (4 workgroups per CU stay resident)
#ifdef HK_PROFILE_SECTIONS
extern "C" __attribute__((visibility("default"))) int hk_debug_read_sections(unsigned long long* out16 /* 24 values */, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(hkd::g_sections), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out16 + 16, HIP_SYMBOL(hkd::g_walk_events), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hkd::g_sections), z, sizeof(z)) != hipSuccess) return 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(hkd::g_walk_events), z, 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

static inline size_t lds_bytes_for(const DScene& sc) { return (size_t)sc.blob_f4 * 16 <= HK_LDS_SCENE_BYTES ? (size_t)sc.blob_f4 * 16 : 0; }

void launch_prepass(hipStream_t st, const DScene& sc, const DFrame& fr, const float* inverse_view_proj, const float* view_proj,
                    const float* prev_view_proj, const float4* prev_models, float jitter_x, float jitter_y, const GBuffer& g, int y0, int y1,
                    unsigned long long* counters, const WideTrees* wide) {
  if (y1 <= y0) return;
  const PrepassParams pp = make_prepass_params(inverse_view_proj, view_proj, prev_view_proj, prev_models, jitter_x, jitter_y, wide);
  dim3 grid = grid_for(fr.dw, y1 - y0);
  const size_t lds = lds_bytes_for(sc);
  // (stage_scene's MODE: 0 global, 1 LDS copy, 2 LDS copy + one-level walk, 3 global + one-level walk for the counting replay)
  const bool flat = sc.flat_mode != 0u && lds;
  if (counters && flat)
    hipLaunchKernelGGL((k_prepass<true, 3>), grid, dim3(256), 0, st, sc, fr, pp, g, y0, y1, counters);
  else if (counters && pp.wide.tlas)
    hipLaunchKernelGGL((k_prepass<true, 4>), grid, dim3(256), 0, st, sc, fr, pp, g, y0, y1, counters);
  else if (counters)
    hipLaunchKernelGGL((k_prepass<true, 0>), grid, dim3(256), 0, st, sc, fr, pp, g, y0, y1, counters);
  else if (flat)
    hipLaunchKernelGGL((k_prepass<false, 2>), grid, dim3(256), lds, st, sc, fr, pp, g, y0, y1, counters);
  else if (lds)
    hipLaunchKernelGGL((k_prepass<false, 1>), grid, dim3(256), lds, st, sc, fr, pp, g, y0, y1, counters);
  else if (pp.wide.tlas)
    hipLaunchKernelGGL((k_prepass<false, 4>), grid, dim3(256), 0, st, sc, fr, pp, g, y0, y1, counters);
  else
    hipLaunchKernelGGL((k_prepass<false, 0>), grid, dim3(256), 0, st, sc, fr, pp, g, y0, y1, counters);
}
void launch_albedo(hipStream_t st, const DScene& sc, const DFrame& fr, const GBuffer& g, void* albedo, int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_full_screen_albedo, grid_for(fr.dw, y1 - y0), dim3(256), 0, st, sc, fr, g, (uint2*)albedo, y0, y1);
}
void launch_direct(hipStream_t st, bool emissive_lit, const DScene& sc, const DFrame& fr, const GBuffer& g, const LightTargets& t, int y0, int y1,
                   unsigned long long* counters) {
  if (y1 <= y0) return;
  dim3 grid = grid_for(fr.rw, y1 - y0);
  const size_t lds = lds_bytes_for(sc);
  const bool flat = sc.flat_mode != 0u && lds;
#define HK_LAUNCH(E)                                                                                                                     \
  if (counters && flat) hipLaunchKernelGGL((k_direct_lit<E, true, 3>), grid, dim3(256), 0, st, sc, fr, g, t, y0, y1, counters);          \
  else if (counters) hipLaunchKernelGGL((k_direct_lit<E, true, 0>), grid, dim3(256), 0, st, sc, fr, g, t, y0, y1, counters);             \
  else if (flat) hipLaunchKernelGGL((k_direct_lit<E, false, 2>), grid, dim3(256), lds, st, sc, fr, g, t, y0, y1, counters);              \
  else if (lds) hipLaunchKernelGGL((k_direct_lit<E, false, 1>), grid, dim3(256), lds, st, sc, fr, g, t, y0, y1, counters);               \
  else hipLaunchKernelGGL((k_direct_lit<E, false, 0>), grid, dim3(256), 0, st, sc, fr, g, t, y0, y1, counters);
  if (emissive_lit) { HK_LAUNCH(true) } else { HK_LAUNCH(false) }
#undef HK_LAUNCH
}
// start / stop (optional): events attached to the dispatch itself (hipExtLaunchKernel) - they take the kernel's own
// begin / end timestamps without the two extra barrier packets of hipEventRecord around it
void launch_indirect(hipStream_t st, bool multiple_bounces, const DScene& sc, const DFrame& fr, const GBuffer& g, const LightTargets& t, int y0, int y1,
                     unsigned long long* counters, hipEvent_t start, hipEvent_t stop) {
  if (y1 <= y0) return;
  dim3 grid = grid_for(fr.rw, y1 - y0);
  const size_t lds = lds_bytes_for(sc);
  const bool flat = sc.flat_mode != 0u && lds;
#define HK_LAUNCH(M)                                                                                                                              \
  if (counters && flat) hipExtLaunchKernelGGL((k_indirect<M, true, 3>), grid, dim3(256), 0, st, start, stop, 0, sc, fr, g, t, y0, y1, counters);  \
  else if (counters) hipExtLaunchKernelGGL((k_indirect<M, true, 0>), grid, dim3(256), 0, st, start, stop, 0, sc, fr, g, t, y0, y1, counters);     \
  else if (flat) hipExtLaunchKernelGGL((k_indirect<M, false, 2>), grid, dim3(256), (uint32_t)lds, st, start, stop, 0, sc, fr, g, t, y0, y1, counters);   \
  else if (lds) hipExtLaunchKernelGGL((k_indirect<M, false, 1>), grid, dim3(256), (uint32_t)lds, st, start, stop, 0, sc, fr, g, t, y0, y1, counters);    \
  else hipExtLaunchKernelGGL((k_indirect<M, false, 0>), grid, dim3(256), 0, st, start, stop, 0, sc, fr, g, t, y0, y1, counters);
  if (multiple_bounces) { HK_LAUNCH(true) } else { HK_LAUNCH(false) }
#undef HK_LAUNCH
}
void launch_resolve_scatter(hipStream_t st, const LightTargets& t, int p0, int p1, int own0, int own1) {
  if (p1 <= p0) return;
  const dim3 grid((unsigned)((p1 - p0 + 255) / 256));
  if (own1 > own0 && (p0 < own0 || p1 > own1)) hipLaunchKernelGGL(k_join_winners, grid, dim3(256), 0, st, t, p0, p1, own0, own1);
  hipLaunchKernelGGL(k_resolve_scatter, grid, dim3(256), 0, st, t, p0, p1);
}
void launch_resolve_scatter_lite(hipStream_t st, const LightTargets& t, const DFrame& fr, const float* depth_plane, bool indirect, int y0, int y1) {
  if (y1 <= y0) return;
  const int p0 = y0 * fr.rw, p1 = y1 * fr.rw;
  hipLaunchKernelGGL(k_resolve_scatter_lite, dim3((unsigned)((p1 - p0 + 255) / 256)), dim3(256), 0, st, t, fr, depth_plane, indirect ? 1 : 0, p0, p1);
}
// windowed: 0 never, 1 always, -1 by the size of the launch - the windowed form pays where the launch is many rounds of workgroups
// (HK_SPATIAL_WINDOWED_MIN_TILES: sixteen rounds of 4 workgroups on 256 CUs); a 1080p frame (8 160 tiles) keeps the plain one
#ifndef HK_SPATIAL_WINDOWED_MIN_TILES
#define HK_SPATIAL_WINDOWED_MIN_TILES 16384u
#endif
bool launch_spatial(hipStream_t st, bool emissive_lit, const DScene& sc, const DFrame& fr, const GBuffer& g, const LightTargets& t, int y0, int y1, int windowed,
                    hipEvent_t start, hipEvent_t stop) {
  if (y1 <= y0) return false;
  dim3 grid = grid_for(fr.rw, y1 - y0);
  SpatialTaps taps{};
  const uint32_t count = emissive_lit ? 8u : 16u;           // light.wgsl:246-252
  const float range = emissive_lit ? 10.0f : 20.0f;
  for (uint32_t i = 1; i <= count; ++i) {
    const float radius = sqrtf((float)i / (float)count) * range;           // light.wgsl:1570
    const float interval = std::fmax(1.0f, radius / 5.0f);                  // light.wgsl:1609 (SPATIAL_REUSE_TAPS + 1 = 5)
    taps.radius[i - 1] = radius;
    taps.tap_interval[i - 1] = interval;
    taps.tap_count[i - 1] = (uint32_t)(radius / interval);                 // light.wgsl:1610
    for (uint32_t j = 1; j <= taps.tap_count[i - 1] && j <= 6u; ++j) taps.march_frac[i - 1][j - 1] = (float)j / (float)(taps.tap_count[i - 1] + 1u);
  }
  const bool window = windowed > 0 || (windowed < 0 && grid.x >= HK_SPATIAL_WINDOWED_MIN_TILES);
  if (emissive_lit && window) hipExtLaunchKernelGGL((k_spatial_reuse<true, true>), grid, dim3(256), 0, st, start, stop, 0, sc, fr, g, t, taps, y0, y1);
  else if (emissive_lit) hipExtLaunchKernelGGL((k_spatial_reuse<true, false>), grid, dim3(256), 0, st, start, stop, 0, sc, fr, g, t, taps, y0, y1);
  else if (window) hipExtLaunchKernelGGL((k_spatial_reuse<false, true>), grid, dim3(256), 0, st, start, stop, 0, sc, fr, g, t, taps, y0, y1);
  else hipExtLaunchKernelGGL((k_spatial_reuse<false, false>), grid, dim3(256), 0, st, start, stop, 0, sc, fr, g, t, taps, y0, y1);
  return window;
}
void launch_tone_mapping(hipStream_t st, const DFrame& fr, const void* direct, const void* emissive, const void* indirect, void* out, int y0, int y1) {
  if (y1 <= y0) return;
  hipLaunchKernelGGL(k_tone_mapping, grid_for(fr.rw, y1 - y0), dim3(256), 0, st, fr, (const uint2*)direct, (const uint2*)emissive,
                     (const uint2*)indirect, (uint2*)out, y0, y1);
}
void launch_debug_math(hipStream_t st, uint32_t op, const float* x, const float* y, float* out, size_t n) {
  hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, op, x, y, out, n);
}

}  // namespace hk
