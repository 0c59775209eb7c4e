// host_logic.cpp - host-side mirrors of reference logic that need no GPU:
//   HikariSettings::default            src/lib.rs:435-455
//   FrameUniform::extract_component    src/view.rs:125-193
//   scaled render size                 src/light.rs:318-319,623-624
//   band partition + halo plan         (this repo's multi-GPU design)
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "hk_internal.hpp"

namespace hk {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

uint32_t buffer_bpp(uint32_t b) {
  if (b == HK_BUF_POSITION || b == HK_BUF_VELOCITY_UV) return 16;
  if (b == HK_BUF_NORMAL) return 4;
  if (b == HK_BUF_DEPTH_GRADIENT || b == HK_BUF_INSTANCE_MATERIAL) return 8;
  if (b == HK_BUF_ALBEDO) return 8;
  if (b >= HK_BUF_VARIANCE0 && b < HK_BUF_VARIANCE0 + 3) return 4;
  if (b >= HK_BUF_RENDER0 && b < HK_BUF_RENDER0 + 3) return 8;
  if (b >= HK_BUF_RESERVOIR0 && b < HK_BUF_RESERVOIR0 + 10) return 64;
  if (b >= HK_BUF_DENOISE_INTERNAL0 && b < HK_BUF_DENOISE_INTERNAL0 + 4) return 8;
  if (b == HK_BUF_DENOISE_INTERNAL_VARIANCE) return 4;
  if (b >= HK_BUF_DENOISE_RENDER0 && b < HK_BUF_DENOISE_RENDER0 + 3) return 8;
  if (b == HK_BUF_TONE_MAPPED || b == HK_BUF_PREVIOUS_TONE_MAPPED) return 8;
  if (b == HK_BUF_PREVIOUS_POSITION || b == HK_BUF_PREVIOUS_VELOCITY_UV) return 16;
  if (b == HK_BUF_UPSCALE_OUTPUT || b == HK_BUF_TAA_OUTPUT || b == HK_BUF_PREVIOUS_TAA_OUTPUT || b == HK_BUF_UPSCALE_SHARPENED) return 8;
  if (b >= HK_BUF_PARKED_TO0 && b < HK_BUF_PARKED_TO0 + 3) return 4;
  if (b >= HK_BUF_PARKED_RECORD0 && b < HK_BUF_PARKED_RECORD0 + 3) return 64;
  return 0;
}
bool buffer_is_full_size(uint32_t b) {
  return b <= HK_BUF_ALBEDO || (b >= HK_BUF_RESERVOIR0 && b < HK_BUF_RESERVOIR0 + 10) || b == HK_BUF_PREVIOUS_POSITION ||
         b == HK_BUF_PREVIOUS_VELOCITY_UV || b == HK_BUF_UPSCALE_SHARPENED;
}
bool buffer_is_upscaled(uint32_t b) { return b == HK_BUF_UPSCALE_OUTPUT || b == HK_BUF_TAA_OUTPUT || b == HK_BUF_PREVIOUS_TAA_OUTPUT; }

void band_rows(uint32_t height, uint32_t i, uint32_t n, uint32_t* b0, uint32_t* b1) {
  uint32_t base = height / n, rem = height % n;
  *b0 = i * base + std::min(i, rem);
  *b1 = *b0 + base + (i < rem ? 1u : 0u);
}
// Band i of n over `rows` rows when the split of the `render_rows` rows of the scaled render image is given explicitly
// (bounds[0] = 0 < bounds[1] < ... < bounds[n] = render_rows; NULL = the equal split).  Planes of another height (the window
// rows FSR1 writes) are cut where the render rows' boundaries fall in them.
void band_rows_in(const uint32_t* bounds, uint32_t render_rows, uint32_t rows, uint32_t i, uint32_t n, uint32_t* b0, uint32_t* b1) {
  if (!bounds) {
    band_rows(rows, i, n, b0, b1);
    return;
  }
  auto cut = [&](uint32_t k) -> uint32_t {
    if (k == 0) return 0u;
    if (k == n) return rows;
    return rows == render_rows ? bounds[k] : (uint32_t)(((uint64_t)bounds[k] * rows) / render_rows);
  };
  *b0 = cut(i);
  *b1 = cut(i + 1);
}
bool band_bounds_valid(const uint32_t* bounds, uint32_t n, uint32_t render_rows) {
  if (!bounds) return true;
  if (bounds[0] != 0u || bounds[n] != render_rows) return false;
  for (uint32_t k = 0; k < n; ++k)
    if (bounds[k + 1] <= bounds[k]) return false;
  return true;
}

// the finished image lives in bands, one per rank; the rank that presents it collects the others' rows.
// Which rows of `buffer` band i owns (bands partition the RENDER rows [b0, b1)):
//   render-size buffers (and the rw x rh records of a reservoir buffer)      [b0, b1)
//   SMAA Tu4x outputs (upscale_output / taa_output, two rows per render row) [2 b0, 2 b1) clamped, the last band to the end
//   FSR1 window-size outputs                                                 the boundaries scaled to the window height
//   full-size planes (G-buffer, albedo) at ratio > 1                         the window rows under [b0, b1)
int band_buffer_rows(uint32_t width, uint32_t height, float ratio, uint32_t upscale_kind, const uint32_t* bounds, uint32_t buffer, uint32_t i, uint32_t n,
                     uint32_t* y0, uint32_t* y1, uint64_t* row_bytes) {
  uint32_t rw, rh;
  int rc = hk_scaled_size(width, height, ratio, &rw, &rh);
  if (rc) return rc;
  HK_REQUIRE(buffer < HK_BUF_COUNT && buffer_bpp(buffer) && i < n && n > 0, HK_E_INVALID, "bad buffer or band");
  HK_REQUIRE(band_bounds_valid(bounds, n, rh), HK_E_INVALID, "band bounds do not fit the render size");
  const float cr = ratio < 1.0f ? 1.0f : (ratio > 2.0f ? 2.0f : ratio);  // Upscale::ratio, lib.rs:500-504
  const float s2 = (1.0f / cr) * 2.0f;                                    // post_process.rs:718-722
  const uint32_t uw = (uint32_t)ceilf((float)width * s2), uh = (uint32_t)ceilf((float)height * s2);
  uint32_t bw = rw, bh = rh;  // context.hip buffer_dims
  if (buffer_is_full_size(buffer)) { bw = width; bh = height; }
  else if (buffer_is_upscaled(buffer) && upscale_kind == HK_UPSCALE_SMAA_TU4X) { bw = uw; bh = uh; }
  else if (buffer == HK_BUF_UPSCALE_OUTPUT) { bw = width; bh = height; }
  const bool reservoir = buffer >= HK_BUF_RESERVOIR0 && buffer < HK_BUF_RESERVOIR0 + 10;
  const uint32_t rows = reservoir ? rh : bh;
  *row_bytes = (uint64_t)(reservoir ? rw : bw) * buffer_bpp(buffer);
  const bool upscaled = buffer_is_upscaled(buffer);
  const bool fsr_window = upscale_kind == HK_UPSCALE_FSR1 && (buffer == HK_BUF_UPSCALE_OUTPUT || buffer == HK_BUF_UPSCALE_SHARPENED);
  uint32_t b0, b1;
  band_rows_in(bounds, rh, rh, i, n, &b0, &b1);
  if (fsr_window) {
    band_rows_in(bounds, rh, height, i, n, y0, y1);
  } else if (rows == rh) {
    *y0 = b0;
    *y1 = b1;
  } else if (upscaled && upscale_kind == HK_UPSCALE_SMAA_TU4X) {
    *y0 = 2 * b0 < rows ? 2 * b0 : rows;
    *y1 = (b1 == rh) ? rows : (2 * b1 < rows ? 2 * b1 : rows);
  } else {
    *y0 = (uint32_t)((uint64_t)b0 * rows / rh);
    *y1 = (b1 == rh) ? rows : (uint32_t)((uint64_t)b1 * rows / rh);
  }
  return HK_OK;
}


// Kernel footprints in scaled render rows:
//  spatial_reuse reads neighbour reservoirs within SPATIAL_REUSE_RANGE px (20 indirect / 10
//  emissive, light.wgsl:246-252); its depth ray-march taps (light.wgsl:1609-1625) can land one
//  more row out after truncation, which only concerns the locally ray-cast G-buffer apron.
//  The four a-trous levels reach 8+4+2+1 = 15 rows (denoise.wgsl:101-114), the variance
//  prefilter one more (denoise.wgsl:152-160).
Aprons band_aprons(const HkSettings* st) {
  Aprons a{0, 0};
  a.spatial = st->indirect_spatial_reuse ? 21u : (st->emissive_spatial_reuse ? 11u : 0u);  // the dispatch runs regardless of bounces (light.rs:676)
  a.denoise = st->denoise ? 16 : 0;
  return a;
}

// ---- direction-threaded skip-link BVHs ------------------------------------------------------------------------------
// The reference's flat BVH (`bvh` 0.7.1 flatten_custom, mod.rs:458-459) fixes ONE depth-first order: left child first,
// whatever the ray's direction, so a closest-hit walk (light.wgsl:400-486) prunes with `t_box < hit.distance` only once it
// has stumbled on a near hit.  The same stackless walk becomes front-to-back-ish if the array is flattened with the
// children of every inner node in the order the ray meets them; that order depends only on the SIGNS of the ray
// direction, so eight flattenings of the same tree - one per direction octant - cover every ray ("multiple-threaded
// BVH"; 8 x 32 B per node is nothing next to 288 GB).  Same nodes, same boxes, same leaves: only entry / exit indices and
// the position of a node in the array change, so a walk visits the same candidates and returns the same closest hit up
// to exact ties between two triangles (which the visiting order breaks differently).
bool rethread_flat_bvh(const HkNode* nodes, uint32_t count, uint32_t oct, HkNode* out) {
  if (count == 0) return true;
  struct Frame { uint32_t begin, end, first, second, nav_out; int stage; };
  std::vector<Frame> stack;
  stack.push_back({0u, count, 0u, 0u, 0u, 0});
  uint32_t out_pos = 0;
  auto is_leaf = [&](uint32_t i) { return nodes[i].entry_index >= HK_BVH_LEAF_FLAG; };
  auto put_nav = [&](uint32_t at, uint32_t child, uint32_t exit_) {
    out[at] = nodes[child];
    out[at].entry_index = at + 1u;
    out[at].exit_index = exit_;
  };
  while (!stack.empty()) {
    Frame& f = stack.back();
    if (f.stage == 0) {
      if (f.end - f.begin == 1u) {  // a single leaf
        if (!is_leaf(f.begin) || out_pos >= count) return false;
        out[out_pos] = nodes[f.begin];
        out[out_pos].exit_index = out_pos + 1u;
        out_pos += 1u;
        stack.pop_back();
        continue;
      }
      const uint32_t a = f.begin;
      if (is_leaf(a)) return false;
      const uint32_t b = nodes[a].exit_index;
      if (!(b > a + 1u && b < f.end) || is_leaf(b) || nodes[b].exit_index != f.end || nodes[a].entry_index != a + 1u || nodes[b].entry_index != b + 1u) return false;
      // the axis along which the two child boxes are furthest apart; `lower` = the child met first by a ray travelling in + direction
      int axis = 0;
      float best = -1.0f, ca_axis = 0.0f, cb_axis = 0.0f;
      for (int k = 0; k < 3; ++k) {
        const float ca = nodes[a].min[k] + nodes[a].max[k], cb = nodes[b].min[k] + nodes[b].max[k];
        const float d = fabsf(ca - cb);
        if (d > best) { best = d; axis = k; ca_axis = ca; cb_axis = cb; }
      }
      const bool a_lower = ca_axis <= cb_axis;
      const bool negative = (oct >> axis) & 1u;
      const bool a_first = a_lower != negative;
      f.first = a_first ? a : b;
      f.second = a_first ? b : a;
      f.stage = 1;
      if (out_pos >= count) return false;
      f.nav_out = out_pos++;
      const Frame child{f.first + 1u, nodes[f.first].exit_index, 0u, 0u, 0u, 0};
      stack.push_back(child);  // (invalidates f)
    } else if (f.stage == 1) {
      put_nav(f.nav_out, f.first, out_pos);
      f.stage = 2;
      if (out_pos >= count) return false;
      f.nav_out = out_pos++;
      const Frame child{f.second + 1u, nodes[f.second].exit_index, 0u, 0u, 0u, 0};
      stack.push_back(child);
    } else {
      put_nav(f.nav_out, f.second, out_pos);
      stack.pop_back();
    }
  }
  return out_pos == count;
}

}  // namespace hk

using namespace hk;

extern "C" {

uint32_t hk_abi_version(void) { return HK_ABI_VERSION; }
const char* hk_last_error(void) { return g_error; }

int hk_settings_default(HkSettings* s) {  // lib.rs:435-455
  HK_REQUIRE(s, HK_E_INVALID, "settings is NULL");
  memset(s, 0, sizeof(*s));
  s->direct_validate_interval = 3;
  s->emissive_validate_interval = 5;
  s->max_temporal_reuse_count = 50;
  s->max_spatial_reuse_count = 800;
  s->max_reservoir_lifetime = 100.0f;
  s->solar_angle = 0.046f;
  s->indirect_bounces = 1;
  s->max_indirect_luminance = 10.0f;
  s->clear_color[0] = 0.4f;
  s->clear_color[1] = 0.4f;
  s->clear_color[2] = 0.4f;
  s->clear_color[3] = 1.0f;
  s->temporal_reuse = 1;
  s->emissive_spatial_reuse = 0;
  s->indirect_spatial_reuse = 1;
  s->denoise = 1;
  s->taa = HK_TAA_JASMINE;
  s->upscale_kind = HK_UPSCALE_SMAA_TU4X;  // Upscale::SMAA_TU_2_0, lib.rs:491-495
  s->upscale_ratio = 2.0f;
  s->upscale_sharpness = 0.0f;
  return HK_OK;
}

static float clamp_ratio(float r) { return r < 1.0f ? 1.0f : (r > 2.0f ? 2.0f : r); }  // Upscale::ratio, lib.rs:500-504

int hk_frame_from_settings(const HkSettings* s, uint32_t frame_number, HkFrame* f) {  // view.rs:125-193
  HK_REQUIRE(s && f, HK_E_INVALID, "NULL argument");
  memset(f, 0, sizeof(*f));
  static const float KERNEL[3][3] = {{0.0625f, 0.125f, 0.0625f}, {0.125f, 0.25f, 0.125f}, {0.0625f, 0.125f, 0.0625f}};
  static const float HALTON[8][4] = {
      {0.000000f, 0.000000f, 0.500000f, 0.333333f}, {0.250000f, 0.666667f, 0.750000f, 0.111111f},
      {0.125000f, 0.444444f, 0.625000f, 0.777778f}, {0.375000f, 0.222222f, 0.875000f, 0.555556f},
      {0.062500f, 0.888889f, 0.562500f, 0.037037f}, {0.312500f, 0.370370f, 0.812500f, 0.703704f},
      {0.187500f, 0.148148f, 0.687500f, 0.481481f}, {0.437500f, 0.814815f, 0.937500f, 0.259259f}};
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) f->kernel[c][r] = KERNEL[c][r];
  memcpy(f->halton, HALTON, sizeof(HALTON));
  memcpy(f->clear_color, s->clear_color, 16);
  f->number = frame_number;
  f->direct_validate_interval = s->direct_validate_interval;
  f->emissive_validate_interval = s->emissive_validate_interval;
  f->indirect_bounces = s->indirect_bounces;
  f->temporal_reuse = s->temporal_reuse ? 1u : 0u;
  f->emissive_spatial_reuse = s->emissive_spatial_reuse ? 1u : 0u;
  f->indirect_spatial_reuse = s->indirect_spatial_reuse ? 1u : 0u;
  f->max_temporal_reuse_count = s->max_temporal_reuse_count;
  f->max_spatial_reuse_count = s->max_spatial_reuse_count;
  f->max_reservoir_lifetime = s->max_reservoir_lifetime;
  f->solar_angle = s->solar_angle;
  f->max_indirect_luminance = s->max_indirect_luminance;
  f->upscale_ratio = clamp_ratio(s->upscale_ratio);
  return HK_OK;
}

int hk_scaled_size(uint32_t width, uint32_t height, float upscale_ratio, uint32_t* sw, uint32_t* sh) {  // light.rs:318-319
  HK_REQUIRE(sw && sh && width && height, HK_E_INVALID, "bad argument");
  HK_REQUIRE(std::isfinite(upscale_ratio), HK_E_INVALID, "upscale ratio is not finite");  // clamp_ratio(NaN) is NaN
  float scale = 1.0f / clamp_ratio(upscale_ratio);
  *sw = (uint32_t)ceilf(scale * (float)width);
  *sh = (uint32_t)ceilf(scale * (float)height);
  HK_REQUIRE(*sw >= 1u && *sh >= 1u, HK_E_INVALID, "scaled size is empty");
  return HK_OK;
}

int hk_band_rows(uint32_t height, uint32_t band_index, uint32_t band_count, uint32_t* row_begin, uint32_t* row_end) {
  HK_REQUIRE(row_begin && row_end && band_count > 0 && band_index < band_count && height >= band_count, HK_E_INVALID, "bad band");
  band_rows(height, band_index, band_count, row_begin, row_end);
  return HK_OK;
}

// Rows [lo,hi) of `buffer` are needed by band `band_index`; emit one op per other band that owns a
// piece of them.
static void emit(const uint32_t* bounds, uint32_t buffer, uint32_t width, uint32_t height, uint32_t lo, uint32_t hi, uint32_t band_index, uint32_t band_count,
                 HkHaloOp* ops, uint32_t* n, uint32_t cap) {
  for (uint32_t j = 0; j < band_count; ++j) {
    if (j == band_index) continue;
    uint32_t o0, o1;
    band_rows_in(bounds, height, height, j, band_count, &o0, &o1);
    uint32_t a = std::max(lo, o0), b = std::min(hi, o1);
    if (a >= b) continue;
    if (ops && *n < cap) {
      ops[*n].buffer = buffer;
      ops[*n].peer = j;
      ops[*n].row_begin = a;
      ops[*n].row_end = b;
      ops[*n].row_bytes = (uint64_t)width * buffer_bpp(buffer);
    }
    *n += 1;
  }
}

int hk_band_plan_for(uint32_t width, uint32_t height, float upscale_ratio, uint32_t band_index, uint32_t band_count, uint32_t stage_arg,
                     uint32_t frame_number, const HkSettings* st, HkHaloOp* ops, uint32_t* n_ops) {
  return hk_band_plan_bounds(width, height, upscale_ratio, nullptr, band_index, band_count, stage_arg, frame_number, st, ops, n_ops);
}

int hk_band_plan_bounds(uint32_t width, uint32_t height, float upscale_ratio, const uint32_t* bounds, uint32_t band_index, uint32_t band_count,
                        uint32_t stage_arg, uint32_t frame_number, const HkSettings* st, HkHaloOp* ops, uint32_t* n_ops) {
  const uint32_t stage = stage_arg & 0xffu, history_rows = stage_arg >> 8;  // HK_STAGE_TEMPORAL_WITH_HISTORY
  HK_REQUIRE(history_rows == 0 || stage == HK_STAGE_TEMPORAL || stage == HK_STAGE_SPATIAL || stage == HK_STAGE_ANTIALIAS, HK_E_INVALID,
             "history rows only apply to the temporal, spatial and antialias stages");
  HK_REQUIRE(st && n_ops && band_count > 0 && band_index < band_count && stage < HK_STAGE_COUNT, HK_E_INVALID, "bad argument");
  uint32_t rw, rh;
  int rc = hk_scaled_size(width, height, upscale_ratio, &rw, &rh);
  if (rc) return rc;
  HK_REQUIRE(rh >= band_count, HK_E_INVALID, "more bands than rows");  // then height >= band_count too
  HK_REQUIRE(band_bounds_valid(bounds, band_count, rh), HK_E_INVALID, "band bounds must run 0 = b[0] < b[1] < ... < b[%u] = %u (scaled render rows)", band_count, rh);
  const uint32_t cap = ops ? *n_ops : 0;
  uint32_t n = 0;
  uint32_t b0, b1;
  band_rows_in(bounds, rh, rh, band_index, band_count, &b0, &b1);
  auto lo = [&](uint32_t a) { return b0 > a ? b0 - a : 0u; };
  auto hi = [&](uint32_t a) { return std::min(rh, b1 + a); };
  // reservoir ping-pong, light.rs:376,480-481: the temporal dispatch writes buf[previous + T]
  const uint32_t previous = 1u - (frame_number % 2u);
  if (stage == HK_STAGE_TEMPORAL && history_rows > 0) {  // (whatever temporal_reuse says: the dispatches load `previous` and store to previous_spatial regardless, light.wgsl:1091-1095)
    // exchange C: what frame n reads as history = what frame n-1 wrote.  light.rs:518-546: previous = buf[current + T],
    // previous_spatial = buf[current + S] with (T, S) = (0, 4) sun, (2, 4) emissive, (6, 8) indirect.
    const uint32_t current = frame_number % 2u;
    const uint32_t temporal[3] = {0u, 2u, 6u};
    for (uint32_t t : temporal) emit(bounds, HK_BUF_RESERVOIR0 + current + t, rw, rh, lo(history_rows), hi(history_rows), band_index, band_count, ops, &n, cap);
    if (st->emissive_spatial_reuse) emit(bounds, HK_BUF_RESERVOIR0 + current + 4u, rw, rh, lo(history_rows), hi(history_rows), band_index, band_count, ops, &n, cap);
    if (st->indirect_spatial_reuse) emit(bounds, HK_BUF_RESERVOIR0 + current + 8u, rw, rh, lo(history_rows), hi(history_rows), band_index, band_count, ops, &n, cap);
  } else if (stage == HK_STAGE_SPATIAL) {
    // reservoirs are allocated at full width (light.rs:344) but indexed with the scaled width
    // (light.wgsl:1061): a "row" of the exchange is rw reservoirs
    if (st->emissive_spatial_reuse) {
      uint32_t buf = HK_BUF_RESERVOIR0 + previous + 2;
      emit(bounds, buf, rw, rh, lo(10), hi(10), band_index, band_count, ops, &n, cap);
    }
    if (st->indirect_spatial_reuse) {
      uint32_t buf = HK_BUF_RESERVOIR0 + previous + 6;
      emit(bounds, buf, rw, rh, lo(20), hi(20), band_index, band_count, ops, &n, cap);
    }
    // SURVEY 8e step 6 (HK_STAGE_SPATIAL_WITH_HISTORY): the parked scatter stores of the temporal dispatches.  A slot this band's
    // spatial pass reads lies at most history_rows rows outside it, and a pixel that stores to such a slot at most history_rows
    // rows beyond the slot: 2 x history_rows rows of the parked planes per side.  Sun (0) and emissive (1) store into the same
    // buffer (S = 4 for both, light.rs:518-546), which only the emissive spatial pass reads: both or neither.
    if (history_rows > 0) {
      const uint32_t reach = 2u * history_rows;
      auto parked = [&](uint32_t channel) {
        emit(bounds, HK_BUF_PARKED_TO0 + channel, rw, rh, lo(reach), hi(reach), band_index, band_count, ops, &n, cap);
        emit(bounds, HK_BUF_PARKED_RECORD0 + channel, rw, rh, lo(reach), hi(reach), band_index, band_count, ops, &n, cap);
      };
      if (st->emissive_spatial_reuse) {
        parked(0u);
        parked(1u);
      }
      if (st->indirect_spatial_reuse) parked(2u);
    }
  } else if (stage == HK_STAGE_POST_PROCESS) {
    if (st->denoise) {
      uint32_t nch = st->indirect_bounces == 0 ? 2u : 3u;  // post_process.rs:949-954
      for (uint32_t ch = 0; ch < nch; ++ch) {
        emit(bounds, HK_BUF_RENDER0 + ch, rw, rh, lo(15), hi(15), band_index, band_count, ops, &n, cap);
        emit(bounds, HK_BUF_VARIANCE0 + ch, rw, rh, lo(16), hi(16), band_index, band_count, ops, &n, cap);
      }
    }
  }
  else if (stage == HK_STAGE_ANTIALIAS) {
    // exchange D.  Footprints (smaa.wgsl, taa.wgsl) in rows of the scaled render image: TAA reads its input 1 row
    // around the pixel; with SMAA Tu4x that input row comes from the extrapolation of the neighbouring quad row,
    // which reads the quads 1 row around it, whose SMAA samples read tone_mapping_output 2 rows around them
    // (2x2 gather at +-2.5 output texels): 4 rows in all.  History is read at the reprojected position:
    // previous tone-mapped rows are already local for a static camera (last frame's exchange D delivered them into
    // the plane that is `previous` now); previous TAA rows are not (TAA only runs on the band) - 5-tap Catmull-Rom
    // over bilinear taps reaches 3 output rows.  history_rows (scaled render rows) extends the history planes.
    const bool smaa = st->upscale_kind == HK_UPSCALE_SMAA_TU4X, taa = st->taa == HK_TAA_JASMINE;
    const uint32_t tm = smaa ? 4u : (taa ? 1u : 0u);
    if (tm) emit(bounds, HK_BUF_TONE_MAPPED, rw, rh, lo(tm), hi(tm), band_index, band_count, ops, &n, cap);
    if (smaa && history_rows) emit(bounds, HK_BUF_PREVIOUS_TONE_MAPPED, rw, rh, lo(tm + history_rows), hi(tm + history_rows), band_index, band_count, ops, &n, cap);
    if (taa) {
      // taa_output rows are output rows: 2 per render row with SMAA Tu4x (ceil(size * 2 / ratio) of them), 1 otherwise
      const uint32_t scale = smaa ? 2u : 1u;
      const float s2 = (1.0f / clamp_ratio(upscale_ratio)) * 2.0f;
      const uint32_t tw = smaa ? (uint32_t)ceilf((float)width * s2) : rw, th = smaa ? (uint32_t)ceilf((float)height * s2) : rh;
      const uint32_t reach = 4u + scale * history_rows;
      const uint32_t need_lo = scale * b0 > reach ? scale * b0 - reach : 0u, need_hi = std::min(th, scale * b1 + reach);
      for (uint32_t j = 0; j < band_count; ++j) {
        if (j == band_index) continue;
        uint32_t o0, o1;
        band_rows_in(bounds, rh, rh, j, band_count, &o0, &o1);
        const uint32_t a = std::max(need_lo, scale * o0), b = std::min(need_hi, std::min(th, scale * o1));
        if (a >= b) continue;
        if (ops && n < cap) {
          ops[n].buffer = HK_BUF_PREVIOUS_TAA_OUTPUT;
          ops[n].peer = j;
          ops[n].row_begin = a;
          ops[n].row_end = b;
          ops[n].row_bytes = (uint64_t)tw * buffer_bpp(HK_BUF_PREVIOUS_TAA_OUTPUT);
        }
        n += 1;
      }
    }
  }
  else if (stage == HK_STAGE_UPSCALE && st->upscale_kind == HK_UPSCALE_FSR1) {
    // exchange E.  EASU (ffx_fsr1.h:315-441) of window row y reads input rows f-1..f+2 with f = floor(y * con0.y + con0.w),
    // the same two f32 operations as the kernel; the band's EASU rows are its window rows +-1 for RCAS's cross.
    uint32_t w0, w1;
    band_rows_in(bounds, rh, height, band_index, band_count, &w0, &w1);
    const uint32_t e0 = w0 > 0 ? w0 - 1 : 0, e1 = std::min(height, w1 + 1);
    const float ivy = (float)rh, osy = (float)height;
    const float con0y = ivy * (1.0f / osy), con0w = 0.5f * ivy * (1.0f / osy) - 0.5f;
    const int f_lo = (int)floorf((float)e0 * con0y + con0w) - 1, f_hi = (int)floorf((float)(e1 - 1) * con0y + con0w) + 2;
    const uint32_t need_lo = (uint32_t)std::max(f_lo, 0), need_hi = (uint32_t)std::min(f_hi, (int)rh - 1) + 1u;
    emit(bounds, st->taa == HK_TAA_JASMINE ? HK_BUF_TAA_OUTPUT : HK_BUF_TONE_MAPPED, rw, rh, need_lo, need_hi, band_index, band_count, ops, &n, cap);
  }
  if (ops && n > cap) {
    *n_ops = n;
    HK_REQUIRE(false, HK_E_INVALID, "ops array too small: need %u", n);
  }
  *n_ops = n;
  return HK_OK;
}


// ---- hk_history_rows_bound (the history halo a frame needs, derived instead of supplied) -----------------
// Everything in double on the f32 uniforms: the bound has to dominate what the kernels compute in f32 from the same matrices,
// and it ends with + 2 rows of slack (deferred-texel jitter, truncation) that the rounding differences vanish in.
namespace {
struct M4 { double m[4][4]; };  // m[row][col]
M4 from_columns(const float* c) {
  M4 r;
  for (int col = 0; col < 4; ++col)
    for (int row = 0; row < 4; ++row) r.m[row][col] = (double)c[4 * col + row];
  return r;
}
M4 mul(const M4& a, const M4& b) {
  M4 r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
struct Box3 { double lo[3], hi[3]; };
// NDC box of a world box under view_proj.  false: nothing of it is in front of the camera.  A box that reaches behind the camera
// (or the near plane) takes the whole screen and every depth up to the near plane (z = 1 in reverse-Z clip space; the prepass
// stores clip z / w and 0 is the background, prepass.wgsl:85, light.wgsl:1057).
bool ndc_box(const M4& vp, const float mn[3], const float mx[3], Box3* out) {
  bool behind = false, front = false;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int k = 0; k < 8; ++k) {
    const double p[4] = {(double)((k & 1) ? mx[0] : mn[0]), (double)((k & 2) ? mx[1] : mn[1]), (double)((k & 4) ? mx[2] : mn[2]), 1.0};
    double c[4];
    for (int i = 0; i < 4; ++i) c[i] = vp.m[i][0] * p[0] + vp.m[i][1] * p[1] + vp.m[i][2] * p[2] + vp.m[i][3];
    if (!(c[3] > 1e-6)) {
      behind = true;
      continue;
    }
    front = true;
    for (int i = 0; i < 3; ++i) {
      const double v = c[i] / c[3];
      lo[i] = std::min(lo[i], v);
      hi[i] = std::max(hi[i], v);
    }
  }
  if (!front) return false;
  for (int i = 0; i < 2; ++i) {
    out->lo[i] = behind ? -1.0 : std::max(-1.0, lo[i]);
    out->hi[i] = behind ? 1.0 : std::min(1.0, hi[i]);
    if (out->lo[i] > out->hi[i]) return false;  // beside the screen
  }
  out->lo[2] = std::max(0.0, lo[2]);
  out->hi[2] = behind ? 1.0 : std::min(1.0, hi[2]);
  if (behind) out->lo[2] = std::min(out->lo[2], 1.0);
  return out->lo[2] <= out->hi[2];
}
// Upper bound, in NDC units, of |y - (M v).y / (M v).w| over v = (x, y, z, 1) in `box`, among the v whose image is on screen.
// Returns a negative number when it cannot bound it (the denominator reaches zero inside the box where the image may be on screen).
double reprojection_bound(const M4& M, const Box3& box, int gx, int gy, int gz, double screen_slack) {
  const double* rx = M.m[0];
  const double* ry = M.m[1];
  const double* rw = M.m[3];
  double worst = 0.0;
  for (int iz = 0; iz < gz; ++iz)
    for (int iy = 0; iy < gy; ++iy)
      for (int ix = 0; ix < gx; ++ix) {
        const double x0 = box.lo[0] + (box.hi[0] - box.lo[0]) * ix / gx, x1 = box.lo[0] + (box.hi[0] - box.lo[0]) * (ix + 1) / gx;
        const double y0 = box.lo[1] + (box.hi[1] - box.lo[1]) * iy / gy, y1 = box.lo[1] + (box.hi[1] - box.lo[1]) * (iy + 1) / gy;
        const double z0 = box.lo[2] + (box.hi[2] - box.lo[2]) * iz / gz, z1 = box.lo[2] + (box.hi[2] - box.lo[2]) * (iz + 1) / gz;
        // the affine forms at the 8 corners: exact extremes over the cell
        double dmin = 1e300, dmax = -1e300, py_max = 0.0;
        double a_min = 1e300, a_max = -1e300, b_min = 1e300, b_max = -1e300, c_min = 1e300, c_max = -1e300, e_min = 1e300, e_max = -1e300;
        const double s = 1.0 + screen_slack;
        for (int k = 0; k < 8; ++k) {
          const double x = (k & 1) ? x1 : x0, y = (k & 2) ? y1 : y0, z = (k & 4) ? z1 : z0;
          const double d = rw[0] * x + rw[1] * y + rw[2] * z + rw[3];
          const double ny = ry[0] * x + ry[1] * y + ry[2] * z + ry[3];
          const double nx = rx[0] * x + rx[1] * y + rx[2] * z + rx[3];
          dmin = std::min(dmin, d);
          dmax = std::max(dmax, d);
          const double a = ny - s * d, b = ny + s * d, c = nx - s * d, e = nx + s * d;
          a_min = std::min(a_min, a); a_max = std::max(a_max, a);
          b_min = std::min(b_min, b); b_max = std::max(b_max, b);
          c_min = std::min(c_min, c); c_max = std::max(c_max, c);
          e_min = std::min(e_min, e); e_max = std::max(e_max, e);
          py_max = std::max(py_max, fabs(d + y * rw[1] - ry[1]));  // dP/dy, affine in v
        }
        // the image (ny / d, nx / d) certainly off screen (|.| > s): the temporal kernels neither load nor store there
        if (dmin > 0.0 && (a_min > 0.0 || b_max < 0.0 || c_min > 0.0 || e_max < 0.0)) continue;
        if (dmax < 0.0 && (b_min > 0.0 || a_max < 0.0 || e_min > 0.0 || c_max < 0.0)) continue;  // (behind the previous camera: the quotient flips)
        const double dabs = dmin > 0.0 ? dmin : -dmax;
        if (!(dabs > 1e-9)) return -1.0;
        // P(v) = y (M v).w - (M v).y in centred form: |P| <= |P(c)| + sum_i max|dP/dv_i| r_i
        const double cx = 0.5 * (x0 + x1), cy = 0.5 * (y0 + y1), cz = 0.5 * (z0 + z1);
        const double pc = cy * (rw[0] * cx + rw[1] * cy + rw[2] * cz + rw[3]) - (ry[0] * cx + ry[1] * cy + ry[2] * cz + ry[3]);
        const double px_max = std::max(fabs(y0 * rw[0] - ry[0]), fabs(y1 * rw[0] - ry[0]));
        const double pz_max = std::max(fabs(y0 * rw[2] - ry[2]), fabs(y1 * rw[2] - ry[2]));
        const double p = fabs(pc) + px_max * 0.5 * (x1 - x0) + py_max * 0.5 * (y1 - y0) + pz_max * 0.5 * (z1 - z0);
        worst = std::max(worst, p / dabs);
      }
  return worst;
}
}  // namespace

int hk_history_rows_bound(const HkView* view, const HkPreviousView* pview, uint32_t render_rows, const float scene_min[3], const float scene_max[3],
                          const HkMovedBox* moved, uint32_t n_moved, uint32_t* rows) {
  HK_REQUIRE(view && pview && rows && scene_min && scene_max && render_rows > 0 && (moved || n_moved == 0), HK_E_INVALID, "bad argument");
  *rows = 0;
  const bool same_view = memcmp(view->view_proj, pview->view_proj, sizeof(view->view_proj)) == 0;
  if (same_view && n_moved == 0) return HK_OK;
  for (int k = 0; k < 3; ++k)
    if (!(scene_min[k] <= scene_max[k])) return HK_OK;  // an empty scene: every pixel is background, nothing reprojects
  const M4 vp = from_columns(view->view_proj), ivp = from_columns(view->inverse_view_proj), pvp = from_columns(pview->view_proj);
  const double slack = 8.0 / (double)render_rows;  // on-screen test of the kernels: previous_uv in [0, 1], +- the deferred jitter
  double worst = 0.0;  // NDC units; a row is 2 / render_rows of them
  bool unbounded = false;
  if (!same_view) {
    Box3 box;
    if (ndc_box(vp, scene_min, scene_max, &box)) {
      const double b = reprojection_bound(mul(pvp, ivp), box, 16, 16, 8, slack);
      if (b < 0.0) unbounded = true; else worst = b;
    }
  }
  for (uint32_t i = 0; i < n_moved && !unbounded; ++i) {
    Box3 box;
    if (!ndc_box(vp, moved[i].min, moved[i].max, &box)) continue;
    // cells of about 1/8 of the screen: the centred form is second-order in the cell size
    const int gx = std::max(1, std::min(16, (int)ceil((box.hi[0] - box.lo[0]) * 4.0))), gy = std::max(1, std::min(16, (int)ceil((box.hi[1] - box.lo[1]) * 4.0)));
    const double b = reprojection_bound(mul(mul(pvp, from_columns(moved[i].previous_from_current)), ivp), box, gx, gy, 2, slack);
    if (b < 0.0) unbounded = true; else worst = std::max(worst, b);
  }
  double r = unbounded ? (double)render_rows : worst * 0.5 * (double)render_rows;
  if (!(r < (double)render_rows)) r = (double)render_rows;   // (also NaN)
  uint32_t h = (uint32_t)ceil(r) + 2u;
  h = (h + 3u) & ~3u;
  *rows = std::min(h, render_rows);
  return HK_OK;
}

int hk_bvh_rethread(const HkNode* nodes, uint32_t count, uint32_t octant, HkNode* out) {
  HK_REQUIRE(nodes && out && octant < 8u && nodes != out, HK_E_INVALID, "bad argument");
  HK_REQUIRE(rethread_flat_bvh(nodes, count, octant, out), HK_E_INVALID, "not a flat BVH in the bvh 0.7.1 flatten_custom layout");
  return HK_OK;
}

int hk_band_schedule(uint32_t width, uint32_t height, float upscale_ratio, uint32_t rank, uint32_t n_ranks, uint32_t stage,
                     uint32_t frame_number, const HkSettings* st, HkTransfer* out, uint32_t* n_out) {
  return hk_band_schedule_bounds(width, height, upscale_ratio, nullptr, rank, n_ranks, stage, frame_number, st, out, n_out);
}

int hk_band_schedule_bounds(uint32_t width, uint32_t height, float upscale_ratio, const uint32_t* bounds, uint32_t rank, uint32_t n_ranks, uint32_t stage,
                            uint32_t frame_number, const HkSettings* st, HkTransfer* out, uint32_t* n_out) {
  HK_REQUIRE(st && n_out && n_ranks > 0 && rank < n_ranks, HK_E_INVALID, "bad argument");
  const uint32_t cap = out ? *n_out : 0;
  uint32_t n = 0;
  std::vector<HkHaloOp> ops;
  for (uint32_t r = 0; r < n_ranks; ++r) {  // the fixed global order: receive plan of rank 0, 1, ...
    uint32_t k = 0;
    int rc = hk_band_plan_bounds(width, height, upscale_ratio, bounds, r, n_ranks, stage, frame_number, st, nullptr, &k);
    if (rc) return rc;
    ops.resize(k);
    if (k && (rc = hk_band_plan_bounds(width, height, upscale_ratio, bounds, r, n_ranks, stage, frame_number, st, ops.data(), &k))) return rc;
    for (uint32_t i = 0; i < k; ++i) {
      const HkHaloOp& op = ops[i];
      const bool recv = r == rank, send = op.peer == rank;
      if (!recv && !send) continue;
      if (out && n < cap) {
        out[n].buffer = op.buffer;
        out[n].peer = recv ? op.peer : r;
        out[n].is_recv = recv ? 1u : 0u;
        out[n]._pad = 0;
        out[n].offset = (uint64_t)op.row_begin * op.row_bytes;
        out[n].bytes = (uint64_t)(op.row_end - op.row_begin) * op.row_bytes;
      }
      n += 1;
    }
  }
  if (out && n > cap) {
    *n_out = n;
    HK_REQUIRE(false, HK_E_INVALID, "transfer array too small: need %u", n);
  }
  *n_out = n;
  return HK_OK;
}

// The gather as a list of transfers for `rank`: the root receives every other band's rows of `buffer` (one transfer per band, in
// band order), band r sends its rows to the root.  Same contract as hk_band_schedule: every rank derives the same global order.
int hk_band_gather_schedule(uint32_t width, uint32_t height, float upscale_ratio, uint32_t upscale_kind, const uint32_t* bounds, uint32_t rank, uint32_t n_ranks,
                            uint32_t root, uint32_t buffer, HkTransfer* out, uint32_t* n_out) {
  HK_REQUIRE(n_out && n_ranks > 0 && rank < n_ranks && root < n_ranks, HK_E_INVALID, "bad argument");
  const uint32_t cap = out ? *n_out : 0;
  uint32_t n = 0;
  for (uint32_t r = 0; r < n_ranks; ++r) {
    if (r == root || (rank != root && rank != r)) continue;
    uint32_t y0, y1;
    uint64_t row_bytes;
    const int rc = band_buffer_rows(width, height, upscale_ratio, upscale_kind, bounds, buffer, r, n_ranks, &y0, &y1, &row_bytes);
    if (rc) return rc;
    if (y1 <= y0) continue;
    if (out && n < cap) {
      out[n].buffer = buffer;
      out[n].peer = rank == root ? r : root;
      out[n].is_recv = rank == root ? 1u : 0u;
      out[n]._pad = 0;
      out[n].offset = (uint64_t)y0 * row_bytes;
      out[n].bytes = (uint64_t)(y1 - y0) * row_bytes;
    }
    n += 1;
  }
  if (out && n > cap) {
    *n_out = n;
    HK_REQUIRE(false, HK_E_INVALID, "transfer array too small: need %u", n);
  }
  *n_out = n;
  return HK_OK;
}

// the image the overlay presents (overlay.rs:226-231) for these settings: what a frame's gather moves
uint32_t hk_final_buffer(const HkSettings* st, uint32_t frame_flags) {
  if (!st || !(frame_flags & HK_FRAME_ANTIALIAS)) return HK_BUF_TONE_MAPPED;
  if (st->upscale_kind == HK_UPSCALE_FSR1) return HK_BUF_UPSCALE_SHARPENED;
  return st->taa == HK_TAA_JASMINE ? HK_BUF_TAA_OUTPUT : HK_BUF_UPSCALE_OUTPUT;
}

// Row boundaries that give every band about the same COST: cost(row) = geometry pixels in it + width x background_cost, the cost
// of a background pixel in units of a geometry pixel (<= 0: 1/16).  Measured (profiles/r03_band_balance_probe.json): a sky band of
// config 4 takes 0.3 ms, a city band 6.3 ms; in the Cornell frame, whose geometry pixels are cheap, a background pixel costs ~1/4.  row_cost
// has cost_rows entries (the rows of the plane it was counted on); the boundaries are in scaled render rows, every band gets at
// least min_rows of them.  Pure host logic, deterministic: ranks that counted the same G-buffer derive the same split.
int hk_balanced_band_bounds(const uint32_t* row_cost, uint32_t cost_rows, uint32_t width, uint32_t render_rows, uint32_t band_count, uint32_t min_rows,
                            float background_cost, uint32_t* bounds) {
  HK_REQUIRE(row_cost && bounds && cost_rows > 0 && band_count > 0 && render_rows >= band_count, HK_E_INVALID, "bad argument");
  min_rows = std::max(min_rows, 1u);
  HK_REQUIRE(std::isfinite(background_cost) && background_cost <= 1.0f, HK_E_INVALID, "background_cost is the cost of a background pixel relative to a geometry pixel (0..1]");
  const double per_row = (double)width * (background_cost > 0.0f ? (double)background_cost : 1.0 / 16.0);
  HK_REQUIRE((uint64_t)min_rows * band_count <= render_rows, HK_E_INVALID, "min_rows x band_count exceeds the %u render rows", render_rows);
  std::vector<double> prefix(render_rows + 1, 0.0);  // cost of render rows [0, k)
  for (uint32_t y = 0; y < render_rows; ++y) {
    const uint32_t src = (uint32_t)(((uint64_t)y * cost_rows) / render_rows);
    prefix[y + 1] = prefix[y] + (double)row_cost[src] + per_row;
  }
  const double total = prefix[render_rows];
  bounds[0] = 0u;
  for (uint32_t k = 1; k < band_count; ++k) {
    const double target = total * (double)k / (double)band_count;
    uint32_t y = (uint32_t)(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
    y = std::max(y, bounds[k - 1] + min_rows);                                 // at least min_rows in the band before
    y = std::min(y, render_rows - (band_count - k) * min_rows);                // ... and in every band after
    bounds[k] = y;
  }
  bounds[band_count] = render_rows;
  return HK_OK;
}

// ---- bands of equal MEASURED time (round 6) -----------------------------------------------------------------
// The split by geometry pixels (hk_balanced_band_bounds) prices every geometry pixel alike; rows of a city-class frame differ three
// times in walk length, and a band's fixed cost does not shrink with its rows.  The controller below needs no model: after some
// frames every rank contributes ONE number - the time its band took - and the boundaries move towards equal times.
//   bounds[band_count + 1]   the split in force (0 = b0 < ... < bN = render_rows)
//   band_ms[band_count]      what each band took (any unit; <= 0 or non-finite entries: no information - the split stays)
//   row_weight[render_rows]  optional prior of how cost is spread INSIDE a band (e.g. geometry pixels per row + a floor): a band's
//                            measured time is distributed over its rows in proportion; NULL = evenly
//   damping in (0, 1]        the fraction of the way to the predicted optimum a boundary moves per call (cost per row is not constant
//                            and part of a band's time is fixed cost: the prediction overshoots; 0.5 converges in 3-5 calls)
//   max_shift                rows a boundary may move per call (0 = no limit): the rows that change owner travel as one migration
//                            (hk_band_migration_plan) and every moved row costs W x 64 B per history buffer
// Deterministic, pure: every rank that evaluates it on the same numbers gets the same boundaries.
int hk_rebalanced_band_bounds(const uint32_t* bounds, const float* band_ms, uint32_t band_count, uint32_t render_rows, const float* row_weight, uint32_t min_rows,
                              uint32_t max_shift, float damping, uint32_t* out) {
  HK_REQUIRE(bounds && band_ms && out && band_count > 0 && render_rows >= band_count, HK_E_INVALID, "bad argument");
  HK_REQUIRE(band_bounds_valid(bounds, band_count, render_rows), HK_E_INVALID, "band bounds must run 0 = b[0] < b[1] < ... < b[%u] = %u", band_count, render_rows);
  min_rows = std::max(min_rows, 1u);
  HK_REQUIRE((uint64_t)min_rows * band_count <= render_rows, HK_E_INVALID, "min_rows x band_count exceeds the %u render rows", render_rows);
  HK_REQUIRE(std::isfinite(damping) && damping > 0.0f && damping <= 1.0f, HK_E_INVALID, "damping must lie in (0, 1]");
  std::copy(bounds, bounds + band_count + 1, out);
  double total = 0.0;
  for (uint32_t i = 0; i < band_count; ++i) {
    if (!(std::isfinite(band_ms[i]) && band_ms[i] > 0.0f)) return HK_OK;  // no information from some band: keep the split
    total += (double)band_ms[i];
  }
  // cumulative cost over the rows: band i's time spread over its rows by the prior
  std::vector<double> prefix(render_rows + 1, 0.0);
  for (uint32_t i = 0; i < band_count; ++i) {
    double wsum = 0.0;
    for (uint32_t y = bounds[i]; y < bounds[i + 1]; ++y) {
      const double w = row_weight ? (double)row_weight[y] : 1.0;
      wsum += (std::isfinite(w) && w > 0.0) ? w : 0.0;
    }
    const uint32_t rows = bounds[i + 1] - bounds[i];
    for (uint32_t y = bounds[i]; y < bounds[i + 1]; ++y) {
      double w = row_weight ? (double)row_weight[y] : 1.0;
      w = (std::isfinite(w) && w > 0.0) ? w : 0.0;
      const double share = wsum > 0.0 ? w / wsum : 1.0 / (double)rows;
      prefix[y + 1] = prefix[y] + (double)band_ms[i] * share;
    }
  }
  for (uint32_t k = 1; k < band_count; ++k) {
    const double target = total * (double)k / (double)band_count;
    // the row position (fractional) where the cumulative cost reaches the target
    uint32_t y = (uint32_t)(std::upper_bound(prefix.begin(), prefix.end(), target) - prefix.begin());  // first prefix > target
    y = std::min(std::max(y, 1u), render_rows);
    const double seg = prefix[y] - prefix[y - 1];
    const double pos = (double)(y - 1) + (seg > 0.0 ? (target - prefix[y - 1]) / seg : 0.0);
    double moved = (double)bounds[k] + (double)damping * (pos - (double)bounds[k]);
    if (max_shift > 0u) moved = std::min(std::max(moved, (double)bounds[k] - (double)max_shift), (double)bounds[k] + (double)max_shift);
    long r = std::lround(moved);
    r = std::max<long>(r, (long)out[k - 1] + (long)min_rows);                              // at least min_rows in the band before
    r = std::min<long>(r, (long)render_rows - (long)(band_count - k) * (long)min_rows);   // ... and in every band after
    out[k] = (uint32_t)r;
  }
  out[band_count] = render_rows;
  return HK_OK;
}

// What travels when the split changes between two frames (round 6).  A band keeps per-pixel state from frame to frame - the
// reservoirs the next frame reads as history (light.rs:518-546: the three temporal outputs, and the spatial outputs whose pass is on) -
// for the rows it OWNS; rows that change owner must reach the new owner before it renders them, or it would find its own stale
// records there.  (With the history halo of a moving camera, exchange C then delivers the rows around the NEW borders from their new
// owners, as always.)  The anti-aliasing tail keeps more state (previous tone-mapped / TAA planes, previous G-buffer planes): a host
// that runs it keeps the split fixed or treats a re-split as a cut.
//   `next_frame_number` = the frame that will be rendered with `new_bounds` (the ping-pong parity of the buffers it reads).
// ops (capacity *n_ops in, count out): rows [row_begin, row_end) of `buffer` that band `band_index` receives from `peer`, the band
// that owned them under `old_bounds`.  NULL bounds = the equal split.
int hk_band_migration_plan(uint32_t width, uint32_t height, float upscale_ratio, const uint32_t* old_bounds, const uint32_t* new_bounds, uint32_t band_index,
                           uint32_t band_count, uint32_t next_frame_number, const HkSettings* st, HkHaloOp* ops, uint32_t* n_ops) {
  HK_REQUIRE(st && n_ops && band_count > 0 && band_index < band_count, HK_E_INVALID, "bad argument");
  uint32_t rw, rh;
  int rc = hk_scaled_size(width, height, upscale_ratio, &rw, &rh);
  if (rc) return rc;
  HK_REQUIRE(rh >= band_count, HK_E_INVALID, "more bands than rows");
  HK_REQUIRE(band_bounds_valid(old_bounds, band_count, rh) && band_bounds_valid(new_bounds, band_count, rh), HK_E_INVALID, "band bounds do not fit the render size");
  const uint32_t cap = ops ? *n_ops : 0;
  uint32_t n = 0;
  uint32_t n0, n1, o0, o1;
  band_rows_in(new_bounds, rh, rh, band_index, band_count, &n0, &n1);
  band_rows_in(old_bounds, rh, rh, band_index, band_count, &o0, &o1);
  const uint32_t current = next_frame_number % 2u;  // as exchange C: previous = buf[current + T], previous_spatial = buf[current + S]
  std::vector<uint32_t> buffers = {HK_BUF_RESERVOIR0 + current + 0u, HK_BUF_RESERVOIR0 + current + 2u, HK_BUF_RESERVOIR0 + current + 6u};
  if (st->emissive_spatial_reuse) buffers.push_back(HK_BUF_RESERVOIR0 + current + 4u);
  if (st->indirect_spatial_reuse) buffers.push_back(HK_BUF_RESERVOIR0 + current + 8u);
  for (uint32_t buffer : buffers)
    for (uint32_t j = 0; j < band_count; ++j) {
      if (j == band_index) continue;
      uint32_t p0, p1;
      band_rows_in(old_bounds, rh, rh, j, band_count, &p0, &p1);
      const uint32_t a = std::max(n0, p0), b = std::min(n1, p1);   // rows this band owns now and band j owned before
      if (a >= b) continue;
      if (ops && n < cap) {
        ops[n].buffer = buffer;
        ops[n].peer = j;
        ops[n].row_begin = a;
        ops[n].row_end = b;
        ops[n].row_bytes = (uint64_t)rw * buffer_bpp(buffer);
      }
      n += 1;
    }
  (void)o0;
  (void)o1;
  if (ops && n > cap) {
    *n_ops = n;
    HK_REQUIRE(false, HK_E_INVALID, "ops array too small: need %u", n);
  }
  *n_ops = n;
  return HK_OK;
}

// ... as ONE global order of sends and receives that every rank derives identically (hk_band_schedule's contract)
int hk_band_migration_schedule(uint32_t width, uint32_t height, float upscale_ratio, const uint32_t* old_bounds, const uint32_t* new_bounds, uint32_t rank,
                               uint32_t n_ranks, uint32_t next_frame_number, const HkSettings* st, HkTransfer* out, uint32_t* n_out) {
  HK_REQUIRE(st && n_out && n_ranks > 0 && rank < n_ranks, HK_E_INVALID, "bad argument");
  const uint32_t cap = out ? *n_out : 0;
  uint32_t n = 0;
  std::vector<HkHaloOp> ops;
  for (uint32_t r = 0; r < n_ranks; ++r) {
    uint32_t k = 0;
    int rc = hk_band_migration_plan(width, height, upscale_ratio, old_bounds, new_bounds, r, n_ranks, next_frame_number, st, nullptr, &k);
    if (rc) return rc;
    ops.resize(k);
    if (k && (rc = hk_band_migration_plan(width, height, upscale_ratio, old_bounds, new_bounds, r, n_ranks, next_frame_number, st, ops.data(), &k))) return rc;
    for (uint32_t i = 0; i < k; ++i) {
      const HkHaloOp& op = ops[i];
      const bool recv = r == rank, send = op.peer == rank;
      if (!recv && !send) continue;
      if (out && n < cap) {
        out[n].buffer = op.buffer;
        out[n].peer = recv ? op.peer : r;
        out[n].is_recv = recv ? 1u : 0u;
        out[n]._pad = 0;
        out[n].offset = (uint64_t)op.row_begin * op.row_bytes;
        out[n].bytes = (uint64_t)(op.row_end - op.row_begin) * op.row_bytes;
      }
      n += 1;
    }
  }
  if (out && n > cap) {
    *n_out = n;
    HK_REQUIRE(false, HK_E_INVALID, "transfer array too small: need %u", n);
  }
  *n_out = n;
  return HK_OK;
}

}  // extern "C"
