// hk_kernels.hpp - kernel argument bundles and host launchers (implemented in kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "hk_device.hpp"

namespace hkd {

// group 1 of the reference pipelines (deferred_bindings.wgsl:3-12), row-major planes
struct GBuffer {
  float4* __restrict__ position;          // rgba32f: world xyz, clip depth
  uint32_t* __restrict__ normal;          // rgba8snorm
  float2* __restrict__ depth_gradient;    // rg32f
  float2* __restrict__ instance_material; // rg32f, id + 0.5
  float4* __restrict__ velocity_uv;       // rgba32f
};
// groups 5 + 6 for one light channel (light.wgsl:26-31,68-75; ping-pong light.rs:518-546)
struct LightTargets {
  const PackedReservoir* __restrict__ previous;  // binding 0
  PackedReservoir* current;                      // binding 1
  PackedReservoir* previous_spatial;             // binding 2
  PackedReservoir* spatial;                      // binding 3
  float* variance;                               // r32f
  uint2* render;                                 // rgba16f
};
// groups 3 + 4 of the denoise pipeline (denoise.wgsl:10-28)
struct DenoiseTargets {
  const uint2* __restrict__ albedo;     // rgba16f, full size
  const float* __restrict__ variance;   // light variance of this channel
  const uint2* __restrict__ render;     // light render of this channel
  const uint2* __restrict__ input;      // internal_texture_<level>
  uint2* __restrict__ output;           // internal_texture_<level+1> / denoise_render[channel] / internal_texture_0 (demodulation)
  float* internal_variance;
};

}  // namespace hkd

namespace hk {
void launch_prepass(hipStream_t st, const hkd::DScene& sc, const hkd::DFrame& fr, const float* inverse_view_proj, const float* view_proj,
                    const float* prev_view_proj, float jitter_x, float jitter_y, const hkd::GBuffer& g, int y0, int y1, unsigned long long* counters);
void launch_albedo(hipStream_t st, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g, void* albedo, int y0, int y1);
void launch_direct(hipStream_t st, bool emissive_lit, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g, const hkd::LightTargets& t,
                   int y0, int y1, unsigned long long* counters);
void launch_indirect(hipStream_t st, bool multiple_bounces, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g,
                     const hkd::LightTargets& t, int y0, int y1, unsigned long long* counters);
void launch_spatial(hipStream_t st, bool emissive_lit, const hkd::DScene& sc, const hkd::DFrame& fr, const hkd::GBuffer& g, const hkd::LightTargets& t,
                    int y0, int y1);
void launch_demodulation(hipStream_t st, const hkd::DFrame& fr, const hkd::DenoiseTargets& d, int y0, int y1);
void launch_denoise(hipStream_t st, int level, bool firefly, const hkd::DFrame& fr, const hkd::GBuffer& g, const hkd::DenoiseTargets& d, int y0, int y1);
void launch_tone_mapping(hipStream_t st, const hkd::DFrame& fr, const void* direct, const void* emissive, const void* indirect, void* out, int y0, int y1);
void launch_debug_math(hipStream_t st, uint32_t op, const float* x, const float* y, float* out, size_t n);
}  // namespace hk
