// probes.hip - the measurement hooks of include/hikari_hip_debug.h: the ceilings bench.py reports beside its rooflines, measured
// in the same run on the same device ("measure the empirical HBM ceiling with a device copy/triad kernel in the same run").
//   hk_measure_hbm     streaming copy / triad                       - the HBM roof of the screen-space kernels
//   hk_measure_valu    register-only v_fma_f32 chains                - the issue roof of the ray kernels of LDS-resident scenes
//   hk_measure_gather  dependent, divergent 16-B .. 128-B gathers    - the roof of the BVH walks of scenes beyond LDS (round 4; 128 B = a
//                                                                     record of the wide walk: round 5)
// None of them touches a context's buffers; they run on its stream between frames.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "hk_internal.hpp"
#include "hk_kernels.hpp"

using namespace hk;

#define HK_HIP(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      ::hk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);        \
      return HK_E_HIP;                                                                                   \
    }                                                                                                    \
  } while (0)
// every probe: the context's device current, everything the context enqueued finished, its stream in `stream`
#define PROBE_BEGIN(c)                                   \
  CtxInfo ci;                                            \
  { const int rc_ = ctx_info(c, &ci); if (rc_) return rc_; } \
  HK_HIP(hipSetDevice(ci.device));                       \
  { const int rc_ = hk_frame_wait(c); if (rc_) return rc_; } \
  hipStream_t stream = (hipStream_t)ci.stream

namespace {

// HBM ceiling probes (hk_measure_hbm): grid-stride float4 streams, 16 B per lane per access
__global__ __launch_bounds__(256) void k_stream_copy(float4* __restrict__ a, const float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = b[i];
}
__global__ __launch_bounds__(256) void k_stream_triad(float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float s, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 x = b[i], y = c[i];
    a[i] = make_float4(fmaf(s, y.x, x.x), fmaf(s, y.y, x.y), fmaf(s, y.z, x.z), fmaf(s, y.w, x.w));
  }
}

// one-shot variants: every thread moves four float4 that are a whole grid apart (four independent 16-B loads in flight per lane,
// every wave-instruction one contiguous 1 KiB), no loop
__global__ __launch_bounds__(256) void k_stream_copy4(float4* __restrict__ a, const float4* __restrict__ b, size_t quarter) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= quarter) return;
  const float4 x0 = b[i], x1 = b[i + quarter], x2 = b[i + 2 * quarter], x3 = b[i + 3 * quarter];
  a[i] = x0; a[i + quarter] = x1; a[i + 2 * quarter] = x2; a[i + 3 * quarter] = x3;
}
__global__ __launch_bounds__(256) void k_stream_triad4(float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float s, size_t quarter) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= quarter) return;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 x = b[i + k * quarter], y = c[i + k * quarter];
    a[i + k * quarter] = make_float4(fmaf(s, y.x, x.x), fmaf(s, y.y, x.y), fmaf(s, y.z, x.z), fmaf(s, y.w, x.w));
  }
}

// the shape that reaches the chip's copy ceiling (tools/ubench.hip, profiles/r03_ubench.json: 6.2 TB/s against 4.6-5.6 for
// the looped / multi-access shapes): ONE 16-B access per lane, no loop, the grid covers the array
__global__ __launch_bounds__(256) void k_stream_copy1(float4* __restrict__ a, const float4* __restrict__ b, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = b[i];
}
__global__ __launch_bounds__(256) void k_stream_triad1(float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float s, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 x = b[i], y = c[i];
  a[i] = make_float4(fmaf(s, y.x, x.x), fmaf(s, y.y, x.y), fmaf(s, y.z, x.z), fmaf(s, y.w, x.w));
}

// VALU issue probe (hk_measure_valu): eight independent v_fma_f32 chains per lane, no memory traffic; the grid decides how
// many waves share a SIMD (one 256-thread workgroup = one wave on each of a CU's four SIMDs)
__global__ __launch_bounds__(256) void k_valu_issue(float* out, float x, float y, int iters) {
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (float)threadIdx.x + k;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  if (s == 123.456f) out[0] = s;
}


// ---- divergent dependent gathers (hk_measure_gather) ------------------------------------------------------------------------
// What a BVH walk of a scene beyond LDS asks of the memory system, with everything else taken away: every lane follows its own
// chain of dependent loads through a table far larger than any cache - the address of step k + 1 comes out of the bytes step k
// loaded - 64 unrelated addresses per wave-level load instruction, `LOADS` adjacent 16-B loads per step (2 = one 32-B node: the
// two float4 of a node step, hk_device.hpp traverse_top; 4 = one 64-B reservoir record, the gather of spatial_reuse's taps).  The table is a random permutation cycle laid out by the host's LCG, so
// no two lanes meet and no prefetcher helps.  Reported: wave-level load instructions per second and bytes per second (lanes x
// 16 B x LOADS per step) at a given number of resident waves per SIMD - the rate NO walk of that shape can exceed on this chip.
template <int LOADS>
__global__ __launch_bounds__(256) void k_gather_chase(const uint4* __restrict__ table, uint32_t n_records, uint32_t steps, uint32_t* __restrict__ sink) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  uint32_t at = (tid * 2654435761u + 12345u) % n_records;  // one start per lane, scattered
  uint32_t acc = 0u;
  for (uint32_t k = 0; k < steps; ++k) {
    const uint4 a = table[(size_t)at * LOADS];
#pragma unroll
    for (int l = 1; l < LOADS; ++l) acc += table[(size_t)at * LOADS + l].w;  // the rest of the record: adjacent 16-B loads, issued together
    acc += a.y;
    at = a.x;  // the next record: known only now
  }
  if (acc == 0x12345678u) sink[0] = at;  // (keeps the chain alive)
}
// Round 5 experiment: the same chains of 128-B records, fetched COOPERATIVELY - in instruction i the eight lanes 8a .. 8a + 7 load the
// eight 16-B pieces of the record that lane 8i + a wants: every wave-level load touches 8 lines (each covered by 8 consecutive lanes)
// instead of 64.  Lane b = 0 of a group holds the record's first piece, with the chain's next pointer: it goes back to its owner
// through LDS.  What it measures: whether the rate of divergent record fetches - 1 lane-load per clock and CU when every lane loads
// its own record (the roofs of bench.py) - is a matter of LINES per instruction (then this runs several times faster) or of lanes.
__global__ __launch_bounds__(256) void k_gather_chase_coop(const uint4* __restrict__ table, uint32_t n_records, uint32_t steps, uint32_t* __restrict__ sink) {
  __shared__ uint32_t next_of[256];
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wave_base = threadIdx.x & ~63u;
  const uint32_t a = lane >> 3, b = lane & 7u;
  uint32_t at = (tid * 2654435761u + 12345u) % n_records;
  uint32_t acc = 0u;
  for (uint32_t k = 0; k < steps; ++k) {
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t want = (uint32_t)__shfl((int)at, (int)(8u * (uint32_t)i + a));  // the record lane 8 i + a is at
      v[i] = table[(size_t)want * 8u + b];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc += v[i].w;
      if (b == 0u) next_of[wave_base + 8u * (uint32_t)i + a] = v[i].x;  // piece 0 of the record of lane 8 i + a: its next pointer
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    at = next_of[threadIdx.x];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  if (acc == 0x12345678u) sink[0] = at;
}
__global__ __launch_bounds__(256) void k_gather_fill(uint4* __restrict__ table, uint32_t n_records, uint32_t loads, uint32_t mult, uint32_t add) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n_records) return;
  // next(i) = (i * mult + add) mod 2^k with mult = 5 mod 8, add odd: a full-period LCG over the 2^k records - one cycle through all of them
  const uint32_t next = (i * mult + add) & (n_records - 1u);
  for (uint32_t l = 0; l < loads; ++l) table[(size_t)i * loads + l] = make_uint4(next, i ^ 0x9E3779B9u, l, i + l);
}

}  // namespace

extern "C" {

int hk_measure_hbm(hk_ctx* c, size_t bytes, uint32_t reps, double* copy_gbs, double* triad_gbs) {
  HK_REQUIRE(c && copy_gbs && triad_gbs && bytes >= 4096 && reps > 0, HK_E_INVALID, "bad argument");
  PROBE_BEGIN(c);
  const size_t n = bytes / 16;
  float4 *a = nullptr, *b = nullptr, *d = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = HK_OK;
  auto fail = [&](const char* what, hipError_t e) { set_error("%s failed: %s", what, hipGetErrorString(e)); rc = HK_E_HIP; };
  hipError_t e;
  if ((e = hipMalloc((void**)&a, n * 16)) != hipSuccess || (e = hipMalloc((void**)&b, n * 16)) != hipSuccess || (e = hipMalloc((void**)&d, n * 16)) != hipSuccess) fail("hipMalloc", e);
  if (!rc && ((e = hipMemsetAsync(a, 0, n * 16, stream)) != hipSuccess || (e = hipMemsetAsync(b, 0, n * 16, stream)) != hipSuccess ||
              (e = hipMemsetAsync(d, 0, n * 16, stream)) != hipSuccess)) fail("hipMemsetAsync", e);
  if (!rc && ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess)) fail("hipEventCreate", e);
  const dim3 grid(256 * 32);  // 32 workgroups per CU of grid-stride work
  const size_t quarter = n / 4;
  const dim3 grid4((unsigned)((quarter + 255) / 256));
  const dim3 grid1((unsigned)((n + 255) / 256));
  *copy_gbs = *triad_gbs = 0.0;
  // three access shapes per probe (a grid-stride loop; a one-shot launch with four independent 16-B accesses per lane; a
  // one-shot launch with ONE access per lane - the shape that reaches the guide's 6.3 TB/s): the ceiling is the best of them
  for (int pass = 0; pass < 6 && !rc; ++pass) {
    const bool triad = pass & 1;
    const int shape = pass >> 1;
    for (uint32_t k = 0; k <= reps && !rc; ++k) {  // k = 0 warms up
      if (k == 1) (void)hipEventRecord(e0, stream);
      if (shape == 0 && !triad) hipLaunchKernelGGL(k_stream_copy, grid, dim3(256), 0, stream, a, (const float4*)b, n);
      else if (shape == 0) hipLaunchKernelGGL(k_stream_triad, grid, dim3(256), 0, stream, a, (const float4*)b, (const float4*)d, 0.5f, n);
      else if (shape == 1 && !triad) hipLaunchKernelGGL(k_stream_copy4, grid4, dim3(256), 0, stream, a, (const float4*)b, quarter);
      else if (shape == 1) hipLaunchKernelGGL(k_stream_triad4, grid4, dim3(256), 0, stream, a, (const float4*)b, (const float4*)d, 0.5f, quarter);
      else if (!triad) hipLaunchKernelGGL(k_stream_copy1, grid1, dim3(256), 0, stream, a, (const float4*)b, n);
      else hipLaunchKernelGGL(k_stream_triad1, grid1, dim3(256), 0, stream, a, (const float4*)b, (const float4*)d, 0.5f, n);
    }
    (void)hipEventRecord(e1, stream);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) { fail("hipStreamSynchronize", e); break; }
    float ms = 0.0f;
    if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) { fail("hipEventElapsedTime", e); break; }
    const double moved = (double)((shape == 1 ? 4 * quarter : n) * 16);
    const double gbs = (double)(triad ? 3 : 2) * moved * reps / ((double)ms * 1e-3) / 1e9;
    if (!triad) *copy_gbs = std::max(*copy_gbs, gbs); else *triad_gbs = std::max(*triad_gbs, gbs);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  if (d) (void)hipFree(d);
  return rc;
}

int hk_measure_valu(hk_ctx* c, uint32_t iters, double ginstr_s[4]) {
  HK_REQUIRE(c && ginstr_s && iters > 0 && iters <= (1u << 20), HK_E_INVALID, "bad argument");
  PROBE_BEGIN(c);
  hipDeviceProp_t prop;
  HK_HIP(hipGetDeviceProperties(&prop, ci.device));
  const int cus = prop.multiProcessorCount;
  float* out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  HK_HIP(hipMalloc((void**)&out, 4));
  int rc = HK_OK;
  hipError_t e;
  if ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) { set_error("hipEventCreate failed: %s", hipGetErrorString(e)); rc = HK_E_HIP; }
  for (int k = 0; k < 4 && !rc; ++k) {
    const int waves_per_simd = 1 << k;
    const dim3 grid(cus * waves_per_simd);
    hipLaunchKernelGGL(k_valu_issue, grid, dim3(256), 0, stream, out, 1.0001f, 0.5f, (int)iters);  // warm-up
    (void)hipEventRecord(e0, stream);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_valu_issue, grid, dim3(256), 0, stream, out, 1.0001f, 0.5f, (int)iters);
    (void)hipEventRecord(e1, stream);
    float ms = 0.0f;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess || (e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) {
      set_error("VALU probe failed: %s", hipGetErrorString(e));
      rc = HK_E_HIP;
      break;
    }
    // wave-instructions: waves x 64 v_fma_f32 per iteration
    ginstr_s[k] = (double)cus * 4.0 * waves_per_simd * 64.0 * iters * 4.0 / ((double)ms * 1e-3) / 1e9;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return rc;
}


// hikari_hip_debug.h
int hk_measure_gather(hk_ctx* c, size_t footprint_bytes, uint32_t bytes_per_step, uint32_t waves_per_simd, uint32_t steps, uint32_t workgroups, double* gloads_s,
                      double* gbytes_s) {
  HK_REQUIRE(c && gloads_s && gbytes_s && (bytes_per_step == 16u || bytes_per_step == 32u || bytes_per_step == 64u || bytes_per_step == 128u || bytes_per_step == 129u) && waves_per_simd >= 1u && waves_per_simd <= 8u && steps >= 16u &&
                 footprint_bytes >= 4096u && footprint_bytes <= ((size_t)32 << 30), HK_E_INVALID, "bad argument");
  PROBE_BEGIN(c);
  const bool coop = bytes_per_step == 129u;  // 128-B records fetched cooperatively (k_gather_chase_coop)
  if (coop) bytes_per_step = 128u;
  uint32_t n_records = 1u;
  while ((size_t)n_records * 2u * bytes_per_step <= footprint_bytes && n_records < (1u << 30)) n_records *= 2u;  // a power of two: the LCG's period
  const uint32_t loads = bytes_per_step / 16u;
  uint4* table = nullptr;
  uint32_t* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipDeviceProp_t prop;
  HK_HIP(hipGetDeviceProperties(&prop, ci.device));
  int rc = HK_OK;
  hipError_t e;
  if ((e = hipMalloc((void**)&table, (size_t)n_records * bytes_per_step)) != hipSuccess || (e = hipMalloc((void**)&sink, 4)) != hipSuccess ||
      (e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess) {
    set_error("gather probe set-up failed: %s", hipGetErrorString(e));
    rc = HK_E_HIP;
  }
  if (!rc) {
    hipLaunchKernelGGL(k_gather_fill, dim3((n_records + 255u) / 256u), dim3(256), 0, stream, table, n_records, loads, 1664525u, 1013904223u);
    // one 256-thread workgroup = one wave on each SIMD of a CU; `workgroups` (0 = CUs x waves_per_simd) lets a sweep load only part of the chip
    const dim3 grid(workgroups ? workgroups : (unsigned)prop.multiProcessorCount * waves_per_simd);
    for (int pass = 0; pass < 2; ++pass) {  // (pass 0 warms up: page tables, clocks)
      (void)hipEventRecord(e0, stream);
      if (coop) hipLaunchKernelGGL(k_gather_chase_coop, grid, dim3(256), 0, stream, (const uint4*)table, n_records, steps, sink);
      else if (loads == 1u) hipLaunchKernelGGL(k_gather_chase<1>, grid, dim3(256), 0, stream, (const uint4*)table, n_records, steps, sink);
      else if (loads == 2u) hipLaunchKernelGGL(k_gather_chase<2>, grid, dim3(256), 0, stream, (const uint4*)table, n_records, steps, sink);
      else if (loads == 4u) hipLaunchKernelGGL(k_gather_chase<4>, grid, dim3(256), 0, stream, (const uint4*)table, n_records, steps, sink);
      else hipLaunchKernelGGL(k_gather_chase<8>, grid, dim3(256), 0, stream, (const uint4*)table, n_records, steps, sink);  // 128 B: a record of the wide walk (hk_wide.hpp)
      (void)hipEventRecord(e1, stream);
    }
    float ms = 0.0f;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess || (e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) {
      set_error("gather probe failed: %s", hipGetErrorString(e));
      rc = HK_E_HIP;
    } else {
      const double waves = 4.0 * (double)grid.x;
      *gloads_s = waves * steps * loads / ((double)ms * 1e-3) / 1e9;
      *gbytes_s = waves * 64.0 * steps * bytes_per_step / ((double)ms * 1e-3) / 1e9;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (table) (void)hipFree(table);
  if (sink) (void)hipFree(sink);
  return rc;
}

// Test hook of the numeric contract (tests/test_math_contract.py)
int hk_debug_math(hk_ctx* c, uint32_t op, const float* x, const float* y, float* out, size_t n) {
  HK_REQUIRE(c && x && out && op <= 20, HK_E_INVALID, "bad argument");
  const size_t xin = (op >= 16 && op <= 19 ? 16 : 1) * n;
  if (n == 0) return HK_OK;
  PROBE_BEGIN(c);
  float *dx = nullptr, *dy = nullptr, *dout = nullptr;
  HK_HIP(hipMalloc((void**)&dx, xin * 4));
  HK_HIP(hipMalloc((void**)&dout, n * 4));
  HK_HIP(hipMemcpy(dx, x, xin * 4, hipMemcpyHostToDevice));
  if (y) {
    HK_HIP(hipMalloc((void**)&dy, n * 4));
    HK_HIP(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice));
  }
  launch_debug_math(stream, op, dx, dy, dout, n);
  HK_HIP(hipStreamSynchronize(stream));
  HK_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(dx);
  (void)hipFree(dout);
  if (dy) (void)hipFree(dy);
  return HK_OK;
}
}  // extern "C"
