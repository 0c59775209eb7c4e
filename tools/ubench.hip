// Stand-alone ceiling probes for MI355X (gfx950): which access shape reaches the HBM copy ceiling, and at what rate a CU's
// SIMDs issue wave64 VALU instructions.  The winners are what hk_measure_hbm / hk_measure_valu (context.hip) run inside
// bench.py; this file is the sweep that picked them (profiles/r03_ubench.json).
//   hipcc --offload-arch=gfx950 -O3 -o build_ab/ubench tools/ubench.hip && build_ab/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

// ---------------------------------------------------------------- HBM
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void k_copy_stride(f4* __restrict__ a, const f4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(&b[i]), &a[i]);
    else a[i] = b[i];
  }
}
// one-shot: U independent 16-B accesses per lane, a whole grid apart
template <int U, int NT>
__global__ __launch_bounds__(256) void k_copy_once(f4* __restrict__ a, const f4* __restrict__ b, size_t part) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= part) return;
  f4 x[U];
#pragma unroll
  for (int k = 0; k < U; ++k) x[k] = NT ? __builtin_nontemporal_load(&b[i + k * part]) : b[i + k * part];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    if (NT) __builtin_nontemporal_store(x[k], &a[i + k * part]);
    else a[i + k * part] = x[k];
  }
}
// block-contiguous: workgroup g owns one contiguous chunk, U x 4 KiB per iteration
template <int U, int NT>
__global__ __launch_bounds__(256) void k_copy_chunk(f4* __restrict__ a, const f4* __restrict__ b, size_t per_block) {
  const size_t base = (size_t)blockIdx.x * per_block;
  for (size_t i = threadIdx.x; i < per_block; i += 256 * U) {
    f4 x[U];
#pragma unroll
    for (int k = 0; k < U; ++k) x[k] = NT ? __builtin_nontemporal_load(&b[base + i + k * 256]) : b[base + i + k * 256];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (NT) __builtin_nontemporal_store(x[k], &a[base + i + k * 256]);
      else a[base + i + k * 256] = x[k];
    }
  }
}
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ b, size_t n, float* sink) {
  f4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const f4 x = b[i];
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  if (s.x + s.y + s.z + s.w == 123.456f) *sink = 1.0f;
}
__global__ __launch_bounds__(256) void k_fill(f4* __restrict__ a, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = f4{v, v, v, v};
}

// ---------------------------------------------------------------- VALU issue
// CHAINS independent v_fma_f32 chains per lane, ITERS x 8 rounds; no memory traffic.  launch_bounds(256, W) only bounds the
// registers: the number of resident waves per SIMD is set by the grid (blocks per CU).
template <int CHAINS>
__global__ __launch_bounds__(256) void k_valu_fma(float* out, float x, float y, int iters) {
  float a[CHAINS];
#pragma unroll
  for (int k = 0; k < CHAINS; ++k) a[k] = (float)threadIdx.x + k;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(x), "v"(y));
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < CHAINS; ++k) s += a[k];
  if (s == 123.456f) out[0] = s;
}
typedef float float2v __attribute__((ext_vector_type(2)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k_valu_pk_fma(float* out, float x, float y, int iters) {
  float2v a[CHAINS];
  float2v xx = {x, x}, yy = {y, y};
#pragma unroll
  for (int k = 0; k < CHAINS; ++k) a[k] = float2v{(float)threadIdx.x + k, 1.0f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(xx), "v"(yy));
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < CHAINS; ++k) s += a[k].x + a[k].y;
  if (s == 123.456f) out[0] = s;
}
// the instruction mix of a BVH node step, roughly: min / max / mul / sub / cmp / cndmask
template <int CHAINS>
__global__ __launch_bounds__(256) void k_valu_mix(float* out, float x, float y, int iters) {
  float a[CHAINS];
#pragma unroll
  for (int k = 0; k < CHAINS; ++k) a[k] = (float)threadIdx.x + k;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int k = 0; k < CHAINS; ++k) {
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(y));
        asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(x));
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(y));
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < CHAINS; ++k) s += a[k];
  if (s == 123.456f) out[0] = s;
}

struct Timer {
  hipEvent_t e0, e1;
  Timer() { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
  template <typename F>
  double ms(F&& f, int reps) {
    f();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t = 0;
    CK(hipEventElapsedTime(&t, e0, e1));
    return t / reps;
  }
};

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f,\n", p.gcnArchName, cus, ghz);
  Timer T;
  // ------------------------------------------------------------ HBM
  printf(" \"hbm\": [\n");
  bool first = true;
  auto row = [&](const char* name, size_t bytes, double moved, double ms) {
    printf("%s  {\"probe\": \"%s\", \"array_mib\": %zu, \"gbs\": %.1f}", first ? "" : ",\n", name, bytes >> 20, moved / (ms * 1e-3) / 1e9);
    first = false;
    fflush(stdout);
  };
  for (size_t bytes : {(size_t)1 << 28, (size_t)1 << 30, (size_t)4 << 30}) {
    const size_t n = bytes / 16;
    f4 *a, *b;
    float* sink;
    CK(hipMalloc((void**)&a, bytes));
    CK(hipMalloc((void**)&b, bytes));
    CK(hipMalloc((void**)&sink, 4));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 1, bytes));
    const int reps = bytes >= ((size_t)4 << 30) ? 4 : 8;
    for (int per_cu : {4, 8, 16, 32, 64}) {
      char nm[64];
      snprintf(nm, sizeof nm, "copy_stride_%dwg_per_cu", per_cu);
      row(nm, bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL(k_copy_stride<0>, dim3(cus * per_cu), dim3(256), 0, 0, a, b, n); }, reps));
    }
    row("copy_stride_nt_32wg_per_cu", bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL(k_copy_stride<1>, dim3(cus * 32), dim3(256), 0, 0, a, b, n); }, reps));
    row("copy_once_1", bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL((k_copy_once<1, 0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n); }, reps));
    row("copy_once_2", bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL((k_copy_once<2, 0>), dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, 0, a, b, n / 2); }, reps));
    row("copy_once_4", bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL((k_copy_once<4, 0>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, a, b, n / 4); }, reps));
    row("copy_once_8", bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL((k_copy_once<8, 0>), dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, 0, a, b, n / 8); }, reps));
    row("copy_once_4_nt", bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL((k_copy_once<4, 1>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, a, b, n / 4); }, reps));
    for (int per_cu : {8, 32}) {
      const size_t blocks = (size_t)cus * per_cu, per_block = n / blocks;  // n is a power of two, blocks = 2^k * cus: exact when cus = 256
      if (per_block * blocks != n || per_block % 1024) continue;
      char nm[64];
      snprintf(nm, sizeof nm, "copy_chunk4_%dwg_per_cu", per_cu);
      row(nm, bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL((k_copy_chunk<4, 0>), dim3((unsigned)blocks), dim3(256), 0, 0, a, b, per_block); }, reps));
      snprintf(nm, sizeof nm, "copy_chunk4_nt_%dwg_per_cu", per_cu);
      row(nm, bytes, 2.0 * bytes, T.ms([&] { hipLaunchKernelGGL((k_copy_chunk<4, 1>), dim3((unsigned)blocks), dim3(256), 0, 0, a, b, per_block); }, reps));
    }
    row("hipMemcpyDtoD", bytes, 2.0 * bytes, T.ms([&] { CK(hipMemcpyAsync(a, b, bytes, hipMemcpyDeviceToDevice, 0)); }, reps));
    row("read_only_32wg_per_cu", bytes, 1.0 * bytes, T.ms([&] { hipLaunchKernelGGL(k_read, dim3(cus * 32), dim3(256), 0, 0, b, n, sink); }, reps));
    row("fill_32wg_per_cu", bytes, 1.0 * bytes, T.ms([&] { hipLaunchKernelGGL(k_fill, dim3(cus * 32), dim3(256), 0, 0, a, n, 2.0f); }, reps));
    CK(hipFree(a));
    CK(hipFree(b));
    CK(hipFree(sink));
  }
  printf("\n ],\n \"valu\": [\n");
  // ------------------------------------------------------------ VALU issue
  float* out;
  CK(hipMalloc((void**)&out, 4));
  first = true;
  const int iters = 2048;
  auto vrow = [&](const char* name, int waves_per_simd, int chains, double instr_per_wave, double ms) {
    const double waves = (double)cus * 4 * waves_per_simd;
    const double ginstr = waves * instr_per_wave / (ms * 1e-3) / 1e9;
    printf("%s  {\"probe\": \"%s\", \"waves_per_simd\": %d, \"chains\": %d, \"ginstr_s\": %.1f, \"cycles_per_wave_instr_per_simd\": %.3f}", first ? "" : ",\n", name,
           waves_per_simd, chains, ginstr, (double)cus * 4 * ghz / ginstr);
    first = false;
    fflush(stdout);
  };
  for (int w : {1, 2, 4, 8}) {
    // w waves per SIMD = w workgroups of 4 waves per CU
    const dim3 g(cus * w);
    vrow("v_fma_f32", w, 8, 64.0 * iters, T.ms([&] { hipLaunchKernelGGL(k_valu_fma<8>, g, dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); }, 4));
    vrow("v_fma_f32", w, 2, 16.0 * iters, T.ms([&] { hipLaunchKernelGGL(k_valu_fma<2>, g, dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); }, 4));
    vrow("v_fma_f32", w, 1, 8.0 * iters, T.ms([&] { hipLaunchKernelGGL(k_valu_fma<1>, g, dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); }, 4));
    vrow("v_pk_fma_f32", w, 8, 64.0 * iters, T.ms([&] { hipLaunchKernelGGL(k_valu_pk_fma<8>, g, dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); }, 4));
    vrow("sub_mul_min_max", w, 8, 64.0 * iters, T.ms([&] { hipLaunchKernelGGL(k_valu_mix<8>, g, dim3(256), 0, 0, out, 1.0001f, 0.5f, iters); }, 4));
  }
  printf("\n ]}\n");
  return 0;
}
