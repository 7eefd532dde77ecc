// Shared device/host helpers for libtartangan_amd (gfx950 only: wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tartangan_amd.h"

#define TG_WAVE 64

#define TG_CHECK_PTR(p) \
  do {                  \
    if ((p) == nullptr) return TG_EINVAL; \
  } while (0)
#define TG_CHECK_POS(v) \
  do {                  \
    if ((v) <= 0) return TG_EINVAL; \
  } while (0)

static inline int tg_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? TG_OK : (int)e;
}

static inline hipStream_t tg_stream(void* s) { return (hipStream_t)s; }

static inline int tg_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// grid size for a bandwidth-bound grid-stride kernel: enough blocks to fill 256 CUs x 8
static inline int tg_ew_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

static inline bool tg_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Sum over the block; result valid in every thread.  `scratch` >= 32 floats of LDS.
// Deterministic (fixed tree) for a fixed block size.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}
__device__ __forceinline__ double block_sum_d(double v, double* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
  return r;
}
