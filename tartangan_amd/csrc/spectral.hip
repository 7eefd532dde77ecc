// Spectral-norm power iteration (north_star item a15; the reference's only trace is prep4web.py:33-51 which strips
// torch.nn.utils.spectral_norm from a checkpoint, so the semantics pinned here are torch's):
//     n_iter times:  v <- normalize(W^T u, eps),  u <- normalize(W v, eps)      (in place)
//     sigma = u^T W v
// W is rows x cols (Cout x Cin*k*k): ONE workgroup of 16 waves, sized for the matrices of the benchmarked configs
// (<= 256 x 2304: 3 passes over <= 2.4 MB, latency-bound by construction).  Any size is computed correctly, but the
// 1024 x 9216 filters of '128big' stream 37.7 MB x 3 through a single CU (~0.3 ms per layer); a multi-workgroup form
// (row / column partials + a second stage, as in planes.h) is the obvious next step if spectral norm is ever turned on for
// those configs -- the reference's trainers never enable it.
// W^T u runs one column per lane (coalesced along the row), W v one row per wave with a wavefront-shuffle
// reduction; the two norms are block reductions.
#include "common.h"

namespace {

constexpr int SNB = 1024;

__device__ __forceinline__ float block_sum_1024(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < SNB / 64; ++i) r += scratch[i];
  return r;
}

__global__ void __launch_bounds__(SNB) sn_power_iter_kernel(const float* __restrict__ W, float* __restrict__ u, float* __restrict__ v,
                                                            float* __restrict__ sigma, int rows, int cols, int n_iter, float eps) {
  __shared__ float scratch[32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < n_iter; ++it) {
    // v = W^T u
    float nrm = 0.f;
    for (int j = threadIdx.x; j < cols; j += SNB) {
      float acc = 0.f;
      for (int i = 0; i < rows; ++i) acc = fmaf(W[(int64_t)i * cols + j], u[i], acc);
      v[j] = acc;
      nrm = fmaf(acc, acc, nrm);
    }
    nrm = sqrtf(block_sum_1024(nrm, scratch));
    const float inv = 1.f / fmaxf(nrm, eps);
    for (int j = threadIdx.x; j < cols; j += SNB) v[j] *= inv;
    __syncthreads();
    // u = W v
    for (int i = wave; i < rows; i += SNB / 64) {
      float acc = 0.f;
      for (int j = lane; j < cols; j += 64) acc = fmaf(W[(int64_t)i * cols + j], v[j], acc);
      acc = wave_sum(acc);
      if (lane == 0) u[i] = acc;
    }
    __syncthreads();
    float un = 0.f;
    for (int i = threadIdx.x; i < rows; i += SNB) un = fmaf(u[i], u[i], un);
    un = sqrtf(block_sum_1024(un, scratch));
    const float uinv = 1.f / fmaxf(un, eps);
    for (int i = threadIdx.x; i < rows; i += SNB) u[i] *= uinv;
    __syncthreads();
  }
  if (sigma != nullptr) {
    float part = 0.f;
    for (int i = wave; i < rows; i += SNB / 64) {
      float acc = 0.f;
      for (int j = lane; j < cols; j += 64) acc = fmaf(W[(int64_t)i * cols + j], v[j], acc);
      acc = wave_sum(acc);
      if (lane == 0) part = fmaf(u[i], acc, part);
    }
    part = block_sum_1024(part, scratch);
    if (threadIdx.x == 0) *sigma = part;
  }
}

__global__ void recip_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) out[i] = 1.f / x[i];
}

}  // namespace

extern "C" {

int tg_sn_power_iter(const float* W, float* u, float* v, float* sigma, int rows, int cols, int n_iter, float eps, void* stream) {
  TG_CHECK_PTR(W); TG_CHECK_PTR(u); TG_CHECK_PTR(v);
  TG_CHECK_POS(rows); TG_CHECK_POS(cols);
  if (n_iter < 0) return TG_EINVAL;
  sn_power_iter_kernel<<<1, SNB, 0, tg_stream(stream)>>>(W, u, v, sigma, rows, cols, n_iter, eps);
  return tg_launch_status();
}

int tg_recip(const float* x, float* out, int64_t n, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(out);
  if (n <= 0) return TG_EINVAL;
  recip_kernel<<<tg_ew_grid(n, 256), 256, 0, tg_stream(stream)>>>(x, out, n);
  return tg_launch_status();
}

}  // extern "C"
