// NCHW "plane" iteration helpers shared by the BatchNorm and channel-reduction kernels.
//
// A plane is one (b, c) image: HW contiguous floats at (b*C + c)*HW.  Two regimes:
//   big   (HW % 4 == 0 && HW >= 1024): 16-byte accesses, no per-element integer division;
//   small (everything else)          : flat index, one division per element (these tensors
//                                      are at most a few MB, the kernels are launch-bound).
#pragma once
#include "common.h"

namespace planes {

// Groups.  A tensor of G*Bg images may be reduced / normalised per (group, channel) instead of per channel: group g is
// images [g*Bg, (g+1)*Bg) -- the discriminator runs the real and the fake batch of a step as ONE tensor whose BatchNorm
// statistics stay those of each half (trainers/cnn.py:122-123 are two forwards).  A "virtual channel" vc = g*C + c indexes
// everything that exists per group (partials, mean, invstd, coefficients); c indexes the parameters.  G = 1: vc == c.
struct Chan {
  int vc, c;
  int64_t base;      // offset of the group's first image
};
__device__ __forceinline__ Chan chan_of(int vc, int C, int Bg, int HW) {
  const int g = vc / C;
  return Chan{vc, vc - g * C, (int64_t)g * Bg * C * HW};
}
__device__ __forceinline__ Chan chan_of_image(int img, int c, int C, int Bg, int HW) {
  const int g = img / Bg;
  return Chan{g * C + c, c, (int64_t)g * Bg * C * HW};
}

constexpr int BLOCK = 256;
constexpr int TILE = BLOCK * 4;   // floats per block-tile in the big regime

static inline bool big(int HW) { return (HW % 4 == 0) && HW >= TILE; }

// ------------------------------------------------------------------ elementwise over planes
// Body: __device__ void vec4(const Chan&, int64_t off)    -- process 4 floats at off
//       __device__ void one(const Chan&, int64_t off)     -- process 1 float at off
template <class Body>
__global__ void __launch_bounds__(BLOCK) map_big(Body body, int C, int HW, int Bg) {
  const int plane = blockIdx.x;
  const int img = plane / C;
  const Chan ch = chan_of_image(img, plane - img * C, C, Bg, HW);
  const int p = (blockIdx.y * BLOCK + threadIdx.x) * 4;
  if (p < HW) body.vec4(ch, (int64_t)plane * HW + p);
}
// Same, for bodies whose per-channel constants come out of a stage-1 reduction: every block first finishes that
// reduction for ITS channel (`begin`: a wave-parallel sum of the S partials, a few hundred bytes from L2) instead
// of a separate one-wave-per-channel kernel between the two passes.  `lead` marks the one block per channel that
// also publishes the per-channel results (statistics, parameter gradients).
// A block owns a CHANNEL and MAP_TILES tiles of it (over the images and the plane), so the channel prologue is paid once
// per MAP_TILES x 4 KiB instead of once per 4 KiB, and the body is split into ld (all loads of the block's tiles are issued
// before the first use: one load per thread in flight capped these passes near 4.5 TB/s) and st (compute + store).
//   Body: void begin(const Chan&, bool lead); typename V; V ld(int64_t off) const; void st(const Chan&, int64_t off, const V&) const
// grid.x = G*C virtual channels; B = images per group.
constexpr int MAP_TILES = 4;
template <class Body>
__global__ void __launch_bounds__(BLOCK) map_big_begin(Body body, int B, int C, int HW) {
  const Chan ch = chan_of(blockIdx.x, C, B, HW);
  const int c = ch.c;
  body.begin(ch, blockIdx.y == 0);
  const int tiles_per_row = (HW + TILE - 1) / TILE;
  const int T = B * tiles_per_row;
  typename Body::V v[MAP_TILES];
  int64_t off[MAP_TILES];
  bool ok[MAP_TILES];
#pragma unroll
  for (int u = 0; u < MAP_TILES; ++u) {
    const int t = blockIdx.y * MAP_TILES + u;
    const int b = t / tiles_per_row;
    const int p = (t - b * tiles_per_row) * TILE + threadIdx.x * 4;
    ok[u] = t < T && p < HW;
    off[u] = ch.base + ((int64_t)b * C + c) * HW + p;
    if (ok[u]) v[u] = body.ld(off[u]);
  }
#pragma unroll
  for (int u = 0; u < MAP_TILES; ++u)
    if (ok[u]) body.st(ch, off[u], v[u]);
}
template <class Body>
static inline void launch_map_begin(Body body, int B, int C, int HW, hipStream_t st, int G = 1) {   // big regime only; B per group
  const int T = B * ((HW + TILE - 1) / TILE);
  dim3 grid(G * C, (T + MAP_TILES - 1) / MAP_TILES);
  map_big_begin<Body><<<grid, BLOCK, 0, st>>>(body, B, C, HW);
}

template <class Body>
__global__ void __launch_bounds__(BLOCK) map_flat(Body body, int C, int HW, int64_t total, int Bg) {
  for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < total; i += gridDim.x * (int64_t)BLOCK) {
    const int64_t plane = i / HW;
    const int img = (int)(plane / C);
    body.one(chan_of_image(img, (int)(plane - (int64_t)img * C), C, Bg, HW), i);
  }
}
// B = ALL images of the tensor; Bg = images per group (0: one group)
template <class Body>
static inline void launch_map(Body body, int B, int C, int HW, hipStream_t st, bool aligned = true, int Bg = 0) {
  if (Bg <= 0) Bg = B;
  if (big(HW) && aligned) {
    dim3 grid(B * C, (HW + TILE - 1) / TILE);
    map_big<Body><<<grid, BLOCK, 0, st>>>(body, C, HW, Bg);
  } else {
    const int64_t total = (int64_t)B * C * HW;
    map_flat<Body><<<tg_ew_grid(total, BLOCK), BLOCK, 0, st>>>(body, C, HW, total, Bg);
  }
}

// ------------------------------------------------------------------ per-channel reductions, stage 1
// Number of partial blocks per (virtual) channel (must agree between workspace sizing and launch).  B = images per
// group, C = number of virtual channels (G * channels).
static inline int splits(int B, int C, int HW) {
  const int64_t n = (int64_t)B * HW;
  int64_t units = (n + TILE - 1) / TILE;
  int64_t cap = 2048 / (C > 0 ? C : 1);
  if (cap < 1) cap = 1;
  // aim for >= 4 tiles per block so that per-thread partial sums stay short
  int64_t s = (units + 3) / 4;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return (int)s;
}

// Red: static constexpr int K (values per element);
//      __device__ void init(const Chan&);
//      typename V;  __device__ V ld4(int64_t off);          the loads of 4 elements
//      __device__ void acc(const V&, float* a /*K*/);       accumulate them
//      __device__ void acc1(int64_t off, float* a);
// partial layout: [G*C][S][K] doubles.  grid.x = G*C virtual channels; B = images per group.
template <class Red>
__global__ void __launch_bounds__(BLOCK) reduce_stage1(Red red, double* __restrict__ partial, int B, int C, int HW, int S, int is_big) {
  constexpr int K = Red::K;
  __shared__ double scratch[32];
  const Chan ch = chan_of(blockIdx.x, C, B, HW);
  const int c = ch.c, s = blockIdx.y;
  float a[K];
#pragma unroll
  for (int k = 0; k < K; ++k) a[k] = 0.f;
  red.init(ch);
  if (is_big) {
    const int tiles_per_row = (HW + TILE - 1) / TILE;
    const int T = B * tiles_per_row;
    // four tiles per trip: their loads are all issued before the first sum (same summation order as one at a time)
    constexpr int U = 4;
    for (int t0 = s; t0 < T; t0 += U * S) {
      typename Red::V v[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u * S;
        const int b = t / tiles_per_row;
        const int p = (t - b * tiles_per_row) * TILE + threadIdx.x * 4;
        ok[u] = t < T && p < HW;
        if (ok[u]) v[u] = red.ld4(ch.base + ((int64_t)b * C + c) * HW + p);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ok[u]) red.acc(v[u], a);
    }
  } else {
    const int64_t n = (int64_t)B * HW;
    for (int64_t e = (int64_t)s * BLOCK + threadIdx.x; e < n; e += (int64_t)S * BLOCK) {
      const int64_t b = e / HW;
      const int64_t p = e - b * HW;
      red.acc1(ch.base + (b * C + c) * HW + p, a);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double r = block_sum_d((double)a[k], scratch);
    if (threadIdx.x == 0) partial[((int64_t)ch.vc * S + s) * K + k] = r;
  }
}

// B = images per group, G groups
template <class Red>
static inline void launch_reduce(Red red, double* partial, int B, int C, int HW, hipStream_t st, bool aligned = true, int G = 1) {
  const int S = splits(B, G * C, HW);
  dim3 grid(G * C, S);
  reduce_stage1<Red><<<grid, BLOCK, 0, st>>>(red, partial, B, C, HW, S, (big(HW) && aligned) ? 1 : 0);
}

// helper for stage 2 kernels (ONE WAVE PER CHANNEL, blockDim = 64): sum the S partials of value k of
// channel c.  Lanes stride over S and combine by wavefront shuffles; result in every lane.  (A serial
// loop of S dependent loads per channel made these tiny kernels cost ~9 us each.)
__device__ __forceinline__ double gather(const double* __restrict__ partial, int c, int S, int K, int k) {
  double r = 0.0;
  for (int s = threadIdx.x & 63; s < S; s += 64) r += partial[((int64_t)c * S + s) * K + k];
  return wave_sum_d(r);
}

}  // namespace planes
