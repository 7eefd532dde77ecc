// Row / channel reductions and broadcasts, softmax (fwd, bwd, second-order term).
// One wave (64 lanes) per row; row sums are wavefront shuffles, no LDS, no atomics.
#include <type_traits>

#include "planes.h"

namespace {

constexpr int RB = 256;              // 4 waves = 4 rows per workgroup
constexpr int ROWS_PER_BLOCK = RB / 64;

static inline int row_grid(int rows) {
  int64_t g = ((int64_t)rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------ channel sum / bcast
struct RedChan {
  static constexpr int K = 1;
  const float* x;
  __device__ void init(const planes::Chan&) {}
  using V = float4;
  __device__ V ld4(int64_t off) const { return *reinterpret_cast<const float4*>(x + off); }
  __device__ void acc(const V& v, float* a) { a[0] += (v.x + v.y) + (v.z + v.w); }
  __device__ void acc1(int64_t off, float* a) { a[0] += x[off]; }
};

__global__ void chan_stage2(const double* __restrict__ partial, float* __restrict__ out, int C, int S, int accumulate) {
  const int c = blockIdx.x;                       // one wave per channel
  const double r = planes::gather(partial, c, S, 1, 0);
  if (threadIdx.x == 0) out[c] = (float)r + (accumulate ? out[c] : 0.f);
}

struct BcastBody {
  const float* v; float* out;
  __device__ void vec4(const planes::Chan& ch, int64_t off) const {
    const float s = v[ch.c];
    *reinterpret_cast<float4*>(out + off) = make_float4(s, s, s, s);
  }
  __device__ void one(const planes::Chan& ch, int64_t off) const { out[off] = v[ch.c]; }
};

// ------------------------------------------------------------------ row sum / bcast / repeat
__global__ void __launch_bounds__(RB) row_sum_kernel(const float* __restrict__ x, float* __restrict__ out, float alpha,
                                                     int rows, int cols) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); r < rows; r += gridDim.x * ROWS_PER_BLOCK) {
    const float* row = x + (int64_t)r * cols;
    float acc = 0.f;
    for (int c = lane; c < cols; c += 64) acc += row[c];
    acc = wave_sum(acc);
    if (lane == 0) out[r] = alpha * acc;
  }
}

__global__ void __launch_bounds__(RB) row_bcast_kernel(const float* __restrict__ v, float* __restrict__ out, float alpha,
                                                       int64_t total, int cols) {
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < total; i += gridDim.x * (int64_t)RB)
    out[i] = alpha * v[i / cols];
}

// out[(g*reps + q)*rc + w] = alpha * x[g*rc + w]: x is G groups of `rows` rows, each group repeated on its own (G = 1: x.repeat(reps, 1))
__global__ void __launch_bounds__(RB) repeat_rows_kernel(const float* __restrict__ x, float* __restrict__ out, float alpha,
                                                         int64_t rc, int reps, int groups) {
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < rc * groups; i += gridDim.x * (int64_t)RB) {
    const int64_t g = i / rc, w = i - g * rc;
    const float v = alpha * x[i];
    for (int q = 0; q < reps; ++q) out[(g * reps + q) * rc + w] = v;
  }
}

__global__ void __launch_bounds__(RB) sum_reps_kernel(const float* __restrict__ x, float* __restrict__ out, float alpha,
                                                      int64_t rc, int reps, int groups) {
  for (int64_t i = blockIdx.x * (int64_t)RB + threadIdx.x; i < rc * groups; i += gridDim.x * (int64_t)RB) {
    const int64_t g = i / rc, w = i - g * rc;
    float acc = 0.f;
    for (int q = 0; q < reps; ++q) acc += x[(g * reps + q) * rc + w];
    out[i] = alpha * acc;
  }
}

// ------------------------------------------------------------------ softmax over the last dim
__global__ void __launch_bounds__(RB) softmax_fwd_kernel(const float* __restrict__ s, float* __restrict__ y, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); r < rows; r += gridDim.x * ROWS_PER_BLOCK) {
    const float* in = s + (int64_t)r * cols;
    float* o = y + (int64_t)r * cols;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, in[c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += expf(in[c] - m);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int c = lane; c < cols; c += 64) o[c] = expf(in[c] - m) * inv;
  }
}

__global__ void __launch_bounds__(RB) softmax_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                         float* __restrict__ gs, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); r < rows; r += gridDim.x * ROWS_PER_BLOCK) {
    const int64_t base = (int64_t)r * cols;
    float d = 0.f;
    for (int c = lane; c < cols; c += 64) d += gy[base + c] * y[base + c];
    d = wave_sum(d);
    for (int c = lane; c < cols; c += 64) gs[base + c] = y[base + c] * (gy[base + c] - d);
  }
}

__global__ void __launch_bounds__(RB) softmax_dbwd_kernel(const float* __restrict__ v, const float* __restrict__ gy,
                                                          const float* __restrict__ y, float* __restrict__ out, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); r < rows; r += gridDim.x * ROWS_PER_BLOCK) {
    const int64_t base = (int64_t)r * cols;
    float d = 0.f, e = 0.f;
    for (int c = lane; c < cols; c += 64) {
      const float yy = y[base + c];
      d += gy[base + c] * yy;
      e += v[base + c] * yy;
    }
    d = wave_sum(d);
    e = wave_sum(e);
    for (int c = lane; c < cols; c += 64) {
      const float vv = v[base + c], g = gy[base + c];
      out[base + c] = vv * g - vv * d - g * e;
    }
  }
}

// Second-order attention, the row-wise middle of it (see tg_attn_dbwd_rows): per row of the (N x M) maps
//   P = exp(S - lse),  delta = sum P gP,  eps = sum P U,  zeta = sum P dP = sum P V + sum P U gP - 2 delta eps,
//   gS = P (gP - delta),  dgP = P (U - eps),  dP = V + U (gP - delta) - gP eps,  dS = P (dP - zeta);
// every output overwrites the input it replaces (same element, same thread).
template <bool VEC>
__global__ void __launch_bounds__(RB) attn_dbwd_rows_kernel(float* __restrict__ s, const float* __restrict__ lse, float* __restrict__ gp,
                                                            float* __restrict__ u, float* __restrict__ v, int rows, int cols) {
  using T = typename std::conditional<VEC, float4, float>::type;
  constexpr int W = VEC ? 4 : 1;
  const int lane = threadIdx.x & 63;
  const int nq = cols / W;
  for (int r = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); r < rows; r += gridDim.x * ROWS_PER_BLOCK) {
    const int64_t base = (int64_t)r * cols;
    const float l = lse[r];
    float d = 0.f, e = 0.f, pv = 0.f, pug = 0.f;
    for (int q = lane; q < nq; q += 64) {
      const T sv = *reinterpret_cast<const T*>(s + base + q * W), gv = *reinterpret_cast<const T*>(gp + base + q * W);
      const T uv = *reinterpret_cast<const T*>(u + base + q * W), vv = *reinterpret_cast<const T*>(v + base + q * W);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const float p = expf(reinterpret_cast<const float*>(&sv)[k] - l);
        const float g = reinterpret_cast<const float*>(&gv)[k], uu = reinterpret_cast<const float*>(&uv)[k];
        d += p * g; e += p * uu; pv += p * reinterpret_cast<const float*>(&vv)[k]; pug += p * uu * g;
      }
    }
    d = wave_sum(d); e = wave_sum(e); pv = wave_sum(pv); pug = wave_sum(pug);
    const float z = pv + pug - 2.f * d * e;
    for (int q = lane; q < nq; q += 64) {
      T sv = *reinterpret_cast<const T*>(s + base + q * W), gv = *reinterpret_cast<const T*>(gp + base + q * W);
      T uv = *reinterpret_cast<const T*>(u + base + q * W), vv = *reinterpret_cast<const T*>(v + base + q * W);
#pragma unroll
      for (int k = 0; k < W; ++k) {
        float& sp = reinterpret_cast<float*>(&sv)[k];
        float& g = reinterpret_cast<float*>(&gv)[k];
        float& uu = reinterpret_cast<float*>(&uv)[k];
        float& ww = reinterpret_cast<float*>(&vv)[k];
        const float p = expf(sp - l);
        const float dp = ww + uu * (g - d) - g * e;
        sp = p;
        ww = p * (dp - z);
        uu = p * (uu - e);
        g = p * (g - d);
      }
      *reinterpret_cast<T*>(s + base + q * W) = sv;
      *reinterpret_cast<T*>(gp + base + q * W) = gv;
      *reinterpret_cast<T*>(u + base + q * W) = uv;
      *reinterpret_cast<T*>(v + base + q * W) = vv;
    }
  }
}

}  // namespace

extern "C" {

int tg_channel_sum(const float* x, float* out, float* workspace, int B, int C, int HW, int accumulate, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(out); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  hipStream_t st = tg_stream(stream);
  double* partial = reinterpret_cast<double*>(workspace);
  planes::launch_reduce(RedChan{x}, partial, B, C, HW, st, tg_aligned16(x));
  chan_stage2<<<C, 64, 0, st>>>(partial, out, C, planes::splits(B, C, HW), accumulate);
  return tg_launch_status();
}

int tg_channel_bcast(const float* v, float* out, int B, int C, int HW, void* stream) {
  TG_CHECK_PTR(v); TG_CHECK_PTR(out);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  planes::launch_map(BcastBody{v, out}, B, C, HW, tg_stream(stream), tg_aligned16(out));
  return tg_launch_status();
}

int tg_row_sum(const float* x, float* out, float alpha, int rows, int cols, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(out); TG_CHECK_POS(rows); TG_CHECK_POS(cols);
  row_sum_kernel<<<row_grid(rows), RB, 0, tg_stream(stream)>>>(x, out, alpha, rows, cols);
  return tg_launch_status();
}

int tg_row_bcast(const float* v, float* out, float alpha, int rows, int cols, void* stream) {
  TG_CHECK_PTR(v); TG_CHECK_PTR(out); TG_CHECK_POS(rows); TG_CHECK_POS(cols);
  const int64_t total = (int64_t)rows * cols;
  row_bcast_kernel<<<tg_ew_grid(total, RB), RB, 0, tg_stream(stream)>>>(v, out, alpha, total, cols);
  return tg_launch_status();
}

int tg_repeat_rows_groups(const float* x, float* out, float alpha, int rows, int cols, int reps, int groups, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(out); TG_CHECK_POS(rows); TG_CHECK_POS(cols); TG_CHECK_POS(reps); TG_CHECK_POS(groups);
  const int64_t rc = (int64_t)rows * cols;
  repeat_rows_kernel<<<tg_ew_grid(rc * groups, RB), RB, 0, tg_stream(stream)>>>(x, out, alpha, rc, reps, groups);
  return tg_launch_status();
}
int tg_repeat_rows(const float* x, float* out, float alpha, int rows, int cols, int reps, void* stream) {
  return tg_repeat_rows_groups(x, out, alpha, rows, cols, reps, 1, stream);
}

int tg_sum_reps_groups(const float* x, float* out, float alpha, int rows, int cols, int reps, int groups, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(out); TG_CHECK_POS(rows); TG_CHECK_POS(cols); TG_CHECK_POS(reps); TG_CHECK_POS(groups);
  const int64_t rc = (int64_t)rows * cols;
  sum_reps_kernel<<<tg_ew_grid(rc * groups, RB), RB, 0, tg_stream(stream)>>>(x, out, alpha, rc, reps, groups);
  return tg_launch_status();
}
int tg_sum_reps(const float* x, float* out, float alpha, int rows, int cols, int reps, void* stream) {
  return tg_sum_reps_groups(x, out, alpha, rows, cols, reps, 1, stream);
}

int tg_softmax_fwd(const float* s, float* y, int rows, int cols, void* stream) {
  TG_CHECK_PTR(s); TG_CHECK_PTR(y); TG_CHECK_POS(rows); TG_CHECK_POS(cols);
  softmax_fwd_kernel<<<row_grid(rows), RB, 0, tg_stream(stream)>>>(s, y, rows, cols);
  return tg_launch_status();
}

int tg_softmax_bwd(const float* gy, const float* y, float* gs, int rows, int cols, void* stream) {
  TG_CHECK_PTR(gy); TG_CHECK_PTR(y); TG_CHECK_PTR(gs); TG_CHECK_POS(rows); TG_CHECK_POS(cols);
  softmax_bwd_kernel<<<row_grid(rows), RB, 0, tg_stream(stream)>>>(gy, y, gs, rows, cols);
  return tg_launch_status();
}

int tg_softmax_dbwd(const float* v, const float* gy, const float* y, float* out, int rows, int cols, void* stream) {
  TG_CHECK_PTR(v); TG_CHECK_PTR(gy); TG_CHECK_PTR(y); TG_CHECK_PTR(out); TG_CHECK_POS(rows); TG_CHECK_POS(cols);
  softmax_dbwd_kernel<<<row_grid(rows), RB, 0, tg_stream(stream)>>>(v, gy, y, out, rows, cols);
  return tg_launch_status();
}

int tg_attn_dbwd_rows(float* s, const float* lse, float* gp, float* u, float* v, int rows, int cols, void* stream) {
  TG_CHECK_PTR(s); TG_CHECK_PTR(lse); TG_CHECK_PTR(gp); TG_CHECK_PTR(u); TG_CHECK_PTR(v); TG_CHECK_POS(rows); TG_CHECK_POS(cols);
  const bool vec = (cols % 4 == 0) && tg_aligned16(s) && tg_aligned16(gp) && tg_aligned16(u) && tg_aligned16(v);
  if (vec) attn_dbwd_rows_kernel<true><<<row_grid(rows), RB, 0, tg_stream(stream)>>>(s, lse, gp, u, v, rows, cols);
  else attn_dbwd_rows_kernel<false><<<row_grid(rows), RB, 0, tg_stream(stream)>>>(s, lse, gp, u, v, rows, cols);
  return tg_launch_status();
}

}  // extern "C"
