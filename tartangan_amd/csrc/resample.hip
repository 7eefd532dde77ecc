// Resampling kernels (all HBM-bound, gather form, no atomics):
//   up2x / pool2            nearest x2 upsample <-> 2x2 sum-pool (transposes of each other)
//   bilinear_half fwd/bwd   F.interpolate(scale .5, bilinear, align_corners=True) and its transpose
//   maxpool2 fwd/bwd/gather 2x2 max-pool with argmax byte, scatter and gather by argmax
#include "common.h"

namespace {

constexpr int RS_BLOCK = 256;

// ------------------------------------------------------------------ up2x / pool2
// vector path: one thread per input PAIR (w, w+1) -> two float4 stores (rows 2h, 2h+1)
__global__ void __launch_bounds__(RS_BLOCK) up2x_vec(const float* __restrict__ x, float* __restrict__ y, float alpha,
                                                     int64_t npairs, int H, int W2 /* = W/2 */) {
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < npairs; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int64_t row = i / W2;            // bc*H + h
    const int wp = (int)(i - row * W2);
    const float2 v = *reinterpret_cast<const float2*>(x + row * (2 * W2) + 2 * wp);
    const float a = alpha * v.x, b = alpha * v.y;
    const float4 o = make_float4(a, a, b, b);
    // output row index = 2*row (since (bc*2H + 2h) = 2*(bc*H + h)), width 4*W2
    float* dst = y + (2 * row) * (int64_t)(4 * W2) + 4 * wp;
    *reinterpret_cast<float4*>(dst) = o;
    *reinterpret_cast<float4*>(dst + 4 * W2) = o;
  }
}
__global__ void __launch_bounds__(RS_BLOCK) up2x_scalar(const float* __restrict__ x, float* __restrict__ y, float alpha,
                                                        int64_t nout, int H, int W) {
  const int OW = 2 * W, OH = 2 * H;
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < nout; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t bc = t / OH;
    y[i] = alpha * x[(bc * H + (oh >> 1)) * W + (ow >> 1)];
  }
}

// vector path: one thread per output PAIR: float4 from row 2h and row 2h+1
__global__ void __launch_bounds__(RS_BLOCK) pool2_vec(const float* __restrict__ x, float* __restrict__ y, float alpha,
                                                      int64_t npairs, int OW2 /* = W/4 */, const float* __restrict__ residual) {
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < npairs; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int64_t orow = i / OW2;          // bc*OH + oh
    const int op = (int)(i - orow * OW2);
    const float* src = x + (2 * orow) * (int64_t)(4 * OW2) + 4 * op;
    const float4 r0 = *reinterpret_cast<const float4*>(src);
    const float4 r1 = *reinterpret_cast<const float4*>(src + 4 * OW2);
    float2 o;
    o.x = alpha * ((r0.x + r0.y) + (r1.x + r1.y));
    o.y = alpha * ((r0.z + r0.w) + (r1.z + r1.w));
    if (residual) {
      const float2 q = *reinterpret_cast<const float2*>(residual + orow * (int64_t)(2 * OW2) + 2 * op);
      o.x = q.x + o.x;
      o.y = q.y + o.y;
    }
    *reinterpret_cast<float2*>(y + orow * (int64_t)(2 * OW2) + 2 * op) = o;
  }
}
__global__ void __launch_bounds__(RS_BLOCK) pool2_scalar(const float* __restrict__ x, float* __restrict__ y, float alpha,
                                                         int64_t nout, int H, int W, const float* __restrict__ residual) {
  const int OW = W / 2, OH = H / 2;
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < nout; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t bc = t / OH;
    const float* s = x + (bc * H + 2 * oh) * W + 2 * ow;
    const float o = alpha * ((s[0] + s[1]) + (s[W] + s[W + 1]));
    y[i] = residual ? residual[i] + o : o;
  }
}

// ------------------------------------------------------------------ bilinear half (align_corners = True)
// ATen's source-index computation in fp32 (UpSample.h area_pixel_compute_source_index, align_corners)
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp src_index(int o, float scale, int in_size) {
  // ATen rounds scale*o to fp32 BEFORE taking the fractional part; an fma(scale, o, -i0) here changes
  // lambda by up to ulp(r) (7.6e-6 at 128 px), which was measurable end to end (gp off by 7e-5).
  // hipcc contracts through __fmul_rn and `#pragma clang fp contract(off)` alike (checked in the ISA:
  // v_pk_fma_f32 scale, o, -i0); the empty asm makes the rounded product opaque, which does stop it.
  float r = scale * (float)o;
  asm volatile("" : "+v"(r));
  Lerp t;
  t.i0 = (int)r;
  if (t.i0 > in_size - 1) t.i0 = in_size - 1;
  t.i1 = t.i0 + ((t.i0 < in_size - 1) ? 1 : 0);
  t.l1 = __fsub_rn(r, (float)t.i0);
  t.l0 = __fsub_rn(1.f, t.l1);
  return t;
}

__global__ void __launch_bounds__(RS_BLOCK) bilinear_half_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                     int64_t nout, int H, int W, int OH, int OW, float sh, float sw) {
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < nout; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t bc = t / OH;
    const Lerp a = src_index(oh, sh, H), b = src_index(ow, sw, W);
    const float* p = x + bc * (int64_t)H * W;
    const float top = b.l0 * p[a.i0 * W + b.i0] + b.l1 * p[a.i0 * W + b.i1];
    const float bot = b.l0 * p[a.i1 * W + b.i0] + b.l1 * p[a.i1 * W + b.i1];
    y[i] = a.l0 * top + a.l1 * bot;
  }
}

// weight with which output index o reads input index `in` along one axis
__device__ __forceinline__ float axis_weight(int o, int in, float scale, int in_size) {
  const Lerp t = src_index(o, scale, in_size);
  float w = 0.f;
  if (t.i0 == in) w += t.l0;
  if (t.i1 == in) w += t.l1;
  return w;
}

__global__ void __launch_bounds__(RS_BLOCK) bilinear_half_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ residual,
                                                                     float* __restrict__ gx, int64_t nin, int H, int W, int OH, int OW, float sh, float sw) {
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < nin; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int w = (int)(i % W);
    const int64_t t = i / W;
    const int h = (int)(t % H);
    const int64_t bc = t / H;
    // outputs o with floor(scale*o) in {in-1, in}: at most 4 candidates per axis (scale >= 2), tested exactly;
    // the per-axis weights are computed once (8 index evaluations instead of up to 25)
    int oh_lo = sh > 0.f ? (int)((float)(h - 1) / sh) - 1 : 0;
    int ow_lo = sw > 0.f ? (int)((float)(w - 1) / sw) - 1 : 0;
    oh_lo = oh_lo < 0 ? 0 : oh_lo; ow_lo = ow_lo < 0 ? 0 : ow_lo;
    float wh[4], ww[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      wh[u] = (oh_lo + u < OH) ? axis_weight(oh_lo + u, h, sh, H) : 0.f;
      ww[u] = (ow_lo + u < OW) ? axis_weight(ow_lo + u, w, sw, W) : 0.f;
    }
    const float* g = gy + bc * (int64_t)OH * OW;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (wh[u] == 0.f) continue;
      float racc = 0.f;
#pragma unroll
      for (int t2 = 0; t2 < 4; ++t2)
        if (ww[t2] != 0.f) racc += ww[t2] * g[(oh_lo + u) * OW + ow_lo + t2];
      acc += wh[u] * racc;
    }
    gx[i] = residual ? residual[i] + acc : acc;
  }
}

// Plane-per-workgroup variant: the per-axis candidate windows / weights are computed ONCE per workgroup into LDS
// (H + W entries instead of 8 index evaluations per element) and the gy plane is read from LDS.
constexpr int BL_MAX_OUT = 8192;     // OH*OW floats of gy staged in LDS (32 KB)
struct AxisEntry { int lo; float w[4]; };
__global__ void __launch_bounds__(RS_BLOCK) bilinear_half_bwd_plane_kernel(const float* __restrict__ gy, const float* __restrict__ residual,
                                                                           float* __restrict__ gx, int BC,
                                                                           int H, int W, int OH, int OW, float sh, float sw) {
  __shared__ float g[BL_MAX_OUT];
  __shared__ AxisEntry rows[128], cols[128];
  for (int i = threadIdx.x; i < H + W; i += RS_BLOCK) {
    const bool is_row = i < H;
    const int in = is_row ? i : i - H;
    const float sc = is_row ? sh : sw;
    const int in_size = is_row ? H : W, out_size = is_row ? OH : OW;
    int lo = sc > 0.f ? (int)((float)(in - 1) / sc) - 1 : 0;
    lo = lo < 0 ? 0 : lo;
    AxisEntry e;
    e.lo = lo;
#pragma unroll
    for (int u = 0; u < 4; ++u) e.w[u] = (lo + u < out_size) ? axis_weight(lo + u, in, sc, in_size) : 0.f;
    if (is_row) rows[in] = e; else cols[in] = e;
  }
  const int nout = OH * OW, nin = H * W;
  for (int plane = blockIdx.x; plane < BC; plane += gridDim.x) {
    __syncthreads();
    const float* gp = gy + (int64_t)plane * nout;
    for (int i = threadIdx.x; i < nout; i += RS_BLOCK) g[i] = gp[i];
    __syncthreads();
    float* xp = gx + (int64_t)plane * nin;
    const float* rp = residual ? residual + (int64_t)plane * nin : nullptr;
    for (int i = threadIdx.x; i < nin; i += RS_BLOCK) {
      const int h = i / W, w = i - h * W;
      const AxisEntry r = rows[h], c = cols[w];
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r.w[u] == 0.f) continue;
        const float* grow = g + (r.lo + u) * OW + c.lo;
        float racc = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (c.w[t] != 0.f) racc += c.w[t] * grow[t];
        acc += r.w[u] * racc;
      }
      xp[i] = rp ? rp[i] + acc : acc;
    }
  }
}

// One-hit variant.  With scale = (in - 1) / (out - 1) >= 2 consecutive outputs read disjoint input pairs
// (floor(s (o + 1)) >= floor(s o) + 2), so every input element receives from AT MOST ONE output per axis: the transpose is
// a plain gather gx[i][j] = wr[i] * wc[j] * gy[ro[i]][co[j]] (+ residual).  The (index, weight) tables are built once per
// workgroup from the same fp32 source-index arithmetic as the forward; rows are written as float4.
constexpr int BL_MAX_AXIS = 512;
struct OneHit { int o; float w; };
__global__ void __launch_bounds__(RS_BLOCK) bilinear_half_bwd_onehit_kernel(const float* __restrict__ gy, const float* __restrict__ residual,
                                                                            float* __restrict__ gx, int BC, int H, int W, int OH, int OW,
                                                                            float sh, float sw) {
  __shared__ OneHit rows[BL_MAX_AXIS], cols[BL_MAX_AXIS];
  for (int i = threadIdx.x; i < H + W; i += RS_BLOCK) {
    const bool is_row = i < H;
    const int in = is_row ? i : i - H;
    const float sc = is_row ? sh : sw;
    const int in_size = is_row ? H : W, out_size = is_row ? OH : OW;
    int lo = (int)((float)(in - 1) / sc) - 1;
    lo = lo < 0 ? 0 : lo;
    OneHit e{0, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float w = (lo + u < out_size) ? axis_weight(lo + u, in, sc, in_size) : 0.f;
      if (w != 0.f) { e.o = lo + u; e.w = w; }
    }
    if (is_row) rows[in] = e; else cols[in] = e;
  }
  __syncthreads();
  const int nin = H * W, nout = OH * OW, q4 = W >> 2;            // W % 4 == 0 (host-checked)
  for (int plane = blockIdx.x; plane < BC; plane += gridDim.x) {
    const float* gp = gy + (int64_t)plane * nout;
    float* xp = gx + (int64_t)plane * nin;
    const float* rp = residual ? residual + (int64_t)plane * nin : nullptr;
    for (int e = threadIdx.x; e < H * q4; e += RS_BLOCK) {
      const int h = e / q4, w0 = (e - h * q4) * 4;
      const OneHit r = rows[h];
      const float* grow = gp + r.o * OW;
      float4 o;
      { const OneHit c = cols[w0];     o.x = r.w * c.w * grow[c.o]; }
      { const OneHit c = cols[w0 + 1]; o.y = r.w * c.w * grow[c.o]; }
      { const OneHit c = cols[w0 + 2]; o.z = r.w * c.w * grow[c.o]; }
      { const OneHit c = cols[w0 + 3]; o.w = r.w * c.w * grow[c.o]; }
      if (rp) {
        const float4 rr = *reinterpret_cast<const float4*>(rp + (int64_t)h * W + w0);
        o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
      }
      *reinterpret_cast<float4*>(xp + (int64_t)h * W + w0) = o;
    }
  }
}

// ------------------------------------------------------------------ 2x2 max pool
__global__ void __launch_bounds__(RS_BLOCK) maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                uint8_t* __restrict__ idx, int64_t nout, int H, int W) {
  const int OW = W / 2, OH = H / 2;
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < nout; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t bc = t / OH;
    const float* s = x + (bc * H + 2 * oh) * W + 2 * ow;
    const float v[4] = {s[0], s[1], s[W], s[W + 1]};
    float m = v[0];
    int k = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (v[j] > m || v[j] != v[j]) { m = v[j]; k = j; }     // first maximum wins; NaN propagates like ATen
    y[i] = m;
    idx[i] = (uint8_t)k;
  }
}

__global__ void __launch_bounds__(RS_BLOCK) maxpool2_bwd_kernel(const float* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                                float* __restrict__ gx, int64_t nout, int H, int W) {
  const int OW = W / 2, OH = H / 2;
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < nout; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t bc = t / OH;
    float* d = gx + (bc * H + 2 * oh) * W + 2 * ow;
    const float g = gy[i];
    const int k = idx[i];
    *reinterpret_cast<float2*>(d) = make_float2(k == 0 ? g : 0.f, k == 1 ? g : 0.f);
    *reinterpret_cast<float2*>(d + W) = make_float2(k == 2 ? g : 0.f, k == 3 ? g : 0.f);
  }
}

__global__ void __launch_bounds__(RS_BLOCK) maxpool2_gather_kernel(const float* __restrict__ x, const uint8_t* __restrict__ idx,
                                                                   float* __restrict__ y, int64_t nout, int H, int W) {
  const int OW = W / 2, OH = H / 2;
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < nout; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t bc = t / OH;
    const int k = idx[i];
    y[i] = x[(bc * H + 2 * oh + (k >> 1)) * W + 2 * ow + (k & 1)];
  }
}

}  // namespace

// ---- Inception input (8f-2): the reference hands its samples to the Inception network through
// accumulate_inception_activations' transform (inception_utils.py:254-258: (s + 1) / 2, then (s - VGG_MEAN) / VGG_STD) and
// WrapInception.forward (:44-50: the SAME normalisation once more, then F.interpolate(size=(299, 299), mode='bilinear',
// align_corners=True)).  One kernel: `stages` normalisations (per source pixel, in the reference's operation order:
// add, divide by 2, subtract, divide), then ATen's align_corners bilinear weights.  Streaming, HBM-bound
// (4 B out per sample; the four source pixels of neighbouring outputs come from L2).
__device__ __forceinline__ float inception_norm(float v, float m, float sd, int stages) {
  for (int k = 0; k < stages; ++k) {
    v = (v + 1.f) / 2.0f;
    v = (v - m) / sd;
  }
  return v;
}
__global__ void __launch_bounds__(RS_BLOCK) inception_preprocess_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                        const float* __restrict__ stdv, float* __restrict__ out, int C, int H,
                                                                        int W, int OH, int OW, int stages, float rh, float rw, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)RS_BLOCK + threadIdx.x; i < total; i += gridDim.x * (int64_t)RS_BLOCK) {
    const int ow = (int)(i % OW);
    const int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t plane = t / OH;
    const int c = (int)(plane % C);
    const float m = mean[c], sd = stdv[c];
    const float* src = x + plane * (int64_t)H * W;
    if (H == OH && W == OW) {
      out[i] = inception_norm(src[(int64_t)oh * W + ow], m, sd, stages);
      continue;
    }
    const float h1r = rh * oh, w1r = rw * ow;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < H - 1 ? 1 : 0, w1p = w1 < W - 1 ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
    const float* p = src + (int64_t)h1 * W + w1;
    const float v00 = inception_norm(p[0], m, sd, stages), v01 = inception_norm(p[w1p], m, sd, stages);
    const float v10 = inception_norm(p[(int64_t)h1p * W], m, sd, stages), v11 = inception_norm(p[(int64_t)h1p * W + w1p], m, sd, stages);
    out[i] = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
  }
}

extern "C" {

int tg_up2x(const float* x, float* y, float alpha, int BC, int H, int W, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(y); TG_CHECK_POS(BC); TG_CHECK_POS(H); TG_CHECK_POS(W);
  hipStream_t st = tg_stream(stream);
  if (W % 2 == 0 && tg_aligned16(y) && ((uintptr_t)x & 7) == 0) {
    const int64_t npairs = (int64_t)BC * H * (W / 2);
    up2x_vec<<<tg_ew_grid(npairs, RS_BLOCK), RS_BLOCK, 0, st>>>(x, y, alpha, npairs, H, W / 2);
  } else {
    const int64_t nout = (int64_t)BC * H * W * 4;
    up2x_scalar<<<tg_ew_grid(nout, RS_BLOCK), RS_BLOCK, 0, st>>>(x, y, alpha, nout, H, W);
  }
  return tg_launch_status();
}

int tg_pool2(const float* x, const float* residual, float* y, float alpha, int BC, int H, int W, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(y); TG_CHECK_POS(BC); TG_CHECK_POS(H); TG_CHECK_POS(W);
  if ((H & 1) || (W & 1)) return TG_EUNSUPPORTED;
  hipStream_t st = tg_stream(stream);
  if (W % 4 == 0 && tg_aligned16(x) && ((uintptr_t)y & 7) == 0 && ((uintptr_t)residual & 7) == 0) {
    const int64_t npairs = (int64_t)BC * (H / 2) * (W / 4);
    pool2_vec<<<tg_ew_grid(npairs, RS_BLOCK), RS_BLOCK, 0, st>>>(x, y, alpha, npairs, W / 4, residual);
  } else {
    const int64_t nout = (int64_t)BC * (H / 2) * (W / 2);
    pool2_scalar<<<tg_ew_grid(nout, RS_BLOCK), RS_BLOCK, 0, st>>>(x, y, alpha, nout, H, W, residual);
  }
  return tg_launch_status();
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

int tg_bilinear_half_fwd(const float* x, float* y, int BC, int H, int W, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(y); TG_CHECK_POS(BC);
  if (H < 2 || W < 2) return TG_EUNSUPPORTED;
  const int OH = H / 2, OW = W / 2;
  const int64_t nout = (int64_t)BC * OH * OW;
  bilinear_half_fwd_kernel<<<tg_ew_grid(nout, RS_BLOCK), RS_BLOCK, 0, tg_stream(stream)>>>(x, y, nout, H, W, OH, OW,
                                                                                          ac_scale(H, OH), ac_scale(W, OW));
  return tg_launch_status();
}

int tg_bilinear_half_bwd(const float* gy, const float* residual, float* gx, int BC, int H, int W, void* stream) {
  TG_CHECK_PTR(gy); TG_CHECK_PTR(gx); TG_CHECK_POS(BC);
  if (H < 2 || W < 2) return TG_EUNSUPPORTED;
  const int OH = H / 2, OW = W / 2;
  const int64_t nin = (int64_t)BC * H * W;
  if (OH >= 2 && OW >= 2 && W % 4 == 0 && H <= BL_MAX_AXIS && W <= BL_MAX_AXIS && H * W >= 256 && tg_aligned16(gx) &&
      (!residual || tg_aligned16(residual))) {
    // scale = (in - 1) / (out - 1) > 2 for out = in / 2: one contributing output per axis
    const int grid = BC < 4096 ? BC : 4096;
    bilinear_half_bwd_onehit_kernel<<<grid, RS_BLOCK, 0, tg_stream(stream)>>>(gy, residual, gx, BC, H, W, OH, OW, ac_scale(H, OH),
                                                                              ac_scale(W, OW));
    return tg_launch_status();
  }
  if (H <= 128 && W <= 128 && OH * OW <= BL_MAX_OUT && H * W >= 256) {
    const int grid = BC < 2048 ? BC : 2048;
    bilinear_half_bwd_plane_kernel<<<grid, RS_BLOCK, 0, tg_stream(stream)>>>(gy, residual, gx, BC, H, W, OH, OW, ac_scale(H, OH),
                                                                             ac_scale(W, OW));
    return tg_launch_status();
  }
  bilinear_half_bwd_kernel<<<tg_ew_grid(nin, RS_BLOCK), RS_BLOCK, 0, tg_stream(stream)>>>(gy, residual, gx, nin, H, W, OH, OW,
                                                                                         ac_scale(H, OH), ac_scale(W, OW));
  return tg_launch_status();
}

int tg_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int BC, int H, int W, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(y); TG_CHECK_PTR(idx); TG_CHECK_POS(BC);
  if (H < 2 || W < 2 || (H & 1) || (W & 1)) return TG_EUNSUPPORTED;
  const int64_t nout = (int64_t)BC * (H / 2) * (W / 2);
  maxpool2_fwd_kernel<<<tg_ew_grid(nout, RS_BLOCK), RS_BLOCK, 0, tg_stream(stream)>>>(x, y, idx, nout, H, W);
  return tg_launch_status();
}

int tg_maxpool2_bwd(const float* gy, const uint8_t* idx, float* gx, int BC, int H, int W, void* stream) {
  TG_CHECK_PTR(gy); TG_CHECK_PTR(idx); TG_CHECK_PTR(gx); TG_CHECK_POS(BC);
  if (H < 2 || W < 2 || (H & 1) || (W & 1) || ((uintptr_t)gx & 7)) return TG_EUNSUPPORTED;
  const int64_t nout = (int64_t)BC * (H / 2) * (W / 2);
  maxpool2_bwd_kernel<<<tg_ew_grid(nout, RS_BLOCK), RS_BLOCK, 0, tg_stream(stream)>>>(gy, idx, gx, nout, H, W);
  return tg_launch_status();
}

int tg_maxpool2_gather(const float* x, const uint8_t* idx, float* y, int BC, int H, int W, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(idx); TG_CHECK_PTR(y); TG_CHECK_POS(BC);
  if (H < 2 || W < 2 || (H & 1) || (W & 1)) return TG_EUNSUPPORTED;
  const int64_t nout = (int64_t)BC * (H / 2) * (W / 2);
  maxpool2_gather_kernel<<<tg_ew_grid(nout, RS_BLOCK), RS_BLOCK, 0, tg_stream(stream)>>>(x, idx, y, nout, H, W);
  return tg_launch_status();
}

int tg_inception_preprocess(const float* x, const float* mean, const float* stdv, float* out, int B, int C, int H, int W, int OH, int OW,
                            int stages, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(stdv); TG_CHECK_PTR(out);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(H); TG_CHECK_POS(W); TG_CHECK_POS(OH); TG_CHECK_POS(OW);
  if (stages < 0 || stages > 4) return TG_EINVAL;
  const int64_t total = (int64_t)B * C * OH * OW;
  // ATen's area_pixel_compute_scale for align_corners=True: (in - 1) / (out - 1) in fp32, 0 for a single output row / column
  const float rh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
  const float rw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
  inception_preprocess_kernel<<<tg_ew_grid(total, RS_BLOCK), RS_BLOCK, 0, tg_stream(stream)>>>(x, mean, stdv, out, C, H, W, OH, OW, stages,
                                                                                                rh, rw, total);
  return tg_launch_status();
}

}  // extern "C"
