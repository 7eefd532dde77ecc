// Bandwidth-bound elementwise kernels, scalar reductions, losses, Adam and EMA.
// All HBM-bound: 16-byte vector accesses (4 floats / lane / instruction), grid-stride,
// <= 2048 workgroups (cdna_hip_programming.md Guideline 11/13).
#include "common.h"

namespace {

constexpr int EW_BLOCK = 256;
constexpr int RED_BLOCK = 256;
constexpr int RED_MAX_BLOCKS = 1024;

// ------------------------------------------------------------------ generic maps
struct OpAdd { __device__ float operator()(float a, float b) const { return a + b; } };
struct OpMul { __device__ float operator()(float a, float b) const { return a * b; } };
struct OpLreluBwd {   // a = g, b = x
  float slope;
  __device__ float operator()(float g, float x) const { return x >= 0.f ? g : g * slope; }
};
struct OpTanhBwd {    // a = g, b = y
  __device__ float operator()(float g, float y) const { return g * (1.f - y * y); }
};
// ELU family (nn.ELU: alpha 1, scale 1; nn.SELU: alpha 1.6732632..., scale 1.0507009...), ATen's elu / elu_backward /
// elu_double_backward with input_scale 1: the negative branch is taken for x <= 0.
struct OpEluBwd {     // a = g, b = x; order 1: g f'(x), order 2: g f''(x)
  float negcoef, poscoef; int order;
  __device__ float operator()(float g, float x) const {
    if (x <= 0.f) return g * negcoef * expf(x);
    return order == 1 ? g * poscoef : 0.f;
  }
};
struct OpScaleAdd {   // s*a + b
  const float* s;
  __device__ float operator()(float a, float b) const { return (*s) * a + b; }
};

template <class Op>
__global__ void __launch_bounds__(EW_BLOCK) ew_binary(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, int64_t n, int vec, Op op) {
  const int64_t tid = blockIdx.x * (int64_t)EW_BLOCK + threadIdx.x;
  const int64_t stride = gridDim.x * (int64_t)EW_BLOCK;
  if (vec) {
    const int64_t n4 = n >> 2;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t i = tid; i < n4; i += stride) {
      float4 x = a4[i], y = b4[i], r;
      r.x = op(x.x, y.x); r.y = op(x.y, y.y); r.z = op(x.z, y.z); r.w = op(x.w, y.w);
      o4[i] = r;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = op(a[i], b[i]);
  } else {
    for (int64_t i = tid; i < n; i += stride) out[i] = op(a[i], b[i]);
  }
}

// out = ((a + b) + c) + d, c / d nullable: the gradient fan-in of a tensor with up to four consumers in ONE pass
// (autograd would run three separate adds: 9 tensor passes instead of 5)
__global__ void __launch_bounds__(EW_BLOCK) ew_add4(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                    const float* __restrict__ d, float* __restrict__ out, int64_t n, int vec) {
  const int64_t tid = blockIdx.x * (int64_t)EW_BLOCK + threadIdx.x;
  const int64_t stride = gridDim.x * (int64_t)EW_BLOCK;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += stride) {
      float4 x = reinterpret_cast<const float4*>(a)[i];
      const float4 y = reinterpret_cast<const float4*>(b)[i];
      x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      if (c) { const float4 z = reinterpret_cast<const float4*>(c)[i]; x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w; }
      if (d) { const float4 z = reinterpret_cast<const float4*>(d)[i]; x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w; }
      reinterpret_cast<float4*>(out)[i] = x;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = ((a[i] + b[i]) + (c ? c[i] : 0.f)) + (d ? d[i] : 0.f);
  } else {
    for (int64_t i = tid; i < n; i += stride) out[i] = ((a[i] + b[i]) + (c ? c[i] : 0.f)) + (d ? d[i] : 0.f);
  }
}

// dst (B, Cd, HW) <- src (B, Cs, HW): channels [0, min(Cs, Cd)) copied, channels >= Cs filled.  One thread per float4 (or
// float when HW % 4) of dst; HBM-bound, one pass.
template <int VEC>
__global__ void __launch_bounds__(EW_BLOCK) copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t planes_dst,
                                                                 int Cs, int Cd, int HW, float fill) {
  const int per = HW / VEC;
  const int64_t total = planes_dst * per;
  for (int64_t i = blockIdx.x * (int64_t)EW_BLOCK + threadIdx.x; i < total; i += gridDim.x * (int64_t)EW_BLOCK) {
    const int64_t plane = i / per;
    const int p = (int)(i - plane * per) * VEC;
    const int64_t b = plane / Cd;
    const int c = (int)(plane - b * Cd);
    if (VEC == 4) {
      float4 v = make_float4(fill, fill, fill, fill);
      if (c < Cs) v = *reinterpret_cast<const float4*>(src + (b * Cs + c) * (int64_t)HW + p);
      *reinterpret_cast<float4*>(dst + plane * (int64_t)HW + p) = v;
    } else {
      dst[plane * (int64_t)HW + p] = c < Cs ? src[(b * Cs + c) * (int64_t)HW + p] : fill;
    }
  }
}

struct OpScale { float alpha; __device__ float operator()(float x) const { return x * alpha; } };
struct OpScaleDev {
  const float* s; float alpha;
  __device__ float operator()(float x) const { return (alpha * (*s)) * x; }
};
struct OpTanh { __device__ float operator()(float x) const { return tanhf(x); } };
struct OpElu {
  float negcoef, poscoef;
  __device__ float operator()(float x) const { return x <= 0.f ? expm1f(x) * negcoef : x * poscoef; }
};
struct OpFill { float v; __device__ float operator()(float) const { return v; } };

template <class Op, bool READ>
__global__ void __launch_bounds__(EW_BLOCK) ew_unary(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                                     int vec, Op op) {
  const int64_t tid = blockIdx.x * (int64_t)EW_BLOCK + threadIdx.x;
  const int64_t stride = gridDim.x * (int64_t)EW_BLOCK;
  if (vec) {
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (int64_t i = tid; i < n4; i += stride) {
      float4 v = READ ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f), r;
      r.x = op(v.x); r.y = op(v.y); r.z = op(v.z); r.w = op(v.w);
      o4[i] = r;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = op(READ ? x[i] : 0.f);
  } else {
    for (int64_t i = tid; i < n; i += stride) out[i] = op(READ ? x[i] : 0.f);
  }
}

template <class Op>
int launch_binary(const float* a, const float* b, float* out, int64_t n, void* stream, Op op) {
  TG_CHECK_PTR(a); TG_CHECK_PTR(b); TG_CHECK_PTR(out);
  if (n < 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  const int vec = tg_aligned16(a) && tg_aligned16(b) && tg_aligned16(out);
  ew_binary<Op><<<tg_ew_grid((n + 3) / 4, EW_BLOCK), EW_BLOCK, 0, tg_stream(stream)>>>(a, b, out, n, vec, op);
  return tg_launch_status();
}
template <class Op, bool READ>
int launch_unary(const float* x, float* out, int64_t n, void* stream, Op op) {
  if (READ) TG_CHECK_PTR(x);
  TG_CHECK_PTR(out);
  if (n < 0) return TG_EINVAL;
  if (n == 0) return TG_OK;
  const int vec = (!READ || tg_aligned16(x)) && tg_aligned16(out);
  ew_unary<Op, READ><<<tg_ew_grid((n + 3) / 4, EW_BLOCK), EW_BLOCK, 0, tg_stream(stream)>>>(x, out, n, vec, op);
  return tg_launch_status();
}

// ------------------------------------------------------------------ scalar reductions (two deterministic stages)
struct RedDot { __device__ double term(const float* a, const float* b, int64_t i) const { return (double)a[i] * (double)b[i]; } };
struct RedSumSq { __device__ double term(const float* a, const float*, int64_t i) const { return (double)a[i] * (double)a[i]; } };

// A reduction that fits one block (<= 1024 elements: the logits of a batch, the quantile rows of an IQN head) is finished by that
// block -- the same (float)(alpha * sum) the second stage would form from its single partial -- instead of by a second launch.
struct Finish {
  float* out; double alpha; int accumulate;           // out == nullptr: several blocks, the partial goes to stage 2
  __device__ void put(double* __restrict__ partial, double acc) const {
    if (out) *out = (float)(alpha * acc) + (accumulate ? *out : 0.f);
    else partial[blockIdx.x] = acc;
  }
};

template <class R>
__global__ void __launch_bounds__(RED_BLOCK) reduce_stage1(const float* __restrict__ a, const float* __restrict__ b,
                                                           double* __restrict__ partial, int64_t n, R r, Finish fin) {
  __shared__ double scratch[32];
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)RED_BLOCK + threadIdx.x; i < n; i += gridDim.x * (int64_t)RED_BLOCK)
    acc += r.term(a, b, i);
  acc = block_sum_d(acc, scratch);
  if (threadIdx.x == 0) fin.put(partial, acc);
}

__global__ void __launch_bounds__(RED_BLOCK) reduce_stage2(const double* __restrict__ partial, int nparts, double alpha,
                                                           float* __restrict__ out, int accumulate = 0) {
  __shared__ double scratch[32];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += RED_BLOCK) acc += partial[i];
  acc = block_sum_d(acc, scratch);
  if (threadIdx.x == 0) *out = (float)(alpha * acc) + (accumulate ? *out : 0.f);
}

static inline int red_grid(int64_t n) {
  int64_t g = (n + RED_BLOCK * 4 - 1) / (RED_BLOCK * 4);
  if (g < 1) g = 1;
  if (g > RED_MAX_BLOCKS) g = RED_MAX_BLOCKS;
  return (int)g;
}

template <class R>
int launch_reduce(const float* a, const float* b, double alpha, float* out, float* ws, int64_t n, void* stream, R r, int accumulate = 0) {
  TG_CHECK_PTR(a); TG_CHECK_PTR(out); TG_CHECK_PTR(ws);
  if (n <= 0) return TG_EINVAL;
  double* partial = reinterpret_cast<double*>(ws);
  const int g = red_grid(n);
  reduce_stage1<R><<<g, RED_BLOCK, 0, tg_stream(stream)>>>(a, b, partial, n, r, Finish{g == 1 ? out : nullptr, alpha, accumulate});
  if (g > 1) reduce_stage2<<<1, RED_BLOCK, 0, tg_stream(stream)>>>(partial, g, alpha, out, accumulate);
  return tg_launch_status();
}

// ------------------------------------------------------------------ losses
// BCE-with-logits, mean reduction: loss_i = (1-t) x + max(-x,0) + log1p(exp(-|x|))
__global__ void __launch_bounds__(RED_BLOCK) bce_stage1(const float* __restrict__ x, const float* __restrict__ t,
                                                        float* __restrict__ dlogits, double* __restrict__ partial, int n, Finish fin) {
  __shared__ double scratch[32];
  double acc = 0.0;
  const float inv_n = 1.f / (float)n;
  for (int i = blockIdx.x * RED_BLOCK + threadIdx.x; i < n; i += gridDim.x * RED_BLOCK) {
    const float xi = x[i], ti = t[i];
    const float l = (1.f - ti) * xi + fmaxf(-xi, 0.f) + log1pf(expf(-fabsf(xi)));
    acc += (double)l;
    const float sig = 1.f / (1.f + expf(-xi));
    dlogits[i] = (sig - ti) * inv_n;
  }
  acc = block_sum_d(acc, scratch);
  if (threadIdx.x == 0) fin.put(partial, acc);
}

// IQN quantile Huber loss (models/iqn.py:111-130), out_dims = 1, row = q*B + b
__global__ void __launch_bounds__(RED_BLOCK) iqn_loss_stage1(const float* __restrict__ preds, const float* __restrict__ target,
                                                             const float* __restrict__ taus, float k, float* __restrict__ dpreds,
                                                             double* __restrict__ partial, int Q, int B, int G, Finish fin) {
  // G independent evaluations back to back (rows g*Q*B + q*B + b, targets g*B + b): the sum of their losses
  __shared__ double scratch[32];
  double acc = 0.0;
  const int n = Q * B;
  const float inv_b = 1.f / (float)B;
  for (int i = blockIdx.x * RED_BLOCK + threadIdx.x; i < n * G; i += gridDim.x * RED_BLOCK) {
    const int g = i / n, b = (i - g * n) % B;
    const float err = target[g * B + b] - preds[i];
    const float a = fabsf(err);
    const bool quad = a <= k;
    const float hub = quad ? 0.5f * err * err : k * (a - 0.5f * k);
    const float dh = quad ? err : (err > 0.f ? k : (err < 0.f ? -k : 0.f));
    const float w = fabsf(taus[i] - (err < 0.f ? 1.f : 0.f));
    acc += (double)(w * hub);
    dpreds[i] = -w * dh * inv_b;
  }
  acc = block_sum_d(acc, scratch);
  if (threadIdx.x == 0) fin.put(partial, acc);
}

__global__ void __launch_bounds__(EW_BLOCK) iqn_cos_embed_kernel(const float* __restrict__ taus, const float* __restrict__ range,
                                                                 float* __restrict__ out, int n, int dims) {
  const float pi = 3.14159265358979323846f;   // np.pi rounded to fp32, as torch does for tensor * python-float
  for (int i = blockIdx.x * EW_BLOCK + threadIdx.x; i < n * dims; i += gridDim.x * EW_BLOCK) {
    const int r = i / dims, c = i - r * dims;
    out[i] = cosf((taus[r] * pi) * range[c]);
  }
}

// ------------------------------------------------------------------ Adam / EMA
// hyper = [lr/bc1, sqrt(bc2), beta1, beta2, 1-beta1, 1-beta2]  (host doubles rounded to fp32)
__global__ void __launch_bounds__(EW_BLOCK) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const float* __restrict__ hyper, float eps, int64_t n) {
  const float step_size = hyper[0], bc2_sqrt = hyper[1], b1 = hyper[2], b2 = hyper[3], omb1 = hyper[4], omb2 = hyper[5];
  (void)b1;
  for (int64_t i = blockIdx.x * (int64_t)EW_BLOCK + threadIdx.x; i < n; i += gridDim.x * (int64_t)EW_BLOCK) {
    const float gi = g[i];
    float mi = m[i], vi = v[i];
    // torch.lerp(m, g, w = 1-beta1)
    const float d = gi - mi;
    mi = (omb1 < 0.5f) ? (mi + omb1 * d) : (gi - d * (1.f - omb1));
    vi = vi * b2;
    vi = vi + (omb2 * gi) * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ void __launch_bounds__(EW_BLOCK) ema_kernel(float* __restrict__ t, const float* __restrict__ p, float lr, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)EW_BLOCK + threadIdx.x; i < n; i += gridDim.x * (int64_t)EW_BLOCK) {
    const float ti = t[i];
    t[i] = ti + (p[i] - ti) * lr;
  }
}

}  // namespace

extern "C" {

int tg_add(const float* a, const float* b, float* out, int64_t n, void* stream) { return launch_binary(a, b, out, n, stream, OpAdd{}); }
int tg_add4(const float* a, const float* b, const float* c, const float* d, float* out, int64_t n, void* stream) {
  TG_CHECK_PTR(a); TG_CHECK_PTR(b); TG_CHECK_PTR(out);
  if (n <= 0) return TG_EINVAL;
  const int vec = tg_aligned16(a) && tg_aligned16(b) && tg_aligned16(out) && (!c || tg_aligned16(c)) && (!d || tg_aligned16(d));
  ew_add4<<<tg_ew_grid((n + 3) / 4, EW_BLOCK), EW_BLOCK, 0, tg_stream(stream)>>>(a, b, c, d, out, n, vec);
  return tg_launch_status();
}
int tg_mul(const float* a, const float* b, float* out, int64_t n, void* stream) { return launch_binary(a, b, out, n, stream, OpMul{}); }
int tg_lrelu_bwd(const float* g, const float* x, float slope, float* out, int64_t n, void* stream) {
  return launch_binary(g, x, out, n, stream, OpLreluBwd{slope});
}
int tg_tanh_bwd(const float* g, const float* y, float* out, int64_t n, void* stream) { return launch_binary(g, y, out, n, stream, OpTanhBwd{}); }
int tg_scale_add_dev(const float* s, const float* a, const float* b, float* out, int64_t n, void* stream) {
  TG_CHECK_PTR(s);
  return launch_binary(a, b, out, n, stream, OpScaleAdd{s});
}
int tg_scale(const float* x, float alpha, float* out, int64_t n, void* stream) {
  return launch_unary<OpScale, true>(x, out, n, stream, OpScale{alpha});
}
int tg_scale_dev(const float* s, float alpha, const float* x, float* out, int64_t n, void* stream) {
  TG_CHECK_PTR(s);
  return launch_unary<OpScaleDev, true>(x, out, n, stream, OpScaleDev{s, alpha});
}
int tg_tanh_fwd(const float* x, float* y, int64_t n, void* stream) { return launch_unary<OpTanh, true>(x, y, n, stream, OpTanh{}); }
int tg_elu_fwd(const float* x, float alpha, float scale, float* y, int64_t n, void* stream) {
  return launch_unary<OpElu, true>(x, y, n, stream, OpElu{alpha * scale, scale});
}
int tg_elu_bwd(const float* g, const float* x, float alpha, float scale, int order, float* out, int64_t n, void* stream) {
  if (order != 1 && order != 2) return TG_EINVAL;
  return launch_binary(g, x, out, n, stream, OpEluBwd{alpha * scale, scale, order});
}
int tg_copy_channels(const float* src, float* dst, int B, int Cs, int Cd, int HW, float fill, void* stream) {
  TG_CHECK_PTR(src); TG_CHECK_PTR(dst);
  TG_CHECK_POS(B); TG_CHECK_POS(Cs); TG_CHECK_POS(Cd); TG_CHECK_POS(HW);
  const int64_t planes_dst = (int64_t)B * Cd;
  if (HW % 4 == 0 && tg_aligned16(src) && tg_aligned16(dst))
    copy_channels_kernel<4><<<tg_ew_grid(planes_dst * (HW / 4), EW_BLOCK), EW_BLOCK, 0, tg_stream(stream)>>>(src, dst, planes_dst, Cs, Cd, HW, fill);
  else
    copy_channels_kernel<1><<<tg_ew_grid(planes_dst * HW, EW_BLOCK), EW_BLOCK, 0, tg_stream(stream)>>>(src, dst, planes_dst, Cs, Cd, HW, fill);
  return tg_launch_status();
}
int tg_fill(float* x, float value, int64_t n, void* stream) { return launch_unary<OpFill, false>(nullptr, x, n, stream, OpFill{value}); }

size_t tg_reduce_workspace(int64_t n) { (void)n; return (size_t)RED_MAX_BLOCKS * sizeof(double); }

int tg_dot(const float* a, const float* b, float alpha, float* out, float* workspace, int64_t n, int accumulate, void* stream) {
  TG_CHECK_PTR(b);
  return launch_reduce(a, b, (double)alpha, out, workspace, n, stream, RedDot{}, accumulate);
}
int tg_sumsq(const float* x, float alpha, float* out, float* workspace, int64_t n, void* stream) {
  return launch_reduce(x, nullptr, (double)alpha, out, workspace, n, stream, RedSumSq{});
}

int tg_bce_logits(const float* logits, const float* targets, float* loss, float* dlogits, float* workspace, int n, void* stream) {
  TG_CHECK_PTR(logits); TG_CHECK_PTR(targets); TG_CHECK_PTR(loss); TG_CHECK_PTR(dlogits); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(n);
  double* partial = reinterpret_cast<double*>(workspace);
  const int g = red_grid(n);
  bce_stage1<<<g, RED_BLOCK, 0, tg_stream(stream)>>>(logits, targets, dlogits, partial, n, Finish{g == 1 ? loss : nullptr, 1.0 / (double)n, 0});
  if (g > 1) reduce_stage2<<<1, RED_BLOCK, 0, tg_stream(stream)>>>(partial, g, 1.0 / (double)n, loss);
  return tg_launch_status();
}

int tg_iqn_loss_groups(const float* preds, const float* target, const float* taus, float k, float* loss, float* dpreds,
                       float* workspace, int Q, int B, int groups, void* stream) {
  TG_CHECK_PTR(preds); TG_CHECK_PTR(target); TG_CHECK_PTR(taus); TG_CHECK_PTR(loss); TG_CHECK_PTR(dpreds); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(Q); TG_CHECK_POS(B); TG_CHECK_POS(groups);
  double* partial = reinterpret_cast<double*>(workspace);
  const int g = red_grid((int64_t)Q * B * groups);
  iqn_loss_stage1<<<g, RED_BLOCK, 0, tg_stream(stream)>>>(preds, target, taus, k, dpreds, partial, Q, B, groups,
                                                          Finish{g == 1 ? loss : nullptr, 1.0 / (double)B, 0});
  if (g > 1) reduce_stage2<<<1, RED_BLOCK, 0, tg_stream(stream)>>>(partial, g, 1.0 / (double)B, loss);
  return tg_launch_status();
}
int tg_iqn_loss(const float* preds, const float* target, const float* taus, float k, float* loss, float* dpreds,
                float* workspace, int Q, int B, void* stream) {
  return tg_iqn_loss_groups(preds, target, taus, k, loss, dpreds, workspace, Q, B, 1, stream);
}

int tg_iqn_cos_embed(const float* taus, const float* range, float* out, int n, int dims, void* stream) {
  TG_CHECK_PTR(taus); TG_CHECK_PTR(range); TG_CHECK_PTR(out);
  TG_CHECK_POS(n); TG_CHECK_POS(dims);
  iqn_cos_embed_kernel<<<tg_ew_grid((int64_t)n * dims, EW_BLOCK), EW_BLOCK, 0, tg_stream(stream)>>>(taus, range, out, n, dims);
  return tg_launch_status();
}

int tg_adam_step(float* p, const float* g, float* m, float* v, const float* hyper, float eps, int64_t n, void* stream) {
  TG_CHECK_PTR(p); TG_CHECK_PTR(g); TG_CHECK_PTR(m); TG_CHECK_PTR(v); TG_CHECK_PTR(hyper);
  if (n <= 0) return TG_EINVAL;
  adam_kernel<<<tg_ew_grid(n, EW_BLOCK), EW_BLOCK, 0, tg_stream(stream)>>>(p, g, m, v, hyper, eps, n);
  return tg_launch_status();
}

int tg_ema(float* t, const float* p, float lr, int64_t n, void* stream) {
  TG_CHECK_PTR(t); TG_CHECK_PTR(p);
  if (n <= 0) return TG_EINVAL;
  ema_kernel<<<tg_ew_grid(n, EW_BLOCK), EW_BLOCK, 0, tg_stream(stream)>>>(t, p, lr, n);
  return tg_launch_status();
}

int tg_version(void) { return 100; }
const char* tg_arch(void) { return "gfx950"; }

}  // extern "C"
