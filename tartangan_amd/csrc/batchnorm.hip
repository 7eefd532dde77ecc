// BatchNorm2d (train / eval) fused with LeakyReLU: forward, backward and the second
// backward needed by the R1 penalty.  HBM-bound; per-channel sums are two-stage and
// deterministic (fp32 per-thread partials over <= ~64 elements, fp64 across threads/blocks).
//
// Notation per channel (n = B*HW): xhat = (x-mean)*invstd, y = gamma*xhat+beta,
// s = lrelu'(y) in {1, slope}, gyh = gz*s.
//   fwd : z = y*s
//   bwd : gbeta = S(gyh); ggamma = S(gyh*xhat); gx = gamma*invstd*(gyh - gbeta/n - xhat*ggamma/n)
//   dbwd: see tg_bn_act_dbwd below (derivation in DESIGN.md "BatchNorm second backward").
#include <stdlib.h>

#include "planes.h"

namespace {

using planes::BLOCK;
using planes::Chan;

// workspace layout: [C*S*5] doubles of partial sums, then [C*COEF] floats of per-channel coefficients
constexpr int COEF = 8;
constexpr int MAXK = 5;

__device__ __forceinline__ float bn_y(float x, float a, float b) { return fmaf(x, a, b); }   // a = gamma*invstd, b = beta - mean*a

struct Parts {
  double* partial;
  float* coef;
};
static inline Parts split_ws(float* ws, int B, int C, int HW) {
  const int S = planes::splits(B, C, HW);
  Parts p;
  p.partial = reinterpret_cast<double*>(ws);
  p.coef = reinterpret_cast<float*>(p.partial + (size_t)C * S * MAXK);
  return p;
}

// ------------------------------------------------------------------ statistics
struct RedStats {
  static constexpr int K = 2;
  const float* x;
  int C, HW;
  float pivot;
  __device__ void init(const Chan& ch) { pivot = x[ch.base + (int64_t)ch.c * HW]; }   // first element of the (group's) channel: shift against cancellation
  using V = float4;
  __device__ V ld4(int64_t off) const { return *reinterpret_cast<const float4*>(x + off); }
  __device__ void acc(const V& v, float* a) {
    const float d0 = v.x - pivot, d1 = v.y - pivot, d2 = v.z - pivot, d3 = v.w - pivot;
    a[0] += (d0 + d1) + (d2 + d3);
    a[1] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  }
  __device__ void acc1(int64_t off, float* a) {
    const float d = x[off] - pivot;
    a[0] += d;
    a[1] += d * d;
  }
};

// mean / variance of virtual channel vc (group g's channel c) from the stage-1 partials; wave-collective
__device__ __forceinline__ void finish_stats(const double* __restrict__ partial, const float* __restrict__ x, int vc, int c, int64_t base,
                                             int B, int HW, int S, double& m, double& var) {
  const double n = (double)B * HW;
  const double pivot = (double)x[base + (int64_t)c * HW];
  const double s1 = planes::gather(partial, vc, S, 2, 0) / n;
  const double s2 = planes::gather(partial, vc, S, 2, 1) / n;
  m = pivot + s1;
  var = s2 - s1 * s1;
  if (var < 0.0) var = 0.0;
}
// one running-statistics update, rounded to fp32 like a separate BatchNorm call would leave it
__device__ __forceinline__ void running_update(float& rmf, float& rvf, double m, double var, double n, int rep, float momentum) {
  const double nr = n * rep;                     // element count of the tensor these statistics stand for
  const double unbiased = nr > 1.0 ? var * (nr / (nr - 1.0)) : var;
  rmf = (float)((1.0 - (double)momentum) * (double)rmf + (double)momentum * m);
  rvf = (float)((1.0 - (double)momentum) * (double)rvf + (double)momentum * unbiased);
}

// one wave per CHANNEL; the groups are finished one after the other, so the running statistics see them in order
__global__ void stats_stage2(const double* __restrict__ partial, const float* __restrict__ x, float* __restrict__ mean,
                             float* __restrict__ invstd, float* __restrict__ rm, float* __restrict__ rv,
                             int64_t* __restrict__ nbt, float momentum, float eps, int B, int C, int HW, int S, int rep, int G) {
  const int c = blockIdx.x;
  float rmf = 0.f, rvf = 0.f;
  if (rm != nullptr) { rmf = rm[c]; rvf = rv[c]; }
  for (int g = 0; g < G; ++g) {
    const int vc = g * C + c;
    double m, var;
    finish_stats(partial, x, vc, c, (int64_t)g * B * C * HW, B, HW, S, m, var);
    if (threadIdx.x == 0) {
      mean[vc] = (float)m;
      invstd[vc] = (float)(1.0 / sqrt(var + (double)eps));
    }
    running_update(rmf, rvf, m, var, (double)B * HW, rep, momentum);
  }
  if (threadIdx.x != 0) return;
  if (c == 0 && nbt != nullptr) *nbt += G;
  if (rm != nullptr) { rm[c] = rmf; rv[c] = rvf; }
}

__global__ void eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ mean,
                                  float* __restrict__ invstd, float eps, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  invstd[c] = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
}

// ------------------------------------------------------------------ forward apply
struct FwdBody {
  const float* x; float* z;
  const float *mean, *invstd, *gamma, *beta;
  float slope;
  __device__ void coeffs(const Chan& ch, float& a, float& b) const {
    a = gamma[ch.c] * invstd[ch.vc];
    b = beta[ch.c] - mean[ch.vc] * a;
  }
  __device__ void vec4(const Chan& ch, int64_t off) const {
    float a, b; coeffs(ch, a, b);
    const float4 v = *reinterpret_cast<const float4*>(x + off);
    float4 r;
    float y;
    y = bn_y(v.x, a, b); r.x = y >= 0.f ? y : y * slope;
    y = bn_y(v.y, a, b); r.y = y >= 0.f ? y : y * slope;
    y = bn_y(v.z, a, b); r.z = y >= 0.f ? y : y * slope;
    y = bn_y(v.w, a, b); r.w = y >= 0.f ? y : y * slope;
    *reinterpret_cast<float4*>(z + off) = r;
  }
  __device__ void one(const Chan& ch, int64_t off) const {
    float a, b; coeffs(ch, a, b);
    const float y = bn_y(x[off], a, b);
    z[off] = y >= 0.f ? y : y * slope;
  }
};

// statistics finished inside the apply pass (see planes::map_big_begin); arithmetic identical to stats_stage2
struct FwdStatsBody {
  const float* x; float* z;
  const double* partial;
  float *mean, *invstd, *rm, *rv;
  int64_t* nbt;
  const float *gamma, *beta;
  float slope, momentum, eps;
  int B, HW, S, rep, C, G;
  float a, b;
  __device__ void begin(const Chan& ch, bool lead) {
    const int c = ch.c;
    double m, var;
    finish_stats(partial, x, ch.vc, c, ch.base, B, HW, S, m, var);
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    a = gamma[c] * rf;
    b = beta[c] - mf * a;
    if (!lead) return;                        // (uniform per block)
    if (threadIdx.x == 0) {
      if (ch.vc == 0 && nbt != nullptr) *nbt += G;
      mean[ch.vc] = mf;
      invstd[ch.vc] = rf;
    }
    if (ch.vc == c && rm != nullptr) {
      // group 0's lead block publishes the running statistics for ALL groups of its channel, in group order (two
      // forwards of the reference update them real-then-fake); the other groups' sums are a few hundred bytes from L2
      float rmf = rm[c], rvf = rv[c];
      running_update(rmf, rvf, m, var, (double)B * HW, rep, momentum);
      for (int g = 1; g < G; ++g) {
        double mg, vg;
        finish_stats(partial, x, g * C + c, c, (int64_t)g * B * C * HW, B, HW, S, mg, vg);
        running_update(rmf, rvf, mg, vg, (double)B * HW, rep, momentum);
      }
      if (threadIdx.x == 0) { rm[c] = rmf; rv[c] = rvf; }
    }
  }
  using V = float4;
  __device__ V ld(int64_t off) const { return *reinterpret_cast<const float4*>(x + off); }
  __device__ void st(const Chan&, int64_t off, const V& v) const {
    float4 r;
    float y;
    y = bn_y(v.x, a, b); r.x = y >= 0.f ? y : y * slope;
    y = bn_y(v.y, a, b); r.y = y >= 0.f ? y : y * slope;
    y = bn_y(v.z, a, b); r.z = y >= 0.f ? y : y * slope;
    y = bn_y(v.w, a, b); r.w = y >= 0.f ? y : y * slope;
    *reinterpret_cast<float4*>(z + off) = r;
  }
};

// ------------------------------------------------------------------ first backward
struct RedBwd {
  static constexpr int K = 2;
  const float *gz, *x, *mean, *invstd, *gamma, *beta;
  float slope;
  float a, b, mu, r;
  __device__ void init(const Chan& ch) {
    r = invstd[ch.vc]; mu = mean[ch.vc];
    a = gamma[ch.c] * r;
    b = beta[ch.c] - mu * a;
  }
  __device__ void elem(float g, float xv, float* acc) {
    const float y = bn_y(xv, a, b);
    const float gyh = y >= 0.f ? g : g * slope;
    acc[0] += gyh;
    acc[1] += gyh * ((xv - mu) * r);
  }
  struct V { float4 g, v; };
  __device__ V ld4(int64_t off) const {
    return V{*reinterpret_cast<const float4*>(gz + off), *reinterpret_cast<const float4*>(x + off)};
  }
  __device__ void acc(const V& q, float* acc) {
    elem(q.g.x, q.v.x, acc); elem(q.g.y, q.v.y, acc); elem(q.g.z, q.v.z, acc); elem(q.g.w, q.v.w, acc);
  }
  __device__ void acc1(int64_t off, float* acc) { elem(gz[off], x[off], acc); }
};

// one wave per CHANNEL: parameter gradients are the sums over all groups, the input-gradient coefficients per group
__global__ void bwd_stage2(const double* __restrict__ partial, float* __restrict__ ggamma, float* __restrict__ gbeta,
                           float* __restrict__ coef, int B, int C, int HW, int S, int accumulate, int G) {
  const int c = blockIdx.x;
  const double n = (double)B * HW;
  double tb = 0.0, tg = 0.0;
  for (int g = 0; g < G; ++g) {
    const int vc = g * C + c;
    const double sb = planes::gather(partial, vc, S, 2, 0);
    const double sg = planes::gather(partial, vc, S, 2, 1);
    tb += sb; tg += sg;
    if (threadIdx.x == 0) {
      coef[vc * COEF + 0] = (float)(sb / n);
      coef[vc * COEF + 1] = (float)(sg / n);
    }
  }
  if (threadIdx.x != 0) return;
  gbeta[c] = (float)tb + (accumulate ? gbeta[c] : 0.f);
  ggamma[c] = (float)tg + (accumulate ? ggamma[c] : 0.f);
}

struct BwdBody {
  const float *gz, *x; float* gx;
  const float *mean, *invstd, *gamma, *beta, *coef;
  float slope; int training;
  const float* add;          // nullable: a second gradient of x (the R1 penalty's second-order term) added in the same pass
  int add_vc_end;            // ... for the virtual channels below this (the leading groups; `add` holds only their images)
  __device__ float elem(float g, float xv, float a, float b, float mu, float r, float k1, float k2) const {
    const float y = bn_y(xv, a, b);
    const float gyh = y >= 0.f ? g : g * slope;
    return training ? a * (gyh - k1 - ((xv - mu) * r) * k2) : a * gyh;
  }
  __device__ void vec4(const Chan& ch, int64_t off) const {
    const float r = invstd[ch.vc], mu = mean[ch.vc], a = gamma[ch.c] * r, b = beta[ch.c] - mu * a;
    const float k1 = coef[ch.vc * COEF + 0], k2 = coef[ch.vc * COEF + 1];
    const float4 g = *reinterpret_cast<const float4*>(gz + off);
    const float4 v = *reinterpret_cast<const float4*>(x + off);
    float4 o;
    o.x = elem(g.x, v.x, a, b, mu, r, k1, k2); o.y = elem(g.y, v.y, a, b, mu, r, k1, k2);
    o.z = elem(g.z, v.z, a, b, mu, r, k1, k2); o.w = elem(g.w, v.w, a, b, mu, r, k1, k2);
    if (add && ch.vc < add_vc_end) { const float4 t = *reinterpret_cast<const float4*>(add + off); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
    *reinterpret_cast<float4*>(gx + off) = o;
  }
  __device__ void one(const Chan& ch, int64_t off) const {
    const float r = invstd[ch.vc], mu = mean[ch.vc], a = gamma[ch.c] * r, b = beta[ch.c] - mu * a;
    gx[off] = elem(gz[off], x[off], a, b, mu, r, coef[ch.vc * COEF + 0], coef[ch.vc * COEF + 1]) + ((add && ch.vc < add_vc_end) ? add[off] : 0.f);
  }
};

// reduction finished inside the apply pass; arithmetic identical to bwd_stage2 + BwdBody
struct BwdSumsBody {
  const float *gz, *x; float* gx;
  const double* partial;
  const float *mean, *invstd, *gamma, *beta;
  float *ggamma, *gbeta;
  float slope; int training, accumulate, B, HW, S;
  float a, b, mu, r, k1, k2;
  const float* add;
  int C, G, add_vc_end;
  bool use_add;
  __device__ void begin(const Chan& ch, bool lead) {
    const int c = ch.c;
    const double n = (double)B * HW;
    const double sb = planes::gather(partial, ch.vc, S, 2, 0);
    const double sg = planes::gather(partial, ch.vc, S, 2, 1);
    r = invstd[ch.vc]; mu = mean[ch.vc];
    a = gamma[c] * r;
    b = beta[c] - mu * a;
    k1 = (float)(sb / n);
    k2 = (float)(sg / n);
    use_add = add != nullptr && ch.vc < add_vc_end;
    if (lead && ch.vc == c) {                  // group 0's lead block publishes the parameter gradients: sums over ALL groups
      double tb = sb, tg = sg;
      for (int g = 1; g < G; ++g) {
        tb += planes::gather(partial, g * C + c, S, 2, 0);
        tg += planes::gather(partial, g * C + c, S, 2, 1);
      }
      if (threadIdx.x == 0) {
        gbeta[c] = (float)tb + (accumulate ? gbeta[c] : 0.f);
        ggamma[c] = (float)tg + (accumulate ? ggamma[c] : 0.f);
      }
    }
  }
  __device__ float elem(float g, float xv) const {
    const float y = bn_y(xv, a, b);
    const float gyh = y >= 0.f ? g : g * slope;
    return training ? a * (gyh - k1 - ((xv - mu) * r) * k2) : a * gyh;
  }
  struct V { float4 g, v, t; };
  __device__ V ld(int64_t off) const {
    return V{*reinterpret_cast<const float4*>(gz + off), *reinterpret_cast<const float4*>(x + off),
             use_add ? *reinterpret_cast<const float4*>(add + off) : make_float4(0.f, 0.f, 0.f, 0.f)};
  }
  __device__ void st(const Chan&, int64_t off, const V& q) const {
    float4 o;
    o.x = elem(q.g.x, q.v.x) + q.t.x; o.y = elem(q.g.y, q.v.y) + q.t.y; o.z = elem(q.g.z, q.v.z) + q.t.z; o.w = elem(q.g.w, q.v.w) + q.t.w;
    *reinterpret_cast<float4*>(gx + off) = o;
  }
};

// ------------------------------------------------------------------ second backward
struct RedDbwd {
  static constexpr int K = 5;   // S1 = S(v), S2 = S(v xhat), S3 = S(gyh), S4 = S(gyh xhat), S5 = S(v gyh)
  const float *v, *gz, *x, *mean, *invstd, *gamma, *beta;
  float slope;
  float a, b, mu, r;
  __device__ void init(const Chan& ch) {
    const int c = ch.c;
    r = invstd[c]; mu = mean[c];
    a = gamma[c] * r;
    b = beta[c] - mu * a;
  }
  __device__ void elem(float vv, float g, float xv, float* acc) {
    const float y = bn_y(xv, a, b);
    const float gyh = y >= 0.f ? g : g * slope;
    const float xh = (xv - mu) * r;
    acc[0] += vv; acc[1] += vv * xh; acc[2] += gyh; acc[3] += gyh * xh; acc[4] += vv * gyh;
  }
  struct V { float4 w, g, q; };
  __device__ V ld4(int64_t off) const {
    return V{*reinterpret_cast<const float4*>(v + off), *reinterpret_cast<const float4*>(gz + off), *reinterpret_cast<const float4*>(x + off)};
  }
  __device__ void acc(const V& t, float* acc) {
    elem(t.w.x, t.g.x, t.q.x, acc); elem(t.w.y, t.g.y, t.q.y, acc); elem(t.w.z, t.g.z, t.q.z, acc); elem(t.w.w, t.g.w, t.q.w, acc);
  }
  __device__ void acc1(int64_t off, float* acc) { elem(v[off], gz[off], x[off], acc); }
};

// coef: 0 S1/n  1 S2/n  2 cg  3 cv  4 qm  5 qx  6 gamma*A*r^2/n
__global__ void dbwd_stage2(const double* __restrict__ partial, const float* __restrict__ gamma, const float* __restrict__ invstd,
                            const float* __restrict__ vgamma, float* __restrict__ adj_gamma, float* __restrict__ coef,
                            int B, int C, int HW, int S, int accumulate) {
  const int c = blockIdx.x;                       // one wave per channel
  const double n = (double)B * HW;
  const double S1 = planes::gather(partial, c, S, 5, 0), S2 = planes::gather(partial, c, S, 5, 1);
  const double S3 = planes::gather(partial, c, S, 5, 2), S4 = planes::gather(partial, c, S, 5, 3);
  const double S5 = planes::gather(partial, c, S, 5, 4);
  if (threadIdx.x != 0) return;
  const double r = (double)invstd[c], g = (double)gamma[c];
  const double vg = vgamma ? (double)vgamma[c] : 0.0;
  const double A = S5 - S1 * S3 / n - S2 * S4 / n;
  const double cg = S4 / n, cv = S2 / n;
  const double qm = -g * r * (cg * S1 / n + cv * S3 / n) + vg * S3 / n;
  const double qx = -g * r * (cg * S2 / n + cv * S4 / n) + vg * S4 / n;
  adj_gamma[c] = (float)(r * A) + (accumulate ? adj_gamma[c] : 0.f);
  float* k = coef + c * COEF;
  k[0] = (float)(S1 / n); k[1] = (float)(S2 / n); k[2] = (float)cg; k[3] = (float)cv;
  k[4] = (float)qm; k[5] = (float)qx; k[6] = (float)(g * A * r * r / n);
}

struct DbwdBody {
  const float *v, *gz, *x; float *adj_gz, *adj_x;
  const float *mean, *invstd, *gamma, *beta, *vgamma, *vbeta, *coef;
  float slope;
  struct Ch { float r, mu, a, b, gr, vg, vb, k0, k1, cg, cv, qm, qx, k6; };
  __device__ Ch load(int c) const {
    Ch h;
    h.r = invstd[c]; h.mu = mean[c]; h.a = gamma[c] * h.r; h.b = beta[c] - h.mu * h.a; h.gr = h.a;
    h.vg = vgamma ? vgamma[c] : 0.f; h.vb = vbeta ? vbeta[c] : 0.f;
    const float* k = coef + c * COEF;
    h.k0 = k[0]; h.k1 = k[1]; h.cg = k[2]; h.cv = k[3]; h.qm = k[4]; h.qx = k[5]; h.k6 = k[6];
    return h;
  }
  __device__ void elem(const Ch& h, float vv, float g, float xv, float& o_gz, float& o_x) const {
    const float y = bn_y(xv, h.a, h.b);
    const float s = y >= 0.f ? 1.f : slope;
    const float gyh = g * s;
    const float xh = (xv - h.mu) * h.r;
    const float pv = vv - h.k0 - xh * h.k1;
    o_gz = (h.gr * pv + h.vg * xh + h.vb) * s;
    const float q = -h.gr * (h.cg * vv + h.cv * gyh) + h.vg * gyh;
    o_x = h.r * (q - h.qm - xh * h.qx) - h.k6 * xh;
  }
  __device__ void vec4(const Chan& ch, int64_t off) const {
    const Ch h = load(ch.c);
    const float4 w = *reinterpret_cast<const float4*>(v + off);
    const float4 g = *reinterpret_cast<const float4*>(gz + off);
    const float4 q = *reinterpret_cast<const float4*>(x + off);
    float4 og, ox;
    elem(h, w.x, g.x, q.x, og.x, ox.x); elem(h, w.y, g.y, q.y, og.y, ox.y);
    elem(h, w.z, g.z, q.z, og.z, ox.z); elem(h, w.w, g.w, q.w, og.w, ox.w);
    *reinterpret_cast<float4*>(adj_gz + off) = og;
    *reinterpret_cast<float4*>(adj_x + off) = ox;
  }
  __device__ void one(const Chan& ch, int64_t off) const {
    const Ch h = load(ch.c);
    elem(h, v[off], gz[off], x[off], adj_gz[off], adj_x[off]);
  }
};

// the same with the reduction finished inside the apply pass (big planes): arithmetic identical to dbwd_stage2 + DbwdBody
struct DbwdSumsBody {
  const float *v, *gz, *x; float *adj_gz, *adj_x;
  const double* partial;
  const float *mean, *invstd, *gamma, *beta, *vgamma, *vbeta;
  float* adj_gamma;
  float slope; int accumulate, B, HW, S;
  DbwdBody::Ch h;
  __device__ void begin(const Chan& ch, bool lead) {
    const int c = ch.c;
    const double n = (double)B * HW;
    const double S1 = planes::gather(partial, c, S, 5, 0), S2 = planes::gather(partial, c, S, 5, 1);
    const double S3 = planes::gather(partial, c, S, 5, 2), S4 = planes::gather(partial, c, S, 5, 3);
    const double S5 = planes::gather(partial, c, S, 5, 4);
    const double r = (double)invstd[c], g = (double)gamma[c];
    const double vg = vgamma ? (double)vgamma[c] : 0.0;
    const double A = S5 - S1 * S3 / n - S2 * S4 / n;
    const double cg = S4 / n, cv = S2 / n;
    const double qm = -g * r * (cg * S1 / n + cv * S3 / n) + vg * S3 / n;
    const double qx = -g * r * (cg * S2 / n + cv * S4 / n) + vg * S4 / n;
    h.r = invstd[c]; h.mu = mean[c]; h.a = gamma[c] * h.r; h.b = beta[c] - h.mu * h.a; h.gr = h.a;
    h.vg = vgamma ? vgamma[c] : 0.f; h.vb = vbeta ? vbeta[c] : 0.f;
    h.k0 = (float)(S1 / n); h.k1 = (float)(S2 / n); h.cg = (float)cg; h.cv = (float)cv;
    h.qm = (float)qm; h.qx = (float)qx; h.k6 = (float)(g * A * r * r / n);
    if (lead && threadIdx.x == 0) adj_gamma[c] = (float)(r * A) + (accumulate ? adj_gamma[c] : 0.f);
  }
  struct V { float4 w, g, q; };
  __device__ V ld(int64_t off) const {
    return V{*reinterpret_cast<const float4*>(v + off), *reinterpret_cast<const float4*>(gz + off), *reinterpret_cast<const float4*>(x + off)};
  }
  __device__ void st(const Chan&, int64_t off, const V& t) const {
    const DbwdBody e{v, gz, x, adj_gz, adj_x, mean, invstd, gamma, beta, vgamma, vbeta, nullptr, slope};
    float4 og, ox;
    e.elem(h, t.w.x, t.g.x, t.q.x, og.x, ox.x); e.elem(h, t.w.y, t.g.y, t.q.y, og.y, ox.y);
    e.elem(h, t.w.z, t.g.z, t.q.z, og.z, ox.z); e.elem(h, t.w.w, t.g.w, t.q.w, og.w, ox.w);
    *reinterpret_cast<float4*>(adj_gz + off) = og;
    *reinterpret_cast<float4*>(adj_x + off) = ox;
  }
};

static inline int chan_grid(int C) { return (C + 63) / 64; }

// ------------------------------------------------------------------ small tensors: one workgroup per channel
// When a channel has <= SMALL_N elements (4x4 ... 16x16 layers at batch 64) the three-launch pipeline above is
// pure launch latency (~20 us for a few hundred KB).  One kernel, one workgroup per channel: reduce, then apply
// (the second read hits L2).  Same arithmetic as the large path (fp32 per-thread partials, fp64 combine).
constexpr int SMALL_N = 16384;
static inline int small_n_limit() {            // TG_BN_SMALL_N: development knob (<= SMALL_N), read once
  static const int v = [] {
    const char* e = getenv("TG_BN_SMALL_N");
    const int k = e ? atoi(e) : SMALL_N;
    return k < 0 ? 0 : (k > SMALL_N ? SMALL_N : k);
  }();
  return v;
}
static inline bool small_case(int B, int C, int HW) { return (int64_t)B * HW <= small_n_limit() && C >= 8; }

// 1024 threads per channel, <= 16 elements per thread held in registers between the reduce and the apply pass:
// every global load of a thread is issued up front (independent), nothing is read twice.
constexpr int SB = 1024, SPER = SMALL_N / SB;

template <int N, int THREADS = SB>
struct SmallIdxN {
  int64_t off[N];
  bool ok[N];
  __device__ __forceinline__ SmallIdxN(int B, int C, int HW, int c, int64_t base = 0) {
    const int n = B * HW;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int e = threadIdx.x + i * THREADS;
      const int b = e / HW, p = e - b * HW;
      ok[i] = e < n;
      off[i] = base + ((int64_t)b * C + c) * HW + p;
    }
  }
};
using SmallIdx = SmallIdxN<SPER>;

__device__ __forceinline__ double block_sum_d1024(double v, double* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < SB / 64; ++i) r += scratch[i];
  return r;
}

// One workgroup per CHANNEL; its G groups (B images each) are normalised one after the other, so the running statistics
// are updated in group order.
template <int GT>
__global__ void __launch_bounds__(SB) bn_small_fwd_kernel(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ rm, float* __restrict__ rv, int64_t* __restrict__ nbt,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                          float momentum, float eps, float* __restrict__ z, int B, int C, int HW, int rep,
                                                          int Grt) {
  __shared__ double scratch[32];
  const int c = blockIdx.x;
  const int G = GT > 0 ? GT : Grt;
  float rmf = 0.f, rvf = 0.f;
  if (rm != nullptr) { rmf = rm[c]; rvf = rv[c]; }
  const double n = (double)B * HW;
  for (int g = 0; g < G; ++g) {
    const int64_t base = (int64_t)g * B * C * HW;
    const SmallIdx ix(B, C, HW, c, base);
    const float pivot = x[base + (int64_t)c * HW];
    float xv[SPER];
#pragma unroll
    for (int i = 0; i < SPER; ++i) xv[i] = ix.ok[i] ? x[ix.off[i]] : pivot;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < SPER; ++i) { const float d = xv[i] - pivot; a0 += d; a1 += d * d; }
    const double s1 = block_sum_d1024((double)a0, scratch) / n;
    const double s2 = block_sum_d1024((double)a1, scratch) / n;
    const double m = (double)pivot + s1;
    double var = s2 - s1 * s1;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
      mean[g * C + c] = mf;
      invstd[g * C + c] = rf;
    }
    running_update(rmf, rvf, m, var, n, rep, momentum);
    const float a = gamma[c] * rf, b = beta[c] - mf * a;
#pragma unroll
    for (int i = 0; i < SPER; ++i)
      if (ix.ok[i]) {
        const float y = bn_y(xv[i], a, b);
        z[ix.off[i]] = y >= 0.f ? y : y * slope;
      }
  }
  if (threadIdx.x == 0) {
    if (rm != nullptr) { rm[c] = rmf; rv[c] = rvf; }
    if (c == 0 && nbt != nullptr) *nbt += G;
  }
}

template <int GT>
__global__ void __launch_bounds__(SB) bn_small_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                          int training, float* __restrict__ gx, float* __restrict__ ggamma,
                                                          float* __restrict__ gbeta, int B, int C, int HW, int accumulate,
                                                          const float* __restrict__ add, int Grt, int add_groups) {
  __shared__ double scratch[32];
  const int c = blockIdx.x;
  const int G = GT > 0 ? GT : Grt;
  const double n = (double)B * HW;
  double tb = 0.0, tg = 0.0;
  for (int g = 0; g < G; ++g) {
    const SmallIdx ix(B, C, HW, c, (int64_t)g * B * C * HW);
    const float r = invstd[g * C + c], mu = mean[g * C + c], a = gamma[c] * r, b = beta[c] - mu * a;
    float gyh[SPER], xh[SPER];
#pragma unroll
    for (int i = 0; i < SPER; ++i) {
      const float xv = ix.ok[i] ? x[ix.off[i]] : mu;
      const float gg = ix.ok[i] ? gz[ix.off[i]] : 0.f;
      gyh[i] = bn_y(xv, a, b) >= 0.f ? gg : gg * slope;
      xh[i] = (xv - mu) * r;
    }
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < SPER; ++i) { a0 += gyh[i]; a1 += gyh[i] * xh[i]; }
    const double sb = block_sum_d1024((double)a0, scratch), sg = block_sum_d1024((double)a1, scratch);
    tb += sb; tg += sg;
    if (gx != nullptr) {
      const float k1 = (float)(sb / n), k2 = (float)(sg / n);
      const bool use_add = add != nullptr && g < add_groups;
#pragma unroll
      for (int i = 0; i < SPER; ++i)
        if (ix.ok[i]) gx[ix.off[i]] = (training ? a * (gyh[i] - k1 - xh[i] * k2) : a * gyh[i]) + (use_add ? add[ix.off[i]] : 0.f);
    }
  }
  if (threadIdx.x == 0) {
    gbeta[c] = (float)tb + (accumulate ? gbeta[c] : 0.f);
    ggamma[c] = (float)tg + (accumulate ? ggamma[c] : 0.f);
  }
}

// ---- two groups side by side: threads [0, 512) own group 0, [512, 1024) group 1 (32 elements per thread), so the paired
// real | fake tensor of the discriminator costs ONE small-kernel latency, not two in a row.  Sums are per group (the eight
// waves of a half); thread 0 then publishes both groups' results in group order.
constexpr int G2 = 2, SEG2 = SB / G2, PER2 = SMALL_N / SEG2;

// per-group block sum: result of the caller's OWN group in every thread; all 1024 threads must call
__device__ __forceinline__ double group_sum_d2(double v, double* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  constexpr int WPG = SEG2 / 64;
  const int w0 = (wid / WPG) * WPG;
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < WPG; ++i) r += scratch[w0 + i];
  return r;
}

__device__ __forceinline__ int small2_off(int e, int HW, int C, int c) {     // offset inside the group (< 2^31: n <= 16384 per channel)
  const int b = e / HW;
  return (b * C + c) * HW + (e - b * HW);
}

__global__ void __launch_bounds__(SB) bn_small_fwd2_kernel(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ invstd,
                                                           float* __restrict__ rm, float* __restrict__ rv, int64_t* __restrict__ nbt,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                           float momentum, float eps, float* __restrict__ z, int B, int C, int HW, int rep) {
  __shared__ double scratch[32];
  __shared__ double stat[G2][2];
  const int c = blockIdx.x;
  const int g = threadIdx.x / SEG2, t = threadIdx.x - g * SEG2;
  const int64_t base = (int64_t)g * B * C * HW;
  const int n_i = B * HW;
  const float pivot = x[base + (int64_t)c * HW];
  float xv[PER2];
#pragma unroll
  for (int i = 0; i < PER2; ++i) {
    const int e = t + i * SEG2;
    xv[i] = e < n_i ? x[base + small2_off(e, HW, C, c)] : pivot;
  }
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < PER2; ++i) { const float d = xv[i] - pivot; a0 += d; a1 += d * d; }
  const double n = (double)n_i;
  const double s1 = group_sum_d2((double)a0, scratch) / n;
  const double s2 = group_sum_d2((double)a1, scratch) / n;
  const double m = (double)pivot + s1;
  double var = s2 - s1 * s1;
  if (var < 0.0) var = 0.0;
  const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
  if (t == 0) {
    mean[g * C + c] = mf;
    invstd[g * C + c] = rf;
    stat[g][0] = m; stat[g][1] = var;
  }
  const float a = gamma[c] * rf, b = beta[c] - mf * a;
#pragma unroll
  for (int i = 0; i < PER2; ++i) {
    const int e = t + i * SEG2;
    if (e < n_i) {
      const float y = bn_y(xv[i], a, b);
      z[base + small2_off(e, HW, C, c)] = y >= 0.f ? y : y * slope;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (rm != nullptr) {
      float rmf = rm[c], rvf = rv[c];
      for (int k = 0; k < G2; ++k) running_update(rmf, rvf, stat[k][0], stat[k][1], n, rep, momentum);
      rm[c] = rmf; rv[c] = rvf;
    }
    if (c == 0 && nbt != nullptr) *nbt += G2;
  }
}

__global__ void __launch_bounds__(SB) bn_small_bwd2_kernel(const float* __restrict__ gz, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                           int training, float* __restrict__ gx, float* __restrict__ ggamma,
                                                           float* __restrict__ gbeta, int B, int C, int HW, int accumulate,
                                                           const float* __restrict__ add, int add_groups) {
  __shared__ double scratch[32];
  __shared__ double tot[G2][2];
  const int c = blockIdx.x;
  const int g = threadIdx.x / SEG2, t = threadIdx.x - g * SEG2;
  const int64_t base = (int64_t)g * B * C * HW;
  const int n_i = B * HW;
  const float r = invstd[g * C + c], mu = mean[g * C + c], a = gamma[c] * r, b = beta[c] - mu * a;
  float gyh[PER2], xh[PER2];
#pragma unroll
  for (int i = 0; i < PER2; ++i) {
    const int e = t + i * SEG2;
    const bool ok = e < n_i;
    const int64_t off = base + (ok ? small2_off(e, HW, C, c) : 0);
    const float xv = ok ? x[off] : mu;
    const float gg = ok ? gz[off] : 0.f;
    gyh[i] = bn_y(xv, a, b) >= 0.f ? gg : gg * slope;
    xh[i] = (xv - mu) * r;
  }
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int i = 0; i < PER2; ++i) { a0 += gyh[i]; a1 += gyh[i] * xh[i]; }
  const double n = (double)n_i;
  const double sb = group_sum_d2((double)a0, scratch), sg = group_sum_d2((double)a1, scratch);
  if (t == 0) { tot[g][0] = sb; tot[g][1] = sg; }
  if (gx != nullptr) {
    const float k1 = (float)(sb / n), k2 = (float)(sg / n);
    const bool use_add = add != nullptr && g < add_groups;
#pragma unroll
    for (int i = 0; i < PER2; ++i) {
      const int e = t + i * SEG2;
      if (e < n_i) {
        const int64_t off = base + small2_off(e, HW, C, c);
        gx[off] = (training ? a * (gyh[i] - k1 - xh[i] * k2) : a * gyh[i]) + (use_add ? add[off] : 0.f);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    gbeta[c] = (float)(tot[0][0] + tot[1][0]) + (accumulate ? gbeta[c] : 0.f);
    ggamma[c] = (float)(tot[0][1] + tot[1][1]) + (accumulate ? ggamma[c] : 0.f);
  }
}

// (<= DSMALL_N elements per channel, 512 threads x 8: the fp64 coefficient algebra needs more than the 128 registers a
// 1024-thread workgroup leaves a lane -- as a 1024 x 16 kernel this one spilled 148 of them)
constexpr int DSMALL_N = 4096, DSB = 512, DPER = DSMALL_N / DSB;
__device__ __forceinline__ double block_sum_d512(double v, double* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < DSB / 64; ++i) r += scratch[i];
  return r;
}
__global__ void __launch_bounds__(DSB) bn_small_dbwd_kernel(const float* __restrict__ v, const float* __restrict__ vgamma,
                                                           const float* __restrict__ vbeta, const float* __restrict__ gz,
                                                           const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float slope, float* __restrict__ adj_gz,
                                                           float* __restrict__ adj_x, float* __restrict__ adj_gamma, int B, int C, int HW,
                                                           int accumulate) {
  __shared__ double scratch[32];
  const int c = blockIdx.x;
  const SmallIdxN<DPER, DSB> ix(B, C, HW, c);
  const float r = invstd[c], mu = mean[c], gr = gamma[c] * r, b = beta[c] - mu * gr;
  float vv[DPER], gyh[DPER], xh[DPER], sl[DPER];
#pragma unroll
  for (int i = 0; i < DPER; ++i) {
    const float xv = ix.ok[i] ? x[ix.off[i]] : mu;
    vv[i] = ix.ok[i] ? v[ix.off[i]] : 0.f;
    const float g = ix.ok[i] ? gz[ix.off[i]] : 0.f;
    sl[i] = bn_y(xv, gr, b) >= 0.f ? 1.f : slope;
    gyh[i] = g * sl[i];
    xh[i] = (xv - mu) * r;
  }
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < DPER; ++i) {
    acc[0] += vv[i]; acc[1] += vv[i] * xh[i]; acc[2] += gyh[i]; acc[3] += gyh[i] * xh[i]; acc[4] += vv[i] * gyh[i];
  }
  const double n = (double)B * HW;
  const double S1 = block_sum_d512((double)acc[0], scratch), S2 = block_sum_d512((double)acc[1], scratch);
  const double S3 = block_sum_d512((double)acc[2], scratch), S4 = block_sum_d512((double)acc[3], scratch);
  const double S5 = block_sum_d512((double)acc[4], scratch);
  const double rd = (double)r, g = (double)gamma[c];
  const double vg = vgamma ? (double)vgamma[c] : 0.0;
  const double A = S5 - S1 * S3 / n - S2 * S4 / n;
  const double cgd = S4 / n, cvd = S2 / n;
  const float k0 = (float)(S1 / n), k1 = (float)(S2 / n), cg = (float)cgd, cv = (float)cvd;
  const float qm = (float)(-g * rd * (cgd * S1 / n + cvd * S3 / n) + vg * S3 / n);
  const float qx = (float)(-g * rd * (cgd * S2 / n + cvd * S4 / n) + vg * S4 / n);
  const float k6 = (float)(g * A * rd * rd / n);
  const float vgf = (float)vg, vbf = vbeta ? vbeta[c] : 0.f;
  if (threadIdx.x == 0) adj_gamma[c] = (float)(rd * A) + (accumulate ? adj_gamma[c] : 0.f);
#pragma unroll
  for (int i = 0; i < DPER; ++i)
    if (ix.ok[i]) {
      const float pv = vv[i] - k0 - xh[i] * k1;
      adj_gz[ix.off[i]] = (gr * pv + vgf * xh[i] + vbf) * sl[i];
      const float q = -gr * (cg * vv[i] + cv * gyh[i]) + vgf * gyh[i];
      adj_x[ix.off[i]] = r * (q - qm - xh[i] * qx) - k6 * xh[i];
    }
}

// ------------------------------------------------------------------ data-parallel BatchNorm (SyncBN)
// Three steps per pass: local per-channel sums (fp64, [C][K] channel-major) -> all-reduce SUM by the caller
// (torch.distributed: RCCL / gloo) -> finish with the global sums.  Equal shard sizes on every rank.
//   statistics K = 3: (mean_r, mean_r^2, var_r) -- the union's mean is the mean of the local means, its biased variance
//                     the mean of the local variances plus the variance of the local means (fp64: no cancellation issue)
//   backward   K = 2: (sum ghat, sum ghat*xhat) with the GLOBAL mean / invstd
//   second bwd K = 5: S1..S5 of tg_bn_act_dbwd
__global__ void sync_stats_pack(const double* __restrict__ partial, const float* __restrict__ x, double* __restrict__ sums,
                                int B, int C, int HW, int S) {
  const int c = blockIdx.x;                       // one wave per channel
  const double n = (double)B * HW;
  const double pivot = (double)x[(int64_t)c * HW];
  const double s1 = planes::gather(partial, c, S, 2, 0) / n;
  const double s2 = planes::gather(partial, c, S, 2, 1) / n;
  if (threadIdx.x != 0) return;
  const double m = pivot + s1;
  double var = s2 - s1 * s1;
  if (var < 0.0) var = 0.0;
  sums[c * 3 + 0] = m; sums[c * 3 + 1] = m * m; sums[c * 3 + 2] = var;
}

__global__ void sync_stats_finish(const double* __restrict__ sums, int world, float* __restrict__ mean, float* __restrict__ invstd,
                                  float* __restrict__ rm, float* __restrict__ rv, int64_t* __restrict__ nbt, float momentum,
                                  float eps, double count, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt != nullptr) *nbt += 1;
  if (c >= C) return;
  const double w = (double)world;
  const double m = sums[c * 3 + 0] / w;
  double var = sums[c * 3 + 2] / w + (sums[c * 3 + 1] / w - m * m);
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rm != nullptr) {
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    rm[c] = (float)((1.0 - (double)momentum) * (double)rm[c] + (double)momentum * m);
    rv[c] = (float)((1.0 - (double)momentum) * (double)rv[c] + (double)momentum * unbiased);
  }
}

template <int K>
__global__ void sync_sums_pack(const double* __restrict__ partial, double* __restrict__ sums, int S) {
  const int c = blockIdx.x;                       // one wave per channel
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = planes::gather(partial, c, S, K, k);
    if (threadIdx.x == 0) sums[c * K + k] = v;
  }
}

// parameter gradients from the LOCAL sums (the gradient all-reduce averages them like every other parameter gradient),
// the input-gradient coefficients from the GLOBAL sums over the global element count
__global__ void sync_bwd_finish(const double* __restrict__ local, const double* __restrict__ global, double count,
                                float* __restrict__ ggamma, float* __restrict__ gbeta, float* __restrict__ coef, int C, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  gbeta[c] = (float)local[c * 2 + 0] + (accumulate ? gbeta[c] : 0.f);
  ggamma[c] = (float)local[c * 2 + 1] + (accumulate ? ggamma[c] : 0.f);
  coef[c * COEF + 0] = (float)(global[c * 2 + 0] / count);
  coef[c * COEF + 1] = (float)(global[c * 2 + 1] / count);
}

// dbwd_stage2 on the global sums; adj_gamma is this rank's 1/world share of r*A (the shares add up to the global adjoint
// under the gradient all-reduce; see DESIGN.md "SyncBN")
__global__ void sync_dbwd_finish(const double* __restrict__ global, double count, int world, const float* __restrict__ gamma,
                                 const float* __restrict__ invstd, float* __restrict__ adj_gamma, float* __restrict__ coef, int C,
                                 int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double n = count;
  const double S1 = global[c * 5 + 0], S2 = global[c * 5 + 1], S3 = global[c * 5 + 2], S4 = global[c * 5 + 3], S5 = global[c * 5 + 4];
  const double r = (double)invstd[c], g = (double)gamma[c];
  const double A = S5 - S1 * S3 / n - S2 * S4 / n;
  const double cg = S4 / n, cv = S2 / n;
  const double qm = -g * r * (cg * S1 / n + cv * S3 / n);
  const double qx = -g * r * (cg * S2 / n + cv * S4 / n);
  adj_gamma[c] = (float)(r * A / (double)world) + (accumulate ? adj_gamma[c] : 0.f);
  float* k = coef + c * COEF;
  k[0] = (float)(S1 / n); k[1] = (float)(S2 / n); k[2] = (float)cg; k[3] = (float)cv;
  k[4] = (float)qm; k[5] = (float)qx; k[6] = (float)(g * A * r * r / n);
}

}  // namespace

extern "C" {

size_t tg_bn_workspace(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return 0;
  const int S = planes::splits(B, C, HW);
  return (size_t)C * S * MAXK * sizeof(double) + (size_t)C * COEF * sizeof(float);
}

static int train_stats_groups(const float* x, float* mean, float* invstd, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float momentum, float eps, float* workspace, int G, int B, int C, int HW,
                              int replicate, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(G); TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW); TG_CHECK_POS(replicate);
  if ((running_mean == nullptr) != (running_var == nullptr)) return TG_EINVAL;
  hipStream_t st = tg_stream(stream);
  Parts p = split_ws(workspace, B, G * C, HW);
  RedStats red{x, C, HW, 0.f};
  planes::launch_reduce(red, p.partial, B, C, HW, st, tg_aligned16(x), G);
  stats_stage2<<<C, 64, 0, st>>>(p.partial, x, mean, invstd, running_mean, running_var, num_batches_tracked, momentum, eps, B, C, HW,
                                            planes::splits(B, G * C, HW), replicate, G);
  return tg_launch_status();
}

int tg_bn_train_stats(const float* x, float* mean, float* invstd, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float momentum, float eps, float* workspace, int B, int C, int HW,
                      int replicate, void* stream) {
  return train_stats_groups(x, mean, invstd, running_mean, running_var, num_batches_tracked, momentum, eps, workspace, 1, B, C, HW,
                            replicate, stream);
}

static int act_fwd_groups(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                          float* z, int G, int B, int C, int HW, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(gamma); TG_CHECK_PTR(beta); TG_CHECK_PTR(z);
  TG_CHECK_POS(G); TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  FwdBody body{x, z, mean, invstd, gamma, beta, slope};
  planes::launch_map(body, G * B, C, HW, tg_stream(stream), tg_aligned16(x) && tg_aligned16(z), B);
  return tg_launch_status();
}

int tg_bn_train_fwd_groups(const float* x, float* mean, float* invstd, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, const float* gamma, const float* beta, float slope, float momentum,
                           float eps, float* z, float* workspace, int groups, int B, int C, int HW, int replicate, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(gamma); TG_CHECK_PTR(beta); TG_CHECK_PTR(z);
  TG_CHECK_PTR(workspace);
  TG_CHECK_POS(groups); TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW); TG_CHECK_POS(replicate);
  if ((running_mean == nullptr) != (running_var == nullptr)) return TG_EINVAL;
  const int G = groups;
  if (small_case(B, C, HW)) {
    if (G == 2)
      bn_small_fwd2_kernel<<<C, SB, 0, tg_stream(stream)>>>(x, mean, invstd, running_mean, running_var, num_batches_tracked,
                                                               gamma, beta, slope, momentum, eps, z, B, C, HW, replicate);
    else if (G == 1)
      bn_small_fwd_kernel<1><<<C, SB, 0, tg_stream(stream)>>>(x, mean, invstd, running_mean, running_var, num_batches_tracked,
                                                                 gamma, beta, slope, momentum, eps, z, B, C, HW, replicate, G);
    else
      bn_small_fwd_kernel<0><<<C, SB, 0, tg_stream(stream)>>>(x, mean, invstd, running_mean, running_var, num_batches_tracked,
                                                                 gamma, beta, slope, momentum, eps, z, B, C, HW, replicate, G);
    return tg_launch_status();
  }
  if (planes::big(HW) && tg_aligned16(x) && tg_aligned16(z)) {
    // two launches: the per-block partial sums, then the apply pass, whose blocks finish the reduction themselves
    hipStream_t st = tg_stream(stream);
    Parts p = split_ws(workspace, B, G * C, HW);
    RedStats red{x, C, HW, 0.f};
    planes::launch_reduce(red, p.partial, B, C, HW, st, true, G);
    FwdStatsBody body{x, z, p.partial, mean, invstd, running_mean, running_var, num_batches_tracked, gamma, beta,
                      slope, momentum, eps, B, HW, planes::splits(B, G * C, HW), replicate, C, G, 0.f, 0.f};
    planes::launch_map_begin(body, B, C, HW, st, G);
    return tg_launch_status();
  }
  if (int rc = train_stats_groups(x, mean, invstd, running_mean, running_var, num_batches_tracked, momentum, eps, workspace, G, B, C,
                                  HW, replicate, stream))
    return rc;
  return act_fwd_groups(x, mean, invstd, gamma, beta, slope, z, G, B, C, HW, stream);
}

int tg_bn_train_fwd(const float* x, float* mean, float* invstd, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, const float* gamma, const float* beta, float slope, float momentum, float eps,
                    float* z, float* workspace, int B, int C, int HW, int replicate, void* stream) {
  return tg_bn_train_fwd_groups(x, mean, invstd, running_mean, running_var, num_batches_tracked, gamma, beta, slope, momentum, eps,
                                z, workspace, 1, B, C, HW, replicate, stream);
}

int tg_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* invstd, float eps, int C,
                     void* stream) {
  TG_CHECK_PTR(running_mean); TG_CHECK_PTR(running_var); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd);
  TG_CHECK_POS(C);
  eval_stats_kernel<<<chan_grid(C), 64, 0, tg_stream(stream)>>>(running_mean, running_var, mean, invstd, eps, C);
  return tg_launch_status();
}

int tg_bn_act_fwd(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                  float* z, int B, int C, int HW, void* stream) {
  return act_fwd_groups(x, mean, invstd, gamma, beta, slope, z, 1, B, C, HW, stream);
}

int tg_bn_act_bwd_groups(const float* gz, const float* x, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, float slope, int training, float* gx, float* ggamma, float* gbeta, float* workspace,
                         int groups, int B, int C, int HW, int accumulate, const float* gx_add, int add_groups, void* stream) {
  TG_CHECK_PTR(gz); TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(gamma); TG_CHECK_PTR(beta);
  TG_CHECK_PTR(ggamma); TG_CHECK_PTR(gbeta); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(groups); TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  if (gx_add != nullptr && (gx == nullptr || add_groups < 1 || add_groups > groups)) return TG_EINVAL;
  const int G = groups;
  hipStream_t st = tg_stream(stream);
  if (small_case(B, C, HW)) {
    if (G == 2)
      bn_small_bwd2_kernel<<<C, SB, 0, st>>>(gz, x, mean, invstd, gamma, beta, slope, training, gx, ggamma, gbeta, B, C, HW,
                                                accumulate, gx_add, add_groups);
    else if (G == 1)
      bn_small_bwd_kernel<1><<<C, SB, 0, st>>>(gz, x, mean, invstd, gamma, beta, slope, training, gx, ggamma, gbeta, B, C, HW,
                                                  accumulate, gx_add, G, add_groups);
    else
      bn_small_bwd_kernel<0><<<C, SB, 0, st>>>(gz, x, mean, invstd, gamma, beta, slope, training, gx, ggamma, gbeta, B, C, HW,
                                                  accumulate, gx_add, G, add_groups);
    return tg_launch_status();
  }
  Parts p = split_ws(workspace, B, G * C, HW);
  const int S = planes::splits(B, G * C, HW);
  const int add_vc_end = add_groups * C;
  RedBwd red{gz, x, mean, invstd, gamma, beta, slope, 0.f, 0.f, 0.f, 0.f};
  planes::launch_reduce(red, p.partial, B, C, HW, st, tg_aligned16(x) && tg_aligned16(gz), G);
  if (gx != nullptr && planes::big(HW) && tg_aligned16(x) && tg_aligned16(gz) && tg_aligned16(gx) && (!gx_add || tg_aligned16(gx_add))) {
    BwdSumsBody body{gz, x, gx, p.partial, mean, invstd, gamma, beta, ggamma, gbeta, slope, training, accumulate,
                     B, HW, S, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, gx_add, C, G, add_vc_end, false};
    planes::launch_map_begin(body, B, C, HW, st, G);
    return tg_launch_status();
  }
  bwd_stage2<<<C, 64, 0, st>>>(p.partial, ggamma, gbeta, p.coef, B, C, HW, S, accumulate, G);
  if (gx != nullptr) {
    BwdBody body{gz, x, gx, mean, invstd, gamma, beta, p.coef, slope, training, gx_add, add_vc_end};
    planes::launch_map(body, G * B, C, HW, st, tg_aligned16(x) && tg_aligned16(gz) && tg_aligned16(gx) && (!gx_add || tg_aligned16(gx_add)), B);
  }
  return tg_launch_status();
}

int tg_bn_act_bwd(const float* gz, const float* x, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, float slope, int training, float* gx, float* ggamma, float* gbeta, float* workspace,
                  int B, int C, int HW, int accumulate, const float* gx_add, void* stream) {
  return tg_bn_act_bwd_groups(gz, x, mean, invstd, gamma, beta, slope, training, gx, ggamma, gbeta, workspace, 1, B, C, HW,
                              accumulate, gx_add, 1, stream);
}

int tg_bn_act_dbwd(const float* v, const float* vgamma, const float* vbeta, const float* gz, const float* x,
                   const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                   float* adj_gz, float* adj_x, float* adj_gamma, float* workspace, int B, int C, int HW, int accumulate,
                   void* stream) {
  TG_CHECK_PTR(v); TG_CHECK_PTR(gz); TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(gamma);
  TG_CHECK_PTR(beta); TG_CHECK_PTR(adj_gz); TG_CHECK_PTR(adj_x); TG_CHECK_PTR(adj_gamma); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  hipStream_t st = tg_stream(stream);
  if (small_case(B, C, HW) && (int64_t)B * HW <= DSMALL_N) {
    bn_small_dbwd_kernel<<<C, DSB, 0, st>>>(v, vgamma, vbeta, gz, x, mean, invstd, gamma, beta, slope, adj_gz, adj_x, adj_gamma,
                                              B, C, HW, accumulate);
    return tg_launch_status();
  }
  Parts p = split_ws(workspace, B, C, HW);
  RedDbwd red{v, gz, x, mean, invstd, gamma, beta, slope, 0.f, 0.f, 0.f, 0.f};
  const bool al = tg_aligned16(x) && tg_aligned16(gz) && tg_aligned16(v) && tg_aligned16(adj_gz) && tg_aligned16(adj_x);
  planes::launch_reduce(red, p.partial, B, C, HW, st, al);
  if (planes::big(HW) && al) {                     // the per-channel coefficients are finished inside the apply pass
    DbwdSumsBody body{v, gz, x, adj_gz, adj_x, p.partial, mean, invstd, gamma, beta, vgamma, vbeta, adj_gamma, slope, accumulate, B, HW,
                      planes::splits(B, C, HW), {}};
    planes::launch_map_begin(body, B, C, HW, st);
    return tg_launch_status();
  }
  dbwd_stage2<<<C, 64, 0, st>>>(p.partial, gamma, invstd, vgamma, adj_gamma, p.coef, B, C, HW,
                                           planes::splits(B, C, HW), accumulate);
  DbwdBody body{v, gz, x, adj_gz, adj_x, mean, invstd, gamma, beta, vgamma, vbeta, p.coef, slope};
  planes::launch_map(body, B, C, HW, st, al);
  return tg_launch_status();
}

int tg_bn_sync_stats_local(const float* x, double* sums, float* workspace, int B, int C, int HW, void* stream) {
  TG_CHECK_PTR(x); TG_CHECK_PTR(sums); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  hipStream_t st = tg_stream(stream);
  Parts p = split_ws(workspace, B, C, HW);
  RedStats red{x, C, HW, 0.f};
  planes::launch_reduce(red, p.partial, B, C, HW, st, tg_aligned16(x));
  sync_stats_pack<<<C, 64, 0, st>>>(p.partial, x, sums, B, C, HW, planes::splits(B, C, HW));
  return tg_launch_status();
}

int tg_bn_sync_stats_finish(const double* sums, int world, float* mean, float* invstd, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float momentum, float eps, int64_t count_global, int replicate, int C,
                            void* stream) {
  TG_CHECK_PTR(sums); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd);
  TG_CHECK_POS(world); TG_CHECK_POS(C); TG_CHECK_POS(replicate);
  if (count_global <= 0 || (running_mean == nullptr) != (running_var == nullptr)) return TG_EINVAL;
  sync_stats_finish<<<chan_grid(C), 64, 0, tg_stream(stream)>>>(sums, world, mean, invstd, running_mean, running_var,
                                                                   num_batches_tracked, momentum, eps,
                                                                   (double)count_global * replicate, C);
  return tg_launch_status();
}

int tg_bn_sync_bwd_local(const float* gz, const float* x, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, float slope, double* sums, float* workspace, int B, int C, int HW, void* stream) {
  TG_CHECK_PTR(gz); TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(gamma); TG_CHECK_PTR(beta);
  TG_CHECK_PTR(sums); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  hipStream_t st = tg_stream(stream);
  Parts p = split_ws(workspace, B, C, HW);
  RedBwd red{gz, x, mean, invstd, gamma, beta, slope, 0.f, 0.f, 0.f, 0.f};
  planes::launch_reduce(red, p.partial, B, C, HW, st, tg_aligned16(x) && tg_aligned16(gz));
  sync_sums_pack<2><<<C, 64, 0, st>>>(p.partial, sums, planes::splits(B, C, HW));
  return tg_launch_status();
}

int tg_bn_sync_bwd_finish(const float* gz, const float* x, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, float slope, const double* local_sums, const double* global_sums,
                          int64_t count_global, float* gx, float* ggamma, float* gbeta, float* workspace, int B, int C, int HW,
                          int accumulate, const float* gx_add, void* stream) {
  TG_CHECK_PTR(gz); TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(gamma); TG_CHECK_PTR(beta);
  TG_CHECK_PTR(local_sums); TG_CHECK_PTR(global_sums); TG_CHECK_PTR(ggamma); TG_CHECK_PTR(gbeta); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  if (count_global <= 0) return TG_EINVAL;
  hipStream_t st = tg_stream(stream);
  Parts p = split_ws(workspace, B, C, HW);
  sync_bwd_finish<<<chan_grid(C), 64, 0, st>>>(local_sums, global_sums, (double)count_global, ggamma, gbeta, p.coef, C, accumulate);
  if (gx != nullptr) {
    BwdBody body{gz, x, gx, mean, invstd, gamma, beta, p.coef, slope, 1, gx_add, C};
    planes::launch_map(body, B, C, HW, st, tg_aligned16(x) && tg_aligned16(gz) && tg_aligned16(gx));
  }
  return tg_launch_status();
}

int tg_bn_sync_dbwd_local(const float* v, const float* gz, const float* x, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, float slope, double* sums, float* workspace, int B, int C,
                          int HW, void* stream) {
  TG_CHECK_PTR(v); TG_CHECK_PTR(gz); TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(gamma);
  TG_CHECK_PTR(beta); TG_CHECK_PTR(sums); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW);
  hipStream_t st = tg_stream(stream);
  Parts p = split_ws(workspace, B, C, HW);
  RedDbwd red{v, gz, x, mean, invstd, gamma, beta, slope, 0.f, 0.f, 0.f, 0.f};
  planes::launch_reduce(red, p.partial, B, C, HW, st, tg_aligned16(x) && tg_aligned16(gz) && tg_aligned16(v));
  sync_sums_pack<5><<<C, 64, 0, st>>>(p.partial, sums, planes::splits(B, C, HW));
  return tg_launch_status();
}

int tg_bn_sync_dbwd_finish(const float* v, const float* gz, const float* x, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, float slope, const double* global_sums, int64_t count_global,
                           int world, float* adj_gz, float* adj_x, float* adj_gamma, float* workspace, int B, int C, int HW,
                           int accumulate, void* stream) {
  TG_CHECK_PTR(v); TG_CHECK_PTR(gz); TG_CHECK_PTR(x); TG_CHECK_PTR(mean); TG_CHECK_PTR(invstd); TG_CHECK_PTR(gamma);
  TG_CHECK_PTR(beta); TG_CHECK_PTR(global_sums); TG_CHECK_PTR(adj_gz); TG_CHECK_PTR(adj_x); TG_CHECK_PTR(adj_gamma);
  TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(C); TG_CHECK_POS(HW); TG_CHECK_POS(world);
  if (count_global <= 0) return TG_EINVAL;
  hipStream_t st = tg_stream(stream);
  Parts p = split_ws(workspace, B, C, HW);
  sync_dbwd_finish<<<chan_grid(C), 64, 0, st>>>(global_sums, (double)count_global, world, gamma, invstd, adj_gamma, p.coef, C, accumulate);
  DbwdBody body{v, gz, x, adj_gz, adj_x, mean, invstd, gamma, beta, nullptr, nullptr, p.coef, slope};
  const bool al = tg_aligned16(x) && tg_aligned16(gz) && tg_aligned16(v) && tg_aligned16(adj_gz) && tg_aligned16(adj_x);
  planes::launch_map(body, B, C, HW, st, al);
  return tg_launch_status();
}

}  // extern "C"
