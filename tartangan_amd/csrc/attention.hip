// Fused SelfAttention2d core (reference models/blocks/attention.py:27-34):
//     beta = softmax_m(theta^T phi)   (N x M, M = N/4)        o = g beta^T
// without ever materialising beta (16 MiB per image at 128:3 in the generator).
//
// Head dims here are tiny (D = C/8 in {1..16}, DV = C/2 in {4..64}): the contraction is far too
// thin for MFMA tiles, the work is exp/VALU-bound.  So: one lane per query
// (forward, dtheta) or per key (dphi, dg), the other side's rows are staged in LDS and read as
// wave-uniform broadcasts (every lane reads the same address: conflict-free), scores / softmax
// weights / accumulators live in registers, online softmax over key tiles.
//
// Layouts are the reference's channel-major views: theta (B, D, N), phi (B, D, M), g (B, DV, M),
// o (B, DV, N); lse (B, N) = log sum_m exp(score) is saved for the backward.
#include "common.h"

namespace {

constexpr int AT = 256;          // threads per workgroup = queries (or keys) per workgroup
constexpr int KT = 64;           // rows of the other side staged per LDS tile
constexpr int SUB = 16;          // scores handled per online-softmax rescale

template <int D, int DV>
__global__ void __launch_bounds__(AT) attn_fwd_kernel(const float* __restrict__ theta, const float* __restrict__ phi,
                                                      const float* __restrict__ g, float* __restrict__ o, float* __restrict__ lse,
                                                      int N, int M) {
  __shared__ float ph[KT][D];      // [key][d]   : one broadcast read fetches a key's whole phi row
  __shared__ float gl[KT][DV];     // [key][dv]
  const int b = blockIdx.y;
  const int n = blockIdx.x * AT + threadIdx.x;
  const bool live = n < N;
  const float* th = theta + (int64_t)b * D * N;
  float q[D];
#pragma unroll
  for (int d = 0; d < D; ++d) q[d] = live ? th[(int64_t)d * N + n] : 0.f;
  float acc[DV];
#pragma unroll
  for (int v = 0; v < DV; ++v) acc[v] = 0.f;
  float mx = -INFINITY, l = 0.f;
  const float* pb = phi + (int64_t)b * D * M;
  const float* gb = g + (int64_t)b * DV * M;

  for (int k0 = 0; k0 < M; k0 += KT) {
    __syncthreads();
    for (int e = threadIdx.x; e < KT * D; e += AT) {
      const int k = e % KT, d = e / KT;                     // coalesced along keys
      ph[k][d] = (k0 + k < M) ? pb[(int64_t)d * M + k0 + k] : 0.f;
    }
    for (int e = threadIdx.x; e < KT * DV; e += AT) {
      const int k = e % KT, v = e / KT;
      gl[k][v] = (k0 + k < M) ? gb[(int64_t)v * M + k0 + k] : 0.f;
    }
    __syncthreads();
    const int kn = min(KT, M - k0);
    for (int ks = 0; ks < kn; ks += SUB) {
      float s[SUB];
      float tmax = -INFINITY;
#pragma unroll
      for (int u = 0; u < SUB; ++u) {
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) sc = fmaf(q[d], ph[ks + u][d], sc);
        sc = (ks + u < kn) ? sc : -INFINITY;
        s[u] = sc;
        tmax = fmaxf(tmax, sc);
      }
      const float mnew = fmaxf(mx, tmax);
      const float resc = __expf(mx - mnew);                 // exp(-inf) = 0 on the first tile
      l *= resc;
#pragma unroll
      for (int v = 0; v < DV; ++v) acc[v] *= resc;
#pragma unroll
      for (int u = 0; u < SUB; ++u) {
        const float p = __expf(s[u] - mnew);
        l += p;
#pragma unroll
        for (int v = 0; v < DV; ++v) acc[v] = fmaf(p, gl[ks + u][v], acc[v]);
      }
      mx = mnew;
    }
  }
  if (live) {
    const float inv = 1.f / l;
    float* ob = o + (int64_t)b * DV * N;
#pragma unroll
    for (int v = 0; v < DV; ++v) ob[(int64_t)v * N + n] = acc[v] * inv;
    lse[(int64_t)b * N + n] = mx + __logf(l);
  }
}

// delta[n] = sum_v go[v][n] * o[v][n]
__global__ void __launch_bounds__(AT) attn_delta_kernel(const float* __restrict__ go, const float* __restrict__ o,
                                                        float* __restrict__ delta, int DV, int N) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * AT + threadIdx.x;
  if (n >= N) return;
  const float* a = go + (int64_t)b * DV * N;
  const float* c = o + (int64_t)b * DV * N;
  float acc = 0.f;
  for (int v = 0; v < DV; ++v) acc = fmaf(a[(int64_t)v * N + n], c[(int64_t)v * N + n], acc);
  delta[(int64_t)b * N + n] = acc;
}

// query-owned: dtheta[d][n] = sum_k ds(n,k) phi[d][k],  ds = p (dp - delta), p = exp(s - lse), dp = go_n . g_k
template <int D, int DV>
__global__ void __launch_bounds__(AT) attn_bwd_q_kernel(const float* __restrict__ go, const float* __restrict__ theta,
                                                        const float* __restrict__ phi, const float* __restrict__ g,
                                                        const float* __restrict__ lse, const float* __restrict__ delta,
                                                        float* __restrict__ dtheta, int N, int M) {
  __shared__ float ph[KT][D];
  __shared__ float gl[KT][DV];
  const int b = blockIdx.y;
  const int n = blockIdx.x * AT + threadIdx.x;
  const bool live = n < N;
  const float* th = theta + (int64_t)b * D * N;
  const float* gob = go + (int64_t)b * DV * N;
  float q[D], dq[D], dout[DV];
#pragma unroll
  for (int d = 0; d < D; ++d) { q[d] = live ? th[(int64_t)d * N + n] : 0.f; dq[d] = 0.f; }
#pragma unroll
  for (int v = 0; v < DV; ++v) dout[v] = live ? gob[(int64_t)v * N + n] : 0.f;
  const float L = live ? lse[(int64_t)b * N + n] : 0.f;
  const float dl = live ? delta[(int64_t)b * N + n] : 0.f;
  const float* pb = phi + (int64_t)b * D * M;
  const float* gb = g + (int64_t)b * DV * M;
  for (int k0 = 0; k0 < M; k0 += KT) {
    __syncthreads();
    for (int e = threadIdx.x; e < KT * D; e += AT) {
      const int k = e % KT, d = e / KT;
      ph[k][d] = (k0 + k < M) ? pb[(int64_t)d * M + k0 + k] : 0.f;
    }
    for (int e = threadIdx.x; e < KT * DV; e += AT) {
      const int k = e % KT, v = e / KT;
      gl[k][v] = (k0 + k < M) ? gb[(int64_t)v * M + k0 + k] : 0.f;
    }
    __syncthreads();
    const int kn = min(KT, M - k0);
    for (int k = 0; k < kn; ++k) {
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) sc = fmaf(q[d], ph[k][d], sc);
#pragma unroll
      for (int v = 0; v < DV; ++v) dp = fmaf(dout[v], gl[k][v], dp);
      const float ds = __expf(sc - L) * (dp - dl);
#pragma unroll
      for (int d = 0; d < D; ++d) dq[d] = fmaf(ds, ph[k][d], dq[d]);
    }
  }
  if (live) {
    float* dt = dtheta + (int64_t)b * D * N;
#pragma unroll
    for (int d = 0; d < D; ++d) dt[(int64_t)d * N + n] = dq[d];
  }
}

// key-owned: dphi[d][k] = sum_n ds(n,k) theta[d][n];  dg[v][k] = sum_n p(n,k) go[v][n]
template <int D, int DV>
__global__ void __launch_bounds__(AT) attn_bwd_k_kernel(const float* __restrict__ go, const float* __restrict__ theta,
                                                        const float* __restrict__ phi, const float* __restrict__ g,
                                                        const float* __restrict__ lse, const float* __restrict__ delta,
                                                        float* __restrict__ dphi, float* __restrict__ dg, int N, int M, int QS,
                                                        int B) {
  // blockIdx.z = query slice: the slice's sums go to partial buffers [QS][B][D or DV][M] (QS > 1) that
  // attn_reduce_k_kernel adds in a fixed order; QS == 1 writes dphi / dg directly
  __shared__ float tq[KT][D];
  __shared__ float dq[KT][DV];
  __shared__ float ls[KT];
  __shared__ float de[KT];
  const int b = blockIdx.y;
  const int k = blockIdx.x * AT + threadIdx.x;
  const bool live = k < M;
  const float* pb = phi + (int64_t)b * D * M;
  const float* gb = g + (int64_t)b * DV * M;
  float kp[D], kg[DV], dkp[D], dkg[DV];
#pragma unroll
  for (int d = 0; d < D; ++d) { kp[d] = live ? pb[(int64_t)d * M + k] : 0.f; dkp[d] = 0.f; }
#pragma unroll
  for (int v = 0; v < DV; ++v) { kg[v] = live ? gb[(int64_t)v * M + k] : 0.f; dkg[v] = 0.f; }
  const float* th = theta + (int64_t)b * D * N;
  const float* gob = go + (int64_t)b * DV * N;
  const int slice = blockIdx.z;
  const int per = ((N + QS - 1) / QS + KT - 1) / KT * KT;
  const int n_begin = slice * per, n_end = min(N, n_begin + per);
  for (int n0 = n_begin; n0 < n_end; n0 += KT) {
    __syncthreads();
    for (int e = threadIdx.x; e < KT * D; e += AT) {
      const int q = e % KT, d = e / KT;
      tq[q][d] = (n0 + q < N) ? th[(int64_t)d * N + n0 + q] : 0.f;
    }
    for (int e = threadIdx.x; e < KT * DV; e += AT) {
      const int q = e % KT, v = e / KT;
      dq[q][v] = (n0 + q < N) ? gob[(int64_t)v * N + n0 + q] : 0.f;
    }
    for (int e = threadIdx.x; e < KT; e += AT) {
      ls[e] = (n0 + e < N) ? lse[(int64_t)b * N + n0 + e] : INFINITY;     // p = exp(s - inf) = 0 for padding queries
      de[e] = (n0 + e < N) ? delta[(int64_t)b * N + n0 + e] : 0.f;
    }
    __syncthreads();
    for (int q = 0; q < KT; ++q) {
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) sc = fmaf(tq[q][d], kp[d], sc);
#pragma unroll
      for (int v = 0; v < DV; ++v) dp = fmaf(dq[q][v], kg[v], dp);
      const float p = __expf(sc - ls[q]);
      const float ds = p * (dp - de[q]);
#pragma unroll
      for (int d = 0; d < D; ++d) dkp[d] = fmaf(ds, tq[q][d], dkp[d]);
#pragma unroll
      for (int v = 0; v < DV; ++v) dkg[v] = fmaf(p, dq[q][v], dkg[v]);
    }
  }
  if (live) {
    float* dp_ = dphi + ((int64_t)slice * B + b) * D * M;
    float* dg_ = dg + ((int64_t)slice * B + b) * DV * M;
#pragma unroll
    for (int d = 0; d < D; ++d) dp_[(int64_t)d * M + k] = dkp[d];
#pragma unroll
    for (int v = 0; v < DV; ++v) dg_[(int64_t)v * M + k] = dkg[v];
  }
}

// out[i] = sum_s part[s][i]  (fixed order)
__global__ void __launch_bounds__(AT) attn_reduce_k_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int QS) {
  for (int64_t i = blockIdx.x * (int64_t)AT + threadIdx.x; i < n; i += gridDim.x * (int64_t)AT) {
    float acc = 0.f;
    for (int sidx = 0; sidx < QS; ++sidx) acc += part[sidx * n + i];
    out[i] = acc;
  }
}

// query slices for the key-owned kernel: enough workgroups to fill the chip
static inline int attn_qsplit(int B, int N, int M) {
  const int64_t base = (int64_t)((M + AT - 1) / AT) * B;
  int qs = (int)((1024 + base - 1) / base);
  const int maxqs = (N + KT - 1) / KT;
  if (qs > maxqs) qs = maxqs;
  if (qs > 16) qs = 16;
  if (qs < 1) qs = 1;
  return qs;
}

template <int D, int DV>
int launch_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse, int B, int N, int M, hipStream_t st) {
  dim3 grid((N + AT - 1) / AT, B);
  attn_fwd_kernel<D, DV><<<grid, AT, 0, st>>>(theta, phi, g, o, lse, N, M);
  return tg_launch_status();
}
template <int D, int DV>
int launch_bwd(const float* go, const float* theta, const float* phi, const float* g, const float* o, const float* lse,
               float* dtheta, float* dphi, float* dg, float* ws, int B, int N, int M, hipStream_t st) {
  const int QS = attn_qsplit(B, N, M);
  float* delta = ws;
  float* pphi = ws + (int64_t)B * N;
  float* pg = pphi + (int64_t)QS * B * D * M;
  dim3 gq((N + AT - 1) / AT, B), gk((M + AT - 1) / AT, B, QS);
  attn_delta_kernel<<<gq, AT, 0, st>>>(go, o, delta, DV, N);
  attn_bwd_q_kernel<D, DV><<<gq, AT, 0, st>>>(go, theta, phi, g, lse, delta, dtheta, N, M);
  if (QS == 1) {
    attn_bwd_k_kernel<D, DV><<<gk, AT, 0, st>>>(go, theta, phi, g, lse, delta, dphi, dg, N, M, 1, B);
  } else {
    attn_bwd_k_kernel<D, DV><<<gk, AT, 0, st>>>(go, theta, phi, g, lse, delta, pphi, pg, N, M, QS, B);
    const int64_t n1 = (int64_t)B * D * M, n2 = (int64_t)B * DV * M;
    attn_reduce_k_kernel<<<tg_ew_grid(n1, AT), AT, 0, st>>>(pphi, dphi, n1, QS);
    attn_reduce_k_kernel<<<tg_ew_grid(n2, AT), AT, 0, st>>>(pg, dg, n2, QS);
  }
  return tg_launch_status();
}

}  // namespace

#define TG_ATTN_DISPATCH(CALL)                   \
  if (D == 4 && DV == 16) return CALL(4, 16);    \
  if (D == 16 && DV == 64) return CALL(16, 64);  \
  if (D == 8 && DV == 32) return CALL(8, 32);    \
  if (D == 2 && DV == 8) return CALL(2, 8);      \
  if (D == 1 && DV == 4) return CALL(1, 4);      \
  return TG_EUNSUPPORTED;

extern "C" {

int tg_attn_supported(int D, int DV) {
  return (D == 4 && DV == 16) || (D == 16 && DV == 64) || (D == 8 && DV == 32) || (D == 2 && DV == 8) || (D == 1 && DV == 4);
}

size_t tg_attn_bwd_workspace(int B, int D, int DV, int N, int M) {
  if (B <= 0 || D <= 0 || DV <= 0 || N <= 0 || M <= 0) return 0;
  const int QS = attn_qsplit(B, N, M);
  return ((size_t)B * N + (size_t)QS * B * (D + DV) * M) * sizeof(float);
}

int tg_attn_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse, int B, int D, int DV, int N, int M,
                void* stream) {
  TG_CHECK_PTR(theta); TG_CHECK_PTR(phi); TG_CHECK_PTR(g); TG_CHECK_PTR(o); TG_CHECK_PTR(lse);
  TG_CHECK_POS(B); TG_CHECK_POS(N); TG_CHECK_POS(M);
  if (B > 65535) return TG_EUNSUPPORTED;
  hipStream_t st = tg_stream(stream);
#define CALL_FWD(d, v) launch_fwd<d, v>(theta, phi, g, o, lse, B, N, M, st)
  TG_ATTN_DISPATCH(CALL_FWD)
#undef CALL_FWD
}

int tg_attn_bwd(const float* go, const float* theta, const float* phi, const float* g, const float* o, const float* lse,
                float* dtheta, float* dphi, float* dg, float* workspace, int B, int D, int DV, int N, int M, void* stream) {
  TG_CHECK_PTR(go); TG_CHECK_PTR(theta); TG_CHECK_PTR(phi); TG_CHECK_PTR(g); TG_CHECK_PTR(o); TG_CHECK_PTR(lse);
  TG_CHECK_PTR(dtheta); TG_CHECK_PTR(dphi); TG_CHECK_PTR(dg); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(N); TG_CHECK_POS(M);
  if (B > 65535) return TG_EUNSUPPORTED;
  hipStream_t st = tg_stream(stream);
#define CALL_BWD(d, v) launch_bwd<d, v>(go, theta, phi, g, o, lse, dtheta, dphi, dg, workspace, B, N, M, st)
  TG_ATTN_DISPATCH(CALL_BWD)
#undef CALL_BWD
}

}  // extern "C"
