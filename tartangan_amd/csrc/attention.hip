// Fused SelfAttention2d core (reference models/blocks/attention.py:27-34):
//     beta = softmax_m(theta^T phi)   (N x M, M = N/4)        o = g beta^T
// without ever materialising beta (16 MiB per image at 128:3 in the generator).
//
// Layouts are the reference's channel-major views: theta (B, D, N), phi (B, D, M), g (B, DV, M),
// o (B, DV, N); lse (B, N) = log sum_m exp(score) is saved for the backward.
//
// Head dims are tiny (D = C/8 in {1..16}, DV = C/2 in {4..64}) but every contraction here is still a matrix product,
// and v_mfma_f32_16x16x4_f32 has exactly the K = 4 that D = 4 needs.  One wave owns a 16-row tile of one side
// (queries for forward / dtheta, keys for dphi / dg) and walks the other side in 16-row tiles; operands are loaded
// straight from global memory (L2-resident: K and V of an image are a few KB) into the MFMA register layouts, scores
// and probabilities never leave registers.  With lane l = (c = l & 15, g = l >> 4):
//     A operand: A[row c][k g]        B operand: B[k g][col c]        result reg r: D[row 4g + r][col c]
// A 16x16 result tile is reused as the B operand of the NEXT product without any data movement: result register r of
// lane group g holds rows 4g + r, so taking "register r" as k-step r just enumerates the contraction index in the
// order (r, g) -> row 4g + r, and the other operand is loaded in that same order (4 consecutive rows = one float4).
// (An earlier one-lane-per-query VALU version spent 8x its issue time waiting on LDS broadcasts.)
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AT = 256;          // threads per workgroup: 4 waves, 16 tile rows each

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); }

// Operand loads.  `full` (wave-uniform): every column this tile touches is in range and rows are 16-byte aligned, so
// the loads are unconditional (rows past the head dim are clamped and zeroed by a select: no exec-mask branches in the
// inner loops); otherwise every element is guarded.
template <bool FULL>
__device__ __forceinline__ float ld1(const float* __restrict__ p, int row, int rows, int64_t stride, int col, int cols) {
  if (FULL) {
    const float v = p[(int64_t)min(row, rows - 1) * stride + col];
    return row < rows ? v : 0.f;
  }
  return (row < rows && col < cols) ? p[(int64_t)row * stride + col] : 0.f;
}
// 4 consecutive columns of one row (columns >= cols read as 0)
template <bool FULL>
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p, int row, int rows, int64_t stride, int col, int cols) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (FULL) {
    const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)min(row, rows - 1) * stride + col);
    if (row < rows) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    return v;
  }
  if (row < rows) {
    const float* q = p + (int64_t)row * stride + col;
    if (col < cols) v[0] = q[0];
    if (col + 1 < cols) v[1] = q[1];
    if (col + 2 < cols) v[2] = q[2];
    if (col + 3 < cols) v[3] = q[3];
  }
  return v;
}
__device__ __forceinline__ float group_max(float v) {      // over the 4 lane groups (same c)
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

template <int D, int DV> struct AttnDims {
  static constexpr int KS = (D + 3) / 4;       // k-steps of the score product
  static constexpr int VT = (DV + 15) / 16;    // 16-row tiles of the value dimension
  static constexpr int VS = (DV + 3) / 4;      // k-steps of a contraction over dv
};
constexpr int JT = 4;                          // 16-row tiles of the streamed side per loop iteration (forward, dtheta)
constexpr int JK2 = 2;                         // ... for dphi / dg, whose per-tile operand set is larger

// A wave owns RT 16-row tiles of its side (RT = 4 when that still leaves >= 4 waves per SIMD, else 1) and reuses every
// streamed operand it loaded for all of them: with RT = 1 the 64x64-pixel attention map of the generator re-read K / V
// (or Q / dO) from L2 once per 16 rows -- 1.4 GB per call -- and ran at a third of the matrix-core rate.

// ---------------------------------------------------------------------------------------------------- forward
// One step = 64 keys: per owned query tile 4 score tiles, 16 exp2 and the PV products alternating between two accumulators
// (halves the dependent-MFMA chain).  Scores are kept in log2 units (theta is scaled by log2 e on load) so that the softmax
// weights are bare v_exp_f32, and RELATIVE to a per-query reference value ref (the score tiles start from -ref in the MFMA's C
// operand: no subtraction per element).  softmax is invariant under the choice of ref, so ref need not be the running
// maximum: it starts as the maximum over the first 16 keys and is raised -- accumulators and partial sum rescaled -- only
// when some score of the wave exceeds it by more than SLACK (weights stay below 2^SLACK: no overflow; ref is always a score
// of the row, so the largest weight of a row is >= 1: no underflow of the sum).  On the generator's 64x64 map that is a
// handful of rescales per row instead of one per 64 keys, each of which dragged the 32 accumulator registers through the
// vector ALU and two cross-lane maxima through the LDS crossbar.
constexpr float SLACK = 10.f;

template <int D, int DV, int RT, bool FULL>
__device__ __forceinline__ void fwd_step(const float* __restrict__ pb, const float* __restrict__ gb, const float (&qB)[RT][(D + 3) / 4],
                                         f32x4 (&acc)[RT][RT == 4 ? 1 : 2][(DV + 15) / 16], float (&ref)[RT], float (&lpart)[RT], int k0, int M,
                                         int c, int g) {
  using A = AttnDims<D, DV>;
  constexpr int NA = RT == 4 ? 1 : 2;          // accumulators per query tile: a wave that owns 4 tiles has chains enough
  float kA[JT][A::KS];
  f32x4 vA[JT][A::VT];
#pragma unroll
  for (int j = 0; j < JT; ++j) {
#pragma unroll
    for (int s = 0; s < A::KS; ++s) kA[j][s] = ld1<FULL>(pb, 4 * s + g, D, M, k0 + 16 * j + c, M);
#pragma unroll
    for (int t = 0; t < A::VT; ++t) vA[j][t] = ld4<FULL>(gb, 16 * t + c, DV, M, k0 + 16 * j + 4 * g, M);
  }
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    f32x4 st[JT];
    const float nr = -ref[q];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      st[j] = f32x4{nr, nr, nr, nr};
#pragma unroll
      for (int s = 0; s < A::KS; ++s) st[j] = mfma16(kA[j][s], qB[q][s], st[j]);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!FULL && k0 + 16 * j + 4 * g + r >= M) st[j][r] = -INFINITY;
        tmax = fmaxf(tmax, st[j][r]);
      }
    if (__builtin_amdgcn_ballot_w64(tmax > SLACK) != 0) {        // wave-uniform, rare after the first steps
      const float up = fmaxf(group_max(tmax), 0.f);              // the same for the 4 lane groups of a query
      const float resc = fast_exp2(-up);
      ref[q] += up;
      lpart[q] *= resc;
#pragma unroll
      for (int t = 0; t < A::VT; ++t)
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[q][a][t] *= resc;
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[j][r] -= up;
    }
    float psum = 0.f;
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[j][r] = fast_exp2(st[j][r]);
        psum += st[j][r];
      }
    lpart[q] += psum;
#pragma unroll
    for (int t = 0; t < A::VT; ++t)
#pragma unroll
      for (int j = 0; j < JT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][j % NA][t] = mfma16(vA[j][t][r], st[j][r], acc[q][j % NA][t]);
  }
}

// registers per lane (vector + accumulator file) that leave room for W waves per SIMD
template <int D, int DV, int RT> struct AttnOcc {
  // the generator's 64x64 map at batch 64 is 4096 waves of 64 queries: 4 per SIMD, resident at once only within 128 registers
  static constexpr int WAVES = (RT == 4 && D <= 4) ? 4 : 1;
};

template <int D, int DV, int RT>
__global__ void __launch_bounds__(AT) __attribute__((amdgpu_waves_per_eu(AttnOcc<D, DV, RT>::WAVES)))
attn_fwd_kernel(const float* __restrict__ theta, const float* __restrict__ phi, const float* __restrict__ g_, float* __restrict__ o,
                float* __restrict__ lse, int N, int M, int vecM) {
  using A = AttnDims<D, DV>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * 16 * RT;
  if (q0 >= N) return;                                      // whole wave; no barriers in this kernel
  const float* th = theta + (int64_t)b * D * N;
  const float* pb = phi + (int64_t)b * D * M;
  const float* gb = g_ + (int64_t)b * DV * M;
  constexpr int NA = RT == 4 ? 1 : 2;
  float qB[RT][A::KS], ref[RT], lpart[RT];
  f32x4 acc[RT][NA][A::VT];
  float k0A[A::KS];                                         // the first 16 keys: where the reference values come from
#pragma unroll
  for (int s = 0; s < A::KS; ++s) k0A[s] = ld1<false>(pb, 4 * s + g, D, M, c, M);
#pragma unroll
  for (int q = 0; q < RT; ++q) {
#pragma unroll
    for (int s = 0; s < A::KS; ++s) qB[q][s] = ld1<false>(th, 4 * s + g, D, N, q0 + 16 * q + c, N) * LOG2E;
#pragma unroll
    for (int t = 0; t < A::VT; ++t)
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[q][a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < A::KS; ++s) s0 = mfma16(k0A[s], qB[q][s], s0);
    float m0 = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r < M) m0 = fmaxf(m0, s0[r]);
    ref[q] = group_max(m0);
    lpart[q] = 0.f;
  }
  int k0 = 0;
  if (vecM)
    for (; k0 + 16 * JT <= M; k0 += 16 * JT) fwd_step<D, DV, RT, true>(pb, gb, qB, acc, ref, lpart, k0, M, c, g);
  for (; k0 < M; k0 += 16 * JT) fwd_step<D, DV, RT, false>(pb, gb, qB, acc, ref, lpart, k0, M, c, g);
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    const float l = group_sum(lpart[q]);
    const float inv = 1.f / l;
    const int n = q0 + 16 * q + c;
    if (n < N) {
      float* ob = o + (int64_t)b * DV * N + n;
#pragma unroll
      for (int t = 0; t < A::VT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int dv = 16 * t + 4 * g + r;
          if (dv < DV) ob[(int64_t)dv * N] = (acc[q][0][t][r] + (NA == 2 ? acc[q][NA - 1][t][r] : 0.f)) * inv;
        }
      if (g == 0) lse[(int64_t)b * N + n] = (ref[q] + __log2f(l)) * LN2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- dtheta
// query-owned: dtheta[d][n] = sum_k ds(n,k) phi[d][k],  ds = p (dp - delta), p = exp(s - lse), dp = go_n . g_k
template <int D, int DV, int RT, bool FULL>
__device__ __forceinline__ void bwd_q_step(const float* __restrict__ pb, const float* __restrict__ gb, const float (&qB)[RT][(D + 3) / 4],
                                           const float (&doB)[RT][(DV + 3) / 4], const float (&L2)[RT], const float (&dl)[RT],
                                           f32x4 (&dq)[RT][RT == 4 ? 1 : 2], int k0, int M, int c, int g) {
  using A = AttnDims<D, DV>;
  constexpr int NA = RT == 4 ? 1 : 2;
  float kA[JT][A::KS], vK[JT][A::VS];
  f32x4 kT[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) {
#pragma unroll
    for (int s = 0; s < A::KS; ++s) kA[j][s] = ld1<FULL>(pb, 4 * s + g, D, M, k0 + 16 * j + c, M);
#pragma unroll
    for (int u = 0; u < A::VS; ++u) vK[j][u] = ld1<FULL>(gb, 4 * u + g, DV, M, k0 + 16 * j + c, M);  // V[key c][dv 4u+g]
    kT[j] = ld4<FULL>(pb, c, D, M, k0 + 16 * j + 4 * g, M);                             // K^T[d c][keys k0+16j+4g+(0..3)]
  }
#pragma unroll
  for (int q = 0; q < RT; ++q)
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < A::KS; ++s) st = mfma16(kA[j][s], qB[q][s], st);
#pragma unroll
      for (int u = 0; u < A::VS; ++u) dp = mfma16(vK[j][u], doB[q][u], dp);              // dP^T[key][query]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (!FULL && k0 + 16 * j + 4 * g + r >= M) ? 0.f : fast_exp2(st[r] - L2[q]);
        dq[q][j % NA] = mfma16(kT[j][r], p * (dp[r] - dl[q]), dq[q][j % NA]);             // dtheta^T[d 4g+r'][query c]
      }
    }
}

template <int D, int DV, int RT>
__global__ void __launch_bounds__(AT) __attribute__((amdgpu_waves_per_eu(AttnOcc<D, DV, RT>::WAVES))) attn_bwd_q_kernel(const float* __restrict__ go, const float* __restrict__ theta,
                                                        const float* __restrict__ phi, const float* __restrict__ g_,
                                                        const float* __restrict__ o, const float* __restrict__ lse,
                                                        float* __restrict__ delta /*out: sum_v go o per query, for the key-owned kernel*/,
                                                        float* __restrict__ dtheta, int N, int M, int vecM) {
  using A = AttnDims<D, DV>;
  static_assert(D <= 16, "one 16-row tile of dtheta");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * 16 * RT;
  if (q0 >= N) return;
  const float* th = theta + (int64_t)b * D * N;
  const float* pb = phi + (int64_t)b * D * M;
  const float* gb = g_ + (int64_t)b * DV * M;
  const float* gob = go + (int64_t)b * DV * N;
  const float* ob = o + (int64_t)b * DV * N;
  float qB[RT][A::KS], doB[RT][A::VS], L2[RT], dl[RT];
  constexpr int NA = RT == 4 ? 1 : 2;
  f32x4 dq[RT][NA];
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    const int n = q0 + 16 * q + c;
#pragma unroll
    for (int s = 0; s < A::KS; ++s) qB[q][s] = ld1<false>(th, 4 * s + g, D, N, n, N) * LOG2E;
#pragma unroll
    for (int u = 0; u < A::VS; ++u) doB[q][u] = ld1<false>(gob, 4 * u + g, DV, N, n, N);      // dO^T[dv 4u+g][query c]
    L2[q] = n < N ? lse[(int64_t)b * N + n] * LOG2E : INFINITY;                               // dead query: p = 0
    // delta[n] = sum_v go[v][n] o[v][n]: this lane's dv rows, then the four lane groups of the query (no launch of its own)
    float dsum = 0.f;
#pragma unroll
    for (int u = 0; u < A::VS; ++u) dsum = fmaf(doB[q][u], ld1<false>(ob, 4 * u + g, DV, N, n, N), dsum);
    dl[q] = group_sum(dsum);
    if (g == 0 && n < N) delta[(int64_t)b * N + n] = dl[q];
#pragma unroll
    for (int a = 0; a < NA; ++a) dq[q][a] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int k0 = 0;
  if (vecM)
    for (; k0 + 16 * JT <= M; k0 += 16 * JT) bwd_q_step<D, DV, RT, true>(pb, gb, qB, doB, L2, dl, dq, k0, M, c, g);
  for (; k0 < M; k0 += 16 * JT) bwd_q_step<D, DV, RT, false>(pb, gb, qB, doB, L2, dl, dq, k0, M, c, g);
#pragma unroll
  for (int q = 0; q < RT; ++q) {
    const int n = q0 + 16 * q + c;
    if (n < N) {
      float* dt = dtheta + (int64_t)b * D * N + n;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < D) dt[(int64_t)(4 * g + r) * N] = dq[q][0][r] + (NA == 2 ? dq[q][NA - 1][r] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------- dphi, dg
// key-owned: dphi[d][k] = sum_n ds(n,k) theta[d][n];  dg[v][k] = sum_n p(n,k) go[v][n]
template <int D, int DV, int RT, bool FULL>
__device__ __forceinline__ void bwd_k_step(const float* __restrict__ th, const float* __restrict__ gob, const float* __restrict__ lb,
                                           const float* __restrict__ db, const float (&kB)[RT][(D + 3) / 4],
                                           const float (&vB)[RT][(DV + 3) / 4], f32x4 (&dk)[RT], f32x4 (&dv_)[RT][(DV + 15) / 16],
                                           int n0, int N, int c, int g) {
  using A = AttnDims<D, DV>;
  constexpr int JK = RT == 4 ? 1 : JK2;          // a wave that owns 4 key tiles streams one query tile at a time (registers)
  float qA[JK][A::KS], goA[JK][A::VS];
  f32x4 L4[JK], d4[JK], qT[JK], goT[JK][A::VT];
#pragma unroll
  for (int j = 0; j < JK; ++j) {
    const int nj = n0 + 16 * j;
#pragma unroll
    for (int s = 0; s < A::KS; ++s) qA[j][s] = ld1<FULL>(th, 4 * s + g, D, N, nj + c, N);        // Q[query c][d 4s+g]
#pragma unroll
    for (int u = 0; u < A::VS; ++u) goA[j][u] = ld1<FULL>(gob, 4 * u + g, DV, N, nj + c, N);    // dO[query c][dv 4u+g]
    L4[j] = ld4<FULL>(lb, 0, 1, 0, nj + 4 * g, N) * LOG2E;
    d4[j] = ld4<FULL>(db, 0, 1, 0, nj + 4 * g, N);
    qT[j] = ld4<FULL>(th, c, D, N, nj + 4 * g, N);                                      // Q^T[d c][queries nj+4g+(0..3)]
#pragma unroll
    for (int t = 0; t < A::VT; ++t) goT[j][t] = ld4<FULL>(gob, 16 * t + c, DV, N, nj + 4 * g, N);   // dO^T[dv 16t+c][queries]
  }
#pragma unroll
  for (int k = 0; k < RT; ++k)
#pragma unroll
    for (int j = 0; j < JK; ++j) {
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < A::KS; ++s) sc = mfma16(qA[j][s], kB[k][s], sc);              // S[query][key c]   (log2 units)
#pragma unroll
      for (int u = 0; u < A::VS; ++u) dp = mfma16(goA[j][u], vB[k][u], dp);             // dP[query][key c]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (!FULL && n0 + 16 * j + 4 * g + r >= N) ? 0.f : fast_exp2(sc[r] - L4[j][r]);
        dk[k] = mfma16(qT[j][r], p * (dp[r] - d4[j][r]), dk[k]);                        // dphi^T[d 4g+r'][key c]
#pragma unroll
        for (int t = 0; t < A::VT; ++t) dv_[k][t] = mfma16(goT[j][t][r], p, dv_[k][t]); // dg^T[dv 16t+4g+r'][key c]
      }
    }
}

template <int D, int DV, int RT>
__global__ void __launch_bounds__(AT) __attribute__((amdgpu_waves_per_eu(AttnOcc<D, DV, RT>::WAVES))) attn_bwd_k_kernel(const float* __restrict__ go, const float* __restrict__ theta,
                                                        const float* __restrict__ phi, const float* __restrict__ g_,
                                                        const float* __restrict__ lse, const float* __restrict__ delta,
                                                        float* __restrict__ dphi, float* __restrict__ dg, int N, int M, int QS,
                                                        int B, int vecN) {
  // blockIdx.z = query slice: the slice's sums go to partial buffers [QS][B][D or DV][M] (QS > 1) that
  // attn_reduce_k_kernel adds in a fixed order; QS == 1 writes dphi / dg directly
  using A = AttnDims<D, DV>;
  static_assert(D <= 16, "one 16-row tile of dphi");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int k0 = (blockIdx.x * 4 + wave) * 16 * RT;
  if (k0 >= M) return;
  const float* th = theta + (int64_t)b * D * N;
  const float* pb = phi + (int64_t)b * D * M;
  const float* gb = g_ + (int64_t)b * DV * M;
  const float* gob = go + (int64_t)b * DV * N;
  const float* lb = lse + (int64_t)b * N;
  const float* db = delta + (int64_t)b * N;
  float kB[RT][A::KS], vB[RT][A::VS];
  f32x4 dk[RT], dv_[RT][A::VT];
#pragma unroll
  for (int k = 0; k < RT; ++k) {
#pragma unroll
    for (int s = 0; s < A::KS; ++s) kB[k][s] = ld1<false>(pb, 4 * s + g, D, M, k0 + 16 * k + c, M) * LOG2E;   // K^T[d 4s+g][key c]
#pragma unroll
    for (int u = 0; u < A::VS; ++u) vB[k][u] = ld1<false>(gb, 4 * u + g, DV, M, k0 + 16 * k + c, M);          // V^T[dv 4u+g][key c]
    dk[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < A::VT; ++t) dv_[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int JK = RT == 4 ? 1 : JK2;
  const int slice = blockIdx.z;
  const int per = ((N + QS - 1) / QS + 16 * JK2 - 1) / (16 * JK2) * (16 * JK2);
  const int n_begin = slice * per, n_end = min(N, n_begin + per);
  int n0 = n_begin;
  if (vecN)
    for (; n0 + 16 * JK <= n_end; n0 += 16 * JK) bwd_k_step<D, DV, RT, true>(th, gob, lb, db, kB, vB, dk, dv_, n0, N, c, g);
  for (; n0 < n_end; n0 += 16 * JK) bwd_k_step<D, DV, RT, false>(th, gob, lb, db, kB, vB, dk, dv_, n0, N, c, g);
#pragma unroll
  for (int k = 0; k < RT; ++k) {
    const int key = k0 + 16 * k + c;
    if (key < M) {
      float* dp_ = dphi + ((int64_t)slice * B + b) * D * M + key;
      float* dg_ = dg + ((int64_t)slice * B + b) * DV * M + key;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < D) dp_[(int64_t)(4 * g + r) * M] = dk[k][r];
#pragma unroll
      for (int t = 0; t < A::VT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * t + 4 * g + r < DV) dg_[(int64_t)(16 * t + 4 * g + r) * M] = dv_[k][t][r];
    }
  }
}

// out[i] = sum_s part[s][i]  (fixed order), for the two partial buffers of a backward pass (d phi, d g) in one launch
__global__ void __launch_bounds__(AT) attn_reduce_k_kernel(const float* __restrict__ part1, float* __restrict__ out1, int64_t n1,
                                                           const float* __restrict__ part2, float* __restrict__ out2, int64_t n2, int QS) {
  for (int64_t i = blockIdx.x * (int64_t)AT + threadIdx.x; i < n1 + n2; i += gridDim.x * (int64_t)AT) {
    const bool first = i < n1;
    const float* part = first ? part1 : part2;
    const int64_t n = first ? n1 : n2, e = first ? i : i - n1;
    float acc = 0.f;
    for (int sidx = 0; sidx < QS; ++sidx) acc += part[sidx * n + e];
    (first ? out1 : out2)[e] = acc;
  }
}

// rows per wave (16 * RT) on the owning side: 4 tiles when that still gives every SIMD ~4 waves
static inline int attn_rt(int B, int rows) { return ((int64_t)B * ((rows + 63) / 64) >= 4096) ? 4 : 1; }
// query slices for the key-owned kernel: enough waves to fill the chip
static inline int attn_qsplit(int B, int N, int M) {
  const int64_t base1 = (int64_t)((M + 15) / 16) * B, base4 = (int64_t)((M + 63) / 64) * B;   // waves at QS = 1 for RT = 1 / 4
  const int64_t base = base4 >= 1024 ? base4 : base1;
  int qs = (int)((4096 + base - 1) / base);
  const int maxqs = (N + 16 * JK2 - 1) / (16 * JK2);
  if (qs > maxqs) qs = maxqs;
  if (qs > 16) qs = 16;
  if (qs < 1) qs = 1;
  return qs;
}
static inline int attn_rt_keys(int B, int M) { return ((int64_t)((M + 63) / 64) * B >= 1024) ? 4 : 1; }
static inline int vec_rows(const void* p, int cols) { return (cols % 4 == 0) && tg_aligned16(p); }

template <int D, int DV>
int launch_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse, int B, int N, int M, hipStream_t st) {
  const int vm = vec_rows(g, M) && vec_rows(phi, M);
  if (attn_rt(B, N) == 4) {
    dim3 grid((N + 255) / 256, B);
    attn_fwd_kernel<D, DV, 4><<<grid, AT, 0, st>>>(theta, phi, g, o, lse, N, M, vm);
  } else {
    dim3 grid((N + 63) / 64, B);
    attn_fwd_kernel<D, DV, 1><<<grid, AT, 0, st>>>(theta, phi, g, o, lse, N, M, vm);
  }
  return tg_launch_status();
}
template <int D, int DV>
int launch_bwd(const float* go, const float* theta, const float* phi, const float* g, const float* o, const float* lse,
               float* dtheta, float* dphi, float* dg, float* ws, int B, int N, int M, hipStream_t st) {
  const int QS = attn_qsplit(B, N, M);
  float* delta = ws;
  float* pphi = ws + (((int64_t)B * N + 3) / 4) * 4;
  float* pg = pphi + (int64_t)QS * B * D * M;
  const int vecM = vec_rows(g, M) && vec_rows(phi, M);
  const int vecN = vec_rows(theta, N) && vec_rows(go, N) && vec_rows(lse, N) && vec_rows(delta, N);
  if (attn_rt(B, N) == 4) {
    dim3 gq((N + 255) / 256, B);
    attn_bwd_q_kernel<D, DV, 4><<<gq, AT, 0, st>>>(go, theta, phi, g, o, lse, delta, dtheta, N, M, vecM);
  } else {
    dim3 gq((N + 63) / 64, B);
    attn_bwd_q_kernel<D, DV, 1><<<gq, AT, 0, st>>>(go, theta, phi, g, o, lse, delta, dtheta, N, M, vecM);
  }
  float* ophi = QS == 1 ? dphi : pphi;
  float* og = QS == 1 ? dg : pg;
  if (attn_rt_keys(B, M) == 4) {
    dim3 gk((M + 255) / 256, B, QS);
    attn_bwd_k_kernel<D, DV, 4><<<gk, AT, 0, st>>>(go, theta, phi, g, lse, delta, ophi, og, N, M, QS, B, vecN);
  } else {
    dim3 gk((M + 63) / 64, B, QS);
    attn_bwd_k_kernel<D, DV, 1><<<gk, AT, 0, st>>>(go, theta, phi, g, lse, delta, ophi, og, N, M, QS, B, vecN);
  }
  if (QS > 1) {
    const int64_t n1 = (int64_t)B * D * M, n2 = (int64_t)B * DV * M;
    attn_reduce_k_kernel<<<tg_ew_grid(n1 + n2, AT), AT, 0, st>>>(pphi, dphi, n1, pg, dg, n2, QS);
  }
  return tg_launch_status();
}

// Wavefront sum on the DPP lanes of the vector ALU (no LDS crossbar round trips as with __shfl_xor): quad swaps, the two
// row mirrors, then the row totals chained over rows 1..3; lane 63 holds the total, handed back wave-uniform.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define TG_DPP_ADD(CTRL, ROWS) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false))
  TG_DPP_ADD(0xB1, 0xf);      // quad_perm [1,0,3,2]
  TG_DPP_ADD(0x4E, 0xf);      // quad_perm [2,3,0,1]
  TG_DPP_ADD(0x141, 0xf);     // row_half_mirror
  TG_DPP_ADD(0x140, 0xf);     // row_mirror: every lane of a 16-lane row has the row's sum
  TG_DPP_ADD(0x142, 0xa);     // row_bcast15 into rows 1, 3
  TG_DPP_ADD(0x143, 0xc);     // row_bcast31 into rows 2, 3
#undef TG_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// =========================================================================== second order (R1 penalty)
// The derivative of the first-order backward above: with (a, b, c) the adjoints of (dtheta, dphi, dg), the adjoints of
// (go, theta, phi, g).  Per row n of the (N x M) maps (all of them functions of the row's and the columns' operands only):
//   S = theta^T phi, P = exp(S - lse), gP = go^T g, U = a^T phi + theta^T b, V = go^T c,
//   delta = sum P gP, eps = sum P U, zeta = sum P V + sum P U gP - 2 delta eps,
//   gS = P (gP - delta), dgP = P (U - eps), dS = P (V + U (gP - delta) - gP eps - zeta),
//   d theta[:, n] = phi dS + b gS,   d go[:, n] = g dgP + c P          (sums over the row)
//   d phi += theta[:, n] dS + a[:, n] gS,   d g += go[:, n] dgP         (sums over rows)
// Nothing N x M ever exists in memory: one wave owns 64 rows of one image; a lane owns CPL columns (m = lane + 64 j) and
// keeps THEIR operands (phi, b, g, c: 2 (D + DV) CPL registers) and THEIR d phi / d g sums ((D + DV) CPL registers)
// for its whole life; the row's operands are wave-uniform (scalar loads); row sums are wavefront reductions.  ~110 FMAs
// per map element on the vector unit (the maps have tiny inner dimensions: MFMA would idle on operand shuffles here).
template <int D, int DV, int CPL>
__global__ void __launch_bounds__(64)
attn_dbwd_kernel(const float* __restrict__ go, const float* __restrict__ theta, const float* __restrict__ phi, const float* __restrict__ g_,
                 const float* __restrict__ lse, const float* __restrict__ a, const float* __restrict__ b_, const float* __restrict__ c_,
                 float* __restrict__ d_go, float* __restrict__ d_theta, float* __restrict__ pphi, float* __restrict__ pg,
                 int N, int M, int B) {
  const int lane = threadIdx.x;
  const int w = blockIdx.x, bi = blockIdx.y;
  const int n0 = w * 64;
  const float* th = theta + (int64_t)bi * D * N;
  const float* ab = a + (int64_t)bi * D * N;
  const float* gob = go + (int64_t)bi * DV * N;
  const float* lb = lse + (int64_t)bi * N;
  float cphi[D][CPL], cb[D][CPL], cg[DV][CPL], cc[DV][CPL], sphi[D][CPL], sg[DV][CPL];
  bool valid[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int m = lane + 64 * j;
    valid[j] = m < M;
    const int mm = valid[j] ? m : 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      cphi[k][j] = valid[j] ? phi[((int64_t)bi * D + k) * M + mm] : 0.f;
      cb[k][j] = valid[j] ? b_[((int64_t)bi * D + k) * M + mm] : 0.f;
      sphi[k][j] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < DV; ++q) {
      cg[q][j] = valid[j] ? g_[((int64_t)bi * DV + q) * M + mm] : 0.f;
      cc[q][j] = valid[j] ? c_[((int64_t)bi * DV + q) * M + mm] : 0.f;
      sg[q][j] = 0.f;
    }
  }
  float o_th[D], o_go[DV];               // lane i keeps the row outputs of row n0 + i
#pragma unroll
  for (int k = 0; k < D; ++k) o_th[k] = 0.f;
#pragma unroll
  for (int q = 0; q < DV; ++q) o_go[q] = 0.f;

  const int nrows = min(64, N - n0);
  for (int i = 0; i < nrows; ++i) {
    const int n = n0 + i;
    float rt[D], ra[D], rg[DV];          // wave-uniform row operands
#pragma unroll
    for (int k = 0; k < D; ++k) { rt[k] = th[k * N + n]; ra[k] = ab[k * N + n]; }
#pragma unroll
    for (int q = 0; q < DV; ++q) rg[q] = gob[q * N + n];
    const float l = lb[n];
    float P[CPL], GP[CPL], U[CPL], V[CPL];
    float d = 0.f, e = 0.f, pv = 0.f, pug = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      float sv = 0.f, uv = 0.f, gv = 0.f, vv = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        sv = fmaf(rt[k], cphi[k][j], sv);
        uv = fmaf(ra[k], cphi[k][j], uv);
        uv = fmaf(rt[k], cb[k][j], uv);
      }
#pragma unroll
      for (int q = 0; q < DV; ++q) {
        gv = fmaf(rg[q], cg[q][j], gv);
        vv = fmaf(rg[q], cc[q][j], vv);
      }
      const float p = valid[j] ? __expf(sv - l) : 0.f;
      P[j] = p; GP[j] = gv; U[j] = uv; V[j] = vv;
      d = fmaf(p, gv, d); e = fmaf(p, uv, e); pv = fmaf(p, vv, pv); pug = fmaf(p * uv, gv, pug);
    }
    d = wave_sum_dpp(d); e = wave_sum_dpp(e); pv = wave_sum_dpp(pv); pug = wave_sum_dpp(pug);
    const float z = pv + pug - 2.f * d * e;
    float GS[CPL], DS[CPL], DG[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const float dp = V[j] + U[j] * (GP[j] - d) - GP[j] * e;
      GS[j] = P[j] * (GP[j] - d);
      DS[j] = P[j] * (dp - z);
      DG[j] = P[j] * (U[j] - e);
    }
    // row outputs
#pragma unroll
    for (int k = 0; k < D; ++k) {
      float r = 0.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) { r = fmaf(DS[j], cphi[k][j], r); r = fmaf(GS[j], cb[k][j], r); }
      r = wave_sum_dpp(r);
      o_th[k] = (lane == i) ? r : o_th[k];
    }
#pragma unroll
    for (int q = 0; q < DV; ++q) {
      float r = 0.f;
#pragma unroll
      for (int j = 0; j < CPL; ++j) { r = fmaf(DG[j], cg[q][j], r); r = fmaf(P[j], cc[q][j], r); }
      r = wave_sum_dpp(r);
      o_go[q] = (lane == i) ? r : o_go[q];
    }
    // column sums
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
#pragma unroll
      for (int k = 0; k < D; ++k) { sphi[k][j] = fmaf(DS[j], rt[k], sphi[k][j]); sphi[k][j] = fmaf(GS[j], ra[k], sphi[k][j]); }
#pragma unroll
      for (int q = 0; q < DV; ++q) sg[q][j] = fmaf(DG[j], rg[q], sg[q][j]);
    }
  }
  if (lane < nrows) {
#pragma unroll
    for (int k = 0; k < D; ++k) d_theta[((int64_t)bi * D + k) * N + n0 + lane] = o_th[k];
#pragma unroll
    for (int q = 0; q < DV; ++q) d_go[((int64_t)bi * DV + q) * N + n0 + lane] = o_go[q];
  }
  // partial column sums [row slice w][B][D or DV][M], added in a fixed order by attn_reduce_k_kernel
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int m = lane + 64 * j;
    if (m < M) {
#pragma unroll
      for (int k = 0; k < D; ++k) pphi[(((int64_t)w * B + bi) * D + k) * M + m] = sphi[k][j];
#pragma unroll
      for (int q = 0; q < DV; ++q) pg[(((int64_t)w * B + bi) * DV + q) * M + m] = sg[q][j];
    }
  }
}

// columns per lane the register file takes: 3 (D + DV) CPL operand / sum registers next to the row's own
static inline int dbwd_cpl(int D, int DV, int M) {
  const int cpl = (M + 63) / 64;
  return (cpl <= 4 && 3 * (D + DV) * cpl <= 256) ? cpl : 0;
}

template <int D, int DV, int CPL>
int launch_dbwd(const float* go, const float* theta, const float* phi, const float* g, const float* lse, const float* a,
                const float* b, const float* c, float* d_go, float* d_theta, float* d_phi, float* d_g, float* ws,
                int B, int N, int M, hipStream_t st) {
  const int S = (N + 63) / 64;
  float* pphi = ws;
  float* pg = ws + (int64_t)S * B * D * M;
  attn_dbwd_kernel<D, DV, CPL><<<dim3(S, B), 64, 0, st>>>(go, theta, phi, g, lse, a, b, c, d_go, d_theta, pphi, pg, N, M, B);
  const int64_t n1 = (int64_t)B * D * M, n2 = (int64_t)B * DV * M;
  attn_reduce_k_kernel<<<tg_ew_grid(n1 + n2, AT), AT, 0, st>>>(pphi, d_phi, n1, pg, d_g, n2, S);
  return tg_launch_status();
}
template <int D, int DV>
int launch_dbwd_cpl(int cpl, const float* go, const float* theta, const float* phi, const float* g, const float* lse, const float* a,
                    const float* b, const float* c, float* d_go, float* d_theta, float* d_phi, float* d_g, float* ws,
                    int B, int N, int M, hipStream_t st) {
  switch (cpl) {
    case 1: return launch_dbwd<D, DV, 1>(go, theta, phi, g, lse, a, b, c, d_go, d_theta, d_phi, d_g, ws, B, N, M, st);
    case 2: if constexpr (3 * (D + DV) * 2 <= 256) return launch_dbwd<D, DV, 2>(go, theta, phi, g, lse, a, b, c, d_go, d_theta, d_phi, d_g, ws, B, N, M, st); break;
    case 3: if constexpr (3 * (D + DV) * 3 <= 256) return launch_dbwd<D, DV, 3>(go, theta, phi, g, lse, a, b, c, d_go, d_theta, d_phi, d_g, ws, B, N, M, st); break;
    case 4: if constexpr (3 * (D + DV) * 4 <= 256) return launch_dbwd<D, DV, 4>(go, theta, phi, g, lse, a, b, c, d_go, d_theta, d_phi, d_g, ws, B, N, M, st); break;
  }
  return TG_EUNSUPPORTED;
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
  return ((((size_t)B * N + 3) / 4) * 4 + (size_t)QS * B * (D + DV) * M) * sizeof(float);
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

int tg_attn_dbwd_supported(int D, int DV, int M) {
  return tg_attn_supported(D, DV) && M > 0 && dbwd_cpl(D, DV, M) > 0;
}

size_t tg_attn_dbwd_workspace(int B, int D, int DV, int N, int M) {
  if (B <= 0 || D <= 0 || DV <= 0 || N <= 0 || M <= 0) return 0;
  return (size_t)((N + 63) / 64) * B * (D + DV) * M * sizeof(float);
}

int tg_attn_dbwd(const float* go, const float* theta, const float* phi, const float* g, const float* lse, const float* a,
                 const float* b, const float* c, float* d_go, float* d_theta, float* d_phi, float* d_g, float* workspace,
                 int B, int D, int DV, int N, int M, void* stream) {
  TG_CHECK_PTR(go); TG_CHECK_PTR(theta); TG_CHECK_PTR(phi); TG_CHECK_PTR(g); TG_CHECK_PTR(lse);
  TG_CHECK_PTR(a); TG_CHECK_PTR(b); TG_CHECK_PTR(c);
  TG_CHECK_PTR(d_go); TG_CHECK_PTR(d_theta); TG_CHECK_PTR(d_phi); TG_CHECK_PTR(d_g); TG_CHECK_PTR(workspace);
  TG_CHECK_POS(B); TG_CHECK_POS(N); TG_CHECK_POS(M);
  if (!tg_attn_dbwd_supported(D, DV, M)) return TG_EUNSUPPORTED;
  if (B > 65535) return TG_EINVAL;
  const int cpl = dbwd_cpl(D, DV, M);
  hipStream_t st = tg_stream(stream);
#define TG_DBWD(D_, DV_) \
  if (D == D_ && DV == DV_) return launch_dbwd_cpl<D_, DV_>(cpl, go, theta, phi, g, lse, a, b, c, d_go, d_theta, d_phi, d_g, workspace, B, N, M, st)
  TG_DBWD(1, 4); TG_DBWD(2, 8); TG_DBWD(4, 16); TG_DBWD(8, 32); TG_DBWD(16, 64);
#undef TG_DBWD
  return TG_EUNSUPPORTED;
}

}  // extern "C"
