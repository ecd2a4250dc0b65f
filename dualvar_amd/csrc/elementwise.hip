// Streaming (HBM-bound) kernels of the DualVar hot path on gfx950: ingest, BatchNorm statistics /
// apply / backward, MaxPool3d, spatial mean, self-gating scale, optimizer.  NDHWC, 16-byte vector
// accesses along the channel axis, fp32 math, wavefront(64)-shuffle + LDS reductions.
#include "common.hpp"

namespace {

constexpr int kThreads = 256;
static inline int grid_for(int64_t work_items, int max_blocks = 4096) {
  int64_t b = (work_items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

// ------------------------------------------------------------------ ingest
template <typename T>
__global__ void ingest_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int C, int T_, int H, int W,
                              int64_t sxn, int ldy, const float* mean3, const float* istd3,
                              const int* perm, int n_seg) {
  const int64_t total = (int64_t)N * T_ * H * W;
  const int64_t plane = (int64_t)H * W;
  const int seg = n_seg > 0 ? T_ / n_seg : T_;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t hw = i % plane;
    int64_t nt = i / plane;
    int t = (int)(nt % T_);
    int n = (int)(nt / T_);
    int ts = t;
    if (perm) ts = perm[n * n_seg + t / seg] * seg + t % seg;
    const float* src = x + (int64_t)n * sxn + (int64_t)ts * plane + hw;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C && c < 4; ++c) {
      float a = src[(int64_t)c * T_ * plane];
      if (mean3) a = (a - mean3[c]) * istd3[c];
      v[c] = a;
    }
    T* dst = y + i * ldy;
    if (sizeof(T) == 4) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst) = o;
    } else {
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      *reinterpret_cast<bf16x4*>(dst) = o;
    }
  }
}

// ------------------------------------------------------------------ block reduce helper
__device__ __forceinline__ float block_sum(float v, float* sh /*>= 4 floats*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}

// ------------------------------------------------------------------ BN forward statistics
// one block per channel: partials [tiles][2][C] -> local [2C+1]
__global__ void bn_reduce_stats_kernel(const float* __restrict__ part, int n_tiles, int tile_rows, int64_t M, int C,
                                       float* __restrict__ out) {
  __shared__ float sh[8];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) s += part[((size_t)i * 2) * C + c];
  const float S = block_sum(s, sh);
  const float mean = S / (float)M;
  float m2 = 0.f;
  for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) {
    int64_t left = M - (int64_t)i * tile_rows;
    float n_i = (float)(left < tile_rows ? left : tile_rows);
    float d = part[((size_t)i * 2) * C + c] / n_i - mean;
    m2 += part[((size_t)i * 2 + 1) * C + c] + n_i * d * d;
  }
  const float M2 = block_sum(m2, sh + 4);
  if (threadIdx.x == 0) {
    out[c] = S;
    out[C + c] = M2;
    if (c == 0) out[2 * C] = (float)M;
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ stats, int R, int C, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean_o, float* invstd_o, float* scale_o,
                                   float* shift_o) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int stride = 2 * C + 1;
  float cnt = 0.f, S = 0.f;
  for (int r = 0; r < R; ++r) { cnt += stats[r * stride + 2 * C]; S += stats[r * stride + c]; }
  const float mean = S / cnt;
  float M2 = 0.f;
  for (int r = 0; r < R; ++r) {
    float n_r = stats[r * stride + 2 * C];
    float d = stats[r * stride + c] / n_r - mean;
    M2 += stats[r * stride + C + c] + n_r * d * d;
  }
  const float var = M2 / cnt;
  const float invstd = rsqrtf(var + eps);
  mean_o[c] = mean;
  invstd_o[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale_o[c] = sc;
  shift_o[c] = beta[c] - mean * sc;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    const float unbiased = cnt > 1.f ? M2 / (cnt - 1.f) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// ------------------------------------------------------------------ BN apply (+residual) (+ReLU)
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                const float* __restrict__ shift, const T* __restrict__ res, int ldr,
                                T* __restrict__ y, int ldy, int64_t M, int C, int CP, int flags) {
  constexpr int V = DT<T>::VEC;
  const int CV = CP / V;
  const int64_t total = M * CV;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / CV;
    const int c0 = (int)(i % CV) * V;
    float v[V], r[V];
    Pack16<T>::load(x + row * ldx + c0, v);
    if (res) Pack16<T>::load(res + row * ldr + c0, r);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int c = c0 + e;
      float o = 0.f;
      if (c < C) {
        o = v[e] * scale[c] + shift[c];
        if (res) o += r[e];
        if (flags & DV_RELU) o = fmaxf(o, 0.f);
      }
      v[e] = o;
    }
    Pack16<T>::store(y + row * ldy + c0, v);
  }
}

// ------------------------------------------------------------------ column reductions over rows
// Generic: rows [r_begin, r_end) of a [rows][ld] matrix, lanes along channel vectors.  F maps
// (row, c0) -> V values for NS sums.  Result: out[s][c] for this block (written by the caller's lambda).
template <int V, int NS, typename F, typename W>
__device__ __forceinline__ void column_reduce(int64_t r_begin, int64_t r_end, int CP, F f, W write) {
  __shared__ float lds[kThreads * V * NS > 4096 ? 4096 : kThreads * V * NS];
  const int CV = CP / V;
  for (int cvb = 0; cvb < CV; cvb += kThreads) {
    const int cvc = min(kThreads, CV - cvb);
    int rg = kThreads / cvc;                       // row groups
    // LDS budget: rg * cvc * V * NS floats <= 4096
    while (rg > 1 && rg * cvc * V * NS > 4096) rg >>= 1;
    const int t = threadIdx.x;
    const int my_cv = t % cvc, my_rg = t / cvc;
    float acc[NS][V];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < V; ++e) acc[s][e] = 0.f;
    if (my_rg < rg)
      for (int64_t r = r_begin + my_rg; r < r_end; r += rg) f(r, (cvb + my_cv) * V, acc);
    __syncthreads();
    if (rg == 1 && cvc * V * NS > 4096) {
      // too wide for LDS staging (cannot happen for CP <= 4096/NS); write directly
      if (my_rg < rg) write((cvb + my_cv) * V, acc);
    } else {
      if (my_rg < rg) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int e = 0; e < V; ++e) lds[((my_rg * NS + s) * cvc + my_cv) * V + e] = acc[s][e];
      }
      __syncthreads();
      if (t < cvc) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int e = 0; e < V; ++e) {
            float a = 0.f;
            for (int g = 0; g < rg; ++g) a += lds[((g * NS + s) * cvc + t) * V + e];
            acc[s][e] = a;
          }
        write((cvb + t) * V, acc);
      }
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                     const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, int64_t M, int C, int CP, int flags,
                                     int64_t rows_per_block, float* __restrict__ part) {
  constexpr int V = DT<T>::VEC;
  const int64_t r0 = blockIdx.x * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  const bool mask = !(flags & DV_NO_RELU_MASK);
  float* outp = part + (size_t)blockIdx.x * 2 * C;
  column_reduce<V, 2>(
      r0, r1, CP,
      [&](int64_t r, int c0, float(&acc)[2][V]) {
        float g[V], yy[V], xx[V];
        Pack16<T>::load(dy + r * lddy + c0, g);
        if (mask) Pack16<T>::load(y + r * ldy + c0, yy);
        Pack16<T>::load(x + r * ldx + c0, xx);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const int c = c0 + e;
          if (c < C) {
            float gg = (mask && !(yy[e] > 0.f)) ? 0.f : g[e];
            acc[0][e] += gg;
            acc[1][e] += gg * (xx[e] - mean[c]) * invstd[c];
          }
        }
      },
      [&](int c0, float(&acc)[2][V]) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (c0 + e < C) { outp[c0 + e] = acc[0][e]; outp[C + c0 + e] = acc[1][e]; }
      });
}

// partials [n_blocks][W] -> sums[W] (W = 2C): 32 columns x 8 row lanes per block
__global__ void reduce_rows_kernel(const float* __restrict__ part, int n_blocks, int Wd, float* __restrict__ out) {
  __shared__ float sh[8][33];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31);
  const int rl = threadIdx.x >> 5;
  float a = 0.f;
  if (col < Wd)
    for (int r = rl; r < n_blocks; r += 8) a += part[(size_t)r * Wd + col];
  sh[rl][threadIdx.x & 31] = a;
  __syncthreads();
  if (rl == 0 && col < Wd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += sh[i][threadIdx.x & 31];
    out[col] = s;
  }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                    const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sums_g, const float* __restrict__ sums_l,
                                    float inv_count, float* dgamma, float* dbeta, T* __restrict__ dx, int lddx,
                                    T* __restrict__ dres, int lddres, int64_t M, int C, int CP, int flags) {
  constexpr int V = DT<T>::VEC;
  if (blockIdx.x == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      dbeta[c] += sums_l[c];
      dgamma[c] += sums_l[C + c];
    }
  }
  const int CV = CP / V;
  const int64_t total = M * CV;
  const bool mask = !(flags & DV_NO_RELU_MASK);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / CV;
    const int c0 = (int)(i % CV) * V;
    float g[V], yy[V], xx[V], o[V], ro[V];
    Pack16<T>::load(dy + row * lddy + c0, g);
    if (mask) Pack16<T>::load(y + row * ldy + c0, yy);
    Pack16<T>::load(x + row * ldx + c0, xx);
    if (dres && (flags & DV_ACCUM)) Pack16<T>::load(dres + row * lddres + c0, ro);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int c = c0 + e;
      float gg = 0.f, d = 0.f;
      if (c < C) {
        gg = (mask && !(yy[e] > 0.f)) ? 0.f : g[e];
        const float xh = (xx[e] - mean[c]) * invstd[c];
        d = gamma[c] * invstd[c] * (gg - sums_g[c] * inv_count - xh * sums_g[C + c] * inv_count);
      }
      o[e] = d;
      if (dres) ro[e] = (flags & DV_ACCUM) ? ro[e] + gg : gg;
    }
    Pack16<T>::store(dx + row * lddx + c0, o);
    if (dres) Pack16<T>::store(dres + row * lddres + c0, ro);
  }
}

// ------------------------------------------------------------------ MaxPool3d
struct PoolArgs {
  int N, Ti, Hi, Wi, C, CP;
  int To, Ho, Wo;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int ldx, ldy;
};

template <typename T>
__global__ void maxpool_fwd_kernel(PoolArgs a, const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx) {
  constexpr int V = DT<T>::VEC;
  const int CV = a.CP / V;
  const int64_t total = (int64_t)a.N * a.To * a.Ho * a.Wo * CV;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % CV) * V;
    int64_t m = i / CV;
    const int wo = (int)(m % a.Wo); int64_t q = m / a.Wo;
    const int ho = (int)(q % a.Ho); q /= a.Ho;
    const int to = (int)(q % a.To);
    const int n = (int)(q / a.To);
    float best[V]; int bi[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    int tap = 0;
    for (int dt = 0; dt < a.kt; ++dt) {
      const int t = to * a.st - a.pt + dt;
      for (int dh = 0; dh < a.kh; ++dh) {
        const int hh = ho * a.sh - a.ph + dh;
        for (int dw = 0; dw < a.kw; ++dw, ++tap) {
          const int w = wo * a.sw - a.pw + dw;
          if ((unsigned)t >= (unsigned)a.Ti || (unsigned)hh >= (unsigned)a.Hi || (unsigned)w >= (unsigned)a.Wi) continue;
          float v[V];
          Pack16<T>::load(x + ((int64_t)((n * a.Ti + t) * a.Hi + hh) * a.Wi + w) * a.ldx + c0, v);
#pragma unroll
          for (int e = 0; e < V; ++e)
            if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = tap; }
          first = false;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) if (c0 + e >= a.C) best[e] = 0.f;
    Pack16<T>::store(y + m * a.ldy + c0, best);
    uint8_t* ip = idx + m * a.CP + c0;
#pragma unroll
    for (int e = 0; e < V; ++e) ip[e] = (uint8_t)bi[e];
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(PoolArgs a, const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                   T* __restrict__ dx, int accumulate) {
  constexpr int V = DT<T>::VEC;
  const int CV = a.CP / V;
  const int64_t total = (int64_t)a.N * a.Ti * a.Hi * a.Wi * CV;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % CV) * V;
    int64_t m = i / CV;
    const int wi = (int)(m % a.Wi); int64_t q = m / a.Wi;
    const int hi = (int)(q % a.Hi); q /= a.Hi;
    const int ti = (int)(q % a.Ti);
    const int n = (int)(q / a.Ti);
    float acc[V];
    if (accumulate) Pack16<T>::load(dx + m * a.ldx + c0, acc);
    else {
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
    }
    int tap = 0;
    for (int dt = 0; dt < a.kt; ++dt) {
      const int tn = ti + a.pt - dt;
      for (int dh = 0; dh < a.kh; ++dh) {
        const int hn = hi + a.ph - dh;
        for (int dw = 0; dw < a.kw; ++dw, ++tap) {
          const int wn = wi + a.pw - dw;
          if ((tn | hn | wn) < 0) continue;
          if ((tn % a.st) | (hn % a.sh) | (wn % a.sw)) continue;
          const int to = tn / a.st, ho = hn / a.sh, wo = wn / a.sw;
          if (to >= a.To || ho >= a.Ho || wo >= a.Wo) continue;
          const int64_t mo = (int64_t)((n * a.To + to) * a.Ho + ho) * a.Wo + wo;
          float g[V];
          Pack16<T>::load(dy + mo * a.ldy + c0, g);
          const uint8_t* ip = idx + mo * a.CP + c0;
#pragma unroll
          for (int e = 0; e < V; ++e)
            if (ip[e] == tap) acc[e] += g[e];
        }
      }
    }
    Pack16<T>::store(dx + m * a.ldx + c0, acc);
  }
}

// ------------------------------------------------------------------ spatial mean / gating
template <typename T>
__global__ void spatial_mean_kernel(const T* __restrict__ x, int ldx, int S, int C, int CP, float* __restrict__ out) {
  constexpr int V = DT<T>::VEC;
  const int n = blockIdx.x;
  const float inv = 1.f / (float)S;
  column_reduce<V, 1>(
      (int64_t)n * S, (int64_t)(n + 1) * S, CP,
      [&](int64_t r, int c0, float(&acc)[1][V]) {
        float v[V];
        Pack16<T>::load(x + r * ldx + c0, v);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[0][e] += v[e];
      },
      [&](int c0, float(&acc)[1][V]) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (c0 + e < C) out[(size_t)n * C + c0 + e] = acc[0][e] * inv;
      });
}

template <typename T>
__global__ void gate_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                       const float* __restrict__ g, int S, int C, int CP, float* __restrict__ dpre) {
  constexpr int V = DT<T>::VEC;
  const int n = blockIdx.x;
  column_reduce<V, 1>(
      (int64_t)n * S, (int64_t)(n + 1) * S, CP,
      [&](int64_t r, int c0, float(&acc)[1][V]) {
        float a[V], b[V];
        Pack16<T>::load(dy + r * lddy + c0, a);
        Pack16<T>::load(x + r * ldx + c0, b);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[0][e] += a[e] * b[e];
      },
      [&](int c0, float(&acc)[1][V]) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (c0 + e < C) {
            float gg = g[(size_t)n * C + c0 + e];
            dpre[(size_t)n * C + c0 + e] = acc[0][e] * gg * (1.f - gg);
          }
      });
}

// MODE 0: y = x*g[n][c]          (gate_scale)
// MODE 1: dx (+)= dy*g + dmean/S  (gate_bwd_apply)
// MODE 2: dx (+)= dout[n][c]/S    (spatial_mean_bwd)
template <typename T, int MODE>
__global__ void rowscale_kernel(const T* __restrict__ a, int lda, const float* __restrict__ g,
                                const float* __restrict__ dm, int N, int S, int C, int CP, T* __restrict__ o,
                                int ldo, int accumulate) {
  constexpr int V = DT<T>::VEC;
  const int CV = CP / V;
  const int64_t total = (int64_t)N * S * CV;
  const float invS = 1.f / (float)S;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / CV;
    const int c0 = (int)(i % CV) * V;
    const int n = (int)(row / S);
    float v[V], old[V];
    if (MODE != 2) Pack16<T>::load(a + row * lda + c0, v);
    if (accumulate) Pack16<T>::load(o + row * ldo + c0, old);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int c = c0 + e;
      float r = 0.f;
      if (c < C) {
        if (MODE == 0) r = v[e] * g[(size_t)n * C + c];
        else if (MODE == 1) r = v[e] * g[(size_t)n * C + c] + dm[(size_t)n * C + c] * invS;
        else r = dm[(size_t)n * C + c] * invS;
      }
      v[e] = accumulate ? old[e] + r : r;
    }
    Pack16<T>::store(o + row * ldo + c0, v);
  }
}

// ------------------------------------------------------------------ small fp32 helpers
__global__ void colsum_kernel(const float* __restrict__ x, int ldx, int R, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < R; ++r) s += x[(size_t)r * ldx + c];
  out[c] += s;
}

// one wave per row
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, int R, int D, float eps, float* __restrict__ y,
                                  float* __restrict__ norm) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) { float v = x[(size_t)row * D + d]; s += v * v; }
  s = wave_sum(s);
  const float nrm = fmaxf(sqrtf(s), eps);
  for (int d = lane; d < D; d += 64) y[(size_t)row * D + d] = x[(size_t)row * D + d] / nrm;
  if (lane == 0) norm[row] = nrm;
}
__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                  const float* __restrict__ norm, int R, int D, float* __restrict__ dx) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += dy[(size_t)row * D + d] * y[(size_t)row * D + d];
  s = wave_sum(s);
  const float inv = 1.f / norm[row];
  for (int d = lane; d < D; d += 64)
    dx[(size_t)row * D + d] = (dy[(size_t)row * D + d] - y[(size_t)row * D + d] * s) * inv;
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int64_t n, float* __restrict__ dx) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ void mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ float sh[8];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

// ------------------------------------------------------------------ optimizer / arenas
template <typename CT>
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n,
                           float lr, float mu, float wd, float gs, CT* __restrict__ copy) {
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = *reinterpret_cast<const f32x4*>(g + i),
            bv = *reinterpret_cast<f32x4*>(buf + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float d = gv[e] * gs + wd * pv[e];
        bv[e] = mu * bv[e] + d;
        pv[e] = pv[e] - lr * bv[e];
      }
      *reinterpret_cast<f32x4*>(p + i) = pv;
      *reinterpret_cast<f32x4*>(buf + i) = bv;
      if (copy)
#pragma unroll
        for (int e = 0; e < 4; ++e) copy[i + e] = (CT)pv[e];
    } else {
      for (int64_t j = i; j < n; ++j) {
        float d = g[j] * gs + wd * p[j];
        buf[j] = mu * buf[j] + d;
        p[j] -= lr * buf[j];
        if (copy) copy[j] = (CT)p[j];
      }
    }
  }
}
template <typename CT>
__global__ void ema_kernel(float* __restrict__ k, const float* __restrict__ q, int64_t n, float m, CT* __restrict__ copy) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = k[i] * m + q[i] * (1.f - m);
    k[i] = v;
    if (copy) copy[i] = (CT)v;
  }
}
template <typename CT>
__global__ void cast_kernel(const float* __restrict__ s, CT* __restrict__ d, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (CT)s[i];
}
// one block per (desc, input channel c): Wd[c][tap][n] = W[n][tap][c]
template <typename CT>
__global__ void pack_dgrad_kernel(const float* __restrict__ master, CT* __restrict__ dst,
                                  const dv_pack_desc* __restrict__ descs, const int* __restrict__ bmap) {
  const dv_pack_desc d = descs[bmap[2 * blockIdx.x]];
  const int c = bmap[2 * blockIdx.x + 1];
  const int rowlen = d.taps * d.cout_pitch;
  const float* src = master + d.src_off;
  CT* out = dst + d.dst_off + (int64_t)c * rowlen;
  for (int i = threadIdx.x; i < rowlen; i += blockDim.x) {
    const int tap = i / d.cout_pitch, n = i % d.cout_pitch;
    float v = 0.f;
    if (n < d.Cout) v = src[((int64_t)n * d.taps + tap) * d.cin_pitch + c];
    out[i] = (CT)v;
  }
}

}  // namespace

#define ST(s) ((hipStream_t)(s))
#define DISPATCH_T(dtype, ...)                                  \
  do {                                                          \
    if ((dtype) == DV_F32) { typedef float T; __VA_ARGS__; }    \
    else if ((dtype) == DV_BF16) { typedef bf16_t T; __VA_ARGS__; } \
    else return DV_EUNSUPPORTED;                                \
  } while (0)

static inline int cp8(int c) { return (c + 7) & ~7; }

extern "C" int dv_abi_version(void) { return DV_ABI_VERSION; }

extern "C" int dv_check_device(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DV_EUNSUPPORTED;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return DV_EUNSUPPORTED;
  const char* a = prop.gcnArchName;
  return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? DV_OK : DV_EUNSUPPORTED;
}

extern "C" int dv_ingest_ncdhw(int32_t dtype, const float* x, void* y, int32_t N, int32_t C, int32_t T_, int32_t H,
                               int32_t W, int64_t sxn, int32_t ldy, const float* mean3, const float* istd3,
                               const int32_t* perm, int32_t n_seg, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || C > 4 || ldy < 4 || ldy % 4) return DV_EINVAL;
  if (perm && (n_seg <= 0 || T_ % n_seg)) return DV_EINVAL;
  if ((mean3 == nullptr) != (istd3 == nullptr)) return DV_EINVAL;
  const int64_t total = (int64_t)N * T_ * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL((ingest_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, ST(stream), x,
                                       (T*)y, N, C, T_, H, W, sxn, ldy, mean3, istd3, perm, n_seg));
  return dv_launch_status();
}

extern "C" int dv_bn_reduce_stats(const float* partials, int32_t n_tiles, int32_t tile_rows, int64_t M, int32_t C,
                                  float* local_stats, void* stream) {
  if (!partials || !local_stats || n_tiles <= 0 || C <= 0 || M <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(bn_reduce_stats_kernel, dim3(C), dim3(kThreads), 0, ST(stream), partials, n_tiles, tile_rows, M, C,
                     local_stats);
  return dv_launch_status();
}

extern "C" int dv_bn_finalize(const float* stats, int32_t R, int32_t C, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                              float* scale, float* shift, void* stream) {
  if (!stats || R <= 0 || C <= 0 || !gamma || !beta || !mean || !invstd || !scale || !shift) return DV_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return DV_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, ST(stream), stats, R, C, gamma, beta, eps,
                     momentum, running_mean, running_var, mean, invstd, scale, shift);
  return dv_launch_status();
}

extern "C" int dv_bn_apply(int32_t dtype, const void* x, int32_t ldx, const float* scale, const float* shift,
                           const void* residual, int32_t ldr, void* y, int32_t ldy, int64_t M, int32_t C, int32_t flags,
                           void* stream) {
  const int CP = cp8(C);
  if (!x || !y || !scale || !shift || M <= 0 || C <= 0 || ldx < CP || ldy < CP || (residual && ldr < CP)) return DV_EINVAL;
  if (!aligned16(x) || !aligned16(y) || (residual && !aligned16(residual))) return DV_EALIGN;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (ldx % V || ldy % V || (residual && ldr % V)) return DV_EALIGN;
    hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(M * (CP / V))), dim3(kThreads), 0, ST(stream), (const T*)x,
                       ldx, scale, shift, (const T*)residual, ldr, (T*)y, ldy, M, C, CP, flags);
  });
  return dv_launch_status();
}

extern "C" int dv_bn_bwd_blocks(int64_t M, int32_t C) {
  (void)C;
  int64_t b = (M + 63) / 64;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int dv_bn_bwd_reduce(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                                int32_t ldx, const float* mean, const float* invstd, int64_t M, int32_t C, int32_t flags,
                                float* partials, void* stream) {
  const int CP = cp8(C);
  const bool mask = !(flags & DV_NO_RELU_MASK);
  if (!dy || !x || (mask && !y) || !mean || !invstd || !partials || M <= 0 || C <= 0) return DV_EINVAL;
  if (lddy < CP || ldx < CP || (mask && ldy < CP)) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(x) || (mask && !aligned16(y))) return DV_EALIGN;
  const int blocks = dv_bn_bwd_blocks(M, C);
  const int64_t rpb = (M + blocks - 1) / blocks;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (lddy % V || ldx % V || (mask && ldy % V)) return DV_EALIGN;
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(blocks), dim3(kThreads), 0, ST(stream), (const T*)dy, lddy,
                       (const T*)y, ldy, (const T*)x, ldx, mean, invstd, M, C, CP, flags, rpb, partials);
  });
  return dv_launch_status();
}

extern "C" int dv_bn_bwd_finalize(const float* partials, int32_t n_blocks, int32_t C, float* sums, void* stream) {
  if (!partials || !sums || n_blocks <= 0 || C <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * C + 31) / 32), dim3(256), 0, ST(stream), partials, n_blocks, 2 * C, sums);
  return dv_launch_status();
}

extern "C" int dv_bn_bwd_apply(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                               int32_t ldx, const float* mean, const float* invstd, const float* gamma,
                               const float* sums_global, const float* sums_local, float inv_count, float* dgamma,
                               float* dbeta, void* dx, int32_t lddx, void* dres, int32_t lddres, int64_t M, int32_t C,
                               int32_t flags, void* stream) {
  const int CP = cp8(C);
  const bool mask = !(flags & DV_NO_RELU_MASK);
  if (!dy || !x || (mask && !y) || !mean || !invstd || !gamma || !sums_global || !dx || M <= 0 || C <= 0) return DV_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr) || (dgamma && !sums_local)) return DV_EINVAL;
  if (lddy < CP || ldx < CP || lddx < CP || (mask && ldy < CP) || (dres && lddres < CP)) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(x) || !aligned16(dx) || (mask && !aligned16(y)) || (dres && !aligned16(dres))) return DV_EALIGN;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (lddy % V || ldx % V || lddx % V || (mask && ldy % V) || (dres && lddres % V)) return DV_EALIGN;
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(grid_for(M * (CP / V))), dim3(kThreads), 0, ST(stream),
                       (const T*)dy, lddy, (const T*)y, ldy, (const T*)x, ldx, mean, invstd, gamma, sums_global,
                       sums_local, inv_count, dgamma, dbeta, (T*)dx, lddx, (T*)dres, lddres, M, C, CP, flags);
  });
  return dv_launch_status();
}

static int pool_args(const dv_pool_desc* d, PoolArgs& a) {
  if (!d) return DV_EINVAL;
  if (d->N <= 0 || d->C <= 0 || d->kt <= 0 || d->kh <= 0 || d->kw <= 0 || d->st <= 0 || d->sh <= 0 || d->sw <= 0) return DV_EINVAL;
  if (d->kt * d->kh * d->kw > 255) return DV_EUNSUPPORTED;
  if ((d->Ti + 2 * d->pt - d->kt) / d->st + 1 != d->To || (d->Hi + 2 * d->ph - d->kh) / d->sh + 1 != d->Ho ||
      (d->Wi + 2 * d->pw - d->kw) / d->sw + 1 != d->Wo) return DV_EINVAL;
  if (2 * d->pt > d->kt || 2 * d->ph > d->kh || 2 * d->pw > d->kw) return DV_EINVAL;
  a.N = d->N; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.C = d->C; a.CP = cp8(d->C);
  a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo;
  a.kt = d->kt; a.kh = d->kh; a.kw = d->kw; a.st = d->st; a.sh = d->sh; a.sw = d->sw;
  a.pt = d->pt; a.ph = d->ph; a.pw = d->pw; a.ldx = d->ldx; a.ldy = d->ldy;
  if (a.ldx < a.CP || a.ldy < a.CP) return DV_EINVAL;
  const int V = d->dtype == DV_F32 ? 4 : 8;
  if (a.ldx % V || a.ldy % V) return DV_EALIGN;
  return DV_OK;
}

extern "C" int dv_maxpool3d_fwd(const dv_pool_desc* d, const void* x, void* y, uint8_t* idx, void* stream) {
  PoolArgs a;
  int rc = pool_args(d, a);
  if (rc) return rc;
  if (!x || !y || !idx) return DV_EINVAL;
  if (!aligned16(x) || !aligned16(y) || (reinterpret_cast<uintptr_t>(idx) & 7)) return DV_EALIGN;
  DISPATCH_T(d->dtype, {
    const int64_t total = (int64_t)a.N * a.To * a.Ho * a.Wo * (a.CP / DT<T>::VEC);
    hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for(total, 16384)), dim3(kThreads), 0, ST(stream), a, (const T*)x,
                       (T*)y, idx);
  });
  return dv_launch_status();
}

extern "C" int dv_maxpool3d_bwd(const dv_pool_desc* d, const void* dy, const uint8_t* idx, void* dx, int32_t flags,
                                void* stream) {
  PoolArgs a;
  int rc = pool_args(d, a);
  if (rc) return rc;
  if (!dy || !dx || !idx) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(dx)) return DV_EALIGN;
  DISPATCH_T(d->dtype, {
    const int64_t total = (int64_t)a.N * a.Ti * a.Hi * a.Wi * (a.CP / DT<T>::VEC);
    hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for(total, 16384)), dim3(kThreads), 0, ST(stream), a, (const T*)dy,
                       idx, (T*)dx, (flags & DV_ACCUM) ? 1 : 0);
  });
  return dv_launch_status();
}

extern "C" int dv_spatial_mean(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t S, int32_t C, float* out,
                               void* stream) {
  const int CP = cp8(C);
  if (!x || !out || N <= 0 || S <= 0 || C <= 0 || ldx < CP) return DV_EINVAL;
  if (!aligned16(x)) return DV_EALIGN;
  DISPATCH_T(dtype, {
    if (ldx % DT<T>::VEC) return DV_EALIGN;
    hipLaunchKernelGGL((spatial_mean_kernel<T>), dim3(N), dim3(kThreads), 0, ST(stream), (const T*)x, ldx, S, C, CP, out);
  });
  return dv_launch_status();
}

extern "C" int dv_spatial_mean_bwd(int32_t dtype, const float* dout, int32_t N, int32_t S, int32_t C, void* dx,
                                   int32_t lddx, int32_t flags, void* stream) {
  const int CP = cp8(C);
  if (!dout || !dx || N <= 0 || S <= 0 || C <= 0 || lddx < CP) return DV_EINVAL;
  if (!aligned16(dx)) return DV_EALIGN;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (lddx % V) return DV_EALIGN;
    hipLaunchKernelGGL((rowscale_kernel<T, 2>), dim3(grid_for((int64_t)N * S * (CP / V))), dim3(kThreads), 0, ST(stream),
                       (const T*)nullptr, 0, (const float*)nullptr, dout, N, S, C, CP, (T*)dx, lddx,
                       (flags & DV_ACCUM) ? 1 : 0);
  });
  return dv_launch_status();
}

extern "C" int dv_gate_scale(int32_t dtype, const void* x, int32_t ldx, const float* g, int32_t N, int32_t S, int32_t C,
                             void* y, int32_t ldy, void* stream) {
  const int CP = cp8(C);
  if (!x || !g || !y || N <= 0 || S <= 0 || C <= 0 || ldx < CP || ldy < CP) return DV_EINVAL;
  if (!aligned16(x) || !aligned16(y)) return DV_EALIGN;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (ldx % V || ldy % V) return DV_EALIGN;
    hipLaunchKernelGGL((rowscale_kernel<T, 0>), dim3(grid_for((int64_t)N * S * (CP / V))), dim3(kThreads), 0, ST(stream),
                       (const T*)x, ldx, g, (const float*)nullptr, N, S, C, CP, (T*)y, ldy, 0);
  });
  return dv_launch_status();
}

extern "C" int dv_gate_bwd_reduce(int32_t dtype, const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* g,
                                  int32_t N, int32_t S, int32_t C, float* dpre, void* stream) {
  const int CP = cp8(C);
  if (!dy || !x || !g || !dpre || N <= 0 || S <= 0 || C <= 0 || lddy < CP || ldx < CP) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(x)) return DV_EALIGN;
  DISPATCH_T(dtype, {
    if (lddy % DT<T>::VEC || ldx % DT<T>::VEC) return DV_EALIGN;
    hipLaunchKernelGGL((gate_bwd_reduce_kernel<T>), dim3(N), dim3(kThreads), 0, ST(stream), (const T*)dy, lddy,
                       (const T*)x, ldx, g, S, C, CP, dpre);
  });
  return dv_launch_status();
}

extern "C" int dv_gate_bwd_apply(int32_t dtype, const void* dy, int32_t lddy, const float* g, const float* dmean, int32_t N,
                                 int32_t S, int32_t C, void* dx, int32_t lddx, int32_t flags, void* stream) {
  const int CP = cp8(C);
  if (!dy || !g || !dmean || !dx || N <= 0 || S <= 0 || C <= 0 || lddy < CP || lddx < CP) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(dx)) return DV_EALIGN;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (lddy % V || lddx % V) return DV_EALIGN;
    hipLaunchKernelGGL((rowscale_kernel<T, 1>), dim3(grid_for((int64_t)N * S * (CP / V))), dim3(kThreads), 0, ST(stream),
                       (const T*)dy, lddy, g, dmean, N, S, C, CP, (T*)dx, lddx, (flags & DV_ACCUM) ? 1 : 0);
  });
  return dv_launch_status();
}

extern "C" int dv_colsum_f32(const float* x, int32_t ldx, int32_t R, int32_t C, float* out, void* stream) {
  if (!x || !out || R <= 0 || C <= 0 || ldx < C) return DV_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 127) / 128), dim3(128), 0, ST(stream), x, ldx, R, C, out);
  return dv_launch_status();
}
extern "C" int dv_l2norm_fwd(const float* x, int32_t R, int32_t D, float eps, float* y, float* norm, void* stream) {
  if (!x || !y || !norm || R <= 0 || D <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, ST(stream), x, R, D, eps, y, norm);
  return dv_launch_status();
}
extern "C" int dv_l2norm_bwd(const float* dy, const float* y, const float* norm, int32_t R, int32_t D, float* dx, void* stream) {
  if (!dy || !y || !norm || !dx || R <= 0 || D <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, ST(stream), dy, y, norm, R, D, dx);
  return dv_launch_status();
}
extern "C" int dv_relu_bwd_f32(const float* dy, const float* y, int64_t n, float* dx, void* stream) {
  if (!dy || !y || !dx || n <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, ST(stream), dy, y, n, dx);
  return dv_launch_status();
}
extern "C" int dv_mean_f32(const float* x, int32_t n, float* out, void* stream) {
  if (!x || !out || n <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, ST(stream), x, n, out);
  return dv_launch_status();
}

extern "C" int dv_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float mu, float wd, float gs,
                               int32_t copy_dtype, void* p_copy, void* stream) {
  if (!p || !g || !buf || n <= 0) return DV_EINVAL;
  if (!aligned16(p) || !aligned16(g) || !aligned16(buf)) return DV_EALIGN;
  const int grid = grid_for((n + 3) / 4, 2048);
  if (p_copy && copy_dtype == DV_BF16)
    hipLaunchKernelGGL((sgd_kernel<bf16_t>), dim3(grid), dim3(kThreads), 0, ST(stream), p, g, buf, n, lr, mu, wd, gs, (bf16_t*)p_copy);
  else
    hipLaunchKernelGGL((sgd_kernel<float>), dim3(grid), dim3(kThreads), 0, ST(stream), p, g, buf, n, lr, mu, wd, gs,
                       (float*)(copy_dtype == DV_F32 ? p_copy : nullptr));
  return dv_launch_status();
}
extern "C" int dv_ema(float* k, const float* q, int64_t n, float m, int32_t copy_dtype, void* k_copy, void* stream) {
  if (!k || !q || n <= 0) return DV_EINVAL;
  const int grid = grid_for(n, 2048);
  if (k_copy && copy_dtype == DV_BF16)
    hipLaunchKernelGGL((ema_kernel<bf16_t>), dim3(grid), dim3(kThreads), 0, ST(stream), k, q, n, m, (bf16_t*)k_copy);
  else
    hipLaunchKernelGGL((ema_kernel<float>), dim3(grid), dim3(kThreads), 0, ST(stream), k, q, n, m,
                       (float*)(copy_dtype == DV_F32 ? k_copy : nullptr));
  return dv_launch_status();
}
extern "C" int dv_cast_arena(int32_t dtype, const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return DV_EINVAL;
  if (dtype == DV_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t>), dim3(grid_for(n, 2048)), dim3(kThreads), 0, ST(stream), src, (bf16_t*)dst, n);
  else if (dtype == DV_F32) hipLaunchKernelGGL((cast_kernel<float>), dim3(grid_for(n, 2048)), dim3(kThreads), 0, ST(stream), src, (float*)dst, n);
  else return DV_EUNSUPPORTED;
  return dv_launch_status();
}
extern "C" int dv_pack_dgrad_weights(int32_t dtype, const float* master, void* dst, const dv_pack_desc* descs,
                                     const int32_t* block_map, int32_t n_blocks, void* stream) {
  if (!master || !dst || !descs || !block_map || n_blocks <= 0) return DV_EINVAL;
  if (dtype == DV_BF16) hipLaunchKernelGGL((pack_dgrad_kernel<bf16_t>), dim3(n_blocks), dim3(kThreads), 0, ST(stream), master, (bf16_t*)dst, descs, block_map);
  else if (dtype == DV_F32) hipLaunchKernelGGL((pack_dgrad_kernel<float>), dim3(n_blocks), dim3(kThreads), 0, ST(stream), master, (float*)dst, descs, block_map);
  else return DV_EUNSUPPORTED;
  return dv_launch_status();
}
