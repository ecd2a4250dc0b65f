// Contrastive objectives of the DualVar hot path on gfx950: similarity products on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32, exact fp32) + wave-per-row masked log-softmax
// cross-entropy that also emits the reference-ordered logits, the top-k rank of the positive
// and the gradient w.r.t. the similarity matrix.  See include/dualvar_hip.h for the citations.
#include "common.hpp"

namespace {

// C[m][n] (+)= alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]; one wave per 32x32 tile.
__global__ __launch_bounds__(256) void gemm_f32_kernel(int M, int N, int K, const float* __restrict__ A, int64_t sam,
                                                        int64_t sak, const float* __restrict__ B, int64_t sbk,
                                                        int64_t sbn, float* __restrict__ C, int64_t ldc, float alpha,
                                                        int accumulate, int tiles_n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * 32, n0 = tn * 32;
  if (m0 >= M) return;
  const int l31 = lane & 31, h = lane >> 5;
  const int m = m0 + l31, n = n0 + l31;
  const float* ap = A + (int64_t)m * sam;
  const float* bp = B + (int64_t)n * sbn;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // 8 k-steps per trip: the 16 loads are issued before the first MFMA consumes them (the operands come
  // straight from global memory / L2, so the loop is latency-bound without this ILP)
  int k = 0;
  for (; k + 16 <= K; k += 16) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k + 2 * u + h;
      a[u] = (m < M) ? ap[(int64_t)kk * sak] : 0.f;
      b[u] = (n < N) ? bp[(int64_t)kk * sbk] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  for (; k < K; k += 2) {
    const int kk = k + h;
    float a = (m < M && kk < K) ? ap[(int64_t)kk * sak] : 0.f;
    float b = (n < N && kk < K) ? bp[(int64_t)kk * sbk] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  if (n < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < M) {
        float* p = C + (int64_t)row * ldc + n;
        float v = alpha * acc[r];
        *p = accumulate ? *p + v : v;
      }
    }
  }
}

// Several independent small GEMMs in one launch (the four self-gating FCs of an Inception block and their
// backward products).  Descriptors live in device memory and are static per plan.
// One workgroup per 32x32 output tile; its four waves split K in interleaved 16-deep chunks (these GEMMs are a few
// dozen tiles with K up to 832: one wave per tile left the chip idle and serialised ~50 dependent load rounds) and
// reduce through LDS.  A K-contiguous operand is read as two float4 per lane and chunk instead of eight strided
// dwords.  Within a chunk lane half h multiplies k = chunk + 8h + u (u = 0..7): any pairing works as long as A and B
// agree, this one makes each lane's eight values contiguous.
__device__ __forceinline__ void gemm_tile4_body(const dv_gemm_desc& d, int tile, float (&red)[3][16][64]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tiles_n = (d.N + 31) / 32;
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  const int l31 = lane & 31, h = lane >> 5;
  const int m = m0 + l31, n = n0 + l31;
  const bool mok = m < d.M, nok = n < d.N;
  const float* ap = d.A + (int64_t)(mok ? m : 0) * d.sam;
  const float* bp = d.B + (int64_t)(nok ? n : 0) * d.sbn;
  const bool avec = d.sak == 1 && (d.sam & 3) == 0 && ((uintptr_t)d.A & 15) == 0;
  const bool bvec = d.sbk == 1 && (d.sbn & 3) == 0 && ((uintptr_t)d.B & 15) == 0;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // U chunks per round: all loads of a round are issued before its MFMAs, so a wave pays the load latency K/(64 U)
  // times instead of K/64 times (the loop is latency bound: 13 dependent rounds at K = 832 were 12-16 us per launch)
  constexpr int U = 4;
  auto load_chunk = [&](int kc, float (&a)[8], float (&b)[8]) {
    const int kb = kc + 8 * h;
    if (kc + 16 <= d.K) {
      if (avec) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(ap + kb), v1 = *reinterpret_cast<const f32x4*>(ap + kb + 4);
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = ap[(int64_t)(kb + u) * d.sak];
      }
      if (bvec) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(bp + kb), v1 = *reinterpret_cast<const f32x4*>(bp + kb + 4);
        b[0] = v0.x; b[1] = v0.y; b[2] = v0.z; b[3] = v0.w; b[4] = v1.x; b[5] = v1.y; b[6] = v1.z; b[7] = v1.w;
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = bp[(int64_t)(kb + u) * d.sbk];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool kok = kb + u < d.K;                  // also covers kc >= K: the whole chunk is zero
        a[u] = kok ? ap[(int64_t)(kb + u) * d.sak] : 0.f;
        b[u] = kok ? bp[(int64_t)(kb + u) * d.sbk] : 0.f;
      }
    }
  };
  for (int kc = wave * 16; kc < d.K; kc += 64 * U) {
    float a[U][8], b[U][8];
#pragma unroll
    for (int q = 0; q < U; ++q) load_chunk(kc + 64 * q, a[q], b[q]);
#pragma unroll
    for (int q = 0; q < U; ++q) {
      if (kc + 64 * q < d.K) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mok ? a[q][u] : 0.f, nok ? b[q][u] : 0.f, acc, 0, 0, 0);
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0 && nok) {
    const float bv = d.bias ? d.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < d.M) {
        float* p = d.C + (int64_t)row * d.ldc + n;
        float v = d.alpha * (((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane]) + bv;
        if (d.flags & DV_ACCUM) v += *p;
        if (d.flags & DV_SIGMOID) v = 1.f / (1.f + __expf(-v));
        if (d.flags & DV_RELU) v = fmaxf(v, 0.f);
        *p = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gemm_f32_grouped_kernel(const dv_gemm_desc* __restrict__ descs, int n_groups) {
  __shared__ float red[3][16][64];
  int tile = blockIdx.x;
  int gi = 0, tstart = 0;
  {  // independent loads of the first eight prefixes instead of a chain of dependent ones
    int e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = k < n_groups ? descs[k].tile_end : 0x7fffffff;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < n_groups && tile >= e[k]) { gi = k + 1; tstart = e[k]; }
    while (gi >= 8 && gi < n_groups && tile >= descs[gi].tile_end) { tstart = descs[gi].tile_end; ++gi; }
  }
  if (gi >= n_groups) return;
  const dv_gemm_desc d = descs[gi];
  gemm_tile4_body(d, tile - tstart, red);
}

// one GEMM, descriptor in the kernel arguments: the same workgroup-per-tile, K-over-four-waves scheme for the single
// products of the heads and losses (128 x 1024 x 1024: 32 workgroups of one wave per tile walking K = 1024 took 65-70 us)
__global__ __launch_bounds__(256) void gemm_f32_tile4_kernel(dv_gemm_desc d) {
  __shared__ float red[3][16][64];
  gemm_tile4_body(d, (int)blockIdx.x, red);
}

// Few output tiles, long K (MoCo's dq = dlogits . queue^T: 32 x 128 outputs over K = 65 536 -- four tiles, i.e. four waves
// walking 4 096 dependent load rounds each: 4 ms): the K range is cut into `splits` slices, one workgroup per (tile,
// slice), its four waves interleave 16-deep chunks of the slice, reduce through LDS and add alpha * partial to C with
// fp32 atomics (C zeroed first unless accumulating).
__global__ __launch_bounds__(256) void gemm_f32_splitk_kernel(int M, int N, int K, const float* __restrict__ A, int64_t sam,
                                                               int64_t sak, const float* __restrict__ B, int64_t sbk,
                                                               int64_t sbn, float* __restrict__ C, int64_t ldc, float alpha,
                                                               int tiles_n, int tiles, int kslice, float* __restrict__ ws,
                                                               int splits, int accumulate) {
  __shared__ float red[3][16][64];
  __shared__ uint32_t last;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  const int l31 = lane & 31, h = lane >> 5;
  const int m = m0 + l31, n = n0 + l31;
  const bool mok = m < M, nok = n < N;
  const float* ap = A + (int64_t)(mok ? m : 0) * sam;
  const float* bp = B + (int64_t)(nok ? n : 0) * sbn;
  const int k_begin = slice * kslice, k_end = min(K, k_begin + kslice);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kc = k_begin + wave * 16; kc < k_end; kc += 64) {
    const int kb = kc + 8 * h;
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool kok = kb + u < k_end;
      a[u] = (kok && mok) ? ap[(int64_t)(kb + u) * sak] : 0.f;
      b[u] = (kok && nok) ? bp[(int64_t)(kb + u) * sbk] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (ws == nullptr) {
    if (wave == 0 && nok) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) atomicAdd(C + (int64_t)row * ldc + n, alpha * (((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane]));
      }
    }
    return;
  }
  // Ordered form: the slice's 32x32 partial goes to ws[slice][tile][16][64]; the workgroup that takes the last ticket of its
  // tile adds the slices in order and writes C -- no float atomics, the result does not depend on the arrival order.
  float* part = ws + ((size_t)slice * tiles + tile) * 1024;
  // (partial tiles and tickets as agent-scope relaxed atomics ordered by s_waitcnt, not by agent-scope fences: see
  // elementwise.hip, ordered_fold)
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      __hip_atomic_store(part + r * 64 + lane, ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  uint32_t* ticket = reinterpret_cast<uint32_t*>(ws + (size_t)splits * tiles * 1024) + tile;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) last = (atomicAdd(ticket, 1u) == (uint32_t)splits - 1) ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int r = i >> 6, ln = i & 63;
    const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), col = n0 + (ln & 31);
    float t = 0.f;
    const float* src = ws + (size_t)tile * 1024 + i;
    const size_t pitch = (size_t)tiles * 1024;
    int sl = 0;
    for (; sl + 8 <= splits; sl += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(const_cast<float*>(src) + (size_t)(sl + u) * pitch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; sl < splits; ++sl) t += __hip_atomic_load(const_cast<float*>(src) + (size_t)sl * pitch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (row < M && col < N) {
      float* c = C + (int64_t)row * ldc + col;
      *c = accumulate ? *c + alpha * t : alpha * t;
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void zero_matrix_kernel(float* __restrict__ C, int64_t ldc, int M, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M * N) C[(int64_t)(i / N) * ldc + i % N] = 0.f;
}

// split-K plan of launch_gemm_f32 (0 splits: the GEMM does not take the split path)
static void gemm_splitk_plan(int M, int N, int K, int& tiles, int& tiles_n, int& splits, int& kslice) {
  const int tiles_m = (M + 31) / 32;
  tiles_n = (N + 31) / 32;
  tiles = tiles_m * tiles_n;
  splits = 0; kslice = 0;
  if (tiles <= 64 && K >= 4096) {
    splits = 1024 / tiles;
    if (splits > K / 256) splits = K / 256;
    kslice = (((K + splits - 1) / splits) + 63) / 64 * 64;
    splits = (K + kslice - 1) / kslice;
  }
}
static int64_t gemm_splitk_ws_bytes(int M, int N, int K) {
  int tiles, tiles_n, splits, kslice;
  gemm_splitk_plan(M, N, K, tiles, tiles_n, splits, kslice);
  return splits ? ((int64_t)splits * tiles * 1024 + ((tiles + 7) & ~7)) * 4 : 0;
}

static int launch_gemm_f32(int M, int N, int K, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
                           int64_t sbn, float* C, int64_t ldc, float alpha, int accumulate, hipStream_t s,
                           float* ws = nullptr, int64_t ws_bytes = 0) {
  int tiles, tiles_n, splits, kslice;
  gemm_splitk_plan(M, N, K, tiles, tiles_n, splits, kslice);
  if (splits) {
    if (ws && ws_bytes >= gemm_splitk_ws_bytes(M, N, K)) {          // ordered: partial tiles + tickets, no atomics, no memset
      hipLaunchKernelGGL(gemm_f32_splitk_kernel, dim3(tiles * splits), dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc,
                         alpha, tiles_n, tiles, kslice, ws, splits, accumulate);
      return dv_launch_status();
    }
    if (!accumulate) {
      hipLaunchKernelGGL(zero_matrix_kernel, dim3((M * N + 255) / 256), dim3(256), 0, s, C, ldc, M, N);
      int rc = dv_launch_status();
      if (rc) return rc;
    }
    hipLaunchKernelGGL(gemm_f32_splitk_kernel, dim3(tiles * splits), dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc,
                       alpha, tiles_n, tiles, kslice, (float*)nullptr, splits, accumulate);
    return dv_launch_status();
  }
  if (tiles <= 1024 && K >= 64) {          // few tiles: one workgroup per tile, K over its four waves
    dv_gemm_desc d;
    d.A = A; d.B = B; d.C = C; d.bias = nullptr;
    d.sam = sam; d.sak = sak; d.sbk = sbk; d.sbn = sbn; d.ldc = ldc;
    d.M = M; d.N = N; d.K = K; d.flags = accumulate ? DV_ACCUM : 0; d.tile_end = tiles; d.alpha = alpha;
    hipLaunchKernelGGL(gemm_f32_tile4_kernel, dim3(tiles), dim3(256), 0, s, d);
    return dv_launch_status();
  }
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, M, N, K, A, sam, sak, B, sbk, sbn, C, ldc,
                     alpha, accumulate, tiles_n);
  return dv_launch_status();
}

// One wave per row.  On entry sim[r][c] holds the scaled similarities (dot * inv_T); on exit the
// gradient of mean_r(loss_r) w.r.t. the unscaled dot products.
__global__ void ntxent_rows_kernel(float* __restrict__ sim, int R, int n_local, int N, int row_index0, float inv_T,
                                   float* __restrict__ logits, float* __restrict__ loss_rows, int* __restrict__ rank0) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const int C2 = 2 * N;
  const int gi = row_index0 + (r / n_local) * N + (r % n_local);
  const int pos = (gi + N) % C2;
  float* s = sim + (size_t)r * C2;
  float mx = -INFINITY;
  for (int c = lane; c < C2; c += 64)
    if (c != gi) mx = fmaxf(mx, s[c]);
  mx = wave_max(mx);
  const float sp = s[pos];
  float se = 0.f, cnt = 0.f;
  for (int c = lane; c < C2; c += 64)
    if (c != gi) {
      float v = s[c];
      se += __expf(v - mx);
      if (c != pos && v > sp) cnt += 1.f;
    }
  se = wave_sum(se);
  cnt = wave_sum(cnt);
  const float lse = logf(se) + mx;
  if (lane == 0) {
    loss_rows[r] = lse - sp;
    rank0[r] = (int)cnt;
  }
  float* lg = logits + (size_t)r * (C2 - 1);
  const float gscale = inv_T / (float)R;
  for (int c = lane; c < C2; c += 64) {
    float v = s[c];
    float grad = 0.f;
    if (c != gi) {
      float p = __expf(v - lse);
      grad = (p - (c == pos ? 1.f : 0.f)) * gscale;
      int j = (c == pos) ? 0 : 1 + c - (c > gi ? 1 : 0) - (c > pos ? 1 : 0);
      lg[j] = v;
    }
    s[c] = grad;
  }
}

// MoCo InfoNCE rows: logits[b][0] = q.k*inv_T computed here, logits[b][1..K] already hold q.queue*inv_T.
__global__ void infonce_rows_kernel(const float* __restrict__ q, const float* __restrict__ k, int B, int D, int K,
                                    float inv_T, float* __restrict__ logits, float* __restrict__ loss_rows,
                                    int* __restrict__ rank0, float* __restrict__ dlogits) {
  const int b = blockIdx.x;
  __shared__ float sh[16];
  const int t = threadIdx.x;
  float d = 0.f;
  for (int i = t; i < D; i += blockDim.x) d += q[(size_t)b * D + i] * k[(size_t)b * D + i];
  d = wave_sum(d);
  if ((t & 63) == 0) sh[t >> 6] = d;
  __syncthreads();
  float l0 = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) l0 += sh[i];
  l0 *= inv_T;
  __syncthreads();
  float* lg = logits + (size_t)b * (K + 1);
  float mx = l0;
  for (int j = 1 + t; j <= K; j += blockDim.x) mx = fmaxf(mx, lg[j]);
  mx = wave_max(mx);
  if ((t & 63) == 0) sh[t >> 6] = mx;
  __syncthreads();
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) mx = fmaxf(mx, sh[i]);
  __syncthreads();
  float se = 0.f, cnt = 0.f;
  for (int j = 1 + t; j <= K; j += blockDim.x) {
    float v = lg[j];
    se += __expf(v - mx);
    if (v > l0) cnt += 1.f;
  }
  se = wave_sum(se);
  cnt = wave_sum(cnt);
  if ((t & 63) == 0) { sh[t >> 6] = se; sh[8 + (t >> 6)] = cnt; }
  __syncthreads();
  se = 0.f; cnt = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { se += sh[i]; cnt += sh[8 + i]; }
  se += __expf(l0 - mx);
  const float lse = logf(se) + mx;
  const float gs = inv_T / (float)B;
  float* dl = dlogits + (size_t)b * (K + 1);
  for (int j = 1 + t; j <= K; j += blockDim.x) dl[j] = __expf(lg[j] - lse) * gs;
  if (t == 0) {
    lg[0] = l0;
    dl[0] = (__expf(l0 - lse) - 1.f) * gs;
    loss_rows[b] = lse - l0;
    rank0[b] = (int)cnt;
  }
}

// dq[b][:] += dlogits[b][0] * k[b][:]
__global__ void axpy_rows_kernel(const float* __restrict__ dl, int64_t ld, const float* __restrict__ k, int B, int D,
                                 float* __restrict__ dq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D;
  dq[i] += dl[(int64_t)b * ld] * k[i];
}

// shuffle-rank margin: one block (64 threads) per sample; feats [Bn][2s][D] (view-major)
__global__ void rank_margin_kernel(const float* __restrict__ f, int Bn, int s, int D, float theta, float clip,
                                   float weight, float* __restrict__ logits, float* __restrict__ loss_part,
                                   float* __restrict__ df) {
  extern __shared__ float sm[];          // S[n2*n2], dS[n2*n2]
  const int b = blockIdx.x, t = threadIdx.x;
  const int n2 = 2 * s;
  float* S = sm;
  float* dS = sm + n2 * n2;
  const float* fb = f + (size_t)b * n2 * D;
  for (int p = t; p < n2 * n2; p += blockDim.x) {
    const int i = p / n2, j = p % n2;
    float a = 0.f;
    for (int d = 0; d < D; ++d) a += fb[i * D + d] * fb[j * D + d];
    S[p] = a;
    dS[p] = 0.f;
  }
  __syncthreads();
  const float count = (float)Bn * n2 * (n2 - 2);
  float lsum = 0.f;
  if (t < n2) {
    const int i = t, pr = (i + s) % n2;
    const float hi = S[i * n2 + pr];
    float* lg = logits + ((size_t)b * n2 + i) * (n2 - 1);
    lg[0] = hi;
    int o = 1;
    for (int j = 0; j < n2; ++j) {
      if (j == i || j == pr) continue;
      const float lo = S[i * n2 + j];
      lg[o++] = lo;
      float z = (lo - hi) / theta;
      float pass = 1.f;
      if (clip > 0.f && z > clip) { z = clip; pass = 0.f; }
      lsum += log1pf(__expf(z));                    // log(1+exp(z)), z <= 5 with clip; fp32 safe to z~88
      const float sg = 1.f / (1.f + __expf(-z));
      const float dz = weight / count * sg * pass / theta;
      dS[i * n2 + j] += dz;
      dS[i * n2 + pr] -= dz;
    }
  }
  lsum = wave_sum(lsum);
  if (t == 0) loss_part[b] = lsum * weight / count;
  __syncthreads();
  // df[i][d] = sum_j (dS[i][j] + dS[j][i]) f[j][d]
  for (int p = t; p < n2 * D; p += blockDim.x) {
    const int i = p / D, d = p % D;
    float a = 0.f;
    for (int j = 0; j < n2; ++j) a += (dS[i * n2 + j] + dS[j * n2 + i]) * fb[j * D + d];
    df[(size_t)b * n2 * D + p] = a;
  }
}

__global__ void sum_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    out[0] = r * scale;
  }
}

// x [R][G][D] -> y [R][D] mean over G;  bwd: dx[r][g][d] = dy[r][d]/G
__global__ void group_mean_kernel(const float* __restrict__ x, int R, int G, int D, float* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * D) return;
  const int r = i / D, d = i % D;
  float a = 0.f;
  for (int g = 0; g < G; ++g) a += x[((size_t)r * G + g) * D + d];
  y[i] = a / (float)G;
}
__global__ void group_mean_bwd_kernel(const float* __restrict__ dy, int R, int G, int D, float* __restrict__ dx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * G * D) return;
  const int d = i % D, r = i / (G * D);
  dx[i] = dy[(size_t)r * D + d] / (float)G;
}

}  // namespace

#define ST(s) ((hipStream_t)(s))

extern "C" int dv_gemm_f32(int32_t M, int32_t N, int32_t K, const float* A, int64_t sam, int64_t sak, const float* B,
                           int64_t sbk, int64_t sbn, float* C, int64_t ldc, float alpha, int32_t accumulate,
                           void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return DV_EINVAL;
  return launch_gemm_f32(M, N, K, A, sam, sak, B, sbk, sbn, C, ldc, alpha, accumulate, ST(stream));
}

extern "C" int dv_gemm_f32_ex(const dv_gemm_desc* desc_host, void* stream) {
  if (!desc_host || !desc_host->A || !desc_host->B || !desc_host->C || desc_host->M <= 0 || desc_host->N <= 0 || desc_host->K <= 0)
    return DV_EINVAL;
  dv_gemm_desc d = *desc_host;
  const int64_t tiles = (int64_t)((d.M + 31) / 32) * ((d.N + 31) / 32);
  if (tiles > 0x7fffffff) return DV_EINVAL;
  d.tile_end = (int32_t)tiles;
  hipLaunchKernelGGL(gemm_f32_tile4_kernel, dim3((unsigned)tiles), dim3(256), 0, ST(stream), d);
  return dv_launch_status();
}

extern "C" int dv_gemm_f32_grouped(const dv_gemm_desc* descs_dev, int32_t n_groups, int32_t total_tiles, void* stream) {
  if (!descs_dev || n_groups <= 0 || total_tiles <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(gemm_f32_grouped_kernel, dim3(total_tiles), dim3(256), 0, ST(stream), descs_dev, n_groups);
  return dv_launch_status();
}

extern "C" int dv_ntxent_fwd(const float* rows, const float* cols, int32_t R, int32_t n_local, int32_t N, int32_t D,
                             int32_t row_index0, float inv_T, float* logits, float* loss_rows, int32_t* rank0, float* dsim,
                             void* stream) {
  if (!rows || !cols || !logits || !loss_rows || !rank0 || !dsim) return DV_EINVAL;
  if (R <= 0 || N <= 1 || D <= 0 || n_local <= 0 || R % n_local || R / n_local != 2) return DV_EINVAL;
  if (row_index0 < 0 || row_index0 + n_local > N) return DV_EINVAL;
  int rc = launch_gemm_f32(R, 2 * N, D, rows, D, 1, cols, 1, D, dsim, 2 * N, inv_T, 0, ST(stream));
  if (rc) return rc;
  hipLaunchKernelGGL(ntxent_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, ST(stream), dsim, R, n_local, N, row_index0, inv_T,
                     logits, loss_rows, rank0);
  return dv_launch_status();
}

extern "C" int64_t dv_infonce_workspace(int32_t B, int32_t D, int32_t K) {
  return (B > 0 && D > 0 && K > 0) ? gemm_splitk_ws_bytes(B, D, K) : 0;
}

extern "C" int dv_infonce_fwd(const float* q, const float* k, const float* queue, int32_t B, int32_t D, int32_t K,
                              float inv_T, float* logits, float* loss_rows, int32_t* rank0, float* dlogits, float* dq,
                              float* workspace, int64_t workspace_bytes, void* stream) {
  if (!q || !k || !queue || !logits || !loss_rows || !rank0 || !dlogits || !dq || B <= 0 || D <= 0 || K <= 0) return DV_EINVAL;
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 3)) return DV_EALIGN;
  // logits[:,1:] = q . queue * inv_T      (queue is [D][K]: B operand strides (K, 1))
  int rc = launch_gemm_f32(B, K, D, q, D, 1, queue, K, 1, logits + 1, K + 1, inv_T, 0, ST(stream));
  if (rc) return rc;
  hipLaunchKernelGGL(infonce_rows_kernel, dim3(B), dim3(256), 0, ST(stream), q, k, B, D, K, inv_T, logits, loss_rows, rank0,
                     dlogits);
  rc = dv_launch_status();
  if (rc) return rc;
  // dq = dlogits[:,1:] . queue^T + dlogits[:,0] * k
  rc = launch_gemm_f32(B, D, K, dlogits + 1, K + 1, 1, queue, 1, K, dq, D, 1.f, 0, ST(stream), workspace, workspace_bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(axpy_rows_kernel, dim3((B * D + 255) / 256), dim3(256), 0, ST(stream), dlogits, (int64_t)(K + 1), k, B, D, dq);
  return dv_launch_status();
}

// Softmax cross-entropy with integer targets (nn.CrossEntropyLoss, mean reduction: classifier.py:330,465): one wave
// per row.  loss_rows[r] = logsumexp(logits[r]) - logits[r][label]; dlogits = (softmax - onehot) / R; rank0[r] = number
// of classes scoring above the target (top-k accuracy without a sort).
__global__ void softmax_ce_rows_kernel(const float* __restrict__ logits, int ld, int R, int K, const int* __restrict__ labels,
                                       float* __restrict__ loss_rows, float* __restrict__ dlogits, int ldd,
                                       int* __restrict__ rank0) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const float* lg = logits + (size_t)r * ld;
  const int y = labels[r];
  const float ly = lg[y];
  float mx = -INFINITY;
  for (int j = lane; j < K; j += 64) mx = fmaxf(mx, lg[j]);
  mx = wave_max(mx);
  float se = 0.f, cnt = 0.f;
  for (int j = lane; j < K; j += 64) {
    const float v = lg[j];
    se += __expf(v - mx);
    if (v > ly) cnt += 1.f;
  }
  se = wave_sum(se);
  cnt = wave_sum(cnt);
  const float lse = logf(se) + mx;
  const float gs = 1.f / (float)R;
  if (dlogits)
    for (int j = lane; j < K; j += 64) dlogits[(size_t)r * ldd + j] = (__expf(lg[j] - lse) - (j == y ? 1.f : 0.f)) * gs;
  if (lane == 0) {
    loss_rows[r] = lse - ly;
    if (rank0) rank0[r] = (int)cnt;
  }
}

extern "C" int dv_softmax_ce_fwd(const float* logits, int32_t ld, int32_t R, int32_t K, const int32_t* labels, float* loss_rows,
                                 float* dlogits, int32_t ldd, int32_t* rank0, void* stream) {
  if (!logits || !labels || !loss_rows || R <= 0 || K <= 0 || ld < K || (dlogits && ldd < K)) return DV_EINVAL;
  hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, ST(stream), logits, ld, R, K, labels, loss_rows,
                     dlogits, ldd, rank0);
  return dv_launch_status();
}

// Row softmax (F.softmax(logit, dim=-1) of the 10-clip test, classifier.py:716): one wave per row.
__global__ void softmax_rows_kernel(const float* __restrict__ logits, int ld, int R, int K, float* __restrict__ probs, int ldp) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const float* lg = logits + (size_t)r * ld;
  float mx = -INFINITY;
  for (int j = lane; j < K; j += 64) mx = fmaxf(mx, lg[j]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < K; j += 64) se += expf(lg[j] - mx);
  se = wave_sum(se);
  const float inv = 1.f / se;
  for (int j = lane; j < K; j += 64) probs[(size_t)r * ldp + j] = expf(lg[j] - mx) * inv;
}

extern "C" int dv_softmax_rows_f32(const float* logits, int32_t ld, int32_t R, int32_t K, float* probs, int32_t ldp, void* stream) {
  if (!logits || !probs || R <= 0 || K <= 0 || ld < K || ldp < K) return DV_EINVAL;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, ST(stream), logits, ld, R, K, probs, ldp);
  return dv_launch_status();
}

// Nearest-neighbour retrieval score (classifier.py:964-981: topk over sim = test . train^T, hit if any of the k nearest
// train samples carries the test label): per test row the rank of its best same-label train sample, i.e. the number of
// train samples scoring strictly above it; the k-NN accuracy is mean(rank < k) for every k at once, without a sort.
// rank = n_train when no train sample has the label.  One wave per row.
__global__ void knn_rank_rows_kernel(const float* __restrict__ sim, int ld, int R, int Nt, const int* __restrict__ train_labels,
                                     const int* __restrict__ test_labels, int* __restrict__ rank) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const float* row = sim + (size_t)r * ld;
  const int y = test_labels[r];
  float best = -INFINITY;
  for (int j = lane; j < Nt; j += 64)
    if (train_labels[j] == y) best = fmaxf(best, row[j]);
  best = wave_max(best);
  float cnt = 0.f;
  if (best > -INFINITY) {
    for (int j = lane; j < Nt; j += 64) cnt += row[j] > best ? 1.f : 0.f;
    cnt = wave_sum(cnt);
  } else {
    cnt = (float)Nt;
  }
  if (lane == 0) rank[r] = (int)cnt;
}

extern "C" int dv_knn_rank(const float* sim, int32_t ld, int32_t R, int32_t n_train, const int32_t* train_labels,
                           const int32_t* test_labels, int32_t* rank, void* stream) {
  if (!sim || !train_labels || !test_labels || !rank || R <= 0 || n_train <= 0 || ld < n_train) return DV_EINVAL;
  hipLaunchKernelGGL(knn_rank_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, ST(stream), sim, ld, R, n_train, train_labels,
                     test_labels, rank);
  return dv_launch_status();
}

extern "C" int dv_rank_margin(const float* feats, int32_t Bn, int32_t s, int32_t D, float theta, float clip, float weight,
                              float* logits, float* loss, float* dfeats, float* scratch /*[Bn]*/, void* stream) {
  if (!feats || !logits || !loss || !dfeats || !scratch || Bn <= 0 || s < 2 || s > 8 || D <= 0 || theta <= 0.f) return DV_EINVAL;
  const int n2 = 2 * s;
  hipLaunchKernelGGL(rank_margin_kernel, dim3(Bn), dim3(64), 2 * n2 * n2 * sizeof(float), ST(stream), feats, Bn, s, D, theta,
                     clip, weight, logits, scratch, dfeats);
  int rc = dv_launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, ST(stream), scratch, Bn, 1.f, loss);
  return dv_launch_status();
}

extern "C" int dv_group_mean_f32(const float* x, int32_t R, int32_t G, int32_t D, float* y, void* stream) {
  if (!x || !y || R <= 0 || G <= 0 || D <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(group_mean_kernel, dim3((R * D + 255) / 256), dim3(256), 0, ST(stream), x, R, G, D, y);
  return dv_launch_status();
}
extern "C" int dv_group_mean_bwd_f32(const float* dy, int32_t R, int32_t G, int32_t D, float* dx, void* stream) {
  if (!dy || !dx || R <= 0 || G <= 0 || D <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(group_mean_bwd_kernel, dim3((R * G * D + 255) / 256), dim3(256), 0, ST(stream), dy, R, G, D, dx);
  return dv_launch_status();
}
