// Streaming (HBM-bound) kernels of the DualVar hot path on gfx950: ingest, BatchNorm statistics /
// apply / backward, MaxPool3d, spatial mean, self-gating scale, optimizer.  NDHWC, 16-byte vector
// accesses along the channel axis, fp32 math, wavefront(64)-shuffle + LDS reductions.
#include "common.hpp"
#include <cstdlib>

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
                              const int* perm, int n_seg, int pad, int Hp, int Wp) {
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
    // destination pixel inside the (optionally zero-bordered) [N*T][Hp][Wp] frame
    const int hh = (int)(hw / W), ww = (int)(hw - (int64_t)hh * W);
    T* dst = y + ((nt * Hp + hh + pad) * (int64_t)Wp + ww + pad) * ldy;
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
__device__ __forceinline__ float block_sum(float v, float* sh /*>= blockDim/64 floats*/) {
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
// one block per channel: partials [2][P][tiles] (a channel's tiles contiguous: coalesced) -> local [2C+1]; with `fin` set (single rank) the block also
// finalises the channel (mean / invstd / scale / shift / running statistics), saving a launch per BN layer
struct BnFinalize {
  const float* gamma; const float* beta; float eps, momentum;
  float* running_mean; float* running_var; float* mean; float* invstd; float* scale; float* shift;
};
__device__ __forceinline__ void bn_reduce_stats_body(const float* __restrict__ part, int n_tiles, int tile_rows, int P,
                                                     int64_t M, int C, float* __restrict__ out, int fin,
                                                     const BnFinalize& f, int c) {
  __shared__ float sh[32];
  float s = 0.f;
  const float* ps = part + (size_t)c * n_tiles;                 // sums of channel c, one per tile
  const float* pq = part + ((size_t)P + c) * n_tiles;           // M2 about the tile mean
  for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) s += ps[i];
  const float S = block_sum(s, sh);
  const float mean = S / (float)M;
  float m2 = 0.f;
  for (int i = threadIdx.x; i < n_tiles; i += blockDim.x) {
    int64_t left = M - (int64_t)i * tile_rows;
    float n_i = (float)(left < tile_rows ? left : tile_rows);
    float d = ps[i] / n_i - mean;
    m2 += pq[i] + n_i * d * d;
  }
  const float M2 = block_sum(m2, sh + 16);
  if (threadIdx.x == 0) {
    out[c] = S;
    out[C + c] = M2;
    if (c == 0) out[2 * C] = (float)M;
    if (fin) {
      const float cnt = (float)M;
      const float var = M2 / cnt;
      const float invstd = rsqrtf(var + f.eps);
      f.mean[c] = mean;
      f.invstd[c] = invstd;
      const float sc = f.gamma[c] * invstd;
      f.scale[c] = sc;
      f.shift[c] = f.beta[c] - mean * sc;
      if (f.running_mean) {
        f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
        const float unbiased = cnt > 1.f ? M2 / (cnt - 1.f) : var;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unbiased;
      }
    }
  }
}

__global__ void bn_reduce_stats_kernel(const float* __restrict__ part, int n_tiles, int tile_rows, int P, int64_t M, int C,
                                       float* __restrict__ out, int fin, BnFinalize f) {
  bn_reduce_stats_body(part, n_tiles, tile_rows, P, M, C, out, fin, f, blockIdx.x);
}

// locate the item a block of a multi-tensor launch belongs to: `end` is the running block-count prefix
// (the first eight prefixes are fetched with independent loads: a `while` over them is a chain of dependent global
// loads, ~1 us each, in front of every block of these latency-bound launches)
template <typename GetEnd>
__device__ __forceinline__ int find_item(int n, uint32_t& bid, uint32_t& nblk, GetEnd end) {
  uint32_t e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = k < n ? (uint32_t)end(k) : 0xffffffffu;
  int i = 0;
  uint32_t start = 0, endv = e[0];
#pragma unroll
  for (int k = 0; k < 7; ++k)
    if (k < n - 1 && bid >= e[k]) { start = e[k]; i = k + 1; endv = e[k + 1]; }
  if (n > 8 && i == 7) {
    while (i < n - 1 && bid >= (uint32_t)end(i)) { start = (uint32_t)end(i); ++i; }
    endv = (uint32_t)end(i);
  }
  nblk = endv - start;
  bid -= start;
  return i;
}

__global__ void bn_stats_multi_kernel(const dv_bn_item* __restrict__ items, int n, int fin) {
  uint32_t bid = blockIdx.x, nblk;
  const int i = find_item(n, bid, nblk, [&](int k) { return items[k].blk_stats; });
  const dv_bn_item& it = items[i];
  BnFinalize f = {it.gamma, it.beta, it.eps, it.momentum, it.running_mean, it.running_var, it.mean, it.invstd, it.scale, it.shift};
  bn_reduce_stats_body(it.partials, it.n_tiles, it.tile_rows, it.pitch, it.M, it.C, it.local_stats, fin, f, (int)bid);
}

__global__ void bn_finalize_kernel(const float* __restrict__ stats, int R, int stride, int C, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean_o, float* invstd_o, float* scale_o,
                                   float* shift_o) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
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

// the members of a group in ONE launch: item i's rows sit in the gathered [R][stride] table at the offset its
// local_stats has inside the group's local row; 128 channels per block, blocks of item i follow those of item i-1
__global__ void bn_finalize_multi_kernel(const dv_bn_item* __restrict__ items, int n, const float* __restrict__ local_base,
                                         const float* __restrict__ gathered, int R, int stride) {
  int bid = blockIdx.x, i = 0;
  while (i < n - 1 && bid >= (items[i].C + 127) / 128) { bid -= (items[i].C + 127) / 128; ++i; }
  const dv_bn_item& it = items[i];
  const int c = bid * 128 + threadIdx.x, C = it.C;
  if (c >= C) return;
  const float* stats = gathered + (it.local_stats - local_base);
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
  const float invstd = rsqrtf(var + it.eps);
  it.mean[c] = mean;
  it.invstd[c] = invstd;
  const float sc = it.gamma[c] * invstd;
  it.scale[c] = sc;
  it.shift[c] = it.beta[c] - mean * sc;
  if (it.running_mean) {
    it.running_mean[c] = (1.f - it.momentum) * it.running_mean[c] + it.momentum * mean;
    const float unbiased = cnt > 1.f ? M2 / (cnt - 1.f) : var;
    it.running_var[c] = (1.f - it.momentum) * it.running_var[c] + it.momentum * unbiased;
  }
}

// ------------------------------------------------------------------ BN apply (+residual) (+ReLU)
// i (vector index) -> (row, first channel) with one mulhi; parameters as 16-byte vector loads (the per-channel
// arrays are padded to a multiple of 8 floats)
template <int V>
__device__ __forceinline__ void load_params(const float* __restrict__ p, int c0, float (&v)[V]) {
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p + c0 + 4 * q);
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}

template <typename T>
__device__ __forceinline__ void bn_apply_body(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                              const float* __restrict__ shift, const T* __restrict__ res, int ldr,
                                              T* __restrict__ y, int ldy, uint32_t total, int C, const FastDiv& fcv,
                                              int flags, uint32_t bid, uint32_t nblk) {
  constexpr int V = DT<T>::VEC;
  const uint32_t CV = fcv.d;
  for (uint32_t i = bid * blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
    const uint32_t row = fd_div(i, fcv);
    const int c0 = (int)(i - row * CV) * V;
    float v[V], r[V], sc[V], sh[V];
    Pack16<T>::load(x + (size_t)row * ldx + c0, v);
    if (res) Pack16<T>::load(res + (size_t)row * ldr + c0, r);
    load_params<V>(scale, c0, sc);
    load_params<V>(shift, c0, sh);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float o = v[e] * sc[e] + sh[e];
      if (res) o += r[e];
      if (flags & DV_RELU) o = fmaxf(o, 0.f);
      v[e] = (c0 + e < C) ? o : 0.f;
    }
    Pack16<T>::store(y + (size_t)row * ldy + c0, v);
  }
}

template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
                                const float* __restrict__ shift, const T* __restrict__ res, int ldr,
                                T* __restrict__ y, int ldy, uint32_t total, int C, FastDiv fcv, int flags) {
  bn_apply_body<T>(x, ldx, scale, shift, res, ldr, y, ldy, total, C, fcv, flags, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ FastDiv fastdiv_dev(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t sft = 0;
  while ((1u << sft) < d) ++sft;
  f.shr = sft;
  f.mul = (uint32_t)((((1ull << sft) - d) << 32) / d + 1);
  f._pad = 0;
  return f;
}

template <typename T>
__global__ void bn_apply_multi_kernel(const dv_bn_item* __restrict__ items, int n) {
  uint32_t bid = blockIdx.x, nblk;
  const int i = find_item(n, bid, nblk, [&](int k) { return items[k].blk_apply; });
  const dv_bn_item& it = items[i];
  constexpr int V = DT<T>::VEC;
  const int CP = (it.C + 7) & ~7;
  bn_apply_body<T>((const T*)it.x, it.ldx, it.scale, it.shift, (const T*)it.residual, it.ldr, (T*)it.y, it.ldy,
                   (uint32_t)(it.M * (CP / V)), it.C, fastdiv_dev((uint32_t)(CP / V)), it.fwd_flags, bid, nblk);
}

// ------------------------------------------------------------------ column reductions over rows
// Generic: rows [r_begin, r_end) of a [rows][ld] matrix, lanes along channel vectors.  F maps
// (row, c0) -> V values for NS sums.  Result: out[s][c] for this block (written by the caller's lambda).
struct NoPre { __device__ __forceinline__ void operator()(int) const {} };

template <int V, int NS, typename F, typename W, typename Pre = NoPre>
__device__ __forceinline__ void column_reduce(int64_t r_begin, int64_t r_end, int CP, F f, W write, Pre pre = Pre()) {
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
    if (my_rg < rg) {
      pre((cvb + my_cv) * V);                       // per-thread constants of this channel vector
      // four rows in flight for 4-element (fp32) vectors, two for 8-element (bf16) ones: the same 16 values per sum and thread.
      // (unroll 4 for bf16 needed > 128 VGPRs: with the 1 024-thread default bound the reduce kernels spilled 112 B per lane and
      // their traffic went 1.08x -> 1.55x algorithmic -- round 3's bf16 regression; tests/test_abi_and_host.py now fails on
      // any kernel with scratch.)
#pragma unroll(V == 4 ? 4 : 2)
      for (int64_t r = r_begin + my_rg; r < r_end; r += rg) f(r, (cvb + my_cv) * V, acc);
    }
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
      // fold the row groups pairwise with every thread (a serial fold by the first cvc threads costs ~10 us
      // when the tensor has few channels and therefore up to 128 row groups)
      for (int n = rg; n > 1;) {
        const int half = (n + 1) >> 1;
        if (my_rg < rg && my_rg + half < n) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < V; ++e)
              lds[((my_rg * NS + s) * cvc + my_cv) * V + e] += lds[(((my_rg + half) * NS + s) * cvc + my_cv) * V + e];
        }
        __syncthreads();
        n = half;
      }
      if (t < cvc) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int e = 0; e < V; ++e) acc[s][e] = lds[(s * cvc + t) * V + e];
        write((cvb + t) * V, acc);
      }
    }
    __syncthreads();
  }
}

// Tail of an ordered reduction over the nblk blocks of one item, each of which has stored NO = 2*CP partial sums to
// ws[bid][NO].  Two levels, so that no single block has to walk hundreds of partial rows: the block that takes the last
// ticket of its group of kFoldGroup blocks adds the group's rows in block order into a group row; the block that takes the
// last group ticket adds the group rows in group order and WRITES out[NO].  The hand-off (rows -> ticket -> reader) is the
// sc1 / s_waitcnt form described below, not agent-scope fences.  Layout of ws: [nblk][NO] rows, [ngrp][NO] group rows,
// [ngrp + 1] ticket words (zero on entry, zero again on exit; the host re-zeroes them after a failed launch:
// dualvar_amd/_lib.py register_ticket_workspace).
constexpr int kFoldGroup = 32;
static inline int64_t ordered_fold_floats(int nblk, int CP) {
  const int ngrp = (nblk + kFoldGroup - 1) / kFoldGroup;
  return (int64_t)(nblk + ngrp) * 2 * CP + ((ngrp + 1 + 7) & ~7);
}
// Rows and tickets travel as agent-scope relaxed atomics (stores / loads with sc1: coherent across the XCDs' L2s on their own),
// ordered by a plain s_waitcnt -- NOT by agent-scope fences: a release fence is a write-back of the whole L2 (buffer_wbl2) and
// an acquire fence an invalidate, from every one of the thousands of blocks of a launch; measured, they took the reduce
// launches of the S3D-G step from 1.4 to 4.0 - 7.9 ms.
// The protocol, spelled out: (1) EVERY store of a handed-off row is an sc1 (write-through, agent-coherent) store; (2) every
// storing wave drains them with s_waitcnt vmcnt(0), then the workgroup's barrier; (3) ONE lane takes the ticket with an
// agent-scope atomic add; (4) only the workgroup whose add returned the last ticket reads, after a barrier its ticket lane
// joins, and EVERY load of the rows is an sc1 load to registers (never cached in the reading CU's L1).  This is the
// "sc1 stores + drained counter + sc1 loads" hand-off measured valid on gfx950 (MI355X_MICROARCH.md, inter-workgroup
// visibility, valid forms); it is NOT a HIP memory-model guarantee and relies on stores retiring under vmcnt on this part.
// Hence the guard: the file refuses to build for any other target (the library's host side refuses to run elsewhere too:
// dv_check_device).  A port would replace coherent_store / stores_done / coherent_load by plain accesses between
// __builtin_amdgcn_fence(__ATOMIC_RELEASE / __ATOMIC_ACQUIRE, "agent").
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "ordered_fold's sc1 / s_waitcnt hand-off is validated on gfx950 only (see the comment above)"
#endif
__device__ __forceinline__ void coherent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float coherent_load(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void fold_rows(const float* __restrict__ rows, int n, int NO, int C, int CP, float* __restrict__ dst,
                                          bool coherent_dst) {
  // all the loads of a batch are in flight together (one trip to the coherence point per kFoldBatch rows, not per row); a
  // short batch re-reads its last row and adds nothing for it, so the order of the additions is the row order
  constexpr int kFoldBatch = 32;
  for (int i = threadIdx.x; i < NO; i += blockDim.x) {
    const int c = i < CP ? i : i - CP;
    float t = 0.f;
    if (c < C) {
      for (int b = 0; b < n; b += kFoldBatch) {
        float v[kFoldBatch];
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) v[u] = coherent_load(rows + (size_t)min(b + u, n - 1) * NO + i);
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) t += (b + u < n) ? v[u] : 0.f;
      }
    }
    if (coherent_dst) coherent_store(dst + i, t);
    else dst[i] = t;
  }
}
__device__ __forceinline__ void ordered_fold(float* __restrict__ ws, uint32_t bid, uint32_t nblk, int C, int CP, float* __restrict__ out) {
  __shared__ uint32_t last;
  const int NO = 2 * CP;
  constexpr uint32_t gfold = kFoldGroup;
  const uint32_t ngrp = (nblk + gfold - 1) / gfold, grp = bid / gfold;
  const uint32_t gsize = min(gfold, nblk - grp * gfold);
  float* grows = ws + (size_t)nblk * NO;
  uint32_t* tick = reinterpret_cast<uint32_t*>(grows + (size_t)ngrp * NO);
  stores_done();                                     // this block's row has reached the coherence point ...
  __syncthreads();
  if (threadIdx.x == 0) last = (atomicAdd(tick + grp, 1u) == gsize - 1) ? 1u : 0u;       // ... before its ticket is taken
  __syncthreads();
  if (!last) return;
  if (ngrp == 1) {                                   // one group: its last block writes the result (0 + the group row: same bits)
    fold_rows(ws, (int)gsize, NO, C, CP, out, false);
    if (threadIdx.x == 0) __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  fold_rows(ws + (size_t)grp * gfold * NO, (int)gsize, NO, C, CP, grows + (size_t)grp * NO, true);
  if (threadIdx.x == 0) __hip_atomic_store(tick + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  stores_done();
  __syncthreads();
  if (threadIdx.x == 0) last = (atomicAdd(tick + ngrp, 1u) == ngrp - 1) ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  fold_rows(grows, (int)ngrp, NO, C, CP, out, false);
  if (threadIdx.x == 0) __hip_atomic_store(tick + ngrp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T>
__device__ __forceinline__ void bn_bwd_reduce_body(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                                   const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, int64_t M, int C, int CP, int flags,
                                                   int64_t rows_per_block, float* __restrict__ sums_all, int n_rep,
                                                   uint32_t bid, const float* __restrict__ scale = nullptr,
                                                   const float* __restrict__ shift = nullptr, float* __restrict__ ws = nullptr,
                                                   uint32_t nblk = 0) {
  constexpr int V = DT<T>::VEC;
  // Ordered mode (ws != nullptr): the block stores its partial sums to ws[bid][2][CP] (sc1 stores); the block that takes the
  // last ticket (ordered_fold: drained stores -> ticket -> sc1 loads) adds the partials in block order into sums_all[0] -- no
  // float atomics, so the sums do not depend on the order blocks finish in.  Legacy mode: atomics, spread over n_rep replicas because atomics on one
  // address serialise at the memory side (~12 ns each).
  float* sums = ws ? ws + (size_t)bid * 2 * CP : sums_all + (size_t)(bid % n_rep) * 2 * CP;
  const int64_t r0 = (int64_t)bid * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  const bool mask = !(flags & DV_NO_RELU_MASK);
  // DV_MASK_FROM_X: the ReLU mask is recomputed from x exactly as the forward computed y = relu(x*scale + shift)
  // (same expression, same rounding), so the output tensor is not read at all: 2 tensor reads instead of 3
  const bool fromx = mask && (flags & DV_MASK_FROM_X);
  float mu[V], is[V], sc[V], sh[V];
  column_reduce<V, 2>(
      r0, r1, CP,
      [&](int64_t r, int c0, float(&acc)[2][V]) {
        float g[V], yy[V], xx[V];
        Pack16<T>::load(dy + r * lddy + c0, g);
        if (mask && !fromx) Pack16<T>::load(y + r * ldy + c0, yy);
        Pack16<T>::load(x + r * ldx + c0, xx);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float act = fromx ? xx[e] * sc[e] + sh[e] : yy[e];
          float gg = (mask && !(act > 0.f)) ? 0.f : g[e];
          acc[0][e] += gg;
          acc[1][e] += gg * (xx[e] - mu[e]) * is[e];
        }
      },
      [&](int c0, float(&acc)[2][V]) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (c0 + e < C) {
            if (ws) { coherent_store(sums + c0 + e, acc[0][e]); coherent_store(sums + CP + c0 + e, acc[1][e]); }
            else { atomicAdd(sums + c0 + e, acc[0][e]); atomicAdd(sums + CP + c0 + e, acc[1][e]); }
          }
      },
      [&](int c0) {
        load_params<V>(mean, c0, mu);
        load_params<V>(invstd, c0, is);
        if (fromx) { load_params<V>(scale, c0, sc); load_params<V>(shift, c0, sh); }
      });
  if (ws) ordered_fold(ws, bid, nblk, C, CP, sums_all);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                     const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, int64_t M, int C, int CP, int flags,
                                     int64_t rows_per_block, float* __restrict__ sums_all, int n_rep, float* __restrict__ ws) {
  bn_bwd_reduce_body<T>(dy, lddy, y, ldy, x, ldx, mean, invstd, M, C, CP, flags, rows_per_block, sums_all, n_rep, blockIdx.x,
                        nullptr, nullptr, ws, gridDim.x);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_multi_kernel(const dv_bn_item* __restrict__ items, int n) {
  uint32_t bid = blockIdx.x, nblk;
  const int i = find_item(n, bid, nblk, [&](int k) { return items[k].blk_red; });
  const dv_bn_item& it = items[i];
  const int CP = (it.C + 7) & ~7;
  const int64_t rpb = (it.M + nblk - 1) / nblk;
  bn_bwd_reduce_body<T>((const T*)it.dy, it.lddy, (const T*)it.y, it.ldy, (const T*)it.x, it.ldx, it.mean, it.invstd, it.M,
                        it.C, CP, it.bwd_flags, rpb, it.sums, it.n_rep, bid, it.scale, it.shift, it.red_ws, nblk);
}

// partials [n_blocks][W] -> out[W] (+=): 32 columns x 8 row lanes per block
__global__ void reduce_rows_kernel(const float* __restrict__ part, int64_t ld, int n_rows, int Wd, float* __restrict__ out,
                                   int accumulate) {
  __shared__ float sh[8][33];
  const int col = blockIdx.x * 32 + (threadIdx.x & 31);
  const int rl = threadIdx.x >> 5;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < Wd) {
    int r = rl;
    for (; r + 24 < n_rows; r += 32) {
      a0 += part[(size_t)r * ld + col];
      a1 += part[(size_t)(r + 8) * ld + col];
      a2 += part[(size_t)(r + 16) * ld + col];
      a3 += part[(size_t)(r + 24) * ld + col];
    }
    for (; r < n_rows; r += 8) a0 += part[(size_t)r * ld + col];
  }
  sh[rl][threadIdx.x & 31] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && col < Wd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += sh[i][threadIdx.x & 31];
    out[col] = accumulate ? out[col] + s : s;
  }
}

// dx = k1[c]*g + k2[c]*x + k3[c] with  k1 = gamma*invstd, k2 = -k1*invstd*sgx/M, k3 = -k1*sg/M - k2*mean
// (sg, sgx = global sums of g and g*xhat).  The table is built once per block in LDS.
template <typename T>
__device__ __forceinline__ void bn_bwd_apply_body(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                                  const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ sums_g, int rep_g, float inv_count,
                                                  float dscale, float* dgamma, float* dbeta, T* __restrict__ dx,
                                                  int lddx, T* __restrict__ dres, int lddres, uint32_t total, int C,
                                                  int CP, const FastDiv& fcv, int flags, uint32_t bid, uint32_t nblk,
                                                  const float* __restrict__ scale = nullptr,
                                                  const float* __restrict__ shift = nullptr) {
  constexpr int V = DT<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float coef[];      // [3][CP] (+ [2][CP] scale, shift with DV_MASK_FROM_X)
  const bool fromx = !(flags & DV_NO_RELU_MASK) && (flags & DV_MASK_FROM_X);
  if (bid == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float sb = 0.f, sg = 0.f;
      for (int r = 0; r < rep_g; ++r) { sb += sums_g[(size_t)r * 2 * CP + c]; sg += sums_g[(size_t)r * 2 * CP + CP + c]; }
      dbeta[c] += dscale * sb;
      dgamma[c] += dscale * sg;
    }
  }
  for (int c = threadIdx.x; c < CP; c += blockDim.x) {
    float k1 = 0.f, k2 = 0.f, k3 = 0.f;
    if (c < C) {
      float sg = 0.f, sgx = 0.f;
      for (int r = 0; r < rep_g; ++r) { sg += sums_g[(size_t)r * 2 * CP + c]; sgx += sums_g[(size_t)r * 2 * CP + CP + c]; }
      k1 = gamma[c] * invstd[c];
      k2 = -k1 * invstd[c] * sgx * inv_count;
      k3 = -k1 * sg * inv_count - k2 * mean[c];
    }
    coef[c] = k1; coef[CP + c] = k2; coef[2 * CP + c] = k3;
    if (fromx) { coef[3 * CP + c] = c < C ? scale[c] : 0.f; coef[4 * CP + c] = c < C ? shift[c] : 0.f; }
  }
  __syncthreads();
  const uint32_t CV = fcv.d;
  const bool mask = !(flags & DV_NO_RELU_MASK);
  for (uint32_t i = bid * blockDim.x + threadIdx.x; i < total; i += nblk * blockDim.x) {
    const uint32_t row = fd_div(i, fcv);
    const int c0 = (int)(i - row * CV) * V;
    float g[V], yy[V], xx[V], o[V], ro[V], k1[V], k2[V], k3[V], sc[V], sh[V];
    Pack16<T>::load(dy + (size_t)row * lddy + c0, g);
    if (mask && !fromx) Pack16<T>::load(y + (size_t)row * ldy + c0, yy);
    Pack16<T>::load(x + (size_t)row * ldx + c0, xx);
    if (dres && (flags & DV_ACCUM)) Pack16<T>::load(dres + (size_t)row * lddres + c0, ro);
    load_params<V>(coef, c0, k1);
    load_params<V>(coef + CP, c0, k2);
    load_params<V>(coef + 2 * CP, c0, k3);
    if (fromx) { load_params<V>(coef + 3 * CP, c0, sc); load_params<V>(coef + 4 * CP, c0, sh); }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float act = fromx ? xx[e] * sc[e] + sh[e] : yy[e];     // the forward's expression: same mask bit for bit
      const float gg = (mask && !(act > 0.f)) ? 0.f : g[e];
      o[e] = k1[e] * gg + k2[e] * xx[e] + k3[e];
      if (dres) ro[e] = (flags & DV_ACCUM) ? ro[e] + gg : gg;
    }
    Pack16<T>::store(dx + (size_t)row * lddx + c0, o);
    if (dres) Pack16<T>::store(dres + (size_t)row * lddres + c0, ro);
  }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                    const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sums_g, int rep_g, float inv_count, float dscale,
                                    float* dgamma, float* dbeta, T* __restrict__ dx, int lddx,
                                    T* __restrict__ dres, int lddres, uint32_t total, int C, int CP, FastDiv fcv,
                                    int flags) {
  bn_bwd_apply_body<T>(dy, lddy, y, ldy, x, ldx, mean, invstd, gamma, sums_g, rep_g, inv_count, dscale, dgamma, dbeta, dx,
                       lddx, dres, lddres, total, C, CP, fcv, flags, blockIdx.x, gridDim.x);
}

template <typename T>
__global__ void bn_bwd_apply_multi_kernel(const dv_bn_item* __restrict__ items, int n) {
  uint32_t bid = blockIdx.x, nblk;
  const int i = find_item(n, bid, nblk, [&](int k) { return items[k].blk_bapply; });
  const dv_bn_item& it = items[i];
  constexpr int V = DT<T>::VEC;
  const int CP = (it.C + 7) & ~7;
  bn_bwd_apply_body<T>((const T*)it.dy, it.lddy, (const T*)it.y, it.ldy, (const T*)it.x, it.ldx, it.mean, it.invstd,
                       it.gamma, it.sums, it.n_rep, it.inv_count, it.dparam_scale, it.dgamma, it.dbeta, (T*)it.dx, it.lddx,
                       (T*)it.dres, it.lddres, (uint32_t)(it.M * (CP / V)), it.C, CP, fastdiv_dev((uint32_t)(CP / V)),
                       it.bwd_flags, bid, nblk, it.scale, it.shift);
}

// ------------------------------------------------------------------ MaxPool3d
// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  With a grid that is a multiple of 8, this gives
// XCD x the logical blocks [x * grid/8, (x+1) * grid/8): neighbouring windows share an L2 instead of each XCD fetching
// every plane.
__device__ __forceinline__ uint32_t xcd_block() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }
static inline int grid8_for(int64_t work_items, int max_blocks) { return (grid_for(work_items, max_blocks) + 7) & ~7; }

struct PoolArgs {
  int N, Ti, Hi, Wi, C, CP;
  int To, Ho, Wo;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  int ldx, ldy;
  FastDiv fcv, fWo, fHo, fTo, fWi, fHi, fTi;
};

template <typename T>
__global__ void maxpool_fwd_kernel(PoolArgs a, const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx) {
  constexpr int V = DT<T>::VEC;
  const uint32_t CV = a.fcv.d;
  const uint32_t total = (uint32_t)a.N * a.To * a.Ho * a.Wo * CV;
  for (uint32_t i = xcd_block() * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t m, cvi, q, wo_, ho_, to_, n_;
    fd_divmod(i, a.fcv, m, cvi);
    const int c0 = (int)cvi * V;
    fd_divmod(m, a.fWo, q, wo_);
    fd_divmod(q, a.fHo, q, ho_);
    fd_divmod(q, a.fTo, n_, to_);
    const int wo = (int)wo_, ho = (int)ho_, to = (int)to_, n = (int)n_;
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
    Pack16<T>::store(y + (size_t)m * a.ldy + c0, best);
    uint8_t* ip = idx + (size_t)m * a.CP + c0;
#pragma unroll
    for (int e = 0; e < V; ++e) ip[e] = (uint8_t)bi[e];
  }
}

template <typename T>
__device__ __forceinline__ void pool_gather(const PoolArgs& a, const T* __restrict__ dyp, const uint8_t* __restrict__ idx,
                                            uint32_t m_in, int c0, float (&g)[DT<T>::VEC]);

template <typename T>
__global__ void maxpool_bwd_kernel(PoolArgs a, const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                   T* __restrict__ dx, int accumulate) {
  constexpr int V = DT<T>::VEC;
  const uint32_t CV = a.fcv.d;
  const uint32_t total = (uint32_t)a.N * a.Ti * a.Hi * a.Wi * CV;
  for (uint32_t i = xcd_block() * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t m, cvi;
    fd_divmod(i, a.fcv, m, cvi);
    const int c0 = (int)cvi * V;
    float acc[V];
    if (accumulate) Pack16<T>::load(dx + (size_t)m * a.ldx + c0, acc);
    else {
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
    }
    pool_gather<T>(a, dy, idx, m, c0, acc);
    Pack16<T>::store(dx + (size_t)m * a.ldx + c0, acc);
  }
}

// Backward of the pools with a 3x3 window, stride 2, padding 1 in (h, w) -- the stem pools (s3dg.py:139,145, 1x3x3 / 1x2x2;
// resnet_2d3d.py:130) and MaxPool_4a (3x3x3 / 2x2x2) -- with any window along t.  A thread owns a 2x2 QUAD of input pixels:
// together they are covered by the four windows (hq, wq) .. (hq+1, wq+1) only, so a thread loads four (dy, idx) vectors per
// t-window and applies them to its pixels in nine compares, the same count for every thread -- the per-pixel gather has 1, 2,
// 2 or 4 windows depending on the parity of the pixel (divergent trip counts) and loads every window vector four times.
// Same sums, same (ascending tap) order as pool_gather.
template <typename T>
__global__ void maxpool_bwd_quad_kernel(PoolArgs a, FastDiv fWq, FastDiv fHq, const T* __restrict__ dy,
                                        const uint8_t* __restrict__ idx, T* __restrict__ dx, int accumulate) {
  constexpr int V = DT<T>::VEC;
  const uint32_t CV = a.fcv.d;
  const uint32_t total = (uint32_t)a.N * a.Ti * fHq.d * fWq.d * CV;
  const int st1 = a.st - 1;
  for (uint32_t i = xcd_block() * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t q, cvi, wq_, hq_, ti_, n_;
    fd_divmod(i, a.fcv, q, cvi);
    fd_divmod(q, fWq, q, wq_);
    fd_divmod(q, fHq, q, hq_);
    fd_divmod(q, a.fTi, n_, ti_);
    const int c0 = (int)cvi * V, wq = (int)wq_, hq = (int)hq_, ti = (int)ti_, n = (int)n_;
    float acc[2][2][V];
    bool pix[2][2];
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        pix[pa][pb] = 2 * hq + pa < a.Hi && 2 * wq + pb < a.Wi;
        const size_t m = ((size_t)(n * a.Ti + ti) * a.Hi + 2 * hq + pa) * a.Wi + 2 * wq + pb;
        if (accumulate && pix[pa][pb]) Pack16<T>::load(dx + m * a.ldx + c0, acc[pa][pb]);
        else {
#pragma unroll
          for (int e = 0; e < V; ++e) acc[pa][pb][e] = 0.f;
        }
      }
    const int tn = ti + a.pt;
    const int t_hi = min(tn >> st1, a.To - 1), t_lo = max((tn - a.kt + 1 + st1) >> st1, 0);
    for (int to = t_hi; to >= t_lo; --to) {
      const int dt = tn - to * a.st;
      float g[2][2][V];
      uint32_t iw[2][2][V / 4];
#pragma unroll
      for (int wy = 0; wy < 2; ++wy)
#pragma unroll
        for (int wx = 0; wx < 2; ++wx) {
          const bool ok = hq + wy < a.Ho && wq + wx < a.Wo;
          const int64_t mo = (int64_t)((n * a.To + to) * a.Ho + hq + wy) * a.Wo + wq + wx;
          if (ok) {
            Pack16<T>::load(dy + mo * a.ldy + c0, g[wy][wx]);
#pragma unroll
            for (int j = 0; j < V / 4; ++j) iw[wy][wx][j] = reinterpret_cast<const uint32_t*>(idx + mo * a.CP + c0)[j];
          } else {
#pragma unroll
            for (int e = 0; e < V; ++e) g[wy][wx][e] = 0.f;
#pragma unroll
            for (int j = 0; j < V / 4; ++j) iw[wy][wx][j] = 0xffffffffu;
          }
        }
      // pixel (pa, pb) lies in window (wy, wx) iff wy <= pa and wx <= pb, at tap (dh, dw) = (pa + 1 - 2 wy, pb + 1 - 2 wx)
#pragma unroll
      for (int pa = 0; pa < 2; ++pa)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
          for (int wy = 1; wy >= 0; --wy)
#pragma unroll
            for (int wx = 1; wx >= 0; --wx) {
              if (wy > pa || wx > pb) continue;
              const uint32_t tap = (uint32_t)((dt * 3 + pa + 1 - 2 * wy) * 3 + pb + 1 - 2 * wx);
#pragma unroll
              for (int e = 0; e < V; ++e)
                if (((iw[wy][wx][e >> 2] >> (8 * (e & 3))) & 0xffu) == tap) acc[pa][pb][e] += g[wy][wx][e];
            }
    }
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        if (!pix[pa][pb]) continue;
        const size_t m = ((size_t)(n * a.Ti + ti) * a.Hi + 2 * hq + pa) * a.Wi + 2 * wq + pb;
        Pack16<T>::store(dx + m * a.ldx + c0, acc[pa][pb]);
      }
  }
}

// ------------------------------------------------------------------ BatchNorm + ReLU + MaxPool3d as one pass each way
// A pool that is the ONLY consumer of y = relu(x*scale + shift) (the stems: s3dg.py:138-151, resnet_2d3d.py:128-131) never
// needs y in memory: the backward of that BatchNorm rebuilds the ReLU mask from x (DV_MASK_FROM_X) and the pool's own
// backward only needs the tap indices.  Forward: the window elements are normalised, rectified and rounded to the storage
// type on the fly, exactly the values bn_apply would have stored (same expression, same rounding), so pooled values and
// indices are bit-identical to the two-pass form.  Backward: g = dL/dy of an input element is the gather of the pooled
// gradients through idx (maxpool_bwd's inner loop), evaluated inside the reduce and the apply pass of the BatchNorm
// backward instead of being written and read back twice.  Saved per stem pool: one write + one read of y forward, one
// write + two reads of dL/dy backward, two launches.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_maxpool_kernel(PoolArgs a, const T* __restrict__ x, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, T* __restrict__ y,
                                                               uint8_t* __restrict__ idx) {
  constexpr int V = DT<T>::VEC;
  const uint32_t CV = a.fcv.d;
  const uint32_t total = (uint32_t)a.N * a.To * a.Ho * a.Wo * CV;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t m, cvi, q, wo_, ho_, to_, n_;
    fd_divmod(i, a.fcv, m, cvi);
    const int c0 = (int)cvi * V;
    fd_divmod(m, a.fWo, q, wo_);
    fd_divmod(q, a.fHo, q, ho_);
    fd_divmod(q, a.fTo, n_, to_);
    const int wo = (int)wo_, ho = (int)ho_, to = (int)to_, n = (int)n_;
    float sc[V], sh[V];
    load_params<V>(scale, c0, sc);
    load_params<V>(shift, c0, sh);
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
          for (int e = 0; e < V; ++e) {
            const float yv = DT<T>::to_f(DT<T>::from_f(fmaxf(v[e] * sc[e] + sh[e], 0.f)));     // what bn_apply stores
            if (first || yv > best[e] || yv != yv) { best[e] = yv; bi[e] = tap; }
          }
          first = false;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) if (c0 + e >= a.C) best[e] = 0.f;
    Pack16<T>::store(y + (size_t)m * a.ldy + c0, best);
    uint8_t* ip = idx + (size_t)m * a.CP + c0;
#pragma unroll
    for (int e = 0; e < V; ++e) ip[e] = (uint8_t)bi[e];
  }
}

// g[e] += dL/dy of input element (row m_in, channels c0..c0+V): the pooled gradients of the windows that chose it.  Only
// the windows that contain the element are visited (o*s - p <= i < o*s - p + k per dimension: 3.4 of the 27 taps of a
// 3x3x3 / stride 2 pool on average), in ascending tap order, which fixes the order of the fp32 sums.
template <typename T>
__device__ __forceinline__ void pool_gather(const PoolArgs& a, const T* __restrict__ dyp, const uint8_t* __restrict__ idx,
                                            uint32_t m_in, int c0, float (&g)[DT<T>::VEC]) {
  constexpr int V = DT<T>::VEC;
  uint32_t q, wi_, hi_, ti_, n_;
  fd_divmod(m_in, a.fWi, q, wi_);
  fd_divmod(q, a.fHi, q, hi_);
  fd_divmod(q, a.fTi, n_, ti_);
  const int n = (int)n_;
  // strides are 1 or 2 (checked on the host): division by the stride is a shift, ceil(x / s) = (x + s - 1) >> (s - 1)
  const int st1 = a.st - 1, sh1 = a.sh - 1, sw1 = a.sw - 1;
  const int tn = (int)ti_ + a.pt, hn = (int)hi_ + a.ph, wn = (int)wi_ + a.pw;
  const int t_hi = min(tn >> st1, a.To - 1), t_lo = max((tn - a.kt + 1 + st1) >> st1, 0);
  const int h_hi = min(hn >> sh1, a.Ho - 1), h_lo = max((hn - a.kh + 1 + sh1) >> sh1, 0);
  const int w_hi = min(wn >> sw1, a.Wo - 1), w_lo = max((wn - a.kw + 1 + sw1) >> sw1, 0);
  for (int to = t_hi; to >= t_lo; --to) {
    const int dt = tn - to * a.st;
    for (int ho = h_hi; ho >= h_lo; --ho) {
      const int dh = hn - ho * a.sh;
      for (int wo = w_hi; wo >= w_lo; --wo) {
        const uint32_t tap = (uint32_t)((dt * a.kh + dh) * a.kw + wn - wo * a.sw);
        const int64_t mo = (int64_t)((n * a.To + to) * a.Ho + ho) * a.Wo + wo;
        float d[V];
        Pack16<T>::load(dyp + mo * a.ldy + c0, d);
        const uint8_t* ip = idx + mo * a.CP + c0;
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (ip[e] == tap) g[e] += d[e];
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_maxpool_kernel(PoolArgs a, const T* __restrict__ dyp, const uint8_t* __restrict__ idx,
                                             const T* __restrict__ x, const float* __restrict__ mean,
                                             const float* __restrict__ invstd, const float* __restrict__ scale,
                                             const float* __restrict__ shift, int64_t M, int C, int CP, int64_t rows_per_block,
                                             float* __restrict__ sums_all, int n_rep) {
  constexpr int V = DT<T>::VEC;
  float* sums = sums_all + (size_t)(blockIdx.x % n_rep) * 2 * CP;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  float mu[V], is[V], sc[V], sh[V];
  column_reduce<V, 2>(
      r0, r1, CP,
      [&](int64_t r, int c0, float(&acc)[2][V]) {
        float g[V], xx[V];
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = 0.f;
        pool_gather<T>(a, dyp, idx, (uint32_t)r, c0, g);
        Pack16<T>::load(x + r * a.ldx + c0, xx);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float act = xx[e] * sc[e] + sh[e];
          const float gg = !(act > 0.f) ? 0.f : g[e];
          acc[0][e] += gg;
          acc[1][e] += gg * (xx[e] - mu[e]) * is[e];
        }
      },
      [&](int c0, float(&acc)[2][V]) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (c0 + e < C) { atomicAdd(sums + c0 + e, acc[0][e]); atomicAdd(sums + CP + c0 + e, acc[1][e]); }
      },
      [&](int c0) {
        load_params<V>(mean, c0, mu);
        load_params<V>(invstd, c0, is);
        load_params<V>(scale, c0, sc);
        load_params<V>(shift, c0, sh);
      });
}

template <typename T>
__global__ void bn_bwd_apply_maxpool_kernel(PoolArgs a, const T* __restrict__ dyp, const uint8_t* __restrict__ idx,
                                            const T* __restrict__ x, const float* __restrict__ mean,
                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                            const float* __restrict__ sums_g, int rep_g, float inv_count, float dscale,
                                            float* dgamma, float* dbeta, T* __restrict__ dx, int lddx, uint32_t total, int C,
                                            int CP) {
  constexpr int V = DT<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) float coef[];      // [5][CP]: k1, k2, k3, scale, shift
  if (blockIdx.x == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float sb = 0.f, sg = 0.f;
      for (int r = 0; r < rep_g; ++r) { sb += sums_g[(size_t)r * 2 * CP + c]; sg += sums_g[(size_t)r * 2 * CP + CP + c]; }
      dbeta[c] += dscale * sb;
      dgamma[c] += dscale * sg;
    }
  }
  for (int c = threadIdx.x; c < CP; c += blockDim.x) {
    float k1 = 0.f, k2 = 0.f, k3 = 0.f;
    if (c < C) {
      float sg = 0.f, sgx = 0.f;
      for (int r = 0; r < rep_g; ++r) { sg += sums_g[(size_t)r * 2 * CP + c]; sgx += sums_g[(size_t)r * 2 * CP + CP + c]; }
      k1 = gamma[c] * invstd[c];
      k2 = -k1 * invstd[c] * sgx * inv_count;
      k3 = -k1 * sg * inv_count - k2 * mean[c];
    }
    coef[c] = k1; coef[CP + c] = k2; coef[2 * CP + c] = k3;
    coef[3 * CP + c] = c < C ? scale[c] : 0.f; coef[4 * CP + c] = c < C ? shift[c] : 0.f;
  }
  __syncthreads();
  const uint32_t CV = a.fcv.d;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    uint32_t row, cvi;
    fd_divmod(i, a.fcv, row, cvi);
    const int c0 = (int)cvi * V;
    float g[V], xx[V], o[V], k1[V], k2[V], k3[V], sc[V], sh[V];
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    pool_gather<T>(a, dyp, idx, row, c0, g);
    Pack16<T>::load(x + (size_t)row * a.ldx + c0, xx);
    load_params<V>(coef, c0, k1);
    load_params<V>(coef + CP, c0, k2);
    load_params<V>(coef + 2 * CP, c0, k3);
    load_params<V>(coef + 3 * CP, c0, sc);
    load_params<V>(coef + 4 * CP, c0, sh);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float act = xx[e] * sc[e] + sh[e];
      const float gg = !(act > 0.f) ? 0.f : g[e];
      o[e] = k1[e] * gg + k2[e] * xx[e] + k3[e];
    }
    Pack16<T>::store(dx + (size_t)row * lddx + c0, o);
  }
  (void)CV;
}

// ------------------------------------------------------------------ MaxPool3d 3x3x3 / stride 1 / padding 1, LDS-staged
// The gather kernels above read every window element through L1 / L2 (27 taps): with workgroups dealt round-robin to
// the 8 XCDs, neighbouring windows land in different L2s and each of them fetches the whole tensor (PMC: 7.2x the
// algorithmic bytes forward, 8.8x backward on the 14x14 fp32 pools).  Here a workgroup owns a TH x TW spatial tile of
// one clip and a chunk of CV 16-byte channel vectors, and walks the T planes once: plane t+1 is in flight to
// registers while plane t-1 is computed from a three-slot ring in LDS that holds planes t-2 .. t with a one-pixel
// halo, so every element leaves L2 once (1.29x with the halo rows) and the 27 taps are LDS reads.  Workgroup order
// is XCD-aware: consecutive tiles of a clip go to the same XCD, so halo rows and the other channel chunks of a
// 128-byte line are L2 hits.
//   forward : values are staged as order-preserving 32-bit keys (key_f32: a plain unsigned compare orders them, padding
//             is key 0 below every real value, a NaN with the sign bit clear wins as in PyTorch; -0.0 orders below
//             +0.0, a tie in PyTorch -- inputs are post-ReLU).  Taps are scanned in PyTorch's order with a strict
//             compare: first maximum wins, same values and uint8 tap index as maxpool_fwd_kernel.
//   backward: (dy, idx) staged with idx = 0xff outside the tensor; an input element adds dy of the windows that chose it
//             in the tap order of maxpool_bwd_kernel, so the fp32 sums are bit-identical to that kernel's.
struct PoolTileArgs {
  int N, T, H, W, C, CP, ldx, ldy;
  int nth, ntw, ncc, cpv;              // tiles along H and W, channel chunks, 16-byte vectors per pixel
  uint32_t groups, per_xcd;            // workgroups with work; ceil(groups / 8)
  FastDiv fcc, ftw, fth;
  int accumulate;
};

__device__ __forceinline__ uint32_t key_f32(uint32_t x) { return x ^ ((uint32_t)((int32_t)x >> 31) | 0x80000000u); }
__device__ __forceinline__ uint32_t unkey_f32(uint32_t k) { return (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k; }

// CV 16-byte global vectors per pixel and workgroup; Q = 4-channel quads per global vector (1 for fp32, 2 for bf16).  LDS
// holds one uint4 of 32-bit entries per quad (bf16 is widened while staging), and a work item is one quad of one pixel.
template <int TH, int TW, int CV, int Q>
struct PoolTile {
  static constexpr int HW = TW + 2, NPX = (TH + 2) * HW, NST = NPX * CV, SR = (NST + kThreads - 1) / kThreads;
  static constexpr int QV = CV * Q, NLE = NPX * QV, NIT = TH * TW * QV, R = (NIT + kThreads - 1) / kThreads;
  int n, h0, w0, cv0;
  __device__ __forceinline__ bool locate(const PoolTileArgs& a) {
    const uint32_t lb = (blockIdx.x & 7u) * a.per_xcd + (blockIdx.x >> 3);
    if (lb >= a.groups) return false;
    uint32_t q, cc, tw, th, nn;
    fd_divmod(lb, a.fcc, q, cc);
    fd_divmod(q, a.ftw, q, tw);
    fd_divmod(q, a.fth, nn, th);
    n = (int)nn; h0 = (int)th * TH; w0 = (int)tw * TW; cv0 = (int)cc * CV;
    return true;
  }
  // staged vector i of a plane -> halo pixel (h, w) and channel vector cv; false when nothing is to be fetched
  __device__ __forceinline__ bool stage_src(const PoolTileArgs& a, int i, int& h, int& w, int& cv) const {
    const int px = i / CV;
    cv = cv0 + (i - px * CV);
    const int hh = px / HW;
    h = h0 - 1 + hh; w = w0 - 1 + (px - hh * HW);
    return i < NST && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W && cv < a.cpv;
  }
  // item it of the tile -> pixel (ph, pw) inside the tile and quad ql of the chunk; false for lanes without an output
  __device__ __forceinline__ bool item(const PoolTileArgs& a, int it, int& ph, int& pw, int& ql) const {
    const int px = it / QV;
    ql = it - px * QV;
    ph = px / TW; pw = px - ph * TW;
    return it < NIT && h0 + ph < a.H && w0 + pw < a.W && cv0 * Q + ql < a.cpv * Q;
  }
};

// the two quads of a 16-byte vector of bf16, widened to fp32 bit patterns
__device__ __forceinline__ void widen_bf16x8(const uint4& v, uint4& q0, uint4& q1) {
  q0 = make_uint4(v.x << 16, v.x & 0xffff0000u, v.y << 16, v.y & 0xffff0000u);
  q1 = make_uint4(v.z << 16, v.z & 0xffff0000u, v.w << 16, v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 key_quad(const uint4& v) { return make_uint4(key_f32(v.x), key_f32(v.y), key_f32(v.z), key_f32(v.w)); }

template <typename T, int TH, int TW, int CV>
__global__ __launch_bounds__(kThreads) void pool333_fwd_tile_kernel(PoolTileArgs a, const T* __restrict__ x,
                                                                    T* __restrict__ y, uint8_t* __restrict__ idx) {
  constexpr int V = DT<T>::VEC, Q = V / 4;
  using PT = PoolTile<TH, TW, CV, Q>;
  __shared__ uint4 slots[2][PT::NLE];
  PT tl;
  if (!tl.locate(a)) return;
  const int tid = threadIdx.x;
  const int64_t plane = (int64_t)a.H * a.W * a.ldx;
  int64_t soff[PT::SR];
  bool sval[PT::SR];
#pragma unroll
  for (int s = 0; s < PT::SR; ++s) {
    int h, w, cv;
    sval[s] = tl.stage_src(a, s * kThreads + tid, h, w, cv);
    soff[s] = ((int64_t)(tl.n * a.T) * a.H * a.W + (int64_t)h * a.W + w) * a.ldx + cv * V;
  }
  uint4 pre[PT::SR];
  auto prefetch = [&](int t) {
#pragma unroll
    for (int s = 0; s < PT::SR; ++s)
      pre[s] = sval[s] ? *reinterpret_cast<const uint4*>(x + soff[s] + t * plane) : make_uint4(0u, 0u, 0u, 0u);
  };
  auto commit = [&](int slot) {
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int s = 0; s < PT::SR; ++s) {
      const int i = s * kThreads + tid;
      if (i >= PT::NST) break;
      if constexpr (Q == 1) {
        slots[slot][i] = sval[s] ? key_quad(pre[s]) : zero;
      } else {
        uint4 q0, q1;
        widen_bf16x8(pre[s], q0, q1);
        slots[slot][2 * i] = sval[s] ? key_quad(q0) : zero;
        slots[slot][2 * i + 1] = sval[s] ? key_quad(q1) : zero;
      }
    }
  };
  // The window maximum is taken plane by plane: the first maximum of the 3x3 spatial window of one plane (key and spatial
  // tap 0..8) is kept in registers for the two previous planes; an output is the first maximum of its three plane maxima in
  // plane order -- the element PyTorch's (dt, dh, dw) scan with a strict compare picks, with 9 + 3 compares, not 27.
  uint32_t k2[PT::R][4], c2[PT::R][4], k1[PT::R][4], c1[PT::R][4], k0[PT::R][4], c0[PT::R][4];
  bool live[PT::R];
  int base[PT::R];
#pragma unroll
  for (int r = 0; r < PT::R; ++r) {
    int ph, pw, ql;
    live[r] = tl.item(a, r * kThreads + tid, ph, pw, ql);
    base[r] = (ph * PT::HW + pw) * PT::QV + ql;
#pragma unroll
    for (int e = 0; e < 4; ++e) { k1[r][e] = 0u; c1[r][e] = 0u; k0[r][e] = 0u; c0[r][e] = 0u; }
  }
  prefetch(0);
  for (int t = 0; t <= a.T; ++t) {
    if (t < a.T) {
      commit(t & 1);
      if (t + 1 < a.T) prefetch(t + 1);
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < PT::R; ++r) {
      if (!live[r]) continue;
      const int it = r * kThreads + tid, px = it / PT::QV, ph = px / TW, pw = px - ph * TW;
      const int h = tl.h0 + ph, w = tl.w0 + pw;
      const uint32_t code0 = (h == 0 ? 3u : 0u) + (w == 0 ? 1u : 0u);
#pragma unroll
      for (int e = 0; e < 4; ++e) { k2[r][e] = k1[r][e]; c2[r][e] = c1[r][e]; k1[r][e] = k0[r][e]; c1[r][e] = c0[r][e]; }
      if (t < a.T) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { k0[r][e] = 0u; c0[r][e] = code0; }
        const uint4* pl = slots[t & 1] + base[r];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) {
            const uint4 k = pl[(dh * PT::HW + dw) * PT::QV];
            const uint32_t kk[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (kk[e] > k0[r][e]) { k0[r][e] = kk[e]; c0[r][e] = (uint32_t)(dh * 3 + dw); }
          }
      }
      if (t == 0) continue;
      const int to = t - 1;
      uint32_t best[4], bi[4];
      // planes to-1 (k2), to (k1), to+1 (k0); k2 / k0 are absent at the ends of the clip
      if (to >= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { best[e] = k2[r][e]; bi[e] = c2[r][e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k1[r][e] > best[e]) { best[e] = k1[r][e]; bi[e] = 9u + c1[r][e]; }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { best[e] = k1[r][e]; bi[e] = 9u + c1[r][e]; }
      }
      if (t < a.T) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k0[r][e] > best[e]) { best[e] = k0[r][e]; bi[e] = 18u + c0[r][e]; }
      }
      const int cc0 = (tl.cv0 * Q + (it - px * PT::QV)) * 4;
      uint32_t ov[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = (cc0 + e < a.C) ? unkey_f32(best[e]) : 0u;
      const int64_t m = ((int64_t)(tl.n * a.T + to) * a.H + h) * a.W + w;
      if constexpr (Q == 1)
        *reinterpret_cast<uint4*>(y + m * a.ldy + cc0) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
      else
        *reinterpret_cast<uint2*>(y + m * a.ldy + cc0) = make_uint2((ov[0] >> 16) | (ov[1] & 0xffff0000u), (ov[2] >> 16) | (ov[3] & 0xffff0000u));
      *reinterpret_cast<uint32_t*>(idx + m * a.CP + cc0) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
  }
}

// acc[e] += g[e] for the elements whose tap byte in iw equals tap.  v_cmpx leaves the compare in EXEC, the add runs on the
// lanes it kept and EXEC comes back from a scalar copy: two vector instructions per element where the compiler's
// and / compare / add / select sequence takes four.
__device__ __forceinline__ void add_if_tap4(float (&acc)[4], const uint4& g, uint32_t iw, uint32_t tap) {
  uint64_t sv;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "v_cmpx_eq_u32_sdwa vcc, %[iw], %[tap] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
      "v_add_f32_e32 %[a0], %[a0], %[g0]\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_eq_u32_sdwa vcc, %[iw], %[tap] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
      "v_add_f32_e32 %[a1], %[a1], %[g1]\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_eq_u32_sdwa vcc, %[iw], %[tap] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
      "v_add_f32_e32 %[a2], %[a2], %[g2]\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      "v_cmpx_eq_u32_sdwa vcc, %[iw], %[tap] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
      "v_add_f32_e32 %[a3], %[a3], %[g3]\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [sv] "=&s"(sv)
      : [iw] "v"(iw), [tap] "s"(tap), [g0] "v"(g.x), [g1] "v"(g.y), [g2] "v"(g.z), [g3] "v"(g.w)
      : "vcc");
}

template <typename T, int TH, int TW, int CV>
__global__ __launch_bounds__(kThreads) void pool333_bwd_tile_kernel(PoolTileArgs a, const T* __restrict__ dy,
                                                                    const uint8_t* __restrict__ idx, T* __restrict__ dx) {
  constexpr int V = DT<T>::VEC, Q = V / 4;
  using PT = PoolTile<TH, TW, CV, Q>;
  __shared__ uint4 gring[3][PT::NLE];
  __shared__ uint32_t iring[3][PT::NLE];
  PT tl;
  if (!tl.locate(a)) return;
  const int tid = threadIdx.x;
  const int64_t plane = (int64_t)a.H * a.W;
  int64_t srow[PT::SR];
  int scv[PT::SR];
  bool sval[PT::SR];
#pragma unroll
  for (int s = 0; s < PT::SR; ++s) {
    int h, w;
    sval[s] = tl.stage_src(a, s * kThreads + tid, h, w, scv[s]);
    srow[s] = (int64_t)(tl.n * a.T) * a.H * a.W + (int64_t)h * a.W + w;
  }
  uint4 pre[PT::SR];
  uint32_t prei[PT::SR][Q];
  auto prefetch = [&](int t) {
#pragma unroll
    for (int s = 0; s < PT::SR; ++s) {
      const int64_t m = srow[s] + t * plane;
      pre[s] = sval[s] ? *reinterpret_cast<const uint4*>(dy + m * a.ldy + scv[s] * V) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int j = 0; j < Q; ++j)
        prei[s][j] = sval[s] ? reinterpret_cast<const uint32_t*>(idx + m * a.CP + scv[s] * V)[j] : 0xffffffffu;
    }
  };
  auto commit = [&](int slot) {
#pragma unroll
    for (int s = 0; s < PT::SR; ++s) {
      const int i = s * kThreads + tid;
      if (i >= PT::NST) break;
      if constexpr (Q == 1) {
        gring[slot][i] = pre[s];
        iring[slot][i] = prei[s][0];
      } else {
        uint4 q0, q1;
        widen_bf16x8(pre[s], q0, q1);
        gring[slot][2 * i] = q0; gring[slot][2 * i + 1] = q1;
        iring[slot][2 * i] = prei[s][0]; iring[slot][2 * i + 1] = prei[s][1];
      }
    }
  };
  auto compute = [&](int ti) {
#pragma unroll
    for (int r = 0; r < PT::R; ++r) {
      int ph, pw, ql;
      if (!tl.item(a, r * kThreads + tid, ph, pw, ql)) continue;
      const int c0 = (tl.cv0 * Q + ql) * 4;
      const int64_t m = ((int64_t)(tl.n * a.T + ti) * a.H + tl.h0 + ph) * a.W + tl.w0 + pw;
      T* dst = dx + m * a.ldx + c0;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.accumulate) {
        if constexpr (Q == 1) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(dst);
          acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w;
        } else {
          const uint2 v = *reinterpret_cast<const uint2*>(dst);
          acc[0] = __uint_as_float(v.x << 16); acc[1] = __uint_as_float(v.x & 0xffff0000u);
          acc[2] = __uint_as_float(v.y << 16); acc[3] = __uint_as_float(v.y & 0xffff0000u);
        }
      }
      // window (dh, dw) of this input pixel is the output at tile offset (ph + 1 - dh, pw + 1 - dw), i.e. halo pixel
      // (ph + 2 - dh, pw + 2 - dw): offsets from the tile-corner base stay non-negative (immediate LDS offsets)
      const int base = (ph * PT::HW + pw) * PT::QV + ql;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const int to = ti + 1 - dt;
        if (to < 0 || to >= a.T) continue;
        const uint4* gs = gring[to % 3] + base;
        const uint32_t* is = iring[to % 3] + base;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) {
            const int o = ((2 - dh) * PT::HW + (2 - dw)) * PT::QV;
            add_if_tap4(acc, gs[o], is[o], (uint32_t)(dt * 9 + dh * 3 + dw));
          }
      }
      if constexpr (Q == 1) {
        f32x4 v = {acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 v = {(bf16_t)acc[0], (bf16_t)acc[1], (bf16_t)acc[2], (bf16_t)acc[3]};
        *reinterpret_cast<bf16x4*>(dst) = v;
      }
    }
  };
  prefetch(0);
  for (int t = 0; t <= a.T; ++t) {
    if (t < a.T) commit(t % 3);
    if (t + 1 < a.T) prefetch(t + 1);
    __syncthreads();
    if (t >= 1) compute(t - 1);
    __syncthreads();
  }
}

static bool pool333_shape(const dv_pool_desc* d) {
  return d->kt == 3 && d->kh == 3 && d->kw == 3 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 1 && d->ph == 1 &&
         d->pw == 1;
}
// tile geometry for a staged launch; false when the plane is too small to be worth a tile (the gather kernels run then)
static bool pool_tile_args(const PoolArgs& p, int V, int TH, int TW, int CV, int accumulate, PoolTileArgs& a) {
  if (p.Hi * p.Wi < 25) return false;
  a.N = p.N; a.T = p.Ti; a.H = p.Hi; a.W = p.Wi; a.C = p.C; a.CP = p.CP; a.ldx = p.ldx; a.ldy = p.ldy;
  a.nth = (p.Hi + TH - 1) / TH; a.ntw = (p.Wi + TW - 1) / TW; a.cpv = p.CP / V; a.ncc = (a.cpv + CV - 1) / CV;
  const int64_t groups = (int64_t)p.N * a.nth * a.ntw * a.ncc;
  if (groups >= (1ll << 28)) return false;
  a.groups = (uint32_t)groups; a.per_xcd = (uint32_t)((groups + 7) / 8);
  a.fcc = make_fastdiv((uint32_t)a.ncc); a.ftw = make_fastdiv((uint32_t)a.ntw); a.fth = make_fastdiv((uint32_t)a.nth);
  a.accumulate = accumulate;
  return true;
}
static int pool_tile_cv(int dflt) { return dflt; }
// Tile width / channel vectors per workgroup.  Measured on the 14x14 and 7x7 pools of S3D-G (tools/pool_sweep.sh): the
// backward is occupancy-bound (three-plane ring of dy + idx), so it takes the 7-wide tile and 64 bytes of dy per pixel; the
// forward (two single-plane slots) is flat between the shapes.
static int pool_tile_tw(int W, bool narrow) { return (W <= 7 || narrow) ? 7 : 14; }
static bool pool_tile_off() { return false; }

// ------------------------------------------------------------------ spatial mean / gating
// One workgroup per (sample, slice of the S positions): with one workgroup per sample (N = 128) half the CUs had no
// work and the others one resident workgroup each (~1.5 TB/s).  Slices are added with fp32 atomics into the zeroed
// output (the host wrappers memset it on the same stream).
// Per-sample column sums over the S positions.  Large levels are split over CHANNEL chunks (grid.y, `ccv` 16-byte vectors =
// >= 256 bytes of a row each), not over positions: every output element has one owner, so there are no float atomics and no
// memset in front -- the results do not depend on the order workgroups finish in.
template <typename T>
__global__ void spatial_mean_kernel(const T* __restrict__ x, int ldx, int S, int C, int CP, int ccv, float* __restrict__ out) {
  constexpr int V = DT<T>::VEC;
  const int n = blockIdx.x;
  const float inv = 1.f / (float)S;
  const int cb = (int)blockIdx.y * ccv * V, cpl = min(ccv * V, CP - cb);
  if (cpl <= 0) return;
  column_reduce<V, 1>(
      (int64_t)n * S, (int64_t)(n + 1) * S, cpl,
      [&](int64_t r, int c0, float(&acc)[1][V]) {
        float v[V];
        Pack16<T>::load(x + r * ldx + cb + c0, v);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[0][e] += v[e];
      },
      [&](int c0, float(&acc)[1][V]) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (cb + c0 + e < C) out[(size_t)n * C + cb + c0 + e] = acc[0][e] * inv;
      });
}

template <typename T>
__global__ void gate_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                       const float* __restrict__ g, int S, int C, int CP, int ccv, float* __restrict__ dpre,
                                       int x_is_output) {
  constexpr int V = DT<T>::VEC;
  const int n = blockIdx.x;
  const int cb = (int)blockIdx.y * ccv * V, cpl = min(ccv * V, CP - cb);
  if (cpl <= 0) return;
  column_reduce<V, 1>(
      (int64_t)n * S, (int64_t)(n + 1) * S, cpl,
      [&](int64_t r, int c0, float(&acc)[1][V]) {
        float a[V], b[V];
        Pack16<T>::load(dy + r * lddy + cb + c0, a);
        Pack16<T>::load(x + r * ldx + cb + c0, b);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[0][e] += a[e] * b[e];
      },
      [&](int c0, float(&acc)[1][V]) {
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (cb + c0 + e < C) {
            const float gg = g[(size_t)n * C + cb + c0 + e];
            dpre[(size_t)n * C + cb + c0 + e] = x_is_output ? acc[0][e] * (1.f - gg) : acc[0][e] * gg * (1.f - gg);
          }
      });
}

// MODE 0: y = x*g[n][c]          (gate_scale)
// MODE 1: dx (+)= dy*g + dmean/S  (gate_bwd_apply)
// MODE 2: dx (+)= dout[n][c]/S    (spatial_mean_bwd)
template <typename T, int MODE>
__global__ void rowscale_kernel(const T* __restrict__ a, int lda, const float* __restrict__ g,
                                const float* __restrict__ dm, uint32_t total, FastDiv fcv, FastDiv fS, int C,
                                T* __restrict__ o, int ldo, int accumulate) {
  constexpr int V = DT<T>::VEC;
  const uint32_t CV = fcv.d;
  const float invS = 1.f / (float)fS.d;
  const bool vec = (C % V) == 0 && (MODE == 2 || ((uintptr_t)g & 15) == 0) && (MODE == 0 || ((uintptr_t)dm & 15) == 0);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t row = fd_div(i, fcv);
    const int c0 = (int)(i - row * CV) * V;
    const int n = (int)fd_div(row, fS);
    float v[V], old[V];
    if (MODE != 2) Pack16<T>::load(a + (size_t)row * lda + c0, v);
    if (accumulate) Pack16<T>::load(o + (size_t)row * ldo + c0, old);
    float gv[V], dv[V];
    if (vec) {                                     // C % V == 0: the [N][C] fp32 rows are read as float4s
      if (MODE != 2) load_params<V>(g + (size_t)n * C, c0, gv);
      if (MODE != 0) load_params<V>(dm + (size_t)n * C, c0, dv);
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const bool in = c0 + e < C;
        gv[e] = (MODE != 2 && in) ? g[(size_t)n * C + c0 + e] : 0.f;
        dv[e] = (MODE != 0 && in) ? dm[(size_t)n * C + c0 + e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float r;
      if (MODE == 0) r = v[e] * gv[e];
      else if (MODE == 1) r = v[e] * gv[e] + dv[e] * invS;
      else r = dv[e] * invS;
      if (!vec && c0 + e >= C) r = 0.f;
      v[e] = accumulate ? old[e] + r : r;
    }
    Pack16<T>::store(o + (size_t)row * ldo + c0, v);
  }
}

// ------------------------------------------------------------------ small fp32 helpers
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
                                       (T*)y, N, C, T_, H, W, sxn, ldy, mean3, istd3, perm, n_seg, 0, H, W));
  return dv_launch_status();
}

extern "C" int dv_ingest_ncdhw_pad(int32_t dtype, const float* x, void* y, int32_t N, int32_t C, int32_t T_, int32_t H,
                                   int32_t W, int64_t sxn, int32_t ldy, const float* mean3, const float* istd3,
                                   const int32_t* perm, int32_t n_seg, int32_t pad, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || C > 4 || ldy < 4 || ldy % 4 || pad < 0) return DV_EINVAL;
  if (perm && (n_seg <= 0 || T_ % n_seg)) return DV_EINVAL;
  if ((mean3 == nullptr) != (istd3 == nullptr)) return DV_EINVAL;
  const int64_t total = (int64_t)N * T_ * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL((ingest_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, ST(stream), x,
                                       (T*)y, N, C, T_, H, W, sxn, ldy, mean3, istd3, perm, n_seg, pad, H + 2 * pad,
                                       W + 2 * pad));
  return dv_launch_status();
}

__global__ void fill_cols_kernel(float* __restrict__ p, int64_t rows, int pitch, int col0, int ncols, float v) {
  const int64_t total = rows * ncols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ncols;
    p[r * pitch + col0 + (int)(i - r * ncols)] = v;
  }
}

extern "C" int dv_fill_cols_f32(float* p, int64_t rows, int32_t pitch, int32_t col0, int32_t ncols, float value, void* stream) {
  if (!p || rows <= 0 || ncols <= 0 || col0 < 0 || col0 + ncols > pitch) return DV_EINVAL;
  hipLaunchKernelGGL(fill_cols_kernel, dim3(grid_for(rows * ncols)), dim3(kThreads), 0, ST(stream), p, rows, pitch, col0, ncols,
                     value);
  return dv_launch_status();
}

// eval-mode BatchNorm (running statistics): scale = gamma * rsqrt(running_var + eps), shift = beta - running_mean * scale;
// arrays are written up to round_up(C, 8) (pad lanes zero) for the 16-byte loads of dv_bn_apply
__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C, int CP,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= CP) return;
  float sc = 0.f, sh = 0.f;
  if (c < C) {
    sc = gamma[c] * rsqrtf(rv[c] + eps);
    sh = beta[c] - rm[c] * sc;
  }
  scale[c] = sc;
  shift[c] = sh;
}

extern "C" int dv_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float eps, int32_t C, float* scale, float* shift, void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0) return DV_EINVAL;
  const int CP = cp8(C);
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((CP + 127) / 128), dim3(128), 0, ST(stream), gamma, beta, running_mean,
                     running_var, eps, C, CP, scale, shift);
  return dv_launch_status();
}

// BatchNorm partials of a small fp32 [M][C] matrix (BatchNorm1d of the classifier head, M = batch): one thread per
// channel, ONE tile: partials [2][C][1] = (sum, M2 about the mean), the format dv_bn_stats_finalize consumes
__global__ void bn_rows_partials_kernel(const float* __restrict__ x, int ldx, int M, int C, float* __restrict__ part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < M; ++r) s += x[(size_t)r * ldx + c];
  const float mu = s / (float)M;
  float q = 0.f;
  for (int r = 0; r < M; ++r) { const float d = x[(size_t)r * ldx + c] - mu; q += d * d; }
  part[c] = s;
  part[C + c] = q;
}

extern "C" int dv_bn_rows_partials_f32(const float* x, int32_t ldx, int32_t M, int32_t C, float* partials, void* stream) {
  if (!x || !partials || M <= 0 || C <= 0 || ldx < C) return DV_EINVAL;
  hipLaunchKernelGGL(bn_rows_partials_kernel, dim3((C + 127) / 128), dim3(128), 0, ST(stream), x, ldx, M, C, partials);
  return dv_launch_status();
}

__global__ void addcmul_kernel(float* __restrict__ y, const float* __restrict__ a, const float* __restrict__ b, float alpha, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += alpha * a[i] * (b ? b[i] : 1.f);
}

extern "C" int dv_addcmul_f32(float* y, const float* a, const float* b, float alpha, int32_t n, void* stream) {
  if (!y || !a || n <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(addcmul_kernel, dim3((n + 255) / 256), dim3(256), 0, ST(stream), y, a, b, alpha, n);
  return dv_launch_status();
}

extern "C" int dv_bn_reduce_stats(const float* partials, int32_t n_tiles, int32_t tile_rows, int32_t pitch, int64_t M,
                                  int32_t C, float* local_stats, void* stream) {
  if (!partials || !local_stats || n_tiles <= 0 || C <= 0 || M <= 0 || pitch < C) return DV_EINVAL;
  BnFinalize f = {};
  hipLaunchKernelGGL(bn_reduce_stats_kernel, dim3(C), dim3(n_tiles >= 2048 ? 1024 : kThreads), 0, ST(stream), partials, n_tiles,
                     tile_rows, pitch, M, C, local_stats, 0, f);
  return dv_launch_status();
}

extern "C" int dv_bn_stats_finalize(const float* partials, int32_t n_tiles, int32_t tile_rows, int32_t pitch, int64_t M,
                                    int32_t C, float* local_stats, const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                    float* shift, void* stream) {
  if (!partials || !local_stats || n_tiles <= 0 || C <= 0 || M <= 0 || pitch < C) return DV_EINVAL;
  if (!gamma || !beta || !mean || !invstd || !scale || !shift) return DV_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return DV_EINVAL;
  BnFinalize f = {gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift};
  hipLaunchKernelGGL(bn_reduce_stats_kernel, dim3(C), dim3(n_tiles >= 2048 ? 1024 : kThreads), 0, ST(stream), partials, n_tiles,
                     tile_rows, pitch, M, C, local_stats, 1, f);
  return dv_launch_status();
}

extern "C" int dv_bn_stats_multi(const dv_bn_item* items, int32_t n, int32_t finalize, int32_t total_blocks, void* stream) {
  if (!items || n <= 0 || total_blocks <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(bn_stats_multi_kernel, dim3(total_blocks), dim3(kThreads), 0, ST(stream), items, n, finalize);
  return dv_launch_status();
}
extern "C" int dv_bn_finalize_multi(const dv_bn_item* items, int32_t n, int32_t total_blocks, const float* local_base,
                                    const float* gathered, int32_t R, int32_t stride, void* stream) {
  if (!items || n <= 0 || total_blocks <= 0 || !local_base || !gathered || R <= 0 || stride <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(bn_finalize_multi_kernel, dim3(total_blocks), dim3(128), 0, ST(stream), items, n, local_base, gathered, R,
                     stride);
  return dv_launch_status();
}
extern "C" int dv_bn_apply_multi(int32_t dtype, const dv_bn_item* items, int32_t n, int32_t total_blocks, void* stream) {
  if (!items || n <= 0 || total_blocks <= 0) return DV_EINVAL;
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_multi_kernel<T>), dim3(total_blocks), dim3(kThreads), 0, ST(stream), items, n));
  return dv_launch_status();
}
extern "C" int dv_bn_bwd_reduce_multi(int32_t dtype, const dv_bn_item* items, int32_t n, int32_t total_blocks, void* stream) {
  if (!items || n <= 0 || total_blocks <= 0) return DV_EINVAL;
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_reduce_multi_kernel<T>), dim3(total_blocks), dim3(kThreads), 0, ST(stream), items, n));
  return dv_launch_status();
}
extern "C" int dv_bn_bwd_apply_multi(int32_t dtype, const dv_bn_item* items, int32_t n, int32_t total_blocks, int32_t max_c,
                                     void* stream) {
  if (!items || n <= 0 || total_blocks <= 0 || max_c <= 0) return DV_EINVAL;
  const int CP = cp8(max_c);
  if (5 * CP * 4 > 60 * 1024) return DV_EUNSUPPORTED;
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_multi_kernel<T>), dim3(total_blocks), dim3(kThreads), 5 * CP * sizeof(float),
                                       ST(stream), items, n));
  return dv_launch_status();
}

extern "C" int dv_bn_finalize(const float* stats, int32_t R, int32_t stride, int32_t C, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                              float* scale, float* shift, void* stream) {
  if (!stats || R <= 0 || C <= 0 || stride < 2 * C + 1 || !gamma || !beta || !mean || !invstd || !scale || !shift) return DV_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return DV_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, ST(stream), stats, R, stride, C, gamma, beta, eps,
                     momentum, running_mean, running_var, mean, invstd, scale, shift);
  return dv_launch_status();
}

extern "C" int dv_bn_apply(int32_t dtype, const void* x, int32_t ldx, const float* scale, const float* shift,
                           const void* residual, int32_t ldr, void* y, int32_t ldy, int64_t M, int32_t C, int32_t flags,
                           void* stream) {
  const int CP = cp8(C);
  if (!x || !y || !scale || !shift || M <= 0 || C <= 0 || ldx < CP || ldy < CP || (residual && ldr < CP)) return DV_EINVAL;
  if (!aligned16(x) || !aligned16(y) || (residual && !aligned16(residual))) return DV_EALIGN;
  if (!aligned16(scale) || !aligned16(shift)) return DV_EALIGN;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (ldx % V || ldy % V || (residual && ldr % V)) return DV_EALIGN;
    const int64_t total = M * (CP / V);
    if (total >= (1ll << 31)) return DV_EINVAL;
    hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, ST(stream), (const T*)x,
                       ldx, scale, shift, (const T*)residual, ldr, (T*)y, ldy, (uint32_t)total, C,
                       make_fastdiv((uint32_t)(CP / V)), flags);
  });
  return dv_launch_status();
}

extern "C" int dv_bn_bwd_blocks(int64_t M, int32_t C) {
  (void)C;
  // Every workgroup ends with an LDS fold and 2*C atomics, so fewer, longer workgroups win as soon as there are enough of
  // them to cover the chip (measured in the S3D-G step: 100k-row layers 83 -> 43 us going from 1024 to 320 workgroups,
  // 12.5k-row layers 25 -> 18 us with 64 instead of 32 rows each); the >= 300k-row layers keep 1024.
  const int rpb = M <= 2048 ? 32 : 64;
  const int cap = M >= 300000 ? 1024 : 320;       // (2048 on the >= 1M-row layers: 105 -> 127 us, 208 -> 216 us)
  int64_t b = (M + rpb - 1) / rpb;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int dv_bn_bwd_reduce(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                                int32_t ldx, const float* mean, const float* invstd, int64_t M, int32_t C, int32_t flags,
                                float* sums, int32_t n_rep, float* ws, void* stream) {
  const int CP = cp8(C);
  const bool mask = !(flags & DV_NO_RELU_MASK);
  if (!dy || !x || (mask && !y) || !mean || !invstd || !sums || M <= 0 || C <= 0 || n_rep <= 0) return DV_EINVAL;
  if (ws && (reinterpret_cast<uintptr_t>(ws) & 3)) return DV_EALIGN;
  if (lddy < CP || ldx < CP || (mask && ldy < CP)) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(x) || (mask && !aligned16(y)) || !aligned16(mean) || !aligned16(invstd)) return DV_EALIGN;
  const int blocks = dv_bn_bwd_blocks(M, C);
  const int64_t rpb = (M + blocks - 1) / blocks;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (lddy % V || ldx % V || (mask && ldy % V)) return DV_EALIGN;
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(blocks), dim3(kThreads), 0, ST(stream), (const T*)dy, lddy,
                       (const T*)y, ldy, (const T*)x, ldx, mean, invstd, M, C, CP, flags, rpb, sums, n_rep, ws);
  });
  return dv_launch_status();
}

extern "C" int64_t dv_bn_bwd_reduce_workspace(int64_t M, int32_t C) {
  if (M <= 0 || C <= 0) return 0;
  return ordered_fold_floats(dv_bn_bwd_blocks(M, C), cp8(C)) * (int64_t)sizeof(float);
}

extern "C" int dv_bn_bwd_apply(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                               int32_t ldx, const float* mean, const float* invstd, const float* gamma,
                               const float* sums_global, int32_t rep_global, float inv_count, float dparam_scale,
                               float* dgamma, float* dbeta, void* dx, int32_t lddx, void* dres,
                               int32_t lddres, int64_t M, int32_t C, int32_t flags, void* stream) {
  const int CP = cp8(C);
  const bool mask = !(flags & DV_NO_RELU_MASK);
  if (!dy || !x || (mask && !y) || !mean || !invstd || !gamma || !sums_global || !dx || M <= 0 || C <= 0) return DV_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr) || rep_global <= 0) return DV_EINVAL;
  if (lddy < CP || ldx < CP || lddx < CP || (mask && ldy < CP) || (dres && lddres < CP)) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(x) || !aligned16(dx) || (mask && !aligned16(y)) || (dres && !aligned16(dres))) return DV_EALIGN;
  if (3 * CP * 4 > 60 * 1024) return DV_EUNSUPPORTED;
  DISPATCH_T(dtype, {
    constexpr int V = DT<T>::VEC;
    if (lddy % V || ldx % V || lddx % V || (mask && ldy % V) || (dres && lddres % V)) return DV_EALIGN;
    const int64_t total = M * (CP / V);
    if (total >= (1ll << 31)) return DV_EINVAL;
    int grid = grid_for(total, 2048);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(grid), dim3(kThreads), 3 * CP * sizeof(float), ST(stream),
                       (const T*)dy, lddy, (const T*)y, ldy, (const T*)x, ldx, mean, invstd, gamma, sums_global,
                       rep_global, inv_count, dparam_scale, dgamma, dbeta, (T*)dx, lddx, (T*)dres, lddres, (uint32_t)total, C, CP,
                       make_fastdiv((uint32_t)(CP / V)), flags);
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
  if (d->st > 2 || d->sh > 2 || d->sw > 2) return DV_EUNSUPPORTED;
  if ((int64_t)a.N * a.Ti * a.Hi * a.Wi * (a.CP / V) >= (1ll << 31)) return DV_EINVAL;
  a.fcv = make_fastdiv((uint32_t)(a.CP / V));
  a.fWo = make_fastdiv((uint32_t)a.Wo); a.fHo = make_fastdiv((uint32_t)a.Ho); a.fTo = make_fastdiv((uint32_t)a.To);
  a.fWi = make_fastdiv((uint32_t)a.Wi); a.fHi = make_fastdiv((uint32_t)a.Hi); a.fTi = make_fastdiv((uint32_t)a.Ti);
  return DV_OK;
}

extern "C" int dv_maxpool3d_fwd(const dv_pool_desc* d, const void* x, void* y, uint8_t* idx, void* stream) {
  PoolArgs a;
  int rc = pool_args(d, a);
  if (rc) return rc;
  if (!x || !y || !idx) return DV_EINVAL;
  if (!aligned16(x) || !aligned16(y) || (reinterpret_cast<uintptr_t>(idx) & 7)) return DV_EALIGN;
  PoolTileArgs ta;
  if (pool333_shape(d) && !pool_tile_off()) {
    const int cv = pool_tile_cv(4), tw = pool_tile_tw(a.Wi, d->dtype == DV_BF16);
#define POOL_FWD_TILE(TW_, CV_)                                                                                             \
  DISPATCH_T(d->dtype, {                                                                                                     \
    if (cv == CV_ && tw == TW_ && pool_tile_args(a, DT<T>::VEC, 7, TW_, CV_, 0, ta)) {                                       \
      hipLaunchKernelGGL((pool333_fwd_tile_kernel<T, 7, TW_, CV_>), dim3(8 * ta.per_xcd), dim3(kThreads), 0, ST(stream), ta, \
                         (const T*)x, (T*)y, idx);                                                                           \
      return dv_launch_status();                                                                                             \
    }                                                                                                                        \
  })
    POOL_FWD_TILE(7, 2); POOL_FWD_TILE(7, 4); POOL_FWD_TILE(7, 8);
    POOL_FWD_TILE(14, 2); POOL_FWD_TILE(14, 4); POOL_FWD_TILE(14, 8);
#undef POOL_FWD_TILE
  }
  DISPATCH_T(d->dtype, {
    const int64_t total = (int64_t)a.N * a.To * a.Ho * a.Wo * (a.CP / DT<T>::VEC);
    hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid8_for(total, 16384)), dim3(kThreads), 0, ST(stream), a, (const T*)x,
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
  PoolTileArgs ta;
  if (pool333_shape(d) && !pool_tile_off() && (reinterpret_cast<uintptr_t>(idx) & 7) == 0) {
    const int cv = pool_tile_cv(d->dtype == DV_BF16 ? 2 : 4), tw = pool_tile_tw(a.Wi, true);
    const int acc = (flags & DV_ACCUM) ? 1 : 0;
#define POOL_BWD_TILE(TW_, CV_)                                                                                             \
  DISPATCH_T(d->dtype, {                                                                                                     \
    if (cv == CV_ && tw == TW_ && pool_tile_args(a, DT<T>::VEC, 7, TW_, CV_, acc, ta)) {                                     \
      hipLaunchKernelGGL((pool333_bwd_tile_kernel<T, 7, TW_, CV_>), dim3(8 * ta.per_xcd), dim3(kThreads), 0, ST(stream), ta, \
                         (const T*)dy, idx, (T*)dx);                                                                         \
      return dv_launch_status();                                                                                             \
    }                                                                                                                        \
  })
    POOL_BWD_TILE(7, 2); POOL_BWD_TILE(7, 4); POOL_BWD_TILE(7, 8);
    POOL_BWD_TILE(14, 2); POOL_BWD_TILE(14, 4); POOL_BWD_TILE(14, 8);
#undef POOL_BWD_TILE
  }
  if (a.kh == 3 && a.kw == 3 && a.sh == 2 && a.sw == 2 && a.ph == 1 && a.pw == 1 && (reinterpret_cast<uintptr_t>(idx) & 7) == 0) {
    const int Hq = (a.Hi + 1) / 2, Wq = (a.Wi + 1) / 2;
    DISPATCH_T(d->dtype, {
      const int64_t total = (int64_t)a.N * a.Ti * Hq * Wq * (a.CP / DT<T>::VEC);
      hipLaunchKernelGGL((maxpool_bwd_quad_kernel<T>), dim3(grid8_for(total, 16384)), dim3(kThreads), 0, ST(stream), a,
                         make_fastdiv((uint32_t)Wq), make_fastdiv((uint32_t)Hq), (const T*)dy, idx, (T*)dx, (flags & DV_ACCUM) ? 1 : 0);
    });
    return dv_launch_status();
  }
  DISPATCH_T(d->dtype, {
    const int64_t total = (int64_t)a.N * a.Ti * a.Hi * a.Wi * (a.CP / DT<T>::VEC);
    hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid8_for(total, 16384)), dim3(kThreads), 0, ST(stream), a, (const T*)dy,
                       idx, (T*)dx, (flags & DV_ACCUM) ? 1 : 0);
  });
  return dv_launch_status();
}

extern "C" int dv_bn_apply_maxpool(const dv_pool_desc* d, const void* x, const float* scale, const float* shift, void* y,
                                   uint8_t* idx, void* stream) {
  PoolArgs a;
  int rc = pool_args(d, a);
  if (rc) return rc;
  if (!x || !scale || !shift || !y || !idx) return DV_EINVAL;
  if (!aligned16(x) || !aligned16(y) || !aligned16(scale) || !aligned16(shift) || (reinterpret_cast<uintptr_t>(idx) & 7)) return DV_EALIGN;
  DISPATCH_T(d->dtype, {
    const int64_t total = (int64_t)a.N * a.To * a.Ho * a.Wo * (a.CP / DT<T>::VEC);
    hipLaunchKernelGGL((bn_apply_maxpool_kernel<T>), dim3(grid_for(total, 16384)), dim3(kThreads), 0, ST(stream), a, (const T*)x,
                       scale, shift, (T*)y, idx);
  });
  return dv_launch_status();
}

extern "C" int dv_bn_bwd_reduce_maxpool(const dv_pool_desc* d, const void* dy_pool, const uint8_t* idx, const void* x,
                                        const float* mean, const float* invstd, const float* scale, const float* shift,
                                        float* sums, int32_t n_rep, void* stream) {
  PoolArgs a;
  int rc = pool_args(d, a);
  if (rc) return rc;
  if (!dy_pool || !idx || !x || !mean || !invstd || !scale || !shift || !sums || n_rep <= 0) return DV_EINVAL;
  if (!aligned16(dy_pool) || !aligned16(x) || !aligned16(mean) || !aligned16(invstd) || !aligned16(scale) || !aligned16(shift)) return DV_EALIGN;
  const int64_t M = (int64_t)a.N * a.Ti * a.Hi * a.Wi;
  const int blocks = dv_bn_bwd_blocks(M, a.C);
  const int64_t rpb = (M + blocks - 1) / blocks;
  DISPATCH_T(d->dtype, hipLaunchKernelGGL((bn_bwd_reduce_maxpool_kernel<T>), dim3(blocks), dim3(kThreads), 0, ST(stream), a,
                                          (const T*)dy_pool, idx, (const T*)x, mean, invstd, scale, shift, M, a.C, a.CP, rpb,
                                          sums, n_rep));
  return dv_launch_status();
}

extern "C" int dv_bn_bwd_apply_maxpool(const dv_pool_desc* d, const void* dy_pool, const uint8_t* idx, const void* x,
                                       const float* mean, const float* invstd, const float* gamma, const float* scale,
                                       const float* shift, const float* sums_global, int32_t rep_global, float inv_count,
                                       float dparam_scale, float* dgamma, float* dbeta, void* dx, int32_t lddx, void* stream) {
  PoolArgs a;
  int rc = pool_args(d, a);
  if (rc) return rc;
  if (!dy_pool || !idx || !x || !mean || !invstd || !gamma || !scale || !shift || !sums_global || !dx || rep_global <= 0) return DV_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr) || lddx < a.CP) return DV_EINVAL;
  if (!aligned16(dy_pool) || !aligned16(x) || !aligned16(dx)) return DV_EALIGN;
  if (5 * a.CP * 4 > 60 * 1024) return DV_EUNSUPPORTED;
  DISPATCH_T(d->dtype, {
    constexpr int V = DT<T>::VEC;
    if (lddx % V) return DV_EALIGN;
    const int64_t total = (int64_t)a.N * a.Ti * a.Hi * a.Wi * (a.CP / V);
    hipLaunchKernelGGL((bn_bwd_apply_maxpool_kernel<T>), dim3(grid_for(total, 2048)), dim3(kThreads), 5 * a.CP * sizeof(float),
                       ST(stream), a, (const T*)dy_pool, idx, (const T*)x, mean, invstd, gamma, scale, shift, sums_global,
                       rep_global, inv_count, dparam_scale, dgamma, dbeta, (T*)dx, lddx, (uint32_t)total, a.C, a.CP);
  });
  return dv_launch_status();
}

// channel chunks per sample (16-byte vectors per chunk): ~1024 workgroups in all, at least 16 vectors (256 bytes of a row) each,
// and one chunk when a sample has fewer than 128 positions (the row groups of one workgroup cover it)
static int c_chunk_vecs(int N, int S, int CV) {
  int chunks = (1024 + N - 1) / N;
  const int cap = (CV + 15) / 16;
  if (chunks > cap) chunks = cap;
  if (chunks < 1 || S < 128) chunks = 1;
  return (CV + chunks - 1) / chunks;
}

extern "C" int dv_spatial_mean(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t S, int32_t C, float* out,
                               void* stream) {
  const int CP = cp8(C);
  if (!x || !out || N <= 0 || S <= 0 || C <= 0 || ldx < CP) return DV_EINVAL;
  if (!aligned16(x)) return DV_EALIGN;
  DISPATCH_T(dtype, {
    if (ldx % DT<T>::VEC) return DV_EALIGN;
    const int CV = CP / DT<T>::VEC, ccv = c_chunk_vecs(N, S, CV);
    hipLaunchKernelGGL((spatial_mean_kernel<T>), dim3(N, (CV + ccv - 1) / ccv), dim3(kThreads), 0, ST(stream), (const T*)x, ldx, S,
                       C, CP, ccv, out);
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
    const int64_t total = (int64_t)N * S * (CP / V);
    if (total >= (1ll << 31)) return DV_EINVAL;
    hipLaunchKernelGGL((rowscale_kernel<T, 2>), dim3(grid_for(total)), dim3(kThreads), 0, ST(stream),
                       (const T*)nullptr, 0, (const float*)nullptr, dout, (uint32_t)total, make_fastdiv((uint32_t)(CP / V)),
                       make_fastdiv((uint32_t)S), C, (T*)dx, lddx, (flags & DV_ACCUM) ? 1 : 0);
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
    const int64_t total = (int64_t)N * S * (CP / V);
    if (total >= (1ll << 31)) return DV_EINVAL;
    hipLaunchKernelGGL((rowscale_kernel<T, 0>), dim3(grid_for(total)), dim3(kThreads), 0, ST(stream),
                       (const T*)x, ldx, g, (const float*)nullptr, (uint32_t)total, make_fastdiv((uint32_t)(CP / V)),
                       make_fastdiv((uint32_t)S), C, (T*)y, ldy, 0);
  });
  return dv_launch_status();
}

extern "C" int dv_gate_bwd_reduce(int32_t dtype, const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* g,
                                  int32_t N, int32_t S, int32_t C, float* dpre, int32_t x_is_output, void* stream) {
  const int CP = cp8(C);
  if (!dy || !x || !g || !dpre || N <= 0 || S <= 0 || C <= 0 || lddy < CP || ldx < CP) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(x)) return DV_EALIGN;
  DISPATCH_T(dtype, {
    if (lddy % DT<T>::VEC || ldx % DT<T>::VEC) return DV_EALIGN;
    const int CV = CP / DT<T>::VEC, ccv = c_chunk_vecs(N, S, CV);
    hipLaunchKernelGGL((gate_bwd_reduce_kernel<T>), dim3(N, (CV + ccv - 1) / ccv), dim3(kThreads), 0, ST(stream), (const T*)dy,
                       lddy, (const T*)x, ldx, g, S, C, CP, ccv, dpre, x_is_output);
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
    const int64_t total = (int64_t)N * S * (CP / V);
    if (total >= (1ll << 31)) return DV_EINVAL;
    hipLaunchKernelGGL((rowscale_kernel<T, 1>), dim3(grid_for(total)), dim3(kThreads), 0, ST(stream),
                       (const T*)dy, lddy, g, dmean, (uint32_t)total, make_fastdiv((uint32_t)(CP / V)),
                       make_fastdiv((uint32_t)S), C, (T*)dx, lddx, (flags & DV_ACCUM) ? 1 : 0);
  });
  return dv_launch_status();
}

extern "C" int dv_colsum_f32(const float* x, int32_t ldx, int32_t R, int32_t C, float* out, void* stream) {
  if (!x || !out || R <= 0 || C <= 0 || ldx < C) return DV_EINVAL;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + 31) / 32), dim3(256), 0, ST(stream), x, (int64_t)ldx, R, C, out, 1);
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

// ------------------------------------------------------------------------------------------
// fp8 quantisation for the pointwise-conv GEMMs (dv_conv3d_fwd_fp8 / dv_conv3d_dgrad_fp8): per-tensor scaling,
//   scale = amax / FMAX (448 for e4m3, 57344 for e5m2; 1 when the tensor is all zero),  q = fp8(x / scale)  (RNE, finite).
// Two launches and no atomics: block maxima -> every block of the second launch folds them (<= 1024 floats) and converts.
#include <hip/hip_fp8.h>
namespace {

constexpr int kAmaxBlocks = 512;

template <typename T>
__global__ void __launch_bounds__(256) amax_partials_kernel(const T* __restrict__ x, int64_t M, int C, int ld, float* __restrict__ part) {
  constexpr int V = DT<T>::VEC;
  const int vpr = C / V;
  const int64_t total = M * vpr;
  float m = 0.f;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / vpr;
    const int cv = (int)(i - r * vpr);
    float v[V];
    Pack16<T>::load(x + r * ld + cv * V, v);
#pragma unroll
    for (int e = 0; e < V; ++e) m = fmaxf(m, fabsf(v[e]));
  }
  m = wave_max(m);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

template <typename T, int FMT>
__global__ void __launch_bounds__(256) quantize_fp8_kernel(const T* __restrict__ x, int64_t M, int C, int ld, const float* __restrict__ part,
                                                           int nparts, unsigned char* __restrict__ q, int ldq, float* __restrict__ scale_out) {
  __shared__ float s_amax;
  {
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
    m = wave_max(m);
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) s_amax = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
  }
  constexpr float FMAX = FMT == 0 ? 448.f : 57344.f;
  const float amax = s_amax;
  const float scale = amax > 0.f ? amax / FMAX : 1.f;
  const float inv = 1.f / scale;
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = scale;
  const int vpr = C / 16;                                       // 16 fp8 per 16-byte store
  const int64_t total = M * vpr;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / vpr;
    const int cv = (int)(i - r * vpr);
    union { unsigned char b[16]; uint4 u; } o;
    constexpr int V = DT<T>::VEC;
#pragma unroll
    for (int hv = 0; hv < 16 / V; ++hv) {
      float v[V];
      Pack16<T>::load(x + r * ld + cv * 16 + hv * V, v);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float t = fminf(fmaxf(v[e] * inv, -FMAX), FMAX);
        o.b[hv * V + e] = __hip_cvt_float_to_fp8(t, __HIP_SATFINITE, FMT == 0 ? __HIP_E4M3 : __HIP_E5M2);
      }
    }
    *reinterpret_cast<uint4*>(q + r * ldq + cv * 16) = o.u;
  }
}

}  // namespace

extern "C" int dv_quantize_fp8_workspace(void) { return kAmaxBlocks * 4; }

extern "C" int dv_quantize_fp8(int32_t dtype, const void* x, int64_t M, int32_t C, int32_t ld, int32_t fmt, void* q, int32_t ldq,
                               float* scale_out, float* workspace, void* stream) {
  if (!x || !q || !scale_out || !workspace || M <= 0 || C <= 0 || C % 16 || ld < C || ldq < C || ldq % 16) return DV_EINVAL;
  if (dtype != DV_F32 && dtype != DV_BF16) return DV_EUNSUPPORTED;
  if (fmt != 0 && fmt != 1) return DV_EINVAL;
  const int es = dtype == DV_F32 ? 4 : 2;
  if (!aligned16(x) || !aligned16(q) || (ld * es) % 16) return DV_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int64_t vecs = M * (C / 16);
  const int blocks = (int)(vecs / 256 + 1 < kAmaxBlocks ? vecs / 256 + 1 : kAmaxBlocks);
#define DV_Q(T_, F_)                                                                                                         \
  do {                                                                                                                       \
    hipLaunchKernelGGL((amax_partials_kernel<T_>), dim3(blocks), dim3(256), 0, st, (const T_*)x, M, C, ld, workspace);        \
    hipLaunchKernelGGL((quantize_fp8_kernel<T_, F_>), dim3(blocks * 2), dim3(256), 0, st, (const T_*)x, M, C, ld, workspace,  \
                       blocks, (unsigned char*)q, ldq, scale_out);                                                          \
  } while (0)
  if (dtype == DV_F32) { if (fmt == 0) DV_Q(float, 0); else DV_Q(float, 1); }
  else { if (fmt == 0) DV_Q(bf16_t, 0); else DV_Q(bf16_t, 1); }
#undef DV_Q
  return dv_launch_status();
}
