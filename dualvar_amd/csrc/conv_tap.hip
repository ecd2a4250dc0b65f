// LDS-staged input tiles for the separable convolutions (fp32 split mode): the stride-1 "same" 1x3x3 and 3x1x1 convs of
// backbone/s3dg.py:30-65 (STConv3d: 1xkxk -> BN -> ReLU -> kx1x1) and backbone/r21d.py:11-70 -- forward and data gradient.
//
// conv_gemm_kernel (conv.hip) is a per-tap implicit GEMM: every (tap, 16-channel) K tile costs one global -> LDS DMA round of the
// activation tile and, in the fp32 mode, one split of its fragments into bf16 triples; the round-3 ablation priced those two at
// 172 of the 523 us of Conv_2c's 1x3x3 forward, and nothing overlapped the matrix pipe.  Here a workgroup stages the 16-channel
// slab of its 256 output positions ONCE per channel chunk -- with the halo its taps reach --, splits it ONCE into bf16 planes in
// MFMA-fragment order, and every tap's A fragments are then plain ds_read_b128 at a row offset into those planes:
//   * spatial (1 x kh x kw): the tile is 256 consecutive tensor rows; the planes hold rows [m0 - HALO, m0 + 256 + HALO), HALO =
//     ph * W + pw.  A tap is the row offset dh' * W + dw'.  Rows of the image border (a tap that leaves the plane) read a shared
//     48-byte ZERO slot instead: one v_cndmask on the LDS address per (row block, tap), no arithmetic on the data, no padded
//     staging layout -- positions are tensor rows, tiles may straddle planes and clips.
//   * temporal (kt x 1 x 1): the tile is P = 256 / T pixels x all T frames (T = 2, 4, 8), no halo; a tap is the planes of the
//     neighbouring frame, and the (frame, tap) pairs that leave the clip are not multiplied at all (wave-uniform skip): 10 of
//     12 for 3x1x1 on four frames.
// Split instructions and activation loads per MFMA drop by the number of taps (x 256 / (256 + 2 HALO) for the halo); the
// weights, pre-split in fragment order (dv_pack_w3), are streamed per (chunk, tap) by LDS DMA exactly as in conv_gemm_kernel.
// LDS: NS x BN x 96 B of weight stages + (256 + 2 HALO) x 96 B of planes + the zero slot = 43 .. 54 KB: three workgroups per CU
// (round 3 measured co-residency worth 1.36x on these kernels: the reason the 96 KB prototype of that round lost).
//
// Planes layout: [k half h][position][hi | mid | lo][8 bf16] = 48 B per (h, position): lane (l31, h) of a 32-row block reads
// position base + l31 -- stride 48 B, conflict-free for ds_read_b128 at ANY base (3 is odd: the 16 lanes of a read group hit 16
// different 16-byte slots), so a tap offset costs nothing.
//
// Variants of the one kernel (template parameters): BM = 128-row tiles for launches of less than one full round of workgroups (the
// 12 544-row levels; tap_bm); BNL = the BatchNorm (+ReLU) in front of the conv applied to the staged slab (dv_conv3d_fwd_bn_in: the
// activation between a 1xkxk -> kx1x1 pair is never written); and, further down, conv_pp_fwd_kernel: the RGB stem's forward as
// tiles of whole output lines over pixel pairs.
//
// Activations go global -> VGPR -> split -> LDS (not LDS-DMA: the fp32 staging copy would cost 16 - 24 KB of LDS per stage and a
// resident workgroup).  The loads of chunk c + 1 are issued at the start of chunk c and consumed at its end; they are inline
// assembly with their own counted s_waitcnt so that the compiler's wait-count pass, which cannot see the LDS-DMA weight pieces,
// does not drain the weight pipeline in front of them.
#include "conv_common.hpp"

namespace {

struct TapArgs {
  const void* src;        // x (forward) or dY (data gradient): [M][lds_] fp32
  const void* w;          // pre-split weights (dv_pack_w3): forward layout for fwd, dgrad layout for dgrad
  void* out;              // [M][ldo] fp32
  float* stats;           // forward + DV_STATS: [2][N][ceil(M / 256)] BatchNorm partials
  int M, N, NP;           // rows (stride 1, same: input rows == output rows), real columns, columns to write
  int lds_, ldo, ldw;     // pitches in elements; ldw = padded row count of the pre-split weights
  int ntn, flags;
  int src_bytes, w_bytes, out_bytes;
  int CP;                 // K per tap (channel pitch of src), multiple of 16
  int npos, npp, halo;    // staged positions per tile, pitch of a k-half plane; positions in front of the tile's first row (spatial)
  int sgn;                // +1 forward, -1 data gradient (dX[m] = sum_d dY[m + p - d] W_d): a tap's displacement is sgn * (d - p)
  int H, W;
  FastDiv fW, fH;
  int T, S, P, lgP, NQ;   // temporal: frames, pixels per frame, pixels per tile (256 / T), log2 P, N * S
  FastDiv fS;
  // temporal, general form (the parity classes of a t-strided data gradient, conv.hip ConvArgs::cls_on == 1: every class is a
  // dense stride-1 problem over T frames of dY): tap j reads frame f + sgn * (j - pt); its weights sit at the ORIGINAL tap
  // wt0 + wts * j; row frame f is frame f * ofs + ofo of an output tensor with oT frames
  int pt, wt0, wts, oT, ofs, ofo;
  // data gradient that is dL/dy of y = relu(BatchNorm(bn_x)) and its only contribution (dv_conv3d_dgrad_bn_ws): the epilogue
  // also forms the BatchNorm backward's sum(g), sum(g * xhat) over the rows it stores -- from the accumulators and ONE read of
  // bn_x, instead of dv_bn_bwd_reduce's second read of dL/dy -- as per-tile rows in bn_ws, folded in tile order by the
  // workgroups that take the last tickets (no float atomics; += into bn_sums [2][bn_cpb], which the caller zeroed)
  const void* bn_x;
  const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift;
  float *bn_sums, *bn_ws;
  int bn_ldx, bn_mask, bn_bytes, bn_cpb;
  // forward whose input is y = [relu](src * in_scale + in_shift), the BatchNorm in front of this conv (dv_conv3d_fwd_bn_in; BNL
  // instantiation, temporal form): applied to the staged slab in front of the split
  const float *in_scale, *in_shift;
  int in_C, in_relu;
};

// rows / tickets of the ordered fold (the protocol of elementwise.hip's ordered_fold: sc1 stores -> s_waitcnt vmcnt(0) -> barrier ->
// one agent-scope ticket -> the last arriver reads with sc1 loads; validated on gfx950, see the comment there)
constexpr int kBnFoldGroup = 32, kBnRow = 128;                  // tiles per group; floats per row: [sum g | sum g xhat] x 64 columns
// Workspace layout: [kBnTickWords ticket words][per column tile: n_mt tile rows, ngrp group rows].  The tickets have a region of
// their own that NO launch ever stores data into: the workspace is shared by launches of different shapes (and by the passes of
// a step), whose row regions overlap -- with the tickets behind each column tile's rows (the first version of this round) a
// launch's partial sums landed on the ticket words of another shape's layout, and from the second pass through a plan on the
// folds ran on garbage counts (gradients 10 % off; tools/probes/fused_reduce_twice.py, test_repeated_passes_...).
constexpr int kBnTickWords = 16384;
__host__ __device__ __forceinline__ int bn_ws_tick_stride(int n_mt) { return (((n_mt + kBnFoldGroup - 1) / kBnFoldGroup) + 1 + 7) & ~7; }
static inline int64_t bn_ws_floats_per_coltile(int n_mt) {
  const int ngrp = (n_mt + kBnFoldGroup - 1) / kBnFoldGroup;
  return (int64_t)(n_mt + ngrp) * kBnRow;
}
__device__ __forceinline__ size_t bn_ws_floats_per_coltile_dev(int n_mt) {
  const int ngrp = (n_mt + kBnFoldGroup - 1) / kBnFoldGroup;
  return (size_t)(n_mt + ngrp) * kBnRow;
}
__device__ __forceinline__ void tap_coherent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float tap_coherent_load(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// displacement of tap t (compile-time index, natural order (dh, dw) / dt; 3 x 3 and 3 x 1 x 1 windows, padding 1): a handful of
// scalar instructions from two kernel arguments -- a table in the argument block was re-loaded (s_load + lgkmcnt(0), which also
// waits for every LDS read in flight) inside the hot loop
template <int KIND> __device__ __forceinline__ int tap_dh(int t) { return KIND == 0 ? t / 3 - 1 : 0; }
template <int KIND> __device__ __forceinline__ int tap_dw(int t) { return KIND == 0 ? t % 3 - 1 : 0; }
template <int KIND> __device__ __forceinline__ int tap_off(const int sgn, const int W, int t, int pt = 1) {
  return KIND == 0 ? sgn * (tap_dh<0>(t) * W + tap_dw<0>(t)) : sgn * (t - pt);
}

// one 16-byte global load into registers, invisible to the compiler's wait-count pass (see the header comment)
__device__ __forceinline__ void gload16(f32x4& dst, dma_rsrc_t rsrc, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void gload16_hi(f32x4& dst, dma_rsrc_t rsrc, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// KIND 0: spatial (1 x kh x kw, NTAPS = kh * kw);  KIND 1: temporal (kt x 1 x 1, NTAPS = kt)
// NU: staging units (8 channels of one position = 32 B) per thread: ceil(2 * npos / 256)
// BNL: BatchNorm (+ReLU) of the layer in front applied to the staged slab (coefficients in an LDS table behind the zero slot)
// BM: rows per tile, 256 (a wave owns 64 rows) or 128 (32): the 128-row form is for launches of less than ~1.5 workgroups per CU
// in the 256-row form (the 12 544-row levels): twice the workgroups, two or three of them resident per CU instead of one
template <int KIND, int NTAPS, int BN, int NS, int NU, bool BNL = false, int BM = 256>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, BM == 128 ? 4 : 3))) void conv_tap_kernel(TapArgs a) {
  static_assert(!BNL || KIND == 1, "BatchNorm on load: temporal form (rows past the end never share a tile row with valid ones)");
  static_assert(BM == 256 || BM == 128, "tile height");
  constexpr int NW = 4, TM = BM / 128, WR = BM / NW, TN = BN / 32;       // WR: rows of a wave
  constexpr int B_BYTES = BN * 96, BPC = B_BYTES / 1024;       // one weight stage: hi|mid|lo of both k halves, in 1 KiB DMA pieces
  static_assert(B_BYTES % 1024 == 0, "weight stage in 1 KiB pieces");
  constexpr int NPW = (BPC + NW - 1) / NW;                      // pieces per wave and stage (the last round may be partial)
  constexpr int D = NS - 1;                                     // prefetch distance of the weight pipeline (steps)
  static_assert(D >= 1 && NTAPS > D, "the staging loads of a chunk are complete once tap D of that chunk has been waited for");
  constexpr int NSL = 2 * NU;                                   // staging loads per thread and chunk
  static_assert((D - 1) * NPW + NSL <= 24, "counted wait");
  constexpr unsigned kOOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(1024))) unsigned char dsm[];
  const int npos = a.npos;
  // positions between the two k-half planes: npos rounded up to 4 mod 8 -- a staging store instruction (ds_write_b128: groups of 8
  // lanes, 32 banks) covers four positions x both halves, 48 B apart within a half; with the halves 3 * npp = 4 mod 8 slots apart
  // the eight 16-byte slots of a group are all different (PMC of the first version: 22 % of the LDS cycles were conflicts)
  const int npp = a.npp;
  const unsigned planes_off = NS * B_BYTES;                     // [2][npp][48]
  const unsigned zero_off = planes_off + 2u * (unsigned)npp * 48u;        // 64 bytes of zeros
  const unsigned coef_off = zero_off + 64u;                               // BNL: [scale | shift][CP] floats

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int lbid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = lbid % a.ntn, tile_m = lbid / a.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const dma_rsrc_t src_rs = dma_make_rsrc(a.src, (unsigned)a.src_bytes), w_rs = dma_make_rsrc(a.w, (unsigned)a.w_bytes);
  const unsigned ldb = (unsigned)a.lds_ * 4u;
  const unsigned dsm_base = lds_addr(dsm);
  const int tsgn = a.sgn, tW = a.W, tT = a.T, tP48 = a.P * 48, tpt = a.pt;

  // ---- staging roles: unit u = tid + 256 k -> (position u >> 1, channel half u & 1)
  unsigned soff[NU], pwr[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int u = tid + 256 * k, pidx = u >> 1, half = u & 1;
    bool ok = pidx < npos;
    unsigned row = 0;
    if constexpr (KIND == 0) {
      const int r = m0 - a.halo + pidx;
      ok = ok && r >= 0 && r < a.M;
      row = (unsigned)r;
    } else {
      const int f = pidx >> a.lgP, p = pidx & (a.P - 1);
      const unsigned q = (unsigned)(tile_m * a.P + p);
      ok = ok && (int)q < a.NQ;
      const unsigned n = fd_div(q, a.fS);
      row = q + n * (unsigned)((a.T - 1) * a.S) + (unsigned)(f * a.S);
    }
    soff[k] = ok ? row * ldb + (unsigned)half * 32u : kOOB;
    pwr[k] = pidx < npos ? planes_off + (unsigned)(half * npp + pidx) * 48u : 0xffffffffu;
  }
  f32x4 sreg[NU][2];
  auto stage_issue = [&](int c0) {
    const unsigned so = (unsigned)c0 * 4u;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      gload16(sreg[k][0], src_rs, soff[k], so);
      gload16_hi(sreg[k][1], src_rs, soff[k], so);
    }
  };
  // split the staged slab (registers -> bf16 triples), then store the triples into the planes (the caller has made sure that
  // nobody reads the planes any more)
  Split3 s3r[NU];
  const float bn_lo = (BNL && !a.in_relu) ? -__builtin_inff() : 0.f;
  auto stage_split = [&](int c0) {
    float sc[8], sh[8];
    if constexpr (BNL) {                                        // every unit of this thread is channel half tid & 1 of the chunk
      const f32x4* cs = reinterpret_cast<const f32x4*>(dsm + coef_off + (unsigned)(c0 + 8 * (tid & 1)) * 4u);
      const f32x4* ch = reinterpret_cast<const f32x4*>(dsm + coef_off + (unsigned)(a.CP + c0 + 8 * (tid & 1)) * 4u);
      const f32x4 s0 = cs[0], s1 = cs[1], h0 = ch[0], h1 = ch[1];
      sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
      sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
    }
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      float v[8] = {sreg[k][0].x, sreg[k][0].y, sreg[k][0].z, sreg[k][0].w, sreg[k][1].x, sreg[k][1].y, sreg[k][1].z, sreg[k][1].w};
      if constexpr (BNL) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], bn_lo);      // dv_bn_apply's expression; bn_lo = 0 | -inf
      }
      s3r[k] = split3(v);
    }
  };
  auto stage_write = [&]() {
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      if (pwr[k] != 0xffffffffu) {
        bf16x8* dst = reinterpret_cast<bf16x8*>(dsm + pwr[k]);
        dst[0] = s3r[k].hi; dst[1] = s3r[k].mid; dst[2] = s3r[k].lo;
      }
    }
  };
  // the staging loads are older than every weight piece still in flight at the points this is called from: <= 2 * NPW pieces
  auto stage_wait = [&]() {
    static_assert(NU <= 3, "operand list");
    if constexpr (NU == 1) asm volatile("s_waitcnt vmcnt(4)" : "+v"(sreg[0][0]), "+v"(sreg[0][1])::"memory");
    else if constexpr (NU == 2)
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(sreg[0][0]), "+v"(sreg[0][1]), "+v"(sreg[1][0]), "+v"(sreg[1][1])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(4)"
                   : "+v"(sreg[0][0]), "+v"(sreg[0][1]), "+v"(sreg[1][0]), "+v"(sreg[1][1]), "+v"(sreg[2][0]), "+v"(sreg[2][1])::"memory");
  };
  static_assert(2 * NPW <= 4, "stage_wait's count");

  // ---- weight pieces of this wave: LDS byte o of a stage <- W3 byte ((K tile * 2 + hh) * ldw + n0) * 48 + (o mod BN * 48)
  unsigned woff[NPW];
  int pieces = 0;
#pragma unroll
  for (int u = 0; u < NPW; ++u) {
    const int p = wave + NW * u;
    const unsigned o = (unsigned)p * 1024u + (unsigned)lane * 16u;
    const unsigned hh = o / (BN * 48u), rem = o - hh * (BN * 48u);
    woff[u] = (hh * (unsigned)a.ldw + (unsigned)n0) * 48u + rem;
    pieces += p < BPC ? 1 : 0;
  }
  const int NC = a.CP >> 4, nsteps = NC * NTAPS;
  auto w_issue = [&](int step, int stage) {
    const int c = step / NTAPS, t = step - c * NTAPS;
    const unsigned kt = (unsigned)((a.wt0 + a.wts * t) * a.CP + c * 16) >> 4;
#pragma unroll
    for (int u = 0; u < NPW; ++u)
      if (wave + NW * u < BPC)
        dma_load16(w_rs, dsm_base + stage * B_BYTES + (wave + NW * u) * 1024, woff[u] + kt * (unsigned)a.ldw * 96u);
  };

  // ---- A-fragment addressing
  unsigned abase[TM];          // LDS byte offset of (h, position of this lane's row in block i) -- spatial: at tap offset 0
  unsigned amask[TM];          // spatial: bit t set = tap t leaves the image for this lane's row
  int bfr[TM], bpb[TM];        // temporal: frame and 32-pixel block of row block i (wave-uniform)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    amask[i] = 0; bfr[i] = 0; bpb[i] = 0;
    if constexpr (KIND == 0) {
      const int rloc = WR * wave + 32 * i + l31;
      abase[i] = planes_off + (unsigned)(h * npp + a.halo + rloc) * 48u;
      uint32_t q_, ww, hh_, q2;
      fd_divmod((uint32_t)(m0 + rloc), a.fW, q_, ww);
      fd_divmod(q_, a.fH, q2, hh_);
      unsigned inv = 0;
#pragma unroll
      for (int t = 0; t < NTAPS; ++t)
        inv |= (((unsigned)((int)hh_ + a.sgn * tap_dh<0>(t)) >= (unsigned)a.H || (unsigned)((int)ww + a.sgn * tap_dw<0>(t)) >= (unsigned)a.W) ? 1u : 0u) << t;
      amask[i] = inv;
    } else {
      // balanced (frame, pixel block) pairs per wave: border frames have a tap less
      if constexpr (BM == 128) {           // one (frame, 32-pixel block) per wave: T = 2 -> P = 64, T = 4 -> P = 32
        if (a.T == 2) { bfr[i] = wave & 1; bpb[i] = wave >> 1; }
        else { bfr[i] = wave; bpb[i] = 0; }
      }
      else if (a.T == 2) { bfr[i] = i; bpb[i] = wave; }
      else if (a.T == 4) {                 // frames {0, 2} and {1, 3} pair up: equal tap counts for 3 taps / padding 1 and 4 / 2
        const int f0 = ((wave & 1) << 1) | (wave >> 1);
        bfr[i] = i == 0 ? f0 : (f0 ^ 2); bpb[i] = i;
      }
      else { bfr[i] = i == 0 ? wave : 7 - wave; bpb[i] = 0; }
      abase[i] = planes_off + (unsigned)(h * npp + bpb[i] * 32 + l31) * 48u;
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned bbase = (unsigned)(h * BN + l31) * 48u;
  // `issue`: the weight DMA of step + D, called once this step's first fragment reads are on their way (the DMA's address
  // arithmetic and m0 traffic then sit under the LDS latency instead of in front of it)
  auto compute = [&](int t, int stage, auto&& issue) {
    Split3 bf[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bf16x8* pb = reinterpret_cast<const bf16x8*>(dsm + stage * B_BYTES + bbase + j * (32 * 48));
      bf[j].hi = pb[0]; bf[j].mid = pb[1]; bf[j].lo = pb[2];
    }
    bool issued = false;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned ad;
      if constexpr (KIND == 0) {
        const unsigned at = abase[i] + (unsigned)(tap_off<0>(tsgn, tW, t) * 48);
        ad = ((amask[i] >> t) & 1u) ? zero_off : at;
      } else {
        const int fs = bfr[i] + tap_off<1>(tsgn, tW, t, tpt);
        if (fs < 0 || fs >= tT) continue;                      // (wave-uniform) this tap leaves the clip for the whole block
        ad = abase[i] + (unsigned)(fs * tP48);
      }
      const bf16x8* pa = reinterpret_cast<const bf16x8*>(dsm + ad);
      Split3 af;
      af.hi = pa[0]; af.mid = pa[1]; af.lo = pa[2];
      if (!issued) { issue(); issued = true; }
#pragma unroll
      for (int j = 0; j < TN; ++j) mma_split3(af, bf[j], acc[i][j]);
    }
    if (!issued) issue();
  };

  // ---- prologue: chunk 0 into the planes, the zero slot, the first D weight stages
  stage_issue(0);
  if (tid < 4) *reinterpret_cast<f32x4*>(dsm + zero_off + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (BNL) {
    float* tab = reinterpret_cast<float*>(dsm + coef_off);
    for (int c = tid; c < a.CP; c += 256) {
      tab[c] = c < a.in_C ? a.in_scale[c] : 0.f;
      tab[a.CP + c] = c < a.in_C ? a.in_shift[c] : 0.f;
    }
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  stage_wait();
  stage_split(0);
  stage_write();
  for (int s = 0; s < D && s < nsteps; ++s) w_issue(s, s);
  int cur = 0, nxt = D % NS;
  for (int c = 0; c < NC; ++c) {
    const bool more = c + 1 < NC;
    if (more) stage_issue((c + 1) * 16);
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const int step = c * NTAPS + t;
      // stage `cur` (step) has landed: younger than it are the next D - 1 steps' pieces and, for the first D taps of a chunk,
      // the staging loads of the next chunk (issued at the chunk's start, after the pieces of step c * NTAPS + D - 1)
      dma_wait_upto(min(D - 1, nsteps - 1 - step) * pieces + ((more && t < D) ? NSL : 0));
      __syncthreads();                         // ... everybody's; stage `nxt` (step - 1) is free; t == 0: the planes are published
      compute(t, cur, [&] { if (step + D < nsteps) w_issue(step + D, nxt); });
      cur = cur + 1 == NS ? 0 : cur + 1;
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
    if (more) {
      stage_wait();
      __syncthreads();                         // every wave has read its last fragments of this chunk's planes
      stage_split((c + 1) * 16);               // (splitting in front of the barrier instead was measured equal)
      stage_write();
    }
  }

  // ---------------- epilogue: fp32 tiles straight from the accumulators (the row term sits in voffset: range-checked)
  float* red = reinterpret_cast<float*>(dsm);                   // statistics scratch: [NW][BN] + [BN] floats in the weight stages
  const int flags = a.flags;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
  const unsigned ldo4 = (unsigned)a.ldo * 4u;
  const bool bn_on = a.bn_x != nullptr;
  const __amdgpu_buffer_rsrc_t bn_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(bn_on ? a.bn_x : a.out), 0,
                                                                           bn_on ? a.bn_bytes : a.out_bytes, 0x00020000);
  const unsigned bn_ldx4 = (unsigned)a.bn_ldx * 4u;
  float bn_mu[TN], bn_is[TN], bn_sc[TN], bn_sh[TN], bn_s1[TN], bn_s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + j * 32 + l31;
    const bool ok = bn_on && col < a.N;
    bn_mu[j] = ok ? a.bn_mean[col] : 0.f;
    bn_is[j] = ok ? a.bn_invstd[col] : 0.f;
    bn_sc[j] = (ok && a.bn_mask) ? a.bn_scale[col] : 0.f;
    bn_sh[j] = (ok && a.bn_mask) ? a.bn_shift[col] : 0.f;
    bn_s1[j] = bn_s2[j] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    unsigned ro[16], rox[16];                                   // byte offsets of this lane's rows in the output / in bn_x
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
      unsigned row;
      bool ok;
      if constexpr (KIND == 0) {
        const int m = m0 + WR * wave + 32 * i + rl;
        ok = m < a.M; row = (unsigned)m;
      } else {
        const unsigned q = (unsigned)(tile_m * a.P + bpb[i] * 32 + rl);
        const unsigned n = fd_div(q, a.fS);
        ok = (int)q < a.NQ;
        row = q + n * (unsigned)((a.oT - 1) * a.S) + (unsigned)((bfr[i] * a.ofs + a.ofo) * a.S);
      }
      ro[r] = ok ? row * ldo4 : kOOB;
      rox[r] = ok ? row * bn_ldx4 : kOOB;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + j * 32 + l31;
      const bool col_ok = col < a.NP;                           // (columns [N, NP) are pad lanes: their weights are zero rows)
      const unsigned cb = (unsigned)col * 4u;
      unsigned old[16];
      if ((flags & DV_ACCUM) && col_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = __builtin_amdgcn_raw_buffer_load_b32(orsrc, (int)(ro[r] + cb), 0, 0);
      }
      if (bn_on) {                                              // the BatchNorm's input at this lane's 16 rows of column `col`
        const unsigned xb = (unsigned)col * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          old[r] = col_ok ? __builtin_amdgcn_raw_buffer_load_b32(bn_rsrc, (int)(rox[r] + xb), 0, 0) : 0u;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = ro[r] == kOOB ? 0.f : acc[i][j][r];           // rows past the end feed neither the output nor the statistics
        acc[i][j][r] = v;
        if (col_ok) {
          if (flags & DV_ACCUM) v += __builtin_bit_cast(float, old[r]);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, (int)(ro[r] + cb), 0, 0);
        }
        if (bn_on) {
          const float xv = __builtin_bit_cast(float, old[r]);
          const float act = xv * bn_sc[j] + bn_sh[j];          // the forward's expression (dv_bn_apply), same rounding
          const float gg = (a.bn_mask && !(act > 0.f)) ? 0.f : v;
          bn_s1[j] += gg;
          bn_s2[j] += gg * (xv - bn_mu[j]) * bn_is[j];
        }
      }
    }
  }
  if (bn_on) {
    // ---- this tile's row of partial sums (fixed order: registers, wave halves, waves), then the ordered fold over tiles
    constexpr int G = kBnFoldGroup;
    static_assert(BN == 64, "a row of partial sums is [sum g | sum g xhat] x 64 columns");
    float* redf = reinterpret_cast<float*>(dsm);                // [NW][2][BN] + the row [2][BN]; one flag word behind it
    __syncthreads();                                            // the weight stages are free
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s1 = bn_s1[j], s2 = bn_s2[j];
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (h == 0) { redf[(wave * 2 + 0) * BN + j * 32 + l31] = s1; redf[(wave * 2 + 1) * BN + j * 32 + l31] = s2; }
    }
    __syncthreads();
    const int n_mt = (a.M + BM - 1) / BM, ngrp = (n_mt + G - 1) / G, grp = tile_m / G;
    const int gsize = min(G, n_mt - grp * G);
    float* rows = a.bn_ws + kBnTickWords + (size_t)tile_n * bn_ws_floats_per_coltile_dev(n_mt);
    float* grows = rows + (size_t)n_mt * kBnRow;
    unsigned* tick = reinterpret_cast<unsigned*>(a.bn_ws) + (size_t)tile_n * bn_ws_tick_stride(n_mt);      // [ngrp group tickets | 1]
    unsigned* lastf = reinterpret_cast<unsigned*>(redf + (NW * 2 + 2) * BN);
    const int which = tid >> 6, cc = tid & 63;                  // tid < 128: (sum index, column)
    const bool mine = tid < kBnRow;
    if (mine) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += redf[(w * 2 + which) * BN + cc];
      tap_coherent_store(rows + (size_t)tile_m * kBnRow + tid, t);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this tile's row has reached the coherence point ...
    __syncthreads();
    if (tid == 0) *lastf = (atomicAdd(tick + grp, 1u) == (unsigned)gsize - 1u) ? 1u : 0u;       // ... before its ticket is taken
    __syncthreads();
    if (!*lastf) return;
    auto fold = [&](const float* src, int n) -> float {         // rows src[0 .. n) of this thread's word, in row order
      float t = 0.f;
      for (int b = 0; b < n; b += G) {
        float v[G];
#pragma unroll
        for (int u = 0; u < G; ++u) v[u] = tap_coherent_load(src + (size_t)min(b + u, n - 1) * kBnRow + tid);
#pragma unroll
        for (int u = 0; u < G; ++u) t += (b + u < n) ? v[u] : 0.f;
      }
      return t;
    };
    auto finish = [&](float t) {                                // += into the caller's sums: [2][bn_cpb], real columns only
      const int c = n0 + cc;
      if (c < a.N) a.bn_sums[(size_t)which * a.bn_cpb + c] += t;
    };
    float t = mine ? fold(rows + (size_t)grp * G * kBnRow, gsize) : 0.f;
    if (ngrp == 1) {
      if (mine) finish(t);
      if (tid == 0) __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (mine) tap_coherent_store(grows + (size_t)grp * kBnRow + tid, t);
    if (tid == 0) __hip_atomic_store(tick + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *lastf = (atomicAdd(tick + ngrp, 1u) == (unsigned)ngrp - 1u) ? 1u : 0u;
    __syncthreads();
    if (!*lastf) return;
    if (mine) finish(fold(grows, ngrp));
    if (tid == 0) __hip_atomic_store(tick + ngrp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (flags & DV_STATS) {
    // per column: sum and M2 about the tile mean of the values as stored (two passes over the accumulators), [2][N][tiles]
    const int n_mt = (a.M + BM - 1) / BM;
    const int rows_here = min(BM, a.M - m0);
    float* meanb = red + NW * BN;
    __syncthreads();                                            // the weight stages are free
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
      s += __shfl_xor(s, 32);
      if (h == 0) red[wave * BN + j * 32 + l31] = s;
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * BN + tid];
      meanb[tid] = s / (float)rows_here;
      if (n0 + tid < a.N) a.stats[(size_t)(n0 + tid) * n_mt + tile_m] = s;
    }
    __syncthreads();
    // validity of this lane's rows again (ro[] is out of scope): bit r of vm[i]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float mu = meanb[j * 32 + l31];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;
          bool ok;
          if constexpr (KIND == 0) ok = m0 + WR * wave + 32 * i + rl < a.M;
          else ok = tile_m * a.P + bpb[i] * 32 + rl < a.NQ;
          const float dlt = acc[i][j][r] - mu;
          s += ok ? dlt * dlt : 0.f;
        }
      s += __shfl_xor(s, 32);
      if (h == 0) red[wave * BN + j * 32 + l31] = s;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.N) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * BN + tid];
      a.stats[(size_t)(a.N + n0 + tid) * n_mt + tile_m] = s;
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Pixel-pair stem form (forward): the network's first conv (backbone/s3dg.py:151 Conv_1a.conv1, 1x7x7 / stride 2 on RGB -- here a
// 1 x 7 x 4 window over 8-channel pixel pairs of the zero-bordered frames, stride (1, 2, 1), no padding).
// With 8 channels a 16-wide K tile straddles two taps, so conv_gemm ran it on the generic gather (per-lane tap decode, one DMA
// round and one split per K tile: 105 TFLOP/s).  Here a workgroup owns G consecutive OUTPUT LINES of one frame (G * Wo <= 256
// rows, 224 = 7 row blocks for 56-pixel lines) x 64 output channels and stages the 2 G + 5 input lines its windows reach ONCE
// -- global -> registers -> bf16 triples -> planes [line][pair][hi|mid|lo][8], 48 B per pair --: the whole K range (14 K tiles =
// 7 kernel rows x two pair-tap pairs) is then resident, a K tile's A fragment is three ds_read_b128 at
// (line 2 j + dh, pair wo + 2 (kk & 1) + h) -- consecutive lanes are consecutive pairs: conflict-free, no masks (the border is
// part of the ingest layout) --, and the K loop only streams the pre-split weights (LDS DMA, two stages) between barriers.
// Output rows of a tile are contiguous in the tensor; epilogue and BatchNorm partials as conv_tap_kernel (per G * Wo rows).
struct PpArgs {
  const void* src;        // [N * T * Hp lines][Wp pairs][8] fp32
  const void* w;          // pre-split weights (dv_pack_w3), K = ((dh * 4 + pair tap) * 8 + channel)
  void* out;              // [M][ldo] fp32
  float* stats;           // DV_STATS: [2][N][M / rows]
  int M, N, NP, ldo, ldw, ntn, flags;
  int src_bytes, w_bytes, out_bytes;
  int Wo, Wp, Hp, G, rows, npos;       // output pixels per line, pairs per input line, input lines per frame, lines per tile, G * Wo, (2 G + 5) * Wp
  FastDiv fWo, fWp, fHg;               // row -> (line, pixel); position -> (line, pair); tile -> (frame, line group)
  int Hg;                              // line groups per frame (Ho / G)
};

template <int NS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_pp_fwd_kernel(PpArgs a) {
  constexpr int BN = 64, NW = 4, TM = 2, TN = 2, NK = 14;
  constexpr int B_BYTES = BN * 96, BPC = B_BYTES / 1024, NPW = (BPC + NW - 1) / NW, D = NS - 1, NU = 3;
  constexpr unsigned kOOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(1024))) unsigned char dsm[];
  const unsigned planes_off = NS * B_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int lbid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = lbid % a.ntn, tile_m = lbid / a.ntn;
  const int n0 = tile_n * BN, m0 = tile_m * a.rows;
  const unsigned frame = fd_div((uint32_t)tile_m, a.fHg);
  const unsigned ho0 = ((unsigned)tile_m - frame * (unsigned)a.Hg) * (unsigned)a.G;
  const dma_rsrc_t src_rs = dma_make_rsrc(a.src, (unsigned)a.src_bytes), w_rs = dma_make_rsrc(a.w, (unsigned)a.w_bytes);
  const unsigned dsm_base = lds_addr(dsm);

  // ---- stage the input lines (the only activation traffic of this workgroup)
  f32x4 sreg[NU][2];
  const unsigned line0 = frame * (unsigned)a.Hp + 2u * ho0;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const unsigned p = (unsigned)(tid + 256 * k);
    uint32_t ln, pr;
    fd_divmod(p, a.fWp, ln, pr);
    const unsigned off = (int)p < a.npos ? ((line0 + ln) * (unsigned)a.Wp + pr) * 32u : kOOB;
    gload16(sreg[k][0], src_rs, off, 0u);
    gload16_hi(sreg[k][1], src_rs, off, 0u);
  }
  // ---- weight pieces of this wave (as conv_tap_kernel)
  unsigned woff[NPW];
  int pieces = 0;
#pragma unroll
  for (int u = 0; u < NPW; ++u) {
    const int p = wave + NW * u;
    const unsigned o = (unsigned)p * 1024u + (unsigned)lane * 16u;
    const unsigned hh = o / (BN * 48u), rem = o - hh * (BN * 48u);
    woff[u] = (hh * (unsigned)a.ldw + (unsigned)n0) * 48u + rem;
    pieces += p < BPC ? 1 : 0;
  }
  auto w_issue = [&](int kk, int stage) {
#pragma unroll
    for (int u = 0; u < NPW; ++u)
      if (wave + NW * u < BPC)
        dma_load16(w_rs, dsm_base + stage * B_BYTES + (wave + NW * u) * 1024, woff[u] + (unsigned)kk * (unsigned)a.ldw * 96u);
  };
  // ---- A-fragment addressing: row r = 32 b + l31 of the tile -> (line j, pixel wo); rows past the end read position 0
  const int nblk = (a.rows + 31) >> 5;
  unsigned abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = (wave + NW * i) * 32 + l31;
    uint32_t j, wo;
    fd_divmod((uint32_t)(r < a.rows ? r : 0), a.fWo, j, wo);
    abase[i] = planes_off + (2u * j * (unsigned)a.Wp + wo + (unsigned)h) * 48u;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int s = 0; s < D; ++s) w_issue(s, s);
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(sreg[0][0]), "+v"(sreg[0][1]), "+v"(sreg[1][0]), "+v"(sreg[1][1]), "+v"(sreg[2][0]), "+v"(sreg[2][1])::"memory");
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int p = tid + 256 * k;
    if (p < a.npos) {
      const float v[8] = {sreg[k][0].x, sreg[k][0].y, sreg[k][0].z, sreg[k][0].w, sreg[k][1].x, sreg[k][1].y, sreg[k][1].z, sreg[k][1].w};
      const Split3 q = split3(v);
      bf16x8* dst = reinterpret_cast<bf16x8*>(dsm + planes_off + (unsigned)p * 48u);
      dst[0] = q.hi; dst[1] = q.mid; dst[2] = q.lo;
    }
  }
  const unsigned bbase = (unsigned)(h * BN + l31) * 48u;
  const unsigned wp48 = (unsigned)a.Wp * 48u;
  int cur = 0, nxt = D % NS;
  for (int kk = 0; kk < NK; ++kk) {
    dma_wait_upto(min(D - 1, NK - 1 - kk) * pieces);          // stage `cur` has landed (this wave's pieces)
    __syncthreads();                                            // ... everybody's; stage `nxt` is free; kk == 0: the planes are published
    if (kk + D < NK) w_issue(kk + D, nxt);
    Split3 bf[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bf16x8* pb = reinterpret_cast<const bf16x8*>(dsm + cur * B_BYTES + bbase + j * (32 * 48));
      bf[j].hi = pb[0]; bf[j].mid = pb[1]; bf[j].lo = pb[2];
    }
    const unsigned koff = (unsigned)(kk >> 1) * wp48 + (unsigned)(kk & 1) * 96u;      // kernel row dh = kk / 2, pair taps 2 (kk & 1) + h
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (wave + NW * i >= nblk) continue;                      // (wave-uniform) 7 row blocks: the last wave carries one
      const bf16x8* pa = reinterpret_cast<const bf16x8*>(dsm + abase[i] + koff);
      Split3 af;
      af.hi = pa[0]; af.mid = pa[1]; af.lo = pa[2];
#pragma unroll
      for (int j = 0; j < TN; ++j) mma_split3(af, bf[j], acc[i][j]);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
    nxt = nxt + 1 == NS ? 0 : nxt + 1;
  }

  // ---------------- epilogue: fp32 tiles straight from the accumulators (row term in voffset: range-checked)
  const int flags = a.flags;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
  const unsigned ldo4 = (unsigned)a.ldo * 4u;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + j * 32 + l31;
      const unsigned cb = (unsigned)col * 4u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = (wave + NW * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool ok = rl < a.rows;
        const float v = ok ? acc[i][j][r] : 0.f;               // rows past the end feed neither the output nor the statistics
        acc[i][j][r] = v;
        if (ok && col < a.NP) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, (int)((unsigned)(m0 + rl) * ldo4 + cb), 0, 0);
      }
    }
  }
  if (flags & DV_STATS) {
    // per column: sum and M2 about the tile mean (two passes over the accumulators), [2][N][tiles]
    float* red = reinterpret_cast<float*>(dsm);                 // [NW][BN] + [BN] floats in the weight stages
    float* meanb = red + NW * BN;
    const int n_mt = a.M / a.rows;
    __syncthreads();                                            // the weight stages are free
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
      s += __shfl_xor(s, 32);
      if (h == 0) red[wave * BN + j * 32 + l31] = s;
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * BN + tid];
      meanb[tid] = s / (float)a.rows;
      if (n0 + tid < a.N) a.stats[(size_t)(n0 + tid) * n_mt + tile_m] = s;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float mu = meanb[j * 32 + l31];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = (wave + NW * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float dlt = acc[i][j][r] - mu;
          s += rl < a.rows ? dlt * dlt : 0.f;
        }
      s += __shfl_xor(s, 32);
      if (h == 0) red[wave * BN + j * 32 + l31] = s;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.N) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * BN + tid];
      a.stats[(size_t)(a.N + n0 + tid) * n_mt + tile_m] = s;
    }
  }
}

// lines per tile of the pixel-pair stem form for this problem, or 0 (not that form / does not fit)
static int pp_lines(const ConvArgs& a, int mode) {
  static const int on = getenv("DUALVAR_CONV_PP_FWD") ? atoi(getenv("DUALVAR_CONV_PP_FWD")) : 1;
  static const int any_size = (getenv("DUALVAR_CONV_TAP_GRID") ? atoi(getenv("DUALVAR_CONV_TAP_GRID")) : 128) <= 1;
  const ConvGeom& g = a.g;
  if (!on || mode != MODE_FWD || !(a.flags & DV_W3) || (a.flags & (DV_BIAS | DV_RELU | DV_SIGMOID | DV_ACCUM)) || a.out_bytes <= 0) return 0;
  if (a.cls_on || a.bn_x != nullptr || a.in_scale != nullptr) return 0;
  if (g.kt != 1 || g.kh != 7 || g.kw != 4 || g.st != 1 || g.sh != 2 || g.sw != 1 || g.pt || g.ph || g.pw || g.CP != 8 || a.lds_ != 8) return 0;
  if (g.rW < 1 || g.rW > 64 || g.sW < g.rW + 3 || g.sH < 2 * g.rH + 5) return 0;
  for (int G = std::min(256 / g.rW, g.rH); G >= 1; --G) {
    if (g.rH % G) continue;
    if ((2 * G + 5) * g.sW > 768) continue;                      // three staging units per thread
    if (2 * 64 * 96 + (size_t)(2 * G + 5) * g.sW * 48 > 54000) continue;       // three workgroups per CU
    if (!any_size && (G * g.rW < 128 || (int64_t)(a.M / (G * g.rW)) * ((a.NP + 63) / 64) < 512)) return 0;
    return G;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------ host side
// 0: not applicable; 1: spatial; 2: temporal
static int tap_kind(const ConvArgs& a, int mode) {
  static const int on = getenv("DUALVAR_CONV_TAP") ? atoi(getenv("DUALVAR_CONV_TAP")) : 1;
  // smallest grid (counted in 256-row x 64-column tiles) the kernel takes.  With the 128-row tiles (tap_bm) the 64-channel
  // branches of the 12 544-row levels (49 tiles) belong here too: their data gradients leave conv_gemm_ks (nine launches, 498 -> 177
  // us) for +287 us of this kernel, forwards likewise, five more BatchNorms are applied on load -- step 16.35 - 16.49 -> 16.10 -
  // 16.22 ms (thresholds 128 / 49 / 25 / 10: 16.35, 16.16, 16.18, 16.36 ms).  Below that (the 1 152-row levels) conv_gemm_ks's K split
  // over the waves fills the chip better.  DUALVAR_CONV_TAP_GRID overrides (tests: 1).
  static const int min_grid = getenv("DUALVAR_CONV_TAP_GRID") ? atoi(getenv("DUALVAR_CONV_TAP_GRID")) : 49;
  if (!on) return 0;
  const ConvGeom& g = a.g;
  if (!(a.flags & DV_W3) || (a.flags & (DV_BIAS | DV_RELU | DV_SIGMOID)) || a.out_bytes <= 0) return 0;
  // the fused BatchNorm-backward reduce: only in its ordered form (workspace given), on a data gradient that is not accumulated
  if (a.bn_x != nullptr && (a.bn_ws == nullptr || a.bn_bytes <= 0 || mode != MODE_DGRAD || (a.flags & DV_ACCUM))) return 0;
  if (a.cls_on == 1 && a.in_scale != nullptr) return 0;
  if (a.cls_on == 1) {
    // one parity class of a t-strided kt x 1 x 1 data gradient (S3D-G's 7x1x1 / stride 2 stem conv, backbone/s3dg.py:151): 3 or 4
    // taps over the four frames of dY, rows = the class's four frames of dX
    if (mode != MODE_DGRAD || g.kh != 1 || g.kw != 1 || a.csh != 1 || a.csw != 1 || a.coh || a.cow || a.crh || a.crw) return 0;
    if (g.st != 1 || g.sh != 1 || g.sw != 1 || g.CP % 16 != 0 || g.rH != g.sH || g.rW != g.sW) return 0;
    if (g.sT != 4 || g.rT != 4 || (g.kt != 3 && g.kt != 4) || g.pt < 0 || g.pt >= g.kt) return 0;
    if ((int64_t)((a.M + 255) / 256) * ((a.NP + 63) / 64) < min_grid) return 0;
    return 2;
  }
  if (a.cls_on) return 0;
  if (mode != MODE_FWD && (a.flags & DV_STATS)) return 0;
  // BatchNorm on load: forward of the stride-1 temporal form only
  if (a.in_scale != nullptr && !(mode == MODE_FWD && g.kt == 3 && g.kh == 1 && g.kw == 1 && g.CP <= 512)) return 0;
  if (g.st != 1 || g.sh != 1 || g.sw != 1 || g.CP % 16 != 0) return 0;
  if (g.rT != g.sT || g.rH != g.sH || g.rW != g.sW) return 0;
  const int64_t grid = (int64_t)((a.M + 255) / 256) * ((a.NP + 63) / 64);
  if (grid < min_grid) return 0;
  if (g.kt == 1 && g.kh == 3 && g.kw == 3 && g.ph == 1 && g.pw == 1) {
    if (g.sW + 1 > 57 || g.sH < 2 || g.sW < 2) return 0;         // planes of 256 + 2 (W + 1) positions: three workgroups per CU
    return 1;
  }
  if (g.kt == 3 && g.kh == 1 && g.kw == 1 && g.pt == 1 && (g.sT == 2 || g.sT == 4 || g.sT == 8)) return 2;
  return 0;
}

template <int KIND, int NTAPS, int NU, bool BNL = false, int BM = 256>
static void launch_tap(const TapArgs& t, int grid, size_t lds, hipStream_t s) {
  hipLaunchKernelGGL((conv_tap_kernel<KIND, NTAPS, 64, 3, NU, BNL, BM>), dim3(grid), dim3(256), lds, s, t);
}

// tile height: 128 rows where the 256-row form would launch fewer than ~1.5 workgroups per CU (the 12 544-row levels: 147 - 245
// workgroups, one wave per SIMD, latency bound): twice the workgroups, two or three resident per CU.  Not for the parity
// classes of the strided stem data gradient (large) nor for eight-frame temporal tiles (128 / 8 pixels < one row block).
static int tap_bm(const ConvArgs& a, int kind) {
  static const int on = getenv("DUALVAR_CONV_TAP_BM128") ? atoi(getenv("DUALVAR_CONV_TAP_BM128")) : 1;
  // below ONE full round of three workgroups per CU (sweep on the headline step, kernel time summed over the step: 384 -> 17.60 -
  // 17.65 ms, 800 -> 17.38 - 17.41, 1 600 -> 17.33 - 17.43; step time equal within noise)
  static const int max_grid = getenv("DUALVAR_CONV_TAP_BM128_GRID") ? atoi(getenv("DUALVAR_CONV_TAP_BM128_GRID")) : 768;
  if (!on || a.cls_on == 1) return 256;
  if (kind == 2 && a.g.sT != 2 && a.g.sT != 4) return 256;
  const int64_t grid = (int64_t)((a.M + 255) / 256) * ((a.NP + 63) / 64);
  return grid < max_grid ? 128 : 256;
}

}  // namespace

// floats of dv_conv3d_dgrad_bn_ws's workspace for a launch of `rows` rows and `np` (padded) columns
int64_t dvt_bn_ws_floats(int64_t rows, int np) {
  const int n_mt = (int)((rows + 127) / 128), ntn = (np + 63) / 64;        // (sized for the 128-row tiles: covers both tile heights)
  if ((int64_t)ntn * bn_ws_tick_stride(n_mt) > kBnTickWords) return 0;     // (more tickets than the ticket region holds: not fused)
  return kBnTickWords + (int64_t)ntn * bn_ws_floats_per_coltile(n_mt);
}

// rows per tile (= rows per BatchNorm partial) when dv_conv3d_fwd runs this problem on the pixel-pair stem form, else 0
int dvt_conv_pp_rows(const void* conv_args, int mode) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(conv_args);
  return pp_lines(a, mode) * a.g.rW;
}

int dvt_conv_pp_launch(const void* conv_args, int mode, void* stream) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(conv_args);
  const int G = pp_lines(a, mode);
  if (!G) return 0;
  const ConvGeom& g = a.g;
  PpArgs t;
  t.src = a.src; t.w = a.w; t.out = a.out; t.stats = a.stats;
  t.M = a.M; t.N = a.N; t.NP = a.NP; t.ldo = a.ldo; t.ldw = a.ldw; t.ntn = (a.NP + 63) / 64;
  t.flags = a.flags & DV_STATS;
  t.src_bytes = a.src_bytes; t.w_bytes = a.w_bytes; t.out_bytes = a.out_bytes;
  t.Wo = g.rW; t.Wp = g.sW; t.Hp = g.sH; t.G = G; t.rows = G * g.rW; t.npos = (2 * G + 5) * g.sW;
  t.Hg = g.rH / G;
  t.fWo = make_fastdiv((uint32_t)g.rW); t.fWp = make_fastdiv((uint32_t)g.sW); t.fHg = make_fastdiv((uint32_t)t.Hg);
  const int grid = t.ntn * (a.M / t.rows);
  const size_t lds = 2 * 64 * 96 + (size_t)t.npos * 48;
  hipLaunchKernelGGL((conv_pp_fwd_kernel<2>), dim3(grid), dim3(256), lds, (hipStream_t)stream, t);
  return 1;
}

// entry points for conv.hip (the argument block is conv_common.hpp's ConvArgs, passed by address)
int dvt_conv_tap_kind(const void* conv_args, int mode) { return tap_kind(*static_cast<const ConvArgs*>(conv_args), mode); }
// rows per tile (= rows per BatchNorm partial) of that launch, 0 when the problem does not run on the kernel
int dvt_conv_tap_rows(const void* conv_args, int mode) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(conv_args);
  const int kind = tap_kind(a, mode);
  return kind ? tap_bm(a, kind) : 0;
}

// launches the LDS-staged kernel for this problem if it is one of its forms; returns 1 when it did
int dvt_conv_tap_launch(const void* conv_args, int mode, void* stream) {
  const ConvArgs& a = *static_cast<const ConvArgs*>(conv_args);
  const int kind = tap_kind(a, mode);
  if (!kind) return 0;
  const ConvGeom& g = a.g;
  TapArgs t;
  t.src = a.src; t.w = a.w; t.out = a.out; t.stats = a.stats;
  t.M = a.M; t.N = a.N; t.NP = a.NP; t.lds_ = a.lds_; t.ldo = a.ldo; t.ldw = a.ldw;
  t.ntn = (a.NP + 63) / 64;
  t.flags = a.flags & (DV_STATS | DV_ACCUM);
  t.src_bytes = a.src_bytes; t.w_bytes = a.w_bytes; t.out_bytes = a.out_bytes;
  t.CP = g.CP;
  t.H = g.sH; t.W = g.sW; t.T = g.sT; t.S = g.sH * g.sW;
  t.fW = make_fastdiv((uint32_t)g.sW); t.fH = make_fastdiv((uint32_t)g.sH); t.fS = make_fastdiv((uint32_t)t.S);
  const int bm = tap_bm(a, kind);
  t.P = bm / g.sT; t.lgP = 0;
  while ((1 << t.lgP) < t.P) ++t.lgP;
  t.NQ = a.M / g.sT;
  t.pt = g.pt; t.wt0 = 0; t.wts = 1; t.oT = g.sT; t.ofs = 1; t.ofo = 0;
  if (a.cls_on == 1) { t.wt0 = a.crt; t.wts = a.cst; t.oT = a.oT; t.ofs = a.cst; t.ofo = a.cot; }
  t.sgn = mode == MODE_FWD ? 1 : -1;
  t.bn_x = a.bn_x; t.bn_mean = a.bn_mean; t.bn_invstd = a.bn_invstd; t.bn_scale = a.bn_scale; t.bn_shift = a.bn_shift;
  t.bn_sums = a.bn_sums; t.bn_ws = a.bn_ws; t.bn_ldx = a.bn_ldx; t.bn_mask = a.bn_mask; t.bn_bytes = a.bn_bytes;
  t.bn_cpb = (a.N + 7) & ~7;
  if (a.bn_x == nullptr) { t.bn_ws = nullptr; t.bn_sums = nullptr; t.bn_bytes = 0; }
  t.in_scale = a.in_scale; t.in_shift = a.in_shift; t.in_C = a.in_C; t.in_relu = a.in_relu;
  const int grid = t.ntn * ((a.M + bm - 1) / bm);
  hipStream_t s = (hipStream_t)stream;
  if (kind == 1) {
    t.halo = g.ph * g.sW + g.pw;
    t.npos = bm + 2 * t.halo;
    t.npp = t.npos + ((4 - t.npos % 8) + 8) % 8;
    const size_t lds = 3 * 64 * 96 + (size_t)t.npp * 96 + 64;
    if (bm == 128) launch_tap<0, 9, 2, false, 128>(t, grid, lds, s);        // (128 + 2 (W + 1) <= 256 positions: two units per thread)
    else if (2 * t.npos <= 512) launch_tap<0, 9, 2>(t, grid, lds, s);
    else launch_tap<0, 9, 3>(t, grid, lds, s);
  } else {
    t.halo = 0;
    t.npos = bm;
    t.npp = t.npos + 4;
    const size_t lds = 3 * 64 * 96 + (size_t)t.npp * 96 + 64;
    if (bm == 128) {
      if (a.in_scale != nullptr) launch_tap<1, 3, 1, true, 128>(t, grid, lds + (size_t)g.CP * 8, s);
      else if (g.kt == 4) launch_tap<1, 4, 1, false, 128>(t, grid, lds, s);
      else launch_tap<1, 3, 1, false, 128>(t, grid, lds, s);
    }
    else if (a.in_scale != nullptr) launch_tap<1, 3, 2, true>(t, grid, lds + (size_t)g.CP * 8, s);
    else if (g.kt == 4) launch_tap<1, 4, 2>(t, grid, lds, s);
    else launch_tap<1, 3, 2>(t, grid, lds, s);
  }
  return 1;
}
