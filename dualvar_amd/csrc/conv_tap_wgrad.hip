// Weight gradient of the stride-1 3x1x1 ("temporal") convs in the fp32 split mode, with LDS-staged operand tiles that are
// split ONCE -- the weight-gradient half of the LDS-staged input-tile design of conv_tap.hip.  Reference: the backward of
// nn.Conv3d(k = (3, 1, 1), padding (1, 0, 0)) in backbone/s3dg.py:41,58-65 (STConv3d.conv2) and backbone/r21d.py:54-70.
//
//   dW[n][dt][c] = sum over clips, pixels p, frames f:  dY[f][p][n] * X[f + dt - 1][p][c]
//
// conv_wgrad_dma_kernel (conv.hip) treats this as a GEMM over rows with J = (tap, channel) columns: every tap's column tile
// fetches and splits its own copy of the x rows (the same rows one frame apart) and of dY.  Here a workgroup owns 64 output
// channels x 64 input channels x ALL THREE taps and a range of pixels; a step stages 64 / T pixels of ALL T frames of both
// operands once (global -> registers -> bf16 triples -> row-major planes in LDS), and each (frame, tap) pair whose source frame
// lies inside the clip is one group of MFMAs on fragments read TRANSPOSED from those planes (ds_read_b64_tr_b16; K = pixels): per
// MFMA a third of the operand loads and splits of the per-tap form, and the pairs that leave the clip (2 of 12 at T = 4) are not
// multiplied.  No halo: a tap is a whole frame away.  48 KB of LDS (2 operands x 3 planes x 64 rows x 128 B): three workgroups
// per CU.  The partial tiles go to the same row-split slabs as conv_wgrad_dma_kernel's and are added by wgrad_reduce_kernel in a
// fixed order (no float atomics).
#include "conv_common.hpp"

namespace {

__device__ __forceinline__ void tw_gload16(f32x4& dst, dma_rsrc_t rsrc, unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void tw_gload16_hi(f32x4& dst, dma_rsrc_t rsrc, unsigned voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}

// ds_read_b64_tr_b16 fragment from a plane of 64-BYTE rows (32 channels): four consecutive rows are one 256-byte LDS line, so the
// reads are conflict-free without a swizzle (wg_frag<128> XORs the 64-byte chunk instead)
__device__ __forceinline__ bf16x8 wg_frag64(const unsigned char* tile, int lane, int ks) {
  const int g16 = lane >> 4, li = lane & 15;
  const int q = li >> 2, p = li & 3;
  const int row = ks * 16 + 8 * (g16 >> 1) + q;
  const unsigned char* ad = tile + row * 64 + (16 * (g16 & 1) + 4 * p) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad + 4 * 64));
  s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}

// YT frames of dY, XT frames of x, KT taps, temporal stride ST, padding PT: tap dt of output frame to reads x frame
// to * ST - PT + dt.  PXS = 64 / YT pixels per step (dY planes: 64 rows x 64 channels).  Two forms:
//   BC = 64 (stride 1, XT = YT = 2 / 4, KT = 3): x planes 64 rows x 64 channels; wave (bi, bj) carries its 32 x 32 block for
//            all KT taps;
//   BC = 32 (the 7x1x1 / stride-2 stem conv, backbone/s3dg.py:151: XT = 8, YT = 4): x planes 128 rows x 32 channels (the same
//            24 KB); wave (bi, parity) carries the 32 x 32 blocks of the taps of its parity -- even taps only ever read odd
//            x frames and vice versa, so a wave touches half of the x planes.
// BNL (dv_conv3d_wgrad_bn_in): x is the INPUT of the BatchNorm (+ReLU) in front of this conv; the x operand is
// y = [relu](x * scale + shift) formed in front of the split with dv_bn_apply's expression (coefficients of the workgroup's BC
// channels in LDS).  Rows past the end carry [relu](shift) instead of zero: their dY rows are zero, the products vanish.
template <int YT, int XT, int KT, int ST, int PT, int BC, bool BNL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BC == 64 ? 3 : 2, BC == 64 ? 3 : 2))) void conv_wgrad_tm_kernel(TmWgradArgs a) {
  constexpr int PXS = 64 / YT, KS = PXS / 16, YROWS = 64, XROWS = XT * PXS;
  constexpr int RBY = 128, RBX = BC * 2;                         // plane row bytes
  constexpr int YPLANE = YROWS * RBY, XPLANE = XROWS * RBX, YOP = 3 * YPLANE, XOP = 3 * XPLANE;
  constexpr int XU = BC / 8;                                     // 8-channel units per x row
  static_assert(YROWS * 8 == 512 && XROWS * XU == 512, "four staging units per thread: two of dY, two of x");
  static_assert(BC == 64 || (BC == 32 && ST == 2), "tap-parity waves belong to the stride-2 form");
  constexpr int NACC = BC == 64 ? KT : (KT + 1) / 2;
  constexpr unsigned kOOB = 0x80000000u;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[YOP + XOP + (BNL ? 2 * BC * 4 : 0)];      // dY planes, then x planes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_c = bid % a.ntc; bid /= a.ntc;
  const int tile_i = bid % a.nti;
  const int split = bid / a.nti;
  const int i0 = tile_i * 64, c0 = tile_c * BC;
  if constexpr (BNL) {
    float* tab = reinterpret_cast<float*>(smem + YOP + XOP);      // [scale | shift][BC]; published by the first step's barrier
    if (tid < BC) {
      const int c = c0 + tid;
      tab[tid] = c < a.in_C ? a.in_scale[c] : 0.f;
      tab[BC + tid] = c < a.in_C ? a.in_shift[c] : 0.f;
    }
    __syncthreads();
  }
  const int bi = wave >> 1, wlo = wave & 1;                      // wlo: column block (BC = 64) / tap parity (BC = 32)
  const int ch_begin = split * a.chunks_per_split, ch_end = min(a.nchunks, ch_begin + a.chunks_per_split);
  const dma_rsrc_t x_rs = dma_make_rsrc(a.x, (unsigned)a.x_bytes), dy_rs = dma_make_rsrc(a.dy, (unsigned)a.dy_bytes);

  // ---- staging roles: 4 units of 8 channels per thread: k = 0, 1 -> dY, k = 2, 3 -> x; unit -> (row, 8-channel group)
  // kept per unit: the part of its byte offset that does not change from step to step (frame, channel group; kOOB beyond the
  // channel pitch) and its place in the planes; the pixel inside the step is re-derived from the thread index
  unsigned ubase[4], uwr[4];
  const unsigned ldxb = (unsigned)a.ldx * 4u, ldyb = (unsigned)a.ldy * 4u;
  auto unit_row = [&](int k) -> int { const int u = (tid + 256 * k) & 511; return k >= 2 ? u / XU : u >> 3; };
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool isx = k >= 2;
    const int u = (tid + 256 * k) & 511;
    const int row = unit_row(k), ch8 = isx ? u % XU : u & 7;
    const int f = row / PXS;
    const int cbase = (isx ? c0 : i0) + ch8 * 8;
    const int climit = isx ? a.CP : a.CoutP;
    ubase[k] = cbase + 8 <= climit ? (unsigned)(f * a.S) * (isx ? ldxb : ldyb) + (unsigned)cbase * 4u : kOOB;
    // dY: plane row, 16-byte slot ch8, 64-byte chunk XOR-swizzled with the row as wg_frag<128> expects; x with 64-byte rows: plain
    if (!isx) uwr[k] = (unsigned)(row * RBY + ((ch8 ^ (wg_swz<RBY>(row) << 2)) << 4));
    else uwr[k] = (unsigned)(YOP + row * RBX + (RBX == 128 ? ((ch8 ^ (wg_swz<128>(row) << 2)) << 4) : (ch8 << 4)));
  }
  f32x4 sreg[4][2];
  auto issue = [&](int chunk) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned q = (unsigned)(chunk * PXS + unit_row(k) % PXS);
      const unsigned n = fd_div(q, a.fS);
      // row of (clip n, frame f, pixel s) in a tensor of FT frames: (n * FT + f) * S + s = q + n * (FT - 1) * S + f * S
      const unsigned grow0 = q + n * (unsigned)(((k >= 2 ? XT : YT) - 1) * a.S);
      const bool ok = (int)q < a.NQ && ubase[k] != kOOB;
      const unsigned off = ok ? grow0 * (k >= 2 ? ldxb : ldyb) + ubase[k] : kOOB;
      tw_gload16(sreg[k][0], k >= 2 ? x_rs : dy_rs, off);
      tw_gload16_hi(sreg[k][1], k >= 2 ? x_rs : dy_rs, off);
    }
  };

  f32x16 acc[NACC];
#pragma unroll
  for (int d = 0; d < NACC; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  auto frag_y = [&](int f, int ks) -> Split3 {
    Split3 s;
    const unsigned char* t = smem + f * PXS * RBY;
    s.hi = wg_frag<RBY>(t, bi * 32, lane, ks);
    s.mid = wg_frag<RBY>(t + YPLANE, bi * 32, lane, ks);
    s.lo = wg_frag<RBY>(t + 2 * YPLANE, bi * 32, lane, ks);
    return s;
  };
  auto frag_x = [&](int f, int ks) -> Split3 {
    Split3 s;
    const unsigned char* t = smem + YOP + f * PXS * RBX;
    if constexpr (RBX == 128) {
      s.hi = wg_frag<128>(t, wlo * 32, lane, ks);
      s.mid = wg_frag<128>(t + XPLANE, wlo * 32, lane, ks);
      s.lo = wg_frag<128>(t + 2 * XPLANE, wlo * 32, lane, ks);
    } else {
      s.hi = wg_frag64(t, lane, ks);
      s.mid = wg_frag64(t + XPLANE, lane, ks);
      s.lo = wg_frag64(t + 2 * XPLANE, lane, ks);
    }
    return s;
  };
  // the (x frame, tap) pairs of one tap parity (PAR < 0: all taps): x frame fx feeds tap dt of dY frame to = (fx + PT - dt) / ST
  auto pairs = [&](auto par_c, int ks) {
    constexpr int PAR = decltype(par_c)::value;
#pragma unroll
    for (int fx = 0; fx < XT; ++fx) {
      bool any = false;
#pragma unroll
      for (int dt = 0; dt < KT; ++dt) {
        const int num = fx + PT - dt;
        if ((PAR >= 0 && (dt & 1) != PAR) || num < 0 || num % ST != 0 || num / ST >= YT) continue;
        any = true;
      }
      if (!any) continue;                                        // (compile time) no tap of this parity reads this x frame
      const Split3 b = frag_x(fx, ks);
#pragma unroll
      for (int dt = 0; dt < KT; ++dt) {
        const int num = fx + PT - dt;
        if ((PAR >= 0 && (dt & 1) != PAR) || num < 0 || num % ST != 0 || num / ST >= YT) continue;
        const Split3 av = frag_y(num / ST, ks);
        mma_split3(av, b, acc[PAR >= 0 ? dt / 2 : dt]);
      }
    }
  };

  if (ch_begin < ch_end) issue(ch_begin);
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    // the slab of this step has landed (the only loads in flight are this wave's own eight)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(sreg[0][0]), "+v"(sreg[0][1]), "+v"(sreg[1][0]), "+v"(sreg[1][1]), "+v"(sreg[2][0]), "+v"(sreg[2][1]),
                   "+v"(sreg[3][0]), "+v"(sreg[3][1])::"memory");
    // the weight-gradient split (hi rounded, mid / lo exact residue halves: conv_common.hpp split3w).  Three accumulator blocks
    // per wave: all four units are split IN FRONT of the barrier (under the tail of the other waves' MFMAs); four blocks (the
    // stem form) leave no room for the 48 registers of that: unit by unit behind it
    constexpr bool PRESPLIT = NACC <= 3;
    Split3 sp[PRESPLIT ? 4 : 1];
    const float bn_lo = (BNL && !a.in_relu) ? -__builtin_inff() : 0.f;
    auto split_unit = [&](int k) -> Split3 {
      float v[8] = {sreg[k][0].x, sreg[k][0].y, sreg[k][0].z, sreg[k][0].w, sreg[k][1].x, sreg[k][1].y, sreg[k][1].z, sreg[k][1].w};
      if constexpr (BNL) {
        if (k >= 2) {
          const int ch8 = ((tid + 256 * k) & 511) % XU;
          const f32x4* cs = reinterpret_cast<const f32x4*>(smem + YOP + XOP + ch8 * 32);
          const f32x4* ch = reinterpret_cast<const f32x4*>(smem + YOP + XOP + BC * 4 + ch8 * 32);
          const f32x4 s0 = cs[0], s1 = cs[1], h0 = ch[0], h1 = ch[1];
          const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], bn_lo);      // dv_bn_apply's expression; bn_lo = 0 | -inf
        }
      }
      return split3w(v);
    };
    if constexpr (PRESPLIT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sp[k] = split_unit(k);
    }
    __syncthreads();                           // every wave has read its last fragments of the previous step's planes
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pl = k >= 2 ? XPLANE : YPLANE;
      if constexpr (!PRESPLIT) sp[0] = split_unit(k);
      const Split3& q3 = sp[PRESPLIT ? k : 0];
      *reinterpret_cast<bf16x8*>(smem + uwr[k]) = q3.hi;
      *reinterpret_cast<bf16x8*>(smem + uwr[k] + pl) = q3.mid;
      *reinterpret_cast<bf16x8*>(smem + uwr[k] + 2 * pl) = q3.lo;
    }
    __syncthreads();
    if (ch + 1 < ch_end) issue(ch + 1);        // in flight underneath this step's MFMAs
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (BC == 64) pairs(std::integral_constant<int, -1>(), ks);
      else if (wlo == 0) pairs(std::integral_constant<int, 0>(), ks);
      else pairs(std::integral_constant<int, 1>(), ks);
    }
  }

  // ---- epilogue: this wave's 32 x 32 blocks -> slab[split] (plain stores) or dW (+=, one split)
  const int nrow0 = i0 + bi * 32, cc = c0 + (BC == 64 ? wlo * 32 : 0) + l31;
  if (cc < a.CP) {
#pragma unroll
    for (int d = 0; d < NACC; ++d) {
      const int dt = BC == 64 ? d : 2 * d + wlo;
      if (dt >= KT) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nrow0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < a.Cout) {
          const size_t e = (size_t)n * a.ldw + (size_t)dt * a.CP + cc;
          if (a.slab) a.slab[(size_t)split * a.slab_stride + e] = acc[d][r];
          else a.dw[e] += acc[d][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Spatial form: dW[n][(dh, dw)][c] = sum over rows m: dY[m][n] * x[m + (dh - 1) W + (dw - 1)][c], zero where the tap leaves the
// image (backbone/s3dg.py:39 STConv3d.conv1, backbone/r21d.py:54: 1x3x3, stride 1, padding 1).  A workgroup owns 64 output
// channels x 64 input channels x the THREE taps of one kernel row dh and a range of 64-row steps: a step stages 64 rows of dY and
// the 66 rows of x its three taps reach -- 64 new ones; the first two are the last two of the previous step, copied inside the
// planes -- and a tap dw is the same planes read one row further (ds_read_b64_tr_b16 takes any row offset; the 64-byte chunk
// swizzle follows the actual row).  Rows whose tap leaves the image read a zero row instead: per step wave 0 decodes the 64
// rows once, three ballots give the invalid-row masks, and a lane redirects the reads of its rows by one v_cndmask each.
// The kernel rows dh are separate workgroups (their x rows are a whole image line apart: staging them together would be the
// 2 x (W + 1)-row halo that makes a one-pass spatial tile pointless), so dY is staged three times instead of nine.
template <int DUMMY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_wgrad_sp_kernel(TmWgradArgs a) {
  constexpr int R = 64, KS = 4, RB = 128, XR = 72;              // rows per step; x planes hold 66 rows (72 allocated)
  constexpr int YPLANE = R * RB, XPLANE = XR * RB, YOP = 3 * YPLANE, XOP = 3 * XPLANE;
  constexpr int ZOFF = YOP + XOP, MOFF = ZOFF + RB;             // a zero row; 6 mask words
  constexpr unsigned kOOB = 0x80000000u;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[MOFF + 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int dh = bid % 3; bid /= 3;
  const int tile_c = bid % a.ntc; bid /= a.ntc;
  const int tile_i = bid % a.nti;
  const int split = bid / a.nti;
  const int i0 = tile_i * 64, c0 = tile_c * 64;
  const int bi = wave >> 1, bj = wave & 1;
  const int ch_begin = split * a.chunks_per_split, ch_end = min(a.nchunks, ch_begin + a.chunks_per_split);
  const dma_rsrc_t x_rs = dma_make_rsrc(a.x, (unsigned)a.x_bytes), dy_rs = dma_make_rsrc(a.dy, (unsigned)a.dy_bytes);
  const unsigned ldxb = (unsigned)a.ldx * 4u, ldyb = (unsigned)a.ldy * 4u;
  const int xshift = (dh - 1) * a.W - 1;                         // x plane row j of a step that starts at row r0 holds x row r0 + xshift + j

  // staging roles: units k = 0, 1 -> dY rows (u >> 3), k = 2, 3 -> the 64 NEW x rows (plane rows 2 .. 65)
  unsigned ucolb[4], uwr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool isx = k >= 2;
    const int u = (tid + 256 * k) & 511, row = u >> 3, ch8 = u & 7;
    const int cbase = (isx ? c0 : i0) + ch8 * 8;
    ucolb[k] = cbase + 8 <= (isx ? a.CP : a.CoutP) ? (unsigned)cbase * 4u : kOOB;
    const int prow = isx ? row + 2 : row;
    uwr[k] = (unsigned)((isx ? YOP : 0) + prow * RB + ((ch8 ^ (wg_swz<RB>(prow) << 2)) << 4));
  }
  f32x4 sreg[4][2];
  auto issue = [&](int chunk) {
    const int r0 = chunk * R;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int u = (tid + 256 * k) & 511;
      const int grow = k >= 2 ? r0 + xshift + 2 + (u >> 3) : r0 + (u >> 3);
      const bool ok = grow >= 0 && grow < a.M && ucolb[k] != kOOB;
      const unsigned off = ok ? (unsigned)grow * (k >= 2 ? ldxb : ldyb) + ucolb[k] : kOOB;
      tw_gload16(sreg[k][0], k >= 2 ? x_rs : dy_rs, off);
      tw_gload16_hi(sreg[k][1], k >= 2 ? x_rs : dy_rs, off);
    }
  };
  auto split_unit = [&](const f32x4 (&r)[2]) -> Split3 {
    const float v[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
    return split3w(v);
  };
  auto write_unit = [&](unsigned off, int plane, const Split3& q3) {
    *reinterpret_cast<bf16x8*>(smem + off) = q3.hi;
    *reinterpret_cast<bf16x8*>(smem + off + plane) = q3.mid;
    *reinterpret_cast<bf16x8*>(smem + off + 2 * plane) = q3.lo;
  };

  f32x16 acc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  // ---- prologue: the zero row, and the two head rows of the first step's x planes (later steps copy them from the tail)
  if (tid < 8) *reinterpret_cast<f32x4*>(smem + ZOFF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (ch_begin < ch_end && tid < 16) {
    const int prow = tid >> 3, ch8 = tid & 7;
    const int grow = ch_begin * R + xshift + prow, cbase = c0 + ch8 * 8;
    const bool ok = grow >= 0 && grow < a.M && cbase + 8 <= a.CP;
    f32x4 t2[2];
    const unsigned off = ok ? (unsigned)grow * ldxb + (unsigned)cbase * 4u : kOOB;
    tw_gload16(t2[0], x_rs, off);
    tw_gload16_hi(t2[1], x_rs, off);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(t2[0]), "+v"(t2[1])::"memory");
    write_unit((unsigned)(YOP + prow * RB + ((ch8 ^ (wg_swz<RB>(prow) << 2)) << 4)), XPLANE, split_unit(t2));
  }
  // this lane's two address rows of a fragment read (ks adds 16): lo = 8 * (g16 >> 1) + q, hi = lo + 4
  const int g16 = lane >> 4, li = lane & 15, fq = li >> 2, fp = li & 3;
  const int frow = 8 * (g16 >> 1) + fq;
  const unsigned fcol = (unsigned)((bj * 32 + 16 * (g16 & 1) + 4 * fp) * 2);
  const unsigned zaddr = (unsigned)ZOFF + fcol % RB;

  if (ch_begin < ch_end) issue(ch_begin);
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(sreg[0][0]), "+v"(sreg[0][1]), "+v"(sreg[1][0]), "+v"(sreg[1][1]), "+v"(sreg[2][0]), "+v"(sreg[2][1]),
                   "+v"(sreg[3][0]), "+v"(sreg[3][1])::"memory");
    Split3 sp[2];                              // (the dY units in front of the barrier, the x units behind it: registers)
#pragma unroll
    for (int k = 0; k < 2; ++k) sp[k] = split_unit(sreg[k]);
    // the tail rows 64, 65 of the x planes become the head rows 0, 1 of this step (same swizzle: (64 >> 1) & 1 == (0 >> 1) & 1)
    bf16x8 carry = {};
    const bool carrier = ch > ch_begin && tid < 48;              // 2 rows x 8 slots x 3 planes
    const unsigned csrc = (unsigned)(YOP + (tid >> 4) * XPLANE + (64 + ((tid >> 3) & 1)) * RB + (tid & 7) * 16);
    if (carrier) carry = *reinterpret_cast<const bf16x8*>(smem + csrc);
    // invalid-row masks of this step (wave 0: lane = row): bit set = tap dw of that row leaves the image (or the row is past M)
    unsigned long long inv0 = 0, inv1 = 0, inv2 = 0;
    if (wave == 0) {
      const int m = ch * R + lane;
      uint32_t q_, ww, hh, q2;
      fd_divmod((uint32_t)m, a.fW, q_, ww);
      fd_divmod(q_, a.fH, q2, hh);
      const bool bad_h = (unsigned)((int)hh + dh - 1) >= (unsigned)a.H || m >= a.M;
      inv0 = __ballot(bad_h || ww == 0);
      inv1 = __ballot(bad_h);
      inv2 = __ballot(bad_h || (int)ww == a.W - 1);
    }
    __syncthreads();                           // every wave has read its last fragments of the previous step's planes
    if (carrier) *reinterpret_cast<bf16x8*>(smem + csrc - 64 * RB) = carry;
#pragma unroll
    for (int k = 0; k < 2; ++k) write_unit(uwr[k], YPLANE, sp[k]);
#pragma unroll
    for (int k = 2; k < 4; ++k) write_unit(uwr[k], XPLANE, split_unit(sreg[k]));
    if (wave == 0 && lane == 0) {
      unsigned* mw = reinterpret_cast<unsigned*>(smem + MOFF);
      mw[0] = (unsigned)inv0; mw[1] = (unsigned)(inv0 >> 32); mw[2] = (unsigned)inv1; mw[3] = (unsigned)(inv1 >> 32);
      mw[4] = (unsigned)inv2; mw[5] = (unsigned)(inv2 >> 32);
    }
    __syncthreads();
    if (ch + 1 < ch_end) issue(ch + 1);        // in flight underneath this step's MFMAs
    const unsigned* mw = reinterpret_cast<const unsigned*>(smem + MOFF);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Split3 av;
      av.hi = wg_frag<RB>(smem, bi * 32, lane, ks);
      av.mid = wg_frag<RB>(smem + YPLANE, bi * 32, lane, ks);
      av.lo = wg_frag<RB>(smem + 2 * YPLANE, bi * 32, lane, ks);
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const unsigned mword = mw[dw * 2 + (ks >> 1)];           // rows 32 * (ks >> 1) .. + 31
        const int r_lo = (ks & 1) * 16 + frow;                   // this lane's address rows inside that word
        const int xr = ks * 16 + frow + dw;                      // plane row of the lo read; the hi read is 4 rows further
        const unsigned a_lo = (unsigned)(YOP + xr * RB) + (fcol ^ ((unsigned)wg_swz<RB>(xr) << 6));
        const unsigned ad_lo = ((mword >> r_lo) & 1u) ? zaddr : a_lo;
        const unsigned ad_hi = ((mword >> (r_lo + 4)) & 1u) ? zaddr : a_lo + 4 * RB;
        Split3 b;
        {
          auto rd = [&](unsigned lo_, unsigned hi_) -> bf16x8 {
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + lo_));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + hi_));
            s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            return __builtin_bit_cast(bf16x8, v);
          };
          // (a redirected lane reads the zero row in every plane: no plane offset for it)
          const unsigned p1l = ad_lo == zaddr ? 0u : (unsigned)XPLANE, p1h = ad_hi == zaddr ? 0u : (unsigned)XPLANE;
          b.hi = rd(ad_lo, ad_hi);
          b.mid = rd(ad_lo + p1l, ad_hi + p1h);
          b.lo = rd(ad_lo + 2 * p1l, ad_hi + 2 * p1h);
        }
        mma_split3(av, b, acc[dw]);
      }
    }
  }

  // ---- epilogue
  const int nrow0 = i0 + bi * 32, cc = c0 + bj * 32 + l31;
  if (cc < a.CP) {
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nrow0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < a.Cout) {
          const size_t e = (size_t)n * a.ldw + (size_t)(dh * 3 + dw) * a.CP + cc;
          if (a.slab) a.slab[(size_t)split * a.slab_stride + e] = acc[dw][r];
          else a.dw[e] += acc[dw][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Pixel-pair stem form: the weight gradient of the network's first conv (backbone/s3dg.py:151 Conv_1a.conv1, 1x7x7 / stride 2 on
// RGB -- here a 1 x 7 x 4 window over 8-channel pixel pairs of the bordered frames, stride (1, 2, 1)), with
// or without the following BatchNorm's backward apply formed on the fly (BNA: dv_conv3d_wgrad_bn -- the conv's input needs no
// gradient, so this kernel is dL/d(conv output)'s only reader).  It is the largest kernel of the step and the last of the
// weight-gradient tail.  One OUTPUT LINE (Wo <= 64 pixels = the K of the MFMAs, padded with zero rows) per step:
//   * dY planes [64 rows][64 channels] as in the other forms (BNA: k1 * g' + k2 * y + k3 per element in front of the split, with
//     dv_bn_bwd_apply's expression; the per-channel coefficients sit in LDS);
//   * the seven input lines the window reaches as planes [dh][pair][8 channels] (16-byte rows): the 32 columns of kernel row dh
//     are its four pair taps x 8 channels, and a lane's transposed read takes its 4-column group from pair (pixel + tap) --
//     the im2col matrix is never formed.  224 columns = 7 blocks; wave (bi, parity) carries the kernel rows of its parity.
template <bool BNA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad_pp_kernel(TmWgradArgs a) {
  constexpr int KH = 7, KS = 4, RBY = 128, XROWS = 68, XRB = 16;
  constexpr int YPLANE = 64 * RBY, YOP = 3 * YPLANE;
  constexpr int XLINE = XROWS * XRB, XPLANE = KH * XLINE, XOP = 3 * XPLANE;
  constexpr int COFF = YOP + XOP;                                // BNA: k1 | k2 | k3 | scale | shift, 64 floats each
  constexpr unsigned kOOB = 0x80000000u;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[COFF + 5 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_i = bid % a.nti;
  const int split = bid / a.nti;
  const int i0 = tile_i * 64;
  const int bi = wave >> 1, par = wave & 1;
  const int ln_begin = split * a.chunks_per_split, ln_end = min(a.nchunks, ln_begin + a.chunks_per_split);
  const dma_rsrc_t x_rs = dma_make_rsrc(a.x, (unsigned)a.x_bytes), dy_rs = dma_make_rsrc(a.dy, (unsigned)a.dy_bytes);
  const dma_rsrc_t bx_rs = dma_make_rsrc(BNA ? a.bn_x : a.dy, (unsigned)a.dy_bytes);
  const unsigned ldyb = (unsigned)a.ldy * 4u;
  const int Wo = a.Wo, Wp = a.Wp;

  // zero everything once: the K padding rows of the dY planes and the pairs past the end of a line are never written again
  for (int o = tid * 16; o < COFF; o += 256 * 16) *reinterpret_cast<f32x4*>(smem + o) = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (BNA) {
    if (tid < 64) {
      const int c = i0 + tid, cpb = (a.Cout + 7) & ~7;
      float k1 = 0.f, k2 = 0.f, k3 = 0.f, sc = 0.f, sh = 0.f;
      if (c < a.Cout) {
        float sg = 0.f, sgx = 0.f;
        for (int r = 0; r < a.bn_rep; ++r) { sg += a.bn_sums[(size_t)r * 2 * cpb + c]; sgx += a.bn_sums[(size_t)r * 2 * cpb + cpb + c]; }
        k1 = a.bn_gamma[c] * a.bn_invstd[c];
        k2 = -k1 * a.bn_invstd[c] * sgx * a.bn_inv_count;
        k3 = -k1 * sg * a.bn_inv_count - k2 * a.bn_mean[c];
        if (a.bn_mask) { sc = a.bn_scale[c]; sh = a.bn_shift[c]; }
        if (split == 0 && a.bn_dgamma) {                         // one workgroup per channel tile: dgamma, dbeta
          a.bn_dbeta[c] += a.bn_dscale * sg;
          a.bn_dgamma[c] += a.bn_dscale * sgx;
        }
      }
      float* cf = reinterpret_cast<float*>(smem + COFF);
      cf[tid] = k1; cf[64 + tid] = k2; cf[128 + tid] = k3; cf[192 + tid] = sc; cf[256 + tid] = sh;
    }
  }

  // staging roles: dY units u = tid, tid + 256 (row u >> 3 < Wo, channel group u & 7 = tid & 7); x units u = tid, tid + 256 < 7 * Wp
  const int ch8 = tid & 7;
  const unsigned ycol = i0 + ch8 * 8 + 8 <= a.CoutP ? (unsigned)(i0 + ch8 * 8) * 4u : kOOB;
  unsigned ywr[2], xwr[2], xsrc[2];
  bool yok[2], xok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int u = tid + 256 * k, row = u >> 3;
    yok[k] = row < Wo && ycol != kOOB;
    ywr[k] = (unsigned)(row * RBY + ((ch8 ^ (wg_swz<RBY>(row) << 2)) << 4));
    const int dh = u / Wp, wp = u - dh * Wp;
    xok[k] = u < KH * Wp;
    xwr[k] = (unsigned)(YOP + dh * XLINE + wp * XRB);
    xsrc[k] = (unsigned)((dh * Wp + wp) * 32);                   // bytes from the first pair of input line (2 ho + 0)
  }
  f32x4 greg[2][2], breg[2][2], xreg[2][2];
  auto issue = [&](int line) {
    const unsigned m0 = (unsigned)line * (unsigned)Wo;           // first output row of the line
    const unsigned img = fd_div((uint32_t)line, a.fH);           // (n * T + t); ho = line - img * Ho
    const unsigned ho = (unsigned)line - img * (unsigned)a.Ho;
    const unsigned xline0 = ((img * (unsigned)a.Hp + 2u * ho) * (unsigned)Wp) * 32u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned yo = yok[k] ? (m0 + (unsigned)((tid + 256 * k) >> 3)) * ldyb + ycol : kOOB;
      tw_gload16(greg[k][0], dy_rs, yo);
      tw_gload16_hi(greg[k][1], dy_rs, yo);
      if constexpr (BNA) {
        tw_gload16(breg[k][0], bx_rs, yo);
        tw_gload16_hi(breg[k][1], bx_rs, yo);
      }
      const unsigned xo = xok[k] ? xline0 + xsrc[k] : kOOB;
      tw_gload16(xreg[k][0], x_rs, xo);
      tw_gload16_hi(xreg[k][1], x_rs, xo);
    }
  };
  auto write3 = [&](unsigned off, int plane, const Split3& q3) {
    *reinterpret_cast<bf16x8*>(smem + off) = q3.hi;
    *reinterpret_cast<bf16x8*>(smem + off + plane) = q3.mid;
    *reinterpret_cast<bf16x8*>(smem + off + 2 * plane) = q3.lo;
  };

  f32x16 acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  // this lane's part of an x fragment address: rows (pixel + pair tap), 8-byte column half
  const int g16 = lane >> 4, li = lane & 15, fq = li >> 2, fp = li & 3;
  const int cg = (g16 & 1) * 4 + fp;
  const unsigned xlane = (unsigned)(YOP + (8 * (g16 >> 1) + fq + (cg >> 1)) * XRB + (cg & 1) * 8);

  __syncthreads();                             // zeros and coefficients are in place
  if (ln_begin < ln_end) issue(ln_begin);
  for (int ln = ln_begin; ln < ln_end; ++ln) {
    if constexpr (BNA)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(greg[0][0]), "+v"(greg[0][1]), "+v"(greg[1][0]), "+v"(greg[1][1]), "+v"(breg[0][0]), "+v"(breg[0][1]),
                     "+v"(breg[1][0]), "+v"(breg[1][1]), "+v"(xreg[0][0]), "+v"(xreg[0][1]), "+v"(xreg[1][0]), "+v"(xreg[1][1])::"memory");
    else
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(greg[0][0]), "+v"(greg[0][1]), "+v"(greg[1][0]), "+v"(greg[1][1]), "+v"(xreg[0][0]), "+v"(xreg[0][1]),
                     "+v"(xreg[1][0]), "+v"(xreg[1][1])::"memory");
    Split3 sy[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float v[8] = {greg[k][0].x, greg[k][0].y, greg[k][0].z, greg[k][0].w, greg[k][1].x, greg[k][1].y, greg[k][1].z, greg[k][1].w};
      if constexpr (BNA) {
        const float xv[8] = {breg[k][0].x, breg[k][0].y, breg[k][0].z, breg[k][0].w, breg[k][1].x, breg[k][1].y, breg[k][1].z, breg[k][1].w};
        const float* cf = reinterpret_cast<const float*>(smem + COFF) + ch8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float act = xv[e] * cf[192 + e] + cf[256 + e];   // the forward's expression (dv_bn_apply), same rounding
          const float gg = (a.bn_mask && !(act > 0.f)) ? 0.f : v[e];
          v[e] = cf[e] * gg + cf[64 + e] * xv[e] + cf[128 + e];  // dv_bn_bwd_apply's expression
        }
      }
      sy[k] = split3w(v);
    }
    __syncthreads();                           // every wave has read its last fragments of the previous line's planes
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (yok[k]) write3(ywr[k], YPLANE, sy[k]);
      if (xok[k]) {
        const float v[8] = {xreg[k][0].x, xreg[k][0].y, xreg[k][0].z, xreg[k][0].w, xreg[k][1].x, xreg[k][1].y, xreg[k][1].z, xreg[k][1].w};
        write3(xwr[k], XPLANE, split3w(v));
      }
    }
    __syncthreads();
    if (ln + 1 < ln_end) issue(ln + 1);        // in flight underneath this line's MFMAs
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      Split3 av;
      av.hi = wg_frag<RBY>(smem, bi * 32, lane, ks);
      av.mid = wg_frag<RBY>(smem + YPLANE, bi * 32, lane, ks);
      av.lo = wg_frag<RBY>(smem + 2 * YPLANE, bi * 32, lane, ks);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int dh = 2 * d + par;
        if (dh >= KH) continue;                                  // (wave-uniform: the odd-parity waves carry three kernel rows)
        const unsigned ad = xlane + (unsigned)(dh * XLINE + ks * 16 * XRB);
        auto rd = [&](unsigned o) -> bf16x8 {
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + o));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + o + 4 * XRB));
          s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          return __builtin_bit_cast(bf16x8, v);
        };
        Split3 b;
        b.hi = rd(ad); b.mid = rd(ad + XPLANE); b.lo = rd(ad + 2 * XPLANE);
        mma_split3(av, b, acc[d]);
      }
    }
  }

  // ---- epilogue: block (bi, dh) is columns dh * 32 .. + 31 of dW (its four pair taps x 8 channels)
  const int nrow0 = i0 + bi * 32;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int dh = 2 * d + par;
    if (dh >= KH) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nrow0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (n < a.Cout) {
        const size_t e = (size_t)n * a.ldw + (size_t)dh * 32 + l31;
        if (a.slab) a.slab[(size_t)split * a.slab_stride + e] = acc[d][r];
        else a.dw[e] += acc[d][r];
      }
    }
  }
}

}  // namespace

// entry point for conv.hip (the argument block is conv_common.hpp's TmWgradArgs, passed by address)
void dvw_wgrad_tm_launch(const void* args, int grid, void* stream) {
  const TmWgradArgs& a = *static_cast<const TmWgradArgs*>(args);
  hipStream_t s = (hipStream_t)stream;
  if (a.kind == 4) {
    if (a.bn_x) hipLaunchKernelGGL((conv_wgrad_pp_kernel<true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_pp_kernel<false>), dim3(grid), dim3(256), 0, s, a);
  } else if (a.kind == 3) hipLaunchKernelGGL((conv_wgrad_sp_kernel<0>), dim3(grid), dim3(256), 0, s, a);
  else if (a.in_scale != nullptr) {
    if (a.kind == 2) hipLaunchKernelGGL((conv_wgrad_tm_kernel<4, 8, 7, 2, 3, 32, true>), dim3(grid), dim3(256), 0, s, a);
    else if (a.T == 4) hipLaunchKernelGGL((conv_wgrad_tm_kernel<4, 4, 3, 1, 1, 64, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_tm_kernel<2, 2, 3, 1, 1, 64, true>), dim3(grid), dim3(256), 0, s, a);
  }
  else if (a.kind == 2) hipLaunchKernelGGL((conv_wgrad_tm_kernel<4, 8, 7, 2, 3, 32>), dim3(grid), dim3(256), 0, s, a);
  else if (a.T == 4) hipLaunchKernelGGL((conv_wgrad_tm_kernel<4, 4, 3, 1, 1, 64>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_wgrad_tm_kernel<2, 2, 3, 1, 1, 64>), dim3(grid), dim3(256), 0, s, a);
}
