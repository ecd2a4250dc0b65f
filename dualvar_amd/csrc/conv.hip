// 3-D convolution (forward, data gradient, weight gradient) as implicit GEMM on the gfx950
// matrix cores.  NDHWC activations, K-contiguous packed weights; see include/dualvar_hip.h.
//
// One gather routine serves all three kernels: a tile of the (virtual) im2col matrix
//     A[row m][k = tap*CP + c]
// is built from 16-byte (8-byte for the 3->4-channel stem) vectors, each lying inside one
// tap.  fwd/dgrad read it K-contiguous (ds_read_b128); wgrad reads the same kind of tile
// transposed (ds_read_b64_tr_b16 for bf16, plain b32 for f32).
//
//   fwd  : Y[m][n]  = sum_k A_x [m][k] * Wf[n][k]            m = output position
//   dgrad: dX[m][c] = sum_k A_dy[m][k] * Wd[c][k]            m = input position, k=(tap,n)
//   wgrad: dW[n][j] = sum_m dY[m][n]  * A_x[m][j]            j = (tap,c)
//
// MFMA: v_mfma_f32_32x32x16_bf16 (bf16 storage) / v_mfma_f32_32x32x2_f32 (f32 parity mode);
// both share the 32x32 C/D layout  col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include "conv_common.hpp"

// the LDS-staged input-tile kernels for stride-1 "same" 1x3x3 / 3x1x1 convs (conv_tap.hip); the argument is a ConvArgs*
int dvt_conv_tap_kind(const void* conv_args, int mode);
int dvt_conv_tap_rows(const void* conv_args, int mode);
int dvt_conv_tap_launch(const void* conv_args, int mode, void* stream);
int64_t dvt_bn_ws_floats(int64_t rows, int np);
int dvt_conv_pp_rows(const void* conv_args, int mode);
int dvt_conv_pp_launch(const void* conv_args, int mode, void* stream);
// dv_conv3d_dgrad_bn's reduce over what a data gradient has written (conv_experiments.hip); the argument is a ConvArgs*
void dvx_dgrad_bn_tail_launch(const void* conv_args, int is_f32, void* stream);

namespace {

// ------------------------------------------------------------------------------------------
// fwd / dgrad kernel.  256 threads = 4 waves arranged WAVES_M x WAVES_N over a BM x BN tile.
// LDS rows hold 64 bytes of K, padded to 80 so that ds_read_b128 fragments are conflict free.
// GM (gather mode): 0 = generic (a K-step may straddle taps; per-lane tap decode through the LDS tap table);
// 1 = uniform tap (channel pitch a multiple of the K-step, <= 32 taps, DMA path): the tap, its source offset and the
// K offset are wave-uniform and live in SGPRs, the separable validity masks are folded into ONE inverted bit mask per
// row in the prologue, and a gather address costs three VALU instructions (add, bfe, lshl_or) instead of ~12 -- the K
// loop was issue bound on exactly that address arithmetic (VALU ~ half of all issued cycles in the PMC profile).
// NS: LDS stages of the DMA pipeline (tiles k+1 .. k+NS-1 are in flight while tile k is multiplied; counted vmcnt).
// Waves per SIMD the register allocator is asked to make room for (the LDS footprint allows that many workgroups; the
// K loop is issue bound and resident waves are what hides it): 6 for the 128x64 / 64x128 bf16 tiles (<= 80 VGPRs instead
// of 84-88), 3 for 128x128 (<= 168 instead of 172), no request otherwise.
constexpr int conv_waves_per_simd(int es, int gvb, int bm, int bn) {
  return (es == 4 && gvb == 16 && bm == 256) ? 3 : (es != 2 || gvb != 16) ? 1 : (bm * bn == 128 * 64) ? 6 : (bm == 128 && bn == 128) ? 3 : 1;
}

// WF (fp32 split mode): the weights arrive PRE-SPLIT in fragment order (dv_pack_w3: [K tile][k half h][row][hi|mid|lo][8]
// bf16, 48 bytes per (row, h)), so a B tile is two contiguous chunks that the DMA copies as they are and a lane's three
// operand fragments are three conflict-free ds_read_b128 -- no vector instruction is spent on the weight operand; only
// the activation fragments are split in the kernel (the kernel was bound by exactly those instructions).
// BNL (dv_conv3d_fwd_bn_in): the gathered tensor is the INPUT of the BatchNorm (+ReLU) in front of this conv; the A fragments are
// y = [relu](x * scale + shift) formed between the LDS read and the split (coefficients in a small LDS table, one inverted tap
// mask per fragment row: a tap that leaves the tensor is a ZERO of y, not [relu](shift)).
template <typename T, int MODE, int GVB, int BM, int BN, int WAVES_M, int WAVES_N, int GM, int NS = 2, bool SPLIT = false,
          bool WF = false, bool BNL = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
__attribute__((amdgpu_waves_per_eu(conv_waves_per_simd(sizeof(T), GVB, BM, BN)))) void conv_gemm_kernel(ConvArgs a) {
  static_assert(!SPLIT || sizeof(T) == 4, "the bf16 split is the fp32 mode's product");
  static_assert(!WF || (SPLIT && GVB == 16), "pre-split weights belong to the fp32 split kernels");
  static_assert(!BNL || (WF && GM == 1 && MODE == MODE_FWD), "BatchNorm on load: forward, uniform-tap gathers, pre-split weights");
  // DMA: 16-byte gathers go global -> LDS directly (buffer_load ... lds), no VGPR staging and no ds_write.  One wave
  // instruction fills 16 rows x 64 B = 1 KiB of a row-linear, UNPADDED tile; bank conflicts of the ds_read_b128 fragment
  // reads are avoided by XOR-swizzling the 16-byte slot with (row>>2)&3, applied on the source side (which k-slot a
  // lane fetches) and on the read side.  The 8-byte-gather instantiation (RGB stem) keeps register staging + padding.
  constexpr bool DMA = (GVB == 16);
  static_assert(GM == 0 || DMA, "uniform-tap gathers are DMA only");
  constexpr int ROWB = 64, PITCH = DMA ? 64 : 80;
  constexpr int BKE = ROWB / (int)sizeof(T);
  constexpr int GV = GVB / (int)sizeof(T);
  constexpr int VPR = ROWB / GVB;           // vectors per row (4 or 8)
  constexpr int NT = WAVES_M * WAVES_N * 64; // threads
  constexpr int RPP = NT / VPR;             // rows per pass
  constexpr int A_PASSES = BM / RPP;
  constexpr int B_PASSES = (BN + RPP - 1) / RPP;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(BN <= NT && BM % RPP == 0, "tile / thread layout");
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int A_G = DMA ? (BM / 16) / NW : A_PASSES;          // 16-row groups of the A tile per wave
  constexpr int BROWB = WF ? 96 : PITCH;                       // bytes of one B row in LDS (WF: hi|mid|lo of both k halves)
  constexpr int BPC = (BN * BROWB + 1023) / 1024;              // 1 KiB DMA pieces of the B tile
  constexpr int B_G = DMA ? (BPC + NW - 1) / NW : B_PASSES;
  static_assert(!DMA || (BM / 16) % NW == 0, "A groups per wave");
  static_assert(TM >= 1 && TN >= 1, "tile");
  typedef typename VecB<GVB>::type vec_t;

  static_assert(NS == 2 || GVB == 16, "deeper pipelines are DMA only");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * (BM * PITCH + BPC * 1024)];
  constexpr int BUFB = BM * PITCH + BPC * 1024;   // A tile then B tile, NS times

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int lbid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = lbid % a.ntn, tile_m = lbid / a.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

  // register-staged path: thread -> (vector slot, row); DMA path: lane -> (row within a 16-row group, swizzled slot)
  const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vslot = DMA ? ((lane & 3) ^ ((lane >> 4) & 3)) : tid % VPR;
  const int vrow = DMA ? (lane >> 2) : tid / VPR;
  auto a_row_of = [&](int p) { return DMA ? 16 * (uwave + p * NW) + vrow : vrow + p * RPP; };
  auto b_row_of = [&](int p) { return DMA ? 16 * (uwave + p * NW) + vrow : vrow + p * RPP; };
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  constexpr int ES = (int)sizeof(T);
  constexpr unsigned kOOB = 0x80000000u;      // beyond every buffer (tensors are < 2 GiB): the load returns zeros

  // ---- gather set-up -------------------------------------------------------------------------------------------
  // The K loop used to spend ~50 scalar/vector instructions and 19 branches per MFMA on coordinate arithmetic and
  // zero-fill selects.  Now: (1) loads go through buffer descriptors, so an invalid tap is just an out-of-range
  // offset and the hardware returns zeros (no branch, no select); (2) tap validity is separable, valid(dt,dh,dw) =
  // vt(dt)&vh(dh)&vw(dw), kept as three small bit masks per row; (3) the source offset of a tap is linear in the tap
  // (for dgrad with stride 2 in the halved coordinates), looked up from a per-kernel LDS table.
  __shared__ int2 taptab[256];
  const int ntaps = g.kt * g.kh * g.kw;
  {
    const int st_s = g.st - 1, sh_s = g.sh - 1, sw_s = g.sw - 1;     // strides are 1 or 2 on the dgrad path
    for (int tp = tid; tp < ntaps; tp += NT) {
      const int dw = tp % g.kw, t2 = tp / g.kw, dh = t2 % g.kh, dt = t2 / g.kh;
      const int toff = MODE == MODE_FWD ? (dt * g.sH + dh) * g.sW + dw
                                        : -(((dt >> st_s) * g.sH + (dh >> sh_s)) * g.sW + (dw >> sw_s));
      const int otap = a.cls_on ? ((a.crt + a.cst * dt) * a.oKH + a.crh + a.csh * dh) * a.oKW + a.crw + a.csw * dw : tp;
      taptab[tp] = make_int2(toff, dt | (dh << 8) | (dw << 16) | (otap << 24));
    }
  }
  const __amdgpu_buffer_rsrc_t src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src), 0, a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, a.w_bytes, 0x00020000);
  const dma_rsrc_t src_dma = dma_make_rsrc(a.src, (unsigned)a.src_bytes), w_dma = dma_make_rsrc(a.w, (unsigned)a.w_bytes);
  const unsigned smem_base = lds_addr(smem);
  const unsigned ldb = (unsigned)a.lds_ * ES;

  unsigned rowoff[A_G], mt[A_G], mh[A_G], mw[A_G];
#pragma unroll
  for (int p = 0; p < A_G; ++p) {
    const RowPos r = decode_row<MODE>((uint32_t)(m0 + a_row_of(p)), a.M, g);
    unsigned bt = 0, bh = 0, bw = 0;
    int pos;
    if (MODE == MODE_FWD) {
      for (int d = 0; d < g.kt; ++d) bt |= ((unsigned)(r.t0 + d) < (unsigned)g.sT ? 1u : 0u) << d;
      for (int d = 0; d < g.kh; ++d) bh |= ((unsigned)(r.h0 + d) < (unsigned)g.sH ? 1u : 0u) << d;
      for (int d = 0; d < g.kw; ++d) bw |= ((unsigned)(r.w0 + d) < (unsigned)g.sW ? 1u : 0u) << d;
      pos = r.base + (r.t0 * g.sH + r.h0) * g.sW + r.w0;
    } else {
      const int st_s = g.st - 1, sh_s = g.sh - 1, sw_s = g.sw - 1;
      for (int d = 0; d < g.kt; ++d) { int v = r.t0 - d; bt |= ((v >= 0 && !(v & st_s) && (v >> st_s) < g.sT) ? 1u : 0u) << d; }
      for (int d = 0; d < g.kh; ++d) { int v = r.h0 - d; bh |= ((v >= 0 && !(v & sh_s) && (v >> sh_s) < g.sH) ? 1u : 0u) << d; }
      for (int d = 0; d < g.kw; ++d) { int v = r.w0 - d; bw |= ((v >= 0 && !(v & sw_s) && (v >> sw_s) < g.sW) ? 1u : 0u) << d; }
      pos = r.base + ((r.t0 >> st_s) * g.sH + (r.h0 >> sh_s)) * g.sW + (r.w0 >> sw_s);
    }
    if (!r.valid) bt = 0;
    mt[p] = bt; mh[p] = bh; mw[p] = bw;
    rowoff[p] = (unsigned)pos * ldb;           // modulo 2^32; exact for every valid (row, tap)
  }
  unsigned woff[B_G];
#pragma unroll
  for (int p = 0; p < B_G; ++p) {
    if constexpr (WF) {
      // LDS byte o of the B region <- W3 byte (h * NPad + n0) * 48 + (o mod BN*48), h = o / (BN*48); a.ldw carries NPad
      const unsigned o = (unsigned)(uwave + p * NW) * 1024u + (unsigned)lane * 16u;
      const unsigned hh = o / (BN * 48u), rem = o - hh * (BN * 48u);
      woff[p] = o < BN * 96u ? (hh * (unsigned)a.ldw + (unsigned)n0) * 48u + rem : kOOB;
    } else {
      const int r = b_row_of(p), n = n0 + r;
      woff[p] = (r < BN && n < a.N) ? (unsigned)n * (unsigned)a.ldw * ES : kOOB;
    }
  }
  // uniform-tap mode: wave-uniform K cursor (next tile to load) and per-row inverted tap masks
  int u_c0 = 0, u_tap = 0, u_dt = 0, u_dh = 0, u_dw = 0;
  int u_otap = a.cls_on ? (a.crt * a.oKH + a.crh) * a.oKW + a.crw : 0;     // weight tap of class tap 0
  unsigned u_toffb = 0;                        // source byte offset of the current tap relative to tap 0
  unsigned imask[A_G];
  if constexpr (GM == 1) {
#pragma unroll
    for (int p = 0; p < A_G; ++p) {
      unsigned inv = 0;
      int tp = 0;
      for (int dt = 0; dt < g.kt; ++dt)
        for (int dh = 0; dh < g.kh; ++dh)
          for (int dw = 0; dw < g.kw; ++dw, ++tp)
            inv |= ((((mt[p] >> dt) & (mh[p] >> dh) & (mw[p] >> dw)) & 1u) ^ 1u) << tp;
      imask[p] = inv;
      rowoff[p] += (unsigned)vslot * GVB;
    }
#pragma unroll
    for (int p = 0; p < B_G; ++p)
      if (!WF && woff[p] != kOOB) woff[p] += (unsigned)vslot * GVB;
  }
  // BNL: scale | shift of the gathered channels (pad lanes: 0, 0) and the inverted tap masks of this lane's FRAGMENT rows
  constexpr int kBnlC = 256;
  float* bncoef = nullptr;
  unsigned cmask[TM];
  int c_c0 = 0, c_tap = 0;                     // wave-uniform K cursor of the tile being multiplied
  if constexpr (BNL) {
    __shared__ __attribute__((aligned(16))) float bntab[2 * kBnlC];
    bncoef = bntab;
    for (int c = tid; c < g.CP; c += NT) {
      bntab[c] = c < a.in_C ? a.in_scale[c] : 0.f;
      bntab[kBnlC + c] = c < a.in_C ? a.in_shift[c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const RowPos r = decode_row<MODE>((uint32_t)(m0 + wm0 + i * 32 + l31), a.M, g);
      unsigned inv = 0;
      int tp = 0;
      for (int dt = 0; dt < g.kt; ++dt)
        for (int dh = 0; dh < g.kh; ++dh)
          for (int dw = 0; dw < g.kw; ++dw, ++tp) {
            const bool ok = r.valid && (unsigned)(r.t0 + dt) < (unsigned)g.sT && (unsigned)(r.h0 + dh) < (unsigned)g.sH &&
                            (unsigned)(r.w0 + dw) < (unsigned)g.sW;
            inv |= (ok ? 0u : 1u) << tp;
          }
      cmask[i] = inv;
    }
  }
  __syncthreads();                             // taptab (and the coefficient table)

  const int nk = (g.Ktot + BKE - 1) / BKE;
  vec_t ra[A_G], rb[B_G];

  // gload(tile, buf): register-staged -> ra/rb (buf ignored); DMA -> straight into LDS buffer `buf`
  auto gload = [&](int kt_idx, int buf) {
    if constexpr (GM == 1) {
      const unsigned s_a = u_toffb + (unsigned)u_c0 * ES;
#pragma unroll
      for (int p = 0; p < A_G; ++p) {
        const unsigned off = ((__builtin_amdgcn_ubfe(imask[p], (unsigned)u_tap, 1u)) << 31) | (rowoff[p] + s_a);
        dma_load16(src_dma, smem_base + buf * BUFB + (uwave + p * NW) * 1024, off);
      }
      const unsigned s_b = (unsigned)(u_otap * g.CP + u_c0) * ES;
#pragma unroll
      for (int p = 0; p < B_G; ++p)
        if (uwave + p * NW < BPC)
          dma_load16(w_dma, smem_base + buf * BUFB + BM * PITCH + (uwave + p * NW) * 1024,
                     WF ? (woff[p] == kOOB ? kOOB : woff[p] + ((unsigned)(u_otap * g.CP + u_c0) / (unsigned)BKE) * (unsigned)a.ldw * 96u)
                        : woff[p] + s_b);                       // (WF: the K tile of the weights' own tap; == kt_idx unless taps are remapped)
      u_c0 += BKE;
      if (u_c0 >= g.CP) {
        u_c0 = 0;
        ++u_tap;
        if (++u_dw == g.kw) {
          u_dw = 0;
          if (++u_dh == g.kh) { u_dh = 0; ++u_dt; }
        }
        const int toff = MODE == MODE_FWD ? (u_dt * g.sH + u_dh) * g.sW + u_dw
                                          : -(((u_dt >> (g.st - 1)) * g.sH + (u_dh >> (g.sh - 1))) * g.sW + (u_dw >> (g.sw - 1)));
        u_toffb = (unsigned)toff * ldb;
        u_otap = a.cls_on ? ((a.crt + a.cst * u_dt) * a.oKH + a.crh + a.csh * u_dh) * a.oKW + a.crw + a.csw * u_dw : u_tap;
      }
      return;
    }
    const unsigned k = (unsigned)(kt_idx * BKE + vslot * GV);
    const unsigned tap = fd_div(k, a.fCP);
    const unsigned c = k - tap * (unsigned)g.CP;
    const bool tin = (int)tap < ntaps;
    const int2 ti = taptab[tin ? tap : 0];
    const unsigned dt = ti.y & 255, dh = (ti.y >> 8) & 255, dw = (ti.y >> 16) & 255, otap = (unsigned)ti.y >> 24;
    const unsigned tb = (unsigned)ti.x * ldb + c * ES;
#pragma unroll
    for (int p = 0; p < A_G; ++p) {
      const unsigned ok = (mt[p] >> dt) & (mh[p] >> dh) & (mw[p] >> dw) & (tin ? 1u : 0u);
      const unsigned off = ok ? rowoff[p] + tb : kOOB;
      if constexpr (DMA) {
        dma_load16(src_dma, smem_base + buf * BUFB + (uwave + p * NW) * 1024, off);
      } else if constexpr (GVB == 16) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(src_rsrc, off, 0, 0);
        ra[p] = make_uint4(v.x, v.y, v.z, v.w);
      } else {
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(src_rsrc, off, 0, 0);
        ra[p] = make_uint2(v.x, v.y);
      }
    }
    const unsigned kb = tin ? (otap * (unsigned)g.CP + c) * ES : kOOB;      // weights sit at the original tap index
#pragma unroll
    for (int p = 0; p < B_G; ++p) {
      const unsigned off = WF ? (woff[p] == kOOB ? kOOB : woff[p] + (unsigned)kt_idx * (unsigned)a.ldw * 96u)
                              : ((woff[p] | kb) >= kOOB ? kOOB : woff[p] + kb);
      if constexpr (DMA) {
        if (uwave + p * NW < BPC)
          dma_load16(w_dma, smem_base + buf * BUFB + BM * PITCH + (uwave + p * NW) * 1024, off);
      } else if constexpr (GVB == 16) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, off, 0, 0);
        rb[p] = make_uint4(v.x, v.y, v.z, v.w);
      } else {
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(w_rsrc, off, 0, 0);
        rb[p] = make_uint2(v.x, v.y);
      }
    }
  };
  auto lstore = [&](int buf) {
    if constexpr (!DMA) {
#pragma unroll
      for (int p = 0; p < A_G; ++p)
        *reinterpret_cast<vec_t*>(smem + buf * BUFB + (vrow + p * RPP) * PITCH + vslot * GVB) = ra[p];
#pragma unroll
      for (int p = 0; p < B_G; ++p) {
        int r = vrow + p * RPP;
        if (r < BN) *reinterpret_cast<vec_t*>(smem + buf * BUFB + (BM + r) * PITCH + vslot * GVB) = rb[p];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // 32-row fragment blocks start at multiples of 32, so (row>>2)&3 of a fragment row is (l31>>2)&3 for A and B alike
  const int swz = DMA ? ((l31 >> 2) & 3) : 0;
  auto compute = [&](int buf) {
    if constexpr (SPLIT) {
      // each fragment is split ONCE per K tile and used by TN (TM) blocks
      Split3 af[TM], bf[TN];
      if constexpr (BNL) {
        const f32x4* cs = reinterpret_cast<const f32x4*>(bncoef + c_c0 + 8 * h);
        const f32x4* ch = reinterpret_cast<const f32x4*>(bncoef + kBnlC + c_c0 + 8 * h);
        const f32x4 s0 = cs[0], s1 = cs[1], h0 = ch[0], h1 = ch[1];
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned char* row = smem + buf * BUFB + (wm0 + i * 32 + l31) * PITCH;
          const f32x4 lo4 = *reinterpret_cast<const f32x4*>(row + ((2 * h) ^ swz) * 16);
          const f32x4 hi4 = *reinterpret_cast<const f32x4*>(row + ((2 * h + 1) ^ swz) * 16);
          float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
          // [relu] and "a tap outside the tensor is a zero of y" as ONE clamp per element: live rows [0 | -inf, +inf], dead rows [0, 0]
          const bool dead = (cmask[i] >> c_tap) & 1u;
          const float lo = (dead || a.in_relu) ? 0.f : -__builtin_inff(), hi = dead ? 0.f : __builtin_inff();
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e] * sc[e] + sh[e], lo, hi);      // dv_bn_apply's expression
          af[i] = split3(v);
        }
        c_c0 += BKE;
        if (c_c0 >= g.CP) { c_c0 = 0; ++c_tap; }
      } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = split3_row(smem + buf * BUFB + (wm0 + i * 32 + l31) * PITCH, h, swz);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (WF) {
          const unsigned char* pb = smem + buf * BUFB + BM * PITCH + ((h * BN + wn0 + j * 32 + l31) * 3) * 16;
          bf[j].hi = *reinterpret_cast<const bf16x8*>(pb);
          bf[j].mid = *reinterpret_cast<const bf16x8*>(pb + 16);
          bf[j].lo = *reinterpret_cast<const bf16x8*>(pb + 32);
        } else {
          bf[j] = split3_row(smem + buf * BUFB + (BM + wn0 + j * 32 + l31) * PITCH, h, swz);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_split3(af[i], bf[j], acc[i][j]);
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          Mma<T>::tile(smem + buf * BUFB + (wm0 + i * 32 + l31) * PITCH,
                       smem + buf * BUFB + (BM + wn0 + j * 32 + l31) * PITCH, h, swz, swz, acc[i][j]);
    }
  };
  if constexpr (DMA) {
    // pieces this wave issues per tile (the B groups may not divide evenly over the waves)
    int pieces = A_G;
#pragma unroll
    for (int p = 0; p < B_G; ++p) pieces += (uwave + p * NW < BPC) ? 1 : 0;
    constexpr int D = NS - 1;                  // prefetch distance
    for (int t = 0; t < D && t < nk; ++t) gload(t, t);
    int cur = 0, nxt = D % NS;                 // stage of tile k / of tile k + D
    for (int kt_idx = 0; kt_idx < nk; ++kt_idx) {
      dma_wait_upto(min(D - 1, nk - 1 - kt_idx) * pieces);   // tile k has landed (this wave's pieces)
      __syncthreads();                         // ... everybody's; and stage `nxt` (tile k-1) is no longer read
      if (kt_idx + D < nk) gload(kt_idx + D, nxt);
      compute(cur);
      cur = cur + 1 == NS ? 0 : cur + 1;
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
  } else {
    gload(0, 0);
    lstore(0);
    __syncthreads();
    for (int kt_idx = 0; kt_idx < nk; ++kt_idx) {
      const int cur = kt_idx & 1;
      if (kt_idx + 1 < nk) gload(kt_idx + 1, cur ^ 1);
      compute(cur);
      if (kt_idx + 1 < nk) lstore(cur ^ 1);
      __syncthreads();
    }
  }

  // ---------------- epilogue
  typedef typename OutOf<T>::type OT;           // what is stored (T, or bf16 for the fp8 GEMMs)
  OT* out = reinterpret_cast<OT*>(a.out);
  const int flags = a.flags;
  if constexpr (sizeof(T) == 1) {               // fp8 operands were scaled per tensor: undo it on the fp32 accumulators
    const float sc = a.sc_a[0] * a.sc_b[0];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= sc;
  }
  // row of the output tensor for row `m` of this launch (identity, or class sub-lattice -> full input: see ConvArgs)
  auto out_row = [&](int m) -> size_t {
    if (a.cls_on != 1) return (size_t)m;           // (2: taps remapped only -- trimmed window, rows are the tensor's)
    uint32_t q_, w_, h_, t_, n_;
    fd_divmod((uint32_t)m, g.dW, q_, w_);
    fd_divmod(q_, g.dH, q_, h_);
    fd_divmod(q_, g.dT, n_, t_);
    return ((size_t)(n_ * a.oT + t_ * a.cst + a.cot) * a.oH + h_ * a.csh + a.coh) * a.oW + w_ * a.csw + a.cow;
  };
  // BatchNorm partials of this tile (fwd + DV_STATS): per column the sum and M2 (about the tile mean) of the values AS
  // STORED, over valid rows.  One pass, fused into the conversion loop: each wave accumulates sum(v - c) and
  // sum((v - c)^2) about a provisional centre c = its first row's value (a sample of the column, so no cancellation
  // problem), which needs no second sweep over the accumulators -- keeping them alive for a two-pass M2 used to cost
  // the forward kernel 16..50 more VGPRs (one resident workgroup per CU less) than the otherwise identical dgrad.
  constexpr bool STATS_MODE = (MODE == MODE_FWD);
  const bool do_stats = STATS_MODE && (flags & DV_STATS);
  const int n_mt = (a.M + BM - 1) / BM;        // partials are stored [2][N][n_mt]: a channel's tiles are contiguous
  const bool full_tile = m0 + BM <= a.M;
  float st_c[TN], st_s1[TN], st_s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) st_c[j] = st_s1[j] = st_s2[j] = 0.f;
  auto stat_acc = [&](int i, int j, int r, float vs, int rl /* row inside the wave strip i */) {
    if constexpr (STATS_MODE) {
      if (i == 0 && r == 0) st_c[j] = __shfl(vs, l31);          // row 0 of the wave's rows (held by the h == 0 half)
      float d = vs - st_c[j];
      if (!full_tile && m0 + wm0 + i * 32 + rl >= a.M) d = 0.f;
      st_s1[j] += d;
      st_s2[j] = fmaf(d, d, st_s2[j]);
    }
  };
  {
    // Each wave stages a strip of its tile in LDS ([row][col], element writes at immediate offsets) and writes it out as
    // 16-byte vectors, a row segment per group of lanes -- instead of one element store and a 64-bit address computation per
    // element (the per-element form was ~1/3 of all instructions of the bf16 kernel, and for f32 -- 64 dword stores, 64
    // address chains and the row-mapping divisions per lane -- more than the K loop of the 1x1x1 layers).
    constexpr int EO = (int)sizeof(OT);
    constexpr int SP = WN * EO + 16;                 // staging row pitch (bytes)
    constexpr int CPR = WN * EO / 16;                // 16-byte chunks per strip row
    constexpr int EPC = 16 / EO;                     // elements per chunk
    constexpr int SROWS = (NW * 32 * SP <= (int)sizeof(smem)) ? 32 : 16;   // strip height that fits the tile buffers
    static_assert(NW * SROWS * SP <= (int)sizeof(smem), "staging fits the tile buffers");
    constexpr int RPS = SROWS / 2;                   // accumulator registers per strip (16 cover 32 rows)
    // fp32 outputs whose rows are the tensor's rows (no parity-class row map): straight from the accumulators.  One register
    // of a 32x32 block is 32 consecutive floats of a row per half wave (two 128-byte segments per store instruction: full
    // rate); the row of register r is a wave-uniform offset (soffset: scalar arithmetic only), the lane's part of the address
    // is fixed per column block.  FULL tiles only: the row term sits in soffset, which the buffer range check does NOT cover
    // (only voffset + the immediate offset are checked), so on a partial last tile rows >= M would be written behind the
    // tensor -- such tiles take the staged path below, which tests every row (the sentinel test is
    // tests/test_ops_gpu.py::test_fp32_partial_tiles_do_not_write_behind_the_output).  No LDS staging, no barrier, no
    // per-element address arithmetic (what made the per-element form of round 1 slow).
    bool direct = false;
    if constexpr (EO == 4) direct = full_tile && a.out_bytes > 0 && a.cls_on != 1;
    if (direct) {
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.out_bytes, 0x00020000);
      const unsigned ldo4 = (unsigned)a.ldo * 4u;
      const bool plain32 = !(flags & (DV_BIAS | DV_RELU | DV_SIGMOID));
      const unsigned rbase = (unsigned)(m0 + wm0 + 4 * h) * ldo4;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn0 + j * 32 + l31;
        const float bv = ((flags & DV_BIAS) && col < a.N) ? a.bias[col] : 0.f;
        const unsigned vo = col < a.NP ? rbase + (unsigned)col * 4u : 0x80000000u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          unsigned old[16];
          if (flags & DV_ACCUM) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              old[r] = __builtin_amdgcn_raw_buffer_load_b32(orsrc, (int)vo, (int)((unsigned)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo4), 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2);
            float v = acc[i][j][r];
            if (!plain32) {
              v = act_apply(v + bv, flags);
              if (col >= a.N) v = 0.f;
            }
            acc[i][j][r] = v;                        // as stored: feeds the two-pass statistics
            if (flags & DV_ACCUM) v += __builtin_bit_cast(float, old[r]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc, (int)vo, (int)((unsigned)(i * 32 + rl) * ldo4), 0);
          }
        }
      }
    }
    if (!direct) {
    __syncthreads();                                 // every wave is done reading the last K tile
    unsigned char* stg = smem + wave * (SROWS * SP);
    const bool plain = EO == 2 && !(flags & (DV_BIAS | DV_RELU | DV_SIGMOID));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int sub = 0; sub < 32 / SROWS; ++sub) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = n0 + wn0 + j * 32 + l31;
          const float bv = ((flags & DV_BIAS) && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
          for (int rr = 0; rr < RPS; ++rr) {
            const int r = sub * RPS + rr;
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * h;          // row inside the 32-row block
            float v = acc[i][j][r];
            if (!plain) {                            // (plain: rows >= M and columns >= N are exact zeros already)
              v = act_apply(v + bv, flags);
              if (col >= a.N || m0 + wm0 + i * 32 + rl >= a.M) v = 0.f;
            }
            const OT tv = DT<OT>::from_f(v);
            *reinterpret_cast<OT*>(stg + (rl - sub * SROWS) * SP + (j * 32 + l31) * EO) = tv;
            if constexpr (EO == 2) {
              if (do_stats) stat_acc(i, j, r, DT<OT>::to_f(tv), rl);
            } else {
              acc[i][j][r] = v;                      // as stored (0 outside the valid region): feeds the two-pass statistics
            }
          }
        }
#pragma unroll
        for (int it = 0; it < (SROWS * CPR) / 64; ++it) {
          const int idx = it * 64 + lane;
          const int rl = idx / CPR, ch = idx % CPR;
          const int row = m0 + wm0 + i * 32 + sub * SROWS + rl, col0 = n0 + wn0 + ch * EPC;
          if (row < a.M && col0 < a.NP) {
            OT* p = out + out_row(row) * a.ldo + col0;
            if constexpr (EO == 2) {
              bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + rl * SP + ch * 16);
              if (flags & DV_ACCUM) {
                const bf16x8 o = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)o[e]);
              }
              *reinterpret_cast<bf16x8*>(p) = v;
            } else {
              f32x4 v = *reinterpret_cast<const f32x4*>(stg + rl * SP + ch * 16);
              if (flags & DV_ACCUM) v += *reinterpret_cast<const f32x4*>(p);
              *reinterpret_cast<f32x4*>(p) = v;
            }
          }
        }
      }
    }
    }   // !direct
  }

  if constexpr (STATS_MODE && sizeof(T) == 4) {
    // f32 parity mode: exact two-pass M2 about the tile mean (register pressure is no concern here)
    if (do_stats) {
      float* red = reinterpret_cast<float*>(smem);            // [WAVES_M][BN]
      float* meanb = red + WAVES_M * BN;                      // [BN]
      const int rows_here = min(BM, a.M - m0);
      const int wmi = wave / WAVES_N;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        s += __shfl_xor(s, 32);
        if (h == 0) red[wmi * BN + wn0 + j * 32 + l31] = s;
      }
      __syncthreads();
      if (tid < BN) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) s += red[w * BN + tid];
        meanb[tid] = s / (float)rows_here;
        if (n0 + tid < a.N) a.stats[(size_t)(n0 + tid) * n_mt + tile_m] = s;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float mu = meanb[wn0 + j * 32 + l31];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float dlt = acc[i][j][r] - mu;
            s += (row < a.M) ? dlt * dlt : 0.f;
          }
        s += __shfl_xor(s, 32);
        if (h == 0) red[wmi * BN + wn0 + j * 32 + l31] = s;
      }
      __syncthreads();
      if (tid < BN && n0 + tid < a.N) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) s += red[w * BN + tid];
        a.stats[(size_t)(a.N + n0 + tid) * n_mt + tile_m] = s;
      }
    }
  }
  if constexpr (STATS_MODE && sizeof(OT) == 2) {
    if (do_stats) {
      // merge: halves of a wave share the centre; the WAVES_M row strips of a column are combined pairwise (Chan)
      float* red = reinterpret_cast<float*>(smem);            // [WAVES_M][2][BN]
      static_assert(WAVES_M * 2 * BN * 4 <= (int)sizeof(smem), "stats scratch");
      const int wmi = wave / WAVES_N;
      const float nw = (float)max(0, min(WM, a.M - (m0 + wm0)));
      __syncthreads();                                         // staging reads are done
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float s1 = st_s1[j], s2 = st_s2[j];
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (h == 0) {
          red[(wmi * 2 + 0) * BN + wn0 + j * 32 + l31] = s1 + nw * st_c[j];                    // sum
          red[(wmi * 2 + 1) * BN + wn0 + j * 32 + l31] = nw > 0.f ? s2 - s1 * s1 / nw : 0.f;   // M2 about the strip mean
        }
      }
      __syncthreads();
      if (tid < BN && n0 + tid < a.N) {
        float n = 0.f, S = 0.f, M2 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) {
          const float nq = (float)max(0, min(WM, a.M - (m0 + w * WM)));
          const float sw = red[(w * 2 + 0) * BN + tid], qw = red[(w * 2 + 1) * BN + tid];
          if (nq > 0.f) {
            if (n > 0.f) {
              const float d = sw / nq - S / n;
              M2 += qw + d * d * (n * nq / (n + nq));
            } else {
              M2 = qw;
            }
            S += sw;
            n += nq;
          }
        }
        a.stats[(size_t)(n0 + tid) * n_mt + tile_m] = S;
        a.stats[(size_t)(a.N + n0 + tid) * n_mt + tile_m] = fmaxf(M2, 0.f);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// fwd / dgrad for FEW ROWS and a LONG K (fp32 split mode; the 1 152- and 12 544-row levels of S3D-G, Mixed_4b .. 5c, and the
// projection heads): "K split over the waves".  With 18 .. 196 row tiles the grid of conv_gemm_kernel cannot fill the chip and
// every workgroup walks 30 .. 216 K tiles one after the other, each step costing a global -> LDS round trip and a barrier
// (measured 0.5 us per 64-byte K step: 12.9 TFLOP/s on the 1x3x3 data gradient of Mixed_5c).  Here the FOUR WAVES of a
// workgroup own the same 64 x BN output tile and a quarter of the K range each: every wave runs a private NS-stage DMA
// pipeline (its own LDS stages: no barrier in the K loop, only counted vmcnt waits), so four K steps are in flight per
// workgroup and the sequential depth is a quarter.  The four partial tiles are added through LDS in wave order (fixed order:
// bit-reproducible), then bias / activation / store / BatchNorm partials as in conv_gemm_kernel.
// Restricted to what the engine's fp32 mode issues for those layers: pre-split weights (DV_W3), uniform-tap gathers
// (channel pitch a multiple of 16, <= 32 taps), no parity classes, no fused BatchNorm-backward reduce.
template <int MODE, int BN, int NS>
__global__ __launch_bounds__(256) void conv_gemm_ks_kernel(ConvArgs a) {
  constexpr int BM = 64, NW = 4, TM = 2, TN = BN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 96;          // one K tile: 16 f32 per row of A; hi|mid|lo of both k halves of B
  constexpr int BPC = B_BYTES / 1024;
  static_assert(B_BYTES % 1024 == 0, "B tile in 1 KiB pieces");
  constexpr int STG = A_BYTES + B_BYTES;
  constexpr int PIECES = A_BYTES / 1024 + BPC;
  constexpr int D = NS - 1;
  static_assert(D >= 1 && (D - 1) * PIECES <= 24, "counted wait");
  constexpr unsigned kOOB = 0x80000000u;
  static_assert(NW * NS * STG >= NW * BM * BN * 4 + BM * BN * 4, "the reduction scratch fits the stage buffers");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NW * NS * STG];

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int lbid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_n = lbid % a.ntn, tile_m = lbid / a.ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const dma_rsrc_t src_dma = dma_make_rsrc(a.src, (unsigned)a.src_bytes), w_dma = dma_make_rsrc(a.w, (unsigned)a.w_bytes);
  unsigned char* wsm = smem + wave * (NS * STG);               // this wave's stages
  const unsigned wbase = lds_addr(wsm);
  const unsigned ldb = (unsigned)a.lds_ * 4u;
  const int vslot = (lane & 3) ^ ((lane >> 4) & 3), vrow = lane >> 2;

  unsigned rowoff[4], imask[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const RowPos r = decode_row<MODE>((uint32_t)(m0 + 16 * p + vrow), a.M, g);
    unsigned bt = 0, bh = 0, bw = 0;
    int pos;
    if (MODE == MODE_FWD) {
      for (int d = 0; d < g.kt; ++d) bt |= ((unsigned)(r.t0 + d) < (unsigned)g.sT ? 1u : 0u) << d;
      for (int d = 0; d < g.kh; ++d) bh |= ((unsigned)(r.h0 + d) < (unsigned)g.sH ? 1u : 0u) << d;
      for (int d = 0; d < g.kw; ++d) bw |= ((unsigned)(r.w0 + d) < (unsigned)g.sW ? 1u : 0u) << d;
      pos = r.base + (r.t0 * g.sH + r.h0) * g.sW + r.w0;
    } else {                                                   // stride 1 (checked on the host)
      for (int d = 0; d < g.kt; ++d) { int v = r.t0 - d; bt |= ((v >= 0 && v < g.sT) ? 1u : 0u) << d; }
      for (int d = 0; d < g.kh; ++d) { int v = r.h0 - d; bh |= ((v >= 0 && v < g.sH) ? 1u : 0u) << d; }
      for (int d = 0; d < g.kw; ++d) { int v = r.w0 - d; bw |= ((v >= 0 && v < g.sW) ? 1u : 0u) << d; }
      pos = r.base + (r.t0 * g.sH + r.h0) * g.sW + r.w0;
    }
    if (!r.valid) bt = 0;
    unsigned inv = 0;
    int tp = 0;
    for (int dt = 0; dt < g.kt; ++dt)
      for (int dh = 0; dh < g.kh; ++dh)
        for (int dw = 0; dw < g.kw; ++dw, ++tp) inv |= ((((bt >> dt) & (bh >> dh) & (bw >> dw)) & 1u) ^ 1u) << tp;
    imask[p] = inv;
    rowoff[p] = (unsigned)pos * ldb + (unsigned)vslot * 16u;
  }
  unsigned woff[BPC];
#pragma unroll
  for (int p = 0; p < BPC; ++p) {
    const unsigned o = (unsigned)p * 1024u + (unsigned)lane * 16u;
    const unsigned hh = o / (BN * 48u), rem = o - hh * (BN * 48u);
    woff[p] = (hh * (unsigned)a.ldw + (unsigned)n0) * 48u + rem;          // + K tile * ldw * 96 (a.ldw = padded rows)
  }
  // this wave's K range and its wave-uniform cursor
  const int nk = g.Ktot / 16;
  const int k_lo = (int)(((long long)nk * wave) / NW), k_hi = (int)(((long long)nk * (wave + 1)) / NW);
  int u_tap = (k_lo * 16) / g.CP, u_c0 = k_lo * 16 - u_tap * g.CP;
  int u_dw = u_tap % g.kw, u_dh = (u_tap / g.kw) % g.kh, u_dt = u_tap / (g.kw * g.kh);
  auto tap_off = [&]() -> unsigned {
    const int toff = MODE == MODE_FWD ? (u_dt * g.sH + u_dh) * g.sW + u_dw : -((u_dt * g.sH + u_dh) * g.sW + u_dw);
    return (unsigned)toff * ldb;
  };
  unsigned u_toffb = tap_off();
  auto w_tap = [&]() -> int { return a.cls_on ? ((a.crt + u_dt) * a.oKH + a.crh + u_dh) * a.oKW + a.crw + u_dw : u_tap; };
  int u_otap = w_tap();
  auto gload = [&](int, int st) {
    const unsigned kt = (unsigned)(u_otap * g.CP + u_c0) >> 4;   // K tile of the weights (their own tap when the window is trimmed)
    const unsigned s_a = u_toffb + (unsigned)u_c0 * 4u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const unsigned off = ((__builtin_amdgcn_ubfe(imask[p], (unsigned)u_tap, 1u)) << 31) | (rowoff[p] + s_a);
      dma_load16(src_dma, wbase + st * STG + p * 1024, off);
    }
#pragma unroll
    for (int p = 0; p < BPC; ++p)
      dma_load16(w_dma, wbase + st * STG + A_BYTES + p * 1024, woff[p] + (unsigned)kt * (unsigned)a.ldw * 96u);
    u_c0 += 16;
    if (u_c0 >= g.CP) {
      u_c0 = 0;
      ++u_tap;
      if (++u_dw == g.kw) { u_dw = 0; if (++u_dh == g.kh) { u_dh = 0; ++u_dt; } }
      u_toffb = tap_off();
      u_otap = w_tap();
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int swz = (l31 >> 2) & 3;
  const int nmine = k_hi - k_lo;
  for (int t = 0; t < D && t < nmine; ++t) gload(k_lo + t, t);
  int cur = 0, nxt = D % NS;
  for (int t = 0; t < nmine; ++t) {
    dma_wait_upto(min(D - 1, nmine - 1 - t) * PIECES);         // tile t of this wave has landed
    if (t + D < nmine) gload(k_lo + t + D, nxt);               // its stage was read by step t - 1 of THIS wave (program order)
    const unsigned char* sa = wsm + cur * STG;
    const unsigned char* sb = sa + A_BYTES;
    Split3 af[TM], bf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = split3_row(sa + (i * 32 + l31) * 64, h, swz);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const unsigned char* pb = sb + ((h * BN + j * 32 + l31) * 3) * 16;
      bf[j].hi = *reinterpret_cast<const bf16x8*>(pb);
      bf[j].mid = *reinterpret_cast<const bf16x8*>(pb + 16);
      bf[j].lo = *reinterpret_cast<const bf16x8*>(pb + 32);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma_split3(af[i], bf[j], acc[i][j]);
    cur = cur + 1 == NS ? 0 : cur + 1;
    nxt = nxt + 1 == NS ? 0 : nxt + 1;
  }
  // ---- the four partial tiles -> LDS [wave][row][col], added in wave order
  __syncthreads();                                             // every wave is done with its stages
  float* red = reinterpret_cast<float*>(smem);                 // [NW][BM][BN]
  float* fin = red + NW * BM * BN;                             // [BM][BN]: the tile as stored (statistics)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        red[(wave * BM + row) * BN + j * 32 + l31] = acc[i][j][r];
      }
  __syncthreads();
  constexpr int CPT = BN / 4;                                  // columns per thread: thread = (row, quarter of the columns)
  const int row = tid >> 2, c0 = (tid & 3) * CPT;
  const int flags = a.flags;
  const bool row_ok = m0 + row < a.M;
  float* out = reinterpret_cast<float*>(a.out);
#pragma unroll
  for (int c4 = 0; c4 < CPT / 4; ++c4) {
    const int col = c0 + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(red + (0 * BM + row) * BN + col);
#pragma unroll
    for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(red + (w * BM + row) * BN + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cg = n0 + col + e;
      float x = v[e];
      if (flags & DV_BIAS) x += cg < a.N ? a.bias[cg] : 0.f;
      x = act_apply(x, flags);
      if (cg >= a.N || !row_ok) x = 0.f;
      v[e] = x;
    }
    *reinterpret_cast<f32x4*>(fin + row * BN + col) = v;
    if (row_ok && n0 + col < a.NP) {
      float* p = out + (size_t)(m0 + row) * a.ldo + n0 + col;
      if (flags & DV_ACCUM) v += *reinterpret_cast<const f32x4*>(p);
      *reinterpret_cast<f32x4*>(p) = v;
    }
  }
  if constexpr (MODE == MODE_FWD) {
    if (flags & DV_STATS) {                                    // per column: sum and M2 about the tile mean of the values as stored
      __syncthreads();
      if (tid < BN && n0 + tid < a.N) {
        const int rows_here = min(BM, a.M - m0);
        const int n_mt = (a.M + BM - 1) / BM;
        float s = 0.f;
        for (int r = 0; r < rows_here; ++r) s += fin[r * BN + tid];
        const float mu = s / (float)rows_here;
        float m2 = 0.f;
        for (int r = 0; r < rows_here; ++r) { const float dlt = fin[r * BN + tid] - mu; m2 += dlt * dlt; }
        a.stats[(size_t)(n0 + tid) * n_mt + tile_m] = s;
        a.stats[(size_t)(a.N + n0 + tid) * n_mt + tile_m] = m2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad kernels.  Each workgroup owns a BI (output channels) x BJ (im2col columns) tile of dW and a contiguous slice of
// rows (a "row split"); 32 rows per step.  DETERMINISTIC two-phase accumulation: a workgroup stores its fp32 partial
// tile with plain stores into slab[split] (caller-owned scratch, [splits][Cout][J]) and wgrad_reduce_kernel adds the
// slabs to dW in a fixed order -- no float atomics (their order changed results from run to run, and at ~1.3 TB/s
// chip-wide they were ~40 % of a mid-size launch).  With one split the tile is added to dW directly.
// dW[e] += sum_s slab[s][e], s ascending inside each of the 16 interleaved groups, groups combined in ascending order:
// a fixed summation tree, so the result does not depend on scheduling.  Thread (cx, sg) of a block sums slabs
// sg, sg+16, ... of float4 column blockIdx*16 + cx (16 lanes = 256 contiguous bytes per slab).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, long long stride, int splits,
                                                           float* __restrict__ dw, long long n4) {
  __shared__ f32x4 part[16][17];
  const int cx = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const long long c = (long long)blockIdx.x * 16 + cx;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
  if (c < n4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(slab) + c;
    const long long st4 = stride / 4;
    int s = sg;
    for (; s + 48 < splits; s += 64) {                 // four independent loads in flight
      const f32x4 v0 = p[(long long)s * st4], v1 = p[(long long)(s + 16) * st4], v2 = p[(long long)(s + 32) * st4],
                  v3 = p[(long long)(s + 48) * st4];
      s0 += v0; s0 += v1; s0 += v2; s0 += v3;
    }
    for (; s < splits; s += 16) s0 += p[(long long)s * st4];
  }
  part[sg][cx] = s0;
  __syncthreads();
  if (sg == 0 && c < n4) {
    f32x4* d = reinterpret_cast<f32x4*>(dw) + c;
    f32x4 t = part[0][cx];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += part[g][cx];
    *d = *d + t;
  }
}

// smallest 16-byte-multiple pitch >= bytes with pitch mod 256 in {64, 192}: four consecutive rows of a
// ds_read_b64_tr_b16 block then fall on four different 64-byte bank groups
constexpr int tr_pitch(int bytes) {
  int p = (bytes + 15) / 16 * 16;
  while (p % 256 != 64 && p % 256 != 192) p += 16;
  return p;
}

template <typename T> struct WgMma;
template <> struct WgMma<bf16_t> {
  // P: [32 rows m][pitchP bytes] holding i-columns; Q likewise for j-columns
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* tile, int pitch, int col0, int lane, int ks) {
    const int g16 = lane >> 4, li = lane & 15;
    const int q = li >> 2, p = li & 3;
    const int row = ks * 16 + 8 * (g16 >> 1) + q;
    const unsigned char* ad = tile + row * pitch + (col0 + 16 * (g16 & 1) + 4 * p) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(ad));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(ad + 4 * pitch));
    s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
  }
};

template <typename T, int GVB, int BI, int BJ>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int ES = (int)sizeof(T);
  constexpr int GV = GVB / ES;
  constexpr int ROWS = 32;
  // pitches: bf16 transposed reads want (pitch mod 256) == 64 or 192 bytes; f32 reads are row-linear
  constexpr int PITCH_P = ES == 2 ? tr_pitch(BI * 2) : BI * 4 + 16;
  constexpr int PITCH_Q = ES == 2 ? tr_pitch(BJ * 2) : BJ * 4 + 16;
  constexpr int PV = BI * ES / 16;               // 16-byte vectors per P row
  constexpr int P_PER_THREAD = ROWS * PV / 256;
  constexpr int QV = BJ * ES / GVB;              // gather vectors per Q row
  constexpr int Q_PER_THREAD = (ROWS * QV + 255) / 256;
  constexpr int WI = BI / 2, WJ = BJ / 2;        // waves 2x2
  constexpr int TI = WI / 32, TJ = WJ / 32;
  static_assert(TI >= 1 && TJ >= 1, "tile");
  static_assert(P_PER_THREAD >= 1, "P tile");
  typedef typename VecB<GVB>::type qvec_t;

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * ROWS * (PITCH_P + PITCH_Q)];
  constexpr int BUFB = ROWS * (PITCH_P + PITCH_Q);   // P tile then Q tile, twice
  constexpr int QOFF = ROWS * PITCH_P;

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_j = bid % a.ntj; bid /= a.ntj;
  const int tile_i = bid % a.nti;
  const int split = bid / a.nti;
  const int i0 = tile_i * BI, j0 = tile_j * BJ;
  const int wi0 = (wave >> 1) * WI, wj0 = (wave & 1) * WJ;
  const int m_begin = split * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  if (m_begin >= m_end) return;

  const T* x = reinterpret_cast<const T*>(a.x);
  const T* dy = reinterpret_cast<const T*>(a.dy);

  // fixed column cursors of this thread's Q vectors
  KCursor qc[Q_PER_THREAD];
  int qrow[Q_PER_THREAD], qcol[Q_PER_THREAD];
  bool qok[Q_PER_THREAD];
#pragma unroll
  for (int u = 0; u < Q_PER_THREAD; ++u) {
    int id = tid + u * 256;
    qrow[u] = id / QV;
    qcol[u] = (id % QV) * GV;
    int j = j0 + qcol[u];
    qok[u] = (qrow[u] < ROWS) && (j < a.J);
    qc[u].init(qok[u] ? j : 0, g);
  }

  uint4 rp[P_PER_THREAD];
  qvec_t rq[Q_PER_THREAD];
  auto gload = [&](int mb) {
#pragma unroll
    for (int u = 0; u < P_PER_THREAD; ++u) {
      int id = tid + u * 256;
      int r = id / PV, cv = (id % PV) * (16 / ES);
      int m = mb + r, n = i0 + cv;
      bool ok = (m < m_end) && (n < a.CoutP);
      rp[u] = ok ? *reinterpret_cast<const uint4*>(dy + (size_t)m * a.ldy + n) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < Q_PER_THREAD; ++u) {
      int m = mb + qrow[u];
      int pos = -1;
      if (qok[u] && m < m_end) {
        RowPos rpos = decode_row<MODE_FWD>((uint32_t)m, a.M, g);
        pos = src_pos<MODE_FWD>(rpos, qc[u], g);
      }
      rq[u] = pos >= 0 ? *reinterpret_cast<const qvec_t*>(x + (size_t)pos * a.ldx + qc[u].c) : VecB<GVB>::zero();
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < P_PER_THREAD; ++u) {
      int id = tid + u * 256;
      int r = id / PV, cb = (id % PV) * 16;
      *reinterpret_cast<uint4*>(smem + buf * BUFB + r * PITCH_P + cb) = rp[u];
    }
#pragma unroll
    for (int u = 0; u < Q_PER_THREAD; ++u)
      if (qrow[u] < ROWS) *reinterpret_cast<qvec_t*>(smem + buf * BUFB + QOFF + qrow[u] * PITCH_Q + qcol[u] * ES) = rq[u];
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = (m_end - m_begin + ROWS - 1) / ROWS;
  gload(m_begin);
  lstore(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) gload(m_begin + (s + 1) * ROWS);
    if constexpr (ES == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[TI], bf[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = WgMma<bf16_t>::frag(smem + cur * BUFB, PITCH_P, wi0 + i * 32, lane, ks);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bf[j] = WgMma<bf16_t>::frag(smem + cur * BUFB + QOFF, PITCH_Q, wj0 + j * 32, lane, ks);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < ROWS / 2; ++ks) {
        float af[TI], bf[TJ];
        const int kr = ks * 2 + h;
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const float*>(smem + cur * BUFB + kr * PITCH_P + (wi0 + i * 32 + l31) * 4);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const float*>(smem + cur * BUFB + QOFF + kr * PITCH_Q + (wj0 + j * 32 + l31) * 4);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if (s + 1 < nsteps) lstore(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
      wgrad_store_block(a, split, i0 + wi0 + i * 32, j0 + wj0 + j * 32 + l31, h, acc[i][j]);
}


// ------------------------------------------------------------------------------------------
// wgrad with global -> LDS DMA gathers (16-byte vectors; bf16 and f32).  The 8-byte-gather case keeps the
// register-staged kernel above.  Differences from it:
//  * both tiles are UNPADDED row-linear LDS images filled by buffer_load ... lds (one wave instruction = 1 KiB), so no
//    VGPR staging and no ds_write.  bf16 reads them transposed (ds_read_b64_tr_b16); the bank conflicts of those reads
//    (4 rows x 64 B per half wave) are removed by XOR-swizzling the 64-byte chunk index with the row (row&3 for 256- and
//    512-byte rows, (row>>1)&1 for 128-byte rows), applied on the source side (which column slot a lane fetches) and on
//    the read side.  f32 reads one dword per lane along a row (conflict free as it is);
//  * a row's im2col coordinates are decoded ONCE per workgroup (not once per 16-byte vector): each thread decodes one of
//    the next 256 rows into an LDS table {byte offset of the row's first tap, separable tap-validity masks}; a lane's
//    column slot -- hence its tap and channel -- is fixed for the whole kernel, so the per-step address is
//    table.offset + constant, or an out-of-range offset (hardware zero fill) when the tap falls into the padding;
//  * wave tiles are 64 x 64 (2 x 2 MFMA blocks: 8 MFMAs per 32-row step and wave instead of 4, for the same barrier),
//    workgroup tiles 128 x 128 or 64 x 256 (f32: 64 x 128 with 32 x 64 wave tiles): half the dY / im2col re-reads of
//    the 128 x 64 tiles this replaces.

// NS: LDS stages -- tiles s+1 .. s+NS-1 are in flight while tile s is multiplied (counted vmcnt; every wave issues the
// same NPW + NQW pieces per tile).  WVI x WVJ waves, each a 64 x 64 tile of dW.
// workgroups per CU the LDS footprint admits (<= 3), and the waves per SIMD that makes (the register allocator is
// asked to make room for them)
constexpr int wgrad_wgs_per_cu(int es, int bi, int bj, int nw, int ns) {
  const int lds = ns * 32 * (bi + bj) * es + 24 * nw * 64;
  const int by_lds = 163840 / lds, by_waves = 32 / nw;
  const int k = by_lds < by_waves ? by_lds : by_waves;
  return k >= 3 ? 3 : (k >= 2 ? 2 : 1);
}
constexpr int wgrad_waves_per_simd(int es, int bi, int bj, int nw, int ns) {
  return (wgrad_wgs_per_cu(es, bi, bj, nw, ns) * nw + 3) / 4;
}

// SHARE (fp32 split mode, 1 x 4 waves): the dY fragments of a step -- which every one of the four waves needs -- are split
// ONCE per workgroup (one fragment per thread, the bf16 triples written to LDS in fragment order, a barrier, operands
// fetched as ds_read_b128); each wave splits only its own 32 columns of x.  Three fragments per wave and step instead of the
// six of the 2 x 2 layout (where the two waves of a row / column each split the fragments they share): the kernel is bound
// by the count of exactly those vector instructions.  12 KB of LDS more: still two workgroups per CU.
// BNA (SHARE only; dv_conv3d_wgrad_bn): the dY tile is dv_bn_bwd_apply's output formed on the fly -- a third DMA tile carries
// the BatchNorm's input beside dL/dy, and the thread that splits a dY fragment (its column = its channel: k1, k2, k3, scale and
// shift are per-thread constants) first forms k1*g' + k2*x + k3 with dv_bn_bwd_apply's expression.  8 KB of LDS more per
// stage; the row tables shrink to 128 rows per round so that two workgroups still fit a CU (80 896 bytes each).
template <typename T, int BI, int BJ, int WVI, int WVJ, int NS, bool SPLIT = false, bool SHARE = false, bool BNA = false>
__global__ __launch_bounds__(WVI * WVJ * 64)
__attribute__((amdgpu_waves_per_eu(wgrad_waves_per_simd(sizeof(T), BI, BJ, WVI * WVJ, NS)))) void conv_wgrad_dma_kernel(WgradDmaArgs aa) {
  static_assert(!SPLIT || sizeof(T) == 4, "the bf16 split is the fp32 mode's product");
  static_assert(!SHARE || SPLIT, "the shared split belongs to the fp32 split mode");
  static_assert(!BNA || (SHARE && BI == 64), "the fused BatchNorm backward rides on the shared dY split (one channel per lane)");
  const WgradArgs& a = aa.w;
  constexpr int ES = (int)sizeof(T);
  constexpr int EPV = 16 / ES;                       // elements per 16-byte DMA slot
  constexpr int ROWS = 32;
  constexpr int NW = WVI * WVJ, NT = NW * 64;
  constexpr int RBP = BI * ES, RBQ = BJ * ES;        // row bytes of the dY (P) and im2col (Q) tiles
  static_assert(RBP % 128 == 0 && RBQ % 128 == 0, "row bytes (swizzle, 1 KiB DMA pieces)");
  constexpr int DP = RBP / 16, DQ = RBQ / 16;        // 16-byte slots per row
  constexpr int QOFF = ROWS * RBP, YOFF = ROWS * (RBP + RBQ), BUFB = YOFF + (BNA ? ROWS * RBP : 0);
  constexpr int PPC = QOFF / 1024, QPC = (YOFF - QOFF) / 1024;                   // 1 KiB DMA pieces per tile
  constexpr int NPW = (PPC + NW - 1) / NW, NQW = (QPC + NW - 1) / NW;             // ... per wave (the last round may be partial)
  constexpr int RT = BNA ? NT / 2 : NT;              // rows per decode round: one row per thread (BNA: per thread of the first half)
  constexpr int SPR = RT / ROWS;                     // steps per round
  constexpr int WI = BI / WVI, WJ = BJ / WVJ, TI = WI / 32, TJ = WJ / 32;
  static_assert(TI >= 1 && TJ >= 1, "wave tile");
  constexpr unsigned kOOB = 0x80000000u;

  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * BUFB];
  __shared__ uint2 rowtab[2][RT];
  __shared__ unsigned rowdy[2][RT];                  // byte offset of the row in dY (kOOB beyond the slice)
  // SHARE: [k half * 2 + h][hi | mid | lo][column] bf16x8 fragments of the current step's dY tile
  constexpr int NFRAG = SHARE ? BI * 4 : 1;
  __shared__ uint4 planes[NFRAG * 3];
  static_assert(!SHARE || (NFRAG == NT && WVI == 1), "one dY fragment per thread, every wave spans all of BI");

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_j = bid % a.ntj; bid /= a.ntj;
  const int tile_i = bid % a.nti;
  const int split = bid / a.nti;
  const int i0 = tile_i * BI, j0 = tile_j * BJ;
  const int wi0 = (wave / WVJ) * WI, wj0 = (wave % WVJ) * WJ;
  const int m_begin = split * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);      // (the host sizes the splits so that none is empty)

  const dma_rsrc_t x_rsrc = dma_make_rsrc(a.x, (unsigned)aa.x_bytes), dy_rsrc = dma_make_rsrc(a.dy, (unsigned)aa.dy_bytes);
  const dma_rsrc_t bnx_rsrc = dma_make_rsrc(BNA ? aa.bn_x : a.dy, (unsigned)aa.dy_bytes);
  const unsigned smem_base = lds_addr(smem);
  const unsigned ldxb = (unsigned)a.ldx * ES, ldyb = (unsigned)a.ldy * ES;

  // fixed per-lane roles: LDS slot -> (row, swizzled source column)
  int prow[NPW]; unsigned pcolb[NPW];
#pragma unroll
  for (int u = 0; u < NPW; ++u) {
    const int sl = (wave + NW * u) * 64 + lane;
    prow[u] = sl / DP;
    const int jj = ES == 2 ? ((sl % DP) ^ (wg_swz<RBP>(prow[u]) << 2)) : (sl % DP);
    const int n = i0 + jj * EPV;
    pcolb[u] = n < a.CoutP ? (unsigned)n * ES : kOOB;
  }
  int qrow[NQW]; unsigned qtb[NQW], qbit[NQW];
#pragma unroll
  for (int u = 0; u < NQW; ++u) {
    const int sl = (wave + NW * u) * 64 + lane;
    qrow[u] = sl / DQ;
    const int jj = ES == 2 ? ((sl % DQ) ^ (wg_swz<RBQ>(qrow[u]) << 2)) : (sl % DQ);
    const int col = j0 + jj * EPV;
    if (col < a.J) {
      const int tap = col / g.CP, c = col - tap * g.CP;
      const int dw = tap % g.kw, t2 = tap / g.kw, dh = t2 % g.kh, dt = t2 / g.kh;
      qbit[u] = (1u << dt) | (1u << (8 + dh)) | (1u << (16 + dw));
      qtb[u] = (unsigned)((dt * g.sH + dh) * g.sW + dw) * ldxb + (unsigned)c * ES;
    } else {
      qbit[u] = 0xffffffffu;                         // never matches a 24-bit mask: zero fill
      qtb[u] = 0;
    }
  }

  // decode RT rows of round `rnd` (one per thread) into rowtab[rnd & 1]
  auto decode = [&](int rnd) {
    if (RT < NT && tid >= RT) return;
    const int q = m_begin + rnd * RT + tid;          // position in the row sequence
    uint2 e = make_uint2(0u, 0u);
    unsigned dyo = kOOB;
    if (q < m_end) {
      const uint32_t m = a.perm.on ? perm_row(a.perm, (uint32_t)q) : (uint32_t)q;
      dyo = m * ldyb;
      const RowPos r = decode_row<MODE_FWD>(m, a.M, g);
      auto range = [](int x0, int k, int lim) -> unsigned {      // bits d in [0,k) with 0 <= x0 + d < lim
        const int lo = max(0, -x0), hi = min(k, lim - x0);
        return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
      };
      const unsigned bt = range(r.t0, g.kt, g.sT), bh = range(r.h0, g.kh, g.sH), bw = range(r.w0, g.kw, g.sW);
      e.x = (unsigned)(r.base + (r.t0 * g.sH + r.h0) * g.sW + r.w0) * ldxb;   // modulo 2^32; exact for valid taps
      e.y = (bt && bh && bw) ? (bt | (bh << 8) | (bw << 16)) : 0u;
    }
    rowtab[rnd & 1][tid] = e;
    rowdy[rnd & 1][tid] = dyo;
  };

  auto issue = [&](int s, int buf) {
    const unsigned* dtab = rowdy[(s / SPR) & 1] + (s % SPR) * ROWS;
    unsigned dyo[NPW];
#pragma unroll
    for (int u = 0; u < NPW; ++u) dyo[u] = dtab[prow[u]];
#pragma unroll
    for (int u = 0; u < NPW; ++u) {
      if (PPC % NW != 0 && wave + NW * u >= PPC) break;           // (wave-uniform)
      const unsigned off = (dyo[u] != kOOB && pcolb[u] != kOOB) ? dyo[u] + pcolb[u] : kOOB;
      dma_load16(dy_rsrc, smem_base + buf * BUFB + (wave + NW * u) * 1024, off);
    }
    if constexpr (BNA) {                             // the BatchNorm's input: same rows, same pitch, its own tile
#pragma unroll
      for (int u = 0; u < NPW; ++u) {
        const unsigned off = (dyo[u] != kOOB && pcolb[u] != kOOB) ? dyo[u] + pcolb[u] : kOOB;
        dma_load16(bnx_rsrc, smem_base + buf * BUFB + YOFF + (wave + NW * u) * 1024, off);
      }
    }
    const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(rowtab[(s / SPR) & 1] + (s % SPR) * ROWS);
    unsigned long long e[NQW];
#pragma unroll
    for (int u = 0; u < NQW; ++u) e[u] = tab[qrow[u] & (ROWS - 1)];   // one ds_read_b64 each, issued back to back
#pragma unroll
    for (int u = 0; u < NQW; ++u) {
      if (QPC % NW != 0 && wave + NW * u >= QPC) break;
      const unsigned ex = (unsigned)e[u], ey = (unsigned)(e[u] >> 32);
      const unsigned off = ((ey & qbit[u]) == qbit[u]) ? ex + qtb[u] : kOOB;
      dma_load16(x_rsrc, smem_base + buf * BUFB + QOFF + (wave + NW * u) * 1024, off);
    }
  };
  // pieces this wave issues per step (wave-uniform)
  int my_pieces = 0;
#pragma unroll
  for (int u = 0; u < NPW; ++u) my_pieces += (wave + NW * u < PPC) ? (BNA ? 2 : 1) : 0;
#pragma unroll
  for (int u = 0; u < NQW; ++u) my_pieces += (wave + NW * u < QPC) ? 1 : 0;

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int h = lane >> 5, l31 = lane & 31;
  // BNA: this thread's channel (= its column of the dY tile) and dv_bn_bwd_apply's coefficients for it, same expressions
  float bk1 = 0.f, bk2 = 0.f, bk3 = 0.f, bsc = 0.f, bsh = 0.f;
  if constexpr (BNA) {
    const int c = i0 + lane, cpb = (a.Cout + 7) & ~7;
    if (c < a.Cout) {
      float sg = 0.f, sgx = 0.f;
      for (int r = 0; r < aa.bn_rep; ++r) { sg += aa.bn_sums[(size_t)r * 2 * cpb + c]; sgx += aa.bn_sums[(size_t)r * 2 * cpb + cpb + c]; }
      bk1 = aa.bn_gamma[c] * aa.bn_invstd[c];
      bk2 = -bk1 * aa.bn_invstd[c] * sgx * aa.bn_inv_count;
      bk3 = -bk1 * sg * aa.bn_inv_count - bk2 * aa.bn_mean[c];
      if (aa.bn_mask) { bsc = aa.bn_scale[c]; bsh = aa.bn_shift[c]; }
      if (split == 0 && tile_j == 0 && wave == 0 && aa.bn_dgamma) {       // one workgroup per channel tile: dgamma, dbeta
        aa.bn_dbeta[c] += aa.bn_dscale * sg;
        aa.bn_dgamma[c] += aa.bn_dscale * sgx;
      }
    }
  }
  const int nsteps = (m_end - m_begin + ROWS - 1) / ROWS;
  constexpr int D = NS - 1;                          // prefetch distance (the row table runs one round = SPR steps ahead)
  static_assert(D >= 1 && D < SPR, "stages");
  decode(0);
  __syncthreads();
  for (int t = 0; t < D && t < nsteps; ++t) issue(t, t);
  int cur = 0, nxt = D % NS;
  for (int s = 0; s < nsteps; ++s) {
    dma_wait_upto(min(D - 1, nsteps - 1 - s) * my_pieces);         // tile s has landed (this wave's pieces)
    __syncthreads();                                 // ... everybody's; stage `nxt` (tile s-1) and the old row table are free
    // table of round R+1 is written during the first step of round R; its previous contents (round R-1) were last read
    // at least one barrier ago, and its first reader (the issue for step SPR(R+1), at step SPR(R+1)-D) is later
    if ((s % SPR) == 0 && (s + SPR) < nsteps) decode(s / SPR + 1);
    if (s + D < nsteps) issue(s + D, nxt);
    const unsigned char* tp = smem + cur * BUFB;
    const unsigned char* tq = tp + QOFF;
    if constexpr (ES == 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[TI], bf[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = wg_frag<RBP>(tp, wi0 + i * 32, lane, ks);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bf[j] = wg_frag<RBQ>(tq, wj0 + j * 32, lane, ks);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (SHARE) {
      // split phase: thread -> dY fragment (kh = k half * 2 + h = its wave, column = its lane): conflict-free dword reads of 8
      // rows of a column, 16-byte writes of the three planes
      {
        const int kh = wave, col = lane;
        const int rr0 = (kh >> 1) * 16 + 8 * (kh & 1);
        const unsigned char* src = tp + rr0 * RBP + col * 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(src + e * RBP);
        if constexpr (BNA) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            // (rows past the end of the split need no special case: their im2col rows are zero-filled too, so whatever this forms
            // for them -- k3 -- is multiplied by zeros)
            const float xv = *reinterpret_cast<const float*>(src + YOFF + e * RBP);
            const float act = xv * bsc + bsh;              // the forward's expression (dv_bn_apply), same rounding
            const float gg = (aa.bn_mask && !(act > 0.f)) ? 0.f : v[e];
            v[e] = bk1 * gg + bk2 * xv + bk3;              // dv_bn_bwd_apply's expression
          }
        }
        const Split3 s3 = split3w(v);
        uint4* dst = planes + kh * 3 * BI + col;
        dst[0] = __builtin_bit_cast(uint4, s3.hi);
        dst[BI] = __builtin_bit_cast(uint4, s3.mid);
        dst[2 * BI] = __builtin_bit_cast(uint4, s3.lo);
      }
      // this wave's own x fragments (both K halves) are split while the other waves finish their dY fragment (splitting K half 1
      // under the MFMAs of K half 0 instead was measured equal)
      Split3 bfr[2][TJ];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(tq + (ks * 16 + 8 * h + e) * RBQ + (wj0 + j * 32 + l31) * 4);
          bfr[ks][j] = split3w(v);
        }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Split3 af[TI];
        const int kh = ks * 2 + h;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
          const uint4* pa = planes + kh * 3 * BI + wi0 + i * 32 + l31;
          af[i].hi = __builtin_bit_cast(bf16x8, pa[0]);
          af[i].mid = __builtin_bit_cast(bf16x8, pa[BI]);
          af[i].lo = __builtin_bit_cast(bf16x8, pa[2 * BI]);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) mma_split3(af[i], bfr[ks][j], acc[i][j]);
      }
    } else if constexpr (SPLIT) {
      // f32 tiles, products on the bf16 matrix cores (split3 above): the K index of this GEMM is the ROW, so a lane's
      // fragment is 8 rows of one column (conflict-free dword reads, immediate offsets)
      // Software pipeline inside the step: the fragments of K half 1 are split while the MFMAs of K half 0 run.  In program order
      // a wave would split everything (vector ALU only), then multiply (matrix pipe only); sched_group_barrier makes the
      // scheduler emit one MFMA, then its share of the split instructions, so both pipes work at once.
      auto frag_a = [&](int ks, int i, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(tp + (ks * 16 + 8 * h + e) * RBP + (wi0 + i * 32 + l31) * 4);
      };
      auto frag_b = [&](int ks, int j, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(tq + (ks * 16 + 8 * h + e) * RBQ + (wj0 + j * 32 + l31) * 4);
      };
      Split3 a0[TI], b0[TJ], a1[TI], b1[TJ];
      float ra[TI][8], rb[TJ][8];
#pragma unroll
      for (int i = 0; i < TI; ++i) { float v[8]; frag_a(0, i, v); a0[i] = split3w(v); }
#pragma unroll
      for (int j = 0; j < TJ; ++j) { float v[8]; frag_b(0, j, v); b0[j] = split3w(v); }
#pragma unroll
      for (int i = 0; i < TI; ++i) frag_a(1, i, ra[i]);
#pragma unroll
      for (int j = 0; j < TJ; ++j) frag_b(1, j, rb[j]);
#pragma unroll
      for (int i = 0; i < TI; ++i) a1[i] = split3w(ra[i]);
#pragma unroll
      for (int j = 0; j < TJ; ++j) b1[j] = split3w(rb[j]);
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) mma_split3(a0[i], b0[j], acc[i][j]);
      {
        constexpr int NM = TI * TJ * 6, VPM = (44 * (TI + TJ) + NM - 1) / NM;
#pragma unroll
        for (int qq = 0; qq < NM; ++qq) {
          __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x2, VPM, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) mma_split3(a1[i], b1[j], acc[i][j]);
    } else {
#pragma unroll 4
      for (int ks = 0; ks < ROWS / 2; ++ks) {
        float af[TI], bf[TJ];
        const int kr = ks * 2 + h;
#pragma unroll
        for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const float*>(tp + kr * RBP + (wi0 + i * 32 + l31) * 4);
#pragma unroll
        for (int j = 0; j < TJ; ++j) bf[j] = *reinterpret_cast<const float*>(tq + kr * RBQ + (wj0 + j * 32 + l31) * 4);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
    nxt = nxt + 1 == NS ? 0 : nxt + 1;
  }

#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
      wgrad_store_block(a, split, i0 + wi0 + i * 32, j0 + wj0 + j * 32 + l31, h, acc[i][j]);
}


// ------------------------------------------------------------------------------------------
// host side
static bool fill_geom(const dv_conv_desc* d, int mode, ConvGeom& g) {
  g.kt = d->kt; g.kh = d->kh; g.kw = d->kw;
  g.st = d->st; g.sh = d->sh; g.sw = d->sw;
  g.pt = d->pt; g.ph = d->ph; g.pw = d->pw;
  if (mode == MODE_FWD) {
    g.rT = d->To; g.rH = d->Ho; g.rW = d->Wo;
    g.sT = d->Ti; g.sH = d->Hi; g.sW = d->Wi;
    g.CP = d->cin_pitch;
  } else {
    g.rT = d->Ti; g.rH = d->Hi; g.rW = d->Wi;
    g.sT = d->To; g.sH = d->Ho; g.sW = d->Wo;
    g.CP = d->cout_pitch;
  }
  g.Ktot = d->kt * d->kh * d->kw * g.CP;
  g.dW = make_fastdiv((uint32_t)g.rW);
  g.dH = make_fastdiv((uint32_t)g.rH);
  g.dT = make_fastdiv((uint32_t)g.rT);
  return true;
}

static int check_desc(const dv_conv_desc* d) {
  if (!d) return DV_EINVAL;
  if (d->dtype != DV_F32 && d->dtype != DV_BF16) return DV_EUNSUPPORTED;
  if (d->N <= 0 || d->Ti <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Cin <= 0 || d->Cout <= 0) return DV_EINVAL;
  if (d->kt <= 0 || d->kh <= 0 || d->kw <= 0 || d->st <= 0 || d->sh <= 0 || d->sw <= 0) return DV_EINVAL;
  if (d->pt < 0 || d->ph < 0 || d->pw < 0) return DV_EINVAL;
  if ((d->Ti + 2 * d->pt - d->kt) / d->st + 1 != d->To) return DV_EINVAL;
  if ((d->Hi + 2 * d->ph - d->kh) / d->sh + 1 != d->Ho) return DV_EINVAL;
  if ((d->Wi + 2 * d->pw - d->kw) / d->sw + 1 != d->Wo) return DV_EINVAL;
  if (d->To <= 0 || d->Ho <= 0 || d->Wo <= 0) return DV_EINVAL;
  const int gv_min = 4;
  if (d->cin_pitch < d->Cin || d->cin_pitch % gv_min) return DV_EINVAL;
  if (d->cout_pitch < d->Cout || d->cout_pitch % 8) return DV_EINVAL;
  if (d->ldx < d->cin_pitch || d->ldy < d->cout_pitch) return DV_EINVAL;
  const int esz = d->dtype == DV_F32 ? 4 : 2;
  if ((d->ldx * esz) % 8 || (d->ldy * esz) % 16) return DV_EALIGN;
  const int64_t mi = (int64_t)d->N * d->Ti * d->Hi * d->Wi, mo = (int64_t)d->N * d->To * d->Ho * d->Wo;
  if (mi >= (1ll << 31) || mo >= (1ll << 31)) return DV_EINVAL;
  return DV_OK;
}

// gather vector bytes for a channel pitch
static int gather_bytes(int dtype, int cp) {
  if (dtype == DV_F32) return 16;                   // 4 floats; cp % 4 == 0 checked
  return (cp % 8 == 0) ? 16 : 8;                    // bf16: 8 elements, or 4 for the padded RGB input
}

// ---- pre-split weights of the fp32 split mode (DV_W3) ---------------------------------------------------------------------
static int w3_rows(int n) { return (n + 127) / 128 * 128; }                         // rows incl. padding (any tile fits)
static int64_t w3_bytes(int n, int ktot) { return (int64_t)((ktot + 15) / 16) * 2 * w3_rows(n) * 48; }

namespace {
// one thread per (K tile, k half, row): 8 floats of W[row][kt*16 + 8h ..] -> hi | mid | lo, 48 bytes
__global__ __launch_bounds__(256) void pack_w3_kernel(const float* __restrict__ base, unsigned char* __restrict__ out_base,
                                                      const dv_w3_desc* __restrict__ descs, const int2* __restrict__ block_map) {
  const int2 bm = block_map[blockIdx.x];                     // (descriptor, first unit of this block)
  const dv_w3_desc d = descs[bm.x];
  const int npad = (d.N + 127) / 128 * 128;
  const long long unit = (long long)bm.y + threadIdx.x;      // unit = (kt * 2 + h) * npad + n
  const long long units = (long long)((d.Ktot + 15) / 16) * 2 * npad;
  if (unit >= units) return;
  const int n = (int)(unit % npad);
  const int kh = (int)(unit / npad), k0 = (kh >> 1) * 16 + (kh & 1) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (n < d.N && k0 + e < d.Ktot) ? base[d.src_off + (long long)n * d.Ktot + k0 + e] : 0.f;
  const Split3 s3 = split3(v);
  bf16x8* o = reinterpret_cast<bf16x8*>(out_base + d.dst_off + unit * 48);
  o[0] = s3.hi; o[1] = s3.mid; o[2] = s3.lo;
}
}  // namespace

extern "C" int64_t dv_w3_bytes(int32_t rows, int32_t ktot) { return rows > 0 && ktot > 0 ? w3_bytes(rows, ktot) : 0; }

extern "C" int dv_pack_w3(const float* base, void* out_base, const dv_w3_desc* descs, const int32_t* block_map, int32_t n_blocks,
                          void* stream) {
  if (!base || !out_base || !descs || !block_map || n_blocks <= 0) return DV_EINVAL;
  hipLaunchKernelGGL(pack_w3_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, base, (unsigned char*)out_base, descs,
                     reinterpret_cast<const int2*>(block_map));
  return dv_launch_status();
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

static bool f32_exact() {
  static const bool v = getenv("DUALVAR_F32_EXACT") && atoi(getenv("DUALVAR_F32_EXACT")) != 0;
  return v;
}

static int pick_bn(int np) {
  if (np <= 32) return 32;
  if (np <= 64) return 64;
  int w128 = (np + 127) / 128 * 128, w64 = (np + 63) / 64 * 64;
  return (w128 <= w64) ? 128 : 64;
}

// Tile rows: 128, or 64 when a 128-row grid would leave most CUs with fewer than ~4 workgroups (the K loop is
// latency bound at low occupancy; twice as many, half as tall workgroups hide it)
static int pick_bm(int M, int ntn) {
  return ((int64_t)((M + 127) / 128) * ntn >= 1024) ? 128 : 64;
}

// Tile (rows x columns) of the fwd / dgrad GEMM over M rows and NP (padded) columns: columns 32 / 64 / 128 by least
// padding, rows 128 when that still gives >= 1024 workgroups, else 64.  (96- and 192-column tiles -- wave tiles 32x96
// and 64x96 -- were measured too: 15-20 % faster on random data in isolation, but equal within noise inside the real
// training step, where post-ReLU activations are half zeros and the clocks are higher; not kept.)
// fp32 through the bf16 split (split3): every fragment costs ~44 vector instructions to split, so the kernel is bound
// by the vector ALU unless a fragment feeds enough MFMAs -- 256 x 64 tiles (wave tiles 64x64: 1 fragment per 32x32 block
// instead of the 1.5 of 128 x 64) wherever the grid still fills the chip twice over (c2c 1x3x3 forward 664 -> 609 us;
// a 256 x 128 tile with 128 x 64 wave tiles spills registers and is far slower).
static void pick_tile(int dtype, int M, int NP, int& bm, int& bn) {
  bn = pick_bn(NP);
  const int ntn = (NP + bn - 1) / bn;
  bm = pick_bm(M, ntn);
  if (dtype == DV_F32 && !f32_exact() && bm == 128 && bn == 64 && (int64_t)((M + 255) / 256) * ntn >= 512) bm = 256;
  // Few rows (the 1 152-row layers of Mixed_5b/5c: 18 row tiles): 128-column tiles leave three quarters of the CUs without a
  // workgroup while each busy CU runs one latency-bound K loop -- narrower tiles until the grid covers the chip
  constexpr int fill = 256;
  if (bm == 64) {
    const int64_t mt = (M + 63) / 64;
    while (bn > 32 && mt * ((NP + bn - 1) / bn) < fill) bn >>= 1;
  }
}

template <typename T, int MODE, int GVB, int GM, int NS, bool SPLIT = false, bool WF = false>
static void launch_gemm_ns(int bm, int bn, const ConvArgs& a, int grid, hipStream_t s) {
  if constexpr (WF) {       // pre-split weights: the instantiations the fp32 split mode uses
    if constexpr (MODE == MODE_FWD && GM == 1 && NS == 2 && sizeof(T) == 4) {
      if (a.in_scale != nullptr) {             // BatchNorm on load (fwd_impl has checked: 256 x 64 tile, channel pitch <= 256)
        hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 256, 64, 4, 1, GM, NS, true, true, true>), dim3(grid), dim3(256), 0, s, a);
        return;
      }
    }
    if (bm == 256) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 256, 64, 4, 1, GM, NS, true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (bm == 64 && bn == 32) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 64, 32, 2, 1, GM, NS, true, true>), dim3(grid), dim3(128), 0, s, a);
    else if (bm == 64 && bn == 64) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 64, 64, 2, 2, GM, NS, true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (bm == 64) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 64, 128, 2, 2, GM, NS, true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (bn == 32) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 128, 32, 4, 1, GM, NS, true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (bn == 64) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 128, 64, 4, 1, GM, NS, true, true>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 128, 128, 2, 2, GM, NS, true, true>), dim3(grid), dim3(256), 0, s, a);
    return;
  }
  if constexpr (SPLIT) {
    if (bm == 256) {
      hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 256, 64, 4, 1, GM, NS, SPLIT>), dim3(grid), dim3(256), 0, s, a);
      return;
    }
  }
  if (bm == 64) {
    if (bn == 32) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 64, 32, 2, 1, GM, NS, SPLIT>), dim3(grid), dim3(128), 0, s, a);
    else if (bn == 64) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 64, 64, 2, 2, GM, NS, SPLIT>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 64, 128, 2, 2, GM, NS, SPLIT>), dim3(grid), dim3(256), 0, s, a);
    return;
  }
  if (bn == 32) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 128, 32, 4, 1, GM, NS, SPLIT>), dim3(grid), dim3(256), 0, s, a);
  else if (bn == 64) hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 128, 64, 4, 1, GM, NS, SPLIT>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_gemm_kernel<T, MODE, GVB, 128, 128, 2, 2, GM, NS, SPLIT>), dim3(grid), dim3(256), 0, s, a);
}



// Two LDS stages where the grid fills the chip several times over: there three to six stages measured equal or slower
// (they cost resident workgroups, and the K loop is issue bound).  Launches of at most ~2 workgroups per CU (64-row tiles
// of the 12 544- and 1 152-row layers) are latency bound instead -- one workgroup per CU waits out every DMA round trip --
// and get four stages: those layers 252 -> 200 us (1 152 rows) and 581 -> 541 us (12 544 rows) per pass, step 10.53 ->
// 10.35 ms.  Three stages, and six on the smallest grids, were measured equal or slower.
template <typename T, int MODE, int GVB, int GM>
static void launch_gemm_gm(int bm, int bn, const ConvArgs& a, int grid, hipStream_t s) {
  if constexpr (GVB == 16 && sizeof(T) == 2) {
    if (bm == 64 && grid <= 512) {
      launch_gemm_ns<T, MODE, GVB, GM, 4>(bm, bn, a, grid, s);
      return;
    }
  }
  if constexpr (sizeof(T) == 4) {
    if constexpr (GM == 1) {                     // (only the pre-split-weight kernels are built with uniform-tap gathers)
      // (four or six stages for the small grids, as for bf16, were measured SLOWER here: Mixed_5c 1x3x3 forward 57 -> 68 -> 71 us;
      // with one workgroup per CU the K step is bound by its own ds_read -> split -> MFMA chain, not by the DMA round trip)
      launch_gemm_ns<T, MODE, GVB, 1, 2, true, true>(bm, bn, a, grid, s);
      return;
    } else {
      if (!f32_exact()) {
        if (a.flags & DV_W3) launch_gemm_ns<T, MODE, GVB, 0, 2, true, true>(bm, bn, a, grid, s);
        else launch_gemm_ns<T, MODE, GVB, 0, 2, true>(bm, bn, a, grid, s);
        return;
      }
    }
  }
  if constexpr (sizeof(T) != 4 || GM == 0) launch_gemm_ns<T, MODE, GVB, GM, 2>(bm, bn, a, grid, s);
}

template <typename T, int MODE, int GVB>
static void launch_gemm(int bm, int bn, const ConvArgs& a, int grid, hipStream_t s) {
  if constexpr (GVB == 16 && (sizeof(T) == 2 || sizeof(T) == 4)) {
    // uniform-tap gathers: every K-step (64 bytes: 32 bf16 / 16 f32) inside one tap, tap bit mask in one register.
    // f32: for the pre-split-weight kernels (the fp32 split mode's hot path; its K loop is bound by vector instructions)
    constexpr int BKE_ = 64 / (int)sizeof(T);
    if (a.g.CP % BKE_ == 0 && a.g.kt * a.g.kh * a.g.kw <= 32 && (sizeof(T) == 2 || ((a.flags & DV_W3) && !f32_exact()))) {
      launch_gemm_gm<T, MODE, GVB, 1>(bm, bn, a, grid, s);
      return;
    }
  }
  launch_gemm_gm<T, MODE, GVB, 0>(bm, bn, a, grid, s);
}

}  // namespace

// Taps that fall into the padding for EVERY row of the launch contribute nothing: run the window of the live taps instead
// (per axis a contiguous range [lo, hi]).  Mixed_5b / 5c of S3D-G at 8-frame clips have ONE frame left: their 3x1x1 convs
// (padding 1) keep one tap of three -- a third of the K loop.  Weights stay where they are: the kernel addresses a live tap
// at its original index (ConvArgs::cls_on == 2).  Stride-1 data gradients and forward convs; not for the generic-gather
// pre-split-weight path (its K tiles straddle taps).
static void trim_dead_taps(ConvArgs& a, int mode, int dtype) {
  ConvGeom& g = a.g;
  if (a.cls_on || g.kt * g.kh * g.kw <= 1) return;
  if (mode == MODE_DGRAD && (g.st > 1 || g.sh > 1 || g.sw > 1)) return;
  if ((a.flags & DV_W3) && (g.CP % 16 != 0 || g.kt * g.kh * g.kw > 32)) return;
  (void)dtype;
  auto live = [&](int k, int rdim, int sdim, int stride, int pad, int& lo, int& hi) {
    lo = k; hi = -1;
    for (int d = 0; d < k; ++d) {
      bool any = false;
      for (int o = 0; o < rdim && !any; ++o) {
        const int v = mode == MODE_FWD ? o * stride - pad + d : o + pad - d;
        any = v >= 0 && v < sdim;
      }
      if (any) { lo = std::min(lo, d); hi = std::max(hi, d); }
    }
  };
  int lt, ht, lh, hh, lw, hw;
  live(g.kt, g.rT, g.sT, g.st, g.pt, lt, ht);
  live(g.kh, g.rH, g.sH, g.sh, g.ph, lh, hh);
  live(g.kw, g.rW, g.sW, g.sw, g.pw, lw, hw);
  if (ht < lt || hh < lh || hw < lw) return;                   // (no live tap at all: leave it to the zero fill)
  if (ht - lt + 1 == g.kt && hh - lh + 1 == g.kh && hw - lw + 1 == g.kw) return;
  a.cls_on = 2;
  a.cst = a.csh = a.csw = 1; a.cot = a.coh = a.cow = 0;
  a.crt = lt; a.crh = lh; a.crw = lw; a.oKH = g.kh; a.oKW = g.kw; a.oT = g.rT; a.oH = g.rH; a.oW = g.rW;
  // tap d' = d - lo: forward t = o*s - p + d = o*s - (p - lo) + d'; data gradient t = o + p - d = o + (p - lo) - d'
  g.kt = ht - lt + 1; g.kh = hh - lh + 1; g.kw = hw - lw + 1;
  g.pt -= lt; g.ph -= lh; g.pw -= lw;
  g.Ktot = g.kt * g.kh * g.kw * g.CP;
}

// conv_gemm_ks_kernel (K split over the waves) takes a fwd / dgrad launch when the ordinary tiling leaves a small grid with a
// long K loop.  Returns the column tile (32 / 64) or 0.  DUALVAR_CONV_KS=0 switches it off (A/B runs).
static int ks_tile(int dtype, const ConvArgs& a, int bm) {
  static const int on = env_int("DUALVAR_CONV_KS", 1);
  constexpr int max_grid = 400, min_nk = 16;       // (a larger grid limit was measured slower: 19.54 -> 20.44 ms at 1024)
  if (!on || dtype != DV_F32 || !(a.flags & DV_W3) || f32_exact() || a.cls_on == 1 || a.bn_x != nullptr || bm != 64) return 0;
  if (a.g.CP % 16 != 0 || a.g.kt * a.g.kh * a.g.kw > 32 || a.g.Ktot / 16 < min_nk) return 0;
  if (a.g.st > 1 || a.g.sh > 1 || a.g.sw > 1) return 0;       // (few-row layers are stride 1; keeps one gather form)
  const int64_t mt = (a.M + 63) / 64;
  const int64_t g32 = mt * ((a.NP + 31) / 32), g64 = mt * ((a.NP + 63) / 64);
  // one workgroup per CU (its four private pipelines fill the LDS): a grid of at most one round, and as many workgroups as that
  // allows; 64-column tiles when 32-column ones would need a second round
  if (g32 <= 256) return 32;
  if (g64 <= max_grid) return g64 <= 256 || g32 > max_grid ? 64 : 32;
  return 0;
}
template <int MODE>
static void launch_ks(int bn, ConvArgs& a, hipStream_t s) {
  a.ntn = (a.NP + bn - 1) / bn;
  const int grid = a.ntn * ((a.M + 63) / 64);
  if (bn == 32) hipLaunchKernelGGL((conv_gemm_ks_kernel<MODE, 32, 4>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_gemm_ks_kernel<MODE, 64, 3>), dim3(grid), dim3(256), 0, s, a);
}

// the parity classes of a strided data gradient (ConvArgs::cls_on == 1): `a` holds the launch-wide fields; -> number of classes
static int dgrad_classes(const dv_conv_desc* d, const ConvArgs& a, ConvArgs (&out)[8]) {
  const int taps_orig = d->kt * d->kh * d->kw;
  int n = 0;
  for (int rt = 0; rt < d->st; ++rt)
    for (int rh = 0; rh < d->sh; ++rh)
      for (int rw = 0; rw < d->sw; ++rw) {
        auto first = [](int r, int p, int st_) { return ((r - p) % st_ + st_) % st_; };   // smallest pos with (pos+p)%s == r
        const int ot = first(rt, d->pt, d->st), oh = first(rh, d->ph, d->sh), ow = first(rw, d->pw, d->sw);
        if (ot >= d->Ti || oh >= d->Hi || ow >= d->Wi) continue;
        ConvArgs& c = out[n++];
        c = a;
        ConvGeom& g = c.g;
        g.rT = (d->Ti - ot + d->st - 1) / d->st; g.rH = (d->Hi - oh + d->sh - 1) / d->sh; g.rW = (d->Wi - ow + d->sw - 1) / d->sw;
        g.kt = (d->kt - rt + d->st - 1) / d->st; g.kh = (d->kh - rh + d->sh - 1) / d->sh; g.kw = (d->kw - rw + d->sw - 1) / d->sw;
        g.st = g.sh = g.sw = 1;
        g.pt = (ot + d->pt - rt) / d->st; g.ph = (oh + d->ph - rh) / d->sh; g.pw = (ow + d->pw - rw) / d->sw;
        g.Ktot = g.kt * g.kh * g.kw * g.CP;
        g.dW = make_fastdiv((uint32_t)g.rW); g.dH = make_fastdiv((uint32_t)g.rH); g.dT = make_fastdiv((uint32_t)g.rT);
        c.M = d->N * g.rT * g.rH * g.rW;
        if (!(a.flags & DV_W3)) c.ldw = taps_orig * g.CP;       // (pre-split weights: ldw stays their padded row count)
        c.cls_on = 1; c.cst = d->st; c.csh = d->sh; c.csw = d->sw; c.cot = ot; c.coh = oh; c.cow = ow;
        c.crt = rt; c.crh = rh; c.crw = rw; c.oKH = d->kh; c.oKW = d->kw; c.oT = d->Ti; c.oH = d->Hi; c.oW = d->Wi;
      }
  return n;
}

// the ConvArgs fields the kernel-choice queries look at (no pointers: nothing is launched)
static void query_args(const dv_conv_desc* d, int dgrad, ConvArgs& a) {
  fill_geom(d, dgrad ? MODE_DGRAD : MODE_FWD, a.g);
  a.M = dgrad ? d->N * d->Ti * d->Hi * d->Wi : d->N * d->To * d->Ho * d->Wo;
  a.N = dgrad ? d->Cin : d->Cout;
  a.NP = dgrad ? d->cin_pitch : d->cout_pitch;
  a.ldo = dgrad ? d->ldx : d->ldy;
  a.flags = d->flags & (dgrad ? (DV_W3 | DV_ACCUM) : (DV_W3 | DV_BIAS | DV_RELU | DV_SIGMOID | DV_STATS));
  a.cls_on = 0;
  a.bn_x = nullptr; a.bn_ws = nullptr; a.bn_bytes = 0;
  const int64_t ob = (((int64_t)a.M - 1) * a.ldo + a.NP) * 4;
  a.out_bytes = (d->dtype == DV_F32 && ob < (1ll << 31)) ? (int)ob : 0;
}

// rows per tile when dv_conv3d_fwd runs this problem on the pixel-pair stem form (conv_tap.hip: conv_pp_fwd_kernel), else 0
static int pp_rows_choice(const dv_conv_desc* d) {
  if (d->dtype != DV_F32 || f32_exact() || !(d->flags & DV_W3)) return 0;
  ConvArgs a;
  query_args(d, 0, a);
  a.lds_ = d->ldx;
  return dvt_conv_pp_rows(&a, MODE_FWD);
}

// 0: conv_gemm / conv_gemm_ks; 1 / 2: the LDS-staged input-tile kernel (conv_tap.hip), spatial / temporal form; 3: its pixel-pair
// stem form (forward only)
static int tap_choice(const dv_conv_desc* d, int dgrad) {
  if (d->dtype != DV_F32 || f32_exact() || !(d->flags & DV_W3)) return 0;
  if (!dgrad && pp_rows_choice(d)) return 3;
  ConvArgs a;
  query_args(d, dgrad, a);
  if (d->st > 1 || d->sh > 1 || d->sw > 1) {
    // a strided data gradient whose parity classes all run on the temporal form (the 7x1x1 / stride-2 stem conv); forward: never
    if (!dgrad || d->st > 2 || d->sh > 2 || d->sw > 2 || !(d->kt >= d->st && d->kh >= d->sh && d->kw >= d->sw)) return 0;
    a.ldw = w3_rows(d->Cin);
    ConvArgs cls[8];
    const int ncls = dgrad_classes(d, a, cls);
    for (int i = 0; i < ncls; ++i)
      if (!dvt_conv_tap_kind(&cls[i], MODE_DGRAD)) return 0;
    return ncls > 0 ? 2 : 0;
  }
  trim_dead_taps(a, dgrad ? MODE_DGRAD : MODE_FWD, d->dtype);
  return dvt_conv_tap_kind(&a, dgrad ? MODE_DGRAD : MODE_FWD);
}

extern "C" int dv_conv3d_tap_kind(const dv_conv_desc* d, int32_t dgrad) {
  if (!d || check_desc(d)) return 0;
  return tap_choice(d, dgrad);
}

extern "C" int dv_conv3d_tap_rows(const dv_conv_desc* d, int32_t dgrad) {
  if (!d || check_desc(d)) return 0;
  const int kind = tap_choice(d, dgrad);
  if (!kind) return 0;
  if (kind == 3) return pp_rows_choice(d);
  if (d->st > 1 || d->sh > 1 || d->sw > 1) return 256;           // (the parity classes of a strided data gradient)
  ConvArgs a;
  query_args(d, dgrad, a);
  trim_dead_taps(a, dgrad ? MODE_DGRAD : MODE_FWD, d->dtype);
  return dvt_conv_tap_rows(&a, dgrad ? MODE_DGRAD : MODE_FWD);
}

extern "C" int dv_conv3d_tile_rows(const dv_conv_desc* d) {
  if (!d) return DV_EINVAL;
  if (!check_desc(d)) {
    if (const int r = pp_rows_choice(d)) return r;
    if (tap_choice(d, 0)) {              // (stride 1 here: the strided forward never runs on the LDS-staged kernel)
      ConvArgs a;
      query_args(d, 0, a);
      trim_dead_taps(a, MODE_FWD, d->dtype);
      return dvt_conv_tap_rows(&a, MODE_FWD);
    }
  }
  const int64_t m = (int64_t)d->N * d->To * d->Ho * d->Wo;
  int bm, bn;
  pick_tile(d->dtype, (int)m, d->cout_pitch, bm, bn);
  return bm;
}

extern "C" int dv_conv3d_tile_shape(const dv_conv_desc* d, int32_t dgrad, int32_t* rows, int32_t* cols) {
  if (!d || !rows || !cols) return DV_EINVAL;
  int bm, bn;
  if (dgrad) {
    const int64_t m = (int64_t)d->N * d->Ti * d->Hi * d->Wi;
    pick_tile(d->dtype, (int)m, d->cin_pitch, bm, bn);
  } else {
    const int64_t m = (int64_t)d->N * d->To * d->Ho * d->Wo;
    pick_tile(d->dtype, (int)m, d->cout_pitch, bm, bn);
  }
  *rows = bm; *cols = bn;
  return DV_OK;
}

extern "C" int dv_conv3d_ksplit_cols(const dv_conv_desc* d, int32_t dgrad) {
  if (!d || check_desc(d)) return 0;
  if (dgrad && (d->st > 1 || d->sh > 1 || d->sw > 1)) return 0;
  ConvArgs a;
  fill_geom(d, dgrad ? MODE_DGRAD : MODE_FWD, a.g);
  a.M = dgrad ? d->N * d->Ti * d->Hi * d->Wi : d->N * d->To * d->Ho * d->Wo;
  a.N = dgrad ? d->Cin : d->Cout;
  a.NP = dgrad ? d->cin_pitch : d->cout_pitch;
  a.flags = d->flags & DV_W3;
  a.cls_on = 0;
  a.bn_x = nullptr; a.bn_ws = nullptr; a.bn_bytes = 0;
  trim_dead_taps(a, dgrad ? MODE_DGRAD : MODE_FWD, d->dtype);
  int bm, bn;
  pick_tile(d->dtype, a.M, a.NP, bm, bn);
  return ks_tile(d->dtype, a, bm);
}

extern "C" int dv_conv3d_stat_tiles(const dv_conv_desc* d) {
  if (!d) return DV_EINVAL;
  const int64_t m = (int64_t)d->N * d->To * d->Ho * d->Wo;
  const int bm = dv_conv3d_tile_rows(d);
  return (int)((m + bm - 1) / bm);
}

// which forward kernel can apply a BatchNorm to its INPUT (dv_conv3d_fwd_bn_in): 1 the LDS-staged temporal form, 2 conv_gemm's
// 256 x 64 uniform-tap form; 0 none
static int fwd_bn_in_path(const dv_conv_desc* d) {
  if (d->dtype != DV_F32 || f32_exact() || !(d->flags & DV_W3) || (d->flags & (DV_BIAS | DV_RELU | DV_SIGMOID))) return 0;
  if (d->cin_pitch % 16 != 0) return 0;
  static const float dummy = 0.f;
  ConvArgs a;
  query_args(d, 0, a);
  a.in_scale = &dummy;
  if (d->st == 1 && d->sh == 1 && d->sw == 1) {
    trim_dead_taps(a, MODE_FWD, d->dtype);
    if (a.cls_on) return 0;
    if (dvt_conv_tap_kind(&a, MODE_FWD)) return 1;
    if (tap_choice(d, 0)) return 0;          // (the plain launch would take the spatial LDS-staged form: keep it)
  }
  if (d->cin_pitch > 256 || d->kt * d->kh * d->kw > 32) return 0;
  int bm, bn;
  pick_tile(d->dtype, a.M, a.NP, bm, bn);
  if (bm != 256 || bn != 64 || ks_tile(d->dtype, a, bm)) return 0;
  return 2;
}

static int fwd_impl(const dv_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* stats,
                    const dv_bn_in* bn_in, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!x || !w || !y) return DV_EINVAL;
  if ((d->flags & DV_BIAS) && !bias) return DV_EINVAL;
  if ((d->flags & DV_STATS) && !stats) return DV_EINVAL;
  if (!aligned16(w) || !aligned16(y) || (reinterpret_cast<uintptr_t>(x) & 7)) return DV_EALIGN;
  ConvArgs a;
  fill_geom(d, MODE_FWD, a.g);
  a.src = x; a.w = w; a.out = y; a.bias = bias; a.stats = stats; a.sc_a = a.sc_b = nullptr; a.bn_x = nullptr; a.bn_ws = nullptr; a.bn_bytes = 0;
  a.M = d->N * d->To * d->Ho * d->Wo;
  a.N = d->Cout; a.NP = d->cout_pitch;
  a.lds_ = d->ldx; a.ldo = d->ldy; a.ldw = a.g.Ktot;
  a.flags = d->flags & (DV_BIAS | DV_RELU | DV_SIGMOID | DV_STATS);
  const bool w3 = (d->flags & DV_W3) != 0;
  if (w3 && (d->dtype != DV_F32 || f32_exact())) return DV_EUNSUPPORTED;
  if (w3) { a.flags |= DV_W3; a.ldw = w3_rows(d->Cout); }
  a.cls_on = 0;
  int bn_path = 0;
  if (bn_in) {
    if (!bn_in->scale || !bn_in->shift) return DV_EINVAL;
    bn_path = fwd_bn_in_path(d);
    if (!bn_path) return DV_EUNSUPPORTED;
    a.in_scale = bn_in->scale; a.in_shift = bn_in->shift; a.in_C = d->Cin; a.in_relu = (bn_in->flags & DV_RELU) ? 1 : 0;
  }
  {
    const int64_t es = d->dtype == DV_F32 ? 4 : 2;
    const int64_t sb = ((int64_t)d->N * d->Ti * d->Hi * d->Wi - 1) * d->ldx * es + (int64_t)d->cin_pitch * es;
    const int64_t wb = w3 ? w3_bytes(d->Cout, a.g.Ktot) : (int64_t)d->Cout * a.g.Ktot * es;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || d->kt > 32 || d->kh > 32 || d->kw > 32 || d->kt * d->kh * d->kw > 256)
      return DV_EUNSUPPORTED;
    a.src_bytes = (int)sb; a.w_bytes = (int)wb;
    {
      const int64_t ob = (((int64_t)a.M - 1) * a.ldo + a.NP) * 4;       // fp32 outputs: direct-store epilogue
      a.out_bytes = (d->dtype == DV_F32 && ob < (1ll << 31)) ? (int)ob : 0;
    }
    a.fCP = make_fastdiv((uint32_t)a.g.CP);
  }
  const int gvb = gather_bytes(d->dtype, d->cin_pitch);
  if (gvb == 16 && !aligned16(x)) return DV_EALIGN;
  const int esz = d->dtype == DV_F32 ? 4 : 2;
  if ((d->ldx * esz) % gvb || (a.ldw * esz) % gvb) return DV_EALIGN;
  if (bn_path != 2) trim_dead_taps(a, MODE_FWD, d->dtype);
  if (d->dtype == DV_F32 && w3 && !bn_path && dvt_conv_pp_launch(&a, MODE_FWD, stream)) return dv_launch_status();
  if (d->dtype == DV_F32 && w3 && bn_path != 2 && dvt_conv_tap_launch(&a, MODE_FWD, stream)) return dv_launch_status();
  if (bn_path == 1) return DV_EUNSUPPORTED;          // (cannot happen: fwd_bn_in_path asked the same question)
  int bm, bn;
  pick_tile(d->dtype, a.M, a.NP, bm, bn);
  a.ntn = (a.NP + bn - 1) / bn;
  const int grid = a.ntn * ((a.M + bm - 1) / bm);
  hipStream_t s = (hipStream_t)stream;
  if (const int kbn = bn_path ? 0 : ks_tile(d->dtype, a, bm)) {
    launch_ks<MODE_FWD>(kbn, a, s);
    return dv_launch_status();
  }
  if (d->dtype == DV_F32) launch_gemm<float, MODE_FWD, 16>(bm, bn, a, grid, s);
  else if (gvb == 16) launch_gemm<bf16_t, MODE_FWD, 16>(bm, bn, a, grid, s);
  else launch_gemm<bf16_t, MODE_FWD, 8>(bm, bn, a, grid, s);
  return dv_launch_status();
}

extern "C" int dv_conv3d_fwd(const dv_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                             float* stats, void* stream) {
  return fwd_impl(d, x, w, bias, y, stats, nullptr, stream);
}

extern "C" int dv_conv3d_fwd_bn_in(const dv_conv_desc* d, const void* x_bn, const dv_bn_in* bn, const void* w, void* y,
                                   float* stats, void* stream) {
  if (!bn) return DV_EINVAL;
  return fwd_impl(d, x_bn, w, nullptr, y, stats, bn, stream);
}

// ---- fp8 pointwise GEMMs (BASELINE configs[4]: the 1x1x1 convs of resnet_2d3d.py's bottleneck blocks) --------------------
template <typename T, int MODE>
static void launch_gemm_fp8(int bm, int bn, const ConvArgs& a, int grid, hipStream_t s) {
  // one tap, channel pitch a multiple of the 64-element K tile: the uniform-tap gather (tap state in SGPRs)
  if (a.g.CP % 64 == 0) launch_gemm_ns<T, MODE, 16, 1, 2>(bm, bn, a, grid, s);
  else launch_gemm_ns<T, MODE, 16, 0, 2>(bm, bn, a, grid, s);
}

static int check_fp8_desc(const dv_conv_desc* d) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (d->dtype != DV_BF16) return DV_EUNSUPPORTED;                       // the bf16 side of the GEMM (output, other tensors)
  if (d->kt != 1 || d->kh != 1 || d->kw != 1 || d->st != 1 || d->sh != 1 || d->sw != 1 || d->pt || d->ph || d->pw)
    return DV_EUNSUPPORTED;                                              // pointwise only
  if (d->cin_pitch % 16 || d->cout_pitch % 16) return DV_EALIGN;          // 16-byte gathers of fp8 channels
  return DV_OK;
}

extern "C" int dv_conv3d_fwd_fp8(const dv_conv_desc* d, const void* x8, const void* w8, const float* scale_x,
                                 const float* scale_w, void* y, float* stats, void* stream) {
  int rc = check_fp8_desc(d);
  if (rc) return rc;
  if (!x8 || !w8 || !y || !scale_x || !scale_w) return DV_EINVAL;
  if ((d->flags & DV_STATS) && !stats) return DV_EINVAL;
  if (!aligned16(x8) || !aligned16(w8) || !aligned16(y) || d->ldx % 16 || (d->ldy * 2) % 16) return DV_EALIGN;
  ConvArgs a;
  fill_geom(d, MODE_FWD, a.g);
  a.src = x8; a.w = w8; a.out = y; a.bias = nullptr; a.stats = stats; a.sc_a = scale_x; a.sc_b = scale_w; a.bn_x = nullptr; a.bn_ws = nullptr; a.bn_bytes = 0;
  a.M = d->N * d->To * d->Ho * d->Wo;
  a.N = d->Cout; a.NP = d->cout_pitch;
  a.lds_ = d->ldx; a.ldo = d->ldy; a.ldw = a.g.Ktot;
  a.flags = d->flags & DV_STATS;
  a.cls_on = 0;
  const int64_t sb = ((int64_t)a.M - 1) * d->ldx + d->cin_pitch, wb = (int64_t)d->Cout * a.g.Ktot;
  if (sb >= (1ll << 31) || wb >= (1ll << 31)) return DV_EUNSUPPORTED;
  a.src_bytes = (int)sb; a.w_bytes = (int)wb;
  a.out_bytes = 0;
  a.fCP = make_fastdiv((uint32_t)a.g.CP);
  int bm, bn;
  pick_tile(DV_BF16, a.M, a.NP, bm, bn);
  a.ntn = (a.NP + bn - 1) / bn;
  launch_gemm_fp8<fp8e4_t, MODE_FWD>(bm, bn, a, a.ntn * ((a.M + bm - 1) / bm), (hipStream_t)stream);
  return dv_launch_status();
}

extern "C" int dv_conv3d_dgrad_fp8(const dv_conv_desc* d, const void* dy8, const void* wd8, const float* scale_dy,
                                   const float* scale_w, void* dx, void* stream) {
  int rc = check_fp8_desc(d);
  if (rc) return rc;
  if (!dy8 || !wd8 || !dx || !scale_dy || !scale_w) return DV_EINVAL;
  if (!aligned16(dy8) || !aligned16(wd8) || !aligned16(dx) || d->ldy % 16 || (d->ldx * 2) % 16) return DV_EALIGN;
  ConvArgs a;
  fill_geom(d, MODE_DGRAD, a.g);
  a.src = dy8; a.w = wd8; a.out = dx; a.bias = nullptr; a.stats = nullptr; a.sc_a = scale_dy; a.sc_b = scale_w; a.bn_x = nullptr; a.bn_ws = nullptr; a.bn_bytes = 0;
  a.M = d->N * d->Ti * d->Hi * d->Wi;
  a.N = d->Cin; a.NP = d->cin_pitch;
  a.lds_ = d->ldy; a.ldo = d->ldx; a.ldw = a.g.Ktot;
  a.flags = d->flags & DV_ACCUM;
  a.cls_on = 0;
  const int64_t sb = ((int64_t)a.M - 1) * d->ldy + d->cout_pitch, wb = (int64_t)d->Cin * a.g.Ktot;
  if (sb >= (1ll << 31) || wb >= (1ll << 31)) return DV_EUNSUPPORTED;
  a.src_bytes = (int)sb; a.w_bytes = (int)wb;
  a.out_bytes = 0;
  a.fCP = make_fastdiv((uint32_t)a.g.CP);
  int bm, bn;
  pick_tile(DV_BF16, a.M, a.NP, bm, bn);
  a.ntn = (a.NP + bn - 1) / bn;
  launch_gemm_fp8<fp8e5_t, MODE_DGRAD>(bm, bn, a, a.ntn * ((a.M + bm - 1) / bm), (hipStream_t)stream);
  return dv_launch_status();
}

// the launches of one data gradient (`a` complete): parity classes / LDS-staged kernel / K split over the waves / conv_gemm
static int dgrad_launch(const dv_conv_desc* d, ConvArgs& a, bool w3, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  a.cls_on = 0;
  const bool strided = d->st > 1 || d->sh > 1 || d->sw > 1;
  if (strided && d->kt >= d->st && d->kh >= d->sh && d->kw >= d->sw) {
    // one dense stride-1 launch per parity class of the input positions (see ConvArgs): every class has >= 1 tap
    ConvArgs cls[8];
    const int ncls = dgrad_classes(d, a, cls);
    if (w3) {
      // pre-split weights: every class on the LDS-staged input-tile kernel (conv_tap.hip, temporal form), or not at all
      for (int i = 0; i < ncls; ++i)
        if (!dvt_conv_tap_kind(&cls[i], MODE_DGRAD)) return DV_EUNSUPPORTED;
      for (int i = 0; i < ncls; ++i) dvt_conv_tap_launch(&cls[i], MODE_DGRAD, stream);
      return dv_launch_status();
    }
    if (a.bn_ws) return DV_EUNSUPPORTED;
    for (int i = 0; i < ncls; ++i) {
      ConvArgs& c = cls[i];
      int bm, bn;
      pick_tile(d->dtype, c.M, c.NP, bm, bn);
      c.ntn = (c.NP + bn - 1) / bn;
      const int grid = c.ntn * ((c.M + bm - 1) / bm);
      if (d->dtype == DV_F32) launch_gemm<float, MODE_DGRAD, 16>(bm, bn, c, grid, s);
      else launch_gemm<bf16_t, MODE_DGRAD, 16>(bm, bn, c, grid, s);
    }
    return dv_launch_status();
  }
  trim_dead_taps(a, MODE_DGRAD, d->dtype);
  if (d->dtype == DV_F32 && w3 && dvt_conv_tap_launch(&a, MODE_DGRAD, stream)) return dv_launch_status();
  if (a.bn_ws) return DV_EUNSUPPORTED;           // (the ordered fused reduce exists on the LDS-staged kernel only)
  int bm, bn;
  pick_tile(d->dtype, a.M, a.NP, bm, bn);
  a.ntn = (a.NP + bn - 1) / bn;
  const int grid = a.ntn * ((a.M + bm - 1) / bm);
  if (const int kbn = ks_tile(d->dtype, a, bm)) {
    launch_ks<MODE_DGRAD>(kbn, a, s);
    return dv_launch_status();
  }
  if (d->dtype == DV_F32) launch_gemm<float, MODE_DGRAD, 16>(bm, bn, a, grid, s);
  else launch_gemm<bf16_t, MODE_DGRAD, 16>(bm, bn, a, grid, s);
  return dv_launch_status();
}

static int dgrad_impl(const dv_conv_desc* d, const void* dy, const void* wd, void* dx, const dv_bn_reduce* bnr, void* stream,
                      void* bn_ws = nullptr, int64_t bn_ws_bytes = 0) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!dy || !wd || !dx) return DV_EINVAL;
  if (bnr) {
    if (!bnr->x || !bnr->mean || !bnr->invstd || !bnr->sums || bnr->n_rep <= 0 || (d->flags & DV_ACCUM)) return DV_EINVAL;
    if (!(bnr->flags & DV_NO_RELU_MASK) && (!bnr->scale || !bnr->shift)) return DV_EINVAL;
    if (!aligned16(bnr->x) || bnr->ldx < d->cin_pitch || (bnr->ldx * (d->dtype == DV_F32 ? 4 : 2)) % 16) return DV_EALIGN;
  }
  if (d->st > 2 || d->sh > 2 || d->sw > 2) return DV_EUNSUPPORTED;
  if (d->cin_pitch % 8) return DV_EUNSUPPORTED;      // the RGB input never needs a data gradient
  if (!aligned16(dy) || !aligned16(wd) || !aligned16(dx)) return DV_EALIGN;
  const int esz = d->dtype == DV_F32 ? 4 : 2;
  if ((d->ldx * esz) % 16) return DV_EALIGN;
  ConvArgs a;
  fill_geom(d, MODE_DGRAD, a.g);
  a.src = dy; a.w = wd; a.out = dx; a.bias = nullptr; a.stats = nullptr; a.sc_a = a.sc_b = nullptr;
  a.M = d->N * d->Ti * d->Hi * d->Wi;
  a.N = d->Cin; a.NP = d->cin_pitch;
  a.lds_ = d->ldy; a.ldo = d->ldx; a.ldw = a.g.Ktot;
  a.flags = d->flags & DV_ACCUM;
  a.bn_x = nullptr; a.bn_ws = nullptr; a.bn_bytes = 0;
  if (bnr) {
    a.bn_x = bnr->x; a.bn_ldx = bnr->ldx; a.bn_mean = bnr->mean; a.bn_invstd = bnr->invstd; a.bn_scale = bnr->scale;
    a.bn_shift = bnr->shift; a.bn_sums = bnr->sums; a.bn_rep = bnr->n_rep; a.bn_mask = (bnr->flags & DV_NO_RELU_MASK) ? 0 : 1;
    if (bn_ws) {                                       // dv_conv3d_dgrad_bn_ws: the ordered form, on the LDS-staged kernel only
      const int64_t need = dvt_bn_ws_floats(a.M, a.NP) * 4;
      const int64_t xb = (((int64_t)a.M - 1) * bnr->ldx + a.NP) * 4;
      if (need <= 0) return DV_EUNSUPPORTED;             // (more tiles than the ticket region of the workspace holds)
      if (bn_ws_bytes < need || !aligned16(bn_ws) || xb >= (1ll << 31) || d->dtype != DV_F32) return DV_EINVAL;
      a.bn_ws = reinterpret_cast<float*>(bn_ws);
      a.bn_bytes = (int)xb;
    }
  }
  const bool w3 = (d->flags & DV_W3) != 0;
  if (w3 && (d->dtype != DV_F32 || f32_exact())) return DV_EUNSUPPORTED;      // (strided: only where every parity class runs on the
                                                                              //  LDS-staged kernel, checked at the class launches)
  if (w3) { a.flags |= DV_W3; a.ldw = w3_rows(d->Cin); }
  {
    const int64_t es = d->dtype == DV_F32 ? 4 : 2;
    const int64_t sb = ((int64_t)d->N * d->To * d->Ho * d->Wo - 1) * d->ldy * es + (int64_t)d->cout_pitch * es;
    const int64_t wb = w3 ? w3_bytes(d->Cin, a.g.Ktot) : (int64_t)d->Cin * a.g.Ktot * es;
    if (sb >= (1ll << 31) || wb >= (1ll << 31) || d->kt > 32 || d->kh > 32 || d->kw > 32 || d->kt * d->kh * d->kw > 256)
      return DV_EUNSUPPORTED;
    a.src_bytes = (int)sb; a.w_bytes = (int)wb;
    {
      const int64_t ob = (((int64_t)a.M - 1) * a.ldo + a.NP) * 4;       // fp32 outputs: direct-store epilogue
      a.out_bytes = (d->dtype == DV_F32 && ob < (1ll << 31)) ? (int)ob : 0;
    }
    a.fCP = make_fastdiv((uint32_t)a.g.CP);
  }
  if (bnr && !bn_ws) {
    // dv_conv3d_dgrad_bn, the atomic form: the plain data gradient on whatever kernel takes it, then the reduce over what it wrote
    ConvArgs tail = a;
    a.bn_x = nullptr;
    const int rc2 = dgrad_launch(d, a, w3, stream);
    if (rc2) return rc2;
    tail.out_bytes = a.out_bytes;
    dvx_dgrad_bn_tail_launch(&tail, d->dtype == DV_F32 ? 1 : 0, stream);
    return dv_launch_status();
  }
  return dgrad_launch(d, a, w3, stream);
}

extern "C" int dv_conv3d_dgrad(const dv_conv_desc* d, const void* dy, const void* wd, void* dx, void* stream) {
  return dgrad_impl(d, dy, wd, dx, nullptr, stream);
}

extern "C" int dv_conv3d_dgrad_bn(const dv_conv_desc* d, const void* dy, const void* wd, void* dx, const dv_bn_reduce* bn,
                                  void* stream) {
  if (!bn) return DV_EINVAL;
  return dgrad_impl(d, dy, wd, dx, bn, stream);
}

extern "C" int64_t dv_conv3d_dgrad_bn_workspace(const dv_conv_desc* d) {
  if (!d || check_desc(d) || (d->flags & DV_ACCUM) || !tap_choice(d, 1)) return 0;
  const int64_t rows = (int64_t)d->N * d->Ti * d->Hi * d->Wi;
  return dvt_bn_ws_floats(rows, d->cin_pitch) * 4;
}

extern "C" int dv_conv3d_dgrad_bn_ws(const dv_conv_desc* d, const void* dy, const void* wd, void* dx, const dv_bn_reduce* bn,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  if (!bn || !workspace) return DV_EINVAL;
  return dgrad_impl(d, dy, wd, dx, bn, stream, workspace, workspace_bytes);
}

// Tile / row-split plan of a weight-gradient problem (shared by the launch and by dv_conv3d_wgrad_workspace).
struct WgradCfg { int BI, BJ, WVI, WVJ, NS; };
// DMA-kernel configurations.  The kernel is bound by what a CU can pull from L2 into LDS (~70 GB/s per CU): the bytes
// filled per unit of work go with 1/BI + 1/BJ, so the largest tile that wastes little padding wins -- see plan_wgrad.
static const WgradCfg kWgBf16[] = {{128, 128, 2, 2, 3}, {64, 256, 1, 4, 3}, {128, 256, 2, 2, 3}, {192, 256, 3, 4, 4}};
// (f32 split mode: a two-wave 64 x 128 workgroup with 64 x 64 wave tiles -- one fragment split per 32x32 block instead of
// 1.5 -- was measured slower than the four-wave one with 32 x 64 wave tiles: 950 vs 827 us on the 7x1x1 stem layer)
static const WgradCfg kWgF32[] = {{128, 128, 2, 2, 2}, {64, 128, 2, 2, 2}};
// conv_wgrad_f32s_kernel (the fp32 split mode's second weight-gradient form) lives in conv_experiments.hip: opt-in only
int dvx_wgrad_f32s_lds_bytes(int cfg);
void dvx_wgrad_f32s_tile(int cfg, int* bi, int* bj);
void dvx_launch_wgrad_f32s(int cfg, const void* args, int grid, void* stream);

struct WgradPlan {
  int BI, BJ, nti, ntj, splits, rows_per_split, gvb, cfg;
  int f32s;                   // >= 0: conv_wgrad_f32s_kernel configuration (conv_experiments.hip)
  bool dma;
  bool tm;                    // conv_wgrad_tm_kernel (conv_tap_wgrad.hip): stride-1 3x1x1, fp32 split mode, T = 2 / 4
  int nchunks, chunks_per_split;
  long long slab_stride;      // elements
};

// the LDS-staged weight gradient of the stride-1 3x1x1 convs (conv_tap_wgrad.hip); the argument is a TmWgradArgs*
void dvw_wgrad_tm_launch(const void* args, int grid, void* stream);

// 0: no; 1: stride 1, 3x1x1, T = 2 / 4; 2: the 7x1x1 / stride-2 stem conv (8 -> 4 frames); 3: stride 1, 1x3x3, padding 1
static int wgrad_tm_kind(const dv_conv_desc* d) {
  static const int on = env_int("DUALVAR_WGRAD_TM", 1);           // (A/B switch; 2: temporal forms only)
  if (!on || d->dtype != DV_F32 || f32_exact()) return 0;
  if (d->cin_pitch % 8 || d->cout_pitch % 8) return 0;
  // 4: the pixel-pair RGB stem conv: window 1 x 7 x 4 over 8-channel pixel pairs, stride (1, 2, 1), no padding
  if (on != 2 && d->kt == 1 && d->kh == 7 && d->kw == 4 && d->st == 1 && d->sh == 2 && d->sw == 1 && !d->pt && !d->ph && !d->pw &&
      d->cin_pitch == 8 && d->ldx == 8 && d->Wo <= 64 && d->Wi <= 68) {
    const int64_t Mx = (int64_t)d->N * d->Ti * d->Hi * d->Wi, M = (int64_t)d->N * d->To * d->Ho * d->Wo;
    static const int min_rows4 = env_int("DUALVAR_CONV_TAP_GRID", 128) <= 1 ? 1 : 8192;
    if (Mx * 32 < (1ll << 31) && (M - 1) * d->ldy * 4 + (int64_t)d->cout_pitch * 4 < (1ll << 31) && M >= min_rows4) return 4;
    return 0;
  }
  if (d->sh != 1 || d->sw != 1) return 0;
  int kind = 0;
  if (d->kh == 1 && d->kw == 1 && !d->ph && !d->pw) {
    if (d->kt == 3 && d->st == 1 && d->pt == 1 && d->To == d->Ti && (d->Ti == 2 || d->Ti == 4)) kind = 1;
    else if (d->kt == 7 && d->st == 2 && d->pt == 3 && d->Ti == 8 && d->To == 4) kind = 2;
  } else if (on != 2 && d->kt == 1 && d->st == 1 && !d->pt && d->kh == 3 && d->kw == 3 && d->ph == 1 && d->pw == 1 &&
             d->Hi >= 2 && d->Wi >= 2) {
    // the spatial form pays on the large maps whose channel count fills its 64-channel x tiles: Conv_2c 710 -> 542 us, Mixed_3c
    // (128 channels) 310 -> 240; Mixed_3b (96 channels: a half-empty second tile) 148 -> 156 and the 12 544-row levels 89 -> 87
    // stay on conv_wgrad_dma_kernel (isolated, one box)
    const int64_t rows = (int64_t)d->N * d->Ti * d->Hi * d->Wi;
    static const int any_size = env_int("DUALVAR_CONV_TAP_GRID", 128) <= 1;
    if ((any_size && d->cin_pitch >= 16) || (rows >= 50000 && (d->cin_pitch + 63) / 64 * 64 * 10 <= d->cin_pitch * 11)) kind = 3;
  }
  if (!kind) return 0;
  const int64_t Mx = (int64_t)d->N * d->Ti * d->Hi * d->Wi, M = (int64_t)d->N * d->To * d->Ho * d->Wo;
  const int64_t xb = (Mx - 1) * d->ldx * 4 + (int64_t)d->cin_pitch * 4, yb = (M - 1) * d->ldy * 4 + (int64_t)d->cout_pitch * 4;
  if (xb >= (1ll << 31) || yb >= (1ll << 31)) return 0;
  // (DUALVAR_CONV_TAP_GRID <= 1, the test knob of the LDS-staged kernels, also lets tiny problems through: tools/tap_check.py)
  static const int min_rows = env_int("DUALVAR_CONV_TAP_GRID", 128) <= 1 ? 1 : 8192;
  return M >= min_rows ? kind : 0;
}


static WgradPlan plan_wgrad(const dv_conv_desc* d) {
  WgradPlan p;
  const int M = d->N * d->To * d->Ho * d->Wo;
  const int J = d->kt * d->kh * d->kw * d->cin_pitch;
  const int es = d->dtype == DV_F32 ? 4 : 2;
  p.gvb = gather_bytes(d->dtype, d->cin_pitch);
  p.dma = p.gvb == 16 && d->kt <= 8 && d->kh <= 8 && d->kw <= 8;
  if (p.dma) {
    const int64_t xb = ((int64_t)d->N * d->Ti * d->Hi * d->Wi - 1) * d->ldx * es + (int64_t)d->cin_pitch * es;
    const int64_t yb = ((int64_t)M - 1) * d->ldy * es + (int64_t)d->cout_pitch * es;
    if (xb >= (1ll << 31) || yb >= (1ll << 31)) p.dma = false;
  }
  int per_cu = 3;
  p.cfg = -1;
  p.f32s = -1;
  p.tm = false; p.nchunks = p.chunks_per_split = 0;
  if (const int tmk = wgrad_tm_kind(d)) {
    // 64 x 64 (n, c) tiles carrying all three taps; the row splits are ranges of 64 / T-pixel steps: one round of three
    // co-resident workgroups per CU, at least 8 steps each, and not more than the slab traffic pays for (as below)
    p.tm = true; p.dma = false;
    const int bc = tmk == 2 ? 32 : 64;                            // channel tile of x (conv_tap_wgrad.hip)
    p.BI = 64; p.BJ = (tmk == 3 ? 3 : d->kt) * bc;
    p.nti = (d->Cout + 63) / 64; p.ntj = (d->cin_pitch + bc - 1) / bc;
    const int krows = tmk == 3 ? 3 : 1;                           // spatial: one workgroup per kernel row dh
    if (tmk == 4) {
      p.BJ = 224; p.ntj = 1;
      p.nchunks = d->N * d->To * d->Ho;                           // output lines
    } else if (tmk == 3) {
      p.nchunks = (M + 63) / 64;                                  // 64-row steps
    } else {
      const int pxs = 64 / d->To;
      p.nchunks = (int)(((int64_t)d->N * d->Hi * d->Wi + pxs - 1) / pxs);
    }
    int splits = std::max(1, ((tmk == 2 || tmk == 4) ? 512 : 768) / (p.nti * p.ntj * krows));      // (the stem forms: two workgroups per CU)
    splits = std::min(splits, std::max(1, p.nchunks / 8));
    const double slab_us = 8.0 * d->Cout * (double)J / 4e6;
    const int s_opt = (int)(std::sqrt(p.nchunks * 1.2 / slab_us) + 0.5);          // ~1.2 us per step of one workgroup
    splits = std::max(1, std::min(splits, s_opt));
    p.chunks_per_split = (p.nchunks + splits - 1) / splits;
    p.splits = (p.nchunks + p.chunks_per_split - 1) / p.chunks_per_split;
    p.rows_per_split = p.chunks_per_split * 64;
    p.slab_stride = ((long long)d->Cout * J + 63) / 64 * 64;
    return p;
  }
  // DUALVAR_WGRAD_F32S = 0 | 2: conv_wgrad_f32s_kernel (conv_experiments.hip) in its 64 x 256 / 128 x 128 form for every fp32
  // weight gradient (sweeps: tools/wgrad_sweep.sh).  Not selected by default: isolated it gains 8 - 10 % on the layers whose
  // shape its tiles fit (128-channel layers on 128 x 128, the 7x1x1 stem conv on 64 x 256) and nothing elsewhere, and a
  // per-layer rule built on that measured 19.39 vs 19.39 ms on the whole step ("Round 3").
  static const int f32s_force = env_int("DUALVAR_WGRAD_F32S", -1);
  const int f32s_pick = (f32s_force == 0 || f32s_force == 2) ? f32s_force : -1;
  if (p.dma && d->dtype == DV_F32 && !f32_exact() && f32s_pick >= 0) {
    p.f32s = f32s_pick;
    dvx_wgrad_f32s_tile(f32s_pick, &p.BI, &p.BJ);
    per_cu = std::max(1, std::min(2, 163840 / dvx_wgrad_f32s_lds_bytes(f32s_pick)));
  } else if (p.dma) {
    // candidate with the least estimated time: LDS-fill bytes at ~18 TB/s chip-wide against padded MFMA work at ~2/3 of
    // peak; a configuration whose tiles cannot even fill the chip once (few rows) pays for the idle CUs
    const WgradCfg* tab = d->dtype == DV_F32 ? kWgF32 : kWgBf16;
    const int ncfg = d->dtype == DV_F32 ? 2 : 4;
    constexpr int minrows_ = 256;
    double best = 0;
    for (int c = 0; c < ncfg; ++c) {
      const WgradCfg& k = tab[c];
      const int nti = (d->Cout + k.BI - 1) / k.BI, ntj = (J + k.BJ - 1) / k.BJ;
      const double fill = (double)M * nti * ntj * (k.BI + k.BJ) * es / 18e12;
      const double mfma = 2.0 * M * (double)(nti * k.BI) * (ntj * k.BJ) / (d->dtype == DV_F32 ? 120e12 : 1600e12);
      const int cap = 256 * wgrad_wgs_per_cu(es, k.BI, k.BJ, k.WVI * k.WVJ, k.NS);
      const int64_t sp = std::max<int64_t>(1, std::min<int64_t>(cap / (nti * ntj), (M + minrows_ - 1) / minrows_));
      const double wgs = (double)nti * ntj * sp;
      const double t = std::max(fill, mfma) * std::max(1.0, 256.0 / wgs);
      if (p.cfg < 0 || t < best) { best = t; p.cfg = c; }
    }
    const WgradCfg& k = tab[p.cfg];
    p.BI = k.BI; p.BJ = k.BJ;
    per_cu = wgrad_wgs_per_cu(es, k.BI, k.BJ, k.WVI * k.WVJ, k.NS);
  } else {
    // tile heights 64 or 128: whichever pads Cout less (144 -> 3 x 64 rather than 2 x 128)
    const bool narrow = (d->Cout + 63) / 64 * 64 < (d->Cout + 127) / 128 * 128;
    p.BI = narrow ? 64 : 128;
    p.BJ = narrow ? 128 : 64;
  }
  p.nti = (d->Cout + p.BI - 1) / p.BI;
  p.ntj = (J + p.BJ - 1) / p.BJ;
  const int tiles = p.nti * p.ntj;
  // Row splits: ONE round of co-resident workgroups (256 CUs x the workgroups per CU the kernel's LDS / registers admit)
  // -- a grid of 1.5 rounds leaves half the chip idle for the second one -- but at least `minrows` rows each: a split's
  // partial tile costs as much traffic as ~128 rows of its inputs.
  constexpr int minrows = 256;
  const int tgt = 256 * per_cu;
  int splits = tgt / tiles;                          // round down: never more workgroups than fit at once
  const int max_splits = (M + minrows - 1) / minrows;
  if (splits > max_splits) splits = max_splits;
  // ... and not more than pays: every split writes (and the reduce re-reads) a whole partial dW.  With S splits a workgroup
  // runs M/(32 S) steps of ~t_step and the slabs cost 2 S |W| 4 bytes at ~4 TB/s: the sum is least at
  // S = sqrt(M/32 * t_step / (8 |W| / 4e6 us)).  (Layers with few rows and a large dW -- Mixed_4/5 -- had slab traffic of
  // 3x their operands.)
  {
    const double t_step = d->dtype == DV_F32 ? 1.5 : 0.7;                        // us per 32-row step of one workgroup
    const double slab_us = 8.0 * d->Cout * (double)J / 4e6;
    const int s_opt = (int)(std::sqrt(M / 32.0 * t_step / slab_us) + 0.5);
    if (splits > s_opt) splits = s_opt;
  }
  if (splits < 1) splits = 1;
  p.rows_per_split = ((M + splits - 1) / splits + 31) / 32 * 32;
  p.splits = (M + p.rows_per_split - 1) / p.rows_per_split;
  p.slab_stride = ((long long)d->Cout * J + 63) / 64 * 64;
  return p;
}

extern "C" int dv_conv3d_wgrad_tile(const dv_conv_desc* d, int32_t* rows, int32_t* cols, int32_t* splits) {
  if (!d || !rows || !cols || !splits) return DV_EINVAL;
  int rc = check_desc(d);
  if (rc) return rc;
  const WgradPlan p = plan_wgrad(d);
  *rows = p.BI; *cols = p.BJ; *splits = p.splits;
  return DV_OK;
}

extern "C" int64_t dv_conv3d_wgrad_workspace(const dv_conv_desc* d) {
  if (check_desc(d)) return 0;
  const WgradPlan p = plan_wgrad(d);
  return p.splits > 1 ? (int64_t)p.splits * p.slab_stride * 4 : 0;
}

// the plan runs on conv_wgrad_dma_kernel<float, 64, 128, 1, 4, 2, true, true>, the form that can carry dv_conv3d_wgrad_bn
static bool wgrad_plan_is_share(const dv_conv_desc* d, const WgradPlan& p) {
  if (p.tm) return wgrad_tm_kind(d) == 4;        // conv_wgrad_pp_kernel<BNA> (conv_tap_wgrad.hip)
  return p.dma && d->dtype == DV_F32 && !f32_exact() && p.f32s < 0 && p.cfg == 1;
}
extern "C" int dv_conv3d_wgrad_bn_ok(const dv_conv_desc* d) {
  if (!d || check_desc(d)) return 0;
  return wgrad_plan_is_share(d, plan_wgrad(d)) ? 1 : 0;
}

static int wgrad_impl(const dv_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace,
                      int64_t workspace_bytes, const dv_bn_bwd* bn, void* stream, const dv_bn_in* bn_in = nullptr) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!x || !dy || !dw) return DV_EINVAL;
  if (!aligned16(dy) || !aligned16(dw) || (reinterpret_cast<uintptr_t>(x) & 7)) return DV_EALIGN;
  const WgradPlan p = plan_wgrad(d);
  if (bn_in) {
    if (bn || !bn_in->scale || !bn_in->shift) return DV_EINVAL;
    const int k = p.tm ? wgrad_tm_kind(d) : 0;
    if (k != 1 && k != 2) return DV_EUNSUPPORTED;
  }
  if (bn) {
    if (!bn->x || !bn->mean || !bn->invstd || !bn->gamma || !bn->sums || bn->n_rep <= 0) return DV_EINVAL;
    if ((bn->dgamma == nullptr) != (bn->dbeta == nullptr)) return DV_EINVAL;
    if (!(bn->flags & DV_NO_RELU_MASK) && (!bn->scale || !bn->shift)) return DV_EINVAL;
    if (bn->ldx != d->ldy) return DV_EINVAL;
    if (!aligned16(bn->x)) return DV_EALIGN;
    if (!wgrad_plan_is_share(d, p)) return DV_EUNSUPPORTED;
  }
  const int64_t need = p.splits > 1 ? (int64_t)p.splits * p.slab_stride * 4 : 0;
  if (need && (!workspace || workspace_bytes < need)) return DV_EINVAL;
  if (need && !aligned16(workspace)) return DV_EALIGN;
  if (p.gvb == 16 && !aligned16(x)) return DV_EALIGN;
  WgradArgs a;
  fill_geom(d, MODE_FWD, a.g);
  a.x = x; a.dy = dy; a.dw = dw;
  a.slab = need ? reinterpret_cast<float*>(workspace) : nullptr;
  a.slab_stride = p.slab_stride;
  a.M = d->N * d->To * d->Ho * d->Wo;
  a.Cout = d->Cout; a.CoutP = d->cout_pitch; a.J = a.g.Ktot;
  a.ldx = d->ldx; a.ldy = d->ldy; a.ldw = a.g.Ktot;
  a.nti = p.nti; a.ntj = p.ntj;
  a.rows_per_split = p.rows_per_split;
  a.perm = make_row_perm(d->kt > 1 && d->To > 1, d->To, d->Ho * d->Wo);
  const int grid = p.nti * p.ntj * p.splits;
  const bool narrow = p.BI == 64;
  hipStream_t s = (hipStream_t)stream;
  if (p.tm) {
    TmWgradArgs t;
    t.x = x; t.dy = dy; t.dw = dw; t.slab = a.slab; t.slab_stride = p.slab_stride;
    t.S = d->Hi * d->Wi; t.NQ = d->N * t.S; t.T = d->To; t.kind = wgrad_tm_kind(d);
    t.Cout = d->Cout; t.CoutP = d->cout_pitch; t.CP = d->cin_pitch;
    t.ldx = d->ldx; t.ldy = d->ldy; t.ldw = a.ldw;
    t.nti = p.nti; t.ntc = p.ntj; t.nchunks = p.nchunks; t.chunks_per_split = p.chunks_per_split;
    t.x_bytes = (int)(((int64_t)d->N * d->Ti * d->Hi * d->Wi - 1) * d->ldx * 4 + (int64_t)d->cin_pitch * 4);
    t.dy_bytes = (int)(((int64_t)a.M - 1) * d->ldy * 4 + (int64_t)d->cout_pitch * 4);
    t.fS = make_fastdiv((uint32_t)t.S);
    t.M = a.M; t.H = d->Hi; t.W = d->Wi;
    t.fW = make_fastdiv((uint32_t)d->Wi); t.fH = make_fastdiv((uint32_t)d->Hi);
    t.Wo = d->Wo; t.Ho = d->Ho; t.Wp = d->Wi; t.Hp = d->Hi;
    t.bn_x = nullptr;
    if (bn_in) { t.in_scale = bn_in->scale; t.in_shift = bn_in->shift; t.in_C = d->Cin; t.in_relu = (bn_in->flags & DV_RELU) ? 1 : 0; }
    if (t.kind == 4) {
      t.fH = make_fastdiv((uint32_t)d->Ho);                      // line -> (image, output line)
      if (bn) {
        t.bn_x = bn->x; t.bn_mean = bn->mean; t.bn_invstd = bn->invstd; t.bn_gamma = bn->gamma; t.bn_scale = bn->scale;
        t.bn_shift = bn->shift; t.bn_sums = bn->sums; t.bn_dgamma = bn->dgamma; t.bn_dbeta = bn->dbeta;
        t.bn_inv_count = bn->inv_count; t.bn_dscale = bn->dparam_scale; t.bn_rep = bn->n_rep;
        t.bn_mask = (bn->flags & DV_NO_RELU_MASK) ? 0 : 1;
      }
    }
    dvw_wgrad_tm_launch(&t, t.kind == 3 ? grid * 3 : grid, stream);
  } else if (p.dma) {
    const int64_t es = d->dtype == DV_F32 ? 4 : 2;
    WgradDmaArgs aa;
    aa.w = a;
    aa.x_bytes = (int)(((int64_t)d->N * d->Ti * d->Hi * d->Wi - 1) * d->ldx * es + (int64_t)d->cin_pitch * es);
    aa.dy_bytes = (int)(((int64_t)a.M - 1) * d->ldy * es + (int64_t)d->cout_pitch * es);
    aa.bn_x = nullptr;
    if (bn) {
      aa.bn_x = bn->x; aa.bn_mean = bn->mean; aa.bn_invstd = bn->invstd; aa.bn_gamma = bn->gamma; aa.bn_scale = bn->scale;
      aa.bn_shift = bn->shift; aa.bn_sums = bn->sums; aa.bn_dgamma = bn->dgamma; aa.bn_dbeta = bn->dbeta;
      aa.bn_inv_count = bn->inv_count; aa.bn_dscale = bn->dparam_scale; aa.bn_rep = bn->n_rep;
      aa.bn_mask = (bn->flags & DV_NO_RELU_MASK) ? 0 : 1;
    }
#define WGD(T_, BI_, BJ_, WI_, WJ_, NS_, ...) \
  hipLaunchKernelGGL((conv_wgrad_dma_kernel<T_, BI_, BJ_, WI_, WJ_, NS_, ##__VA_ARGS__>), dim3(grid), dim3(WI_ * WJ_ * 64), 0, s, aa)
    if (p.f32s >= 0) {
      dvx_launch_wgrad_f32s(p.f32s, &aa, grid, s);
    } else if (d->dtype == DV_BF16) {
      switch (p.cfg) {
        case 0: WGD(bf16_t, 128, 128, 2, 2, 3); break;
        case 1: WGD(bf16_t, 64, 256, 1, 4, 3); break;
        case 2: WGD(bf16_t, 128, 256, 2, 2, 3); break;
        default: WGD(bf16_t, 192, 256, 3, 4, 4); break;
      }
    } else {
      if (f32_exact()) {
        if (p.cfg == 1) WGD(float, 64, 128, 2, 2, 2);
        else WGD(float, 128, 128, 2, 2, 2);
      } else {
        if (bn) WGD(float, 64, 128, 1, 4, 2, true, true, true);
        else if (p.cfg == 1) WGD(float, 64, 128, 1, 4, 2, true, true);
        else WGD(float, 128, 128, 2, 2, 2, true);
      }
    }
#undef WGD
  } else {
#define WG_LAUNCH(T_, G_)                                                                                   \
  do {                                                                                                      \
    if (narrow) hipLaunchKernelGGL((conv_wgrad_kernel<T_, G_, 64, 128>), dim3(grid), dim3(256), 0, s, a);   \
    else hipLaunchKernelGGL((conv_wgrad_kernel<T_, G_, 128, 64>), dim3(grid), dim3(256), 0, s, a);          \
  } while (0)
    if (d->dtype == DV_F32) WG_LAUNCH(float, 16);
    else if (p.gvb == 16) WG_LAUNCH(bf16_t, 16);
    else WG_LAUNCH(bf16_t, 8);
#undef WG_LAUNCH
  }
  // (folding this reduce into the weight-gradient kernel -- VERDICT round 3 item 6 -- was priced first: with EVERY reduce launch
  // skipped the step went 17.69 -> 17.66 ms, two A/B pairs on one box; the launches overlap on the side stream.  Not built.)
  if (need) {
    const long long n4 = (long long)d->Cout * a.ldw / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, s, a.slab, p.slab_stride,
                       p.splits, dw, n4);
  }
  return dv_launch_status();
}

extern "C" int dv_conv3d_wgrad(const dv_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  return wgrad_impl(d, x, dy, dw, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int dv_conv3d_wgrad_bn(const dv_conv_desc* d, const void* x, const void* g, float* dw, void* workspace,
                                  int64_t workspace_bytes, const dv_bn_bwd* bn, void* stream) {
  if (!bn) return DV_EINVAL;
  return wgrad_impl(d, x, g, dw, workspace, workspace_bytes, bn, stream);
}

extern "C" int dv_conv3d_wgrad_bn_in(const dv_conv_desc* d, const void* x_bn, const dv_bn_in* bn, const void* dy, float* dw,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
  if (!bn) return DV_EINVAL;
  return wgrad_impl(d, x_bn, dy, dw, workspace, workspace_bytes, nullptr, stream, bn);
}

extern "C" int dv_conv3d_bn_in_ok(const dv_conv_desc* d) {
  if (!d || check_desc(d)) return 0;
  const int k = wgrad_tm_kind(d);
  if (k != 1 && k != 2) return 0;
  return fwd_bn_in_path(d);
}
