// Device-side pieces shared by the convolution translation units (conv.hip: the hot path; conv_tap.hip: the LDS-staged
// input-tile kernels for the separable convs; conv_experiments.hip: measured-and-not-selected kernels kept under test).
// Everything sits in an anonymous namespace: each translation unit gets its own copy, nothing is exported.
#pragma once
#include "common.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>


namespace {


enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct ConvGeom {
  // "row space" (what m enumerates) and "source space" (the tensor the gather reads)
  int rT, rH, rW;        // row-space dims
  int sT, sH, sW;        // source dims
  int kt, kh, kw;
  int st, sh, sw;        // strides (conv strides)
  int pt, ph, pw;
  int CP;                // channel pitch used to decode k -> (tap, c)
  int Ktot;              // kt*kh*kw*CP
  FastDiv dW, dH, dT;    // fast division by rW, rH, rT
};

// Row ORDER of a launch whose window has taps along t (on): rows are enumerated
//   for clip n: for 32-pixel chunk c of the (h, w) plane: for t: rows (n, t, c*32 .. c*32+31)
// instead of the tensor's [n][t][h][w] order, so the rows a tap reads at t-1 / t+1 are rows the neighbouring steps of the
// weight-gradient kernel read at their own t: L2 hits.  In tensor order they are a whole plane (784 .. 3 136 rows)
// away and every tap fetched its own copy of the activation: PMC 2.2x the algorithmic bytes on the 3x1x1 / 7x1x1 weight
// gradients (1.0 - 1.25x with this order, and 3 - 9 % faster).  A sum over rows: any fixed bijection of [0, M) will do.
// (The same order for the tiles of the forward / data-gradient GEMMs was measured 2 - 6 % SLOWER: not used there.)
struct RowPerm {
  int on, pT, pS, nfull, full_span;
  FastDiv fB, f32T, fwt;
};
static RowPerm make_row_perm(bool on, int T, int S) {
  RowPerm p;
  p.on = on ? 1 : 0; p.pT = T; p.pS = S; p.nfull = S / 32; p.full_span = p.nfull * 32 * T;
  p.fB = make_fastdiv((uint32_t)(T * S)); p.f32T = make_fastdiv((uint32_t)(32 * T));
  p.fwt = make_fastdiv((uint32_t)(S % 32 ? S % 32 : 1));
  return p;
}
__device__ __forceinline__ uint32_t perm_row(const RowPerm& a, uint32_t q) {
  uint32_t n, p, t, pix;
  fd_divmod(q, a.fB, n, p);
  if (p < (uint32_t)a.full_span) {
    uint32_t c, w;
    fd_divmod(p, a.f32T, c, w);
    t = w >> 5; pix = c * 32 + (w & 31);
  } else {
    uint32_t r;
    fd_divmod(p - (uint32_t)a.full_span, a.fwt, t, r);
    pix = (uint32_t)a.nfull * 32 + r;
  }
  return (n * (uint32_t)a.pT + t) * (uint32_t)a.pS + pix;
}

struct ConvArgs {
  const void* src;       // gathered tensor (x for fwd, dy for dgrad)
  const void* w;         // [N rows][Ktot] K-contiguous, pitch ldw
  void* out;             // [M][ldo]
  const float* bias;
  float* stats;          // [2][N][m_tiles]
  int M, N, NP;          // rows, real cols, cols to write (zeros beyond N)
  int lds_, ldo, ldw;    // pitches in elements (src, out, weights)
  int ntn;               // number of N tiles
  int flags;
  const float* sc_a;     // fp8 GEMMs: device scalars, result = acc * sc_a[0] * sc_b[0] (per-tensor scales of the operands)
  const float* sc_b;
  int src_bytes, w_bytes; // extents for the buffer descriptors (< 2 GiB)
  int out_bytes;          // extent of the output rows [0, M) from `out` (0: not known to be < 2 GiB -> staged epilogue)
  FastDiv fCP;           // k -> (tap, c)
  ConvGeom g;
  // Strided dgrad, one launch per PARITY CLASS (cls_on): input positions t = t'*cst + cot (same for h, w) only receive
  // the kernel taps d = crt + cst*j, so in (t', j) coordinates the class is a dense stride-1 problem -- `g` describes
  // it (row space = the class's sub-lattice, kernel = its taps, padding (cot + pt - crt) / cst) -- instead of gathering
  // all taps and multiplying zeros for the (st*sh*sw - 1)/(st*sh*sw) that miss.  The weights are not repacked: a class tap
  // reads the ORIGINAL tap ((crt+cst*jt)*oKH + crh+csh*jh)*oKW + crw+csw*jw; output rows map back to the full input.
  // cls_on == 2: only the TAPS are remapped (cst = csh = csw = 1, crt / crh / crw = first live tap): a window whose outer taps
  // fall into the padding for EVERY row (3x1x1 with padding 1 on a one-frame map: two of three taps) runs as the smaller
  // window of its live taps -- trim_dead_taps() on the host.
  int cls_on, cst, csh, csw, cot, coh, cow, crt, crh, crw, oKH, oKW, oT, oH, oW;
  // dgrad whose output is dL/dy of y = relu(x_bn * scale + shift), the BatchNorm in front of this conv, and is that
  // gradient's only contribution (dv_conv3d_dgrad_bn): the epilogue also accumulates the BatchNorm backward's
  // sum(g), sum(g * xhat) into bn_sums[tile_m % bn_rep][2][CP] -- what dv_bn_bwd_reduce would compute from a second read of
  // dL/dy.  bn_x == nullptr: plain dgrad.
  const void* bn_x;
  const float *bn_mean, *bn_invstd, *bn_scale, *bn_shift;
  float* bn_sums;
  int bn_ldx, bn_rep, bn_mask;
  // dv_conv3d_dgrad_bn_ws (conv_tap.hip): per-tile partial sums + tickets of the ORDERED form of that reduce (no float atomics);
  // bn_bytes = extent of bn_x for its buffer descriptor
  float* bn_ws;
  int bn_bytes;
  // forward whose input is y = [relu](src * in_scale + in_shift) -- the BatchNorm (+ReLU) in front of this conv, whose output is
  // never materialised (dv_conv3d_fwd_bn_in): src is that BatchNorm's INPUT and the affine map is applied where the operand
  // fragments are formed, with dv_bn_apply's expression (same bits as the two-launch plan).  Channels >= in_C are pad lanes (zero).
  const float* in_scale = nullptr;
  const float* in_shift = nullptr;
  int in_C = 0, in_relu = 0;
};

template <int BYTES> struct VecB;
template <> struct VecB<16> { typedef uint4 type; static __device__ __forceinline__ uint4 zero() { return make_uint4(0, 0, 0, 0); } };
template <> struct VecB<8> { typedef uint2 type; static __device__ __forceinline__ uint2 zero() { return make_uint2(0, 0); } };

// Per-thread cursor over the k axis of the im2col matrix for one fixed vector slot.
struct KCursor {
  int c, dt, dh, dw, tap;
  __device__ __forceinline__ void init(int k0, const ConvGeom& g) {
    tap = k0 / g.CP;
    c = k0 - tap * g.CP;
    dw = tap % g.kw;
    int t2 = tap / g.kw;
    dh = t2 % g.kh;
    dt = t2 / g.kh;
  }
  __device__ __forceinline__ void advance(int step, const ConvGeom& g) {
    c += step;
    while (c >= g.CP) {
      c -= g.CP;
      ++tap;
      if (++dw == g.kw) {
        dw = 0;
        if (++dh == g.kh) { dh = 0; ++dt; }
      }
    }
  }
};

// Row of the im2col matrix: decoded once per thread.
struct RowPos {
  int base;      // n * sT*sH*sW
  int t0, h0, w0;
  bool valid;
};

template <int MODE>
__device__ __forceinline__ RowPos decode_row(uint32_t m, int M, const ConvGeom& g) {
  RowPos r;
  r.valid = (int)m < M;
  uint32_t q, wo, ho, to, n;
  fd_divmod(m, g.dW, q, wo);
  fd_divmod(q, g.dH, q, ho);
  fd_divmod(q, g.dT, n, to);
  r.base = (int)n * g.sT * g.sH * g.sW;
  if (MODE == MODE_FWD) {
    r.t0 = (int)to * g.st - g.pt;
    r.h0 = (int)ho * g.sh - g.ph;
    r.w0 = (int)wo * g.sw - g.pw;
  } else {
    r.t0 = (int)to + g.pt;
    r.h0 = (int)ho + g.ph;
    r.w0 = (int)wo + g.pw;
  }
  return r;
}

// source position (in elements/ld units) of (row, tap) or -1
template <int MODE>
__device__ __forceinline__ int src_pos(const RowPos& r, const KCursor& k, const ConvGeom& g) {
  int t, h, w;
  if (MODE == MODE_FWD) {
    t = r.t0 + k.dt; h = r.h0 + k.dh; w = r.w0 + k.dw;
  } else {
    t = r.t0 - k.dt; h = r.h0 - k.dh; w = r.w0 - k.dw;
    // strides are 1 or 2 (checked on the host)
    if (((t & (g.st - 1)) | (h & (g.sh - 1)) | (w & (g.sw - 1))) != 0) return -1;
    if ((t | h | w) < 0) return -1;
    t >>= (g.st - 1); h >>= (g.sh - 1); w >>= (g.sw - 1);
  }
  bool ok = r.valid && (k.dt < g.kt) && (unsigned)t < (unsigned)g.sT && (unsigned)h < (unsigned)g.sH &&
            (unsigned)w < (unsigned)g.sW;
  return ok ? r.base + (t * g.sH + h) * g.sW + w : -1;
}

// Blocks are dealt round-robin over the 8 XCDs (private L2 each): blocks b and b+8 share an XCD.  Remap so that each
// XCD gets a CONTIGUOUS range of logical tiles -- the tiles that re-read the same rows (all N tiles of one M tile,
// all (i,j) tiles of one wgrad row split) then hit the same L2.  Bijective for any grid size.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- global -> LDS DMA (buffer_load_dwordx4 ... lds) as inline assembly -------------------------------------------
// The __builtin_amdgcn_raw_ptr_buffer_load_lds form is tracked by the compiler's wait-count pass, which cannot prove
// that the following ds_reads touch the OTHER buffer and therefore drains vmcnt(0) in front of them: the prefetch of
// tile k+1 is then waited for before tile k is even read, i.e. no overlap inside a workgroup.  Issued as opaque
// assembly the load is invisible to that pass; the kernels wait for it themselves (dma_wait_all) right before the
// barrier that publishes the tile.  One wave instruction moves 64 lanes x 16 B to LDS [m0, m0 + 1 KiB).
typedef __attribute__((ext_vector_type(4))) unsigned int dma_rsrc_t;
__device__ __forceinline__ dma_rsrc_t dma_make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  dma_rsrc_t r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
  return r;
}
__device__ __forceinline__ void dma_load16(dma_rsrc_t rsrc, unsigned lds_base, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(lds_base), "v"(voff), "s"(rsrc)
               : "memory", "m0");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most n (wave-uniform, <= 24) of this wave's DMA pieces are still in flight
__device__ __forceinline__ void dma_wait_upto(int n) {
#define DV_W(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
  switch (n) {
    DV_W(1) DV_W(2) DV_W(3) DV_W(4) DV_W(5) DV_W(6) DV_W(7) DV_W(8) DV_W(9) DV_W(10) DV_W(11) DV_W(12)
    DV_W(13) DV_W(14) DV_W(15) DV_W(16) DV_W(17) DV_W(18) DV_W(19) DV_W(20) DV_W(21) DV_W(22) DV_W(23) DV_W(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef DV_W
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  // one K-tile = 64 bytes = 32 bf16 per row: two 32x32x16 steps
  // sa / sb: XOR masks on the 16-byte slot index (0 for the padded layout, (row>>2)&3 for the swizzled DMA layout)
  static __device__ __forceinline__ void tile(const unsigned char* a_row, const unsigned char* b_row, int h, int sa, int sb,
                                               f32x16& acc) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a = *reinterpret_cast<const bf16x8*>(a_row + ((2 * ks + h) ^ sa) * 16);
      bf16x8 b = *reinterpret_cast<const bf16x8*>(b_row + ((2 * ks + h) ^ sb) * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
};
template <> struct Mma<float> {
  // one K-tile = 64 bytes = 16 f32 per row: eight 32x32x2 steps
  static __device__ __forceinline__ void tile(const unsigned char* a_row, const unsigned char* b_row, int h, int sa, int sb,
                                               f32x16& acc) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      f32x4 a = *reinterpret_cast<const f32x4*>(a_row + (gq ^ sa) * 16);
      f32x4 b = *reinterpret_cast<const f32x4*>(b_row + (gq ^ sb) * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a.y : a.x, h ? b.y : b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a.w : a.z, h ? b.w : b.z, acc, 0, 0, 0);
    }
  }
};

// fp8 operands (1x1x1 convs of the bottleneck blocks, BASELINE configs[4]): one K tile = 64 bytes = 64 fp8 per row = ONE
// v_mfma_f32_32x32x64_f8f6f4 (twice the bf16 rate per clock); lane (r = lane&31, h = lane>>5) holds k = 32h .. 32h+31 of
// its row in eight VGPRs.  FA / FB: 0 = e4m3, 1 = e5m2 (cbsz / blgp).  Scales 0 select the unscaled form of the instruction.
template <int FA, int FB>
__device__ __forceinline__ void mma_fp8(const unsigned char* a_row, const unsigned char* b_row, int h, int sa, int sb, f32x16& acc) {
  const i32x4 a0 = *reinterpret_cast<const i32x4*>(a_row + ((2 * h) ^ sa) * 16), a1 = *reinterpret_cast<const i32x4*>(a_row + ((2 * h + 1) ^ sa) * 16);
  const i32x4 b0 = *reinterpret_cast<const i32x4*>(b_row + ((2 * h) ^ sb) * 16), b1 = *reinterpret_cast<const i32x4*>(b_row + ((2 * h + 1) ^ sb) * 16);
  const i32x8 a = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, b = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, FA, FB, 0, 0, 0, 0);
}
template <> struct Mma<fp8e4_t> {       // forward: x (e4m3) * w (e4m3)
  static __device__ __forceinline__ void tile(const unsigned char* a_row, const unsigned char* b_row, int h, int sa, int sb, f32x16& acc) {
    mma_fp8<0, 0>(a_row, b_row, h, sa, sb, acc);
  }
};
template <> struct Mma<fp8e5_t> {       // data gradient: dy (e5m2) * w (e4m3)
  static __device__ __forceinline__ void tile(const unsigned char* a_row, const unsigned char* b_row, int h, int sa, int sb, f32x16& acc) {
    mma_fp8<1, 0>(a_row, b_row, h, sa, sb, acc);
  }
};
template <typename T> struct OutOf { typedef T type; };
template <> struct OutOf<fp8e4_t> { typedef bf16_t type; };          // fp8 GEMMs write bf16 activations / gradients
template <> struct OutOf<fp8e5_t> { typedef bf16_t type; };

// ---- fp32 operands on the bf16 matrix cores ("3 x bf16 split") ----------------------------------------------------
// gfx950 has no xf32 / TF32 path and its f32-input MFMA runs at the vector rate, 1/16 of the bf16 MFMA.  An fp32 value
// is EXACTLY hi + mid + lo with three bf16 (8 significant bits each = fp32's 24; round-to-nearest residues), and a
// product a*b is the sum of nine partial products of which the six of weight >= 2^-16 are kept: the three dropped ones
// are <= 2^-24 |a||b| each, i.e. at the level of fp32's own rounding of the product.  Every partial product of two bf16
// is exact in fp32 and the MFMA accumulates in fp32, small terms first.  Six bf16 MFMAs (32 cycles each) replace eight
// 32x32x2 f32 MFMAs (64 cycles each) per 32x32x16 block: 2.67x less matrix-pipe time at fp32-level accuracy
// (DUALVAR_F32_EXACT=1 selects the exact-f32 MFMA kernels instead; tests/test_ops_gpu.py compares both with torch fp32).
struct Split3 { bf16x8 hi, mid, lo; };
__device__ __forceinline__ Split3 split3(const float (&v)[8]) {
  Split3 s;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bf16_t h = (bf16_t)v[e];
    const float r1 = v[e] - (float)h;
    const bf16_t m = (bf16_t)r1;
    const float r2 = r1 - (float)m;
    s.hi[e] = h; s.mid[e] = m; s.lo[e] = (bf16_t)r2;
  }
  return s;
}
// The weight-gradient kernels' split (they are bound by the vector-instruction count of exactly this function):
// hi = bf16(x), rounded to nearest (one v_cvt_pk_bf16_f32 per pair); r1 = x - hi is exact, has <= 16 significant bits and a
// sign that does not follow x's; mid = the TOP 16 BITS of r1 (truncated: the bf16 is the upper half of the fp32 word, so the
// pair is one v_perm_b32 and widening it back is one v_and per value); lo = r1 - mid has <= 8 significant bits and is exact
// in bf16 (again the upper half).  Per pair of values: 1 conversion, 2 widenings, 2 v_and, 2 packed subtractions, 2 v_perm =
// 36 vector instructions per 8 values; rounding mid and lo as well takes 42, and these kernels are bound by that count
// (weight gradients -6.6 %, step -1.1 %).  The forward / data-gradient kernels keep the fully rounded split3 above: they are
// MFMA bound, and with the truncated mid the loss after two SGD steps moved from 1.1e-3 to 1.6e-3 of the reference's on the
// R(2+1)D fixture (a weight gradient enters the next step scaled by the learning rate; an activation enters it directly).  Truncating hi too would cost the same 36 but biases every dropped partial product
// (mid*lo, lo*mid, lo*lo) towards the sign of x*y: in the long, cancelling sums of a weight gradient that bias showed as
// 1e-1 relative differences between a batch and its two halves (1e-3 .. 1e-2 with the rounded hi).
// one pair of values -> the pair's dword of each of the three planes (the step split3w repeats four times)
__device__ __forceinline__ void split3w_pair(float x0, float x1, unsigned& H, unsigned& Mi, unsigned& Lo) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const u32x2 mask = {0xffff0000u, 0xffff0000u};
  const f32x2 x = {x0, x1};
  const bf16x2 hb = __builtin_convertvector(x, bf16x2);              // one v_cvt_pk_bf16_f32
  const unsigned hw = __builtin_bit_cast(unsigned, hb);
  const u32x2 hwide = {hw << 16, hw & 0xffff0000u};
  const f32x2 r1 = x - __builtin_bit_cast(f32x2, hwide);
  const u32x2 rb = __builtin_bit_cast(u32x2, r1);
  const f32x2 r2 = r1 - __builtin_bit_cast(f32x2, rb & mask);
  const u32x2 qb = __builtin_bit_cast(u32x2, r2);
  H = hw;
  Mi = __builtin_amdgcn_perm(rb.y, rb.x, 0x07060302u);
  Lo = __builtin_amdgcn_perm(qb.y, qb.x, 0x07060302u);
}
__device__ __forceinline__ Split3 split3w(const float (&v)[8]) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
  u32x4v H, Mi, Lo;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned h_, m_, l_;
    split3w_pair(v[2 * p], v[2 * p + 1], h_, m_, l_);
    H[p] = h_; Mi[p] = m_; Lo[p] = l_;
  }
  Split3 s;
  s.hi = __builtin_bit_cast(bf16x8, H); s.mid = __builtin_bit_cast(bf16x8, Mi); s.lo = __builtin_bit_cast(bf16x8, Lo);
  return s;
}
// compile-time loop: f(std::integral_constant<int, 0>()) ... f(std::integral_constant<int, N - 1>())
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>()), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>()); }
__device__ __forceinline__ void mma_split3(const Split3& a, const Split3& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
// K-contiguous LDS row (64 bytes = 16 f32 per K tile, 16-byte slots XOR-swizzled by s): the 8 floats k = 8h .. 8h+7
__device__ __forceinline__ Split3 split3_row(const unsigned char* row, int h, int s) {
  const f32x4 lo4 = *reinterpret_cast<const f32x4*>(row + ((2 * h) ^ s) * 16);
  const f32x4 hi4 = *reinterpret_cast<const f32x4*>(row + ((2 * h + 1) ^ s) * 16);
  const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  return split3(v);
}

__device__ __forceinline__ float act_apply(float v, int flags) {
  if (flags & DV_RELU) v = fmaxf(v, 0.f);
  if (flags & DV_SIGMOID) v = 1.f / (1.f + __expf(-v));
  return v;
}

// ---- weight-gradient argument blocks (conv.hip, conv_experiments.hip) ----------------------------------------------------
struct WgradArgs {
  const void* x;
  const void* dy;
  float* dw;
  float* slab;            // [splits][slab_stride] partial sums, or nullptr (splits == 1: dW += tile)
  long long slab_stride;  // elements between consecutive splits (>= Cout * ldw)
  int M, Cout, CoutP, J;  // rows, output channels (padded), J = taps*CP
  int ldx, ldy, ldw;
  int nti, ntj;
  int rows_per_split;
  ConvGeom g;
  RowPerm perm;           // row order of the DMA kernel (see RowPerm)
};

// one 32x32 accumulator block -> slab / dW.  Per register a half wave stores 32 consecutive floats (128 B).
__device__ __forceinline__ void wgrad_store_block(const WgradArgs& a, int split, int row0, int col, int h, const f32x16& acc) {
  if (col >= a.J) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (row < a.Cout) {
      const size_t e = (size_t)row * a.ldw + col;
      if (a.slab) a.slab[(size_t)split * a.slab_stride + e] = acc[r];
      else a.dw[e] += acc[r];
    }
  }
}


struct WgradDmaArgs {
  WgradArgs w;
  int x_bytes, dy_bytes;
  // BNA (dv_conv3d_wgrad_bn): the dY operand is formed from g = w.dy and the BatchNorm's input bn_x (same rows / pitch)
  const void* bn_x;
  const float *bn_mean, *bn_invstd, *bn_gamma, *bn_scale, *bn_shift, *bn_sums;
  float *bn_dgamma, *bn_dbeta;
  float bn_inv_count, bn_dscale;
  int bn_rep, bn_mask;
};

// rows of RB bytes: four consecutive rows must fall on four different 64-byte bank groups of the 256-byte LDS line
// (RB a multiple of 256: XOR the 64-byte chunk index with row&3; RB = 128 mod 256, i.e. 128 or 384: rows 0/1 already
// differ by 128 bytes, XOR chunk bit 0 with (row>>1)&1)
template <int RB> __device__ __forceinline__ int wg_swz(int row) { return (RB % 256) ? ((row >> 1) & 1) : (row & 3); }

template <int RB>
__device__ __forceinline__ bf16x8 wg_frag(const unsigned char* tile, int col0, int lane, int ks) {
  const int g16 = lane >> 4, li = lane & 15;
  const int q = li >> 2, p = li & 3;
  const int row = ks * 16 + 8 * (g16 >> 1) + q;              // wg_swz(row) == wg_swz(q) == wg_swz(row + 4)
  const int colb = ((col0 + 16 * (g16 & 1) + 4 * p) * 2) ^ (wg_swz<RB>(q) << 6);
  const unsigned char* ad = tile + row * RB + colb;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ad + 4 * RB));
  s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- weight gradient of the stride-1 3x1x1 convs with LDS-staged, once-split operand tiles (conv_tap_wgrad.hip) -------------
struct TmWgradArgs {
  const void* x;          // [N][T][S][ldx] fp32
  const void* dy;         // [N][T][S][ldy] fp32
  float* dw;              // [Cout][taps * CP] fp32 (+=) when splits == 1
  float* slab;            // [splits][slab_stride] partial tiles, or nullptr
  long long slab_stride;
  int NQ, S, T;           // N * S pixels, pixels per frame, frames of dY (2 or 4)
  int Cout, CoutP, CP;    // output channels (and pitch of dY's channels), channel pitch of x
  int ldx, ldy, ldw;      // pitches in elements; ldw = taps * CP
  int nti, ntc;           // 64-channel tiles of Cout; 64- (kind 1) / 32-channel (kind 2) tiles of CP
  int nchunks, chunks_per_split;      // steps (64 / T pixels each) in all / per row split
  int x_bytes, dy_bytes;
  int kind;               // 1: stride 1, 3 taps, T = 2 / 4 frames; 2: 7 taps, stride 2, 8 -> 4 frames (the stem conv);
                          // 3: spatial 1x3x3, stride 1, padding 1 (M, H, W, fW, fH below; nchunks = 64-row steps)
  FastDiv fS;
  int M, H, W;
  FastDiv fW, fH;
  // kind 4: the pixel-pair RGB stem conv (window 1 x 7 x 4 over 8-channel pixel pairs, stride (1, 2, 1), no padding): one output
  // line per step.  Wo / Ho output pixels per line / lines per frame, Wp / Hp pairs per input line / input lines per frame,
  // nchunks = N * T * Ho lines.  bn_x != nullptr: the dY operand is dv_bn_bwd_apply's output formed on the fly (dv_conv3d_wgrad_bn)
  int Wo, Ho, Wp, Hp;
  const void* bn_x;
  const float *bn_mean, *bn_invstd, *bn_gamma, *bn_scale, *bn_shift, *bn_sums;
  float *bn_dgamma, *bn_dbeta;
  float bn_inv_count, bn_dscale;
  int bn_rep, bn_mask;
  // kinds 1, 2: the x operand is y = [relu](x * in_scale + in_shift) formed on the fly (dv_conv3d_wgrad_bn_in; see ConvArgs)
  const float* in_scale = nullptr;
  const float* in_shift = nullptr;
  int in_C = 0, in_relu = 0;
};

}  // namespace
