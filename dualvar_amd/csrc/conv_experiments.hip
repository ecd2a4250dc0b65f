// Kernels that were built, measured and NOT selected (DESIGN.md section 4 records why), kept compiling and under test so
// that the record stays reproducible: the second fp32 weight-gradient form (conv_wgrad_f32s_kernel, opt-in through
// DUALVAR_WGRAD_F32S).  The hot path is conv.hip / conv_tap.hip.
#include "conv_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// fp32 weight gradient, second form ("f32s"): products on the bf16 matrix cores as above, but laid out around the count of
// vector instructions the operand SPLITS cost, which is what bounded conv_wgrad_dma_kernel<float, 64, 128, 1, 4, 2, true, true>
// (PMC: vector issue 77 % busy, matrix pipe 42 %; ~210 vector instructions against 24 MFMAs per wave and 32-row step):
//  * 1 x 4 waves, wave tiles BI x BJ/4 with BJ/4 = 64 (or 32): a wave splits its OWN x columns once per step and uses each
//    fragment for TI = BI/32 blocks; the dY fragments, which every wave needs, are split once per WORKGROUP (a 64-lane
//    "fragment op" per 8 rows x 64 columns, dealt over the waves) and shared through LDS planes in fragment order
//    ([k group][hi|mid|lo][column] bf16x8: conflict-free ds_write_b128 / ds_read_b128).  Split instructions per MFMA:
//    64x256: 3.75, 128x128: 3.0, 128x256: 2.25 (the 64x128 form above: 4.5, and its 2x2 predecessor 9);
//  * ONE barrier per step instead of two: the planes are double buffered, and the loop is rotated -- iteration s multiplies
//    step s (planes[s & 1], the x fragments split one iteration earlier) and, in the same basic block, splits step s + 1
//    (vector ALU and LDS work the scheduler can place between the MFMAs), then waits for the DMA of step s + 2 and meets the
//    barrier.  Three LDS stages of fp32 tiles (prefetch distance 2): a tile has a whole iteration to land;
//  * ROWS = 16 rows per step keep a workgroup at <= 80 KB of LDS, i.e. two workgroups per CU whose phases interleave.
// Everything else (row table, tap masks, t-inner row order, slab epilogue) is the DMA kernel's.
constexpr int wgrad_f32s_lds(int bi, int bj, int rows, int ns) {
  return ns * rows * (bi + bj) * 4 + 2 * (rows / 8) * 3 * bi * 16 + 2 * 256 * 12;
}
constexpr int wgrad_f32s_waves_per_simd(int bi, int bj, int rows, int ns) {
  return 163840 / wgrad_f32s_lds(bi, bj, rows, ns) >= 2 ? 2 : 1;
}

template <int BI, int BJ, int ROWS, int NS>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(wgrad_f32s_waves_per_simd(BI, BJ, ROWS, NS)))) void conv_wgrad_f32s_kernel(WgradDmaArgs aa) {
  const WgradArgs& a = aa.w;
  constexpr int NW = 4, NT = 256;
  constexpr int RBP = BI * 4, RBQ = BJ * 4;          // row bytes of the dY (P) and im2col (Q) tiles
  constexpr int DP = RBP / 16, DQ = RBQ / 16;        // 16-byte slots per row
  constexpr int QOFF = ROWS * RBP, BUFB = ROWS * (RBP + RBQ);
  static_assert(QOFF % 1024 == 0 && BUFB % 1024 == 0, "1 KiB DMA pieces");
  constexpr int PPC = QOFF / 1024, QPC = (BUFB - QOFF) / 1024;
  constexpr int NPW = (PPC + NW - 1) / NW, NQW = (QPC + NW - 1) / NW;
  constexpr int RT = NT, SPR = RT / ROWS;
  constexpr int WJ = BJ / NW, TI = BI / 32, TJ = WJ / 32;
  constexpr int NKS = ROWS / 16, NKH = ROWS / 8;     // MFMA k steps / 8-row k groups per step
  constexpr int NPF = NKH * (BI / 64);               // dY fragment ops (8 rows x 64 columns) per step
  constexpr int PLN = NKH * 3 * BI;                  // uint4 entries of one planes buffer
  constexpr int D = NS - 1;
  static_assert(TI >= 1 && TJ >= 1 && BI % 64 == 0 && ROWS % 16 == 0, "tile");
  static_assert(D >= 2 && D < SPR, "the rotated loop needs a prefetch distance of two steps");
  constexpr unsigned kOOB = 0x80000000u;

  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS * BUFB];
  __shared__ uint2 rowtab[2][RT];
  __shared__ unsigned rowdy[2][RT];
  __shared__ uint4 planes[2 * PLN];

  const ConvGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_j = bid % a.ntj; bid /= a.ntj;
  const int tile_i = bid % a.nti;
  const int split = bid / a.nti;
  const int i0 = tile_i * BI, j0 = tile_j * BJ;
  const int wj0 = wave * WJ;
  const int m_begin = split * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);

  const dma_rsrc_t x_rsrc = dma_make_rsrc(a.x, (unsigned)aa.x_bytes), dy_rsrc = dma_make_rsrc(a.dy, (unsigned)aa.dy_bytes);
  const unsigned smem_base = lds_addr(smem);
  const unsigned ldxb = (unsigned)a.ldx * 4u, ldyb = (unsigned)a.ldy * 4u;

  int prow[NPW]; unsigned pcolb[NPW];
#pragma unroll
  for (int u = 0; u < NPW; ++u) {
    const int sl = (wave + NW * u) * 64 + lane;
    prow[u] = sl / DP;
    const int n = i0 + (sl % DP) * 4;
    pcolb[u] = n < a.CoutP ? (unsigned)n * 4u : kOOB;
  }
  int qrow[NQW]; unsigned qtb[NQW], qbit[NQW];
#pragma unroll
  for (int u = 0; u < NQW; ++u) {
    const int sl = (wave + NW * u) * 64 + lane;
    qrow[u] = sl / DQ;
    const int col = j0 + (sl % DQ) * 4;
    if (col < a.J) {
      const int tap = col / g.CP, c = col - tap * g.CP;
      const int dw = tap % g.kw, t2 = tap / g.kw, dh = t2 % g.kh, dt = t2 / g.kh;
      qbit[u] = (1u << dt) | (1u << (8 + dh)) | (1u << (16 + dw));
      qtb[u] = (unsigned)((dt * g.sH + dh) * g.sW + dw) * ldxb + (unsigned)c * 4u;
    } else {
      qbit[u] = 0xffffffffu;
      qtb[u] = 0;
    }
  }

  auto decode = [&](int rnd) {
    const int q = m_begin + rnd * RT + tid;
    uint2 e = make_uint2(0u, 0u);
    unsigned dyo = kOOB;
    if (q < m_end) {
      const uint32_t m = a.perm.on ? perm_row(a.perm, (uint32_t)q) : (uint32_t)q;
      dyo = m * ldyb;
      const RowPos r = decode_row<MODE_FWD>(m, a.M, g);
      auto range = [](int x0, int k, int lim) -> unsigned {
        const int lo = max(0, -x0), hi = min(k, lim - x0);
        return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
      };
      const unsigned bt = range(r.t0, g.kt, g.sT), bh = range(r.h0, g.kh, g.sH), bw = range(r.w0, g.kw, g.sW);
      e.x = (unsigned)(r.base + (r.t0 * g.sH + r.h0) * g.sW + r.w0) * ldxb;
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
      if (PPC % NW != 0 && wave + NW * u >= PPC) break;
      const unsigned off = (dyo[u] != kOOB && pcolb[u] != kOOB) ? dyo[u] + pcolb[u] : kOOB;
      dma_load16(dy_rsrc, smem_base + buf * BUFB + (wave + NW * u) * 1024, off);
    }
    const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(rowtab[(s / SPR) & 1] + (s % SPR) * ROWS);
    unsigned long long e[NQW];
#pragma unroll
    for (int u = 0; u < NQW; ++u) e[u] = tab[qrow[u] & (ROWS - 1)];
#pragma unroll
    for (int u = 0; u < NQW; ++u) {
      if (QPC % NW != 0 && wave + NW * u >= QPC) break;
      const unsigned ex = (unsigned)e[u], ey = (unsigned)(e[u] >> 32);
      const unsigned off = ((ey & qbit[u]) == qbit[u]) ? ex + qtb[u] : kOOB;
      dma_load16(x_rsrc, smem_base + buf * BUFB + QOFF + (wave + NW * u) * 1024, off);
    }
  };
  int my_pieces = 0;
#pragma unroll
  for (int u = 0; u < NPW; ++u) my_pieces += (wave + NW * u < PPC) ? 1 : 0;
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
  const int nsteps = (m_end - m_begin + ROWS - 1) / ROWS;

  // split step s: this wave's share of the dY fragment ops -> planes[s & 1]; its own x fragments -> bq
  auto split_step = [&](int s, Split3 (&bq)[NKS][TJ]) {
    const unsigned char* tp = smem + (s % NS) * BUFB;
    const unsigned char* tq = tp + QOFF;
    uint4* pl = planes + (s & 1) * PLN;
    static_assert(NPF % NW == 0 || NW % NPF == 0, "fragment ops deal evenly over the waves");
#pragma unroll
    for (int f0 = 0; f0 < NPF; f0 += NW) {
      // fewer ops than waves: the spare waves repeat one (same values to the same place) rather than branch around it --
      // the step takes as long as its slowest wave either way, and a branch would cut the block the scheduler interleaves
      const int f = NPF % NW == 0 ? f0 + wave : (f0 + wave) % NPF;
      {
        const int kh = f % NKH, col = (f / NKH) * 64 + lane;
        const unsigned char* src = tp + (kh * 8) * RBP + col * 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(src + e * RBP);
        const Split3 s3 = split3w(v);
        uint4* dst = pl + kh * 3 * BI + col;
        dst[0] = __builtin_bit_cast(uint4, s3.hi);
        dst[BI] = __builtin_bit_cast(uint4, s3.mid);
        dst[2 * BI] = __builtin_bit_cast(uint4, s3.lo);
      }
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(tq + (ks * 16 + 8 * h + e) * RBQ + (wj0 + j * 32 + l31) * 4);
        bq[ks][j] = split3w(v);
      }
  };
  auto mma_step = [&](int s, const Split3 (&bq)[NKS][TJ]) {
    const uint4* pl = planes + (s & 1) * PLN;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int kh = ks * 2 + h;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const uint4* pa = pl + kh * 3 * BI + i * 32 + l31;
        Split3 af;
        af.hi = __builtin_bit_cast(bf16x8, pa[0]);
        af.mid = __builtin_bit_cast(bf16x8, pa[BI]);
        af.lo = __builtin_bit_cast(bf16x8, pa[2 * BI]);
#pragma unroll
        for (int j = 0; j < TJ; ++j) mma_split3(af, bq[ks][j], acc[i][j]);
      }
    }
  };
  // iteration s (-1 <= s < nsteps): multiply step s, split step s + 1, prefetch step s + 1 + D, make step s + 2 visible.
  // Steady iterations (0 <= s, s + 1 + D < nsteps: every part exists, no conditions) run steady_hand below; the first and the
  // last D + 1 iterations take the conditional form `iteration`.
  static_assert(PPC % NW == 0 && QPC % NW == 0, "every wave issues the same pieces per step (compile-time wait counts)");
  constexpr int PIECES = PPC / NW + QPC / NW;
  static_assert((D - 1) * PIECES <= 24, "counted wait");
  auto wait_tail = [&](int tiles) { dma_wait_upto(tiles * PIECES); };
  // The steady iteration in HAND-PLACED order (HAND): an in-order wave only fills the 24 idle issue cycles behind an MFMA
  // with what follows it in ITS OWN instruction stream, and the compiler's scheduler clusters the MFMAs of a block (24 back to
  // back, then the ~190 vector instructions of the split: matrix pipe and vector ALU take turns, PMC: matrix pipe 52 % busy
  // with two waves per SIMD).  Here every MFMA is followed by one UNIT of the next step's work -- a pair of values split
  // (9 vector instructions), the three plane stores of a dY fragment, or one DMA piece's address and issue -- and a
  // sched_barrier(0) pins that order.  All LDS reads of the iteration are issued up front.
  constexpr int NPFW = NPF >= NW ? NPF / NW : 1;                   // dY fragment ops per wave and step
  constexpr int NM = NKS * TI * TJ * 6;                            // MFMAs per wave and step
  constexpr int U_P = NPFW * 5, U_Q = NKS * TJ * 4, U_ALL = U_P + U_Q + PIECES;
  auto steady_hand = [&](int s, int st_next, int st_issue, const Split3 (&bcur)[NKS][TJ], Split3 (&bnext)[NKS][TJ]) {
    const int sn = s + 1;
    if ((sn % SPR) == 0 && (sn + SPR) < nsteps) decode(sn / SPR + 1);
    const uint4* pl = planes + (s & 1) * PLN;
    uint4* pln = planes + (sn & 1) * PLN;
    const unsigned char* tp = smem + st_next * BUFB;
    const unsigned char* tq = tp + QOFF;
    Split3 af[NKS][TI];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const uint4* pa = pl + (ks * 2 + h) * 3 * BI + i * 32 + l31;
        af[ks][i].hi = __builtin_bit_cast(bf16x8, pa[0]);
        af[ks][i].mid = __builtin_bit_cast(bf16x8, pa[BI]);
        af[ks][i].lo = __builtin_bit_cast(bf16x8, pa[2 * BI]);
      }
    float pv[NPFW][8], qv[NKS][TJ][8];
    int pdst[NPFW];
#pragma unroll
    for (int f0 = 0; f0 < NPFW; ++f0) {
      const int f = NPF % NW == 0 ? f0 * NW + wave : wave % NPF;
      const int kh = f % NKH, col = (f / NKH) * 64 + lane;
      pdst[f0] = kh * 3 * BI + col;
      const unsigned char* src = tp + (kh * 8) * RBP + col * 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[f0][e] = *reinterpret_cast<const float*>(src + e * RBP);
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          qv[ks][j][e] = *reinterpret_cast<const float*>(tq + (ks * 16 + 8 * h + e) * RBQ + (wj0 + j * 32 + l31) * 4);
    const int si = sn + D;
    const unsigned* dtab = rowdy[(si / SPR) & 1] + (si % SPR) * ROWS;
    const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(rowtab[(si / SPR) & 1] + (si % SPR) * ROWS);
    unsigned dyo[NPW];
    unsigned long long te[NQW];
#pragma unroll
    for (int u = 0; u < NPW; ++u) dyo[u] = dtab[prow[u]];
#pragma unroll
    for (int u = 0; u < NQW; ++u) te[u] = tab[qrow[u] & (ROWS - 1)];
    const unsigned stage_base = smem_base + st_issue * BUFB;
    __builtin_amdgcn_sched_barrier(0);
    unsigned PH[NPFW][4], PM[NPFW][4], PL[NPFW][4], QH[NKS][TJ][4], QM[NKS][TJ][4], QL[NKS][TJ][4];
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4v;
    auto unit = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k < U_P) {
        constexpr int f = k / 5, q = k % 5;
        if constexpr (q < 4) {
          split3w_pair(pv[f][2 * q], pv[f][2 * q + 1], PH[f][q], PM[f][q], PL[f][q]);
        } else {
          uint4* dst = pln + pdst[f];
          dst[0] = make_uint4(PH[f][0], PH[f][1], PH[f][2], PH[f][3]);
          dst[BI] = make_uint4(PM[f][0], PM[f][1], PM[f][2], PM[f][3]);
          dst[2 * BI] = make_uint4(PL[f][0], PL[f][1], PL[f][2], PL[f][3]);
        }
      } else if constexpr (k < U_P + U_Q) {
        constexpr int idx = k - U_P, fr = idx / 4, q = idx % 4, ks = fr / TJ, j = fr % TJ;
        split3w_pair(qv[ks][j][2 * q], qv[ks][j][2 * q + 1], QH[ks][j][q], QM[ks][j][q], QL[ks][j][q]);
        // (pin the results HERE: their only reader is the next iteration, and the machine-sink pass otherwise moves the whole split
        // behind the barrier into the successor block -- correct, but it is this placement between the MFMAs that is wanted)
        asm volatile("" : "+v"(QH[ks][j][q]), "+v"(QM[ks][j][q]), "+v"(QL[ks][j][q]));
        if constexpr (q == 3) {
          const u32x4v vh = {QH[ks][j][0], QH[ks][j][1], QH[ks][j][2], QH[ks][j][3]};
          const u32x4v vm = {QM[ks][j][0], QM[ks][j][1], QM[ks][j][2], QM[ks][j][3]};
          const u32x4v vl = {QL[ks][j][0], QL[ks][j][1], QL[ks][j][2], QL[ks][j][3]};
          bnext[ks][j].hi = __builtin_bit_cast(bf16x8, vh);
          bnext[ks][j].mid = __builtin_bit_cast(bf16x8, vm);
          bnext[ks][j].lo = __builtin_bit_cast(bf16x8, vl);
        }
      } else {
        constexpr int u = k - U_P - U_Q;
        if constexpr (u < NPW) {
          const unsigned off = (dyo[u] != kOOB && pcolb[u] != kOOB) ? dyo[u] + pcolb[u] : kOOB;
          dma_load16(dy_rsrc, stage_base + (wave + NW * u) * 1024, off);
        } else {
          constexpr int v = u - NPW;
          const unsigned ex = (unsigned)te[v], ey = (unsigned)(te[v] >> 32);
          const unsigned off = ((ey & qbit[v]) == qbit[v]) ? ex + qtb[v] : kOOB;
          dma_load16(x_rsrc, stage_base + QOFF + (wave + NW * v) * 1024, off);
        }
      }
    };
    static_for<NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int blk = m / 6, prod = m % 6, j = blk % TJ, i = (blk / TJ) % TI, ks = blk / (TJ * TI);
      const Split3& A = af[ks][i];
      const Split3& B = bcur[ks][j];
      if constexpr (prod == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.mid, B.mid, acc[i][j], 0, 0, 0);
      if constexpr (prod == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.lo, acc[i][j], 0, 0, 0);
      if constexpr (prod == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.lo, B.hi, acc[i][j], 0, 0, 0);
      if constexpr (prod == 3) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.mid, acc[i][j], 0, 0, 0);
      if constexpr (prod == 4) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.mid, B.hi, acc[i][j], 0, 0, 0);
      if constexpr (prod == 5) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.hi, acc[i][j], 0, 0, 0);
      constexpr int k0 = (m * U_ALL + NM - 1) / NM, k1 = ((m + 1) * U_ALL + NM - 1) / NM;     // units of this slot
      static_for<(k1 - k0)>([&](auto dc) { unit(std::integral_constant<int, k0 + decltype(dc)::value>()); });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else wait_tail(D - 1);
    __syncthreads();
  };
  // the first and the last D + 1 iterations: every part behind its condition (compiler-scheduled)
  auto iteration = [&](int s, const Split3 (&bcur)[NKS][TJ], Split3 (&bnext)[NKS][TJ]) {
    const int sn = s + 1;
    if (sn < nsteps && (sn % SPR) == 0 && (sn + SPR) < nsteps) decode(sn / SPR + 1);
    if (sn + D < nsteps) issue(sn + D, (sn + D) % NS);          // its stage was last read by split_step(s), one barrier ago
    if (s >= 0) mma_step(s, bcur);
    if (sn < nsteps) split_step(sn, bnext);
    if (sn + 1 < nsteps) wait_tail(min(D - 1, nsteps - 2 - sn));  // step s + 2 has landed (this wave's pieces)
    __syncthreads();
  };

  decode(0);
  if (SPR < nsteps) decode(1);                       // (the loop decodes round r + 1 at the first step of round r, r >= 1)
  __syncthreads();
  for (int t = 0; t < D && t < nsteps; ++t) issue(t, t);
  wait_tail(min(D - 1, nsteps - 1));                 // step 0 has landed
  __syncthreads();
  Split3 b0[NKS][TJ], b1[NKS][TJ];
  {
    // iteration -1 without its decode (round 1 is decoded above)
    if (D < nsteps) issue(D, D % NS);
    split_step(0, b0);
    if (1 < nsteps) wait_tail(min(D - 1, nsteps - 2));
    __syncthreads();
  }
  // steady iterations: s + 1 + D < nsteps, in pairs (the x fragments alternate between two register sets)
  const int n_steady = max(0, nsteps - 1 - D) & ~1;
  int s = 0;
  {
    int st_next = 1 % NS, st_issue = (1 + D) % NS;               // stages of steps s + 1 and s + 1 + D at s = 0
    auto adv = [&] { st_next = st_next + 1 == NS ? 0 : st_next + 1; st_issue = st_issue + 1 == NS ? 0 : st_issue + 1; };
    for (; s < n_steady; s += 2) {
      steady_hand(s, st_next, st_issue, b0, b1); adv();
      steady_hand(s + 1, st_next, st_issue, b1, b0); adv();
    }
  }
  for (; s < nsteps; s += 2) {
    iteration(s, b0, b1);
    if (s + 1 < nsteps) iteration(s + 1, b1, b0);
  }

#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
      wgrad_store_block(a, split, i0 + i * 32, j0 + wj0 + j * 32 + l31, h, acc[i][j]);
}


// ------------------------------------------------------------------------------------------------------------------------
// dv_conv3d_dgrad_bn, the ATOMIC form of the BatchNorm-backward reduce behind a data gradient (engine.FUSE_BN_REDUCE, off by
// default; rounds 1 - 3 carried it as a tail inside conv_gemm_kernel): a pass over the tile rows the data gradient has just
// written (L2-resident for the layers it was meant for) and the BatchNorm's input, sum g and sum g * xhat per column, one atomic
// per column and 128-row tile into replica tile % n_rep.  Measured a loss on every step it was tried on; kept as a
// tested entry point.  The ORDERED form that the default plan uses lives in conv_tap.hip (dv_conv3d_dgrad_bn_ws).
struct BnTailArgs {
  const void* g;          // dL/dy as written by the data gradient, [M][ldg]
  const void* x;          // the BatchNorm's input, [M][ldx]
  const float *mean, *invstd, *scale, *shift;
  float* sums;            // [n_rep][2][cp8(N)]
  int M, N, NP, ldg, ldx, n_rep, mask;
};

template <typename T>
__global__ __launch_bounds__(256) void dgrad_bn_tail_kernel(BnTailArgs a) {
  constexpr int BM = 128, BN = 64, NT = 256, EO = (int)sizeof(T), EPC = 16 / EO, CPT = BN * EO / 16, RG = NT / CPT;
  __shared__ float red[RG * 2 * BN];
  const int tid = threadIdx.x;
  const int ntn = (a.NP + BN - 1) / BN;
  const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ch = tid % CPT, rg = tid / CPT, col0 = n0 + ch * EPC;
  float mu[EPC], is[EPC], sc[EPC], sh[EPC], s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const bool ok = col0 + e < a.N;
    mu[e] = ok ? a.mean[col0 + e] : 0.f;
    is[e] = ok ? a.invstd[col0 + e] : 0.f;
    sc[e] = (ok && a.mask) ? a.scale[col0 + e] : 0.f;
    sh[e] = (ok && a.mask) ? a.shift[col0 + e] : 0.f;
    s1[e] = s2[e] = 0.f;
  }
  if (col0 < a.NP) {
    const int rows_here = min(BM, a.M - m0);
    const T* g = reinterpret_cast<const T*>(a.g);
    const T* x = reinterpret_cast<const T*>(a.x);
    for (int r = rg; r < rows_here; r += RG) {
      const size_t row = (size_t)(m0 + r);
      float gq[EPC], xq[EPC];
      Pack16<T>::load(g + row * a.ldg + col0, gq);
      Pack16<T>::load(x + row * a.ldx + col0, xq);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float act = xq[e] * sc[e] + sh[e];          // the forward's expression (dv_bn_apply), same rounding
        const float gg = (a.mask && !(act > 0.f)) ? 0.f : gq[e];
        s1[e] += gg;
        s2[e] += gg * (xq[e] - mu[e]) * is[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    red[(rg * 2 + 0) * BN + ch * EPC + e] = s1[e];
    red[(rg * 2 + 1) * BN + ch * EPC + e] = s2[e];
  }
  __syncthreads();
  if (tid < BN && n0 + tid < a.N) {
    float t1 = 0.f, t2 = 0.f;
    for (int w = 0; w < RG; ++w) { t1 += red[(w * 2 + 0) * BN + tid]; t2 += red[(w * 2 + 1) * BN + tid]; }
    const int cpb = (a.N + 7) & ~7;
    float* dst = a.sums + (size_t)(tile_m % a.n_rep) * 2 * cpb + n0 + tid;
    atomicAdd(dst, t1);
    atomicAdd(dst + cpb, t2);
  }
}

}  // namespace

// ---- entry points for conv.hip (plain C++ symbols of the shared object's own translation units; the argument block is
// conv_common.hpp's WgradDmaArgs, passed by address because each translation unit has its own anonymous-namespace copy of
// the type).  Configurations: 0 = 64 x 256 tile, 16 rows per step; 2 = 128 x 128, 16 rows (the two the sweep kept: the other
// four shapes of round 3 -- 32-row steps, 128 x 256, 64 x 128 -- were never faster anywhere and are gone).
int dvx_wgrad_f32s_lds_bytes(int cfg) {
  return cfg == 0 ? wgrad_f32s_lds(64, 256, 16, 3) : cfg == 2 ? wgrad_f32s_lds(128, 128, 16, 3) : 0;
}
void dvx_wgrad_f32s_tile(int cfg, int* bi, int* bj) {
  *bi = cfg == 0 ? 64 : 128;
  *bj = cfg == 0 ? 256 : 128;
}
void dvx_launch_wgrad_f32s(int cfg, const void* args, int grid, void* stream) {
  const WgradDmaArgs& aa = *static_cast<const WgradDmaArgs*>(args);
  hipStream_t s = (hipStream_t)stream;
  if (cfg == 0) hipLaunchKernelGGL((conv_wgrad_f32s_kernel<64, 256, 16, 3>), dim3(grid), dim3(256), 0, s, aa);
  else hipLaunchKernelGGL((conv_wgrad_f32s_kernel<128, 128, 16, 3>), dim3(grid), dim3(256), 0, s, aa);
}

// entry point for conv.hip (dv_conv3d_dgrad_bn): the reduce over what the data gradient `a` (ConvArgs) has written
void dvx_dgrad_bn_tail_launch(const void* conv_args, int is_f32, void* stream) {
  const ConvArgs& c = *static_cast<const ConvArgs*>(conv_args);
  BnTailArgs t;
  t.g = c.out; t.x = c.bn_x; t.mean = c.bn_mean; t.invstd = c.bn_invstd; t.scale = c.bn_scale; t.shift = c.bn_shift;
  t.sums = c.bn_sums; t.M = c.M; t.N = c.N; t.NP = c.NP; t.ldg = c.ldo; t.ldx = c.bn_ldx; t.n_rep = c.bn_rep; t.mask = c.bn_mask;
  const int grid = ((c.NP + 63) / 64) * ((c.M + 127) / 128);
  if (is_f32) hipLaunchKernelGGL((dgrad_bn_tail_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, t);
  else hipLaunchKernelGGL((dgrad_bn_tail_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, t);
}
