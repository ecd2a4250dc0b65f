/*
 * dualvar_hip.h -- C ABI of libdualvar_hip.so, the MI355X (gfx950) kernels underneath the
 * DualVar pretrain hot path.
 *
 * The reference (lzhangbj/DualVar) has no FFI of its own: its "operator API" for this path is
 * the set of torch calls made by backbone/{s3dg,r21d,r3d,resnet_2d3d}.py, model/simclr.py, model/moco.py and
 * pretrain.py:394-451.  Each entry point below replaces the library kernel behind one of those
 * calls (cited per function).  The Python host (dualvar_amd/) binds them with ctypes; a
 * maintainer of the reference would bind the same symbols.
 *
 * Conventions
 *   - plain C: pointers, sizes, PODs.  No torch types.
 *   - every pointer is a DEVICE pointer unless named host_*.  The caller owns every buffer;
 *     nothing is allocated, freed or synchronised inside.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - return value: 0 on success, a negative DV_E* code for a rejected argument (nothing is
 *     launched), a positive hipError_t if the launch itself failed.
 *   - activations are NDHWC ("channels last"): element (n,t,h,w,c) of a view lives at
 *     ptr[((n*T+t)*H+h)*W+w)*ld + c]; ld >= round_up(C,8) elements, and the lanes
 *     [C, round_up(C,8)) hold zeros (kernels that produce a view write those zeros).
 *     A channel slice of a wider buffer is a view with ptr advanced by the channel offset.
 *   - dtype: DV_F32 (parity mode) or DV_BF16 (storage bf16, fp32 accumulate).  Statistics,
 *     losses, gradients of parameters and optimizer state are always fp32.
 */
#ifndef DUALVAR_HIP_H
#define DUALVAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on every incompatible change of a signature or struct below; dv_abi_version() returns the value the library was
   built with and a consumer must refuse a library whose value differs from the header it was compiled against.
   2: dv_conv3d_wgrad (workspace, workspace_bytes), dv_bn_bwd_reduce (ws), dv_infonce_fwd (workspace, bytes),
      dv_augment_ingest (blur, blur_scratch) gained arguments; dv_bn_item grew by red_ws (round 2 of this build).
   (dv_conv3d_ksplit_cols, dv_conv3d_wgrad_bn, dv_conv3d_wgrad_bn_ok were ADDED under version 2: additions do not bump it.) */
#define DV_ABI_VERSION 2

enum { DV_F32 = 0, DV_BF16 = 1 };

enum {
  DV_OK = 0,
  DV_EINVAL = -1,      /* inconsistent shapes / unsupported parameter */
  DV_EALIGN = -2,      /* pointer or pitch not 16-byte aligned */
  DV_EUNSUPPORTED = -3
};

/* epilogue / behaviour flags */
enum {
  DV_BIAS = 1,         /* add bias[n]                                     */
  DV_RELU = 2,         /* max(.,0)                                        */
  DV_SIGMOID = 4,      /* 1/(1+exp(-.))                                   */
  DV_ACCUM = 8,        /* out += result (dgrad into a shared input)       */
  DV_STATS = 16,       /* conv fwd: also emit per-tile BatchNorm partials */
  DV_NO_RELU_MASK = 32, /* bn backward: activation was identity            */
  DV_MASK_FROM_X = 64,  /* bn backward (multi-tensor forms, no residual): recompute the ReLU mask from x with the item's
                           scale / shift -- relu(x*scale+shift) exactly as the forward -- instead of reading y */
  DV_W3 = 128           /* conv fwd / stride-1 dgrad, DV_F32: the weight pointer is the PRE-SPLIT layout of dv_pack_w3 */
};

int dv_abi_version(void);
/* Required device: gfx950.  Returns 0 when the current device can run the kernels. */
int dv_check_device(void);

/* ---------------------------------------------------------------------------------------
 * 3-D convolution as implicit GEMM on MFMA.  Replaces nn.Conv3d (bias=False) forward /
 * backward at backbone/s3dg.py:11,39,41  r21d.py:54,64  r3d.py:33  resnet_2d3d.py:11-29,124-173
 * and the 1x1x1 projection heads model/simclr.py:47-49,176-180, model/moco.py:58-60,284-308
 * (with DV_BIAS), and SelfGating's nn.Linear (s3dg.py:71) as a 1x1x1 conv over [N,1,1,1,C].
 *
 * Geometry: input [N,Ti,Hi,Wi,Cin] (pitch ldx), output [N,To,Ho,Wo,Cout] (pitch ldy).
 * Weights:  w_fwd  [Cout][kt*kh*kw][CinP]   (K-contiguous per output channel), CinP = cin_pitch
 *           w_dgrad[Cin ][kt*kh*kw][CoutP]  (made by dv_pack_dgrad_weights)
 *           dw     [Cout][kt*kh*kw][CinP]   fp32, same layout as the fp32 master weights
 * cin_pitch / cout_pitch are the zero-padded channel counts the gathers decode with
 * (multiples of 8, or 4 for the 3-channel network input).
 */
typedef struct dv_conv_desc {
  int32_t dtype;
  int32_t N, Ti, Hi, Wi, Cin;
  int32_t To, Ho, Wo, Cout;
  int32_t kt, kh, kw;
  int32_t st, sh, sw;
  int32_t pt, ph, pw;
  int32_t cin_pitch, cout_pitch;
  int32_t ldx, ldy;
  int32_t flags;
} dv_conv_desc;

/* number of M-tiles dv_conv3d_fwd emits BatchNorm partials for (rows of `stats`) and the rows per tile (128, or 64
 * for small problems) -- both are what dv_bn_reduce_stats / dv_bn_stats_finalize need */
int dv_conv3d_stat_tiles(const dv_conv_desc* d);
int dv_conv3d_tile_rows(const dv_conv_desc* d);
/* the GEMM tile (rows x columns) dv_conv3d_fwd (dgrad = 0) or dv_conv3d_dgrad (dgrad = 1) will use for this
 * problem: informational (profiling labels, grid size = ceil(M/rows) * ceil(Npitch/cols)) */
int dv_conv3d_tile_shape(const dv_conv_desc* d, int32_t dgrad, int32_t* rows, int32_t* cols);
/* informational: the column tile (32 / 64) of the "K split over the waves" kernel when dv_conv3d_fwd (dgrad = 0) / dv_conv3d_dgrad
 * (1) will run this problem on it (few rows, long K: the 1 152- and 12 544-row levels of S3D-G), else 0 */
int dv_conv3d_ksplit_cols(const dv_conv_desc* d, int32_t dgrad);
/* informational: 1 / 2 when dv_conv3d_fwd (dgrad = 0) / dv_conv3d_dgrad (1) will run this problem on the LDS-staged input-tile
 * kernel (csrc/conv_tap.hip) in its spatial (1x3x3) / temporal (3x1x1) form -- DV_F32 with DV_W3, stride 1, "same" padding,
 * channel pitch % 16 == 0, at least two rounds of 256-row x 64-column tiles: the separable pairs of backbone/s3dg.py:30-65 and
 * backbone/r21d.py:11-70 on the large maps -- else 0.  Its BatchNorm partials are per 256 rows (dv_conv3d_tile_rows).
 * 3 (dgrad = 0 only): the pixel-pair stem form of that kernel -- the RGB stem conv (backbone/s3dg.py:151) as a 1x7x4 window over
 * 8-channel pixel pairs, stride (1,2,1): tiles of G whole output lines (dv_conv3d_tile_rows = G * Wo, e.g. 224). */
int dv_conv3d_tap_kind(const dv_conv_desc* d, int32_t dgrad);
/* informational: rows per tile of that launch (256; 128 where the 256-row form would not fill the chip: the 12 544-row levels;
 * the pixel-pair stem form: G * Wo), 0 when dv_conv3d_tap_kind is 0.  For dgrad = 0 this is dv_conv3d_tile_rows. */
int dv_conv3d_tap_rows(const dv_conv_desc* d, int32_t dgrad);
/* y = conv(x, w) [+bias][act]; with DV_STATS also stats[2][Cout][tiles] = (sum, M2 about the
 * tile mean) of the values as stored.  */
int dv_conv3d_fwd(const dv_conv_desc* d, const void* x, const void* w_fwd, const float* bias,
                  void* y, float* stats, void* stream);
/* dx (+)= conv_transpose(dy, w).  strides must be 1 or 2. */
int dv_conv3d_dgrad(const dv_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, void* stream);
/* The same data gradient when dx is dL/dy of y = relu(BatchNorm(x_bn)) -- the layer in front of this conv
 * (backbone/s3dg.py:24-28,58-65: conv -> bn -> relu -> conv) -- and this conv is y's ONLY consumer: the epilogue also adds
 * the BatchNorm backward's two sums over the rows it writes, sums[tile % n_rep][0][c] += sum g, [1][c] += sum g*xhat with
 * g = dx masked by (x_bn*scale + shift > 0) (flags: DV_NO_RELU_MASK for a BatchNorm without ReLU), i.e. exactly what
 * dv_bn_bwd_reduce(DV_MASK_FROM_X) computes from a second read of dx; dv_bn_bwd_apply then reads `sums` as usual.
 * x_bn has d->dtype, the dims of dx and pitch ldx (elements); sums is [n_rep][2][cp8(Cin)] fp32, zeroed by the caller.
 * DV_ACCUM is not allowed (the sums must see the complete gradient). */
typedef struct dv_bn_reduce {
  const void* x;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  float* sums;
  int32_t ldx, n_rep, flags, _pad;
} dv_bn_reduce;
int dv_conv3d_dgrad_bn(const dv_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const dv_bn_reduce* bn,
                       void* stream);
/* The ORDERED form of the same fusion, on the LDS-staged input-tile kernel (dv_conv3d_tap_kind != 0: DV_F32 + DV_W3 stride-1
 * "same" 1x3x3 / 3x1x1 convs and the t-strided 7x1x1 stem conv): the two sums are formed from the accumulators and ONE read of
 * x_bn -- dv_bn_bwd_reduce's read of dL/dy disappears --, every 256-row tile stores its row of partial sums into `workspace` and
 * the workgroups that take the last tickets add the rows in tile order: no float atomics, two runs give the same bits.  The
 * result is ADDED to sums[0][2][cp8(Cin)] (n_rep is ignored; the caller zeroes `sums`).  dv_conv3d_dgrad_bn_workspace returns
 * the bytes `workspace` must hold (0: this problem does not run on that kernel -- use dv_bn_bwd_reduce); its ticket words must be
 * ZERO before the first call and are left zero (memset the buffer once; it may be shared by consecutive calls on one stream, of
 * any shapes: the tickets are the first 64 KiB of the buffer and no launch stores anything else there). */
int64_t dv_conv3d_dgrad_bn_workspace(const dv_conv_desc* d);
int dv_conv3d_dgrad_bn_ws(const dv_conv_desc* d, const void* dy, const void* w_dgrad, void* dx, const dv_bn_reduce* bn,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* dw += x^T * dy into the fp32 gradient arena (caller zeroes at zero_grad).  Deterministic: the rows are split over
 * workgroups whose partial tiles go to `workspace` ([splits][Cout][taps*CinP] fp32, plain stores) and are then added to dw
 * in a fixed order -- no float atomics, so two runs on the same inputs give the same bits.  dv_conv3d_wgrad_workspace
 * returns the bytes `workspace` must hold for this problem (0: none needed, workspace may be NULL); the buffer can be
 * shared by launches on one stream. */
int64_t dv_conv3d_wgrad_workspace(const dv_conv_desc* d);
/* the dW tile (output channels x im2col columns) and the number of row splits dv_conv3d_wgrad will use: informational */
int dv_conv3d_wgrad_tile(const dv_conv_desc* d, int32_t* rows, int32_t* cols, int32_t* splits);
int dv_conv3d_wgrad(const dv_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace,
                    int64_t workspace_bytes, void* stream);
/* The same weight gradient for a conv whose output feeds y = [relu](BatchNorm(conv)) and whose INPUT needs no gradient (the
 * first conv of a network: backbone/s3dg.py:151 conv1 on the clip): then dL/d(conv output) -- what dv_bn_bwd_apply would
 * write -- has this launch as its only reader, and it is formed on the fly instead: `g` is dL/dy (the BatchNorm's incoming
 * gradient), bn->x the conv's forward output (same dims and pitch as g), and the kernel multiplies
 *   k1*g' + k2*x + k3,   g' = g masked by (x*scale + shift > 0),   k1..k3 from the GLOBAL sums as in dv_bn_bwd_apply
 * with dv_bn_bwd_apply's own expression (bit-identical operand, bit-identical dw), and adds dgamma / dbeta.  One read of g and
 * x instead of a read of both, a write of dx and a read of dx.  fp32 split mode only (dv_conv3d_wgrad_bn_ok says whether
 * this problem runs on the kernel that carries it; otherwise DV_EUNSUPPORTED: call dv_bn_bwd_apply + dv_conv3d_wgrad). */
typedef struct dv_bn_bwd {
  const void* x;                 /* BatchNorm input = conv output, [M][ldx], d->dtype; ldx must equal d->ldy */
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* scale;            /* gamma*invstd, beta - mean*scale: the forward's affine map (ReLU mask) */
  const float* shift;
  const float* sums;             /* [n_rep][2][cp8(Cout)]: dv_bn_bwd_reduce's result (all-reduced over the ranks) */
  float* dgamma;                 /* += dparam_scale * sum g*xhat (NULL: skip both) */
  float* dbeta;
  float inv_count, dparam_scale; /* 1 / (rows over all ranks), 1 / ranks */
  int32_t ldx, n_rep, flags, _pad; /* flags: DV_NO_RELU_MASK */
} dv_bn_bwd;
int dv_conv3d_wgrad_bn_ok(const dv_conv_desc* d);
int dv_conv3d_wgrad_bn(const dv_conv_desc* d, const void* x, const void* g, float* dw, void* workspace,
                       int64_t workspace_bytes, const dv_bn_bwd* bn, void* stream);

/* BatchNorm ON LOAD: conv -> BatchNorm (+ReLU) -> conv chains (backbone/s3dg.py:30-65 STConv3d: conv1 -> bn1 -> relu -> conv2;
 * backbone/r21d.py:54-70) in which the second conv is the ONLY reader of y = [relu](x_bn * scale + shift).  Then y need not
 * exist in memory: the second conv's forward and its weight gradient read x_bn (the BatchNorm's input, i.e. the first conv's
 * raw output) and apply the affine map where their operand fragments are formed, with dv_bn_apply's expression
 * (x * scale + shift, then max(., 0) with DV_RELU in `flags`): same bits as the two-launch plan, one write and one read of the
 * activation less per pair.  The padding of the conv is a padding of y (zeros), not of x_bn.  scale / shift are the
 * [cp8(Cin)] arrays dv_bn_stats_finalize / dv_bn_finalize wrote.  The data gradient is unchanged (it never reads y); the
 * BatchNorm backward takes its ReLU mask from x_bn (DV_MASK_FROM_X).
 * dv_conv3d_bn_in_ok: 0 = this problem cannot run that way (then call dv_bn_apply and the plain entry points); 1 / 2 = both
 * dv_conv3d_fwd_bn_in and dv_conv3d_wgrad_bn_in take it (1: the LDS-staged temporal kernels, stride-1 3x1x1; 2: conv_gemm's
 * 256 x 64 tile, e.g. the 7x1x1 / stride-2 stem conv of backbone/s3dg.py:151).  DV_F32 with DV_W3, no bias / activation flags;
 * DV_STATS as in dv_conv3d_fwd.  Both return DV_EUNSUPPORTED where dv_conv3d_bn_in_ok says 0. */
typedef struct dv_bn_in {
  const float* scale;
  const float* shift;
  int32_t flags, _pad;           /* DV_RELU */
} dv_bn_in;
int dv_conv3d_bn_in_ok(const dv_conv_desc* d);
int dv_conv3d_fwd_bn_in(const dv_conv_desc* d, const void* x_bn, const dv_bn_in* bn, const void* w_fwd, void* y, float* stats,
                        void* stream);
int dv_conv3d_wgrad_bn_in(const dv_conv_desc* d, const void* x_bn, const dv_bn_in* bn, const void* dy, float* dw,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* ---- fp8 pointwise path (BASELINE configs[4]: "fp8 MFMA pointwise convs" of the 2D3D-ResNet-50 bottlenecks,
 * resnet_2d3d.py:130,167,173).  A 1x1x1 stride-1 conv is the GEMM Y[M, Cout] = X[M, Cin] W[Cout, Cin]^T; with OCP fp8
 * operands it runs on v_mfma_f32_32x32x64_f8f6f4 (fp32 accumulate, twice the bf16 rate per clock).  Operands are quantised
 * per TENSOR: dv_quantize_fp8 computes amax, scale = amax / 448 (e4m3; 57344 for e5m2) and q = fp8_rne(x / scale) into a
 * dense [M][ldq] byte tensor (channel pitch a multiple of 16); the GEMMs multiply the fp32 accumulators by the two scales
 * (device scalars: no host round trip).  Forward: x, w e4m3 -> y bf16 (+ DV_STATS partials like dv_conv3d_fwd); data
 * gradient: dy e5m2, w (dgrad layout [Cin][CoutP]) e4m3 -> dx bf16 (DV_ACCUM as dv_conv3d_dgrad).  The weight gradient stays
 * on the bf16 path (dv_conv3d_wgrad on the bf16 tensors): its reduction runs over up to 4e5 rows.
 * `d` describes the bf16 side (dtype DV_BF16, pitches of y / dx in elements); ldx / ldy of the fp8 operand are in BYTES. */
int dv_quantize_fp8_workspace(void);            /* bytes of `workspace` for dv_quantize_fp8 */
int dv_quantize_fp8(int32_t dtype, const void* x, int64_t M, int32_t C, int32_t ld, int32_t fmt /* 0 e4m3, 1 e5m2 */, void* q,
                    int32_t ldq, float* scale_out, float* workspace, void* stream);
int dv_conv3d_fwd_fp8(const dv_conv_desc* d, const void* x8, const void* w8, const float* scale_x, const float* scale_w,
                      void* y, float* stats, void* stream);
int dv_conv3d_dgrad_fp8(const dv_conv_desc* d, const void* dy8, const void* wd8, const float* scale_dy, const float* scale_w,
                        void* dx, void* stream);

/* ---- fp32 products on the bf16 matrix cores.  gfx950 has no TF32 path and its f32-input MFMA runs at 1/16 of the bf16 rate,
 * so DV_F32 convolutions split every fp32 operand EXACTLY into three bf16 (hi + mid + lo, 8 significant bits each) and
 * accumulate the six partial products of weight >= 2^-16 in fp32 (error at fp32 rounding level; DUALVAR_F32_EXACT=1 in the
 * environment selects exact-f32 MFMA kernels instead).  Activations are split inside the kernels; the WEIGHTS can be handed
 * over already split and in fragment order (flag DV_W3 on dv_conv3d_fwd / stride-1 dv_conv3d_dgrad), made by dv_pack_w3 for
 * many tensors in one launch: [K tile of 16][k half][rows padded to 128][hi|mid|lo][8] bf16 = dv_w3_bytes(rows, Ktot) bytes
 * per tensor (rows = Cout of the forward layout [Cout][Ktot], Cin of the dgrad layout [Cin][Ktot]). */
typedef struct dv_w3_desc {
  int64_t src_off;   /* element offset of the fp32 [rows][Ktot] tensor from `base` */
  int64_t dst_off;   /* BYTE offset of its split copy from `out_base` (multiple of 16) */
  int32_t N, Ktot;   /* rows, K */
} dv_w3_desc;
int64_t dv_w3_bytes(int32_t rows, int32_t ktot);
/* block_map (device): [n_blocks][2] = (descriptor, first 48-byte unit of the block); a block converts 256 units */
int dv_pack_w3(const float* base, void* out_base, const dv_w3_desc* descs /*device*/, const int32_t* block_map /*device*/,
               int32_t n_blocks, void* stream);

/* master fp32 [Cout][taps][CinP] -> compute-dtype [Cin][taps][CoutP] for n_desc tensors at once */
typedef struct dv_pack_desc {
  int64_t src_off;   /* element offset into the fp32 master arena */
  int64_t dst_off;   /* element offset into the dgrad-layout arena */
  int32_t Cout, Cin, taps, cin_pitch, cout_pitch, _pad;
} dv_pack_desc;
int dv_pack_dgrad_weights(int32_t dtype, const float* master, void* dst, const dv_pack_desc* descs /*device*/,
                          const int32_t* block_map /*device: [n_blocks][2] = (desc, first output row)*/,
                          int32_t n_blocks, void* stream);
/* fp32 -> compute dtype copy of a flat arena (n elements) */
int dv_cast_arena(int32_t dtype, const float* src, void* dst, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------
 * Input ingest: NCDHW fp32 clips -> NDHWC (C padded to 4), optional (x-mean)/std
 * (utils/transforms.py:57-63 via pretrain.py:386-389) and optional per-sample temporal
 * segment permutation (torch.gather at simclr.py:378-383 / moco.py:543-549): with perm != NULL
 * output frame t of sample n comes from frame perm[n][t / seg]*seg + t % seg.
 */
int dv_ingest_ncdhw(int32_t dtype, const float* x, void* y, int32_t N, int32_t C, int32_t T, int32_t H,
                    int32_t W, int64_t x_stride_n, int32_t ldy, const float* mean3, const float* istd3,
                    const int32_t* perm, int32_t n_seg, void* stream);
/* Same, into frames with a border of `pad` pixels on every side of H and W: y is [N*T][H+2*pad][W+2*pad][ldy] and
 * only the interior is written (the caller zeroes the buffer once).  With the zero padding of the RGB stem conv
 * (1x7x7 / 3x7x7, stride 2, padding 3: s3dg.py:137, r21d.py:201, r3d.py:116, resnet_2d3d.py:128) materialised like
 * this, pixel pairs of the bordered frame are an 8-channel tensor [.., H+6, (W+6)/2, 8] on which that conv is a dense
 * (kt x 7 x 4)-tap, stride (st, 2, 1), padding-free convolution over 16-byte vectors -- the fast gather path -- with
 * the kernel's rows stored 8 wide (tap kw = 7 is a structural zero: dv_fill_cols_f32 keeps its gradient zero). */
int dv_ingest_ncdhw_pad(int32_t dtype, const float* x, void* y, int32_t N, int32_t C, int32_t T, int32_t H,
                        int32_t W, int64_t x_stride_n, int32_t ldy, const float* mean3, const float* istd3,
                        const int32_t* perm, int32_t n_seg, int32_t pad, void* stream);
/* p[r][col0 .. col0+ncols) = value for r < rows (row pitch `pitch` floats) */
int dv_fill_cols_f32(float* p, int64_t rows, int32_t pitch, int32_t col0, int32_t ncols, float value, void* stream);

/* Augmenting ingest (replaces the CPU PIL/DataLoader pipeline of pretrain.py:491-564 with the
 * reference's own tensor-side definitions, utils/transforms.py:13-63,66-78,90-163,201-312): decoded uint8 frames
 * [n_src][Hs][Ws][3] -> for every output frame f = n*T + t one table row: source frame, crop window (resized to H x W
 * with bilinear / align_corners=False when its size differs), horizontal flip, then up to five colour ops in table
 * order on [0,1] floats -- brightness / contrast / saturation `clamp(f*x + (1-f)*ref)` with ref = 0 / mean luma of the
 * frame at that point / luma of the pixel, grayscale (x = luma), or the hue rotation RGB -> HSV -> h = (h + f) mod 1
 * -> RGB of utils/augmentation.py:26-106 (`adjust_hue_np`, without its uint8 re-quantisation) -- then Normalize, and the NDHWC store of
 * dv_ingest_ncdhw_pad (4th channel zero, optional zero border, optional segment shuffle of table rows).
 * `table` and `perm` are DEVICE arrays; indices and windows are clamped into the source, so a bad row cannot fault.
 * At most one contrast op per frame.  scratch: N*T floats (mean luma in front of the contrast op).  Two launches (three
 * with `blur`). */
enum { DV_AUG_NONE = 0, DV_AUG_BRIGHTNESS = 1, DV_AUG_CONTRAST = 2, DV_AUG_SATURATION = 3, DV_AUG_GRAY = 4, DV_AUG_HUE = 5 };
#define DV_AUG_MAX_OPS 5
typedef struct dv_aug_frame {
  int32_t src;                     /* index of the source frame */
  int32_t crop_i, crop_j;          /* top-left corner of the window in the source frame */
  int32_t crop_h, crop_w;          /* its size; == (H, W): plain crop, else bilinear resize to H x W */
  int32_t flip;                    /* 1: horizontal flip of the window */
  int32_t op[DV_AUG_MAX_OPS];      /* DV_AUG_*, applied in this order */
  float factor[DV_AUG_MAX_OPS];    /* blend ratio of op[k]; hue shift in turns for DV_AUG_HUE; unused for GRAY / NONE */
} dv_aug_frame;                    /* 64 bytes per row */
/* Gaussian blur of the finished (colour-jittered, re-quantised) frame, as the reference applies it through PIL
 * (utils/augmentation.py:706-721: ToPILImage -> ImageFilter.GaussianBlur(radius=sigma) -> ToTensor, one sigma per clip).
 * Pillow approximates the Gaussian by three passes of an extended box filter per axis in 8.24 fixed point
 * (src/libImaging/BoxBlur.c); a row carries that filter's integer parameters, derived from sigma on the host with Pillow's
 * own float32 arithmetic (dualvar_amd/utils/transforms.py: box_blur_params), so the device side is exact integer work.
 * ww == 0: the frame is not blurred. */
typedef struct dv_aug_blur {
  int32_t radius;                  /* integer part of the box radius */
  uint32_t ww, fw;                 /* 8.24 weights of the 2*radius+1 inner taps / of the two outer taps */
  int32_t _pad;
} dv_aug_blur;                     /* 16 bytes per row, rows parallel to `table` */
/* blur: NULL, or N*T rows; blur_scratch: N*T*H*W*3 bytes (needed when blur != NULL; H*W*2 bytes of LDS per workgroup) */
int dv_augment_ingest(int32_t dtype, const uint8_t* frames, int32_t n_src, int32_t Hs, int32_t Ws,
                      const dv_aug_frame* table, int32_t N, int32_t T, int32_t H, int32_t W, void* y, int32_t ldy,
                      int32_t pad, const float* mean3, const float* istd3, const int32_t* perm, int32_t n_seg,
                      float* scratch, const dv_aug_blur* blur, uint8_t* blur_scratch, void* stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm3d, training mode (nn.BatchNorm3d at s3dg.py:16,46-47, r21d.py:56,99,106,111,228, ...;
 * SyncBatchNorm math torch/nn/modules/_functions.py:39-200).
 * Forward is split so that the cross-rank exchange can sit between the two calls:
 *   dv_bn_reduce_stats : conv-epilogue partials [2][C][tiles] -> local (sum, M2, count) [2*C+1]
 *   dv_bn_finalize     : R ranks' (sum, M2, count) -> mean, invstd, scale=gamma*invstd,
 *                        shift=beta-mean*scale; running stats updated with momentum
 *                        (unbiased variance, PyTorch semantics).
 *   dv_bn_stats_finalize : both of the above in one launch for the single-rank case.
 *   dv_bn_apply        : y = act(x*scale + shift [+ residual]) into a (possibly sliced) view.
 * Backward:
 *   dv_bn_bwd_reduce   : g = dy*(y>0); sums[0][c] += sum(g), sums[1][c] += sum(g*xhat)  (block-reduced, then
 *                        one fp32 atomic per block and channel, block b into replica b % n_rep so that the
 *                        memory-side atomics do not serialise; the caller zeroes `sums` [n_rep][2][CP] first)
 *   dv_bn_bwd_apply    : dx = scale*(g - sum_g/M - xhat*sum_gx/M) with the GLOBAL sums over the M = R*M_local rows;
 *                        dgamma += dparam_scale*sum(g*xhat), dbeta += dparam_scale*sum(g).  dparam_scale = 1/R:
 *                        every rank then holds (1/R)*SUM_r local_r, which is exactly what DDP's averaging of
 *                        SyncBatchNorm's per-rank (local-sum) parameter gradients produces.
 *                        optional dres (+)= g for the residual branch.
 */
/* Per-channel fp32 arrays read by the apply / backward kernels (scale, shift, mean, invstd, gamma, sums) are
 * accessed with 16-byte loads: they must be 16-byte aligned and readable up to CP = round_up(C, 8) floats;
 * `sums` arrays are laid out [2][CP]. */
/* partials: [2][pitch][n_tiles] (a channel's tiles are contiguous, so the one-workgroup-per-channel reduction reads
 * coalesced); the pointer is at this layer's first channel, i.e. base + first_channel * n_tiles (pitch > C when
 * several convolutions that share their input were run as one GEMM) */
int dv_bn_reduce_stats(const float* partials, int32_t n_tiles, int32_t tile_rows, int32_t pitch, int64_t M, int32_t C,
                       float* local_stats /*[2*C+1]*/, void* stream);
int dv_bn_stats_finalize(const float* partials, int32_t n_tiles, int32_t tile_rows, int32_t pitch, int64_t M, int32_t C,
                         float* local_stats /*[2*C+1]*/, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                         float* scale, float* shift, void* stream);
/* Eval mode (module.eval(): nn.BatchNorm3d with running statistics, classifier.py's test / retrieval passes and the
 * 'last'-layer finetune): the per-channel affine map of dv_bn_apply from the running statistics,
 * scale = gamma * rsqrt(running_var + eps), shift = beta - running_mean * scale, written up to round_up(C, 8). */
/* partials [2][C][1] = (sum, M2 about the mean) of a small fp32 [M][C] matrix: train-mode BatchNorm1d of the classifier
 * head (model/classifier.py:29-32) then runs through dv_bn_stats_finalize / dv_bn_apply / dv_bn_bwd_* like any other */
int dv_bn_rows_partials_f32(const float* x, int32_t ldx, int32_t M, int32_t C, float* partials, void* stream);
/* y[i] += alpha * a[i] * (b ? b[i] : 1): per-channel fix-ups for a conv BIAS in front of a BatchNorm (c3d.py:15-47) --
 * train mode: the bias cancels in the normalised output and only moves the running mean (running_mean += momentum*bias);
 * eval mode: shift += scale * bias */
int dv_addcmul_f32(float* y, const float* a, const float* b, float alpha, int32_t n, void* stream);
int dv_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int32_t C, float* scale, float* shift, void* stream);
int dv_bn_finalize(const float* stats /*[R] rows of (sum[C], M2[C], count), row pitch `stride` floats*/, int32_t R,
                   int32_t stride, int32_t C, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                   float* mean, float* invstd, float* scale, float* shift, void* stream);
int dv_bn_apply(int32_t dtype, const void* x, int32_t ldx, const float* scale, const float* shift,
                const void* residual, int32_t ldr, void* y, int32_t ldy, int64_t M, int32_t C,
                int32_t flags, void* stream);
/* Multi-tensor forms: the independent BatchNorm layers of one group (e.g. the four branch-entry BNs of an Inception
 * block) in ONE launch per phase.  `items` is a DEVICE array; blk_* are running block-count prefixes per phase
 * (stats: C blocks per item; apply / bwd_apply: ceil(M*CP/V/256) capped; bwd_reduce: dv_bn_bwd_blocks). */
typedef struct dv_bn_item {
  const float* partials; float* local_stats; const float* gamma; const float* beta;
  float* running_mean; float* running_var; float* mean; float* invstd; float* scale; float* shift;
  const void* x; const void* residual; void* y; const void* dy; void* dx; void* dres;
  float* sums; float* dgamma; float* dbeta;
  int64_t M;
  int32_t n_tiles, tile_rows, pitch, C, ldx, ldr, ldy, lddy, lddx, lddres;
  int32_t fwd_flags, bwd_flags, n_rep;
  float eps, momentum, inv_count, dparam_scale;
  int32_t blk_stats, blk_apply, blk_red, blk_bapply;
  float* red_ws;   /* ordered backward reduce: dv_bn_bwd_reduce_workspace(M, C) bytes (see dv_bn_bwd_reduce), or NULL */
} dv_bn_item;
int dv_bn_stats_multi(const dv_bn_item* items, int32_t n, int32_t finalize, int32_t total_blocks, void* stream);
/* dv_bn_finalize for every member of the group in one launch (multi-rank step, after the all-gather): item i's rows of the
 * gathered [R][stride] table start at the offset (items[i].local_stats - local_base); total_blocks = sum ceil(C_i / 128) */
int dv_bn_finalize_multi(const dv_bn_item* items, int32_t n, int32_t total_blocks, const float* local_base,
                         const float* gathered, int32_t R, int32_t stride, void* stream);
int dv_bn_apply_multi(int32_t dtype, const dv_bn_item* items, int32_t n, int32_t total_blocks, void* stream);
int dv_bn_bwd_reduce_multi(int32_t dtype, const dv_bn_item* items, int32_t n, int32_t total_blocks, void* stream);
int dv_bn_bwd_apply_multi(int32_t dtype, const dv_bn_item* items, int32_t n, int32_t total_blocks, int32_t max_c,
                          void* stream);
int dv_bn_bwd_blocks(int64_t M, int32_t C);
/* sums[2][CP] = (sum g, sum g*xhat).  ORDERED form (ws != NULL): every block stores its partial sums to ws, the block that
 * arrives last adds them in block order and WRITES sums[0 .. 2*CP) -- no float atomics, the result does not depend on the
 * order blocks finish in; n_rep is ignored and dv_bn_bwd_apply takes n_rep = 1.  ws: dv_bn_bwd_reduce_workspace(M, C) bytes
 * ([blocks][2][CP] partials, one row per group of 32 blocks, ticket words), ZERO before the first use (the kernel leaves the
 * tickets zero).
 * Atomic form (ws == NULL): sums[n_rep][2][CP] pre-zeroed by the caller, accumulated with float atomics over n_rep replicas. */
int64_t dv_bn_bwd_reduce_workspace(int64_t M, int32_t C);
int dv_bn_bwd_reduce(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                     int32_t ldx, const float* mean, const float* invstd, int64_t M, int32_t C,
                     int32_t flags, float* sums, int32_t n_rep, float* ws, void* stream);
int dv_bn_bwd_apply(int32_t dtype, const void* dy, int32_t lddy, const void* y, int32_t ldy, const void* x,
                    int32_t ldx, const float* mean, const float* invstd, const float* gamma,
                    const float* sums /*[n_rep][2][CP], global (all-reduced) sums*/, int32_t n_rep, float inv_count,
                    float dparam_scale, float* dgamma, float* dbeta, void* dx, int32_t lddx, void* dres, int32_t lddres,
                    int64_t M, int32_t C, int32_t flags, void* stream);

/* ---------------------------------------------------------------------------------------
 * MaxPool3d (s3dg.py:105,151,162,173,190; resnet_2d3d.py:212,280): -inf padding, first maximum
 * in (t,h,w) scan order wins (PyTorch CPU semantics).  idx holds the winning tap per element.
 */
typedef struct dv_pool_desc {
  int32_t dtype;
  int32_t N, Ti, Hi, Wi, C;
  int32_t To, Ho, Wo;
  int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
  int32_t ldx, ldy;
} dv_pool_desc;
int dv_maxpool3d_fwd(const dv_pool_desc* d, const void* x, void* y, uint8_t* idx, void* stream);
/* dx (+)= scatter of dy through idx (gather formulation, deterministic); flags: DV_ACCUM */
int dv_maxpool3d_bwd(const dv_pool_desc* d, const void* dy, const uint8_t* idx, void* dx, int32_t flags,
                     void* stream);
/* BatchNorm + ReLU + MaxPool3d in one pass each way, for a pool that is the only consumer of y = relu(x*scale + shift)
 * (the stems: s3dg.py:138-151, resnet_2d3d.py:128-131): y is never materialised.  d describes the pool (d->ldx = pitch of
 * the conv output x that the BatchNorm normalises, d->ldy = pitch of the pooled tensor).  Forward: pooled values and tap
 * indices bit-identical to dv_bn_apply (DV_RELU) followed by dv_maxpool3d_fwd.  Backward: the two passes of the BatchNorm
 * backward (dv_bn_bwd_reduce / dv_bn_bwd_apply with DV_MASK_FROM_X) with dL/dy gathered from the pooled gradient through
 * idx on the fly. */
int dv_bn_apply_maxpool(const dv_pool_desc* d, const void* x, const float* scale, const float* shift, void* y, uint8_t* idx,
                        void* stream);
int dv_bn_bwd_reduce_maxpool(const dv_pool_desc* d, const void* dy_pool, const uint8_t* idx, const void* x, const float* mean,
                             const float* invstd, const float* scale, const float* shift, float* sums, int32_t n_rep,
                             void* stream);
int dv_bn_bwd_apply_maxpool(const dv_pool_desc* d, const void* dy_pool, const uint8_t* idx, const void* x, const float* mean,
                            const float* invstd, const float* gamma, const float* scale, const float* shift,
                            const float* sums_global, int32_t rep_global, float inv_count, float dparam_scale, float* dgamma,
                            float* dbeta, void* dx, int32_t lddx, void* stream);

/* ---------------------------------------------------------------------------------------
 * Spatio-temporal mean and broadcast scaling: nn.AdaptiveAvgPool3d((1,1,1)) (simclr.py:44,166,
 * moco.py:55,66) and SelfGating (s3dg.py:68-78).
 *   dv_spatial_mean      : out[n][c] = mean_s x[n][s][c]                      (fp32 out, pitch C)
 *   dv_spatial_mean_bwd  : dx[n][s][c] (+)= dout[n][c]/S
 *   dv_gate_scale        : y[n][s][c] = x[n][s][c]*g[n][c]
 *   dv_gate_bwd_reduce   : dpre[n][c] = (sum_s dy*x) * g*(1-g); with x_is_output the second operand is the
 *                          gated output x*g (in-place gating) and dpre = (sum_s dy*out) * (1-g)
 *   dv_gate_bwd_apply    : dx[n][s][c] (+)= dy*g[n][c] + dmean[n][c]/S
 */
int dv_spatial_mean(int32_t dtype, const void* x, int32_t ldx, int32_t N, int32_t S, int32_t C, float* out,
                    void* stream);
int dv_spatial_mean_bwd(int32_t dtype, const float* dout, int32_t N, int32_t S, int32_t C, void* dx,
                        int32_t lddx, int32_t flags, void* stream);
int dv_gate_scale(int32_t dtype, const void* x, int32_t ldx, const float* g, int32_t N, int32_t S, int32_t C,
                  void* y, int32_t ldy, void* stream);
int dv_gate_bwd_reduce(int32_t dtype, const void* dy, int32_t lddy, const void* x, int32_t ldx, const float* g,
                       int32_t N, int32_t S, int32_t C, float* dpre, int32_t x_is_output, void* stream);
int dv_gate_bwd_apply(int32_t dtype, const void* dy, int32_t lddy, const float* g, const float* dmean,
                      int32_t N, int32_t S, int32_t C, void* dx, int32_t lddx, int32_t flags, void* stream);

/* ---------------------------------------------------------------------------------------
 * Small fp32 helpers for the projection heads and parameter gradients.
 *   dv_colsum_f32  : out[c] += sum_r x[r][c]                 (bias gradients)
 *   dv_l2norm_fwd  : y = x / max(||x||_2, eps) over the last dim (F.normalize, simclr.py:117,359,367,393)
 *   dv_l2norm_bwd  : dx = (dy - y*<dy,y>) / max(||x||,eps)
 *   dv_relu_bwd_f32: dx = dy * (y > 0)
 */
int dv_colsum_f32(const float* x, int32_t ldx, int32_t R, int32_t C, float* out, void* stream);
int dv_l2norm_fwd(const float* x, int32_t R, int32_t D, float eps, float* y, float* norm, void* stream);
int dv_l2norm_bwd(const float* dy, const float* y, const float* norm, int32_t R, int32_t D, float* dx, void* stream);
int dv_relu_bwd_f32(const float* dy, const float* y, int64_t n, float* dx, void* stream);

/* ---------------------------------------------------------------------------------------
 * Contrastive objectives, fused similarity (fp32 MFMA) + masked log-softmax cross-entropy.
 *
 * dv_ntxent_fwd: NT-Xent over the gathered negative set, simclr.py:56-99,183-229 (clip head:
 *   rows = all 2N view-major features) and simclr.py:280-337 (tc head: rows = this rank's 2B
 *   entries; features are the series-mean vectors).
 *     rows  [R][D] fp32, row r is global (view-major) index row_index0 + (r / n_local)*N + r % n_local
 *     cols  [2N][D] fp32 view-major
 *     logits[R][2N-1]   = [positive, negatives in column order] * inv_T   (reference layout)
 *     loss_rows[R]      = logsumexp(logits) - logits[:,0]
 *     rank0[R]          = number of negatives whose logit is > the positive's (top-k accuracy:
 *                         utils/utils.py:75-92; hit@k <=> rank0 < k)
 *     dsim  [R][2N]     = d(mean_r loss)/d sim[r][c]  (softmax - onehot) * inv_T / R, 0 at self
 * dv_infonce_fwd: MoCo InfoNCE against a queue, moco.py:222-229,404-438:
 *     logits[B][1+K] = [q.k, q.queue] * inv_T, queue stored [D][K] as in the reference.
 *     dlogits[B][1+K] = d(mean loss)/d(unscaled dot products);  dq[B][D] = d(mean loss)/dq.
 * dv_rank_margin: shuffle-rank margin loss simclr.py:231-278 / moco.py:440-480 on
 *     feats[Bn][2 views][s][D] (view-major per sample): loss = w*mean softplus(min(z,clip)),
 *     z=(other-highest)/theta; also margin logits [Bn*2s][2s-1] and dfeats.
 */
int dv_ntxent_fwd(const float* rows, const float* cols, int32_t R, int32_t n_local, int32_t N, int32_t D,
                  int32_t row_index0, float inv_T, float* logits, float* loss_rows, int32_t* rank0,
                  float* dsim, void* stream);
/* workspace (optional): dv_infonce_workspace(B, D, K) bytes, ZERO before the first use (the kernel leaves its ticket words
 * zero).  With it the K-split dq = dlogits . queue^T sums its slices in slice order (the last workgroup of a tile adds the
 * partial tiles): no float atomics, run-to-run identical bits.  NULL: fp32 atomics. */
int64_t dv_infonce_workspace(int32_t B, int32_t D, int32_t K);
int dv_infonce_fwd(const float* q, const float* k, const float* queue, int32_t B, int32_t D, int32_t K,
                   float inv_T, float* logits, float* loss_rows, int32_t* rank0, float* dlogits /*[B][1+K]*/,
                   float* dq, float* workspace, int64_t workspace_bytes, void* stream);
/* NN-retrieval score (classifier.py:964-981, torch.topk over the test x train similarity): rank[r] = number of train
 * samples scoring strictly above test row r's best same-label train sample (n_train if the label is absent); the k-NN
 * accuracy of the reference is mean(rank < k), for every k from one pass */
int dv_knn_rank(const float* sim, int32_t ld, int32_t R, int32_t n_train, const int32_t* train_labels,
                const int32_t* test_labels, int32_t* rank, void* stream);
/* nn.CrossEntropyLoss (mean) over integer targets, the downstream classifier's criterion (classifier.py:330,465):
 * loss_rows[r] = logsumexp(logits[r]) - logits[r][labels[r]], dlogits (optional) = (softmax - onehot) / R,
 * rank0 (optional) = classes scoring above the target (top-k accuracy without a sort) */
int dv_softmax_ce_fwd(const float* logits, int32_t ld, int32_t R, int32_t K, const int32_t* labels, float* loss_rows,
                      float* dlogits, int32_t ldd, int32_t* rank0, void* stream);
/* probs[r][:] = softmax(logits[r][:K])  (F.softmax(logit, dim=-1), classifier.py:716: the 10-clip test averages these) */
int dv_softmax_rows_f32(const float* logits, int32_t ld, int32_t R, int32_t K, float* probs, int32_t ldp, void* stream);
int dv_rank_margin(const float* feats /*[Bn][2s][D]*/, int32_t Bn, int32_t s, int32_t D, float theta,
                   float clip /*<=0: none*/, float weight, float* logits, float* loss /*[1]*/, float* dfeats,
                   float* scratch /*[Bn]*/, void* stream);
/* Strided fp32 GEMM on v_mfma_f32_32x32x2_f32 for the small head / loss products:
 *   C[m][n] (+)= alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]      (C row pitch ldc) */
int dv_gemm_f32(int32_t M, int32_t N, int32_t K, const float* A, int64_t sam, int64_t sak, const float* B,
                int64_t sbk, int64_t sbn, float* C, int64_t ldc, float alpha, int32_t accumulate, void* stream);
/* Several such GEMMs in one launch (the four self-gating FCs of an Inception block, forward and backward).
 * `descs` is a DEVICE array; tile_end is the running sum of ceil(M/32)*ceil(N/32) over the groups.
 *   C = act(alpha*A.B + bias [+ C])   flags: DV_ACCUM, DV_SIGMOID, DV_RELU */
typedef struct dv_gemm_desc {
  const float* A; const float* B; float* C; const float* bias;
  int64_t sam, sak, sbk, sbn, ldc;
  int32_t M, N, K, flags, tile_end;
  float alpha;
} dv_gemm_desc;
int dv_gemm_f32_grouped(const dv_gemm_desc* descs /*device*/, int32_t n_groups, int32_t total_tiles, void* stream);
/* ONE such GEMM with its descriptor in HOST memory (passed to the kernel by value; tile_end is ignored): the heads'
 * Linear / 1x1x1-conv forward with bias and ReLU in the epilogue (simclr.py:45-50,167-180; model/classifier.py:49-62) */
int dv_gemm_f32_ex(const dv_gemm_desc* desc /*host*/, void* stream);
/* y[r][d] = mean_g x[r][g][d]  (series-mean vectors of the tc head, simclr.py:297-304 ==
 * <mean_i row_i, mean_j col_j>);  bwd: dx[r][g][d] = dy[r][d]/G */
int dv_group_mean_f32(const float* x, int32_t R, int32_t G, int32_t D, float* y, void* stream);
int dv_group_mean_bwd_f32(const float* dy, int32_t R, int32_t G, int32_t D, float* dx, void* stream);
/* out[0] = mean(x[0..n)) */
int dv_mean_f32(const float* x, int32_t n, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer / momentum encoder over flat fp32 arenas (pretrain.py:272,451: SGD momentum 0.9 with
 * weight decay on every tensor; moco.py:104-107,329-334: k = m*k + (1-m)*q).
 *   buf = mu*buf + (g*grad_scale + wd*p);  p -= lr*buf;  optional compute-dtype copy of p.
 */
int dv_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float mu, float wd,
                    float grad_scale, int32_t copy_dtype, void* p_copy, void* stream);
int dv_ema(float* k, const float* q, int64_t n, float m, int32_t copy_dtype, void* k_copy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DUALVAR_HIP_H */
