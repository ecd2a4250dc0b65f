// Augmenting ingest: decoded uint8 frames -> crop / resize / flip / colour jitter / grayscale / Gaussian blur / Normalize ->
// NDHWC activations of the RGB stem, in two launches (three with blurred frames).  The arithmetic follows the reference's tensor-side
// definitions, utils/transforms.py:13-31 (crop, hflip), :33-42 (bilinear resize, align_corners=False), :49-51 (/255),
// :57-63 (normalize), :66-78 (luma), :90-163 (brightness / contrast / saturation blends with clamp to [0, 1]).
#include "common.hpp"

namespace {

constexpr int kThreads = 256;

struct AugArgs {
  const uint8_t* frames;            // [n_src][Hs][Ws][3]
  const dv_aug_frame* tab;          // [F]
  const int* perm;                  // optional segment shuffle [N][n_seg] (simclr.py:378-383)
  const float* mean3;
  const float* istd3;
  float* cmean;                     // [F] mean luma in front of the contrast op
  const dv_aug_blur* blur;          // optional [F]: Gaussian blur of the finished frame (ww == 0: none)
  uint8_t* u8tmp;                   // [F][H][W][3] quantised frames in front of the blur
  int n_src, Hs, Ws, F, T, n_seg, H, W, ldy, pad, Hp, Wp;
};

struct Rgb { float r, g, b; };

__device__ __forceinline__ float luma(const Rgb& p) { return 0.2989f * p.r + 0.5870f * p.g + 0.1140f * p.b; }
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__device__ __forceinline__ Rgb load_px(const uint8_t* __restrict__ fr, int Ws, int h, int w) {
  const uint8_t* p = fr + ((size_t)h * Ws + w) * 3;
  return Rgb{(float)p[0] / 255.f, (float)p[1] / 255.f, (float)p[2] / 255.f};
}

// table row of output frame f (the segment shuffle moves whole segments of a clip)
__device__ __forceinline__ int table_row(const AugArgs& a, int f) {
  if (!a.perm) return f;
  const int n = f / a.T, t = f - n * a.T, seg = a.T / a.n_seg;
  const int sg = min(max(a.perm[n * a.n_seg + t / seg], 0), a.n_seg - 1);      // a bad permutation entry cannot leave the table
  return n * a.T + sg * seg + t % seg;
}

// pixel (hh, ww) of the H x W output window of a frame, before the colour ops
__device__ __forceinline__ Rgb sample(const AugArgs& a, const dv_aug_frame& q, int hh, int ww) {
  const int src = min(max(q.src, 0), a.n_src - 1);
  const uint8_t* fr = a.frames + (size_t)src * a.Hs * a.Ws * 3;
  const int ch = min(max(q.crop_h, 1), a.Hs), cw = min(max(q.crop_w, 1), a.Ws);
  const int ci = min(max(q.crop_i, 0), a.Hs - ch), cj = min(max(q.crop_j, 0), a.Ws - cw);
  if (q.flip) ww = a.W - 1 - ww;
  if (ch == a.H && cw == a.W) return load_px(fr, a.Ws, ci + hh, cj + ww);
  // F.interpolate(mode='bilinear', align_corners=False) of the crop window (transforms.py:33-42,245-246)
  const float sh = (float)ch / (float)a.H, sw = (float)cw / (float)a.W;
  const float fy = fmaxf(sh * ((float)hh + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)ww + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)fy, ch - 1), x0 = min((int)fx, cw - 1);
  const int y1 = y0 + (y0 < ch - 1 ? 1 : 0), x1 = x0 + (x0 < cw - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0, my = 1.f - ly, mx = 1.f - lx;
  const Rgb p00 = load_px(fr, a.Ws, ci + y0, cj + x0), p01 = load_px(fr, a.Ws, ci + y0, cj + x1);
  const Rgb p10 = load_px(fr, a.Ws, ci + y1, cj + x0), p11 = load_px(fr, a.Ws, ci + y1, cj + x1);
  Rgb o;
  o.r = my * (mx * p00.r + lx * p01.r) + ly * (mx * p10.r + lx * p11.r);
  o.g = my * (mx * p00.g + lx * p01.g) + ly * (mx * p10.g + lx * p11.g);
  o.b = my * (mx * p00.b + lx * p01.b) + ly * (mx * p10.b + lx * p11.b);
  return o;
}

// colour ops op[from .. to) of a frame; `cm` = mean luma for the contrast op
__device__ __forceinline__ Rgb colour_ops(Rgb p, const dv_aug_frame& q, int to, float cm) {
  for (int k = 0; k < to; ++k) {
    const float f = q.factor[k], g = 1.f - f;
    switch (q.op[k]) {
      case DV_AUG_BRIGHTNESS:                                   // _blend(vid, 0, f)
        p.r = clamp01(f * p.r + g * 0.f); p.g = clamp01(f * p.g + g * 0.f); p.b = clamp01(f * p.b + g * 0.f);
        break;
      case DV_AUG_CONTRAST:                                     // _blend(vid, mean luma of the frame, f)
        p.r = clamp01(f * p.r + g * cm); p.g = clamp01(f * p.g + g * cm); p.b = clamp01(f * p.b + g * cm);
        break;
      case DV_AUG_SATURATION: {                                 // _blend(vid, luma, f)
        const float l = luma(p);
        p.r = clamp01(f * p.r + g * l); p.g = clamp01(f * p.g + g * l); p.b = clamp01(f * p.b + g * l);
        break;
      }
      case DV_AUG_GRAY: {                                       // gray * mask + vid * (1 - mask), mask = 1
        const float l = luma(p);
        p.r = p.g = p.b = l;
        break;
      }
      case DV_AUG_HUE: {                                        // utils/augmentation.py:26-106 (_rgb2hsv_np, _hsv2rgb_np)
        const float maxc = fmaxf(p.r, fmaxf(p.g, p.b)), minc = fminf(p.r, fminf(p.g, p.b));
        const bool eqc = maxc == minc;
        const float cr = maxc - minc;
        const float s = cr / (eqc ? 1.f : maxc);
        const float cd = eqc ? 1.f : cr;
        const float rc = (maxc - p.r) / cd, gc = (maxc - p.g) / cd, bc = (maxc - p.b) / cd;
        const float hr = (maxc == p.r) ? (bc - gc) : 0.f;
        const float hg = (maxc == p.g && maxc != p.r) ? (2.0f + rc - bc) : 0.f;
        const float hb = (maxc != p.g && maxc != p.r) ? (4.0f + gc - rc) : 0.f;
        float hh = fmodf((hr + hg + hb) / 6.0f + 1.0f, 1.0f);
        hh = hh + f;
        hh = hh - floorf(hh);                                   // numpy's float `% 1.0`
        const float h6 = hh * 6.0f, fi = floorf(h6), ff = h6 - fi;
        const float v = maxc;
        const float pp = clamp01(v * (1.0f - s)), qq = clamp01(v * (1.0f - s * ff)), tt = clamp01(v * (1.0f - s * (1.0f - ff)));
        const int i = ((int)fi) % 6;
        p.r = i == 0 ? v : i == 1 ? qq : i == 2 ? pp : i == 3 ? pp : i == 4 ? tt : v;
        p.g = i == 0 ? tt : i == 1 ? v : i == 2 ? v : i == 3 ? qq : i == 4 ? pp : pp;
        p.b = i == 0 ? pp : i == 1 ? pp : i == 2 ? tt : i == 3 ? v : i == 4 ? v : qq;
        break;
      }
      default: break;
    }
  }
  return p;
}

__device__ __forceinline__ int contrast_pos(const dv_aug_frame& q) {
  for (int k = 0; k < DV_AUG_MAX_OPS; ++k)
    if (q.op[k] == DV_AUG_CONTRAST) return k;
  return -1;
}

// one workgroup per output frame: mean luma of the window as the contrast op sees it (fixed summation order)
__global__ void __launch_bounds__(kThreads) aug_contrast_mean_kernel(AugArgs a) {
  const int f = blockIdx.x;
  const dv_aug_frame q = a.tab[table_row(a, f)];
  const int kc = contrast_pos(q);
  if (kc < 0) return;
  __shared__ float sh[kThreads / 64];
  float s = 0.f;
  const int px = a.H * a.W;
  for (int i = threadIdx.x; i < px; i += kThreads) {
    const int hh = i / a.W, ww = i - hh * a.W;
    s += luma(colour_ops(sample(a, q, hh, ww), q, kc, 0.f));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < kThreads / 64; ++i) t += sh[i];
    a.cmean[f] = t / (float)px;
  }
}

template <typename T>
__global__ void __launch_bounds__(kThreads) aug_apply_kernel(AugArgs a, T* __restrict__ y) {
  const int64_t px = (int64_t)a.H * a.W, total = px * a.F;
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
    const int f = (int)(i / px);
    const int hw = (int)(i - (int64_t)f * px);
    const int hh = hw / a.W, ww = hw - hh * a.W;
    const dv_aug_frame q = a.tab[table_row(a, f)];
    Rgb p = colour_ops(sample(a, q, hh, ww), q, DV_AUG_MAX_OPS, contrast_pos(q) >= 0 ? a.cmean[f] : 0.f);
    if (a.blur && a.blur[table_row(a, f)].ww != 0) {
      // this frame goes through the blur: ToPILImage() of the float frame = mul(255).byte() (truncation), utils/augmentation.py:719
      uint8_t* u = a.u8tmp + ((int64_t)f * px + hw) * 3;
      u[0] = (uint8_t)(p.r * 255.f); u[1] = (uint8_t)(p.g * 255.f); u[2] = (uint8_t)(p.b * 255.f);
      continue;
    }
    if (a.mean3) {
      p.r = (p.r - a.mean3[0]) * a.istd3[0];
      p.g = (p.g - a.mean3[1]) * a.istd3[1];
      p.b = (p.b - a.mean3[2]) * a.istd3[2];
    }
    T* dst = y + (((int64_t)f * a.Hp + hh + a.pad) * a.Wp + ww + a.pad) * a.ldy;
    if (sizeof(T) == 4) {
      f32x4 o = {p.r, p.g, p.b, 0.f};
      *reinterpret_cast<f32x4*>(dst) = o;
    } else {
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      bf16x4 o = {(bf16_t)p.r, (bf16_t)p.g, (bf16_t)p.b, (bf16_t)0.f};
      *reinterpret_cast<bf16x4*>(dst) = o;
    }
  }
}

// Gaussian blur as PIL's ImageFilter.GaussianBlur does it (utils/augmentation.py:706-721 calls it on the uint8 frame): three
// horizontal then three vertical passes of an "extended box filter" in 8.24 fixed point (Pillow, src/libImaging/BoxBlur.c:
// out[x] = (ww * sum_{|d| <= r} in[x+d] + fw * (in[x-r-1] + in[x+r+1]) + 2^23) >> 24, indices clamped to the line, the result
// of every pass re-quantised to uint8).  r, ww, fw come from the host (dualvar_amd/utils/transforms.py restates Pillow's
// float32 radius arithmetic), so the kernel is integer only: bit-exact with PIL.  One workgroup per (frame, channel), the
// plane ping-pongs between two LDS images; then ToTensor (/255), Normalize and the store into the stem's NDHWC input.
template <typename T>
__global__ void __launch_bounds__(kThreads) aug_blur_kernel(AugArgs a, T* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_blur[];
  const int f = blockIdx.x / 3, c = blockIdx.x - f * 3;
  const dv_aug_blur b = a.blur[table_row(a, f)];
  if (b.ww == 0) return;
  const int H = a.H, W = a.W, px = H * W, r = b.radius;
  uint8_t* cur = lds_blur;
  uint8_t* nxt = lds_blur + px;
  const uint8_t* src = a.u8tmp + (int64_t)f * px * 3 + c;
  for (int i = threadIdx.x; i < px; i += kThreads) cur[i] = src[(int64_t)i * 3];
  __syncthreads();
  for (int pass = 0; pass < 6; ++pass) {
    const bool horiz = pass < 3;
    const int n = horiz ? W : H, stride = horiz ? 1 : W;
    for (int i = threadIdx.x; i < px; i += kThreads) {
      const int hh = i / W, wx = i - hh * W;
      const int x = horiz ? wx : hh;
      const uint8_t* line = cur + (horiz ? hh * W : wx);
      uint32_t acc = 0;
      for (int d = -r; d <= r; ++d) acc += line[min(max(x + d, 0), n - 1) * stride];
      const uint32_t far = (uint32_t)line[min(max(x - r - 1, 0), n - 1) * stride] + (uint32_t)line[min(max(x + r + 1, 0), n - 1) * stride];
      const uint32_t bulk = acc * b.ww + far * b.fw;                       // (mod 2^32, as the UINT32 arithmetic of BoxBlur.c)
      nxt[i] = (uint8_t)((bulk + (1u << 23)) >> 24);
    }
    __syncthreads();
    uint8_t* t = cur; cur = nxt; nxt = t;
  }
  const float m = a.mean3 ? a.mean3[c] : 0.f, is = a.mean3 ? a.istd3[c] : 1.f;
  for (int i = threadIdx.x; i < px; i += kThreads) {
    const int hh = i / W, wx = i - hh * W;
    float v = (float)cur[i] / 255.f;                                        // ToTensor()
    if (a.mean3) v = (v - m) * is;
    y[(((int64_t)f * a.Hp + hh + a.pad) * a.Wp + wx + a.pad) * a.ldy + c] = DT<T>::from_f(v);
  }
}

}  // namespace

extern "C" int dv_augment_ingest(int32_t dtype, const uint8_t* frames, int32_t n_src, int32_t Hs, int32_t Ws,
                                 const dv_aug_frame* table, int32_t N, int32_t T_, int32_t H, int32_t W, void* y, int32_t ldy,
                                 int32_t pad, const float* mean3, const float* istd3, const int32_t* perm, int32_t n_seg,
                                 float* scratch, const dv_aug_blur* blur, uint8_t* blur_scratch, void* stream) {
  if (blur && (!blur_scratch || (int64_t)H * W * 2 > 160 * 1024)) return DV_EINVAL;
  if (!frames || !table || !y || !scratch || n_src <= 0 || Hs <= 0 || Ws <= 0 || N <= 0 || T_ <= 0 || H <= 0 || W <= 0 ||
      ldy < 4 || ldy % 4 || pad < 0)
    return DV_EINVAL;
  if ((int64_t)N * T_ > 0x7fffffff / 4) return DV_EINVAL;
  if (perm && (n_seg <= 0 || T_ % n_seg)) return DV_EINVAL;
  if ((mean3 == nullptr) != (istd3 == nullptr)) return DV_EINVAL;
  if (dtype != DV_F32 && dtype != DV_BF16) return DV_EUNSUPPORTED;
  AugArgs a;
  a.frames = frames; a.tab = table; a.perm = perm; a.mean3 = mean3; a.istd3 = istd3; a.cmean = scratch;
  a.blur = blur; a.u8tmp = blur_scratch;
  a.n_src = n_src; a.Hs = Hs; a.Ws = Ws; a.F = N * T_; a.T = T_; a.n_seg = perm ? n_seg : 1;
  a.H = H; a.W = W; a.ldy = ldy; a.pad = pad; a.Hp = H + 2 * pad; a.Wp = W + 2 * pad;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(aug_contrast_mean_kernel, dim3(a.F), dim3(kThreads), 0, st, a);
  int rc = dv_launch_status();
  if (rc) return rc;
  const int64_t total = (int64_t)a.F * H * W;
  const int blocks = (int)((total + kThreads - 1) / kThreads < 8192 ? (total + kThreads - 1) / kThreads : 8192);
  if (dtype == DV_F32)
    hipLaunchKernelGGL((aug_apply_kernel<float>), dim3(blocks), dim3(kThreads), 0, st, a, (float*)y);
  else
    hipLaunchKernelGGL((aug_apply_kernel<bf16_t>), dim3(blocks), dim3(kThreads), 0, st, a, (bf16_t*)y);
  if (blur) {
    rc = dv_launch_status();
    if (rc) return rc;
    const size_t lds = (size_t)H * W * 2;
    if (dtype == DV_F32) {
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)aug_blur_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((aug_blur_kernel<float>), dim3(a.F * 3), dim3(kThreads), lds, st, a, (float*)y);
    } else {
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)aug_blur_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((aug_blur_kernel<bf16_t>), dim3(a.F * 3), dim3(kThreads), lds, st, a, (bf16_t*)y);
    }
  }
  return dv_launch_status();
}
