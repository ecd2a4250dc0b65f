// Hardware / compiler probe (development aid): does a chain of DEPENDENT v_mfma_f32_32x32x16_bf16 (same accumulator, back to
// back, as the compiler emits them with its own wait states) accumulate exactly when several waves contend for the SIMD's
// matrix pipe?  A = B = 1.0: every MFMA adds exactly 16 to every element, so after n of them every element must be 16 n.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_probe.hip -o tools/probes/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: one accumulator, 6 dependent MFMAs per iteration; MODE 1: the same with two vector instructions between them;
// MODE 2: two accumulators interleaved (each MFMA depends on the one before the previous)
template <int MODE>
__global__ __launch_bounds__(256) void chain(int iters, unsigned* bad) {
  const int lane = threadIdx.x & 63;
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)1.0f; }
  float v = (float)lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      if (MODE == 2 && (q & 1)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      if (MODE == 1) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(1.0f)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(1.0f)); }
    }
  }
  const float want0 = 16.f * iters * (MODE == 2 ? 3 : 6), want1 = MODE == 2 ? 16.f * iters * 3 : 0.f;
  unsigned nbad = 0;
  for (int r = 0; r < 16; ++r) nbad += (acc0[r] != want0) + (acc1[r] != want1);
  if (nbad) { atomicAdd(bad, nbad); atomicAdd(bad + 1 + (threadIdx.x >> 6 & 3), 1u); }
  if (v == 12345.f) bad[15] = 1;
}

int main() {
  unsigned* bad;
  CHECK(hipMalloc(&bad, 64));
  const int iters = 2000;
  for (int blocks : {256, 768, 1280}) {
    for (int mode = 0; mode < 3; ++mode) {
      CHECK(hipMemset(bad, 0, 64));
      if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(blocks), dim3(256), 0, 0, iters, bad);
      if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(256), 0, 0, iters, bad);
      if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(blocks), dim3(256), 0, 0, iters, bad);
      CHECK(hipDeviceSynchronize());
      unsigned h[16];
      CHECK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost));
      printf("workgroups %4d (waves per SIMD %d) mode %d: wrong accumulator elements %u\n", blocks, blocks / 256, mode, h[0]);
    }
  }
  return 0;
}
