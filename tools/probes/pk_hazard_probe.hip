// Hardware / compiler probe (development aid): is the result of v_pk_add_f32 safe to read ONE wait state later (the `s_nop 0`
// the compiler puts between a packed-f32 instruction and a v_cndmask that consumes its result) on gfx950, also when other waves
// keep the SIMD's matrix pipe and vector ALU busy?
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/pk_hazard_probe.hip -o tools/probes/pk_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// NOPS: 0 = consumer directly behind the packed add (no wait state), 1 = `s_nop 0`, 2 = `s_nop 1`; CONS: 0 v_cndmask (vcc), 1 v_mov
template <int NOPS, int CONS>
__global__ __launch_bounds__(256) void probe(int iters, unsigned* bad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned nbad = 0, badlanes = 0;
  if ((blockIdx.x & 1) == 0) {
    f32x2 k = {3.0f, 5.0f};
    for (int it = 0; it < iters; ++it) {
      f32x2 y = {(float)(it + lane), (float)(2 * it + lane)};
      float r;
      // v[20:21] = k + y; then the low half through the consumer.  The previous value of v20 is NOT k.x + y.x (it is set to -1 first).
      if (NOPS == 0 && CONS == 0)
        asm volatile("s_mov_b64 vcc, -1\n\tv_mov_b32 v20, -1.0\n\tv_mov_b32 v21, -1.0\n\ts_nop 4\n\tv_pk_add_f32 v[20:21], %1, %2\n\tv_cndmask_b32_e32 %0, 0, v20, vcc" : "=v"(r) : "v"(k), "v"(y) : "v20", "v21", "vcc");
      if (NOPS == 1 && CONS == 0)
        asm volatile("s_mov_b64 vcc, -1\n\tv_mov_b32 v20, -1.0\n\tv_mov_b32 v21, -1.0\n\ts_nop 4\n\tv_pk_add_f32 v[20:21], %1, %2\n\ts_nop 0\n\tv_cndmask_b32_e32 %0, 0, v20, vcc" : "=v"(r) : "v"(k), "v"(y) : "v20", "v21", "vcc");
      if (NOPS == 2 && CONS == 0)
        asm volatile("s_mov_b64 vcc, -1\n\tv_mov_b32 v20, -1.0\n\tv_mov_b32 v21, -1.0\n\ts_nop 4\n\tv_pk_add_f32 v[20:21], %1, %2\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, 0, v20, vcc" : "=v"(r) : "v"(k), "v"(y) : "v20", "v21", "vcc");
      if (NOPS == 0 && CONS == 1)
        asm volatile("v_mov_b32 v20, -1.0\n\tv_mov_b32 v21, -1.0\n\ts_nop 4\n\tv_pk_add_f32 v[20:21], %1, %2\n\tv_mov_b32 %0, v20" : "=v"(r) : "v"(k), "v"(y) : "v20", "v21");
      if (NOPS == 1 && CONS == 1)
        asm volatile("v_mov_b32 v20, -1.0\n\tv_mov_b32 v21, -1.0\n\ts_nop 4\n\tv_pk_add_f32 v[20:21], %1, %2\n\ts_nop 0\n\tv_mov_b32 %0, v20" : "=v"(r) : "v"(k), "v"(y) : "v20", "v21");
      const bool wrong = r != k.x + y.x;
      nbad += wrong;
      badlanes |= wrong ? 1u : 0u;
    }
  } else {
    // neighbours on the same SIMDs: MFMAs and vector instructions
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)1.0f; }
    float v = (float)lane;
    for (int it = 0; it < iters; ++it) {
      acc[it & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[it & 1], 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(1.0f));
    }
    if (acc[0][0] + acc[1][0] + v == 12345.f) bad[15] = 1;
  }
  if (nbad) { atomicAdd(bad, nbad); atomicAdd(bad + 1 + (lane >> 4), 1u); }
  (void)wave;
}

int main() {
  unsigned* bad;
  CHECK(hipMalloc(&bad, 64));
  const int iters = 20000;
#define RUN(N_, C_) { CHECK(hipMemset(bad, 0, 64)); hipLaunchKernelGGL((probe<N_, C_>), dim3(1536), dim3(256), 0, 0, iters, bad); CHECK(hipDeviceSynchronize()); \
    unsigned h[16]; CHECK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost)); \
    printf("wait states %d, consumer %s: stale reads %u (threads with any, by lane quarter: %u %u %u %u)\n", N_, C_ ? "v_mov_b32" : "v_cndmask_b32 (vcc)", h[0], h[1], h[2], h[3], h[4]); }
  RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(0, 1) RUN(1, 1)
  return 0;
}
