// Hardware probe (development aid, not part of the library): do the matrix pipe and the vector ALU of a gfx950 SIMD overlap
//  (a) across waves -- one wave issues only MFMAs, its SIMD neighbours only vector instructions,
//  (b) inside a wave -- k independent vector instructions after every MFMA?
// One workgroup of 12 waves per CU (96 KB of LDS keeps a second one out): wave w sits on SIMD w % 4, role = w / 4.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/coissue_probe.hip -o tools/probes/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int K>
__device__ __forceinline__ void valu_k(float (&v)[8], float a, float b) {
#pragma unroll
  for (int u = 0; u < K; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(a), "v"(b));
}

__device__ __forceinline__ void mfma4(f32x16 (&acc)[4], bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
}

// roles: bit0 of mask = role 0 runs MFMAs, bit1 = role 1 runs vector ops, bit2 = role 2 runs vector ops, bit3: role 1 runs LDS reads + vector ops
__global__ __launch_bounds__(768) void cross_wave(int mask, int iters, float* out, long long* clk) {
  extern __shared__ unsigned char smem[];
  const int wave = threadIdx.x >> 6, role = wave >> 2, lane = threadIdx.x & 63;
  float res = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  if (role == 0 && (mask & 1)) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    for (int it = 0; it < iters; ++it) mfma4(acc, a, b);           // 4 independent accumulators: back to back
    for (int j = 0; j < 4; ++j) res += acc[j][0];
  } else if ((role == 1 && (mask & 2)) || (role == 2 && (mask & 4))) {
    float v[8];
    for (int u = 0; u < 8; ++u) v[u] = (float)(lane + u);
    for (int it = 0; it < iters; ++it) valu_k<32>(v, 1.0001f, 0.5f);      // 32 vector instructions = 128 cycles = 4 MFMAs' worth of time
    for (int u = 0; u < 8; ++u) res += v[u];
  } else if (role == 1 && (mask & 8)) {
    float v[8];
    for (int u = 0; u < 8; ++u) v[u] = (float)(lane + u);
    const float4* p = reinterpret_cast<const float4*>(smem) + lane;
    for (int it = 0; it < iters; ++it) {
      float4 q0 = p[0], q1 = p[64], q2 = p[128], q3 = p[192];
      asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x));
      v[0] += q0.x; v[1] += q1.x; v[2] += q2.x; v[3] += q3.x;
      valu_k<24>(v, 1.0001f, 0.5f);
    }
    for (int u = 0; u < 8; ++u) res += v[u];
  }
  if (res == 12345.678f) out[0] = res;
  if (blockIdx.x == 7 && lane == 0 && (wave & 3) == 0) {           // waves 0, 4, 8: the three roles on one SIMD
    clk[role * 3 + 0] = clock64() - c0; clk[role * 3 + 1] = wall_clock64() - w0;
    clk[role * 3 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
  }
}

// every wave: 4 MFMAs, each followed by K independent vector instructions
template <int K>
__global__ __launch_bounds__(256) void in_wave(int iters, float* out, long long* clk) {
  const int lane = threadIdx.x & 63;
  const long long c0 = clock64(), w0 = wall_clock64();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  float v[8];
  for (int u = 0; u < 8; ++u) v[u] = (float)(lane + u);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      valu_k<K>(v, 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float res = 0.f;
  for (int j = 0; j < 4; ++j) res += acc[j][0];
  for (int u = 0; u < 8; ++u) res += v[u];
  if (res == 12345.678f) out[0] = res;
  if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
  if (threadIdx.x == 0) {                                          // placement log: where and when this workgroup ran
    long long* lg = clk + 16 + (size_t)blockIdx.x * 4;
    lg[0] = w0; lg[1] = wall_clock64(); lg[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); lg[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
}

// the same with other instruction kinds after every MFMA (OP: 0 v_fma_f32, 1 v_and_b32, 2 v_perm_b32, 3 v_cvt_pk_bf16_f32,
// 4 v_pk_add_f32, 5 v_mov_b32, 6 v_add_u32, 7 ds_read_b128 (1 per 4 slots), 8 s_add_u32, 9 v_sub_f32, 10 v_pk_mul_f32 with MFMA 16x16x32)
template <int OP, int K, bool MF = true>
__global__ __launch_bounds__(256) void in_wave_op(int iters, float* out) {
  __shared__ float4 lds[256];
  const int lane = threadIdx.x & 63;
  lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  float v[8];
  unsigned w[8];
  for (int u = 0; u < 8; ++u) { v[u] = (float)(lane + u); w[u] = lane * 77 + u; }
  unsigned sreg = 1;
  float2 pk[4];
  for (int u = 0; u < 4; ++u) pk[u] = make_float2(v[u], v[u + 4]);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (MF) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < K; ++u) {
        if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(1.0001f), "v"(0.5f));
        if constexpr (OP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[u & 7]) : "v"(0xfffffff7u));
        if constexpr (OP == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(w[u & 7]) : "v"(w[(u + 1) & 7]), "v"(0x07060302u));
        if constexpr (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[u & 7]) : "v"(v[u & 7]), "v"(v[(u + 1) & 7]));
        if constexpr (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[u & 3]) : "v"(pk[(u + 1) & 3]));
        if constexpr (OP == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(w[u & 7]) : "v"(w[(u + 1) & 7]));
        if constexpr (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[u & 7]) : "v"(3u));
        if constexpr (OP == 7) { if ((u & 3) == 0) { float4 q = lds[(lane + u) & 255]; asm volatile("" : "+v"(q.x)); v[u & 7] += q.x; } }
        if constexpr (OP == 8) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sreg));
        if constexpr (OP == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[u & 7]) : "v"(0.5f));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float res = (float)sreg;
  for (int j = 0; j < 4; ++j) res += acc[j][0];
  for (int u = 0; u < 8; ++u) res += v[u] + (float)w[u];
  for (int u = 0; u < 4; ++u) res += pk[u].x + pk[u].y;
  if (res == 12345.678f) out[0] = res;
}

template <typename F>
static float timed(F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  float* out;
  CHECK(hipMalloc(&out, 64));
  long long *clk_d, clk[9];
  CHECK(hipMalloc(&clk_d, 128 + 768 * 32));
  const int iters = 20000;                                       // x 4 MFMAs x 32 cycles = 2.56 M cycles
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cross_wave), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  const char* names[] = {"", "mfma only (1 wave/SIMD)", "vector only (1 wave/SIMD)", "mfma + vector", "", "", "vector x2 (2 waves/SIMD)",
                         "mfma + vector x2", "", "mfma + (LDS reads + vector)"};
  for (int mask : {1, 2, 3, 6, 7, 8, 9}) {
    const float ms = timed([&] { hipLaunchKernelGGL(cross_wave, dim3(256), dim3(768), 96 * 1024, 0, mask, iters, out, clk_d); });
    CHECK(hipMemcpy(clk, clk_d, 72, hipMemcpyDeviceToHost));
    printf("cross-wave mask %d  %-32s %8.3f ms | cycles per MFMA-time-slot (iters*4): role0 %.1f role1 %.1f role2 %.1f | MHz %.0f %.0f %.0f | simd %lld %lld %lld\n",
           mask, mask < 10 ? names[mask] : "", ms, clk[0] / (iters * 4.0), clk[3] / (iters * 4.0), clk[6] / (iters * 4.0),
           100.0 * clk[0] / clk[1], 100.0 * clk[3] / clk[4], 100.0 * clk[6] / clk[7], (clk[2] >> 4) & 3, (clk[5] >> 4) & 3, (clk[8] >> 4) & 3);
  }

#define INW(K_) { const float ms = timed([&] { hipLaunchKernelGGL(in_wave<K_>, dim3(256), dim3(256), 0, 0, iters, out, clk_d); }); \
    CHECK(hipMemcpy(clk, clk_d, 16, hipMemcpyDeviceToHost)); \
    printf("in-wave: 1 MFMA + %2d vector instructions  %8.3f ms  clock64 %.1f per MFMA, wall_clock64 %.2f per MFMA (ratio %.2f)\n", K_, ms, \
           clk[0] / (iters * 4.0), clk[1] / (iters * 4.0), (double)clk[0] / clk[1]); }
  INW(0) INW(2) INW(4) INW(6) INW(7) INW(8) INW(10) INW(12) INW(16)
  // the same with three workgroups per CU (3 waves per SIMD, every wave alternating)
#define INW3(K_) { const float ms = timed([&] { hipLaunchKernelGGL(in_wave<K_>, dim3(768), dim3(256), 0, 0, iters, out, clk_d); }); \
    CHECK(hipMemcpy(clk, clk_d, 16, hipMemcpyDeviceToHost)); \
    printf("in-wave x3 waves/SIMD: 1 MFMA + %2d vector  %8.3f ms  clock64 %.1f per MFMA and SIMD, wall_clock64 %.2f (ratio %.2f)\n", K_, ms, \
           clk[0] / (iters * 12.0), clk[1] / (iters * 12.0), (double)clk[0] / clk[1]); }
  auto placement = [&](int nblk) {
    static long long lg[768 * 4];
    CHECK(hipMemcpy(lg, clk_d + 16, nblk * 32, hipMemcpyDeviceToHost));
    int per_cu[8 * 64] = {0}, hist[16] = {0};
    long long t0 = lg[0], t1 = lg[1], dmin = 1ll << 60, dmax = 0;
    for (int b = 0; b < nblk; ++b) {
      const unsigned hw = (unsigned)lg[b * 4 + 2], xcc = (unsigned)lg[b * 4 + 3] & 15;
      const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_cu[(xcc & 7) * 64 + ((se & 3) * 16 + cu) % 64]++;
      if (lg[b * 4] < t0) t0 = lg[b * 4];
      if (lg[b * 4 + 1] > t1) t1 = lg[b * 4 + 1];
      const long long d = lg[b * 4 + 1] - lg[b * 4];
      if (d < dmin) dmin = d;
      if (d > dmax) dmax = d;
      (void)sh;
    }
    for (int i = 0; i < 8 * 64; ++i) if (per_cu[i]) hist[per_cu[i] < 15 ? per_cu[i] : 15]++;
    printf("   placement of %d workgroups: span %.3f ms, workgroup duration %.3f .. %.3f ms; CUs holding n workgroups:", nblk, (t1 - t0) / 1e5, dmin / 1e5, dmax / 1e5);
    for (int n = 1; n < 16; ++n) if (hist[n]) printf(" n=%d: %d", n, hist[n]);
    printf("  (hw_id of block 0: 0x%llx xcc 0x%llx)\n", lg[2], lg[3]);
  };
  INW3(0) placement(768); INW3(4) INW3(7) INW3(8) INW3(12) INW3(16)
  const char* ops[] = {"v_fma_f32", "v_and_b32", "v_perm_b32", "v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_mov_b32", "v_add_u32", "ds_read_b128 (K/4)", "s_add_u32", "v_sub_f32"};
#define INOP(OP_) { const float m0 = timed([&] { hipLaunchKernelGGL((in_wave_op<OP_, 8, true>), dim3(768), dim3(256), 0, 0, iters, out); }); \
    const float m1 = timed([&] { hipLaunchKernelGGL((in_wave_op<OP_, 8, false>), dim3(768), dim3(256), 0, 0, iters, out); }); \
    const float m2 = timed([&] { hipLaunchKernelGGL((in_wave_op<OP_, 16, true>), dim3(768), dim3(256), 0, 0, iters, out); }); \
    printf("x3 waves/SIMD, per MFMA 8 x %-20s with MFMA %7.3f ms, without %7.3f ms; 16 x with MFMA %7.3f ms\n", ops[OP_], m0, m1, m2); }
  INOP(0) INOP(1) INOP(2) INOP(3) INOP(4) INOP(5) INOP(6) INOP(7) INOP(8) INOP(9)
  return 0;
}
