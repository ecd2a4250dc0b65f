// Hardware probe (development aid): after `s_waitcnt vmcnt(0)` + `s_barrier`, is ALL of an LDS-DMA piece
// (buffer_load_dwordx4 ... lds, 64 lanes x 16 B) visible to a ds_read issued right behind the barrier -- including the bytes of
// the last lanes of the last piece a wave issued -- when a second workgroup shares the CU and several pieces are in flight?
// Each step every wave DMAs PIECES pieces of fresh data (values encode step, piece, lane) into its own LDS region, waits, meets the
// barrier, and the NEXT wave (wave + 1) & 3 checks them.  DELAY: s_nop cycles between the barrier and the reads.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/dma_land_probe.hip -o tools/probes/dma_land_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int rsrc_t;

__device__ __forceinline__ void dma16(rsrc_t rsrc, unsigned lds_base, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_base), "v"(voff), "s"(rsrc) : "memory", "m0");
}

template <int PIECES, int DELAY, int LDSB>
__global__ __launch_bounds__(256) void land(const unsigned* __restrict__ src, unsigned src_bytes, int steps, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDSB];            // 2 stages x 4 waves x PIECES KiB used
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long a = (unsigned long long)src;
  const rsrc_t rsrc = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, src_bytes, 0x00020000u};
  const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)smem;
  unsigned nbad = 0, bad_lane_q = 0;
  // source: word w of the array holds w; step s, wave w, piece p reads 1 KiB at ((blockIdx * steps + s) * 4 + w) * PIECES + p
  for (int s = 0; s < steps; ++s) {
    const int st = s & 1;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const unsigned kib = (((unsigned)blockIdx.x * steps + s) * 4 + wave) * PIECES + p;
      dma16(rsrc, base + ((st * 4 + wave) * PIECES + p) * 1024, (kib % (src_bytes / 1024)) * 1024 + lane * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (DELAY == 1) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    if (DELAY == 2) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory"); }
    const int rw = (wave + 1) & 3;                 // read the neighbour wave's pieces, last piece first
#pragma unroll
    for (int p = PIECES - 1; p >= 0; --p) {
      const uint4 v = *reinterpret_cast<const uint4*>(smem + ((st * 4 + rw) * PIECES + p) * 1024 + lane * 16);
      const unsigned kib = ((((unsigned)blockIdx.x * steps + s) * 4 + rw) * PIECES + p) % (src_bytes / 1024);
      const unsigned w0 = kib * 256 + lane * 4;
      if (v.x != w0 || v.y != w0 + 1 || v.z != w0 + 2 || v.w != w0 + 3) { ++nbad; bad_lane_q |= 1u << (lane >> 4); }
    }
    __builtin_amdgcn_s_barrier();                  // WAR: the stage is refilled two steps later
  }
  if (nbad) { atomicAdd(out, nbad); for (int q = 0; q < 4; ++q) if (bad_lane_q >> q & 1) atomicAdd(out + 1 + q, 1u); }
}

int main() {
  const unsigned src_bytes = 256u << 20;
  unsigned *src, *out;
  CHECK(hipMalloc(&src, src_bytes));
  CHECK(hipMalloc(&out, 64));
  unsigned* h = (unsigned*)malloc(src_bytes);
  for (unsigned i = 0; i < src_bytes / 4; ++i) h[i] = i;
  CHECK(hipMemcpy(src, h, src_bytes, hipMemcpyHostToDevice));
#define RUN(P_, D_, L_, G_) { CHECK(hipMemset(out, 0, 64)); hipLaunchKernelGGL((land<P_, D_, L_>), dim3(G_), dim3(256), 0, 0, src, src_bytes, 400, out); \
    CHECK(hipDeviceSynchronize()); unsigned r[8]; CHECK(hipMemcpy(r, out, 32, hipMemcpyDeviceToHost)); \
    printf("pieces/wave %d, delay %d, LDS %6d B, grid %4d: stale 16-byte reads %u (threads by lane quarter: %u %u %u %u)\n", P_, D_, L_, G_, r[0], r[1], r[2], r[3], r[4]); }
  RUN(2, 0, 16384, 256) RUN(2, 0, 16384, 2048) RUN(6, 0, 49152, 768) RUN(6, 0, 80896, 512) RUN(6, 1, 80896, 512) RUN(6, 2, 80896, 512)
  RUN(8, 0, 65536, 512) RUN(8, 0, 80896, 512) RUN(8, 2, 80896, 512)
  return 0;
}
