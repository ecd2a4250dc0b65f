// Hardware probe (development aid): do two workgroups that are co-resident on a CU and together use (almost) all of its 160 KB
// of LDS really get disjoint allocations?  Every workgroup fills its whole static LDS array with a pattern derived from its id,
// waits (so that its neighbour does the same), and checks the pattern; a mismatch is an overlap.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_overlap_probe.hip -o tools/probes/lds_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BYTES>
__global__ __launch_bounds__(256) void fill_check(unsigned* out, int rounds) {
  __shared__ unsigned lds[BYTES / 4];
  const unsigned tag = (blockIdx.x + 1) * 0x9E3779B1u;
  unsigned first_bad = 0xffffffffu, nbad = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < BYTES / 4; i += 256) lds[i] = tag ^ (unsigned)(i * 2654435761u) ^ (unsigned)r;
    __syncthreads();
    for (int w = 0; w < 200; ++w) __builtin_amdgcn_s_sleep(20);           // let the neighbour write
    __syncthreads();
    for (int i = threadIdx.x; i < BYTES / 4; i += 256) {
      const unsigned want = tag ^ (unsigned)(i * 2654435761u) ^ (unsigned)r;
      if (lds[i] != want) { ++nbad; if ((unsigned)i < first_bad) first_bad = (unsigned)i; }
    }
    __syncthreads();
  }
  if (nbad) { atomicAdd(out, nbad); atomicMin(out + 1, first_bad * 4); atomicMax(out + 2, first_bad * 4); atomicAdd(out + 3, 1u); }
}

int main() {
  unsigned* out;
  CHECK(hipMalloc(&out, 64));
#define RUN(B_) { unsigned init[4] = {0, 0xffffffffu, 0, 0}; CHECK(hipMemcpy(out, init, 16, hipMemcpyHostToDevice)); \
    int nb = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fill_check<B_>, 256, 0)); \
    hipLaunchKernelGGL(fill_check<B_>, dim3(1024), dim3(256), 0, 0, out, 6); CHECK(hipDeviceSynchronize()); \
    unsigned h[4]; CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); \
    printf("static LDS %6d B (%.2f x 1280, %.2f x 512), runtime says %d workgroups per CU: wrong words %u in %u threads, lowest first-bad byte offset %u, highest %u\n", \
           B_, B_ / 1280.0, B_ / 512.0, nb, h[0], h[3], h[0] ? h[1] : 0, h[2]); }
  RUN(67584) RUN(79360) RUN(80384) RUN(80640) RUN(80896) RUN(81408) RUN(81920) RUN(54272) RUN(54528) RUN(40960) RUN(40704)
  return 0;
}
