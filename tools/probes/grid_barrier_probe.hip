// How much does a device-wide barrier cost inside ONE persistent kernel on MI355X (256 work-groups, 8 XCDs with their own L2s),
// against the boundary between two dependent kernels of a stream?  Answers whether a cooperative per-step kernel could beat the
// ~200 dependent launches of a single-image polishing step.
//   build/grid_barrier_probe [work-groups] [threads] [barriers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);  // agent scope: writes of this work-group are visible device-wide first
    unsigned spins = 0;
    while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > 40000000u) break;  // never hang the box
    }
  }
  __syncthreads();
}

// every phase: each work-group reads what its neighbour wrote in the previous phase (a real cross-XCD dependency), adds, writes
__global__ __launch_bounds__(1024) void chain(unsigned* ctr, float* buf, int phases) {
  const int wg = blockIdx.x, n = gridDim.x;
  float v = 0.f;
  for (int p = 0; p < phases; ++p) {
    if (threadIdx.x == 0) {
      v += __builtin_nontemporal_load(buf + ((wg + 1) % n) * 32 + (p & 1) * 16);
      __builtin_nontemporal_store(v + 1.f, buf + wg * 32 + ((p + 1) & 1) * 16);
    }
    grid_barrier(ctr, (unsigned)(p + 1) * n);
  }
  if (threadIdx.x == 0) buf[wg * 32 + 8] = v;
}

__global__ void tiny(float* buf, int p) {
  const int wg = blockIdx.x, n = gridDim.x;
  if (threadIdx.x == 0) buf[wg * 32 + ((p + 1) & 1) * 16] = buf[((wg + 1) % n) * 32 + (p & 1) * 16] + 1.f;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 256, thr = argc > 2 ? atoi(argv[2]) : 256, phases = argc > 3 ? atoi(argv[3]) : 200;
  unsigned* ctr; float* buf;
  CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&buf, wgs * 128));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(ctr, 0, 64)); CK(hipMemset(buf, 0, wgs * 128));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(chain, dim3(wgs), dim3(thr), 0, 0, ctr, buf, phases);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    float h; CK(hipMemcpy(&h, buf + 8, 4, hipMemcpyDeviceToHost));
    printf("persistent kernel: %d work-groups x %d threads, %d barriers: %.3f ms = %.2f us per barrier (chain value %.0f)\n", wgs, thr, phases, ms, ms * 1e3 / phases, h);
  }
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(buf, 0, wgs * 128));
    CK(hipEventRecord(a));
    for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(thr), 0, 0, buf, p);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    printf("dependent launches: %d kernels of %d work-groups: %.3f ms = %.2f us per kernel\n", phases, wgs, ms, ms * 1e3 / phases);
  }
  return 0;
}
