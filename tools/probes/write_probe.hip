// HBM write-rate probe (gfx950).  Question: the tower's write-heavy kernels (gemm_wreg: 1.05 GB written + 0.33 GB read per
// launch, clip_embed: writes only) see ~3 TB/s while the read-heavy LayerNorm pass sees ~6 TB/s -- is that the access
// pattern of those kernels or what this memory system gives stores?
//   ./write_probe [MB]        (default 1280 MB, larger than L2 + Infinity Cache)
// Patterns (every one moves the whole buffer once per launch, 512 threads per work-group, grid-stride over 256 x 8 WGs):
//   fill      each wave stores 1 KiB contiguous per instruction (global_store_dwordx4), WG-contiguous 8 KiB
//   fill_nt   the same through __builtin_nontemporal_store
//   tile64    the weight-stationary GEMM's output pattern: a WG owns 32 rows x 512 B at a 4096-B row pitch, each wave
//             64 B of every row (16 rows x 64 B per instruction)
//   tile128   the same block as full 128-B lines: a pair of waves owns 128 B of every row (8 rows x 128 B per instruction)
//   copy      read one buffer, write another (1 : 1)
//   read      read only (sum into a sink)
//   r2w1      read two buffers, write one (the LayerNorm pass's ratio)
//   memset    hipMemsetAsync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_fill(u32x4* dst, long n16, int nt) {
  const long stride = (long)gridDim.x * 512;
  const u32x4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
  for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n16; i += stride) {
    if (nt) __builtin_nontemporal_store(v, dst + i);
    else dst[i] = v;
  }
}

// block of 32 rows x 512 B at pitch 4096 B: 8 column groups per row range, row ranges of 32 rows
__global__ __launch_bounds__(512) void k_tile(unsigned char* dst, long rows, int wide) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nblk = (rows / 32) * 8;
  const u32x4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
  for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
    const long rb = b >> 3; const int cg = (int)(b & 7);
    unsigned char* base = dst + rb * 32 * 4096L + cg * 512;
    if (!wide) {  // wave: 64 B of each row, two instructions of 16 rows
#pragma unroll
      for (int p = 0; p < 2; ++p)
        *(u32x4*)(base + (long)(p * 16 + (lane >> 2)) * 4096 + wave * 64 + (lane & 3) * 16) = v;
    } else {      // wave pair: 128 B of each row; wave (2q + h) stores rows h*16 .. h*16+15, two instructions of 8 rows
      const int q = wave >> 1, h = wave & 1;
#pragma unroll
      for (int p = 0; p < 2; ++p)
        *(u32x4*)(base + (long)(h * 16 + p * 8 + (lane >> 3)) * 4096 + q * 128 + (lane & 7) * 16) = v;
    }
  }
}

__global__ __launch_bounds__(512) void k_copy(const u32x4* a, const u32x4* b, u32x4* dst, long n16, int mode, unsigned* sink) {
  const long stride = (long)gridDim.x * 512;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (long i = (long)blockIdx.x * 512 + threadIdx.x; i < n16; i += stride) {
    u32x4 v = a[i];
    if (mode == 2) { const u32x4 w = b[i]; v.x ^= w.x; v.y ^= w.y; v.z ^= w.z; v.w ^= w.w; }
    if (mode == 0) { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    else dst[i] = v;
  }
  if (mode == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

int main(int argc, char** argv) {
  const long mb = argc > 1 ? atol(argv[1]) : 1280;
  const long bytes = mb << 20, n16 = bytes / 16, rows = bytes / 4096;
  unsigned char *a, *b, *c; unsigned* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(c, 3, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  auto timeit = [&](const char* name, double moved, auto&& launch) -> int {
    std::vector<float> t;
    for (int r = 0; r < 9; ++r) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-8s %6ld MB buffer  median %.4f ms  %.2f TB/s (bytes moved %.2f GB)\n", name, mb, t[t.size() / 2],
           moved / (t[t.size() / 2] * 1e-3) / 1e12, moved / 1e9);
    return 0;
  };
  timeit("fill", (double)bytes, [&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(512), 0, 0, (u32x4*)c, n16, 0); });
  timeit("fill_nt", (double)bytes, [&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(512), 0, 0, (u32x4*)c, n16, 1); });
  timeit("tile64", (double)bytes, [&] { hipLaunchKernelGGL(k_tile, dim3(grid), dim3(512), 0, 0, c, rows, 0); });
  timeit("tile128", (double)bytes, [&] { hipLaunchKernelGGL(k_tile, dim3(grid), dim3(512), 0, 0, c, rows, 1); });
  timeit("copy", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(512), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)c, n16, 1, sink); });
  timeit("read", (double)bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(512), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)c, n16, 0, sink); });
  timeit("r2w1", 3.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(512), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)c, n16, 2, sink); });
  timeit("memset", (double)bytes, [&] { (void)hipMemsetAsync(c, 0, bytes, 0); });
  return 0;
}
