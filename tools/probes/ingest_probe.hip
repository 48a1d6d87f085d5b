// Per-CU global->register ingest probe (gfx950).  Question it answers: how many bytes per clock can
// the waves of one CU pull from L2 / HBM with plain global_load_dwordx4 in the access pattern of a
// weight-stationary GEMM (each wave streams its own 32 activation rows, 128 bytes per row per step)?
//   ./ingest_probe <waves_per_wg> <depth> <mode> <rows_total> <K>
//   mode 0: every WG streams its own slice (HBM);  1: the 16 WGs of a group stream the same slice
//   (first one misses, the rest hit L2 when co-located);  2: every WG re-reads one small slice (L2-hot)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(1024) void ingest(const unsigned char* A, long rows, int K, int mode, int rep, int pat, long pitch_arg,
                                               unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int wg = blockIdx.x;
  // group id / member: mode 1 puts 16 consecutive-in-XCD WGs on one slice
  const int xcd = wg & 7, loc = wg >> 3;
  int slice, nslices;
  if (mode == 0 || mode == 3) { slice = wg; nslices = gridDim.x; }
  else if (mode == 1) { slice = xcd * (32 / 16) + loc / 16; nslices = 8 * 2; }
  else { slice = 0; nslices = 1; }
  const long rows_per_slice = rows / nslices;
  const long row0 = (long)slice * rows_per_slice;
  const int rt_count = (int)(rows_per_slice / 32);
  const long pitch = pitch_arg;
  const int chunks = K / 64;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint4 buf[DEPTH][4];
  // flatten (row tile, chunk) into one stream for this wave
  const int my_rt = (rt_count - wave + nw - 1) / nw;  // row tiles wave, wave+nw, ...
  const long per_pass = (long)my_rt * chunks;
  const long total = per_pass * rep;
  auto addr = [&](long s) {
    const long sp = s % per_pass;
    const long t = sp / chunks; const int c = (int)(sp - t * chunks);
    const long rbase = row0 + (wave + t * nw) * 32;
    // every pattern moves the same 32 rows x 128 B per step (4 x dwordx4 per lane), lanes mapped differently
    if (pat == 0) return A + (rbase + (lane & 31)) * pitch + c * 128 + (lane >> 5) * 64;   // row per lane, 64 B per lane
    return A + rbase * pitch + c * 4096L + lane * 16;  // pat 1: 4 KiB contiguous per step (pitch ignored)
  };
  auto joff = [&](int j) -> long {
    if (pat == 0) return j * 16;
    if (pat == 1) return j * 1024;
    return 0;
  };
  auto addr2 = [&](long s, int j) {
    const long sp = s % per_pass;
    const long t = sp / chunks; const int c = (int)(sp - t * chunks);
    const long rbase = row0 + (wave + t * nw) * 32;
    if (pat == 2) return A + (rbase + j * 8 + (lane >> 3)) * pitch + c * 128 + (lane & 7) * 16;        // 8 rows x 128 B per instr
    /* pat 3 */ return A + (rbase + (j >> 1) * 16 + (lane >> 2)) * pitch + c * 128 + (j & 1) * 64 + (lane & 3) * 16;  // 16 rows x 64 B
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < total) {
#pragma unroll
      for (int j = 0; j < 4; ++j) buf[d][j] = pat < 2 ? *(const uint4*)(addr(d) + joff(j)) : *(const uint4*)addr2(d, j);
    }
  for (long s = 0; s < total; s += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (s + d < total) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc.x ^= buf[d][j].x; acc.y += buf[d][j].y; acc.z ^= buf[d][j].z; acc.w += buf[d][j].w; }
        if (s + d + DEPTH < total) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            buf[d][j] = pat < 2 ? *(const uint4*)(addr(s + d + DEPTH) + joff(j)) : *(const uint4*)addr2(s + d + DEPTH, j);
        }
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 12;
  const int depth = argc > 2 ? atoi(argv[2]) : 2;
  const int mode = argc > 3 ? atoi(argv[3]) : 0;
  const long rows = argc > 4 ? atol(argv[4]) : 311808;
  const int K = argc > 5 ? atoi(argv[5]) : 512;
  const int rep = argc > 6 ? atoi(argv[6]) : 1;
  const int pat = argc > 7 ? atoi(argv[7]) : 0;
  const long pitch = argc > 8 ? atol(argv[8]) : (long)K * 2;
  unsigned char* A; unsigned* sink;
  const size_t bytes = (size_t)rows * K * 2;
  const size_t alloc = (size_t)rows * pitch + (1 << 20);
  CK(hipMalloc(&A, alloc)); CK(hipMemset(A, 1, alloc)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = argc > 9 ? atoi(argv[9]) : 256;
  auto launch = [&]() {
    if (depth == 1) hipLaunchKernelGGL(ingest<1>, dim3(grid), dim3(waves * 64), 0, 0, A, rows, K, mode, rep, pat, pitch, sink);
    else if (depth == 2) hipLaunchKernelGGL(ingest<2>, dim3(grid), dim3(waves * 64), 0, 0, A, rows, K, mode, rep, pat, pitch, sink);
    else if (depth == 3) hipLaunchKernelGGL(ingest<3>, dim3(grid), dim3(waves * 64), 0, 0, A, rows, K, mode, rep, pat, pitch, sink);
    else hipLaunchKernelGGL(ingest<4>, dim3(grid), dim3(waves * 64), 0, 0, A, rows, K, mode, rep, pat, pitch, sink);
  };
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int it = 5;
  for (int i = 0; i < it; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
  // bytes pulled by all CUs per launch
  double pulled;
  if (mode == 0 || mode == 3) pulled = (double)bytes;
  else if (mode == 1) pulled = (double)bytes / 16 * 256;   // 16 slices, 256 WGs each streaming one
  else pulled = (double)bytes * grid;
  pulled *= rep;
  printf("pat=%d pitch=%ld waves=%d depth=%d mode=%d rows=%ld K=%d: %.3f ms  %.2f TB/s aggregate  %.1f GB/s/CU  (%.1f B/clk @2.1GHz)\n", pat, pitch, waves, depth,
         mode, rows, K, ms, pulled / ms / 1e9, pulled / ms / 1e6 / grid, pulled / ms / 1e6 / grid / 2.1);
  return 0;
}
