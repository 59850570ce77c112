// What can 256 CUs pull into LDS by LDS-DMA (global_load_lds_dwordx4), chip-wide, with the GEMM's own access shape and
// nothing else in the loop?  One 512-thread workgroup per CU; per "half-tile" every wave issues two 1-KiB pieces (8 rows x
// 128 B of a row-major bf16 matrix with leading dimension LD), then waits with a counted vmcnt that leaves INFL pieces in
// flight -- the cadence of gemm3_kernel's LOAD segments without fragment reads, MFMAs or barriers.
//   mode 0: every workgroup of an XCD walks the SAME panel (L2 hits after the first touch: the GEMM's shared operand panels)
//   mode 1: every workgroup walks its own panel (streamed from HBM / Infinity Cache: split-K dW)
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_bw lds_dma_bw.hip ; run: ./lds_dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define GLB_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

template <int INFL>
__global__ __launch_bounds__(512, 2) void dma_kernel(const char* base, long panel_bytes, int mode, int ld_bytes, int iters,
                                                     int rows_per_panel, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7;
  const char* panel = base + (mode == 0 ? (long)xcd : (long)blockIdx.x) * panel_bytes;
  // piece: rows wave*8 + lane/8 (+64 for the second piece), 16 B at (lane & 7) * 16 of a 128-B K-slice
  const long off0 = (long)(wave * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16;
  const long off1 = off0 + 64L * ld_bytes;
  int slot = 0, row = 0, kcol = 0;
  const int kcols = ld_bytes / 128;
  for (int it = 0; it < iters; ++it) {
    const char* src = panel + (long)row * ld_bytes + (long)kcol * 128;
    char* dst = smem + slot * 16384 + wave * 1024;
    __builtin_amdgcn_global_load_lds((GLB_AS void*)(src + off0), (LDS_AS void*)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((GLB_AS void*)(src + off1), (LDS_AS void*)(dst + 8192), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL) : "memory");
    slot = (slot + 1) & 7;
    if (++kcol == kcols) {  // walk K first (like a K loop), then the next 128 rows of the panel
      kcol = 0;
      row += 128;
      if (row + 128 > rows_per_panel) row = 0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && smem[0] == 123 && smem[777] == 45) sink[0] = 1;
}

template <int INFL>
static float run(const char* d, long panel_bytes, int mode, int ld_bytes, int iters, int rows, int* sink, int nwg) {
  auto fn = dma_kernel<INFL>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fn, dim3(nwg), dim3(512), 131072, 0, d, panel_bytes, mode, ld_bytes, iters, rows, sink);
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(fn, dim3(nwg), dim3(512), 131072, 0, d, panel_bytes, mode, ld_bytes, iters, rows, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const int nwg = 256, ld_bytes = 768 * 2;
  // a panel = 256 rows x 768 bf16 (384 KiB): the A panel of one output-tile row of a K = 768 GEMM
  const int rows = 256;
  const long panel_bytes = (long)rows * ld_bytes;
  char* d; int* sink;
  const long total = panel_bytes * nwg;
  hipMalloc(&d, total + (1 << 20));
  hipMemset(d, 1, total + (1 << 20));
  hipMalloc(&sink, 64);
  const int iters = 4000;  // half-tiles per workgroup: 16 KiB each -> 62.5 MiB per workgroup
  const double bytes = (double)nwg * iters * 16384.0;
  for (int mode = 0; mode < 2; ++mode) {
    float t4 = run<4>(d, panel_bytes, mode, ld_bytes, iters, rows, sink, nwg);
    float t8 = run<8>(d, panel_bytes, mode, ld_bytes, iters, rows, sink, nwg);
    float t12 = run<12>(d, panel_bytes, mode, ld_bytes, iters, rows, sink, nwg);
    float t14 = run<14>(d, panel_bytes, mode, ld_bytes, iters, rows, sink, nwg);
    printf("mode %d (%s): pieces in flight/wave 4: %.2f TB/s | 8: %.2f TB/s | 12: %.2f TB/s | 14: %.2f TB/s   (per CU at 8: %.1f GB/s, %.2f us per 16 KiB)\n",
           mode, mode == 0 ? "one 384-KiB panel per XCD, L2-resident" : "own 384-KiB panel per workgroup, 96 MiB in all",
           bytes / t4 / 1e9, bytes / t8 / 1e9, bytes / t12 / 1e9, bytes / t14 / 1e9, bytes / t8 / 1e6 / nwg, t8 * 1e3 / iters);
  }
  // HBM stream: every workgroup walks a private 16 MiB region once (4 GiB in all)
  {
    const long big = 16L << 20;
    char* h; hipMalloc(&h, big * nwg + (1 << 20)); hipMemset(h, 1, big * nwg);
    const int rows_b = (int)(big / ld_bytes) / 128 * 128;
    const int it_b = rows_b / 128 * (ld_bytes / 128);
    float t8 = run<8>(h, big, 1, ld_bytes, it_b, rows_b, sink, nwg);
    float t14 = run<14>(h, big, 1, ld_bytes, it_b, rows_b, sink, nwg);
    const double b2 = (double)nwg * it_b * 16384.0;
    printf("mode 2 (own 16-MiB region per workgroup, streamed once from HBM): 8 in flight %.2f TB/s | 14 in flight %.2f TB/s\n", b2 / t8 / 1e9, b2 / t14 / 1e9);
  }
  return 0;
}
