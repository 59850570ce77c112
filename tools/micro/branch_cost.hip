// What does a taken backward branch cost on gfx950?  A loop whose body is BODY independent VALU instructions (an
// .rept block, so the code is really that long), 1 or 2 waves per SIMD, all CUs busy: cycles per iteration minus the cycles
// the same instructions take inside a longer body = the price of the back edge.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/branch_cost.hip -o tools/micro/branch_cost && tools/micro/branch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BODY>
__global__ void k(unsigned* out, unsigned long long* cyc, int iters) {
  unsigned a = threadIdx.x, b = blockIdx.x + 7, c = 3, d = 5;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if (BODY == 64) asm volatile(".rept 16\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (BODY == 256) asm volatile(".rept 64\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (BODY == 1024) asm volatile(".rept 256\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    if (BODY == 4096) asm volatile(".rept 1024\n\tv_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  unsigned* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4 << 20); (void)hipMalloc(&cyc, 8);
  for (int waves = 1; waves <= 2; ++waves)
    for (int body : {64, 256, 1024, 4096}) {
      const int total = 1 << 18, iters = total / body;
      unsigned long long h = 0;
      for (int rep = 0; rep < 2; ++rep) {
        dim3 g(256), b(256 * waves);
        if (body == 64) hipLaunchKernelGGL(k<64>, g, b, 0, 0, out, cyc, iters);
        if (body == 256) hipLaunchKernelGGL(k<256>, g, b, 0, 0, out, cyc, iters);
        if (body == 1024) hipLaunchKernelGGL(k<1024>, g, b, 0, 0, out, cyc, iters);
        if (body == 4096) hipLaunchKernelGGL(k<4096>, g, b, 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      }
      printf("%d wave(s)/SIMD, body %4d instructions (%5d B): %8.1f cycles per iteration = %5.2f per instruction\n", waves, body, body * 4,
             (double)h / iters, (double)h / iters / body);
    }
  return 0;
}
