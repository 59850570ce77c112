// VALU issue rates on gfx950: cycles per wave64 instruction for the integer / transcendental ops the dropout hash and the
// softmax use (one wave per SIMD, 8 independent chains, s_memtime around 8 x 512 instructions).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(unsigned* out, unsigned long long* cyc, unsigned seed, int iters) {
  unsigned a[8];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed * (i + 3) + threadIdx.x; f[i] = (float)(a[i] & 1023) * 0.001f; }
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#define STEP(i)                                                                                              \
  if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));                             \
  if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(seed));                            \
  if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));                                \
  if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));                                                \
  if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));                        \
  if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));                  \
  if (OP == 6) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));                             \
  if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));                                                \
  if (OP == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "+v"(a[i]) : "v"(f[i]));                        \
  if (OP == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&f[i & 6]));
    REP8(STEP)
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + __float_as_uint(f[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  unsigned* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const char* names[] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_xor_b32", "v_exp_f32", "v_mad_u32_u24", "v_fma_f32", "v_mul_hi_u32", "v_rcp_f32", "v_cvt_pk_bf16_f32", "v_pk_fma_f32"};
  for (int waves = 1; waves <= 4; ++waves) {
    for (int op = 0; op < 10; ++op) {
      unsigned long long h = 0;
      float ms = 0.f;
      const int iters = 1 << 16;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        dim3 g(256), b(256 * waves);
        switch (op) {
#define C(N) case N: hipLaunchKernelGGL(k<N>, g, b, 0, 0, out, cyc, 12345u, iters); break;
          C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9)
        }
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      }
      printf("%d wave(s)/SIMD  %-18s %6.2f s_memtime ticks, %6.2f ns per instruction per wave\n", waves, names[op], (double)h / (8.0 * iters), ms * 1e6 / (8.0 * iters));
    }
  }
  return 0;
}
