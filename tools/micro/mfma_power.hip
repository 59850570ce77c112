// Sustained dense-MFMA rate, clock and package power of the two bf16 MFMA shapes with NOTHING else in the kernel (registers only):
// is the 2.5 PFLOP/s dense bf16 peak reachable inside the 1 400 W cap, and does the block shape matter?
//   v_mfma_f32_16x16x32_bf16 (8 passes, 16 K FLOP)   vs   v_mfma_f32_32x32x16_bf16 (16 passes, 32 K FLOP)
// 256 workgroups x 512 threads (2 waves per SIMD, the GEMM core's occupancy), 4 independent accumulator chains per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip ; run: ./mfma_power [seconds per shape]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// RANDOM = 1: eight operand pairs of pseudo-random bf16 values in [-2, 2), a different pair per MFMA (operand buses and
// multiplier inputs toggle as they do on real data); RANDOM = 0: one constant pair (nothing toggles: the floor of MFMA power)
__device__ __forceinline__ bf16x8 rnd8(unsigned s) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    r[i] = (short)(((s >> 16) & 0x807F) | 0x3F80);  // sign + 7 mantissa bits, exponent of 1.0: values in +-[1, 2)
  }
  return r;
}

template <int SHAPE, int RANDOM>
__global__ __launch_bounds__(512) void mfma_kernel(float* out, int iters) {
  bf16x8 A[8], Bv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    A[i] = RANDOM ? rnd8(threadIdx.x * 977u + blockIdx.x * 31u + i * 7919u) : (bf16x8){1, 2, 3, 4, 5, 6, 7, 8};
    Bv[i] = RANDOM ? rnd8(threadIdx.x * 613u + blockIdx.x * 17u + i * 104729u + 5u) : (bf16x8){7, 6, 5, 4, 3, 2, 1, 9};
  }
#define a A
#define b Bv
  if (SHAPE == 16) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    for (int i = 0; i < iters; ++i) {
      // RANDOM == 2: the A operand stays for four consecutive MFMAs (a GEMM wave sweeping the N sub-tiles of one A fragment)
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RANDOM == 2 ? 0 : 1], b[1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RANDOM == 2 ? 0 : 2], b[2], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RANDOM == 2 ? 0 : 3], b[3], c3, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[4], b[RANDOM == 2 ? 0 : 4], c4, 0, 0, 0);
      c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RANDOM == 2 ? 4 : 5], b[RANDOM == 2 ? 1 : 5], c5, 0, 0, 0);
      c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RANDOM == 2 ? 4 : 6], b[RANDOM == 2 ? 2 : 6], c6, 0, 0, 0);
      c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[RANDOM == 2 ? 4 : 7], b[RANDOM == 2 ? 3 : 7], c7, 0, 0, 0);
    }
    f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s[0] == 12345.f) out[0] = s[1];
  } else {
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[3], b[3], c3, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[4], b[4], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[5], b[5], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[6], b[6], c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[7], b[7], c3, 0, 0, 0);
    }
    f32x16 s = c0 + c1 + c2 + c3;
    if (s[0] == 12345.f) out[0] = s[1];
  }
}

#undef a
#undef b
template <int SHAPE, int RANDOM>
static void run(double seconds, float* out) {
  const int iters = 20000;
  const double flop_per_launch = 256.0 * 8 * iters * (SHAPE == 16 ? 8 * 2.0 * 16 * 16 * 32 : 8 * 2.0 * 32 * 32 * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_kernel<SHAPE, RANDOM>), dim3(256), dim3(512), 0, 0, out, iters);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int launches = 0; double ms_total = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipEventRecord(e0);
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((mfma_kernel<SHAPE, RANDOM>), dim3(256), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms_total += ms; launches += 20;
  }
  printf("shape %s, %s operands: %.1f TFLOP/s sustained over %.1f s (%d launches)\n", SHAPE == 16 ? "16x16x32" : "32x32x16",
         RANDOM == 2 ? "random, A held for 4 MFMAs" : (RANDOM ? "random" : "constant"),
         flop_per_launch * launches / (ms_total * 1e-3) / 1e12, ms_total * 1e-3, launches);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 6.0;
  float* out; hipMalloc(&out, 1024);
  run<16, 0>(secs, out);
  run<32, 0>(secs, out);
  run<16, 1>(secs, out);
  run<32, 1>(secs, out);
  run<16, 2>(secs, out);
  return 0;
}
