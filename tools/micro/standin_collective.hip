// Stand-in for an RCCL ring all-reduce kernel on ONE GPU (tools/standin_sweep.py; VERDICT r4 #4 iii): no multi-GPU node was ever
// available to this build, so what an overlapping collective does to the backward is rehearsed with a kernel that has the
// collective's footprint and none of its links: `wgs` persistent workgroups of 256 threads (RCCL channels), each streaming its
// share of the gradient bucket through the CU (read + write in place, x * 1.0f: twice the bucket in HBM traffic, what a ring
// step costs the local memory), PACED against the 100 MHz wall clock so that the bucket takes bytes * 2 (n - 1) / n / busbw
// seconds -- the time xGMI would need -- however fast HBM could serve it.  FOOTPRINT = RCCL's own: the gfx950 code object of
// this image's librccl.so (torch/lib, RCCL 2.26.6) holds rcclGenericKernel<1|2|4, *> with 256 threads per workgroup,
// 19 744 bytes of LDS and 261-280 VGPRs (17-32 of them AGPRs) -- read from its .amdgpu_metadata notes -- so a channel cannot
// share a CU with a ping-pong GEMM workgroup (160 KiB of LDS, 2 x 232 VGPRs per SIMD) or with the pair-pipelined attention
// backward: the stand-in allocates the same LDS and clobbers v255 / a7 so that its descriptor asks for 264 registers.
// Every workgroup reaches the end of its share
// (bounded loop, bounded waits: at most `ticks_total` ticks past its start), so the grid always drains.
// Build (tools/standin_sweep.py does it): hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libstandin.so standin_collective.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void standin_kernel(float* buf, long n4, long chunk4, unsigned long long ticks_per_chunk) {
  __shared__ float lds[19744 / 4];
  lds[threadIdx.x] = 0.f;                                  // the allocation is what matters: a resident channel's LDS
  asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a7, 0" ::: "v255", "a7");  // 256 + 8 registers in the descriptor
  const unsigned long long t0 = wall_clock64();
  const long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  f32x4* p = (f32x4*)buf;
  long done = 0;
  for (long c = lo; c < hi; c += chunk4) {
    const long e = c + chunk4 < hi ? c + chunk4 : hi;
    for (long i = c + threadIdx.x; i < e; i += 256) {
      f32x4 v = p[i];
      v *= 1.0f;
      p[i] = v;
    }
    ++done;
    // pace: chunk k may not end before t0 + k * ticks_per_chunk (bounded: the deadline is a fixed time, not a condition on
    // other workgroups)
    const unsigned long long deadline = t0 + (unsigned long long)done * ticks_per_chunk;
    while (wall_clock64() < deadline) __builtin_amdgcn_s_sleep(32);
  }
  if (lds[threadIdx.x] != 0.f) buf[0] = 0.f;  // never true: keeps the LDS array alive
}

extern "C" int standin_launch(void* buf, long n_floats, int wgs, double seconds, void* stream) {
  if (!buf || n_floats < 4 || wgs < 1) return -1;
  const long n4 = n_floats / 4;
  const long per = (n4 + wgs - 1) / wgs;
  const long chunk4 = 4096;  // 64 KiB per workgroup and pacing step
  const long chunks = (per + chunk4 - 1) / chunk4;
  const double ticks = seconds * 100e6 / (double)(chunks > 0 ? chunks : 1);  // wall_clock64: 100 MHz on gfx950
  hipLaunchKernelGGL(standin_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (float*)buf, n4, chunk4,
                     (unsigned long long)(ticks > 1.0 ? ticks : 1.0));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
