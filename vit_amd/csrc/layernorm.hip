// LayerNorm forward / backward for gfx950: HBM-bound row kernels, one wave (64 lanes) per row.
// x is the fp32 residual stream; statistics, affine and the backward sums are fp32 (eps = 1e-12 is far below bf16
// resolution: SURVEY.md section 7).  Rows stay in registers between the statistics and the normalisation, so x is
// read from HBM exactly once per pass.  Cross-lane sums use wave shuffles (DPP/permute), no LDS.
#include <algorithm>

#include "common.h"

#ifndef VIT_LNF_BLOCKS4
#define VIT_LNF_BLOCKS4 1792
#endif

namespace vit {

void* ctx_workspace(vit_handle h, size_t* bytes);

// NV = float4 chunks per lane held in registers; supports D <= 256*NV
// RES: 0 = plain; 1 / 2 = the row normalised is x + delta (delta bf16 / f32: the projection underneath a
// "dropout(Linear(.)) + residual"), and the sum is also written to xsum -- the new residual stream.
template <int NV, int OUT_BF16, int RES>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int D, float eps, const void* __restrict__ delta,
                                                     float* __restrict__ xsum) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nvec = D >> 2;
  const float invD = 1.0f / (float)D;
  for (int row = wave; row < rows; row += nwaves) {
    const float* xr = x + (long)row * D;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < nvec) ? *(const f32x4*)(xr + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
      if (RES && c < nvec) {
        if (RES == 1) {
          const bf16x4 t = *(const bf16x4*)((const short*)delta + (long)row * D + 4 * c);
          v[i] += (f32x4){bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
        } else {
          v[i] += *(const f32x4*)((const float*)delta + (long)row * D + 4 * c);
        }
        *(f32x4*)(xsum + (long)row * D + 4 * c) = v[i];
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 d = v[i] - mu;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
    const float rs = rsqrtf(wave_sum(q) * invD + eps);
    if (lane == 0) {
      if (mean) mean[row] = mu;
      if (rstd) rstd[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        const f32x4 g = *(const f32x4*)(gamma + 4 * c);
        const f32x4 b = *(const f32x4*)(beta + 4 * c);
        const f32x4 o = (v[i] - mu) * rs * g + b;
        if (OUT_BF16) {
          u32x2 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
          *(u32x2*)((short*)y + (long)row * D + 4 * c) = pk;
        } else {
          *(f32x4*)((float*)y + (long)row * D + 4 * c) = o;
        }
      }
    }
  }
}

// backward: dx = rstd * (g - mean(g) - xhat * mean(g*xhat)) + dres,  g = dy*gamma, xhat = (x-mean)*rstd
// per-wave running sums of dy*xhat (dgamma) and dy (dbeta) over the rows the wave visits; block-combined through LDS
// and written as one partial row per block: part[blk][0][D] (dgamma), part[blk][1][D] (dbeta).
// FUSE: additionally emit dyn = dropout_mask(drop) * dx as bf16 (the gradient wrt the Linear output that sits under the
// "dropout(.) + residual" this LayerNorm's input came from) and its column sums (that Linear's bias gradient) as a
// third partial row -- what vit_dropout_bwd_cast + vit_colsum would otherwise re-read dx for.
// FUSE: 0 = plain, 1 = + bf16 dyn, 2 = + f32 dyn
template <int NV, int DY_BF16, int FUSE>
__global__ __launch_bounds__(1024) void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, float* __restrict__ part, int rows, int D,
                                                     void* __restrict__ dyn, DropCfg drop) {
  resolve_drop(drop);
  constexpr int NP = FUSE ? 3 : 2;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][NP][D], used by four waves at a time
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nvec = D >> 2;
  const float invD = 1.0f / (float)D;
  // gamma: in registers up to D = 768; from D = 1024 (NV = 4, ViT-L) in the LDS behind the reduction buffer -- with three
  // accumulator sets of NV x 4 registers the fused form spilled 14 registers per lane at 128 VGPRs (4 waves per SIMD) and its
  // scratch traffic sat inside the row loop (r03: 4.2 TB/s at ViT-L against 6.0 TB/s at ViT-B)
  constexpr bool GAM_LDS = NV >= 4 && FUSE != 0;
  float* gam_s = red + 4 * NP * D;
  f32x4 gam[GAM_LDS ? 1 : NV], dg[NV], db[NV], dbias[FUSE ? NV : 1];
  if (GAM_LDS) {
    for (int i = threadIdx.x; i < nvec; i += blockDim.x) *(f32x4*)(gam_s + 4 * i) = *(const f32x4*)(gamma + 4 * i);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (!GAM_LDS) gam[i] = (c < nvec) ? *(const f32x4*)(gamma + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    dg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    db[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (FUSE) dbias[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int row = wave; row < rows; row += nwaves) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[NV], g[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 d;
        if (DY_BF16) {
          bf16x4 t = *(const bf16x4*)((const short*)dy + (long)row * D + 4 * c);
          d = (f32x4){bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
        } else {
          d = *(const f32x4*)((const float*)dy + (long)row * D + 4 * c);
        }
        xh[i] = (*(const f32x4*)(x + (long)row * D + 4 * c) - mu) * rs;
        g[i] = d * (GAM_LDS ? *(const f32x4*)(gam_s + 4 * c) : gam[GAM_LDS ? 0 : i]);
        dg[i] += d * xh[i];
        db[i] += d;
        s1 += (g[i][0] + g[i][1]) + (g[i][2] + g[i][3]);
        const f32x4 t2 = g[i] * xh[i];
        s2 += (t2[0] + t2[1]) + (t2[2] + t2[3]);
      } else {
        xh[i] = g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        f32x4 o = (g[i] - c1 - xh[i] * c2) * rs;
        if (dres) o += *(const f32x4*)(dres + (long)row * D + 4 * c);
        *(f32x4*)(dx + (long)row * D + 4 * c) = o;
        if (FUSE) {
          if (drop.thr) {
            float k0, k1, k2, k3;
            drop_pair(drop, (unsigned long long)row, (unsigned)(D >> 1), (unsigned)(4 * c), k0, k1);
            drop_pair(drop, (unsigned long long)row, (unsigned)(D >> 1), (unsigned)(4 * c) + 2, k2, k3);
            o[0] *= k0; o[1] *= k1; o[2] *= k2; o[3] *= k3;
          }
          if (FUSE == 1) {
            u32x2 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
            *(u32x2*)((short*)dyn + (long)row * D + 4 * c) = pk;
            // sum what was stored (bf16-rounded), exactly like colsum over the bf16 tensor would
            dbias[i] += (f32x4){bf2f((short)(pk[0] & 0xFFFF)), bf2f((short)(pk[0] >> 16)), bf2f((short)(pk[1] & 0xFFFF)),
                                bf2f((short)(pk[1] >> 16))};
          } else {
            *(f32x4*)((float*)dyn + (long)row * D + 4 * c) = o;
            dbias[i] += o;
          }
        }
      }
    }
  }
  // block combine, four waves at a time through the same 4 x NP x D floats (a 16-wave block = one partial row per CU: 256
  // rows for the reducer instead of 1024, no first reduction stage); each thread owns the columns tid, tid + blockDim, ...
  const int nw = blockDim.x >> 6, ngrp = (nw + 3) >> 2;
  constexpr int KMAX = 16;  // columns per thread: NP * D <= 4096 = 16 * 256 at every shape the dispatch accepts
  float tot[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) tot[k] = 0.f;
  for (int gq = 0; gq < ngrp; ++gq) {
    if ((wib >> 2) == gq) {
      const int w4 = wib & 3;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
          *(f32x4*)(red + (w4 * NP + 0) * D + 4 * c) = dg[i];
          *(f32x4*)(red + (w4 * NP + 1) * D + 4 * c) = db[i];
          if (FUSE) *(f32x4*)(red + (w4 * NP + 2) * D + 4 * c) = dbias[i];
        }
      }
    }
    __syncthreads();
    const int nw4 = min(4, nw - gq * 4);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int i = threadIdx.x + k * blockDim.x;
      if (i < NP * D) {
        float a = 0.f;
        for (int w = 0; w < nw4; ++w) a += red[w * NP * D + i];
        tot[k] += a;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int i = threadIdx.x + k * blockDim.x;
    if (i < NP * D) part[(long)blockIdx.x * NP * D + i] = tot[k];
  }
}

template <int OUT_BF16, int RES>
static int ln_fwd_dispatch(const float* x, const float* g, const float* b, void* y, float* mean, float* rstd, int rows,
                           int D, float eps, hipStream_t st, const void* delta = nullptr, float* xsum = nullptr) {
  const int nv = cdiv(D, 256);
  // one round of resident 4-wave blocks: 8 per CU up to D = 768 (<= 64 VGPRs), 7 at D = 1024 (72 VGPRs)
  const int blocks = std::min(cdiv(rows, 4), nv >= 4 ? VIT_LNF_BLOCKS4 : 2048);
#define LAUNCH(NV) hipLaunchKernelGGL((ln_fwd_kernel<NV, OUT_BF16, RES>), dim3(blocks), dim3(256), 0, st, x, g, b, y, mean, rstd, rows, D, eps, delta, xsum)
  if (nv <= 1) LAUNCH(1);
  else if (nv <= 2) LAUNCH(2);
  else if (nv <= 3) LAUNCH(3);
  else if (nv <= 4) LAUNCH(4);
  else if (nv <= 8) LAUNCH(8);
  else { set_error("vit_layernorm_fwd: D=%d > 2048 not supported", D); return VIT_ERR_UNSUPPORTED; }
#undef LAUNCH
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

template <int DY_BF16, int FUSE>
static int ln_bwd_dispatch(const void* dy, const float* x, const float* g, const float* mean, const float* rstd,
                           const float* dres, float* dx, float* part, int rows, int D, int blocks, int threads,
                           void* dyn, DropCfg drop, hipStream_t st) {
  const int nv = cdiv(D, 256);
  const size_t sh = (size_t)4 * (FUSE ? 3 : 2) * D * sizeof(float) + (FUSE && nv >= 4 ? (size_t)D * sizeof(float) : 0);
  if (sh > 64 * 1024) { set_error("vit_layernorm_bwd: D=%d needs %zu bytes of LDS", D, sh); return VIT_ERR_UNSUPPORTED; }
#define LAUNCH(NV) hipLaunchKernelGGL((ln_bwd_kernel<NV, DY_BF16, FUSE>), dim3(blocks), dim3(threads), sh, st, dy, x, g, mean, rstd, dres, dx, part, rows, D, dyn, drop)
  if (nv <= 1) LAUNCH(1);
  else if (nv <= 2) LAUNCH(2);
  else if (nv <= 3) LAUNCH(3);
  else if (nv <= 4) LAUNCH(4);
  else if (nv <= 8 && !FUSE) LAUNCH(8);
  else { set_error("vit_layernorm_bwd: D=%d not supported (max 2048, 1024 for the fused form)", D); return VIT_ERR_UNSUPPORTED; }
#undef LAUNCH
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

static int ln_bwd_common(vit_handle h, const void* dy, int dy_dtype, const float* x, const float* gamma,
                         const float* mean, const float* rstd, const float* dres, float* dx, float* dgamma,
                         float* dbeta, int rows, int D, void* dyn, int dyn_dtype, float* dbias, DropCfg drop,
                         hipStream_t st) {
  // 16 waves per CU: with only two waves per SIMD the pass was latency-bound at 4.3 TB/s (bytes in flight / HBM latency).
  // They are TWO blocks of 512 threads per CU when there are rows for it: a block's waves combine through the LDS four at a
  // time (36 KiB at D = 768), so the pass leaves 512 partial rows instead of 1024 and the reducer needs no first stage (25
  // launches of 7 us per step).  Pass + reducers on one box: 4-wave blocks and two reducer stages ~123 us, one 16-wave block
  // per CU 126.1 us (the pass itself got 5 % slower), 8-wave blocks 122.0 us.  Small inputs keep 4-wave blocks.
  const int np = dyn ? 3 : 2;
  const bool big = rows >= 256 * 16;
  const int threads = big ? 512 : 256;
  const int blocks = big ? 512 : std::min(cdiv(rows, 16), 1024);
  size_t wsb = 0;
  float* part = (float*)ctx_workspace(h, &wsb);
  const size_t need = (size_t)blocks * np * D * sizeof(float);
  VIT_CHECK(part && wsb >= need, VIT_ERR_WORKSPACE, "vit_layernorm_bwd: needs %zu workspace bytes, have %zu", need, wsb);
  int rc;
  if (dyn && dyn_dtype == VIT_BF16)
    rc = dy_dtype == VIT_BF16 ? ln_bwd_dispatch<1, 1>(dy, x, gamma, mean, rstd, dres, dx, part, rows, D, blocks, threads, dyn, drop, st)
                              : ln_bwd_dispatch<0, 1>(dy, x, gamma, mean, rstd, dres, dx, part, rows, D, blocks, threads, dyn, drop, st);
  else if (dyn)
    rc = dy_dtype == VIT_BF16 ? ln_bwd_dispatch<1, 2>(dy, x, gamma, mean, rstd, dres, dx, part, rows, D, blocks, threads, dyn, drop, st)
                              : ln_bwd_dispatch<0, 2>(dy, x, gamma, mean, rstd, dres, dx, part, rows, D, blocks, threads, dyn, drop, st);
  else rc = dy_dtype == VIT_BF16 ? ln_bwd_dispatch<1, 0>(dy, x, gamma, mean, rstd, dres, dx, part, rows, D, blocks, threads, dyn, drop, st)
                                 : ln_bwd_dispatch<0, 0>(dy, x, gamma, mean, rstd, dres, dx, part, rows, D, blocks, threads, dyn, drop, st);
  if (rc != VIT_OK) return rc;
  if (!dyn) return launch_reduce_partials(part, blocks, 2 * D, dgamma, D, dbeta, 0, st, np * D);
  return launch_reduce_partials(part, blocks, 3 * D, dgamma, D, dbeta, 0, st, np * D, 2 * D, dbias);  // one launch for all three
}

}  // namespace vit

extern "C" {

int vit_layernorm_fwd(vit_handle h, const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                      float* mean, float* rstd, int rows, int D, float eps, vit_stream stream) {
  using namespace vit;
  (void)h;
  VIT_CHECK(x && gamma && beta && y, VIT_ERR_ARG, "vit_layernorm_fwd: null pointer");
  VIT_CHECK(rows > 0 && D > 0 && (D % 4) == 0, VIT_ERR_ARG, "vit_layernorm_fwd: rows=%d D=%d (D must be a multiple of 4)", rows, D);
  VIT_CHECK(y_dtype == VIT_BF16 || y_dtype == VIT_F32, VIT_ERR_ARG, "vit_layernorm_fwd: bad y_dtype");
  hipStream_t st = (hipStream_t)stream;
  return y_dtype == VIT_BF16 ? ln_fwd_dispatch<1, 0>(x, gamma, beta, y, mean, rstd, rows, D, eps, st)
                             : ln_fwd_dispatch<0, 0>(x, gamma, beta, y, mean, rstd, rows, D, eps, st);
}

int vit_layernorm_fwd_residual(vit_handle h, const float* x, const void* delta, int delta_dtype, float* xsum,
                               const float* gamma, const float* beta, void* y, int y_dtype, float* mean, float* rstd,
                               int rows, int D, float eps, vit_stream stream) {
  using namespace vit;
  (void)h;
  VIT_CHECK(x && delta && xsum && gamma && beta && y, VIT_ERR_ARG, "vit_layernorm_fwd_residual: null pointer");
  VIT_CHECK(rows > 0 && D > 0 && (D % 4) == 0, VIT_ERR_ARG, "vit_layernorm_fwd_residual: rows=%d D=%d (D must be a multiple of 4)", rows, D);
  VIT_CHECK((y_dtype == VIT_BF16 || y_dtype == VIT_F32) && (delta_dtype == VIT_BF16 || delta_dtype == VIT_F32), VIT_ERR_ARG,
            "vit_layernorm_fwd_residual: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  if (delta_dtype == VIT_BF16)
    return y_dtype == VIT_BF16 ? ln_fwd_dispatch<1, 1>(x, gamma, beta, y, mean, rstd, rows, D, eps, st, delta, xsum)
                               : ln_fwd_dispatch<0, 1>(x, gamma, beta, y, mean, rstd, rows, D, eps, st, delta, xsum);
  return y_dtype == VIT_BF16 ? ln_fwd_dispatch<1, 2>(x, gamma, beta, y, mean, rstd, rows, D, eps, st, delta, xsum)
                             : ln_fwd_dispatch<0, 2>(x, gamma, beta, y, mean, rstd, rows, D, eps, st, delta, xsum);
}

int vit_layernorm_bwd(vit_handle h, const void* dy, int dy_dtype, const float* x, const float* gamma,
                      const float* mean, const float* rstd, const float* dres, float* dx, float* dgamma,
                      float* dbeta, int rows, int D, vit_stream stream) {
  using namespace vit;
  VIT_CHECK(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, VIT_ERR_ARG, "vit_layernorm_bwd: null pointer");
  VIT_CHECK(rows > 0 && D > 0 && (D % 4) == 0, VIT_ERR_ARG, "vit_layernorm_bwd: rows=%d D=%d", rows, D);
  return ln_bwd_common(h, dy, dy_dtype, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, D, nullptr, VIT_BF16, nullptr,
                       make_drop(0.f, 0, 0), (hipStream_t)stream);
}

int vit_layernorm_bwd_fused(vit_handle h, const void* dy, int dy_dtype, const float* x, const float* gamma,
                            const float* mean, const float* rstd, const float* dres, float* dx, float* dgamma,
                            float* dbeta, int rows, int D, void* dyn, int dyn_dtype, float* dbias, float dropout_p,
                            uint64_t seed, uint64_t site, vit_stream stream) {
  using namespace vit;
  VIT_CHECK(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && dyn && dbias, VIT_ERR_ARG,
            "vit_layernorm_bwd_fused: null pointer");
  VIT_CHECK(rows > 0 && D > 0 && (D % 4) == 0, VIT_ERR_ARG, "vit_layernorm_bwd_fused: rows=%d D=%d", rows, D);
  VIT_CHECK(dropout_p >= 0.f && dropout_p < 1.f, VIT_ERR_ARG, "vit_layernorm_bwd_fused: dropout_p out of [0,1)");
  return ln_bwd_common(h, dy, dy_dtype, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, D, dyn, dyn_dtype, dbias,
                       make_drop_h(h, dropout_p, seed, site), (hipStream_t)stream);
}

}  // extern "C"
