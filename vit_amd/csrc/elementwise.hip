// HBM-bound side kernels of the ViT step for gfx950: unfold+cast, embedding finish (CLS / pos-emb / dropout) and its
// backward, dropout-backward cast, deterministic column sums (bias gradients), head + loss, global grad norm, AdamW.
// All are vectorised (8-16 B per lane), grid-stride, and free of float atomics (reference: deterministic=True).
#include <algorithm>

#include "common.h"

namespace vit {

void* ctx_workspace(vit_handle h, size_t* bytes);

static inline int grid_for(long n, int block = 256, int cap = 4096) {
  return (int)std::max<long>(1, std::min<long>((n + block - 1) / block, cap));
}

// ------------------------------------------------------------------------------------------ unfold + cast
// patches[(b*N+n)*P + p] = bf16(x[b*L + n*S + p]); a window that does not fit entirely inside the signal is ALL zero:
// Tensor.unfold drops partial windows and the reference appends whole zero patches (tokenization.py:45-49)
template <int OUT_BF16>
__global__ void unfold_cast_kernel(const float* __restrict__ x, void* __restrict__ out, int B, int L, int P, int S,
                                   int N) {
  const long total = (long)B * N * (P >> 2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int pv = (int)(i % (P >> 2)) << 2;
    const long bn = i / (P >> 2);
    const int n = (int)(bn % N);
    const long b = bn / N;
    const int pos = n * S + pv;
    const float* src = x + b * L + pos;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (n * S + P <= L) ? src[k] : 0.f;
    if (OUT_BF16) {
      u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *(u32x2*)((short*)out + bn * P + pv) = pk;
    } else {
      *(f32x4*)((float*)out + bn * P + pv) = (f32x4){v[0], v[1], v[2], v[3]};
    }
  }
}

// ------------------------------------------------------------------------------------------ embedding finish
__global__ void embed_finish_kernel(float* __restrict__ tok, const float* __restrict__ cls,
                                    const float* __restrict__ pos, int B, int T, int D, DropCfg drop) {
  resolve_drop(drop);
  const int dv = D >> 2;
  const long total = (long)B * T * dv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % dv) << 2;
    const long row = i / dv;  // b*T + t
    const int t = (int)(row % T);
    float* p = tok + row * D + d;
    f32x4 v = (t == 0) ? *(const f32x4*)(cls + d) : *(const f32x4*)p;
    if (pos) v += *(const f32x4*)(pos + (long)t * D + d);
    if (drop.thr) {
      float k0, k1, k2, k3;
      drop_pair(drop, (unsigned long long)row, (unsigned)(D >> 1), (unsigned)d, k0, k1);
      drop_pair(drop, (unsigned long long)row, (unsigned)(D >> 1), (unsigned)d + 2, k2, k3);
      v[0] *= k0; v[1] *= k1; v[2] *= k2; v[3] *= k3;
    }
    *(f32x4*)p = v;
  }
}

// one thread per (t, 4 columns): walks the batch, applies the embedding-dropout mask, emits the bf16 gradient of the
// patch projection output (rows t >= 1) and the batch-summed gradients of cls_token (t == 0) / position_embeddings.
template <int OUT_BF16>
__global__ void embed_finish_bwd_kernel(const float* __restrict__ dtok, void* __restrict__ dpatch,
                                        float* __restrict__ part, int B, int T, int D, DropCfg drop, int bchunk) {
  resolve_drop(drop);
  // blockIdx.y walks a chunk of the batch; its (t, 4 columns) sums go to part[blockIdx.y][T*D] (reduced afterwards in a
  // fixed order: deterministic) -- one thread per (t, 4 columns) walking the WHOLE batch left 40 % of the CUs idle and
  // serialised 256 loads per thread (205 us at B = 256)
  const int dv = D >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * dv) return;
  const int d = (i % dv) << 2;
  const int t = i / dv;
  const int N = T - 1;
  const int b0 = blockIdx.y * bchunk, b1 = min(B, b0 + bchunk);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int b = b0; b < b1; ++b) {
    const long row = (long)b * T + t;
    f32x4 v = *(const f32x4*)(dtok + row * D + d);
    if (drop.thr) {
      float k0, k1, k2, k3;
      drop_pair(drop, (unsigned long long)row, (unsigned)(D >> 1), (unsigned)d, k0, k1);
      drop_pair(drop, (unsigned long long)row, (unsigned)(D >> 1), (unsigned)d + 2, k2, k3);
      v[0] *= k0; v[1] *= k1; v[2] *= k2; v[3] *= k3;
    }
    acc += v;
    if (t > 0) {
      if (OUT_BF16) {
        u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(u32x2*)((short*)dpatch + ((long)b * N + (t - 1)) * D + d) = pk;
      } else {
        *(f32x4*)((float*)dpatch + ((long)b * N + (t - 1)) * D + d) = v;
      }
    }
  }
  *(f32x4*)(part + ((long)blockIdx.y * T + t) * D + d) = acc;
}

// ------------------------------------------------------------------------------------------ dropout bwd + cast
template <int OUT_BF16>
__global__ void dropout_bwd_cast_kernel(const float* __restrict__ dx, void* __restrict__ dy, long rows, int cols,
                                        DropCfg drop) {
  resolve_drop(drop);
  const int cv = cols >> 2;
  const long total = rows * cv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) << 2;
    const long row = i / cv;
    f32x4 v = *(const f32x4*)(dx + row * cols + c);
    if (drop.thr) {
      float k0, k1, k2, k3;
      drop_pair(drop, (unsigned long long)row, (unsigned)(cols >> 1), (unsigned)c, k0, k1);
      drop_pair(drop, (unsigned long long)row, (unsigned)(cols >> 1), (unsigned)c + 2, k2, k3);
      v[0] *= k0; v[1] *= k1; v[2] *= k2; v[3] *= k3;
    }
    if (OUT_BF16) {
      u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *(u32x2*)((short*)dy + row * cols + c) = pk;
    } else {
      *(f32x4*)((float*)dy + row * cols + c) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------ column sums
// stage 1: block (64 x 4): 64 lanes x 4 columns each = 256 columns, 4 row-lanes; grid.y row chunks
template <int BF16>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const void* __restrict__ a, long lda, float* __restrict__ part,
                                                            int rows, int cols) {
  __shared__ f32x4 red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + tx) << 2;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    for (int r = blockIdx.y * 4 + ty; r < rows; r += gridDim.y * 4) {
      if (BF16) {
        bf16x4 t = *(const bf16x4*)((const short*)a + (long)r * lda + c);
        acc += (f32x4){bf2f(t[0]), bf2f(t[1]), bf2f(t[2]), bf2f(t[3])};
      } else {
        acc += *(const f32x4*)((const float*)a + (long)r * lda + c);
      }
    }
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && c < cols) {
    f32x4 s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    *(f32x4*)(part + (long)blockIdx.y * cols + c) = s;
  }
}
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ part, int nblk, int width,
                                                               float* __restrict__ out0, int split,
                                                               float* __restrict__ out1, int accumulate, int stride,
                                                               int split2, float* __restrict__ out2) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a0 = 0.f, a1 = 0.f;
  if (c < width) {
    // 8 rows per pass, all 8 loads issued before the first add: with two loads per pass the 512-row reductions of a step were
    // a chain of 16 L2 round trips (6.7 us per launch, ~50 launches per step; r03).  Fixed summation order per (ty, column).
    int k = ty;
    for (; k + 7 * 16 < nblk; k += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(k + u * 16) * stride + c];
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        a0 += v[u];
        a1 += v[u + 1];
      }
    }
    for (; k + 16 < nblk; k += 32) {
      a0 += part[(long)k * stride + c];
      a1 += part[(long)(k + 16) * stride + c];
    }
    if (k < nblk) a0 += part[(long)k * stride + c];
  }
  red[ty][tx] = a0 + a1;
  __syncthreads();
  if (ty == 0 && c < width) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][tx];
    // columns [0, split) -> out0, [split, split2) -> out1, [split2, width) -> out2
    float* o = c < split ? out0 + c : (c < split2 ? out1 + (c - split) : out2 + (c - split2));
    if (accumulate) s += *o;
    *o = s;
  }
}

// stage 1 of a tall reduction: blockIdx.y owns rows [y * chunk, (y + 1) * chunk) and leaves their sum IN its first row
// (only this block touches those rows of its 64 columns); 64 columns x 16 row groups per block like the final stage
__global__ __launch_bounds__(1024) void reduce_partials_stage1_kernel(float* __restrict__ part, int nblk, int width,
                                                                      int stride, int chunk) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * chunk, r1 = min(nblk, r0 + chunk);
  float a = 0.f;
  if (c < width) {
    int k = r0 + ty;
    for (; k + 3 * 16 < r1; k += 4 * 16) {  // four loads in flight, added in row order
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = part[(long)(k + u * 16) * stride + c];
#pragma unroll
      for (int u = 0; u < 4; ++u) a += v[u];
    }
    for (; k < r1; k += 16) a += part[(long)k * stride + c];
  }
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && c < width) {
    float s_ = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s_ += red[k][tx];
    part[(long)r0 * stride + c] = s_;
  }
}

int launch_reduce_partials(const float* part, int nblk, int width, float* out0, int split, float* out1, int accumulate,
                           hipStream_t st, int stride, int split2, float* out2) {
  if (!stride) stride = width;
  if (nblk > 512) {
    // a tall partial matrix (one row per wave of the attention backward, per block of the LayerNorm backward): 36 blocks
    // of one launch left most CUs idle; fold groups of rows first, in place, then reduce the group sums (fixed order)
    const int groups = std::min(32, nblk / 32), chunk = cdiv(nblk, groups);
    hipLaunchKernelGGL(reduce_partials_stage1_kernel, dim3(cdiv(width, 64), cdiv(nblk, chunk)), dim3(1024), 0, st,
                       const_cast<float*>(part), nblk, width, stride, chunk);
    VIT_LAUNCH_CHECK();
    nblk = cdiv(nblk, chunk);
    stride *= chunk;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(width, 64)), dim3(1024), 0, st, part, nblk, width, out0, split,
                     out1, accumulate, stride, split2 > 0 ? split2 : width, out2);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, short* __restrict__ dst, long n) {
  const long nv = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = *(const f32x4*)(src + 4 * i);
    u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *(u32x2*)(dst + 4 * i) = pk;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(nv << 2) + threadIdx.x] = f2bf(src[(nv << 2) + threadIdx.x]);
}

// dst (f32) = scale * src (bf16): the receive side of a bf16 gradient exchange (vit_amd/ddp.py)
__global__ void cast_bf16_f32_kernel(const short* __restrict__ src, float* __restrict__ dst, long n, float scale) {
  const long nv = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const u32x2 pk = *(const u32x2*)(src + 4 * i);
    f32x4 v = {__builtin_bit_cast(float, pk[0] << 16) * scale, __builtin_bit_cast(float, pk[0] & 0xFFFF0000u) * scale,
               __builtin_bit_cast(float, pk[1] << 16) * scale, __builtin_bit_cast(float, pk[1] & 0xFFFF0000u) * scale};
    *(f32x4*)(dst + 4 * i) = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(nv << 2) + threadIdx.x] = bf2f(src[(nv << 2) + threadIdx.x]) * scale;
}

// ------------------------------------------------------------------------------------------ head + loss
// one wave per sample: logits[b, c] = <last_hidden[b, 0, :], W[c, :]> + bias[c]
__global__ __launch_bounds__(64) void head_logits_kernel(const float* __restrict__ last, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ logits,
                                                         int T, int D, int C) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* x = last + (long)b * T * D;
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += x[d] * W[(long)c * D + d];
    s = wave_sum(s);
    if (lane == 0) logits[(long)b * C + c] = s + bias[c];
  }
}
// single block: mean loss over the batch (fixed summation order)
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ logits, const void* __restrict__ labels,
                                                   float* __restrict__ loss, int B, int C, int kind) {
  __shared__ float red[256];
  float a = 0.f;
  if (kind == VIT_LOSS_CE) {
    const long long* lab = (const long long*)labels;
    for (int b = threadIdx.x; b < B; b += 256) {
      const float* z = logits + (long)b * C;
      float mx = z[0];
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += __expf(z[c] - mx);
      a += (mx + __logf(se)) - z[lab[b]];
    }
  } else {
    const float* lab = (const float*)labels;
    for (int i = threadIdx.x; i < B * C; i += 256) {
      const float d = logits[i] - lab[i];
      a += (kind == VIT_LOSS_L1) ? fabsf(d) : d * d;
    }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0] / (float)(kind == VIT_LOSS_CE ? B : B * C);
}
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long n) {
  const long nv = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x)
    *(f32x4*)(p + 4 * i) = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(nv << 2) + threadIdx.x] = 0.f;
}
// dlogits (workspace, [B,C]) and the CLS rows of d(last_hidden); one wave per sample
__global__ __launch_bounds__(64) void head_bwd_rows_kernel(const float* __restrict__ W, const float* __restrict__ logits,
                                                           const void* __restrict__ labels,
                                                           const float* __restrict__ dloss, float* __restrict__ dlogits,
                                                           float* __restrict__ dlast, int B, int T, int D, int C,
                                                           int kind) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float up = dloss[0];
  const float* z = logits + (long)b * C;
  float mx = 0.f, se = 1.f;
  if (kind == VIT_LOSS_CE) {
    mx = z[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
    se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(z[c] - mx);
  }
  for (int c = 0; c < C; ++c) {
    float g;
    if (kind == VIT_LOSS_CE) {
      const long long y = ((const long long*)labels)[b];
      g = (__expf(z[c] - mx) / se - (c == y ? 1.f : 0.f)) / (float)B;
    } else {
      const float d = z[c] - ((const float*)labels)[(long)b * C + c];
      g = (kind == VIT_LOSS_L1) ? ((d > 0.f) - (d < 0.f)) / (float)(B * C) : 2.f * d / (float)(B * C);
    }
    g *= up;
    if (lane == 0) dlogits[(long)b * C + c] = g;
    for (int d = lane; d < D; d += 64) {
      float* o = dlast + (long)b * T * D + d;
      *o = (c == 0 ? 0.f : *o) + g * W[(long)c * D + d];
    }
  }
}
// dW[c, d] = sum_b dlogits[b,c] * cls[b,d];  db[c] = sum_b dlogits[b,c]
__global__ void head_bwd_params_kernel(const float* __restrict__ last, const float* __restrict__ dlogits,
                                       float* __restrict__ dW, float* __restrict__ db, int B, int T, int D, int C,
                                       int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * D) {
    const int c = i / D, d = i - c * D;
    float a = 0.f;
    int b = 0;
    for (; b + 7 < B; b += 8) {  // the CLS rows are T * D floats apart: eight misses in flight instead of one; sums in batch order
      float x[8], y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x[u] = dlogits[(long)(b + u) * C + c];
        y[u] = last[(long)(b + u) * T * D + d];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) a += x[u] * y[u];
    }
    for (; b < B; ++b) a += dlogits[(long)b * C + c] * last[(long)b * T * D + d];
    if (accumulate) a += dW[i];
    dW[i] = a;
  }
  if (i < C) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dlogits[(long)b * C + i];
    if (accumulate) a += db[i];
    db[i] = a;
  }
}

// ------------------------------------------------------------------------------------------ grad norm + AdamW
__global__ __launch_bounds__(256) void sqnorm_stage1_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float red[4];
  const long nv = n >> 2;
  float a = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(g + 4 * i);
    a += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(nv << 2) + threadIdx.x];
    a += v * v;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sqnorm_stage2_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out,
                                                            int accumulate) {
  __shared__ float red[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) a += part[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + red[0] : red[0];
}

// torch.optim.AdamW (single-tensor form): p *= 1 - lr*wd; m,v EMA; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    short* __restrict__ pb, long n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float rsqrt_bc2,
                                                    const float* __restrict__ sqnorm, float max_norm) {
  float clip = 1.f;
  if (sqnorm) clip = fminf(1.f, max_norm / (sqrtf(sqnorm[0]) + 1e-6f));
  const float step = lr / bc1, decay = 1.f - lr * wd;
  const long nv = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = *(const f32x4*)(p + 4 * i);
    const f32x4 gg = *(const f32x4*)(g + 4 * i) * clip;
    f32x4 mm = *(const f32x4*)(m + 4 * i);
    f32x4 vv = *(const f32x4*)(v + 4 * i);
    mm = mm * b1 + gg * (1.f - b1);
    vv = vv * b2 + gg * gg * (1.f - b2);
#pragma unroll
    for (int k = 0; k < 4; ++k) pp[k] = pp[k] * decay - step * mm[k] / (sqrtf(vv[k]) * rsqrt_bc2 + eps);
    *(f32x4*)(p + 4 * i) = pp;
    *(f32x4*)(m + 4 * i) = mm;
    *(f32x4*)(v + 4 * i) = vv;
    if (pb) {
      u32x2 pk = {pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3])};
      *(u32x2*)(pb + 4 * i) = pk;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (nv << 2) + threadIdx.x;
    const float gg = g[i] * clip;
    const float mm = m[i] * b1 + gg * (1.f - b1);
    const float vv = v[i] * b2 + gg * gg * (1.f - b2);
    const float pp = p[i] * decay - step * mm / (sqrtf(vv) * rsqrt_bc2 + eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (pb) pb[i] = f2bf(pp);
  }
}

// ---- backward of the tokenizer's unfold (tokenization.py:43-49 / the Conv1d windows of :66-69) for a TRAINABLE input
// preprocessor: dx[b, l] = sum over the windows n that cover l (n*S <= l < n*S + P, and the window lies entirely inside
// the signal -- the others were zero patches, not taken from x) of dpatches[b, n, l - n*S].  Gather form: deterministic.
__global__ __launch_bounds__(256) void fold_add_kernel(const float* __restrict__ dp, float* __restrict__ dx, int B, int L,
                                                       int P, int S, int N) {
  const long total = (long)B * L;
  const int nvalid = min(N, (L - P) / S + 1);  // windows that fit
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), l = (int)(i - (long)b * L);
    const int n_hi = min(nvalid - 1, l / S);
    const int n_lo = max(0, (l - P + S) / S);  // ceil((l - P + 1) / S) for l - P + 1 > 0
    float acc = 0.f;
    for (int n = n_lo; n <= n_hi; ++n) acc += dp[((long)b * N + n) * P + (l - n * S)];
    dx[i] = acc;
  }
}

// ---- training-time noise injection (ViTLModule.training_step, src/vit.py:86-88): out = flux + N(0,1) * error * level.
// The reference draws from torch's generator (implementation-defined stream); here a counter-based generator keyed on
// (seed, element index): two 32-bit hashes -> Box-Muller pair, 4 elements per work item, so the noise is reproducible
// for a given seed and independent of the launch geometry.
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ flux, const float* __restrict__ error,
                                                        float* __restrict__ out, long nvec, float level, unsigned k0,
                                                        unsigned k1) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const f32x4 f = *(const f32x4*)(flux + 4 * i), e = *(const f32x4*)(error + 4 * i);
    f32x4 z;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const unsigned h1 = drop_hash(k0, k1, (unsigned long long)(4 * i + 2 * p));
      const unsigned h2 = drop_hash(k0, k1, (unsigned long long)(4 * i + 2 * p + 1));
      const float u1 = ((float)(h1 >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
      const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);            // [0, 1)
      const float r = sqrtf(-2.0f * __logf(u1));
      float sn, cs;
      __sincosf(6.28318530717958647692f * u2, &sn, &cs);
      z[2 * p] = r * cs;
      z[2 * p + 1] = r * sn;
    }
    *(f32x4*)(out + 4 * i) = f + z * e * level;
  }
}

// ---- rotary position embedding on the q and k thirds of the token-major qkv buffer, in place (rope.py:66-98 with
// _rotate_half: element i of a head pairs with element i + dh/2).  cos/sin: f32 [T, dh/2] (the reference's cached table
// is this half-width table twice along the last dim).  One work item = 4 pairs of one (row, q|k, head): 16-byte loads
// of both halves.  inverse: rotate by -angle, i.e. the backward of the forward rotation (a rotation's transpose).
template <int BF16>
__global__ __launch_bounds__(256) void rope_qk_kernel(void* __restrict__ qkv, const float* __restrict__ cs,
                                                      const float* __restrict__ sn, long rows, int T, int H, int dh,
                                                      long ld, float sign) {
  const int hv = dh >> 3;                 // vectors of 4 pairs per head
  const long per_row = 2L * H * hv;
  const long total = rows * per_row;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / per_row;
    const int rem = (int)(idx - row * per_row);
    const int which = rem / (H * hv), r2 = rem - which * (H * hv);
    const int head = r2 / hv, v = r2 - head * hv;
    const int t = (int)(row % T);
    const f32x4 c = *(const f32x4*)(cs + (long)t * (dh >> 1) + 4 * v);
    const f32x4 s = *(const f32x4*)(sn + (long)t * (dh >> 1) + 4 * v) * sign;
    const long off = row * ld + (long)which * H * dh + (long)head * dh + 4 * v;
    f32x4 x1, x2;
    if (BF16) {
      const bf16x4 a = *(const bf16x4*)((const short*)qkv + off);
      const bf16x4 b = *(const bf16x4*)((const short*)qkv + off + (dh >> 1));
      x1 = (f32x4){bf2f(a[0]), bf2f(a[1]), bf2f(a[2]), bf2f(a[3])};
      x2 = (f32x4){bf2f(b[0]), bf2f(b[1]), bf2f(b[2]), bf2f(b[3])};
    } else {
      x1 = *(const f32x4*)((const float*)qkv + off);
      x2 = *(const f32x4*)((const float*)qkv + off + (dh >> 1));
    }
    const f32x4 o1 = x1 * c - x2 * s;   // x * cos + rotate_half(x) * sin, first half: rotate_half = -x2
    const f32x4 o2 = x2 * c + x1 * s;   //                                 second half: rotate_half = +x1
    if (BF16) {
      *(u32x2*)((short*)qkv + off) = (u32x2){pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3])};
      *(u32x2*)((short*)qkv + off + (dh >> 1)) = (u32x2){pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3])};
    } else {
      *(f32x4*)((float*)qkv + off) = o1;
      *(f32x4*)((float*)qkv + off + (dh >> 1)) = o2;
    }
  }
}

}  // namespace vit

extern "C" {
using namespace vit;

int vit_unfold_cast(vit_handle h, const float* x, void* patches, int out_dtype, int B, int L, int P, int S, int N,
                    vit_stream stream) {
  (void)h;
  VIT_CHECK(x && patches, VIT_ERR_ARG, "vit_unfold_cast: null pointer");
  VIT_CHECK(B > 0 && L > 0 && P > 0 && S > 0 && N > 0 && (P % 4) == 0, VIT_ERR_ARG,
            "vit_unfold_cast: B=%d L=%d P=%d S=%d N=%d (P must be a multiple of 4)", B, L, P, S, N);
  VIT_CHECK((long)(N - 1) * S < L, VIT_ERR_ARG, "vit_unfold_cast: patch %d starts past the signal", N - 1);
  const long total = (long)B * N * (P / 4);
  if (out_dtype == VIT_BF16)
    hipLaunchKernelGGL(unfold_cast_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, patches, B, L, P, S, N);
  else
    hipLaunchKernelGGL(unfold_cast_kernel<0>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, patches, B, L, P, S, N);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_fold_add(vit_handle h, const float* dpatches, float* dx, int B, int L, int P, int S, int N, vit_stream stream) {
  (void)h;
  VIT_CHECK(dpatches && dx, VIT_ERR_ARG, "vit_fold_add: null pointer");
  VIT_CHECK(B > 0 && L > 0 && P > 0 && S > 0 && N > 0 && P <= L, VIT_ERR_ARG, "vit_fold_add: B=%d L=%d P=%d S=%d N=%d", B, L,
            P, S, N);
  hipLaunchKernelGGL(fold_add_kernel, dim3(grid_for((long)B * L)), dim3(256), 0, (hipStream_t)stream, dpatches, dx, B, L, P,
                     S, N);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_add_noise(vit_handle h, const float* flux, const float* error, float* out, long n, float noise_level,
                  uint64_t seed, vit_stream stream) {
  (void)h;
  VIT_CHECK(flux && error && out, VIT_ERR_ARG, "vit_add_noise: null pointer");
  VIT_CHECK(n > 0 && (n % 4) == 0, VIT_ERR_ARG, "vit_add_noise: n=%ld must be a positive multiple of 4", n);
  const DropCfg d = make_drop(0.f, seed, 0x6e6f697365ull /* "noise" */);
  hipLaunchKernelGGL(add_noise_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, flux, error, out, n / 4,
                     noise_level, d.k0, d.k1);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_rope_qk(vit_handle h, void* qkv, int dtype, const float* cos_half, const float* sin_half, long rows, int T,
                int H, int dh, long ld, int inverse, vit_stream stream) {
  (void)h;
  VIT_CHECK(qkv && cos_half && sin_half, VIT_ERR_ARG, "vit_rope_qk: null pointer");
  VIT_CHECK(rows > 0 && T > 0 && H > 0 && dh > 0 && (dh % 8) == 0 && ld >= 3L * H * dh && (ld % 4) == 0, VIT_ERR_ARG,
            "vit_rope_qk: rows=%ld T=%d H=%d dh=%d ld=%ld (head_dim must be a multiple of 8)", rows, T, H, dh, ld);
  VIT_CHECK(dtype == VIT_BF16 || dtype == VIT_F32, VIT_ERR_ARG, "vit_rope_qk: bad dtype");
  const long total = rows * 2L * H * (dh >> 3);
  const float sign = inverse ? -1.f : 1.f;
  if (dtype == VIT_BF16)
    hipLaunchKernelGGL(rope_qk_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, qkv, cos_half, sin_half,
                       rows, T, H, dh, ld, sign);
  else
    hipLaunchKernelGGL(rope_qk_kernel<0>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, qkv, cos_half, sin_half,
                       rows, T, H, dh, ld, sign);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_embed_finish(vit_handle h, float* tokens, const float* cls, const float* pos, int B, int T, int D,
                     float dropout_p, uint64_t seed, uint64_t site, vit_stream stream) {
  (void)h;
  VIT_CHECK(tokens && cls, VIT_ERR_ARG, "vit_embed_finish: null pointer");
  VIT_CHECK(B > 0 && T > 1 && D > 0 && (D % 4) == 0, VIT_ERR_ARG, "vit_embed_finish: B=%d T=%d D=%d", B, T, D);
  VIT_CHECK(dropout_p >= 0.f && dropout_p < 1.f, VIT_ERR_ARG, "vit_embed_finish: dropout_p out of [0,1)");
  const long total = (long)B * T * (D / 4);
  hipLaunchKernelGGL(embed_finish_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tokens, cls, pos, B,
                     T, D, make_drop_h(h, dropout_p, seed, site));
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_embed_finish_bwd(vit_handle h, const float* dtokens, void* dpatch_out, int dpatch_dtype, float* dcls, float* dpos,
                         int B, int T, int D, float dropout_p, uint64_t seed, uint64_t site, int accumulate,
                         vit_stream stream) {
  VIT_CHECK(dtokens && dpatch_out && dcls, VIT_ERR_ARG, "vit_embed_finish_bwd: null pointer");
  VIT_CHECK(B > 0 && T > 1 && D > 0 && (D % 4) == 0, VIT_ERR_ARG, "vit_embed_finish_bwd: B=%d T=%d D=%d", B, T, D);
  const int nchunk = std::min(B, 16), bchunk = cdiv(B, nchunk), ny = cdiv(B, bchunk);
  size_t wsb = 0;
  float* part = (float*)ctx_workspace(h, &wsb);
  const size_t need = (size_t)ny * T * D * sizeof(float);
  VIT_CHECK(part && wsb >= need, VIT_ERR_WORKSPACE, "vit_embed_finish_bwd: needs %zu workspace bytes, have %zu", need, wsb);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(cdiv((long)T * (D / 4), 256), ny);
  if (dpatch_dtype == VIT_BF16)
    hipLaunchKernelGGL(embed_finish_bwd_kernel<1>, grid, dim3(256), 0, st, dtokens, dpatch_out, part, B, T, D,
                       make_drop_h(h, dropout_p, seed, site), bchunk);
  else
    hipLaunchKernelGGL(embed_finish_bwd_kernel<0>, grid, dim3(256), 0, st, dtokens, dpatch_out, part, B, T, D,
                       make_drop_h(h, dropout_p, seed, site), bchunk);
  VIT_LAUNCH_CHECK();
  // dcls = sum over the batch of row t = 0; dpos (if any) = the sums of every row
  int rc = launch_reduce_partials(part, ny, D, dcls, D, dcls, accumulate, st, T * D);
  if (rc != VIT_OK || !dpos) return rc;
  return launch_reduce_partials(part, ny, T * D, dpos, T * D, dpos, accumulate, st, T * D);
}

int vit_dropout_bwd_cast(vit_handle h, const float* dx, void* dy, int dy_dtype, int rows, int cols, float dropout_p,
                         uint64_t seed, uint64_t site, vit_stream stream) {
  (void)h;
  VIT_CHECK(dx && dy, VIT_ERR_ARG, "vit_dropout_bwd_cast: null pointer");
  VIT_CHECK(rows > 0 && cols > 0 && (cols % 4) == 0, VIT_ERR_ARG, "vit_dropout_bwd_cast: rows=%d cols=%d", rows, cols);
  const long total = (long)rows * (cols / 4);
  if (dy_dtype == VIT_BF16)
    hipLaunchKernelGGL(dropout_bwd_cast_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dx, dy,
                       (long)rows, cols, make_drop_h(h, dropout_p, seed, site));
  else
    hipLaunchKernelGGL(dropout_bwd_cast_kernel<0>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dx, dy,
                       (long)rows, cols, make_drop_h(h, dropout_p, seed, site));
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_colsum(vit_handle h, const void* a, int a_dtype, int64_t lda, float* out, int rows, int cols, int accumulate,
               vit_stream stream) {
  VIT_CHECK(a && out, VIT_ERR_ARG, "vit_colsum: null pointer");
  VIT_CHECK(rows > 0 && cols > 0 && (cols % 4) == 0 && (lda % 4) == 0 && lda >= cols, VIT_ERR_ARG,
            "vit_colsum: rows=%d cols=%d lda=%ld", rows, cols, (long)lda);
  const int gx = cdiv(cols, 256);
  const int gy = std::max(1, std::min(cdiv(rows, 64), 2048 / gx));
  size_t wsb = 0;
  float* part = (float*)ctx_workspace(h, &wsb);
  const size_t need = (size_t)gy * cols * sizeof(float);
  VIT_CHECK(part && wsb >= need, VIT_ERR_WORKSPACE, "vit_colsum: needs %zu workspace bytes, have %zu", need, wsb);
  hipStream_t st = (hipStream_t)stream;
  if (a_dtype == VIT_BF16)
    hipLaunchKernelGGL(colsum_stage1_kernel<1>, dim3(gx, gy), dim3(256), 0, st, a, (long)lda, part, rows, cols);
  else
    hipLaunchKernelGGL(colsum_stage1_kernel<0>, dim3(gx, gy), dim3(256), 0, st, a, (long)lda, part, rows, cols);
  VIT_LAUNCH_CHECK();
  return launch_reduce_partials(part, gy, cols, out, cols, out, accumulate, st);
}

int vit_cast_f32_bf16(vit_handle h, const float* src, void* dst, int64_t n, vit_stream stream) {
  (void)h;
  VIT_CHECK(src && dst && n > 0, VIT_ERR_ARG, "vit_cast_f32_bf16: bad arguments");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src, (short*)dst,
                     (long)n);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_cast_bf16_f32(vit_handle h, const void* src, float* dst, int64_t n, float scale, vit_stream stream) {
  (void)h;
  VIT_CHECK(src && dst && n > 0, VIT_ERR_ARG, "vit_cast_bf16_f32: bad arguments");
  VIT_CHECK((((uintptr_t)src) & 7) == 0 && (((uintptr_t)dst) & 15) == 0, VIT_ERR_ARG, "vit_cast_bf16_f32: src must be 8-byte, dst 16-byte aligned");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, (const short*)src, dst,
                     (long)n, scale);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_head_loss_fwd(vit_handle h, const float* last_hidden, const float* W, const float* b, const void* labels,
                      float* logits, float* loss_out, int B, int T, int D, int C, int loss_kind, vit_stream stream) {
  (void)h;
  VIT_CHECK(last_hidden && W && b && logits, VIT_ERR_ARG, "vit_head_loss_fwd: null pointer");
  VIT_CHECK(B > 0 && T > 0 && D > 0 && C > 0, VIT_ERR_ARG, "vit_head_loss_fwd: B=%d T=%d D=%d C=%d", B, T, D, C);
  VIT_CHECK(loss_kind >= VIT_LOSS_MSE && loss_kind <= VIT_LOSS_CE, VIT_ERR_ARG, "vit_head_loss_fwd: bad loss_kind");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(head_logits_kernel, dim3(B), dim3(64), 0, st, last_hidden, W, b, logits, T, D, C);
  VIT_LAUNCH_CHECK();
  if (labels) {
    VIT_CHECK(loss_out, VIT_ERR_ARG, "vit_head_loss_fwd: labels without loss_out");
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, st, logits, labels, loss_out, B, C, loss_kind);
    VIT_LAUNCH_CHECK();
  }
  return VIT_OK;
}

int vit_head_loss_bwd(vit_handle h, const float* last_hidden, const float* W, const float* logits, const void* labels,
                      const float* dloss, float* dlast_hidden, float* dW, float* db, int B, int T, int D, int C,
                      int loss_kind, int accumulate, vit_stream stream) {
  VIT_CHECK(last_hidden && W && logits && labels && dloss && dlast_hidden && dW && db, VIT_ERR_ARG,
            "vit_head_loss_bwd: null pointer");
  VIT_CHECK(B > 0 && T > 0 && D > 0 && C > 0, VIT_ERR_ARG, "vit_head_loss_bwd: B=%d T=%d D=%d C=%d", B, T, D, C);
  size_t wsb = 0;
  float* dlog = (float*)ctx_workspace(h, &wsb);
  VIT_CHECK(dlog && wsb >= (size_t)B * C * 4, VIT_ERR_WORKSPACE, "vit_head_loss_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  // a kernel, not hipMemsetAsync: inside a captured hipGraph the memset node did not reliably run before the kernels that
  // follow it (replays kept stale rows), and a fill kernel is ordered like every other node
  hipLaunchKernelGGL(zero_f32_kernel, dim3(grid_for((long)B * T * D / 4 + 1)), dim3(256), 0, st, dlast_hidden, (long)B * T * D);
  VIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_bwd_rows_kernel, dim3(B), dim3(64), 0, st, W, logits, labels, dloss, dlog, dlast_hidden, B, T,
                     D, C, loss_kind);
  VIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_bwd_params_kernel, dim3(cdiv((long)C * D, 256)), dim3(256), 0, st, last_hidden, dlog, dW, db,
                     B, T, D, C, accumulate);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

static int grad_sqnorm_impl(vit_handle h, const float* g, int64_t n, float* out, int accumulate, vit_stream stream) {
  VIT_CHECK(g && out && n > 0, VIT_ERR_ARG, "vit_grad_sqnorm: bad arguments");
  const int blocks = grid_for(n / 4 + 1, 256, 1024);
  size_t wsb = 0;
  float* part = (float*)ctx_workspace(h, &wsb);
  VIT_CHECK(part && wsb >= (size_t)blocks * 4, VIT_ERR_WORKSPACE, "vit_grad_sqnorm: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_stage1_kernel, dim3(blocks), dim3(256), 0, st, g, (long)n, part);
  VIT_LAUNCH_CHECK();
  hipLaunchKernelGGL(sqnorm_stage2_kernel, dim3(1), dim3(256), 0, st, part, blocks, out, accumulate);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}
int vit_grad_sqnorm(vit_handle h, const float* g, int64_t n, float* out, vit_stream stream) {
  return grad_sqnorm_impl(h, g, n, out, 0, stream);
}
int vit_grad_sqnorm_acc(vit_handle h, const float* g, int64_t n, float* out, vit_stream stream) {
  return grad_sqnorm_impl(h, g, n, out, 1, stream);
}

// ---- per-step state in device memory (hipGraph replays): one thread advances the step counter and derives from it the
// dropout keys of the step and AdamW's bias corrections; lr is whatever the host last wrote into the state
__global__ void step_advance_kernel(StepState* s, unsigned long long base_seed, float b1, float b2) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned step = s->step + 1;
    unsigned long long z = base_seed + 0x9E3779B97F4A7C15ull * (unsigned long long)step;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    s->key0 = (unsigned)z;
    s->key1 = (unsigned)(z >> 32);
    // in double, like the host computes them for the eager vit_adamw_step: graph and eager steps stay bit-identical
    s->bc1 = (float)(1.0 - pow((double)b1, (double)step));
    s->rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)step)));
    s->step = step;
  }
}
__global__ __launch_bounds__(256) void adamw_dyn_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        short* __restrict__ pb, long n, const StepState* __restrict__ st,
                                                        float b1, float b2, float eps, float wd,
                                                        const float* __restrict__ sqnorm, float max_norm) {
  float clip = 1.f;
  if (sqnorm) clip = fminf(1.f, max_norm / (sqrtf(sqnorm[0]) + 1e-6f));
  const float lr = st->lr, rsqrt_bc2 = st->rsqrt_bc2;
  const float step = lr / st->bc1, decay = 1.f - lr * wd;
  const long nv = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = *(const f32x4*)(p + 4 * i);
    const f32x4 gg = *(const f32x4*)(g + 4 * i) * clip;
    f32x4 mm = *(const f32x4*)(m + 4 * i);
    f32x4 vv = *(const f32x4*)(v + 4 * i);
    mm = mm * b1 + gg * (1.f - b1);
    vv = vv * b2 + gg * gg * (1.f - b2);
#pragma unroll
    for (int k = 0; k < 4; ++k) pp[k] = pp[k] * decay - step * mm[k] / (sqrtf(vv[k]) * rsqrt_bc2 + eps);
    *(f32x4*)(p + 4 * i) = pp;
    *(f32x4*)(m + 4 * i) = mm;
    *(f32x4*)(v + 4 * i) = vv;
    if (pb) {
      u32x2 pk = {pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3])};
      *(u32x2*)(pb + 4 * i) = pk;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (nv << 2) + threadIdx.x;
    const float gg = g[i] * clip;
    const float mm = m[i] * b1 + gg * (1.f - b1);
    const float vv = v[i] * b2 + gg * gg * (1.f - b2);
    const float pp = p[i] * decay - step * mm / (sqrtf(vv) * rsqrt_bc2 + eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (pb) pb[i] = f2bf(pp);
  }
}

int vit_step_advance(vit_handle h, uint64_t base_seed, float beta1, float beta2, vit_stream stream) {
  const StepState* st = ctx_step_state(h);
  VIT_CHECK(st, VIT_ERR_ARG, "vit_step_advance: no step state bound (vit_step_state_bind)");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, const_cast<StepState*>(st),
                     (unsigned long long)base_seed, beta1, beta2);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_adamw_step_dyn(vit_handle h, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float beta1,
                       float beta2, float eps, float weight_decay, const float* sqnorm, float max_norm, vit_stream stream) {
  const StepState* st = ctx_step_state(h);
  VIT_CHECK(st, VIT_ERR_ARG, "vit_adamw_step_dyn: no step state bound (vit_step_state_bind)");
  VIT_CHECK(p && g && m && v && n > 0, VIT_ERR_ARG, "vit_adamw_step_dyn: bad arguments");
  hipLaunchKernelGGL(adamw_dyn_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (short*)p_bf16, (long)n, st, beta1, beta2, eps, weight_decay, sqnorm, max_norm);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_adamw_step(vit_handle h, float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, const float* sqnorm,
                   float max_norm, vit_stream stream) {
  (void)h;
  VIT_CHECK(p && g && m && v && n > 0 && step >= 1, VIT_ERR_ARG, "vit_adamw_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     (short*)p_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                     (float)(1.0 / sqrt(bc2)), sqnorm, max_norm);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

}  // extern "C"
