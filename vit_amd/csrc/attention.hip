// Fused softmax attention (forward + backward) for gfx950, flash-style: scores / probabilities never reach HBM.
//
// Reads Q, K, V straight out of the fused QKV projection's output [B*T, 3*H*dh] (token-major, head h at columns
// h*dh of each third) and writes the merged-head context [B*T, H*dh], so no head split / merge copies exist.
//
// One wave owns 16 query rows (forward, dQ) or 16 keys (dK/dV); 4 waves share the 64-row K/V (or Q/dO) tiles staged
// in LDS.  All MFMAs (v_mfma_f32_16x16x32_bf16) are issued "swapped" so the 16 owned rows sit on lane&15:
//     S^T = K * Q^T,   O^T = V^T * P^T,   dP^T = V * dO^T,   dQ^T = K^T * dS^T          (owner = query)
//     S   = Q * K^T,   dP  = dO * V^T,    dV^T = dO^T * P,   dK^T = Q^T * dS            (owner = key)
// With that orientation (a) the softmax statistics m, l, lse, delta of a row live in the lane that owns the row
// (only the 4 lane groups lane>>4 have to be combined: two xor-shuffles), (b) the P / dS accumulator tile is already
// the B operand of the next MFMA (4 consecutive reduction indices per 16-tile per lane: k-slot j<4 -> tile 2u,
// j>=4 -> tile 2u+1), and (c) the other operand of that product is a transposed read of the row-major LDS tile
// (ds_read_b64_tr_b16) with the same slot order.  No P round trip through LDS, no permutes.
//
// LDS tile image: [64 rows][DH] bf16, 16-byte chunk c of row r at r*2*DH + ((c ^ swz(r)) << 4); row fragments by
// ds_read_b128, transposed fragments by ds_read_b64_tr_b16 inside the chunks.
#include <algorithm>
#include <math.h>

#include <type_traits>
#include "common.h"

namespace vit {

void* ctx_workspace(vit_handle h, size_t* bytes);

constexpr int AW = 4;    // waves per workgroup
constexpr int RT = 64;   // rows per LDS tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnArgs {
  const short* qkv; short* ctx; float* lse;
  short* ctx_lo;  // optional bf16 [B*T, H*dh]: the rounding residual ctx_exact - bf16(ctx_exact), so that the backward's
                  // delta = rowsum(dO * O) sees O to ~16 mantissa bits (with the 8-bit O its error is common to a whole
                  // score row and survives the sum over keys in dQ / dK: measured 5e-2 on ViT-L's deep query weights)
  const short* dctx; float* delta; short* dqkv;
  int B, H, T, dh;
  float scale;
  DropCfg drop;
  int nsplit, wpw;  // resident kernels: workgroups per (batch, head) and waves per workgroup (row tiles are dealt in order)
  float* csum_part;  // resident backward kernels: [B * nsplit * wpw][3 * H * dh] per-wave column sums of dqkv as stored, or NULL
};

// XOR applied to the 16-byte chunk index of row r.  Two kinds of read share an image and both must be free of bank conflicts
// (r03: the earlier swizzles served the row reads only; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE read 0.27 / 0.43 in the
// forward / backward kernels while the GEMM images read 0.00):
//  * row fragments, ds_read_b128: a 16-lane group = 8 rows of lane group lg at chunk c0 and 8 rows of lg ^ 1 at chunk c0 ^ 1
//    (rows r and r ^ 8 never share a chunk) -> the 16 (row, chunk) slots must tile the 256-byte bank row;
//  * transposed fragments, ds_read_b64_tr_b16: a 32-lane half = 8 consecutive rows x one 32-byte chunk PAIR {2dt, 2dt + 1}
//    -> the 8 rows must land on 8 different 32-byte regions of the bank row, so the pair index (chunk >> 1) has to be
//    XORed with something that differs between rows that share their position (r mod rows-per-bank-row).
// DH = 64 (128-byte rows, 2 per bank row): (r & 6) -- pair index ^ ((r >> 1) & 3), the row's parity picks the half.
// DH = 32 (64-byte rows, 4 per bank row): rows r and r + 4 share a quarter -> pair index ^ ((r >> 2) & 1).
// DH = 128 (256-byte rows): pair index ^ (r & 7).
template <int DH>
__device__ __forceinline__ int swz(int r) {
  return DH == 32 ? ((r >> 1) & 2) : (DH == 64 ? (r & 6) : ((r & 7) << 1));
}
template <int DH>
__device__ __forceinline__ int tile_off(int r, int c) {
  return r * (DH * 2) + ((c ^ swz<DH>(r)) << 4);
}

// 8 bf16 of a head's row from column `col` (a multiple of 8), zero past the head size.  A head size that is a multiple of 4
// but not of 8 (the reference's sweep reaches hidden 32 / 8 heads = 4, configs/sweep.yaml:13-18) puts heads at 8-byte offsets
// and ends them in half a chunk: those take 8-byte loads; every other shape keeps its single 16-byte load.
__device__ __forceinline__ i32x4 ld_head8(const short* p, int col, int dh) {
  i32x4 v = {0, 0, 0, 0};
  if ((dh & 7) == 0) {
    if (col < dh) v = *(const i32x4*)(p + col);
    return v;
  }
  if (col < dh) {
    const i32x2 h = *(const i32x2*)(p + col);
    v[0] = h[0]; v[1] = h[1];
  }
  if (col + 4 < dh) {
    const i32x2 h = *(const i32x2*)(p + col + 4);
    v[2] = h[0]; v[3] = h[1];
  }
  return v;
}

// cooperative load of rows [row0, row0+64) x [0, DH) of a strided bf16 matrix into an LDS image (zero fill outside)
template <int DH>
__device__ __forceinline__ void load_tile(char* img, const short* g, long ld, int row0, int nrows, int dh, int tid) {
  constexpr int CPR = DH / 8;
#pragma unroll
  for (int i = 0; i < (RT * CPR) / (AW * 64); ++i) {
    const int q = tid + AW * 64 * i;
    const int r = q / CPR, c = q % CPR;
    const int row = row0 + r;
    i32x4 v = {0, 0, 0, 0};
    if (row < nrows) v = ld_head8(g + (long)row * ld, c * 8, dh);
    *(i32x4*)(img + tile_off<DH>(r, c)) = v;
  }
}

// 16 rows starting at rb (tile-local), reduction index = columns s*32 + 8*(lane>>4) + j
template <int DH>
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int rb, int s, int l15, int lg) {
  return *(const bf16x8*)(img + tile_off<DH>(rb + l15, s * 4 + lg));
}
// transposed: MFMA rows = columns cb..cb+15 of the tile, reduction slots j<4 -> row rb0 + 4*lg + j, j>=4 -> rb1 + ...
template <int DH>
__device__ __forceinline__ bf16x8 frag_cols(const char* img, int rb0, int rb1, int cb, int l15, int lg) {
  const int tq = l15 >> 2, tp = l15 & 3;
  const int col = cb + 4 * tp;
  const int c = col >> 3, half = (col >> 2) & 1;
  const int r0 = rb0 + 4 * lg + tq, r1 = rb1 + 4 * lg + tq;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(img + tile_off<DH>(r0, c) + half * 8));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(img + tile_off<DH>(r1, c) + half * 8));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// the 16 rows a wave owns, straight from global memory into fragment registers (row = r0 + lane&15)
template <int DH>
__device__ __forceinline__ void load_own(bf16x8 (&f)[DH / 32], const short* g, long ld, int r0, int nrows, int dh,
                                         int l15, int lg) {
#pragma unroll
  for (int s = 0; s < DH / 32; ++s) {
    const int col = s * 32 + lg * 8;
    i32x4 v = {0, 0, 0, 0};
    if (r0 + l15 < nrows) v = ld_head8(g + (long)(r0 + l15) * ld, col, dh);
    f[s] = __builtin_bit_cast(bf16x8, v);
  }
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  u32x4 r = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
// lo = bf16(v - bf16(v)) for 4 values already packed as pk
__device__ __forceinline__ void store_lo(short* dst, const f32x4& v, const u32x2& pk) {
  const float h0 = __builtin_bit_cast(float, pk[0] << 16), h1 = __builtin_bit_cast(float, pk[0] & 0xFFFF0000u);
  const float h2 = __builtin_bit_cast(float, pk[1] << 16), h3 = __builtin_bit_cast(float, pk[1] & 0xFFFF0000u);
  u32x2 lo = {pack2bf(v[0] - h0, v[1] - h1), pack2bf(v[2] - h2, v[3] - h3)};
  *(u32x2*)dst = lo;
}

// ------------------------------------------------------------------------------------------------ forward
template <int DH>
__global__ __launch_bounds__(AW * 64) void attn_fwd_kernel(AttnArgs p) {
  resolve_drop(p.drop);
  __shared__ __attribute__((aligned(16))) char smem[2 * RT * DH * 2];
  char* Kimg = smem;
  char* Vimg = smem + RT * DH * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int T = p.T, dh = p.dh;
  const long ld = 3L * p.H * dh;
  const short* qb = p.qkv + (long)b * T * ld + h * dh;
  const short* kb_ = qb + p.H * dh;
  const short* vb = kb_ + p.H * dh;
  const int q0 = (blockIdx.x * AW + wave) * 16;

  bf16x8 qf[DH / 32];
  load_own<DH>(qf, qb, ld, q0, T, dh, l15, lg);
  const float c = p.scale * LOG2E;
  float m = -INFINITY, l = 0.f;
  f32x4 ot[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) ot[i] = zero4();
  const unsigned long long drow = (unsigned long long)bh * T + (q0 + l15);
  const unsigned half_cols = (unsigned)((T + 1) >> 1);

  for (int kb = 0; kb < T; kb += RT) {
    if (kb) __syncthreads();
    load_tile<DH>(Kimg, kb_, ld, kb, T, dh, tid);
    load_tile<DH>(Vimg, vb, ld, kb, T, dh, tid);
    __syncthreads();
    f32x4 st[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kb + j * 16 < T) {
        f32x4 a = zero4();
#pragma unroll
        for (int s = 0; s < DH / 32; ++s)
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Kimg, j * 16, s, l15, lg), qf[s], a, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb + j * 16 + lg * 4 + r;
          a[r] = key < T ? a[r] * c : -INFINITY;
          mx = fmaxf(mx, a[r]);
        }
        st[j] = a;
      } else {
        st[j] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = exp2f(m - mn);
    m = mn;
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[j][r] = exp2f(st[j][r] - mn);
        ls += st[j][r];
      }
    l = l * alpha + ls;
#pragma unroll
    for (int i = 0; i < DH / 16; ++i) ot[i] *= alpha;
    if (p.drop.thr) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned key = kb + j * 16 + lg * 4;
        float k0, k1, k2, k3;
        drop_pair(p.drop, drow, half_cols, key, k0, k1);
        drop_pair(p.drop, drow, half_cols, key + 2, k2, k3);
        st[j][0] *= k0; st[j][1] *= k1; st[j][2] *= k2; st[j][3] *= k3;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (kb + u * 32 < T) {
        const bf16x8 pf = pack8(st[2 * u], st[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt)
          ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Vimg, u * 32, u * 32 + 16, dt * 16, l15, lg),
                                                           pf, ot[dt], 0, 0, 0);
      }
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const int q = q0 + l15;
  if (q < T) {
    const float inv = 1.0f / l;
    short* o = p.ctx + ((long)b * T + q) * (p.H * dh) + h * dh;
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      const int d = dt * 16 + lg * 4;
      if (d < dh) {
        const f32x4 v = ot[dt] * inv;
        u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(u32x2*)(o + d) = pk;
        if (p.ctx_lo) store_lo(p.ctx_lo + (o - p.ctx) + d, v, pk);
      }
    }
    if (lg == 0) p.lse[(long)bh * T + q] = (m + log2f(l)) * LN2;
  }
}

// ------------------------------------------------------------------------------------------------ dQ
template <int DH>
__global__ __launch_bounds__(AW * 64) void attn_bwd_dq_kernel(AttnArgs p) {
  resolve_drop(p.drop);
  __shared__ __attribute__((aligned(16))) char smem[2 * RT * DH * 2];
  char* Kimg = smem;
  char* Vimg = smem + RT * DH * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int T = p.T, dh = p.dh;
  const long ld = 3L * p.H * dh, ldc = (long)p.H * dh;
  const short* qb = p.qkv + (long)b * T * ld + h * dh;
  const short* kb_ = qb + p.H * dh;
  const short* vb = kb_ + p.H * dh;
  const short* dob = p.dctx + (long)b * T * ldc + h * dh;
  const int q0 = (blockIdx.x * AW + wave) * 16;
  const int q = q0 + l15;

  bf16x8 qf[DH / 32], dof[DH / 32];
  load_own<DH>(qf, qb, ld, q0, T, dh, l15, lg);
  load_own<DH>(dof, dob, ldc, q0, T, dh, l15, lg);
  const float c = p.scale * LOG2E;
  const float lse2 = q < T ? p.lse[(long)bh * T + q] * LOG2E : INFINITY;
  // delta[q] = rowsum(dO * O): this lane holds 8 columns per 32-column step of its row; the 4 lane groups complete the
  // row with two xor-shuffles.  Written out for the dK/dV kernel that runs next on the stream.
  float del = 0.f;
  {
    const short* ob = p.ctx + (long)b * T * ldc + h * dh;
#pragma unroll
    for (int s = 0; s < DH / 32; ++s) {
      const int col = s * 32 + lg * 8;
      if (q < T && col < dh) {
        const bf16x8 o = __builtin_bit_cast(bf16x8, ld_head8(ob + (long)q * ldc, col, dh));
#pragma unroll
        for (int e = 0; e < 8; ++e) del += bf2f(o[e]) * bf2f(dof[s][e]);
        if (p.ctx_lo) {
          const bf16x8 ol = __builtin_bit_cast(bf16x8, ld_head8(p.ctx_lo + (ob - p.ctx) + (long)q * ldc, col, dh));
#pragma unroll
          for (int e = 0; e < 8; ++e) del += bf2f(ol[e]) * bf2f(dof[s][e]);
        }
      }
    }
    del += __shfl_xor(del, 16, 64);
    del += __shfl_xor(del, 32, 64);
    if (q < T && lg == 0) p.delta[(long)bh * T + q] = del;
  }
  f32x4 dqt[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) dqt[i] = zero4();
  const unsigned long long drow = (unsigned long long)bh * T + q;
  const unsigned half_cols = (unsigned)((T + 1) >> 1);

  for (int kb = 0; kb < T; kb += RT) {
    if (kb) __syncthreads();
    load_tile<DH>(Kimg, kb_, ld, kb, T, dh, tid);
    load_tile<DH>(Vimg, vb, ld, kb, T, dh, tid);
    __syncthreads();
    f32x4 ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ds[j] = zero4();
      if (kb + j * 16 < T) {
        f32x4 s_ = zero4(), dp = zero4();
#pragma unroll
        for (int s = 0; s < DH / 32; ++s) {
          s_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Kimg, j * 16, s, l15, lg), qf[s], s_, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Vimg, j * 16, s, l15, lg), dof[s], dp, 0, 0, 0);
        }
        const unsigned key0 = kb + j * 16 + lg * 4;
        float k[4] = {1.f, 1.f, 1.f, 1.f};
        if (p.drop.thr) {
          drop_pair(p.drop, drow, half_cols, key0, k[0], k[1]);
          drop_pair(p.drop, drow, half_cols, key0 + 2, k[2], k[3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = ((int)key0 + r < T) ? exp2f(s_[r] * c - lse2) : 0.f;
          ds[j][r] = pr * (dp[r] * k[r] - del);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (kb + u * 32 < T) {
        const bf16x8 df = pack8(ds[2 * u], ds[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt)
          dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Kimg, u * 32, u * 32 + 16, dt * 16, l15, lg),
                                                            df, dqt[dt], 0, 0, 0);
      }
    }
  }
  if (q < T) {
    short* o = p.dqkv + ((long)b * T + q) * ld + h * dh;
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      const int d = dt * 16 + lg * 4;
      if (d < dh) {
        const f32x4 v = dqt[dt] * p.scale;
        u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(u32x2*)(o + d) = pk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int DH>
__global__ __launch_bounds__(AW * 64) void attn_bwd_dkv_kernel(AttnArgs p) {
  resolve_drop(p.drop);
  __shared__ __attribute__((aligned(16))) char smem[2 * RT * DH * 2 + 2 * RT * 4];
  char* Qimg = smem;
  char* Oimg = smem + RT * DH * 2;
  float* lse_s = (float*)(smem + 2 * RT * DH * 2);
  float* del_s = lse_s + RT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int T = p.T, dh = p.dh;
  const long ld = 3L * p.H * dh, ldc = (long)p.H * dh;
  const short* qb = p.qkv + (long)b * T * ld + h * dh;
  const short* kb_ = qb + p.H * dh;
  const short* vb = kb_ + p.H * dh;
  const short* dob = p.dctx + (long)b * T * ldc + h * dh;
  const int k0w = (blockIdx.x * AW + wave) * 16;
  const int key = k0w + l15;

  bf16x8 kf[DH / 32], vf[DH / 32];
  load_own<DH>(kf, kb_, ld, k0w, T, dh, l15, lg);
  load_own<DH>(vf, vb, ld, k0w, T, dh, l15, lg);
  const float c = p.scale * LOG2E;
  f32x4 dkt[DH / 16], dvt[DH / 16];
#pragma unroll
  for (int i = 0; i < DH / 16; ++i) dkt[i] = dvt[i] = zero4();
  const unsigned half_cols = (unsigned)((T + 1) >> 1);

  for (int qb0 = 0; qb0 < T; qb0 += RT) {
    if (qb0) __syncthreads();
    load_tile<DH>(Qimg, qb, ld, qb0, T, dh, tid);
    load_tile<DH>(Oimg, dob, ldc, qb0, T, dh, tid);
    if (tid < RT) {
      const int qq = qb0 + tid;
      lse_s[tid] = qq < T ? p.lse[(long)bh * T + qq] * LOG2E : INFINITY;
      del_s[tid] = qq < T ? p.delta[(long)bh * T + qq] : 0.f;
    }
    __syncthreads();
    f32x4 pd[4], ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pd[j] = ds[j] = zero4();
      if (qb0 + j * 16 < T) {
        f32x4 s_ = zero4(), dp = zero4();
#pragma unroll
        for (int s = 0; s < DH / 32; ++s) {
          s_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Qimg, j * 16, s, l15, lg), kf[s], s_, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows<DH>(Oimg, j * 16, s, l15, lg), vf[s], dp, 0, 0, 0);
        }
        const f32x4 l4 = *(const f32x4*)(lse_s + j * 16 + lg * 4);
        const f32x4 d4 = *(const f32x4*)(del_s + j * 16 + lg * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = exp2f(s_[r] * c - l4[r]);  // rows past T carry lse = +inf -> 0
          float mk = 1.f;
          if (p.drop.thr) {
            const unsigned long long row = (unsigned long long)bh * T + (qb0 + j * 16 + lg * 4 + r);
            const unsigned hsh = drop_bits(drop_rowkey(p.drop, row), (unsigned)key >> 1);
            const unsigned r16 = (key & 1) ? (hsh >> 16) : (hsh & 0xFFFFu);
            mk = r16 >= p.drop.thr ? p.drop.scale : 0.f;
          }
          pd[j][r] = pr * mk;
          ds[j][r] = pr * (dp[r] * mk - d4[r]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (qb0 + u * 32 < T) {
        const bf16x8 pf = pack8(pd[2 * u], pd[2 * u + 1]);
        const bf16x8 df = pack8(ds[2 * u], ds[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
          dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Oimg, u * 32, u * 32 + 16, dt * 16, l15, lg),
                                                            pf, dvt[dt], 0, 0, 0);
          dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols<DH>(Qimg, u * 32, u * 32 + 16, dt * 16, l15, lg),
                                                            df, dkt[dt], 0, 0, 0);
        }
      }
    }
  }
  if (key < T) {
    short* ok = p.dqkv + ((long)b * T + key) * ld + p.H * dh + h * dh;
    short* ov = ok + p.H * dh;
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      const int d = dt * 16 + lg * 4;
      if (d < dh) {
        const f32x4 a = dkt[dt] * p.scale, v = dvt[dt];
        u32x2 pk = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
        u32x2 pv = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(u32x2*)(ok + d) = pk;
        *(u32x2*)(ov + d) = pv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ probabilities
// eval-time attention maps [B,H,T,T] f32 (output_attentions=True, consumed by the reference's viz callbacks);
// a plain VALU kernel off the training path: one wave per (b, h, q) row.
__global__ __launch_bounds__(256) void attn_probs_kernel(const short* __restrict__ qkv, float* __restrict__ probs, int B,
                                                         int H, int T, int dh, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // (b*H + h)*T + q
  if (row >= (long)B * H * T) return;
  const int q = (int)(row % T);
  const long bh = row / T;
  const int h = (int)(bh % H);
  const long b = bh / H;
  const long ld = 3L * H * dh;
  const short* qp = qkv + (b * T + q) * ld + h * dh;
  const short* kp = qkv + b * T * ld + H * dh + h * dh;
  float* out = probs + row * T;
  float mx = -INFINITY;
  for (int k = lane; k < T; k += 64) {
    float s = 0.f;
    for (int d = 0; d < dh; d += 8) {
      const bf16x8 x = __builtin_bit_cast(bf16x8, ld_head8(qp, d, dh)), y = __builtin_bit_cast(bf16x8, ld_head8(kp + (long)k * ld, d, dh));
#pragma unroll
      for (int e = 0; e < 8; ++e) s += bf2f(x[e]) * bf2f(y[e]);
    }
    s *= scale;
    out[k] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < T; k += 64) {
    const float e = __expf(out[k] - mx);
    out[k] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int k = lane; k < T; k += 64) out[k] *= inv;
}

// ======================================================================================= resident kernels (T <= 256)
// When a (batch, head)'s whole K and V (or Q and dO) fit the LDS -- T <= 256: every config of BASELINE.json's bench
// line -- one workgroup owns the (batch, head): the tiles are staged ONCE, there is a single barrier, and every wave
// then runs its key loop without any synchronisation.  Each wave owns RQ = 2 sixteen-row tiles and feeds both from
// every K / V fragment it reads, which halves the LDS read traffic per MFMA (the 4-wave tiled kernels above were
// LDS-read and barrier bound at ~135 TFLOP/s).
// Stage two [T, dh] matrices (all their 64-row tiles) at once: every global load of a thread is issued before its first
// LDS store, so a workgroup pays ONE memory latency for its whole working set (a load->store loop paid ten).
template <int DH>
__device__ __forceinline__ void load_all_tiles2(char* imgA, const short* ga, long lda, char* imgB, const short* gb,
                                                long ldb, int T, int dh, int rows_alloc, int tid, int nthr) {
  constexpr int CPR = DH / 8, MAXI = 8;
  const int total = rows_alloc * CPR;  // rows staged (zero beyond T): a multiple of 16, not necessarily of the 64-row tile
  for (int base = 0; base < total; base += MAXI * nthr) {
    i32x4 va[MAXI], vb[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int q = base + tid + i * nthr;
      const int r = q / CPR, c = q % CPR;
      va[i] = vb[i] = (i32x4){0, 0, 0, 0};
      if (q < total && r < T && c * 8 < dh) {
        va[i] = *(const i32x4*)(ga + (long)r * lda + c * 8);
        vb[i] = *(const i32x4*)(gb + (long)r * ldb + c * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int q = base + tid + i * nthr;
      if (q < total) {
        const int r = q / CPR, c = q % CPR;
        const int off = (r >> 6) * (RT * DH * 2) + tile_off<DH>(r & 63, c);
        *(i32x4*)(imgA + off) = va[i];
        *(i32x4*)(imgB + off) = vb[i];
      }
    }
  }
}

// LDS-DMA staging of a [rows x 128 B] tile image (head_dim 64): 1 KiB of consecutive LDS per wave-instruction, the XOR
// swizzle applied to the GLOBAL address of each lane; rows past T are clamped to row T - 1 (a DMA has no bounds check).
#define GLB_AS __attribute__((address_space(1)))

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {  // n is wave-uniform; a smaller count than asked for is always safe
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
    case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
    case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
    case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
    case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
  }
}
__device__ __forceinline__ void dma_rows64(char* img, const short* g, long ld, int row0, int nrows_img, int T, int wave,
                                           int lane, int nwaves = 8) {
  // image = nrows_img rows of 128 B, tile layout (row r at r * 128, chunk c at ((c ^ swz(r)) << 4))
  const int p = lane & 7;
  const unsigned img_a = lds_addr_of(img);
  for (int j = wave; j < (nrows_img >> 3); j += nwaves) {
    const int r = (j << 3) + (lane >> 3);
    const int c = p ^ swz<64>(r & 63);
    const int grow = min(row0 + r, T - 1);
    // uniform base + 32-bit lane offset (rows x row stride x 2 B stays far below 2^32 inside one head's rows), raw LDS address
    lds_dma16_s(g, __umul24((unsigned)grow, (unsigned)(ld * 2)) + (unsigned)(c * 16), img_a + j * 1024);
  }
}

// lo = bf16(v - bf16(v)) for 4 values already packed as pk (the context residual), returned packed
__device__ __forceinline__ u32x2 pack_lo(const f32x4& v, const u32x2& pk) {
  const float h0 = __builtin_bit_cast(float, pk[0] << 16), h1 = __builtin_bit_cast(float, pk[0] & 0xFFFF0000u);
  const float h2 = __builtin_bit_cast(float, pk[1] << 16), h3 = __builtin_bit_cast(float, pk[1] & 0xFFFF0000u);
  return (u32x2){pack2bf(v[0] - h0, v[1] - h1), pack2bf(v[2] - h2, v[3] - h3)};
}

#ifdef VIT_FWD_STAMP  // diagnostic build only (tools/fwd_stamps.py): where a wave's lifetime goes, one record per wave
#define FWD_ST_WAVES (1 << 16)
__device__ unsigned long long g_fwd_st[FWD_ST_WAVES * 8];
#define FWD_ST(K)                                                                                   \
  {                                                                                                 \
    unsigned long long t_;                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    st_[K] = t_ - tprev_;                                                                           \
    tprev_ = t_;                                                                                    \
  }
#else
#define FWD_ST(K)
#endif

// The key loop of one wave: RQ 16-row query tiles (fragments qf) against the staged K / V images of a head; running max m,
// row sums l and the transposed output accumulators ot are the caller's.  pre_pv() runs once, before the first V fragment read.
template <int DH, int RQ, int TPC, class PrePV>  // TPC: the padded length 64 n + 16 when TPC - 16 < T <= TPC is known at compile time (208: ViT-B, 592: ViT-L), else 0
__device__ __forceinline__ void fwd_keyloop(const char* Kimg, const char* Vimg, const bf16x8 (&qf)[RQ][DH / 32], float (&m)[RQ],
                                            float (&l)[RQ], f32x4 (&ot)[RQ][DH / 16], int T, float c, const DropCfg& drop, int bh,
                                            int q00, int l15, int lg, PrePV&& pre_pv) {
  constexpr int TILE = RT * DH * 2;
  // One 64-key tile.  NJ = its 16-key blocks that hold keys (compile-time: the full tiles run a body with no validity test,
  // no -inf fills and no edge select at all; the LAST tile runs the body for its own block count, so a T = 197 head does
  // 3 x 4 + 1 blocks of softmax / dropout work instead of 4 x 4), EDGE = the last block straddles T (per-key select).
  // Blocks that are left out would have contributed exp(-inf) = 0 to the row sums and zero rows to P V: same results.
  // HOOK (compile-time): the caller's pre_pv() runs between this tile's scores and its first V fragment read.
  auto tile = [&](auto njc, auto edgec, auto hookc, int kt) {
    constexpr int NJ = decltype(njc)::value;
    constexpr bool EDGE = decltype(edgec)::value;
    const int kb = kt * RT;
    const char* Kt = Kimg + kt * TILE;
    const char* Vt = Vimg + kt * TILE;
    f32x4 st[RQ][NJ];
    float mx[RQ];
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) mx[rq] = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 a[RQ];
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) a[rq] = zero4();
#pragma unroll
      for (int s = 0; s < DH / 32; ++s) {
        const bf16x8 kf = frag_rows<DH>(Kt, j * 16, s, l15, lg);
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) a[rq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[rq][s], a[rq], 0, 0, 0);
      }
      // raw scores: the scale rides in the exp2's FMA below (max(c s) = c max(s), c > 0)
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (EDGE && j == NJ - 1 && kb + j * 16 + lg * 4 + r >= T) a[rq][r] = -INFINITY;
          mx[rq] = fmaxf(mx[rq], a[rq][r]);
        }
        st[rq][j] = a[rq];
      }
    }
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
      const float mn = fmaxf(m[rq], grp4_max(mx[rq]));  // running max of the RAW scores
      const float alpha = fast_exp2((m[rq] - mn) * c);
      m[rq] = mn;
      const float mnc = mn * c;
      float ls = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[rq][j][r] = fast_exp2(fmaf(st[rq][j][r], c, -mnc));
          ls += st[rq][j][r];
        }
      l[rq] = l[rq] * alpha + ls;
#pragma unroll
      for (int i = 0; i < DH / 16; ++i) ot[rq][i] *= alpha;
      if (drop.thr) {
        // keep <=> the element's 16-bit draw >= thr: the high draw by ONE unsigned compare of the whole word against thr << 16,
        // the low draw after one shift; dropped probabilities become 0 by a select, and the 1 / (1 - p) of the kept ones is
        // applied once per row at the end (it rides in `inv`): 2.5 instead of 4 VALU per element in this VALU-bound kernel
        const unsigned rkey = drop_rowkey(drop, (unsigned long long)bh * T + (q00 + rq * 16 + l15));
        const unsigned thr16 = drop.thr << 16;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const unsigned key = kb + j * 16 + lg * 4;
          const unsigned ha = drop_bits(rkey, key >> 1), hb = drop_bits(rkey, (key >> 1) + 1);
          st[rq][j][0] = (ha << 16) >= thr16 ? st[rq][j][0] : 0.f;
          st[rq][j][1] = ha >= thr16 ? st[rq][j][1] : 0.f;
          st[rq][j][2] = (hb << 16) >= thr16 ? st[rq][j][2] : 0.f;
          st[rq][j][3] = hb >= thr16 ? st[rq][j][3] : 0.f;
        }
      }
    }
    if constexpr (decltype(hookc)::value) pre_pv();
#pragma unroll
    for (int u = 0; u < (NJ + 1) / 2; ++u) {
      const bool two = 2 * u + 1 < NJ;  // compile-time after unrolling
      bf16x8 pf[RQ];
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) pf[rq] = two ? pack8(st[rq][2 * u], st[rq][2 * u + 1 < NJ ? 2 * u + 1 : 2 * u]) : pack8(st[rq][2 * u], zero4());
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        // a 16-row block with no key in it is not staged: point its half of the fragment at the first block (its P is 0)
        const bf16x8 vf = frag_cols<DH>(Vt, u * 32, two ? u * 32 + 16 : u * 32, dt * 16, l15, lg);
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq)
          ot[rq][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[rq], ot[rq][dt], 0, 0, 0);
      }
    }
  };
  using std::integral_constant;
  using no = integral_constant<bool, false>;
  using yes = integral_constant<bool, true>;
  if constexpr (TPC != 0) {  // TPC = 64 n + 16: n full tiles and one 16-key block
    static_assert(TPC % 64 == 16, "compile-time sequence lengths end in one 16-key block");
    constexpr int NFULL = TPC / 64;
    tile(integral_constant<int, 4>{}, no{}, yes{}, 0);
#pragma clang loop unroll(disable)  // unrolled, the scheduler overlaps the tiles and spills 50 registers per lane
    for (int kt = 1; kt < NFULL; ++kt) tile(integral_constant<int, 4>{}, no{}, no{}, kt);
    tile(integral_constant<int, 1>{}, yes{}, no{}, NFULL);
    return;
  }
  const int nfull = T / RT;
  // the first tile is peeled (the hook sits inside it); a sequence shorter than one full tile runs the hook before its only tile
  if (nfull > 0) tile(integral_constant<int, 4>{}, no{}, yes{}, 0);
  else pre_pv();
  for (int kt = 1; kt < nfull; ++kt) tile(integral_constant<int, 4>{}, no{}, no{}, kt);
  const int rem = T - nfull * RT;
  if (rem > 0) {
    const int nj = (rem + 15) >> 4;
    if (nj == 1) tile(integral_constant<int, 1>{}, yes{}, no{}, nfull);
    else if (nj == 2) tile(integral_constant<int, 2>{}, yes{}, no{}, nfull);
    else if (nj == 3) tile(integral_constant<int, 3>{}, yes{}, no{}, nfull);
    else tile(integral_constant<int, 4>{}, yes{}, no{}, nfull);
  }
}

// Normalise, store the context rows (+ their rounding residual) and the row statistics of one wave's query tiles.
template <int DH, int RQ, bool FULL = false, int HC = 0>  // FULL: dh == DH is known at compile time (no row-per-lane store path compiled); HC: head count, if known
__device__ __forceinline__ void fwd_finish(const AttnArgs& p, const float (&m)[RQ], const float (&l)[RQ], f32x4 (&ot)[RQ][DH / 16], int b,
                                           int h, int bh, int q00, float c, int l15, int lg) {
  const int T = p.T, dh = FULL ? DH : p.dh, NH = HC ? HC : p.H;
#pragma unroll
  for (int rq = 0; rq < RQ; ++rq) {
    if (q00 + rq * 16 >= T) continue;  // uniform: a tile with no row below T has nothing to store
    const float lt = grp4_sum(l[rq]);
    const int q = q00 + rq * 16 + l15;
    const float inv = (p.drop.thr ? p.drop.scale : 1.0f) / lt;  // the kept probabilities' 1 / (1 - p) rides here
    short* o = p.ctx + ((long)b * T + q) * (NH * dh) + h * dh;
    if ((FULL || dh == DH) && (DH % 32) == 0) {
      // 16-byte stores: two adjacent 16-column tiles per instruction (row-per-lane stores are issue-bound)
#pragma unroll
      for (int dp = 0; dp < DH / 32; ++dp) {
        const f32x4 v0 = ot[rq][dp * 2] * inv, v1 = ot[rq][dp * 2 + 1] * inv;
        u32x2 p0 = {pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3])}, p1 = {pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3])};
        u32x2 l0 = pack_lo(v0, p0), l1 = pack_lo(v1, p1);
        const int col = widen_pair(p0, p1, lg);
        widen_pair(l0, l1, lg);
        if (q < T) {
          *(u32x4*)(o + dp * 32 + col) = (u32x4){p0[0], p0[1], p1[0], p1[1]};
          if (p.ctx_lo) *(u32x4*)(p.ctx_lo + (o - p.ctx) + dp * 32 + col) = (u32x4){l0[0], l0[1], l1[0], l1[1]};
        }
      }
    } else if (q < T) {
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        const int d = dt * 16 + lg * 4;
        if (d < dh) {
          const f32x4 v = ot[rq][dt] * inv;
          u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *(u32x2*)(o + d) = pk;
          if (p.ctx_lo) store_lo(p.ctx_lo + (o - p.ctx) + d, v, pk);
        }
      }
    }
    if (q < T && lg == 0) p.lse[(long)bh * T + q] = (m[rq] * c + log2f(lt)) * LN2;
  }
}

// DMA: dh == DH == 64 (compile-time, so that the untracked-load prologue below shares no control flow with tracked loads: the
// compiler waits vmcnt(0) wherever a tracked load MIGHT be pending, and would drain the V image with it)
// TPC / HC / NSP / WPWC: a shape known at compile time (TPC = 208: 192 < T <= 208, 12 heads, 2 workgroups x 4 waves per head:
// ViT-B): piece counts, waits, row strides and the tile sequence are constants.  (The same for ViT-L -- 592, 16 heads, 2 x 10
// waves -- measured no gain: its nine-tile key loop dominates and is the same code.) (r03: the same specialisation took 7 %
// off the pair-pipelined backward)
template <int DH, int RQ, bool DMA, int TPC = 0, int HC = 0, int NSP = 0, int WPWC = 0>
__global__ __launch_bounds__(768, 3) void attn_fwd_res_kernel(AttnArgs p) {
  resolve_drop(p.drop);
#ifdef VIT_FWD_STAMP
  unsigned long long tprev_, st_[5] = {0, 0, 0, 0, 0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev_)::"memory");
  const unsigned long long tstart_ = tprev_;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups go to the XCDs round-robin (blockIdx % 8): deal each XCD a contiguous run of logical ids, so the nsplit
  // workgroups that stage the SAME head's K / V sit on one XCD, back to back, and the second one finds them in that L2
  const int wg = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
  const int NH = HC ? HC : p.H, NSPLIT = TPC ? NSP : p.nsplit, WPW = TPC ? WPWC : p.wpw;
  const int bh = wg / NSPLIT, part = wg - bh * NSPLIT, b = bh / NH, h = bh - b * NH;
  const int T = p.T, dh = DMA ? DH : p.dh, ntl = (T + RT - 1) / RT;
  const long ld = 3L * NH * dh;
  const short* qb = p.qkv + (long)b * T * ld + h * dh;
  const short* kb_ = qb + NH * dh;
  const short* vb = kb_ + NH * dh;
  char* Kimg = smem;
  // only the 16-row blocks that hold keys are staged: 208 rows at T = 197 -> 52 KiB per workgroup, so THREE workgroups
  // share a CU's 160 KiB (whole 64-row tiles took 64 KiB: two)
  const int rows_alloc = TPC ? TPC : ((T + 15) & ~15);
  char* Vimg = smem + rows_alloc * (DH * 2);
  const int q00 = (part * WPW + wave) * RQ * 16;
  bf16x8 qf[RQ][DH / 32];
  float m[RQ], l[RQ];
  f32x4 ot[RQ][DH / 16];
  // The wave's Q rows first (plain loads, oldest in the vmcnt order), then K, then V: the waits below are counted, so that
  // the three latencies overlap and the first tile's scores start when K is in (stamps, r03: a wave spent 30 % of its life
  // waiting for K + V together and another 10 % for Q fragments requested only after that)
#pragma unroll
  for (int rq = 0; rq < RQ; ++rq) {
    if constexpr (DMA) {
      // loads the compiler does not track (it would wait vmcnt(0) at their first use and drain V with them): rows past T read
      // the last row again (never stored); the counted wait below covers them -- they are the oldest operations in flight
#pragma unroll
      for (int s = 0; s < DH / 32; ++s) {
        i32x4 v;
        const short* src = qb + (long)min(q00 + rq * 16 + l15, T - 1) * ld + s * 32 + lg * 8;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(src) : "memory");
        qf[rq][s] = __builtin_bit_cast(bf16x8, v);
      }
    } else {
      load_own<DH>(qf[rq], qb, ld, q00 + rq * 16, T, dh, l15, lg);
    }
  }
  bool v_pending = false;
  if constexpr (DMA) {
    // no registers, no zero-fill moves, no address arithmetic per chunk: this kernel saturates the VALU (PMC: 3 waves x 33 %
    // VALU-active per SIMD) and the register-staged form spent ~300 VALU instructions per wave here.  Keys past T are
    // masked to -inf in the edge tile, so the clamped duplicate rows are never used.
    const int nwv = TPC ? WPWC : (int)(blockDim.x >> 6), npc = rows_alloc >> 3;
    dma_rows64(Kimg, kb_, ld, 0, rows_alloc, T, wave, lane, nwv);
    dma_rows64(Vimg, vb, ld, 0, rows_alloc, T, wave, lane, nwv);
    wait_vmcnt_dyn(wave < npc ? (npc - wave + nwv - 1) / nwv : 0);  // all but this wave's V pieces: Q and K are in
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq)
#pragma unroll
      for (int s = 0; s < DH / 32; ++s) asm volatile("" : "+v"(qf[rq][s]));  // uses of Q stay behind the wait
    __builtin_amdgcn_sched_barrier(0);
    v_pending = true;
  } else {
    load_all_tiles2<DH>(Kimg, kb_, ld, Vimg, vb, ld, T, dh, rows_alloc, tid, blockDim.x);
  }
  FWD_ST(0)  // Q + K landed (this wave's pieces)
  __syncthreads();
  FWD_ST(1)  // barrier
  if (q00 >= T) {  // a wave with no query rows: it still owes the workgroup its V pieces and the barrier that publishes them
    if (v_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    return;
  }
#pragma unroll
  for (int rq = 0; rq < RQ; ++rq) {
    m[rq] = -INFINITY;
    l[rq] = 0.f;
#pragma unroll
    for (int i = 0; i < DH / 16; ++i) ot[rq][i] = zero4();
  }
  const float c = p.scale * LOG2E;
#ifdef VIT_FWD_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::"v"(qf[0][0]), "v"(qf[RQ - 1][DH / 32 - 1]));
#endif
  FWD_ST(2)  // Q fragments in registers

  fwd_keyloop<DH, RQ, TPC>(Kimg, Vimg, qf, m, l, ot, T, c, p.drop, bh, q00, l15, lg, [&]() {
    if (v_pending) {  // V in and published before its first fragment read
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  });
  FWD_ST(3)  // key loop
  fwd_finish<DH, RQ, DMA, HC>(p, m, l, ot, b, h, bh, q00, c, l15, lg);
#ifdef VIT_FWD_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FWD_ST(4)  // normalise + stores retired
  {
    const unsigned wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && wid < FWD_ST_WAVES) {
#pragma unroll
      for (int k = 0; k < 5; ++k) g_fwd_st[wid * 8 + k] = st_[k];
      g_fwd_st[wid * 8 + 5] = tstart_;
      g_fwd_st[wid * 8 + 6] = tprev_;
      g_fwd_st[wid * 8 + 7] = 1;
    }
  }
#endif
}

// Dropout keep FLAGS of 4 consecutive query rows at ONE key for the key-owner orientation of the backward kernels, from the
// rows' keys `rk`.  The mask word belongs to a (row, key pair): lanes l15 and l15 ^ 1 hold the two keys of a pair; the even lane
// evaluates rows 0, 1, the odd lane rows 2, 3, and they trade results across the lane pair by DPP (2 words + 2 moves per 4
// elements).  Flags, not multipliers (r03): the draw of (row r, this lane's key) sits in the low or high half of the pair's word by
// the key's parity (= the lane's); shifting the word left by 16 for even keys puts it in the high half either way, and
// "draw >= thr" becomes ONE unsigned compare against thr << 16.  The caller selects with the flags and applies the kept
// elements' 1 / (1 - p) where it is cheapest (an FMA operand, the dV accumulator at the end): 3 VALU per element less than
// building {0, scale} multipliers.
__device__ __forceinline__ void drop_keep4_keyowner(const DropCfg& d, const u32x4& rk, unsigned key, int l15, bool (&keep)[4]) {
  const unsigned odd = (unsigned)l15 & 1u;
  // dropout off: thr << 16 = 0 and every compare below is true -- the flags are formed OUTSIDE the uniform branch, so they are
  // plain compare results (inside it the compiler merged them with the "off" default through a dozen scalar mask instructions
  // per tile: ISA of the A stage, 90 of 586 instructions)
  unsigned h[4] = {~0u, ~0u, ~0u, ~0u};
  if (d.thr) {
    const unsigned ha = drop_bits(odd ? rk[2] : rk[0], key >> 1);
    const unsigned hb = drop_bits(odd ? rk[3] : rk[1], key >> 1);
    const unsigned oa = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ha, 0xB1, 0xF, 0xF, false);
    const unsigned ob = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hb, 0xB1, 0xF, 0xF, false);
    h[0] = odd ? oa : ha; h[1] = odd ? ob : hb; h[2] = odd ? ha : oa; h[3] = odd ? hb : ob;
  }
  const unsigned sh = odd ? 0u : 16u, thr16 = d.thr << 16;
#pragma unroll
  for (int r = 0; r < 4; ++r) keep[r] = (h[r] << sh) >= thr16;
}

// The staged-image prologue of the two resident backward kernels, DMA form (dh == DH == 64).  One workgroup per CU at
// T = 577 (148 KiB of images), six of them one after the other: the register-staged prologue (global -> registers -> LDS, two
// or three full round trips for 148 KiB) sat in the open in front of each one's loop, ~6 of its ~35 us.  Here every piece
// of the two images is requested up front by LDS-DMA IN THE ORDER THE LOOP READS THEM (tile 0 of both images, tile 1, ...),
// the wave's own rows before them by loads the compiler does not track (it would wait vmcnt(0) at their first use and drain
// the images with them), and the loop waits, tile by tile, with a COUNTED vmcnt for this wave's pieces of that tile and a
// barrier that publishes everybody's: the first tile's arithmetic starts when 16 KiB have landed, the other 130 KiB arrive
// underneath it.  Piece jg (8 rows x 128 B) of an image belongs to wave jg % nwaves; vmcnt retires in order, so "all but my
// pieces of later tiles" is one number per tile.
struct ImgDma {
  int nwv, npc, wave, tot;
  __device__ __forceinline__ int mine_below(int lim) const { return lim > wave ? (lim - wave + nwv - 1) / nwv : 0; }
  // outstanding operations this wave may leave when tile kt (pieces < 8 (kt + 1)) is about to be read; PER = DMA instructions per piece
  __device__ __forceinline__ int allowed(int kt, int per) const { return tot - per * mine_below(min(8 * (kt + 1), npc)); }
};
__device__ __forceinline__ void dma_piece64(unsigned img_a, const short* g, long ld, int jg, int T, int lane) {
  const int r = (jg << 3) + (lane >> 3);
  const int c = (lane & 7) ^ swz<64>(r & 63);
  lds_dma16_s(g, __umul24((unsigned)min(r, T - 1), (unsigned)(ld * 2)) + (unsigned)(c * 16), img_a + jg * 1024);
}
__device__ __forceinline__ i32x4 load16_untracked(const short* src) {
  i32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(src) : "memory");
  return v;
}

template <int DH, int RQ, bool DMA = false>
__global__ __launch_bounds__(512) void attn_bwd_dq_res_kernel(AttnArgs p) {
  resolve_drop(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = RT * DH * 2;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = DMA ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int bh = blockIdx.x / p.nsplit, part = blockIdx.x - bh * p.nsplit, b = bh / p.H, h = bh - b * p.H;
  const int T = p.T, dh = DMA ? DH : p.dh, ntl = (T + RT - 1) / RT;
  const long ld = 3L * p.H * dh, ldc = (long)p.H * dh;
  const short* qb = p.qkv + (long)b * T * ld + h * dh;
  const short* kb_ = qb + p.H * dh;
  const short* vb = kb_ + p.H * dh;
  const short* dob = p.dctx + (long)b * T * ldc + h * dh;
  const short* ob = p.ctx + (long)b * T * ldc + h * dh;
  char* Kimg = smem;
  // only the 16-row blocks that hold keys are staged: 208 rows at T = 197 -> 52 KiB per workgroup, so THREE workgroups
  // share a CU's 160 KiB (whole 64-row tiles took 64 KiB: two)
  const int rows_alloc = (T + 15) & ~15;
  char* Vimg = smem + rows_alloc * (DH * 2);
  const int q00 = (part * p.wpw + wave) * RQ * 16;
  float* csum = p.csum_part ? p.csum_part + ((long)(bh / p.H) * p.nsplit * p.wpw + part * p.wpw + wave) * ld + h * dh : nullptr;
  const bool idle = q00 >= T;  // a wave with no query rows (the last workgroup of a head): DMA form, it still owes its pieces and barriers

  bf16x8 qf[RQ][DH / 32], dof[RQ][DH / 32];
  float lse2[RQ], del[RQ];
  f32x4 dqt[RQ][DH / 16];
  ImgDma dm = {(int)(blockDim.x >> 6), rows_alloc >> 3, wave, 0};
  if constexpr (DMA) {
    i32x4 ov[RQ][DH / 32], lv[RQ][DH / 32];
    float lraw[RQ];
    const bool has_lo = p.ctx_lo != nullptr;
    if (!idle) {
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) {
        const long row = min(q00 + rq * 16 + l15, T - 1);  // rows past T read the last row again (masked through lse = +inf)
#pragma unroll
        for (int s = 0; s < DH / 32; ++s) {
          const int col = s * 32 + lg * 8;
          qf[rq][s] = __builtin_bit_cast(bf16x8, load16_untracked(qb + row * ld + col));
          dof[rq][s] = __builtin_bit_cast(bf16x8, load16_untracked(dob + row * ldc + col));
          ov[rq][s] = load16_untracked(ob + row * ldc + col);
          if (has_lo) lv[rq][s] = load16_untracked(p.ctx_lo + (ob - p.ctx) + row * ldc + col);
        }
        asm volatile("global_load_dword %0, %1, off" : "=v"(lraw[rq]) : "v"(p.lse + (long)bh * T + row) : "memory");
      }
    }
    const unsigned Ka = lds_addr_of(Kimg), Va = lds_addr_of(Vimg);
    for (int jg = wave; jg < dm.npc; jg += dm.nwv) {  // ascending piece index = tile order
      dma_piece64(Ka, kb_, ld, jg, T, lane);
      dma_piece64(Va, vb, ld, jg, T, lane);
      dm.tot += 2;
    }
    wait_vmcnt_dyn(dm.allowed(0, 2));  // my own rows (older than every piece) and my pieces of tile 0
    __builtin_amdgcn_sched_barrier(0);
    if (!idle) {
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) {
        asm volatile("" : "+v"(lraw[rq]));
#pragma unroll
        for (int s = 0; s < DH / 32; ++s) {
          asm volatile("" : "+v"(qf[rq][s]), "+v"(dof[rq][s]), "+v"(ov[rq][s]));  // their uses stay behind the wait
          if (has_lo) asm volatile("" : "+v"(lv[rq][s]));
        }
        const int q = q00 + rq * 16 + l15;
        lse2[rq] = q < T ? lraw[rq] * LOG2E : INFINITY;
        float d_ = 0.f;
#pragma unroll
        for (int s = 0; s < DH / 32; ++s) {
          const bf16x8 o = __builtin_bit_cast(bf16x8, ov[rq][s]);
#pragma unroll
          for (int e = 0; e < 8; ++e) d_ += bf2f(o[e]) * bf2f(dof[rq][s][e]);
          if (has_lo) {
            const bf16x8 ol = __builtin_bit_cast(bf16x8, lv[rq][s]);
#pragma unroll
            for (int e = 0; e < 8; ++e) d_ += bf2f(ol[e]) * bf2f(dof[rq][s][e]);
          }
        }
        d_ = grp4_sum(d_);
        if (q < T && lg == 0) p.delta[(long)bh * T + q] = d_;
        del[rq] = q < T ? d_ : 0.f;
#pragma unroll
        for (int i = 0; i < DH / 16; ++i) dqt[rq][i] = zero4();
      }
    }
  } else {
    load_all_tiles2<DH>(Kimg, kb_, ld, Vimg, vb, ld, T, dh, rows_alloc, tid, blockDim.x);
    __syncthreads();
    if (idle) {
      if (csum && lane < DH / 4 && lane * 4 < dh) *(f32x4*)(csum + lane * 4) = zero4();  // an idle wave's partial row
      return;
    }
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
      const int q = q00 + rq * 16 + l15;
      load_own<DH>(qf[rq], qb, ld, q00 + rq * 16, T, dh, l15, lg);
      load_own<DH>(dof[rq], dob, ldc, q00 + rq * 16, T, dh, l15, lg);
      lse2[rq] = q < T ? p.lse[(long)bh * T + q] * LOG2E : INFINITY;
      float d_ = 0.f;
#pragma unroll
      for (int s = 0; s < DH / 32; ++s) {
        const int col = s * 32 + lg * 8;
        if (q < T && col < dh) {
          const bf16x8 o = *(const bf16x8*)(ob + (long)q * ldc + col);
#pragma unroll
          for (int e = 0; e < 8; ++e) d_ += bf2f(o[e]) * bf2f(dof[rq][s][e]);
          if (p.ctx_lo) {
            const bf16x8 ol = *(const bf16x8*)(p.ctx_lo + (ob - p.ctx) + (long)q * ldc + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) d_ += bf2f(ol[e]) * bf2f(dof[rq][s][e]);
          }
        }
      }
      d_ = grp4_sum(d_);
      if (q < T && lg == 0) p.delta[(long)bh * T + q] = d_;
      del[rq] = d_;
#pragma unroll
      for (int i = 0; i < DH / 16; ++i) dqt[rq][i] = zero4();
    }
  }
  const float c = p.scale * LOG2E;
  const unsigned half_cols = (unsigned)((T + 1) >> 1);

  // dropout row keys of this wave's rows; keep flags by compares against thr << 16 (high draw: the word itself, low draw: the
  // word shifted up), the kept elements' 1 / (1 - p) as the FMA's multiplier (r03: 3 VALU per element less than multipliers)
  unsigned rkey[RQ];
#pragma unroll
  for (int rq = 0; rq < RQ; ++rq)
    rkey[rq] = p.drop.thr ? drop_rowkey(p.drop, (unsigned long long)bh * T + (q00 + rq * 16 + l15)) : 0u;
  const unsigned thr16 = p.drop.thr << 16;
  const float dscale = p.drop.thr ? p.drop.scale : 1.0f;
  // One 64-key tile; EDGE = the tile straddles T (per-key validity select).  Full tiles run the body without it (r03: the
  // forward's peeling applied here -- ViT-L's T = 577 walks nine full tiles and one edge tile).
  auto tile = [&](auto edgec, int kt) {
    constexpr bool EDGE = decltype(edgec)::value;
    const int kb = kt * RT;
    const char* Kt = Kimg + kt * TILE;
    const char* Vt = Vimg + kt * TILE;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (EDGE && kb + u * 32 >= T) continue;
      f32x4 ds[RQ][2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * u + jj;
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) ds[rq][jj] = zero4();
        if (!EDGE || kb + j * 16 < T) {
          f32x4 s_[RQ], dp[RQ];
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) s_[rq] = dp[rq] = zero4();
#pragma unroll
          for (int s = 0; s < DH / 32; ++s) {
            const bf16x8 kf = frag_rows<DH>(Kt, j * 16, s, l15, lg);
            const bf16x8 vf = frag_rows<DH>(Vt, j * 16, s, l15, lg);
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) {
              s_[rq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[rq][s], s_[rq], 0, 0, 0);
              dp[rq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[rq][s], dp[rq], 0, 0, 0);
            }
          }
          const unsigned key0 = kb + j * 16 + lg * 4;
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) {
            unsigned ha = ~0u, hb = ~0u;  // dropout off: thr16 = 0, every compare true
            if (p.drop.thr) {
              ha = drop_bits(rkey[rq], key0 >> 1);
              hb = drop_bits(rkey[rq], (key0 >> 1) + 1);
            }
            const bool keep[4] = {(ha << 16) >= thr16, ha >= thr16, (hb << 16) >= thr16, hb >= thr16};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pr = fast_exp2(s_[rq][r] * c - lse2[rq]);
              if (EDGE) pr = ((int)key0 + r < T) ? pr : 0.f;
              ds[rq][jj][r] = pr * fmaf(keep[r] ? dp[rq][r] : 0.f, dscale, -del[rq]);
            }
          }
        }
      }
      bf16x8 df[RQ];
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) df[rq] = pack8(ds[rq][0], ds[rq][1]);
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        const bf16x8 ktf = frag_cols<DH>(Kt, u * 32, (!EDGE || kb + u * 32 + 16 < T) ? u * 32 + 16 : u * 32, dt * 16, l15, lg);
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq)
          dqt[rq][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, df[rq], dqt[rq][dt], 0, 0, 0);
      }
    }
  };
  // DMA form: before a tile is read, this wave's pieces of it have landed (counted wait) and everybody's are published
  // (barrier).  Three rendezvous only -- tile 0; tiles 1-2; everything else -- all while the waves are still in step anyway:
  // a barrier in front of EVERY tile kept the seven waves in lock-step through the whole loop (all of them in their MFMA
  // chains, then all in their exp / dropout arithmetic) and cost more than the prologue it hid (T = 577: 418 -> 434 us).
  auto arrive = [&](int kt) {
    if constexpr (DMA) {
      if (kt == 0) {
        __builtin_amdgcn_s_barrier();  // the wait for tile 0 was the prologue's
      } else if (kt == 1) {
        wait_vmcnt_dyn(dm.allowed(2, 2));
        __builtin_amdgcn_s_barrier();
      } else if (kt == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  };
  {
    using std::integral_constant;
    const int nfull = T / RT;
    for (int kt = 0; kt < nfull; ++kt) {
      arrive(kt);
      if (!DMA || !idle) tile(integral_constant<bool, false>{}, kt);
    }
    if (nfull * RT < T) {
      arrive(nfull);
      if (!DMA || !idle) tile(integral_constant<bool, true>{}, nfull);
    }
  }
  if (DMA && idle) {
    if (csum && lane < DH / 4) *(f32x4*)(csum + lane * 4) = zero4();  // an idle wave's partial row
    return;
  }
  f32x4 cs[DH / 16];
#pragma unroll
  for (int dt = 0; dt < DH / 16; ++dt) cs[dt] = zero4();
#pragma unroll
  for (int rq = 0; rq < RQ; ++rq) {
    const int q = q00 + rq * 16 + l15;
    if (q < T) {
      short* o = p.dqkv + ((long)b * T + q) * ld + h * dh;
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        const int d = dt * 16 + lg * 4;
        if (d < dh) {
          const f32x4 v = dqt[rq][dt] * p.scale;
          u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *(u32x2*)(o + d) = pk;
          cs[dt] += bf_round4(pk);
        }
      }
    }
  }
  if (csum) {  // the query third's bias gradient: this wave's rows, reduced over the batch afterwards
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      const f32x4 t = rows16_sum(cs[dt]);
      const int d = dt * 16 + lg * 4;
      if (l15 == 0 && d < dh) *(f32x4*)(csum + d) = t;
    }
  }
}

template <int DH, int RQ, bool DMA = false>
__global__ __launch_bounds__(512) void attn_bwd_dkv_res_kernel(AttnArgs p) {
  resolve_drop(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = RT * DH * 2;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = DMA ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int bh = blockIdx.x / p.nsplit, part = blockIdx.x - bh * p.nsplit, b = bh / p.H, h = bh - b * p.H;
  const int T = p.T, dh = DMA ? DH : p.dh, ntl = (T + RT - 1) / RT;
  const long ld = 3L * p.H * dh, ldc = (long)p.H * dh;
  const short* qb = p.qkv + (long)b * T * ld + h * dh;
  const short* kb_ = qb + p.H * dh;
  const short* vb = kb_ + p.H * dh;
  const short* dob = p.dctx + (long)b * T * ldc + h * dh;
  // only the 16-row blocks that hold queries are staged (like K / V in the other two kernels): T = 577 -> 592 rows,
  // 2 x 74 KiB + 7.5 KiB of row statistics (each array padded to whole 64-row DMA pieces) fit the CU's 160 KiB
  const int rows_alloc = (T + 15) & ~15, rows_st = (T + 63) & ~63;
  char* Qimg = smem;
  char* Oimg = smem + rows_alloc * (DH * 2);
  float* lse_s = (float*)(smem + 2 * rows_alloc * (DH * 2));
  float* del_s = lse_s + rows_st;
  unsigned* rk_s = (unsigned*)(del_s + rows_st);  // dropout row keys of the head's query rows
  const int k00 = (part * p.wpw + wave) * RQ * 16;
  float* csum = p.csum_part ? p.csum_part + ((long)(bh / p.H) * p.nsplit * p.wpw + part * p.wpw + wave) * ld + p.H * dh + h * dh
                            : nullptr;
  const bool idle = k00 >= T;  // DMA form: an idle wave still owes the workgroup its pieces and barriers
  bf16x8 kf[RQ][DH / 32], vf[RQ][DH / 32];
  f32x4 dkt[RQ][DH / 16], dvt[RQ][DH / 16];
  ImgDma dm = {(int)(blockDim.x >> 6), rows_alloc >> 3, wave, 0};
  if constexpr (DMA) {
    // see the dQ kernel: own rows (untracked), then the raw row statistics (64 rows x 4 B per piece), then the Q / dO images in
    // the order the query loop reads them
    if (!idle) {
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) {
        const long row = min(k00 + rq * 16 + l15, T - 1);  // keys past T: nothing of theirs is stored
#pragma unroll
        for (int s = 0; s < DH / 32; ++s) {
          kf[rq][s] = __builtin_bit_cast(bf16x8, load16_untracked(kb_ + row * ld + s * 32 + lg * 8));
          vf[rq][s] = __builtin_bit_cast(bf16x8, load16_untracked(vb + row * ld + s * 32 + lg * 8));
        }
      }
    }
    const unsigned La = lds_addr_of(lse_s), Da = lds_addr_of(del_s);
    for (int jp = wave; jp < (rows_st >> 6); jp += dm.nwv) {
      const unsigned off = (unsigned)min(jp * 64 + lane, T - 1) * 4u;
      lds_dma4_s(p.lse + (long)bh * T, off, La + jp * 256);
      lds_dma4_s(p.delta + (long)bh * T, off, Da + jp * 256);
    }
    const unsigned Qa = lds_addr_of(Qimg), Oa = lds_addr_of(Oimg);
    for (int jg = wave; jg < dm.npc; jg += dm.nwv) {
      dma_piece64(Qa, qb, ld, jg, T, lane);
      dma_piece64(Oa, dob, ldc, jg, T, lane);
      dm.tot += 2;
    }
    wait_vmcnt_dyn(dm.tot);  // everything older than the image pieces: my own rows and my statistics pieces
    __builtin_amdgcn_sched_barrier(0);
    if (!idle) {
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq)
#pragma unroll
        for (int s = 0; s < DH / 32; ++s) asm volatile("" : "+v"(kf[rq][s]), "+v"(vf[rq][s]));  // uses stay behind the wait
    }
    __builtin_amdgcn_s_barrier();  // everybody's statistics pieces are in
    for (int i = tid; i < rows_alloc; i += blockDim.x) {
      const float lr = lse_s[i], dr = del_s[i];
      lse_s[i] = i < T ? lr * LOG2E : INFINITY;
      del_s[i] = i < T ? dr : 0.f;
      rk_s[i] = p.drop.thr ? drop_rowkey(p.drop, (unsigned long long)bh * T + i) : 0u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // published by the first tile's barrier below
  } else {
    load_all_tiles2<DH>(Qimg, qb, ld, Oimg, dob, ldc, T, dh, rows_alloc, tid, blockDim.x);
    for (int i = tid; i < rows_alloc; i += blockDim.x) {
      lse_s[i] = i < T ? p.lse[(long)bh * T + i] * LOG2E : INFINITY;
      del_s[i] = i < T ? p.delta[(long)bh * T + i] : 0.f;
      rk_s[i] = p.drop.thr ? drop_rowkey(p.drop, (unsigned long long)bh * T + i) : 0u;
    }
    __syncthreads();
    if (idle) {
      if (csum && lane < DH / 4 && lane * 4 < dh) {
        *(f32x4*)(csum + lane * 4) = zero4();
        *(f32x4*)(csum + p.H * dh + lane * 4) = zero4();
      }
      return;
    }
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) {
      load_own<DH>(kf[rq], kb_, ld, k00 + rq * 16, T, dh, l15, lg);
      load_own<DH>(vf[rq], vb, ld, k00 + rq * 16, T, dh, l15, lg);
    }
  }
#pragma unroll
  for (int rq = 0; rq < RQ; ++rq)
#pragma unroll
    for (int i = 0; i < DH / 16; ++i) dkt[rq][i] = dvt[rq][i] = zero4();
  const float c = p.scale * LOG2E;
  const float dscale = p.drop.thr ? p.drop.scale : 1.0f;

  for (int qt = 0; qt < ntl; ++qt) {
    const int qb0 = qt * RT;
    if constexpr (DMA) {  // three rendezvous, as in the dQ kernel: tile 0; tiles 1-2; the rest
      if (qt == 0 || qt == 1) {
        wait_vmcnt_dyn(dm.allowed(qt ? 2 : 0, 2));
        __builtin_amdgcn_s_barrier();
      } else if (qt == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (idle) continue;
    }
    const char* Qt = Qimg + qt * TILE;
    const char* Ot = Oimg + qt * TILE;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (qb0 + u * 32 >= T) continue;
      u32x2 pdh[RQ][2], dsh[RQ][2];  // P*mask and dS, packed to bf16 as soon as they exist (register pressure)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * u + jj;
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) pdh[rq][jj] = dsh[rq][jj] = (u32x2){0u, 0u};
        if (qb0 + j * 16 < T) {
          f32x4 s_[RQ], dp[RQ];
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) s_[rq] = dp[rq] = zero4();
#pragma unroll
          for (int s = 0; s < DH / 32; ++s) {
            const bf16x8 qfr = frag_rows<DH>(Qt, j * 16, s, l15, lg);
            const bf16x8 ofr = frag_rows<DH>(Ot, j * 16, s, l15, lg);
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) {
              s_[rq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[rq][s], s_[rq], 0, 0, 0);
              dp[rq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[rq][s], dp[rq], 0, 0, 0);
            }
          }
          const f32x4 l4 = *(const f32x4*)(lse_s + qb0 + j * 16 + lg * 4);
          const f32x4 d4 = *(const f32x4*)(del_s + qb0 + j * 16 + lg * 4);
          const u32x4 rk4 = *(const u32x4*)(rk_s + qb0 + j * 16 + lg * 4);
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) {
            const unsigned key = k00 + rq * 16 + l15;
            float pdv[4], dsv[4];
            bool keep[4];
            // the lane pair (l15, l15 ^ 1) holds the two keys of a mask word: two hashes per four elements, traded by DPP, and
            // compare-only flags (key tiles start at even keys: key parity == lane parity)
            drop_keep4_keyowner(p.drop, rk4, key, l15, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pr = fast_exp2(s_[rq][r] * c - l4[r]);  // rows past T carry lse = +inf -> 0
              pdv[r] = keep[r] ? pr : 0.f;  // 1 / (1 - p) goes onto dV once, at the end
              dsv[r] = pr * fmaf(keep[r] ? dp[rq][r] : 0.f, dscale, -d4[r]);
            }
            pdh[rq][jj] = (u32x2){pack2bf(pdv[0], pdv[1]), pack2bf(pdv[2], pdv[3])};
            dsh[rq][jj] = (u32x2){pack2bf(dsv[0], dsv[1]), pack2bf(dsv[2], dsv[3])};
          }
        }
      }
      bf16x8 pf[RQ], df[RQ];
#pragma unroll
      for (int rq = 0; rq < RQ; ++rq) {
        pf[rq] = __builtin_bit_cast(bf16x8, (u32x4){pdh[rq][0][0], pdh[rq][0][1], pdh[rq][1][0], pdh[rq][1][1]});
        df[rq] = __builtin_bit_cast(bf16x8, (u32x4){dsh[rq][0][0], dsh[rq][0][1], dsh[rq][1][0], dsh[rq][1][1]});
      }
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        const int rb1 = (qb0 + u * 32 + 16 < T) ? u * 32 + 16 : u * 32;  // an un-staged block: its P and dS are 0
        const bf16x8 otf = frag_cols<DH>(Ot, u * 32, rb1, dt * 16, l15, lg);
        const bf16x8 qtf = frag_cols<DH>(Qt, u * 32, rb1, dt * 16, l15, lg);
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
          dvt[rq][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf, pf[rq], dvt[rq][dt], 0, 0, 0);
          dkt[rq][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, df[rq], dkt[rq][dt], 0, 0, 0);
        }
      }
    }
  }
  if (DMA && idle) {
    if (csum && lane < DH / 4) {
      *(f32x4*)(csum + lane * 4) = zero4();
      *(f32x4*)(csum + p.H * dh + lane * 4) = zero4();
    }
    return;
  }
  f32x4 csk[DH / 16], csv[DH / 16];
#pragma unroll
  for (int dt = 0; dt < DH / 16; ++dt) csk[dt] = csv[dt] = zero4();
#pragma unroll
  for (int rq = 0; rq < RQ; ++rq) {
    const int key = k00 + rq * 16 + l15;
    if (key < T) {
      short* ok = p.dqkv + ((long)b * T + key) * ld + p.H * dh + h * dh;
      short* ov = ok + p.H * dh;
#pragma unroll
      for (int dt = 0; dt < DH / 16; ++dt) {
        const int d = dt * 16 + lg * 4;
        if (d < dh) {
          const f32x4 a = dkt[rq][dt] * p.scale, v = dvt[rq][dt] * dscale;
          u32x2 pk = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
          u32x2 pv = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *(u32x2*)(ok + d) = pk;
          *(u32x2*)(ov + d) = pv;
          csk[dt] += bf_round4(pk);
          csv[dt] += bf_round4(pv);
        }
      }
    }
  }
  if (csum) {  // the key and value thirds' bias gradients
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) {
      const f32x4 tk = rows16_sum(csk[dt]), tv = rows16_sum(csv[dt]);
      const int d = dt * 16 + lg * 4;
      if (l15 == 0 && d < dh) {
        *(f32x4*)(csum + d) = tk;
        *(f32x4*)(csum + p.H * dh + d) = tv;
      }
    }
  }
}

// ======================================================================================= pipelined fused backward (r03)
// The single-kernel backward (one pass, dS through the LDS, five products instead of the two-kernel path's seven).  Its two
// predecessors -- attn_bwd_fused_kernel<DH, NW> (one workgroup per head, r02) and attn_bwd_persist_kernel (one workgroup per CU
// walking heads, r02) -- ran their pieces one after the other: per (head, half) a phase A, a barrier, global loads for the next
// head, a phase B far too short to cover them, another barrier (140 of the persistent form's 364 us were that exposed chain).
// They were removed in r04 once this form had carried the benchmarked shape for a round: what they still served (dh 64 with
// T < 64 or 209 .. 240, dh 32) now takes the two-kernel resident path, which every test also covers.  This form removes every
// global -> register load and every phase boundary from the critical path: the unit of work is a PAIR of query tiles (32 rows), one barrier per pair,
// and in each barrier interval every stage of the backward runs for a DIFFERENT pair, on different waves:
//
//   iteration g:  top      B(g-1)   waves 4-7: dQ of pair g-1 (wave 4 + j: query tile j >> 1, 16-column tiles 2 (j & 1), + 1)
//                                   = dS(g-1) K over all keys (dS image
//                                   [key][32 q] written by A(g-1), K^T by transposing reads of the head's K image); stores
//                          E        every wave, when pair g-1 ended its head: dK / dV of its key tiles + bias-gradient sums
//                 issue    L(g+2)   LDS-DMA of pair g+2's rows into ring slot (g+2) % 3: Q, dO, O, O_lo, lse of 8 rows per wave,
//                                   ALL issued by waves 0-3 (waves 4-7 issue nothing: they carry B); KV(h+1): the next head's
//                                   K and V images, a few pieces per issuing wave per iteration (pairs 1 .. np-1 of head h)
//                 A(g)              every wave, owner = key (tiles w and w + 8): S = Q K^T, dP = dO V^T, P, dS; dV += P^T dO,
//                                   dK += dS^T Q in registers; dS (bf16) -> dS image g % 2.  K / V fragments are read from the
//                                   LDS images when a head starts.  This is the VALU-bound stage; all else hides under it.
//                 wait              s_waitcnt vmcnt(n): n = what THIS iteration issued, so L(g+1) (one iteration old) is in; in a
//                                   head's last iteration n excludes the K / V pieces issued in it (issued first: they are
//                                   read at the top of the next iteration's A stage, before that iteration's wait)
//                 D(g+1)   waves 0-3: delta = rowsum(dO (O + O_lo)), lse * log2 e, dropout row keys of the 8 rows whose
//                                   data the wave loaded ITSELF (its own vmcnt wait orders them: no barrier needed)
//                 barrier           publishes dS(g), statistics(g+1), the landed rows of pair g+1
//
// Nothing younger than an iteration's DMA is a store (stores sit at the top of the next iteration), so the counted wait
// never drains a store or a prefetch.  13 key tiles at T = 197: waves 0-3 and wave 7 own two, waves 4-6 one -- waves 0-3 carry
// the DMA issue and D, waves 4-7 the B stage, so the four SIMDs (waves w and w + 4) are loaded about evenly.  Rows past T: DMA sources are clamped to row
// T - 1, their probabilities are zero through lse = +inf (queries) / +inf added on the key side.
// LDS: ring 3 x 16 KiB + lse staging 3 KiB + 2 K images + V image + 2 dS images [R][32] + statistics = 156 KiB at R = 208.
// dh = 64, 64 <= T <= 208.  Deterministic, no atomics (basemodule.py:250).
// one LDS-DMA piece: 8 rows x 128 B (image rows row_img .. + 7 of a [rows][64] bf16 matrix whose image row 0 is global row
// grow0) into 1 KiB of consecutive LDS; SWZ: the tile image's XOR swizzle, applied to the lane's GLOBAL chunk
template <bool SWZ>
__device__ __forceinline__ void dma_piece(char* dst, const short* g, long ld, int row_img, int grow0, int T, int lane) {
  const int r = row_img + (lane >> 3), pc = lane & 7;
  const int c = SWZ ? (pc ^ swz<64>(r & 63)) : pc;
  const int grow = min(grow0 + r, T - 1);
  lds_dma16_s(g, __umul24((unsigned)grow, (unsigned)(ld * 2)) + (unsigned)(c * 16), lds_addr_of(dst));
}
// dS image of one pair: [key][32 queries] bf16, 64 B per key row, 8-byte slots (4 queries of one key) XOR-swizzled so that the
// phase-A store (16 consecutive keys at one slot, banks mod 32) and the transposing read (8 consecutive keys x 4 adjacent
// slots, banks mod 64) are both conflict-free: rows k and k + 2 share a 128-byte bank row half, rows k and k + 4 a quarter
// of the 256-byte bank row -> slot ^ bits (k2, k3, k1)
__device__ __forceinline__ int ds2_swz(int key) { return (((key >> 2) & 1) << 2) | (((key >> 3) & 1) << 1) | ((key >> 1) & 1); }
__device__ __forceinline__ int ds2_off(int key, int slot) { return key * 64 + ((slot ^ ds2_swz(key)) << 3); }

constexpr int PIPE_SLOT = 16384, PIPE_NS = 3;
#ifndef VIT_PIPE_SKIP  // timing variants only (python -m vit_amd.build --defs -DVIT_PIPE_SKIP=n --tag ..): 1 no B, 2 no A, 4 no DMA, 8 no D
#define VIT_PIPE_SKIP 0
#endif
static size_t pipe_smem(int T) {
  const size_t R = (T + 15) & ~15;
  return PIPE_NS * PIPE_SLOT + PIPE_NS * 4 * 256 + 2 * R * 128 + R * 128 + 2 * R * 64 + 2 * 96 * 4;
}

#ifdef VIT_PIPE_STAMP  // diagnostic build only (tools/pipe_stamps.py): where an iteration of workgroup 0 goes, per wave
__device__ unsigned long long g_pipe_st[8 * 8];
#define PIPE_ST(K)                                                                   \
  {                                                                                  \
    unsigned long long t_;                                                           \
    __builtin_amdgcn_sched_barrier(0);                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
    __builtin_amdgcn_sched_barrier(0);                                               \
    st_[K] += t_ - tprev_;                                                           \
    tprev_ = t_;                                                                     \
  }
#else
#define PIPE_ST(K)
#endif

struct PipeHead { int bh, b, hh; };

// HC: the head count when it is known at compile time (12: ViT-B -- row strides and head divisions become constants), else 0;
// LOC: the context residual is present (the bf16 training path always passes it)
template <bool FULL7, int HC, bool LOC>
__global__ __launch_bounds__(512) void attn_bwd_pipe_kernel(AttnArgs p) {
  resolve_drop(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DH = 64, TILE = RT * DH * 2, RQ = 2, ND = DH / 16;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NH = HC ? HC : p.H;
  const int T = p.T, BH = p.B * NH;
  const long ld = 3L * NH * DH, ldc = (long)NH * DH, HD = (long)NH * DH;
  // FULL7: 192 < T <= 208 -- every quantity derived from the padded length is a compile-time constant (LDS offsets become
  // immediates, the key-step and tile loops lose their bounds tests)
  const int R = FULL7 ? 208 : ((T + 15) & ~15), nq = R >> 4, np = (nq + 1) >> 1, nks = FULL7 ? 7 : ((T + 31) >> 5);
  char* ring = smem;
  const unsigned ring_a = lds_addr_of(smem);   // DMA destinations are raw LDS addresses
  char* lse_raw = ring + PIPE_NS * PIPE_SLOT;  // [slot][row group][64 words]: raw lse, word l = lse of row (l >> 3) of the group
  char* Kimg0 = lse_raw + PIPE_NS * 4 * 256;   // two K images (heads alternate)
  char* Vimg = Kimg0 + 2 * R * 128;
  char* dSb = Vimg + R * 128;                  // two dS images
  float* stats = (float*)(dSb + 2 * R * 64);   // two sets of [lse 32 | delta 32 | dropout row key 32]
  if ((int)blockIdx.x >= BH) return;
  const int nheads = (BH - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int G = nheads * np;  // pairs this workgroup walks
  const float c = p.scale * LOG2E;
  const bool has_lo = LOC || p.ctx_lo != nullptr;
  const float dscale = p.drop.thr ? p.drop.scale : 1.0f;  // 1 / (1 - p) of the kept probabilities
  // roles: waves 0-3 issue EVERY DMA piece (Q / dO / O / O_lo / lse of 8 rows each, and the next head's K / V) and derive the
  // statistics of the rows they loaded (D); waves 4-7 issue nothing and run the dQ stage (B): wave 4 + j takes query tile j >> 1 of the pair and the two 16-column tiles 2 (j & 1), 2 (j & 1) + 1 of dQ
  const bool is_d = wave < 4, is_b = wave >= 4;
  const int grp = wave & 3;
  // this wave's key tiles: tile w, and of the tiles past 8 first the four for waves 0-3, then wave 7, 6, 5, 4 -- waves 4-7 carry
  // the dQ stage, the heavier extra (stamps), and the second query tile's pair (6, 7) is idle in a head's last iteration
  const int kt0 = wave, kt1 = wave < 4 ? 8 + wave : 19 - wave;
  const bool own0 = kt0 * 16 < R, own1 = kt1 * 16 < R;
  const float kinf[RQ] = {(kt0 * 16 + l15 < T) ? 0.f : INFINITY, (kt1 * 16 + l15 < T) ? 0.f : INFINITY};
  // Lane constants of the stages OUTSIDE the A stage are derived from an opaque copy of the lane id inside each iteration
  // (`ln` below): hoisted out of the loop they stayed live across the A stage, whose registers then spilled to scratch --
  // and a scratch reload is a vector-memory load the compiler waits for with vmcnt(0), draining the LDS-DMA just issued.
  // A DMA piece is 8 rows x 128 B: lane -> (row rl8 of the piece, 16-byte chunk); the image swizzle of rows 8 j + rl8
  // depends on rl8 only.
  int rl8, csw8, clin8;

  auto head_of = [&](int hidx) -> PipeHead {  // the integer division happens here, once per head and pipeline position
    PipeHead h;
    h.bh = (int)blockIdx.x + hidx * (int)gridDim.x;
    h.b = h.bh / NH;
    h.hh = h.bh - h.b * NH;
    return h;
  };
  auto qoff_of = [&](const PipeHead& h) -> long { return (long)h.b * T * ld + (long)h.hh * DH; };
  auto coff_of = [&](const PipeHead& h) -> long { return (long)h.b * T * ldc + (long)h.hh * DH; };

  // byte offsets of the LDS regions (integers: DMA destinations are raw LDS addresses, see lds_dma16_s)
  const int off_lse = PIPE_NS * PIPE_SLOT, off_K = off_lse + PIPE_NS * 4 * 256, off_V = off_K + 2 * R * 128;
  auto issue_L = [&](int bh, long qo, long co, int pp, int s) -> int {
    // waves 0-3 issue every piece of their 8 rows (Q too), waves 4-7 carry the dQ stage -- they issue nothing.  Uniform 64-bit
    // bases (qo, co: element offsets of the head, computed once per head) + 32-bit lane offsets (bytes inside the head's rows:
    // < 208 rows x 3 D x 2 B): the first form spent ~180 cycles per piece, most of it 64-bit address arithmetic (stamps)
    if (is_b) return 0;
    const unsigned slot = ring_a + s * PIPE_SLOT + grp * 1024;
    const unsigned row = (unsigned)min(pp * 32 + grp * 8 + rl8, T - 1);
    const unsigned oq = __umul24(row, (unsigned)(ld * 2)) + (unsigned)csw8 * 2, oc = __umul24(row, (unsigned)(ldc * 2));
    lds_dma16_s(p.qkv + qo, oq, slot);
    lds_dma16_s(p.dctx + co, oc + (unsigned)csw8 * 2, slot + 4096);
    lds_dma16_s(p.ctx + co, oc + (unsigned)clin8 * 2, slot + 8192);
    if (has_lo) lds_dma16_s(p.ctx_lo + co, oc + (unsigned)clin8 * 2, slot + 12288);
    lds_dma4_s(p.lse + (long)bh * T, row * 4u, ring_a + off_lse + (s * 4 + grp) * 256);
    return has_lo ? 5 : 4;
  };
  // ---- KV: pieces [j0, j0 + n) of a head's K image (buffer kbuf) and V image; piece j < R/8: K rows 8j.., else V
  auto issue_KV = [&](long qo, int kbuf, int j0, int n) -> int {
    const int nk = R >> 3;
    const short* kb_ = p.qkv + qo + HD;
    const unsigned Kd = ring_a + off_K + kbuf * (R * 128), Vd = ring_a + off_V;
    int cnt = 0;
    for (int j = j0; j < j0 + n && j < 2 * nk; ++j) {
      const bool isk = j < nk;
      const int jj = isk ? j : j - nk;
      const unsigned row = (unsigned)min(jj * 8 + rl8, T - 1);
      lds_dma16_s(isk ? kb_ : kb_ + HD, __umul24(row, (unsigned)(ld * 2)) + (unsigned)csw8 * 2, (isk ? Kd : Vd) + jj * 1024);
      ++cnt;
    }
    return cnt;
  };
  const int kv_total = 2 * (R >> 3);
  const int kvp = (kv_total + 4 * (np - 1) - 1) / (4 * (np - 1));  // pieces per issuing wave (0-3) per iteration pp = 1 .. np - 1

  bf16x8 kf[RQ][DH / 32], vf[RQ][DH / 32];
  f32x4 dkt[RQ][ND], dvt[RQ][ND], csq[2];
  csq[0] = csq[1] = zero4();
#pragma unroll
  for (int i = 0; i < ND; ++i) {
#pragma unroll
    for (int rq = 0; rq < RQ; ++rq) dkt[rq][i] = dvt[rq][i] = zero4();
  }
  const int bq = (wave >> 1) & 1, bd = wave & 1;  // B stage: query tile of the pair, dt pair (waves 4-7)

  // (head ordinal, pair) of g - 1, g, g + 1, g + 2; g runs from -2
  int hm = 0, pm = -3, h0 = 0, p0 = -2, h1 = 0, p1 = -1, h2 = 0, p2 = 0;
  PipeHead Hm = head_of(0), H0 = Hm, H1 = Hm, H2 = Hm, Hn = Hm;
  long q2 = qoff_of(H2), c2 = coff_of(H2), qn = q2;  // element offsets of heads H2 / Hn: 64-bit products, once per head
#ifdef VIT_PIPE_STAMP
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev_)::"memory");
#endif
  for (int g = -2; g <= G; ++g) {
    const bool vm = g - 1 >= 0 && g - 1 < G, v0 = g >= 0 && g < G, v1 = g + 1 >= 0 && g + 1 < G, v2 = g + 2 < G;
    int ln = lane;
    asm volatile("" : "+v"(ln));  // opaque: what derives from it is recomputed per iteration, not kept across the A stage
    const int l15o = ln & 15, lgo = ln >> 4;
    rl8 = ln >> 3;
    csw8 = ((ln & 7) ^ (rl8 & 6)) * 8;
    clin8 = (ln & 7) * 8;
    // ------------------------------------------------------------------ top: B(g-1), head-end epilogue (stores)
    if (vm) {
      const bool head_done = pm == np - 1;
      if (is_b && !(VIT_PIPE_SKIP & 1)) {
        const int qt = pm * 2 + bq;
        if (qt < nq) {
          f32x4 dq0 = zero4(), dq1 = zero4();
          // the transposing reads of the dS image (keys kb + 4 lg + tq (+ 16), this wave's query tile) and of the K image
          // (same keys, the wave's two 16-column tiles); kb is a multiple of 32, which leaves both swizzles alone
          const int tq = l15o >> 2, tp = l15o & 3, krow = 4 * lgo + tq;
          int k_lane[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int col = (bd * 2 + i) * 16 + 4 * tp;
            k_lane[i] = krow * 128 + ((((col >> 3)) ^ (krow & 6)) << 4) + ((col >> 2) & 1) * 8;
          }
          const char* dcol = dSb + ((g - 1) & 1) * (R * 64) + ds2_off(krow, bq * 4 + tp);
          const char* Kh = Kimg0 + (hm & 1) * (R * 128);
          const char* ka = Kh + k_lane[0];
          const char* kb2 = Kh + k_lane[1];
          // T <= 208: at most 7 key steps of 32.  Branch-free (a step past the last reads step 0 again and its dS fragment is
          // zeroed; a 16-key block that is not staged reads the block before it, zeroed likewise), so that the fragment reads
          // of two steps are in flight while the MFMAs of the two steps before them run: as one basic block per step the
          // stage was a chain of 7 LDS round trips (3 900 cycles per pair in the stamps, the critical path of the iteration).
          struct BFrag { bf16x8 ds, a, b; };
          auto bload = [&](int ks) -> BFrag {
            // FULL7 (192 < T <= 208, the ViT-B sequence): 7 key steps, the last one half full -- known at compile time, so the
            // offsets are immediates and nothing is selected (70 of the stage's 118 VALU instructions were these adds / selects)
            const bool on = FULL7 || ks < nks, hi_ok = FULL7 ? ks < 6 : (on && ks * 32 + 16 < R);
            const int od = on ? ks * 2048 : 0, okk = on ? ks * 4096 : 0, oh = hi_ok ? 1 : 0;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(dcol + od));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(dcol + od + oh * 1024));
            const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(ka + okk));
            const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(ka + okk + oh * 2048));
            const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(kb2 + okk));
            const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(kb2 + okk + oh * 2048));
            const bf16x4 z = {0, 0, 0, 0};
            lo = on ? lo : z;
            hi = hi_ok ? hi : z;
            BFrag f;
            f.ds = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            f.a = (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            f.b = (bf16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            return f;
          };
          auto bmma = [&](const BFrag& f) {
            dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a, f.ds, dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b, f.ds, dq1, 0, 0, 0);
          };
          {  // BD steps of fragments in flight (12 VGPRs each); 3 and 4 measured the same as 2 (r03: 307 us each)
            constexpr int BD = 2;
            BFrag f[BD];
#pragma unroll
            for (int i = 0; i < BD; ++i) f[i] = bload(i);
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
              bmma(f[ks % BD]);
              if (ks + BD < 7) f[ks % BD] = bload(ks + BD);
            }
          }
          const int q = qt * 16 + l15o;
          const f32x4 v0_ = dq0 * p.scale, v1_ = dq1 * p.scale;
          u32x2 pa = {pack2bf(v0_[0], v0_[1]), pack2bf(v0_[2], v0_[3])};
          u32x2 pb = {pack2bf(v1_[0], v1_[1]), pack2bf(v1_[2], v1_[3])};
          if (q < T) {
            csq[0] += bf_round4(pa);
            csq[1] += bf_round4(pb);
          }
          const int col = widen_pair(pa, pb, lgo);
          if (q < T)
            *(u32x4*)(p.dqkv + ((long)Hm.b * T + q) * ld + Hm.hh * DH + bd * 32 + col) = (u32x4){pa[0], pa[1], pb[0], pb[1]};
        }
      }
      PIPE_ST(6)  // B(g-1) alone
      if (head_done) {  // dK, dV of this wave's key tiles of head hm; per-wave column sums of everything this wave stored
        f32x4 csk[ND], csv[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) csk[dt] = csv[dt] = zero4();
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
          const int key = (rq ? kt1 : kt0) * 16 + l15o;
          const bool okk = key < T;
          short* ok = p.dqkv + ((long)Hm.b * T + key) * ld + HD + Hm.hh * DH;
#pragma unroll
          for (int dp = 0; dp < 2; ++dp) {
            u32x2 pk[2], pv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const f32x4 a = dkt[rq][dp * 2 + i] * p.scale, v = dvt[rq][dp * 2 + i] * dscale;
              pk[i] = (u32x2){pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
              pv[i] = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
              if (okk) {
                csk[dp * 2 + i] += bf_round4(pk[i]);
                csv[dp * 2 + i] += bf_round4(pv[i]);
              }
            }
            const int col = widen_pair(pk[0], pk[1], lgo);
            widen_pair(pv[0], pv[1], lgo);
            if (okk) {
              *(u32x4*)(ok + dp * 32 + col) = (u32x4){pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
              *(u32x4*)(ok + HD + dp * 32 + col) = (u32x4){pv[0][0], pv[0][1], pv[1][0], pv[1][1]};
            }
          }
        }
        if (p.csum_part) {  // one partial row per (batch, wave): [q third | k third | v third], this head's 64 columns of each
          float* csum = p.csum_part + ((long)Hm.b * 8 + wave) * ld + Hm.hh * DH;
#pragma unroll
          for (int dt = 0; dt < ND; ++dt) {
            f32x4 tq_ = zero4();  // a B wave summed dQ over its two 16-column tiles only
            if (is_b && (dt >> 1) == bd) tq_ = rows16_sum(csq[dt & 1]);
            const f32x4 tk = rows16_sum(csk[dt]), tv = rows16_sum(csv[dt]);
            const int d = dt * 16 + lgo * 4;
            if (l15o == 0) {
              *(f32x4*)(csum + d) = tq_;
              *(f32x4*)(csum + HD + d) = tk;
              *(f32x4*)(csum + 2 * HD + d) = tv;
            }
          }
        }
        csq[0] = csq[1] = zero4();
#pragma unroll
        for (int i = 0; i < ND; ++i) {
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) dkt[rq][i] = dvt[rq][i] = zero4();
        }
      }
    }
    PIPE_ST(0)  // top: B(g-1) + head-end epilogue
    // ------------------------------------------------------------------ issue: next head's K / V images, pair g + 2
    // INVARIANT of every hand-counted wait below: a DMA piece may be read only after a wait of the wave that issued it AND a
    // barrier, both at least one iteration newer than its issue.  L pieces: issued in iteration g for pair g + 2, covered by
    // the wait of iteration g + 1 (which leaves only ITS OWN issues in flight), read from iteration g + 2 on.  K / V pieces of
    // the next head are read at the TOP of that head's first A stage, i.e. before that iteration's wait: the ones issued in a
    // head's LAST iteration are therefore waited for in that same iteration (they are issued before the L pieces and vmcnt
    // retires in order, so the wait leaves only the L pieces in flight; the whole A stage lies between issue and wait).
    int nissued = 0, nkv_now = 0;
    if (g == -2) {  // prologue: the first head's images, spread over the waves
      const int per = (kv_total + 7) >> 3;
      nissued += issue_KV(q2, 0, wave * per, per);
    } else if (is_d && v0 && p0 >= 1 && h0 + 1 < nheads) {
      if (p0 == 1) {
        Hn = head_of(h0 + 1);
        qn = qoff_of(Hn);
      }
      const int n_ = issue_KV(qn, (h0 + 1) & 1, ((p0 - 1) * 4 + wave) * kvp, kvp);
      nissued += n_;
      if (p0 == np - 1) nkv_now = n_;  // the head's last iteration: these must have landed before its closing barrier
    }
    if (v2 && !(VIT_PIPE_SKIP & 4)) nissued += issue_L(H2.bh, q2, c2, p2, (g + 2) % PIPE_NS);
    PIPE_ST(1)  // DMA issue
    // ------------------------------------------------------------------ A(g)
    if (v0) {
      if (p0 == 0) {  // a head starts: this wave's K / V rows out of the images (landed and published an iteration ago or more)
        const char* Kh = Kimg0 + (h0 & 1) * (R * 128);
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
          const int k0 = (rq ? kt1 : kt0) * 16;
          const bool ex = rq ? own1 : own0;
#pragma unroll
          for (int s = 0; s < DH / 32; ++s) {
            kf[rq][s] = ex ? frag_rows<DH>(Kh + (k0 >> 6) * TILE, k0 & 63, s, l15, lg) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            vf[rq][s] = ex ? frag_rows<DH>(Vimg + (k0 >> 6) * TILE, k0 & 63, s, l15, lg) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
          }
        }
      }
      if (own0 && !(VIT_PIPE_SKIP & 2)) {
        const char* Qt = ring + (g % PIPE_NS) * PIPE_SLOT;
        const char* Ot = Qt + 4096;
        const float* lse_s = stats + (g & 1) * 96;
        const float* del_s = lse_s + 32;
        const unsigned* rk_s = (const unsigned*)(del_s + 32);
        char* dSw = dSb + (g & 1) * (R * 64);
        u32x2 pdh[RQ][2], dsh[RQ][2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) pdh[rq][jj] = dsh[rq][jj] = (u32x2){0u, 0u};
          if (p0 * 32 + jj * 16 < R) {
            f32x4 s_[RQ], dp[RQ];
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) s_[rq] = dp[rq] = zero4();
#pragma unroll
            for (int s = 0; s < DH / 32; ++s) {
              const bf16x8 qfr = frag_rows<DH>(Qt, jj * 16, s, l15, lg);
              const bf16x8 ofr = frag_rows<DH>(Ot, jj * 16, s, l15, lg);
#pragma unroll
              for (int rq = 0; rq < RQ; ++rq) {
                s_[rq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[rq][s], s_[rq], 0, 0, 0);
                dp[rq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ofr, vf[rq][s], dp[rq], 0, 0, 0);
              }
            }
            const f32x4 l4 = *(const f32x4*)(lse_s + jj * 16 + lg * 4);
            const f32x4 d4 = *(const f32x4*)(del_s + jj * 16 + lg * 4);
            const u32x4 rk4 = *(const u32x4*)(rk_s + jj * 16 + lg * 4);
#pragma unroll
            for (int rq = 0; rq < RQ; ++rq) {
              if (rq == 1 && !own1) continue;
              const unsigned key = (rq ? kt1 : kt0) * 16 + l15;
              float pdv[4], dsv[4];
              bool keep[4];
              drop_keep4_keyowner(p.drop, rk4, key, l15, keep);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                // queries past T carry lse = +inf, keys past T add +inf: probability 0 either way
                const float pr = fast_exp2(s_[rq][r] * c - (l4[r] + kinf[rq]));
                pdv[r] = keep[r] ? pr : 0.f;  // the kept elements' 1 / (1 - p) is applied to dV once, when the head ends
                dsv[r] = pr * fmaf(keep[r] ? dp[rq][r] : 0.f, dscale, -d4[r]);
              }
              pdh[rq][jj] = (u32x2){pack2bf(pdv[0], pdv[1]), pack2bf(pdv[2], pdv[3])};
              dsh[rq][jj] = (u32x2){pack2bf(dsv[0], dsv[1]), pack2bf(dsv[2], dsv[3])};
              *(u32x2*)(dSw + ds2_off((int)key, jj * 4 + lg)) = dsh[rq][jj];
            }
          }
        }
        bf16x8 pf[RQ], df[RQ];
#pragma unroll
        for (int rq = 0; rq < RQ; ++rq) {
          pf[rq] = __builtin_bit_cast(bf16x8, (u32x4){pdh[rq][0][0], pdh[rq][0][1], pdh[rq][1][0], pdh[rq][1][1]});
          df[rq] = __builtin_bit_cast(bf16x8, (u32x4){dsh[rq][0][0], dsh[rq][0][1], dsh[rq][1][0], dsh[rq][1][1]});
        }
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) {
          // rows 16..31 of the slot always hold rows (clamped duplicates past T; their P and dS are 0)
          const bf16x8 otf = frag_cols<DH>(Ot, 0, 16, dt * 16, l15, lg);
          const bf16x8 qtf = frag_cols<DH>(Qt, 0, 16, dt * 16, l15, lg);
#pragma unroll
          for (int rq = 0; rq < RQ; ++rq) {
            if (rq == 1 && !own1) continue;
            dvt[rq][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf, pf[rq], dvt[rq][dt], 0, 0, 0);
            dkt[rq][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, df[rq], dkt[rq][dt], 0, 0, 0);
          }
        }
      }
    }
    PIPE_ST(2)  // A(g)
    // ------------------------------------------------------------------ my pieces of pair g + 1 (one iteration old) are in
    wait_vmcnt_dyn(nissued - nkv_now);
    PIPE_ST(3)  // counted wait
    // ------------------------------------------------------------------ D(g+1): statistics of the 8 rows this wave loaded
    if (v1 && is_d && !(VIT_PIPE_SKIP & 8)) {
      const int s1 = (g + 1) % PIPE_NS;
      const char* slot = ring + s1 * PIPE_SLOT;
      const int rl = grp * 8 + rl8, ch = ln & 7;
      const bf16x8 d8 = *(const bf16x8*)(slot + 4096 + tile_off<DH>(rl, ch));
      const bf16x8 o8 = *(const bf16x8*)(slot + 8192 + rl * 128 + ch * 16);
      bf16x8 l8 = {0, 0, 0, 0, 0, 0, 0, 0};
      if (has_lo) l8 = *(const bf16x8*)(slot + 12288 + rl * 128 + ch * 16);
      const float lraw = *(const float*)(lse_raw + (s1 * 4 + grp) * 256 + ln * 4);
      float d_ = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d_ += (bf2f(o8[e]) + bf2f(l8[e])) * bf2f(d8[e]);
      d_ = sum_lanes_cpr<8>(d_);
      const int grow = p1 * 32 + rl;
      if (ch == 0) {
        float* st = stats + ((g + 1) & 1) * 96;
        st[rl] = grow < T ? lraw * LOG2E : INFINITY;
        st[32 + rl] = grow < T ? d_ : 0.f;
        ((unsigned*)st)[64 + rl] = p.drop.thr ? drop_rowkey(p.drop, (unsigned long long)H1.bh * T + min(grow, T - 1)) : 0u;
        if (grow < T) p.delta[(long)H1.bh * T + grow] = d_;
      }
    }
    PIPE_ST(4)  // D(g+1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PIPE_ST(5)  // barrier
    hm = h0; pm = p0; h0 = h1; p0 = p1; h1 = h2; p1 = p2;
    Hm = H0; H0 = H1; H1 = H2;
    if (++p2 == np) {
      p2 = 0;
      ++h2;
      if (h2 < nheads) {
        H2 = head_of(h2);
        q2 = qoff_of(H2);
        c2 = coff_of(H2);
      }
    }
  }
#ifdef VIT_PIPE_STAMP
  if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) g_pipe_st[wave * 8 + k] = st_[k];
    g_pipe_st[wave * 8 + 7] = (unsigned long long)(G + 3);
  }
#endif
}
#ifdef VIT_FWD_STAMP
}  // namespace vit
extern "C" int vit_debug_fwd_stamps(unsigned long long* host, int reset) {  // host: FWD_ST_WAVES * 8 words
  if (reset) {
    void* d = nullptr;
    if (hipGetSymbolAddress(&d, HIP_SYMBOL(vit::g_fwd_st)) != hipSuccess) return -1;
    return (int)hipMemset(d, 0, sizeof(unsigned long long) * FWD_ST_WAVES * 8);
  }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vit::g_fwd_st), sizeof(unsigned long long) * FWD_ST_WAVES * 8);
}
namespace vit {
#endif
#ifdef VIT_PIPE_STAMP
}  // namespace vit
extern "C" int vit_debug_pipe_stamps(unsigned long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(vit::g_pipe_st), 64 * sizeof(unsigned long long));
}
namespace vit {
#endif

static bool pipe_fits(int T, int dh) { return dh == 64 && T >= 64 && T <= 208 && pipe_smem(T) <= 160 * 1024; }

// vit_set_option("attn_bwd_fused"): 0 = two-kernel backward everywhere; non-zero (default 4; 1 .. 3 named forms that no longer
// exist and mean the same) = the pair-pipelined single kernel where it fits (dh 64, 64 <= T <= 208), the two-kernel path elsewhere
int g_attn_bwd_fused = 4;
int g_attn_fwd_waves = 12;  // vit_set_option("attn_fwd_waves"): most waves per workgroup of the resident forward (8 or 12)

// resident kernels: a (batch, head)'s whole K/V (or Q/dO) in the LDS -- at head_dim 64 up to T = 592 rows (2 x 74 KiB; the
// backward adds 4.6 KiB of row statistics): ViT-L/16 384^2 (T = 577) fits, one workgroup per CU, three workgroups of 7
// waves per head
constexpr int RES_MAX_T = 592, RES_MAX_DH = 64, RES_RQ = 2;
int g_attn_res_max_t = RES_MAX_T;  // vit_set_option("attn_res_max_t"): larger T goes to the tiled kernels
int g_attn_split = 2;  // vit_set_option("attn_split"): workgroups per (batch, head) in the resident kernels

static bool res_fits(int T, int dh) {  // the dK/dV kernel's LDS: the staged rows of Q and dO + three f32 rows of statistics
  if (dh & 7) return false;  // 8-byte head offsets: the tiled kernels (ld_head8); the resident ones stage 16-byte pieces
  const size_t dhp = dh <= 32 ? 32 : 64, rows = (T + 15) & ~15, rows_st = (T + 63) & ~63;
  return 2 * rows * dhp * 2 + 3 * rows_st * 4 <= 160 * 1024;
}

// max_waves: 8 for the backward kernels (their csum partial rows share one geometry; dK/dV needs 216 VGPRs = 2 waves per
// SIMD), 12 for the forward (166 VGPRs = 3 per SIMD).  It matters where ONE workgroup fills the LDS (T = 577: 148 KiB of
// K / V): 19 waves' worth of query tiles as 3 x 7 waves left a CU with 1.75 waves per SIMD in an issue-bound kernel; 2 x 10 is
// 2.5 per SIMD and stages K / V twice per head instead of three times (r03).
static void res_geometry(int T, int* nsplit, int* wpw, int max_waves = 8) {
  const int nq = cdiv(T, 16), nw = cdiv(nq, RES_RQ);
  *nsplit = std::max(1, std::min(std::max(g_attn_split, cdiv(nw, max_waves)), nw));
  *wpw = cdiv(nw, *nsplit);
  *nsplit = cdiv(nw, *wpw);
}

template <void (*FN)(AttnArgs)>
static int launch_res(const AttnArgs& a, size_t smem, hipStream_t st, int max_waves = 8) {
  // dynamic LDS above 64 KiB needs the attribute; set it to the CU's 160 KiB once per KERNEL (the template parameter is the
  // kernel itself, not its type: all resident kernels share one function-pointer type)
  static bool done = false;
  if (!done) {
    VIT_HIP(hipFuncSetAttribute((const void*)FN, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done = true;
  }
  // several workgroups per (batch, head), each staging the whole K / V (or Q / dO) but owning a share of the row tiles:
  // with 4-wave workgroups three of them fit a CU (150 KiB of LDS, 12 of the 12 wave slots 152 VGPRs leave), so the
  // staging latency of one hides behind the key loops of the others; one 7-wave workgroup per CU paid it in the open.
  AttnArgs b = a;
  res_geometry(a.T, &b.nsplit, &b.wpw, max_waves);
  hipLaunchKernelGGL(FN, dim3(a.B * a.H * b.nsplit), dim3(b.wpw * 64), smem, st, b);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

#define DISPATCH_RES(KERNEL, a, smem_expr, st, rc)                                          \
  do {                                                                                      \
    if (a.dh <= 32) { constexpr int DH_ = 32; rc = launch_res<KERNEL<32, RES_RQ>>(a, smem_expr, st); }        \
    else { constexpr int DH_ = 64; rc = launch_res<KERNEL<64, RES_RQ>>(a, smem_expr, st); }                    \
  } while (0)
// the two resident backward kernels: head_dim exactly 64 takes the DMA prologue (images requested in reading order, per-tile
// counted waits); g_attn_bwd_dma = 0 (vit_set_option("attn_bwd_dma")) keeps the register-staged form for A/B runs
int g_attn_bwd_dma = 1;
#define DISPATCH_RES_BWD(KERNEL, a, smem_expr, st, rc)                                      \
  do {                                                                                      \
    if (a.dh == 64 && g_attn_bwd_dma) { constexpr int DH_ = 64; rc = launch_res<KERNEL<64, RES_RQ, true>>(a, smem_expr, st); } \
    else if (a.dh <= 32) { constexpr int DH_ = 32; rc = launch_res<KERNEL<32, RES_RQ, false>>(a, smem_expr, st); } \
    else { constexpr int DH_ = 64; rc = launch_res<KERNEL<64, RES_RQ, false>>(a, smem_expr, st); }             \
  } while (0)

// ======================================================================================= fp32 attention (precision '32')
// Exact-arithmetic path behind the reference's default precision: fp32 in, fp32 FMAs, fp32 out; one wave per query row
// (forward, dQ) or key row (dK/dV); scores / probabilities of the row live in the wave's LDS slice.  The fp32 matrix
// instructions run at the vector rate on gfx950, so nothing is lost by using the vector units; this path is for
// parity-grade runs, the bf16 kernels above are the throughput path.  qkv: f32 [B*T, 3*H*dh].
struct Attn32Args {
  const float* qkv; float* ctx; float* lse; float* probs;
  const float* dctx; float* delta; float* dqkv;
  int B, H, T, dh, Tp;
  float scale;
  DropCfg drop;
};

__device__ __forceinline__ float drop_mult(const DropCfg& d, unsigned long long row, unsigned half_cols, unsigned col) {
  if (!d.thr) return 1.f;
  (void)half_cols;
  const unsigned h = drop_bits(drop_rowkey(d, row), col >> 1);
  const unsigned r16 = (col & 1) ? (h >> 16) : (h & 0xFFFFu);
  return r16 >= d.thr ? d.scale : 0.f;
}
__device__ __forceinline__ float dot_row(const float* __restrict__ a_lds, const float* __restrict__ g, int dh) {
  float s = 0.f;
  for (int d = 0; d < dh; d += 4) {
    const f32x4 x = *(const f32x4*)(a_lds + d), y = *(const f32x4*)(g + d);
    s = fmaf(x[0], y[0], s); s = fmaf(x[1], y[1], s); s = fmaf(x[2], y[2], s); s = fmaf(x[3], y[3], s);
  }
  return s;
}

// MODE 0: forward (ctx, lse, optional probs)   MODE 1: dQ (+ delta)
template <int MODE>
__global__ __launch_bounds__(256) void attn32_row_kernel(Attn32Args p) {
  resolve_drop(p.drop);
  extern __shared__ __attribute__((aligned(16))) float sm32[];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wib;  // (b*H + h)*T + q
  if (row >= (long)p.B * p.H * p.T) return;      // whole waves only; no block barrier is used below
  const int T = p.T, dh = p.dh, H = p.H;
  const int q = (int)(row % T);
  const long bh = row / T;
  const int h = (int)(bh % H);
  const long b = bh / H;
  const long ld = 3L * H * dh, ldc = (long)H * dh;
  float* pr = sm32 + wib * (2 * p.Tp + 2 * 128);  // [Tp] p or p*mask, [Tp] ds, [128] q row, [128] dO row
  float* ds = pr + p.Tp;
  float* qrow = ds + p.Tp;
  float* dorow = qrow + 128;
  const float* qp = p.qkv + (b * T + q) * ld + h * dh;
  const float* kbase = p.qkv + b * T * ld + H * dh + h * dh;
  const float* vbase = kbase + H * dh;
  for (int d = lane; d < dh; d += 64) {
    qrow[d] = qp[d];
    if (MODE == 1) dorow[d] = p.dctx[(b * T + q) * ldc + h * dh + d];
  }
  __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order; this only pins the compiler's schedule
  const unsigned half_cols = (unsigned)((T + 1) >> 1);
  const unsigned long long drow = (unsigned long long)row;
  if (MODE == 0) {
    float mx = -INFINITY;
    for (int k = lane; k < T; k += 64) {
      const float sc = dot_row(qrow, kbase + (long)k * ld, dh) * p.scale;
      pr[k] = sc;
      mx = fmaxf(mx, sc);
    }
    __builtin_amdgcn_wave_barrier();
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < T; k += 64) {
      const float e = expf(pr[k] - mx);
      pr[k] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int k = lane; k < T; k += 64) {
      const float pk = pr[k] * inv;
      if (p.probs) p.probs[row * T + k] = pk;
      pr[k] = pk * drop_mult(p.drop, drow, half_cols, (unsigned)k);
    }
    __builtin_amdgcn_wave_barrier();
    if (p.lse && lane == 0) p.lse[row] = mx + logf(sum);
    if (p.ctx) {
      for (int d = lane; d < dh; d += 64) {
        float acc = 0.f;
        for (int k = 0; k < T; ++k) acc = fmaf(pr[k], vbase[(long)k * ld + d], acc);
        p.ctx[(b * T + q) * ldc + h * dh + d] = acc;
      }
    }
  } else {
    const float lse = p.lse[row];
    float dl = 0.f;
    for (int d = lane; d < dh; d += 64) dl = fmaf(dorow[d], p.ctx[(b * T + q) * ldc + h * dh + d], dl);
    dl = wave_sum(dl);
    if (lane == 0) p.delta[row] = dl;
    for (int k = lane; k < T; k += 64) {
      const float sc = dot_row(qrow, kbase + (long)k * ld, dh) * p.scale;
      const float pk = expf(sc - lse);
      const float dp = dot_row(dorow, vbase + (long)k * ld, dh);
      ds[k] = pk * (dp * drop_mult(p.drop, drow, half_cols, (unsigned)k) - dl);
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < dh; d += 64) {
      float acc = 0.f;
      for (int k = 0; k < T; ++k) acc = fmaf(ds[k], kbase[(long)k * ld + d], acc);
      p.dqkv[(b * T + q) * ld + h * dh + d] = acc * p.scale;
    }
  }
}

// dK, dV: one wave per key row
__global__ __launch_bounds__(256) void attn32_dkv_kernel(Attn32Args p) {
  resolve_drop(p.drop);
  extern __shared__ __attribute__((aligned(16))) float sm32[];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wib;  // (b*H + h)*T + key
  if (row >= (long)p.B * p.H * p.T) return;
  const int T = p.T, dh = p.dh, H = p.H;
  const int key = (int)(row % T);
  const long bh = row / T;
  const int h = (int)(bh % H);
  const long b = bh / H;
  const long ld = 3L * H * dh, ldc = (long)H * dh;
  float* pd = sm32 + wib * (2 * p.Tp + 2 * 128);
  float* ds = pd + p.Tp;
  float* krow = ds + p.Tp;
  float* vrow = krow + 128;
  const float* qbase = p.qkv + b * T * ld + h * dh;
  const float* kp = qbase + (long)key * ld + H * dh;
  const float* dobase = p.dctx + b * T * ldc + h * dh;
  for (int d = lane; d < dh; d += 64) {
    krow[d] = kp[d];
    vrow[d] = kp[H * dh + d];
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned half_cols = (unsigned)((T + 1) >> 1);
  for (int q = lane; q < T; q += 64) {
    const float sc = dot_row(krow, qbase + (long)q * ld, dh) * p.scale;
    const float pk = expf(sc - p.lse[bh * T + q]);
    const float dp = dot_row(vrow, dobase + (long)q * ldc, dh);
    const float m = drop_mult(p.drop, (unsigned long long)(bh * T + q), half_cols, (unsigned)key);
    pd[q] = pk * m;
    ds[q] = pk * (dp * m - p.delta[bh * T + q]);
  }
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < dh; d += 64) {
    float av = 0.f, ak = 0.f;
    for (int q = 0; q < T; ++q) {
      av = fmaf(pd[q], dobase[(long)q * ldc + d], av);
      ak = fmaf(ds[q], qbase[(long)q * ld + d], ak);
    }
    float* o = p.dqkv + (b * T + key) * ld + H * dh + h * dh + d;
    o[0] = ak * p.scale;
    o[H * dh] = av;
  }
}

// ------------------------------------------------------------------------------------------------ fp32 attention on MFMA (r03)
// The one-wave-per-row kernels above are exact but slow: 258 of the 368 ms of a ViT-B step in precision '32' (r03 profile).
// gfx950 has f32-input matrix instructions (v_mfma_f32_16x16x4_f32: exact f32 products and f32 accumulation, bit for bit
// a k-ordered fmaf chain, at the f32 vector rate per instruction but 64 lanes x 16 results each), so the same flash-style
// tiling as the bf16 kernels runs on them: 4 waves x 16 rows, 64-row K / V (or Q / dO) tiles of f32 in the LDS, the swapped
// orientation that keeps the softmax statistics lane-local, and the accumulator tile of one product being the B operand of
// the next (a lane's register r IS the k-slot (lane >> 4) of MFMA step r: no lane movement).  head_dim 64 only (ViT-B / -L);
// other head sizes and the attention-map output stay on the kernels above.
// Operand maps of v_mfma_f32_16x16x4_f32: A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15] (one float per lane each),
// C / D as for every 16x16 MFMA.  The contraction over d (64) takes 16 steps; step s uses d = 16 g + s for lane group g, so a
// lane's 16 operand values are 64 CONTIGUOUS bytes of its row (4 x ds_read_b128).
// LDS tile [64 rows][64 f32]: row r at r * 256, 16-byte chunk c at ((c ^ sw(r)) << 4), sw(r) = (r & 3) | ((r & 8) ? 12 : 0):
// the row reads (a 16-lane group = 8 rows at chunk i of one lane group and 8 rows at chunk i + 4 of the next) tile the
// 256-byte bank row; the per-element reads of the third product are 2-way at worst, one per 32-cycle MFMA.
__device__ __forceinline__ int t32_off(int r, int c) { return r * 256 + ((c ^ ((r & 3) | ((r & 8) ? 12 : 0))) << 4); }
__device__ __forceinline__ void load_tile32(char* img, const float* g, long ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i, r = q >> 4, c = q & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + r < nrows) v = *(const f32x4*)(g + (long)(row0 + r) * ld + c * 4);
    *(f32x4*)(img + t32_off(r, c)) = v;
  }
}
// the 16 operand values of row (rb + l15) for the 16 contraction steps: d = 16 g + s
__device__ __forceinline__ void frag32_rows(float (&f)[16], const char* img, int rb, int l15, int lg) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = *(const f32x4*)(img + t32_off(rb + l15, 4 * lg + i));
    f[4 * i] = v[0]; f[4 * i + 1] = v[1]; f[4 * i + 2] = v[2]; f[4 * i + 3] = v[3];
  }
}
__device__ __forceinline__ void load_own32(float (&f)[16], const float* g, long ld, int row, int nrows, int lg) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < nrows) v = *(const f32x4*)(g + (long)row * ld + 16 * lg + 4 * i);
    f[4 * i] = v[0]; f[4 * i + 1] = v[1]; f[4 * i + 2] = v[2]; f[4 * i + 3] = v[3];
  }
}
__device__ __forceinline__ float t32_elem(const char* img, int r, int col) {
  return *(const float*)(img + t32_off(r, col >> 2) + (col & 3) * 4);
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(256) void attn32m_fwd_kernel(Attn32Args p) {
  resolve_drop(p.drop);
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 256];
  char* Kimg = smem;
  char* Vimg = smem + 64 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int T = p.T;
  const long ld = 3L * p.H * 64, ldc = (long)p.H * 64;
  const float* qb = p.qkv + (long)b * T * ld + h * 64;
  const float* kb_ = qb + p.H * 64;
  const float* vb = kb_ + p.H * 64;
  const int q0 = (blockIdx.x * 4 + wave) * 16, q = q0 + l15;
  float qf[16];
  load_own32(qf, qb, ld, q, T, lg);
  const float c = p.scale * LOG2E;
  float m = -INFINITY, l = 0.f;
  f32x4 ot[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ot[i] = zero4();
  const unsigned long long drow = (unsigned long long)bh * T + q;
  for (int kb = 0; kb < T; kb += 64) {
    if (kb) __syncthreads();
    load_tile32(Kimg, kb_, ld, kb, T, tid);
    load_tile32(Vimg, vb, ld, kb, T, tid);
    __syncthreads();
    f32x4 st[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kb + j * 16 < T) {
        float kf[16];
        frag32_rows(kf, Kimg, j * 16, l15, lg);
        f32x4 a = zero4();
#pragma unroll
        for (int s = 0; s < 16; ++s) a = MFMA32(kf[s], qf[s], a);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a[r] = (kb + j * 16 + lg * 4 + r < T) ? a[r] * c : -INFINITY;
          mx = fmaxf(mx, a[r]);
        }
        st[j] = a;
      } else {
        st[j] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
    }
    mx = grp4_max(mx);
    const float mn = fmaxf(m, mx);
    const float alpha = exp2f(m - mn);
    m = mn;
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[j][r] = exp2f(st[j][r] - mn);
        ls += st[j][r];
      }
    l = l * alpha + ls;
#pragma unroll
    for (int i = 0; i < 4; ++i) ot[i] *= alpha;
    if (p.drop.thr) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned key = kb + j * 16 + lg * 4;
        float k0, k1, k2, k3;
        drop_pair(p.drop, drow, 0u, key, k0, k1);
        drop_pair(p.drop, drow, 0u, key + 2, k2, k3);
        st[j][0] *= k0; st[j][1] *= k1; st[j][2] *= k2; st[j][3] *= k3;
      }
    }
    // O^T[d][q] += sum over keys V^T[d][key] P^T[key][q]: MFMA step (j, r) has k-slot g = key 16 j + 4 g + r, whose
    // probability is this lane's register st[j][r]; the A operand is V[that key][dt * 16 + l15]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kb + j * 16 < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kr = j * 16 + lg * 4 + r;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) ot[dt] = MFMA32(t32_elem(Vimg, kr, dt * 16 + l15), st[j][r], ot[dt]);
        }
      }
    }
  }
  l = grp4_sum(l);
  if (q < T) {
    const float inv = 1.0f / l;
    float* o = p.ctx + ((long)b * T + q) * ldc + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *(f32x4*)(o + dt * 16 + lg * 4) = ot[dt] * inv;
    if (lg == 0 && p.lse) p.lse[(long)bh * T + q] = (m + log2f(l)) * LN2;
  }
}

__global__ __launch_bounds__(256) void attn32m_dq_kernel(Attn32Args p) {
  resolve_drop(p.drop);
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 256];
  char* Kimg = smem;
  char* Vimg = smem + 64 * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int T = p.T;
  const long ld = 3L * p.H * 64, ldc = (long)p.H * 64;
  const float* qb = p.qkv + (long)b * T * ld + h * 64;
  const float* kb_ = qb + p.H * 64;
  const float* vb = kb_ + p.H * 64;
  const float* dob = p.dctx + (long)b * T * ldc + h * 64;
  const float* ob = p.ctx + (long)b * T * ldc + h * 64;
  const int q0 = (blockIdx.x * 4 + wave) * 16, q = q0 + l15;
  float qf[16], dof[16];
  load_own32(qf, qb, ld, q, T, lg);
  load_own32(dof, dob, ldc, q, T, lg);
  const float c = p.scale * LOG2E;
  const float lse2 = q < T ? p.lse[(long)bh * T + q] * LOG2E : INFINITY;
  float del = 0.f;  // delta[q] = rowsum(dO o O): this lane's 16 columns, then the 4 lane groups
  {
    float of[16];
    load_own32(of, ob, ldc, q, T, lg);
#pragma unroll
    for (int s = 0; s < 16; ++s) del = fmaf(of[s], dof[s], del);
    del = grp4_sum(del);
    if (q < T && lg == 0) p.delta[(long)bh * T + q] = del;
  }
  f32x4 dqt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dqt[i] = zero4();
  const unsigned long long drow = (unsigned long long)bh * T + q;
  for (int kb = 0; kb < T; kb += 64) {
    if (kb) __syncthreads();
    load_tile32(Kimg, kb_, ld, kb, T, tid);
    load_tile32(Vimg, vb, ld, kb, T, tid);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (kb + j * 16 >= T) continue;
      float kf[16], vf[16];
      frag32_rows(kf, Kimg, j * 16, l15, lg);
      frag32_rows(vf, Vimg, j * 16, l15, lg);
      f32x4 s_ = zero4(), dp = zero4();
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        s_ = MFMA32(kf[s], qf[s], s_);
        dp = MFMA32(vf[s], dof[s], dp);
      }
      const unsigned key0 = kb + j * 16 + lg * 4;
      float k[4] = {1.f, 1.f, 1.f, 1.f};
      if (p.drop.thr) {
        drop_pair(p.drop, drow, 0u, key0, k[0], k[1]);
        drop_pair(p.drop, drow, 0u, key0 + 2, k[2], k[3]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = ((int)key0 + r < T) ? exp2f(s_[r] * c - lse2) : 0.f;
        const float ds = pr * (dp[r] * k[r] - del);
        const int kr = j * 16 + lg * 4 + r;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dqt[dt] = MFMA32(t32_elem(Kimg, kr, dt * 16 + l15), ds, dqt[dt]);
      }
    }
  }
  if (q < T) {
    float* o = p.dqkv + ((long)b * T + q) * ld + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *(f32x4*)(o + dt * 16 + lg * 4) = dqt[dt] * p.scale;
  }
}

__global__ __launch_bounds__(256) void attn32m_dkv_kernel(Attn32Args p) {
  resolve_drop(p.drop);
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 256 + 2 * 64 * 4];
  char* Qimg = smem;
  char* Oimg = smem + 64 * 256;
  float* lse_s = (float*)(smem + 2 * 64 * 256);
  float* del_s = lse_s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int T = p.T;
  const long ld = 3L * p.H * 64, ldc = (long)p.H * 64;
  const float* qb = p.qkv + (long)b * T * ld + h * 64;
  const float* kb_ = qb + p.H * 64;
  const float* vb = kb_ + p.H * 64;
  const float* dob = p.dctx + (long)b * T * ldc + h * 64;
  const int key = (blockIdx.x * 4 + wave) * 16 + l15;
  float kf[16], vf[16];
  load_own32(kf, kb_, ld, key, T, lg);
  load_own32(vf, vb, ld, key, T, lg);
  const float c = p.scale * LOG2E;
  f32x4 dkt[4], dvt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dkt[i] = dvt[i] = zero4();
  for (int qb0 = 0; qb0 < T; qb0 += 64) {
    if (qb0) __syncthreads();
    load_tile32(Qimg, qb, ld, qb0, T, tid);
    load_tile32(Oimg, dob, ldc, qb0, T, tid);
    if (tid < 64) {
      const int qq = qb0 + tid;
      lse_s[tid] = qq < T ? p.lse[(long)bh * T + qq] * LOG2E : INFINITY;
      del_s[tid] = qq < T ? p.delta[(long)bh * T + qq] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (qb0 + j * 16 >= T) continue;
      float qfr[16], ofr[16];
      frag32_rows(qfr, Qimg, j * 16, l15, lg);
      frag32_rows(ofr, Oimg, j * 16, l15, lg);
      f32x4 s_ = zero4(), dp = zero4();
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        s_ = MFMA32(qfr[s], kf[s], s_);
        dp = MFMA32(ofr[s], vf[s], dp);
      }
      const f32x4 l4 = *(const f32x4*)(lse_s + j * 16 + lg * 4);
      const f32x4 d4 = *(const f32x4*)(del_s + j * 16 + lg * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = exp2f(s_[r] * c - l4[r]);  // rows past T carry lse = +inf -> 0
        float mk = 1.f;
        if (p.drop.thr) {
          const unsigned long long row = (unsigned long long)bh * T + (qb0 + j * 16 + lg * 4 + r);
          const unsigned hsh = drop_bits(drop_rowkey(p.drop, row), (unsigned)key >> 1);
          const unsigned r16 = (key & 1) ? (hsh >> 16) : (hsh & 0xFFFFu);
          mk = r16 >= p.drop.thr ? p.drop.scale : 0.f;
        }
        const float pd = pr * mk, ds = pr * (dp[r] * mk - d4[r]);
        const int qr = j * 16 + lg * 4 + r;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dvt[dt] = MFMA32(t32_elem(Oimg, qr, dt * 16 + l15), pd, dvt[dt]);
          dkt[dt] = MFMA32(t32_elem(Qimg, qr, dt * 16 + l15), ds, dkt[dt]);
        }
      }
    }
  }
  if (key < T) {
    float* ok = p.dqkv + ((long)b * T + key) * ld + p.H * 64 + h * 64;
    float* ov = ok + p.H * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *(f32x4*)(ok + dt * 16 + lg * 4) = dkt[dt] * p.scale;
      *(f32x4*)(ov + dt * 16 + lg * 4) = dvt[dt];
    }
  }
}
#undef MFMA32

int g_attn32_mfma = 1;  // vit_set_option("attn32_mfma"): 0 = the one-wave-per-row fp32 kernels for every shape

static int launch_attn32(int which, Attn32Args& a, hipStream_t st) {
  VIT_CHECK(a.T <= 4096 && a.dh <= 128 && (a.dh % 4) == 0, VIT_ERR_UNSUPPORTED,
            "fp32 attention supports T <= 4096 and dh <= 128 (multiple of 4); got T=%d dh=%d", a.T, a.dh);
  if (g_attn32_mfma && a.dh == 64 && !a.probs) {  // head_dim 64, no attention-map output: the f32-MFMA kernels
    dim3 grid(cdiv(cdiv(a.T, 16), 4), a.B * a.H);
    if (which == 0) hipLaunchKernelGGL(attn32m_fwd_kernel, grid, dim3(256), 0, st, a);
    else if (which == 1) hipLaunchKernelGGL(attn32m_dq_kernel, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(attn32m_dkv_kernel, grid, dim3(256), 0, st, a);
    VIT_LAUNCH_CHECK();
    return VIT_OK;
  }
  a.Tp = (a.T + 63) & ~63;
  const size_t smem = (size_t)4 * (2 * a.Tp + 256) * sizeof(float);
  const long rows = (long)a.B * a.H * a.T;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  static bool attr[3] = {false, false, false};
  const void* fns[3] = {(const void*)attn32_row_kernel<0>, (const void*)attn32_row_kernel<1>, (const void*)attn32_dkv_kernel};
  if (!attr[which]) {
    VIT_HIP(hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr[which] = true;
  }
  if (which == 0) hipLaunchKernelGGL(attn32_row_kernel<0>, grid, block, smem, st, a);
  else if (which == 1) hipLaunchKernelGGL(attn32_row_kernel<1>, grid, block, smem, st, a);
  else hipLaunchKernelGGL(attn32_dkv_kernel, grid, block, smem, st, a);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

static int check_attn(const char* fn, int B, int H, int T, int dh, float p) {
  VIT_CHECK(B > 0 && H > 0 && T > 0 && dh > 0, VIT_ERR_ARG, "%s: B=%d H=%d T=%d dh=%d", fn, B, H, T, dh);
  VIT_CHECK((dh % 4) == 0 && dh <= 128, VIT_ERR_UNSUPPORTED, "%s: head dim %d (need a multiple of 4, <= 128)", fn, dh);
  VIT_CHECK(p >= 0.f && p < 1.f, VIT_ERR_ARG, "%s: dropout_p out of [0,1)", fn);
  return VIT_OK;
}

#define DISPATCH_DH(KERNEL, grid, st, a)                                                         \
  do {                                                                                           \
    if (a.dh <= 32) hipLaunchKernelGGL((KERNEL<32>), grid, dim3(AW * 64), 0, st, a);              \
    else if (a.dh <= 64) hipLaunchKernelGGL((KERNEL<64>), grid, dim3(AW * 64), 0, st, a);         \
    else hipLaunchKernelGGL((KERNEL<128>), grid, dim3(AW * 64), 0, st, a);                        \
  } while (0)

}  // namespace vit

extern "C" {
using namespace vit;

int vit_attention_fwd(vit_handle h, const void* qkv, void* ctx, float* lse, int io_dtype, int B, int H, int T, int dh,
                      float scale, float dropout_p, uint64_t seed, uint64_t site, vit_stream stream) {
  return vit_attention_fwd_lo(h, qkv, ctx, nullptr, lse, io_dtype, B, H, T, dh, scale, dropout_p, seed, site, stream);
}

int vit_attention_fwd_lo(vit_handle h, const void* qkv, void* ctx, void* ctx_lo, float* lse, int io_dtype, int B, int H,
                         int T, int dh, float scale, float dropout_p, uint64_t seed, uint64_t site, vit_stream stream) {
  VIT_CHECK(qkv && ctx && lse, VIT_ERR_ARG, "vit_attention_fwd: null pointer");
  int rc = check_attn("vit_attention_fwd", B, H, T, dh, dropout_p);
  if (rc != VIT_OK) return rc;
  if (io_dtype == VIT_F32) {
    Attn32Args a32 = {};
    a32.qkv = (const float*)qkv; a32.ctx = (float*)ctx; a32.lse = lse;
    a32.B = B; a32.H = H; a32.T = T; a32.dh = dh; a32.scale = scale;
    a32.drop = make_drop_h(h, dropout_p, seed, site);
    return launch_attn32(0, a32, (hipStream_t)stream);
  }
  AttnArgs a = {};
  a.qkv = (const short*)qkv; a.ctx = (short*)ctx; a.ctx_lo = (short*)ctx_lo; a.lse = lse;
  a.B = B; a.H = H; a.T = T; a.dh = dh; a.scale = scale;
  a.drop = make_drop_h(h, dropout_p, seed, site);
  if (T <= g_attn_res_max_t && T <= RES_MAX_T && dh <= RES_MAX_DH && res_fits(T, dh)) {
    const size_t img = 2 * (size_t)((T + 15) & ~15) * 2;  // K + V images: rows x dh_padded x 2 bytes each
    int ns_ = 0, wp_ = 0;
    res_geometry(T, &ns_, &wp_, g_attn_fwd_waves);
    if (a.dh == 64 && T > 192 && T <= 208 && H == 12 && ns_ == 2 && wp_ == 4)
      rc = launch_res<attn_fwd_res_kernel<64, RES_RQ, true, 208, 12, 2, 4>>(a, img * 64, (hipStream_t)stream, g_attn_fwd_waves);
    else if (a.dh == 64) rc = launch_res<attn_fwd_res_kernel<64, RES_RQ, true>>(a, img * 64, (hipStream_t)stream, g_attn_fwd_waves);
    else if (a.dh <= 32) rc = launch_res<attn_fwd_res_kernel<32, RES_RQ, false>>(a, img * 32, (hipStream_t)stream, g_attn_fwd_waves);
    else rc = launch_res<attn_fwd_res_kernel<64, RES_RQ, false>>(a, img * 64, (hipStream_t)stream, g_attn_fwd_waves);
    return rc;
  }
  dim3 grid(cdiv(cdiv(T, 16), AW), B * H);
  DISPATCH_DH(attn_fwd_kernel, grid, (hipStream_t)stream, a);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

static int attention_bwd_impl(vit_handle h, const void* qkv, const void* ctx, const void* ctx_lo, const void* dctx,
                              const float* lse, float* delta, void* dqkv, int io_dtype, int B, int H, int T, int dh,
                              float scale, float dropout_p, uint64_t seed, uint64_t site, float* colsum_part, vit_stream stream);

int vit_attention_bwd(vit_handle h, const void* qkv, const void* ctx, const void* dctx, const float* lse,
                      float* delta, void* dqkv, int io_dtype, int B, int H, int T, int dh, float scale,
                      float dropout_p, uint64_t seed, uint64_t site, vit_stream stream) {
  return attention_bwd_impl(h, qkv, ctx, nullptr, dctx, lse, delta, dqkv, io_dtype, B, H, T, dh, scale, dropout_p, seed, site,
                            nullptr, stream);
}

int vit_attention_bwd_colsum(vit_handle h, const void* qkv, const void* ctx, const void* dctx, const float* lse,
                             float* delta, void* dqkv, int io_dtype, int B, int H, int T, int dh, float scale,
                             float dropout_p, uint64_t seed, uint64_t site, float* dqkv_colsum, vit_stream stream) {
  VIT_CHECK(dqkv_colsum, VIT_ERR_ARG, "vit_attention_bwd_colsum: null dqkv_colsum");
  return vit_attention_bwd_lo(h, qkv, ctx, nullptr, dctx, lse, delta, dqkv, io_dtype, B, H, T, dh, scale, dropout_p, seed, site,
                              dqkv_colsum, stream);
}

int vit_attention_bwd_lo(vit_handle h, const void* qkv, const void* ctx, const void* ctx_lo, const void* dctx,
                         const float* lse, float* delta, void* dqkv, int io_dtype, int B, int H, int T, int dh, float scale,
                         float dropout_p, uint64_t seed, uint64_t site, float* dqkv_colsum, vit_stream stream) {
  if (!dqkv_colsum)
    return attention_bwd_impl(h, qkv, ctx, ctx_lo, dctx, lse, delta, dqkv, io_dtype, B, H, T, dh, scale, dropout_p, seed, site,
                              nullptr, stream);
  const int D3 = 3 * H * dh;
  if (io_dtype == VIT_BF16 && T <= g_attn_res_max_t && T <= RES_MAX_T && dh <= RES_MAX_DH && (dh % 4) == 0 && res_fits(T, dh)) {
    int nsplit, wpw;
    res_geometry(T, &nsplit, &wpw);
    size_t wsb = 0;
    float* part = (float*)ctx_workspace(h, &wsb);
    const bool pipe = g_attn_bwd_fused && pipe_fits(T, dh);  // partial rows per batch: one per wave of the pipelined kernel
    const int prow = pipe ? B * 8 : B * nsplit * wpw;
    if (part && wsb >= (size_t)prow * D3 * sizeof(float)) {
      // the resident kernels leave one partial row per wave: column sums of what they stored
      int rc = attention_bwd_impl(h, qkv, ctx, ctx_lo, dctx, lse, delta, dqkv, io_dtype, B, H, T, dh, scale, dropout_p, seed,
                                  site, part, stream);
      if (rc != VIT_OK) return rc;
      return launch_reduce_partials(part, prow, D3, dqkv_colsum, D3, dqkv_colsum, 0, (hipStream_t)stream);
    }
  }
  int rc = attention_bwd_impl(h, qkv, ctx, ctx_lo, dctx, lse, delta, dqkv, io_dtype, B, H, T, dh, scale, dropout_p, seed, site,
                              nullptr, stream);
  if (rc != VIT_OK) return rc;
  return vit_colsum(h, dqkv, io_dtype, D3, dqkv_colsum, B * T, D3, 0, stream);
}

static int attention_bwd_impl(vit_handle h, const void* qkv, const void* ctx, const void* ctx_lo, const void* dctx,
                              const float* lse, float* delta, void* dqkv, int io_dtype, int B, int H, int T, int dh,
                              float scale, float dropout_p, uint64_t seed, uint64_t site, float* colsum_part, vit_stream stream) {
  VIT_CHECK(qkv && ctx && dctx && lse && delta && dqkv, VIT_ERR_ARG, "vit_attention_bwd: null pointer");
  int rc = check_attn("vit_attention_bwd", B, H, T, dh, dropout_p);
  if (rc != VIT_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (io_dtype == VIT_F32) {
    Attn32Args a32 = {};
    a32.qkv = (const float*)qkv; a32.ctx = (float*)const_cast<void*>(ctx); a32.lse = const_cast<float*>(lse);
    a32.dctx = (const float*)dctx; a32.delta = delta; a32.dqkv = (float*)dqkv;
    a32.B = B; a32.H = H; a32.T = T; a32.dh = dh; a32.scale = scale;
    a32.drop = make_drop_h(h, dropout_p, seed, site);
    rc = launch_attn32(1, a32, st);
    if (rc != VIT_OK) return rc;
    return launch_attn32(2, a32, st);
  }
  AttnArgs a = {};
  a.qkv = (const short*)qkv; a.lse = const_cast<float*>(lse); a.ctx = (short*)const_cast<void*>(ctx);
  a.ctx_lo = (short*)const_cast<void*>(ctx_lo);
  a.dctx = (const short*)dctx; a.delta = delta; a.dqkv = (short*)dqkv;
  a.B = B; a.H = H; a.T = T; a.dh = dh; a.scale = scale;
  a.drop = make_drop_h(h, dropout_p, seed, site);
  a.csum_part = colsum_part;
  if (g_attn_bwd_fused && T <= g_attn_res_max_t && pipe_fits(T, dh)) {
    static bool attr = false;
    if (!attr) {
      VIT_HIP(hipFuncSetAttribute((const void*)attn_bwd_pipe_kernel<true, 12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      VIT_HIP(hipFuncSetAttribute((const void*)attn_bwd_pipe_kernel<true, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      VIT_HIP(hipFuncSetAttribute((const void*)attn_bwd_pipe_kernel<false, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = true;
    }
    const dim3 grid(std::min(B * H, ctx_num_cus(h)));
    if (T > 192 && H == 12 && a.ctx_lo)  // the ViT-B shape: everything the padded length and the head count determine is constant
      hipLaunchKernelGGL((attn_bwd_pipe_kernel<true, 12, true>), grid, dim3(512), pipe_smem(T), st, a);
    else if (T > 192) hipLaunchKernelGGL((attn_bwd_pipe_kernel<true, 0, false>), grid, dim3(512), pipe_smem(T), st, a);
    else hipLaunchKernelGGL((attn_bwd_pipe_kernel<false, 0, false>), grid, dim3(512), pipe_smem(T), st, a);
    VIT_LAUNCH_CHECK();
    return VIT_OK;
  }
  if (T <= g_attn_res_max_t && T <= RES_MAX_T && dh <= RES_MAX_DH && res_fits(T, dh)) {
    DISPATCH_RES_BWD(attn_bwd_dq_res_kernel, a, (2 * (size_t)((T + 15) & ~15) * DH_ * 2), st, rc);
    if (rc != VIT_OK) return rc;
    DISPATCH_RES_BWD(attn_bwd_dkv_res_kernel, a, (2 * (size_t)((T + 15) & ~15) * DH_ * 2 + 3 * (size_t)((T + 63) & ~63) * 4), st, rc);
    return rc;
  }
  dim3 grid(cdiv(cdiv(T, 16), AW), B * H);
  DISPATCH_DH(attn_bwd_dq_kernel, grid, st, a);
  VIT_LAUNCH_CHECK();
  DISPATCH_DH(attn_bwd_dkv_kernel, grid, st, a);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int vit_attention_probs(vit_handle h, const void* qkv, float* probs, int io_dtype, int B, int H, int T, int dh,
                        float scale, vit_stream stream) {
  VIT_CHECK(qkv && probs, VIT_ERR_ARG, "vit_attention_probs: null pointer");
  int rc = check_attn("vit_attention_probs", B, H, T, dh, 0.f);
  if (rc != VIT_OK) return rc;
  if (io_dtype == VIT_F32) {
    Attn32Args a32 = {};
    a32.qkv = (const float*)qkv; a32.probs = probs;
    a32.B = B; a32.H = H; a32.T = T; a32.dh = dh; a32.scale = scale;
    a32.drop = make_drop(0.f, 0, 0);
    return launch_attn32(0, a32, (hipStream_t)stream);
  }
  const long rows = (long)B * H * T;
  hipLaunchKernelGGL(attn_probs_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const short*)qkv, probs, B, H, T, dh, scale);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

}  // extern "C"
