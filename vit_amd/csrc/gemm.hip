// bf16 MFMA GEMM for gfx950 with fused epilogues: the dense contractions of the ViT step.
//
//   C[M,N] = epilogue(alpha * op(A)[M,K] * op(B)[K,N])
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// Operands reach LDS through registers (global -> VGPR issued one K-tile ahead, ds_write after the MFMAs of the
// current tile: the "issue early / write late" split), two LDS stages, one barrier per K-tile.
//
// Two LDS images, chosen per operand by how it is stored in HBM, so that forward (X*W^T), dX (dY*W) and
// dW (dY^T*X) all run from the tensors as they are, with no transposed copies:
//   K-contiguous  ([rows][K]):  image [128 rows][64 k], 128-B rows, 16-B chunk index XORed with (row>>1)&7;
//                               fragments by ds_read_b128 (conflict-free for the 16x16x32 operand map).
//   transposed    ([K][rows]):  image [64 k][128 rows], 256-B rows, 8-B chunk index XORed with
//                               ((k&3) | ((k>>3)&1)<<2) << 2; fragments by 2 x ds_read_b64_tr_b16 (hardware
//                               transpose; conflict-free: a 32-lane half covers all 64 banks exactly once).
// Both give fragment element j of lane l the k index 8*(l>>4)+j, so any A/B image pairing is consistent.
//
// The MFMA is issued with the operands swapped (B-fragment as matrix A), so a lane ends up with 4 CONSECUTIVE
// columns n of one row m: bias/aux/residual loads and the C store are 8/16-byte vectors.
#include <algorithm>
#include <stdio.h>

#include "common.h"

namespace vit {

constexpr int BM = 128, BN = 128, BK = 64, NTHR = 256;
constexpr int STAGE_BYTES = (BM * BK + BN * BK) * 2;  // 32 KiB
constexpr int A_BYTES = BM * BK * 2;

struct GemmArgs {
  const char* A; const char* B; char* C;
  long lda, ldb, ldc;       // elements
  long a_bs, b_bs, c_bs;    // batch strides, elements
  int M, N, K;
  int tiles_m, tiles_n;
  int splits, k_per_split;
  float* slab;              // split-K partials [batch*splits][M][N]
  const float* bias;
  const short* aux_in; short* aux_out; long ldaux;
  const float* residual; long ldres;
  float alpha;
  int act, c_dtype;
  DropCfg drop;
  int rpb, orb, roff;
};

__device__ __forceinline__ int tr_swz(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 2; }

// Epilogue shared by the bf16 and the split-bf16 ("x3") kernels: lane owns row m = .. + l15 and the 4 consecutive
// columns n = .. + lg*4 + {0..3} of each 16x16 accumulator tile.  AUX_F32: the saved pre-activation is f32 (f32 mode).
template <int AUX_F32>
__device__ __forceinline__ void generic_epilogue(f32x4 (&acc)[4][4], const GemmArgs& p, int m0, int n0, int wm, int wn,
                                                 int l15, int lg, int z, int batch) {
  if (p.splits > 1) {
    float* slab = p.slab + (long)z * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + lg * 4;
        if (n < p.N) *(f32x4*)(slab + (long)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
  char* Cb = p.C + (long)batch * p.c_bs * (p.c_dtype == VIT_BF16 ? 2 : 4);
  const unsigned half_cols = (unsigned)(p.N >> 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
    if (m >= p.M) continue;
    long orow = m;
    if (p.rpb > 0) {
      const int b = m / p.rpb;
      orow = (long)b * p.orb + (m - b * p.rpb) + p.roff;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + lg * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j] * p.alpha;
      if (p.bias) v += *(const f32x4*)(p.bias + n);
      if (p.act == VIT_ACT_GELU || p.act == VIT_ACT_GELU_GRAD) {
        f32x4 sv = v;  // what aux_out receives: the pre-activation, or gelu'(pre-activation)
        if (p.act == VIT_ACT_GELU_GRAD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float g_, d_;
            gelu_both(v[r], g_, d_);
            v[r] = g_;
            sv[r] = d_;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        }
        if (p.aux_out) {
          if (AUX_F32) {
            *(f32x4*)((float*)p.aux_out + orow * p.ldaux + n) = sv;
          } else {
            u32x2 pk = {pack2bf(sv[0], sv[1]), pack2bf(sv[2], sv[3])};
            *(u32x2*)(p.aux_out + orow * p.ldaux + n) = pk;
          }
        }
      } else if (p.act == VIT_ACT_DGELU || p.act == VIT_ACT_MUL_AUX) {
        f32x4 u;
        if (AUX_F32) {
          u = *(const f32x4*)((const float*)p.aux_in + orow * p.ldaux + n);
        } else {
          const bf16x4 ub = *(const bf16x4*)(p.aux_in + orow * p.ldaux + n);
          u = (f32x4){bf2f(ub[0]), bf2f(ub[1]), bf2f(ub[2]), bf2f(ub[3])};
        }
        if (p.act == VIT_ACT_DGELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] = dgelu_erf(u[r]);
        }
        v *= u;
      }
      if (p.drop.thr) {
        float k0, k1, k2, k3;
        drop_pair(p.drop, (unsigned long long)orow, half_cols, (unsigned)n, k0, k1);
        drop_pair(p.drop, (unsigned long long)orow, half_cols, (unsigned)n + 2, k2, k3);
        v[0] *= k0; v[1] *= k1; v[2] *= k2; v[3] *= k3;
      }
      if (p.residual) v += *(const f32x4*)(p.residual + orow * p.ldres + n);
      if (p.c_dtype == VIT_BF16) {
        u32x2 pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *(u32x2*)(Cb + (orow * p.ldc + n) * 2) = pk;
      } else {
        *(f32x4*)(Cb + (orow * p.ldc + n) * 4) = v;
      }
    }
  }
}

template <int TRANS>
struct Loader {
  // per-thread addressing of one operand tile: 4 x 16-byte global loads and 4 x ds_write_b128
  unsigned voff[4];
  unsigned lds_off[4];
  bool col_ok[4];
  int kchunk[4];  // TRANS=0: k offset (elements) of the chunk inside the tile, for the K-tail predicate
  __device__ __forceinline__ void init(int tid, long ld, int rows_total_minus_r0, int kdummy) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int q = tid + NTHR * i;
      if (TRANS == 0) {
        int r = q >> 3, c = q & 7;
        voff[i] = (unsigned)(r * ld * 2 + c * 16);
        lds_off[i] = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
        kchunk[i] = c * 8;
        col_ok[i] = true;
      } else {
        int kr = q >> 4, c16 = q & 15;
        voff[i] = (unsigned)(kr * ld * 2 + c16 * 16);
        lds_off[i] = kr * 256 + (((2 * c16) ^ tr_swz(kr)) << 3);
        kchunk[i] = 0;
        col_ok[i] = (c16 * 8) < rows_total_minus_r0;
      }
    }
  }
};

template <int A_T, int B_T>
__global__ __launch_bounds__(NTHR, 2) void gemm_bf16_kernel(GemmArgs p) {
  resolve_drop(p.drop);
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- XCD-aware tile order: blocks b and b+8 share an XCD (its L2); give each XCD a contiguous run of logical
  // tiles, n fastest, so neighbours re-use the same A panel from that L2 (bijective form for any tile count).
  const int ntile = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = ntile >> 3, r = ntile & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int batch = z / p.splits, split = z - batch * p.splits;
  const int k_begin = split * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int nk = (k_end - k_begin + BK - 1) / BK;

  // ---- operand bases and buffer resources (hardware zero-fill past the last valid row)
  const char* Ab;
  const char* Bb;
  unsigned long long a_bytes, b_bytes;
  if (A_T == 0) {
    Ab = p.A + ((long)batch * p.a_bs + (long)m0 * p.lda + k_begin) * 2;
    a_bytes = (unsigned long long)(p.M - m0) * p.lda * 2;
  } else {
    Ab = p.A + ((long)batch * p.a_bs + (long)k_begin * p.lda + m0) * 2;
    a_bytes = (unsigned long long)(k_end - k_begin) * p.lda * 2;
  }
  if (B_T == 0) {
    Bb = p.B + ((long)batch * p.b_bs + (long)n0 * p.ldb + k_begin) * 2;
    b_bytes = (unsigned long long)(p.N - n0) * p.ldb * 2;
  } else {
    Bb = p.B + ((long)batch * p.b_bs + (long)k_begin * p.ldb + n0) * 2;
    b_bytes = (unsigned long long)(k_end - k_begin) * p.ldb * 2;
  }
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bb, b_bytes);

  Loader<A_T> la;
  Loader<B_T> lb;
  la.init(tid, p.lda, p.M - m0, 0);
  lb.init(tid, p.ldb, p.N - n0, 0);
  const unsigned a_step = (A_T == 0) ? BK * 2 : (unsigned)(BK * p.lda * 2);
  const unsigned b_step = (B_T == 0) ? BK * 2 : (unsigned)(BK * p.ldb * 2);
  const int klen = k_end - k_begin;

  i32x4 sa[4], sb[4];
  auto issue = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool ok = (A_T == 0) ? (kt * BK + la.kchunk[i] < klen) : la.col_ok[i];
      unsigned vo = ok ? la.voff[i] + (unsigned)kt * a_step : OOB;
      sa[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bool ok = (B_T == 0) ? (kt * BK + lb.kchunk[i] < klen) : lb.col_ok[i];
      unsigned vo = ok ? lb.voff[i] + (unsigned)kt * b_step : OOB;
      sb[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, vo, 0, 0);
    }
  };
  auto commit = [&](int stage) {
    char* As = smem + stage * STAGE_BYTES;
    char* Bs = As + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(i32x4*)(As + la.lds_off[i]) = sa[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *(i32x4*)(Bs + lb.lds_off[i]) = sb[i];
  };

  // ---- fragment read offsets (bytes inside an operand image), constant over the K loop
  // K-contiguous image: row = base16 + l15, chunk = s*4 + lg  ->  row*128 + ((chunk ^ (l15>>1)) << 4)
  const int kc_row = l15 * 128;
  const int kc_c0 = ((lg ^ (l15 >> 1)) << 4);          // s = 0
  const int kc_c1 = (((4 + lg) ^ (l15 >> 1)) << 4);    // s = 1
  // transposed image: lane (g=lg, q=l15>>2, p=l15&3) addresses k = s*32 + g*8 + q (+4), 4 elements at m = mb + 4p
  const int tq = l15 >> 2, tp = l15 & 3;
  const int tr_f = (tq | ((lg & 1) << 2)) << 2;
  const int tr_k0 = (lg * 8 + tq) * 256;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto read_frag = [&](const char* img, int trans, int base16, int s) -> bf16x8 {
    if (!trans) {
      const char* pa = img + (base16 * 128 + kc_row) + (s ? kc_c1 : kc_c0);
      return *(const bf16x8*)pa;
    } else {
      const int ch = (((base16 >> 2) + tp) ^ tr_f) << 3;
      const char* pa = img + tr_k0 + s * (32 * 256) + ch;
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)pa);
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(pa + 4 * 256));
      return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };

  if (nk > 0) {
    issue(0);
    commit(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1);
    const char* As = smem + cur * STAGE_BYTES;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag(As, A_T, wm * 64 + i * 16, s);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = read_frag(Bs, B_T, wn * 64 + j * 16, s);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) commit(cur ^ 1);
    __syncthreads();
  }

  generic_epilogue<0>(acc, p, m0, n0, wm, wn, l15, lg, z, batch);
}

// ------------------------------------------------------------------------------------------------ f32 operands ("x3")
// fp32-class GEMM on the bf16 matrix cores: every f32 operand element x is split on its way into the LDS into
// hi = bf16(x) and lo = bf16(x - hi) (two images per operand), and each fragment pair issues three MFMAs,
// acc += Ahi*Bhi + Ahi*Blo + Alo*Bhi  (the lo*lo term is below 2^-16 relative).  Operands keep ~16 mantissa bits and
// accumulation is fp32, so results agree with an fp32 GEMM to ~1e-5 relative: this is the arithmetic behind
// precision='32' (the reference's default, basemodule.py:233).  Tile 128x128x32, same images / fragment maps / epilogue
// as the bf16 kernel.
constexpr int XBK = 32;
constexpr int XIMG = 128 * XBK * 2;          // one bf16 image of one operand: 8 KiB
constexpr int XSTAGE = 4 * XIMG;             // A_hi, A_lo, B_hi, B_lo

// 8 consecutive f32 (two 16-byte vectors) -> one bf16x8 chunk of the "hi" image and one of the "lo" image
__device__ __forceinline__ void split8(const i32x4& v0, const i32x4& v1, i32x4& out_hi, i32x4& out_lo) {
  const f32x4 a = __builtin_bit_cast(f32x4, v0), b = __builtin_bit_cast(f32x4, v1);
  const unsigned h0 = pack2bf(a[0], a[1]), h1 = pack2bf(a[2], a[3]), h2 = pack2bf(b[0], b[1]), h3 = pack2bf(b[2], b[3]);
  const f32x4 ra = {a[0] - __builtin_bit_cast(float, h0 << 16), a[1] - __builtin_bit_cast(float, h0 & 0xFFFF0000u),
                    a[2] - __builtin_bit_cast(float, h1 << 16), a[3] - __builtin_bit_cast(float, h1 & 0xFFFF0000u)};
  const f32x4 rb = {b[0] - __builtin_bit_cast(float, h2 << 16), b[1] - __builtin_bit_cast(float, h2 & 0xFFFF0000u),
                    b[2] - __builtin_bit_cast(float, h3 << 16), b[3] - __builtin_bit_cast(float, h3 & 0xFFFF0000u)};
  out_hi = (i32x4){(int)h0, (int)h1, (int)h2, (int)h3};
  out_lo = (i32x4){(int)pack2bf(ra[0], ra[1]), (int)pack2bf(ra[2], ra[3]), (int)pack2bf(rb[0], rb[1]),
                   (int)pack2bf(rb[2], rb[3])};
}

template <int A_T, int B_T>
__global__ __launch_bounds__(NTHR, 2) void gemm_f32x3_kernel(GemmArgs p) {
  resolve_drop(p.drop);
  __shared__ __attribute__((aligned(16))) char smem[2 * XSTAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int batch = z / p.splits, split = z - batch * p.splits;
  const int k_begin = split * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int klen = k_end - k_begin;
  const int nk = (klen + XBK - 1) / XBK;

  const char* Ab;
  const char* Bb;
  unsigned long long a_bytes, b_bytes;
  if (A_T == 0) { Ab = p.A + ((long)batch * p.a_bs + (long)m0 * p.lda + k_begin) * 4; a_bytes = (unsigned long long)(p.M - m0) * p.lda * 4; }
  else { Ab = p.A + ((long)batch * p.a_bs + (long)k_begin * p.lda + m0) * 4; a_bytes = (unsigned long long)klen * p.lda * 4; }
  if (B_T == 0) { Bb = p.B + ((long)batch * p.b_bs + (long)n0 * p.ldb + k_begin) * 4; b_bytes = (unsigned long long)(p.N - n0) * p.ldb * 4; }
  else { Bb = p.B + ((long)batch * p.b_bs + (long)k_begin * p.ldb + n0) * 4; b_bytes = (unsigned long long)klen * p.ldb * 4; }
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bb, b_bytes);

  // per-thread: 2 bf16 chunks (8 elements = 32 bytes of f32) per operand per K-tile
  unsigned voffA[2], voffB[2], ldsA[2], ldsB[2];
  bool okA[2], okB[2];
  int kcA[2], kcB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = tid + NTHR * i;
    if (A_T == 0) { const int r = q >> 2, c = q & 3; voffA[i] = (unsigned)(r * p.lda * 4 + c * 32); ldsA[i] = r * 64 + ((c ^ ((-(r >> 2)) & 3)) << 4); kcA[i] = c * 8; okA[i] = true; }
    else { const int kr = q >> 4, c16 = q & 15; voffA[i] = (unsigned)(kr * p.lda * 4 + c16 * 32); ldsA[i] = kr * 256 + (((2 * c16) ^ tr_swz(kr)) << 3); kcA[i] = 0; okA[i] = (c16 * 8) < (p.M - m0); }
    if (B_T == 0) { const int r = q >> 2, c = q & 3; voffB[i] = (unsigned)(r * p.ldb * 4 + c * 32); ldsB[i] = r * 64 + ((c ^ ((-(r >> 2)) & 3)) << 4); kcB[i] = c * 8; okB[i] = true; }
    else { const int kr = q >> 4, c16 = q & 15; voffB[i] = (unsigned)(kr * p.ldb * 4 + c16 * 32); ldsB[i] = kr * 256 + (((2 * c16) ^ tr_swz(kr)) << 3); kcB[i] = 0; okB[i] = (c16 * 8) < (p.N - n0); }
  }
  const unsigned a_step = (A_T == 0) ? XBK * 4 : (unsigned)(XBK * p.lda * 4);
  const unsigned b_step = (B_T == 0) ? XBK * 4 : (unsigned)(XBK * p.ldb * 4);

  i32x4 sa[2][2], sb[2][2];
  auto issue = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = (A_T == 0) ? (kt * XBK + kcA[i] < klen) : okA[i];
      const unsigned vo = ok ? voffA[i] + (unsigned)kt * a_step : OOB;
      sa[i][0] = __builtin_amdgcn_raw_buffer_load_b128(ra, vo, 0, 0);
      sa[i][1] = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? vo + 16 : OOB, 0, 0);
      const bool okb = (B_T == 0) ? (kt * XBK + kcB[i] < klen) : okB[i];
      const unsigned vb = okb ? voffB[i] + (unsigned)kt * b_step : OOB;
      sb[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rb, vb, 0, 0);
      sb[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rb, okb ? vb + 16 : OOB, 0, 0);
    }
  };
  auto commit = [&](int stage) {
    char* st = smem + stage * XSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      i32x4 h, l;
      split8(sa[i][0], sa[i][1], h, l);
      *(i32x4*)(st + ldsA[i]) = h;
      *(i32x4*)(st + XIMG + ldsA[i]) = l;
      split8(sb[i][0], sb[i][1], h, l);
      *(i32x4*)(st + 2 * XIMG + ldsB[i]) = h;
      *(i32x4*)(st + 3 * XIMG + ldsB[i]) = l;
    }
  };

  const int kc_off = l15 * 64 + ((lg ^ ((-(l15 >> 2)) & 3)) << 4);
  const int tq = l15 >> 2, tp = l15 & 3;
  const int tr_f = (tq | ((lg & 1) << 2)) << 2;
  auto read_frag = [&](const char* img, int trans, int base16) -> bf16x8 {
    if (!trans) {
      return *(const bf16x8*)(img + base16 * 64 + kc_off);
    } else {
      const char* pa = img + (lg * 8 + tq) * 256 + ((((base16 >> 2) + tp) ^ tr_f) << 3);
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)pa);
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(pa + 4 * 256));
      return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    issue(0);
    commit(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1);
    const char* st = smem + cur * XSTAGE;
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = read_frag(st, A_T, wm * 64 + i * 16);
      al[i] = read_frag(st + XIMG, A_T, wm * 64 + i * 16);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 bh = read_frag(st + 2 * XIMG, B_T, wn * 64 + j * 16);
      const bf16x8 bl = read_frag(st + 3 * XIMG, B_T, wn * 64 + j * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[i], acc[i][j], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) commit(cur ^ 1);
    __syncthreads();
  }
  generic_epilogue<1>(acc, p, m0, n0, wm, wn, l15, lg, z, batch);
}

// C (f32) (+)= alpha * sum over split slabs; one float4 per thread
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, long ldc, int M, int N,
                                     int splits, float alpha, int accumulate) {
  const long nvec = (long)M * (N >> 2);
  const long stride = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (N >> 2);
    const int n = (int)(i - m * (N >> 2)) << 2;
    const float* s = slab + m * N + n;
    f32x4 a = *(const f32x4*)s;
    for (int k = 1; k < splits; ++k) a += *(const f32x4*)(s + k * stride);
    a *= alpha;
    float* c = C + m * ldc + n;
    if (accumulate) a += *(const f32x4*)c;
    *(f32x4*)c = a;
  }
}

void* ctx_workspace(vit_handle h, size_t* bytes);
int gemm2_try_launch(vit_handle h, const vit_gemm_desc* d, hipStream_t st, int* rc);  // gemm2.hip

int launch_splitk_reduce(const float* slab, float* C, long ldc, int M, int N, int splits, float alpha, int accumulate,
                         hipStream_t st) {
  const long nvec = (long)M * (N / 4);
  const int blocks = (int)std::min<long>((nvec + 255) / 256, 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, C, ldc, M, N, splits, alpha, accumulate);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

thread_local int g_colsum_fused = 0;  // set by a core whose epilogue produced d->colsum_out itself
thread_local int g_rope_fused = 0;    // set by a core whose epilogue applied d->rope_* itself
static int gemm_launch_core(vit_handle h, const vit_gemm_desc* d, hipStream_t st);

int gemm_launch(vit_handle h, const vit_gemm_desc* d, hipStream_t st) {
  g_colsum_fused = 0;
  g_rope_fused = 0;
  if (d && d->rope_cos) {
    VIT_CHECK(d->rope_sin && d->rope_T > 0 && d->rope_dh >= 8 && (d->rope_dh % 8) == 0 && d->rope_cols > 0 &&
                  (d->rope_cols % (2 * d->rope_dh)) == 0 && d->rope_cols <= d->N && d->rows_per_batch == 0 && !d->colsum_out &&
                  d->act == VIT_ACT_NONE && !d->residual && d->split_k <= 1 && d->dropout_p == 0.f,
              VIT_ERR_ARG, "vit_gemm: rope needs sin, T > 0, head_dim %% 8 == 0, rope_cols = 2 x heads x head_dim <= N, and a "
                           "projection (a bias is fine) without dropout, row map, column sums, residual or split-K: the pass "
                           "behind a dropped-out product would rotate AFTER the dropout, which is not the reference's order");
  }
  int rc = gemm_launch_core(h, d, st);
  if (rc == VIT_OK && d->rope_cos && !g_rope_fused)  // the core had no rotating epilogue for this shape: the separate pass
    rc = vit_rope_qk(h, d->C, d->c_dtype, d->rope_cos, d->rope_sin, d->M, d->rope_T, d->rope_cols / (2 * d->rope_dh), d->rope_dh,
                     d->ldc, 0, (vit_stream)st);
  if (rc != VIT_OK || !d->colsum_out || g_colsum_fused) return rc;
  VIT_CHECK(d->rows_per_batch == 0, VIT_ERR_ARG, "vit_gemm: colsum_out with a row map is not supported");
  return vit_colsum(h, d->C, d->c_dtype, d->ldc, d->colsum_out, d->M, d->N, 0, (vit_stream)st);
}

static int gemm_launch_core(vit_handle h, const vit_gemm_desc* d, hipStream_t st) {
  VIT_CHECK(d && d->A && d->B && d->C, VIT_ERR_ARG, "vit_gemm: null operand");
  VIT_CHECK(d->M > 0 && d->N > 0 && d->K > 0, VIT_ERR_ARG, "vit_gemm: empty problem M=%d N=%d K=%d", d->M, d->N, d->K);
  VIT_CHECK(d->ab_dtype == VIT_BF16 || d->ab_dtype == VIT_F32, VIT_ERR_ARG, "vit_gemm: bad ab_dtype");
  const bool f32in = d->ab_dtype == VIT_F32;
  VIT_CHECK(d->c_dtype == VIT_BF16 || d->c_dtype == VIT_F32, VIT_ERR_ARG, "vit_gemm: bad c_dtype");
  // 16-byte vectors along each operand's contiguous dimension: K for a K-contiguous operand, M / N for a transposed one
  // (whose K is a row index, bounded by the buffer resource, so any K works there: dW sums over B*T tokens).
  VIT_CHECK((d->lda % 8) == 0 && (d->ldb % 8) == 0 && (d->ldc % 4) == 0 && (d->N % 4) == 0, VIT_ERR_ARG,
            "vit_gemm: lda, ldb must be multiples of 8, ldc and N of 4 (N=%d lda=%ld ldb=%ld ldc=%ld)", d->N,
            (long)d->lda, (long)d->ldb, (long)d->ldc);
  VIT_CHECK(d->a_trans ? (d->M % 8) == 0 : (d->K % 8) == 0, VIT_ERR_ARG,
            "vit_gemm: A needs %s %% 8 == 0 (M=%d K=%d)", d->a_trans ? "M" : "K", d->M, d->K);
  VIT_CHECK(d->b_trans ? (d->N % 8) == 0 : (d->K % 8) == 0, VIT_ERR_ARG,
            "vit_gemm: B needs %s %% 8 == 0 (N=%d K=%d)", d->b_trans ? "N" : "K", d->N, d->K);
  VIT_CHECK(al16(d->A) && al16(d->B) && al16(d->C), VIT_ERR_ARG, "vit_gemm: operands must be 16-byte aligned");
  VIT_CHECK(d->lda >= (d->a_trans ? d->M : d->K) && d->ldb >= (d->b_trans ? d->N : d->K) && d->ldc >= d->N,
            VIT_ERR_ARG, "vit_gemm: leading dimension smaller than the row length");
  if (d->bias) VIT_CHECK(al16(d->bias), VIT_ERR_ARG, "vit_gemm: bias must be 16-byte aligned");
  if (d->residual) VIT_CHECK(al16(d->residual) && (d->ldres % 4) == 0, VIT_ERR_ARG, "vit_gemm: residual alignment");
  VIT_CHECK(d->act >= VIT_ACT_NONE && d->act <= VIT_ACT_MUL_AUX, VIT_ERR_ARG, "vit_gemm: bad act %d", d->act);
  if (d->act == VIT_ACT_DGELU || d->act == VIT_ACT_MUL_AUX)
    VIT_CHECK(d->aux_in && (d->ldaux % 4) == 0, VIT_ERR_ARG, "vit_gemm: ACT_DGELU / ACT_MUL_AUX need aux_in");
  if (d->act == VIT_ACT_GELU_GRAD) VIT_CHECK(d->aux_out, VIT_ERR_ARG, "vit_gemm: ACT_GELU_GRAD needs aux_out");
  if (d->aux_out) VIT_CHECK((d->ldaux % 4) == 0 && d->ldaux >= d->N, VIT_ERR_ARG, "vit_gemm: bad ldaux");
  VIT_CHECK(d->dropout_p >= 0.f && d->dropout_p < 1.f, VIT_ERR_ARG, "vit_gemm: dropout_p out of [0,1)");

  if (d->dropout_p > 0.f) VIT_CHECK((d->N % 2) == 0, VIT_ERR_ARG, "vit_gemm: dropout needs an even N");
  if (!f32in) {
    int rc2 = VIT_OK;  // tile-aligned problems go to the LDS-DMA / persistent core
    if (gemm2_try_launch(h, d, st, &rc2)) return rc2;
  }

  GemmArgs a;
  a.A = (const char*)d->A; a.B = (const char*)d->B; a.C = (char*)d->C;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.a_bs = a.b_bs = a.c_bs = 0;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.tiles_m = cdiv(d->M, BM); a.tiles_n = cdiv(d->N, BN);
  const int bk = f32in ? XBK : BK;
  const int ktiles = cdiv(d->K, bk);
  int splits = d->split_k;
  if (splits < 0) {
    const int tiles = a.tiles_m * a.tiles_n;
    splits = 1;
    if (tiles < 256) splits = std::min(std::max(1, 512 / tiles), std::max(1, ktiles / 4));
  }
  if (splits < 1) splits = 1;
  if (splits > ktiles) splits = ktiles;
  int kps = cdiv(ktiles, splits) * bk;
  splits = cdiv(d->K, kps);
  a.splits = splits; a.k_per_split = kps;
  a.slab = nullptr;
  if (splits > 1) {
    VIT_CHECK(d->c_dtype == VIT_F32 && !d->bias && d->act == VIT_ACT_NONE && d->dropout_p == 0.f && !d->residual &&
                  d->rows_per_batch == 0,
              VIT_ERR_ARG, "vit_gemm: split_k supports only alpha and an f32 C");
    size_t wsb = 0;
    void* ws = ctx_workspace(h, &wsb);
    size_t need = (size_t)splits * d->M * d->N * 4;
    VIT_CHECK(ws && wsb >= need, VIT_ERR_WORKSPACE, "vit_gemm: split-K needs %zu workspace bytes, have %zu", need, wsb);
    a.slab = (float*)ws;
  }
  a.bias = d->bias;
  a.aux_in = (const short*)d->aux_in; a.aux_out = (short*)d->aux_out; a.ldaux = d->ldaux;
  a.residual = d->residual; a.ldres = d->ldres;
  if (d->accumulate && splits == 1) {  // C += result through the residual port of the epilogue
    VIT_CHECK(d->c_dtype == VIT_F32 && !d->residual && d->rows_per_batch == 0, VIT_ERR_ARG,
              "vit_gemm: accumulate needs an f32 C and no residual");
    a.residual = (const float*)d->C; a.ldres = d->ldc;
  }
  a.alpha = d->alpha;
  a.act = d->act; a.c_dtype = d->c_dtype;
  a.drop = make_drop_h(h, d->dropout_p, d->seed, d->site);
  if (a.drop.thr) VIT_CHECK((d->N % 2) == 0, VIT_ERR_ARG, "vit_gemm: dropout needs an even N");
  a.rpb = d->rows_per_batch; a.orb = d->out_batch_rows; a.roff = d->out_row_offset;

  dim3 grid(a.tiles_m * a.tiles_n, splits), block(NTHR);
  const int v = d->a_trans * 2 + d->b_trans;
  snprintf(g_last_gemm, sizeof(g_last_gemm), "%s<%d, %d>", f32in ? "gemm_f32x3_kernel" : "gemm_bf16_kernel", d->a_trans,
           d->b_trans);
  if (f32in) {
    switch (v) {
      case 0: hipLaunchKernelGGL((gemm_f32x3_kernel<0, 0>), grid, block, 0, st, a); break;
      case 1: hipLaunchKernelGGL((gemm_f32x3_kernel<0, 1>), grid, block, 0, st, a); break;
      case 2: hipLaunchKernelGGL((gemm_f32x3_kernel<1, 0>), grid, block, 0, st, a); break;
      default: hipLaunchKernelGGL((gemm_f32x3_kernel<1, 1>), grid, block, 0, st, a); break;
    }
  } else
  switch (v) {
    case 0: hipLaunchKernelGGL((gemm_bf16_kernel<0, 0>), grid, block, 0, st, a); break;
    case 1: hipLaunchKernelGGL((gemm_bf16_kernel<0, 1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((gemm_bf16_kernel<1, 0>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((gemm_bf16_kernel<1, 1>), grid, block, 0, st, a); break;
  }
  VIT_LAUNCH_CHECK();
  if (splits > 1)
    return launch_splitk_reduce(a.slab, (float*)d->C, (long)d->ldc, d->M, d->N, splits, d->alpha, d->accumulate, st);
  return VIT_OK;
}

}  // namespace vit

extern "C" {

int vit_gemm(vit_handle h, const vit_gemm_desc* d, vit_stream stream) {
  return vit::gemm_launch(h, d, (hipStream_t)stream);
}

int vit_linear_fwd(vit_handle h, const void* x, const void* W, const float* bias, void* y, int y_dtype, int M, int N,
                   int K, int act, void* aux_out, float dropout_p, uint64_t seed, uint64_t site,
                   const float* residual, vit_stream stream) {
  vit_gemm_desc d = {};
  d.M = M; d.N = N; d.K = K; d.ab_dtype = VIT_BF16;
  d.A = x; d.lda = K; d.B = W; d.ldb = K; d.C = y; d.ldc = N; d.c_dtype = y_dtype;
  d.alpha = 1.f; d.bias = bias; d.act = act; d.aux_out = aux_out; d.ldaux = N;
  d.dropout_p = dropout_p; d.seed = seed; d.site = site; d.residual = residual; d.ldres = N;
  return vit::gemm_launch(h, &d, (hipStream_t)stream);
}

int vit_linear_bwd_dx(vit_handle h, const void* dy, const void* W, void* dx, int dx_dtype, int M, int N, int K,
                      const void* dgelu_aux_in, vit_stream stream) {
  vit_gemm_desc d = {};
  d.M = M; d.N = K; d.K = N; d.ab_dtype = VIT_BF16;   // dX[M,K] = dY[M,N] * W[N,K]: B stored [k'=N][n'=K]
  d.A = dy; d.lda = N; d.B = W; d.ldb = K; d.b_trans = 1; d.C = dx; d.ldc = K; d.c_dtype = dx_dtype;
  d.alpha = 1.f;
  if (dgelu_aux_in) { d.act = VIT_ACT_DGELU; d.aux_in = dgelu_aux_in; d.ldaux = K; }
  return vit::gemm_launch(h, &d, (hipStream_t)stream);
}

int vit_linear_bwd_dw(vit_handle h, const void* dy, const void* x, float* dW, int M, int N, int K, int accumulate,
                      vit_stream stream) {
  vit_gemm_desc d = {};
  d.M = N; d.N = K; d.K = M; d.ab_dtype = VIT_BF16;   // dW[N,K] = dY[M,N]^T * X[M,K]: both stored [k'=M][.]
  d.A = dy; d.lda = N; d.a_trans = 1; d.B = x; d.ldb = K; d.b_trans = 1; d.C = dW; d.ldc = K; d.c_dtype = VIT_F32;
  d.alpha = 1.f; d.split_k = -1; d.accumulate = accumulate;
  return vit::gemm_launch(h, &d, (hipStream_t)stream);
}

}  // extern "C"
