// Ping-pong bf16 MFMA GEMM core for gfx950, used when the problem is tile-aligned (the ViT-B / ViT-L shapes are: M = B*T
// padded to 256 rows, N multiples of 256, K multiples of 64).  Same contract, operand layouts, MFMA operand maps and fused
// epilogues as gemm.hip; what changes is how operands reach the LDS, how long they may be in flight, and how results leave
// the CU:
//
//   * 256x256x64 tiles, 8 waves = 512 threads, one workgroup per CU; the two wave halves run one barrier out of phase
//     (LOAD segment beside MFMA segment): gemm3_kernel below; gemm3h_kernel takes the tiles of a partial last round;
//   * LDS-DMA staging (global_load_lds_dwordx4 by inline asm, common.h: lds_dma16): 1 KiB per wave-instruction straight into
//     a ring of 8 half-tile slots, no staging registers.  The DMA destination is lane-linear, so the XOR swizzle of both
//     image kinds is applied to the per-lane SOURCE address and to the fragment reads (never to the destination);
//   * counted waits (s_waitcnt vmcnt(N), raw s_barrier): four half-tiles = 64 KiB stay in flight across barriers and across
//     output-tile boundaries;
//   * persistent over output tiles: a workgroup walks its tiles in one flat (tile, k-tile) iteration space, so the
//     loads of the next tile's first K-tiles are already in flight during the current tile's epilogue;
//   * XCD-aware tile order inside each round of concurrently running tiles (neighbours in n share the A panel in L2);
//   * epilogue through a per-wave LDS transpose (the 32 KiB the ring leaves free): accumulators go to LDS in MFMA layout and
//     come back row-major, so every bias / aux / residual load and every output store of a wave-instruction covers whole
//     128- or 256-byte row segments instead of 16 rows x 32 bytes.
// (The earlier lock-step geometries -- 256x256 / 256x128, 2-4 stage rings, selectable as gemm_core 2 / 3 / 4 / 6 -- were never
// chosen automatically once the ping-pong form existed and were removed in round 3.)
//
// LDS images (byte offsets inside one operand image of R rows):
//   K-contiguous, BK=64: row r, 16-B chunk c at r*128 + ((c ^ ((r>>1)&7)) << 4)
//   transposed:          k-row k, 8-B chunk ch at k*2R + ((ch ^ tr_swz(k)) << 3),  tr_swz(k) = ((k&3) | ((k>>3)&1)<<2) << 2
// all conflict-free for the 16x16x32 operand reads (ds_read_b128 / ds_read_b64_tr_b16): SQ_LDS_BANK_CONFLICT = 0.
#include <algorithm>
#include <stdio.h>

#include "common.h"

namespace vit {

#define GLB_AS __attribute__((address_space(1)))

struct Gemm2Args {
  const char* A; const char* B; char* C;
  long lda, ldb, ldc;
  int M, N, K;
  int tiles_m, tiles_n, nblk;
  int splits, k_per_split;
  float* slab;
  const float* bias;
  const short* aux_in; short* aux_out; long ldaux;
  const float* residual; long ldres;
  float alpha;
  int act, c_dtype;
  DropCfg drop;
  int rpb, orb, roff;
#ifdef VIT_PP_DIAG
  int debug;  // diagnostic twin build only (python -m vit_amd.build --diag, tools/pp_diag.py): pieces of the K loop switched off
#endif
  int lin_split;  // ping-pong kernel, split-K with one tile per workgroup: 1-D grid of tiles x splits, XCD-contiguous
  float* colsum_part;  // ping-pong kernel, bf16 epilogues: [tiles_m * 2][N] per-wave-row column sums of C, or NULL
  int tile_limit;      // ping-pong kernel: walk only the first tile_limit tiles (0 = all); the half-tile kernel takes the rest
  int tail_first, tail_n;  // half-tile kernel: tiles [tail_first, tail_first + tail_n), two workgroups each
  int slab_tiles;  // split-K over the tail tiles: the slab is compact, [split][tile - tail_first][256][256] f32 (0: [split][M][N])
  int grp2;  // ping-pong kernel: XCDs 0-3 walk the lower half of the N-tiles, XCDs 4-7 the upper half (see tile_coords)
  const float* rope_cos; const float* rope_sin;  // epilogue 8: rotary embedding of columns [0, rope_cols) (vit_gemm_desc)
  int rope_T, rope_dh, rope_cols;
#ifdef VIT_PP_STAMP
  unsigned long long* stamps;  // diagnostic build: [8 waves][8] summed s_memtime deltas of workgroup `stamp_block`
  int stamp_block;
#endif
};

__device__ __forceinline__ int tr_swz2(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 2; }

// Epilogue of one wave's (WM*16) x (WN*16) accumulator tile through its private LDS scratch (see file header).
template <int WM, int WN, int CW, int EPI>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[WM][WN], char* scr, const Gemm2Args& p, int m0, int n0,
                                              int split, int lane) {
  const int l15 = lane & 15, lg = lane >> 4;
  constexpr int CPR = CW / 4;         // 16-byte chunks per scratch row
  constexpr int RPI = 64 / CPR;       // rows one wave-instruction covers on the way out
  const unsigned half_cols = (unsigned)(p.N >> 1);
  float* slab = p.splits > 1 ? p.slab + (long)split * p.M * p.N : nullptr;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int jc = 0; jc < (WN * 16) / CW; ++jc) {
      // accumulators (MFMA layout: row l15, columns 16*j + 4*lg..+3) -> scratch
#pragma unroll
      for (int jj = 0; jj < CW / 16; ++jj) {
        const int c16 = jj * 4 + lg;
        *(f32x4*)(scr + l15 * (CW * 4) + ((c16 ^ (l15 & (CPR - 1))) << 4)) = acc[i][jc * (CW / 16) + jj];
        acc[i][jc * (CW / 16) + jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      // scratch -> row-major: lane owns 4 consecutive columns of one row, CPR lanes cover a whole row segment
#pragma unroll
      for (int rr = 0; rr < 16 / RPI; ++rr) {
        const int row = rr * RPI + lane / CPR, c16 = lane % CPR;
        f32x4 v = *(const f32x4*)(scr + row * (CW * 4) + ((c16 ^ (row & (CPR - 1))) << 4));
        const int m = m0 + i * 16 + row, n = n0 + jc * CW + c16 * 4;
        if (slab) {
          *(f32x4*)(slab + (long)m * p.N + n) = v;
          continue;
        }
        long orow = m;
        if (p.rpb > 0) {
          const int b = m / p.rpb;
          orow = (long)b * p.orb + (m - b * p.rpb) + p.roff;
        }
        v *= p.alpha;
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if (EPI == 1) {
          f32x4 sv = v;  // aux_out: the pre-activation (ACT_GELU) or gelu' of it (ACT_GELU_GRAD)
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
            u32x2 pk = {pack2bf(sv[0], sv[1]), pack2bf(sv[2], sv[3])};
            *(u32x2*)(p.aux_out + orow * p.ldaux + n) = pk;
          }
        } else if (EPI == 2) {
          bf16x4 u = *(const bf16x4*)(p.aux_in + orow * p.ldaux + n);
          if (p.act == VIT_ACT_MUL_AUX) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= bf2f(u[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= dgelu_erf(bf2f(u[r]));
          }
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
          *(u32x2*)(p.C + (orow * p.ldc + n) * 2) = pk;
        } else {
          *(f32x4*)(p.C + (orow * p.ldc + n) * 4) = v;
        }
      }
    }
  }
}

// defined in gemm.hip / api.hip
int launch_splitk_reduce(const float* slab, float* C, long ldc, int M, int N, int splits, float alpha, int accumulate,
                         hipStream_t st);
void* ctx_workspace(vit_handle h, size_t* bytes);

// Specialised epilogues of the ping-pong kernel (FAST): what the five hot GEMM kinds of a ViT layer need, with every
// decision at compile time so that no load sits behind a runtime branch (hipcc waits vmcnt(0) after each such load,
// and, beside LDS-DMA in flight, after ANY load: 32 serialised HBM round trips per tile in the generic epilogue):
//   3 = (+bias) (+dropout) -> bf16      Y = X W^T of the QKV / attention-output / FC2 projections
//   4 = +bias, erf-GELU (pre-activation saved to aux_out when given) -> bf16        FC1
//   5 = * gelu'(aux_in) -> bf16         dX of FC2; the wave's aux rows are fetched in two batches of 8 loads, up front
//   6 = plain -> bf16                   dX = dY W
//   7 = plain -> f32 (C or split-K slab)  dW = dY^T X
//   8 = +bias, rotary embedding of the q / k columns -> bf16      the fused QKV projection under pos_encoding_type 'rope'
//       (src/models/vit_with_rope.py:58-60): a wave's 64 output columns are whole heads (head_dim 16 / 32 / 64), so the
//       rotation partner i + head_dim / 2 of a lane's 8 columns sits 1 / 2 / 4 lanes away in the row-major layout behind the
//       LDS transpose: one ds_bpermute per value, the f32 values rotated BEFORE the one rounding to bf16
// Same per-wave LDS transpose as tile_epilogue on the way in; on the way out a lane owns 16 bytes of output.
template <int FAST, int NI>
__device__ __forceinline__ void pp_epilogue(f32x4 (&acc)[NI][4], char* scr, const Gemm2Args& p, int m0, int n0, int split,
                                            int lane) {
  const int l15 = lane & 15, lg = lane >> 4;
  if (FAST == 7) {
    // f32 out: a lane owns 4 consecutive columns of rows rr*4 + lane/16 (16-byte stores, 256-byte row segments)
    float* cf = p.splits > 1 ? p.slab + (long)split * p.M * p.N : (float*)p.C;
    long ldc = p.splits > 1 ? (long)p.N : p.ldc;
    if (p.slab_tiles) {  // tail tiles only: one 256 x 256 block per (K-slice, tile)
      const int tl = (m0 >> 8) * p.tiles_n + (n0 >> 8) - p.tail_first;
      cf = p.slab + ((long)split * p.slab_tiles + tl) * 65536;
      ldc = 256;
      m0 &= 255;
      n0 &= 255;
    }
    const int n = n0 + l15 * 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *(f32x4*)(scr + l15 * 256 + (((j * 4 + lg) ^ l15) << 4)) = acc[i][j];
        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = rr * 4 + lg;
        *(f32x4*)(cf + (long)(m0 + i * 16 + row) * ldc + n) = *(const f32x4*)(scr + row * 256 + ((l15 ^ row) << 4));
      }
    }
    return;
  }
  // bf16 out: a lane owns 8 consecutive columns of rows rr*8 + lane/8, so every store is 16 bytes per lane and a
  // wave-instruction covers 8 whole 128-byte row segments (the epilogue is store-ISSUE bound: half as many, twice as wide)
  const int cg = lane & 7, rsub = lane >> 3;
  const int n = n0 + cg * 8;
  // The outputs leave through raw buffer stores with the sc1 bit: write-through, and the line is NOT kept in the XCD's L2
  // (MI355X_MICROARCH.md, "stores of each flavour").  An epilogue's output is a write-once stream of 77-620 MB per launch that
  // nobody in this launch reads again; kept in the 4 MiB L2 it evicts the operand panels the XCD's other workgroups are about
  // to re-read.  r04, same-box A/B against plain stores (tools/gemm_bench.py, two interleaved repetitions): FC1 + GELU (two
  // outputs, 620 MB) 284-286 -> 271-275 us, the K = 768 / N = 768 products 70 -> 66 and 62 -> 60.7 us, QKV 160.6 -> 159.
#ifndef VIT_EPI_PLAIN_STORES
  const __amdgpu_buffer_rsrc_t rc_ = make_rsrc(p.C, (unsigned long long)p.M * p.ldc * 2);
  const __amdgpu_buffer_rsrc_t ra_ = make_rsrc(p.aux_out ? (const void*)p.aux_out : (const void*)p.C, (unsigned long long)p.M * p.ldaux * 2);
#define EPI_STORE_C(M_, PK) __builtin_amdgcn_raw_buffer_store_b128(PK, rc_, (unsigned)(((M_) * p.ldc + n) * 2), 0, 16)
#define EPI_STORE_AUX(M_, PK) __builtin_amdgcn_raw_buffer_store_b128(PK, ra_, (unsigned)(((M_) * p.ldaux + n) * 2), 0, 16)
#else  // A/B variant build only
#define EPI_STORE_C(M_, PK) *(u32x4*)(p.C + ((M_) * p.ldc + n) * 2) = PK
#define EPI_STORE_AUX(M_, PK) *(u32x4*)(p.aux_out + (M_) * p.ldaux + n) = PK
#endif
  const unsigned half_cols = (unsigned)(p.N >> 1);
  f32x4 bv0 = (f32x4){0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
  if ((FAST == 3 || FAST == 4 || FAST == 8) && p.bias) {
    bv0 = *(const f32x4*)(p.bias + n);
    bv1 = *(const f32x4*)(p.bias + n + 4);
  }
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // column sums of the stored (bf16-rounded) values
  const bool want_cs = p.colsum_part != nullptr;
  u32x4 au[4][2];  // FAST == 5: the aux rows of four 16-row blocks at a time (two batches per tile; one batch spills)
  // FAST == 8: cos / sin of this lane's 8 frequencies at its current token (rc*, rs*), advanced by a rotation of 8 tokens per row
  // chunk (k8*: table row 8) instead of 64 table loads per tile, each of which would be waited for in the open (first form,
  // r04: QKV 167 -> 217 us, exactly what the separate pass had cost)
  f32x4 rc0, rc1, rs0, rs1, k8c0, k8c1, k8s0, k8s1;
  int rt = 0;
  const int rope_peer = (lane ^ (p.rope_dh >> 4)) << 2;  // ds_bpermute byte address of the lane that holds column i +- dh / 2
  if (FAST == 8 && n0 < p.rope_cols) {
    const int hl_ = p.rope_dh >> 4, hd = p.rope_dh >> 1, io = (cg & (hl_ - 1)) * 8;
    rt = (m0 + rsub) % p.rope_T;
    const long to = (long)rt * hd + io, t8 = 8L * hd + io;
    rc0 = *(const f32x4*)(p.rope_cos + to); rc1 = *(const f32x4*)(p.rope_cos + to + 4);
    rs0 = *(const f32x4*)(p.rope_sin + to); rs1 = *(const f32x4*)(p.rope_sin + to + 4);
    k8c0 = *(const f32x4*)(p.rope_cos + t8); k8c1 = *(const f32x4*)(p.rope_cos + t8 + 4);
    k8s0 = *(const f32x4*)(p.rope_sin + t8); k8s1 = *(const f32x4*)(p.rope_sin + t8 + 4);
  }
  // the aux operand (gelu' of FC1's pre-activation, 310 MB) is read exactly once: nt + sc1 keeps it out of L1 / low priority in
  // L2 (r04 A/B on dX * gelu', same box: plain 338 us, nt 333, nt + sc1 327-330).  0 = plain loads (A/B variant)
#ifndef VIT_EPI_AUX_POLICY
#define VIT_EPI_AUX_POLICY 18
#endif
#if VIT_EPI_AUX_POLICY
  const __amdgpu_buffer_rsrc_t rx_ = make_rsrc(p.aux_in ? (const void*)p.aux_in : (const void*)p.C, (unsigned long long)p.M * p.ldaux * 2);
#endif
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (FAST == 5 && (i & 3) == 0) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#if VIT_EPI_AUX_POLICY
          au[ii][rr] = __builtin_amdgcn_raw_buffer_load_b128(rx_, (unsigned)(((long)(m0 + (i + ii) * 16 + rr * 8 + rsub) * p.ldaux + n) * 2), 0, VIT_EPI_AUX_POLICY);
#else
          au[ii][rr] = *(const u32x4*)(p.aux_in + (long)(m0 + (i + ii) * 16 + rr * 8 + rsub) * p.ldaux + n);
#endif
    }
    f32x4 v[2][2];
#ifdef VIT_EPI_NOLDS  // timing experiment (compile-time variant): no LDS transpose, values land in the wrong places
    {
      v[0][0] = acc[i][0]; v[0][1] = acc[i][1]; v[1][0] = acc[i][2]; v[1][1] = acc[i][3];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (false)
#endif
    {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *(f32x4*)(scr + l15 * 256 + (((j * 4 + lg) ^ l15) << 4)) = acc[i][j];
      acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = rr * 8 + rsub;
      v[rr][0] = *(const f32x4*)(scr + row * 256 + (((2 * cg) ^ row) << 4));
      v[rr][1] = *(const f32x4*)(scr + row * 256 + (((2 * cg + 1) ^ row) << 4));
    }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const long m = m0 + i * 16 + rr * 8 + rsub;
      float o[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = v[rr][0][r] + bv0[r];
        o[4 + r] = v[rr][1][r] + bv1[r];
      }
      if (FAST == 4) {
        float sv[8];  // aux_out: the pre-activation (ACT_GELU) or gelu' of it (ACT_GELU_GRAD: two FMAs on top of the GELU)
#ifdef VIT_EPI_NOGELU  // timing experiment: no GELU arithmetic (values are wrong)
        if (true) {
#pragma unroll
          for (int r = 0; r < 8; ++r) sv[r] = o[r] * 0.5f;
        } else
#endif
        if (p.act == VIT_ACT_GELU_GRAD) {
#pragma unroll
          for (int r = 0; r < 8; r += 2) {
            f32x2 g_, d_;
            gelu_both2((f32x2){o[r], o[r + 1]}, g_, d_);
            o[r] = g_[0]; o[r + 1] = g_[1];
            sv[r] = d_[0]; sv[r + 1] = d_[1];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 8; r += 2) {
            sv[r] = o[r]; sv[r + 1] = o[r + 1];
            const f32x2 g_ = gelu_erf2((f32x2){o[r], o[r + 1]});
            o[r] = g_[0]; o[r + 1] = g_[1];
          }
        }
#ifdef VIT_EPI_NOAUX  // timing experiment: the second output is not stored
        if (p.aux_out && sv[0] == 1.2345e30f) {
#else
        if (p.aux_out) {
#endif
          u32x4 pk = {pack2bf(sv[0], sv[1]), pack2bf(sv[2], sv[3]), pack2bf(sv[4], sv[5]), pack2bf(sv[6], sv[7])};
          EPI_STORE_AUX(m, pk);
        }
      }
      if (FAST == 5) {
        const u32x4 u = au[i & 3][rr];
        if (p.act == VIT_ACT_MUL_AUX) {  // aux_in already holds gelu'(.)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[2 * r] *= __builtin_bit_cast(float, u[r] << 16);
            o[2 * r + 1] *= __builtin_bit_cast(float, u[r] & 0xFFFF0000u);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[2 * r] *= dgelu_erf(__builtin_bit_cast(float, u[r] << 16));
            o[2 * r + 1] *= dgelu_erf(__builtin_bit_cast(float, u[r] & 0xFFFF0000u));
          }
        }
      }
      if (FAST == 8 && n0 < p.rope_cols) {  // wave-uniform: this wave's 64 columns are q or k heads
        const int hl = p.rope_dh >> 4;      // lanes per half head: 1, 2 or 4
        const float sg = (cg & hl) ? 1.f : -1.f;  // first half: x1 cos - x2 sin; second half: x2 cos + x1 sin
        const f32x4 c0 = rc0, c1 = rc1, s0 = rs0, s1 = rs1;  // this row chunk's angles (the recurrence below)
        float px[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)  // the partner lane's value: lane ^ hl inside the 8-lane row segment (LDS crossbar, no memory)
          px[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rope_peer, __builtin_bit_cast(int, o[r])));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = fmaf(o[r], c0[r], sg * px[r] * s0[r]);
          o[4 + r] = fmaf(o[4 + r], c1[r], sg * px[4 + r] * s1[r]);
        }
        // next row chunk of this lane = 8 tokens on: rotate the angles by 8 theta (table row 8); a lane whose token index
        // wrapped into the next sample reads its row again -- every lane does, behind a WAVE-uniform test, so the loads sit in
        // one rarely taken block (a 128-row wave tile crosses at most one sample boundary per lane when T >= 128)
        rt += 8;
        const bool wrapped = rt >= p.rope_T;
        if (wrapped) rt -= p.rope_T;
        {
          const f32x4 nc0 = rc0 * k8c0 - rs0 * k8s0, ns0 = rs0 * k8c0 + rc0 * k8s0;
          const f32x4 nc1 = rc1 * k8c1 - rs1 * k8s1, ns1 = rs1 * k8c1 + rc1 * k8s1;
          rc0 = nc0; rs0 = ns0; rc1 = nc1; rs1 = ns1;
        }
        if (__builtin_amdgcn_ballot_w64(wrapped)) {
          const long to = (long)rt * (p.rope_dh >> 1) + (cg & (hl - 1)) * 8;
          rc0 = *(const f32x4*)(p.rope_cos + to); rc1 = *(const f32x4*)(p.rope_cos + to + 4);
          rs0 = *(const f32x4*)(p.rope_sin + to); rs1 = *(const f32x4*)(p.rope_sin + to + 4);
        }
      }
      if (FAST == 3 && p.drop.thr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float k0, k1;
          drop_pair(p.drop, (unsigned long long)m, half_cols, (unsigned)n + 2 * r, k0, k1);
          o[2 * r] *= k0;
          o[2 * r + 1] *= k1;
        }
      }
      u32x4 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
      EPI_STORE_C(m, pk);
      if (want_cs) {  // uniform: only the launches that fuse a bias gradient pay the unpack + add per stored vector
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cs[2 * r] += __builtin_bit_cast(float, pk[r] << 16);
          cs[2 * r + 1] += __builtin_bit_cast(float, pk[r] & 0xFFFF0000u);
        }
      }
    }
  }
  if (p.colsum_part) {
    // the 8 lanes with the same lane & 7 hold the same 8 columns over different rows: fold them, lanes 0-7 store the
    // wave's 128-row sums; the two wave rows of a tile write separate partial rows (fixed order -> deterministic)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float v = cs[r];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      cs[r] = v;
    }
    if (lane < 8) {
      float* dst = p.colsum_part + (long)(m0 >> 7) * p.N + n;
      *(f32x4*)dst = (f32x4){cs[0], cs[1], cs[2], cs[3]};
      *(f32x4*)(dst + 4) = (f32x4){cs[4], cs[5], cs[6], cs[7]};
    }
  }
#undef EPI_STORE_C
#undef EPI_STORE_AUX
}

// ------------------------------------------------------------------------------------------------ ping-pong variant
// 256x256x64 tiles, 8 waves as 2 (M) x 4 (N), the "8-phase" structure: the two wave groups (waves 0-3 / 4-7: partners on
// the same four SIMDs) run ONE BARRIER out of phase, so that in every barrier interval one group is in a LOAD segment
// (its LDS fragment reads + its share of one half-tile's LDS-DMA) while the other is in an MFMA segment (16 MFMAs = one
// 64x32 quadrant of its 128x64 output over the whole K-tile, at s_setprio 1).  The matrix pipe of every SIMD always has a
// wave feeding it and fragment-read latency is never in front of an MFMA.
//
// LDS ring: 2 K-tiles x 4 HALF-tiles (A0, B0, B1, A1) of 16 KiB.  Half h of A holds rows wr*128 + h*64 .. +64 of both
// wave rows, half h of B the columns wc*64 + h*32 .. +32 of all four wave columns, so one phase needs one (A-half, B-half)
// pair and a half-tile is dead -- restageable -- long before the K-tile is finished.  Per K-tile `it`:
//   phase  MFMA quadrant   LDS reads (b128)        LDS-DMA issued (stream index p + 5)
//   P1     (A0, B0)        B0(it) 4, A0(it) 8      B1(it+1)
//   P2     (A0, B1)        B1(it) 4                A1(it+1)
//   P3     (A1, B1)        A1(it) 8                A0(it+2)
//   P4     (A1, B0)        -                       B0(it+2)
// Half-tiles stream in the order they are first read (A0, B0, B1, A1 per K-tile); stream index h is issued in phase h - 5
// (the first phase that is >= 2 phases after the last read of the slot it overwrites) and every phase ends its LOAD
// segment with s_waitcnt vmcnt(8): everything up to stream index p + 1 -- what phase p + 1 reads -- has landed, four
// half-tiles (64 KiB per CU) stay in flight across the barriers.  A wait in phase w orders reads in phase >= w + 1 for
// both groups (the staggered group reads one barrier later still).  The loop is flat over (output tile, K-tile), so the
// stream runs through tile boundaries; at a boundary the groups re-align (one extra barrier each, on opposite sides of
// the epilogue) so that all eight waves run their epilogues concurrently, and the first four phases after it count the
// epilogue's >= 32 stores per wave into the vmcnt budget instead of draining them.
template <int A_T, int B_T, int EPI, int NSLOT>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(Gemm2Args p) {
  resolve_drop(p.drop);
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int HALF = 128 * BK * 2;     // one half-tile image: 16 KiB
  // NSLOT half-tile slots, stream index h lives in slot h % NSLOT and is issued in phase h - DEPTH, DEPTH = NSLOT - 3 (the
  // first phase >= 2 after the last read of the slot it overwrites); each phase's wait leaves DEPTH - 1 half-tiles in flight.
  // NSLOT = 8: 64 KiB in flight, separate 32 KiB epilogue scratch.  NSLOT = 10: the ring takes all 160 KiB, 96 KiB in
  // flight (operands that miss L2 -- dW streams every byte from HBM / Infinity Cache -- are latency-bound: bytes in flight
  // over latency), and the epilogue scratch aliases the two slots that are free at a tile boundary, B1 and A1 of the
  // K-tile just finished (restaged in phases 1 and 2 of the next K-tile; waves 0-3 own the first, 4-7 the second, and a
  // wave half restages only after its own epilogue, the other half's DMA into the same slot comes a barrier later).
  constexpr int DEPTH = NSLOT - 3, INFL = 2 * (DEPTH - 1);
  constexpr int SCR = 4096, CW = 64;     // epilogue scratch per wave
  constexpr int XA0 = 0, XB0 = 1, XB1 = 2, XA1 = 3;
  // vmcnt budget of the four phases after an epilogue: the usual 8 + the FEWEST vector-memory operations that epilogue
  // issues per wave (its stores: 16 x 16 B for the bf16 kinds, 32 otherwise) -- a lower bound keeps the wait conservative
  constexpr int RELAX = INFL + (((EPI >= 3 && EPI <= 6) || EPI == 8) ? 16 : 32);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, grp = wr;
  const int l15 = lane & 15, lg = lane >> 4;

  const int ntile = p.tile_limit > 0 ? p.tile_limit : p.tiles_m * p.tiles_n;
  int split = blockIdx.y, bx = blockIdx.x;
  if (p.lin_split) {
    // split-K (dW = dY^T X: K = all tokens, a few dozen output tiles): one (tile, K-slice) per workgroup on a 1-D grid.
    // Workgroups are dealt to the 8 XCDs round-robin by linear id, so give each XCD a CONTIGUOUS run of the
    // (slice-major) work list: the ~32 tiles an XCD runs then belong to one or two K-slices and fetch that slice's
    // tiles_m + tiles_n operand panels once into its L2 instead of once per tile (2 x 32 panels).
    const int total = ntile * p.splits, lin = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = lin & 7, within = lin >> 3;
    const int pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    split = pos / ntile;
    bx = pos - split * ntile;
  }
  const int k_begin = split * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int nk = (k_end - k_begin) / BK;
  const int nblk = p.nblk;
  const int my_tiles = (ntile - bx + nblk - 1) / nblk;
  const int T = my_tiles * nk;  // K-tiles this workgroup walks

  // per-thread DMA source offsets (elements) of the two 8-KiB rounds of a half-tile, relative to (tile origin, k0, half 0)
  int offA[2], offB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (A_T == 0) {
      const int rimg = i * 64 + wave * 8 + (lane >> 3);
      offA[i] = (i * 128 + wave * 8 + (lane >> 3)) * (int)p.lda + (((lane & 7) ^ ((rimg >> 1) & 7)) << 3);
    } else {
      const int k = i * 32 + wave * 4 + (lane >> 4);
      const int c16 = (lane & 15) ^ (tr_swz2(k) >> 1);
      offA[i] = k * (int)p.lda + (c16 >> 3) * 128 + ((c16 & 7) << 3);
    }
    if (B_T == 0) {
      const int rimg = i * 64 + wave * 8 + (lane >> 3);
      offB[i] = ((rimg >> 5) * 64 + (rimg & 31)) * (int)p.ldb + (((lane & 7) ^ ((rimg >> 1) & 7)) << 3);
    } else {
      const int k = i * 32 + wave * 4 + (lane >> 4);
      const int c16 = (lane & 15) ^ (tr_swz2(k) >> 1);
      offB[i] = k * (int)p.ldb + (c16 >> 2) * 64 + ((c16 & 3) << 3);
    }
  }
  const long halfA = (A_T == 0) ? 64 * p.lda : 64, halfB = (B_T == 0) ? 32 * p.ldb : 32;  // elements to half 1

  auto tile_coords = [&](int j, int& tm, int& tn) {
    const int round0 = j * nblk;
    const int n_here = min(nblk, ntile - round0);
    const int q = n_here >> 3, r = n_here & 7, xcd = bx & 7, within = bx >> 3;
    const int pos = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    if (p.grp2) {
      // N = 3072 (12 tiles): the B operand (4.7 MB of weights) does not fit an XCD's 4 MiB L2, and with every XCD walking
      // all 12 N-tiles each round it was re-fetched per XCD per round (PMC: 482 MB read against an 82 MB operand set).
      // Split the N-tiles in two halves of 2.4 MB: the first half of a round's positions (XCDs 0-3) keeps walking the lower
      // half, the second (XCDs 4-7) the upper half, M-panel by M-panel, so each XCD's weights stay L2-resident across
      // rounds and an A panel is fetched by two XCD groups instead of ~1.3 -- 393 KB more per panel against 4.7 MB less per
      // XCD-round.  nblk and the tile count are even (host), so each round splits exactly and the map is a bijection.
      const int hgrp = n_here >> 1, hn = p.tiles_n >> 1;
      const int g = pos >= hgrp;
      const int u = (round0 >> 1) + (g ? pos - hgrp : pos);
      tm = u / hn;
      tn = g * hn + (u - tm * hn);
      return;
    }
    const int t = p.lin_split ? p.tail_first + bx : round0 + pos;  // tail_first: 0 unless this launch is the tail's K-slices
    tm = t / p.tiles_n;
    tn = t - tm * p.tiles_n;
  };
  // The K-tile stream cursor: global bases (A, B) of the next K-tile to stage.  Everything on this path is SCALAR code that
  // sits in a LOAD segment, in front of the barrier the other wave group's MFMA segment ends on, and a wave issues one
  // instruction per ~4-5 cycles: in-kernel stamps (tools/pp_stamps.py) showed the per-K-tile bookkeeping -- bases recomputed
  // with 64-bit multiplies, ring-slot arithmetic, three branches per phase -- costing as much as the 16 MFMAs it hides
  // behind.  So: inside an output tile the bases advance by a constant (two 64-bit adds); past the end of the walk the
  // cursor stays on the last K-tile (its re-fetches land in ring slots nobody reads again: no "is there a next K-tile"
  // branch in any phase); the ring-slot offsets are one XOR / one masked add per K-tile or phase.
  const long dA = (A_T == 0) ? (long)BK * 2 : (long)BK * p.lda * 2;  // bytes from one K-tile to the next
  const long dB = (B_T == 0) ? (long)BK * 2 : (long)BK * p.ldb * 2;
  int cj = 0, ckt = 0;
  const char *ca, *cb;
  auto tile_bases = [&](int j) {
    int tm, tn;
    tile_coords(j, tm, tn);
    ca = (A_T == 0) ? p.A + ((long)tm * BM * p.lda + k_begin) * 2 : p.A + ((long)k_begin * p.lda + (long)tm * BM) * 2;
    cb = (B_T == 0) ? p.B + ((long)tn * BN * p.ldb + k_begin) * 2 : p.B + ((long)k_begin * p.ldb + (long)tn * BN) * 2;
  };
  tile_bases(0);
  auto cursor_next = [&](const char*& ab, const char*& bb) {  // bases of the cursor's K-tile, then advance
    ab = ca;
    bb = cb;
    if (++ckt == nk) {
      if (cj + 1 < my_tiles) {
        ++cj;
        ckt = 0;
        tile_bases(cj);
      } else {
        ckt = nk - 1;  // end of the walk: stay
      }
    } else {
      ca += dA;
      cb += dB;
    }
  };
  // one half-tile = 2 LDS-DMA instructions per thread; destination: slot + round*8 KiB + wave*1 KiB (+ lane*16)
  static_assert(NSLOT == 8, "the ring arithmetic below is for 8 slots (two K-tiles of four half-tiles)");
  unsigned dofs = 0;  // byte offset of the slot the next stream index goes to (stream order: A0 B0 B1 A1 per K-tile)
  // uniform 64-bit base + the thread's constant 32-bit byte offset, raw LDS address: three instructions per piece (a generic
  // `char*` destination cost an aperture compare + null check per piece, the per-lane source a 64-bit VALU add; r03)
  const unsigned smem_a = lds_addr_of(smem) + wave * 1024;
  auto issue_half = [&](const char* base, long half_off, const int (&off)[2]) {
    const unsigned dst = smem_a + dofs;
    dofs = (dofs + HALF) & (NSLOT * HALF - 1);
#ifdef VIT_PP_DIAG  // diagnostic build only (tools/pp_diag.py): 16 = no operand DMA, 32 = no fragment reads, 64 = no MFMAs, 128 = no epilogue
    if (p.debug & 16) return;
#endif
    const char* sb = base + half_off * 2;
    lds_dma16_s(sb, (unsigned)off[0] * 2u, dst);
    lds_dma16_s(sb, (unsigned)off[1] * 2u, dst + 8192);
  };

  // ---- fragment reads
  const int tq = l15 >> 2, tp = l15 & 3;
  const int tr_f = (tq | ((lg & 1) << 2)) << 2;
  int kc_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) kc_off[s] = l15 * 128 + (((s * 4 + lg) ^ (l15 >> 1)) << 4);
  auto read_frag = [&](const char* img, int trans, int base16, int s) -> bf16x8 {
#ifdef VIT_PP_DIAG
    if (p.debug & 32) return (bf16x8){(short)base16, (short)s, 0, 0, 0, 0, 0, 0};
#endif
    if (!trans) {
      return *(const bf16x8*)(img + base16 * (BK * 2) + kc_off[s]);
    } else {
      const char* pa = img + (s * 32 + lg * 8 + tq) * 256 + ((((base16 >> 2) + tp) ^ tr_f) << 3);
      bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)pa);
      bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS bf16x4*)(pa + 4 * 256));
      return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], b0[2][2], b1[2][2];

// In-kernel stamps (diagnostic build -DVIT_PP_STAMP=1|2 only; tools/pp_stamps.py): where a barrier interval goes.
//   slot 0 fragment reads issued AND landed   1 LDS-DMA issue   2 counted vmcnt wait   (level 1: all three in slot 2)
//   slot 3 barrier that ends the LOAD segment + lgkmcnt   4 MFMA segment (issue)   5 closing barrier   6 epilogue + re-align
#ifdef VIT_PP_STAMP
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev_)::"memory");
#define PP_STX(K)                                                                                \
  {                                                                                              \
    unsigned long long t_;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    st_[K] += t_ - tprev_;                                                                       \
    tprev_ = t_;                                                                                 \
  }
#define PP_CNT st_[7] += 1;
#if VIT_PP_STAMP == 5
// level 5 (least intrusive): a stamp after every closing barrier (slot 1 = everything else) and ONE more at the end of P1's
// LOAD segment: slot 0 = loop tail + P1's fragment reads + its LDS-DMA issue + counted wait
#define PP_ST(K) PP_ST5_##K
#define PP_ST5_2
#define PP_ST5_3
#define PP_ST5_4
#define PP_ST5_5 PP_STX(1)
#define PP_ST5_6 PP_STX(1)
#define PP_ST2(K)
#define PP_ST3(PH)
#define PP_STL(PH) PP_STL5_##PH
#define PP_STL5_0 PP_STX(0)
#define PP_STL5_1
#define PP_STL5_2
#define PP_STL5_3
#elif VIT_PP_STAMP == 4
// level 4: slots 0-3 = the whole LOAD segment of phases P1..P4 (P1 includes the loop tail), 4 = barrier + lgkmcnt,
// 5 = MFMA segment, 6 = closing barrier + epilogue
#define PP_ST(K) PP_STX(((K) == 3 ? 4 : (K) == 4 ? 5 : 6))
#define PP_ST2(K)
#define PP_ST3(PH)
#define PP_STL(PH) PP_STX(PH)
#elif VIT_PP_STAMP == 3
// level 3: slots 0-3 = "fragment reads landed" of phases P1..P4, 4 = rest of the LOAD segment, 5 = barrier + lgkmcnt,
// 6 = MFMA segment + closing barrier + epilogue
#define PP_ST(K) PP_STX(((K) == 2 ? 4 : (K) == 3 ? 5 : 6))
#define PP_ST2(K)
#define PP_ST3(PH) PP_STX(PH)
#define PP_STL(PH) PP_ST(2)
#else
#define PP_ST(K) PP_STX(K)
#define PP_ST3(PH)
#define PP_STL(PH) PP_ST(2)
#if VIT_PP_STAMP >= 2
#define PP_ST2(K) PP_STX(K)
#else
#define PP_ST2(K)
#endif
#endif
#else
#define PP_ST(K)
#define PP_ST2(K)
#define PP_ST3(PH)
#define PP_STL(PH)
#define PP_CNT
#endif
#ifdef VIT_PP_NOPRIO
#define PP_SETPRIO1
#else
#define PP_SETPRIO1 __builtin_amdgcn_s_setprio(1);
#endif
#ifdef VIT_PP_DIAG
#define PP_DO_MFMA (!(p.debug & 64))
#else
#define PP_DO_MFMA true
#endif
#define PP_READ_A(IMG)                                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) \
      af[i_][s_] = read_frag(IMG, A_T, wr * 64 + i_ * 16, s_);
#define PP_READ_B(BF, IMG)                                                                       \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) \
      BF[j_][s_] = read_frag(IMG, B_T, wc * 32 + j_ * 16, s_);
// end of a LOAD segment -> MFMA segment on quadrant (HA, HB) -> closing barrier
#define PP_MFMA(HA, HB, BF)                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  __builtin_amdgcn_s_barrier();                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  PP_ST(3)                                                                                       \
  PP_SETPRIO1                                                                 \
  if (PP_DO_MFMA)                                                                                \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                            \
          acc[HA * 4 + i_][HB * 2 + j_] =                                                        \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j_][s_], af[i_][s_], acc[HA * 4 + i_][HB * 2 + j_], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  PP_ST(4)                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                  \
  PP_ST(5)                                                                                       \
  PP_CNT
// stage one half-tile and leave four half-tiles in flight: vmcnt(INFL), or vmcnt(RELAX) in the K-tile after an epilogue
// (`relaxed` != 0): its stores were issued after the three older half-tiles still in flight -- count them in instead of
// draining them.  One asm statement with its own two-instruction branch: an if / else in C++ made the compiler clone the
// K-tile body (256 VGPRs + spills).
#define PP_STAGE(PH, BASE, HOFF, OFF)                                                            \
  {                                                                                              \
  PP_ST2(0)                                                                                      \
  PP_ST3(PH)                                                                                     \
  issue_half(BASE, HOFF, OFF);                                                                   \
  PP_ST2(1)                                                                                      \
  asm volatile("s_cmp_lg_u32 %0, 0\n\t"                                                          \
               "s_cbranch_scc1 .Lpp_relaxed_%=\n\t"                                              \
               "s_waitcnt vmcnt(%1)\n\t"                                                         \
               "s_branch .Lpp_waited_%=\n"                                                       \
               ".Lpp_relaxed_%=:\n\t"                                                            \
               "s_waitcnt vmcnt(%2)\n"                                                           \
               ".Lpp_waited_%=:" ::"s"(relaxed), "n"(INFL), "n"(RELAX) : "scc", "memory");        \
  PP_STL(PH)                                                                                     \
  }
// Two forms of one K-tile, chosen per operand layout (MERGED below).
// FOUR phases of 16 MFMAs.  Phase x issues stream index 4 it + 6 + x: B1, A1 of K-tile it+1, then A0, B0 of K-tile it+2
#define PP_KTILE4                                                                                \
  PP_READ_B(b0, sp + XB0 * HALF)                                                                 \
  PP_READ_A(sp + XA0 * HALF)                                                                     \
  PP_STAGE(0, bb1, halfB, offB)                                                                  \
  PP_MFMA(0, 0, b0)                                                                              \
  PP_READ_B(b1, sp + XB1 * HALF)                                                                 \
  PP_STAGE(1, a1, halfA, offA)                                                                   \
  PP_MFMA(0, 1, b1)                                                                              \
  PP_READ_A(sp + XA1 * HALF)                                                                     \
  PP_STAGE(2, a2, 0, offA)                                                                       \
  PP_MFMA(1, 1, b1)                                                                              \
  PP_STAGE(3, bb2, 0, offB)                                                                      \
  PP_MFMA(1, 0, b0)
// TWO phases per K-tile, 32 MFMAs each (half the barriers):
//   I   reads B0, A0, B1 (16 fragments)  issues A1 of K-tile it+1           MFMA (A0,B0), (A0,B1)
//   II  reads A1 (8 fragments)           issues A0, B0, B1 of K-tile it+2   MFMA (A1,B1), (A1,B0)
// Stream order A0 B0 B1 A1 as before; every wait leaves four half-tiles in flight: after I the newest four are A0 B0 B1 A1
// of K-tile it+1, so A1(it) has landed (read in II); after II they are A1(it+1), A0 B0 B1(it+2), so A0 B0 B1 of it+1 have
// landed (read in the next I).  Slot reuse: the wait for a phase's fragment reads (lgkmcnt) sits BEFORE the barrier that
// ends its LOAD segment, so when that barrier releases the fragments are in registers; A0 B0 B1 of K-tile it are read in
// phase I by both groups (the staggered group's LOAD segment ends one barrier later, still before the other group's
// phase-II LOAD segment starts issuing their replacements), A1(it) in phase II, replaced from phase I of it+1.
#define PP_LOADED(RX)                                                                            \
  asm volatile("s_cmp_lg_u32 %0, 0\n\t"                                                          \
               "s_cbranch_scc1 .Lpp_relaxed_%=\n\t"                                              \
               "s_waitcnt vmcnt(%1)\n\t"                                                         \
               "s_branch .Lpp_waited_%=\n"                                                       \
               ".Lpp_relaxed_%=:\n\t"                                                            \
               "s_waitcnt vmcnt(%2)\n"                                                           \
               ".Lpp_waited_%=:\n\t"                                                             \
               "s_waitcnt lgkmcnt(0)" ::"s"(relaxed), "n"(INFL), "n"(RX) : "scc", "memory");       \
  PP_ST(2)
#define PP_MFMA2(HA, BFA, HBA, BFB, HBB)                                                         \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  __builtin_amdgcn_s_barrier();                                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  PP_ST(3)                                                                                       \
  PP_SETPRIO1                                                                                    \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                            \
          acc[HA * 4 + i_][HBA * 2 + j_] =                                                       \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(BFA[j_][s_], af[i_][s_], acc[HA * 4 + i_][HBA * 2 + j_], 0, 0, 0); \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                            \
          acc[HA * 4 + i_][HBB * 2 + j_] =                                                       \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(BFB[j_][s_], af[i_][s_], acc[HA * 4 + i_][HBB * 2 + j_], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  PP_ST(4)                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                  \
  PP_ST(5)                                                                                       \
  PP_CNT
#define PP_KTILE2                                                                                \
  PP_READ_B(b0, sp + XB0 * HALF)                                                                 \
  PP_READ_A(sp + XA0 * HALF)                                                                     \
  PP_READ_B(b1, sp + XB1 * HALF)                                                                 \
  issue_half(a1, halfA, offA);                                                                   \
  PP_LOADED(RELAX)                                                                               \
  PP_MFMA2(0, b0, 0, b1, 1)                                                                      \
  PP_READ_A(sp + XA1 * HALF)                                                                     \
  issue_half(a2, 0, offA);                                                                       \
  issue_half(bb2, 0, offB);                                                                      \
  issue_half(bb2, halfB, offB);                                                                  \
  PP_LOADED(RELAX)                                                                               \
  PP_MFMA2(1, b1, 1, b0, 0)
  // Measured per layout (tools/gemm_bench.py, ViT-B layer shapes): the two-phase form is worth 11-17 % on dW (both
  // operands through the transposing reads: its 24-instruction LOAD segments were the long pole of a 16-MFMA interval) and
  // 1-2 % on dX; the forward GEMMs (K = 768: an epilogue every 12 K-tiles) are 0-4 % FASTER with four phases.
  constexpr bool MERGED = A_T || B_T;

  // ---- prologue: stream indices 0..5 (K-tile 0, and A0, B0 of K-tile 1)
  static_assert(DEPTH == 5, "stage schedule written out for the 8-slot ring");
  const char *a1, *bb1, *a2, *bb2;  // bases of K-tiles it+1, it+2 (past the end: the last K-tile again)
  {
    const char *a0, *bb0;
    cursor_next(a0, bb0);
    issue_half(a0, 0, offA);
    issue_half(bb0, 0, offB);
    issue_half(bb0, halfB, offB);
    issue_half(a0, halfA, offA);
    cursor_next(a1, bb1);
    issue_half(a1, 0, offA);
    issue_half(bb1, 0, offB);
    if constexpr (MERGED) issue_half(bb1, halfB, offB);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFL) : "memory");
    cursor_next(a2, bb2);
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();

  unsigned kb = 0;  // byte offset of the current K-tile's four slots: 0 or 4 * HALF
  int relaxed = 0;
  // tile loop around a K loop: the K loop's body is contiguous code with ONE backward branch per K-tile (with the epilogue
  // inside a flat loop every K-tile jumped over ~8 KB of epilogue code and back: two far taken branches on the P1 critical path)
  for (int jt = 0; jt < my_tiles; ++jt) {
    for (int kt = 0; kt < nk; ++kt) {
      const char* sp = smem + kb;
      if constexpr (MERGED) {
        PP_KTILE2
      } else {
        PP_KTILE4
      }
      relaxed = 0;
      a1 = a2;
      bb1 = bb2;
      cursor_next(a2, bb2);
      kb ^= 4 * HALF;
    }
    {
      int tm, tn;
      tile_coords(jt, tm, tn);
      if (grp == 0) __builtin_amdgcn_s_barrier();  // re-align: the other group finishes its last MFMA segment
      char* scr = smem + 8 * HALF + wave * SCR;
#if defined(VIT_EPI_NONE)  // timing experiment (compile-time variant): no epilogue at all
      if (true) {
        if (acc[0][0][0] == 1.2345f) p.C[0] = 1;  // keep the accumulators alive
      } else
#elif defined(VIT_PP_DIAG)
      if (p.debug & 128) {
        if (acc[0][0][0] == 1.2345f) p.C[0] = 1;  // keep the accumulators alive
      } else
#endif
      if constexpr (EPI >= 3) pp_epilogue<EPI, 8>(acc, scr, p, tm * BM + wr * 128, tn * BN + wc * 64, split, lane);
      else tile_epilogue<8, 4, CW, EPI>(acc, scr, p, tm * BM + wr * 128, tn * BN + wc * 64, split, lane);
      if (grp == 1) __builtin_amdgcn_s_barrier();  // re-stagger
      relaxed = 1;
      PP_ST(6)
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();  // balance the stagger barrier of the other group
#ifdef VIT_PP_STAMP
  if (p.stamps && (int)blockIdx.x == p.stamp_block && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) p.stamps[wave * 8 + k] = st_[k];
  }
#endif
#undef PP_ST
#undef PP_ST2
#undef PP_ST3
#undef PP_STL
#undef PP_STX
#undef PP_CNT
#undef PP_READ_A
#undef PP_READ_B
#undef PP_MFMA
#undef PP_STAGE
#undef PP_KTILE2
#undef PP_KTILE4
#undef PP_LOADED
#undef PP_MFMA2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-fetches past the end of the walk are still landing in the LDS
}

// ------------------------------------------------------------------------------------------------ half-tile tail kernel
// The tiles of a partial last round (591 tiles = 2 rounds of 256 + 79) would keep 79 workgroups busy for a whole tile-time
// while 177 CUs idle.  This kernel runs them as 2 x 79 HALF tiles instead: workgroup u takes tile tail_first + u/2 and, in
// each wave row, the 64-row half h = u & 1 -- every wave computes a 64 x 64 block, the workgroup 128 rows x 256 columns.
// Same ping-pong structure with TWO phases per K-tile, P1 = (A_h, B0), P2 = (A_h, B1), three half-tiles per K-tile in the
// stream order A B0 B1, each issued three phases before its first read (odd phases issue one half-tile, even phases
// two), 8-slot ring (slot = stream index % 8: restaged >= 2 phases after the last read), vmcnt(6) = three half-tiles in
// flight at every wait.  A separate code object on purpose: the main kernel's register allocation is untouched.
template <int A_T, int B_T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm3h_kernel(Gemm2Args p) {
  resolve_drop(p.drop);
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int HALF = 128 * BK * 2, NSLOT = 8;
  constexpr int SCR = 4096, CW = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3, grp = wr;
  const int l15 = lane & 15, lg = lane >> 4;
  // XCD-contiguous unit order: the two halves of a tile (same B half-tiles at the same time) land on one XCD
  const int total = 2 * p.tail_n, lin = blockIdx.x;
  const int q_ = total >> 3, r_ = total & 7, xcd = lin & 7, within = lin >> 3;
  const int unit = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + within;
  const int t = p.tail_first + (unit >> 1), h = unit & 1;
  const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const int nk = p.K / BK;

  int offA[2], offB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (A_T == 0) {
      const int rimg = i * 64 + wave * 8 + (lane >> 3);
      offA[i] = (i * 128 + wave * 8 + (lane >> 3)) * (int)p.lda + (((lane & 7) ^ ((rimg >> 1) & 7)) << 3);
    } else {
      const int k = i * 32 + wave * 4 + (lane >> 4);
      const int c16 = (lane & 15) ^ (tr_swz2(k) >> 1);
      offA[i] = k * (int)p.lda + (c16 >> 3) * 128 + ((c16 & 7) << 3);
    }
    if (B_T == 0) {
      const int rimg = i * 64 + wave * 8 + (lane >> 3);
      offB[i] = ((rimg >> 5) * 64 + (rimg & 31)) * (int)p.ldb + (((lane & 7) ^ ((rimg >> 1) & 7)) << 3);
    } else {
      const int k = i * 32 + wave * 4 + (lane >> 4);
      const int c16 = (lane & 15) ^ (tr_swz2(k) >> 1);
      offB[i] = k * (int)p.ldb + (c16 >> 2) * 64 + ((c16 & 3) << 3);
    }
  }
  const long halfA = ((A_T == 0) ? 64 * p.lda : 64) * h, halfB = (B_T == 0) ? 32 * p.ldb : 32;
  // operand bases of the next K-tile to stage, advanced by a constant per K-tile; past the last K-tile they stay (the
  // re-fetched half-tiles land in ring slots nobody reads again), so no phase carries an "is there a next K-tile" branch --
  // the same scalar diet as gemm3_kernel's loop (one instruction per ~4-5 cycles per wave: bookkeeping is not free)
  const char* ca = ((A_T == 0) ? p.A + ((long)tm * BM * p.lda) * 2 : p.A + ((long)tm * BM) * 2) + halfA * 2;
  const char* cb = (B_T == 0) ? p.B + ((long)tn * BN * p.ldb) * 2 : p.B + ((long)tn * BN) * 2;
  const long dA = (A_T == 0) ? (long)BK * 2 : (long)BK * p.lda * 2, dB = (B_T == 0) ? (long)BK * 2 : (long)BK * p.ldb * 2;
  int ckt = 0;
  auto cursor_next = [&](const char*& ab, const char*& bb) {
    ab = ca;
    bb = cb;
    if (ckt + 1 < nk) {
      ++ckt;
      ca += dA;
      cb += dB;
    }
  };

  unsigned dofs = 0;
  const unsigned smem_a = lds_addr_of(smem) + wave * 1024;
  auto issue_half = [&](const char* base, long off_el, const int (&off)[2]) {
    const unsigned dst = smem_a + dofs;
    dofs = (dofs + HALF) & (NSLOT * HALF - 1);
    const char* sb = base + off_el * 2;
    lds_dma16_s(sb, (unsigned)off[0] * 2u, dst);
    lds_dma16_s(sb, (unsigned)off[1] * 2u, dst + 8192);
  };

  const int tq = l15 >> 2, tp = l15 & 3;
  const int tr_f = (tq | ((lg & 1) << 2)) << 2;
  int kc_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) kc_off[s] = l15 * 128 + (((s * 4 + lg) ^ (l15 >> 1)) << 4);
  auto read_frag = [&](const char* img, int trans, int base16, int s) -> bf16x8 {
    if (!trans) {
      return *(const bf16x8*)(img + base16 * (BK * 2) + kc_off[s]);
    } else {
      const char* pa = img + (s * 32 + lg * 8 + tq) * 256 + ((((base16 >> 2) + tp) ^ tr_f) << 3);
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
  bf16x8 af[4][2], bq[2][2];

#define HT_MFMA(HB)                                                                              \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  __builtin_amdgcn_s_barrier();                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  __builtin_amdgcn_s_setprio(1);                                                                 \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) \
      _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                            \
          acc[i_][HB * 2 + j_] =                                                                 \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq[j_][s_], af[i_][s_], acc[i_][HB * 2 + j_], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  __builtin_amdgcn_s_barrier();

  // prologue: everything first read in phases 1..3 = A B0 B1 of K-tile 0, A B0 of K-tile 1 (stream indices 0..4)
  const char *a1, *b1, *a2, *b2;  // bases of K-tiles kt + 1, kt + 2 (clamped to the last one)
  {
    const char *a0, *b0;
    cursor_next(a0, b0);
    issue_half(a0, 0, offA);
    issue_half(b0, 0, offB);
    issue_half(b0, halfB, offB);
    cursor_next(a1, b1);
    issue_half(a1, 0, offA);
    issue_half(b1, 0, offB);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    cursor_next(a2, b2);
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();

  unsigned rofs = 0;  // byte offset of A of the current K-tile in the ring (stream index 3 kt)
  constexpr unsigned RING = NSLOT * HALF - 1;
  for (int kt = 0; kt < nk; ++kt) {
    const char* sA = smem + rofs;
    const char* sB0 = smem + ((rofs + HALF) & RING);
    const char* sB1 = smem + ((rofs + 2 * HALF) & RING);
    // P1 = (A, B0); issues B1 of K-tile kt + 1 (first read three phases on)
#pragma unroll
    for (int j_ = 0; j_ < 2; ++j_)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) bq[j_][s_] = read_frag(sB0, B_T, wc * 32 + j_ * 16, s_);
#pragma unroll
    for (int i_ = 0; i_ < 4; ++i_)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) af[i_][s_] = read_frag(sA, A_T, wr * 64 + i_ * 16, s_);
    issue_half(b1, halfB, offB);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    HT_MFMA(0)
    // P2 = (A, B1); issues A and B0 of K-tile kt + 2
#pragma unroll
    for (int j_ = 0; j_ < 2; ++j_)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) bq[j_][s_] = read_frag(sB1, B_T, wc * 32 + j_ * 16, s_);
    issue_half(a2, 0, offA);
    issue_half(b2, 0, offB);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    HT_MFMA(1)
    rofs = (rofs + 3 * HALF) & RING;
    a1 = a2;
    b1 = b2;
    cursor_next(a2, b2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped re-fetches are still landing in the ring
  if (grp == 0) __builtin_amdgcn_s_barrier();  // balance the stagger barrier: everyone is done with the ring
#undef HT_MFMA
  char* scr = smem + 8 * HALF + wave * SCR;
  const int m0 = tm * BM + wr * 128 + h * 64, n0 = tn * BN + wc * 64;
  if constexpr (EPI >= 3) pp_epilogue<EPI, 4>(acc, scr, p, m0, n0, 0, lane);
  else tile_epilogue<4, 4, CW, EPI>(acc, scr, p, m0, n0, 0, lane);
}

template <int AT, int BT, int EPI>
static int launch_half(const Gemm2Args& a, hipStream_t st) {
  constexpr int smem = 160 * 1024;
  static bool attr_done = false;
  auto fn = gemm3h_kernel<AT, BT, EPI>;
  if (!attr_done) {
    VIT_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, dim3(2 * a.tail_n), dim3(512), smem, st, a);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}
// instantiated for Y = X W^T with bias / dropout (3) or GELU (4) and for dX = dY W (6); the GELU kind only on request
// (gemm_half_tail = 2): at 9.2 rounds the 0.2 of a round it could save is less than the second launch costs (measured)
static int launch_half_cfg(const Gemm2Args& a, int epi, hipStream_t st) {
  if (epi == 3) return launch_half<0, 0, 3>(a, st);
  if (epi == 4) return launch_half<0, 0, 4>(a, st);
  return launch_half<0, 1, 6>(a, st);
}

// Sum of the tail tiles' K-slices + the epilogue the ping-pong kernel would have run (EPI 3: +bias, dropout -> bf16; 6: plain ->
// bf16): block = 8 rows x 256 columns of one tile, a thread owns 8 consecutive columns; slices are added in index order.
template <int EPI>
__global__ __launch_bounds__(256) void tail_reduce_kernel(Gemm2Args p) {
  resolve_drop(p.drop);
  const int tl = blockIdx.x >> 5, r = ((blockIdx.x & 31) << 3) + (threadIdx.x >> 5), cg = threadIdx.x & 31;
  const int t = p.tail_first + tl, tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
  const float* src = p.slab + (long)tl * 65536 + r * 256 + cg * 8;
  f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
  for (int s = 1; s < p.splits; ++s) {
    const float* q = src + (long)s * p.slab_tiles * 65536;
    v0 += *(const f32x4*)q;
    v1 += *(const f32x4*)(q + 4);
  }
  const long m = (long)tm * 256 + r;
  const int n = tn * 256 + cg * 8;
  float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  if (EPI == 3) {
    if (p.bias) {
      const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] += b0[i];
        o[4 + i] += b1[i];
      }
    }
    if (p.drop.thr) {
      const unsigned half_cols = (unsigned)(p.N >> 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float k0, k1;
        drop_pair(p.drop, (unsigned long long)m, half_cols, (unsigned)n + 2 * i, k0, k1);
        o[2 * i] *= k0;
        o[2 * i + 1] *= k1;
      }
    }
  }
  const u32x4 pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
  *(u32x4*)(p.C + (m * p.ldc + n) * 2) = pk;
}
static int launch_tail_reduce(const Gemm2Args& a, int epi, hipStream_t st) {
  if (epi == 3) hipLaunchKernelGGL(tail_reduce_kernel<3>, dim3(a.slab_tiles * 32), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(tail_reduce_kernel<6>, dim3(a.slab_tiles * 32), dim3(256), 0, st, a);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}

int g_split_tail = 1;  // vit_set_option("gemm_split_tail"): K-slices instead of half tiles for a short tail of a long-K GEMM
int g_grp2 = 1;  // vit_set_option("gemm_ngroups"): 1 = two N-groups for weights larger than an L2 (see tile_coords)
// vit_set_option("gemm_half_tail").  Default 0 since r05: the second launch shortens the N = 768 products themselves (-5...9 % in
// the micro-benchmark) but spends 5 % more CU-time (a half tile costs 0.7 of a tile), and the step is power-limited: with it off,
// one launch of ceil(tiles / rounds) workgroups leaves 59 CUs idle for the kernel and every OTHER kernel of the step runs 1-3 %
// faster (FC1 299 -> 292 us, attention backward 288 -> 281, dW 145.3 -> 143.3; step -0.4 ms at ViT-B on three boxes, ViT-L with
// its second stream 53.6 -> 51.7 ms)
int g_half_tail = 0;
int g_balance_wgs = 1;  // vit_set_option("gemm_balance_wgs")
int g_pp_slots = 8;  // vit_set_option("gemm_pp_slots"): half-tile slots of the ping-pong ring, 8 (default) or 10
template <int AT, int BT, int EPI, int NSLOT>
static int launch_stag_n(const Gemm2Args& a, dim3 grid, hipStream_t st) {
  constexpr int smem = 160 * 1024;
  static bool attr_done = false;
  auto fn = gemm3_kernel<AT, BT, EPI, NSLOT>;
  if (!attr_done) {
    VIT_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_done = true;
  }
  hipLaunchKernelGGL(fn, grid, dim3(512), smem, st, a);
  VIT_LAUNCH_CHECK();
  return VIT_OK;
}
template <int AT, int BT, int EPI>
static int launch_stag(const Gemm2Args& a, dim3 grid, hipStream_t st) {
  return launch_stag_n<AT, BT, EPI, 8>(a, grid, st);
}
static int launch_stag_cfg(const Gemm2Args& a, int at, int bt, int epi, dim3 grid, hipStream_t st) {
  if (epi == 1) return launch_stag<0, 0, 1>(a, grid, st);
  if (epi == 2) return launch_stag<0, 1, 2>(a, grid, st);
  if (epi == 3) return launch_stag<0, 0, 3>(a, grid, st);
  if (epi == 4) return launch_stag<0, 0, 4>(a, grid, st);
  if (epi == 5) return launch_stag<0, 1, 5>(a, grid, st);
  if (epi == 6) return launch_stag<0, 1, 6>(a, grid, st);
  if (epi == 8) return launch_stag<0, 0, 8>(a, grid, st);
  if (epi == 7) {
    if (at && bt) return launch_stag<1, 1, 7>(a, grid, st);
    if (!at && !bt) return launch_stag<0, 0, 7>(a, grid, st);  // the next two: K-slices of a forward / dX GEMM's tail tiles
    if (!at && bt) return launch_stag<0, 1, 7>(a, grid, st);
    return VIT_ERR_ARG;
  }
  if (!at && !bt) return launch_stag<0, 0, 0>(a, grid, st);
  if (!at && bt) return launch_stag<0, 1, 0>(a, grid, st);
  if (at && !bt) return launch_stag<1, 0, 0>(a, grid, st);
  return launch_stag<1, 1, 0>(a, grid, st);
}

extern thread_local int g_colsum_fused, g_rope_fused;  // gemm.hip
#ifdef VIT_PP_STAMP
unsigned long long* g_pp_stamps = nullptr;
int g_pp_stamp_block = 0;
#endif
int g_gemm2_mode = -1;  // -1: read VIT_GEMM2 from the environment on first use
#ifdef VIT_PP_DIAG
int g_gemm2_debug = 0;  // diagnostic twin build only: vit_debug_pp_diag()
#endif

// returns 1 if handled (rc in *rc), 0 if the shape is not eligible
int gemm2_try_launch(vit_handle h, const vit_gemm_desc* d, hipStream_t st, int* rc) {
  if (g_gemm2_mode < 0) {
    const char* e = getenv("VIT_GEMM2");
    g_gemm2_mode = e ? atoi(e) : 1;
  }
  // 0 = off (generic 128x128 core only); 1 = automatic, 5 = the same choice named explicitly: the 256x256x64 ping-pong kernel
  const int mode = g_gemm2_mode;
  if (mode == 0) return 0;
  if (d->M % 256 || d->N % 256 || d->K % 64) return 0;  // other shapes stay on the register-staged 128x128 core
  if (d->lda * 256 >= (1L << 30) || d->ldb * 256 >= (1L << 30)) return 0;  // int offsets inside a tile
  int epi = 0;
  if (d->act == VIT_ACT_GELU || d->act == VIT_ACT_GELU_GRAD) {
    if (d->a_trans || d->b_trans) return 0;
    epi = 1;
  } else if (d->act == VIT_ACT_DGELU || d->act == VIT_ACT_MUL_AUX) {
    if (d->a_trans || !d->b_trans) return 0;
    epi = 2;
  }
  constexpr int cfg = 5;
  const int bn = 256;
  const int slots = ctx_num_cus(h);  // one workgroup per CU

  Gemm2Args a;
  a.A = (const char*)d->A; a.B = (const char*)d->B; a.C = (char*)d->C;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.tiles_m = d->M / 256; a.tiles_n = d->N / bn;
  const int ntile = a.tiles_m * a.tiles_n;
  const int ktiles = d->K / 64;
  int splits = d->split_k;
  if (splits < 0) {
    splits = 1;
    if (ntile < slots) splits = std::min(std::max(1, slots / ntile), std::max(1, ktiles / 8));
  }
  if (splits < 1) splits = 1;
  if (splits > ktiles) splits = ktiles;
  const int kps = cdiv(ktiles, splits) * 64;
  splits = cdiv(d->K, kps);
  a.splits = splits; a.k_per_split = kps;
  a.slab = nullptr;
  if (splits > 1) {
    if (!(d->c_dtype == VIT_F32 && !d->bias && d->act == VIT_ACT_NONE && d->dropout_p == 0.f && !d->residual &&
          d->rows_per_batch == 0)) return 0;
    size_t wsb = 0;
    void* ws = ctx_workspace(h, &wsb);
    const size_t need = (size_t)splits * d->M * d->N * 4;
    if (!ws || wsb < need) {
      set_error("vit_gemm: split-K needs %zu workspace bytes, have %zu", need, wsb);
      *rc = VIT_ERR_WORKSPACE;
      return 1;
    }
    a.slab = (float*)ws;
  }
  a.nblk = std::min(ntile, slots);
  a.tile_limit = 0; a.tail_first = 0; a.tail_n = 0; a.slab_tiles = 0;
  a.bias = d->bias;
  a.aux_in = (const short*)d->aux_in; a.aux_out = (short*)d->aux_out; a.ldaux = d->ldaux;
  a.residual = d->residual; a.ldres = d->ldres;
  if (d->accumulate && splits == 1) {
    if (!(d->c_dtype == VIT_F32 && !d->residual && d->rows_per_batch == 0)) return 0;
    a.residual = (const float*)d->C; a.ldres = d->ldc;
  }
  a.alpha = d->alpha;
  a.rope_cos = d->rope_cos; a.rope_sin = d->rope_sin; a.rope_T = d->rope_T; a.rope_dh = d->rope_dh; a.rope_cols = d->rope_cols;
  a.act = d->act; a.c_dtype = d->c_dtype;
  a.drop = make_drop_h(h, d->dropout_p, d->seed, d->site);
  a.rpb = d->rows_per_batch; a.orb = d->out_batch_rows; a.roff = d->out_row_offset;
#ifdef VIT_PP_DIAG
  a.debug = g_gemm2_debug;
#endif
#ifdef VIT_PP_STAMP
  a.stamps = g_pp_stamps;
  a.stamp_block = g_pp_stamp_block;
#endif

  dim3 grid(a.nblk, splits);
  a.lin_split = 0;
  a.colsum_part = nullptr;
  if (cfg == 5 && splits > 1 && a.nblk == ntile) {
    a.lin_split = 1;
    grid = dim3(ntile * splits, 1);
  }
  int epi5 = epi;  // ping-pong kernel: the specialised epilogue when the descriptor is one of the five hot kinds
  if (cfg == 5) {
    const bool plain = !a.residual && a.rpb == 0 && d->alpha == 1.0f;
    if (epi == 1 && plain && a.bias && d->c_dtype == VIT_BF16 && !a.drop.thr && splits == 1) epi5 = 4;
    else if (epi == 2 && plain && !a.bias && d->c_dtype == VIT_BF16 && !a.drop.thr && splits == 1) epi5 = 5;
    else if (epi == 0 && plain && d->c_dtype == VIT_BF16 && !d->a_trans && !d->b_trans && splits == 1) epi5 = 3;
    else if (epi == 0 && plain && !a.bias && !a.drop.thr && d->c_dtype == VIT_BF16 && !d->a_trans && d->b_trans &&
             splits == 1) epi5 = 6;
    else if (epi == 0 && plain && !a.bias && !a.drop.thr && d->c_dtype == VIT_F32 && d->a_trans && d->b_trans) epi5 = 7;
    // rotary embedding requested: the rotating epilogue where a wave's 64 columns are whole heads, else the caller (gemm_launch)
    // runs the separate pass behind this launch
    if (d->rope_cos) {
      if (epi5 == 3 && !a.drop.thr && (d->rope_dh == 16 || d->rope_dh == 32 || d->rope_dh == 64) && (d->rope_cols % 64) == 0 &&
          d->rope_T >= 9)  // the epilogue steps its angles by table row 8
        epi5 = 8;
    }
    // the bf16 fast epilogues store through 32-bit buffer offsets (raw buffer stores, sc1): outputs past 2 GiB keep the
    // generic epilogue (a clamped descriptor would drop the stores beyond it silently)
    if (((epi5 >= 3 && epi5 <= 6) || epi5 == 8) &&
        ((unsigned long long)d->M * d->ldc * 2 >= 0x7FFFFFF0ull ||
         ((d->aux_out || d->aux_in) && (unsigned long long)d->M * d->ldaux * 2 >= 0x7FFFFFF0ull)))  // aux_in: epilogue 5's loads
      epi5 = epi;
  }
  // the tiles of a partial last round go to the half-tile kernel (two workgroups per tile) when the epilogue is one of
  // the kinds instantiated for it and nothing else rides on the launch
  int half_tail = 0;
  if (cfg == 5 && g_half_tail && (epi5 == 3 || epi5 == 6 || (epi5 == 4 && g_half_tail > 1)) && splits == 1 && ntile > slots && !d->colsum_out) {
    const int tail = ntile % slots;
    if (tail > 0 && 2 * tail <= slots && d->K / 64 >= 4) half_tail = tail;
  }
  if (cfg == 5 && g_balance_wgs && splits == 1 && ntile - half_tail > slots) {
    // multi-round persistent walk: the makespan is ceil(tiles / slots) tile-times whatever the workgroup count, so launch
    // just enough workgroups for that many rounds: the idle CUs' power budget goes to the busy ones' clock
    a.nblk = cdiv(ntile - half_tail, cdiv(ntile - half_tail, slots));
    grid = dim3(a.nblk, splits);
  }
  a.grp2 = 0;
  if (cfg == 5 && g_grp2 && splits == 1 && !half_tail && !a.lin_split && (a.tiles_n % 2) == 0 &&
      (size_t)d->N * d->K * 2 > (size_t)4 << 20 && ntile > slots && (ntile % 2) == 0) {
    a.grp2 = 1;
    if (a.nblk & 1) {  // each round must split into two equal halves
      a.nblk += 1;
      grid = dim3(a.nblk, splits);
    }
  }
  if (half_tail) {
    a.tile_limit = ntile - half_tail;
    a.tail_first = ntile - half_tail;
    a.tail_n = half_tail;
  }
  // A SHORT tail of a LONG-K product (ViT-L: 36 of 292 tiles, K = 3072 / 4096) runs as K-slices instead: every workgroup does a
  // whole 256 x 256 tile at the main loop's efficiency over K / s, s = slots / tail, and a small kernel adds the slices and runs
  // the epilogue.  The f32 partials cost 2 x s x 256 KiB of traffic per tile, so it only pays from s >= 4 (priced in DESIGN
  // section 7: a wash at ViT-B's 79-tile tails with s = 3).
  int st_splits = 0, st_kps = 0;
  if (half_tail && g_split_tail && (epi5 == 3 || epi5 == 6)) {
    const int s_ = slots / half_tail;
    if (s_ >= (g_split_tail >= 2 ? 3 : 4) && ktiles / s_ >= 6) {  // gemm_split_tail = 2: from 3 slices (ViT-B's 79-tile tails; measured a wash)
      st_kps = cdiv(ktiles, s_) * 64;
      st_splits = cdiv(d->K, st_kps);
      size_t wsb = 0;
      void* ws = ctx_workspace(h, &wsb);
      if (!ws || wsb < (size_t)st_splits * half_tail * 65536 * sizeof(float)) st_splits = 0;
    }
  }
  if (d->colsum_out && cfg == 5 && (epi5 == 3 || epi5 == 5 || epi5 == 6)) {
    size_t wsb = 0;
    void* ws = ctx_workspace(h, &wsb);
    if (ws && wsb >= (size_t)a.tiles_m * 2 * d->N * sizeof(float)) a.colsum_part = (float*)ws;
  }
  {
    const int at = epi ? 0 : d->a_trans, bt = epi == 1 ? 0 : (epi == 2 ? 1 : d->b_trans);
    snprintf(g_last_gemm, sizeof(g_last_gemm), "gemm3_kernel<%d, %d, %d, %d>", at, bt, epi5, 8);
  }
  if (epi5 == 8) g_rope_fused = 1;
  int r = launch_stag_cfg(a, d->a_trans, d->b_trans, epi5, grid, st);
  if (r == VIT_OK && half_tail && st_splits > 1) {
    Gemm2Args b = a;
    size_t wsb = 0;
    b.slab = (float*)ctx_workspace(h, &wsb);
    b.slab_tiles = half_tail;
    b.splits = st_splits; b.k_per_split = st_kps;
    b.lin_split = 1; b.grp2 = 0;
    b.tile_limit = half_tail;  // the kernel's tile count; tile ids are tail_first + 0 .. half_tail - 1
    b.nblk = half_tail;
    r = launch_stag_cfg(b, 0, d->b_trans, 7, dim3(half_tail * st_splits), st);
    if (r == VIT_OK) r = launch_tail_reduce(b, epi5, st);
  } else if (r == VIT_OK && half_tail) {
    r = launch_half_cfg(a, epi5, st);
  }
  if (r == VIT_OK && a.colsum_part) {
    r = launch_reduce_partials(a.colsum_part, a.tiles_m * 2, d->N, d->colsum_out, d->N, d->colsum_out, 0, st);
    g_colsum_fused = 1;
  }
  if (r == VIT_OK && splits > 1)
    r = launch_splitk_reduce(a.slab, (float*)d->C, (long)d->ldc, d->M, d->N, splits, d->alpha, d->accumulate, st);
  *rc = r;
  return 1;
}

}  // namespace vit
