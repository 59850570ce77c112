// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of vit_amd.
// Wave = 64 lanes everywhere; bf16 MFMA 16x16x32; LDS images are XOR-swizzled per access kind.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vit_amd.h"

namespace vit {

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // MFMA A/B fragment (8 bf16 = 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;   // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define LDS_AS __attribute__((address_space(3)))

// ---- error plumbing (host) -------------------------------------------------------------------
void set_error(const char* fmt, ...);
extern thread_local char g_last_gemm[96];
#define VIT_CHECK(cond, code, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      ::vit::set_error(__VA_ARGS__);           \
      return (code);                           \
    }                                          \
  } while (0)
#define VIT_HIP(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      ::vit::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return VIT_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)
#define VIT_LAUNCH_CHECK() VIT_HIP(hipGetLastError())

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(short h) {
  return __builtin_bit_cast(float, ((unsigned)(unsigned short)h) << 16);
}
__device__ __forceinline__ short f2bf(float f) {  // RNE; lowers to v_cvt_pk_bf16_f32 on gfx950, keeps NaN a NaN
  return __builtin_bit_cast(short, (__bf16)f);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 v = {lo, hi};
  bf2 r = __builtin_convertvector(v, bf2);
  return __builtin_bit_cast(unsigned, r);
}

// ---- dropout: counter-based, stateless, identical in every kernel that needs the same mask -----------
// Two levels.  Every ROW of a [rows, ncols] tensor has a 32-bit key, a full-strength hash of (seed keys, 64-bit row index):
// drop_rowkey.  Inside the row, one 32-bit word per PAIR of adjacent columns, drop_bits(rowkey, col / 2): element (row, col)
// draws its low (col even) or high (col odd) 16 bits.  keep <=> r16 >= thr, thr = round(p * 65536); kept values are scaled
// by 65536 / (65536 - thr) so the mask is exactly unbiased.  The row key costs ~10 integer ops and is amortised over the
// row (kernels keep it in a register or stage it in the LDS); the per-pair word is 7 ops (xor, 2 x multiply-xorshift) --
// the softmax / epilogue kernels are VALU-bound and the single-level hash of (row * ncols/2 + col/2) cost 14 per pair with
// its 64-bit index arithmetic.  One multiply-xorshift round per pair is NOT enough: adjacent pairs correlate at 0.07.
// The reference's own masks come from torch's Philox stream and are implementation-defined (they differ between
// its CPU and CUDA runs too), so only the distribution is part of the contract.
__device__ __forceinline__ unsigned drop_hash(unsigned k0, unsigned k1, unsigned long long idx) {
  unsigned x = (unsigned)idx ^ k0 ^ ((unsigned)(idx >> 32) * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x7feb352du; x += k1; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned drop_bits(unsigned rowkey, unsigned colpair) {
  unsigned x = (colpair ^ rowkey) * 0x2C1B3C6Du;
  x ^= x >> 15; x *= 0x297A2D39u; x ^= x >> 16;
  return x;
}
struct DropCfg {
  unsigned thr;   // 0 => dropout off
  float scale;    // 1/(1-p_eff)
  unsigned k0, k1;
  // optional per-step keys in DEVICE memory (vit_step_state_bind): XORed into (k0, k1) at kernel entry, so a captured
  // hipGraph draws fresh masks on every replay although its kernel arguments are frozen
  const unsigned* dyn;
};
__device__ __forceinline__ void resolve_drop(DropCfg& d) {
  if (d.thr && d.dyn) {
    d.k0 ^= d.dyn[0];
    d.k1 ^= d.dyn[1];
  }
}
__host__ __device__ inline DropCfg make_drop(float p, uint64_t seed, uint64_t site) {
  DropCfg d;
  unsigned thr = (p <= 0.f) ? 0u : (unsigned)(p * 65536.0f + 0.5f);
  if (thr > 65535u) thr = 65535u;
  d.thr = thr;
  d.scale = 65536.0f / (float)(65536u - thr);
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (site + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  d.k0 = (unsigned)z;
  d.k1 = (unsigned)(z >> 32);
  d.dyn = nullptr;
  return d;
}
__device__ __forceinline__ unsigned drop_rowkey(const DropCfg& d, unsigned long long row) { return drop_hash(d.k0, d.k1, row); }
// multiplier (0 or scale) for 2 adjacent columns starting at even column `col` (`half_cols` is no longer part of the hash;
// the parameter stays for the call sites' sake).  `row` is loop-invariant at most call sites: the compiler hoists its key.
__device__ __forceinline__ void drop_pair(const DropCfg& d, unsigned long long row, unsigned half_cols, unsigned col,
                                          float& m0, float& m1) {
  (void)half_cols;
  const unsigned h = drop_bits(drop_rowkey(d, row), col >> 1);
  m0 = ((h & 0xFFFFu) >= d.thr) ? d.scale : 0.f;
  m1 = ((h >> 16) >= d.thr) ? d.scale : 0.f;
}

// ---- erf GELU (HF hidden_act="gelu", src/models/builder.py:246): 0.5 x (1 + erf(x / sqrt 2)) ---------------------
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-rounding level and 4 orders below bf16 resolution):
// one v_rcp + one v_exp + 6 FMAs instead of libm's branchy erff (the GELU epilogue of the FC1 GEMM was VALU-bound).
// cdf(x) = Phi(x) and pdf-exponential e^{-x^2/2} share the same exponential, so gelu' costs no second transcendental.
__device__ __forceinline__ void phi_parts(float x, float& cdf, float& ex) {
  // t = 1 / (1 + p |x| / sqrt2) by a bare v_rcp_f32 (1 ulp; __frcp_rn expands to a 12-instruction IEEE divide)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
  ex = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // e^{-x^2/2} = 2^{-x^2 log2(e) / 2}
  float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);  // the 0.5 of 0.5 erfc folded into the coefficients
  poly = fmaf(poly, t, 0.5f * 1.421413741f);
  poly = fmaf(poly, t, 0.5f * -0.284496736f);
  poly = fmaf(poly, t, 0.5f * 0.254829592f);
  const float half_erfc = poly * t * ex;     // 0.5 * erfc(|x|/sqrt2)
  cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, ex;
  phi_parts(x, cdf, ex);
  return x * cdf;
}
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {  // gelu(x) and gelu'(x) from one phi_parts
  float cdf, ex;
  phi_parts(x, cdf, ex);
  g = x * cdf;
  dg = fmaf(x * 0.39894228040143267794f, ex, cdf);
}
// Two elements at a time on packed f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two FMAs per issue slot).
// The FC1 epilogue is pure VALU with every MFMA pipe idle and each wave issue-bound (one instruction per ~4-5 cycles): per pair of
// elements 14 packed + 8 single instructions instead of ~42 (r03).  Same formula as phi_parts; the sign select of the cdf
// becomes cdf = 0.5 + copysign(0.5 - half_erfc, x) (v_bfi), which is the same value: x >= 0 -> 1 - h, x < 0 -> h.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void gelu_both2(f32x2 x, f32x2& g, f32x2& dg) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 den = ax * (0.3275911f * 0.70710678118654752440f) + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const f32x2 e2 = (x * x) * -0.72134752044448170368f;
  const f32x2 ex = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
  f32x2 poly = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
  poly = poly * t + (0.5f * 1.421413741f);
  poly = poly * t + (0.5f * -0.284496736f);
  poly = poly * t + (0.5f * 0.254829592f);
  const f32x2 h = (poly * t) * ex;      // 0.5 * erfc(|x| / sqrt 2)
  const f32x2 d = 0.5f - h;             // >= 0
  const f32x2 cdf = {0.5f + __builtin_copysignf(d[0], x[0]), 0.5f + __builtin_copysignf(d[1], x[1])};
  g = x * cdf;
  dg = (x * 0.39894228040143267794f) * ex + cdf;
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  f32x2 g, dg;
  gelu_both2(x, g, dg);
  return g;
}
__device__ __forceinline__ float dgelu_erf(float x) {
  float cdf, ex;
  phi_parts(x, cdf, ex);
  return fmaf(x * 0.39894228040143267794f, ex, cdf);
}

// ---- wave reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- lane movement inside 16-lane rows (DPP) and between them (v_permlane16_swap); shared by the attention kernels and the
// GEMM epilogues
// column sums of a wave's 16-row x 4-column register tile over its rows (lanes that share lane >> 4), bf16-rounded like
// the stored values; the four lanes with l15 == 0 hold the result
__device__ __forceinline__ float dpp_add(float v, int ctrl_sel) {
  // v + (v moved within its 16-lane row): DPP moves are plain VALU ops; __shfl_xor goes through ds_bpermute (LDS crossbar)
  int m;
  const int iv = __builtin_bit_cast(int, v);
  switch (ctrl_sel) {
    case 0: m = __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xF, 0xF, false); break;   // row_ror:8
    case 1: m = __builtin_amdgcn_update_dpp(0, iv, 0x124, 0xF, 0xF, false); break;   // row_ror:4
    case 2: m = __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    default: m = __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
  }
  return v + __builtin_bit_cast(float, m);
}
// sum over groups of CPR (4 or 8) consecutive lanes, result in every lane of the group: quad swaps, then (8) the mirror of the
// 8-lane half row brings the other quad's sum
template <int CPR>
__device__ __forceinline__ float sum_lanes_cpr(float x) {
  x = dpp_add(x, 3);
  x = dpp_add(x, 2);
  if (CPR == 8) {
    const int m = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, false);  // row_half_mirror
    x += __builtin_bit_cast(float, m);
  }
  return x;
}
__device__ __forceinline__ f32x4 rows16_sum(f32x4 v) {  // every lane ends with the sum over its 16-lane row
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float x = v[c];
    x = dpp_add(x, 0); x = dpp_add(x, 1); x = dpp_add(x, 2); x = dpp_add(x, 3);
    v[c] = x;
  }
  return v;
}
__device__ __forceinline__ f32x4 bf_round4(u32x2 pk) {
  return (f32x4){__builtin_bit_cast(float, pk[0] << 16), __builtin_bit_cast(float, pk[0] & 0xFFFF0000u),
                 __builtin_bit_cast(float, pk[1] << 16), __builtin_bit_cast(float, pk[1] & 0xFFFF0000u)};
}

// Two adjacent 16-column tiles of one 16-row block, packed to bf16 (a lane holds 4 consecutive columns of each: 8 bytes + 8
// bytes), into ONE 16-byte store per lane: v_permlane16_swap trades the odd lane groups' first-tile data for the even
// groups' second-tile data, so an even group ends with 8 consecutive columns of the first tile, an odd group with 8 of the
// second.  Row-per-lane stores are issue-bound (each instruction touches 16 rows): half the instructions, half the time.
// Returns this lane's first column within the 32-column pair.
__device__ __forceinline__ int widen_pair(u32x2& a, u32x2& b, int lg) {
  auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  a[0] = r0[0]; b[0] = r0[1];
  a[1] = r1[0]; b[1] = r1[1];
  return (lg & 1) ? 16 + 4 * (lg - 1) : 4 * lg;
}

// ---- reductions across the 4 lane groups lane>>4 (all 64 lanes end with the result)
__device__ __forceinline__ float grp4_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  return fmaxf(x, __shfl_xor(x, 32, 64));
}
__device__ __forceinline__ float grp4_sum(float x) {
  x += __shfl_xor(x, 16, 64);
  return x + __shfl_xor(x, 32, 64);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // bare v_exp_f32

// ---- raw buffer resources: hardware bounds check, out-of-range lanes read 0 --------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned long long bytes) {
  unsigned n = bytes > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
constexpr unsigned OOB = 0x80000000u;  // any voffset >= num_records reads as zero

// ---- LDS-DMA by inline asm: global_load_lds_dwordx4 / _dword (16 / 4 bytes per lane from a per-lane global address to
// LDS[M0 + lane * size], no VGPR destination).  NOT __builtin_amdgcn_global_load_lds: with the builtin the compiler knows an
// LDS write is pending and, unable to prove that a transposing read does not alias it, writes `s_waitcnt vmcnt(0)` in front
// of the first ds_read_b64_tr_b16 that follows -- once per K-tile phase in every GEMM with a transposed operand (dX, dW), and
// in the attention kernels: the whole ring was drained although the hand-counted waits left 64 KiB in flight (r03 finding:
// ISA of gemm3_kernel<*,1,*,*>; the plain ds_read_b128 forms were never affected).  The asm form is invisible to that pass;
// these loads still count in vmcnt, in order, and every wait for them is a hand-counted s_waitcnt.  M0 is written in the
// statement that reads it (s_nop: M0 write -> use) and named in the clobber list (r04, advisor finding): the compiler's own
// users of M0 (its LDS-DMA builtin in the VIT_DMA_BUILTIN variant, v_readlane / v_movrel indexing, s_sendmsg) then see a
// definition here and neither keep a value live across the statement nor drop their own M0 write as redundant.  clang notes
// that M0 is a reserved register (-Winline-asm, silenced in build.py): it is not allocatable, the clobber is still recorded.
#ifdef VIT_DMA_BUILTIN  // A/B variant build only (python -m vit_amd.build --defs -DVIT_DMA_BUILTIN --tag dmab): the builtin form
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma4(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 4, 0, 0);
}
#else
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst /* wave-uniform */) {
  const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(a) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma4(const void* gsrc, void* lds_dst /* wave-uniform */) {
  const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(a) : "memory", "m0");
}
#endif
// The same two with a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset: the lane address is one
// 32-bit multiply-add instead of a 64-bit one per piece (the attention backward's DMA issue was ~180 cycles per piece, most of
// it address arithmetic: stamps).  base + off is the byte address; off < 2^32.
// lds_addr: the destination as a raw LDS byte address (lds_addr_of), wave-uniform -- a generic `char*` costs a null check and
// an aperture compare per piece when the compiler cannot see the address space through a select.
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ void lds_dma16_s(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds_addr /* wave-uniform */) {
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(a) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma4_s(const void* sbase /* wave-uniform */, unsigned voff, unsigned lds_addr /* wave-uniform */) {
  const unsigned a = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(a) : "memory", "m0");
}

// the handle's bound per-step state (api.hip): [key0, key1, lr, bc1, rsqrt_bc2, step] in device memory, or NULL
struct StepState { unsigned key0, key1; float lr, bc1, rsqrt_bc2; unsigned step; };
const StepState* ctx_step_state(vit_handle h);
int ctx_num_cus(vit_handle h);  // compute units of the handle's device (api.hip)
static inline DropCfg make_drop_h(vit_handle h, float p, uint64_t seed, uint64_t site) {
  DropCfg d = make_drop(p, seed, site);
  const StepState* s = h ? ctx_step_state(h) : nullptr;
  d.dyn = s ? &s->key0 : nullptr;
  return d;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// out[c] (+)= sum_{k < nblk} part[k * stride + c]; columns [0, split) go to out0, [split, split2) to out1 (- split),
// [split2, width) to out2 (- split2; split2 = 0: no third output).  64 columns x 16 row groups per block, fixed
// summation order (deterministic).  Defined in elementwise.hip.
int launch_reduce_partials(const float* part, int nblk, int width, float* out0, int split, float* out1, int accumulate,
                           hipStream_t st, int stride = 0 /* floats between partial rows; 0 = width */, int split2 = 0,
                           float* out2 = nullptr);

}  // namespace vit
