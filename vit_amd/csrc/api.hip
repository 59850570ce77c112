// Handle, error string and workspace plumbing of libvit_amd.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "common.h"

struct vit_ctx {
  int device;
  void* ws;
  size_t ws_bytes;
  const void* step_state;  // device memory: vit::StepState, or NULL (vit_step_state_bind)
  int num_cus;             // compute units of the device (read once in vit_create): grid size of the persistent kernels
  int reserve_cus;         // vit_handle_set_option("reserve_cus"): -1 = the process default (vit_set_option), else this handle's own
};

namespace vit {

extern int g_gemm2_mode, g_pp_slots, g_balance_wgs, g_half_tail, g_split_tail, g_grp2;  // gemm2.hip
extern int g_attn_split, g_attn_res_max_t, g_attn_bwd_fused, g_attn_fwd_waves, g_attn32_mfma, g_attn_bwd_dma;  // attention.hip

static thread_local char g_err[512] = "";
thread_local char g_last_gemm[96] = "";  // symbol of the kernel the last vit_gemm on this thread launched

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const StepState* ctx_step_state(vit_handle h) { return h ? (const StepState*)h->step_state : nullptr; }

// vit_set_option("reserve_cus", n): the persistent kernels (one workgroup per CU: ping-pong GEMMs, pair-pipelined attention
// backward) size their grids for n fewer CUs.  For data-parallel runs: an RCCL kernel that overlaps the backward needs CUs of
// its own (a 512-thread GEMM workgroup owns its CU's whole register file), and a persistent grid that does not fit runs its
// last workgroups as a second round -- twice the kernel time instead of n / 256 more.  Default 0; to be tuned on a measured
// scaling curve.
int g_reserve_cus = 0;
int ctx_num_cus(vit_handle h) {
  // without a handle (kernel-level calls of the C ABI that pass NULL): the current device, asked once
  static int dflt = 0;
  if (h) return std::max(1, h->num_cus - (h->reserve_cus >= 0 ? h->reserve_cus : g_reserve_cus));
  if (!dflt) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      dflt = n;
    else
      dflt = 256;
  }
  return std::max(1, dflt - g_reserve_cus);
}

void* ctx_workspace(vit_handle h, size_t* bytes) {
  if (!h) {
    *bytes = 0;
    return nullptr;
  }
  *bytes = h->ws_bytes;
  return h->ws;
}

}  // namespace vit

extern "C" {

int vit_version(void) { return VIT_AMD_VERSION; }

const char* vit_last_error(void) { return vit::g_err; }

int vit_create(vit_handle* out, int device) {
  VIT_CHECK(out, VIT_ERR_ARG, "vit_create: null out pointer");
  int n = 0;
  VIT_HIP(hipGetDeviceCount(&n));
  VIT_CHECK(device >= 0 && device < n, VIT_ERR_ARG, "vit_create: device %d out of range (%d visible)", device, n);
  hipDeviceProp_t prop;
  VIT_HIP(hipGetDeviceProperties(&prop, device));
  VIT_CHECK(strncmp(prop.gcnArchName, "gfx950", 6) == 0, VIT_ERR_UNSUPPORTED,
            "vit_create: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  vit_ctx* c = new vit_ctx();
  c->device = device;
  c->ws = nullptr;
  c->ws_bytes = 0;
  c->step_state = nullptr;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->reserve_cus = -1;
  *out = c;
  return VIT_OK;
}

int vit_destroy(vit_handle h) {
  delete h;
  return VIT_OK;
}

const char* vit_last_gemm_kernel(void) { return vit::g_last_gemm; }

int vit_set_option(const char* name, int value) {
  VIT_CHECK(name, VIT_ERR_ARG, "vit_set_option: null name");
  if (strcmp(name, "gemm_core") == 0) {
    VIT_CHECK(value == 0 || value == 1 || value == 5, VIT_ERR_ARG,
              "vit_set_option: gemm_core must be 0 (generic core), 1 (automatic) or 5 (ping-pong core; what 1 selects on aligned "
              "problems); the lock-step geometries 2 / 3 / 4 / 6 were removed in round 3");
    vit::g_gemm2_mode = value;
    return VIT_OK;
  }
  if (strcmp(name, "attn_split") == 0) {
    if (value < 1 || value > 8) return VIT_ERR_ARG;
    vit::g_attn_split = value;
    return VIT_OK;
  }
  if (strcmp(name, "attn_bwd_fused") == 0) {
    vit::g_attn_bwd_fused = value;
    return VIT_OK;
  }
  if (strcmp(name, "attn_bwd_dma") == 0) {
    vit::g_attn_bwd_dma = value ? 1 : 0;
    return VIT_OK;
  }
  if (strcmp(name, "reserve_cus") == 0) {
    if (value < 0 || value > 128) {
      vit::set_error("vit_set_option: reserve_cus takes 0 .. 128");
      return VIT_ERR_ARG;
    }
    vit::g_reserve_cus = value;
    return VIT_OK;
  }
  if (strcmp(name, "attn_fwd_waves") == 0) {
    if (value != 8 && value != 12) {
      vit::set_error("vit_set_option: attn_fwd_waves takes 8 or 12");
      return VIT_ERR_ARG;
    }
    vit::g_attn_fwd_waves = value;
    return VIT_OK;
  }
  if (strcmp(name, "attn32_mfma") == 0) {
    vit::g_attn32_mfma = value;
    return VIT_OK;
  }
  if (strcmp(name, "attn_res_max_t") == 0) {
    vit::g_attn_res_max_t = value;
    return VIT_OK;
  }
  if (strcmp(name, "gemm_split_tail") == 0) {
    vit::g_split_tail = value;
    return VIT_OK;
  }
  if (strcmp(name, "gemm_half_tail") == 0) {
    vit::g_half_tail = value;
    return VIT_OK;
  }
  if (strcmp(name, "gemm_ngroups") == 0) {
    vit::g_grp2 = value;
    return VIT_OK;
  }
  if (strcmp(name, "gemm_balance_wgs") == 0) {
    vit::g_balance_wgs = value;
    return VIT_OK;
  }
  if (strcmp(name, "gemm_pp_slots") == 0) {
    if (value != 8) return VIT_ERR_ARG;  // the 10-slot ring (measured 0-15 % slower) was removed with the r02 loop rewrite
    vit::g_pp_slots = value;
    return VIT_OK;
  }
  vit::set_error("vit_set_option: unknown option '%s'", name);
  return VIT_ERR_ARG;
}

int vit_handle_set_option(vit_handle h, const char* name, int value) {
  VIT_CHECK(h && name, VIT_ERR_ARG, "vit_handle_set_option: null handle or name");
  if (strcmp(name, "reserve_cus") == 0) {
    VIT_CHECK(value >= -1 && value <= 128, VIT_ERR_ARG, "vit_handle_set_option: reserve_cus takes -1 (process default) or 0 .. 128");
    h->reserve_cus = value;
    return VIT_OK;
  }
  vit::set_error("vit_handle_set_option: '%s' is not a per-handle option (only launch geometry is: reserve_cus)", name);
  return VIT_ERR_ARG;
}

int vit_step_state_bind(vit_handle h, void* state) {
  VIT_CHECK(h, VIT_ERR_ARG, "vit_step_state_bind: null handle");
  VIT_CHECK(((uintptr_t)state & 15) == 0, VIT_ERR_ARG, "vit_step_state_bind: state must be 16-byte aligned");
  h->step_state = state;
  return VIT_OK;
}

int vit_set_workspace(vit_handle h, void* ws, size_t bytes) {
  VIT_CHECK(h, VIT_ERR_ARG, "vit_set_workspace: null handle");
  VIT_CHECK(((uintptr_t)ws & 255) == 0, VIT_ERR_ARG, "vit_set_workspace: workspace must be 256-byte aligned");
  h->ws = ws;
  h->ws_bytes = bytes;
  return VIT_OK;
}

}  // extern "C"

#ifdef VIT_PP_DIAG
// diagnostic twin build only (python -m vit_amd.build --diag; tools/pp_diag.py): switch pieces of the ping-pong K loop off
namespace vit { extern int g_gemm2_debug; }
extern "C" int vit_debug_pp_diag(int bits) {
  vit::g_gemm2_debug = bits;
  return 0;
}
#endif
#ifdef VIT_PP_STAMP
// diagnostic build only (python -m vit_amd.build --stamps N): where gemm3_kernel's workgroup `block` spends its cycles
namespace vit { extern unsigned long long* g_pp_stamps; extern int g_pp_stamp_block; }
extern "C" int vit_debug_pp_stamps(void* device_buf_64x_u64, int block) {
  vit::g_pp_stamps = (unsigned long long*)device_buf_64x_u64;
  vit::g_pp_stamp_block = block;
  return 0;
}
#endif
