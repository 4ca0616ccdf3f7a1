// Error reporting and ABI version for libmvlt_hip.so.
#include "common.h"
#include "../../include/mvlt_hip.h"
#include <stdarg.h>
#include <stdlib.h>
#include <cxxabi.h>

static thread_local char g_err[512] = "";

void mvlt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local hipError_t g_lds_err = hipSuccess;
void mvlt_note_lds_error(hipError_t e) { g_lds_err = e; }

int mvlt_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    if (g_lds_err != hipSuccess)
      mvlt_set_error("%s: launch failed: %s (raising the kernel's dynamic-LDS limit to 160 KB had failed on this device: %s)", what, hipGetErrorString(e), hipGetErrorString(g_lds_err));
    else
      mvlt_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    g_lds_err = hipSuccess;
    return MVLT_ERR_LAUNCH;
  }
  return MVLT_OK;
}

extern "C" const char* mvlt_last_error(void) { return g_err; }

// the instantiation MVLT_LAUNCH launched last on this thread: the runtime's name for the host function pointer, demangled
// ("void (anonymous namespace)::mlp_wgrad2_kernel<64, 4>(mvlt_mlp_args, int, int, int)")
static thread_local const void* g_last_kernel = nullptr;
static thread_local char g_last_kernel_name[512] = "";
void mvlt_note_kernel(const void* host_function) { g_last_kernel = host_function; }
extern "C" const char* mvlt_last_kernel(void) {
  g_last_kernel_name[0] = 0;
  if (!g_last_kernel) return g_last_kernel_name;
  const char* mangled = hipKernelNameRefByPtr(g_last_kernel, nullptr);
  if (!mangled) return g_last_kernel_name;
  int status = 0;
  char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
  snprintf(g_last_kernel_name, sizeof(g_last_kernel_name), "%s", (status == 0 && dem) ? dem : mangled);
  free(dem);
  return g_last_kernel_name;
}

// Bumped whenever an exported signature or argument struct changes (include/mvlt_hip.h MVLT_ABI_VERSION; mvlt_amd/_lib.py refuses a library whose
// number differs from the binding's: a stale build loaded through MVLT_HIP_LIB would otherwise be called with shifted positional arguments).
extern "C" int mvlt_abi_version(void) { return MVLT_ABI_VERSION; }

// sizeof() of every argument struct, so a foreign-language binding can verify its mirror of include/mvlt_hip.h
extern "C" int mvlt_sizeof(const char* name) {
#define S(T) if (strcmp(name, #T) == 0) return (int)sizeof(T);
  S(mvlt_rowmap) S(mvlt_prep_desc) S(mvlt_gemm_nt_args) S(mvlt_gemm_tn_args) S(mvlt_layernorm_args) S(mvlt_layernorm_bwd_args)
  S(mvlt_attn_args) S(mvlt_attn_bwd_args) S(mvlt_mlp_args)
#undef S
  return -1;
}
