// Shared device helpers for the MVLT gfx950 kernels (CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <string.h>
#include <stdio.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Streaming global accesses (`global_store_* ... nt`, `global_load_* ... nt`).  An activation or gradient tensor written by a GEMM epilogue
// is read back by a LATER launch, long after it would have left the 4 MB L2 of an XCD; written the normal way (write-back, line allocated)
// it pushes the operand tiles the running workgroups share -- the weights, the A rows of the n-tiles of one m-tile -- out of that L2.
// Measured (tools/ubench_mlpgemm.py, MI355X): stage-3 fc1 + GELU 212 -> 180 us, stage-4 190 -> 155 us, stage-4 gelu'-dgrad 224 -> 188 us,
// stage-3 fc1 dgrad 130 -> 110 us; whole step 23.05 -> 22.75 ms with the GEMM epilogues alone (bits 0 and 6).  The same hint on the pure
// streaming kernels is a LOSS: LayerNorm outputs (norm.hip) +0.27 ms -- their consumer is the next launch and the 256 MB Infinity Cache
// serves part of it, which `nt` gives up; the MIM decoder's BatchNorm / upsample kernels +0.05 ms; the fused-MLP and attention epilogues
// are neutral.  MVLT_NT_MASK selects the groups (A/B builds: tools/build_alt.sh ... -DMVLT_NT_MASK=0x..); the default is what measured best.
#ifndef MVLT_NT_MASK
#define MVLT_NT_MASK 0x41
#endif
#define MVLT_NT_GEMM ((MVLT_NT_MASK >> 0) & 1)
#define MVLT_NT_NORM ((MVLT_NT_MASK >> 1) & 1)
#define MVLT_NT_MLP ((MVLT_NT_MASK >> 2) & 1)
#define MVLT_NT_ATTN ((MVLT_NT_MASK >> 3) & 1)
#define MVLT_NT_EW ((MVLT_NT_MASK >> 4) & 1)
#define MVLT_NT_MIM ((MVLT_NT_MASK >> 5) & 1)
#define MVLT_NT_LD ((MVLT_NT_MASK >> 6) & 1)       // read-once operands of epilogues
#define MVLT_NT_NORM_LD ((MVLT_NT_MASK >> 7) & 1)  // rows a LayerNorm launch reads once
template <bool NTF, typename V> __device__ __forceinline__ void st_g(V* p, const V& v) {
  if constexpr (!NTF) *p = v;
  else if constexpr (sizeof(V) == 16) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), (u32x4*)p);
  else if constexpr (sizeof(V) == 8) __builtin_nontemporal_store(__builtin_bit_cast(u32x2, v), (u32x2*)p);
  else if constexpr (sizeof(V) == 4) __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, v), (uint32_t*)p);
  else { static_assert(sizeof(V) == 2, "st_g: 2 / 4 / 8 / 16-byte values"); __builtin_nontemporal_store(__builtin_bit_cast(uint16_t, v), (uint16_t*)p); }
}
template <bool NTF, typename V> __device__ __forceinline__ V ld_g(const V* p) {
  if constexpr (!NTF) return *p;
  else if constexpr (sizeof(V) == 16) return __builtin_bit_cast(V, __builtin_nontemporal_load((const u32x4*)p));
  else if constexpr (sizeof(V) == 8) return __builtin_bit_cast(V, __builtin_nontemporal_load((const u32x2*)p));
  else if constexpr (sizeof(V) == 4) return __builtin_bit_cast(V, __builtin_nontemporal_load((const uint32_t*)p));
  else { static_assert(sizeof(V) == 2, "ld_g: 2 / 4 / 8 / 16-byte values"); return __builtin_bit_cast(V, __builtin_nontemporal_load((const uint16_t*)p)); }
}

#define MVLT_OK 0
#define MVLT_ERR_ARG -1
#define MVLT_ERR_LAUNCH -2
#define MVLT_ERR_UNSUPPORTED -3

// thread-local last-error text (mvlt_last_error())
void mvlt_set_error(const char* fmt, ...);
int mvlt_check_launch(const char* what);

// Every kernel launch of the library goes through MVLT_LAUNCH, which remembers WHICH instantiation was launched last (its host-side function pointer, one
// store; the name is looked up and demangled only when asked for): mvlt_last_kernel() hands it to the caller, so that bench.py's `roofline` entries name the kernel that actually ran instead of a literal
// (after a dispatch change the prose could silently lie).
void mvlt_note_kernel(const void* host_function);
#define MVLT_LAUNCH(K, ...)                  \
  do {                                       \
    mvlt_note_kernel((const void*)(K));      \
    hipLaunchKernelGGL(K, __VA_ARGS__);      \
  } while (0)

// dynamic-LDS ceiling of a kernel instantiation raised to the 160 KB of a CU ONCE (the per-launch hipFuncSetAttribute calls of rounds 1-4 cost host time on
// ~55 launches per step); what a launch occupies is still the size it asks for
// -- once per (instantiation, DEVICE): the attribute is per device (a process that later launched on another GPU kept the 64 KB default there).  A failure is
// remembered on the thread and named by mvlt_check_launch when the launch that needed the room then fails (instead of an opaque "invalid argument").
void mvlt_note_lds_error(hipError_t e);
template <auto K> inline void mvlt_max_lds() {
  static std::atomic<unsigned> done{0};            // bit d: device d has the ceiling raised
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned bit = 1u << (dev & 31);
  if (done.load(std::memory_order_relaxed) & bit) return;
  const hipError_t e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { (void)hipGetLastError(); mvlt_note_lds_error(e); return; }
  done.fetch_or(bit, std::memory_order_relaxed);
}

// partial-tile reductions of the weight-gradient kernels (gemm.hip): the scratch region for `need` bytes of bf16 partial tiles of the output a.C [a.N1][a.N2] (nullptr: they do
// not fit / no scratch), and the fold of `splits` tiles [splits][N1][N2] into it (at once, or appended to the scratch's pending table: a.defer_fold).  Also used by mlp.hip.
struct mvlt_gemm_tn_args;
bf16* mvlt_fold_acquire_ext(const mvlt_gemm_tn_args& a, long need, hipStream_t s, int descriptors);       // room for that many fold_launch calls behind it
void mvlt_fold_launch_ext(const mvlt_gemm_tn_args& a, const bf16* part, int splits, hipStream_t s);

#define MVLT_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      mvlt_set_error(__VA_ARGS__);              \
      return MVLT_ERR_ARG;                      \
    }                                           \
  } while (0)

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7): one exp + one reciprocal + 5 FMAs instead of libm's
// ~40-instruction erff.  The GELU epilogues are VALU-bound otherwise (554 M activations per stage-1 fc1 launch).
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);   // v_rcp_f32 (1 ulp): __frcp_rn expands to the full IEEE division sequence
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float y = 1.0f - poly * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// d/dx [0.5 x (1+erf(x/sqrt2))] = 0.5(1+erf(x/sqrt2)) + x * exp(-x^2/2)/sqrt(2 pi)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// Two activations at a time on the packed-f32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 run 2 lanes-worth per issue): same
// A&S 7.1.26 arithmetic as erf_as, 10 instead of 17 instructions per activation; exp(-x^2/2) is shared between the
// erf tail and the Gaussian term of the derivative.  x * erf(x/sqrt2) = |x| * erf(|x|/sqrt2) removes the sign fix-up.
__device__ __forceinline__ void erf_parts2(f32x2 x, f32x2& erf_abs, f32x2& gauss) {
  const f32x2 az = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const f32x2 d = az * 0.3275911f + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2 poly = ((((t * 1.061405429f - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const f32x2 a = az * az * -1.4426950408889634f;
  gauss = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};       // exp(-x^2/2)
  erf_abs = 1.0f - poly * gauss;                                                      // erf(|x|/sqrt2)
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  f32x2 y, e;
  erf_parts2(x, y, e);
  const f32x2 hx = x * 0.5f;
  return hx + __builtin_elementwise_abs(hx) * y;
}
__device__ __forceinline__ f32x2 gelu_erf_grad2(f32x2 x) {
  f32x2 y, e;
  erf_parts2(x, y, e);
  const f32x2 hs = {copysignf(0.5f, x[0]), copysignf(0.5f, x[1])};
  return (hs * y + 0.5f) + x * (e * 0.39894228040143267794f);
}
__device__ __forceinline__ void gelu_erf_both2(f32x2 x, f32x2& g, f32x2& dg) {
  f32x2 y, e;
  erf_parts2(x, y, e);
  const f32x2 hx = x * 0.5f;
  g = hx + __builtin_elementwise_abs(hx) * y;
  const f32x2 hs = {copysignf(0.5f, x[0]), copysignf(0.5f, x[1])};
  dg = (hs * y + 0.5f) + x * (e * 0.39894228040143267794f);
}

// GELU for the bf16 fused-MLP kernels, which are VALU-bound on this function (one activation per 2 MFMA-flops at
// C = 64): Phi(x) ~ sigmoid(x * (p0 + p1 x^2 + p2 x^4)), a minimax fit of the logit on |x| <= 7 (clamped beyond, where
// Phi is 0 / 1 to 1e-11).  |gelu error| <= 3.0e-5, |gelu' error| <= 9.5e-5 absolute: 1-2 % of the bf16 rounding
// the result undergoes right after (relative L2 4e-6 / 7e-5 on N(0, 1.5^2) inputs); the fp32 parity path keeps the
// erf form above.  9 VALU ops (two transcendental) instead of 17; the derivative is the exact derivative of the
// approximant, so forward and backward stay consistent: g' = s + x s (1 - s) u'(x), 1 - s = e s.
#define MVLT_GP0 1.594965410743013f
#define MVLT_GP1 0.07397063459225359f
#define MVLT_GP2 -0.0006933278070537189f
#define MVLT_LOG2E 1.4426950408889634f
// Transcendental-free forms (MVLT_GELU_POLY bit 0: GELU' of the input-gradient kernels, bit 1: GELU of the forward, bit 2: the pair of the
// weight-gradient kernels).  Phi(x) - 1/2 and GELU'(x) - 1/2 are odd: x * P(x^2) on |x| <= 4, clamped beyond (Phi(4) = 1 - 3.2e-5,
// GELU'(4) = 1 + 5.0e-4), near-minimax fits (the fitting recipe is in DESIGN 6.0).  v_exp_f32 / v_rcp_f32 cost 2.3 plain VALU
// instructions each on gfx950 (profiles/r03_valu_rates.txt): GELU' is 10 plain instructions here against 15 + 2 transcendentals
// (19.6 units) above.  |GELU' error| <= 2.8e-4 + the 5e-4 tail, |Phi error| <= 1.0e-4 absolute (fp32 Horner).
#ifndef MVLT_GELU_POLY
#define MVLT_GELU_POLY 3
#endif
__device__ __forceinline__ float gelu_poly_phi_s(float xc, float s) {          // Phi(xc) for |xc| <= 4, s = xc^2
  float r = 2.816099979e-08f;
  r = __builtin_fmaf(r, s, -1.891889492e-06f);
  r = __builtin_fmaf(r, s, 5.419039730e-05f);
  r = __builtin_fmaf(r, s, -8.789809206e-04f);
  r = __builtin_fmaf(r, s, 9.112951399e-03f);
  r = __builtin_fmaf(r, s, -6.538835501e-02f);
  r = __builtin_fmaf(r, s, 3.985269119e-01f);
  return __builtin_fmaf(xc, r, 0.5f);
}
__device__ __forceinline__ float gelu_poly_dg_s(float xc, float s) {           // GELU'(xc) for |xc| <= 4
  float r = -1.641974535e-08f;
  r = __builtin_fmaf(r, s, 1.213803683e-06f);
  r = __builtin_fmaf(r, s, -3.845951304e-05f);
  r = __builtin_fmaf(r, s, 6.876447493e-04f);
  r = __builtin_fmaf(r, s, -7.687437410e-03f);
  r = __builtin_fmaf(r, s, 5.591481231e-02f);
  r = __builtin_fmaf(r, s, -2.620298260e-01f);
  r = __builtin_fmaf(r, s, 7.967216338e-01f);
  return __builtin_fmaf(xc, r, 0.5f);
}
// The Phi polynomial overshoots past its fit range (Phi_poly(4) = 1 + 7.2e-5, Phi_poly(-4) = -7.2e-5): clamped at +-4 the forward returned
// +7.2e-5 |x| for x < -4 -- wrong sign, growing with |x|.  It crosses 1 / 0 at |x| = 3.98504; clamped at 3.985 it saturates at
// 1 - 6e-7 / 6e-7 (true Phi(3.985) = 1 - 3.4e-5: inside the fit's 1e-4 bound), so the tail is x * 6e-7 with the right sign, no extra instruction.
#define MVLT_GELU_PHI_CLAMP 3.985f
__device__ __forceinline__ float gelu_poly1(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -MVLT_GELU_PHI_CLAMP, MVLT_GELU_PHI_CLAMP);
  return x * gelu_poly_phi_s(xc, xc * xc);
}
__device__ __forceinline__ float gelu_poly_grad1(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
  return gelu_poly_dg_s(xc, xc * xc);
}
// GELU and GELU' together: Phi by the polynomial, the Gaussian term by one exponential (12 plain + 1 transcendental against 12 + 2)
__device__ __forceinline__ void gelu_poly_both1(float x, float& g, float& dg) {
  const float xc = __builtin_amdgcn_fmed3f(x, -MVLT_GELU_PHI_CLAMP, MVLT_GELU_PHI_CLAMP);
  const float s = xc * xc;
  const float ph = gelu_poly_phi_s(xc, s);
  const float e = __builtin_amdgcn_exp2f(s * (-0.5f * MVLT_LOG2E));
  g = x * ph;
  dg = __builtin_fmaf(xc * 0.39894228040143267794f, e, ph);
}

// (Round 5 built packed-f16 forms of the two polynomials -- centred variable t = (x/4)^2 - 1/2, last Horner step in f32: |dGELU| <= 4.2e-4, |dGELU'| <= 8.2e-4,
//  17 instead of 21-23 instructions per pair -- and measured no gain in the fused-MLP kernels (docs/experiments_r5.md 1; code: commit 75ee330;
//  error certificate: tools/probes/gelu_f16_probe.py -> profiles/r05_gelu_f16_probe.txt).  Not kept in the product.)
__device__ __forceinline__ void gelu_fast_parts2(f32x2 x, f32x2& xc2, f32x2& e, f32x2& sg) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -7.0f, 7.0f), __builtin_amdgcn_fmed3f(x[1], -7.0f, 7.0f)};
  xc2 = xc * xc;
  const f32x2 t = xc * ((xc2 * (-MVLT_LOG2E * MVLT_GP2) + (-MVLT_LOG2E * MVLT_GP1)) * xc2 + (-MVLT_LOG2E * MVLT_GP0));
  e = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};            // exp(-u)
  const f32x2 d = e + 1.0f;
  sg = f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};             // sigmoid(u) ~ Phi(x)
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
#if MVLT_GELU_POLY & 2
  return f32x2{gelu_poly1(x[0]), gelu_poly1(x[1])};
#endif
  f32x2 s2, e, sg;
  gelu_fast_parts2(x, s2, e, sg);
  return x * sg;
}
__device__ __forceinline__ void gelu_fast_both2(f32x2 x, f32x2& g, f32x2& dg) {
#if MVLT_GELU_POLY & 4
  {                                            // the polynomial form on the packed-f32 ops: 16 instructions per PAIR of activations
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -MVLT_GELU_PHI_CLAMP, MVLT_GELU_PHI_CLAMP), __builtin_amdgcn_fmed3f(x[1], -MVLT_GELU_PHI_CLAMP, MVLT_GELU_PHI_CLAMP)};
    const f32x2 s = xc * xc;
    f32x2 r = s * 2.816099979e-08f + -1.891889492e-06f;
    r = r * s + 5.419039730e-05f;
    r = r * s + -8.789809206e-04f;
    r = r * s + 9.112951399e-03f;
    r = r * s + -6.538835501e-02f;
    r = r * s + 3.985269119e-01f;
    const f32x2 ph = xc * r + 0.5f;
    const f32x2 a = s * (-0.5f * MVLT_LOG2E);
    const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    g = x * ph;
    dg = (xc * 0.39894228040143267794f) * e + ph;
    return;
  }
#endif
  f32x2 s2, e, sg;
  gelu_fast_parts2(x, s2, e, sg);
  g = x * sg;
  const f32x2 up = (s2 * (5.0f * MVLT_GP2) + (3.0f * MVLT_GP1)) * s2 + MVLT_GP0;    // u'(x)
  dg = (x * up) * (sg * sg * e) + sg;
}
__device__ __forceinline__ f32x2 gelu_fast_grad2(f32x2 x) {
#if MVLT_GELU_POLY & 1
  return f32x2{gelu_poly_grad1(x[0]), gelu_poly_grad1(x[1])};
#endif
  f32x2 g, dg;
  gelu_fast_both2(x, g, dg);
  return dg;
}

// ---- LDS-DMA (global_load_lds) + transposed LDS reads, shared by the GEMM and fused-MLP kernels
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 tr_frag(const char* lds_addr, int rowb) {
  typedef __attribute__((address_space(3))) s16x4* lptr;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds_addr));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds_addr + 4 * rowb));
  unsigned long long l = __builtin_bit_cast(unsigned long long, lo), h = __builtin_bit_cast(unsigned long long, hi);
  return u32x4{(unsigned)l, (unsigned)(l >> 32), (unsigned)h, (unsigned)(h >> 32)};
}
// LDS-DMA of 16 B per lane to lds_wave_base + 16*lane_id.  Inline asm on purpose: for the builtin hipcc orders every
// later LDS read after the DMA with s_waitcnt vmcnt(0) (it cannot prove the two buffers disjoint), which serialises
// the prefetch behind the MFMAs; the kernel waits vmcnt(0) itself, once per tile, in front of its barrier.
// same, second read 16 rows of 128 B below (the k-slot order of accumulator-born MFMA operands: two 16-row tiles paired)
__device__ __forceinline__ u32x4 tr_frag16(const char* lds_addr) {
  typedef __attribute__((address_space(3))) s16x4* lptr;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds_addr));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds_addr + 16 * 128));
  unsigned long long l = __builtin_bit_cast(unsigned long long, lo), h = __builtin_bit_cast(unsigned long long, hi);
  return u32x4{(unsigned)l, (unsigned)(l >> 32), (unsigned)h, (unsigned)(h >> 32)};
}
#ifndef MVLT_GLDS_MOD
#define MVLT_GLDS_MOD ""             // cache-policy bits of the LDS-DMA loads (" sc1", " nt", " sc0 sc1": measured, no gain -- DESIGN 6.0)
#endif
__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MVLT_GLDS_MOD "\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_wave_base) : "memory");
}
// zero source for LDS slots whose row / column / 3x3 tap does not exist: every thread issues the same number of DMAs per
// tile (the pipelines count them with s_waitcnt vmcnt(N)); 64 KB so that the reads spread over L2 channels
static __device__ __attribute__((aligned(256))) unsigned char g_zero_page[65536];

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// Row addressing shared by the GEMM operands / outputs.
//  mode 0: phys_row = m                                  (rows_per_batch == 0)
//          phys_row = (m / rpb) * batch_stride + offset + m % rpb
//  mode 1: non-overlapping r x r patch gather over a token grid (conv with kernel==stride):
//          logical row m = (b, oi, oj) over B x Ho x Wo, logical column segment s = (di, dj),
//          phys_row = b * tokens_in + (oi*r + di) * w_in + (oj*r + dj)
//  mode 2: 3 x 3, stride 1, zero-padded neighbourhood gather over an H x W pixel grid stored pixel-major:
//          logical row m = (b, y, x), logical column segment s = (dy, dx) in 0..2 x 0..2,
//          phys_row = b * tokens_in + (y+dy-1) * w_in + (x+dx-1), and the segment reads as ZERO outside the grid
struct RowMap {
  int mode;
  int rows_per_batch;   // mode 0: logical rows per batch (0 = identity)
  int batch_stride;     // mode 0: physical rows per batch
  int offset;           // mode 0: first physical row inside a batch
  int r, w_in, tokens_in, hw_out, w_out, c_seg;   // mode 1 / 2
  int h_in;             // mode 2
};

__device__ __forceinline__ long rowmap_base(const RowMap& rm, int m) {
  if (rm.mode == 0) {
    if (rm.rows_per_batch == 0) return m;
    int b = m / rm.rows_per_batch;
    return (long)b * rm.batch_stride + rm.offset + (m - b * rm.rows_per_batch);
  }
  int b = m / rm.hw_out;
  int rem = m - b * rm.hw_out;
  int oi = rem / rm.w_out, oj = rem - oi * rm.w_out;
  if (rm.mode == 2) return (long)b * rm.tokens_in + (long)oi * rm.w_in + oj;     // centre pixel
  return (long)b * rm.tokens_in + (long)(oi * rm.r) * rm.w_in + oj * rm.r;
}
// mode 2: pixel coordinates of logical row m
__device__ __forceinline__ void rowmap_yx(const RowMap& rm, int m, int& y, int& x) {
  int rem = m % rm.hw_out;
  y = rem / rm.w_out;
  x = rem - y * rm.w_out;
}
// mode 2: is neighbour (dy, dx) of pixel (y, x) inside the grid, and its physical-row offset from the centre
__device__ __forceinline__ bool rowmap_nb(const RowMap& rm, int seg, int y, int x, int& off) {
  int dy = seg / 3, dx = seg - dy * 3;
  off = (dy - 1) * rm.w_in + (dx - 1);
  return (unsigned)(y + dy - 1) < (unsigned)rm.h_in && (unsigned)(x + dx - 1) < (unsigned)rm.w_in;
}
// extra physical-row offset for patch segment s = di*r + dj (mode 1 only)
__device__ __forceinline__ int rowmap_seg(const RowMap& rm, int seg) {
  int di = seg / rm.r, dj = seg - di * rm.r;
  return di * rm.w_in + dj;
}
