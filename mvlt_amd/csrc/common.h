// Shared device helpers for the MVLT gfx950 kernels (CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
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

#define MVLT_OK 0
#define MVLT_ERR_ARG -1
#define MVLT_ERR_LAUNCH -2
#define MVLT_ERR_UNSUPPORTED -3

// thread-local last-error text (mvlt_last_error())
void mvlt_set_error(const char* fmt, ...);
int mvlt_check_launch(const char* what);

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
