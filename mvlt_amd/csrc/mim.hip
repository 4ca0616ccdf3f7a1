// HBM-bound kernels of the MIM decoder (reference libs/vl_heads.py:107-165, "ITGHead"): train-mode BatchNorm over
// pixel-major [M = B*H*W, C] fp32 matrices (the conv3x3 themselves are mvlt_gemm_nt with the 3x3 row map), the
// align_corners=True bilinear resizes, and the feature products.  fp32 math throughout; bf16 copies are written only
// where the next consumer is an MFMA operand.
#include "common.h"
#include "../../include/mvlt_hip.h"

namespace {

constexpr int NT = 256;

template <typename TO> __device__ __forceinline__ void store4(TO* p, const f32x4& v);
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const f32x4& v) {
  st_g<MVLT_NT_MIM>((bf16x4*)p, bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]});
}
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4& v) { st_g<MVLT_NT_MIM>((f32x4*)p, v); }
// four consecutive gradient values as fp32 from an fp32 or bf16 tensor (the decoder's first backward stages hand their activations'
// gradients over in bf16: they are read twice by the BatchNorm backward and once by the producer)
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <> __device__ __forceinline__ f32x4 load4<_Float16>(const _Float16* p) {     // the fp16 pre-BatchNorm conv output (mvlt_gemm_nt out_dtype 2)
  const f16x4 v = *(const f16x4*)p;
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
  const bf16x4 v = *(const bf16x4*)p;
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// eight consecutive values (one 16-byte access of a two-byte type): the fp16-z forms of the BatchNorm kernels below move 16 bytes per lane and tensor
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8]) {
  const bf16x8 a = *(const bf16x8*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}
template <> __device__ __forceinline__ void load8<_Float16>(const _Float16* p, float (&v)[8]) {
  const f16x8 a = *(const f16x8*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}
template <typename TO> __device__ __forceinline__ void store8(TO* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  st_g<MVLT_NT_MIM>((f32x4*)p, f32x4{v[0], v[1], v[2], v[3]});
  st_g<MVLT_NT_MIM>((f32x4*)(p + 4), f32x4{v[4], v[5], v[6], v[7]});
}
template <> __device__ __forceinline__ void store8<_Float16>(_Float16* p, const float (&v)[8]) {
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
  st_g<MVLT_NT_MIM>((f16x8*)p, o);
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
  st_g<MVLT_NT_MIM>((bf16x8*)p, o);
}

// ---- per-column sum / sum of squares (BatchNorm batch statistics) and the two backward reductions
// mode 0: s1 += sum z, s2 += sum z^2          mode 1: s1 += sum dy, s2 += sum dy * xhat  (xhat = (z-mean)*rstd)
template <int MODE>
__global__ __launch_bounds__(NT) void col_reduce_kernel(const float* z, int ldz, const float* dy, int lddy, const float* mean, const float* rstd,
                                                        int M, int C, float* s1, float* s2) {
  extern __shared__ float sm[];             // [2][C]
  for (int i = threadIdx.x; i < 2 * C; i += NT) sm[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int MAXC = 4;                   // C <= 256
  float a1[MAXC], a2[MAXC], mu[MAXC], rs[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    a1[k] = 0.f; a2[k] = 0.f;
    int c = lane + 64 * k;
    mu[k] = (MODE == 1 && c < C) ? mean[c] : 0.f;
    rs[k] = (MODE == 1 && c < C) ? rstd[c] : 0.f;
  }
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
      int c = lane + 64 * k;
      if (c < C) {
        float zv = z[row * ldz + c];
        if (MODE == 0) { a1[k] += zv; a2[k] += zv * zv; }
        else { float d = dy[row * lddy + c]; a1[k] += d; a2[k] += d * (zv - mu[k]) * rs[k]; }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    int c = lane + 64 * k;
    if (c < C) { atomicAdd(&sm[c], a1[k]); atomicAdd(&sm[C + c], a2[k]); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += NT) { atomicAdd(&s1[i], sm[i]); atomicAdd(&s2[i], sm[C + i]); }
}

// The same two reductions with 16-byte loads: C/4 lanes span a row, NT/(C/4) rows per pass, four passes in flight per thread
// (the 4-byte form above keeps too few bytes in flight to reach HBM speed at C = 64).  C % 4 == 0, ld % 4 == 0, 16-byte aligned.
template <int MODE, int NT, typename TDY = float, typename TZ = float>
__global__ __launch_bounds__(NT) void col_reduce4_kernel(const TZ* z, int ldz, const TDY* dy, int lddy, const float* mean, const float* rstd,
                                                         long M, int C, float* s1, float* s2, int rows_per_wg) {
  extern __shared__ float sm[];             // [2][C]
  for (int i = threadIdx.x; i < 2 * C; i += NT) sm[i] = 0.f;
  __syncthreads();
  const int cq = C >> 2, rpp = NT / cq;
  const int tr = threadIdx.x / cq, tc = (threadIdx.x - tr * cq) * 4;
  if (tr < rpp) {
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, mu = a1, rs = a1;
    if (MODE == 1) { mu = *(const f32x4*)(mean + tc); rs = *(const f32x4*)(rstd + tc); }
    const long r0 = (long)blockIdx.x * rows_per_wg;
    const long r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
    long r = r0 + tr;
    constexpr int U = 4;
    for (; r + (long)(U - 1) * rpp < r1; r += (long)U * rpp) {
      f32x4 zv[U], dv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        zv[u] = load4<TZ>(z + (r + (long)u * rpp) * ldz + tc);
        if (MODE == 1) dv[u] = load4<TDY>(dy + (r + (long)u * rpp) * lddy + tc);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (MODE == 0) { a1[e] += zv[u][e]; a2[e] += zv[u][e] * zv[u][e]; }
          else { a1[e] += dv[u][e]; a2[e] += dv[u][e] * (zv[u][e] - mu[e]) * rs[e]; }
        }
    }
    for (; r < r1; r += rpp) {
      f32x4 zv = load4<TZ>(z + r * ldz + tc), dv = zv;
      if (MODE == 1) dv = load4<TDY>(dy + r * lddy + tc);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (MODE == 0) { a1[e] += zv[e]; a2[e] += zv[e] * zv[e]; }
        else { a1[e] += dv[e]; a2[e] += dv[e] * (zv[e] - mu[e]) * rs[e]; }
      }
    }
    if (64 % cq == 0) {                       // (see col_reduce8_kernel)
      for (int o = cq; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { a1[e] += __shfl_xor(a1[e], o); a2[e] += __shfl_xor(a2[e], o); }
      if ((threadIdx.x & 63) < cq) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { atomicAdd(&sm[tc + e], a1[e]); atomicAdd(&sm[C + tc + e], a2[e]); }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { atomicAdd(&sm[tc + e], a1[e]); atomicAdd(&sm[C + tc + e], a2[e]); }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += NT) { atomicAdd(&s1[i], sm[i]); atomicAdd(&s2[i], sm[C + i]); }
}

// mean / rstd from (sum, sumsq) and the running-stat update of nn.BatchNorm2d (momentum 0.1, unbiased running var)
__global__ void bn_finalize_kernel(const float* sum, const float* sumsq, int copies, int M, int C, float eps, float momentum,
                                   float* mean, float* rstd, float* running_mean, float* running_var) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < copies; ++k) { s1 += sum[(long)k * C + c]; s2 += sumsq[(long)k * C + c]; }
  float m = s1 / M;
  float var = fmaxf(s2 / M - m * m, 0.f);
  mean[c] = m;
  rstd[c] = rsqrtf(var + eps);
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
}

// y = (z - mean) * rstd * gamma + beta   -> fp32 (optional) and / or bf16 (optional), each with its own row stride
template <typename TO, typename TZ = float>
__global__ __launch_bounds__(NT) void bn_norm_kernel(const TZ* z, int ldz, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                     long M, int C, float* y32, int ld32, TO* y16, int ld16) {
  const int cq = C / 4;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * cq; i += (long)gridDim.x * NT) {
    long r = i / cq; int c = (int)(i - r * cq) * 4;
    f32x4 v = load4<TZ>(z + r * ldz + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean[c + e]) * rstd[c + e] * gamma[c + e] + beta[c + e];
    if (y32) st_g<MVLT_NT_MIM>((f32x4*)(y32 + r * ld32 + c), o);
    if (y16) store4<TO>(y16 + r * ld16 + c, o);
  }
}

// dz = gamma * rstd * (dy - s1/M - xhat * s2/M)   (bf16: it is the A operand of the conv dgrad / wgrad GEMMs)
template <typename TO, typename TDY = float, typename TZ = float>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const TDY* dy, int lddy, const TZ* z, int ldz, const float* mean, const float* rstd,
                                                          const float* gamma, const float* s1, const float* s2, long M, int C, TO* dz, int lddz,
                                                          float* g_beta, float* g_gamma) {
  const int cq = C / 4;
  const float invM = 1.0f / (float)M;
  if (g_beta && blockIdx.x == 0)             // BatchNorm parameter gradients: d beta += s1, d gamma += s2 (one writer)
    for (int c = threadIdx.x; c < C; c += NT) { g_beta[c] += s1[c]; g_gamma[c] += s2[c]; }
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * cq; i += (long)gridDim.x * NT) {
    long r = i / cq; int c = (int)(i - r * cq) * 4;
    f32x4 d = load4<TDY>(dy + r * lddy + c), zv = load4<TZ>(z + r * ldz + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xh = (zv[e] - mean[c + e]) * rstd[c + e];
      o[e] = gamma[c + e] * rstd[c + e] * (d[e] - s1[c + e] * invM - xh * s2[c + e] * invM);
    }
    store4<TO>(dz + r * lddz + c, o);
  }
}

// ---- the fp16-z forms (bf16 path; C % 8 == 0, strides multiples of 8, 16-byte aligned rows): eight channels per lane
// mode 1 reduction: s1 += sum dy, s2 += sum dy * xhat.  C % 64 == 0: a wave covers 8 rows x one 64-column group (lane = 8 * row + chunk), the waves of a
// workgroup are dealt to the C / 64 groups in turn, so the per-column partial sums fold inside the wave (xor 8 / 16 / 32) and only eight lanes per wave touch
// the LDS accumulators (the 4-wide kernel's LDS atomics were 32 .. 64-way contended: at the few rows per thread of these launches, most of its time)
template <int NT, typename TDY>
__global__ __launch_bounds__(NT) void col_reduce8_kernel(const _Float16* z, int ldz, const TDY* dy, int lddy, const float* mean, const float* rstd,
                                                         long M, int C, float* s1, float* s2, int rows_per_wg) {
  extern __shared__ float sm[];             // [2][C]
  for (int i = threadIdx.x; i < 2 * C; i += NT) sm[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = C >> 6, g = wave % G, wr = wave / G;
  const int rpp = (NT / 64 / G) * 8;                    // rows per pass
  const int tc = g * 64 + (lane & 7) * 8, tr = wr * 8 + (lane >> 3);
  float a1[8], a2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
  if (tr < rpp) {
    float mu[8], rs[8];
    load8<float>(mean + tc, mu);
    load8<float>(rstd + tc, rs);
    const long r0 = (long)blockIdx.x * rows_per_wg;
    const long r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
    long r = r0 + tr;
    constexpr int U = 4;
    for (; r + (long)(U - 1) * rpp < r1; r += (long)U * rpp) {
      float zv[U][8], dv[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        load8<_Float16>(z + (r + (long)u * rpp) * ldz + tc, zv[u]);
        load8<TDY>(dy + (r + (long)u * rpp) * lddy + tc, dv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a1[e] += dv[u][e]; a2[e] += dv[u][e] * (zv[u][e] - mu[e]) * rs[e]; }
    }
    for (; r < r1; r += rpp) {
      float zv[8], dv[8];
      load8<_Float16>(z + r * ldz + tc, zv);
      load8<TDY>(dy + r * lddy + tc, dv);
#pragma unroll
      for (int e = 0; e < 8; ++e) { a1[e] += dv[e]; a2[e] += dv[e] * (zv[e] - mu[e]) * rs[e]; }
    }
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] += __shfl_xor(a1[e], o); a2[e] += __shfl_xor(a2[e], o); }
  if (lane < 8 && tr < rpp) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(&sm[tc + e], a1[e]); atomicAdd(&sm[C + tc + e], a2[e]); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += NT) { atomicAdd(&s1[i], sm[i]); atomicAdd(&s2[i], sm[C + i]); }
}

template <typename TY>
__global__ __launch_bounds__(NT) void bn_norm8_kernel(const _Float16* z, int ldz, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                      long M, int C, TY* y32, int ld32, bf16* y16, int ld16) {
  const int cq = C / 8;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * cq; i += (long)gridDim.x * NT) {
    long r = i / cq; int c = (int)(i - r * cq) * 8;
    float v[8], mu[8], rs[8], ga[8], be[8], o[8];
    load8<_Float16>(z + r * ldz + c, v);
    load8<float>(mean + c, mu); load8<float>(rstd + c, rs); load8<float>(gamma + c, ga); load8<float>(beta + c, be);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[e] - mu[e]) * rs[e] * ga[e] + be[e];
    if (y32) store8<TY>(y32 + r * ld32 + c, o);
    if (y16) store8<bf16>(y16 + r * ld16 + c, o);
  }
}

// the same with bn_finalize_kernel's work in its prologue (eleven 64..192-thread launches of ~6.5 us each per step otherwise): every workgroup derives mean / rstd of
// the C <= 256 channels from the conv epilogue's sums into LDS -- same operation order as bn_finalize_kernel, so the statistics are bit-identical -- and workgroup 0
// stores them for the backward pass and updates the running statistics
template <typename TY>
__global__ __launch_bounds__(NT) void bn_fin_norm8_kernel(const _Float16* z, int ldz, const float* sum, const float* sumsq, int copies, long Mstat, float eps, float momentum,
                                                          float* mean, float* rstd, float* running_mean, float* running_var, const float* gamma, const float* beta,
                                                          long M, int C, TY* y32, int ld32, bf16* y16, int ld16) {
  __shared__ __attribute__((aligned(16))) float s_mu[256], s_rs[256];
  for (int c = threadIdx.x; c < C; c += NT) {
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < copies; ++k) { s1 += sum[(long)k * C + c]; s2 += sumsq[(long)k * C + c]; }
    const float m = s1 / (int)Mstat;
    const float var = fmaxf(s2 / (int)Mstat - m * m, 0.f);
    const float r = rsqrtf(var + eps);
    s_mu[c] = m; s_rs[c] = r;
    if (blockIdx.x == 0) {
      mean[c] = m; rstd[c] = r;
      if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
        const float unb = Mstat > 1 ? var * ((float)(int)Mstat / (float)((int)Mstat - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
      }
    }
  }
  __syncthreads();
  const int cq = C / 8;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * cq; i += (long)gridDim.x * NT) {
    long r = i / cq; int c = (int)(i - r * cq) * 8;
    float v[8], ga[8], be[8], o[8];
    load8<_Float16>(z + r * ldz + c, v);
    load8<float>(gamma + c, ga); load8<float>(beta + c, be);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (v[e] - s_mu[c + e]) * s_rs[c + e] * ga[e] + be[e];
    if (y32) store8<TY>(y32 + r * ld32 + c, o);
    if (y16) store8<bf16>(y16 + r * ld16 + c, o);
  }
}

template <typename TDY>
__global__ __launch_bounds__(NT) void bn_bwd_apply8_kernel(const TDY* dy, int lddy, const _Float16* z, int ldz, const float* mean, const float* rstd,
                                                           const float* gamma, const float* s1, const float* s2, long M, int C, bf16* dz, int lddz,
                                                           float* g_beta, float* g_gamma) {
  const int cq = C / 8;
  const float invM = 1.0f / (float)M;
  if (g_beta && blockIdx.x == 0)             // BatchNorm parameter gradients: d beta += s1, d gamma += s2 (one writer)
    for (int c = threadIdx.x; c < C; c += NT) { g_beta[c] += s1[c]; g_gamma[c] += s2[c]; }
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * cq; i += (long)gridDim.x * NT) {
    long r = i / cq; int c = (int)(i - r * cq) * 8;
    float d[8], zv[8], mu[8], rs[8], ga[8], t1[8], t2[8], o[8];
    load8<TDY>(dy + r * lddy + c, d);
    load8<_Float16>(z + r * ldz + c, zv);
    load8<float>(mean + c, mu); load8<float>(rstd + c, rs); load8<float>(gamma + c, ga); load8<float>(s1 + c, t1); load8<float>(s2 + c, t2);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (zv[e] - mu[e]) * rs[e];
      o[e] = ga[e] * rs[e] * (d[e] - t1[e] * invM - xh * t2[e] * invM);
    }
    store8<bf16>(dz + r * lddz + c, o);
  }
}

// out (+)= a * b (* c)   fp32, optional bf16 copy of the final value
template <typename TO, typename TI = float>
__global__ __launch_bounds__(NT) void ew_mul_kernel(float* out, int ldo, const TI* a, int lda, const TI* b, int ldb, const TI* c3, int ldc,
                                                    long M, int C, int accumulate, TO* o16, int ld16) {
  const int cq = C / 4;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * cq; i += (long)gridDim.x * NT) {
    long r = i / cq; int c = (int)(i - r * cq) * 4;
    f32x4 v = load4<TI>(a + r * lda + c);
    f32x4 w = load4<TI>(b + r * ldb + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= w[e];
    if (c3) {
      f32x4 u = load4<TI>(c3 + r * ldc + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= u[e];
    }
    if (accumulate) {
      f32x4 o = *(const f32x4*)(out + r * ldo + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += o[e];
    }
    if (out) st_g<MVLT_NT_MIM>((f32x4*)(out + r * ldo + c), v);
    if (o16) store4<TO>(o16 + r * ld16 + c, v);
  }
}

// gradients of the three-way product y = a * b * c (reference libs/vl_heads.py:152: conv_upsample2(..) * conv_upsample3(..) * low):
// da = dy b c, db = dy a c, dc = dy a b in one pass (three ew_mul launches read dy and two of the factors each)
template <typename TDY, typename TO, typename TI = float>
__global__ __launch_bounds__(NT) void ew_mul3_bwd_kernel(const TDY* dy, int lddy, const TI* a, const TI* b, const TI* c, int ld,
                                                         TO* da, TO* db, TO* dc, long M, int C) {
  const int cq = C / 4;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * cq; i += (long)gridDim.x * NT) {
    long r = i / cq; int col = (int)(i - r * cq) * 4;
    const f32x4 g = load4<TDY>(dy + r * lddy + col);
    const f32x4 va = load4<TI>(a + r * ld + col), vb = load4<TI>(b + r * ld + col), vc = load4<TI>(c + r * ld + col);
    f32x4 oa, ob, oc;
#pragma unroll
    for (int e = 0; e < 4; ++e) { oa[e] = g[e] * vb[e] * vc[e]; ob[e] = g[e] * va[e] * vc[e]; oc[e] = g[e] * va[e] * vb[e]; }
    store4<TO>(da + r * ld + col, oa);
    store4<TO>(db + r * ld + col, ob);
    store4<TO>(dc + r * ld + col, oc);
  }
}

// ---- bilinear resize by an integer factor, align_corners=True (nn.Upsample in reference libs/vl_heads.py:114,134)
// forward: x fp32 pixel-major [B, H, W, C] (row stride ldx)  ->  out [B, sH, sW, C] (bf16 or fp32, row stride ldo)
//          or NCHW fp32 [B, C, sH, sW] when nchw != 0
template <typename TO>
__global__ __launch_bounds__(NT) void upsample_fwd_kernel(const float* x, int ldx, int B, int H, int W, int C, int s, TO* out, int ldo, int nchw) {
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long total = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    int c, ox, oy, b;
    if (nchw) { ox = (int)(i % Wo); long t = i / Wo; oy = (int)(t % Ho); t /= Ho; c = (int)(t % C); b = (int)(t / C); }
    else { c = (int)(i % C); long t = i / C; ox = (int)(t % Wo); t /= Wo; oy = (int)(t % Ho); b = (int)(t / Ho); }
    float fy = oy * ry, fx = ox * rx;
    int y0 = (int)fy, x0 = (int)fx;
    int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    float wy = fy - y0, wx = fx - x0;
    const float* xb = x + (long)b * H * W * ldx + c;
    float v00 = xb[((long)y0 * W + x0) * ldx], v01 = xb[((long)y0 * W + x1) * ldx];
    float v10 = xb[((long)y1 * W + x0) * ldx], v11 = xb[((long)y1 * W + x1) * ldx];
    float v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
    if (nchw) out[i] = (TO)v;
    else out[(((long)b * Ho + oy) * Wo + ox) * ldo + c] = (TO)v;
  }
}

// backward (adjoint, gather form): dx[b, iy, ix, c] (+)= sum over the output pixels whose bilinear footprint touches (iy, ix)
__global__ __launch_bounds__(NT) void upsample_bwd_kernel(const float* dy, int lddy, int nchw, int B, int H, int W, int C, int s, float* dx, int lddx, int accumulate) {
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    int c = (int)(i % C); long t = i / C; int ix = (int)(t % W); t /= W; int iy = (int)(t % H); int b = (int)(t / H);
    // output rows oy with |oy*ry - iy| < 1 : conservative integer window, exact weights decide
    int oy_lo = ry > 0.f ? max(0, (int)floorf((iy - 1) / ry)) : 0, oy_hi = ry > 0.f ? min(Ho - 1, (int)ceilf((iy + 1) / ry)) : Ho - 1;
    int ox_lo = rx > 0.f ? max(0, (int)floorf((ix - 1) / rx)) : 0, ox_hi = rx > 0.f ? min(Wo - 1, (int)ceilf((ix + 1) / rx)) : Wo - 1;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      float fy = oy * ry;
      int y0 = (int)fy; int y1 = min(y0 + 1, H - 1); float wy = fy - y0;
      float wyi = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);      // y0 == y1 at the last row: weights add to 1
      if (wyi == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        float fx = ox * rx;
        int x0 = (int)fx; int x1 = min(x0 + 1, W - 1); float wx = fx - x0;
        float wxi = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
        if (wxi == 0.f) continue;
        float g = nchw ? dy[(((long)b * C + c) * Ho + oy) * Wo + ox] : dy[(((long)b * Ho + oy) * Wo + ox) * lddy + c];
        acc += wyi * wxi * g;
      }
    }
    float* d = dx + (((long)b * H + iy) * W + ix) * lddx + c;
    *d = accumulate ? *d + acc : acc;
  }
}

// ---- row-per-workgroup versions of the two kernels above (the generic ones spend ~4 64-bit divisions per element on
// index decomposition).  A workgroup = one output row (b, oy) [forward] / input row (b, iy) [backward]; threads cover
// (pixel, 4-channel group) pairs with float4 accesses (pixel-major layout, C % 4 == 0), or pixels of one channel plane
// (NCHW side of the final x8 upsample, C = 3).  Row interpolation weights are uniform per workgroup.
template <typename TO>
__global__ __launch_bounds__(NT) void upsample_fwd_row_kernel(const float* x, int ldx, int H, int W, int C, int s, TO* out, int ldo, int chunks) {
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int b = row / Ho, oy = row - b * Ho;
  const int C4 = C >> 2;
  const int idx = chunk * NT + threadIdx.x;
  if (idx >= Wo * C4) return;
  const int ox = idx / C4, c = (idx - ox * C4) << 2;
  const float fy = oy * ry, fx = ox * rx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float wy = fy - y0, wx = fx - x0;
  const float* xb = x + (long)b * H * W * ldx + c;
  const f32x4 v00 = *(const f32x4*)(xb + ((long)y0 * W + x0) * ldx), v01 = *(const f32x4*)(xb + ((long)y0 * W + x1) * ldx);
  const f32x4 v10 = *(const f32x4*)(xb + ((long)y1 * W + x0) * ldx), v11 = *(const f32x4*)(xb + ((long)y1 * W + x1) * ldx);
  const f32x4 v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
  TO* o = out + (((long)b * Ho + oy) * Wo + ox) * ldo + c;
  if constexpr (sizeof(TO) == 4) st_g<MVLT_NT_MIM>((f32x4*)o, v);
  else st_g<MVLT_NT_MIM>((bf16x4*)o, bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]});
}
// NCHW fp32 output: workgroup = (b, c, oy) plane row, threads = ox
__global__ __launch_bounds__(NT) void upsample_fwd_nchw_kernel(const float* x, int ldx, int H, int W, int C, int s, float* out, int nrows) {
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  for (int rr = 0; rr < 8; ++rr) {                 // 8 plane rows per workgroup (a row alone is too little work per launch slot)
    const int row = blockIdx.x * 8 + rr;           // (b * C + c) * Ho + oy
    if (row >= nrows) return;
    const int oy = row % Ho, bc = row / Ho;
    const int c = bc % C, b = bc / C;
    const float fy = oy * ry;
    const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
    const float wy = fy - y0;
    const float* xb = x + (long)b * H * W * ldx + c;
    for (int ox = threadIdx.x; ox < Wo; ox += NT) {
      const float fx = ox * rx;
      const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
      const float wx = fx - x0;
      const float v00 = xb[((long)y0 * W + x0) * ldx], v01 = xb[((long)y0 * W + x1) * ldx];
      const float v10 = xb[((long)y1 * W + x0) * ldx], v11 = xb[((long)y1 * W + x1) * ldx];
      out[(long)row * Wo + ox] = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
    }
  }
}
// the same for W <= 64 and Wo % 4 == 0 (the x8 score upsampling to the image: W = 32, Wo = 256): the two source rows of each of
// the workgroup's 8 plane rows go through LDS once, a thread interpolates four neighbouring outputs and stores 16 bytes
// (4-byte stores, one per lane, held the first version at 2 TB/s of output)
__global__ __launch_bounds__(NT) void upsample_fwd_nchw4_kernel(const float* x, int ldx, int H, int W, int C, int s, float* out, int nrows) {
  __shared__ float src[8][2][64];
  __shared__ float s_wy[8];
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int row_base = blockIdx.x * 8;
  for (int i = threadIdx.x; i < 8 * 2 * W; i += NT) {
    const int rr = i / (2 * W), rem = i - rr * 2 * W, yy = rem / W, xx = rem - yy * W;
    const int row = row_base + rr;
    if (row < nrows) {
      const int oy = row % Ho, bc = row / Ho;
      const int c = bc % C, b = bc / C;
      const float fy = oy * ry;
      const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
      src[rr][yy][xx] = x[((long)b * H * W + (long)(yy ? y1 : y0) * W + xx) * ldx + c];
      if (rem == 0) s_wy[rr] = fy - y0;
    }
  }
  __syncthreads();
  const int q = threadIdx.x & 63, rsub = threadIdx.x >> 6;
  if (q * 4 >= Wo) return;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rr = pass * 4 + rsub, row = row_base + rr;
    if (row >= nrows) continue;
    const float wy = s_wy[rr];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ox = q * 4 + e;
      const float fx = ox * rx;
      const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
      const float wx = fx - x0;
      o[e] = (1.f - wy) * ((1.f - wx) * src[rr][0][x0] + wx * src[rr][0][x1]) + wy * ((1.f - wx) * src[rr][1][x0] + wx * src[rr][1][x1]);
    }
    st_g<MVLT_NT_MIM>((f32x4*)(out + (long)row * Wo + q * 4), o);
  }
}
// backward, pixel-major dy: workgroup = input row (b, iy), threads = (ix, 4-channel group)
template <typename TDY>
__global__ __launch_bounds__(NT) void upsample_bwd_row_kernel(const TDY* dy, int lddy, int H, int W, int C, int s, float* dx, int lddx, int accumulate, int chunks) {
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int b = row / H, iy = row - b * H;
  const int C4 = C >> 2;
  const int idx = chunk * NT + threadIdx.x;
  if (idx >= W * C4) return;
  const int ix = idx / C4, c = (idx - ix * C4) << 2;
  const int oy_lo = ry > 0.f ? max(0, (int)floorf((iy - 1) / ry)) : 0, oy_hi = ry > 0.f ? min(Ho - 1, (int)ceilf((iy + 1) / ry)) : Ho - 1;
  const int ox_lo = rx > 0.f ? max(0, (int)floorf((ix - 1) / rx)) : 0, ox_hi = rx > 0.f ? min(Wo - 1, (int)ceilf((ix + 1) / rx)) : Wo - 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float fy = oy * ry;
    const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
    const float wy = fy - y0;
    const float wyi = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
    if (wyi == 0.f) continue;
    const TDY* drow = dy + ((long)b * Ho + oy) * Wo * lddy + c;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float fx = ox * rx;
      const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
      const float wx = fx - x0;
      const float wxi = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
      if (wxi == 0.f) continue;
      acc += (wyi * wxi) * load4<TDY>(drow + (long)ox * lddy);
    }
  }
  float* d = dx + (((long)b * H + iy) * W + ix) * lddx + c;
  if (accumulate) acc += *(const f32x4*)d;
  st_g<MVLT_NT_MIM>((f32x4*)d, acc);
}
// the same with the scale a template parameter: an input pixel is touched by at most 2 S + 1 output rows / columns, so the two tap-weight
// vectors are computed ONCE per thread into registers (the generic kernel recomputes the column weights inside the row loop: ~500 VALU
// instructions per thread for 25 taps of which ~9-16 are non-zero -- the x2 resizes of the decoder were VALU-bound at ~2 TB/s)
template <typename TDY, int S>
__global__ __launch_bounds__(NT) void upsample_bwd_row_s_kernel(const TDY* dy, int lddy, int H, int W, int C, float* dx, int lddx, int accumulate, int chunks) {
  constexpr int T = 2 * S + 1;
  const int Ho = H * S, Wo = W * S;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int row = blockIdx.x / chunks, chunk = blockIdx.x - row * chunks;
  const int b = row / H, iy = row - b * H;
  const int C4 = C >> 2;
  const int idx = chunk * NT + threadIdx.x;
  if (idx >= W * C4) return;
  const int ix = idx / C4, c = (idx - ix * C4) << 2;
  const int oy_lo = ry > 0.f ? max(0, (int)floorf((iy - 1) / ry)) : 0, ox_lo = rx > 0.f ? max(0, (int)floorf((ix - 1) / rx)) : 0;
  float wyv[T], wxv[T];
#pragma unroll
  for (int k = 0; k < T; ++k) {
    const int oy = oy_lo + k, ox = ox_lo + k;
    const float fy = oy * ry, fx = ox * rx;
    const int y0 = (int)fy, y1 = min(y0 + 1, H - 1), x0 = (int)fx, x1 = min(x0 + 1, W - 1);
    const float wy = fy - y0, wx = fx - x0;
    wyv[k] = oy < Ho ? (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f) : 0.f;
    wxv[k] = ox < Wo ? (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f) : 0.f;
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const TDY* base = dy + (((long)b * Ho + oy_lo) * Wo + ox_lo) * lddy + c;
#pragma unroll
  for (int ky = 0; ky < T; ++ky) {
    if (wyv[ky] == 0.f) continue;
    const TDY* drow = base + (long)ky * Wo * lddy;
#pragma unroll
    for (int kx = 0; kx < T; ++kx) {
      if (wxv[kx] == 0.f) continue;
      acc += (wyv[ky] * wxv[kx]) * load4<TDY>(drow + (long)kx * lddy);
    }
  }
  float* d = dx + (((long)b * H + iy) * W + ix) * lddx + c;
  if (accumulate) acc += *(const f32x4*)d;
  st_g<MVLT_NT_MIM>((f32x4*)d, acc);
}
// backward, NCHW dy (final x8 upsample, C = 3): workgroup = (b, c, iy).  Pass 1: thread ox folds its output column over
// the rows that touch iy (coalesced plane-row reads) into LDS; pass 2: thread ix folds the <= 2s+1 columns that touch it.
template <typename TO>
__global__ __launch_bounds__(NT) void upsample_bwd_nchw_kernel(const float* dy, int H, int W, int C, int s, TO* dx, int lddx, int accumulate) {
  extern __shared__ float colsum[];                // [Wo]
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int row = blockIdx.x;                      // (b * C + c) * H + iy
  const int iy = row % H, bc = row / H;
  const int c = bc % C, b = bc / C;
  const int oy_lo = ry > 0.f ? max(0, (int)floorf((iy - 1) / ry)) : 0, oy_hi = ry > 0.f ? min(Ho - 1, (int)ceilf((iy + 1) / ry)) : Ho - 1;
  const float* plane = dy + (long)bc * Ho * Wo;
  for (int ox = threadIdx.x; ox < Wo; ox += NT) {
    float t = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const float fy = oy * ry;
      const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
      const float wy = fy - y0;
      const float wyi = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
      t += wyi * plane[(long)oy * Wo + ox];
    }
    colsum[ox] = t;
  }
  __syncthreads();
  for (int ix = threadIdx.x; ix < W; ix += NT) {
    const int ox_lo = rx > 0.f ? max(0, (int)floorf((ix - 1) / rx)) : 0, ox_hi = rx > 0.f ? min(Wo - 1, (int)ceilf((ix + 1) / rx)) : Wo - 1;
    float t = 0.f;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float fx = ox * rx;
      const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
      const float wx = fx - x0;
      const float wxi = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
      t += wxi * colsum[ox];
    }
    TO* d = dx + (((long)b * H + iy) * W + ix) * lddx + c;
    *d = from_f32<TO>(accumulate ? to_f32<TO>(*d) + t : t);
  }
}

// the same for Wo % 4 == 0, Wo <= 256: 16-byte loads, the ~2s contributing output rows split over the four waves
template <typename TO>
__global__ __launch_bounds__(NT) void upsample_bwd_nchw4_kernel(const float* dy, int H, int W, int C, int s, TO* dx, int lddx, int accumulate) {
  __shared__ __attribute__((aligned(16))) float part[4][256];
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int row = blockIdx.x;                      // (b * C + c) * H + iy
  const int iy = row % H, bc = row / H;
  const int c = bc % C, b = bc / C;
  const int oy_lo = ry > 0.f ? max(0, (int)floorf((iy - 1) / ry)) : 0, oy_hi = ry > 0.f ? min(Ho - 1, (int)ceilf((iy + 1) / ry)) : Ho - 1;
  const float* plane = dy + (long)bc * Ho * Wo;
  const int q = threadIdx.x & 63, g = threadIdx.x >> 6;
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  if (q * 4 < Wo) {
    for (int oy = oy_lo + g; oy <= oy_hi; oy += 4) {
      const float fy = oy * ry;
      const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
      const float wy = fy - y0;
      const float wyi = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
      t += wyi * *(const f32x4*)(plane + (long)oy * Wo + q * 4);
    }
    *(f32x4*)(&part[g][q * 4]) = t;
  }
  __syncthreads();
  for (int ix = threadIdx.x; ix < W; ix += NT) {
    const int ox_lo = rx > 0.f ? max(0, (int)floorf((ix - 1) / rx)) : 0, ox_hi = rx > 0.f ? min(Wo - 1, (int)ceilf((ix + 1) / rx)) : Wo - 1;
    float acc = 0.f;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float fx = ox * rx;
      const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
      const float wx = fx - x0;
      const float wxi = (x0 == ix ? 1.f - wx : 0.f) + (x1 == ix ? wx : 0.f);
      acc += wxi * (part[0][ox] + part[1][ox] + part[2][ox] + part[3][ox]);
    }
    TO* d = dx + (((long)b * H + iy) * W + ix) * lddx + c;
    *d = from_f32<TO>(accumulate ? to_f32<TO>(*d) + acc : acc);
  }
}

// ---- the MIM loss without the image-sized logits (training): SmoothL1(mean) of the x s bilinear upsample (align_corners=True) of the
// low-resolution score map x[B,H,W,C] (pixel-major, row stride ldx) against the NCHW target -- the upsampled prediction (201 MB at
// batch 256) is never written: the forward reduces the loss while it interpolates (same tiling as upsample_fwd_nchw4_kernel), the
// backward recomputes the prediction of the <= 2s+1 output rows that touch a score row and folds clamp(pred - target) back onto the
// score map in one pass (same tiling as upsample_bwd_nchw4_kernel).  Reference: libs/vl_heads.py:163-165 + engine_grid_masking.py:99.
__global__ __launch_bounds__(NT) void upsample_l1_fwd_kernel(const float* x, int ldx, int H, int W, int C, int s, const float* target, int nrows,
                                                             float* loss_sum) {
  __shared__ float src[8][2][64];
  __shared__ float s_wy[8];
  __shared__ float s_w[NT / 64];
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int q = threadIdx.x & 63, rsub = threadIdx.x >> 6;
  float acc = 0.f;
  // a workgroup walks several 8-row groups and ends with ONE atomic (24576 same-address atomics, one per row group, cost more than
  // the whole interpolation)
  for (int row_base = blockIdx.x * 8; row_base < nrows; row_base += gridDim.x * 8) {
    __syncthreads();                                 // the previous group's reads of src are done
    for (int i = threadIdx.x; i < 8 * 2 * W; i += NT) {
      const int rr = i / (2 * W), rem = i - rr * 2 * W, yy = rem / W, xx = rem - yy * W;
      const int row = row_base + rr;
      if (row < nrows) {
        const int oy = row % Ho, bc = row / Ho;
        const int c = bc % C, b = bc / C;
        const float fy = oy * ry;
        const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
        src[rr][yy][xx] = x[((long)b * H * W + (long)(yy ? y1 : y0) * W + xx) * ldx + c];
        if (rem == 0) s_wy[rr] = fy - y0;
      }
    }
    __syncthreads();
    if (q * 4 < Wo) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int rr = pass * 4 + rsub, row = row_base + rr;
        if (row >= nrows) continue;
        const float wy = s_wy[rr];
        const f32x4 t = *(const f32x4*)(target + (long)row * Wo + q * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ox = q * 4 + e;
          const float fx = ox * rx;
          const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
          const float wx = fx - x0;
          const float o = (1.f - wy) * ((1.f - wx) * src[rr][0][x0] + wx * src[rr][0][x1]) + wy * ((1.f - wx) * src[rr][1][x0] + wx * src[rr][1][x1]);
          const float d = fabsf(o - t[e]);
          acc += d < 1.0f ? 0.5f * d * d : d - 0.5f;
        }
      }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, s_w[0] + s_w[1] + s_w[2] + s_w[3]);
}

template <typename TO>
__global__ __launch_bounds__(NT) void upsample_l1_bwd_kernel(const float* x, int ldx, int H, int W, int C, int s, const float* target, const float* gscale,
                                                             float inv_n, TO* dx, int lddx) {
  __shared__ __attribute__((aligned(16))) float part[4][256];
  __shared__ float lr[3][64];                      // score rows iy - 1, iy, iy + 1 (clamped) of channel c
  const int Ho = H * s, Wo = W * s;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const int row = blockIdx.x;                      // (b * C + c) * H + iy
  const int iy = row % H, bc = row / H;
  const int c = bc % C, b = bc / C;
  for (int i = threadIdx.x; i < 3 * W; i += NT) {
    const int k = i / W, xx = i - k * W;
    const int y = min(max(iy - 1 + k, 0), H - 1);
    lr[k][xx] = x[((long)b * H * W + (long)y * W + xx) * ldx + c];
  }
  __syncthreads();
  const int oy_lo = ry > 0.f ? max(0, (int)floorf((iy - 1) / ry)) : 0, oy_hi = ry > 0.f ? min(Ho - 1, (int)ceilf((iy + 1) / ry)) : Ho - 1;
  const float* plane = target + (long)bc * Ho * Wo;
  const int q = threadIdx.x & 63, g = threadIdx.x >> 6;
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  if (q * 4 < Wo) {
    int x0[4]; float wx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float fx = (q * 4 + e) * rx; x0[e] = (int)fx; wx[e] = fx - x0[e]; }
    for (int oy = oy_lo + g; oy <= oy_hi; oy += 4) {
      const float fy = oy * ry;
      const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
      const float wy = fy - y0;
      const float wyi = (y0 == iy ? 1.f - wy : 0.f) + (y1 == iy ? wy : 0.f);
      if (wyi == 0.f) continue;
      const float* r0 = lr[y0 - iy + 1];           // y0, y1 are within iy - 1 .. iy + 1 whenever wyi != 0
      const float* r1 = lr[y1 - iy + 1];
      const f32x4 tg = *(const f32x4*)(plane + (long)oy * Wo + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int xa = x0[e], xb = min(xa + 1, W - 1);
        const float o = (1.f - wy) * ((1.f - wx[e]) * r0[xa] + wx[e] * r0[xb]) + wy * ((1.f - wx[e]) * r1[xa] + wx[e] * r1[xb]);
        t[e] += wyi * __builtin_amdgcn_fmed3f(o - tg[e], -1.0f, 1.0f);
      }
    }
    *(f32x4*)(&part[g][q * 4]) = t;
  }
  __syncthreads();
  const float sc = gscale[0] * inv_n;
  for (int ix = threadIdx.x; ix < W; ix += NT) {
    const int ox_lo = rx > 0.f ? max(0, (int)floorf((ix - 1) / rx)) : 0, ox_hi = rx > 0.f ? min(Wo - 1, (int)ceilf((ix + 1) / rx)) : Wo - 1;
    float acc = 0.f;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float fx = ox * rx;
      const int xa = (int)fx, xb = min(xa + 1, W - 1);
      const float w = fx - xa;
      const float wxi = (xa == ix ? 1.f - w : 0.f) + (xb == ix ? w : 0.f);
      acc += wxi * (part[0][ox] + part[1][ox] + part[2][ox] + part[3][ox]);
    }
    dx[(((long)b * H + iy) * W + ix) * lddx + c] = from_f32<TO>(acc * sc);
  }
}

inline int grid_for(long work, int cap = 8192) {
  long g = (work + NT - 1) / NT;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

constexpr int RNT = 1024, RWGS = 256;     // the vectorised reductions: one workgroup per CU (every extra one adds 2C same-line atomics)
inline bool vec4_ok(const float* p, int ld, int C) { return C % 4 == 0 && C <= 256 && ld % 4 == 0 && ((uintptr_t)p & 15) == 0; }
// about 1024 workgroups, each a whole number of four-pass iterations
inline int reduce_rows_per_wg(long M, int C, int nt, int wgs) {
  const int step = 4 * (nt / (C / 4));
  long rows = (M + wgs - 1) / wgs;
  rows = (rows + step - 1) / step * step;
  return (int)rows;
}

}  // namespace

extern "C" int mvlt_col_stats(const float* z, int ldz, long M, int C, float* sum, float* sumsq, void* stream) {
  MVLT_REQUIRE(z && sum && sumsq && C > 0 && C <= 256 && ldz >= C, "mvlt_col_stats: bad arguments (C <= 256)");
  if (M <= 0) return MVLT_OK;
  if (vec4_ok(z, ldz, C)) {
    int rows_per_wg = reduce_rows_per_wg(M, C, RNT, RWGS);
    int grid = (int)((M + rows_per_wg - 1) / rows_per_wg);
    MVLT_LAUNCH((col_reduce4_kernel<0, RNT>), dim3(grid), dim3(RNT), 2 * C * sizeof(float), (hipStream_t)stream, z, ldz, (const float*)nullptr, 0, nullptr, nullptr, M, C, sum, sumsq, rows_per_wg);
    return mvlt_check_launch("mvlt_col_stats");
  }
  int grid = (int)((M + 63) / 64); if (grid > 2048) grid = 2048;
  MVLT_LAUNCH((col_reduce_kernel<0>), dim3(grid), dim3(NT), 2 * C * sizeof(float), (hipStream_t)stream, z, ldz, nullptr, 0, nullptr, nullptr, (int)M, C, sum, sumsq);
  return mvlt_check_launch("mvlt_col_stats");
}

extern "C" int mvlt_bn_finalize(const float* sum, const float* sumsq, int copies, long M, int C, float eps, float momentum, float* mean, float* rstd,
                                float* running_mean, float* running_var, void* stream) {
  MVLT_REQUIRE(sum && sumsq && mean && rstd && M > 0 && C > 0 && copies >= 1, "mvlt_bn_finalize: bad arguments");
  MVLT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mvlt_bn_finalize: running_mean and running_var go together");
  MVLT_LAUNCH(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sum, sumsq, copies, (int)M, C, eps, momentum, mean, rstd, running_mean, running_var);
  return mvlt_check_launch("mvlt_bn_finalize");
}

extern "C" int mvlt_bn_norm(const void* z_, int ldz, int z_dtype, const float* mean, const float* rstd, const float* gamma, const float* beta, long M, int C,
                            void* y32_, int ld32, int y32_dtype, void* y16, int ld16, int op_dtype, void* stream) {
  float* y32 = (float*)y32_;
  MVLT_REQUIRE(!y32_ || y32_dtype == 1 || y32_dtype == 2, "mvlt_bn_norm: y32_dtype is 1 (fp32) or 2 (fp16)");
  MVLT_REQUIRE(z_ && mean && rstd && gamma && beta && (y32 || y16) && C % 4 == 0 && ldz % 4 == 0, "mvlt_bn_norm: bad arguments (C, ld multiples of 4)");
  MVLT_REQUIRE((!y32 || ld32 % 4 == 0) && (!y16 || ld16 % 4 == 0), "mvlt_bn_norm: output strides must be multiples of 4");
  MVLT_REQUIRE(z_dtype == 1 || (z_dtype == 2 && op_dtype == 0), "mvlt_bn_norm: z is fp32 (z_dtype 1), or fp16 (2) on the bf16 path");
  if (M <= 0) return MVLT_OK;
  const float* z = (const float*)z_;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (z_dtype == 2 && C % 8 == 0 && ldz % 8 == 0 && (!y32 || (ld32 % 4 == 0 && al16(y32))) && (!y16 || (ld16 % 8 == 0 && al16(y16))) && al16(z_) && al16(mean) && al16(rstd) &&
      al16(gamma) && al16(beta))
  {
    if (y32_ && y32_dtype == 2) {
      MVLT_REQUIRE(ld32 % 8 == 0, "mvlt_bn_norm: an fp16 y needs a row stride that is a multiple of 8");
      MVLT_LAUNCH((bn_norm8_kernel<_Float16>), dim3(grid_for(M * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const _Float16*)z_, ldz, mean, rstd, gamma, beta, M, C, (_Float16*)y32_, ld32, (bf16*)y16, ld16);
    } else MVLT_LAUNCH((bn_norm8_kernel<float>), dim3(grid_for(M * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const _Float16*)z_, ldz, mean, rstd, gamma, beta, M, C, y32, ld32, (bf16*)y16, ld16);
  }
  else if (y32_ && y32_dtype == 2) { mvlt_set_error("mvlt_bn_norm: an fp16 y exists with an fp16 z on the 8-wide path only (C, strides multiples of 8, 16-byte aligned)"); return MVLT_ERR_UNSUPPORTED; }
  else if (z_dtype == 2) MVLT_LAUNCH((bn_norm_kernel<bf16, _Float16>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, (const _Float16*)z_, ldz, mean, rstd, gamma, beta, M, C, y32, ld32, (bf16*)y16, ld16);
  else if (op_dtype == 0) MVLT_LAUNCH((bn_norm_kernel<bf16>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, z, ldz, mean, rstd, gamma, beta, M, C, y32, ld32, (bf16*)y16, ld16);
  else MVLT_LAUNCH((bn_norm_kernel<float>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, z, ldz, mean, rstd, gamma, beta, M, C, y32, ld32, (float*)y16, ld16);
  return mvlt_check_launch("mvlt_bn_norm");
}

extern "C" int mvlt_bn_finalize_norm(const void* z_, int ldz, const float* sum, const float* sumsq, int copies, float eps, float momentum, float* mean, float* rstd,
                                     float* running_mean, float* running_var, const float* gamma, const float* beta, long M, int C,
                                     void* y32_, int ld32, int y32_dtype, void* y16, int ld16, void* stream) {
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  MVLT_REQUIRE(z_ && sum && sumsq && mean && rstd && gamma && beta && (y32_ || y16) && M > 0 && M < (1L << 31) && copies >= 1, "mvlt_bn_finalize_norm: bad arguments");
  MVLT_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "mvlt_bn_finalize_norm: running_mean and running_var go together");
  MVLT_REQUIRE(C % 8 == 0 && C <= 256 && ldz % 8 == 0 && al16(z_) && al16(gamma) && al16(beta) && (!y32_ || (al16(y32_) && ld32 % (y32_dtype == 2 ? 8 : 4) == 0 && (y32_dtype == 1 || y32_dtype == 2))) &&
               (!y16 || (al16(y16) && ld16 % 8 == 0)),
               "mvlt_bn_finalize_norm: fp16 z, C a multiple of 8 and <= 256, row strides multiples of 8, 16-byte aligned tensors (the bf16 training path of the MIM decoder)");
  // every workgroup re-derives the C channels' statistics from copies x 2 x C floats before its first row: with the default cap (8192 workgroups: three 8-element chunks per
  // thread at 262144 x 192) that prologue outweighed the rows (round 6: the launches ran at 2.2-2.8 x their HBM time, tools/ceiling_table.py)
  static const int fin_cap = getenv("MVLT_BN_FIN_CAP") ? atoi(getenv("MVLT_BN_FIN_CAP")) : 2048;
  const dim3 grid(grid_for(M * (C / 8), fin_cap));
  if (y32_ && y32_dtype == 2)
    MVLT_LAUNCH((bn_fin_norm8_kernel<_Float16>), grid, dim3(NT), 0, (hipStream_t)stream, (const _Float16*)z_, ldz, sum, sumsq, copies, M, eps, momentum, mean, rstd, running_mean, running_var,
                       gamma, beta, M, C, (_Float16*)y32_, ld32, (bf16*)y16, ld16);
  else
    MVLT_LAUNCH((bn_fin_norm8_kernel<float>), grid, dim3(NT), 0, (hipStream_t)stream, (const _Float16*)z_, ldz, sum, sumsq, copies, M, eps, momentum, mean, rstd, running_mean, running_var,
                       gamma, beta, M, C, (float*)y32_, ld32, (bf16*)y16, ld16);
  return mvlt_check_launch("mvlt_bn_finalize_norm");
}

extern "C" int mvlt_bn_bwd_reduce(const void* dy_, int lddy, const void* z_, int ldz, int z_dtype, const float* mean, const float* rstd, long M, int C,
                                  float* s1, float* s2, int dy_dtype, void* stream) {
  MVLT_REQUIRE(dy_ && z_ && mean && rstd && s1 && s2 && C > 0 && C <= 256 && (dy_dtype == 0 || dy_dtype == 1) && (z_dtype == 1 || z_dtype == 2),
               "mvlt_bn_bwd_reduce: bad arguments (C <= 256)");
  if (M <= 0) return MVLT_OK;
  const float* dy = (const float*)dy_;
  const float* z = (const float*)z_;
  const bool dy_vec = dy_dtype == 1 ? vec4_ok(dy, lddy, C) : (C % 4 == 0 && C <= 256 && lddy % 4 == 0 && ((uintptr_t)dy_ & 7) == 0);
  const bool z_vec = z_dtype == 1 ? vec4_ok(z, ldz, C) : (C % 4 == 0 && ldz % 4 == 0 && ((uintptr_t)z_ & 7) == 0);
  MVLT_REQUIRE(dy_dtype == 1 || (dy_vec && z_vec), "mvlt_bn_bwd_reduce: bf16 dy needs C, ld multiples of 4 and aligned rows");
  if (z_vec && dy_vec && (((uintptr_t)mean | (uintptr_t)rstd) & 15) == 0) {
    // measured at M = 262144: C = 64 34.7 -> 30.6 us with 512 threads, C = 192 98.8 -> 80.0 us with 1024 (5.0 TB/s)
    const int nt = C <= 64 ? 512 : 1024;
    int rows_per_wg = reduce_rows_per_wg(M, C, nt, RWGS);
    int grid = (int)((M + rows_per_wg - 1) / rows_per_wg);
    const size_t lds = 2 * C * sizeof(float);
    const bf16* dyh = (const bf16*)dy_;
    const _Float16* zh = (const _Float16*)z_;
#define MVLT_RED_LAUNCH(NT_, TDY_, TZ_, DY_, Z_) \
    MVLT_LAUNCH((col_reduce4_kernel<1, NT_, TDY_, TZ_>), dim3(grid), dim3(NT_), lds, (hipStream_t)stream, Z_, ldz, DY_, lddy, mean, rstd, M, C, s1, s2, rows_per_wg)
    const bool wide = z_dtype == 2 && C % 64 == 0 && ldz % 8 == 0 && ((uintptr_t)z_ & 15) == 0 && ((uintptr_t)dy_ & 15) == 0 && lddy % (dy_dtype == 0 ? 8 : 4) == 0;
    if (wide) {
      const int nt8 = C == 64 ? 512 : C == 192 ? 768 : 1024;           // whole waves per 64-column group
      const int rpp8 = (nt8 / 64 / (C / 64)) * 8;
      long rows8 = (M + RWGS - 1) / RWGS;
      rows8 = (rows8 + 4 * rpp8 - 1) / (4 * rpp8) * (4 * rpp8);
      const int grid8 = (int)((M + rows8 - 1) / rows8);
#define MVLT_RED8(NT_, TDY_, DY_) MVLT_LAUNCH((col_reduce8_kernel<NT_, TDY_>), dim3(grid8), dim3(NT_), lds, (hipStream_t)stream, zh, ldz, DY_, lddy, mean, rstd, M, C, s1, s2, (int)rows8)
      if (dy_dtype == 0) { if (nt8 == 512) MVLT_RED8(512, bf16, dyh); else if (nt8 == 768) MVLT_RED8(768, bf16, dyh); else MVLT_RED8(1024, bf16, dyh); }
      else { if (nt8 == 512) MVLT_RED8(512, float, dy); else if (nt8 == 768) MVLT_RED8(768, float, dy); else MVLT_RED8(1024, float, dy); }
#undef MVLT_RED8
    } else if (z_dtype == 2) {
      if (dy_dtype == 0) { if (nt == 512) MVLT_RED_LAUNCH(512, bf16, _Float16, dyh, zh); else MVLT_RED_LAUNCH(1024, bf16, _Float16, dyh, zh); }
      else { if (nt == 512) MVLT_RED_LAUNCH(512, float, _Float16, dy, zh); else MVLT_RED_LAUNCH(1024, float, _Float16, dy, zh); }
    } else if (dy_dtype == 0) {
      if (nt == 512) MVLT_RED_LAUNCH(512, bf16, float, dyh, z); else MVLT_RED_LAUNCH(1024, bf16, float, dyh, z);
    } else {
      if (nt == 512) MVLT_RED_LAUNCH(512, float, float, dy, z); else MVLT_RED_LAUNCH(1024, float, float, dy, z);
    }
#undef MVLT_RED_LAUNCH
    return mvlt_check_launch("mvlt_bn_bwd_reduce");
  }
  MVLT_REQUIRE(dy_dtype == 1 && z_dtype == 1, "mvlt_bn_bwd_reduce: bf16 dy / fp16 z need the vectorised path (16-byte aligned mean / rstd)");
  int grid = (int)((M + 63) / 64); if (grid > 2048) grid = 2048;
  MVLT_LAUNCH((col_reduce_kernel<1>), dim3(grid), dim3(NT), 2 * C * sizeof(float), (hipStream_t)stream, z, ldz, dy, lddy, mean, rstd, (int)M, C, s1, s2);
  return mvlt_check_launch("mvlt_bn_bwd_reduce");
}

extern "C" int mvlt_bn_bwd_apply(const void* dy_, int lddy, const void* z_, int ldz, int z_dtype, const float* mean, const float* rstd, const float* gamma,
                                 const float* s1, const float* s2, long M, int C, void* dz_bf16, int lddz, float* g_beta, float* g_gamma,
                                 int op_dtype, int dy_dtype, void* stream) {
  MVLT_REQUIRE(dy_ && z_ && mean && rstd && gamma && s1 && s2 && dz_bf16 && C % 4 == 0 && lddy % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0 && (dy_dtype == 0 || dy_dtype == 1),
               "mvlt_bn_bwd_apply: bad arguments");
  MVLT_REQUIRE((g_beta == nullptr) == (g_gamma == nullptr), "mvlt_bn_bwd_apply: g_beta and g_gamma go together");
  MVLT_REQUIRE(z_dtype == 1 || (z_dtype == 2 && op_dtype == 0), "mvlt_bn_bwd_apply: z is fp32 (z_dtype 1), or fp16 (2) on the bf16 path");
  if (M <= 0) return MVLT_OK;
  const dim3 grid(grid_for(M * (C / 4)));
  const float* dy = (const float*)dy_;
  const bf16* dyh = (const bf16*)dy_;
  const float* z = (const float*)z_;
  const _Float16* zh = (const _Float16*)z_;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  const bool wide = z_dtype == 2 && C % 8 == 0 && ldz % 8 == 0 && lddz % 8 == 0 && lddy % (dy_dtype == 0 ? 8 : 4) == 0 && al16(dy_) && al16(z_) && al16(dz_bf16) && al16(mean) &&
                    al16(rstd) && al16(gamma) && al16(s1) && al16(s2);
  const dim3 grid8(grid_for(M * (C / 8)));
  if (wide && dy_dtype == 0) MVLT_LAUNCH((bn_bwd_apply8_kernel<bf16>), grid8, dim3(NT), 0, (hipStream_t)stream, dyh, lddy, zh, ldz, mean, rstd, gamma, s1, s2, M, C, (bf16*)dz_bf16, lddz, g_beta, g_gamma);
  else if (wide) MVLT_LAUNCH((bn_bwd_apply8_kernel<float>), grid8, dim3(NT), 0, (hipStream_t)stream, dy, lddy, zh, ldz, mean, rstd, gamma, s1, s2, M, C, (bf16*)dz_bf16, lddz, g_beta, g_gamma);
  else if (z_dtype == 2 && dy_dtype == 0) MVLT_LAUNCH((bn_bwd_apply_kernel<bf16, bf16, _Float16>), grid, dim3(NT), 0, (hipStream_t)stream, dyh, lddy, zh, ldz, mean, rstd, gamma, s1, s2, M, C, (bf16*)dz_bf16, lddz, g_beta, g_gamma);
  else if (z_dtype == 2) MVLT_LAUNCH((bn_bwd_apply_kernel<bf16, float, _Float16>), grid, dim3(NT), 0, (hipStream_t)stream, dy, lddy, zh, ldz, mean, rstd, gamma, s1, s2, M, C, (bf16*)dz_bf16, lddz, g_beta, g_gamma);
  else if (op_dtype == 0 && dy_dtype == 0) MVLT_LAUNCH((bn_bwd_apply_kernel<bf16, bf16>), grid, dim3(NT), 0, (hipStream_t)stream, dyh, lddy, z, ldz, mean, rstd, gamma, s1, s2, M, C, (bf16*)dz_bf16, lddz, g_beta, g_gamma);
  else if (op_dtype == 0) MVLT_LAUNCH((bn_bwd_apply_kernel<bf16>), grid, dim3(NT), 0, (hipStream_t)stream, dy, lddy, z, ldz, mean, rstd, gamma, s1, s2, M, C, (bf16*)dz_bf16, lddz, g_beta, g_gamma);
  else if (dy_dtype == 0) MVLT_LAUNCH((bn_bwd_apply_kernel<float, bf16>), grid, dim3(NT), 0, (hipStream_t)stream, dyh, lddy, z, ldz, mean, rstd, gamma, s1, s2, M, C, (float*)dz_bf16, lddz, g_beta, g_gamma);
  else MVLT_LAUNCH((bn_bwd_apply_kernel<float>), grid, dim3(NT), 0, (hipStream_t)stream, dy, lddy, z, ldz, mean, rstd, gamma, s1, s2, M, C, (float*)dz_bf16, lddz, g_beta, g_gamma);
  return mvlt_check_launch("mvlt_bn_bwd_apply");
}

extern "C" int mvlt_ew_mul(float* out, int ldo, const void* a_, int lda, const void* b_, int ldb, const void* c_, int ldc, int in_dtype, long M, int C,
                           int accumulate, void* out_bf16, int ld16, int op_dtype, void* stream) {
  const float* a = (const float*)a_; const float* b = (const float*)b_; const float* c = (const float*)c_;
  MVLT_REQUIRE(in_dtype == 1 || (in_dtype == 2 && op_dtype == 0), "mvlt_ew_mul: the factors are fp32 (in_dtype 1), or fp16 (2) on the bf16 path");
  MVLT_REQUIRE((out || out_bf16) && a && b && C % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && (!c || ldc % 4 == 0) && (!out || ldo % 4 == 0) &&
               (!out_bf16 || ld16 % 4 == 0), "mvlt_ew_mul: bad arguments");
  MVLT_REQUIRE(!accumulate || out, "mvlt_ew_mul: accumulate needs the fp32 output");
  if (M <= 0) return MVLT_OK;
  if (in_dtype == 2) MVLT_LAUNCH((ew_mul_kernel<bf16, _Float16>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, out, ldo, (const _Float16*)a_, lda, (const _Float16*)b_, ldb, (const _Float16*)c_, ldc, M, C, accumulate, (bf16*)out_bf16, ld16);
  else if (op_dtype == 0) MVLT_LAUNCH((ew_mul_kernel<bf16>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, out, ldo, a, lda, b, ldb, c, ldc, M, C, accumulate, (bf16*)out_bf16, ld16);
  else MVLT_LAUNCH((ew_mul_kernel<float>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, out, ldo, a, lda, b, ldb, c, ldc, M, C, accumulate, (float*)out_bf16, ld16);
  return mvlt_check_launch("mvlt_ew_mul");
}

extern "C" int mvlt_ew_mul3_bwd(const void* dy, int lddy, const void* a_, const void* b_, const void* c_, int ld, int in_dtype, void* da, void* db, void* dc,
                                long M, int C, int dy_dtype, void* stream) {
  const float* a = (const float*)a_; const float* b = (const float*)b_; const float* c = (const float*)c_;
  MVLT_REQUIRE(in_dtype == 1 || (in_dtype == 2 && dy_dtype == 0), "mvlt_ew_mul3_bwd: the factors are fp32 (in_dtype 1), or fp16 (2) beside a bf16 dy");
  /* da / db / dc take dy's dtype */
  MVLT_REQUIRE(dy && a && b && c && da && db && dc && C % 4 == 0 && lddy % 4 == 0 && ld % 4 == 0 && ld >= C && (dy_dtype == 0 || dy_dtype == 1), "mvlt_ew_mul3_bwd: bad arguments");
  if (M <= 0) return MVLT_OK;
  if (in_dtype == 2) MVLT_LAUNCH((ew_mul3_bwd_kernel<bf16, bf16, _Float16>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, (const bf16*)dy, lddy, (const _Float16*)a_, (const _Float16*)b_, (const _Float16*)c_, ld, (bf16*)da, (bf16*)db, (bf16*)dc, M, C);
  else if (dy_dtype == 0) MVLT_LAUNCH((ew_mul3_bwd_kernel<bf16, bf16>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, (const bf16*)dy, lddy, a, b, c, ld, (bf16*)da, (bf16*)db, (bf16*)dc, M, C);
  else MVLT_LAUNCH((ew_mul3_bwd_kernel<float, float>), dim3(grid_for(M * (C / 4))), dim3(NT), 0, (hipStream_t)stream, (const float*)dy, lddy, a, b, c, ld, (float*)da, (float*)db, (float*)dc, M, C);
  return mvlt_check_launch("mvlt_ew_mul3_bwd");
}

extern "C" int mvlt_upsample_fwd(const float* x, int ldx, int B, int H, int W, int C, int scale, void* out, int ldo, int out_dtype, int nchw, void* stream) {
  MVLT_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && scale >= 1, "mvlt_upsample_fwd: bad arguments");
  MVLT_REQUIRE(!nchw || out_dtype == 1, "mvlt_upsample_fwd: NCHW output is fp32");
  long total = (long)B * H * scale * W * scale * C;
  if (nchw && (long)B * C * H * scale < (1L << 31) && W <= 64 && (W * scale) % 4 == 0 && W * scale <= 256 && ((uintptr_t)out & 15) == 0) {
    MVLT_LAUNCH(upsample_fwd_nchw4_kernel, dim3((unsigned)((B * C * H * scale + 7) / 8)), dim3(NT), 0, (hipStream_t)stream, x, ldx, H, W, C, scale, (float*)out,
                       B * C * H * scale);
    return mvlt_check_launch("mvlt_upsample_fwd");
  }
  if (nchw && (long)B * C * H * scale < (1L << 31)) {
    MVLT_LAUNCH(upsample_fwd_nchw_kernel, dim3((unsigned)((B * C * H * scale + 7) / 8)), dim3(NT), 0, (hipStream_t)stream, x, ldx, H, W, C, scale, (float*)out,
                       B * C * H * scale);
    return mvlt_check_launch("mvlt_upsample_fwd");
  }
  if (!nchw && C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 7) == 0) {
    const int chunks = (W * scale * (C / 4) + NT - 1) / NT;
    const unsigned grid = (unsigned)((long)B * H * scale * chunks);
    if (out_dtype == 0) MVLT_LAUNCH((upsample_fwd_row_kernel<bf16>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x, ldx, H, W, C, scale, (bf16*)out, ldo, chunks);
    else MVLT_LAUNCH((upsample_fwd_row_kernel<float>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, x, ldx, H, W, C, scale, (float*)out, ldo, chunks);
    return mvlt_check_launch("mvlt_upsample_fwd");
  }
  if (out_dtype == 0) MVLT_LAUNCH((upsample_fwd_kernel<bf16>), dim3(grid_for(total, 16384)), dim3(NT), 0, (hipStream_t)stream, x, ldx, B, H, W, C, scale, (bf16*)out, ldo, nchw);
  else MVLT_LAUNCH((upsample_fwd_kernel<float>), dim3(grid_for(total, 16384)), dim3(NT), 0, (hipStream_t)stream, x, ldx, B, H, W, C, scale, (float*)out, ldo, nchw);
  return mvlt_check_launch("mvlt_upsample_fwd");
}

extern "C" int mvlt_upsample_bwd(const void* dy_, int lddy, int nchw, int B, int H, int W, int C, int scale, void* dx_, int lddx, int accumulate, int dx_dtype,
                                 int dy_dtype, void* stream) {
  MVLT_REQUIRE(dy_ && dx_ && B > 0 && H > 0 && W > 0 && C > 0 && scale >= 1 && (dy_dtype == 0 || dy_dtype == 1), "mvlt_upsample_bwd: bad arguments");
  const float* dy = (const float*)dy_;
  if (dy_dtype == 0) {
    // bf16 dy: the pixel-major x2 resizes inside the decoder (their dy is a conv input gradient in the operand dtype)
    MVLT_REQUIRE(!nchw && dx_dtype == 1 && C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)dy_ & 7) == 0 && ((uintptr_t)dx_ & 15) == 0,
                 "mvlt_upsample_bwd: bf16 dy needs the pixel-major layout, fp32 dx, C / ld multiples of 4");
    const int chunks = (W * (C / 4) + NT - 1) / NT;
    if (scale == 2) MVLT_LAUNCH((upsample_bwd_row_s_kernel<bf16, 2>), dim3((unsigned)((long)B * H * chunks)), dim3(NT), 0, (hipStream_t)stream, (const bf16*)dy_, lddy, H, W,
                                       C, (float*)dx_, lddx, accumulate, chunks);
    else MVLT_LAUNCH(upsample_bwd_row_kernel<bf16>, dim3((unsigned)((long)B * H * chunks)), dim3(NT), 0, (hipStream_t)stream, (const bf16*)dy_, lddy, H, W, C, scale,
                            (float*)dx_, lddx, accumulate, chunks);
    return mvlt_check_launch("mvlt_upsample_bwd");
  }
  MVLT_REQUIRE(dx_dtype == 1 || (dx_dtype == 0 && nchw), "mvlt_upsample_bwd: bf16 dx only behind the NCHW (final x8) upsample");
  float* dx = (float*)dx_;
  long total = (long)B * H * W * C;
  if (nchw && (size_t)W * scale * sizeof(float) <= 64 * 1024) {
    const dim3 grid((unsigned)(B * C * H));
    const size_t lds = (size_t)W * scale * sizeof(float);
    const bool v4 = (W * scale) % 4 == 0 && W * scale <= 256 && ((uintptr_t)dy & 15) == 0;
    if (dx_dtype == 0) {
      if (v4) MVLT_LAUNCH(upsample_bwd_nchw4_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)stream, dy, H, W, C, scale, (bf16*)dx_, lddx, accumulate);
      else MVLT_LAUNCH(upsample_bwd_nchw_kernel<bf16>, grid, dim3(NT), lds, (hipStream_t)stream, dy, H, W, C, scale, (bf16*)dx_, lddx, accumulate);
    } else {
      if (v4) MVLT_LAUNCH(upsample_bwd_nchw4_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, dy, H, W, C, scale, dx, lddx, accumulate);
      else MVLT_LAUNCH(upsample_bwd_nchw_kernel<float>, grid, dim3(NT), lds, (hipStream_t)stream, dy, H, W, C, scale, dx, lddx, accumulate);
    }
    return mvlt_check_launch("mvlt_upsample_bwd");
  }
  MVLT_REQUIRE(dx_dtype == 1, "mvlt_upsample_bwd: bf16 dx needs the row-buffered NCHW path (W * scale * 4 <= 64 KB)");
  if (!nchw && C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0) {
    const int chunks = (W * (C / 4) + NT - 1) / NT;
    if (scale == 2) MVLT_LAUNCH((upsample_bwd_row_s_kernel<float, 2>), dim3((unsigned)((long)B * H * chunks)), dim3(NT), 0, (hipStream_t)stream, dy, lddy, H, W, C, dx, lddx,
                                       accumulate, chunks);
    else MVLT_LAUNCH(upsample_bwd_row_kernel<float>, dim3((unsigned)((long)B * H * chunks)), dim3(NT), 0, (hipStream_t)stream, dy, lddy, H, W, C, scale, dx, lddx,
                            accumulate, chunks);
    return mvlt_check_launch("mvlt_upsample_bwd");
  }
  MVLT_LAUNCH(upsample_bwd_kernel, dim3(grid_for(total, 16384)), dim3(NT), 0, (hipStream_t)stream, dy, lddy, nchw, B, H, W, C, scale, dx, lddx, accumulate);
  return mvlt_check_launch("mvlt_upsample_bwd");
}

extern "C" int mvlt_upsample_l1_fwd(const float* x, int ldx, int B, int H, int W, int C, int scale, const float* target, float* loss_sum, void* stream) {
  MVLT_REQUIRE(x && target && loss_sum && B > 0 && H > 0 && W > 0 && C > 0 && scale >= 1, "mvlt_upsample_l1_fwd: bad arguments");
  MVLT_REQUIRE(W <= 64 && (W * scale) % 4 == 0 && W * scale <= 256 && ((uintptr_t)target & 15) == 0 && (long)B * C * H * scale < (1L << 31),
               "mvlt_upsample_l1_fwd: needs W <= 64, W * scale a multiple of 4 and <= 256, 16-byte aligned target");
  const int nrows = B * C * H * scale;
  const int groups = (nrows + 7) / 8;
  MVLT_LAUNCH(upsample_l1_fwd_kernel, dim3((unsigned)(groups < 2048 ? groups : 2048)), dim3(NT), 0, (hipStream_t)stream, x, ldx, H, W, C, scale, target, nrows, loss_sum);
  return mvlt_check_launch("mvlt_upsample_l1_fwd");
}

extern "C" int mvlt_upsample_l1_bwd(const float* x, int ldx, int B, int H, int W, int C, int scale, const float* target, const float* gscale, void* dx, int lddx,
                                    int dx_dtype, void* stream) {
  MVLT_REQUIRE(x && target && gscale && dx && B > 0 && H > 0 && W > 0 && C > 0 && scale >= 1 && (dx_dtype == 0 || dx_dtype == 1), "mvlt_upsample_l1_bwd: bad arguments");
  MVLT_REQUIRE(W <= 64 && (W * scale) % 4 == 0 && W * scale <= 256 && ((uintptr_t)target & 15) == 0, "mvlt_upsample_l1_bwd: needs W <= 64, W * scale a multiple of 4 and <= 256");
  const float inv_n = 1.0f / ((float)B * C * H * scale * W * scale);
  const dim3 grid((unsigned)(B * C * H));
  if (dx_dtype == 0) MVLT_LAUNCH(upsample_l1_bwd_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)stream, x, ldx, H, W, C, scale, target, gscale, inv_n, (bf16*)dx, lddx);
  else MVLT_LAUNCH(upsample_l1_bwd_kernel<float>, grid, dim3(NT), 0, (hipStream_t)stream, x, ldx, H, W, C, scale, target, gscale, inv_n, (float*)dx, lddx);
  return mvlt_check_launch("mvlt_upsample_l1_bwd");
}
