// LayerNorm forward / backward for the MVLT token matrices (gfx950).  HBM-bound: every row is read once with
// 16-byte loads, kept in registers for the two-pass mean/variance, and written once.
// A row (C = 64..768 channels) is owned by a group of G = 8/16/32/64 lanes so that small C still fills the wave.
#include "common.h"
#include "../../include/mvlt_hip.h"

namespace {

constexpr int NT = 256;
constexpr int VN = 8;                  // elements per chunk
constexpr int MAXIT = 2;               // chunks per lane: C <= 64 * 2 * 8 = 1024

__device__ __forceinline__ RowMap rm0(const mvlt_rowmap& m) {
  RowMap r{};
  r.mode = 0; r.rows_per_batch = m.rows_per_batch; r.batch_stride = m.batch_stride; r.offset = m.offset;
  return r;
}

template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T> struct Vec;
template <> struct Vec<bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16* p, float* f) {
    bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  static __device__ __forceinline__ void load_s(const bf16* p, float* f) {        // streaming: a row read once by this launch
    bf16x8 v = ld_g<MVLT_NT_NORM_LD>((const bf16x8*)p);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  static __device__ __forceinline__ void store(bf16* p, const float* f) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    st_g<MVLT_NT_NORM>((bf16x8*)p, v);
  }
};
template <> struct Vec<float> {
  static constexpr int N = 8;            // same 8-element chunk as bf16 (two 16-B accesses) so mixed-dtype rows line up
  static __device__ __forceinline__ void load(const float* p, float* f) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void load_s(const float* p, float* f) {
    f32x4 a = ld_g<MVLT_NT_NORM_LD>((const f32x4*)p), b = ld_g<MVLT_NT_NORM_LD>((const f32x4*)(p + 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float* p, const float* f) {
    f32x4 a = {f[0], f[1], f[2], f[3]}, b = {f[4], f[5], f[6], f[7]};
    st_g<MVLT_NT_NORM>((f32x4*)p, a);
    st_g<MVLT_NT_NORM>((f32x4*)(p + 4), b);
  }
};

template <typename T, typename TY, int G>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(mvlt_layernorm_args p) {
  const int gl = threadIdx.x % G;                 // lane inside the row group
  const int grp = threadIdx.x / G;                // row group inside the block
  constexpr int GROUPS = NT / G;
  const int nchunk = p.C / VN;
  const RowMap xm = rm0(p.x_map), ym = rm0(p.y_map);
  const float inv_c = 1.0f / (float)p.C;
  // a lane owns the same columns in every row: keep their gamma / beta in registers
  float gam[MAXIT][VN], bet[MAXIT][VN];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    int c = gl + it * G;
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      gam[it][e] = c < nchunk ? p.gamma[c * VN + e] : 0.f;
      bet[it][e] = c < nchunk ? p.beta[c * VN + e] : 0.f;
    }
  }
  for (int row = blockIdx.x * GROUPS + grp; row < p.rows; row += gridDim.x * GROUPS) {
    const T* xr = (const T*)p.x + rowmap_base(xm, row) * p.ldx;
    TY* yr = (TY*)p.y + rowmap_base(ym, row) * p.ldy;
    float v[MAXIT][VN];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int c = gl + it * G;
      if (c < nchunk) {
        Vec<T>::load_s(xr + c * VN, v[it]);
#pragma unroll
        for (int e = 0; e < VN; ++e) s += v[it][e];
      }
    }
    const float mean = group_sum<G>(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int c = gl + it * G;
      if (c < nchunk) {
#pragma unroll
        for (int e = 0; e < VN; ++e) { float d = v[it][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(group_sum<G>(q) * inv_c + p.eps);
    if (gl == 0) {
      if (p.mean) p.mean[row] = mean;
      if (p.rstd) p.rstd[row] = rstd;
    }
    const float* addr = p.add ? p.add + (long)(row % p.add_rows) * p.C : nullptr;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      int c = gl + it * G;
      if (c < nchunk) {
        float o[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) {
          int col = c * VN + e;
          o[e] = (v[it][e] - mean) * rstd * gam[it][e] + bet[it][e];
          if (addr) o[e] += addr[col];
        }
        Vec<TY>::store(yr + c * VN, o);
      }
    }
  }
}

// The same forward with the row geometry fixed at compile time: G lanes x ITS chunks of 8 columns cover the row exactly
// (C == 8 G ITS: no idle lanes at C = 320 = 8 x 5 x 8, where the power-of-two group of 64 lanes left 24 idle), RU rows of a
// lane group are loaded before the first is reduced (twice the bytes in flight per lane), gamma / beta sit in LDS when ITS > 2,
// and the grid is sized for a few row passes per workgroup so that the per-workgroup prologue is paid once per several rows
// (one pass per workgroup at C = 512 spent as long fetching gamma / beta as normalising).
template <typename T, typename TY, int G, int ITS, int RU>
__global__ __launch_bounds__(NT) void ln_fwd_fixed_kernel(mvlt_layernorm_args p) {
  const int gl = threadIdx.x % G, grp = threadIdx.x / G;
  constexpr int GROUPS = NT / G;
  constexpr bool KEEP = ITS <= 2;                  // gamma / beta in registers
  const RowMap xm = rm0(p.x_map), ym = rm0(p.y_map);
  const float inv_c = 1.0f / (float)p.C;
  float gam[KEEP ? ITS : 1][VN], bet[KEEP ? ITS : 1][VN];
  __shared__ __attribute__((aligned(16))) float s_gb[KEEP ? 8 : 2 * G * ITS * VN];      // [gamma | beta] when they do not stay in registers
  if constexpr (KEEP) {
#pragma unroll
    for (int it = 0; it < ITS; ++it)
#pragma unroll
      for (int e = 0; e < VN; ++e) { gam[it][e] = p.gamma[(gl + it * G) * VN + e]; bet[it][e] = p.beta[(gl + it * G) * VN + e]; }
  } else {
    for (int i = threadIdx.x; i < G * ITS * VN; i += NT) { s_gb[i] = p.gamma[i]; s_gb[G * ITS * VN + i] = p.beta[i]; }
    __syncthreads();
  }
  const int stride = gridDim.x * GROUPS;
  for (int row0 = blockIdx.x * GROUPS + grp; row0 < p.rows; row0 += stride * RU) {
    float v[RU][ITS][VN];
    float s[RU];
    bool live[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const int row = row0 + u * stride;
      live[u] = row < p.rows;
      s[u] = 0.f;
      if (live[u]) {
        const T* xr = (const T*)p.x + rowmap_base(xm, row) * p.ldx;
#pragma unroll
        for (int it = 0; it < ITS; ++it) Vec<T>::load_s(xr + (gl + it * G) * VN, v[u][it]);
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      if (!live[u]) continue;                        // uniform over the lane group
      const int row = row0 + u * stride;
#pragma unroll
      for (int it = 0; it < ITS; ++it)
#pragma unroll
        for (int e = 0; e < VN; ++e) s[u] += v[u][it][e];
      const float mean = group_sum<G>(s[u]) * inv_c;
      float q = 0.f;
#pragma unroll
      for (int it = 0; it < ITS; ++it)
#pragma unroll
        for (int e = 0; e < VN; ++e) { float d = v[u][it][e] - mean; q += d * d; }
      const float rstd = rsqrtf(group_sum<G>(q) * inv_c + p.eps);
      if (gl == 0) {
        if (p.mean) p.mean[row] = mean;
        if (p.rstd) p.rstd[row] = rstd;
      }
      const float* addr = p.add ? p.add + (long)(row % p.add_rows) * p.C : nullptr;
      TY* yr = (TY*)p.y + rowmap_base(ym, row) * p.ldy;
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int c0 = (gl + it * G) * VN;
        float o[VN], ga[VN], be[VN];
        if constexpr (KEEP) {
#pragma unroll
          for (int e = 0; e < VN; ++e) { ga[e] = gam[KEEP ? it : 0][e]; be[e] = bet[KEEP ? it : 0][e]; }
        } else {
          Vec<float>::load(s_gb + c0, ga);
          Vec<float>::load(s_gb + G * ITS * VN + c0, be);
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = (v[u][it][e] - mean) * rstd * ga[e] + be[e];
        if (addr) {                                  // "+ pos_embed": two 16-byte loads instead of eight scalar ones
          float ad[VN];
          Vec<float>::load_s(addr + c0, ad);
#pragma unroll
          for (int e = 0; e < VN; ++e) o[e] += ad[e];
        }
        Vec<TY>::store(yr + c0, o);
        if (p.y2) {
#pragma unroll
          for (int e = 0; e < VN; ++e) v[u][it][e] = o[e];
        }
      }
      if (p.y2) {
        // chained second LayerNorm of the row just written (the first block's norm1 behind the patch / text embedding's LN + pos):
        // the output row is still in registers, so that block needs no LayerNorm launch (and no re-read of the fp32 row) of its own
        float s2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITS; ++it)
#pragma unroll
          for (int e = 0; e < VN; ++e) s2 += v[u][it][e];
        const float mean2 = group_sum<G>(s2) * inv_c;
        float q2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITS; ++it)
#pragma unroll
          for (int e = 0; e < VN; ++e) { float d = v[u][it][e] - mean2; q2 += d * d; }
        const float rstd2 = rsqrtf(group_sum<G>(q2) * inv_c + p.eps2);
        const long prow = rowmap_base(ym, row);
        if (gl == 0) { p.mean2[prow] = mean2; p.rstd2[prow] = rstd2; }
        bf16* y2r = (bf16*)p.y2 + prow * p.ldy;
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          const int c0 = (gl + it * G) * VN;
          float ga[VN], be[VN], o[VN];
          Vec<float>::load(p.gamma2 + c0, ga);
          Vec<float>::load(p.beta2 + c0, be);
#pragma unroll
          for (int e = 0; e < VN; ++e) o[e] = (v[u][it][e] - mean2) * rstd2 * ga[e] + be[e];
          Vec<bf16>::store(y2r + c0, o);
        }
      }
    }
  }
}

// The launch is one 1024-thread workgroup per CU (launch_bwd): four waves per SIMD, 128 registers each.  Left to itself hipcc aims the two-chunk
// variants at EIGHT waves per SIMD (64 registers) and spills 36 bytes per lane inside the row loop (rocprofv3 Scratch_Size, round 4: 98304 x 320 ran
// at 3.6-4.0 TB/s); the occupancy the launch can actually reach is stated instead.
#ifndef MVLT_LN_BWD_WAVES
#define MVLT_LN_BWD_WAVES(NT_) __attribute__((amdgpu_waves_per_eu((NT_) / 256, (NT_) / 256)))
#endif
// ITS = chunks per lane (1 when the lane group covers the row: C <= 8 G; the second slot of the arrays would only hold
// registers: 126 -> ~90 VGPRs, 4 -> 5 waves per SIMD on an HBM-bound kernel)
template <typename T, typename TX, typename TDX, int G, int ITS, int NT>
__global__ __launch_bounds__(NT) MVLT_LN_BWD_WAVES(NT) void ln_bwd_kernel(mvlt_layernorm_bwd_args p) {
  constexpr int GROUPS = NT / G, NW = NT / 64;
  // LDS: [NW][2][C] per-wave partial sums of dgamma / dbeta in [e][chunk] order, then (two chunks per lane) gamma [C].  The partials used to be LDS
  // atomics into one [2][C] block: sixteen waves adding to the same words cost 12 / 20 / 29 us per launch at C = 320 / 512 / 768 (round 4 ablation),
  // more than the rows themselves on the short launches.
  extern __shared__ __attribute__((aligned(16))) float s_part[];
  const int gl = threadIdx.x % G, grp = threadIdx.x / G;
  const int nchunk = p.C / VN;
  const RowMap dym = rm0(p.dy_map), xm = rm0(p.x_map), dxm = rm0(p.dx_map);
  const float inv_c = 1.0f / (float)p.C;
  float* const s_gam = s_part + (size_t)NW * 2 * p.C;
  float dg[ITS][VN], db[ITS][VN], gam[ITS == 1 ? 1 : 1][VN];
#pragma unroll
  for (int it = 0; it < ITS; ++it)
#pragma unroll
    for (int e = 0; e < VN; ++e) { dg[it][e] = 0.f; db[it][e] = 0.f; }
  if constexpr (ITS == 1) {
#pragma unroll
    for (int e = 0; e < VN; ++e) gam[0][e] = gl < nchunk ? p.gamma[gl * VN + e] : 0.f;
  } else {
    // two chunks per lane: gamma is re-read from LDS per row instead of living in 16 registers (with them the kernel needed more than the 128
    // registers of a 1024-thread workgroup and spilled inside the row loop)
    for (int i = threadIdx.x; i < p.C; i += NT) s_gam[i] = p.gamma[i];
    __syncthreads();
  }

  for (int row = blockIdx.x * GROUPS + grp; row < p.rows; row += gridDim.x * GROUPS) {
    const T* dyr = (const T*)p.dy + rowmap_base(dym, row) * p.lddy;
    const TX* xr = (const TX*)p.x + rowmap_base(xm, row) * p.ldx;
    TDX* dxr = (TDX*)p.dx + rowmap_base(dxm, row) * p.lddx;
    const float mean = p.mean[row], rstd = p.rstd[row];
    const float sc2 = p.dx2 ? p.dx2_scale[row / p.dx2_rows_per_scale] : 0.f;
    float g[ITS][VN], xh[ITS][VN];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      int c = gl + it * G;
      if (c < nchunk) {
        float dyv[VN], xv[VN], gm[VN];
        Vec<T>::load_s(dyr + c * VN, dyv);
        Vec<TX>::load_s(xr + c * VN, xv);
        if constexpr (ITS == 1) {
#pragma unroll
          for (int e = 0; e < VN; ++e) gm[e] = gam[0][e];
        } else {
          const f32x4 g0 = *(const f32x4*)(s_gam + c * VN), g1 = *(const f32x4*)(s_gam + c * VN + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { gm[e] = g0[e]; gm[4 + e] = g1[e]; }
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) {
          float h = (xv[e] - mean) * rstd;
          float gg = dyv[e] * gm[e];
          xh[it][e] = h; g[it][e] = gg;
          s1 += gg; s2 += gg * h;
          dg[it][e] += dyv[e] * h;
          db[it][e] += dyv[e];
        }
      }
    }
    s1 = group_sum<G>(s1) * inv_c;
    s2 = group_sum<G>(s2) * inv_c;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      int c = gl + it * G;
      if (c < nchunk) {
        float o[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) o[e] = rstd * (g[it][e] - s1 - xh[it][e] * s2);
        if (p.dx_accumulate) {
          float old[VN];
          Vec<TDX>::load_s(dxr + c * VN, old);
#pragma unroll
          for (int e = 0; e < VN; ++e) o[e] += old[e];
        }
        Vec<TDX>::store(dxr + c * VN, o);
        if (p.dx2) {
#pragma unroll
          for (int e = 0; e < VN; ++e) o[e] *= sc2;
          Vec<T>::store((T*)p.dx2 + (long)row * p.lddx2 + c * VN, o);
        }
      }
    }
  }
  if (p.dgamma) {
    // the 64 / G row groups of a wave hold partial sums of the same columns: combine them in registers first; lanes 0 .. G-1 then hold the wave's
    // sums of every column and park them in the wave's own LDS slice ([e][chunk]: the lanes of a store hit consecutive words)
#pragma unroll
    for (int off = G; off < 64; off <<= 1)
#pragma unroll
      for (int it = 0; it < ITS; ++it)
#pragma unroll
        for (int e = 0; e < VN; ++e) { dg[it][e] += __shfl_xor(dg[it][e], off); db[it][e] += __shfl_xor(db[it][e], off); }
    float* const mine = s_part + (size_t)(threadIdx.x >> 6) * 2 * p.C;
    if ((threadIdx.x & 63) < G) {
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        int c = gl + it * G;
        if (c < nchunk) {
#pragma unroll
          for (int e = 0; e < VN; ++e) {
            mine[e * nchunk + c] = dg[it][e];
            mine[p.C + e * nchunk + c] = db[it][e];
          }
        }
      }
    }
    __syncthreads();
    const long cp = p.dg_copies > 1 ? (long)(blockIdx.x % p.dg_copies) * p.dg_copy_stride : 0;
    const bool own = p.dg_copies >= (int)gridDim.x;
    // a copy per workgroup (own): no other workgroup of this launch touches these floats and launches are stream-ordered, so the sums are added
    // without atomics; mvlt_fold_copies sums the copies afterwards
    for (int i = threadIdx.x; i < 2 * p.C; i += NT) {
      const int col = i < p.C ? i : i - p.C;
      const int si = (i < p.C ? 0 : p.C) + (col % VN) * nchunk + col / VN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += s_part[(size_t)w * 2 * p.C + si];
      float* dst = (i < p.C ? p.dgamma : p.dbeta) + cp + col;
      if (own) *dst += t; else atomicAdd(dst, t);
    }
  }
}

__global__ __launch_bounds__(NT) void fold_copies_kernel(float* arena, int copies, long stride, const int* dst_index, int j0, int j1, float* dst) {
  const int j = j0 + blockIdx.x * NT + threadIdx.x;
  if (j >= j1) return;
  float t = 0.f;
  for (int k = 0; k < copies; ++k) { t += arena[(long)k * stride + j]; arena[(long)k * stride + j] = 0.f; }
  dst[dst_index[j]] += t;
}

// the same for many copies (one per workgroup of the LayerNorm backward launches): a workgroup owns 64 slots, its 16 waves take every 16th copy
__global__ __launch_bounds__(1024) void fold_copies_wide_kernel(float* arena, int copies, long stride, const int* dst_index, int j0, int j1, float* dst) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
  const int j = j0 + blockIdx.x * 64 + lane;
  float t = 0.f;
  if (j < j1) {
    // sixteen copies per thread and pass, all requested before the first is used (a load / test / store chain per copy was sixteen round trips)
    for (int k0 = kq; k0 < copies; k0 += 256) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int k = k0 + 16 * i; v[i] = k < copies ? arena[(long)k * stride + j] : 0.f; }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        t += v[i];
        if (v[i] != 0.f) arena[(long)(k0 + 16 * i) * stride + j] = 0.f;    // (most copies of a short launch were never written: read-only then)
      }
    }
  }
  part[kq][lane] = t;
  __syncthreads();
  if (kq == 0 && j < j1) {
#pragma unroll
    for (int k = 1; k < 16; ++k) t += part[k][lane];
    dst[dst_index[j]] += t;
  }
}

// out[r, c] = sum_b in[b*batch_stride_rows + r][c]   (gradient of a broadcast "+ pos_embed"); fp32 out
// A workgroup owns 16 (row, 8-column chunk) items and splits the batch over its 16 thread groups (the first version
// gave every item to ONE thread looping over the whole batch: 132 workgroups at stage 1, latency-bound at 2.6x the
// HBM time); partial sums meet in LDS.
template <typename T>
__global__ __launch_bounds__(NT) void batch_sum_kernel(const T* in, float* out, int B, int R, int C, long batch_stride, int ld, float* acc2, int split) {
  __shared__ float part[16][16][VN + 1];
  const int nchunk = C / VN;
  const long total = (long)R * nchunk;
  const int item = threadIdx.x & 15, bs = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + item;
  float acc[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) acc[e] = 0.f;
  int r = 0, c = 0;
  if (i < total) {
    r = (int)(i / nchunk); c = (int)(i - (long)r * nchunk);
    for (int b = bs; b < B; b += 16) {
      float v[VN];
      Vec<T>::load_s(in + ((long)b * batch_stride + r) * ld + c * VN, v);
#pragma unroll
      for (int e = 0; e < VN; ++e) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < VN; ++e) part[bs][item][e] = acc[e];
  __syncthreads();
  if (bs == 0 && i < total) {
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += part[k][item][e];
      if (acc2 && r >= split) acc2[(long)(r - split) * C + c * VN + e] += t;      // rows from `split` on are ADDED to a second destination (text_pos_embed's gradient)
      else out[(long)r * C + c * VN + e] = t;
    }
  }
}

int pick_group(int C) {
  int chunks = C / VN;
  int g = 8;
  while (g < 64 && g * 2 <= chunks) g *= 2;      // largest power of two <= chunks (>= 8)
  while (g * MAXIT < chunks && g < 64) g *= 2;
  return g;
}

template <typename T, typename TY, int G, int ITS, int RU> void launch_fwd_fixed(const mvlt_layernorm_args& a, hipStream_t s) {
  constexpr int GROUPS = NT / G;
  static const int cap = getenv("MVLT_LN_GRID") ? atoi(getenv("MVLT_LN_GRID")) : 8192;
  long grid = ((long)a.rows + GROUPS * RU - 1) / (GROUPS * RU);
  if (grid > cap) grid = cap;
  MVLT_LAUNCH((ln_fwd_fixed_kernel<T, TY, G, ITS, RU>), dim3((unsigned)grid), dim3(NT), 0, s, a);
}

template <typename T, typename TY> int launch_fwd(const mvlt_layernorm_args& a_in, hipStream_t s) {
  // the widths of this model get an exact (lanes, chunks) split; everything else takes the power-of-two group kernel
  static const bool fixed_ok = !getenv("MVLT_LN_GENERIC");
  const bool fixed_c = a_in.C == 64 || a_in.C == 128 || a_in.C == 320 || a_in.C == 512 || a_in.C == 768;
  const mvlt_layernorm_args& a = a_in;
  MVLT_REQUIRE(!a.y2 || (fixed_ok && fixed_c), "mvlt_layernorm_fwd: the chained second LayerNorm (y2) needs C in {64, 128, 320, 512, 768}, got %d", a.C);
  if (fixed_ok) {
    switch (a.C) {
      case 64: launch_fwd_fixed<T, TY, 8, 1, 2>(a, s); return mvlt_check_launch("mvlt_layernorm_fwd");
      case 128: launch_fwd_fixed<T, TY, 16, 1, 2>(a, s); return mvlt_check_launch("mvlt_layernorm_fwd");
      case 320: launch_fwd_fixed<T, TY, 8, 5, 1>(a, s); return mvlt_check_launch("mvlt_layernorm_fwd");
      case 512: launch_fwd_fixed<T, TY, 16, 4, 1>(a, s); return mvlt_check_launch("mvlt_layernorm_fwd");
      case 768: launch_fwd_fixed<T, TY, 32, 3, 1>(a, s); return mvlt_check_launch("mvlt_layernorm_fwd");
      default: break;
    }
  }
  int g = pick_group(a.C);
  MVLT_REQUIRE(g * MAXIT * VN >= a.C, "mvlt_layernorm_fwd: C=%d too large", a.C);
  int groups = NT / g;
  int grid = (a.rows + groups - 1) / groups;
  if (grid > 8192) grid = 8192;
  switch (g) {
    case 8: MVLT_LAUNCH((ln_fwd_kernel<T, TY, 8>), dim3(grid), dim3(NT), 0, s, a); break;
    case 16: MVLT_LAUNCH((ln_fwd_kernel<T, TY, 16>), dim3(grid), dim3(NT), 0, s, a); break;
    case 32: MVLT_LAUNCH((ln_fwd_kernel<T, TY, 32>), dim3(grid), dim3(NT), 0, s, a); break;
    default: MVLT_LAUNCH((ln_fwd_kernel<T, TY, 64>), dim3(grid), dim3(NT), 0, s, a); break;
  }
  return mvlt_check_launch("mvlt_layernorm_fwd");
}

template <typename T, typename TX, typename TDX> int launch_bwd(const mvlt_layernorm_bwd_args& a_in, hipStream_t s) {
  const mvlt_layernorm_bwd_args& a = a_in;
  int g = pick_group(a.C);
  MVLT_REQUIRE(g * MAXIT * VN >= a.C, "mvlt_layernorm_bwd: C=%d too large", a.C);
  MVLT_REQUIRE((size_t)(1024 / 64 * 2 + 1) * a.C * sizeof(float) <= 160 * 1024, "mvlt_layernorm_bwd: C=%d needs more than 160 KB of LDS for the per-wave slices", a.C);
  // every workgroup ends with 2*C atomics on the same few cache lines: the rows are spread over few, large workgroups (1024
  // threads, one per CU).  (Storing the partial rows plainly and adding them in a second launch was measured: same time.)  Same time as 1024 x 256 threads
  // on most shapes, 123 -> 96 us at 98304 x 320 (fp32 x, dx +=); more workgroups of either size are slower (1536 x 256: +10 %)
  // only the two instantiated workgroup sizes: the per-wave LDS slices below are sized from this value (512 used to size 8 slices for a 16-wave launch)
  static const int nt = (getenv("MVLT_LN_BWD_NT") && atoi(getenv("MVLT_LN_BWD_NT")) == 256) ? 256 : 1024;
  static const int bcap = getenv("MVLT_LN_BWD_GRID") ? atoi(getenv("MVLT_LN_BWD_GRID")) : 256;
  int groups = nt / g;
  int grid = (a.rows + groups - 1) / groups;
  if (grid > bcap) grid = bcap;
  if (a.dgamma && a.dg_copies >= 64 && grid > a.dg_copies) grid = a.dg_copies;     // a copy per workgroup: plain adds (see the kernel's tail)
  const bool one = g * VN >= a.C;               // one chunk per lane covers the row
  size_t lds = ((size_t)(nt / 64) * 2 + (one ? 0 : 1)) * a.C * sizeof(float);      // per-wave dgamma / dbeta slices (+ gamma): 101 KB at C = 768
  // the dynamic-LDS ceiling is raised ONCE per instantiation (160 KB: whatever C a later call brings) and its result checked
#define MVLT_LN_BWD_K(KERN_, NT_)                                                                                    \
  do {                                                                                                               \
    static const hipError_t attr = hipFuncSetAttribute((const void*)KERN_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    MVLT_REQUIRE(lds <= 65536 || attr == hipSuccess, "mvlt_layernorm_bwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr)); \
    MVLT_LAUNCH(KERN_, dim3(grid), dim3(NT_), lds, s, a);                                                     \
  } while (0)
#define MVLT_LN_BWD_N(G_, NT_)                                                                                       \
  do {                                                                                                               \
    if (one) MVLT_LN_BWD_K((ln_bwd_kernel<T, TX, TDX, G_, 1, NT_>), NT_);                                            \
    else MVLT_LN_BWD_K((ln_bwd_kernel<T, TX, TDX, G_, MAXIT, NT_>), NT_);                                            \
  } while (0)
#define MVLT_LN_BWD(G_)                                                                                              \
  do {                                                                                                               \
    if (nt == 256) MVLT_LN_BWD_N(G_, 256);                                                                           \
    else MVLT_LN_BWD_N(G_, 1024);                                                                                    \
  } while (0)
  switch (g) {
    case 8: MVLT_LN_BWD(8); break;
    case 16: MVLT_LN_BWD(16); break;
    case 32: MVLT_LN_BWD(32); break;
    default: MVLT_LN_BWD(64); break;
  }
#undef MVLT_LN_BWD
#undef MVLT_LN_BWD_N
#undef MVLT_LN_BWD_K
  return mvlt_check_launch("mvlt_layernorm_bwd");
}

}  // namespace

extern "C" int mvlt_layernorm_fwd(const mvlt_layernorm_args* a, void* stream) {
  MVLT_REQUIRE(a && a->x && a->y && a->gamma && a->beta, "mvlt_layernorm_fwd: null pointer");
  MVLT_REQUIRE((a->dtype == 0 || a->dtype == 1) && (a->y_dtype == 0 || a->y_dtype == 1), "mvlt_layernorm_fwd: bad dtype");
  MVLT_REQUIRE(a->C > 0 && a->C % 8 == 0 && a->ldx % 8 == 0 && a->ldy % 8 == 0, "mvlt_layernorm_fwd: C/ldx/ldy must be multiples of 8");
  MVLT_REQUIRE(a->x_map.mode == 0 && a->y_map.mode == 0, "mvlt_layernorm_fwd: only mode-0 row maps");
  MVLT_REQUIRE(!a->add || a->add_rows > 0, "mvlt_layernorm_fwd: add needs add_rows");
  if (a->rows <= 0) return MVLT_OK;
  hipStream_t s = (hipStream_t)stream;
  if (a->dtype == 0) return a->y_dtype == 0 ? launch_fwd<bf16, bf16>(*a, s) : launch_fwd<bf16, float>(*a, s);
  return a->y_dtype == 0 ? launch_fwd<float, bf16>(*a, s) : launch_fwd<float, float>(*a, s);
}

extern "C" int mvlt_fold_copies(float* arena, int copies, long stride, const int* dst_index, int j0, int j1, float* dst, void* stream) {
  MVLT_REQUIRE(arena && dst_index && dst && copies >= 1 && j0 >= 0 && j1 >= j0 && stride >= j1, "mvlt_fold_copies: bad arguments");
  if (j1 == j0) return MVLT_OK;
  if (copies >= 32) MVLT_LAUNCH(fold_copies_wide_kernel, dim3((j1 - j0 + 63) / 64), dim3(1024), 0, (hipStream_t)stream, arena, copies, stride, dst_index, j0, j1, dst);
  else MVLT_LAUNCH(fold_copies_kernel, dim3((j1 - j0 + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, arena, copies, stride, dst_index, j0, j1, dst);
  return mvlt_check_launch("mvlt_fold_copies");
}

extern "C" int mvlt_layernorm_bwd(const mvlt_layernorm_bwd_args* a, void* stream) {
  MVLT_REQUIRE(a && a->dy && a->x && a->dx && a->gamma && a->mean && a->rstd, "mvlt_layernorm_bwd: null pointer");
  MVLT_REQUIRE((a->dgamma == nullptr) == (a->dbeta == nullptr), "mvlt_layernorm_bwd: dgamma and dbeta go together");
  MVLT_REQUIRE((a->dtype == 0 || a->dtype == 1) && (a->x_dtype == 0 || a->x_dtype == 1) && (a->dx_dtype == 0 || a->dx_dtype == 1),
               "mvlt_layernorm_bwd: bad dtype");
  MVLT_REQUIRE(a->C > 0 && a->C % 8 == 0 && a->ldx % 8 == 0 && a->lddy % 8 == 0 && a->lddx % 8 == 0, "mvlt_layernorm_bwd: C/ld* must be multiples of 8");
  MVLT_REQUIRE(a->x_map.mode == 0 && a->dy_map.mode == 0 && a->dx_map.mode == 0, "mvlt_layernorm_bwd: only mode-0 row maps");
  MVLT_REQUIRE(a->dg_copies <= 1 || (a->dgamma && a->dg_copy_stride >= a->C), "mvlt_layernorm_bwd: dg_copies needs dgamma / dbeta and dg_copy_stride >= C");
  MVLT_REQUIRE(!a->dx2 || (a->dx2_scale && a->dx2_rows_per_scale > 0 && a->lddx2 % 8 == 0 && a->lddx2 >= a->C),
               "mvlt_layernorm_bwd: dx2 needs dx2_scale, dx2_rows_per_scale > 0 and lddx2 (multiple of 8) >= C");
  if (a->rows <= 0) return MVLT_OK;
  hipStream_t s = (hipStream_t)stream;
  const int key = a->dtype * 4 + a->x_dtype * 2 + a->dx_dtype;
  switch (key) {
    case 0: return launch_bwd<bf16, bf16, bf16>(*a, s);
    case 1: return launch_bwd<bf16, bf16, float>(*a, s);
    case 2: return launch_bwd<bf16, float, bf16>(*a, s);
    case 3: return launch_bwd<bf16, float, float>(*a, s);
    case 4: return launch_bwd<float, bf16, bf16>(*a, s);
    case 5: return launch_bwd<float, bf16, float>(*a, s);
    case 6: return launch_bwd<float, float, bf16>(*a, s);
    default: return launch_bwd<float, float, float>(*a, s);
  }
}

extern "C" int mvlt_batch_sum(const void* in, float* out, int B, int R, int C, long batch_stride_rows, int ld, int dtype, float* acc2, int split, void* stream) {
  MVLT_REQUIRE(in && out && B > 0 && R >= 0 && C > 0 && (!acc2 || (split >= 0 && split <= R)), "mvlt_batch_sum: bad arguments");
  MVLT_REQUIRE(C % 8 == 0 && ld % 8 == 0, "mvlt_batch_sum: C/ld must be multiples of 8");
  if (R == 0) return MVLT_OK;
  long total = (long)R * (C / 8);
  int grid = (int)((total + 15) / 16);
  if (dtype == 0) MVLT_LAUNCH((batch_sum_kernel<bf16>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, (const bf16*)in, out, B, R, C, batch_stride_rows, ld, acc2, split);
  else MVLT_LAUNCH((batch_sum_kernel<float>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, (const float*)in, out, B, R, C, batch_stride_rows, ld, acc2, split);
  return mvlt_check_launch("mvlt_batch_sum");
}
