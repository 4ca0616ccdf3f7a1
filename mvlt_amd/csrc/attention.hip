// Spatial-reduction attention core for MVLT (gfx950): O = softmax(Q K^T * scale) V, head_dim 64, M <= 320 keys.
// Replaces reference libs/pvlt.py:113-117 (two bmm + softmax that materialise the (B,h,N,M) score tensor).
//
// PVT's spatial reduction keeps M = (H/r)^2 + T small (192 at 256 px, 272 at 384 px) in every stage, so the whole
// K (M x 64) and V^T (64 x M) of one (batch, head) live in LDS for the lifetime of a workgroup and the softmax is
// single-pass: a wave owns 32 queries, computes S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (so each lane holds the
// scores of ONE query: the row reduction is lane-local plus one cross-half shuffle), exponentiates in registers,
// and feeds P^T straight back as the B operand of O^T = V^T P^T.  Nothing but Q, K, V, O and the log-sum-exp
// touches HBM.  fp32 instantiation (exact-f32 MFMA 32x32x2) serves the 1e-3 parity bar.
#include "common.h"
#include <type_traits>
#include "../../include/mvlt_hip.h"

namespace {

constexpr int NT = 256;
constexpr int HD = 64;

template <typename T> struct Frag;                 // 8 consecutive-K elements of one MFMA operand row
template <> struct Frag<bf16> { bf16x8 v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ void mma32(f32x16& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ Frag<T> load_frag8(const T* p);      // 8 contiguous elements
template <> __device__ __forceinline__ Frag<bf16> load_frag8<bf16>(const bf16* p) {
  Frag<bf16> f; f.v = *(const bf16x8*)p; return f;
}
template <> __device__ __forceinline__ Frag<float> load_frag8<float>(const float* p) {
  Frag<float> f;
  f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
  return f;
}
template <typename T> __device__ __forceinline__ Frag<T> zero_frag() {
  Frag<T> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = (T)0.f;
  return f;
}
// 4 + 4 contiguous elements from two places (the V^T / transposed-tile fragments)
template <typename T> __device__ __forceinline__ Frag<T> load_frag44(const T* p0, const T* p1);
template <> __device__ __forceinline__ Frag<bf16> load_frag44<bf16>(const bf16* p0, const bf16* p1) {
  bf16x4 a = *(const bf16x4*)p0, b = *(const bf16x4*)p1;
  Frag<bf16> f;
  f.v = bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return f;
}
template <> __device__ __forceinline__ Frag<float> load_frag44<float>(const float* p0, const float* p1) {
  f32x4 a = *(const f32x4*)p0, b = *(const f32x4*)p1;
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
  return f;
}

template <typename T> struct Lds {
  static constexpr int PC = 16 / sizeof(T);             // elements per 16-B chunk
  static constexpr int NCH = HD / PC;                   // chunks per 64-wide row (8 / 16)
  static __device__ __forceinline__ int swz(int row, int chunk) {
    // fp32: an 8-element fragment spans two adjacent chunks, so only even XOR masks keep it contiguous
    return (sizeof(T) == 2) ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row & 7) << 1));
  }
  // element offset of (row, d) in a [rows][64] row-major swizzled tile; d must be a multiple of PC
  static __device__ __forceinline__ int off(int row, int d) { return row * HD + swz(row, d / PC) * PC; }
};

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int NKT>
__global__ __launch_bounds__(NT, 2) void attn_fwd_kernel(mvlt_attn_args p, int nq_chunks, int q_per_wg) {
  constexpr int MP = NKT * 32;                // padded key count
  constexpr int VS = MP + 4;                  // V^T row stride (elements): conflict-free 8-B / 16-B column reads
  constexpr int PC = Lds<T>::PC, NCH = Lds<T>::NCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = (T*)smem;                           // [MP][64] swizzled
  T* sVt = sK + MP * HD;                      // [64][VS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 5, l31 = lane & 31;
  // XCD-aware placement: all query chunks of one (batch, head) run on the same XCD (block id % 8) and share K/V in its L2
  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  const int gidx = xcd + 8 * (j / nq_chunks);
  const int chunk_id = j % nq_chunks;
  if (gidx >= p.B * p.H) return;
  const int b = gidx / p.H, h = gidx % p.H;

  const T* Qg = (const T*)p.Q + (long)b * p.N * p.ldq + h * HD;
  const T* Kg = (const T*)p.KV + (long)b * p.M * p.ldkv + p.k_off + h * HD;
  const T* Vg = (const T*)p.KV + (long)b * p.M * p.ldkv + p.v_off + h * HD;
  T* Og = (T*)p.O + (long)b * p.N * p.ldo + h * HD;

  // ---- stage K (swizzled rows) and V^T (transposed) in LDS; rows >= M are zero
  for (int u = tid; u < MP * NCH; u += NT) {
    int r = u / NCH, c = u % NCH;
    u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
    if (r < p.M) {
      kv = *(const u32x4*)(Kg + (long)r * p.ldkv + c * PC);
      vv = *(const u32x4*)(Vg + (long)r * p.ldkv + c * PC);
    }
    *(u32x4*)(sK + Lds<T>::off(r, c * PC)) = kv;
    T ve[PC];
    *(u32x4*)ve = vv;
#pragma unroll
    for (int e = 0; e < PC; ++e) sVt[(c * PC + e) * VS + r] = ve[e];
  }
  __syncthreads();

  const float sl2 = p.scale * 1.44269504088896340736f;
  const int q_begin = chunk_id * q_per_wg;
  const int q_end = min(p.N, q_begin + q_per_wg);
  // Keys are walked in NBLK blocks of <= BT tiles (32 keys each).  NKT <= 7 is ONE block: plain single-pass softmax.
  // Larger M (384-px inputs: 272 keys) uses two blocks merged with the usual running-max rescale, which keeps the
  // score accumulators at <= 80 registers instead of 144-160 (no spills, 2 waves per SIMD).
  constexpr int NBLK = (NKT <= 7) ? 1 : 2;
  constexpr int BT = (NKT + NBLK - 1) / NBLK;
  for (int q0 = q_begin + wave * 32; q0 < q_end; q0 += 4 * 32) {
    const int q = q0 + l31;
    const bool q_ok = q < q_end;
    Frag<T> qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = q_ok ? load_frag8<T>(Qg + (long)q * p.ldq + 16 * s + 8 * g) : zero_frag<T>();

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    }
    float m_run = -INFINITY, sum_loc = 0.f;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int t0 = blk * BT;
      f32x16 acc[BT];
      // K fragments one tile ahead of their MFMAs (two tiles' worth live): a tile's four LDS reads issued right in front of the MFMAs
      // that use them put one LDS latency in front of every tile, and both waves of a SIMD stall the same way
      Frag<T> kf[2][4];
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[0][s] = load_frag8<T>(sK + Lds<T>::off(t0 * 32 + l31, 16 * s + 8 * g));
#pragma unroll
      for (int ti = 0; ti < BT; ++ti) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
        if (ti + 1 < BT && t0 + ti + 1 < NKT) {
#pragma unroll
          for (int s = 0; s < 4; ++s) kf[(ti + 1) & 1][s] = load_frag8<T>(sK + Lds<T>::off((t0 + ti + 1) * 32 + l31, 16 * s + 8 * g));
        }
        if (t0 + ti < NKT) {
#pragma unroll
          for (int s = 0; s < 4; ++s) mma32(acc[ti], kf[ti & 1][s], qf[s]);          // rows = keys, cols = queries
        }
        __builtin_amdgcn_sched_barrier(0);      // keep only two tiles' K fragments live (register budget: 2 waves/SIMD)
      }
      // acc[ti][r] = S[q = l31][key = 32 (t0+ti) + (r&3) + 8 (r>>2) + 4 g]
      float mb = -INFINITY;
      if (p.M == MP && NKT % NBLK == 0) {         // no padded key anywhere (M = 192 at 256 px): no masking pass
#pragma unroll
        for (int ti = 0; ti < BT; ++ti)
#pragma unroll
          for (int r = 0; r < 16; ++r) mb = fmaxf(mb, acc[ti][r]);
      } else {
#pragma unroll
        for (int ti = 0; ti < BT; ++ti)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int key = 32 * (t0 + ti) + (r & 3) + 8 * (r >> 2) + 4 * g;
            float sv = (t0 + ti < NKT && key < p.M) ? acc[ti][r] : -INFINITY;
            acc[ti][r] = sv;
            mb = fmaxf(mb, sv);
          }
      }
      mb = fmaxf(mb, __shfl_xor(mb, 32));
      const float m_new = fmaxf(m_run, mb);     // finite: every block holds >= 1 valid key
      if (blk > 0) {
        const float alpha = exp2f((m_run - m_new) * sl2);
        sum_loc *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
      }
      m_run = m_new;
      // bf16: the bare v_exp_f32 (libm's exp2f wraps it in a denormal-range rescale, five more VALU per score; results below 2^-126
      // feed a bf16 P and a sum >= 1 either way); the fp32 parity path keeps libm's
      const float mneg = -m_new * sl2;
#pragma unroll
      for (int ti = 0; ti < BT; ++ti)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float e;
          if constexpr (sizeof(T) == 2) e = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[ti][r], sl2, mneg));
          else e = exp2f((acc[ti][r] - m_new) * sl2);
          acc[ti][r] = e;
          sum_loc += e;
        }
      // PV: step u = (key tile, 16-key half); the V^T fragments of step u + 1 are read while step u's MFMAs run
      Frag<T> vf[2][2];
      {
        const int kb = 32 * t0 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) { const T* vr = sVt + (32 * dt + l31) * VS + kb; vf[0][dt] = load_frag44<T>(vr, vr + 8); }
      }
#pragma unroll
      for (int u = 0; u < 2 * BT; ++u) {
        const int ti = u >> 1, s2 = u & 1;
        if (t0 + ti >= NKT) continue;
        if (u + 1 < 2 * BT && t0 + ((u + 1) >> 1) < NKT) {
          const int kb = 32 * (t0 + ((u + 1) >> 1)) + 16 * ((u + 1) & 1) + 4 * g;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) { const T* vr = sVt + (32 * dt + l31) * VS + kb; vf[(u + 1) & 1][dt] = load_frag44<T>(vr, vr + 8); }
        }
        Frag<T> pf;                             // B operand: P^T, k-slot (g, jj) <-> key 32kt + 16 s2 + 4g + 8 (jj>>2) + (jj&3)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) pf.v[jj] = (T)acc[ti][8 * s2 + jj];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) mma32(oacc[dt], vf[u & 1][dt], pf);            // rows = d, cols = queries
        __builtin_amdgcn_sched_barrier(0);      // do not hoist every V^T fragment above the first PV MFMA
      }
    }
    const float sum = sum_loc + __shfl_xor(sum_loc, 32);
    const float inv = 1.0f / sum;
    // oacc[dt][r] = O[q = l31][d = 32 dt + (r&3) + 8 (r>>2) + 4 g]
    if (q_ok) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          T o4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o4[e] = (T)(oacc[dt][4 * rq + e] * inv);
          T* dst = Og + (long)q * p.ldo + 32 * dt + 8 * rq + 4 * g;
          if constexpr (sizeof(T) == 2) *(u32x2*)dst = *(u32x2*)o4; else *(u32x4*)dst = *(u32x4*)o4;
        }
      if (g == 0 && p.lse) p.lse[((long)b * p.H + h) * p.N + q] = m_run * p.scale + logf(sum);
    }
  }
}

template <typename T, int NKT> int launch_fwd_n(const mvlt_attn_args& a, hipStream_t s) {
  constexpr int MP = NKT * 32;
  const size_t lds = (size_t)(MP * HD + HD * (MP + 4)) * sizeof(T);
  MVLT_REQUIRE(lds <= 160 * 1024, "mvlt_sr_attention_fwd: M=%d needs %zu B of LDS (>160 KB) in this dtype", a.M, lds);
  int q_per_wg = 256;
  if (a.N <= 512) q_per_wg = 128 * ((a.N + 127) / 128);
  const int nq = (a.N + q_per_wg - 1) / q_per_wg;
  const int groups = a.B * a.H;
  const int grid = 8 * ((groups + 7) / 8) * nq;
  mvlt_max_lds<(attn_fwd_kernel<T, NKT>)>();
  MVLT_LAUNCH((attn_fwd_kernel<T, NKT>), dim3(grid), dim3(NT), lds, s, a, nq, q_per_wg);
  return mvlt_check_launch("mvlt_sr_attention_fwd");
}

template <typename T> int launch_fwd(const mvlt_attn_args& a, hipStream_t s) {
  switch ((a.M + 31) / 32) {
    case 1: return launch_fwd_n<T, 1>(a, s);
    case 2: return launch_fwd_n<T, 2>(a, s);
    case 3: return launch_fwd_n<T, 3>(a, s);
    case 4: return launch_fwd_n<T, 4>(a, s);
    case 5: return launch_fwd_n<T, 5>(a, s);
    case 6: return launch_fwd_n<T, 6>(a, s);
    case 7: return launch_fwd_n<T, 7>(a, s);
    case 8: return launch_fwd_n<T, 8>(a, s);
    case 9: return launch_fwd_n<T, 9>(a, s);
    case 10: return launch_fwd_n<T, 10>(a, s);
    default: break;
  }
  mvlt_set_error("mvlt_sr_attention_fwd: M=%d keys exceeds the 320-key LDS-resident design", a.M);
  return MVLT_ERR_UNSUPPORTED;
}


// ------------------------------------------------------------------------------------------------ forward, round 3 (bf16)
// Same math as attn_fwd_kernel, re-organised around what tools/probes/valu_rates.hip measured on gfx950: a wave issues one VALU instruction
// per ~3.5 ns on its own, the SIMD retires one per ~1.55 ns (exp: 3.6 ns), a 32x32x16 MFMA occupies the matrix pipe for 16 ns, and the
// two only overlap when their instructions are interleaved in the stream.  A 32-query tile needs 48 MFMAs (0.77 us) and, at best, ~4
// VALU + 1 exp per score (96 scores per lane: ~0.85 us); the kernel above spends ~4 us per tile and SIMD (counters: 47 % of the wave
// cycles parked at waits, 27 % issuing).  Changes:
//  * the keys are cut into two blocks; the block-1 score MFMAs run under block 0's exponentials, block 0's P V MFMAs under block 1's, and
//    the NEXT tile's first score MFMAs under this tile's normalise-and-store (slots pinned with sched_barrier(0));
//  * the row maximum used as the exponent's reference is block 0's; block 1 only forces the usual rescale when its maximum is more than
//    2^24 above it (a wave-uniform branch that LayerNorm-ed inputs never take) -- no multiply pass over the O accumulators;
//  * a workgroup keeps K / V^T of its (batch, head) in LDS for a whole chunk of several hundred queries (a wave walks 4 - 9 tiles) and
//    fetches the next tile's Q rows while the current tile computes;
//  * V^T rows are stored with the keys of a 32-key tile permuted so that an MFMA A fragment is ONE 16-byte read (was two 8-byte reads,
//    29 % bank conflicts), rows padded to a stride of 16 B mod 256 B (conflict-free for the 16-lane groups of ds_read_b128);
//  * accumulators start from an inline zero (128 v_mov per tile in the kernel above), no masking code unless keys are padded.
template <int NKT, bool PADDED, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_fwd2_kernel(mvlt_attn_args p, int nq_chunks, int q_per_wg) {
  typedef bf16 T;
  constexpr int NTH = NW * 64;       // 4 waves while two workgroups fit a CU (M <= 192 keys), 8 waves in one workgroup per CU beyond
  constexpr int MP = NKT * 32;
  constexpr int BT0 = (NKT + 1) / 2, BT1 = NKT - BT0;
  constexpr int VPAD = 8;
  constexpr int VS = MP + VPAD;               // V^T row stride (elements): 64 NKT + 16 bytes = an odd multiple of 16 B, so 16 consecutive rows
                                              // at one column fall on 16 different 16-byte slots of the 256-byte bank window
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = (T*)smem;                           // [MP][64] swizzled
  T* sVt = sK + MP * HD;                      // [64][VS], keys permuted inside each 32-key tile
  char* sQ = (char*)(sVt + HD * VS);          // [NW waves][32 rows][128 B]: a wave's next Q tile, 16-B chunks XOR-swizzled by (row >> 1) & 7
  char* sO = sQ + NW * 4096;                  // [NW waves][16 rows][144 B]: O staging, half a tile at a time

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 5, l31 = lane & 31;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, j = bid >> 3;
  const int gidx = xcd + 8 * (j / nq_chunks);
  const int chunk_id = j % nq_chunks;
  if (gidx >= p.B * p.H) return;
  const int b = gidx / p.H, h = gidx % p.H;

  const T* Qg = (const T*)p.Q + (long)b * p.N * p.ldq + h * HD;
  const T* Kg = (const T*)p.KV + (long)b * p.M * p.ldkv + p.k_off + h * HD;
  const T* Vg = (const T*)p.KV + (long)b * p.M * p.ldkv + p.v_off + h * HD;
  T* Og = (T*)p.O + (long)b * p.N * p.ldo + h * HD;

  const int q_begin = chunk_id * q_per_wg;
  const int q_end = min(p.N, q_begin + q_per_wg);
  // Q rows and O rows go through LDS in full 128-byte lines: a row-per-lane access (each lane its own row, 16 or 8 bytes per instruction)
  // makes every wave instruction touch 64 different lines -- with the stores and loads in that form the kernel ran 107 us on the stage-1
  // shape, 68 us without the stores, 83 us with the loads served from cache (ablations, one box).  Q: four LDS-DMA instructions of eight
  // rows each per tile, issued one tile ahead into the wave's own buffer; O: through a 16-row staging tile, 16 bytes per lane, eight
  // lanes per row.
  char* const myQ = sQ + wave * 4096;
  char* const myO = sO + wave * 2304;
  const unsigned myQ_lds = (unsigned)(uintptr_t)myQ;
  const int qd_row = lane >> 3, qd_chunk = lane & 7;
  const char* zsrc = (const char*)g_zero_page + ((lane * 16 + wave * 1024) & 65535);
  auto dma_q = [&](int q0) {
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int row = 8 * k4 + qd_row, q = q0 + row;
      const char* src = (const char*)(Qg + (long)q * p.ldq) + ((qd_chunk ^ ((row >> 1) & 7)) << 4);
      glds16(q < q_end ? src : zsrc, __builtin_amdgcn_readfirstlane(myQ_lds + k4 * 1024));
    }
  };
  dma_q(q_begin + wave * 32);
  for (int u = tid; u < MP * 8; u += NTH) {
    const int r = u >> 3, c = u & 7;
    u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
    if (r < p.M) {
      kv = *(const u32x4*)(Kg + (long)r * p.ldkv + c * 8);
      vv = *(const u32x4*)(Vg + (long)r * p.ldkv + c * 8);
    }
    *(u32x4*)(sK + Lds<T>::off(r, c * 8)) = kv;
    // key r = 32 t + 16 s2 + 8 hi + 4 gg + lo  ->  column 32 t + 16 s2 + 8 gg + 4 hi + lo
    const int col = (r & ~15) | (((r >> 2) & 1) << 3) | (((r >> 3) & 1) << 2) | (r & 3);
    T ve[8];
    *(u32x4*)ve = vv;
#pragma unroll
    for (int e = 0; e < 8; ++e) sVt[(c * 8 + e) * VS + col] = ve[e];
  }
  __syncthreads();

  const float sl2 = p.scale * 1.44269504088896340736f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 accA[BT0], accB[BT1 > 0 ? BT1 : 1], oacc[2];
  bf16x8 qf[4];
  // Operand fragments of one 4-MFMA step (a key tile of K, or a key tile of V^T for both halves of d) are read ONE STEP AHEAD into a
  // two-deep register ring: a read issued right in front of its MFMA puts a full LDS latency in front of each of the 48 MFMAs of a tile.
  // A tile is 2 NKT steps (block-1 scores, block-0 P V, block-1 P V, next tile's block-0 scores), so the ring parity is static.
  bf16x8 fr[2][4];
  auto readK = [&](auto parc, int t) {
    constexpr int par = decltype(parc)::value;
#pragma unroll
    for (int s = 0; s < 4; ++s) fr[par][s] = *(const bf16x8*)(sK + Lds<T>::off(t * 32 + l31, 16 * s + 8 * g));
  };
  auto readV = [&](auto parc, int t) {
    constexpr int par = decltype(parc)::value;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) fr[par][2 * s2 + dt] = *(const bf16x8*)(sVt + (32 * dt + l31) * VS + 32 * t + 16 * s2 + 8 * g);
  };
  auto mmaS = [&](auto parc, f32x16& acc) {                       // acc = K tile . Q^T   (rows = keys, cols = queries)
    constexpr int par = decltype(parc)::value;
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[par][s], qf[s], s == 0 ? zero16 : acc, 0, 0, 0);
  };
  auto mmaPV = [&](auto parc, const f32x16& pa, bool first) {     // O^T += V^T[:, key tile] . P^T, P from the tile's score accumulators
    constexpr int par = decltype(parc)::value;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 pf;                             // B operand: P^T, k-slot (g, jj) <-> key 32 t + 16 s2 + 4 g + 8 (jj >> 2) + (jj & 3)
#pragma unroll
      for (int jj = 0; jj < 8; jj += 2) {
        const bf16x2 pr = __builtin_convertvector(f32x2{pa[8 * s2 + jj], pa[8 * s2 + jj + 1]}, bf16x2);
        pf[jj] = pr[0]; pf[jj + 1] = pr[1];
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[par][2 * s2 + dt], pf, (first && s2 == 0) ? zero16 : oacc[dt], 0, 0, 0);
    }
  };
  auto mask = [&](f32x16& acc, int t) {                           // padded keys (>= M) score -inf; only instantiated when M < MP
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
      if (key >= p.M) acc[r] = -INFINITY;
    }
  };
  auto rowmax = [&](const f32x16* acc, int nt) {
    float m = acc[0][0];
#pragma unroll
    for (int ti = 0; ti < BT0; ++ti) {
      if (ti < nt) {
#pragma unroll
        for (int r = (ti == 0 ? 1 : 0); r < 16; r += 2)
          m = (r + 1 < 16) ? __builtin_fmaxf(__builtin_fmaxf(m, acc[ti][r]), acc[ti][r + 1]) : __builtin_fmaxf(m, acc[ti][r]);
      }
    }
    return fmaxf(m, __shfl_xor(m, 32));
  };
  float mneg, sum, mref;
  auto expo = [&](f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[r], sl2, mneg));
      acc[r] = e;
      sum += e;
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // step index -> ring parity: steps of a tile run [0, BT1) block-1 scores, [BT1, NKT) block-0 P V, [NKT, NKT + BT1) block-1 P V,
  // [NKT + BT1, 2 NKT) next tile's block-0 scores; the fragments of step n sit in ring slot n & 1
#define ATTN_PAR(n) std::integral_constant<int, ((n) & 1)>{}

  auto take_q = [&](int q_after) {              // the landed tile -> fragments; then the buffer is free for the tile after it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(myQ + l31 * 128 + (((2 * s + g) ^ ((l31 >> 1) & 7)) << 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (q_after < q_end) dma_q(q_after);
  };
  int q0 = q_begin + wave * 32;
  if (q0 < q_end) {
    take_q(q0 + NW * 32);
    // the first tile's block-0 scores, un-overlapped (ring parity as in the steady state: step NKT + BT1 + k)
    readK(ATTN_PAR(NKT + BT1), 0);
#pragma unroll
    for (int k = 0; k < BT0; ++k) {
      if (k + 1 < BT0) { if ((NKT + BT1 + k + 1) & 1) readK(P1{}, k + 1); else readK(P0{}, k + 1); }
      else if (BT1 > 0) readK(P0{}, BT0);              // step 0 of the steady state: parity 0
      else readV(P0{}, 0);
      if ((NKT + BT1 + k) & 1) mmaS(P1{}, accA[k]); else mmaS(P0{}, accA[k]);
    }
  }
  for (; q0 < q_end; q0 += NW * 32) {
    const int q = q0 + l31;
    const bool has_next = q0 + NW * 32 < q_end;
    // ---- block-1 scores (MFMA)  ||  block-0 numerators relative to block 0's row maximum (VALU)
    if (PADDED) {
#pragma unroll
      for (int ti = 0; ti < BT0; ++ti) mask(accA[ti], ti);
    }
    const float m0 = rowmax(accA, BT0);
    mneg = -m0 * sl2;
    sum = 0.f;
    mref = m0;
#pragma unroll
    for (int k = 0; k < BT1; ++k) {
      if (k + 1 < BT1) { if ((k + 1) & 1) readK(P1{}, BT0 + k + 1); else readK(P0{}, BT0 + k + 1); }
      else { if (BT1 & 1) readV(P1{}, 0); else readV(P0{}, 0); }
      if (k & 1) mmaS(P1{}, accB[k]); else mmaS(P0{}, accB[k]);
      // block-0 tiles spread over the BT1 steps (the last step takes what is left)
#pragma unroll
      for (int ti = 0; ti < BT0; ++ti)
        if (ti == k || (k == BT1 - 1 && ti > k)) expo(accA[ti]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BT1 == 0) {
#pragma unroll
      for (int ti = 0; ti < BT0; ++ti) expo(accA[ti]);
    }
    // ---- block-0 P V (MFMA)  ||  block-1 numerators (VALU)
#pragma unroll
    for (int k = 0; k < BT0; ++k) {
      const int n = BT1 + k;
      if (k + 1 < BT0) { if ((n + 1) & 1) readV(P1{}, k + 1); else readV(P0{}, k + 1); }
      else if (BT1 > 0) { if ((n + 1) & 1) readV(P1{}, BT0); else readV(P0{}, BT0); }
      else { if ((n + 1) & 1) readK(P1{}, 0); else readK(P0{}, 0); }
      if (n & 1) mmaPV(P1{}, accA[k], k == 0); else mmaPV(P0{}, accA[k], k == 0);
      if (BT1 > 0) {
        if (k == 0) {
          if (PADDED) {
#pragma unroll
            for (int ti = 0; ti < BT1; ++ti) mask(accB[ti], BT0 + ti);
          }
          const float m1 = rowmax(accB, BT1);
          // block 0's maximum stays the reference unless block 1 tops it by more than 2^24 (exp2 arguments up to 24 lose nothing in
          // fp32 and are far from bf16's range end); otherwise the standard rescale of what block 0 has left behind -- after its P V
          // MFMAs of this very step have been accounted for: the branch sits behind them in program order
          const bool far = (m1 - m0) * sl2 > 24.0f;
          if (__builtin_amdgcn_ballot_w64(far) != 0) {
            const float alpha = far ? __builtin_amdgcn_exp2f((m0 - m1) * sl2) : 1.0f;
            mref = far ? m1 : m0;
            mneg = -mref * sl2;
            sum *= alpha;
            // the remaining block-0 numerators (tiles 1 .. BT0-1 are still to be multiplied into O) and what tile 0 already added
#pragma unroll
            for (int ti = 1; ti < BT0; ++ti)
#pragma unroll
              for (int r = 0; r < 16; ++r) accA[ti][r] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
              for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
          }
        }
#pragma unroll
        for (int ti = 0; ti < BT1; ++ti)
          if (ti == k || (k == BT0 - 1 && ti > k)) expo(accB[ti]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- block-1 P V
#pragma unroll
    for (int k = 0; k < BT1; ++k) {
      const int n = NKT + k;
      if (k + 1 < BT1) { if ((n + 1) & 1) readV(P1{}, BT0 + k + 1); else readV(P0{}, BT0 + k + 1); }
      else { if ((n + 1) & 1) readK(P1{}, 0); else readK(P0{}, 0); }
      if (n & 1) mmaPV(P1{}, accB[k], false); else mmaPV(P0{}, accB[k], false);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the next tile's block-0 scores (MFMA)  ||  normalise and store this tile (VALU)
    if (has_next) take_q(q0 + 2 * NW * 32);
#pragma unroll
    for (int k = 0; k < BT0; ++k) {
      const int n = NKT + BT1 + k;
      if (k + 1 < BT0) { if ((n + 1) & 1) readK(P1{}, k + 1); else readK(P0{}, k + 1); }
      else if (BT1 > 0) readK(P0{}, BT0);
      else readV(P0{}, 0);
      if (has_next) { if (n & 1) mmaS(P1{}, accA[k]); else mmaS(P0{}, accA[k]); }
      if (k == 0) {
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        // oacc[dt][r] = O[q = l31][d = 32 dt + (r&3) + 8 (r>>2) + 4 g]: sixteen rows at a time into the staging tile (8-byte pieces), back out
        // as 16 bytes per lane with eight lanes per row, stored as whole 128-byte lines
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          if ((l31 >> 4) == pass) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
              for (int rq = 0; rq < 4; ++rq) {
                const bf16x2 a = __builtin_convertvector(f32x2{oacc[dt][4 * rq] * inv, oacc[dt][4 * rq + 1] * inv}, bf16x2);
                const bf16x2 c = __builtin_convertvector(f32x2{oacc[dt][4 * rq + 2] * inv, oacc[dt][4 * rq + 3] * inv}, bf16x2);
                *(u32x2*)(myO + (l31 & 15) * 144 + (32 * dt + 8 * rq + 4 * g) * 2) = u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, c)};
              }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int row = 8 * hh + (lane >> 3), qq = q0 + 16 * pass + row;
            const u32x4 v = *(const u32x4*)(myO + row * 144 + (lane & 7) * 16);
            if (qq < q_end) st_g<MVLT_NT_ATTN>((u32x4*)(Og + (long)qq * p.ldo + (lane & 7) * 8), v);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        if (q < q_end && g == 0 && p.lse) p.lse[((long)b * p.H + h) * p.N + q] = mref * p.scale + __logf(sum);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef ATTN_PAR
}

template <int NKT, int NW> int launch_fwd2_nw(const mvlt_attn_args& a, hipStream_t s) {
  constexpr int MP = NKT * 32;
  constexpr int VPAD = 8;
  const size_t lds = (size_t)(MP * HD + HD * (MP + VPAD)) * 2 + NW * 4096 + NW * 2304;
  const int groups = a.B * a.H;
  // a workgroup's waves walk its query chunk in 32-query tiles; chunks as long as the grid allows (K / V staging is per chunk):
  // at least ~1024 workgroups (two per CU, two rounds) when the problem has them
  int nq = 1;
  while (nq < 8 && groups * nq < 1024 && (a.N + nq) / (nq + 1) >= 512) ++nq;
  int q_per_wg = ((a.N + nq - 1) / nq + 32 * NW - 1) / (32 * NW) * (32 * NW);
  nq = (a.N + q_per_wg - 1) / q_per_wg;
  const int grid = 8 * ((groups + 7) / 8) * nq;
  if (a.M == MP) {
    mvlt_max_lds<(attn_fwd2_kernel<NKT, false, NW>)>();
    MVLT_LAUNCH((attn_fwd2_kernel<NKT, false, NW>), dim3(grid), dim3(NW * 64), lds, s, a, nq, q_per_wg);
  } else {
    mvlt_max_lds<(attn_fwd2_kernel<NKT, true, NW>)>();
    MVLT_LAUNCH((attn_fwd2_kernel<NKT, true, NW>), dim3(grid), dim3(NW * 64), lds, s, a, nq, q_per_wg);
  }
  return mvlt_check_launch("mvlt_sr_attention_fwd");
}
// four waves while two workgroups fit the 160 KB of a CU (up to 192 keys: 75.8 KB each), eight waves in ONE workgroup beyond (272 keys at
// 384 px: 100 KB with four waves would leave one wave per SIMD)
template <int NKT> int launch_fwd2_n(const mvlt_attn_args& a, hipStream_t s) {
  if constexpr (NKT <= 6) return launch_fwd2_nw<NKT, 4>(a, s);
  else return launch_fwd2_nw<NKT, 8>(a, s);
}

int launch_fwd2(const mvlt_attn_args& a, hipStream_t s) {
  switch ((a.M + 31) / 32) {
    case 1: return launch_fwd2_n<1>(a, s);
    case 2: return launch_fwd2_n<2>(a, s);
    case 3: return launch_fwd2_n<3>(a, s);
    case 4: return launch_fwd2_n<4>(a, s);
    case 5: return launch_fwd2_n<5>(a, s);
    case 6: return launch_fwd2_n<6>(a, s);
    default: break;
  }
  mvlt_set_error("mvlt_sr_attention_fwd: the round-3 kernel takes M <= 192 keys, got %d", a.M);
  return MVLT_ERR_UNSUPPORTED;
}


// ------------------------------------------------------------------------------------------------ backward
// dQ = dS K, dK = dS^T Q, dV = P^T dO with P = exp(S*scale - lse), dS = P (dP - D) scale, dP = dO V^T, D = rowsum(dO*O).
// Work split (per (batch, head, query chunk) workgroup): wave w OWNS key tiles [w*TPW, (w+1)*TPW) x 16 keys: its K/V
// fragments stay in registers for the whole kernel and its dK/dV tiles accumulate in registers over all query
// tiles; one fp32 atomic flush at the end.  Per 32-query tile: S and dP via 16x16x32 MFMA (rows = queries), P/dS
// are re-used from the accumulators as the A operand of dV/dK (k-slots = queries), dS is also parked in LDS as
// [q][key] so that dQ (k = all keys) is computed by all waves against the LDS-resident K^T.
template <typename T, int NW, int TPW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_bwd_kernel(mvlt_attn_bwd_args p, int nq_chunks, int q_per_wg) {
  constexpr int NTH = NW * 64;
  constexpr int MP = NW * TPW * 16;           // padded keys (multiple of 32)
  constexpr int PC = Lds<T>::PC, NCH = Lds<T>::NCH;
  constexpr int PAD = 16 / sizeof(T);         // 16 B of padding per row
  constexpr int KS = MP + PAD;                // row stride of sKt and sdS (elements)
  constexpr int QS = 32 + PAD;                // row stride of sQt / sdOt
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sKt = (T*)smem;                          // [64 d][KS]   K^T
  T* sdS = sKt + HD * KS;                     // [32 q][KS]
  T* sQt = sdS + 32 * KS;                     // [64 d][QS]   Q^T tile
  T* sdOt = sQt + HD * QS;                    // [64 d][QS]   dO^T tile
  float* sD = (float*)(sdOt + HD * QS);       // [32]
  float* sL = sD + 32;                        // [32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, jb = bid >> 3;
  const int gidx = xcd + 8 * (jb / nq_chunks);
  const int chunk_id = jb % nq_chunks;
  if (gidx >= p.B * p.H) return;
  const int b = gidx / p.H, h = gidx % p.H;

  const T* Qg = (const T*)p.Q + (long)b * p.N * p.ldq + h * HD;
  const T* Og = (const T*)p.O + (long)b * p.N * p.ldo + h * HD;
  const T* dOg = (const T*)p.dO + (long)b * p.N * p.ldo + h * HD;
  const T* Kg = (const T*)p.KV + (long)b * p.M * p.ldkv + p.k_off + h * HD;
  const T* Vg = (const T*)p.KV + (long)b * p.M * p.ldkv + p.v_off + h * HD;
  T* dQg = (T*)p.dQ + (long)b * p.N * p.ldq + h * HD;
  float* dKg = (float*)p.dKV + (long)b * p.M * p.lddkv + p.k_off + h * HD;
  float* dVg = (float*)p.dKV + (long)b * p.M * p.lddkv + p.v_off + h * HD;
  const float* Lg = p.lse + ((long)b * p.H + h) * p.N;

  // K^T into LDS (all keys), zero beyond M
  for (int u = tid; u < MP * NCH; u += NTH) {
    int r = u / NCH, c = u % NCH;
    u32x4 kv = {0u, 0u, 0u, 0u};
    if (r < p.M) kv = *(const u32x4*)(Kg + (long)r * p.ldkv + c * PC);
    T ke[PC];
    *(u32x4*)ke = kv;
#pragma unroll
    for (int e = 0; e < PC; ++e) sKt[(c * PC + e) * KS + r] = ke[e];
  }
  // this wave's K / V fragments (B operands: n = key = fr, k = d)
  Frag<T> kreg[TPW][2], vreg[TPW][2];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    int key = (wave * TPW + t) * 16 + fr;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (key < p.M) {
        kreg[t][s] = load_frag8<T>(Kg + (long)key * p.ldkv + 32 * s + 8 * fg);
        vreg[t][s] = load_frag8<T>(Vg + (long)key * p.ldkv + 32 * s + 8 * fg);
      } else {
        kreg[t][s] = zero_frag<T>();
        vreg[t][s] = zero_frag<T>();
      }
    }
  }
  f32x4 dKacc[TPW][4], dVacc[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dKacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dVacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const float sl2 = p.scale * 1.44269504088896340736f;
  const float l2e = 1.44269504088896340736f;
  const int q_begin = chunk_id * q_per_wg;
  const int q_end = min(p.N, q_begin + q_per_wg);

  for (int q0 = q_begin; q0 < q_end; q0 += 32) {
    // ---- (a) stage Q^T, dO^T tiles, D and lse for 32 queries
    for (int u = tid; u < 32 * NCH; u += NTH) {
      int r = u / NCH, c = u % NCH;
      int q = q0 + r;
      u32x4 qv = {0u, 0u, 0u, 0u}, dov = {0u, 0u, 0u, 0u}, ov = {0u, 0u, 0u, 0u};
      if (q < q_end) {
        qv = *(const u32x4*)(Qg + (long)q * p.ldq + c * PC);
        dov = *(const u32x4*)(dOg + (long)q * p.ldo + c * PC);
        ov = *(const u32x4*)(Og + (long)q * p.ldo + c * PC);
      }
      T qe[PC], de[PC], oe[PC];
      *(u32x4*)qe = qv; *(u32x4*)de = dov; *(u32x4*)oe = ov;
      float dsum = 0.f;
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        sQt[(c * PC + e) * QS + r] = qe[e];
        sdOt[(c * PC + e) * QS + r] = de[e];
        dsum += (float)de[e] * (float)oe[e];
      }
#pragma unroll
      for (int o = NCH / 2; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o);
      if (c == 0) sD[r] = dsum;
    }
    if (tid < 32) sL[tid] = (q0 + tid < q_end) ? Lg[q0 + tid] : 0.f;
    // A-operand fragments of Q and dO (rows = queries) straight from global
    Frag<T> qf[2][2], dof[2][2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      int q = q0 + qs * 16 + fr;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (q < q_end) {
          qf[qs][s] = load_frag8<T>(Qg + (long)q * p.ldq + 32 * s + 8 * fg);
          dof[qs][s] = load_frag8<T>(dOg + (long)q * p.ldo + 32 * s + 8 * fg);
        } else {
          qf[qs][s] = zero_frag<T>();
          dof[qs][s] = zero_frag<T>();
        }
      }
    }
    __syncthreads();                                   // (b)

    // ---- (c) per owned key tile: S, dP -> P, dS ; dV += P^T dO ; dK += dS^T Q ; park dS in LDS
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int key = (wave * TPW + t) * 16 + fr;
      const bool key_ok = key < p.M;
      Frag<T> pfrag, dsfrag;
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          mma16(sacc, qf[qs][s], kreg[t][s]);          // S[q = 16 qs + 4 fg + r][key]
          mma16(pacc, dof[qs][s], vreg[t][s]);         // dP
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qs * 16 + 4 * fg + r;
          float pv = key_ok ? exp2f(sacc[r] * sl2 - sL[ql] * l2e) : 0.f;
          float dsv = pv * (pacc[r] - sD[ql]) * p.scale;
          pfrag.v[qs * 4 + r] = (T)pv;
          dsfrag.v[qs * 4 + r] = (T)dsv;
          sdS[ql * KS + key] = (T)dsv;
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // B operands (n = d): k-slot (fg, j) <-> q = 16 (j>>2) + 4 fg + (j&3); re-read per tile to stay under 256 VGPRs
        const T* a = sQt + (dt * 16 + fr) * QS + 4 * fg;
        const T* c2 = sdOt + (dt * 16 + fr) * QS + 4 * fg;
        Frag<T> dotf = load_frag44<T>(c2, c2 + 16);
        Frag<T> qtf = load_frag44<T>(a, a + 16);
        mma16(dVacc[t][dt], pfrag, dotf);              // dV[key = tile*16 + 4 fg + r][d = 16 dt + fr]
        mma16(dKacc[t][dt], dsfrag, qtf);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                   // (d)

    // ---- (e) dQ[32 x 64] = dS[32 x MP] K[MP x 64]: 8 output tiles (16 x 16) dealt round-robin to the waves
    for (int tile = wave; tile < 8; tile += NW) {
      const int qs = tile >> 2, dt = tile & 3;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int ks = 0; ks < MP / 32; ++ks) {
        Frag<T> a = load_frag8<T>(sdS + (qs * 16 + fr) * KS + 32 * ks + 8 * fg);
        Frag<T> bb = load_frag8<T>(sKt + (dt * 16 + fr) * KS + 32 * ks + 8 * fg);
        mma16(acc, a, bb);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int q = q0 + qs * 16 + 4 * fg + r;
        if (q < q_end) dQg[(long)q * p.ldq + dt * 16 + fr] = (T)acc[r];
      }
    }
  }
  // ---- flush dK / dV
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = (wave * TPW + t) * 16 + 4 * fg + r;
        if (key < p.M) {
          if (nq_chunks == 1) {        // this workgroup saw every query of its (batch, head): the sums are final
            dKg[(long)key * p.lddkv + dt * 16 + fr] = dKacc[t][dt][r];
            dVg[(long)key * p.lddkv + dt * 16 + fr] = dVacc[t][dt][r];
          } else {
            atomicAdd(&dKg[(long)key * p.lddkv + dt * 16 + fr], dKacc[t][dt][r]);
            atomicAdd(&dVg[(long)key * p.lddkv + dt * 16 + fr], dVacc[t][dt][r]);
          }
        }
      }
}

// ------------------------------------------------------------------------------------------------ backward, bf16, LDS-DMA
// Same algorithm and work split as attn_bwd_kernel; what changes is how the per-tile operands reach the MFMAs.  The Q, dO
// and O rows of a 32-query tile arrive by LDS-DMA into a 2-deep ring of natural-layout [query][64] tiles, one tile ahead
// of the math (the kernel above paid a full global round trip per tile: load -> transposing 2-byte LDS stores -> barrier).
// Row fragments (A operands of S / dP) are 16-byte reads of those tiles, the transposed fragments (B operands of dV / dK,
// k = queries) are ds_read_b64_tr_b16 of the same tiles; D = rowsum(dO * O) is computed by every wave for itself from
// LDS (no workgroup barrier), dQ leaves through an LDS tile as 16-byte row stores one iteration later.
template <int NW, int TPW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_bwd_dma_kernel(mvlt_attn_bwd_args p, int nq_chunks, int q_per_wg) {
  // (round 5 bounded a split backward with two timing variants of this template -- the key-split half alone; the transposed fragments read once per
  //  query tile -- and rejected it: docs/experiments_r5.md 2, code in commit 2c01256)
  typedef bf16 T;
  constexpr int NTH = NW * 64;
  constexpr int MP = NW * TPW * 16;           // padded keys (multiple of 32)
  constexpr int KS = MP + 8;                  // row stride of sKt and sdS (elements; 16 B of padding)
  constexpr int TILE = 32 * 128;              // bytes of one [32 q][64 d] tile
  constexpr int STAGE = 3 * TILE;             // Q | dO | O
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sKt = (T*)smem;                          // [64 d][KS]   K^T
  T* sdS = sKt + HD * KS;                     // [32 q][KS]
  char* ring = (char*)(sdS + 32 * KS);        // [2][Q | dO | O]
  T* sdQ = (T*)(ring + 2 * STAGE);            // [32 q][64 d]
  float* sDw = (float*)(sdQ + 32 * HD);       // [NW][32]
  float* sLw = sDw + NW * 32;                 // [NW][32]
  const unsigned ring_lds = (unsigned)(uintptr_t)ring;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, jb = bid >> 3;
  const int gidx = xcd + 8 * (jb / nq_chunks);
  const int chunk_id = jb % nq_chunks;
  if (gidx >= p.B * p.H) return;
  const int b = gidx / p.H, h = gidx % p.H;

  const T* Qg = (const T*)p.Q + (long)b * p.N * p.ldq + h * HD;
  const T* Og = (const T*)p.O + (long)b * p.N * p.ldo + h * HD;
  const T* dOg = (const T*)p.dO + (long)b * p.N * p.ldo + h * HD;
  const T* Kg = (const T*)p.KV + (long)b * p.M * p.ldkv + p.k_off + h * HD;
  const T* Vg = (const T*)p.KV + (long)b * p.M * p.ldkv + p.v_off + h * HD;
  T* dQg = (T*)p.dQ + (long)b * p.N * p.ldq + h * HD;
  float* dKg = (float*)p.dKV + (long)b * p.M * p.lddkv + p.k_off + h * HD;
  float* dVg = (float*)p.dKV + (long)b * p.M * p.lddkv + p.v_off + h * HD;
  const float* Lg = p.lse + ((long)b * p.H + h) * p.N;
  const int q_begin = chunk_id * q_per_wg;
  const int q_end = min(p.N, q_begin + q_per_wg);

  // ---- DMA geometry: thread (row = tid >> 3, slot = tid & 7) of the first 256 threads fills one 16-B slot per tensor
  const int l_row = (tid >> 3) & 31;
  const int l_chunk = (tid & 7) ^ (((l_row >> 1) & 3) << 1);
  const char* zsrc = (const char*)g_zero_page + ((tid * 16 + (bid & 15) * 4096) & 65535);
  auto issue = [&](int q0, int slot) {
    if (NW > 4 && wave >= 4) return;
    const int q = q0 + l_row;
    const bool ok = q < q_end;
    const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + slot * STAGE + wave * 1024);
    glds16(ok ? (const void*)(Qg + (long)q * p.ldq + l_chunk * 8) : (const void*)zsrc, dst);
    glds16(ok ? (const void*)(dOg + (long)q * p.ldo + l_chunk * 8) : (const void*)zsrc, dst + TILE);
    glds16(ok ? (const void*)(Og + (long)q * p.ldo + l_chunk * 8) : (const void*)zsrc, dst + 2 * TILE);
  };
  issue(q_begin, 0);
  float lse_cur = (lane < 32 && q_begin + lane < q_end) ? Lg[q_begin + lane] : 0.f;

  // K^T into LDS (all keys), zero beyond M
  for (int u = tid; u < MP * 8; u += NTH) {
    int r = u >> 3, c = u & 7;
    u32x4 kv = {0u, 0u, 0u, 0u};
    if (r < p.M) kv = *(const u32x4*)(Kg + (long)r * p.ldkv + c * 8);
    T ke[8];
    *(u32x4*)ke = kv;
#pragma unroll
    for (int e = 0; e < 8; ++e) sKt[(c * 8 + e) * KS + r] = ke[e];
  }
  // this wave's K / V fragments (B operands: n = key = fr, k = d)
  Frag<T> kreg[TPW][2], vreg[TPW][2];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    int key = (wave * TPW + t) * 16 + fr;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (key < p.M) {
        kreg[t][s] = load_frag8<T>(Kg + (long)key * p.ldkv + 32 * s + 8 * fg);
        vreg[t][s] = load_frag8<T>(Vg + (long)key * p.ldkv + 32 * s + 8 * fg);
      } else {
        kreg[t][s] = zero_frag<T>();
        vreg[t][s] = zero_frag<T>();
      }
    }
  }
  f32x4 dKacc[TPW][4], dVacc[TPW][4];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dKacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dVacc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const float sl2 = p.scale * 1.44269504088896340736f;
  const float l2e = 1.44269504088896340736f;

  // ---- fragment geometry inside a tile (byte offsets)
  const int hs_r = (fr >> 1) & 3;                                           // rows 16 qs + fr
  int roff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) roff[s] = fr * 128 + (((s * 4 + fg) ^ (hs_r << 1)) << 4);
  const int trow = 4 * fg + (fr >> 2);                                      // transposed reads: rows trow, trow + 16
  const int hs_t = (trow >> 1) & 3;
  int toffs[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) toffs[dt] = trow * 128 + ((dt ^ hs_t) << 5) + ((fr & 3) << 3);
  const int dq_row = lane >> 1, dq_half = lane & 1;                         // per-wave D: lane = (query, half of d)
  const int hs_d = (dq_row >> 1) & 3;
  float* myD = sDw + wave * 32;
  float* myL = sLw + wave * 32;

  int slot = 0;
  int q_prev = -1;
  for (int q0 = q_begin; q0 < q_end; q0 += 32, slot ^= 1) {
    // lgkmcnt(0) explicitly: hipcc emitted this barrier without draining the wave's own ds_write_b16 of the dQ tile issued just
    // before the loop back-edge, and the deferred dQ store below then read two or four stale rows about once in 100 launches
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                   // (A) tile q0 landed; previous tile's dS / dQ tiles are complete
    if (q_prev >= 0 && tid < 256) {                    // deferred dQ store of the previous tile: 16 B per thread
      const int r = tid >> 3, c = tid & 7;
      if (q_prev + r < q_end) st_g<MVLT_NT_ATTN>((u32x4*)(dQg + (long)(q_prev + r) * p.ldq + c * 8), *(const u32x4*)(sdQ + r * HD + c * 8));
    }
    const float lse_now = lse_cur;
    if (q0 + 32 < q_end) {
      lse_cur = (lane < 32 && q0 + 32 + lane < q_end) ? Lg[q0 + 32 + lane] : 0.f;
      issue(q0 + 32, slot ^ 1);
    }
    const char* tQ = ring + slot * STAGE;
    const char* tdO = tQ + TILE;
    const char* tO = tQ + 2 * TILE;

    // ---- per-wave D = rowsum(dO * O) and lse into this wave's scratch
    {
      float dsum = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int off = dq_row * 128 + (((dq_half * 4 + c) ^ (hs_d << 1)) << 4);
        const bf16x8 a = *(const bf16x8*)(tdO + off), o = *(const bf16x8*)(tO + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)a[e] * (float)o[e];
      }
      dsum += __shfl_xor(dsum, 1);
      if (dq_half == 0) myD[dq_row] = dsum;
      if (lane < 32) myL[lane] = lse_now * l2e;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    f32x4 dv[2], lv[2];                                // D and lse*log2e of this lane's 8 query rows (16 qs + 4 fg + r)
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      dv[qs] = *(const f32x4*)(myD + qs * 16 + 4 * fg);
      lv[qs] = *(const f32x4*)(myL + qs * 16 + 4 * fg);
    }
    // A-operand fragments of Q and dO (rows = queries)
    Frag<T> qf[2][2], dof[2][2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        qf[qs][s].v = *(const bf16x8*)(tQ + qs * 16 * 128 + roff[s]);
        dof[qs][s].v = *(const bf16x8*)(tdO + qs * 16 * 128 + roff[s]);
      }

    // ---- per owned key tile: S, dP -> P, dS ; dV += P^T dO ; dK += dS^T Q ; park dS in LDS
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int key = (wave * TPW + t) * 16 + fr;
      const bool key_ok = key < p.M;
      Frag<T> pfrag, dsfrag;
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          mma16(sacc, qf[qs][s], kreg[t][s]);          // S[q = 16 qs + 4 fg + r][key]
          mma16(pacc, dof[qs][s], vreg[t][s]);         // dP
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qs * 16 + 4 * fg + r;
          float pv = key_ok ? __builtin_amdgcn_exp2f(sacc[r] * sl2 - lv[qs][r]) : 0.f;
          float dsv = pv * (pacc[r] - dv[qs][r]) * p.scale;
          pfrag.v[qs * 4 + r] = (T)pv;
          dsfrag.v[qs * 4 + r] = (T)dsv;
          sdS[ql * KS + key] = (T)dsv;
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // B operands (n = d): k-slot (fg, j) <-> q = 16 (j>>2) + 4 fg + (j&3): two transposed 8-byte reads, 16 rows apart
        Frag<T> dotf, qtf;
        dotf.v = __builtin_bit_cast(bf16x8, tr_frag16(tdO + toffs[dt]));
        qtf.v = __builtin_bit_cast(bf16x8, tr_frag16(tQ + toffs[dt]));
        mma16(dVacc[t][dt], pfrag, dotf);              // dV[key = tile*16 + 4 fg + r][d = 16 dt + fr]
        mma16(dKacc[t][dt], dsfrag, qtf);
      }
    }
    __syncthreads();                                   // (C) every wave's dS columns are parked

    // ---- dQ[32 x 64] = dS[32 x MP] K[MP x 64]: 8 output tiles (16 x 16) dealt round-robin to the waves -> sdQ
    if constexpr (NW == 4) {
      // four waves: wave w owns output tiles (qs = 0, dt = w) and (qs = 1, dt = w) -- the same K^T fragments serve both, the two
      // accumulator chains are independent, and a k-step's three fragment reads are issued one step ahead of its two MFMAs
      const int dt = wave;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const T* pa0 = sdS + fr * KS + 8 * fg;
      const T* pa1 = sdS + (16 + fr) * KS + 8 * fg;
      const T* pb = sKt + (dt * 16 + fr) * KS + 8 * fg;
      Frag<T> a0 = load_frag8<T>(pa0), a1 = load_frag8<T>(pa1), bb = load_frag8<T>(pb);
#pragma unroll
      for (int ks = 0; ks < MP / 32; ++ks) {
        Frag<T> n0 = a0, n1 = a1, nb = bb;
        if (ks + 1 < MP / 32) {
          n0 = load_frag8<T>(pa0 + 32 * (ks + 1)); n1 = load_frag8<T>(pa1 + 32 * (ks + 1)); nb = load_frag8<T>(pb + 32 * (ks + 1));
        }
        mma16(acc0, a0, bb);
        mma16(acc1, a1, bb);
        a0 = n0; a1 = n1; bb = nb;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sdQ[(4 * fg + r) * HD + dt * 16 + fr] = (T)acc0[r];
        sdQ[(16 + 4 * fg + r) * HD + dt * 16 + fr] = (T)acc1[r];
      }
    } else {
    for (int tile = wave; tile < 8; tile += NW) {
      const int qs = tile >> 2, dt = tile & 3;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int ks = 0; ks < MP / 32; ++ks) {
        Frag<T> a = load_frag8<T>(sdS + (qs * 16 + fr) * KS + 32 * ks + 8 * fg);
        Frag<T> bb = load_frag8<T>(sKt + (dt * 16 + fr) * KS + 32 * ks + 8 * fg);
        mma16(acc, a, bb);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sdQ[(qs * 16 + 4 * fg + r) * HD + dt * 16 + fr] = (T)acc[r];
    }
    }
    q_prev = q0;
  }
  __syncthreads();
  if (q_prev >= 0 && tid < 256) {
    const int r = tid >> 3, c = tid & 7;
    if (q_prev + r < q_end) st_g<MVLT_NT_ATTN>((u32x4*)(dQg + (long)(q_prev + r) * p.ldq + c * 8), *(const u32x4*)(sdQ + r * HD + c * 8));
  }
  // ---- flush dK / dV.  One query chunk per (batch, head) and a bf16 dKV (stages 2-4): the accumulator layout gives a lane one column and four ROWS --
  //      stored directly that was 32 two-byte stores per key tile and lane, 12 / 20 / 35 us of the 152 / 152 / 146 us launches of stages 2 / 3 / 4 (ablation,
  //      round 4).  Each wave parks a [16 keys][K 64 | V 64] tile in its own LDS slice (the ring is free now) and stores whole 16-byte row pieces.
  if (nq_chunks == 1 && p.dkv_dtype == 0) {
    constexpr int LDW = 128 + 8;
    bf16* sX = (bf16*)smem + wave * 16 * LDW;
    __syncthreads();                                   // every wave is done with the K^T / dS tiles this overlays
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sX[(4 * fg + r) * LDW + dt * 16 + fr] = (bf16)dKacc[t][dt][r];
          sX[(4 * fg + r) * LDW + 64 + dt * 16 + fr] = (bf16)dVacc[t][dt][r];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = i * 64 + lane, row = q >> 4, c = q & 15;
        const int key = (wave * TPW + t) * 16 + row;
        if (key < p.M)
          *(u32x4*)((bf16*)p.dKV + ((long)b * p.M + key) * p.lddkv + (c < 8 ? p.k_off : p.v_off) + h * HD + (c & 7) * 8) = *(const u32x4*)(sX + row * LDW + c * 8);
      }
    }
    return;
  }
  // ---- flush dK / dV
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = (wave * TPW + t) * 16 + 4 * fg + r;
        if (key < p.M) {
          if (nq_chunks == 1 && p.dkv_dtype == 0) {       // bf16 dKV: the operand of the kv-projection gradients, no cast pass
            ((bf16*)p.dKV)[((long)b * p.M + key) * p.lddkv + p.k_off + h * HD + dt * 16 + fr] = (bf16)dKacc[t][dt][r];
            ((bf16*)p.dKV)[((long)b * p.M + key) * p.lddkv + p.v_off + h * HD + dt * 16 + fr] = (bf16)dVacc[t][dt][r];
          } else if (nq_chunks == 1) {
            dKg[(long)key * p.lddkv + dt * 16 + fr] = dKacc[t][dt][r];
            dVg[(long)key * p.lddkv + dt * 16 + fr] = dVacc[t][dt][r];
          } else {
            atomicAdd(&dKg[(long)key * p.lddkv + dt * 16 + fr], dKacc[t][dt][r]);
            atomicAdd(&dVg[(long)key * p.lddkv + dt * 16 + fr], dVacc[t][dt][r]);
          }
        }
      }
}

// Query chunks per (batch, head) of the backward.  With ONE chunk a workgroup sees every query of its (batch, head) and stores dK / dV plainly; every extra chunk
// adds M x 128 fp32 atomics per (batch, head) and pays the prologue (K^T staging, K / V fragments) again.  Round 1-5 took ceil(512 / groups) -- fill the chip once --
// which at pvlt_medium's stage 3 (320 (batch, head) pairs, 704 queries, 272 keys, one six-wave workgroup per CU) meant two chunks = 3 rounds of 11 query tiles + 22 M
// atomics: 175 us, where ONE chunk (2 rounds of 22 tiles, no atomics, no zero fill, no cast) takes 149 us (tools/ubench_attn.py sweep, profiles/r06_attn_bwd_chunks.txt).
// Round 6: the chunk count minimises a cost model fitted to that sweep and to the 256-px shapes -- rounds of the chip x (32-query tiles per chunk + 2 tiles of fixed
// cost) x 3.2 us, + chunks x groups x M x 128 atomics at 0.5 per ns when there is more than one chunk.  `slots` = workgroups the chip holds at once (two per CU for
// the four-wave bf16 instantiations, one otherwise).  It reproduces ceil(512 / groups) on every 256-px shape of the BASELINE configurations.
inline int attn_bwd_chunks(int groups, int N, int M, int slots, bool one_chunk_only, int* q_per_wg) {
  static const int force = getenv("MVLT_ATTN_BWD_NQ") ? atoi(getenv("MVLT_ATTN_BWD_NQ")) : 0;      // measurement switch (tools/ubench_attn.py)
  int best_nq = 1;
  double best = -1.0;
  const int max_nq = one_chunk_only ? 1 : min(128, max(1, (N + 63) / 64));       // at least two query tiles per chunk
  for (int nq = 1; nq <= max_nq; ++nq) {
    const int qpw = ((N + nq - 1) / nq + 31) / 32 * 32;
    if ((N + qpw - 1) / qpw != nq) continue;                        // the rounding to whole tiles merged two chunks
    const long rounds = ((long)groups * nq + slots - 1) / slots;
    double cost = (double)rounds * (qpw / 32 + 2) * 3.2;
    if (nq > 1) cost += (double)nq * groups * M * 128.0 / 500.0 * 1e-3;
    if (force > 0 && !one_chunk_only) cost = nq == force ? 0.0 : 1.0;
    if (best < 0.0 || cost < best) { best = cost; best_nq = nq; }
  }
  *q_per_wg = ((N + best_nq - 1) / best_nq + 31) / 32 * 32;
  return (N + *q_per_wg - 1) / *q_per_wg;
}
template <typename T, int NW> constexpr int attn_bwd_slots() { return (sizeof(T) == 2 && NW == 4) ? 512 : 256; }

template <typename T, int NW, int TPW> int launch_bwd_n(const mvlt_attn_bwd_args& a, hipStream_t s) {
  constexpr int MP = NW * TPW * 16;
  constexpr int PAD = 16 / sizeof(T);
  const size_t lds = (size_t)(HD * (MP + PAD) + 32 * (MP + PAD) + 2 * HD * (32 + PAD)) * sizeof(T) + 64 * sizeof(float);
  MVLT_REQUIRE(lds <= 160 * 1024, "mvlt_sr_attention_bwd: LDS %zu B > 160 KB", lds);
  const int groups = a.B * a.H;
  int q_per_wg = 0;
  const int nq = attn_bwd_chunks(groups, a.N, a.M, attn_bwd_slots<T, NW>(), a.dkv_dtype == 0, &q_per_wg);
  const int grid = 8 * ((groups + 7) / 8) * nq;
  if constexpr (sizeof(T) == 2) {
    {
      const size_t lds2 = (size_t)(HD * (MP + 8) + 32 * (MP + 8)) * 2 + 2 * 3 * 4096 + 4096 + 2 * NW * 32 * sizeof(float);
      MVLT_REQUIRE(lds2 <= 160 * 1024, "mvlt_sr_attention_bwd: LDS %zu B > 160 KB", lds2);
      mvlt_max_lds<(attn_bwd_dma_kernel<NW, TPW>)>();
      MVLT_LAUNCH((attn_bwd_dma_kernel<NW, TPW>), dim3(grid), dim3(NW * 64), lds2, s, a, nq, q_per_wg);
      return mvlt_check_launch("mvlt_sr_attention_bwd");
    }
  } else {                                             // fp32 parity path
    mvlt_max_lds<(attn_bwd_kernel<T, NW, TPW>)>();
    MVLT_LAUNCH((attn_bwd_kernel<T, NW, TPW>), dim3(grid), dim3(NW * 64), lds, s, a, nq, q_per_wg);
    return mvlt_check_launch("mvlt_sr_attention_bwd");
  }
}

// workgroups per CU class of the instantiation launch_bwd picks for M keys (same ladder as below)
template <typename T> int attn_bwd_slots_for(int M) { return (sizeof(T) == 2 && M <= 192) ? 512 : 256; }

template <typename T> int launch_bwd(const mvlt_attn_bwd_args& a, hipStream_t s) {
  const int M = a.M;
  if (M <= 64) return launch_bwd_n<T, 4, 1>(a, s);
  if (M <= 128) return launch_bwd_n<T, 4, 2>(a, s);
  if (M <= 192) return launch_bwd_n<T, 4, 3>(a, s);
  if (M <= 256) return launch_bwd_n<T, 8, 2>(a, s);
  if (M <= 288) return launch_bwd_n<T, 6, 3>(a, s);
  if (M <= 320) return launch_bwd_n<T, 8, 3>(a, s);
  mvlt_set_error("mvlt_sr_attention_bwd: M=%d keys exceeds the 320-key LDS-resident design", M);
  return MVLT_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int mvlt_sr_attention_fwd(const mvlt_attn_args* a, void* stream) {
  MVLT_REQUIRE(a && a->Q && a->KV && a->O, "mvlt_sr_attention_fwd: null pointer");
  MVLT_REQUIRE(a->B > 0 && a->H > 0 && a->N > 0 && a->M > 0, "mvlt_sr_attention_fwd: bad shape");
  MVLT_REQUIRE(a->dtype == 0 || a->dtype == 1, "mvlt_sr_attention_fwd: bad dtype");
  const int pc = a->dtype == 0 ? 8 : 4;
  MVLT_REQUIRE(a->ldq % pc == 0 && a->ldkv % pc == 0 && a->ldo % pc == 0 && a->k_off % pc == 0 && a->v_off % pc == 0,
               "mvlt_sr_attention_fwd: strides/offsets must be multiples of %d elements", pc);
  // round-3 kernel up to 192 keys (every 256-px configuration); beyond (272 keys at 384 px) its two score blocks no longer fit the
  // register budget of two waves per SIMD next to the O accumulators (144 + 32 of 256), and the round-2 kernel is the faster one
  if (a->dtype == 0 && a->M <= 192) return launch_fwd2(*a, (hipStream_t)stream);
  return a->dtype == 0 ? launch_fwd<bf16>(*a, (hipStream_t)stream) : launch_fwd<float>(*a, (hipStream_t)stream);
}

extern "C" int mvlt_sr_attention_bwd_chunks(int B, int H, int N, int M, int dtype) {
  int q_per_wg = 0;
  if (B <= 0 || H <= 0 || N <= 0 || M <= 0) return 0;
  return attn_bwd_chunks(B * H, N, M, dtype == 0 ? attn_bwd_slots_for<bf16>(M) : attn_bwd_slots_for<float>(M), false, &q_per_wg);
}

extern "C" int mvlt_sr_attention_bwd(const mvlt_attn_bwd_args* a, void* stream) {
  MVLT_REQUIRE(a && a->Q && a->KV && a->O && a->dO && a->lse && a->dQ && a->dKV, "mvlt_sr_attention_bwd: null pointer");
  MVLT_REQUIRE(a->dkv_dtype == 1 || (a->dkv_dtype == 0 && a->dtype == 0),
               "mvlt_sr_attention_bwd: bf16 dKV needs bf16 operands (it forces one query chunk per (batch, head): plain stores)");
  MVLT_REQUIRE(a->B > 0 && a->H > 0 && a->N > 0 && a->M > 0, "mvlt_sr_attention_bwd: bad shape");
  MVLT_REQUIRE(a->dtype == 0 || a->dtype == 1, "mvlt_sr_attention_bwd: bad dtype");
  const int pc = a->dtype == 0 ? 8 : 4;
  MVLT_REQUIRE(a->ldq % pc == 0 && a->ldkv % pc == 0 && a->ldo % pc == 0 && a->k_off % pc == 0 && a->v_off % pc == 0,
               "mvlt_sr_attention_bwd: strides/offsets must be multiples of %d elements", pc);
  return a->dtype == 0 ? launch_bwd<bf16>(*a, (hipStream_t)stream) : launch_bwd<float>(*a, (hipStream_t)stream);
}
