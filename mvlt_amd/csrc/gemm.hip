// MFMA GEMMs for the MVLT hot path (gfx950 / CDNA4, wave64).
//
//  mvlt_gemm_nt : C[M,N] = epilogue(A[M,K] . B[N,K]^T)      Linear forward / dgrad (with W^T) and the
//                 kernel==stride convolutions (PatchEmbed, Attention.sr) as gathered-row GEMMs.
//                 Replaces F.linear / nn.Conv2d call sites of reference libs/pvlt.py:66-69,98,104,108,118,168
//                 and libs/vl_heads.py:31,67,85,102.
//  mvlt_gemm_tn : C[N1,N2] += A[M,N1]^T . B[M,N2]  (fp32 atomics)   weight gradients (+ fused bias gradient).
//
// Tiling: 256 threads = 4 waves (2x2), block tile 128 x BN (BN = 128 or 64), wave tile 64 x BN/2 built from
// v_mfma_f32_16x16x32_bf16 (bf16) or 8 x v_mfma_f32_16x16x4_f32 (fp32: exact-f32 path for the 1e-3 parity bar).
// LDS rows are 128 B (64 bf16 / 32 fp32 of K) = 8 x 16-B chunks, XOR-swizzled by row so the ds_read_b128
// fragment reads are conflict-free; global->register->LDS staging is double-buffered (one barrier per K tile).
#ifndef MVLT_GELU_POLY
#define MVLT_GELU_POLY 3     // GELU / GELU' of the bf16 GEMM epilogues (EPI 3 / 4) by the transcendental-free polynomials of common.h.  Round 3 measured no gain on the
#endif                       // 128-wide kernels (four workgroups per CU hide the epilogue's VALU work); on the 8-wave kernels (two waves per SIMD, VALU-bound epilogue)
                             // GELU' is 10 instructions instead of 19.6 issue units: stage-4 GELU' dgrad 179 -> 157 us, stage-3 204 -> 181 us (same-box A/B, round 4)
#include "common.h"
#include <type_traits>
#include <map>
#include <mutex>
#ifndef MVLT_NT_EARLY_DEFAULT
#define MVLT_NT_EARLY_DEFAULT 0x100  // early slot release in the NT K-loop: logits GEMM 172 -> 163 us, step -0.16 ms (same-box A/B, MVLT_NT_EARLY=0 / 1)
#endif
#ifndef MVLT_ABL
#define MVLT_ABL 0                   // timing ablations of the NT K-loop (wrong results): 1 no DMA, 2 no MFMA, 3 no fragment reads, 4 every DMA reads the zero page
#endif
#ifndef MVLT_TN_EARLY
#define MVLT_TN_EARLY 0
#endif
#include "../../include/mvlt_hip.h"

namespace {

constexpr int BM = 128;
constexpr int NTHREADS = 256;
constexpr int ROW_BYTES = 128;        // one LDS row = 128 B of K
constexpr int CHUNKS = 8;             // 16-B chunks per LDS row

template <typename T> struct Elem;
template <> struct Elem<bf16> { static constexpr int BK = 64; static constexpr int PER_CHUNK = 8; };
template <> struct Elem<float> { static constexpr int BK = 32; static constexpr int PER_CHUNK = 4; };

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ RowMap to_rowmap(const mvlt_rowmap& m) {
  RowMap r;
  r.mode = m.mode; r.rows_per_batch = m.rows_per_batch; r.batch_stride = m.batch_stride; r.offset = m.offset;
  r.r = m.r; r.w_in = m.w_in; r.tokens_in = m.tokens_in; r.hw_out = m.hw_out; r.w_out = m.w_out; r.c_seg = m.c_seg;
  r.h_in = m.h_in;
  return r;
}

// one k32 MFMA step on a 16x16 tile: a/b fragments = 8 consecutive K elements of row (lane&15) at K offset 8*(lane>>4)
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, bf16*) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc, 0, 0, 0);
  (void)a1; (void)b1;
}
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1, float*) {
  f32x4 fa0 = __builtin_bit_cast(f32x4, a0), fa1 = __builtin_bit_cast(f32x4, a1);
  f32x4 fb0 = __builtin_bit_cast(f32x4, b0), fb1 = __builtin_bit_cast(f32x4, b1);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[j], fb0[j], acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[j], fb1[j], acc, 0, 0, 0);
}

// out_dtype 2 (fp16; EPI 5 only: the pre-BatchNorm conv output z, which no MFMA reads -- 11 significand bits at half the bytes of fp32, the
// reference's own autocast type for it, reference libs/vl_heads.py:15-21 under torch.cuda.amp): values saturate at the fp16 range instead of
// turning into inf, and the column statistics are taken from the ROUNDED values (mean / rstd describe z as the normalisation reads it)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void round_f16_8(float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)(_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
}
template <bool NTF> __device__ __forceinline__ void store_f16_8(void* base, long ix, const float (&v)[8]) {
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (_Float16)v[e];
  st_g<NTF>((f16x8*)((_Float16*)base + ix), o);
}
template <typename T> __device__ __forceinline__ void store_out(void* base, long idx, float v, int out_fp32) {
  if (out_fp32) ((float*)base)[idx] = v; else ((bf16*)base)[idx] = (bf16)v;
}
__device__ __forceinline__ float load_out(const void* base, long idx, int out_fp32) {
  return out_fp32 ? ((const float*)base)[idx] : (float)((const bf16*)base)[idx];
}

// ---------------- NT epilogue (shared by the register-staged and the LDS-DMA main loops).  Must be entered after a
// workgroup barrier that follows the last operand-tile read: the staging below reuses the tile LDS.
template <typename T, int BN>
__device__ __forceinline__ void nt_epilogue(const mvlt_gemm_nt_args& p, f32x4 (&acc)[4][BN / 32], char* smem, int m0, int n0,
                                            int wave, int lane) {
  constexpr int WN = BN / 2;
  constexpr int TN_ = WN / 16;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  // ---------------- epilogue: acc[i][j][r] = C[m0 + wm*64 + i*16 + 4*fg + r][n0 + wn*WN + j*16 + fr]
  // The MFMA C layout gives each lane one column and 4 rows: stored directly that is 2-byte pieces, 32 B per row.
  // Instead every wave parks its tile in the (now free) staging LDS, 32 rows at a time, and re-reads it row-major:
  // each lane then owns 8 consecutive columns of one row, so bias / GELU / residual / H traffic and the C store are
  // 16-byte accesses that cover a full 128-B (bf16) or 256-B (fp32) row segment per 8 lanes.
  const RowMap cmap = to_rowmap(p.c_map);
  const int ofp32 = p.out_dtype;
  constexpr int LDW = WN + 4;                         // fp32 words per staged row (pad: <=2-way ds_write conflicts)
  constexpr int CPR = WN / 8;                         // 8-column chunks per row
  constexpr int RPI = 64 / CPR;                       // rows covered per wave iteration
  float* stage = (float*)smem + wave * 32 * LDW;
  const bool vec_ok = (p.ldc % 8 == 0) && (((uintptr_t)p.C & 15) == 0) && (!p.R || ((uintptr_t)p.R & 15) == 0) &&
                      (!p.H || ((uintptr_t)p.H & 15) == 0);
  // this lane's 8 output columns are the same in every iteration: fetch their bias once (two 16-B loads when aligned)
  const int nc_lane = n0 + wn * WN + (lane % CPR) * 8;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  if (p.bias && p.split_k <= 1) {
    if (nc_lane + 8 <= p.N && (((uintptr_t)p.bias & 15) == 0)) {
      f32x4 b0 = *(const f32x4*)(p.bias + nc_lane), b1 = *(const f32x4*)(p.bias + nc_lane + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (nc_lane + e < p.N) bias8[e] = p.bias[nc_lane + e];
    }
  }
  // optional per-column sum / sum of squares of the stored values (BatchNorm batch statistics of a conv output)
  float cs8[8], cq8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cs8[e] = 0.f; cq8[e] = 0.f; }
#pragma unroll
  // (the K loop ended with a workgroup barrier: nobody reads the operand tiles any more.  From here on every wave
  //  touches only its own staging slice, so only wave-level ordering is needed and the waves drift apart freely.)
  for (int half = 0; half < 2; ++half) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < TN_; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(ii * 16 + 4 * fg + r) * LDW + j * 16 + fr] = acc[half * 2 + ii][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (p.split_k > 1) {
      // partial tile of a K split: fp32 atomics into the caller-zeroed C, one staged row per instruction so that the
      // lanes of a wave hit consecutive floats (whole 64-B atomic requests; the row-major 8-columns-per-lane layout
      // below would touch every request eight times)
      const int col = n0 + wn * WN + lane;
      if (lane < WN && col < p.N) {
        const float bv = (p.bias && blockIdx.y == 0) ? p.bias[col] : 0.f;
        for (int r = 0; r < 32; ++r) {
          const int m = m0 + wm * 64 + half * 32 + r;
          if (m < p.M) atomicAdd((float*)p.C + (long)m * p.ldc + col, stage[r * LDW + lane] + bv);
        }
      }
      continue;
    }
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int rl = it * RPI + lane / CPR;           // row inside this 32-row half
      const int ch = lane % CPR;
      const int m = m0 + wm * 64 + half * 32 + rl;
      const int nc = n0 + wn * WN + ch * 8;           // first of this lane's 8 columns
      if (m >= p.M || nc >= p.N) continue;
      f32x4 v0 = *(const f32x4*)(stage + rl * LDW + ch * 8), v1 = *(const f32x4*)(stage + rl * LDW + ch * 8 + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      int seg_rows = 0, ncol = nc;
      if (cmap.mode == 1) {                           // scatter back through the patch map (dgrad of a kernel==stride conv)
        int seg = nc / cmap.c_seg;
        ncol = nc - seg * cmap.c_seg;
        seg_rows = rowmap_seg(cmap, seg);
      }
      const long idx = (rowmap_base(cmap, m) + seg_rows) * p.ldc + ncol;
      const bool full = vec_ok && (nc + 8 <= p.N);
      const float rs = p.row_scale ? p.row_scale[m / p.rows_per_scale] : 1.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias8[e];
      if (full) {
        auto load8 = [&](const void* base, float* o) {
          if (ofp32) {
            f32x4 a = *(const f32x4*)((const float*)base + idx), b = *(const f32x4*)((const float*)base + idx + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
          } else {
            bf16x8 a = *(const bf16x8*)((const bf16*)base + idx);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (float)a[e];
          }
        };
        auto store8 = [&](void* base, const float* o) {
          if (ofp32) {
            st_g<MVLT_NT_GEMM>((f32x4*)((float*)base + idx), f32x4{o[0], o[1], o[2], o[3]});
            st_g<MVLT_NT_GEMM>((f32x4*)((float*)base + idx + 4), f32x4{o[4], o[5], o[6], o[7]});
          } else {
            bf16x8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (bf16)o[e];
            st_g<MVLT_NT_GEMM>((bf16x8*)((bf16*)base + idx), a);
          }
        };
        if (p.act == 1) {
          if (p.H) store8(p.H, v);
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2 gv = gelu_erf2(f32x2{v[e], v[e + 1]});
            v[e] = gv[0]; v[e + 1] = gv[1];
          }
        } else if (p.act == 2) {
          float h8[8];
          load8(p.H, h8);
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2 dv = gelu_erf_grad2(f32x2{h8[e], h8[e + 1]});
            v[e] *= dv[0]; v[e + 1] *= dv[1];
          }
        }
        if (p.row_scale) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= rs;
        }
        if (p.R) {
          float r8[8];
          load8(p.R, r8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r8[e];
        }
        if (p.col_sum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { cs8[e] += v[e]; cq8[e] += v[e] * v[e]; }
        }
        store8(p.C, v);
      } else {                                        // ragged N (vocabulary tail, 2/48/122-way heads) or unaligned rows
        for (int e = 0; e < 8; ++e) {
          if (nc + e >= p.N) break;
          float x = v[e];
          if (p.act == 1) {
            if (p.H) store_out<T>(p.H, idx + e, x, ofp32);
            x = gelu_erf(x);
          } else if (p.act == 2) {
            x *= gelu_erf_grad(load_out(p.H, idx + e, ofp32));
          }
          x *= rs;
          if (p.R) x += load_out(p.R, idx + e, ofp32);
          if (p.col_sum) { cs8[e] += x; cq8[e] += x * x; }
          store_out<T>(p.C, idx + e, x, ofp32);
        }
      }
    }
  }
  if (p.col_sum) {
    // lanes with equal lane % CPR own the same 8 columns: fold them, then turn the CPR x 8 totals of this wave into one
    // 64-lane (32 for the narrow tile) atomic per statistic through the wave's staging slice
#pragma unroll
    for (int off = CPR; off < 64; off <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs8[e] += __shfl_xor(cs8[e], off); cq8[e] += __shfl_xor(cq8[e], off); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { stage[lane * 8 + e] = cs8[e]; stage[WN + lane * 8 + e] = cq8[e]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // every row tile of the launch adds into the same N floats; same-line atomics serialise at the memory side, so the
    // caller provides col_copies interleaved accumulators ([copy][N]) and sums them when it finalises the statistics
    const int col = n0 + wn * WN + lane;
    const long cpy = p.col_copies > 1 ? (long)((m0 / BM) % p.col_copies) * p.N : 0;
    if (lane < WN && col < p.N) {
      atomicAdd(&p.col_sum[cpy + col], stage[lane]);
      atomicAdd(&p.col_sumsq[cpy + col], stage[WN + lane]);
    }
  }
}

// ---------------- lean NT epilogues (bf16 LDS-DMA kernels).  The generic epilogue above decides every feature at run time
// inside the per-row loop and recomputes row / column addressing per 8 outputs: ~100 instructions per 8 outputs, 55 % of
// the stage-3 fc1 GEMM's time.  The call sites use five shapes of epilogue; each gets a compile-time variant in which the
// column side (bias, chunk index) is fixed per lane, the row side is a multiply, and the operands a variant needs (R, H,
// DropPath factor) are requested for a whole 32-row half before it is processed.  Preconditions (checked by the host
// dispatch): identity, batch-strided or patch-scatter c_map, N % 8 == 0, ldc % 8 == 0, 16-byte aligned C / R / H, no split-K.
//   EPI 1: C = AB^T (+bias)                      EPI 2: C = (AB^T + bias) * row_scale + R
//   EPI 3: H = AB^T + bias ; C = gelu(H)         EPI 4: C = AB^T * gelu'(H)          EPI 5: EPI 1 + column sum / sum of squares
//   EPI 6 / 7: EPI 1 / 2 written through a patch-scatter c_map (dgrad of the kernel==stride convs)
//   EPI 8: EPI 2 + LayerNorm of the finished row (N == BN: the tile holds whole rows, split over the two waves of a row pair):
//          post_y = LN(C row; post_gamma, post_beta, post_eps) in bf16 + the row statistics -- Block.norm2 behind attn.proj
//          (reference libs/pvlt.py:140-142) without a pass of its own over the fp32 mid stream
__device__ __forceinline__ int fdiv24(int m, int d, float inv) {      // exact m / d for 0 <= m < 2^24, inv = 1.0f / d
  int q = (int)((float)m * inv);
  int r = m - q * d;
  return q + (r >= d) - (r < 0);
}
// NWN = waves along N (2: the 4-wave 2 x 2 kernels; 4: the 8-wave 256 x 256 kernel, whose wave tile is (TM * 16) x 64 = "BN 128" here)
// FULL: the launch has whole tiles only (no row / column bound checks: the 8-wave kernels, whose epilogue is issue-bound at two waves per SIMD)
template <int BN, int EPIX, int TM = 4, int NWN = 2, bool FULL = false>
__device__ __forceinline__ void nt_epilogue_lean(const mvlt_gemm_nt_args& p, f32x4 (&acc)[TM][BN / 32], char* smem, int m0, int n0,
                                                 int wave, int lane) {
  static_assert(NWN == 2 || EPIX != 8, "the LayerNorm epilogue pairs the two waves of a row: 2 x 2 wave grids only");
  constexpr bool SCAT = (EPIX == 6 || EPIX == 7);     // EPI 6 / 7 = EPI 1 / 2 with a patch-scatter c_map (mode 1)
  constexpr bool POST = EPIX == 8;                    // EPI 8 = EPI 2 + LayerNorm of the output row
  constexpr int EPI = EPIX == 6 ? 1 : (EPIX == 7 || EPIX == 8) ? 2 : EPIX;
  constexpr int WN = BN / 2;
  constexpr int TN_ = WN / 16;
  constexpr int LDW = WN + 4;
  constexpr int CPR = WN / 8;
  constexpr int RPI = 64 / CPR;
  constexpr int NIT = 32 / RPI;
  const int wm = wave / NWN, wn = wave % NWN;
  const int fr = lane & 15, fg = lane >> 4;
  // the GELU / GELU' epilogues write (and read) operand-dtype tensors only -- the host dispatch guarantees it -- so their fp32 store / load
  // paths and the second half of every prefetch slot are not compiled in (EPI 4: 119 -> fewer registers, a third workgroup per CU)
  const int ofp32 = (EPI == 3 || EPI == 4) ? 0 : p.out_dtype;
  const int rfp32 = (EPI == 2 && !POST) ? (ofp32 | p.r_fp32) : ofp32;      // EPI 2: the residual may be fp32 beside a bf16 C (mvlt_gemm_nt_args.r_fp32)
  float* stage = (float*)smem + wave * 32 * LDW;
  const int ch = lane % CPR;
  const int nc = n0 + wn * WN + ch * 8;
  const bool col_ok = FULL || nc < p.N;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
  if (p.bias && col_ok) {
    if (!FULL && nc + 8 > p.N) {                       // the last, partial chunk of a row whose length is not a multiple of 8 (EPI 1 only: host dispatch)
#pragma unroll
      for (int e = 0; e < 8; ++e) if (nc + e < p.N) bias8[e] = p.bias[nc + e];
    } else if (((uintptr_t)p.bias & 15) == 0) {
      f32x4 b0 = *(const f32x4*)(p.bias + nc), b1 = *(const f32x4*)(p.bias + nc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) bias8[e] = p.bias[nc + e];
    }
  }
  constexpr int NH = TM / 2;                          // 32-row halves of the wave tile (2, or 4 under the 256-row tile)
  const int m_first = m0 + wm * (TM * 16) + lane / CPR;
  const int rpb = p.c_map.rows_per_batch;
  const float inv_rpb = rpb > 0 ? 1.0f / (float)rpb : 0.f;
  // patch scatter (c_map mode 1, dgrad of a kernel==stride conv): this lane's 8 columns lie in one (di, dj) segment
  int seg_rows = 0, ncol = nc;
  float inv_hw = 0.f, inv_w = 0.f;
  if constexpr (SCAT) {
    const int seg = nc / p.c_map.c_seg;
    ncol = nc - seg * p.c_map.c_seg;
    const int di = seg / p.c_map.r;
    seg_rows = di * p.c_map.w_in + (seg - di * p.c_map.r);
    inv_hw = 1.0f / (float)p.c_map.hw_out;
    inv_w = 1.0f / (float)p.c_map.w_out;
  }
  const float inv_rps = (EPI == 2 && p.rows_per_scale > 0) ? 1.0f / (float)p.rows_per_scale : 0.f;
  float cs8[8], cq8[8];
  if (EPI == 5) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs8[e] = 0.f; cq8[e] = 0.f; }
  }
  // R (EPI 2) / H (EPI 4) of two 32-row halves are requested before the first one is staged (two rotating slots): their HBM
  // latency hides behind the LDS staging and the previous half instead of being paid once per half
  long idx[2][NIT];
  bool ok[2][NIT];
  float rs[2][NIT];
  u32x4 raw[2][NIT][2];
  auto request = [&](int half) {
    const int sl = half & 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m_first + half * 32 + it * RPI;
      ok[sl][it] = FULL || (m < p.M && col_ok);
      const int mm = (FULL || ok[sl][it]) ? m : 0;
      long phys = mm;
      if constexpr (SCAT) {
        const int b = fdiv24(mm, p.c_map.hw_out, inv_hw);
        const int rem = mm - b * p.c_map.hw_out;
        const int oi = fdiv24(rem, p.c_map.w_out, inv_w), oj = rem - oi * p.c_map.w_out;
        phys = (long)b * p.c_map.tokens_in + (long)(oi * p.c_map.r) * p.c_map.w_in + oj * p.c_map.r + seg_rows;
      } else if (rpb > 0) {
        const int b = fdiv24(mm, rpb, inv_rpb);
        phys = (long)b * p.c_map.batch_stride + p.c_map.offset + (mm - b * rpb);
      }
      idx[sl][it] = phys * p.ldc + ncol;
      rs[sl][it] = 1.0f;
      if (EPI == 2 && p.row_scale) rs[sl][it] = p.row_scale[fdiv24(mm, p.rows_per_scale, inv_rps)];
      if ((EPI == 2 || EPI == 4) && (FULL || ok[sl][it])) {
        const void* src = (EPI == 2) ? p.R : p.H;
        if (rfp32) { raw[sl][it][0] = ld_g<MVLT_NT_LD>((const u32x4*)((const float*)src + idx[sl][it])); raw[sl][it][1] = ld_g<MVLT_NT_LD>((const u32x4*)((const float*)src + idx[sl][it] + 4)); }
        else raw[sl][it][0] = ld_g<MVLT_NT_LD>((const u32x4*)((const bf16*)src + idx[sl][it]));
      }
    }
  };
  constexpr bool PREFETCH = (EPI == 2 || EPI == 4);
  if (PREFETCH) { request(0); request(1); }
  float pgam[8], pbet[8];
  float* const xch = (float*)smem + 4 * 32 * LDW;    // POST: [128 rows][2 column halves] row sums | the same for the squared deviations
  if constexpr (POST) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { pgam[e] = p.post_gamma[nc + e]; pbet[e] = p.post_beta[nc + e]; }
  }
#pragma unroll
  for (int half = 0; half < NH; ++half) {
    const int sl = half & 1;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < TN_; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(ii * 16 + 4 * fg + r) * LDW + j * 16 + fr] = acc[half * 2 + ii][j][r];
    if (!PREFETCH) request(half);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (POST) {
      // the row's other half lives in the partner wave (wn ^ 1): two-pass statistics with one LDS exchange + workgroup barrier per pass
      const float inv_n = 1.0f / (float)p.N;
      float keep[NIT][8];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int rl = it * RPI + lane / CPR;
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) keep[it][e] = 0.f;
        if (ok[sl][it]) {
          const f32x4 v0 = *(const f32x4*)(stage + rl * LDW + ch * 8), v1 = *(const f32x4*)(stage + rl * LDW + ch * 8 + 4);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          float o8[8];
          if (ofp32) {
            const f32x4 a = __builtin_bit_cast(f32x4, raw[sl][it][0]), b = __builtin_bit_cast(f32x4, raw[sl][it][1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o8[e] = a[e]; o8[4 + e] = b[e]; }
          } else {
            const bf16x8 a = __builtin_bit_cast(bf16x8, raw[sl][it][0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = (float)a[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[e] = (v[e] + bias8[e]) * rs[sl][it] + o8[e]; keep[it][e] = v[e]; sum += v[e]; }
          const long ix = idx[sl][it];
          if (ofp32) {
            st_g<MVLT_NT_GEMM>((f32x4*)((float*)p.C + ix), f32x4{v[0], v[1], v[2], v[3]});
            st_g<MVLT_NT_GEMM>((f32x4*)((float*)p.C + ix + 4), f32x4{v[4], v[5], v[6], v[7]});
          } else {
            bf16x8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (bf16)v[e];
            st_g<MVLT_NT_GEMM>((bf16x8*)((bf16*)p.C + ix), a);
          }
        }
#pragma unroll
        for (int o = 1; o < CPR; o <<= 1) sum += __shfl_xor(sum, o);
        if (ch == 0) xch[((wm * 2 + half) * 32 + rl) * 2 + wn] = sum;
      }
      __syncthreads();
      float mean[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = (wm * 2 + half) * 32 + it * RPI + lane / CPR;
        mean[it] = (xch[row * 2] + xch[row * 2 + 1]) * inv_n;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { keep[it][e] -= mean[it]; q += keep[it][e] * keep[it][e]; }
#pragma unroll
        for (int o = 1; o < CPR; o <<= 1) q += __shfl_xor(q, o);
        if (ch == 0) xch[256 + row * 2 + wn] = q;
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        if (!ok[sl][it]) continue;
        const int rl = it * RPI + lane / CPR;
        const int row = (wm * 2 + half) * 32 + rl;
        const int m = m_first + half * 32 + it * RPI;
        const float rstd = rsqrtf((xch[256 + row * 2] + xch[256 + row * 2 + 1]) * inv_n + p.post_eps);
        bf16x8 y;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (bf16)(keep[it][e] * rstd * pgam[e] + pbet[e]);
        st_g<MVLT_NT_GEMM>((bf16x8*)((bf16*)p.post_y + (long)m * p.post_ld + nc), y);
        if (wn == 0 && ch == 0) { p.post_mean[m] = mean[it]; p.post_rstd[m] = rstd; }
      }
      if (PREFETCH && half + 2 < NH) request(half + 2);
      continue;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (!FULL && !ok[sl][it]) continue;
      const int rl = it * RPI + lane / CPR;
      const f32x4 v0 = *(const f32x4*)(stage + rl * LDW + ch * 8), v1 = *(const f32x4*)(stage + rl * LDW + ch * 8 + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bias8[e];
      const long ix = idx[sl][it];
      auto store8 = [&](void* base, const float* o) {
        if (EPI == 5 && ofp32 == 2) {
          const float o8h[8] = {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
          store_f16_8<MVLT_NT_GEMM>(base, ix, o8h);
        } else if (ofp32) {
          st_g<MVLT_NT_GEMM>((f32x4*)((float*)base + ix), f32x4{o[0], o[1], o[2], o[3]});
          st_g<MVLT_NT_GEMM>((f32x4*)((float*)base + ix + 4), f32x4{o[4], o[5], o[6], o[7]});
        } else {
          bf16x8 a;
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = (bf16)o[e];
          st_g<MVLT_NT_GEMM>((bf16x8*)((bf16*)base + ix), a);
        }
      };
      float o8[8];
      if (EPI == 2 || EPI == 4) {
        if (rfp32) {
          const f32x4 a = __builtin_bit_cast(f32x4, raw[sl][it][0]), b = __builtin_bit_cast(f32x4, raw[sl][it][1]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { o8[e] = a[e]; o8[4 + e] = b[e]; }
        } else {
          const bf16x8 a = __builtin_bit_cast(bf16x8, raw[sl][it][0]);
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = (float)a[e];
        }
      }
      if (EPI == 3) {
        if (p.H) store8(p.H, v);
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 gv = gelu_fast2(f32x2{v[e], v[e + 1]});
          v[e] = gv[0]; v[e + 1] = gv[1];
        }
      }
      if (EPI == 4) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const f32x2 dv = gelu_fast_grad2(f32x2{o8[e], o8[e + 1]});
          v[e] *= dv[0]; v[e + 1] *= dv[1];
        }
      }
      if (EPI == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * rs[sl][it] + o8[e];
      }
      if (EPI == 5) {
        if (ofp32 == 2) round_f16_8(v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { cs8[e] += v[e]; cq8[e] += v[e] * v[e]; }
      }
      if constexpr (EPI == 1 && !FULL && !SCAT) {
        if (nc + 8 > p.N) {                            // partial last chunk: its valid columns one by one, nothing past column N - 1 is touched
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (nc + e < p.N) { if (ofp32) ((float*)p.C)[ix + e] = v[e]; else ((bf16*)p.C)[ix + e] = (bf16)v[e]; }
          continue;
        }
      }
      store8(p.C, v);
    }
    if (PREFETCH && half + 2 < NH) request(half + 2);
  }
  if (EPI == 5) {
#pragma unroll
    for (int off = CPR; off < 64; off <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs8[e] += __shfl_xor(cs8[e], off); cq8[e] += __shfl_xor(cq8[e], off); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { stage[lane * 8 + e] = cs8[e]; stage[WN + lane * 8 + e] = cq8[e]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int col = n0 + wn * WN + lane;
    const long cpy = p.col_copies > 1 ? (long)((m0 / BM) % p.col_copies) * p.N : 0;
    if (lane < WN && col < p.N) {
      atomicAdd(&p.col_sum[cpy + col], stage[lane]);
      atomicAdd(&p.col_sumsq[cpy + col], stage[WN + lane]);
    }
  }
}

// BN = 192 (wave tile 64 x 96, for N % 192 == 0 outputs that the 128-wide tile would pad to the next 256): the wave's 96 columns
// leave as three 32-column groups, 4 lanes x 8 columns per row and 16 rows per instruction.  Only the two epilogues the
// 192-channel convolutions use: EPI 1 (plain (+bias)) and EPI 5 (the same + column sum / sum of squares).
template <int EPI>
__device__ __forceinline__ void nt_epilogue_192(const mvlt_gemm_nt_args& p, f32x4 (&acc)[4][6], char* smem, int m0, int n0, int wave,
                                                int lane) {
  static_assert(EPI == 1 || EPI == 5, "192-wide tiles carry the plain and the column-statistics epilogue only");
  constexpr int WN = 96, TN_ = 6, LDW = WN + 4, NG = 3;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int ofp32 = p.out_dtype;
  float* stage = (float*)smem + wave * 32 * LDW;
  const int ch = lane & 3, rl0 = lane >> 2;
  const int nbase = n0 + wn * WN + ch * 8;
  const int rpb = p.c_map.rows_per_batch;
  const float inv_rpb = rpb > 0 ? 1.0f / (float)rpb : 0.f;
  float cs[NG][8], cq[NG][8];
  if (EPI == 5) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs[g][e] = 0.f; cq[g][e] = 0.f; }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < TN_; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(ii * 16 + 4 * fg + r) * LDW + j * 16 + fr] = acc[half * 2 + ii][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int rl = it * 16 + rl0;
      const int m = m0 + wm * 64 + half * 32 + rl;
      if (m >= p.M) continue;
      long phys = m;
      if (rpb > 0) {
        const int b = fdiv24(m, rpb, inv_rpb);
        phys = (long)b * p.c_map.batch_stride + p.c_map.offset + (m - b * rpb);
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int nc = nbase + g * 32;
        if (nc >= p.N) continue;
        const f32x4 v0 = *(const f32x4*)(stage + rl * LDW + g * 32 + ch * 8), v1 = *(const f32x4*)(stage + rl * LDW + g * 32 + ch * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += p.bias[nc + e];
        }
        if (EPI == 5) {
          if (ofp32 == 2) round_f16_8(v);
#pragma unroll
          for (int e = 0; e < 8; ++e) { cs[g][e] += v[e]; cq[g][e] += v[e] * v[e]; }
        }
        const long ix = phys * p.ldc + nc;
        if (EPI == 5 && ofp32 == 2) {
          store_f16_8<MVLT_NT_GEMM>(p.C, ix, v);
        } else if (ofp32) {
          st_g<MVLT_NT_GEMM>((f32x4*)((float*)p.C + ix), f32x4{v[0], v[1], v[2], v[3]});
          st_g<MVLT_NT_GEMM>((f32x4*)((float*)p.C + ix + 4), f32x4{v[4], v[5], v[6], v[7]});
        } else {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          st_g<MVLT_NT_GEMM>((bf16x8*)((bf16*)p.C + ix), o);
        }
      }
    }
  }
  if (EPI == 5) {
#pragma unroll
    for (int off = 4; off < 64; off <<= 1)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) { cs[g][e] += __shfl_xor(cs[g][e], off); cq[g][e] += __shfl_xor(cq[g][e], off); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 4) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) { stage[g * 32 + lane * 8 + e] = cs[g][e]; stage[WN + g * 32 + lane * 8 + e] = cq[g][e]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const long cpy = p.col_copies > 1 ? (long)((m0 / BM) % p.col_copies) * p.N : 0;
    for (int c = lane; c < WN; c += 64) {
      const int col = n0 + wn * WN + c;
      if (col < p.N) {
        atomicAdd(&p.col_sum[cpy + col], stage[c]);
        atomicAdd(&p.col_sumsq[cpy + col], stage[WN + c]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ NT
template <typename T, int BN>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(mvlt_gemm_nt_args p, int nbuf) {
  constexpr int BK = Elem<T>::BK;
  constexpr int PC = Elem<T>::PER_CHUNK;
  constexpr int WN = BN / 2;            // wave tile N
  constexpr int TN_ = WN / 16;          // 16-wide MFMA tiles per wave in N
  constexpr int A_ITERS = BM * CHUNKS / NTHREADS;   // 4
  constexpr int B_ITERS = BN * CHUNKS / NTHREADS;   // 4 or 2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                                   // nbuf x BM x 128 B  (nbuf = 1 when K fits one tile: 2x the
  char* sB = smem + nbuf * BM * ROW_BYTES;           // nbuf x BN x 128 B   workgroups per CU for the K=64 stage-1 GEMMs)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order (block b runs on XCD b % 8): the column tiles of one 128-row panel get consecutive slots on ONE XCD,
  // so the A panel is fetched into that XCD's L2 once and a row block of C is written as a whole (all its column tiles
  // close together in time) instead of as 256-byte slivers revisited tiles_m workgroups later.
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int tile_m = (slot / tiles_n) * 8 + xcd, tile_n = slot % tiles_n;
  if (tile_m >= tiles_m) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const RowMap amap = to_rowmap(p.a_map);

  const int chunk = tid & 7;
  const int row_in = tid >> 3;          // 0..31
  const T* Ag = (const T*)p.A;
  const T* Bg = (const T*)p.B;

  long a_base[A_ITERS];
  bool a_ok[A_ITERS];
  int a_y[A_ITERS], a_x[A_ITERS];       // mode 2 only: pixel coordinates (zero-padding test per 3x3 tap)
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    int m = m0 + row_in + 32 * i;
    a_ok[i] = m < p.M;
    a_base[i] = a_ok[i] ? rowmap_base(amap, m) : 0;
    a_y[i] = 0; a_x[i] = 0;
    if (amap.mode == 2 && a_ok[i]) rowmap_yx(amap, m, a_y[i], a_x[i]);
  }
  long b_base[B_ITERS];
  bool b_ok[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    int n = n0 + row_in + 32 * i;
    b_ok[i] = n < p.N;
    b_base[i] = b_ok[i] ? (long)n * p.ldb : 0;
  }

  u32x4 ra[A_ITERS], rb[B_ITERS];
  auto gload = [&](int kt) {
    const int k = kt * BK + chunk * PC;
    const bool k_ok = k < p.K;          // K % PER_CHUNK == 0 is required by the host wrapper
    int seg_rows = 0, kk = k, seg = 0;
    if (amap.mode != 0) {
      seg = k / amap.c_seg;
      kk = k - seg * amap.c_seg;
      if (amap.mode == 1) seg_rows = rowmap_seg(amap, seg);
    }
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      bool ok = a_ok[i] && k_ok;
      int off = seg_rows;
      if (amap.mode == 2) ok = ok && rowmap_nb(amap, seg, a_y[i], a_x[i], off);
      if (ok) v = *(const u32x4*)(Ag + (a_base[i] + off) * p.lda + kk);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (b_ok[i] && k_ok) v = *(const u32x4*)(Bg + b_base[i] + k);
      rb[i] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      int r = row_in + 32 * i;
      *(u32x4*)(sA + buf * BM * ROW_BYTES + r * ROW_BYTES + swz(r, chunk) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      int r = row_in + 32 * i;
      *(u32x4*)(sB + buf * BN * ROW_BYTES + r * ROW_BYTES + swz(r, chunk) * 16) = rb[i];
    }
  };

  f32x4 acc[4][TN_];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* a_s = sA + buf * BM * ROW_BYTES + (wm * 64) * ROW_BYTES;
    const char* b_s = sB + buf * BN * ROW_BYTES + (wn * WN) * ROW_BYTES;
    constexpr int KSTEPS = (sizeof(T) == 2) ? 2 : 1;     // k32 steps per LDS tile
    constexpr int CPS = (sizeof(T) == 2) ? 1 : 2;        // 16-B chunks per fragment
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      u32x4 fa[4][2], fb[TN_][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int r = i * 16 + fr;
        int rr = wm * 64 + r;           // swizzle uses the tile-level row
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
          int ch = (sizeof(T) == 2) ? (ks * 4 + fg) : (fg * 2 + c);
          fa[i][c] = *(const u32x4*)(a_s + r * ROW_BYTES + swz(rr, ch) * 16);
        }
        if (CPS == 1) fa[i][1] = fa[i][0];
      }
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        int r = j * 16 + fr;
        int rr = wn * WN + r;
#pragma unroll
        for (int c = 0; c < CPS; ++c) {
          int ch = (sizeof(T) == 2) ? (ks * 4 + fg) : (fg * 2 + c);
          fb[j][c] = *(const u32x4*)(b_s + r * ROW_BYTES + swz(rr, ch) * 16);
        }
        if (CPS == 1) fb[j][1] = fb[j][0];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j) mma16(acc[i][j], fa[i][0], fa[i][1], fb[j][0], fb[j][1], (T*)nullptr);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  nt_epilogue<T, BN>(p, acc, smem, m0, n0, wave, lane);
}

// ------------------------------------------------------------------------------------------------ TN (wgrad)
// C[n1, n2] += sum_m A[m, n1] * B[m, n2].  The reduction index m is the slow (row) index of both operands, so
// tiles are transposed on their way into LDS (At[n1][m], Bt[n2][m], 64 m per tile) and the MFMA fragments again
// read 8 consecutive m.  The m range is split across gridDim.z; partial tiles are combined with fp32 atomics.
constexpr int TBK = 64;                         // m per LDS tile (both dtypes)
template <typename T> struct TElem;
template <> struct TElem<bf16> { static constexpr int ROWB = (TBK + 8) * 2; };     // 144 B rows (pad keeps 16-B alignment)
template <> struct TElem<float> { static constexpr int ROWB = (TBK + 4) * 4; };    // 272 B rows

// bf16 transposed tiles hold (m, m+1) pairs as 32-bit words: row n = output index, 32 pair-columns.  Every 8 rows share
// the 4 low bank bits (row stride 144 B = 36 dwords), so the 16 lanes that write the same pair-column of 16 different
// row-groups would hit ONE bank; XOR-ing the pair-column with the row-group index (in units of 4 pairs = one 16-B
// fragment read, which therefore stays contiguous) spreads them over 8 banks (2-way on ds_write_b32 is free).
__device__ __forceinline__ int tsw(int n, int pair) { return pair ^ (((n >> 3) & 7) << 2); }

// destination column of logical output column n2 (mvlt_gemm_tn_args.c_taps / c_seg: [tap][cin] -> [cin][tap])
__device__ __forceinline__ int tn_dst_col(const mvlt_gemm_tn_args& p, int n2) {
  if (p.c_taps <= 1) return n2;
  const int tap = n2 / p.c_seg;
  return (n2 - tap * p.c_seg) * p.c_taps + tap;
}

template <typename T, int BN>
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(mvlt_gemm_tn_args p, int m_per_split) {
  constexpr int PC = Elem<T>::PER_CHUNK;             // elements per 16-B global chunk
  constexpr int ROWB = TElem<T>::ROWB;
  constexpr int WN = BN / 2;
  constexpr int TN_ = WN / 16;
  constexpr int A_CH = BM / PC;                      // chunks per tile row (A): 16 (bf16) / 32 (fp32)
  constexpr int B_CH = BN / PC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                                   // BM rows (n1) x ROWB
  char* sB = smem + BM * ROWB;                       // BN rows (n2) x ROWB
  float* s_colsum = (float*)(smem + (BM + BN) * ROWB);   // [BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n1_0 = blockIdx.x * BM, n2_0 = blockIdx.y * BN;
  const int m_begin = blockIdx.z * m_per_split;
  const int m_end = min(p.M, m_begin + m_per_split);
  const RowMap amap = to_rowmap(p.a_map), bmap = to_rowmap(p.b_map);
  const T* Ag = (const T*)p.A;
  const T* Bg = (const T*)p.B;
  const bool do_colsum = p.colsum_a != nullptr && blockIdx.y == 0;
  const bool do_colsum_b = p.colsum_b != nullptr && blockIdx.x == 0;
  if ((do_colsum || do_colsum_b) && tid < BM) s_colsum[tid] = 0.f;

  f32x4 acc[4][TN_];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // loader geometry: a "unit" = one 16-B chunk of columns x 2 consecutive m rows (bf16) or 1 row (fp32)
  constexpr int RPU = (sizeof(T) == 2) ? 2 : 1;      // m rows per unit
  constexpr int A_UNITS = A_CH * (TBK / RPU);
  constexpr int B_UNITS = B_CH * (TBK / RPU);
  constexpr int A_IT = A_UNITS / NTHREADS;           // bf16: 16*32/256 = 2 ; fp32: 32*64/256 = 8
  constexpr int B_IT = (B_UNITS + NTHREADS - 1) / NTHREADS;

  // per-thread fixed column chunk for B (patch gather resolves the segment once)
  const int fr = lane & 15, fg = lane >> 4;
  float colsum_local[PC], colsum_local_b[PC];
#pragma unroll
  for (int e = 0; e < PC; ++e) { colsum_local[e] = 0.f; colsum_local_b[e] = 0.f; }

  u32x4 va[A_IT][RPU], vb[B_IT][RPU];
  auto gload = [&](int mt) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int u = tid + it * NTHREADS;
      int c = u % A_CH, pr = u / A_CH;
      int n1 = n1_0 + c * PC;
#pragma unroll
      for (int q = 0; q < RPU; ++q) {
        int m = mt + pr * RPU + q;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m < m_end && n1 < p.N1) v = *(const u32x4*)(Ag + rowmap_base(amap, m) * p.lda + n1);
        va[it][q] = v;
      }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      int u = tid + it * NTHREADS;
      int c = u % B_CH, pr = u / B_CH;
      int n2 = n2_0 + c * PC;
      int seg_rows = 0, col = n2, seg = 0;
      if (bmap.mode != 0) {
        seg = n2 / bmap.c_seg;
        col = n2 - seg * bmap.c_seg;
        if (bmap.mode == 1) seg_rows = rowmap_seg(bmap, seg);
      }
#pragma unroll
      for (int q = 0; q < RPU; ++q) {
        int m = mt + pr * RPU + q;
        u32x4 v = {0u, 0u, 0u, 0u};
        bool ok = u < B_UNITS && m < m_end && n2 < p.N2;
        int off = seg_rows;
        if (bmap.mode == 2 && ok) {
          int y, x;
          rowmap_yx(bmap, m, y, x);
          ok = rowmap_nb(bmap, seg, y, x, off);
        }
        if (ok) v = *(const u32x4*)(Bg + (rowmap_base(bmap, m) + off) * p.ldb + col);
        vb[it][q] = v;
      }
    }
  };
  gload(m_begin);
  for (int mt = m_begin; mt < m_end; mt += TBK) {
    __syncthreads();          // previous tile's fragment reads are done
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int u = tid + it * NTHREADS;
      int c = u % A_CH, pr = u / A_CH;
      if constexpr (sizeof(T) == 2) {
        bf16x8 x0 = __builtin_bit_cast(bf16x8, va[it][0]), x1 = __builtin_bit_cast(bf16x8, va[it][1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bf16x2 pk = {x0[e], x1[e]};
          *(bf16x2*)(sA + (c * 8 + e) * ROWB + tsw(c * 8 + e, pr) * 4) = pk;
          if (do_colsum) colsum_local[e] += (float)x0[e] + (float)x1[e];
        }
      } else {
        f32x4 x0 = __builtin_bit_cast(f32x4, va[it][0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          *(float*)(sA + (c * 4 + e) * ROWB + pr * 4) = x0[e];
          if (do_colsum) colsum_local[e] += x0[e];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      int u = tid + it * NTHREADS;
      if (u >= B_UNITS) continue;
      int c = u % B_CH, pr = u / B_CH;
      if constexpr (sizeof(T) == 2) {
        bf16x8 x0 = __builtin_bit_cast(bf16x8, vb[it][0]), x1 = __builtin_bit_cast(bf16x8, vb[it][1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bf16x2 pk = {x0[e], x1[e]};
          *(bf16x2*)(sB + (c * 8 + e) * ROWB + tsw(c * 8 + e, pr) * 4) = pk;
          if (do_colsum_b) colsum_local_b[e] += (float)x0[e] + (float)x1[e];
        }
      } else {
        f32x4 x0 = __builtin_bit_cast(f32x4, vb[it][0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          *(float*)(sB + (c * 4 + e) * ROWB + pr * 4) = x0[e];
          if (do_colsum_b) colsum_local_b[e] += x0[e];
        }
      }
    }
    __syncthreads();
    if (mt + TBK < m_end) gload(mt + TBK);       // next tile's HBM reads fly while this tile's MFMAs run
    // fragments: row = n1 (or n2) index, 8 consecutive m at offset 32*ks + 8*fg
#pragma unroll
    for (int ks = 0; ks < TBK / 32; ++ks) {
      u32x4 fa[4][2], fb[TN_][2];
      constexpr int EB = sizeof(T);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = wm * 64 + i * 16 + fr;
        const char* ptr = (EB == 2) ? sA + n * ROWB + tsw(n, ks * 16 + fg * 4) * 4 : sA + n * ROWB + (ks * 32 + fg * 8) * EB;
        fa[i][0] = *(const u32x4*)ptr;
        fa[i][1] = (EB == 4) ? *(const u32x4*)(ptr + 16) : fa[i][0];
      }
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        const int n = wn * WN + j * 16 + fr;
        const char* ptr = (EB == 2) ? sB + n * ROWB + tsw(n, ks * 16 + fg * 4) * 4 : sB + n * ROWB + (ks * 32 + fg * 8) * EB;
        fb[j][0] = *(const u32x4*)ptr;
        fb[j][1] = (EB == 4) ? *(const u32x4*)(ptr + 16) : fb[j][0];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j) mma16(acc[i][j], fa[i][0], fa[i][1], fb[j][0], fb[j][1], (T*)nullptr);
    }
  }

  if (do_colsum) {
    // every A unit of this thread has the same column chunk c (A_CH divides NTHREADS)
    int c = tid % A_CH;
#pragma unroll
    for (int e = 0; e < PC; ++e) atomicAdd(&s_colsum[c * PC + e], colsum_local[e]);
    __syncthreads();
    if (tid < BM && n1_0 + tid < p.N1) atomicAdd(&p.colsum_a[n1_0 + tid], s_colsum[tid]);
  }
  if (do_colsum_b) {                 // only one of colsum_a / colsum_b is used per call (host wrapper checks)
    if (B_IT * NTHREADS == B_UNITS || tid < B_UNITS) {
      int c = tid % B_CH;
#pragma unroll
      for (int e = 0; e < PC; ++e) atomicAdd(&s_colsum[c * PC + e], colsum_local_b[e]);
    }
    __syncthreads();
    if (tid < BN && n2_0 + tid < p.N2) atomicAdd(&p.colsum_b[n2_0 + tid], s_colsum[tid]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n1 = n1_0 + wm * 64 + i * 16 + 4 * fg + r;
        int n2 = n2_0 + wn * WN + j * 16 + fr;
        if (n1 < p.N1 && n2 < p.N2) atomicAdd(p.trans_c ? &p.C[(long)n2 * p.ldc + n1] : &p.C[(long)n1 * p.ldc + tn_dst_col(p, n2)], acc[i][j][r]);
      }
}

// ------------------------------------------------------------------------------------------------ TN, bf16, LDS-DMA
// Same contract as gemm_tn_kernel<bf16,BN>, different data path: tiles stay in their natural [m][n] layout and are
// filled by global_load_lds_dwordx4 (no VGPR round trip, no ds_write: the 2-byte transposing stores of the kernel
// above cost as many LDS cycles as the MFMAs they feed); the MFMA fragments (8 consecutive m of one n per lane) come
// from two ds_read_b64_tr_b16 each.  LDS-DMA writes lane-linear (base + 16*lane), so the bank swizzle is applied to
// the SOURCE column chunk: LDS slot s of tile row m holds global 16-B chunk s ^ (h(m) << 1), h spreading the 8 rows a
// 32-lane read group touches (m..m+3 and m+8..m+11) over the eight 32-B bank windows.  Rows past m_end / columns past
// N / out-of-image 3x3 taps are zero-filled by the owning lane.  Two buffers, one barrier per 64-row tile.  Column sums
// (bias gradients) are one extra MFMA against an all-ones fragment, shared between the two waves of a tile row.
// Logical row m of a row map as (b, y, x) digits, advanced by a fixed stride with carries instead of divisions: the
// loader of a thread visits rows m0, m0+64, m0+128, ... and one division per row would cost more than the MFMAs it
// feeds.  plain rows: x = m; batch-strided rows: (b, x) base rows_per_batch; patch / 3x3 maps: (b, y, x) in the
// output grid.
struct RowIt { int b, y, x; };
struct RowStep { int w, h, db, dy, dx; };
__device__ __forceinline__ RowStep row_step(const RowMap& rm, int delta) {
  RowStep s;
  if (rm.mode == 0) {
    if (rm.rows_per_batch == 0) { s.w = 0x7fffffff; s.h = 1; s.db = 0; s.dy = 0; s.dx = delta; }
    else { s.w = rm.rows_per_batch; s.h = 1; s.db = delta / s.w; s.dy = 0; s.dx = delta - s.db * s.w; }
  } else {
    s.w = rm.w_out; s.h = rm.hw_out / rm.w_out;
    s.db = delta / rm.hw_out;
    int rem = delta - s.db * rm.hw_out;
    s.dy = rem / s.w; s.dx = rem - s.dy * s.w;
  }
  return s;
}
__device__ __forceinline__ RowIt row_init(const RowMap& rm, int m) {
  RowIt it;
  if (rm.mode == 0) {
    if (rm.rows_per_batch == 0) { it.b = 0; it.y = 0; it.x = m; }
    else { it.b = m / rm.rows_per_batch; it.y = 0; it.x = m - it.b * rm.rows_per_batch; }
  } else {
    it.b = m / rm.hw_out;
    int rem = m - it.b * rm.hw_out;
    it.y = rem / rm.w_out; it.x = rem - it.y * rm.w_out;
  }
  return it;
}
__device__ __forceinline__ void row_advance(RowIt& it, const RowStep& s) {
  it.x += s.dx;
  int c = it.x >= s.w;
  it.x -= c ? s.w : 0;
  it.y += s.dy + c;
  int c2 = it.y >= s.h;
  it.y -= c2 ? s.h : 0;
  it.b += s.db + c2;
}
// physical row for the thread's column segment (tap dy,dx for the 3x3 map; seg_rows for the patch map); false = zero.
// MODE is a template parameter so that the per-tile loader has no mode branches (plain rows are mode 0 with b = 0).
template <int MODE>
__device__ __forceinline__ bool row_phys(const RowMap& rm, const RowIt& it, int tap_dy, int tap_dx, int seg_rows, int& phys) {
  if constexpr (MODE == 0) {
    phys = it.b * rm.batch_stride + rm.offset + it.x;
    return true;
  } else if constexpr (MODE == 1) {
    phys = it.b * rm.tokens_in + (it.y * rm.r) * rm.w_in + it.x * rm.r + seg_rows;
    return true;
  } else {
    int y = it.y + tap_dy, x = it.x + tap_dx;
    phys = it.b * rm.tokens_in + y * rm.w_in + x;
    return (unsigned)y < (unsigned)rm.h_in && (unsigned)x < (unsigned)rm.w_in;
  }
}

// geometry of one [64 m][W n] bf16 tile (W = 128 or 64): 16-B slots per row, rows per 256-thread pass, bank hash
template <int W> struct DmaTile {
  static constexpr int CH = W / 8, RPP = NTHREADS / CH, IT = TBK / RPP, ROWB = W * 2, BYTES = TBK * ROWB;
  // 256-B rows (W=128): rows m..m+3, m+8..m+11 of a read group all start on bank 0 -> spread by (m&3, bit 3);
  // 128-B rows (W=64): bit 0 of the row already picks the bank half -> spread by (bit 1, bit 3)
  static __device__ __forceinline__ int h(int row) {
    return W == 128 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
  }
  static __device__ __forceinline__ int frag_off(int frow, int window, int L) {
    return frow * ROWB + ((window ^ h(frow)) << 5) + ((L & 3) << 3);
  }
};

// DG (plain rows, one BMT x BN = C x C output tile): the same pass over A = dY also produces the Linear's INPUT gradient dX = dY W -- the 64-row
// A tile is read a second time from LDS, row-wise, as the A operand of a [64 x C] x [C x C] product against W^T fragments that stay in
// registers (wave w owns output columns [w C/4, (w + 1) C/4)); the result leaves through an LDS tile as whole 16-byte row pieces one
// iteration later (so that its stores are a tile time old when the next counted vmcnt wait sees them).
template <int BMT, int BN, int BMODE, int NS, bool DG = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_dma_kernel(mvlt_gemm_tn_args p, int m_per_split, int t1, int t2, int splits, bf16* part = nullptr) {
  using TA = DmaTile<BMT>;
  using TB = DmaTile<BN>;
  static_assert(!DG || (BMODE == 3 && BMT == BN && !MVLT_TN_EARLY), "dgrad rides on the plain-row single-tile kernels");
  constexpr int WM = BMT / 2, TM_ = WM / 16;             // wave tile rows: 64 (4 fragments) or 32 (2)
  constexpr int WN = BN / 2, TN_ = WN / 16;
  constexpr int STAGE = TA::BYTES + TB::BYTES;
  constexpr int LPT = TA::IT + TB::IT;                    // DMA instructions per thread per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [NS][A tile | B tile]

  // XCD-aware order: workgroup b runs on XCD b % 8.  All t1*t2 output tiles of one m-split read the same rows of A
  // and B, so a whole split is given to ONE XCD (its tiles are consecutive in that XCD's queue and co-resident) and
  // the operands come through that XCD's L2 once instead of once per tile.
  // With fewer than 8 splits (M small, many output tiles: the vocabulary decoder) the natural order is kept instead:
  // x fastest, so the tiles sharing an A column slab (same x) meet on XCD x % 8.
  const int txy = t1 * t2;
  int bz, xy;
  if (splits >= 8) {
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int zq = kq / txy;
    xy = kq - zq * txy;
    bz = zq * 8 + xcd;
    if (bz >= splits) return;
  } else {
    bz = blockIdx.x / txy;
    xy = blockIdx.x - bz * txy;
  }
  const int bx = xy % t1, by = xy / t1;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n1_0 = bx * BMT, n2_0 = by * BN;
  const unsigned smem_lds = (unsigned)(uintptr_t)smem;      // LDS byte address of the dynamic segment
  const int m_begin = bz * m_per_split;
  const int m_end = min(p.M, m_begin + m_per_split);
  const RowMap amap = to_rowmap(p.a_map), bmap = to_rowmap(p.b_map);
  // bias gradients: every tile row (column) of workgroups sees the same A (B) rows; the t2 (t1) workgroups that share
  // them take turns, one 64-row tile each, so the extra MFMAs are spread evenly over the launch
  const bool do_colsum = p.colsum_a != nullptr;
  const bool do_colsum_b = p.colsum_b != nullptr;

  // ---- loader geometry (fixed per thread): LDS slot -> source column chunk
  const int a_row0 = tid / TA::CH;                                        // + RPP * j
  const int a_col = n1_0 + ((((tid % TA::CH)) ^ (TA::h(a_row0) << 1)) << 3);
  const bool a_col_ok = a_col < p.N1;
  const int b_row0 = tid / TB::CH;
  const int b_colg = n2_0 + ((((tid % TB::CH)) ^ (TB::h(b_row0) << 1)) << 3);
  const bool b_col_ok = b_colg < p.N2;
  int b_col = b_colg, b_seg_rows = 0, tap_dy = 0, tap_dx = 0;
  if constexpr (BMODE == 1 || BMODE == 2) {
    const int b_seg = b_colg / bmap.c_seg;
    b_col = b_colg - b_seg * bmap.c_seg;
    if constexpr (BMODE == 1) b_seg_rows = rowmap_seg(bmap, b_seg);
    else { tap_dy = b_seg / 3 - 1; tap_dx = b_seg - (b_seg / 3) * 3 - 1; }
  }
  const char* zsrc = (const char*)g_zero_page + ((tid * 16 + (blockIdx.x & 15) * 4096) & 65535);
  const char* a_src = (const char*)p.A + 2 * a_col;            // + physical row * row bytes
  const char* b_src = (const char*)p.B + 2 * b_col;
  const unsigned a_rowb = 2u * (unsigned)p.lda, b_rowb = 2u * (unsigned)p.ldb;
  const RowStep astep = row_step(amap, TBK), bstep = row_step(bmap, TBK);
  RowIt ait[TA::IT], bit[TB::IT];
#pragma unroll
  for (int j = 0; j < TA::IT; ++j) ait[j] = row_init(amap, m_begin + j * TA::RPP + a_row0);
#pragma unroll
  for (int j = 0; j < TB::IT; ++j) bit[j] = row_init(bmap, m_begin + j * TB::RPP + b_row0);

  // BMODE 3 = both operands with identity rows (most weight gradients): the source of a slot is a pointer that advances
  // by 64 rows per tile, and only the last (partial) tile of a split looks at row indices at all
  constexpr bool PLAIN = (BMODE == 3);
  const char* a_ptr[TA::IT];
  const char* b_ptr[TB::IT];
  const long a_adv = a_col_ok ? (long)TBK * a_rowb : 0, b_adv = b_col_ok ? (long)TBK * b_rowb : 0;
  if constexpr (PLAIN) {
#pragma unroll
    for (int j = 0; j < TA::IT; ++j) a_ptr[j] = a_col_ok ? a_src + (unsigned long long)(unsigned)(m_begin + j * TA::RPP + a_row0) * a_rowb : zsrc;
#pragma unroll
    for (int j = 0; j < TB::IT; ++j) b_ptr[j] = b_col_ok ? b_src + (unsigned long long)(unsigned)(m_begin + j * TB::RPP + b_row0) * b_rowb : zsrc;
  }
  // tiles are issued in order m_begin, m_begin + 64, ... (the row iterators advance by one tile per call)
  auto issue = [&](int mt, int slot) {
    if constexpr (PLAIN) {
      const bool whole = mt + TBK <= m_end;               // uniform
#pragma unroll
      for (int j = 0; j < TA::IT; ++j) {
        const int woff = slot * STAGE + (j * NTHREADS + wave * 64) * 16;
        const char* src = (whole || mt + j * TA::RPP + a_row0 < m_end) ? a_ptr[j] : zsrc;
        glds16(src, __builtin_amdgcn_readfirstlane(smem_lds + woff));
        a_ptr[j] += a_adv;
      }
#pragma unroll
      for (int j = 0; j < TB::IT; ++j) {
        const int woff = slot * STAGE + TA::BYTES + (j * NTHREADS + wave * 64) * 16;
        const char* src = (whole || mt + j * TB::RPP + b_row0 < m_end) ? b_ptr[j] : zsrc;
        glds16(src, __builtin_amdgcn_readfirstlane(smem_lds + woff));
        b_ptr[j] += b_adv;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TA::IT; ++j) {
      int m = mt + j * TA::RPP + a_row0;
      int phys;
      bool ok = row_phys<0>(amap, ait[j], 0, 0, 0, phys) && m < m_end && a_col_ok;
      row_advance(ait[j], astep);
      const int woff = slot * STAGE + (j * NTHREADS + wave * 64) * 16;
      glds16(ok ? a_src + (unsigned long long)(unsigned)phys * a_rowb : zsrc, __builtin_amdgcn_readfirstlane(smem_lds + woff));
    }
#pragma unroll
    for (int j = 0; j < TB::IT; ++j) {
      int m = mt + j * TB::RPP + b_row0;
      int phys;
      bool ok = row_phys<(BMODE == 3 ? 0 : BMODE)>(bmap, bit[j], tap_dy, tap_dx, b_seg_rows, phys) && m < m_end && b_col_ok;
      row_advance(bit[j], bstep);
      const int woff = slot * STAGE + TA::BYTES + (j * NTHREADS + wave * 64) * 16;
      glds16(ok ? b_src + (unsigned long long)(unsigned)phys * b_rowb : zsrc, __builtin_amdgcn_readfirstlane(smem_lds + woff));
    }
  };

  // ---- fragment geometry: lane (g = lane>>4, L = lane&15) supplies tile row 8g + (L>>2), 8-B piece (L&3) of the
  //      16-column window of the n tile; second read 4 rows below; ks adds 32 rows
  const int g = lane >> 4, L = lane & 15;
  const int frow = 8 * g + (L >> 2);
  int aoff[TM_], boff[TN_];
#pragma unroll
  for (int i = 0; i < TM_; ++i) aoff[i] = TA::frag_off(frow, wm * TM_ + i, L);
#pragma unroll
  for (int j = 0; j < TN_; ++j) boff[j] = TB::frag_off(frow, wn * TN_ + j, L);

  f32x4 acc[TM_][TN_];
#pragma unroll
  for (int i = 0; i < TM_; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  bf16* const nb = nullptr;
  // ---- DG: W^T fragments of this wave's output columns (B operand: row n = input feature, 8 consecutive k = output features)
  constexpr int DGN = DG ? BN / 64 : 1, DGK = BMT / 32;
  u32x4 wtf[DGN][DGK];
  bf16* const sdX = (bf16*)(smem + NS * STAGE);              // [64][BN] parked dgrad tile
  if constexpr (DG) {
#pragma unroll
    for (int jn = 0; jn < DGN; ++jn)
#pragma unroll
      for (int ks = 0; ks < DGK; ++ks)
        wtf[jn][ks] = *(const u32x4*)((const bf16*)p.dgrad_wt + (long)((wave * DGN + jn) * 16 + (lane & 15)) * BMT + ks * 32 + (lane >> 4) * 8);
  }
  auto dg_store = [&](int mt_prev) {                         // the parked tile of rows mt_prev .. mt_prev + 63 -> dgrad_out, 16 bytes per thread and pass
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int u = tid; u < 64 * CPR; u += NTHREADS) {
      const int r = u / CPR, c = u - r * CPR;
      if (mt_prev + r < m_end) *(u32x4*)((bf16*)p.dgrad_out + (long)(mt_prev + r) * p.dgrad_ld + c * 8) = *(const u32x4*)(sdX + r * BN + c * 8);
    }
  };

  // NS-deep ring: tile t+NS-1 is issued while tile t is consumed; the wait leaves the NS-2 younger tiles in flight.
  // Near the end fewer tiles are in flight than the count assumes, so the wait falls back to vmcnt(0) there; nothing
  // is issued past the last tile, so no DMA is outstanding when the epilogue reuses the LDS.
  // MVLT_TN_EARLY (early slot release, as in gemm_nt_dma_kernel): all fragments of a tile are read up front, a second barrier frees its slot and
  // the refill (tile t + NS) is issued in front of the MFMAs: NS tiles in flight per workgroup instead of NS - 1
  constexpr bool EARLY = MVLT_TN_EARLY != 0;
#pragma unroll
  for (int st = 0; st < (EARLY ? NS : NS - 1); ++st)
    if (m_begin + st * TBK < m_end) issue(m_begin + st * TBK, st);
  int slot = 0, islot = EARLY ? 0 : NS - 1;
  for (int mt = m_begin; mt < m_end; mt += TBK) {
    constexpr int YOUNG = EARLY ? NS - 1 : NS - 2;       // younger tiles that may still be in flight when tile `mt` is needed
    if (YOUNG > 0 && mt + YOUNG * TBK < m_end) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPT * (YOUNG > 0 ? YOUNG : 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();    // tile `mt` has landed for every wave; everyone is done reading the slot refilled next
    asm volatile("" ::: "memory");
    if constexpr (DG) { if (mt > m_begin) dg_store(mt - TBK); }      // (parked by every wave before this barrier)
    if (!EARLY) {
      if (mt + (NS - 1) * TBK < m_end) issue(mt + (NS - 1) * TBK, islot);
      islot = islot + 1 == NS ? 0 : islot + 1;
    }
    const char* sA = smem + slot * STAGE;
    const char* sB = sA + TA::BYTES;
    slot = slot + 1 == NS ? 0 : slot + 1;
    const int cs_turn = ((mt - m_begin) / TBK) % (do_colsum ? t2 : t1);
    u32x4 fa_all[TBK / 32][TM_], fb_all[TBK / 32][TN_];
    if (EARLY) {
#pragma unroll
      for (int ks = 0; ks < TBK / 32; ++ks) {
#pragma unroll
        for (int i = 0; i < TM_; ++i) fa_all[ks][i] = tr_frag(sA + aoff[i] + ks * 32 * TA::ROWB, TA::ROWB);
#pragma unroll
        for (int j = 0; j < TN_; ++j) fb_all[ks][j] = tr_frag(sB + boff[j] + ks * 32 * TB::ROWB, TB::ROWB);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // every wave holds its fragments: the slot may be refilled
      asm volatile("" ::: "memory");
      if (mt + NS * TBK < m_end) issue(mt + NS * TBK, islot);
      islot = islot + 1 == NS ? 0 : islot + 1;
    }
#pragma unroll
    for (int ks = 0; ks < TBK / 32; ++ks) {
      u32x4 fa[TM_], fb[TN_];
#pragma unroll
      for (int i = 0; i < TM_; ++i) fa[i] = EARLY ? fa_all[ks][i] : tr_frag(sA + aoff[i] + ks * 32 * TA::ROWB, TA::ROWB);
#pragma unroll
      for (int j = 0; j < TN_; ++j) fb[j] = EARLY ? fb_all[ks][j] : tr_frag(sB + boff[j] + ks * 32 * TB::ROWB, TB::ROWB);
#pragma unroll
      for (int i = 0; i < TM_; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j) mma16(acc[i][j], fa[i], fa[i], fb[j], fb[j], nb);
      // column sums: the TM_ (TN_) fragments of a tile row (column) are shared by two waves; each takes half of them
      if (do_colsum && cs_turn == by) {
#pragma unroll
        for (int t = 0; t < TM_ / 2; ++t) {
          if (wn == 0) mma16(cs[t], fa[t], fa[t], ones, ones, nb);
          else mma16(cs[t], fa[TM_ / 2 + t], fa[TM_ / 2 + t], ones, ones, nb);
        }
      }
      if (do_colsum_b && cs_turn == bx) {
#pragma unroll
        for (int t = 0; t < TN_ / 2; ++t) {
          if (wm == 0) mma16(cs[t], ones, ones, fb[t], fb[t], nb);
          else mma16(cs[t], ones, ones, fb[TN_ / 2 + t], fb[TN_ / 2 + t], nb);
        }
      }
    }
    if constexpr (DG) {
      // dX tile [64 rows][BN]: row-wise A fragments of the dY tile (slot c of row r holds source chunk c ^ (h(r) << 1): an involution)
      const int dfr = lane & 15, dfg = lane >> 4;
      f32x4 dacc[4][DGN];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < DGN; ++jn) dacc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DGK; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 16 + dfr;
          const u32x4 af = *(const u32x4*)(sA + row * TA::ROWB + (((ks * 4 + dfg) ^ (TA::h(row) << 1)) << 4));
#pragma unroll
          for (int jn = 0; jn < DGN; ++jn) mma16(dacc[i][jn], af, af, wtf[jn][ks], wtf[jn][ks], nb);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // every thread has read the previous parked tile (its store-out sits in front of this iteration's MFMAs)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < DGN; ++jn)
#pragma unroll
          for (int r = 0; r < 4; ++r) sdX[(i * 16 + 4 * dfg + r) * BN + (wave * DGN + jn) * 16 + dfr] = (bf16)dacc[i][jn][r];
    }
  }
  if constexpr (DG) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (m_end > m_begin) dg_store(m_begin + ((m_end - m_begin - 1) / TBK) * TBK);
    __syncthreads();                     // (the column sums below reuse the front of the LDS)
  }

  const int fr = lane & 15, fg = lane >> 4;
  // column sums leave through LDS so that a workgroup sends one 64-lane atomic per 64 columns: every workgroup of the
  // launch adds into the same few cache lines, and those requests serialise at the memory side
  if (do_colsum || do_colsum_b) {
    float* s_cs = (float*)smem;
    __syncthreads();                 // all tile reads are done (uniform branch: by / bx are per workgroup)
    if (do_colsum && fr == 0) {      // cs[t][r]: tile row (wn*TM_/2+t)*16 + 4*fg + r, identical in every lane column
#pragma unroll
      for (int t = 0; t < TM_ / 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_cs[wm * WM + (wn * (TM_ / 2) + t) * 16 + 4 * fg + r] = cs[t][r];
    }
    if (do_colsum_b && fg == 0) {    // cs[t][0]: tile column (wm*TN_/2+t)*16 + fr
#pragma unroll
      for (int t = 0; t < TN_ / 2; ++t) s_cs[wn * WN + (wm * (TN_ / 2) + t) * 16 + fr] = cs[t][0];
    }
    __syncthreads();
    if (do_colsum && tid < BMT && n1_0 + tid < p.N1) atomicAdd(&p.colsum_a[n1_0 + tid], s_cs[tid]);
    if (do_colsum_b && tid < BN && n2_0 + tid < p.N2) atomicAdd(&p.colsum_b[n2_0 + tid], s_cs[tid]);
  }
  if (part) {
    // PARTIAL-TILE mode (round 5; the host chooses it for outputs that many m-splits meet on: the q / proj weight gradients of stages 3-4 spent 14-21 us of 53-56 us in
    // 5.7-8.4 M fp32 atomics): this split's tile goes to part[split][N1][N2] in bf16 -- 16 rows at a time through a per-wave LDS tile, out as 16-byte pieces of contiguous
    // rows -- and tn_fold_kernel adds the splits in order (deterministic).  N2 % 8 == 0 (host), rows past N1 and 8-column pieces past N2 are not stored.
    constexpr int LDP = WN + 8;
    __syncthreads();                                    // every wave is done with the operand tiles (and the column sums) this overlays
    bf16* const st = (bf16*)smem + wave * 16 * LDP;
    bf16* const P = part + (size_t)bz * p.N1 * p.N2;
#pragma unroll
    for (int i = 0; i < TM_; ++i) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < TN_; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(4 * fg + r) * LDP + j * 16 + fr] = (bf16)acc[i][j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = lane; q < 16 * (WN / 8); q += 64) {
        const int row = q / (WN / 8), ch = q - row * (WN / 8);
        const int n1 = n1_0 + wm * WM + i * 16 + row, n2 = n2_0 + wn * WN + ch * 8;
        if (n1 < p.N1 && n2 < p.N2) st_g<MVLT_NT_GEMM>((u32x4*)(P + (size_t)n1 * p.N2 + n2), *(const u32x4*)(st + row * LDP + ch * 8));
      }
    }
    return;
  }
  if (p.c_overwrite) {
    // round 6: C = (not +=) this launch's single m-split (the caller vouches for a zeroed C: the vocabulary decoder's weight gradient, 30522 x 768 = 23 M outputs whose
    // fire-and-forget fp32 atomics were most of the launch).  16 rows of the wave tile at a time through a per-wave LDS tile, out as 16-byte pieces of contiguous rows.
    constexpr int LDP = WN + 4;
    constexpr int CPR = WN / 4;
    __syncthreads();                                   // every wave is done with the operand tiles this overlays
    float* const st = (float*)smem + wave * 16 * LDP;
#pragma unroll
    for (int i = 0; i < TM_; ++i) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < TN_; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(4 * fg + r) * LDP + j * 16 + fr] = acc[i][j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = lane; q < 16 * CPR; q += 64) {
        const int row = q / CPR, ch = q - row * CPR;
        const int n1 = n1_0 + wm * WM + i * 16 + row, n2 = n2_0 + wn * WN + ch * 4;
        if (n1 < p.N1 && n2 < p.N2) st_g<MVLT_NT_GEMM>((f32x4*)(p.C + (size_t)n1 * p.ldc + n2), *(const f32x4*)(st + row * LDP + ch * 4));
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM_; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int n1 = n1_0 + wm * WM + i * 16 + 4 * fg + r;
        int n2 = n2_0 + wn * WN + j * 16 + fr;
        if (n1 < p.N1 && n2 < p.N2) atomicAdd(p.trans_c ? &p.C[(long)n2 * p.ldc + n1] : &p.C[(long)n1 * p.ldc + tn_dst_col(p, n2)], acc[i][j][r]);
      }
}

// ------------------------------------------------------------------------------------------------ conv3x3 weight gradient, LDS halo
// dW[o][tap][c] += sum over pixels dz[pixel][o] * x[pixel + tap][c]: the weight gradient of the MIM decoder's conv3x3 (reference
// libs/vl_heads.py:116-129) on pixel-major data.  As the generic TN GEMM with the 3x3 gather on B (BMODE 2 above) every tap re-fetches
// its shifted copy of the input rows through the LDS-DMA path -- 9 x 64 rows of B per 64-pixel k-tile -- and that path, not the MFMAs,
// bounds the GEMM loops of this chip (round-2 ablations): 96 B/clk/CU asked of a path that sustains ~16-20.
// Here a workgroup owns a 64 (out) x 9 (taps) x 64 (in) block of dW in registers (36 accumulator tiles per wave) and, per k-tile of
// 64 pixels (= 64 / W whole image rows), loads the pixels' HALO once -- (64/W + 2) x (W + 2) input rows of 64 channels, 17 KB at
// W = 32 instead of 72 KB -- and forms all nine taps' B fragments from it by transposed LDS reads at shifted row addresses (pixels of
// a fragment are consecutive in one image row, so a tap is a constant row offset; out-of-image halo rows come from the zero page).
// 25 KB through the DMA path per 4.7 MFLOP: ~22 B/clk/CU at full MFMA rate.
template <int W>
__global__ __launch_bounds__(NTHREADS, 2) void conv3_wgrad_kernel(mvlt_gemm_tn_args p, int tiles_per_split, int n_o, int n_c, int splits, bf16* part) {
  using TA = DmaTile<64>;
  constexpr int R = 64 / W;                         // image rows per 64-pixel k-tile
  constexpr int HW2 = W + 2, HR = (R + 2) * HW2;    // halo rows (one pixel each, 64 channels = 128 B)
  constexpr int H_IT = (HR * 8 + NTHREADS - 1) / NTHREADS;          // DMA instructions per thread for the halo (last one partly idle)
  constexpr int HALO_BYTES = H_IT * NTHREADS * 16;
  constexpr int STAGE = TA::BYTES + HALO_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][dz tile | halo tile]
  const int txy = n_o * n_c;
  int bz, xy;
  if (splits >= 8) {
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int zq = kq / txy;
    xy = kq - zq * txy;
    bz = zq * 8 + xcd;
    if (bz >= splits) return;
  } else {
    bz = blockIdx.x / txy;
    xy = blockIdx.x - bz * txy;
  }
  const int bo = xy % n_o, bc = xy / n_o;
  const int o0 = bo * 64, c0 = bc * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned smem_lds = (unsigned)(uintptr_t)smem;
  const int ntiles = p.M / 64;
  const int t_begin = bz * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);
  const int Himg = p.b_map.h_in, cin = p.b_map.c_seg, tokens_in = p.b_map.tokens_in;
  const int tiles_per_img = Himg * W / 64;
  const char* zsrc = (const char*)g_zero_page + ((tid * 16 + (blockIdx.x & 15) * 4096) & 65535);

  // ---- loader geometry.  dz tile: as the TN kernel's A tile (64 pixels x 64 outputs, chunks swizzled by TA::h on the source side)
  const int a_row0 = tid / TA::CH;
  const int a_col = o0 + (((tid % TA::CH) ^ (TA::h(a_row0) << 1)) << 3);
  const char* a_src = (const char*)p.A + 2 * a_col;
  const unsigned a_rowb = 2u * (unsigned)p.lda, b_rowb = 2u * (unsigned)p.ldb;
  // halo tile: LDS row hr = (hy, hx) of the padded (R+2) x (W+2) window; slot s of row hr holds source chunk s ^ (hh(hr) << 1).
  // The bank hash of this tile uses bit 1 of the row only: a fragment's second read sits 4 rows below its first at ANY row
  // alignment here (taps shift the rows by +-1 and +-(W+2)), so the hash must not change under +4 (TA::h does when bit 2 is set).
  // Rows r..r+3 land on four distinct (bank half, window) pairs, the +8 group repeats them: 2-way conflicts, LDS stays off the
  // critical path (72 transposed reads per 36 MFMAs).
  auto hh = [](int row) { return (row >> 1) & 1; };
  int h_dy[H_IT], h_off[H_IT];
  bool h_xok[H_IT];
#pragma unroll
  for (int j = 0; j < H_IT; ++j) {
    const int q = tid + j * NTHREADS, hr = q >> 3, sl = q & 7;
    const int hy = hr / HW2, hx = hr - hy * HW2;
    h_dy[j] = hy - 1;
    h_xok[j] = hr < HR && (unsigned)(hx - 1) < (unsigned)W;
    h_off[j] = (hx - 1) * (int)b_rowb + 2 * (c0 + ((sl ^ (hh(hr) << 1)) << 3));
  }
  auto issue = [&](int tile, int slot) {
    const int img = tile / tiles_per_img, y0 = (tile - img * tiles_per_img) * R;
#pragma unroll
    for (int j = 0; j < TA::IT; ++j) {
      const unsigned m = (unsigned)(tile * 64 + j * TA::RPP + a_row0);
      glds16(a_src + (unsigned long long)m * a_rowb, __builtin_amdgcn_readfirstlane(smem_lds + slot * STAGE + (j * NTHREADS + wave * 64) * 16));
    }
    const char* img_base = (const char*)p.B + (unsigned long long)(unsigned)(img * tokens_in) * b_rowb;
#pragma unroll
    for (int j = 0; j < H_IT; ++j) {
      const int y = y0 + h_dy[j];
      const bool ok = h_xok[j] && (unsigned)y < (unsigned)Himg;
      glds16(ok ? img_base + (long)(y * W) * (long)b_rowb + h_off[j] : zsrc,
             __builtin_amdgcn_readfirstlane(smem_lds + slot * STAGE + TA::BYTES + (j * NTHREADS + wave * 64) * 16));
    }
  };

  // ---- fragment geometry (transposed reads: lane (g, L) supplies k-row 8g + (L >> 2) and the row 4 below, 8-B piece L & 3)
  const int g = lane >> 4, L = lane & 15;
  const int frow = 8 * g + (L >> 2);
  int aoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aoff[i] = TA::frag_off(frow, wm * 2 + i, L);
  // B fragment of (k32 step ks, tap t, channel tile j): halo row of pixel ks*32 + frow shifted by the tap
  int boff[2][9];                                   // channel tile j = 0; j = 1 is the neighbouring 32-B window: offset ^ 32
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int pix = ks * 32 + frow, py = pix / W, px = pix - py * W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hr = (py + t / 3) * HW2 + px + t % 3;             // (py + 1 + dy, px + 1 + dx) with dy = t/3 - 1, dx = t%3 - 1
      boff[ks][t] = TA::BYTES + hr * 128 + (((wn * 2) ^ hh(hr)) << 5) + ((L & 3) << 3);
    }
  }
  f32x4 acc[2][9][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16* const nb = nullptr;

  if (t_begin < t_end) issue(t_begin, 0);
  int slot = 0;
  for (int tile = t_begin; tile < t_end; ++tile, slot ^= 1) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // tile has landed for every wave; everyone is done reading the slot refilled next
    asm volatile("" ::: "memory");
    if (tile + 1 < t_end) issue(tile + 1, slot ^ 1);
    const char* sS = smem + slot * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = tr_frag(sS + aoff[i] + ks * 32 * TA::ROWB, TA::ROWB);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        u32x4 fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = tr_frag(sS + (boff[ks][t] ^ (j << 5)), 128);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mma16(acc[i][t][j], fa[i], fa[i], fb[j], fb[j], nb);
      }
    }
  }
  const int fr = lane & 15, fg = lane >> 4;
  if (part) {
    // PARTIAL-TILE mode (round 5): the flush of the 64 x 9 x 64 blocks was 35 % of this kernel family (18.6 M fp32 atomics at 192 -> 192 / 32 x 32: 193 us with, 155 us without
    // them; 64 -> 64: 75 / 32 us -- profiles/r05_conv_wgrad_atomics_ablation.txt).  The split's block goes to part[split][N1][N2] in bf16 -- 16 output rows at a time through a
    // per-wave LDS tile [16][9 taps x 32 columns], out as 16-byte pieces (64 contiguous bytes per row and tap) -- and tn_fold_kernel adds the splits in order.
    constexpr int LDP = 9 * 32 + 8;
    __syncthreads();                                    // every wave is done with the dz tiles / halos this overlays
    bf16* const st = (bf16*)smem + wave * 16 * LDP;
    bf16* const P = part + (size_t)bz * p.N1 * p.N2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) st[(4 * fg + r) * LDP + t * 32 + j * 16 + fr] = (bf16)acc[i][t][j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int q = lane + 64 * k, row = q / 36, rem = q - row * 36, t = rem >> 2, pc = rem & 3;
        const int n1 = o0 + wm * 32 + i * 16 + row, n2 = t * cin + c0 + wn * 32 + pc * 8;
        st_g<MVLT_NT_GEMM>((u32x4*)(P + (size_t)n1 * p.N2 + n2), *(const u32x4*)(st + row * LDP + t * 32 + pc * 8));
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n1 = o0 + wm * 32 + i * 16 + 4 * fg + r;
          const int n2 = t * cin + c0 + wn * 32 + j * 16 + fr;
          atomicAdd(&p.C[(long)n1 * p.ldc + n2], acc[i][t][j][r]);
        }
}

__global__ void tn_fold_kernel(const bf16* __restrict__ part, int splits, int N1, int N2, float* __restrict__ C, int ldc);      // below, with the 8-phase TN kernel

// ---- deferred folds (mvlt_gemm_tn_args.defer_fold): a fold is a ~8 us memory-bound launch between two long MFMA-bound ones (30 per step, 2.5 TB/s).  A deferring caller's
// partial tiles stay in its scratch -- each launch takes the next free region -- and up to FOLD_MAX of them are folded by ONE launch of tn_fold_multi_kernel: when the table or
// the scratch is full, when a non-deferring launch wants the scratch, or when the caller says mvlt_tn_fold_flush (before anything reads the gradients).  Same stream throughout.
constexpr int FOLD_MAX = 32;
struct FoldDesc { const bf16* part; float* C; int splits, N1, N2, ldc, wg0; };
struct FoldBatch { FoldDesc d[FOLD_MAX]; int n, total_wgs; };
__global__ void tn_fold_multi_kernel(FoldBatch b);
// One pending table PER SCRATCH (= per owner: a FlatStore holds one scratch; ADVICE r5 -- a single process-global table let the start of one model's backward pass discard another
// model's pending folds, and a launch on another stream reused the scratch while the old stream's fold still read it).  The map and its tables are guarded by g_fold_mu; the launches
// themselves are asynchronous, so the lock is held for table bookkeeping + one enqueue.
struct FoldPending {
  FoldBatch b = {};
  long used = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;                         // recorded behind a fold when ANOTHER stream is about to reuse the scratch / read the gradients
};
std::mutex g_fold_mu;
std::map<const void*, FoldPending> g_fold;         // key: mvlt_gemm_tn_args.partials
int fold_wgs(const FoldDesc& d) { return (int)(((long)d.N1 * d.N2 / 8 + 31) / 32); }
// fold what is pending in `f` on the stream its producers ran on; `reader` (if it is another stream) is made to wait for that fold
void fold_flush_locked(FoldPending& f, hipStream_t reader, bool order_reader) {
  if (f.b.n > 0) {
    f.b.total_wgs = f.b.d[f.b.n - 1].wg0 + fold_wgs(f.b.d[f.b.n - 1]);
    MVLT_LAUNCH(tn_fold_multi_kernel, dim3((unsigned)f.b.total_wgs), dim3(256), 0, f.stream, f.b);
    if (order_reader && reader != f.stream) {
      if (!f.ev) (void)hipEventCreateWithFlags(&f.ev, hipEventDisableTiming);
      if (f.ev && hipEventRecord(f.ev, f.stream) == hipSuccess) (void)hipStreamWaitEvent(reader, f.ev, 0);
    }
  }
  f.b.n = 0;
  f.used = 0;
}
// the scratch region of this launch's partial tiles, or nullptr when they do not fit
bf16* fold_acquire(const mvlt_gemm_tn_args& a, long need, hipStream_t s, int room = 1) {
  if (!a.partials || ((uintptr_t)a.partials & 15) || need > a.partials_bytes) return nullptr;
  std::lock_guard<std::mutex> lk(g_fold_mu);
  FoldPending& f = g_fold[a.partials];
  bool flush = f.b.n > 0 && (!a.defer_fold || f.stream != s || f.b.n + room > FOLD_MAX || f.used + need > a.partials_bytes);
  // two pending folds into the same gradient would be two unordered read-modify-writes in one launch (the kv weight gradient takes its text rows and its image rows from two
  // GEMMs): an output that overlaps a pending one folds the pending ones first
  const float* c_lo = a.C;
  const float* c_hi = a.C + (size_t)(a.N1 - 1) * a.ldc + a.N2;
  for (int i = 0; i < f.b.n && !flush; ++i) {
    const FoldDesc& d = f.b.d[i];
    const float* d_lo = d.C;
    const float* d_hi = d.C + (size_t)(d.N1 - 1) * d.ldc + d.N2;
    if (c_lo < d_hi && d_lo < c_hi) flush = true;
  }
  if (flush) fold_flush_locked(f, s, true);        // a launch on another stream writes region 0 only behind the old stream's fold
  f.stream = s;
  if (!a.defer_fold) return (bf16*)a.partials;
  return (bf16*)((char*)a.partials + f.used);
}
void fold_launch(const mvlt_gemm_tn_args& a, const bf16* part, int splits, hipStream_t s) {
  const long groups = (long)a.N1 * a.N2 / 8;
  if (!a.defer_fold) {
    MVLT_LAUNCH(tn_fold_kernel, dim3((unsigned)((groups + 31) / 32)), dim3(256), 0, s, part, splits, a.N1, a.N2, a.C, a.ldc);
    return;
  }
  std::lock_guard<std::mutex> lk(g_fold_mu);
  FoldPending& f = g_fold[a.partials];
  FoldDesc& d = f.b.d[f.b.n];
  d.part = part; d.C = a.C; d.splits = splits; d.N1 = a.N1; d.N2 = a.N2; d.ldc = a.ldc;
  d.wg0 = f.b.n == 0 ? 0 : f.b.d[f.b.n - 1].wg0 + fold_wgs(f.b.d[f.b.n - 1]);
  ++f.b.n;
  f.used += ((long)splits * a.N1 * a.N2 * 2 + 255) & ~255L;
}

template <int W> int launch_conv3_wgrad(const mvlt_gemm_tn_args& a, hipStream_t s) {
  constexpr int R = 64 / W, HR = (R + 2) * (W + 2), H_IT = (HR * 8 + NTHREADS - 1) / NTHREADS;
  const size_t lds = (size_t)2 * (DmaTile<64>::BYTES + H_IT * NTHREADS * 16);
  const int n_o = a.N1 / 64, n_c = a.b_map.c_seg / 64, ntiles = a.M / 64;
  // every split flushes its whole 64 x 576 block with fp32 atomics: at least 16 k-tiles of work per flush, about two workgroups per
  // CU when the shape allows (262144 x 64 x 576: 640 splits of 7 tiles 108 us, 256 splits of 16 tiles measured below)
  static const int min_tiles = getenv("MVLT_CONV_WGRAD_MINT") ? atoi(getenv("MVLT_CONV_WGRAD_MINT")) : 16;
  int splits = (512 + n_o * n_c - 1) / (n_o * n_c);
  if (splits > ntiles / min_tiles) splits = ntiles / min_tiles;
  if (splits < 1) splits = 1;
  if (splits >= 8) splits = (splits + 4) / 8 * 8;
  if (splits > ntiles) splits = ntiles;
  const int tps = (ntiles + splits - 1) / splits;
  splits = (ntiles + tps - 1) / tps;
  dim3 grid((unsigned)((splits >= 8 ? 8 * ((splits + 7) / 8) : splits) * n_o * n_c)), block(NTHREADS);
  mvlt_max_lds<(conv3_wgrad_kernel<W>)>();
  // the caller's scratch takes the splits' blocks (bf16) and an ordered fold adds them to C: no atomics (MVLT_TN_P8=0 keeps them)
  static const bool part_ok = !(getenv("MVLT_TN_P8") && atoi(getenv("MVLT_TN_P8")) == 0);
  bf16* const part = (part_ok && a.partials && splits >= 4 && a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0 && lds >= (size_t)4 * 16 * (9 * 32 + 8) * 2)
                         ? fold_acquire(a, (long)splits * a.N1 * a.N2 * 2, s) : nullptr;
  MVLT_LAUNCH((conv3_wgrad_kernel<W>), grid, block, lds, s, a, tps, n_o, n_c, splits, part);
  if (part) fold_launch(a, part, splits, s);
  return mvlt_check_launch("mvlt_gemm_tn");
}

// ------------------------------------------------------------------------------------------------ NT, bf16, LDS-DMA
// gemm_nt_kernel<bf16,BN> with the operand tiles filled by global_load_lds_dwordx4 instead of load -> VGPR ->
// ds_write_b128 (LDS stores run at ~80 B/clk/CU, a third of the read rate, and were as expensive as the MFMAs).  The
// LDS image is the same (128-B rows of 64 k, 16-B chunks XOR-swizzled by row); because the DMA writes lane-linear the
// swizzle is applied to the source chunk each lane fetches.  ns-deep ring with a counted vmcnt wait, as in the TN kernel.
template <int BN, int AMODE, int EPI, int BK, int BMT = BM>
__global__ __launch_bounds__(NTHREADS, BMT == 256 ? 2 : 1) void gemm_nt_dma_kernel(mvlt_gemm_nt_args p, int ns_flags) {
  const int ns = ns_flags & 0xff;                   // ring depth
  const bool early = (ns_flags & 0x100) != 0;       // early slot release (below)
  constexpr int ROWB = BK * 2;                      // LDS row: BK k-values of one tile row
  constexpr int CH = BK / 8;                        // 16-B chunks per row (8 or 4)
  constexpr int RPL = NTHREADS / CH;                // tile rows one DMA instruction of the workgroup covers
  constexpr int WN = BN / 2, TN_ = WN / 16;
  // BMT = 256 (wave tile 128 x 64 with 32-wide K stages, a quarter less operand traffic per MFMA) was measured and is not
  // dispatched: -5 % on 49152x2048x512 and the 256-channel conv shape, +13 % on 98304x320x1280, worse under every R / H epilogue
  constexpr int WM = BMT / 2, TM_ = WM / 16;        // wave tile rows: 64
  constexpr int A_ITERS = BMT * CH / NTHREADS;      // 4 (BK 64) or 2 (BK 32) at 128 rows
  constexpr int B_ITERS = BN * CH / NTHREADS;
  constexpr int LPT = A_ITERS + B_ITERS;
  constexpr int STAGE = (BMT + BN) * ROWB;
  // chunk swizzle by row: 16 consecutive rows x one logical chunk must spread over all 64 banks
  // (four chunks per row, BK = 32: a ds_read_b128 lane group holds rows {r, r+12} at one chunk and {r+4, r+8} at the next for each r & 3 --
  // the mask must differ between those four row quads in a way that keeps chunk ^ mask distinct: (-(row >> 2)) & 3 does, (row >> 2) & 3, round
  // 2's choice, maps them onto two slots: 43 % LDS bank conflicts in the stage-3 fc1 + GELU launch)
  auto swzk = [](int row, int chunk) { return CH == 8 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((0 - (row >> 2)) & 3)); };
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (p.M + BMT - 1) / BMT;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, bslot = bid >> 3;
  const int tile_m = (bslot / tiles_n) * 8 + xcd, tile_n = bslot % tiles_n;
  if (tile_m >= tiles_m) return;
  const int m0 = tile_m * BMT, n0 = tile_n * BN;
  const RowMap amap = to_rowmap(p.a_map);
  const unsigned smem_lds = (unsigned)(uintptr_t)smem;
#ifdef MVLT_NT_STAGGER
  // experiment: the workgroups of the first round that share a CU start a quarter of a tile time apart, so that their store phases
  // do not coincide (MVLT_NT_STAGGER = shift that picks the co-resident index out of the block id, 3 or 8)
  if constexpr (EPI == 3 || EPI == 4) {
    if (bid < 1024) {
      const int phase = (bid >> MVLT_NT_STAGGER) & 3;
      for (int q = 0; q < phase; ++q) __builtin_amdgcn_s_sleep(127);
    }
  }
#endif

  const int row_in = tid / CH;                                  // 0..RPL-1 (+RPL i)
  const int chunk = swzk(row_in, tid % CH);                     // source chunk of this thread's LDS slot: swz is an involution
  const char* zsrc = (const char*)g_zero_page + ((tid * 16 + (bid & 15) * 4096) & 65535);

  const char* a_ptr[A_ITERS];
  bool a_ok[A_ITERS];
  int a_y[A_ITERS], a_x[A_ITERS];
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    int m = m0 + row_in + RPL * i;
    a_ok[i] = m < p.M;
    RowIt it = row_init(amap, a_ok[i] ? m : 0);
    int phys;
    if constexpr (AMODE == 0) phys = it.b * amap.batch_stride + amap.offset + it.x;
    else if constexpr (AMODE == 1) phys = it.b * amap.tokens_in + (it.y * amap.r) * amap.w_in + it.x * amap.r;
    else phys = it.b * amap.tokens_in + it.y * amap.w_in + it.x;                   // centre pixel
    a_y[i] = it.y; a_x[i] = it.x;
    a_ptr[i] = (const char*)p.A + (unsigned long long)(unsigned)phys * (2u * (unsigned)p.lda);
  }
  const char* b_ptr[B_ITERS];
  bool b_ok[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    int n = n0 + row_in + RPL * i;
    b_ok[i] = n < p.N;
    b_ptr[i] = (const char*)p.B + (unsigned long long)(unsigned)(b_ok[i] ? n : 0) * (2u * (unsigned)p.ldb);
  }

  // K position of this thread's chunk, kept as (segment, offset in segment) for the gather maps; advanced per tile
  // K split (blockIdx.y): this workgroup reduces k-tiles [kt0, kt0 + nk) and adds its partial tile atomically
  const int nk_all = (p.K + BK - 1) / BK;
  const int nsplit = p.split_k > 1 ? p.split_k : 1;
  const int kt_per = (nk_all + nsplit - 1) / nsplit;
  const int kt0 = blockIdx.y * kt_per;
  int kpos = kt0 * BK + chunk * 8, seg = 0, kk = kpos;                   // (a K split of a gathered A starts inside a later segment)
  if constexpr (AMODE != 0) { seg = kk / amap.c_seg; kk -= seg * amap.c_seg; }
  auto issue = [&](int slot) {                                  // tiles are issued in K order
    const bool k_ok = kpos < p.K;
    int off_bytes = (AMODE == 0 ? kpos : kk) * 2, tdy = 0, tdx = 0;
    if constexpr (AMODE == 1) off_bytes += rowmap_seg(amap, seg) * (2 * p.lda);
    if constexpr (AMODE == 2) {
      const int dy = seg / 3;
      tdy = dy - 1; tdx = seg - dy * 3 - 1;
      off_bytes += (tdy * amap.w_in + tdx) * (2 * p.lda);
    }
    if (MVLT_ABL == 1) { kpos += BK; return; }          // ablation: no DMA at all
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      bool ok = a_ok[i] && k_ok;
      if constexpr (AMODE == 2) ok = ok && (unsigned)(a_y[i] + tdy) < (unsigned)amap.h_in && (unsigned)(a_x[i] + tdx) < (unsigned)amap.w_in;
      if (MVLT_ABL == 4) ok = false;                      // ablation: every DMA instruction is issued, all of them read the (L2-resident) zero page
      glds16(ok ? a_ptr[i] + off_bytes : zsrc, __builtin_amdgcn_readfirstlane(smem_lds + slot * STAGE + (i * NTHREADS + wave * 64) * 16));
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i)
      glds16((b_ok[i] && k_ok && MVLT_ABL != 4) ? b_ptr[i] + kpos * 2 : zsrc,
             __builtin_amdgcn_readfirstlane(smem_lds + slot * STAGE + BMT * ROWB + (i * NTHREADS + wave * 64) * 16));
    kpos += BK;
    if constexpr (AMODE != 0) {
      kk += BK;
      while (kk >= amap.c_seg) { kk -= amap.c_seg; ++seg; }
    }
  };

  f32x4 acc[TM_][TN_];
#pragma unroll
  for (int i = 0; i < TM_; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = min(kt_per, nk_all - kt0);
  if (nk <= 0) return;
  const int fr = lane & 15, fg = lane >> 4;
  // Early slot release: every fragment of a stage is read into registers before its first MFMA, so the stage's slot is free as soon as all
  // four waves have done those reads -- a second barrier right behind them -- and the refill (stage kt + ns) is issued THERE, in front of
  // the MFMAs, instead of behind the barrier of the next k-step: ns stages in flight per workgroup instead of ns - 1.  With the
  // two-slot ring the K-loop otherwise runs at one LDS-DMA latency per k-step (Little's law with the LDS as the window).
  for (int st = 0; st < (early ? ns : ns - 1) && st < nk; ++st) issue(st);
  if (ns == 1 && !early) issue(0);
  int slot = 0, islot = early ? 0 : ns - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed; up to ns - 2 (early: ns - 1) later tiles may still be in flight (none near the tail)
    const int ahead = min(nk - 1 - kt, early ? ns - 1 : ns - 2);
    if (ahead >= 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * LPT) : "memory");
    else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * LPT) : "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPT) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!early) {
      if (ns > 1 && kt + ns - 1 < nk) issue(islot);
      islot = islot + 1 >= ns ? 0 : islot + 1;
    }
    const char* a_s = smem + slot * STAGE + (wm * WM) * ROWB;
    const char* b_s = smem + slot * STAGE + BMT * ROWB + (wn * WN) * ROWB;
    slot = slot + 1 >= ns ? 0 : slot + 1;
    // every fragment of the tile is requested before the first MFMA (both 32-k halves): the compiler otherwise recycles one
    // fragment set and waits out the LDS latency three times per half
    constexpr int KS = BK / 32;
    u32x4 fa[KS][TM_], fb[KS][TN_];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < TM_; ++i) {
        int r = i * 16 + fr;
        fa[ks][i] = MVLT_ABL == 3 ? u32x4{(unsigned)r, 1u, 2u, 3u} : *(const u32x4*)(a_s + r * ROWB + swzk(wm * WM + r, ks * 4 + fg) * 16);
      }
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        int r = j * 16 + fr;
        fb[ks][j] = MVLT_ABL == 3 ? u32x4{(unsigned)r, 5u, 6u, 7u} : *(const u32x4*)(b_s + r * ROWB + swzk(wn * WN + r, ks * 4 + fg) * 16);
      }
    }
    if (early) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // every wave holds its fragments of stage kt: the slot may be refilled
      asm volatile("" ::: "memory");
      if (kt + ns < nk) issue(islot);
      islot = islot + 1 >= ns ? 0 : islot + 1;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __builtin_amdgcn_sched_barrier(0);             // keeps every ds_read ahead of the MFMAs (one wait per tile)
      if (MVLT_ABL == 2) {                              // ablation: no MFMAs (the fragments are consumed by one XOR each)
#pragma unroll
        for (int i = 0; i < TM_; ++i) acc[i][0][0] += __builtin_bit_cast(float, fa[ks][i][0] ^ fb[ks][i % TN_][1]);
        continue;
      }
#pragma unroll
      for (int i = 0; i < TM_; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j) mma16(acc[i][j], fa[ks][i], fa[ks][i], fb[ks][j], fb[ks][j], (bf16*)nullptr);
    }
  }
  __syncthreads();                   // last tile's reads are done before the epilogue reuses the LDS
  static_assert(BMT == BM || (BN != 192 && EPI != 0), "the 256-row tile carries the lean epilogues only");
  if constexpr (BN == 192) nt_epilogue_192<EPI>(p, acc, smem, m0, n0, wave, lane);
  else if constexpr (EPI == 0) nt_epilogue<bf16, BN>(p, acc, smem, m0, n0, wave, lane);
  else nt_epilogue_lean<BN, EPI, TM_>(p, acc, smem, m0, n0, wave, lane);
}

// ------------------------------------------------------------------------------------------------ NT, bf16, 8 waves, 8-phase K-loop (256 x 256 x 64 and relatives)
// The 128 x 128 kernel above asks the CU's vector-memory path for 32 KB per 64-deep k-step and workgroup -- 60 B per clock and CU at matrix-pipe
// speed against the 64 B per clock the texture / L1 path is specified at (round 3: TA busy 54-68 %, TCP_PENDING_STALL ~50 % of the launch) --
// and runs its whole workgroup in lockstep (DMA issue, barrier, fragment reads, MFMAs).  This kernel halves the operand bytes per FLOP (256 x 256
// tile: 64 KB per k-step for 4x the MFMAs) and replaces the lockstep by the phase schedule of cdna_hip_programming.md 5 ("256^2 8-phase template"):
//   * 8 waves as 2 (M) x 4 (N); a wave owns (2 HM) x (HN0 + HN1) accumulator tiles of 16 x 16 -- 8 x 4 for the 256 x 256 tile, 6 x 4 for
//     192 x 256, 6 x 5 for 192 x 320; one workgroup per CU, two k-tile buffers of {A0, A1, B0, B1} half-tiles in LDS (128 KB at most).  Half h
//     of A holds the rows {wr * 2 HM 16 + h * HM 16 + r} of both wave rows, half h of B the columns of all four wave columns: every wave reads
//     ITS part of a half-tile, and a half-tile is free for its refill as soon as the one phase that reads it is over.
//   * a k-tile is four phases, one accumulator quadrant (HM x HNh tiles x K 64: 16 MFMAs at 256 x 256) each: (A0, B0) -> (A0, B1) -> (A1, B1) ->
//     (A1, B0); a phase is {fragment reads of the half-tile(s) it needs, ONE half-tile refill (2-3 LDS-DMA instructions per thread), barrier,
//     its MFMAs at raised priority, barrier}.  The refills run two k-tiles ahead; the only vmcnt wait is in phase 4 and leaves the three
//     youngest half-tiles in flight -- never zero inside the loop.
//   * the two wave rows run half a phase apart (wr == 1 takes one extra barrier up front, wr == 0 one at the end): while one wave of a SIMD
//     issues its MFMAs the other reads fragments and issues DMA, so the matrix pipe sees a new MFMA group as soon as the last one drains.
// Hazards (who may touch which half-tile when): a refill targets a half-tile whose last reads were retired by an lgkmcnt wait at least one barrier
// earlier in EVERY wave (B0: the counted lgkmcnt of phase 1, refilled in phase 2; A0 / B1 / A1: read in phase 1 / 2 / 3, refilled in phase
// 3 / 4 / 1); a half-tile is read one phase or more after the barrier that follows the vmcnt wait that retired its DMA in every wave (the wait
// of phase 4 covers k-tile t + 1 whole: the three half-tiles issued after its last one, A1(t + 1) in phase 1, are B0 / A0 / B1 of k-tile t + 2).
// Tile shapes: the launches of this model are 1.5 .. 10 tiles per CU, so whole rounds matter as much as the loop: 49152 x 512 is 384 tiles of
// 256 x 256 (1.5 rounds of 256 CUs) but 512 tiles of 192 x 256 (2 rounds of 3/4 the work each); 98304 x 320 is 512 tiles of 192 x 320.
// Preconditions (host dispatch): bf16, identity a_map, M % BM == 0, N % BN == 0, K % 64 == 0, lean epilogue (EPI 1-5; the 80-column wave
// tile of BN 320 carries EPI 1 / 2 only).
__device__ __forceinline__ void glds16_s(const char* sbase, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}
#define MVLT_BAR()                                  \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    __builtin_amdgcn_s_barrier();                   \
    asm volatile("" ::: "memory");                  \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// epilogue of the 80-column wave tile (BN 320 = 4 waves x 5 accumulator tiles): a 32-row half of the wave tile is 32 x 10 chunks of 8 columns =
// 5 chunks per lane.  EPI 1: C = AB^T (+bias); EPI 2: C = (AB^T + bias) * row_scale + R.  Identity or batch-strided c_map, fp32 / bf16 output.
// CHK (round 6): the last row tile is ragged (M % 192 != 0: pvlt_medium at 384 px has 45056 rows per stage-3 launch) -- requests are clamped to the last row, rows past M are not stored.
template <int EPI, int TM, bool CHK = false>
__device__ __forceinline__ void nt_epilogue_w80(const mvlt_gemm_nt_args& p, f32x4 (&acc)[TM][5], char* smem, int m0, int n0, int wave, int lane) {
  static_assert(EPI == 1 || EPI == 2, "the 80-column wave tile carries the plain and the residual epilogue");
  constexpr int WN = 80, LDW = WN + 4, NH = TM / 2, NIT = 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int ofp32 = p.out_dtype;
  const int rfp32 = ofp32 | p.r_fp32;                  // the residual may be fp32 beside a bf16 C (mvlt_gemm_nt_args.r_fp32)
  float* stage = (float*)smem + wave * 32 * LDW;
  const int rpb = p.c_map.rows_per_batch;
  const float inv_rpb = rpb > 0 ? 1.0f / (float)rpb : 0.f;
  const float inv_rps = (EPI == 2 && p.rows_per_scale > 0) ? 1.0f / (float)p.rows_per_scale : 0.f;
  int rl[NIT], cc[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int sidx = it * 64 + lane;
    rl[it] = sidx / 10;
    cc[it] = (sidx - rl[it] * 10) * 8;
  }
  // the residual rows (EPI 2) of a 32-row half are requested one half ahead, in two rotating slots: their HBM latency hides behind the
  // previous half's staging, arithmetic and stores (the fragment registers of the K-loop are dead here)
  long idx[2][NIT];
  float rs[2][NIT];
  u32x4 raw[2][NIT][2];
  auto request = [&](int half) {
    const int sl = half & 1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = CHK ? min(m0 + wm * (TM * 16) + half * 32 + rl[it], p.M - 1) : m0 + wm * (TM * 16) + half * 32 + rl[it];
      long phys = m;
      if (rpb > 0) {
        const int b = fdiv24(m, rpb, inv_rpb);
        phys = (long)b * p.c_map.batch_stride + p.c_map.offset + (m - b * rpb);
      }
      idx[sl][it] = phys * p.ldc + n0 + wn * WN + cc[it];
      rs[sl][it] = 1.0f;
      if (EPI == 2) {
        if (p.row_scale) rs[sl][it] = p.row_scale[fdiv24(m, p.rows_per_scale, inv_rps)];
        if (rfp32) { raw[sl][it][0] = ld_g<MVLT_NT_LD>((const u32x4*)((const float*)p.R + idx[sl][it])); raw[sl][it][1] = ld_g<MVLT_NT_LD>((const u32x4*)((const float*)p.R + idx[sl][it] + 4)); }
        else raw[sl][it][0] = ld_g<MVLT_NT_LD>((const u32x4*)((const bf16*)p.R + idx[sl][it]));
      }
    }
  };
  request(0);
#pragma unroll
  for (int half = 0; half < NH; ++half) {
    const int sl = half & 1;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) stage[(ii * 16 + 4 * fg + r) * LDW + j * 16 + fr] = acc[half * 2 + ii][j][r];
    if (half + 1 < NH) request(half + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (CHK && m0 + wm * (TM * 16) + half * 32 + rl[it] >= p.M) continue;
      const f32x4 v0 = *(const f32x4*)(stage + rl[it] * LDW + cc[it]), v1 = *(const f32x4*)(stage + rl[it] * LDW + cc[it] + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (p.bias) {
        const f32x4 b0 = *(const f32x4*)(p.bias + n0 + wn * WN + cc[it]), b1 = *(const f32x4*)(p.bias + n0 + wn * WN + cc[it] + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
      }
      if (EPI == 2) {
        float o8[8];
        if (rfp32) {
          const f32x4 a = __builtin_bit_cast(f32x4, raw[sl][it][0]), b = __builtin_bit_cast(f32x4, raw[sl][it][1]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { o8[e] = a[e]; o8[4 + e] = b[e]; }
        } else {
          const bf16x8 a = __builtin_bit_cast(bf16x8, raw[sl][it][0]);
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = (float)a[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * rs[sl][it] + o8[e];
      }
      if (ofp32) {
        st_g<MVLT_NT_GEMM>((f32x4*)((float*)p.C + idx[sl][it]), f32x4{v[0], v[1], v[2], v[3]});
        st_g<MVLT_NT_GEMM>((f32x4*)((float*)p.C + idx[sl][it] + 4), f32x4{v[4], v[5], v[6], v[7]});
      } else {
        bf16x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (bf16)v[e];
        st_g<MVLT_NT_GEMM>((bf16x8*)((bf16*)p.C + idx[sl][it]), a);
      }
    }
  }
}

template <int EPI, int HM, int HN0, int HN1, bool RAG = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_p8_kernel(mvlt_gemm_nt_args p) {
  // RAG: M and / or N are not whole tiles (the MLM logits: ~1500 selected rows x 30522 words).  The loader then points the rows past the end at
  // the last valid row -- their products are never stored: the epilogue runs with its bound checks -- which costs nothing inside the K-loop.
  constexpr int WMT = 2 * HM, WNT = HN0 + HN1;                    // accumulator tiles per wave
  constexpr int BMT = 2 * WMT * 16, BNT = 4 * WNT * 16;           // workgroup tile
  constexpr int AH = 2 * HM * 16, BH0 = 4 * HN0 * 16, BH1 = 4 * HN1 * 16;      // rows of the A / B0 / B1 half-tiles
  constexpr int A_IT = (AH + 63) / 64, B_IT0 = BH0 / 64, B_IT1 = BH1 / 64;     // DMA instructions per thread (the last A one: waves 0-3 only when AH = 96)
  constexpr bool A_PART = AH % 64 != 0;
  static_assert(BH0 % 64 == 0 && BH1 % 64 == 0 && (!A_PART || AH % 64 == 32), "half-tile geometry");
  constexpr int OFF_A1 = AH * 128, OFF_B0 = 2 * AH * 128, OFF_B1 = OFF_B0 + BH0 * 128, BUF = OFF_B1 + BH1 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];     // [2 buffers][A0 | A1 | B0 | B1]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_m = RAG ? (p.M + BMT - 1) / BMT : p.M / BMT, tiles_n = RAG ? (p.N + BNT - 1) / BNT : p.N / BNT;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, bslot = bid >> 3;
  const int tile_m = (bslot / tiles_n) * 8 + xcd, tile_n = bslot % tiles_n;      // the n-tiles of an m-tile are neighbours on one XCD
  if (tile_m >= tiles_m) return;
  const int m0 = tile_m * BMT, n0 = tile_n * BNT;
  const unsigned smem_lds = (unsigned)(uintptr_t)smem;
  const int nk = p.K >> 6;

  // ---- loader: DMA instruction i of a half-tile covers its rows 64 i .. 64 i + 63, thread -> (row (tid >> 3) + 64 i, slot tid & 7); the slot holds
  //      source chunk slot ^ ((row >> 1) & 7) (the read-side swizzle, an involution; rows 64 apart share the mask).  Source address = a scalar
  //      base per (operand, half, k-tile) + a per-thread 32-bit offset per instruction.
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((lrow >> 1) & 7);
  const unsigned a_rs = 2u * (unsigned)p.lda, b_rs = 2u * (unsigned)p.ldb;
  unsigned a_voff[A_IT], a_voff1[RAG ? A_IT : 1], b_voff0[B_IT0], b_voff1[B_IT1];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int lr = lrow + 64 * i, w_ = lr / (HM * 16), rem = lr - w_ * (HM * 16);
    const int trow = w_ * (WMT * 16) + rem;                     // row inside the tile, half 0; half 1 is HM * 16 rows further
    a_voff[i] = (unsigned)(RAG ? min(trow, p.M - 1 - m0) : trow) * a_rs + chunk * 16;
    if (RAG) a_voff1[i] = (unsigned)min(trow + HM * 16, p.M - 1 - m0) * a_rs + chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < B_IT0; ++i) {
    const int lr = lrow + 64 * i, w_ = lr / (HN0 * 16), rem = lr - w_ * (HN0 * 16);
    const int trow = w_ * (WNT * 16) + rem;
    b_voff0[i] = (unsigned)(RAG ? min(trow, p.N - 1 - n0) : trow) * b_rs + chunk * 16;
  }
#pragma unroll
  for (int i = 0; i < B_IT1; ++i) {
    const int lr = lrow + 64 * i, w_ = lr / (HN1 * 16), rem = lr - w_ * (HN1 * 16);
    const int trow = w_ * (WNT * 16) + HN0 * 16 + rem;
    b_voff1[i] = (unsigned)(RAG ? min(trow, p.N - 1 - n0) : trow) * b_rs + chunk * 16;
  }
  const char* const a_base = (const char*)p.A + (size_t)m0 * a_rs;
  const char* const b_base = (const char*)p.B + (size_t)n0 * b_rs;
  const unsigned dst_wave = smem_lds + wave * 1024;
  // which: 0 = A0, 1 = A1, 2 = B0, 3 = B1 (compile-time at every call site)
  auto stage = [&](int which, int t, int buf) {
    if (which < 2) {
      const char* sb = a_base + (RAG ? (size_t)0 : (size_t)(which * HM * 16) * a_rs) + t * 128;
      const unsigned dst = dst_wave + buf * BUF + which * OFF_A1;
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (!A_PART || i + 1 < A_IT || wave < 4) glds16_s(sb, (RAG && which == 1) ? a_voff1[i] : a_voff[i], dst + i * 8192);
    } else if (which == 2) {
      const char* sb = b_base + t * 128;
      const unsigned dst = dst_wave + buf * BUF + OFF_B0;
#pragma unroll
      for (int i = 0; i < B_IT0; ++i) glds16_s(sb, b_voff0[i], dst + i * 8192);
    } else {
      const char* sb = b_base + t * 128;
      const unsigned dst = dst_wave + buf * BUF + OFF_B1;
#pragma unroll
      for (int i = 0; i < B_IT1; ++i) glds16_s(sb, b_voff1[i], dst + i * 8192);
    }
  };
  // DMA instructions of THIS wave in the three youngest half-tiles (B0, A0, B1) = what the wait of phase 4 leaves in flight
  constexpr int INFL0 = B_IT0 + A_IT + B_IT1, INFL1 = B_IT0 + (A_PART ? A_IT - 1 : A_IT) + B_IT1;
  auto wait_tile = [&](bool more) {
    if (!more) wait_vm<0>();
    else if (wr == 0) wait_vm<INFL0>();
    else wait_vm<INFL1>();
  };

  // ---- fragment geometry: lane (fr, fg) reads row fr of a 16-row tile, chunk (ks * 4 + fg) ^ ((fr >> 1) & 7)
  const int fr = lane & 15, fg = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const char* const fa_base = smem + (wr * HM * 16 + fr) * 128;
  const char* const fb0_base = smem + OFF_B0 + (wc * HN0 * 16 + fr) * 128;
  const char* const fb1_base = smem + OFF_B1 + (wc * HN1 * 16 + fr) * 128;
  const int koff0 = ((fg ^ sw) & 7) << 4, koff1 = (((4 + fg) ^ sw) & 7) << 4;

  f32x4 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fa[2][HM], fb0[2][HN0], fb1[2][HN1];

#define MVLT_LDA(BUFI, MH)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < HM; ++i) {                                                                 \
    fa[0][i] = *(const u32x4*)(fa_base + (BUFI) * BUF + (MH) * OFF_A1 + i * 2048 + koff0);                         \
    fa[1][i] = *(const u32x4*)(fa_base + (BUFI) * BUF + (MH) * OFF_A1 + i * 2048 + koff1);                         \
  }
#define MVLT_LDB0(BUFI)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < HN0; ++j) {                                                                \
    fb0[0][j] = *(const u32x4*)(fb0_base + (BUFI) * BUF + j * 2048 + koff0);                                       \
    fb0[1][j] = *(const u32x4*)(fb0_base + (BUFI) * BUF + j * 2048 + koff1);                                       \
  }
#define MVLT_LDB1(BUFI)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < HN1; ++j) {                                                                \
    fb1[0][j] = *(const u32x4*)(fb1_base + (BUFI) * BUF + j * 2048 + koff0);                                       \
    fb1[1][j] = *(const u32x4*)(fb1_base + (BUFI) * BUF + j * 2048 + koff1);                                       \
  }
#define MVLT_MMA(MH, JBASE, HN, FB)                                                                                \
  do {                                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
      _Pragma("unroll") for (int i = 0; i < HM; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < (HN); ++j)                                                           \
          acc[(MH) * HM + i][(JBASE) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                               \
              __builtin_bit_cast(bf16x8, fa[ks][i]), __builtin_bit_cast(bf16x8, FB[ks][j]), acc[(MH) * HM + i][(JBASE) + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  } while (0)

  // ---- prologue: k-tile 0 whole, and the first three half-tiles of k-tile 1, in the loop's own issue order (B0, A0, B1, A1)
  stage(2, 0, 0); stage(0, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
  if (nk > 1) { stage(2, 1, 1); stage(0, 1, 1); stage(3, 1, 1); }
  wait_tile(nk > 1);
  MVLT_BAR();
  if (wr == 1) MVLT_BAR();                        // the second wave row runs half a phase behind the first

  auto ktile = [&](auto bufc, int t) {
    constexpr int B = decltype(bufc)::value;
    // phase 1: quadrant (0, 0); refill A1 of k-tile t + 1 (other buffer: last read in phase 3 of k-tile t - 1)
    MVLT_LDB0(B)
    __builtin_amdgcn_sched_barrier(0);
    MVLT_LDA(B, 0)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nk) stage(1, t + 1, B ^ 1);
    wait_lgkm<2 * HM>();                          // the B0 reads (issued first) are back: B0 may be refilled in phase 2
    MVLT_BAR();
    MVLT_MMA(0, 0, HN0, fb0);
    MVLT_BAR();
    // phase 2: quadrant (0, 1); refill B0 of k-tile t + 2
    MVLT_LDB1(B)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nk) stage(2, t + 2, B);
    MVLT_BAR();
    MVLT_MMA(0, HN0, HN1, fb1);
    MVLT_BAR();
    // phase 3: quadrant (1, 1); refill A0 of k-tile t + 2
    MVLT_LDA(B, 1)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nk) stage(0, t + 2, B);
    MVLT_BAR();
    MVLT_MMA(1, HN0, HN1, fb1);
    MVLT_BAR();
    // phase 4: quadrant (1, 0) from registers; refill B1 of k-tile t + 2; k-tile t + 1 must have landed when this phase ends
    if (t + 2 < nk) stage(3, t + 2, B);
    wait_tile(t + 2 < nk);
    MVLT_BAR();
    MVLT_MMA(1, 0, HN0, fb0);
    MVLT_BAR();
  };
  for (int t = 0; t < nk; t += 2) {
    ktile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nk) ktile(std::integral_constant<int, 1>{}, t + 1);
  }
  if (wr == 0) MVLT_BAR();
  MVLT_BAR();                                      // every wave is out of the loop: the epilogue may reuse the LDS
#undef MVLT_LDA
#undef MVLT_LDB0
#undef MVLT_LDB1
#undef MVLT_MMA
  // (whole tiles only: the bound checks go, -20 us on the GELU' launches; not for the residual epilogue, which measured 9 us SLOWER without them --
  //  49152 x 512 x 2048 + R 120 -> 129 us, same box, two passes: its prefetched rows are then requested in a different order)
  if constexpr (WNT == 4) nt_epilogue_lean<128, EPI, WMT, 4, EPI != 2 && !RAG>(p, acc, smem, m0, n0, wave, lane);
  else nt_epilogue_w80<EPI, WMT, RAG>(p, acc, smem, m0, n0, wave, lane);
}

template <int EPI, int HM, int HN0, int HN1, bool RAG = false> void launch_nt_p8(const mvlt_gemm_nt_args& a, hipStream_t s) {
  constexpr int BMT = 64 * HM, BNT = 64 * (HN0 + HN1);
  constexpr int LDS_LOOP = 2 * (2 * (2 * HM * 16) + 64 * (HN0 + HN1)) * 128;
  constexpr int LDS_EPI = 8 * 32 * (16 * (HN0 + HN1) + 4) * 4;
  constexpr int LDS = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
  static bool once = (hipFuncSetAttribute((const void*)gemm_nt_p8_kernel<EPI, HM, HN0, HN1, RAG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess);
  (void)once;
  const int tiles_m = (a.M + BMT - 1) / BMT, tiles_n = (a.N + BNT - 1) / BNT;
  dim3 grid((unsigned)(8 * ((tiles_m + 7) / 8) * tiles_n)), block(512);
  MVLT_LAUNCH((gemm_nt_p8_kernel<EPI, HM, HN0, HN1, RAG>), grid, block, LDS, s, a);
}
template <int HM, int HN0, int HN1> bool dispatch_nt_p8(const mvlt_gemm_nt_args& a, int epi, hipStream_t s) {
  switch (epi) {
    case 1: launch_nt_p8<1, HM, HN0, HN1>(a, s); return true;
    case 2: launch_nt_p8<2, HM, HN0, HN1>(a, s); return true;
    default: break;
  }
  if constexpr (HN0 + HN1 == 4) {
    switch (epi) {
      case 3: launch_nt_p8<3, HM, HN0, HN1>(a, s); return true;
      case 4: launch_nt_p8<4, HM, HN0, HN1>(a, s); return true;
      case 5: launch_nt_p8<5, HM, HN0, HN1>(a, s); return true;
      default: break;
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------------ TN (weight gradients), bf16, 8 waves, 8-phase loop
// C[N1, N2] += A[M, N1]^T B[M, N2] on the schedule of gemm_nt_p8_kernel: the reduction runs over the ROWS of both operands (k-tile = 64 rows), the
// operand tiles keep their natural [m][n] layout in LDS (LDS-DMA cannot transpose) and the MFMA fragments -- 8 consecutive m of one n per lane --
// come from two ds_read_b64_tr_b16 each, as in gemm_tn_dma_kernel.  Half h of A holds the columns {wr * 2 HM 16 + h * HM 16 + c} of both wave rows
// as a [64 m][2 HM 16] image, half h of B the columns of all four wave columns as [64 m][4 HNh 16]; row pitches 128 / 256 / 384 B, the 16-byte
// chunks of a row XOR-ed (on the source side) with a hash of the row so that the eight rows a 32-lane transposed read touches (m .. m+3, m+8 ..
// m+11) fall into eight different 32-byte bank windows: pitch 256 -> (row & 3) | bit 3 of the row; pitch 128 and 384 (both 4 windows mod 8 per
// row) -> bit 1 | bit 3.  Tiles: 256 x 256 (2048 x 512 at stage 4) and 128 x 320 (1280 x 320 at stage 3); an output whose 320-multiple side is N1 is
// computed as its transpose (operands swapped by the host, the MFMA operand order flipped so that a lane's 16 consecutive outputs stay contiguous
// in memory).  The m range is split over workgroups (one per CU), partial tiles meet by fp32 atomics; bias gradients = ones-fragment MFMAs,
// taken in turns by the workgroups that share an operand column range.
template <int PITCH> __device__ __forceinline__ int tn_hash(int row) {
  // pitch 192 (round 6, the 96-column A half-tile of the 192 x 320 tile): a row advances 6 windows mod 8, so rows m .. m+3 sit in windows {0, 6, 4, 2} + w and rows
  // m+8 .. m+11 in the same ones: bit 3 of the row flips the window's low bit ({1, 7, 5, 3} + w) -- eight distinct windows for either read of a fragment
  if (PITCH == 192) return (row >> 3) & 1;
  return PITCH == 256 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}
// Round 6: the 192 x 320 tile (<3, 3, 2>: 30 accumulator tiles per wave, as in gemm_nt_p8_kernel) for the fc weight gradients of stage 3 -- 1280 x 320 and 320 x 1280 over
// 98304 (pvlt_medium at 384 px: 45056) rows.  The 320 side is one column tile; the 1280 side is 6.67 row tiles: RAG1 = the last row tile is ragged (its loader columns are
// clamped to the last 8 valid ones, its rows / column sums past N1 are not stored: 7 tiles for 6.67 of work).  SWAP = the caller's N1 is the 320 side: the host swaps the
// operands (A' = B, B' = A) and this instantiation stores its partial tiles TRANSPOSED, i.e. in the caller's [N1][N2] layout (needs the un-flipped MFMA operand order: a
// lane then owns four consecutive kernel-n1 = caller-n2 of one caller-n1).
template <int HM, int HN0, int HN1, bool TRANS, bool RAG1 = false, bool SWAP = false>
__global__ __launch_bounds__(512, 2) void gemm_tn_p8_kernel(mvlt_gemm_tn_args p, int kt_per, int t1, int t2, int splits, bf16* part) {
  constexpr int WMT = 2 * HM, WNT = HN0 + HN1;
  constexpr int AC = 2 * HM * 16, BC0 = 4 * HN0 * 16, BC1 = 4 * HN1 * 16;          // columns of the A / B0 / B1 half-tiles
  constexpr int PA = AC * 2, PB0 = BC0 * 2, PB1 = BC1 * 2;                         // row pitches in bytes
  constexpr int A_IT = (64 * PA + 8191) / 8192, B_IT0 = 64 * PB0 / 8192, B_IT1 = 64 * PB1 / 8192;
  constexpr bool A_PART = (64 * PA) % 8192 != 0;                                   // 12 KB half-tile: the second DMA instruction belongs to waves 0-3 only
  static_assert((PA == 128 || PA == 192 || PA == 256) && (PB0 == 256 || PB0 == 384) && PB1 == 256, "half-tile pitches with a bank hash");
  static_assert(!SWAP || !TRANS, "the transposed partial store needs the un-flipped accumulator layout");
  constexpr int OFF_A1 = 64 * PA, OFF_B0 = 2 * 64 * PA, OFF_B1 = OFF_B0 + 64 * PB0, BUF = OFF_B1 + 64 * PB1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // all output tiles of one m-split on ONE XCD (workgroup b runs on XCD b % 8): the operand rows pass that L2 once
  const int txy = t1 * t2;
  int bz, xy;
  if (splits >= 8) {
    const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
    const int zq = kq / txy;
    xy = kq - zq * txy;
    bz = zq * 8 + xcd;
    if (bz >= splits) return;
  } else {
    bz = blockIdx.x / txy;
    xy = blockIdx.x - bz * txy;
  }
  const int bx = xy % t1, by = xy / t1;
  const int n1_0 = bx * (2 * WMT * 16), n2_0 = by * (4 * WNT * 16);
  const int nkt = p.M >> 6;
  const int kt0 = bz * kt_per;
  const int nk = min(kt_per, nkt - kt0);
  if (nk <= 0) return;
  const unsigned smem_lds = (unsigned)(uintptr_t)smem;
  const unsigned a_rs = 2u * (unsigned)p.lda, b_rs = 2u * (unsigned)p.ldb;

  // ---- loader: chunk q = tid + 512 i of a half-tile = (row q / CPR, slot q % CPR); the slot holds source chunk slot ^ (hash(row) << 1)
  unsigned a_voff[A_IT], a_voff1[RAG1 ? A_IT : 1], b_voff0[B_IT0], b_voff1[B_IT1];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    constexpr int CPR = PA / 16;
    const int q = tid + 512 * i, row = (q / CPR) & 63, lc = ((q - (q / CPR) * CPR) ^ (tn_hash<PA>(row) << 1)) * 8;      // (& 63: the idle half of a partial instruction)
    const int w_ = lc / (HM * 16);
    const int col = w_ * (WMT * 16) + (lc - w_ * (HM * 16));                      // column inside the tile, half 0; half 1 is HM * 16 further
    a_voff[i] = (unsigned)row * a_rs + 2u * (unsigned)(RAG1 ? min(col, p.N1 - 8 - n1_0) : col);
    if (RAG1) a_voff1[i] = (unsigned)row * a_rs + 2u * (unsigned)min(col + HM * 16, p.N1 - 8 - n1_0);
  }
#pragma unroll
  for (int i = 0; i < B_IT0; ++i) {
    constexpr int CPR = PB0 / 16;
    const int q = tid + 512 * i, row = q / CPR, lc = ((q - row * CPR) ^ (tn_hash<PB0>(row) << 1)) * 8;
    const int w_ = lc / (HN0 * 16);
    b_voff0[i] = (unsigned)row * b_rs + 2u * (unsigned)(w_ * (WNT * 16) + (lc - w_ * (HN0 * 16)));
  }
#pragma unroll
  for (int i = 0; i < B_IT1; ++i) {
    constexpr int CPR = PB1 / 16;
    const int q = tid + 512 * i, row = q / CPR, lc = ((q - row * CPR) ^ (tn_hash<PB1>(row) << 1)) * 8;
    const int w_ = lc / (HN1 * 16);
    b_voff1[i] = (unsigned)row * b_rs + 2u * (unsigned)(w_ * (WNT * 16) + HN0 * 16 + (lc - w_ * (HN1 * 16)));
  }
  const char* const a_base = (const char*)p.A + (size_t)(kt0 * 64) * a_rs + (size_t)n1_0 * 2;
  const char* const b_base = (const char*)p.B + (size_t)(kt0 * 64) * b_rs + (size_t)n2_0 * 2;
  const unsigned dst_wave = smem_lds + wave * 1024;
  auto stage = [&](int which, int t, int buf) {
    if (which < 2) {
      const char* sb = a_base + (size_t)(t * 64) * a_rs + (RAG1 ? 0 : which * (HM * 16 * 2));
      const unsigned dst = dst_wave + buf * BUF + which * OFF_A1;
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        if (!A_PART || i + 1 < A_IT || wave < 4) glds16_s(sb, (RAG1 && which == 1) ? a_voff1[i] : a_voff[i], dst + i * 8192);
    } else if (which == 2) {
      const char* sb = b_base + (size_t)(t * 64) * b_rs;
      const unsigned dst = dst_wave + buf * BUF + OFF_B0;
#pragma unroll
      for (int i = 0; i < B_IT0; ++i) glds16_s(sb, b_voff0[i], dst + i * 8192);
    } else {
      const char* sb = b_base + (size_t)(t * 64) * b_rs;
      const unsigned dst = dst_wave + buf * BUF + OFF_B1;
#pragma unroll
      for (int i = 0; i < B_IT1; ++i) glds16_s(sb, b_voff1[i], dst + i * 8192);
    }
  };
  // DMA instructions of THIS wave in the three youngest half-tiles (B0, A0, B1): what the wait of phase 4 leaves in flight
  constexpr int INFL0 = B_IT0 + A_IT + B_IT1, INFL1 = B_IT0 + (A_PART ? A_IT - 1 : A_IT) + B_IT1;
  auto wait_infl = [&]() { if (wr == 0) wait_vm<INFL0>(); else wait_vm<INFL1>(); };

  // ---- fragment geometry (transposed reads): lane (g, L) supplies k-row 8 g + (L >> 2) and the row 4 below, 8-byte piece L & 3 of a 16-column
  //      window; the k32 step ks adds 32 rows
  const int g = lane >> 4, L = lane & 15;
  const int frow = 8 * g + (L >> 2);
  int aoff[HM], boff0[HN0], boff1[HN1];
#pragma unroll
  for (int i = 0; i < HM; ++i) aoff[i] = frow * PA + (((wr * HM + i) ^ tn_hash<PA>(frow)) << 5) + ((L & 3) << 3);
#pragma unroll
  for (int j = 0; j < HN0; ++j) boff0[j] = OFF_B0 + frow * PB0 + (((wc * HN0 + j) ^ tn_hash<PB0>(frow)) << 5) + ((L & 3) << 3);
#pragma unroll
  for (int j = 0; j < HN1; ++j) boff1[j] = OFF_B1 + frow * PB1 + (((wc * HN1 + j) ^ tn_hash<PB1>(frow)) << 5) + ((L & 3) << 3);

  f32x4 acc[WMT][WNT];
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradients: column sums of A (colsum_a) or of B (colsum_b) by one MFMA against an all-ones fragment.  The A fragments of accumulator row tile
  // i are read by the four waves of a wave row: wave wc takes i == wc; the B fragments of column tile j by the two wave rows: j & 1 == wr.
  f32x4 csa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  f32x4 csb[(WNT + 1) / 2];
#pragma unroll
  for (int j = 0; j < (WNT + 1) / 2; ++j) csb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
  const bool do_csa = p.colsum_a != nullptr, do_csb = p.colsum_b != nullptr;
  u32x4 fa[2][HM], fb0[2][HN0], fb1[2][HN1];

#define MVLT_TLDA(BUFI, MH)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < HM; ++i) {                                                                 \
    fa[0][i] = tr_frag(smem + (BUFI) * BUF + (MH) * OFF_A1 + aoff[i], PA);                                         \
    fa[1][i] = tr_frag(smem + (BUFI) * BUF + (MH) * OFF_A1 + aoff[i] + 32 * PA, PA);                               \
  }
#define MVLT_TLDB0(BUFI)                                                                                           \
  _Pragma("unroll") for (int j = 0; j < HN0; ++j) {                                                                \
    fb0[0][j] = tr_frag(smem + (BUFI) * BUF + boff0[j], PB0);                                                      \
    fb0[1][j] = tr_frag(smem + (BUFI) * BUF + boff0[j] + 32 * PB0, PB0);                                           \
  }
#define MVLT_TLDB1(BUFI)                                                                                           \
  _Pragma("unroll") for (int j = 0; j < HN1; ++j) {                                                                \
    fb1[0][j] = tr_frag(smem + (BUFI) * BUF + boff1[j], PB1);                                                      \
    fb1[1][j] = tr_frag(smem + (BUFI) * BUF + boff1[j] + 32 * PB1, PB1);                                           \
  }
#define MVLT_TMMA(MH, JBASE, HN, FB)                                                                               \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                 \
    _Pragma("unroll") for (int i = 0; i < HM; ++i)                                                                 \
      _Pragma("unroll") for (int j = 0; j < (HN); ++j) {                                                           \
        f32x4& c_ = acc[(MH) * HM + i][(JBASE) + j];                                                               \
        if (TRANS) c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, FB[ks][j]), __builtin_bit_cast(bf16x8, fa[ks][i]), c_, 0, 0, 0); \
        else c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ks][i]), __builtin_bit_cast(bf16x8, FB[ks][j]), c_, 0, 0, 0);       \
      }
#define MVLT_TCSA(MH)                                                                                              \
  if (csa_now) {                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
      _Pragma("unroll") for (int i = 0; i < HM; ++i)                                                               \
        if (i == wc) csa[MH] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ks][i]), ones, csa[MH], 0, 0, 0); \
  }
#define MVLT_TCSB(JBASE, HN, FB)                                                                                   \
  if (csb_now) {                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
      _Pragma("unroll") for (int j = 0; j < (HN); ++j)                                                             \
        if ((((JBASE) + j) & 1) == wr) csb[((JBASE) + j) >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, FB[ks][j]), csb[((JBASE) + j) >> 1], 0, 0, 0); \
  }

  stage(2, 0, 0); stage(0, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
  if (nk > 1) { stage(2, 1, 1); stage(0, 1, 1); stage(3, 1, 1); wait_infl(); }
  else wait_vm<0>();
  MVLT_BAR();
  if (wr == 1) MVLT_BAR();

  auto ktile = [&](auto bufc, int t) {
    constexpr int B = decltype(bufc)::value;
    const bool csa_now = do_csa && (t % t2) == by, csb_now = do_csb && (t % t1) == bx;       // the workgroups sharing A (B) columns take turns
    MVLT_TLDB0(B)
    __builtin_amdgcn_sched_barrier(0);
    MVLT_TLDA(B, 0)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nk) stage(1, t + 1, B ^ 1);
    wait_lgkm<(4 * HM < 15 ? 4 * HM : 15)>();     // the B0 reads (two 8-byte reads per fragment, issued first) are back (the counter has 4 bits)
    MVLT_BAR();
    __builtin_amdgcn_s_setprio(1);
    MVLT_TMMA(0, 0, HN0, fb0)
    MVLT_TCSA(0)
    MVLT_TCSB(0, HN0, fb0)
    __builtin_amdgcn_s_setprio(0);
    MVLT_BAR();
    MVLT_TLDB1(B)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nk) stage(2, t + 2, B);
    MVLT_BAR();
    __builtin_amdgcn_s_setprio(1);
    MVLT_TMMA(0, HN0, HN1, fb1)
    MVLT_TCSB(HN0, HN1, fb1)
    __builtin_amdgcn_s_setprio(0);
    MVLT_BAR();
    MVLT_TLDA(B, 1)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nk) stage(0, t + 2, B);
    MVLT_BAR();
    __builtin_amdgcn_s_setprio(1);
    MVLT_TMMA(1, HN0, HN1, fb1)
    MVLT_TCSA(1)
    __builtin_amdgcn_s_setprio(0);
    MVLT_BAR();
    if (t + 2 < nk) { stage(3, t + 2, B); wait_infl(); }
    else wait_vm<0>();
    MVLT_BAR();
    __builtin_amdgcn_s_setprio(1);
    MVLT_TMMA(1, 0, HN0, fb0)
    __builtin_amdgcn_s_setprio(0);
    MVLT_BAR();
  };
  for (int t = 0; t < nk; t += 2) {
    ktile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nk) ktile(std::integral_constant<int, 1>{}, t + 1);
  }
  if (wr == 0) MVLT_BAR();
#undef MVLT_TLDA
#undef MVLT_TLDB0
#undef MVLT_TLDB1
#undef MVLT_TMMA
#undef MVLT_TCSA
#undef MVLT_TCSB
  const int fr = lane & 15, fg = lane >> 4;
  if (do_csa && fr == 0 && wc < HM) {            // csa[mh][r]: column n1 = tile (mh * HM + wc), row 4 fg + r of it (identical in every lane column)
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n1 = n1_0 + wr * (WMT * 16) + (mh * HM + wc) * 16 + 4 * fg + r;
        if (!RAG1 || n1 < p.N1) atomicAdd(&p.colsum_a[n1], csa[mh][r]);
      }
  }
  if (do_csb && fg == 0) {                        // csb[j >> 1][0]: column n2 = tile j, lane column fr
#pragma unroll
    for (int j = 0; j < WNT; ++j)
      if ((j & 1) == wr) atomicAdd(&p.colsum_b[n2_0 + wc * (WNT * 16) + j * 16 + fr], csb[j >> 1][0]);
  }
  if constexpr (SWAP) {
    // PARTIAL-TILE mode, operands swapped by the host (kernel n1 = the caller's n2 and vice versa): the split's tile goes to part[split][caller N1 = p.N2][caller N2 = p.N1],
    // the caller's layout, so that the same fold serves it.  Un-flipped accumulators: acc[i][j][r] = C'[n1 = tile i row 4 fg + r][n2 = tile j column fr]: a lane owns four
    // consecutive kernel-n1 of one kernel-n2 = four consecutive columns of one row of the caller's matrix.  Per wave and kernel-n2 tile j: a [16 n2][WMT * 16 n1] LDS tile,
    // out as whole 16-byte pieces of contiguous rows.
    bf16* const P = part + (size_t)bz * p.N1 * p.N2;
    constexpr int LDP = WMT * 16 + 8;
    constexpr int CPR = WMT * 2;
    MVLT_BAR();
    bf16* const st = (bf16*)smem + wave * 16 * LDP;
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < WMT; ++i) {
        const bf16x2 lo = __builtin_convertvector(f32x2{acc[i][j][0], acc[i][j][1]}, bf16x2), hi = __builtin_convertvector(f32x2{acc[i][j][2], acc[i][j][3]}, bf16x2);
        *(u32x2*)(st + fr * LDP + i * 16 + 4 * fg) = u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = lane; q < 16 * CPR; q += 64) {
        const int row = q / CPR, ch = q - row * CPR;
        const int n2 = n2_0 + wc * (WNT * 16) + j * 16 + row, n1 = n1_0 + wr * (WMT * 16) + ch * 8;
        if (!RAG1 || n1 < p.N1) st_g<MVLT_NT_GEMM>((u32x4*)(P + (size_t)n2 * p.N1 + n1), *(const u32x4*)(st + row * LDP + ch * 8));
      }
    }
    return;
  } else if (part) {
    // PARTIAL-TILE mode (round 5): no atomics.  The split's tile goes to part[split][N1][N2] in bf16 -- the MFMA operands are flipped (TRANS instantiation), so a lane owns
    // four consecutive n2 of one n1: one 8-byte store per accumulator tile -- and tn_fold_kernel adds the splits' tiles into C in a fixed order (deterministic).
    static_assert(TRANS, "partial tiles are stored from the flipped accumulator layout");
    bf16* const P = part + (size_t)bz * p.N1 * p.N2;
    // through a per-wave LDS tile [16 n1][WNT * 16 n2] so that the partial tile leaves as whole 16-byte pieces of contiguous rows (8-byte pieces 32 B apart cost 22 us per launch)
    constexpr int LDP = WNT * 16 + 8;              // bf16 elements per staged row (16 B of padding)
    constexpr int CPR = WNT * 2;                   // 16-byte chunks per row
    MVLT_BAR();                                    // every wave is out of the loop: the k-tile buffers are free
    bf16* const st = (bf16*)smem + wave * 16 * LDP;
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const bf16x2 lo = __builtin_convertvector(f32x2{acc[i][j][0], acc[i][j][1]}, bf16x2), hi = __builtin_convertvector(f32x2{acc[i][j][2], acc[i][j][3]}, bf16x2);
        *(u32x2*)(st + fr * LDP + j * 16 + 4 * fg) = u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = lane; q < 16 * CPR; q += 64) {
        const int row = q / CPR, ch = q - row * CPR;
        const int n1 = n1_0 + wr * (WMT * 16) + i * 16 + row, n2 = n2_0 + wc * (WNT * 16) + ch * 8;
        if (!RAG1 || n1 < p.N1) st_g<MVLT_NT_GEMM>((u32x4*)(P + (size_t)n1 * p.N2 + n2), *(const u32x4*)(st + row * LDP + ch * 8));
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < WMT; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (TRANS) {                              // acc[r] = C[n2 tile j row 4 fg + r][n1 tile i column fr], stored transposed: consecutive lanes = consecutive n1
          const int n2 = n2_0 + wc * (WNT * 16) + j * 16 + 4 * fg + r, n1 = n1_0 + wr * (WMT * 16) + i * 16 + fr;
          atomicAdd(&p.C[(long)n2 * p.ldc + n1], acc[i][j][r]);
        } else {
          const int n1 = n1_0 + wr * (WMT * 16) + i * 16 + 4 * fg + r, n2 = n2_0 + wc * (WNT * 16) + j * 16 + fr;
          atomicAdd(&p.C[(long)n1 * p.ldc + n2], acc[i][j][r]);
        }
      }
}

// C[n1][n2] += sum over the splits of part[split][n1][n2] (bf16 partial tiles of gemm_tn_p8_kernel / gemm_tn_dma_kernel).  A workgroup covers 32 groups of eight consecutive n2;
// its 256 threads are 32 groups x 8 split subsets (a 320 x 320 output over 56 splits is only 12800 groups: one thread per group left 200 CUs idle and ran the 56 loads of a
// group one behind the other), the subsets meet in LDS and are added in a FIXED order: the result does not depend on timing.
__device__ __forceinline__ void tn_fold_body(const bf16* __restrict__ part, int splits, int N1, int N2, float* __restrict__ C, int ldc, int blk) {
  __shared__ float red[8][32][9];
  const int eg = threadIdx.x & 31, sk = threadIdx.x >> 5;
  const long g = (long)blk * 32 + eg;
  const long per = (long)N1 * N2;
  const bool ok = g * 8 < per;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ok) {
    for (int z = sk; z < splits; z += 8) {
      const bf16x8 v = __builtin_bit_cast(bf16x8, ld_g<MVLT_NT_LD>((const u32x4*)(part + (size_t)z * per + g * 8)));
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[sk][eg][e] = s[e];
  __syncthreads();
  if (sk == 0 && ok) {
#pragma unroll
    for (int k = 1; k < 8; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += red[k][eg][e];
    const int n1 = (int)((g * 8) / N2), n2 = (int)((g * 8) - (long)n1 * N2);
    float* dst = C + (long)n1 * ldc + n2;
    const f32x4 a = *(const f32x4*)dst, b = *(const f32x4*)(dst + 4);
    *(f32x4*)dst = f32x4{a[0] + s[0], a[1] + s[1], a[2] + s[2], a[3] + s[3]};
    *(f32x4*)(dst + 4) = f32x4{b[0] + s[4], b[1] + s[5], b[2] + s[6], b[3] + s[7]};
  }
}

__global__ __launch_bounds__(256) void tn_fold_kernel(const bf16* __restrict__ part, int splits, int N1, int N2, float* __restrict__ C, int ldc) {
  tn_fold_body(part, splits, N1, N2, C, ldc, blockIdx.x);
}
// several folds in one launch: workgroup b belongs to the descriptor whose [wg0, next wg0) holds it
__global__ __launch_bounds__(256) void tn_fold_multi_kernel(FoldBatch b) {
  int i = 0;
#pragma unroll 1
  for (int k = 1; k < b.n; ++k) if ((int)blockIdx.x >= b.d[k].wg0) i = k;
  const FoldDesc& d = b.d[i];
  tn_fold_body(d.part, d.splits, d.N1, d.N2, d.C, d.ldc, (int)blockIdx.x - d.wg0);
}

template <int HM, int HN0, int HN1> int launch_tn_p8_partial(const mvlt_gemm_tn_args& a, hipStream_t s) {
  constexpr int BM1 = 64 * HM, BN2 = 64 * (HN0 + HN1);
  constexpr int LDS = 2 * 64 * 2 * (2 * (2 * HM * 16) + BN2);
  mvlt_max_lds<(gemm_tn_p8_kernel<HM, HN0, HN1, true>)>();
  const int t1 = a.N1 / BM1, t2 = a.N2 / BN2, nkt = a.M / 64;
  int splits = 256 / (t1 * t2);                   // one workgroup per CU (the caller's scratch is sized for exactly this)
  if (splits > nkt) splits = nkt;
  const int kt_per = (nkt + splits - 1) / splits;
  splits = (nkt + kt_per - 1) / kt_per;
  bf16* const scratch = fold_acquire(a, (long)splits * a.N1 * a.N2 * 2, s);      // [splits][N1][N2] bf16 in the caller's scratch (mvlt_gemm_tn checked its size)
  dim3 grid((unsigned)((splits >= 8 ? 8 * ((splits + 7) / 8) : splits) * t1 * t2)), block(512);
  MVLT_LAUNCH((gemm_tn_p8_kernel<HM, HN0, HN1, true>), grid, block, LDS, s, a, kt_per, t1, t2, splits, scratch);
  fold_launch(a, scratch, splits, s);
  return mvlt_check_launch("mvlt_gemm_tn");
}

// 192 x 320 tiles (round 6): the caller's matrix is [N1][N2] with ONE side a multiple of 320 (taken as the kernel's column side, whole tiles) and the other covered by
// ceil(. / 192) row tiles, the last one ragged.  swap: the 320 side is the caller's N1 -- operands exchanged, partial tiles stored transposed (in the caller's layout).
template <bool SWAP> int launch_tn_p8_320(const mvlt_gemm_tn_args& a, hipStream_t s) {
  constexpr int LDS = 2 * 64 * 2 * (2 * (2 * 3 * 16) + 320);
  mvlt_gemm_tn_args k = a;
  if (SWAP) {
    k.A = a.B; k.B = a.A; k.lda = a.ldb; k.ldb = a.lda; k.N1 = a.N2; k.N2 = a.N1;
    k.colsum_a = a.colsum_b; k.colsum_b = a.colsum_a;
  }
  const int t1 = (k.N1 + 191) / 192, t2 = k.N2 / 320, nkt = k.M / 64;
  // all tiles of an m-split run on ONE XCD (32 CUs, one workgroup each): whole splits per XCD, the same number on each -- 7 tiles: 4 splits per XCD = 32 splits = 224
  // workgroups (36 splits = 252 workgroups put 35 on four of the XCDs: a second round there, 184 us instead of the 128-wide kernel's 137)
  int splits = (32 / (t1 * t2)) * 8;
  if (splits < 8) splits = 256 / (t1 * t2);
  if (splits > nkt) splits = nkt;
  const int kt_per = (nkt + splits - 1) / splits;
  splits = (nkt + kt_per - 1) / kt_per;
  bf16* const scratch = fold_acquire(a, (long)splits * a.N1 * a.N2 * 2, s);      // [splits][caller N1][caller N2] bf16
  if (!scratch) return 1;                                                          // (does not fit: the caller falls through to the 128-wide kernel)
  dim3 grid((unsigned)((splits >= 8 ? 8 * ((splits + 7) / 8) : splits) * t1 * t2)), block(512);
  const bool rag = k.N1 % 192 != 0;
  if (SWAP) {
    if (rag) { mvlt_max_lds<(gemm_tn_p8_kernel<3, 3, 2, false, true, true>)>(); MVLT_LAUNCH((gemm_tn_p8_kernel<3, 3, 2, false, true, true>), grid, block, LDS, s, k, kt_per, t1, t2, splits, scratch); }
    else { mvlt_max_lds<(gemm_tn_p8_kernel<3, 3, 2, false, false, true>)>(); MVLT_LAUNCH((gemm_tn_p8_kernel<3, 3, 2, false, false, true>), grid, block, LDS, s, k, kt_per, t1, t2, splits, scratch); }
  } else {
    if (rag) { mvlt_max_lds<(gemm_tn_p8_kernel<3, 3, 2, true, true, false>)>(); MVLT_LAUNCH((gemm_tn_p8_kernel<3, 3, 2, true, true, false>), grid, block, LDS, s, k, kt_per, t1, t2, splits, scratch); }
    else { mvlt_max_lds<(gemm_tn_p8_kernel<3, 3, 2, true, false, false>)>(); MVLT_LAUNCH((gemm_tn_p8_kernel<3, 3, 2, true, false, false>), grid, block, LDS, s, k, kt_per, t1, t2, splits, scratch); }
  }
  fold_launch(a, scratch, splits, s);
  return mvlt_check_launch("mvlt_gemm_tn");
}

// ------------------------------------------------------------------------------------------------ conv3x3 forward / dgrad, LDS halo
// C[pixel][n] = sum over taps t and channels c of x[pixel + tap t][c] * B[n][t*cin + c]: the MIM decoder's conv3x3 (and its input
// gradient, the same gather with flipped taps) as the NT GEMM with a_map mode 2.  In gemm_nt_dma_kernel every k-step fetches its own
// shifted copy of the 128 input rows through the LDS-DMA path, nine times per 64 channels.  Here the K loop runs channel slice
// outermost: per 64-channel slice the tile's HALO -- (128/W + 2) x (W + 2) input rows, 26 KB at W = 32 -- is loaded once and all
// nine taps read their A fragments from it at shifted row addresses (ds_read_b128, the row's 16-B chunks XOR-swizzled by the halo
// row, conflict-free from any first row); only the weight tiles still stream per k-step.  A traffic per 128 x 192-channel tile: 78 KB instead of 432 KB.  Same tile
// and wave geometry as gemm_nt_dma_kernel (128 x BN, 4 waves, two workgroups per CU), so its epilogues are used unchanged.
template <int W, int BN, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void conv3_nt_kernel(mvlt_gemm_nt_args p) {
  constexpr int BK = 64, ROWB = 128, CH = 8;
  constexpr int RPL = NTHREADS / CH;
  constexpr int WN = BN / 2, TN_ = WN / 16, TM_ = 4;
  constexpr int B_ITERS = BN * CH / NTHREADS;
  constexpr int R = BM / W, HW2 = W + 2, HR = (R + 2) * HW2;          // image rows per tile; halo rows (one pixel, 64 channels = 128 B)
  constexpr int H_IT = (HR * 8 + NTHREADS - 1) / NTHREADS;
  constexpr int HALO_BYTES = H_IT * NTHREADS * 16;
  constexpr int BSTAGE = BN * ROWB;
  auto swzk = [](int row, int chunk) { return chunk ^ ((row >> 1) & 7); };
  // halo rows are read at arbitrary (tap-shifted) offsets: chunk ^ (((row >> 1) & 3) << 1) keeps every 16-lane group of a
  // ds_read_b128 -- 16 consecutive rows, lanes 4..11 of them one chunk further -- on 16 distinct 16-B bank slots from ANY first row
  // (the (row >> 1) & 7 form used for the 16-aligned tiles is 2-way conflicted from three first rows in four)
  auto swzh = [](int row) { return ((row >> 1) & 3) << 1; };
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [halo slice | B stage 0 | B stage 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = p.M / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, bgroup = bid >> 3;
  const int tile_m = (bgroup / tiles_n) * 8 + xcd, tile_n = bgroup % tiles_n;
  if (tile_m >= tiles_m) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const unsigned smem_lds = (unsigned)(uintptr_t)smem;
  const int Himg = p.a_map.h_in, cin = p.a_map.c_seg, tokens_in = p.a_map.tokens_in;
  const int tiles_per_img = Himg * W / BM;
  const int img = tile_m / tiles_per_img, y0 = (tile_m - img * tiles_per_img) * R;
  const char* zsrc = (const char*)g_zero_page + ((tid * 16 + (bid & 15) * 4096) & 65535);
  const unsigned a_rowb = 2u * (unsigned)p.lda;
  const char* img_base = (const char*)p.A + (unsigned long long)(unsigned)(img * tokens_in) * a_rowb;

  // ---- halo loader: LDS row hr = (hy, hx) of the padded window; slot s of row hr holds source chunk swzk(hr, s)
  const char* h_src[H_IT];
#pragma unroll
  for (int j = 0; j < H_IT; ++j) {
    const int q = tid + j * NTHREADS, hr = q >> 3, sl = q & 7;
    const int hy = hr / HW2, hx = hr - hy * HW2;
    const int y = y0 + hy - 1, x = hx - 1;
    const bool ok = hr < HR && (unsigned)y < (unsigned)Himg && (unsigned)x < (unsigned)W;
    h_src[j] = ok ? img_base + (long)(y * W + x) * (long)a_rowb + 2 * ((sl ^ swzh(hr)) << 3) : nullptr;
  }
  auto issue_halo = [&](int kc) {
#pragma unroll
    for (int j = 0; j < H_IT; ++j)
      glds16(h_src[j] ? h_src[j] + kc * 128 : zsrc, __builtin_amdgcn_readfirstlane(smem_lds + (j * NTHREADS + wave * 64) * 16));
  };
  // ---- weight loader (as gemm_nt_dma_kernel): row n of the tile, chunk swizzled by row
  const int row_in = tid / CH;
  const int chunk = swzk(row_in, tid % CH);
  const char* b_ptr[B_ITERS];
  bool b_ok[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    int n = n0 + row_in + RPL * i;
    b_ok[i] = n < p.N;
    b_ptr[i] = (const char*)p.B + (unsigned long long)(unsigned)(b_ok[i] ? n : 0) * (2u * (unsigned)p.ldb) + chunk * 16;
  }
  auto issue_b = [&](int kc, int t, int slot) {
    const int koff = (t * cin + kc * 64) * 2;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i)
      glds16(b_ok[i] ? b_ptr[i] + koff : zsrc, __builtin_amdgcn_readfirstlane(smem_lds + HALO_BYTES + slot * BSTAGE + (i * NTHREADS + wave * 64) * 16));
  };

  f32x4 acc[TM_][TN_];
#pragma unroll
  for (int i = 0; i < TM_; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  // halo row of this lane's pixel in each of the wave's four 16-pixel tiles, tap (0, 0)
  int hbase[TM_];
#pragma unroll
  for (int i = 0; i < TM_; ++i) {
    const int pix = wm * 64 + i * 16 + fr, py = pix / W, px = pix - py * W;
    hbase[i] = (py + 1) * HW2 + px + 1;
  }
  const int nkc = cin / 64;
  int bslot = 0;
  for (int kc = 0; kc < nkc; ++kc) {
    // the previous slice's A reads (and the B stage refilled below) are done for every wave
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_halo(kc);
    if (kc == 0) issue_b(0, 0, bslot);                     // later slices: their first weight tile was issued during the previous tap 8
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // weight tile (kc, t) (and a fresh halo) landed; reads of the other stage done
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // all pieces of the next weight tile right behind the barrier: one piece between each pair of MFMA groups instead was 5 % slower
      if (t + 1 < 9) issue_b(kc, t + 1, bslot ^ 1);
      else if (kc + 1 < nkc) issue_b(kc + 1, 0, bslot ^ 1);
      const int tapoff = (t / 3 - 1) * HW2 + (t % 3 - 1);
      const char* b_s = smem + HALO_BYTES + bslot * BSTAGE + (wn * WN) * ROWB;
      bslot ^= 1;
      u32x4 fa[2][TM_], fb[2][TN_];
#pragma unroll
      for (int i = 0; i < TM_; ++i) {
        const int hr = hbase[i] + tapoff;
        const char* rowp = smem + hr * 128;
        const int sw = swzh(hr);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[ks][i] = *(const u32x4*)(rowp + (((ks * 4 + fg) ^ sw) << 4));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < TN_; ++j) {
          int r = j * 16 + fr;
          fb[ks][j] = *(const u32x4*)(b_s + r * ROWB + swzk(wn * WN + r, ks * 4 + fg) * 16);
        }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM_; ++i)
#pragma unroll
          for (int j = 0; j < TN_; ++j) mma16(acc[i][j], fa[ks][i], fa[ks][i], fb[ks][j], fb[ks][j], (bf16*)nullptr);
      }
    }
  }
  __syncthreads();                   // last reads are done before the epilogue reuses the LDS
  if constexpr (BN == 192) nt_epilogue_192<EPI>(p, acc, smem, m0, n0, wave, lane);
  else nt_epilogue_lean<BN, EPI, TM_>(p, acc, smem, m0, n0, wave, lane);
}

template <int W, int BN, int EPI> void launch_conv3_nt(const mvlt_gemm_nt_args& a, hipStream_t s) {
  constexpr int R = BM / W, HR = (R + 2) * (W + 2), H_IT = (HR * 8 + NTHREADS - 1) / NTHREADS;
  size_t lds = (size_t)H_IT * NTHREADS * 16 + (size_t)2 * BN * ROW_BYTES;
  const size_t stage = (size_t)4 * 32 * (BN / 2 + 4) * sizeof(float);
  if (lds < stage) lds = stage;
  const int tiles_m = a.M / BM, tiles_n = (a.N + BN - 1) / BN;
  dim3 grid((unsigned)(8 * ((tiles_m + 7) / 8) * tiles_n)), block(NTHREADS);
  mvlt_max_lds<(conv3_nt_kernel<W, BN, EPI>)>();
  MVLT_LAUNCH((conv3_nt_kernel<W, BN, EPI>), grid, block, lds, s, a);
}
template <int W> bool dispatch_conv3_nt(const mvlt_gemm_nt_args& a, int epi, hipStream_t s) {
  const bool wide = a.N % 192 == 0 && a.N % 128 != 0 && (epi == 1 || epi == 5) && a.c_map.mode == 0;
  if (wide) { if (epi == 1) launch_conv3_nt<W, 192, 1>(a, s); else launch_conv3_nt<W, 192, 5>(a, s); return true; }
  if (a.N <= 64) {
    switch (epi) {
      case 1: launch_conv3_nt<W, 64, 1>(a, s); return true;
      case 2: launch_conv3_nt<W, 64, 2>(a, s); return true;
      case 5: launch_conv3_nt<W, 64, 5>(a, s); return true;
      default: return false;
    }
  }
  switch (epi) {
    case 1: launch_conv3_nt<W, 128, 1>(a, s); return true;
    case 2: launch_conv3_nt<W, 128, 2>(a, s); return true;
    case 5: launch_conv3_nt<W, 128, 5>(a, s); return true;
    default: return false;
  }
}

int check_rowmap(const mvlt_rowmap& m, const char* who) {
  if (m.mode == 0) {
    MVLT_REQUIRE(m.rows_per_batch >= 0, "%s: rows_per_batch < 0", who);
  } else if (m.mode == 1) {
    MVLT_REQUIRE(m.r > 0 && m.w_in > 0 && m.tokens_in > 0 && m.hw_out > 0 && m.w_out > 0 && m.c_seg > 0 && m.c_seg % 16 == 0,
                 "%s: bad patch map (c_seg must be a positive multiple of 16)", who);
  } else if (m.mode == 2) {
    MVLT_REQUIRE(m.r == 3 && m.w_in > 0 && m.h_in > 0 && m.tokens_in >= m.h_in * m.w_in && m.hw_out == m.h_in * m.w_in &&
                 m.w_out == m.w_in && m.c_seg > 0 && m.c_seg % 8 == 0, "%s: bad 3x3 neighbourhood map", who);
  } else {
    MVLT_REQUIRE(false, "%s: unknown rowmap mode %d", who, m.mode);
  }
  return MVLT_OK;
}

}  // namespace

extern "C" int mvlt_gemm_nt(const mvlt_gemm_nt_args* a, void* stream) {
  MVLT_REQUIRE(a && a->A && a->B && a->C, "mvlt_gemm_nt: null operand");
  MVLT_REQUIRE(a->M >= 0 && a->N > 0 && a->K > 0, "mvlt_gemm_nt: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
  MVLT_REQUIRE(a->dtype == 0 || a->dtype == 1, "mvlt_gemm_nt: dtype must be 0 (bf16) or 1 (fp32)");
  const int pc = a->dtype == 0 ? 8 : 4;
  MVLT_REQUIRE(a->K % pc == 0 && a->lda % pc == 0 && a->ldb % pc == 0, "mvlt_gemm_nt: K/lda/ldb must be multiples of %d elements (16 B)", pc);
  MVLT_REQUIRE(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "mvlt_gemm_nt: A/B must be 16-byte aligned");
  MVLT_REQUIRE(a->act >= 0 && a->act <= 2, "mvlt_gemm_nt: bad act");
  MVLT_REQUIRE(a->act != 2 || a->H, "mvlt_gemm_nt: act=2 (gelu') needs H");
  MVLT_REQUIRE(!a->row_scale || a->rows_per_scale > 0, "mvlt_gemm_nt: row_scale needs rows_per_scale");
  MVLT_REQUIRE((a->col_sum == nullptr) == (a->col_sumsq == nullptr), "mvlt_gemm_nt: col_sum and col_sumsq come together");
  MVLT_REQUIRE(a->split_k <= 1 || (a->dtype == 0 && a->out_dtype == 1 && a->act == 0 && !a->H && !a->R && !a->row_scale && !a->col_sum &&
                                   a->c_map.mode == 0 && a->split_k <= 64),
               "mvlt_gemm_nt: split_k needs bf16 operands, fp32 C (zeroed by the caller) and a plain epilogue (A may be gathered)");
  MVLT_REQUIRE(a->col_copies >= 0, "mvlt_gemm_nt: col_copies < 0");
  MVLT_REQUIRE(!a->r_fp32 || (a->R && a->R != a->C && a->dtype == 0 && a->out_dtype == 0 && a->act == 0 && !a->col_sum && !a->post_y && a->split_k <= 1 && a->c_map.mode == 0 &&
                              a->N % 8 == 0 && a->ldc % 8 == 0 && (((uintptr_t)a->C | (uintptr_t)a->R) & 15) == 0 && a->M < (1 << 24) && !getenv("MVLT_NT_GENERIC_EPI")),
               "mvlt_gemm_nt: r_fp32 (fp32 residual beside a bf16 C) exists in the residual epilogue of the bf16 LDS-DMA kernels only (plain c_map, N % 8 == 0, 16-byte aligned C / R)");
  MVLT_REQUIRE(a->out_dtype >= 0 && a->out_dtype <= 2, "mvlt_gemm_nt: out_dtype is 0 (bf16), 1 (fp32) or 2 (fp16, with col_sum only)");
  MVLT_REQUIRE(a->out_dtype != 2 || (a->dtype == 0 && a->col_sum && a->act == 0 && !a->R && !a->row_scale && a->split_k <= 1 && a->c_map.mode == 0 && a->N % 8 == 0 &&
                                     a->ldc % 8 == 0 && ((uintptr_t)a->C & 15) == 0 && a->M < (1 << 24) && !getenv("MVLT_NT_GENERIC_EPI")),
               "mvlt_gemm_nt: fp16 output exists in the column-statistics epilogue of the bf16 LDS-DMA kernels only (plain c_map, N % 8 == 0, 16-byte aligned C)");
  if (int e = check_rowmap(a->a_map, "mvlt_gemm_nt a_map")) return e;
  if (int e = check_rowmap(a->c_map, "mvlt_gemm_nt c_map")) return e;
  MVLT_REQUIRE(a->a_map.mode == 0 || a->K == a->a_map.r * a->a_map.r * a->a_map.c_seg, "mvlt_gemm_nt: gather K != r*r*c_seg");
  MVLT_REQUIRE(a->c_map.mode == 0 || a->N == a->c_map.r * a->c_map.r * a->c_map.c_seg, "mvlt_gemm_nt: scatter N != r*r*c_seg");
  MVLT_REQUIRE(a->c_map.mode != 2, "mvlt_gemm_nt: the 3x3 map is a gather only (its dgrad is a gather with flipped taps)");
  // EPI 8 (LayerNorm of the finished row) exists in the bf16 LDS-DMA kernel only: every other launch path refuses instead of skipping it silently
  MVLT_REQUIRE(!a->post_y || (a->dtype == 0 && a->split_k <= 1),
               "mvlt_gemm_nt: post_y needs the bf16 LDS-DMA path (no fp32 operands, no split_k)");
  if (a->M == 0) return MVLT_OK;
  hipStream_t s = (hipStream_t)stream;
  const int tiles_m = (a->M + BM - 1) / BM;
  const bool narrow = a->N <= 64;
  const int bn = narrow ? 64 : 128;
  const int tiles_n = (a->N + bn - 1) / bn;
  const int bk = a->dtype == 0 ? 64 : 32;
  const int nbuf = a->K <= bk ? 1 : 2;
  size_t lds = (size_t)nbuf * (BM + bn) * ROW_BYTES;
  const size_t stage = (size_t)4 * 32 * (bn / 2 + 4) * sizeof(float);      // epilogue staging (4 waves x 32 rows)
  if (lds < stage) lds = stage;
  dim3 grid((unsigned)(8 * ((tiles_m + 7) / 8) * tiles_n), (unsigned)(a->split_k > 1 ? a->split_k : 1)), block(NTHREADS);
  if (a->dtype == 0) {
    const int nk = ((a->K + 63) / 64 + (a->split_k > 1 ? a->split_k : 1) - 1) / (a->split_k > 1 ? a->split_k : 1);
    // compile-time epilogue variant (see nt_epilogue_lean); 0 = generic
    int epi = 0;
    // (N % 8 != 0 -- the 30522-word MLM logits -- takes the plain lean epilogue too: its last chunk of a row is stored column by column)
    const bool plain_epi = a->act == 0 && !a->R && !a->row_scale && !a->col_sum && !a->H && a->c_map.mode == 0;
    const bool lean_ok = (a->c_map.mode == 0 || (a->c_map.mode == 1 && a->c_map.c_seg % 8 == 0)) && a->split_k <= 1 && (a->N % 8 == 0 || plain_epi) && a->ldc % 8 == 0 && ((uintptr_t)a->C & 15) == 0 &&
                         (!a->R || ((uintptr_t)a->R & 15) == 0) && (!a->H || ((uintptr_t)a->H & 15) == 0) && a->M < (1 << 24) &&
                         !getenv("MVLT_NT_GENERIC_EPI");
    if (lean_ok && a->c_map.mode == 1) {
      if (a->act == 0 && !a->col_sum && !a->R && !a->row_scale) epi = 6;
      else if (a->act == 0 && !a->col_sum && a->R) epi = 7;
    } else if (lean_ok) {
      if (a->act == 1 && !a->R && !a->row_scale && !a->col_sum && a->out_dtype == 0) epi = 3;
      else if (a->act == 2 && !a->R && !a->row_scale && !a->col_sum && a->out_dtype == 0) epi = 4;
      else if (a->act == 0 && a->R && !a->col_sum) epi = 2;
      else if (a->act == 0 && !a->R && !a->row_scale && a->col_sum) epi = 5;
      else if (a->act == 0 && !a->R && !a->row_scale && !a->col_sum) epi = 1;
    }
    // ring depth: 2 stages (64 KB, two workgroups per CU).  A single stage at four workgroups per CU was tried: same time in
    // the step, and its extra in-loop issue path cost 34 VGPRs (a wave per SIMD on the K <= 64 launches)
    // The GELU epilogue (EPI 3) on the short-K fc1 GEMMs is epilogue-bound: 32-wide K stages halve the ring (32 KB) so a third
    // workgroup fits per CU and covers it (98304x1280x320: 240 -> 220 us, 49152x2048x512: 234 -> 201 us).  Long K loses to the
    // doubled barrier count (K = 2048: 103 -> 125 us), other epilogues are neutral; EPI 4 prefers its two-half H prefetch,
    // whose registers allow two workgroups per CU either way (220 / 213 us against 227 / 217 us).
    const int bkd = (!narrow && a->a_map.mode == 0 && (epi == 3 || (epi == 4 && !getenv("MVLT_NT_EPI4_BK64"))) && a->K <= 512 && !getenv("MVLT_NT_BK64")) ? 32 : 64;
    const int nkd = bkd == 32 ? (a->K + 31) / 32 : nk;
    int ns = nkd < 2 ? nkd : 2;
    if (const char* e = getenv("MVLT_NT_NS")) { ns = atoi(e); if (ns > nkd) ns = nkd; if (ns < 2) ns = nkd < 2 ? nkd : 2; if (ns > 6) ns = 6; }
    size_t lds2 = (size_t)ns * (BM + bn) * (bkd * 2);
    if (lds2 < stage) lds2 = stage;
    static const int early_flag = getenv("MVLT_NT_EARLY") ? (atoi(getenv("MVLT_NT_EARLY")) ? 0x100 : 0) : MVLT_NT_EARLY_DEFAULT;
    const int ns_lds = ns;
    if (nkd >= 2) ns |= early_flag;                // (a single k-step has nothing to refill)
    if (a->post_y) {                            // EPI 8: attn.proj + residual + Block.norm2 (whole rows in one tile)
      MVLT_REQUIRE(epi == 2 && a->N == bn && a->a_map.mode == 0 && a->c_map.mode == 0 && a->c_map.rows_per_batch == 0 && a->post_gamma && a->post_beta &&
                   a->post_mean && a->post_rstd && a->post_ld % 8 == 0 && ((uintptr_t)a->post_y & 15) == 0 && ((uintptr_t)a->post_gamma & 15) == 0,
                   "mvlt_gemm_nt: post_y needs bf16 operands, a residual (R), N == 64 or 128, identity row maps, 16-byte aligned outputs");
      if (lds2 < stage + 2048) lds2 = stage + 2048;
      if (narrow) MVLT_LAUNCH((gemm_nt_dma_kernel<64, 0, 8, 64>), grid, block, lds2, s, *a, ns);
      else MVLT_LAUNCH((gemm_nt_dma_kernel<128, 0, 8, 64>), grid, block, lds2, s, *a, ns);
      return mvlt_check_launch("mvlt_gemm_nt");
    }
#define MVLT_NT_LAUNCH_E(BN_, AM_, BK_)                                                                                 \
  do {                                                                                                               \
    switch (epi) {                                                                                                   \
      case 1: MVLT_LAUNCH((gemm_nt_dma_kernel<BN_, AM_, 1, BK_>), grid, block, lds2, s, *a, ns); break;            \
      case 2: MVLT_LAUNCH((gemm_nt_dma_kernel<BN_, AM_, 2, BK_>), grid, block, lds2, s, *a, ns); break;            \
      case 3: MVLT_LAUNCH((gemm_nt_dma_kernel<BN_, AM_, 3, BK_>), grid, block, lds2, s, *a, ns); break;            \
      case 4: MVLT_LAUNCH((gemm_nt_dma_kernel<BN_, AM_, 4, BK_>), grid, block, lds2, s, *a, ns); break;            \
      case 5: MVLT_LAUNCH((gemm_nt_dma_kernel<BN_, AM_, 5, BK_>), grid, block, lds2, s, *a, ns); break;            \
      case 6: MVLT_LAUNCH((gemm_nt_dma_kernel<BN_, AM_, 6, BK_>), grid, block, lds2, s, *a, ns); break;            \
      case 7: MVLT_LAUNCH((gemm_nt_dma_kernel<BN_, AM_, 7, BK_>), grid, block, lds2, s, *a, ns); break;            \
      default: MVLT_LAUNCH((gemm_nt_dma_kernel<BN_, AM_, 0, BK_>), grid, block, lds2, s, *a, ns);                  \
    }                                                                                                                \
  } while (0)
#define MVLT_NT_LAUNCH(BN_, BK_)                                                                                     \
  do {                                                                                                               \
    if (a->a_map.mode == 0) MVLT_NT_LAUNCH_E(BN_, 0, BK_);                                                           \
    else if (a->a_map.mode == 1) MVLT_NT_LAUNCH_E(BN_, 1, BK_);                                                      \
    else MVLT_NT_LAUNCH_E(BN_, 2, BK_);                                                                              \
  } while (0)
    // conv3x3 (a_map mode 2) on an LDS-resident, channel-sliced halo: grids of width 16 / 32 / 64, whole 128-pixel tiles inside one
    // image, 64-multiples of gathered channels, lean epilogue, identity or batch-strided output rows
    static const bool conv_nt_ok = !getenv("MVLT_NO_CONV_NT");
    if (conv_nt_ok && a->a_map.mode == 2 && (epi == 1 || epi == 2 || epi == 5) && a->split_k <= 1 && a->a_map.c_seg % 64 == 0 && a->c_map.mode == 0 &&
        (a->a_map.w_in == 16 || a->a_map.w_in == 32 || a->a_map.w_in == 64) && (a->a_map.h_in * a->a_map.w_in) % BM == 0 && a->M % BM == 0 &&
        a->M % (a->a_map.h_in * a->a_map.w_in) == 0 && a->a_map.hw_out == a->a_map.h_in * a->a_map.w_in && a->a_map.w_out == a->a_map.w_in &&
        a->K == 9 * a->a_map.c_seg && a->lda >= a->a_map.c_seg && a->ldb >= a->K) {
      bool done = a->a_map.w_in == 32 ? dispatch_conv3_nt<32>(*a, epi, s) : a->a_map.w_in == 16 ? dispatch_conv3_nt<16>(*a, epi, s) : dispatch_conv3_nt<64>(*a, epi, s);
      if (done) return mvlt_check_launch("mvlt_gemm_nt");
    }
    // 8-wave kernels with the 8-phase K-loop (gemm_nt_p8_kernel) for the MFMA-bound shapes of the stage 3-4 MLPs: one workgroup per CU, so the
    // tile height is chosen by whole rounds of 256 CUs (rows x rounds = time): 256 x 256, 192 x 256, or 192 x 320 for N % 320 == 0
    static const int ntp8 = getenv("MVLT_NT_P8") ? atoi(getenv("MVLT_NT_P8")) : 15;      // bit 0: 256 x 256, bit 1: 192 x 256, bit 2: 192 x 320, bit 3: ragged 256 x 256 (EPI 1)
    if (ntp8 && a->a_map.mode == 0 && a->a_map.rows_per_batch == 0 && a->c_map.mode == 0 && epi >= 1 && epi <= 5 && a->K % 64 == 0 && a->K >= 128) {
      auto cost = [&](int bm, int bn) -> long {                         // rows x rounds; 0 = shape does not fit
        if (a->M % bm || a->N % bn) return 0;
        const long tiles = (long)(a->M / bm) * (a->N / bn);
        if (tiles < 192) return 0;
        return ((tiles + 255) / 256) * (long)bm * bn;
      };
      const long c256 = (ntp8 & 1) ? cost(256, 256) : 0, c192 = (ntp8 & 2) ? cost(192, 256) : 0;
      // (192 x 320 with the fp32 residual epilogue pays from K = 640 on: 98304 x 320 x 320 + R 93 us against 73 us on the 128-wide kernel, K = 1280 138 against 155)
      const long c320 = ((ntp8 & 4) && a->N % 256 != 0 && (epi == 1 || (epi == 2 && (a->K >= 640 || (ntp8 & 16))))) ? cost(192, 320) : 0;
      bool done = false;
      // ragged M / N (the MLM logits, ~1500 x 30522 x 768, fp32 out): 256 x 256 tiles with the loader clamped at the last row and the checked epilogue
      if (!c320 && !c256 && !c192 && epi == 1 && (ntp8 & 8) && (a->M % 256 || a->N % 256)) {
        const long tiles = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
        // ... when the whole rounds of padded tiles are at least 85 % real work: 294912 x 128 would pad half of every 256-wide tile (36 -> 56 us), and
        // 1558 selected rows make 7 x 120 tiles = 4 rounds where 1490 rows make 3 (the 128-wide kernel then wins)
        const long rounds = (tiles + 255) / 256;
        if (tiles >= 384 && (double)a->M * a->N >= 0.85 * (double)rounds * 256.0 * 65536.0) { launch_nt_p8<1, 4, 2, 2, true>(*a, s); done = true; }
      }
      // ragged M on the 192 x 320 tile (round 6; N % 320 == 0, same epilogue rules as the whole-tile form): pvlt_medium at 384 px runs its 18 stage-3 blocks at
      // M = 45056 = 234.67 x 192 rows -- 235 tiles = ONE round at 92 % -- and took the 128-wide kernels for every N = 320 product (fc2 76 us, fc1 input gradient 66 us,
      // q / proj 27 us: profiles/r06_medium384_gemm_shapes.txt); taken when the whole rounds of padded tiles are >= 85 % real work
      if (!done && !c320 && !c256 && !c192 && (ntp8 & 4) && a->N % 320 == 0 && a->N % 256 != 0 && a->M % 192 != 0 && (epi == 1 || (epi == 2 && (a->K >= 640 || (ntp8 & 16))))) {
        const long tiles = (long)((a->M + 191) / 192) * (a->N / 320);
        const long rounds = (tiles + 255) / 256;
        if (tiles >= 192 && (double)a->M * a->N >= 0.85 * (double)rounds * 256.0 * 192.0 * 320.0) {
          if (epi == 1) launch_nt_p8<1, 3, 3, 2, true>(*a, s);
          else launch_nt_p8<2, 3, 3, 2, true>(*a, s);
          done = true;
        }
      }
      if (done) return mvlt_check_launch("mvlt_gemm_nt");
      if (c320) done = dispatch_nt_p8<3, 3, 2>(*a, epi, s);
      else if (c256 && (!c192 || c256 <= c192)) done = dispatch_nt_p8<4, 2, 2>(*a, epi, s);
      else if (c192) done = dispatch_nt_p8<3, 2, 2>(*a, epi, s);
      if (done) return mvlt_check_launch("mvlt_gemm_nt");
    }
    // N % 192 == 0 (the 192-channel convolutions): one 192-wide tile instead of 128 + a half-empty 128
    const bool wide = !narrow && a->N % 192 == 0 && a->N % 128 != 0 && (epi == 1 || epi == 5) && a->c_map.mode == 0 && a->a_map.mode != 1 &&
                      a->K >= 128 && !getenv("MVLT_NT_NO192");
    if (wide) {
      const int tn192 = a->N / 192;
      dim3 grid192((unsigned)(8 * ((tiles_m + 7) / 8) * tn192), 1);
      size_t lds3 = (size_t)ns_lds * (BM + 192) * ROW_BYTES;
      const size_t stage192 = (size_t)4 * 32 * 100 * sizeof(float);
      if (lds3 < stage192) lds3 = stage192;
      if (a->a_map.mode == 0 && epi == 1) MVLT_LAUNCH((gemm_nt_dma_kernel<192, 0, 1, 64>), grid192, block, lds3, s, *a, ns);
      else if (a->a_map.mode == 0) MVLT_LAUNCH((gemm_nt_dma_kernel<192, 0, 5, 64>), grid192, block, lds3, s, *a, ns);
      else if (epi == 1) MVLT_LAUNCH((gemm_nt_dma_kernel<192, 2, 1, 64>), grid192, block, lds3, s, *a, ns);
      else MVLT_LAUNCH((gemm_nt_dma_kernel<192, 2, 5, 64>), grid192, block, lds3, s, *a, ns);
    } else if (narrow) MVLT_NT_LAUNCH(64, 64);
    else if (bkd == 32 && epi == 3) MVLT_LAUNCH((gemm_nt_dma_kernel<128, 0, 3, 32>), grid, block, lds2, s, *a, ns);
    else if (bkd == 32) MVLT_LAUNCH((gemm_nt_dma_kernel<128, 0, 4, 32>), grid, block, lds2, s, *a, ns);
    else MVLT_NT_LAUNCH(128, 64);
#undef MVLT_NT_LAUNCH_E
#undef MVLT_NT_LAUNCH
  } else {
    if (narrow) MVLT_LAUNCH((gemm_nt_kernel<float, 64>), grid, block, lds, s, *a, nbuf);
    else MVLT_LAUNCH((gemm_nt_kernel<float, 128>), grid, block, lds, s, *a, nbuf);
  }
  return mvlt_check_launch("mvlt_gemm_nt");
}

bf16* mvlt_fold_acquire_ext(const mvlt_gemm_tn_args& a, long need, hipStream_t s, int descriptors) { return fold_acquire(a, need, s, descriptors); }
void mvlt_fold_launch_ext(const mvlt_gemm_tn_args& a, const bf16* part, int splits, hipStream_t s) { fold_launch(a, part, splits, s); }

extern "C" int mvlt_tn_fold_discard(const void* partials) {
  // forget the pending folds of one scratch (NULL: of every scratch) without launching anything: their producers belong to a pass that was abandoned
  std::lock_guard<std::mutex> lk(g_fold_mu);
  for (auto& kv : g_fold)
    if (!partials || kv.first == partials) { kv.second.b.n = 0; kv.second.used = 0; }
  return MVLT_OK;
}

extern "C" int mvlt_tn_fold_flush(const void* partials, void* stream) {
  // the pending folds run on the stream their producers ran on; `stream` -- the stream of whoever reads the gradients next -- waits for them when it is another one
  std::lock_guard<std::mutex> lk(g_fold_mu);
  for (auto& kv : g_fold)
    if (!partials || kv.first == partials) fold_flush_locked(kv.second, (hipStream_t)stream, true);
  return mvlt_check_launch("mvlt_tn_fold_flush");
}

extern "C" int mvlt_gemm_tn(const mvlt_gemm_tn_args* a, void* stream) {
  MVLT_REQUIRE(a && a->A && a->B && a->C, "mvlt_gemm_tn: null operand");
  MVLT_REQUIRE(a->M >= 0 && a->N1 > 0 && a->N2 > 0, "mvlt_gemm_tn: bad shape");
  MVLT_REQUIRE(a->dtype == 0 || a->dtype == 1, "mvlt_gemm_tn: dtype must be 0 (bf16) or 1 (fp32)");
  const int pc = a->dtype == 0 ? 8 : 4;
  // N1/N2 may be ragged (30522 vocabulary rows) as long as the padded row (lda/ldb) covers the last 16-B chunk
  MVLT_REQUIRE(a->lda % pc == 0 && a->ldb % pc == 0, "mvlt_gemm_tn: lda/ldb must be multiples of %d elements (16 B)", pc);
  MVLT_REQUIRE(a->lda >= (a->N1 + pc - 1) / pc * pc, "mvlt_gemm_tn: lda must cover N1 rounded up to %d", pc);
  MVLT_REQUIRE(a->b_map.mode != 0 || a->ldb >= (a->N2 + pc - 1) / pc * pc, "mvlt_gemm_tn: ldb must cover N2 rounded up to %d", pc);
  MVLT_REQUIRE(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "mvlt_gemm_tn: A/B must be 16-byte aligned");
  if (int e = check_rowmap(a->a_map, "mvlt_gemm_tn a_map")) return e;
  if (int e = check_rowmap(a->b_map, "mvlt_gemm_tn b_map")) return e;
  MVLT_REQUIRE(a->a_map.mode == 0, "mvlt_gemm_tn: A cannot be a patch gather");
  MVLT_REQUIRE(!(a->colsum_a && a->colsum_b), "mvlt_gemm_tn: at most one of colsum_a / colsum_b");
  MVLT_REQUIRE(a->b_map.mode == 0 || a->N2 == a->b_map.r * a->b_map.r * a->b_map.c_seg, "mvlt_gemm_tn: gather N2 != r*r*c_seg");
  MVLT_REQUIRE(a->c_taps <= 1 || (a->trans_c == 0 && a->c_seg > 0 && a->N2 == a->c_taps * a->c_seg), "mvlt_gemm_tn: c_taps needs trans_c == 0 and N2 == c_taps*c_seg");
  MVLT_REQUIRE(!a->dgrad_out || (a->dtype == 0 && a->M < (1 << 24) && a->b_map.mode == 0 && a->a_map.mode == 0),
               "mvlt_gemm_tn: dgrad_out rides on the bf16 LDS-DMA kernel only (bf16 operands, M < 2^24, plain row maps): every other path would leave it unwritten");
  if (a->M == 0) return MVLT_OK;
  hipStream_t s = (hipStream_t)stream;
  const int mtiles = (a->M + TBK - 1) / TBK;
  // conv3x3 weight gradient with an LDS-resident halo (conv3_wgrad_kernel): 3x3 gather on B over a W x H grid with W in {8, 16, 32, 64},
  // whole 64-pixel k-tiles inside one image, 64-multiples of channels, plain [out][tap*cin + c] output
  static const bool conv_wgrad_ok = !getenv("MVLT_NO_CONV_WGRAD");
  if (conv_wgrad_ok && a->dtype == 0 && a->b_map.mode == 2 && a->a_map.mode == 0 && a->a_map.rows_per_batch == 0 && !a->trans_c && a->c_taps <= 1 &&
      !a->colsum_a && !a->colsum_b && a->N1 % 64 == 0 && a->b_map.c_seg % 64 == 0 && a->N2 == 9 * a->b_map.c_seg && a->M % 64 == 0 &&
      (a->b_map.w_in == 8 || a->b_map.w_in == 16 || a->b_map.w_in == 32 || a->b_map.w_in == 64) && (a->b_map.h_in * a->b_map.w_in) % 64 == 0 &&
      a->M % (a->b_map.h_in * a->b_map.w_in) == 0 && a->ldb >= a->b_map.c_seg) {
    hipStream_t s2 = (hipStream_t)stream;
    if (a->b_map.w_in == 64) return launch_conv3_wgrad<64>(*a, s2);
    if (a->b_map.w_in == 32) return launch_conv3_wgrad<32>(*a, s2);
    if (a->b_map.w_in == 16) return launch_conv3_wgrad<16>(*a, s2);
    return launch_conv3_wgrad<8>(*a, s2);
  }
  // (Round 4 built this kernel's 8-wave / 8-phase sibling -- 256 x 256 and 128 x 320 output tiles, one workgroup per CU, commit 6d0e8d3 -- and
  //  removed it again: its main loop ran at 1.1-1.16 PFLOP/s, but one workgroup per CU means 16 m-splits of the 2048 x 512 outputs, and the fp32
  //  atomics that combine the splits complete at ~0.3 floats per ns chip-wide whatever their scope or coalescing: 55 us of tail per launch,
  //  147 us against the 122 us of the 128 x 128 tiles below with their 8 splits and a second workgroup per CU to hide the tail.  DESIGN.md 6.)
  // whole 256 x 256 output tiles, 16 .. 64 of them (= 16 .. 4 m-splits for one workgroup per CU), and a scratch buffer from the caller: the 8-wave / 8-phase TN loop with
  // bf16 partial tiles + an ordered fold instead of fp32 atomics (gemm_tn_p8_kernel; MVLT_TN_P8=0 keeps the atomic path).  Fewer tiles mean more splits than the fold is
  // worth (8 tiles: 75 us either way), more do not occur in this model.
  static const bool tnp8 = !(getenv("MVLT_TN_P8") && atoi(getenv("MVLT_TN_P8")) == 0);
  if (tnp8 && a->partials && a->dtype == 0 && a->a_map.mode == 0 && a->b_map.mode == 0 && a->a_map.rows_per_batch == 0 && a->b_map.rows_per_batch == 0 && a->c_taps <= 1 &&
      !a->trans_c && !a->dgrad_out && a->M % 64 == 0 && a->N1 % 256 == 0 && a->N2 % 256 == 0 && a->ldc % 4 == 0 && ((uintptr_t)a->C & 15) == 0 && ((uintptr_t)a->partials & 15) == 0) {
    const int tiles = (a->N1 / 256) * (a->N2 / 256), nkt = a->M / 64;
    if (tiles >= 16 && tiles <= 64 && nkt / (256 / tiles) >= 16 && a->partials_bytes >= (long)(256 / tiles) * a->N1 * a->N2 * 2)
      return launch_tn_p8_partial<4, 2, 2>(*a, s);
  }
  // round 6: one side a multiple of 320, the other >= 1024 (the fc weight gradients of stage 3: 1280 x 320 and 320 x 1280): 192 x 320 tiles of the 8-phase TN loop, 6-16 of
  // them, the last row tile ragged when it is >= 85 % full overall; bf16 partial tiles + the ordered fold (MVLT_TN_P8_320=0: the 128-wide kernel as in round 5)
  static const bool tn320 = !(getenv("MVLT_TN_P8_320") && atoi(getenv("MVLT_TN_P8_320")) == 0);
  if (tnp8 && tn320 && a->partials && a->dtype == 0 && a->a_map.mode == 0 && a->b_map.mode == 0 && a->a_map.rows_per_batch == 0 && a->b_map.rows_per_batch == 0 && a->c_taps <= 1 &&
      !a->trans_c && !a->dgrad_out && a->M % 64 == 0 && a->N1 % 8 == 0 && a->N2 % 8 == 0 && a->ldc % 4 == 0 && ((uintptr_t)a->C & 15) == 0 && ((uintptr_t)a->partials & 15) == 0 &&
      a->lda % 8 == 0 && a->ldb % 8 == 0 && (((uintptr_t)a->A | (uintptr_t)a->B) & 15) == 0) {
    const bool direct = a->N2 % 320 == 0 && a->N1 >= 1024, swapped = !direct && a->N1 % 320 == 0 && a->N2 >= 1024;
    if (direct || swapped) {
      const int rows = direct ? a->N1 : a->N2, cols = direct ? a->N2 : a->N1;
      const int tiles = ((rows + 191) / 192) * (cols / 320), nkt = a->M / 64;
      if (tiles >= 6 && tiles <= 16 && nkt / (256 / tiles) >= 16 && (double)rows >= 0.85 * 192.0 * ((rows + 191) / 192)) {
        const int rc = direct ? launch_tn_p8_320<false>(*a, s) : launch_tn_p8_320<true>(*a, s);
        if (rc <= 0) return rc;
      }
    }
  }
  if (a->dtype == 0 && a->M < (1 << 24)) {
    // LDS-DMA kernel: A tile 128 or 64 wide, B tile 128 or 64 wide; ~1024 workgroups, splits a multiple of the 8 XCDs
    // outputs of at most 128 x 128 take 64 x 64 tiles: every output cache line receives one atomic request per m-split, those
    // serialise at the memory side (~100 ns each), and four small tiles need a quarter of the splits of one big tile for the same
    // number of workgroups (294912 x 128 x 128: 61 -> 35 us)
    if (a->dgrad_out) {
      // weight gradient + input gradient of a C x C Linear from one pass over dY (gemm_tn_dma_kernel<.., DG>): one output tile, plain rows
      MVLT_REQUIRE(a->dgrad_wt && a->N1 == a->N2 && (a->N1 == 64 || a->N1 == 128) && a->a_map.rows_per_batch == 0 && a->b_map.mode == 0 &&
                   a->b_map.rows_per_batch == 0 && !a->trans_c && a->c_taps <= 1 && !a->colsum_b && a->dgrad_ld % 8 == 0 &&
                   (((uintptr_t)a->dgrad_out | (uintptr_t)a->dgrad_wt) & 15) == 0,
                   "mvlt_gemm_tn: dgrad_out needs bf16 operands, plain rows, N1 == N2 == 64 or 128, no transposed / tap output, 16-byte aligned buffers");
      int splits = a->splits > 0 ? a->splits : (a->N1 == 64 ? 256 : 384);       // 1081344 x 64: 128 / 192 / 256 / 384 / 512 splits 136 / 90 / 83 / 91 / 85 us; 294912 x 128: 75 / 63 / 60 / 59 / 68
      static const int min_tiles = getenv("MVLT_TN_MINT") ? atoi(getenv("MVLT_TN_MINT")) : 8;
      if (min_tiles > 1 && splits > mtiles / min_tiles) splits = mtiles / min_tiles;
      if (splits >= 8) splits = (splits + 4) / 8 * 8;
      if (splits > mtiles) splits = mtiles;
      if (splits < 1) splits = 1;
      const int m_per_split = ((mtiles + splits - 1) / splits) * TBK;
      splits = (a->M + m_per_split - 1) / m_per_split;
      dim3 grid((unsigned)(splits >= 8 ? 8 * ((splits + 7) / 8) : splits)), block(NTHREADS);
      if (a->N1 == 64) {
        const size_t lds = (size_t)4 * TBK * 128 * 2 + 64 * 64 * 2;
        static bool once = (hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<64, 64, 3, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess);
        (void)once;
        MVLT_LAUNCH((gemm_tn_dma_kernel<64, 64, 3, 4, true>), grid, block, lds, s, *a, m_per_split, 1, 1, splits, (bf16*)nullptr);
      } else {
        const size_t lds = (size_t)2 * TBK * 256 * 2 + 64 * 128 * 2;
        static bool once = (hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<128, 128, 3, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess);
        (void)once;
        MVLT_LAUNCH((gemm_tn_dma_kernel<128, 128, 3, 2, true>), grid, block, lds, s, *a, m_per_split, 1, 1, splits, (bf16*)nullptr);
      }
      return mvlt_check_launch("mvlt_gemm_tn");
    }
    MVLT_REQUIRE(!a->c_overwrite || (!a->trans_c && a->c_taps <= 1 && a->N2 % 4 == 0 && a->ldc % 4 == 0 && ((uintptr_t)a->C & 15) == 0),
                 "mvlt_gemm_tn: c_overwrite needs the plain output layout, N2 % 4 == 0, ldc % 4 == 0 and a 16-byte aligned C");
    const bool small_out = a->N1 <= 128 && a->N2 <= 128 && !getenv("MVLT_TN_NO64");
    const int bmt = (a->N1 <= 64 || small_out) ? 64 : 128, bn = (a->N2 <= 64 || small_out) ? 64 : 128;
    const int t1 = (a->N1 + bmt - 1) / bmt, t2 = (a->N2 + bn - 1) / bn;
    int splits = a->c_overwrite ? 1 : a->splits;
    if (splits <= 0) {
      // 2 workgroups per CU in one round: every extra split is another N1*N2 fp32 atomics through the fabric
      splits = ((t1 * t2 == 1 ? 384 : 512) + t1 * t2 - 1) / (t1 * t2);     // a single output tile: 384 (1081344 x 64 x 64: 54.7 -> 48.9 us)
      if (splits >= 8) splits = (splits + 4) / 8 * 8;
      if (splits > 4096) splits = 4096;
      // ... and a split must bring at least 8 k-tiles of work to its N1*N2 atomics: the kv / text-row weight gradients (32768 or 16384
      // rows, 256 x 128 outputs) ran 256 splits of 1-2 k-tiles, 49 us where 32-64 splits take 20-26 us
      static const int min_tiles = getenv("MVLT_TN_MINT") ? atoi(getenv("MVLT_TN_MINT")) : 8;
      if (min_tiles > 1 && splits > mtiles / min_tiles) splits = mtiles / min_tiles;
    }
    if (splits > mtiles) splits = mtiles;
    if (splits < 1) splits = 1;
    int m_per_split = ((mtiles + splits - 1) / splits) * TBK;
    splits = (a->M + m_per_split - 1) / m_per_split;
    const int ns = (bmt + bn == 256) ? 2 : (bmt + bn == 192) ? 3 : 4;      // 64 / 72 / 64 KB of LDS: 2 workgroups per CU
    const size_t lds = (size_t)ns * TBK * (bmt + bn) * 2;
    dim3 grid((unsigned)((splits >= 8 ? 8 * ((splits + 7) / 8) : splits) * t1 * t2)), block(NTHREADS);
    // several splits meeting on an output of >= 65536 elements (>= 8: the C x C and the fc weight gradients of stages 3-4; 24 while every fold was a launch of its own -- with the
    // batched folds 16 / 8 / 4 / 2 all measure 12.98-13.00 k pairs/s against 12.90 at 24): bf16 partial tiles into the caller's scratch
    // + an ordered fold instead of the atomics (14-21 us of a 53-56 us launch, profiles/r05_tn_small_atomics_ablation.txt).  Fewer splits: the atomics are cheaper than a fold launch.
    static const int part_min = getenv("MVLT_TN_PART_MIN") ? atoi(getenv("MVLT_TN_PART_MIN")) : 8;
    // ... and from 16384 output elements on (65536 through round 5): the SMALL outputs are where many splits hurt most -- text_embed1's 64 x 768 weight gradient over 32768 rows:
    // 64 splits x 49152 atomics on the same 1536 cache lines = 56 us where its operands are 9 us of HBM time (tools/ubench_tn_smallk.py; MVLT_TN_PART_MINOUT)
    static const long part_min_out = getenv("MVLT_TN_PART_MINOUT") ? atol(getenv("MVLT_TN_PART_MINOUT")) : 16384;
    bf16* const part = (tnp8 && a->partials && !a->c_overwrite && splits >= part_min && !a->trans_c && a->c_taps <= 1 && a->N2 % 8 == 0 && a->ldc % 4 == 0 && ((uintptr_t)a->C & 15) == 0 &&
                        (long)a->N1 * a->N2 >= part_min_out)
                           ? fold_acquire(*a, (long)splits * a->N1 * a->N2 * 2, s) : nullptr;
#define MVLT_TN_LAUNCH(BMT_, BN_, NS_)                                                                                          \
  do {                                                                                                                         \
    if (a->b_map.mode == 0 && a->b_map.rows_per_batch == 0 && a->a_map.rows_per_batch == 0)                                                        \
      MVLT_LAUNCH((gemm_tn_dma_kernel<BMT_, BN_, 3, NS_>), grid, block, lds, s, *a, m_per_split, t1, t2, splits, part);                     \
    else if (a->b_map.mode == 0) MVLT_LAUNCH((gemm_tn_dma_kernel<BMT_, BN_, 0, NS_>), grid, block, lds, s, *a, m_per_split, t1, t2, splits, part); \
    else if (a->b_map.mode == 1) MVLT_LAUNCH((gemm_tn_dma_kernel<BMT_, BN_, 1, NS_>), grid, block, lds, s, *a, m_per_split, t1, t2, splits, part); \
    else MVLT_LAUNCH((gemm_tn_dma_kernel<BMT_, BN_, 2, NS_>), grid, block, lds, s, *a, m_per_split, t1, t2, splits, part);                   \
  } while (0)
    if (bmt == 128 && bn == 128) MVLT_TN_LAUNCH(128, 128, 2);
    else if (bmt == 128) MVLT_TN_LAUNCH(128, 64, 3);
    else if (bn == 128) MVLT_TN_LAUNCH(64, 128, 3);
    else MVLT_TN_LAUNCH(64, 64, 4);
#undef MVLT_TN_LAUNCH
    if (part) fold_launch(*a, part, splits, s);
    return mvlt_check_launch("mvlt_gemm_tn");
  }
  const bool narrow = a->N2 <= 64;
  const int bn = narrow ? 64 : 128;
  const int t1 = (a->N1 + BM - 1) / BM, t2 = (a->N2 + bn - 1) / bn;
  int splits = a->splits;
  if (splits <= 0) {
    splits = (1024 + t1 * t2 - 1) / (t1 * t2);       // ~4 workgroups per CU in total
    if (splits > mtiles) splits = mtiles;
    if (splits > 4096) splits = 4096;
    if (splits < 1) splits = 1;
  }
  int m_per_split = ((mtiles + splits - 1) / splits) * TBK;
  splits = (a->M + m_per_split - 1) / m_per_split;
  const int rowb = a->dtype == 0 ? TElem<bf16>::ROWB : TElem<float>::ROWB;
  const size_t lds = (size_t)(BM + bn) * rowb + BM * sizeof(float);
  dim3 grid((unsigned)t1, (unsigned)t2, (unsigned)splits), block(NTHREADS);
  if (a->dtype == 0) {
    if (narrow) MVLT_LAUNCH((gemm_tn_kernel<bf16, 64>), grid, block, lds, s, *a, m_per_split);
    else MVLT_LAUNCH((gemm_tn_kernel<bf16, 128>), grid, block, lds, s, *a, m_per_split);
  } else {
    if (narrow) MVLT_LAUNCH((gemm_tn_kernel<float, 64>), grid, block, lds, s, *a, m_per_split);
    else MVLT_LAUNCH((gemm_tn_kernel<float, 128>), grid, block, lds, s, *a, m_per_split);
  }
  return mvlt_check_launch("mvlt_gemm_tn");
}
