// Fused MLP kernels for the narrow stages (C = 64 / 128: stage 1 and 2 of PVLT), gfx950.
//
// Reference libs/pvlt.py:65-71 runs fc1 -> GELU -> fc2 as three kernels and round-trips the (tokens x hidden) activation
// through HBM three times (18.4 MB per pair per block in stage 1).  With C = 64/128 the GEMMs have
// K = C: far below the MFMA/HBM ridge, so the hidden activation is the whole cost.  Here a workgroup owns 128 (C=64)
// or 64 (C=128) token rows, keeps their x tile in LDS, and walks the hidden dimension in chunks of 64 units:
//
//   forward  (mode 0):  G_c = gelu(x W1_c^T + b1_c)           producer MFMAs -> bf16 tile in LDS
//                       out += G_c W2[:, c]^T                 consumer MFMAs (accumulators live across chunks)
//                       out = (out + b2) * droppath + residual            (fp32 residual stream)
//   backward (mode 1):  H_c = x W1_c^T + b1_c ; dG_c = dy W2[:, c]        two producers
//                       dH_c = dG_c * gelu'(H_c)                          -> bf16 tile in LDS
//                       dx += dH_c W1_c                                   consumer; dx *= droppath
//
// so the hidden activation never leaves the CU (forward can optionally still store the pre-activation for the unfused
// backward).  Producers are computed transposed (rows = hidden units, columns = tokens) so that each lane ends up with
// four consecutive hidden units of one token: one 8-byte LDS store into the consumer's A-operand tile.
#include "common.h"
#include "../../include/mvlt_hip.h"
#include <type_traits>

namespace {

// Activation by table (MVLT_GELU_LUT bit 0: the weight-gradient kernels, bit 1: the input-gradient kernels, bit 2: the forward): the
// pre-activation is rounded to bf16 -- what the stage 3-4 path stores and what torch autocast hands nn.GELU -- and { Phi - 1/2, GELU' - 1/2 }
// of its magnitude comes out of a 13 KB LDS copy of g_gelu_lut (tools/gen_gelu_lut.py); the sign is applied by one FMA each: 9.5 VALU
// instructions + one ds_read_b64 per hidden activation in the weight-gradient kernels instead of 17, 7 + one ds_read_b32 instead of 11-12 in the others
// Measured (same box, tools/ubench_mlp2.py / ab_bench_libs.sh): weight gradients 470 -> 363 us (C = 64) and 381 -> 358 us (C = 128), step -0.24 ms;
// input gradients 325 -> 324 / 301 -> 306 us and forward 265 -> 283 / 204 -> 210 us (three to four waves per SIMD already hide the polynomial;
// the gather's LDS latency is not hidden): bits 1 and 2 stay off.
#ifndef MVLT_GELU_LUT
#define MVLT_GELU_LUT 1
#endif
#include "gelu_lut.inc"
constexpr int GELU_LUT_BYTES = MVLT_GELU_LUT_PAD * 8;          // whole 1 KB LDS-DMA instructions
// the table into LDS by LDS-DMA, 1 KB per wave instruction, dealt round-robin to the NWV waves (lands with the kernel's first vmcnt(0) + barrier)
template <int NWV> __device__ __forceinline__ void gelu_lut_dma(unsigned lds_base, int wave, int lane) {
  for (int k = wave; k < GELU_LUT_BYTES / 1024; k += NWV)
    glds16((const char*)&g_gelu_lut[0][0] + k * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_base + k * 1024));
}
// index of a bf16 magnitude (15 bits) as a byte offset from lut0 = table base - 8 * MVLT_GELU_LUT_BASE
__device__ __forceinline__ unsigned gelu_lut_off(unsigned u) {
  return min(max(u, (unsigned)MVLT_GELU_LUT_BASE), (unsigned)MVLT_GELU_LUT_TOP) << 3;
}
__device__ __forceinline__ float gelu_lut_sign(float h) { return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, h) & 0x80000000u) | 0x3f800000u); }
// one of the two functions alone (WHICH = 0: Phi, 1: GELU') of two pre-activations
template <int WHICH> __device__ __forceinline__ void gelu_lut_one2(const char* lut0, float h0, float h1, float& r0, float& r1) {
  const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{h0, h1}, bf16x2));
  const float t0 = *(const float*)(lut0 + gelu_lut_off(pk & 0x7fffu) + 4 * WHICH), t1 = *(const float*)(lut0 + gelu_lut_off((pk >> 16) & 0x7fffu) + 4 * WHICH);
  r0 = __builtin_fmaf(gelu_lut_sign(h0), t0, 0.5f);
  r1 = __builtin_fmaf(gelu_lut_sign(h1), t1, 0.5f);
}
// Phi(h) and GELU'(h) of two pre-activations; lut0 = LDS table base - 8 * MVLT_GELU_LUT_BASE
__device__ __forceinline__ void gelu_lut_both2(const char* lut0, float h0, float h1, float& ph0, float& d0, float& ph1, float& d1) {
  const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{h0, h1}, bf16x2));
  const f32x2 t0 = *(const f32x2*)(lut0 + gelu_lut_off(pk & 0x7fffu)), t1 = *(const f32x2*)(lut0 + gelu_lut_off((pk >> 16) & 0x7fffu));
  const float s0 = gelu_lut_sign(h0), s1 = gelu_lut_sign(h1);
  ph0 = __builtin_fmaf(s0, t0[0], 0.5f); d0 = __builtin_fmaf(s0, t0[1], 0.5f);
  ph1 = __builtin_fmaf(s1, t1[0], 0.5f); d1 = __builtin_fmaf(s1, t1[1], 0.5f);
}

constexpr int NT = 256;
template <int C> struct MlpGeo { static constexpr int JC = (C == 64) ? 64 : 32; };   // hidden units per chunk: C*JC = 4096 MACs per token either way

// element offset inside a [rows][K] bf16 tile with 16-B chunks XOR-swizzled by row (conflict-free fragment reads)
template <int K> __device__ __forceinline__ int toff(int row, int k) {
  constexpr int NCH = K / 8;                 // 16-B chunks per row: 4 (K=32), 8 (K=64) or 16 (K=128)
  int ch = k >> 3;
  int sw = (NCH == 4) ? (ch ^ ((row >> 2) & 3)) : (NCH == 8) ? (ch ^ ((row >> 1) & 7)) : (ch ^ (row & 15));
  return row * K + sw * 8 + (k & 7);
}

__device__ __forceinline__ bf16x8 ldfrag(const bf16* tile_base_elem) { return *(const bf16x8*)tile_base_elem; }

// ---- B-operand fragments of a wave's 32 token rows (x, and dy for the backward): 8 consecutive channels of token fr per lane, straight
// from global memory (used by every hidden chunk, never re-read: no LDS copy); forward: optionally with Block.norm2 folded in
template <int C, int MODE>
__device__ __forceinline__ void mlp_load_rows(const mvlt_mlp_args& p, int m0, int wave, int fr, int fg, bf16x8 (&xfr)[2][C / 32],
                                              bf16x8 (&yfr)[MODE == 1 ? 2 : 1][C / 32]) {
  constexpr int MT = 2, WR = 32, KS_C = C / 32;
  const bf16* X = (const bf16*)p.x;
  const bf16* DY = (const bf16*)p.dy;
  if (MODE == 0 && p.ln_x) {
    // LayerNorm folded into the operand load (Block.norm2): the four fg-lanes of a token hold its whole fp32 row (KS_C x 8 channels
    // each), so the row statistics are two xor-shuffles away; the normalised row is this wave's fc1 operand AND is stored for the
    // backward passes, which replaces a separate LayerNorm launch (fp32 row read + bf16 row written once more)
    const float inv_c = 1.0f / (float)C;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + wave * WR + mt * 16 + fr;
      const bool ok = m < p.M;
      float v[KS_C][8];
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS_C; ++ks) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const float* src = p.ln_x + (long)m * C + ks * 32 + fg * 8;
          a = *(const f32x4*)src; b = *(const f32x4*)(src + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[ks][e] = a[e]; v[ks][4 + e] = b[e]; sum += a[e] + b[e]; }
      }
      sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
      const float mean = sum * inv_c;
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS_C; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mean; q += d * d; }
      q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
      const float rstd = rsqrtf(q * inv_c + p.ln_eps);
      if (ok && fg == 0) { p.ln_mean[m] = mean; p.ln_rstd[m] = rstd; }
#pragma unroll
      for (int ks = 0; ks < KS_C; ++ks) {
        const float* gp = p.ln_gamma + ks * 32 + fg * 8;
        const float* bp = p.ln_beta + ks * 32 + fg * 8;
        const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4), be0 = *(const f32x4*)bp, be1 = *(const f32x4*)(bp + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16)((v[ks][e] - mean) * rstd * g0[e] + be0[e]);
          o[4 + e] = (bf16)((v[ks][4 + e] - mean) * rstd * g1[e] + be1[e]);
        }
        if (!ok) o = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        xfr[mt][ks] = o;
        if (ok) st_g<MVLT_NT_MLP>((bf16x8*)((bf16*)p.ln_y + (long)m * C + ks * 32 + fg * 8), o);
      }
    }
  } else {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ks = 0; ks < KS_C; ++ks) {
      const int m = m0 + wave * WR + mt * 16 + fr;
      u32x4 v = {0u, 0u, 0u, 0u}, w = {0u, 0u, 0u, 0u};
      if (m < p.M) {
        v = *(const u32x4*)(X + (long)m * C + ks * 32 + fg * 8);
        if (MODE == 1) w = *(const u32x4*)(DY + (long)m * C + ks * 32 + fg * 8);
      }
      xfr[mt][ks] = __builtin_bit_cast(bf16x8, v);
      if (MODE == 1) yfr[mt][ks] = __builtin_bit_cast(bf16x8, w);
    }
  }
}

// ---- epilogue shared by the fused kernels: oacc[i][j][r] = out[token wave*32 + 16 i + 4 fg + r][c = 16 j + fr]
template <int C, int MODE, bool PFALL = true>
__device__ __forceinline__ void mlp_epilogue(const mvlt_mlp_args& p, char* smem, int m0, int tid, f32x4 (&oacc)[2][C / 16]) {
  constexpr int MT = 2, WR = 32, CT = C / 16;
  const int lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  // staged per wave through LDS (the operand tiles are dead) so that global traffic is 16-byte, row-contiguous
  // (one 16-token tile at a time: 4 waves x 16 rows x (C+4) floats)
  constexpr int LDW = C + 4, CPR = C / 8, RPI = 64 / CPR;
  float* stage = (float*)smem + wave * 16 * LDW;
  const int ch = lane % CPR;
  const int nc = ch * 8;
  float b2v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) b2v[e] = (MODE == 0) ? p.b2[nc + e] : 0.f;
  // MODE 1 with lnb_x: the backward of the LayerNorm in front of this MLP (Block.norm2) rides on the epilogue -- the row of
  // d(LN output) is in registers here (CPR lanes x 8 channels), so its two row reductions are log2(CPR) xor-shuffles
  const bool lnb = MODE == 1 && p.lnb_x != nullptr;
  float lgam[8], accg[8], accb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { lgam[e] = lnb ? p.lnb_gamma[nc + e] : 0.f; accg[e] = 0.f; accb[e] = 0.f; }
  // Everything the row passes below read from HBM -- the fp32 residual rows (forward), or the LayerNorm input rows, their statistics and
  // the gradient stream's current rows (backward with lnb_x) -- is requested HERE for all of the wave's 32 rows, before the first staging
  // pass: read inside the passes, each of the 2 x NIT row groups paid its own HBM round trip behind the previous group's stores (the
  // epilogue was 52 of 280 us of the stage-1 forward and ~120 of 530 us of the stage-1 input-gradient launch, by ablation)
  constexpr int NIT = 16 / RPI;
  f32x4 pre_a[MT][NIT][2];       // forward: residual row chunk; backward: LayerNorm input row chunk
  u32x4 pre_o[MT][NIT];          // backward: current gradient-stream row chunk (bf16 x 8)
  float pre_mean[MT][NIT], pre_rstd[MT][NIT], pre_rs[MT][NIT], pre_sc[MT][NIT];
  // PFALL = false (three waves per SIMD: the other waves cover the round trip) requests one 16-row tile at a time: half the registers
  auto prefetch = [&](int i) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + wave * WR + i * 16 + it * RPI + lane / CPR;
      const bool ok = m < p.M;
      const long idx = (long)(ok ? m : 0) * C + nc;
      const float* src = (MODE == 0) ? (const float*)p.residual + idx : p.lnb_x + idx;
      pre_a[i][it][0] = ld_g<MVLT_NT_LD && MVLT_NT_MLP>((const f32x4*)src);
      pre_a[i][it][1] = ld_g<MVLT_NT_LD && MVLT_NT_MLP>((const f32x4*)(src + 4));
      pre_rs[i][it] = p.row_scale ? p.row_scale[(ok ? m : 0) / p.rows_per_scale] : 1.0f;
      if (MODE == 1) {
        pre_o[i][it] = ld_g<MVLT_NT_LD && MVLT_NT_MLP>((const u32x4*)((const bf16*)p.lnb_dx + idx));
        pre_mean[i][it] = p.lnb_mean[ok ? m : 0];
        pre_rstd[i][it] = p.lnb_rstd[ok ? m : 0];
        pre_sc[i][it] = p.lnb_dx2 ? p.lnb_dx2_scale[(ok ? m : 0) / p.lnb_dx2_rows_per_scale] : 0.f;
      }
    }
  };
  if (PFALL && (MODE == 0 || lnb)) {
#pragma unroll
    for (int i = 0; i < MT; ++i) prefetch(i);
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (!PFALL && (MODE == 0 || lnb)) prefetch(i);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();               // previous tile's reads are done before it is overwritten
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[(4 * fg + r) * LDW + j * 16 + fr] = oacc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 16 / RPI; ++it) {
      const int rl = it * RPI + lane / CPR;
      const int m = m0 + wave * WR + i * 16 + rl;
      if (m >= p.M) continue;
      f32x4 v0 = *(const f32x4*)(stage + rl * LDW + ch * 8), v1 = *(const f32x4*)(stage + rl * LDW + ch * 8 + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const float rs = (MODE == 0 || lnb) ? pre_rs[i][it] : (p.row_scale ? p.row_scale[m / p.rows_per_scale] : 1.0f);
      const long idx = (long)m * C + nc;
      if (MODE == 0) {
        const f32x4 r0 = pre_a[i][it][0], r1 = pre_a[i][it][1];
        float rr[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (v[e] + b2v[e]) * rs + rr[e];
        if (p.out) {
          float* O = (float*)p.out + idx;
          st_g<MVLT_NT_MLP>((f32x4*)O, f32x4{v[0], v[1], v[2], v[3]});
          st_g<MVLT_NT_MLP>((f32x4*)(O + 4), f32x4{v[4], v[5], v[6], v[7]});
        }
        if (p.out_op) {                  // MFMA-operand copy of the block output (the last block of a stage writes only this one)
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
          st_g<MVLT_NT_MLP>((bf16x8*)((bf16*)p.out_op + idx), o);
        }
        if (p.post_y) {
          // LayerNorm of the OUTPUT row (the next block's norm1, reference libs/pvlt.py:141) while the row is here: its CPR lanes hold
          // 8 channels each, so the two-pass statistics are log2(CPR) xor-shuffles; saves that block's LayerNorm launch (fp32 row read)
          float sum = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += v[e];
#pragma unroll
          for (int o = 1; o < CPR; o <<= 1) sum += __shfl_xor(sum, o);
          const float mean = sum * (1.0f / (float)C);
          float q = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
#pragma unroll
          for (int o = 1; o < CPR; o <<= 1) q += __shfl_xor(q, o);
          const float rstd = rsqrtf(q * (1.0f / (float)C) + p.post_eps);
          if (ch == 0) { p.post_mean[m] = mean; p.post_rstd[m] = rstd; }
          const f32x4 g0 = *(const f32x4*)(p.post_gamma + nc), g1 = *(const f32x4*)(p.post_gamma + nc + 4);
          const f32x4 be0 = *(const f32x4*)(p.post_beta + nc), be1 = *(const f32x4*)(p.post_beta + nc + 4);
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (bf16)((v[e] - mean) * rstd * g0[e] + be0[e]);
            o[4 + e] = (bf16)((v[4 + e] - mean) * rstd * g1[e] + be1[e]);
          }
          st_g<MVLT_NT_MLP>((bf16x8*)((bf16*)p.post_y + idx), o);
        }
      } else if (!lnb) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)(v[e] * rs);
        st_g<MVLT_NT_MLP>((bf16x8*)((bf16*)p.out + idx), o);
      } else {
        // dx (+)= LayerNorm backward of d(LN output) = v * rs; optional second output dx2 = dx * DropPath factor of the other branch
        const f32x4 x0 = pre_a[i][it][0], x1 = pre_a[i][it][1];
        const bf16x8 old = __builtin_bit_cast(bf16x8, pre_o[i][it]);
        const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        const float mean = pre_mean[i][it], rstd = pre_rstd[i][it];
        float g[8], xh[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dyn = v[e] * rs;
          xh[e] = (xs[e] - mean) * rstd;
          g[e] = dyn * lgam[e];
          s1 += g[e]; s2 += g[e] * xh[e];
          accg[e] += dyn * xh[e]; accb[e] += dyn;
        }
#pragma unroll
        for (int o = 1; o < CPR; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        s1 *= 1.0f / (float)C; s2 *= 1.0f / (float)C;
        float dxv[8];
        bf16x8 o1;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dxv[e] = rstd * (g[e] - s1 - xh[e] * s2) + (float)old[e]; o1[e] = (bf16)dxv[e]; }
        st_g<MVLT_NT_MLP>((bf16x8*)((bf16*)p.lnb_dx + idx), o1);
        if (p.lnb_dx2) {
          const float sc = pre_sc[i][it];
          bf16x8 o2;
#pragma unroll
          for (int e = 0; e < 8; ++e) o2[e] = (bf16)(dxv[e] * sc);
          st_g<MVLT_NT_MLP>((bf16x8*)((bf16*)p.lnb_dx2 + idx), o2);
        }
      }
    }
  }
  if (MODE == 1 && lnb) {
    // column sums of this workgroup's rows (d gamma | d beta): lanes with the same channel chunk, then the four waves through LDS,
    // stored plainly per workgroup; mvlt_add_column_sums adds them to the gradient (no same-address atomics from 8000 workgroups)
#pragma unroll
    for (int o = CPR; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) { accg[e] += __shfl_xor(accg[e], o); accb[e] += __shfl_xor(accb[e], o); }
    __syncthreads();                                  // every wave is done with its staging rows
    float* part = (float*)smem;                      // [4 waves][2 C]
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { part[wave * 2 * C + nc + e] = accg[e]; part[wave * 2 * C + C + nc + e] = accb[e]; }
    }
    __syncthreads();
    for (int c = tid; c < 2 * C; c += NT)
      p.lnb_partials[(long)blockIdx.x * 2 * C + c] = part[c] + part[2 * C + c] + part[4 * C + c] + part[6 * C + c];
  }
}

template <int C, int MODE>
__global__ __launch_bounds__(NT) void mlp_fused_kernel(mvlt_mlp_args p) {
  constexpr int JC = MlpGeo<C>::JC;
  constexpr int BM = 128;                    // token rows per workgroup
  constexpr int WR = BM / 4;                 // token rows per wave: each wave is self-contained (producer AND consumer of its rows)
  constexpr int MT = WR / 16;                // 16-token tiles per wave
  constexpr int CT = C / 16;                 // 16-col output tiles (all of C)
  constexpr int KS_C = C / 32;               // k32 steps over C
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sWa = (bf16*)smem;                                   // [2][JC][C]   W1 chunk
  bf16* sWb = sWa + 2 * JC * C;                              // [2][C][JC]   W2[:, chunk] (mode 0) / W1^T[:, chunk] (mode 1)
  bf16* sWc = sWb + 2 * C * JC;                              // [2][JC][C]   W2^T chunk   (mode 1 only)
  float* sB1 = (float*)(sWc + (MODE == 1 ? 2 * JC * C : 0)); // [hid]        fc1 bias: a global load inside the chunk loop would
                                                             //              wait (vmcnt is in-order) for the weight prefetch too
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * BM;
  const bf16* X = (const bf16*)p.x;
  const bf16* DY = (const bf16*)p.dy;
  const bf16* Wa = (const bf16*)p.w1;        // [hid][C]
  const bf16* Wb = (const bf16*)p.wb;        // [C][hid]
  const bf16* Wc = (const bf16*)p.wc;        // [hid][C]
  const int hid = p.hid;

  for (int u = tid; u < p.hid; u += NT) sB1[u] = p.b1[u];
  constexpr bool LUT = MODE == 1 && ((MVLT_GELU_LUT >> 1) & 1);     // GELU' by table (the launch sizes the LDS for it): visible after the first barrier below
  float* const sLut = sB1 + hid;
  const char* const lut0 = (const char*)sLut - 8 * MVLT_GELU_LUT_BASE;
  if (LUT)
    for (int u = tid; u < MVLT_GELU_LUT_N * 2; u += NT) sLut[u] = (&g_gelu_lut[0][0])[u];
  // ---- B-operand fragments of this wave's tokens (x, and dy for the backward): 8 consecutive channels of token fr per
  // lane, straight from global memory (used by every hidden chunk, never re-read: no LDS copy)
  bf16x8 xfr[MT][KS_C], yfr[MODE == 1 ? MT : 1][KS_C];
  mlp_load_rows<C, MODE>(p, m0, wave, fr, fg, xfr, yfr);
  // ---- weight-chunk staging (global -> registers -> LDS), double buffered
  constexpr int WA_IT = JC * (C / 8) / NT;   // 2 (C=64) / 4 (C=128)
  constexpr int WB_IT = C * (JC / 8) / NT;   // 2 / 4
  u32x4 ra[WA_IT], rb[WB_IT], rc[MODE == 1 ? WA_IT : 1];
  auto wload = [&](int jc) {
#pragma unroll
    for (int it = 0; it < WA_IT; ++it) {
      int u = tid + it * NT, r = u / (C / 8), ch = u % (C / 8);
      ra[it] = *(const u32x4*)(Wa + (long)(jc * JC + r) * C + ch * 8);
      if (MODE == 1) rc[it] = *(const u32x4*)(Wc + (long)(jc * JC + r) * C + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < WB_IT; ++it) {
      int u = tid + it * NT, r = u / (JC / 8), ch = u % (JC / 8);
      rb[it] = *(const u32x4*)(Wb + (long)r * hid + jc * JC + ch * 8);
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int it = 0; it < WA_IT; ++it) {
      int u = tid + it * NT, r = u / (C / 8), ch = u % (C / 8);
      *(u32x4*)(sWa + buf * JC * C + toff<C>(r, ch * 8)) = ra[it];
      if (MODE == 1) *(u32x4*)(sWc + buf * JC * C + toff<C>(r, ch * 8)) = rc[it];
    }
#pragma unroll
    for (int it = 0; it < WB_IT; ++it) {
      int u = tid + it * NT, r = u / (JC / 8), ch = u % (JC / 8);
      *(u32x4*)(sWb + buf * C * JC + toff<JC>(r, ch * 8)) = rb[it];
    }
  };
  wload(0);
  wstore(0);
  __syncthreads();

  f32x4 oacc[MT][CT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = hid / JC;
  for (int jc = 0; jc < nchunks; ++jc) {
    const int buf = jc & 1;
    if (jc + 1 < nchunks) wload(jc + 1);
    // ---- producers, transposed: tile (jt, mt): rows = hidden units 16 jt.. of the chunk, cols = this wave's tokens 16 mt..
    // A lane ends up with hidden units 16 jt + 4 fg + r of token fr.  Two hidden tiles (jt = 2 pair, 2 pair + 1) give it 8
    // values of one token = one A-operand fragment of the consumer MFMA, with k-slot (fg, jj) <-> hidden unit
    // 32 pair + 16 (jj >> 2) + 4 fg + (jj & 3); the consumer's B operand is read from the W chunk in the same order
    // (two 8-byte LDS reads), so the activation goes from accumulator to operand without touching LDS.
    const bf16* wa = sWa + buf * JC * C;
    const bf16* wc = sWc + buf * JC * C;
    bf16x8 gfrag[JC / 32][MT];
#pragma unroll
    for (int jt = 0; jt < JC / 16; ++jt) {
      const int jrow = jt * 16 + fr;                   // A-operand row (hidden unit inside the chunk)
      bf16x8 af[KS_C], cf[MODE == 1 ? KS_C : 1];
#pragma unroll
      for (int ks = 0; ks < KS_C; ++ks) {
        af[ks] = ldfrag(wa + toff<C>(jrow, ks * 32 + fg * 8));
        if (MODE == 1) cf[ks] = ldfrag(wc + toff<C>(jrow, ks * 32 + fg * 8));
      }
      const int jl = jt * 16 + 4 * fg;                 // the four hidden units this lane ends up with: jl + r
      f32x4 b1v = *(const f32x4*)(sB1 + jc * JC + jl);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int mrow = wave * WR + mt * 16 + fr;     // token (column fr of the tile)
        f32x4 h = {0.f, 0.f, 0.f, 0.f}, dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS_C; ++ks) {
          h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], xfr[mt][ks], h, 0, 0, 0);
          if (MODE == 1) dg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[ks], yfr[mt][ks], dg, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 hv = f32x2{h[r], h[r + 1]} + f32x2{b1v[r], b1v[r + 1]};
          f32x2 gv;
          if (LUT) {
            float d0, d1;
            gelu_lut_one2<1>(lut0, hv[0], hv[1], d0, d1);
            gv = f32x2{dg[r] * d0, dg[r + 1] * d1};
          } else gv = (MODE == 0) ? gelu_fast2(hv) : f32x2{dg[r], dg[r + 1]} * gelu_fast_grad2(hv);
          gfrag[jt >> 1][mt][(jt & 1) * 4 + r] = (bf16)gv[0];
          gfrag[jt >> 1][mt][(jt & 1) * 4 + r + 1] = (bf16)gv[1];
          h[r] = hv[0]; h[r + 1] = hv[1];
        }
        if (MODE == 0 && p.h_out && m0 + mrow < p.M) {
          bf16x4 h4 = {(bf16)h[0], (bf16)h[1], (bf16)h[2], (bf16)h[3]};
          *(bf16x4*)((bf16*)p.h_out + (long)(m0 + mrow) * hid + jc * JC + jl) = h4;
        }
      }
    }
    // ---- consumer: out[this wave's tokens][C] += G[tokens][64] . Wb[C][64]^T
    const bf16* wb = sWb + buf * C * JC;
#pragma unroll
    for (int pair = 0; pair < JC / 32; ++pair) {
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const bf16x4 b0 = *(const bf16x4*)(wb + toff<JC>(j * 16 + fr, pair * 32 + 4 * fg));
        const bf16x4 b1 = *(const bf16x4*)(wb + toff<JC>(j * 16 + fr, pair * 32 + 16 + 4 * fg));
        const bf16x8 bfr = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int i = 0; i < MT; ++i) oacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfrag[pair][i], bfr, oacc[i][j], 0, 0, 0);
      }
    }
    if (jc + 1 < nchunks) wstore(buf ^ 1);
    __syncthreads();                                   // weight double buffer: next chunk visible, this one free next time
  }

  mlp_epilogue<C, MODE>(p, smem, m0, tid, oacc);
}

// ------------------------------------------------------------------------------------------------ software-pipelined fused MLP
// Round 3.  The kernel above runs, per wave and hidden chunk, 16 producer MFMAs, then ~270 VALU instructions of GELU, then 16 consumer
// MFMAs -- strictly one after the other (ISA: tools/isa_mix.py), so the matrix pipe idles through every activation block and the VALU
// through every MFMA block unless ANOTHER wave of the SIMD happens to be in the opposite phase (counters: MFMA busy 0.20-0.28 of SIMD
// cycles, VALU active ~0.25 per wave).  Here the three stages of a 32-hidden-unit slice u are spread over three loop iterations of ONE
// wave -- iteration i issues the producer MFMAs of slice i+1, the activation of slice i and the consumer MFMAs of slice i-1, which are
// independent instruction streams the scheduler may interleave (MFMAs issue asynchronously: the VALU work of a wave runs in the shadow
// of its own matrix instructions):
//
//     P(i+1): H = W1[slice] x^T + b1      (+ dG = W2^T[slice] dy^T)     -> hacc[(i+1)&1]      [MFMA]
//     G(i)  : g = gelu(H)                 (dH = dG * gelu'(H))          -> gfrag[i&1] (bf16)  [VALU]
//     C(i-1): out += g W2[:, slice]^T     (dx += dH W1^T[:, slice])                            [MFMA]
//
// Producer tiles are transposed (rows = hidden units, columns = tokens) and the hidden unit of tile row rho is chosen as
// 32 u + 8 (rho >> 2) + 4 h + (rho & 3) for the two tiles h = 0, 1 of a slice: lane (fr, fg) then ends up with the EIGHT CONSECUTIVE
// hidden units 32 u + 8 fg .. + 7 of token fr = one A-operand fragment of the consumer MFMA in natural k order, and the consumer's B
// fragment is ONE 16-byte LDS read (the kernel above pairs tiles 16 apart: two 8-byte reads that hipcc fuses into ds_read2st64_b64,
// which is 2-way bank conflicted: lds_conflict_frac 0.25-0.31).  fc1's bias is the producer's accumulator initialiser (no VALU add).
// Weight slices arrive by LDS-DMA (no staging registers, no ds_write), two small rings of two slots: the producer side (W1 [, W2^T]
// rows of slice i+2) and the consumer side (Wb columns of slice i) are fetched in iteration i, one barrier per iteration.
template <int C> __device__ __forceinline__ int pipe_fP(int rho) {       // 16-B slot swizzle of a producer-tile row (row = hidden unit in slice)
  return C == 64 ? (((rho >> 1) & 1) | (((rho >> 3) & 3) << 1)) : ((rho & 3) | (((rho >> 3) & 3) << 2));
}
__device__ __forceinline__ int pipe_gC(int n) { return (0 - (n >> 2)) & 3; }   // same for a consumer-tile row (row = channel, 64-B rows)

__device__ __forceinline__ float gelu_fast1(float x) {
#if MVLT_GELU_POLY & 2
  return gelu_poly1(x);
#endif
  const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
  const float x2 = xc * xc;
  const float t = xc * __builtin_fmaf(__builtin_fmaf(x2, -MVLT_LOG2E * MVLT_GP2, -MVLT_LOG2E * MVLT_GP1), x2, -MVLT_LOG2E * MVLT_GP0);
  const float e = __builtin_amdgcn_exp2f(t);
  return x * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float gelu_fast_grad1(float x) {
#if MVLT_GELU_POLY & 1
  return gelu_poly_grad1(x);
#endif
  const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
  const float x2 = xc * xc;
  const float t = xc * __builtin_fmaf(__builtin_fmaf(x2, -MVLT_LOG2E * MVLT_GP2, -MVLT_LOG2E * MVLT_GP1), x2, -MVLT_LOG2E * MVLT_GP0);
  const float e = __builtin_amdgcn_exp2f(t);
  const float sg = __builtin_amdgcn_rcpf(e + 1.0f);
  const float up = __builtin_fmaf(__builtin_fmaf(x2, 5.0f * MVLT_GP2, 3.0f * MVLT_GP1), x2, MVLT_GP0);
  return __builtin_fmaf(x * up, sg * sg * e, sg);
}

// Waves per SIMD the register allocation is held to.  A wave of these kernels issues one VALU instruction per ~3.5 ns however few waves share
// its SIMD (tools/probes/valu_rates.hip: 16 x {MFMA + 8 v_fma} takes 38.8 / 27.5 / 21.5 ticks per SIMD at 2 / 3 / 4 waves), so occupancy is
// throughput here.  The C = 64 input-gradient kernel needs 171 registers with the polynomial GELU' (174 with the sigmoid form); held to 168
// = THREE waves per SIMD it spills two dwords outside the slice loop (8 B of scratch per lane, none in the loop) and its epilogue requests
// the LayerNorm-backward operands one 16-row tile at a time (PFALL = false in mlp_epilogue: half the registers; the third wave covers the
// round trip): bare launch 359 -> 326 us, step -0.1 ms (same-box A/B).  The sigmoid form at three waves with the all-rows prefetch spilled
// inside the epilogue: +0.5 ms per step.  Four waves (128 registers) spill 328 B per lane: 1778 us.  C = 128 forward at three waves: 260 B
// of scratch, 213 -> 390 us.
#ifndef MVLT_PIPE_WAVES_64_1
#define MVLT_PIPE_WAVES_64_1 3
#endif
template <int C, int MODE>
__global__ __launch_bounds__(NT, (C == 64 && MODE == 1) ? MVLT_PIPE_WAVES_64_1 : 2) void mlp_pipe_kernel(mvlt_mlp_args p) {
  constexpr int MT = 2, CT = C / 16, KS_C = C / 32;
  constexpr int NP = MODE == 1 ? 2 : 1;          // producer-side tiles per slice: W1 (, W2^T)
  constexpr int RB = 2 * C;                      // bytes per producer-tile row
  constexpr int SPR = RB / 16;                   // 16-B slots per row: 8 / 16
  constexpr int RPI = 64 / SPR;                  // rows per DMA instruction: 8 / 4
  constexpr int IPW = 32 / RPI / 4;              // DMA instructions per wave per producer tile: 1 / 2
  constexpr int PB = 32 * RB;                    // producer tile bytes: 4 / 8 KB
  constexpr int CB = C * 64;                     // consumer tile [C][32] bytes: 4 / 8 KB
  constexpr int IPWC = C / 16 / 4;               // DMA instructions per wave per consumer tile: 1 / 2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2][NP][PB] producer ring | [2][CB] consumer ring | b1 [hid] floats
  char* const sP = smem;
  char* const sC = smem + 2 * NP * PB;
  float* const sB1 = (float*)(sC + 2 * CB);
  constexpr bool LUT = (MVLT_GELU_LUT >> (MODE == 1 ? 1 : 2)) & 1;
  const unsigned sP_lds = (unsigned)(uintptr_t)sP, sC_lds = (unsigned)(uintptr_t)sC;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * 128;
  const int hid = p.hid, NU = hid / 32;

  // ---- DMA geometry: per-lane source pointers of slice 0, advanced by a wave-uniform slice offset
  const char* srcP[NP][IPW];
  const char* srcC[IPWC];
#pragma unroll
  for (int it = 0; it < IPW; ++it) {
    const int q = wave + 4 * it, row = q * RPI + lane / SPR, slot = lane % SPR;
    const long off = (long)row * RB + ((slot ^ pipe_fP<C>(row)) << 4);
    srcP[0][it] = (const char*)p.w1 + off;
    if (MODE == 1) srcP[NP - 1][it] = (const char*)p.wc + off;
  }
#pragma unroll
  for (int it = 0; it < IPWC; ++it) {
    const int q = wave + 4 * it, n = q * 16 + (lane >> 2), slot = lane & 3;
    srcC[it] = (const char*)p.wb + (long)n * hid * 2 + ((slot ^ pipe_gC(n)) << 4);
  }
  auto dmaP = [&](int u, int ring) {             // producer tile(s) of slice u -> ring slot
    const long so = (long)u * PB;
#pragma unroll
    for (int t = 0; t < NP; ++t)
#pragma unroll
      for (int it = 0; it < IPW; ++it)
        glds16(srcP[t][it] + so, __builtin_amdgcn_readfirstlane(sP_lds + (ring * NP + t) * PB + (wave + 4 * it) * 1024));
  };
  auto dmaC = [&](int u, int ring) {
#pragma unroll
    for (int it = 0; it < IPWC; ++it)
      glds16(srcC[it] + u * 64, __builtin_amdgcn_readfirstlane(sC_lds + ring * CB + (wave + 4 * it) * 1024));
  };
  dmaP(0, 0);
  if (NU > 1) dmaP(1, 1);
  const char* const lut0 = (const char*)(sB1 + hid) - 8 * MVLT_GELU_LUT_BASE;      // activation table behind the bias (the launch sizes the LDS for it)
  if (LUT) gelu_lut_dma<4>((unsigned)(uintptr_t)(sB1 + hid), wave, lane);
  for (int u = tid; u < hid; u += NT) sB1[u] = p.b1[u];

  bf16x8 xfr[MT][KS_C], yfr[MODE == 1 ? MT : 1][KS_C];
  mlp_load_rows<C, MODE>(p, m0, wave, fr, fg, xfr, yfr);

  // ---- fragment geometry (byte offsets inside a tile)
  const int rho0 = 8 * (fr >> 2) + (fr & 3);                         // tile h: row rho0 + 4 h
  int poff[KS_C];
#pragma unroll
  for (int ks = 0; ks < KS_C; ++ks) poff[ks] = rho0 * RB + (((ks * 4 + fg) ^ pipe_fP<C>(rho0)) << 4);
  const int coff = fr * 64 + ((fg ^ pipe_gC(fr)) << 4);             // tile j: + j * 16 * 64

  f32x4 oacc[MT][CT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 hacc[2][2][MT], dgacc[MODE == 1 ? 2 : 1][2][MT];
  u32x4 gfrag[2][MT];                            // bf16 pairs: word 2 h + (r >> 1) of token tile mt = hidden units 8 fg + 4 h + r

  // One iteration = NSLOT slots of {a few MFMAs, two activations per lane}, pinned apart by sched_barrier(0): hipcc otherwise emits an
  // iteration's MFMAs as one block and its ~150 VALU instructions as another (checked in the ISA; sched_group_barrier requests did
  // not change that), which is exactly the serialisation this kernel exists to remove.  Inside a slot the MFMAs come first, so the
  // activation's VALU / transcendental instructions issue while the matrix pipe works.
  //   PAR = parity of the slice being ACTIVATED (i): hacc[PAR] -> gfrag[PAR]; producers write hacc[PAR ^ 1] (slice i + 1) from producer
  //   ring slot PAR ^ 1; consumers read gfrag[PAR ^ 1] (slice i - 1) and consumer ring slot PAR ^ 1.
  constexpr int NSLOT = 8;
  constexpr int PK = KS_C / 2;                   // producer k-steps per slot (a tile takes two slots)
  constexpr int CPS = CT * MT / NSLOT;           // consumer MFMAs per slot: 1 / 2
  auto iteration = [&](auto parc, auto dopc, auto docc, auto dogc, int u_next) {
    constexpr int PAR = decltype(parc)::value;
    constexpr bool DO_P = decltype(dopc)::value, DO_C = decltype(docc)::value, DO_G = decltype(dogc)::value;
    const char* tp = sP + (PAR ^ 1) * NP * PB;
    const char* tc = sC + (PAR ^ 1) * CB;
    // operand fragments of slot sl are read one slot ahead into set sl & 1 (only slot 0's reads wait for LDS in front of their MFMAs)
    bf16x8 af[2][PK], cf[MODE == 1 ? 2 : 1][PK], bfr[2][CPS];
    f32x4 b1v;
    auto read_frags = [&](auto slc) {
      constexpr int sl = decltype(slc)::value, t = sl >> 1, half = sl & 1, h = t >> 1;
      if (DO_P) {
#pragma unroll
        for (int k = 0; k < PK; ++k) {
          af[sl & 1][k] = *(const bf16x8*)(tp + h * 4 * RB + poff[half * PK + k]);
          if (MODE == 1) cf[sl & 1][k] = *(const bf16x8*)(tp + PB + h * 4 * RB + poff[half * PK + k]);
        }
        if ((sl & 3) == 0) b1v = *(const f32x4*)(sB1 + u_next * 32 + 8 * fg + 4 * h);       // first slot of the two tiles sharing h
      }
      if (DO_C) {
#pragma unroll
        for (int k = 0; k < CPS; ++k) bfr[sl & 1][k] = *(const bf16x8*)(tc + ((sl * CPS + k) / MT) * 16 * 64 + coff);
      }
    };
    read_frags(std::integral_constant<int, 0>{});
    auto slot = [&](auto slc) {
      constexpr int sl = decltype(slc)::value;
      constexpr int t = sl >> 1, half = sl & 1;  // producer tile (h, mt) = (t >> 1, t & 1) and activation group: same order
      constexpr int h = t >> 1, mt = t & 1;
      if constexpr (sl + 1 < NSLOT) read_frags(std::integral_constant<int, sl + 1>{});
      if (DO_P) {
        f32x4 hh = half == 0 ? b1v : hacc[PAR ^ 1][h][mt];
        f32x4 dg = half == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : dgacc[MODE == 1 ? (PAR ^ 1) : 0][h][mt];
#pragma unroll
        for (int k = 0; k < PK; ++k) {
          hh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[sl & 1][k], xfr[mt][half * PK + k], hh, 0, 0, 0);
          if (MODE == 1) dg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf[sl & 1][k], yfr[mt][half * PK + k], dg, 0, 0, 0);
        }
        hacc[PAR ^ 1][h][mt] = hh;
        if (MODE == 1) dgacc[PAR ^ 1][h][mt] = dg;
      }
      if (DO_C) {
#pragma unroll
        for (int k = 0; k < CPS; ++k) {
          constexpr int c0 = sl * CPS;
          const int c = c0 + k, j = c / MT, i = c % MT;
          oacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, gfrag[PAR ^ 1][i]), bfr[sl & 1][k], oacc[i][j], 0, 0, 0);
        }
      }
      if (DO_G) {
        const float h0 = hacc[PAR][h][mt][2 * half], h1 = hacc[PAR][h][mt][2 * half + 1];
        float g0, g1;
        if (LUT) {
          float a0, a1;
          gelu_lut_one2<MODE>(lut0, h0, h1, a0, a1);
          g0 = (MODE == 0 ? h0 : dgacc[MODE == 1 ? PAR : 0][h][mt][2 * half]) * a0;
          g1 = (MODE == 0 ? h1 : dgacc[MODE == 1 ? PAR : 0][h][mt][2 * half + 1]) * a1;
        } else if (MODE == 0) { g0 = gelu_fast1(h0); g1 = gelu_fast1(h1); }
        else {
          g0 = dgacc[MODE == 1 ? PAR : 0][h][mt][2 * half] * gelu_fast_grad1(h0);
          g1 = dgacc[MODE == 1 ? PAR : 0][h][mt][2 * half + 1] * gelu_fast_grad1(h1);
        }
        gfrag[PAR][mt][2 * h + half] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{g0, g1}, bf16x2));
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    slot(std::integral_constant<int, 0>{}); slot(std::integral_constant<int, 1>{}); slot(std::integral_constant<int, 2>{});
    slot(std::integral_constant<int, 3>{}); slot(std::integral_constant<int, 4>{}); slot(std::integral_constant<int, 5>{});
    slot(std::integral_constant<int, 6>{}); slot(std::integral_constant<int, 7>{});
  };
  auto fence = [&]() {                           // this iteration's DMAs have landed for every wave; its LDS reads are done
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using Y = std::true_type;
  using N = std::false_type;

  fence();                                       // slices 0, 1 of the producer side and the bias are in LDS
  iteration(P1{}, Y{}, N{}, N{}, 0);             // "iteration -1": producers of slice 0 (ring slot 0) -> hacc[0]
  fence();                                       // ring slot 0 may be refilled
  // iteration 0: P(1), G(0)
  dmaP(min(2, NU - 1), 0);
  dmaC(0, 0);
  iteration(P0{}, Y{}, N{}, Y{}, 1);
  fence();
  // steady state, two iterations per trip: i odd, then i + 1 even (NU is even and >= 4: hid % 64 == 0, hid >= 128)
  for (int i = 1; i + 1 < NU - 1; i += 2) {
    dmaP(min(i + 2, NU - 1), 1);
    dmaC(i, 1);
    iteration(P1{}, Y{}, Y{}, Y{}, i + 1);
    fence();
    dmaP(min(i + 3, NU - 1), 0);
    dmaC(i + 1, 0);
    iteration(P0{}, Y{}, Y{}, Y{}, i + 2);
    fence();
  }
  // last iteration i = NU - 1 (odd): G(NU-1), C(NU-2); then the consumers of the last slice alone
  dmaC(NU - 1, 1);
  iteration(P1{}, N{}, Y{}, Y{}, 0);
  fence();
  iteration(P0{}, N{}, Y{}, N{}, 0);
  __syncthreads();                               // the epilogue stages through the same LDS
  mlp_epilogue<C, MODE, !(C == 64 && MODE == 1 && MVLT_PIPE_WAVES_64_1 >= 3)>(p, smem, m0, tid, oacc);
}

// ------------------------------------------------------------------------------------------------ weight gradients
// dW1[j][c] += sum_m dh[m][j] x[m][c]      db1[j] += sum_m dh[m][j]
// dW2[c][j] += sum_m dy[m][c] g[m][j]      db2[c] += sum_m dy[m][c]          (dy scaled by the per-sample DropPath factor)
// A workgroup owns 128 hidden units and a range of tokens; W1 / W2^T rows of its hidden range stay in LDS.  Per 64-token
// tile it recomputes h and dg for its hidden units (MFMA, rows = tokens), turns them into g and dh in registers and
// feeds those straight back as MFMA operands of the two weight-gradient products (k = tokens): the C layout (4
// consecutive tokens per lane) IS the operand layout once two 16-token tiles are paired.  The other operands, x^T / dy^T
// fragments, are transposed reads (ds_read_b64_tr_b16) of the same natural-layout [token][C] tiles the producers read
// row-wise; those tiles arrive by LDS-DMA in a 2-deep ring (the kernel runs one or two waves per SIMD: without the
// prefetch every tile paid a full HBM/L2 round trip).  Nothing of size (tokens x hidden) is ever written.
//
// Tile layout: row = token, 16-B chunks XOR-swizzled by hs(token) << 1 on the DMA source side; hs is chosen so that
// both access patterns are bank-conflict free: the b128 row reads (16 tokens x 4 chunks per instruction) and the
// transposed reads (8 consecutive tokens x one 32-B window per 32-lane group).
template <int C> __device__ __forceinline__ int wg_hs(int row) { return C == 64 ? ((row >> 1) & 3) : (row & 7); }

__device__ __forceinline__ float sload_f32(const float* ptr) {      // scalar load: must not touch vmcnt (the DMA ring counts it)
  float v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
  return v;
}

template <int C>
__global__ __launch_bounds__(NT, C == 64 ? 2 : 1) void mlp_wgrad_kernel(mvlt_mlp_args p, int m_per_split, int splits, int ny) {
  constexpr int KS_C = C / 32, CT16 = C / 16;
  constexpr int ROWB = 2 * C;                  // bytes per token row
  constexpr int CH = C / 8;                    // 16-B slots per row
  constexpr int RPP = NT / CH;                 // rows per 256-thread pass
  constexpr int IT = 64 / RPP;                 // passes per tile
  constexpr int TILE = 64 * ROWB;
  constexpr int STAGE = 2 * TILE;              // x tile | dy tile
  constexpr int LPT = 2 * IT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sW1 = (bf16*)smem;                     // [128][C]
  bf16* sW2T = sW1 + 128 * C;                  // [128][C]
  char* sT = smem + 2 * 128 * C * 2;           // [2][x tile | dy tile]
  const unsigned sT_lds = (unsigned)(uintptr_t)sT;

  // one token split (all its ny hidden blocks) per XCD: x / dy rows come through that L2 once
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int zq = kq / ny, by = kq - zq * ny;
  const int bz = zq * 8 + xcd;
  if (bz >= splits) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int j0 = by * 128;
  const int m_begin = bz * m_per_split, m_end = min(p.M, m_begin + m_per_split);

  // ---- DMA geometry
  const int l_row0 = tid / CH;
  const int l_chunk = (tid % CH) ^ (wg_hs<C>(l_row0) << 1);
  const char* xsrc = (const char*)p.x + l_chunk * 16;
  const char* ysrc = (const char*)p.dy + l_chunk * 16;
  const char* zsrc = (const char*)g_zero_page + ((tid * 16 + (blockIdx.x & 15) * 4096) & 65535);
  auto issue = [&](int mt, int slot) {
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int m = mt + j * RPP + l_row0;
      const bool ok = m < m_end;
      const unsigned long long off = (unsigned long long)(unsigned)m * ROWB;
      const unsigned dst = __builtin_amdgcn_readfirstlane(sT_lds + slot * STAGE + (j * NT + wave * 64) * 16);
      glds16(ok ? xsrc + off : zsrc, dst);
      glds16(ok ? ysrc + off : zsrc, dst + TILE);
    }
  };
  issue(m_begin, 0);

  for (int u = tid; u < 128 * (C / 8); u += NT) {
    int r = u / (C / 8), ch = u % (C / 8);
    *(u32x4*)(sW1 + toff<C>(r, ch * 8)) = *(const u32x4*)((const bf16*)p.w1 + (long)(j0 + r) * C + ch * 8);
    *(u32x4*)(sW2T + toff<C>(r, ch * 8)) = *(const u32x4*)((const bf16*)p.wc + (long)(j0 + r) * C + ch * 8);
  }
  float b1v[2];
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) b1v[jt] = p.b1[j0 + wave * 32 + jt * 16 + fr];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                             // weights visible (and tile 0 has landed)

  // ---- fragment geometry
  const int hs_row = wg_hs<C>(fr);                                   // producers: token row 16 mt + fr
  int xoff[KS_C];
#pragma unroll
  for (int ks = 0; ks < KS_C; ++ks) xoff[ks] = fr * ROWB + (((ks * 4 + fg) ^ (hs_row << 1)) << 4);
  const int L = lane & 15;
  const int trow = 4 * fg + (L >> 2);                                // transposed reads: token row 32 pair + trow (+16)
  const int hs_t = wg_hs<C>(trow);
  int toffs[CT16];
#pragma unroll
  for (int ct = 0; ct < CT16; ++ct) toffs[ct] = trow * ROWB + ((ct ^ hs_t) << 5) + ((L & 3) << 3);

  f32x4 dw1[2][CT16], dw2[CT16][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < CT16; ++b) { dw1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; dw2[b][a] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float db1acc[2] = {0.f, 0.f};
  float db2acc[CT16 / 4];
#pragma unroll
  for (int b = 0; b < CT16 / 4; ++b) db2acc[b] = 0.f;
  const bool do_db2 = by == 0;                 // column sums of dy: the four waves take every fourth 16-channel tile each
  const bool scaled = p.row_scale != nullptr;

  int slot = 0;
  for (int mt0 = m_begin; mt0 < m_end; mt0 += 64, slot ^= 1) {
    if (mt0 != m_begin) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // tile mt0 landed for every wave; the other slot is free again
      asm volatile("" ::: "memory");
    }
    if (mt0 + 64 < m_end) issue(mt0 + 64, slot ^ 1);
    const char* tX = sT + slot * STAGE;
    const char* tY = tX + TILE;

    // per-sample DropPath factor of this lane's 16 token rows (16 mt + 4 fg + r): a tile spans at most two samples
    float sc[4][4];
    if (scaled) {
      const int b0 = __builtin_amdgcn_readfirstlane(mt0 / p.rows_per_scale);
      const int bound = (b0 + 1) * p.rows_per_scale;
      const int b1i = __builtin_amdgcn_readfirstlane(min(b0 + 1, (p.M - 1) / p.rows_per_scale));
      const float sA = sload_f32(p.row_scale + b0), sB = sload_f32(p.row_scale + b1i);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[mt][r] = (mt0 + mt * 16 + 4 * fg + r < bound) ? sA : sB;
    }

    // ---- producers: h, dg for this wave's 32 hidden units x 64 tokens; rows = tokens (4 fg + r), cols = hidden (fr)
    bf16x8 gfrag[2][2], dhfrag[2][2];          // [hidden tile][token-tile pair]: k-slot (fg, jj) <-> token 32 pair + 16 (jj>>2) + 4 fg + (jj&3)
    bf16x8 w1f[2][KS_C], w2f[2][KS_C];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int ks = 0; ks < KS_C; ++ks) {
        w1f[jt][ks] = ldfrag(sW1 + toff<C>(wave * 32 + jt * 16 + fr, ks * 32 + fg * 8));
        w2f[jt][ks] = ldfrag(sW2T + toff<C>(wave * 32 + jt * 16 + fr, ks * 32 + fg * 8));
      }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      bf16x8 xf[KS_C], yf[KS_C];
#pragma unroll
      for (int ks = 0; ks < KS_C; ++ks) {
        xf[ks] = *(const bf16x8*)(tX + mt * 16 * ROWB + xoff[ks]);
        yf[ks] = *(const bf16x8*)(tY + mt * 16 * ROWB + xoff[ks]);
      }
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        f32x4 h = {0.f, 0.f, 0.f, 0.f}, dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS_C; ++ks) {
          h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks], w1f[jt][ks], h, 0, 0, 0);
          dg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[ks], w2f[jt][ks], dg, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 hv = f32x2{h[r], h[r + 1]} + b1v[jt];
          f32x2 gv, dgv;
          gelu_fast_both2(hv, gv, dgv);
          f32x2 dhv = f32x2{dg[r], dg[r + 1]} * dgv;
          if (scaled) {                        // dy's factor moves onto the two products it enters linearly
            const f32x2 s2 = {sc[mt][r], sc[mt][r + 1]};
            dhv *= s2; gv *= s2;
          }
          db1acc[jt] += dhv[0] + dhv[1];
          gfrag[jt][mt >> 1][(mt & 1) * 4 + r] = (bf16)gv[0];
          gfrag[jt][mt >> 1][(mt & 1) * 4 + r + 1] = (bf16)gv[1];
          dhfrag[jt][mt >> 1][(mt & 1) * 4 + r] = (bf16)dhv[0];
          dhfrag[jt][mt >> 1][(mt & 1) * 4 + r + 1] = (bf16)dhv[1];
        }
      }
    }
    // ---- weight-gradient products, k = tokens (two k32 steps per 64-token tile)
#pragma unroll
    for (int pair = 0; pair < 2; ++pair) {
#pragma unroll
      for (int ct = 0; ct < CT16; ++ct) {
        typedef __attribute__((address_space(3))) s16x4* lptr;
        const char* ax = tX + pair * 32 * ROWB + toffs[ct];
        const char* ay = tY + pair * 32 * ROWB + toffs[ct];
        s16x4 xa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)ax), xb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ax + 16 * ROWB));
        s16x4 ya = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)ay), yb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ay + 16 * ROWB));
        const unsigned long long xl = __builtin_bit_cast(unsigned long long, xa), xh = __builtin_bit_cast(unsigned long long, xb);
        const unsigned long long yl = __builtin_bit_cast(unsigned long long, ya), yh = __builtin_bit_cast(unsigned long long, yb);
        const bf16x8 xt = __builtin_bit_cast(bf16x8, u32x4{(unsigned)xl, (unsigned)(xl >> 32), (unsigned)xh, (unsigned)(xh >> 32)});
        const bf16x8 yt = __builtin_bit_cast(bf16x8, u32x4{(unsigned)yl, (unsigned)(yl >> 32), (unsigned)yh, (unsigned)(yh >> 32)});
        if (do_db2 && (ct & 3) == wave) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float sv = scaled ? sc[2 * pair + (e >> 2)][e & 3] : 1.0f;
            db2acc[ct >> 2] += (float)yt[e] * sv;
          }
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          dw1[jt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dhfrag[jt][pair], xt, dw1[jt][ct], 0, 0, 0);   // rows j, cols c
          dw2[ct][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yt, gfrag[jt][pair], dw2[ct][jt], 0, 0, 0);    // rows c, cols j
        }
      }
    }
  }
  // ---- flush
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int jb = j0 + wave * 32 + jt * 16;
#pragma unroll
    for (int ct = 0; ct < CT16; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        atomicAdd(&p.dw1[(long)(jb + 4 * fg + r) * C + ct * 16 + fr], dw1[jt][ct][r]);
        atomicAdd(&p.dw2[(long)(ct * 16 + 4 * fg + r) * p.hid + jb + fr], dw2[ct][jt][r]);
      }
    float v = db1acc[jt];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (fg == 0) atomicAdd(&p.db1[jb + fr], v);
  }
  if (do_db2) {
#pragma unroll
    for (int q = 0; q < CT16 / 4; ++q) {
      float v = db2acc[q];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (fg == 0) atomicAdd(&p.db2[(q * 4 + wave) * 16 + fr], v);
    }
  }
}

// ---- round 3: the same computation with (a) the activation's VALU work of token sub-tile mt issued in the shadow of sub-tile mt + 1's
// producer MFMAs and of the weight-gradient MFMAs (slots pinned by sched_barrier(0), as in mlp_pipe_kernel), (b) the per-sample
// DropPath factor taken per 64-token TILE (the launch takes this path when rows_per_scale % 64 == 0: a tile then never straddles two
// samples): one scalar per tile instead of sixteen compare / select pairs, and a tile whose factor is 0 -- the sample's branch was
// dropped -- is skipped altogether, (c) fc1's bias as the producers' accumulator initialiser and db1 = dh^T 1 as one more MFMA column
// block instead of VALU adds, (d) NW = 8 waves of ONE 16-unit hidden tile each at C = 128, where the four-wave form needs 446 registers
// per lane (one wave per SIMD: a VALU-bound loop then issues one instruction per ~7 cycles instead of one per ~3).
__device__ __forceinline__ void gelu_fast_both1(float x, float& g, float& dg) {
#if MVLT_GELU_POLY & 4
  gelu_poly_both1(x, g, dg);
  return;
#endif
  const float xc = __builtin_amdgcn_fmed3f(x, -7.0f, 7.0f);
  const float x2 = xc * xc;
  const float t = xc * __builtin_fmaf(__builtin_fmaf(x2, -MVLT_LOG2E * MVLT_GP2, -MVLT_LOG2E * MVLT_GP1), x2, -MVLT_LOG2E * MVLT_GP0);
  const float e = __builtin_amdgcn_exp2f(t);
  const float sg = __builtin_amdgcn_rcpf(e + 1.0f);
  const float up = __builtin_fmaf(__builtin_fmaf(x2, 5.0f * MVLT_GP2, 3.0f * MVLT_GP1), x2, MVLT_GP0);
  g = x * sg;
  dg = __builtin_fmaf(x * up, __builtin_fmaf(-sg, sg, sg), sg);          // s + x u' s (1 - s)
}

template <int C, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 2 : 2) void mlp_wgrad2_kernel(mvlt_mlp_args p, int m_per_split, int splits, int ny, void* part1_, void* part2_) {
  // (void*: a __bf16* parameter makes rocprofv3 print the kernel's MANGLED name -- its demangler does not know DF16b -- and every name-keyed tool of tools/ loses the kernel)
  bf16* const part1 = (bf16*)part1_;
  bf16* const part2 = (bf16*)part2_;
  constexpr int NTH = NW * 64;
  constexpr int JT = 8 / NW;                   // 16-unit hidden tiles per wave: 2 (4 waves) / 1 (8 waves)
  constexpr int KS_C = C / 32, CT16 = C / 16;
  constexpr int ROWB = 2 * C;                  // bytes per token row
  constexpr int CH = C / 8;                    // 16-B slots per row
  constexpr int RPP = NTH / CH;                // rows per pass of the whole workgroup
  constexpr int IT = 64 / RPP;                 // passes per tile
  constexpr int TILE = 64 * ROWB;
  constexpr int STAGE = 2 * TILE;              // x tile | dy tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sW1 = (bf16*)smem;                     // [128][C]
  bf16* sW2T = sW1 + 128 * C;                  // [128][C]
  char* sT = smem + 2 * 128 * C * 2;           // [2][x tile | dy tile]
  constexpr int TILE_ = 64 * 2 * C;
  float* sLut = (float*)(sT + 2 * 2 * TILE_);  // activation table (MVLT_GELU_LUT)
  const char* const lut0 = (const char*)sLut - 8 * MVLT_GELU_LUT_BASE;
  const unsigned sT_lds = (unsigned)(uintptr_t)sT;

  // one token split (all its ny hidden blocks) per XCD: x / dy rows come through that L2 once
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int zq = kq / ny, by = kq - zq * ny;
  const int bz = zq * 8 + xcd;
  if (bz >= splits) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int j0 = by * 128, jw = wave * 16 * JT;                   // this wave's hidden units: j0 + jw + 16 jt + (column)
  const int m_begin = bz * m_per_split, m_end = min(p.M, m_begin + m_per_split);

  // ---- DMA geometry (same image as mlp_wgrad_kernel)
  const int l_row0 = tid / CH;
  const int l_chunk = (tid % CH) ^ (wg_hs<C>(l_row0) << 1);
  const char* xsrc = (const char*)p.x + l_chunk * 16;
  const char* ysrc = (const char*)p.dy + l_chunk * 16;
  const char* zsrc = (const char*)g_zero_page + ((tid * 16 + (blockIdx.x & 15) * 4096) & 65535);
  auto issue = [&](int mt, int slot) {
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int m = mt + j * RPP + l_row0;
      const bool ok = m < m_end;
      const unsigned long long off = (unsigned long long)(unsigned)m * ROWB;
      const unsigned dst = __builtin_amdgcn_readfirstlane(sT_lds + slot * STAGE + (j * NTH + wave * 64) * 16);
      glds16(ok ? xsrc + off : zsrc, dst);
      glds16(ok ? ysrc + off : zsrc, dst + TILE);
    }
  };
  issue(m_begin, 0);

  for (int u = tid; u < 128 * (C / 8); u += NTH) {
    int r = u / (C / 8), ch = u % (C / 8);
    *(u32x4*)(sW1 + toff<C>(r, ch * 8)) = *(const u32x4*)((const bf16*)p.w1 + (long)(j0 + r) * C + ch * 8);
    *(u32x4*)(sW2T + toff<C>(r, ch * 8)) = *(const u32x4*)((const bf16*)p.wc + (long)(j0 + r) * C + ch * 8);
  }
  if (MVLT_GELU_LUT & 1) gelu_lut_dma<NW>((unsigned)(uintptr_t)sLut, wave, lane);
  float b1v[JT];
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) b1v[jt] = p.b1[j0 + jw + jt * 16 + fr];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                             // weights visible (and tile 0 has landed)

  // ---- fragment geometry
  const int hs_row = wg_hs<C>(fr);                                   // producers: token row 16 mt + fr
  int xoff[KS_C];
#pragma unroll
  for (int ks = 0; ks < KS_C; ++ks) xoff[ks] = fr * ROWB + (((ks * 4 + fg) ^ (hs_row << 1)) << 4);
  const int L = lane & 15;
  const int trow = 4 * fg + (L >> 2);                                // transposed reads: token row 32 pair + trow (+16)
  const int hs_t = wg_hs<C>(trow);
  int toffs[CT16];
#pragma unroll
  for (int ct = 0; ct < CT16; ++ct) toffs[ct] = trow * ROWB + ((ct ^ hs_t) << 5) + ((L & 3) << 3);

  // the hidden-unit operands of the producers stay in registers: 2 JT KS_C fragments (32 registers at either geometry)
  bf16x8 w1f[JT][KS_C], w2f[JT][KS_C];
#pragma unroll
  for (int jt = 0; jt < JT; ++jt)
#pragma unroll
    for (int ks = 0; ks < KS_C; ++ks) {
      w1f[jt][ks] = ldfrag(sW1 + toff<C>(jw + jt * 16 + fr, ks * 32 + fg * 8));
      w2f[jt][ks] = ldfrag(sW2T + toff<C>(jw + jt * 16 + fr, ks * 32 + fg * 8));
    }

  f32x4 dw1[JT][CT16], dw2[CT16][JT], db1m[JT];
#pragma unroll
  for (int a = 0; a < JT; ++a) {
    db1m[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < CT16; ++b) { dw1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; dw2[b][a] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  // db2[c] = sum_m s_m dy[m][c]: wave w owns channel tile w (CT16 == NW at both geometries) and adds one MFMA against a ones fragment per
  // token-tile pair -- every column of that product holds the row sums.  Every workgroup computes it (two MFMAs per tile, no branch in
  // the loop); the by == 0 ones flush it.
  // The per-sample factor s never touches the per-element path: the accumulators hold UNSCALED sums under the invariant
  // true sum = run_sc x accumulator.  Tiles of dropped samples (s = 0) are skipped; when a tile's factor differs from run_sc (DropPath
  // hands out one non-zero value per block, so in practice never) the accumulators are rescaled by run_sc / s first.  The flush multiplies
  // by run_sc.
  static_assert(CT16 == NW, "one channel tile of dy per wave");
  f32x4 db2m = {0.f, 0.f, 0.f, 0.f};
  float run_sc = 0.0f;                         // 0: nothing accumulated yet
  const int toff_own = trow * ROWB + ((wave ^ hs_t) << 5) + ((L & 3) << 3);
  const bool do_db2 = by == 0;
  const bool scaled = p.row_scale != nullptr;
  const bf16x8 ones = {(bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f, (bf16)1.f};

  int slot = 0;
  for (int mt0 = m_begin; mt0 < m_end; mt0 += 64, slot ^= 1) {
    if (mt0 != m_begin) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // tile mt0 landed for every wave; the other slot is free again
      asm volatile("" ::: "memory");
    }
    if (mt0 + 64 < m_end) issue(mt0 + 64, slot ^ 1);
    // DropPath factor of this tile's sample (wave-uniform); a dropped sample contributes nothing to any of the four gradients
    float sc = 1.0f;
    if (scaled) {
      sc = sload_f32(p.row_scale + __builtin_amdgcn_readfirstlane(mt0 / p.rows_per_scale));
      if (sc == 0.0f) continue;
    }
    if (sc != run_sc) {
      if (run_sc != 0.0f) {
        const float f = run_sc / sc;
#pragma unroll
        for (int a = 0; a < JT; ++a) {
          db1m[a] *= f;
#pragma unroll
          for (int b = 0; b < CT16; ++b) { dw1[a][b] *= f; dw2[b][a] *= f; }
        }
        db2m *= f;
      }
      run_sc = sc;
    }
    const char* tX = sT + slot * STAGE;
    const char* tY = tX + TILE;

    f32x4 hd[2][JT][2];                        // [sub-tile parity][hidden tile][h | dg]
    u32x4 gfrag[JT][2], dhfrag[JT][2];         // [hidden tile][token-tile pair]: k-slot (fg, jj) <-> token 32 pair + 16 (jj>>2) + 4 fg + (jj&3)
    auto produce = [&](auto mtc) {
      constexpr int mt = decltype(mtc)::value;
      bf16x8 xf[KS_C], yf[KS_C];
#pragma unroll
      for (int ks = 0; ks < KS_C; ++ks) {
        xf[ks] = *(const bf16x8*)(tX + mt * 16 * ROWB + xoff[ks]);
        yf[ks] = *(const bf16x8*)(tY + mt * 16 * ROWB + xoff[ks]);
      }
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) {
        f32x4 h = {b1v[jt], b1v[jt], b1v[jt], b1v[jt]}, dg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS_C; ++ks) {
          h = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ks], w1f[jt][ks], h, 0, 0, 0);
          dg = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[ks], w2f[jt][ks], dg, 0, 0, 0);
        }
        hd[mt & 1][jt][0] = h;
        hd[mt & 1][jt][1] = dg;
      }
    };
    auto activate = [&](auto mtc) {
      constexpr int mt = decltype(mtc)::value;
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          float g0, g1, d0, d1;
          if (MVLT_GELU_LUT & 1) {
            const float h0 = hd[mt & 1][jt][0][r], h1 = hd[mt & 1][jt][0][r + 1];
            float ph0, ph1;
            gelu_lut_both2(lut0, h0, h1, ph0, d0, ph1, d1);
            g0 = h0 * ph0; g1 = h1 * ph1;
          } else {
            gelu_fast_both1(hd[mt & 1][jt][0][r], g0, d0);
            gelu_fast_both1(hd[mt & 1][jt][0][r + 1], g1, d1);
          }
          const float q0 = hd[mt & 1][jt][1][r], q1 = hd[mt & 1][jt][1][r + 1];
          gfrag[jt][mt >> 1][(mt & 1) * 2 + (r >> 1)] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{g0, g1}, bf16x2));
          dhfrag[jt][mt >> 1][(mt & 1) * 2 + (r >> 1)] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{q0 * d0, q1 * d1}, bf16x2));
        }
    };
    auto consume = [&](auto pairc, auto ct0c, auto ct1c) {       // weight-gradient products of token-tile pair `pair`, channel tiles [ct0, ct1)
      constexpr int pair = decltype(pairc)::value, ct0 = decltype(ct0c)::value, ct1 = decltype(ct1c)::value;
#pragma unroll
      for (int ct = ct0; ct < ct1; ++ct) {
        typedef __attribute__((address_space(3))) s16x4* lptr;
        const char* ax = tX + pair * 32 * ROWB + toffs[ct];
        const char* ay = tY + pair * 32 * ROWB + toffs[ct];
        s16x4 xa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)ax), xb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ax + 16 * ROWB));
        s16x4 ya = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)ay), yb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ay + 16 * ROWB));
        const unsigned long long xl = __builtin_bit_cast(unsigned long long, xa), xh = __builtin_bit_cast(unsigned long long, xb);
        const unsigned long long yl = __builtin_bit_cast(unsigned long long, ya), yh = __builtin_bit_cast(unsigned long long, yb);
        const bf16x8 xt = __builtin_bit_cast(bf16x8, u32x4{(unsigned)xl, (unsigned)(xl >> 32), (unsigned)xh, (unsigned)(xh >> 32)});
        const bf16x8 yt = __builtin_bit_cast(bf16x8, u32x4{(unsigned)yl, (unsigned)(yl >> 32), (unsigned)yh, (unsigned)(yh >> 32)});
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
          dw1[jt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, dhfrag[jt][pair]), xt, dw1[jt][ct], 0, 0, 0);   // rows j, cols c
          dw2[ct][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yt, __builtin_bit_cast(bf16x8, gfrag[jt][pair]), dw2[ct][jt], 0, 0, 0);    // rows c, cols j
        }
      }
      if (ct0 == 0) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)        // db1[j] += sum over the pair's 32 tokens of dh: one more column block, against ones
          db1m[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, dhfrag[jt][pair]), ones, db1m[jt], 0, 0, 0);
        typedef __attribute__((address_space(3))) s16x4* lptr;
        const char* ay = tY + pair * 32 * ROWB + toff_own;
        s16x4 ya = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)ay), yb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(ay + 16 * ROWB));
        const unsigned long long yl = __builtin_bit_cast(unsigned long long, ya), yh = __builtin_bit_cast(unsigned long long, yb);
        const bf16x8 yo = __builtin_bit_cast(bf16x8, u32x4{(unsigned)yl, (unsigned)(yl >> 32), (unsigned)yh, (unsigned)(yh >> 32)});
        db2m = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yo, ones, db2m, 0, 0, 0);
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using IH = std::integral_constant<int, CT16 / 2>;
    using IF = std::integral_constant<int, CT16>;
    produce(I0{});
    __builtin_amdgcn_sched_barrier(0);
    produce(I1{}); activate(I0{});
    __builtin_amdgcn_sched_barrier(0);
    produce(I2{}); activate(I1{});
    __builtin_amdgcn_sched_barrier(0);
    produce(I3{}); consume(I0{}, I0{}, IH{}); activate(I2{});
    __builtin_amdgcn_sched_barrier(0);
    consume(I0{}, IH{}, IF{}); activate(I3{});
    __builtin_amdgcn_sched_barrier(0);
    consume(I1{}, I0{}, IF{});
  }
  // ---- flush (true sums = run_sc x accumulators)
  if (part1) {
    // round 6: no atomics for the two weight gradients -- this token split's sums leave as bf16 PARTIAL tiles, part1[bz][hid][C] and part2[bz][C][hid], and the ordered fold of
    // the TN GEMMs (tn_fold_multi_kernel) adds the splits into dw1 / dw2: 8.4 M fp32 atomics per launch less, and bit-identical fc gradients from run to run.  Through a
    // per-wave LDS tile so that the partials leave as 16-byte pieces of contiguous rows (the operand tiles are dead: every wave is past the token loop).
    constexpr int LD1 = C + 8;                 // dW1 rows: [16 hidden units][C]
    constexpr int LD2 = 16 * JT + 8;           // dW2 rows: [16 channels][this wave's 16 JT hidden units]
    constexpr int PERW = 16 * (LD1 > LD2 ? LD1 : LD2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    bf16* const st = (bf16*)smem + wave * PERW;
    bf16* const P1 = part1 + (size_t)bz * p.hid * C;
    bf16* const P2 = part2 + (size_t)bz * p.hid * C;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      const int jb = j0 + jw + jt * 16;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ct = 0; ct < CT16; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(4 * fg + r) * LD1 + ct * 16 + fr] = (bf16)(dw1[jt][ct][r] * run_sc);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = lane; q < 16 * (C / 8); q += 64) {
        const int row = q / (C / 8), ch = q - row * (C / 8);
        *(u32x4*)(P1 + (size_t)(jb + row) * C + ch * 8) = *(const u32x4*)(st + row * LD1 + ch * 8);
      }
    }
#pragma unroll
    for (int ct = 0; ct < CT16; ++ct) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(4 * fg + r) * LD2 + jt * 16 + fr] = (bf16)(dw2[ct][jt][r] * run_sc);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < 16 * 2 * JT) {
        const int row = lane / (2 * JT), ch = lane - row * (2 * JT);
        *(u32x4*)(P2 + (size_t)(ct * 16 + row) * p.hid + j0 + jw + ch * 8) = *(const u32x4*)(st + row * LD2 + ch * 8);
      }
    }
  }
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) {
    const int jb = j0 + jw + jt * 16;
    if (!part1) {
#pragma unroll
    for (int ct = 0; ct < CT16; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        atomicAdd(&p.dw1[(long)(jb + 4 * fg + r) * C + ct * 16 + fr], dw1[jt][ct][r] * run_sc);
        atomicAdd(&p.dw2[(long)(ct * 16 + 4 * fg + r) * p.hid + jb + fr], dw2[ct][jt][r] * run_sc);
      }
    }
    if (fr == 0) {                             // every column of the ones-product holds the same sums: column 0 adds them
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&p.db1[jb + 4 * fg + r], db1m[jt][r] * run_sc);
    }
  }
  if (do_db2 && fr == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&p.db2[wave * 16 + 4 * fg + r], db2m[r] * run_sc);
  }
}

template <int C> int launch_wgrad(const mvlt_mlp_args& a, hipStream_t s) {
  const size_t lds = (size_t)2 * 128 * C * 2 + (size_t)2 * 2 * 64 * 2 * C;      // weights + 2 x (x tile | dy tile)
  const int ny = a.hid / 128;
  // one round of workgroups (2 per CU at C = 64, 1 per CU at C = 128): every extra split costs hid*C*2 fp32 atomics
  int splits = ((C == 64 ? 512 : 256) + ny - 1) / ny;
  splits = (splits + 7) / 8 * 8;
  const int mtiles = (a.M + 63) / 64;
  if (splits > mtiles) splits = mtiles;
  if (splits < 1) splits = 1;
  const int m_per_split = ((mtiles + splits - 1) / splits) * 64;
  splits = (a.M + m_per_split - 1) / m_per_split;
  if (!a.row_scale || a.rows_per_scale % 64 == 0) {      // tile-uniform DropPath factor: the round-3 kernel (other sample lengths -- 96-px inputs, T = 20 -- take the round-2 one below)
    constexpr int NW = C == 64 ? 4 : 8;
    const size_t lds_t = lds + ((MVLT_GELU_LUT & 1) ? GELU_LUT_BYTES : 0);
    mvlt_max_lds<(mlp_wgrad2_kernel<C, NW>)>();
    // the token splits' partial tiles of dW1 and dW2 in the caller's scratch (ONE region for both, room for two descriptors) + two folds behind the kernel -- launched at
    // once, or appended to the scratch's pending table when the caller defers (MVLT_MLP_DW_PARTIALS=0: fp32 atomics as in round 5)
    static const bool part_ok = !(getenv("MVLT_MLP_DW_PARTIALS") && atoi(getenv("MVLT_MLP_DW_PARTIALS")) == 0);
    bf16 *part1 = nullptr, *part2 = nullptr;
    mvlt_gemm_tn_args f1 = {}, f2 = {};
    const long pbytes = (long)splits * a.hid * C * 2;
    if (part_ok && a.partials && ((uintptr_t)a.dw1 & 15) == 0 && ((uintptr_t)a.dw2 & 15) == 0) {
      f1.C = a.dw1; f1.N1 = a.hid; f1.N2 = C; f1.ldc = C;
      f2.C = a.dw2; f2.N1 = C; f2.N2 = a.hid; f2.ldc = a.hid;
      f1.partials = f2.partials = a.partials; f1.partials_bytes = f2.partials_bytes = a.partials_bytes; f1.defer_fold = f2.defer_fold = a.defer_fold;
      part1 = mvlt_fold_acquire_ext(f1, 2 * pbytes, s, 2);
      if (part1) part2 = part1 + pbytes / 2;
    }
    MVLT_LAUNCH((mlp_wgrad2_kernel<C, NW>), dim3(8 * ((splits + 7) / 8) * ny), dim3(NW * 64), lds_t, s, a, m_per_split, splits, ny, (void*)part1, (void*)part2);
    if (part1) {
      mvlt_fold_launch_ext(f1, part1, splits, s);
      mvlt_fold_launch_ext(f2, part2, splits, s);
    }
    return mvlt_check_launch("mvlt_mlp_bwd_dw");
  }
  mvlt_max_lds<(mlp_wgrad_kernel<C>)>();
  MVLT_LAUNCH((mlp_wgrad_kernel<C>), dim3(8 * ((splits + 7) / 8) * ny), dim3(NT), lds, s, a, m_per_split, splits, ny);
  return mvlt_check_launch("mvlt_mlp_bwd_dw");
}

template <int C, int MODE> int launch(const mvlt_mlp_args& a, hipStream_t s) {
  constexpr int BM = 128, JC = MlpGeo<C>::JC;
  size_t lds = (size_t)(2 * JC * C * (MODE == 1 ? 2 : 1) + 2 * C * JC) * 2 + (size_t)a.hid * 4;
  if (MODE == 1 && ((MVLT_GELU_LUT >> 1) & 1)) lds += GELU_LUT_BYTES;
  const size_t stage = (size_t)4 * 16 * (C + 4) * 4;       // epilogue staging (4 waves x 16 rows) reuses the weight buffers
  if (lds < stage) lds = stage;
  // the software-pipelined kernel (no pre-activation store: nothing in the step asks for one); at C = 128 the input gradient fits since the polynomial GELU'
  // (44 B of scratch, 297 us against the round-2 kernel's 339 us).  The round-2 kernel below keeps the launches with a pre-activation output or hid == 64.
  if (!a.h_out && a.hid >= 128) {
    size_t l2 = (size_t)2 * (MODE == 1 ? 2 : 1) * 32 * 2 * C + (size_t)2 * C * 64 + (size_t)a.hid * 4;
    if ((MVLT_GELU_LUT >> (MODE == 1 ? 1 : 2)) & 1) l2 += GELU_LUT_BYTES;
    if (l2 < stage) l2 = stage;
    mvlt_max_lds<(mlp_pipe_kernel<C, MODE>)>();
    MVLT_LAUNCH((mlp_pipe_kernel<C, MODE>), dim3((a.M + BM - 1) / BM), dim3(NT), l2, s, a);
    return mvlt_check_launch(MODE == 0 ? "mvlt_mlp_fwd" : "mvlt_mlp_bwd_dx");
  }
  mvlt_max_lds<(mlp_fused_kernel<C, MODE>)>();
  MVLT_LAUNCH((mlp_fused_kernel<C, MODE>), dim3((a.M + BM - 1) / BM), dim3(NT), lds, s, a);
  return mvlt_check_launch(MODE == 0 ? "mvlt_mlp_fwd" : "mvlt_mlp_bwd_dx");
}

int check(const mvlt_mlp_args* a, const char* who) {
  MVLT_REQUIRE(a && (a->x || a->ln_x) && a->w1 && a->wb && a->b1 && (a->out || a->out_op || a->lnb_x), "%s: null pointer", who);
  MVLT_REQUIRE(a->C == 64 || a->C == 128, "%s: C must be 64 or 128 (stage 1 / 2), got %d", who, a->C);
  MVLT_REQUIRE(a->hid > 0 && a->hid % 64 == 0, "%s: hidden size must be a multiple of 64", who);
  MVLT_REQUIRE(!a->row_scale || a->rows_per_scale > 0, "%s: row_scale needs rows_per_scale", who);
  return MVLT_OK;
}

}  // namespace

extern "C" int mvlt_mlp_fwd(const mvlt_mlp_args* a, void* stream) {
  if (int e = check(a, "mvlt_mlp_fwd")) return e;
  MVLT_REQUIRE(a->b2 && a->residual, "mvlt_mlp_fwd: b2 and residual are required");
  MVLT_REQUIRE(!a->out_op || ((uintptr_t)a->out_op & 15) == 0, "mvlt_mlp_fwd: out_op must be 16-byte aligned");
  MVLT_REQUIRE(!a->post_y || (a->post_gamma && a->post_beta && a->post_mean && a->post_rstd && ((uintptr_t)a->post_y & 15) == 0),
               "mvlt_mlp_fwd: post_y needs post_gamma / post_beta / post_mean / post_rstd and 16-byte alignment");
  MVLT_REQUIRE(!a->ln_x || (a->ln_gamma && a->ln_beta && a->ln_y && a->ln_mean && a->ln_rstd && ((uintptr_t)a->ln_x & 15) == 0 && ((uintptr_t)a->ln_y & 15) == 0),
               "mvlt_mlp_fwd: the folded LayerNorm needs gamma, beta, y, mean, rstd and 16-byte aligned rows");
  if (a->M <= 0) return MVLT_OK;
  return a->C == 64 ? launch<64, 0>(*a, (hipStream_t)stream) : launch<128, 0>(*a, (hipStream_t)stream);
}

extern "C" int mvlt_mlp_bwd_dx(const mvlt_mlp_args* a, void* stream) {
  if (int e = check(a, "mvlt_mlp_bwd_dx")) return e;
  MVLT_REQUIRE(a->dy && a->wc, "mvlt_mlp_bwd_dx: dy and wc (W2^T) are required");
  MVLT_REQUIRE(!a->lnb_x || (a->lnb_mean && a->lnb_rstd && a->lnb_gamma && a->lnb_dx && a->lnb_partials && (!a->lnb_dx2 || (a->lnb_dx2_scale && a->lnb_dx2_rows_per_scale > 0)) &&
                             ((uintptr_t)a->lnb_x & 15) == 0 && ((uintptr_t)a->lnb_dx & 15) == 0 && ((uintptr_t)a->lnb_dx2 & 15) == 0),
               "mvlt_mlp_bwd_dx: lnb_x needs lnb_mean / lnb_rstd / lnb_gamma / lnb_dx / lnb_partials (and a scale for lnb_dx2), 16-byte aligned");
  if (a->M <= 0) return MVLT_OK;
  return a->C == 64 ? launch<64, 1>(*a, (hipStream_t)stream) : launch<128, 1>(*a, (hipStream_t)stream);
}

extern "C" int mvlt_mlp_bwd_dw(const mvlt_mlp_args* a, void* stream) {
  MVLT_REQUIRE(a && a->x && a->dy && a->w1 && a->wc && a->b1 && a->dw1 && a->db1 && a->dw2 && a->db2, "mvlt_mlp_bwd_dw: null pointer");
  MVLT_REQUIRE(a->C == 64 || a->C == 128, "mvlt_mlp_bwd_dw: C must be 64 or 128, got %d", a->C);
  MVLT_REQUIRE(a->hid > 0 && a->hid % 128 == 0, "mvlt_mlp_bwd_dw: hidden size must be a multiple of 128");
  MVLT_REQUIRE(!a->row_scale || a->rows_per_scale > 0, "mvlt_mlp_bwd_dw: row_scale needs rows_per_scale");
  if (a->M <= 0) return MVLT_OK;
  return a->C == 64 ? launch_wgrad<64>(*a, (hipStream_t)stream) : launch_wgrad<128>(*a, (hipStream_t)stream);
}
