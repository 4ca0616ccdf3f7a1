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
#include "common.h"
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

template <typename T> __device__ __forceinline__ void store_out(void* base, long idx, float v, int out_fp32) {
  if (out_fp32) ((float*)base)[idx] = v; else ((bf16*)base)[idx] = (bf16)v;
}
__device__ __forceinline__ float load_out(const void* base, long idx, int out_fp32) {
  return out_fp32 ? ((const float*)base)[idx] : (float)((const bf16*)base)[idx];
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
  if (p.bias) {
    if (nc_lane + 8 <= p.N && (((uintptr_t)p.bias & 15) == 0)) {
      f32x4 b0 = *(const f32x4*)(p.bias + nc_lane), b1 = *(const f32x4*)(p.bias + nc_lane + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (nc_lane + e < p.N) bias8[e] = p.bias[nc_lane + e];
    }
  }
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
            *(f32x4*)((float*)base + idx) = f32x4{o[0], o[1], o[2], o[3]};
            *(f32x4*)((float*)base + idx + 4) = f32x4{o[4], o[5], o[6], o[7]};
          } else {
            bf16x8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (bf16)o[e];
            *(bf16x8*)((bf16*)base + idx) = a;
          }
        };
        if (p.act == 1) {
          if (p.H) store8(p.H, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
        } else if (p.act == 2) {
          float h8[8];
          load8(p.H, h8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_grad(h8[e]);
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
          store_out<T>(p.C, idx + e, x, ofp32);
        }
      }
    }
  }
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
        if (n1 < p.N1 && n2 < p.N2) atomicAdd(p.trans_c ? &p.C[(long)n2 * p.ldc + n1] : &p.C[(long)n1 * p.ldc + n2], acc[i][j][r]);
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
  if (int e = check_rowmap(a->a_map, "mvlt_gemm_nt a_map")) return e;
  if (int e = check_rowmap(a->c_map, "mvlt_gemm_nt c_map")) return e;
  MVLT_REQUIRE(a->a_map.mode == 0 || a->K == a->a_map.r * a->a_map.r * a->a_map.c_seg, "mvlt_gemm_nt: gather K != r*r*c_seg");
  MVLT_REQUIRE(a->c_map.mode == 0 || a->N == a->c_map.r * a->c_map.r * a->c_map.c_seg, "mvlt_gemm_nt: scatter N != r*r*c_seg");
  MVLT_REQUIRE(a->c_map.mode != 2, "mvlt_gemm_nt: the 3x3 map is a gather only (its dgrad is a gather with flipped taps)");
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
  dim3 grid((unsigned)(8 * ((tiles_m + 7) / 8) * tiles_n)), block(NTHREADS);
  if (a->dtype == 0) {
    if (narrow) hipLaunchKernelGGL((gemm_nt_kernel<bf16, 64>), grid, block, lds, s, *a, nbuf);
    else hipLaunchKernelGGL((gemm_nt_kernel<bf16, 128>), grid, block, lds, s, *a, nbuf);
  } else {
    if (narrow) hipLaunchKernelGGL((gemm_nt_kernel<float, 64>), grid, block, lds, s, *a, nbuf);
    else hipLaunchKernelGGL((gemm_nt_kernel<float, 128>), grid, block, lds, s, *a, nbuf);
  }
  return mvlt_check_launch("mvlt_gemm_nt");
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
  if (a->M == 0) return MVLT_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool narrow = a->N2 <= 64;
  const int bn = narrow ? 64 : 128;
  const int t1 = (a->N1 + BM - 1) / BM, t2 = (a->N2 + bn - 1) / bn;
  const int mtiles = (a->M + TBK - 1) / TBK;
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
    if (narrow) hipLaunchKernelGGL((gemm_tn_kernel<bf16, 64>), grid, block, lds, s, *a, m_per_split);
    else hipLaunchKernelGGL((gemm_tn_kernel<bf16, 128>), grid, block, lds, s, *a, m_per_split);
  } else {
    if (narrow) hipLaunchKernelGGL((gemm_tn_kernel<float, 64>), grid, block, lds, s, *a, m_per_split);
    else hipLaunchKernelGGL((gemm_tn_kernel<float, 128>), grid, block, lds, s, *a, m_per_split);
  }
  return mvlt_check_launch("mvlt_gemm_tn");
}
