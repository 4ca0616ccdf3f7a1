// HBM-bound helper kernels of the MVLT hot path (gfx950): BERT embedding (+LN +dropout), image patchify,
// masked-index selection, row gather/scatter, cross-entropy (fwd + in-place grad), fused AdamW + bf16 re-cast.
// Everything is coalesced 16-byte traffic with fp32 math; none of this is reshaped into a GEMM.
#include "common.h"
#include "../../include/mvlt_hip.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------ BERT embeddings
// y[b,t,:] = dropout(LN(word[ids[b,t]] + type[0] + pos[t])), hidden = 768 (one wave per token, 12 floats per lane).
// Replaces transformers BertEmbeddings.forward (call site reference libs/pvlt.py:326).
template <typename T, int HID>
__global__ __launch_bounds__(NT) void bert_embed_fwd_kernel(const long* ids, const float* word, const float* pos, const float* type0,
                                                            const float* gamma, const float* beta, const uint8_t* keep, float inv_keep,
                                                            T* y, float* mean_out, float* rstd_out, int rows, int Tlen, float eps) {
  constexpr int PER = HID / 64;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long id = ids[row];
  const int t = row % Tlen;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int c = lane + 64 * i;
    v[i] = word[id * HID + c] + type0[c] + pos[(long)t * HID + c];
    s += v[i];
  }
  const float mean = wave_sum(s) * (1.0f / HID);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) { float d = v[i] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / HID) + eps);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int c = lane + 64 * i;
    float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
    if (keep) o = keep[(long)row * HID + c] ? o * inv_keep : 0.f;
    y[(long)row * HID + c] = (T)o;
  }
}

// backward: recompute x = word+type+pos, LN backward, then scatter: word[id] (skipping padding_idx 0, as
// nn.Embedding(padding_idx=0) does), pos[t], type[0], gamma, beta -- all fp32 atomics into zeroed grads.
// 1024-thread workgroups, at most one per CU: every workgroup ends with 3 x 768 atomics on the same 72 cache lines (dgamma, dbeta,
// dtype0), which the memory side serves one at a time at ~100 ns -- 2048 workgroups of 256 threads spent 0.2 of the kernel's
// 0.26 ms there
template <typename T, int HID, int NT>
__global__ __launch_bounds__(NT) void bert_embed_bwd_kernel(const T* dy, const long* ids, const float* word, const float* pos, const float* type0,
                                                            const float* gamma, const uint8_t* keep, float inv_keep,
                                                            const float* mean_in, const float* rstd_in,
                                                            float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                                                            int rows, int Tlen) {
  constexpr int PER = HID / 64;
  __shared__ float s_w[NT / 64][HID];       // one slice per wave (LDS atomics from sixteen waves onto the same words were most of this kernel's tail)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[PER], ab[PER], at[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) { ag[i] = 0.f; ab[i] = 0.f; at[i] = 0.f; }
  // a workgroup owns ONE position t and a slice of the batch (its waves take every (NT/64)-th sample): the position-embedding
  // gradient of that slice is summed in registers (at[] doubles as it: dtype0 = sum over all rows of the same dx) and leaves as one
  // 768-float flush per workgroup instead of one per row
  const int t = blockIdx.x % Tlen, split = blockIdx.x / Tlen, nsplit = gridDim.x / Tlen;
  const int nb = rows / Tlen;
  for (int bb = split * (NT / 64) + (threadIdx.x >> 6); bb < nb; bb += nsplit * (NT / 64)) {
    const int row = bb * Tlen + t;
    const long id = ids[row];
    const float mean = mean_in[row], rstd = rstd_in[row];
    float g[PER], xh[PER];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int c = lane + 64 * i;
      float d = (float)dy[(long)row * HID + c];
      if (keep) d = keep[(long)row * HID + c] ? d * inv_keep : 0.f;
      float x = word[id * HID + c] + type0[c] + pos[(long)t * HID + c];
      float h = (x - mean) * rstd;
      float gg = d * gamma[c];
      xh[i] = h; g[i] = gg;
      s1 += gg; s2 += gg * h;
      ag[i] += d * h; ab[i] += d;
    }
    s1 = wave_sum(s1) * (1.0f / HID);
    s2 = wave_sum(s2) * (1.0f / HID);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int c = lane + 64 * i;
      float dx = rstd * (g[i] - s1 - xh[i] * s2);
      if (id != 0) atomicAdd(&dword[id * HID + c], dx);
      at[i] += dx;
    }
  }
  // the three per-column sums of the workgroup, one after the other through the per-wave slices
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int i = 0; i < PER; ++i) s_w[wave][lane + 64 * i] = q == 0 ? ag[i] : q == 1 ? ab[i] : at[i];
    __syncthreads();
    for (int i = threadIdx.x; i < HID; i += NT) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) v += s_w[w][i];
      if (q == 0) atomicAdd(&dgamma[i], v);
      else if (q == 1) atomicAdd(&dbeta[i], v);
      else { atomicAdd(&dtype0[i], v); atomicAdd(&dpos[(long)t * HID + i], v); }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ image patchify (stage-1 PatchEmbed operand)
// P[(b, oi, oj), (c, di, dj)] = img[b, c, oi*k+di, oj*k+dj]   (NCHW fp32 image -> row-major patch matrix, K = C*k*k)
template <typename T>
__global__ __launch_bounds__(NT) void patchify_kernel(const float* img, T* out, int B, int Cin, int H, int W, int k) {
  const int Ho = H / k, Wo = W / k, K = Cin * k * k;
  const long total = (long)B * Ho * Wo * Cin * k;      // one thread per (row, c, di): k contiguous pixels
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    int di = (int)(i % k);
    long r1 = i / k;
    int oj = (int)(r1 % Wo); r1 /= Wo;
    int c = (int)(r1 % Cin); r1 /= Cin;
    int oi = (int)(r1 % Ho);
    int b = (int)(r1 / Ho);
    const float* src = img + (((long)b * Cin + c) * H + (oi * k + di)) * W + oj * k;
    T* dst = out + (((long)b * Ho + oi) * Wo + oj) * K + (c * k + di) * k;
    for (int dj = 0; dj < k; ++dj) dst[dj] = (T)src[dj];
  }
}

// k = 4, Cin = 3 (every configuration of the model): one workgroup per (b, oi) strip.  The strip's 12 image rows (3 channels x 4 rows of W
// floats) come in as whole rows, 16 B per lane on consecutive addresses, pass through LDS, and leave as the strip's W/4 patch rows of 48
// values -- contiguous in P, 16 B (bf16) per lane.  The generic kernel above writes 8-byte pieces 96 B apart: 96 us at batch 256 against
// the ~60 us the 300 MB take.
template <typename T>
__global__ __launch_bounds__(NT) void patchify_strip_kernel(const float* img, T* out, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float ptile[];      // [12][W + 4]
  const int Ho = H / 4, Wo = W / 4, LDT = W + 4;
  const int b = blockIdx.x / Ho, oi = blockIdx.x - b * Ho;
  for (int q = threadIdx.x; q < 12 * Wo; q += NT) {
    const int r = q / Wo, x4 = q - r * Wo;               // r = c * 4 + di
    *(f32x4*)(ptile + r * LDT + x4 * 4) = *(const f32x4*)(img + (((long)b * 3 + (r >> 2)) * H + (oi * 4 + (r & 3))) * W + x4 * 4);
  }
  __syncthreads();
  T* dst = out + ((long)b * Ho + oi) * Wo * 48;
  for (int q = threadIdx.x; q < Wo * 6; q += NT) {       // 8 values per thread: columns ch * 8 .. + 7 = image rows (c, di), (c, di + 1) x dj 0..3
    const int oj = q / 6, ch = q - oj * 6;
    const f32x4 a = *(const f32x4*)(ptile + (ch * 2) * LDT + oj * 4), c = *(const f32x4*)(ptile + (ch * 2 + 1) * LDT + oj * 4);
    T* d = dst + (long)oj * 48 + ch * 8;
    if constexpr (sizeof(T) == 2) {
      *(bf16x8*)d = bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)c[0], (bf16)c[1], (bf16)c[2], (bf16)c[3]};
    } else {
      *(f32x4*)d = a;
      *(f32x4*)(d + 4) = c;
    }
  }
}

// ------------------------------------------------------------------ masked-index selection (bit-exact, ordered)
// idx[0..count) = ascending positions p with labels[p] != ignore; one workgroup, ballot + prefix.
// any n (more than 64 labels per thread): 1024 labels per trip
__global__ __launch_bounds__(1024) void masked_select_loop_kernel(const long* labels, int n, long ignore, int* idx, int* count) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int start = 0; start < n; start += 1024) {
    int p = start + threadIdx.x;
    bool sel = p < n && labels[p] != ignore;
    unsigned long long m = __ballot(sel);
    int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (sel) idx[off + within] = p;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_wave[w]; s_base += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = s_base;
}
__global__ __launch_bounds__(1024) void masked_select_kernel(const long* labels, int n, long ignore, int* idx, int* count) {
  // wave w owns the contiguous labels [w * per * 64, (w + 1) * per * 64), per <= 64: row k of it is ONE coalesced 512-byte load (lane l takes label k * 64 + l; the rows in
  // batches of eight, all in flight), its ballot a wave-uniform 64-bit mask; a selected lane's place = the wave's offset + the selected of the rows before + the selected
  // lanes below it -- scalar arithmetic on the masks.  One pass, one barrier, ascending order.  (Round 5 first had 64 contiguous labels per THREAD: 21 us for 32768 labels,
  // every load instruction touching 64 cache lines; before that 1024 labels per trip on one workgroup, a load and three barriers per trip: 40 us.)
  __shared__ int s_wave[16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per = (n + 1023) / 1024;                   // rows of 64 labels per wave
  const long w0 = (long)wave * per * 64;
  int total = 0;
  // pass 1: the wave's count (pass 2 recomputes the masks from the same, now cached, loads: up to 64 of them would not stay in scalar registers)
  for (int b = 0; b < per; b += 8) {
    long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const long p = w0 + (long)(b + u) * 64 + lane; v[u] = (b + u < per && p < n) ? labels[p] : ignore; }
#pragma unroll
    for (int u = 0; u < 8; ++u) total += __popcll(__ballot(v[u] != ignore));
  }
  if (lane == 0) s_wave[wave] = total;
  __syncthreads();
  int off = 0, all = 0;
  for (int w = 0; w < 16; ++w) { const int c = s_wave[w]; if (w < wave) off += c; all += c; }
  // pass 2: places
  for (int b = 0; b < per; b += 8) {
    long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const long p = w0 + (long)(b + u) * 64 + lane; v[u] = (b + u < per && p < n) ? labels[p] : ignore; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned long long m = __ballot(v[u] != ignore);
      if (v[u] != ignore) idx[off + __popcll(m & ((1ull << lane) - 1ull))] = (int)(w0 + (long)(b + u) * 64 + lane);
      off += __popcll(m);
    }
  }
  if (threadIdx.x == 0) *count = all;
}

// ------------------------------------------------------------------ row gather / scatter-add
template <typename T>
__global__ __launch_bounds__(NT) void gather_rows_kernel(const T* src, const int* idx, T* dst, int rows, int C, int ld_src, RowMap smap) {
  constexpr int PC = 16 / sizeof(T);
  const int nch = C / PC;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (long)rows * nch; i += (long)gridDim.x * NT) {
    int r = (int)(i / nch), c = (int)(i % nch);
    *(u32x4*)(dst + (long)r * C + c * PC) = *(const u32x4*)(src + rowmap_base(smap, idx[r]) * ld_src + c * PC);
  }
}
// dst[map(idx[r])] (+)= src[r]   (rows in idx are unique)
template <typename T>
__global__ __launch_bounds__(NT) void scatter_rows_kernel(const T* src, const int* idx, T* dst, int rows, int C, int ld_dst, RowMap dmap, int accumulate) {
  constexpr int PC = 16 / sizeof(T);
  const int nch = C / PC;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (long)rows * nch; i += (long)gridDim.x * NT) {
    int r = (int)(i / nch), c = (int)(i % nch);
    T* d = dst + rowmap_base(dmap, idx[r]) * ld_dst + c * PC;
    T v[PC];
    *(u32x4*)v = *(const u32x4*)(src + (long)r * C + c * PC);
    if (accumulate) {
      T o[PC];
      *(u32x4*)o = *(const u32x4*)d;
#pragma unroll
      for (int e = 0; e < PC; ++e) v[e] = (T)((float)v[e] + (float)o[e]);
    }
    *(u32x4*)d = *(u32x4*)v;
  }
}

// ------------------------------------------------------------------ cross entropy over rows (one workgroup per row)
// fwd : lse[r] = logsumexp(logits[r,:]); loss_sum += (lse - logits[r,label]) for label != ignore; count += 1
// bwd : dlogits[r,c] = (exp(logits - lse) - [c==label]) * gscale[0] / max(count,1)   (0 for ignored rows)
template <typename T>
__global__ __launch_bounds__(NT) void ce_fwd_kernel(const T* logits, const long* labels, long ignore, float* lse, float* loss_sum, float* count,
                                                    int rows, int V, int ld) {
  __shared__ float s_m[NT / 64], s_s[NT / 64];
  const int row = blockIdx.x;
  const T* lr = logits + (long)row * ld;
  float m = -INFINITY, s = 0.f;
  int c_first = 0;
  if constexpr (sizeof(T) == 4) {
    // fp32 rows on 16-byte boundaries (the 30522-wide MLM logits, ld = 30528): four columns per load, one rescale per four
    if ((ld & 3) == 0 && (((uintptr_t)logits) & 15) == 0) {
      // eight independent 16-byte loads per thread in flight, then ONE rescale per 32 columns: with one load per trip the thread's 30 trips over a 30522-wide
      // row each waited out a full memory round trip behind the (m, s) dependency (82 us for 182 MB of logits: 2.2 TB/s)
      const int V4 = V & ~3;
      constexpr int U = 8;
      int c = threadIdx.x * 4;
      for (; c + (U - 1) * NT * 4 < V4; c += U * NT * 4) {
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *(const f32x4*)((const float*)lr + c + u * NT * 4);
        float nm = m;
#pragma unroll
        for (int u = 0; u < U; ++u) nm = fmaxf(nm, fmaxf(fmaxf(x[u][0], x[u][1]), fmaxf(x[u][2], x[u][3])));
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) a += (__expf(x[u][0] - nm) + __expf(x[u][1] - nm)) + (__expf(x[u][2] - nm) + __expf(x[u][3] - nm));
        s = s * __expf(m - nm) + a;
        m = nm;
      }
      for (; c < V4; c += NT * 4) {
        const f32x4 x = *(const f32x4*)((const float*)lr + c);
        const float nm = fmaxf(m, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
        s = s * __expf(m - nm) + (__expf(x[0] - nm) + __expf(x[1] - nm)) + (__expf(x[2] - nm) + __expf(x[3] - nm));
        m = nm;
      }
      c_first = V4;
    }
  }
  for (int c = c_first + threadIdx.x; c < V; c += NT) {
    float x = (float)lr[c];
    float nm = fmaxf(m, x);
    s = (m == -INFINITY ? 0.f : s * __expf(m - nm)) + __expf(x - nm);
    m = nm;
  }
  // combine (m, s) pairs: wave, then block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(m, o), os = __shfl_xor(s, o);
    float nm = fmaxf(m, om);
    s = (m == -INFINITY ? 0.f : s * __expf(m - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
    m = nm;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_m[wave] = m; s_s[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = s_m[0], S = s_s[0];
    for (int w = 1; w < NT / 64; ++w) {
      float nm = fmaxf(M, s_m[w]);
      S = (M == -INFINITY ? 0.f : S * __expf(M - nm)) + (s_m[w] == -INFINITY ? 0.f : s_s[w] * __expf(s_m[w] - nm));
      M = nm;
    }
    lse[row] = M + logf(S);
  }
}
// loss_sum += sum over the rows with a label of (lse - logits[row, label]); count += their number.  ONE workgroup, in a fixed order: the row kernel above used to end every
// workgroup with two atomics on the same two words -- 2980 of them for the 1490 MLM rows, served one after the other by the memory side: they, not the 182 MB of logits,
// were its 80 us (2.4 TB/s with eight loads in flight per thread just as with one).
template <typename T>
__global__ __launch_bounds__(1024) void ce_loss_finish_kernel(const T* logits, const long* labels, long ignore, const float* lse, float* loss_sum, float* count, int rows, int ld) {
  __shared__ float s_l[16], s_c[16];
  float l = 0.f, c = 0.f;
  for (int r = threadIdx.x; r < rows; r += 1024) {
    const long lab = labels[r];
    if (lab != ignore) { l += lse[r] - (float)logits[(long)r * ld + lab]; c += 1.f; }
  }
  l = wave_sum(l); c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { s_l[threadIdx.x >> 6] = l; s_c[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f, Cn = 0.f;
    for (int w = 0; w < 16; ++w) { L += s_l[w]; Cn += s_c[w]; }
    loss_sum[0] += L;
    count[0] += Cn;
  }
}

template <typename T, typename TO>
__global__ __launch_bounds__(NT) void ce_bwd_kernel(const T* logits, const long* labels, long ignore, const float* lse, const float* gscale,
                                                    const float* count, TO* dlogits, int rows, int V, int ld, int ldd) {
  const int row = blockIdx.x;
  const T* lr = logits + (long)row * ld;
  TO* dr = dlogits + (long)row * ldd;
  const long lab = labels[row];
  const float l = lse[row];
  const float sc = (lab == ignore) ? 0.f : gscale[0] / fmaxf(count[0], 1.0f);
  int c_first = 0;
  if constexpr (sizeof(T) == 4) {
    if ((ld & 3) == 0 && (ldd & 3) == 0 && (((uintptr_t)logits) & 15) == 0 && (((uintptr_t)dlogits) & 15) == 0) {
      const int V4 = V & ~3;
      for (int c = threadIdx.x * 4; c < V4; c += NT * 4) {
        const f32x4 x = *(const f32x4*)((const float*)lr + c);
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = (__expf(x[e] - l) - (c + e == lab ? 1.f : 0.f)) * sc;
        if constexpr (sizeof(TO) == 2) *(bf16x4*)((bf16*)dr + c) = bf16x4{(bf16)g[0], (bf16)g[1], (bf16)g[2], (bf16)g[3]};
        else *(f32x4*)((float*)dr + c) = g;
      }
      c_first = V4;
    }
  }
  for (int c = c_first + threadIdx.x; c < ldd; c += NT) {
    float g = 0.f;
    if (c < V) g = (__expf((float)lr[c] - l) - (c == lab ? 1.f : 0.f)) * sc;
    dr[c] = (TO)g;                        // padding columns [V, ldd) are zeroed
  }
}

// ------------------------------------------------------------------ SmoothL1 (beta = 1) over fp32 tensors: sum, then gradient
// loss_sum += sum_i l(p_i - t_i), l(d) = 0.5 d^2 for |d| < 1, |d| - 0.5 otherwise (torch.nn.functional.smooth_l1_loss, the
// T2I loss of reference engine_grid_masking.py:99); no per-element loss tensor is written.
__global__ __launch_bounds__(NT) void smooth_l1_sum_kernel(const float* pred, const float* target, long n, float* loss_sum) {
  __shared__ float s_w[NT / 64];
  float acc = 0.f;
  const long stride = (long)gridDim.x * NT * 4;
  long i = ((long)blockIdx.x * NT + threadIdx.x) * 4;
  for (; i + 3 * stride < n; i += 4 * stride) {                // four independent 16-byte loads per tensor in flight
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = *(const f32x4*)(pred + i + u * stride); b[u] = *(const f32x4*)(target + i + u * stride); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = fabsf(a[u][e] - b[u][e]);
        acc += d < 1.0f ? 0.5f * d * d : d - 0.5f;
      }
  }
  for (; i < n; i += stride) {
    const f32x4 a = *(const f32x4*)(pred + i), b = *(const f32x4*)(target + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = fabsf(a[e] - b[e]);
      acc += d < 1.0f ? 0.5f * d * d : d - 0.5f;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += s_w[w];
    atomicAdd(loss_sum, t);
  }
}
// grad_i = clamp(p_i - t_i, -1, 1) * gscale[0] * inv_n   (d/dp of mean-reduced SmoothL1 times the incoming scalar gradient)
__global__ __launch_bounds__(NT) void smooth_l1_grad_kernel(const float* pred, const float* target, long n, const float* gscale, float inv_n,
                                                            float* grad) {
  const float sc = gscale[0] * inv_n;
  for (long i = ((long)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += (long)gridDim.x * NT * 4) {
    const f32x4 a = *(const f32x4*)(pred + i), b = *(const f32x4*)(target + i);
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = __builtin_amdgcn_fmed3f(a[e] - b[e], -1.0f, 1.0f) * sc;
    *(f32x4*)(grad + i) = g;
  }
}

// ------------------------------------------------------------------ fused AdamW over a flat fp32 buffer (+ bf16 re-cast)
// torch.optim.AdamW semantics (reference main_vl.py:308 via timm create_optimizer): decoupled weight decay,
// bias-corrected moments.  lr / step-dependent scalars come from a small device array so that a captured graph replays.
// hp = {lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale}
__global__ __launch_bounds__(NT) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16* p16, long n, const float* hp,
                                                   const uint8_t* decay_mask) {
  const float lr = hp[0], b1 = hp[1], b2 = hp[2], eps = hp[3], wd = hp[4], bc1 = hp[5], bc2 = hp[6], gs = hp[7];
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  for (long i = ((long)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += (long)gridDim.x * NT * 4) {
    f32x4 pv = *(f32x4*)(p + i), gv = *(const f32x4*)(g + i), mv = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
    const uint32_t dm = decay_mask ? *(const uint32_t*)(decay_mask + i) : 0x01010101u;   // 1 byte per parameter
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gv[e] * gs;
      float pp = ((dm >> (8 * e)) & 1u) ? pv[e] * (1.0f - lr * wd) : pv[e];
      float mm = b1 * mv[e] + (1.0f - b1) * gr;
      float v2 = b2 * vv[e] + (1.0f - b2) * gr * gr;
      float denom = sqrtf(v2) * inv_sqrt_bc2 + eps;
      pv[e] = pp - step_size * mm / denom;
      mv[e] = mm; vv[e] = v2;
    }
    *(f32x4*)(p + i) = pv; *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv;
    if (p16) {
      bf16x4 o = {(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
      *(bf16x4*)(p16 + i) = o;
    }
  }
}

__global__ __launch_bounds__(NT) void cast_f32_bf16_kernel(const float* src, bf16* dst, long n) {
  for (long i = ((long)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += (long)gridDim.x * NT * 4) {
    f32x4 v = *(const f32x4*)(src + i);
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *(bf16x4*)(dst + i) = o;
  }
}

// out[row, :] = x[row, :] * scale[row / rows_per_scale]   (DropPath factor of the sample a row belongs to; contiguous [M, C])
template <typename T>
__global__ __launch_bounds__(NT) void row_scale_kernel(const T* x, const float* scale, long per_scale, long n, T* out) {
  constexpr int V = 16 / sizeof(T);
  for (long i = ((long)blockIdx.x * NT + threadIdx.x) * V; i < n; i += (long)gridDim.x * NT * V) {
    const float s = scale[i / per_scale];
    if constexpr (sizeof(T) == 2) {
      bf16x8 v = *(const bf16x8*)(x + i), o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)v[e] * s);
      *(bf16x8*)(out + i) = o;
    } else {
      f32x4 v = *(const f32x4*)(x + i);
      *(f32x4*)(out + i) = f32x4{v[0] * s, v[1] * s, v[2] * s, v[3] * s};
    }
  }
}

// ---- small classification heads (ITM: B x 2, CLS: B x 48 / B x 122; reference libs/vl_heads.py:73-104), backward: the padded operand-dtype copy of dlogits
// that the weight- and input-gradient GEMMs read, and db += column sums of dlogits into the head's TWO bias parameters (`linear.bias` and `linear_bias`), in one
// launch of one workgroup (six ATen launches before: zeros, cast, slice copy, sum, two adds -- each a drained pipeline between two large kernels)
template <typename T>
__global__ __launch_bounds__(NT) void head_grad_prep_kernel(const float* dlogits, int B, int n, int n_pad, T* dl, float* db1, float* db2) {
  __shared__ float part[NT];
  const int c = threadIdx.x % n_pad, r0 = threadIdx.x / n_pad, rstep = NT / n_pad;
  float acc = 0.f;
  if (r0 < rstep) {
    for (int r = r0; r < B; r += rstep) {
      const float v = c < n ? dlogits[(long)r * n + c] : 0.f;
      dl[(long)r * n_pad + c] = from_f32<T>(v);
      acc += v;
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < n) {
    float t = 0.f;
    for (int k = 0; k < rstep; ++k) t += part[k * n_pad + threadIdx.x];
    db1[threadIdx.x] += t;
    if (db2) db2[threadIdx.x] += t;
  }
}

// out[c][r] = (T) in[r][c]   (fp32 master weight [R,C] -> transposed compute copy for the dgrad GEMMs)
template <typename T>
__global__ __launch_bounds__(NT) void transpose_cast_kernel(const float* in, T* out, int R, int Ccols, int ld_out) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < Ccols) ? in[(long)r * Ccols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (c < Ccols && r < R) out[(long)c * ld_out + r] = (T)tile[tx][j];
  }
}

// table-driven version of the above plus a strided 3-D gather: one workgroup = one 64x64 tile or 256 gathered elements.
// The block -> descriptor search runs on an LDS copy of blk_start (one coalesced load instead of log2(ndesc) dependent global
// loads per workgroup: with 32x32 tiles and the search in global memory the launch spent most of its 115 us there).
template <typename T>
__global__ __launch_bounds__(NT) void weight_prep_kernel(const mvlt_prep_desc* descs, const int* blk_start, int ndesc, const int* blk_desc) {
  constexpr int TS = 64, MAXD = 1024;
  __shared__ float tile[TS][TS + 1];
  __shared__ int s_start[MAXD];
  const int b = blockIdx.x;
  int lo;
  if (blk_desc) {
    // the host's block -> descriptor table: two dependent SCALAR loads (uniform addresses) in front of the tile's loads instead of a 4 KB copy of blk_start
    // into LDS, a barrier, a ten-step search and a vector load of the descriptor (three full memory round trips per 8 KB tile: 97 us for ~190 MB)
    lo = __builtin_amdgcn_readfirstlane(blk_desc[b]);
  } else {
    const bool in_lds = ndesc <= MAXD;
    if (in_lds) {
      for (int i = threadIdx.x; i < ndesc; i += NT) s_start[i] = blk_start[i];
      __syncthreads();
    }
    int hi = ndesc - 1;                                // last descriptor whose first block is <= b
    lo = 0;
    while (lo < hi) {
      int mid = (lo + hi + 1) >> 1;
      if ((in_lds ? s_start[mid] : blk_start[mid]) <= b) lo = mid; else hi = mid - 1;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
  }
  const mvlt_prep_desc d = descs[lo];
  const int lb = b - __builtin_amdgcn_readfirstlane(blk_start[lo]);
  T* out = (T*)d.dst;
  if (d.kind == 0) {
    const int tiles_c = (d.C + TS - 1) / TS;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    const int c0 = (lb % tiles_c) * TS, r0 = (lb / tiles_c) * TS;
    float v[TS / 4];
#pragma unroll
    for (int jj = 0; jj < TS / 4; ++jj) {                       // all 16 loads of the thread in flight before the first LDS write
      int r = r0 + ty + 4 * jj, c = c0 + tx;
      v[jj] = (r < d.R && c < d.C) ? d.src[(long)r * d.C + c] : 0.f;
    }
#pragma unroll
    for (int jj = 0; jj < TS / 4; ++jj) tile[ty + 4 * jj][tx] = v[jj];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < TS / 4; ++jj) {
      int c = c0 + ty + 4 * jj, r = r0 + tx;
      if (c < d.C && r < d.R) out[(long)c * d.ld_out + r] = (T)tile[tx][ty + 4 * jj];
    }
  } else if (d.kind == 2) {
    // transpose of a bf16 source (the compute-dtype copy of the parameters the fused optimizer step writes): 16-byte loads and stores, half the
    // bytes read; C % 8 == 0, ld_out % 8 == 0.  Rows of the tile are 64 + 8 bf16 apart in LDS (144 B: the column reads of eight rows spread over banks).
    if constexpr (sizeof(T) == 2) {
      constexpr int LDT = TS + 8;
      bf16* t16 = (bf16*)&tile[0][0];                            // 64 x 72 bf16 = 9 KB of the 16.6 KB tile
      const bf16* src = (const bf16*)d.src;
      const int tiles_c = (d.C + TS - 1) / TS;
      const int c0 = (lb % tiles_c) * TS, r0 = (lb / tiles_c) * TS;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int q = threadIdx.x + k * NT, rr = q >> 3, cc = (q & 7) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + rr < d.R && c0 + cc < d.C) v = *(const u32x4*)(src + (long)(r0 + rr) * d.C + c0 + cc);
        *(u32x4*)(t16 + rr * LDT + cc) = v;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int q = threadIdx.x + k * NT, cc = q >> 3, rr = (q & 7) * 8;
        if (c0 + cc < d.C && r0 + rr < d.ld_out) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = t16[(rr + e) * LDT + cc];          // rows >= R were loaded as zeros: the padding columns of W^T stay 0
          *(bf16x8*)((bf16*)d.dst + (long)(c0 + cc) * d.ld_out + r0 + rr) = o;
        }
      }
    }
  } else {
    const long i = (long)lb * NT + threadIdx.x;
    const long n = (long)d.d0 * d.d1 * d.d2;
    if (i < n) {
      const int i2 = (int)(i % d.d2);
      const long t = i / d.d2;
      const int i1 = (int)(t % d.d1), i0 = (int)(t / d.d1);
      out[(long)i0 * d.ds0 + (long)i1 * d.ds1 + (long)i2 * d.ds2] =
          (T)d.src[(long)d.src_off + (long)i0 * d.ss0 + (long)i1 * d.ss1 + (long)i2 * d.ss2];
    }
  }
}

RowMap host_rowmap(const mvlt_rowmap* m) {
  RowMap r{};
  if (m) { r.mode = m->mode; r.rows_per_batch = m->rows_per_batch; r.batch_stride = m->batch_stride; r.offset = m->offset; }
  return r;
}

inline int grid_for(long work, int cap = 4096) {
  long g = (work + NT - 1) / NT;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}


// ------------------------------------------------------------------ bilinear resize of a token-major [Hin*Win, C] fp32 map
// PyTorch's upsample_bilinear2d index rule for align_corners=False (what F.interpolate(mode="bilinear") runs at reference
// libs/pvlt.py:295-297 on the learned position embeddings): src = (dst + 0.5) * in/out - 0.5 clamped at 0, neighbours
// floor(src) and min(floor(src)+1, in-1).  ADJ = false: out[(y,x), c] = sum of 4 weighted inputs.  ADJ = true: the adjoint --
// `in` is the gradient w.r.t. the resized map, every element is scattered (atomics) into the gradient of the source map `out`.
template <bool ADJ>
__global__ __launch_bounds__(NT) void resize_tokens_kernel(const float* in, int ld_in, float* out, int ld_out, int Hin, int Win, int Hout, int Wout,
                                                           int C, float sh, float sw) {
  const long n = (long)Hout * Wout * C;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const int c = (int)(i % C);
    const int px = (int)(i / C);
    const int y = px / Wout, x = px - y * Wout;
    const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    if constexpr (!ADJ) {
      out[(long)px * ld_out + c] = w00 * in[(long)(y0 * Win + x0) * ld_in + c] + w01 * in[(long)(y0 * Win + x1) * ld_in + c] +
                                   w10 * in[(long)(y1 * Win + x0) * ld_in + c] + w11 * in[(long)(y1 * Win + x1) * ld_in + c];
    } else {
      const float g = in[(long)px * ld_in + c];
      atomicAdd(&out[(long)(y0 * Win + x0) * ld_out + c], w00 * g);
      atomicAdd(&out[(long)(y0 * Win + x1) * ld_out + c], w01 * g);
      atomicAdd(&out[(long)(y1 * Win + x0) * ld_out + c], w10 * g);
      atomicAdd(&out[(long)(y1 * Win + x1) * ld_out + c], w11 * g);
    }
  }
}

// up to four resizes in one launch (blockIdx.y = which): the four stages' position embeddings at the start of a forward pass, their four adjoints at the end of
// the backward pass (eight 5-us launches per step otherwise, each a drained pipeline between two large kernels)
struct ResizeMulti { const float* in[4]; float* out[4]; int ld_in[4], ld_out[4], Hin[4], Win[4], Hout[4], Wout[4], C[4]; };
template <bool ADJ>
__global__ __launch_bounds__(NT) void resize_tokens_multi_kernel(ResizeMulti a) {
  const int k = blockIdx.y;
  const float* in = a.in[k];
  float* out = a.out[k];
  const int ld_in = a.ld_in[k], ld_out = a.ld_out[k], Hin = a.Hin[k], Win = a.Win[k], Hout = a.Hout[k], Wout = a.Wout[k], C = a.C[k];
  const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
  const long n = (long)Hout * Wout * C;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
    const int c = (int)(i % C);
    const int px = (int)(i / C);
    const int y = px / Wout, x = px - y * Wout;
    const float fy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    if constexpr (!ADJ) {
      out[(long)px * ld_out + c] = w00 * in[(long)(y0 * Win + x0) * ld_in + c] + w01 * in[(long)(y0 * Win + x1) * ld_in + c] +
                                   w10 * in[(long)(y1 * Win + x0) * ld_in + c] + w11 * in[(long)(y1 * Win + x1) * ld_in + c];
    } else {
      const float g = in[(long)px * ld_in + c];
      atomicAdd(&out[(long)(y0 * Win + x0) * ld_out + c], w00 * g);
      atomicAdd(&out[(long)(y0 * Win + x1) * ld_out + c], w01 * g);
      atomicAdd(&out[(long)(y1 * Win + x0) * ld_out + c], w10 * g);
      atomicAdd(&out[(long)(y1 * Win + x1) * ld_out + c], w11 * g);
    }
  }
}

// out = dy * gelu'(h) (exact-erf GELU; BertHeadTransform backward, reference libs/vl_heads.py:13-14,31-32)
template <typename T>
__global__ __launch_bounds__(NT) void gelu_bwd_kernel(const T* dy, const T* h, T* out, long n) {
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT)
    out[i] = (T)((float)dy[i] * gelu_erf_grad((float)h[i]));
}
// out[0] = sum_i w[i] * (*p[i]), out[1 + i] = w[i] * (*p[i]) (0 where p[i] is null): the loss composition of reference
// engine_grid_masking.py:81-102 (mlm + itm + sup_cls + sub_cls + 10 * t2i) in one launch instead of a chain of 0-dim ATen ops
struct LossPtrs { const float* p[5]; float w[5]; };
__global__ void loss_compose_kernel(LossPtrs a, float* out, float* total_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float v = a.p[i] ? a.w[i] * *a.p[i] : 0.f;
    out[1 + i] = v;
    total += v;
  }
  out[0] = total;
  *total_out = total;
}
// dst[c] += sum over rows of in[r][c].  Round 3: 32 workgroups of 1024 threads = (row group, column): every thread strides the slab of its
// workgroup, the row groups meet in LDS, ONE atomic per column and workgroup.  (Round 2: 256 threads = columns walking 64 rows each, 132
// workgroups at stage 1 = 132 same-address atomics per column at ~100 ns: 26 us for 4 MB.)
__global__ __launch_bounds__(1024) void add_column_sums_kernel(const float* in, long rows, int cols, int ld, float* dst0, int n0, float* dst1, int rows_per_wg) {
  __shared__ float part[1024];
  const int rg_n = 1024 / cols > 0 ? 1024 / cols : 1;                 // row groups (cols <= 1024 is checked by the host)
  const int c = threadIdx.x % cols, rg = threadIdx.x / cols;
  const long r0 = (long)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
  float t = 0.f;
  if (rg < rg_n)
  {
    long r = r0 + rg;
    for (; r + 3L * rg_n < r1; r += 4L * rg_n) {       // four independent loads per trip (one per trip waited out a memory round trip each)
      const float a0 = in[r * ld + c], a1 = in[(r + rg_n) * ld + c], a2 = in[(r + 2L * rg_n) * ld + c], a3 = in[(r + 3L * rg_n) * ld + c];
      t += (a0 + a1) + (a2 + a3);
    }
    for (; r < r1; r += rg_n) t += in[r * ld + c];
  }
  part[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x < cols) {
    for (int g = 1; g < rg_n; ++g) t += part[g * cols + threadIdx.x];
    atomicAdd(threadIdx.x < n0 ? dst0 + threadIdx.x : dst1 + (threadIdx.x - n0), t);
  }
}
}  // namespace

extern "C" int mvlt_bert_embed_fwd(const long* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                                   const float* beta, const uint8_t* keep, float drop_p, void* y, float* mean, float* rstd,
                                   int rows, int T, int hidden, float eps, int dtype, void* stream) {
  MVLT_REQUIRE(ids && word && pos && type0 && gamma && beta && y && mean && rstd, "mvlt_bert_embed_fwd: null pointer");
  MVLT_REQUIRE(hidden == 768, "mvlt_bert_embed_fwd: hidden must be 768 (bert-base), got %d", hidden);
  MVLT_REQUIRE(T > 0 && T <= 512, "mvlt_bert_embed_fwd: T must be in (0,512]");
  if (rows <= 0) return MVLT_OK;
  const float inv_keep = 1.0f / (1.0f - drop_p);
  dim3 grid((rows + 3) / 4), block(NT);
  if (dtype == 0) MVLT_LAUNCH((bert_embed_fwd_kernel<bf16, 768>), grid, block, 0, (hipStream_t)stream, ids, word, pos, type0, gamma, beta, keep, inv_keep, (bf16*)y, mean, rstd, rows, T, eps);
  else MVLT_LAUNCH((bert_embed_fwd_kernel<float, 768>), grid, block, 0, (hipStream_t)stream, ids, word, pos, type0, gamma, beta, keep, inv_keep, (float*)y, mean, rstd, rows, T, eps);
  return mvlt_check_launch("mvlt_bert_embed_fwd");
}

extern "C" int mvlt_bert_embed_bwd(const void* dy, const long* ids, const float* word, const float* pos, const float* type0,
                                   const float* gamma, const uint8_t* keep, float drop_p, const float* mean, const float* rstd,
                                   float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                                   int rows, int T, int hidden, int dtype, void* stream) {
  MVLT_REQUIRE(dy && ids && word && pos && type0 && gamma && mean && rstd && dword && dpos && dtype0 && dgamma && dbeta, "mvlt_bert_embed_bwd: null pointer");
  MVLT_REQUIRE(hidden == 768, "mvlt_bert_embed_bwd: hidden must be 768");
  if (rows <= 0) return MVLT_OK;
  const float inv_keep = 1.0f / (1.0f - drop_p);
  MVLT_REQUIRE(rows % T == 0, "mvlt_bert_embed_bwd: rows must be a multiple of T");
  int nsplit = 256 / T; if (nsplit < 1) nsplit = 1;
  const int nbatch = rows / T;
  if (nsplit > (nbatch + 15) / 16) nsplit = (nbatch + 15) / 16;
  dim3 grid(T * nsplit), block(1024);
  if (dtype == 0) MVLT_LAUNCH((bert_embed_bwd_kernel<bf16, 768, 1024>), grid, block, 0, (hipStream_t)stream, (const bf16*)dy, ids, word, pos, type0, gamma, keep, inv_keep, mean, rstd, dword, dpos, dtype0, dgamma, dbeta, rows, T);
  else MVLT_LAUNCH((bert_embed_bwd_kernel<float, 768, 1024>), grid, block, 0, (hipStream_t)stream, (const float*)dy, ids, word, pos, type0, gamma, keep, inv_keep, mean, rstd, dword, dpos, dtype0, dgamma, dbeta, rows, T);
  return mvlt_check_launch("mvlt_bert_embed_bwd");
}

extern "C" int mvlt_patchify(const float* img, void* out, int B, int Cin, int H, int W, int k, int dtype, void* stream) {
  MVLT_REQUIRE(img && out && B > 0 && Cin > 0 && k > 0 && H % k == 0 && W % k == 0, "mvlt_patchify: bad arguments (H, W must be divisible by k)");
  // strip kernel: 4 image rows x 3 channels of a sample in LDS; widths whose strip exceeds the 64 KB default dynamic-LDS limit (W > 1361)
  // and widths that are not whole 16-byte groups take the generic kernel
  if (k == 4 && Cin == 3 && W % 4 == 0 && (size_t)12 * (W + 4) * sizeof(float) <= 65536 && ((uintptr_t)img & 15) == 0 && ((uintptr_t)out & 15) == 0) {
    const size_t lds = (size_t)12 * (W + 4) * sizeof(float);
    dim3 sgrid((unsigned)(B * (H / 4))), sblock(NT);
    if (dtype == 0) MVLT_LAUNCH((patchify_strip_kernel<bf16>), sgrid, sblock, lds, (hipStream_t)stream, img, (bf16*)out, H, W);
    else MVLT_LAUNCH((patchify_strip_kernel<float>), sgrid, sblock, lds, (hipStream_t)stream, img, (float*)out, H, W);
    return mvlt_check_launch("mvlt_patchify");
  }
  long total = (long)B * (H / k) * (W / k) * Cin * k;
  dim3 grid(grid_for(total, 16384)), block(NT);
  if (dtype == 0) MVLT_LAUNCH((patchify_kernel<bf16>), grid, block, 0, (hipStream_t)stream, img, (bf16*)out, B, Cin, H, W, k);
  else MVLT_LAUNCH((patchify_kernel<float>), grid, block, 0, (hipStream_t)stream, img, (float*)out, B, Cin, H, W, k);
  return mvlt_check_launch("mvlt_patchify");
}

extern "C" int mvlt_resize_bilinear_tokens(const float* in, int ld_in, float* out, int ld_out, int Hin, int Win, int Hout, int Wout, int C, int adjoint,
                                           void* stream) {
  MVLT_REQUIRE(in && out && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0 && ld_in >= C && ld_out >= C, "mvlt_resize_bilinear_tokens: bad arguments");
  const long n = (long)Hout * Wout * C;
  const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
  if (adjoint) MVLT_LAUNCH((resize_tokens_kernel<true>), dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, in, ld_in, out, ld_out, Hin, Win, Hout, Wout, C, sh, sw);
  else MVLT_LAUNCH((resize_tokens_kernel<false>), dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, in, ld_in, out, ld_out, Hin, Win, Hout, Wout, C, sh, sw);
  return mvlt_check_launch("mvlt_resize_bilinear_tokens");
}

extern "C" int mvlt_resize_bilinear_tokens_multi(const float* const* in, const int* ld_in, float* const* out, const int* ld_out, const int* Hin, const int* Win,
                                                 const int* Hout, const int* Wout, const int* C, int count, int adjoint, void* stream) {
  MVLT_REQUIRE(in && out && ld_in && ld_out && Hin && Win && Hout && Wout && C && count >= 1 && count <= 4, "mvlt_resize_bilinear_tokens_multi: 1..4 resizes per call");
  ResizeMulti a{};
  long nmax = 0;
  for (int k = 0; k < count; ++k) {
    MVLT_REQUIRE(in[k] && out[k] && Hin[k] > 0 && Win[k] > 0 && Hout[k] > 0 && Wout[k] > 0 && C[k] > 0 && ld_in[k] >= C[k] && ld_out[k] >= C[k],
                 "mvlt_resize_bilinear_tokens_multi: bad arguments");
    a.in[k] = in[k]; a.out[k] = out[k]; a.ld_in[k] = ld_in[k]; a.ld_out[k] = ld_out[k];
    a.Hin[k] = Hin[k]; a.Win[k] = Win[k]; a.Hout[k] = Hout[k]; a.Wout[k] = Wout[k]; a.C[k] = C[k];
    const long n = (long)Hout[k] * Wout[k] * C[k];
    if (n > nmax) nmax = n;
  }
  const dim3 grid(grid_for(nmax), (unsigned)count);
  if (adjoint) MVLT_LAUNCH((resize_tokens_multi_kernel<true>), grid, dim3(NT), 0, (hipStream_t)stream, a);
  else MVLT_LAUNCH((resize_tokens_multi_kernel<false>), grid, dim3(NT), 0, (hipStream_t)stream, a);
  return mvlt_check_launch("mvlt_resize_bilinear_tokens_multi");
}

extern "C" int mvlt_gelu_bwd(const void* dy, const void* h, void* out, long n, int dtype, void* stream) {
  MVLT_REQUIRE(dy && h && out && n >= 0 && (dtype == 0 || dtype == 1), "mvlt_gelu_bwd: bad arguments");
  if (n == 0) return MVLT_OK;
  if (dtype == 0) MVLT_LAUNCH((gelu_bwd_kernel<bf16>), dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)h, (bf16*)out, n);
  else MVLT_LAUNCH((gelu_bwd_kernel<float>), dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, (const float*)dy, (const float*)h, (float*)out, n);
  return mvlt_check_launch("mvlt_gelu_bwd");
}

extern "C" int mvlt_masked_select(const long* labels, int n, long ignore_index, int* idx, int* count, void* stream) {
  MVLT_REQUIRE(labels && idx && count && n >= 0, "mvlt_masked_select: bad arguments");
  if (n <= 65536) MVLT_LAUNCH(masked_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, n, ignore_index, idx, count);
  else MVLT_LAUNCH(masked_select_loop_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, n, ignore_index, idx, count);
  return mvlt_check_launch("mvlt_masked_select");
}

extern "C" int mvlt_gather_rows(const void* src, const int* idx, void* dst, int rows, int C, int ld_src, const mvlt_rowmap* src_map, int dtype, void* stream) {
  MVLT_REQUIRE(src && idx && dst, "mvlt_gather_rows: null pointer");
  const int pc = dtype == 0 ? 8 : 4;
  MVLT_REQUIRE(C % pc == 0 && ld_src % pc == 0, "mvlt_gather_rows: C/ld must be multiples of %d", pc);
  if (rows <= 0) return MVLT_OK;
  dim3 grid(grid_for((long)rows * (C / pc))), block(NT);
  RowMap m = host_rowmap(src_map);
  if (dtype == 0) MVLT_LAUNCH((gather_rows_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16*)src, idx, (bf16*)dst, rows, C, ld_src, m);
  else MVLT_LAUNCH((gather_rows_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)src, idx, (float*)dst, rows, C, ld_src, m);
  return mvlt_check_launch("mvlt_gather_rows");
}

extern "C" int mvlt_scatter_rows(const void* src, const int* idx, void* dst, int rows, int C, int ld_dst, const mvlt_rowmap* dst_map, int accumulate, int dtype, void* stream) {
  MVLT_REQUIRE(src && idx && dst, "mvlt_scatter_rows: null pointer");
  const int pc = dtype == 0 ? 8 : 4;
  MVLT_REQUIRE(C % pc == 0 && ld_dst % pc == 0, "mvlt_scatter_rows: C/ld must be multiples of %d", pc);
  if (rows <= 0) return MVLT_OK;
  dim3 grid(grid_for((long)rows * (C / pc))), block(NT);
  RowMap m = host_rowmap(dst_map);
  if (dtype == 0) MVLT_LAUNCH((scatter_rows_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16*)src, idx, (bf16*)dst, rows, C, ld_dst, m, accumulate);
  else MVLT_LAUNCH((scatter_rows_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)src, idx, (float*)dst, rows, C, ld_dst, m, accumulate);
  return mvlt_check_launch("mvlt_scatter_rows");
}

extern "C" int mvlt_cross_entropy_fwd(const void* logits, const long* labels, long ignore_index, float* lse, float* loss_sum, float* count,
                                      int rows, int V, int ld, int dtype, void* stream) {
  MVLT_REQUIRE(logits && labels && lse && loss_sum && count && V > 0 && ld >= V, "mvlt_cross_entropy_fwd: bad arguments");
  if (rows <= 0) return MVLT_OK;
  dim3 grid(rows), block(NT);
  if (dtype == 0) {
    MVLT_LAUNCH((ce_fwd_kernel<bf16>), grid, block, 0, (hipStream_t)stream, (const bf16*)logits, labels, ignore_index, lse, loss_sum, count, rows, V, ld);
    MVLT_LAUNCH((ce_loss_finish_kernel<bf16>), dim3(1), dim3(1024), 0, (hipStream_t)stream, (const bf16*)logits, labels, ignore_index, (const float*)lse, loss_sum, count, rows, ld);
  } else {
    MVLT_LAUNCH((ce_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)logits, labels, ignore_index, lse, loss_sum, count, rows, V, ld);
    MVLT_LAUNCH((ce_loss_finish_kernel<float>), dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)logits, labels, ignore_index, (const float*)lse, loss_sum, count, rows, ld);
  }
  return mvlt_check_launch("mvlt_cross_entropy_fwd");
}

extern "C" int mvlt_cross_entropy_bwd(const void* logits, const long* labels, long ignore_index, const float* lse, const float* gscale,
                                      const float* count, void* dlogits, int rows, int V, int ld, int ldd, int dtype, int out_dtype, void* stream) {
  MVLT_REQUIRE(logits && labels && lse && gscale && count && dlogits && V > 0 && ld >= V && ldd >= V, "mvlt_cross_entropy_bwd: bad arguments");
  if (rows <= 0) return MVLT_OK;
  dim3 grid(rows), block(NT);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 0 && out_dtype == 0) MVLT_LAUNCH((ce_bwd_kernel<bf16, bf16>), grid, block, 0, s, (const bf16*)logits, labels, ignore_index, lse, gscale, count, (bf16*)dlogits, rows, V, ld, ldd);
  else if (dtype == 1 && out_dtype == 0) MVLT_LAUNCH((ce_bwd_kernel<float, bf16>), grid, block, 0, s, (const float*)logits, labels, ignore_index, lse, gscale, count, (bf16*)dlogits, rows, V, ld, ldd);
  else if (dtype == 1 && out_dtype == 1) MVLT_LAUNCH((ce_bwd_kernel<float, float>), grid, block, 0, s, (const float*)logits, labels, ignore_index, lse, gscale, count, (float*)dlogits, rows, V, ld, ldd);
  else MVLT_LAUNCH((ce_bwd_kernel<bf16, float>), grid, block, 0, s, (const bf16*)logits, labels, ignore_index, lse, gscale, count, (float*)dlogits, rows, V, ld, ldd);
  return mvlt_check_launch("mvlt_cross_entropy_bwd");
}

extern "C" int mvlt_smooth_l1_fwd(const float* pred, const float* target, long n, float* loss_sum, void* stream) {
  MVLT_REQUIRE(pred && target && loss_sum && n > 0 && n % 4 == 0 && (((uintptr_t)pred | (uintptr_t)target) & 15) == 0,
               "mvlt_smooth_l1_fwd: bad arguments (n multiple of 4, 16-byte aligned)");
  MVLT_LAUNCH(smooth_l1_sum_kernel, dim3(grid_for(n / 4, 2048)), dim3(NT), 0, (hipStream_t)stream, pred, target, n, loss_sum);
  return mvlt_check_launch("mvlt_smooth_l1_fwd");
}

extern "C" int mvlt_smooth_l1_bwd(const float* pred, const float* target, long n, const float* gscale, float* grad, void* stream) {
  MVLT_REQUIRE(pred && target && gscale && grad && n > 0 && n % 4 == 0 && (((uintptr_t)pred | (uintptr_t)target | (uintptr_t)grad) & 15) == 0,
               "mvlt_smooth_l1_bwd: bad arguments (n multiple of 4, 16-byte aligned)");
  MVLT_LAUNCH(smooth_l1_grad_kernel, dim3(grid_for(n / 4, 8192)), dim3(NT), 0, (hipStream_t)stream, pred, target, n, gscale, 1.0f / (float)n, grad);
  return mvlt_check_launch("mvlt_smooth_l1_bwd");
}

extern "C" int mvlt_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, const float* hp,
                               const uint8_t* decay_mask, void* stream) {
  MVLT_REQUIRE(p && g && m && v && hp && n >= 0 && n % 4 == 0, "mvlt_adamw_step: bad arguments (n must be a multiple of 4)");
  if (n == 0) return MVLT_OK;
  MVLT_LAUNCH(adamw_kernel, dim3(grid_for(n / 4, 8192)), dim3(NT), 0, (hipStream_t)stream, p, g, m, v, (bf16*)p_bf16, n, hp, decay_mask);
  return mvlt_check_launch("mvlt_adamw_step");
}

extern "C" int mvlt_cast_bf16(const float* src, void* dst, long n, void* stream) {
  MVLT_REQUIRE(src && dst && n >= 0 && n % 4 == 0, "mvlt_cast_bf16: bad arguments (n must be a multiple of 4)");
  if (n == 0) return MVLT_OK;
  MVLT_LAUNCH(cast_f32_bf16_kernel, dim3(grid_for(n / 4, 8192)), dim3(NT), 0, (hipStream_t)stream, src, (bf16*)dst, n);
  return mvlt_check_launch("mvlt_cast_bf16");
}

extern "C" int mvlt_row_scale(const void* x, const float* scale, int rows_per_scale, long M, int C, void* out, int dtype, void* stream) {
  MVLT_REQUIRE(x && scale && out && rows_per_scale > 0 && M >= 0 && C > 0 && C % 8 == 0, "mvlt_row_scale: bad arguments (C must be a multiple of 8)");
  MVLT_REQUIRE(dtype == 0 || dtype == 1, "mvlt_row_scale: bad dtype");
  if (M == 0) return MVLT_OK;
  const long n = M * C, per = (long)rows_per_scale * C;
  if (dtype == 0) MVLT_LAUNCH((row_scale_kernel<bf16>), dim3(grid_for(n / 8, 8192)), dim3(NT), 0, (hipStream_t)stream, (const bf16*)x, scale, per, n, (bf16*)out);
  else MVLT_LAUNCH((row_scale_kernel<float>), dim3(grid_for(n / 4, 8192)), dim3(NT), 0, (hipStream_t)stream, (const float*)x, scale, per, n, (float*)out);
  return mvlt_check_launch("mvlt_row_scale");
}

extern "C" int mvlt_head_grad_prep(const float* dlogits, int B, int n, int n_pad, void* dl, float* db1, float* db2, int dtype, void* stream) {
  MVLT_REQUIRE(dlogits && dl && db1 && B > 0 && n > 0 && n_pad >= n && n_pad <= 256 && (dtype == 0 || dtype == 1), "mvlt_head_grad_prep: bad arguments (n_pad <= 256)");
  if (dtype == 0) MVLT_LAUNCH((head_grad_prep_kernel<bf16>), dim3(1), dim3(NT), 0, (hipStream_t)stream, dlogits, B, n, n_pad, (bf16*)dl, db1, db2);
  else MVLT_LAUNCH((head_grad_prep_kernel<float>), dim3(1), dim3(NT), 0, (hipStream_t)stream, dlogits, B, n, n_pad, (float*)dl, db1, db2);
  return mvlt_check_launch("mvlt_head_grad_prep");
}

extern "C" int mvlt_weight_prep(const mvlt_prep_desc* descs, const int* blk_start, int ndesc, int total_blocks, const int* blk_desc, int dtype, void* stream) {
  MVLT_REQUIRE(descs && blk_start && ndesc > 0 && total_blocks > 0, "mvlt_weight_prep: bad arguments");
  if (dtype == 0) MVLT_LAUNCH((weight_prep_kernel<bf16>), dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, descs, blk_start, ndesc, blk_desc);
  else MVLT_LAUNCH((weight_prep_kernel<float>), dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, descs, blk_start, ndesc, blk_desc);
  return mvlt_check_launch("mvlt_weight_prep");
}

extern "C" int mvlt_transpose_cast(const float* in, void* out, int R, int Ccols, int ld_out, int dtype, void* stream) {
  MVLT_REQUIRE(in && out && R > 0 && Ccols > 0 && ld_out >= R, "mvlt_transpose_cast: bad arguments");
  dim3 grid((Ccols + 31) / 32, (R + 31) / 32), block(NT);
  if (dtype == 0) MVLT_LAUNCH((transpose_cast_kernel<bf16>), grid, block, 0, (hipStream_t)stream, in, (bf16*)out, R, Ccols, ld_out);
  else MVLT_LAUNCH((transpose_cast_kernel<float>), grid, block, 0, (hipStream_t)stream, in, (float*)out, R, Ccols, ld_out);
  return mvlt_check_launch("mvlt_transpose_cast");
}

extern "C" int mvlt_loss_compose(const float* const* losses, const float* weights, float* out, float* total, void* stream) {
  MVLT_REQUIRE(losses && weights && out && total, "mvlt_loss_compose: null argument");
  LossPtrs a;
  for (int i = 0; i < 5; ++i) { a.p[i] = losses[i]; a.w[i] = weights[i]; }
  MVLT_LAUNCH(loss_compose_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out, total);
  return mvlt_check_launch("mvlt_loss_compose");
}

extern "C" int mvlt_add_column_sums(const float* in, long rows, int cols, int ld, float* dst0, int n0, float* dst1, void* stream) {
  MVLT_REQUIRE(in && dst0 && rows >= 0 && cols > 0 && ld >= cols && n0 >= 0 && n0 <= cols && (n0 == cols || dst1), "mvlt_add_column_sums: bad arguments");
  if (rows == 0) return MVLT_OK;
  MVLT_REQUIRE(cols <= 1024, "mvlt_add_column_sums: at most 1024 columns, got %d", cols);
  long nwg = rows / 64 < 32 ? (rows + 63) / 64 : 32;
  const int rows_per_wg = (int)((rows + nwg - 1) / nwg);
  MVLT_LAUNCH(add_column_sums_kernel, dim3((unsigned)((rows + rows_per_wg - 1) / rows_per_wg)), dim3(1024), 0, (hipStream_t)stream, in, rows, cols, ld, dst0, n0,
                     dst1, rows_per_wg);
  return mvlt_check_launch("mvlt_add_column_sums");
}
