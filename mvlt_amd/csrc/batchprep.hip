// Device-side batch preparation: what the reference's dataset does per sample on the host before a batch
// reaches the engine -- grid-mask generation + masked_fill(1e-6) (mcloader/fashion_gen.py:176,225-254) and BERT-style token
// masking (:383-409) -- as HBM-bound integer / byte kernels on a counter-based generator (Philox4x32-10), so that a sample's
// masks depend on (seed, sample id) only and oracle/batchprep_oracle.py can restate them bit for bit.
#include "common.h"
#include "../../include/mvlt_hip.h"

namespace {
constexpr int NT = 256;

struct u4 { uint32_t x, y, z, w; };

// Philox4x32-10 (Salmon et al., SC'11; Random123 constants).  counter = (element, sample_lo, stream, sample_hi), key = seed.
__device__ __forceinline__ u4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return u4{c0, c1, c2, c3};
}
__device__ __forceinline__ u4 draws(uint64_t seed, uint64_t sample, uint32_t element, uint32_t stream) {
  return philox(element, (uint32_t)sample, stream, (uint32_t)(sample >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
}

constexpr int MAXP = 4096;        // patches per sample (1024 px / 16 squared)

// One workgroup per sample: patch flags (1 = masked) into flags[b, gh*gw].  Ranks come from an all-pairs comparison of the
// 32-bit keys held in LDS (P <= 4096: at most 16 M compares per sample, 65 k at 256 px) -- no sort, no atomics.
__global__ __launch_bounds__(NT) void grid_flags_kernel(uint8_t* flags, int gh, int gw, int num_mask, int mode, uint64_t seed, uint64_t sample0) {
  __shared__ uint32_t key[MAXP];
  __shared__ uint8_t shuf[MAXP];
  const int P = gh * gw, b = blockIdx.x;
  const uint64_t sample = sample0 + b;
  for (int p = threadIdx.x; p < P; p += NT) key[p] = draws(seed, sample, p, 0).x;
  __syncthreads();
  uint8_t* out = flags + (size_t)b * P;
  for (int e = threadIdx.x; e < P; e += NT) {
    const uint32_t k = key[e];
    int rank = 0;
    for (int q = 0; q < P; ++q) rank += (key[q] < k) || (key[q] == k && q < e);
    if (mode == 0) out[e] = rank < num_mask;                      // the num_mask smallest keys
    else shuf[rank] = e >= P - num_mask;                          // element e of [0]*(P-n) + [1]*n lands at position rank
  }
  if (mode == 0) return;
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += NT) key[p] = draws(seed, sample, p, 1).x;       // second shuffle, within each patch row
  __syncthreads();
  for (int p = threadIdx.x; p < P; p += NT) {
    const int i = p / gw, c = p - i * gw;
    const uint32_t k = key[p];
    int rank = 0;
    for (int q = 0; q < gw; ++q) { const uint32_t kq = key[i * gw + q]; rank += (kq < k) || (kq == k && q < c); }
    out[i * gw + rank] = shuf[i + c];                              // row i re-shuffles the WINDOW shuffled[i : i+gw] (reference quirk)
  }
}

// masked[b,c,y,x] = flags[b, y/patch, x/patch] ? fill : image[b,c,y,x]   (16-byte accesses; W % 4 == 0, patch % 4 == 0)
__global__ __launch_bounds__(NT) void grid_apply_kernel(const float* image, const uint8_t* flags, float* masked, long n4, int C, int H, int W,
                                                        int patch, float fill) {
  const int gw = W / patch, gh = H / patch, W4 = W / 4;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
    const int x4 = (int)(i % W4);
    const long r = i / W4;                 // (b*C + c)*H + y
    const int y = (int)(r % H);
    const long b = r / H / C;
    const f32x4 v = *(const f32x4*)(image + i * 4);
    const bool m = flags[b * gh * gw + (y / patch) * gw + (x4 * 4) / patch];
    *(f32x4*)(masked + i * 4) = m ? f32x4{fill, fill, fill, fill} : v;
  }
}

// BERT-style masking of one caption position per thread (integer decisions on 24-bit draws: bit-exact on any host)
__global__ __launch_bounds__(NT) void token_mask_kernel(const long* ori, long* ids, long* labels, long n, int T, uint64_t seed, uint64_t sample0,
                                                        int vocab, int t15, int t80, int t90) {
  const long i = (long)blockIdx.x * NT + threadIdx.x;
  if (i >= n) return;
  const long b = i / T;
  const int t = (int)(i - b * T);
  const long id = ori[i];
  const u4 d = draws(seed, sample0 + b, t, 2);
  const uint32_t r1 = d.x >> 8, r2 = d.y >> 8;
  const bool cand = t >= 1 && id != 0 && id != 101 && id != 102;
  const bool sel = cand && r1 < (uint32_t)t15;
  long out = id;
  if (sel && r2 < (uint32_t)t80) out = 103;
  else if (sel && r2 < (uint32_t)t90) out = (long)(((uint64_t)d.z * (uint64_t)vocab) >> 32);
  ids[i] = out;
  labels[i] = sel ? id : -1;
}
// ---- train-mode masks of the step itself (nn.Dropout inside BertEmbeddings, timm DropPath per block; reference libs/pvlt.py:135,233),
// from the same counter-based generator: one launch each instead of ATen's rand / compare / cast / divide chains.
// keep[i] = draw_i >= drop_p on 16-bit draws (eight per Philox call, one 8-byte store per thread); counter = (i / 8, call, 4, call >> 32)
// (streams 0-2 belong to the per-sample dataset masks; a stream of its own keeps the dropout draws of call n apart from the token-masking
// draws of sample id n when both generators run under the same seed).
__global__ __launch_bounds__(NT) void keep_mask_kernel(uint8_t* keep, long n, uint32_t thr16, uint64_t seed, uint64_t call) {
  const long g = (long)blockIdx.x * NT + threadIdx.x;
  if (g * 8 >= n) return;
  const u4 d = draws(seed, call, (uint32_t)g, 4);
  const uint32_t w[4] = {d.x, d.y, d.z, d.w};
  uint8_t k[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) k[e] = ((w[e >> 1] >> (16 * (e & 1))) & 0xFFFFu) >= thr16;
  if (g * 8 + 8 <= n) {
    uint64_t v = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) v |= (uint64_t)k[e] << (8 * e);
    *(uint64_t*)(keep + g * 8) = v;
  } else {
    for (int e = 0; g * 8 + e < n; ++e) keep[g * 8 + e] = k[e];
  }
}
// out[r][j] = (draw >= rate[r]) / (1 - rate[r]) on 24-bit draws: DropPath's per-sample keep factor (timm drop_path: x / keep_prob * mask);
// counter = (r * per + j, call, 5, call >> 32)
__global__ __launch_bounds__(NT) void droppath_scales_kernel(float* out, const float* rates, int nrate, int per, uint64_t seed, uint64_t call) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= nrate * per) return;
  const float rate = rates[i / per];
  const uint32_t thr = (uint32_t)(rate * 16777216.0f);
  const uint32_t d = draws(seed, call, (uint32_t)i, 5).x >> 8;
  out[i] = d >= thr ? 1.0f / (1.0f - rate) : 0.0f;
}

}  // namespace

extern "C" int mvlt_grid_mask_flags(uint8_t* flags, int B, int gh, int gw, int num_mask, int mode, uint64_t seed, uint64_t sample0, void* stream) {
  MVLT_REQUIRE(flags && B >= 0 && gh > 0 && gw > 0 && gh * gw <= MAXP, "mvlt_grid_mask_flags: bad arguments (at most %d patches per sample)", MAXP);
  MVLT_REQUIRE(num_mask >= 0 && num_mask <= gh * gw && (mode == 0 || mode == 1), "mvlt_grid_mask_flags: bad num_mask / mode");
  if (B == 0) return MVLT_OK;
  MVLT_LAUNCH(grid_flags_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, flags, gh, gw, num_mask, mode, seed, sample0);
  return mvlt_check_launch("mvlt_grid_mask_flags");
}

extern "C" int mvlt_grid_mask_apply(const float* image, const uint8_t* flags, float* masked, int B, int C, int H, int W, int patch, float fill,
                                    void* stream) {
  MVLT_REQUIRE(image && flags && masked && B >= 0 && C > 0 && H > 0 && W > 0 && patch > 0, "mvlt_grid_mask_apply: bad arguments");
  MVLT_REQUIRE(H % patch == 0 && W % patch == 0 && patch % 4 == 0, "mvlt_grid_mask_apply: H, W must be multiples of patch, patch of 4");
  const long n4 = (long)B * C * H * W / 4;
  if (n4 == 0) return MVLT_OK;
  const long blocks = (n4 + NT - 1) / NT;
  MVLT_LAUNCH(grid_apply_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(NT), 0, (hipStream_t)stream, image, flags, masked, n4,
                     C, H, W, patch, fill);
  return mvlt_check_launch("mvlt_grid_mask_apply");
}

extern "C" int mvlt_token_mask(const long* ori_ids, long* input_ids, long* labels, int B, int T, uint64_t seed, uint64_t sample0, int vocab,
                               void* stream) {
  MVLT_REQUIRE(ori_ids && input_ids && labels && B >= 0 && T > 0 && vocab > 0, "mvlt_token_mask: bad arguments");
  const long n = (long)B * T;
  if (n == 0) return MVLT_OK;
  // r < T  <=>  r / 2^24 < p for p = 0.15 / 0.8 / 0.9 (fashion_gen.py:390-398)
  MVLT_LAUNCH(token_mask_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, ori_ids, input_ids, labels, n, T, seed,
                     sample0, vocab, 2516583, 13421773, 15099495);
  return mvlt_check_launch("mvlt_token_mask");
}

extern "C" int mvlt_keep_mask(uint8_t* keep, long n, float drop_p, uint64_t seed, uint64_t call, void* stream) {
  MVLT_REQUIRE(keep && n >= 0 && drop_p >= 0.f && drop_p < 1.f && ((uintptr_t)keep & 7) == 0, "mvlt_keep_mask: bad arguments");
  if (n == 0) return MVLT_OK;
  const long groups = (n + 7) / 8;
  MVLT_LAUNCH(keep_mask_kernel, dim3((unsigned)((groups + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, keep, n, (uint32_t)(drop_p * 65536.0f), seed,
                     call);
  return mvlt_check_launch("mvlt_keep_mask");
}

extern "C" int mvlt_droppath_scales(float* out, const float* rates, int nrate, int per, uint64_t seed, uint64_t call, void* stream) {
  MVLT_REQUIRE(out && rates && nrate >= 0 && per >= 0, "mvlt_droppath_scales: bad arguments");
  if (nrate * per == 0) return MVLT_OK;
  MVLT_LAUNCH(droppath_scales_kernel, dim3((unsigned)((nrate * per + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, out, rates, nrate, per, seed, call);
  return mvlt_check_launch("mvlt_droppath_scales");
}
