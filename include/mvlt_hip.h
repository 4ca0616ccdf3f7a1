/* mvlt_hip.h -- C ABI of libmvlt_hip.so: the MI355X (gfx950) kernels behind the MVLT hot path.
 *
 * The reference (GewelsJI/MVLT) is pure PyTorch: it has no FFI of its own.  Each entry point below names the
 * reference call site(s) whose ATen kernels it replaces; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - the caller owns every buffer (device pointers, contiguous in the stated layout); kernels never allocate
 *   - everything is asynchronous on the hipStream_t passed as `void* stream` (NULL = default stream); no syncs
 *   - returns 0 on success, <0 on error; mvlt_last_error() returns a thread-local description
 *   - dtype codes: 0 = bf16, 1 = fp32.  All reductions/accumulations are fp32.
 */
#ifndef MVLT_HIP_H
#define MVLT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MVLT_DT_BF16 0
#define MVLT_DT_FP32 1

/* Version of this header: bumped whenever an exported signature or an argument struct changes.  mvlt_abi_version() returns the number the library was
 * built with; a binding compares it with the number it was written against (mvlt_amd/_lib.py ABI_VERSION) before any other call.
 *   2 (round 5): positional signatures of mvlt_batch_sum / mvlt_bn_norm / mvlt_bn_bwd_reduce / mvlt_bn_bwd_apply / mvlt_ew_mul / mvlt_ew_mul3_bwd as of
 *                round 4's second half (the number had stayed 1 through those changes: ADVICE r4)
 *   3 (round 5): mvlt_gemm_tn_args.partials / partials_bytes; mvlt_last_kernel()
 *   4: mvlt_weight_prep blk_desc      5: mvlt_gemm_tn_args.defer_fold, mvlt_tn_fold_flush(), mvlt_tn_fold_discard()
 *   6 (round 6): mvlt_tn_fold_flush(partials, stream) / mvlt_tn_fold_discard(partials): the pending-fold table is kept per scratch (= per owner);
 *                mvlt_sr_attention_bwd_chunks(); mvlt_gemm_tn_args.c_overwrite; mvlt_mlp_args.partials / partials_bytes / defer_fold */
#define MVLT_ABI_VERSION 6
const char* mvlt_last_error(void);
int mvlt_abi_version(void);
/* the kernel instantiation the library launched last on the calling thread, as the HIP runtime names it, demangled (e.g. "void (anonymous
 * namespace)::mlp_wgrad2_kernel<64, 4>(mvlt_mlp_args, int, int, int)"); "" before the first launch.  For measurement records (bench.py names its roofline launches with it). */
const char* mvlt_last_kernel(void);
/* sizeof(struct <name>) for binding self-checks; -1 if unknown */
int mvlt_sizeof(const char* name);

/* Row addressing of a GEMM operand / result.
 *  mode 0: phys_row = m                                   when rows_per_batch == 0
 *          phys_row = (m / rows_per_batch) * batch_stride + offset + m % rows_per_batch    otherwise
 *          (a token sub-range of a (B, N, C) buffer: image tokens [0,HW) or text tokens [HW,HW+T))
 *  mode 1: r x r non-overlapping patch gather over the image-token grid of a (B, tokens_in, c_seg) buffer:
 *          logical row m = (b, oi, oj), logical column = (di*r + dj) * c_seg + c,
 *          phys_row = b*tokens_in + (oi*r+di)*w_in + (oj*r+dj).  This is nn.Conv2d(kernel=stride=r) on
 *          token-major data (reference libs/pvlt.py:92,104 Attention.sr and :162,168 PatchEmbed.proj);
 *          the conv weight is used as [out][di][dj][c].
 *  mode 2: 3x3 / stride 1 / zero-padded neighbourhood gather over an h_in x w_in pixel grid of a (B, tokens_in, c_seg)
 *          buffer (hw_out = h_in*w_in, w_out = w_in): logical row m = (b, y, x), logical column = (dy*3+dx)*c_seg + c,
 *          phys_row = b*tokens_in + (y+dy-1)*w_in + (x+dx-1), zero outside the grid.  This is nn.Conv2d(3, padding=1)
 *          of the MIM decoder on pixel-major data (reference libs/vl_heads.py:116-129,148-152); weight as [out][dy][dx][c]. */
typedef struct mvlt_rowmap {
  int mode, rows_per_batch, batch_stride, offset;
  int r, w_in, tokens_in, hw_out, w_out, c_seg;
  int h_in;
} mvlt_rowmap;

/* C[M,N] = epi(A[M,K] . B[N,K]^T).   Replaces nn.Linear forward (reference libs/pvlt.py:66,69,98,108,118;
 * libs/vl_heads.py:31,67,85,102), its input gradient (B = W^T), and the kernel==stride convolutions.
 *   epi(v) = act(v + bias[n]) * row_scale[m / rows_per_scale] + R[m,n]
 *   act 0: identity; 1: exact-erf GELU (pre-activation optionally stored to H); 2: v * gelu'(H[m,n])
 *   R may alias C (accumulate).  C, R, H share c_map/ldc and out_dtype. */
typedef struct mvlt_gemm_nt_args {
  const void* A; const void* B; void* C;
  int M, N, K, lda, ldb, ldc;
  int dtype;                 /* of A and B */
  int out_dtype;             /* of C, R, H: 0 bf16, 1 fp32; 2 = fp16 C, with col_sum only (the conv output z that BatchNorm normalises) */
  mvlt_rowmap a_map, c_map;
  const float* bias;         /* [N] fp32 or NULL */
  int act;
  void* H;
  const float* row_scale;    /* [ceil(M / rows_per_scale)] fp32 or NULL (DropPath keep/(1-p) per sample) */
  int rows_per_scale;
  const void* R;
  float* col_sum;            /* optional [N] += sum over rows of the stored C values, and */
  float* col_sumsq;          /* [N] += sum of their squares (fp32 atomics into caller-zeroed buffers): the batch statistics
                                of BatchNorm2d behind a conv (reference libs/vl_heads.py:107-120 BasicConv2d) without a
                                second pass over the conv output */
  int col_copies;            /* 0 / 1: one accumulator; k > 1: col_sum / col_sumsq are [k][N] and row tile t adds into copy
                                t % k (spreads the same-address atomics; mvlt_bn_finalize sums the copies) */
  int split_k;               /* 0 / 1: off; s > 1: K is cut into s ranges handled by separate workgroups that add their partial
                                tiles into the fp32, caller-zeroed C with atomics (few output tiles, long K: the input
                                gradient of the tied 30522-word MLM decoder, reference libs/vl_heads.py:31-36).  bf16
                                operands, plain epilogue (bias allowed), identity row maps */
  /* optional LayerNorm of the finished output row while it is on chip (bf16 operands, R given, N == 64 or 128 so that one tile holds
   * whole rows, identity row maps): post_y[M, post_ld] (bf16) = LN(C row; post_gamma, post_beta, post_eps), statistics to post_mean /
   * post_rstd [M].  attn.proj + DropPath + residual followed by Block.norm2 (reference libs/pvlt.py:140-142): the fused MLP then reads
   * its operand in bf16 instead of normalising the fp32 mid stream itself. */
  void* post_y; int post_ld;
  const float* post_gamma; const float* post_beta; float post_eps;
  float* post_mean; float* post_rstd;
  /* 1: R is fp32 while C (out_dtype 0) is bf16 -- the last block of a stage adds the fp32 residual stream and hands the next stage / the heads
   * the MFMA-operand copy directly (no fp32 stage output + cast pass); 0: R has C's dtype */
  int r_fp32;
} mvlt_gemm_nt_args;
int mvlt_gemm_nt(const mvlt_gemm_nt_args* args, void* stream);

/* C[N1,N2] += A[M,N1]^T . B[M,N2]  (fp32, atomic accumulate into a caller-zeroed buffer) and optionally
 * colsum_a[N1] += sum_m A[m,:] (or colsum_b over B).   Weight / bias gradients of nn.Linear and of the kernel==stride convs
 * (autograd of the call sites above).  b_map may be a patch gather (N2 = r*r*c_seg). */
typedef struct mvlt_gemm_tn_args {
  const void* A; const void* B; float* C;
  int M, N1, N2, lda, ldb, ldc;
  int dtype;
  mvlt_rowmap a_map, b_map;
  float* colsum_a;
  int splits;                /* 0 = choose */
  float* colsum_b;           /* optional [N2] += sum_m B[m,:] (at most one of colsum_a / colsum_b) */
  int trans_c;               /* store C transposed: C[n2*ldc + n1] (lets the narrow operand take the 64-wide tile side) */
  int c_taps, c_seg;         /* c_taps > 1: logical column n2 = tap*c_seg + c is stored at column c*c_taps + tap -- a conv weight
                              * gradient computed in the gather's [out][tap][cin] order lands in nn.Conv2d's [out][cin][kh][kw]
                              * layout directly (N2 == c_taps*c_seg, trans_c == 0) */
  /* optional: the INPUT gradient of the same nn.Linear from the same pass over A (= dY): dgrad_out[M, N2] (bf16, row stride dgrad_ld) =
   * A[M, N1] . W[N1, N2], dgrad_wt = W^T [N2][N1] bf16 (the `::T` operand copy, row stride N1).  bf16, plain rows on both operands,
   * N1 == N2 == 64 or 128, trans_c == 0: the q / proj projections of stages 1-2 (reference libs/pvlt.py:98,118 and their autograd) read dY
   * once for both gradients instead of once per gradient (138 MB per Linear at stage 1). */
  const void* dgrad_wt; void* dgrad_out; int dgrad_ld;
  /* optional scratch for split reductions WITHOUT atomics (round 5): every m-split stores its output tile to partials[split][N1][N2] in bf16 and an ordered fold
   * (tn_fold_kernel) adds the splits to C -- deterministic (bit-identical from launch to launch), one bf16 rounding per split's partial sum (max-norm error 2e-3 of the
   * gradient instead of 1e-5).  Taken by (a) outputs of 16 .. 64 whole 256 x 256 tiles with M a multiple of 64 (the fc1 / fc2 weight gradients of a stage-4 block, 2048 x 512
   * over 49152 rows: 8-wave / 8-phase TN kernel, 114 against 132 us), (b) the 128-wide kernel when >= 8 m-splits meet on an output of >= 65536 elements (q / proj / kv
   * weight gradients of stages 3-4: 40-45 against 53-56 us), (c) the conv3x3 weight-gradient kernel with >= 4 m-splits; bf16 operands, plain rows (b_map mode 2 for (c)),
   * trans_c == 0, c_taps <= 1.  A launch whose splits x N1 x N2 x 2 bytes exceed partials_bytes, or partials == NULL, takes the atomic path.  Sizing: the largest single
   * launch of the BASELINE configurations needs 37 MiB (56 splits x 192 x 1728); with defer_fold several launches SHARE the scratch -- each takes the next free region, the
   * library folds by itself when the next one does not fit -- so the host side holds 256 MiB per parameter store (MVLT_TN_SCRATCH_MIB): ~4 fold launches per step. */
  void* partials; long partials_bytes;
  /* 1: leave this launch's partial tiles in the scratch (each deferring launch takes the next free region of it) and fold them together with the next ones -- up to 32 per fold
   * launch; the library folds by itself when its table or the scratch is full or when a non-deferring launch needs the scratch, and when the caller says
   * mvlt_tn_fold_flush(): REQUIRED before anything reads a gradient a deferring launch produced.  0: fold right behind the GEMM (ABI 5). */
  int defer_fold;
  /* 1: C = A^T B instead of C += (round 6, ABI 6): the caller knows C holds zeros -- the vocabulary decoder's weight gradient, the first writer of its 94 MB slice of a gradient
   * buffer zeroed at the start of the backward pass -- so the ONE m-split the launch is then made of stores its tiles plainly (16-byte row pieces through LDS) instead of 23 M
   * fire-and-forget fp32 atomics (141 -> ~80 us).  bf16 operands, plain or mapped rows, trans_c == 0, c_taps <= 1, N2 % 4 == 0, ldc % 4 == 0, C 16-byte aligned; forces splits = 1,
   * ignores partials. */
  int c_overwrite;
} mvlt_gemm_tn_args;
int mvlt_gemm_tn(const mvlt_gemm_tn_args* args, void* stream);
/* fold the deferred partial-tile reductions of the scratch `partials` (NULL: of every scratch) now: one launch per scratch on the stream their producers ran on; when `stream`
 * -- the stream of whoever reads the gradients next -- is another one, it is made to wait for that launch (event).  Nothing pending: no launch.  The table of pending folds is
 * kept PER SCRATCH under a mutex: two parameter stores (two models in one backward pass, two host threads) never see each other's entries (ABI 6). */
int mvlt_tn_fold_flush(const void* partials, void* stream);
/* forget the pending folds of `partials` (NULL: all) without launching them: a backward pass that raised leaves descriptors of gradients nobody will use (and whose buffers
 * may be gone by the time the next pass starts) -- the start of a pass discards ITS OWN store's entries, it never folds */
int mvlt_tn_fold_discard(const void* partials);

/* y = LayerNorm(x) * gamma + beta (+ add[(row % add_rows)] ) over the last dim C; rows addressed through maps.
 * Replaces nn.LayerNorm at reference libs/pvlt.py:105,141,142,169,208 and libs/vl_heads.py:33 (eps differs per
 * site: 1e-6 block norms, 1e-5 others, 1e-12 BERT) plus the "+ pos_embed" and torch.cat of libs/pvlt.py:346
 * (y_map writes straight into the concatenated token buffer).  mean/rstd (fp32, [rows]) are saved for backward. */
typedef struct mvlt_layernorm_args {
  const void* x; void* y;
  const float* gamma; const float* beta;
  float* mean; float* rstd;           /* optional outputs [rows] */
  const float* add;                   /* optional fp32 [add_rows, C] added after the affine (pos-embed) */
  int add_rows;
  int rows, C, ldx, ldy;
  mvlt_rowmap x_map, y_map;           /* mode 0 only */
  float eps;
  int dtype;                          /* of x */
  int y_dtype;                        /* of y (bf16 GEMM operand out of an fp32 residual stream, or the reverse) */
  /* optional chained second LayerNorm of the row just produced (the first block's norm1 behind the patch / text embedding's
   * LayerNorm + pos-embed, reference libs/pvlt.py:141,169,208): y2 (bf16, rows laid out like y: same y_map / ldy) =
   * LN(y; gamma2, beta2, eps2); mean2 / rstd2 are indexed by the PHYSICAL row of y.  C in {64, 128, 320, 512, 768} only. */
  void* y2; const float* gamma2; const float* beta2; float eps2; float* mean2; float* rstd2;
} mvlt_layernorm_args;
int mvlt_layernorm_fwd(const mvlt_layernorm_args* args, void* stream);

/* dx = LN backward; dgamma/dbeta (fp32 [C]) accumulate atomically into caller-zeroed buffers (the "+ add" term's
 * gradient is mvlt_batch_sum of dy).  If dx_accumulate != 0, dx += result (several consumers of one tensor). */
typedef struct mvlt_layernorm_bwd_args {
  const void* dy; const void* x; void* dx;
  const float* gamma; const float* mean; const float* rstd;
  float* dgamma; float* dbeta;
  int rows, C, lddy, ldx, lddx;
  mvlt_rowmap dy_map, x_map, dx_map;
  int dx_accumulate;
  int dtype;                          /* of dy */
  int x_dtype, dx_dtype;
  /* optional second output: dx2[row,:] = (final dx)[row,:] * dx2_scale[row / dx2_rows_per_scale], dtype of dy, plain rows with
   * stride lddx2.  It is the DropPath-scaled gradient the next branch's GEMMs read (timm drop_path backward), written while
   * the row is still in registers instead of by a separate pass over dx. */
  void* dx2; const float* dx2_scale; int dx2_rows_per_scale; int lddx2;
  /* dg_copies > 1: dgamma / dbeta are the first of dg_copies interleaved accumulators dg_copy_stride floats apart and workgroup
   * b adds into copy b % dg_copies (mvlt_fold_copies sums them): every workgroup of a launch adds 2*C floats to the same few
   * cache lines, and those requests serialise at the memory side at ~100 ns each (16-27 us per call with one copy).  With dg_copies >= 64
   * the launch uses at most dg_copies workgroups, one copy each, and adds without atomics. */
  int dg_copies; long dg_copy_stride;
} mvlt_layernorm_bwd_args;
int mvlt_layernorm_bwd(const mvlt_layernorm_bwd_args* args, void* stream);
/* dst[dst_index[j]] += sum_k arena[k * stride + j] for j in [j0, j1), and those arena elements are zeroed again: folds the
 * interleaved LayerNorm parameter-gradient accumulators into the flat gradient buffer (one launch per backward stage). */
int mvlt_fold_copies(float* arena, int copies, long stride, const int* dst_index, int j0, int j1, float* dst, void* stream);

/* out[r,c] (fp32) = sum_b in[(b*batch_stride_rows + r) * ld + c]: gradient of the broadcast "+ pos_embed /
 * text_pos_embed" of reference libs/pvlt.py:346 (reduction over the batch). */
int mvlt_batch_sum(const void* in, float* out, int B, int R, int C, long batch_stride_rows, int ld, int dtype,
                   float* acc2 /* nullable: rows r >= split are ADDED to acc2[(r - split), :] instead of stored to out (text_pos_embed's gradient) */, int split, void* stream);

/* Spatial-reduction attention core: O = softmax(Q K^T * scale) V per (batch, head), head_dim = 64,
 * M <= 320 keys (whole K/V of a head stays in LDS; single-pass softmax).  No mask (reference
 * libs/pvlt.py:113-117 applies none).  Q: (B,N,ldq) with head h at columns [64h,64h+64); K,V: (B,M,ldkv)
 * rows, head h at columns k_off+64h / v_off+64h of the kv buffer; O like Q.  lse[B,H,N] fp32 saved for bwd.
 * lse = ref * scale + log(sum of exp((s - ref) * scale)) where ref is a row maximum of the scores (bf16, M <= 192: the maximum over the
 * first half of the keys unless the second half tops it by 2^24 -- the value of lse does not depend on which). */
typedef struct mvlt_attn_args {
  const void* Q; const void* KV; void* O; float* lse;
  int B, H, N, M;
  int ldq, ldkv, ldo;        /* row strides in elements */
  int k_off, v_off;          /* column offsets of K and V inside a kv row */
  float scale;
  int dtype;
} mvlt_attn_args;
int mvlt_sr_attention_fwd(const mvlt_attn_args* args, void* stream);

/* Backward of the above: dQ (like Q) and dKV (B,M,ldkv fp32).  With B*H < 512 the queries of a (batch, head) are split over
 * workgroups that accumulate dKV atomically (caller-zeroed buffer); from B*H >= 512 on one workgroup sees all queries and
 * stores every dKV element exactly once (no zero fill needed). */
typedef struct mvlt_attn_bwd_args {
  const void* Q; const void* KV; const void* O; const void* dO; const float* lse;
  void* dQ; void* dKV;
  int B, H, N, M;
  int ldq, ldkv, ldo, lddkv;
  int k_off, v_off;
  float scale;
  int dtype;
  int dkv_dtype;   /* 1 (default use): dKV is fp32.  0: dKV is bf16 [B, M, lddkv] and takes the plain stores of the one-chunk case
                      directly (dtype == 0 only; FORCES one query chunk per (batch, head): no atomics, so no fp32 staging buffer and no cast --
                      worth it exactly when mvlt_sr_attention_bwd_chunks() says 1 anyway) */
} mvlt_attn_bwd_args;
int mvlt_sr_attention_bwd(const mvlt_attn_bwd_args* args, void* stream);
/* number of query chunks per (batch, head) the backward would split an fp32-dKV launch of this shape into (dtype: MVLT_DT_*): 1 = every dK / dV element is
 * stored exactly once -- the caller may then hand a bf16 dKV (dkv_dtype 0) and skip the zero fill and the cast; > 1 = the chunks meet in fp32 atomics on a
 * caller-zeroed buffer.  Cost model in csrc/attention.hip (whole rounds of the chip x query tiles + atomics), fitted in round 6 (ABI 6). */
int mvlt_sr_attention_bwd_chunks(int B, int H, int N, int M, int dtype);

/* ---- HBM-bound helpers (mvlt_amd/csrc/elementwise.hip) ------------------------------------------------------- */

/* y[row,:] = dropout(LayerNorm(word[ids[row]] + type0 + pos[row % T])); hidden must be 768.  keep: optional
 * uint8 [rows,768] keep-mask (train mode), drop_p its probability.  Replaces transformers BertEmbeddings.forward
 * (call site reference libs/pvlt.py:326).  mean/rstd saved for backward. */
int mvlt_bert_embed_fwd(const long* ids, const float* word, const float* pos, const float* type0, const float* gamma,
                        const float* beta, const uint8_t* keep, float drop_p, void* y, float* mean, float* rstd,
                        int rows, int T, int hidden, float eps, int dtype, void* stream);
/* Backward of the above into caller-zeroed fp32 grads (atomics).  word row 0 (padding_idx) gets no gradient. */
int mvlt_bert_embed_bwd(const void* dy, const long* ids, const float* word, const float* pos, const float* type0,
                        const float* gamma, const uint8_t* keep, float drop_p, const float* mean, const float* rstd,
                        float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                        int rows, int T, int hidden, int dtype, void* stream);

/* out[(b,oi,oj), (c,di,dj)] = img[b,c,oi*k+di,oj*k+dj]: the stage-1 PatchEmbed conv (reference libs/pvlt.py:162,168,
 * kernel=stride=4 on the NCHW fp32 image) becomes a K = 3*4*4 = 48 GEMM over this matrix. */
int mvlt_patchify(const float* img, void* out, int B, int Cin, int H, int W, int k, int dtype, void* stream);

/* Masked-index selection (bit-exact): idx[0..*count) = ascending p with labels[p] != ignore_index.  This is the row
 * set CrossEntropyLoss(ignore_index=-1) averages over (reference engine_grid_masking.py:84). */
int mvlt_masked_select(const long* labels, int n, long ignore_index, int* idx, int* count, void* stream);

/* Bilinear resize (align_corners=False, PyTorch index rule) of a token-major fp32 map in[Hin*Win, C] -> out[Hout*Wout, C]:
 * F.interpolate on the learned position embeddings, reference libs/pvlt.py:291-297 (`_get_pos_embed`).  adjoint != 0: `in` is
 * the gradient w.r.t. the resized [Hout*Wout, C] map and is ACCUMULATED (atomics) into out[Hin*Win, C], the gradient of the
 * source map. */
int mvlt_resize_bilinear_tokens(const float* in, int ld_in, float* out, int ld_out, int Hin, int Win, int Hout, int Wout, int C, int adjoint,
                                void* stream);
/* up to four of them in one launch (host arrays of `count` entries each): the position embeddings of the four stages, and their adjoints */
int mvlt_resize_bilinear_tokens_multi(const float* const* in, const int* ld_in, float* const* out, const int* ld_out, const int* Hin, const int* Win,
                                      const int* Hout, const int* Wout, const int* C, int count, int adjoint, void* stream);

/* out = dy * gelu'(h), exact-erf GELU, elementwise over n values (autograd of reference libs/vl_heads.py:13-14,31-32) */
int mvlt_gelu_bwd(const void* dy, const void* h, void* out, long n, int dtype, void* stream);
/* The engine's loss composition (reference engine_grid_masking.py:81-102: total = mlm + itm + sup_cls + sub_cls + 10 * t2i) in one
 * launch: losses = HOST array of 5 device pointers to fp32 scalars (NULL = head off), weights = HOST array of 5 floats;
 * out (device, 6 floats) = [total, w0 * l0, ..., w4 * l4]; total (device, 1 float) = the total once more, as the tensor the
 * backward pass starts from. */
int mvlt_loss_compose(const float* const* losses, const float* weights, float* out, float* total, void* stream);

/* ---- device-side batch preparation (what the reference's dataset does per sample on the host, mcloader/fashion_gen.py) ----
 * All three draw from Philox4x32-10 with key = seed and counter = (element, sample id, stream): sample b of a call has id
 * sample0 + b, so results do not depend on batch composition; oracle/batchprep_oracle.py restates them bit for bit.
 *
 * mvlt_grid_mask_flags: flags[b, gh*gw] (1 = masked patch).  mode 0: exactly num_mask patches, uniformly.  mode 1: the
 *   reference generator `generate_grid_mask` (fashion_gen.py:225-254): one shuffle of [0]*(P-num_mask)+[1]*num_mask, then patch
 *   row i re-shuffles the window shuffled[i : i+gw].  At most 4096 patches per sample.
 * mvlt_grid_mask_apply: masked[b,c,y,x] = flags[b, y/patch, x/patch] ? fill : image[b,c,y,x] on NCHW fp32 -- the
 *   `image.clone().masked_fill_(mask, 1e-6)` of fashion_gen.py:176 (fill = 1e-6, patch = 16).
 * mvlt_token_mask: `random_masking_features` (fashion_gen.py:383-409) on token ids: positions t >= 1 whose id is not
 *   PAD(0)/CLS(101)/SEP(102) are selected with probability 0.15; selected -> [MASK]=103 (80 %), uniform id in [0, vocab) (10 %),
 *   unchanged (10 %); labels = original id where selected, -1 elsewhere.  ori_ids / input_ids / labels: int64 [B, T]. */
int mvlt_grid_mask_flags(uint8_t* flags, int B, int gh, int gw, int num_mask, int mode, uint64_t seed, uint64_t sample0, void* stream);
int mvlt_grid_mask_apply(const float* image, const uint8_t* flags, float* masked, int B, int C, int H, int W, int patch, float fill, void* stream);
int mvlt_token_mask(const long* ori_ids, long* input_ids, long* labels, int B, int T, uint64_t seed, uint64_t sample0, int vocab, void* stream);
/* Train-mode masks of the step, from the same generator (key = seed, counter = (element, call, stream 2 / 3)), `call` = the
 * caller's running count of draws:
 * mvlt_keep_mask: keep[i] = 1 with probability 1 - drop_p (16-bit draws) -- nn.Dropout(0.1) inside BertEmbeddings (reference
 *   libs/pvlt.py:232-233,326 via transformers); the forward / backward kernels take it as their `keep` operand.
 * mvlt_droppath_scales: out[r][j] = keep / (1 - rates[r]), keep ~ Bernoulli(1 - rates[r]) per (block r, sample j) on 24-bit
 *   draws -- timm DropPath (libs/pvlt.py:135,141-142): x / keep_prob * mask, one mask value per sample. */
int mvlt_keep_mask(uint8_t* keep, long n, float drop_p, uint64_t seed, uint64_t call, void* stream);
int mvlt_droppath_scales(float* out, const float* rates, int nrate, int per, uint64_t seed, uint64_t call, void* stream);

/* dst[r,:] = src[map(idx[r]),:]  and  dst[map(idx[r]),:] (+)= src[r,:]   (idx rows unique; map = mode-0 rowmap or NULL) */
int mvlt_gather_rows(const void* src, const int* idx, void* dst, int rows, int C, int ld_src, const mvlt_rowmap* src_map, int dtype, void* stream);
int mvlt_scatter_rows(const void* src, const int* idx, void* dst, int rows, int C, int ld_dst, const mvlt_rowmap* dst_map, int accumulate, int dtype, void* stream);

/* Row-wise cross entropy.  fwd: lse[r]; loss_sum += lse - logit[label], count += 1 over rows with label != ignore.
 * bwd: dlogits[r,c] = (softmax - onehot) * gscale[0] / max(count[0],1) (0 for ignored rows; columns [V,ldd) zeroed).
 * Replaces torch CrossEntropyLoss at reference engine_grid_masking.py:84,90,94,95. */
int mvlt_cross_entropy_fwd(const void* logits, const long* labels, long ignore_index, float* lse, float* loss_sum, float* count,
                           int rows, int V, int ld, int dtype, void* stream);
int mvlt_cross_entropy_bwd(const void* logits, const long* labels, long ignore_index, const float* lse, const float* gscale,
                           const float* count, void* dlogits, int rows, int V, int ld, int ldd, int dtype, int out_dtype, void* stream);

/* SmoothL1 (beta = 1, mean reduction) between fp32 tensors of n elements: the T2I loss of reference
 * engine_grid_masking.py:99 (F.smooth_l1_loss(t2i_logits, images)).  fwd: *loss_sum += sum of the element losses (caller
 * zeroes it and divides by n); bwd: grad = clamp(pred - target, -1, 1) * gscale[0] / n. */
int mvlt_smooth_l1_fwd(const float* pred, const float* target, long n, float* loss_sum, void* stream);
int mvlt_smooth_l1_bwd(const float* pred, const float* target, long n, const float* gscale /* device scalar */, float* grad, void* stream);

/* torch.optim.AdamW step over a flat fp32 buffer (+ optional bf16 re-cast of the updated parameters).
 * hp (device, fp32[8]) = {lr, beta1, beta2, eps, weight_decay, 1-beta1^t, 1-beta2^t, grad_scale}.
 * Replaces timm create_optimizer('adamw') stepping (reference main_vl.py:308, engine_grid_masking.py:126). */
int mvlt_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, const float* hp,
                    const uint8_t* decay_mask /* [n] 1 = apply weight decay (NULL = all), timm's no-decay split */, void* stream);
int mvlt_cast_bf16(const float* src, void* dst, long n, void* stream);
/* out[row,:] = x[row,:] * scale[row / rows_per_scale] over contiguous [M, C]: the per-sample DropPath factor applied to a
 * block's incoming gradient before its branch GEMMs (timm drop_path backward, reference libs/pvlt.py:133-134). */
int mvlt_row_scale(const void* x, const float* scale, int rows_per_scale, long M, int C, void* out, int dtype, void* stream);
/* backward of the small classification heads (reference libs/vl_heads.py:73-104): dl[B, n_pad] (operand dtype, columns >= n zero) = dlogits[B, n] (fp32),
 * db1[n] += column sums of dlogits, db2[n] += the same (nullable: the heads have two bias parameters, `linear.bias` and `linear_bias`); n_pad <= 256 */
int mvlt_head_grad_prep(const float* dlogits, int B, int n, int n_pad, void* dl, float* db1, float* db2, int dtype, void* stream);
/* out[c*ld_out + r] = in[r*C + c] (fp32 master weight -> transposed compute-dtype operand for the dgrad GEMMs) */
int mvlt_transpose_cast(const float* in, void* out, int R, int C, int ld_out, int dtype, void* stream);

/* All per-step derived weight copies in ONE launch (they used to be ~150 small transposes / permutes / casts per step):
 * a device table of descriptors, each either a 2-D transpose-cast (fp32 [R][C] -> compute dtype [C][ld_out], the W^T dgrad
 * operands) or a 3-D strided gather-cast (dst[i0*ds0 + i1*ds1 + i2*ds2] = src[src_off + i0*ss0 + i1*ss1 + i2*ss2]: the
 * [out][kh][kw][cin] / flipped-tap re-orderings of the conv weights of reference libs/pvlt.py:104,168 and
 * libs/vl_heads.py:107-165).  blk_start[ndesc + 1] = prefix sums of the workgroups each descriptor needs
 * (kind 0 / 2: ceil(R/64)*ceil(C/64), kind 1: ceil(d0*d1*d2 / 256)); blk_desc[total_blocks] (nullable, ABI 4) = the descriptor
 * index of every workgroup, i.e. the inverse of blk_start: with it a workgroup finds its descriptor by two scalar loads instead of a search. */
typedef struct mvlt_prep_desc {
  const float* src; void* dst;
  int kind;                 /* 0 = transpose, 1 = gather, 2 = transpose of a bf16 source (src points at bf16; C % 8 == 0, ld_out % 8 == 0, bf16 dst) */
  int R, C, ld_out;         /* kind 0 */
  int d0, d1, d2;           /* kind 1: loop extents (i2 fastest) */
  int src_off, ss0, ss1, ss2;
  int ds0, ds1, ds2;
} mvlt_prep_desc;
int mvlt_weight_prep(const mvlt_prep_desc* descs /* device */, const int* blk_start /* device, [ndesc + 1] */, int ndesc,
                     int total_blocks, const int* blk_desc /* device, [total_blocks] or NULL */, int dtype /* of every dst */, void* stream);

/* ---- fused MLP for the narrow stages (mvlt_amd/csrc/mlp.hip), bf16 operands, C = 64 or 128 ------------------------
 * Replaces fc1 -> nn.GELU -> fc2 (+ DropPath + residual) of reference libs/pvlt.py:65-71,142 and their autograd: the
 * (tokens x hidden) activation stays in LDS / registers.
 *   mvlt_mlp_fwd    : out[M,C] (fp32) = (gelu(x W1^T + b1) W2^T + b2) * row_scale + residual ; optional h_out = x W1^T + b1
 *                     x [M,C] bf16, w1 = W1 [hid,C], wb = W2 [C,hid]
 *   mvlt_mlp_bwd_dx : out[M,C] (bf16) = ((dy W2) * gelu'(x W1^T + b1)) W1 * row_scale
 *                     w1 = W1 [hid,C], wb = W1^T [C,hid], wc = W2^T [hid,C]
 *   mvlt_mlp_bwd_dw : dW1[hid,C] += dh^T x, db1 += colsum dh, dW2[C,hid] += (dy*row_scale)^T gelu(h), db2 += colsum dy*row_scale
 *                     (fp32 atomics into caller-zeroed buffers; dh, h recomputed on chip)  w1 = W1, wc = W2^T
 * Kernel selection (same results either way, tests/test_kernels_gpu.py::test_fused_mlp): without h_out the forward and the input gradient run
 * the software-pipelined kernel (hid >= 128); rows_per_scale % 64 == 0 (or no row_scale) lets the weight gradients take the per-tile
 * DropPath factor path -- a 64-token tile then never straddles two samples, tiles of samples whose factor is 0 are skipped. */
typedef struct mvlt_mlp_args {
  const void* x; const void* dy;
  const void* w1; const void* wb; const void* wc;
  const float* b1; const float* b2;
  const void* residual;               /* fwd: fp32 [M,C] */
  const float* row_scale; int rows_per_scale;
  void* out;
  void* h_out;                        /* fwd: optional bf16 [M,hid] pre-activation */
  float* dw1; float* db1; float* dw2; float* db2;      /* bwd_dw */
  int M, C, hid;
  /* fwd only, optional: LayerNorm folded into the kernel's operand load (Block.norm2 + Mlp, reference libs/pvlt.py:142).  When
   * ln_x != NULL the kernel reads the fp32 rows ln_x[M,C] (= `residual`, the block's mid stream) instead of `x`, normalises them
   * (two-pass statistics per row, eps = ln_eps, affine ln_gamma / ln_beta), uses the result as the fc1 operand and stores it to
   * ln_y[M,C] (bf16, what `x` would have held: the backward passes read it) together with the row statistics ln_mean / ln_rstd. */
  const float* ln_x; const float* ln_gamma; const float* ln_beta; float ln_eps;
  void* ln_y; float* ln_mean; float* ln_rstd;
  /* fwd only, optional: bf16 [M,C] copy of the output (the MFMA-operand form the next stage's convolutions and the heads read).  `out`
   * may then be NULL: the last block of a stage has no fp32 consumer (reference libs/pvlt.py:331-345: the stage output goes to
   * the next patch embedding / the heads only). */
  void* out_op;
  /* fwd only, optional: LayerNorm of the OUTPUT rows (the NEXT block's norm1, reference libs/pvlt.py:141) computed in the epilogue
   * while a row is in registers: post_y[M,C] (bf16) = LN(out; post_gamma, post_beta, post_eps), row statistics to post_mean /
   * post_rstd -- that block then needs no LayerNorm launch of its own. */
  const float* post_gamma; const float* post_beta; float post_eps;
  void* post_y; float* post_mean; float* post_rstd;
  /* bwd_dx only, optional: the backward of the LayerNorm in FRONT of the MLP (Block.norm2) from the epilogue, where the row of
   * d(LN output) is in registers.  lnb_x [M,C] fp32 = the LayerNorm's input (the block's mid stream), lnb_mean / lnb_rstd [M] its row
   * statistics, lnb_gamma [C]; lnb_dx [M,C] bf16 is the gradient stream, updated in place (dx += LN backward; it may be the tensor
   * passed as `dy`: a workgroup reads its rows of dy before it writes them); lnb_dx2 (optional, bf16 [M,C]) = the updated rows times
   * lnb_dx2_scale[row / lnb_dx2_rows_per_scale] (the attention branch's DropPath factor); lnb_partials [ceil(M / 128)][2 C] fp32
   * receives every workgroup's column sums (d gamma | d beta) for mvlt_add_column_sums.  `out` is not written then. */
  const float* lnb_x; const float* lnb_mean; const float* lnb_rstd; const float* lnb_gamma;
  void* lnb_dx; void* lnb_dx2; const float* lnb_dx2_scale; int lnb_dx2_rows_per_scale;
  float* lnb_partials;
  /* bwd_dw only, optional (round 6, ABI 6): scratch for the token-split reduction WITHOUT atomics, shared with mvlt_gemm_tn (same pointer = same pending-fold table).  Every
   * token split stores its dW1 [hid][C] and dW2 [C][hid] partial sums in bf16 ([splits][hid][C] + [splits][C][hid], 2 x splits x hid x C x 2 bytes) and the ordered fold adds
   * them to dw1 / dw2 -- deterministic, and no 8.4 M fp32 atomics per launch; db1 / db2 (hid + C floats per split) keep their atomics.  defer_fold as in mvlt_gemm_tn_args.
   * Taken by the kernel with a DropPath factor per 64-token tile (rows_per_scale % 64 == 0 or no row_scale); NULL or too small: atomics. */
  void* partials; long partials_bytes; int defer_fold;
} mvlt_mlp_args;
int mvlt_mlp_fwd(const mvlt_mlp_args* args, void* stream);
int mvlt_mlp_bwd_dx(const mvlt_mlp_args* args, void* stream);
int mvlt_mlp_bwd_dw(const mvlt_mlp_args* args, void* stream);
/* dst0[c] += sum over rows of in[r][c] (c < n0), dst1[c - n0] += the same for c >= n0; in: fp32 [rows][cols], row stride ld.  Folds the
 * per-workgroup LayerNorm parameter-gradient partials of mvlt_mlp_bwd_dx (lnb_partials) into the two gradient slices. */
int mvlt_add_column_sums(const float* in, long rows, int cols, int ld, float* dst0, int n0, float* dst1, void* stream);

/* ---- MIM decoder helpers (mvlt_amd/csrc/mim.hip): train-mode BatchNorm over pixel-major [M, C] fp32 matrices, the
 * align_corners=True bilinear resizes and the feature products of reference libs/vl_heads.py:136-165.  The conv3x3
 * themselves are mvlt_gemm_nt / mvlt_gemm_tn with the mode-2 row map. --------------------------------------------- */
int mvlt_col_stats(const float* z, int ldz, long M, int C, float* sum, float* sumsq, void* stream);         /* += */
int mvlt_bn_finalize(const float* sum, const float* sumsq, int copies /* accumulators [copies][C], summed here */, long M, int C, float eps, float momentum, float* mean, float* rstd,
                     float* running_mean, float* running_var /* nullable pair: updated like nn.BatchNorm2d */, void* stream);
/* z, the pre-BatchNorm conv output, is fp32 (z_dtype 1) or fp16 (z_dtype 2: what mvlt_gemm_nt writes with out_dtype 2 -- no MFMA reads z, and it is
 * written once and read three times per step, so the bf16 path keeps it at half the bytes in the type the reference's autocast gives it) */
int mvlt_bn_norm(const void* z, int ldz, int z_dtype, const float* mean, const float* rstd, const float* gamma, const float* beta, long M, int C,
                 void* y32, int ld32, int y32_dtype /* 1 fp32; 2 fp16 (with an fp16 z: the factors of the decoder's three-way feature product, which only
                 elementwise kernels read) */, void* y_op, int ld_op, int op_dtype /* dtype of y_op: the MFMA-operand copy */, void* stream);
/* mvlt_bn_finalize + mvlt_bn_norm in one launch for an fp16 z (the bf16 training path): every workgroup derives the statistics from the conv epilogue's sums,
 * workgroup 0 stores mean / rstd for the backward pass and updates the running statistics (bit-identical to the two separate calls) */
int mvlt_bn_finalize_norm(const void* z /* fp16 */, int ldz, const float* sum, const float* sumsq, int copies, float eps, float momentum, float* mean, float* rstd,
                          float* running_mean, float* running_var /* nullable pair */, const float* gamma, const float* beta, long M, int C,
                          void* y32, int ld32, int y32_dtype, void* y_op /* bf16 */, int ld_op, void* stream);
/* dy (the gradient w.r.t. the BatchNorm output) is fp32 (dy_dtype 1) or bf16 (dy_dtype 0: what the decoder's first backward stages hand
 * over -- a gradient tensor is read twice here and written once by its producer) */
int mvlt_bn_bwd_reduce(const void* dy, int lddy, const void* z, int ldz, int z_dtype, const float* mean, const float* rstd, long M, int C,
                       float* s1 /* += sum dy = dbeta */, float* s2 /* += sum dy*xhat = dgamma */, int dy_dtype, void* stream);
int mvlt_bn_bwd_apply(const void* dy, int lddy, const void* z, int ldz, int z_dtype, const float* mean, const float* rstd, const float* gamma,
                      const float* s1, const float* s2, long M, int C, void* dz_op, int lddz,
                      float* g_beta, float* g_gamma /* nullable pair: += s1, += s2 (the BatchNorm parameter gradients) */,
                      int op_dtype, int dy_dtype, void* stream);
/* out (+)= a*b(*c) elementwise over [M, C] fp32 with row strides; optional operand-dtype copy of the result */
int mvlt_ew_mul(float* out, int ldo, const void* a, int lda, const void* b, int ldb, const void* c, int ldc, int in_dtype /* of a, b, c: 1 fp32, 2 fp16 */,
                long M, int C, int accumulate, void* out_op, int ld_op, int op_dtype, void* stream);
/* gradients of y = a*b*c (all [M, C] fp32, row stride ld; dy row stride lddy): da = dy*b*c, db = dy*a*c, dc = dy*a*b
 * (the three-way feature product of reference libs/vl_heads.py:152) */
int mvlt_ew_mul3_bwd(const void* dy /* fp32 or bf16 (dy_dtype) */, int lddy, const void* a, const void* b, const void* c, int ld, int in_dtype /* 1 fp32, 2 fp16 */,
                     void* da, void* db, void* dc /* dy's dtype */, long M, int C, int dy_dtype, void* stream);
/* bilinear resize by an integer factor, align_corners=True.  x fp32 [B,H,W,C] (row stride ldx) -> [B,sH,sW,C] (bf16/fp32,
 * row stride ldo) or NCHW fp32 [B,C,sH,sW]; bwd is the exact adjoint in gather form (no atomics); its dx is fp32 (dx_dtype 1) or,
 * behind the NCHW upsample only, bf16 (dx_dtype 0: the [pixels][8]-padded operand of the score conv's gradient GEMMs). */
int mvlt_upsample_fwd(const float* x, int ldx, int B, int H, int W, int C, int scale, void* out, int ldo, int out_dtype, int nchw, void* stream);
int mvlt_upsample_bwd(const void* dy /* fp32; bf16 (dy_dtype 0) for the pixel-major resizes */, int lddy, int nchw, int B, int H, int W, int C, int scale, void* dx,
                      int lddx, int accumulate, int dx_dtype, int dy_dtype, void* stream);
/* The MIM loss without the image-sized prediction (training): SmoothL1(beta 1, mean) between the x scale bilinear upsample
 * (align_corners=True) of the score map x[B,H,W,C] (pixel-major fp32, row stride ldx) and the NCHW fp32 target [B,C,H*scale,W*scale]
 * (reference libs/vl_heads.py:163-165 + engine_grid_masking.py:99).  fwd: *loss_sum += the SUM over all elements (the caller divides by
 * their number); bwd: dx[B*H*W, lddx] (bf16 or fp32, columns 0..C-1) = d(mean loss)/d(x) * gscale[0], recomputing the prediction.
 * Geometry of the final x8 upsample only: W <= 64, W*scale a multiple of 4 and <= 256. */
int mvlt_upsample_l1_fwd(const float* x, int ldx, int B, int H, int W, int C, int scale, const float* target, float* loss_sum, void* stream);
int mvlt_upsample_l1_bwd(const float* x, int ldx, int B, int H, int W, int C, int scale, const float* target, const float* gscale, void* dx, int lddx,
                         int dx_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif
