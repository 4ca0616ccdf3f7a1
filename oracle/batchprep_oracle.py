"""CPU oracle for the device-side batch preparation (TEST INFRASTRUCTURE -- never shipped, never timed as the product).

numpy restatement of what the reference's dataset does per sample on the host before a batch reaches the engine:
  grid mask        mcloader/fashion_gen.py:225-254 (`generate_grid_mask`: patch_size 16, int(mask_ratio * patches) masked
                   patches, one global shuffle, then per patch-row a window of the shuffled list re-shuffled) and :176
                   (`image.clone().masked_fill_(mask, 1e-6)`)
  token masking    mcloader/fashion_gen.py:383-409 (`random_masking_features`: 15 % of the caption tokens; of those 80 % ->
                   [MASK], 10 % -> a uniformly random vocabulary entry, 10 % kept; label = original id, -1 elsewhere) and the
                   label / id assembly at :340-364

The reference draws from numpy's / Python's global Mersenne Twisters inside DataLoader workers.  The device path uses a
counter-based generator instead -- Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11;
Random123 reference constants), keyed by (seed) and counted by (sample id, element, stream) -- so a sample's masks do not
depend on batch composition, rank count or launch geometry, and this file can restate the device result bit for bit.

Parity pin: the *decision logic* is pinned against the reference's own functions run in the build container with their
random draws replaced by the Philox-derived ones (tests/golden/make_golden.py: run_batchprep_case ->
tests/golden/batchprep_ref.npz); the generator is pinned by the published Philox4x32-10 known-answer vectors
(tests/test_batchprep_cpu.py).  The random streams themselves cannot equal the reference's (different generator): that part
is "parity unpinned" by construction and stated so in DESIGN.md.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

STREAM_GRID, STREAM_ROW, STREAM_TOKEN = 0, 1, 2
STREAM_DROPOUT, STREAM_DROPPATH = 4, 5
TOK_CLS, TOK_SEP, TOK_MASK, TOK_PAD, VOCAB = 101, 102, 103, 0, 30522
# integer thresholds on 24-bit draws: r < T  <=>  r / 2^24 < p  for p in {0.15, 0.8, 0.9}
T15, T80, T90 = 2516583, 13421773, 15099495


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """vectorised Philox4x32: counters / keys are uint32-valued arrays (broadcastable); returns 4 uint32 arrays"""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & MASK32 for v in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(rounds):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK32, p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & MASK32, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & MASK32, lo0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def draws(seed, sample, element, stream):
    """the 4 x 32 random bits of (sample id, element index, stream) under `seed`: counter = (element, sample_lo, stream,
    sample_hi), key = (seed_lo, seed_hi)"""
    sample = np.asarray(sample, dtype=np.uint64)
    return philox4x32(element, sample & MASK32, stream, sample >> np.uint64(32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def perm_from_keys(keys):
    """the permutation a key sort defines: position k receives the element with the k-th smallest key (ties by index)"""
    return np.argsort(keys, kind="stable")


def grid_flags(seed, sample, gh, gw, num_mask, mode, perms=None):
    """(gh, gw) uint8 patch flags of one sample (1 = masked).
    mode 0: exactly `num_mask` of the gh*gw patches, uniformly (the variant bench.py / SURVEY.md 8d use).
    mode 1: the reference's generator (fashion_gen.py:236-251): the list [0]*(P-num_mask) + [1]*num_mask is shuffled once;
            patch-row i then takes the WINDOW shuffled[i : i+gw] (not [i*gw : (i+1)*gw] -- the reference's quirk, which makes
            the realised ratio vary) and shuffles it again.
    perms: optional (perm_global, [perm_row_i ...]) to replace the Philox-derived shuffles (how the reference is driven when
    the goldens are made)."""
    P = gh * gw
    if perms is None:
        k1 = draws(seed, sample, np.arange(P), STREAM_GRID)[0]
        pg = perm_from_keys(k1)
        k2 = draws(seed, sample, np.arange(P), STREAM_ROW)[0].reshape(gh, gw)
        prow = [perm_from_keys(k2[i]) for i in range(gh)]
    else:
        pg, prow = perms
    if mode == 0:
        flags = np.zeros(P, dtype=np.uint8)
        flags[pg[:num_mask]] = 1                    # the num_mask patches with the smallest keys
        return flags.reshape(gh, gw)
    orig = np.concatenate([np.zeros(P - num_mask, np.uint8), np.ones(num_mask, np.uint8)])
    shuffled = orig[pg]
    out = np.zeros((gh, gw), dtype=np.uint8)
    for i in range(gh):
        cur = shuffled[i:i + gw]
        out[i] = cur[prow[i]]
    return out


def apply_grid_mask(image, flags, patch=16, fill=1e-6):
    """image (C, H, W) float32, flags (gh, gw) -> masked copy: masked pixels = fill (fashion_gen.py:176)"""
    m = np.repeat(np.repeat(flags.astype(bool), patch, axis=0), patch, axis=1)
    return np.where(m[None], np.float32(fill), image).astype(np.float32)


def mask_tokens(seed, sample, ori_ids, draws3=None):
    """ori_ids (T,) int64 = [CLS] caption [SEP] pad...  ->  (input_ids, mlm_labels).
    Caption positions (t >= 1, id not PAD / CLS / SEP) are selected with probability 0.15; a selected position becomes [MASK]
    (80 %), a uniform vocabulary id (10 %) or stays (10 %); its label is the original id, every other label is -1.
    draws3: optional (r1, r2, r3) arrays replacing the Philox draws (24-bit r1, r2; 32-bit r3)."""
    T = ori_ids.shape[0]
    if draws3 is None:
        x0, x1, x2, _ = draws(seed, sample, np.arange(T), STREAM_TOKEN)
        r1, r2, r3 = x0 >> np.uint32(8), x1 >> np.uint32(8), x2
    else:
        r1, r2, r3 = draws3
    t = np.arange(T)
    cand = (t >= 1) & (ori_ids != TOK_PAD) & (ori_ids != TOK_CLS) & (ori_ids != TOK_SEP)
    sel = cand & (r1 < T15)
    ids = ori_ids.copy()
    rnd = ((r3.astype(np.uint64) * np.uint64(VOCAB)) >> np.uint64(32)).astype(np.int64)      # uniform in [0, VOCAB)
    ids = np.where(sel & (r2 < T80), TOK_MASK, ids)
    ids = np.where(sel & (r2 >= T80) & (r2 < T90), rnd, ids)
    labels = np.where(sel, ori_ids, -1).astype(np.int64)
    return ids.astype(np.int64), labels


def prepare_batch(seed, sample0, images, ori_ids, num_mask, mode, patch=16):
    """whole-batch restatement of mvlt_amd.batchprep.DeviceBatchPrep: sample b has id sample0 + b"""
    B, C, H, W = images.shape
    gh, gw = H // patch, W // patch
    flags = np.stack([grid_flags(seed, sample0 + b, gh, gw, num_mask, mode) for b in range(B)])
    masked = np.stack([apply_grid_mask(images[b], flags[b], patch) for b in range(B)])
    tok = [mask_tokens(seed, sample0 + b, ori_ids[b]) for b in range(B)]
    input_ids = np.stack([t[0] for t in tok])
    labels = np.stack([t[1] for t in tok])
    positions = np.nonzero(labels.reshape(-1) != -1)[0].astype(np.int32)
    return dict(masked_images=masked, patch_flags=flags, input_ids=input_ids, mlm_labels=labels, mlm_positions=positions)


# ---- train-mode masks of the step (mvlt_keep_mask / mvlt_droppath_scales): same generator, streams 4 / 5 (their own: stream 2 is the
# token masking's, and with equal seeds call n would otherwise reuse the draws of sample id n), counted by the caller's
# running draw count `call` in the place of the sample id
def keep_mask(seed, call, n, drop_p):
    """n uint8 keep flags, Bernoulli(1 - drop_p) on 16-bit draws: nn.Dropout(drop_p) of BertEmbeddings (reference libs/pvlt.py:232-233
    via transformers' BertEmbeddings.dropout); element i uses half (i % 2) of word (i % 8) // 2 of Philox call i // 8"""
    groups = (n + 7) // 8
    w = np.stack(draws(seed, call, np.arange(groups), STREAM_DROPOUT), axis=1)                 # (groups, 4) uint32
    lo, hi = w & np.uint32(0xFFFF), w >> np.uint32(16)
    d16 = np.stack([lo, hi], axis=2).reshape(groups, 8)                          # e -> word e >> 1, half e & 1
    thr = np.uint32(int(np.float32(drop_p) * np.float32(65536.0)))
    return (d16 >= thr).astype(np.uint8).reshape(-1)[:n]


def droppath_scales(seed, call, rates, per):
    """(len(rates), per) float32: keep / (1 - rate) with keep ~ Bernoulli(1 - rate) on 24-bit draws -- timm DropPath's per-sample factor
    (reference libs/pvlt.py:135,141-142: x / keep_prob * floor(keep_prob + U))"""
    rates = np.asarray(rates, dtype=np.float32)
    n = rates.shape[0] * per
    d = draws(seed, call, np.arange(n), STREAM_DROPPATH)[0] >> np.uint32(8)
    r = np.repeat(rates, per)
    thr = (r * np.float32(16777216.0)).astype(np.uint32)
    return np.where(d >= thr, np.float32(1.0) / (np.float32(1.0) - r), np.float32(0.0)).astype(np.float32).reshape(rates.shape[0], per)
