"""Deterministic, platform-independent tensor filler (TEST INFRASTRUCTURE).

The reference ships no weights and no fixtures, and its
Python cannot travel to the GPU box.  Golden vectors are therefore captured
with *synthetic* weights that both sides can regenerate bit-for-bit from a
name and a seed: an integer hash (splitmix64 finaliser) of
``(seed, crc32(name), flat_index)`` mapped to [0,1) with 24 bits, then shaped
per tensor kind so activations stay O(1) through the 8..28 blocks.

Only numpy integer arithmetic is used: identical on every host.
Nothing in the product package imports this module.
"""
import zlib

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def unit(seed, name, n):
    """n float64 values in [0,1), a pure function of (seed, name, index)."""
    h0 = np.uint64((zlib.crc32(name.encode("utf-8")) ^ ((seed * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        x = (np.arange(1, n + 1, dtype=np.uint64) + h0 * np.uint64(0x100000001)) * _GOLD
        x ^= x >> np.uint64(30)
        x *= _M1
        x ^= x >> np.uint64(27)
        x *= _M2
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def sym(seed, name, n):
    """n float64 values in [-1,1)."""
    return 2.0 * unit(seed, name, n) - 1.0


def fill_param(seed, name, shape):
    """Value for one state_dict entry, chosen by the entry's role (its name)."""
    shape = tuple(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if name.endswith("position_ids"):
        return np.arange(n, dtype=np.int64).reshape(shape)
    if name.endswith("running_var"):
        v = 0.5 + unit(seed, name, n)
    elif name.endswith("running_mean"):
        v = 0.1 * sym(seed, name, n)
    elif "pos_embed" in name:
        v = 0.2 * sym(seed, name, n)
    elif "embeddings.weight" in name:  # word / position / token-type tables
        v = 0.1 * sym(seed, name, n)
    elif len(shape) == 1 and name.endswith(".weight"):  # LayerNorm / BatchNorm gain
        v = 1.0 + 0.2 * sym(seed, name, n)
    elif len(shape) == 1:  # every bias, incl. mlm_head.bias and *.linear_bias
        v = 0.1 * sym(seed, name, n)
    else:  # Linear (out,in) or Conv (out,in,kh,kw): variance-preserving uniform
        fan_in = int(np.prod(shape[1:]))
        v = np.sqrt(3.0 / fan_in) * sym(seed, name, n)
        if name.startswith("t2i_head.score"):   # keep the MIM output O(1): SmoothL1 stays in its quadratic zone
            v = 0.05 * v
    return v.reshape(shape).astype(np.float32)


def fill_state_dict(seed, shapes):
    """shapes: ordered mapping name -> shape.  Tied names must be de-duplicated
    by the caller (the tied decoder takes the word-embedding tensor)."""
    return {k: fill_param(seed, k, s) for k, s in shapes.items()}


def make_batch(seed, batch, img_size=256, num_tokens=128, vocab=30522, n_sup=48, n_sub=122):
    """Synthetic batch with the schema the engine consumes
    (reference engine_grid_masking.py:42-56, mcloader/fashion_gen.py:192-209)."""
    B, S, T = batch, img_size, num_tokens
    image = unit(seed, "image", B * 3 * S * S).reshape(B, 3, S, S).astype(np.float32)
    # grid mask: 16x16-pixel patches, exactly half masked, masked pixels = 1e-6
    g = S // 16
    order = np.argsort(unit(seed, "grid", B * g * g).reshape(B, g * g), axis=1, kind="stable")
    pm = np.zeros((B, g * g), dtype=bool)
    np.put_along_axis(pm, order[:, : (g * g) // 2], True, axis=1)
    pm = np.repeat(np.repeat(pm.reshape(B, g, g), 16, axis=1), 16, axis=2)
    masked = np.where(pm[:, None, :, :], np.float32(1e-6), image).astype(np.float32)

    r_len = unit(seed, "caplen", B)
    r_tok = unit(seed, "tok", B * T).reshape(B, T)
    r_sel = unit(seed, "sel", B * T).reshape(B, T)
    r_how = unit(seed, "how", B * T).reshape(B, T)
    r_rnd = unit(seed, "rnd", B * T).reshape(B, T)
    ori = np.zeros((B, T), dtype=np.int64)
    ids = np.zeros((B, T), dtype=np.int64)
    lab = -np.ones((B, T), dtype=np.int64)
    for b in range(B):
        L = 20 + int(r_len[b] * 41) if T >= 64 else max(1, T - 2)
        L = min(L, T - 2)
        ori[b, 0] = 101
        ori[b, 1 : 1 + L] = 1000 + (r_tok[b, :L] * (vocab - 1000)).astype(np.int64)
        ori[b, 1 + L] = 102
        ids[b] = ori[b]
        sel = np.nonzero(r_sel[b, :L] < 0.15)[0] + 1
        if len(sel) == 0:
            sel = np.array([1])
        for p in sel:
            lab[b, p] = ori[b, p]
            if r_how[b, p] < 0.8:
                ids[b, p] = 103
            elif r_how[b, p] < 0.9:
                ids[b, p] = 1000 + int(r_rnd[b, p] * (vocab - 1000))
    itm = (unit(seed, "itm", B) < 0.5).astype(np.int64).reshape(B, 1)
    sup = (unit(seed, "sup", B) * n_sup).astype(np.int64).reshape(B, 1)
    sub = (unit(seed, "sub", B) * n_sub).astype(np.int64).reshape(B, 1)
    return dict(image=image, masked_images=masked, input_ids=ids, ori_input_ids=ori,
                mlm_labels=lab, i2t_labels=ori.copy(), itm_labels=itm,
                sup_cls_labels=sup, sub_cls_labels=sub)
