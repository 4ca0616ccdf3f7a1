"""CPU: the batch-preparation oracle (oracle/batchprep_oracle.py) -- its generator against the published Philox4x32-10
known-answer vectors, its decision logic against the fixture the REAL reference functions produced
(tests/golden/batchprep_ref.npz: mcloader/fashion_gen.py generate_grid_mask / random_masking_features driven with the same
draws, see make_golden.py:run_batchprep_case), plus the properties the domain offers."""
import os

import numpy as np

from oracle import batchprep_oracle as BP


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds"""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = BP.philox4x32(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(v[0]) for v in got) == want


def test_thresholds_are_the_reference_probabilities():
    for T, p in ((BP.T15, 0.15), (BP.T80, 0.8), (BP.T90, 0.9)):
        assert (T - 1) / 2 ** 24 < p <= T / 2 ** 24            # r < T  <=>  r / 2^24 < p


def test_grid_mask_and_token_logic_equal_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "batchprep_ref.npz"))
    seed = int(g["meta"][0])
    for ci, (S, ratio) in enumerate(g["grid/cases"]):
        gsz = int(S) // 16
        for sample in range(4):
            mine = BP.grid_flags(seed, sample, gsz, gsz, int(ratio * gsz * gsz), 1)
            assert np.array_equal(mine, g[f"grid/{ci}/{sample}"]), (S, ratio, sample)
    for sample in range(6):
        ids, lab = BP.mask_tokens(seed, sample, g[f"tok/{sample}/ori"])
        assert np.array_equal(ids, g[f"tok/{sample}/ids"]) and np.array_equal(lab, g[f"tok/{sample}/labels"])


def test_exact_mode_masks_exactly_half_and_fill_value():
    for sample in range(5):
        f = BP.grid_flags(11, sample, 16, 16, 128, 0)
        assert f.sum() == 128 and f.shape == (16, 16)
    assert not np.array_equal(BP.grid_flags(11, 0, 16, 16, 128, 0), BP.grid_flags(11, 1, 16, 16, 128, 0))
    assert not np.array_equal(BP.grid_flags(11, 0, 16, 16, 128, 0), BP.grid_flags(12, 0, 16, 16, 128, 0))
    img = np.random.default_rng(0).random((3, 32, 32), dtype=np.float32)
    flags = np.array([[1, 0], [0, 1]], dtype=np.uint8)
    m = BP.apply_grid_mask(img, flags)
    assert (m[:, :16, :16] == np.float32(1e-6)).all() and (m[:, 16:, 16:] == np.float32(1e-6)).all()
    assert np.array_equal(m[:, :16, 16:], img[:, :16, 16:]) and np.array_equal(m[:, 16:, :16], img[:, 16:, :16])


def test_token_masking_conventions_and_rates():
    T, n = 128, 400
    ori = np.zeros((n, T), dtype=np.int64)
    ori[:, 0] = BP.TOK_CLS
    ori[:, 1:101] = 2000 + np.arange(100)[None]
    ori[:, 101] = BP.TOK_SEP
    sel = msk = rnd = keep = 0
    for b in range(n):
        ids, lab = BP.mask_tokens(5, b, ori[b])
        s = lab != -1
        assert not s[0] and not s[101:].any()                       # [CLS], [SEP] and padding are never selected
        assert np.array_equal(lab[s], ori[b][s])                    # label = original id
        assert np.array_equal(ids[~s], ori[b][~s])                  # unselected positions are untouched
        sel += s.sum()
        msk += (ids[s] == BP.TOK_MASK).sum()
        keep += (ids[s] == ori[b][s]).sum()
        rnd += ((ids[s] != BP.TOK_MASK) & (ids[s] != ori[b][s])).sum()
        assert ids.min() >= 0 and ids.max() < BP.VOCAB
    tot = n * 100
    assert abs(sel / tot - 0.15) < 0.01 and abs(msk / sel - 0.8) < 0.02 and abs(rnd / sel - 0.1) < 0.015 and abs(keep / sel - 0.1) < 0.015


def test_prepare_batch_is_independent_of_batch_composition():
    rng = np.random.default_rng(1)
    img = rng.random((4, 3, 32, 32), dtype=np.float32)
    ori = np.zeros((4, 16), dtype=np.int64)
    ori[:, 0], ori[:, 1:9], ori[:, 9] = 101, rng.integers(1000, 30000, (4, 8)), 102
    whole = BP.prepare_batch(3, 10, img, ori, 2, 0)
    part = BP.prepare_batch(3, 12, img[2:], ori[2:], 2, 0)
    assert np.array_equal(whole["masked_images"][2:], part["masked_images"]) and np.array_equal(whole["mlm_labels"][2:], part["mlm_labels"])
    assert np.array_equal(whole["mlm_positions"], np.nonzero(whole["mlm_labels"].reshape(-1) != -1)[0])
