"""GPU: device-side batch preparation (csrc/batchprep.hip through mvlt_amd.batchprep) -- bit-exact against the numpy oracle
and the reference-made fixture, and the side-stream prefetcher feeding the engine loop."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import batchprep_oracle as BP
from oracle import filler
from oracle import pvlt_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("S,ratio,mode", [(256, 0.5, "exact"), (256, 0.5, "reference"), (384, 0.5, "reference"), (384, 0.75, "exact"),
                                          (96, 0.5, "reference"), (1024, 0.5, "exact")])
def test_batch_prep_is_bit_exact_against_the_oracle(S, ratio, mode):
    from mvlt_amd.batchprep import DeviceBatchPrep
    B, T, seed, sample0 = (5, 128, 77, 1000) if S < 1024 else (2, 16, 3, (1 << 33) + 5)
    nb = filler.make_batch(9, B, S, T)
    prep = DeviceBatchPrep(seed, mask_ratio=ratio, mode=mode)
    out = prep(torch.from_numpy(nb["image"]).to(DEV), torch.from_numpy(nb["ori_input_ids"]).to(DEV), sample0=sample0)
    torch.cuda.synchronize()
    g = S // 16
    want = BP.prepare_batch(seed, sample0, nb["image"], nb["ori_input_ids"], int(ratio * g * g), 0 if mode == "exact" else 1)
    assert np.array_equal(out["patch_flags"].cpu().numpy(), want["patch_flags"])
    assert np.array_equal(out["masked_images"].cpu().numpy(), want["masked_images"])            # 1e-6 fill, every other pixel untouched
    assert np.array_equal(out["input_ids"].cpu().numpy(), want["input_ids"])
    assert np.array_equal(out["mlm_labels"].cpu().numpy(), want["mlm_labels"])
    n = int(out["mlm_count_dev"].item())
    assert np.array_equal(out["mlm_positions_buf"][:n].cpu().numpy(), want["mlm_positions"])   # masked-index selection
    if mode == "exact":
        assert (out["patch_flags"].reshape(B, -1).sum(1) == int(ratio * g * g)).all()


def test_device_masks_equal_the_reference_made_fixture(golden_dir):
    from mvlt_amd import ops
    g = np.load(os.path.join(golden_dir, "batchprep_ref.npz"))
    seed = int(g["meta"][0])
    for ci, (S, ratio) in enumerate(g["grid/cases"]):
        gs = int(S) // 16
        flags = torch.empty(4, gs * gs, dtype=torch.uint8, device=DEV)
        ops.grid_mask_flags(flags, 4, gs, gs, int(ratio * gs * gs), 1, seed, 0)
        for sample in range(4):
            assert np.array_equal(flags[sample].view(gs, gs).cpu().numpy(), g[f"grid/{ci}/{sample}"]), (S, ratio, sample)
    ori = torch.from_numpy(np.stack([g[f"tok/{s}/ori"] for s in range(6)])).to(DEV)
    ids, lab = torch.empty_like(ori), torch.empty_like(ori)
    ops.token_mask(ori, ids, lab, seed, 0)
    for s in range(6):
        assert np.array_equal(ids[s].cpu().numpy(), g[f"tok/{s}/ids"]) and np.array_equal(lab[s].cpu().numpy(), g[f"tok/{s}/labels"])


@pytest.mark.parametrize("n,p,seed,call", [(8 * 1000, 0.1, 5, 0), (32768 * 768, 0.1, (1 << 40) + 9, (1 << 33) + 2), (1003, 0.5, 1, 7), (8, 0.0, 2, 3)])
def test_step_dropout_mask_is_bit_exact(n, p, seed, call):
    """mvlt_keep_mask (the nn.Dropout of BertEmbeddings, reference libs/pvlt.py:232-233) against the numpy restatement"""
    from mvlt_amd import ops
    keep = torch.full((n + 8,), 7, dtype=torch.uint8, device=DEV)
    ops.keep_mask(keep[:n], p, seed, call)
    want = BP.keep_mask(seed, call, n, p)
    assert np.array_equal(keep[:n].cpu().numpy(), want)
    assert (keep[n:] == 7).all()                                  # nothing written past n
    if n > 100000:
        assert abs(float(want.mean()) - (1.0 - p)) < 1e-3


def test_step_droppath_scales_are_bit_exact():
    """mvlt_droppath_scales (timm DropPath's per-sample keep / keep_prob, reference libs/pvlt.py:135,141-142)"""
    from mvlt_amd import ops
    rates = np.linspace(0.0, 0.3, 8).astype(np.float32)
    out = torch.empty(8, 2, 256, device=DEV)
    ops.droppath_scales(out, torch.from_numpy(rates).to(DEV), 123, 45)
    want = BP.droppath_scales(123, 45, rates, 512).reshape(8, 2, 256)
    assert np.array_equal(out.cpu().numpy(), want)
    assert (want[0] == 1.0).all()                                 # rate 0 keeps everything
    kept = (want[-1] > 0).mean()
    assert abs(kept - 0.7) < 0.08 and np.allclose(want[-1][want[-1] > 0], 1.0 / 0.7, rtol=1e-6)


def _cpu_batches(n, B, S, T):
    return [O.to_torch_batch(filler.make_batch(40 + i, B, S, T)) for i in range(n)]


def test_prefetcher_delivers_batches_and_counts():
    from mvlt_amd.batchprep import DeviceBatchPrep, DevicePrefetcher
    B, S, T = 3, 64, 32
    batches = _cpu_batches(5, B, S, T)
    got = list(DevicePrefetcher(batches, DEV))
    assert len(got) == 5
    for b, g_ in zip(batches, got):
        for k in ("image", "masked_images", "input_ids", "mlm_labels", "itm_labels"):
            assert g_[k].is_cuda and torch.equal(g_[k].cpu(), b[k]), k
        pos = torch.nonzero(b["mlm_labels"].reshape(-1) != -1).flatten()
        assert g_["mlm_count"] == pos.numel() and torch.equal(g_["mlm_positions"].cpu().long(), pos)
    # with on-device preparation: only image + ori_input_ids + labels travel, the rest is made on the GPU
    slim = [{k: v for k, v in b.items() if k not in ("masked_images", "input_ids", "mlm_labels")} for b in batches]
    got = list(DevicePrefetcher(slim, DEV, prep=DeviceBatchPrep(21, 0.5, "exact")))
    for i, g_ in enumerate(got):
        want = BP.prepare_batch(21, i * B, batches[i]["image"].numpy(), batches[i]["ori_input_ids"].numpy(), 8, 0)
        assert np.array_equal(g_["masked_images"].cpu().numpy(), want["masked_images"])
        assert np.array_equal(g_["input_ids"].cpu().numpy(), want["input_ids"])
        assert np.array_equal(g_["mlm_labels"].cpu().numpy(), want["mlm_labels"])
        assert g_["mlm_count"] == len(want["mlm_positions"]) and np.array_equal(g_["mlm_positions"].cpu().numpy(), want["mlm_positions"])


def test_engine_loop_runs_on_the_prefetcher(parity):
    """train_one_epoch_vl over DevicePrefetcher(loader) gives the losses of the plain loader (same batches; fp32 atomics aside)"""
    from mvlt_amd import pvlt
    from mvlt_amd.batchprep import DevicePrefetcher
    from mvlt_amd.engine import BF16Scaler, train_one_epoch_vl
    from mvlt_amd.optim import FusedAdamW
    lt = dict(mlm=1, itm=1, t2i=1, cls=0)
    B, S, T = 2, 64, 32
    batches = _cpu_batches(4, B, S, T)
    cfg = O.Cfg("pvlt_tiny", lt, 224, 768, T, 0.0)
    res = []
    for wrap in (False, True):
        m = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=T, loss_type=lt, pretrained_pth=None, drop_path_rate=0.0,
                           compute_dtype=torch.float32)
        m.load_state_dict(O.filled_state_dict(cfg, 6), strict=True)
        m.cuda()
        m.injected_masks = dict(bert=torch.ones(B, T, 768), droppath=[torch.ones(B)] * 8, droppath2=[torch.ones(B)] * 8)
        opt = FusedAdamW(m, lr=1e-4, weight_decay=0.01)
        loader = DevicePrefetcher(batches, DEV) if wrap else batches
        res.append(train_one_epoch_vl(m, None, loader, opt, torch.device(DEV), 0, BF16Scaler(), None, None, None, True, False,
                                      types.SimpleNamespace(loss_type=lt)))
    for k in res[0]:
        assert parity(f"prefetch/{k}", abs(res[0][k] - res[1][k]) / max(1.0, abs(res[0][k])), 1e-5), (k, res[0][k], res[1][k])
