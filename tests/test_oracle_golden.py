"""CPU: the oracle (oracle/pvlt_oracle.py) against the committed golden vectors captured from
the real reference by tests/golden/make_golden.py.  This is the oracle's pin:
the reference has no tests or fixtures of its own."""
import os

import numpy as np
import pytest
import torch

from oracle import filler
from oracle.hostinfo import usable_cores
from oracle import pvlt_oracle as O

CASES = {
    "tiny256_pretrain": dict(variant="pvlt_tiny", lt=dict(mlm=1, itm=1, t2i=1, cls=0), train=True),
    "tiny256_ft": dict(variant="pvlt_tiny", lt=dict(mlm=0, itm=0, t2i=0, cls=1), train=True),
    "tiny256_all": dict(variant="pvlt_tiny", lt=dict(mlm=1, itm=1, t2i=1, cls=1), train=False),
    "tiny224_pretrain": dict(variant="pvlt_tiny", lt=dict(mlm=1, itm=1, t2i=1, cls=0), train=False),
    "tiny384_pretrain": dict(variant="pvlt_tiny", lt=dict(mlm=1, itm=1, t2i=1, cls=0), train=False),
    "small96_T20_ragged": dict(variant="pvlt_small", lt=dict(mlm=1, itm=1, t2i=1, cls=1), train=True),
    "large96_T20": dict(variant="pvlt_large", lt=dict(mlm=1, itm=1, t2i=1, cls=1), train=True),        # the fourth factory (round 6); the batch-64 / medium batch-8
}                                                                                                          # fixtures are GPU-suite cases (minutes per pass on CPU)
TOL = 2e-4   # fp32 CPU vs fp32 CPU on another host/thread count: summation-order noise only


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, B, img, T, dp = g["meta"]
    return g, int(seed), int(B), int(img), int(T), float(dp)


def sample(t, n=64):
    f = t.detach().reshape(-1).to(torch.float32)
    stride = max(1, f.numel() // n) | 1
    return f[::stride][:n].numpy()


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(1e-6, np.abs(b).max())
    return np.abs(a - b).max() / scale <= tol


@pytest.mark.parametrize("name", list(CASES))
def test_eval_forward_matches_reference_golden(golden_dir, name):
    torch.set_num_threads(min(8, usable_cores()))
    c = CASES[name]
    g, seed, B, img, T, dp = load(golden_dir, name)
    cfg = O.Cfg(c["variant"], c["lt"], 224, 768, T, dp)
    sd = O.filled_state_dict(cfg, seed)
    batch = O.to_torch_batch(filler.make_batch(seed, B, img, T))
    taps = {}
    with torch.no_grad():
        out = O.forward(sd, cfg, batch["image"], batch["input_ids"], taps=taps)
    for k in g.files:
        if k.startswith("eval/tap/") and k.endswith("/sample"):
            tap = k.split("/")[2]
            assert close(sample(taps[tap], 1024), g[k]), k
        if k.startswith("eval/out/") and k.endswith("/sample"):
            key = k.split("/")[2]
            assert close(sample(out[key], 4096), g[k]), k
        if k.startswith("eval/full/"):
            assert close(out[k.split("/")[2]].numpy(), g[k]), k
    pos = O.masked_positions(batch["mlm_labels"])
    assert np.array_equal(pos.numpy(), g["masked_positions"])          # bit-exact index selection
    if out["mlm_logits"] is not None:
        rows = out["mlm_logits"].reshape(-1, O.VOCAB)[pos]
        tv, ti = rows.topk(8, dim=-1)
        assert close(tv.numpy(), g["eval/mlm/top8_val"])
        assert (ti.numpy()[:, 0] == g["eval/mlm/top8_idx"][:, 0]).mean() > 0.99
    ls = O.losses(out, batch)
    for k, v in ls.items():
        assert abs(float(v) - float(g[f"eval/loss/{k}"])) <= TOL * max(1.0, abs(float(g[f"eval/loss/{k}"]))), k


@pytest.mark.parametrize("name", ["tiny256_pretrain", "small96_T20_ragged", "large96_T20"])
def test_train_step_matches_reference_golden(golden_dir, name):
    """loss + every parameter gradient of one train-mode step with injected dropout/DropPath masks."""
    from tests.golden.make_golden import make_masks
    torch.set_num_threads(min(8, usable_cores()))
    c = CASES[name]
    g, seed, B, img, T, dp = load(golden_dir, name)
    cfg = O.Cfg(c["variant"], c["lt"], 224, 768, T, dp)
    sd = O.filled_state_dict(cfg, seed)
    batch = O.to_torch_batch(filler.make_batch(seed, B, img, T))
    step_idx = 1
    masks = make_masks(cfg, B, T, seed + step_idx)
    sdg = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v)
           for k, v in sd.items() if k != O.TIED[0]}
    sdg[O.TIED[0]] = sdg[O.TIED[1]]
    bn = {}
    ls, _ = O.step_loss(sdg, cfg, batch, step_idx, train=True, masks=masks, bn_out=bn)
    ls["total_loss"].backward()
    for k, v in ls.items():
        ref = float(g[f"train{step_idx}/loss/{k}"])
        assert abs(float(v) - ref) <= TOL * max(1.0, abs(ref)), k
    n = 0
    for k, v in sdg.items():
        gk = f"train{step_idx}/grad/{k}/norm"
        if gk not in g.files:
            continue
        ref = float(g[gk])
        assert abs(v.grad.double().norm().item() - ref) <= 2e-3 * max(1e-6, ref), k
        n += 1
    assert n > 100
    for k, v in bn.items():
        assert close(v.numpy(), g[f"train{step_idx}/bn/{k}"]), k


@pytest.mark.parametrize("name,variant,lt", [("tiny256_loop", "pvlt_tiny", dict(mlm=1, itm=1, t2i=1, cls=0)),
                                             ("tiny256_ft_loop", "pvlt_tiny", dict(mlm=0, itm=0, t2i=0, cls=1))])
def test_engine_loop_matches_reference_golden(golden_dir, name, variant, lt):
    """oracle.train_loop (engine iteration order + AdamW with timm's split) against the REAL reference model driven the same
    way: per-iteration losses, every parameter's delta after the AdamW steps, BatchNorm counters."""
    from tests.golden.make_golden import loop_view, make_masks
    torch.set_num_threads(min(8, usable_cores()))
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, B, img, T, dp, iters, lr, wd = g["meta"]
    seed, B, img, T, iters = int(seed), int(B), int(img), int(T), int(iters)
    cfg = O.Cfg(variant, lt, 224, 768, T, float(dp))
    sd = O.filled_state_dict(cfg, seed)
    batches = [O.to_torch_batch(filler.make_batch(seed + 100 * it, B, img, T)) for it in range(iters)]
    masks = [make_masks(cfg, B, T, seed + it) for it in range(iters)]
    hist, sd_o = O.train_loop(sd, cfg, batches, masks, float(lr), float(wd))
    for it, h in enumerate(hist):
        for k, v in h.items():
            ref = float(g[f"loop/loss/{it}/{k}"])
            assert abs(v - ref) <= TOL * max(1.0, abs(ref)), (it, k)
    n = 0
    for k, v in sd_o.items():
        if k == O.TIED[0]:
            continue
        if not v.is_floating_point():
            assert int(v) == int(g[f"loop/int/{k}"]) == iters
            continue
        refn = float(g[f"loop/delta/{k}/norm"])
        if refn == 0.0:
            continue
        d = loop_view(k, (v - sd[k]).double())
        assert abs(d.norm().item() - refn) <= 1e-2 * refn, k
        rs = g[f"loop/delta/{k}/sample"]
        assert np.linalg.norm(sample(d, 64) - rs) <= 2e-2 * np.linalg.norm(rs) + 1e-12, k
        n += 1
    assert n > 50
