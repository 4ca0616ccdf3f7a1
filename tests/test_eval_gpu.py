"""GPU: the eval callers (mvlt_amd/evaluate.py) on the HIP model against numbers the REAL reference produced for the same
weights and batches (tests/golden/tiny128_eval.npz, written by make_golden.py:run_eval_case with the reference's own
libs/vl_scores.py and sklearn): evaluate_vl's per-epoch averages, the 101-candidate ITM ranking of evaluate_retrieval, the
predictions and F1 / accuracy numbers of evaluate_recognition."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import filler
from oracle import pvlt_oracle as O

pytestmark = pytest.mark.gpu
LT = dict(mlm=1, itm=1, t2i=1, cls=1)


def _setup(golden_dir, dtype):
    from mvlt_amd import pvlt
    g = np.load(os.path.join(golden_dir, "tiny128_eval.npz"))
    seed, B, img, T, nb, nq, nc = (int(v) for v in g["meta"])
    cfg = O.Cfg("pvlt_tiny", LT, 224, 768, T, 0.0)
    model = pvlt.pvlt_tiny(pretrained=True, token_hidden_size=768, num_text_tokens=T, loss_type=LT, pretrained_pth=None,
                           drop_path_rate=0.0, drop_rate=0.0, num_classes=1000, in_chans=3, compute_dtype=dtype)
    model.load_state_dict(O.filled_state_dict(cfg, seed), strict=True)
    model.cuda()
    return model, g, seed, B, img, T, nb, nq, nc


def _vl_batches(g, seed, B, img, T, nb):
    out = []
    for it in range(nb):
        b = O.to_torch_batch(filler.make_batch(seed + 100 * it, B, img, T))
        for k in ("mlm_labels", "itm_labels", "sup_cls_labels", "sub_cls_labels"):      # the half-right labels of the golden run
            b[k] = torch.from_numpy(g[f"vl/{it}/{k}"])
        out.append(b)
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_evaluate_vl_matches_reference(golden_dir, parity, dtype):
    from mvlt_amd.evaluate import evaluate_vl
    model, g, seed, B, img, T, nb, nq, nc = _setup(golden_dir, dtype)
    args = types.SimpleNamespace(loss_type=LT)
    res = evaluate_vl(_vl_batches(g, seed, B, img, T, nb), model, torch.device("cuda:0"), args)
    assert set(res) >= {"mlm_acc", "itm_acc", "sup_cls_acc", "sub_cls_acc", "t2i_psnr", "total_loss"}
    fp32 = dtype == torch.float32
    # accuracies: the labels were built so that half the argmaxes are right; one flipped argmax moves an accuracy by 1/count
    for k, step in (("mlm_acc", 1.0 / 30), ("itm_acc", 1.0 / (B * nb)), ("sup_cls_acc", 1.0 / (B * nb)), ("sub_cls_acc", 1.0 / (B * nb))):
        ref = float(g[f"vl/avg/{k}"])
        assert parity(f"vl/{k}", abs(res[k] - ref), 1e-6 if fp32 else 2.01 * step), (k, res[k], ref)
    ref = float(g["vl/avg/t2i_psnr"])
    assert parity("vl/t2i_psnr(dB)", abs(res["t2i_psnr"] - ref), 0.01 if fp32 else 0.2), (res["t2i_psnr"], ref)
    ref = float(g["vl/avg/total_loss"])
    assert parity("vl/total_loss", abs(res["total_loss"] - ref) / abs(ref), 1e-3 if fp32 else 2e-2), (res["total_loss"], ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_evaluate_retrieval_ranking_matches_reference(golden_dir, parity, dtype):
    from mvlt_amd.evaluate import evaluate_retrieval, rank_candidates
    from tests.golden.make_golden import retrieval_query
    model, g, seed, B, img, T, nb, nq, nc = _setup(golden_dir, dtype)
    model.eval()
    dev = torch.device("cuda:0")
    fp32 = dtype == torch.float32
    items = [retrieval_query(seed, qi, nc, img, T) for qi in range(nq)]
    for qi, item in enumerate(items):
        with torch.no_grad():
            score, order, rank0 = rank_candidates(model, item["images_101"].to(dev).squeeze(), item["ori_input_ids_101"].to(dev).squeeze())
        ref = g[f"retr/{qi}/score"]
        assert parity(f"retr/score{qi}", np.abs(score.cpu().numpy() - ref).max(), 1e-3 if fp32 else 2e-2)
        ref_order = g[f"retr/{qi}/order"]
        if fp32:
            # identical order wherever neighbouring reference scores are further apart than the fp32 tolerance
            srt = ref[ref_order]
            gaps_ok = np.concatenate([[True], (srt[:-1] - srt[1:]) > 2e-3]) & np.concatenate([(srt[:-1] - srt[1:]) > 2e-3, [True]])
            assert np.array_equal(order.cpu().numpy()[gaps_ok], ref_order[gaps_ok])
        assert parity(f"retr/rank0-{qi}", abs(rank0 - int(g[f"retr/{qi}/rank0"])), 0 if fp32 else 6), (rank0, int(g[f"retr/{qi}/rank0"]))
    args = types.SimpleNamespace(eval_retrieval_tir=True, eval_retrieval_itr=False)
    res = evaluate_retrieval(items, model, dev, args)
    hits = {k: sum(int(g[f"retr/{qi}/rank0"]) < k for qi in range(nq)) for k in (1, 5, 10)}
    if fp32:
        assert res == {f"acc@{k}": v / 1000 for k, v in hits.items()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_evaluate_recognition_matches_reference(golden_dir, parity, dtype):
    from mvlt_amd.evaluate import evaluate_recognition
    model, g, seed, B, img, T, nb, nq, nc = _setup(golden_dir, dtype)
    loader = []
    for it, b in enumerate(_vl_batches(g, seed, B, img, T, nb)):
        loader.append(dict(images=b["image"], ori_input_ids=b["ori_input_ids"], sup_cls_labels=b["sup_cls_labels"],
                           sub_cls_labels=b["sub_cls_labels"], info_list=[f"{it}_{j}" for j in range(B)]))
    res = evaluate_recognition(loader, model, torch.device("cuda:0"), types.SimpleNamespace())
    fp32 = dtype == torch.float32
    for tag in ("sup", "sub"):
        ref_p = np.concatenate([g[f"recog/{it}/{tag}_pred"] for it in range(nb)])
        got_p = np.asarray(res[f"{tag}_cls_preds"])
        top2 = np.concatenate([g[f"recog/{it}/{tag}_top2"] for it in range(nb)])
        clear = (top2[:, 0] - top2[:, 1]) > (1e-3 if fp32 else 5e-2) * np.abs(top2).max()     # predictions with a clear margin must agree
        assert np.array_equal(got_p[clear], ref_p[clear]), (tag, got_p, ref_p)
        parity(f"recog/{tag}_pred_disagreement", float((got_p != ref_p).mean()), 0.0 if fp32 else 0.2)
        if np.array_equal(got_p, ref_p):
            ref_m = g[f"recog/{tag}_metrics"]
            got_m = [res[f"{tag}_{k}"] for k in ("accuracy", "macro_f1", "micro_f1", "weighted_f1")]
            assert np.allclose(got_m, ref_m, atol=1e-9), (tag, got_m, ref_m)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_decoder_with_folded_batchnorm_tracks_the_unfolded_one(golden_dir, parity, monkeypatch, dtype):
    """Inference mode folds each BatchNorm of the MIM decoder into its conv (mvlt_amd/mim.py:MimStep._folded, reference
    libs/vl_heads.py:147-152 under model.eval()).  The folded forward must equal the pass that keeps z and normalises it -- and keep
    doing so after the running statistics and the weights have moved (the folded copies are cached)."""
    from mvlt_amd import mim
    from mvlt_amd.engine import train_step
    from mvlt_amd.optim import FusedAdamW
    model, g, seed, B, img, T, nb, nq, nc = _setup(golden_dir, dtype)
    b = {k: v.cuda() for k, v in O.to_torch_batch(filler.make_batch(seed + 7, B, img, T)).items()}
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2

    def both():
        model.eval()
        with torch.no_grad():
            monkeypatch.setattr(mim, "_NO_BN_FOLD", False)
            a = model(b["image"], b["input_ids"])["t2i_logits"].float()
            monkeypatch.setattr(mim, "_NO_BN_FOLD", True)
            c = model(b["image"], b["input_ids"])["t2i_logits"].float()
        return a, c

    a0, c0 = both()
    assert parity("bn-fold/initial", ((a0 - c0).norm() / c0.norm()).item(), tol)
    # one training step: batch statistics move the running buffers (through the C ABI, no torch version bump), AdamW moves the weights
    model.train()
    opt = FusedAdamW(model, lr=1e-2, weight_decay=0.0)
    total, _ = train_step(model, b, 0, True)
    total.backward()
    opt.step()
    a1, c1 = both()
    assert ((c1 - c0).norm() / c0.norm()).item() > 10 * tol            # the step did change the decoder's output ...
    assert parity("bn-fold/after-step", ((a1 - c1).norm() / c1.norm()).item(), tol)      # ... and the folded copies followed
