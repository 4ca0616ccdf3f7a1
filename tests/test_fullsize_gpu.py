"""GPU: BASELINE configuration #2 at its FULL size (pvlt_tiny, 256x256 + 128 tokens, batch 256, bf16 operands) -- the shapes
bench.py times.  The golden fixtures pin this configuration at B = 4 (the CPU reference needs minutes per step beyond that), so
at B = 256 the checks are the size-independent ones (reference engine_grid_masking.py:69-102):
  * one iteration of `train_one_epoch_vl`: every loss finite, and the MLM / ITM losses equal to what the same 256 pairs give when
    they go through the eval-mode model as 64 independent batches of 4 (the configuration the fixtures DO pin);
  * the masked-index selection over the 32768 labels bit-equal to torch.nonzero;
  * the same step (same batch, same draws) repeated with freed memory poisoned in between: every gradient equal to the first run's
    up to fp32 atomic-order noise (a race or an uninitialised read shows up here);
  * round 6: every parameter gradient of a batch-256 step equal to the mean over its four batch-64 quarters (the size at which the goldens
    DO pin the gradients to the reference's), on both numeric paths, for the MLM + ITM heads and for the CLS heads."""
import argparse
import contextlib
import io

import pytest
import torch
import torch.nn.functional as F

import bench

pytestmark = pytest.mark.gpu
B, IMG, T = 256, 256, 128
LT = dict(mlm=1, itm=1, t2i=1, cls=0)


def _model():
    from mvlt_amd import pvlt
    torch.manual_seed(77)
    m = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=T, loss_type=LT, pretrained_pth=None,
                       drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3)
    return m.cuda()


def test_masked_select_equals_nonzero_at_full_size():
    from mvlt_amd import ops
    dev = torch.device("cuda:0")
    labels = bench.synth_batch(B, 32, T, dev, 5)["mlm_labels"].reshape(-1).contiguous()
    assert labels.numel() == 32768
    idx = torch.empty(labels.numel(), device=dev, dtype=torch.int32)
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    ops.masked_select(labels, idx, cnt)
    want = torch.nonzero(labels != -1).flatten().to(torch.int32)
    n = int(cnt.item())
    assert n == want.numel() and n > 1000
    assert torch.equal(idx[:n], want)


def test_full_batch_engine_iteration_matches_chunked_eval(parity):
    import engine_grid_masking as E
    from mvlt_amd.engine import BF16Scaler
    from mvlt_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    model = _model()
    batch = bench.synth_batch(B, IMG, T, dev, 4321)
    # the same 256 pairs as 64 eval-mode batches of 4: full (4, 128, 30522) MLM logits, the reference's CrossEntropyLoss(ignore_index=-1)
    model.eval()
    ce_sum, n_sel, itm_sum = 0.0, 0, 0.0
    with torch.no_grad():
        for c in range(0, B, 4):
            out = model(batch["image"][c:c + 4], batch["input_ids"][c:c + 4])
            lab = batch["mlm_labels"][c:c + 4].reshape(-1)
            ce_sum += F.cross_entropy(out["mlm_logits"].reshape(-1, 30522).float(), lab, ignore_index=-1, reduction="sum").item()
            n_sel += int((lab != -1).sum())
            itm_sum += F.cross_entropy(out["itm_logits"].reshape(-1, 2).float(), batch["itm_labels"][c:c + 4].reshape(-1), reduction="sum").item()
    want_mlm, want_itm = ce_sum / n_sel, itm_sum / B
    # one engine iteration at full size; DropPath / dropout draws replaced by ones so that train mode == eval mode for MLM / ITM
    # (BatchNorm, the only other train / eval difference, lives in the MIM decoder alone)
    model.injected_masks = dict(bert=torch.ones(B, T, 768), droppath=[torch.ones(B)] * 8, droppath2=[torch.ones(B)] * 8)
    opt = FusedAdamW(model, lr=1e-5, weight_decay=0.01)
    with contextlib.redirect_stdout(io.StringIO()):
        stats = E.train_one_epoch_vl(model, None, [batch], opt, dev, 0, BF16Scaler(), None, None, None, True, False,
                                     argparse.Namespace(loss_type=LT))
    torch.cuda.synchronize()
    for k in ("total_loss", "loss_mlm", "loss_itm", "loss_t2i"):
        assert stats[k] == stats[k] and abs(stats[k]) < 1e4, (k, stats[k])
    assert parity("full256/loss_mlm vs 64 eval chunks", abs(stats["loss_mlm"] - want_mlm) / want_mlm, 2e-2), (stats["loss_mlm"], want_mlm)
    assert parity("full256/loss_itm vs 64 eval chunks", abs(stats["loss_itm"] - want_itm) / want_itm, 2e-2), (stats["loss_itm"], want_itm)
    assert abs(stats["total_loss"] - (stats["loss_mlm"] + stats["loss_itm"] + stats["loss_t2i"])) <= 1e-3 * abs(stats["total_loss"])
    assert all(torch.isfinite(p).all() for p in model.parameters())          # the AdamW step of that iteration


@pytest.mark.parametrize("name,img,batch_size,lt", [("pvlt_medium", 384, 64, dict(mlm=1, itm=1, t2i=1, cls=0)),
                                                    ("pvlt_tiny", 256, 256, dict(mlm=0, itm=0, t2i=0, cls=1))], ids=["config4_medium384_b64", "config5_finetune_b256"])
def test_other_configs_full_size_iteration_matches_chunked_eval(parity, name, img, batch_size, lt):
    """BASELINE configurations #4 (pvlt_medium, 384 px: 272 keys per query, 18 stage-3 blocks) and #5 (the CLS-head fine-tune step) at the sizes `bench.py`'s
    `other_configs` runs them (round 5; the fixtures pin them at B = 4): one `train_one_epoch_vl` iteration whose per-head losses equal what the same pairs give as
    eval-mode batches of 4 -- every kernel variant these shapes select (attention with 272 keys on the 6-wave backward, the ragged-free 128-wide GEMMs at 45056 rows,
    the recognition heads) against the configuration the goldens do pin."""
    import engine_grid_masking as E
    from mvlt_amd import pvlt
    from mvlt_amd.engine import BF16Scaler
    from mvlt_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    torch.manual_seed(78)
    model = getattr(pvlt, name)(pretrained=False, token_hidden_size=768, num_text_tokens=T, loss_type=lt, pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0,
                                num_classes=1000, in_chans=3).cuda()
    Bc = batch_size
    batch = bench.synth_batch(Bc, img, T, dev, 4321)
    model.eval()
    acc = dict(mlm=0.0, itm=0.0, sup=0.0, sub=0.0)
    n_sel = 0
    with torch.no_grad():
        for c in range(0, Bc, 4):
            out = model(batch["image"][c:c + 4], batch["input_ids"][c:c + 4])
            if lt["mlm"]:
                lab = batch["mlm_labels"][c:c + 4].reshape(-1)
                acc["mlm"] += F.cross_entropy(out["mlm_logits"].reshape(-1, 30522).float(), lab, ignore_index=-1, reduction="sum").item()
                n_sel += int((lab != -1).sum())
                acc["itm"] += F.cross_entropy(out["itm_logits"].reshape(-1, 2).float(), batch["itm_labels"][c:c + 4].reshape(-1), reduction="sum").item()
            if lt["cls"]:
                acc["sup"] += F.cross_entropy(out["sup_cls_logits"].reshape(-1, 48).float(), batch["sup_cls_labels"][c:c + 4].reshape(-1), reduction="sum").item()
                acc["sub"] += F.cross_entropy(out["sub_cls_logits"].reshape(-1, 122).float(), batch["sub_cls_labels"][c:c + 4].reshape(-1), reduction="sum").item()
    nblk = sum(model.depths)
    model.injected_masks = dict(bert=torch.ones(Bc, T, 768), droppath=[torch.ones(Bc)] * nblk, droppath2=[torch.ones(Bc)] * nblk)
    opt = FusedAdamW(model, lr=1e-5, weight_decay=0.01)
    with contextlib.redirect_stdout(io.StringIO()):
        stats = E.train_one_epoch_vl(model, None, [batch], opt, dev, 0, BF16Scaler(), None, None, None, True, False, argparse.Namespace(loss_type=lt))
    torch.cuda.synchronize()
    assert stats["total_loss"] == stats["total_loss"] and abs(stats["total_loss"]) < 1e4
    if lt["mlm"]:
        assert parity(f"{name}-{img}/loss_mlm vs eval chunks", abs(stats["loss_mlm"] - acc["mlm"] / n_sel) / (acc["mlm"] / n_sel), 2e-2), (stats["loss_mlm"], acc["mlm"] / n_sel)
        assert parity(f"{name}-{img}/loss_itm vs eval chunks", abs(stats["loss_itm"] - acc["itm"] / Bc) / (acc["itm"] / Bc), 2e-2), (stats["loss_itm"], acc["itm"] / Bc)
    if lt["cls"]:
        assert parity(f"{name}-{img}/loss_sup_cls vs eval chunks", abs(stats["loss_sup_cls"] - acc["sup"] / Bc) / (acc["sup"] / Bc), 2e-2), (stats["loss_sup_cls"], acc["sup"] / Bc)
        assert parity(f"{name}-{img}/loss_sub_cls vs eval chunks", abs(stats["loss_sub_cls"] - acc["sub"] / Bc) / (acc["sub"] / Bc), 2e-2), (stats["loss_sub_cls"], acc["sub"] / Bc)
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_full_batch_step_is_repeatable(parity):
    from mvlt_amd.engine import train_step
    dev = torch.device("cuda:0")
    model = _model()
    model.train()
    batch = bench.synth_batch(B, IMG, T, dev, 1)
    ref, worst = None, (0.0, "")
    for it in range(3):
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(1234)
        torch.cuda.manual_seed(1234)
        model._rng_calls = 0                    # the step's own Philox draws (BERT dropout, DropPath) restart with the seed
        junk = [torch.full((1 << 26,), float("nan"), device=dev) for _ in range(8)]          # poison freed memory (2 GB)
        del junk
        total, _ = train_step(model, batch, 1, True)
        total.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(total)
        cur = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
        if ref is None:
            ref = cur
            assert len(ref) > 150
            continue
        for k, v in ref.items():
            d = (cur[k] - v).norm().item() / max(v.norm().item(), 1e-20)
            assert d == d, k
            worst = max(worst, (d, k))
    # Measured on the r03 build: 3e-3 .. 8e-3 on the attention q / kv parameters of stages 1-2, 1e-5 and below elsewhere.  The stage-1
    # dK / dV sums (one head: B * heads < 512) go through fp32 atomics before their bf16 cast, so an order-dependent last fp32 bit can flip
    # a bf16 rounding (4e-3 of that element), and the q / k gradients are small differences of large terms.  A race or an uninitialised
    # read shows up as NaN or as deviations of order one (the round-2 dQ race: 2-4 stale rows = 3e-1 on block1 gradients).
    assert parity(f"full256/repeat-step worst gradient deviation ({worst[1]})", worst[0], 3e-2), worst


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("heads,lt,variant,img,bsz,nchunk", [("mlm+itm", dict(mlm=1, itm=1, t2i=0, cls=0), "pvlt_tiny", IMG, B, 4),
                                                             ("cls", dict(mlm=0, itm=0, t2i=0, cls=1), "pvlt_tiny", IMG, B, 4),
                                                             ("mlm+itm", dict(mlm=1, itm=1, t2i=0, cls=0), "pvlt_medium", 384, 64, 8)])
def test_full_batch_gradients_equal_the_mean_of_its_chunks(parity, heads, lt, variant, img, bsz, nchunk, dtype):
    """The BACKWARD pass at the benchmark's size against the size the goldens pin: every loss of the step is a mean over pairs, so the
    gradients of one batch-256 step equal the mean of the gradients of its four batch-64 quarters -- same weights, same per-pair dropout / DropPath draws (injected) --
    and batch 64 is the size at which `tiny256_pretrain_b64` / `tiny256_ft_b64` meet the reference's own gradients.  The quarters run the small-M kernel variants
    (fewer m-splits, other tile rounds, more query chunks per head), the full batch the large-M ones, so a variant that only the bench's shapes select is
    compared with one that is pinned.  Heads: MLM + ITM (the MLM mean is over the selected positions: the labels of the first quarter are reused by the others so
    that every quarter has the same count) and the two CLS heads (BASELINE configuration #5); the MIM decoder stays out -- its BatchNorm uses batch statistics,
    for which no such identity holds (at full size it is covered by the loss check and the repeatability test above, and -- forward, BN statistics, every gradient --
    against PyTorch's own ops at batch 256 by tests/test_model_gpu.py::test_mim_decoder_hip_vs_torch_twin).  BASELINE configuration #4 the same way:
    pvlt_medium at 384 px, batch 64 against its eight batch-8 chunks, the size of `medium384_pretrain_b8`."""
    from mvlt_amd import pvlt
    from mvlt_amd.engine import train_step
    dev = torch.device("cuda:0")
    torch.manual_seed(78)
    B, Q = bsz, bsz // nchunk
    model = getattr(pvlt, variant)(pretrained=False, token_hidden_size=768, num_text_tokens=T, loss_type=lt, pretrained_pth=None,
                                   drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda()
    if dtype == torch.float32:
        model.set_compute_dtype(torch.float32)
    model.train()
    batch = bench.synth_batch(B, img, T, dev, 99)
    batch["mlm_labels"] = batch["mlm_labels"][:Q].repeat(nchunk, 1).contiguous()
    g = torch.Generator().manual_seed(5)
    nblk = sum(model.depths) if hasattr(model, "depths") else {"pvlt_tiny": 8, "pvlt_medium": 28}[variant]
    dpr = [0.1 * k / (nblk - 1) for k in range(nblk)]
    masks = dict(bert=(torch.rand(B, T, 768, generator=g) >= 0.1).float(),
                 droppath=[(torch.rand(B, generator=g) >= r).float() for r in dpr], droppath2=[(torch.rand(B, generator=g) >= r).float() for r in dpr])

    def step(lo, hi):
        for p in model.parameters():
            p.grad = None
        model.injected_masks = dict(bert=masks["bert"][lo:hi], droppath=[m[lo:hi] for m in masks["droppath"]], droppath2=[m[lo:hi] for m in masks["droppath2"]])
        total, _ = train_step(model, {k: v[lo:hi].contiguous() for k, v in batch.items()}, 0, False)
        total.backward()
        torch.cuda.synchronize()
        return float(total.detach()), {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.grad is not None}

    loss_full, g_full = step(0, B)
    loss_q, g_q = 0.0, None
    for c in range(nchunk):
        l, gc = step(c * Q, (c + 1) * Q)
        loss_q += l / nchunk
        g_q = gc if g_q is None else {k: g_q[k] + gc[k] for k in gc}
    assert len(g_full) > (150 if lt["mlm"] else 120) and set(g_full) == set(g_q)
    tag = f"{variant} b{B}[{heads}]"
    # measured: 8e-7 .. 2.2e-6 (fp32 path); 1.9e-3 / 2.6e-3 on pvlt_tiny, 7.3e-3 on pvlt_medium's 28 blocks (bf16 path: bf16 partial tiles of the weight-gradient reductions cut at
    # other rows, q / kv gradients small differences of large terms) -- bounds at 4-10x that
    ltol, gtol = (1e-5, 2e-5) if dtype == torch.float32 else (2e-3, 3e-2)
    assert parity(f"{tag}/loss vs mean of {nchunk} chunks", abs(loss_full - loss_q) / abs(loss_q), ltol), (loss_full, loss_q)
    worst = (0.0, "")
    for k, v in g_full.items():
        n = v.norm().item()
        if n < 1e-9:
            continue
        worst = max(worst, (((g_q[k] / nchunk) - v).norm().item() / n, k))
    assert parity(f"{tag}/gradients vs mean of {nchunk} chunks, worst parameter ({worst[1]})", worst[0], gtol), worst
