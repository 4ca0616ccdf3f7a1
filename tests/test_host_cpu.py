"""CPU: host-side logic -- the C-ABI library loads and exports every symbol include/mvlt_hip.h declares (no compute
calls without a GPU), the model's state_dict schema equals the reference's, the product path fails loudly without a
GPU, loss composition equals the oracle's, metric bookkeeping, and the reference import surface."""
import os
import re

import pytest
import torch

from oracle import filler
from oracle import pvlt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import mvlt_amd._lib as L
    hdr = open(os.path.join(ROOT, "include", "mvlt_hip.h")).read()
    declared = set(re.findall(r"\b(mvlt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(L.lib, name), f"{name} declared in mvlt_hip.h but not exported by libmvlt_hip.so"
    assert set(L.EXPORTS) <= declared
    assert L.lib.mvlt_abi_version() == 1
    assert L.lib.mvlt_sizeof(b"no_such_struct") == -1


def test_bad_arguments_return_error_codes_not_crashes():
    import ctypes as C
    import mvlt_amd._lib as L
    a = L.GemmNTArgs()
    assert L.lib.mvlt_gemm_nt(C.byref(a), None) < 0
    assert b"null operand" in L.lib.mvlt_last_error()
    b = L.AttnArgs()
    assert L.lib.mvlt_sr_attention_fwd(C.byref(b), None) < 0


@pytest.mark.parametrize("variant", ["pvlt_tiny", "pvlt_small", "pvlt_medium", "pvlt_large"])
@pytest.mark.parametrize("lt", [dict(mlm=1, itm=1, t2i=1, cls=0), dict(mlm=0, itm=0, t2i=0, cls=1), dict(mlm=1, itm=1, t2i=1, cls=1)])
def test_state_dict_schema_equals_reference(variant, lt):
    from mvlt_amd import pvlt
    if variant in ("pvlt_medium", "pvlt_large") and lt["cls"] == 0:
        pytest.skip("one loss_type per big variant is enough")
    m = getattr(pvlt, variant)(pretrained=True, token_hidden_size=768, num_text_tokens=128, loss_type=lt, pretrained_pth=None,
                               drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3)
    cfg = O.Cfg(variant, lt, 224, 768, 128, 0.1)
    want = [(k, tuple(s)) for k, s in O.param_shapes(cfg).items()]
    got = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert got == want
    if lt["mlm"]:
        sd = m.state_dict()
        assert sd["mlm_head.mlm_decoder.weight"].data_ptr() == sd["text_embeddings.word_embeddings.weight"].data_ptr()
    assert hasattr(m, "default_cfg")
    assert m.dpr == cfg.dpr


def test_constructor_errors_match_reference():
    from mvlt_amd import pvlt
    with pytest.raises(KeyError):          # loss_type must hold all four keys (reference libs/pvlt.py:242-275)
        pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=8, loss_type=dict(mlm=1, itm=1), pretrained_pth=None)
    with pytest.raises(AssertionError):    # img/patch divisibility (reference libs/pvlt.py:158)
        pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=8, loss_type=dict(mlm=1, itm=1, t2i=0, cls=0),
                       pretrained_pth=None, img_size=230)


def test_load_legacy_checkpoint_and_pretrained_pth(tmp_path):
    from mvlt_amd import pvlt
    lt = dict(mlm=1, itm=1, t2i=0, cls=0)
    cfg = O.Cfg("pvlt_tiny", lt, 224, 768, 16, 0.0)
    sd = O.filled_state_dict(cfg, 3)
    sd["text_embeddings.position_ids"] = torch.arange(512)[None]       # transformers==4.10.2-era key
    m = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=16, loss_type=lt, pretrained_pth=None)
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.state_dict()["block1.0.attn.q.weight"], sd["block1.0.attn.q.weight"])
    # PVT-v1 style backbone-only checkpoint through pretrained_pth (non-strict, reference libs/pvlt.py:426-428)
    backbone = {k: v for k, v in sd.items() if k.startswith(("patch_embed", "pos_embed", "block"))}
    backbone["cls_token"] = torch.zeros(1, 1, 512)
    pth = tmp_path / "pvt_tiny.pth"
    torch.save(backbone, pth)
    m2 = pvlt.pvlt_tiny(pretrained=True, token_hidden_size=768, num_text_tokens=16, loss_type=lt, pretrained_pth=str(pth))
    assert torch.equal(m2.state_dict()["block4.1.mlp.fc2.weight"], sd["block4.1.mlp.fc2.weight"])


def test_no_cpu_fallback():
    from mvlt_amd import ops, pvlt
    from mvlt_amd._lib import MVLTError
    m = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=16, loss_type=dict(mlm=1, itm=1, t2i=0, cls=0), pretrained_pth=None)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 16, dtype=torch.long))
    with pytest.raises(MVLTError):
        x = torch.zeros(8, 64)
        ops.gemm_nt(x, x, x.clone(), 8, 8, 64, 64, 64, 64)


def test_loss_composition_equals_oracle():
    from mvlt_amd.engine import compute_losses
    torch.manual_seed(0)
    B, T = 3, 16
    nb = filler.make_batch(5, B, 64, T)
    batch = O.to_torch_batch(nb)
    out = dict(mlm_logits=torch.randn(B, T, 30522), itm_logits=torch.randn(B, 1, 2), sup_cls_logits=torch.randn(B, 1, 48),
               sub_cls_logits=torch.randn(B, 1, 122), t2i_logits=torch.rand(B, 3, 64, 64))
    total, parts = compute_losses(out, batch["image"], batch["mlm_labels"], batch["itm_labels"], batch["sup_cls_labels"], batch["sub_cls_labels"])
    ref = O.losses(out, batch)
    assert abs(float(total) - float(ref["total_loss"])) < 1e-5
    for k in ("loss_mlm", "loss_itm", "loss_sup_cls", "loss_sub_cls", "loss_t2i"):
        assert abs(float(parts[k]) - float(ref[k])) < 1e-5
    # fused-loss form
    out2 = dict(out, mlm_logits=None, mlm_loss=ref["loss_mlm"])
    total2, _ = compute_losses(out2, batch["image"], batch["mlm_labels"], batch["itm_labels"], batch["sup_cls_labels"], batch["sub_cls_labels"])
    assert abs(float(total2) - float(total)) < 1e-5


def test_metric_logger_surface():
    from mvlt_amd.metrics import MetricLogger, SmoothedValue
    lg = MetricLogger(delimiter="  ")
    lg.add_meter("lr", SmoothedValue(window_size=1, fmt="{value:.6f}"))
    for i in range(30):
        lg.update(total_loss=float(i), lr=0.1)
    assert lg.meters["total_loss"].global_avg == pytest.approx(14.5)
    assert lg.total_loss.median == pytest.approx(19.0)      # window of 20 -> values 10..29 (torch lower median)
    assert "lr: 0.100000" in str(lg)
    assert list(lg.log_every(range(3), 10, "x")) == [0, 1, 2]
    lg.synchronize_between_processes("cpu")                  # no process group: no-op


def test_reference_import_surface():
    import engine_grid_masking as E
    for name in ("evaluate_vl", "train_one_epoch_vl", "visual_vl", "evaluate_retrieval", "evaluate_recognition", "train_one_epoch"):
        assert callable(getattr(E, name))
    from libs import pvlt
    assert pvlt.__all__ == ['pvlt_tiny', 'pvlt_small', 'pvlt_medium', 'pvlt_large']
    import inspect
    sig = inspect.signature(E.train_one_epoch_vl)
    assert list(sig.parameters)[:7] == ["model", "criterion", "data_loader", "optimizer", "device", "epoch", "loss_scaler"]
    assert sig.parameters["max_norm"].default == 0 and sig.parameters["fp32"].default is False


def test_masked_positions_definition():
    lab = torch.tensor([[-1, 5, -1], [7, -1, 0]])
    assert O.masked_positions(lab).tolist() == [1, 3, 5]


def test_zero_pool_hands_out_zeroed_disjoint_scratch():
    """params.ZeroPool: step-scoped zero-initialised scratch (one fill per step instead of one per buffer)."""
    import torch
    from mvlt_amd.params import ZeroPool
    pool = ZeroPool(torch.device("cpu"))
    pool.reset()
    a = pool.take((3, 5), torch.float32)          # first step: nothing pooled yet -> plain zeros, need recorded
    a += 1.0
    pool.reset()                                  # second step: buffer sized from the recorded need
    b = pool.take((3, 5), torch.float32)
    c = pool.take((7,), torch.bfloat16)
    assert b.abs().sum() == 0 and c.float().abs().sum() == 0
    b += 2.0
    c += 1.0
    assert b.data_ptr() != c.data_ptr() and float(c.float().sum()) == 7.0 and float(b.sum()) == 30.0
    pool.reset()
    d = pool.take((3, 5), torch.float32)
    assert d.abs().sum() == 0                     # re-zeroed
    big = pool.take((1 << 22,), torch.float32)    # does not fit: falls back to torch.zeros, pool grows next step
    assert big.numel() == 1 << 22 and big.abs().sum() == 0


def test_no_barrier_with_undrained_lds_writes():
    """tools/isa_barrier_check.py over every kernel source (hipcc -S, no GPU): no s_barrier may be reached over a loop back-edge
    while the wave's own LDS writes are still in flight (the attention-backward race of round 1)."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_barrier_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
