"""GPU, two processes sharing the one MI355X of the test box (gloo backend on device tensors -- RCCL refuses two ranks on one
device): the data-parallel path with the REAL model and kernels.  Each rank steps its own batch through
mvlt_amd.dist.DataParallel (ranges all-reduced as the backward announces them) or through stock
torch.nn.parallel.DistributedDataParallel (what reference main_vl.py:298-302 constructs); the gradients both ranks end up
with must be the mean of the two single-rank gradients -- DDP's mean-of-means, including unequal masked-token counts per
rank -- and the parameters after a FusedAdamW step must match a single-process step on that mean gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
LT = dict(mlm=1, itm=1, t2i=1, cls=0)
T, IMG, B = 32, 64, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed=9):
    from mvlt_amd import pvlt
    from oracle import pvlt_oracle as O
    cfg = O.Cfg("pvlt_tiny", LT, 224, 768, T, 0.0)
    m = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=T, loss_type=LT, pretrained_pth=None, drop_path_rate=0.0,
                       compute_dtype=torch.float32)
    m.load_state_dict(O.filled_state_dict(cfg, seed), strict=True)
    m.cuda()
    m.train()
    m.injected_masks = dict(bert=torch.ones(B, T, 768), droppath=[torch.ones(B)] * 8, droppath2=[torch.ones(B)] * 8)
    return m


def _batch(rank):
    from oracle import filler
    from oracle import pvlt_oracle as O
    b = O.to_torch_batch(filler.make_batch(70 + rank, B, IMG, T))
    if rank == 1:                       # make the masked-token counts differ between the ranks (mean of means != global mean)
        b["mlm_labels"][0, 3] = b["ori_input_ids"][0, 3]
        b["mlm_labels"][1, 4] = b["ori_input_ids"][1, 4]
    return {k: v.cuda() for k, v in b.items()}


def _grads(model, batch, idx=1):
    from mvlt_amd.engine import train_step
    total, _ = train_step(model, batch, idx, True)
    for p in model.parameters():
        p.grad = None
    total.backward()
    torch.cuda.synchronize()
    return total


def _worker(rank, world, port, kind, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvlt_amd.dist import DataParallel
        from mvlt_amd.optim import FusedAdamW
        core = _model(seed=9 + rank)                 # different start weights: the wrapper must broadcast rank 0's
        p_init = _model(seed=9).state_dict()
        if kind in ("ours", "ours_scaler"):
            model = DataParallel(core)
        else:
            model = torch.nn.parallel.DistributedDataParallel(core, device_ids=[0])
        opt = FusedAdamW(core, lr=1e-3, weight_decay=0.05)
        batch = _batch(rank)
        S = core.store
        launches = []
        if kind.startswith("ours_scaler"):
            from mvlt_amd import ops as _ops
            _adamw = _ops.adamw_step
            _ops.adamw_step = lambda p_, *a_, **k_: (launches.append(p_.numel()), _adamw(p_, *a_, **k_))[1]
            # the engine's own path: BF16Scaler runs backward + FusedAdamW.step with the 1/world factor riding in the optimizer kernel
            # (G keeps the rank SUM; nothing may be owed once the call returns)
            from mvlt_amd.engine import BF16Scaler, train_step
            total, _ = train_step(model, batch, 1, True)
            for p_ in model.parameters():
                p_.grad = None
            p_before = S.P.clone()
            BF16Scaler()(total, opt)
            _ops.adamw_step = _adamw                 # (the single-process reference below steps through the plain entry)
            assert S.pending_grad_scale == 1.0 and not S.scale_in_optimizer and not S.grad_works
            g_mine = (S.G / world).cpu()
            # the same step as ONE launch over everything, on the same (now final) gradient sums, moments from zero, the optimizer's own scalar row
            # (it still holds lr, betas, bias corrections and the 1/world gradient scale of this step): must equal the phased step bit for bit
            p_one, m_one, v_one = p_before.clone(), torch.zeros_like(S.P), torch.zeros_like(S.P)
            _adamw(p_one, S.G, m_one, v_one, None if S.C is None else torch.empty_like(S.C), S.total, opt._hp, opt._wd_mask)
            torch.cuda.synchronize()
            same_as_one_launch = bool(torch.equal(p_one, S.P) and torch.equal(m_one, opt._m) and torch.equal(v_one, opt._v))
        else:
            _grads(model, batch)
            S.sync_grads()
            scale = S.pending_grad_scale             # 1.0: outside the engine's scaler the wrapper hands out final (mean) gradients
            assert scale == 1.0
            g_mine = (S.G * scale).cpu()
            opt.step()
        torch.cuda.synchronize()
        # single-process reference, computed by every rank for itself: both batches through a plain model with rank 0's weights,
        # gradients averaged, one FusedAdamW step on the mean
        ref = _model(seed=9)
        gs = []
        for r in range(world):
            _grads(ref, _batch(r))
            gs.append(ref.store.G.clone())
        gmean = sum(gs) / world
        ropt = FusedAdamW(ref, lr=1e-3, weight_decay=0.05)
        ref.store.G.copy_(gmean)
        for n_, p_ in ref.store.params.items():
            p_.grad = ref.store.grad(n_)
        ropt.step()
        torch.cuda.synchronize()
        e_g = ((g_mine.double() - gmean.cpu().double()).norm() / gmean.cpu().double().norm()).item()
        # error of the AdamW step relative to the step itself (AdamW turns rounding-level gradient differences on elements whose
        # gradient is ~0 -- e.g. the key half of attn.kv.bias -- into O(lr) differences: the bound is 5 % of the update's norm)
        upd = sum(((ref.state_dict()[k].double() - v.double().cuda()) ** 2).sum() for k, v in p_init.items() if v.is_floating_point()).sqrt()
        e_p = ((S.P.double() - ref.store.P.double()).norm() / upd).item()
        import hashlib
        q.put(dict(rank=rank, e_g=e_g, e_p=e_p, p_sum=float(S.P.double().sum()), p_abs=float(S.P.double().abs().sum()),
                   p_sha=hashlib.sha256(S.P.cpu().numpy().tobytes()).hexdigest(), launches=launches, total=S.total,
                   same_as_one_launch=same_as_one_launch if kind == "ours_scaler" else None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["ours", "ours_scaler", "torch_ddp"])
def test_two_ranks_average_gradients_like_ddp(kind, parity):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = res
    for r in res:
        assert parity(f"dp-{kind}/grad-rank{r['rank']}", r["e_g"], 1e-5), r
        assert parity(f"dp-{kind}/params-after-step-rank{r['rank']}", r["e_p"], 5e-2), r
    assert abs(r0["p_sum"] - r1["p_sum"]) <= 1e-9 * r0["p_abs"]           # both ranks hold the same parameters after the step


def _run_kind(kind):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_phased_adamw_equals_one_launch_bit_for_bit():
    """VERDICT r4 #5b with the real model and the real kernel: FusedAdamW steps the ranges whose collectives went out during the backward in a first
    launch (after their wait) and the tail in a second; parameters and moments equal, bit for bit, those of ONE launch over everything on the same
    gradient sums, and both ranks end with identical parameters."""
    res = _run_kind("ours_scaler")
    for r in res:
        assert r["same_as_one_launch"], (r["rank"], r["launches"])
        assert len(r["launches"]) >= 2 and sum(r["launches"]) == r["total"], r["launches"]
    assert res[0]["p_sha"] == res[1]["p_sha"]


def test_bench_self_launch_runs_the_rccl_path():
    """`python bench.py --gpus N` with no launcher around it starts its own ranks as child processes BEFORE touching the GPU
    (VERDICT r2 #3).  One GPU here: `--spawn` takes the same route at world size 1 with every collective of the N > 1 path
    issued over RCCL (MVLT_DP_FORCE_COLLECTIVES).  The child's JSON line must say which ranks it saw."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and len(out["per_rank_pairs_s"]) == 1
    assert out["value"] > 0 and out["config"]["parallelism"] == "dp1"
