"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (mvlt_amd/dist.py) -- ranges announced in
gradient-ready order are all-reduced as they complete, the rest at the end, and the result is the DDP average
(reference main_vl.py:298-302 semantics); rank-0 parameters are broadcast at start."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(nn.Module):
    """smallest module with the stage-named parameters the store's range logic looks for"""

    def __init__(self):
        super().__init__()
        from mvlt_amd.params import FlatStore, Holder
        self.pos_embed1 = nn.Parameter(torch.zeros(1, 4, 8))
        for i in range(4):
            setattr(self, f"patch_embed{i+1}", Holder(weight=(8, 8), bias=(8,)))
            setattr(self, f"text_embed{i+1}", Holder(weight=(8, 8)))
            setattr(self, f"block{i+1}", Holder(weight=(16, 8), bias=(3,)))
        self.head = Holder(weight=(5, 8))
        self._store = FlatStore(self, torch.float32)

    @property
    def store(self):
        return self._store


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvlt_amd.dist import DataParallel, allreduce_meter
        torch.manual_seed(100 + rank)
        m = _Toy()
        for p in m.parameters():
            p.data.normal_()
        S = m.store
        S.materialize(torch.device("cpu"))
        m.register_buffer("running_mean", torch.full((5,), float(rank + 1)))
        m.register_buffer("running_var", torch.full((3, 2), 10.0 * (rank + 1)))
        m.register_buffer("num_batches_tracked", torch.tensor(rank + 7))           # integer buffers stay per rank
        dp = DataParallel(m, broadcast_buffers=False)
        dp._sync_init()
        dp._sync_buffers()
        ok_buf = bool((m.running_mean == 1.0).all() and (m.running_var == 10.0).all() and int(m.num_batches_tracked) == rank + 7)
        p0 = S.P.clone()
        # emulate one backward: every rank writes rank-dependent gradients, stages complete 4 -> 1
        S.G.copy_(torch.arange(S.total, dtype=torch.float32) * (rank + 1))
        for i in (3, 2, 1, 0):
            S.announce_stage(i)
        S._finalize()
        expect = torch.arange(S.total, dtype=torch.float32) * (sum(r + 1 for r in range(world)) / world)
        ok_grad = torch.allclose(S.G, expect)
        lo, hi = S.stage_range(2)
        names = [n for n, (o, k, _) in S.offsets.items() if lo <= o < hi]
        cnt, tot = allreduce_meter(3, 1.5 * (rank + 1), "cpu")
        q.put((rank, ok_grad and ok_buf, p0.sum().item(), names, cnt, tot))
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "all-reduced gradients are not the cross-rank average"
    assert res[0][2] == pytest.approx(res[1][2]), "parameters were not broadcast from rank 0"
    assert res[0][3] == ["patch_embed3.weight", "patch_embed3.bias", "text_embed3.weight", "block3.weight", "block3.bias"]
    assert res[0][4] == 6 and res[0][5] == pytest.approx(4.5)


def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvlt_amd.dist import DataParallel
        m = _Toy()
        S = m.store
        S.materialize(torch.device("cpu"))
        dp = DataParallel(m, broadcast_buffers=False, grad_payload=torch.bfloat16)
        dp._sync_init()
        g = torch.Generator().manual_seed(5 + rank)
        mine = torch.randn(S.total, generator=g)
        S.G.copy_(mine)
        for i in (3, 2):                               # two stages announced early, the rest reduced at the end of the pass
            S.announce_stage(i)
        S._finalize()
        both = [torch.randn(S.total, generator=torch.Generator().manual_seed(5 + r)) for r in range(world)]
        want = sum(both) / world
        q.put((rank, ((S.G - want).norm() / want.norm()).item(), bool(S.G.dtype == torch.float32)))
    finally:
        dist.destroy_process_group()


def test_bf16_gradient_payload_world2():
    """the optional bf16 payload of the gradient all-reduce (half the bytes on the xGMI ring) against the fp32 reduce:
    both ranks end with the cross-rank mean to bf16 rounding (each rank's range is rounded once before, the sum once after the
    reduction: ~1.5e-3 relative L2 on Gaussian gradients), G itself stays fp32."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bf16, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, is_f32 in res:
        assert is_f32 and 1e-5 < err < 3e-3, (rank, err)
    assert res[0][1] == pytest.approx(res[1][1])


def _worker_phased(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvlt_amd.dist import DataParallel
        from mvlt_amd.optim import phased_ranges

        def adamw(P, G, M, V, lo, hi, gscale, step=1, lr=1e-2, b1=0.9, b2=0.999, eps=1e-8, wd=0.05):
            g = G[lo:hi] * gscale
            M[lo:hi].mul_(b1).add_(g, alpha=1 - b1)
            V[lo:hi].mul_(b2).addcmul_(g, g, value=1 - b2)
            P[lo:hi].mul_(1 - lr * wd).addcdiv_(M[lo:hi] / (1 - b1 ** step), (V[lo:hi] / (1 - b2 ** step)).sqrt_().add_(eps), value=-lr)

        out = {}
        for mode in ("phased", "one"):
            torch.manual_seed(3)
            m = _Toy()
            for p in m.parameters():
                p.data.normal_()
            S = m.store
            S.materialize(torch.device("cpu"))
            dp = DataParallel(m, broadcast_buffers=False)
            dp.MIN_BYTES = 256                              # the toy's stages are a few hundred bytes: two of them start a collective
            dp._sync_init()
            S.G.copy_(torch.randn(S.total, generator=torch.Generator().manual_seed(11 + rank)))
            S.scale_in_optimizer = mode == "phased"         # what engine.BF16Scaler sets around backward + FusedAdamW.step
            for i in (3, 2):
                S.announce_stage(i)                         # these go out DURING the pass
            S._finalize()                                   # ... everything else at its end
            handed = [(lo, hi, early) for _, lo, hi, _, _, early in S.grad_works]
            M, V = torch.zeros_like(S.P), torch.zeros_like(S.P)
            gscale = S.pending_grad_scale if mode == "phased" else 1.0
            ranges = list(phased_ranges(S))
            for lo, hi in ranges:
                adamw(S.P, S.G, M, V, lo, hi, gscale)
            S.scale_in_optimizer = False
            S.pending_grad_scale = 1.0
            out[mode] = (S.P.clone(), ranges, handed, not S.grad_works)
        same = bool(torch.equal(out["phased"][0], out["one"][0]))
        q.put((rank, same, out["phased"][1], out["phased"][2], out["one"][1], out["one"][2], out["phased"][3], int(m.store.total)))
    finally:
        dist.destroy_process_group()


def test_phased_optimizer_step_world2():
    """VERDICT r4 #5b: with the fused optimizer as the next reader the wrapper hands its collectives over UN-WAITED and `phased_ranges` steps the ranges
    whose collectives went out during the backward first (after their wait), the tail after its own wait -- at least two ranges, covering the buffer
    exactly once; the parameters equal, bit for bit, what wait-all + one range gives (element-wise AdamW restated in torch: the HIP kernel needs a GPU;
    tests/test_dist_gpu.py repeats this with the real kernel).  Outside the optimizer's scope the wrapper still returns final gradients."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_phased, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, ranges, handed, ranges_one, handed_one, drained, total in res:
        assert same, "phased step differs from the one-range step"
        assert len(ranges) >= 2 and sorted(ranges)[0][0] == 0 and sorted(ranges)[-1][1] == total
        srt = sorted(ranges)
        assert all(a[1] == b[0] for a, b in zip(srt, srt[1:])), ranges
        assert any(e for _, _, e in handed) and any(not e for _, _, e in handed), handed
        early_cov = sorted((lo, hi) for lo, hi, e in handed if e)
        n_early = len(ranges) - sum(1 for lo, hi, e in handed if not e)          # ranges stepped before the tail's wait
        assert n_early >= 1 and ranges[0][0] == early_cov[0][0]                   # the early collectives' ranges come first
        assert ranges_one == [(0, total)] and handed_one == [], (ranges_one, handed_one)
        assert drained


# ------------------------------------------------------------------------------------------------------------------------------------
# World 4 / 8 with the REAL parameter layouts (VERDICT r3 #6b): what travels when, in which order, and what is left for the end of the pass
def _worker_real(rank, world, port, q, variant, payload):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvlt_amd import pvlt
        from mvlt_amd.dist import DataParallel
        lt = dict(mlm=1, itm=1, t2i=1, cls=0)
        m = getattr(pvlt, variant)(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=lt, pretrained_pth=None,
                                   compute_dtype=torch.float32)
        S = m.store
        S.materialize(torch.device("cpu"))
        dp = DataParallel(m, broadcast_buffers=True, grad_payload=payload)
        seen, sent = [], []
        inner = S.on_range_ready
        S.on_range_ready = lambda st, lo, hi: (seen.append((lo, hi)), inner(st, lo, hi))
        inner_reduce = dp._reduce
        dp._reduce = lambda st, lo, hi: (sent.append((lo, hi, len(seen))), inner_reduce(st, lo, hi))
        dp._sync_init()
        if rank != 0:
            for b in m.buffers():
                if b.is_floating_point():
                    b.fill_(float(rank))
        dp._sync_buffers()                                   # rank 0's BatchNorm statistics everywhere, through ONE broadcast
        slab = dp._slab
        bufs_follow_rank0 = all(bool((b == (0.0 if "mean" in n else 1.0)).all()) for n, b in m.named_buffers() if b.is_floating_point())
        views_alias = all(b.data_ptr() >= slab.data_ptr() and b.data_ptr() < slab.data_ptr() + slab.numel() * 4
                          for b in m.buffers() if b.is_floating_point())
        # one backward pass the way the schedule announces it: MIM decoder, MLM head, ITM head (heads run first), then stages 4 -> 1
        S.G.copy_(torch.arange(S.total, dtype=torch.float32).remainder_(1024.0) * (rank + 1))
        S._ranges_done = []
        S.announce_prefix("t2i_head.")
        S.announce_prefix("mlm_head_embed.", "mlm_head.")
        S.announce_prefix("itm_head_embed.", "itm_head.")
        for i in (3, 2, 1, 0):
            S.announce_stage(i)
        S.announce_prefix("pos_embed", "text_pos_embed")     # TrunkStep.backward, after stage 1
        S._finalize()
        expect = torch.arange(S.total, dtype=torch.float32).remainder_(1024.0) * (sum(r + 1 for r in range(world)) / world)
        err = ((S.G - expect).norm() / expect.norm()).item()
        # what was NOT announced = reduced at the end of the pass
        covered = sorted(seen)
        tail, cur = [], 0
        for lo, hi in covered + [(S.total, S.total)]:
            if lo > cur:
                tail.append((cur, lo))
            cur = max(cur, hi)
        tail_names = sorted({n for n, (o, k, _) in S.offsets.items() for lo, hi in tail if lo <= o < hi})
        q.put((rank, err, bufs_follow_rank0, views_alias, seen, tail, tail_names, sent,
               {n: S.offsets[n][0] for n in ("text_embeddings.word_embeddings.weight", "pos_embed1", "text_pos_embed4")}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant,world,payload", [("pvlt_tiny", 8, torch.float32), ("pvlt_medium", 4, torch.float32), ("pvlt_tiny", 4, torch.bfloat16)])
def test_real_layout_ranges_world_4_and_8(variant, world, payload):
    """The gradient exchange of the pre-train configurations at world 4 / 8 (gloo, CPU; unmeasured on xGMI): every rank ends with the
    cross-rank mean; the ranges travel in gradient-ready order (MIM decoder, MLM head, ITM head, stage 4 .. 1, position embeddings); announcements
    are held back until 16 MB are pending and neighbours merge, so that at most six collectives of >= 11 MB go out per step (the last,
    position embeddings + stage 1, excepted); what is left for the end of the pass is the BERT embedding block alone (the tied word table, which
    the embedding-lookup gradient -- the last kernel of the backward -- still writes), 95 MB in fp32."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_real, args=(r, world, port, q, variant, payload)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    tol = 1e-6 if payload is torch.float32 else 5e-3
    for rank, err, bufs_ok, alias_ok, seen, tail, tail_names, sent, offs in res:
        assert err < tol, (rank, err)
        assert bufs_ok and alias_ok, (rank, bufs_ok, alias_ok)
    rank, err, _, _, seen, tail, tail_names, sent, offs = res[0]
    assert all(r[4] == seen and r[7] == sent for r in res), "ranks announce / send different ranges"
    assert len(seen) == 8 and all(hi > lo for lo, hi in seen)
    # announcements: heads first, then the stages back to front, then the learned position embeddings (root parameters: the first 2 MB of
    # the flat buffer, in front of stage 1)
    los = [lo for lo, _ in seen[3:]]
    assert los == sorted(los, reverse=True)
    assert seen[7][0] == 0 and seen[7][0] <= offs["pos_embed1"] < seen[7][1] and seen[7][0] <= offs["text_pos_embed4"] < seen[7][1]
    assert seen[7][1] == seen[6][0], "position embeddings and stage 1 are neighbours: one collective"
    srt = sorted(seen)
    assert all(a[1] <= b[0] for a, b in zip(srt, srt[1:]))
    # what was not announced = the BERT embedding block alone (tied word table + position / type tables + LayerNorm), 91 MB in fp32
    assert tail_names and all(n.startswith("text_embeddings.") for n in tail_names), tail_names
    assert len(tail) == 1 and tail[0][0] <= offs["text_embeddings.word_embeddings.weight"] < tail[0][1]
    assert 88 < (tail[0][1] - tail[0][0]) * 4 / 2 ** 20 < 100
    # what went on the wire: the three head ranges wait for stage 4 and travel as ONE range next to it; nothing under 11 MB is sent before the
    # end of the pass; at most six collectives per step; together they cover the whole buffer exactly once
    mb = [(hi - lo) * 4 / 2 ** 20 for lo, hi, _ in sent]
    early = [(m, n) for m, (_, _, n) in zip(mb, sent) if n < 8]
    assert len(sent) <= 6, mb
    assert early and all(m >= 11 for m, _ in early), mb
    first = [m for m, (_, _, n) in zip(mb, sent) if n == 4]                # the flush that stage 4's arrival triggers: stage 4 + the merged heads
    heads_mb = sum((hi - lo) * 4 / 2 ** 20 for lo, hi in seen[:3])
    assert len(first) == 2 and any(abs(m - heads_mb) < 1e-6 for m in first), (first, heads_mb, mb)
    cov = sorted((lo, hi) for lo, hi, _ in sent)
    assert cov[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))


def test_bench_spawn_ranks_plumbing(monkeypatch):
    """`python bench.py --gpus 8` on a node with 8 GPUs (VERDICT r3 #6e): the launcher command and environment it builds -- one rank per GPU
    through torch.distributed.run on 127.0.0.1, the ranks as CHILD processes, dmabuf IPC for RCCL, the caller's arguments passed on
    without --spawn, no forced collectives above world size 1 -- and the refusal on a smaller node."""
    import subprocess
    import sys
    import bench
    calls = []

    class _R:
        returncode = 0

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, **k: (calls.append((cmd, env)), _R())[1])
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2", "--spawn"])
    monkeypatch.delenv("MVLT_DP_FORCE_COLLECTIVES", raising=False)
    assert bench.spawn_ranks(8) == 0
    (cmd, env), = calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "7", "--warmup", "2"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and int(env["OMP_NUM_THREADS"]) >= 1
    assert "MVLT_DP_FORCE_COLLECTIVES" not in env
    # world size 1 through the launcher = the RCCL code path of N > 1
    calls.clear()
    assert bench.spawn_ranks(1) == 0
    assert calls[0][1]["MVLT_DP_FORCE_COLLECTIVES"] == "1" and "--nproc-per-node=1" in calls[0][0]
    # a node with fewer GPUs than asked for: refused before anything is started
    calls.clear()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    assert bench.spawn_ranks(8) == 2 and not calls


@pytest.mark.parametrize("n", [2, 4])
def test_bench_spawn_ranks_plumbing_for_2_and_4(monkeypatch, n):
    """the driver's N = 2 / 4 points of the scaling curve through `python bench.py --gpus N` (VERDICT r4 #5c): N ranks on 127.0.0.1, no forced
    collectives, the caller's arguments passed on"""
    import subprocess
    import sys
    import bench
    calls = []

    class _R:
        returncode = 0

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, **k: (calls.append((cmd, env)), _R())[1])
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(n), "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("MVLT_DP_FORCE_COLLECTIVES", raising=False)
    assert bench.spawn_ranks(n) == 0
    (cmd, env), = calls
    assert f"--nproc-per-node={n}" in cmd and "--nnodes=1" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index(os.path.abspath(bench.__file__)) + 1:] == ["--gpus", str(n), "--steps", "20", "--warmup", "5"]
    assert "MVLT_DP_FORCE_COLLECTIVES" not in env and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.parametrize("world,local", [(2, 1), (4, 3), (8, 5)])
def test_bench_rank_binds_its_own_device_before_anything_else(monkeypatch, world, local):
    """Under a launcher every rank must bind LOCAL_RANK's GPU BEFORE its first allocation, its process group and its first kernel -- a rank that
    allocates on the default device first leaves 8 contexts on GPU 0 (and RCCL then binds the wrong device).  bench.main() is run with the GPU
    entry points replaced by recorders: the first thing it does to the GPU runtime is set_device(LOCAL_RANK)."""
    import sys
    import bench

    class _Stop(Exception):
        pass

    events = []

    def bind(d):
        events.append(("set_device", d))
        raise _Stop()

    monkeypatch.setenv("WORLD_SIZE", str(world))
    monkeypatch.setenv("RANK", str(local))
    monkeypatch.setenv("LOCAL_RANK", str(local))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1"])
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", bind)
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **k: events.append(("init_process_group",)))
    for fn in ("empty", "zeros", "randn", "rand", "full", "tensor"):
        orig = getattr(torch, fn)
        monkeypatch.setattr(torch, fn, (lambda o, f: lambda *a, **k: (events.append((f, k.get("device"))), o(*a, **k))[1])(orig, fn))
    with pytest.raises(_Stop):
        bench.main()
    assert events == [("set_device", local)], events
