"""BERT embedding forward / backward kernels at the step's shape (256 x 128 tokens, hidden 768, vocabulary 30522)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, T, H, V = 256, 128, 768, 30522
torch.manual_seed(0)
ids = torch.randint(1, V, (B, T), device=dev)
word, pos, typ = torch.randn(V, H, device=dev), torch.randn(512, H, device=dev), torch.randn(2, H, device=dev)
gamma = torch.ones(H, device=dev)
dy = torch.randn(B * T, H, device=dev).to(bf)
mean, rstd = torch.zeros(B * T, device=dev), torch.ones(B * T, device=dev)
dword, dpos, dtyp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
step_ids = bench.synth_batch(B, 32, T, dev, 1)["input_ids"]          # the bench's own captions: ~11 % [MASK] (103), 101 / 102 once per row, PAD (0) behind
for name, idv in (("random ids", ids), ("32 distinct ids", ids % 32 + 1), ("the step's ids", step_ids)):
    t = timeit(lambda: ops.bert_embed_bwd(dy, idv, word, pos, typ[0], gamma, None, 0.1, mean, rstd, dword, dpos, dtyp[0], dg, db, B * T, T))
    print(f"bert_embed_bwd {name}: {t:.1f} us")
