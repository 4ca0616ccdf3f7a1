"""SR-attention forward / backward microbench on the four stage shapes of pvlt_tiny at batch 256 (256 px, T = 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = int(os.environ.get("B", "256"))          # B=64 KEYS=272: pvlt_medium at 384 px (stage shapes 9344 / 2432 / 704 / 272 queries)
SHAPES = ((1, 4224, 192), (2, 1152, 192), (5, 384, 192), (8, 192, 192)) if os.environ.get("KEYS", "192") == "192" else ((1, 9344, 272), (2, 2432, 272), (5, 704, 272), (8, 272, 272))
for H, N, M in SHAPES:
    C = 64 * H
    q, kv = torch.randn(B, N, C, device=dev).to(bf), torch.randn(B, M, 2 * C, device=dev).to(bf)
    o, lse = torch.empty_like(q), torch.empty(B, H, N, device=dev)
    do, dq = torch.randn(B, N, C, device=dev).to(bf), torch.empty_like(q)
    dkv = torch.empty(B, M, 2 * C, device=dev, dtype=bf) if B * H >= 512 else torch.zeros(B, M, 2 * C, device=dev)
    tf = timeit(lambda: ops.sr_attention_fwd(q, kv, o, lse, B, H, N, M, C, 2 * C, C, 0, C, 0.125))
    tb = timeit(lambda: ops.sr_attention_bwd(q, kv, o, do, lse, dq, dkv, B, H, N, M, C, 2 * C, C, 2 * C, 0, C, 0.125))
    fl = 4.0 * B * H * N * M * 64
    print(f'H={H} N={N} M={M}: fwd {tf*1e3:.1f} us {fl/tf/1e9:.0f} TF/s | bwd {tb*1e3:.1f} us {2.5*fl/tb/1e9:.0f} TF/s')
