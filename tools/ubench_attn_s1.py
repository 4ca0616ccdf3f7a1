import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B, H, N, M = 256, 1, 4224, 192
C = 64
q, kv = torch.randn(B, N, C, device=dev).to(bf), torch.randn(B, M, 2 * C, device=dev).to(bf)
o, lse = torch.empty_like(q), torch.empty(B, H, N, device=dev)
do, dq = torch.randn(B, N, C, device=dev).to(bf), torch.empty_like(q)
dkv = torch.zeros(B, M, 2 * C, device=dev)
ops.sr_attention_fwd(q, kv, o, lse, B, H, N, M, C, 2 * C, C, 0, C, 0.125)
tb = timeit(lambda: ops.sr_attention_bwd(q, kv, o, do, lse, dq, dkv, B, H, N, M, C, 2 * C, C, 2 * C, 0, C, 0.125))
print('NQ', os.environ.get('MVLT_ATTN_BWD_NQ'), 'stage-1 bwd %.1f us' % (tb * 1e3))
