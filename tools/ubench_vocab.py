"""The MLM decoder's logits GEMM (selected rows x 30522 words x 768, fp32 logits) alone, e.g. against the ring depth: MVLT_NT_NS=3 python tools/ubench_vocab.py"""
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
for M in (1490, 4915):
    N, K = 30522, 768
    x = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.05).to(bf); b = torch.randn(N, device=dev)
    ld = (N + 7) // 8 * 8
    out = torch.empty(M, ld, device=dev)
    t = timeit(lambda: ops.gemm_nt(x, w, out, M, N, K, K, K, ld, bias=b))
    ref = x.float() @ w.float().t() + b
    err = ((out[:, :N] - ref).abs().max() / ref.abs().max()).item()
    print(f'logits {M} x {N} x {K}: {t:.1f} us  {2.0 * M * N * K / t / 1e6:.0f} TF/s   max-norm error {err:.2e}')
    outb = torch.empty(M, ld, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm_nt(x, w, outb, M, N, K, K, K, ld, bias=b))
    print(f'   bf16 logits: {t:.1f} us  {2.0 * M * N * K / t / 1e6:.0f} TF/s')
