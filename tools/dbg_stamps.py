import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops, _lib
dev = torch.device('cuda:0'); bf = torch.bfloat16
M, N, K = 98304, 1280, 320
A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
out = torch.empty(M, N, device=dev, dtype=bf)
for _ in range(3): ops.gemm_nt(A, W, out, M, N, K, K, K, N)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * (64 * 16))()
_lib.lib.mvlt_debug_read(buf, 64 * 16)
for s in range(8):
    v = [buf[s * 16 + k] for k in range(14)]
    base = v[0]
    print('wg', s * 997, ' '.join('%6d' % (x - base) if x else '     -' for x in v))
