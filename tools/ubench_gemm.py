import sys, torch
sys.path.insert(0, '.')
from mvlt_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = 256
M = B * 4224
bf = torch.bfloat16
x64 = torch.randn(M, 64, device=dev).to(bf)
w512 = (torch.randn(512, 64, device=dev) * 0.1).to(bf)
b512 = torch.randn(512, device=dev)
o512 = torch.empty(M, 512, device=dev, dtype=bf); h512 = torch.empty_like(o512)
big = torch.empty(M * 512, device=dev, dtype=bf)
print('fill 1.1GB      %.3f ms -> %.0f GB/s' % ((t := timeit(lambda: big.fill_(1.0))), big.numel() * 2 / t / 1e6))
print('copy 1.1GB r+w  %.3f ms -> %.0f GB/s (r+w)' % ((t := timeit(lambda: o512.view(-1).copy_(big))), 2 * big.numel() * 2 / t / 1e6))
for name, kw in [('fc1 gelu+H (2.2GB w)', dict(bias=b512, act=1, H=h512)), ('fc1 gelu noH (1.1GB w)', dict(bias=b512, act=1)),
                 ('fc1 linear (1.1GB w)', dict(bias=b512)), ]:
    t = timeit(lambda: ops.gemm_nt(x64, w512, o512, M, 512, 64, 64, 64, 512, **kw))
    wbytes = M * 512 * 2 * (2 if 'H' in kw else 1) + M * 64 * 2
    print('%-26s %.3f ms -> %.0f GB/s' % (name, t, wbytes / t / 1e6))
# fc2-like: K=512 N=64 with residual fp32
g = o512; w2 = (torch.randn(64, 512, device=dev) * 0.05).to(bf); b64 = torch.randn(64, device=dev)
r32 = torch.randn(M, 64, device=dev); out32 = torch.empty_like(r32)
t = timeit(lambda: ops.gemm_nt(g, w2, out32, M, 64, 512, 512, 512, 64, bias=b64, R=r32))
print('fc2 K=512 N=64 +R fp32     %.3f ms -> %.0f GB/s' % (t, (M * 512 * 2 + M * 64 * 8) / t / 1e6))
# q-like: K=64,N=64
o64 = torch.empty(M, 64, device=dev, dtype=bf); w64 = (torch.randn(64, 64, device=dev) * 0.1).to(bf)
t = timeit(lambda: ops.gemm_nt(x64, w64, o64, M, 64, 64, 64, 64, 64, bias=b64))
print('q K=64 N=64               %.3f ms -> %.0f GB/s' % (t, (M * 64 * 4) / t / 1e6))
# wgrad fc1: dW[512,64] = dh^T x ; wgrad fc2: dW[64,512] = dy^T g
dW = torch.zeros(512, 64, device=dev)
t = timeit(lambda: ops.gemm_tn(o512, x64, dW, M, 512, 64, 512, 64, 64))
print('wgrad fc1 [512,64]        %.3f ms -> %.0f GB/s' % (t, (M * 576 * 2) / t / 1e6))
dW2 = torch.zeros(64, 512, device=dev)
t = timeit(lambda: ops.gemm_tn(x64, o512, dW2, M, 64, 512, 64, 512, 512))
print('wgrad fc2 [64,512] (swap) %.3f ms -> %.0f GB/s' % (t, (M * 576 * 2) / t / 1e6))
# LN fp32->bf16
from mvlt_amd._lib import rowmap
xf = torch.randn(M, 64, device=dev); y = torch.empty(M, 64, device=dev, dtype=bf); gm = torch.ones(64, device=dev); bt = torch.zeros(64, device=dev)
t = timeit(lambda: ops.layernorm_fwd(xf, y, gm, bt, M, 64, 64, 64, 1e-6))
print('LN fwd C=64 f32->bf16      %.3f ms -> %.0f GB/s' % (t, (M * 64 * 6) / t / 1e6))
