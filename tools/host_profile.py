"""Where the HOST time of a training step goes (cProfile over 8 steady-state steps, autograd's backward forced onto the calling thread so that the
profiler sees it): python tools/host_profile.py > profiles/rNN_host_profile.txt"""
import cProfile, io, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvlt_amd import pvlt
from mvlt_amd.engine import BF16Scaler, train_step
from mvlt_amd.optim import FusedAdamW
dev = torch.device('cuda', 0)
model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=dict(mlm=1, itm=1, t2i=1, cls=0),
                       pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
model.train()
batch = bench.synth_batch(256, 256, 128, dev, 1)
batch["mlm_count"] = int((batch["mlm_labels"] != -1).sum())
opt = FusedAdamW(model, lr=1e-4, weight_decay=0.01); scaler = BF16Scaler()
def step(i):
    total, _ = train_step(model, batch, i, True)
    opt.zero_grad(); scaler(total, opt, clip_grad=None, parameters=None)
for i in range(4): step(i)
torch.cuda.synchronize()
N = 8
for single in (False, True):
    torch.autograd.set_multithreading_enabled(not single)
    for i in range(2): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"autograd on {'the calling thread' if single else 'its own thread'}: host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step, with the GPU {1e3 * (time.perf_counter() - t0) / N:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(N): step(i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("tottime")
st.print_stats(45)
print(f"(per step: divide by {N}; the profiler itself roughly doubles the time of small Python functions)")
print(s.getvalue())
