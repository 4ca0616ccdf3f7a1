"""The fine-tune step's first ~20 iterations are ~15 % slower than its steady state: per-iteration wall time (a synchronize
after every iteration), host enqueue time, allocator state and GPU clock for the first 45 iterations of the CLS-head fine-tune step and of the pre-train
step, in a fresh process each.    python tools/ft_ramp.py [finetune|pretrain] [iterations]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            if "sclk" in line:
                return line.split("(")[-1].split(")")[0]
    except Exception as e:           # noqa: BLE001
        return f"n/a ({type(e).__name__})"
    return "n/a"


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "finetune"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    import bench
    from mvlt_amd import pvlt
    from mvlt_amd.engine import BF16Scaler, train_step
    from mvlt_amd.optim import FusedAdamW
    dev = torch.device("cuda", 0)
    lt = dict(mlm=0, itm=0, t2i=0, cls=1) if task == "finetune" else dict(mlm=1, itm=1, t2i=1, cls=0)
    torch.manual_seed(4321)
    model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=lt, pretrained_pth=None, drop_path_rate=0.1,
                           drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
    model.train()
    batch = bench.synth_batch(256, 256, 128, dev, 99)
    batch["mlm_count"] = int((batch["mlm_labels"] != -1).sum())
    opt = FusedAdamW(model, lr=1e-4, weight_decay=0.01)
    scaler = BF16Scaler()
    print(f"{task}: iteration, wall ms (synchronised), host enqueue ms, reserved MB, allocated MB, sclk at iterations 0 / 10 / 20 / 30 / 40")
    for i in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total, _ = train_step(model, batch, i, lt.get("t2i", 0) == 1)
        opt.zero_grad()
        scaler(total, opt, clip_grad=None, parameters=None)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        clk = sclk() if i % 10 == 0 else ""
        print(f"  {i:3d}  {1e3 * (t2 - t0):7.2f}  {1e3 * (t1 - t0):7.2f}  {torch.cuda.memory_reserved() / 2**20:9.0f}  {torch.cuda.memory_allocated() / 2**20:9.0f}  {clk}")


if __name__ == "__main__":
    main()
