"""Metric bookkeeping for the train / eval loops: the counterpart of the reference's SmoothedValue / MetricLogger
(libs/utils.py:18-161) with the same public surface (update, meters[name].global_avg, log_every,
synchronize_between_processes), written around a running (count, total) pair plus a bounded window."""
import datetime
import time
from collections import defaultdict, deque

import torch

from .dist import allreduce_meter


class SmoothedValue:
    def __init__(self, window_size=20, fmt=None):
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"
        self.window = deque(maxlen=window_size)
        self.count, self.total = 0, 0.0

    def update(self, value, n=1):
        self.window.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self, device="cuda"):
        self.count, self.total = allreduce_meter(self.count, self.total, device)

    @property
    def median(self):
        return torch.tensor(list(self.window)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.window), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.window)

    @property
    def value(self):
        return self.window[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)      # `logger.meters['x'].update(v, n=...)` creates the meter, like the reference's
        self.delimiter = delimiter

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def update(self, **kw):
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int)), (k, type(v))
            self.meters[k].update(v)

    def __getattr__(self, name):
        meters = self.__dict__.get("meters", {})
        if name in meters:
            return meters[name]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __str__(self):
        return self.delimiter.join(f"{k}: {m}" for k, m in self.meters.items())

    def synchronize_between_processes(self, device="cuda"):
        """(count, total) of every meter summed over the ranks -- all meters in ONE all-reduce (the reference issues a barrier and
        an all-reduce per meter, libs/utils.py:38-47; same result)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        ms = list(self.meters.values())
        t = torch.tensor([v for m in ms for v in (m.count, m.total)], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        t = t.tolist()
        for i, m in enumerate(ms):
            m.count, m.total = int(t[2 * i]), t[2 * i + 1]

    def log_every(self, iterable, print_freq, header=""):
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable) if hasattr(iterable, "__len__") else None
        start = end = time.time()
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or (n is not None and i == n - 1):
                parts = [header, f"[{i}/{n}]" if n is not None else f"[{i}]"]
                if n is not None:
                    parts.append("eta: " + str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i)))))
                parts += [str(self), f"time: {iter_time}", f"data: {data_time}"]
                if torch.cuda.is_available():
                    parts.append(f"max mem: {torch.cuda.max_memory_allocated() / 2**20:.0f}")
                print(self.delimiter.join(parts))
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(1, n or 1):.4f} s / it)")
