"""Learning-rate schedule of the reference's training runs: `timm.scheduler.create_scheduler(args, optimizer)`
with `--sched cosine` (reference main_vl.py:69,310), stepped ONCE PER EPOCH with `lr_scheduler.step(epoch)` (main_vl.py:439).

timm==0.3.2 is not vendored under /root/reference (requirements.txt:1); this is a restatement of its published
`CosineLRScheduler` / `create_scheduler` behaviour for the arguments the reference passes -- parity unpinned by the reference, pinned
here by closed-form values (tests/test_host_cpu.py):

    epoch <  warmup_epochs : lr = warmup_lr + epoch * (base_lr - warmup_lr) / warmup_epochs
    epoch <  epochs        : lr = min_lr + 0.5 * (base_lr - min_lr) * (1 + cos(pi * epoch / epochs))      (one cycle, no prefix shift)
    epoch >= epochs        : lr = min_lr                                                                  (the cool-down epochs)

`num_epochs` = epochs + cooldown_epochs is what create_scheduler returns as the run length.  Works on any torch optimizer, including
FusedAdamW (both parameter groups get the same lr, which its kernel requires).
"""
import math


class CosineLRScheduler:
    def __init__(self, optimizer, epochs, min_lr=1e-5, warmup_lr=1e-6, warmup_epochs=5, cooldown_epochs=10):
        self.optimizer = optimizer
        self.epochs, self.min_lr, self.warmup_lr, self.warmup_epochs = int(epochs), float(min_lr), float(warmup_lr), int(warmup_epochs)
        self.num_epochs = self.epochs + int(cooldown_epochs)
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        if self.warmup_epochs:                       # timm sets the warm-up start value at construction
            self._set([self.warmup_lr] * len(self.base_lrs))

    def lr_at(self, epoch, base_lr):
        if epoch < self.warmup_epochs:
            return self.warmup_lr + epoch * (base_lr - self.warmup_lr) / self.warmup_epochs
        if epoch < self.epochs:
            return self.min_lr + 0.5 * (base_lr - self.min_lr) * (1.0 + math.cos(math.pi * epoch / self.epochs))
        return self.min_lr

    def _set(self, lrs):
        for g, lr in zip(self.optimizer.param_groups, lrs):
            g["lr"] = lr

    def step(self, epoch, metric=None):
        self._set([self.lr_at(epoch, b) for b in self.base_lrs])

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


def create_scheduler(args, optimizer):
    """(scheduler, num_epochs) like timm.scheduler.create_scheduler for `--sched cosine` with the reference's argument names
    (main_vl.py:69-87: epochs, min_lr, warmup_lr, warmup_epochs, cooldown_epochs)."""
    assert getattr(args, "sched", "cosine") == "cosine", "the reference configurations use the cosine schedule"
    s = CosineLRScheduler(optimizer, args.epochs, getattr(args, "min_lr", 1e-5), getattr(args, "warmup_lr", 1e-6),
                          getattr(args, "warmup_epochs", 5), getattr(args, "cooldown_epochs", 10))
    return s, s.num_epochs
