"""``LRSchedule`` of ``src/dagr/utils/learning_rate_scheduler.py:8-47``: the multiplier handed to
``torch.optim.lr_scheduler.LambdaLR`` (train_ncaltech101.py:136-140) -- quadratic warm-up over ``warmup_epochs``, then a
cosine from 1 down to ``min_lr_ratio`` over the rest of the run, halved (``reduction_at_step``) from every iteration listed
in ``steps_at_iteration`` on."""
import math


class LRSchedule:
    def __init__(self, warmup_epochs, num_iters_per_epoch, tot_num_epochs, min_lr_ratio=0.05, warmup_lr_start=0,
                 steps_at_iteration=(50000,), reduction_at_step=0.5):
        self.warmup_iters = num_iters_per_epoch * warmup_epochs
        self.total_iters = tot_num_epochs * num_iters_per_epoch
        self.floor, self.start = float(min_lr_ratio), float(warmup_lr_start)
        self.steps, self.reduction = tuple(steps_at_iteration), float(reduction_at_step)

    def __call__(self, iters):
        if iters < self.warmup_iters:
            factor = (1 - self.start) * (iters / float(self.warmup_iters)) ** 2 + self.start
        else:
            phase = math.pi * (iters - self.warmup_iters) / (self.total_iters - self.warmup_iters)
            factor = self.floor + 0.5 * (1 - self.floor) * (1.0 + math.cos(phase))
        for step in self.steps:
            if iters >= step:
                factor *= self.reduction
        return factor
