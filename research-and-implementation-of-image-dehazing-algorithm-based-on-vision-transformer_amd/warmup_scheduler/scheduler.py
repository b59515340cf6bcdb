"""Per-epoch linear warm-up followed by a wrapped scheduler (warmup_scheduler/scheduler.py:5-63;
used by My_train.py as warmup(3 epochs, multiplier 1) -> CosineAnnealingLR).

Learning-rate factor during warm-up epoch e of E:  multiplier == 1:  e / E   (so the very first epoch runs at lr = 0, a
quirk of the reference that My_train's logs show);  multiplier m > 1:  1 + (m - 1) e / E.  After epoch E the wrapped
scheduler takes over with its base rates multiplied by m."""
from torch.optim.lr_scheduler import _LRScheduler


class GradualWarmupScheduler(_LRScheduler):
    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        if multiplier < 1.:
            raise ValueError('multiplier should be greater thant or equal to 1.')
        self.multiplier = multiplier
        self.total_epoch = total_epoch
        self.after_scheduler = after_scheduler
        self.finished = False                  # True once the wrapped scheduler has been handed its base rates
        super().__init__(optimizer)

    def _warm_factor(self):
        frac = float(self.last_epoch) / self.total_epoch
        return frac if self.multiplier == 1.0 else 1. + (self.multiplier - 1.) * frac

    def get_lr(self):
        warming = self.last_epoch <= self.total_epoch
        if warming:
            f = self._warm_factor()
            return [base * f for base in self.base_lrs]
        scaled = [base * self.multiplier for base in self.base_lrs]
        if self.after_scheduler is None:
            return scaled
        if not self.finished:
            self.after_scheduler.base_lrs = scaled
            self.finished = True
        return self.after_scheduler.get_lr()

    def step(self, epoch=None, metrics=None):
        handed_over = self.finished and self.after_scheduler is not None
        if not handed_over:
            return super().step(epoch)
        self.after_scheduler.step(None if epoch is None else epoch - self.total_epoch)
