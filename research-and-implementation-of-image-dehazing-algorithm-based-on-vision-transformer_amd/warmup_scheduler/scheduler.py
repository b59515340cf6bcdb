"""Per-epoch linear warm-up followed by a wrapped scheduler (warmup_scheduler/scheduler.py:5-63;
used by My_train.py as warmup(3 epochs, multiplier 1) -> CosineAnnealingLR)."""
from torch.optim.lr_scheduler import _LRScheduler


class GradualWarmupScheduler(_LRScheduler):
    def __init__(self, optimizer, multiplier, total_epoch, after_scheduler=None):
        if multiplier < 1.:
            raise ValueError('multiplier should be greater thant or equal to 1.')
        self.multiplier, self.total_epoch, self.after_scheduler = multiplier, total_epoch, after_scheduler
        self.finished = False
        super().__init__(optimizer)

    def get_lr(self):
        if self.last_epoch > self.total_epoch:
            if self.after_scheduler:
                if not self.finished:
                    self.after_scheduler.base_lrs = [b * self.multiplier for b in self.base_lrs]
                    self.finished = True
                return self.after_scheduler.get_lr()
            return [b * self.multiplier for b in self.base_lrs]
        if self.multiplier == 1.0:
            return [b * (float(self.last_epoch) / self.total_epoch) for b in self.base_lrs]
        return [b * ((self.multiplier - 1.) * self.last_epoch / self.total_epoch + 1.) for b in self.base_lrs]

    def step(self, epoch=None, metrics=None):
        if self.finished and self.after_scheduler:
            if epoch is None:
                self.after_scheduler.step(None)
            else:
                self.after_scheduler.step(epoch - self.total_epoch)
        else:
            return super().step(epoch)
