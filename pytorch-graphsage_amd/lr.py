"""
lr.py -- learning-rate schedules as functions of training progress (reference lr.py:11-42).

Values are kept identical to the reference, including its quirks:
`epochs` never reaches the schedule from GSSupervised so `linear` decays over progress in [0,1),
`cyclical` returns 0.05 for all progress < 1, and `step` takes no lr_init (selecting it through
GSSupervised raises TypeError, as in the reference).
"""
import math


class LRSchedule(object):
    @staticmethod
    def set_lr(optimizer, lr):
        for group in optimizer.param_groups:
            if hasattr(group['lr'], 'fill_'):      # tensor lr of a graph-captured optimizer
                group['lr'].fill_(float(lr))
            else:
                group['lr'] = lr

    @staticmethod
    def constant(x, lr_init=0.1, epochs=1):
        return lr_init

    @staticmethod
    def step(x, breaks=(150, 250)):
        for bound, value in zip(breaks, (0.1, 0.01)):
            if x < bound:
                return value
        return 0.001

    @staticmethod
    def linear(x, lr_init=0.1, epochs=1):
        return lr_init * float(epochs - x) / epochs

    @staticmethod
    def cyclical(x, lr_init=0.1, epochs=1):
        if x < 1:
            return 0.05
        frac = x - math.floor(x)
        return lr_init * (1 - frac) * (epochs - math.floor(x)) / epochs
