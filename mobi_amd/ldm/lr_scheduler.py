"""Learning-rate multiplier schedules of the reference (`ldm/lr_scheduler.py`), the objects `scheduler_config` names in every
MObI config (`configs/mobi_nusc_512.yaml:54-61`: `LambdaLinearScheduler`, 200 warm-up steps): a callable step -> factor that
`configure_optimizers` (ddpm.py:1651-1668) hands to a LambdaLR.  Same class names, constructor arguments, attribute names
(`last_lr` / `last_f`, `cum_cycles`) and values -- host arithmetic in float64 in the reference's operation order, pinned to
the reference's own objects by tests/golden/lr_schedules.npz (bit for bit).

Behaviours kept on purpose: a step that sits exactly on a cycle boundary belongs to the EARLIER cycle (`n <= boundary`), and a
step beyond the last cycle is an error (the reference indexes with None there)."""

import numpy as np


def _say(interval, n, last, cycle=None):
    if interval > 0 and n % interval == 0:
        tail = "" if cycle is None else f", current cycle {cycle}"
        print(f"current step: {n}, recent lr-multiplier: {last}{tail}")


class LambdaWarmUpCosineScheduler:
    """lr_scheduler.py:4-34: linear warm-up from lr_start to lr_max over warm_up_steps, then half a cosine down to lr_min at
    max_decay_steps (use with a base lr of 1.0)."""

    def __init__(self, warm_up_steps, lr_min, lr_max, lr_start, max_decay_steps, verbosity_interval=0):
        self.lr_warm_up_steps, self.lr_start, self.lr_min, self.lr_max = warm_up_steps, lr_start, lr_min, lr_max
        self.lr_max_decay_steps = max_decay_steps
        self.last_lr = 0.
        self.verbosity_interval = verbosity_interval

    def schedule(self, n, **kwargs):
        _say(self.verbosity_interval, n, self.last_lr)
        if n < self.lr_warm_up_steps:
            lr = (self.lr_max - self.lr_start) / self.lr_warm_up_steps * n + self.lr_start
        else:
            t = min((n - self.lr_warm_up_steps) / (self.lr_max_decay_steps - self.lr_warm_up_steps), 1.0)
            lr = self.lr_min + 0.5 * (self.lr_max - self.lr_min) * (1 + np.cos(t * np.pi))
        self.last_lr = lr
        return lr

    __call__ = schedule


class LambdaWarmUpCosineScheduler2:
    """lr_scheduler.py:37-79: the same per cycle, every argument a list with one entry per cycle."""

    def __init__(self, warm_up_steps, f_min, f_max, f_start, cycle_lengths, verbosity_interval=0):
        assert len(warm_up_steps) == len(f_min) == len(f_max) == len(f_start) == len(cycle_lengths)
        self.lr_warm_up_steps, self.f_start, self.f_min, self.f_max = warm_up_steps, f_start, f_min, f_max
        self.cycle_lengths = cycle_lengths
        self.cum_cycles = np.cumsum([0] + list(self.cycle_lengths))
        self.last_f = 0.
        self.verbosity_interval = verbosity_interval

    def find_in_interval(self, n):
        for cycle, end in enumerate(self.cum_cycles[1:]):
            if n <= end:
                return cycle
        return None

    def _locate(self, n):
        cycle = self.find_in_interval(n)
        if cycle is None:
            raise IndexError(f"step {n} lies beyond the last cycle ({int(self.cum_cycles[-1])} steps)")
        n = n - self.cum_cycles[cycle]
        _say(self.verbosity_interval, n, self.last_f, cycle)
        return cycle, n

    def _warm_up(self, cycle, n):
        return (self.f_max[cycle] - self.f_start[cycle]) / self.lr_warm_up_steps[cycle] * n + self.f_start[cycle]

    def _after(self, cycle, n):
        t = min((n - self.lr_warm_up_steps[cycle]) / (self.cycle_lengths[cycle] - self.lr_warm_up_steps[cycle]), 1.0)
        return self.f_min[cycle] + 0.5 * (self.f_max[cycle] - self.f_min[cycle]) * (1 + np.cos(t * np.pi))

    def schedule(self, n, **kwargs):
        cycle, n = self._locate(n)
        f = self._warm_up(cycle, n) if n < self.lr_warm_up_steps[cycle] else self._after(cycle, n)
        self.last_f = f
        return f

    def __call__(self, n, **kwargs):
        return self.schedule(n, **kwargs)


class LambdaLinearScheduler(LambdaWarmUpCosineScheduler2):
    """lr_scheduler.py:82-98: linear warm-up, then a LINEAR ramp from f_max at the start of the cycle towards f_min at its end
    (MObI: f_min = f_max = 1 and one practically endless cycle, i.e. 200 warm-up steps and a constant rate)."""

    def _after(self, cycle, n):
        return self.f_min[cycle] + (self.f_max[cycle] - self.f_min[cycle]) * (self.cycle_lengths[cycle] - n) / (self.cycle_lengths[cycle])


