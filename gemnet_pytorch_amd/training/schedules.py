"""Learning-rate schedules (counterparts of gemnet/training/schedules.py:3-46 and the plateau decay of
gemnet/training/trainer.py:523-717)."""
import logging

import numpy as np
from torch.optim.lr_scheduler import LambdaLR


class LinearWarmupExponentialDecay(LambdaLR):
    """lr(step) = base * min((step+1)/warmup, 1) * decay_rate ** (step/decay_steps)  [floor if staircase]."""

    def __init__(self, optimizer, warmup_steps, decay_steps, decay_rate, staircase=False, last_step=-1,
                 verbose=False):
        assert decay_rate <= 1
        warmup_steps = max(int(warmup_steps), 1)

        def factor(step):
            exponent = step / decay_steps
            if staircase:
                exponent = int(exponent)
            return min((step + 1) / warmup_steps, 1) * decay_rate ** exponent

        super().__init__(optimizer, factor, last_epoch=last_step)  # torch >= 2.7 dropped `verbose`


class ReduceLROnPlateau:
    """Multiply the schedules' base learning rates by `factor` when the monitored value has not improved
    (relative threshold) for more than `patience` evaluations; `cooldown` evaluations are ignored after
    each reduction.  Works on one or several (optimizer, schedule) pairs."""

    def __init__(self, optimizer, scheduler, factor=0.1, patience=10, threshold=1e-4, max_reduce=10, cooldown=0,
                 threshold_mode="rel", min_lr=0, eps=1e-8, mode="min", verbose=False):
        if factor >= 1.0:
            raise ValueError(f"Factor should be < 1.0 but is {factor}.")
        if mode not in ("min", "max"):
            raise ValueError("mode " + mode + " is unknown!")
        if threshold_mode not in ("rel", "abs"):
            raise ValueError("threshold mode " + threshold_mode + " is unknown!")
        unwrap = lambda x: list(getattr(x, "wrapped", x if isinstance(x, (list, tuple)) else [x]))
        self.optimizer, self.scheduler = unwrap(optimizer), unwrap(scheduler)
        assert len(self.optimizer) == len(self.scheduler)
        self.factor, self.patience, self.cooldown, self.verbose = factor, patience, cooldown, verbose
        self.mode, self.threshold, self.threshold_mode, self.eps = mode, threshold, threshold_mode, eps
        self.best = np.inf if mode == "min" else -np.inf
        self.cooldown_counter = 0
        self.num_bad_steps = 0
        self.last_step = 0
        self._reduce_counter = 0

    @property
    def in_cooldown(self):
        return self.cooldown_counter > 0

    def is_better(self, a, best):
        if self.threshold_mode == "rel":
            return a < best * (1.0 - self.threshold) if self.mode == "min" else a > best * (1.0 + self.threshold)
        return a < best - self.threshold if self.mode == "min" else a > best + self.threshold

    def step(self, metrics):
        current = float(metrics)
        self.last_step += 1
        if self.is_better(current, self.best):
            self.best, self.num_bad_steps = current, 0
        else:
            self.num_bad_steps += 1
        if self.in_cooldown:
            self.cooldown_counter -= 1
            self.num_bad_steps = 0
        if self.num_bad_steps > self.patience:
            self._reduce_counter += 1
            for schedule in self.scheduler:
                schedule.base_lrs = [lr * self.factor for lr in schedule.base_lrs]
            if self.verbose:
                logging.info(f"Step {self.last_step}: reducing on plateu by {self.factor}.")
            self.cooldown_counter, self.num_bad_steps = self.cooldown, 0

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k not in ("optimizer", "scheduler")}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)
