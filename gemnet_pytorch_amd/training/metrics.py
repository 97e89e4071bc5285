"""Running metrics of the training harness.

Call surface = what the reference's train.ipynb and Trainer use (gemnet/training/metrics.py:6-159 is the
counterpart): `Metrics(tag, keys, ex)` with `update_state(nsamples, **values)`, `result(append_tag)`, `.loss`,
`write(summary_writer, step)`, `reset_states()`; `BestMetrics(path, metrics, assert_exist)` with
`inititalize()` (sic), `restore()`, `update(step, metrics)`, `items()`, `write(...)`, `.loss`, `.step`.

Own design: every tracked quantity is one row of a float64 table (column 0: sum of weight * value, column 1: sum
of the weights); values arriving as device tensors are reduced to one Python float per update (the mean over their
elements — what the reference takes at read-out), so the table never holds device memory.
"""
import logging
import os

import numpy as np
import torch


def _scalar(v):
    """Mean over the elements of a tensor / array / number, as a Python float."""
    if torch.is_tensor(v):
        return float(v.detach().double().mean())
    return float(np.mean(v))


class MeanMetric:
    """Weighted running mean of one quantity (a view of one row of the owning table, or stand-alone)."""
    __slots__ = ("_row",)

    def __init__(self, row=None):
        self._row = np.zeros(2) if row is None else row

    def update_state(self, values, sample_weight):
        self._row += (sample_weight * _scalar(values), sample_weight)

    def result(self):
        total, weight = self._row
        return total / weight if weight else float("nan")

    def reset_states(self):
        self._row[:] = 0.0


class Metrics:
    def __init__(self, tag, keys, ex=None):
        if "loss" not in keys:
            raise AssertionError("the tracked keys must contain 'loss'")
        self.tag, self.keys, self.ex = tag, list(keys), ex
        self._table = np.zeros((len(self.keys), 2))
        self._index = {k: i for i, k in enumerate(self.keys)}
        self.mean_metrics = {k: MeanMetric(self._table[i]) for k, i in self._index.items()}

    def update_state(self, nsamples, **updates):
        unknown = set(updates) - set(self._index)
        if unknown:
            raise AssertionError(f"untracked metrics: {sorted(unknown)}")
        for name, value in updates.items():
            self._table[self._index[name]] += (nsamples * _scalar(value), nsamples)

    def _mean(self, name):
        total, weight = self._table[self._index[name]]
        return float(total / weight) if weight else float("nan")

    def result(self, append_tag=True):
        suffix = f"_{self.tag}" if append_tag else ""
        return {name + suffix: self._mean(name) for name in self.keys}

    @property
    def loss(self):
        return self._mean("loss")

    def write(self, summary_writer, step):
        info = None if self.ex is None else self.ex.current_run.info
        for name, value in self.result().items():
            summary_writer.add_scalar(name, value, global_step=step)
            if info is not None:
                info.setdefault(name, []).append(value)
        if info is not None:
            info.setdefault(f"step_{self.tag}", []).append(step)

    def reset_states(self):
        self._table[:] = 0.0


class BestMetrics:
    """Best validation metrics so far, mirrored in `<path>/best_metrics.npz` after every change."""
    FILE = "best_metrics.npz"

    def __init__(self, path, metrics, assert_exist=True):
        self.path = os.path.join(path, self.FILE)
        self.metrics = metrics
        self.assert_exist = assert_exist
        self.state = {}

    def _save(self):
        np.savez(self.path, **self.state)

    def inititalize(self):  # (sic) the reference's spelling is part of the call surface
        self.state = dict.fromkeys(self.metrics.result(), np.inf)
        self.state["step"] = 0
        self._save()

    initialize = inititalize

    def restore(self):
        try:
            with np.load(self.path) as stored:
                self.state = {name: stored[name].item() for name in stored.files}
        except FileNotFoundError:
            if self.assert_exist:
                raise UserWarning(f"no best-metrics file at {self.path}") from None
            logging.warning("no best-metrics file at %s: starting from fresh best metrics", self.path)
            self.inititalize()

    def items(self):
        return self.state.items()

    def update(self, step, metrics):
        self.state = {**self.state, **metrics.result(), "step": step}
        self._save()

    def write(self, summary_writer, step):
        for name, value in self.state.items():
            if name != "step":
                summary_writer.add_scalar(f"{name}_best", value, step)

    @property
    def loss(self):
        return self.state["loss_val"]

    @property
    def step(self):
        return self.state["step"]
