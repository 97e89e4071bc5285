"""Running metrics (counterparts of gemnet/training/metrics.py:6-159): sample-weighted means of the
tracked quantities, best-so-far bookkeeping in an npz, TensorBoard / Sacred writers."""
import logging
import os

import numpy as np
import torch


class MeanMetric:
    def __init__(self):
        self.reset_states()

    def update_state(self, values, sample_weight):
        self.values += sample_weight * values
        self.sample_weights += sample_weight

    def result(self):
        return self.values / self.sample_weights

    def reset_states(self):
        self.sample_weights = 0
        self.values = 0


class Metrics:
    def __init__(self, tag, keys, ex=None):
        assert "loss" in keys
        self.tag, self.keys, self.ex = tag, keys, ex
        self.mean_metrics = {k: MeanMetric() for k in keys}

    def update_state(self, nsamples, **updates):
        assert set(updates).issubset(self.keys)
        for k, v in updates.items():
            self.mean_metrics[k].update_state(torch.as_tensor(v).detach().cpu(), sample_weight=nsamples)

    def result(self, append_tag=True):
        return {(f"{k}_{self.tag}" if append_tag else k): float(torch.as_tensor(self.mean_metrics[k].result()).mean())
                for k in self.keys}

    @property
    def loss(self):
        return float(torch.as_tensor(self.mean_metrics["loss"].result()).mean())

    def write(self, summary_writer, step):
        for k, v in self.result().items():
            summary_writer.add_scalar(k, v, global_step=step)
            if self.ex is not None:
                self.ex.current_run.info.setdefault(k, []).append(v)
        if self.ex is not None:
            self.ex.current_run.info.setdefault(f"step_{self.tag}", []).append(step)

    def reset_states(self):
        for m in self.mean_metrics.values():
            m.reset_states()


class BestMetrics:
    def __init__(self, path, metrics, assert_exist=True):
        self.path = os.path.join(path, "best_metrics.npz")
        self.metrics, self.assert_exist, self.state = metrics, assert_exist, {}

    def inititalize(self):  # (sic) the reference's spelling is part of the call surface
        self.state = {f"{k}_{self.metrics.tag}": np.inf for k in self.metrics.keys}
        self.state["step"] = 0
        np.savez(self.path, **self.state)

    initialize = inititalize

    def restore(self):
        if os.path.isfile(self.path):
            self.state = {k: v.item() for k, v in np.load(self.path).items()}
            return
        msg = f"Best metrics can not be restored as the file does not exist in the given path: {self.path}"
        if self.assert_exist:
            raise UserWarning(msg)
        logging.warning(msg + "\n Will initialize the best metrics.")
        self.inititalize()

    def items(self):
        return self.state.items()

    def update(self, step, metrics):
        self.state["step"] = step
        self.state.update(metrics.result())
        np.savez(self.path, **self.state)

    def write(self, summary_writer, step):
        for k, v in self.state.items():
            if k != "step":
                summary_writer.add_scalar(k + "_best", v, step)

    @property
    def loss(self):
        return self.state["loss_val"]

    @property
    def step(self):
        return self.state["step"]
