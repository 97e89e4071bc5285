"""Per-layer scale factors (counterpart of gemnet/model/layers/scaling.py).

`ScalingFactor` is a non-trainable scalar Parameter loaded by name from the scale json at
construction (scaling.py:150-174, lookup :68-81).  `AutomaticFit`/`AutoScaleFit` reproduce the
reference's class-level fitting queue (scaling.py:7-147) used by fit_scaling.py: variables are
fitted one at a time, in creation order, from observed input/output variances.
"""
import logging

import numpy as np
import torch

from .utils import read_value_json, update_json


class AutomaticFit:
    activeVar = None
    queue = None
    fitting_mode = False

    def __init__(self, variable, scale_file, name):
        self.variable = variable
        self.scale_file = scale_file
        self._name = name
        self._fitted = False
        self.load_maybe()
        if AutomaticFit.fitting_mode and not self._fitted:
            if AutomaticFit.activeVar is None:
                AutomaticFit.activeVar = self
                AutomaticFit.queue = []
            else:
                self._add2queue()

    def reset():
        AutomaticFit.activeVar = None
        AutomaticFit.all_processed = False

    def fitting_completed():
        return AutomaticFit.queue is None

    def set2fitmode():
        AutomaticFit.reset()
        AutomaticFit.fitting_mode = True

    def _add2queue(self):
        for var in AutomaticFit.queue:
            if self._name == var._name:
                raise ValueError(f"Variable with the same name ({self._name}) was already added to queue!")
        AutomaticFit.queue += [self]

    def set_next_active(self):
        queue = AutomaticFit.queue
        if len(queue) == 0:
            AutomaticFit.queue = None
            AutomaticFit.activeVar = None
            return
        AutomaticFit.activeVar = queue.pop(0)

    def load_maybe(self):
        value = read_value_json(self.scale_file, self._name)
        if value is None:
            logging.info(f"Initialize variable {self._name}' to {float(self.variable):.3f}")
        else:
            self._fitted = True
            with torch.no_grad():
                self.variable.copy_(torch.tensor(value))


class AutoScaleFit(AutomaticFit):
    def __init__(self, variable, scale_file, name):
        super().__init__(variable, scale_file, name)
        if not self._fitted:
            self.variance_in = 0
            self.variance_out = 0
            self.nSamples = 0

    def observe(self, x, y):
        if self._fitted or AutomaticFit.activeVar is not self:
            return
        n = y.shape[0]
        with torch.no_grad():
            self.variance_in += torch.mean(torch.var(x, dim=0)) * n
            self.variance_out += torch.mean(torch.var(y, dim=0)) * n
        self.nSamples += n

    def fit(self):
        if AutomaticFit.activeVar is not self:
            return
        if self.nSamples == 0:
            raise ValueError(f"Did not track the variable {self._name}. "
                             "Add observe calls to track the variance before and after.")
        v_in = self.variance_in / self.nSamples
        v_out = self.variance_out / self.nSamples
        value = np.sqrt(1 / float(v_out / v_in), dtype="float32")
        logging.info(f"Variable: {self._name}, Var_in: {float(v_in):.3f}, Var_out: {float(v_out):.3f} "
                     f"=> Scaling factor: {value:.3f}")
        with torch.no_grad():
            self.variable.copy_(self.variable * value)
        update_json(self.scale_file, {self._name: float(self.variable.cpu().numpy())})
        self.set_next_active()


class ScalingFactor(torch.nn.Module):
    def __init__(self, scale_file, name, device=None):
        super().__init__()
        self.scale_factor = torch.nn.Parameter(torch.tensor(1.0, device=device), requires_grad=False)
        self.autofit = AutoScaleFit(self.scale_factor, scale_file, name)

        self._cached = None  # (tensor version, python float)

    def forward(self, x_ref, y):
        y = y * self.scale_factor
        self.autofit.observe(x_ref, y)
        return y

    def value(self) -> float:
        """The factor as a host scalar (folded into GEMM epilogues by the fused path).  Read back from
        the device once per change of the parameter (tensor version counter), never per forward."""
        sf = self.scale_factor
        c = self._cached
        if c is None or c[0] != sf._version or c[2] != sf.data_ptr():
            self._cached = c = (sf._version, float(sf.detach().cpu()), sf.data_ptr())
        return c[1]
