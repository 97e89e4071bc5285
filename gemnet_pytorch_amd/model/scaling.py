"""Per-layer scale factors and their one-at-a-time fitting protocol.

What the callers need (the reference's gemnet/model/layers/scaling.py:7-174 as used by fit_scaling.py:94,153-159 and
by the layer modules):

* `ScalingFactor(scale_file, name)` — a non-trainable scalar looked up by name in the scale json at construction;
  `forward(x_ref, y)` scales `y` and, while this factor is the one being fitted, records the variances of `x_ref`
  and of the scaled `y`.
* `AutomaticFit.set2fitmode()` before the model is built, then repeatedly
  `while not AutomaticFit.fitting_completed(): <run batches>; AutomaticFit.activeVar.fit()`:
  factors that are not yet in the json are fitted strictly in creation order, each from the batches that ran while it
  was active, and written to the json as they are fitted.

Implementation: the fitting order lives in one explicit `_FitSchedule` object; `AutomaticFit`'s class-level names
(`activeVar`, `queue`, `fitting_mode`) are read-through properties of it, so there is no scattered class state.
"""
import collections
import logging
import math

import numpy as np
import torch

from .utils import read_value_json, update_json


class _FitSchedule:
    """Creation-ordered schedule of the factors still to be fitted."""

    def __init__(self, enabled=False):
        self.enabled = enabled          # fit mode: newly created, not-yet-fitted factors enrol themselves
        self.current = None             # the factor whose statistics are being collected
        self.waiting = collections.deque()
        self.opened = False             # a factor has been enrolled since the last reset

    def enrol(self, fitter):
        if fitter.name in {f.name for f in self.waiting} | ({self.current.name} if self.current else set()):
            raise ValueError(f"Variable with the same name ({fitter.name}) was already added to queue!")
        if not self.opened:
            self.opened, self.current = True, fitter
        else:
            self.waiting.append(fitter)

    def advance(self):
        """The current factor is done: hand over to the next one in creation order (None when none is left)."""
        self.current = self.waiting.popleft() if self.waiting else None
        if self.current is None:
            self.opened = False

    @property
    def finished(self):
        return not self.opened


class _FitNames(type):
    """Class-level view of the schedule under the names fit_scaling.py and the layers use."""

    @property
    def activeVar(cls):
        return cls._schedule.current

    @property
    def queue(cls):
        s = cls._schedule
        return None if s.finished else list(s.waiting)

    @property
    def fitting_mode(cls):
        return cls._schedule.enabled

    @fitting_mode.setter
    def fitting_mode(cls, on):
        cls._schedule.enabled = bool(on)


class AutomaticFit(metaclass=_FitNames):
    """Base of a fittable scalar: loads its value from the json if present, otherwise (in fit mode) joins the schedule."""
    _schedule = _FitSchedule()

    def __init__(self, variable, scale_file, name):
        self.variable = variable
        self.scale_file = scale_file
        self.name = name
        stored = read_value_json(scale_file, name)
        self.fitted = stored is not None
        if self.fitted:
            with torch.no_grad():
                variable.copy_(torch.as_tensor(stored, dtype=variable.dtype))
        else:
            logging.info(f"Initialize variable {name}' to {float(variable):.3f}")
            if AutomaticFit._schedule.enabled:
                AutomaticFit._schedule.enrol(self)

    @property
    def _name(self):   # the attribute name the reference's tooling prints
        return self.name

    @staticmethod
    def reset():
        """Forget any schedule in progress (GemNet.__init__ calls this; fit mode itself is kept)."""
        AutomaticFit._schedule = _FitSchedule(enabled=AutomaticFit._schedule.enabled)

    @staticmethod
    def set2fitmode():
        AutomaticFit._schedule = _FitSchedule(enabled=True)

    @staticmethod
    def fitting_completed():
        return AutomaticFit._schedule.finished

    def is_active(self):
        return AutomaticFit._schedule.current is self


class AutoScaleFit(AutomaticFit):
    """Fits the factor so that the scaled output has the variance of the reference input:
    factor *= sqrt(mean Var(x_ref) / mean Var(y)), sample-weighted over the observed batches."""

    def __init__(self, variable, scale_file, name):
        super().__init__(variable, scale_file, name)
        self._sum_var_ref = 0.0
        self._sum_var_out = 0.0
        self._rows = 0

    def observe(self, x_ref, y):
        if self.fitted or not self.is_active():
            return
        rows = y.shape[0]
        with torch.no_grad():
            self._sum_var_ref = self._sum_var_ref + rows * torch.var(x_ref, dim=0).mean()
            self._sum_var_out = self._sum_var_out + rows * torch.var(y, dim=0).mean()
        self._rows += rows

    def fit(self):
        if not self.is_active():
            return
        if self._rows == 0:
            raise ValueError(f"Did not track the variable {self.name}. "
                             "Add observe calls to track the variance before and after.")
        var_ref = float(self._sum_var_ref) / self._rows
        var_out = float(self._sum_var_out) / self._rows
        correction = float(np.float32(math.sqrt(var_ref / var_out)))   # the json holds float32-rounded factors
        logging.info(f"Variable: {self.name}, Var_in: {var_ref:.3f}, Var_out: {var_out:.3f} "
                     f"=> Scaling factor: {correction:.3f}")
        with torch.no_grad():
            self.variable.mul_(correction)
        self.fitted = True
        update_json(self.scale_file, {self.name: float(self.variable.detach().cpu())})
        AutomaticFit._schedule.advance()


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
