"""Deferred, grouped weight gradients for the training step.

In the final (non-differentiable) backward of force training every `dW = X^T Y` product — ~300 of them per step for
GemNet-T: the first-order term and the double-backward terms of every Dense (trainer.py:346) — is a LEAF of the
autograd graph: nothing reads it except the accumulation into `param.grad`.  Instead of ~300 split-K launches +
~300 fold launches + ~300 autograd accumulations, `ops._MM.backward` enqueues (param, X, Y) here and `flush()`
runs ONE grouped split-K launch and ONE grouped fold that adds straight into the flat gradient buffer
(csrc/gemm_tn.hip, gn_gemm_tn_grouped_f32).  Fold order per parameter is enqueue order: deterministic.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream

PROB = np.dtype([("X", "<u8"), ("Y", "<u8"), ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("ldx", "<i4"), ("ldy", "<i4"),
                 ("splitk", "<i4"), ("kchunk", "<i4"), ("wg_begin", "<i4"), ("ws_off", "<i8")])
TARGET = np.dtype([("out", "<u8"), ("n", "<i8"), ("slice_begin", "<i4"), ("slice_end", "<i4"), ("wg_begin", "<i4"),
                   ("pad", "<i4")])
assert PROB.itemsize == 56 and TARGET.itemsize == 32


def _rowmajor(t):
    return t if (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]) else t.contiguous()


class WeightGradQueue:
    def __init__(self):
        self.items = []
        self._host = self._dev = None
        self._keep = None

    def add(self, param, X, Y):
        """param.grad (M,N) += X^T @ Y with X (K,M), Y (K,N)."""
        self.items.append((param, _rowmajor(X), _rowmajor(Y)))

    def flush(self):
        items, self.items = self.items, []
        if not items:
            return
        dev = items[0][1].device
        probs = np.zeros(len(items), dtype=PROB)
        by_param = {}
        wg = 0
        ws_off = 0
        for i, (P, X, Y) in enumerate(items):
            K, M = X.shape
            N = Y.shape[1]
            assert Y.shape[0] == K and tuple(P.shape) == (M, N) and P.grad is not None and P.grad.is_contiguous()
            splitk = max(1, min(64, K // 512))
            kchunk = (-(-K // splitk) + 15) // 16 * 16
            splitk = -(-K // kchunk)
            tiles = -(-M // 64) * -(-N // 64)
            probs[i] = (X.data_ptr(), Y.data_ptr(), M, N, K, X.stride(0), Y.stride(0), splitk, kchunk, wg, ws_off)
            by_param.setdefault(id(P), (P, []))[1].extend(ws_off + z * M * N for z in range(splitk))
            wg += tiles * splitk
            ws_off += splitk * M * N
        targets = np.zeros(len(by_param), dtype=TARGET)
        slices = []
        fold_wg = 0
        for j, (P, offs) in enumerate(by_param.values()):
            targets[j] = (P.grad.data_ptr(), P.numel(), len(slices), len(slices) + len(offs), fold_wg, 0)
            slices.extend(offs)
            fold_wg += -(-P.numel() // 64)
        slice_off = np.asarray(slices, dtype=np.int64)
        blob = probs.tobytes() + targets.tobytes() + slice_off.tobytes()
        nbytes = len(blob)
        capturing = torch.cuda.is_current_stream_capturing()
        if self._host is None or self._host.numel() < nbytes or self._frozen:
            # the table of a captured graph is read from the pinned buffer at every replay: never overwrite it
            assert not capturing, "run one eager step before capturing (sizes the tables), capture only once"
            self._host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
            self._dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._frozen = False
        self._host.numpy()[:nbytes] = np.frombuffer(blob, dtype=np.uint8)
        self._dev[:nbytes].copy_(self._host[:nbytes], non_blocking=True)
        ws = torch.empty(ws_off, dtype=torch.float32, device=dev)
        base = self._dev.data_ptr()
        o_t = probs.nbytes
        o_s = o_t + targets.nbytes
        check(_lib.load().gn_gemm_tn_grouped_f32(base, len(items), wg, base + o_t, len(by_param), fold_wg, base + o_s,
                                                 ptr(ws), stream()), "gn_gemm_tn_grouped_f32")
        self._keep = (items, ws)  # operands stay allocated until the launches (or the captured graph) no longer run
        self._frozen = capturing

    _frozen = False
