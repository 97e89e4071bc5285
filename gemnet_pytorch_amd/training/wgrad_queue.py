"""Deferred, grouped weight gradients for the training step.

In the final (non-differentiable) backward of force training every `dW = X^T Y` product — ~300 of them per step for
GemNet-T: the first-order term and the double-backward terms of every Dense (trainer.py:346) — is a LEAF of the
autograd graph: nothing reads it except the accumulation into `param.grad`.  Instead of ~300 split-K launches +
~300 fold launches + ~300 autograd accumulations, `ops._MM.backward` enqueues (param, X, Y) here and `flush()`
runs ONE grouped split-K launch and ONE grouped fold that adds straight into the flat gradient buffer
(csrc/gemm_tn.hip, gn_gemm_tn_grouped_f32).  Fold order per parameter is enqueue order: deterministic.
"""

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream

PROB = np.dtype([("X", "<u8"), ("Y", "<u8"), ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("ldx", "<i4"), ("ldy", "<i4"),
                 ("splitk", "<i4"), ("kchunk", "<i4"), ("wg_begin", "<i4"), ("ws_off", "<i8"), ("alpha", "<f4"), ("pad", "<i4")])
TARGET = np.dtype([("out", "<u8"), ("n", "<i8"), ("slice_begin", "<i4"), ("slice_end", "<i4"), ("wg_begin", "<i4"),
                   ("cols", "<i4"), ("ld", "<i4"), ("pad", "<i4")])
assert PROB.itemsize == 64 and TARGET.itemsize == 40
# rows of the contraction per split-K slice.  A step queues ~480 products at once (1 900 output tiles of 64 x 64: the grid
# is full without any split), and every slice costs a (M, N) partial written and folded again: see profiles/r3_ab.txt
import os as _os
SPLIT_ROWS = int(_os.environ.get("GEMNET_WGRAD_SPLIT_ROWS", "2048"))


def grad_target(P):
    """Where the gradient of the 2-D weight-like tensor P accumulates: (address, rows, cols, ld) inside the `.grad` of a
    leaf parameter — P itself, or the leaf P is a row-pitch-preserving 2-D view of (the column blocks `W[:, a:b]` of a
    concat-Dense weight, embedding_block.py:60-75) — or None when P's gradient has to travel through autograd."""
    if not (P.is_cuda and P.dim() == 2 and P.requires_grad):
        return None
    if P.is_leaf:
        g = P.grad
        if g is None or not g.is_contiguous():
            return None
        return (g.data_ptr(), P.shape[0], P.shape[1], P.shape[1])
    base = P._base
    if base is None or not base.is_leaf or base.dim() != 2 or not base.is_contiguous():
        return None
    g = base.grad
    if g is None or not g.is_contiguous() or P.stride(1) != 1 or P.stride(0) != base.stride(0):
        return None
    off = P.storage_offset() - base.storage_offset()
    return (g.data_ptr() + 4 * off, P.shape[0], P.shape[1], base.shape[1])



def _rowmajor(t):
    return t if (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]) else t.contiguous()


class _TableSlot:
    """One pinned host table + its device copy + the event that marks the copy's completion."""

    def __init__(self, nbytes, dev):
        self.host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.copied = torch.cuda.Event()
        self.in_flight = False


class WeightGradQueue:
    """Eager steps rotate through `RING` table slots and wait on a slot's copy event before rewriting its pinned
    host buffer (the CPU may run a step ahead of the GPU: rewriting the buffer of a copy that has not executed yet
    would hand the earlier step the later step's operand addresses).  A slot used while a hipGraph is being captured
    belongs to that graph — the captured memcpy re-reads its pinned buffer at every replay — so it is moved to
    `_captured`, kept alive for the lifetime of the queue and never written again."""
    RING = 3

    def __init__(self):
        self.items = []
        self._ring = []
        self._next = 0
        self._spare = None       # pre-allocated table for the next capture
        self._captured = []
        self._keep = None

    def add(self, param, X, Y, alpha=1.0):
        """param.grad (M,N) += alpha X^T @ Y with X (K,M), Y (K,N); `param`: a leaf parameter or a column-block view of one
        (see `grad_target`, which must accept it)."""
        tgt = grad_target(param)
        assert tgt is not None and tgt[1:3] == (X.shape[1], Y.shape[1]), "not a queueable weight-gradient target"
        self.items.append((tgt, _rowmajor(X), _rowmajor(Y), param, float(alpha)))

    def add_region(self, tgt, X, Y, keep=None, alpha=1.0):
        """The same for an explicit target region (address, rows, cols, ld) of a parameter's .grad — e.g. the (C, O) block
        of one i of a bilinear weight (C, I, O): rows c with pitch I * O."""
        assert tgt[1:3] == (X.shape[1], Y.shape[1])
        self.items.append((tuple(tgt), _rowmajor(X), _rowmajor(Y), keep, float(alpha)))

    def _slot(self, nbytes, dev, capturing):
        if capturing:
            # pinned memory cannot be allocated while a stream is capturing: the graph takes the spare slot that the
            # last eager flush left behind (TrainStep.capture runs eager warm-up steps first), and owns it from now on
            slot, self._spare = self._spare, None
            if slot is None or slot.host.numel() < nbytes:
                raise RuntimeError("WeightGradQueue: run one eager step of the same batch before capturing a hipGraph")
            self._captured.append(slot)
            return slot
        if self._spare is None or self._spare.host.numel() < nbytes:
            self._spare = _TableSlot(nbytes, dev)
        if len(self._ring) < self.RING:
            self._ring.append(_TableSlot(nbytes, dev))
            return self._ring[-1]
        i = self._next
        self._next = (i + 1) % self.RING
        slot = self._ring[i]
        if slot.in_flight:
            slot.copied.synchronize()       # the previous use of this pinned buffer has been read by the GPU
            slot.in_flight = False
        if slot.host.numel() < nbytes:
            slot = self._ring[i] = _TableSlot(nbytes, dev)
        return slot

    def flush(self):
        items, self.items = self.items, []
        if not items:
            return
        dev = items[0][1].device
        probs = np.zeros(len(items), dtype=PROB)
        by_param = {}
        wg = 0
        ws_off = 0
        for i, (tgt, X, Y, _, alpha) in enumerate(items):
            K, M = X.shape
            N = Y.shape[1]
            assert Y.shape[0] == K
            splitk = max(1, min(64, K // SPLIT_ROWS))
            kchunk = (-(-K // splitk) + 15) // 16 * 16
            splitk = -(-K // kchunk)
            tiles = -(-M // 64) * -(-N // 64)
            probs[i] = (X.data_ptr(), Y.data_ptr(), M, N, K, X.stride(0), Y.stride(0), splitk, kchunk, wg, ws_off, alpha, 0)
            # keyed by the target REGION: fresh view objects of one weight block must fold into one accumulator
            by_param.setdefault(tgt, (tgt, []))[1].extend(ws_off + z * M * N for z in range(splitk))
            wg += tiles * splitk
            ws_off += splitk * M * N
        targets = np.zeros(len(by_param), dtype=TARGET)
        slices = []
        fold_wg = 0
        for j, ((addr, rows, cols, ld), offs) in enumerate(by_param.values()):
            strided = ld != cols
            targets[j] = (addr, rows * cols, len(slices), len(slices) + len(offs), fold_wg, cols if strided else 0,
                          ld if strided else 0, 0)
            slices.extend(offs)
            fold_wg += -(-(rows * cols) // 64)
        slice_off = np.asarray(slices, dtype=np.int64)
        blob = probs.tobytes() + targets.tobytes() + slice_off.tobytes()
        nbytes = len(blob)
        capturing = torch.cuda.is_current_stream_capturing()
        slot = self._slot(nbytes, dev, capturing)
        slot.host.numpy()[:nbytes] = np.frombuffer(blob, dtype=np.uint8)
        slot.dev[:nbytes].copy_(slot.host[:nbytes], non_blocking=True)
        if not capturing:
            slot.copied.record()
            slot.in_flight = True
        ws = torch.empty(ws_off, dtype=torch.float32, device=dev)
        base = _lib.addr(slot.dev)
        # the operand addresses sit in the device table: tell a recorder (hbcheck.py) what this launch reads and writes
        if _lib.TRACE is not None:
            _lib.note(reads=[x for it in items for x in it[1:3]],
                      writes=[(a, 4 * ((rows - 1) * ld + cols)) for (a, rows, cols, ld), _ in by_param.values()])
        o_t = probs.nbytes
        o_s = o_t + targets.nbytes
        check(_lib.load().gn_gemm_tn_grouped_f32(base, len(items), wg, base + o_t, len(by_param), fold_wg, base + o_s,
                                                 ptr(ws), stream()), "gn_gemm_tn_grouped_f32")
        # operands stay allocated until the launches have been enqueued after them in stream order (eager: the
        # caching allocator is stream-ordered) or, for a captured graph, for as long as the graph may be replayed
        if capturing:
            slot.keep = (items, ws)
        else:
            self._keep = (items, ws)
