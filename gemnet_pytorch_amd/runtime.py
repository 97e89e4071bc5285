"""Static-shape inference runtime: hipGraphs of forward+force, replayed concurrently on dedicated HIP streams.

At the headline batch (32 molecules x 32 atoms: 18 k edges) every kernel of the path is latency-bound — a
128x128 layer over 18 k rows is ~4 us of MFMA work behind ~4 us of launch/drain — so a single in-order
stream leaves the chip partly idle between kernels.  Molecules are independent (block-diagonal index
arrays), so the batch is cut into sub-batches, each captured ONCE into its own hipGraph
(`torch.cuda.CUDAGraph`, including the autograd pass that yields F = -dE/dR) on its own stream; a step
launches all graphs back-to-back and the hardware interleaves their kernels.  No nested capture, no
collective, no host synchronisation inside a step.

The graphs read the model parameters and the `R` tensors of the sub-batches in place: `set_positions`
copies new coordinates into the captured buffers (MD with a fixed neighbour list); a changed graph
(new neighbour list) needs a new runner.
"""
import torch


class ForceGraphs:
    def __init__(self, model, batches, warmup=2):
        """`batches`: list of input dicts (reference keys, already on the HIP device), one per sub-batch."""
        if not batches:
            raise ValueError("at least one sub-batch")
        self.model = model
        # Private position buffers: autograd ties a leaf's gradient accumulator to the stream of its first backward
        # and keeps it on the tensor; positions that already went through an eager forward on the DEFAULT stream make
        # the capture below crash inside hipStreamEndCapture (the engine synchronises the capturing stream with the
        # default stream).  Fresh leaves see their first backward on this runner's own streams.
        self.batches = [dict(b, R=b["R"].detach().clone()) for b in batches]
        for b in self.batches:
            b.pop("_plan", None)
        dev = self.batches[0]["R"].device
        if dev.type != "cuda":
            raise RuntimeError("ForceGraphs needs a HIP device (no CPU fallback)")
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.batches]
        self.graphs = []
        self.outputs = []
        model.eval()
        cur = torch.cuda.current_stream(dev)
        for inputs, st in zip(self.batches, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                for _ in range(warmup):
                    model(inputs)
            cur.wait_stream(st)
        torch.cuda.synchronize(dev)
        for inputs, st in zip(self.batches, self.streams):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                out = model(inputs)
            self.graphs.append(g)
            self.outputs.append(out)
        torch.cuda.synchronize(dev)

    def set_positions(self, i, R):
        self.batches[i]["R"].detach().copy_(R)

    def replay(self):
        """Enqueue one step (all sub-batches); the calling stream waits for all of them."""
        cur = torch.cuda.current_stream()
        for g, st in zip(self.graphs, self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                g.replay()
        for st in self.streams:
            cur.wait_stream(st)
        return self.outputs

    __call__ = replay

    def energies_forces(self):
        """Concatenated (E (nMol, T), F (nAtoms, ...)) of the last replay, sub-batch order."""
        return (torch.cat([o[0] for o in self.outputs]), torch.cat([o[1] for o in self.outputs]))
