"""Happens-before checker for a captured hipGraph (debug tool; never on the product path).

The reference runs its whole forward / backward on ONE stream (gemnet/model/gemnet.py:453-615): whatever order the
Python code has is the order of the device.  This build forks side streams (output blocks, the head of the forward),
sums gradients in place across consumers and lets torch's caching allocator recycle blocks inside a capture — and a
replayed hipGraph really runs its branches concurrently, which the eager run (bound by ~100 us of host time per launch)
never does.  Three "replay != eager" findings of round 3 were worked around by bisection.  This module replaces the
bisection with a proof obligation:

    every pair of device operations of a captured graph that touch overlapping memory, at least one of them writing,
    must be connected by a path of graph edges.

How it gets the facts (nothing is modelled; all of it is read back from the runtime):
  * memory accesses of OUR launches: every launcher passes its operands through `_lib.ptr()` and calls the library through
    `_lib.load()`; with a recorder installed both report here.  Which pointer argument is read and which is written comes
    from the `const` qualifiers of include/gemnet_hip.h (parsed below, struct fields included); launches whose operands sit
    in device-resident tables (grouped weight gradients / grouped weight packing) declare them with `note()`.
  * memory accesses of ATen operations (the autograd engine's gradient sums, copies, fills, the loss): a
    TorchDispatchMode — it follows the engine into its worker thread — reports inputs (reads), outputs and mutated
    arguments (writes) of every operation that added a node to the graph.
  * the graph: `hipStreamGetCaptureInfo_v2` hands out the hipGraph_t while the capture is in progress;
    `hipGraphGetNodes` after every operation attributes the new nodes to it; `hipGraphGetEdges` at the end gives the
    edges the capture really recorded (stream order, event waits, the engine's syncs, allocator-inserted waits).

Use:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        with hbcheck.record() as rec:
            out = step()
    races = rec.races()          # [] == every conflicting pair is ordered
    print(rec.format(races))
"""
import bisect
import contextlib
import ctypes
import os
import re
import traceback

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "gemnet_hip.h")


# ------------------------------------------------------------------------------------------ the C ABI, from the header
def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _param_kind(decl):
    """One parameter / field declaration -> (name, kind): kind 'r' / 'w' (pointer to const / to mutable data),
    'ra' (host array of pointers to const data), 'struct:<type>' (pointer to a struct), 'stream', or None (a value)."""
    decl = decl.strip()
    m = re.match(r"(.*?)([A-Za-z_]\w*)\s*(\[\w*\])?$", decl, flags=re.S)
    typ, name = m.group(1).strip(), m.group(2)
    stars = typ.count("*")
    if stars == 0:
        return name, None
    if name == "stream":
        return name, "stream"
    base = typ.replace("*", " ").replace("const", " ").split()
    if stars == 2:
        return name, "ra" if typ.lstrip().startswith("const") else "wa"
    if base and base[0].startswith("gn_") and base[0] not in ("gn_pack_job", "gn_tn_problem", "gn_tn_target"):
        return name, "struct:" + base[0]
    return name, "r" if re.match(r"const\b", typ) else "w"


def parse_header(path=_HEADER):
    """-> (functions {name: [(param, kind)]}, structs {name: [(field, kind)]}) of the C ABI."""
    text = _strip_comments(open(path).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            if "*" not in stmt:      # `int M, N, K` — values
                continue
            fields.append(_param_kind(stmt))
        structs[m.group(2)] = fields
    funcs = {}
    for m in re.finditer(r"\b(?:int|int64_t)\s+(gn_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = [p for p in (q.strip() for q in m.group(2).split(",")) if p and p != "void"]
        funcs[m.group(1)] = [_param_kind(p) for p in params]
    return funcs, structs


# ------------------------------------------------------------------------------------------------- HIP graph read-back
class _Hip:
    _inst = None

    def __init__(self):
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("hbcheck: libamdhip64 is not loaded in this process")
        self.lib = lib = ctypes.CDLL(path)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        lib.hipStreamGetCaptureInfo_v2.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong),
                                                   ctypes.POINTER(vp), ctypes.POINTER(ctypes.POINTER(vp)),
                                                   ctypes.POINTER(sz)]
        lib.hipGraphGetNodes.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz)]
        lib.hipGraphGetEdges.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(sz)]
        lib.hipGraphNodeGetType.argtypes = [vp, ctypes.POINTER(ctypes.c_int)]

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def capture_graph(self, stream_handle):
        status, gid, graph = ctypes.c_int(0), ctypes.c_ulonglong(0), ctypes.c_void_p(0)
        rc = self.lib.hipStreamGetCaptureInfo_v2(ctypes.c_void_p(stream_handle), ctypes.byref(status), ctypes.byref(gid),
                                                 ctypes.byref(graph), None, None)
        if rc != 0 or status.value != 1 or not graph.value:
            raise RuntimeError(f"hbcheck: the current stream is not capturing (rc {rc}, status {status.value})")
        return graph.value

    def nodes(self, graph):
        n = ctypes.c_size_t(0)
        if self.lib.hipGraphGetNodes(ctypes.c_void_p(graph), None, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetNodes failed")
        if n.value == 0:
            return []
        arr = (ctypes.c_void_p * n.value)()
        if self.lib.hipGraphGetNodes(ctypes.c_void_p(graph), arr, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetNodes failed")
        return [arr[i] for i in range(n.value)]

    def n_nodes(self, graph):
        n = ctypes.c_size_t(0)
        if self.lib.hipGraphGetNodes(ctypes.c_void_p(graph), None, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetNodes failed")
        return n.value

    def edges(self, graph):
        n = ctypes.c_size_t(0)
        if self.lib.hipGraphGetEdges(ctypes.c_void_p(graph), None, None, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetEdges failed")
        if n.value == 0:
            return []
        a, b = (ctypes.c_void_p * n.value)(), (ctypes.c_void_p * n.value)()
        if self.lib.hipGraphGetEdges(ctypes.c_void_p(graph), a, b, ctypes.byref(n)) != 0:
            raise RuntimeError("hipGraphGetEdges failed")
        return [(a[i], b[i]) for i in range(n.value)]

    def node_type(self, node):
        t = ctypes.c_int(-1)
        self.lib.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t))
        return t.value


NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record"}


# --------------------------------------------------------------------------------------------------------- the records
class Op:
    """One device operation as the host issued it: a C-ABI launch or an ATen call."""
    __slots__ = ("idx", "name", "stream", "reads", "writes", "nodes", "where", "unresolved")

    def __init__(self, idx, name, stream, where):
        self.idx, self.name, self.stream, self.where = idx, name, stream, where
        self.reads, self.writes, self.nodes, self.unresolved = [], [], [], []

    def __repr__(self):
        return f"#{self.idx} {self.name} [stream {self.stream:#x}] at {self.where}"


def _span_bytes(t):
    """Bytes from data_ptr() to the end of the last element the tensor can address."""
    if t.numel() == 0:
        return 0
    last = sum((s - 1) * st for s, st in zip(t.shape, t.stride()))
    return (last + 1) * t.element_size()


def _where(skip_prefixes=("hbcheck.py", "_lib.py")):
    """Innermost frames inside this package (not the recorder's own) — enough to name the launch site."""
    out = []
    for fr in reversed(traceback.extract_stack(limit=40)):
        base = os.path.basename(fr.filename)
        if base in skip_prefixes or "-packages/torch/" in fr.filename or "/contextlib.py" in fr.filename:
            continue
        out.append(f"{base}:{fr.lineno}:{fr.name}")
        if len(out) == 4:
            break
    return " < ".join(out)


class Recorder:
    """Installed as `_lib.TRACE` while a capture is recorded."""

    def __init__(self, graph_source=None, keep=False):
        self.funcs, self.structs = parse_header()
        # keep: hold a reference to every tensor an operation touches (nothing is recycled inside the capture, every
        # intermediate survives the replay) and remember which operation read / wrote which tensor — `replay_diff`
        self.keep = bool(keep)
        self.tensors = {}        # address -> tensor, of the extents in `known`
        self.touched = []        # (op index, 'r' | 'w', description, tensor)
        self.ops = []
        self.pending = {}        # address -> (bytes, description) of the tensors touched since the last launch
        self.known = {}          # every extent ever touched: fallback for derived addresses (base + offset)
        self._known_sorted = None
        self.notes = []
        self.graph_source = graph_source     # test hook: object with .n_nodes() / .nodes() / .edges() / .node_type()
        self.graph = None
        self.seen = set()
        self.n_seen = 0
        self.orphans = 0
        self.edges_ = None
        self.types_ = {}
        self._proxy = None

    # -- graph side
    def _begin(self):
        if self.graph_source is None:
            hip = _Hip.get()
            g = hip.capture_graph(torch.cuda.current_stream().cuda_stream)

            class _Src:
                n_nodes = staticmethod(lambda: hip.n_nodes(g))
                nodes = staticmethod(lambda: hip.nodes(g))
                edges = staticmethod(lambda: hip.edges(g))
                node_type = staticmethod(hip.node_type)
            self.graph_source = _Src
        self.seen = set(self.graph_source.nodes())     # whatever the capture held before the recorder started
        self.n_seen = len(self.seen)

    def _new_nodes(self):
        if self.graph_source.n_nodes() == self.n_seen:
            return []
        now = self.graph_source.nodes()
        new = [n for n in now if n not in self.seen]
        self.seen.update(new)
        self.n_seen = len(now)
        return new

    def _sweep_orphans(self):
        """Nodes that appeared outside any recorded operation (something launched behind the recorder's back)."""
        new = self._new_nodes()
        if new:
            op = Op(len(self.ops), "<unrecorded>", 0, _where())
            op.nodes = new
            self.ops.append(op)
            self.orphans += len(new)

    def _finish(self):
        self._sweep_orphans()
        self.edges_ = self.graph_source.edges()
        self.types_ = {n: self.graph_source.node_type(n) for n in self.seen}

    # -- access side
    def touch(self, t):
        n = _span_bytes(t)
        if n:
            a = t.data_ptr()
            d = (n, f"{tuple(t.shape)} {str(t.dtype).replace('torch.', '')}")
            old = self.pending.get(a)
            if old is None or old[0] < n:
                self.pending[a] = d
            old = self.known.get(a)
            if old is None or old[0] < n:
                self.known[a] = d
                self._known_sorted = None
                if self.keep:
                    self.tensors[a] = t
            elif self.keep and a not in self.tensors:
                self.tensors[a] = t

    def note(self, reads=(), writes=()):
        """Operands of the NEXT launch that the recorder cannot see in its argument list (device-resident tables).
        Entries: tensors, or (address, bytes) pairs."""
        self.notes.append((list(reads), list(writes)))

    def _extent(self, addr):
        """(lo, hi, description) of the access that starts at `addr`, or None."""
        hit = self.pending.get(addr) or self.known.get(addr)
        if hit is not None:
            return addr, addr + hit[0], hit[1]
        if self._known_sorted is None:
            self._known_sorted = sorted(self.known)
        i = bisect.bisect_right(self._known_sorted, addr) - 1
        if i >= 0:
            base = self._known_sorted[i]
            n, desc = self.known[base]
            if addr < base + n:
                return addr, base + n, desc + f" +{addr - base}"
        return None

    def _add(self, op, addr, kind, what):
        if not addr:
            return
        ext = self._extent(addr)
        if ext is None:
            op.unresolved.append((what, addr))
            return
        (op.writes if kind == "w" else op.reads).append((ext[0], ext[1], f"{what} {ext[2]}"))
        if self.keep:
            base = addr if addr in self.tensors else next((b for b in (self._known_sorted or ()) if b <= addr < b + self.known[b][0]
                                                          and b in self.tensors), None)
            if base is not None:
                self.touched.append((op.idx, kind, f"{what} {ext[2]}", self.tensors[base]))

    @staticmethod
    def _value(arg):
        if arg is None:
            return 0
        if isinstance(arg, int):
            return arg
        v = getattr(arg, "value", None)
        return int(v) if v else 0

    def _struct_fields(self, op, obj, sname, prefix):
        for fname, kind in self.structs[sname]:
            if kind in ("r", "w"):
                self._add(op, getattr(obj, fname) or 0, kind, prefix + fname)

    def launch(self, name, args):
        from . import _lib
        sig = self.funcs.get(name)
        op = Op(len(self.ops), name, 0, _where())
        if sig is None or len(sig) != len(args):
            raise RuntimeError(f"hbcheck: {name} called with {len(args)} arguments, header declares "
                               f"{None if sig is None else len(sig)}")
        for (pname, kind), arg in zip(sig, args):
            if kind is None:
                continue
            if kind == "stream":
                op.stream = self._value(arg)
            elif kind in ("r", "w"):
                self._add(op, self._value(arg), kind, pname)
            elif kind in ("ra", "wa"):
                for i in range(len(arg)):
                    self._add(op, arg[i] or 0, kind[0], f"{pname}[{i}]")
            elif kind == "struct:gn_gemm_args":
                self._struct_fields(op, arg._obj, "gn_gemm_args", "args.")
            elif kind == "struct:gn_chain_args":
                ca = _lib.ChainArgs.from_address(self._value(arg))
                for i in range(ca.n_ops):
                    self._struct_fields(op, ca.ops[i], "gn_chain_op", f"ops[{i}].")
            else:
                raise RuntimeError(f"hbcheck: no rule for parameter {pname} ({kind}) of {name}")
        for reads, writes in self.notes:
            for lst, kind in ((reads, "r"), (writes, "w")):
                for x in lst:
                    if torch.is_tensor(x):
                        self.touch(x)
                        self._add(op, x.data_ptr(), kind, "table operand")
                    else:
                        (op.writes if kind == "w" else op.reads).append((x[0], x[0] + x[1], "table operand region"))
        self.notes = []
        self.pending = {}
        op.nodes = self._new_nodes()
        if op.nodes:
            self.ops.append(op)

    def aten(self, func, args, kwargs, out):
        new = self._new_nodes()
        if not new:
            return
        where = _where()
        if not where:      # issued by the autograd engine itself (gradient fan-in sums, AccumulateGrad): name the node at work
            try:
                node = torch._C._current_autograd_node()
                where = "engine, after " + (node.name() if node is not None else "?")
            except Exception:  # noqa: BLE001
                where = "engine"
        op = Op(len(self.ops), str(func), torch.cuda.current_stream().cuda_stream, where)
        op.nodes = new
        schema = getattr(func, "_schema", None)
        mutated = set()
        flat_args = []

        def walk(x, is_w):
            if torch.is_tensor(x):
                flat_args.append((x, is_w))
            elif isinstance(x, (list, tuple)):
                for y in x:
                    walk(y, is_w)

        if schema is not None:
            sargs = schema.arguments
            for i, a in enumerate(args):
                is_w = i < len(sargs) and sargs[i].alias_info is not None and sargs[i].alias_info.is_write
                walk(a, is_w)
            for k, a in (kwargs or {}).items():
                sa = next((s for s in sargs if s.name == k), None)
                walk(a, sa is not None and sa.alias_info is not None and sa.alias_info.is_write)
        else:
            for a in list(args) + list((kwargs or {}).values()):
                walk(a, False)
        for t, is_w in flat_args:
            if t.is_cuda and t.numel():
                n = _span_bytes(t)
                desc = f"{tuple(t.shape)} {str(t.dtype).replace('torch.', '')}"
                (op.writes if is_w else op.reads).append((t.data_ptr(), t.data_ptr() + n, "arg " + desc))
                if self.keep:
                    self.touched.append((op.idx, "w" if is_w else "r", "arg " + desc, t))
                if is_w:
                    mutated.add(t.data_ptr())
        outs = []
        walk_out = [out]
        while walk_out:
            x = walk_out.pop()
            if torch.is_tensor(x):
                outs.append(x)
            elif isinstance(x, (list, tuple)):
                walk_out.extend(x)
        for t in outs:
            if t.is_cuda and t.numel() and t.data_ptr() not in mutated:
                # a fresh result — or a view of an input (no write then; views do not add nodes, but an op may return
                # an alias of an argument it only read)
                if any(t.data_ptr() == a.data_ptr() and not w for a, w in flat_args if a.is_cuda):
                    continue
                n = _span_bytes(t)
                op.writes.append((t.data_ptr(), t.data_ptr() + n,
                                  f"result {tuple(t.shape)} {str(t.dtype).replace('torch.', '')}"))
                if self.keep:
                    self.touched.append((op.idx, "w", f"result {tuple(t.shape)}", t))
        self.ops.append(op)

    # -- replay-to-replay comparison of every intermediate (needs keep=True)
    def snapshot(self):
        """Device copies of every tensor some operation wrote (call after a replay + synchronize)."""
        seen, out = set(), []
        for idx, kind, what, t in self.touched:
            if kind == "w" and id(t) not in seen:
                seen.add(id(t))
                out.append((t, t.detach().clone()))
        return out

    def dump(self, path):
        """The captured graph as JSON: operations (index, name, stream, launch site, node ids, accesses) and edges."""
        import json
        with open(path, "w") as f:
            json.dump(dict(ops=[dict(idx=o.idx, name=o.name, stream=o.stream, where=o.where, nodes=o.nodes,
                                     reads=o.reads, writes=o.writes) for o in self.ops],
                           edges=self.edges_, types={str(k): v for k, v in self.types_.items()}), f)

    def replay_diff(self, snap, limit=12):
        """Compare the current contents with `snap` (another replay): the operations, in issue order, that wrote a tensor
        whose contents differ, and for each whether any tensor it READ differs too.  An operation whose inputs are
        identical and whose output is not is where the two replays part."""
        differs = {id(t): not torch.equal(t, c) for t, c in snap}
        lines, n = [], 0
        by_op = {}
        for idx, kind, what, t in self.touched:
            by_op.setdefault(idx, []).append((kind, what, t))
        ops = {o.idx: o for o in self.ops}
        for idx in sorted(by_op):
            w_bad = []
            for kind, what, t in by_op[idx]:
                if kind == "w" and differs.get(id(t)):
                    c = next(c for tt, c in snap if tt is t)
                    d = (t != c)
                    rows = d.reshape(d.shape[0], -1).any(dim=1).nonzero().flatten() if d.dim() > 0 and d.shape[0] else d.nonzero()
                    w_bad.append(f"{what}: {int(d.sum())} of {d.numel()} elements in {rows.numel()} rows "
                                 f"[{int(rows.min()) if rows.numel() else -1}..{int(rows.max()) if rows.numel() else -1}], "
                                 f"max |d| {float((t.float() - c.float()).abs().max()):.2e} of max |x| {float(c.float().abs().max()):.2e}")
            if not w_bad:
                continue
            r_bad = [what for kind, what, t in by_op[idx] if kind == "r" and differs.get(id(t))]
            n += 1
            if n <= limit:
                lines.append(f"  {ops.get(idx)!r}\n      writes differing: {w_bad}\n      reads differing: {r_bad or 'NONE'}")
        return n, "\n".join(lines)

    # -- the library proxy
    def proxy(self, lib):
        if self._proxy is None or self._proxy._lib is not lib:
            self._proxy = _LibProxy(lib, self)
        return self._proxy

    # -- analysis
    def races(self):
        return find_races(self.ops, self.edges_)

    def summary(self):
        kinds = {}
        for t in self.types_.values():
            kinds[NODE_TYPES.get(t, str(t))] = kinds.get(NODE_TYPES.get(t, str(t)), 0) + 1
        streams = sorted({o.stream for o in self.ops})
        unresolved = sum(len(o.unresolved) for o in self.ops)
        return dict(ops=len(self.ops), nodes=len(self.seen), edges=len(self.edges_ or ()), node_kinds=kinds,
                    streams=len(streams), unrecorded_nodes=self.orphans, unresolved_pointers=unresolved)

    def format(self, races, limit=20):
        lines = [f"hbcheck: {self.summary()}"]
        if not races:
            lines.append("hbcheck: every conflicting pair of operations is ordered by graph edges")
        seen = set()
        for r in races:
            key = (r["a"].name, r["a"].where, r["b"].name, r["b"].where, r["what_a"], r["what_b"])
            if key in seen:
                continue
            seen.add(key)
            if len(seen) > limit:
                lines.append(f"... {len(races)} unordered pairs in total")
                break
            lines.append(f"UNORDERED {r['kind']}: bytes [{r['lo']:#x}, {r['hi']:#x})\n    {r['a']!r}\n      {r['what_a']}\n"
                         f"    {r['b']!r}\n      {r['what_b']}")
        return "\n".join(lines)


class _LibProxy:
    def __init__(self, lib, rec):
        self._lib, self._rec, self._cache = lib, rec, {}

    def __getattr__(self, name):
        hit = self._cache.get(name)
        if hit is not None:
            return hit
        fn = getattr(self._lib, name)
        sig = self._rec.funcs.get(name)
        if sig is None or not any(k == "stream" for _, k in sig):
            self._cache[name] = fn       # queries (gn_abi_version, gn_gemm_tn_splitk, ...): no launch
            return fn
        rec = self._rec

        def call(*args):
            rec._sweep_orphans()         # nodes that appeared since the last recorded operation are nobody's
            rc = fn(*args)
            rec.launch(name, args)
            return rc
        self._cache[name] = call
        return call


def find_races(ops, edges):
    """ops: [Op] with .nodes / .reads / .writes ((lo, hi, what)); edges: [(from_node, to_node)].
    -> list of dicts (a, b, kind, lo, hi, what_a, what_b) for every conflicting pair without a path between them."""
    nodes = []
    for o in ops:
        nodes.extend(o.nodes)
    for a, b in edges:
        nodes.append(a), nodes.append(b)
    index = {}
    for n in nodes:
        if n not in index:
            index[n] = len(index)
    N = len(index)
    succ = [[] for _ in range(N)]
    indeg = [0] * N
    for a, b in edges:
        succ[index[a]].append(index[b])
        indeg[index[b]] += 1
    order = [i for i in range(N) if indeg[i] == 0]
    for i in order:                       # Kahn; `order` grows while it is walked
        for j in succ[i]:
            indeg[j] -= 1
            if indeg[j] == 0:
                order.append(j)
    if len(order) != N:
        raise RuntimeError("hbcheck: the captured graph has a cycle?")
    reach = [0] * N                        # bit j of reach[i]: a path i -> j exists
    for i in reversed(order):
        r = 0
        for j in succ[i]:
            r |= reach[j] | (1 << j)
        reach[i] = r

    def ordered(x, y):
        xs, ys = [index[n] for n in x.nodes], [index[n] for n in y.nodes]
        fwd = all((reach[i] >> j) & 1 for i in xs for j in ys)
        return fwd or all((reach[j] >> i) & 1 for i in xs for j in ys)

    acc = []
    for o in ops:
        if not o.nodes:
            continue
        for lo, hi, what in o.reads:
            acc.append((lo, hi, 0, o, what))
        for lo, hi, what in o.writes:
            acc.append((lo, hi, 1, o, what))
    acc.sort(key=lambda a: (a[0], a[1]))
    races, checked = [], {}
    active = []
    for cur in acc:
        lo, hi, w, o, what = cur
        active = [a for a in active if a[1] > lo]
        for a in active:
            if a[3] is o or not (w or a[2]):
                continue
            key = (a[3].idx, o.idx) if a[3].idx < o.idx else (o.idx, a[3].idx)
            ok = checked.get(key)
            if ok is None:
                ok = checked[key] = ordered(a[3], o)
            if not ok:
                first, second = (a, cur) if a[3].idx < o.idx else (cur, a)
                kind = {(1, 1): "WAW", (1, 0): "RAW", (0, 1): "WAR"}[(first[2], second[2])]
                races.append(dict(a=first[3], b=second[3], kind=kind, lo=max(a[0], lo), hi=min(a[1], hi),
                                  what_a=first[4], what_b=second[4]))
        active.append(cur)
    return races


@contextlib.contextmanager
def record(keep=False):
    """Record every device operation issued inside the block; must be entered while the current stream is capturing.
    keep: see `Recorder` (for `replay_diff`)."""
    import torch.utils._python_dispatch  # noqa: F401  (the attribute is not loaded by `import torch` alone)
    from . import _lib
    if _lib.TRACE is not None:
        raise RuntimeError("hbcheck: a recorder is already installed")
    rec = Recorder(keep=keep)
    rec._begin()
    _lib.TRACE = rec
    try:
        with _spy(rec):
            yield rec
    finally:
        _lib.TRACE = None
    rec._finish()


def _spy(rec):
    from torch.utils._python_dispatch import TorchDispatchMode

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            rec._sweep_orphans()
            out = func(*args, **(kwargs or {}))
            rec.aten(func, args, kwargs, out)
            return out
    return Spy()
