"""The happens-before checker (gemnet_pytorch_amd/hbcheck.py) on the CPU: header parsing against the ctypes signatures,
the graph analysis on hand-made graphs, and the recorder end to end on a fake library + fake capture graph.  The run
over real captured steps is tests/test_gpu_hbcheck.py."""
import ctypes

import torch

from gemnet_pytorch_amd import _lib, hbcheck
from gemnet_pytorch_amd import kernels as K


def test_header_covers_every_bound_entry_point():
    funcs, structs = hbcheck.parse_header()
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in funcs, name
        assert len(funcs[name]) == len(argtypes), (name, funcs[name])
        for (pname, kind), ct in zip(funcs[name], argtypes):
            is_ptr = ct is ctypes.c_void_p or hasattr(ct, "contents") or (isinstance(ct, type) and issubclass(ct, ctypes._Pointer))
            assert (kind is not None) == is_ptr, (name, pname, kind, ct)
        assert sum(k == "stream" for _, k in funcs[name]) == (0 if name == "gn_gemm_tn_splitk" else 1), name
    # const-correctness the checker relies on, spot-checked
    g = dict(funcs["gn_rbf_aggregate_bwd_f32"])
    assert g["g_out"] == "r" and g["m"] == "r" and g["g_m"] == "w" and g["g_rbf"] == "w"
    s = dict(structs["gn_chain_op"])
    assert s["src"] == "r" and s["W"] == "r" and s["pre_out"] == "w" and s["out"] == "w" and s["out2"] == "w"
    assert s["srcP"] == "r" and s["res_g"] == "r" and s["rows"] == "r"
    a = dict(structs["gn_gemm_args"])
    assert a["A"] == "r" and a["C"] == "w" and a["splitk_ws"] == "w" and a["ridx"] == "r"
    assert dict(funcs["gn_bil_dy_multi_f32"])["dSm_list"] == "ra"
    assert dict(funcs["gn_gemm_f32_cfg"])["args"] == "struct:gn_gemm_args"
    assert dict(funcs["gn_chain_split_f32"])["args"] == "struct:gn_chain_args"
    # every field of the ctypes mirrors that is a pointer is classified
    for cls, sname in ((_lib.GemmArgs, "gn_gemm_args"), (_lib.ChainOp, "gn_chain_op")):
        ptr_fields = [n for n, ct in cls._fields_ if ct is ctypes.c_void_p]
        assert sorted(ptr_fields) == sorted(n for n, _ in structs[sname]), sname


def _op(i, nodes, reads=(), writes=()):
    o = hbcheck.Op(i, f"op{i}", 0, "here")
    o.nodes, o.reads, o.writes = list(nodes), [(a, b, "r") for a, b in reads], [(a, b, "w") for a, b in writes]
    return o


def test_find_races_on_hand_made_graphs():
    # chain 1 -> 2 -> 3 and a side branch 1 -> 4; 3 joins nothing
    edges = [(1, 2), (2, 3), (1, 4)]
    ops = [_op(0, [1], writes=[(0, 100)]),                     # producer
           _op(1, [2], reads=[(0, 100)], writes=[(100, 200)]),
           _op(2, [3], reads=[(100, 200)], writes=[(0, 50)]),  # reuses the producer's block: ordered after op1 (2 -> 3)
           _op(3, [4], reads=[(40, 60)])]                      # side-branch reader of the block op2 rewrites: unordered (issued later: "RAW")
    races = hbcheck.find_races(ops, edges)
    assert [(r["a"].idx, r["b"].idx, r["kind"]) for r in races] == [(2, 3, "RAW")]
    assert (races[0]["lo"], races[0]["hi"]) == (40, 50)
    # the join 4 -> 3 orders the pair
    assert hbcheck.find_races(ops, edges + [(4, 3)]) == []
    # read-read never conflicts; two unordered writers do
    ops = [_op(0, [1], writes=[(0, 8)]), _op(1, [2], reads=[(0, 8)]), _op(2, [4], reads=[(0, 8)]),
           _op(3, [3], writes=[(300, 400)]), _op(4, [5], writes=[(350, 360)])]
    races = hbcheck.find_races(ops, [(1, 2), (1, 4), (2, 3), (1, 5)])
    assert [(r["a"].idx, r["b"].idx, r["kind"]) for r in races] == [(3, 4, "WAW")]
    # an operation with two nodes (split-K + fold) is ordered only if every node is
    ops = [_op(0, [1, 2], writes=[(0, 8)]), _op(1, [3], reads=[(0, 8)])]
    assert len(hbcheck.find_races(ops, [(1, 2), (1, 3)])) == 1
    assert hbcheck.find_races(ops, [(1, 2), (2, 3)]) == []


class _FakeGraph:
    """Stands in for the hipGraph under capture: one node per launch; `stream` decides the edges (same stream: chained)."""

    def __init__(self):
        self.nodes_, self.edges_, self.tail = [], [], {}

    def add(self, stream, after=()):
        n = len(self.nodes_) + 1
        self.nodes_.append(n)
        if stream in self.tail:
            self.edges_.append((self.tail[stream], n))
        for a in after:
            self.edges_.append((self.tail[a], n))
        self.tail[stream] = n

    def n_nodes(self):
        return len(self.nodes_)

    def nodes(self):
        return list(self.nodes_)

    def edges(self):
        return list(self.edges_)

    def node_type(self, n):
        return 0


class _FakeLib:
    def __init__(self, graph):
        self.graph, self.cur, self.after = graph, 1, ()

    def __getattr__(self, name):
        def fn(*args):
            self.graph.add(self.cur, self.after)
            self.after = ()
            return 0
        return fn


def test_recorder_classifies_launch_operands_and_finds_a_cross_stream_reuse(monkeypatch):
    g = _FakeGraph()
    lib = _FakeLib(g)
    rec = hbcheck.Recorder(graph_source=g)
    rec._begin()
    monkeypatch.setattr(_lib, "_lib", lib)
    monkeypatch.setattr(_lib, "TRACE", rec)
    monkeypatch.setattr(_lib, "stream", lambda: ctypes.c_void_p(lib.cur))
    monkeypatch.setattr(_lib, "require_device", lambda *t: None)
    monkeypatch.setattr(K, "stream", _lib.stream)
    monkeypatch.setattr(K, "require_device", _lib.require_device)
    x = torch.zeros(8, 16)
    idx = torch.arange(4, dtype=torch.int32)
    monkeypatch.setattr(torch, "empty", lambda *a, **k: torch.zeros(*a, **{q: v for q, v in k.items() if q != "device"}))
    y = K.gather(x, idx)                     # stream 1: reads x, idx; writes y
    lib.cur = 2
    z = K.gather(y, idx)                     # stream 2, NO edge from stream 1: RAW on y
    lib.cur = 1
    K.gather_mul(x, idx, y)                  # stream 1 again: reads y (no conflict with the reader on stream 2)
    rec._finish()
    assert [o.name for o in rec.ops] == ["gn_gather_rows_f32", "gn_gather_rows_f32", "gn_gather_mul_f32"]
    o0 = rec.ops[0]
    assert [(a, b - a) for a, b, _ in o0.reads] == [(x.data_ptr(), 512), (idx.data_ptr(), 16)]
    assert [(a, b - a) for a, b, _ in o0.writes] == [(y.data_ptr(), 256)]
    races = rec.races()
    assert [(r["a"].idx, r["b"].idx, r["kind"]) for r in races] == [(0, 1, "RAW")]
    assert "UNORDERED RAW" in rec.format(races)
    # with the event wait (an edge stream 1 -> stream 2) the same sequence is clean
    g2 = _FakeGraph()
    lib2 = _FakeLib(g2)
    rec2 = hbcheck.Recorder(graph_source=g2)
    rec2._begin()
    monkeypatch.setattr(_lib, "_lib", lib2)
    monkeypatch.setattr(_lib, "TRACE", rec2)
    monkeypatch.setattr(_lib, "stream", lambda: ctypes.c_void_p(lib2.cur))
    monkeypatch.setattr(K, "stream", _lib.stream)
    y = K.gather(x, idx)
    lib2.cur, lib2.after = 2, (1,)
    z = K.gather(y, idx)
    rec2._finish()
    assert rec2.races() == [] and rec2.summary()["unresolved_pointers"] == 0
    del z


def test_recorder_reads_chain_programs_and_struct_arguments(monkeypatch):
    g = _FakeGraph()
    lib = _FakeLib(g)
    rec = hbcheck.Recorder(graph_source=g)
    rec._begin()
    monkeypatch.setattr(_lib, "_lib", lib)
    monkeypatch.setattr(_lib, "TRACE", rec)
    monkeypatch.setattr(_lib, "stream", lambda: ctypes.c_void_p(1))
    monkeypatch.setattr(K, "stream", _lib.stream)
    monkeypatch.setattr(K, "require_device", lambda *t: None)

    class T:   # a "device" tensor for `_mat`
        pass
    monkeypatch.setattr(K, "_mat", lambda t, cols=None: _lib.addr(t))
    M = 32
    x, W, pre, out = torch.zeros(M, 128), torch.zeros(128, 128), torch.zeros(M, 128), torch.zeros(M, 128)
    rows = torch.arange(M, dtype=torch.int32)
    prog = K.ChainProgram(M)
    prog.load(0, x, rows=rows)
    prog.gemm(W, 0, 1, act=True, pre_out=pre, out=out)
    K.chain(prog, mode="f32")
    rec._finish()
    (op,) = rec.ops
    assert op.name == "gn_chain_f32"
    assert sorted(a for a, _, _ in op.reads) == sorted([x.data_ptr(), rows.data_ptr(), W.data_ptr()])
    assert sorted(a for a, _, _ in op.writes) == sorted([pre.data_ptr(), out.data_ptr()])
    assert not op.unresolved
