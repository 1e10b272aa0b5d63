"""gq_amd.exchange.WireExchange on its NON-STAGED code path -- the one RCCL takes on a multi-GPU node, which no box of the
build pool has run (one GPU each; over gloo the exchange stages through host buffers: other code) -- driven by a fake
`torch.distributed` that emulates N ranks inside this process and records every call.  What is checked, for every
transport and 2 ... 8 ranks:

  * a rank only ever SENDS bytes of its own rows and only ever RECEIVES into its peers' rows of its own buffer -- by
    storage offsets, so an off-by-one row or byte range cannot hide;
  * the in-place all-gather's input IS this rank's slice of its output (no staging copy);
  * nothing is readable before `wait()`: the fake delivers a transfer's bytes when its Work is waited for, never at
    queueing time, and the buffers end up equal to the expected [world * users, user_bytes] matrix on every rank;
  * PSQuantizer.apply() calls wait() on every pending transfer BEFORE the first decode touches the gathered buffer.

No GPU, no process group: the call order and the addresses are what an 8-GPU run will hit first."""
import os
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "gradient-quantization_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


class FakeWork(object):
    def __init__(self, world, deliver):
        self.world, self.deliver, self.waited = world, deliver, False

    def wait(self):
        if not self.waited:
            self.waited = True
            self.world.log.append(("wait", self.world.current))
            self.deliver()
        return True


class FakeP2POp(object):
    def __init__(self, op, tensor, peer, group=None):
        self.op, self.tensor, self.peer, self.group = op, tensor, peer, group


class FakeWorld(object):
    """`torch.distributed` for N ranks in one process.  `current` is the rank whose code is running; sends are parked until
    the matching receive's Work is waited for (delivery at wait time: reading a row early reads zeros)."""

    def __init__(self, n):
        self.n, self.current, self.log = n, 0, []
        self.sent = {}           # (src, dst) -> list of byte tensors (clones taken when the RECEIVER waits: the sender's kernels are done by then)
        self.exchanges = {}      # rank -> its WireExchange

    # -- the API surface exchange.py uses --
    def get_backend(self, group=None):
        return "nccl"

    def get_global_rank(self, group, r):
        return r

    def isend(self, *a, **k):
        raise AssertionError("isend is only passed to P2POp")

    def irecv(self, *a, **k):
        raise AssertionError("irecv is only passed to P2POp")

    P2POp = FakeP2POp

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        assert async_op, "the exchange must not block the host"
        rank = self.current
        self.log.append(("all_gather", rank, out.data_ptr(), out.numel(), inp.data_ptr(), inp.numel()))
        assert out.is_contiguous() and inp.is_contiguous() and out.numel() == self.n * inp.numel()
        n = inp.numel()

        def deliver():
            for r in range(self.n):
                if r != rank:
                    ex = self.exchanges[r]
                    rows = n // ex.user_bytes
                    out[r * n:(r + 1) * n].copy_(ex.local[:rows].reshape(-1))
                elif out[r * n:(r + 1) * n].data_ptr() != inp.data_ptr():      # not in place: the rank's own slice arrives too
                    out[r * n:(r + 1) * n].copy_(inp)
        return FakeWork(self, deliver)

    def batch_isend_irecv(self, ops):
        rank = self.current
        works = []
        for op in ops:
            kind = "send" if op.op == self.isend else "recv"
            self.log.append((kind, rank, op.peer, op.tensor.data_ptr(), op.tensor.numel()))
            if kind == "send":
                self.sent.setdefault((rank, op.peer), []).append(op.tensor)
                works.append(FakeWork(self, lambda: None))
            else:
                src, dst_t = op.peer, op.tensor

                def deliver(src=src, dst_t=dst_t, rank=rank):
                    q = self.sent[(src, rank)]
                    t = [x for x in q if x.numel() == dst_t.numel()][0]
                    q.remove(t)
                    dst_t.copy_(t)
                works.append(FakeWork(self, deliver))
        return works


@pytest.fixture
def fake_dist(monkeypatch):
    import torch.distributed as dist
    holder = {}

    def install(n):
        w = FakeWorld(n)
        w.isend, w.irecv = dist.isend, dist.irecv          # identity objects exchange.py hands to P2POp
        for name in ("get_backend", "get_global_rank", "all_gather_into_tensor", "batch_isend_irecv"):
            monkeypatch.setattr(dist, name, getattr(w, name))
        monkeypatch.setattr(dist, "P2POp", FakeP2POp)
        holder["w"] = w
        return w
    return install


def _fill(ex, rank, users, user_bytes):
    rng = np.random.RandomState(100 + rank)
    ex.local.copy_(torch.from_numpy(rng.randint(0, 256, (users, user_bytes)).astype(np.uint8)))


def _expected(world, users, user_bytes, rows):
    out = np.zeros((world * rows, user_bytes), np.uint8)
    for r in range(world):
        full = np.random.RandomState(100 + r).randint(0, 256, (users, user_bytes)).astype(np.uint8)
        out[r * rows:(r + 1) * rows] = full[:rows]
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("mode,users,rows", [("allgather", 1, 1), ("allgather", 3, 3), ("allgather", 3, 2), ("direct", 1, 1), ("direct", 2, 2),
                                             ("direct", 3, 1), ("split", 1, 1), ("pipelined", 1, 1)])
def test_every_transport_moves_exactly_the_rows_it_owns(fake_dist, world, mode, users, rows):
    from gq_amd import exchange
    w = fake_dist(world)
    user_bytes = 4096 + 64
    dev = torch.device("cpu")
    exs = []
    for r in range(world):
        w.current = r
        ex = exchange.WireExchange(world, r, users, user_bytes, dev)
        assert not ex._staged                                  # the RCCL path: device buffers go to the collective as they are
        _fill(ex, r, users, user_bytes)
        w.exchanges[r] = ex
        exs.append(ex)
    pend, bufs = {}, {}
    for r in range(world):                                     # every rank QUEUES its transfers ...
        w.current = r
        bufs[r], pend[r] = exs[r].start(mode, rows, cut=2048, cuts=[1024, 2048, 3072])
        assert len(pend[r]) == {"allgather": 1, "direct": 1, "split": 2, "pipelined": 4}[mode]
    want = _expected(world, users, user_bytes, rows)
    for r in range(world):                                     # ... and nothing has arrived before a wait()
        got = bufs[r].numpy()                                  # (rows < users: a compact buffer that starts uninitialised)
        for peer in range(world):
            if peer != r and rows == users:
                assert not got[peer * rows:(peer + 1) * rows].any(), "rank %d sees rank %d's rows before wait()" % (r, peer)
    for r in range(world):
        w.current = r
        for p in pend[r]:
            p.wait()
    for r in range(world):
        assert np.array_equal(bufs[r].numpy(), want), "rank %d" % r
    # addresses: sends leave the rank's own rows, receives land in the sender's rows, byte ranges as asked
    for r in range(world):
        buf = bufs[r]
        base, row_b = buf.data_ptr(), buf.stride(0)
        own = (base + r * rows * row_b, base + (r + 1) * rows * row_b)
        for ev in w.log:
            if ev[0] == "send" and ev[1] == r:
                _, _, peer, ptr, n = ev
                assert own[0] <= ptr and ptr + n <= own[1], "rank %d sends bytes outside its rows" % r
            if ev[0] == "recv" and ev[1] == r:
                _, _, peer, ptr, n = ev
                lo, hi = base + peer * rows * row_b, base + (peer + 1) * rows * row_b
                assert lo <= ptr and ptr + n <= hi, "rank %d receives rank %d's bytes outside that rank's rows" % (r, peer)
            if ev[0] == "all_gather" and ev[1] == r:
                _, _, optr, on, iptr, inn = ev
                if rows == users:      # in place: the input is this rank's slice of the output
                    assert optr == base and iptr == own[0] and inn == rows * user_bytes
                else:                  # fewer records than slots: a compact buffer, the rank's rows copied into ... nothing: gathered from `local`
                    assert on == world * inn
    if mode in ("split", "pipelined"):
        edges = [0, 2048, user_bytes] if mode == "split" else [0, 1024, 2048, 3072, user_bytes]
        sizes = sorted(set(ev[4] for ev in w.log if ev[0] == "send"))
        assert sizes == sorted(set(b - a for a, b in zip(edges[:-1], edges[1:])))
    sends = [ev for ev in w.log if ev[0] == "send"]
    if mode != "allgather":
        per = {"direct": 1, "split": 2, "pipelined": 4}[mode]
        assert len(sends) == world * (world - 1) * per         # one message per peer and byte range, no more
        for r in range(world):                                 # rotated peer order: at step s rank r sends to r + s
            peers = [ev[2] for ev in sends if ev[1] == r][:world - 1]
            assert peers == [(r + s) % world for s in range(1, world)]


def test_apply_waits_for_the_transfers_before_it_decodes(fake_dist, monkeypatch):
    """PSQuantizer.apply() on two fake ranks (CPU, oracle codecs): the exchange is queued, every pending transfer is waited
    for, and only then the decode reads the gathered buffer -- in that order on both ranks, for every transport; the result
    equals the two-user single process."""
    from oracle_codec import oracle_codec_factory
    import torch.distributed as dist
    from gq_amd import quantizers
    from gq_amd.compressors import NearestNeighborCompressor
    shapes = [(64, 32), (10,), (48, 64), (7,)]

    def make(users):
        params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
        args = Namespace(c_dim=16, k_bit=8, n_bit=6, no_cuda=True, random=0, ef=False, two_phase=False, scale="exp", num_users=users,
                         mode="ps", cr=256)
        return quantizers.Quantizer(NearestNeighborCompressor, params, args, codec_factory=oracle_codec_factory), params

    def grads(user):
        g = torch.Generator().manual_seed(40 + user)
        return [torch.randn(s, generator=g) * 1e-2 for s in shapes]
    qs, ps = make(2)
    for u in range(2):
        for p, x in zip(ps, grads(u)):
            p.grad = x.clone()
        qs.record(u, epoch=1)
    qs.apply()
    want = [p.grad.data.clone() for p in ps]
    for mode in ("allgather", "direct", "split", "pipelined"):
        w = fake_dist(2)
        monkeypatch.setenv("GQ_EXCHANGE", mode)
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "is_available", lambda: True)
        monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
        monkeypatch.setattr(dist, "get_rank", lambda group=None: w.current)
        ranks = []
        for r in range(2):
            w.current = r
            q, params = make(1)
            for p, x in zip(params, grads(r)):
                p.grad = x.clone()
            q.record(0, epoch=1)
            w.exchanges[r] = q._ex
            ranks.append((q, params))
        # the decode of either rank must come after ITS waits: wrap the codecs' decode to log it
        for r, (q, params) in enumerate(ranks):
            for c in q.codecs:
                orig = c.decode_mean

                def logged(*a, _orig=orig, _r=r, **k):
                    w.log.append(("decode", _r))
                    return _orig(*a, **k)
                c.decode_mean = logged
        # rank 1 queues first (its sends are parked), then rank 0 runs its whole apply, then rank 1 finishes
        outs = {}
        for r in (0, 1):
            w.current = r
            q, params = ranks[r]
            pre = len(w.log)
            if r == 0:      # make rank 1's rows available to rank 0's waits: rank 1 queues its exchange now
                w.current = 1
                _, pend1 = ranks[1][0]._ex.start(mode, 1, ranks[1][0].cut, cuts=ranks[1][0].cuts)
                w.current = 0
            if r == 1:
                monkeypatch.setattr(q._ex, "start", lambda *a, **k: (q._ex.gathered, pend1))     # (already queued above)
            q.apply()
            events = [e for e in w.log[pre:] if e[0] in ("wait", "decode") and e[1] == r]
            first_decode = [i for i, e in enumerate(events) if e[0] == "decode"]
            waits = [i for i, e in enumerate(events) if e[0] == "wait"]
            assert first_decode and waits and waits[0] < first_decode[0], (mode, r, events[:6])
            if mode in ("allgather", "direct"):
                assert max(waits) < first_decode[0]             # one range: everything has arrived before anything is decoded
            outs[r] = [p.grad.data.clone() for p in params]
        for r in (0, 1):
            for a, b in zip(outs[r], want):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (mode, r)
