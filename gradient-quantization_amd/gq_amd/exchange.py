"""The exchange step of the parameter-server aggregate: every rank's compressed wire to every rank.

The reference has no exchange at all -- its `num_users` live in one process and `apply()` stacks their
DECODED tensors (quantizers/ps_quantizer.py:44-48).  Here each rank owns `users` rows of ONE buffer

    gathered  uint8 [world * users, user_bytes]        (row = rank * users + user)

and its kernels write the wire straight into those rows, so there is no staging copy on either side
of the exchange.  Three ways to fill the other ranks' rows, all bit-identical in their result:

    "allgather"  one in-place all_gather_into_tensor (RCCL picks ring / tree).
    "direct"     all-pairs: one grouped isend / irecv per peer (RCCL: ncclGroupStart .. ncclSend/ncclRecv
                 .. ncclGroupEnd).  xGMI on an MI355X node is point-to-point -- 7 links of ~153 GB/s per
                 GPU -- so every peer's row can travel over its own link at the same time, whereas a ring
                 pushes world-1 rows through one link one after the other.
    "split"      "direct" in two byte ranges [0, cut) and [cut, user_bytes): both are queued at once, the
                 caller decodes the first range while the second is still in flight (needs one user per
                 rank so that a row's byte range is contiguous; otherwise it degrades to "direct").
    "pipelined"  "split" with any number of byte ranges (`cuts`, ascending): chunk k is decoded while the chunks
                 behind it are in flight.  WireExchange.send_range queues one range on its own, so that a caller
                 whose compress launches fill the wire range by range (bench.py's flat tensor: the codes are
                 final one launch before the levels) can start a range's transfer under its next launch.
                 Opt-in only: never part of "auto".

`GQ_EXCHANGE` selects the mode; the default is "allgather", the one collective every backend has ("auto" is
opt-in: time all of them on the first exchange and keep the fastest, every rank taking the same decision from the
all-reduced maxima -- "direct" and "split" have not run on RCCL with more than one rank yet).  Collectives run on the process
group's own stream; `Work.wait()` makes torch's current stream wait for them without blocking the host.
"""
import os
import time

import torch

MODES = ("allgather", "direct", "split")       # what "auto" times
OPT_IN_MODES = ("pipelined",)


def configured_mode(default="allgather"):
    mode = os.environ.get("GQ_EXCHANGE", default)
    if mode not in MODES + OPT_IN_MODES + ("auto",):
        raise ValueError("GQ_EXCHANGE must be one of %s or 'auto', got %r" % (", ".join(MODES + OPT_IN_MODES), mode))
    return mode


class _Pending(object):
    """Outstanding transfers of one byte range; wait() orders the current stream behind them."""

    def __init__(self, works, after=(), lo=0, hi=None):
        self.works = works
        self.after = list(after)      # (device view, host buffer) pairs to copy once the transfers are done
        self.lo, self.hi = lo, hi     # the byte range of a row these transfers fill (hi None: to the end)

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        for dst, src in self.after:
            dst.copy_(src)      # staged (gloo test) path only; blocking, so that the host buffer can be reused
        self.after = []


class WireExchange(object):
    """Owns the [world * users, user_bytes] buffer of one quantizer / bench and moves the rows."""

    def __init__(self, world, rank, users, user_bytes, device, group=None):
        self.world, self.rank, self.users, self.user_bytes = world, rank, users, user_bytes
        self.group = group
        self.gathered = torch.zeros((world * users, user_bytes), dtype=torch.uint8, device=device)
        self.timings_ms = None          # filled by autotune()
        # Point-to-point messages between device buffers are RCCL's business.  Under the gloo TEST backend (several
        # ranks on one GPU: tests, GQ_BENCH_BACKEND=gloo) they are staged through pinned host memory: gloo would
        # write the device buffer from the CPU, unordered with the kernels that read it.
        self._staged = False
        if world > 1 and device.type == "cuda":
            import torch.distributed as dist
            self._staged = dist.get_backend(group) != "nccl"
        self._host = {}

    # ---- views ------------------------------------------------------------------------------
    @property
    def local(self):
        """This rank's own rows: the kernels write the wire here."""
        return self.gathered[self.rank * self.users:(self.rank + 1) * self.users]

    def _peer(self, group_rank):
        import torch.distributed as dist
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)

    # ---- the three transports ---------------------------------------------------------------
    def _allgather(self, rows, dry=False):
        import torch.distributed as dist
        n = rows * self.user_bytes
        flat = self.gathered.view(-1)
        if rows == self.users:
            out = flat
            inp = flat[self.rank * n:(self.rank + 1) * n]
        else:   # fewer records than slots this step: gather the used rows of every rank, rank-major
            out = self._partial_buffer(rows).view(-1)
            inp = self.local[:rows].reshape(-1)
        if dry:
            assert out.numel() == self.world * inp.numel() and out.is_contiguous() and inp.is_contiguous()
            return None
        w = dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)
        return _Pending([w])

    def _partial_buffer(self, rows):
        buf = getattr(self, "_partial", None)
        if buf is None or buf.shape[0] != self.world * rows:
            buf = self._partial = torch.empty((self.world * rows, self.user_bytes), dtype=torch.uint8,
                                              device=self.gathered.device)
        return buf

    def _direct(self, rows, lo, hi, dry=False):
        """Grouped point-to-point transfers of bytes [lo, hi) of the first `rows` rows of every rank.
        dry: build and validate every operation (views, peers, host buffers) but queue nothing."""
        import torch.distributed as dist
        ops, after = [], []
        whole = lo == 0 and hi == self.user_bytes
        target = self.gathered if rows == self.users else self._partial_buffer(rows)
        if rows != self.users and not dry:
            target[self.rank * rows:(self.rank + 1) * rows].copy_(self.local[:rows])
        for step in range(1, self.world):
            # peers in a rotated order: at step s every rank sends to rank+s and receives from rank-s, so
            # that no two ranks aim at the same receiver in the same position of their groups
            dst = (self.rank + step) % self.world
            src = (self.rank - step) % self.world
            mine = target[self.rank * rows:(self.rank + 1) * rows]
            theirs = target[src * rows:(src + 1) * rows]
            if whole:
                send_t, recv_t = mine.view(-1), theirs.view(-1)
            else:
                assert rows == 1, "a byte range of several rows is not contiguous"
                send_t, recv_t = mine[0, lo:hi], theirs[0, lo:hi]
            if self._staged:
                if step == 1:     # one host copy of the outgoing bytes serves every peer
                    out_h = self._host_buffer(("out", lo, hi), send_t.numel())
                    if not dry:
                        out_h.copy_(send_t)       # synchronous: the kernels that wrote the wire have finished
                in_h = self._host_buffer(("in", src, lo, hi), recv_t.numel())
                after.append((recv_t, in_h))
                send_t, recv_t = out_h, in_h
            ops.append(dist.P2POp(dist.isend, send_t, self._peer(dst), group=self.group))
            ops.append(dist.P2POp(dist.irecv, recv_t, self._peer(src), group=self.group))
        if dry:
            return None
        return _Pending(dist.batch_isend_irecv(ops) if ops else [], after, lo, hi)

    def send_range(self, lo, hi):
        """Queue bytes [lo, hi) of this rank's (single) row to every peer and the peers' into `gathered`, behind the
        work queued on the current stream so far -> the pending transfer."""
        assert self.users == 1 and 0 <= lo < hi <= self.user_bytes
        return self._direct(1, lo, hi)

    def _host_buffer(self, key, n):
        buf = self._host.get(key)
        if buf is None or buf.numel() != n:
            buf = self._host[key] = torch.empty(n, dtype=torch.uint8).pin_memory()
        return buf

    # ---- public -----------------------------------------------------------------------------
    def start(self, mode, rows=None, cut=None, dry_run=False, cuts=None):
        """Queue the exchange of the first `rows` rows per rank.  Returns (buffer, [pending...]): one pending
        transfer for "allgather" / "direct", two for "split" (bytes [0, cut) then [cut, user_bytes)), len(cuts) + 1
        for "pipelined" (the ranges between consecutive `cuts`; each pending carries its .lo / .hi).
        dry_run: everything but the transfers themselves -- the buffers, views and operation lists are built and
        checked, nothing is queued and no peer is engaged (autotune's preflight); returns (buffer, [])."""
        rows = self.users if rows is None else rows
        buf = self.gathered if rows == self.users else self._partial_buffer(rows)
        if self.world == 1:
            return (self.gathered[:rows], [])
        if mode == "split" and (rows != 1 or not cut or cut <= 0 or cut >= self.user_bytes):
            mode = "direct"
        if mode == "pipelined":
            cuts = sorted(set(c for c in (cuts or ()) if 0 < c < self.user_bytes))
            if rows != 1 or not cuts:
                mode = "direct"
        if mode == "allgather":
            pend = [self._allgather(rows, dry_run)]
        elif mode == "direct":
            pend = [self._direct(rows, 0, self.user_bytes, dry_run)]
        elif mode == "split":
            pend = [self._direct(rows, 0, cut, dry_run), self._direct(rows, cut, self.user_bytes, dry_run)]
        elif mode == "pipelined":
            edges = [0] + cuts + [self.user_bytes]
            pend = [self._direct(rows, a, b, dry_run) for a, b in zip(edges[:-1], edges[1:])]
        else:
            raise ValueError(mode)
        return buf, ([] if dry_run else pend)

    def run(self, mode, rows=None):
        """Exchange and wait (stream-ordered): the whole gathered buffer is valid for kernels queued next."""
        buf, pending = self.start(mode, rows, cut=self.user_bytes // 2 // 16 * 16)
        for p in pending:
            p.wait()
        return buf

    def autotune(self, step_fn, rounds=10, preflight=None):
        """Time step_fn(mode) -- the caller's exchange (+ decode) for that transport; it is idempotent -- for
        every mode and return the fastest.  Times are maxima over ranks: every rank returns the same mode.

        Every rank issues the SAME sequence of collectives whatever happens locally.  Per mode:
          1. preflight(mode) -- by default start(mode, dry_run=True): the transport's buffers, views and operation
             lists are built and checked, NOTHING is queued and no peer is engaged.  A rank on which this raises
             (a transport its backend or its arguments refuse) reports ok = 0.
          2. ONE all-reduce(MIN) of the ok-flags, outside any try, on every rank.  If any rank failed, the transport
             is dropped on ALL ranks: nobody calls step_fn for it, nobody enters its barrier.
          3. otherwise, on every rank: two untimed calls, barrier, the timed loop.
        An error in step 3 is NOT caught: once a rank has queued its half of a transfer its peers are waiting in
        theirs, an RCCL failure is asynchronous and poisons the communicator -- the exception propagates, the process
        exits non-zero and the launcher ends the job.  (Round 2 caught it per rank and went on to an all-reduce
        while the healthy ranks sat in a barrier: mismatched collectives, a hang.)"""
        import sys
        import torch.distributed as dist
        if self.world == 1:
            return "allgather"
        dev = self.gathered.device
        cuda = dev.type == "cuda"
        if preflight is None:
            def preflight(mode):
                self.start(mode, cut=self.user_bytes // 2 // 16 * 16, dry_run=True)
        res = {}
        for mode in MODES:
            ok = 1.0
            try:
                preflight(mode)
            except Exception as e:
                print("gq_amd.exchange: rank %d: transport %r failed its preflight and is dropped on every rank: %s"
                      % (self.rank, mode, e), file=sys.stderr)
                ok = 0.0
            flag = torch.tensor([ok], dtype=torch.float64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if float(flag.item()) == 0.0:
                res[mode] = 1e30
                continue
            for _ in range(2):
                step_fn(mode)
            if cuda:
                torch.cuda.synchronize()
            dist.barrier(group=self.group)
            t0 = time.perf_counter()
            for _ in range(rounds):
                step_fn(mode)
            if cuda:
                torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t0) / rounds * 1e3
        t = torch.tensor([res[m] for m in MODES], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        self.timings_ms = {m: float(v) for m, v in zip(MODES, t.tolist())}
        return min(MODES, key=lambda m: (self.timings_ms[m], MODES.index(m)))
