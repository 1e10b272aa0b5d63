"""Quantizer layer with the reference's contract:

    q = Quantizer(Compressor, model.parameters(), args)      # args.mode in {'ps','ring'}
    for user in ...: loss.backward(); q.record(user, epoch=epoch)
    q.apply(); optimizer.step()

PSQuantizer mirrors quantizers/ps_quantizer.py:6-65, re-designed around a real wire:

* `record` compresses every gradient straight into this user's slot of ONE wire buffer
  (codes | levels | lb,ub per tensor; raw f32 for the <=1000-element tensors).  Nothing is
  decoded unless error feedback needs the residual.
* `apply` (alias `aggregate`) all-gathers the wire buffers of all ranks with ONE RCCL
  collective when torch.distributed is initialised (world_size > 1: every rank is one or
  more of the reference's `num_users`), then runs decode+mean on the GPU: payloads are
  summed in (rank, user) order and divided by their count -- the same arithmetic as the
  reference's torch.stack(decoded).mean(0) -- and `param.grad.data` is rebound to it.

The single-process case (no process group) is the reference's simulated-users loop with
identical results; the wire simply never leaves the GPU.

Codecs (gq_amd/codecs.py: per-tensor and multi-tensor forms of every compressor's wire format) are
the only objects that touch device memory; the product codecs call the HIP library (gq_amd.native).
`codec_factory` exists so that the host logic here can be exercised without a GPU by the tests
(with the CPU oracle as the checker codec).
"""
import math
import operator
import os

import torch

from . import exchange, native
from .compressors import IdenticalCompressor, _next_seed, shared_seeds

# The C++ walks of the parameter list (csrc/host_ext.cpp -> gq_amd/_gq_host.so, built by build.py): the grads, their
# addresses as one bytes key and the "all plain f32" flag in one pass; `.data =` for all parameters in another.  A step
# whose launches replay from graphs is host-bound, and these walks are most of the host's work.  GQ_HOST_EXT=0: the
# Python walks (same results; tests/test_host_logic.py compares the two).
_HOST = None
if os.environ.get("GQ_HOST_EXT", "1") != "0":
    try:
        from . import _gq_host as _HOST
    except ImportError as _e:      # not built: the Python walks below do the same, slower; say so once
        import warnings
        warnings.warn("gq_amd: the host helper _gq_host.so is not built (%s); run gradient-quantization_amd/build.py" % (_e,))


from .codecs import (  # noqa: F401  (re-exported: tests and tools import the codecs from here)
    BatchedHSQ, BatchedQSGD, DenseCodec, GenericCodec, HSQCodec, QSGDCodec, _BatchedBase, _DATA_PTR,
    _DTYPE_OF, _F32_ONLY, _GET_DEVICE, _IS_CONTIGUOUS, _esize, _kernel_copy, _up, aggregate_fma, default_codec_factory,
    wire_levels_mode)


# --------------------------------------------------------------------------------------
# Parameter-server quantizer
# --------------------------------------------------------------------------------------
def _ef_scale(args, epoch):
    # ps_quantizer.py:28-31
    if args.scale == 'exp':
        return 2 / (math.exp(-epoch) + 1) - 1
    return float(args.scale)


def _dist_world(process_group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(process_group), dist.get_rank(process_group)
    return 1, 0


class _capturing(object):
    """`with _capturing(graph): ...` = torch.cuda.graph(graph, capture_error_mode="thread_local") with the cyclic garbage collector
    held off for the duration.  torch.cuda.graph collects once BEFORE a capture begins because freeing device objects inside one is
    not allowed; a collection that the interpreter starts by itself in the middle of the capture (an allocation count crossing its
    threshold) can still finalize an older quantizer's graphs, plans and private memory pools there -- round 6 saw the process abort
    in exactly that spot.  (thread_local: other threads -- RCCL's watchdog -- may call into HIP meanwhile.)"""

    def __init__(self, graph):
        self._ctx = torch.cuda.graph(graph, capture_error_mode="thread_local")
        self._gc = False

    def __enter__(self):
        import gc
        self._gc = gc.isenabled()
        gc.disable()
        try:
            return self._ctx.__enter__()
        except BaseException:
            if self._gc:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self._ctx.__exit__(*exc)
        finally:
            if self._gc:
                gc.enable()


class PSQuantizer(object):
    def __init__(self, Compressor, parameters, args, process_group=None, codec_factory=None):
        self.parameters = list(parameters)
        self.num_layers = len(self.parameters)
        self.args = args
        self.error_feedback = args.ef
        self.two_phase = args.two_phase
        self.process_group = process_group
        factory = codec_factory or default_codec_factory
        self.aggregate_fma = aggregate_fma(args)     # opt-in fused accumulation of the decode-mean (R >= 2 only)
        self.wire_levels = wire_levels_mode(args, _dist_world(process_group)[0])      # "bytes" | "packed6" (6-bit levels where the configuration allows)
        if self.wire_levels == "packed6":
            base_factory = factory
            factory = lambda comp, n, shape: base_factory(comp, n, shape, packed6=True)
        self.compressors = []
        self.codecs = []
        for param in self.parameters:
            param_size = param.flatten().shape[0]
            comp = Compressor(param_size, param.shape, args) if param_size > 1000 else IdenticalCompressor()
            self.compressors.append(comp)
            self.codecs.append(factory(comp, param_size, param.shape))
            if self.aggregate_fma and isinstance(self.codecs[-1], HSQCodec):
                self.codecs[-1].fma = True
            if self.error_feedback:
                param.error = [torch.zeros_like(param) for _ in range(args.num_users)]
            if self.error_feedback and self.two_phase:
                param.server_error = torch.zeros_like(param)
        # wire layout of one user: 16-byte aligned sections for the compressed tensors first, then ONE
        # packed region with the raw f32 of all identity-compressed (<= 1000 element) tensors
        self.offsets = [0] * self.num_layers
        off = 0
        self.dense_idx = [i for i, c in enumerate(self.codecs) if type(c) is DenseCodec]
        for i, c in enumerate(self.codecs):
            if i not in self.dense_idx:
                self.offsets[i] = off
                off = _up(off + c.nbytes)
        self.dense_off = off
        for i in self.dense_idx:
            self.offsets[i] = off
            off += self.codecs[i].nbytes
        self.dense_bytes = off - self.dense_off
        self._dense_mean, self._dense_views, self._dense_turn = [None, None], [None, None], 0
        self._dense_in = {}                 # (wire pointer, slot) -> views of the slot's dense region
        self.user_bytes = _up(off)          # one user's payload (all tensors)
        # tensors served by multi-tensor kernels: (class, parameter indices), built at the first record()
        self._groups = []
        g = getattr(args, "gq_graph", None)
        self.use_graphs = bool(int(os.environ.get("GQ_GRAPH", "1"))) if g is None else bool(g)      # (see below: gq_graph)
        self._fuse_steps = type(self) is PSQuantizer and os.environ.get("GQ_FUSE_STEP", "1") != "0"   # (see below: _step_graphs)
        BatchedQSGD.place_lone_buckets(self.codecs)
        for cls in (BatchedHSQ, BatchedQSGD):
            keyed = {}
            for i, c in enumerate(self.codecs):
                if cls.eligible(c):
                    keyed.setdefault(cls.group_key(c), []).append(i)
            for key in sorted(keyed):
                idx = keyed[key]
                if len(idx) >= 2 and not getattr(args, "gq_no_batch", False):
                    for part in self._overlap_chunks(idx, args):
                        self._groups.append([cls, part, None])
        self.batch_idx = [i for g in self._groups for i in g[1]]
        self._pick_dense = operator.itemgetter(*self.dense_idx) if len(self.dense_idx) >= 2 else None
        self._pick_group = {}
        # gq_graph (args.gq_graph or $GQ_GRAPH=1): the device work of a record() for a set of gradient addresses seen before
        # is replayed as ONE HIP graph launch (the step is a launch-bound loop: ten launches and copies against ~85 us of
        # kernels).  Needs launches whose arguments do not change between records: deterministic rounding or gq_rng="keyed".
        # gq_rng = "device" (the default): the multi-tensor launches draw from streams keyed by { seed, step } pairs in device
        # memory, one pair per (tensor group, user slot) (GQ_RANDOM_DEVICE_COUNTER); every aggregate adds one to the step
        # words.  The launches' arguments never change -- they replay from a HIP graph -- and the draws are fresh every
        # step whatever the gradients are (the reference draws per call: probabilistic_scalar_compressor.py:22-26).
        self._rng_state = None
        self._ticket = None          # gq_hsq_levels_decode_batched's last-workgroup counter (one device word, zero between launches)
        self._rec_graphs = {}        # (slot, user, scale, gradient addresses) -> [sightings, graph or None, keep-alive]
        self._apply_graphs = {}      # (users recorded, wire, output-buffer turns) -> [sightings, graph or None, decoded list]
        # One rank, one user per step (args.num_users == 1, no process group): record() is always followed by the apply() of
        # exactly that payload, so the two replay as ONE graph from record() -- compress and decode-mean launches back to
        # back, one graph launch less per step (5 us of ~70, tools/graph_pieces.py) -- and apply() only rebinds the gradients.
        # $GQ_FUSE_STEP=0 keeps the two graphs.
        self._step_graphs = {}       # (record key, apply key) -> [sightings, graph or None, decoded list]
        self._fused = None           # the decoded list of a step whose record() has already replayed its apply()
        self._side_stream = None     # the second branch of a chunked step's graph (_overlap_fractions)
        self._phase2_base = None     # the seed base all ranks share for the replicated second phase (_second_phase_base)
        self._fast_ok = None         # _replay_known_step applies to this quantizer (None: not decided yet)
        self._apply_key_memo, self._dense_gen = {}, 0
        self._fast_misses = 0
        self.record_paths = {"graph": 0, "graph_any_address": 0, "whole_step": 0, "whole_step_any_address": 0, "eager": 0}   # how record() ran, by count
        self._phase2_calls = 0
        # gq_rng = "reference": the reference draws r = torch.rand(M) per compressed tensor, in parameter order, from
        # the CPU generator (probabilistic_scalar_compressor.py:23).  torch.rand is one sequential stream, so ONE
        # torch.rand(sum of M) per record (and one per two-phase apply) gives every tensor the same numbers; the
        # multi-tensor kernels and the per-tensor path both take their slices from it.
        self._draw_off, n = {}, 0
        for i, c in enumerate(self.codecs):
            if isinstance(c, HSQCodec) and c.uses_reference_draws():
                self._draw_off[i] = n
                n += c.M
        self._draw_total = n
        self._draw_host = None
        self._assembled = {}
        self._plan = None
        self.capacity = max(1, int(args.num_users))
        self.recorded = 0                   # record() calls since the last apply()
        self._wire = None
        self._ex = None                     # exchange.WireExchange when torch.distributed is initialised
        self.exchange_mode = exchange.configured_mode()    # $GQ_EXCHANGE: allgather | direct | split | auto
        # split exchange: bytes [0, cut) travel (and are decoded) first; the cut lies on a tensor boundary near the
        # middle of the compressed part of the wire
        bounds = [self.offsets[i] for i in range(self.num_layers) if i not in self.dense_idx]
        later = [o for o in bounds if o >= self.dense_off // 2]
        self.cut = min(later) if later else 0
        # pipelined exchange ($GQ_EXCHANGE=pipelined, opt-in): $GQ_PIPELINE_CHUNKS byte ranges (default 4) of about equal size
        # with their boundaries on tensor boundaries; the identity-compressed tensors ride in the last one
        nchunks = max(2, int(os.environ.get("GQ_PIPELINE_CHUNKS", "4")))
        self.cuts = []
        for k in range(1, nchunks):
            inner = [o for o in bounds if o > 0]
            if inner:
                c = min(inner, key=lambda o: (abs(o - self.dense_off * k // nchunks), o))      # the tensor boundary nearest to k / nchunks
                if c not in self.cuts:
                    self.cuts.append(c)
        self.cuts.sort()

    # ---- overlap: the tensor list in chunks, a chunk's level / decode launch under the next chunk's encode ----------------
    OVERLAP_MIN_ELEMENTS = 4 << 20      # below this a step is launch-bound: one group, as before

    def _overlap_fractions(self, args):
        """The shares of the compressed elements that the chunks of a tensor group take, or None (one group: the default).
        args.gq_overlap / $GQ_OVERLAP: the shares, e.g. "0.58,0.42"; "0" (default): off.
        EXPERIMENT, measured and not the default (profiles/r06_overlap_ab.txt): the compress of a tensor list is VALU-bound, its
        level / decode launch HBM-bound, and a tensor's levels need only THAT tensor's (lb, ub) (ps_quantizer.py:33-44 treats
        the tensors as independent), so with the list in chunks, chunk i's level + decode launch can run on a second stream -- a
        parallel branch of the step's graph -- while chunk i + 1 is encoded.  The kernels do run side by side, but every extra
        encode launch costs ~10 us of fixed time and every edge between two hardware queues 5-9 us: 94 against 68.5 us per
        ResNet-50 step.  Only where a whole step replays as one graph (one rank, one user per step, no second phase)."""
        spec = getattr(args, "gq_overlap", None)
        spec = os.environ.get("GQ_OVERLAP", "0") if spec is None else str(spec)
        if spec in ("0", "off", "") or not self._fuse_steps or not self.use_graphs or self.two_phase:
            return None
        if int(args.num_users) != 1 or _dist_world(self.process_group)[0] != 1:
            return None
        fr = [float(x) for x in spec.split(",")]
        if len(fr) < 2 or min(fr) <= 0:
            return None
        return [f / sum(fr) for f in fr]

    def _overlap_chunks(self, idx, args):
        """`idx` (parameter indices of one tensor group, in wire order) cut at tensor boundaries into runs whose element counts
        are nearest to the configured shares; every run keeps at least two tensors."""
        fr = self._overlap_fractions(args)
        sizes = [self.codecs[i].numel for i in idx]
        total = sum(sizes)
        if fr is None or total < self.OVERLAP_MIN_ELEMENTS or len(idx) < 2 * len(fr):
            return [idx]
        cum, acc = [], 0
        for n in sizes:
            acc += n
            cum.append(acc)
        cuts, target = [], 0.0
        for f in fr[:-1]:
            target += f * total
            lo = (cuts[-1] if cuts else 0) + 2
            cand = [k for k in range(lo, len(idx) - 1) if len(idx) - k >= 2]
            if not cand:
                return [idx]
            cuts.append(min(cand, key=lambda k: abs(cum[k - 1] - target)))
        parts, a = [], 0
        for c in cuts + [len(idx)]:
            parts.append(idx[a:c])
            a = c
        return parts if all(len(p) >= 2 for p in parts) else [idx]

    # ---- buffers -------------------------------------------------------------------------
    def _ensure_wire(self, device, slots):
        """This rank's [capacity, user_bytes] wire.  Under torch.distributed it is this rank's block of rows of
        the exchange buffer (gq_amd.exchange): the kernels write where the collective reads."""
        world, rank = _dist_world(self.process_group)
        if (self._wire is None or self._wire.device != device or self._wire.shape[0] < slots
                or (world > 1) != (self._ex is not None)):
            cap = max(self.capacity, slots)
            old = self._wire
            if world > 1:
                self._ex = exchange.WireExchange(world, rank, cap, self.user_bytes, device, self.process_group)
                new = self._ex.local
            else:
                self._ex = None
                new = torch.zeros((cap, self.user_bytes), dtype=torch.uint8, device=device)
            if old is not None and old.device == device:
                new[:min(cap, old.shape[0])].copy_(old[:cap])
            self._wire = new
            self.capacity = cap
        return self._wire

    def wire_bytes_per_user(self):
        return self.user_bytes

    RNG_SLOTS = 17      # { seed, step } pairs per group: 16 user slots + the two-phase re-compress (the last one)
    TWO_PHASE_RNG_SLOT = RNG_SLOTS - 1

    def _rng_pairs_for(self, device, group_index):
        """This group's rows of the quantizer's { seed, step } array (made on first use; seeds from torch's seed, the
        rank, the group and the slot; steps start at 0).  The two-phase slot's seed comes from the base ALL ranks share
        (_second_phase_base) and leaves the rank out: the second phase runs replicated on every rank and must round identically
        everywhere (ps_quantizer.py:52-61 runs it once, on the server); the step words advance in lockstep on all ranks."""
        if self._rng_state is None or self._rng_state.device != device:
            world, rank = _dist_world(self.process_group)
            n = max(1, len(self._groups)) * self.RNG_SLOTS
            host = torch.zeros((n, 2), dtype=torch.int64)
            base = _next_seed()
            shared = self._second_phase_base(device) if self.two_phase else base
            for i in range(n):
                if i % self.RNG_SLOTS == self.TWO_PHASE_RNG_SLOT:
                    host[i, 0] = ((shared ^ ((i + 1) * 0x9E3779B97F4A7C15)) & (2 ** 63 - 1))
                else:
                    host[i, 0] = ((base ^ ((rank * 1000003 + i + 1) * 0x9E3779B97F4A7C15)) & (2 ** 63 - 1))
            self._rng_state = host.to(device)
        return self._rng_state[group_index * self.RNG_SLOTS:(group_index + 1) * self.RNG_SLOTS]

    def _second_phase_base(self, device):
        """The seed base of the replicated second phase: rank 0's, broadcast ONCE over the process group (a collective: every
        rank reaches it at its first two-phase record / apply).  Ranks that seed torch differently -- per-rank data augmentation
        is the usual reason -- would otherwise round the second phase differently, apply different gradients and carry different
        server residuals, silently."""
        if self._phase2_base is None:
            base = _next_seed()
            world, rank = _dist_world(self.process_group)
            if world > 1:
                import torch.distributed as dist
                on_gpu = device.type == "cuda" and dist.get_backend(self.process_group) != "gloo"
                t = torch.tensor([base], dtype=torch.int64, device=device if on_gpu else "cpu")
                dist.broadcast(t, src=0 if self.process_group is None else dist.get_global_rank(self.process_group, 0),
                               group=self.process_group)
                base = int(t.item())
            self._phase2_base, self._phase2_calls = base, 0
        return self._phase2_base

    def _second_phase_seed(self):
        """The _next_seed() of the second phase (shared_seeds): the shared base and a call count that advances in lockstep on all
        ranks (the second phase makes the same calls everywhere)."""
        self._phase2_calls += 1
        return (self._phase2_base * 0x9E3779B97F4A7C15 + self._phase2_calls * 0xD1B54A32D192ED03) & (2 ** 63 - 1)

    def _draws(self, device):
        """One torch.rand for all reference-parity tensors of this record / two-phase apply -> (device tensor, offsets)."""
        if not self._draw_total:
            return None
        if device.type != "cuda":
            return torch.rand(self._draw_total), self._draw_off
        if self._draw_host is None:
            self._draw_host = [torch.empty(self._draw_total).pin_memory() for _ in range(2)]
            self._draw_turn = 0
            self._draw_events = [None, None]
        k = self._draw_turn
        self._draw_turn ^= 1
        if self._draw_events[k] is not None:
            self._draw_events[k].synchronize()      # the previous copy out of this pinned buffer
        torch.rand(self._draw_total, out=self._draw_host[k])     # straight into pinned memory (a fresh 6 MB tensor per
        dev_r = self._draw_host[k].to(device, non_blocking=True)  # record cost 20 ms of page faults on the GPU box)
        ev = torch.cuda.Event()
        ev.record()
        self._draw_events[k] = ev
        return dev_r, self._draw_off

    # ---- reference protocol -------------------------------------------------------------
    def _replay_known_step(self, user):
        """The short way through record() for a step seen before (a host-bound path: round 6 measured 64 us of host time per
        replayed ResNet-50 step against 49 us of QSGD kernels): when the gradients sit at addresses whose whole-step graph exists,
        replay it without building the list of gradient objects (the C++ helper hands back the addresses as one bytes key).  One
        rank, one user per step, no error feedback (its key holds the residuals' addresses too); everything else takes record()'s
        long way, which ends in the same replays.  -> True when the step has been replayed."""
        if (self._fast_ok is None or not self._fast_ok) and not self._fast_check():
            return False
        if self._fast_misses >= 8:      # a caller whose gradients move every step (a training loop): look again every 32nd record only
            self._fast_misses += 1
            if self._fast_misses & 31:
                return False
        if (self.recorded != 0 or not self.use_graphs or not self._fuse_steps or torch.cuda.is_current_stream_capturing()
                or _dist_world(self.process_group)[0] != 1):      # (use_graphs / _fuse_steps: a failed capture, or a caller, may switch them off)
            return False
        key, ok, index = _HOST.scan_key(self.parameters)
        wire = self._wire
        if not ok or index != wire.device.index or index != torch._C._cuda_getDevice():
            self._fast_misses += 1
            return False
        graph_key = (0, user, wire.data_ptr(), key)
        ent = self._rec_graphs.get(graph_key)
        fent = self._step_graphs.get((graph_key, self._apply_key_for(1, wire.data_ptr()))) if (ent is not None and ent[1] is not None) else None
        if fent is None or fent[1] is None:
            self._fast_misses += 1
            return False
        self._fast_misses = 0
        for g in self._groups:
            g[2].ensure_clean()
        fent[1].replay()
        for g in self._groups:
            g[2]._last_ptrs = None
            g[2]._out_turn ^= 1
        if len(self.dense_idx) >= 2:
            self._dense_turn ^= 1
        self._fused = fent[2]
        self.recorded = 1
        self.record_paths["whole_step"] += 1
        return True

    def _fast_check(self):
        """Can _replay_known_step ever apply to this quantizer?  (False: decided for good; None: not yet -- the groups are built by
        the first record.)"""
        if self._fast_ok is False:
            return False
        if (_HOST is None or not hasattr(_HOST, "scan_key") or not self.use_graphs or not self._fuse_steps or self.error_feedback
                or self.two_phase or self._draw_total or self.capacity != 1 or not self._groups):
            self._fast_ok = False
            return False
        if self._wire is None or self._plan is None or self._plan[2] or _dist_world(self.process_group)[0] != 1:
            return False      # (not yet; a process group may still be initialised later: checked again by the long way's own conditions)
        if not all(g[2] is not None and g[2].ready and g[2].graphable() for g in self._groups):
            return False
        self._fast_ok = True
        return True

    def record(self, user, epoch):
        if self._fast_ok is not False and self._replay_known_step(user):
            return
        scale = _ef_scale(self.args, epoch)
        scan = _HOST.scan_grads(self.parameters) if _HOST is not None else None     # (grads, addresses as bytes, all plain f32)
        all_grads = scan[0] if scan is not None else [p.grad for p in self.parameters]     # (p.grad.data builds an alias tensor per access: ~1 us each)
        dev = all_grads[0].device
        slot = self.recorded
        wire = self._ensure_wire(dev, slot + 1)[slot]
        world, rank = _dist_world(self.process_group)
        salt = ((rank * 1000003 + user) * 0x9E3779B1) & (2 ** 62 - 1)
        skip = set()
        draws = self._draws(dev)
        # gq_graph: a record whose gradient addresses were seen before replays its device work as ONE graph launch
        graph_key = None
        if (self.use_graphs and dev.type == "cuda" and not self._draw_total and slot < self.TWO_PHASE_RNG_SLOT
                and all(g[2] is not None and g[2].ready and g[2].graphable() for g in self._groups)
                and not torch.cuda.is_current_stream_capturing()):      # (inside a caller's own capture the launches are simply recorded)
            graph_key = (slot, user, self._wire.data_ptr(), scan[1] if scan is not None else tuple(map(_DATA_PTR, all_grads)))
            if self.error_feedback:     # the residual buffers' addresses are in the header too (a per-tensor step replaces them)
                graph_key += (scale, tuple(p.error[user].data_ptr() for p in self.parameters))
            ent = self._rec_graphs.get(graph_key)
            plain_f32 = ent is not None and ent[1] is not None and (
                scan[2] if scan is not None else (all(map(_IS_CONTIGUOUS, all_grads)) and set(map(_DTYPE_OF, all_grads)) == _F32_ONLY))
            step_key = None
            if (plain_f32 and self._fuse_steps and world == 1 and slot == 0 and self.capacity == 1 and not self.two_phase
                    and self._plan is not None and not self._plan[2]):      # (the plan: everything decodes through multi-tensor launches)
                step_key = (graph_key, self._apply_key(self._wire[:1]))
                fent = self._step_graphs.get(step_key)
                if fent is not None and fent[1] is not None:      # compress + decode-mean of this step in one launch
                    for g in self._groups:
                        g[2].ensure_clean()
                    fent[1].replay()
                    for g in self._groups:
                        g[2]._last_ptrs = None
                        g[2]._out_turn ^= 1
                    if len(self.dense_idx) >= 2:
                        self._dense_turn ^= 1
                    self._fused = fent[2]
                    self.recorded += 1
                    self.record_paths["whole_step"] += 1
                    return
            if plain_f32:
                for g in self._groups:
                    g[2].ensure_clean()
                ent[1].replay()
                self.record_paths["graph"] += 1
                for g in self._groups:
                    g[2]._last_ptrs = None      # the device header now holds this graph's table: the next eager call re-sends its own
                self.recorded += 1
                if step_key is not None:
                    fent = self._graph_entry(self._step_graphs, step_key)
                    if fent is not None and fent[0] >= 2 and fent[1] is None:
                        self._capture_step(fent, ent[2], all_grads, wire, slot, user, salt, scale, dev)
                return
            # No graph for THESE gradient addresses (a training loop whose backward allocates the gradients anew sees a new set
            # nearly every step: round 6 measured 98 sets in 150 iterations of driver.one_iter, 48 captures that were hardly
            # ever replayed and eager launches from then on).  The ADDRESS-FREE graph: the same launches reading the shared
            # device header, which is refreshed -- pointers, accumulator resets -- by one pinned copy in front of the replay,
            # exactly what an eager step sends.  One graph per (slot, user) serves every address set.
            generic_key = ("any", slot, user, self._wire.data_ptr()) + ((scale,) if self.error_feedback else ())
            gent = self._rec_graphs.get(generic_key)
            if (gent is not None and gent[1] is not None
                    and (scan[2] if scan is not None else (all(map(_IS_CONTIGUOUS, all_grads)) and set(map(_DTYPE_OF, all_grads)) == _F32_ONLY))
                    and self._upload_headers(all_grads, slot, user)):
                # an address set that has come back gets a graph of its own (no header copy in front of its replays); the capture
                # executes nothing, this step still replays the address-free graph
                ent = self._graph_entry(self._rec_graphs, graph_key, max_captured=self.MAX_ADDRESS_GRAPHS)
                if ent is not None and ent[0] >= 2 and ent[1] is None:
                    self._capture_record(ent, all_grads, wire, slot, user, salt, scale, dev)
                gstep = None
                if (self._fuse_steps and world == 1 and slot == 0 and self.capacity == 1 and not self.two_phase
                        and self._plan is not None and not self._plan[2]):
                    gstep = (generic_key, self._apply_key(self._wire[:1]))
                    fent = self._step_graphs.get(gstep)
                    if fent is not None and fent[1] is not None:
                        fent[1].replay()
                        for g in self._groups:
                            g[2]._out_turn ^= 1
                        if len(self.dense_idx) >= 2:
                            self._dense_turn ^= 1
                        self._fused = fent[2]
                        self.recorded += 1
                        self.record_paths["whole_step_any_address"] += 1
                        return
                gent[1].replay()
                self.recorded += 1
                self.record_paths["graph_any_address"] += 1
                if gstep is not None:
                    fent = self._graph_entry(self._step_graphs, gstep)
                    if fent is not None and fent[0] >= 2 and fent[1] is None:
                        self._capture_step(fent, None, all_grads, wire, slot, user, salt, scale, dev)
                return
        self.record_paths["eager"] += 1
        skip = self._record_launches(all_grads, wire, slot, user, salt, scale, draws, dev)
        if len(skip) == self.num_layers:     # the usual case: everything went through the multi-tensor launches
            self.recorded += 1
            if graph_key is not None:
                # the address-free graph first (it serves every later step); a graph of its own for an address set only when
                # the set has come back (callers whose gradients keep their storage: one copy node less per step)
                generic_key = ("any", slot, user, self._wire.data_ptr()) + ((scale,) if self.error_feedback else ())
                gent = self._graph_entry(self._rec_graphs, generic_key, max_captured=1 << 30)
                if gent is not None and gent[1] is None and gent[0] >= 2 and self._generic_ok():
                    self._capture_record(gent, all_grads, wire, slot, user, salt, scale, dev, generic=True)
                ent = self._graph_entry(self._rec_graphs, graph_key, max_captured=self.MAX_ADDRESS_GRAPHS)
                if ent is not None and ent[0] >= 2 and ent[1] is None:      # the second sighting: worth a capture
                    self._capture_record(ent, all_grads, wire, slot, user, salt, scale, dev)
            return
        for i, param in enumerate(self.parameters):
            if i in skip:
                continue
            codec, off = self.codecs[i], self.offsets[i]
            grad = param.grad.data
            if self.error_feedback:
                # ps_quantizer.py:35-39:  grad += scale*error ; error = grad - decoded
                if grad.device.type == "cuda" and grad.is_contiguous() and grad.dtype == torch.float32:
                    native.axpy_inplace(grad, param.error[user].contiguous(), scale)
                else:
                    grad.add_(scale * param.error[user])
                if hasattr(codec, "encode_decode_into"):
                    decoded = torch.empty(grad.numel(), dtype=torch.float32, device=grad.device)
                    codec.encode_decode_into(grad, wire, off, salt, decoded, **self._slice(draws, i))
                    decoded = decoded.view(param.shape)
                elif hasattr(codec, "decode_wire"):
                    codec.encode_into(grad, wire, off, salt, **self._slice(draws, i))
                    decoded = torch.empty(grad.numel(), dtype=torch.float32, device=grad.device)
                    codec.decode_wire(wire, off, decoded)
                    decoded = decoded.view(param.shape)
                else:      # (round 6: such a codec's payload never reached the wire under error feedback -- zeros were aggregated)
                    raise TypeError("%s has neither encode_decode_into nor decode_wire: under error feedback its payload must be both "
                                    "written to the wire and decoded (ps_quantizer.py:37)" % type(codec).__name__)
                if grad.device.type == "cuda" and grad.is_contiguous():
                    err = torch.empty_like(grad)
                    native.sub(grad, decoded.contiguous(), err)
                    param.error[user].data = err
                else:
                    param.error[user].data = grad - decoded
            else:
                codec.encode_into(grad, wire, off, salt, **self._slice(draws, i))
        self.recorded += 1

    MAX_ADDRESS_GRAPHS = 8      # graphs tied to one set of gradient addresses (the address-free ones are not counted)

    def graph_counts(self):
        """Captured graphs by kind: record / whole step tied to a set of gradient addresses, their address-free forms, apply."""
        def split(cache, pick):
            keys = [k for k, e in cache.items() if e[1] is not None]
            free = sum(1 for k in keys if pick(k)[0] == "any")
            return len(keys) - free, free
        rec, rec_any = split(self._rec_graphs, lambda k: k)
        step, step_any = split(self._step_graphs, lambda k: k[0])
        return {"record": rec, "record_any_address": rec_any, "apply": sum(1 for e in self._apply_graphs.values() if e[1] is not None),
                "whole_step": step, "whole_step_any_address": step_any}

    def _generic_ok(self):
        """An address-free record graph holds kernel launches only: the dense tensors must ride in the first group's launch
        (their own copy is a torch op on the gradients themselves)."""
        return bool(self._groups) and (len(self.dense_idx) < 2 or bool(self._groups[0][2] is not None and self._groups[0][2].ndense))

    def _upload_headers(self, all_grads, slot, user):
        """Every group's header for these gradients to the device (in front of an address-free graph's replay)."""
        for grp in self._groups:
            cls, idxs, obj = grp
            grads = list(self._pick_group[id(grp)](all_grads))
            errs = [self.parameters[i].error[user] for i in idxs] if self.error_feedback else None
            dense = list(self._pick_dense(all_grads)) if obj.ndense else None
            if not obj.upload(grads, slot, errs, dense):
                return False
        return True

    def _make_group(self, grp, dev):
        """The multi-tensor launch object of one group, built the same way whoever needs it first (a record, or a ring rank
        that decodes before it has encoded anything): dense copy table, draws' { seed, step } pairs, aggregate form."""
        cls, idxs = grp[0], grp[1]
        # the first group's compress launch also copies the identity-compressed tensors into the wire
        dense = ([(self.offsets[i], self.codecs[i].numel) for i in self.dense_idx]
                 if (grp is self._groups[0] and len(self.dense_idx) >= 2) else None)
        obj = grp[2] = cls(self.codecs, self.offsets, idxs, dev, self.capacity, self.user_bytes, dense=dense)
        obj.fma = bool(self.aggregate_fma and cls is BatchedHSQ)
        if dev.type == "cuda":
            self._ticket_for(dev, 0)      # (allocated and zeroed HERE, eagerly: a first use under stream capture would put it in a graph's pool)
        if getattr(obj, "counter", False) and len(self._groups) * self.RNG_SLOTS <= 256:
            obj.rng_pairs = self._rng_pairs_for(dev, self._groups.index(grp))
        return obj

    def _ticket_for(self, dev, gi):
        """Group gi's last-workgroup counters (gq_step_tail.ticket: zero between launches; one set per group -- the groups of a
        chunked step run their launches side by side)."""
        if self._ticket is None or self._ticket.device != dev:
            assert not torch.cuda.is_current_stream_capturing(), "the ticket words are made by the first eager record (_make_group)"
            self._ticket = torch.zeros((max(1, len(self._groups)), native.TICKET_WORDS), dtype=torch.int32, device=dev)
        return self._ticket[gi]

    def _record_launches(self, all_grads, wire, slot, user, salt, scale, draws, dev, headers=None, defer_resets=None, fuse_levels=False,
                         overlap=None, table_current=False):
        """The multi-tensor launches of a record (+ the dense tensors' copy into the wire) -> the set of parameters served.
        headers (stream capture): one device-resident header per group, see BatchedHSQ.encode.
        fuse_levels (whole-step capture, _can_fuse_levels): the group's level launch is left to the aggregate's decode.
        overlap (whole-step capture of a chunked list, a dict): an event is recorded behind every group's compress
        (overlap["events"][group]) and its accumulators' reset is kept per group (overlap["resets"][group]): _decode_all puts
        the group's level / decode launch on the side stream behind that event."""
        skip = set()
        skip_groups = []
        for grp in (self._groups if dev.type == "cuda" else []):
            cls, idxs, obj = grp
            if obj is None:
                obj = self._make_group(grp, dev)
            pick = self._pick_group.get(id(grp))      # operator.itemgetter over the group's indices, built once
            if pick is None:
                pick = self._pick_group[id(grp)] = operator.itemgetter(*idxs)     # (a group has at least two tensors)
            grads = list(pick(all_grads))
            # error feedback (ps_quantizer.py:35,39) rides in the same launches: grad += scale*error
            # before the encode, error = grad - decoded after it, both in place
            errs = [self.parameters[i].error[user] for i in idxs] if self.error_feedback else None
            hdr = headers[len(skip_groups)] if headers is not None else None
            skip_groups.append(obj)
            dense = list(self._pick_dense(all_grads)) if obj.ndense else None
            kw = {"skip_levels": True} if fuse_levels else {}
            if table_current:      # (capture of an address-free graph: launches only, the shared device header is current)
                kw["table_current"] = True
            mine = [] if overlap is not None else defer_resets
            if obj.encode(grads, wire, slot, salt, errs, scale, draws=draws, graph_header=hdr, dense=dense, defer_reset=mine, **kw):
                skip.update(idxs)
                if dense is not None:
                    skip.update(self.dense_idx)      # (copied by that launch)
                if overlap is not None:
                    ev = torch.cuda.Event()
                    ev.record()
                    overlap["events"][id(obj)] = ev
                    overlap["resets"][id(obj)] = mine
        if len(self.dense_idx) >= 2 and self.dense_idx[0] not in skip:
            # all small tensors with one concatenation straight into the packed wire region.  Under
            # error feedback their residual is identically zero (decoded == grad), so nothing else to do.
            key = (self._wire.data_ptr(), slot)
            views = self._dense_in.get(key)
            if views is None:       # parameter-shaped views of this slot's dense region, built once
                region = wire[self.dense_off:self.dense_off + self.dense_bytes].view(torch.float32)
                views, o = [], 0
                for i in self.dense_idx:
                    n = self.codecs[i].numel
                    views.append(region[o:o + n].view(self.codecs[i].shape))
                    o += n
                self._dense_in = {k: v for k, v in self._dense_in.items() if k[0] == key[0]}   # drop a replaced wire's
                self._dense_in[key] = views
            with torch.no_grad():
                torch._foreach_copy_(views, list(self._pick_dense(all_grads)))
            skip.update(self.dense_idx)
        return skip

    @staticmethod
    def _new_graph():
        """The graph object of a capture.  $GQ_DIRECT_REPLAY (default 1): kept as a graph (keep_graph) so that _replayable can read its
        kernel nodes back."""
        if os.environ.get("GQ_DIRECT_REPLAY", "1") != "0":
            try:
                return torch.cuda.CUDAGraph(keep_graph=True)
            except TypeError:      # (a torch without keep_graph: the graph's own replay)
                pass
        return torch.cuda.CUDAGraph()

    @staticmethod
    def _replayable(graph):
        """What an entry's replay() runs: the captured kernel nodes as PLAIN launches on the current stream (native.LaunchPlan --
        a replayed graph pays 4-5 us of boundary between two replays that launches on a stream do not: 62.4 against 57.6 us per
        two-kernel ResNet-50 step, tools/direct_vs_graph.py), or the graph itself where it is not one chain of kernel launches
        (two branches, a torch without raw graph access) or $GQ_DIRECT_REPLAY=0."""
        if os.environ.get("GQ_DIRECT_REPLAY", "1") != "0" and hasattr(graph, "raw_cuda_graph"):
            try:
                return native.LaunchPlan(graph)
            except Exception:
                pass
        try:
            graph.instantiate()      # (keep_graph: not instantiated by capture_end)
        except Exception:
            pass
        return graph

    @staticmethod
    def _graph_entry(cache, key, max_captured=48, max_counting=64):
        """[sightings, graph or None, keep-alive] of `key`, its sighting counted.  A few address sets recur (the allocator
        hands the same blocks out again); captured graphs are never evicted -- once max_captured of them exist, new sets keep
        their eager launches (None) instead of displacing one another capture by capture -- and among the entries that
        only count sightings the oldest goes first."""
        ent = cache.get(key)
        if ent is None:
            counting = [k for k, e in cache.items() if e[1] is None]
            if sum(1 for k, e in cache.items() if e[1] is not None and not (type(k) is tuple and k and k[0] == "any")) >= max_captured:
                return None      # (address-free graphs, keys ("any", ...), are not counted)
            if len(counting) >= max_counting:
                cache.pop(counting[0])
            ent = cache[key] = [0, None, None]
        ent[0] += 1
        return ent

    def _capture_record(self, ent, all_grads, wire, slot, user, salt, scale, dev, generic=False):
        """Stream-capture the launches the record just made eagerly, with copies of the headers it has just sent
        (the shared pinned buffers are rewritten by later records, a graph's memcpy node reads its source at every replay).
        generic: the address-free form -- the launches read the SHARED device header (record() refreshes it in front of every
        replay), so the graph holds no pointer of these gradients."""
        try:
            # (device-resident: the graph's copy node is device-to-device.  A host-to-device node -- pinned memory over PCIe --
            # cost 15 us of every replayed step, tools/graph_pieces.py; the 5 KB header per captured graph is nothing)
            headers = None if generic else [g[2]._host[g[2]._last_slot].to(dev) for g in self._groups]
            graph = self._new_graph()
            with _capturing(graph):
                self._record_launches(all_grads, wire, slot, user, salt, scale, None, dev, headers=headers, table_current=generic)
            graph = self._replayable(graph)
        except Exception as e:      # a capture that fails leaves the eager path as it was (this record has already run eagerly)
            self.use_graphs = False
            import warnings
            warnings.warn("gq_graph: capturing a record failed (%s); continuing with eager launches" % (e,))
            return
        ent[1], ent[2] = graph, headers

    def _capture_step(self, fent, headers, all_grads, wire, slot, user, salt, scale, dev):
        """One graph for a whole step: the record's launches (headers: the device copies its own graph keeps) and the
        decode-mean launches the following apply() would make, captured from record() after this record has run.  Nothing
        executes here; the output-buffer turns the capture advances are put back for the apply() that is still to come."""
        after = ([g[2]._out_turn for g in self._groups], self._dense_turn)
        try:
            graph = self._new_graph()
            with _capturing(graph):
                resets = []      # the groups' accumulator resets ride in the step's last launch
                fuse = self._can_fuse_levels()      # one rank, one user: level launch + decode of that payload as ONE launch
                overlap = None
                if len(self._groups) >= 2 and all(g[2].takes_tail for g in self._groups) and os.environ.get("GQ_STEP_TAIL", "1") != "0":
                    # a chunked list (_overlap_fractions): every group's level / decode launch (with the group's OWN tail) on the
                    # side stream behind its compress -- a parallel branch of this graph
                    if self._side_stream is None:
                        self._side_stream = torch.cuda.Stream(device=dev)
                    overlap = {"events": {}, "resets": {}, "side": self._side_stream if os.environ.get("GQ_OVERLAP_STREAMS", "1") != "0" else None}
                self._record_launches(all_grads, wire, slot, user, salt, scale, None, dev, headers=headers, defer_resets=resets,
                                      fuse_levels=fuse, overlap=overlap, table_current=headers is None)
                decoded = self._decode_all(self._wire[:1], False, (), resets=resets, fused_levels=fuse, overlap=overlap)
            fent[1], fent[2] = self._replayable(graph), decoded
        except Exception as e:      # the two-graph replay keeps working
            self._fuse_steps = False
            for g in self._groups:      # a launch that failed between a group's encode and its level launch: back to the shared header
                if g[2] is not None:
                    g[2]._graph_tables_abort()
                    if getattr(g[2], "_pending_levels", None) is not None:
                        g[2]._pending_levels = None
            import warnings
            warnings.warn("gq_graph: capturing a whole step failed (%s); record and apply keep their own graphs" % (e,))
        finally:
            for g, t in zip(self._groups, after[0]):
                g[2]._out_turn = t
            self._dense_turn = after[1]

    def _can_fuse_levels(self):
        """A whole step of one rank and one user whose tensors all go through ONE HSQ group (+ the dense tensors riding in
        its launches): encode, then gq_hsq_levels_decode_batched.  $GQ_FUSE_LEVELS=0 keeps the three launches."""
        if os.environ.get("GQ_FUSE_LEVELS", "1") == "0" or not self._groups:
            return False
        if self.error_feedback:
            # measured twice (profiles/r05_experiments.txt, 2 and 8): with error feedback the one launch moves three streams (updated
            # gradient in, residual and decoded tensor out) and is SLOWER than the level launch + the decode (0.1203 against
            # 0.1175 ms per step); the opt-in that forced it ($GQ_FUSE_LEVELS=ef) went in round 6
            return False
        obj = self._groups[0][2]      # (several groups: the chunks of ONE tensor group, _overlap_chunks; the first carries the dense tensors)
        return (all(isinstance(g[2], BatchedHSQ) and g[2].fusable_levels() for g in self._groups)
                and (not self.dense_idx or (len(self.dense_idx) >= 2 and obj.ndense))
                and os.environ.get("GQ_STEP_TAIL", "1") != "0")

    def _apply_key(self, gathered):
        """What an apply()'s captured launches depend on: payload count, the wire, and which output buffers are next."""
        return self._apply_key_for(gathered.shape[0], gathered.data_ptr())

    def _apply_key_for(self, R, ptr):
        # (the two tuples of buffer addresses change only when a buffer is allocated: remembered per (turns, allocation count))
        turns = tuple([g[2]._out_turn for g in self._groups])
        gen = (turns, self._dense_turn, self._dense_gen, tuple([g[2]._out_gen for g in self._groups]))
        hit = self._apply_key_memo.get(gen)
        if hit is None:
            if len(self._apply_key_memo) > 16:
                self._apply_key_memo.clear()
            hit = self._apply_key_memo[gen] = (tuple(0 if o is None else o.data_ptr() for g in self._groups for o in g[2]._outs),
                                               tuple(0 if m is None else m.data_ptr() for m in self._dense_mean))
        return (R, ptr, turns, self._dense_turn, hit[0], hit[1])

    def _slice(self, draws, i):
        """This parameter's share of the record's draws as a keyword for the codec (nothing for the other codecs)."""
        if draws is None or i not in self._draw_off:
            return {}
        o = self._draw_off[i]
        return {"r": draws[0][o:o + self.codecs[i].M]}

    def _decode_all(self, gathered, two_phase, pending=(), plain=False, resets=None, fused_levels=False, overlap=None, phase2_headers=None):
        """Mean of the R = gathered.shape[0] user payloads for every parameter (ps_quantizer.py:47-61),
        as a list of tensors in parameter order.  `pending`: the transfers that fill `gathered`
        (exchange.WireExchange.start) -- one, or one per byte range for a split / pipelined exchange, in which case the
        tensors of a range are decoded while the ranges behind it are still in flight."""
        R = gathered.shape[0]
        done = {}
        pending = list(pending)
        chunked = len(pending) >= 2      # split / pipelined: byte ranges of the wire, each decoded as soon as it has arrived
        on_gpu = gathered.device.type == "cuda"
        # a group that did not encode in multi-tensor form this run (unaligned tensors, too many of them) is not
        # `ready`: its tensors take the per-tensor decode below
        key = (on_gpu,) + tuple(g[2] is not None and g[2].ready for g in self._groups)
        if self._plan is None or self._plan[0] != key:      # who decodes what: rebuilt only when a group's state changes
            groups = [g for g in (self._groups if on_gpu else []) if g[2] is not None and g[2].ready]
            batched = set(i for g in groups for i in g[1])
            dense = set(self.dense_idx) if len(self.dense_idx) >= 2 else set()
            single = [i for i in range(self.num_layers) if i not in batched and i not in dense]
            self._plan = (key, groups, single, {})
        _, groups, single, seg_ranges = self._plan
        single = list(single)
        group_views = {}

        # The aggregate's small per-step work -- the mean of the identity-compressed tensors' rows, one step of the draws'
        # { seed, step } words, the accumulators' reset (whole-step capture) -- rides in the LAST multi-tensor decode launch
        # that can take it (native.StepTail: gq_hsq_decode_sum_batched_tail): one kernel and one boundary less per step.
        # Not with a chunked exchange (several decode launches per group), not with two-phase (its re-compress draws from
        # the step words and folds into the accumulators AFTER this decode).
        step_rng = self._rng_state is not None and on_gpu     # one step of the device draws per aggregate (GQ_RANDOM_DEVICE_COUNTER)
        dense_job = None
        if len(self.dense_idx) >= 2:
            # identity tensors: two-phase / error feedback leave them unchanged (roundtrip == clone)
            rows = gathered[:, self.dense_off:self.dense_off + self.dense_bytes].view(torch.float32)
            k = self._dense_turn
            self._dense_turn ^= 1
            if self._dense_mean[k] is None or self._dense_mean[k].device != rows.device:
                mean = torch.empty(rows.shape[1], dtype=torch.float32, device=rows.device)
                views, o = [], 0
                for i in self.dense_idx:
                    n = self.codecs[i].numel
                    views.append(mean[o:o + n].view(self.codecs[i].shape))
                    o += n
                self._dense_mean[k], self._dense_views[k] = mean, views
                self._dense_gen += 1
            dense_job = (rows, k)
        tail, tail_group = None, -1
        tails = None
        if overlap is not None:
            # whole-step capture of a chunked list: every group's launch carries its OWN tail -- the step of its own draws' words,
            # the reset of its own accumulators (both behind the group's last reader, whatever the other branch is doing), and the
            # first group's also the mean of the dense tensors its compress launch copied into the wire
            assert on_gpu and not chunked and not two_phase and [g[2] for g in groups] == [g[2] for g in self._groups]
            mean_in_tail = dense_job is not None and not (plain and R == 1)
            tails = {}
            for gi, g in enumerate(groups):
                obj = g[2]
                rs = overlap["resets"].get(id(obj)) or []
                pairs = (self._rng_state[gi * self.RNG_SLOTS:(gi + 1) * self.RNG_SLOTS]
                         if (step_rng and obj.rng_pairs is not None) else None)
                first = gi == 0 and mean_in_tail
                if first or pairs is not None or rs:
                    tails[gi] = native.StepTail(rows=dense_job[0] if first else None, out=self._dense_mean[dense_job[1]] if first else None,
                                                rng_state=pairs, reset=rs[0] if rs else None, ticket=self._ticket_for(gathered.device, gi))
                for extra in rs[1:]:
                    resets.append(extra)
            if step_rng and all(g[2].rng_pairs is not None for g in groups):
                step_rng = False
            if mean_in_tail:
                dense_job = (None, dense_job[1])
        elif on_gpu and not chunked and not two_phase and os.environ.get("GQ_STEP_TAIL", "1") != "0":
            takers = [gi for gi, g in enumerate(groups) if g[2].takes_tail]
            mean_in_tail = dense_job is not None and not (plain and R == 1)
            if takers and (mean_in_tail or step_rng or resets):
                tail_group = takers[-1]
                tail = native.StepTail(rows=dense_job[0] if mean_in_tail else None,
                                       out=self._dense_mean[dense_job[1]] if mean_in_tail else None,
                                       rng_state=self._rng_state if step_rng else None, reset=resets.pop(0) if resets else None,
                                       ticket=self._ticket_for(gathered.device, 0))
                step_rng = False
                if mean_in_tail:
                    dense_job = (None, dense_job[1])      # (done by the decode launch)

        main_stream = torch.cuda.current_stream() if overlap is not None else None

        def decode_range(lo, hi, first):
            """The tensors whose wire section starts in [lo, hi) (None: all of them)."""
            for gi, (cls, idxs, obj) in enumerate(groups):
                part = None
                if lo is not None:
                    part = seg_ranges.get((gi, lo, hi))
                    if part is None:      # a group's tensors are in wire order: the range is a run of them
                        part = seg_ranges[(gi, lo, hi)] = (sum(1 for i in idxs if self.offsets[i] < lo),
                                                          sum(1 for i in idxs if self.offsets[i] < hi))
                    part = part + (first,)
                t = tails.get(gi) if tails is not None else (tail if gi == tail_group else None)
                side = overlap["side"] if overlap is not None else None
                if side is not None:      # this group's branch: behind its compress, next to the following groups' compresses
                    side.wait_event(overlap["events"][id(obj)])
                    torch.cuda.set_stream(side)
                try:
                    if fused_levels and lo is None and getattr(obj, "_pending_levels", None) is not None:
                        group_views[gi] = obj.levels_decode(plain, t)      # levels + decode (+ tail): one launch
                    else:
                        group_views[gi] = obj.decode_mean(gathered, R, part, plain=plain, tail=t)
                finally:
                    if side is not None:
                        torch.cuda.set_stream(main_stream)
            if overlap is not None and overlap["side"] is not None:
                main_stream.wait_stream(overlap["side"])      # the branches join: the step's graph ends behind all of them
            for i in single:
                if lo is None or lo <= self.offsets[i] < hi:
                    done[i] = self.codecs[i].decode_mean(gathered, self.offsets[i], R, plain=plain)

        if not chunked:
            if pending:
                pending.pop(0).wait()
            decode_range(None, None, True)
        else:
            for k, pnd in enumerate(pending):
                pnd.wait()
                decode_range(pnd.lo, self.user_bytes if pnd.hi is None else pnd.hi, k == 0)
        draws2 = self._draws(gathered.device) if two_phase else None     # the second phase compresses again: new draws
        if two_phase:
            self._second_phase_base(gathered.device)
        if phase2_headers is not None and resets is None:
            resets = []     # (capture of a two-phase apply) the second phase's accumulator resets ride in the aggregate's last small launch
        sources = []        # the lists of output views this call's result is assembled from (persistent objects, see below)
        for gi, (cls, idxs, obj) in enumerate(groups):
            gs = group_views[gi]
            if two_phase:
                # ps_quantizer.py:52-61, replicated on every rank (salt 0, the ranks' shared seed stream); with error
                # feedback g += server_error and server_error = g - decoded happen inside the launches
                serr = [self.parameters[i].server_error for i in idxs] if self.error_feedback else None
                hdr = phase2_headers[gi] if phase2_headers is not None else None      # (capture: the device copy of THIS launch's table)
                with shared_seeds(self._second_phase_seed):
                    dec = obj.roundtrip(list(gs), self.capacity, 0, serr, 1.0, draws=draws2, rng_slot=self.TWO_PHASE_RNG_SLOT,
                                        graph_header=hdr, defer_reset=resets if hdr is not None else None)
                if dec is None:     # not batchable this step: per-tensor second phase below
                    for i, g in zip(idxs, gs):
                        done[i] = g
                        single.append(i)
                    continue
                gs = dec
            sources.append((idxs, gs))
        if dense_job is not None:
            rows, k = dense_job
            if rows is None:
                pass                                     # the mean rode in a decode launch (tail)
            elif plain and R == 1:
                self._dense_mean[k].copy_(rows[0])      # the ring's hop: the payload as it is (a -0 stays -0)
            elif rows.device.type == "cuda":
                # stack().mean(0) with the CPU's arithmetic (true division); the same launch steps the draws' step words
                native.mean_rows(rows, self._dense_mean[k], rng_state=self._rng_state if step_rng else None,
                                 reset=resets.pop(0) if resets else None)
                step_rng = False
            else:
                torch.mean(rows, dim=0, out=self._dense_mean[k])   # stack().mean(0) of the reference, all at once
            sources.append((self.dense_idx, self._dense_views[k]))
        if step_rng:        # no identity-compressed tensors to average (or the ring's plain hop): a launch of its own
            native.rng_step(self._rng_state, reset=resets.pop(0) if resets else None)
        for dst, src in (resets or ()):      # (whole-step capture) what no launch of the aggregate took along
            _kernel_copy(dst, src)
        if not single and not done:
            # everything came out of multi-tensor launches: the result is a fixed interleaving of a few PERSISTENT view lists
            # (two output buffers per group used in turn, their per-tensor views built once), so the parameter-ordered
            # list is assembled once per combination and reused (161 dictionary stores + look-ups per step otherwise)
            key = tuple(id(v) for _, v in sources)
            hit = self._assembled.get(key)
            if hit is not None and all(a is b for a, (_, b) in zip(hit[0], sources)):
                return hit[1]
            for idxs, vs in sources:
                for i, v in zip(idxs, vs):
                    done[i] = v
            out = [done[i] for i in range(self.num_layers)]
            if len(self._assembled) > 8:
                self._assembled.clear()
            self._assembled[key] = ([v for _, v in sources], out)      # (holding the lists keeps their ids from being reused)
            return out
        for idxs, vs in sources:
            for i, v in zip(idxs, vs):
                done[i] = v
        if two_phase:
            for i in single:
                # ps_quantizer.py:52-61 -- identical on every rank (salt 0, the ranks' shared seed stream)
                param, codec, g = self.parameters[i], self.codecs[i], done[i]
                kw = self._slice(draws2, i)
                with shared_seeds(self._second_phase_seed):
                    if self.error_feedback:
                        g = g + param.server_error
                        decoded = codec.roundtrip(g, 0, **kw)
                        param.server_error = g - decoded
                        g = decoded
                    else:
                        g = codec.roundtrip(g, 0, **kw)
                done[i] = g
        return [done[i] for i in range(self.num_layers)]

    def apply(self, refresh_grads=False):
        """ps_quantizer.py:46-65.  refresh_grads: accepted for callers of earlier versions; `param.grad` is always evaluated here."""
        if self.recorded == 0:
            return
        world, rank = _dist_world(self.process_group)
        if world > 1:
            # ONE exchange per step: every rank's [users, bytes] block, rank-major (gq_amd.exchange)
            ex = self._ex
            if self.exchange_mode == "auto":
                def step(mode):
                    buf, pend = ex.start(mode, self.recorded, self.cut)
                    self._decode_all(buf, False, pend)
                self.exchange_mode = ex.autotune(
                    step, preflight=lambda mode: ex.start(mode, self.recorded, self.cut, dry_run=True))
            gathered, pending = ex.start(self.exchange_mode, self.recorded, self.cut, cuts=self.cuts)
        else:
            gathered, pending = self._wire[:self.recorded], ()
        decoded = None
        graph_key = None
        fused, self._fused = self._fused, None
        if fused is not None and self.recorded == 1 and world == 1:
            decoded = fused      # record() has replayed this step's decode-mean already (self._step_graphs)
        elif (self.use_graphs and len(pending) <= 1 and gathered.device.type == "cuda"
                and all(g[2] is not None and g[2].ready for g in self._groups) and not torch.cuda.is_current_stream_capturing()
                and (not self.two_phase or (not self._draw_total and all(g[2].graphable() for g in self._groups)))):
            # gq_graph: the decode-mean launches (+ the dense tensors' mean) of an apply that has been seen with these buffers
            # before replay as ONE graph launch; the two output buffers are used in turn, so two graphs alternate.  With
            # several ranks the exchange stays outside: its one transfer is waited for first (the split transport, whose
            # decode is interleaved with its second transfer, keeps its eager launches)
            for pnd in pending:
                pnd.wait()
            pending = ()
            graph_key = self._apply_key(gathered)
            if self.two_phase and self.error_feedback:      # (the server residuals' addresses are in the second phase's table)
                graph_key += (tuple(p.server_error.data_ptr() for p in self.parameters),)
            ent = self._apply_graphs.get(graph_key)
            if ent is not None and ent[1] is not None:
                if self.two_phase:      # the re-compress folds into the accumulators an eager record() may have left used
                    for g in self._groups:
                        g[2].ensure_clean()
                ent[1].replay()
                for g in self._groups:
                    if not self.two_phase:      # (two-phase: decode-mean and the re-decode took one output buffer each -- back to the first)
                        g[2]._out_turn ^= 1
                    g[2]._last_ptrs = None      # (two-phase: the last eager upload is not what the device header was last used with)
                if len(self.dense_idx) >= 2:
                    self._dense_turn ^= 1
                decoded = ent[2]
        if decoded is None:
            decoded = self._decode_all(gathered, self.two_phase, pending)
            if graph_key is not None and self._plan is not None and not self._plan[2]:     # (no per-tensor decodes in the plan)
                ent = self._graph_entry(self._apply_graphs, graph_key)
                if ent is not None and ent[0] >= 2 and ent[1] is None:
                    after = ([g[2]._out_turn for g in self._groups], self._dense_turn)
                    try:
                        for g, t in zip(self._groups, graph_key[2]):      # the capture re-issues the launches of THIS apply
                            g[2]._out_turn = t
                        self._dense_turn = graph_key[3]
                        # two-phase (ps_quantizer.py:52-61): the second phase's encode reads a device copy of the table its eager run
                        # has just sent (the decode-mean's output buffers, the server residuals), which belongs to this graph
                        hdrs2 = [g[2]._host[g[2]._last_slot].to(gathered.device) for g in self._groups] if self.two_phase else None
                        calls = self._phase2_calls
                        graph = self._new_graph()
                        with _capturing(graph):
                            again = self._decode_all(gathered, self.two_phase, (), phase2_headers=hdrs2)
                        assert self._phase2_calls == calls, "a captured second phase must not take per-call seeds"
                        if len(again) == len(decoded) and all(a is b for a, b in zip(again, decoded)):
                            ent[1], ent[2] = self._replayable(graph), decoded
                            ent.append(hdrs2)      # (kept alive with the graph)
                    except Exception as e:      # this apply has already run eagerly
                        self.use_graphs = False
                        for g in self._groups:
                            g[2]._graph_tables_abort()
                        import warnings
                        warnings.warn("gq_graph: capturing an apply failed (%s); continuing with eager launches" % (e,))
                    finally:
                        for g, t in zip(self._groups, after[0]):
                            g[2]._out_turn = t
                        self._dense_turn = after[1]
        # ps_quantizer.py:63 `param.grad.data = g`: `param.grad` is evaluated HERE, at apply time -- a caller that replaced a
        # parameter's .grad object after the last record() gets the mean in the object it holds now.  The C++ helper reads
        # p.grad() without building Python objects (the 161 attribute look-ups were the step's largest host cost, which is
        # why earlier rounds rebound the objects record() had seen); without the helper the look-ups are paid.
        if _HOST is not None and hasattr(_HOST, "set_grad_data") and type(decoded) is list and len(decoded) == len(self.parameters):
            _HOST.set_grad_data(self.parameters, decoded)
        else:
            for p, g in zip(self.parameters, decoded):
                p.grad.data = g
        self.recorded = 0

    aggregate = apply


# --------------------------------------------------------------------------------------
# Ring quantizer (the reference's other --mode; sequential by construction)
# --------------------------------------------------------------------------------------
class RingQuantizer(PSQuantizer):
    """quantizers/ring_quantizer.py:7-49: user k adds user k-1's decoded running sum to its own
    gradient and re-compresses; the result is the LAST user's decode (a sum, not a mean).

    Built on PSQuantizer's wire and multi-tensor kernels: one record() is [grad += running] + the
    parameter-server record (error feedback included, ring_quantizer.py:33-40 == ps_quantizer.py:34-39)
    + a decode of the wire just written.  Under torch.distributed the ring is real: the users are
    numbered rank-major, the compressed wire (not the decoded sum) travels rank -> rank+1 as ONE
    point-to-point message over xGMI when a rank's users are done, and the last rank broadcasts the
    final wire, which every rank decodes.  The chain is sequential by construction (each hop
    re-compresses the sum of everything before it), so it costs `world` encode latencies."""

    def __init__(self, Compressor, parameters, args, process_group=None, codec_factory=None):
        two_phase = args.two_phase
        args.two_phase = False           # ring_quantizer.py has no second phase and no server residual
        try:
            super().__init__(Compressor, parameters, args, process_group, codec_factory)
        finally:
            args.two_phase = two_phase
        self.two_phase = False
        self.running = None              # decoded running sum, one tensor per parameter
        self._inbox = None

    def record(self, user, epoch):
        world, rank = _dist_world(self.process_group)
        dev = self.parameters[0].grad.device
        if self.running is None and rank > 0:
            # first local user of a rank > 0: the running sum arrives compressed from the previous rank
            import torch.distributed as dist
            if self._inbox is None or self._inbox.device != dev:
                self._inbox = torch.empty((1, self.user_bytes), dtype=torch.uint8, device=dev)
            dist.recv(self._inbox.view(-1), src=self._peer(rank - 1), group=self.process_group)
            self._ready_for_wire(dev)
            self.running = self._decode_all(self._inbox, False, plain=True)
        if self.running is not None:     # ring_quantizer.py:31-32 (user != 0)
            torch._foreach_add_([p.grad.data for p in self.parameters], list(self.running))
        self.recorded = 0                # every user re-uses wire slot 0
        super().record(user, epoch)
        self.running = self._decode_all(self._wire[:1], False, plain=True)

    def _peer(self, group_rank):
        import torch.distributed as dist
        return group_rank if self.process_group is None else dist.get_global_rank(self.process_group, group_rank)

    def _ready_for_wire(self, dev):
        """A rank may have to decode a wire before it has encoded anything: the multi-tensor kernels'
        segment tables (the layout part; pointers are not needed for a decode) must exist."""
        for grp in (self._groups if dev.type == "cuda" else []):
            if grp[2] is None:
                self._make_group(grp, dev)
            if not grp[2].ready:
                grp[2].upload_layout()

    def apply(self):
        world, rank = _dist_world(self.process_group)
        if world > 1:
            import torch.distributed as dist
            if self.recorded == 0 and self.running is None:
                return
            dev = self._wire.device
            if rank < world - 1:
                dist.send(self._wire[0], dst=self._peer(rank + 1), group=self.process_group)
            final = self._wire[:1] if rank == world - 1 else self._inbox_for(dev)
            dist.broadcast(final.view(-1), src=self._peer(world - 1), group=self.process_group)
            if rank != world - 1:
                self.running = self._decode_all(final, False, plain=True)
        if self.running is not None:     # ring_quantizer.py:45-46
            if _HOST is not None and type(self.running) is list and len(self.running) == len(self.parameters):
                _HOST.set_grad_data(self.parameters, self.running)      # (the C++ walk: 30 -> ~18 us for 161 parameters)
            else:
                for param, g in zip(self.parameters, self.running):
                    param.grad.data = g
        self.running = None
        self.recorded = 0

    def _inbox_for(self, dev):
        if self._inbox is None or self._inbox.device != dev:
            self._inbox = torch.empty((1, self.user_bytes), dtype=torch.uint8, device=dev)
        return self._inbox

    aggregate = apply


def Quantizer(Compressor, parameters, args, **kw):
    """quantizers/base_quantizer.py:5-10."""
    if args.mode == 'ps':
        return PSQuantizer(Compressor, parameters, args, **kw)
    elif args.mode == 'ring':
        return RingQuantizer(Compressor, parameters, args, **kw)
    assert False, "mode {} not recognized".format(args.mode)


__all__ = ["Quantizer", "PSQuantizer", "RingQuantizer"]
