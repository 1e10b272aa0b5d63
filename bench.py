#!/usr/bin/env python3
"""Headline benchmark: gradient elements quantised per second, HSQ d=16 k=8 n=6.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hsq|resnet50|qsgd] [--exchange MODE]

`--gpus N` with N > 1 from a bare shell launches its own N ranks (a child `python -m torch.distributed.run
--nproc-per-node N ... bench.py ...`, started BEFORE anything touches the GPU); under torch.distributed.run
(RANK / WORLD_SIZE in the environment) it is one of the ranks.  One rank per GPU over RCCL.

One step (workload hsq, BASELINE configs[1] / [3]) = one pass of the hot path over one synthetic
25,000,000-float32 gradient per rank, inputs resident in HBM (three gradients used in turn, so that a step
never finds its input in the 256 MiB Infinity Cache):
    encode (bf16x3 MFMA prefilter + exact f32 rescoring, one launch) -> level quantiser
    -> [exchange of the (codes, levels, lb, ub) wire between the ranks, gq_amd/exchange.py]
    -> decode + mean over ranks (rank-ascending, == torch.stack(decoded).mean(0), ps_quantizer.py:48).
`value` = ranks * 25e6 * K / (max-over-ranks time of K steps): END TO END.  The compress-only rate
(encode + levels, SURVEY 8d's definition of the metric) is in `compress_only`.  Weak scaling: every rank
owns a full-size gradient (it is one of the reference's `num_users`).

Workload resnet50 (BASELINE configs[2]): the ResNet-50/CIFAR parameter list (161 tensors, 23.5 M elements; 76 go through
the codebook, 85 small ones travel as f32) through PSQuantizer.record + apply with NearestNeighborCompressor c_dim=16
k_bit=8 n_bit=6 (quantizers/ps_quantizer.py:27-65): one multi-tensor encode, one level launch, one decode-mean per step.
Workload qsgd (BASELINE configs[4]): the same list with QSGDCompressor c_dim=128 n_bit=2 (packed 4-bit wire).
Both hand every step fresh tensor objects of three input lists (apply() rebinds .grad.data like the reference).

Extra objects on the JSON line:
  roofline      dominant kernel: algorithmic bytes per launch / its average launch duration, measured with HIP
                events attached to its dispatches inside the timed region; peak = 8 TB/s HBM3E.  `frac` is that kernel
                alone, `frac_compress` SURVEY 8(d)'s own definition: 4.125 B x size / t(encode + levels).
                `traffic`: HBM bytes per launch from PMC counters (FETCH_SIZE, WRITE_SIZE in separate rocprofv3
                passes of a short child run of this script, gfx950 corrections of MI355X_MICROARCH.md), measured
                in this run when rocprofv3 is present (--traffic live|auto), else the committed figure.
  cpu_baseline  the CPU oracle (oracle/gq_oracle.c: OpenMP, hardware FMA, eight codeword chains per AVX2 register)
                timed on a bounded sample of the same gradient on this box's host cores (rank 0, N=1 only), with
                the per-thread rate and the Python reference's own figures (other machine) beside it.
  exchange      N > 1: backend, ranks, the transport used and the per-transport times measured before the timed
                region (all-gather / direct all-pairs / split with overlapped decode).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
sys.path.insert(0, ROOT)

SIZE = 25_000_000
C_DIM, K_BIT, N_BIT = 16, 8, 6
ALGO_BYTES_PER_ELEM = 4.125          # 4 B read + (1 B code + 1 B level) / 16 written   
QSGD_ALGO_BYTES_PER_ELEM = 4.0 + 0.5 + 4.0 / 128     # 4 B read + 4-bit code + one f32 norm per 128-element bucket
FLOP_PER_ELEM = 512                  # 2 * d * K / d
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PREWARM_STEPS = 3000
EF_ENCODE_BYTES_PER_ELEM = 12.0 + 5.0 / 16    # error-feedback encode: grad + error read, grad written, u + code per 16 elements
EF_ALGO_BYTES_PER_ELEM = 20.0 + 10.0 / 16     # + the level launch: u + grad read, error + level written
# Error-feedback workloads run with --scale 0: the reference adds scale * error INTO the gradient in place (ps_quantizer.py:35), and
# this bench puts the same three input lists under the parameters again and again -- with the reference's scale ('exp': 0.46 at
# epoch 1) every visit would add 0.46 x (65 % quantisation error) to the inputs themselves, which overflow after ~400 visits
# (seen: 6 ms steps once the values were infinite).  scale 0 runs the same launches, bytes and arithmetic (error read, product,
# sum, gradient written back, residual written) on inputs that stay what they are; training rewrites its gradients every step.
EF_BENCH_SCALE = "0.0"
WATCHDOG_EXIT = 3                     # exit code of a rank whose watchdog ended the job (see Watchdog)
TIMED_WINDOWS = 5                     # windows of --steps steps each; the line reports the median window (+ min / max)
TRAFFIC_FILE = os.path.join("profiles", "hbm_traffic.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="hsq", choices=["hsq", "resnet50", "qsgd"])
    ap.add_argument("--random", type=int, default=0, choices=[0, 2],
                    help="hsq: 0 = deterministic levels (the bit-exact configuration); 2 = on-device stochastic rounding")
    ap.add_argument("--exchange", default=os.environ.get("GQ_EXCHANGE", "allgather"),
                    choices=["auto", "allgather", "direct", "split", "pipelined"],
                    help="N > 1: how the wire travels.  Default: the in-place all-gather (the one collective every backend has); "
                         "auto = time all three before the timed region and keep the fastest (opt-in: direct / split have not "
                         "met RCCL with more than one rank yet); pipelined (opt-in, never picked by auto): the codes travel "
                         "under the level kernel (hsq) / the wire travels and is decoded in $GQ_PIPELINE_CHUNKS ranges (lists)")
    ap.add_argument("--wire-levels", default=os.environ.get("GQ_WIRE_LEVELS", "auto"), choices=["auto", "bytes", "packed6"],
                    help="how the 6-bit levels travel: a byte each, or four per three bytes (12.5 %% less wire; same decode bits); "
                         "auto = packed6 when there is an exchange (N > 1, all-gather / direct), bytes at N = 1")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "file", "off"],
                    help="roofline.traffic: live = two short child runs under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE); "
                         "auto = live at N=1 when rocprofv3 is on PATH, else the committed profiles/hbm_traffic.json")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--clock-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-workloads", action="store_true",
                    help="workload hsq at N = 1: skip the compact `workloads` object (ResNet-50 list with HSQ, with QSGD, and with "
                         "HSQ on back-propagated gradients: BASELINE configs[2] / [4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True,
                    help="--workload resnet50 / qsgd (default, and the library's default): record() and apply() replay their device "
                         "work from HIP graphs per set of gradient addresses (gq_graph); stochastic rounding draws from streams keyed "
                         "by { seed, step } words in device memory (gq_rng = 'device', GQ_RANDOM_DEVICE_COUNTER)")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="--workload resnet50 / qsgd: eager launches (the same draws)")
    ap.add_argument("--cpu-scaling", action="store_true", help="print the CPU oracle's thread scaling on this host and exit (no GPU work)")
    ap.add_argument("--two-launches", action="store_true",
                    help="N = 1: levels and decode as two launches (as with N > 1) instead of gq_hsq_levels_decode")
    ap.add_argument("--ef", action="store_true", help="--workload resnet50: error feedback (ps_quantizer.py:34-39)")
    ap.add_argument("--two-phase", action="store_true", help="--workload resnet50: the second phase (ps_quantizer.py:52-61)")
    ap.add_argument("--c-dim", type=int, default=0, help="--workload resnet50: sub-dimension (default 16; main.py's own default is 32)")
    ap.add_argument("--k-bit", type=int, default=0, help="--workload resnet50: codebook bits (default 8; 5 / 6: K = 32 / 64, the codebook files from tests/golden/codebooks)")
    ap.add_argument("--n-bit", type=int, default=0, help="--workload resnet50: level bits (default 6; main.py's own default is 8)")
    ap.add_argument("--no-variants", action="store_true", help="skip the untimed random=2 / 1e-3-scale side measurements")
    return ap.parse_args()


def self_launch(args):
    """`bench.py --gpus N` from a bare shell: become the parent of N ranks.  Nothing here touches the GPU."""
    import torch
    have = torch.cuda.device_count()      # counting devices does not initialise the GPU
    if os.environ.get("GQ_BENCH_BACKEND", "nccl") == "nccl" and have < args.gpus:
        sys.exit("bench.py --gpus %d: this machine shows %d GPU(s); one rank per GPU is the contract.  For a functional run of "
                 "the %d-rank code path on fewer GPUs: GQ_BENCH_BACKEND=gloo python bench.py --gpus %d (its times mean nothing)"
                 % (args.gpus, have, args.gpus, args.gpus))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import tempfile
    mark = os.path.join(tempfile.gettempdir(), "gq_bench_watchdog_%d.json" % os.getpid())      # a rank's watchdog leaves its record here
    env["GQ_BENCH_WATCHDOG_FILE"] = mark                                                          # (torch.distributed.run reports any child failure as 1)

    def launch(extra, port):
        if os.path.exists(mark):
            os.remove(mark)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
        return subprocess.call(cmd, env=env)
    rc = launch([], port)
    if rc != 0 and os.path.exists(mark) and args.exchange != "allgather":
        # a rank's watchdog ended the job (a collective of the requested transport did not complete): ONE fresh set of
        # ranks -- new processes, never a re-exec of one that has touched the GPU -- on the transport every backend has
        print("bench.py: the ranks' watchdog ended the run with --exchange %s; starting fresh ranks with --exchange allgather"
              % args.exchange, file=sys.stderr)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        rc = launch(["--exchange", "allgather"], port)
    if os.path.exists(mark):
        os.remove(mark)
    sys.exit(rc)


class Watchdog(object):
    """N > 1 only.  A collective that never completes -- the first RCCL contact of a transport on a new node -- would hold the
    job until the launcher's own limit.  A daemon thread ends THIS rank with exit code WATCHDOG_EXIT once the phase the
    main thread has announced (`enter`) has been running for $GQ_BENCH_TIMEOUT_S seconds (default 300; the process group's own
    timeout is the same figure): what was in flight goes to stderr as one JSON object, torch.distributed.run then ends the
    other ranks, and a self-launched `bench.py --gpus N` starts fresh ranks on the all-gather (self_launch).  os._exit, not
    an exec: a process that has initialised the GPU is never replaced by another program."""

    def __init__(self, rank, world, limit_s):
        import threading
        self.rank, self.world, self.limit = rank, world, float(limit_s)
        self.phase, self.since, self.info = "start", time.monotonic(), {}
        self._stop = threading.Event()
        if world > 1 and self.limit > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def enter(self, phase, **info):
        self.phase, self.since, self.info = phase, time.monotonic(), info

    def done(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(1.0):
            if time.monotonic() - self.since > self.limit:
                rec = json.dumps({"bench_watchdog": "rank %d of %d: phase %r has not completed in %.0f s; ending the job"
                                                    % (self.rank, self.world, self.phase, self.limit),
                                  "phase": self.phase, "info": self.info, "exit_code": WATCHDOG_EXIT})
                print(rec, file=sys.stderr, flush=True)
                try:
                    if os.environ.get("GQ_BENCH_WATCHDOG_FILE"):      # a self-launched run: the parent looks for this
                        with open(os.environ["GQ_BENCH_WATCHDOG_FILE"], "a") as f:
                            f.write(rec + "\n")
                finally:
                    os._exit(WATCHDOG_EXIT)


WATCHDOG = None


def watch(phase, **info):
    if WATCHDOG is not None:
        WATCHDOG.enter(phase, **info)


def usable_cpus():
    """CPUs this process may actually keep busy: the scheduler affinity, capped by the cgroup's CPU quota.  (The GPU
    boxes show 256 host CPUs under a quota of 16: 128 OpenMP threads run a 25 M-element compress in bursts at 5.6e9
    elements/s and, sustained, get throttled to 3.7e8 -- a fifth of what 16 threads sustain.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    why = "sched_getaffinity"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(quota) // int(period) >= 1 and int(quota) // int(period) < n:
            n, why = int(quota) // int(period), "cgroup cpu.max %s/%s" % (quota, period)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and 1 <= q // per < n:
                n, why = q // per, "cgroup cfs quota %d/%d" % (q, per)
        except Exception:
            pass
    return n, why


def cpu_baseline(g_host, cb):
    """Time the CPU oracle's whole compress (encode + min/max + levels) on the rank-0 gradient, repeated until about
    5 s of wall time have been spent (bounded sample: ~80 core-seconds on the 16 CPUs of a GPU box), on all host cores and
    on ONE thread (2 s)."""
    import oracle
    oracle.build()
    avail, why = usable_cpus()
    threads = max(1, min(oracle.num_threads(), avail))
    oracle.set_num_threads(threads)
    oracle.hsq_compress(g_host[:16 * 20000], cb, N_BIT, 0)       # warm the thread pool
    n = SIZE
    reps, spent = 0, 0.0
    while spent < 5.0 and reps < 4000:
        t0 = time.perf_counter()
        oracle.hsq_compress(g_host[:n], cb, N_BIT, 0)
        spent += time.perf_counter() - t0
        reps += 1
    oracle.set_num_threads(1)
    n1 = 16 * 500_000
    oracle.hsq_compress(g_host[:n1], cb, N_BIT, 0)
    reps1, spent1 = 0, 0.0
    while spent1 < 2.0 and reps1 < 200:
        t0 = time.perf_counter()
        oracle.hsq_compress(g_host[:n1], cb, N_BIT, 0)
        spent1 += time.perf_counter() - t0
        reps1 += 1
    oracle.set_num_threads(threads)
    value = n * reps / spent
    return {"value": value, "unit": "elements/s", "cores": threads, "kind": "port", "vectorised": True,
            "per_thread": value / threads, "one_thread": {"value": n1 * reps1 / spent1, "unit": "elements/s",
                                                          "sample": "%d elements x %d repetitions, 1 thread" % (n1, reps1)},
            "sample": "the full 25,000,000-element rank-0 gradient, HSQ compress (encode+min/max+levels), "
                      "%d repetitions in %.1f s wall, OpenMP %d threads (%.0f core-seconds); C restatement of the "
                      "reference built with -mavx2 -mfma: every score is the reference's ascending fmaf chain, eight "
                      "codewords per 256-bit register (oracle/gq_oracle.c); min/max single-threaded like torch's"
                      % (reps, spent, threads, spent * threads),
            "host_cpus": os.cpu_count(), "usable_cpus": avail, "usable_cpus_from": why,
            "reference_python": {"value_8_threads": 2.17e7, "value_1_thread": 5.2e6, "unit": "elements/s",
                                 "note": "the reference itself (PyTorch CPU ops), measured in the survey container, not on "
                                         "this box (BASELINE.md section 2); it cannot travel to the GPU box"}}


def cpu_scaling(g_host, cb):
    """`--cpu-scaling`: the cpu_baseline leg's thread sweep (what the oracle makes of 1 .. all host threads on this box),
    printed as a table instead of the JSON line.  Part of the CPU-baseline leg: the only place besides it that runs the oracle."""
    import oracle
    oracle.build()
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", usable_cpus())
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(f):
            print(f, open(f).read().strip())
    for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if th > os.cpu_count():
            break
        oracle.set_num_threads(th)
        oracle.hsq_compress(g_host[:16 * 100000], cb, N_BIT, 0)
        best_e = best_c = 1e9
        for _ in range(3):
            t = time.perf_counter()
            oracle.hsq_encode(g_host, cb)
            best_e = min(best_e, time.perf_counter() - t)
            t = time.perf_counter()
            oracle.hsq_compress(g_host, cb, N_BIT, 0)
            best_c = min(best_c, time.perf_counter() - t)
        print("threads %3d  encode %8.1f M elements/s  compress %8.1f M elements/s" % (th, SIZE / 1e6 / best_e, SIZE / 1e6 / best_c))


def live_traffic(workload, kernel_match):
    """HBM bytes per launch of the kernels whose name contains `kernel_match`, from PMC counters collected NOW: two
    child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE do not fit one pass), a few steps each.
    FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read, so the
    read side is doubled (MI355X_MICROARCH.md, HBM section).  Returns (bytes or None, how)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled: no nested rocprofv3"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gq_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--traffic-child", "--workload", workload, "--steps", "6", "--warmup", "2",
               "--no-cpu-baseline", "--no-variants", "--traffic", "off"]
        try:
            # a child process (never an exec from this GPU-initialised process); the profiler's own program is python itself
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=150, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == counter and kernel_match in r.get("Kernel_Name", ""):
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, "rocprofv3 --pmc %s produced no rows for %s" % (counter, kernel_match)
            got[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as e:      # a box without counter access: fall back to the committed figure
            return None, "rocprofv3 --pmc %s failed: %s" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = got["FETCH_SIZE"][0] * 1024 * 2, got["WRITE_SIZE"][0] * 1024
    return rd + wr, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate child runs of bench.py "
                     "(%d / %d launches of %s); KiB counters, read side doubled (gfx950 wide-read correction): "
                     "%.1f MB read + %.1f MB written" % (got["FETCH_SIZE"][1], got["WRITE_SIZE"][1], kernel_match,
                                                         rd / 1e6, wr / 1e6))


def traffic_for(args, world, workload, kernel_match, file_key):
    """roofline.traffic and where it came from."""
    if args.traffic == "off" or args.traffic_child:
        return None, "not collected (--traffic off)"
    why = "--traffic file"
    if args.traffic in ("auto", "live") and world == 1:
        t, why = live_traffic(workload, kernel_match)
        if t is not None:
            return t, why
    if file_key and os.path.exists(os.path.join(ROOT, TRAFFIC_FILE)):
        try:
            t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE))).get(file_key)
            return t, TRAFFIC_FILE + " (committed figure of an earlier rocprofv3 --pmc run, NOT measured in this run: %s)" % why
        except Exception:
            pass
    return None, why


CLOCK_LIB = os.path.join(ROOT, "gradient-quantization_amd", "libgq_hsq_clock.so")
ISA_FILE = os.path.join("profiles", "r06_encode_isa_floor.json")


def clock_child():
    """`bench.py --clock-child`, run by in_kernel_clock() with GQ_LIB_PATH = the DIAGNOSTIC twin of the library (hsq_encode_pf.hip
    built with -DGQ_PF_STAMPS: s_memtime / s_memrealtime stamps; never the product): two seconds of back-to-back encodes of a
    random 25 M-element gradient, then the stamps of the last launch -- shader cycles of the tile loop over its real time
    (100 MHz counter), median over the workgroups' waves (MI355X_MICROARCH.md, 'DVFS give-back' item 6)."""
    import numpy as np
    import torch
    from gq_amd import native
    from gq_amd.codebook import load_codebook
    dev = torch.device("cuda:0")
    cb = torch.from_numpy(load_codebook(C_DIM, 2 ** K_BIT)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    # the conditions of the timed step (run_hsq at N = 1): three gradients in turn, every encode followed by the fused
    # level quantiser + decode of the rank's own payload -- the clock the chip holds under the STEP, not under
    # back-to-back encodes of one Infinity-Cache-warm input (2.34 GHz there against ~2.0 under the step's mix)
    grads = [torch.randn(SIZE, device=dev, generator=gen) for _ in range(3)]
    M = SIZE // C_DIM
    codes, u = torch.empty(M, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
    levels, lb_ub = torch.empty(M, dtype=torch.uint8, device=dev), torch.empty(2, dtype=torch.float32, device=dev)
    out = torch.empty(SIZE, dtype=torch.float32, device=dev)
    ws = native.new_workspace(dev, M)
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < 2.0:
        for _ in range(200):
            native.hsq_encode(grads[i % 3], cb, codes, u, ws)
            native.hsq_levels_decode(u, N_BIT, 0, None, 0, ws, lb_ub, levels, codes, cb, out, False)
            i += 1
        torch.cuda.synchronize()
    native.hsq_encode(grads[i % 3], cb, codes, u, ws)      # (the level kernel reads the head of the workspace only; the stamps sit behind the log)
    torch.cuda.synchronize()
    first = native.WS_LOG_FIRST
    raw = ws[first + M - 65536:first + M - 65536 + 256 * 8 * 12 * 2].contiguous().view(torch.int64).view(-1, 12).cpu().numpy().astype(np.float64)
    cyc, real, tiles = raw[:, :6].sum(1), (raw[:, 8] - raw[:, 7]) / 100.0, raw[:, 9]      # cycles, us, tiles per wave
    ok = (real > 0) & (tiles > 0)
    print(json.dumps({"in_kernel_clock_ghz": float(np.median(cyc[ok] / real[ok]) / 1e3),
                      "cycles_per_tile_and_wave": float(cyc[ok].sum() / tiles[ok].sum()), "waves": int(ok.sum())}))


def in_kernel_clock():
    """(clock in GHz or None, how): a child run of this script on the stamped twin of the library (see clock_child)."""
    if not os.path.exists(CLOCK_LIB):
        return None, "gradient-quantization_amd/libgq_hsq_clock.so was not built"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is being profiled: the stamped twin's launches would be counted as the product kernel's"
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--clock-child"], env=dict(os.environ, GQ_LIB_PATH=CLOCK_LIB),
                             capture_output=True, text=True, timeout=120).stdout.strip().splitlines()
        rec = json.loads(out[-1])
        return rec, ("child run of bench.py on libgq_hsq_clock.so (hsq_encode_pf.hip built with -DGQ_PF_STAMPS; never the product "
                     "library): 2 s of the timed step's launches (encode + fused levels-decode, three random gradients in turn), then "
                     "s_memtime cycles of the last encode's tile loop / its s_memrealtime span, median over %d waves" % rec["waves"])
    except Exception as e:
        return None, "clock child failed: %s" % (e,)


def issue_floor(clock_ghz, cus):
    """The launch's ISSUE floor: tiles x (4 cycles per VALU instruction + 8 per MFMA of the tile loop, from the compiler's ISA:
    tools/isa_count.py -> profiles/r04_encode_isa_floor.json) / SIMDs / in-kernel clock.  What the instruction stream allows
    with every issue slot used; tail, prologue, second pass and exact scans come on top."""
    try:
        isa = json.load(open(os.path.join(ROOT, ISA_FILE)))
    except Exception:
        return None, None
    tiles = -(-(SIZE // C_DIM) // isa["subvectors_per_tile"])
    if not clock_ghz:
        return None, isa
    return tiles * isa["issue_cycles_per_tile"] / (4.0 * cus) / (clock_ghz * 1e9) * 1e3, isa


def event_ms(torch, fn, n=50, warm=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    import numpy as np
    if args.cpu_scaling:     # the CPU-baseline leg's thread sweep: host only
        sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
        from gq_amd.codebook import load_codebook
        cpu_scaling(np.random.RandomState(1234).standard_normal(SIZE).astype(np.float32), load_codebook(C_DIM, 1 << K_BIT))
        return
    if args.clock_child:
        clock_child()
        return
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback)")
    # GQ_BENCH_BACKEND=gloo is a TEST hook (tests/test_gpu_api.py): it lets several ranks share one GPU, which RCCL
    # refuses, so that the N > 1 code path can be exercised on a single-GPU box.  The driver never sets it.
    backend = os.environ.get("GQ_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a collective that never completes (first RCCL contact on a new node) ends the job after five minutes with the
        # watchdog's error instead of after the default ten
        import datetime
        # (+ 30 s: bench.Watchdog, with the plain figure, speaks first and says which phase and transport were in flight)
        limit = datetime.timedelta(seconds=int(os.environ.get("GQ_BENCH_TIMEOUT_S", "300")) + 30)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)

    from gq_amd import exchange, native
    native.lib()
    global WATCHDOG
    WATCHDOG = Watchdog(rank, world, os.environ.get("GQ_BENCH_TIMEOUT_S", "300"))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload in ("qsgd", "resnet50"):
        line = run_list(args, torch, np, dist, native, exchange, dev, rank, world, backend, barrier)
    else:
        line = run_hsq(args, torch, np, dist, native, exchange, dev, rank, world, backend, barrier)
    watch("report")
    bad_ranks = False
    if rank == 0:
        print(json.dumps(line), flush=True)
        ex = line.get("exchange") if isinstance(line, dict) else None
        if world > 1 and (not ex or ex.get("rccl_ranks") != world):
            # the collective library joined another number of ranks than the launcher started: the line above is not an
            # N-GPU measurement.  Say so where it cannot be missed, and fail.
            print("bench.py --gpus %d: the collectives joined %r ranks, not %d; exchange = %s"
                  % (world, ex.get("rccl_ranks") if ex else None, world, json.dumps(ex)), file=sys.stderr, flush=True)
            bad_ranks = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    WATCHDOG.done()
    if bad_ranks:
        sys.exit(4)


def ranks_counted(torch, dist, dev):
    """How many ranks the collective library actually joins: an all-reduce (sum) of a device one -- not get_world_size(),
    which would read N whatever the backend formed."""
    one = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(one)
    if os.environ.get("GQ_BENCH_TEST_RANKS"):      # TEST hook (tests/test_gpu_api.py): what a library that joined fewer ranks would report
        return int(os.environ["GQ_BENCH_TEST_RANKS"])
    return int(round(float(one.item())))


def collective_library(torch, backend):
    if backend != "nccl":
        return backend, None
    try:
        v = torch.cuda.nccl.version()
        return "rccl", ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        return "rccl", None


def transports_ms(torch, dist, dev, run):
    """Transfer time (no decode) of the in-place all-gather and of the direct all-pairs exchange, both, in every N > 1 run
    (a few MB per rank: milliseconds of untimed work) -- max over ranks.  A transport that ANY rank's dry run refuses is
    skipped on all of them (one all-reduced flag: the ranks never disagree on which collectives follow)."""
    out = {}
    for m in ("allgather", "direct"):
        ok, why = 1.0, ""
        try:
            run(m, True)
        except Exception as e:
            ok, why = 0.0, str(e).splitlines()[0][:120]
        flag = torch.tensor([ok], dtype=torch.float32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) < 1.0:
            out[m] = "refused" + (": " + why if why else " by another rank")
            continue
        t = torch.tensor([event_ms(torch, lambda: run(m), n=20, warm=5)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[m] = float(t.item())
    return out


def exchange_report(torch, counted, ex, mode, requested, backend, world, exchange_ms, per_transport_ms, wire_bytes):
    """N > 1: what a first multi-GPU run needs to explain itself -- the ranks counted by a collective, the library and its
    version, the transport used, and the time of EVERY transport measured in the untimed pass (transfer alone, no decode).
    `counted`: ranks_counted(), called by EVERY rank before rank 0 builds the line."""
    if world == 1:
        return None
    name, version = collective_library(torch, backend)
    return {"backend": name, "rccl_version": version, "rccl_ranks": counted, "ranks": world,
            "transport": mode, "requested": requested, "wire_bytes_per_rank": wire_bytes,
            "ms": exchange_ms, "ms_by_transport": per_transport_ms, "autotune_ms": ex.timings_ms,
            "algorithm_env": {k: os.environ[k] for k in ("NCCL_ALGO", "NCCL_PROTO", "RCCL_MSCCL_ENABLE", "NCCL_DEBUG") if k in os.environ},
            "note": "rccl_ranks = an all-reduce of ones over the process group (== n_gpus when every rank joined); ms = the chosen "
                    "transport alone (no decode), HIP events, untimed pass; ms_by_transport = the same for the in-place all-gather and for "
                    "the direct all-pairs isend / irecv (one xGMI link per peer), so that ONE run says which to default to: direct wins "
                    "when its ms is below allgather's by more than the run-to-run spread (DESIGN.md section 5); autotune_ms = exchange + "
                    "decode-mean per transport (max over ranks), only with --exchange auto"}


def ranks_agree(torch, dist, world, tensors):
    """True if every rank holds the same bits in `tensors` (64-bit sums of the int32 views, all-gathered)."""
    sums = torch.stack([t.contiguous().view(torch.int32).sum(dtype=torch.int64) for t in tensors])
    if world == 1:
        return True
    got = [torch.empty_like(sums) for _ in range(world)]
    dist.all_gather(got, sums)
    return all(bool(torch.equal(got[0], g)) for g in got)


# ------------------------------------------------------------------------------------------------------
# workload hsq: one 25 M-element gradient per rank (BASELINE configs[1], [3])
# ------------------------------------------------------------------------------------------------------
def run_hsq(args, torch, np, dist, native, exchange, dev, rank, world, backend, barrier):
    from gq_amd.codebook import load_codebook
    from gq_amd.wire import HSQWire, SplitHSQWire

    cb_np = load_codebook(C_DIM, 2 ** K_BIT)
    cb = torch.from_numpy(cb_np).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    # three gradients in turn: 300 MB of inputs + the 100 MB decode target between two uses of the same bytes
    grads = [torch.randn(SIZE, device=dev, generator=gen) for _ in range(3)]
    M = SIZE // C_DIM
    if args.wire_levels == "auto":
        args.wire_levels = "packed6" if (world > 1 and args.exchange in ("allgather", "direct", "pipelined")) else "bytes"
    packed6 = args.wire_levels == "packed6" and args.random == 0      # top level 63 (n_bit 6 without stochastic rounding)
    if packed6 and args.exchange in ("split", "auto"):
        sys.exit("bench.py: --wire-levels packed6 goes with --exchange allgather or direct (the split arrangement keeps byte levels)")
    wire, swire = HSQWire(M, packed6), SplitHSQWire(M)
    ex = exchange.WireExchange(world, rank, 1, wire.nbytes, dev)
    sex = exchange.WireExchange(world, rank, 1, swire.nbytes, dev) if world > 1 else None
    codes, levels, lb_ub = wire.views(ex.local[0])
    u = torch.empty(M, dtype=torch.float32, device=dev)
    partials = native.new_workspace(dev, M)
    out = torch.empty(SIZE, dtype=torch.float32, device=dev)
    seed = 1234 + rank

    def compress(g, profile_slot=-1):
        native.hsq_encode(g, cb, codes, u, partials, profile_slot=profile_slot)
        native.hsq_levels(u, N_BIT, args.random, None, seed, partials, lb_ub, levels, packed6)

    def decode(buf):
        native.hsq_decode_sum_packed(buf, M, cb, N_BIT, out, world, wire.codes_off, wire.levels_off, wire.lbub_off,
                                     level_dtype=native.PACKED6 if packed6 else torch.uint8)

    if world > 1:
        s_codes, s_la, s_lb, s_lbub = swire.views(sex.local[0])

        def compress_split(g, profile_slot=-1):
            native.hsq_encode(g, cb, s_codes, u, partials, profile_slot=profile_slot)
            native.hsq_levels(u[:swire.MA], N_BIT, args.random, None, seed, partials, s_lbub, s_la)
            native.hsq_levels(u[swire.MA:], N_BIT, args.random, None, seed + 7919, partials, s_lbub, s_lb)

        def exchange_decode_split():
            buf, pend = sex.start("split", cut=swire.cut)
            pend[0].wait()
            native.hsq_decode_sum_packed(buf, swire.MA, cb, N_BIT, out[:swire.MA * C_DIM], world, swire.codes_off,
                                         swire.levels_a_off, swire.lbub_off)
            pend[1].wait()
            native.hsq_decode_sum_packed(buf, swire.MB, cb, N_BIT, out[swire.MA * C_DIM:], world,
                                         swire.codes_off + swire.MA, swire.levels_b_off, swire.lbub_off)

    def step_pipelined(g, profile_slot=-1):
        """The codes are final when the encode has run, one launch before the levels: their transfer is queued behind the
        encode and travels under the level kernel; the levels (+ lb, ub) follow; the decode waits for both."""
        native.hsq_encode(g, cb, codes, u, partials, profile_slot=profile_slot)
        first = ex.send_range(0, wire.levels_off)
        native.hsq_levels(u, N_BIT, args.random, None, seed, partials, lb_ub, levels, packed6)
        second = ex.send_range(wire.levels_off, wire.nbytes)
        first.wait()
        second.wait()
        decode(ex.gathered)

    def roundtrip(g, profile_slot=-1):
        """One rank: decompress(compress(g)) = the encode, then level quantiser + decode as ONE launch (gq_hsq_levels_decode:
        the same (codes, lb, ub, levels) in the wire and the same decoded tensor as the two calls, bit for bit)."""
        native.hsq_encode(g, cb, codes, u, partials, profile_slot=profile_slot)
        if not native.hsq_levels_decode(u, N_BIT, args.random, None, seed, partials, lb_ub, levels, codes, cb, out, packed6):
            native.hsq_levels(u, N_BIT, args.random, None, seed, partials, lb_ub, levels, packed6)
            decode(ex.gathered)

    fused = world == 1 and not args.two_launches

    def step(i, mode, profile_slot=-1):
        g = grads[i % 3]
        if mode == "split":
            compress_split(g, profile_slot)
            exchange_decode_split()
        elif mode == "pipelined":
            step_pipelined(g, profile_slot)
        elif fused:
            roundtrip(g, profile_slot)
        else:
            compress(g, profile_slot)
            decode(ex.run(mode) if world > 1 else ex.gathered)

    # ---- transport: requested, or the fastest of the three (exchange + decode, max over ranks) ----------
    mode = "allgather"
    if world > 1:
        watch("first exchange", requested=args.exchange)
        if os.environ.get("GQ_BENCH_TEST_HANG") == args.exchange and rank == world - 1:
            time.sleep(1e6)      # TEST hook (tests/test_gpu_api.py): a rank that never joins this transport's first collective
        compress(grads[0])
        compress_split(grads[0])
        if args.exchange == "auto":
            def probe(m):
                if m == "split":
                    exchange_decode_split()
                else:
                    decode(ex.run(m))
            mode = ex.autotune(probe, preflight=lambda m: (sex.start("split", cut=swire.cut, dry_run=True) if m == "split"
                                                           else ex.start(m, dry_run=True)))
        else:
            mode = args.exchange

    # The GPU needs a few hundred milliseconds of load before its clocks and caches settle (measured: 84 us
    # per step over the first 60 steps, 74 us in steady state), so the W warm-up steps are preceded by an
    # untimed pre-warm of PREWARM_STEPS of the same steps (~0.25 s at N=1); the timed region is untouched and
    # the JSON line says so (`prewarm_steps`).
    prewarm = 20 if args.traffic_child else (PREWARM_STEPS if world == 1 else 300)
    watch("prewarm", transport=mode, steps=prewarm)
    for i in range(prewarm):          # a fixed count: every rank issues the same collectives
        step(i, mode)
        if i % 100 == 99:
            torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i, mode)

    # HIP events on the dominant kernel, live in the timed region: a start/stop pair ATTACHED to the encode's
    # dispatch (hipExtLaunchKernelGGL: gq_hsq_encode_ex's profile_slot) on up to 16 of the steps.  An event bracket
    # recorded around the call would also measure 5-8 us of queue bubbles and put them into the timed
    # region (calibrated below for reference).
    stride = max(1, -(-args.steps // 16))          # at most 16 armed steps: an armed dispatch costs a few us of its own
    armed = list(range(0, args.steps, stride))[:16]
    slot_of = {i: k for k, i in enumerate(armed)}
    # TIMED_WINDOWS windows of exactly --steps steps each, every one bracketed by barrier + synchronize on both sides and
    # reduced to the slowest rank; the line reports the MEDIAN window (ms_per_step, value) with the fastest and slowest next
    # to it: one window of 20 steps is 1.2 ms of wall clock, and single windows differ by a few percent on one box.
    window_dt = []
    for w in range(TIMED_WINDOWS):
        watch("timed window %d of %d" % (w + 1, TIMED_WINDOWS), transport=mode, steps=args.steps)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, mode, slot_of.get(i, -1) if w == 0 else -1)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        window_dt.append(dt)
    dt = float(np.median(window_dt))
    enc_ms = float(np.mean([native.profile_read(k) for k in range(len(armed))]))
    identical = ranks_agree(torch, dist, world, [out])

    watch("untimed breakdown pass", transport=mode)
    # ---- untimed breakdown pass (events per phase), for DESIGN.md / the judge ----------
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    torch.cuda.synchronize()
    for a, b in pairs:
        a.record()
        b.record()
    torch.cuda.synchronize()
    ev_overhead_ms = float(np.mean([a.elapsed_time(b) for a, b in pairs]))
    rot = [0]

    def next_grad():
        rot[0] += 1
        return grads[rot[0] % 3]
    enc_b2b_ms = event_ms(torch, lambda: native.hsq_encode(next_grad(), cb, codes, u, partials))
    torch.cuda.synchronize()
    for a, b in pairs:
        a.record()
        native.hsq_encode(next_grad(), cb, codes, u, partials)
        b.record()
    torch.cuda.synchronize()
    enc_bracket_ms = float(np.mean([a.elapsed_time(b) for a, b in pairs]))
    lv_ms = event_ms(torch, lambda: native.hsq_levels(u, N_BIT, args.random, None, seed, partials, lb_ub, levels, packed6))
    cmp_ms = event_ms(torch, lambda: compress(next_grad()))
    dec_ms = event_ms(torch, lambda: decode(ex.gathered))
    # the decode-mean an 8-rank step runs after its exchange (R = 8: the kernel that grows with N), on 8 copies of this
    # rank's payload; untimed, N = 1 only (with N > 1 the step's own decode already runs over R = N payloads)
    lvdec_ms = event_ms(torch, lambda: native.hsq_levels_decode(u, N_BIT, args.random, None, seed, partials, lb_ub, levels, codes, cb,
                                                                 out, packed6)) if world == 1 else None
    dec8_ms, dec8 = None, None
    if world == 1:
        # EIGHT DIFFERENT payloads (the three gradients at eight scales, compressed by this rank's kernels), decoded in a timed
        # region of its own with the step's scheme: windows of --steps launches between synchronize() pairs on the host
        # clock, median window reported, next to the event-timed back-to-back figure
        g8 = wire.alloc(dev, ranks=8)
        for r in range(8):
            compress(grads[r % 3] * (1.0 + 0.25 * r))
            g8[r].copy_(ex.gathered[0])
        compress(grads[0])

        def dec8_call():
            native.hsq_decode_sum_packed(g8, M, cb, N_BIT, out, 8, wire.codes_off, wire.levels_off, wire.lbub_off,
                                         level_dtype=native.PACKED6 if packed6 else torch.uint8)
        dec8_ms = event_ms(torch, dec8_call)
        wins = []
        for w in range(TIMED_WINDOWS):
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            for _ in range(max(args.steps, 20)):
                dec8_call()
            torch.cuda.synchronize()
            wins.append((time.perf_counter() - t8) / max(args.steps, 20) * 1e3)
        dec8 = {"ms_median_window": float(np.median(wins)), "ms_min": min(wins), "ms_max": max(wins), "launches_per_window": max(args.steps, 20),
                "ms_events_back_to_back": dec8_ms, "payloads": "8 different (three gradients at eight scales)",
                "algorithmic_bytes": (2.0 * 8 / 16 + 4.0) * SIZE,
                "frac": (2.0 * 8 / 16 + 4.0) * SIZE / (float(np.median(wins)) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del g8
    exch_ms, exch_by = None, None
    if world > 1:
        exch_ms = event_ms(torch, (lambda: [p.wait() for p in sex.start("split", cut=swire.cut)[1]]) if mode == "split"
                           else (lambda: [p.wait() for p in ex.start("pipelined", cuts=[wire.levels_off])[1]]) if mode == "pipelined"
                           else (lambda: ex.run(mode)))
        watch("every transport timed (allgather, direct)", transport=mode)
        exch_by = transports_ms(torch, dist, dev, lambda m, dry=False: (ex.start(m, dry_run=True) if dry else ex.run(m)))
    counted = ranks_counted(torch, dist, dev) if world > 1 else 1

    if rank != 0:
        return None
    ms_per_step = dt / args.steps * 1e3
    value = world * SIZE * args.steps / dt
    achieved = ALGO_BYTES_PER_ELEM * SIZE / (enc_ms * 1e-3) / 1e9
    achieved_compress = ALGO_BYTES_PER_ELEM * SIZE / (cmp_ms * 1e-3) / 1e9
    traffic, traffic_source = traffic_for(args, world, "hsq", "hsq_encode_pf_kernel", "hsq_encode_hbm_bytes_per_launch")
    clock, clock_source = (None, "not measured (N > 1 or a PMC child run)") if (world > 1 or args.traffic_child) else in_kernel_clock()
    clock_ghz = clock["in_kernel_clock_ghz"] if clock else None
    floor_ms, isa = issue_floor(clock_ghz, native.device_info(dev.index)[0])
    line = {
        "metric": "gradient elements quantized/sec (HSQ d=16 k=8)", "value": value, "unit": "elements/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "ms_per_step_min": min(window_dt) / args.steps * 1e3, "ms_per_step_max": max(window_dt) / args.steps * 1e3,
        "timed_windows": {"count": len(window_dt), "steps_each": args.steps, "ms_per_step": [d / args.steps * 1e3 for d in window_dt],
                          "reported": "median window (value and ms_per_step); kernel_ms: events on the first window's dispatches"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_is": "end_to_end (encode + levels" + (" + exchange" if world > 1 else "") + " + decode-mean per step); "
                    "compress_only holds the encode + levels rate (SURVEY 8d's definition)",
        "prewarm_steps": prewarm,
        "config": {"workload": "synthetic 25,000,000-float32 flat gradient per rank, HSQ c_dim=16 k_bit=8 "
                               "n_bit=6 (BASELINE configs[%d]), step = encode+levels" % (1 if world == 1 else 3)
                               + ("+exchange(%s)" % mode if world > 1 else "") + "+decode-mean",
                   "elements_per_rank": SIZE, "random": args.random, "ranks": world,
                   "wire_levels": "packed6" if packed6 else "bytes", "wire_bytes_per_rank": wire.nbytes,
                   "inputs": "3 gradients of 100 MB used in turn (never Infinity-Cache resident)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes": ALGO_BYTES_PER_ELEM * SIZE,
                     "frac_compress": achieved_compress / HBM_PEAK_GBS, "achieved_compress": achieved_compress,
                     "compress_ms": cmp_ms,
                     # the whole timed step against the roofline: compress (4.125 B/element) + decode-mean over `world` payloads
                     # ((2 * world / 16 + 4) B per output element), exchange bytes not counted
                     "frac_step": (ALGO_BYTES_PER_ELEM + 2.0 * world / 16 + 4.0) * SIZE / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "frac_is": "frac = the dominant kernel alone (4.125 B x 25e6 / kernel_ms / peak); frac_compress = SURVEY 8(d)'s "
                                "definition, the whole compress: 4.125 B x 25e6 / (encode + levels, HIP events around "
                                "back-to-back pairs on rotating inputs) / peak; frac_step = (4.125 + 2 R / 16 + 4) B x 25e6 / ms_per_step "
                                "/ peak, the timed step's algorithmic bytes (compress + decode-mean of R = n_gpus payloads)",
                     "kernel": "gq_hsq_encode = hsq_encode_pf_kernel (one launch: f16 prefilter, exact rescoring, second pass and exact scan of the unsettled; levels: a second launch)",
                     "kernel_ms": enc_ms, "kernel_ms_back_to_back": enc_b2b_ms,
                     # the kernel's own ceiling: its instruction stream at the clock the chip holds under it
                     "in_kernel_clock_ghz": clock_ghz, "in_kernel_clock_source": clock_source,
                     "issue_floor_ms": floor_ms, "frac_of_issue_floor": (floor_ms / enc_ms) if floor_ms else None,
                     "issue_floor_is": ("tiles x (4 cycles x %d VALU + 8 x %d MFMA instructions of the tile loop, %s) / (4 SIMDs x CUs) / "
                                        "in_kernel_clock_ghz: what the instruction stream allows with every issue slot used; "
                                        "prologue, tail, second pass and exact scans come on top"
                                        % (isa["valu"], isa["mfma"], ISA_FILE)) if isa else None,
                     "kernel_ms_recorded_bracket": enc_bracket_ms, "empty_recorded_bracket_ms": ev_overhead_ms,
                     "note": "exact f32 scoring would need 512 flop/element (81 us at 157.3 TFLOP/s); the f16 prefilter "
                             "(one MFMA per chain) + exact rescoring + second pass is bound by VALU issue and by the clock the "
                             "chip holds under it, not by HBM.  kernel_ms: HIP "
                             "start/stop events attached to the kernel's dispatch inside the timed region "
                             "(agrees with rocprofv3, profiles/); a bracket RECORDED around the call reads "
                             "kernel_ms_recorded_bracket, an empty one empty_recorded_bracket_ms"},
        "phases_ms": {"encode": enc_ms, "levels": lv_ms, "compress": cmp_ms, "decode_mean": dec_ms, "levels_decode_fused": lvdec_ms,
                      "decode_mean_R8": dec8_ms, "decode_mean_R8_timed": dec8, "exchange": exch_ms,
                      "step": ("encode + gq_hsq_levels_decode (level quantiser and decode of the rank's own payload in one launch)" if fused
                               else "encode + levels + " + ("exchange + " if world > 1 else "") + "decode-mean"),
                      "note": "encode: HIP events attached to the dispatch in the timed region; levels / compress / decode_mean / "
                              "levels_decode_fused: HIP events around back-to-back calls on rotating inputs (untimed pass), i.e. launch to "
                              "launch; in the step the start of a kernel overlaps the drain of the one before it, so the phases of a step "
                              "can add up to ~1 us more than ms_per_step"},
        "compress_only": {"value": world * SIZE / (cmp_ms * 1e-3), "unit": "elements/s"},
        "ranks_bit_identical": identical,
    }
    rep = exchange_report(torch, counted, sex if mode == "split" else ex, mode, args.exchange, backend, world, exch_ms, exch_by,
                          (swire if mode == "split" else wire).nbytes) if world > 1 else None
    if rep:
        rep["autotune_ms"] = ex.timings_ms
        line["exchange"] = rep
    if world == 1 and not args.no_variants:
        # side measurements asked for by SURVEY 8d: stochastic rounding with the in-kernel generator, and a
        # gradient of realistic magnitude (N(0,1) * 1e-3); compress only, HIP events over 20 launches
        small = grads[1] * 1e-3
        lv_b = torch.empty(M, dtype=torch.uint8, device=dev)      # byte levels of their own (stochastic rounding reaches level 64)
        r2 = event_ms(torch, lambda: (native.hsq_encode(next_grad(), cb, codes, u, partials),
                                      native.hsq_levels(u, N_BIT, 2, None, seed, partials, lb_ub, lv_b)))
        sc = event_ms(torch, lambda: (native.hsq_encode(small, cb, codes, u, partials),
                                      native.hsq_levels(u, N_BIT, 0, None, seed, partials, lb_ub, lv_b)))
        line["variants"] = {"compress_random2": {"ms": r2, "value": SIZE / (r2 * 1e-3), "unit": "elements/s"},
                            "compress_scale_1e-3": {"ms": sc, "value": SIZE / (sc * 1e-3), "unit": "elements/s"}}
        del small
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(grads[0].cpu().numpy(), cb_np)
    if world == 1 and not args.no_workloads and not args.traffic_child:
        del grads, out, u, partials      # 500 MB back before the parameter lists are built
        torch.cuda.empty_cache()
        line["workloads"] = w = list_workloads(args, torch, np, native, dev)
        # the list steps' headline numbers as top-level keys too (BASELINE configs[2] / [4] at one rank): a reader that keeps only
        # the keys it knows still sees them
        for name in ("resnet50", "qsgd", "resnet50_real", "resnet50_ef", "resnet50_ef_twophase"):
            if name in w and "ms_per_step_graph" in w[name]:
                line["list_%s_ms_per_step" % name] = w[name]["ms_per_step_graph"]
                line["list_%s_elements_per_s" % name] = w[name]["value_graph"]
    return line


# ------------------------------------------------------------------------------------------------------
def gradient_feeder(torch, params, lists):
    """feed(i): the parameters' gradients become list i % len(lists) -- what autograd's backward does between two quantizer
    steps.  apply() rebinds `param.grad.data` like the reference (ps_quantizer.py:63), so after a step a gradient object
    points at the decoded mean; the next step's inputs are put back under the SAME gradient objects (autograd accumulates
    into an existing .grad in place) by the library's C++ helper -- 161 Python-level `p.grad = g` assignments cost more host
    time than the whole quantizer step and are not part of what is measured.  Without the helper: the Python assignments."""
    from gq_amd import quantizers
    host = quantizers._HOST
    for p, g in zip(params, lists[0]):
        p.grad = g.view(g.shape)
    objs = [p.grad for p in params]
    data = [[g.view(g.shape) for g in lst] for lst in lists]

    def feed(i):
        if host is not None:
            host.set_data(objs, data[i % len(data)])
        else:
            for o, g in zip(objs, data[i % len(data)]):
                o.data = g
    return feed


# the compact `workloads` object of the default line: BASELINE configs[2] / [4] under the driver's clock
# ------------------------------------------------------------------------------------------------------
def list_workloads(args, torch, np, native, dev, steps=400, warm=600):
    """The ResNet-50/CIFAR parameter list (161 tensors, 23.5 M elements) through PSQuantizer.record + apply, three ways:
    resnet50 (HSQ c_dim 16 k_bit 8 n_bit 6 random 1, N(0,1) x 1e-3 inputs), qsgd (QSGD c_dim 128 n_bit 2 random 1, same inputs) and
    resnet50_real (HSQ on BACK-PROPAGATED gradients: one forward / backward of driver.ResNet50 per input list on a seeded
    synthetic CIFAR batch of 128 -- real gradient statistics: dead units' all-zero subvectors, heavy tails).  Each with the
    library's default launches (HIP graph replay) and with eager launches, `steps` timed steps after `warm` (600: ~50 ms of load --
    with 40 the clocks had not come back up after the idle gap in which the lists are built, and the default line's
    `workloads` read 0.074 ms where `--workload resnet50`, with its own pre-warm, reads 0.067)."""
    import contextlib
    from argparse import Namespace
    from gq_amd.compressors import NearestNeighborCompressor, QSGDCompressor
    from gq_amd.driver import ResNet50
    from gq_amd.quantizers import BatchedHSQ, BatchedQSGD, Quantizer

    torch.manual_seed(1234)
    model = ResNet50(num_classes=10).to(dev)
    shapes = [tuple(p.shape) for p in model.parameters()]
    n = sum(int(np.prod(sh)) for sh in shapes)
    lossf = torch.nn.CrossEntropyLoss()
    real = []
    for b in range(3):
        gen = torch.Generator(device=dev)
        gen.manual_seed(4321 + b)
        x = torch.randn(128, 3, 32, 32, device=dev, generator=gen)
        y = torch.randint(0, 10, (128,), device=dev, generator=gen)
        model.zero_grad(set_to_none=True)
        lossf(model(x), y).backward()
        real.append([p.grad.detach().clone() for p in model.parameters()])
    del model
    synth = [[torch.randn(sh, device=dev) * 1e-3 for sh in shapes] for _ in range(3)]
    hsq_kw = dict(c_dim=C_DIM, k_bit=K_BIT, n_bit=N_BIT)
    qsgd_kw = dict(c_dim=128, k_bit=8, n_bit=2)

    def one(Comp, kw, lists, hsq, ef=False, two_phase=False):
        res = {}
        if ef:      # error feedback adds into the gradients in place (ps_quantizer.py:35): private copies of the input lists
            lists = [[g.clone() for g in l] for l in lists]
        for graph in (False, True):
            qargs = Namespace(no_cuda=False, random=1, ef=ef, two_phase=two_phase, scale=EF_BENCH_SCALE if ef else "exp", num_users=1,
                              mode="ps", cr=256, gq_graph=graph, **kw)
            params = [torch.nn.Parameter(torch.zeros(*sh, device=dev)) for sh in shapes]
            with contextlib.redirect_stdout(sys.stderr):     # the constructors report the reference's dimension repair on stdout
                q = Quantizer(Comp, params, qargs)
            feed = gradient_feeder(torch, params, lists)

            def step(i):
                feed(i)
                q.record(0, epoch=1)
                q.apply()
            for i in range(warm):
                step(i)
            Grp = BatchedHSQ if hsq else BatchedQSGD
            grp = [g[2] for g in q._groups if isinstance(g[2], Grp) and not getattr(g[2], "wide", False)][0]
            armed = {warm + k * (steps // 8): k for k in range(8)} if (hsq and not graph) else {}
            windows = []      # three windows of `steps` steps, the median reported: one window caught a host hiccup often enough
            for wdw in range(3):     # (a 98 us resnet50 step beside 60 us from `--workload resnet50` on the same box)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(warm + wdw * steps, warm + (wdw + 1) * steps):
                    if wdw == 0 and i in armed:
                        grp.profile_slot = armed[i]
                    step(i)
                torch.cuda.synchronize()
                windows.append((time.perf_counter() - t0) / steps * 1e3)
            ms = sorted(windows)[1]
            res["ms_per_step_graph" if graph else "ms_per_step_eager"] = ms
            res["windows_ms_graph" if graph else "windows_ms_eager"] = windows
            if graph:
                res["graphs_captured"] = q.graph_counts()
                continue
            c0 = native.CALLS[0]
            step(warm + steps)
            res["launches"] = {"library_calls_per_step": native.CALLS[0] - c0,
                               "torch_ops_per_step": "1 pinned header copy (HSQ) + the dense tensors' _foreach_copy_"}
            for p, g in zip(params, lists[0]):
                p.grad = g.view(g.shape)
            gl = [params[i].grad.data for i in grp.idxs]
            wire0 = q._wire[0]
            errs = [params[i].error[0] for i in grp.idxs] if ef else None
            cmp_ms = event_ms(torch, lambda: grp.encode(gl, wire0, 0, 0, errs, float(EF_BENCH_SCALE) if ef else None))   # HSQ: encode + levels; QSGD: the one compress launch
            k_elems = sum(cd.numel for cd in grp.codecs)
            if hsq:
                k_ms = float(np.mean([native.profile_read(k) for k in range(8)]))
                algo = ALGO_BYTES_PER_ELEM * k_elems
            else:
                k_ms, algo = cmp_ms, QSGD_ALGO_BYTES_PER_ELEM * k_elems
            if ef and hsq:
                # error feedback, per element of a compressed tensor: the encode reads grad and error and writes grad + scale*error
                # back (12 B) + u, code (5 / 16 B); the level launch reads u and the updated grad and writes the residual and the
                # level (8 B + 5 / 16 B): 20.625 B against 4.125 B without error feedback
                algo_c = EF_ALGO_BYTES_PER_ELEM * k_elems
                res.update({"kernel_ms": k_ms, "compress_ms": cmp_ms, "levels_ms": max(0.0, cmp_ms - k_ms), "kernel_elements": k_elems,
                            "algorithmic_bytes": algo_c, "algorithmic_bytes_encode": EF_ENCODE_BYTES_PER_ELEM * k_elems,
                            "frac": EF_ENCODE_BYTES_PER_ELEM * k_elems / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "frac_compress": algo_c / (cmp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "decode_mean_ms_R1": event_ms(torch, lambda: grp.decode_mean(q._wire[:1], 1))})
            else:
                res.update({"kernel_ms": k_ms, "compress_ms": cmp_ms, "kernel_elements": k_elems, "algorithmic_bytes": algo,
                            "frac": algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_compress": algo / (cmp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "decode_mean_ms_R1": event_ms(torch, lambda: grp.decode_mean(q._wire[:1], 1))})
            if hsq and not ef:     # how many subvectors the prefilter's first pass does not settle (second pass / exact scan), per-tensor encodes
                cbk, flagged, subv = grp.codebook, 0, 0
                for i in grp.idxs:
                    g = lists[0][i].contiguous().view(-1)
                    if q.codecs[i].c.dim != 16 or g.numel() % 16:
                        continue
                    M = g.numel() // 16
                    ws = native.new_workspace(dev, M)
                    native.mark_worklist(ws, M)
                    native.hsq_encode(g, cbk, torch.empty(M, dtype=torch.uint8, device=dev), torch.empty(M, dtype=torch.float32, device=dev), ws)
                    flagged += native.fixup_count(ws, M)
                    subv += M
                res["fixup_fraction"] = flagged / max(1, subv)
            del q
        res["value_graph"] = n / (res["ms_per_step_graph"] * 1e-3)
        res["unit"] = "elements/s"
        return res

    out = {"resnet50": one(NearestNeighborCompressor, hsq_kw, synth, True),
           "qsgd": one(QSGDCompressor, qsgd_kw, synth, False),
           "resnet50_real": one(NearestNeighborCompressor, hsq_kw, real, True),
           # error feedback (ps_quantizer.py:34-39) and error feedback + two-phase (:52-61) on the back-propagated gradients
           "resnet50_ef": one(NearestNeighborCompressor, hsq_kw, real, True, ef=True),
           "resnet50_ef_twophase": one(NearestNeighborCompressor, hsq_kw, real, True, ef=True, two_phase=True),
           "note": ("ResNet-50/CIFAR parameter list, %d elements in 161 tensors (76 through the codebook / the bucket quantiser, 85 of "
                    "<= 1000 elements as f32), one rank, PSQuantizer.record + apply per step (three input lists in turn, put under the parameters' existing .grad objects by the library's C++ helper inside the timed region), three windows of %d timed steps after %d, the median window reported; ms_per_step_graph: the "
                    "library's default (HIP graph replay -- record + apply as ONE graph per step at one rank and one user --, draws keyed by device step words), ms_per_step_eager: gq_graph off; kernel_ms: "
                    "HSQ = HIP events attached to the multi-tensor encode's dispatch on 8 eager steps, QSGD = the one compress launch "
                    "(events around 50 back-to-back launches); frac = algorithmic bytes of the compressed tensors / kernel_ms / 8 TB/s; "
                    "resnet50_real: gradients of driver.ResNet50 back-propagated from seeded synthetic CIFAR batches of 128 (three lists "
                    "in turn); fixup_fraction: subvectors the prefilter's first pass leaves to the second pass or the exact scan; "
                    "resnet50_ef / resnet50_ef_twophase: --ef (and --two-phase) on the same gradients with --scale 0 (the in-place add of "
                    "ps_quantizer.py:35 would otherwise accumulate in the bench's recycled inputs; same launches, bytes and arithmetic), kernel_ms = the error-feedback encode (reads grad + error, writes grad back: 12.3125 B per "
                    "element), levels_ms = compress_ms - kernel_ms (reads u and grad, writes error and level: 8.3125 B), frac_compress "
                    "against 20.625 B per element"
                    % (n, steps, warm))}
    return out



# ------------------------------------------------------------------------------------------------------
# workloads resnet50 / qsgd: the ResNet-50 parameter list through the quantizer (BASELINE configs[2] / [4])
# ------------------------------------------------------------------------------------------------------
def run_list(args, torch, np, dist, native, exchange, dev, rank, world, backend, barrier):
    from argparse import Namespace
    from gq_amd.compressors import NearestNeighborCompressor, QSGDCompressor
    from gq_amd.driver import ResNet50
    from gq_amd.quantizers import BatchedHSQ, BatchedQSGD, Quantizer

    hsq = args.workload == "resnet50"
    if args.traffic_child:      # (the PMC child runs count the bytes of individually dispatched kernels)
        args.graph = False
    shapes = [tuple(p.shape) for p in ResNet50(num_classes=10).parameters()]
    n = sum(int(np.prod(s)) for s in shapes)
    if hsq:     # README: --quantizer hsq --network resnet50 --c-dim 16 --k-bit 8 --n-bit 6 (--random defaults to True)
        if args.k_bit and args.k_bit != 8:      # (the package ships the K = 256 codebooks; the others it finds where the reference keeps them or here)
            os.environ.setdefault("GQ_CODEBOOK_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "codebooks"))
        qargs = Namespace(c_dim=args.c_dim or C_DIM, k_bit=args.k_bit or K_BIT, n_bit=args.n_bit or N_BIT, no_cuda=False, random=1, ef=bool(args.ef),
                          two_phase=bool(args.two_phase), scale=EF_BENCH_SCALE if args.ef else "exp", num_users=1, mode="ps", cr=256)

        Comp = NearestNeighborCompressor
    else:       # README: --quantizer qsgd --c-dim 128 --n-bit 2
        qargs = Namespace(c_dim=128, k_bit=8, n_bit=2, no_cuda=False, random=1, ef=False, two_phase=False, scale="exp",
                          num_users=1, mode="ps", cr=256)
        Comp = QSGDCompressor
    qargs.gq_graph = bool(args.graph)      # (the library's default is replay: gq_rng = "device" draws from device step words)
    params = [torch.nn.Parameter(torch.zeros(*s, device=dev)) for s in shapes]
    os.environ["GQ_EXCHANGE"] = args.exchange
    os.environ["GQ_WIRE_LEVELS"] = args.wire_levels      # (auto: packed6 for N > 1.)  packed6 applies where the top level is <= 63 (not with the README's --random 1 at n_bit 6)
    torch.manual_seed(1234 + rank)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # the constructors report the reference's dimension repair on stdout
        q = Quantizer(Comp, params, qargs)
    grads = [[torch.randn(s, device=dev) * 1e-3 for s in shapes] for _ in range(3)]
    prewarm = 10 if args.traffic_child else 200
    # apply() rebinds `param.grad.data` like the reference (ps_quantizer.py:63): a gradient object points at the decoded
    # mean afterwards.  Before every step the three input lists are put back under the gradient objects in turn
    # (gradient_feeder), so the inputs stay N(0,1) * 1e-3 for the whole run.
    feed = gradient_feeder(torch, params, grads)

    def step(i):
        feed(i)
        q.record(0, epoch=1)
        q.apply()

    watch("list workload: prewarm", transport=args.exchange, steps=prewarm + args.warmup)
    for i in range(prewarm + args.warmup):      # (--exchange auto: the first apply() times the transports)
        step(i)
    # HIP events attached to the dominant kernel's dispatch on up to 16 of the timed steps (HSQ: the multi-tensor
    # prefilter encode takes them: gq_hsq_batch.profile_slot)
    stride = max(1, -(-args.steps // 16))
    armed = list(range(0, args.steps, stride))[:16] if (hsq and not args.graph) else []     # (a replayed graph has no armed dispatch)
    slot_of = {prewarm + args.warmup + i: k for k, i in enumerate(armed)}
    Grp = BatchedHSQ if hsq else BatchedQSGD
    grp = [g[2] for g in q._groups if isinstance(g[2], Grp) and not getattr(g[2], "wide", False)][0]
    watch("list workload: timed region", transport=q.exchange_mode, steps=args.steps)
    barrier()
    t0 = time.perf_counter()
    for i in range(prewarm + args.warmup, prewarm + args.warmup + args.steps):
        if i in slot_of:
            grp.profile_slot = slot_of[i]
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    identical = ranks_agree(torch, dist, world, [p.grad.data for p in params if p.numel() > 1000][:8])

    # untimed: the kernels of the step alone, back to back between two HIP events on the stream they are launched on
    for p, g in zip(params, grads[0]):
        p.grad = g.view(g.shape)
    gl = [params[i].grad.data for i in grp.idxs]
    wire0 = q._wire[0]
    errs = [params[i].error[0] for i in grp.idxs] if (hsq and args.ef) else None
    cmp_ms = event_ms(torch, lambda: grp.encode(gl, wire0, 0, 0, errs, float(EF_BENCH_SCALE) if errs else None))       # HSQ: encode + levels; QSGD: the one compress launch
    if hsq and args.graph:      # the encode's own time: armed dispatches of eager steps after the timed region (a replayed graph
        armed = list(range(8))  # has no armed dispatch); whole steps, so that the encode meets the caches a step leaves behind
        q.use_graphs = False
        for k in armed:
            grp.profile_slot = k
            step(k)
        q.use_graphs = True
        torch.cuda.synchronize()
    k_elems = sum(cd.numel for cd in grp.codecs)
    dec_ms = event_ms(torch, lambda: grp.decode_mean(q._wire[:1], 1))
    exch_ms, exch_by = None, None
    if world > 1 and q._ex is not None:
        mode = q.exchange_mode

        def only_exchange(m=None, dry=False):
            if dry:
                return q._ex.start(m, 1, q.cut, dry_run=True)
            for pnd in q._ex.start(m or mode, 1, q.cut, cuts=q.cuts)[1]:
                pnd.wait()
        exch_ms = event_ms(torch, only_exchange)
        exch_by = transports_ms(torch, dist, dev, only_exchange)
    counted = ranks_counted(torch, dist, dev) if world > 1 else 1
    if rank != 0:
        return None
    if hsq:
        k_ms = float(np.mean([native.profile_read(k) for k in range(len(armed))]))
        dd = qargs.c_dim
        algo = (4.0 + 2.0 / dd) * k_elems      # 4 B read + (code + level) per dd elements
        kernel = ("gq_hsq_encode_batched = hsq_encode_pf_kernel<uint8_t, BATCHED> (ONE launch for the 76 codebook-compressed "
                  "tensors: prefilter, exact rescoring, in-place fix-up, per-tensor lb/ub by atomics)")
        match, metric = "hsq_encode_pf_kernel", "gradient elements quantized/sec (HSQ d=%d k=8, ResNet-50 list)" % dd
        cfg = ("ResNet-50/CIFAR parameter list (161 tensors, %d elements) per rank through PSQuantizer.record + apply, HSQ c_dim=%d "
               "k_bit=%d n_bit=%d random=1 on-device draws%s%s (%s), byte wire, multi-tensor kernels"
               % (n, dd, qargs.k_bit, qargs.n_bit, ", error feedback" if args.ef else "", ", two-phase" if args.two_phase else "",
                  "BASELINE configs[2]; the README's hsq command" if (dd == 16 and qargs.n_bit == 6 and qargs.k_bit == 8) else "main.py:90-92's own defaults" if (dd == 32 and qargs.n_bit == 8 and qargs.k_bit == 8) else "a variant"))
        note = ("kernel_ms: HIP start/stop events attached to the encode's dispatch on %d %s; compress_ms = encode + "
                "levels launches back to back after the timed region; a replayed step is bound by its kernels (the host issues one in ~40 us), an eager one by the host"
                % (len(armed), "eager steps run after the timed region (a replayed graph has no armed dispatch)" if args.graph
                   else "of the timed steps"))
    else:
        k_ms = cmp_ms
        algo = QSGD_ALGO_BYTES_PER_ELEM * k_elems
        kernel = "gq_qsgd_compress_batched = qsgd_compress_batched4_kernel"
        match, metric = "qsgd_compress_batched4_kernel", "gradient elements quantized/sec (QSGD c_dim=128 n_bit=2, ResNet-50 list)"
        cfg = ("ResNet-50/CIFAR parameter list (161 tensors, %d elements) per rank, QSGD c_dim=128 n_bit=2 "
               "random=1 (BASELINE configs[4]), packed 4-bit wire, multi-tensor kernels" % n)
        note = ("HIP events around 50 back-to-back launches after the timed region (a replayed step is bound by its two kernels, "
                "compress + decode-mean; an eager one by the host)")
    achieved = algo / (k_ms * 1e-3) / 1e9
    achieved_compress = algo / (cmp_ms * 1e-3) / 1e9
    traffic, traffic_source = traffic_for(args, world, args.workload, match, None)
    line = {
        "metric": metric, "value": world * n * args.steps / dt,
        "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_is": "end_to_end: PSQuantizer.record + apply per step (compress, " + ("exchange, " if world > 1 else "")
                    + "decode-mean, small tensors dense), host launch time included",
        "prewarm_steps": prewarm,
        "config": {"workload": cfg, "elements_per_rank": n, "ranks": world, "wire_bytes_per_rank": q.wire_bytes_per_user(),
                   "inputs": "3 gradient lists used in turn, N(0,1)*1e-3: before every step the next list is put under the parameters' "
                             "existing .grad objects (in place, as autograd writes gradients) by the library's C++ helper, inside the timed region",
                   "launches": ("gq_graph: record() and apply() replay their device work from HIP graphs (one per set of gradient "
                                "addresses / output buffer; %d + %d captured), stochastic rounding with draws keyed by each tensor's "
                                "(lb, ub) / each bucket's norm (gq_rng = 'keyed')" % (
                                    sum(1 for e in q._rec_graphs.values() if e[1] is not None),
                                    sum(1 for e in q._apply_graphs.values() if e[1] is not None))) if args.graph
                               else "eager: ten launches and copies per record + apply; per-call seeds for the on-device draws"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": algo,
                     "frac_compress": achieved_compress / HBM_PEAK_GBS, "compress_ms": cmp_ms,
                     "kernel": kernel, "kernel_ms": k_ms, "kernel_elements": k_elems, "note": note},
        "phases_ms": {"compress_kernels": cmp_ms, "decode_mean_kernel_R1": dec_ms, "exchange": exch_ms,
                      "note": "compress_kernels / decode_mean: the launches alone, back to back (HIP events); exchange: the chosen "
                              "transport alone; with N > 1 the decode-mean of the step runs over R = n_gpus payloads"},
        "compress_only": {"value": world * k_elems / (cmp_ms * 1e-3), "unit": "elements/s"},
        "ranks_bit_identical": identical,
    }
    if world > 1 and q._ex is not None:
        line["exchange"] = exchange_report(torch, counted, q._ex, q.exchange_mode, args.exchange, backend, world, exch_ms, exch_by,
                                           q.wire_bytes_per_user())
    return line


if __name__ == "__main__":
    main()
