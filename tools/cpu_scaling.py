"""Thread scaling of the CPU oracle's HSQ compress on this box's host (what bench.py's cpu_baseline sees)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
cb = np.load(os.path.join(ROOT, "tests", "golden", "codebook_d16_k256_normalized.npy"))
x = np.random.RandomState(0).standard_normal(25_000_000).astype(np.float32)
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > os.cpu_count():
        break
    oracle.set_num_threads(th)
    oracle.hsq_compress(x[:16 * 100000], cb, 6, 0)
    best_e = best_c = 1e9
    for _ in range(3):
        t = time.perf_counter(); oracle.hsq_encode(x, cb); best_e = min(best_e, time.perf_counter() - t)
        t = time.perf_counter(); oracle.hsq_compress(x, cb, 6, 0); best_c = min(best_c, time.perf_counter() - t)
    print("threads %3d  encode %8.1f M elements/s  compress %8.1f M elements/s" % (th, 25 / best_e, 25 / best_c))
