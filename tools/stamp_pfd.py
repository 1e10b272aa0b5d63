"""In-kernel stamps for hsq_encode_pfd_kernel (d = 8 / 32 prefilter encode), like stamp_build.py / stamp_read.py for d = 16:
    python tools/stamp_pfd.py                                                  # build container -> tools/exp/libgq_pfdstamp.so
    GQ_LIB_PATH=tools/exp/libgq_pfdstamp.so GQ_AB_D=32 python tools/stamp_pfd.py read      # on the GPU box
Cycles per phase of a tile, in-kernel clock, prologue length, loop-end skew.  Never shipped, never timed."""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["loop top: prefetch issue", "16 chains (MFMA + keys)", "tracker merge", "own-subvector swaps", "exact rescoring + safety",
         "next-tile bf16 split", "fix-up (rare) + stores + draw"]
NP = len(NAMES)

def build():
    SRC = os.path.join(ROOT, "gradient-quantization_amd", "csrc")
    TMP = "/tmp/gq_pfdstamp_src"
    shutil.rmtree(TMP, ignore_errors=True)
    shutil.copytree(SRC, TMP)
    p = os.path.join(TMP, "hsq_encode_pfd.hip")
    s = open(p).read()
    stamp = ('        __builtin_amdgcn_sched_barrier(0);\n'
             '        { unsigned long long ts_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); '
             '__builtin_amdgcn_sched_barrier(0); stamp_acc[ID] += (ts_ - ts_prev); ts_prev = ts_; }\n')
    def rep(old, new):
        nonlocal s
        assert old in s, old
        s = s.replace(old, new, 1)
    rep("    const int npages = PAGED ? a.npages : 1;\n", "    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();\n    const int npages = PAGED ? a.npages : 1;\n")
    rep("    while (t < tile_end) {\n        const int64_t tnn = BATCHED ? draw() : 0;",
        "    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();\n    unsigned long long ntl = 0;\n"
        "    unsigned long long stamp_acc[%d] = {%s}; unsigned long long ts_prev; "
        "asm volatile(\"s_memtime %%0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(ts_prev) :: \"memory\");\n"
        "    while (t < tile_end) {\n        const int64_t tnn = BATCHED ? draw() : 0;" % (NP, ",".join("0" * NP)))
    for i, marker in enumerate(["        // ---- prefilter: 16 chains in the order",
                                "        // ---- per block: merge the two trackers",
                                "        // ---- this lane's own full subvector",
                                "        // ---- exact rescoring of the better of the two halves",
                                "        // consume the prefetched tile BEFORE this tile's stores",
                                "        // ---- exact fix-up, in place and wave-wide"]):
        rep(marker, stamp.replace("ID", str(i)) + marker)
    rep("        ti = tin;\n        t = tn;\n        tn = BATCHED ? tnn : draw();\n    }\n    if (BATCHED) {\n        flush_minmax();\n        return;",
        "        ti = tin;\n        t = tn;\n        tn = BATCHED ? tnn : draw();\n" + stamp.replace("ID", str(NP - 1)) + "        ++ntl;\n    }\n"
        "    { const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();\n"
        "      if (!BATCHED && !PAGED && lane == 0 && (int)blockIdx.x < (32768 / (%d * WAVES) < 256 ? 32768 / (%d * WAVES) : 256)) {\n"
        "          unsigned long long *o = reinterpret_cast<unsigned long long *>(ws_worklist(ws) + (M - 65536)) + (blockIdx.x * WAVES + wave) * %d;\n"
        "          for (int i = 0; i < %d; ++i) o[i] = stamp_acc[i];\n"
        "          o[%d] = rt_entry; o[%d] = rt0; o[%d] = rt1; o[%d] = ntl; } }\n    if (BATCHED) {\n        flush_minmax();\n        return;"
        % (NP + 4, NP + 4, NP + 4, NP, NP, NP + 1, NP + 2, NP + 3))
    rep("if (!BATCHED && lane == 0) worklist[", "if (false) worklist[")
    open(p, "w").write(s)
    out = os.path.join(ROOT, "tools", "exp")
    os.makedirs(out, exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared", "-std=c++17",
           "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
           "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + TMP, "-o", os.path.join(out, "libgq_pfdstamp.so")] + sorted(glob.glob(TMP + "/*.hip"))
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    print(os.path.join(out, "libgq_pfdstamp.so"))

def read():
    sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
    import torch, numpy as np
    from gq_amd import native
    from gq_amd.codebook import load_codebook
    dev = torch.device("cuda:0")
    D = int(os.environ.get("GQ_AB_D", "32"))
    W = 8 if D == 32 else 12
    cb = torch.from_numpy(load_codebook(D, 256)).to(dev)
    torch.manual_seed(1234)
    g = torch.randn(25_000_000, device=dev)
    M = g.numel() // D
    codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = native.new_workspace(dev, M)
    for _ in range(3):
        native.hsq_encode(g, cb, codes, u, ws, impl=4)
    torch.cuda.synchronize()
    wl = ws[native.WS_LOG_FIRST:native.WS_LOG_FIRST + M]
    n = NP + 4
    nblk = min(256, 32768 // (n * W))
    raw = wl[M - 65536:M - 65536 + nblk * W * n * 2].contiguous().view(torch.int64).view(-1, n).cpu().numpy().astype(np.float64)
    seg, entry, rt0, rt1, tiles_w = raw[:, :NP], raw[:, NP], raw[:, NP + 1], raw[:, NP + 2], raw[:, NP + 3]
    tiles = tiles_w.mean()
    cyc = seg.sum(1).mean()
    loop_us = (rt1 - rt0).mean() / 100
    print("d = %d: loop %.0f shader cycles per wave in %.1f us -> in-kernel clock %.2f GHz; %.2f tiles per wave, %.0f cycles per tile per wave"
          % (D, cyc, loop_us, cyc / loop_us / 1e3, tiles, cyc / tiles))
    for nme, v in zip(NAMES, seg.mean(0)):
        print("  %-32s %6.0f cycles/tile  %5.1f %%" % (nme, v / tiles, 100 * v / cyc))
    print("prologue per wave %.1f us (min %.1f, max %.1f); first entry -> last loop end %.1f us; loop-end skew %.1f us"
          % ((rt0 - entry).mean() / 100, (rt0 - entry).min() / 100, (rt0 - entry).max() / 100, (rt1.max() - entry.min()) / 100, (rt1.max() - rt1.min()) / 100))
    print("entry spread %.1f us; loop end: mean %.1f us, min %.1f, max %.1f after the first entry"
          % ((entry.max() - entry.min()) / 100, ((rt1 - entry.min()) / 100).mean(), ((rt1 - entry.min()) / 100).min(), ((rt1 - entry.min()) / 100).max()))

if __name__ == "__main__":
    read() if sys.argv[1:] == ["read"] else build()
