"""Diagnostic build of the prefilter kernel with s_memrealtime stamps (100 MHz) through its PROLOGUE
(codebook -> LDS, A fragments, barriers, first tile) -- never shipped, never timed:
    python tools/stamp_prologue.py                 # -> tools/exp/libgq_pstamp.so
    GQ_LIB_PATH=tools/exp/libgq_pstamp.so python tools/stamp_prologue.py read      (on the GPU box)"""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1:] == ["read"]:
    sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
    import torch, numpy as np
    from gq_amd import native
    from gq_amd.codebook import load_codebook
    dev = torch.device("cuda:0")
    cb = torch.from_numpy(load_codebook(16, 256)).to(dev)
    torch.manual_seed(1234)
    g = torch.randn(25_000_000, device=dev)
    M = g.numel() // 16
    codes = torch.empty(M, dtype=torch.uint8, device=dev); u = torch.empty(M, dtype=torch.float32, device=dev)
    ws = native.new_workspace(dev, M)
    for _ in range(200):
        native.hsq_encode(g, cb, codes, u, ws, impl=4)
    torch.cuda.synchronize()
    wl = ws[2 * native.GQ_MAX_PARTIALS + 4:2 * native.GQ_MAX_PARTIALS + 4 + M]
    raw = wl[M - 65536:M - 65536 + 256 * 8 * 8 * 2].contiguous().view(torch.int64).view(-1, 8).cpu().numpy().astype(np.float64)
    t0 = raw[:, 0].min()
    names = ["entry (after first wave's entry)", "all prologue loads issued", "LDS image + own row block's fragments written", "barrier passed",
             "fragments read back, ||c||_1", "first tile split (loop starts)", "loop ends", "end of wave (partials written)"]
    for i, n in enumerate(names):
        c = (raw[:, i] - t0) / 100
        print("  %-42s mean %6.2f us  min %6.2f  max %6.2f" % (n, c.mean(), c.min(), c.max()))
    sys.exit(0)
SRC = os.path.join(ROOT, "gradient-quantization_amd", "csrc")
TMP = "/tmp/gq_pstamp_src"
shutil.rmtree(TMP, ignore_errors=True)
shutil.copytree(SRC, TMP)
p = os.path.join(TMP, "hsq_encode_pf.hip")
s = open(p).read()
def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
ST = '    __builtin_amdgcn_sched_barrier(0); pst[%d] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0);\n'
rep("    const float *__restrict__ cb = a.cb;\n    float *__restrict__ ws = a.ws;",
    "    unsigned long long pst[8];\n" + ST % 0 + "    const float *__restrict__ cb = a.cb;\n    float *__restrict__ ws = a.ws;")
rep("    if (threadIdx.x == 0) s_next = PF_WAVES;", ST % 1 + "    if (threadIdx.x == 0) s_next = PF_WAVES;")           # all loads issued
rep("    __syncthreads();\n    // A fragments of v_mfma", ST % 2 + "    __syncthreads();\n" + ST % 3 + "    // A fragments of v_mfma")   # LDS written / barrier passed
rep("    int64_t tn = draw();                // the tile after", ST % 4 + "    int64_t tn = draw();                // the tile after")   # fragments read
rep("    while (t < tile_end) {\n        // single tensor: tn was drawn", ST % 5 + "    while (t < tile_end) {\n        // single tensor: tn was drawn")
rep("    write_minmax_partials<PF_WAVES>(lmin, lmax, ws, sawnan);",
    ST % 6 + "    write_minmax_partials<PF_WAVES>(lmin, lmax, ws, sawnan);\n" + ST % 7 +
    "    if (lane == 0 && blockIdx.x < 256) {\n"
    "        unsigned long long *o = reinterpret_cast<unsigned long long *>(ws_worklist(ws) + (M - 65536)) + (blockIdx.x * 8 + wave) * 8;\n"
    "        for (int i = 0; i < 8; ++i) o[i] = pst[i];\n    }")
# diagnostics marks off: the fix-up log shares the workspace region the stamps are written to
rep("if (!BATCHED && lane == 0) worklist[", "if (false) worklist[")
open(p, "w").write(s)
out = os.path.join(ROOT, "tools", "exp")
os.makedirs(out, exist_ok=True)
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared", "-std=c++17",
       "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
       "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + TMP, "-o", os.path.join(out, "libgq_pstamp.so")] + sorted(glob.glob(TMP + "/*.hip"))
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
print(os.path.join(out, "libgq_pstamp.so"))
