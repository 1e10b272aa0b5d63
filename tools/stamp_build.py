"""Diagnostic build of the prefilter kernel with s_memtime / s_memrealtime stamps around the phases
of a tile (cdna_hip_programming.md section 7, in-kernel stamps).  Never shipped, never timed:
    python tools/stamp_build.py           # -> tools/exp/libgq_stamp.so
    GQ_LIB_PATH=tools/exp/libgq_stamp.so python tools/stamp_read.py      (on the GPU box)
The stamps are written behind the worklist, which no other code of the kernel reads."""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gradient-quantization_amd", "csrc")
TMP = "/tmp/gq_stamp_src"
shutil.rmtree(TMP, ignore_errors=True)
shutil.copytree(SRC, TMP)
p = os.path.join(TMP, "hsq_encode_pf.hip")
s = open(p).read()
stamp = ('        __builtin_amdgcn_sched_barrier(0);\n'
         '        { unsigned long long ts_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(ts_) :: "memory"); '
         '__builtin_amdgcn_sched_barrier(0); stamp_acc[ID] += (ts_ - ts_prev); ts_prev = ts_; }\n')
def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
rep("    const float *__restrict__ cb = a.cb;\n    float *__restrict__ ws = a.ws;\n    float *__restrict__ u = a.u;\n    const int64_t M = a.M;\n    // f32 codebook",
    "    const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();\n    const float *__restrict__ cb = a.cb;\n    float *__restrict__ ws = a.ws;\n    float *__restrict__ u = a.u;\n    const int64_t M = a.M;\n    // f32 codebook")
rep("    while (t < tile_end) {\n        // single tensor: tn was drawn at the end of the previous tile; batched: a whole tile ago",
    "    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();\n    unsigned long long ntl = 0;\n    unsigned long long stamp_acc[6] = {0,0,0,0,0,0}; unsigned long long ts_prev; "
    "asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(ts_prev) :: \"memory\");\n    while (t < tile_end) {\n        // single tensor: tn was drawn at the end of the previous tile; batched: a whole tile ago")
for i, marker in enumerate(["        // ---- prefilter: 16 (block, row block) chains",
                            "        // ---- per block: merge the two trackers",
                            "        // ---- exact rescoring of the better of the two",
                            "        // Consume the prefetched tile (convert it to the next B fragments)",
                            "        // ---- the few subvectors the bound could not settle"]):
    rep(marker, stamp.replace("ID", str(i)) + marker)
rep("        tn = BATCHED ? tnn : draw();\n        sigma_t = sigma_n;\n    }\n    if (BATCHED) flush_minmax();",
    "        tn = BATCHED ? tnn : draw();\n        sigma_t = sigma_n;\n" + stamp.replace("ID", "5") + "        ++ntl;\n    }\n"
    "    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();\n    if (BATCHED) flush_minmax();")
rep("    auto scan4 = [&](int first, int n) {\n", "    unsigned long long ndrained = 0;\n    auto scan4 = [&](int first, int n) {\n        ndrained += 1000 * n;\n")
rep("    auto second_pass = [&](int first, int n) {\n", "    auto second_pass = [&](int first, int n) {\n        ndrained += n;\n")
rep("    if (BATCHED) return;\n",
    "    { const unsigned long long rt2 = __builtin_amdgcn_s_memrealtime();\n"
    "      if (lane == 0 && blockIdx.x < 256) {\n"
    "          unsigned long long *o = reinterpret_cast<unsigned long long *>(ws_worklist(ws) + (M - 65536)) + (blockIdx.x * 8 + wave) * 12;\n"
    "          for (int i = 0; i < 6; ++i) o[i] = stamp_acc[i];\n"
    "          o[6] = rt_entry; o[7] = rt0; o[8] = rt1; o[9] = ntl; o[10] = rt2; o[11] = ndrained; } }\n    if (BATCHED) return;\n")
# diagnostics marks off: the fix-up log shares the workspace region the stamps are written to
s = s.replace("if (!BATCHED && lane == 0) worklist[", "if (false) worklist[").replace("                    worklist[m[2]] = (int)m[2];", "                    ;").replace("                worklist[mt[2]] = (int)mt[2];", "                ;")
OUT_NAME = "libgq_stamp.so"
if os.environ.get("GQ_STAMP_VARIANT"):
    # the stamps on one of tools/pf_variants.py's diagnostic builds (GQ_STAMP_VARIANT=zb4: the DVFS diagnostic of
    # MI355X_MICROARCH.md 'DVFS give-back' item 6 -- cycles per tile stay, the clock the chip then holds is what the stamps read)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pf_variants
    s = pf_variants.VARIANTS[os.environ["GQ_STAMP_VARIANT"]](s)
    OUT_NAME = "libgq_stamp_%s.so" % os.environ["GQ_STAMP_VARIANT"]
open(p, "w").write(s)
out = os.path.join(ROOT, "tools", "exp")
os.makedirs(out, exist_ok=True)
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared", "-std=c++17",
       "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + TMP, "-o", os.path.join(out, OUT_NAME)] + sorted(glob.glob(TMP + "/*.hip"))
subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
print(os.path.join(out, OUT_NAME))
