#!/usr/bin/env python3
"""Build libgq_hsq.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python gradient-quantization_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU
box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgq_hsq.so")
SOURCES = ["gq_common.hip", "gq_api.hip", "hsq_encode.hip", "hsq_encode_pf.hip", "hsq_encode_pfd.hip", "hsq_levels.hip", "hsq_batched.hip", "hsq_decode.hip", "qsgd.hip", "qsgd_batched.hip", "qsgd_wide.hip", "pvq.hip"]
# -ffp-contract=off: the reference's elementwise ops are separately rounded; hipcc's
# default ("fast") would fuse the decode's mul/add and the level quantiser's sub/div.
# -packed-fp32-ops (target feature off): no v_pk_{fma,mul,add}_f32.  One instantiation of the encode kernel came out
# with a packed FMA whose destination pair was also its multiplicand pair, read crosswise through op_sel, and on
# the MI355X one result in ~1e5 (low half, lanes 48-63 only) was wrong (tools/fuzz_batched.py found it; see
# hsq_pf_common.hpp).  The compiler emits the same crosswise in-place form for `vector * scalar` in the decode
# kernels; rather than trust them, the library is built without packed f32 (decode R=8: see DESIGN.md).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
         "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


HOST_EXT = os.path.join(HERE, "gq_amd", "_gq_host.so")


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(os.path.join(HERE, "libgq_hsq_clock.so")) or not os.path.exists(HOST_EXT):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(HOST_EXT))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "gq_hsq.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


# per-file extra flags
EXTRA = {
    # lets fmaxf on MFMA outputs lower to v_max3_f32 without canonicalising v_max x,x,x (see the file)
    "hsq_encode_pf.hip": ["-fno-honor-nans"],
    # + MFMA results straight into VGPRs: with 512 registers available (d = 32 runs one wave per SIMD) the
    # compiler otherwise accumulates in AGPRs and pays 16 v_accvgpr_read per chain (69 -> 61 us)
    "hsq_encode_pfd.hip": ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    # every score of the f32 MFMA is read by the VALU (two sequential sums per subvector): keep them out of the AGPRs
    "pvq.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
}


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"]
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc()] + compile_flags + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd), obj))
    objs = []
    for cmd, p, obj in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
        objs.append(obj)
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(link))
    host = None
    try:
        host = build_host_ext(verbose, wait=False)      # g++ against libtorch: ~25 s, next to the link and the clock twin
    except Exception as e:      # (no libtorch headers, no g++): the helper is optional, quantizers.py walks in Python without it
        print("build.py: host helper not built (%s); gq_amd falls back to its Python walks" % (e,), file=sys.stderr)
    try:
        subprocess.check_call(link)
        build_clock_lib(objs, verbose)
    finally:
        if host is not None and host[1].wait() != 0:     # (always reaped, also when the link above raised)
            print("build.py: host helper failed to compile (exit %d); gq_amd falls back to its Python walks" % host[1].returncode,
                  file=sys.stderr)
    return LIB


def build_host_ext(verbose=False, wait=True):
    """gq_amd/_gq_host.so: the CPython helper of the quantizer's host loop (csrc/host_ext.cpp; no device code)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
           + ["-I" + d for d in ce.include_paths() + [sysconfig.get_paths()["include"]]]
           + [os.path.join(CSRC, "host_ext.cpp"), "-o", HOST_EXT]
           + ["-L" + d for d in ce.library_paths()] + ["-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10"]
           + ["-Wl,-rpath," + d for d in ce.library_paths()])
    if verbose:
        print(" ".join(cmd))
    p = subprocess.Popen(cmd)
    if wait and p.wait() != 0:
        raise subprocess.CalledProcessError(p.returncode, cmd)
    return cmd, p


CLOCK_LIB = os.path.join(HERE, "libgq_hsq_clock.so")


def build_clock_lib(objs=None, verbose=False, source=None, out=None):
    """DIAGNOSTIC twin of the library: the same objects, with hsq_encode_pf.hip compiled with -DGQ_PF_STAMPS (s_memtime /
    s_memrealtime stamps around the phases of a tile; see the macro's comment in the file).  Never loaded by the product:
    `bench.py` runs it in a child process to read the in-kernel clock (roofline.in_kernel_clock_ghz), tools/stamp_read.py
    prints the whole breakdown.  `source` / `out`: a variant of the file (tools/stamp_build.py)."""
    objdir = os.path.join(HERE, "build")
    if objs is None:
        objs = [os.path.join(objdir, f.replace(".hip", ".o")) for f in SOURCES]
    src = source or os.path.join(CSRC, "hsq_encode_pf.hip")
    obj = os.path.join(objdir, "stamps", "hsq_encode_pf_stamps.o" if out is None else os.path.basename(out) + ".o")   # (a directory of its own: build/*.o are the product's objects)
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    flags = [f for f in FLAGS if f != "-shared"] + EXTRA.get("hsq_encode_pf.hip", [])
    cmd = [hipcc()] + flags + ["-DGQ_PF_STAMPS", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, stderr=None if verbose else subprocess.DEVNULL)
    others = [o for o in objs if os.path.basename(o) != "hsq_encode_pf.o"]
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or CLOCK_LIB, obj] + others)
    return out or CLOCK_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
