"""Diagnostic build of the prefilter kernel with s_memtime / s_memrealtime stamps around the phases of a tile
(cdna_hip_programming.md section 7, in-kernel stamps): hsq_encode_pf.hip compiled with -DGQ_PF_STAMPS (the macro in the file).
The default build (gradient-quantization_amd/build.py) already makes gradient-quantization_amd/libgq_hsq_clock.so; this tool
makes the same for one of tools/pf_variants.py's experimental sources.  Never shipped, never timed:
    python tools/stamp_build.py                          # -> tools/exp/libgq_stamp.so (a copy of the clock library)
    GQ_STAMP_VARIANT=zb2 python tools/stamp_build.py     # -> tools/exp/libgq_stamp_zb2.so
    GQ_LIB_PATH=tools/exp/libgq_stamp.so python tools/stamp_read.py      (on the GPU box)
The stamps are written behind the diagnostics log of the workspace, which the stamped build does not write."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gq_build", os.path.join(ROOT, "gradient-quantization_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
b.build()      # the objects the diagnostic library links against
out_dir = os.path.join(ROOT, "tools", "exp")
os.makedirs(out_dir, exist_ok=True)
variant = os.environ.get("GQ_STAMP_VARIANT")
src = None
name = "libgq_stamp.so"
if variant:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pf_variants
    os.makedirs("/tmp/gq_pfv", exist_ok=True)
    src = "/tmp/gq_pfv/hsq_encode_pf_stamp_%s.hip" % variant
    open(src, "w").write(pf_variants.VARIANTS[variant](pf_variants.SRC))
    name = "libgq_stamp_%s.so" % variant
print(b.build_clock_lib(source=src, out=os.path.join(out_dir, name)))
