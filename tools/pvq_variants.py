"""Diagnostic builds of the one-sweep PVQ kernel (pvq.hip, PVQ_DIAG bits) for tools/pvq_ab.py -- never shipped:
    python tools/pvq_variants.py diag1 diag3 diag7 ...      # -> tools/exp/libgq_pvq_<name>.so each
  diagN   -DPVQ_DIAG=N: 1 = the walk reduced to one codeword, 2 = no boundary sums, 4 = no l1 chain, 8 = no swaps
  any other name=flags pair builds the shipped source with those flags:  bpc2="-DPVQ_X=1" """
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gradient-quantization_amd", "csrc", "pvq.hip")
if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tools", "exp"), exist_ok=True)
    for arg in sys.argv[1:]:
        name, _, flags = arg.partition("=")
        extra = flags.split() if flags else ["-DPVQ_DIAG=%d" % int(name[4:])]
        env = dict(os.environ, VARIANT_OF="pvq")
        subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_variant.sh"), os.path.join(ROOT, "tools", "exp", "libgq_pvq_%s.so" % name), SRC,
                               "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + extra, env=env, stderr=subprocess.DEVNULL)
