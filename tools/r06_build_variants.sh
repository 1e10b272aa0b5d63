#!/bin/bash
# round 6 A/B builds of the d16 prefilter encode (profiles/r06_encode_ab.txt): tools/exp/libgq_{mask,pkfma,pkfma_mask}.so against the
# in-tree objects.  The variant code (-DGQ_PF_N2NORM: n2 as the plain squared norm, the product since; -DGQ_PF_PKFMA: the rescoring as
# v_pk_fma_f32) lived in the sources of commit 33ef004 -- tools/experiments/r06_encode_variants.patch is the diff that puts it back --
# and is built from there:   n2norm = the block-I variant (today's product), pkfma / pkfma_n2norm = block J.
cd /root/repo/gradient-quantization_amd || exit 1
mkdir -p ../tools/exp /tmp/vb/csrc
cp csrc/*.hpp csrc/*.h /tmp/vb/csrc/
git show 33ef004:gradient-quantization_amd/csrc/hsq_encode_pf.hip > /tmp/vb/csrc/hsq_encode_pf.hip || exit 1
git show 33ef004:gradient-quantization_amd/csrc/hsq_pf_common.hpp > /tmp/vb/csrc/hsq_pf_common.hpp || exit 1
BASE="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -I../include -I/tmp/vb/csrc -fno-honor-nans"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
OBJS=$(ls build/*.o | grep -v "/hsq_encode_pf.o")
one() {  # name, flags...
  n=$1; shift
  hipcc $BASE "$@" -c /tmp/vb/csrc/hsq_encode_pf.hip -o /tmp/vb/$n.o 2>/tmp/vb/$n.err && hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/exp/libgq_$n.so /tmp/vb/$n.o $OBJS || { echo "$n FAILED"; grep -i error /tmp/vb/$n.err | head -5; }
}
one round5 $NOPK &                            # (the masked n2 of rounds 4-5: block I's baseline)
one n2norm $NOPK -DGQ_PF_N2NORM &
one pkfma -DGQ_PF_PKFMA &
one pkfma_n2norm -DGQ_PF_PKFMA -DGQ_PF_N2NORM &
wait
ls -la ../tools/exp/*.so
