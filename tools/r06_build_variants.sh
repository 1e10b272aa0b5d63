#!/bin/bash
# round 6 A/B builds of the d16 prefilter encode (profiles/r06_encode_ab.txt): tools/exp/libgq_{n2norm,pkfma,pkfma_n2norm}.so
cd /root/repo/gradient-quantization_amd || exit 1
mkdir -p ../tools/exp /tmp/vb
BASE="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -fvisibility=hidden -I../include -Icsrc -fno-honor-nans"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
OBJS=$(ls build/*.o | grep -v "/hsq_encode_pf.o")
one() {  # name, flags...
  n=$1; shift
  hipcc $BASE "$@" -c csrc/hsq_encode_pf.hip -o /tmp/vb/$n.o 2>/tmp/vb/$n.err && hipcc --offload-arch=gfx950 -shared -fPIC -o ../tools/exp/libgq_$n.so /tmp/vb/$n.o $OBJS || { echo "$n FAILED"; grep -i error /tmp/vb/$n.err | head -5; }
}
one n2norm $NOPK -DGQ_PF_N2NORM &
one pkfma -DGQ_PF_PKFMA &
one pkfma_n2norm -DGQ_PF_PKFMA -DGQ_PF_N2NORM &
wait
ls -la ../tools/exp/*.so
