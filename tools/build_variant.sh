#!/bin/bash
# A/B experiments on the d16 prefilter encode: build a variant of hsq_encode_pf.hip (an edited copy) against the objects
# of the in-tree build, then time both on one box:
#     tools/build_variant.sh tools/exp/libgq_X.so /tmp/my_copy_of_hsq_encode_pf.hip [extra hipcc flags]
#     gpurun -- 'timeout 300 python tools/ab_time.py product tools/exp/libgq_X.so'
# (keep the edited copy OUTSIDE a directory that holds stale copies of the headers: the compiler looks there first)
# VARIANT_OF=pvq (or any other source's stem) swaps that object instead: VARIANT_OF=pvq tools/build_variant.sh out.so /tmp/pvq_copy.hip
OUT=$1; PF=$2; shift 2
STEM=${VARIANT_OF:-hsq_encode_pf}
NANS=-fno-honor-nans; [ "$STEM" = hsq_encode_pf ] || [ "$STEM" = hsq_encode_pfd ] || NANS=""
cd /root/repo/gradient-quantization_amd
mkdir -p /tmp/vb_$$
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -fvisibility=hidden -I../include -Icsrc"
hipcc $FLAGS $NANS "$@" -c $PF -o /tmp/vb_$$/pf.o || exit 1
OBJS=$(ls build/*.o | grep -v "/$STEM.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT /tmp/vb_$$/pf.o $OBJS
rm -rf /tmp/vb_$$
ls -la $OUT
