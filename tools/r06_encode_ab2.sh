# round 6, encode A/B block L: tile indices as 32-bit integers (scalar compares at the loop top) + the first chain's MFMA issued before
# the tile's bookkeeping, against the source before the change (tools/exp/libgq_before_i32.so)
mkdir -p gpurun_out/r06
O=gpurun_out/r06/encode_ab_L.txt
python tools/ab_time.py product tools/exp/libgq_before_i32.so > $O 2>&1
GQ_AB_D=32 python tools/ab_time.py product tools/exp/libgq_before_i32.so >> $O 2>&1
GQ_AB_D=8 python tools/ab_time.py product tools/exp/libgq_before_i32.so >> $O 2>&1
python tools/ab_script.py tools/batched_vs_flat.py product tools/exp/libgq_before_i32.so 2>/dev/null | grep "76 tensors, 23.50" >> $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -2 >> $O
python tools/fuzz_prefilter.py 3000 808 2>&1 | tail -1 >> $O
GQ_LIB_PATH=gradient-quantization_amd/libgq_hsq_clock.so python tools/stamp_read.py 2>/dev/null | head -9 >> $O
cat $O
