# round 6: K <= 256 on the prefilter / specialised level + decode kernels: tests, fuzzers, timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "smaller or k64 or k32 or _k5 or _k6 or descriptor or prefilter" 2>&1 | tail -15
timeout 600 python tools/fuzz_prefilter.py 600 5 smallk 2>&1 | tail -5 | tee $O/fuzz_prefilter_smallk.txt
timeout 600 python tools/fuzz_batched.py 150 3 2>&1 | tail -3 | tee $O/fuzz_batched_smallk.txt
timeout 600 python tools/time_generic.py 16,64 16,32 16,16 8,32 32,64 32,32 24,64 8,8 2>&1 | grep -v amdgpu | tee $O/time_small_k.txt
timeout 900 python tools/bench_resnet50.py 2>/dev/null | grep "K32\|K64\|segment table\|c-dim 8 batched" | tee $O/resnet50_small_k.txt
