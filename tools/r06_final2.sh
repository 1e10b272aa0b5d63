# round 6, the final code: GPU suite, the three default bench lines, rocprofv3 kernel stats of the driver's command, stamps
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
python -m pytest tests -q -m gpu > $O/gputests_final.txt 2>&1
grep -E "passed|failed" $O/gputests_final.txt | tail -1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-workloads --traffic off > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> /dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_resnet50 -- python3 $GRAFT_REPO_ROOT/bench.py --workload resnet50 --steps 200 --warmup 20 --traffic off > $GRAFT_REPO_ROOT/$O/bench_resnet50_under_rocprof.json 2> /dev/null)
python tools/kstats.py $O 100 > $O/kernel_stats_short.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --workload resnet50 > $O/bench_resnet50.json 2> $O/bench_resnet50.err
python bench.py --workload qsgd > $O/bench_qsgd.json 2> /dev/null
GQ_LIB_PATH=gradient-quantization_amd/libgq_hsq_clock.so python tools/stamp_read.py > $O/stamps.txt 2>&1
python tools/time_pf_d.py 32 8 16 12 24 2>&1 | grep -v amdgpu > $O/time_pf_d_all.txt
python tools/batched_vs_flat.py 2>/dev/null | tail -5 > $O/batched_vs_flat.txt
python - <<'PY'
import json
for f in ("bench", "bench_resnet50", "bench_qsgd"):
    d = json.loads(open("gpurun_out/r06/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "ms_per_step %.5f" % d["ms_per_step"], "value %.4g" % d["value"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
PY
cat $O/time_pf_d_all.txt | grep impl=4
