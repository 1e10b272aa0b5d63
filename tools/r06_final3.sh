# round 6, the code with direct replay: r06_final2.sh + the host-vs-device and graph-vs-direct measurements
cd $GRAFT_REPO_ROOT
bash tools/r06_final2.sh
O=gpurun_out/r06
python tools/step_host_vs_device.py 2>/dev/null > $O/step_host_vs_device.txt
GQ_DIRECT_REPLAY=0 python tools/step_host_vs_device.py 2>/dev/null > $O/step_host_vs_device_graphlaunch.txt
python tools/direct_vs_graph.py 2>/dev/null > $O/direct_vs_graph.txt
python bench.py --workload resnet50 --ef > $O/bench_resnet50_ef.json 2>/dev/null
python bench.py --workload resnet50 --ef --two-phase > $O/bench_resnet50_ef_twophase.json 2>/dev/null
tail -3 $O/step_host_vs_device.txt $O/step_host_vs_device_graphlaunch.txt $O/direct_vs_graph.txt
python - <<'PY'
import json
for f in ("bench_resnet50_ef", "bench_resnet50_ef_twophase"):
    try:
        d = json.loads(open("gpurun_out/r06/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "ms_per_step %.5f" % d["ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
