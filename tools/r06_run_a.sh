# round 6, call A: the whole GPU suite + the two-phase list step (graphs on) with its kernel stats
mkdir -p gpurun_out/r06
python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06/gputests_a.txt
python bench.py --workload resnet50 --ef --two-phase --steps 400 --warmup 50 > gpurun_out/r06/bench_ef_twophase.json 2> gpurun_out/r06/bench_ef_twophase.err
python bench.py --workload resnet50 --two-phase --steps 400 --warmup 50 > gpurun_out/r06/bench_twophase.json 2> gpurun_out/r06/bench_twophase.err
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06/prof_ef_twophase -- python $R/bench.py --workload resnet50 --ef --two-phase --steps 200 --warmup 20 > $R/gpurun_out/r06/prof_ef_twophase.log 2>&1
cd $R
find gpurun_out/r06/prof_ef_twophase -name "*kernel_trace.csv" -delete
tail -5 gpurun_out/r06/gputests_a.txt
for f in bench_ef_twophase bench_twophase; do python - <<PY
import json
d=json.loads(open('gpurun_out/r06/$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['config']['launches'][:200])
PY
done
find gpurun_out/r06/prof_ef_twophase -name "*kernel_stats.csv" | head -1 | xargs head -14
