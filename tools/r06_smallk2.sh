cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "smaller or k64 or k32 or _k5 or _k6 or descriptor or prefilter" 2>&1 | tail -5
for cfg in "8 5" "8 8" "16 6"; do set -- $cfg
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_d$1_k$2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload resnet50 --c-dim $1 --k-bit $2 --steps 200 --warmup 20 --traffic off > $GRAFT_REPO_ROOT/$O/bench_resnet50_d$1_k$2.json 2> /dev/null)
f=$(ls -t $O/stats_d$1_k$2/*/*kernel_stats.csv | head -1)
echo "== c_dim $1 k_bit $2"; python3 -c "
import json,csv,sys
d=json.loads(open('$O/bench_resnet50_d$1_k$2.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])
for i,r in enumerate(csv.DictReader(open('$f'))):
    if i<4: print('   %-100s calls %5s avg %8.1f us'%(r['Name'][:100],r['Calls'],float(r['AverageNs'])/1e3))
"
done
