#!/bin/bash
# Everything profiles/ holds for one round, in one gpurun call:  gpurun --timeout 2400 -- 'bash tools/prof_all.sh'
# -> gpurun_out/r02/{stats (rocprofv3 --kernel-trace --stats of bench.py), hbm_traffic.json, pmc_show.txt, stamps.txt,
#    prologue_stamps.txt, bench.json, resnet50_steps.txt, host_breakdown.txt}; copy what is to be judged into profiles/.
# Needs tools/exp/libgq_stamp.so and libgq_pstamp.so (python tools/stamp_build.py; python tools/stamp_prologue.py).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02/bench_under_rocprof.json 2> /dev/null)
bash tools/hbm_traffic.sh
python tools/hbm_traffic.py > gpurun_out/r02/hbm_traffic.txt 2>&1
cp profiles/hbm_traffic.json gpurun_out/r02/
bash tools/pmc_any.sh r02b tools/exp_time.py
bash tools/pmc_overlap.sh r02b tools/exp_time.py
python tools/pmc_show.py r02b > gpurun_out/r02/pmc_show.txt 2>&1
GQ_LIB_PATH=tools/exp/libgq_stamp.so python tools/stamp_read.py > gpurun_out/r02/stamps.txt 2>&1
GQ_LIB_PATH=tools/exp/libgq_pstamp.so python tools/stamp_prologue.py read > gpurun_out/r02/prologue_stamps.txt 2>&1
python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err
python tools/bench_resnet50.py > gpurun_out/r02/resnet50_steps.txt 2>&1
python tools/host_breakdown.py > gpurun_out/r02/host_breakdown.txt 2>&1
ls gpurun_out/r02
python tools/decode_r.py > gpurun_out/r02/decode_r.txt 2>&1
python tools/time_generic.py > gpurun_out/r02/time_generic.txt 2>&1
python tools/pvq_time.py > gpurun_out/r02/pvq_time.txt 2>&1
python tools/qsgd_r.py > gpurun_out/r02/qsgd_r.txt 2>&1
python bench.py --workload qsgd --steps 300 --warmup 30 > gpurun_out/r02/bench_qsgd.json 2> /dev/null
GQ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r02/bench_2ranks_gloo.json 2> /dev/null
