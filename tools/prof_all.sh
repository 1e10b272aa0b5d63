#!/bin/bash
# Everything profiles/ holds for one round, in one gpurun call:  gpurun --timeout 2400 -- 'bash tools/prof_all.sh r05'
# -> gpurun_out/<tag>/{stats (rocprofv3 --kernel-trace --stats of bench.py), bench*.json, pmc_show.txt, stamps.txt, ...};
# copy what is to be judged into profiles/.  The stamp section reads the stamped twin build.py makes (libgq_hsq_clock.so).
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
# 1. the driver's command under the kernel tracer (the profiler's own program is python3 itself)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-workloads --traffic off > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> /dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_resnet50 -- python3 $GRAFT_REPO_ROOT/bench.py --workload resnet50 --steps 200 --warmup 20 --traffic off > $GRAFT_REPO_ROOT/$O/bench_resnet50_under_rocprof.json 2> /dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_qsgd -- python3 $GRAFT_REPO_ROOT/bench.py --workload qsgd --steps 200 --warmup 20 --traffic off > $GRAFT_REPO_ROOT/$O/bench_qsgd_under_rocprof.json 2> /dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_resnet50_ef -- python3 $GRAFT_REPO_ROOT/bench.py --workload resnet50 --ef --steps 200 --warmup 20 --traffic off > $GRAFT_REPO_ROOT/$O/bench_resnet50_ef_under_rocprof.json 2> /dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_resnet50_ef_twophase -- python3 $GRAFT_REPO_ROOT/bench.py --workload resnet50 --ef --two-phase --steps 200 --warmup 20 --traffic off > $GRAFT_REPO_ROOT/$O/bench_resnet50_ef_twophase_under_rocprof.json 2> /dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_resnet50_main_defaults -- python3 $GRAFT_REPO_ROOT/bench.py --workload resnet50 --c-dim 32 --n-bit 8 --steps 200 --warmup 20 --traffic off > $GRAFT_REPO_ROOT/$O/bench_resnet50_main_defaults_under_rocprof.json 2> /dev/null)
python tools/kstats.py $O 100 > $O/kernel_stats_short.txt 2>&1
# 2. the default lines (live PMC traffic, CPU baseline) of the three workloads
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --workload resnet50 > $O/bench_resnet50.json 2> $O/bench_resnet50.err
python bench.py --workload qsgd > $O/bench_qsgd.json 2> /dev/null
python bench.py --workload resnet50 --ef --traffic off > $O/bench_resnet50_ef.json 2> /dev/null
python bench.py --workload resnet50 --ef --two-phase --traffic off > $O/bench_resnet50_ef_twophase.json 2> /dev/null
python bench.py --workload resnet50 --c-dim 32 --n-bit 8 --traffic off > $O/bench_resnet50_main_defaults.json 2> /dev/null
python bench.py --workload resnet50 --c-dim 8 --traffic off > $O/bench_resnet50_d8.json 2> /dev/null
python bench.py --workload resnet50 --k-bit 6 --traffic off > $O/bench_resnet50_k6.json 2> /dev/null
python bench.py --workload resnet50 --k-bit 5 --traffic off > $O/bench_resnet50_k5.json 2> /dev/null
python bench.py --workload resnet50 --c-dim 8 --k-bit 5 --traffic off > $O/bench_resnet50_d8_k5.json 2> /dev/null
python bench.py --workload resnet50 --no-graph --traffic off > $O/bench_resnet50_eager.json 2> /dev/null
python bench.py --workload qsgd --no-graph --traffic off > $O/bench_qsgd_eager.json 2> /dev/null
python bench.py --wire-levels packed6 --no-cpu-baseline --traffic off > $O/bench_packed6.json 2> /dev/null
GQ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo.json 2> /dev/null
# 3. SQ counters of the encode (three passes) and the in-kernel stamps
bash tools/pmc_any.sh $TAG tools/exp_time.py
bash tools/pmc_overlap.sh $TAG tools/exp_time.py
python tools/pmc_show.py $TAG > $O/pmc_show.txt 2>&1
GQ_LIB_PATH=gradient-quantization_amd/libgq_hsq_clock.so python tools/stamp_read.py > $O/stamps.txt 2>&1
# 4. side measurements
python tools/decode_r.py 1 2 4 8 16 > $O/decode_r.txt 2>&1
python tools/hsq_batched_r.py > $O/hsq_batched_r.txt 2>&1
python tools/time_generic.py > $O/time_generic.txt 2>&1
python tools/time_pf_d.py 32 8 16 12 24 >> $O/time_generic.txt 2>&1
python tools/pvq_time.py > $O/pvq_time.txt 2>&1
python tools/qsgd_r.py > $O/qsgd_r.txt 2>&1
python tools/bench_resnet50.py 2>&1 | grep -v 'alternate dimension' > $O/resnet50_steps.txt
python tools/host_breakdown.py > $O/host_breakdown.txt 2>&1
python tools/config_sweep.py > $O/config_sweep.txt 2>/dev/null
(GQ_AB_ROUNDS=1 python tools/pvq_ab.py product; GQ_AB_D=32 GQ_AB_ROUNDS=1 python tools/pvq_ab.py product; GQ_AB_D=8 GQ_AB_ROUNDS=1 python tools/pvq_ab.py product; GQ_PVQ_TWO_SWEEPS=1 GQ_AB_ROUNDS=1 python tools/pvq_ab.py product) > $O/pvq_ab.txt 2>&1
python tools/cpu_scaling.py > $O/cpu_scaling.txt 2>&1
python tools/graph_pieces.py > $O/graph_pieces.txt 2>&1
python tools/batched_vs_flat.py > $O/batched_vs_flat.txt 2>&1
(for kb in 6 5; do echo "== k_bit $kb"; GQ_AB_KBIT=$kb python tools/batched_vs_flat.py 2>/dev/null | tail -1; done) >> $O/batched_vs_flat.txt
python tools/step_host_vs_device.py 2>/dev/null > $O/step_host_vs_device.txt
GQ_DIRECT_REPLAY=0 python tools/step_host_vs_device.py 2>/dev/null > $O/step_host_vs_device_graphlaunch.txt
python tools/direct_vs_graph.py 2>/dev/null > $O/direct_vs_graph.txt
python tools/cold_vs_warm.py 2>/dev/null > $O/cold_vs_warm.txt
GQ_LIB_PATH=$PWD/gradient-quantization_amd/libgq_hsq_clock.so python tools/cold_vs_warm.py 2>/dev/null >> $O/cold_vs_warm.txt
GQ_FUSE_STEP=0 python bench.py --workload resnet50 --no-workloads --traffic off > $O/bench_resnet50_two_graphs.json 2> /dev/null
GQ_AGGREGATE=fma GQ_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo_fma.json 2> /dev/null
ls $O
