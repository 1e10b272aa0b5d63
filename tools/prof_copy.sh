#!/bin/bash
# After `gpurun -- 'bash tools/prof_all.sh r05'`: copy what is to be judged from gpurun_out/<tag>/ into profiles/<tag>_*.
#     bash tools/prof_copy.sh r05
TAG=${1:-r05}
O=gpurun_out/$TAG
P=profiles
for f in step_host_vs_device step_host_vs_device_graphlaunch direct_vs_graph cold_vs_warm batched_vs_flat cpu_scaling decode_r graph_pieces host_breakdown hsq_batched_r kernel_stats_short qsgd_r time_generic config_sweep pvq_time; do
    [ -f $O/$f.txt ] && cp $O/$f.txt $P/${TAG}_$f.txt
done
[ -f $O/stamps.txt ] && cp $O/stamps.txt $P/${TAG}_pf_kernel_stamps.txt
[ -f $O/pmc_show.txt ] && cp $O/pmc_show.txt $P/${TAG}_pmc_sq.txt
[ -f $O/resnet50_steps.txt ] && cp $O/resnet50_steps.txt $P/${TAG}_resnet50_quantizer_steps.txt
[ -f $O/pvq_ab.txt ] && cp $O/pvq_ab.txt $P/${TAG}_pvq_ab_final.txt
for f in bench bench_packed6 bench_qsgd bench_qsgd_eager bench_resnet50 bench_resnet50_d8 bench_resnet50_k6 bench_resnet50_k5 bench_resnet50_d8_k5 bench_resnet50_eager bench_resnet50_ef bench_resnet50_ef_twophase bench_resnet50_main_defaults bench_resnet50_two_graphs; do
    [ -f $O/$f.json ] && cp $O/$f.json $P/${TAG}_$f.json
done
[ -f $O/bench_2ranks_gloo.json ] && cp $O/bench_2ranks_gloo.json $P/${TAG}_bench_2ranks_gloo_one_gpu.json
for d in stats:kernel_stats stats_qsgd:qsgd_kernel_stats stats_resnet50:resnet50_kernel_stats stats_resnet50_ef:resnet50_ef_kernel_stats stats_resnet50_ef_twophase:resnet50_ef_twophase_kernel_stats stats_resnet50_main_defaults:resnet50_main_defaults_kernel_stats; do
    src=${d%%:*}; dst=${d##*:}
    f=$(ls -t $O/$src/*/*kernel_stats.csv 2>/dev/null | head -1)      # (the newest: gpurun merges a run into what earlier runs left)
    [ -n "$f" ] && cp $f $P/${TAG}_$dst.csv
done
ls $P | grep "^${TAG}_" | wc -l
