# round 6, last call: the whole GPU suite (summary to a file), and the three list benches the profile run lost to a bug in record()'s short way
mkdir -p gpurun_out/r06
python -m pytest tests -q -m gpu > gpurun_out/r06/gputests_final.txt 2>&1
tail -3 gpurun_out/r06/gputests_final.txt | head -2
O=gpurun_out/r06
python bench.py --workload resnet50 > $O/bench_resnet50.json 2> $O/bench_resnet50.err
python bench.py --workload resnet50 --c-dim 32 --n-bit 8 --traffic off > $O/bench_resnet50_main_defaults.json 2> /dev/null
python bench.py --workload resnet50 --c-dim 8 --traffic off > $O/bench_resnet50_d8.json 2> /dev/null
ls -la $O/bench_resnet50.json $O/bench_resnet50_main_defaults.json $O/bench_resnet50_d8.json
