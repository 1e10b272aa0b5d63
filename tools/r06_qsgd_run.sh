mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "qsgd" 2>&1 | tail -15 > gpurun_out/r06/qsgd_tests.txt
python tools/qsgd_r.py > gpurun_out/r06/qsgd_r.txt 2>&1
python bench.py --workload qsgd --steps 400 --warmup 50 > gpurun_out/r06/bench_qsgd.json 2>gpurun_out/r06/bench_qsgd.err
tail -15 gpurun_out/r06/qsgd_tests.txt; tail -12 gpurun_out/r06/qsgd_r.txt; python -c "
import json; d=json.loads(open('gpurun_out/r06/bench_qsgd.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['phases_ms'])"
