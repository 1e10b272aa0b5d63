"""A short run of the chunked ResNet-50 list step for `rocprofv3 --kernel-trace`: do a chunk's level + decode launch and the next
chunk's encode overlap in time?  (tools/overlap_trace_read.py prints the last steps' kernel intervals from the trace.)
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06/trace -- python tools/overlap_trace.py hsq 0.58,0.42 1"""
import os, sys
sys.argv, spec, streams = sys.argv[:2], sys.argv[2], sys.argv[3]
os.environ["OVERLAP_SPECS"] = spec
os.environ["GQ_OVERLAP_STREAMS"] = streams
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib.util
src = open(os.path.join(ROOT, "tools", "overlap_ab.py")).read().split("\nfor name in (sys.argv[1:]")[0]
exec(compile(src, "overlap_ab_head", "exec"))
Comp, kw, ef = CONFIGS[sys.argv[1]]
q, params, step = build(Comp, kw, ef, spec, 1)
for i in range(300):
    step(i)
torch.cuda.synchronize()
print("whole-step graphs:", sum(1 for e in q._step_graphs.values() if e[1] is not None), "groups:", len(q._groups))
