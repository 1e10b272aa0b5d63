"""Where the HOST time of the quantizer goes inside a real training loop (driver.one_iter: gradients that autograd allocates anew
every backward): wall time of record() / apply() per iteration without a device sync, how the graph caches fill, and a cProfile of
the quantizer calls in steady state.
    python tools/train_host_profile.py [hsq|qsgd|sgd] [users] [iterations]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
import torch
from gq_amd import driver

quant = sys.argv[1] if len(sys.argv) > 1 else "hsq"
users = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 400
extra = {"hsq": ["--c-dim", "16", "--k-bit", "8", "--n-bit", "6"], "qsgd": ["--c-dim", "128", "--n-bit", "2"], "sgd": []}[quant]
args = driver.build_parser().parse_args(["--quantizer", quant, "--network", "resnet50", "--dataset", "cifar10", "--num-users", str(users),
                                         "--batch-size", "32"] + extra)
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = driver.ResNet50(10).to(dev)
q = driver.Quantizer(driver.quantizer_choices[quant], model.parameters(), args)
opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4)
lossf = torch.nn.CrossEntropyLoss()
x = torch.randn(users * 32, 3, 32, 32, device=dev)
y = torch.randint(0, 10, (users * 32,), device=dev)
prof = cProfile.Profile()
t_rec = t_app = 0.0
marks = []
for it in range(iters):
    steady = it >= iters - 100
    for u in range(users):
        opt.zero_grad()
        lossf(model(x[u * 32:(u + 1) * 32]), y[u * 32:(u + 1) * 32]).backward()
        t0 = time.perf_counter()
        if steady: prof.enable()
        q.record(u, epoch=1)
        if steady: prof.disable()
        t_rec += time.perf_counter() - t0
    t0 = time.perf_counter()
    if steady: prof.enable()
    q.apply()
    if steady: prof.disable()
    t_app += time.perf_counter() - t0
    opt.step()
    if (it + 1) % 50 == 0:
        torch.cuda.synchronize()
        marks.append((it + 1, t_rec / 50 * 1e3, t_app / 50 * 1e3,
                      sum(1 for e in q._rec_graphs.values() if e[1] is not None), len(q._rec_graphs),
                      sum(1 for e in q._apply_graphs.values() if e[1] is not None), sum(1 for e in q._step_graphs.values() if e[1] is not None)))
        t_rec = t_app = 0.0
print("%s, %d user(s): host ms per iteration in record() / apply(), graphs captured (record / address sets seen / apply / whole step)" % (quant, users))
for m in marks:
    print("  iter %4d  record %.3f ms  apply %.3f ms   graphs %d / %d / %d / %d" % m)
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
