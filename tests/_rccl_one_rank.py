"""Child of tests/test_gpu_api.py::test_rccl_single_rank_exchange_calls: the NON-staged transports of gq_amd.exchange on the real
collective library with ONE rank (several ranks cannot share a GPU under RCCL): the calls as the product makes them -- an in-place
all_gather_into_tensor of uint8 rows on the process group's stream behind kernels of the current stream, Work.wait(), a kernel that
reads the result -- and the collectives bench.py uses around its timed windows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

port = int(sys.argv[1])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
from gq_amd import exchange

n = 3_000_000      # one row: about the wire of a 25 M-element gradient
ex = exchange.WireExchange(1, 0, 2, n, dev)
assert not ex._staged
src = torch.randint(0, 255, (8, n), device=dev, dtype=torch.uint8)
for it in range(40):
    ex.local[0].copy_(src[it % 8])                 # a kernel of the current stream writes the row ...
    pend = ex._allgather(1)                        # ... the collective (fewer rows than slots: out of place) runs on the group's stream ...
    pend.wait()                                    # ... and the current stream waits for it
    got = ex._partial_buffer(1)[0].to(torch.int32).sum()      # a kernel that reads what the collective wrote
    ex.local[0].zero_()                            # the next writer of the row, queued behind the collective's reads
    assert int(got.item()) == int(src[it % 8].to(torch.int32).sum().item()), it
ex.local.copy_(src[:2])
ref = ex.gathered.clone()
p = ex._allgather(2)                               # all rows: in place (input is a view of the output)
p.wait()
torch.cuda.synchronize()
assert torch.equal(ex.gathered, ref)
p = ex._direct(2, 0, n)                            # no peers: nothing queued
assert p.works == []
p.wait()
one = torch.ones(1, device=dev)
dist.all_reduce(one)
assert one.item() == 1.0
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 1.5
dist.destroy_process_group()
print("rccl one rank ok")
