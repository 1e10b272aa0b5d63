"""Worker for test_autotune_drops_a_transport_that_fails_on_one_rank (gloo, CPU).  TEST-ONLY."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "gradient-quantization_amd"))

if __name__ == "__main__":
    rank, world, out, bad_rank = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    bad_modes = sys.argv[5].split(",")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gq_amd import exchange
    ex = exchange.WireExchange(world, rank, 1, 4096, torch.device("cpu"))
    ex.local.fill_(rank + 1)
    calls = []
    if rank == bad_rank:
        # the injected failure: on this rank only, building the transport raises (what a backend that lacks the
        # operation, or a refused argument, looks like) -- its peers' transports are healthy
        good_direct, good_allgather = ex._direct, ex._allgather

        def bad_direct(*a, **k):
            if "direct" in bad_modes or "split" in bad_modes:
                raise RuntimeError("injected: this rank refuses point-to-point transports")
            return good_direct(*a, **k)

        def bad_allgather(*a, **k):
            if "allgather" in bad_modes:
                raise RuntimeError("injected: this rank refuses the all-gather")
            return good_allgather(*a, **k)
        ex._direct, ex._allgather = bad_direct, bad_allgather

    def step(mode):
        calls.append(mode)
        buf, pend = ex.start(mode, cut=2048)
        for p in pend:
            p.wait()
        assert all(int(buf[r, 0]) == r + 1 and int(buf[r, -1]) == r + 1 for r in range(world))

    mode = ex.autotune(step, rounds=3)
    with open(out + "_rank%d.txt" % rank, "w") as f:
        f.write(mode + "\n" + repr(sorted(ex.timings_ms.items())) + "\n" + ",".join(calls) + "\n")
    dist.barrier()
    dist.destroy_process_group()
