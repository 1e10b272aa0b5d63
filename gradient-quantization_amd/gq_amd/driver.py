"""Training driver: the counterpart of the reference's main.py loop (main.py:79-233) for boxes
without torchvision / TensorFlow / network access.

Same flag names and defaults as main.py:83-122, same iteration order as one_iter (main.py:216-233):
    for user in range(num_users): zero_grad -> forward -> backward -> quantizer.record(user, epoch)
    quantizer.apply() -> optimizer.step()
and the same user split of each batch (main.py:189-193).  Differences, all host-side plumbing:
  * data: `--data synthetic` (default): a learnable classification set with the named dataset's shapes
    (`--dataset mnist|cifar10`), labels from a fixed random teacher -- no downloads;  `--data disk`: the real
    MNIST (IDX files) / CIFAR-10 (python pickles) under `--data-root`, read without torchvision, normalised and
    augmented on the device as dataloaders.py:5-47 does (gq_amd/datasets.py);
  * evaluation: main.py:236-255's test() -- accuracy and summed batch loss over the test split -- at every log
    point, as main.py:197-211 logs loss and accuracy together;
  * models: `fcn` (784-256-10, models/fcn.py:12-13) and a CIFAR bottleneck `resnet50` with the
    reference's parameter-shape list (tests/golden/resnet50_cifar_shapes.json), plain torch.nn;
  * schedule: main.py:136-163's learning-rate steps (new optimizer at epochs 51 and 71, none for MNIST, SignSGD's
    own constants) are reproduced; the number of epochs stays a flag (main.py hard-codes 20 / 150 / 1000);
  * logging: one JSON object per log point on stdout / --logfile instead of TF1 summaries;
  * real data parallelism: launched under torch.distributed.run every rank is one (or
    --num-users) of the reference's users; the quantizer all-gathers the wire once per step.
The quantization itself is the HIP path (gq_amd.compressors / gq_amd.quantizers): no CPU fallback.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from .datasets import OnDiskClassification
from .compressors import (IdenticalCompressor, NearestNeighborCompressor, QSGDCompressor, SignSGDCompressor,
                          TopKSparsificationCompressor)
from .quantizers import Quantizer

quantizer_choices = {          # main.py:20-26
    'sgd': IdenticalCompressor,
    'qsgd': QSGDCompressor,
    'hsq': NearestNeighborCompressor,
    'sign': SignSGDCompressor,
    'topk': TopKSparsificationCompressor,
}


class FCN(nn.Module):
    """784-256-10 (models/fcn.py)."""

    def __init__(self, num_classes=10):
        super().__init__()
        self.fc1 = nn.Linear(784, 256)
        self.fc2 = nn.Linear(256, num_classes)

    def forward(self, x):
        return self.fc2(F.relu(self.fc1(x.view(x.shape[0], -1))))


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, self.expansion * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(self.expansion * planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(nn.Conv2d(in_planes, self.expansion * planes, 1, stride=stride, bias=False),
                                          nn.BatchNorm2d(self.expansion * planes))

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + self.shortcut(x))


class ResNet50(nn.Module):
    """CIFAR-style ResNet-50: 3x3 stem, stages [3,4,6,3] of bottleneck blocks, 23.5 M parameters."""

    def __init__(self, num_classes=10):
        super().__init__()
        self.in_planes = 64
        self.conv1 = nn.Conv2d(3, 64, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._stage(64, 3, 1)
        self.layer2 = self._stage(128, 4, 2)
        self.layer3 = self._stage(256, 6, 2)
        self.layer4 = self._stage(512, 3, 2)
        self.linear = nn.Linear(512 * _Bottleneck.expansion, num_classes)

    def _stage(self, planes, blocks, stride):
        layers = []
        for s in [stride] + [1] * (blocks - 1):
            layers.append(_Bottleneck(self.in_planes, planes, s))
            self.in_planes = planes * _Bottleneck.expansion
        return nn.Sequential(*layers)

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.layer4(self.layer3(self.layer2(self.layer1(out))))
        out = F.adaptive_avg_pool2d(out, 1).flatten(1)
        return self.linear(out)


network_choices = {'fcn': FCN, 'resnet50': ResNet50}
dataset_shapes = {'mnist': (1, 28, 28), 'cifar10': (3, 32, 32)}


class SyntheticClassification(object):
    """Fixed random inputs with labels from a random linear teacher: learnable, no files."""

    def __init__(self, dataset, n, num_classes, device, seed, x=None, y=None):
        if x is None:
            g = torch.Generator().manual_seed(seed)
            shape = dataset_shapes[dataset]
            x = torch.randn((n,) + shape, generator=g)
            teacher = torch.randn(x[0].numel(), num_classes, generator=g)
            y = (x.view(n, -1) @ teacher).argmax(1)
        self.x, self.y = x.to(device), y.to(device)
        self.n = int(self.x.shape[0])

    def split(self, n_test):
        """(train, test): the last n_test samples become a held-out set labelled by the same teacher."""
        k = self.n - n_test
        return (SyntheticClassification(None, 0, 0, self.x.device, 0, self.x[:k], self.y[:k]),
                SyntheticClassification(None, 0, 0, self.x.device, 0, self.x[k:], self.y[k:]))

    def batches(self, batch, epoch_seed, rank=0, world=1, shuffle=True, min_share=1):
        """As OnDiskClassification.batches: the short last batch is kept (the reference's loader has drop_last=False) unless it
        cannot give every rank `min_share` samples -- then it is dropped on all ranks alike."""
        from .datasets import rank_slices
        g = torch.Generator().manual_seed(epoch_seed)
        perm = (torch.randperm(self.n, generator=g) if shuffle else torch.arange(self.n)).to(self.x.device)
        for lo, hi in rank_slices(self.n, batch, rank, world, min_share):
            idx = perm[lo:hi]
            yield self.x[idx], self.y[idx]


def build_parser():
    p = argparse.ArgumentParser(description='Gradient quantization on MI355X (driver)')
    p.add_argument('--network', type=str, default='fcn', choices=sorted(network_choices))
    p.add_argument('--dataset', type=str, default='mnist', choices=sorted(dataset_shapes))
    p.add_argument('--quantizer', type=str, default='hsq', choices=sorted(quantizer_choices))
    p.add_argument('--mode', type=str, default='ps', choices=['ps', 'ring'])
    p.add_argument('--scale', type=str, default="exp")
    p.add_argument('--c-dim', type=int, default=32)
    p.add_argument('--k-bit', type=int, default=8)
    p.add_argument('--n-bit', type=int, default=8)
    p.add_argument('--cr', type=int, default=256)
    p.add_argument('--random', type=int, default=True)
    p.add_argument('--num-users', type=int, default=8)
    p.add_argument('--batch-size', type=int, default=32)
    p.add_argument('--epochs', type=int, default=2)
    p.add_argument('--momentum', type=float, default=0.9)
    p.add_argument('--weight-decay', type=float, default=5e-4)
    p.add_argument('--lr', type=float, default=0.1)
    p.add_argument('--no-cuda', action='store_true', default=False)
    p.add_argument('--ef', action='store_true', default=False)
    p.add_argument('--two-phase', action='store_true', default=False)
    p.add_argument('--seed', type=int, default=1)
    p.add_argument('--train-size', type=int, default=4096, help='synthetic samples per epoch')
    p.add_argument('--log-interval', type=int, default=8, help='iterations between JSON log lines')
    p.add_argument('--logfile', type=str, default=None)
    p.add_argument('--gq-rng', type=str, default=None, choices=[None, 'device', 'reference', 'keyed'])
    p.add_argument('--gq-graph', dest='gq_graph', action='store_true', default=None,
                   help='replay the quantizer step from HIP graphs per set of gradient addresses (the default, also $GQ_GRAPH=1; '
                        'with --gq-rng reference the draws come from the host and the launches stay eager)')
    p.add_argument('--no-gq-graph', dest='gq_graph', action='store_false', help='eager launches')
    p.add_argument('--data', type=str, default='synthetic', choices=['synthetic', 'disk'],
                   help="disk: MNIST idx files / CIFAR-10 pickles under --data-root (no torchvision)")
    p.add_argument('--data-root', type=str, default='./data')
    p.add_argument('--test-batch-size', type=int, default=1000)
    p.add_argument('--test-size', type=int, default=0, help='synthetic data: held-out samples for the accuracy loop (0 = none)')
    p.add_argument('--timing', action='store_true', default=False,
                   help="log lines carry ms_per_iter as the wall time per iteration since the previous log line (device drained at both "
                        "ends), quantizer_ms (HIP events around record / apply, summed per iteration) and quantizer_share")
    return p


def test(model, loss_func, test_data, test_batch_size):
    """main.py:236-255: accuracy over the test split; the loss is the reference's sum of per-batch MEAN losses
    divided by the number of samples (LOSS_FUNC is CrossEntropyLoss with mean reduction, `.sum()` of a scalar)."""
    model.eval()
    test_loss, correct = 0.0, 0
    with torch.no_grad():
        for data, target in test_data.batches(test_batch_size, 0, shuffle=False):
            output = model(data)
            test_loss += loss_func(output, target).sum().item()
            pred = output.argmax(dim=1, keepdim=True)
            correct += pred.eq(target.view_as(pred)).sum().item()
    model.train()
    return correct / test_data.n, test_loss / test_data.n


def one_iter(model, loss_func, optimizer, quantizer, train_data, epoch, spans=None):
    """main.py:216-233.  spans (a list, --timing): (start, end) HIP event pairs around every quantizer call of the iteration --
    the quantizer's share of the iteration's DEVICE time (record per user, apply once)."""
    model.train()
    losses = []

    def timed(fn):
        if spans is None:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        spans.append((a, b))

    for user_id, (data, target) in enumerate(train_data):
        optimizer.zero_grad()
        loss = loss_func(model(data), target)
        losses.append(loss.detach())
        loss.backward()
        timed(lambda: quantizer.record(user_id, epoch=epoch))
    timed(quantizer.apply)
    optimizer.step()
    return torch.stack(losses).mean()


def train(args, log=None):
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if args.no_cuda or not torch.cuda.is_available():
        raise RuntimeError("gq_amd runs on MI355X only: there is no CPU path (drop --no-cuda)")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)      # the HIP library launches on the current device's current stream
    torch.manual_seed(args.seed)
    num_classes = 10
    model = network_choices[args.network](num_classes=num_classes).to(device)
    quantizer = Quantizer(quantizer_choices[args.quantizer], model.parameters(), args)
    optimizer = optim.SGD(model.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)
    # main.py:136-163: the learning-rate steps are NEW optimizers (momentum buffers start over), weight decay 5e-4
    # from then on; none for MNIST; SignSGD has its own constants
    steps = {} if args.dataset == 'mnist' else {51: 0.01, 71: 0.005}
    momentum = args.momentum
    if args.quantizer == 'sign':
        steps, momentum = {51: 0.0005, 71: 0.0001}, 0.0
        optimizer = optim.SGD(model.parameters(), lr=1e-3, momentum=0.0, weight_decay=0.1)
    loss_func = nn.CrossEntropyLoss()
    test_data = None
    if getattr(args, "data", "synthetic") == "disk":
        data = OnDiskClassification(args.dataset, args.data_root, device, train=True)
        test_data = OnDiskClassification(args.dataset, args.data_root, device, train=False)
    else:
        n_test = int(getattr(args, "test_size", 0))
        data = SyntheticClassification(args.dataset, args.train_size + n_test, num_classes, device, args.seed)
        if n_test:
            data, test_data = data.split(n_test)
    out = open(args.logfile, "a") if (args.logfile and rank == 0) else None
    history = []
    it = 0
    timing, mark, mark_it, spans = bool(getattr(args, "timing", False)), None, 0, []
    for epoch in range(1, args.epochs + 1):
        if epoch in steps:
            optimizer = optim.SGD(model.parameters(), lr=steps[epoch], momentum=momentum, weight_decay=5e-4)
        # the loader yields num_users*batch_size samples per rank; split across users as main.py:189-193
        # (a last batch too short to give every user of EVERY rank a sample is dropped by the loader on all ranks alike:
        # the reference would average an empty batch, and ranks that disagreed would leave one another alone in a collective)
        for x, y in data.batches(args.batch_size * args.num_users, 1000 * args.seed + epoch, rank, world, min_share=args.num_users):
            ub = x.shape[0] // args.num_users
            users = [(x[u * ub:(u + 1) * ub], y[u * ub:(u + 1) * ub]) for u in range(args.num_users - 1)]
            users.append((x[(args.num_users - 1) * ub:], y[(args.num_users - 1) * ub:]))
            t0 = time.perf_counter()
            if timing and mark is None:
                torch.cuda.synchronize()
                mark, mark_it, spans = time.perf_counter(), it, []
            loss = one_iter(model, loss_func, optimizer, quantizer, users, epoch, spans if timing else None)
            it += 1
            if it % args.log_interval == 0 or it == 1:
                rec = {"iter": it, "epoch": epoch, "loss": float(loss), "ms_per_iter": (time.perf_counter() - t0) * 1e3,
                       "ranks": world, "users_per_rank": args.num_users}
                if timing:      # the interval since the last log line, device drained at both ends
                    torch.cuda.synchronize()
                    n_it = it - mark_it
                    q_ms = sum(a.elapsed_time(b) for a, b in spans) / n_it
                    rec["ms_per_iter"] = (time.perf_counter() - mark) * 1e3 / n_it
                    rec["quantizer_ms"] = q_ms
                    rec["quantizer_share"] = q_ms / rec["ms_per_iter"]
                    rec["iters_timed"] = n_it
                    mark = None
                if test_data is not None and rank == 0:      # main.py:197-211: loss and test accuracy are logged together (once: rank 0)
                    acc, tl = test(model, loss_func, test_data, getattr(args, "test_batch_size", 1000))
                    rec["accuracy(%)"] = 100.0 * acc
                    rec["test_loss"] = tl
                history.append(rec)
                if rank == 0:
                    line = json.dumps(rec)
                    print(line)
                    if out:
                        out.write(line + "\n")
                if log is not None:
                    log.append(rec)
    if out:
        out.close()
    return model, quantizer, history


def main(argv=None):
    args = build_parser().parse_args(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    train(args)


if __name__ == "__main__":
    main(sys.argv[1:])
