"""On-disk dataset readers of the training driver (gq_amd/datasets.py): IDX (MNIST) and python-pickle (CIFAR-10)
files written here in the formats' own layout, read back without torchvision; the reference's normalisation
constants and augmentation (dataloaders.py:5-47)."""
import gzip
import os
import pickle
import struct
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "gradient-quantization_amd"))


def write_idx(path, arr, gz=False):
    arr = np.ascontiguousarray(arr, np.uint8)
    head = bytes([0, 0, 0x08, arr.ndim]) + struct.pack(">" + "I" * arr.ndim, *arr.shape)
    (gzip.open if gz else open)(path, "wb").write(head + arr.tobytes())


def make_mnist(root, n_train=512, n_test=128, seed=0, gz=False, layout="raw"):
    """A learnable stand-in in MNIST's files: the label is the brightest of ten 2x14 stripes."""
    rng = np.random.RandomState(seed)
    d = os.path.join(root, "MNIST", "raw") if layout == "raw" else root
    os.makedirs(d, exist_ok=True)
    out = {}
    for stem, n in (("train", n_train), ("t10k", n_test)):
        y = rng.randint(0, 10, n).astype(np.uint8)
        x = rng.randint(0, 60, (n, 28, 28)).astype(np.uint8)
        for i in range(n):
            x[i, 2 * y[i]:2 * y[i] + 2, :14] = 250
        suffix = ".gz" if gz else ""
        write_idx(os.path.join(d, stem + "-images-idx3-ubyte" + suffix), x, gz)
        write_idx(os.path.join(d, stem + "-labels-idx1-ubyte" + suffix), y, gz)
        out[stem] = (x, y)
    return out


def make_cifar(root, n_train=256, n_test=64, seed=0):
    rng = np.random.RandomState(seed)
    d = os.path.join(root, "cifar-10-batches-py")
    os.makedirs(d, exist_ok=True)
    out = {}
    for name, n in (("data_batch_1", n_train), ("test_batch", n_test)):
        y = rng.randint(0, 10, n)
        x = rng.randint(0, 40, (n, 3, 32, 32)).astype(np.uint8)
        for i in range(n):
            x[i, y[i] % 3, 3 * y[i]:3 * y[i] + 3, :] = 240
        pickle.dump({b"data": x.reshape(n, 3072), b"labels": [int(v) for v in y], b"batch_label": b"t"},
                    open(os.path.join(d, name), "wb"))
        out[name] = (x, y)
    return out


@pytest.mark.parametrize("gz,layout", [(False, "raw"), (True, "raw"), (False, "flat")])
def test_mnist_idx_reader(tmp_path, gz, layout):
    from gq_amd import datasets
    ref = make_mnist(str(tmp_path), gz=gz, layout=layout)
    x, y = datasets.load_mnist(str(tmp_path), train=True)
    assert x.shape == (512, 1, 28, 28) and x.dtype == np.uint8 and y.dtype == np.int64
    assert np.array_equal(x[:, 0], ref["train"][0]) and np.array_equal(y, ref["train"][1])
    xt, yt = datasets.load_mnist(str(tmp_path), train=False)
    assert xt.shape == (128, 1, 28, 28) and np.array_equal(yt, ref["t10k"][1])


def test_idx_reader_errors(tmp_path):
    from gq_amd import datasets
    p = str(tmp_path / "bad")
    open(p, "wb").write(b"\x00\x01\x08\x01" + struct.pack(">I", 4) + b"abcd")
    with pytest.raises(ValueError):
        datasets.read_idx(p)
    open(p, "wb").write(b"\x00\x00\x0d\x01" + struct.pack(">I", 1) + b"abcd")      # float element type
    with pytest.raises(ValueError):
        datasets.read_idx(p)
    open(p, "wb").write(b"\x00\x00\x08\x01" + struct.pack(">I", 9) + b"abcd")      # truncated
    with pytest.raises(ValueError):
        datasets.read_idx(p)
    with pytest.raises(FileNotFoundError):
        datasets.load_mnist(str(tmp_path / "nowhere"))
    with pytest.raises(FileNotFoundError):
        datasets.load_cifar10(str(tmp_path / "nowhere"))


def test_cifar_pickle_reader_and_batches(tmp_path):
    from gq_amd import datasets
    ref = make_cifar(str(tmp_path))
    x, y = datasets.load_cifar10(str(tmp_path), train=True)
    assert x.shape == (256, 3, 32, 32) and np.array_equal(x, ref["data_batch_1"][0]) and np.array_equal(y, ref["data_batch_1"][1])
    ds = datasets.OnDiskClassification("cifar10", str(tmp_path), "cpu", train=False)
    seen = 0
    for xb, yb in ds.batches(24, 0, shuffle=False):       # evaluation order: everything, last batch short
        assert xb.dtype == torch.float32 and xb.shape[1:] == (3, 32, 32)
        k = xb.shape[0]
        want = (torch.from_numpy(ref["test_batch"][0][seen:seen + k]).float() / 255.0
                - torch.tensor(datasets.CIFAR_MEAN).view(1, 3, 1, 1)) / torch.tensor(datasets.CIFAR_STD).view(1, 3, 1, 1)
        assert torch.equal(xb, want) and np.array_equal(yb.numpy(), ref["test_batch"][1][seen:seen + k])
        seen += k
    assert seen == 64
    # training batches: two ranks see disjoint halves of every global batch; augmentation keeps the value set
    tr = datasets.OnDiskClassification("cifar10", str(tmp_path), "cpu", train=True)
    a = list(tr.batches(16, 5, rank=0, world=2))
    b = list(tr.batches(16, 5, rank=1, world=2))
    assert len(a) == len(b) == 256 // 32 and all(x.shape == (16, 3, 32, 32) for x, _ in a + b)
    g = torch.Generator().manual_seed(5)
    perm = torch.randperm(256, generator=g)
    assert np.array_equal(a[0][1].numpy(), ref["data_batch_1"][1][perm[:16].numpy()])
    assert np.array_equal(b[0][1].numpy(), ref["data_batch_1"][1][perm[16:32].numpy()])
    # a crop of the zero-padded image, possibly mirrored: every output row is a shifted (mirrored) input row or padding
    x0 = a[0][0][0] * torch.tensor(datasets.CIFAR_STD).view(3, 1, 1) + torch.tensor(datasets.CIFAR_MEAN).view(3, 1, 1)
    src = torch.nn.functional.pad(torch.from_numpy(ref["data_batch_1"][0][perm[0]]).float() / 255.0, (4, 4, 4, 4))
    found = False
    for dy in range(9):
        for dx in range(9):
            crop = src[:, dy:dy + 32, dx:dx + 32]
            found = found or torch.allclose(x0, crop, atol=1e-6) or torch.allclose(x0, crop.flip(2), atol=1e-6)
    assert found


@pytest.mark.gpu
@pytest.mark.parametrize("dataset", ["mnist", "cifar10"])
def test_driver_one_epoch_on_disk_data_logs_loss_and_accuracy(tmp_path, dataset):
    """one epoch of the driver on a small on-disk sample (real file formats, no torchvision) through
    the HSQ quantizer on the GPU; every log line carries the loss and the test accuracy (main.py:197-211, 236-255),
    and the model has learnt the (easy) labels."""
    import json
    from gq_amd import driver
    if dataset == "mnist":
        make_mnist(str(tmp_path), n_train=2048, n_test=256)
        net = "fcn"
    else:
        make_cifar(str(tmp_path), n_train=1024, n_test=128)
        net = "resnet50"
    log = str(tmp_path / "log.jsonl")
    argv = ["--network", net, "--dataset", dataset, "--quantizer", "hsq", "--c-dim", "16", "--k-bit", "8", "--n-bit", "6",
            "--num-users", "2", "--batch-size", "16", "--epochs", "2" if dataset == "mnist" else "1", "--lr", "0.05",
            "--data", "disk", "--data-root", str(tmp_path), "--log-interval", "8", "--logfile", log, "--test-batch-size", "100"]
    args = driver.build_parser().parse_args(argv)
    torch.manual_seed(0)
    model, q, hist = driver.train(args)
    lines = [json.loads(ln) for ln in open(log)]
    assert lines and all("loss" in r and "accuracy(%)" in r and "test_loss" in r for r in lines)
    assert type(q.codecs[0]).__name__ == "HSQCodec"
    if dataset == "mnist":
        assert lines[-1]["accuracy(%)"] > 80.0, lines[-1]


def test_rank_slices_keep_the_short_last_batch_like_the_reference_loader():
    """The reference's DataLoader has drop_last=False and main.py:189-193 gives every user len // num_users samples, the
    last user the rest.  rank_slices does the same over ranks: every sample of the epoch is used exactly once, all ranks
    take the same number of steps, a last batch with fewer samples than ranks is skipped everywhere."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gradient-quantization_amd"))
    from gq_amd.datasets import rank_slices
    for n, batch, world in ((1000, 32, 4), (1003, 32, 4), (96, 32, 3), (97, 32, 3), (130, 64, 2), (10, 4, 1), (9, 4, 4)):
        per_rank = [list(rank_slices(n, batch, r, world)) for r in range(world)]
        assert len(set(len(p) for p in per_rank)) == 1, "ranks would take different numbers of steps"
        seen = sorted(i for p in per_rank for lo, hi in p for i in range(lo, hi))
        full, rest = divmod(n, batch * world)
        expect = n if rest == 0 or rest // world > 0 else full * batch * world
        assert seen == list(range(expect)), (n, batch, world)
        for step in range(len(per_rank[0])):
            sizes = [per_rank[r][step][1] - per_rank[r][step][0] for r in range(world)]
            assert len(set(sizes[:-1])) <= 1 and sizes[-1] >= sizes[0] and sizes[-1] - sizes[0] < world


def test_rank_slices_drop_a_short_batch_on_every_rank_or_on_none():
    """ADVICE r3 (driver.py): with several users per rank a short last batch must be skipped by ALL ranks or by none -- the
    last rank's share is larger (len // world + len % world), so a per-rank test `my samples < num_users` let it run a
    step, and enter the exchange's collectives, alone (n = 11, batch 2 x 2 users, 2 ranks: rank 0 held 1 sample, rank 1 two)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gradient-quantization_amd"))
    from gq_amd.datasets import rank_slices
    for n, batch, users, world in ((11, 2, 2, 2), (11, 1, 2, 2), (23, 4, 4, 3), (100, 8, 2, 4), (37, 3, 3, 2), (9, 2, 2, 4)):
        per_rank = [list(rank_slices(n, batch * users, r, world, min_share=users)) for r in range(world)]
        assert len(set(len(p) for p in per_rank)) == 1, ("ranks would take different numbers of steps", n, batch, users, world)
        for step in range(len(per_rank[0])):
            for r in range(world):
                lo, hi = per_rank[r][step]
                assert hi - lo >= users, "a rank holds fewer samples than users in a step that runs"
        # nothing is dropped that could have run: the only batch that may go is the short last one
        used = sum(hi - lo for p in per_rank for lo, hi in p)
        rest = n % (batch * users * world)
        assert used == (n if rest == 0 or rest // world >= users else n - rest)
