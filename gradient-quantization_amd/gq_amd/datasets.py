"""On-disk datasets for the training driver, without torchvision (the reference's dataloaders.py:5-47 needs
torchvision and a download; neither exists on the GPU boxes).

Readers for the two formats the reference's README commands use, in the directory layout torchvision
leaves behind (so a `./data` directory filled by the reference works as it is):

    MNIST      <root>/MNIST/raw/{train,t10k}-{images-idx3,labels-idx1}-ubyte[.gz]     (or directly in <root>)
    CIFAR-10   <root>/cifar-10-batches-py/{data_batch_1..5,test_batch}                (python pickles)

The images stay uint8 in HBM; a batch is normalised on the device with the reference's constants
(dataloaders.py:10-11, 27-28) and, for CIFAR-10 training batches, augmented as the reference does
(RandomCrop(32, padding=4) + RandomHorizontalFlip, dataloaders.py:24-25) -- vectorised over the batch.
"""
import gzip
import os
import pickle
import struct

import numpy as np
import torch

MNIST_MEAN, MNIST_STD = (0.1307,), (0.3081,)
CIFAR_MEAN, CIFAR_STD = (0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010)


def read_idx(path):
    """One IDX file (MNIST's format: magic 0x0000 <dtype 0x08> <ndim>, big-endian dims, raw bytes); .gz transparently."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        head = f.read(4)
        if len(head) != 4 or head[0] != 0 or head[1] != 0:
            raise ValueError("%s: not an IDX file (bad magic)" % path)
        if head[2] != 0x08:
            raise ValueError("%s: IDX element type 0x%02x is not unsigned byte" % (path, head[2]))
        dims = struct.unpack(">" + "I" * head[3], f.read(4 * head[3]))
        data = np.frombuffer(f.read(), dtype=np.uint8)
    n = int(np.prod(dims)) if dims else 0
    if data.size != n:
        raise ValueError("%s: %d data bytes for dimensions %s" % (path, data.size, dims))
    return data.reshape(dims)


def _find(root, names):
    for d in (root, os.path.join(root, "MNIST", "raw"), os.path.join(root, "raw")):
        for n in names:
            for suffix in ("", ".gz"):
                p = os.path.join(d, n + suffix)
                if os.path.exists(p):
                    return p
    raise FileNotFoundError("none of %s found under %s (also looked in MNIST/raw)" % (names, root))


def load_mnist(root, train=True):
    """-> (images uint8 [N, 1, 28, 28], labels int64 [N])."""
    stem = "train" if train else "t10k"
    x = read_idx(_find(root, [stem + "-images-idx3-ubyte", stem + "-images.idx3-ubyte"]))
    y = read_idx(_find(root, [stem + "-labels-idx1-ubyte", stem + "-labels.idx1-ubyte"]))
    if x.ndim != 3 or y.ndim != 1 or x.shape[0] != y.shape[0]:
        raise ValueError("MNIST files under %s do not belong together: images %s, labels %s" % (root, x.shape, y.shape))
    return x[:, None, :, :].copy(), y.astype(np.int64)


def load_cifar10(root, train=True):
    """-> (images uint8 [N, 3, 32, 32], labels int64 [N]) from the python-pickle batches."""
    d = os.path.join(root, "cifar-10-batches-py")
    if not os.path.isdir(d):
        d = root
    names = ["data_batch_%d" % i for i in range(1, 6)] if train else ["test_batch"]
    xs, ys = [], []
    for n in names:
        p = os.path.join(d, n)
        if not os.path.exists(p):
            if train and xs:      # a partial copy (tests ship one batch): use what is there
                continue
            raise FileNotFoundError(p)
        with open(p, "rb") as f:
            b = pickle.load(f, encoding="bytes")
        data = b[b"data"] if b"data" in b else b["data"]
        labels = b[b"labels"] if b"labels" in b else b["labels"]
        xs.append(np.asarray(data, dtype=np.uint8).reshape(-1, 3, 32, 32))
        ys.append(np.asarray(labels, dtype=np.int64))
    return np.concatenate(xs), np.concatenate(ys)


LOADERS = {"mnist": (load_mnist, MNIST_MEAN, MNIST_STD, False), "cifar10": (load_cifar10, CIFAR_MEAN, CIFAR_STD, True)}


class OnDiskClassification(object):
    """A whole split resident on the device as uint8; batches are normalised (and augmented) there."""

    def __init__(self, dataset, root, device, train=True):
        load, mean, std, augment = LOADERS[dataset]
        x, y = load(root, train)
        self.x = torch.from_numpy(x).to(device)
        self.y = torch.from_numpy(y).to(device)
        self.n = int(self.x.shape[0])
        self.mean = torch.tensor(mean, dtype=torch.float32, device=device).view(1, -1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32, device=device).view(1, -1, 1, 1)
        self.augment = augment and train

    def _prepare(self, xb, gen):
        xb = xb.float().div_(255.0)                       # transforms.ToTensor
        if self.augment:
            n, c, hgt, wid = xb.shape
            pad = torch.nn.functional.pad(xb, (4, 4, 4, 4))
            dy = torch.randint(0, 9, (n,), generator=gen, device="cpu").to(xb.device)
            dx = torch.randint(0, 9, (n,), generator=gen, device="cpu").to(xb.device)
            rows = dy.view(n, 1) + torch.arange(hgt, device=xb.device).view(1, hgt)
            cols = dx.view(n, 1) + torch.arange(wid, device=xb.device).view(1, wid)
            flip = (torch.rand(n, generator=gen, device="cpu") < 0.5).to(xb.device)
            cols = torch.where(flip.view(n, 1), cols.flip(1), cols)
            idx_n = torch.arange(n, device=xb.device).view(n, 1, 1, 1)
            idx_c = torch.arange(c, device=xb.device).view(1, c, 1, 1)
            xb = pad[idx_n, idx_c, rows.view(n, 1, hgt, 1), cols.view(n, 1, 1, wid)]
        return (xb - self.mean) / self.std                 # transforms.Normalize

    def batches(self, batch, epoch_seed, rank=0, world=1, shuffle=True, min_share=1):
        """This rank's share of every global batch (world * batch consecutive entries of the epoch's permutation, as the
        reference's loader yields batch_size * num_users and main.py:189-193 splits it).  The loader's drop_last is False
        (dataloaders.py: DataLoader defaults): the short last batch is kept and split the same way -- see rank_slices.
        min_share: samples every rank needs of a batch (one per user); a short batch that cannot give that to EVERY rank is
        dropped on ALL ranks, so that the ranks always agree on whether a step -- and its collectives -- happens."""
        gen = torch.Generator().manual_seed(epoch_seed)
        order = torch.randperm(self.n, generator=gen) if shuffle else torch.arange(self.n)
        order = order.to(self.x.device)
        for lo, hi in rank_slices(self.n, batch, rank, world, min_share):
            idx = order[lo:hi]
            yield self._prepare(self.x[idx], gen), self.y[idx]


def rank_slices(n, batch, rank, world, min_share=1):
    """[lo, hi) of this rank's samples in every global batch of `world * batch` consecutive entries.  A full batch gives
    every rank `batch`.  The short last one (n % (world * batch) samples; DataLoader keeps it, drop_last=False) is split
    like main.py:189-193 splits a batch over its users: every rank but the last gets len // world, the last rank the
    rest; when the smallest share (len // world) is below `min_share` it is skipped on EVERY rank -- decided from the
    global size, never from a rank's own share: the last rank's share is larger, and a rank that ran the step alone would
    enter the exchange's collectives alone."""
    per = batch * world
    for i in range(0, n, per):
        size = min(per, n - i)
        share = size // world
        if share < max(1, min_share):
            continue
        lo = i + rank * share
        hi = i + size if rank == world - 1 else lo + share
        yield lo, hi
