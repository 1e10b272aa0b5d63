#!/usr/bin/env python3
"""CLI of the training driver (gq_amd/driver.py): the reference's main.py flags, synthetic data.

    python train.py --quantizer hsq --network fcn --dataset mnist --c-dim 16 --k-bit 8 --n-bit 6 --num-users 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --num-users 1 --network resnet50 --dataset cifar10 ...
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gradient-quantization_amd"))
from gq_amd.driver import main  # noqa: E402

if __name__ == "__main__":
    main(sys.argv[1:])
