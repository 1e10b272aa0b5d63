# round 6: end-to-end training iterations (main.py:216-233: forward, backward, record per user, apply, step) with the README's
# commands on synthetic data of the datasets' shapes; JSONL -> gpurun_out/r06/train_*.jsonl.  ITERS iterations, a log line per 100.
mkdir -p gpurun_out/r06
ITERS=${ITERS:-600}
for net in "fcn mnist" "resnet50 cifar10"; do
  set -- $net
  for q in "sgd" "hsq --c-dim 16 --k-bit 8 --n-bit 6" "qsgd --c-dim 128 --n-bit 2"; do
    tag=$1_$(echo $q | cut -d" " -f1)
    rm -f gpurun_out/r06/train_$tag.jsonl
    python train.py --quantizer $q --network $1 --dataset $2 --num-users 8 --batch-size 32 --epochs 1 --train-size $((256 * ITERS)) \
        --log-interval 100 --timing --logfile gpurun_out/r06/train_$tag.jsonl > /dev/null 2> gpurun_out/r06/train_$tag.err
    tail -1 gpurun_out/r06/train_$tag.jsonl
  done
done
# one user per step (what one rank of an 8-GPU job runs): batch 32
for q in "sgd" "hsq --c-dim 16 --k-bit 8 --n-bit 6" "qsgd --c-dim 128 --n-bit 2"; do
    tag=resnet50_u1_$(echo $q | cut -d" " -f1)
    rm -f gpurun_out/r06/train_$tag.jsonl
    python train.py --quantizer $q --network resnet50 --dataset cifar10 --num-users 1 --batch-size 32 --epochs 1 --train-size $((32 * ITERS)) \
        --log-interval 100 --timing --logfile gpurun_out/r06/train_$tag.jsonl > /dev/null 2> gpurun_out/r06/train_$tag.err
    tail -1 gpurun_out/r06/train_$tag.jsonl
done
