"""CPU: the dependency-free IDX reader behind the MNIST / FashionMNIST loaders (mnist/train.py:159-165
reads the same files through torchvision)."""
import gzip
import struct

import numpy as np
import pytest

import mvae_amd  # noqa: F401
from mvae_amd.train_common import _find_idx, read_idx


def write_idx(path, arr, gz=False):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    head = struct.pack('>HBB', 0, 0x08, arr.ndim) + struct.pack('>' + 'I' * arr.ndim, *arr.shape)
    (gzip.open if gz else open)(path, 'wb').write(head + arr.tobytes())


def test_read_idx_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    imgs = rng.randint(0, 256, (7, 28, 28)).astype(np.uint8)
    lbls = rng.randint(0, 10, (7,)).astype(np.uint8)
    write_idx(str(tmp_path / 'train-images-idx3-ubyte'), imgs)
    write_idx(str(tmp_path / 'train-labels-idx1-ubyte.gz'), lbls, gz=True)
    assert np.array_equal(read_idx(str(tmp_path / 'train-images-idx3-ubyte')), imgs)
    assert np.array_equal(read_idx(str(tmp_path / 'train-labels-idx1-ubyte.gz')), lbls)
    assert _find_idx(str(tmp_path), 'train-labels-idx1-ubyte').endswith('.gz')
    assert _find_idx(str(tmp_path), 't10k-images-idx3-ubyte') is None


def test_read_idx_rejects_garbage(tmp_path):
    p = str(tmp_path / 'bad')
    open(p, 'wb').write(b'\x00\x00\x0d\x03' + b'\x00' * 40)       # float32 type code
    with pytest.raises(ValueError):
        read_idx(p)
    write_idx(p, np.zeros((3, 4, 4)))
    open(p, 'ab').write(b'\x01')                                    # one byte too many
    with pytest.raises(ValueError):
        read_idx(p)


@pytest.mark.parametrize('n,world,batch', [(10, 3, 3), (9, 2, 4), (8, 2, 4), (7, 4, 1), (5, 1, 2), (60000, 8, 512)])
def test_every_rank_gets_the_same_number_of_batches(n, world, batch):
    """``order[rank::world]`` gave rank 0 one sample -- sometimes one BATCH -- more than the last rank;
    the extra batch's all-reduces have no peer (deadlock).  The shared permutation is padded by wrapping, like
    DistributedSampler: equal shares, every sample still seen once per epoch."""
    import torch
    from mvae_amd.train_common import shard_len, shard_order
    order = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    shares = [shard_order(order, r, world) for r in range(world)]
    assert len({int(s.numel()) for s in shares}) == 1
    assert shares[0].numel() == shard_len(n, world)
    n_batches = {(int(s.numel()) + batch - 1) // batch for s in shares}
    assert len(n_batches) == 1
    seen = torch.cat(shares)
    assert set(seen.tolist()) == set(range(n)) and seen.numel() - n < world
    assert shard_order(list(range(n)), 0, world)[:2] == list(range(n))[0::world][:2]      # plain lists too
