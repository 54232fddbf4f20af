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
