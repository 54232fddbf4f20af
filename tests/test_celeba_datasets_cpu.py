"""CPU: the CelebA partition / attribute parsing of the drop-in (celeba/datasets.py:93-135) on a small
synthetic dataset tree in the real file formats, against a literal restatement of the reference's rules."""
import os

import numpy as np
import pytest
import torch

import mvae_amd  # noqa: F401
from mvae_amd.celeba import datasets as D


def make_tree(root, n=23, seed=0):
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'Eval')); os.makedirs(os.path.join(root, 'Anno'))
    names = ['%06d.jpg' % (i + 1) for i in range(n)]
    part = rng.randint(0, 3, size=n)
    with open(os.path.join(root, 'Eval/list_eval_partition.txt'), 'w') as f:
        for nm, p in zip(names, part):
            f.write('%s %d\n' % (nm, p))
    attrs = rng.choice([-1, 1], size=(n, 40))
    with open(os.path.join(root, 'Anno/list_attr_celeba.txt'), 'w') as f:
        f.write('%d\n' % n)
        f.write(' '.join(D.CELEBA_ATTR_NAMES) + '\n')
        for nm, row in zip(names, attrs):
            f.write('%s  %s\n' % (nm, ' '.join('%2d' % v for v in row)))     # the real file pads with spaces
    return names, part, attrs


def reference_rules(names, part, attrs, partition):
    """celeba/datasets.py:101-134 restated literally: membership filter in attribute-file order, -1 -> 0,
    int64 -> float, the 18 kept columns."""
    want = {'train': 0, 'val': 1, 'test': 2}[partition]
    paths = [nm for nm, p in zip(names, part) if p == want]
    rows = []
    for nm, row in zip(names, attrs):
        if nm in paths:
            r = np.array(row).astype(int); r[r < 0] = 0
            rows.append(r)
    data = torch.from_numpy(np.vstack(rows).astype(np.int64)).float()
    keep = [4, 5, 8, 9, 11, 12, 15, 17, 18, 20, 21, 22, 26, 28, 31, 32, 33, 35]
    return paths, data[:, keep]


@pytest.mark.parametrize('partition', ['train', 'val', 'test'])
def test_partition_and_attribute_parsing(tmp_path, partition):
    names, part, attrs = make_tree(str(tmp_path))
    ref_paths, ref_attr = reference_rules(names, part, attrs, partition)
    paths = D.load_eval_partition(partition, data_dir=str(tmp_path))
    assert paths == ref_paths
    got = D.load_attributes(paths, partition, data_dir=str(tmp_path))
    assert got.dtype == torch.float32 and got.shape == (len(ref_paths), 18) == (len(paths), D.N_ATTRS)
    assert torch.equal(got, ref_attr) and set(got.unique().tolist()) <= {0.0, 1.0}
    ds = D.CelebAttributes(partition, str(tmp_path))
    assert len(ds) == len(ref_paths) and torch.equal(ds.attr_data, ref_attr)


def test_cached_npy_wins_and_names(tmp_path):
    names, part, attrs = make_tree(str(tmp_path))
    cached = np.zeros((3, 40), dtype=np.int64); cached[1, 20] = 1; cached[2, 35] = 1
    np.save(os.path.join(str(tmp_path), 'Anno/attr_train.npy'), cached)
    got = D.load_attributes(['ignored'], 'train', data_dir=str(tmp_path))
    assert got.shape == (3, 18) and got[1, D.ATTR_IX_TO_KEEP.index(20)] == 1 and got.sum() == 2
    assert D.tensor_to_attributes(got[1]) == ['Male'] and D.tensor_to_attributes(got[2]) == ['Wearing_Hat']
    assert D.tensor_to_attributes(torch.full((18,), 0.49)) == []
    assert [D.IX_TO_ATTR_DICT[i] for i in D.ATTR_IX_TO_KEEP][:4] == ['Bald', 'Bangs', 'Black_Hair', 'Blond_Hair']
    assert D.ATTR_TO_IX_DICT['Young'] == 39 and D.ATTR_TO_IX_DICT['Smiling'] == 31 and len(D.ATTR_TO_IX_DICT) == 40
    with pytest.raises(AssertionError):
        D.CelebAttributes('dev', str(tmp_path))
    with pytest.raises(KeyError):
        D.load_eval_partition('dev', data_dir=str(tmp_path))
