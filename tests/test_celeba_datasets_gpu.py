"""GPU: CelebaLoader -- host JPEG decode, ONE uint8 copy + ONE resize/crop/scale launch per batch -- yields what
the reference's DataLoader + Compose([Resize(64), CenterCrop(64), ToTensor()]) yields (celeba/train.py:146-156),
checked against the per-image Pillow arithmetic of oracle/preprocess.py (byte-exact)."""
import os

import numpy as np
import pytest
import torch

import mvae_amd  # noqa: F401
from mvae_amd.celeba import datasets as D
from oracle import preprocess as OP
from test_celeba_datasets_cpu import make_tree

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip('PIL.Image')


def test_loader_batches_match_the_per_image_transform(tmp_path):
    root = str(tmp_path)
    names, part, attrs = make_tree(root, n=21, seed=3)
    os.makedirs(os.path.join(root, 'img_align_celeba'))
    rng = np.random.RandomState(5)
    for nm in names:
        PIL.fromarray(rng.randint(0, 256, size=(218, 178, 3), dtype=np.uint8)).save(
            os.path.join(root, 'img_align_celeba', nm[:-4] + '.png'))
    # lossless files so that the decoded bytes are known: rename the list entries to .png
    for rel in ('Eval/list_eval_partition.txt', 'Anno/list_attr_celeba.txt'):
        path = os.path.join(root, rel)
        text = open(path).read().replace('.jpg', '.png')
        open(path, 'w').write(text)
    loader = D.CelebaLoader('train', root, batch_size=4, shuffle=False, device=torch.device('cuda'))
    ds = D.CelebAttributes('train', root)
    seen = 0
    for image, attr in loader:
        assert image.dtype == torch.float32 and image.shape[1:] == (3, 64, 64) and attr.shape[1] == 18
        for i in range(image.shape[0]):
            u8 = np.asarray(ds.load_rgb(seen + i), dtype=np.uint8)
            ref = OP.resize_center_crop_to_tensor(u8, 64)
            assert np.array_equal(image[i].cpu().numpy(), ref), 'image %d differs from the per-image transform' % (seen + i)
            assert torch.equal(attr[i].cpu(), ds.attr_data[seen + i])
        seen += image.shape[0]
    assert seen == len(ds) and len(loader) == (len(ds) + 3) // 4
    # two data-parallel ranks cover the partition exactly once
    a = D.CelebaLoader('train', root, 4, True, torch.device('cuda'), seed=9, rank=0, world=2)
    b = D.CelebaLoader('train', root, 4, True, torch.device('cuda'), seed=9, rank=1, world=2)
    assert sum(x.shape[0] for x, _ in a) + sum(x.shape[0] for x, _ in b) == len(ds)
