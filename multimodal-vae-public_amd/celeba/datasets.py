"""Drop-in for the reference's ``celeba/datasets.py``: the one-time parsing of CelebA's partition and
attribute files (celeba/datasets.py:93-135), the ``CelebAttributes`` dataset surface (:37-90) and
``tensor_to_attributes`` (:138-152) -- plus ``CelebaLoader``, which feeds the fused step: JPEGs are
decoded on the host (Pillow), the uint8 batch goes to HBM once and ``Resize(64) + CenterCrop(64) +
ToTensor`` (celeba/train.py:146-148) runs as ONE HIP launch per batch (``preprocess.py``), byte-exact
with the per-image torchvision transforms.

Only host-side integer / string work lives here; it follows the reference's semantics exactly:
  * ``Eval/list_eval_partition.txt``: "<file> <0|1|2>" per line, partitions train / val / test;
  * ``Anno/list_attr_celeba.txt``: two header lines, then "<file> <40 values in {-1, 1}>"; rows are kept
    in FILE order when their file belongs to the partition, -1 becomes 0;
  * a cached ``Anno/attr_<partition>.npy`` wins over the text file;
  * the 18 visually distinctive attributes of ``ATTR_IX_TO_KEEP`` are returned as float32 {0, 1}.
"""
import os

import numpy as np
import torch

VALID_PARTITIONS = {'train': 0, 'val': 1, 'test': 2}
# the 40 attribute names in the column order of list_attr_celeba.txt (dataset metadata)
CELEBA_ATTR_NAMES = (
    '5_o_Clock_Shadow', 'Arched_Eyebrows', 'Attractive', 'Bags_Under_Eyes', 'Bald', 'Bangs', 'Big_Lips', 'Big_Nose',
    'Black_Hair', 'Blond_Hair', 'Blurry', 'Brown_Hair', 'Bushy_Eyebrows', 'Chubby', 'Double_Chin', 'Eyeglasses',
    'Goatee', 'Gray_Hair', 'Heavy_Makeup', 'High_Cheekbones', 'Male', 'Mouth_Slightly_Open', 'Mustache',
    'Narrow_Eyes', 'No_Beard', 'Oval_Face', 'Pale_Skin', 'Pointy_Nose', 'Receding_Hairline', 'Rosy_Cheeks',
    'Sideburns', 'Smiling', 'Straight_Hair', 'Wavy_Hair', 'Wearing_Earrings', 'Wearing_Hat', 'Wearing_Lipstick',
    'Wearing_Necklace', 'Wearing_Necktie', 'Young')
ATTR_TO_IX_DICT = {name: ix for ix, name in enumerate(CELEBA_ATTR_NAMES)}
IX_TO_ATTR_DICT = {ix: name for name, ix in ATTR_TO_IX_DICT.items()}
# the 18 attributes the MVAE models (celeba/datasets.py:33)
ATTR_IX_TO_KEEP = [4, 5, 8, 9, 11, 12, 15, 17, 18, 20, 21, 22, 26, 28, 31, 32, 33, 35]
N_ATTRS = len(ATTR_IX_TO_KEEP)
ATTR_TO_PLOT = ['Heavy_Makeup', 'Male', 'Mouth_Slightly_Open', 'Smiling', 'Wavy_Hair']


def load_eval_partition(partition, data_dir='./data'):
    """File names of one partition, in the order of Eval/list_eval_partition.txt (celeba/datasets.py:93-108)."""
    want = VALID_PARTITIONS[partition]
    names = []
    with open(os.path.join(data_dir, 'Eval/list_eval_partition.txt')) as fp:
        for row in fp:
            row = row.strip()
            if not row:
                continue
            path, label = row.split(' ')
            if int(label) == want:
                names.append(path)
    return names


def load_attributes(paths, partition, data_dir='./data'):
    """float32 [n, 18] attribute matrix of a partition (celeba/datasets.py:111-135).  Rows follow the
    ATTRIBUTE FILE's order (the reference filters the file by membership in ``paths``), values are
    {0, 1} (-1 -> 0), columns are ``ATTR_IX_TO_KEEP``."""
    cached = os.path.join(data_dir, 'Anno/attr_%s.npy' % partition)
    if os.path.isfile(cached):
        attr_data = np.load(cached)
    else:
        wanted = set(paths)                 # the reference tests `path in paths` on a list: same result, O(1)
        rows = []
        with open(os.path.join(data_dir, 'Anno/list_attr_celeba.txt')) as fp:
            for ix, row in enumerate(fp):
                if ix < 2:                  # image count, attribute names
                    continue
                fields = row.strip().split()
                if not fields or fields[0] not in wanted:
                    continue
                values = np.array(fields[1:]).astype(int)
                values[values < 0] = 0
                rows.append(values)
        if not rows:
            raise ValueError('no attribute rows of partition %r under %s' % (partition, data_dir))
        attr_data = np.vstack(rows).astype(np.int64)
    return torch.from_numpy(np.asarray(attr_data)).float()[:, ATTR_IX_TO_KEEP]


def tensor_to_attributes(tensor):
    """Names of the attributes whose (rounded) value exceeds 0.5 (celeba/datasets.py:138-152)."""
    tensor = torch.round(tensor)
    return [IX_TO_ATTR_DICT[ATTR_IX_TO_KEEP[i]] for i in range(tensor.size(0)) if tensor[i] > 0.5]


class CelebAttributes(object):
    """The reference's Dataset surface (celeba/datasets.py:37-90): ``dataset[i] -> (image, attrs)`` with the
    optional per-item transforms applied on the host.  The fused step does not go through it -- see
    ``CelebaLoader`` -- but ``sample.py``-style code that indexes single items does."""

    def __init__(self, partition='train', data_dir='./data', image_transform=None, attr_transform=None):
        if partition not in VALID_PARTITIONS:
            raise AssertionError(partition)
        self.partition, self.data_dir = partition, data_dir
        self.image_transform, self.attr_transform = image_transform, attr_transform
        self.image_paths = load_eval_partition(partition, data_dir=data_dir)
        self.attr_data = load_attributes(self.image_paths, partition, data_dir=data_dir)
        self.size = int(len(self.image_paths))

    def load_rgb(self, index):
        from PIL import Image
        path = os.path.join(self.data_dir, 'img_align_celeba', self.image_paths[index])
        return Image.open(path).convert('RGB')

    def __getitem__(self, index):
        image, attr = self.load_rgb(index), self.attr_data[index]
        if self.image_transform is not None:
            image = self.image_transform(image)
        if self.attr_transform is not None:
            attr = self.attr_transform(attr)
        return image, attr

    def __len__(self):
        return self.size


class CelebaLoader(object):
    """(image float32 [B, 3, 64, 64], attrs float32 [B, 18]) batches for the CelebA / CelebA-19 train loops:
    the DataLoader + ``Compose([Resize(64), CenterCrop(64), ToTensor()])`` of celeba/train.py:146-156 with the
    transform on the GPU.  Per batch: Pillow decodes the JPEGs (host threads), ONE uint8 NHWC copy to HBM, ONE
    resize+crop+scale launch.  All images of a batch must share a size (aligned CelebA: 218 x 178).
    ``rank`` / ``world``: data-parallel sharding of one shared permutation, like ``IdxLoader``."""

    def __init__(self, partition, data_dir, batch_size, shuffle, device, seed=0, rank=0, world=1, size=64,
                 decode_threads=8):
        from ..preprocess import ResizeCenterCropToTensor
        self.data = CelebAttributes(partition, data_dir)
        self.batch_size, self.shuffle, self.device = int(batch_size), bool(shuffle), device
        self.rank, self.world = int(rank), int(world)
        from ..train_common import shard_len
        self.dataset = range(shard_len(len(self.data), self.world))        # samples per rank, equal on all ranks
        self._gen = torch.Generator().manual_seed(seed)
        self._transform = ResizeCenterCropToTensor(size)
        self._threads = max(1, int(decode_threads))

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _decode(self, index):
        return np.asarray(self.data.load_rgb(index), dtype=np.uint8)

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        n_total = len(self.data)
        order = torch.randperm(n_total, generator=self._gen) if self.shuffle else torch.arange(n_total)
        from ..train_common import shard_order
        order = shard_order(order, self.rank, self.world).tolist()
        with ThreadPoolExecutor(self._threads) as pool:
            for i in range(0, len(order), self.batch_size):
                idx = order[i:i + self.batch_size]
                frames = list(pool.map(self._decode, idx))
                if any(f.shape != frames[0].shape for f in frames):
                    raise ValueError('images of one batch differ in size: %s' % sorted({f.shape for f in frames}))
                u8 = torch.from_numpy(np.stack(frames))
                if torch.cuda.is_available():
                    u8 = u8.pin_memory()
                image = self._transform(u8.to(self.device, non_blocking=True))
                yield image, self.data.attr_data[idx].to(self.device, non_blocking=True)
