"""Epoch driver shared by the four ``train.py`` drop-ins: the reference's ``__main__`` block
(mnist/train.py:132-268) with the per-batch body replaced by the fused HIP step.

Kept from the reference: every CLI flag and default, the KL-annealing schedule
(mnist/train.py:180-186; fashionmnist's is one epoch ahead, fashionmnist/train.py:182), the log
lines, AverageMeter (mnist/train.py:97-112), the checkpoint dict
{'state_dict','best_loss','n_latents','optimizer'} (mnist/train.py:263-268) and
load_checkpoint (mnist/train.py:124-129).  Opt-in additions: ``--synthetic`` (random-pixel /
random-label batches, SURVEY.md section 8d -- the image has no datasets and no network),
``--steps-per-epoch``, ``--no-graph``, and the ``RANK/WORLD_SIZE`` environment of
``torch.distributed.run`` for data-parallel replicas.
"""
import os
import shutil
import sys

import torch


class AverageMeter(object):
    """Computes and stores the average and current value (mnist/train.py:97-112)."""
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def save_checkpoint(state, is_best, folder='./', filename='checkpoint.pth.tar'):
    if not os.path.isdir(folder):
        os.mkdir(folder)
    torch.save(state, os.path.join(folder, filename))
    if is_best:
        shutil.copyfile(os.path.join(folder, filename),
                        os.path.join(folder, 'model_best.pth.tar'))


def make_load_checkpoint(mvae_cls):
    def load_checkpoint(file_path, use_cuda=False):
        checkpoint = torch.load(file_path) if use_cuda else \
            torch.load(file_path, map_location=lambda storage, location: storage)
        model = mvae_cls(checkpoint['n_latents'])
        model.load_state_dict(checkpoint['state_dict'])
        return model
    return load_checkpoint


# The reference's command lines as data: (flag, type, default, how the default is printed, metavar, text).
# mnist/train.py:133-154 (fashionmnist: same), celeba/train.py:119-140, celeba19/train.py:181-204.
_SMALL = dict(n_latents=(64, '64'), epochs=(500, '500'), annealing=(200, '200'), lr=(1e-3, '1e-3'))
_CELEBA = dict(n_latents=(100, '100'), epochs=(100, '100'), annealing=(20, '20'), lr=(1e-4, '1e-4'))


def _reference_flags(kind):
    d = _SMALL if kind in ('mnist', 'fashionmnist') else _CELEBA
    label = 'text' if kind in ('mnist', 'fashionmnist') else 'attrs'
    label_words = 'text' if label == 'text' else 'attributes'
    flags = [
        ('--n-latents', int, d['n_latents'], None, 'size of the latent embedding'),
        ('--batch-size', int, (100, '100'), 'N', 'input batch size for training'),
        ('--epochs', int, d['epochs'], 'N', 'number of epochs to train'),
        ('--annealing-epochs', int, d['annealing'], 'N', 'number of epochs to anneal KL for'),
        ('--lr', float, d['lr'], 'LR', 'learning rate'),
        ('--log-interval', int, (10, '10'), 'N', 'how many batches to wait before logging training status'),
    ]
    if kind == 'celeba19':
        flags.append(('--approx-m', int, (1, '1'), None, 'number of ELBO terms to approx. the full MVAE objective'))
    flags += [
        ('--lambda-image', float, (1., '1'), None, 'multipler for image reconstruction'),
        ('--lambda-%s' % label, float, (10., '10'), None, 'multipler for %s reconstruction' % label_words),
    ]
    return flags


def reference_parser(kind):
    """argparse parser with exactly the reference's flags, defaults and help texts for ``kind``'s
    train.py, plus the opt-in extras of ``add_extra_flags``."""
    import argparse
    parser = argparse.ArgumentParser()
    for flag, typ, (default, shown), metavar, text in _reference_flags(kind):
        kw = dict(type=typ, default=default, help='%s [default: %s]' % (text, shown))
        if metavar:
            kw['metavar'] = metavar
        parser.add_argument(flag, **kw)
    parser.add_argument('--cuda', action='store_true', default=False, help='enables CUDA training [default: False]')
    add_extra_flags(parser)
    if kind == 'celeba19':
        # SURVEY Appendix B-4, decided explicitly: every model() call of the reference runs the image decoder
        # (celeba19/model.py:52-61), also for the 18 attribute-only terms whose image output nobody reads
        # (celeba19/train.py:278-283) -- their only effect is 18 more BatchNorm running-statistics updates per step.
        parser.add_argument('--bn-stats', choices=('reference', 'loss-bearing'), default='reference',
                            help="'reference': the image decoder's BatchNorm running statistics advance for all 20 + M "
                                 "terms, as in the reference (18 decodes whose output is unused). 'loss-bearing': only the "
                                 "2 + M terms with an image loss are decoded -- same ELBO and gradients, faster step, "
                                 "running_mean / running_var (eval-mode behaviour, checkpoints) differ from the reference "
                                 "[default: reference]")
    return parser


def add_extra_flags(parser):
    parser.add_argument('--synthetic', action='store_true', default=False,
                        help='random-pixel / random-label batches instead of the dataset')
    parser.add_argument('--steps-per-epoch', type=int, default=100,
                        help='mini-batches per epoch with --synthetic [default: 100]')
    parser.add_argument('--synthetic-last-batch', type=int, default=0,
                        help='with --synthetic: make the last mini-batch of an epoch this short (dataset tail)')
    parser.add_argument('--no-graph', action='store_true', default=False,
                        help='launch kernels eagerly instead of replaying the captured hipGraph')
    parser.add_argument('--out-dir', type=str, default='./trained_models')
    parser.add_argument('--data-dir', type=str, default='./data',
                        help='where the MNIST / FashionMNIST IDX files live (without --synthetic)')


class SyntheticLoader(object):
    """len()/iteration surface of the DataLoader the reference builds (mnist/train.py:159-165)."""
    def __init__(self, kind, batch_size, n_batches, seed, device, last_batch=0):
        self.kind, self.batch_size, self.n, self.seed, self.device = kind, batch_size, n_batches, seed, device
        self.last_batch = int(last_batch)       # > 0: the final batch is this short (a dataset tail)
        self.dataset = range(batch_size * n_batches - (batch_size - self.last_batch if self.last_batch else 0))

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for i in range(self.n):
            bs = self.last_batch if (self.last_batch and i == self.n - 1) else self.batch_size
            if self.kind in ('mnist', 'fashionmnist'):
                image = torch.rand(bs, 1, 28, 28, generator=g)
                label = torch.randint(0, 10, (bs,), generator=g)
            else:
                image = torch.rand(bs, 3, 64, 64, generator=g)
                label = torch.randint(0, 2, (bs, 18), generator=g).float()
            yield image.to(self.device, non_blocking=True), label.to(self.device, non_blocking=True)


def read_idx(path):
    """Parse an IDX file (the raw MNIST / FashionMNIST distribution format, optionally .gz):
    magic 0x0000 <dtype 0x08 = uint8> <ndim>, big-endian uint32 dims, then the bytes."""
    import gzip
    import struct
    import numpy as np
    op = gzip.open if path.endswith('.gz') else open
    with op(path, 'rb') as f:
        raw = f.read()
    zero, dtype, ndim = struct.unpack('>HBB', raw[:4])
    if zero != 0 or dtype != 0x08:
        raise ValueError('%s is not a uint8 IDX file' % path)
    dims = struct.unpack('>' + 'I' * ndim, raw[4:4 + 4 * ndim])
    data = np.frombuffer(raw, dtype=np.uint8, offset=4 + 4 * ndim)
    if data.size != int(np.prod(dims)):
        raise ValueError('%s: %d bytes for dims %s' % (path, data.size, dims))
    return data.reshape(dims)


def shard_order(order, rank, world):
    """This rank's share of one shared permutation: ``order`` is padded by wrapping around to a multiple of
    ``world`` (what ``DistributedSampler`` does) and dealt out round-robin, so EVERY rank sees the same number of
    samples and therefore the same number of batches -- a rank with one batch more would enter a round of
    gradient all-reduces its peers never join (deadlock), on a BatchNorm batch of one."""
    n = int(order.shape[0]) if hasattr(order, 'shape') else len(order)
    if world <= 1 or n == 0:
        return order
    per_rank = (n + world - 1) // world
    pad = per_rank * world - n
    if pad:
        order = torch.cat([order, order[:pad]]) if torch.is_tensor(order) else list(order) + list(order[:pad])
    return order[rank::world]


def shard_len(n_total, world):
    return (n_total + world - 1) // world if world > 1 else n_total


class IdxLoader(object):
    """MNIST / FashionMNIST from the raw IDX files, without torchvision: the whole split sits in HBM as
    uint8 (47 MB), a batch is an index gather + ``preprocess.to_tensor`` on the device -- the
    DataLoader + ``transforms.ToTensor()`` of mnist/train.py:159-165 with no host work per batch.
    Same iteration surface (len(), .dataset, (image float [B,1,28,28], label int64 [B]) batches, the
    short last batch kept, reshuffled every epoch when ``shuffle``)."""

    def __init__(self, images_path, labels_path, batch_size, shuffle, device, seed=0, rank=0, world=1,
                 n_classes=10):
        from . import preprocess
        self._to_tensor = preprocess.to_tensor
        images, labels = read_idx(images_path), read_idx(labels_path)
        if images.ndim != 3 or labels.ndim != 1 or images.shape[0] != labels.shape[0]:
            raise ValueError('unexpected IDX shapes %s / %s' % (images.shape, labels.shape))
        if labels.size and int(labels.max()) >= n_classes:
            # the loss kernel indexes the logits row with the label: refuse a corrupt file here, loudly
            raise ValueError('%s: label %d outside 0..%d' % (labels_path, int(labels.max()), n_classes - 1))
        self.images = torch.from_numpy(images.copy()).to(device)
        self.labels = torch.from_numpy(labels.astype('int64')).to(device)
        self.batch_size, self.shuffle, self.device = int(batch_size), bool(shuffle), device
        self.rank, self.world = int(rank), int(world)
        # data parallel: every rank draws the SAME permutation (shared seed) and keeps order[rank::world],
        # so one epoch is one pass over the data across all ranks (what a DistributedSampler does)
        self.dataset = range(shard_len(images.shape[0], self.world))     # len() = samples per rank, equal on all ranks
        self._n_total = images.shape[0]
        self._gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n_total = self._n_total
        order = torch.randperm(n_total, generator=self._gen) if self.shuffle else torch.arange(n_total)
        order = shard_order(order, self.rank, self.world).to(self.device)
        n = order.numel()
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size]
            yield self._to_tensor(self.images[idx]), self.labels[idx]


def _find_idx(root, stem):
    for name in (stem, stem + '.gz', stem.replace('-idx', '.idx'), stem.replace('-idx', '.idx') + '.gz'):
        for sub in ('', 'raw', 'MNIST/raw', 'FashionMNIST/raw'):
            p = os.path.join(root, sub, name)
            if os.path.exists(p):
                return p
    return None


def _real_loaders(kind, batch_size, device, rank=0, data_dir='./data', world=1):
    """The loaders of mnist/train.py:159-165 / fashionmnist/train.py:159-165 from the IDX files under
    ``data_dir`` (where torchvision's ``download=True`` puts them), and of celeba/train.py:146-156 /
    celeba19/train.py from the aligned-CelebA folder layout (``img_align_celeba/``, ``Anno/``, ``Eval/``) via
    ``celeba.datasets.CelebaLoader``.  This box has no network, so the files must already be there."""
    if kind in ('celeba', 'celeba19'):
        from .celeba.datasets import CelebaLoader
        if not os.path.isfile(os.path.join(data_dir, 'Eval/list_eval_partition.txt')):
            raise SystemExit('no CelebA files under %s (no network to download them): run with --synthetic' % data_dir)
        return (CelebaLoader('train', data_dir, batch_size, True, device, seed=1234, rank=rank, world=world),
                CelebaLoader('val', data_dir, batch_size, False, device))
    paths = [_find_idx(data_dir, stem) for stem in ('train-images-idx3-ubyte', 'train-labels-idx1-ubyte',
                                                    't10k-images-idx3-ubyte', 't10k-labels-idx1-ubyte')]
    if any(p is None for p in paths):
        raise SystemExit('no %s IDX files under %s (no network to download them): run with --synthetic' % (kind, data_dir))
    return (IdxLoader(paths[0], paths[1], batch_size, True, device, seed=1234, rank=rank, world=world),
            IdxLoader(paths[2], paths[3], batch_size, False, device))


def run(kind, mvae_cls, test_total, args, lambda_label, annealing_epoch_offset=0, make_engine=None):
    """The reference's main loop.  ``annealing_epoch_offset``: 0 for mnist/celeba
    ((epoch - 1) * N, mnist/train.py:182), 1 for fashionmnist (epoch * N, fashionmnist/train.py:182).
    ``make_engine(model, args, rank, batch_size)``: the fused step to use instead of ``BimodalStep`` (celeba19)."""
    import torch.distributed as dist
    from .engine import BimodalStep
    from .optim import FusedAdam
    from .parallel import DataParallel

    # N ranks on one node exchange gradients through RCCL: this platform's driver only supports dmabuf IPC, and the
    # HSA runtime reads the switch when the process first touches the GPU (the is_available() call below) -- without
    # it RCCL fails with hipIpcGetMemHandle: invalid argument
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    args.cuda = args.cuda and torch.cuda.is_available()
    if not args.cuda:
        raise SystemExit('this drop-in runs the MVAE step as HIP kernels: pass --cuda on a ROCm GPU box '
                         '(the CPU path is the reference itself)')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group('nccl', device_id=device)
    if rank == 0 and not os.path.isdir(args.out_dir):
        os.makedirs(args.out_dir)

    if args.synthetic:
        train_loader = SyntheticLoader(kind, args.batch_size, args.steps_per_epoch, 1234 + rank, device,
                                       last_batch=args.synthetic_last_batch)
        test_loader = SyntheticLoader(kind, args.batch_size, max(1, args.steps_per_epoch // 10), 4321, device)
    else:
        train_loader, test_loader = _real_loaders(kind, args.batch_size, device, rank, args.data_dir, world)
    N_mini_batches = len(train_loader)

    model = mvae_cls(args.n_latents)
    model.cuda(device)
    optimizer = FusedAdam(model.parameters(), lr=args.lr, grad_scale=1.0 / world)
    def build_engine(batch_size):
        if make_engine is not None:
            return make_engine(model, args, rank, batch_size)
        return BimodalStep(model, batch_size, args.lambda_image, lambda_label, seed=1 + rank)

    engine = build_engine(args.batch_size)
    dp = DataParallel(model, engine) if world > 1 else None
    captured = [False]
    ragged = {}           # the reference trains on the loader's short last batch too: eager engine per size

    def train(epoch):
        model.train()
        train_loss_meter = AverageMeter()
        pending = []          # device-side losses; read back only at the log interval
        for batch_idx, (image, label) in enumerate(train_loader):
            if epoch < args.annealing_epochs:
                annealing_factor = (float(batch_idx + (epoch - 1 + annealing_epoch_offset) * N_mini_batches + 1) /
                                    float(args.annealing_epochs * N_mini_batches))
            else:
                annealing_factor = 1.0
            image, label = image.to(device), label.to(device)
            if len(image) != args.batch_size:
                eng = ragged.get(len(image))
                if eng is None:
                    eng = ragged[len(image)] = build_engine(len(image))
                    eng.counter = engine.counter      # one Philox stream: never replay the main engine's draws
                    if dp is not None:
                        eng.configure_buckets(dp.n_buckets)
                        eng.on_bucket_ready = dp.buckets.launch
                elbo = eng.step(image, label, annealing_factor)
                if dp is not None:
                    dp.finish(optimizer)      # Adam per gradient bucket as its all-reduce lands
                else:
                    optimizer.step()
            elif not args.no_graph:
                if not captured[0]:
                    engine.capture(optimizer, image.shape[1:], label, comm=dp)
                    captured[0] = True
                elbo = engine.replay(image, label, annealing_factor)
            else:
                elbo = engine.step(image, label, annealing_factor)
                if dp is not None:
                    dp.finish(optimizer)      # Adam per gradient bucket as its all-reduce lands
                else:
                    optimizer.step()
            pending.append((elbo[-1].clone(), len(image)))
            if batch_idx % args.log_interval == 0:
                # ONE device->host sync per log line
                for v, n in zip(torch.stack([q[0] for q in pending]).tolist(), [q[1] for q in pending]):
                    train_loss_meter.update(v, n)
                pending = []
                if rank == 0:
                    print('Train Epoch: {} [{}/{} ({:.0f}%)]\tLoss: {:.6f}\tAnnealing-Factor: {:.3f}'.format(
                        epoch, batch_idx * len(image), len(train_loader.dataset),
                        100. * batch_idx / len(train_loader), train_loss_meter.avg, annealing_factor))
        if pending:
            for v, n in zip(torch.stack([q[0] for q in pending]).tolist(), [q[1] for q in pending]):
                train_loss_meter.update(v, n)
        if rank == 0:
            print('====> Epoch: {}\tLoss: {:.4f}'.format(epoch, train_loss_meter.avg))

    def test(epoch):
        model.eval()
        test_loss_meter = AverageMeter()
        with torch.no_grad():
            for batch_idx, (image, label) in enumerate(test_loader):
                image, label = image.to(device), label.to(device)
                # celeba19/train.py:338 divides the sum of batch means by len(test_loader); the others weight by B
                test_loss_meter.update(test_total(model, image, label, args).item(),
                                       1 if kind == 'celeba19' else len(image))
        if rank == 0:
            print('====> Test Loss: {:.4f}'.format(test_loss_meter.avg))
        return test_loss_meter.avg

    best_loss = sys.maxsize
    for epoch in range(1, args.epochs + 1):
        train(epoch)
        test_loss = test(epoch)
        is_best = test_loss < best_loss
        best_loss = min(test_loss, best_loss)
        if rank == 0:
            save_checkpoint({
                'state_dict': model.state_dict(),
                'best_loss': best_loss,
                'n_latents': args.n_latents,
                'optimizer': optimizer.state_dict(),
            }, is_best, folder=args.out_dir)
    if world > 1:
        dist.destroy_process_group()
