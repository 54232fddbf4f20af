"""GPU: the four ``train.py`` drop-ins run end to end with the reference's CLI flags on synthetic
batches -- one short epoch of fused HIP steps (hipGraph replay), the eval-mode test pass through the
module surface, and a checkpoint that ``load_checkpoint`` reads back (mnist/train.py:124-129,
263-268)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    ('mnist', ['--n-latents', '16', '--lambda-text', '50', '--synthetic-last-batch', '5']),    # + a short dataset tail
    ('fashionmnist', ['--n-latents', '16', '--lambda-text', '50']),
    ('celeba', ['--n-latents', '20', '--lambda-attrs', '10', '--synthetic-last-batch', '3']),
    ('celeba19', ['--n-latents', '20', '--lambda-attrs', '10', '--approx-m', '2']),
]


@pytest.mark.parametrize('kind,extra', CASES)
def test_train_cli(kind, extra, tmp_path):
    script = os.path.join(ROOT, 'multimodal-vae-public_amd', kind, 'train.py')
    cmd = [sys.executable, script, '--cuda', '--synthetic', '--epochs', '1', '--steps-per-epoch', '6',
           '--batch-size', '8', '--annealing-epochs', '2', '--log-interval', '2', '--out-dir', str(tmp_path)] + extra
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'Train Epoch: 1' in out.stdout and '====> Test Loss:' in out.stdout
    losses = [float(line.split('Loss: ')[1].split()[0]) for line in out.stdout.splitlines() if 'Train Epoch' in line]
    assert all(l == l and abs(l) < 1e7 for l in losses)
    ckpt = os.path.join(str(tmp_path), 'checkpoint.pth.tar')
    assert os.path.exists(ckpt) and os.path.exists(os.path.join(str(tmp_path), 'model_best.pth.tar'))
    state = torch.load(ckpt, map_location='cpu', weights_only=False)
    assert set(state) == {'state_dict', 'best_loss', 'n_latents', 'optimizer'}


def test_mnist_cli_on_idx_files(tmp_path):
    """Without --synthetic: the loaders read raw IDX files (no torchvision), keep the split in HBM as
    uint8 and apply ToTensor on the device; the 37-image split leaves a short last batch."""
    import numpy as np
    from test_loader_cpu import write_idx
    rng = np.random.RandomState(1)
    data = tmp_path / 'data'
    data.mkdir()
    for stem, n in (('train', 37), ('t10k', 11)):
        write_idx(str(data / ('%s-images-idx3-ubyte' % stem)), rng.randint(0, 256, (n, 28, 28)))
        write_idx(str(data / ('%s-labels-idx1-ubyte' % stem)), rng.randint(0, 10, (n,)))
    script = os.path.join(ROOT, 'multimodal-vae-public_amd', 'mnist', 'train.py')
    cmd = [sys.executable, script, '--cuda', '--epochs', '2', '--batch-size', '8', '--n-latents', '16',
           '--annealing-epochs', '2', '--log-interval', '2', '--data-dir', str(data), '--out-dir', str(tmp_path / 'out')]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'Train Epoch: 2 [16/37' in out.stdout and '====> Test Loss:' in out.stdout
