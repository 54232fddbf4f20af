"""Drop-in for the reference's ``celeba19/train.py``: same CLI (--n-latents --batch-size --epochs
--annealing-epochs --lr --log-interval --approx-m --lambda-image --lambda-attrs --cuda), same
function names, log lines and checkpoint format.  The per-batch body -- complete + image-only +
18 single-attribute + ``approx_m`` sampled-subset ELBO terms, celeba19/train.py:257-308 -- is
``engine.Celeba19Step``: one batched HIP pass instead of 20 + M ``model()`` calls."""
import os
import sys

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import mvae_amd  # noqa: F401
    __package__ = 'multimodal-vae-public_amd.celeba19'

import numpy as np  # noqa: E402

from ..engine import Celeba19Step, sample_subsets  # noqa: E402
from ..functional import binary_cross_entropy_with_logits  # noqa: E402,F401
from ..functional import elbo_loss_multi as elbo_loss  # noqa: E402
from ..train_common import AverageMeter, make_load_checkpoint, reference_parser, run, save_checkpoint  # noqa: E402,F401
from .model import MVAE, N_ATTRS  # noqa: E402,F401

load_checkpoint = make_load_checkpoint(MVAE)


def tensor_2d_to_list(x):
    """[B, 18] -> list of 18 [B] columns (celeba19/train.py:63-69)."""
    return [x[:, i] for i in range(x.size(1))]


def enumerate_combinations(n):
    """All subsets of n modalities with 2 .. n-1 members as an [n_subsets, n] boolean matrix ordered by size
    (celeba19/train.py:87-108).  The fused step never materialises this pool -- ``engine.sample_subsets`` makes the
    same generator calls and un-ranks the drawn row numbers -- the function is kept for callers that
    index into it."""
    from itertools import combinations
    rows = []
    for size in range(2, n):
        picks = list(combinations(range(n), size))
        block = np.zeros((len(picks), n), dtype=bool)
        for r, members in enumerate(picks):
            block[r, list(members)] = True
        rows.append(block)
    return np.concatenate(rows)


def sample_combinations(pool, size=1):
    """``size`` rows of ``pool`` exactly as celeba19/train.py:111-142 picks them from the global
    ``np.random`` state: a subset size per sample (uniform over the sizes present in the pool, with
    replacement), then for each drawn size, in increasing order, that many distinct rows of the size's
    block.  The fused step draws the same rows without a pool (``engine.sample_subsets``: identical
    generator calls, row number -> members by un-ranking)."""
    pool = np.asarray(pool)
    n = pool.shape[1]
    row_size = pool.sum(axis=1)
    present = np.flatnonzero(np.bincount(row_size))
    drawn = np.bincount(np.random.choice(present, size, replace=True), minlength=n)
    picked = []
    for k in range(n):
        if drawn[k] > 0:
            block = np.flatnonzero(row_size == k)
            picked.append(block[np.random.choice(block.size, size=int(drawn[k]), replace=False)])
    return pool[np.concatenate(picked)]


def _test_total(model, image, attrs, args):
    """celeba19/train.py:325-335: only the complete-data ELBO, default lambdas, beta = 1."""
    cols = tensor_2d_to_list(attrs.float())
    recon_image, recon_attrs, mu, logvar = model(image, cols)
    return elbo_loss([recon_image] + recon_attrs, [image] + cols, mu, logvar)


def _make_engine(model, args, rank, batch_size):
    # subsets come from the global numpy generator, as in the reference (celeba19/train.py:286): the same
    # np.random.seed gives the same subsets.  Data parallel: every rank seeds it identically once, so all
    # ranks draw the same subsets each step and do equal work.
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not getattr(_make_engine, 'seeded', False):
        np.random.seed(681307)
        _make_engine.seeded = True
    return Celeba19Step(model, batch_size, args.lambda_image, args.lambda_attrs,
                        approx_m=args.approx_m, seed=1 + rank, rng=np.random,
                        faithful_bn_stats=getattr(args, 'bn_stats', 'reference') == 'reference')


if __name__ == "__main__":
    args = reference_parser('celeba19').parse_args()
    run('celeba19', MVAE, _test_total, args, args.lambda_attrs, make_engine=_make_engine)
