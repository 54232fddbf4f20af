"""multimodal-vae-public_amd -- the MVAE train step of mhw32/multimodal-vae-public as hand-written
HIP kernels for MI355X (gfx950), behind the reference's own Python surface.

Import as ``import mvae_amd`` (alias module at the repo root; the directory name carries a
hyphen) or ``importlib.import_module('multimodal-vae-public_amd')``.

    mvae_amd.mnist.model.MVAE / .fashionmnist / .celeba / .celeba19   drop-in nn.Modules
    mvae_amd.functional.elbo_loss_*                                   the reference's loss functions
    mvae_amd.engine.BimodalStep / Celeba19Step                        fused, graph-captured train step
    mvae_amd.optim.FusedAdam                                          one-launch Adam over the arena
    mvae_amd.parallel.DataParallel                                    RCCL gradient all-reduce
    mvae_amd.capture_step(body, example_args, model=, optimizer=)     the reference's unchanged loop body as ONE hipGraph
"""
from . import _lib, kernels, arena, layers, functional, base, engine, optim, parallel, graph  # noqa: F401
from .graph import capture_step  # noqa: F401
from . import mnist, fashionmnist, celeba, celeba19  # noqa: F401

__all__ = ['kernels', 'arena', 'layers', 'functional', 'base', 'engine', 'optim', 'parallel', 'graph', 'capture_step',
           'mnist', 'fashionmnist', 'celeba', 'celeba19']
