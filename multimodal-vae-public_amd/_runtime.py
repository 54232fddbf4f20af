"""ROCm runtime settings this package asks for BEFORE the HIP runtime starts (no torch import here).

``GPU_MAX_HW_QUEUES=2`` for a single-process job: the fused step is a fork/join DAG on two streams, replayed as a
hipGraph; with the runtime's default of four hardware queues the graph's branches are spread over more queues than
there are branches and every fork / join edge is a cross-queue signal (~9 us each on MI355X, six on the MNIST step's
critical path).  Two queues -- one per branch -- measured (profiles/r03_runtime_env.txt): MNIST B=512 0.314 -> 0.300
ms/step, CelebA B=256 2.592 -> 2.562, FashionMNIST / CelebA-19 unchanged; one queue does not run, eight are 2.5x slower.
Not applied when WORLD_SIZE > 1: there the communicator's stream is a third concurrent branch and must not share a
hardware queue with compute.  An explicit GPU_MAX_HW_QUEUES in the environment always wins; MVAE_RUNTIME_ENV=0
leaves the runtime alone.  The setting only takes effect if no HIP call was made before this module is imported
(import the package before touching torch.cuda)."""
import os
import sys


def configure():
    applied = {}
    if os.environ.get('MVAE_RUNTIME_ENV', '1') == '0':
        return applied
    try:
        world = int(os.environ.get('WORLD_SIZE', '1') or 1)
    except ValueError:
        world = 1
    torch = sys.modules.get('torch')
    started = bool(torch is not None and torch.cuda.is_initialized())
    auto = os.environ.get('_MVAE_HWQ_AUTO')
    if auto and auto != str(os.getpid()) and world > 1:
        # inherited from a single-process parent that spawned this rank: its choice, not the user's
        os.environ.pop('GPU_MAX_HW_QUEUES', None)
        os.environ.pop('_MVAE_HWQ_AUTO', None)
    if world == 1 and not started and 'GPU_MAX_HW_QUEUES' not in os.environ:
        os.environ['GPU_MAX_HW_QUEUES'] = '2'
        os.environ['_MVAE_HWQ_AUTO'] = str(os.getpid())
        applied['GPU_MAX_HW_QUEUES'] = '2'
    return applied


APPLIED = configure()
