"""CPU: the ROCm runtime settings the package asks for before the first HIP call (multimodal-vae-public_amd/_runtime.py):
two hardware queues for a single-process job, nothing for a rank of a multi-process job, never over the user's own
setting, and a spawned rank drops what it inherited from a single-process parent."""
import importlib.util
import os

import pytest

PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-vae-public_amd', '_runtime.py')


def configure(monkeypatch, **env):
    for k in ('GPU_MAX_HW_QUEUES', '_MVAE_HWQ_AUTO', 'WORLD_SIZE', 'MVAE_RUNTIME_ENV'):
        monkeypatch.setenv(k, 'x')          # so that the undo list restores whatever the module sets
        monkeypatch.delenv(k)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    spec = importlib.util.spec_from_file_location('_mvae_runtime_probe', PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.APPLIED, os.environ.get('GPU_MAX_HW_QUEUES')


def test_single_process_job_gets_two_queues(monkeypatch):
    applied, value = configure(monkeypatch)
    assert applied == {'GPU_MAX_HW_QUEUES': '2'} and value == '2'
    assert os.environ['_MVAE_HWQ_AUTO'] == str(os.getpid())


@pytest.mark.parametrize('env,value', [({'WORLD_SIZE': '8'}, None),
                                       ({'GPU_MAX_HW_QUEUES': '4'}, '4'),
                                       ({'WORLD_SIZE': '8', 'GPU_MAX_HW_QUEUES': '3'}, '3'),
                                       ({'MVAE_RUNTIME_ENV': '0'}, None)])
def test_left_alone(monkeypatch, env, value):
    applied, got = configure(monkeypatch, **env)
    assert applied == {} and got == value


def test_rank_drops_what_a_single_process_parent_set(monkeypatch):
    applied, value = configure(monkeypatch, WORLD_SIZE='2', GPU_MAX_HW_QUEUES='2', _MVAE_HWQ_AUTO='1')
    assert applied == {} and value is None and '_MVAE_HWQ_AUTO' not in os.environ
