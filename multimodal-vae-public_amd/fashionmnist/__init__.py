"""HIP drop-in for the reference's ``fashionmnist/`` experiment (model.py, train.py)."""
from . import model  # noqa: F401
