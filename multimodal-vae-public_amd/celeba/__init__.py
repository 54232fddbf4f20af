"""HIP drop-in for the reference's ``celeba/`` experiment (model.py, train.py)."""
from . import model  # noqa: F401
