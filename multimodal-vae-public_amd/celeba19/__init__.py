"""HIP drop-in for the reference's ``celeba19/`` experiment (model.py, train.py)."""
from . import model  # noqa: F401
