"""HIP drop-in for the reference's ``mnist/`` experiment (model.py, train.py)."""
from . import model  # noqa: F401
