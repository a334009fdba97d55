from .dgraph import DGraph  # noqa: F401
