from .dgraph import CSRAdjacency, DGraph  # noqa: F401
