"""rechorus_amd: the ReChorus ranking hot path on MI355X."""
from . import graph as _graph

_graph.enable()  # selects the safe hipGraph launch path if HIP is not initialised yet (rechorus_amd/graph.py)
