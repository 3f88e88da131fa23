"""Drop-in import shim: put this directory's parent (``compat/``) on PYTHONPATH and the reference's scripts'
``from models import UMNNMAFFlow`` / ``from models.UMNN import MonotonicNN, IntegrandNN`` resolve to umnn_amd."""
from models.UMNN import UMNNMAFFlow, MADE, ParallelNeuralIntegral, NeuralIntegral  # noqa: F401
