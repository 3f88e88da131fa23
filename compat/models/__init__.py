"""Drop-in import shim: put this directory's parent (``compat/``) on PYTHONPATH *before* the reference checkout and the
reference's scripts' ``from models import UMNNMAFFlow`` / ``from models.UMNN import MonotonicNN, IntegrandNN`` resolve to
umnn_amd.  Sub-packages this shim does not provide (``models.vae_lib`` of TrainVaeFlow.py, whose ``flows.MMAF`` itself
does ``from models import UMNNMAFFlow``) are looked up in the next ``models/`` directory on ``sys.path``."""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
for _p in list(_sys.path):
    _cand = _os.path.abspath(_os.path.join(_p or ".", "models"))
    if _cand != _here and _os.path.isdir(_os.path.join(_cand, "vae_lib")) and _cand not in __path__:
        __path__.append(_cand)
        break

from models.UMNN import UMNNMAFFlow, MADE, ParallelNeuralIntegral, NeuralIntegral  # noqa: E402,F401
