"""umnn_amd -- MI355X-native (gfx950) neural-integration hot path of UMNN behind the reference's module API.

Import surface mirrors models/UMNN/__init__.py:1-6 of the reference.
"""
from .flow import UMNNMAFFlow, UMNNMAF, EmbeddingNetwork, IntegrandNetwork, ListModule
from .monotonic import MonotonicNN, IntegrandNN
from .made import MADE, ConditionnalMADE, MaskedLinear
from .integral import NeuralIntegral, ParallelNeuralIntegral, IntegralWithJacobian, integrate, path_taken
from .quadrature import compute_cc_weights
from .graphs import GraphedLL, GraphedTrainStep
from ._lib import set_forward_precision, get_forward_precision, set_backward_precision, get_backward_precision

__all__ = ["UMNNMAFFlow", "UMNNMAF", "EmbeddingNetwork", "IntegrandNetwork", "ListModule", "MonotonicNN",
           "IntegrandNN", "MADE", "ConditionnalMADE", "MaskedLinear", "NeuralIntegral", "ParallelNeuralIntegral",
           "IntegralWithJacobian", "integrate", "compute_cc_weights", "path_taken", "GraphedLL", "GraphedTrainStep",
           "set_forward_precision", "get_forward_precision", "set_backward_precision", "get_backward_precision"]
