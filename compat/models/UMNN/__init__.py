from umnn_amd import (UMNNMAFFlow, MonotonicNN, IntegrandNN, IntegrandNetwork, UMNNMAF, MADE,  # noqa: F401
                      NeuralIntegral, ParallelNeuralIntegral)
