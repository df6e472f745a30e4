"""Registry hierarchy of the quantizer path, same names as the reference
(vq/registries.py:18-35, vq/tasks/image_tokenization/registries.py:11-16, .../runners/registries.py:11-16,
vq/tasks/image_tokenization/models/registries.py,
.../quantizers/registries.py:9-14, vq/algorithms/vq/distances.py:19-20, vq/algorithms/cvqvae/registries.py:8)."""
from .config import Registry


class ModelRegistry(Registry):
    """Stands in for todd.registries.ModelRegistry (resolves 'torch_nn_modules_sparse_Embedding')."""


class InitRegistry(Registry):
    """Stands in for todd.registries.InitRegistry: weight initialisers by name ('uniform_', 'normal_', ...)."""


class VQRegistry(Registry):
    pass


class VQModelRegistry(VQRegistry, ModelRegistry):
    pass


class VQITQuantizerRegistry(VQModelRegistry):
    pass


class VQITConnectorRegistry(VQModelRegistry):
    pass


class VQITQuantizerDistanceRegistry(VQITQuantizerRegistry):
    pass


class VQITQuantizerCallbackRegistry(VQITQuantizerRegistry):
    pass


class VQITQuantizerLossRegistry(VQITQuantizerRegistry):
    pass


class AnchorRegistry(Registry):
    pass


class RunnerRegistry(Registry):
    """Stands in for todd.registries.RunnerRegistry."""


class VQRunnerRegistry(VQRegistry, RunnerRegistry):
    pass


class VQITRunnerRegistry(VQRunnerRegistry):
    pass


class VQITCallbackRegistry(VQITRunnerRegistry):
    pass


class VQITMetricRegistry(VQITRunnerRegistry):
    pass
