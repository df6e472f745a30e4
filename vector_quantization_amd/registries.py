"""Registry hierarchy of the quantizer path, same names as the reference
(vq/registries.py:18-35, vq/tasks/image_tokenization/registries.py:11-16, .../runners/registries.py:11-16,
vq/tasks/image_tokenization/models/registries.py,
.../quantizers/registries.py:9-14, vq/algorithms/vq/distances.py:19-20, vq/algorithms/cvqvae/registries.py:8)."""
from .config import Registry


class ModelRegistry(Registry):
    """Stands in for todd.registries.ModelRegistry (resolves 'torch_nn_modules_sparse_Embedding')."""


class InitRegistry(Registry):
    """Stands in for todd.registries.InitRegistry: weight initialisers by name.  Any in-place initialiser of ``torch.nn.init``
    resolves by its own name — ``dict(type='trunc_normal_', std=0.02)``, ``dict(type='xavier_uniform_')`` ... — as a factory
    ``(**kwargs) -> (tensor -> tensor)``, what the generic branch of the reference hands to ``InitRegistry.build(config)``
    (vq/algorithms/vq/quantizers.py:87-90); explicitly registered names take precedence."""

    @classmethod
    def resolve(cls, type_):
        if isinstance(type_, str):
            key = type_.rsplit('.', 1)[-1]
            if cls._lookup(key) is None:
                from torch.nn import init
                fn = getattr(init, key, None)
                if callable(fn) and key.endswith('_') and not key.startswith('_'):
                    return lambda **kwargs: (lambda w: fn(w, **kwargs))
        return super().resolve(type_)


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
