"""Run the reference's OWN quantizer-path source files (TEST INFRASTRUCTURE — build container only).

    from oracle import ref_import
    ref = ref_import.load()            # needs /root/reference; raises ReferenceUnavailable elsewhere
    q = ref.build_quantizer(dict(type='VQGANQuantizer', ...))

What this is.  The reference package cannot be imported as a whole here: ``import vq`` executes every
``__init__.py`` (datasets, encoders, runners, ...) and the first ``import todd`` fails (todd_ai @ed2a3ae is
un-vendored).  But the files ON the hot path are small, pure Python over ATen, and use todd only
for plumbing.  This module therefore

  1. puts a stand-in ``todd`` on ``sys.modules`` that supplies STRUCTURE ONLY (Config, Registry/RegistryMeta,
     BuildPreHookMixin, HolderMixin, PriorityQueue, ModuleDict, rank helpers, Store flags) — see ``_install_todd``;
  2. imports the reference's source files from where they lie (``importlib`` on /root/reference/vq/...; nothing is
     copied), with the package ``__init__.py`` files replaced by the sub-set of their star re-exports that the
     path needs (``_PACKAGES``: each entry cites the ``__init__.py`` line it mirrors);
  3. exposes the loaded classes, so that ``oracle/make_golden.py`` generates every fixture by calling the real
     ``L2Distance.forward``, ``CosineDistance.forward``, ``VectorQuantizer._encode/_decode/forward``, ``ste``,
     ``BaseQuantizer.forward``, ``VQGANLoss``, ``QuantStatistics``, ``NormalizeCallback``, ``VQKDCallback``,
     ``CVQVAECallback``, ``NearestAnchor`` ... through the reference's own registries and config dicts.

What stays defined-by-SURVEY (un-vendored todd arithmetic — three things, kept together in ``_ToddArithmetic``):
``todd.utils.ema(a,b,γ)=a·γ+b·(1−γ)``, ``todd.utils.EMA`` (default γ=0.99, first call with ``None`` state returns
``b``), and ``todd.models.losses.MSELoss(norm=)`` (mean-reduced, weight 1; ``norm=True`` L2-normalises both
arguments along dim 1).  Everything else that computes is the reference's or ATen's.

Nothing here travels to the GPU box as behaviour: /root/reference does not exist there; only the generated
``tests/golden/*.npz`` do.  The product package never imports this module.
"""
from __future__ import annotations

import functools
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types
from typing import Any, Generic, Iterable, Mapping, TypeVar

REFERENCE_ROOT = os.environ.get('VQ_REFERENCE_ROOT', '/root/reference')


class ReferenceUnavailable(RuntimeError):
    pass


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'vq', 'algorithms', 'vq', 'distances.py'))


# =====================================================================================================
# 1. stand-in todd: structure only
# =====================================================================================================

class _ToddArithmetic:
    """The ONLY arithmetic the stand-in supplies — todd_ai is un-vendored, these follow SURVEY.md §8c."""

    @staticmethod
    def ema(a, b, decay):
        return a * decay + b * (1 - decay)

    class EMA:
        def __init__(self, *args, decay: float = 0.99, **kwargs) -> None:
            self._decay = decay

        @property
        def decay(self):
            return self._decay

        def __call__(self, a, b):
            return b if a is None else _ToddArithmetic.ema(a, b, self._decay)

    @staticmethod
    def mse(pred, target, norm: bool):
        import torch.nn.functional as F
        if norm:
            pred, target = F.normalize(pred), F.normalize(target)
        return F.mse_loss(pred, target)


def _install_todd() -> types.ModuleType:
    if 'todd' in sys.modules and getattr(sys.modules['todd'], '__standin__', False):
        return sys.modules['todd']
    if 'todd' in sys.modules:
        raise ReferenceUnavailable('a real todd is importable: use it, not the stand-in')
    import torch
    import torch.distributed as dist
    from torch import nn

    # ---- Config: attribute dict ----
    class Config(dict):
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            for k in list(self):
                dict.__setitem__(self, k, Config._lift(dict.__getitem__(self, k)))

        @staticmethod
        def _lift(v):
            if type(v) is dict:
                return Config(v)
            if type(v) in (list, tuple):
                return type(v)(Config._lift(i) for i in v)
            return v

        def __setitem__(self, k, v):
            dict.__setitem__(self, k, Config._lift(v))

        def __getattr__(self, k):
            if k.startswith('__'):
                raise AttributeError(k)
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k) from None

        __setattr__ = __setitem__

        def get_config(self, key):
            v = self.get(key)
            return Config() if v is None else Config(v)

        def update(self, *a, **kw):
            for k, v in dict(*a, **kw).items():
                self[k] = v

    # ---- registries: a metaclass holding name -> item, children visible from parents ----
    class RegistryMeta(type):
        def __init__(cls, name, bases, ns):
            super().__init__(name, bases, ns)
            cls._table = {}

        def register_(cls, *names, force=False):
            def deco(item):
                for n in names or (item.__name__,):
                    assert force or n not in cls._table, n
                    cls._table[n] = item
                return item
            return deco

        def _find(cls, key, seen):
            if cls in seen:
                return None
            seen.add(cls)
            if key in cls.__dict__.get('_table', {}):
                return cls._table[key]
            for rel in list(cls.__subclasses__()) + [b for b in cls.__mro__[1:] if isinstance(b, RegistryMeta)]:
                hit = rel._find(key, seen)
                if hit is not None:
                    return hit
            return None

        def _resolve(cls, type_):
            if not isinstance(type_, str):
                return type_
            key = type_.rsplit('.', 1)[-1]
            item = cls._find(key, set())
            if item is None and key.startswith('torch_'):       # todd registers torch classes under mangled paths
                parts = key.split('_')
                for cut in range(len(parts) - 1, 0, -1):
                    try:
                        item = getattr(importlib.import_module('.'.join(parts[:cut])), '_'.join(parts[cut:]))
                        break
                    except (ImportError, AttributeError):
                        continue
            if item is None:
                raise KeyError(f'{type_!r} not found from {cls.__name__}')
            return item

        def build(cls, config, **kwargs):
            config = Config(config)
            config.update(kwargs)
            item = cls._resolve(config.pop('type'))
            if isinstance(item, type) and issubclass(item, BuildPreHookMixin):
                config = item.build_pre_hook(config, cls, item)
            return item(**config)

        def build_or_return(cls, obj, **kwargs):
            return cls.build(obj, **kwargs) if isinstance(obj, dict) else obj

    class Registry(metaclass=RegistryMeta):
        pass

    class BuildPreHookMixin:
        @classmethod
        def build_pre_hook(cls, config, registry, item):
            return config

    def _registry(name):
        return RegistryMeta(name, (Registry,), {})

    class InitRegistry(Registry):
        """``InitRegistry.build(dict(type='uniform_', a=..., b=...))`` -> callable applying torch.nn.init.<type>."""
        @classmethod
        def build(cls, config, **kwargs):
            config = dict(config, **kwargs)
            return functools.partial(getattr(nn.init, config.pop('type')), **config)

    # ---- holder / queue / module dict ----
    T = TypeVar('T')

    class HolderMixin(Generic[T]):
        def __init__(self, *args, instance=None, **kwargs):
            super().__init__(*args, **kwargs)
            if instance is not None:
                self._instance = instance

        def bind(self, instance):
            self._instance = instance

    class PriorityQueue:
        def __init__(self, priorities: Iterable[Mapping[str, int]], items: Iterable[Any]):
            self._rows = [(dict(p), it) for p, it in zip(priorities, items)]

        def __call__(self, key):
            order = sorted(range(len(self._rows)), key=lambda i: (self._rows[i][0].get(key, 0), i))
            return [self._rows[i][1] for i in order]

    class ModuleDict(nn.ModuleDict):
        def forward(self, *args, **kwargs):
            return {k: m(*args, **kwargs) for k, m in self.items()}

    class ModuleList(nn.ModuleList):
        pass

    class Sequential(nn.Sequential):
        pass

    class BaseLoss(BuildPreHookMixin, nn.Module):          # VQGANLoss.build_pre_hook calls super() through it (losses.py:107)
        def __init__(self, *args, **kwargs):
            super().__init__()

    class MSELoss(BaseLoss):
        def __init__(self, *args, norm: bool = False, **kwargs):
            super().__init__()
            self._norm = norm

        def forward(self, pred, target):
            return _ToddArithmetic.mse(pred, target, self._norm)

    def get_world_size():
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def get_rank():
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def all_gather(t):
        out = [torch.empty_like(t) for _ in range(get_world_size())]
        dist.all_gather(out, t)
        return out

    def is_sync(t):
        if get_world_size() <= 1:
            return True
        ts = all_gather(t.detach().contiguous())
        return all(torch.equal(ts[0], u) for u in ts[1:])

    class _StoreT(type):
        DRY_RUN = property(lambda cls: bool(os.environ.get('DRY_RUN')))
        cuda = property(lambda cls: torch.cuda.is_available())

    class Store(metaclass=_StoreT):
        pass

    class StoreMeta(type):
        pass

    class _Logger:
        def info(self, *a, **k):
            pass

        debug = warning = info

    # ---- runner-side holders (structure only): what a callback / metric is bound to, and todd's string accessor ----
    class _RunnerHolder(Generic[T]):
        def __init__(self, *args, **kwargs) -> None:
            super().__init__()

        def bind(self, runner) -> None:
            self._runner = runner

        @property
        def runner(self):
            return self._runner

    class BaseCallback(_RunnerHolder[T]):
        def before_run_iter(self, batch, memo) -> None:
            pass

        def after_run_iter(self, batch, memo) -> None:
            pass

    class BaseMetric(_RunnerHolder[T]):
        pass

    def get_(obj, attr: str):
        """todd.patches.py_.get_: the configs pass accessor strings such as '["quantizer"]["quant"]'
        (configs/vqgan/runner.py:123)."""
        return eval('__o' + attr, {'__o': obj})     # noqa: S307 — build-container test infrastructure, fixed strings

    known = {
        'todd': dict(Config=Config, Registry=Registry, RegistryMeta=RegistryMeta, Store=Store, logger=_Logger()),
        'todd.bases': {},
        'todd.bases.registries': dict(BuildPreHookMixin=BuildPreHookMixin, Item=Any, RegistryMeta=RegistryMeta),
        'todd.bases.registries.base': dict(Item=Any),
        'todd.registries': dict(InitRegistry=InitRegistry, ModelRegistry=_registry('ModelRegistry'),
                                DatasetRegistry=_registry('DatasetRegistry'),
                                RunnerRegistry=_registry('RunnerRegistry'), TaskRegistry=_registry('TaskRegistry')),
        'todd.models': dict(LossRegistry=_registry('LossRegistry')),
        'todd.models.losses': dict(BaseLoss=BaseLoss, MSELoss=MSELoss),
        'todd.patches': {},
        'todd.patches.torch': dict(ModuleDict=ModuleDict, ModuleList=ModuleList, Sequential=Sequential,
                                   get_world_size=get_world_size, get_rank=get_rank, all_gather=all_gather),
        'todd.runners': dict(Memo=dict),
        'todd.runners.callbacks': dict(BaseCallback=BaseCallback),
        'todd.runners.metrics': dict(BaseMetric=BaseMetric),
        'todd.patches.py_': dict(get_=get_),
        'todd.runners.utils': dict(PriorityQueue=PriorityQueue),
        'todd.utils': dict(EMA=_ToddArithmetic.EMA, ema=_ToddArithmetic.ema, HolderMixin=HolderMixin, is_sync=is_sync,
                           StoreMeta=StoreMeta, EnvRegistry=_registry('EnvRegistry')),
        'todd.configs': {},
    }

    class _Permissive(types.ModuleType):
        """Names the path never executes (PyConfig, load_state_dict, ...) resolve to inert placeholders."""
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            full = f'{self.__name__}.{name}'
            if full in sys.modules:
                return sys.modules[full]
            ph = type(name, (), {'__doc__': f'inert placeholder for {full}'})
            setattr(self, name, ph)
            return ph

    class _ToddFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, fullname, path=None, target=None):
            if fullname == 'todd' or fullname.startswith('todd.'):
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            m = _Permissive(spec.name)
            m.__path__ = []
            m.__standin__ = True
            return m

        def exec_module(self, module):
            for k, v in known.get(module.__name__, {}).items():
                setattr(module, k, v)

    sys.meta_path.insert(0, _ToddFinder())
    todd = importlib.import_module('todd')
    for name in known:
        importlib.import_module(name)
    return todd


# =====================================================================================================
# 2. the reference's files, with the package __init__.py files narrowed to the path
# =====================================================================================================

# package -> ordered steps mirroring the real __init__.py; ('pkg', name) = "from . import name",
# ('star', name) = "from .name import *".  Lines of the real __init__.py that lead off the path are left out.
_PACKAGES = {
    'vq': [('star', 'registries')],                                              # vq/__init__.py:4 (utils: below)
    'vq.utils': [('star', 'builders'), ('star', 'misc')],                        # vq/utils/__init__.py:1,3
    'vq.models': [('star', 'registries')],                                       # vq/models/__init__.py:2
    'vq.tasks': [('star', 'registries')],                                        # vq/tasks/__init__.py
    'vq.tasks.image_tokenization': [('star', 'registries')],                     # .../image_tokenization/__init__.py:2
    'vq.tasks.image_tokenization.models': [('pkg', 'quantizers'), ('star', 'registries')],   # models/__init__.py:1,3
    'vq.tasks.image_tokenization.runners': [('star', 'callbacks'), ('star', 'metrics'),      # runners/__init__.py:1-4
                                            ('star', 'registries'), ('star', 'tokenizer')],
    'vq.tasks.image_tokenization.models.connectors': [                           # connectors/__init__.py:1-3
        ('star', 'base'), ('star', 'composed'), ('star', 'conv')],
    'vq.tasks.image_tokenization.models.quantizers': [                           # quantizers/__init__.py:1-4
        ('pkg', 'callbacks'), ('pkg', 'utils'), ('star', 'base'), ('star', 'losses'), ('star', 'registries')],
    'vq.tasks.image_tokenization.models.quantizers.callbacks': [                 # callbacks/__init__.py:1-3
        ('star', 'base'), ('star', 'composed'), ('star', 'lazy_init_weights')],
    'vq.tasks.image_tokenization.models.quantizers.utils': [                     # utils/__init__.py:1-2
        ('star', 'quantizer_holder'), ('star', 'ste')],
    'vq.algorithms': [],
    'vq.algorithms.vq': [('pkg', 'callbacks'), ('star', 'distances'), ('star', 'losses'),    # vq/__init__.py:1-5
                         ('star', 'quantizers'), ('star', 'utils')],
    'vq.algorithms.vq.callbacks': [('star', 'normalize'), ('star', 'update')],   # callbacks/__init__.py:1-2
    'vq.algorithms.vqgan': [('star', 'quantizer')],                              # vqgan/__init__.py:4
    'vq.algorithms.vqkd': [('pkg', 'quantizers')],                               # vqkd/__init__.py:1
    'vq.algorithms.vqkd.quantizers': [('star', 'base'), ('star', 'callbacks')],  # quantizers/__init__.py:1-2
    'vq.algorithms.cvqvae': [('star', 'anchors'), ('star', 'quantizer_callback'), ('star', 'registries')],
}


# Packages the model-level file (image_tokenization/models/base.py) imports for type annotations only (datasets,
# encoders, runners): inert placeholder modules, nothing of theirs is ever executed.
_INERT = ('vq.datasets', 'vq.models.autoencoders', 'vq.runners')


class _InertModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        ph = type(name, (), {'__doc__': f'inert placeholder for {self.__name__}.{name}',
                             '__class_getitem__': classmethod(lambda cls, item: cls)})      # Generic[...] in annotations
        setattr(self, name, ph)
        return ph


class _RefFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Packages listed in _PACKAGES are created bare (their __init__.py is NOT executed) and then re-export what
    _PACKAGES says; every other vq.* name is the reference's own .py file executed unchanged."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname != 'vq' and not fullname.startswith('vq.'):
            return None
        rel = os.path.join(REFERENCE_ROOT, *fullname.split('.'))
        if fullname in _INERT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        if fullname in _PACKAGES:
            spec = importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            spec.submodule_search_locations = [rel]
            return spec
        if os.path.isfile(rel + '.py'):
            return importlib.util.spec_from_file_location(fullname, rel + '.py')
        raise ReferenceUnavailable(f'{fullname} is outside the quantizer path loaded by oracle/ref_import.py')

    def create_module(self, spec):
        if spec.name in _INERT:
            m = _InertModule(spec.name)
            m.__path__ = []
            return m
        return None

    def exec_module(self, module):
        if module.__name__ in _INERT:
            return
        for kind, name in _PACKAGES[module.__name__]:
            sub = importlib.import_module(f'{module.__name__}.{name}')
            if kind == 'star':
                for n in getattr(sub, '__all__', ()):
                    setattr(module, n, getattr(sub, n))


@functools.lru_cache(1)
def load() -> types.SimpleNamespace:
    """Import the reference's quantizer-path files; returns their public names in one namespace."""
    if not available():
        raise ReferenceUnavailable(f'{REFERENCE_ROOT} is not present (build container only)')
    todd = _install_todd()
    import torch.serialization as ts
    if not hasattr(ts, 'FILE_LIKE'):      # vq/utils/misc.py:67 annotates with the pre-2.6 name of torch's FileLike alias
        ts.FILE_LIKE = getattr(ts, 'FileLike', Any)
    if not any(isinstance(f, _RefFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _RefFinder())
    mods = {}
    for name in ('vq.algorithms.vq', 'vq.algorithms.vqgan', 'vq.algorithms.vqkd.quantizers', 'vq.algorithms.cvqvae',
                 'vq.tasks.image_tokenization.models.quantizers',
                 'vq.tasks.image_tokenization.models.quantizers.utils',
                 'vq.tasks.image_tokenization.models.quantizers.callbacks',
                 'vq.tasks.image_tokenization.models', 'vq.algorithms.vq.callbacks',
                 'vq.tasks.image_tokenization.models.connectors', 'vq.tasks.image_tokenization.models.base'):
        mods[name] = importlib.import_module(name)
    ns = types.SimpleNamespace(todd=todd, Config=todd.Config, modules=mods, files={})
    for m in mods.values():
        for n, v in vars(m).items():
            if not n.startswith('_') and not isinstance(v, types.ModuleType):
                setattr(ns, n, v)
    for n, m in sorted(sys.modules.items()):
        f = getattr(m, '__file__', None)
        if n.startswith('vq.') and f and f.startswith(REFERENCE_ROOT):
            ns.files[n] = os.path.relpath(f, REFERENCE_ROOT)

    def build_quantizer(config: dict, init_weights: dict | None = None):
        """``VQITQuantizerRegistry.build`` of a reference quantizer config (configs/vqgan/model.py:19-23 etc.), then
        ``init_weights`` as the model does (image_tokenization/models/base.py: quantizer.init_weights(config))."""
        cfg = todd.Config(config)
        iw = cfg.pop('init_weights', None) if init_weights is None else init_weights
        q = ns.VQITQuantizerRegistry.build(cfg)
        q.init_weights(todd.Config(iw or {}))
        return q

    ns.build_quantizer = build_quantizer
    return ns


@functools.lru_cache(1)
def load_runners() -> types.SimpleNamespace:
    """The caller side of the path, also from the reference's own files: ``Tokenizer``,
    ``TokenizeCallback``, ``CodebookUsageMetric``, ``CodebookPPLMetric`` (vq/tasks/image_tokenization/runners/*.py) and the
    LlamaGen ``TokenizeCallback`` of tools/tokenize_llamagen.py.  ``vq.runners`` (BaseValidator, the runner registries'
    other parent) stays inert: the classes are driven by hand in oracle/make_golden.py, exactly the methods that touch
    tokens and files."""
    ref = load()
    models = sys.modules['vq.tasks.image_tokenization.models']
    models.BaseModel = ref.BaseModel                                              # models/__init__.py:2 (from .base import *)
    runners = importlib.import_module('vq.tasks.image_tokenization.runners')
    it = sys.modules['vq.tasks.image_tokenization']
    it.runners = runners                                                          # image_tokenization/__init__.py:1
    for stub in ('torchvision', 'torchvision.transforms'):                        # imported by the tool, used by its dataset hook only
        if stub not in sys.modules:
            m = _InertModule(stub)
            m.__path__ = []
            sys.modules[stub] = m
    path = os.path.join(REFERENCE_ROOT, 'tools', 'tokenize_llamagen.py')
    spec = importlib.util.spec_from_file_location('ref_tools_tokenize_llamagen', path)
    tool = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = tool
    spec.loader.exec_module(tool)
    return types.SimpleNamespace(Tokenizer=runners.Tokenizer, TokenizeCallback=runners.TokenizeCallback, Tokens=runners.Tokens,
                                 CodebookUsageMetric=runners.CodebookUsageMetric, CodebookPPLMetric=runners.CodebookPPLMetric,
                                 VQITCallbackRegistry=runners.VQITCallbackRegistry, VQITMetricRegistry=runners.VQITMetricRegistry,
                                 LlamaGenTokenizeCallback=tool.TokenizeCallback, LlamaGenTokenizer=tool.Tokenizer,
                                 files=['vq/tasks/image_tokenization/runners/callbacks.py',
                                        'vq/tasks/image_tokenization/runners/metrics.py',
                                        'vq/tasks/image_tokenization/runners/tokenizer.py', 'tools/tokenize_llamagen.py'])


if __name__ == '__main__':
    r = load()
    print('reference files executed:')
    for n, f in r.files.items():
        print(f'  {n:75s} {f}')
