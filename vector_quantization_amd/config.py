"""Minimal counterparts of the todd symbols the quantizer path touches: ``Config``,
``Registry``/``RegistryMeta``, ``BuildPreHookMixin``.  todd_ai is un-vendored and absent here; when the real
todd is importable, ``vector_quantization_amd.integration.register_into_reference()`` registers the classes of
this package into the reference's own registries instead."""
from __future__ import annotations

import importlib
from typing import Any, Callable


class Config(dict):
    """Attribute-access dict (todd.Config).  Nested dicts become Configs on access."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def get_config(self, key: str) -> 'Config':
        """``config.get(key, Config())`` guaranteed to be a Config (todd.Config.get_config)."""
        v = self.get(key)
        return Config() if v is None else Config(v)

    def copy(self) -> 'Config':
        return Config(super().copy())


Item = Callable[..., Any]


class BuildPreHookMixin:
    """Classes may rewrite their build config (sub-dicts → built objects) before instantiation."""

    @classmethod
    def build_pre_hook(cls, config: Config, registry: 'RegistryMeta', item: Item) -> Config:
        return config


class RegistryMeta(type):
    """Class-level registry: every subclass of ``Registry`` owns a name → item table; lookups fall through
    to parent registries (child registries see their parents' entries, as todd's do)."""

    def __init__(cls, name, bases, ns):
        super().__init__(name, bases, ns)
        cls._items = {}

    def register_(cls, *names: str, force: bool = False):
        def deco(item):
            keys = names or (item.__name__,)
            for k in keys:
                if k in cls._items and not force:
                    raise KeyError(f'{k} is already registered in {cls.__name__}')
                cls._items[k] = item
            return item
        return deco

    def _lookup(cls, key: str):
        for klass in cls.__mro__:
            items = klass.__dict__.get('_items')
            if items and key in items:
                return items[key]
        # registries also see what child registries registered (todd resolves 'Child.Name' paths; plain names
        # are searched depth-first here)
        for sub in cls.__subclasses__():
            found = sub._lookup_down(key)
            if found is not None:
                return found
        return None

    def _lookup_down(cls, key: str):
        items = cls.__dict__.get('_items')
        if items and key in items:
            return items[key]
        for sub in cls.__subclasses__():
            found = sub._lookup_down(key)
            if found is not None:
                return found
        return None

    def resolve(cls, type_) -> Item:
        if not isinstance(type_, str):
            return type_
        key = type_.rsplit('.', 1)[-1]           # 'VQITQuantizerRegistry.VectorQuantizer' → class name
        item = cls._lookup(key)
        if item is None and key.startswith('torch_'):
            # todd auto-registers torch classes under mangled names, e.g. torch_nn_modules_sparse_Embedding
            parts = key.split('_')
            for split in range(len(parts) - 1, 0, -1):
                try:
                    mod = importlib.import_module('.'.join(parts[:split]))
                    item = getattr(mod, '_'.join(parts[split:]))
                    break
                except (ImportError, AttributeError):
                    continue
        if item is None:
            raise KeyError(f'{type_} is not registered in {cls.__name__}')
        return item

    def build(cls, config, **kwargs):
        config = Config(config)
        config.update(kwargs)
        item = cls.resolve(config.pop('type'))
        if isinstance(item, type) and issubclass(item, BuildPreHookMixin):
            config = item.build_pre_hook(config, cls, item)
        return item(**config)

    def build_or_return(cls, obj, **kwargs):
        if isinstance(obj, dict):
            return cls.build(obj, **kwargs)
        return obj


class Registry(metaclass=RegistryMeta):
    pass
