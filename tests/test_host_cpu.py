"""Host-side logic that needs no GPU: config / registry shim, building the reference's quantizer configs, hook
ordering, state-dict layout, and the guarantee that the product refuses to run without the HIP path."""
import os

import pytest
import torch

from vector_quantization_amd import (Config, ModelRegistry, VQITQuantizerCallbackRegistry, VQITQuantizerLossRegistry,
                                     VQITQuantizerRegistry, _lib, build_quantizer)
from vector_quantization_amd import quantizers as Q

EMB = 'torch_nn_modules_sparse_Embedding'


def test_config_and_registry():
    c = Config(a=dict(b=[dict(c=1)]), d=2)
    assert c.a.b[0].c == 1 and c.get_config('missing') == {} and c.get_config('a').b[0].c == 1
    c.e = dict(f=3)
    assert isinstance(c.e, Config) and c.e.f == 3
    emb = ModelRegistry.build(dict(type=EMB, num_embeddings=7, embedding_dim=3))
    assert isinstance(emb, torch.nn.Embedding) and emb.weight.shape == (7, 3)
    assert VQITQuantizerRegistry.resolve('VectorQuantizer') is Q.VectorQuantizer
    assert VQITQuantizerRegistry.resolve('VQITQuantizerRegistry.VQGANQuantizer') is Q.VQGANQuantizer
    assert VQITQuantizerLossRegistry.resolve('VQGANLoss') is Q.VQGANLoss
    assert VQITQuantizerCallbackRegistry.resolve('CVQVAECallback') is Q.CVQVAECallback
    with pytest.raises(KeyError):
        VQITQuantizerRegistry.resolve('NoSuchQuantizer')
    with pytest.raises(KeyError):                              # same name twice without force
        VQITQuantizerRegistry.register_()(Q.VectorQuantizer)


def _vqgan(K=64, D=16, **extra):
    return build_quantizer(dict(type='VQGANQuantizer', embedding=dict(type=EMB, num_embeddings=K, embedding_dim=D),
                                distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss')),
                                **extra))


def test_build_reference_configs_and_state_dict():
    q = _vqgan()
    assert isinstance(q, Q.VQGANQuantizer) and q.codebook_size == 64 and q.embedding_dim == 16
    assert isinstance(q.distance, Q.L2Distance) and isinstance(q.embedding, torch.nn.Embedding)
    assert q._fusable()
    q.init_weights(Config(type='vqgan'))
    assert float(q.embedding.weight.detach().abs().max()) <= 1 / 64
    assert q.embeddings.data_ptr() != q.embedding.weight.data_ptr()         # .embeddings is a clone
    kd = build_quantizer(dict(type='VQKDQuantizer', embedding=dict(type=EMB, num_embeddings=32, embedding_dim=8),
                              distance=dict(type='CosineDistance'), callbacks=[dict(type='VQKDCallback', ema=dict())],
                              losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True)))))
    assert not kd._fusable()                                                # norm=True MSE → general path
    assert kd._callbacks.callbacks[0].with_ema and kd._callbacks.callbacks[0]._ema.decay == 0.99
    assert set(kd.state_dict()) == {'_embedding.weight', '_losses.commitment_loss._weight._steps',
                                    '_losses.commitment_loss._mse._weight._steps'}
    kd.train()
    kd.init_weights(Config())
    assert len(kd._forward_pre_hooks) == 1                                  # k-means lazy init armed
    cvq = _vqgan(callbacks=[dict(type='CVQVAECallback', ema=dict(decay=0.9), anchor=dict(type='NearestAnchor'))])
    cvq.train()
    cvq.init_weights(Config(type='vqgan'))
    assert '_probability' in cvq.state_dict() and cvq.get_buffer('_probability').shape == (64,)
    assert isinstance(cvq._callbacks.callbacks[0]._anchor, Q.NearestAnchor)


def test_callback_priority_order():
    calls = []

    class Rec(Q.BaseCallback):
        def __init__(self, name):
            super().__init__()
            self.name = name

        def before_encode(self, x, memo):
            calls.append(self.name)
            return x

    cc = Q.ComposedCallback(priorities=[dict(before_encode=5), dict(), dict(before_encode=-3)],
                            callbacks=[Rec('a'), Rec('b'), Rec('c')])
    cc.before_encode(torch.zeros(1), {})
    assert calls == ['c', 'b', 'a']
    assert not cc.overrides_decode_or_loss()

    class Dec(Q.BaseCallback):
        def after_decode(self, z, memo):
            return z

    assert Q.ComposedCallback(priorities=[dict()], callbacks=[Dec()]).overrides_decode_or_loss()


def test_no_cpu_fallback():
    q = _vqgan()
    with pytest.raises(_lib.VqhipError):
        q(torch.zeros(4, 16), {})
    with pytest.raises(_lib.VqhipError):
        q.decode(torch.zeros(4, dtype=torch.long), {})


def test_integration_table_names_exist():
    """Every class INTEGRATION.md promises to register exists under the reference's name."""
    from vector_quantization_amd import connectors, integration
    for reg, names in integration.REPLACED.items():
        for n in names:
            assert hasattr(connectors if reg == 'VQITConnectorRegistry' else Q, n), n
    with pytest.raises(ImportError):           # the reference package is not in this image
        integration.register_into_reference()


def test_ordered_sum_policy():
    """ops.use_ordered: explicit request, torch's deterministic flag, size-based default, environment override."""
    import torch
    from vector_quantization_amd import ops
    assert ops.use_ordered(16384, 256, True) and not ops.use_ordered(16384, 256, False)
    assert not ops.use_ordered(16384, 256, None, 3072) and ops.use_ordered(16384, 256, None, 524288)
    assert ops.use_ordered(16384, 256, None, 65536) and not ops.use_ordered(16384, 256, None, 65536, backward=True)
    assert ops.use_ordered(1024, 256, None, 65536, backward=True) and not ops.use_ordered(65536, 256, None, 524288)
    with pytest.raises(ValueError):
        ops.use_ordered(65536, 256, True)
    prev = torch.are_deterministic_algorithms_enabled()
    try:
        torch.use_deterministic_algorithms(True)
        assert ops.use_ordered(16384, 256, None, 16)
        with pytest.raises(RuntimeError):
            ops.use_ordered(16384, 6, None, 16)
    finally:
        torch.use_deterministic_algorithms(prev)


def test_committed_bench_line_has_the_contract_fields():
    """The JSON line bench.py printed on the MI355X (committed under profiles/) carries every field of the driver's
    contract, the roofline object and the CPU baseline object."""
    import glob
    import json
    import os
    root = os.path.join(os.path.dirname(__file__), '..', 'profiles')
    path = sorted(glob.glob(os.path.join(root, 'r01_*_bench.json')))[-1]
    line = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in line, key
    assert line['scaling'] == 'weak' and line['vs_baseline'] is None and line['data'] == 'synthetic'
    assert 'workload' in line['config'] and 'model' not in line['config']
    roof = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in roof, key
    assert roof['bound'] in ('hbm', 'mfma') and abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9
    tokens = line['config']['tokens_per_gpu_per_step'] * line['n_gpus'] * 1e3 / line['ms_per_step']
    assert abs(tokens - line['value']) / line['value'] < 1e-6


def test_bench_gpus_n_self_launch_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with no launcher environment starts the ranks itself (torch.distributed.run child
    process); here there is no GPU, so the ranks fail and the parent must exit non-zero without printing a JSON line
    (never a silent 1-rank number)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert '"metric"' not in p.stdout
    assert '2-rank launch failed' in p.stderr
    # ... and it is refused BEFORE anything is spawned: more ranks than devices is never turned into a smaller run
    assert 'this node has 0 GPU(s); refusing to start' in p.stderr and 'torch.distributed' not in p.stderr


def test_bench_counts_gpus_from_sysfs_without_touching_a_device(tmp_path, monkeypatch):
    """The launching parent of `bench.py --gpus N` counts KFD topology nodes with SIMDs (round-5 review: no torch.cuda call, no
    device opened) and honours a HIP / ROCR visibility mask."""
    import glob as globmod
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    nodes = []
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):           # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / 'properties').write_text(f'cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n')
        nodes.append(str(d / 'properties'))
    monkeypatch.setattr(os.path, 'isdir', lambda p: True if p == '/sys/class/kfd' else os.path.exists(p))
    monkeypatch.setattr(globmod, 'glob', lambda pat: nodes)
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    assert bench.count_gpus_sysfs() == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,2')
    assert bench.count_gpus_sysfs() == 2


def test_init_registry_resolves_any_torch_nn_init_name():
    """vq/algorithms/vq/quantizers.py:87-90: the generic branch builds `InitRegistry.build(config)` for whatever initialiser the
    config names; here every in-place function of torch.nn.init resolves by its own name (round-5 review, missing #3)."""
    from vector_quantization_amd import Config, build_quantizer
    emb = dict(type='torch_nn_modules_sparse_Embedding', num_embeddings=64, embedding_dim=16)
    for init, check in ((dict(type='trunc_normal_', std=0.02, a=-0.04, b=0.04), lambda w: float(w.abs().max()) <= 0.04 + 1e-6),
                        (dict(type='constant_', val=0.25), lambda w: bool((w == 0.25).all())),
                        (dict(type='uniform_', a=-0.5, b=0.5), lambda w: float(w.abs().max()) <= 0.5),
                        (dict(type='xavier_uniform_'), lambda w: float(w.std()) > 0)):
        q = build_quantizer(dict(type='VectorQuantizer', embedding=emb, distance=dict(type='L2Distance')))
        q.init_weights(Config(init))
        assert check(q.embedding.weight.detach()), init
    with pytest.raises(Exception):
        q.init_weights(Config(type='no_such_initialiser_'))


class _StubDistance(Q.BaseDistance):
    """torch.cdist on whatever device the operands live on — only to exercise LazyDistance's tensor behaviour on CPU
    (the product distances have no CPU path)."""
    metric = 'L2'

    def __init__(self):
        super().__init__()
        self.materialisations = 0

    def forward(self, x, e):
        self.materialisations += 1
        return torch.cdist(x, e)

    def argmin(self, x, e, hist=None, prepared=None):
        return torch.cdist(x.detach(), e.detach()).argmin(-1)


def test_lazy_distance_is_a_tensor_for_unknown_consumers():
    """memo['distance'] must work for callbacks/losses written against the reference's plain tensor:
    metadata without materialising, fused argmin without materialising, everything else on the real matrix, with
    autograd to both operands."""
    import einops
    dist = _StubDistance()
    x = torch.randn(10, 4, requires_grad=True)
    e = torch.randn(6, 4, requires_grad=True)
    d = Q.LazyDistance(dist, x, e)
    assert isinstance(d, torch.Tensor) and d.shape == (10, 6) and d.dtype == torch.float32 and d.dim() == 2 and len(d) == 10
    assert torch.equal(d.argmin(-1), torch.cdist(x, e).argmin(-1)) and torch.equal(torch.argmin(d, dim=1), d.argmin(-1))
    assert dist.materialisations == 0
    ref = torch.cdist(x, e)
    t = einops.rearrange(d, 'x e -> e x')                       # MultinomialAnchor's first line (anchors.py:98)
    assert type(t) is torch.Tensor and torch.equal(t, ref.t()) and dist.materialisations == 1
    assert torch.equal(d[2], ref[2]) and torch.equal(1 - d, 1 - ref) and torch.equal(d / 0.5, ref / 0.5)
    assert torch.equal(torch.cat([d, d]), torch.cat([ref, ref])) and torch.equal(d.T.softmax(1), ref.T.softmax(1))
    assert dist.materialisations == 1                           # materialised once, cached
    (d / 0.5).softmax(-1)[:, 0].sum().backward()                # EntropyLoss-style use: gradients reach x and e
    assert x.grad.abs().sum() > 0 and e.grad.abs().sum() > 0
    assert 'LazyDistance' in repr({'distance': d})


def test_composed_callback_threads_values_in_priority_order():
    class Add(Q.BaseCallback):
        def __init__(self, k):
            super().__init__()
            self.k = k

        def before_encode(self, x, memo):
            memo.setdefault('order', []).append(self.k)
            return x * 10 + self.k

        def before_loss(self, z, x, memo):
            return z + self.k, x - self.k

        def after_init_weights(self, config, recursive):
            return recursive and self.k != 2

    cc = Q.ComposedCallback(priorities=[dict(before_encode=5), dict(), dict(before_encode=-1)], callbacks=[Add(1), Add(2), Add(3)])
    memo = {}
    out = cc.before_encode(torch.zeros(()), memo)
    assert memo['order'] == [3, 2, 1] and float(out) == 321.0
    z, x = cc.before_loss(torch.zeros(()), torch.zeros(()), {})
    assert float(z) == 6.0 and float(x) == -6.0
    assert cc.after_init_weights(Config(), True) is False
    assert cc.after_decode('z', {}) == 'z' and cc.before_init_weights(Config()) is None
    sentinel = object()
    cc.bind(sentinel)
    assert cc.quantizer is sentinel and all(cb.quantizer is sentinel for cb in cc.callbacks)


def test_cosine_distance_follows_autocast(monkeypatch):
    """CosineDistance() built from an unchanged config ('auto'): the bf16-autocast metric exactly while the caller is inside
    torch.autocast('cuda', bfloat16) — the region the reference's AutocastCallback opens (vq/runners/base.py:30-48) — and
    the fp32 definition otherwise (fp16 autocast included).  No GPU here: the autocast state is stubbed."""
    from vector_quantization_amd.quantizers import distances as DM
    q = build_quantizer(dict(type='VQKDQuantizer', embedding=dict(type=EMB, num_embeddings=8, embedding_dim=4),
                             distance=dict(type='CosineDistance'), callbacks=[dict(type='VQKDCallback', ema=dict())],
                             losses=dict(commitment_loss=dict(type='CommitmentLoss', mse=dict(norm=True)))))
    d = q.distance
    assert d.metric == 'Cosine'
    state = dict(on=True, dtype=torch.bfloat16)
    monkeypatch.setattr(torch, 'is_autocast_enabled', lambda *a: state['on'])
    monkeypatch.setattr(torch, 'get_autocast_dtype', lambda *a: state['dtype'])
    assert DM.autocast_bf16_active() and d.metric == 'CosineBF16'
    assert Q.CosineDistance(autocast=None).metric == 'Cosine' and Q.L2Distance().metric == 'L2'
    state['dtype'] = torch.float16
    assert d.metric == 'Cosine'
    state['on'] = False
    assert d.metric == 'Cosine' and Q.CosineDistance(autocast='bf16').metric == 'CosineBF16'
    with pytest.raises(ValueError):
        Q.CosineDistance(autocast='fp16')


def test_autocast_auto_keeps_fp32_definition_where_the_bf16_form_does_not_exist(monkeypatch):
    """'auto' must not turn a config that ran outside autocast into a failure inside it — at a D without the
    proposal image (D % 8 != 0 or D > 1024) the fused encode keeps the fp32 definition; an explicit 'bf16' still asks for it."""
    monkeypatch.setattr(torch, 'is_autocast_enabled', lambda *a: True)
    monkeypatch.setattr(torch, 'get_autocast_dtype', lambda *a: torch.bfloat16)
    d = Q.CosineDistance()
    assert d.metric == 'CosineBF16'
    assert d.metric_for(32) == 'CosineBF16' and d.metric_for(1024) == 'CosineBF16'
    assert d.metric_for(12) == 'Cosine' and d.metric_for(1032) == 'Cosine'
    assert Q.CosineDistance(autocast='bf16').metric_for(12) == 'CosineBF16'
    assert Q.L2Distance().metric_for(12) == 'L2'


def test_exchange_route_selection(monkeypatch):
    """rccl.py: which route the packed all-reduce takes.  No process group: none (callers skip the exchange); an invalid
    VQHIP_ALLREDUCE is refused; a CPU tensor never takes the library's communicator; VQ_FORCE_EXCHANGE only matters inside a
    process group."""
    from vector_quantization_amd import rccl, utils
    monkeypatch.delenv('VQHIP_ALLREDUCE', raising=False)
    assert rccl.mode() == 'auto'
    monkeypatch.setenv('VQHIP_ALLREDUCE', 'sideways')
    with pytest.raises(ValueError):
        rccl.mode()
    monkeypatch.setenv('VQHIP_ALLREDUCE', 'direct')
    assert rccl.communicator(torch.zeros(4)) is None            # no process group here
    monkeypatch.setenv('VQ_FORCE_EXCHANGE', '1')
    assert utils.exchanging() is False and utils.get_world_size() == 1
    assert rccl.status()['direct'] is False


def test_map_route_declines_modules_with_forward_hooks():
    """the NCHW map entry points are called directly (not through nn.Module.__call__), so a module with a
    registered forward (pre-)hook must take the token route, where the hook runs."""
    q = build_quantizer(dict(type='VQGANQuantizer', embedding=dict(type=EMB, num_embeddings=64, embedding_dim=8),
                             distance=dict(type='L2Distance'), losses=dict(vqgan_loss=dict(type='VQGANLoss'))))

    class FakeMap:                                               # what map_fusable looks at, without a device
        shape = (2, 8, 4, 4)
        dtype = torch.float32
        is_cuda = True
        def dim(self): return 4
        def is_contiguous(self): return True
        def data_ptr(self): return 0
    assert q.map_fusable(FakeMap()) is True
    h = q.register_forward_pre_hook(lambda m, a: None)
    assert q.map_fusable(FakeMap()) is False
    h.remove()
    assert q.map_fusable(FakeMap()) is True
    h = q.register_forward_hook(lambda m, a, o: None)
    assert q.map_fusable(FakeMap()) is False
