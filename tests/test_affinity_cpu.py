"""Host-thread placement helper (vector_quantization_amd/affinity.py): parsing, slice grouping, bind + restore on this machine."""
import os

import pytest

from vector_quantization_amd import affinity


def test_cpulist_parsing():
    assert affinity._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert affinity._parse_cpulist('') == []
    assert affinity._parse_cpulist('5') == [5]


def test_l3_slices_cover_the_set_once():
    allowed = set(os.sched_getaffinity(0))
    groups = affinity.l3_slices(allowed)
    flat = [c for g in groups for c in g]
    assert sorted(flat) == sorted(allowed)
    assert len(flat) == len(set(flat))


@pytest.mark.skipif(not hasattr(os, 'sched_setaffinity'), reason='no sched_setaffinity')
def test_bind_rank_and_restore():
    before = set(os.sched_getaffinity(0))
    try:
        info = affinity.bind_rank(device_index=0, local_rank=1)
        assert info is not None and set(info['cpus']) <= before and info['cpus']
        assert set(os.sched_getaffinity(0)) == set(info['cpus'])
        # every thread of the process was moved, not only the caller
        for tid in os.listdir('/proc/self/task'):
            assert set(os.sched_getaffinity(int(tid))) == set(info['cpus'])
        affinity.restore(info['previous'])
        assert set(os.sched_getaffinity(0)) == before
    finally:
        affinity.set_process_affinity(before)


def test_ranks_take_successive_slices():
    before = set(os.sched_getaffinity(0))
    groups = affinity.l3_slices(before)
    try:
        a = affinity.bind_rank(0, 0)
        affinity.restore(a['previous'])
        b = affinity.bind_rank(0, 1)
        affinity.restore(b['previous'])
        if len(groups) > 1 and not a['numa_local']:
            assert set(a['cpus']).isdisjoint(b['cpus'])
    finally:
        affinity.set_process_affinity(before)


def test_ranks_sharing_a_numa_node_take_successive_slices_by_position(monkeypatch):
    """Round-5 advisor: with GPUs 0 and 2 on one NUMA node (NPS4-style topologies) the raw local rank put ranks 0 and 2 on the same
    slice.  The slice index is the rank's position among the local ranks whose GPU hangs off the same node."""
    node_of = {0: [0, 1, 2, 3], 1: [4, 5, 6, 7], 2: [0, 1, 2, 3], 3: [4, 5, 6, 7]}        # GPUs 0, 2 on one node; 1, 3 on the other
    monkeypatch.setattr(affinity, 'gpu_local_cpus', lambda i: node_of.get(i))
    assert affinity.node_position(0, 0, 4) == (0, 2) and affinity.node_position(2, 2, 4) == (1, 2)
    assert affinity.node_position(1, 1, 4) == (0, 2) and affinity.node_position(3, 3, 4) == (1, 2)
    assert affinity.node_position(0, 3, 4) == (3, 4)              # a rank that does not drive the device of its number: raw local rank
    monkeypatch.setattr(affinity, 'gpu_local_cpus', lambda i: None)
    assert affinity.node_position(2, 2, 8) == (2, 8)              # topology unreadable: raw local rank
