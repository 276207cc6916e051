"""CPU: the per-rank shares of BASELINE configs 3-5 (bench_support/configs.py) partition the job with no
overlap -- disjoint streams, disjoint joint states, disjoint URDFs -- and the union over the ranks is the whole job."""
import numpy as np
import pytest

from realtime_urdf_filter_amd import sharding
from bench_support import configs as CF


def test_shard_helpers():
    for total, world in ((512, 8), (7, 2), (5, 3), (3, 4)):
        parts = [sharding.shard_range(total, world, r) for r in range(world)]
        assert sum(c for _, c in parts) == total
        assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1)) and parts[0][0] == 0
    assert [sharding.models_for_rank(64, 8, r) for r in (0, 7)] == [list(range(0, 64, 8)), list(range(7, 64, 8))]
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def test_c4_shares_are_disjoint_blocks_with_distinct_joint_states():
    world, total = 3, 7
    shares = [CF.build("c4", world, r, streams=total, triangles=1500, width=160, height=96) for r in range(world)]
    assert [s.n for s in shares] == [3, 2, 2] and all(s.total_streams == total and s.scaling == "strong" for s in shares)
    firsts = [s.groups[0].global_first for s in shares]
    assert firsts == [0, 3, 5]
    q = np.concatenate([s.groups[0].variants[0].joint_q for s in shares])
    assert len({tuple(np.round(row, 12)) for row in q}) == total          # no joint state repeats across ranks
    # a share depends only on which streams it holds: rank 1 of 3 == streams 3..4 of the single-GPU job
    whole = CF.build("c4", 1, 0, streams=total, triangles=1500, width=160, height=96)
    assert np.array_equal(whole.groups[0].variants[0].joint_q[3:5], shares[1].groups[0].variants[0].joint_q)
    # two models per stream: the robot and the static walls
    assert len(shares[0].wl0.models) == 2 and shares[0].wl0.link_tf[1].shape[1] == 2


def test_c5_urdfs_live_on_rank_m_mod_world():
    world, urdfs, per = 2, 5, 3
    shares = [CF.build("c5", world, r, streams=per, triangles=3000, urdfs=urdfs, width=160, height=120) for r in range(world)]
    assert [[g.robot_index for g in s.groups] for s in shares] == [[0, 2, 4], [1, 3]]
    assert [s.n for s in shares] == [9, 6] and all(s.total_streams == 15 for s in shares)
    for s in shares:
        assert [g.first for g in s.groups] == [per * i for i in range(len(s.groups))]
        assert all(g.global_first == per * g.robot_index for g in s.groups)
    # distinct robots: different geometry per URDF number
    tri = {g.robot_index: g.variants[0].n_triangles() for s in shares for g in s.groups}
    assert len(tri) == urdfs and len(set(tri.values())) >= 3


def test_c3_is_weak_scaling_with_rank_distinct_streams():
    a, b = CF.build("c3", 2, 0, streams=4, triangles=1500, width=96, height=72), CF.build("c3", 2, 1, streams=4, triangles=1500, width=96, height=72)
    assert a.n == b.n == 4 and a.total_streams == 8 and a.scaling == "weak"
    assert not np.array_equal(a.wl0.joint_q, b.wl0.joint_q)
    assert not np.array_equal(a.depth_host(0)[0], b.depth_host(0)[0])
    assert a.triangles_per_stream().tolist() == [a.wl0.n_triangles()] * 4
