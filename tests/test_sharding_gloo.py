"""N>1 path on CPU: two gloo ranks each own a shard of envs (env_index_base = rank * B).
There is no data-path collective; the only collectives are bench.py's barrier and the
max-over-ranks of the elapsed time.  Checks: the union of the shards equals one
process running 2B envs (respawn jitter streams are keyed by GLOBAL env index)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'examples', 'drone_pilot', 'drone_pilot.yaml')
B = 6


def rollout(num_envs, base, steps=5):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    from oracle_backend import OracleBackend
    env = DIYGym(CFG, num_envs=num_envs, seed=77, env_index_base=base, backend_factory=OracleBackend)
    gen = torch.Generator().manual_seed(5)
    acts = torch.rand((steps, 2 * B, 4), generator=gen)
    for s in range(steps):
        env.sim.step(env._all_slots, acts[s, base:base + num_envs])
    env.sim.reset(None)  # second episode: new jitter draw per env
    return env.sim.get_state()


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dist.barrier()
    state = rollout(B, rank * B)
    elapsed = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)  # the bench's max-over-ranks
    np.save(os.path.join(out, 'state_%d.npy' % rank), state)
    np.save(os.path.join(out, 'elapsed_%d.npy' % rank), elapsed.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_one_big_batch(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    shards = np.concatenate([np.load(tmp_path / ('state_%d.npy' % r)) for r in range(2)], axis=0)
    whole = rollout(2 * B, 0)
    assert np.array_equal(shards, whole)
    # target respawn jitter differs between envs and between ranks
    assert len({tuple(np.round(r, 6)) for r in shards}) == 2 * B
    assert np.load(tmp_path / 'elapsed_0.npy')[0] == 1.5 == np.load(tmp_path / 'elapsed_1.npy')[0]
