"""`python bench.py --gpus N` without torchrun's environment must run N ranks itself.  Exercised here on CPU through
the launcher's self-test body (same spawn / rendezvous / barrier / max-over-ranks, gloo instead of RCCL, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra)
    return env


def test_gpus_2_spawns_two_ranks_and_rank0_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--selftest-launcher', '--envs-per-gpu', '96'], env=clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1  # only rank 0 reports
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2
    assert sorted(tuple(x) for x in out['ranks']) == [(0, 0, 96), (1, 1, 96)]  # (RANK, LOCAL_RANK, envs per GPU)
    assert out['elapsed_max'] == 1.25  # max over ranks of 0.25 + rank
    assert out['cuda_initialized'] is False
    assert 'parent cuda_initialized=False' in r.stderr  # the parent stayed GPU-free


def test_a_dying_rank_fails_the_whole_run():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--selftest-launcher'], env=clean_env(DG_BENCH_SELFTEST_FAIL_RANK='1'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'rank 1 exited with status 3' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_more_gpus_than_visible_is_refused():
    # no GPU in the CPU container (and never 64 on one node): the launcher must refuse, not run fewer ranks
    r = subprocess.run([sys.executable, BENCH, '--gpus', '64', '--steps', '1'], env=clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert 'only' in r.stderr and 'visible' in r.stderr


def test_world_size_mismatch_is_refused_even_for_world_1():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--selftest-launcher'], env=clean_env(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'WORLD_SIZE=1' in r.stderr


def test_under_torchrun_environment_it_is_a_rank_not_a_launcher():
    # the driver's `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` form
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29731', BENCH, '--gpus', '2', '--selftest-launcher'], env=clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2
