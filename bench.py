#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched DIYGym step path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ur_high_5] [--envs-per-gpu 16384]

One "step" = one DIYGym.step() over the whole batch: controller addons (batched
IK for ur_high_5) -> one 1/240 s physics step (2 substeps, <=150 PGS iterations)
-> sensor / reward / terminal addons, followed by the masked auto-reset of the
envs whose terminal fired.  Inputs are synthetic uniform random
actions within each addon's declared action_space, generated before the timed
region and already resident in HBM.  Weak scaling: every rank owns
--envs-per-gpu independent envs; there is no collective on the data path (envs
never interact); the only collectives are the barrier and the max-over-ranks of
the elapsed time that the bench contract asks for.

Rank 0 prints ONE JSON line (schema in the task statement) with two extra
objects: `roofline` (dominant kernel = step_kernel, timed live with HIP events
on the launch stream) and `cpu_baseline` (the C oracle -- a port, NOT pybullet
-- on the box's host cores, N=1 only, bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (config file, metric config description)
    'ur_high_5': ('examples/ur_high_5/ur_high_5.yaml', 'ur_high_5.yaml as in the reference: 2x UR5, ik_controller(use_orientation) + joint_state_sensor + object_state_sensor + reach_target'),
    'ur_high_5_joint': ('examples/ur_high_5/ur_high_5_joint.yaml', 'VARIANT of ur_high_5 with joint_controller(position) instead of ik_controller'),
    'drone_pilot': ('examples/drone_pilot/drone_pilot.yaml', 'drone_pilot.yaml as in the reference: quadrotor + 4 propellor + fell_over + reach_target'),
    'r2d2_maze': ('examples/r2d2_maze/r2d2_maze.yaml', 'r2d2_maze: R2D2 stand-in (mass 50, 4 velocity-driven wheels) among 119 fixed walls, tools/generate_maze.py --seed 7'),
    'from_the_readme': ('examples/from_the_readme/from_the_readme.yaml', 'from_the_readme.yaml: Jaco + table + 1:10 R2D2 with a 200x200 gripper camera (rendered every step)'),
    'marbles': ('tests/golden/basic_env_nocam.yaml', 'reference test fixture basic_env.yaml minus the camera: 3 marbles + plane + external_force'),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def action_bounds(env):
    from diy_gym_amd.utils import flatten, get_bounds_for_space
    lo = flatten(get_bounds_for_space(env.action_space, True))
    hi = flatten(get_bounds_for_space(env.action_space, False))
    return torch.as_tensor(lo, dtype=torch.float32), torch.as_tensor(hi, dtype=torch.float32)


def algorithmic_bytes_per_env_step(layout):
    """Compulsory HBM traffic of one env-step: persistent state read once and written once,
    actions read, observations / rewards / terminals / collapsed outputs written (DESIGN.md 'Measurement')."""
    return 2 * 4 * layout.state_dim + 4 * layout.act_dim + 4 * layout.obs_dim + 4 * layout.rew_dim + layout.term_dim + 4 + 1


def cpu_baseline(cfg, act_dim, lo, hi, seconds=12.0):
    """Times the C oracle (same algorithm, fp64) on the host cores with the same workload."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_backend
    from diy_gym_amd import DIYGym
    omp = os.path.join(ROOT, 'oracle', 'libdgsim_oracle_omp.so')
    cores = usable_cores()
    os.environ['OMP_NUM_THREADS'] = str(cores)
    if os.path.isfile(omp):
        oracle_backend._LIB = None
        oracle_backend.ORACLE_LIB = omp
    else:
        cores = 1
    envs = 64 * cores
    env = DIYGym(cfg, num_envs=envs, seed=1234, backend_factory=oracle_backend.OracleBackend)
    gen = torch.Generator().manual_seed(99)
    act = lo + (hi - lo) * torch.rand((envs, act_dim), generator=gen)
    env.sim.step(env._all_slots, act)  # warm
    steps, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        env.sim.step(env._all_slots, act)
        steps += 1
    dt = time.time() - t0
    return {'value': envs * steps / dt, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d envs x %d steps of the same config in %.1f s, C oracle (fp64, OpenMP over envs); pybullet itself is %s' %
                      (envs, steps, dt, pybullet_status())}


def usable_cores():
    """Host cores this job may actually use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def pybullet_status():
    try:
        import pybullet  # noqa: F401
        return 'importable here but not timed (no reference code travels to the GPU box)'
    except Exception:
        return 'unavailable on this box'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--workload', default='ur_high_5', choices=sorted(WORKLOADS))
    ap.add_argument('--envs-per-gpu', type=int, default=None, help='default: the size BASELINE.json quotes for the workload')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-auto-reset', action='store_true')
    ap.add_argument('--eager', action='store_true', help='time the eager launch loop instead of a replayed hipGraph')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', device_id=device)

    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    cfg_rel, cfg_desc = WORKLOADS[args.workload]
    cfg = os.path.join(ROOT, cfg_rel)
    B = args.envs_per_gpu or {'r2d2_maze': 4096, 'from_the_readme': 1024}.get(args.workload, 16384)
    env = DIYGym(cfg, num_envs=B, device=device, seed=1234, env_index_base=rank * B)
    lo, hi = action_bounds(env)
    gen = torch.Generator().manual_seed(1234 + rank)
    ring = [(lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to(device) for _ in range(8)]
    sim, slots = env.sim, env._all_slots
    auto_reset = not args.no_auto_reset

    def one_step(i):
        sim.step(slots, ring[i % len(ring)])
        if auto_reset:
            sim.reset(sim.term_flag)

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    # The timed region replays a hipGraph of len(ring) consecutive steps (step + masked auto-reset each): the work
    # is identical to the eager loop, but a busy host cannot stretch the gaps between the ~0.3 ms kernels.
    graph, R = None, len(ring)
    if not args.eager:
        try:
            cap = torch.cuda.Stream(device=device)
            cap.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(cap):
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=cap):
                    for i in range(R):
                        one_step(i)
            torch.cuda.current_stream(device).wait_stream(cap)
            torch.cuda.synchronize()
        except Exception as exc:  # pragma: no cover
            print('graph capture failed (%s); timing the eager loop' % exc, file=sys.stderr)
            graph = None
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    if graph is not None:
        for _ in range(args.steps // R):
            graph.replay()
        done = (args.steps // R) * R
    for i in range(done, args.steps):
        one_step(i)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # dominant-kernel duration: HIP events around dg_world_step on the launch stream, same inputs, right after the
    # timed region (events cannot be recorded per launch inside a replayed graph)
    ncal = min(64, args.steps)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ncal)]
    for i in range(ncal):
        ev[i][0].record()
        sim.step(slots, ring[i % R])
        ev[i][1].record()
        if auto_reset:
            sim.reset(sim.term_flag)
    torch.cuda.synchronize()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    resets = float(sim.state[1, :B].sum().item())  # DG_ST_EPISODE summed over envs

    if rank == 0:
        total_envs = B * world
        value = total_envs * args.steps / elapsed
        bytes_unit = algorithmic_bytes_per_env_step(env.layout)
        achieved = bytes_unit * B / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'r1_pmc_traffic.json')
        if os.path.isfile(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload)
            except Exception:
                traffic = None
        # informative: share of the chip's VALU issue slots the kernel used (wave-instructions from the committed PMC
        # pass of the same command x 4 cycles each, over 256 CUs x 4 SIMDs x the live kernel time at 2.4 GHz)
        valu_frac = None
        pmc_sq = os.path.join(ROOT, 'profiles', 'r1_ur_high_5_16384_pmc_step_kernel.json')
        if args.workload == 'ur_high_5' and B == 16384 and os.path.isfile(pmc_sq):
            try:
                valu = json.load(open(pmc_sq))['per_launch_means'].get('SQ_INSTS_VALU')
                valu_frac = valu * 4.0 / (1024 * kernel_ms * 1e-3 * 2.4e9) if valu else None
            except Exception:
                valu_frac = None
        out = {
            'metric': 'env steps/sec (whole node)', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s x %d envs per GPU' % (args.workload, B), 'what': cfg_desc, 'envs_total': total_envs,
                       'timestep': 1.0 / 240.0, 'substeps': env.layout.substeps, 'solver_iterations': int(env.builder.solver_iterations),
                       'auto_reset': auto_reset, 'launch': 'hipGraph replay of %d-step segments' % R if graph is not None else 'eager', 'episodes_finished_rank0': resets - B, 'parallelism': 'independent env shards x%d, no collective' % world,
                       'envs_per_wavefront': sim.lanes, 'lds_bytes_per_workgroup': sim.lds_bytes},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'kernel': 'step_kernel_par' if getattr(sim, 'lanes', 0) == 64 and args.workload.startswith('ur_high_5') else 'step_kernel', 'kernel_ms': kernel_ms, 'bytes_per_env_step': bytes_unit,
                         'survey_bytes_per_env_step': 449 if args.workload.startswith('ur_high_5') else None,
                         'valu_issue_frac': valu_frac,
                         'note': 'the step kernel keeps all per-env scratch in LDS; it is bound by instruction issue and latency of one '
                                 'wavefront per SIMD (three per workgroup), not by HBM (DESIGN.md Measurement)'},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, lo.numel(), lo, hi)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
