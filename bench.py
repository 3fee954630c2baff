#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched DIYGym step path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ur_high_5] [--envs-per-gpu 16384]

One "step" = one DIYGym.step() over the whole batch: controller addons (batched
IK for ur_high_5) -> one 1/240 s physics step (2 substeps, <=150 PGS iterations)
-> sensor / reward / terminal addons (and the camera render for workloads with a
camera), followed by the masked auto-reset of the envs whose terminal fired
(SURVEY.md 8d).  The timed region drives the backend entry points
(dg_world_step + dg_world_reset) the way a trainer would, from a replayed
hipGraph; the eager ``env.step()`` API rates (dict and flat) are measured
separately after it and reported as ``api_eager``.  Inputs are synthetic uniform
random actions within each addon's declared action_space, generated before the
timed region and already resident in HBM.

N > 1: weak scaling, one process per GPU, every rank owns --envs-per-gpu
independent envs; no collective on the data path (envs never interact); the only
collectives are the barrier and the max-over-ranks of the elapsed time the bench
contract asks for.  ``python bench.py --gpus N`` WITHOUT torchrun's environment
spawns the N ranks itself (the parent never touches a GPU); under
``python -m torch.distributed.run`` it is a rank.

Rank 0 prints ONE JSON line (schema in the task statement) with extra objects:
``roofline`` (dominant kernel timed live with HIP events on the launch stream and, at N=1, by rocprofv3 child runs of
this command that also collect the HBM / SQ counters -- ``--no-pmc`` opts out),
``solver`` (live Gauss-Seidel / IK iteration statistics from the kernel's
diagnostics buffer), ``aged`` (the same timed loop after --age-steps more
steps), ``api_eager`` and ``cpu_baseline`` (the C oracle -- a port, NOT pybullet
-- on the box's host cores, N=1 only, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (config file, metric config description)
    'ur_high_5': ('examples/ur_high_5/ur_high_5.yaml', 'ur_high_5.yaml as in the reference: 2x UR5, ik_controller(use_orientation) + joint_state_sensor + object_state_sensor + reach_target'),
    'ur_high_5_joint': ('examples/ur_high_5/ur_high_5_joint.yaml', 'VARIANT of ur_high_5 with joint_controller(position) instead of ik_controller'),
    'drone_pilot': ('examples/drone_pilot/drone_pilot.yaml', 'drone_pilot.yaml as in the reference: quadrotor + 4 propellor + fell_over + reach_target'),
    'r2d2_maze': ('examples/r2d2_maze/r2d2_maze.yaml', 'r2d2_maze: R2D2 stand-in (mass 50, 4 velocity-driven wheels) among 119 fixed walls, tools/generate_maze.py --seed 7'),
    'from_the_readme': ('examples/from_the_readme/from_the_readme.yaml', 'from_the_readme.yaml: Jaco + table + 1:10 R2D2; the 200x200 gripper camera (rgb + depth) is rendered inside every timed step'),
    'marbles': ('tests/golden/basic_env_nocam.yaml', 'reference test fixture basic_env.yaml minus the camera: 3 marbles + plane + external_force'),
    'ur5_gripper': ('tests/golden/ur5_gripper.yaml', 'UR5 with the two-finger gripper asset (12-DoF tree) next to a ground plane, joint_controller: random targets lay the arm on the ground (contact-rich)'),
    'ur5_child_gripper': ('tests/golden/ur5_child_gripper.yaml', 'UR5 with the robotiq_2f gripper attached as a child model (12-DoF tree, no ground): the contact-free arm + gripper case'),
}
DEFAULT_ENVS = {'r2d2_maze': 4096, 'from_the_readme': 1024}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMD-32; MI355X_MICROARCH.md


# ----------------------------------------------------------------------------------------------- launcher
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch(args, argv):
    """Parent of an N-rank run started as plain ``python bench.py --gpus N``: spawns one child per GPU with
    torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), forwards their output and exits with the
    first non-zero child status (the other children are then terminated by PID).  The parent makes no HIP / CUDA
    call -- ``torch.cuda.device_count()`` does not initialise the runtime on this image."""
    n = args.gpus
    if not args.selftest_launcher:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print('bench.py: --gpus %d but only %d GPU(s) visible; refusing to run fewer ranks than asked' % (n, have), file=sys.stderr)
            return 2
    port = args.master_port or free_port()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc, pending = 0, set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print('bench.py: rank %d exited with status %d; stopping the other ranks' % (r, code), file=sys.stderr)
                deadline = time.time() + 10.0  # a rank blocked in a collective after its peer died may ignore SIGTERM
                for o in pending:
                    procs[o].terminate()
                for o in sorted(pending):
                    try:
                        procs[o].wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        procs[o].kill()
                        procs[o].wait()
                pending = set()
                break
        time.sleep(0.05)
    if args.selftest_launcher:
        import torch
        print('launcher: parent cuda_initialized=%s' % torch.cuda.is_initialized(), file=sys.stderr)
    return rc


def selftest_rank(args):
    """CPU-only rank body for tests/test_bench_launcher.py: same rendezvous, barrier and max-over-ranks as the real
    bench (gloo instead of RCCL), no GPU call."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = int(os.environ['RANK']), int(os.environ['LOCAL_RANK']), int(os.environ['WORLD_SIZE'])
    if os.environ.get('DG_BENCH_SELFTEST_FAIL_RANK') == str(rank):
        sys.exit(3)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dist.barrier()
    elapsed = torch.tensor([0.25 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    seen = [None] * world
    dist.all_gather_object(seen, (rank, local_rank, args.envs_per_gpu or 16384))
    if rank == 0:
        print(json.dumps({'selftest': True, 'n_gpus': world, 'ranks': seen, 'elapsed_max': float(elapsed.item()),
                          'cuda_initialized': torch.cuda.is_initialized()}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------- helpers
def action_bounds(env):
    import torch
    from diy_gym_amd.utils import flatten, get_bounds_for_space
    space = getattr(env, 'original_action_space', env.action_space)
    lo = flatten(get_bounds_for_space(space, True))
    hi = flatten(get_bounds_for_space(space, False))
    return torch.as_tensor(lo, dtype=torch.float32), torch.as_tensor(hi, dtype=torch.float32)


def algorithmic_bytes_per_env_step(layout, image_bytes=0):
    """Compulsory HBM traffic of one env-step: persistent state read once and written once, actions read,
    observations / rewards / terminals / collapsed outputs written, camera images written (DESIGN.md 'Measurement')."""
    return 2 * 4 * layout.state_dim + 4 * layout.act_dim + 4 * layout.obs_dim + 4 * layout.rew_dim + layout.term_dim + 4 + 1 + image_bytes


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


def cpu_baseline(cfg, act_dim, lo, hi, seconds=4.0):
    """Times the C oracle (a port of the same algorithm, NOT pybullet) on the host cores with the same workload, as
    SURVEY.md 8d specifies: the fp32 build on 1 core and on all cores, and the fp64 build (the parity checker) on all cores."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_backend
    from diy_gym_amd import DIYGym
    cores = usable_cores()
    os.environ['OMP_NUM_THREADS'] = str(cores)

    def rate(flavour, envs, secs):
        factory = oracle_backend.flavour(flavour)
        if factory is None:
            return None
        env = DIYGym(cfg, num_envs=envs, seed=1234, backend_factory=factory)
        gen = torch.Generator().manual_seed(99)
        act = lo + (hi - lo) * torch.rand((envs, act_dim), generator=gen)
        env.sim.step(env._all_slots, act)  # warm
        steps, t0 = 0, time.time()
        while time.time() - t0 < secs:
            env.sim.step(env._all_slots, act)
            steps += 1
        dt = time.time() - t0
        env.close()
        return {'value': envs * steps / dt, 'envs': envs, 'steps': steps, 'seconds': round(dt, 2)}

    f32_1 = rate('f32', 64, seconds)
    f32_n = rate('f32_omp', 64 * cores, seconds)
    f64_n = rate('f64_omp', 64 * cores, seconds)
    head = f32_n or f32_1 or f64_n
    return {'value': head['value'], 'unit': 'env-steps/s', 'cores': cores if head is not f32_1 else 1, 'kind': 'port', 'nproc': os.cpu_count(),
            'fp32_1_core': f32_1, 'fp32_all_cores': f32_n, 'fp64_all_cores': f64_n,
            'sample': 'same config, random actions, ~%.0f s each: C oracle built as fp32 on 1 core (64 envs), fp32 with OpenMP over envs on %d cores (%d envs) '
                      '[= value], fp64 on %d cores; pybullet itself is %s' % (seconds, cores, 64 * cores, cores, pybullet_status())}


def quantiles(t):
    """mean / p50 / p99 / max of an integer tensor, as plain floats."""
    f = t.float().flatten()
    return {'mean': round(float(f.mean()), 2), 'p50': float(f.median()), 'p99': float(f.quantile(0.99)) if f.numel() < (1 << 24) else None,
            'max': float(f.max())}


def profile_children(args, argv):
    """N = 1, default on (``--no-pmc`` opts out): per-launch figures of the dominant kernel measured NOW by running this very
    command (fewer steps, timed loop only) under rocprofv3 in fresh child processes -- started before this process touches
    the GPU, ``python bench.py ...`` directly after ``--`` -- one pass each: a kernel trace (average duration), then the
    HBM counters and the SQ counters in their own passes (never a trace and counters together).  Corrections per
    MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB, and gfx950's FETCH_SIZE counts half of a coalesced read.
    Bounded: a pass that fails or runs out of time leaves its fields null."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    out, t_begin = {}, time.time()
    skip = ('--pmc', '--no-pmc')
    base = [a for a in argv if a not in skip]
    inner = ['--steps', '40', '--warmup', '10', '--inner']
    env = dict(os.environ, TMPDIR=os.environ.get('TMPDIR', '/tmp'))
    passes = [('trace', ['--kernel-trace', '--stats']), ('fetch', ['--pmc', 'FETCH_SIZE']), ('write', ['--pmc', 'WRITE_SIZE']),
              ('sq', ['--pmc', 'SQ_INSTS_VALU', 'SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_BUSY_CYCLES', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM'])]
    for name, flags in passes:
        left = args.pmc_budget - (time.time() - t_begin)
        if left < 20:
            print('bench.py: profiling budget (%d s) spent before the %s pass; its fields stay null' % (args.pmc_budget, name), file=sys.stderr)
            break
        d = tempfile.mkdtemp(prefix='dg_prof_', dir=env['TMPDIR'])
        cmd = ['rocprofv3'] + flags + ['--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__)] + base + inner
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left, check=True, cwd=env['TMPDIR'], env=env)
        except Exception as exc:
            print('bench.py: rocprofv3 %s pass failed (%s); its fields stay null' % (name, type(exc).__name__), file=sys.stderr)
            shutil.rmtree(d, ignore_errors=True)
            continue
        if name == 'trace':
            for f in glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if args.kernel_filter in r['Name'] and 'reset' not in r['Name']:
                        out['trace'] = {'kernel': r['Name'].split('(')[0].replace('void ', ''), 'calls': int(r['Calls']), 'average_ns': float(r['AverageNs']),
                                        'min_ns': float(r['MinNs']), 'max_ns': float(r['MaxNs'])}
                        break
        else:
            agg = {}
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if args.kernel_filter in r['Kernel_Name'] and 'reset' not in r['Kernel_Name']:
                        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
            for k, v in agg.items():
                out[k] = sum(v) / len(v)
        shutil.rmtree(d, ignore_errors=True)
    out['seconds'] = round(time.time() - t_begin, 1)
    return out


# ----------------------------------------------------------------------------------------------- one rank
def run_rank(args, argv):
    import numpy as np
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    distributed = world > 1

    pmc = None
    if not args.no_pmc and not distributed and not args.inner:
        args.kernel_filter = 'render_kernel' if args.workload == 'from_the_readme' else 'step_kernel'
        pmc = profile_children(args, argv)  # child processes; this process has not touched the GPU yet

    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', device_id=device)

    import diy_gym_amd.examples  # noqa: F401
    from diy_gym_amd import DIYGym
    from diy_gym_amd.config import Configuration
    cfg_rel, cfg_desc = WORKLOADS[args.workload]
    cfg = os.path.join(ROOT, cfg_rel)
    B = args.envs_per_gpu or DEFAULT_ENVS.get(args.workload, 16384)
    env = DIYGym(cfg, num_envs=B, device=device, seed=1234, env_index_base=rank * B)
    lo, hi = action_bounds(env)
    gen = torch.Generator().manual_seed(1234 + rank)
    ring = [(lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to(device) for _ in range(8)]
    sim, slots = env.sim, env._all_slots
    auto_reset = not args.no_auto_reset
    cameras = [a for r in env.receptors.values() for a in r.addons.values() if hasattr(a, 'camera_index')]
    image_bytes = 0
    for cam in cameras:  # allocate the image buffers once; the timed step renders into them
        cam.observe()
        image_bytes += sum(t.numel() * t.element_size() for t in cam._buffers if t is not None) // B

    def one_step(i):
        sim.step(slots, ring[i % len(ring)])
        for cam in cameras:
            sim.render(cam.camera_index, *cam._buffers)
        if auto_reset:
            sim.reset(sim.term_flag)

    # Warm-up: --warmup untimed steps in all.  The last ones are the FIRST replay of each graph the timed region uses (below):
    # the first launch of a freshly instantiated hipGraph can pay a one-off upload of tens of milliseconds, which in a timed
    # region of ~30 ms in all showed up once as 0.26 ms per step instead of 0.108 (profiles/, round 3).
    R = len(ring)
    n_graph_warm = 0 if args.eager else R + (args.steps % R)
    if args.warmup < n_graph_warm:
        n_graph_warm = 0  # (too few warm-up steps asked for to spend them on the graphs: the first replay is timed)
    for i in range(args.warmup - n_graph_warm):
        one_step(i)
    torch.cuda.synchronize()
    # The timed region replays hipGraphs of consecutive steps (step [+ render] + masked auto-reset each): the work is
    # identical to the eager loop, but a busy host cannot stretch the gaps between the ~0.1 ms kernels.  EXACTLY --steps
    # steps, all of them replayed: whole segments of len(ring) steps plus one shorter segment for the remainder.
    def capture(n_steps, body):
        cap = torch.cuda.Stream(device=device)
        cap.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(cap):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                for i in range(n_steps):
                    body(i)
        torch.cuda.current_stream(device).wait_stream(cap)
        torch.cuda.synchronize()
        return g

    graph = graph_rest = graph_kernel = None
    if not args.eager:
        try:
            graph = capture(R, one_step)
            if args.steps % R:
                graph_rest = capture(args.steps % R, one_step)
            graph_kernel = capture(R, lambda i: sim.step(slots, ring[i % R]))  # the step kernel alone, for kernel_times()
        except Exception as exc:  # pragma: no cover
            print('graph capture failed (%s); timing the eager loop' % exc, file=sys.stderr)
            graph = graph_rest = graph_kernel = None
    if graph is not None and n_graph_warm:  # the remaining warm-up steps, as the graphs' first replays
        graph.replay()
        if graph_rest is not None:
            graph_rest.replay()
        torch.cuda.synchronize()
    elif graph is None:
        for i in range(n_graph_warm):
            one_step(i)
        torch.cuda.synchronize()

    def timed(steps):
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graph is not None:
            for _ in range(steps // R):
                graph.replay()
            if steps % R:
                graph_rest.replay()
        else:
            for i in range(steps):
                one_step(i)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed(args.steps)
    episodes_main = float(sim.state[1, :B].sum().item()) - B  # DG_ST_EPISODE summed over envs

    def kernel_times(n):
        """Average duration of the dominant kernel(s) by HIP events on the launch stream (torch's current stream IS the
        stream the C-ABI launches on), same inputs, right after the timed region.  The step kernel: events around
        replays of a graph of len(ring) back-to-back step launches (no host launch gaps inside); eager launches
        bracketed one by one -- launch overhead included -- are reported next to it.  The render kernel: eager."""
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
        for i in range(n):
            ev[i][0].record()
            sim.step(slots, ring[i % R])
            ev[i][1].record()
            for cam in cameras:
                sim.render(cam.camera_index, *cam._buffers)
            ev[i][2].record()
            if auto_reset:
                sim.reset(sim.term_flag)
        torch.cuda.synchronize()
        eager = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        render = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        replayed = None
        if graph_kernel is not None:
            reps = max(1, n // R)
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(reps):
                graph_kernel.replay()
            g1.record()
            torch.cuda.synchronize()
            replayed = g0.elapsed_time(g1) / (reps * R)
            if auto_reset:
                sim.reset(sim.term_flag)
        return (replayed if replayed is not None else eager), render, eager

    step_ms, render_ms, step_eager_ms = kernel_times(max(R, min(64, args.steps)))

    # live solver statistics from the kernel's diagnostics buffer (a separate, untimed segment: the production
    # launches above carry no diagnostics)
    def solver_stats():
        d = sim.enable_diagnostics()
        acc = []
        for i in range(16):
            one_step(i)
            acc.append(d.clone())
        torch.cuda.synchronize()
        sim.disable_diagnostics()
        D = torch.stack(acc)  # [16, B, 8]
        n_ik = sum(1 for r in env.receptors.values() for a in r.addons.values() if type(a).__name__ == 'InverseKinematicsController')
        per_wave = D[:, :, sim.DIAG_PGS_ITERS].reshape(16, -1, min(sim.envs_per_wave, B)).max(2).values if B % sim.envs_per_wave == 0 else None
        return {'iteration_cap': int(env.builder.solver_iterations), 'substeps': env.layout.substeps,
                'pgs_iterations_last_substep': quantiles(D[:, :, sim.DIAG_PGS_ITERS]),
                'pgs_iterations_first_substep': quantiles(D[:, :, sim.DIAG_PGS_ITERS_FIRST]),
                'pgs_iterations_wavefront_max': quantiles(per_wave) if per_wave is not None else None,
                'contacts_per_env': quantiles(D[:, :, sim.DIAG_CONTACTS]),
                'envs_with_contacts_frac': float((D[:, :, sim.DIAG_CONTACTS] > 0).float().mean()),
                'ik_iteration_cap': int(env.builder.params['ik_iterations']) if n_ik else None,
                'ik_iterations': quantiles(D[:, :, sim.DIAG_IK_ITERS:sim.DIAG_IK_ITERS + min(n_ik, sim.DIAG_N_IK)]) if n_ik else None,
                'sample': '16 steps after the timed region, every env'}

    solver = solver_stats() if rank == 0 and not args.inner else None

    # (measured BEFORE the aged segment, so that the rates are those of the API on the same young rollout as `value`)
    # eager public-API rates: env.step() with the reference's dict actions, and with flatten_actions /
    # flatten_observations + collapsed reward / terminal (the trainer-facing fast path); auto-reset as above
    api = None
    if rank == 0 and not args.inner and not args.no_api:
        from diy_gym_amd.utils import unflatten
        n_api = 100
        dict_ring = [unflatten(r, env.action_space, batch_dims=1) for r in ring] if lo.numel() else [{} for _ in ring]
        for i in range(5):
            env.step(dict_ring[i % R])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_api):
            _, _, term, _ = env.step(dict_ring[i % R])
            if auto_reset:
                env.reset(sim.term_flag)
        torch.cuda.synchronize()
        dict_ms = (time.perf_counter() - t0) / n_api * 1e3
        api = {'steps': n_api, 'dict_api_ms_per_step': dict_ms, 'dict_api_env_steps_per_s': B / dict_ms * 1e3}
        try:
            conf = Configuration.from_file(cfg)
            for k, v in (('flatten_actions', True), ('flatten_observations', True), ('sum_rewards', True)):
                conf.set(k, v)
            if not (conf.get('terminal_if_any', False) or conf.get('terminal_if_all', False)):
                conf.set('terminal_if_any', True)
            env2 = DIYGym(conf, num_envs=B, device=device, seed=1234, env_index_base=rank * B)
            for i in range(5):
                env2.step(ring[i % R])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_api):
                _, _, term, _ = env2.step(ring[i % R])
                if auto_reset:
                    env2.reset(term)
            torch.cuda.synchronize()
            flat_ms = (time.perf_counter() - t0) / n_api * 1e3
            api.update({'flat_api_ms_per_step': flat_ms, 'flat_api_env_steps_per_s': B / flat_ms * 1e3})
            env2.close()
        except Exception as exc:  # pragma: no cover
            api['flat_api_error'] = repr(exc)

    # aged segment: the same loop after --age-steps more (untimed) steps of the same random-action rollout
    aged = None
    if args.age_steps > 0 and not args.inner:
        for i in range(args.age_steps // R if graph is not None else 0):
            graph.replay()
        for i in range((args.age_steps // R) * R if graph is not None else 0, args.age_steps):
            one_step(i)
        el = timed(args.steps)
        a_step_ms, a_render_ms, _ = kernel_times(max(R, min(32, args.steps)))
        aged = {'after_steps': args.warmup + args.steps + 80 + args.age_steps + (105 if api else 0), 'ms_per_step_aged': el / args.steps * 1e3,
                'value_aged': B * world * args.steps / el, 'kernel_ms_aged': a_step_ms,
                'episodes_finished_rank0': float(sim.state[1, :B].sum().item()) - B,
                'solver': solver_stats() if rank == 0 else None}

    if rank == 0:
        total_envs = B * world
        value = total_envs * args.steps / elapsed
        render_bound = bool(cameras) and render_ms > step_ms * 0.2
        state_bytes = algorithmic_bytes_per_env_step(env.layout)
        # the dominant kernel for the roofline: the step kernel, except for camera workloads whose image writes are
        # the HBM-bound part of the step -- there the render kernel is quoted
        if args.workload == 'from_the_readme':
            kname, kms, kbytes = 'render_kernel', render_ms, image_bytes
        else:
            kname, kms, kbytes = ('step_kernel_par' if getattr(sim, 'par', False) else 'step_kernel'), step_ms, state_bytes
        # kernel_ms: this kernel's average launch duration -- from the rocprofv3 kernel trace of a child run of this very
        # command when there is one (it cannot include launch gaps), else from the HIP events above
        kms_events, ksource = kms, 'HIP events around %s' % ('graph replays of back-to-back launches' if (graph_kernel is not None and kname != 'render_kernel') else 'eager launches')
        trace = pmc.get('trace') if pmc else None
        if trace:
            kms, ksource = trace['average_ns'] * 1e-6, 'rocprofv3 --kernel-trace --stats of a child run of this command (%d launches)' % trace['calls']
        achieved = kbytes * B / (kms * 1e-3) / 1e9
        traffic = valu_frac = wait_frac = issue = None
        if pmc:
            if 'FETCH_SIZE' in pmc and 'WRITE_SIZE' in pmc:
                traffic = (2.0 * pmc['FETCH_SIZE'] + pmc['WRITE_SIZE']) * 1024.0
            if 'SQ_INSTS_VALU' in pmc:  # one VALU wave-instruction occupies a SIMD-32 for 2 cycles (MI355X_MICROARCH.md)
                valu_frac = pmc['SQ_INSTS_VALU'] * 2.0 / (SIMDS * kms * 1e-3 * CLOCK_HZ)
            if pmc.get('SQ_WAVE_CYCLES'):
                wait_frac = pmc.get('SQ_WAIT_ANY', 0.0) / pmc['SQ_WAVE_CYCLES']
            if pmc.get('SQ_WAVES') and 'SQ_INSTS_VALU' in pmc:
                # the bound that applies to these kernels: one wavefront per SIMD issues one instruction per ~4.5 cycles
                # whatever its ILP (tools/micro/valu_issue*.hip, profiles/r2_micro_valu_issue_*.txt); the kernel's time is
                # its longest wavefront's instruction stream at that rate.  Mean over ALL wavefronts of the launch here
                # (helper wavefronts that wait at barriers included), so 1.0 would mean every wavefront issues flat out.
                insts = sum(pmc.get(k, 0.0) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM')) / pmc['SQ_WAVES']
                cycles = kms * 1e-3 * CLOCK_HZ
                issue = {'bound': 'lone-wavefront instruction issue', 'instructions_per_wavefront_mean': insts, 'kernel_cycles': cycles,
                         'cycles_per_instruction_mean': cycles / insts if insts else None, 'lone_wavefront_limit_cycles_per_instruction': 4.5,
                         'frac': 4.5 * insts / cycles if cycles else None, 'wavefronts': pmc['SQ_WAVES']}
        out = {
            'metric': 'env steps/sec (whole node)', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s x %d envs per GPU' % (args.workload, B), 'what': cfg_desc, 'envs_total': total_envs,
                       'timestep': 1.0 / 240.0, 'substeps': env.layout.substeps, 'solver_iteration_cap': int(env.builder.solver_iterations),
                       'auto_reset': auto_reset, 'timed_path': 'backend entry points dg_world_step%s + dg_world_reset(term_flag); env.step() rates are in api_eager' % (' + dg_world_render' if cameras else ''),
                       'launch': 'hipGraph replay of %d-step segments' % R if graph is not None else 'eager',
                       'episodes_finished_rank0': episodes_main, 'parallelism': 'independent env shards x%d, no collective' % world,
                       'envs_per_wavefront': sim.lanes, 'lds_bytes_per_workgroup': sim.lds_bytes,
                       'parity': 'vs the C oracle only; parity with pybullet itself is UNPINNED (DESIGN.md 4)'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'kernel': kname, 'kernel_ms': kms, 'kernel_ms_source': ksource, 'kernel_ms_hip_events': kms_events,
                         'bytes_per_env_step': kbytes,
                         'step_kernel_ms': step_ms, 'step_kernel_ms_event_bracketed_eager_launch': step_eager_ms, 'render_kernel_ms': render_ms if cameras else None,
                         'survey_bytes_per_env_step': 449 if args.workload.startswith('ur_high_5') else None,
                         'limiter': ('NOT HBM either: VALU issue of the culling and intersection tests around 655 MB of image writes per launch (DESIGN.md 6)' if args.workload == 'from_the_readme' else
                                     'NOT HBM: instruction issue and latency of one wavefront per SIMD; the hbm fraction is reported because the contract asks for it'),
                         'valu_issue_frac_2cyc': valu_frac, 'wave_wait_frac': wait_frac, 'issue': issue,
                         'pmc_source': ('rocprofv3 child runs of this command, this invocation (%.0f s)' % pmc['seconds']) if pmc else None},
            'solver': solver, 'aged': aged, 'api_eager': api,
        }
        if render_bound and args.workload != 'from_the_readme':
            out['roofline']['note'] = 'camera render takes %.2f ms of the step' % render_ms
        if world == 1 and not args.no_cpu_baseline and not args.inner:
            out['cpu_baseline'] = cpu_baseline(cfg, lo.numel(), lo, hi)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=304)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--workload', default='ur_high_5', choices=sorted(WORKLOADS))
    ap.add_argument('--envs-per-gpu', type=int, default=None, help='default: the size BASELINE.json quotes for the workload')
    ap.add_argument('--age-steps', type=int, default=4000, help='untimed steps before the second (aged) timed segment; 0 disables it')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-auto-reset', action='store_true')
    ap.add_argument('--no-api', action='store_true', help='skip the eager env.step() API measurements')
    ap.add_argument('--eager', action='store_true', help='time the eager launch loop instead of a replayed hipGraph')
    ap.add_argument('--pmc', action='store_true', help='(default at N=1; kept for old command lines)')
    ap.add_argument('--no-pmc', action='store_true', help='N=1: skip the rocprofv3 child runs (kernel trace + HBM / SQ counter passes) that fill roofline.traffic / kernel_ms')
    ap.add_argument('--pmc-budget', type=int, default=360, help='seconds the rocprofv3 child runs may take in total')
    ap.add_argument('--inner', action='store_true', help=argparse.SUPPRESS)  # the profiled child of --pmc: timed loop only
    ap.add_argument('--master-port', type=int, default=0)
    ap.add_argument('--selftest-launcher', action='store_true', help='CPU-only rendezvous test of the N-rank launcher (gloo)')
    args = ap.parse_args()
    argv = sys.argv[1:]
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    world_env = os.environ.get('WORLD_SIZE')
    if world_env is None:
        if args.gpus > 1:
            sys.exit(launch(args, argv))  # this process stays GPU-free
    elif int(world_env) != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%s; refusing to report a different GPU count than asked' % (args.gpus, world_env))
    if args.selftest_launcher:
        if world_env is None:
            os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))
        return selftest_rank(args)
    run_rank(args, argv)


if __name__ == '__main__':
    main()
