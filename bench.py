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
    'ur_high_5': ('examples/ur_high_5/ur_high_5.yaml', "ur_high_5.yaml, semantically equal to the reference's (yaml.load gives the same tree; key order and layout differ): 2x UR5, ik_controller(use_orientation) + joint_state_sensor + object_state_sensor + reach_target"),
    'ur_high_5_joint': ('examples/ur_high_5/ur_high_5_joint.yaml', 'VARIANT of ur_high_5 with joint_controller(position) instead of ik_controller'),
    'drone_pilot': ('examples/drone_pilot/drone_pilot.yaml', "drone_pilot.yaml, semantically equal to the reference's: quadrotor + 4 propellor + fell_over + reach_target"),
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


def pybullet_baseline(seconds=10.0):
    """SURVEY 8(d)'s opportunistic leg: when the pybullet module happens to be importable on the box, a loop WRITTEN HERE that
    issues the call sequence of one reference env-step of ur_high_5 (diy_gym/diy_gym.py:187-209 with the addons of
    examples/ur_high_5/ur_high_5.yaml: per arm getLinkState -> calculateInverseKinematics with the null-space lists ->
    setJointMotorControlArray(POSITION_CONTROL); stepSimulation; per arm getJointStates; reach_target's getLinkState pairs for
    reward and terminal) on one core, from this repo's own URDF files.  Nothing of the reference is imported.  pybullet is
    absent from this image, so this function cannot be exercised here: any failure is reported as a string, never raised."""
    try:
        import pybullet as p
    except Exception:
        return {'status': 'pybullet unavailable on this box'}
    try:
        import numpy as np
        cid = p.connect(p.DIRECT)
        p.resetSimulation(physicsClientId=cid)
        p.setPhysicsEngineParameter(fixedTimeStep=1.0 / 240.0, numSolverIterations=150, numSubSteps=2, physicsClientId=cid)
        p.setGravity(0.0, 0.0, -9.81, physicsClientId=cid)
        urdf = os.path.join(ROOT, 'diy_gym_amd', 'data', 'ur5', 'ur5_robot.urdf')
        rest = [-0.17, -0.73, -1.93, -0.36, -0.03, -0.06]
        arms = []
        for xyz, yaw in (([-0.55, 0.4, 0.0], -1.57), ([0.55, 0.4, 0.0], 1.57)):
            uid = p.loadURDF(urdf, physicsClientId=cid)
            p.resetBasePositionAndOrientation(uid, xyz, p.getQuaternionFromEuler([0.0, 0.0, yaw]), physicsClientId=cid)
            info = [p.getJointInfo(uid, i, physicsClientId=cid) for i in range(p.getNumJoints(uid, physicsClientId=cid))]
            ee = [i[0] for i in info if i[1].decode() == 'ee_fixed_joint'][0]
            joints = [i for i in info if i[3] > -1 and i[0] <= ee]
            ids = [i[0] for i in joints]
            for j, q in zip(ids, rest):
                p.resetJointState(uid, j, q, physicsClientId=cid)
            arms.append(dict(uid=uid, ee=ee, ids=ids, lo=[i[8] for i in joints], hi=[i[9] for i in joints], rng=[i[9] - i[8] for i in joints],
                             force=[i[10] for i in joints]))
        rng = np.random.RandomState(0)
        steps, t0 = 0, time.time()
        while time.time() - t0 < seconds:
            for a in arms:
                ls = p.getLinkState(a['uid'], a['ee'], physicsClientId=cid)
                pos = [c + d for c, d in zip(ls[0], rng.uniform(-0.01, 0.01, 3))]
                orn = p.multiplyTransforms([0, 0, 0], ls[1], [0, 0, 0], p.getQuaternionFromEuler(list(rng.uniform(-0.01, 0.01, 3))))[1]
                q = p.calculateInverseKinematics(a['uid'], a['ee'], pos, orn, lowerLimits=a['lo'], upperLimits=a['hi'], jointRanges=a['rng'], restPoses=rest, physicsClientId=cid)
                p.setJointMotorControlArray(a['uid'], a['ids'], p.POSITION_CONTROL, targetPositions=list(q)[:len(a['ids'])], positionGains=[0.015] * len(a['ids']),
                                            velocityGains=[1.0] * len(a['ids']), forces=a['force'], physicsClientId=cid)
            p.stepSimulation(physicsClientId=cid)
            for a in arms:
                p.getJointStates(a['uid'], a['ids'], physicsClientId=cid)
            for _ in range(3):  # reach_target: reward + terminal, object_state_sensor
                p.getLinkState(arms[0]['uid'], arms[0]['ee'], computeLinkVelocity=1, physicsClientId=cid); p.getLinkState(arms[1]['uid'], arms[1]['ee'], computeLinkVelocity=1, physicsClientId=cid)
            steps += 1
        dt = time.time() - t0
        p.disconnect(cid)
        return {'status': 'timed', 'value': steps / dt, 'unit': 'env-steps/s', 'cores': 1, 'steps': steps, 'seconds': round(dt, 2),
                'what': 'build-written single-env loop over the pybullet module: the call sequence of one ur_high_5 reference step'}
    except Exception as exc:  # pragma: no cover
        return {'status': 'pybullet importable but the loop failed: %r' % (exc, )}


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
    pyb = pybullet_baseline()
    return {'value': head['value'], 'unit': 'env-steps/s', 'cores': cores if head is not f32_1 else 1, 'kind': 'port', 'nproc': os.cpu_count(),
            'fp32_1_core': f32_1, 'fp32_all_cores': f32_n, 'fp64_all_cores': f64_n, 'pybullet': pyb,
            'sample': 'same config, random actions, ~%.0f s each: C oracle built as fp32 on 1 core (64 envs), fp32 with OpenMP over envs on %d cores (%d envs) '
                      '[= value], fp64 on %d cores; pybullet itself: %s' % (seconds, cores, 64 * cores, cores, pyb['status'])}


def quantiles(t):
    """mean / p50 / p99 / max of an integer tensor, as plain floats."""
    f = t.float().flatten()
    return {'mean': round(float(f.mean()), 2), 'p50': float(f.median()), 'p99': float(f.quantile(0.99)) if f.numel() < (1 << 24) else None,
            'max': float(f.max())}


# ----------------------------------------------------------------------------------------------- legs
# A LEG = one world + one action ring + the loop the timed region runs on it.  The headline line is the leg `main`; the
# default N = 1 run adds short legs for the other tail of the headline scene and for the other BASELINE configs:
#   in_contact   ur_high_5, every env started from the crossed-forearms pose of tests/golden/ur_arms_touching_ik.yaml
#   mixed        ur_high_5, every 100th env started from that pose (one touching env per 1.6 wavefronts)
#   in_contact_capsule_contacts   in_contact with hull_contacts = 0 (hulls collide through their fitted capsules, as in rounds 1-3)
#   reference_solver_settings   ur_high_5 with motor_guess = 0, warmstart = 0.85: zero-started sweeps as the reference runs them [R]
#   configs.*    r2d2_maze x 4 096, drone_pilot x 16 384, from_the_readme x 1 024 (BASELINE.json cfg 2, 4, 5)
CROSSED = [1.35, -1.08, 1.03, -0.01, 0.09, 0.86]   # rest_position of tests/golden/ur_arms_touching_ik.yaml
REFERENCE_SETTINGS = {'motor_guess': 0.0, 'warmstart': 0.85}
LEGS = {
    # name: (workload, envs or None (BASELINE size), engine overrides, fraction of envs put into the crossed pose, steps timed)
    'in_contact': ('ur_high_5', None, None, 1.0, 200),
    'mixed': ('ur_high_5', None, None, 0.01, 200),
    # the same start with the narrow phase of rounds 1-3 (the capsule fitted to each hull instead of GJK / EPA on the hulls): what the
    # hull-hull contacts of round 4 cost in this regime
    'in_contact_capsule_contacts': ('ur_high_5', None, {'hull_contacts': 0.0}, 1.0, 200),
    'reference_solver_settings': ('ur_high_5', None, REFERENCE_SETTINGS, 0.0, 304),
    'r2d2_maze': ('r2d2_maze', None, None, 0.0, 200),
    'drone_pilot': ('drone_pilot', None, None, 0.0, 304),
    'from_the_readme': ('from_the_readme', None, None, 0.0, 104),
    # not part of 'all' (no BASELINE config): the arm + gripper tree of SURVEY 8f N2, `--legs ur5_child_gripper`
    'ur5_child_gripper': ('ur5_child_gripper', None, None, 0.0, 200),
}
DEFAULT_LEGS = ('in_contact', 'mixed', 'in_contact_capsule_contacts', 'reference_solver_settings', 'r2d2_maze', 'drone_pilot', 'from_the_readme')
CONFIG_LEGS = ('r2d2_maze', 'drone_pilot', 'from_the_readme')
KERNEL_OF = {'from_the_readme': 'render_kernel'}   # dominant kernel quoted in a leg's roofline (default: the step kernel)


class Leg:
    R = 8  # action batches in the ring = steps per replayed graph segment
    WHOLE_MAX = 96  # a timed region of at most this many steps is captured as one graph

    def __init__(self, name, workload, device, rank=0, envs=None, engine=None, crossed_frac=0.0, auto_reset=True):
        import torch
        import diy_gym_amd.examples  # noqa: F401
        from diy_gym_amd import DIYGym
        self.torch, self.name, self.workload, self.device = torch, name, workload, device
        cfg_rel, self.desc = WORKLOADS[workload]
        self.cfg = os.path.join(ROOT, cfg_rel)
        self.B = B = envs or DEFAULT_ENVS.get(workload, 16384)
        self.engine = dict(engine) if engine else None
        self.env = env = DIYGym(self.cfg, num_envs=B, device=device, seed=1234, env_index_base=rank * B, engine=self.engine)
        self.lo, self.hi = action_bounds(env)
        gen = torch.Generator().manual_seed(1234 + rank)
        self.ring = [(self.lo + (self.hi - self.lo) * torch.rand((B, self.lo.numel()), generator=gen)).to(device) for _ in range(self.R)]
        self.sim, self.slots, self.auto_reset = env.sim, env._all_slots, auto_reset
        self.cameras = [a for r in env.receptors.values() for a in r.addons.values() if hasattr(a, 'camera_index')]
        self.image_bytes = 0
        for cam in self.cameras:  # allocate the image buffers once; the timed step renders into them
            cam.observe()
            self.image_bytes += sum(t.numel() * t.element_size() for t in cam._buffers if t is not None) // B
        self.crossed_frac = crossed_frac
        if crossed_frac > 0.0:  # joint angles of the chosen envs = the crossed-forearms pose, at rest
            from diy_gym_amd.scene import K
            every = max(1, int(round(1.0 / crossed_frac)))
            idx = torch.arange(0, B, every, device=device)
            L = env.layout
            for arm in range(2):
                for j, q in enumerate(CROSSED):
                    o = L.link_state_off[6 * arm + j]
                    self.sim.state[o + K.LS_Q, idx] = q
                    self.sim.state[o + K.LS_QD, idx] = 0.0
                    self.sim.state[o + K.LS_TARGET_POS, idx] = q
            self.crossed_envs = int(idx.numel())
        self.graph = self.graph_rest = self.graph_kernel = self.graph_whole = None; self.graph_whole_steps = 0
        self.kernel_name = KERNEL_OF.get(workload, 'step_kernel_par' if getattr(self.sim, 'par', False) else 'step_kernel')
        self.step_launches = 0   # step-kernel launches so far (eager or replayed): the profiled child's manifest counts them

    def one_step(self, i):
        self.sim.step(self.slots, self.ring[i % self.R])
        for cam in self.cameras:
            self.sim.render(cam.camera_index, *cam._buffers)
        if self.auto_reset:
            self.sim.reset(self.sim.term_flag)

    def eager(self, n, start=0):
        for i in range(start, start + n):
            self.one_step(i)
        self.step_launches += n

    def warm_for(self, seconds, min_steps=48, max_steps=4000):
        """Untimed eager steps until the GPU has been busy for `seconds` (at least min_steps): a leg starts right after its world
        was built on the host, the GPU idle for a second or two, and with 0.1 ms kernels a 48-step warm-up is over before the
        clocks are back up -- the first run of these legs timed drone_pilot and the zero-started ur_high_5 at 3-4x their kernel
        time while the legs with long kernels were consistent."""
        torch, n, t0 = self.torch, 0, time.perf_counter()
        while n < min_steps or (time.perf_counter() - t0 < seconds and n < max_steps):
            self.eager(16, n); n += 16
            torch.cuda.synchronize()
        return n

    def capture(self, steps):
        """hipGraphs of the timed loop: segments of R consecutive steps (+ a shorter one for steps % R), and the step kernel
        alone for kernel_times().  Every graph gets ONE UNTIMED replay here, whatever --warmup says: the first launch of a
        freshly instantiated hipGraph can pay a one-off upload of tens of milliseconds (profiles/, round 3)."""
        torch, R = self.torch, self.R

        def cap(n_steps, body):
            st = torch.cuda.Stream(device=self.device)
            st.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(st):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for i in range(n_steps):
                        body(i)
            torch.cuda.current_stream(self.device).wait_stream(st)
            torch.cuda.synchronize()
            return g

        try:
            self.graph = cap(R, self.one_step)
            if steps % R:
                self.graph_rest = cap(steps % R, self.one_step)
            self.graph_kernel = cap(R, lambda i: self.sim.step(self.slots, self.ring[i % R]))
            # a SHORT timed region (the driver's --steps 20 is 2 ms) as ONE graph: every replay starts ~25 us after its launch, and
            # three of them (8 + 8 + 4 steps) were 3 % of such a region -- 0.1004 ms per step where 304 steps give 0.0972
            if R < steps <= self.WHOLE_MAX:
                self.graph_whole = cap(steps, self.one_step); self.graph_whole_steps = steps
        except Exception as exc:  # pragma: no cover
            print('graph capture failed (%s); timing the eager loop' % exc, file=sys.stderr)
            self.graph = self.graph_rest = self.graph_kernel = self.graph_whole = None
            return 0
        n = 0
        if self.graph_whole is not None:
            self.graph_whole.replay(); n += steps
        self.graph.replay(); n += R
        if self.graph_rest is not None:
            self.graph_rest.replay(); n += steps % R
        self.graph_kernel.replay(); n += R
        torch.cuda.synchronize()
        self.step_launches += n
        return n

    def run(self, steps):
        """EXACTLY `steps` steps, no synchronisation (the caller brackets it)."""
        R = self.R
        if self.graph_whole is not None and steps == self.graph_whole_steps:
            self.graph_whole.replay()
        elif self.graph is not None:
            for _ in range(steps // R):
                self.graph.replay()
            if steps % R:
                self.graph_rest.replay()
        else:
            for i in range(steps):
                self.one_step(i)
        self.step_launches += steps

    def timed(self, steps):
        torch = self.torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.run(steps)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def kernel_times(self, n):
        """Average duration of the dominant kernel(s) by HIP events on the launch stream (torch's current stream IS the
        stream the C-ABI launches on), same inputs, right after the timed region.  The step kernel: events around
        replays of a graph of R back-to-back step launches (no host launch gaps inside); eager launches bracketed one by
        one -- launch overhead included -- are reported next to it.  The render kernel: eager."""
        import numpy as np
        torch, R, sim = self.torch, self.R, self.sim
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
        for i in range(n):
            ev[i][0].record()
            sim.step(self.slots, self.ring[i % R])
            ev[i][1].record()
            for cam in self.cameras:
                sim.render(cam.camera_index, *cam._buffers)
            ev[i][2].record()
            if self.auto_reset:
                sim.reset(sim.term_flag)
        torch.cuda.synchronize()
        self.step_launches += n
        eager = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        render = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        replayed = None
        if self.graph_kernel is not None:
            reps = max(1, n // R)
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(reps):
                self.graph_kernel.replay()
            g1.record()
            torch.cuda.synchronize()
            self.step_launches += reps * R
            replayed = g0.elapsed_time(g1) / (reps * R)
            if self.auto_reset:
                sim.reset(sim.term_flag)
        return (replayed if replayed is not None else eager), render, eager

    def solver_stats(self, n=16):
        """live Gauss-Seidel / IK statistics from the kernel's diagnostics buffer (a separate, untimed segment: the
        production launches carry no diagnostics)"""
        torch, sim, env, B = self.torch, self.sim, self.env, self.B
        d = sim.enable_diagnostics()
        acc = []
        for i in range(n):
            self.one_step(i)
            acc.append(d.clone())
        torch.cuda.synchronize()
        sim.disable_diagnostics()
        self.step_launches += n
        D = torch.stack(acc)  # [n, B, 8]
        n_ik = sum(1 for r in env.receptors.values() for a in r.addons.values() if type(a).__name__ == 'InverseKinematicsController')
        per_wave = D[:, :, sim.DIAG_PGS_ITERS].reshape(n, -1, min(sim.envs_per_wave, B)).max(2).values if B % sim.envs_per_wave == 0 else None
        return {'iteration_cap': int(env.builder.solver_iterations), 'substeps': env.layout.substeps,
                'pgs_iterations_last_substep': quantiles(D[:, :, sim.DIAG_PGS_ITERS]),
                'pgs_iterations_first_substep': quantiles(D[:, :, sim.DIAG_PGS_ITERS_FIRST]),
                'pgs_iterations_wavefront_max': quantiles(per_wave) if per_wave is not None else None,
                'contacts_per_env': quantiles(D[:, :, sim.DIAG_CONTACTS]),
                'envs_with_contacts_frac': float((D[:, :, sim.DIAG_CONTACTS] > 0).float().mean()),
                'ik_iteration_cap': int(env.builder.params['ik_iterations']) if n_ik else None,
                'ik_iterations': quantiles(D[:, :, sim.DIAG_IK_ITERS:sim.DIAG_IK_ITERS + min(n_ik, sim.DIAG_N_IK)]) if n_ik else None,
                'sample': '%d steps after the timed region, every env' % n}

    def bytes_per_env_step(self):
        return self.image_bytes if self.kernel_name == 'render_kernel' else algorithmic_bytes_per_env_step(self.env.layout)

    def close(self):
        self.graph = self.graph_rest = self.graph_kernel = self.graph_whole = None
        self.env.close()


def make_leg(name, device, rank=0, envs=None, auto_reset=True):
    wl, n, engine, frac, _ = LEGS[name]
    return Leg(name, wl, device, rank, envs or n, engine, frac, auto_reset)


def leg_names(args):
    if args.legs == 'none' or args.gpus > 1 or args.workload != 'ur_high_5' or args.envs_per_gpu not in (None, 16384):
        return []   # the extra legs belong to the default headline run on one GPU
    return list(DEFAULT_LEGS) if args.legs == 'all' else [n for n in args.legs.split(',') if n in LEGS]


# ----------------------------------------------------------------------------------------------- profiled child
INNER_WARM, INNER_STEPS = 24, 40


def inner_run(args):
    """The command rocprofv3 profiles (``bench.py ... --inner --manifest PATH``): the eager loop of the main workload and of
    every extra leg, one after the other in ONE process, nothing else.  The manifest lists the legs in launch order with
    the number of step / render launches each made, so that the parent can cut the trace (or the counter rows) of this
    process into legs by dispatch order -- the last INNER_STEPS launches of each leg are its sample."""
    import torch
    torch.cuda.set_device(0)
    device = torch.device('cuda', 0)
    manifest = []
    names = ['main'] + leg_names(args)
    for name in names:
        leg = Leg('main', args.workload, device, 0, args.envs_per_gpu, None, 0.0, not args.no_auto_reset) if name == 'main' else make_leg(name, device, auto_reset=not args.no_auto_reset)
        leg.eager(INNER_WARM)
        torch.cuda.synchronize()
        leg.eager(INNER_STEPS, INNER_WARM)
        torch.cuda.synchronize()
        manifest.append({'leg': name, 'step_launches': leg.step_launches, 'render_launches': leg.step_launches * len(leg.cameras) + len(leg.cameras), 'sample': INNER_STEPS,
                         'kernel': leg.kernel_name})
        leg.close()
        del leg
    if args.manifest:
        json.dump(manifest, open(args.manifest, 'w'))


def profile_children(args, argv):
    """N = 1, default on (``--no-pmc`` opts out): per-launch figures of every leg's dominant kernel measured NOW by running
    this very command's eager loops (inner_run) under rocprofv3 in fresh child processes -- started before this process
    touches the GPU, ``python bench.py ...`` directly after ``--`` -- one pass each: a kernel trace (durations), then the HBM
    counters and the SQ counters in their own passes (never a trace and counters together).  Corrections per
    MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB, and gfx950's FETCH_SIZE counts half of a coalesced read.
    The rows of a pass are cut into legs by dispatch order with the child's manifest.  Bounded: a pass that fails or runs
    out of time leaves its fields null."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None
    out, t_begin = {}, time.time()
    skip = ('--pmc', '--no-pmc')
    base = [a for a in argv if a not in skip]
    env = dict(os.environ, TMPDIR=os.environ.get('TMPDIR', '/tmp'))
    passes = [('trace', ['--kernel-trace']), ('fetch', ['--pmc', 'FETCH_SIZE']), ('write', ['--pmc', 'WRITE_SIZE']),
              ('sq', ['--pmc', 'SQ_INSTS_VALU', 'SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_BUSY_CYCLES', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM'])]

    def is_step(n):
        return 'step_kernel' in n and 'reset' not in n

    for pname, flags in passes:
        left = args.pmc_budget - (time.time() - t_begin)
        if left < 30:
            print('bench.py: profiling budget (%d s) spent before the %s pass; its fields stay null' % (args.pmc_budget, pname), file=sys.stderr)
            break
        d = tempfile.mkdtemp(prefix='dg_prof_', dir=env['TMPDIR'])
        man = os.path.join(d, 'manifest.json')
        cmd = ['rocprofv3'] + flags + ['--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__)] + base + ['--inner', '--manifest', man]
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=left, check=True, cwd=env['TMPDIR'], env=env)
            manifest = json.load(open(man))
        except Exception as exc:
            print('bench.py: rocprofv3 %s pass failed (%s); its fields stay null' % (pname, type(exc).__name__), file=sys.stderr)
            shutil.rmtree(d, ignore_errors=True)
            continue
        # dispatches in launch order: (kernel name, value dict)
        rows = []
        if pname == 'trace':
            for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    rows.append((int(r['Start_Timestamp']), r['Kernel_Name'], {'ns': float(int(r['End_Timestamp']) - int(r['Start_Timestamp']))}))
        else:
            per = {}
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = int(r['Dispatch_Id'])
                    per.setdefault(k, [r['Kernel_Name'], {}])[1][r['Counter_Name']] = per.get(k, [None, {}])[1].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            rows = [(k, v[0], v[1]) for k, v in per.items()]
        rows.sort(key=lambda t: t[0])
        steps = [(n, v) for _, n, v in rows if is_step(n)]
        renders = [(n, v) for _, n, v in rows if 'render_kernel' in n]
        so = ro = 0
        for m in manifest:
            mine_s, mine_r = steps[so:so + m['step_launches']], renders[ro:ro + m['render_launches']]
            so += m['step_launches']; ro += m['render_launches']
            sample = (mine_r if m['kernel'] == 'render_kernel' else mine_s)[-m['sample']:]
            if len(sample) < m['sample']:
                continue  # (the trace does not hold what the manifest says: leave the leg's fields null)
            rec = out.setdefault(m['leg'], {})
            rec['kernel'] = sample[-1][0].split('(')[0].replace('void ', '')
            if pname == 'trace':
                v = [s_[1]['ns'] for s_ in sample]
                rec['trace'] = {'calls': len(v), 'average_ns': sum(v) / len(v), 'min_ns': min(v), 'max_ns': max(v)}
                if m['kernel'] == 'render_kernel' and len(mine_s) >= m['sample']:
                    w = [s_[1]['ns'] for s_ in mine_s[-m['sample']:]]
                    rec['step_trace'] = {'kernel': mine_s[-1][0].split('(')[0].replace('void ', ''), 'calls': len(w), 'average_ns': sum(w) / len(w)}
            else:
                for cname in sample[0][1]:
                    rec[cname] = sum(s_[1].get(cname, 0.0) for s_ in sample) / len(sample)
        if so != len(steps):
            print('bench.py: the %s pass holds %d step launches, the manifest %d' % (pname, len(steps), so), file=sys.stderr)
        shutil.rmtree(d, ignore_errors=True)
    out['seconds'] = round(time.time() - t_begin, 1)
    return out


def roofline_of(leg, kms_events, step_ms, render_ms, step_eager_ms, prof):
    """The `roofline` object of one leg: its dominant kernel's average launch duration (from the rocprofv3 kernel trace of the
    child run when there is one -- it cannot include launch gaps -- else from the HIP events), algorithmic bytes per launch over
    that, and the counters of the child passes."""
    B, kname, kbytes = leg.B, leg.kernel_name, leg.bytes_per_env_step()
    kms, ksource = kms_events, 'HIP events around %s' % ('graph replays of back-to-back launches' if (leg.graph_kernel is not None and kname != 'render_kernel') else 'eager launches')
    trace = prof.get('trace') if prof else None
    if trace:
        kms, ksource = trace['average_ns'] * 1e-6, 'rocprofv3 --kernel-trace of a child run of this command (%d launches)' % trace['calls']
    achieved = kbytes * B / (kms * 1e-3) / 1e9
    traffic = valu_frac = wait_frac = issue = None
    if prof:
        if 'FETCH_SIZE' in prof and 'WRITE_SIZE' in prof:
            traffic = (2.0 * prof['FETCH_SIZE'] + prof['WRITE_SIZE']) * 1024.0
        if 'SQ_INSTS_VALU' in prof:  # one VALU wave-instruction occupies a SIMD-32 for 2 cycles (MI355X_MICROARCH.md)
            valu_frac = prof['SQ_INSTS_VALU'] * 2.0 / (SIMDS * kms * 1e-3 * CLOCK_HZ)
        if prof.get('SQ_WAVE_CYCLES'):
            wait_frac = prof.get('SQ_WAIT_ANY', 0.0) / prof['SQ_WAVE_CYCLES']
        if prof.get('SQ_WAVES') and 'SQ_INSTS_VALU' in prof:
            # the bound that applies to these kernels: one wavefront per SIMD issues one instruction per ~4.5 cycles
            # whatever its ILP (tools/micro/valu_issue*.hip, profiles/r2_micro_valu_issue_*.txt); the kernel's time is
            # its longest wavefront's instruction stream at that rate.  Mean over ALL wavefronts of the launch here
            # (helper wavefronts that wait at barriers included), so 1.0 would mean every wavefront issues flat out.
            insts = sum(prof.get(k, 0.0) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM')) / prof['SQ_WAVES']
            cycles = kms * 1e-3 * CLOCK_HZ
            issue = {'bound': 'lone-wavefront instruction issue', 'instructions_per_wavefront_mean': insts, 'kernel_cycles': cycles,
                     'cycles_per_instruction_mean': cycles / insts if insts else None, 'lone_wavefront_limit_cycles_per_instruction': 4.5,
                     'frac': 4.5 * insts / cycles if cycles else None, 'wavefronts': prof['SQ_WAVES']}
    r = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
         'kernel': (prof or {}).get('kernel', kname), 'kernel_ms': kms, 'kernel_ms_source': ksource, 'kernel_ms_hip_events': kms_events, 'bytes_per_env_step': kbytes,
         'step_kernel_ms': step_ms, 'step_kernel_ms_event_bracketed_eager_launch': step_eager_ms, 'render_kernel_ms': render_ms if leg.cameras else None,
         'valu_issue_frac_2cyc': valu_frac, 'wave_wait_frac': wait_frac, 'issue': issue}
    if prof and prof.get('step_trace'):
        r['step_kernel_trace'] = prof['step_trace']
    return r


def measure_leg(name, device, args, prof):
    """One extra leg, start to finish: world, warm-up, graphs (first replay untimed), timed replay, kernel events, solver."""
    import torch
    steps = LEGS[name][4]
    leg = make_leg(name, device, auto_reset=not args.no_auto_reset)
    untimed = leg.warm_for(0.25)
    untimed += leg.capture(steps)
    leg.run(2 * leg.R); untimed += 2 * leg.R   # (two more untimed segments through the replayed graph)
    # the kernel's HIP-event time is taken on BOTH sides of the timed region and averaged: a rollout ages (drones land, arms press
    # harder), and a kernel time from after the region alone can exceed the region's own step time (round 3's ur5_child_gripper line)
    k_before = leg.kernel_times(16); untimed += 16 + (16 if leg.graph_kernel is not None else 0)
    el = leg.timed(steps)
    k_after = leg.kernel_times(16)
    step_ms, render_ms, step_eager_ms = [0.5 * (a + b) for a, b in zip(k_before, k_after)]
    out = {'workload': '%s x %d envs' % (leg.workload, leg.B), 'what': leg.desc, 'engine': leg.engine, 'steps': steps, 'untimed_steps_before': untimed,
           'ms_per_step': el / steps * 1e3, 'value': leg.B * steps / el, 'unit': 'env-steps/s',
           'envs_per_wavefront': leg.sim.lanes, 'lds_bytes_per_workgroup': leg.sim.lds_bytes,
           'roofline': roofline_of(leg, render_ms if leg.kernel_name == 'render_kernel' else step_ms, step_ms, render_ms, step_eager_ms, prof),
           'solver': leg.solver_stats()}
    out['roofline']['step_kernel_ms_before_and_after_the_timed_region'] = [k_before[0], k_after[0]]
    if leg.crossed_frac > 0:
        out['started_in_the_crossed_forearms_pose'] = {'envs': leg.crossed_envs, 'of': leg.B}
    leg.close()
    return out


# ----------------------------------------------------------------------------------------------- one rank
def run_rank(args, argv):
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    distributed = world > 1

    pmc = None
    if not args.no_pmc and not distributed:
        pmc = profile_children(args, argv)  # child processes; this process has not touched the GPU yet

    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group(backend='nccl', device_id=device)

    from diy_gym_amd import DIYGym
    from diy_gym_amd.config import Configuration
    auto_reset = not args.no_auto_reset
    main = Leg('main', args.workload, device, rank, args.envs_per_gpu, None, 0.0, auto_reset)
    env, sim, B, R, ring, cfg = main.env, main.sim, main.B, main.R, main.ring, main.cfg

    # Warm-up: --warmup untimed eager steps, then -- ALWAYS, whatever --warmup says -- one untimed replay of every graph the
    # timed region uses (Leg.capture); the number of untimed steps actually run is reported.
    main.eager(args.warmup)
    torch.cuda.synchronize()
    untimed = args.warmup
    if not args.eager:
        untimed += main.capture(args.steps)

    def timed(steps):
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        main.run(steps)
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
    step_ms, render_ms, step_eager_ms = main.kernel_times(max(R, min(64, args.steps)))
    solver = main.solver_stats() if rank == 0 else None

    # (measured BEFORE the aged segment, so that the rates are those of the API on the same young rollout as `value`)
    # eager public-API rates: env.step() with the reference's dict actions, and with flatten_actions /
    # flatten_observations + collapsed reward / terminal (the trainer-facing fast path); auto-reset as above
    api = None
    if rank == 0 and not args.no_api:
        from diy_gym_amd.utils import unflatten
        n_api = 100
        dict_ring = [unflatten(r, env.action_space, batch_dims=1) for r in ring] if main.lo.numel() else [{} for _ in ring]
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
    if args.age_steps > 0:
        main.run((args.age_steps // R) * R)
        main.eager(args.age_steps % R)
        el = timed(args.steps)
        a_step_ms, a_render_ms, _ = main.kernel_times(max(R, min(32, args.steps)))
        aged = {'after_steps': untimed + args.steps + 80 + args.age_steps + (105 if api else 0), 'ms_per_step_aged': el / args.steps * 1e3,
                'ms_per_step_aged_over_young': (el / args.steps) / (elapsed / args.steps),
                'value_aged': B * world * args.steps / el, 'kernel_ms_aged': a_step_ms,
                'episodes_finished_rank0': float(sim.state[1, :B].sum().item()) - B,
                'solver': main.solver_stats() if rank == 0 else None}

    if rank == 0:
        total_envs = B * world
        value = total_envs * args.steps / elapsed
        kms_events = render_ms if main.kernel_name == 'render_kernel' else step_ms
        roof = roofline_of(main, kms_events, step_ms, render_ms, step_eager_ms, pmc.get('main') if pmc else None)
        roof['survey_bytes_per_env_step'] = 449 if args.workload.startswith('ur_high_5') else None
        roof['limiter'] = ('NOT HBM either: VALU issue of the culling and intersection tests around 655 MB of image writes per launch (DESIGN.md 6)' if args.workload == 'from_the_readme' else
                           'NOT HBM: instruction issue and latency of one wavefront per SIMD; the hbm fraction is reported because the contract asks for it')
        roof['pmc_source'] = ('rocprofv3 child runs of this command, this invocation (%.0f s)' % pmc['seconds']) if pmc else None
        if main.cameras and render_ms > step_ms * 0.2 and args.workload != 'from_the_readme':
            roof['note'] = 'camera render takes %.2f ms of the step' % render_ms
        out = {
            'metric': 'env steps/sec (whole node)', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s x %d envs per GPU' % (args.workload, B), 'what': main.desc, 'envs_total': total_envs,
                       'timestep': 1.0 / 240.0, 'substeps': env.layout.substeps, 'solver_iteration_cap': int(env.builder.solver_iterations),
                       'solver_start': {k: env.builder.params[k] for k in ('motor_guess', 'limit_guess', 'warmstart', 'warmstart_friction')},
                       'narrow_phase': {k: env.builder.params[k] for k in ('hull_contacts', 'hull_margin', 'contact_margin')},
                       'auto_reset': auto_reset, 'timed_path': 'backend entry points dg_world_step%s + dg_world_reset(term_flag); env.step() rates are in api_eager' % (' + dg_world_render' if main.cameras else ''),
                       'launch': ('one hipGraph of the %d timed steps' % args.steps if main.graph_whole is not None else 'hipGraph replay of %d-step segments' % R) if main.graph is not None else 'eager',
                       'untimed_steps_before_the_timed_region': untimed,
                       'episodes_finished_rank0': episodes_main, 'parallelism': 'independent env shards x%d, no collective' % world,
                       'envs_per_wavefront': sim.lanes, 'lds_bytes_per_workgroup': sim.lds_bytes,
                       'parity': 'vs the C oracle only; parity with pybullet itself is UNPINNED (DESIGN.md 4)'},
            'roofline': roof, 'solver': solver, 'aged': aged, 'api_eager': api,
        }
        main.close()
        # the other tail of the headline scene, the reference's solver settings, the other BASELINE configs (N = 1 default run)
        t_legs = time.time()
        for name in leg_names(args):
            try:
                rec = measure_leg(name, device, args, pmc.get(name) if pmc else None)
            except Exception as exc:  # pragma: no cover
                rec = {'error': repr(exc)}
            if name in CONFIG_LEGS or name not in DEFAULT_LEGS:
                out.setdefault('configs', {})[name] = rec
            else:
                out[name] = rec
                if 'ms_per_step' in rec:
                    rec['ms_per_step_over_young'] = rec['ms_per_step'] / out['ms_per_step']
        if leg_names(args):
            out['legs_seconds'] = round(time.time() - t_legs, 1)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, main.lo.numel(), main.lo, main.hi)
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
    ap.add_argument('--pmc-budget', type=int, default=420, help='seconds the rocprofv3 child runs may take in total')
    ap.add_argument('--inner', action='store_true', help=argparse.SUPPRESS)  # the profiled child: eager loops of every leg, nothing else
    ap.add_argument('--manifest', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--legs', default='all', help="extra legs of the default N=1 ur_high_5 run: 'all', 'none' or a comma-separated subset of " + ','.join(LEGS))
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
    if args.inner:
        return inner_run(args)
    run_rank(args, argv)


if __name__ == '__main__':
    main()
