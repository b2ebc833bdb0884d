#!/usr/bin/env python
"""bench.py -- depth-crops/sec of the hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no launcher: re-executes itself under torch.distributed.run with N ranks)

The number of ranks must equal --gpus: a mismatch is an error, never a silent one-GPU measurement.

A "step" is one pass of the hot path over one batch of 40 synthetic 128x128 crops per GPU, inputs
resident in HBM before the timed region:
  --mode train (default when the handle supports it): NYU S=2 F=128 J=14 B=40 train micro-step =
        forward (batch-stat BatchReNorm, dropout) + loss + backward; every `sub_batch`-th step also
        all-reduces the flat gradient over RCCL (N>1) and applies clip+Adam -- BASELINE.json config 3/4.
  --mode infer: ICVL S=2 F=128 J=16 B=40 forward(eval) + vote -> xyz -- BASELINE.json config 2.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed in a separate profiled pass over the same workload), `cpu_baseline` (the CPU
oracle restatement timed on the host cores, rank 0 / N=1 only) and, in train mode, `forward_vote`: the
other north-star figure (ICVL forward(eval)+vote crops/s, same N, same process) -- `metric`/`value`
stay the fwd+bwd headline.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from densereg_amd.data.synthetic import DATASETS, make_crops  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--mode', choices=['train', 'infer'], default=os.environ.get('DR_BENCH_MODE', 'train'))
    ap.add_argument('--batch', type=int, default=40)
    ap.add_argument('--sub_batch', type=int, default=5)
    # the headline workload is the default; the other BASELINE configs (e.g. config 5: --num_stack 4 --num_fea 256 --in_hw 256
    # --dataset nyu --precision bf16) are reachable for measurement and say so in `metric` / `config.workload`
    ap.add_argument('--num_stack', type=int, default=2)
    ap.add_argument('--num_fea', type=int, default=128)
    ap.add_argument('--in_hw', type=int, default=128, choices=[128, 256, 512])
    ap.add_argument('--dataset', default='', choices=['', 'icvl', 'nyu', 'msra'], help='default: nyu for train, icvl for infer')
    ap.add_argument('--precision', choices=['f32', 'bf16'], default='f32',
                    help='matrix-core arithmetic of the convolutions; bf16 = BASELINE config 5\'s conv path (fp32 stays the headline)')
    ap.add_argument('--groups', type=int, default=int(os.environ.get('DR_BENCH_GROUPS', '-1')),
                    help='train mode: micro-steps of an accumulation window run as ONE pass of launches (dr_set_groups). '
                         '-1 = auto (sub_batch where the engine supports it and the window fits), 1 = one micro-step per pass')
    ap.add_argument('--merge', type=int, default=int(os.environ.get('DR_BENCH_MERGE', '5')),
                    help='forward(eval)+vote: consecutive submitted batches run as one launch of this many batches (ReplicaPool merge)')
    ap.add_argument('--replicas', type=int, default=2,
                    help='forward(eval)+vote: inference replicas per GPU, consecutive batches alternate between them '
                         '(densereg_amd/serving.py; 1 = one engine, one stream)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-forward-vote', action='store_true', help='train mode: skip the forward(eval)+vote leg')
    ap.add_argument('--detail', default='', help='write a per-layer timing table (markdown) to this path')
    return ap.parse_args()


def cpu_baseline(mode, cfg_tuple, B, dataset):
    """The CPU oracle (PyTorch-CPU restatement of the reference graph, NOT TF1.3) on the host cores: the full B-crop step of
    the benchmarked workload (B=40).  forward(eval)+vote (~1-2 s per iteration): SURVEY's 3 warm-up + 10 timed
    iterations; the training step (~6 s per iteration): 1 warm-up + at least 3 timed, more while a ~35 s budget lasts.
    `value` is the MEDIAN; min / max and every sample are reported next to it (the figure wanders 10-25 % run to run on a
    shared host -- it is context, not a target)."""
    from oracle import net, pose, train
    from oracle.graph import NetConfig
    S, F, J = cfg_tuple
    cfg = NetConfig(S, F, J)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    dm, poses, cfgs, coms, _ = make_crops(B, dataset, seed=999)
    poses = np.ascontiguousarray(poses[:, :3 * J])
    ndm = pose.norm_dm(dm, coms)
    params = net.init_params(cfg, 7)

    def one():
        if mode == 'infer':
            ep = net.forward_eval(cfg, params, ndm)
            pose.estimate_pose_mm(ep['hm_outs'][-1], ep['hm3_outs'][-1], ep['um_outs'][-1], ndm, cfgs, coms)
        else:
            train.loss_and_grads(cfg, params, ndm, poses, cfgs, coms)

    # at most 32 threads: oneDNN stops scaling there, and on a many-core box whose cgroup grants fewer CPUs than it shows
    # an all-cores run thrashes (measured on the 256-thread MI355X host: one B=40 training step did not finish in minutes)
    ncores = max(1, min(avail, 32))
    torch.set_num_threads(ncores)
    n_warm, n_min, n_max, budget_s = (3, 10, 10, 60.0) if mode == 'infer' else (1, 3, 10, 35.0)
    t_start = time.time()
    warm = []
    for _ in range(n_warm):                                 # oneDNN primitive creation, allocator
        t0 = time.time()
        one()
        warm.append(time.time() - t0)
        if warm[-1] > 40.0:                                 # pathological host: stop warming, one timed sample below
            n_min = 1
            break
    times = []
    while len(times) < n_max and (len(times) < n_min or time.time() - t_start + min(warm + times) < budget_s):
        t0 = time.time()
        one()
        times.append(time.time() - t0)
        if times[-1] > 40.0:
            break
    med = float(np.median(times))
    return {'value': B / med, 'unit': 'crops/s', 'cores': ncores, 'kind': 'port',
            'min': B / max(times), 'max': B / min(times), 'seconds_per_iteration': [round(t, 3) for t in times],
            'sample': '%d timed iteration(s) after %d warm-up of the full B=%d %s step on the CPU oracle (PyTorch-CPU fp32, '
                      'oneDNN); value = median, min / max next to it; %d threads of %d visible cores (capped at 32: more does not '
                      'scale and thrashes a cgroup-limited host)'
                      % (len(times), len(warm), B, 'fwd(eval)+vote' if mode == 'infer' else 'fwd+loss+bwd', ncores, avail)}


def pmc_traffic(mode, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/rocpd_pmc.py --json; rocprofv3 cannot collect
    counters from inside this process) -- quoted only while the passes describe THIS build: the entry's stamp carries the hash
    of the kernel sources the passes ran with (densereg_amd/buildinfo.py; comments and whitespace do not count).  Returns
    (bytes or None, provenance dict)."""
    from densereg_amd.buildinfo import kernel_source_hash
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        m = json.load(open(path))[mode]
        e, stamp = m[kernel], m.get('_stamp', {})
    except Exception:
        return None, {'note': 'no PMC pass of this mode / kernel under profiles/pmc_traffic.json'}
    here = kernel_source_hash()
    prov = {'pmc_kernel_source_hash': stamp.get('kernel_source_hash'), 'this_build_kernel_source_hash': here,
            'pmc_git_head': stamp.get('git_head'), 'pmc_date': stamp.get('date'), 'pmc_kernels': stamp.get('kernels', {}).get(kernel)}
    if stamp.get('kernel_source_hash') != here:
        prov['note'] = 'kernel sources changed since the PMC passes (or the passes are unstamped): traffic not quoted'
        return None, prov
    return e['read_bytes_per_launch'] + e['write_bytes_per_launch'], prov


# DR_BENCH_DRYRUN=1: the N > 1 HOST logic of this file without a GPU -- spawn_ranks, the launcher's environment, rank -> device
# binding, the process group (gloo instead of RCCL), DataParallelTrainer's window step and its all-reduce of the flat gradient,
# barrier + MAX-over-ranks timing, the one JSON line of rank 0 -- around an engine stand-in that runs NO kernels and does no
# arithmetic (_DryRunEngine).  It exists so that the first real 8-GPU launch cannot die on host logic (tests/test_bench_dryrun.py);
# its line says "dry_run": true and measures nothing.
DRY_RUN = os.environ.get('DR_BENCH_DRYRUN') == '1'


class _DryRunEngine:
    """What bench.py and DataParallelTrainer call on an Engine, with no device and no numerics: the flat gradient is a CPU tensor that
    ``backward`` fills with rank + 1, so the all-reduce that follows can be CHECKED (every element = 1 + 2 + ... + world)."""
    pipeline = 1

    def __init__(self, rank, world, n_param=4096):
        self.rank, self.world = rank, world
        self.device = torch.device('cpu')
        self._grad = torch.zeros(n_param)
        self.checked_reductions = 0

    def set_precision(self, _p): pass
    def load_params(self, _params): pass
    def param_infos(self): return []
    def norm_dm(self, dm, _com): return dm
    def new(self, *shape): return torch.empty(*shape)
    def flat_view(self, which):
        assert which == 'grad'
        return self._grad
    def zero_grad(self): self._grad.zero_()
    def set_groups(self, _g): pass
    def groups_supported(self, _bg, _g): return True
    def forward_train(self, _dm, _mode, _mask, _seed): pass
    def loss(self, dm, _pose, _cfg, _com): return torch.zeros(4 * self._g(dm))
    def _g(self, dm): return max(1, dm.shape[0] // self.micro_batch)
    def backward(self, _b): self._grad.fill_(float(self.rank + 1))
    def apply_adam(self, _lr, div, _step, _clip):
        want = float(self.world * (self.world + 1) // 2)
        assert bool((self._grad == want).all()), 'dry run: the all-reduce left %r, expected %r on every element' % (float(self._grad[0]), want)
        assert div == float(self.sub_batch * self.world), div
        self.checked_reductions += 1
    def conv_flops_per_crop(self): return 0.0
    def close(self): pass


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: run N ranks under torch.distributed.run (one per GPU, RCCL)."""
    ngpu = args.gpus if DRY_RUN else torch.cuda.device_count()
    if ngpu < args.gpus:
        sys.stderr.write('bench.py: --gpus %d but only %d GPU(s) are visible; refusing to measure fewer ranks than asked\n'
                         % (args.gpus, ngpu))
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit('bench.py: --gpus must be >= 1')
    # dmabuf IPC (RCCL / cross-process device memory on this driver): also when an external launcher started the ranks without it --
    # the HIP runtime reads it at its first call, which is further down
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    # stdout carries exactly ONE line, the JSON: anything else written to file descriptor 1 during the run (RCCL prints a
    # version banner there from C when the first communicator is created) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree\n' % (args.gpus, world))
        sys.exit(2)
    # one rank per GPU; if the launcher narrows device visibility to one GPU per process, that GPU is index 0
    # (dry run: as many pretend devices as DR_BENCH_DRYRUN_DEVICES says -- 1 = the narrowed-visibility case -- default one per rank)
    ndev = int(os.environ.get('DR_BENCH_DRYRUN_DEVICES', str(args.gpus))) if DRY_RUN else torch.cuda.device_count()
    local = int(os.environ.get('LOCAL_RANK', '0')) % max(ndev, 1)
    if not DRY_RUN:
        torch.cuda.set_device(local)                    # before the process group: RCCL binds to the current device
    dist = None
    if world > 1 or os.environ.get('DR_FORCE_ALLREDUCE') == '1':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('gloo' if DRY_RUN else 'nccl', rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus or world == 1
    dev = torch.device('cpu') if DRY_RUN else torch.device('cuda', local)
    if DRY_RUN:
        if args.mode != 'train' or args.precision != 'f32':
            sys.exit('bench.py: the dry run covers the training path (the one with a collective)')
        args.no_profile = args.no_forward_vote = args.no_cpu_baseline = True
        sys.stderr.write('bench.py DRY RUN rank %d/%d: LOCAL_RANK=%s -> pretend device %d of %d, backend gloo, MASTER %s:%s\n' % (
            rank, world, os.environ.get('LOCAL_RANK', '0'), local, ndev, os.environ.get('MASTER_ADDR'), os.environ.get('MASTER_PORT')))

    from densereg_amd import _lib
    from densereg_amd.engine import Engine
    from densereg_amd.parallel import DataParallelTrainer

    mode = args.mode
    dataset = args.dataset or ('nyu' if mode == 'train' else 'icvl')
    J = DATASETS[dataset]['jnt_num']
    S, F, B, HW = args.num_stack, args.num_fea, args.batch, args.in_hw
    # training: the `sub_batch` micro-steps between two optimizer steps as micro-batch groups of one pass (dr_set_groups) where
    # the window fits comfortably: up to 262 144 pixels per full-resolution layer (5 x 40 crops at 32x32 maps: 204 800)
    G = 1
    if mode == 'train':
        from densereg_amd.parallel import window_groups
        try:
            G = window_groups(B, args.sub_batch, HW, args.groups)
        except ValueError as e:
            raise SystemExit('bench.py: %s' % e)
    MG = max(1, args.merge) if mode == 'infer' else 1       # forward(eval)+vote: batches per launch (ReplicaPool merge)
    if DRY_RUN:
        eng = _DryRunEngine(rank, world)
        eng.micro_batch, eng.sub_batch = B, args.sub_batch
    else:
        eng = Engine(S, F, J, HW, 3, B * max(G, MG), local, training=(mode == 'train'))
    bf16 = args.precision == 'bf16'
    if bf16:
        eng.set_precision('bf16')
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS

    # random-init weights of the named architecture (values are irrelevant to dense conv speed)
    def random_params(e):
        rng = np.random.default_rng(7)
        params = {}
        for name, shape, _ in e.param_infos():
            leaf = name.rsplit('/', 1)[1]
            if leaf == 'weights':
                params[name] = (rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))).astype(np.float32)
            elif leaf in ('gamma', 'moving_variance', 'r_max'):
                params[name] = np.ones(shape, np.float32)
            else:
                params[name] = np.zeros(shape, np.float32)
        return params
    eng.load_params(random_params(eng))

    dm, poses, cfgs, coms, _ = make_crops(B, dataset, seed=20240, rank=rank, hw=HW)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_dm_mm, d_pose, d_cfg, d_com = t(dm), t(poses), t(cfgs), t(coms)
    d_dm = eng.norm_dm(d_dm_mm, d_com)
    xyz = eng.new(B, 3 * J)
    trainer = DataParallelTrainer(eng, dataset=dataset, sub_batch=args.sub_batch, dist=dist) if mode == 'train' else None
    if G > 1:                                                # the G micro-batches of a window (different crops), resident like d_dm
        wdm, wposes, wcfgs, wcoms, _ = make_crops(B * G, dataset, seed=20240, rank=rank, hw=HW)
        w_pose, w_cfg, w_com = t(wposes), t(wcfgs), t(wcoms)
        w_dm = eng.norm_dm(t(wdm), w_com)
    if MG > 1:                                               # what a replica launches: MG batches side by side (the roofline leg times this)
        m_dm, m_cfg, m_com = (torch.cat([x] * MG) for x in (d_dm, d_cfg, d_com))
        m_xyz = eng.new(B * MG, 3 * J)
    pending = [0]                                            # micro-steps handed in since the last window / flush
    live_pools = []

    # forward(eval)+vote throughput: `--replicas` engines with the same weights, each on its own stream, take the batches in turn
    # (north-star: inference = replicas; the single-engine figure is reported next to it)
    def make_pool(Jn):
        from densereg_amd.serving import ReplicaPool
        pool = ReplicaPool(args.replicas, S, F, Jn, HW, 3, B, local, merge=args.merge)
        if bf16:
            pool.set_precision('bf16')
        pool.load_params(random_params(pool))
        return pool, [pool.engines[0].new(B, 3 * Jn) for _ in range(args.replicas * args.merge)]
    pool, pool_out = make_pool(J) if (mode == 'infer' and args.replicas * args.merge > 1) else (None, None)
    if pool is not None:
        live_pools.append(pool)

    def step(i):
        if mode == 'infer':
            if pool is not None:
                pool.submit(d_dm, d_cfg, d_com, out=pool_out[i % len(pool_out)])
            else:
                eng.infer(d_dm, d_cfg, d_com, out=xyz)
        elif G > 1:                                          # every G-th micro-step launches the window they form
            pending[0] += 1
            if pending[0] == G:
                trainer.window_step(w_dm, w_pose, w_cfg, w_com, seed=i)
                pending[0] = 0
        else:
            trainer.micro_step(d_dm, d_pose, d_cfg, d_com, seed=i)

    def flush():
        """micro-steps that do not fill a window (K or W not a multiple of G) run one by one: exactly K steps are timed"""
        for _ in range(pending[0]):
            trainer.micro_step(d_dm, d_pose, d_cfg, d_com, seed=0)
        pending[0] = 0
        for p in live_pools:                                   # a merged group still filling: launched, so exactly K batches are timed
            p.flush()

    def barrier():
        if not DRY_RUN:
            torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        if not DRY_RUN:
            torch.cuda.synchronize(dev)

    def timed(fn, steps=None, warmup=None):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize, MAX over ranks (seconds)."""
        steps = args.steps if steps is None else steps
        warmup = args.warmup if warmup is None else warmup
        for i in range(warmup):
            fn(i)
        flush()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(warmup + i)
        flush()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    dt = timed(step)
    single = None
    if pool is not None:                                     # the same workload on ONE engine / stream
        sdt = timed(lambda i: eng.infer(d_dm, d_cfg, d_com, out=xyz))
        single = {'value': B * world * args.steps / sdt, 'ms_per_step': sdt / args.steps * 1e3}
        pool.close()
        live_pools.remove(pool)
        pool = None

    # ---- roofline leg: separate profiled pass (events around every op), same workload ------------
    roof = None
    if not args.no_profile:
        eng.h.profile(True)
        nprof = max(2, min(5, args.steps))                     # profiled passes (training with groups: whole windows)
        for i in range(nprof * G):
            if MG > 1:
                eng.infer(m_dm, m_cfg, m_com, out=m_xyz)       # a merged launch, as the replicas run it
            else:
                step(1000 + i)
        if args.detail and rank == 0:
            rows = sorted(eng.h.profile_detail(), key=lambda r: -r['total_ms'])
            tot = sum(r['total_ms'] for r in rows)
            with open(args.detail, 'w') as f:
                f.write('# per-op timing (HIP events), %s mode, %d crops per pass of launches (%d x B=%d), %d profiled passes\n\n'
                        % (mode, B * max(G, MG), max(G, MG), B, nprof))
                f.write('| op | launches/step | us/launch | ms/step | % | TFLOP/s | GB/s (algorithmic) |\n|---|---:|---:|---:|---:|---:|---:|\n')
                for r in rows:
                    ms = r['total_ms'] / nprof
                    f.write('| %s | %.1f | %.1f | %.3f | %.1f | %s | %s |\n' % (
                        r['name'], r['launches'] / nprof, r['total_ms'] * 1e3 / r['launches'], ms, 100 * r['total_ms'] / tot,
                        ('%.1f' % (r['flops'] / (r['total_ms'] * 1e-3) / 1e12)) if r['flops'] else '-',
                        ('%.0f' % (r['bytes'] / (r['total_ms'] * 1e-3) / 1e9)) if r['bytes'] else '-'))
        stats = eng.h.profile_read()
        eng.h.profile(False)
        convs = [s for s in stats if s['name'].startswith('conv_') and s['flops'] > 0]
        if convs:
            dom = max(convs, key=lambda s: s['total_ms'])
            ach = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
            # conv_x3.h forms every fp32 product from SIX bf16 products on v_mfma_f32_32x32x16_bf16: its ceiling is the dense bf16
            # matrix-core rate / 6 (416.7 TFLOP/s of algorithmic fp32-accurate FLOPs), not the fp32-MFMA rate it replaces
            def peak_of(name):
                return (PEAK_BF16_MFMA_TFLOPS / 6.0, 'bf16 MFMA dense %.0f TFLOP/s / 6 products per fp32-accurate product (conv_x3.h, conv_wgrad_x3.h)' % PEAK_BF16_MFMA_TFLOPS) \
                    if name.startswith(('conv_x3', 'conv_wgrad_x3')) else (peak, ('bf16' if bf16 else 'fp32') + ' MFMA dense')
            peak_dom, peak_basis = peak_of(dom['name'])
            pmc_mode = mode + ('_bf16' if bf16 else '') + ('' if (S, F, HW) == (2, 128, 128) else '_s%df%dhw%d' % (S, F, HW))
            traffic, traffic_prov = pmc_traffic(pmc_mode, dom['name'])
            roof = {'kernel': dom['name'], 'bound': 'mfma', 'achieved': ach, 'peak': peak_dom, 'peak_basis': peak_basis,
                    'unit': 'TFLOP/s', 'frac': ach / peak_dom, 'traffic': traffic, 'traffic_provenance': traffic_prov,
                    # GPU-busy evidence that does not depend on an external sampler: HIP-event time of every kernel of a profiled pass
                    # (one stream, nothing overlaps) next to the wall time of a timed step
                    'kernel_ms_per_profiled_pass': sum(s['total_ms'] for s in stats) / nprof,
                    'timed_ms_per_pass': dt / args.steps * 1e3 * (G if mode == 'train' else 1),
                    'launches_per_step': dom['launches'] / nprof, 'avg_launch_us': dom['total_ms'] * 1e3 / dom['launches'],
                    'micro_steps_per_profiled_step': G, 'batches_per_profiled_step': MG,
                    'algorithmic_gflop_per_launch': dom['flops'] / dom['launches'] / 1e9,
                    'share_of_step_time': dom['total_ms'] / max(sum(s['total_ms'] for s in stats), 1e-9),
                    # the runner-up family, same accounting (training: the forward/dgrad tile and the weight gradients trade places)
                    'runner_up': (lambda r: {'kernel': r['name'], 'achieved': r['flops'] / (r['total_ms'] * 1e-3) / 1e12,
                                             'frac': r['flops'] / (r['total_ms'] * 1e-3) / 1e12 / peak_of(r['name'])[0], 'peak': peak_of(r['name'])[0],
                                             'launches_per_step': r['launches'] / nprof,
                                             'avg_launch_us': r['total_ms'] * 1e3 / r['launches']})(
                        sorted(convs, key=lambda s: -s['total_ms'])[1]) if len(convs) > 1 else None,
                    'all_kernels': {s['name']: {'ms_per_step': s['total_ms'] / nprof, 'launches': s['launches'] // nprof,
                                                'tflops': (s['flops'] / (s['total_ms'] * 1e-3) / 1e12) if s['flops'] else None,
                                                'gbs': (s['bytes'] / (s['total_ms'] * 1e-3) / 1e9) if s['bytes'] else None}
                                    for s in stats}}

    eng_flops, eng_pipeline = eng.conv_flops_per_crop(), getattr(eng, 'pipeline', 1)
    if mode == 'train':
        eng.close()                                          # its streams and buffers are not needed by the next leg
    # ---- the other north-star figure, same process, same N: forward(eval) + vote on ICVL crops (replicas, no collective) ----
    fwd_vote = None
    if mode == 'train' and not args.no_forward_vote:
        Ji = DATASETS['icvl']['jnt_num']
        ieng = Engine(S, F, Ji, HW, 3, B, local, training=False)
        if bf16:
            ieng.set_precision('bf16')
        ieng.load_params(random_params(ieng))
        idm, _ip, icfg, icom, _ = make_crops(B, 'icvl', seed=20240, rank=rank, hw=HW)
        i_dm, i_cfg, i_com = ieng.norm_dm(t(idm), t(icom)), t(icfg), t(icom)
        i_xyz = ieng.new(B, 3 * Ji)
        # This leg runs its OWN number of batches: a merged pool launches once per `replicas x merge` batches, so the contract's
        # default of 20 steps would be four launches, two per replica -- ramp, not steady state (the driver's 20-step line said
        # 8370 crops/s where 100 steps say 10.4k).  At least 100 timed batches after at least 10 warm-up ones, whatever --steps is;
        # the leg reports the counts it used.
        fv_steps, fv_warmup = max(args.steps, 100), max(args.warmup, 10)
        idt1 = timed(lambda i: ieng.infer(i_dm, i_cfg, i_com, out=i_xyz), fv_steps, fv_warmup)
        idt = idt1

        def latency_ms(submit_group, n=15):
            """unloaded latency: the device is idle, a batch (or the batches of one merged group) is handed in, the host waits for
            the FIRST batch's joints -- median over n repetitions"""
            ts = []
            for _ in range(n + 2):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                submit_group()
                torch.cuda.synchronize(dev)
                ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts[2:]))
        lat = {'one_engine_one_batch': latency_ms(lambda: ieng.infer(i_dm, i_cfg, i_com, out=i_xyz))}
        if args.replicas * args.merge > 1:
            ipool, iout = make_pool(Ji)
            live_pools.append(ipool)
            idt = timed(lambda i: ipool.submit(i_dm, i_cfg, i_com, out=iout[i % len(iout)]), fv_steps, fv_warmup)

            def one_group():
                tk = [ipool.submit(i_dm, i_cfg, i_com, out=iout[k])[1] for k in range(args.merge)]
                ipool.wait(tk[0])
            # a batch of a merged group leaves with its group: the launch of `merge` x B crops, plus -- in a serving loop -- the
            # wait for the group to fill, which depends on the arrival rate and is not part of this figure
            lat['merged_group_of_%d_batches' % args.merge] = latency_ms(one_group)
            live_pools.remove(ipool)
            ipool.close()
        # `value` is the configuration as BASELINE.json states it -- ONE engine, one batch of B crops per launch.  The serving form (k
        # replicas on k streams, each running `merge` consecutive batches as one launch) is a named sub-object, not the headline of the leg
        # (review of round 5: "quote that one for config 2").  `single_replica` stays as an alias of the top-level figure.
        fwd_vote = {'metric': 'depth-crops/sec fwd(eval)+vote, %d-stack fea=%d @%dx%d' % (S, F, HW, HW),
                    'value': B * world * fv_steps / idt1, 'unit': 'crops/s', 'ms_per_step': idt1 / fv_steps * 1e3,
                    'steps': fv_steps, 'warmup': fv_warmup,
                    'workload': 'ICVL S=%d F=%d J=%d B=%d/GPU %dx%d forward(eval) + vote -> xyz mm, %d GPU(s), one engine per GPU, one batch per launch'
                                % (S, F, Ji, B, HW, HW, world),
                    'single_replica': {'value': B * world * fv_steps / idt1, 'ms_per_step': idt1 / fv_steps * 1e3},
                    'serving_pool': ({'value': B * world * fv_steps / idt, 'unit': 'crops/s', 'ms_per_step': idt / fv_steps * 1e3,
                                      'replicas_per_gpu': args.replicas, 'batches_per_launch': args.merge,
                                      'workload': '%d replica(s) per GPU, each on its own stream; a replica runs %d consecutive batches of %d as one launch'
                                                  % (args.replicas, args.merge, B)} if args.replicas * args.merge > 1 else None),
                    'latency_ms_unloaded': lat,
                    'conv_gflop_per_crop_fwd': ieng.conv_flops_per_crop() / 1e9}
        ieng.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and HW == 128:
        cpu = cpu_baseline(mode, (S, F, J), B, dataset)
        if fwd_vote is not None:                            # the forward(eval)+vote leg gets its own CPU figure (3 + 10 iterations)
            fwd_vote['cpu_baseline'] = cpu_baseline('infer', (S, F, DATASETS['icvl']['jnt_num']), B, 'icvl')

    rccl = None
    if dist is not None:
        try:
            rccl = 'none (dry run over gloo)' if DRY_RUN else '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = 'unknown'

    if rank == 0:
        crops = B * world * args.steps
        out = {
            'metric': ('DRY RUN (host plumbing over gloo, no kernels, measures nothing): ' if DRY_RUN else '') +
                      'depth-crops/sec %s, %d-stack fea=%d @%dx%d' % ('fwd+bwd' if mode == 'train' else 'fwd(eval)+vote', S, F, HW, HW),
            'value': crops / dt, 'unit': 'crops/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': ('%s S=%d F=%d J=%d B=%d/GPU %dx%d ' % (dataset.upper(), S, F, J, B, HW, HW)) +
                       ('train micro-step fwd+loss+bwd, RCCL all-reduce + clip + Adam every %d steps' % args.sub_batch +
                        (' (the %d micro-batches of a window run as one pass of launches, BatchReNorm per micro-batch)' % G if G > 1 else '')
                        if mode == 'train' else 'forward(eval) + vote -> xyz mm') +
                       (', bf16 matrix cores on fp32 tensors (fp32 accumulate, epilogues, vote)' if bf16 else ''),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world,
                       'micro_steps_in_flight': eng_pipeline if mode == 'train' else None,
                       'micro_steps_per_pass': G if mode == 'train' else None,
                       'replicas_per_gpu': args.replicas if mode == 'infer' else None,
                       'batches_per_launch': args.merge if mode == 'infer' else None, 'single_replica': single,
                       'world_size': dist.get_world_size() if dist is not None else 1, 'rccl_version': rccl,
                       'conv_gflop_per_crop_fwd': eng_flops / 1e9},
            'roofline': roof, 'cpu_baseline': cpu, 'forward_vote': fwd_vote,
        }
        if DRY_RUN:
            out['dry_run'] = True
            out['data'] = 'none (dry run)'
            out['config']['checked_all_reduces'] = eng.checked_reductions
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
