#!/usr/bin/env python3
"""
bench.py -- the headline benchmark of BASELINE.json: Mray-bounces/s on the Sandia NSTTF heliostat field.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: `--rays` source rays (default 1e8 =
configs[2], "Sandia NSTTF field, 1e8 rays, 1xMI355X") generated on the device from the Buie-sunshape source
descriptor, traced through the 218-heliostat field + receiver with the Kd-tree, every bounce until the rays
escape or are absorbed, tallies + 50x50 receiver flux map + receiver hit list accumulated on the device.
Rays are sharded over the GPUs by stream id (weak scaling: every rank traces `--rays` rays per step); the only
exchange is ONE all-reduce (RCCL) of the packed tally buffer at the end of the job.

Prints ONE JSON line on rank 0: metric/value/unit per BASELINE.json, plus
  roofline      algorithmic HBM bytes of the fast engine (112 B per ray segment) over the time of its
                kernels for one step, measured with HIP events on the launch stream, against the 8 TB/s HBM peak;
                `traffic` = HBM bytes per step from the rocprofv3 PMC passes kept in profiles/traffic.json
  api_level     the same step through the public entry point, TracerEngine.ray_tracer(tree=False, accel=True): scene
                signature check, hit buffer sized by the engine; the receiver's hits stay on the device until its
                accountants are read -- timed without and with that read after every call
                (N=1 only, after the timed region; `value` above is the C-ABI call trc_trace_fast)
  other_configs the other single-GPU configurations of BASELINE.json through the same C-ABI call, for the record (the
                headline is unchanged): configs[1], dish + receiver at 1e7 rays, and one rank's share of configs[4],
                dish into the spectral cavity at 1.25e8 rays
  check         receiver power, hit fractions; the process exits with status 3 when the receiver power is more than
                5 sigma from the reference's own Monte-Carlo mean (tests/golden/mc_reference.npz)
  cpu_baseline  the oracle (NumPy restatement of the reference's algorithm = the reference's own CPU path, which
                is NumPy too) timed on this host's cores (one process per core, independent batches) on a bounded
                sample of the same workload, rank 0 at N=1 only, before the GPU part starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_SEG = 112.0          # algorithmic bytes per ray segment: read + write of x,y,z,dx,dy,dz,E in float64
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_worker(n_rays, seed, batches):
    """one CPU worker: the oracle, brute force like the reference's default path, `batches` bundles of n_rays NSTTF source rays"""
    import numpy as N
    from tracer_amd import scenes
    from tracer_amd.scene import compile_scene
    from oracle import engine as oracle_engine
    plant, field, rec, src = scenes.nsttf_field()
    cs = compile_scene(plant)
    t0 = time.time()
    seg, kw = 0, 0.
    for k in range(batches):
        b = scenes.nsttf_source(n_rays, src, seed=seed, ray_offset=k * n_rays)
        with N.errstate(all='ignore'):
            ref = oracle_engine.trace_from_compiled(cs, b.source_args(), reps=100, min_energy=1e-10)
        seg += int(ref['segments'])
        kw += float(ref['absorbed'][-1] / 1e3) / batches
    dt = time.time() - t0
    return dict(segments=seg, seconds=dt, receiver_kW=kw)


def cpu_baseline(n_rays, workers, batches):
    """
    oracle (kind "port") timed on this host: `workers` independent processes, one core each, every one tracing its own
    batch of n_rays NSTTF source rays (the reference's multi-core driver, tracer_engine_mp.py, also runs independent
    batches per process).  Must run BEFORE this process touches the GPU: the workers are started with exec.
    """
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', str(n_rays), str(99 + w), str(batches)],
                              stdout=subprocess.PIPE, env=env, cwd=ROOT) for w in range(workers)]
    res = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('CPU baseline worker failed')
        res.append(json.loads(out.decode().strip().splitlines()[-1]))
    wall = time.time() - t0
    seg = sum(r['segments'] for r in res)
    slowest = max(r['seconds'] for r in res)
    per_core = [r['segments'] / r['seconds'] / 1e6 for r in res]
    return dict(value=seg / slowest / 1e6, unit='Mray-bounces/s', cores=workers, kind='port',
                per_core=sum(per_core) / len(per_core),
                sample='NSTTF 218 heliostats + receiver, brute force (the reference default), %d processes x %d bundles of %d '
                       'source rays (%d segments in total), slowest worker %.1f s, wall incl. start-up %.1f s'
                       % (workers, batches, n_rays, seg, slowest, wall),
                receiver_kW=sum(r['receiver_kW'] for r in res) / len(res))


def traffic_for(n, accel):
    """HBM bytes per step from the committed PMC passes (profiles/traffic.json, made by tools/pmc_traffic.py on this workload):
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, the same with FETCH_SIZE as counted), or (None, None) when the
    passes were taken on another workload"""
    tf = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        tj = json.load(open(tf))
        if int(tj.get('rays_per_launch', 0)) == int(n) and bool(tj.get('accel', True)) == bool(accel):
            return tj.get('hbm_bytes_per_launch'), tj.get('hbm_bytes_per_launch_fetch_as_counted')
    except Exception:
        pass
    return None, None


def other_configs(ctx, log):
    """
    The other single-GPU configurations of BASELINE.json through the same C-ABI call (trc_trace_fast, streaming form), third
    run of three each: segments per second by the time of the kernels (HIP events) and by the wall clock of the call.
      configs[1]  parabolic dish + circular receiver, Buie sunshape, 1e7 rays from the source descriptor
      configs[4]  one rank's share (1.25e8 of 1e9 rays over 8 GPUs) of the dish into the cavity with spectral optics and slope
                  error; every ray carries a wavelength, so the bundle is handed over as host arrays (7 columns over PCIe in
                  the wall-clock figure, not in the kernel one)
    """
    import numpy as N
    from tracer_amd import scenes
    from tracer_amd.scene import compile_scene, DeviceScene
    from tracer_amd.ray_bundle import RayBundle
    out = {}
    asm, dish_s, rec_s, src = scenes.dish()
    dev = DeviceScene(compile_scene(asm), ctx)
    n1 = 10 ** 7
    for r in range(3):
        t0 = time.time()
        st, _ = dev.trace_fast(scenes.dish_source(n1, src, seed=3 + r), 100, 1e-10, 3 + r, accel=True, stream=True)
        wall = time.time() - t0
    a, rcv, h = dev.get_tallies()
    dev.close()
    out['configs[1]'] = {'workload': 'parabolic dish (D 5 m, f 3 m, slope error 2 mrad) + circular receiver, Buie sunshape CSR 0.05, 1e7 rays',
                         'rays': n1, 'segments': int(st.segments), 'kernel_ms': st.kernel_ms, 'wall_ms': wall * 1e3,
                         'Gsegments_per_s_kernels': st.segments / st.kernel_ms / 1e6, 'Gsegments_per_s_wall': st.segments / wall / 1e9,
                         'roofline_frac_algorithmic': st.segments * B_SEG / (st.kernel_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                         'intercept': float(a[1] / (a[0] / 0.06 if a[0] > 0 else 1.))}
    log('configs[1]: %.2f ms of kernels, %.1f G segments/s' % (st.kernel_ms, st.segments / st.kernel_ms / 1e6))
    ts, src = scenes.dish_cavity()
    n4 = 125000000
    b0 = scenes.dish_source(n4, src, seed=9)
    v, d, e = N.asarray(b0.get_vertices()), N.asarray(b0.get_directions()), N.asarray(b0.get_energy())
    wl = N.random.default_rng(4).uniform(0.3e-6, 2.5e-6, n4)
    dev = DeviceScene(ts, ctx)
    for r in range(3):
        dev.reset_tallies()
        t0 = time.time()
        st, _ = dev.trace_fast(RayBundle(vertices=v, directions=d, energy=e, wavelengths=wl), 12, 1e-3 * e[0], 31, stream=True)
        wall = time.time() - t0
    a, rcv, h = dev.get_tallies()
    dev.close()
    out['configs[4]'] = {'workload': 'one rank\'s share of 1e9 rays over 8 GPUs: dish with slope error into a cavity of seven walls with angle- and '
                                     'wavelength-tabulated optics, rays with wavelengths handed over as host arrays, reps=12',
                         'rays': n4, 'segments': int(st.segments), 'kernel_ms': st.kernel_ms, 'wall_ms': wall * 1e3,
                         'Gsegments_per_s_kernels': st.segments / st.kernel_ms / 1e6, 'Gsegments_per_s_wall': st.segments / wall / 1e9,
                         'roofline_frac_algorithmic': st.segments * B_SEG / (st.kernel_ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
                         'absorbed_share': float(a.sum() / e.sum()), 'launches': int(st.launches)}
    log('configs[4]: %.1f ms of kernels, %.1f G segments/s' % (st.kernel_ms, st.segments / st.kernel_ms / 1e6))
    # a mesh of 1e5 faces (one object of arrays on the host, the large grid on the device)
    from tracer_amd import sources
    t0 = time.time()
    asm, nf, (center, direction, radius, csr) = scenes.relief_mesh()
    cs = compile_scene(asm)
    t_build = time.time() - t0
    dev = DeviceScene(cs, ctx)
    nm = 10 ** 7
    for r in range(3):
        t0 = time.time()
        st, _ = dev.trace_fast(sources.buie_sunshape(nm, center, direction, radius, csr, flux=1., seed=23 + r), 6, 1e-10, 23 + r, accel=True)
        wall = time.time() - t0
    dev.close()
    out['mesh'] = {'workload': 'relief of %d mirror triangles (models/triangulated_surface.py) under a black lid, Buie sunshape, 1e7 rays, reps=6' % nf,
                   'rays': nm, 'segments': int(st.segments), 'kernel_ms': st.kernel_ms, 'wall_ms': wall * 1e3,
                   'Gsegments_per_s_kernels': st.segments / st.kernel_ms / 1e6, 'Gsegments_per_s_wall': st.segments / wall / 1e9,
                   'build_and_compile_s': t_build, 'launches': int(st.launches)}
    log('mesh: %.1f ms of kernels, %.2f G segments/s' % (st.kernel_ms, st.segments / st.kernel_ms / 1e6))
    return out


def sq_counters_for(n):
    """per-kernel SQ counter ratios of this workload from the committed single-batch PMC passes (profiles/sq_counters.json, made by
    tools/pmc_kernels.py), or None when they were taken on another workload"""
    try:
        sj = json.load(open(os.path.join(ROOT, 'profiles', 'sq_counters.json')))
        if int(sj.get('rays_per_launch', 0)) != int(n):
            return None
        return dict((k, dict((a, round(b, 4)) for a, b in v.items() if a != 'counters')) for k, v in sj['per_kernel'].items()
                    if v.get('counters', {}).get('SQ_WAVES', 0) >= 64)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--rays', type=float, default=1e8, help='source rays per step per GPU')
    ap.add_argument('--cpu-rays', type=int, default=250000, help='source rays per bundle of a CPU baseline worker (0 = skip)')
    ap.add_argument('--cpu-batches', type=int, default=6, help='bundles traced by every CPU baseline worker')
    ap.add_argument('--cpu-workers', type=int, default=0, help='CPU baseline processes (0 = one per core, at most 16)')
    ap.add_argument('--cpu-worker', nargs=3, metavar=('RAYS', 'SEED', 'BATCHES'), help=argparse.SUPPRESS)
    ap.add_argument('--api-steps', type=int, default=2, help='steps timed through TracerEngine.ray_tracer (N=1 only; 0 = skip)')
    ap.add_argument('--no-accel', action='store_true', help='brute force instead of the accelerated candidate search')
    ap.add_argument('--no-extras', action='store_true', help='skip the other single-GPU configurations (other_configs; N=1 only)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help="weak: every rank traces --rays rays per step (the driver's contract); strong: the ranks share the --rays rays of "
                         "a step (distributed.shard), results identical for every number of GPUs")
    ap.add_argument('--kernel', choices=['auto', 'stream', 'megakernel'], default='auto',
                    help='fast-engine form: streaming kernels (default at this size) or the single persistent kernel')
    args = ap.parse_args()

    if args.cpu_worker:
        print(json.dumps(cpu_worker(int(args.cpu_worker[0]), int(args.cpu_worker[1]), int(args.cpu_worker[2]))), flush=True)
        return

    exit_code = 0
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        log('note: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE' % (world, args.gpus))
    n = int(args.rays)

    cpu = None
    if rank == 0 and world == 1 and args.cpu_rays > 0:
        # before anything initialises the GPU in this process (the workers are separate programs)
        workers = args.cpu_workers or min(os.cpu_count() or 1, 16)
        log('timing the CPU baseline (oracle, %d processes x %d x %d rays) ...' % (workers, args.cpu_batches, args.cpu_rays))
        cpu = cpu_baseline(args.cpu_rays, workers, args.cpu_batches)

    import torch
    import torch.distributed as dist
    # TRC_BENCH_BACKEND=gloo rehearses the multi-rank logic on a box with fewer GPUs than ranks (ranks share GPUs)
    backend = os.environ.get('TRC_BENCH_BACKEND', 'nccl')
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend=backend)

    import numpy as N
    from tracer_amd import _cabi, scenes
    from tracer_amd.accel_tree import KdTree
    from tracer_amd.scene import compile_scene, DeviceScene
    from tracer_amd.distributed import reduce_scene_tallies, all_reduce_sum

    _cabi.set_default_device(local)
    ctx = _cabi.get_context(local)
    plant, field, rec, src = scenes.nsttf_field()
    cs = compile_scene(plant)
    dev = DeviceScene(cs, ctx)
    accel = not args.no_accel
    if accel:
        dev.set_kdtree(KdTree(plant, 8 + 1.3 * N.log(cs.n_surf), min_leaf=1))
    ue, ve = scenes.nsttf_fluxmap_edges()
    dev.set_fluxmap(218, ue, ve)
    dev.set_hit_capacity(int(0.08 * n * (args.steps + args.warmup)) + 4096 + 16 * 4096)   # receiver hits ~6.4 % of the source rays

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    strong = args.scaling == 'strong'
    n_total = n                     # rays of a step over all ranks when they share it (strong scaling)
    if strong:
        from tracer_amd.distributed import shard
        lo, hi = shard(n_total, rank, world)
        n = hi - lo

    def step(k):
        # stream ids: disjoint per (step, rank) -- results do not depend on the number of GPUs
        offset = (k * n_total + lo) if strong else (k * world + rank) * n
        if strong:
            # (the energy of a ray is flux x area / rays of the WHOLE step, whichever rank traces it)
            b = scenes.nsttf_source(n, src, seed=2024, ray_offset=offset, n_total=n_total)
        else:
            b = scenes.nsttf_source(n, src, seed=2024, ray_offset=offset)
        stats, _ = dev.trace_fast(b, 100, 1e-10, 2024, accel=accel, stream={'auto': None, 'stream': True, 'megakernel': False}[args.kernel])
        return stats

    for k in range(args.warmup):
        step(k)
    if world > 1:
        reduce_scene_tallies(dev)      # communicator set-up outside the timed region
    dev.reset_tallies()
    dev.lib.trc_scene_clear_hits(dev.handle)

    barrier()
    t0 = time.time()
    seg = 0
    kms = 0.0
    for k in range(args.steps):
        st = step(args.warmup + k)
        seg += st.segments
        kms += st.kernel_ms
        launches = st.launches
    # the single exchange of the job: sum the tally buffers of all ranks (per-surface energies, counts, flux map)
    reduce_scene_tallies(dev)
    barrier()
    dt = time.time() - t0

    if world > 1:
        total_segments = float(all_reduce_sum(N.array([float(seg)]))[0])
        times = all_reduce_sum(N.eye(world)[rank] * dt)        # every rank's time; the job takes the slowest
        dt_max = float(times.max())
    else:
        total_segments, dt_max = float(seg), dt

    api = None
    if rank == 0:
        a, r, h = dev.get_tallies()
        fm = dev.get_fluxmap(218)
        if world == 1 and args.api_steps > 0 and args.kernel == 'auto':
            # the public entry point on the same workload (a scene of its own: the engine compiles the assembly itself)
            dev.close()
            dev = None
            from tracer_amd.tracer_engine import TracerEngine
            eng = TracerEngine(plant)
            eng.set_fluxmap(218, ue, ve)
            mk = lambda k: scenes.nsttf_source(n, src, seed=2024, ray_offset=(args.steps + args.warmup + 1 + k) * n)
            rec_opt = plant.get_surfaces()[218].get_optics_manager()
            eng.ray_tracer(mk(0), reps=100, min_energy=1e-10, tree=False, accel=accel, seed=2024)       # warm-up: allocations
            rec_opt.get_all_hits()                                                                      # ... and the page-locked blocks
            plant.reset_all_optics()
            torch.cuda.synchronize()

            def api_loop(read):
                t1 = time.time()
                aseg, got = 0, 0
                for k in range(args.api_steps):
                    eng.ray_tracer(mk(1 + k), reps=100, min_energy=1e-10, tree=False, accel=accel, seed=2024)
                    aseg += eng.stats['segments']
                    if read:
                        got += len(rec_opt.get_all_hits()[0])     # absorbed energies and hit points of the step, on the host
                        plant.reset_all_optics()
                torch.cuda.synchronize()
                adt = time.time() - t1
                plant.reset_all_optics()
                return adt / args.api_steps * 1e3, aseg / adt / 1e6, got
            ms_unread, v_unread, _ = api_loop(False)
            ms_read, v_read, got = api_loop(True)
            api = {'entry': 'TracerEngine.ray_tracer(bundle, reps=100, min_energy=1e-10, tree=False, accel=%r)' % accel,
                   'steps': args.api_steps, 'ms_per_step': ms_unread, 'value': v_unread, 'unit': 'Mray-bounces/s',
                   'ms_per_step_hits_read': ms_read, 'value_hits_read': v_read, 'hits_read_per_step': got // max(args.api_steps, 1),
                   'includes': 'scene signature check, hit buffer kept or grown (the hits of successive calls stay on the device until an '
                               'accountant is read), trc_trace_fast, marks handed to the receiver\'s accountants; hits_read: plus '
                               'get_all_hits() of the receiver after every call -- the step\'s receiver hits (absorbed energy + hit point, '
                               '36 B each) packed on the device and copied into page-locked host arrays -- and reset_all_optics()'}
        total_rays = float(n_total) * args.steps * (1 if strong else world)
        e_ray = 1000. * N.pi * src['radius'] ** 2 / n_total  # energy per ray of ONE step's bundle
        n_bundles = args.steps * (1 if strong else world)     # independent bundles of n_total rays the job traced
        receiver_kw = a[218] / n_bundles / 1e3                # mean over the independent batches
        ach = (seg * B_SEG / 1e9) / (kms / 1e3) if kms > 0 else 0.0     # GB/s, this rank's launches
        traffic, traffic_raw = traffic_for(n, accel)
        out = {
            'metric': 'Mray-bounces/s on Sandia NSTTF field',
            'value': total_segments / dt_max / 1e6,
            'unit': 'Mray-bounces/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt_max / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'Sandia NSTTF heliostat field: 218 heliostats + receiver, Buie sunshape CSR 0.01, '
                                   '%.0e source rays per step per GPU, %s, reps=100, min_energy=1e-10; tallies + 50x50 '
                                   'flux map + receiver hit list on device'
                                   % (n, 'accel=True (Kd-tree built on the host as in the reference; the device searches a '
                                         'uniform grid over the same geometry boxes)' if accel else 'brute force'),
                       'rays_per_step_per_gpu': n, 'segments_per_step_per_gpu': seg / args.steps, 'accel': accel,
                       'parallelism': 'rays sharded by stream id over %d GPU(s) (%s), one all-reduce of tallies at the end' % (world, 'the ranks share the rays of a step' if strong else 'every rank its own rays per step')},
            'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_fetch_as_counted': traffic_raw,
                         'kernel': 'k_trace_coop<512>' if launches == 1 else
                                   'fast engine, streaming form: k_s_cull + k_s_fresh2 (+ the general path for the aureole) + k_s_shade_c<mirror>, '
                                   'then k_s_bounce (the hits on the receiver finished inside) + k_s_shade_c<mirror> per bounce '
                                   '(%d launches per step)' % launches,
                         'kernel_ms_per_launch': kms / args.steps,
                         'algorithmic_bytes_per_launch': seg / args.steps * B_SEG,
                         # what the counters say moves through HBM, over the same kernel time: the fraction of the 8 TB/s actually used
                         # (`frac` above prices every segment at 112 B, and 94 % of the segments end in registers)
                         'hbm_measured_frac': (traffic / (kms / args.steps * 1e-3) / (HBM_PEAK_GBS * 1e9)) if (traffic and kms > 0) else None,
                         'hbm_measured_frac_fetch_as_counted': (traffic_raw / (kms / args.steps * 1e-3) / (HBM_PEAK_GBS * 1e9)) if (traffic_raw and kms > 0) else None,
                         # what binds each kernel: vector instructions issued per resident wave cycle and the share of those cycles spent
                         # waiting, from the single-batch PMC passes kept in profiles/sq_counters.json (tools/pmc_kernels.py)
                         'kernels': sq_counters_for(n)},
            'check': {'receiver_kW': receiver_kw, 'receiver_hits': int(h[218]), 'heliostat_hits': int(h[:218].sum()),
                      'segments_total': int(round(total_segments)), 'receiver_hits_per_ray': h[218] / total_rays,
                      'heliostat_hits_per_ray': float(h[:218].sum()) / total_rays,
                      'fluxmap_sum_kW': float(fm.sum()) / n_bundles / 1e3, 'energy_per_ray_W': e_ray},
        }
        out['api_level'] = api
        out['cpu_baseline'] = cpu
        if world == 1 and not args.no_extras and args.kernel == 'auto' and n >= 10 ** 7:
            try:
                if dev is not None:
                    dev.close()
                    dev = None
                out['other_configs'] = other_configs(ctx, log)
            except Exception as err:        # (for the record only: the headline stands without them)
                log('other_configs skipped: %r' % (err,))
                out['other_configs'] = None
        # the reference's own Monte-Carlo mean of the receiver power (10 runs of 1e5 rays): 5 sigma of both estimates
        ok = True
        try:
            mc = N.load(os.path.join(ROOT, 'tests', 'golden', 'mc_reference.npz'))
            p_ref, se_ref = float(mc['nsttf_receiver_mean']) / 1e3, float(mc['nsttf_receiver_se']) / 1e3
            se_gpu = e_ray * N.sqrt(max(h[218], 1.)) / n_bundles / 1e3
            sig = float(N.sqrt(se_ref ** 2 + se_gpu ** 2))
            out['check']['reference_receiver_kW'] = p_ref
            out['check']['sigma_kW'] = sig
            out['check']['deviation_sigma'] = abs(receiver_kw - p_ref) / sig
            ok = abs(receiver_kw - p_ref) <= 5. * sig and abs(float(fm.sum()) / n_bundles / 1e3 - receiver_kw) <= 1e-6 * receiver_kw
            out['check']['ok'] = bool(ok)
        except Exception as err:        # no fixture: nothing to compare with
            out['check']['ok'] = None
            log('check skipped: %s' % err)
        print(json.dumps(out), flush=True)
        if not ok:
            log('CHECK FAILED: receiver power %.1f kW vs reference %.1f kW' % (receiver_kw, out['check'].get('reference_receiver_kW', float('nan'))))
            exit_code = 3
    if dev is not None:
        dev.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == '__main__':
    main()
