#!/usr/bin/env python3
"""bench.py - MPix/s of the end-to-end 5-stage fixed ISP forward on 256x256 Bayer patches
(BASELINE.json metric), one process per GPU, batch sharded across ranks (weak scaling, no
data-path collective: every image is independent).

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python bench.py --gpus N ...          # starts N fresh rank processes itself (torch.distributed.run, RCCL)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one forward of the pipeline over one batch of 64 synthetic 256x256 RGGB patches that is
already resident in HBM.  Workloads (SURVEY.md section 8d / BASELINE.md):
  isp (headline `value`)  Demosaic_01_sRGB_07_11_01_14 = nearest demosaic -> bilateral denoise ->
                          WbManual -> Gamma -> GtmManual (OriginUniversal; the literal demosaic / denoise /
                          white-balance / gamma / tone-map ordering), every stage output materialised,
                          ONE fused launch (risp_bilateral_chain_fwd); HBM roofline
  pointwise (`extra.pointwise_*`)  Bayer_02_Demosaic_01_sRGB_11_01_14 = skip -> nearest demosaic ->
                          WbManual -> Gamma -> GtmManual (no neighbourhood stage), one risp_chain_fwd launch
  cnn (`extra.cnn_*`)     Bayer_01_Demosaic_03_sRGB_01_13_11 (options/train/SID_isp.yml:28) =
                          Path-Restore-Bayer -> proxy demosaic -> Gamma -> WbQuadratic -> WbManual;
                          fp32-MFMA roofline
  search (`extra.search_step`)  BASELINE config 4: one DARTS iteration (optimize_alphas + optimize_parameters,
                          train.py:230-246, darts_model.py:159-324) of the 4-slot super-net (n_step 2) on a GLOBAL batch
                          of 32 train + 32 val 256x256 patches, sharded 32/N per rank, gradients averaged over RCCL
                          (strong scaling of a fixed job); the same job on one GPU is timed beside it
Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
MFMA_F16_PEAK_TFLOPS = 2516.6   # dense f16 / bf16 MFMA: 1024 FLOP / clk / SIMD x 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PF)
MFMA_F16_SUSTAINED_TFLOPS = 1698.0   # what the bare f16 stream sustains on random halves under the power limit (profiles/r04_mfma_f16_power.txt)

ARCH_HBM = 'Bayer_02_Demosaic_01_sRGB_11_01_14'           # element-wise only: one fused launch
ARCH_DENOISE = 'Demosaic_01_sRGB_07_11_01_14'             # nearest demosaic, bilateral, WbManual, Gamma, GtmManual
ARCH_CNN = 'Bayer_01_Demosaic_03_sRGB_01_13_11'
# algorithmic HBM bytes per pixel of a fused launch: read the mosaic once (4 B) and write each materialised
# BGR stage output (12 B each); `skip` aliases its input (0 B).  SURVEY 8d's stage-by-stage figures
# (112 / 88 B/pix) also count the intermediate re-reads that the fused launches avoid.
BYTES_PER_PIX_ISP = 4 + 5 * 12          # demosaic, bilateral, wb, gamma, tone curve outputs
BYTES_PER_PIX_FUSED = 4 + 4 * 12        # point-wise pipeline: demosaic, wb, gamma, tone curve outputs
BYTES_PER_PIX_UNFUSED = 88
FLOP_PER_PIX_CNN = 239680      # 223488 (Path14lBayer) + 16192 (SRCNNDemosaic)


def build_pipeline(arch, device, which='IspUniversal'):
    from reconfigisp_amd.codes.models import networks
    opt = {'network_G': {'which_model_G': which, 'architecture': arch, 'module_path': None,
                         'individual_module_paths': [None] * 8}}
    torch.manual_seed(10)
    net = networks.define_G(opt).to(device)
    net.eval()
    return net


def timed(fn, steps, warmup, device, world):
    # The cyclic GC is collected once BEFORE the warm-up and paused until the timed region ends (as timeit does): a
    # collection of this process's heap is a ~40 ms host stall, and 40 ms of idle GPU right before the timed region
    # drops the device into its low-power clocks, which the first ~2 ms of timed launches then pay for.
    gc.collect()
    gc.disable()
    # Spin-up (untimed, before the W warm-up steps): ~50 ms of the same work issued WITHOUT intermediate synchronisation.
    # After an idle period - and while launches are interleaved with host synchronisations - the device runs slower for a
    # few hundred launches (tools/diag_ramp.py, diag_ramp2.py: 46 -> 60 us per step); without this a short timed window
    # (K = 200) measures that transient, not the throughput.
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_a.record()
    for _ in range(3):
        fn()
    ev_b.record()
    ev_b.synchronize()
    est_ms = max(ev_a.elapsed_time(ev_b) / 3, 1e-3)
    timed.spin_up_launches = 3 + min(4000, int(50.0 / est_ms) + 1)
    for _ in range(timed.spin_up_launches - 3):
        fn()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    marks = []
    for _ in range(steps):
        fn()
        marks.append(time.perf_counter())
    ev1.record()
    t_issued = time.perf_counter()
    # host time to ISSUE a step (diagnostic): mean, median and worst single step
    gaps = sorted(b - a for a, b in zip([t0] + marks[:-1], marks))
    timed.host_us = (marks[-1] - t0) / steps * 1e6
    timed.host_med_us, timed.host_max_us = gaps[len(gaps) // 2] * 1e6, gaps[-1] * 1e6
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    gc.enable()
    dev_ms = ev0.elapsed_time(ev1)
    if os.environ.get('RISP_BENCH_TIMELINE') == '1':      # diagnostic: where a short window's fixed cost goes
        sys.stderr.write('window: first step issued after %.1f us, all %d after %.1f us, wall %.1f us, device ev0->ev1 %.1f us\n'
                         % ((marks[0] - t0) * 1e6, steps, (t_issued - t0) * 1e6, wall * 1e6, dev_ms * 1e3))
    if world > 1:
        t = torch.tensor([wall], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = t.item()
    return wall, dev_ms / steps


def kernel_time_ms(net, batches, reps, device, isp):
    """Average launch duration of the dominant kernel (the one fused launch a step consists of), measured
    live with a HIP event pair on the launch stream around `reps` back-to-back launches issued straight
    through the C ABI (one ctypes call each, so the stream never drains and the figure is kernel + the
    ~1.5 us same-stream launch boundary, not host time).  The launches rotate over `batches` (each with its own
    output buffers): with one batch the 285 MB working set is largely served by the 256 MiB Infinity Cache,
    with several it streams from / to HBM."""
    import reconfigisp_amd.functional as F
    n = batches[0].shape[0]
    with torch.no_grad():
        pars = list(net._stage_params(n))
    plans = []
    for bay in batches:
        if isp:     # demosaic | bilateral | wbmanual, gamma, gtmmanual
            d = net.all_modules[1]._params(pars[1].detach(), {})
            plans.append(F.BilateralChainPlan(bay, True, d['window_length'].to(torch.int32), d['sigma_color'],
                                              d['sigma_space'], int(d['window_length'].max().item()),
                                              [F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL], [pars[2] * 5, pars[3], pars[4]]))
        else:       # skip | demosaic, wbmanual, gamma, gtmmanual (the kernel takes the gain = params * 5)
            plans.append(F.ChainPlan(bay, [F.OP_DEMOSAIC_NEAREST, F.OP_WB_MANUAL, F.OP_GAMMA, F.OP_GTM_MANUAL],
                                     [None, pars[2] * 5, pars[3], pars[4]]))
    # same untimed spin-up as timed(): ~50 ms of launches without intermediate synchronisation, then straight into
    # the measured launches (no host synchronisation in between: it would re-introduce the clock transient)
    for k in range(max(2 * len(plans), 1000)):
        plans[k % len(plans)].launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        plans[k % len(plans)].launch()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def launched_kernel():
    """the instance of the fused kernel a bench step launches, as rocprofv3 names it (asked of the library, not assumed)"""
    from reconfigisp_amd import lib as L
    return L.load().risp_bilateral_chain_kernel(1, 3, 0).decode()


def measured_traffic(algorithmic_bytes):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, regenerated each round by
    tools/collect_r05.py from that round's counter passes) - only when they were taken on THIS kernel instance and this very
    workload; otherwise null (a stale file must not be reported as a measurement)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
        if t.get('kernel') == launched_kernel() and t.get('algorithmic_bytes_per_launch') == algorithmic_bytes:
            return t['traffic_bytes_per_launch']
    except (OSError, ValueError, KeyError, RuntimeError, AttributeError):
        pass
    return None


def host_cpu():
    """(model string, physical cores, logical cores) from /proc/cpuinfo"""
    model, cores = 'unknown', set()
    try:
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for ln in f:
                k, _, v = ln.partition(':')
                k, v = k.strip(), v.strip()
                if k == 'model name':
                    model = v
                elif k == 'physical id':
                    phys = v
                elif k == 'core id':
                    core = v
                elif not k and phys is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) or logical), logical


ARCH_CONFIG1 = 'Demosaic_01_sRGB_11_01'                    # BASELINE configs[0]: demosaic -> WB -> gamma


def cpu_baseline_config1(size, seconds=4.0):
    """BASELINE.json configs[0]: test.py on ONE 256 x 256 synthetic Bayer patch through the 3-stage
    fixed pipeline demosaic -> WbManual -> Gamma, the oracle (torch CPU fp32) on the host cores.  One patch is 65 k pixels of
    element-wise work: threads beyond a few only add overhead, so 1, 4 and 8 threads are tried for `seconds` / 3 each."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import isp_oracle as O
    from reconfigisp_amd.codes.data.synthetic_raw import make_batch
    bay, _ = make_batch(1, size, size, seed=10)
    names = O.parse_architecture(ARCH_CONFIG1)
    raw = [torch.tensor(O.PARAM_INIT[k]) for k in names]
    best = None
    with torch.no_grad():
        for th in (1, 4, 8):
            torch.set_num_threads(th)
            O.fixed_pipeline(bay, names, raw, [None] * len(names), origin=True)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < seconds / 3:
                O.fixed_pipeline(bay, names, raw, [None] * len(names), origin=True)
                reps += 1
            rate = size * size * reps / (time.perf_counter() - t0) / 1e6
            if best is None or rate > best[0]:
                best = (rate, th, reps)
    return {'value': round(best[0], 2), 'unit': 'MPix/s', 'cores': best[1], 'kind': 'port',
            'sample': 'one %dx%d patch x %d repetitions of %s through oracle/isp_oracle.py (torch CPU fp32) at %d thread(s), the '
                      'best of 1 / 4 / 8' % (size, size, best[2], ARCH_CONFIG1, best[1])}


def cpu_baseline(bay, arch, point_s=3.0, min_iters=10):
    """The CPU oracle (restatement of the reference's torch-CPU path, oracle/isp_oracle.py) on the WHOLE batch the GPU
    leg times, on the host cores of this box.  Thread sweep 8, 16, ... up to every logical core,
    `point_s` seconds per point; then at least `min_iters` timed repetitions at the best count.  About 25-35 s in total."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import isp_oracle as O
    model, physical, logical = host_cpu()
    names = O.parse_architecture(arch)
    raw = [torch.tensor(O.PARAM_INIT[k]) for k in names]
    sample = bay.cpu()
    n = sample.shape[0]
    run = lambda: O.fixed_pipeline(sample, names, raw, [None] * len(names), origin=True)

    def rate(threads, seconds, min_reps):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        run()                                           # warm-up repetition, also the guard below
        if time.perf_counter() - t0 > 4 * seconds:      # pathological point (hundreds of threads on small ops): one rep is the rate
            return 1, time.perf_counter() - t0
        reps, t0 = 0, time.perf_counter()
        while reps < min_reps or time.perf_counter() - t0 < seconds:
            run()
            reps += 1
        return reps, time.perf_counter() - t0

    points = sorted({t for t in (8, 16, 32, 64, 128, 256, 512) if t < logical} | {logical})
    sweep = {}
    with torch.no_grad():
        for th in points:
            reps, dt = rate(th, point_s, 1)
            sweep[th] = reps / dt
            if sweep[th] < 0.5 * max(sweep.values()):   # past the knee: more threads only oversubscribe these small ops
                break
        best = max(sweep, key=sweep.get)
        reps, dt = rate(best, point_s, min_iters)
    pix = n * bay.shape[2] * bay.shape[3]
    return {'value': round(pix * reps / dt / 1e6, 2), 'unit': 'MPix/s', 'cores': best, 'kind': 'port',
            'cpu_model': model, 'physical_cores': physical, 'logical_cores': logical,
            'thread_sweep_MPix_s': {str(t): round(pix * r / 1e6, 2) for t, r in sorted(sweep.items())},
            'sample': 'the whole batch of %d 256x256 patches x %d repetitions of %s through oracle/isp_oracle.py (torch CPU '
                      'fp32) at the best thread count of the sweep (%d threads; %s, %d physical / %d logical cores)'
                      % (n, reps, arch, best, model, physical, logical)}


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """GPUs this process tree may use, WITHOUT loading the HIP runtime: the *_VISIBLE_DEVICES lists when set, else the
    KFD topology nodes that have SIMDs (CPU nodes report simd_count 0)."""
    for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(',') if t.strip()])
    count = 0
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(ln.split(None, 1) for ln in f.read().splitlines() if ' ' in ln)
            count += int(props.get('simd_count', '0').strip()) > 0
    except OSError:
        return 0
    return count


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N FRESH rank processes (torch.distributed.run, one per
    GPU, rendezvous on 127.0.0.1) as children of this process, which has not touched the GPU and never will; wait,
    pass rank 0's JSON line through, exit with the children's code.  Nothing is exec'ed."""
    have = visible_gpus()                       # sysfs / environment only: the parent never touches the HIP runtime
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if have < n and env.get('RISP_BENCH_LAUNCH_ONLY') != '1':
        if env.get('RISP_BENCH_ONE_DEVICE') != '1':
            sys.stderr.write('bench.py: --gpus %d but %d GPU(s) visible.  (RISP_BENCH_ONE_DEVICE=1 RISP_BENCH_BACKEND=gloo runs '
                             'all ranks on device 0 as a plumbing dry run; its numbers are not an N-GPU measurement.)\n' % (n, have))
            return 2
        env.setdefault('RISP_BENCH_BACKEND', 'gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = child.communicate(timeout=float(env.get('RISP_BENCH_TIMEOUT', '1500')))
    except subprocess.TimeoutExpired:           # a rank stuck in a collective: end the whole child tree, keep what was printed
        import signal
        os.killpg(child.pid, signal.SIGKILL)
        out, _ = child.communicate()
        sys.stderr.write('bench.py: the ranks did not finish within the time limit (RISP_BENCH_TIMEOUT); killed\n')
    lines = [ln for ln in out.splitlines() if ln.startswith('{') and '"metric"' in ln]
    for ln in out.splitlines():
        if ln not in lines:
            sys.stderr.write(ln + '\n')
    if lines:
        print(lines[-1])
    return child.returncode if child.returncode or lines else 1


def search_opt(n_step, distributed):
    from collections import OrderedDict
    return OrderedDict(model='darts', gpu_ids=[0], dist=distributed, is_train=True,
                       network_G=dict(which_model_G='SuperPruneFifteenDemosFourBayerTwo', n_step=n_step, n_modules=15,
                                      prune_threshold=0.2, module_path=None),
                       path=dict(pretrain_model_G=None, strict_load=True),
                       train=dict(lr_G=1e-4, momentum_G=0.9, lr_meta=1e-4, beta1=0.9, beta2=0.99, pixel_criterion='l2',
                                  lr_scheme='MultiStepLR', lr_steps=[100000], restarts=None, restart_weights=None,
                                  lr_gamma=0.5, clear_state=False))


def _family(name, args):
    """kernel family of a C-ABI call (the convolution entries by their descriptor's filter size)"""
    k = 0
    if name.startswith('risp_conv2d') and args and hasattr(args[0], '_obj'):
        k = getattr(args[0]._obj, 'ksize', 0)
    if name in ('risp_conv2d_f16x2', 'risp_conv2d_f16x2_uniform'):
        return 'wide %dx%d layers (split precision)' % (k, k)
    if name.startswith(('risp_conv2d_tapout', 'risp_conv2d_toep', 'risp_conv2d_thin5')):
        return 'few-channel layer ends (split precision)'
    if name.startswith('risp_conv') or name.startswith('risp_rect_sums') or name.startswith('risp_srcnn') or name == 'risp_group_sum':
        return 'other convolution launches (fp32 kernels, sums for the constant planes)'
    if name.startswith(('risp_slot', 'risp_mix', 'risp_chain', 'risp_prune', 'risp_param_blocks')) or name.endswith(('_fwd', '_bwd')):
        return 'slot mixtures and element-wise operators'
    return 'step glue (losses, virtual step, optimizers, reductions)'


def search_step_times(device, rank, world, distributed, global_batch, size, n_step, iters, counts=None):
    """Seconds per DARTS iteration (train.py's loop body: feed_data, update_learning_rate, optimize_alphas,
    optimize_parameters) on this rank's shard of the global batch, and the seconds of it spent inside the gradient
    all-reduces (a second set of iterations with the collectives bracketed by device synchronisations).  ``counts``: a dict
    that receives, from one more iteration, the matrix FLOPs its launches issue and its C-ABI calls."""
    from reconfigisp_amd.codes.models import create_model
    from reconfigisp_amd.codes.data.synthetic_raw import make_batch
    torch.manual_seed(10)
    model = create_model(search_opt(n_step, distributed))
    a, ga = make_batch(global_batch, size, size, seed=1)
    b, gb = make_batch(global_batch, size, size, seed=2)
    shard = slice(rank, None, world) if distributed else slice(None)       # data_sampler.py:88-99: indices[rank::world]
    data = tuple(t[shard].to(device) for t in (a, ga, b, gb))

    def step(i):
        model.feed_data(data)
        model.update_learning_rate(i, warmup_iter=-1)
        model.optimize_alphas()
        model.optimize_parameters()

    def run(k0):
        torch.cuda.synchronize(device)
        if distributed and world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(iters):
            step(k0 + i)
        torch.cuda.synchronize(device)
        if distributed and world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if distributed and world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt / iters

    step(0)
    step(1)
    sec = run(2)
    comm = 0.0
    if distributed and world > 1:
        model.comm_seconds = 0.0
        run(2 + iters)
        comm, model.comm_seconds = model.comm_seconds / iters, None
    loss = model.log_dict['loss']
    per_rank = data[0].shape[0]
    if counts is not None:
        import reconfigisp_amd.convnets as CN
        from reconfigisp_amd import lib as L
        CN.MFMA_ISSUED, CN.MFMA_ISSUED_F16, L.CALLS = [0.0], [0.0], {}
        try:
            step(2 + 2 * iters)
            torch.cuda.synchronize(device)
            counts.update(f32_flop=CN.MFMA_ISSUED[0], f16_flop=CN.MFMA_ISSUED_F16[0], c_abi_calls=sum(L.CALLS.values()))
        finally:
            CN.MFMA_ISSUED = CN.MFMA_ISSUED_F16 = L.CALLS = None
        if counts.get('families'):
            # one more iteration with every C-ABI call between two events on the stream it is issued to: kernel time by family
            marks, real = [], L.call

            def timed(name, *a):
                st = torch.cuda.current_stream(device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                real(name, *a)
                e1.record(st)
                marks.append((_family(name, a), e0, e1))

            from reconfigisp_amd.codes.models.modules import super_prune_fifteen_demos_four_bayer_two as SP
            L.call, streams, SP.SLOT_STREAMS = timed, SP.SLOT_STREAMS, 1      # (one stream: intervals on two streams would overlap)
            try:
                step(3 + 2 * iters)
                torch.cuda.synchronize(device)
            finally:
                L.call, SP.SLOT_STREAMS = real, streams
            fam = {}
            for f, e0, e1 in marks:
                fam[f] = fam.get(f, 0.0) + e0.elapsed_time(e1)
            counts['family_ms'] = fam
    del model, data
    torch.cuda.empty_cache()
    return sec, comm, loss, per_rank


def search_step_leg(device, rank, world, global_batch=32, size=256, n_step=2, iters=2):
    """BASELINE config 4 (train.py DDP 4-slot search): a fixed global batch sharded over the ranks."""
    pix = 2 * global_batch * size * size                       # train + val pixels of one iteration
    out = {'workload': 'DARTS iteration (optimize_alphas + optimize_parameters: 5 forwards + 5 backwards of the %d-slot '
                       'super-net, n_step %d), global batch %d train + %d val %dx%d patches, alpha = 0, nothing pruned'
                       % (n_step + 2, n_step, global_batch, global_batch, size, size), 'scaling': 'strong'}
    if world > 1:
        sec, comm, loss, per_rank = search_step_times(device, rank, world, True, global_batch, size, n_step, iters)
        out.update(n_gpus=world, per_rank_batch=per_rank, s_per_step=round(sec, 4), steps_per_s=round(1.0 / sec, 3),
                   MPix_s=round(pix / sec / 1e6, 2), allreduce_s_per_step=round(comm, 5),
                   allreduce_calls_per_step=4, loss_rank0=round(loss, 6))
    # the same job on ONE GPU (the ratio's denominator).  EVERY rank runs it, each on its own GPU and after the
    # distributed leg, so no rank waits in a collective while another computes (rank 0's figure is the one reported)
    sec1, _, loss1, _ = search_step_times(device, 0, 1, False, global_batch, size, n_step, iters)
    one = {'s_per_step': round(sec1, 4), 'MPix_s': round(pix / sec1 / 1e6, 2), 'loss': round(loss1, 6)}
    if world > 1:
        out.update(one_gpu=one, speedup_vs_one_gpu=round(one['s_per_step'] / out['s_per_step'], 3))
    else:
        out.update(n_gpus=1, per_rank_batch=global_batch, steps_per_s=round(1.0 / one['s_per_step'], 3), **one)
        if global_batch % 8 == 0 and global_batch >= 8:
            # what ONE rank of the 8-GPU configuration computes per iteration (its shard of the same global batch), timed on
            # this GPU without collectives: the compute-side ceiling of the 8-GPU speed-up, NOT a measurement of it
            sec8, _, _, _ = search_step_times(device, 0, 1, False, global_batch // 8, size, n_step, max(iters, 5))
            out['rank_of_8_shard'] = {'batch': global_batch // 8, 's_per_step': round(sec8, 4),
                                      'compute_ceiling_of_8_gpu_speedup': round(sec1 / sec8, 2),
                                      'note': 'one GPU, no collectives: an upper bound, not an 8-GPU measurement'}
    return out


def config3_leg(device, batch=32, size=256, n_step=3, iters=2):
    """BASELINE config 3: the 5-slot search step (n_step 3) at batch 32 on one GPU - what train.py:162,220 prints as 'Average time
    per iter'.  The matrix pipes' share: FLOPs the step's launches issue over the step time, f16 against the nameplate and against
    what the pipe sustains on random halves under the chip's power limit (profiles/r04_mfma_f16_power.txt)."""
    c = {'families': True}
    sec, _, loss, _ = search_step_times(device, 0, 1, False, batch, size, n_step, iters, counts=c)
    pix = 2 * batch * size * size
    total = sum(c['family_ms'].values()) or 1.0
    tf16, tf32 = c['f16_flop'] / sec / 1e12, c['f32_flop'] / sec / 1e12
    return {'workload': 'DARTS iteration, 5-slot super-net (n_step %d, prune 0.2, alpha = 0), batch %d train + %d val %dx%d'
                        % (n_step, batch, batch, size, size),
            's_per_step': round(sec, 4), 'MPix_s': round(pix / sec / 1e6, 2), 'loss': round(loss, 6),
            'c_abi_calls_per_step': c['c_abi_calls'],
            'mfma_f16_issued_TFLOPs': round(tf16, 1), 'mfma_f32_issued_TFLOPs': round(tf32, 2),
            'mfma_f16_issued_of_nameplate': round(tf16 / MFMA_F16_PEAK_TFLOPS, 4),
            'mfma_f16_issued_of_sustained': round(tf16 / MFMA_F16_SUSTAINED_TFLOPS, 4),
            'mfma_f32_issued_of_peak': round(tf32 / MFMA_F32_PEAK_TFLOPS, 4),
            'kernel_ms_per_step_between_events': round(total, 1),
            'kernel_time_share': {k: round(v / total, 3) for k, v in sorted(c['family_ms'].items(), key=lambda kv: -kv[1])},
            'note': 'issued = FLOPs of the matrix instructions the launches execute over the WHOLE step time (element-wise launches, '
                    'reductions and host gaps included); nameplate %.1f, sustained %.0f TFLOP/s; kernel_time_share: one more iteration with every C-ABI call between '
                    'two events on its stream (kernel by kernel: profiles/r06_config3_darts_step_kernel_stats.txt)' % (MFMA_F16_PEAK_TFLOPS, MFMA_F16_SUSTAINED_TFLOPS)}


def shipped_search_leg(device, rank=0, world=1, iters=10):
    """The search iteration at the geometry the reference ships (options/train/SID_search.yml:16-17,31-34, S7ISP_search.yml: batch 4,
    48 x 48 crops, n_step 3, prune 0.2; README.md:10 launches it on 4 ranks): launch-bound - iterations per second and C-ABI calls
    per iteration.  N > 1: the global batch of 4 sharded over the ranks (8 ranks: 8, one crop each - data/__init__.py:15 wants
    batch %% world == 0), with the four gradient all-reduces."""
    gb = 4 if 4 % world == 0 else world
    c = {}
    sec, comm, loss, per_rank = search_step_times(device, rank, world, world > 1, gb, 48, 3, iters, counts=c)
    out = {'workload': 'DARTS iteration, 5-slot super-net (n_step 3, prune 0.2), global batch %d train + %d val 48x48 (SID_search.yml)' % (gb, gb),
           'n_gpus': world, 'per_rank_batch': per_rank, 's_per_iter': round(sec, 5), 'iters_per_s': round(1.0 / sec, 2),
           'c_abi_calls_per_iter': c['c_abi_calls'], 'loss_rank0': round(loss, 6)}
    if world > 1:
        out['allreduce_s_per_iter'] = round(comm, 6)
    return out


def config5_leg(device, tile_batch=21, reps=3):
    """BASELINE config 5: test_split.py on one 4000 x 3000 frame (patch 512 / stride 480 -> 63 tiles, SID_test.yml:18-19) through
    Bayer_01_Demosaic_02_sRGB_13; MPix/s on the 12 MPix frame.  First frame (packs built, buffers allocated) and steady state."""
    import contextlib
    from collections import OrderedDict
    from reconfigisp_amd.codes.models import create_model
    from reconfigisp_amd.codes.test_split import run_frame as run_frame_
    def run_frame(*a):                                           # (test_split.py prints 'Split into 63 patches': the line is stdout's only one)
        with contextlib.redirect_stdout(sys.stderr):
            return run_frame_(*a)
    H, W = 3000, 4000
    opt = OrderedDict(model='isp', gpu_ids=[0], dist=False, is_train=False,
                      network_G=dict(which_model_G='IspUniversal', architecture='Bayer_01_Demosaic_02_sRGB_13',
                                     individual_module_paths=[None] * 3, module_path=None),
                      path=dict(pretrain_model_G=None, strict_load=True))
    torch.manual_seed(10)
    model = create_model(opt)
    g = torch.Generator().manual_seed(1)
    frame = (torch.randint(0, 1024, (1, 1, H, W), generator=g).float() / 1023.).to(device)
    torch.cuda.synchronize(device)
    t = time.perf_counter()
    out = run_frame(model, frame, (512, 512), (480, 480), tile_batch)
    torch.cuda.synchronize(device)
    first = time.perf_counter() - t
    run_frame(model, frame, (512, 512), (480, 480), tile_batch)
    torch.cuda.synchronize(device)
    t = time.perf_counter()
    for _ in range(reps):
        out = run_frame(model, frame, (512, 512), (480, 480), tile_batch)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t) / reps
    flop = 239680.0 * 63 * 512 * 512
    res = {'workload': 'test_split: 1x1x3000x4000 RAW, 63 tiles of 512 (stride 480), tile batch %d, Bayer_01_Demosaic_02_sRGB_13' % tile_batch,
           'ms_per_frame': round(dt * 1e3, 2), 'first_frame_ms': round(first * 1e3, 1), 'MPix_s_on_12MPix_frame': round(H * W / dt / 1e6, 1),
           'effective_TFLOPs': round(flop / dt / 1e12, 1), 'finite': bool(torch.isfinite(out).all())}
    del model, frame, out
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000,
                    help='timed steps (default 2000 = ~0.1 s: right after an idle period the device spends some 100 '
                         'launches in a boost-then-clamp power transient, tools/diag_ramp.py)')
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--batch', type=int, default=64, help='patches per GPU per step')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--launch', choices=('stream', 'graph'), default='stream',
                    help="stream: every step is a Python call of the registry forward (one C-ABI launch on the stream; "
                         "measured faster than hipGraph on this stack, whose kernel nodes are ~9 us apart); "
                         "graph: hipGraph replay of `--queue` forwards")
    ap.add_argument('--eager', action='store_true', help='alias of --launch stream')
    ap.add_argument('--queue', type=int, default=4, help='resident batches the steps rotate over, each with its own '
                                                          'stage-output buffers; a step is ONE forward over ONE batch')
    ap.add_argument('--no-cnn', action='store_true', help='skip the MFMA-bound reference-YAML pipeline')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--no-search', action='store_true', help='skip the distributed search-step leg (BASELINE config 4)')
    ap.add_argument('--search-batch', type=int, default=32, help='GLOBAL batch of the search-step leg')
    ap.add_argument('--no-configs', action='store_true', help='skip BASELINE configs 3 / 5 and the shipped search geometry')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    one_device = os.environ.get('RISP_BENCH_ONE_DEVICE') == '1'
    if os.environ.get('RISP_BENCH_LAUNCH_ONLY') == '1':
        # launcher self-test (tests/test_bench_launcher_cpu.py, no GPU needed): rendezvous, one collective, one line
        dist.init_process_group(backend='gloo')
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({'metric': 'launch-only self-test', 'value': None, 'n_gpus': world, 'launch_only': True,
                              'rank_sum': t.item()}))
        dist.destroy_process_group()
        return
    if world > 1:
        # RCCL on ROCm.  RISP_BENCH_BACKEND=gloo + RISP_BENCH_ONE_DEVICE=1 is a dry-run mode for boxes with a
        # single GPU (every rank on device 0) used only to exercise this code path.
        dist.init_process_group(backend=os.environ.get('RISP_BENCH_BACKEND', 'nccl'))
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    from reconfigisp_amd.codes.data.synthetic_raw import make_batch
    bay_cpu, _ = make_batch(args.batch, args.size, args.size, seed=10 + rank)
    bay = bay_cpu.to(device)
    pix_per_step = args.batch * args.size * args.size

    from reconfigisp_amd.graphs import GraphedForward, GraphedQueue
    net = build_pipeline(ARCH_DENOISE, device, 'OriginUniversal')
    graph = args.launch == 'graph' and not args.eager
    queue = max(1, args.queue)
    if graph and (args.steps % queue or args.warmup % queue):
        queue = 1
    # `queue` different batches resident in HBM, each with its own stage-output buffers
    batches = [bay] + [make_batch(args.batch, args.size, args.size, seed=100 + 10 * rank + k)[0].to(device)
                       for k in range(1, queue)]

    def rotating(fwd):          # step k runs the registry forward on resident batch k mod queue
        state = [0]
        def step():
            fwd(batches[state[0] % queue])
            state[0] += 1
        return step

    with torch.no_grad():
        if graph:               # one replay = `queue` steps (forwards), each on its own resident batch
            gq = GraphedQueue(net, batches)
            wall, dev_ms = timed(lambda: gq(), args.steps // queue, args.warmup // queue, device, world)
            dev_ms /= queue
        else:
            wall, dev_ms = timed(rotating(net), args.steps, args.warmup, device, world)
        host_us = timed.host_us / (queue if graph else 1)
        spin_up = timed.spin_up_launches * (queue if graph else 1)
        b2b_ms = kernel_time_ms(net, batches[:queue], max(args.steps, 100), device, True)
        b2b_ms_cached = kernel_time_ms(net, batches[:1], max(args.steps, 100), device, True)
    value = world * pix_per_step * args.steps / wall / 1e6
    ms_per_step = wall / args.steps * 1e3
    # The dominant kernel's average launch duration = HIP events on the launch stream over the SAME timed region as
    # `value` (a step is exactly one launch of it), i.e. an upper bound of the kernel's own duration that includes
    # the launch boundary.  By construction it cannot exceed ms_per_step (the events sit inside the wall bracket).
    kernel_ms = dev_ms
    kernel_within_step = bool(kernel_ms <= ms_per_step * 1.001)      # reported, not asserted: a diagnostic must not cost the line
    achieved = BYTES_PER_PIX_ISP * pix_per_step / (kernel_ms * 1e-3) / 1e9

    pw = build_pipeline(ARCH_HBM, device)
    with torch.no_grad():
        pstep = GraphedForward(pw, bay) if graph else rotating(pw)
        wall_p, dev_ms_p = timed(lambda: pstep(), args.steps, args.warmup, device, world)
        host_us_p = timed.host_us
        kernel_ms_p = kernel_time_ms(pw, batches[:queue], max(args.steps, 100), device, False)
    hbm = lambda bpp, ms: bpp * pix_per_step / (ms * 1e-3) / 1e9          # GB/s
    extra = {'kernel_ms': round(kernel_ms, 5), 'kernel_ms_within_step': kernel_within_step, 'kernel_ms_source': 'HIP events on the launch stream around the timed region / steps',
             'host_issue_us_per_step': round(host_us, 1),
             'resident_batches': queue,
             'kernel_ms_back_to_back_c_abi': round(b2b_ms, 5),
             'hbm_frac_back_to_back_c_abi': round(hbm(BYTES_PER_PIX_ISP, b2b_ms) / HBM_PEAK_GBS, 4),
             'kernel_ms_one_batch_cache_assisted': round(b2b_ms_cached, 5),
             'hbm_frac_one_batch_cache_assisted': round(hbm(BYTES_PER_PIX_ISP, b2b_ms_cached) / HBM_PEAK_GBS, 4),
             'spin_up_launches': spin_up,       # untimed, before the W warm-up steps: ~50 ms of the same launches (see timed())
             'launch': ('hipGraph replay, %d resident batches (steps) per replay' % queue) if graph else
                       ('one Python forward() per step on the stream, rotating over %d resident batches' % queue),
             'pointwise_arch': ARCH_HBM,
             'pointwise_MPix_s': round(world * pix_per_step * args.steps / wall_p / 1e6, 1),
             'pointwise_kernel_ms': round(dev_ms_p, 5), 'pointwise_kernel_ms_back_to_back_c_abi': round(kernel_ms_p, 5),
             'pointwise_host_issue_us_per_step': round(host_us_p, 1), 'pointwise_host_issue_us_med_max': [round(timed.host_med_us, 1), round(timed.host_max_us, 1)],
             'pointwise_hbm_GBs': round(hbm(BYTES_PER_PIX_FUSED, dev_ms_p), 1),
             'pointwise_hbm_frac': round(hbm(BYTES_PER_PIX_FUSED, dev_ms_p) / HBM_PEAK_GBS, 4)}
    if not args.no_cnn:
        import reconfigisp_amd.convnets as CN
        cnn = build_pipeline(ARCH_CNN, device)
        steps_c = max(3, args.steps // 20)
        cstep = GraphedForward(cnn, bay) if graph else (lambda: cnn(bay))
        def cnn_leg(arith):
            # the wide 3x3 layers in split precision on the f16 matrix pipe (the product's default) or on the fp32 matrix cores
            keep, CN.CONV_ARITH = CN.CONV_ARITH, arith
            try:
                with torch.no_grad():
                    cstep()                           # (a graphed forward was captured with the default arithmetic)
                    wall, dev_ms = timed(lambda: cstep(), steps_c, 2, device, world)
                    CN.MFMA_ISSUED, CN.MFMA_ISSUED_F16 = [0.0], [0.0]      # one more forward, counting the FLOPs its launches issue
                    cnn(bay)
                    f32, f16 = CN.MFMA_ISSUED[0], CN.MFMA_ISSUED_F16[0]
            finally:
                CN.MFMA_ISSUED = CN.MFMA_ISSUED_F16 = None
                CN.CONV_ARITH = keep
            return wall, dev_ms, f32, f16
        wall_c, dev_ms_c, issued, issued16 = cnn_leg(CN.CONV_ARITH)
        tf = lambda flop, ms: flop / (ms * 1e-3) / 1e12
        extra.update(cnn_arch=ARCH_CNN, cnn_MPix_s=round(world * pix_per_step * steps_c / wall_c / 1e6, 1),
                     cnn_ms_per_step=round(dev_ms_c, 3),
                     cnn_arith=('2 x f16 split (3 products, f32 accumulate) for the wide 3x3 / 5x5 layers and, on rows of >= 192 pixels, the 9x9 / 5x5 few-channel ends; f32 elsewhere'
                                if CN.CONV_ARITH == 'f16x2' else 'f32'),
                     cnn_effective_TFLOPs=round(tf(FLOP_PER_PIX_CNN * pix_per_step, dev_ms_c), 2),
                     cnn_mfma_issued_TFLOPs=round(tf(issued, dev_ms_c), 2),
                     cnn_mfma_f16_issued_TFLOPs=round(tf(issued16, dev_ms_c), 2),
                     cnn_mfma_issued_frac=round(tf(issued, dev_ms_c) / MFMA_F32_PEAK_TFLOPS + tf(issued16, dev_ms_c) / MFMA_F16_PEAK_TFLOPS, 4),
                     cnn_note='effective = direct-convolution FLOPs (SURVEY 8d) / time (exceeds the fp32 matrix peak where Winograd or the '
                              'f16 pipe do the work: a rate, not a utilisation); issued = FLOPs of the MFMA '
                              'instructions the launches execute (Winograd halves the fp32 3x3 layers, cout padded to 32; the '
                              'split-precision layers issue 3 f16 products per tap; the small-cout layers run on vector FMAs '
                              'and count 0); cnn_mfma_issued_frac = f32 issued / %.1f + f16 issued / %.1f TFLOP/s: the share of the '
                              'time the matrix pipes are busy at their nameplate rates' % (MFMA_F32_PEAK_TFLOPS, MFMA_F16_PEAK_TFLOPS))
        if CN.CONV_ARITH != 'f32' and not graph:
            wall_f, dev_ms_f, issued_f, _ = cnn_leg('f32')
            extra.update(cnn_f32_MPix_s=round(world * pix_per_step * steps_c / wall_f / 1e6, 1), cnn_f32_ms_per_step=round(dev_ms_f, 3),
                         cnn_f32_mfma_issued_frac=round(tf(issued_f, dev_ms_f) / MFMA_F32_PEAK_TFLOPS, 4))
    if not args.no_search:
        # a secondary leg must never cost the headline line: at N = 1 a failure is reported instead of raised.  At N > 1 a
        # failure can be rank-local (out of memory on one GPU, an RCCL error) while the peers sit in a collective: the ranks
        # cannot agree on anything any more, so the exception goes up and the launcher ends the job (launch_ranks also
        # carries a time limit that kills the child tree).
        try:
            leg = search_step_leg(device, rank, world, global_batch=args.search_batch, size=args.size)
        except Exception as e:                                  # noqa: BLE001
            if world > 1:
                raise
            leg = {'error': '%s: %s' % (type(e).__name__, e)}
        if rank == 0:
            extra['search_step'] = leg
        try:
            leg = shipped_search_leg(device, rank, world)
        except Exception as e:                                  # noqa: BLE001
            if world > 1:
                raise
            leg = {'error': '%s: %s' % (type(e).__name__, e)}
        if rank == 0:
            extra['search_step_shipped'] = leg
    if not args.no_configs:
        # BASELINE configs 3 and 5, driver-run (every rank runs them on its own GPU, after the collectives of the legs above: no
        # rank waits in a collective meanwhile; rank 0's figures are reported)
        for key, fn in (('config3', config3_leg), ('config5', config5_leg)):
            try:
                leg = fn(device)
            except Exception as e:                              # noqa: BLE001
                leg = {'error': '%s: %s' % (type(e).__name__, e)}
            if rank == 0:
                extra[key] = leg

    proof = None
    if world > 1:
        # Proof of ranks: what each rank ran on, gathered over the process group itself - and, measured in the SAME run, what one
        # rank alone delivers while the others wait in a barrier (the N = 1 figure beside the N-rank one; the scaling efficiency
        # is the reader's to compute).
        props = torch.cuda.get_device_properties(device)
        me = {'rank': rank, 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'device_index': local,
              'device_name': props.name, 'pci_bus_id': '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), getattr(props, 'pci_bus_id', 0),
                                                                         getattr(props, 'pci_device_id', 0)),
              'uuid': str(getattr(props, 'uuid', '')), 'value_MPix_s': round(pix_per_step * args.steps / wall / 1e6, 1)}
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        solo = None
        if rank == 0:
            with torch.no_grad():
                wall_1, _ = timed(rotating(net), args.steps, args.warmup, device, 1)
            solo = round(pix_per_step * args.steps / wall_1 / 1e6, 1)
        dist.barrier()
        proof = {'ranks': ranks, 'backend': dist.get_backend(), 'world_as_seen_by_pg': dist.get_world_size(),
                 'distinct_devices': len({r['pci_bus_id'] + r['uuid'] for r in ranks}),
                 'one_rank_alone_same_run_MPix_s': solo}
    if rank == 0:
        line = {
            'metric': 'MPix/s end-to-end 5-stage ISP forward, 256x256 Bayer', 'value': round(value, 1),
            'unit': 'MPix/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 5), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'batch=%d %dx%d Bayer per GPU, 5-stage fixed ISP forward %s (nearest demosaic, '
                                   'bilateral denoise, WbManual, Gamma, GtmManual; OriginUniversal), all stage '
                                   'outputs materialised' % (args.batch, args.size, args.size, ARCH_DENOISE),
                       'global_batch': args.batch * world, 'parallelism': 'batch-sharded x%d, no collective' % world},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': measured_traffic(BYTES_PER_PIX_ISP * pix_per_step),
                         'kernel': '%s (risp_bilateral_chain_fwd): one launch per step, %d B/pix algorithmic'
                                   % (launched_kernel(), BYTES_PER_PIX_ISP)},
            'extra': extra,
        }
        if proof is not None:
            line['proof_of_ranks'] = proof
        if one_device and world > 1:
            line['dry_run_all_ranks_on_one_device'] = True
        if not args.no_cpu and world == 1:
            line['cpu_baseline'] = cpu_baseline(bay_cpu, ARCH_DENOISE)
            line['cpu_baseline_config1'] = cpu_baseline_config1(args.size)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
