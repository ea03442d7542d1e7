#!/usr/bin/env python
"""bench.py — SR output megapixels/s at x4 (128 -> 512) on N MI355X (BASELINE.json metric).

A step = one pass of the hot path (`FeMaSRNet.test`) over one batch of synthetic input already resident
in HBM: BASELINE config[1] — x4, batch 16 of 128x128 LR tiles per GPU, random-init (synthetic) weights.
Weak scaling: every rank processes its own 16 tiles per step (tiles are independent units); with N > 1
the step also contains the path's one real exchange, the RCCL all-gather of the upscaled tiles that
precedes the paste (femasr_amd/distributed.py), unless --no-gather.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant kernel,
HIP-event timed on the launch stream inside the timed region) and `cpu_baseline` (the CPU oracle timed
on a bounded sample on this box's host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PMC_TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')   # from tools/rocpd_pmc_summary.py (rocprofv3 --pmc passes)
PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0       # same guide: bf16 MFMA dense peak (the 5 PF headline includes 2:1 sparsity)
TILE_GFLOP = 964.47                  # algorithmic GFLOP per x4 128^2 tile (SURVEY 8d / BASELINE.md 3)


def rocprof_kernel_name(bench_name):
    """'conv3x3_halo<8x16x128,FEMASR_PRO_GN_SILU,up2=false,waves=4x2>' -> 'conv3x3_halo_kernel<128, 4, 2, 1, false>'."""
    import re
    pro = {'FEMASR_PRO_NONE': 0, 'FEMASR_PRO_GN_SILU': 1, 'FEMASR_PRO_LN': 2}
    m = re.match(r'conv3x3_halo<8x16x(\d+),(\w+),up2=(\w+),waves=(\d)x(\d)>', bench_name)
    if m:
        return f'conv3x3_halo_kernel<{m.group(1)}, {m.group(4)}, {m.group(5)}, {pro[m.group(2)]}, {m.group(3)}>'
    m = re.match(r'conv3x3_halo_bf16x3<8x16x(\d+),(\w+),up2=(\w+),waves=(\d)x(\d)>', bench_name)
    if m:
        return f'conv3x3_halo_bf16x3_kernel<{m.group(1)}, {m.group(4)}, {m.group(5)}, {pro[m.group(2)]}, {m.group(3)}>'
    m = re.match(r'conv_igemm<(\d+)x(\d+),(\w+),cinvec=(\w+),vq=(\w+),k1=(\w+),waves=(\d)x(\d)>', bench_name)
    if m:
        return (f'conv_igemm_kernel<{m.group(1)}, {m.group(2)}, {m.group(7)}, {m.group(8)}, {pro[m.group(3)]}, '
                f'{m.group(4)}, {m.group(5)}, {m.group(6)}>')
    return bench_name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16, help='128x128 LR tiles per GPU per step')
    ap.add_argument('--streams', type=int, default=2, help='sub-batch streams inside one forward (femasr_set_streams)')
    ap.add_argument('--profile-steps', type=int, default=2, help='extra serialized steps (streams=1) for the roofline object')
    ap.add_argument('--decoder-math', choices=['fp32', 'bf16x3'], default='bf16x3',
                    help="'bf16x3': convs behind the VQ lookup on the bf16 matrix cores (3-term split, within 1e-3)")
    ap.add_argument('--no-gather', action='store_true', help='N>1: skip the all-gather of upscaled tiles')
    ap.add_argument('--force-gather', action='store_true',
                    help='N=1: still run the all-gather path through a one-rank RCCL group (exercises the N>1 code on one GPU)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-exact-leg', action='store_true', help='skip the extra all-fp32 (bit-exact mode) timing')
    ap.add_argument('--no-profile', action='store_true', help='do not record per-kernel HIP events')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from femasr_amd import distributed as fd
    from femasr_amd import synth
    from helpers import synth_weights
    import gpu_utils as G

    rank, world, local = fd.init_from_env()
    assert world == args.gpus or world == 1 and args.gpus == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    weights = synth_weights('x4', 0, 'trained')
    net = G.build_net('x4', weights, dev)
    net.num_streams = args.streams
    net.decoder_math = args.decoder_math
    B = args.batch
    x = torch.from_numpy(synth.synth_input(1000 + rank, (B, 3, 128, 128))).to(dev)
    if args.force_gather and world == 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29577')
        dist.init_process_group(backend='nccl', rank=0, world_size=1)
    use_pg = world > 1 or args.force_gather
    do_gather = use_pg and not args.no_gather
    # The all-gather of step k runs on RCCL's stream while step k+1 computes (double-buffered receive lists): the
    # upscaled tiles of a step are only consumed by the paste, so a serving loop pipelines exactly like this.
    gathered = [[torch.empty((B, 3, 512, 512), dtype=torch.float32, device=dev) for _ in range(world)] for _ in range(2)] \
        if do_gather else None
    pending = []            # (work handle, tensors kept alive until the collective has run)
    nstep = [0]

    def step():
        y = net.test(x)
        if do_gather:
            w = dist.all_gather(gathered[nstep[0] & 1], y, async_op=True)
            pending.append((w, y))
            if len(pending) > 1:            # the buffer set about to be re-used next step must be free
                pending.pop(0)[0].wait()
        nstep[0] += 1
        return y

    def fence():
        while pending:
            pending.pop(0)[0].wait()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # Per-kernel roofline: HIP events around every launch on the launch stream.  With >1 sub-batch streams kernels
    # of different streams overlap and a per-kernel duration is not separable, so the events are recorded in extra
    # SERIALIZED steps (streams=1) run right after the timed region, on rank 0 only.
    prof = {}
    if rank == 0 and not args.no_profile and args.profile_steps > 0:
        net.num_streams = 1
        net.test(x)
        torch.cuda.synchronize(dev)
        net.enable_profile(True)
        tp0 = time.perf_counter()
        for _ in range(args.profile_steps):
            net.test(x)
        torch.cuda.synchronize(dev)
        prof_ms_per_step = (time.perf_counter() - tp0) / args.profile_steps * 1e3
        prof = net.profile()
        net.enable_profile(False)
        net.num_streams = args.streams
    net.decoder_math = args.decoder_math
    assert torch.isfinite(y).all()

    out_mpix = world * B * 512 * 512 / 1e6
    value = out_mpix * args.steps / dt
    res = {
        'metric': 'SR output megapixels/sec at x4 (128->512)', 'value': round(value, 4), 'unit': 'MPix/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.decoder_math == 'fp32' else 'f32 for everything feeding the VQ argmin + bf16x3 (3-pass split-bf16 MFMA, fp32 accumulate) for the 3x3 convs that do not',
        'data': 'synthetic',
        'config': {'workload': f'x4 SR FeMaSRNet.test, batch {B} of 128x128 LR tiles per GPU -> 512x512 (padded 144->576 '
                               'inside, reference geometry), synthetic random-init weights (seed 0), inputs resident in HBM',
                   'global_batch': B * world, 'tile': '128x128->512x512', 'parallelism': f'tile-parallel x{world}',
                   'gather': bool(do_gather), 'gather_overlap': 'all-gather of step k overlaps step k+1', 'streams': args.streams, 'decoder_math': args.decoder_math,
                   'algorithmic_gflop_per_tile': TILE_GFLOP,
                   'end_to_end_tflops': round(TILE_GFLOP * B * world * args.steps / dt / 1e3, 2)},
    }
    if rank == 0:
        if prof:
            convs = {k: v for k, v in prof.items() if k.startswith('conv')}      # the MFMA kernels
            psteps = args.profile_steps
            pmc = json.load(open(PMC_TRAFFIC_JSON)) if os.path.exists(PMC_TRAFFIC_JSON) else {}

            def roof(name):
                ms, n, fl, _ = convs[name]
                split = name.startswith('conv3x3_halo_bf16x3')
                peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
                ach = fl / (ms * 1e-3) / 1e12
                rec = pmc.get(rocprof_kernel_name(name))
                out = {'bound': 'mfma', 'kernel': name, 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                       'frac': round(ach / peak, 4),
                       'traffic': round((rec['fetch_bytes_corrected'] + rec['write_bytes']) / 1e9, 4) if rec else None,
                       'launches': n, 'avg_launch_ms': round(ms / n, 4), 'gflop_per_launch': round(fl / n / 1e9, 3),
                       'peak_basis': ('bf16 dense MFMA peak; ALGORITHMIC flops: each multiply-add costs 3 MFMA passes '
                                      '(hi*hi + hi*lo + lo*hi), so MFMA issue fraction = 3 x frac') if split else
                                     'fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32)'}
                if split:
                    out['mfma_issue_frac'] = round(3 * ach / peak, 4)
                if rec:
                    out['traffic_source'] = ('GB per launch, rocprofv3 --pmc FETCH_SIZE (doubled per MI355X_MICROARCH.md HBM '
                                             'section) + WRITE_SIZE passes of this command (profiles/pmc_traffic.json)')
                return out
            dom = max(convs, key=lambda k: convs[k][0])
            res['roofline'] = roof(dom)
            fp32k = [k for k in convs if not k.startswith('conv3x3_halo_bf16x3')]
            if fp32k and dom not in fp32k:
                res['roofline']['largest_fp32_kernel'] = roof(max(fp32k, key=lambda k: convs[k][0]))
            tot_ms = sum(v[0] for v in convs.values())
            tot_fl = sum(v[2] for v in convs.values())
            res['roofline']['all_mfma_conv_kernels'] = {
                'achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                'share_of_serialized_step_time': round(tot_ms / psteps / prof_ms_per_step, 4)}
            res['roofline']['measured_in'] = (f'{psteps} serialized steps (streams=1, {prof_ms_per_step:.1f} ms/step) right after '
                                              'the timed region')
            res['roofline']['per_kernel_ms_per_step'] = {k: round(v[0] / psteps, 3)
                                                         for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        if world == 1 and args.decoder_math != 'fp32' and not args.no_exact_leg:
            # the same workload with EVERY layer in exact fp32 (the mode the parity tests check bit-for-bit)
            net.decoder_math = 'fp32'
            net.test(x)
            torch.cuda.synchronize(dev)
            te0 = time.perf_counter()
            for _ in range(args.steps):
                ye = net.test(x)
            torch.cuda.synchronize(dev)
            te = (time.perf_counter() - te0) / args.steps
            res['exact_fp32_mode'] = {'value': round(B * 512 * 512 / 1e6 / te, 4), 'unit': 'MPix/s', 'ms_per_step': round(te * 1e3, 3),
                                      'max_abs_vs_timed_mode': float((ye - y).abs().max())}
            net.decoder_math = args.decoder_math
        if world == 1 and not args.no_cpu_baseline:
            from helpers import oracle_net
            onet = oracle_net('x4', weights)
            xs = x[:1].cpu().numpy()
            t1 = time.perf_counter()
            yo = onet.test(xs)
            tc = time.perf_counter() - t1
            res['cpu_baseline'] = {
                'value': round(512 * 512 / 1e6 / tc, 5), 'unit': 'MPix/s', 'cores': os.cpu_count(), 'kind': 'port',
                'sample': f'1 of the {B} tiles of one step (x4 128x128->512x512, {TILE_GFLOP} GFLOP) through oracle/ '
                          f'(C, OpenMP, fp32 fmaf) in {tc:.1f} s',
                'max_abs_vs_gpu': float(np.abs(yo - y[:1].cpu().numpy()).max()),
            }
        try:        # RCCL's banner goes through C stdio: flush it first so that the JSON line is the LAST line on stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
