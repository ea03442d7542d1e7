#!/usr/bin/env python
"""bench.py — SR output megapixels/s at x4 (128 -> 512) on N MI355X (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic input already resident in HBM.

  --workload tiles16 (default; BASELINE config[1], the configuration the metric is quoted on):
      `FeMaSRNet.test` on 16 tiles of 128x128 per GPU.  WEAK scaling: every rank processes its own 16 tiles per
      step (tiles are independent units); with N > 1 the step also contains the path's one real exchange, the RCCL
      all-gather of the upscaled tiles that precedes the paste (femasr_amd/distributed.py), unless --no-gather.
  --workload tile2048 (BASELINE config[2]/3a): ONE 2048x2048 LR image -> `test_tile(tile_size=128, tile_pad=0)`,
      256 tiles sharded over the ranks, all-gather + paste into the 8192x8192 canvas INSIDE the timed region.
      STRONG scaling (total work fixed).

The arithmetic of record is fp32 (`decoder_math='fp32'`: every layer on v_mfma_f32_32x32x2_f32; `fp32_strict` is bit-identical
to the CPU oracle, `dtype: "f32"`).  The split-bf16 mode (3-pass hi/lo products for the convs that do not feed the
VQ argmin; within the 1e-3 bound but narrower products than fp32) is timed in the same run and reported under the
secondary key `bf16x3_mode` — never as `value`.

`python bench.py --gpus N` with WORLD_SIZE unset launches the N ranks ITSELF (re-exec under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); under torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  `--backend gloo --dry-net` runs the same launch /
barrier / gather / paste code with a stand-in network on CPU (tests/test_bench_launch.py).

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant kernel by summed time,
HIP-event timed on the launch stream) and `cpu_baseline` (stock-torch CPU restatement and the C oracle timed on a
bounded sample on this box's host cores; N=1 only).
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

PMC_TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')   # from tools/rocpd_pmc_summary.py (rocprofv3 --pmc passes)
PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0       # same guide: bf16 MFMA dense peak (the 5 PF headline includes 2:1 sparsity)
TILE_GFLOP = 964.47                  # algorithmic GFLOP per x4 128^2 tile (SURVEY 8d / BASELINE.md 3)
X4_CFG = dict(type='FeMaSRNet', codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
# the other single-GPU BASELINE configs as first-class workloads: same JSON schema, their own roofline and CPU sample
WORKLOADS = {
    'x2b32': dict(cfg=dict(type='FeMaSRNet', codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=2), batch=32, hw=256, fn='test', out_hw=512,
                  gflop=1075.05, tile='256x256->512x512', metric='SR output megapixels/sec at x2 (256->512)',
                  text='BASELINE config 4: x2 SR FeMaSRNet.test (scale_factor=2 encoder head: 3->128 in_conv, TWO stride-2 stages 128->256->256, femasr_arch.py:255-256), batch {B} of 256x256 LR tiles -> 512x512 '
                       '(padded 288->576 inside)'),
    'hq8': dict(cfg=dict(type='FeMaSRNet', codebook_params=[[32, 1024, 512]], LQ_stage=False), batch=8, hw=512, fn='forward', out_hw=512,
                gflop=544.21, tile='512x512->512x512', metric='autoencoded megapixels/sec (HQ pretrain stage, 512x512)',
                text='BASELINE config 5: HQ autoencode FeMaSRNet.forward (encode -> VQ -> decode, no LR encoder / Swin stage, no pad), batch {B} of 512x512 images'),
}


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
    m = re.match(r'conv_igemm<(\d+)x(\d+),(\w+),cinvec=(\w+),waves=(\d)x(\d)>', bench_name)
    if m:
        return f'conv_igemm_kernel<{m.group(1)}, {m.group(2)}, {m.group(5)}, {m.group(6)}, {pro[m.group(3)]}, {m.group(4)}>'
    if bench_name.startswith('conv3x3_wino_up2<'):
        return 'conv3x3_wino_up2_kernel<'          # (a prefix: pmc_record averages the instantiations launch-weighted)
    m = re.match(r'conv3x3_wino4<2x16x16px x64,(\w+),(\w+),res=(\*|\d),waves=8>', bench_name)
    if m:      # (res=*: the merged slot of the three residual-operand instantiations, see merge_wino_slots / pmc_record)
        return f'conv3x3_wino4_kernel<{pro[m.group(1)]}, {m.group(2)}' + ('' if m.group(3) == '*' else f', {m.group(3)}>')
    m = re.match(r'conv3x3_wino4<16x16px x128,(\w+),(\w+),res=(\*|\d),waves=8>', bench_name)
    if m:      # the 16x16-pixel x 128-channel block shape (kernels_wino_c128.hip)
        return f'conv3x3_wino4c_kernel<{pro[m.group(1)]}, {m.group(2)}' + ('' if m.group(3) == '*' else f', {m.group(3)}>')
    m = re.match(r'gemm_dma<tile=64\*(\d),k=8\*(\d),stages=(\d),act=(\d),nres=(\d),vq=(\w+)>', bench_name)
    if m:
        return f'gemm_dma_kernel<{m.group(1)}, {m.group(2)}, {m.group(3)}, {m.group(4)}, {m.group(5)}, {m.group(6)}>'
    return bench_name


def merge_wino_slots(convs):
    """The F(4x4,3x3) kernel is instantiated per number of residual operands of its epilogue (res=0/1/2: compile-time instead of
    per-pixel selects); for the roofline it is ONE kernel - same main loop, same flops per launch - so its profile slots are
    summed into one entry named ...,res=*,... ((ms, launches, flops, extra) per slot)."""
    import re
    out = {}
    for k, v in convs.items():
        kk = re.sub(r'^(conv3x3_wino4<.*),res=\d,', r'\1,res=*,', k)
        if kk in out:
            o = out[kk]
            out[kk] = (o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3])
        else:
            out[kk] = tuple(v)
    return out


def pmc_record(pmc, kname):
    """profiles/pmc_traffic.json record of a kernel; a name without its closing '>' (merged instantiations) = the launch-weighted
    mean over every instantiation that starts with it."""
    if kname in pmc:
        return pmc[kname]
    rs = [r for k, r in pmc.items() if k.startswith(kname)]
    n = sum(r['launches'] for r in rs)
    if not rs or not n:
        return None
    return {'fetch_bytes_corrected': sum(r['fetch_bytes_corrected'] * r['launches'] for r in rs) / n,
            'write_bytes': sum(r['write_bytes'] * r['launches'] for r in rs) / n, 'launches': n}


def measure_traffic(args, timeout_s=150):
    """HBM traffic per launch of every kernel, measured on THIS box in THIS run: two `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE
    need separate passes: TCC counter slots, MI355X_MICROARCH.md) over two serialized steps of the same workload, launched from here
    after the timed region.  Returns {kernel name: {'fetch_bytes_corrected', 'write_bytes', 'launches'}} (the format of
    profiles/pmc_traffic.json; FETCH_SIZE doubled per the guide's gfx950 correction) or None when rocprofv3 is missing / fails."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if not exe:
        return None
    data = {}
    with tempfile.TemporaryDirectory(dir='/tmp') as d:
        cmd = [sys.executable, os.path.abspath(__file__), '--steps', '2', '--warmup', '1', '--streams', '1', '--no-cpu-baseline', '--no-profile',
               '--no-bf16x3-leg', '--no-measure-traffic', '--workload', args.workload, '--decoder-math', args.decoder_math,
               '--linear-math', args.linear_math, '--batch', str(args.batch)]
        env = dict(os.environ, TMPDIR='/tmp')
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(d, ctr)
            try:
                r = subprocess.run([exe, '--pmc', ctr, '--kernel-trace', '-d', out, '-o', ctr.lower(), '--'] + cmd, cwd='/tmp', env=env,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except Exception:
                return None
            dbs = glob.glob(os.path.join(out, '**', '*.db'), recursive=True)
            if r.returncode != 0 or not dbs:
                return None
            cur = sqlite3.connect(dbs[0]).cursor()
            try:
                rows = list(cur.execute('select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name'))
            except Exception:
                return None
            for name, c, n, val in rows:
                name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
                data.setdefault(name, {})[c] = (n, val)
    recs = {}
    for name, dd in data.items():
        if 'FETCH_SIZE' in dd and 'WRITE_SIZE' in dd:          # (KiB per dispatch)
            recs[name] = {'fetch_bytes_corrected': 2 * dd['FETCH_SIZE'][1] * 1024, 'write_bytes': dd['WRITE_SIZE'][1] * 1024, 'launches': dd['FETCH_SIZE'][0]}
    return recs or None


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` outside torchrun: spawn the N ranks (one process per GPU) and relay their exit code."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this driver (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


class DryNet:
    """Stand-in for the network when no GPU is present (--dry-net): same `test` / `test_tile` host code path
    (tiling, partition, gather, paste come from the product module), a trivial per-tile function instead of the
    HIP forward.  Exists so that the N>1 launch path can be exercised on a CPU box."""

    def __init__(self, net):
        import types
        import torch.nn.functional as F

        def fake_test(self_, t, out=None):
            y = F.interpolate(t, scale_factor=4, mode='nearest') * 0.5 + t.amax(dim=(1, 2, 3), keepdim=True)
            return y if out is None else out.copy_(y)
        net.test = types.MethodType(fake_test, net)
        self.net = net


class PowerWatch:
    """Package power and engine clock of one GPU while steps run: the amdgpu hwmon files (power1_average / power1_input in uW,
    freq1_input in Hz, power1_cap) read by a host thread at ~25 Hz - no GPU work, no effect on the timed region; `rocm-smi` polled
    at ~2 Hz when the files are not there.  Round 4 found the MFMA kernels of this network within 1-6 % of the 1400 W package limit
    (profiles/r04_k_power_clock.txt): a step time is not interpretable without the clock it ran at, so the line carries both."""

    def __init__(self, pci=None):
        """pci = (domain, bus, device) of the GPU (torch.cuda.get_device_properties): /sys/class/drm lists EVERY card of the host, also
        inside a one-GPU container, so the card is matched by its PCI address; without a match nothing is read from sysfs."""
        import glob
        self.dir, self.samples, self._stop, self._thr = None, [], False, None
        for d in sorted(glob.glob('/sys/class/drm/card*/device')):
            hw = sorted(glob.glob(os.path.join(d, 'hwmon', 'hwmon*')))
            if not hw or pci is None:
                continue
            try:
                dom, bus, devfn = os.path.basename(os.path.realpath(d)).split(':')
                if (int(dom, 16), int(bus, 16), int(devfn.split('.')[0], 16)) != tuple(int(v) for v in pci):
                    continue
            except (ValueError, IndexError):
                continue
            for h in hw:
                if any(os.path.exists(os.path.join(h, f)) for f in ('power1_input', 'power1_average')):
                    self.dir = h
                    break
            if self.dir:
                break
        self.source = f'amdgpu hwmon ({self.dir})' if self.dir else 'rocm-smi --showpower --showclocks'

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _sample(self):
        if self.dir:
            p = self._read('power1_input')            # (MI300-class: the socket's current power, label PPT; older parts: power1_average)
            if p is None:
                p = self._read('power1_average')
            f = self._read('freq1_input')
            return (p / 1e6 if p else None, f / 1e6 if f else None)
        import re
        import subprocess
        try:
            out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=10).stdout
        except (OSError, subprocess.SubprocessError):
            return (None, None)
        mp = re.search(r'GPU\[0\].*Power \(W\):\s*([\d.]+)', out)
        mc = re.search(r'GPU\[0\].*sclk clock level:.*\((\d+)Mhz\)', out)
        return (float(mp.group(1)) if mp else None, float(mc.group(1)) if mc else None)

    def _loop(self):
        while not self._stop:
            self.samples.append(self._sample())
            time.sleep(0.04 if self.dir else 0.1)

    def start(self):
        import threading
        self.samples, self._stop = [], False
        self._thr = threading.Thread(target=self._loop, daemon=True)
        self._thr.start()

    def stop(self):
        self._stop = True
        if self._thr:
            self._thr.join(timeout=15)
        pw = [a for a, _ in self.samples if a]
        ck = sorted(b for _, b in self.samples if b)
        cap = self._read('power1_cap') if self.dir else None
        if not pw and not ck:
            return None
        return {'package_w_mean': round(sum(pw) / len(pw), 1) if pw else None, 'package_w_max': round(max(pw), 1) if pw else None,
                'cap_w': round(cap / 1e6, 1) if cap else None, 'sclk_mhz_median': round(ck[len(ck) // 2], 1) if ck else None,
                'sclk_mhz_min_max': [round(ck[0], 1), round(ck[-1], 1)] if ck else None, 'samples': len(self.samples), 'source': self.source}


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', choices=['tiles16', 'tile2048', 'x2b32', 'hq8'], default='tiles16')
    ap.add_argument('--batch', type=int, default=None, help='128x128 LR tiles per GPU per step (tiles16) / per batched test() call (tile2048)')
    ap.add_argument('--image', type=int, default=2048, help='tile2048: LR image side')
    ap.add_argument('--streams', type=int, default=3, help='sub-batch streams inside one forward (femasr_set_streams)')
    ap.add_argument('--profile-steps', type=int, default=2, help='extra serialized steps (streams=1) for the roofline object')
    ap.add_argument('--decoder-math', choices=['fp32', 'fp32_strict', 'bf16x3', 'fp32_direct'], default='fp32',
                    help="'fp32' (bench of record, the product default): every layer fp32; the 3x3 convs behind the VQ lookup in the Winograd "
                         "F(4x4,3x3) form with the SiLU of their GroupNorm prologue on the hardware exp2 / rcp units.  'fp32_strict': the same with "
                         "the IEEE-exact SiLU (bit-identical to the oracle).  'fp32_direct': every conv in the direct form.  'bf16x3': convs behind "
                         'the VQ lookup on the bf16 matrix cores (3-term split, within 1e-3) - a secondary mode, reported as such')
    ap.add_argument('--linear-math', choices=['bf16_split', 'fp32'], default='bf16_split',
                    help="arithmetic of the 1x1 convs / nn.Linear layers: 'bf16_split' (product default: exact 3-term bf16 split, 6 products, fp32 accumulation "
                         "on the bf16 matrix pipe - fp32-grade, bit-identical to the oracle) or 'fp32' (fp32 MFMA fmaf chain)")
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default=None)
    ap.add_argument('--dry-net', action='store_true', help='CPU stand-in network (launch-path test; no GPU work, not a measurement)')
    ap.add_argument('--no-gather', action='store_true', help='tiles16, N>1: skip the all-gather of upscaled tiles')
    ap.add_argument('--force-gather', action='store_true',
                    help='N=1: still run the all-gather path through a one-rank RCCL group (exercises the N>1 code on one GPU)')
    ap.add_argument('--no-strong-leg', action='store_true',
                    help='N > 1, tiles16: do not append the strong-scaling leg (BASELINE config 3a: ONE 2048x2048 LR image, 256 tiles of 128, sharded '
                         'over the ranks; all-gather + paste timed) that makes the same JSON line carry both scalings')
    ap.add_argument('--strong-steps', type=int, default=2, help='timed steps of the strong-scaling leg (after one warm-up step)')
    ap.add_argument('--no-other-configs', action='store_true', help='tiles16, one GPU: skip the short x2b32 / hq8 legs (BASELINE configs 4 and 5, ~15 s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-bf16x3-leg', '--no-exact-leg', dest='no_second_leg', action='store_true',
                    help='skip the extra timing of the other decoder-math mode')
    ap.add_argument('--no-profile', action='store_true', help='do not record per-kernel HIP events')
    ap.add_argument('--no-measure-traffic', action='store_true',
                    help='do not run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; ~20 s each) that make roofline.traffic a number of THIS run; '
                         'the line then carries the figure of profiles/pmc_traffic.json, stamped as such')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from femasr_amd import distributed as fd
    from femasr_amd import synth
    from femasr_amd.archs import build_network

    dry = args.dry_net
    backend = args.backend or ('gloo' if dry else 'nccl')
    rank, world, local = fd.init_from_env(backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if dry:
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)

    def sync():
        if not dry:
            torch.cuda.synchronize(dev)

    wl = WORKLOADS.get(args.workload)
    if args.batch is None:
        args.batch = wl['batch'] if wl else 16
    net = build_network(dict(wl['cfg'] if wl else X4_CFG))
    if dry:
        net = DryNet(net).net
    else:
        sd = synth.fill_state_dict(net.state_dict(), seed=0, codebook='trained')
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        net = net.to(dev).eval()
        net.num_streams = args.streams
        net.decoder_math = args.decoder_math
        net.linear_math = args.linear_math
    B = args.batch
    if args.force_gather and world == 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        dist.init_process_group(backend=backend, rank=0, world_size=1)
    use_pg = dist.is_initialized()

    sg = None
    if args.workload == 'tiles16':
        x = torch.from_numpy(synth.synth_input(1000 + rank, (B, 3, 128, 128))).to(dev)
        do_gather = use_pg and not args.no_gather
        # The all-gather of step k runs on RCCL's stream while step k+1 computes (double-buffered persistent receive
        # buffers, all_gather_into_tensor): the upscaled tiles of a step are only consumed by the paste, so a serving
        # loop pipelines exactly like this.
        sg = fd.StepGather((B, 3, 512, 512), torch.float32, dev) if do_gather else None      # persistent double-buffered send / receive
        nstep = [0]

        def step():
            if do_gather:
                k = nstep[0]
                y = net.test(x, out=sg.send(k))      # the forward's last kernel writes the send buffer: no copy
                sg.launch(k)
            else:
                y = net.test(x)
            nstep[0] += 1
            return y
        out_mpix = world * B * 512 * 512 / 1e6
        scaling = 'weak'
        workload = (f'x4 SR FeMaSRNet.test, batch {B} of 128x128 LR tiles per GPU -> 512x512 (padded 144->576 inside, reference '
                    'geometry), synthetic random weights (seed 0; codebook drawn at the scale of z), inputs resident in HBM')
        units_per_step = B * world
    elif wl:
        x = torch.from_numpy(synth.synth_input(1000 + rank, (B, 3, wl['hw'], wl['hw']))).to(dev)
        do_gather = False

        def step():
            return net.test(x) if wl['fn'] == 'test' else net(x)[0]
        out_mpix = world * B * wl['out_hw'] * wl['out_hw'] / 1e6
        scaling = 'weak'
        workload = wl['text'].format(B=B) + ' per GPU, synthetic random weights (seed 0; codebook drawn at the scale of z), inputs resident in HBM'
        units_per_step = B * world
    else:
        S = args.image
        do_gather = world > 1
        img = torch.from_numpy(synth.synth_input(2000, (1, 3, S, S))).to(dev)    # replicated LR image (48 MB at 2048^2)
        net.max_tile_batch = B
        net.time_split = not dry

        def step():
            return fd.test_tile_parallel(net, img, 128, 0)       # partition -> batched test() -> ONE all-gather -> paste
        out_mpix = (4 * S) * (4 * S) / 1e6
        scaling = 'strong'
        workload = (f'x4 SR of ONE {S}x{S} LR image: test_tile(tile_size=128, tile_pad=0) = {(S // 128) ** 2} tiles of 128x128 sharded '
                    f'over {world} rank(s) in batches of {B}, RCCL all-gather of the upscaled tiles + paste into the {4 * S}x{4 * S} canvas '
                    'inside the timed region; synthetic random weights (seed 0), LR image resident in HBM on every rank')
        units_per_step = (S // 128) ** 2

    def fence():
        if args.workload == 'tiles16' and sg is not None:
            sg.wait_all()
        if use_pg:
            dist.barrier()
        sync()

    def clock_probe():
        """Clock the chip sustains under back-to-back fp32 MFMAs right now (femasr_clock_probe: ticks of s_memtime / wall time)."""
        if dry:
            return None
        from femasr_amd import _lib
        lib = _lib.load()
        ticks = torch.zeros(int(lib.femasr_clock_probe_entries()), dtype=torch.int64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream(dev)
        # two back-to-back probes, the SECOND one timed: launch and clock ramp are behind it (a single cold probe read low)
        _lib.check(lib.femasr_clock_probe(st.cuda_stream, 40000, _lib.ptr(ticks)))
        e0.record(st)
        _lib.check(lib.femasr_clock_probe(st.cuda_stream, 40000, _lib.ptr(ticks)))       # ~1.1 ms of MFMAs per wave
        e1.record(st)
        sync()
        return round(float(ticks.double().mean().item()) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 3)

    clk_before = clock_probe()       # ahead of the warm-up: the timed steps follow the warm-up steps directly (with the probe between them the
    for _ in range(args.warmup):     # first timed step ran 4 ms slow - the chip leaves the probe's pure-MFMA loop in another power state)
        step()
    fence()
    watch = None
    if rank == 0 and not dry:
        pr = torch.cuda.get_device_properties(dev)
        watch = PowerWatch((pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id) if hasattr(pr, 'pci_bus_id') else None)
        if watch.dir:                      # file reads on a host thread: safe inside the timed region (rocm-smi is sampled in its own loop below)
            watch.start()
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if not dry else None
    t0 = time.perf_counter()
    if step_ev:
        step_ev[0].record(torch.cuda.current_stream(dev))
    for i in range(args.steps):
        y = step()
        if step_ev:
            step_ev[i + 1].record(torch.cuda.current_stream(dev))
    fence()
    dt = time.perf_counter() - t0
    dt_rank = dt
    power = watch.stop() if (watch and watch.dir) else None
    clk_after = clock_probe()
    per_rank_ms = [round(dt / args.steps * 1e3, 3)]
    if world > 1:
        tall = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dt], dtype=torch.float64, device=dev))
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 3) for t in tall]
        dt = max(float(t.item()) for t in tall)                 # the MAX over ranks is the job's time
    assert torch.isfinite(y).all()
    step_ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)] if step_ev else []

    unit_gflop = wl['gflop'] if wl else TILE_GFLOP          # algorithmic GFLOP of one unit
    value = out_mpix * args.steps / dt
    res = {
        'metric': wl['metric'] if wl else 'SR output megapixels/sec at x4 (128->512)', 'value': round(value, 4), 'unit': 'MPix/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
        'dtype': (('f32' if args.linear_math == 'fp32' else
                   'f32 (1x1 convs / nn.Linear: every fp32 operand split EXACTLY into three bf16 terms, six partial products, fp32 accumulation on the '
                   'bf16 matrix pipe - closer to fp64 than the fp32 fmaf chain, bit-identical to the CPU oracle)')
                  if args.decoder_math in ('fp32', 'fp32_strict', 'fp32_direct') else 'f32 + bf16x3 split (secondary mode, not the bench of record)'),
        'data': 'synthetic' if not dry else 'dry-net stand-in on CPU (launch-path check, NOT a measurement)',
        'config': {'workload': workload, 'workload_name': args.workload,
                   'global_batch': units_per_step, 'tile': wl['tile'] if wl else '128x128->512x512', 'parallelism': f'tile-parallel x{world}',
                   'backend': ('RCCL (torch.distributed nccl)' if backend == 'nccl' else backend) if use_pg else 'none (single process)',
                   'gather': bool(do_gather), 'streams': args.streams, 'decoder_math': args.decoder_math, 'linear_math': args.linear_math,
                   'decoder_math_note': ("product default: all fp32; the SiLU of the Winograd convs' GroupNorm prologue on the hardware exp2 / rcp units - "
                                         "VQ indices exact, image within 1e-5 of 'fp32_strict' (the mode that is bit-identical to the CPU oracle, timed as a secondary leg)"
                                         if args.decoder_math == 'fp32' else None),
                   'algorithmic_gflop_per_tile': unit_gflop,
                   'end_to_end_algorithmic_tflops': None if dry else round(unit_gflop * units_per_step * args.steps / dt / 1e3, 2),
                   'flops_note': ('algorithmic = the layer DEFINITIONS (964.47 GFLOP per tile, direct form); the default fp32 mode ISSUES far '
                                  'fewer (behind the VQ lookup: Winograd F(4x4,3x3) 36/144, nearest-x2 convs in the 25-product form 25/144; phase-filter x2 convs elsewhere: 4/9) - the issued figure and '
                                  'the physical MFMA fraction are in `roofline`')},
        'timed_region': {'per_rank_ms_per_step': per_rank_ms, 'world_size': world,
                         'rccl_version': ('.'.join(map(str, torch.cuda.nccl.version())) if (use_pg and backend == 'nccl') else None),
                         'step_ms_first_median_last': ([round(step_ms[0], 2), round(sorted(step_ms)[len(step_ms) // 2], 2), round(step_ms[-1], 2)]
                                                       if step_ms else None),
                         'step_ms_min_max': [round(min(step_ms), 2), round(max(step_ms), 2)] if step_ms else None,
                         'mfma_clock_ghz_before_after': [clk_before, clk_after],
                         'clock_note': ('clock the chip sustains under back-to-back fp32 MFMAs (femasr_clock_probe), probed before the warm-up steps and '
                                        'right after the timed steps: the hot kernels are clock-bound, so box-to-box / thermal differences '
                                        'show here')},
    }
    if watch and not watch.dir and world == 1:      # no hwmon files: rocm-smi (slow to poll) while ~3 s of extra, untimed steps run
        watch.start()
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < 3.0:
            step()
        fence()
        power = watch.stop()
        if power:
            power['sampled_in'] = 'about 3 s of extra steps right after the timed region (rocm-smi is too slow to poll inside it)'
    if power:
        power.setdefault('sampled_in', 'the timed steps')
        power['note'] = ('package power and engine clock while the steps run.  Stand-alone, the fp32 MFMA kernels of this network draw 1.3-1.4 kW '
                         '(a pure MFMA stream: 0.72 kW at 98 % of the peak) - within 1-6 % of the 1400 W package limit; a kernel that moves '
                         'more bytes per MFMA is clocked down (DESIGN.md 5, "last experiment")')
        res['power'] = power
    if args.workload == 'tile2048' and not dry:
        # where the last step went on every rank: batched test() calls / the RCCL all-gather / the paste of all ranks' tiles
        split = net.last_split_ms
        if world > 1:
            allsp = [None] * world
            dist.all_gather_object(allsp, split)
            split = allsp
        res['timed_region']['last_step_split_ms_per_rank'] = split
    if args.workload == 'tiles16' and do_gather:
        res['config']['gather_overlap'] = 'all-gather of step k overlaps step k+1'

    # ---------------------------------------------------------------- strong-scaling leg (all ranks; after the timed region of record)
    # the driver runs ONE command per N.  With N > 1 the default (weak-scaling) workload therefore appends BASELINE
    # config 3a - one 2048x2048 LR image = 256 tiles of 128x128, sharded over the ranks by the work-balanced partition, ONE RCCL all-gather
    # of the upscaled tiles, paste on rank 0 - so that the same JSON line records a point of the strong curve too, with each rank's
    # compute / gather / paste split.  (`--workload tile2048` is the same thing as the workload of record, with every rank pasting.)
    if world > 1 and args.workload == 'tiles16' and not args.no_strong_leg and use_pg:
        if sg is not None:
            sg.wait_all()
        S = 512 if dry else args.image
        img = torch.from_numpy(synth.synth_input(2000, (1, 3, S, S))).to(dev)
        keep_batch = net.max_tile_batch
        net.max_tile_batch = B
        net.time_split = not dry

        def strong_step():
            return fd.test_tile_parallel(net, img, 128, 0, root_only=True)
        strong_step()
        dist.barrier()
        sync()
        ts0 = time.perf_counter()
        for _ in range(max(1, args.strong_steps)):
            ys = strong_step()
        dist.barrier()
        sync()
        dts = (time.perf_counter() - ts0) / max(1, args.strong_steps)
        tall = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dts], dtype=torch.float64, device=dev))
        dts_max = max(float(t.item()) for t in tall)
        split = [None] * world
        dist.all_gather_object(split, net.last_split_ms if not dry else None)
        from femasr_amd import tiling
        cls = tiling.shape_classes(tiling.enumerate_tiles(S, S, 128, 0))
        if rank == 0:
            assert ys is not None and tuple(ys.shape) == (1, 3, 4 * S, 4 * S)
        res['strong_scaling'] = {
            'workload': (f'x4 SR of ONE {S}x{S} LR image: test_tile(128, 0) = {(S // 128) ** 2} tiles sharded over {world} ranks in batches of {B}, ONE RCCL '
                         f'all-gather of the upscaled fp32 tiles, paste into the {4 * S}x{4 * S} canvas on rank 0 - all inside the timed step (BASELINE config 3a)'),
            'scaling': 'strong', 'steps': max(1, args.strong_steps), 'warmup': 1,
            'ms_per_step': round(dts_max * 1e3, 3), 'value': round((4 * S) * (4 * S) / 1e6 / dts_max, 4), 'unit': 'MPix/s',
            'per_rank_ms_per_step': [round(float(t.item()) * 1e3, 3) for t in tall],
            'last_step_split_ms_per_rank': split,
            'tiles_per_rank': [sum(len(tl) for tl in o.values()) for o in tiling.assign(cls, world, 4)],
            'partition_bound': round(tiling.balance_bound(cls, world, 4), 3),
            'one_gpu_reference': 'profiles/r04_tile2048_bench.json: 1.147 s per step = 58.5 MPix/s on one MI355X (round 4; gather 0.005 ms, paste 0.62 ms)',
        }
        net.max_tile_batch = keep_batch
        net.time_split = False
        del img, ys

    # ---------------------------------------------------------------- rank-0 extras (not part of the timed region)
    if rank == 0 and not dry:
        x16 = x if (args.workload == 'tiles16' or wl) else torch.from_numpy(synth.synth_input(1000, (B, 3, 128, 128))).to(dev)
        run16 = (lambda t: net(t)[0]) if (wl and wl['fn'] == 'forward') else net.test          # the unit forward of this workload
        # Per-kernel roofline: HIP events around every launch on the launch stream.  With >1 sub-batch streams kernels
        # of different streams overlap and a per-kernel duration is not separable, so the events are recorded in extra
        # SERIALIZED steps (streams=1) run right after the timed region.
        if not args.no_profile and args.profile_steps > 0:
            net.num_streams = 1
            run16(x16)
            sync()
            net.enable_profile(True)
            tp0 = time.perf_counter()
            for _ in range(args.profile_steps):
                run16(x16)
            sync()
            prof_ms_per_step = (time.perf_counter() - tp0) / args.profile_steps * 1e3
            prof = net.profile()
            net.enable_profile(False)
            net.num_streams = args.streams
            convs = merge_wino_slots({k: v for k, v in prof.items() if k.startswith('conv')})      # the MFMA kernels
            psteps = args.profile_steps
            pmc_live = None
            if world == 1 and not args.no_measure_traffic:
                try:
                    pmc_live = measure_traffic(args)
                except Exception:
                    pmc_live = None
            pmc = pmc_live or (json.load(open(PMC_TRAFFIC_JSON)) if os.path.exists(PMC_TRAFFIC_JSON) else {})

            def issued_share(name):      # MFMA flops issued / algorithmic flops of the layer definition
                if name.startswith('conv3x3_wino_up2'):
                    return 25.0 / 144.0          # nearest-x2 + 3x3 conv, Winograd-type form: 25 multiplies per 4x4 outputs
                if name.startswith('conv3x3_wino'):
                    return 36.0 / 144.0          # Winograd F(4x4,3x3): 36 multiplies per 4x4 outputs instead of 16 x 9
                if name.startswith('conv3x3_halo<') and 'up2=true' in name:
                    return 4.0 / 9.0             # nearest-x2 + 3x3 conv as four 2x2-tap phase filters
                if name.startswith('conv3x3_halo_bf16x3'):
                    return 3.0                   # three bf16 MFMA passes per multiply-add
                if name.startswith('gemm_bf16s') or name.startswith('conv3x3_bf16s'):
                    # six bf16 MFMA passes per multiply-add of the definition, on a pipe 15.9x as fast: in units of the fp32-MFMA peak (so that the
                    # sums below stay physical pipe-time fractions <= 1)
                    return 6.0 * PEAK_FP32_MFMA_TFLOPS / PEAK_BF16_MFMA_TFLOPS
                return 1.0

            def roof(name):
                """`achieved` / `frac` are PHYSICAL: the flops the MFMA pipe executes per second and their share of the dense peak
                (<= 1).  The layer definition's (algorithmic) flops over the same time are reported beside them."""
                ms, n, fl, alg_bytes = convs[name]
                split = name.startswith('conv3x3_halo_bf16x3')
                peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
                alg = fl / (ms * 1e-3) / 1e12
                share = issued_share(name)
                rec = pmc_record(pmc, rocprof_kernel_name(name))
                out = {'bound': 'mfma', 'kernel': name, 'achieved': round(alg * share, 2), 'peak': peak, 'unit': 'TFLOP/s',
                       'frac': round(alg * share / peak, 4),
                       'traffic': round((rec['fetch_bytes_corrected'] + rec['write_bytes']) / 1e9, 4) if rec else None,
                       'algorithmic_GB_per_launch': round(alg_bytes / n / 1e9, 4) if alg_bytes else None,
                       'launches': n, 'avg_launch_ms': round(ms / n, 4),
                       'issued_gflop_per_launch': round(fl * share / n / 1e9, 3), 'algorithmic_gflop_per_launch': round(fl / n / 1e9, 3),
                       'issued_share_of_algorithmic': round(share, 4),
                       'algorithmic_tflops': round(alg, 2), 'algorithmic_over_peak': round(alg / peak, 4),
                       'peak_basis': ('bf16 dense MFMA peak; each multiply-add of the definition costs 3 MFMA passes (hi*hi + hi*lo + lo*hi)'
                                      if split else 'fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32)'),
                       'basis': ('achieved = flops the MFMA pipe EXECUTES (algorithmic flops of the layer definition x issued share) / HIP-event '
                                 'time of the launches; frac = achieved / peak = the physical MFMA issue fraction')}
                if name.startswith('conv3x3_wino_up2'):
                    out['form'] = 'nearest-x2 + 3x3 conv in a Winograd-type form on the low-resolution input, all fp32: 25 multiplies per 4x4 outputs where the definition has 144'
                elif name.startswith('conv3x3_wino'):
                    out['form'] = 'Winograd F(4x4,3x3), all fp32: 36 multiplies per 4x4 outputs where the definition has 144'
                elif share == 4.0 / 9.0:
                    out['form'] = 'nearest-x2 folded into four 2x2-tap phase filters: 4 multiplies per output where the definition has 9'
                if rec:
                    out['traffic_source'] = (('GB per launch MEASURED IN THIS RUN on this box: two rocprofv3 --pmc passes (FETCH_SIZE, doubled per '
                                              'MI355X_MICROARCH.md HBM section, and WRITE_SIZE) over two serialized steps of this workload, launched by bench.py '
                                              'after the timed region') if pmc_live else
                                             ('GB per launch from profiles/pmc_traffic.json (rocprofv3 --pmc passes on the BUILD box, commit-stamped there); NOT '
                                              're-measured in this run: rocprofv3 missing / failed or --no-measure-traffic'))
                    out['traffic_measured_live'] = bool(pmc_live)
                    if alg_bytes and out['traffic']:
                        # (every operand once: input, residual, output, weights; the kernel reads its input once per 64-channel column block
                        # and a 18x18 halo per 16x16 pixels - what of that misses L2 / the Infinity Cache shows up here)
                        out['traffic_over_algorithmic'] = round(out['traffic'] / (alg_bytes / n / 1e9), 3)
                return out
            dom = max(convs, key=lambda k: convs[k][0])
            res['roofline'] = roof(dom)
            # every MFMA kernel of the step (convs + the GEMMs of the Swin linears) and the step as a whole
            mf = {k: v for k, v in prof.items() if v[2] > 0 and not k.startswith('vq(')}
            tot_ms = sum(v[0] for v in mf.values())
            alg_fl = sum(v[2] for v in mf.values())
            iss_fl = sum(v[2] * issued_share(k) for k, v in mf.items())
            step_alg_gf = sum(v[2] for v in prof.values()) / psteps / 1e9
            step_iss_gf = sum(v[2] * issued_share(k) for k, v in prof.items()) / psteps / 1e9
            res['roofline']['all_mfma_kernels'] = {
                'issued_tflops': round(iss_fl / (tot_ms * 1e-3) / 1e12, 2), 'frac_of_fp32_mfma_peak': round(iss_fl / (tot_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                'algorithmic_tflops': round(alg_fl / (tot_ms * 1e-3) / 1e12, 2),
                'share_of_serialized_step_time': round(tot_ms / psteps / prof_ms_per_step, 4)}
            t_step = dt / args.steps
            res['roofline']['end_to_end'] = {
                'issued_gflop_per_step': round(step_iss_gf, 1), 'algorithmic_gflop_per_step': round(step_alg_gf, 1),
                'end_to_end_issued_tflops': round(step_iss_gf * (units_per_step / world / B) / t_step / 1e3, 2),
                'end_to_end_issued_over_fp32_mfma_peak': round(step_iss_gf * (units_per_step / world / B) / t_step / 1e3 / PEAK_FP32_MFMA_TFLOPS, 4),
                'end_to_end_algorithmic_tflops': round(step_alg_gf * (units_per_step / world / B) / t_step / 1e3, 2),
                'note': 'per GPU; flops of one profiled step (kernel launch records) over the TIMED step time of this rank'}
            res['roofline']['measured_in'] = (f'{psteps} serialized steps (streams=1, {prof_ms_per_step:.1f} ms/step, batch {B} of 128x128 tiles, '
                                              f"decoder_math={args.decoder_math}) right after the timed region")
            # the split GEMM on ITS pipe: six bf16 passes per multiply-add of the definition over the dense bf16 peak - at the
            # nominal 2.4 GHz and at the engine clock the power sampler read during the timed steps (the launch is power-bound: profiles/r06_gemm_experiments.txt).
            # Price per v_mfma_f32_32x32x16_bf16: 32 cycles per SIMD (SQ_VALU_MFMA_BUSY_CYCLES = 32 x MFMAs in every profiled launch) = the 2.5 PF figure.
            sclk = (power or {}).get('sclk_mhz_median')

            def bf16_pipe(k, v):
                if not (k.startswith('gemm_bf16s') or k.startswith('conv3x3_bf16s')) or v[2] <= 0:
                    return {}
                pf = 6.0 * v[2] / (v[0] * 1e-3) / 1e12
                return {'bf16_pipe_tflops': round(pf, 1), 'bf16_pipe_frac_at_2.4GHz': round(pf / PEAK_BF16_MFMA_TFLOPS, 4),
                        **({'bf16_pipe_frac_at_measured_clock': round(pf / (PEAK_BF16_MFMA_TFLOPS * sclk / 2400.0), 4)} if sclk else {})}
            res['roofline']['per_kernel'] = {
                k: {'ms_per_step': round(v[0] / psteps, 3), 'launches_per_step': v[1] // psteps,
                    **({'issued_tflops': round(v[2] * issued_share(k) / (v[0] * 1e-3) / 1e12, 1),
                        'algorithmic_tflops': round(v[2] / (v[0] * 1e-3) / 1e12, 1)} if v[2] > 0 else {}), **bf16_pipe(k, v)}
                for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
            lin = {k: v for k, v in prof.items() if k.startswith('gemm_bf16s') or k.startswith('gemm_dma') or k == 'layernorm'}
            if lin:
                res['roofline']['swin_linear_bucket_ms_per_step'] = round(sum(v[0] for v in lin.values()) / psteps, 3)
            vq = prof.get('vq(codebook lookup)')
            if vq:          # the north-star's VQ figure: algorithmic HBM bytes (23.37 MB per tile) / time
                res['roofline']['vq'] = {'ms_per_step': round(vq[0] / psteps, 3), 'algorithmic_GB_per_step': round(vq[3] / psteps / 1e9, 4),
                                         'hbm_GB_s': round(vq[3] / (vq[0] * 1e-3) / 1e9, 1), 'frac_of_8TB_s': round(vq[3] / (vq[0] * 1e-3) / 8e12, 4),
                                         'algorithmic_tflops': round(vq[2] / (vq[0] * 1e-3) / 1e12, 1),
                                         'search': ('single-pass fp32 MFMA (FEMASR_VQ=gemm)' if os.environ.get('FEMASR_VQ') == 'gemm' else
                                                    'two-pass exact: bf16 MFMA candidates + fp32 chain re-check (kernels_vq.hip)')}
        if world == 1 and not args.no_second_leg and args.workload == 'tiles16':
            legs = ['fp32_strict', 'bf16x3', 'fp32_direct'] if args.decoder_math == 'fp32' else ['fp32']
            if args.linear_math == 'bf16_split':
                legs = ['fp32_linears'] + legs
            for other in legs:
                if other == 'fp32_linears':          # the timed mode with the 1x1 / Linear layers as fp32 MFMA fmaf chains (the round-4 arithmetic)
                    net.linear_math = 'fp32'
                else:
                    net.linear_math = args.linear_math
                    net.decoder_math = other
                net.test(x16)
                sync()
                te0 = time.perf_counter()
                for _ in range(args.steps):
                    ye = net.test(x16)
                sync()
                te = (time.perf_counter() - te0) / args.steps
                leg = {'value': round(B * 512 * 512 / 1e6 / te, 4), 'unit': 'MPix/s', 'ms_per_step': round(te * 1e3, 3),
                       'max_abs_vs_timed_mode': float((ye - y).abs().max()),
                       'end_to_end_algorithmic_tflops': round(TILE_GFLOP * B / te / 1e3, 2)}
                if other == 'fp32_strict':
                    leg['note'] = ('the timed mode with the IEEE-exact SiLU in the Winograd convs (polynomial exp + division instead of v_exp_f32 / '
                                   'v_rcp_f32): bit-identical to the CPU oracle; VQ indices identical to the timed mode')
                if other == 'bf16x3':
                    leg['note'] = ('secondary mode: 698 of 964 GFLOP per tile as 3-pass split-bf16 MFMA (products narrower than fp32); '
                                   'everything feeding the VQ argmin stays exact fp32; NOT the bench of record')
                if other == 'fp32_direct':
                    leg['note'] = ('the same fp32 network without the Winograd form (the 3x3 convs behind the VQ lookup in the direct form; the x2 '
                                   'convs still as phase filters): bit-identical to OracleNet(winograd=False); VQ indices identical, output '
                                   'within fp32 rounding of the timed mode')
                if other == 'fp32_linears':
                    leg['note'] = ("the timed mode with linear_math='fp32': every 1x1 conv / nn.Linear one fp32 fmaf chain per output on the fp32 MFMA "
                                   '(the arithmetic of rounds 1-4); both modes are bit-identical to the CPU oracle in their arithmetic')
                res[{'bf16x3': 'bf16x3_mode', 'fp32_direct': 'fp32_direct_mode', 'fp32_strict': 'fp32_strict_mode', 'fp32': 'fp32_mode',
                     'fp32_linears': 'fp32_linears_mode'}[other]] = leg
            net.decoder_math = args.decoder_math
            net.linear_math = args.linear_math
        if world == 1 and not args.no_second_leg and args.workload == 'tiles16' and not args.no_other_configs:
            # BASELINE configs 4 and 5 as short legs of the DEFAULT run (VERDICT r5 "missing" 3: the driver runs this command only, so the
            # x2 / HQ numbers were the builder's alone): the same step functions as `--workload x2b32 | hq8`, a few steps each, the product
            # default modes, inputs resident in HBM.  Secondary information - `value` above is config 2.
            others = {}
            for name, w2 in WORKLOADS.items():
                try:
                    net2 = build_network(dict(w2['cfg']))
                    sd2 = synth.fill_state_dict(net2.state_dict(), seed=0, codebook='trained')
                    net2.load_state_dict({k: torch.from_numpy(v) for k, v in sd2.items()}, strict=False)
                    net2 = net2.to(dev).eval()
                    net2.num_streams, net2.decoder_math, net2.linear_math = args.streams, args.decoder_math, args.linear_math
                    x2 = torch.from_numpy(synth.synth_input(1000, (w2['batch'], 3, w2['hw'], w2['hw']))).to(dev)
                    run2 = (lambda t: net2.test(t)) if w2['fn'] == 'test' else (lambda t: net2(t)[0])
                    for _ in range(2):
                        run2(x2)
                    sync()
                    n2 = 5
                    tq = time.perf_counter()
                    for _ in range(n2):
                        y2 = run2(x2)
                    sync()
                    t2 = (time.perf_counter() - tq) / n2
                    assert torch.isfinite(y2).all()
                    others[name] = {'metric': w2['metric'], 'value': round(w2['batch'] * w2['out_hw'] ** 2 / 1e6 / t2, 4), 'unit': 'MPix/s',
                                    'ms_per_step': round(t2 * 1e3, 3), 'steps': n2, 'warmup': 2, 'batch': w2['batch'], 'tile': w2['tile'],
                                    'end_to_end_algorithmic_tflops': round(w2['gflop'] * w2['batch'] / t2 / 1e3, 2)}
                    del net2, x2, y2
                    torch.cuda.empty_cache()
                except Exception as e:          # (a leg must never cost the line of record)
                    others[name] = {'error': f'{type(e).__name__}: {e}'[:300]}
            res['other_configs'] = others
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(net, x16, y if (args.workload == 'tiles16' or wl) else None, B, wl)
            res['vq_index_match'] = res['cpu_baseline'].pop('vq_index_match')      # the metric's second half (BASELINE.json), top level
    if rank == 0:
        try:        # RCCL's banner goes through C stdio: flush it first so that the JSON line is the LAST line on stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and res.get('vq_index_match', {}).get('vq_index_mismatches_outside_rule', 0) > 0:
        sys.stderr.write('bench.py: VQ index mismatches OUTSIDE the near-tie rule - see vq_index_match in the line above\n')
        sys.exit(3)


def cpu_baseline(net, x16, y_gpu, B, wl=None):
    """The same path on this box's host cores, on ONE of the step's tiles (bounded sample): (a) the stock-torch CPU
    restatement (oracle/torch_ref.py: ATen/oneDNN ops, i.e. the reference's own arithmetic library; bit-identical to
    the reference goldens), all physical cores, 1 warm-up + median of 3; (b) the C oracle (scalar fmaf chains, the
    bit-exact checker).  Reported baselines, not targets."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    from oracle.torch_ref import TorchRefNet
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    xs = x16[:1].cpu()
    cores = physical_cores()
    prev = torch.get_num_threads()
    kw = {k: v for k, v in (wl['cfg'] if wl else X4_CFG).items() if k != 'type'}
    fwd = bool(wl and wl['fn'] == 'forward')
    unit_px = (wl['out_hw'] if wl else 512) ** 2
    unit_text = (f"{wl['tile']}, {wl['gflop']} GFLOP" if wl else f'x4 128x128->512x512, {TILE_GFLOP} GFLOP')
    tnet = TorchRefNet(sd, **kw)
    tnet_run = (lambda t: tnet.forward(t)[0]) if fwd else tnet.test
    # the best thread count for one 128x128 tile is not "all cores" on a 2-socket box (small convs): probe a few settings
    # (1 warm-up + 1 timed run each), then 1 warm-up + median of 3 at the fastest
    probe = {}
    for nt in sorted({cores, max(cores // 2, 1), max(cores // 4, 1), min(cores, 16)}, reverse=True):
        torch.set_num_threads(nt)
        tnet_run(xs)
        t1 = time.perf_counter()
        tnet_run(xs)
        probe[nt] = time.perf_counter() - t1
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    tnet_run(xs)
    ts = []
    for _ in range(3):
        t1 = time.perf_counter()
        yt = tnet_run(xs)
        ts.append(time.perf_counter() - t1)
    torch.set_num_threads(prev)
    tt = sorted(ts)[1]
    # BASELINE.json's metric names "VQ index bit-match": the index map of this tile in the reference's arithmetic (stock torch
    # ops: femasr_arch.py:35-38,58-66) against the HIP path's, each difference classified with the reference-side distances
    tnet.keep_vq_dist = True
    _, it = tnet.forward(xs) if fwd else tnet.test(xs, return_indices=True)
    tnet.keep_vq_dist = False
    vq = vq_index_report(net, xs, it.numpy().reshape(-1), tnet.vq_dist[0].numpy(), forward=fwd)
    onet = orc.OracleNet({k: v for k, v in sd.items() if not k.endswith(('relative_position_index', 'attn_mask'))},
                         linear_math=net.linear_math, **kw)
    t1 = time.perf_counter()
    yo = onet.forward(xs.numpy())[0] if fwd else onet.test(xs.numpy())
    tc = time.perf_counter() - t1
    out = {
        'value': round(unit_px / 1e6 / tt, 5), 'unit': 'MPix/s', 'cores': best, 'kind': 'port',
        'impl': 'stock torch CPU (ATen/oneDNN/MKL fp32, the arithmetic library the reference itself runs on); module restated in '
                'oracle/torch_ref.py from the reference semantics, bit-identical to the reference-recorded goldens',
        'cpu': cpu_model_name(), 'physical_cores': cores, 'torch_threads': best,
        'thread_probe_s': {str(k): round(v, 2) for k, v in probe.items()},
        'sample': f'1 of the {B} units of one step ({unit_text}): fastest of the probed thread counts, 1 warm-up + median of 3 = {tt:.2f} s',
        'c_oracle': {'value': round(unit_px / 1e6 / tc, 5), 'unit': 'MPix/s', 'cores': os.cpu_count(), 'kind': 'port',
                     'sample': (f'the same tile through oracle/femasr_oracle.c (C, OpenMP: the bit-exact checker - fp32 fmaf chains, and for the 1x1 / Linear '
                                f"layers in linear_math='bf16_split' the restated matrix-instruction arithmetic, ~30x the work of a chain) in {tc:.1f} s")},
    }
    out['vq_index_match'] = vq
    if y_gpu is not None:
        yg = y_gpu[:1].cpu().numpy()
        out['max_abs_torch_cpu_vs_gpu'] = float(np.abs(yt.numpy() - yg).max())
        out['c_oracle']['max_abs_vs_gpu'] = float(np.abs(yo - yg).max())
    try:
        if wl:
            raise RuntimeError('measured on the tiles16 workload only')
        out['host_throughput'] = cpu_host_throughput(sd, xs.numpy(), cores, best)
    except Exception as e:                                  # a reported extra, never a reason to lose the bench line
        out['host_throughput'] = {'error': f'{type(e).__name__}: {e}'[:200]}
    return out


def vq_index_report(net, xs_cpu, idx_ref, dist_ref, rule_ulp=None, forward=False):
    """The HIP path's VQ index map of one tile against the reference arithmetic's (`idx_ref`, `dist_ref` = the (tokens, n_e)
    fp32 distance matrix torch computed on the CPU).  A differing token is 'within the rule' when, in the REFERENCE's own
    distances, the code the HIP path picked is within `rule_ulp` ulp of the reference's minimum (hard part 1: encoder
    summation-order differences of ~1e-7 relative decide exact and near ties differently; anything else is a real mismatch)."""
    import numpy as np
    import torch
    from oracle.near_tie import NEAR_TIE_ULP, histogram      # the one rule of tests/, tools/parity_report.py and this line
    rule_ulp = NEAR_TIE_ULP if rule_ulp is None else rule_ulp
    dev = next(net.parameters()).device
    if forward:
        ig = net(xs_cpu.to(dev))[3][0]
    else:
        _, ig = net.test_with_indices(xs_cpu.to(dev))
    torch.cuda.synchronize()
    ig = ig.cpu().numpy().reshape(-1)
    assert ig.shape == idx_ref.shape, (ig.shape, idx_ref.shape)
    bad = np.nonzero(ig != idx_ref)[0]
    gaps = []
    for r in bad:
        dmin = np.float32(dist_ref[r, idx_ref[r]])
        gaps.append(float((np.float32(dist_ref[r, ig[r]]) - dmin) / np.spacing(np.abs(dmin))))
    # how tie-prone the tile is in the reference's own arithmetic: tokens whose runner-up is within the rule of the winner
    part = np.partition(dist_ref, 1, axis=1)[:, :2]
    near = int(np.count_nonzero((part[:, 1] - part[:, 0]) <= rule_ulp * np.spacing(np.abs(part[:, 0]))))
    outside = int(sum(1 for g in gaps if g > rule_ulp))
    return {'tokens': int(idx_ref.size), 'n_codes': int(dist_ref.shape[1]),
            'vq_index_mismatches_vs_reference_arith': int(bad.size),
            'near_tie_rule_ulp': rule_ulp,
            'vq_index_mismatches_within_rule': int(bad.size) - outside,
            'accepted_gap_histogram_ulp': histogram([g for g in gaps if g <= rule_ulp]),
            'vq_index_mismatches_outside_rule': outside,
            'bit_match_fraction': round(1.0 - bad.size / idx_ref.size, 6),
            'mismatch_gaps_ulp_in_reference_distances': [round(g, 2) for g in gaps[:16]],
            'reference_tokens_with_runner_up_within_rule': near,
            'basis': 'tile 0 of the timed step: index map of the stock-torch CPU restatement (the reference arithmetic, bit-identical to the '
                     'reference goldens) vs the HIP path; a difference counts as within the rule when the reference\'s own fp32 distance of the '
                     'code the HIP path picked is <= near_tie_rule_ulp ulp above its minimum (oracle/near_tie.py: the rule the test suite uses)'}


_HOST_WORKER = r'''
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from oracle.torch_ref import TorchRefNet
d, idx, nthreads, k = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
torch.set_num_threads(nthreads)
z = np.load(os.path.join(d, 'state.npz'))
net = TorchRefNet({n: z[n] for n in z.files if n != '__x__'}, LQ_stage=True, scale_factor=4)
x = torch.from_numpy(z['__x__'])
net.test(x)                                            # warm-up
open(os.path.join(d, f'ready_{idx}'), 'w').close()
while not os.path.exists(os.path.join(d, 'go')):
    time.sleep(0.005)
for _ in range(k):
    net.test(x)
open(os.path.join(d, f'done_{idx}'), 'w').close()
'''


def cpu_host_throughput(sd, x_np, cores, threads_each, tiles_each=2, timeout_s=180):
    """What the HOST sustains when every core works: cores // threads_each concurrent processes, each running the stock-torch
    restatement on its own copy of the tile with `threads_each` threads (the fastest single-tile setting), started together
    behind a file barrier after their warm-ups; value = all tiles / wall time.  """
    import subprocess
    import tempfile
    import numpy as np
    nproc = max(1, min(16, cores // max(1, threads_each)))
    with tempfile.TemporaryDirectory() as d:
        np.savez(os.path.join(d, 'state.npz'), __x__=x_np, **sd)
        procs = [subprocess.Popen([sys.executable, '-c', _HOST_WORKER, ROOT, d, str(i), str(threads_each), str(tiles_each)],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(nproc)]
        try:
            t_end = time.time() + timeout_s
            while not all(os.path.exists(os.path.join(d, f'ready_{i}')) for i in range(nproc)):
                if time.time() > t_end or any(p.poll() not in (None, 0) for p in procs):
                    raise RuntimeError('workers did not get ready')
                time.sleep(0.01)
            t0 = time.perf_counter()
            open(os.path.join(d, 'go'), 'w').close()
            while not all(os.path.exists(os.path.join(d, f'done_{i}')) for i in range(nproc)):
                if time.time() > t_end:
                    raise RuntimeError('workers did not finish')
                time.sleep(0.005)
            wall = time.perf_counter() - t0
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
                p.wait()
    n = nproc * tiles_each
    return {'value': round(n * 512 * 512 / 1e6 / wall, 5), 'unit': 'MPix/s', 'processes': nproc, 'threads_each': threads_each,
            'tiles': n, 'seconds': round(wall, 2),
            'sample': f'{nproc} concurrent processes x {tiles_each} tiles, {threads_each} torch threads each, started together after their warm-ups'}


if __name__ == '__main__':
    main()
