"""Package power and engine clock under a PURE fp32 MFMA stream (femasr_clock_probe: one wave per SIMD, four independent accumulator
tiles, back-to-back v_mfma_f32_32x32x2_f32, no memory traffic) - the price of the multiplies alone, against which the conv / GEMM kernels'
package power is read ("the power limit").      python tools/power_probe.py [--seconds 6]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import bench  # noqa: E402
from femasr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=6.0)
    ap.add_argument('--mfmas', type=int, default=400000, help='MFMAs per wave and launch')
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    n = int(lib.femasr_clock_probe_entries())
    ticks = torch.zeros(n, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)
    pr = torch.cuda.get_device_properties(dev)
    watch = bench.PowerWatch((pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
    for _ in range(3):
        _lib.check(lib.femasr_clock_probe(st.cuda_stream, a.mfmas, _lib.ptr(ticks)))
    torch.cuda.synchronize()
    watch.start()
    t0 = time.perf_counter()
    launches = 0
    while time.perf_counter() - t0 < a.seconds:
        for _ in range(8):
            _lib.check(lib.femasr_clock_probe(st.cuda_stream, a.mfmas, _lib.ptr(ticks)))
        torch.cuda.synchronize()
        launches += 8
    dt = time.perf_counter() - t0
    pw = watch.stop()
    flops = launches * n * (a.mfmas // 4 * 4) * 4096.0
    cyc = float(ticks.double().mean().item())
    print(f'pure fp32 MFMA stream: {n} waves (one per SIMD), {launches} launches in {dt:.2f} s: {flops / dt / 1e12:.1f} TFLOP/s; '
          f'{cyc / (a.mfmas // 4 * 4):.2f} cycles per MFMA; power / clock: {pw}')


if __name__ == '__main__':
    main()
