"""bench.PowerWatch: the amdgpu hwmon reader behind the `power` object of the bench line (host-only logic, no GPU)."""
import os

import bench


def _fake_card(root, name, pci, power_uw, sclk_hz):
    dev = root / 'devices' / f'pci0000:{pci[1]:02x}' / f'{pci[0]:04x}:{pci[1]:02x}:{pci[2]:02x}.0'
    hw = dev / 'hwmon' / 'hwmon7'
    hw.mkdir(parents=True)
    (hw / 'power1_input').write_text(f'{power_uw}\n')
    (hw / 'power1_cap').write_text('1400000000\n')
    (hw / 'freq1_input').write_text(f'{sclk_hz}\n')
    card = root / 'class' / 'drm' / name
    card.mkdir(parents=True)
    os.symlink(dev, card / 'device')


def test_power_watch_reads_the_card_with_the_matching_pci_address(tmp_path, monkeypatch):
    """/sys/class/drm lists every card of the host, also inside a one-GPU container (measured: the first card is somebody else's):
    the sampler must pick the card whose PCI address torch reports, read power1_input (uW) / freq1_input (Hz) / power1_cap."""
    import glob as _glob
    _fake_card(tmp_path, 'card0', (0, 0x72, 0), 292000000, 2401000000)      # another GPU of the host
    _fake_card(tmp_path, 'card16', (0, 0x5a, 0), 1371000000, 2370000000)    # ours
    real_glob = _glob.glob
    monkeypatch.setattr(_glob, 'glob', lambda pat, *a, **k: real_glob(pat.replace('/sys/class/drm', str(tmp_path / 'class' / 'drm')), *a, **k))
    w = bench.PowerWatch((0, 0x5a, 0))
    assert w.dir is not None and 'card16' in w.dir
    w.start()
    import time
    time.sleep(0.2)
    got = w.stop()
    assert got['package_w_mean'] == 1371.0 and got['cap_w'] == 1400.0 and got['sclk_mhz_median'] == 2370.0 and got['samples'] >= 2
    assert bench.PowerWatch((0, 0x11, 0)).dir is None        # no such card: nothing is read from sysfs (rocm-smi fallback)
    assert bench.PowerWatch(None).dir is None
