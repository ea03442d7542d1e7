"""Build libfemasr_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python femasr_amd/csrc/build.py [--force]

Flags that matter for the parity contract: -ffp-contract=off (the kernels place
every fmaf explicitly) and correctly-rounded fp32 divide / sqrt.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['kernels_conv.hip', 'kernels_gemm.hip', 'kernels_gemm_bf16.hip', 'kernels_vq.hip', 'kernels_wino.hip', 'kernels_wino_up2.hip', 'kernels_conv_bf16.hip', 'kernels_misc.hip', 'model.hip']
HEADERS = ['common.h', 'conv_common.h', 'wino_common.h', 'detmath.h', '../../include/femasr_hip.h', '../../include/femasr_hip_debug.h']
SO = os.path.join(HERE, 'libfemasr_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
         '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wno-unused-result']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, gate='fail'):
    """gate: what the scratch-memory check (kernel_meta.py) does when a kernel of the default schedule newly uses scratch: 'fail' raises
    (the developer's build), 'warn' prints the table (the driver's build check: a toolchain drift must not hide the library), 'off' skips it."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            jobs.append((src, subprocess.Popen(cmd)))       # translation units compile in parallel
        objs.append(o)
    failed = [src for src, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError(f'hipcc failed for {failed}')
    if force or _stale(SO, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', SO] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if gate != 'off' and os.environ.get('FEMASR_SKIP_SCRATCH_CHECK') != '1':
        # gates (every build, also one that compiled nothing): kernel_meta.py reads the code objects' metadata and ISA -
        #   * no kernel of the default schedule may (newly) use scratch memory;
        #   * the main loop of the split GEMM holds hand-counted `s_waitcnt vmcnt(N)` waits: no VMEM instruction other than the
        #     inline-asm loads / LDS-DMA copies may appear between them, and the wait values must be the source's.
        sys.path.insert(0, HERE)
        import kernel_meta
        problems = []
        try:
            _, bad = kernel_meta.check()
            if bad:
                problems.append('kernels of the default schedule use (more) scratch memory:\n' + kernel_meta.table(bad) +
                                '(fix the spill, or list the instantiation in kernel_meta.py with the reason)')
            problems += kernel_meta.check_counted_waits()
        except Exception as e:              # (llvm tools missing, another object layout: the gate is a check, not a dependency)
            print(f'kernel_meta: checks skipped ({type(e).__name__}: {e})', flush=True)
        if problems and gate == 'warn':
            print('WARNING: ' + '\n'.join(problems), flush=True)
        elif problems:
            os.remove(SO)                   # a library that failed its gate must not be picked up by a later import
            raise RuntimeError('\n'.join(problems))
    return SO


if __name__ == '__main__':
    build(force='--force' in sys.argv)
