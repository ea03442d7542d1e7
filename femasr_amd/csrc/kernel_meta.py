"""Register / scratch metadata of every gfx950 kernel in the built objects (llvm-readelf --notes on the code object inside each .o).

    python femasr_amd/csrc/kernel_meta.py [--table out.txt]        # print the table; exit 1 if a kernel outside ALLOW_SCRATCH uses scratch

`build.py` runs `check()` after compiling: a kernel of the DEFAULT launch schedule must not touch scratch memory (spilled VGPRs turn
into HBM traffic: VERDICT r4 found 228 B per thread in the codebook lookup and accumulator tiles spilled inside a Winograd main loop).
Opt-in / experiment instantiations are allow-listed by name prefix.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
# kernels that may use scratch: not on the default schedule (opt-in modes, test hooks, the C-ABI-only instantiations)
ALLOW_SCRATCH = (
    'conv3x3_halo_bf16x3_kernel',       # decoder_math='bf16x3' (secondary mode): 2-8 dwords
)


def device_elf(obj_path):
    """The gfx950 code object inside a hipcc .o (clang offload bundle in .hip_fatbin)."""
    data = open(obj_path, 'rb').read()
    i = data.find(b'__CLANG_OFFLOAD_BUNDLE__')
    if i < 0:
        return None
    n = struct.unpack_from('<Q', data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from('<QQQ', data, off)
        off += 24
        triple = data[off:off + tl].decode()
        off += tl
        if 'gfx950' in triple and sz:
            return data[i + o:i + o + sz]
    return None


def demangle_short(name):
    m = re.match(r'_ZN\d+_GLOBAL__N_1(\d+)', name)
    if m:
        n = int(m.group(1))
        st = m.end()
        return name[st:st + n] + ('<' + name[st + n:][:60] + '>' if name[st + n:].startswith('I') else '')
    return name[:80]


def kernels():
    rows = []
    for f in sorted(os.listdir(HERE)):
        if not f.endswith('.o'):
            continue
        elf = device_elf(os.path.join(HERE, f))
        if not elf:
            continue
        with tempfile.NamedTemporaryFile(suffix='.co') as t:
            t.write(elf)
            t.flush()
            txt = subprocess.run([READELF, '--notes', t.name], capture_output=True, text=True).stdout
        cur = {}
        for line in txt.splitlines():
            m = re.match(r'\s+-?\s*\.(\w+):\s+(.*)$', line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            if k == 'agpr_count' and cur.get('name'):          # (first key of a kernel record in this LLVM's ordering is .agpr_count)
                rows.append(cur)
                cur = {}
            if k in ('name', 'private_segment_fixed_size', 'sgpr_count', 'sgpr_spill_count', 'vgpr_count', 'vgpr_spill_count', 'agpr_count',
                     'group_segment_fixed_size'):
                cur[k] = v
            cur['file'] = f
        if cur.get('name'):
            rows.append(cur)
    # records can interleave: keep complete ones, unique by name
    out, seen = [], set()
    for r in rows:
        if 'name' in r and 'private_segment_fixed_size' in r and r['name'] not in seen:
            seen.add(r['name'])
            out.append(r)
    return out


def table(rows):
    lines = ['%-28s %-84s %5s %5s %5s %7s %7s %8s' % ('object', 'kernel', 'vgpr', 'agpr', 'sgpr', 'vspill', 'sspill', 'scratchB')]
    for r in sorted(rows, key=lambda r: (r['file'], r['name'])):
        lines.append('%-28s %-84s %5s %5s %5s %7s %7s %8s' % (r['file'], demangle_short(r['name'])[:84], r.get('vgpr_count', '?'), r.get('agpr_count', '?'),
                                                               r.get('sgpr_count', '?'), r.get('vgpr_spill_count', '?'), r.get('sgpr_spill_count', '?'),
                                                               r.get('private_segment_fixed_size', '?')))
    return '\n'.join(lines) + '\n'


# kernels ON the default schedule that may touch scratch, with the bytes per thread (a regression gate: the build fails when one grows or
# a new one appears).  Empty since round 5: the Winograd output stages (border blocks fetch their residual rows where they are used) and the
# codebook search (running minima as one row per lane through an explicit LDS pointer) are spill-free; DESIGN.md 5 "scratch".
KNOWN_SCRATCH = {
}


def check(verbose=False):
    rows = kernels()

    def over(r):
        sz = int(r['private_segment_fixed_size'])
        if sz == 0 or any(a in r['name'] for a in ALLOW_SCRATCH):
            return False
        for k, lim in KNOWN_SCRATCH.items():
            if k in r['name']:
                return sz > lim
        return True
    bad = [r for r in rows if over(r)]
    if verbose:
        print(table(rows))
    return rows, bad


OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
VMEM_PREFIXES = ('global_', 'buffer_', 'scratch_', 'flat_', 'tbuffer_')
# kernels_gemm_bf16.hip: what one trip of the MFMA loop may contain: two chunks = four steps = twelve W copies (LDS-DMA) + two 4-load row
# fetches, waits 14 / 10 / 10 per chunk.  The numbers are the source's (`s_waitcnt vmcnt(N)` in the inline asm); check_counted_waits() also
# reads them there.
COUNTED = {'loops': 1, 'global_load_lds_dwordx4': 12, 'global_load_dwordx4': 8, 'vmcnt': {14: 2, 10: 4}}
N_GEMM_BF16S = 9                # instantiations (activation x residual operands, + the 3x3 conv form x residual operands)


def disassemble(obj_path):
    """{kernel symbol: [(address, mnemonic, operands)]} of the gfx950 code object inside a hipcc .o."""
    elf = device_elf(obj_path)
    if not elf:
        return {}
    with tempfile.NamedTemporaryFile(suffix='.co') as t:
        t.write(elf)
        t.flush()
        txt = subprocess.run([OBJDUMP, '-d', t.name], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):', line)
        if m and cur is not None:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def mfma_loops(insts):
    """Innermost backward-branch loops that contain MFMAs: [(first index, branch index)]."""
    addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
    loops = []
    for i, (a, op, args) in enumerate(insts):
        if not op.startswith('s_cbranch') and op != 's_branch':
            continue
        try:
            simm = int(args.split()[0])
        except (ValueError, IndexError):
            continue
        if simm >= 32768:
            simm -= 65536
        if simm >= 0:
            continue
        tgt = addr_index.get(a + 4 + 4 * simm)
        if tgt is None:
            continue
        if any(o.startswith('v_mfma') for _, o, _ in insts[tgt:i]):
            loops.append((tgt, i))
    return [l for l in loops if not any(m != l and l[0] <= m[0] and m[1] <= l[1] for m in loops)]      # innermost only


def check_counted_waits(obj=None):
    """The split GEMM's main loops hold hand-counted `s_waitcnt vmcnt(N)` waits around inline-asm loads / LDS-DMA copies: a VMEM instruction
    the COMPILER adds between them (a spill, a hoisted bias load, a future scheduler) would make the counts wrong and feed stale operands to the
    MFMAs - silently, only the bit-exact GPU tests would tell.  Returns a list of problems (empty = fine)."""
    obj = obj or os.path.join(HERE, 'kernels_gemm_bf16.o')
    problems = []
    src = open(os.path.join(HERE, 'kernels_gemm_bf16.hip')).read()
    src_counts = sorted({int(v) for v in re.findall(r's_waitcnt vmcnt\((\d+)\)', src)})
    want_counts = sorted({0} | set(COUNTED['vmcnt']))
    if src_counts != want_counts:
        problems.append(f'kernels_gemm_bf16.hip waits for vmcnt {src_counts}, kernel_meta.COUNTED expects {want_counts}: update both together')
    seen = 0
    for name, insts in disassemble(obj).items():
        if 'gemm_bf16s_kernel' not in name:
            continue
        seen += 1
        want = COUNTED
        loops = mfma_loops(insts)
        if len(loops) != want['loops']:
            problems.append(f'{demangle_short(name)}: expected {want["loops"]} MFMA loop(s), found {len(loops)}')
            continue
        for lo, hi in loops:
            body = insts[lo:hi + 1]
            vmem, waits = {}, {}
            for _, op, args in body:
                if op.startswith(VMEM_PREFIXES):
                    vmem[op] = vmem.get(op, 0) + 1
                if op == 's_waitcnt':
                    for n in re.findall(r'vmcnt\((\d+)\)', args):
                        waits[int(n)] = waits.get(int(n), 0) + 1
            want_vmem = {k: v for k, v in want.items() if k not in ('vmcnt', 'loops')}
            if vmem != want_vmem:
                problems.append(f'{demangle_short(name)}: VMEM instructions in the MFMA loop {vmem}, the counted waits assume exactly {want_vmem}')
            if waits != want['vmcnt']:
                problems.append(f'{demangle_short(name)}: vmcnt waits in the MFMA loop {waits}, the source has {want["vmcnt"]}')
            if not any(op == 's_barrier' for _, op, _ in body):
                problems.append(f'{demangle_short(name)}: no s_barrier in the MFMA loop')
    if seen < N_GEMM_BF16S:
        problems.append(f'only {seen} gemm_bf16s_kernel instantiations found in {obj}')
    return problems


if __name__ == '__main__':
    rows, bad = check(verbose='--quiet' not in sys.argv)
    if '--table' in sys.argv:
        open(sys.argv[sys.argv.index('--table') + 1], 'w').write(table(rows))
    if bad:
        print('kernels of the default schedule that use scratch memory:')
        print(table(bad))
        sys.exit(1)
    probs = check_counted_waits()
    if probs:
        print('\n'.join(probs))
        sys.exit(1)
