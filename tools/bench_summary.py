"""Pretty-print the per-kernel breakdown of a bench.py JSON line (file argument, or stdin)."""
import json
import sys

for line in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    line = line.strip()
    if not line.startswith('{'):
        print(line[:200])
        continue
    d = json.loads(line)
    r = d.get('roofline', {})
    print('MPix/s', d['value'], 'ms/step', d['ms_per_step'], '| dominant', r.get('kernel'), r.get('achieved'), r.get('frac'), '|', r.get('measured_in', '')[:60])
    for k, v in r.get('per_kernel', {}).items():
        print('   %8.3f ms  %3d x  %6s TF  %s' % (v['ms_per_step'], v['launches_per_step'], v.get('tflops', ''), k))
    for key in ('bf16x3_mode', 'fp32_direct_mode', 'exact_fp32_mode'):
        if key in d:
            print('  ', key, d[key]['value'], 'MPix/s', d[key]['ms_per_step'], 'ms')
    if 'cpu_baseline' in d:
        c = d['cpu_baseline']
        print('   cpu torch', c['value'], c.get('sample', '')[-30:], '| c oracle', c.get('c_oracle', {}).get('value'))
