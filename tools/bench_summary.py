import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        r = d.get('roofline', {})
        print('MPix/s', d['value'], 'ms/step', d['ms_per_step'], '| dominant', r.get('kernel'), r.get('achieved'), r.get('frac'))
        for k, v in r.get('per_kernel_ms_per_step', {}).items():
            print('   %8.3f  %s' % (v, k))
        if 'cpu_baseline' in d:
            print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline']['max_abs_vs_gpu'])
    else:
        print(line[:200])
