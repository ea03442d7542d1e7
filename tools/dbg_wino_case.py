import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from femasr_amd import synth
from oracle import oracle as orc
import gpu_utils as G
cin, cout, (b, h, w) = 32, 192, (2, 5, 33)
x = synth.uniform(21, 'wix', (b, h, w, cin), -2.0, 2.0)
wt = synth.uniform(21, 'wiw', (3, 3, cin, cout), -0.1, 0.1)
bias = synth.uniform(21, 'wib', (cout,), -0.5, 0.5)
y = G.conv2d(x, wt, bias, 3, 1, 1, wino=True)
yr = orc.conv2d(x, wt, bias, 3, 1, 1, wino=True)
bad = np.argwhere(y != yr)
print('mismatches', len(bad)); 
import collections
print(collections.Counter((int(n), int(yy), int(xx)) for n, yy, xx, c in bad).most_common(12))
print('channels', sorted(set(int(c) for *_, c in bad))[:40])
n, yy, xx, c = bad[0]; print(y[n, yy, xx, c:c+8], yr[n, yy, xx, c:c+8])
# where do the wrong values come from?
for (n, yy, xx, c) in bad[:6]:
    v = y[n, yy, xx, c]
    hits = np.argwhere(np.isclose(yr, v, rtol=0, atol=1e-6))
    nb = yr[n, yy, xx, c] - bias[c]
    hits2 = np.argwhere(np.isclose(yr - bias[None, None, None, :], v - bias[c], rtol=0, atol=1e-6))
    print((n, yy, xx, c), 'got', v, 'ref', yr[n, yy, xx, c], 'same value in ref at', hits[:4].tolist(), 'same pre-bias at', hits2[:4].tolist())
