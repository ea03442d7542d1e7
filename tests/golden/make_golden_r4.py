"""Round-4 golden vectors: the reference's WHOLE testset/ directory through the CLI's arithmetic (development container only;
see make_golden.py for the rules - the reference is imported from /root/reference, only DATA is written).

  python tests/golden/make_golden_r4.py          # ~25 min on 8 cores; per-image results are cached under /tmp and reused

BASELINE config 1 = `inference_femasr.py -s 4` on ./testset (38 images, 112x112 .. 800x592; two of them take the
`test_tile(240, 16)` branch, inference_femasr.py:58-63).  For every image, in the CLI's sorted order:
    decode (PIL, as femasr_amd/inference.py does; the reference's cv2 is not installed) -> /255 -> net.test() or net.test_tile()
    -> tensor2img arithmetic (clamp, x255, round) -> uint8 RGB
with the synthetic 'trained' weights of seed 12 (the seed of png_OST_120_tiled, whose numbers this file must reproduce).
Stored in testset_all.npz: the image FILE BYTES (data; concatenated), and per image the SHA-256 of the uint8 output, the fp32
output and the uint8 output strided by 16, the fp32 mean, the VQ index map of every `test()` call in call order (one per
un-tiled image, one per tile otherwise) and, for every token whose runner-up distance is within 16 ulp of the best in the
REFERENCE's own arithmetic, its position, the reference's 8 best codes and their gaps in ulp (the near-tie rule's evidence).
"""
import hashlib
import io
import os
import pickle
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from make_golden_r2 import build_ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
TESTSET = '/root/reference/testset'
SEED = 12
CACHE = '/tmp/golden_r4_cache'
STRIDE = 16


def run_image(net, fname):
    from PIL import Image
    raw = open(os.path.join(TESTSET, fname), 'rb').read()
    rgb = np.asarray(Image.open(io.BytesIO(raw)).convert('RGB'))
    h, w = rgb.shape[:2]
    x = torch.from_numpy(rgb.transpose(2, 0, 1).astype(np.float32)).unsqueeze(0) / 255.
    calls = []

    def vq_hook(mod, inp, out):
        z = inp[0].detach()
        zf = z.permute(0, 2, 3, 1).reshape(-1, mod.e_dim)
        d = mod.dist(zf, mod.embedding.weight.detach())
        top = torch.topk(d, 8, dim=1, largest=False)
        calls.append(dict(hw=tuple(z.shape[2:]), idx=out[2].detach().numpy().reshape(-1).astype(np.int16),
                          top_d=top.values.numpy(), top_i=top.indices.numpy().astype(np.int16)))
    hk = net.quantize_group[0].register_forward_hook(vq_hook)
    tiled = h * w >= 600 ** 2                      # inference_femasr.py:57-63 (max_size 600)
    with torch.no_grad():
        y = net.test_tile(x) if tiled else net.test(x)
    hk.remove()
    t = y.squeeze(0).float().clone().clamp_(0, 1)                          # basicsr/utils/img_util.py:66-90 (RGB kept: the file order is the CLI's business)
    out_u8 = (t.numpy().transpose(1, 2, 0) * 255.0).round().astype(np.uint8)
    yn = y.numpy()
    return dict(name=fname, raw=raw, hw=(h, w), tiled=tiled, calls=calls, out_u8=out_u8, yn=yn)


def main():
    torch.set_num_threads(int(os.environ.get('GOLDEN_THREADS', '8')))
    os.makedirs(CACHE, exist_ok=True)
    names = sorted(os.listdir(TESTSET))
    cfg = dict(codebook_params=[[32, 1024, 512]], LQ_stage=True, scale_factor=4)
    net = None
    recs = []
    for i, fn in enumerate(names):
        cp = os.path.join(CACHE, fn + '.pkl')
        if os.path.exists(cp):
            r = pickle.load(open(cp, 'rb'))
        else:
            if net is None:
                net = build_ref(cfg, SEED, 'trained', 'default')
            t0 = time.time()
            full = run_image(net, fn)
            top_d = np.concatenate([c['top_d'] for c in full['calls']])
            top_i = np.concatenate([c['top_i'] for c in full['calls']])
            gap = (top_d[:, 1] - top_d[:, 0]) / np.spacing(np.abs(top_d[:, 0]))
            near = np.nonzero(gap <= 16.0)[0]
            r = dict(name=fn, raw=full['raw'], hw=full['hw'], tiled=full['tiled'],
                     call_hw=np.array([c['hw'] for c in full['calls']], np.int32),
                     indices=np.concatenate([c['idx'] for c in full['calls']]),
                     near_pos=near.astype(np.int64), near_codes=top_i[near],
                     near_gaps=((top_d[near] - top_d[near, :1]) / np.spacing(np.abs(top_d[near, :1]))).astype(np.float32),
                     n_within2=int((gap <= 2).sum()), n_exact=int((gap == 0).sum()),
                     sha=hashlib.sha256(full['out_u8'].tobytes()).hexdigest(),
                     out_shape=np.array(full['yn'].shape), out_mean=float(full['yn'].astype(np.float64).mean()),
                     out_absmax=float(np.abs(full['yn']).max()),
                     f32s=full['yn'][0, :, ::STRIDE, ::STRIDE].copy(), u8s=full['out_u8'][::STRIDE, ::STRIDE].copy())
            pickle.dump(r, open(cp, 'wb'))
            print(f'[{i + 1}/{len(names)}] {fn}: {r["hw"]} {"tiled" if r["tiled"] else ""} {len(r["indices"])} tokens, {r["n_within2"]} within 2 ulp of a tie '
                  f'({r["n_exact"]} exact), {len(r["near_pos"])} within 16; {time.time() - t0:.0f} s', flush=True)
        recs.append(r)
    cat = lambda key, dt=None: np.concatenate([np.asarray(r[key]).reshape(-1) if dt is None else np.asarray(r[key], dt).reshape(-1) for r in recs])
    off = lambda key: np.cumsum([0] + [np.asarray(r[key]).size for r in recs]).astype(np.int64)
    tok_off = off('indices')
    np.savez_compressed(
        os.path.join(OUT, 'testset_all.npz'), seed=np.array(SEED), codebook=np.array('trained'), stride=np.array(STRIDE),
        tile_size=np.array(240), tile_pad=np.array(16),
        names=np.array([r['name'] for r in recs]), file_bytes=np.frombuffer(b''.join(r['raw'] for r in recs), np.uint8),
        file_off=np.cumsum([0] + [len(r['raw']) for r in recs]).astype(np.int64),
        hw=np.array([r['hw'] for r in recs], np.int32), tiled=np.array([r['tiled'] for r in recs]),
        n_calls=np.array([len(r['call_hw']) for r in recs], np.int32), call_hw=np.concatenate([r['call_hw'] for r in recs]),
        indices=cat('indices'), token_off=tok_off,
        near_pos=np.concatenate([r['near_pos'] + tok_off[i] for i, r in enumerate(recs)]),       # global token positions
        near_codes=np.concatenate([r['near_codes'] for r in recs]), near_gaps_ulp=np.concatenate([r['near_gaps'] for r in recs]),
        out_sha256=np.array([r['sha'] for r in recs]), out_shape=np.array([r['out_shape'] for r in recs]),
        out_mean=np.array([r['out_mean'] for r in recs]), out_absmax=np.array([r['out_absmax'] for r in recs]),
        f32_strided=cat('f32s'), f32_off=off('f32s'), u8_strided=cat('u8s'), u8_off=off('u8s'))
    tot = int(tok_off[-1])
    print(f'testset_all: {len(recs)} images, {tot} tokens, {sum(r["n_within2"] for r in recs)} within 2 ulp of a tie in the reference, '
          f'{sum(r["n_exact"] for r in recs)} exact ties; file {os.path.getsize(os.path.join(OUT, "testset_all.npz")) / 1e6:.1f} MB')


if __name__ == '__main__':
    main()
