"""Thin wrappers over the C ABI for the GPU parity tests (torch only moves memory)."""
import ctypes

import numpy as np
import torch

from femasr_amd import _lib


def dev(a, device='cuda'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def lib_weight_layout(w_khwc):
    """Oracle layout [kh][kw][Cin][Cout] -> the library's packed FRAGMENT-MAJOR layout (numpy restatement of
    femasr_repack_oihw, include/femasr_hip.h): out[q][ntile][lane][kk], zero padded."""
    kh, kw, cin, cout = w_khwc.shape
    if kh == 1 and kw == 1 and cin % 32 == 0:       # GEMM layout of the 1x1 / linear layers: [q][ntile][j][lane][t]
        npad = (cout + 31) // 32 * 32
        full = np.zeros((cin, npad), np.float32)
        full[:, :cout] = w_khwc.reshape(cin, cout)
        # k = 32q + 8j + 4h + t, n = 32*ntile + c:  [q][j][h][t][ntile][c] -> [q][ntile][j][h][c][t]
        t = full.reshape(cin // 32, 4, 2, 4, npad // 32, 32).transpose(0, 4, 1, 2, 5, 3)
        return np.ascontiguousarray(t).reshape(-1)
    if cin % 32 == 0:       # K order: channel blocks of 32 outermost
        rows = w_khwc.reshape(kh, kw, cin // 32, 32, cout).transpose(2, 0, 1, 3, 4).reshape(-1, cout)
    else:
        rows = w_khwc.reshape(-1, cout)
    k = rows.shape[0]
    kp, npad = (k + 31) // 32 * 32, (cout + 31) // 32 * 32
    full = np.zeros((kp, npad), np.float32)
    full[:k, :cout] = rows
    # [q][kk][kslot][ntile][col] -> [q][ntile][kslot][col][kk]
    t = full.reshape(kp // 32, 16, 2, npad // 32, 32).transpose(0, 3, 2, 4, 1)
    out = np.ascontiguousarray(t).reshape(-1)
    if cout <= 4 and kh == 3 and kw == 3 and cin % 32 == 0:       # compact [k][4] copy for the direct (VALU) out_conv kernel
        tail = np.zeros((k, 4), np.float32)
        tail[:, :cout] = rows
        out = np.concatenate((out, tail.reshape(-1)))
    return out


def conv2d(x, w_khwc, bias, ksz, stride=1, pad=0, up2=False, prologue=0, pro=(None, None, None), act=0,
           res1=None, res2=None, bf16x3=False, gn_part=False, wino=False, fast_act=False, in_add=None, bf16s=False):
    """x NHWC numpy -> numpy, through femasr_conv2d (weights given in the oracle's [kh][kw][Cin][Cout] layout).
    bf16s: the 1x1 / Linear layer on the bf16 matrix pipe (femasr_conv_args.w_bf16s, three-term split)."""
    lib = _lib.load()
    cout = np.asarray(w_khwc).shape[-1]
    wlin3 = None
    if bf16s and ksz == 3:       # the 3x3 conv as the split-bf16 GEMM over K = 9 Cin: planes packed on the GPU from OIHW
        w_oihw = dev(np.ascontiguousarray(np.asarray(w_khwc).transpose(3, 2, 0, 1)))
        wlin3 = torch.empty(int(lib.femasr_packed_weight_conv3x3_bf16s_bytes(cout, w_oihw.shape[1])), dtype=torch.uint8, device='cuda')
        _lib.check(lib.femasr_repack_oihw_bf16s(None, _lib.ptr(w_oihw), cout, w_oihw.shape[1], _lib.ptr(wlin3)))
    elif bf16s:
        w_oi = dev(np.ascontiguousarray(np.asarray(w_khwc).reshape(-1, cout).T))
        wlin3 = torch.empty(int(lib.femasr_packed_weight_bf16s_bytes(cout, w_oi.shape[1])), dtype=torch.uint8, device='cuda')
        _lib.check(lib.femasr_repack_k1_bf16s(None, _lib.ptr(w_oi), cout, w_oi.shape[1], _lib.ptr(wlin3)))
    wsplit = None
    if bf16x3:      # opt-in split-bf16 path: weights repacked on the GPU from OIHW
        w_oihw = dev(np.ascontiguousarray(np.asarray(w_khwc).transpose(3, 2, 0, 1)))
        o_, i_, kh_, kw_ = w_oihw.shape
        wsplit = torch.empty(int(lib.femasr_packed_weight_bf16x3_bytes(o_, i_, kh_, kw_)), dtype=torch.uint8, device='cuda')
        _lib.check(lib.femasr_repack_oihw_bf16x3(None, _lib.ptr(w_oihw), o_, i_, kh_, kw_, _lib.ptr(wsplit)))
    b, h, w, cin = x.shape
    wup2 = None
    if up2 and ksz == 3 and stride == 1 and pad == 1 and cin % 32 == 0 and not bf16x3:    # phase-filter form of nearest-x2 + conv
        w_oihw = dev(np.ascontiguousarray(np.asarray(w_khwc).transpose(3, 2, 0, 1)))
        wup2 = torch.empty(int(lib.femasr_up2_weight_floats(cout, cin)), dtype=torch.float32, device='cuda')
        _lib.check(lib.femasr_repack_oihw_up2(None, _lib.ptr(w_oihw), cout, cin, _lib.ptr(wup2)))
    wwino = None
    if wino:        # Winograd-domain weights, transformed and packed on the GPU from OIHW
        w_oihw = dev(np.ascontiguousarray(np.asarray(w_khwc).transpose(3, 2, 0, 1)))
        if up2:     # the 25-component form of nearest-x2 + conv (kernels_wino_up2.hip)
            wwino = torch.empty(int(lib.femasr_wino_up2_weight_floats(cout, cin)), dtype=torch.float32, device='cuda')
            _lib.check(lib.femasr_repack_oihw_wino_up2(None, _lib.ptr(w_oihw), cout, cin, _lib.ptr(wwino)))
        else:
            wwino = torch.empty(int(lib.femasr_wino_weight_floats(cout, cin)), dtype=torch.float32, device='cuda')
            _lib.check(lib.femasr_repack_oihw_wino(None, _lib.ptr(w_oihw), cout, cin, _lib.ptr(wwino)))
    w_khwc = lib_weight_layout(np.asarray(w_khwc))
    hv, wv = (2 * h, 2 * w) if up2 else (h, w)
    ho, wo = (hv + 2 * pad - ksz) // stride + 1, (wv + 2 * pad - ksz) // stride + 1
    tx, tw, tb = dev(x), dev(w_khwc), dev(bias)
    tp = [None if p is None else dev(p) for p in pro]
    t1 = None if res1 is None else dev(res1)
    t2 = None if res2 is None else dev(res2)
    out = torch.full((b, ho, wo, cout), float('nan'), dtype=torch.float32, device='cuda')
    a = _lib.ConvArgs()
    a.in_ = tx.data_ptr(); a.B, a.H, a.W, a.Cin = b, h, w, cin
    a.w = tw.data_ptr(); a.bias = tb.data_ptr()
    a.Cout, a.ksz, a.stride, a.pad, a.up2, a.prologue = cout, ksz, stride, pad, int(up2), prologue
    a.pro_a = None if tp[0] is None else tp[0].data_ptr()
    a.pro_b = None if tp[1] is None else tp[1].data_ptr()
    a.pro_c = None if tp[2] is None else tp[2].data_ptr()
    a.act = act
    a.res1 = None if t1 is None else t1.data_ptr()
    a.res2 = None if t2 is None else t2.data_ptr()
    a.out = out.data_ptr(); a.Ho, a.Wo = ho, wo
    a.w_bf16x3 = None if wsplit is None else wsplit.data_ptr()
    a.w_up2 = None if wup2 is None else wup2.data_ptr()
    a.w_wino = None if wwino is None else wwino.data_ptr()
    a.fast_act = int(fast_act)      # 0 exact, 1 hardware SiLU
    a.w_bf16s = None if wlin3 is None else wlin3.data_ptr()
    tadd = None if in_add is None else dev(in_add)
    a.in_add = None if tadd is None else tadd.data_ptr()
    part = None
    if gn_part:
        tiles = ((ho + 7) // 8) * ((wo + 15) // 16)
        if wup2 is not None:        # phase-filter x2 conv: one partial per half-resolution tile and phase
            tiles = 4 * ((h + 7) // 8) * ((w + 15) // 16)
        if wwino is not None:       # Winograd conv: one partial per 16x16-pixel sub-block
            tiles = ((ho + 15) // 16) * ((wo + 15) // 16)
        part = torch.full((b, tiles, 32, 2), float('nan'), dtype=torch.float64, device='cuda')
        a.gn_part = part.data_ptr()
    _lib.check(lib.femasr_conv2d(None, ctypes.byref(a)))
    torch.cuda.synchronize()
    if gn_part:
        return out.cpu().numpy(), part
    return out.cpu().numpy()


def gn_silu_apply(x, a, b):
    """silu(fmaf(x, a[n][c], b[n][c])) on an NHWC array through femasr_gn_silu_apply."""
    lib = _lib.load()
    bb, h, w, c = x.shape
    tx, ta, tb = dev(x), dev(a), dev(b)
    y = torch.full(x.shape, float('nan'), dtype=torch.float32, device='cuda')
    _lib.check(lib.femasr_gn_silu_apply(None, _lib.ptr(tx), bb, h, w, c, _lib.ptr(ta), _lib.ptr(tb), _lib.ptr(y)))
    torch.cuda.synchronize()
    return y.cpu().numpy()


def gn_coeffs(x, gamma, beta, eps=1e-6, groups=32):
    lib = _lib.load()
    b, h, w, c = x.shape
    tx, tg, tb = dev(x), dev(gamma), dev(beta)
    a = torch.empty((b, c), dtype=torch.float32, device='cuda')
    bb = torch.empty((b, c), dtype=torch.float32, device='cuda')
    scratch = torch.empty(int(lib.femasr_gn_scratch_bytes(b, h, w, c, groups)), dtype=torch.uint8, device='cuda')
    _lib.check(lib.femasr_gn_coeffs(None, _lib.ptr(tx), b, h, w, c, groups, _lib.ptr(tg), _lib.ptr(tb), eps,
                                    _lib.ptr(a), _lib.ptr(bb), _lib.ptr(scratch)))
    torch.cuda.synchronize()
    return a.cpu().numpy(), bb.cpu().numpy()


def gn_coeffs_from_partials(part, h, w, c, gamma, beta, eps=1e-6):
    lib = _lib.load()
    b, tiles = part.shape[0], part.shape[1]
    tg, tb = dev(gamma), dev(beta)
    a = torch.empty((b, c), dtype=torch.float32, device='cuda')
    bb = torch.empty((b, c), dtype=torch.float32, device='cuda')
    _lib.check(lib.femasr_gn_coeffs_from_partials(None, _lib.ptr(part), b, tiles, h, w, c, 32, _lib.ptr(tg), _lib.ptr(tb), eps,
                                                  _lib.ptr(a), _lib.ptr(bb)))
    torch.cuda.synchronize()
    return a.cpu().numpy(), bb.cpu().numpy()


def ln_stats(x, eps=1e-5):
    lib = _lib.load()
    rows, c = x.shape
    tx = dev(x)
    st = torch.empty((rows, 2), dtype=torch.float32, device='cuda')
    _lib.check(lib.femasr_ln_stats(None, _lib.ptr(tx), rows, c, eps, _lib.ptr(st)))
    torch.cuda.synchronize()
    return st.cpu().numpy()


def layernorm(x, gamma, beta, eps=1e-5):
    lib = _lib.load()
    rows, c = x.shape
    tx, tg, tb = dev(x), dev(gamma), dev(beta)
    y = torch.full((rows, c), float('nan'), dtype=torch.float32, device='cuda')
    _lib.check(lib.femasr_layernorm(None, _lib.ptr(tx), rows, c, _lib.ptr(tg), _lib.ptr(tb), eps, _lib.ptr(y)))
    torch.cuda.synchronize()
    return y.cpu().numpy()


def window_attention(qkv, b, h, w, c, heads, shift, table):
    lib = _lib.load()
    tq, tt = dev(qkv), dev(table)
    out = torch.full((b, h * w, c), float('nan'), dtype=torch.float32, device='cuda')
    _lib.check(lib.femasr_window_attention(None, _lib.ptr(tq), b, h, w, c, heads, shift, _lib.ptr(tt), _lib.ptr(out)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def vq(z_rows, codebook, mode='gemm'):
    """mode 'gemm': femasr_vq (single pass, fp32 MFMA); 'twopass': femasr_vq_twopass (bf16 candidates + exact re-check)."""
    lib = _lib.load()
    m, d = z_rows.shape
    n_e = codebook.shape[0]
    tz, tcb = dev(z_rows), dev(codebook)
    cbt = torch.empty(int(lib.femasr_packed_weight_floats(n_e, d, 1, 1)), dtype=torch.float32, device='cuda')
    ee = torch.empty((n_e,), dtype=torch.float32, device='cuda')
    _lib.check(lib.femasr_repack_oihw(None, _lib.ptr(tcb), n_e, d, 1, 1, _lib.ptr(cbt)))
    _lib.check(lib.femasr_row_sqsum(None, _lib.ptr(tcb), n_e, d, _lib.ptr(ee)))
    idx = torch.full((m,), -1, dtype=torch.int64, device='cuda')
    zq = torch.full((m, d), float('nan'), dtype=torch.float32, device='cuda')
    scratch = torch.empty(int(lib.femasr_vq_scratch_bytes(m, n_e)), dtype=torch.uint8, device='cuda')
    if mode == 'gemm':
        _lib.check(lib.femasr_vq(None, _lib.ptr(tz), m, d, _lib.ptr(tcb), _lib.ptr(cbt), _lib.ptr(ee), n_e,
                                 _lib.ptr(idx), _lib.ptr(zq), _lib.ptr(scratch)))
    else:
        assert lib.femasr_vq_twopass_ok(n_e, d)
        aux = torch.empty(int(lib.femasr_vq_aux_bytes(n_e, d)), dtype=torch.uint8, device='cuda')
        _lib.check(lib.femasr_vq_prepare(None, _lib.ptr(tcb), _lib.ptr(ee), n_e, d, _lib.ptr(aux)))
        _lib.check(lib.femasr_vq_twopass(None, _lib.ptr(tz), m, d, _lib.ptr(tcb), _lib.ptr(aux), _lib.ptr(ee), n_e,
                                         _lib.ptr(idx), _lib.ptr(zq), _lib.ptr(scratch)))
    torch.cuda.synchronize()
    return idx.cpu().numpy(), zq.cpu().numpy(), cbt.cpu().numpy(), ee.cpu().numpy()


def vq_candidates(z_rows, codebook):
    """Pass 1 of the two-pass search: (cand (M,32) uint16, cnt (M,) uint16; 0xFFFF = every code)."""
    lib = _lib.load()
    m, d = z_rows.shape
    n_e = codebook.shape[0]
    tz, tcb = dev(z_rows), dev(codebook)
    ee = torch.empty((n_e,), dtype=torch.float32, device='cuda')
    _lib.check(lib.femasr_row_sqsum(None, _lib.ptr(tcb), n_e, d, _lib.ptr(ee)))
    aux = torch.empty(int(lib.femasr_vq_aux_bytes(n_e, d)), dtype=torch.uint8, device='cuda')
    _lib.check(lib.femasr_vq_prepare(None, _lib.ptr(tcb), _lib.ptr(ee), n_e, d, _lib.ptr(aux)))
    cand = torch.zeros((m, 32), dtype=torch.int16, device='cuda')
    cnt = torch.zeros((m,), dtype=torch.int16, device='cuda')
    _lib.check(lib.femasr_vq_candidates(None, _lib.ptr(tz), m, d, _lib.ptr(aux), _lib.ptr(ee), n_e, _lib.ptr(cand), _lib.ptr(cnt)))
    torch.cuda.synchronize()
    return cand.cpu().numpy().view('uint16'), cnt.cpu().numpy().view('uint16')


def build_net(cfg_name, weights, device='cuda', decoder_math='fp32_strict', linear_math='bf16_split'):
    """decoder_math defaults to 'fp32_strict' HERE (the tests' bit-exact comparisons with the oracle): the product default
    'fp32' runs the SiLU of the Winograd convs on the hardware exp2 / rcp units and is compared with a tolerance instead
    (test_gpu_network.py::test_default_mode_vs_oracle_and_reference)."""
    from femasr_amd.archs import build_network
    from helpers import CONFIGS
    net = build_network(dict(type='FeMaSRNet', **CONFIGS[cfg_name]))
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=False)
    assert not missing.unexpected_keys
    net.decoder_math = decoder_math
    net.linear_math = linear_math
    return net.to(device).eval()
