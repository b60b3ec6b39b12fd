"""A/B of kernel-variant choices (hoig_set_tuning keys) on single convolution launches: interleaved rounds in ONE process on random
data (cdna_hip_programming.md rules 24, 25), HIP events on the launch stream, median and min per variant.
Usage: python tools/ab_conv.py "mfma16=0" "mfma16=1" [--shapes dominant] [--rounds 7] [--iters 20] [--prec bf16x3:f16x2]
Shape sets: 'dominant' (3x3 512->512 at 32x32, 16 and 8 images), 's1' (every 3x3 stride-1 shape of the step with >= 0.2 ms)."""
import argparse
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hoig_amd import ops, _lib as L          # noqa: E402

SHAPES = {
    'dominant': [(16, 512, 512, 32, 32, 3), (8, 512, 512, 32, 32, 3)],
    's1': [(16, 512, 512, 32, 32, 3), (8, 512, 512, 32, 32, 3), (8, 128, 1024, 32, 32, 3), (16, 128, 1024, 32, 32, 3),
           (8, 256, 256, 64, 64, 3), (8, 128, 256, 128, 128, 3), (16, 128, 256, 128, 128, 3), (8, 128, 128, 128, 128, 3),
           (8, 128, 512, 64, 64, 3), (16, 128, 512, 64, 64, 3), (8, 64, 64, 256, 256, 3), (8, 64, 128, 128, 128, 3)],
    'wg': [(16, 512, 512, 32, 32, 3), (8, 512, 512, 32, 32, 3), (8, 128, 1024, 32, 32, 3), (16, 128, 1024, 32, 32, 3),
           (8, 256, 256, 64, 64, 3), (8, 128, 256, 128, 128, 3), (16, 128, 256, 128, 128, 3), (8, 64, 64, 256, 256, 3)],
    # (B, Ci, Co, H, W, k, stride, transposed): the stride-2 3x3 layers of the generator (Conv2d s2 p1 / ConvTranspose2d s2 p1 op1)
    's2': [(16, 64, 128, 256, 256, 3, 2, 0), (16, 128, 256, 128, 128, 3, 2, 0), (16, 256, 512, 64, 64, 3, 2, 0),
           (16, 512, 256, 32, 32, 3, 2, 1), (16, 256, 128, 64, 64, 3, 2, 1), (16, 128, 64, 128, 128, 3, 2, 1),
           (8, 64, 128, 256, 256, 3, 2, 0), (8, 128, 64, 128, 128, 3, 2, 1)],
    's2small': [(2, 64, 128, 256, 256, 3, 2, 0), (4, 64, 128, 256, 256, 3, 2, 0), (2, 128, 256, 128, 128, 3, 2, 0), (4, 128, 256, 128, 128, 3, 2, 0),
                (4, 256, 512, 64, 64, 3, 2, 0), (8, 256, 512, 64, 64, 3, 2, 0)],
    'attn5': [(8, 512, 128, 40, 40, 5), (8, 512, 128, 36, 36, 5), (8, 128, 128, 136, 136, 5), (8, 256, 128, 72, 72, 5)],
}


def parse(v):
    return dict((k, int(x)) for k, x in (kv.split('=') for kv in v.split(',') if kv))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('variants', nargs='+')
    ap.add_argument('--shapes', default='dominant')
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--prec', default='bf16x3:f16x2')
    ap.add_argument('--kinds', default='fwd,dgrad,wgrad')
    a = ap.parse_args()
    ops.set_precision(a.prec)
    variants = [parse(v) for v in a.variants]
    # every key any variant names starts each variant from its default (a key set by one variant must not leak into the next)
    defaults = {k: L.set_tuning(k, -1) for v in variants for k in v}
    variants = [dict(defaults, **v) for v in variants]
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    print('precision %s, %d rounds x %d launches, random data' % (a.prec, a.rounds, a.iters))
    for shp in SHAPES[a.shapes]:
        B, Ci, Co, H, W, k = shp[:6]
        stride, tr = (shp[6], bool(shp[7])) if len(shp) > 6 else (1, False)
        pad = 1 if k == 3 else 0
        if tr:
            Ho, Wo = 2 * H, 2 * W
        else:
            Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        x = torch.randn(B, H, W, Ci, device='cuda')
        w = ops.pack_weight(torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), device='cuda') * 0.02, transposed=tr)
        dy = torch.randn(B, Ho, Wo, Co, device='cuda')
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.zeros_like(w)
        d = L.ConvDesc(B, H, W, Ci, Ho, Wo, Co, k, k, stride, pad, 1 if tr else 0, 0, 0.0, ops.precision)
        d_dg, d_wg = ops._bwd_descs(d)
        hi, lo = ops._packed_planes(w, tr, False)
        thi, tlo = ops._packed_planes(w, tr, True)
        fns = {
            'fwd': lambda: L.call('hoig_conv2d_fwd_packed', ctypes.byref(d), p(x), p(hi), p(lo), None, p(y), st),
            'dgrad': lambda: L.call('hoig_conv2d_bwd_data_packed', ctypes.byref(d_dg), p(dy), p(thi), p(tlo), p(dx), st),
            'wgrad': lambda: ops.wgrad_call('hoig_conv2d_bwd_weight', d_wg, p(x), p(dy), p(dw), None, st),
        }
        if 'wgrad_split' in a.kinds or 'split' in a.kinds or 'dgrad_split' in a.kinds or 'pair' in a.kinds:      # pre-split dy (round 5)
            dys = torch.empty(B, Ho, Wo, 2, Co, dtype=torch.bfloat16, device='cuda')
            fns['split'] = lambda: L.call('hoig_split_planes_bf16', p(dy), p(dys), B * Ho * Wo, Co, st)
            fns['split']()
            fns['wgrad_split'] = lambda: L.call('hoig_conv2d_bwd_weight_split', ctypes.byref(d_wg), p(x), p(dys), p(dw), st)
            if 'pair' in a.kinds:           # grouped launches: this problem twice (two tensors, two weights) as one grid vs one after the other
                x2, dy2, dys2 = torch.randn_like(x), torch.randn_like(dy), torch.empty_like(dys)
                w2 = ops.pack_weight(torch.randn_like(w) * 0.02)
                L.call('hoig_split_planes_bf16', p(dy2), p(dys2), B * Ho * Wo, Co, st)
                hi2, lo2 = ops._packed_planes(w2, tr, False)
                thi2, tlo2 = ops._packed_planes(w2, tr, True)
                y2, dx2, dw2 = torch.empty_like(y), torch.empty_like(dx), torch.zeros_like(dw)
                fns['fwd_2x'] = lambda: (fns['fwd'](), L.call('hoig_conv2d_fwd_packed', ctypes.byref(d), p(x2), p(hi2), p(lo2), None, p(y2), st))
                fns['fwd_pair'] = lambda: L.call('hoig_conv2d_fwd_packed_pair', ctypes.byref(d), p(x), p(x2), p(hi), p(lo), p(hi2), p(lo2), None, None, p(y), p(y2), st)
                fns['dgrad_2x'] = lambda: (L.call('hoig_conv2d_bwd_data_packed_split', ctypes.byref(d_dg), p(dys), p(thi), p(tlo), None, p(dx), st),
                                           L.call('hoig_conv2d_bwd_data_packed_split', ctypes.byref(d_dg), p(dys2), p(thi2), p(tlo2), None, p(dx2), st))
                fns['dgrad_pair'] = lambda: L.call('hoig_conv2d_bwd_data_packed_split_pair', ctypes.byref(d_dg), p(dys), p(dys2), p(thi), p(tlo), p(thi2), p(tlo2), None, None, p(dx), p(dx2), st)
                fns['wgrad_2x'] = lambda: (L.call('hoig_conv2d_bwd_weight_split', ctypes.byref(d_wg), p(x), p(dys), p(dw), st),
                                           L.call('hoig_conv2d_bwd_weight_split', ctypes.byref(d_wg), p(x2), p(dys2), p(dw2), st))
                fns['wgrad_pair'] = lambda: L.call('hoig_conv2d_bwd_weight_split_pair', ctypes.byref(d_wg), p(x), p(x2), p(dys), p(dys2), p(dw), p(dw2), st)
            if hasattr(L.lib, 'hoig_conv2d_bwd_data_packed_split'):
                fns['dgrad_split'] = lambda: L.call('hoig_conv2d_bwd_data_packed_split', ctypes.byref(d_dg), p(dys), p(thi), p(tlo), None, p(dx), st)
        flop = 2.0 * B * (H * W if tr else Ho * Wo) * Ci * Co * k * k
        kinds = [k for k in a.kinds.split(',') if k != 'pair'] + (['fwd_2x', 'fwd_pair', 'dgrad_2x', 'dgrad_pair', 'wgrad_2x', 'wgrad_pair'] if 'pair' in a.kinds else [])
        for kind in kinds:
            fn = fns[kind]
            times = [[] for _ in variants]
            for r in range(a.rounds + 1):
                for vi, v in enumerate(variants):
                    for key, val in v.items():
                        L.set_tuning(key, val)
                    fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    if r:                      # round 0 warms up
                        times[vi].append(e0.elapsed_time(e1) / a.iters * 1e3)
            line = '%-5s %2d %3dx%-3d %4d->%-4d k%d s%d%s |' % (kind, B, H, W, Ci, Co, k, stride, 'T' if tr else ' ')
            for v, t in zip(a.variants, times):
                med, mn = statistics.median(t), min(t)
                line += '  [%s] med %7.1f us %6.1f TF  min %7.1f us |' % (v, med, flop / med / 1e6, mn)
            print(line, flush=True)


if __name__ == '__main__':
    main()
