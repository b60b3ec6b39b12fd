"""Bisect helper: run-to-run differences of the eval forward per generator output and per sample (TEST INFRASTRUCTURE)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from common import product_trainer
from hoig_amd import ops
ops.set_precision(os.environ.get('PREC', 'bf16x3'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = product_trainer('generator_spade_attn', B, 256)
m.set_eval()
n = m._n
def run():
    with torch.no_grad():
        outs = m._G.forward_nhwc(n['input_G_bg'], n['src_obj_rgb'], n['tsf_obj_rgb'], n['src_hand_rgb'], n['tsf_hand_rgb'], n['T'],
                                 n['src_obj_cond'], n['src_hand_cond'], n['tsf_obj_cond'], n['tsf_hand_cond'], n.get('armask_src'), n.get('armask_tsf'))
        torch.cuda.synchronize()
        return [o.clone() for o in outs]
a, b = run(), run()
names = ['src_bg', 'tsf_bg', 'src_obj', 'src_hand', 'src_mbg', 'src_mh', 'tsf_obj', 'tsf_hand', 'tsf_mbg', 'tsf_mh']
for i, nm in enumerate(names):
    d = (a[i] - b[i]).abs().flatten(1).max(1)[0] / a[i].abs().max()
    bad = [(j, '%.1e' % float(v)) for j, v in enumerate(d) if v > 1e-3]
    print('%-9s max %.2e  bad samples %s' % (nm, float(d.max()), bad[:12]))
inp = n['src_obj_rgb']
print('src_obj_rgb per-sample abs max', [round(float(v), 3) for v in inp.abs().flatten(1).max(1)[0]])
print('tsf_obj_rgb per-sample abs max', [round(float(v), 3) for v in n['tsf_obj_rgb'].abs().flatten(1).max(1)[0]])
