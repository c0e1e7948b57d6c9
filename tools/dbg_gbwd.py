import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fd-gan_amd"), os.path.join(ROOT, "tests")]
import torch
import models.dehaze1113 as net
from oracle import dehaze1113_ref as ref
from oracle.detweights import det_input, fill_state_dict
from hiputil import rel_rms, emulate_kernel_operands
og = ref.FDGAN(); fill_state_dict(og, seed=0)
g = net.FDGAN(); g.load_state_dict(og.state_dict()); g = g.to("cuda:0")
emulate_kernel_operands(og)
x = det_input((2, 3, 64, 64), seed=1234); tgt = det_input((2, 3, 64, 64), seed=4321, lo=-1.0, hi=1.0)
((og(x.clone()) - tgt) ** 2).mean().backward()
y = g(x.to("cuda:0")); ((y - tgt.to("cuda:0")) ** 2).mean().backward(); torch.cuda.synchronize()
order = ["conv_refin3", "trans_block6", "dense_block6", "trans_block5", "dense_block5", "trans_block4", "dense_block4", "conv_refin6",
         "conv_refin5", "trans_block3", "dense_block3.denselayer24", "dense_block3.denselayer23", "dense_block3.denselayer1.", "trans_block2",
         "dense_block2.denselayer12", "dense_block2.denselayer1.", "conv_refine4", "trans_block1", "conv_refin2", "dense_block1.denselayer6", "dense_block1.denselayer1.", "conv_refin1"]
P = dict(g.named_parameters()); Q = dict(og.named_parameters())
for pre in order:
    for k in P:
        if k.startswith(pre) and Q[k].grad is not None:
            print("%-48s err %.4f  |ref| %.3e" % (k, rel_rms(P[k].grad.cpu(), Q[k].grad), float(Q[k].grad.abs().mean())))
