"""Which 16-bit storage format can meet the north star's metric budget?  CPU experiment on the fp32 oracle.

The oracle generator (oracle/dehaze1113_ref.FDGAN, deterministic weights, train-mode BatchNorm) is run with every
convolution's filter, (activated) input and stored output rounded to a 16-bit format -- exactly the three places the HIP
kernels round -- per stage and per format, and the output is compared with the unrounded fp32 run.  What it showed
(profiles/r3_precision_study.txt; B=2 @ 256x256):

    bf16 everywhere                     46.8 dB   (the HIP path of rounds 1-2 measured 47.2 dB: the kernels add nothing)
    bf16 only in the 64x64 / 32x32 stages  62.1 dB   -> keeping the DEEP stages in fp32 would change nothing (46.9 dB):
                                                     the error is made in the 256x256 / 128x128 stages and amplified
    fp16 everywhere                     64.4 dB
    fp16 activations, bf16 filters      51.7 dB;  bf16 activations, fp16 filters 48.1 dB  (both operands matter)

|PSNR_hip - PSNR_ref| <= 0.02 dB against a ground truth the reference scores at 30 dB needs >= 53.4 dB between the two
outputs (25 dB: 48.3 dB).  So the forward pass stores fp16 (same bytes, same MFMA rate); gradients stay bf16.

Run: python tools/precision_study.py [batch] [size]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch
import torch.nn as nn

from oracle import dehaze1113_ref as ref
from oracle.detweights import det_input, fill_state_dict

DEEP = ("dense_block3", "trans_block3", "conv_refin5", "conv_refin6", "dense_block4", "trans_block4")


def make(sel_w, sel_in, sel_out, dtw, dta):
    g = ref.FDGAN()
    fill_state_dict(g, seed=0)
    g.train()
    rw, ra = (lambda t: t.to(dtw).float()), (lambda t: t.to(dta).float())
    for name, m in g.named_modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if sel_w(name):
                with torch.no_grad():
                    m.weight.copy_(rw(m.weight))
            if sel_in(name):
                m.register_forward_pre_hook(lambda mod, inp: (ra(inp[0]),))
            if sel_out(name):
                m.register_forward_hook(lambda mod, inp, out: ra(out))
        if isinstance(m, (nn.ReLU, nn.LeakyReLU)):
            m.inplace = False
    return g


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    x = det_input((B, 3, S, S), seed=1234)
    g0 = ref.FDGAN()
    fill_state_dict(g0, seed=0)
    g0.train()
    with torch.no_grad():
        t0 = {}
        y0 = g0(x.clone(), taps=t0)
    ALL, NONE = (lambda n: True), (lambda n: False)
    deep = lambda n: any(n.startswith(p) for p in DEEP)
    shallow = lambda n: not deep(n)
    bf, fh = torch.bfloat16, torch.float16
    print("oracle FDGAN, train-mode BatchNorm, batch %d @ %dx%d; PSNR of the rounded run vs the fp32 run (peak-to-peak 2), "
          "rel-rms of the taps" % (B, S, S))
    for label, args in (("bf16: filter, input, stored output", (ALL, ALL, ALL, bf, bf)),
                        ("bf16: filters only", (ALL, NONE, NONE, bf, bf)),
                        ("bf16: conv inputs only", (NONE, ALL, NONE, bf, bf)),
                        ("bf16: stored outputs only", (NONE, NONE, ALL, bf, bf)),
                        ("bf16 in the 256^2/128^2 stages, fp32 deep", (shallow, shallow, shallow, bf, bf)),
                        ("bf16 in the 64^2/32^2 stages only", (deep, deep, deep, bf, bf)),
                        ("fp16: filter, input, stored output", (ALL, ALL, ALL, fh, fh)),
                        ("bf16 filters, fp16 activations", (ALL, ALL, ALL, bf, fh)),
                        ("fp16 filters, bf16 activations", (ALL, ALL, ALL, fh, bf))):
        g = make(*args)
        with torch.no_grad():
            t = {}
            y = g(x.clone(), taps=t)
        psnr = 10 * np.log10(4.0 / float(((y.double() - y0.double()) ** 2).mean()))
        rr = " ".join("%s %.2f%%" % (k, 100 * float(((t[k] - t0[k]) ** 2).mean().sqrt() / (t0[k] ** 2).mean().sqrt()))
                      for k in ("x0", "x1", "x2", "x3", "x4", "x5", "x6"))
        print("%-46s %6.2f dB   %s" % (label, psnr, rr), flush=True)


if __name__ == "__main__":
    main()
