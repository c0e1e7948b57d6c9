"""Sustained timing of the MFMA-bound data gradients (fdgan_conv2d_bwd_data: the forward kernel on dy with the flipped filter + the masked
row-phase epilogue) next to the forward conv of the same shape: bwd_mfma_bench.py [reps].  VGG16 layers (activation-only prologue, the
input's only consumer: store mode) and D's 4x4 144 -> 288 (BatchNorm + LeakyReLU prologue)."""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "fd-gan_amd"))
from fdgan_hip import engine as E, lib as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, dev = 16, "cuda"
SHAPES = [("VGG conv1_2", 256, 64, 64, 3, 1, False), ("VGG conv2_2", 128, 128, 128, 3, 1, False), ("VGG conv3_2", 64, 256, 256, 3, 1, False),
          ("VGG conv4_2", 32, 512, 512, 3, 1, False), ("D layer3 72->144", 128, 72, 144, 3, 1, True), ("D layer4 144->288", 128, 144, 288, 4, 1, True)]


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, hw, cin, cout, ks, pad, bn in SHAPES:
    ho = hw + 2 * pad - ks + 1
    x = torch.randn(N, hw, hw, cin, device=dev).half()
    y = torch.empty(N, ho, ho, (cout + 7) // 8 * 8, device=dev).half()
    G = torch.zeros(N, hw, hw, cin, device=dev).bfloat16()
    dy = torch.randn(N, ho, ho, (cout + 7) // 8 * 8, device=dev).bfloat16()
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.05
    pwf = E.PackedWeight(w, cout, cin, ks)
    pwf.pack()
    pwb = E.PackedWeight(w, cin, cout, ks, transposed=False, flip=True, stride=1, layout=L.WLAYOUT_CHUNK32)
    pwb.pack()
    kw = dict(act=L.ACT_LEAKY02 if bn else L.ACT_RELU)
    if bn:
        kw.update(mean=torch.zeros(cin, device=dev), var=torch.ones(cin, device=dev), gamma=torch.ones(cin, device=dev), beta=torch.zeros(cin, device=dev))
    pro = E.make_prologue(**kw)
    ws = torch.empty(1 << 24, device=dev)
    descf = E.conv_desc(ks, 1, pad, cout=cout, w_layout=pwf.layout)
    descb = E.conv_desc(ks, 1, ks - 1 - pad, cout=cin, w_layout=L.WLAYOUT_CHUNK32)
    xv, yv, gv, dv = E.View(x, 0, cin), E.View(y, 0, (cout + 3) // 4 * 4), E.View(G, 0, cin), E.View(dy, 0, cout)
    tf = timeit(lambda: E.conv2d(xv.fd, pwf, None, pro, yv.fd, descf, None))
    tb = timeit(lambda: E.conv_bwd_data(dv.fd, pwb, xv.fd, pro, gv.fd, descb, ws, accumulate=2))
    fl = 2.0 * N * ho * ho * cin * cout * ks * ks
    print(json.dumps({"shape": name, "fwd_us": round(tf, 1), "fwd_TF": round(fl / tf / 1e6, 0), "bwd_data_us": round(tb, 1), "bwd_TF": round(fl / tb / 1e6, 0),
                      "epilogue_cost_us": round(tb - tf, 1)}), flush=True)
