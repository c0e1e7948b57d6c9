"""TEST INFRASTRUCTURE (oracle).  fp32 PyTorch-CPU restatement of the two pix2pix U-Nets among the reference's DCPDN-era
networks: /root/reference/models/dehaze22.py `G` (:205-362) and `G2` (:364-488), built from `blockUNet` (:51-65).

Written functionally so that every quirk is explicit instead of inherited from `inplace=True`:
  * the encoder's LeakyReLU(0.2, inplace=True) (:54-56) runs on the tensor the skip connection later concatenates, so the
    decoder sees leaky_relu(out_k), not out_k (:320-333);
  * the decoder's ReLU(inplace=True) then runs on the concatenation;
  * Dropout2d(0.5) sits after BatchNorm in dlayer7 / dlayer6 and after the bare transposed conv in dlayer8 (:262-273);
    `masks` = three (N, C) tensors of {0, 2} in the order dlayer8, 7, 6 -- None draws them exactly as F.dropout2d does,
    which reproduces the reference bit for bit under the same torch seed;
  * G's head (:336-356): avg_pool2d 16 / 8 / 4 / 2 -> Conv2d(20, 1, 1) -> LeakyReLU(0.2) -> nearest upsampling, concatenated
    IN FRONT of the 20 decoder channels, Conv2d(24, out, 3, 1, 1, bias=False), tanh;  G2's tail (:384-386):
    ConvTranspose2d(2 nf, out, 4, 2, 1) -> LeakyReLU(0.2) (the module is merely called `tanh`).
Pinned against the real reference by oracle/make_golden.py (max |oracle - reference| recorded in MANIFEST.json)."""
import torch
import torch.nn.functional as F


def _bn(x, sd, prefix, training, momentum=0.1, eps=1e-5):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training:
        y = F.batch_norm(x, rm, rv, w, b, True, momentum, eps)        # updates rm / rv in place, as the module does
        sd[prefix + ".num_batches_tracked"] += 1
        return y
    return F.batch_norm(x, rm, rv, w, b, False, momentum, eps)


def unet_forward(sd, x, training, kind, masks=None):
    """sd: state_dict (tensors are updated in place in training mode: running statistics).  kind: "G" or "G2".
    Returns (output, masks used)."""
    sd = dict(sd)
    used = []
    outs = [None] * 9
    t = F.conv2d(x, sd["layer1.layer1.weight"], None, 2, 1)
    outs[1] = t
    for k in range(2, 9):
        a = F.leaky_relu(outs[k - 1], 0.2)
        outs[k - 1] = a                                               # in place in the reference: the skip tensor changes too
        t = F.conv2d(a, sd["layer%d.layer%d.conv.weight" % (k, k)], None, 2, 1)
        outs[k] = _bn(t, sd, "layer%d.layer%d.bn" % (k, k), training)
    d = outs[8]
    for k in range(8, 1, -1):
        d = F.relu(d)
        d = F.conv_transpose2d(d, sd["dlayer%d.dlayer%d.tconv.weight" % (k, k)], None, 2, 1)
        if k <= 7:
            d = _bn(d, sd, "dlayer%d.dlayer%d.bn" % (k, k), training)
        if k >= 6 and training:
            if masks is None:
                m = F.dropout2d(torch.ones(d.shape[0], d.shape[1], 1, 1), 0.5, True)
            else:
                m = masks[8 - k].view(d.shape[0], d.shape[1], 1, 1)
            used.append(m[:, :, 0, 0].clone())
            d = d * m
        d = torch.cat([d, outs[k - 1]], 1)
    d = F.relu(d)
    d = F.conv_transpose2d(d, sd["dlayer1.dlayer1.tconv.weight"], None, 2, 1)
    if kind == "G2":
        return F.leaky_relu(d, 0.2), used
    size = d.shape[2:4]
    pyr = []
    for k, nm in ((16, "conv1010"), (8, "conv1020"), (4, "conv1030"), (2, "conv1040")):
        p = F.conv2d(F.avg_pool2d(d, k), sd[nm + ".weight"], sd[nm + ".bias"])
        pyr.append(F.interpolate(F.leaky_relu(p, 0.2), size=size, mode="nearest"))
    d = torch.cat(pyr + [d], 1)
    return torch.tanh(F.conv2d(d, sd["dlayerfinal.dlayer1.conv.weight"], None, 1, 1)), used


# ------------------------------------------------------------------------------------------------------------------
# `Dense`: the DCPDN dehazing network, three spellings of one topology --
#   /root/reference/models/dehaze1113.py Dense  :431-570   tail: conv_refin -> batchnorm20 -> LeakyReLU -> refine3(20 -> 3) -> tanh
#   /root/reference/models/dehaze1113.py Dense2 :572-699 } tail: conv_refin -> LeakyReLU -> four-scale pooling head (avg_pool 32 / 16 /
#   /root/reference/models/dehaze22.py   Dense  :531-660 }       8 / 4, Conv2d(20, 1, 1), LeakyReLU, nearest upsampling) in FRONT of the
#                                                                20 channels -> refine3(24 -> 3) -> tanh
# Encoder: torchvision DenseNet-121 stem (conv0 7x7 stride 2, norm0, relu0, MaxPool2d(3, 2, 1)) + dense blocks 1-3 with their
# transitions; decoder: BottleneckBlock (BN-ReLU-1x1, BN-ReLU-3x3, concat) / TransitionBlock (BN-ReLU-ConvTranspose 1x1, nearest x2)
# with the skip concatenations x42 = [x4, x2], x52 = [x5, x1], x8 = [x8, x] (dehaze22.py:491-529, :604-632).
# ------------------------------------------------------------------------------------------------------------------
def dense_forward(sd, x, training, tail, taps=None):
    """sd: state_dict (running statistics updated in place in training mode).  tail: "bn" (dehaze1113.Dense) or "pyramid".
    taps: optional dict that receives conv_refin's raw output (`x9pre`, gradient retained) for gradient debugging."""
    sd = dict(sd)
    bn = lambda t, p: _bn(t, sd, p, training)

    def dense_block(t, name, layers):
        for i in range(1, layers + 1):
            p = "%s.denselayer%d." % (name, i)
            y = F.conv2d(F.relu(bn(t, p + "norm1")), sd[p + "conv1.weight"])
            y = F.conv2d(F.relu(bn(y, p + "norm2")), sd[p + "conv2.weight"], None, 1, 1)
            t = torch.cat([t, y], 1)
        return t

    def transition(t, name):
        return F.avg_pool2d(F.conv2d(F.relu(bn(t, name + ".norm")), sd[name + ".conv.weight"]), 2)

    def bottleneck(t, name):
        y = F.conv2d(F.relu(bn(t, name + ".bn1")), sd[name + ".conv1.weight"])
        y = F.conv2d(F.relu(bn(y, name + ".bn2")), sd[name + ".conv2.weight"], None, 1, 1)
        return torch.cat([t, y], 1)

    def transup(t, name):
        y = F.conv_transpose2d(F.relu(bn(t, name + ".bn1")), sd[name + ".conv1.weight"])
        return F.interpolate(y, scale_factor=2, mode="nearest")

    x0 = F.max_pool2d(F.relu(bn(F.conv2d(x, sd["conv0.weight"], None, 2, 3), "norm0")), 3, 2, 1)
    x1 = transition(dense_block(x0, "dense_block1", 6), "trans_block1")
    x2 = transition(dense_block(x1, "dense_block2", 12), "trans_block2")
    x3 = transition(dense_block(x2, "dense_block3", 24), "trans_block3")
    x4 = transup(bottleneck(x3, "dense_block4"), "trans_block4")
    x5 = transup(bottleneck(torch.cat([x4, x2], 1), "dense_block5"), "trans_block5")
    x6 = transup(bottleneck(torch.cat([x5, x1], 1), "dense_block6"), "trans_block6")
    x7 = transup(bottleneck(x6, "dense_block7"), "trans_block7")
    x8 = transup(bottleneck(x7, "dense_block8"), "trans_block8")
    x9 = F.conv2d(torch.cat([x8, x], 1), sd["conv_refin.weight"], sd["conv_refin.bias"], 1, 1)
    if taps is not None:
        if x9.requires_grad:
            x9.retain_grad()
        taps["x9pre"] = x9
    if tail == "bn":
        x9 = F.leaky_relu(bn(x9, "batchnorm20"), 0.2)
        return torch.tanh(F.conv2d(x9, sd["refine3.weight"], sd["refine3.bias"], 1, 1))
    x9 = F.leaky_relu(x9, 0.2)
    size = x9.shape[2:4]
    pyr = []
    for k, nm in ((32, "conv1010"), (16, "conv1020"), (8, "conv1030"), (4, "conv1040")):
        p = F.conv2d(F.avg_pool2d(x9, k), sd[nm + ".weight"], sd[nm + ".bias"])
        pyr.append(F.interpolate(F.leaky_relu(p, 0.2), size=size, mode="nearest"))
    return torch.tanh(F.conv2d(torch.cat(pyr + [x9], 1), sd["refine3.weight"], sd["refine3.bias"], 1, 1))


def dehaze_forward(sd, x, training, masks=None):
    """/root/reference/models/dehaze22.py `dehaze` :662-753: transmission = Dense(x), airlight = G2(x) pooled over H x H windows,
    J = (x - A) / (|t| + 1e-10) + A (:699-715), then refine1 / refine2 (LeakyReLU), the four-scale head (32 / 16 / 8 / 4) and
    tanh(refine3).  `tran_est` (a G) is registered but never called (:665).  Returns (dehaze, tran, atp, dehaze2, masks used)."""
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    tran = dense_forward(sub("tran_dense."), x, training, "pyramid")
    atp, used = unet_forward(sub("atp_est."), x, training, "G2", masks)
    zz = torch.abs(tran) + (10 ** -10)
    size = atp.shape[2:4]
    atp = F.interpolate(F.leaky_relu(F.avg_pool2d(atp, atp.shape[2]), 0.2), size=size, mode="nearest")
    dehaze2 = (x - atp) / zz + atp
    d = torch.cat([dehaze2, x], 1)
    d = F.leaky_relu(F.conv2d(d, sd["refine1.weight"], sd["refine1.bias"], 1, 1), 0.2)
    d = F.leaky_relu(F.conv2d(d, sd["refine2.weight"], sd["refine2.bias"], 1, 1), 0.2)
    pyr = []
    for k, nm in ((32, "conv1010"), (16, "conv1020"), (8, "conv1030"), (4, "conv1040")):
        p = F.conv2d(F.avg_pool2d(d, k), sd[nm + ".weight"], sd[nm + ".bias"])
        pyr.append(F.interpolate(F.leaky_relu(p, 0.2), size=size, mode="nearest"))
    out = torch.tanh(F.conv2d(torch.cat(pyr + [d], 1), sd["refine3.weight"], sd["refine3.bias"], 1, 1))
    return out, tran, atp, dehaze2, used
