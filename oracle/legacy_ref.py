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
