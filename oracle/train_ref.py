"""TEST INFRASTRUCTURE (oracle).  CPU fp32 restatement of the training step fd-gan_amd/train.py runs on the HIP path,
composed from the other oracle modules (each a restatement of a reference file, cited there):

    D step:  BCE(D(F(gt)), 1) + BCE(D(F(fake.detach())), 0)                                   -> Adam(D)
    G step:  L1(fake, gt) + (1 - SSIM(fake, gt)) + sum_k MSE(VGG_k(fake), VGG_k(gt)) + w_adv BCE(D(F(fake)), 1)  -> Adam(G)
    F(img) = cat[img, Blur15(img), Laplacian3(img)]

The reference ships no training loop; what it pins are the pieces: Adam lr 2e-4 / beta1 0.5
(/root/reference/demo.py:43-46), the component losses (models/pytorch_ssim, myutils/vgg16.py, the Blur / Laplacian
filters) and the Fusion-discriminator input (facades/network.png).  Only tests/ and bench.py's cpu_baseline leg
import this; the image pool is left out (it returns its input unchanged until 50 images have passed).  parity
unpinned against the reference for the COMPOSITION (there is nothing to pin it to); every component is pinned by its
own golden vectors."""
import torch
import torch.nn.functional as F

from . import dehaze1113_ref as ref
from . import freqsplit_ref, ssim_ref, vgg16_ref


def fusion_input(img):
    return torch.cat([img, freqsplit_ref.blur(img), freqsplit_ref.laplacian(img)], 1)


class TrainStepRef:
    def __init__(self, netG_state=None, netD_state=None, vgg_state=None, lrG=2e-4, lrD=2e-4, beta1=0.5, w_adv=0.01, w_perc=1.0,
                 w_ssim=1.0, w_l1=1.0):
        self.netG, self.netD, self.vgg = ref.FDGAN(), ref.D(9, 36), vgg16_ref.Vgg16()
        if netG_state is not None:
            self.netG.load_state_dict(netG_state)
        if netD_state is not None:
            self.netD.load_state_dict(netD_state)
        if vgg_state is not None:
            self.vgg.load_state_dict(vgg_state)
        for p in self.vgg.parameters():
            p.requires_grad_(False)
        self.optG = torch.optim.Adam([p for p in self.netG.parameters()], lr=lrG, betas=(beta1, 0.999))
        self.optD = torch.optim.Adam(self.netD.parameters(), lr=lrD, betas=(beta1, 0.999))
        self.w = dict(adv=w_adv, perc=w_perc, ssim=w_ssim, l1=w_l1)

    def _set_d_grad(self, flag):
        for p in self.netD.parameters():
            p.requires_grad_(flag)

    def step(self, haze, gt):
        out = {}
        fake = self.netG(haze.clone())                 # the reference's in-place ReLUs modify their input
        self._set_d_grad(True)
        self.optD.zero_grad()
        with torch.no_grad():
            real_in, fake_in = fusion_input(gt), fusion_input(fake.detach())
        p_real = self.netD(real_in)
        l_real = F.binary_cross_entropy(p_real, torch.ones_like(p_real))
        l_real.backward()
        p_fake = self.netD(fake_in)
        l_fake = F.binary_cross_entropy(p_fake, torch.zeros_like(p_fake))
        l_fake.backward()
        self.optD.step()
        out["lossD"] = float((l_real + l_fake).detach())
        self._set_d_grad(False)
        self.optG.zero_grad()
        with torch.no_grad():
            feats_gt = self.vgg(gt)
        feats = self.vgg(fake)
        l_perc = sum(F.mse_loss(a, b) for a, b in zip(feats, feats_gt))
        l_ssim = 1.0 - ssim_ref.ssim(fake, gt)
        l_l1 = (fake - gt).abs().mean()
        p_adv = self.netD(fusion_input(fake))
        l_adv = F.binary_cross_entropy(p_adv, torch.ones_like(p_adv))
        lossG = self.w["l1"] * l_l1 + self.w["ssim"] * l_ssim + self.w["perc"] * l_perc + self.w["adv"] * l_adv
        lossG.backward()
        self.optG.step()
        out.update({k: float(v.detach()) for k, v in dict(lossG=lossG, l1=l_l1, ssim=1.0 - l_ssim, perc=l_perc, adv=l_adv).items()})
        return out
