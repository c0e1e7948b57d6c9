"""One FD-GAN training step on the HIP path.

The reference ships no training script: what its tree pins are the Adam hyper-parameters
(lrD = lrG = 2e-4, beta1 = 0.5, demo.py:43-46), weights_init N(0, 0.02) (misc.py:16-22), ImagePool(50)
(misc.py:140-161), the linear LR decay helper (misc.py:164-172), the component losses (SSIM, VGG16 features,
Blur / Laplacian) and the figure showing D on [img, LF(img), HF(img)].  This step is therefore OUR composition of
those pieces -- the loss weights are not recoverable from the reference:

    D step:  BCE(D(F(gt)), 1) + BCE(D(F(pool(fake.detach()))), 0)                          -> Adam(D)
    G step:  L1(fake, gt) + (1 - SSIM(fake, gt)) + sum_k MSE(VGG_k(fake), VGG_k(gt)) + w_adv * BCE(D(F(fake)), 1)  -> Adam(G)
    F(img) = cat[img, Blur15(img), Laplacian3(img)]

Every network forward / backward, the frequency split, SSIM and the scalar losses (L1, BCE, the perceptual MSE on
Vgg16's NHWC fp16 feature maps: fdgan_hip/losses.py) run through libfdgan_hip.so; Adam is one HIP kernel over a flat
fp32 parameter buffer (fdgan_hip/optim.py), whose flat gradient is also what RCCL all-reduces.  Nothing in a step
synchronises with the host until its six loss values are read, once, at the end.
Activations live in the modules' plan buffers, so each module's backward runs before its next forward (two
backward calls for the two halves of the D loss).
"""
import argparse
import os
import time

import torch

import misc
import models.dehaze1113 as net
import models.pytorch_ssim as pytorch_ssim
from fdgan_hip.dp import DpContext
from fdgan_hip.losses import bce_loss, l1_loss, vgg_perceptual, vgg_targets
from fdgan_hip.optim import FlatAdam
from loss import fusion_input
from myutils.vgg16 import Vgg16


class TrainStep:
    def __init__(self, device, lrG=2e-4, lrD=2e-4, beta1=0.5, w_adv=0.01, w_perc=1.0, w_ssim=1.0, w_l1=1.0, pool_size=50,
                 dp=None, vgg_weights=None, densenet_weights=None, synthetic=False):
        """vgg_weights: vgg16.weight / torchvision VGG16 state_dict / vgg16.t7 (myutils.utils.load_vgg16_weights);
        densenet_weights: torchvision densenet121 state_dict for the generator's encoder.  The reference trains on an
        ImageNet-pretrained encoder (dehaze1113.py:707) and a pretrained, frozen VGG16 (myutils/utils.py:84-94): without the
        files both are RANDOM here, which only `synthetic=True` (benchmarks, tests) accepts silently."""
        import warnings
        device = torch.device(device)
        if device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("TrainStep: the FD-GAN HIP path needs an MI355X (`cuda`) device; there is no CPU fallback (got %s)" % device)
        self.dev = device
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                 # the pretrained=True warning is replaced by the explicit check below
            self.netG = net.FDGAN()
        self.netG.apply(misc.weights_init)
        if densenet_weights:
            from models.tv_densenet121 import load_densenet121_weights
            load_densenet121_weights(self.netG, densenet_weights)
        self.netG = self.netG.to(device)
        self.netD = net.D(9, 36).to(device)
        self.netD.apply(misc.weights_init)
        self.vgg = Vgg16()                                  # the reference loads pretrained VGG16 weights; frozen
        if vgg_weights:
            from myutils.utils import load_vgg16_weights
            load_vgg16_weights(self.vgg, vgg_weights)
        self.vgg = self.vgg.to(device)
        if not synthetic and not (vgg_weights and densenet_weights):
            warnings.warn("TrainStep: %s randomly initialised (no weight file given) -- the reference uses ImageNet-pretrained "
                          "networks there; pass vgg_weights= / densenet_weights= (train.py --vgg / --densenet)"
                          % " and ".join(n for n, w in (("the VGG16 perceptual network is", vgg_weights),
                                                         ("the DenseNet-121 encoder is", densenet_weights)) if not w))
        for p in self.vgg.parameters():
            p.requires_grad_(False)
        # Only parameters that can receive a gradient go into the optimizer: FDGAN registers 2.2 M that never do
        # (conv0, dense_block31, dense_norm31, the dy blocks' bn1 / bn2 -- SURVEY 8e).  One dry forward + backward
        # at a tiny size finds them.
        self._init_trace("networks on the device")
        self.g_params = self._params_with_grad(self.netG, device)
        self._init_trace("probe forward + backward done, plans released")
        self.optG = FlatAdam(self.g_params, lr=lrG, betas=(beta1, 0.999))
        self.optD = FlatAdam(list(self.netD.parameters()), lr=lrD, betas=(beta1, 0.999))
        self.pool = misc.ImagePool(pool_size)
        self.w = dict(adv=w_adv, perc=w_perc, ssim=w_ssim, l1=w_l1)
        self.dp = dp
        # second HIP stream: work that does not depend on the generator's output (the discriminator's real half, VGG16 on
        # the ground truth) is enqueued there and fills the CUs the generator's small-grid launches leave idle
        # (FDGAN_NO_SIDE_STREAM: profiling aid -- everything on one stream, so that a kernel trace's durations are not inflated by
        # two networks sharing the CUs; with FDGAN_NO_WGRAD_STREAM it gives the strictly serial step of profiles/*_serial_*.)
        self.side = torch.cuda.current_stream(device) if os.environ.get("FDGAN_NO_SIDE_STREAM") else torch.cuda.Stream(device=device)
        # D's fake half on the side stream too (measured round 5, same box: 25.70 -> 25.59, 25.87 -> 25.79 ms; FDGAN_D_FAKE_MAIN=1: the
        # round-4 placement).  Also measured and NOT adopted: the whole step on a high-priority stream so that the side streams only
        # fill idle CUs (+0.55 ms), the weight-gradient stream at high priority (46 ms: it starves the chain).
        self.d_fake_side = os.environ.get("FDGAN_D_FAKE_MAIN") is None
        self._init_trace("optimizers built")
        if dp is not None and dp.world > 1:
            self.sync_replicas()

    @staticmethod
    def _init_trace(what):
        """FDGAN_DEBUG_INIT_TRACE=1 (tools/ranks8_loop.sh): drain the device and say on stderr how far this rank's start-up got -- the
        last line of a rank that dies names the phase whose GPU work raised the abort (profiles/r6_oversubscription.txt)."""
        if os.environ.get("FDGAN_DEBUG_INIT_TRACE"):
            import sys
            import time
            torch.cuda.synchronize()
            sys.stderr.write("[init-trace rank %s %.3f] %s\n" % (os.environ.get("RANK", "0"), time.time() % 1000.0, what))
            sys.stderr.flush()

    def sync_replicas(self):
        """Data-parallel replicas must START identical: only gradients are exchanged afterwards.  Rank 0's parameters
        (the optimizers' flat buffers) and BatchNorm buffers are broadcast, then every rank checks a checksum."""
        import torch.distributed as dist
        bufs = [self.optG.flat, self.optD.flat]
        bufs += [b for m in (self.netG, self.netD) for b in m.buffers() if b.dtype.is_floating_point]
        bufs += [p.data for p in self.netG.parameters() if not any(p is q for q in self.optG.params)]   # never-trained tensors
        bufs += [p.data for p in self.vgg.parameters()]                                                  # frozen, but must agree
        self._init_trace("sync_replicas: %d tensors to broadcast" % len(bufs))
        for b in bufs:
            dist.broadcast(b, src=0)
        self._init_trace("broadcasts done")
        # the checksums are formed on the HOST (60 MB once): with eight processes sharing one GPU (the tests' dry run) PyTorch's fp64
        # convert / reduce kernels, first used right here on a drained device, lost a rank to an illegal-instruction queue abort in
        # 3 of ~90 launches (profiles/r6_oversubscription.txt: FDGAN_DEBUG_INIT_TRACE pins the window)
        chk = torch.stack([b.detach().cpu().double().sum() for b in bufs[:2]]).to(bufs[0].device)
        lo, hi = chk.clone(), chk.clone()
        self._init_trace("checksums computed")
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        self._init_trace("checksums reduced")
        if not torch.equal(lo, hi):
            raise RuntimeError("data-parallel replicas differ after the initial broadcast")

    @staticmethod
    def _params_with_grad(module, device):
        was = {n: (m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone())
               for n, m in module.named_modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)}
        module(torch.rand(1, 3, 32, 32, device=device)).mean().backward()
        got = [p for p in module.parameters() if p.grad is not None]
        for p in module.parameters():
            p.grad = None
        for n, m in module.named_modules():                          # the probe must not count as a training step
            if n in was:
                m.running_mean.copy_(was[n][0]), m.running_var.copy_(was[n][1]), m.num_batches_tracked.copy_(was[n][2])
        module.release_plans()                                       # the 1x3x32x32 probe plan and its buffers
        return got

    @staticmethod
    def _cross(stream, *tensors):
        """Tensors allocated on one stream and used on `stream`: explicit waits order the WORK; record_stream makes torch's caching
        allocator, which hands a freed block back to its allocating stream at once, also wait for `stream`'s use before the memory
        is handed out again (VERDICT r5 #2d)."""
        for t in tensors:
            t = getattr(t, "buf", t)                    # engine.View (VGG16's target features live in its plan's buffers)
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)

    def _set_d_grad(self, flag):
        for p in self.netD.parameters():
            p.requires_grad_(flag)

    LOSS_NAMES = ("lossD", "lossG", "l1", "ssim", "perc", "adv")

    def step(self, haze, gt, sync=True):
        """haze, gt: (B,3,H,W) float in [0,1] on the device.  sync=True: a dict of python floats (ONE host sync, at the
        end).  sync=False: the six loss values as ONE device tensor in LOSS_NAMES order and no host synchronisation at
        all -- a training loop reads it every k steps (`losses_dict`): under data parallelism a per-step .tolist() is a
        bubble in every rank's launch queue."""
        # ---- D step: two backward calls (D's activations live in its plan buffers).  The real half and VGG16's target
        # features depend on `gt` only: they go to the side stream and run beside the generator's forward.
        self._set_d_grad(True)
        self.optD.zero_grad()
        main = torch.cuda.current_stream(self.dev)
        two = self.side is not main and self.side != main
        self.side.wait_stream(main)
        if two:
            gt.record_stream(self.side)                 # the caller's batch is read over there (allocator lifetime, see _cross)
        with torch.cuda.stream(self.side):
            with torch.no_grad():
                real_in = fusion_input(gt)
            l_real = bce_loss(self.netD(real_in), 1.0)
            l_real.backward()
            feats_gt = vgg_targets(self.vgg, gt)
        fake = self.netG(haze)                                                     # autograd graph of the generator
        if two:
            self._cross(main, l_real, *getattr(feats_gt, "taps", ()))   # produced on the side stream (VGG16's target-slot plan), consumed on the main one
            self._cross(self.side, fake)                # and the other way round
        if self.d_fake_side:
            # D's fake half, its Adam and the adversarial branch as ONE chain on the side stream, the perceptual / SSIM / L1 branch
            # meanwhile on the main one: the main stream only needs VGG16's target features from the side stream (an event), not D
            ev_gt = self.side.record_event()
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                with torch.no_grad():
                    fake_in = fusion_input(self.pool.query(fake.detach()))
                l_fake = bce_loss(self.netD(fake_in), 0.0)
                l_fake.backward()
                self.optD.allreduce_end(self.optD.allreduce_begin(self.dp))
                self.optD.step()
                self._set_d_grad(False)
                l_adv = bce_loss(self.netD(fusion_input(fake)), 1.0)
            main.wait_event(ev_gt)
        else:
            main.wait_stream(self.side)
            with torch.no_grad():
                fake_in = fusion_input(self.pool.query(fake.detach()))
            l_fake = bce_loss(self.netD(fake_in), 0.0)
            l_fake.backward()
            # ---- D's gradient exchange + Adam(D) + the G step's adversarial branch (frequency split -> D -> BCE) on the side
            # stream; the perceptual / SSIM / L1 branch, which does not need D, meanwhile on the main one.  So D's all-reduce
            # (3.2 MB: latency-priced on xGMI) is hidden under VGG16's forward instead of waited for.  The two branches only
            # meet at `fake`; autograd replays a node on its forward's stream, so the adversarial BACKWARD runs beside VGG16's too.
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self.optD.allreduce_end(self.optD.allreduce_begin(self.dp))
                self.optD.step()
                self._set_d_grad(False)                                                # D is a fixed critic here: no dW work
                l_adv = bce_loss(self.netD(fusion_input(fake)), 1.0)
        if two:
            self._cross(main, l_fake, l_adv)
        self.optG.zero_grad()
        l_perc = vgg_perceptual(self.vgg, fake, feats_gt)
        ssim = pytorch_ssim.ssim(fake, gt)
        l_l1 = l1_loss(fake, gt)
        main.wait_stream(self.side)
        lossG = self.w["l1"] * l_l1 + self.w["ssim"] * (1.0 - ssim) + self.w["perc"] * l_perc + self.w["adv"] * l_adv
        with self.optG.overlap(self.dp):            # slices of the flat gradient are all-reduced while the backward still runs
            lossG.backward()
        self.optG.step()
        vals = torch.stack([t.detach().float() for t in (l_real + l_fake, lossG, l_l1, ssim, l_perc, l_adv)])
        return self.losses_dict(vals) if sync else vals

    @classmethod
    def losses_dict(cls, vals):
        """The device tensor step(sync=False) returned -> dict of python floats (this is the host synchronisation)."""
        return dict(zip(cls.LOSS_NAMES, vals.tolist()))

    # ---- checkpoints in the format demo.py loads (/root/reference/demo.py:78-86: an nn.DataParallel state_dict) ----------
    def save_checkpoint(self, out_dir, epoch):
        """netG_epoch_<e>.pth / netD_epoch_<e>.pth with `module.`-prefixed keys, as torch.save(netG.state_dict()) of the
        reference's nn.DataParallel-wrapped networks would write them."""
        import os
        os.makedirs(out_dir, exist_ok=True)
        paths = []
        for name, netw in (("netG", self.netG), ("netD", self.netD)):
            sd = {"module." + k: v.detach().cpu().clone() for k, v in netw.state_dict().items()}
            path = os.path.join(out_dir, "%s_epoch_%d.pth" % (name, epoch))
            torch.save(sd, path)
            paths.append(path)
        return paths


def build_parser():
    ap = argparse.ArgumentParser(description="FD-GAN training on the HIP path (flags as /root/reference/demo.py:28-60 names them)")
    ap.add_argument("--dataset", default="pix2pix")
    ap.add_argument("--dataroot", default="", help="directory of <i>.h5 pairs (datasets/pix2pix.py); empty: synthetic images")
    ap.add_argument("--batchSize", type=int, default=16)
    ap.add_argument("--originalSize", type=int, default=256)
    ap.add_argument("--imageSize", type=int, default=256)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--niter", type=int, default=5, help="epochs over --dataroot (synthetic: steps)")
    ap.add_argument("--lrG", type=float, default=0.0002)
    ap.add_argument("--lrD", type=float, default=0.0002)
    ap.add_argument("--beta1", type=float, default=0.5)
    ap.add_argument("--annealStart", type=int, default=0, help="first epoch of the linear LR decay (misc.adjust_learning_rate)")
    ap.add_argument("--annealEvery", type=int, default=400, help="epochs over which the LR decays to zero")
    ap.add_argument("--poolSize", type=int, default=50)
    ap.add_argument("--exp", "--save", dest="exp", default="", help="directory for netG_epoch_<e>.pth / netD_epoch_<e>.pth")
    ap.add_argument("--evalIter", type=int, default=1, help="save a checkpoint every this many epochs")
    ap.add_argument("--display", type=int, default=5, help="read the losses back (one host sync) every this many steps")
    ap.add_argument("--vgg", default="", help="pretrained VGG16: vgg16.weight, a torchvision state_dict or vgg16.t7")
    ap.add_argument("--densenet", default="", help="torchvision densenet121 state_dict for the generator's encoder")
    ap.add_argument("--fixedBatch", action="store_true", help="synthetic runs: ONE batch of smooth images reused every step (an overfit / learning check)")
    return ap


def run(opt):
    """Trains and returns (checkpoint paths written, last loss dict).  With --dataroot: epochs over the .h5 pairs through
    misc.getLoader (N ranks: fdgan_hip.dp.RankBatches, equal step counts, ragged tail dropped); without: --niter steps on synthetic images."""
    dp = DpContext.from_env()
    dev = dp.device or torch.device("cuda", 0)
    synthetic = not opt.dataroot
    if synthetic and dp.rank == 0 and not (opt.vgg and opt.densenet):
        print("train.py: no --dataroot / weight files: a SYNTHETIC run on random images with randomly initialised "
              "VGG16 / DenseNet-121 encoder (throughput and plumbing only)", flush=True)
    ts = TrainStep(dev, opt.lrG, opt.lrD, opt.beta1, pool_size=opt.poolSize, dp=dp, vgg_weights=opt.vgg or None,
                   densenet_weights=opt.densenet or None, synthetic=synthetic)
    written, last, it = [], None, 0
    t0 = time.time()

    def one(haze, gt):
        nonlocal last, it, t0
        vals = ts.step(haze, gt, sync=False)
        it += 1
        if it % max(1, opt.display) == 0:
            last = ts.losses_dict(vals)                  # the only host synchronisation of the loop
            if dp.rank == 0:
                dt = (time.time() - t0) / max(1, opt.display)
                print("[%d] %.1f ms/step  %s" % (it, 1e3 * dt, {k: round(v, 4) for k, v in last.items()}), flush=True)
            t0 = time.time()
        return vals

    if synthetic:
        g = torch.Generator(device="cpu").manual_seed(1234 + dp.rank)
        vals = None
        fixed = None
        if getattr(opt, "fixedBatch", False):      # smooth images (8 x 8 noise, bilinear): something a generator can fit, unlike white noise
            lo = torch.rand(opt.batchSize, 3, 8, 8, generator=g)
            fixed = torch.nn.functional.interpolate(lo, size=(opt.imageSize, opt.imageSize), mode="bilinear", align_corners=False).to(dev)
        for _ in range(opt.niter):
            gt = fixed if fixed is not None else torch.rand(opt.batchSize, 3, opt.imageSize, opt.imageSize, generator=g).to(dev)
            vals = one((gt * 0.6 + 0.3).clamp(0, 1), gt)
        if vals is not None:
            last = ts.losses_dict(vals)
        if opt.exp and dp.rank == 0:
            written += ts.save_checkpoint(opt.exp, 0)
    else:
        loader = misc.getLoader(opt.dataset, opt.dataroot, opt.originalSize, opt.imageSize, opt.batchSize, opt.workers,
                                split="train", shuffle=True, seed=1234)
        sampler = None
        if dp.world > 1:
            # one process: the reference's loader as it is (shuffled, ragged last batch).  N ranks: equal step counts on every
            # rank or the all-reduces of the odd step out have no peer (fdgan_hip.dp.RankBatches)
            from fdgan_hip.dp import RankBatches
            sampler = RankBatches(len(loader.dataset), opt.batchSize, dp.world, dp.rank, seed=1234)
            loader = torch.utils.data.DataLoader(loader.dataset, batch_sampler=sampler, num_workers=int(opt.workers))
        for epoch in range(opt.niter):
            if epoch > opt.annealStart:                  # linear decay, as misc.adjust_learning_rate is written to be used
                misc.adjust_learning_rate(ts.optG, opt.lrG, epoch, None, opt.annealEvery)
                misc.adjust_learning_rate(ts.optD, opt.lrD, epoch, None, opt.annealEvery)
            vals = None
            if sampler is not None:
                sampler.set_epoch(epoch)
            for haze, gt in loader:
                vals = one(haze.float().to(dev, non_blocking=True), gt.float().to(dev, non_blocking=True))
            if vals is not None:
                last = ts.losses_dict(vals)
            if opt.exp and dp.rank == 0 and (epoch + 1) % max(1, opt.evalIter) == 0:
                written += ts.save_checkpoint(opt.exp, epoch)
    torch.cuda.synchronize()
    dp.close()
    return written, last


def main(argv=None):
    written, last = run(build_parser().parse_args(argv))
    if last is not None:
        print("final losses:", {k: round(v, 4) for k, v in last.items()}, flush=True)
    for pth in written:
        print("wrote", pth, flush=True)


if __name__ == "__main__":
    main()
