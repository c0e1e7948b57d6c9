"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU fp32 restatement of the FD-GAN conv hot path (generator,
Fusion-discriminator, VGG16 features, Gaussian/Laplacian frequency split, SSIM
loss) used as the *checker* for the HIP path.  Nothing under fd-gan_amd/ may
import this package; only tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py do.

Pinning status (see DESIGN.md "Oracle"):
  * FDGAN wiring, D, BottleneckBlockdy/TransitionBlockdy, Vgg16, pytorch_ssim:
    pinned -- validated in the build container against the reference's own
    modules imported from /root/reference behind import shims
    (oracle/make_golden.py), outputs committed under tests/golden/.
  * DenseNet-121 topology (third-party torchvision, absent from the reference
    tree and from this image): parity unpinned by the reference; pinned here by
    torchvision's published key names / parameter counts.
  * Blur / Laplacian: source absent from the reference (only CPython-3.6
    bytecode, not loadable); restated from the disassembly in SURVEY Appendix B
    and pinned by its known-answer values (kernel sum/centre/corner, constant
    image responses).
"""
