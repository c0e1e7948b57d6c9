"""GPU parity tests of the fused implicit-GEMM convolution (through the C ABI) against a
torch-CPU statement of the same math (tests/hiputil.py), on seeded inputs.

Tolerance: operands are rounded to fp16 exactly where the kernel rounds them, so the
only differences are fp32 accumulation order, rare 1-ulp flips of the activated
operand and the final fp16 rounding of the stored result (2^-11 relative)."""
import os

import pytest
import torch

from hiputil import (ACT_LEAKY02, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, f16_round, fused_conv_ref, rel_rms,
                     seeded)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from fdgan_hip import engine
    from fdgan_hip import lib
    lib.load()
    assert torch.cuda.is_available()
    arch = lib.load().fdgan_device_arch()
    assert arch is not None and arch.decode().startswith("gfx950"), arch
    return engine


def _run(E, n, h, w, cin, cout, k, stride=1, pad=0, pitch_in=None, c0_in=0, pitch_out=None, c0_out=0, bias=False,
         p_act=ACT_NONE, bn=False, pool=False, e_act=ACT_NONE, upsample=False, transposed=False, stats=False,
         nchw_out=False, seed=0, running=False, layout=None):
    from fdgan_hip import lib as L
    dev = torch.device("cuda:0")
    pitch_in = pitch_in or (cin + 7) // 8 * 8
    hin, win = (h // 2, w // 2) if pool else (h, w)
    ho, wo = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
    up = 2 if upsample else 1
    cstore = cout if nchw_out else (cout + 3) // 4 * 4
    pitch_out = pitch_out or (cstore + 7) // 8 * 8
    x = f16_round(seeded((n, cin, h, w), seed, -1.5, 1.5))
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    wt = seeded(wshape, seed + 1, -1.0, 1.0) * (2.0 / (cin * k * k)) ** 0.5
    b = seeded((cout,), seed + 2, -0.5, 0.5) if bias else None
    bnp = None
    if bn:
        bnp = dict(mean=seeded((cin,), seed + 3, -0.3, 0.3), var=seeded((cin,), seed + 4, 0.5, 1.5),
                   gamma=seeded((cin,), seed + 5, 0.5, 1.5), beta=seeded((cin,), seed + 6, -0.3, 0.3), eps=1e-5)
    y_ref, m_ref, v_ref = fused_conv_ref(x, wt, b, k, stride, pad, p_act, bnp, pool, e_act, upsample, transposed)

    xbuf = torch.full((n, h, w, pitch_in), 7.0, dtype=torch.float16, device=dev)   # poison outside the slice
    xbuf[..., c0_in:c0_in + cin] = x.permute(0, 2, 3, 1).to(dev).to(torch.float16)
    if (cin % 8) and c0_in + (cin + 7) // 8 * 8 <= pitch_in:
        xbuf[..., c0_in + cin:c0_in + (cin + 7) // 8 * 8] = 0     # padded channels must be finite
    xv = E.View(xbuf, c0_in, cin)
    wparam = wt.to(dev).contiguous()
    pw = E.PackedWeight(wparam, cout, cin, k, transposed, stride=stride, layout=layout)
    pw.pack()
    bd = b.to(dev) if bias else None
    pro, run = None, None
    if bn or p_act != ACT_NONE or pool:
        kw = dict(act=p_act, pool=pool)
        if bn:
            d = {kk: vv.to(dev) for kk, vv in bnp.items() if kk != "eps"}
            kw.update(mean=d["mean"], var=d["var"], gamma=d["gamma"], beta=d["beta"], eps=bnp["eps"])
            if running:
                run = dict(rm=torch.zeros(cin, device=dev), rv=torch.ones(cin, device=dev),
                           nbt=torch.zeros((), dtype=torch.int64, device=dev))
                kw.update(running_mean=run["rm"], running_var=run["rv"], nbt=run["nbt"], momentum=0.1,
                          count=n * h * w)
        pro = E.make_prologue(**kw)
    if nchw_out:
        ybuf = torch.full((n, cout, ho * up, wo * up), 9.0, dtype=torch.float32, device=dev)
        yfd = E.nchw_f32_view(ybuf)
    else:
        ybuf = torch.full((n, ho * up, wo * up, pitch_out), 9.0, dtype=torch.float16, device=dev)
        yv = E.View(ybuf, c0_out, cstore)
        yfd = yv.fd
    desc = E.conv_desc(k, stride, pad, e_act, upsample, cout=cout, w_layout=pw.layout)
    ws = torch.zeros(1 << 22, dtype=torch.float32, device=dev) if stats else None
    info = E.conv2d(xv.fd, pw, bd, pro, yfd, desc, ws)
    out = {}
    if stats:
        mean = torch.zeros(cout, device=dev)
        var = torch.zeros(cout, device=dev)
        E.bn_finalize(ws, info, cout, n * ho * wo, mean, var)
        out["mean"], out["var"] = mean.cpu(), var.cpu()
    torch.cuda.synchronize()
    if nchw_out:
        y = ybuf.cpu()
    else:
        yb = ybuf.float().cpu()
        y = yb[..., c0_out:c0_out + cout].permute(0, 3, 1, 2).contiguous()
        # untouched channels keep their poison; padded store channels are zero
        if c0_out > 0:
            assert (yb[..., :c0_out] == 9.0).all()
        if c0_out + cstore < pitch_out:
            assert (yb[..., c0_out + cstore:] == 9.0).all()
        if cstore > cout:
            assert (yb[..., c0_out + cout:c0_out + cstore] == 0).all()
    assert torch.isfinite(y).all()
    err = rel_rms(y, y_ref)
    scale = float(y_ref.abs().max())
    maxerr = float((y - y_ref).abs().max())
    tol_max = 0.004 * max(scale, 1e-3) + 1e-3
    assert err < (1.5e-3 if not nchw_out else 1e-3), "rel rms %.3g" % err
    assert maxerr < tol_max, "max err %.3g (scale %.3g)" % (maxerr, scale)
    if stats:
        assert (out["mean"] - m_ref).abs().max() < 2e-4 * max(1.0, float(m_ref.abs().max()))
        assert ((out["var"] - v_ref).abs() / (v_ref + 1e-6)).max() < 1e-3
    if run is not None:
        torch.cuda.synchronize()
        cnt = n * h * w
        rm = 0.9 * 0.0 + 0.1 * bnp["mean"]
        rv = 0.9 * 1.0 + 0.1 * bnp["var"] * cnt / (cnt - 1)
        assert (run["rm"].cpu() - rm).abs().max() < 1e-6
        assert (run["rv"].cpu() - rv).abs().max() < 1e-5
        assert int(run["nbt"].item()) == 1
    return err


# the headline shape: dense-layer growth conv, BN+ReLU prologue, slice output, statistics
def test_conv3x3_128_to_32_bn_relu_stats(E):
    _run(E, 2, 32, 48, 128, 32, 3, pad=1, bn=True, p_act=ACT_RELU, stats=True, pitch_out=256, c0_out=96, running=True)


def test_conv3x3_persistent_filter_kernel(E):
    """conv3x3_pw: ragged tiles, several tiles per workgroup (persistence + register statistics),
    partial last chunk, NHWC fallback stores."""
    _run(E, 3, 40, 24, 128, 32, 3, pad=1, bn=True, p_act=ACT_RELU, stats=True, pitch_out=96, c0_out=64, seed=40)
    _run(E, 16, 128, 128, 128, 32, 3, pad=1, bn=True, p_act=ACT_RELU, stats=True, pitch_out=64, c0_out=32, seed=41)
    _run(E, 2, 33, 17, 72, 20, 3, pad=1, p_act=ACT_LEAKY02, bias=True, stats=True, seed=42)
    _run(E, 1, 64, 64, 16, 3, 3, pad=1, bias=True, e_act=ACT_TANH, nchw_out=True, seed=43)


def test_conv3x3_row_streaming_kernel(E):
    """conv3x3_rs (Cin == 128, Cout <= 32): ragged strips and rows, row segments, more work items than
    workgroups, input as a channel slice of a wider buffer, narrow outputs, no-BN and LeakyReLU prologues."""
    from fdgan_hip import engine, lib as L
    info = engine.conv_info(engine.View(torch.empty((16, 64, 64, 128), dtype=torch.float16, device="cuda:0")).fd,
                            engine.View(torch.empty((16, 64, 64, 32), dtype=torch.float16, device="cuda:0")).fd, 32,
                            engine.conv_desc(3, 1, 1))
    assert (info.grid_x, info.stats_rows, info.stats_cpad) == (256, 256, 32)      # 4 strips x 4 segments x 16 images
    _run(E, 3, 40, 24, 128, 32, 3, pad=1, bn=True, p_act=ACT_RELU, stats=True, pitch_out=96, c0_out=64, seed=50)
    _run(E, 2, 21, 19 + 5, 128, 32, 3, pad=1, bn=True, p_act=ACT_RELU, stats=True, pitch_in=192, c0_in=64, seed=51)
    _run(E, 5, 256, 256, 128, 32, 3, pad=1, bn=True, p_act=ACT_RELU, stats=True, pitch_out=64, c0_out=32, seed=52)
    _run(E, 1, 8, 8, 128, 32, 3, pad=1, bn=True, p_act=ACT_RELU, stats=True, seed=53, running=True)
    _run(E, 2, 36, 40, 128, 20, 3, pad=1, p_act=ACT_LEAKY02, bias=True, stats=True, seed=54)
    _run(E, 2, 16, 32, 128, 3, 3, pad=1, bias=True, e_act=ACT_TANH, nchw_out=True, seed=55)
    _run(E, 1, 24, 16, 128, 32, 3, pad=1, stats=True, seed=56)


def test_conv3x3_row_streaming_second_generation(E):
    """conv3x3_rs2 (the fast path of the growth convs: 32 filters, no bias, widths of whole 16-pixel strips, whole row groups):
    every iteration count 1 .. 13 per work item -- the pipeline's first steps, its 4-step compute loop and 4-step helper loop
    and all their remainders, for both helper parities -- with the three prologue kinds, an image-edge strip in every item
    (16-pixel-wide images: both zero columns), a sliced output, and items with different lengths in one launch."""
    from fdgan_hip import engine
    for i, h in enumerate((4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52)):
        info = engine.conv_info(engine.View(torch.empty((256, h, 16, 128), dtype=torch.float16, device="cuda:0")).fd,
                                engine.View(torch.empty((256, h, 16, 32), dtype=torch.float16, device="cuda:0")).fd, 32,
                                engine.conv_desc(3, 1, 1))
        assert info.grid_x == 256 and info.lds_bytes > 140 * 1024      # one whole-height item per workgroup, the 12-wave kernel
        kind = i % 3
        _run(E, 256, h, 16, 128, 32, 3, pad=1, bn=kind != 2, p_act=(ACT_RELU, ACT_LEAKY02, ACT_NONE)[kind], stats=True,
             pitch_out=64 if i % 2 else None, c0_out=32 if i % 2 else 0, seed=60 + i)
    _run(E, 3, 52, 48, 128, 32, 3, pad=1, bn=True, p_act=ACT_RELU, stats=True, seed=80)        # segments of 8 rows and one of 4
    _run(E, 40, 20, 32, 128, 32, 3, pad=1, bn=True, p_act=ACT_LEAKY02, stats=True, pitch_in=192, c0_in=64, seed=81)


def test_last_workgroup_finalizes_the_statistics(E):
    """The in-kernel finalize (FdStats.mean: the last workgroup to arrive reduces every workgroup's partial row; relaxed
    agent-scope atomics instead of device-scope fences) of the three kernels that have it against the separate
    fdgan_bn_finalize launch on the same partial rows: 300 back-to-back launches each, every one must give the same mean /
    variance (a row read before its writer's store had landed, or a stale counter, shows up as a different sum)."""
    dev = torch.device("cuda:0")
    for k, cin, hw in ((3, 128, 128), (3, 128, 19), (1, 224, 128)):       # conv3x3_rs2, conv3x3_rs, conv1x1_ds
        cout = 32 if k == 3 else 128
        x = (torch.randn(16, hw, hw, cin, device=dev) * 0.7).to(torch.float16)
        y = torch.empty(16, hw, hw, cout, dtype=torch.float16, device=dev)
        pw = E.PackedWeight(torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5, cout, cin, k)
        pw.pack()
        pro = E.make_prologue(act=ACT_RELU, mean=torch.randn(cin, device=dev) * 0.1, var=torch.rand(cin, device=dev) + 0.5,
                              gamma=torch.rand(cin, device=dev) + 0.5, beta=torch.randn(cin, device=dev) * 0.1)
        desc = E.conv_desc(k, 1, k // 2, cout=cout, w_layout=pw.layout)
        xv, yv = E.View(x, 0, cin), E.View(y, 0, cout)
        ws = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
        count = 16 * hw * hw
        info = E.conv2d(xv.fd, pw, None, pro, yv.fd, desc, ws)
        assert info.fused_finalize == 1
        mean0, var0 = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
        E.bn_finalize(ws, info, cout, count, mean0, var0)
        counter = torch.zeros(1, dtype=torch.int32, device=dev)
        means, vars_ = torch.zeros(300, cout, device=dev), torch.zeros(300, cout, device=dev)
        for i in range(300):
            E.conv2d(xv.fd, pw, None, pro, yv.fd, desc, ws, (means[i].data_ptr(), vars_[i].data_ptr(), counter.data_ptr(), count))
        torch.cuda.synchronize()
        assert int(counter.item()) == 0
        assert bool((means == means[0]).all()) and bool((vars_ == vars_[0]).all()), "launches disagree (k=%d)" % k
        assert float((means[0] - mean0).abs().max()) <= 1e-6 * max(1.0, float(mean0.abs().max()))
        assert float(((vars_[0] - var0).abs() / (var0 + 1e-12)).max()) <= 1e-5


# dense-layer bottleneck: 1x1 over a channel prefix of a wider concat buffer
def test_conv1x1_prefix_to_128_bn_relu_stats(E):
    _run(E, 2, 32, 32, 96, 128, 1, pitch_in=256, bn=True, p_act=ACT_RELU, stats=True, seed=3)
    _run(E, 1, 16, 16, 992, 128, 1, pitch_in=1024, bn=True, p_act=ACT_RELU, stats=True, seed=4)


def test_conv1x1_legacy_chunk32_path(E):
    """1x1 convs whose filter does not fit LDS fall back to the LDS-tiled kernel (CHUNK32)."""
    _run(E, 2, 32, 32, 96, 128, 1, pitch_in=256, bn=True, p_act=ACT_RELU, stats=True, seed=3, layout=0)
    _run(E, 1, 16, 16, 992, 128, 1, pitch_in=1024, bn=True, p_act=ACT_RELU, stats=True, seed=4, layout=0)
    _run(E, 1, 16, 32, 64, 32, 1, pool=True, bias=True, pitch_out=160, seed=6, layout=0)
    _run(E, 2, 8, 8, 768, 128, 1, p_act=ACT_RELU, upsample=True, transposed=True, pitch_out=512, seed=8, layout=0)


def test_conv1x1_xstream_shapes(E):
    """x-stream kernel: ragged pixel counts, every BN width, NCHW fp32 output, many tiles per workgroup."""
    from fdgan_hip import lib as L
    assert L.load().fdgan_conv_weight_layout(128, 224, 1, 1) == L.WLAYOUT_X64
    assert L.load().fdgan_conv_weight_layout(128, 992, 1, 1) == L.WLAYOUT_X64      # streamed in k-groups
    assert L.load().fdgan_conv_weight_layout(32, 128, 3, 1) == L.WLAYOUT_CHUNK32
    _run(E, 3, 19, 23, 224, 128, 1, pitch_in=256, bn=True, p_act=ACT_RELU, stats=True, seed=21, running=True)
    _run(E, 2, 10, 14, 160, 64, 1, bias=True, e_act=ACT_RELU, stats=True, seed=22)
    _run(E, 1, 24, 40, 64, 32, 1, p_act=ACT_LEAKY02, seed=23)
    _run(E, 2, 12, 12, 384, 512, 1, p_act=ACT_RELU, seed=24)                      # grid.y = 4
    _run(E, 1, 16, 16, 40, 3, 1, bias=True, e_act=ACT_TANH, nchw_out=True, seed=25)
    _run(E, 8, 128, 128, 64, 128, 1, bn=True, p_act=ACT_RELU, stats=True, seed=26)  # > 1 tile per workgroup
    _run(E, 2, 36, 20, 512, 256, 1, bn=True, p_act=ACT_RELU, pool=True, stats=True, seed=27)
    # filters larger than LDS: streamed k-groups, 8-wave workgroups, 32-pixel wave tiles
    _run(E, 2, 24, 24, 992, 128, 1, pitch_in=1024, bn=True, p_act=ACT_RELU, stats=True, seed=28)
    _run(E, 1, 16, 16, 1024, 512, 1, bn=True, p_act=ACT_RELU, pool=True, seed=29)
    _run(E, 16, 64, 64, 768, 128, 1, p_act=ACT_RELU, upsample=True, transposed=True, pitch_out=512, seed=30)
    _run(E, 16, 128, 128, 480, 128, 1, pitch_in=512, bn=True, p_act=ACT_RELU, stats=True, seed=31)


def test_transition_pool_prologue(E):
    _run(E, 2, 32, 32, 256, 128, 1, bn=True, p_act=ACT_RELU, pool=True, stats=True, seed=5, running=True)
    _run(E, 1, 16, 32, 64, 32, 1, pool=True, bias=True, pitch_out=160, seed=6)           # conv_refin2
    _run(E, 1, 8, 8, 1024, 512, 1, bn=True, p_act=ACT_RELU, pool=True, seed=7)            # trans_block3


def test_decoder_convT_relu_upsample(E):
    _run(E, 2, 8, 8, 768, 128, 1, p_act=ACT_RELU, upsample=True, transposed=True, pitch_out=512, seed=8)
    _run(E, 1, 16, 16, 96, 16, 1, p_act=ACT_RELU, upsample=True, transposed=True, seed=9)


def test_wide_3x3_with_bias_and_partial_tiles(E):
    _run(E, 1, 20, 28, 160, 128, 3, pad=1, bias=True, stats=True, seed=10)                # conv_refine4, ragged
    _run(E, 2, 8, 8, 640, 512, 3, pad=1, bias=True, seed=11)                              # conv_refin6
    _run(E, 1, 8, 8, 1024, 256, 3, pad=1, p_act=ACT_RELU, pitch_out=768, c0_out=512, seed=12)


def test_first_and_last_conv(E):
    _run(E, 2, 32, 32, 3, 64, 3, pad=1, bias=True, e_act=ACT_RELU, stats=True, pitch_out=256, seed=13)
    _run(E, 2, 32, 32, 16, 3, 3, pad=1, bias=True, e_act=ACT_TANH, nchw_out=True, seed=14)


def test_image_reading_first_layers_small_cin_kernel(E):
    """csrc/conv_sc.hip (round 5): k = (tap, channel) with 4 / 16 channels per tap instead of one 32-channel chunk per tap."""
    _run(E, 1, 37, 45, 3, 64, 3, pad=1, bias=True, e_act=ACT_RELU, stats=True, pitch_out=256, seed=131)   # ragged tiles, slice of a concat buffer
    _run(E, 2, 64, 64, 3, 64, 3, pad=1, bias=True, e_act=ACT_RELU, seed=132)                              # VGG16 conv1_1
    _run(E, 1, 70, 66, 9, 36, 4, stride=2, pad=1, pitch_in=16, seed=133)                                   # D layer1, ragged
    _run(E, 1, 32, 32, 12, 40, 4, stride=2, pad=1, pitch_in=16, stats=True, seed=134)
    _run(E, 1, 16, 16, 4, 48, 3, pad=1, seed=135)


def test_discriminator_convs(E):
    _run(E, 2, 64, 64, 9, 36, 4, stride=2, pad=1, pitch_in=16, seed=15)                    # D layer1
    _run(E, 2, 32, 32, 36, 72, 3, pad=1, p_act=ACT_LEAKY02, stats=True, pitch_in=40, seed=16)
    _run(E, 1, 17, 19, 144, 288, 4, pad=1, bn=True, p_act=ACT_LEAKY02, seed=17)            # odd sizes (127-like)
    _run(E, 1, 16, 18, 288, 1, 4, pad=1, p_act=ACT_LEAKY02, e_act=ACT_SIGMOID, nchw_out=True, seed=18)
    # the one-filter conv on the matrix pipe (round 5, conv_cout1_mfma_kernel): two column blocks and three row bands with ragged ends,
    # 512 channels (A fragments from LDS) behind a BatchNorm prologue, and the 3x3 form
    _run(E, 2, 21, 140, 288, 1, 4, pad=1, p_act=ACT_LEAKY02, nchw_out=True, seed=181)
    _run(E, 1, 31, 31, 512, 1, 4, pad=1, bn=True, p_act=ACT_LEAKY02, e_act=ACT_SIGMOID, nchw_out=True, seed=182)
    _run(E, 1, 20, 130, 64, 1, 3, pad=1, p_act=ACT_RELU, nchw_out=True, seed=183)
    _run(E, 3, 127, 127, 288, 1, 4, pad=1, p_act=ACT_LEAKY02, e_act=ACT_SIGMOID, nchw_out=True, seed=184)


def test_plan_replay_matches_eager(E):
    """Record -> replay (and hipGraph) gives the same bytes as the eager launch."""
    dev = torch.device("cuda:0")
    x = torch.randn(1, 16, 16, 64, device=dev).to(torch.float16)
    w = torch.randn(32, 64, 3, 3, device=dev) * 0.05
    pw = E.PackedWeight(w, 32, 64, 3)
    pw.pack()
    y0 = torch.zeros(1, 16, 16, 32, dtype=torch.float16, device=dev)
    y1 = torch.zeros_like(y0)
    d = E.conv_desc(3, 1, 1, cout=32)
    E.conv2d(E.View(x).fd, pw, None, None, E.View(y0).fd, d)
    plan = E.Plan()
    with plan.record():
        E.conv2d(E.View(x).fd, pw, None, None, E.View(y1).fd, d)
    torch.cuda.synchronize()
    assert (y1 == 0).all() and len(plan) == 1 and plan.kernel_names() == ["conv3x3_pw_bn32"]
    plan.launch()
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    y1.zero_()
    plan.instantiate_graph()
    plan.launch()
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
