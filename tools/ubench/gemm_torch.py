"""What does the vendor GEMM (hipBLASLt through torch.matmul) reach on this board at the shapes of our MFMA-bound convs?
(Calibration for the MFMA-family fractions in DESIGN.md: D's 4x4 144->288 conv is M = 16*127*127 pixels, N = 288, K = 2304.)"""
import torch
dev = "cuda:0"
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for dt in (torch.bfloat16, torch.float16):
    for name, M, N, K in (("square 8192", 8192, 8192, 8192), ("D layer4 fwd", 16 * 127 * 127, 288, 2304), ("D layer4 dgrad", 16 * 128 * 128, 144, 4608),
                          ("D layer4 wgrad", 288, 2304, 16 * 127 * 127), ("VGG 3x3 256->256 @64^2", 16 * 64 * 64, 256, 2304),
                          ("VGG 3x3 128->128 @128^2", 16 * 128 * 128, 128, 1152), ("growth 3x3 128->32 @256^2", 16 * 256 * 256, 32, 1152)):
        a = torch.randn(M, K, device=dev, dtype=dt); b = torch.randn(K, N, device=dev, dtype=dt)
        s = t(lambda: torch.matmul(a, b))
        print("%-8s %-28s M %8d N %5d K %8d  %8.1f us  %6.1f TFLOP/s" % (str(dt).split(".")[1], name, M, N, K, s * 1e6, 2.0 * M * N * K / s / 1e12), flush=True)
        del a, b
