"""Is the guard allocator itself sound on this box?  (tools/dbg/guard_alloc.cpp; round 6: under it several GPU tests returned wrong
numbers without any fault.)  torch fills / copies of tensors of growing size, checked on the host.
    hipcc -O2 -w -shared -fPIC -o /tmp/libguard_alloc.so tools/dbg/guard_alloc.cpp && python tools/dbg/guard_alloc_selftest.py"""
import torch
alloc = torch.cuda.memory.CUDAPluggableAllocator("/tmp/libguard_alloc.so", "guard_malloc", "guard_free")
torch.cuda.memory.change_current_allocator(alloc)
bad = 0
for n in (1000, 4096, 100_000, 1 << 20, (1 << 20) + 8, 3_000_000, 1 << 23, 50_000_000):
    x = torch.full((n,), 9.0, device="cuda")
    ok_fill = bool((x.cpu() == 9.0).all())
    h = torch.arange(n, dtype=torch.float32)
    d = h.cuda()
    ok_h2d = bool((d.cpu() == h).all())
    y = (d * 2.0)
    ok_kernel = bool((y.cpu() == h * 2.0).all())
    z = x.half().float()
    ok_cast = bool((z.cpu() == 9.0).all())
    first_bad = -1
    if not ok_fill:
        first_bad = int((x.cpu() != 9.0).nonzero()[0])
    print("n = %9d: fill %s  h2d/d2h %s  kernel %s  cast %s  first bad element of the fill %d" % (n, ok_fill, ok_h2d, ok_kernel, ok_cast, first_bad), flush=True)
    bad += not (ok_fill and ok_h2d and ok_kernel and ok_cast)
print("guard allocator self-test:", "FAILED on %d sizes" % bad if bad else "passed")
