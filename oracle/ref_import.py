"""TEST INFRASTRUCTURE (oracle).  Imports the *real* reference modules from
/root/reference behind the three shims of SURVEY Appendix C.  Only usable in
the build container (the reference never travels to the GPU box); used by
oracle/make_golden.py to validate the restatement and to generate the golden
vectors committed under tests/golden/.
"""
import ast
import sys
import types

import numpy as np
import torch.nn as nn

REF_ROOT = "/root/reference"


def import_reference():
    """Returns (dehaze1113, dehaze22, Vgg16, pytorch_ssim) reference modules."""
    sys.dont_write_bytecode = True                      # reference tree is read-only
    from . import densenet121 as _dn
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.densenet121 = _dn.densenet121                   # torchvision is absent: restated topology
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    orig = nn.Module.add_module

    def add_module(self, name, module):                 # legacy dotted child names (D, dehaze22.*)
        if "." in name:
            self._modules[name] = module
        else:
            orig(self, name, module)
    nn.Module.add_module = add_module
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import models.dehaze1113 as ref1113
    import models.dehaze22 as ref22
    import models.pytorch_ssim as ref_ssim
    from myutils.vgg16 import Vgg16 as RefVgg16
    return ref1113, ref22, RefVgg16, ref_ssim


def import_reference_metrics():
    """ast-extracts crop / compare_ssim / output_psnr_mse from the reference's
    PSNRSSIM.py (the file runs argparse + os.listdir at import, :15-18,:253)."""
    from scipy.ndimage import gaussian_filter, uniform_filter
    src = open(REF_ROOT + "/PSNRSSIM.py").read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef)
            and n.name in ("crop", "compare_ssim", "output_psnr_mse")]
    mod = ast.Module(body=keep, type_ignores=[])
    dtype_range = {np.bool_: (False, True), np.uint8: (0, 255), np.uint16: (0, 65535),
                   np.int8: (-128, 127), np.int16: (-32768, 32767),
                   np.float16: (-1, 1), np.float32: (-1, 1), np.float64: (-1, 1)}
    ns = {"np": np, "uniform_filter": uniform_filter, "gaussian_filter": gaussian_filter,
          "dtype_range": dtype_range,
          "_validate_lengths": lambda ar, w: [(w, w)] * np.ndim(ar)}
    exec(compile(mod, REF_ROOT + "/PSNRSSIM.py", "exec"), ns)

    def crop(ar, crop_width, copy=False, order="K"):     # numpy-2: index with a tuple of slices
        ar = np.asarray(ar)
        return ar[tuple(slice(crop_width, s - crop_width) for s in ar.shape)]
    ns["crop"] = crop
    return ns["compare_ssim"], ns["output_psnr_mse"]
