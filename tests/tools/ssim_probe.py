"""Runs the fused SSIM forward / backward kernels alone at 1080p (for rocprofv3 --pmc and quick timing).
usage: python3 tests/tools/ssim_probe.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sparse-view-3dgs-pack_amd"))
import torch  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402

api = dgr._C.backend.api
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
H, W = 1080, 1920
g = torch.Generator().manual_seed(0)
a = torch.rand((3, H, W), generator=g).cuda()
b = torch.rand((3, H, W), generator=g).cuda()
d1, d2, d3, grad = (torch.empty_like(a) for _ in range(4))
s = torch.zeros(4, device="cuda")
coef = torch.full((1,), 1e-6, device="cuda")
st = None
import ctypes as C  # noqa: E402
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
s9 = torch.zeros(16, device="cuda")
part = torch.empty(int(api.raw("ssim_partials_count")(1, 3, H, W)), device="cuda")
for name in ("fwd", "fwdmap", "fwdpart", "bwd", "dwt"):
    for it in range(reps + 3):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if name == "fwd":
            api.call("ssim_fwd_sum", a.data_ptr(), b.data_ptr(), 1, 3, H, W, 1e-4, 9e-4, s.data_ptr(), d1.data_ptr(),
                     d2.data_ptr(), d3.data_ptr(), st)
        elif name == "fwdmap":  # the same kernel without the per-workgroup atomic (writes the map instead)
            api.call("ssim_fwd", a.data_ptr(), b.data_ptr(), 1, 3, H, W, 1e-4, 9e-4, grad.data_ptr(), d1.data_ptr(),
                     d2.data_ptr(), d3.data_ptr(), st)
        elif name == "fwdpart":
            api.call("ssim_fwd_partials", a.data_ptr(), b.data_ptr(), 1, 3, H, W, 1e-4, 9e-4, part.data_ptr(),
                     d1.data_ptr(), d2.data_ptr(), d3.data_ptr(), st)
        elif name == "dwt":
            api.call("l1_dwt2_fwd", a.data_ptr(), b.data_ptr(), 3, H, W, s9.data_ptr(), s9[2:].data_ptr(), st)
        else:
            api.call("ssim_bwd_uniform", a.data_ptr(), b.data_ptr(), 1, 3, H, W, coef.data_ptr(), d1.data_ptr(),
                     d2.data_ptr(), d3.data_ptr(), grad.data_ptr(), 0, a.data_ptr(), st)
    torch.cuda.synchronize()
    print(name, "%.1f us" % ((time.perf_counter() - t0) / reps * 1e6))
