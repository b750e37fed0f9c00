"""How far are gfx950's one-instruction v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 from the correctly rounded results the numerics contract uses?
Evaluates them over every mantissa (2^23 values in [1, 2) for rcp; [1, 4) for rsq / sqrt: both exponent parities) through
nrdHipEvalNumerics ops 16-18 and compares with the correctly rounded answers (float64 on the host, rounded once). Prints the histogram of
the deviation in ulps and how well a per-mantissa delta table compresses -- the data needed to decide whether a table-emulated oracle could
follow kernels that use the hardware instructions. usage: python tools/hw_transcendentals.py"""
import os
import sys
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracingdenoiser_amd import api  # noqa: E402


def main():
    lib = api.load_library()
    stream = torch.cuda.current_stream().cuda_stream
    for op, name, lo, hi, exact in ((16, "v_rcp_f32", 1.0, 2.0, lambda x: 1.0 / x), (17, "v_rsq_f32", 1.0, 4.0, lambda x: 1.0 / np.sqrt(x)), (18, "v_sqrt_f32", 1.0, 4.0, np.sqrt)):
        bits = np.arange(np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32), dtype=np.uint32)
        x = torch.from_numpy(bits.view(np.float32).copy()).cuda()
        out = torch.empty_like(x)
        assert lib.nrdHipEvalNumerics(op, x.data_ptr(), None, out.data_ptr(), x.numel(), stream) == 0
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        want = exact(bits.view(np.float32).astype(np.float64)).astype(np.float32)  # one rounding of the (practically exact) float64 result
        delta = got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)
        hist = {int(k): int(v) for k, v in zip(*np.unique(delta, return_counts=True))}
        packed = zlib.compress(delta.astype(np.int8).tobytes(), 9)
        if os.environ.get("NRD_HW_TABLE_DIR") and op in (17, 18):  # the tables the oracle emulates v_rsq_f32 / v_sqrt_f32 with (oracle/hw_math.h)
            with open(os.path.join(os.environ["NRD_HW_TABLE_DIR"], {17: "hw_rsq.i8.z", 18: "hw_sqrt.i8.z"}[op]), "wb") as fp:
                fp.write(packed)
        print("%s over %d inputs in [%g, %g): deviation from the correctly rounded result in ulps -> count: %s; exact %.2f %%; int8 delta table %d bytes zlib-compressed" %
              (name, len(bits), lo, hi, hist, 100.0 * hist.get(0, 0) / len(bits), len(packed)))
    # special inputs
    specials = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1e-45, 1e-39, -1e-39, 3.4e38, -1.0], dtype=torch.float32, device="cuda")
    for op, name in ((16, "v_rcp_f32"), (17, "v_rsq_f32"), (18, "v_sqrt_f32")):
        out = torch.empty_like(specials)
        lib.nrdHipEvalNumerics(op, specials.data_ptr(), None, out.data_ptr(), specials.numel(), stream)
        torch.cuda.synchronize()
        print(name, "specials", dict(zip([float(v) for v in specials.cpu()], [float(v) for v in out.cpu()])))


if __name__ == "__main__":
    main()
