"""The bf16-matrix-core form of the fp32 GEMMs (tuning value gemm_bf16x3; csrc/mnk_common.h) against the fp32 MFMA chain it
replaces -- tests/test_kernels_conv.py runs every kernel test of the family in both modes with one set of tolerances; this file
compares the two modes' errors directly on the layer shapes of the BASELINE configurations (MI355X only)."""
import pytest

from _util import from_nhwc
from test_kernels_conv import _inputs, _ref_fwd, _run_fwd


@pytest.mark.gpu
def test_the_bf16_split_gemm_is_as_accurate_as_the_fp32_mfma_chain():
    """`gemm_bf16x3` is not a reduced-precision mode: on the layer shapes of BASELINE configs[1] / [2] (batch 4) the forward
    convolution's error against an fp64 convolution with the three-way-split bf16 products is within 1.25 x the error of the
    fp32 MFMA chain (measured: 0.8 ... 1.05 x -- each bf16 x bf16 product is exact, the three dropped cross terms are below the
    rounding of an fp32 product, and the MFMA adds in fp32 either way)."""
    from conftest import Backend
    be = Backend("hip")
    shapes = [(4, 32, 32, 64, 0, 128), (4, 16, 16, 128, 0, 256), (4, 8, 8, 256, 0, 512), (4, 4, 4, 512, 0, 1024),
              (4, 2, 2, 1024, 0, 1024), (4, 32, 32, 138, 128, 64), (4, 64, 64, 64, 0, 64), (4, 8, 8, 522, 0, 128)]
    rows = []
    for (n, h, w, c0, c1, cout) in shapes:
        case = (n, h, w, c0, c1, cout, 0, True, False)
        x0, x1, wt, b, r = _inputs(case, seed=3)
        ref = _ref_fwd(case, x0, x1, wt, b, r)
        errs = []
        for mode in (0, 1):
            be.lib.call("mnk_set_tuning", b"gemm_bf16x3", mode)
            try:
                y = from_nhwc(_run_fwd(be, case, x0, x1, wt, b, r, clean=True), cout)
            finally:
                be.lib.call("mnk_set_tuning", b"gemm_bf16x3", 0)
            errs.append(float((y.double() - ref).abs().max()))
        rows.append((case[:6], errs[0], errs[1]))
        assert errs[1] <= 1.25 * errs[0] + 1e-7, (case, errs)
    print("max |conv - fp64|, fp32 MFMA vs bf16 split:", rows)
