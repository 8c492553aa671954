"""The C-ABI shared library loads without a GPU and exports every symbol include/monkeynet_hip.h declares;
argument validation returns MNK_EINVAL before anything is launched."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def real_lib():
    import __graft_entry__ as ge
    from mnk import _lib
    if not os.path.exists(_lib.DEFAULT_LIB):
        ge.build()
    return _lib.Library(_lib.DEFAULT_LIB, strict=True)


def test_every_declared_symbol_is_exported(real_lib):
    from mnk import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 40
    for name in protos:
        assert hasattr(real_lib.cdll, name), name
    assert real_lib.cdll.mnk_version() >= 100
    assert real_lib.is_device_build


def test_invalid_arguments_are_rejected_before_launch(real_lib):
    from mnk import _lib
    rc = real_lib.cdll.mnk_bn_stats(None, 4, 10, 3, None, None, 0, None)
    assert rc == -1
    assert b"invalid argument" in real_lib.cdll.mnk_last_error()
    with pytest.raises(_lib.MnkError):
        real_lib.call("mnk_conv3x3_fwd", None, 4, 3, None, 0, 0, 0, None, None, None, 0, None, 4, 1, 8, 8, 4, None, 0,
                      None, None)
    assert real_lib.query("mnk_conv3x3_packed_floats", 64, 3, 0) == 64 * 9 * 16
    assert real_lib.query("mnk_conv3x3_workspace_floats", 32, 64, 64, 64, 0, 64) == 0      # no split-K needed
    assert real_lib.query("mnk_conv3x3_workspace_floats", 32, 4, 4, 1024, 0, 1024) > 0     # deep level: split-K


def test_generated_cpython_binding_is_the_same_abi(real_lib):
    """monkey-net_amd/_mnkfast (csrc/gen_fastcall.py, generated from the header): every entry point without a char* in its
    prototype has a wrapper bound to THIS library's address; the wrapper returns what the ctypes call returns, reports a
    wrong argument count, passes status codes on, and hands an argument it does not take (a ctypes object) back to ctypes."""
    import ctypes
    import _mnkfast
    from mnk import _lib
    names = set(_mnkfast.names())
    assert len(names) >= len(real_lib.protos) - 8 and names <= set(real_lib.protos)
    assert set(real_lib.fast) == names
    for args in ((64, 3, 0), (128, 64, 32), (45, 45, 0)):
        assert real_lib.fast["mnk_conv3x3_packed_floats"](*args) == real_lib.cdll.mnk_conv3x3_packed_floats(*args)
    assert real_lib.fast["mnk_version"]() == real_lib.cdll.mnk_version()
    with pytest.raises(TypeError):
        real_lib.fast["mnk_conv3x3_packed_floats"](64, 3)
    assert real_lib.fast["mnk_bn_stats"](None, 4, 10, 3, None, None, 0, None) == -1            # MNK_EINVAL, nothing launched
    assert real_lib.fast["mnk_conv3x3_packed_floats"](64.5, 3, 0) is NotImplemented              # not an int: ctypes' turn
    plan = (ctypes.c_int * 8)()
    assert real_lib.fast["mnk_conv2d_wgrad_plan2"](32, 8, 8, 64, 64, 3, 3, 1, 64, 0, ctypes.byref(plan)) is NotImplemented
    assert real_lib.query("mnk_conv2d_wgrad_plan2", 32, 8, 8, 64, 64, 3, 3, 1, 64, 0, ctypes.byref(plan)) == 0
    # a second library (here: the same file under a second handle) binds its own addresses
    other = _lib.Library(_lib.DEFAULT_LIB, strict=True)
    assert other.query("mnk_conv3x3_packed_floats", 64, 3, 0) == 64 * 9 * 16


def test_missing_library_fails_loudly(tmp_path):
    from mnk import _lib
    with pytest.raises(_lib.MnkError):
        _lib.Library(str(tmp_path / "nope.so"))


def test_product_never_imports_the_oracle():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/."""
    pkg = os.path.join(ROOT, "monkey-net_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)
                assert "hipemu" not in text, os.path.join(dirpath, f)


def test_integration_stub_runs(be):
    """The ctypes binding printed in INTEGRATION.md section 2 is executed as written (library path, device and stream
    substituted for the backend under test) and reproduces DownBlock3D.forward of the reference."""
    import re
    import torch
    import torch.nn.functional as F
    from torch import nn
    from _util import to_nhwc
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch\n.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libmonkeynet_hip.so")', "ctypes.CDLL(LIB_PATH)")
    if be.kind == "emu":
        code = code.replace("torch.cuda.current_stream().cuda_stream", "None")
    ns = {"LIB_PATH": be.lib.path}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    torch.manual_seed(2)
    n, h, w, cin, cout = 3, 8, 8, 5, 14
    conv = nn.Conv3d(cin, cout, (1, 3, 3), padding=(0, 1, 1)).to(be.device)
    norm = nn.BatchNorm3d(cout).to(be.device)
    x = torch.rand(n, cin, h, w)
    z = ns["down_block"](be.t(to_nhwc(x)), conv, norm, n, h, w, cin, cout)
    be.sync()
    ref = F.avg_pool2d(F.relu(F.batch_norm(F.conv2d(x.double(), conv.weight.detach().cpu()[:, :, 0].double(),
                                                    conv.bias.detach().cpu().double(), padding=1),
                                           None, None, torch.ones(cout).double(), torch.zeros(cout).double(), True)), 2)
    assert float((z.cpu()[..., :cout].permute(0, 3, 1, 2).double() - ref).abs().max()) < 2e-5
