import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "monkey-net_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


import _util  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "trajectory: long multi-iteration training runs; collected LAST so that a failure "
                            "there (pytest -x) cannot hide the kernel / module / oracle tests")


# files whose tests run several full-size training iterations: they go to the end of the collection, after every kernel,
# module and oracle test (the driver runs `pytest -x`: in round 4 one such test stopped the run in front of 361 others)
_LATE_FILES = ("test_fullsize.py", "test_dist_graph_gpu.py", "test_p2p_gpu.py", "test_train_sanity.py", "test_bench_guard.py")


def pytest_collection_modifyitems(config, items):
    def rank(item):
        if item.get_closest_marker("trajectory") is not None:
            return 2
        return 1 if os.path.basename(str(item.fspath)) in _LATE_FILES else 0
    items.sort(key=rank)            # stable: the order inside each class stays the collection order


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`pytest tests -m "not gpu"` (the CPU suite: kernels on the fiber emulator, ~10 min in one process) runs on four worker
    processes when pytest-xdist is installed and no -n / -p no:xdist was given -- about 3 min.  The GPU suite (-m gpu) and any
    other selection are left alone: its tests share one device (and some spawn processes of their own).  MNK_TESTS_SERIAL=1
    keeps one process."""
    opt = config.option
    if (os.environ.get("MNK_TESTS_SERIAL") == "1" or os.environ.get("PYTEST_XDIST_WORKER")
            or not config.pluginmanager.hasplugin("xdist") or getattr(opt, "numprocesses", None) is not None
            or (getattr(opt, "markexpr", "") or "").strip() != "not gpu" or getattr(opt, "collectonly", False)
            or getattr(opt, "usepdb", False) or (os.cpu_count() or 1) < 4):
        return None
    opt.numprocesses = 4
    return None


_EMU = {}


def emu_library_path():
    """Build (once per session) the CPU emulation of the kernels -- tests/hipemu, test infrastructure only."""
    if "path" not in _EMU:
        import fcntl
        os.makedirs(os.path.join(ROOT, "tests", "hipemu", "build"), exist_ok=True)
        # one build at a time: the workers of `pytest -n N` all get here, and a worker must not dlopen the library while another
        # one's link step is rewriting it
        with open(os.path.join(ROOT, "tests", "hipemu", "build", ".lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            out = subprocess.run([os.path.join(ROOT, "tests", "hipemu", "build.sh")], capture_output=True, text=True,
                                 stdin=subprocess.DEVNULL)
            fcntl.flock(lock, fcntl.LOCK_UN)
        if out.returncode != 0:
            raise RuntimeError("hipemu build failed:\n" + out.stdout + out.stderr)
        _EMU["path"] = out.stdout.strip().splitlines()[-1]
    return _EMU["path"]


class Backend:
    """Where a kernel test runs: 'emu' = kernel sources compiled for the CPU emulator (CPU tensors),
    'hip' = the real gfx950 library on cuda:0."""

    def __init__(self, kind):
        import torch
        from mnk import _lib
        self.kind = kind
        if kind == "emu":
            self.lib = _util.set_library(emu_library_path(), strict=False)
            self.device = torch.device("cpu")
        else:
            if not torch.cuda.is_available():
                pytest.skip("no GPU")
            self.lib = _util.set_library(None) or _lib.lib()
            assert self.lib.is_device_build
            self.device = torch.device("cuda:0")

    def t(self, x):
        return x.to(self.device).contiguous()

    def zeros(self, *shape):
        import torch
        return torch.zeros(*shape, device=self.device)

    def empty(self, *shape):
        import torch
        return torch.full(shape, float("nan"), device=self.device)

    def stream(self):
        import torch
        return torch.cuda.current_stream().cuda_stream if self.kind == "hip" else 0

    def call(self, name, *args):
        import torch
        conv = [a.data_ptr() if torch.is_tensor(a) else a for a in args]
        self.lib.call(name, *conv, self.stream())

    def query(self, name, *args):
        return self.lib.query(name, *args)

    def sync(self):
        import torch
        if self.kind == "hip":
            torch.cuda.synchronize()


@pytest.fixture(params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    b = Backend(request.param)
    yield b
    from mnk import _lib
    _util.set_library(None)


@pytest.fixture
def make_backend():
    """Backend(kind) for tests that name their backends themselves (so that no gpu-marked instance of a test builds the CPU
    emulator); the product's library handle is put back afterwards like `be` does."""
    yield Backend
    _util.set_library(None)
