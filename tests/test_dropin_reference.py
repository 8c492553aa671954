"""The drop-in claim, executed: the reference's OWN callers -- run.py's imports, train.train() with its optimisers /
DataLoader / Logger / DataParallelWithCallback(device_ids=...), reconstruction.generate, transfer.transfer_one +
normalize_kp, Logger.save_cpk / load_cpk -- run unmodified on top of monkey-net_amd/modules and
monkey-net_amd/sync_batchnorm (kernels on the CPU emulator) and reproduce what the reference recorded with its own
modules.  Needs the reference tree (authoring container); each scenario is a fresh process (tests/dropin_worker.py)
because the package names collide with the ones this pytest session has already imported."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MNK_REFERENCE_ROOT", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "modules")),
                                reason="the reference tree exists only in the authoring container")


def _run(scenario, timeout=600):
    env = dict(os.environ, OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_worker.py"), scenario], env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DROPIN_RESULT ")][-1]
    return json.loads(line[len("DROPIN_RESULT "):])


def test_reference_run_py_imports_resolve_to_the_drop_in():
    assert _run("imports")["ok"]


def test_reference_train_loop_and_checkpoint_round_trip():
    res = _run("train")
    assert len(res["report"]) == 3


def test_reference_reconstruction_generate():
    res = _run("reconstruction")
    assert res["max_err"] < 2e-5


def test_reference_transfer_one_with_normalize_kp():
    res = _run("transfer")
    assert res["video_prediction"] < 5e-5


def test_batched_transfer_and_device_normalize_kp_match_the_reference_loop():
    res = _run("transfer_batched")
    assert res["video_prediction"] < 2e-5 and res["repair_var"] < 2e-6


def test_launcher_puts_the_drop_in_first(tmp_path):
    """monkey-net_amd/run_reference.py: `python run.py` would resolve `modules` in the script's own directory."""
    script = tmp_path / "probe.py"
    script.write_text("import modules.generator, modules.prediction_module, sync_batchnorm, sys\n"
                      "print(modules.generator.__file__); print(modules.prediction_module.__file__); "
                      "print(sync_batchnorm.__file__); print(sys.argv[1:])\n")
    os.symlink(os.path.join(REF, "modules"), tmp_path / "modules")
    os.symlink(os.path.join(REF, "sync_batchnorm"), tmp_path / "sync_batchnorm")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "monkey-net_amd", "run_reference.py"), str(script), "--config",
                          "x.yaml"], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert "monkey-net_amd/modules/generator.py" in lines[0]
    assert str(tmp_path) in lines[1] and "prediction_module" in lines[1]
    assert "monkey-net_amd/sync_batchnorm" in lines[2]
    assert lines[3] == "['--config', 'x.yaml']"
