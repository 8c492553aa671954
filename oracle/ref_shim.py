"""Import the *unmodified* reference (AliaksandrSiarohin/monkey-net) on CPU.  TEST INFRASTRUCTURE ONLY.

Works only where /root/reference exists (the authoring container); nothing that runs on the GPU box
may call this.  Three external pins are installed before the reference is imported
(SURVEY.md section 8c):

* ``torch.gesv`` was removed from torch; ``modules/util.py:223`` calls ``torch.gesv(eye, b_mat)``
  and takes element 0 -> ``torch.linalg.solve(b_mat, eye)``.
* ``F.grid_sample`` defaulted to ``align_corners=True`` in torch 0.4.1 (the version the reference
  pins, requirements.txt:25); call sites ``modules/movement_embedding.py:85`` and
  ``modules/generator.py:57`` rely on it because ``make_coordinate_grid`` (``modules/util.py:26-42``)
  builds an align_corners=True grid.
* third-party modules that are not installed here (imageio, skimage, ...) are stubbed so that
  ``train.py``'s ``GeneratorFullModel`` / ``DiscriminatorFullModel`` can be imported.
"""
import os
import sys
import types

import torch
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("MNK_REFERENCE_ROOT", "/root/reference")

_installed = False


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modules"))


def install_torch_pins():
    """torch.gesv and the align_corners default of F.grid_sample as torch 0.4.1 had them.  Idempotent."""
    if not hasattr(torch, "gesv"):
        torch.gesv = lambda B, A: (torch.linalg.solve(A, B), None)
    if not getattr(F.grid_sample, "_mnk_pinned", False):
        _orig = F.grid_sample

        def grid_sample(input, grid, mode="bilinear", padding_mode="zeros", align_corners=None):
            return _orig(input, grid, mode=mode, padding_mode=padding_mode, align_corners=True)

        grid_sample._mnk_pinned = True
        F.grid_sample = grid_sample


def install_stubs():
    """Empty stand-ins for the third-party packages of the reference's callers that are not installed here."""
    for name in ("imageio", "skimage", "skimage.draw", "skimage.io", "skimage.transform", "skimage.color",
                 "skimage.util", "matplotlib", "matplotlib.pyplot", "sklearn", "sklearn.model_selection", "torchvision",
                 "PIL", "pandas", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                m.__dict__.update(circle=None, imread=None, mimread=None, resize=None, rotate=None,
                                  gray2rgb=None, img_as_ubyte=None, img_as_float32=None, img_as_float=None, pad=None,
                                  train_test_split=None, io=None, use=lambda *a, **k: None)
                sys.modules[name] = m


def install():
    """Install the pins and put the reference on sys.path.  Idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    install_torch_pins()
    install_stubs()
    # our drop-in packages are also called `modules` / `sync_batchnorm`: make sure the reference's win here
    for name in list(sys.modules):
        if name == "modules" or name.startswith("modules.") or name == "sync_batchnorm" or \
                name.startswith("sync_batchnorm."):
            del sys.modules[name]
    sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


def load():
    """Return a namespace with the reference classes used by the golden generator."""
    install()
    from modules.generator import MotionTransferGenerator
    from modules.keypoint_detector import KPDetector, kp2gaussian, gaussian2kp
    from modules.dense_motion_module import DenseMotionModule
    from modules.movement_embedding import MovementEmbeddingModule
    from modules.discriminator import Discriminator
    from modules import util, losses
    ns = types.SimpleNamespace(MotionTransferGenerator=MotionTransferGenerator, KPDetector=KPDetector,
                               kp2gaussian=kp2gaussian, gaussian2kp=gaussian2kp,
                               DenseMotionModule=DenseMotionModule,
                               MovementEmbeddingModule=MovementEmbeddingModule,
                               Discriminator=Discriminator, util=util, losses=losses)
    try:
        import train as ref_train
        ns.GeneratorFullModel = ref_train.GeneratorFullModel
        ns.DiscriminatorFullModel = ref_train.DiscriminatorFullModel
    except Exception as e:  # pragma: no cover
        ns.train_import_error = e
    return ns
