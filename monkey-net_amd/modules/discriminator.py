"""Patch discriminator of the training step (modules/discriminator.py; SURVEY.md section 8f-1, the first "next" row).

`Discriminator` / `DownBlock3D` are the gfx950-kernel classes of `mnk/discriminator_hip.py` (4x4 no-pad implicit-GEMM
convolutions, fused InstanceNorm + LeakyReLU + avg-pool, 1x1 score head).  They keep the reference's constructor,
state_dict keys (5-D conv weights) and forward signature, so checkpoints and train.py interoperate; the comparison path
of the tests is the oracle (oracle/restate.py), there is no second backend."""
from mnk.discriminator_hip import Discriminator, DownBlock3D  # noqa: F401
