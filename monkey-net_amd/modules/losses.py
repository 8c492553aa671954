"""Training losses (modules/losses.py): L1 feature matching over discriminator maps + LSGAN terms; same function names,
argument order and return conventions as the reference so train.py runs unchanged.  The L1 terms of fp32 tensors on the
library's device run the one-launch kernels of csrc/losses.hip (forward: |a - b| reduced per sample, backward: the sign
pass for whichever side asks for a gradient); anything else (the fp64 checks of the tests, the few-element LSGAN terms)
is stock PyTorch arithmetic.  mnk.engine.TrainStep goes further and takes the feature-matching terms straight from the
NHWC activations of the batched discriminator pass (ops.PairL1Fn, ops.GanTermsFn)."""
import torch

from mnk import _lib, ops


def mean_batch(val):
    return val.reshape(val.shape[0], -1).mean(-1)


def _on_library_device(*tensors):
    """fp32 tensors of one shape, where the kernels run (cuda for the gfx950 build)."""
    first = tensors[0]
    if not all(torch.is_tensor(t) and t.dtype == torch.float32 and t.shape == first.shape and t.dim() >= 2 and
               t.numel() > 0 for t in tensors):
        return False
    return all(t.is_cuda for t in tensors) if _lib.lib().is_device_build else not any(t.is_cuda for t in tensors)


def reconstruction_loss(prediction, target, weight):
    if weight == 0:
        return 0
    if _on_library_device(prediction, target):
        return ops.L1MeanFn.apply(prediction, target, weight)
    return weight * mean_batch((prediction - target).abs())


def generator_gan_loss(discriminator_maps_generated, weight):
    return weight * mean_batch((1 - discriminator_maps_generated[-1]) ** 2)


def discriminator_gan_loss(discriminator_maps_generated, discriminator_maps_real, weight):
    return weight * mean_batch((1 - discriminator_maps_real[-1]) ** 2 + discriminator_maps_generated[-1] ** 2)


def generator_loss_names(loss_weights):
    names = []
    if loss_weights['reconstruction_deformed'] != 0:
        names.append("rec_def")
    if loss_weights['reconstruction'] is not None:
        names += ["layer-%s_rec" % i for i, wgt in enumerate(loss_weights['reconstruction']) if wgt != 0]
    names.append("gen_gan")
    return names


def discriminator_loss_names():
    return ['disc_gan']


def generator_loss(discriminator_maps_generated, discriminator_maps_real, video_deformed, loss_weights):
    values = []
    if loss_weights['reconstruction_deformed'] != 0:
        values.append(reconstruction_loss(discriminator_maps_real[0], video_deformed,
                                          loss_weights['reconstruction_deformed']))
    if loss_weights['reconstruction'] != 0:
        pairs = zip(discriminator_maps_real[:-1], discriminator_maps_generated[:-1])
        for i, (real, fake) in enumerate(pairs):
            if loss_weights['reconstruction'][i] != 0:
                values.append(reconstruction_loss(fake, real, weight=loss_weights['reconstruction'][i]))
    values.append(generator_gan_loss(discriminator_maps_generated, weight=loss_weights['generator_gan']))
    return values


def discriminator_loss(discriminator_maps_generated, discriminator_maps_real, loss_weights):
    return [discriminator_gan_loss(discriminator_maps_generated, discriminator_maps_real,
                                   loss_weights['discriminator_gan'])]
