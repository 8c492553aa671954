"""Training losses (modules/losses.py): L1 feature matching over discriminator maps + LSGAN terms.  Tiny
element-wise reductions on stock PyTorch ops (out of the hot path); same function names,
argument order and return conventions as the reference so train.py runs unchanged."""
import torch


def mean_batch(val):
    return val.reshape(val.shape[0], -1).mean(-1)


def reconstruction_loss(prediction, target, weight):
    if weight == 0:
        return 0
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
