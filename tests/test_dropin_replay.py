"""The reference's caller sequence WITHOUT the reference tree (so that it also runs on the MI355X box, where /root/reference
does not exist; tests/test_dropin_reference.py runs the unmodified reference files themselves, authoring container only).

What train.train() does with the drop-in packages (train.py:78-153), statement for statement: three torch.optim.Adam
optimisers with MultiStepLR schedulers, a DataLoader over a Dataset of {'source', 'video'} dicts, the two full models wrapped
in DataParallelWithCallback(device_ids=[0]), the loop `out = generator_full_par(x) ... loss.backward(); optimizer.step()`
with host copies of the losses every iteration, a checkpoint in Logger.save_cpk's dict layout (logger.py:43-47) and its
restore by Logger.load_cpk's statements (:49-66) -- against the loss history the REFERENCE recorded with its own modules
(tests/golden/step_tiny.pt, oracle/make_golden.py::step_case)."""
import copy
import os

import torch
from torch.optim.lr_scheduler import MultiStepLR
from torch.utils.data import DataLoader, Dataset

from oracle import cases
from test_modules import build, load


import pytest


@pytest.mark.parametrize("adam", ["torch", "mnk"])
def test_train_py_caller_sequence_replayed_on_the_drop_in_modules(be, tmp_path, adam):
    """adam = "mnk": the same loop with train.py:81-83's three constructors replaced by mnk.optim.MnkAdam (INTEGRATION.md
    section 1.5: the one change to the reference's loop that is worth making -- weight-gradient GEMMs of a backward pass launched
    together into the optimiser's flat buffer, one Adam launch per optimiser that also writes the packed weights)."""
    from mnk.engine import GeneratorFullModel, DiscriminatorFullModel          # train.py:24-75 restated (mnk/engine.py)
    from mnk.optim import MnkAdam
    from sync_batchnorm import DataParallelWithCallback
    Adam = torch.optim.Adam if adam == "torch" else MnkAdam
    gold = load("step_tiny")
    config = copy.deepcopy(gold["cfg"])
    train_params = config["train_params"]
    train_params.update(num_epochs=3, epoch_milestones=[], batch_size=gold["batch"])
    generator, discriminator, kp_detector = build(config)
    generator.load_state_dict(gold["state"]["generator"])
    discriminator.load_state_dict(gold["state"]["discriminator"])
    kp_detector.load_state_dict(gold["state"]["kp_detector"])
    for m in (generator, discriminator, kp_detector):          # run.py:52,58,64: .to(opt.device_ids[0])
        m.to(be.device)
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])

    class Pairs(Dataset):
        def __len__(self):
            return src.shape[0]

        def __getitem__(self, i):
            return {"source": src[i], "video": drv[i]}

    # train.py:81-99
    optimizer_generator = Adam(generator.parameters(), lr=train_params['lr'], betas=(0.5, 0.999))
    optimizer_discriminator = Adam(discriminator.parameters(), lr=train_params['lr'], betas=(0.5, 0.999))
    optimizer_kp_detector = Adam(kp_detector.parameters(), lr=train_params['lr'], betas=(0.5, 0.999))
    schedulers = [MultiStepLR(o, train_params['epoch_milestones'], gamma=0.1, last_epoch=-1)
                  for o in (optimizer_generator, optimizer_discriminator, optimizer_kp_detector)]
    dataloader = DataLoader(Pairs(), batch_size=train_params['batch_size'], shuffle=False, num_workers=0, drop_last=True)
    # train.py:101-105
    generator_full = GeneratorFullModel(kp_detector, generator, discriminator, train_params)
    discriminator_full = DiscriminatorFullModel(kp_detector, generator, discriminator, train_params)
    device_ids = [0] if be.kind == "hip" else None
    generator_full_par = DataParallelWithCallback(generator_full, device_ids=device_ids)
    discriminator_full_par = DataParallelWithCallback(discriminator_full, device_ids=device_ids)
    history = []
    for epoch in range(train_params['num_epochs']):            # train.py:108-141
        for x in dataloader:                                    # host batches: the wrapper moves them to the module's device
            out = generator_full_par(x)
            loss_values = out[:-2]
            generated = out[-2]
            kp_joined = out[-1]
            loss_values = [val.mean() for val in loss_values]
            loss = sum(loss_values)
            loss.backward(retain_graph=not train_params['detach_kp_discriminator'])
            optimizer_generator.step()
            optimizer_generator.zero_grad()
            optimizer_discriminator.zero_grad()
            if train_params['detach_kp_discriminator']:
                optimizer_kp_detector.step()
                optimizer_kp_detector.zero_grad()
            generator_loss_values = [val.detach().cpu().numpy() for val in loss_values]
            loss_values = discriminator_full_par(x, kp_joined, generated)
            loss_values = [val.mean() for val in loss_values]
            loss = sum(loss_values)
            loss.backward()
            optimizer_discriminator.step()
            optimizer_discriminator.zero_grad()
            if not train_params['detach_kp_discriminator']:
                optimizer_kp_detector.step()
                optimizer_kp_detector.zero_grad()
            discriminator_loss_values = [val.detach().cpu().numpy() for val in loss_values]
            history.append([float(v) for v in generator_loss_values] + [float(v) for v in discriminator_loss_values])
        for s in schedulers:
            s.step()
    be.sync()
    for it, (mine, ref, ref64) in enumerate(zip(history, gold["history"], gold["history64"])):
        r32 = ref["generator"] + ref["discriminator"]
        r64 = ref64["generator"] + ref64["discriminator"]
        spread = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(r32, r64))
        err = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(mine, r64))
        assert err <= 16.0 * spread + 2e-5, (it, err, spread)          # the yard-stick of tests/test_step.py
    # Logger.save_cpk (logger.py:43-47)
    models = {'generator': generator, 'discriminator': discriminator, 'kp_detector': kp_detector,
              'optimizer_generator': optimizer_generator, 'optimizer_discriminator': optimizer_discriminator,
              'optimizer_kp_detector': optimizer_kp_detector}
    cpk = {k: v.state_dict() for k, v in models.items()}
    cpk['epoch'], cpk['it'] = 2, 3
    path = os.path.join(tmp_path, '%s-checkpoint.pth.tar' % str(2).zfill(3))
    torch.save(cpk, path)
    # Logger.load_cpk (logger.py:49-66) into fresh objects
    g2, d2, k2 = build(config)
    for m in (g2, d2, k2):
        m.to(be.device)
    og2 = Adam(g2.parameters(), lr=train_params['lr'], betas=(0.5, 0.999))
    checkpoint = torch.load(path, weights_only=False)
    g2.load_state_dict(checkpoint['generator'])
    k2.load_state_dict(checkpoint['kp_detector'])
    d2.load_state_dict(checkpoint['discriminator'])
    og2.load_state_dict(checkpoint['optimizer_generator'])
    assert (checkpoint['epoch'], checkpoint['it']) == (2, 3)
    for a, b in zip(g2.state_dict().items(), generator.state_dict().items()):
        assert a[0] == b[0] and torch.equal(a[1].cpu(), b[1].cpu()), a[0]
    assert og2.state_dict()['state'][0]['step'] == optimizer_generator.state_dict()['state'][0]['step']
    # the restored networks give the trained networks' evaluation forward (reconstruction.py:45-61's wrappers: no device_ids)
    gp, kp_ = DataParallelWithCallback(g2), DataParallelWithCallback(k2)
    gp.eval(), kp_.eval(), generator.eval(), kp_detector.eval()
    with torch.no_grad():
        xs, xd = be.t(src), be.t(drv)
        a = gp(xs, kp_driving=kp_(xd), kp_source=kp_(xs))['video_prediction']
        b = generator(xs, kp_driving=kp_detector(xd), kp_source=kp_detector(xs))['video_prediction']
    be.sync()
    assert torch.equal(a.cpu(), b.cpu())
