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


# ---- round 6: the loop above served by mnk.dropin.TrainPairRunner (hipGraphs on the MI355X, the same three phases as eager
# launches on the CPU emulator) -------------------------------------------------------------------------------------------------
def _reference_loop(be, adam, iterations, gold, batches=None):
    """train.py:81-136 on the drop-in modules for `iterations` batches; returns (history, networks, optimisers, wrappers)"""
    from mnk.engine import GeneratorFullModel, DiscriminatorFullModel
    from mnk.optim import MnkAdam
    from sync_batchnorm import DataParallelWithCallback
    Adam = torch.optim.Adam if adam == "torch" else MnkAdam
    config = copy.deepcopy(gold["cfg"])
    tp = config["train_params"]
    generator, discriminator, kp_detector = build(config)
    generator.load_state_dict(gold["state"]["generator"])
    discriminator.load_state_dict(gold["state"]["discriminator"])
    kp_detector.load_state_dict(gold["state"]["kp_detector"])
    for m in (generator, discriminator, kp_detector):
        m.to(be.device)
    og = Adam(generator.parameters(), lr=tp['lr'], betas=(0.5, 0.999))
    od = Adam(discriminator.parameters(), lr=tp['lr'], betas=(0.5, 0.999))
    ok = Adam(kp_detector.parameters(), lr=tp['lr'], betas=(0.5, 0.999))
    gpar = DataParallelWithCallback(GeneratorFullModel(kp_detector, generator, discriminator, tp), device_ids=[0] if be.kind == "hip" else None)
    dpar = DataParallelWithCallback(DiscriminatorFullModel(kp_detector, generator, discriminator, tp), device_ids=[0] if be.kind == "hip" else None)
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])
    history = []
    for it in range(iterations):
        x = {"source": src.clone(), "video": drv.clone()} if batches is None else batches[it]
        out = gpar(x)
        loss_values = [val.mean() for val in out[:-2]]
        generated, kp_joined = out[-2], out[-1]
        sum(loss_values).backward(retain_graph=not tp['detach_kp_discriminator'])
        og.step(), og.zero_grad(), od.zero_grad()
        if tp['detach_kp_discriminator']:
            ok.step(), ok.zero_grad()
        g_host = [float(val.detach().cpu()) for val in loss_values]
        loss_values = [val.mean() for val in dpar(x, kp_joined, generated)]
        sum(loss_values).backward()
        od.step(), od.zero_grad()
        if not tp['detach_kp_discriminator']:
            ok.step(), ok.zero_grad()
        history.append(g_host + [float(val.detach().cpu()) for val in loss_values])
    be.sync()
    return history, (generator, discriminator, kp_detector), (og, od, ok), (gpar, dpar)


@pytest.mark.parametrize("adam", ["torch", "mnk"])
def test_the_loop_is_served_by_the_dropin_runner_and_equals_the_modules_run_as_they_are(be, monkeypatch, adam):
    """The same loop twice: through mnk.dropin.TrainPairRunner (default: three captured hipGraphs on the MI355X, the same three
    phases as eager launches on the emulator; stock optimisers stepped by mnk.optim.AdoptedAdam) and with MNK_DROPIN_GRAPH=0 (the
    wrapped modules called as they are, two discriminator passes, the stock optimiser steps).  Same loss history and parameters
    to fp32 summation-order level; every call of the loop was served (no fall-back); the stock optimisers' state_dict()s agree."""
    from mnk import dropin, optim as moptim
    gold = load("step_tiny")
    monkeypatch.setenv("MNK_DROPIN_GRAPH", "0")
    h0, nets0, opts0, _ = _reference_loop(be, adam, 3, gold)
    monkeypatch.delenv("MNK_DROPIN_GRAPH")
    h1, nets1, opts1, (gpar, dpar) = _reference_loop(be, adam, 3, gold)
    runner = dropin.runner_for(gpar.module)
    assert runner is not None and runner is dropin.runner_for(dpar.module)
    served = runner.stats["graph_calls"] if be.kind == "hip" else runner.stats["phase_calls"]
    assert served == 3 and runner.stats["fallbacks"] == 0 and runner.stats["d_fallbacks"] == 0, runner.stats
    if be.kind == "hip":
        assert runner.stats["captures"] == 1
    for it, (a, b) in enumerate(zip(h1, h0)):
        for va, vb in zip(a, b):
            assert abs(va - vb) <= 2e-4 * max(1.0, abs(vb)), (it, a, b)
    # Adam's first updates are sign-like: an element whose gradient is ~eps (the bias in front of a training-mode BatchNorm: an
    # analytically zero gradient, i.e. rounding noise) moves by lr either way in every step.  Per network: the mean over ALL its
    # elements stays a fraction of lr, and no element is further apart than the 2 x 3 steps allow
    lr = gold["cfg"]["train_params"]["lr"]
    for m1, m0 in zip(nets1, nets0):
        tot = cnt = 0.0
        for (n1, p1), (_, p0) in zip(m1.named_parameters(), m0.named_parameters()):
            d = (p1.detach().cpu() - p0.detach().cpu()).abs()
            tot, cnt = tot + float(d.sum()), cnt + d.numel()
            assert float(d.max()) <= 6.5 * lr, (n1, float(d.max()))
        assert tot / cnt <= 0.25 * lr, (type(m1).__name__, tot / cnt)
    if adam == "torch":
        for o1, o0 in zip(opts1, opts0):
            ad = moptim.adopted(o1)
            assert ad is not None and ad.steps_taken == 3 and moptim.adopted(o0) is None
            s1, s0 = o1.state_dict(), o0.state_dict()
            assert s1["param_groups"] == s0["param_groups"] and set(s1["state"]) == set(s0["state"])
            for k in s1["state"]:
                assert float(s1["state"][k]["step"]) == float(s0["state"][k]["step"]) == 3.0
                assert s1["state"][k]["exp_avg"].shape == s0["state"][k]["exp_avg"].shape


def test_the_dropin_runner_falls_back_on_what_it_does_not_serve(be, monkeypatch):
    """gradients that were not zeroed, evaluation mode, a discriminator call on tensors that are not this iteration's outputs: the
    wrapped modules run as they are (same values as MNK_DROPIN_GRAPH=0)"""
    from mnk import dropin
    gold = load("step_tiny")
    _, nets, opts, (gpar, dpar) = _reference_loop(be, "torch", 1, gold)
    runner = dropin.runner_for(gpar.module)
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])
    x = {"source": src, "video": drv}
    before = dict(runner.stats)
    # 1. a stale gradient on one parameter
    p = next(nets[0].parameters())
    p.grad = torch.zeros_like(p)
    out = gpar(x)
    assert runner.stats["fallbacks"] == before["fallbacks"] + 1 and out[0].grad_fn is not None
    p.grad = None
    # 2. the discriminator pass on other tensors than this iteration's
    out = gpar(x)
    other = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out[-2].items()}
    d = dpar(x, out[-1], other)
    assert runner.stats["d_fallbacks"] == before["d_fallbacks"] + 1 and d[0].grad_fn is not None
    d2 = dpar(x, out[-1], out[-2])
    assert runner.stats["d_fallbacks"] == before["d_fallbacks"] + 1
    be.sync()
    assert abs(float(d[0].mean()) - float(d2[0].mean())) <= 1e-5 * max(1.0, abs(float(d2[0].mean())))
    # 3. evaluation mode: the module itself
    for m in nets:
        m.eval()
    gpar.eval()
    n = runner.stats["fallbacks"]
    with torch.no_grad():
        gpar(x)
    assert runner.stats["fallbacks"] in (n, n + 1)      # (an eval wrapper does not even ask the runner)
    # 4. a backward pass through an earlier call's outputs is refused, not served wrongly
    for m in nets:
        m.train()
    gpar.train()
    for o in opts:
        o.zero_grad()
    first = gpar(x)
    gpar(x)
    with pytest.raises(RuntimeError, match="earlier"):
        sum(v.mean() for v in first[:-2]).backward()


def test_adopted_adam_follows_schedulers_checkpoints_and_batch_size_changes(be, monkeypatch, tmp_path):
    """The drop-in runner + AdoptedAdam under what train.py does around the loop: a MultiStepLR milestone between two iterations
    (train.py:88-96,143-145), Logger.save_cpk / load_cpk of the three STOCK optimisers (logger.py:43-66) into a fresh set of
    objects that then continues, and a last batch of another size (re-capture).  Reference run: the same statements with
    MNK_DROPIN_GRAPH=0 MNK_ADOPT_ADAM=0 (stock steps on the modules as they are)."""
    from mnk import dropin, optim as moptim
    gold = load("step_tiny")
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])

    def run(tag):
        config = copy.deepcopy(gold["cfg"])
        tp = config["train_params"]
        from mnk.engine import GeneratorFullModel, DiscriminatorFullModel
        from sync_batchnorm import DataParallelWithCallback

        def fresh():
            nets = build(config)
            for m, k in zip(nets, ("generator", "discriminator", "kp_detector")):
                m.load_state_dict(gold["state"][k])
                m.to(be.device)
            opts = [torch.optim.Adam(m.parameters(), lr=tp["lr"], betas=(0.5, 0.999)) for m in nets]
            scheds = [MultiStepLR(o, [1], gamma=0.1, last_epoch=-1) for o in opts]
            g, d, k = nets
            ids = [0] if be.kind == "hip" else None
            return nets, opts, scheds, (DataParallelWithCallback(GeneratorFullModel(k, g, d, tp), device_ids=ids),
                                        DataParallelWithCallback(DiscriminatorFullModel(k, g, d, tp), device_ids=ids))

        def iteration(nets, opts, pars, x):
            og, od, ok = opts
            out = pars[0](x)
            vals = [v.mean() for v in out[:-2]]
            sum(vals).backward(retain_graph=not tp["detach_kp_discriminator"])
            og.step(), og.zero_grad(), od.zero_grad()
            if tp["detach_kp_discriminator"]:
                ok.step(), ok.zero_grad()
            dv = [v.mean() for v in pars[1](x, out[-1], out[-2])]
            sum(dv).backward()
            od.step(), od.zero_grad()
            if not tp["detach_kp_discriminator"]:
                ok.step(), ok.zero_grad()
            return [float(v.detach().cpu()) for v in vals + dv]

        nets, opts, scheds, pars = fresh()
        x = {"source": src, "video": drv}
        hist = [iteration(nets, opts, pars, x)]
        for s in scheds:                     # epoch boundary: lr 2e-4 -> 2e-5
            s.step()
        hist.append(iteration(nets, opts, pars, x))
        cpk = {"g": nets[0].state_dict(), "d": nets[1].state_dict(), "k": nets[2].state_dict(), "o": [o.state_dict() for o in opts]}
        path = os.path.join(tmp_path, tag + ".pth.tar")
        torch.save(cpk, path)
        nets2, opts2, _, pars2 = fresh()     # Logger.load_cpk into new objects
        ck = torch.load(path, weights_only=False)
        nets2[0].load_state_dict(ck["g"]), nets2[1].load_state_dict(ck["d"]), nets2[2].load_state_dict(ck["k"])
        for o, sd in zip(opts2, ck["o"]):
            o.load_state_dict(sd)
        hist.append(iteration(nets2, opts2, pars2, x))
        half = {"source": src[:1].repeat(3, 1, 1, 1, 1), "video": drv[:1].repeat(3, 1, 1, 1, 1)}     # a last batch of another size
        hist.append(iteration(nets2, opts2, pars2, half))
        hist.append(iteration(nets2, opts2, pars2, x))
        be.sync()
        steps = [float(opts2[0].state_dict()["state"][0]["step"]), opts2[0].param_groups[0]["lr"]]
        return hist, steps, pars2, opts2

    monkeypatch.setenv("MNK_DROPIN_GRAPH", "0")
    monkeypatch.setenv("MNK_ADOPT_ADAM", "0")
    want, wsteps, _, _ = run("stock")
    monkeypatch.delenv("MNK_DROPIN_GRAPH")
    monkeypatch.delenv("MNK_ADOPT_ADAM")
    got, gsteps, pars, opts = run("runner")
    runner = dropin.runner_for(pars[0].module)
    assert runner.stats["fallbacks"] == 0 and runner.stats["d_fallbacks"] == 0
    assert (runner.stats["graph_calls"] if be.kind == "hip" else runner.stats["phase_calls"]) == 3
    if be.kind == "hip":
        assert runner.stats["captures"] == 2               # the golden's batch and the batch of three
    assert all(moptim.adopted(o) is not None and moptim.adopted(o).steps_taken == 3 for o in opts)
    assert gsteps == wsteps == [5.0, gsteps[1]]
    for it, (a, b) in enumerate(zip(got, want)):
        for va, vb in zip(a, b):
            # two fp32 trajectories under Adam's sign-like first updates separate step by step (tests/test_step.py): 5e-4 relative
            # at the fifth iteration measured; the structure assertions above are the point of this test
            assert abs(va - vb) <= 2e-4 * (it + 1) ** 2 * max(1.0, abs(vb)), (it, a, b)


def test_adopted_adam_equals_the_stock_step(be):
    """mnk.optim.AdoptedAdam (mnk_adam_multi on the stock optimiser's own state tensors) against torch.optim.Adam itself: same
    parameters, exp_avg, exp_avg_sq and step after three steps on the same gradients (2e-6 relative, the bound of MnkAdam's test)"""
    from mnk import optim as moptim
    g = torch.Generator().manual_seed(2)
    shapes = [(7,), (5, 3), (33, 17, 1, 3, 3), (1024,), (4099,)]

    def params():
        g2 = torch.Generator().manual_seed(9)
        return [torch.nn.Parameter(be.t(torch.randn(s, generator=g2))) for s in shapes]

    pa, pb = params(), params()
    oa = torch.optim.Adam(pa, lr=3e-3, betas=(0.5, 0.999))
    ob = torch.optim.Adam(pb, lr=3e-3, betas=(0.5, 0.999))
    sinks = moptim.GradSinks(pb)
    moptim.install_adam_adoption()
    for it in range(3):
        grads = [torch.randn(s, generator=g) * (10.0 ** (it - 1)) for s in shapes]
        for p, gr in zip(pa, grads):
            p.grad = be.t(gr)
        for p, gr in zip(pb, grads):
            sinks.sink(p).copy_(be.t(gr))
            p.grad = sinks.sink(p)
        oa.step(), ob.step()
        assert all(p.grad is not None for p in pb)         # handed back after the (empty) stock step
        oa.zero_grad(), ob.zero_grad()
    be.sync()
    assert moptim.adopted(ob).steps_taken == 3 and moptim.adopted(oa) is None
    for a, b in zip(pa, pb):
        assert float((a.detach().cpu() - b.detach().cpu()).abs().max()) <= 2e-6 * max(1.0, float(a.detach().abs().max()))
        for k, tol in (("exp_avg", 1e-5), ("exp_avg_sq", 1e-5)):
            x, y = oa.state[a][k].cpu(), ob.state[b][k].cpu()
            assert float((x - y).abs().max()) <= tol * max(1e-30, float(x.abs().max())), k
        assert float(oa.state[a]["step"]) == float(ob.state[b]["step"]) == 3.0
