"""The benchmark workload of the hot path, product side: the synthetic input protocol of BASELINE.md section 2 and the
algorithmic FLOP accounting of the convolution stack that bench.py / the tools quote.  Pure
arithmetic on the config; nothing here runs on the GPU.  (tests/test_oracle_golden.py checks this count against the
oracle's own accounting.)"""
import torch


def synthetic_pair(batch, height, width, seed=1234, channels=3):
    """BASELINE.md section 2 protocol: source, video ~ U[0,1) float32 (B,3,1,H,W) from a seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    source = torch.rand(batch, channels, 1, height, width, generator=g)
    video = torch.rand(batch, channels, 1, height, width, generator=g)
    return source, video


def conv_flops_hot_path(cfg, height, width, frames_kp=2):
    """Forward conv FLOPs (2*MAC) of KPDetector (on `frames_kp` frames) + generator (1 frame) for ONE
    training pair, derived from the channel ladders of modules/util.py:142-143,169-171.
    Returns dict(kp=, gen=, total=, layers=[(name, cin, cout, h, w, k, flops)])."""
    mp = cfg["model_params"]
    common = mp["common_params"]
    layers = []

    def add(name, cin, cout, h, w, k=3, frames=1, groups=1):
        layers.append((name, cin, cout, h, w, k, 2 * (cin // groups) * cout * k * k * h * w * frames))

    def hourglass_layers(name, be, cin, cout, nb, mx, h, w, frames, last=True, extra=0, enc=True, cin_dec=None):
        chans = [cin]
        hh, ww = h, w
        if enc:
            for i in range(nb):
                co = min(mx, be * 2 ** (i + 1))
                add("%s.enc%d" % (name, i), chans[-1], co, hh, ww, frames=frames)
                chans.append(co)
                hh, ww = hh // 2, ww // 2
        else:
            for i in range(nb):
                chans.append(min(mx, be * 2 ** (i + 1)))
                hh, ww = hh // 2, ww // 2
        for j, i in enumerate(range(nb)[::-1]):
            ci = (1 if i == nb - 1 else 2) * min(mx, be * 2 ** (i + 1)) + extra
            co = min(mx, be * 2 ** i)
            hh, ww = hh * 2, ww * 2
            add("%s.dec%d" % (name, j), ci, co, hh, ww, frames=frames)
        if last:
            add(name + ".last", be + cin + extra, cout, h, w, frames=frames)

    kpp = mp["kp_detector_params"]
    s = kpp.get("scale_factor", 1)
    hourglass_layers("kp", kpp["block_expansion"], common["num_channels"], common["num_kp"], kpp["num_blocks"],
                     kpp["max_features"], int(height * s), int(width * s), frames_kp)
    n_kp = len(layers)
    gp = mp["generator_params"]
    be, mx, nb = gp["block_expansion"], gp["max_features"], gp["num_blocks"]
    cin = common["num_channels"]
    hh, ww, c = height, width, cin
    for i in range(nb):
        co = min(mx, be * 2 ** (i + 1))
        add("gen.app%d" % i, c, co, hh, ww)
        c, hh, ww = co, hh // 2, ww // 2
    dm = gp.get("dense_motion_params")
    K = common["num_kp"]
    if dm is not None:
        s = dm.get("scale_factor", 1)
        me = dm["mask_embedding_params"]
        per = int(me.get("use_heatmap", True)) + 2 * int(me.get("use_difference", False)) + \
            cin * int(me.get("use_deformed_source_image", False))
        cemb = per * (K + 1)
        for i in range(dm.get("num_group_blocks", 0)):
            add("gen.dm.group%d" % i, cemb, cemb, int(height * s), int(width * s), k=1, groups=K + 1)
        hourglass_layers("gen.dm", dm["block_expansion"], cemb, (K + 1) * dm["use_mask"] + 2 * dm["use_correction"],
                         dm["num_blocks"], dm["max_features"], int(height * s), int(width * s), 1)
    kpe = gp.get("kp_embedding_params")
    extra = 0
    if kpe is not None:
        extra = (int(kpe.get("use_heatmap", True)) + 2 * int(kpe.get("use_difference", False)) +
                 cin * int(kpe.get("use_deformed_source_image", False))) * (K + int(kpe.get("add_bg_feature_map", False)))
    hourglass_layers("gen.dec", be, cin, cin, nb, mx, height, width, 1, last=False, extra=extra, enc=False)
    cref = be + cin + extra
    for i in range(gp["num_refinement_blocks"]):
        add("gen.ref%d.conv1" % i, cref, cref, height, width)
        add("gen.ref%d.conv2" % i, cref, cref, height, width)
    add("gen.conv-last", cref, cin, height, width, k=1)
    kp = sum(l[-1] for l in layers[:n_kp])
    gen = sum(l[-1] for l in layers[n_kp:])
    return {"kp": kp, "gen": gen, "total": kp + gen, "layers": layers}
