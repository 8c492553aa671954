#!/usr/bin/env python
"""Times the warps of one generator pass (ops.WarpAllFn: mnk_warp_levels_fwd / _bwd, the deterministic gather-form backward)
on the MI355X for the level set of a configuration, optionally over values of a tuning knob.

    python tools/warp_bench.py [--config moving-gif] [--batch 32] [--size 64] [--knob warp_gather_tile=0,4,8,16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "monkey-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="moving-gif")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--knob", default="")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--no-emb", action="store_true")
    ap.add_argument("--no-field-grad", action="store_true")
    ap.add_argument("--no-input-grad", action="store_true")
    a = ap.parse_args()
    from mnk import configs, ops, _lib
    cfg = configs.get(a.config)
    gp = cfg["model_params"]["generator_params"]
    cp = cfg["model_params"]["common_params"]
    mode = {"nearest": 0, "trilinear": 1}[gp.get("interpolation_mode", "nearest")]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    nc, be, mf, nb = cp["num_channels"], gp["block_expansion"], gp["max_features"], gp["num_blocks"]
    chans = [nc] + [min(mf, be * 2 ** (i + 1)) for i in range(nb)]       # Encoder: the frame, then one skip per DownBlock3D
    ke = 0
    if gp.get("kp_embedding_params") is not None and not a.no_emb:
        ke = cp["num_kp"] + 1          # heat-map channels of the embedding (use_heatmap) -- close enough for a timing
    sf = (gp.get("dense_motion_params") or {}).get("scale_factor", 1)
    hf = int(a.size * sf)
    shapes = [(c, a.size >> i, a.size >> i) for i, c in enumerate(chans)] + [(nc, a.size, a.size)]
    specs = tuple((c, ke) for c, _, _ in shapes[:-1]) + ((nc, 0),)
    n = a.batch
    ident = torch.stack(torch.meshgrid(torch.linspace(-1, 1, hf), torch.linspace(-1, 1, hf), indexing="ij")[::-1], -1)
    field = (ident.view(1, hf, hf, 2) + 0.1 * torch.randn(n, hf, hf, 2, generator=g)).to(dev)
    emb = torch.randn(n, hf, hf, ops.ceil4(ke), generator=g).to(dev) if ke else None
    inps = []
    for c, h, w in shapes:
        t = torch.zeros(n, h, w, ops.ceil4(c))
        t[..., :c] = torch.randn(n, h, w, c, generator=g)
        inps.append(t.to(dev))
    douts = [torch.randn(n, h, w, ops.ceil4(c + k), generator=g).to(dev) for (c, h, w), (_, k) in zip(shapes, specs)]
    print("levels (C, h, w):", shapes, "mode", mode, "field", hf, "emb channels", ke)

    def run():
        f = field.clone().requires_grad_(not a.no_field_grad)
        e = emb.clone().requires_grad_(True) if emb is not None else None
        xs = [t.clone().requires_grad_(0 < i < len(inps) - 1 and not a.no_input_grad) for i, t in enumerate(inps)]      # (the frame itself needs no gradient)
        outs = ops.WarpAllFn.apply(f, e, mode, specs, *xs)
        torch.autograd.backward(list(outs), douts)

    lib = _lib.lib()
    lib.cdll.mnk_prof_reset()
    name, values = (a.knob.split("=") + [""])[:2] if a.knob else ("", "")
    for v in ([int(x) for x in values.split(",")] if values else [None]):
        if v is not None:
            lib.call("mnk_set_tuning", name.encode(), v)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        import ctypes
        lib.cdll.mnk_prof_reset()
        lib.cdll.mnk_prof_enable(1)
        for _ in range(a.iters):
            run()
        torch.cuda.synchronize()
        lib.cdll.mnk_prof_enable(0)
        for k in range(lib.cdll.mnk_prof_num_kernels()):
            cnt, ms, work = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
            lib.cdll.mnk_prof_query(k, ctypes.byref(cnt), ctypes.byref(ms), ctypes.byref(work))
            if cnt.value and lib.cdll.mnk_prof_kernel_name(k).decode() == "deform":
                print("%s=%s: deform group %.1f us per pass (fwd + bwd, %d launches)" % (name or "default", v, ms.value / a.iters * 1e3,
                                                                                    cnt.value // a.iters))


if __name__ == "__main__":
    main()
