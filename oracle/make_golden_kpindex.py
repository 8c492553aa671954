"""The integer key-point records of the REAL reference, for every golden case that holds key points:

    python -m oracle.make_golden_kpindex            -> tests/golden/kp_index.pt

north_star's only bit-exact criterion is "bit-exact keypoint indices".  The reference forms integers from key points in one
place: the Visualizer's pixel position `spatial_size * (mean + 1) / 2` (logger.py:99-100), rasterised by
skimage.draw.circle (:104); SURVEY.md section 8c adds the arg-max of the soft-max heat-map (keypoint_detector.py:103-104).
This script runs the unmodified reference KPDetector (oracle/ref_shim.py) in fp32 and fp64 on the inputs of every
golden case (same seeds, weights and frames as oracle/make_golden.py / make_golden_full.py) with a forward hook on its
hourglass, and records per case

    mean32 / mean64   (B,D,K,2)   the soft-argmax means;
    argmax32 / 64     (B,D,K)     h * W + w of the largest heat-map logit (first occurrence);
    top2_gap64        (B,D,K)     largest minus second-largest logit in fp64 -- a key point whose two best pixels are closer
                                  than any fp32 implementation's error has no well-defined arg-max: tests list those;
    frame             (W, H)      the frame size the Visualizer would draw on.

TEST INFRASTRUCTURE ONLY (authoring container; the GPU box has no /root/reference)."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim, cases  # noqa: E402
from oracle.make_golden import build_reference, load_cfg, save  # noqa: E402


ALT_FRAME = (96, 80)     # a frame size that is not a power of two: the product size * (mean + 1) is not exact in fp32 there


def pixel_index(mean, frame):
    """logger.py:99-100 `spatial_size * (kp_array + 1) / 2` in numpy's arithmetic: `kp_array + 1` stays in the key points'
    dtype (an array plus a Python int), the int64 `spatial_size` array times a float32 array is promoted to float64, and so
    is the division; then floor.  spatial_size = (W, H).  A pure function of the recorded means: `--patch` recomputes the
    integer records of an existing tests/golden/kp_index.pt without running the reference again."""
    size = torch.tensor([frame[0], frame[1]], dtype=torch.float64)
    t = mean + 1
    return torch.floor(size * t.double() / 2).to(torch.int64)


def patch():
    """re-derive pixel32 / pixel64 (numpy promotion) and add the ALT_FRAME records to the existing golden"""
    path = os.path.join(ROOT, "tests", "golden", "kp_index.pt")
    gold = torch.load(path, weights_only=False)
    changed = 0
    for name, g in gold.items():
        if name.startswith("_"):
            continue
        for tag in ("32", "64"):
            new = pixel_index(g["mean" + tag], g["frame"])
            changed += int((new != g["pixel" + tag]).sum())
            g["pixel" + tag] = new
            g["pixel%s_alt" % tag] = pixel_index(g["mean" + tag], ALT_FRAME)
        g["frame_alt"] = ALT_FRAME
    print("patched %s: %d integer(s) changed by the float64 promotion; alt frame %s added" % (path, changed, ALT_FRAME))
    save("kp_index", gold)


def run_kp(kpd, frames, dtype, train):
    """-> mean (B,D,K,2), argmax (B,D,K) int64, top-2 gap (B,D,K) of the heat-map logits."""
    kpd.to(dtype).train(train)
    grabbed = {}
    h = kpd.predictor.register_forward_hook(lambda m, i, o: grabbed.__setitem__("heat", o.detach()))
    with torch.no_grad():
        kp = kpd(frames.to(dtype))
    h.remove()
    heat = grabbed["heat"]                                   # (B, K, D, h, w)
    b, k, d, hh, ww = heat.shape
    flat = heat.permute(0, 2, 1, 3, 4).reshape(b, d, k, hh * ww)
    top = flat.topk(2, dim=-1)
    # torch.argmax's tie rule is not documented; the first occurrence of the maximum is what the test compares
    first = (flat == top.values[..., :1]).to(torch.int64).argmax(dim=-1)
    return kp["mean"].detach(), first, (top.values[..., 0] - top.values[..., 1]).double(), (hh, ww)


def case(ref, cfg, frames, train, frame_size):
    out = {}
    for tag, dtype in (("32", torch.float32), ("64", torch.float64)):
        _, _, kpd, _ = build_reference(ref, cfg)             # fresh weights per precision (train mode moves running stats)
        mean, am, gap, heat_hw = run_kp(kpd, frames, dtype, train)
        out["mean" + tag] = mean
        out["argmax" + tag] = am
        out["pixel" + tag] = pixel_index(mean, frame_size)
        out["pixel%s_alt" % tag] = pixel_index(mean, ALT_FRAME)
        if tag == "64":
            out["top2_gap64"] = gap
    out["frame"] = tuple(frame_size)
    out["frame_alt"] = ALT_FRAME
    out["heat_hw"] = heat_hw
    return out


def main():
    assert ref_shim.available(), "run this in the authoring container (needs /root/reference)"
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    ref = ref_shim.load()
    gold = {}
    # module cases of oracle/make_golden.py: smooth frames, [source | driving] joined on the time axis, train and eval
    mod = (("tiny", cases.TINY, 2, 32), ("tiny2", cases.TINY2, 3, 16), ("shapes", None, 2, 64), ("taichi", None, 2, 64),
           ("moving-gif", None, 2, 64), ("bair", None, 2, 64), ("vox", None, 2, 128), ("vox256", "vox", 2, 256))
    for name, cfg, batch, size in mod:
        cfg = copy.deepcopy(cfg) if isinstance(cfg, dict) else load_cfg(cfg or name)
        src, drv = cases.smooth_pair(batch, size, size)
        frames = torch.cat([src, drv], dim=2)
        for mode in ("train", "eval"):
            gold["%s/%s" % (name, mode)] = case(ref, cfg, frames, mode == "train", (size, size))
        print(name, "done", flush=True)
    # full-iteration cases of oracle/make_golden_full.py: U[0,1) frames, training mode
    for name, cfg_name, batch, size in (("fullstep_moving-gif_b32", "moving-gif", 32, 64), ("fullstep_taichi_b32", "taichi", 32, 64),
                                        ("fullstep_tiny_b4", "tiny", 4, 32)):
        cfg = copy.deepcopy(cases.TINY) if cfg_name == "tiny" else load_cfg(cfg_name)
        src, drv = cases.synthetic_pair(batch, size, size)
        gold[name + "/train"] = case(ref, cfg, torch.cat([src, drv], dim=2), True, (size, size))
        print(name, "done", flush=True)
    # batched inference (BASELINE configs[4]): the driving frames' key points, one frame per call (reconstruction.py:57-59)
    src, drv = cases.synthetic_pair(512, 64, 64, seed=4321)
    gold["infer_bair_b512/eval"] = case(ref, load_cfg("bair"), drv, False, (64, 64))
    # how well defined the integers are in the reference itself
    lines = []
    for name, g in gold.items():
        n = g["pixel64"].numel()
        lines.append("%-34s key points %5d  pixel32 != pixel64: %d  argmax32 != argmax64: %d  smallest top-2 logit gap %.3e" % (
            name, n // 2, int((g["pixel32"] != g["pixel64"]).sum()), int((g["argmax32"] != g["argmax64"]).sum()),
            float(g["top2_gap64"].min())))
    print("\n".join(lines))
    gold["_report"] = lines
    save("kp_index", gold)


if __name__ == "__main__":
    patch() if "--patch" in sys.argv else main()
