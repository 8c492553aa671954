#!/usr/bin/env python
"""The reference's per-frame evaluation loop on the drop-in modules (reconstruction.py:45-62: for every frame of a video,
kp_detector(frame) and generator(source, kp_driving, kp_source) at batch 1, no_grad, eval mode, behind
DataParallelWithCallback) -- wall time per frame with eager launches, and where the host time goes (cProfile).
Usage (GPU box): python tools/frame_loop_probe.py [--config taichi] [--frames 60]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from mnk import configs  # noqa: E402
from sync_batchnorm import DataParallelWithCallback  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="taichi")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--frames", type=int, default=60)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen, disc, kpd = bench.build_models(configs.get(a.config), dev)
    generator, kp_detector = DataParallelWithCallback(gen), DataParallelWithCallback(kpd)       # reconstruction.py:45-46
    generator.eval(), kp_detector.eval()
    video = torch.rand(1, 3, a.frames, a.size, a.size)

    def loop():                                   # reconstruction.py:52-62
        out = []
        with torch.no_grad():
            kp_source = kp_detector(video[:, :, :1])
            for i in range(a.frames):
                d = video[:, :, i:i + 1]
                kp_driving = kp_detector(d)
                o = generator(source_image=video[:, :, :1], kp_driving=kp_driving, kp_source=kp_source)
                out.append(o["video_prediction"])
        return out

    loop()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.frames
    print("%s @ %d: %.3f ms per frame (%.0f frames/s), eager launches, batch 1" % (a.config, a.size, dt * 1e3, 1.0 / dt))
    pr = cProfile.Profile()
    pr.enable()
    loop()
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(35)
        print("==== by %s\n%s" % (key, s.getvalue()[:7000]))


if __name__ == "__main__":
    main()
