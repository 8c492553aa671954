#!/usr/bin/env python
"""Where the per-frame evaluation loop's time goes (reconstruction.py:45-62 on the drop-in modules, batch 1): the whole loop as
bench.py's `eval_frame_loop` times it, against back-to-back replays of the two frozen-weight hipGraphs alone (device-bound part)
and the host-side statements of one wrapper call with the replay taken out.  Run on the MI355X:
    python tools/eval_loop_breakdown.py [config] [size]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from mnk import configs, dropin  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "moving-gif"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    frames = 40
    cfg = configs.get(name)
    device = torch.device("cuda:0")
    from sync_batchnorm import DataParallelWithCallback
    gen, _, kpd = bench.build_models(cfg, device)
    generator, kp_detector = DataParallelWithCallback(gen), DataParallelWithCallback(kpd)
    generator.eval(), kp_detector.eval()
    video = torch.rand(1, 3, frames, size, size)

    def loop(v):
        with torch.no_grad():
            kp_source = kp_detector(v[:, :, :1])
            for i in range(frames):
                kp_driving = kp_detector(v[:, :, i:i + 1])
                out = generator(source_image=v[:, :, :1], kp_driving=kp_driving, kp_source=kp_source)
        return out["video_prediction"]

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    loop(video)
    print("whole loop, host frames     : %.3f ms per frame" % (timed(lambda: loop(video)) / frames * 1e3))
    dvideo = video.to(device)
    loop(dvideo)
    print("whole loop, device frames   : %.3f ms per frame" % (timed(lambda: loop(dvideo)) / frames * 1e3))
    runners = [dropin.eval_runner_for_wrapper(w) for w in (kp_detector, generator)]
    if os.environ.get("MNK_BREAKDOWN_REPLAYS"):      # under rocprofv3: N replays of one graph, nothing else (kernel counts / N)
        which, n = os.environ["MNK_BREAKDOWN_REPLAYS"].split(":")
        r = runners[0 if which == "kp" else 1]
        for prog in r.programs.values():
            for _ in range(int(n)):
                prog["graph"].replay()
        torch.cuda.synchronize()
        return
    total = 0.0
    for label, r in zip(("kp_detector", "generator"), runners):
        for key, prog in r.programs.items():
            g = prog["graph"]
            dt = timed(lambda: [g.replay() for _ in range(200)]) / 200
            total += dt
            print("  %-12s graph replay alone: %.3f ms" % (label, dt * 1e3))
    print("both graphs back to back    : %.3f ms per frame (device-bound part)" % (total * 1e3))

    # the host statements of the loop with the replays taken out
    class _NoReplay:
        def replay(self):
            pass
    saved = []
    for r in runners:
        for prog in r.programs.values():
            saved.append((prog, prog["graph"]))
            prog["graph"] = _NoReplay()
    print("loop without the replays    : %.3f ms per frame (host frames)" % (timed(lambda: loop(video)) / frames * 1e3))
    print("loop without the replays    : %.3f ms per frame (device frames)" % (timed(lambda: loop(dvideo)) / frames * 1e3))
    for prog, g in saved:
        prog["graph"] = g
    with torch.no_grad():
        os.environ["MNK_EVAL_GRAPH"] = "0"
        try:
            from mnk import knobs
            loop(video)
            print("eager launches (MNK_EVAL_GRAPH=0): %.3f ms per frame" % (timed(lambda: loop(video)) / frames * 1e3))
        finally:
            os.environ.pop("MNK_EVAL_GRAPH")


if __name__ == "__main__":
    main()
