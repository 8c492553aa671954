"""Worker of tests/test_dist_graph_gpu.py: one rank of a (forced) process group on the MI355X.  The captured iteration of
several ranks -- linear hipGraphs with the gradient exchange started / awaited by host calls between them and the
generator-side optimiser steps replayed from their descriptor tables (mnk.engine.TrainStep._cut) -- must give the losses and
the parameters of the eager iteration on the same batches, with the same learning-rate change in the middle."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "monkey-net_amd"))


def main():
    import torch.distributed as tdist
    from mnk import configs, engine, workload, dist as mdist
    from modules.generator import MotionTransferGenerator
    from modules.discriminator import Discriminator
    from modules.keypoint_detector import KPDetector
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    tdist.init_process_group("nccl")
    assert mdist.grads_active(), "MNK_DIST_FORCE=1 must switch the collective paths on"
    cfg = configs.get("moving-gif")
    mp = cfg["model_params"]
    blur = torch.nn.AvgPool2d(9, stride=1, padding=4)
    batches = []
    for i in range(3):
        src, drv = workload.synthetic_pair(8, 64, 64, seed=200 + i)
        batches.append({"source": blur(src[:, :, 0]).unsqueeze(2).contiguous().cuda(),
                        "video": blur(drv[:, :, 0]).unsqueeze(2).contiguous().cuda()})

    def build():
        torch.manual_seed(5)
        gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"]).cuda()
        disc = Discriminator(**mp["discriminator_params"], **mp["common_params"]).cuda()
        kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"]).cuda()
        return gen, disc, kpd

    def flat(mods):
        return [torch.cat([p.detach().flatten() for p in m.parameters()]).double() for m in mods]

    start = flat(build())

    def run(use_graph):
        mods = build()
        step = engine.TrainStep(*([mods[0], mods[1], mods[2]]), cfg["train_params"], use_graph=use_graph)
        hist = []
        for it, i in enumerate((0, 1, 2, 0, 1)):
            if it == 3:
                for o in (step.opt_g, step.opt_d, step.opt_k):
                    o.param_groups[0]["lr"] *= 0.5
            g_l, d_l, _ = step.step(batches[i])
            hist.append([float(v) for v in g_l] + [float(v) for v in d_l])
        torch.cuda.synchronize()
        return hist, flat(mods), step

    h_e, p_e, _ = run(False)
    h_g, p_g, step = run(True)
    pieces = step._graph
    ngraphs = sum(isinstance(x, torch.cuda.CUDAGraph) for x in pieces)
    print("pieces: %d hipGraphs, %d host calls" % (ngraphs, len(pieces) - ngraphs))
    assert (ngraphs, len(pieces) - ngraphs) == (2, 2), pieces
    # every optimiser ticked once per replayed iteration -- also the two whose step is a host call of the replay
    counts = [float(o.hyper[7]) for o in (step.opt_g, step.opt_d, step.opt_k)]
    assert counts == [5.0, 5.0, 5.0], counts

    def dev(a, b):
        return max(abs(u - v) / max(1.0, abs(v)) for u, v in zip(a, b))
    print("loss deviation per iteration:", ["%.2e" % dev(a, b) for a, b in zip(h_g, h_e)])
    assert dev(h_g[0], h_e[0]) < 1e-3, (h_g[0], h_e[0])            # before any update: only the atomics' rounding differs
    for k in range(1, 5):              # two fp32 trajectories separate by ~1e-2 per three iterations (test_train_sanity);
        assert dev(h_g[k], h_e[k]) < (3e-2 if k < 3 else 1e-1), (k, h_g[k], h_e[k])     # the updates are checked below
    # the five updates themselves: same length and direction per network as the eager run's (a skipped, doubled or stale
    # generator-side step -- it is replayed from a descriptor table -- would show here, not in the losses)
    for name, s0, a, b in zip(("generator", "discriminator", "kp_detector"), start, p_g, p_e):
        da, db = a - s0, b - s0
        cos = float((da * db).sum() / (da.norm() * db.norm()))
        ratio = float(da.norm() / db.norm())
        print("%s: update cosine %.4f, length ratio %.4f" % (name, cos, ratio))
        assert cos > 0.85 and 0.9 < ratio < 1.1, (name, cos, ratio)
    tdist.destroy_process_group()
    print("DIST-GRAPH-OK")


if __name__ == "__main__":
    main()
